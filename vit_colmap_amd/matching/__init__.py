from .hip_matcher import (knn_top2, match_pairs, mutual_ratio, prepare_descriptors, theta_table,
                          exhaustive_pairs)

__all__ = ["knn_top2", "match_pairs", "mutual_ratio", "prepare_descriptors", "theta_table",
           "exhaustive_pairs"]
