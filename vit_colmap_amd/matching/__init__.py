from .exhaustive import match_exhaustive
from .hip_matcher import (exhaustive_pairs, knn_top2, match_pairs, mutual_ratio, prepare_descriptors, theta_table)

__all__ = ["exhaustive_pairs", "knn_top2", "match_exhaustive", "match_pairs", "mutual_ratio",
           "prepare_descriptors", "theta_table"]
