"""CPU oracle for the vit-colmap hot path — TEST INFRASTRUCTURE ONLY.

Restates the reference's algorithm for the path (select_oracle: vit_extractor.py:168-653 and
dummy_extractor.py:95-111; matcher_oracle: the COLMAP brute-force matcher semantics behind
run_pipeline.py:351-363; vit_oracle: the DINOv2 forward behind vit_extractor.py:135-146).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package `vit_colmap_amd` must never import from here.
"""
