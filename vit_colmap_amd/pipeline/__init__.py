from .run_pipeline import Pipeline, main

__all__ = ["Pipeline", "main"]
