"""Evaluation helpers next to the sampling path (the reference's src/Utils): tile-quality metrics on the GPU."""
