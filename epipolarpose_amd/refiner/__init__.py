"""Refinement MLP (BASELINE config 5 "post-lift") -- mirror of the reference's ``refiner`` package (refiner/model.py, main.py,
utils.py, data.py) on the MI355X kernels of ``libepipolar_hip``."""
from .model import LinearModelPG, LinearPG, get_model, weight_init  # noqa: F401
