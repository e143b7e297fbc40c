"""Solver registry (reference: src/solver/__init__.py:11-16): ``collections[name](orig_image_shape,
crop_image_shape, calibration_parameter=..., solver_config=..., visualize_module=...)``."""
from .base import SolverBase
from .contrast_maximization import (ContrastMaximization, ContrastMaximizationMixin, make_solver_class, patch_grid_shape,
                                    register_into)
from .window_pipeline import WindowPipeline

collections = {
    "contrast_maximization": ContrastMaximization,
    "cmax": ContrastMaximization,
}
