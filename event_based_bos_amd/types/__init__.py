"""numpy/torch type helpers of the plugin surface (reference: src/types/__init__.py:8-47).

``nt_max`` / ``nt_min`` on GPU tensors stay torch reductions; the hot path never calls them (the
time range of an event batch comes from ``ebos_time_range_*``).
"""
from typing import Union

import numpy as np
import torch

from .flow_patch import FlowPatch, patch_bounds

NUMPY_TORCH = Union[np.ndarray, torch.Tensor]
FLOAT_TORCH = Union[float, torch.Tensor]

__all__ = ["NUMPY_TORCH", "FLOAT_TORCH", "FlowPatch", "patch_bounds", "is_torch", "is_numpy", "nt_max", "nt_min"]


def is_torch(obj) -> bool:
    return isinstance(obj, torch.Tensor)


def is_numpy(obj) -> bool:
    return isinstance(obj, np.ndarray)


def nt_max(array: NUMPY_TORCH, dim: int) -> NUMPY_TORCH:
    """Maximum along ``dim`` for either array type (None for anything else, like the reference)."""
    if is_numpy(array):
        return array.max(axis=dim)
    if is_torch(array):
        return torch.max(array, dim).values
    return None


def nt_min(array: NUMPY_TORCH, dim: int) -> NUMPY_TORCH:
    """Minimum along ``dim`` for either array type."""
    if is_numpy(array):
        return array.min(axis=dim)
    if is_torch(array):
        return torch.min(array, dim).values
    return None
