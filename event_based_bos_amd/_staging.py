"""Host <-> device staging for the plugin surface.

The reference's classes accept numpy arrays and torch tensors on any device.  Here every array is
computed on the GPU by the HIP kernels: numpy arrays and CPU tensors are uploaded, processed and the
result is handed back in the caller's container type (numpy stays numpy, CPU tensors stay CPU tensors
and keep their autograd link through ``.to()``).  There is no CPU compute path.
"""
from __future__ import annotations

from typing import Any, Optional

import numpy as np
import torch

from . import _hip

NUMPY, CPU, GPU = "numpy", "cpu", "gpu"


def default_device() -> torch.device:
    _hip.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def kind_of(x: Any) -> str:
    if isinstance(x, np.ndarray):
        return NUMPY
    if isinstance(x, torch.Tensor):
        return GPU if x.is_cuda else CPU
    raise TypeError(f"expected numpy.ndarray or torch.Tensor, got {type(x)}")


def to_gpu(x: Any, device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """numpy / CPU tensor / GPU tensor -> floating GPU tensor (float32 stays float32, everything else
    becomes float64 unless ``dtype`` is given)."""
    if isinstance(x, np.ndarray):
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        t = torch.from_numpy(np.ascontiguousarray(x))
    elif isinstance(x, torch.Tensor):
        t = x
        if not t.is_floating_point():
            t = t.double()
    else:
        t = torch.as_tensor(x, dtype=torch.float64)
    if not t.is_cuda:
        t = t.to(device or default_device())
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.dtype not in (torch.float32, torch.float64):
        t = t.double()
    return t


def back(t: torch.Tensor, kind: str):
    """GPU result -> the caller's container type."""
    if kind == NUMPY:
        return t.detach().cpu().numpy()
    if kind == CPU:
        return t.cpu()
    return t
