"""The four image-domain costs shipped by the reference (regularisers / data terms used next to a
contrast cost inside ``HybridCost``).  They touch one [2, H, W] flow or one image per call -- a few
MB -- and stay plain tensor expressions on whatever device their input lives on.

reference: src/costs/diff_norm.py:27-57, flow_norm.py:45-56, flow_norm_pxy.py:12-43, image_gradient.py:60-75
"""
import logging
from typing import Union

import numpy as np
import torch

from .base import CostBase

logger = logging.getLogger(__name__)


def _unsupported(value):
    e = f"Unsupported input type. {type(value)}."
    logger.error(e)
    return NotImplementedError(e)


class DifferenceNorm(CostBase):
    """1-norm of prediction - measurement.  For 2-D inputs ``norm(ord=1)`` is the MATRIX 1-norm (largest
    absolute column sum), exactly as the reference evaluates it; the ``weights`` key must be present
    although it is unused (src/costs/diff_norm.py:38,52)."""

    name = "diff_norm"
    required_keys = ["prediction", "measurement"]

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        prediction, measurement = arg["prediction"], arg["measurement"]
        weights = arg["weights"]
        if isinstance(prediction, torch.Tensor):
            return self.calculate_torch(prediction, measurement, weights)
        if isinstance(prediction, np.ndarray):
            return self.calculate_numpy(prediction, measurement, weights)
        raise _unsupported(prediction)

    def calculate_torch(self, prediction, measurement, weights) -> torch.Tensor:
        loss = torch.linalg.norm(prediction - measurement, ord=1)
        if self.direction == "minimize":
            return loss
        logger.warning("The loss is specified as maximize direction")
        return -loss

    def calculate_numpy(self, prediction, measurement, weights) -> float:
        return np.linalg.norm(prediction - measurement, ord=1)  # same value for every direction (:66-67)


class FlowNorm(CostBase):
    """Mean per-pixel L2 magnitude of a [2, H, W] flow."""

    name = "flow_norm"
    required_keys = ["flow"]

    def __init__(self, direction="minimize", store_history: bool = False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    def _dispatch(self, value):
        if isinstance(value, torch.Tensor):
            return self.calculate_torch(value)
        if isinstance(value, np.ndarray):
            return self.calculate_numpy(value)
        raise _unsupported(value)

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        return self._dispatch(arg["flow"])

    def calculate_torch(self, flow: torch.Tensor) -> torch.Tensor:
        loss = torch.linalg.norm(flow, dim=0).mean()
        if self.direction == "minimize":
            return loss
        logger.warning("The loss is specified as maximize direction")
        return -loss

    def calculate_numpy(self, flow: np.ndarray) -> float:
        return np.linalg.norm(flow, axis=0).mean()


class FlowNormPxy(FlowNorm):
    """``FlowNorm`` evaluated on the ``pxy`` key."""

    name = "flow_norm_pxy"
    required_keys = ["pxy"]

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        return self._dispatch(arg["pxy"])


class ImageGradient(CostBase):
    """Weighted total variation of the flow: mean(|d flow/d row * w| + |d flow/d col * w|) with
    ``torch.gradient`` (central differences).  Tensor input only, like the reference (its numpy branch
    calls an undefined method)."""

    name = "image_gradient"
    required_keys = ["flow", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 visualize_intermediate=False, *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        flow, omit_boundary, weights = arg["flow"], arg["omit_boundary"], arg["weights"]
        if isinstance(flow, torch.Tensor):
            return self.calculate_torch(flow, weights, omit_boundary)
        if isinstance(flow, np.ndarray):
            return self.calculate_numpy(flow, weights, omit_boundary)
        raise _unsupported(flow)

    def calculate_torch(self, flow: torch.Tensor, weights, omit_boundary: bool) -> torch.Tensor:
        d_row = torch.gradient(flow, dim=1)[0] * weights
        d_col = torch.gradient(flow, dim=2)[0] * weights
        loss = torch.mean(torch.abs(d_row) + torch.abs(d_col))
        if self.direction == "minimize":
            return loss
        logger.warning("The loss is specified as maximize direction")
        return -loss
