"""Contrast costs of the contrast-maximisation loop: ``image_variance`` and ``gradient_magnitude``.

The release of the reference references these but does not ship them (``costs.NormalizedImageVariance``
at src/solver/base.py:337 is undefined).  They are defined on the reference's own
primitives -- ``torch.var`` (unbiased) and ``SobelTorch(ksize=3)/8`` with replicate padding
(src/utils/stat_utils.py:48-139) -- behind the reference's ``CostBase`` plugin API, with the argument
keys the reference's dead call site documents: ``iwe`` and ``omit_boundary`` (src/solver/base.py:337-339).

Both are evaluated by HIP kernels (``ebos_image_variance_*`` / ``ebos_gradient_magnitude_*``, fp64
partial sums) and are differentiable; numpy arrays / CPU tensors are staged through the GPU.
Sign convention of the shipped costs (src/costs/diff_norm.py:54-57): ``direction="minimize"`` returns
the quantity to minimise, i.e. the NEGATIVE contrast; "maximize"/"natural" return the contrast.
"""
import logging
from typing import Union

import numpy as np
import torch

from .. import fusion, ops
from .._staging import NUMPY, back, kind_of, to_gpu
from .base import CostBase

logger = logging.getLogger(__name__)


class _ContrastCost(CostBase):
    required_keys = ["iwe", "omit_boundary"]

    def __init__(self, direction="minimize", store_history: bool = False, cuda_available=False, precision="32",
                 *args, **kwargs):
        super().__init__(direction=direction, store_history=store_history)

    def _contrast(self, iwe_gpu: torch.Tensor, omit_boundary: bool) -> torch.Tensor:
        raise NotImplementedError

    def _evaluate(self, arg: dict):
        iwe, omit_boundary = arg["iwe"], arg["omit_boundary"]
        if type(iwe) is fusion.LazyIwe:  # the idiom's third step on an image nobody has read: the objective's one native call
            loss = fusion.fused_variance(iwe, bool(omit_boundary), self.name, -1.0 if self.direction == "minimize" else 1.0)
            if loss is not None:
                return loss
        if not isinstance(iwe, (torch.Tensor, np.ndarray)):
            e = f"Unsupported input type. {type(iwe)}."
            logger.error(e)
            raise NotImplementedError(e)
        kind = kind_of(iwe)
        iwe_gpu = to_gpu(iwe)
        # the image of the fused idiom, untouched (or never computed): the objective's one native call, the direction's sign applied
        # inside its kernels
        loss = fusion.fused_variance(iwe_gpu, bool(omit_boundary), self.name, -1.0 if self.direction == "minimize" else 1.0)
        if loss is None:
            value = self._contrast(iwe_gpu, bool(omit_boundary))
            loss = -value if self.direction == "minimize" else value
        if kind == NUMPY:
            return float(loss.item()) if loss.dim() == 0 else loss.detach().cpu().numpy()
        return back(loss, kind)


class ImageVariance(_ContrastCost):
    """Variance of the IWE (unbiased, ``torch.var``); ``omit_boundary`` drops the outermost pixel ring.
    ``iwe`` may be one image [H, W] or a stack [K, H, W] (one value per image)."""

    name = "image_variance"

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        return self._evaluate(arg)

    def _contrast(self, iwe_gpu, omit_boundary):
        return ops.image_variance(iwe_gpu, omit_boundary)


class GradientMagnitude(_ContrastCost):
    """mean(gx^2 + gy^2) of the IWE with (gx, gy) = Sobel 3x3 / 8, replicate padding."""

    name = "gradient_magnitude"

    @CostBase.register_history
    @CostBase.catch_key_error
    def calculate(self, arg: dict) -> Union[float, torch.Tensor]:
        return self._evaluate(arg)

    def _contrast(self, iwe_gpu, omit_boundary):
        return ops.gradient_magnitude(iwe_gpu, omit_boundary)
