"""FlowPatch: one cell of the patch-flow grid (reference: src/types/flow_patch.py:9-91).

Field names, derived bounds and the dict-style access are the reference's; the patch grid itself
(``prepare_patch``, src/solver/patch_eklt.py:70-95) lives in ``event_based_bos_amd.solver``.
"""
import copy
import math
from dataclasses import dataclass
from typing import Any

import numpy as np


@dataclass
class FlowPatch:
    x: float            # centre, height (row) coordinate
    y: float            # centre, width (column) coordinate
    shape: tuple        # (height, width) of the patch
    u: float = 0.0      # flow, height component
    v: float = 0.0      # flow, width component

    def __getitem__(self, key):
        return getattr(self, key)

    # size
    h = property(lambda self: self.shape[0])
    w = property(lambda self: self.shape[1])
    # bounds: ceil on the low side, floor on the high side (src/types/flow_patch.py:33-47)
    x_min = property(lambda self: int(self.x - math.ceil(self.shape[0] / 2)))
    x_max = property(lambda self: int(self.x + math.floor(self.shape[0] / 2)))
    y_min = property(lambda self: int(self.y - math.ceil(self.shape[1] / 2)))
    y_max = property(lambda self: int(self.y + math.floor(self.shape[1] / 2)))
    xmin = property(lambda self: self.x_min)
    xmax = property(lambda self: self.x_max)
    ymin = property(lambda self: self.y_min)
    ymax = property(lambda self: self.y_max)
    position = property(lambda self: np.array([self.x, self.y]))
    flow = property(lambda self: np.array([self.u, self.v]))

    def update_flow(self, u: float, v: float) -> None:
        self.u, self.v = u, v

    def new_ones(self) -> np.ndarray:
        return np.ones(self.shape)

    def copy(self) -> Any:
        return copy.deepcopy(self)


def patch_bounds(image_size, patch_size, sliding_window):
    """(x_min [gh], x_max [gh], y_min [gw], y_max [gw]) int64 numpy: the crop window of every patch row / column of the
    grid -- centres ``arange(0, size - patch + slide, slide) + patch / 2`` (src/solver/patch_eklt.py:85-86) with the
    bounds of FlowPatch (src/types/flow_patch.py:33-47: ``int(c - ceil(h / 2))`` .. ``int(c + floor(h / 2))``, int()
    truncating toward zero)."""
    out = []
    for size, patch, slide in zip(image_size, patch_size, sliding_window):
        centre = np.arange(0, size - patch + slide, slide) + patch / 2
        out.append(np.trunc(centre - np.ceil(patch / 2)).astype(np.int64))
        out.append(np.trunc(centre + np.floor(patch / 2)).astype(np.int64))
    return tuple(out)
