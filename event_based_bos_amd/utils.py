"""Event-array helpers of the path (reference: src/utils/event_utils.py).

Only ``crop_event`` is on the path: it selects the events of a region of interest (the CROP filter,
src/utils/event_filters.py:182-202) and of a patch (src/solver/patch_eklt.py:118-124).  The per-patch use -- one pass
over the whole event array per patch, only to count -- is replaced by ``EventPlan.patch_event_counts``.
"""
from .types import NUMPY_TORCH


def crop_event(events: NUMPY_TORCH, x0: int, x1: int, y0: int, y1: int) -> NUMPY_TORCH:
    """Events with ``x0 <= x < x1`` and ``y0 <= y < y1`` (x = row = events[..., 0], y = column = events[..., 1]);
    numpy arrays and torch tensors on any device, order kept.  src/utils/event_utils.py:109-129."""
    mask = (x0 <= events[..., 0]) & (events[..., 0] < x1) & (y0 <= events[..., 1]) & (events[..., 1] < y1)
    return events[mask]
