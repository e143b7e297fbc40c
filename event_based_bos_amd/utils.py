"""Event-array helpers of the path (reference: src/utils/event_utils.py).

Only ``crop_event`` is on the path: it selects the events of a region of interest (the CROP filter,
src/utils/event_filters.py:182-202) and of a patch (src/solver/patch_eklt.py:118-124).  The per-patch use -- one pass
over the whole event array per patch, only to count -- is replaced by ``EventPlan.patch_event_counts``.

``propagate_config`` is the driver's config plumbing for this path (reference: src/utils/config_utils.py:42-88): the
solver reads its region of interest from keys that only exist after that propagation.
"""
from .types import NUMPY_TORCH

ROI_KEYS = ("xmin", "xmax", "ymin", "ymax")


def propagate_config(config: dict) -> dict:
    """In place, like the reference's (src/utils/config_utils.py:42-88): the region of interest of ``common_params``
    (x = rows, y = columns) is copied into ``data`` and ``solver.filter.parameters``; ``data.crop_height / crop_width``,
    the ``pad_{x0,x1,y0,y1}`` margins (into ``solver`` and every ``params_<frame method>`` section), ``solver.crop_*``,
    ``solver.params_opencv_flow / params_openpiv`` and ``evaluation.dt = common_params.n_frames`` are derived.  Missing
    optional sections (``solver.filter``, ``params_*``) are created instead of raising ``KeyError``.  Returns ``config``."""
    common, data = config["common_params"], config["data"]
    solver = config.get("solver")
    for key in ROI_KEYS:
        data[key] = common[key]
        if solver is not None:
            solver.setdefault("filter", {}).setdefault("parameters", {})[key] = common[key]
    data["crop_height"] = data["xmax"] - data["xmin"]
    data["crop_width"] = data["ymax"] - data["ymin"]
    pad = {"pad_x0": common["xmin"], "pad_x1": data["height"] - common["xmax"],
           "pad_y0": common["ymin"], "pad_y1": data["width"] - common["ymax"]}
    if solver is not None:
        for sect in ("params_opencv_flow", "params_openpiv"):
            if sect in config:
                solver[sect] = config[sect]
        solver.update(pad)
        solver["crop_height"], solver["crop_width"] = data["crop_height"], data["crop_width"]
    if "evaluation" in config and "n_frames" in common:
        config["evaluation"]["dt"] = common["n_frames"]
    for k in ("opencv_flow", "openpiv", "rife", "flowformer"):
        if f"params_{k}" in config:
            config[f"params_{k}"].update(pad)
        else:
            config[f"params_{k}"] = dict(pad)
    return config


def crop_event(events: NUMPY_TORCH, x0: int, x1: int, y0: int, y1: int) -> NUMPY_TORCH:
    """Events with ``x0 <= x < x1`` and ``y0 <= y < y1`` (x = row = events[..., 0], y = column = events[..., 1]);
    numpy arrays and torch tensors on any device, order kept.  src/utils/event_utils.py:109-129."""
    mask = (x0 <= events[..., 0]) & (events[..., 0] < x1) & (y0 <= events[..., 1]) & (events[..., 1] < y1)
    return events[mask]
