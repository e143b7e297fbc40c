"""``SolverBase`` -- the solver plugin surface the driver talks to (reference: src/solver/base.py:54-378).

Kept from the reference: the constructor signature, ``preprocess(events) -> (events, time_period)``,
``estimate(events, *args, **kwargs) -> np.ndarray [2, H, W]`` and the owned helpers ``orig_imager`` /
``crop_imager`` / ``orig_warper`` / ``crop_warper`` (always ``normalize_t=True``, :98-100).  Visualisation,
flow-error bookkeeping and the event-filter zoo of the reference are outside the accelerated path
 -- only the CROP step of the filter pipeline (always prepended by the reference's
``EventFilter``, src/utils/event_filters.py:182-202) is kept because it defines which events enter the window.
"""
from __future__ import annotations

import logging
from typing import Optional, Tuple

import numpy as np
import torch

from .. import costs, event_image_converter, warp
from .._staging import to_gpu

logger = logging.getLogger(__name__)


class SolverBase(object):
    """Args (same positions/names as the reference):
        orig_image_shape (tuple) ... (H, W) of the sensor.
        crop_image_shape (tuple) ... (H, W) of the region of interest.
        calibration_parameter (dict | None) ... unused by this path, stored.
        solver_config (dict) ... the ``solver`` section of the YAML.
        visualize_module ... optional object, stored as ``visualizer``.
    """

    def __init__(self, orig_image_shape: tuple, crop_image_shape: tuple, calibration_parameter: Optional[dict] = None,
                 solver_config: Optional[dict] = None, visualize_module=None):
        self.orig_image_shape = tuple(orig_image_shape)
        self.crop_image_shape = tuple(crop_image_shape)
        self.calib_param = calibration_parameter
        self.slv_config = dict(solver_config or {})
        self.visualizer = visualize_module
        self.sequential_video_list: list = []
        self.evaluation_text_list: list = []
        self.pad = int(self.slv_config.get("outer_padding", 0))
        self.orig_imager = event_image_converter.EventImageConverter(self.orig_image_shape, outer_padding=self.pad)
        self.crop_imager = event_image_converter.EventImageConverter(self.crop_image_shape, outer_padding=self.pad)
        self.orig_warper = warp.Warp(self.orig_image_shape, normalize_t=True, calib_param=calibration_parameter)
        self.crop_warper = warp.Warp(self.crop_image_shape, normalize_t=True, calib_param=calibration_parameter)
        self.warp_direction = self.slv_config.get("warp_direction", "first")
        self.motion_model = self.slv_config.get("motion_model", "dense-flow")
        self.roi = self._roi_from_config(self.slv_config)
        self.previous_best = None

    @staticmethod
    def _roi_from_config(cfg: dict) -> Optional[Tuple[int, int, int, int]]:
        """(xmin, xmax, ymin, ymax) of the CROP filter; x = rows, y = columns
        (keys propagated by src/utils/config_utils.py:42-88 into solver.filter.parameters)."""
        p = (cfg.get("filter") or {}).get("parameters") or {}
        keys = ("xmin", "xmax", "ymin", "ymax")
        return tuple(int(p[k]) for k in keys) if all(k in p for k in keys) else None

    def preprocess(self, events) -> tuple:
        """CROP the window to the ROI and report its time period (src/solver/base.py:123-139)."""
        ev = to_gpu(events)
        if self.roi is not None:
            x0, x1, y0, y1 = self.roi
            keep = (ev[:, 0] >= x0) & (ev[:, 0] < x1) & (ev[:, 1] >= y0) & (ev[:, 1] < y1)
            ev = ev[keep]
        period = float((ev[:, 2].max() - ev[:, 2].min()).item()) if ev.shape[0] else 0.0
        if isinstance(events, np.ndarray):
            return ev.cpu().numpy(), period
        return (ev if events.is_cuda else ev.cpu()), period

    def estimate(self, events, *args, **kwargs) -> np.ndarray:
        raise NotImplementedError

    def set_previous_frame_best_estimation(self, previous_best):
        self.previous_best = previous_best  # warm start hook (src/solver/base.py:355-361)
