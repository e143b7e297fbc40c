"""``Warp`` -- per-event warp to a reference time, computed by hand-written gfx950 kernels.

Drop-in for the reference's ``src/warp.py`` (class ``Warp``, :55-383 under /root/reference): same
constructor, methods, argument meaning, return shapes (incl. the ``.squeeze()`` quirks) and
exceptions.  numpy arrays, CPU tensors and GPU tensors are all accepted; whatever comes in is
warped on the GPU through libebos_hip.so and returned in the caller's container type.  There is
no CPU compute path: without the library or a GPU every call raises ``HipUnavailableError``.

Semantics kept from the reference:
  * event = (x, y, t, p), x = row, y = column;  dense flow [(b,) 2, H, W], channel 0 = rows
  * dense-flow:    x' = x - dt * F0[trunc(x), trunc(y)],  y' = y - dt * F1[...]        (:330-342)
  * 2d-translation / rigid-optical-flow:  x' = x + dt * th0,  y' = y + dt * th1       (:364-383)
  * t' = dt = t - t_ref, divided by (max dt - min dt) when ``normalize_t``            (:283-287)
  * ``direction`` must be a python ``float`` or one of first/middle/last/random/before/after (:245-262)
On equal dtypes the warped coordinates are bit-identical to the reference (same operation order,
no FMA contraction).
"""
from __future__ import annotations

import logging
import os
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import _hip, fusion, ops
from ._staging import CPU, GPU, NUMPY, back, kind_of, to_gpu
from .event_plan import parse_direction
from .types import FLOAT_TORCH, NUMPY_TORCH, is_numpy, is_torch

logger = logging.getLogger(__name__)

TRANSLATION_MODELS = ("2d-translation", "rigid-optical-flow")


class MotionModelKeyError(Exception):
    """Unknown motion model (logged, then raised -- src/warp.py:18-22)."""

    def __init__(self, message):
        logger.error(message)
        super().__init__(message)


class FeatureCalculatorMock:
    """The release of the reference ships only a stub for warp features (src/warp.py:25-50); the
    same five keys with ``value=None`` are returned here."""

    _KEYS = (("determinant", True), ("trace", True), ("divergence", True), ("straint", True), ("absement", False))

    def __init__(self, *args, **kwargs):
        pass

    def skip(self) -> dict:
        return {k: {"per_event": per_event, "value": None} for k, per_event in self._KEYS}

    def calculate_feature(self, *args, skip: bool = False, **kwargs) -> dict:
        if not skip:
            logger.warning("Feature calculation is disabled in this source code!")
        return self.skip()


class Warp(object):
    """Warp functions with different motion models.

    Args:
        image_size (tuple[int, int]) ... (H, W).  ``image_size[1]`` is the row stride of the flow lookup.
        calculate_feature (bool) ... kept for API parity; features are stubbed like in the reference.
        normalize_t (bool) ... normalise dt to the window period.
        calib_param ... stored, unused (as in the reference).
        strict (bool | None) ... raise IndexError when an event's source pixel is outside the flow field
            (torch.gather raises in the reference).  The check reads one counter back from the GPU, so
            by default it is immediate for numpy / CPU inputs (which synchronise anyway) and DEFERRED for GPU
            tensors: the count stays on the device, comes back asynchronously and a later warp_event call (or
            ``check_out_of_range()``) raises.  strict=True / EBOS_STRICT=1: immediate everywhere;
            strict=False / EBOS_STRICT=0: never checked.
    """

    def __init__(self, image_size: tuple, calculate_feature: bool = False, normalize_t: bool = False,
                 calib_param: Optional[np.ndarray] = None, strict: Optional[bool] = None):
        self.update_property(image_size, calculate_feature, normalize_t, calib_param)
        self.feature_2dof = FeatureCalculatorMock()
        self.feature_dense = FeatureCalculatorMock()
        self.strict = strict
        # deferred out-of-range check of GPU-tensor calls: device counter per device, (page-locked copy, event, device) in flight
        self._oob_dev, self._oob_pending, self._oob_calls = {}, [], 0
        self._flow_snapshots = fusion.FlowSnapshots()  # copies of the flow at warp_event time, for deferred results (fusion.py)

    def update_property(self, image_size: Optional[tuple] = None, calculate_feature: Optional[bool] = None,
                        normalize_t: Optional[bool] = None, calib_param: Optional[np.ndarray] = None):
        if image_size is not None:
            self.image_size = image_size
        if calculate_feature is not None:
            self.calculate_feature = calculate_feature
        if normalize_t is not None:
            self.normalize_t = normalize_t
        if calib_param is not None:
            logger.info("Set camera matrix K.")
            self.calib_param = calib_param

    # ------------------------------------------------------------------ motion-model helpers (:95-190)
    def get_key_names(self, motion_model: str) -> list:
        if motion_model == "dense-flow":
            logger.warning(f"Assume only rigid transformation {motion_model = }")
            return ["trans_x", "trans_y"]
        if motion_model in TRANSLATION_MODELS:
            return ["trans_x", "trans_y"]
        if motion_model == "scaler":
            return ["scaler"]
        raise MotionModelKeyError(f"{motion_model = } not supported")

    def get_motion_vector_size(self, motion_model: str) -> int:
        zero = {k: 0.0 for k in self.get_key_names(motion_model)}
        return len(self.motion_model_to_motion(motion_model, zero))

    def motion_model_to_motion(self, motion_model: str, params: dict) -> np.ndarray:
        if motion_model == "dense-flow":
            logger.warning(f"Assume only rigid transformation {motion_model = }")
            return self.get_flow_from_motion(np.array([params["trans_x"], params["trans_y"]]), "2d-translation")
        if motion_model in TRANSLATION_MODELS:
            return np.array([params["trans_x"], params["trans_y"]])
        if motion_model == "scaler":
            return np.array([params["scaler"]])
        raise MotionModelKeyError(f"{motion_model = } not supported")

    def motion_model_from_motion(self, motion: np.ndarray, motion_model: str) -> dict:
        if motion_model == "dense-flow":
            logger.warning(f"Assume only rigid transformation {motion_model = }")
            return {"trans_x": motion[0], "trans_y": motion[1]}
        if motion_model in TRANSLATION_MODELS:
            return {"trans_x": motion[0], "trans_y": motion[1]}
        if motion_model == "scaler":
            return {"scaler": motion[0]}
        raise MotionModelKeyError(f"{motion_model = } not supported")

    def get_flow_from_motion(self, motion: NUMPY_TORCH, motion_model: str) -> NUMPY_TORCH:
        """Dense flow [2, H, W] equivalent to ``motion``: one synthetic event per pixel at t = 1 (plus one
        at t = 0 that pins the reference time) is warped and the displacement negated (:167-190)."""
        H, W = int(self.image_size[0]), int(self.image_size[1])
        dev = motion.device if is_torch(motion) and motion.is_cuda else None
        m = to_gpu(motion, device=dev)
        rows = torch.arange(H, dtype=torch.float64, device=m.device).repeat_interleave(W)
        cols = torch.arange(W, dtype=torch.float64, device=m.device).repeat(H)
        ones = torch.ones_like(rows)
        grid = torch.stack([rows, cols, ones, ones], dim=1)
        events = torch.cat([torch.zeros((1, 4), dtype=torch.float64, device=m.device), grid], dim=0)
        warped, _ = self.warp_event(events, m.double(), motion_model)
        u = -(warped[1:, 0] - events[1:, 0]).reshape(H, W)[None]
        v = -(warped[1:, 1] - events[1:, 1]).reshape(H, W)[None]
        flow = torch.cat([u, v], dim=0)
        if is_torch(motion):
            return flow if motion.is_cuda else flow.cpu()
        return flow.cpu().numpy()

    # ------------------------------------------------------------------ warp (:193-228)
    def warp_event(self, events: NUMPY_TORCH, motion: NUMPY_TORCH, motion_model: str,
                   direction: Union[str, float] = "first",
                   flow_propagate_bin: Optional[int] = None) -> Tuple[NUMPY_TORCH, dict]:
        """Warp events [(b,) n, 4] with ``motion`` under ``motion_model``; returns (warped, feature dict)."""
        ref_mode, frac = parse_direction(direction)  # ValueError first, like calculate_reftime (:218)
        if motion_model == "dense-flow":
            if is_torch(events) and is_torch(motion) and fusion.lazy_eligible(events, motion, self.image_size) and \
                    not self._strict_for(GPU):
                # opt-in: the coordinates are computed when (if) something reads them; EventImageConverter does not (fusion.py)
                # (a device copy of the flow as it is NOW: a result read after optimizer.step() holds the old flow's values, as
                # the reference's eager tensor does, src/warp.py:330-342)
                slot = self._flow_snapshots.take(motion)
                prov = fusion.Provenance(events, events._version, motion, ref_mode, frac, direction, bool(self.normalize_t),
                                         (int(self.image_size[0]), int(self.image_size[1])), flow_snapshot=slot[0], flow_slot=slot)
                lazy = fusion.LazyWarped.make(events, motion, prov,
                                              lambda fl: self.warp_event_now(events, fl, ref_mode, frac, direction))
                fusion.FlowSnapshots.own(slot, lazy)
                return lazy, self.feature_dense.calculate_feature(skip=not self.calculate_feature)
            warped, feat = self._warp_dense(events, motion, ref_mode, frac, None)
            if is_torch(events) and is_torch(motion) and fusion.eligible(events, motion, self.image_size):
                # remember where these coordinates came from: EventImageConverter can then build the image with the
                # fused kernels instead of splatting the materialised coordinates (see fusion.py)
                fusion.tag(warped, fusion.Provenance(events, events._version, motion, ref_mode, frac, direction,
                                                     bool(self.normalize_t), (int(self.image_size[0]), int(self.image_size[1]))))
            return warped, feat
        if motion_model in TRANSLATION_MODELS:
            assert motion.shape[-1] == 2
            return self._warp_2dof(events, motion, ref_mode, frac, None, None)
        raise MotionModelKeyError(f"{motion_model = } not supported")

    def warp_event_now(self, events, motion, ref_mode, frac, direction) -> torch.Tensor:
        """The dense-flow warp of GPU tensors, computed and provenance-tagged (what ``warp_event`` returns unless it is lazy)."""
        warped, _ = self._warp_dense(events, motion, ref_mode, frac, None)
        if fusion.eligible(events, motion, self.image_size):
            fusion.tag(warped, fusion.Provenance(events, events._version, motion, ref_mode, frac, direction,
                                                 bool(self.normalize_t), (int(self.image_size[0]), int(self.image_size[1]))))
        return warped

    def _snapshot_oob(self, device) -> None:
        host = torch.empty(1, dtype=torch.int32, pin_memory=True)
        host.copy_(self._oob_dev[device], non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(device))
        self._oob_pending.append((host, done, device))
        del self._oob_pending[:-4]

    def _raise_pending_oob(self, wait: bool = False) -> None:
        keep, bad = [], 0
        for host, done, device in self._oob_pending:
            if wait:
                done.synchronize()
            if done.query():
                bad = max(bad, int(host.item()))  # (snapshots of one running counter)
            else:
                keep.append((host, done, device))
        self._oob_pending[:] = keep
        if bad:
            for c in self._oob_dev.values():
                c.zero_()
            self._oob_pending.clear()
            raise IndexError(f"{bad} event(s) of earlier warp_event call(s) have a source pixel outside the flow field "
                             f"(index out of range in gather, src/warp.py:334-336); they were left un-warped.  "
                             f"Warp(..., strict=True) or EBOS_STRICT=1 raises at the call itself, EBOS_STRICT=0 disables the check")

    def check_out_of_range(self) -> None:
        """Wait for the out-of-range count of the GPU-tensor calls made so far and raise ``IndexError`` if any event had a
        source pixel outside the flow field (the reference raises inside ``torch.gather``, src/warp.py:334-336; for GPU
        tensors the check is deferred so that a solver loop does not synchronise on it)."""
        for device in list(self._oob_dev):
            self._snapshot_oob(device)
        self._raise_pending_oob(wait=True)

    def _strict_for(self, kind: str) -> bool:
        if os.environ.get("EBOS_STRICT", "") == "1":
            return True
        return (kind != GPU) if self.strict is None else bool(self.strict)

    def _warp_dense(self, event, flow, ref_mode, frac, timebase) -> Tuple[NUMPY_TORCH, dict]:
        kind = kind_of(event)
        if kind == NUMPY:
            assert is_numpy(flow)
        else:
            assert is_torch(flow)
        ev = to_gpu(event)
        fl = to_gpu(flow, device=ev.device, dtype=ev.dtype)
        if ev.dim() == 2:
            ev, fl = ev[None], fl[None]
        assert ev.dim() == fl.dim() - 1 == 3  # same shape contract as :312
        strict = self._strict_for(kind)
        lazy = (not strict) and self.strict is None and os.environ.get("EBOS_STRICT", "") != "0" \
            and not torch.cuda.is_current_stream_capturing()
        oob = torch.zeros(1, dtype=torch.int32, device=ev.device) if strict else None
        if lazy:
            # GPU tensors: every call adds to one device-resident counter; every 16th call (and the first) sends it through
            # page-locked memory behind the kernel, and a later call that finds such a read-back complete looks at it -- the
            # loop never waits for the device.  check_out_of_range() is the blocking form.
            self._raise_pending_oob()
            oob = self._oob_dev.get(ev.device)
            if oob is None:
                oob = self._oob_dev[ev.device] = torch.zeros(1, dtype=torch.int32, device=ev.device)
        tmm = None
        if kind == GPU and timebase is None:
            # (min t, max t) of a window does not change between solver iterations: remembered on the caller's tensor object
            # together with its version counter (a reduction over all events + ~50 us of host work per warp otherwise)
            memo = getattr(event, "_ebos_time_range", None)
            if memo is not None and memo[0] == event._version and memo[1].dtype == ev.dtype:
                tmm = memo[1]
            else:
                tmm = ops.time_range(ev)
                try:
                    event._ebos_time_range = (event._version, tmm)
                except AttributeError:
                    pass
        warped = ops.warp_dense(ev, fl, ref_mode, frac, self.normalize_t, int(self.image_size[1]), oob, timebase, tmm)
        if strict and int(oob.item()) > 0:
            raise IndexError(f"{int(oob.item())} event(s) have a source pixel outside the flow field "
                             f"(index out of range in gather, src/warp.py:334-336)")
        if lazy:
            self._oob_calls += 1
            if self._oob_calls % 16 == 1:
                self._snapshot_oob(ev.device)
        feat = self.feature_dense.calculate_feature(skip=not self.calculate_feature)
        return back(warped.squeeze(), kind), feat

    def _warp_2dof(self, event, translation, ref_mode, frac, timebase, time_period) -> Tuple[NUMPY_TORCH, dict]:
        kind = kind_of(event)
        if kind == NUMPY:
            assert is_numpy(translation)
        else:
            assert is_torch(translation)
        ev = to_gpu(event)
        if ev.dim() == 1:
            ev = ev[None, :]
        th = to_gpu(translation, device=ev.device, dtype=ev.dtype)
        tp = None if time_period is None else to_gpu(time_period, device=ev.device, dtype=ev.dtype)
        warped = ops.warp_2dof(ev, th, ref_mode, frac, self.normalize_t, tp, timebase)
        feat = self.feature_2dof.calculate_feature(skip=not self.calculate_feature)
        return back(warped, kind), feat

    # ------------------------------------------------------------------ reference time / dt (:230-288)
    def calculate_reftime(self, events: NUMPY_TORCH, direction: Union[str, float] = "first") -> FLOAT_TORCH:
        """Reference time of the warp: a scalar (un-batched) or [b] (batched)."""
        ref_mode, frac = parse_direction(direction)
        kind = kind_of(events)
        ev = to_gpu(events)
        batched = ev.dim() == 3
        tmm = ops.time_range(ev if batched else ev.reshape(1, -1, 4))
        tmin, tmax = tmm[:, 0], tmm[:, 1]
        if ref_mode == _hip.REF_FIRST:
            ref = tmin
        elif ref_mode == _hip.REF_LAST:
            ref = tmax
        else:
            ref = tmin + (tmax - tmin) * frac
        ref = ref if batched else ref[0]
        out = back(ref, kind)
        return out[()] if kind == NUMPY and not batched else out

    def _timebase(self, ev3: torch.Tensor, reference_time, time_period=None) -> torch.Tensor:
        """(t_ref, period) rows for an explicit reference time; period = max dt - min dt (:285-286)."""
        b = ev3.shape[0]
        ref = to_gpu(reference_time, device=ev3.device, dtype=ev3.dtype).reshape(-1)
        ref = ref.expand(b) if ref.numel() == 1 else ref[:b]
        if time_period is None:
            tmm = ops.time_range(ev3)
            period = (tmm[:, 1] - ref) - (tmm[:, 0] - ref)
        else:
            period = to_gpu(time_period, device=ev3.device, dtype=ev3.dtype).reshape(-1).expand(b)
        return torch.stack([ref, period], dim=1)

    def calculate_dt(self, event: NUMPY_TORCH, reference_time: FLOAT_TORCH,
                     time_period: Optional[FLOAT_TORCH] = None) -> NUMPY_TORCH:
        """dt [(b,) n] = t - reference_time (/ period when ``normalize_t``).  Evaluated by the warp kernel
        with a zero motion: its third output column is exactly dt."""
        kind = kind_of(event)
        ev = to_gpu(event)
        batched = ev.dim() == 3
        ev3 = ev if batched else ev[None]
        tb = self._timebase(ev3, reference_time, time_period)
        zero = torch.zeros((ev3.shape[0], 2, 1, 1), dtype=ev3.dtype, device=ev3.device)
        dt = ops.warp_dense(ev3, zero, _hip.REF_TIMEBASE, 0.0, self.normalize_t, 1, None, tb)[..., 2]
        return back(dt if batched else dt[0], kind)

    # ------------------------------------------------------------------ explicit-reference-time entry points
    def warp_event_from_optical_flow(self, event: NUMPY_TORCH, flow: NUMPY_TORCH,
                                     reference_time: FLOAT_TORCH) -> Tuple[NUMPY_TORCH, dict]:
        """Dense-flow warp with an explicit reference time (:292-342)."""
        ev = to_gpu(event)
        tb = self._timebase(ev if ev.dim() == 3 else ev[None], reference_time)
        return self._warp_dense(event, flow, _hip.REF_TIMEBASE, 0.0, tb)

    def warp_event_2dof_xy(self, event: NUMPY_TORCH, translation: NUMPY_TORCH, reference_time: FLOAT_TORCH,
                           time_period: Optional[FLOAT_TORCH] = None) -> Tuple[NUMPY_TORCH, dict]:
        """2-DoF warp with an explicit reference time / period (:344-383).  Un-batched."""
        ev = to_gpu(event)
        ev2 = ev[None, :] if ev.dim() == 1 else ev
        tb = self._timebase(ev2[None], reference_time, time_period)
        return self._warp_2dof(event, translation, _hip.REF_TIMEBASE, 0.0, tb, None)
