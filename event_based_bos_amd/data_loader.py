"""Event ingest for the hot path: a raw-column event store.

The reference reads CCS recordings from HDF5 (``raw_events/{x: int16, y: int16, t: int32 us, p: bool}``,
src/data_loader/ccs.py:57-66) and hands every window to the solver as a float64 ``[n, 4]`` array
``(row = y, col = x, t / 1e6, p)`` (:289-297).  ``h5py`` is not part of this image, so the store keeps the same
four columns in an uncompressed ``.npz`` (keys ``raw_events_x/y/t/p``; an ``.hdf5`` path is read directly where ``h5py`` exists,
``_hdf5.py``) -- and, more to the point, it can hand a
window to the GPU *as raw columns* (9 B/event instead of 32 B/event over PCIe), where ``EventPlan.build_raw``
expands it with the same fp64 time arithmetic.

    store = RawEventStore("recording.npz")
    events = store.load_event(i0, i1)                    # reference format, for any reference-style caller
    plan = store.plan(i0, i1, (720, 1280), "first")      # fast path: raw columns -> device -> EventPlan

Index helpers follow the reference loader: ``index_to_time`` (:319-330), ``time_to_index`` = searchsorted - 1 (:343-356).
"""
from __future__ import annotations

import logging
from typing import Dict, Tuple, Union

import numpy as np
import torch

from . import _hip
from .event_plan import EventPlan

logger = logging.getLogger(__name__)

COLUMNS = (("x", np.int16), ("y", np.int16), ("t", np.int32), ("p", np.bool_))


class RawEventStore(object):
    NAME = "RAW_COLUMNS"
    TICKS_PER_SECOND = 1e6  # microsecond timestamps, src/data_loader/ccs.py:295

    def __init__(self, source: Union[str, Dict[str, np.ndarray]]):
        if isinstance(source, dict):
            data = source
        elif str(source).lower().endswith((".hdf5", ".h5")):   # the reference's own recordings, read as its h5py_loader does (needs h5py)
            from ._hdf5 import read_raw_events

            data = read_raw_events(str(source), wide_time=True)
        else:
            with np.load(source) as f:
                data = {k: f["raw_events_" + k] for k, _ in COLUMNS}
        self.event_data = {}
        for k, dt in COLUMNS:
            col = np.asarray(data[k])
            if k == "t" and col.dtype == np.int64:
                pass  # recordings beyond 2^31 us keep 64-bit ticks (the reference only warns, :60-62)
            else:
                col = col.astype(dt, copy=False)
            self.event_data[k] = np.ascontiguousarray(col)
        n = len(self.event_data["t"])
        if any(len(v) != n or v.ndim != 1 for v in self.event_data.values()):
            raise ValueError("raw event columns must be 1-D and of equal length")
        self._time_cache = None
        self._pinned = None

    @staticmethod
    def save(path: str, x, y, t, p) -> None:
        """Write the four raw columns (sensor x = column, sensor y = row, t in microseconds, polarity)."""
        t = np.asarray(t)
        np.savez(path, raw_events_x=np.asarray(x, dtype=np.int16), raw_events_y=np.asarray(y, dtype=np.int16),
                 raw_events_t=t.astype(np.int64 if t.size and np.abs(t).max() > np.iinfo(np.int32).max else np.int32),
                 raw_events_p=np.asarray(p, dtype=np.bool_))

    def __len__(self) -> int:
        return len(self.event_data["x"])

    # (what set_sequence leaves on the reference's loader, src/data_loader/ccs.py:213-217)
    @property
    def min_ts(self) -> float:
        return self.event_data["t"].min() / self.TICKS_PER_SECOND

    @property
    def max_ts(self) -> float:
        return self.event_data["t"].max() / self.TICKS_PER_SECOND

    @property
    def data_duration(self) -> float:
        return self.max_ts - self.min_ts

    # ------------------------------------------------------------------ reference-format window
    def _check(self, start_index: int, end_index: int) -> None:
        if end_index > len(self):
            e = f"Specified {start_index} to {end_index} index, but there are only {len(self)} events."
            logger.error(e)
            raise IndexError(e)
        if end_index - start_index <= 0 or start_index >= len(self):
            e = f"Specified {start_index} to {end_index} index, but no events."
            logger.error(e)
            raise IndexError(e)

    def load_event(self, start_index: int, end_index: int, *args, **kwargs) -> np.ndarray:
        """float64 [n, 4] = (row, col, t in seconds, p), as src/data_loader/ccs.py:247-297."""
        self._check(start_index, end_index)
        n = end_index - start_index
        events = np.zeros((n, 4), dtype=np.float64)
        sl = slice(start_index, end_index)
        events[:, 0] = self.event_data["y"][sl]
        events[:, 1] = self.event_data["x"][sl]
        events[:, 2] = self.event_data["t"][sl] / self.TICKS_PER_SECOND
        events[:, 3] = self.event_data["p"][sl]
        return events

    # ------------------------------------------------------------------ raw window on the device
    def pin(self) -> "RawEventStore":
        """Page-lock the four columns once (9 B/event of host memory): every later window upload is then a plain
        asynchronous DMA from the recording itself -- no per-window staging buffer, no pinned allocation (tens of
        milliseconds each) on the ingest path."""
        if not self._pinned:
            _hip.require_gpu()
            self._pinned = {k: torch.from_numpy(self.event_data[k].view(np.uint8) if k == "p" else self.event_data[k]).pin_memory()
                            for k in ("x", "y", "t", "p")}
        return self

    def load_raw(self, start_index: int, end_index: int, device="cuda") -> Tuple[torch.Tensor, ...]:
        """(col int16, row int16, t int32|int64, pol uint8) of the window on ``device``: asynchronous copies on the
        current stream straight from the page-locked columns (9 B/event)."""
        self._check(start_index, end_index)
        dev = torch.device(device)
        if self._pinned is None:
            try:
                self.pin()
            except RuntimeError as err:  # recording too large to page-lock whole: stage window by window instead
                logger.warning(f"could not page-lock the recording ({err}); staging every window separately")
                self._pinned = False
        if self._pinned:
            return tuple(self._pinned[k][start_index:end_index].to(dev, non_blocking=True) for k in ("x", "y", "t", "p"))
        sl = slice(start_index, end_index)
        cols = (torch.from_numpy(self.event_data[k][sl].view(np.uint8) if k == "p" else self.event_data[k][sl]) for k in "xytp")
        return tuple(c.pin_memory().to(dev, non_blocking=True) for c in cols)

    def plan(self, start_index: int, end_index: int, image_size: Tuple[int, int], direction="first",
             normalize_t: bool = True, tile="auto", device="cuda", deferred: bool = False, emit: str = "full") -> EventPlan:
        col, row, t, pol = self.load_raw(start_index, end_index, device)
        return EventPlan.build_raw(col, row, t, pol, image_size, direction, normalize_t, tile, self.TICKS_PER_SECOND,
                                   deferred=deferred, emit=emit)

    # ------------------------------------------------------------------ index <-> time
    def _times(self) -> np.ndarray:
        if self._time_cache is None:
            self._time_cache = self.event_data["t"] / self.TICKS_PER_SECOND
        return self._time_cache

    def index_to_time(self, index: int) -> float:
        return self._times()[index]

    def time_to_index(self, time: float) -> int:
        return int(np.searchsorted(self._times(), time)) - 1


collections = {RawEventStore.NAME: RawEventStore}
