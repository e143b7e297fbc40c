"""HDF5 recordings of the reference's co-capture loader, read the way it reads them (torch-free on purpose: this file also runs on
its own under an interpreter that has ``h5py`` but no torch -- tests/golden/make_golden_loader.py checks it against the reference's
``h5py_loader`` that way).

Layout (src/data_loader/ccs.py:48-66): group ``raw_events`` with datasets ``x`` (sensor column), ``y`` (sensor row), ``t``
(microseconds) and ``p`` (polarity), cast on load to int16 / int16 / int32 / bool.  The reference only WARNS when a recording is
longer than int32 microseconds can count (:57-59) and then wraps; ``wide_time=True`` keeps such a recording's ticks as int64 instead
(``RawEventStore``'s default: a wrapped clock is of no use to anyone).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

GROUP = "raw_events"


def read_raw_events(path: str, wide_time: bool = False) -> Dict[str, np.ndarray]:
    try:
        import h5py
    except ImportError as err:  # the GPU image has no h5py: convert once with tools/hdf5_to_npz.py where it is installed
        raise ImportError(f"reading {path} needs h5py (not in this image); RawEventStore also takes the columns as a dict or an "
                          f".npz written by RawEventStore.save / tools/hdf5_to_npz.py") from err
    with h5py.File(path, "r") as f:
        g = f[GROUP]
        t = np.array(g["t"])
        fits = t.size == 0 or (int(t.max()) <= np.iinfo(np.int32).max and int(t.min()) >= np.iinfo(np.int32).min)
        return {
            "x": np.array(g["x"], dtype=np.int16),
            "y": np.array(g["y"], dtype=np.int16),
            "t": t.astype(np.int64) if (wide_time and not fits) else np.array(g["t"], dtype=np.int32),
            "p": np.array(g["p"], dtype=bool),
        }
