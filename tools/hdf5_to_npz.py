#!/usr/bin/env python3
"""Convert a recording of the reference's loader (``<sequence>/prophesee_0/events.hdf5``, src/data_loader/ccs.py:48-66) into the
``.npz`` of raw columns that ``RawEventStore`` reads without h5py.  Runs under any interpreter with numpy + h5py (no torch needed):

    python tools/hdf5_to_npz.py events.hdf5 recording.npz
"""
import importlib.util
import os
import sys

import numpy as np

spec = importlib.util.spec_from_file_location("ebos_hdf5", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                         "event_based_bos_amd", "_hdf5.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    d = mod.read_raw_events(src, wide_time=True)
    np.savez(dst, raw_events_x=d["x"], raw_events_y=d["y"], raw_events_t=d["t"], raw_events_p=d["p"])
    print(f"{dst}: {len(d['t'])} events, t {d['t'].dtype}")
