#!/usr/bin/env python3
"""golden_loader.npz + loader_recording.hdf5: the REFERENCE's co-capture loader (src/data_loader/ccs.py) run on a synthetic recording.

The build container's default interpreter has no h5py; its Anaconda interpreter has (numpy 1.26, h5py 3.3, no torch):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_loader.py

What runs is the reference's own code on the reference's own file format with the real h5py: ``h5py_loader`` (:48-66),
``CcsDataLoader.set_sequence`` (:209-217: min / max time stamps), ``__len__`` / ``set_len_cache`` (:88-91, :133-136), ``load_event``
-> ``load_event_from_hdf`` (:247-297), ``index_to_time`` (:319-330), ``time_to_index`` (:343-356).  Packages that module chain
imports but this path never calls (cv2, torch, openpiv, ...) are absent from that interpreter and stand in as EMPTY modules -- none of
their attributes is touched (any call would raise).  The fixture pins ``event_based_bos_amd.data_loader.RawEventStore`` and the
oracle's ``events_from_raw_columns`` (tests/test_loader_golden.py); the script also checks ``event_based_bos_amd/_hdf5.py`` (the
HDF5 reader ``RawEventStore`` uses where h5py exists) against ``h5py_loader`` on the same file.
"""
import importlib
import importlib.util
import os
import shutil
import sys
import tempfile
import types
import warnings

import h5py
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True


class _Empty(types.ModuleType):
    """Stand-in for a package the loader path imports but never calls: attribute access yields an inert class (enough for
    ``from pkg import name`` and type annotations), calling anything real is impossible."""

    __path__ = []  # (a package: `import pkg.sub` resolves through sys.modules)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        if item[:1].islower():   # `pkg.sub.Name`: a lower-case attribute stands for a submodule (or a function nobody calls)
            sub = _Empty(self.__name__ + "." + item)
            sys.modules.setdefault(sub.__name__, sub)
            setattr(self, item, sub)
            return sub
        cls = type(item, (), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(RuntimeError(f"stub {item} called"))})
        setattr(self, item, cls)
        return cls


def import_loader():
    sys.path.insert(0, REF)
    stubbed = []
    for _ in range(200):  # import, stub whatever third-party package is missing, try again
        try:
            mod = importlib.import_module("src.data_loader.ccs")
            return mod, stubbed
        except ModuleNotFoundError as e:
            name = e.name
            if name is None or name.startswith("src"):
                raise
            parts = name.split(".")
            for i in range(1, len(parts) + 1):
                sub = ".".join(parts[:i])
                if sub not in sys.modules:
                    sys.modules[sub] = _Empty(sub)
                    stubbed.append(sub)
                    if i > 1:
                        setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], sys.modules[sub])
            for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
                del sys.modules[k]   # (a half-imported package would hide the next missing name)
    raise RuntimeError("could not import the reference loader")


def main():
    ccs, stubbed = import_loader()
    print("stand-ins for absent, unused packages:", sorted(set(s.split(".")[0] for s in stubbed)))
    H, W, n = 260, 346, 8000
    rs = np.random.RandomState(77)
    x = rs.randint(0, W, n).astype(np.uint16)                       # sensor column
    y = rs.randint(0, H, n).astype(np.uint16)                       # sensor row
    t = np.sort(rs.randint(1_000_000, 1_400_000, n)).astype(np.int64)   # microseconds, with ties
    t[:3] = t[0]                                                    # ... a run of equal stamps at the very start
    p = rs.randint(0, 2, n).astype(np.uint8)
    root = tempfile.mkdtemp()
    try:
        seq = os.path.join(root, "CCS", "seq0", "prophesee_0")
        os.makedirs(seq)
        path = os.path.join(seq, "events.hdf5")
        with h5py.File(path, "w") as f:
            g = f.create_group("raw_events")
            for k, v in (("x", x), ("y", y), ("t", t), ("p", p)):
                g.create_dataset(k, data=v, compression="gzip")
        loader = ccs.CcsDataLoader({"height": H, "width": W, "root": root, "dataset": "CCS"})
        loader.set_sequence("seq0")
        out = {"size": np.array([H, W]), "file_x": x, "file_y": y, "file_t": t, "file_p": p}
        for k in "xytp":   # what h5py_loader holds (the casts of :61-66)
            out["held_" + k] = loader.event_data[k]
            out["held_" + k + "_dtype"] = np.array(str(loader.event_data[k].dtype))
        out["len"] = np.array(len(loader))
        out["min_ts"], out["max_ts"], out["duration"] = np.array(loader.min_ts), np.array(loader.max_ts), np.array(loader.data_duration)
        windows = [(0, 100), (1234, 6000), (n - 500, n), (0, n), (4000, 4001)]
        out["windows"] = np.array(windows)
        for i, (a, b) in enumerate(windows):
            ev = loader.load_event(a, b)
            assert ev.dtype == np.float64 and ev.shape == (b - a, 4)
            out[f"window{i}"] = ev
        bad = [(0, n + 1), (n, n), (5, 5), (n + 3, n + 9)]
        out["bad_windows"] = np.array(bad)
        raised = []
        for a, b in bad:
            try:
                loader.load_event(a, b)
                raised.append(0)
            except IndexError:
                raised.append(1)
        out["bad_windows_raise_index_error"] = np.array(raised)
        idx = np.array([0, 1, 2, 3, 17, 4000, n - 1, -1])
        out["index_to_time_in"] = idx
        out["index_to_time_out"] = np.array([loader.index_to_time(int(i)) for i in idx])
        times = np.concatenate([[0.5, 1.0, float(t[0]) / 1e6, float(t[5]) / 1e6, float(t[4000]) / 1e6, float(t[-1]) / 1e6, 1.39999995, 2.0],
                                rs.uniform(0.99, 1.41, 24)])
        out["time_to_index_in"] = times
        out["time_to_index_out"] = np.array([int(loader.time_to_index(float(v))) for v in times])
        # the package's own HDF5 reader (torch-free file, loaded on its own) against the reference's h5py_loader on the same file
        spec = importlib.util.spec_from_file_location("ebos_hdf5", os.path.join(os.path.dirname(os.path.dirname(HERE)), "event_based_bos_amd", "_hdf5.py"))
        mine = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mine)
        theirs, ours = ccs.h5py_loader(path), mine.read_raw_events(path)
        for k in "xytp":
            assert ours[k].dtype == theirs[k].dtype and np.array_equal(ours[k], theirs[k]), k
        out["hdf5_reader_equals_h5py_loader"] = np.array(1)
        np.savez_compressed(os.path.join(HERE, "golden_loader.npz"), **out)
        shutil.copy(path, os.path.join(HERE, "loader_recording.hdf5"))
        print("golden_loader.npz:", len(out), "arrays;", "loader_recording.hdf5:", os.path.getsize(path), "bytes")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
