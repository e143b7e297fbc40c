"""The event-ingest restatement (event_based_bos_amd.data_loader.RawEventStore, oracle.events_from_raw_columns) against the REFERENCE's
co-capture loader run on a synthetic HDF5 recording with the real h5py (tests/golden/make_golden_loader.py, generated under the
container's Anaconda interpreter; src/data_loader/ccs.py:48-66, 88-91, 133-136, 209-217, 247-297, 319-356).  CPU only."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ebos_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    with np.load(os.path.join(HERE, "golden", "golden_loader.npz")) as f:
        return {k: f[k] for k in f.files}


def _store(gold, as_held: bool):
    import event_based_bos_amd as ebos

    src = "held_" if as_held else "file_"   # the columns as the reference holds them / as they sit in the file (uint16, int64, uint8)
    return ebos.data_loader.RawEventStore({k: gold[src + k] for k in "xytp"})


@pytest.mark.parametrize("as_held", [True, False])
def test_store_equals_the_reference_loader(gold, as_held):
    s = _store(gold, as_held)
    n = int(gold["len"])
    assert len(s) == n
    for k in "xyp":   # the casts of h5py_loader (:61-66); t may stay 64-bit (same values: the reference wraps beyond 2^31 us, we do not)
        assert s.event_data[k].dtype == np.dtype(str(gold[f"held_{k}_dtype"])) and np.array_equal(s.event_data[k], gold["held_" + k])
    assert np.array_equal(s.event_data["t"], gold["held_t"])
    assert (s.min_ts, s.max_ts, s.data_duration) == (float(gold["min_ts"]), float(gold["max_ts"]), float(gold["duration"]))
    for i, (a, b) in enumerate(gold["windows"]):
        ev = s.load_event(int(a), int(b))
        assert ev.dtype == np.float64 and np.array_equal(ev, gold[f"window{i}"]), (a, b)     # bit for bit
        assert np.array_equal(O.events_from_raw_columns(*(gold["held_" + k] for k in "xytp"), int(a), int(b)), gold[f"window{i}"])
    for (a, b), raises in zip(gold["bad_windows"], gold["bad_windows_raise_index_error"]):
        assert raises == 1
        with pytest.raises(IndexError):
            s.load_event(int(a), int(b))
    for i, want in zip(gold["index_to_time_in"], gold["index_to_time_out"]):
        assert s.index_to_time(int(i)) == want
    for tm, want in zip(gold["time_to_index_in"], gold["time_to_index_out"]):
        assert s.time_to_index(float(tm)) == int(want), tm


def _python_with_h5py():
    for exe in (sys.executable, "/opt/conda/bin/python3.9", shutil.which("python3.9") or ""):
        if exe and os.path.exists(exe):
            r = subprocess.run([exe, "-c", "import h5py, numpy"], capture_output=True)
            if r.returncode == 0:
                return exe
    return None


def test_hdf5_recording_through_the_converter(gold, tmp_path):
    """The committed HDF5 recording (written by h5py for the generator's run of the reference) -> tools/hdf5_to_npz.py (the package's
    torch-free reader, under whichever interpreter of this machine has h5py) -> RawEventStore: the columns the reference's h5py_loader
    held, and its windows."""
    import event_based_bos_amd as ebos

    assert int(gold["hdf5_reader_equals_h5py_loader"]) == 1   # (checked by the generator: _hdf5.read_raw_events == h5py_loader)
    exe = _python_with_h5py()
    if exe is None:
        pytest.skip("no interpreter with h5py on this machine (the GPU image's python has none)")
    out = str(tmp_path / "rec.npz")
    r = subprocess.run([exe, os.path.join(ROOT, "tools", "hdf5_to_npz.py"), os.path.join(HERE, "golden", "loader_recording.hdf5"), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    s = ebos.data_loader.RawEventStore(out)
    for k in "xytp":
        assert np.array_equal(s.event_data[k], gold["held_" + k]), k
    a, b = (int(v) for v in gold["windows"][1])
    assert np.array_equal(s.load_event(a, b), gold["window1"])
