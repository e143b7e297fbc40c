"""Event ingest: the raw-column store against the oracle's restatement of the reference loader,
and -- on the GPU -- the raw-column plan against the plan of the float64 window (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import ebos_oracle as O

H, W = 60, 78


def test_store_round_trip_and_reference_format(tmp_path):
    from event_based_bos_amd.data_loader import RawEventStore, collections

    x, y, t, p = O.synth_raw_columns(5000, H, W, seed=3)
    path = str(tmp_path / "rec.npz")
    RawEventStore.save(path, x, y, t, p)
    store = collections["RAW_COLUMNS"](path)
    assert len(store) == 5000 and store.event_data["t"].dtype == np.int32 and store.event_data["x"].dtype == np.int16
    for a, b in ((0, 5000), (17, 1234), (4999, 5000)):
        got = store.load_event(a, b)
        assert got.dtype == np.float64 and np.array_equal(got, O.events_from_raw_columns(x, y, t, p, a, b))
    with pytest.raises(IndexError):
        store.load_event(10, 5001)
    with pytest.raises(IndexError):
        store.load_event(7, 7)
    # index <-> time: searchsorted - 1 (src/data_loader/ccs.py:355-356)
    times = t / 1e6
    assert store.index_to_time(42) == times[42]
    for q in (times[0] - 1.0, times[100], 0.5 * (times[200] + times[201]), times[-1] + 1.0):
        assert store.time_to_index(q) == int(np.searchsorted(times, q)) - 1


def test_store_keeps_64_bit_ticks():
    from event_based_bos_amd.data_loader import RawEventStore

    t = np.array([2 ** 31 + 5, 2 ** 31 + 9, 2 ** 33], dtype=np.int64)
    store = RawEventStore({"x": np.zeros(3), "y": np.zeros(3), "t": t, "p": np.ones(3)})
    assert store.event_data["t"].dtype == np.int64
    assert store.load_event(0, 3)[2, 2] == 2 ** 33 / 1e6


def test_build_raw_validates_without_gpu():
    import event_based_bos_amd as ebos

    if torch.cuda.is_available():
        pytest.skip("argument checks of the GPU-less container")
    z = torch.zeros(4, dtype=torch.int16)
    with pytest.raises(ebos.HipUnavailableError):
        ebos.EventPlan.build_raw(z, z, torch.zeros(4, dtype=torch.int32), torch.zeros(4, dtype=torch.bool), (H, W))


@pytest.mark.gpu
@pytest.mark.parametrize("direction,normalize,t64", [("first", True, False), ("middle", True, True), (0.25, False, False),
                                                    ("last", True, False)])
def test_raw_plan_equals_float64_plan(direction, normalize, t64):
    import event_based_bos_amd as ebos

    x, y, t, p = O.synth_raw_columns(40000, H, W, seed=4)
    if t64:
        t = t.astype(np.int64) + 2 ** 33
    store = ebos.data_loader.RawEventStore({"x": x, "y": y, "t": t, "p": p})
    a, b = 123, 39000
    ref = ebos.EventPlan.build(torch.from_numpy(store.load_event(a, b)).cuda(), (H, W), direction, normalize, tile=None)
    col, row, tt, pol = store.load_raw(a, b)
    raw = ebos.EventPlan.build_raw(col, row, tt, pol, (H, W), direction, normalize, tile=None)
    for name in ("x", "y", "dt", "p"):
        assert torch.equal(getattr(raw, name), getattr(ref, name)), name  # BIT-EXACT
    # and through the whole fused path: same IWE as the oracle on the reference-format window
    flow = O.synth_dense_flow(H, W, seed=5, max_val=6.0)
    plan = store.plan(a, b, (H, W), "first", True, tile="auto")
    iwe = plan.iwe_dense(torch.from_numpy(flow).float().cuda())
    expect = O.iwe_dense(torch.from_numpy(store.load_event(a, b)), torch.from_numpy(flow), (H, W))
    assert O.rel_l2(iwe.cpu().numpy(), expect.numpy()) < 1e-4  # north_star tolerance, fp32 path vs fp64


@pytest.mark.gpu
def test_raw_plan_rejects_bad_input():
    import event_based_bos_amd as ebos

    z16 = torch.zeros(4, dtype=torch.int16, device="cuda")
    t = torch.zeros(4, dtype=torch.int32, device="cuda")
    pol = torch.zeros(4, dtype=torch.bool, device="cuda")
    with pytest.raises(ValueError):
        ebos.EventPlan.build_raw(z16.float(), z16, t, pol, (H, W))
    with pytest.raises(ValueError):
        ebos.EventPlan.build_raw(z16, z16, t[:3], pol, (H, W))
    with pytest.raises(IndexError):
        ebos.EventPlan.build_raw(z16[:0], z16[:0], t[:0], pol[:0], (H, W))


@pytest.mark.gpu
def test_deferred_plan_equals_synchronous_plan():
    """bin(deferred=True) never reads back to the host; with out-of-image events present it must give the same IWE
    (those events are in no tile range) and report them through counts()."""
    import event_based_bos_amd as ebos

    x, y, t, p = O.synth_raw_columns(30000, H, W, seed=9)
    x[::97] = W + 3          # outside the sensor: dropped by the binning
    y[5::101] = -2
    store = ebos.data_loader.RawEventStore({"x": x, "y": y, "t": t, "p": p})
    flow = torch.from_numpy(O.synth_dense_flow(H, W, seed=5, max_val=6.0)).float().cuda()
    sync = store.plan(0, 30000, (H, W), "first", True, tile="auto")
    lazy = store.plan(0, 30000, (H, W), "first", True, tile="auto", deferred=True)
    dropped = int(((x >= W) | (y < 0)).sum())
    assert sync.n == 30000 - dropped and lazy.n == 30000 and lazy.compact
    assert lazy.counts() == (dropped, 0) and sync.counts() == (dropped, 0)
    assert torch.equal(lazy.iwe_dense(flow), sync.iwe_dense(flow))  # BIT-EXACT (fixed-point accumulation)
    f1, f2 = flow.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    lazy.contrast_dense(f1).backward()
    sync.contrast_dense(f2).backward()
    assert O.rel_l2(f1.grad.cpu().numpy(), f2.grad.cpu().numpy()) < 1e-6
    with pytest.raises(NotImplementedError):  # per-event weights need the exact event count of a synchronous build
        lazy.iwe_dense(flow, weight=torch.ones(30000, device="cuda"))
    # a float plan with fractional coordinates cannot be deferred
    ev = np.stack([y + 0.25, x, t / 1e6, p], 1).astype(np.float64)
    frac = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=None).bin((32, 32), deferred=True)
    with pytest.raises(ValueError):
        frac.counts()
