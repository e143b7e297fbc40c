"""Event ingest: the raw-column store against the oracle's restatement of the reference loader,
and -- on the GPU -- the raw-column plan against the plan of the float64 window (bit-exact)."""
import os

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
@pytest.mark.parametrize("emit", ["full", "compact"])
def test_raw_ingest_of_the_reference_loaders_recording(emit):
    """The raw columns of the recording the REFERENCE's loader read (tests/golden/golden_loader.npz: its h5py_loader's arrays and its
    load_event windows, src/data_loader/ccs.py:48-66, 247-297) through the device-side ingest: the SoA of the plan is the reference
    window bit for bit, and the IWE of the fused path is the oracle's on the reference's own [n, 4] array."""
    import event_based_bos_amd as ebos

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_loader.npz")) as f:
        g = {k: f[k] for k in f.files}
    h, w = (int(v) for v in g["size"])
    store = ebos.data_loader.RawEventStore({k: g["file_" + k] for k in "xytp"})   # (as they sit in the HDF5 file: uint16 / int64 / uint8)
    a, b = (int(v) for v in g["windows"][1])
    window = g["window1"]                                                            # the reference loader's load_event(a, b)
    ref = ebos.EventPlan.build(torch.from_numpy(window).cuda(), (h, w), "first", True, tile=None)
    raw = ebos.EventPlan.build_raw(*store.load_raw(a, b), (h, w), "first", True, tile=None)
    for name in ("x", "y", "dt", "p"):
        assert torch.equal(getattr(raw, name), getattr(ref, name)), name
    flow = O.synth_dense_flow(h, w, seed=5, max_val=6.0)
    plan = store.plan(a, b, (h, w), "first", True, tile="auto", emit=emit)
    iwe = plan.iwe_dense(torch.from_numpy(flow).float().cuda(), halo=32 if emit == "compact" else "auto")
    expect = O.iwe_dense(torch.from_numpy(window), torch.from_numpy(flow), (h, w))
    assert O.rel_l2(iwe.cpu().numpy(), expect.numpy()) < 1e-4


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
    # (integer accumulation: bit-identical per work item; two BUILDS may differ by one ulp where a tile is cut into parts)
    a_img, b_img = lazy.iwe_dense(flow), sync.iwe_dense(flow)
    assert float((a_img - b_img).abs().max()) <= 4e-7 * float(b_img.abs().max())
    f1, f2 = flow.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    lazy.contrast_dense(f1).backward()
    sync.contrast_dense(f2).backward()
    # (two builds order the events inside a source pixel differently: the backward kernel's runs, each rounded to the fixed-point
    # unit of its tile, group differently)
    assert O.rel_l2(f1.grad.cpu().numpy(), f2.grad.cpu().numpy()) < 2e-5
    with pytest.raises(NotImplementedError):  # per-event weights need the exact event count of a synchronous build
        lazy.iwe_dense(flow, weight=torch.ones(30000, device="cuda"))
    # a lean plan is valid iff no source coordinate is fractional; a deferred build never looks: float [n, 4] sources are refused
    # (ADVICE r02), the raw int16 columns above are integers by construction
    ev_f = torch.from_numpy(np.stack([y, x, t / 1e6, p], 1).astype(np.float64)).cuda()
    with pytest.raises(ValueError, match="deferred=True needs integer source coordinates"):
        ebos.EventPlan.build(ev_f, (H, W), "first", True, tile="auto", emit="compact", deferred=True)
    # a float plan with fractional coordinates cannot be deferred
    ev = np.stack([y + 0.25, x, t / 1e6, p], 1).astype(np.float64)
    frac = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=None).bin((32, 32), deferred=True)
    with pytest.raises(ValueError):
        frac.counts()


# ---------------------------------------------------------------------------------------------------------------------
# lean plan build (emit="compact", ebos_plan_lean): same compact plan as the full build
# ---------------------------------------------------------------------------------------------------------------------
def _canon(plan):
    """(key_offsets, grp_offsets, per-slot pixel, per-pixel sorted dt) of a compact plan -- the order of the events inside
    one source pixel is unspecified (atomic cursors), so dt is sorted inside every pixel run before comparing."""
    ko = plan.key_offsets.cpu().numpy().astype(np.int64)
    go = plan.grp_offsets.cpu().numpy().astype(np.int64)
    th, tw = plan.tile
    cpix = plan.cpix.cpu().numpy().view(np.uint16)
    cdt = plan.cdt.cpu().numpy()
    pix_all, dt_all = [], []
    for t in range(len(go) - 1):
        n_t = ko[(t + 1) * th * tw] - ko[t * th * tw]
        s0 = 4 * go[t]
        px, dt = cpix[s0:s0 + n_t].astype(np.int64), cdt[s0:s0 + n_t]
        pad_dt = cdt[s0 + n_t:4 * go[t + 1]]
        assert np.isnan(pad_dt).all() and len(pad_dt) < 4, "padding slots of a tile carry dt = NaN"
        key = (px >> 8) * tw + (px & 255)
        assert (np.diff(key) >= 0).all(), "events of a tile are sorted by source pixel"
        order = np.lexsort((dt, key))
        pix_all.append(px[order])
        dt_all.append(dt[order])
    return ko, go, np.concatenate(pix_all), np.concatenate(dt_all)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["f64", "f32", "raw32", "raw64", "oob", "skew", "small_tiles", "middle"])
def test_lean_plan_is_the_compact_part_of_the_full_plan(case):
    """emit="compact" (two-level counting sort, no SoA / perm) against the full build: key_offsets and grp_offsets equal,
    cpix / cdt BIT-IDENTICAL once the events inside each source pixel are put in a canonical order, same counts, same
    part table -- and bit-identical images (integer accumulation)."""
    import event_based_bos_amd as ebos

    h, w, n, tile, direction, norm = 180, 240, 150_000, (45, 80), "first", True
    if case == "small_tiles":
        h, w, tile = 100, 150, (32, 32)
    if case == "middle":
        direction, norm = "middle", False
    x, y, t, p = O.synth_raw_columns(n, h, w, seed=12)
    if case == "oob":
        x[::53] = w + 2
        y[7::61] = -1
    if case == "skew":  # most events in one corner: bins far beyond the LDS staging are gathered chunk by chunk of their pixels
        n = 400_000
        x, y, t, p = O.synth_raw_columns(n, h, w, seed=13)
        x[: n * 3 // 4] = x[: n * 3 // 4] % 70
        y[: n * 3 // 4] = y[: n * 3 // 4] % 20
    if case == "raw64":
        t = t.astype(np.int64) + 2 ** 33
    store = ebos.data_loader.RawEventStore({"x": x, "y": y, "t": t, "p": p})
    if case.startswith("raw") or case in ("oob", "skew"):
        full = store.plan(0, n, (h, w), direction, norm, tile=tile)
        lean = store.plan(0, n, (h, w), direction, norm, tile=tile, emit="compact")
    else:
        ev = torch.from_numpy(store.load_event(0, n)).cuda()
        if case == "f32":
            ev = ev.float()
        full = ebos.EventPlan.build(ev, (h, w), direction, norm, tile=tile)
        lean = ebos.EventPlan.build(ev, (h, w), direction, norm, tile=tile, emit="compact")
    assert lean.lean and lean.compact and not full.lean and lean.x is None and lean.perm is None
    assert (lean.n, lean.n_input, lean.n_dropped, lean.tile) == (full.n, full.n_input, full.n_dropped, full.tile)
    assert lean.counts() == full.counts()
    a, b = _canon(full), _canon(lean)
    for u, v, name in zip(a, b, ("key_offsets", "grp_offsets", "cpix", "cdt")):
        assert np.array_equal(u, v), name
    assert torch.equal(lean.part_table, full.part_table)
    flow = torch.from_numpy(O.synth_dense_flow(h, w, seed=5, max_val=9.0)).float().cuda()
    halo = 32 if tile == (45, 80) else 16
    # Images are bit-identical between two plans of the same window when every tile is ONE work item.  When tiles are cut into
    # parts (adaptive work items: few tiles on many CUs, or skewed windows) the parts are ranges of the tile's groups, so the
    # unspecified order of the events inside a source pixel decides which part an event lands in; each part's integer field
    # sums are converted to f32 on their own, and a pixel may then differ by one ulp between two BUILDS (never between two
    # evaluations of one plan).  "skew" also overflows the fixed-point fields: its exact f64 redo sums floats in event order.
    one_item_per_tile = lean.resolve_splits(None) == 1 and full.resolve_splits(None) == 1
    a_img, b_img = lean.iwe_dense(flow, halo=halo), full.iwe_dense(flow, halo=halo)
    if one_item_per_tile and case != "skew":
        assert torch.equal(a_img, b_img)
    else:
        assert float((a_img - b_img).abs().max()) <= 4e-7 * float(b_img.abs().max()) and O.rel_l2(a_img.cpu().numpy(), b_img.cpu().numpy()) < 1e-7
    if case != "skew":
        assert torch.equal(a_img, lean.iwe_dense(flow, halo=halo))  # one plan, two evaluations: always bit-identical
    f1, f2 = flow.clone().requires_grad_(True), flow.clone().requires_grad_(True)
    lean.contrast_dense(f1, halo=halo).backward()
    full.contrast_dense(f2, halo=halo).backward()
    # (two builds order the events inside a source pixel differently, so the backward kernel's runs -- each rounded to the
    # fixed-point unit of its tile -- group differently; ONE plan evaluated twice gives the same bits: integer scatter)
    assert O.rel_l2(f1.grad.cpu().numpy(), f2.grad.cpu().numpy()) < 2e-5
    if case != "skew":
        f3 = flow.clone().requires_grad_(True)
        lean.contrast_dense(f3, halo=halo).backward()
        assert torch.equal(f1.grad, f3.grad)
    th = torch.tensor([[2.5, -4.0]], device="cuda")
    assert O.rel_l2(lean.iwe_2dof(th, halo=halo).cpu().numpy(), full.iwe_2dof(th, halo=halo).cpu().numpy()) < (
        1e-30 if (one_item_per_tile and case != "skew") else 1e-6)
    assert torch.equal(lean.pixel_event_counts(), full.pixel_event_counts())
    with pytest.raises(NotImplementedError):   # what a lean plan cannot do says so
        lean.iwe_dense(flow, weight=torch.ones(n, device="cuda"))
    with pytest.raises(NotImplementedError):
        lean.iwe_dense(flow, halo=None)


@pytest.mark.gpu
def test_lean_build_falls_back_for_fractional_sources_and_runs_deferred():
    import event_based_bos_amd as ebos

    h, w, n = 96, 128, 20_000
    ev = O.synth_events(n, h, w, seed=3)
    ev[::4, 0] += 0.5
    plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (h, w), "first", True, tile=(32, 32), emit="compact")
    assert not plan.lean and not plan.compact  # fractional coordinates: the (x, y, dt) format of the full build
    x, y, t, p = O.synth_raw_columns(n, h, w, seed=4)
    x[::97] = w + 3
    store = ebos.data_loader.RawEventStore({"x": x, "y": y, "t": t, "p": p})
    flow = torch.from_numpy(O.synth_dense_flow(h, w, seed=5, max_val=6.0)).float().cuda()
    sync = store.plan(0, n, (h, w), "first", True, tile=(32, 32), emit="compact")
    lazy = store.plan(0, n, (h, w), "first", True, tile=(32, 32), emit="compact", deferred=True)  # no host read-back at all
    dropped = int((x >= w).sum())
    assert sync.n == n - dropped and lazy.n == n and lazy.lean and lazy.counts() == (dropped, 0)
    a_img, b_img = lazy.iwe_dense(flow, halo=16), sync.iwe_dense(flow, halo=16)
    assert float((a_img - b_img).abs().max()) <= 4e-7 * float(b_img.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "hot_pixels", "blob"])
def test_lean_builds_of_one_window_are_identical_and_solve_identically(kind):
    """The lean build's cursors hand out a pixel's slots in the order their atomics arrive; every pixel's run leaves the bin sort in
    ascending dt (ranked in LDS; hot pixels by a bitonic network; overfull bins in chunks): two builds of one window hold the same
    arrays, each run is sorted, the events are those of the full build -- and the 2-DoF Adam loop, whose backward sweep sums a
    group's events in slot order, walks the same trajectory bit for bit (two solves of one integer window drifted apart after ~35
    iterations before)."""
    import event_based_bos_amd as ebos
    from event_based_bos_amd.solver.fused_loop import Fused2dofLoop

    h, w, n = 720, 1280, 1_000_000
    rs = np.random.RandomState(3)
    if kind == "blob":   # 28 events per pixel in the middle: overfull bins, chunked
        r, c = np.empty(0), np.empty(0)
        while len(r) < n:
            rr, cc = np.rint(rs.normal(h / 2, 56, n)), np.rint(rs.normal(w / 2, 100, n))
            ok = (rr >= 0) & (rr < h) & (cc >= 0) & (cc < w)
            r, c = np.concatenate([r, rr[ok]]), np.concatenate([c, cc[ok]])
        r, c = r[:n], c[:n]
    else:
        r, c = rs.randint(0, h, n).astype(np.float64), rs.randint(0, w, n).astype(np.float64)
    if kind == "hot_pixels":
        r[:3000], c[:3000] = 100, 200          # one sensor pixel firing 3 000 times, another 12 000 times
        r[3000:15000], c[3000:15000] = 300, 700
    ev = np.stack([r, c, rs.uniform(0, 0.5, n), rs.randint(0, 2, n)], 1)
    ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
    plans = [ebos.EventPlan.build(ev, (h, w), "first", True, tile="auto", emit="compact") for _ in range(3)]
    assert all(p.lean for p in plans)
    used = 4 * int(plans[0].grp_offsets[-1])
    for p in plans[1:]:
        assert torch.equal(plans[0].cpix[:used], p.cpix[:used]) and torch.equal(plans[0].cdt[:used].view(torch.int32), p.cdt[:used].view(torch.int32))
    full = ebos.EventPlan.build(ev, (h, w), "first", True, tile="auto", emit="full")
    th, tw = plans[0].tile
    ko, grp = plans[0].key_offsets.cpu().numpy(), plans[0].grp_offsets.cpu().numpy().astype(np.int64)
    cdt, fdt = plans[0].cdt.cpu().numpy(), full.cdt.cpu().numpy()
    for t in range(len(grp) - 1):
        offs = ko[t * th * tw:(t + 1) * th * tw + 1] - ko[t * th * tw]
        seg, fseg = cdt[4 * grp[t]:4 * grp[t] + offs[-1]], fdt[4 * grp[t]:4 * grp[t] + offs[-1]]
        runs = np.repeat(np.arange(th * tw), np.diff(offs))
        np.testing.assert_array_equal(seg, fseg[np.lexsort((fseg, runs))])     # the full build's events, every run in ascending dt
    losses = []
    for p in plans:
        loop = Fused2dofLoop(p, torch.zeros(2), 1.0, False, 0, "auto", lr=0.05, capacity=128, blur_sigma=3.0)
        losses.append(loop.run(100).cpu().numpy().copy())
    np.testing.assert_array_equal(losses[0], losses[1])
    np.testing.assert_array_equal(losses[0], losses[2])


@pytest.mark.gpu
def test_lean_build_fuzz_canonical_order_and_same_events_as_the_full_build():
    """Random sensors, tiles, event counts and distributions (uniform; a hot pixel of up to 15 000 events; a blob that overfills bins; runs
    of equal timestamps with events outside the image): two lean builds hold identical arrays, every pixel's run is in ascending dt, and
    the events are those of the full build."""
    import event_based_bos_amd as ebos

    rs = np.random.RandomState(123)
    for trial in range(24):
        H, W = int(rs.randint(40, 800)), int(rs.randint(40, 1300))
        n = int(rs.choice([2, 7, 1000, 50_000, 400_000, 1_500_000]))
        kind = rs.randint(0, 4)
        r, c = rs.randint(0, H, n).astype(np.float64), rs.randint(0, W, n).astype(np.float64)
        if kind == 1 and n > 10:   # a hot pixel
            k = min(n // 3, 15000)
            r[:k], c[:k] = rs.randint(0, H), rs.randint(0, W)
        elif kind == 2:            # a blob
            r = np.clip(np.rint(rs.normal(H / 2, H / 12, n)), 0, H - 1)
            c = np.clip(np.rint(rs.normal(W / 2, W / 12, n)), 0, W - 1)
        elif kind == 3:            # some events outside the image
            r[: n // 10] = -3
        t = rs.uniform(0, 0.5, n)
        if kind == 3:
            t = np.round(t, 2)     # ... and runs of equal timestamps
        ev = np.stack([r, c, t, rs.randint(0, 2, n)], 1)
        ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
        tile = [(32, 32), (45, 80), (32, 64), "auto"][rs.randint(0, 4)]
        plans = [ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="compact") for _ in range(2)]
        assert plans[0].lean, (trial, H, W, n, kind, tile)
        full = ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="full")
        used = 4 * int(plans[0].grp_offsets[-1])
        assert torch.equal(plans[0].cpix[:used], plans[1].cpix[:used]), (trial, H, W, n, kind, tile)
        assert torch.equal(plans[0].cdt[:used].view(torch.int32), plans[1].cdt[:used].view(torch.int32)), (trial, H, W, n, kind, tile)
        th, tw = plans[0].tile
        ko, grp = plans[0].key_offsets.cpu().numpy(), plans[0].grp_offsets.cpu().numpy().astype(np.int64)
        np.testing.assert_array_equal(ko, full.key_offsets.cpu().numpy())
        cdt, fdt, cpx, fpx = plans[0].cdt.cpu().numpy(), full.cdt.cpu().numpy(), plans[0].cpix.cpu().numpy(), full.cpix.cpu().numpy()
        for t_ in range(len(grp) - 1):
            offs = ko[t_ * th * tw:(t_ + 1) * th * tw + 1] - ko[t_ * th * tw]
            a, b = 4 * grp[t_], 4 * grp[t_] + offs[-1]
            runs = np.repeat(np.arange(th * tw), np.diff(offs))
            np.testing.assert_array_equal(cdt[a:b], fdt[a:b][np.lexsort((fdt[a:b], runs))], err_msg=str((trial, H, W, n, kind, tile)))
            np.testing.assert_array_equal(cpx[a:b], fpx[a:b])


@pytest.mark.gpu
@pytest.mark.parametrize("emit", ["full", "compact"])
def test_plan_facts_read_back(emit):
    """ebos_plan_facts (the build's one read-back: dropped, fractional, work items in use, fullest tile) against the plan's own arrays."""
    import event_based_bos_amd as ebos

    h, w, n, tile = 180, 240, 60_000, (45, 80)
    x, y, t, p = O.synth_raw_columns(n, h, w, seed=21)
    x[::41] = w + 1            # outside the image
    x[: n // 2] %= 60          # one crowded corner
    y[: n // 2] %= 30
    store = ebos.data_loader.RawEventStore({"x": x, "y": y, "t": t, "p": p})
    plan = store.plan(0, n, (h, w), "first", True, tile=tile, emit=emit)
    dropped = int(((x >= w) | (x < 0) | (y >= h) | (y < 0)).sum())
    assert plan.counts() == (dropped, 0) and plan.n == n - dropped
    ko = plan.key_offsets.cpu().numpy().astype(np.int64)
    n_tiles = (len(ko) - 1) // (tile[0] * tile[1])
    assert plan.__dict__["_fullest_tile"] == int(np.diff(ko[:: tile[0] * tile[1]]).max())
    assert plan.__dict__["_parts_used"] == int(plan.part_table[n_tiles])
    if emit == "full":   # the C entry on its own: counts / part_table are optional (NULL: zeros), the fullest tile always comes
        from event_based_bos_amd import _hip
        from event_based_bos_amd._hip import check, ptr, stream_ptr

        facts = torch.full((4,), -1, dtype=torch.int32, device="cuda")
        check(_hip.require_gpu().ebos_plan_facts(ptr(plan.key_offsets), h, w, tile[0], tile[1], None, None, ptr(facts), stream_ptr()), "ebos_plan_facts")
        assert facts.tolist() == [0, 0, 0, plan.__dict__["_fullest_tile"]]
        assert _hip.require_gpu().ebos_plan_facts(None, h, w, tile[0], tile[1], None, None, ptr(facts), stream_ptr()) != 0   # NULL key_offsets: refused


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["two_chunks", "dense", "one_run_beyond_the_staging_area"])
def test_lean_build_of_dense_windows_gathers_overfull_bins_in_chunks(case):
    """Windows beyond ~10 k events per tile leave the bins LARGER than the bin sort's LDS staging on purpose (fewer, longer (chunk, bin)
    runs: the stage kernel's table and the gather were what the build cost): an overfull bin is gathered once for its histogram and once
    per chunk of whole pixels.  Same arrays from two builds, every run in ascending dt, the events of the full build; a single run larger
    than the whole staging area (a pixel firing 30 000 times) is written in the order it arrives -- same events, order unspecified."""
    import event_based_bos_amd as ebos

    H, W, tile = 128, 128, (64, 64)
    n = {"two_chunks": 120_000, "dense": 900_000, "one_run_beyond_the_staging_area": 150_000}[case]
    rs = np.random.RandomState(31)
    r, c = rs.randint(0, H, n).astype(np.float64), rs.randint(0, W, n).astype(np.float64)
    if case == "one_run_beyond_the_staging_area":
        r[:30_000], c[:30_000] = 70, 9
    ev = np.stack([r, c, rs.uniform(0, 0.5, n), rs.randint(0, 2, n)], 1)
    ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
    plans = [ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="compact") for _ in range(2)]
    full = ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="full")
    assert plans[0].lean and plans[0].tile == tile
    th, tw = tile
    ko, grp = plans[0].key_offsets.cpu().numpy(), plans[0].grp_offsets.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(ko, full.key_offsets.cpu().numpy())
    cdt, cdt2, fdt = plans[0].cdt.cpu().numpy(), plans[1].cdt.cpu().numpy(), full.cdt.cpu().numpy()
    cpx, cpx2, fpx = plans[0].cpix.cpu().numpy(), plans[1].cpix.cpu().numpy(), full.cpix.cpu().numpy()
    for t_ in range(len(grp) - 1):
        offs = ko[t_ * th * tw:(t_ + 1) * th * tw + 1] - ko[t_ * th * tw]
        a, b = 4 * grp[t_], 4 * grp[t_] + offs[-1]
        runs = np.repeat(np.arange(th * tw), np.diff(offs))
        want = fdt[a:b][np.lexsort((fdt[a:b], runs))]
        np.testing.assert_array_equal(cpx[a:b], fpx[a:b])
        np.testing.assert_array_equal(cpx2[a:b], fpx[a:b])
        if case == "one_run_beyond_the_staging_area":
            giant = np.diff(offs) >= 30_000
            keep = ~giant[runs]
            np.testing.assert_array_equal(cdt[a:b][keep], want[keep])
            np.testing.assert_array_equal(cdt2[a:b][keep], want[keep])
            np.testing.assert_array_equal(cdt[a:b][np.lexsort((cdt[a:b], runs))], want)
        else:
            np.testing.assert_array_equal(cdt[a:b], want)
            np.testing.assert_array_equal(cdt2[a:b], want)


@pytest.mark.gpu
def test_fractional_compact_layout_fuzz_canonical_order():
    """The compact layout with fractions per slot (EventPlan.frac_compact): two builds hold identical arrays, every pixel's run is in
    ascending (dt, fx, fy) -- ranked by the fill for short runs, sorted in LDS for hot pixels of up to 4 096 events --, and the slots are
    the plan's events (same multiset per pixel as the (x, y, dt) arrays)."""
    import event_based_bos_amd as ebos

    rs = np.random.RandomState(321)
    for trial in range(16):
        H, W = int(rs.randint(40, 500)), int(rs.randint(40, 700))
        n = int(rs.choice([7, 1000, 50_000, 300_000]))
        r, c = rs.uniform(0, H - 1, n), rs.uniform(0, W - 1, n)
        if trial % 3 == 1 and n > 100:   # a hot source pixel with a few distinct fractions
            k = min(n // 3, 4000)
            r[:k] = rs.randint(0, H - 1) + rs.choice([0.25, 0.5, 0.75], k)
            c[:k] = rs.randint(0, W - 1) + rs.choice([0.125, 0.5], k)
        t = rs.uniform(0, 0.5, n)
        if trial % 3 == 2:
            t = np.round(t, 2)           # equal timestamps: the fractions decide
        ev = np.stack([r, c, t, rs.randint(0, 2, n)], 1)
        ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
        tile = [(32, 32), (45, 80), (32, 64), "auto"][rs.randint(0, 4)]
        _check_frac_layout(ev, (H, W), tile, (trial, H, W, n, tile))


def _check_frac_layout(ev, size, tile, tag):
    import event_based_bos_amd as ebos

    H, W = size
    plans = [ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile) for _ in range(2)]
    assert plans[0].__dict__["_frac"] is None and plans[0].__dict__["_frac_pending"]   # built on the first access, not by the build
    fa, fb = plans[0].frac_compact, plans[1].frac_compact
    assert fa is not None and fb is not None and plans[0].frac_compact is fa
    used = 4 * int(fa[0][-1])
    for a, b in zip(fa[1:], fb[1:]):
        assert torch.equal(a[:used].view(torch.int32) if a.dtype == torch.float32 else a[:used], b[:used].view(torch.int32) if b.dtype == torch.float32 else b[:used]), tag
    th, tw = plans[0].tile
    ko, grp = plans[0].key_offsets.cpu().numpy(), fa[0].cpu().numpy().astype(np.int64)
    cdt, cfx, cfy = (a.cpu().numpy() for a in fa[2:])
    xs, ys, dts = plans[0].x.cpu().numpy(), plans[0].y.cpu().numpy(), plans[0].dt.cpu().numpy()
    for t_ in range(len(grp) - 1):
        beg = ko[t_ * th * tw]
        offs = ko[t_ * th * tw:(t_ + 1) * th * tw + 1] - beg
        a, b = 4 * grp[t_], 4 * grp[t_] + offs[-1]
        runs = np.repeat(np.arange(th * tw), np.diff(offs))
        ex, ey, ed = xs[beg:beg + offs[-1]], ys[beg:beg + offs[-1]], dts[beg:beg + offs[-1]]
        efx, efy = ex - np.trunc(ex), ey - np.trunc(ey)
        order = np.lexsort((efy, efx, ed, runs))
        np.testing.assert_array_equal(cdt[a:b], ed[order], err_msg=str(tag))
        np.testing.assert_array_equal(cfx[a:b], efx[order].astype(np.float32))
        np.testing.assert_array_equal(cfy[a:b], efy[order].astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["many_hot_pixels", "very_long_run", "long_run_exact_chunks"])
def test_fractional_compact_layout_is_canonical_on_noisy_sensors(case):
    """More than 64 hot pixels in ONE tile (the hot-pixel pass used to canonicalise the first 64 its atomics listed), and a run longer
    than the 4 096 events one LDS sort holds (used to keep its scatter order): both canonical now."""
    rs = np.random.RandomState({"many_hot_pixels": 1, "very_long_run": 2, "long_run_exact_chunks": 3}[case])
    H, W = 90, 160
    n_bg = 20_000
    r, c = rs.uniform(0, H - 1, n_bg), rs.uniform(0, W - 1, n_bg)
    if case == "many_hot_pixels":   # 150 hot pixels of 70 .. 400 events inside the tile at (0, 0) of a 45 x 80 tiling
        px = rs.choice(45 * 80, 150, replace=False)
        lens = rs.randint(70, 400, 150)
        hr = np.repeat(px // 80, lens) + rs.choice([0.0, 0.25, 0.5, 0.75], lens.sum())
        hc = np.repeat(px % 80, lens) + rs.choice([0.125, 0.5, 0.875], lens.sum())
    else:
        k = 11_111 if case == "very_long_run" else 3 * 2048
        hr = 50 + rs.choice([0.0, 0.25, 0.5, 0.75], k)
        hc = 100 + rs.choice([0.125, 0.5, 0.875], k)
    r, c = np.concatenate([r, hr]), np.concatenate([c, hc])
    t = np.round(rs.uniform(0, 0.5, len(r)), 3)   # many equal timestamps: the fractions decide
    ev = np.stack([r, c, t, rs.randint(0, 2, len(r))], 1)
    ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
    _check_frac_layout(ev, (H, W), (45, 80), case)
