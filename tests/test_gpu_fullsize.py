"""Parity at BASELINE.json's full sizes (10 M events, 1280x720), on the GPU box.

Config 2 (variance) and config 3 (gradient magnitude, cost + gradient vs CPU autograd) are compared with the
CPU oracle directly (a few seconds of host time each), plus size-independent properties of the domain:
mass conservation, linearity in the per-event weight, additivity over event subsets, bit-reproducibility of the
fixed-point path, and agreement of every kernel organisation (tile-private / atomic tiled / general)."""
import numpy as np
import pytest
import torch

from oracle import ebos_oracle as O

pytestmark = pytest.mark.gpu

H, W, N = 720, 1280, 10_000_000


@pytest.fixture(scope="module")
def setup():
    import event_based_bos_amd as ebos

    dev = torch.device("cuda:0")
    ev = O.synth_events(N, H, W, seed=0)
    fl = O.synth_dense_flow(H, W, seed=1, max_val=30.0)
    plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile="auto")
    assert plan.tile == (45, 80)  # 16 x 16 = 256 tiles: one per CU
    flow = torch.from_numpy(fl).float().to(dev)
    return ebos, ev, fl, plan, flow


def test_config2_variance_full_size_vs_oracle(setup):
    ebos, ev, fl, plan, flow = setup
    assert plan.n == N and plan.compact
    f = torch.from_numpy(fl).requires_grad_(True)
    iwe_ref = O.iwe_dense(torch.from_numpy(ev), f, (H, W))
    loss_ref = O.image_variance(iwe_ref)
    loss_ref.backward()
    fg = flow.clone().requires_grad_(True)
    loss = -plan.contrast_dense(fg, "image_variance")
    loss.backward()
    iwe = plan.iwe_dense(flow)
    assert O.rel_l2(iwe.cpu().numpy(), iwe_ref.detach().numpy()) < 1e-5          # north_star bar: 1e-4
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())
    e_grad = O.rel_l2(fg.grad.cpu().numpy(), f.grad.numpy())
    print(f"[config 2, un-filtered stream of {N} events] IWE rel-L2 {O.rel_l2(iwe.cpu().numpy(), iwe_ref.detach().numpy()):.2e}, "
          f"loss rel {abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()):.2e}, d loss / d flow rel-L2 {e_grad:.2e}")
    assert e_grad < 1e-3                                                           # SURVEY 8d bar
    # ~1.6 % of the mass leaves the image (IWE sum 9 838 422.9 of 10 M)
    assert abs(iwe.sum().item() - 9_838_422.9) < 50.0


def test_config3_gradient_magnitude_full_size_vs_cpu_autograd(setup):
    ebos, ev, fl, plan, flow = setup
    f = torch.from_numpy(fl).requires_grad_(True)
    loss_ref = O.gradient_magnitude(O.iwe_dense(torch.from_numpy(ev), f, (H, W)))
    loss_ref.backward()
    fg = flow.clone().requires_grad_(True)
    loss = -plan.contrast_dense(fg, "gradient_magnitude")
    loss.backward()
    e_grad = O.rel_l2(fg.grad.cpu().numpy(), f.grad.numpy())
    print(f"[config 3, un-filtered stream of {N} events] loss rel {abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()):.2e}, "
          f"d loss / d flow rel-L2 {e_grad:.2e}")
    assert abs(loss.item() - loss_ref.item()) < 1e-5 * abs(loss_ref.item())       # cost rel err < 1e-5
    assert e_grad < 1e-3                                                           # gradient rel-L2 < 1e-3 (no event filtered)


def test_kernel_organisations_agree_and_are_reproducible(setup):
    ebos, ev, fl, plan, flow = setup
    a = plan.iwe_dense(flow, halo=32)            # tile-private slabs, fixed point
    b = plan.iwe_dense(flow, halo=32)
    assert torch.equal(a, b)                     # integer accumulation + fixed combine order: bit-reproducible
    saved, plan.cpix = plan.cpix, None           # 12 B/event (x, y, dt) format
    c = plan.iwe_dense(flow, halo=32)
    plan.cpix = saved
    assert (torch.linalg.norm(c - a) / torch.linalg.norm(a)).item() < 1e-6
    e = plan.iwe_dense(flow, halo=None)          # general kernel, global atomics
    assert (torch.linalg.norm(e - a) / torch.linalg.norm(a)).item() < 1e-6
    p64 = ebos.EventPlan.build(torch.from_numpy(ev).to(flow.device), (H, W), "first", True, tile=(64, 64))
    d = p64.iwe_dense(flow, halo=64)             # atomic-flush tiled kernel (f32 LDS, 64 px halo)
    g = p64.iwe_dense(flow, halo=16)             # halo smaller than the 30 px flow: taps spill, still exact
    k = p64.iwe_dense(flow, halo=32)             # other tile size, tile-private
    for other in (d, g, k):
        assert (torch.linalg.norm(other - a) / torch.linalg.norm(a)).item() < 1e-6


def test_domain_properties_full_size(setup):
    ebos, ev, fl, plan, flow = setup
    dev = flow.device
    # mass conservation with padding wide enough to catch every tap: sum(IWE) == N
    padded = plan.iwe_dense(flow, pad=(32, 32))
    assert abs(padded.double().sum().item() - N) < 1e-6 * N
    # linearity in the weight: IWE(2.5 w) == 2.5 IWE(w);  additivity: IWE(w1) + IWE(w2) == IWE(w1 + w2)
    w1 = torch.rand(N, device=dev)
    w2 = 1.0 - w1
    i1, i2 = plan.iwe_dense(flow, weight=w1), plan.iwe_dense(flow, weight=w2)
    unit = plan.iwe_dense(flow)
    assert (torch.linalg.norm(i1 + i2 - unit) / torch.linalg.norm(unit)).item() < 1e-6
    assert (torch.linalg.norm(plan.iwe_dense(flow, weight=2.5 * w1) - 2.5 * i1) / torch.linalg.norm(i1)).item() < 1e-6
    # zero flow: the IWE is the event histogram (integer counts, exact)
    hist = plan.iwe_dense(torch.zeros_like(flow))
    counts = np.bincount((ev[:, 0].astype(np.int64) * W + ev[:, 1].astype(np.int64)), minlength=H * W).reshape(H, W)
    assert np.array_equal(hist.cpu().numpy(), counts.astype(np.float32))
    # d(variance)/d(flow) of the zero-weight events is zero; d/d(weight) matches finite differences in aggregate
    wz = torch.ones(N, device=dev, requires_grad=True)
    ebos.ops.image_variance(plan.iwe_dense(flow, weight=wz)).backward()
    eps = 1e-2
    up = ebos.ops.image_variance(plan.iwe_dense(flow, weight=torch.full((N,), 1.0 + eps, device=dev)))
    dn = ebos.ops.image_variance(plan.iwe_dense(flow, weight=torch.full((N,), 1.0 - eps, device=dev)))
    fd = (up - dn).item() / (2 * eps)
    assert abs(wz.grad.sum().item() - fd) < 1e-3 * abs(fd)


def test_config4_patch_grid_route_full_size(setup):
    """BASELINE configs[3] shape (2 M events, 30x40 patch grid -> 1280x720 flow, U(-30, 30)): the grid-sampling event kernels
    against the materialised route (IWE / loss 1e-6, d loss / d theta rel-L2 1e-5) and against the fp64 oracle through
    upsample_patch_flow + iwe_dense + var + flow_norm (IWE rel-L2 < 1e-4, loss < 1e-5, gradient rel-L2 < 1e-3); then 20 Adam
    iterations of both routes stay together (losses 1e-4 relative)."""
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    ebos, ev, fl, plan10, flow = setup
    n = 2_000_000
    from _kinks import off_the_kinks_patch

    rs = np.random.RandomState(107)
    theta = rs.uniform(-30, 30, (2, 30, 40))
    # (events AT a kink of the piecewise-linear vote are dropped, ~0.2 %: there the f32 and fp64 paths may legitimately take
    # different one-sided gradients, and the gradient w.r.t. the 2400 patch parameters is a small difference of large sums)
    ev2 = off_the_kinks_patch(O.synth_events(n, H, W, seed=7), theta, (H, W), (24, 32), (24, 32))
    n = len(ev2)
    dev = flow.device
    plan = ebos.EventPlan.build(torch.from_numpy(ev2).to(dev), (H, W), "first", True, tile="auto")
    out = {}
    for grid in (True, False):
        loop = FusedPatchLoop(plan, (24, 32), (24, 32), torch.from_numpy(theta).float().to(dev), 1.0, 0.01, 0.0, capacity=20, lr=0.1,
                              sample_grid=grid)
        assert loop.sample_grid == grid and loop.fuse_norm == grid
        loss, grad = loop.value_and_grad(torch.from_numpy(theta).float().to(dev))
        iwe = loop.iwe.cpu().double().numpy()
        hist = loop.run(20).cpu().numpy()
        out[grid] = (iwe, float(loss), grad.cpu().double().numpy(), hist)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    assert rel(out[True][0], out[False][0]) < 1e-6 and abs(out[True][1] - out[False][1]) <= 1e-6 * abs(out[False][1])
    assert rel(out[True][2], out[False][2]) < 1e-5
    np.testing.assert_allclose(out[True][3], out[False][3], rtol=1e-4)
    tt = torch.from_numpy(theta).requires_grad_(True)
    dense = O.upsample_patch_flow(tt, (H, W), (24, 32), (24, 32))
    iwe_ref = O.iwe_dense(torch.from_numpy(ev2), dense, (H, W))
    loss_ref = -torch.var(iwe_ref) + 0.01 * O.flow_norm(dense)
    loss_ref.backward()
    assert rel(out[True][0], iwe_ref.detach().numpy()) < 1e-4
    assert abs(out[True][1] - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    assert rel(out[True][2], tt.grad.numpy()) < 1e-3
    # mass: every event whose four taps stay inside the image adds exactly one unit
    assert abs(out[True][0].sum() - iwe_ref.sum().item()) <= 1e-6 * n
    # The same comparison on the UN-filtered stream (the ~0.2 % of events within 5e-4 px of a kink kept): value bars unchanged;
    # the gradient w.r.t. the 2400 patch parameters is reported and held to 3e-3 (measured 1.4e-3: a regression of 2x is seen) -- at a
    # kink the f32 path and the fp64 oracle may take different one-sided derivatives of the piecewise-linear vote
    ev3 = O.synth_events(2_000_000, H, W, seed=7)
    plan3 = ebos.EventPlan.build(torch.from_numpy(ev3).to(dev), (H, W), "first", True, tile="auto")
    loop3 = FusedPatchLoop(plan3, (24, 32), (24, 32), torch.from_numpy(theta).float().to(dev), 1.0, 0.01, 0.0, capacity=4, lr=0.1,
                           sample_grid=True)
    loss3, grad3 = loop3.value_and_grad(torch.from_numpy(theta).float().to(dev))
    t3 = torch.from_numpy(theta).requires_grad_(True)
    dense3 = O.upsample_patch_flow(t3, (H, W), (24, 32), (24, 32))
    iwe3 = O.iwe_dense(torch.from_numpy(ev3), dense3, (H, W))
    ref3 = -torch.var(iwe3) + 0.01 * O.flow_norm(dense3)
    ref3.backward()
    e_iwe, e_loss = rel(loop3.iwe.cpu().double().numpy(), iwe3.detach().numpy()), abs(float(loss3) - ref3.item()) / abs(ref3.item())
    e_grad = rel(grad3.cpu().double().numpy(), t3.grad.numpy())
    print(f"[config 4, un-filtered stream of {len(ev3)} events ({len(ev3) - n} near a kink)] IWE rel-L2 {e_iwe:.2e}, loss rel {e_loss:.2e}, "
          f"d loss / d theta rel-L2 {e_grad:.2e} (filtered stream: {rel(out[True][2], tt.grad.numpy()):.2e})")
    assert e_iwe < 1e-4 and e_loss < 1e-5 and e_grad < 3e-3


def test_config4_batched_windows_full_size(setup):
    """BASELINE configs[3] as ebos_iwe_slab_batch_f32 runs it (bench.py --config 4): five independent windows of different sizes at
    1280x720 with the shipped 45x80 + 32 tile, 30x40 patch grids, adaptive work items -- images and variances bit-identical to one
    ebos_iwe_patch_slab_f32 call per window (more than 16 windows: tests/test_gpu_parity.py), the first window also against the
    fp64 oracle (IWE rel-L2 < 1e-4, variance < 1e-5)."""
    ebos, ev, fl, plan10, flow = setup
    from event_based_bos_amd import _hip

    lib = _hip.require_gpu()
    dev = flow.device
    rs = np.random.RandomState(41)
    sizes = [2_000_000, 300_000, 1_000_000, 5_000, 2_000_000]
    evs = [O.synth_events(n, H, W, seed=300 + k) for k, n in enumerate(sizes)]
    plans = [ebos.EventPlan.build(torch.from_numpy(e).to(dev), (H, W), "first", True, tile="auto", emit="compact") for e in evs]
    grids = [torch.from_numpy(rs.uniform(-30, 30, (2, 30, 40))).float().to(dev) for _ in sizes]
    for splits, halo in ((None, 32), (1, 32), (None, "auto"), (1, "auto")):  # adaptive work items / one workgroup per tile; built halo / run-time windows
        batch = ebos.SlabBatch(plans, grids, patch=((24, 32), (24, 32)), splits=splits, halo=halo)
        var = batch.run(tail_stream=torch.cuda.Stream(device=dev).cuda_stream).cpu().numpy()
        torch.cuda.synchronize()
        th, tw = plans[0].tile
        hcode = batch.halo
        nws = int(lib.ebos_iwe_slab_workspace_bytes(H, W, th, tw, hcode, batch.splits, 0, 0))
        ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
        iwe = torch.empty((H, W), dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        mom = torch.empty((1, 2), dtype=torch.float64, device=dev)
        P = lambda t: None if t is None else t.data_ptr()
        for k, (pl, g) in enumerate(zip(plans, grids)):
            _hip.check(lib.ebos_iwe_patch_slab_f32(*pl._compact_ptrs(), P(pl.key_offsets), pl.n, P(g), 30, 40, 24, 32, 24, 32, H, W, th, tw, hcode,
                                                   batch.splits, 0, 0, P(ws), nws, P(iwe), 1, 0, P(out), P(mom),
                                                   P(pl.part_table) if batch.splits == 0 else None, _hip.stream_ptr()), "ebos_iwe_patch_slab")
            assert torch.equal(iwe, batch.iwes[k]), (splits, halo, k)
            assert out.item() == var[k], (splits, halo, k)
    dense = O.upsample_patch_flow(grids[0].cpu().double(), (H, W), (24, 32), (24, 32))
    ref = O.iwe_dense(torch.from_numpy(evs[0]), dense, (H, W))
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    assert rel(batch.iwes[0].cpu().double().numpy(), ref.numpy()) < 1e-4
    assert abs(var[0] - torch.var(ref).item()) <= 1e-5 * torch.var(ref).item()


def test_general_event_formats_at_1280x720_vs_oracle():
    """The formats the headline does not read, at the sensor's size: 2 M events whose source coordinates lie on a 1/64 px grid
    (undistorted events: the 12 B/event (x, y, dt) plan; the flow is looked up at the TRUNCATED coordinate, src/warp.py:334) with
    random per-event weights (src/event_image_converter.py:576-577, 608-613: the f64 LDS accumulation), against the fp64 oracle:
    IWE rel-L2 < 1e-4, d loss / d flow and d loss / d weight rel-L2 < 1e-3 (events within 5e-4 px of a kink of the piecewise-linear
    vote dropped, as in the other gradient tests; the un-filtered errors are printed)."""
    import event_based_bos_amd as ebos

    n = 2_000_000
    rs = np.random.RandomState(21)
    ev = np.stack([rs.randint(0, H, n) + rs.randint(0, 64, n) / 64.0, rs.randint(0, W, n) + rs.randint(0, 64, n) / 64.0,
                   np.sort(rs.uniform(0.0, 0.5, n)), rs.randint(0, 2, n)], 1)
    fl = np.random.RandomState(22).uniform(-12.0, 12.0, (2, H, W))
    wt = rs.uniform(0.25, 2.0, n)
    warped = O.warp_dense_torch(torch.from_numpy(ev), torch.from_numpy(fl), "first", True).numpy().reshape(-1, 4)
    keep = ~(np.abs(warped[:, :2] - np.rint(warped[:, :2])) < 5e-4).any(1)
    keep[[0, -1]] = True      # (the first and the last event fix the time base: they stay)
    evk, wtk = ev[keep], wt[keep]

    def oracle(e, wv):
        f = torch.from_numpy(fl).requires_grad_(True)
        wg = torch.from_numpy(wv).requires_grad_(True)
        iwe = O.iwe_dense(torch.from_numpy(e), f, (H, W), weight=wg)
        loss = O.image_variance(iwe)
        loss.backward()
        return iwe.detach().numpy(), loss.item(), f.grad.numpy(), wg.grad.numpy()

    def hip(e, wv):
        plan = ebos.EventPlan.build(torch.from_numpy(e).float().cuda(), (H, W), "first", True, tile="auto")
        assert plan.binned and not plan.compact            # fractional sources: the (x, y, dt) format
        f = torch.from_numpy(fl).float().cuda().requires_grad_(True)
        wg = torch.from_numpy(wv).float().cuda().requires_grad_(True)
        iwe = plan.iwe_dense(f, weight=wg)
        loss = -ebos.ops.image_variance(iwe)
        loss.backward()
        return iwe.detach().cpu().numpy(), loss.item(), f.grad.cpu().numpy(), wg.grad.cpu().numpy()

    iwe_r, l_r, gf_r, gw_r = oracle(evk, wtk)
    iwe_g, l_g, gf_g, gw_g = hip(evk, wtk)
    e_iwe, e_f, e_w = O.rel_l2(iwe_g, iwe_r), O.rel_l2(gf_g, gf_r), O.rel_l2(gw_g, gw_r)
    print(f"[fractional sources + weights, {len(evk)} of {n} events away from kinks] IWE rel-L2 {e_iwe:.2e}, loss rel "
          f"{abs(l_g - l_r) / abs(l_r):.2e}, d/d flow {e_f:.2e}, d/d weight {e_w:.2e}")
    assert e_iwe < 1e-4 and abs(l_g - l_r) < 1e-5 * abs(l_r) and e_f < 1e-3 and e_w < 1e-3
    iwe_r2, l_r2, gf_r2, gw_r2 = oracle(ev, wt)
    iwe_g2, l_g2, gf_g2, gw_g2 = hip(ev, wt)
    print(f"[the same, un-filtered] IWE rel-L2 {O.rel_l2(iwe_g2, iwe_r2):.2e}, d/d flow {O.rel_l2(gf_g2, gf_r2):.2e}, "
          f"d/d weight {O.rel_l2(gw_g2, gw_r2):.2e}")
    assert O.rel_l2(iwe_g2, iwe_r2) < 1e-4 and O.rel_l2(gw_g2, gw_r2) < 1e-3


def test_weighted_integer_pixel_events_at_1280x720_vs_oracle():
    """Per-event weights on raw sensor events (integer source pixels), at the sensor's size: the compact 6 B/event format with the
    weights riding along in plan order (10 B/event: csrc/iwe_tile_core.h load_weights4, the fixed-point loop in units of the slice's
    max |w|) instead of the 16 B/event (x, y, dt, w) format -- src/event_image_converter.py:576-577, 608-613.  2 M events, weights
    U(0.25, 2) with exact zeros among them (an event of weight 0 still has a d loss / d weight), against the fp64 oracle: IWE rel-L2
    < 1e-4, d loss / d flow and d loss / d weight rel-L2 < 1e-3 (events within 5e-4 px of a kink of the vote dropped).  One
    negative weight sends its tile through the exact f64 pass: same bars."""
    import event_based_bos_amd as ebos

    n = 2_000_000
    rs = np.random.RandomState(31)
    ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0.0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
    fl = np.random.RandomState(32).uniform(-12.0, 12.0, (2, H, W))
    wt = rs.uniform(0.25, 2.0, n)
    wt[rs.randint(0, n, 1000)] = 0.0
    warped = O.warp_dense_torch(torch.from_numpy(ev), torch.from_numpy(fl), "first", True).numpy().reshape(-1, 4)
    keep = ~(np.abs(warped[:, :2] - np.rint(warped[:, :2])) < 5e-4).any(1)
    keep[[0, -1]] = True
    ev, wt = ev[keep], wt[keep]

    def oracle(wv):
        f = torch.from_numpy(fl).requires_grad_(True)
        wg = torch.from_numpy(wv).requires_grad_(True)
        iwe = O.iwe_dense(torch.from_numpy(ev), f, (H, W), weight=wg)
        loss = O.image_variance(iwe)
        loss.backward()
        return iwe.detach().numpy(), loss.item(), f.grad.numpy(), wg.grad.numpy()

    def hip(wv):
        plan = ebos.EventPlan.build(torch.from_numpy(ev).float().cuda(), (H, W), "first", True, tile="auto")
        assert plan.binned and plan.compact and not plan.lean     # the full build: compact events AND the permutation
        f = torch.from_numpy(fl).float().cuda().requires_grad_(True)
        wg = torch.from_numpy(wv).float().cuda().requires_grad_(True)
        iwe = plan.iwe_dense(f, weight=wg)
        loss = -ebos.ops.image_variance(iwe)
        loss.backward()
        return iwe.detach().cpu().numpy(), loss.item(), f.grad.cpu().numpy(), wg.grad.cpu().numpy()

    for tag, wv in (("weights >= 0", wt), ("one negative weight", np.concatenate([wt[:1000], [-1.5], wt[1001:]]))):
        iwe_r, l_r, gf_r, gw_r = oracle(wv)
        iwe_g, l_g, gf_g, gw_g = hip(wv)
        e_iwe, e_f, e_w = O.rel_l2(iwe_g, iwe_r), O.rel_l2(gf_g, gf_r), O.rel_l2(gw_g, gw_r)
        print(f"[integer pixels + weights ({tag}), {len(ev)} events] IWE rel-L2 {e_iwe:.2e}, loss rel {abs(l_g - l_r) / abs(l_r):.2e}, "
              f"d/d flow {e_f:.2e}, d/d weight {e_w:.2e}")
        assert e_iwe < 1e-4 and abs(l_g - l_r) < 1e-5 * abs(l_r) and e_f < 1e-3 and e_w < 1e-3
        assert np.isfinite(gw_g).all() and np.abs(gw_g[wv == 0.0]).max() > 0.0   # (weight 0: the sampled upstream image, not a skipped slot)
