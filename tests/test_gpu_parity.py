"""GPU parity tests proper: the HIP path (through the C ABI, via the plugin surface) against the CPU
oracle on the same seeded inputs, and against the golden fixtures captured from the reference.

Tolerances (stated per assertion):
  * warped events: BIT-EXACT on equal dtype (elementwise, same op order, no FMA contraction)
  * images, fp64 kernels: rel-L2 <= 1e-12 (float atomics reorder the sum)
  * images, fp32 fused path vs the fp64 reference: rel-L2 < 1e-4 (north_star), typically ~1e-6
  * costs: rel < 1e-5;  gradients (fp32 atomics): rel-L2 < 1e-3
"""
import os

import numpy as np
import pytest
import torch

from oracle import ebos_oracle as O

from _kinks import off_the_kinks as _off_the_kinks, off_the_kinks_patch as _off_the_kinks_patch

pytestmark = pytest.mark.gpu

H, W = 24, 32
DIRS = ["first", "middle", "last", 0.25, "before", "after"]


@pytest.fixture(scope="module")
def ebos():
    import event_based_bos_amd as pkg

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    pkg.load_library()
    return pkg


def dev():
    return torch.device("cuda:0")


def G(a, dtype=None):
    t = torch.from_numpy(np.asarray(a)).to(dev())
    return t if dtype is None else t.to(dtype)


def rel(a, b):
    return O.rel_l2(a, b)


# ------------------------------------------------------------------------------ warp
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("d", DIRS)
def test_warp_dense_bit_exact_vs_golden(ebos, golden_small, norm, d):
    g = golden_small
    ev, fl = g["g2_events"], g["g2_flow"]
    wp = ebos.Warp((H, W), normalize_t=norm)
    tag = f"g2_warp_dense_n{int(norm)}_{d}"
    out_np, feat = wp.warp_event(ev, fl, "dense-flow", d)  # numpy in -> numpy out (staged through the GPU)
    assert isinstance(out_np, np.ndarray) and out_np.dtype == np.float64
    np.testing.assert_array_equal(out_np, g[tag + "_numpy"])
    out_t, _ = wp.warp_event(G(ev), G(fl), "dense-flow", d)  # GPU tensors
    assert out_t.is_cuda
    np.testing.assert_array_equal(out_t.cpu().numpy(), g[tag + "_torch"])
    assert set(feat) == {"determinant", "trace", "divergence", "straint", "absement"}
    assert all(v["value"] is None for v in feat.values())


def test_warp_misc_vs_golden(ebos, golden_small):
    g = golden_small
    ev, fl = g["g2_events"], g["g2_flow"]
    th = np.array([3.0, -2.0])
    for norm in (False, True):
        wp = ebos.Warp((H, W), normalize_t=norm)
        np.testing.assert_array_equal(wp.warp_event(ev, th, "2d-translation", "middle")[0],
                                      g[f"g2_warp_2dof_n{int(norm)}_middle_numpy"])
        np.testing.assert_array_equal(wp.warp_event(G(ev), G(th), "rigid-optical-flow", "first")[0].cpu().numpy(),
                                      g[f"g2_warp_2dof_n{int(norm)}_first_torch"])
    wp = ebos.Warp((H, W), normalize_t=True)
    # float32 tensors: bit-exact with the reference's float32 run
    out32 = wp.warp_event(G(ev).float(), G(fl).float(), "dense-flow", "first")[0]
    assert out32.dtype == torch.float32
    np.testing.assert_array_equal(out32.cpu().numpy(), g["g2_warp_dense_n1_first_torch_f32"])
    # batched
    eb, fb = g["g2_events_b"], g["g2_flow_b"]
    np.testing.assert_array_equal(wp.warp_event(eb, fb, "dense-flow", "middle")[0], g["g2_warp_dense_b_n1_middle_numpy"])
    np.testing.assert_array_equal(wp.warp_event(G(eb), G(fb), "dense-flow", "middle")[0].cpu().numpy(),
                                  g["g2_warp_dense_b_n1_middle_torch"])
    # CPU tensors are staged through the GPU and come back as CPU tensors
    out_c = wp.warp_event(torch.from_numpy(eb), torch.from_numpy(fb), "dense-flow", "middle")[0]
    assert not out_c.is_cuda
    np.testing.assert_array_equal(out_c.numpy(), g["g2_warp_dense_b_n1_middle_torch"])


def test_warp_micro_known_answers(ebos, golden_small):
    g = golden_small
    ev, fl = g["g1_warp_events"], g["g1_warp_flow"]
    wp = ebos.Warp((4, 5))
    for d in ["first", "middle", "last", "before", "after"]:
        np.testing.assert_array_equal(wp.warp_event(ev, fl, "dense-flow", d)[0], g[f"g1_warp_dense_{d}_numpy"])
    np.testing.assert_array_equal(wp.warp_event(ev, fl, "dense-flow", 0.25)[0], g["g1_warp_dense_f025_numpy"])
    np.testing.assert_array_equal(wp.warp_event(ev, np.array([1.0, 2.0]), "2d-translation", "first")[0],
                                  g["g1_warp_2dof_first_numpy"])
    np.testing.assert_array_equal(wp.warp_event(g["g1_warp_frac_events"], g["g1_warp_frac_flow"], "dense-flow")[0],
                                  g["g1_warp_frac_numpy"])
    np.testing.assert_allclose(wp.get_flow_from_motion(np.array([1.0, 2.0]), "2d-translation"), g["g1_flow_from_motion"], atol=0)
    assert wp.get_motion_vector_size("2d-translation") == 2
    assert wp.motion_model_from_motion(np.array([1.0, 2.0]), "2d-translation") == {"trans_x": 1.0, "trans_y": 2.0}


def test_warp_errors(ebos):
    wp = ebos.Warp((4, 5))
    ev = O.synth_events(10, 4, 5)
    fl = np.zeros((2, 4, 5))
    for bad in (1, np.float64(0.5), "sideways"):
        with pytest.raises(ValueError):
            wp.warp_event(ev, fl, "dense-flow", bad)
    with pytest.raises(ebos.MotionModelKeyError):
        wp.warp_event(ev, fl, "affine")
    with pytest.raises(ebos.MotionModelKeyError):
        wp.get_key_names("nope")
    with pytest.raises(AssertionError):
        wp.warp_event(ev, np.zeros(3), "2d-translation")
    bad_ev = ev.copy()
    bad_ev[3, 0] = 100.0  # source pixel outside the flow field: torch.gather raises in the reference
    with pytest.raises(IndexError):
        wp.warp_event(bad_ev, fl, "dense-flow")
    # GPU tensors: the same error, deferred -- the count stays on the device and surfaces without a sync in the loop
    # (check_out_of_range() is the blocking form; a later call raises once the read-back has landed)
    wp2 = ebos.Warp((4, 5))
    wp2.warp_event(G(bad_ev), G(fl), "dense-flow")
    with pytest.raises(IndexError):
        wp2.check_out_of_range()
    wp2.warp_event(G(ev), G(fl), "dense-flow")
    wp2.check_out_of_range()  # the counter was reset by the raise; in-range events leave it at zero
    wp3 = ebos.Warp((4, 5))
    wp3.warp_event(G(bad_ev), G(fl), "dense-flow")
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        wp3.warp_event(G(ev), G(fl), "dense-flow")  # found by the next call, no explicit check
    wp4 = ebos.Warp((4, 5), strict=True)
    with pytest.raises(IndexError):
        wp4.warp_event(G(bad_ev), G(fl), "dense-flow")
    wp5 = ebos.Warp((4, 5), strict=False)
    wp5.warp_event(G(bad_ev), G(fl), "dense-flow")
    wp5.check_out_of_range()


def test_reftime_and_dt(ebos, golden_small):
    g = golden_small
    ev = g["g2_events"]
    wp = ebos.Warp((H, W), normalize_t=True)
    for d in DIRS:
        assert wp.calculate_reftime(ev, d) == O.reference_time(ev, d)
        assert wp.calculate_reftime(G(ev), d).item() == O.reference_time(ev, d)
    ref = O.reference_time(ev, "middle")
    np.testing.assert_array_equal(wp.calculate_dt(ev, ref), O.delta_t(ev, ref, True))
    eb = g["g2_events_b"]
    np.testing.assert_array_equal(wp.calculate_reftime(eb, 0.25), O.reference_time(eb, 0.25))
    out, _ = wp.warp_event_from_optical_flow(ev, g["g2_flow"], ref)
    np.testing.assert_array_equal(out, g["g2_warp_dense_n1_middle_numpy"])


# ------------------------------------------------------------------------------ images
@pytest.mark.parametrize("pad", [0, 2])
def test_iwe_vs_golden(ebos, golden_small, pad):
    g = golden_small
    warped, wgt = g["g2_warp_dense_n1_first_numpy"], g["g2_weight"]
    ic = ebos.EventImageConverter((H, W), outer_padding=pad)
    assert ic.image_size == (H + 2 * pad, W + 2 * pad)
    tol = 1e-12
    assert rel(ic.bilinear_vote_numpy(warped), g[f"g2_iwe_p{pad}_numpy"]) < tol
    assert rel(ic.bilinear_vote_tensor(G(warped)).cpu().numpy(), g[f"g2_iwe_p{pad}_torch"]) < tol
    assert rel(ic.bilinear_vote_numpy(warped, weight=wgt), g[f"g2_iwe_p{pad}_w_numpy"]) < tol
    assert rel(ic.bilinear_vote_tensor(G(warped), weight=G(wgt)).cpu().numpy(), g[f"g2_iwe_p{pad}_w_torch"]) < tol
    assert rel(ic.bilinear_vote_tensor(G(warped), weight=0.5).cpu().numpy(), g[f"g2_iwe_p{pad}_w05_torch"]) < tol
    np.testing.assert_array_equal(ic.count_event_numpy(warped), g[f"g2_count_p{pad}_numpy"])
    np.testing.assert_array_equal(ic.count_event_tensor(G(warped)).cpu().numpy(), g[f"g2_count_p{pad}_numpy"])
    assert rel(ic.create_iwe(warped, method="polarity", sigma=0), g[f"g2_polarity_p{pad}_numpy"]) < tol
    np.testing.assert_array_equal(ic.create_eventmask(warped), g[f"g2_mask_p{pad}_numpy"])


def test_iwe_variants_vs_golden(ebos, golden_small):
    g = golden_small
    warped = g["g2_warp_dense_n1_first_numpy"]
    ic = ebos.EventImageConverter((H, W))
    out32 = ic.bilinear_vote_tensor(G(warped).float())
    assert out32.dtype == torch.float32
    assert rel(out32.cpu().numpy(), g["g2_iwe_f32_torch"]) < 1e-6
    for s in (1, 3):
        assert rel(ic.create_iwe(warped, method="bilinear_vote", sigma=s), g[f"g2_iwe_sigma{s}_numpy"]) < 1e-12
    assert rel(ic.create_iwe(warped), g["g2_iwe_default_numpy"]) < 1e-12  # default sigma = 1 for numpy
    assert rel(ic.create_iwe(G(warped), sigma=0).cpu().numpy(), g["g2_iwe_default_torch"]) < 1e-12
    wb = g["g2_warp_dense_b_n1_middle_numpy"]
    assert rel(ic.bilinear_vote_numpy(wb), g["g2_iwe_b_numpy"]) < 1e-12
    assert rel(ic.bilinear_vote_tensor(G(wb)).cpu().numpy(), g["g2_iwe_b_torch"]) < 1e-12
    with pytest.raises(NotImplementedError):
        ic.create_iwe(warped, method="nope")
    with pytest.raises(NotImplementedError):
        ic.create_iwe(G(warped), method="nope")
    with pytest.raises(RuntimeError):
        ic.create_iwe([1, 2, 3])
    # torch blur (3 taps, reflect) against the oracle's restatement
    blurred = ic.create_image_from_events_tensor(G(warped), sigma=1)
    expect = O.gaussian_blur3_torch(torch.from_numpy(g["g2_iwe_default_torch"]), 1.0)
    assert rel(blurred.cpu().numpy(), expect.numpy()) < 1e-12


def test_derived_images_vs_golden(ebos, golden_small):
    """A12 through the plugin surface: averaged (iwa / iwd / iwt) and weighted (timeimage / probability) images."""
    g = golden_small
    warped, val = g["g2_warp_dense_n1_first_numpy"], g["g2_derived_values"]
    ic = ebos.EventImageConverter((H, W))
    fns = {"iwa": ic.create_iwa, "iwd": ic.create_iwd, "iwt": ic.create_iwt, "timeimage": ic.create_timeimage,
           "prob": ic.create_probability_iwe}
    for name, fn in fns.items():
        for s in (0, 1):
            out = fn(warped, val, sigma=s)
            assert isinstance(out, np.ndarray)
            assert rel(out, g[f"g2_{name}_s{s}_numpy"]) < 1e-12, (name, s)
        out_t = fn(G(warped), G(val), sigma=0)
        assert out_t.is_cuda and tuple(out_t.shape) == g[f"g2_{name}_s0_torch"].shape
        assert rel(out_t.cpu().numpy(), g[f"g2_{name}_s0_torch"]) < 1e-12, name
    assert ic.create_iat(warped, val, 0) is None
    with pytest.raises(RuntimeError):
        ic.create_timeimage([1, 2], val)
    # create_eventrate against the reference's per-event loop (src/event_image_converter.py:304-327), restated inline
    ev = g["g2_events"].copy()
    ev[:, :2] = np.floor(ev[:, :2])
    ev[::7, 2] = ev[1::7, 2][: len(ev[::7])]  # some equal timestamps: dt == 0 pairs are ignored
    rate, last = np.zeros((H, W)), np.full((H, W), np.inf)
    for e in ev:
        r, c = int(e[0]), int(e[1])
        d = e[2] - last[r, c]
        if d > 0:
            rate[r, c] = max(rate[r, c], 1.0 / d)
        last[r, c] = e[2]
    out = ic.create_eventrate(ev)
    np.testing.assert_allclose(out, rate, rtol=1e-12)
    with pytest.raises(RuntimeError):
        ic.create_eventrate(G(ev))


def test_micro_vote_known_answer(ebos, golden_small):
    g = golden_small
    ev = g["g1_events"]
    ic = ebos.EventImageConverter((4, 5))
    np.testing.assert_allclose(ic.bilinear_vote_tensor(G(ev)).cpu().numpy(), g["g1_vote_torch"], atol=1e-15)
    np.testing.assert_allclose(ic.bilinear_vote_numpy(ev), g["g1_vote_numpy"], atol=1e-15)
    np.testing.assert_array_equal(ic.count_event_numpy(ev), g["g1_count_numpy"])
    np.testing.assert_allclose(ic.bilinear_vote_numpy(ev, weight=np.arange(6.0)), g["g1_vote_weighted_numpy"], atol=1e-15)
    ic2 = ebos.EventImageConverter((4, 5), outer_padding=2)
    np.testing.assert_allclose(ic2.bilinear_vote_tensor(G(ev)).cpu().numpy(), g["g1_vote_pad2_torch"], atol=1e-15)
    np.testing.assert_array_equal(ic.create_eventmask(G(ev)).cpu().numpy(), g["g1_mask_torch"])
    # empty and single-event inputs
    assert ic.bilinear_vote_numpy(np.zeros((0, 4))).sum() == 0
    one = ic.bilinear_vote_numpy(np.array([[1.5, 2.5, 0, 1.0]]))
    assert abs(one.sum() - 1.0) < 1e-15 and one[1, 2] == 0.25


# ------------------------------------------------------------------------------ costs + autograd through the API path
def _cost(ebos, name):
    return ebos.costs.functions[name](direction="minimize")


@pytest.mark.parametrize("cost", ["var", "gm"])
@pytest.mark.parametrize("omit", [False, True])
def test_api_path_costs_and_gradients_fp64(ebos, golden_small, cost, omit):
    g = golden_small
    tag = f"g2_{cost}_omit{int(omit)}"
    name = {"var": "image_variance", "gm": "gradient_magnitude"}[cost]
    wp, ic = ebos.Warp((H, W), normalize_t=True), ebos.EventImageConverter((H, W))
    ev = G(g["g2_events"])
    fl = G(g["g2_flow"]).requires_grad_(True)
    wt = G(g["g2_weight"]).requires_grad_(True)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "first")
    iwe = ic.bilinear_vote_tensor(warped, weight=wt)
    loss = _cost(ebos, name).calculate({"iwe": iwe, "omit_boundary": omit})
    loss.backward()
    assert abs(loss.item() - g[tag + "_loss"]) <= 1e-12 * abs(g[tag + "_loss"])
    assert rel(fl.grad.cpu().numpy(), g[tag + "_dflow"]) < 1e-10
    assert rel(wt.grad.cpu().numpy(), g[tag + "_dweight"]) < 1e-10
    th = torch.tensor([3.0, -2.0], dtype=torch.float64, device=dev(), requires_grad=True)
    w2, _ = wp.warp_event(ev, th, "2d-translation", "first")
    loss2 = _cost(ebos, name).calculate({"iwe": ic.bilinear_vote_tensor(w2), "omit_boundary": omit})
    loss2.backward()
    assert abs(loss2.item() - g[tag + "_2dof_loss"]) <= 1e-12 * abs(g[tag + "_2dof_loss"])
    np.testing.assert_allclose(th.grad.cpu().numpy(), g[tag + "_2dof_dtheta"], rtol=1e-9)
    # the cost kernels in isolation: value and d(loss)/d(iwe)
    x = G(g["g2_iwe_p0_torch"]).requires_grad_(True)
    _cost(ebos, name).calculate({"iwe": x, "omit_boundary": omit}).backward()
    assert rel(x.grad.cpu().numpy(), g[tag + "_diwe"]) < 1e-12
    # numpy in -> python float out
    v = _cost(ebos, name).calculate({"iwe": g["g2_iwe_p0_torch"], "omit_boundary": omit})
    assert isinstance(v, float)


def test_cost_plugin_surface(ebos, golden_small):
    g = golden_small
    c = ebos.costs
    assert sorted(c.functions) == ["diff_norm", "flow_norm", "flow_norm_pxy", "gradient_magnitude", "image_gradient",
                                   "image_variance"]
    with pytest.raises(ValueError):
        c.functions["image_variance"](direction="sideways")
    cost = c.functions["image_variance"](direction="maximize", store_history=True)
    with pytest.raises(KeyError):
        cost.calculate({"iwe": G(g["g2_iwe_p0_torch"])})
    v = cost.calculate({"iwe": G(g["g2_iwe_p0_torch"]), "omit_boundary": False})
    assert v.item() > 0 and cost.get_history()["loss"] == [v.item()]
    fl = G(g["g2_flow"])
    wmap = G(g["g2_cost_weights"])
    assert abs(c.functions["flow_norm"]().calculate({"flow": fl}).item() - g["g2_cost_flow_norm"]) < 1e-12
    assert abs(c.functions["image_gradient"]().calculate({"flow": fl, "omit_boundary": False, "weights": wmap}).item()
               - g["g2_cost_image_gradient"]) < 1e-12
    dn = c.functions["diff_norm"]().calculate({"prediction": G(g["g2_iwe_p0_torch"]), "measurement": G(g["g2_iwe_p0_w_torch"]),
                                               "weights": None})
    assert abs(dn.item() - g["g2_cost_diff_norm"]) < 1e-10
    hy = c.HybridCost("minimize", {"flow_norm": 0.5, "image_gradient": "inv"}, store_history=True)
    v = hy.calculate({"flow": fl, "omit_boundary": False, "weights": wmap})
    assert abs(v.item() - g["g2_cost_hybrid"]) < 1e-11
    assert sorted(hy.get_history().keys()) == ["flow_norm", "image_gradient", "loss"]
    hy2 = c.HybridCost("minimize", {"image_variance": 1.0, "flow_norm": 0.1})
    assert set(hy2.required_keys) == {"iwe", "omit_boundary", "flow"}


# ------------------------------------------------------------------------------ fused hot path (f32 SoA plan)
def _oracle_objective(ev, fl, size, cost, omit=False, pad=(0, 0), weight=None):
    f = torch.from_numpy(fl).clone().requires_grad_(True)
    w = 1.0 if weight is None else torch.from_numpy(weight).clone().requires_grad_(True)
    iwe = O.iwe_dense(torch.from_numpy(ev), f, size, pad=pad, weight=w)
    L = O.image_variance(iwe, omit) if cost == "var" else O.gradient_magnitude(iwe, omit)
    L.backward()
    return iwe.detach().numpy(), L.item(), f.grad.numpy(), (None if weight is None else w.grad.numpy())


@pytest.mark.parametrize("tile,halo", [((64, 64), 32), ((32, 64), 32), ((32, 32), 8), ((64, 64), None), (None, None)])
def test_fused_dense_forward_backward(ebos, tile, halo):
    h, w, n = 130, 173, 60_000
    ev = O.synth_events(n, h, w, seed=3)
    fl = O.synth_dense_flow(h, w, seed=4, max_val=12.0)  # exceeds halo 8: exercises the global-atomic fallback
    iwe_ref, loss_ref, dflow_ref, _ = _oracle_objective(ev, fl, (h, w), "var")
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=tile)
    assert plan.n == n and plan.binned == (tile is not None)
    flow = G(fl, torch.float32).requires_grad_(True)
    iwe = plan.iwe_dense(flow, halo=halo)
    loss = -ebos.ops.image_variance(iwe)
    loss.backward()
    assert rel(iwe.detach().cpu().numpy(), iwe_ref) < 1e-5  # bar: 1e-4
    assert abs(loss.item() - loss_ref) < 1e-5 * abs(loss_ref)
    assert rel(flow.grad.cpu().numpy(), dflow_ref) < 1e-3
    # fused objective (variance gradient folded into the backward event kernel)
    f2 = G(fl, torch.float32).requires_grad_(True)
    l2 = -plan.contrast_dense(f2, "image_variance", halo=halo)
    l2.backward()
    assert abs(l2.item() - loss_ref) < 1e-5 * abs(loss_ref)
    assert rel(f2.grad.cpu().numpy(), dflow_ref) < 1e-3


@pytest.mark.parametrize("cost", ["var", "gm"])
@pytest.mark.parametrize("omit", [False, True])
def test_fused_dense_costs_padding_weights(ebos, cost, omit):
    h, w, n = 64, 80, 20_000
    ev = O.synth_events(n, h, w, seed=5)
    ev[:, 0] += np.random.RandomState(6).uniform(0, 0.99, n) * (np.arange(n) % 2 == 0)
    fl = O.synth_dense_flow(h, w, seed=7, max_val=6.0)
    wgt = np.random.RandomState(8).uniform(0.2, 1.8, n)
    pad = (3, 3)
    iwe_ref, loss_ref, dflow_ref, dw_ref = _oracle_objective(ev, fl, (h, w), cost, omit, pad, wgt)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    flow = G(fl, torch.float32).requires_grad_(True)
    wt = G(wgt, torch.float32).requires_grad_(True)
    iwe = plan.iwe_dense(flow, pad=pad, weight=wt, halo=16)
    assert iwe.shape == (h + 6, w + 6)
    fn = ebos.ops.image_variance if cost == "var" else ebos.ops.gradient_magnitude
    loss = -fn(iwe, omit)
    loss.backward()
    assert rel(iwe.detach().cpu().numpy(), iwe_ref) < 1e-5
    assert abs(loss.item() - loss_ref) < 1e-5 * abs(loss_ref)
    assert rel(flow.grad.cpu().numpy(), dflow_ref) < 1e-3
    assert rel(wt.grad.cpu().numpy(), dw_ref) < 1e-3


def test_objective_backward_without_the_engine_keeps_autograd_semantics(ebos):
    """``plan.contrast_dense(flow).backward()`` on a plain float32 leaf stores the gradient that came with the value without
    entering the autograd engine (event_plan._EagerLoss).  Everything observable must be what the engine would have done: the same
    gradient bit for bit, accumulation into an existing ``.grad``, scaling by ``gradient=``, graphs built on top of the result,
    tensor hooks (which only the engine can fire), value reads, and an error on a second backward."""
    h, w, n = 96, 128, 30_000
    ev = O.synth_events(n, h, w, seed=5)
    fl = G(O.synth_dense_flow(h, w, seed=6, max_val=6.0), torch.float32)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    from event_based_bos_amd.event_plan import _EagerLoss, _FusedVarianceDense

    def engine(scale=1.0):  # the ordinary autograd node, as any non-leaf / hooked / double flow gets it
        f = fl.clone().requires_grad_(True)
        v = _FusedVarianceDense.apply(f, plan, (0, 0), False, 32, plan.resolve_splits(None))
        (v * scale).backward()
        return v.detach(), f.grad

    v_ref, g_ref = engine()
    f = fl.clone().requires_grad_(True)
    loss = plan.contrast_dense(f)
    assert type(loss) is _EagerLoss and loss._ebos[2] is None
    assert loss.item() == v_ref.item() and loss._ebos[2] is None and f"{loss:.3e}" == f"{v_ref.item():.3e}"
    loss.backward()
    assert loss._ebos[2] is None                                          # no autograd node was ever made
    assert torch.equal(f.grad, g_ref)
    with pytest.raises(RuntimeError):
        loss.backward()                                                   # consumed, as a freed graph would be
    plan.contrast_dense(f).backward()                                     # a second evaluation accumulates
    assert torch.equal(f.grad, g_ref + g_ref)
    f.grad = None
    plan.contrast_dense(f).backward(gradient=torch.tensor(-0.5, device=f.device))
    assert torch.equal(f.grad, g_ref * -0.5)
    f.grad = None
    neg = -plan.contrast_dense(f)                                         # the sign / weight a caller applies: still no engine
    assert type(neg) is _EagerLoss and neg.item() == -v_ref.item()
    neg.backward()
    assert neg._ebos[2] is None and torch.equal(f.grad, -g_ref)
    f.grad = None
    (0.25 * plan.contrast_dense(f) / 2.0).backward(retain_graph=True)
    assert torch.equal(f.grad, g_ref * 0.125)
    # used in a graph: the node is attached on first use, gradients flow through the composite
    f2 = fl.clone().requires_grad_(True)
    total = -plan.contrast_dense(f2) * 2.0 + (f2 * f2).sum() * 1e-3
    assert type(total) is torch.Tensor and total.grad_fn is not None
    total.backward()
    assert rel(f2.grad.cpu().numpy(), (g_ref * -2.0 + 2e-3 * fl).cpu().numpy()) < 1e-6
    l3 = plan.contrast_dense(fl.clone().requires_grad_(True))
    assert l3.requires_grad and l3.grad_fn is not None and l3.shape == torch.Size([])
    stacked = torch.stack([plan.contrast_dense(f2), plan.contrast_dense(f2)])
    assert stacked.grad_fn is not None
    # a tensor hook on the flow: only the engine fires it -- the short cut must step aside
    f3 = fl.clone().requires_grad_(True)
    seen = []
    f3.register_hook(lambda g: seen.append(1))
    l = plan.contrast_dense(f3)
    assert type(l) is torch.Tensor
    l.backward()
    assert seen == [1] and torch.equal(f3.grad, g_ref)
    with torch.no_grad():
        assert not plan.contrast_dense(f).requires_grad
    # the functional entry points: autograd.grad dispatches through __torch_function__; autograd.backward (the function) does not
    # consult it and must fail loudly on the bare result, never run silently past the flow
    f4 = fl.clone().requires_grad_(True)
    (g4,) = torch.autograd.grad(plan.contrast_dense(f4), f4)
    assert torch.equal(g4, g_ref) and f4.grad is None
    with pytest.raises(RuntimeError):
        torch.autograd.backward([plan.contrast_dense(f4)])
    torch.autograd.backward([plan.contrast_dense(f4) + 0.0])             # (any expression that is not a bare sign / weight attaches the node)
    assert torch.equal(f4.grad, g_ref)


def test_reference_idiom_is_fused_lazily(ebos, monkeypatch):
    """warp_event + create_iwe, written exactly as against the reference, run on the fused kernels for float32
    GPU tensors: same image and flow gradient as the unfused path, plan built once per event window."""
    h, w, n = 120, 160, 80_000
    ev = G(O.synth_events(n, h, w, seed=41), torch.float32)
    fl_np = O.synth_dense_flow(h, w, seed=42, max_val=9.0)
    wp, ic = ebos.Warp((h, w), normalize_t=True), ebos.EventImageConverter((h, w), outer_padding=2)
    cost = ebos.costs.functions["image_variance"]()

    def run():
        fl = G(fl_np, torch.float32).requires_grad_(True)
        warped, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
        iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
        loss = cost.calculate({"iwe": iwe, "omit_boundary": False})
        loss.backward()
        return warped, iwe.detach(), loss.item(), fl.grad

    ebos.fusion.clear_cache()
    before = dict(ebos.fusion.stats)
    monkeypatch.setenv("EBOS_FUSE_API", "off")
    w0, iwe0, l0, g0 = run()
    assert ebos.fusion.stats == before
    monkeypatch.setenv("EBOS_FUSE_API", "f32")
    for _ in range(3):
        w1, iwe1, l1, g1 = run()
    assert ebos.fusion.stats["plan_builds"] == before["plan_builds"] + 1
    assert ebos.fusion.stats["plan_hits"] == before["plan_hits"] + 2
    assert ebos.fusion.stats["fused_images"] == before["fused_images"] + 3
    assert torch.equal(w0, w1)  # the materialised warped events are the same public result
    assert iwe1.shape == iwe0.shape == (h + 4, w + 4)
    assert rel(iwe1.cpu().numpy(), iwe0.cpu().numpy()) < 1e-5
    assert abs(l1 - l0) < 1e-5 * abs(l0)
    assert rel(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-3
    # the third step -- the variance cost of that very image -- is the objective's one native call (fusion.fused_variance):
    # no autograd engine in `loss.backward()`; an image that was touched, or whose own gradient is wanted, steps aside
    from event_based_bos_amd.event_plan import _EagerLoss
    assert ebos.fusion.stats.get("fused_costs", 0) >= 3
    fl = G(fl_np, torch.float32).requires_grad_(True)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    loss = cost.calculate({"iwe": iwe, "omit_boundary": False})           # direction "minimize": the negated contrast
    assert type(loss) is _EagerLoss and abs(loss.item() - l0) < 1e-5 * abs(l0)
    hy = ebos.costs.HybridCost("minimize", {"image_variance": 2.0, "flow_norm": 0.1})
    total = hy.calculate({"iwe": iwe, "omit_boundary": False, "flow": fl})  # combined with another term: an ordinary graph
    total.backward()
    fl_ref = G(fl_np, torch.float32).requires_grad_(True)
    monkeypatch.setenv("EBOS_FUSE_API", "off")
    w_ref, _ = wp.warp_event(ev, fl_ref, "dense-flow", "middle")
    hy.calculate({"iwe": ic.create_iwe(w_ref, "bilinear_vote", sigma=0), "omit_boundary": False, "flow": fl_ref}).backward()
    monkeypatch.setenv("EBOS_FUSE_API", "f32")
    assert rel(fl.grad.cpu().numpy(), fl_ref.grad.cpu().numpy()) < 1e-3
    n_costs = ebos.fusion.stats["fused_costs"]
    fl = G(fl_np, torch.float32).requires_grad_(True)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    iwe.retain_grad()                                                      # the caller wants d loss / d image: the engine's job
    loss = cost.calculate({"iwe": iwe, "omit_boundary": False})
    loss.backward()
    assert type(loss) is torch.Tensor and iwe.grad is not None and ebos.fusion.stats["fused_costs"] == n_costs
    assert rel(fl.grad.cpu().numpy(), g0.cpu().numpy()) < 1e-3
    fl = G(fl_np, torch.float32).requires_grad_(True)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    with torch.no_grad():
        iwe[5, 5] += 100.0                                                # modified in place: no longer the fused image
    l_mod = cost.calculate({"iwe": iwe, "omit_boundary": False})
    assert ebos.fusion.stats["fused_costs"] == n_costs and abs(l_mod.item() - l0) > 1e-6 * abs(l0)
    # EBOS_FUSE_API=lazy (the default): warped events and image are only computed if something reads them; the cost step does not
    monkeypatch.setenv("EBOS_FUSE_API", "lazy")
    n_img = ebos.fusion.stats["fused_images"]
    fl = G(fl_np, torch.float32).requires_grad_(True)
    w2, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    i2 = ic.create_iwe(w2, "bilinear_vote", sigma=0)
    assert type(w2) is ebos.fusion.LazyWarped and type(i2) is ebos.fusion.LazyIwe and i2.shape == iwe1.shape and i2.dtype == iwe1.dtype
    loss2 = cost.calculate({"iwe": i2, "omit_boundary": False})
    loss2.backward()
    assert not w2.computed and not i2.computed and ebos.fusion.stats["fused_images"] == n_img   # the loop looked at neither
    assert w2.shape == w0.shape and w2.dtype == w0.dtype and w2.device == w0.device and not w2.computed
    assert loss2.item() == l1 and torch.equal(fl.grad, g1)
    assert torch.equal(i2.detach(), iwe1) and i2.computed and not w2.computed                   # read: the same fused image
    w2, iwe2, l2, g2 = run()
    assert torch.equal(iwe2, iwe1) and l2 == l1 and torch.equal(g2, g1)
    fl = G(fl_np, torch.float32).requires_grad_(True)
    wl, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    assert torch.equal(wl, w0) and wl._ebos_lazy[1] is not None              # read: computed, the same public result
    assert torch.equal(ic.create_iwe(wl, "bilinear_vote", sigma=0).detach(), iwe1)  # ... and still the fused image
    wl[:, 0] += 1.0                                                           # modified after it was read: splatted as given
    n_fused = ebos.fusion.stats["fused_images"]
    shifted = ic.create_iwe(wl, "bilinear_vote", sigma=0)
    assert ebos.fusion.stats["fused_images"] == n_fused
    assert rel(shifted[3:-1, 2:-2].detach().cpu().numpy(), iwe0[2:-2, 2:-2].cpu().numpy()) < 1e-5
    fl = G(fl_np, torch.float32).requires_grad_(True)
    wl, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    il = ic.create_iwe(wl, "bilinear_vote", sigma=0)
    with torch.no_grad():
        fl.mul_(0.5)                                                          # an optimiser step before anyone read them
    assert torch.equal(wl, w0) and torch.equal(il, iwe1)                      # ... and they hold the OLD flow's values
    ev2 = ev.clone()
    wl, _ = wp.warp_event(ev2, fl, "dense-flow", "middle")
    ev2[:, 2] += 1.0                                                          # the EVENTS modified before the first read: not copied
    with pytest.raises(RuntimeError, match="EVENTS were modified"):
        wl.sum()
    monkeypatch.setenv("EBOS_FUSE_API", "f32")
    # a modified copy of the warped events loses the provenance and is splatted as given
    fl = G(fl_np, torch.float32)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "middle")
    warped[:, 0] += 1.0
    n_fused = ebos.fusion.stats["fused_images"]
    shifted = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    assert ebos.fusion.stats["fused_images"] == n_fused
    assert rel(shifted[3:-1, 2:-2].cpu().numpy(), iwe0[2:-2, 2:-2].cpu().numpy()) < 1e-5
    # float64 callers keep the float64 kernels unless EBOS_FUSE_API=all
    w64, _ = wp.warp_event(ev.double(), fl.double(), "dense-flow", "middle")
    assert not hasattr(w64, "_ebos_provenance")


def test_fusion_plan_cache_is_bound_to_the_tensor_not_its_address(ebos, monkeypatch):
    """A per-window loop frees its events tensor and the caching allocator hands the same address (same shape, version 0)
    to a later window: the plan cache must not serve the earlier window's plan (address reuse / ABA)."""
    monkeypatch.setenv("EBOS_FUSE_API", "f32")
    ebos.fusion.clear_cache()
    h, w, n = 96, 128, 30_000
    fl_np = O.synth_dense_flow(h, w, seed=3, max_val=5.0)
    wp, ic = ebos.Warp((h, w), normalize_t=True), ebos.EventImageConverter((h, w))
    fl = G(fl_np, torch.float32)
    ptrs, reused = [], 0
    for seed in range(6):
        ev_np = O.synth_events(n, h, w, seed=100 + seed)
        ev = torch.from_numpy(ev_np).float().to(dev())  # fresh tensor per window, freed at the end of the iteration
        reused += ev.data_ptr() in ptrs
        ptrs.append(ev.data_ptr())
        warped, _ = wp.warp_event(ev, fl, "dense-flow", "first")
        n_fused = ebos.fusion.stats["fused_images"]
        iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
        assert ebos.fusion.stats["fused_images"] == n_fused + 1
        ref = O.iwe_dense(torch.from_numpy(ev_np), torch.from_numpy(fl_np), (h, w)).numpy()
        assert rel(iwe.cpu().numpy(), ref) < 1e-5, f"window {seed}: image of another window's events"
        del ev, warped, iwe
    assert reused >= 1, "the allocator never reused an address: the test did not exercise the hazard"
    assert len(ebos.fusion._plans) <= 1  # entries die with their tensors


def test_fusion_sees_in_place_flow_updates(ebos, monkeypatch):
    """A flow modified in place between warp_event and create_iwe: the image must come from the materialised (old-flow)
    coordinates, like the reference's, not from the fused kernels reading the new flow."""
    monkeypatch.setenv("EBOS_FUSE_API", "f32")
    h, w, n = 96, 128, 30_000
    ev_np = O.synth_events(n, h, w, seed=9)
    fl_np = O.synth_dense_flow(h, w, seed=10, max_val=5.0)
    wp, ic = ebos.Warp((h, w), normalize_t=True), ebos.EventImageConverter((h, w))
    ev, fl = G(ev_np, torch.float32), G(fl_np, torch.float32)
    warped, _ = wp.warp_event(ev, fl, "dense-flow", "first")
    fl.mul_(-1.0)  # e.g. optimizer.step() / clamp_()
    n_fused = ebos.fusion.stats["fused_images"]
    iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    assert ebos.fusion.stats["fused_images"] == n_fused
    ref = O.iwe_dense(torch.from_numpy(ev_np), torch.from_numpy(fl_np), (h, w)).numpy()
    assert rel(iwe.cpu().numpy(), ref) < 1e-5


def test_fixed_point_tile_overflow_falls_back_exactly(ebos):
    """The tile-private forward accumulates unit-weight events in verified fixed point (2048 units of weight
    per LDS cell per workgroup).  A hot pixel beyond that must be detected and redone in f64."""
    h, w = 64, 64
    n_hot = 20_000
    ev = O.synth_events(30_000, h, w, seed=21)
    hot = np.tile(np.array([[10.0, 12.0, 0.0, 1.0]]), (n_hot, 1))
    hot[:, 2] = np.linspace(0.0, 0.5, n_hot)
    ev = np.concatenate([ev, hot], 0)
    ev = ev[np.argsort(ev[:, 2], kind="stable")]
    fl = np.zeros((2, h, w))  # zero flow: all 20k hot events land on pixel (10, 12) with weight 1
    fl[:, 40:, :] = O.synth_dense_flow(h, w, seed=22, max_val=3.0)[:, 40:, :]
    ref = O.iwe_dense(torch.from_numpy(ev), torch.from_numpy(fl), (h, w)).numpy()
    assert ref[10, 12] > 20_000
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    iwe = plan.iwe_dense(G(fl, torch.float32), halo=16)
    assert rel(iwe.cpu().numpy(), ref) < 1e-6
    assert abs(iwe[10, 12].item() - ref[10, 12]) < 1e-2
    # determinism of the fixed-point path: two runs are bit-identical
    ev2 = O.synth_events(40_000, h, w, seed=23)
    fl2 = O.synth_dense_flow(h, w, seed=24, max_val=8.0)
    plan2 = ebos.EventPlan.build(G(ev2), (h, w), "first", True, tile=(32, 32))
    a = plan2.iwe_dense(G(fl2, torch.float32), halo=16)
    b = plan2.iwe_dense(G(fl2, torch.float32), halo=16)
    assert torch.equal(a, b)


def test_fused_2dof_hypotheses(ebos):
    h, w, n = 100, 120, 50_000
    ev = O.synth_events(n, h, w, seed=9)
    thetas = np.array([[3.0, -2.0], [0.0, 0.0], [-7.5, 4.25], [12.0, 12.0]])
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=None)
    th = G(thetas, torch.float32).requires_grad_(True)
    iwes = plan.iwe_2dof(th)
    loss = -ebos.ops.image_variance(iwes)
    loss.sum().backward()
    for k in range(len(thetas)):
        t = torch.tensor(thetas[k], dtype=torch.float64, requires_grad=True)
        iwe_ref = O.iwe_2dof(torch.from_numpy(ev), t, (h, w))
        L = O.image_variance(iwe_ref)
        L.backward()
        assert rel(iwes[k].detach().cpu().numpy(), iwe_ref.detach().numpy()) < 1e-5
        assert abs(loss[k].item() - L.item()) < 1e-5 * abs(L.item())
        np.testing.assert_allclose(th.grad[k].cpu().numpy(), t.grad.numpy(), rtol=2e-3, atol=1e-6)


def test_2dof_tile_private_sweep(ebos):
    """Hypothesis sweep on a binned plan (tile-private pipeline, uniform motion) vs the oracle."""
    h, w, n = 96, 128, 60_000
    ev = O.synth_events(n, h, w, seed=31)
    grid = np.stack(np.meshgrid(np.linspace(-30, 30, 4), np.linspace(-30, 30, 3), indexing="ij"), -1).reshape(-1, 2)
    grid = np.concatenate([grid, [[45.0, -50.0]]])  # beyond the halo: spills to global atomics
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    assert plan.compact
    var = plan.variance_2dof(G(grid, torch.float32), halo=32, chunk=5)
    iwes = plan.iwe_2dof(G(grid, torch.float32), halo=32)
    # variance_2dof runs the PERSISTENT accumulate pass over hypotheses (ebos_iwe_2dof_slab_batch_f32: one launch per chunk, every
    # workgroup keeps its tile and walks the chunk), iwe_2dof one launch per hypothesis: the same variances, bit for bit, in any
    # chunking, with built halos and run-time windows (the spilling hypothesis aside: global float atomics)
    per_hyp = torch.stack([ebos.ops.image_variance(iwes[k]) for k in range(len(grid))])
    for chunk, halo in ((5, 32), (16, 32), (1, 32), (3, "auto"), (16, "auto")):
        vv = plan.variance_2dof(G(grid, torch.float32), halo=halo, chunk=chunk, n_streams=2)
        assert torch.equal(vv[:-1], var[:-1]), (chunk, halo)
        assert abs(vv[-1].item() - var[-1].item()) <= 1e-6 * var[-1].item()
    assert float((per_hyp[:-1] - var[:-1]).abs().max()) <= 2e-6 * float(var[:-1].abs().max())  # (ops.image_variance: another reduction order)
    for k, th in enumerate(grid):
        ref = O.iwe_2dof(torch.from_numpy(ev), torch.from_numpy(th), (h, w))
        assert rel(iwes[k].cpu().numpy(), ref.numpy()) < 1e-5, k
        v = torch.var(ref).item()
        assert abs(var[k].item() - v) < 1e-5 * v, k
    # gradient of the hypotheses through the tile-private backward
    th = G(grid[:5], torch.float32).requires_grad_(True)
    (-ebos.ops.image_variance(plan.iwe_2dof(th, halo=32))).sum().backward()
    for k in range(5):
        t = torch.tensor(grid[k], dtype=torch.float64, requires_grad=True)
        O.image_variance(O.iwe_2dof(torch.from_numpy(ev), t, (h, w))).backward()
        np.testing.assert_allclose(th.grad[k].cpu().numpy(), t.grad.numpy(), rtol=2e-3, atol=2e-6)
    # fractional source coordinates: the plan keeps the (x, y, dt) format and still agrees
    ev2 = ev.copy()
    ev2[::3, 0] += 0.25
    plan2 = ebos.EventPlan.build(G(ev2), (h, w), "first", True, tile=(32, 32))
    assert not plan2.compact
    i2 = plan2.iwe_2dof(G(grid[:3], torch.float32), halo=32)
    for k in range(3):
        ref = O.iwe_2dof(torch.from_numpy(ev2), torch.from_numpy(grid[k]), (h, w))
        assert rel(i2[k].cpu().numpy(), ref.numpy()) < 1e-5


def test_spill_image_is_consumed_by_the_call_that_wrote_it(ebos):
    """Taps beyond the halo go to the workspace's spill image and the combine pass reads that image only when THIS call's
    accumulate pass stamped the workspace (csrc/iwe_tiled.hip, SpillEpoch).  Alternate calls that spill with calls that do not, on
    one plan (= one workspace): every result must match the oracle, in either order, and the calls that do not spill repeat
    bit for bit whatever ran before them."""
    h, w, n = 96, 128, 40_000
    ev = O.synth_events(n, h, w, seed=77)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    thetas = [[3.0, -2.0], [60.0, -55.0], [0.5, 0.25], [-70.0, 48.0], [3.0, -2.0], [-70.0, 48.0], [0.5, 0.25]]  # halo 16: the big ones spill
    seen = {}
    for k, th in enumerate(thetas):
        iwe = plan.iwe_2dof(G(np.array([th]), torch.float32), halo=16)[0].cpu().numpy()
        ref = O.iwe_2dof(torch.from_numpy(ev), torch.tensor(th, dtype=torch.float64), (h, w)).numpy()
        assert rel(iwe, ref) < 1e-5, (k, th)
        key = tuple(th)
        if key in seen and max(abs(th[0]), abs(th[1])) <= 16:  # (spill taps are global float atomics: order-dependent rounding)
            assert np.array_equal(seen[key], iwe), (k, th)
        seen[key] = iwe
    # dense flow: a field whose left half stays inside the halo and whose right half leaves it, then a small field again
    rs = np.random.RandomState(5)
    small = rs.uniform(-8, 8, (2, h, w))
    big = small.copy()
    big[:, :, w // 2:] = rs.uniform(40, 70, (2, h, w - w // 2))
    for k, fl in enumerate([small, big, small, big]):
        iwe = plan.iwe_dense(G(fl, torch.float32), halo=16).cpu().numpy()
        ref = O.iwe_dense(torch.from_numpy(ev), torch.from_numpy(fl), (h, w)).numpy()
        assert rel(iwe, ref) < 1e-5, k


@pytest.mark.parametrize("n_windows", [3, 18])
def test_batched_windows_equal_per_window_calls(ebos, n_windows):
    """ebos_iwe_slab_batch_f32 (SlabBatch): several independent windows per launch -- dense fields and patch grids, windows of
    different event counts (one of them empty-ish), more windows than one launch holds (16): images and variances are
    bit-identical to one call per window, and the first window also matches the oracle."""
    h, w = 96, 128
    rs = np.random.RandomState(3)
    evs = [O.synth_events(int(n), h, w, seed=100 + k) for k, n in enumerate(rs.randint(50, 30_000, n_windows))]
    plans = [ebos.EventPlan.build(G(e), (h, w), "first", True, tile=(32, 32), emit="compact") for e in evs]
    assert all(p.compact for p in plans)
    # dense fields (one of them leaves the halo: spill path inside a batch)
    flows = [rs.uniform(-12, 12, (2, h, w)) for _ in range(n_windows)]
    flows[1][:, :, : w // 2] = 45.0
    fl_gpu = [G(f, torch.float32) for f in flows]
    batch = ebos.SlabBatch(plans, fl_gpu, halo=16)
    var = batch.run().cpu().numpy()
    for k in range(n_windows):
        iwe = plans[k].iwe_dense(fl_gpu[k], halo=16)
        if k != 1:  # (spill taps are global float atomics: not bit-reproducible)
            assert torch.equal(iwe, batch.iwes[k]), k
        else:
            assert rel(batch.iwes[k].cpu().numpy(), iwe.cpu().numpy()) < 1e-6
        v = torch.var(iwe.double()).item()
        assert abs(var[k] - v) <= 1e-5 * abs(v) + 1e-12, k
    ref = O.iwe_dense(torch.from_numpy(evs[0]), torch.from_numpy(flows[0]), (h, w)).numpy()
    assert rel(batch.iwes[0].cpu().numpy(), ref) < 1e-5
    again = batch.run().cpu().numpy()  # same workspaces, second call
    keep = np.arange(n_windows) != 1
    assert np.array_equal(var[keep], again[keep])
    # run-time windows (EBOS_HALO_AUTO) through the persistent batched pass: the images of the largest built halo
    auto = ebos.SlabBatch(plans, fl_gpu, halo="auto")
    big = ebos.SlabBatch(plans, fl_gpu, halo=32)
    va, vb = auto.run().cpu().numpy(), big.run().cpu().numpy()
    torch.cuda.synchronize()
    for k in range(n_windows):
        if k != 1:
            assert torch.equal(auto.iwes[k], big.iwes[k]) and va[k] == vb[k], k
    # patch grids
    patch, slide = (8, 8), (8, 8)
    gh, gw = ebos.solver.patch_grid_shape((h, w), patch, slide)
    grids = [G(rs.uniform(-10, 10, (2, gh, gw)), torch.float32) for _ in range(n_windows)]
    lib = ebos._hip.require_gpu()
    if not lib.ebos_patch_fused_supported(32, 32, 16, *slide):
        pytest.skip("grid-sampling kernels not built for this tile")
    pb = ebos.SlabBatch(plans, grids, patch=(patch, slide), halo=16)
    pv = pb.run().cpu().numpy()
    for k in range(n_windows):
        dense = ebos.ops.upsample_patch_flow(grids[k], patch, slide, (h, w))
        iwe = plans[k].iwe_dense(dense, halo=16)
        assert rel(pb.iwes[k].cpu().numpy(), iwe.cpu().numpy()) < 1e-5, k
        v = torch.var(iwe.double()).item()
        assert abs(pv[k] - v) <= 2e-5 * abs(v) + 1e-12, k


def _window_table(ebos, plan, ws):
    """(hr, hc) per tile as the accumulate pass of a run-time-window call left them behind the SpillEpoch word of its workspace"""
    th, tw = plan.tile
    tiles = -(-plan.image_size[0] // th) * -(-plan.image_size[1] // tw)
    tab = ws[-((tiles * 4 + 255) // 256 * 256):].view(torch.int32)[:tiles].cpu().numpy()
    return tab & 255, tab >> 8


# (150 k events on 96 x 128: 12 per pixel -- dense tiles behind small windows take the PAIRS lane mapping with same-cell events merged
# before the add; the built halo runs the plain loop: the images must still be the same bits)
@pytest.mark.parametrize("size,tile,n", [((96, 128), (32, 32), 30_000), ((720, 1280), (45, 80), 400_000), ((96, 128), (32, 32), 150_000)])
def test_run_time_windows_give_the_images_of_the_largest_built_halo(ebos, size, tile, n):
    """halo="auto" (EBOS_HALO_AUTO: every work item sizes its LDS window from a bound on its OWN displacements) against the built
    32 px halo on the same plan -- dense flow, 2-DoF, patch grid, forward and backward, alternating small and large flows on ONE
    plan / workspace: images and variances BIT-identical whenever nothing spills (a window only decides where the integer sums
    are kept), gradients to 1e-6 (unordered f64 adds); the windows the kernels chose are read back and checked against the bound
    (hr = ceil(max |u| over the tile) + 1, hc the same for v rounded up to 4, both <= 32)."""
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    h, w = size
    rs = np.random.RandomState(5)
    ev = O.synth_events(n, h, w, seed=21)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=tile, emit="compact")
    assert plan.dt_bound == 1.0
    code = ebos.event_plan.resolve_halo(plan, "auto")
    assert code == -(32 + 256 * 64)
    th, tw = tile
    for amp in (0.4, 30.0, 3.0, 11.0, 45.0, 0.0, 6.0):  # small / large alternate on the same workspaces
        fl = rs.uniform(-amp, amp, (2, h, w)) if amp else np.zeros((2, h, w))
        fl[:, : h // 2, : w // 3] *= 0.1  # tiles differ: windows are per tile
        flow = G(fl, torch.float32)
        a = plan.iwe_dense(flow, halo="auto")
        b = plan.iwe_dense(flow, halo=32)
        if amp < 31.0:
            assert torch.equal(a, b), amp
        else:  # taps beyond 32 px spill through global float atomics in both: not bit-reproducible
            assert rel(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
        hr, hc = _window_table(ebos, plan, plan.__dict__["_workspaces"][(code, plan.resolve_splits(None), 0, 0)])
        ty, tx = -(-h // th), -(-w // tw)
        au = np.abs(fl).reshape(2, h, w)
        for t in range(ty * tx):
            r0, c0 = (t // tx) * th, (t % tx) * tw
            mu, mv = au[0, r0:r0 + th, c0:c0 + tw].max(), au[1, r0:r0 + th, c0:c0 + tw].max()
            want_r = min(32, int(np.ceil(np.float32(mu))) + 1)
            want_c = min(32, (int(np.ceil(np.float32(mv))) + 1 + 3) // 4 * 4)
            assert (hr[t], hc[t]) == (max(want_r, 1), max(want_c, 4)), (amp, t, hr[t], hc[t], mu, mv)
        fa, fb = flow.clone().requires_grad_(True), flow.clone().requires_grad_(True)
        va, vb = plan.contrast_dense(fa, halo="auto"), plan.contrast_dense(fb, halo=32)
        va.backward(), vb.backward()
        if amp < 31.0:
            assert va.item() == vb.item(), amp
        # (the backward scatter is fixed point with a scale per tile from max |upstream| over the tile's WINDOW: two window sizes
        # quantise differently -- 2^-20 of the largest possible contribution per run)
        assert rel(fa.grad.cpu().numpy(), fb.grad.cpu().numpy()) < 2e-5, amp
    # 2-DoF hypotheses: one theta for the whole image, windows from |theta| x max |dt|
    thetas = G(np.array([[0.3, -0.2], [5.0, 0.5], [-12.0, 29.0], [0.0, 0.0], [33.0, -2.0]]), torch.float32)
    ia, ib = plan.iwe_2dof(thetas, halo="auto"), plan.iwe_2dof(thetas, halo=32)
    assert torch.equal(ia[:4], ib[:4]) and rel(ia[4].cpu().numpy(), ib[4].cpu().numpy()) < 1e-6
    assert torch.equal(plan.variance_2dof(thetas[:4], halo="auto"), plan.variance_2dof(thetas[:4], halo=32))
    ta, tb = thetas.clone().requires_grad_(True), thetas.clone().requires_grad_(True)
    ebos.ops.image_variance(plan.iwe_2dof(ta, halo="auto")).sum().backward()
    ebos.ops.image_variance(plan.iwe_2dof(tb, halo=32)).sum().backward()
    assert rel(ta.grad.cpu().numpy(), tb.grad.cpu().numpy()) < 1e-5
    # patch grid: forward + backward + Adam, windows from the cells a tile interpolates
    patch = slide = (8, 8) if h < 200 else (24, 32)
    lib = ebos._hip.require_gpu()
    if not lib.ebos_patch_fused_supported(th, tw, 32, *slide):
        return
    gh, gw = ebos.solver.patch_grid_shape((h, w), patch, slide)
    for amp in (0.5, 9.0, 30.0):
        theta = G(rs.uniform(-amp, amp, (2, gh, gw)), torch.float32)
        out = {}
        for halo in ("auto", 32):
            loop = FusedPatchLoop(plan, patch, slide, theta, 1.0, 0.01, 0.002, capacity=12, lr=0.05, halo=halo, sample_grid=True)
            loss, grad = loop.value_and_grad(theta)
            out[halo] = (loop.iwe.clone(), float(loss), grad.clone(), loop.run(10).clone())
        assert torch.equal(out["auto"][0], out[32][0]) and out["auto"][1] == out[32][1], amp
        assert rel(out["auto"][2].cpu().numpy(), out[32][2].cpu().numpy()) < 1e-5, amp
        np.testing.assert_allclose(out["auto"][3].cpu().numpy(), out[32][3].cpu().numpy(), rtol=1e-4)


@pytest.mark.parametrize("size,tile,n", [((96, 128), (32, 32), 20_000), ((720, 1280), (45, 80), 400_000)])
def test_slab_handoff_never_reads_a_previous_calls_slabs(ebos, size, tile, n):
    """The accumulate pass stores its slabs write-through (sc1) and the combine pass reads them with sc1 loads (csrc/iwe_tiled.hip):
    nothing of an earlier call's slabs -- same addresses, possibly still cached somewhere -- may reach a later image.  Two different
    windows (and two flows each) share ONE workspace and alternate 150 times, at a size where every slab line of the previous call
    still fits the L2s and at the full image size: every image and variance must equal, bit for bit, the one computed with a
    workspace of its own."""
    from event_based_bos_amd import _hip

    lib = _hip.require_gpu()
    h, w = size
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(11)
    plans = [ebos.EventPlan.build(G(O.synth_events(n + 977 * k, h, w, seed=50 + k)), (h, w), "first", True, tile=tile, emit="compact")
             for k in range(2)]
    flows = [G(rs.uniform(-20, 20, (2, h, w)), torch.float32) for _ in range(2)]
    halo, splits = 32, 1
    nws = int(lib.ebos_iwe_slab_workspace_bytes(h, w, tile[0], tile[1], halo, splits, 0, 0))
    P = lambda t: None if t is None else t.data_ptr()

    def run(pl, fl, ws, iwe, out):
        _hip.check(lib.ebos_iwe_dense_slab_f32(None, None, None, None, *pl._compact_ptrs(), P(pl.key_offsets), pl.n, P(fl), h, w, tile[0], tile[1],
                                               halo, splits, 0, 0, P(ws), nws, P(iwe), 1, 0, P(out), None, None, _hip.stream_ptr()),
                   "ebos_iwe_dense_slab")

    want = {}
    for a in range(2):
        for b in range(2):
            ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
            iwe, out = torch.empty((h, w), dtype=torch.float32, device=dev), torch.empty(1, dtype=torch.float32, device=dev)
            run(plans[a], flows[b], ws, iwe, out)
            want[a, b] = (iwe.clone(), out.clone())
    assert not torch.equal(want[0, 0][0], want[1, 0][0]) and not torch.equal(want[0, 0][0], want[0, 1][0])
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)
    iwe, out = torch.empty((h, w), dtype=torch.float32, device=dev), torch.empty(1, dtype=torch.float32, device=dev)
    bad = 0
    for it in range(150):
        a, b = it % 2, (it // 2) % 2
        run(plans[a], flows[b], ws, iwe, out)
        bad += int(not torch.equal(iwe, want[a, b][0])) + int(not torch.equal(out, want[a, b][1]))
    assert bad == 0, f"{bad} of 300 results differ from the fresh-workspace ones"


@pytest.mark.parametrize("emit", ["full", "compact"])
def test_empty_and_three_event_windows_through_the_plan_paths(ebos, emit):
    """A window without events, and one with three: plan build (both builds), tile-private forward, value + gradient in one call, and
    the batched entry.  No event -> zero image, zero variance, zero gradient; three events -> mass 3 and the oracle's variance."""
    h, w = 96, 128
    dev = torch.device("cuda:0")
    for n in (0, 3):
        ev = np.zeros((n, 4))
        if n:
            ev[:, 0], ev[:, 1], ev[:, 2] = np.arange(n) + 5, 7, np.linspace(0, 1, n)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (h, w), "first", True, tile=(32, 32), emit=emit)
        assert plan.n == n
        fl = torch.full((2, h, w), 1.5, device=dev)
        iwe = plan.iwe_dense(fl, halo=16)
        v, g = plan.variance_and_grad_dense(fl)
        assert abs(float(iwe.sum()) - n) < 1e-5 and torch.isfinite(g).all()
        if n == 0:
            assert float(v) == 0.0 and float(g.abs().sum()) == 0.0
        else:
            ref = O.iwe_dense(torch.from_numpy(ev), fl.cpu().double(), (h, w))
            assert rel(iwe.cpu().numpy(), ref.numpy()) < 1e-5 and abs(float(v) - torch.var(ref).item()) < 1e-5 * torch.var(ref).item()
        if plan.compact:
            batch = ebos.SlabBatch([plan, plan], [fl, fl], halo=16)
            assert batch.run().tolist() == [float(v), float(v)] and torch.equal(batch.iwes[0], iwe) and torch.equal(batch.iwes[1], iwe)


def test_non_finite_events_are_contained(ebos):
    """NaN / Inf coordinates and timestamps: the reference poisons pixel 0 (NaN * 0 in the masked scatter); here
    such taps are dropped and every other pixel is unaffected."""
    h, w, n = 40, 48, 5000
    ev = O.synth_events(n, h, w, seed=51)
    fl = O.synth_dense_flow(h, w, seed=52, max_val=4.0)
    ref = O.iwe_dense(torch.from_numpy(ev), torch.from_numpy(fl), (h, w)).numpy()
    warped = O.warp_dense_numpy(ev, fl, "first", True)
    bad = warped.copy()
    bad = np.concatenate([bad, [[np.nan, 3.0, 0.1, 1], [5.0, np.inf, 0.2, 0], [-np.inf, np.nan, 0.3, 1], [1e30, 2.0, 0.4, 0]]])
    ic = ebos.EventImageConverter((h, w))
    img = ic.bilinear_vote_numpy(bad)
    assert np.isfinite(img).all() and rel(img, ic.bilinear_vote_numpy(warped)) < 1e-14
    img_t = ic.bilinear_vote_tensor(G(bad, torch.float32))
    assert torch.isfinite(img_t).all()
    # fused path: NaN timestamp on one event must not leak beyond that event
    ev2 = ev.copy()
    plan = ebos.EventPlan.build(G(ev2), (h, w), "first", True, tile=(32, 32))
    plan.dt[7] = float("nan")
    if plan.compact:
        plan.cdt[7] = float("nan")
    out = plan.iwe_dense(G(fl, torch.float32), halo=8)
    assert torch.isfinite(out).all()
    assert abs(out.sum().item() - ref.sum()) < 1.5  # at most that one event is missing


@pytest.mark.parametrize("emit", ["full", "compact"])
def test_inf_and_huge_flow_entries_mask_their_events_out(ebos, emit):
    """a flow that has diverged at some pixels (+-Inf, +-1e12, +-3e9 -- beyond what a 32-bit column << 2 holds --
    NaN) in u only, v only, or both.  The reference's vote masks such events out (their coordinates fail the in-image test,
    src/event_image_converter.py:596-608); here the image and the variance must equal those of the window WITHOUT the events
    that sit on a bad pixel, in every kernel organisation, and the gradient must agree with the value (zero at the bad pixels)."""
    h, w, n = 96, 128, 60_000
    ev = O.synth_events(n, h, w, seed=71)
    fl = O.synth_dense_flow(h, w, seed=72, max_val=5.0)
    rs = np.random.RandomState(73)
    first = ev[np.argmin(ev[:, 2])]  # (dt = 0 there: 0 * 1e12 = 0 keeps a finite-flow event in place -- keep its pixel clean)
    bads = [np.inf, -np.inf, 1e12, -1e12, 3e9, -3e9, 2.0 ** 30, -(2.0 ** 31), np.nan]
    bad = np.zeros((h, w), dtype=bool)
    fb = fl.copy()
    for k in range(270):
        r, c = rs.randint(0, h), rs.randint(0, w)
        if (r, c) == (int(first[0]), int(first[1])):
            continue
        comp = (0,), (1,), (0, 1)
        for ch in comp[k % 3]:
            fb[ch, r, c] = bads[k % len(bads)]
        bad[r, c] = True
    keep = ~bad[ev[:, 0].astype(int), ev[:, 1].astype(int)]
    assert 0 < (~keep).sum() < n // 10
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32), emit=emit)
    # the expected image: the clean flow, the kept events, ON THE SAME TIME BASE (weights zero the dropped events out)
    plan_full = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    wgt = G(keep.astype(np.float32))
    want = plan_full.iwe_dense(G(fl, torch.float32), weight=wgt, halo=32)
    for halo in (32, 8):
        got = plan.iwe_dense(G(fb, torch.float32), halo=halo)
        assert torch.isfinite(got).all()
        assert rel(got.cpu().numpy(), want.cpu().numpy()) < 1e-6, (emit, halo)
        f = G(fb, torch.float32).requires_grad_(True)
        v = plan.contrast_dense(f, "image_variance", halo=halo)
        v.backward()
        assert abs(v.item() - want.var().item()) <= 1e-5 * want.var().item()
        g = f.grad.cpu().numpy()
        assert np.isfinite(g).all() and not g[:, bad].any()
        fc = G(fl, torch.float32).requires_grad_(True)
        ebos.ops.image_variance(plan_full.iwe_dense(fc, weight=wgt, halo=32)).backward()
        assert rel(g[:, ~bad], fc.grad.cpu().numpy()[:, ~bad]) < 1e-4


def test_plan_binning_properties(ebos):
    h, w, n = 70, 90, 30_000
    ev = O.synth_events(n, h, w, seed=11)
    ev[5, 0] = -3.0   # source outside the image: dropped from a binned plan
    ev[6, 1] = 1e9
    plan = ebos.EventPlan.build(G(ev), (h, w), "middle", True, tile=(32, 32))
    assert plan.n == n - 2 and plan.n_dropped == 2
    perm = plan.perm.cpu().numpy()
    assert len(np.unique(perm)) == plan.n and 5 not in perm and 6 not in perm
    # sortedness: tile-major keys are non-decreasing
    x, y = plan.x.cpu().numpy(), plan.y.cpu().numpy()
    r, c = x.astype(np.int64), y.astype(np.int64)
    tiles_x = (w + 31) // 32
    key = ((r // 32) * tiles_x + c // 32) * 1024 + (r % 32) * 32 + (c % 32)
    assert np.all(np.diff(key) >= 0)
    ko = plan.key_offsets.cpu().numpy()
    assert ko[0] == 0 and ko[-1] == plan.n and np.all(np.diff(ko) >= 0)
    np.testing.assert_array_equal(np.bincount(key, minlength=len(ko) - 1), np.diff(ko))
    # the permutation really maps planned events to input events; dt evaluated in fp64 then rounded
    np.testing.assert_array_equal(x, ev[perm, 0].astype(np.float32))
    dt_ref = O.delta_t(ev, O.reference_time(ev, "middle"), True)
    np.testing.assert_array_equal(plan.dt.cpu().numpy(), dt_ref[perm].astype(np.float32))


def test_upsample_patch_flow(ebos):
    img, patch, slide = (720, 1280), (24, 32), (24, 32)
    grid = np.random.RandomState(100).uniform(-30, 30, (2, 30, 40))
    ref = O.upsample_patch_flow(torch.from_numpy(grid), img, patch, slide)
    g = G(grid, torch.float32).requires_grad_(True)
    dense = ebos.ops.upsample_patch_flow(g, patch, slide, img)
    assert dense.shape == (2, 720, 1280)
    assert rel(dense.detach().cpu().numpy(), ref.numpy()) < 1e-6
    # adjoint against torch autograd of the oracle
    up = np.random.RandomState(101).uniform(-1, 1, (2, 720, 1280))
    (dense * G(up, torch.float32)).sum().backward()
    gr = torch.from_numpy(grid).clone().requires_grad_(True)
    (O.upsample_patch_flow(gr, img, patch, slide) * torch.from_numpy(up)).sum().backward()
    assert rel(g.grad.cpu().numpy(), gr.grad.numpy()) < 1e-5
    # overlapping windows (patch 48x64, slide 24x32)
    gh, gw = O.patch_grid_shape(img, (48, 64), slide)
    grid2 = np.random.RandomState(102).uniform(-5, 5, (2, gh, gw))
    ref2 = O.upsample_patch_flow(torch.from_numpy(grid2), img, (48, 64), slide)
    d2 = ebos.ops.upsample_patch_flow(G(grid2, torch.float32), (48, 64), slide, img)
    assert rel(d2.cpu().numpy(), ref2.numpy()) < 1e-6


@pytest.mark.parametrize("shape,axis,radius,boundary", [
    ((24, 32), 0, 1, 1), ((24, 32), 1, 1, 1), ((2, 3, 9, 7), 2, 1, 1),      # torch 'reflect', 3 taps
    ((24, 32), 0, 4, 0), ((24, 32), 1, 12, 0), ((5, 6), 0, 12, 0),           # scipy 'reflect'; radius > 2L too
    ((1, 8), 0, 2, 0), ((3, 2, 5), 1, 3, 0),
])
def test_blur_pass_adjoint(ebos, shape, axis, radius, boundary):
    """ebos_gauss1d_bwd is the exact transpose of ebos_gauss1d: checked against the dense operator matrix built from
    the library the reference calls (scipy correlate1d 'reflect' = gaussian_filter's pass; torch reflect pad + conv
    for the tensor branch).  fp64, rel-L2 <= 1e-13."""
    from scipy.ndimage import correlate1d

    rng = np.random.default_rng(5)
    taps = rng.uniform(0.1, 1.0, 2 * radius + 1)
    taps = (taps + taps[::-1]) / 2
    L = shape[axis]
    eye = np.eye(L)
    if boundary == 0:
        A = correlate1d(eye, taps, axis=0, mode="reflect")                  # A[i, j] = d out[i] / d in[j]
    else:
        pad = torch.nn.functional.pad(torch.from_numpy(eye.T)[None], (radius, radius), mode="reflect")[0]
        A = torch.nn.functional.conv1d(pad[:, None], torch.from_numpy(taps)[None, None])[:, 0].numpy().T
    x = rng.normal(size=shape)
    gy = rng.normal(size=shape)
    xt = G(x).requires_grad_(True)
    y = ebos.ops.gauss1d(xt, axis, torch.from_numpy(taps), boundary)
    expect_y = np.moveaxis(np.tensordot(A, np.moveaxis(x, axis, 0), axes=1), 0, axis)
    assert rel(y.detach().cpu().numpy(), expect_y) <= 1e-13
    y.backward(G(gy))
    expect_gx = np.moveaxis(np.tensordot(A.T, np.moveaxis(gy, axis, 0), axes=1), 0, axis)
    assert rel(xt.grad.cpu().numpy(), expect_gx) <= 1e-13


def test_upsample_and_blur_vs_reference_code_with_shimmed_torchvision(ebos):
    """HIP upsample kernel (ebos_upsample_patch_flow_f32), the grid-sampling event kernels' flow (through a zero-displacement
    check) and the GPU 3-tap blur against tests/golden/golden_upsample.npz -- the reference's own
    interpolate_dense_flow_from_patch_tensor / create_image_from_events_tensor run with torchvision's resize / gaussian_blur
    shimmed (make_golden.py --upsample).  f32 kernels: 1e-4 px absolute on a +-30 px field (f32 lerp of values up to 30: a few
    1e-5); f64 blur: 1e-12."""
    from test_oracle_golden import _upsample_cases, check_dense_against_fixture

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_upsample.npz"), allow_pickle=False))
    for tag, size, patch, slide in _upsample_cases(g):
        grid = G(g[tag + "_grid"], torch.float32).requires_grad_(True)
        dense = ebos.ops.upsample_patch_flow(grid, patch, slide, size)
        check_dense_against_fixture(g, tag, dense.detach().double().cpu().numpy(), rtol=1e-5, atol=1e-4)
        # adjoint: <upsample(grid), probe> differentiated = upsample^T probe, against torch autograd of the oracle map
        probe = np.random.RandomState(7).normal(size=tuple(dense.shape))
        (dense * G(probe, torch.float32)).sum().backward()
        gt = torch.from_numpy(g[tag + "_grid"]).requires_grad_(True)
        (O.upsample_patch_flow(gt, size, patch, slide) * torch.from_numpy(probe)).sum().backward()
        assert rel(grid.grad.double().cpu().numpy(), gt.grad.numpy()) < 1e-5, tag
    Hh, Ww = 24, 32
    for pad in (0, 2):
        ic = ebos.EventImageConverter((Hh, Ww), outer_padding=pad)
        for sigma in (1, 3):
            for key, ev, dt, tol in ((f"b_p{pad}_s{sigma}", g["b_events"], torch.float64, 1e-12),
                                     (f"b_p{pad}_s{sigma}_batched", g["b_events_batched"], torch.float64, 1e-12),
                                     (f"b_p{pad}_s{sigma}_f32", g["b_events"], torch.float32, 2e-6)):
                img = ic.create_image_from_events_tensor(G(ev, dt), "bilinear_vote", sigma=sigma)
                assert tuple(img.shape) == tuple(g[key].shape) and img.dtype == dt
                assert rel(img.double().cpu().numpy(), g[key].astype(np.float64)) < tol, key
    d = ebos.EventImageConverter((Hh, Ww)).create_iwe(G(g["b_events"]))  # default sigma of create_iwe is 1 (:55)
    assert rel(d.cpu().numpy(), g["b_create_iwe_default"]) < 1e-12


def test_blurred_iwe_is_differentiable(ebos):
    """create_iwe(warped, sigma=1) under autograd, as the reference's tensor branch allows
    (src/event_image_converter.py:399-404): gradient of var(blur(IWE)) w.r.t. the flow against the oracle's
    torch-CPU autograd through the same chain (fp64 general path, rel-L2 <= 1e-9)."""
    ev = O.synth_events(3000, H, W, seed=21)
    flow = O.synth_dense_flow(H, W, seed=22, max_val=2.0)
    ft = torch.from_numpy(flow).requires_grad_(True)
    w_ref = O.warp_dense_torch(torch.from_numpy(ev), ft, "first", True)
    img = O.gaussian_blur3_torch(O.bilinear_vote_torch(w_ref, (H, W)), 1.0)
    O.image_variance(img, False, "minimize").backward()

    fg = G(flow).requires_grad_(True)
    warper = ebos.Warp((H, W), normalize_t=True)
    ic = ebos.EventImageConverter((H, W))
    warped, _ = warper.warp_event(G(ev), fg, "dense-flow", direction="first")
    iwe = ic.create_iwe(warped, method="bilinear_vote", sigma=1)
    cost = ebos.costs.functions["image_variance"](direction="minimize")
    cost.calculate({"iwe": iwe, "omit_boundary": False}).backward()
    assert rel(fg.grad.cpu().numpy(), ft.grad.numpy()) <= 1e-9


@pytest.mark.parametrize("shape,w_norm,w_tv", [((24, 32), 0.7, 0.0), ((24, 32), 0.0, 1.3), ((7, 5), 0.4, 0.9),
                                                ((2, 2), 0.0, 1.0), ((720, 1280), 0.01, 0.5)])
def test_flow_regularisers_value_and_gradient(ebos, shape, w_norm, w_tv):
    """ebos_flow_regularisers_f32 (through the C ABI) against torch-CPU autograd of the oracle's flow_norm /
    image_gradient restatements (fp64).  Value rel < 1e-5; gradient rel-L2 < 1e-5 (f32 kernel; sign() of an exactly
    zero difference is 0 on both sides: the inputs contain flat runs and exact zeros on purpose)."""
    from event_based_bos_amd._hip import check, ptr, stream_ptr

    lib = ebos.load_library()
    h, w = shape
    rng = np.random.default_rng(8)
    flow = rng.uniform(-3, 3, (2, h, w)).astype(np.float32)
    flow[:, : h // 3, : w // 2] = 1.25      # flat patch: zero differences
    flow[:, -1, -1] = 0.0                    # exact zero vector: norm sub-gradient 0
    ft = torch.from_numpy(flow.astype(np.float64)).requires_grad_(True)
    val = w_norm * O.flow_norm(ft) + w_tv * O.image_gradient_tv(ft, torch.ones(h, w, dtype=torch.float64))
    val.backward()
    fg = G(flow)
    d = torch.empty_like(fg)
    n_part = lib.ebos_flow_regularisers_partials()
    parts = torch.zeros(n_part, dtype=torch.float64, device=dev())
    check(lib.ebos_flow_regularisers_f32(ptr(fg), h, w, w_norm, w_tv, ptr(d), ptr(parts), None, 0, 0, None, None, stream_ptr()), "reg")
    assert abs(parts.sum().item() - val.item()) <= 1e-5 * abs(val.item())
    assert rel(d.cpu().numpy(), ft.grad.numpy()) < 1e-5


def test_adam_step_kernel_matches_torch_adam(ebos):
    """ebos_cmax_adam_step_f32 against torch.optim.Adam (CPU, f32) over 25 steps with a changing gradient:
    parameters agree to 1e-6 absolute (same update formula; f32 rounding of the bias corrections differs), the step
    counter advances and losses[t] = scale * contrast + sum(partials) is recorded."""
    from event_based_bos_amd._hip import check, ptr, stream_ptr

    lib = ebos.load_library()
    rng = np.random.default_rng(9)
    n, steps, lr = 2 * 7 * 9, 25, 0.05
    x0 = rng.normal(size=n).astype(np.float32)
    grads = rng.normal(size=(steps, n)).astype(np.float32) * np.linspace(1.0, 0.01, steps, dtype=np.float32)[:, None]
    ref = torch.from_numpy(x0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=lr)
    theta, m, v = G(x0.copy()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    losses = torch.zeros(steps, device=dev())
    contrast = torch.zeros(1, device=dev())
    parts = torch.zeros(4, dtype=torch.float64, device=dev())
    for t in range(steps):
        ref.grad = torch.from_numpy(grads[t].copy())
        opt.step()
        contrast.fill_(float(t))
        parts.fill_(0.25 * t)
        check(lib.ebos_cmax_adam_step_f32(ptr(theta), ptr(G(grads[t])), ptr(m), ptr(v), n, lr, 0.9, 0.999, 1e-8, ptr(step),
                                          ptr(contrast), -2.0, ptr(parts), 4, ptr(losses), steps, stream_ptr()), "adam")
        assert np.abs(theta.cpu().numpy() - ref.detach().numpy()).max() < 1e-6, t
    assert step.item() == steps
    np.testing.assert_allclose(losses.cpu().numpy(), [-2.0 * t + t for t in range(steps)], rtol=1e-6)


def test_upsample_adjoint_adam_with_grad_mask(ebos):
    """ebos_upsample_patch_flow_bwd_adam_f32 with grad_mask [gh, gw]: d_grid == mask * (adjoint of the oracle's upsample)
    (rel-L2 < 1e-5, zeros exact), masked grid cells keep their value bit-exactly over 10 steps, the others follow
    torch.optim.Adam fed the masked gradient (1e-5 absolute)."""
    from event_based_bos_amd._hip import check, ptr, stream_ptr

    lib = ebos.load_library()
    rng = np.random.default_rng(21)
    H, W, patch, slide = 70, 90, (16, 20), (8, 10)
    gh, gw = len(np.arange(0, H - patch[0] + slide[0], slide[0])), len(np.arange(0, W - patch[1] + slide[1], slide[1]))
    mask = (rng.uniform(size=(gh, gw)) > 0.4).astype(np.float32)
    x0 = rng.normal(size=(2, gh, gw)).astype(np.float32)
    ref = torch.from_numpy(x0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.1)
    theta, m, v = G(x0.copy()), torch.zeros((2, gh, gw), device=dev()), torch.zeros((2, gh, gw), device=dev())
    d_grid = torch.empty((2, gh, gw), device=dev())
    step = torch.zeros(1, dtype=torch.int32, device=dev())
    scratch = torch.empty(int(lib.ebos_upsample_bwd_scratch_bytes(gh, W)) // 4, device=dev())
    gmask = G(mask)
    for t in range(1, 11):
        up = rng.normal(size=(2, H, W)).astype(np.float32)
        probe = torch.zeros((2, gh, gw), dtype=torch.float64, requires_grad=True)
        (O.upsample_patch_flow(probe, (H, W), patch, slide) * torch.from_numpy(up).double()).sum().backward()
        want = probe.grad.numpy() * mask
        ref.grad = torch.from_numpy(want.astype(np.float32))
        opt.step()
        check(lib.ebos_upsample_patch_flow_bwd_adam_f32(ptr(G(up)), gh, gw, patch[0], patch[1], slide[0], slide[1], H, W, ptr(scratch),
                                                        ptr(d_grid), ptr(theta), ptr(m), ptr(v), 0.1, 0.9, 0.999, 1e-8, t, ptr(step),
                                                        None, 0.0, None, 0, None, 0, ptr(gmask), stream_ptr()), "bwd_adam")
        got = d_grid.cpu().numpy()
        assert np.all(got[:, mask == 0] == 0.0) and rel(got, want) < 1e-5, t
        th = theta.cpu().numpy()
        assert np.array_equal(th[:, mask == 0], x0[:, mask == 0]), t
        assert np.abs(th - ref.detach().numpy()).max() < 1e-5, t
    assert step.item() == 10


def test_tiled_backward_addend(ebos):
    """addend [2, H, W] of ebos_iwe_dense_tiled_bwd_f32 is added to d_flow exactly once (bit-exact: one f32 add)."""
    from event_based_bos_amd import event_plan as EP

    ev = O.synth_events(20000, 64, 96, seed=31)
    flow = G(O.synth_dense_flow(64, 96, seed=32, max_val=5.0)).float()
    plan = ebos.EventPlan.build(G(ev), (64, 96), "first", True, tile="auto")
    iwe, var, mom = EP._launch_iwe_dense_slab(plan, flow, None, (0, 0), 32, 1, True, False)
    up = torch.full((1,), -1.0, device=dev())
    base, _ = EP._launch_dense_bwd(plan, flow, None, (0, 0), iwe, None, 0, False, 32, mom, up)
    addend = torch.randn(2, 64, 96, device=dev())
    lib = ebos.load_library()
    out = torch.empty_like(base)
    EP.check(lib.ebos_iwe_dense_tiled_bwd_f32(EP.ptr(plan.x), EP.ptr(plan.y), EP.ptr(plan.dt), None, *plan._compact_ptrs(),
                                              EP.ptr(plan.key_offsets), plan.n, EP.ptr(flow), 64, 96, plan.tile[0], plan.tile[1],
                                              32, 0, 0, EP.ptr(iwe), None, 0, EP.ptr(out), None, EP.ptr(mom), EP.ptr(up),
                                              EP.ptr(addend), None, 0, None, EP.stream_ptr()), "bwd")
    assert torch.equal(out, base + addend)
    # the same through the adaptive work items (partial tiles + combine): the addend is applied by the combine kernel
    ws = EP._workspace(plan, (0, 0), 32, 0)
    out2 = torch.empty_like(base)
    EP.check(lib.ebos_iwe_dense_tiled_bwd_f32(EP.ptr(plan.x), EP.ptr(plan.y), EP.ptr(plan.dt), None, *plan._compact_ptrs(),
                                              EP.ptr(plan.key_offsets), plan.n, EP.ptr(flow), 64, 96, plan.tile[0], plan.tile[1],
                                              32, 0, 0, EP.ptr(iwe), None, 0, EP.ptr(out2), None, EP.ptr(mom), EP.ptr(up),
                                              EP.ptr(addend), EP.ptr(ws), ws.numel(), EP.ptr(plan.part_table), EP.stream_ptr()), "bwd")
    assert rel(out2.cpu().numpy(), (base + addend).cpu().numpy()) < 1e-6


def test_backward_fixed_point_scatter_is_exact_or_redone_in_f64(ebos):
    """The tile-private backward kernel scatters (d/du, d/dv) of a run of events as ONE ds_add_u64 of two signed fixed-point
    fields (csrc/iwe_tiled.hip, bwd_compact_slice ACC_FX).  Integer adds commute: the gradient is BIT-reproducible.  The mode is
    exact by construction -- the unit follows the fullest source pixel of the tile, every run sum is range-checked -- and a
    workgroup that fails a test redoes its slice with f64 accumulators: a hot pixel (3000 events), an upstream image with a
    1e6 spike that only spilled taps reach (their runs leave the range the tile's own window promised), and |dt| twice the bound
    the kernel assumes (reference time "before": dt in [1, 2]) all give the gradient of the general (global-atomic) kernels."""
    h, w, n = 96, 128, 40_000
    rs = np.random.RandomState(17)
    ev = O.synth_events(n, h, w, seed=81)
    fl = O.synth_dense_flow(h, w, seed=82, max_val=5.0)
    flow = G(fl, torch.float32)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))

    def grads(pl, f, halo, upstream=None, cost=True):
        fg = f.clone().requires_grad_(True)
        if cost:
            pl.contrast_dense(fg, "image_variance", halo=halo).backward()
        else:
            pl.iwe_dense(fg, halo=halo).backward(gradient=upstream)
        return fg.grad

    a, b = grads(plan, flow, 32), grads(plan, flow, 32)
    assert torch.equal(a, b)                                            # integer scatter: bit-reproducible
    ref = grads(plan, flow, None)                                       # general kernels: f32 global atomics
    assert rel(a.cpu().numpy(), ref.cpu().numpy()) < 2e-5
    f64 = torch.from_numpy(fl).requires_grad_(True)
    O.image_variance(O.iwe_dense(torch.from_numpy(ev), f64, (h, w))).backward()
    assert rel(a.cpu().numpy(), -f64.grad.numpy()) < 1e-3               # (un-filtered stream: the SURVEY 8d bar)
    # hot pixel: 3000 events on one source pixel -> its tile is refused the fixed-point mode (>= 1024 events on a pixel)
    hot = np.concatenate([ev, np.stack([np.full(3000, 40.0), np.full(3000, 50.0), rs.uniform(ev[:, 2].min(), ev[:, 2].max(), 3000),
                                        np.ones(3000)], 1)])
    hp = ebos.EventPlan.build(G(hot), (h, w), "first", True, tile=(32, 32))
    assert rel(grads(hp, flow, 32).cpu().numpy(), grads(hp, flow, None).cpu().numpy()) < 2e-5
    # an upstream spike that a tile only reaches through spilled taps (flow of 30 px against an 8 px halo)
    big = G(rs.uniform(-30, 30, (2, h, w)), torch.float32)
    up = G(rs.uniform(-1e-3, 1e-3, (h, w)), torch.float32)
    up[10, 100] = 1e6
    got, want = grads(plan, big, 8, up, cost=False), grads(plan, big, None, up, cost=False)
    assert rel(got.cpu().numpy(), want.cpu().numpy()) < 2e-5
    # |dt| up to 2 (reference time before the window) where the kernel assumes 1 without a hint: range checks, not trust
    pb = ebos.EventPlan.build(G(ev), (h, w), "before", True, tile=(32, 32))
    assert pb.dt_bound == 2.0
    for halo in (32, "auto"):
        assert rel(grads(pb, flow, halo).cpu().numpy(), grads(pb, flow, None).cpu().numpy()) < 2e-5, halo


def _blob_events(n, h, w, sigma, seed):
    rs = np.random.RandomState(seed)
    r = np.clip(np.rint(rs.normal(h / 2, sigma, n)), 0, h - 1)
    c = np.clip(np.rint(rs.normal(w / 2, sigma, n)), 0, w - 1)
    return np.stack([r, c, np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)


@pytest.mark.parametrize("shape,sigma,n", [((96, 128), 9.0, 60000), ((260, 346), 20.0, 200000), ((96, 128), 1000.0, 40000)])
def test_adaptive_work_items(ebos, shape, sigma, n):
    """Windows whose events sit in a few tiles: ebos_plan_parts cuts the heavy tiles into parts (splits = 0).  The part
    table must be a consistent partition, and IWE / variance / 2-DoF images must equal the one-part-per-tile result
    (fixed-point tile sums are exact; only the f32 slab combine order differs: rel-L2 < 1e-6) and the oracle (< 1e-4)."""
    h, w = shape
    ev = _blob_events(n, h, w, sigma, seed=41) if sigma < 100 else O.synth_events(n, h, w, seed=41)  # blob | uniform
    plan = ebos.EventPlan.build(G(ev), shape, "first", True, tile="auto")
    th, tw = plan.tile
    n_tiles = -(-h // th) * -(-w // tw)
    pt = plan.part_table.cpu().numpy()
    part_off, item_tile, item_part = pt[:n_tiles + 1], pt[n_tiles + 1:3 * n_tiles + 1], pt[3 * n_tiles + 1:]
    used = part_off[-1]
    assert part_off[0] == 0 and np.all(np.diff(part_off) >= 1) and n_tiles <= used <= 2 * n_tiles
    assert np.all(item_tile[used:] == -1) and np.all(item_tile[:used] >= 0)
    slabs = part_off[item_tile[:used]] + item_part[:used]
    assert sorted(slabs.tolist()) == list(range(used))                       # every (tile, part) exactly once
    assert np.all(item_part[:used] < np.diff(part_off)[item_tile[:used]])
    loads = np.diff(plan.key_offsets.cpu().numpy()[::th * tw])
    per_item = np.ceil(loads[item_tile[:used]] / np.diff(part_off)[item_tile[:used]])
    if sigma < 100:
        assert np.all(np.diff(per_item) <= 0)                                # heaviest first (the launch order is the schedule)
        assert used > n_tiles and plan.resolve_splits(None) == 0             # the blob's tiles are split ...
        assert per_item.max() < 0.5 * loads.max()
    else:
        assert used == n_tiles and plan.resolve_splits(None) == 1            # ... a uniform window is not
    flow = G(O.synth_dense_flow(h, w, seed=42, max_val=6.0)).float()
    one = plan.iwe_dense(flow, splits=1)
    ada = plan.iwe_dense(flow, splits=0)
    assert rel(ada.cpu().numpy(), one.cpu().numpy()) < 1e-6
    expect = O.iwe_dense(torch.from_numpy(ev), torch.from_numpy(flow.cpu().numpy().astype(np.float64)), shape)
    assert rel(ada.cpu().numpy(), expect.numpy()) < 1e-4
    v0, v1 = plan.contrast_dense(flow, splits=0).item(), plan.contrast_dense(flow, splits=1).item()
    assert abs(v0 - v1) <= 1e-6 * abs(v1)
    thetas = G(np.array([[2.0, -1.0], [-4.0, 3.0]])).float()
    assert rel(plan.iwe_2dof(thetas, splits=0).cpu().numpy(), plan.iwe_2dof(thetas, splits=1).cpu().numpy()) < 1e-6
    assert rel(plan.variance_2dof(thetas, splits=0).cpu().numpy(), plan.variance_2dof(thetas, splits=1).cpu().numpy()) < 1e-6
    # gradient through the default (adaptive) path against the oracle's autograd
    ft = torch.from_numpy(flow.cpu().numpy().astype(np.float64)).requires_grad_(True)
    torch.var(O.iwe_dense(torch.from_numpy(ev), ft, shape)).backward()
    fg = flow.clone().requires_grad_(True)
    plan.contrast_dense(fg).backward()
    assert rel(fg.grad.cpu().numpy(), ft.grad.numpy()) < 1e-3


@pytest.mark.parametrize("splits", [1, 0, 3])
def test_hot_pixels_overflow_high_and_low_fields(ebos, splits):
    """Cells that collect tens of thousands of events wrap the 32-bit fixed-point fields many times over, in the low
    AND in the high field of a paired word (a high-field wrap carries out of the 64-bit word and is invisible to a
    checksum modulo 2^32 -- the bug this test pins).  The exact f64 redo must take over: rel-L2 < 1e-6 vs the oracle,
    mass conserved."""
    h, w = 96, 128
    rs = np.random.RandomState(13)
    n = 300000
    r, c = rs.randint(0, h, n), rs.randint(0, w, n)
    hot = [(40, 61), (40, 62), (17, 100), (70, 7)]            # even and odd columns: both planes, both fields
    idx = rs.choice(n, 4 * 50000, replace=False)
    r[idx] = np.repeat([p[0] for p in hot], 50000)
    c[idx] = np.repeat([p[1] for p in hot], 50000)
    ev = np.stack([r, c, np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
    flow = rs.uniform(-1.5, 1.5, (2, h, w))
    flow[:, 40, 61] = 0.0                                       # all 50 000 events of this pixel land on one cell
    expect = O.iwe_dense(torch.from_numpy(ev), torch.from_numpy(flow), (h, w)).numpy()
    assert expect.max() >= 50000
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    got = plan.iwe_dense(G(flow).float(), splits=splits).cpu().numpy()
    assert rel(got, expect) < 1e-6
    assert abs(got.sum() - expect.sum()) < 1e-6 * expect.sum()
    var = plan.contrast_dense(G(flow).float(), splits=splits).item()
    assert abs(var - expect.var(ddof=1)) < 1e-5 * expect.var(ddof=1)


def test_fuzz_fused_path_against_oracle(ebos):
    """Seeded fuzz over the knobs that interact in the tile-private pipeline: image size (tiles cut by the border), tile
    configuration, halo (taps beyond it spill), event clustering (hot pixels, blobs, borders), flow magnitude (beyond the
    halo, out of the image), padding, omit_boundary, splits (uniform, adaptive).  IWE rel-L2 < 1e-4, variance rel
    < 1e-5, flow gradient rel-L2 < 1e-3 against the fp64 oracle -- the north_star tolerances, at every flow amplitude
    (events ON a kink of the piecewise-linear vote are taken out first, see _off_the_kinks)."""
    from event_based_bos_amd import _hip

    configs = _hip.slab_configs()
    rs = np.random.RandomState(int(os.environ.get("EBOS_FUZZ_SEED", 2024)))  # soak runs override the seed
    unfiltered = []
    for case in range(64):
        h, w = int(rs.randint(20, 150)), int(rs.randint(20, 200))
        th, tw, halo = configs[rs.randint(len(configs))]
        n = int(rs.choice([2, 7, 500, 5000, 60000]))  # (a single event trips the squeeze() quirk of the reference path)
        kind = rs.randint(5)
        if kind == 4:     # fractional source coordinates (undistorted events): the 12 B/event (x, y, dt) format.  Multiples
            # of 1/64 are exact in f32: an event whose f64 coordinate sits within f32 rounding of an integer would be
            # looked up at another source pixel by ANY f32 path (the flow is random per pixel), one lost event = 2e-3
            r, c = rs.randint(0, h * 64, n) / 64.0, rs.randint(0, w * 64, n) / 64.0
        elif kind == 0:
            r, c = rs.randint(0, h, n), rs.randint(0, w, n)
        elif kind == 1:   # blob
            r = np.clip(np.rint(rs.normal(h / 2, 4, n)), 0, h - 1)
            c = np.clip(np.rint(rs.normal(w / 3, 6, n)), 0, w - 1)
        elif kind == 2:   # a few hot pixels
            k = rs.randint(0, 5, n)
            r, c = np.array([0, h - 1, h // 2, 3, h // 2])[k], np.array([0, w - 1, w // 2, w - 2, w // 2 + 1])[k]
        else:             # borders
            r = rs.choice([0, 1, h - 2, h - 1], n)
            c = rs.randint(0, w, n)
        t = np.sort(rs.uniform(3.0, 3.5, n))
        ev = np.stack([r, c, t, rs.randint(0, 2, n)], 1).astype(np.float64)
        amp = float(rs.choice([0.0, 0.7, 5.0, 40.0, 90.0]))
        flow = rs.uniform(-amp, amp, (2, h, w))
        pad = int(rs.choice([0, 0, 4]))
        omit = bool(rs.randint(2))
        splits = int(rs.choice([0, 1, 2, 5]))
        direction = ["first", "middle", "last", 0.3][rs.randint(4)]
        if case % 3 == 2:  # every third case: run-time windows per tile (EBOS_HALO_AUTO), bounded by the tile's largest built halo
            halo = "auto"
        tag = f"case {case}: {h}x{w} tile {th}x{tw} halo {halo} n {n} kind {kind} amp {amp} pad {pad} splits {splits} {direction}"
        ev_raw = ev
        ev = _off_the_kinks(ev, flow, direction, amp)
        if len(ev) < len(ev_raw) and n >= 5000 and amp > 0 and kind != 4:
            # the SAME case on the un-filtered stream: values keep their bars, the flow-gradient error is recorded (reported
            # at the end; at a kink of the piecewise-linear vote f32 and f64 may take different one-sided derivatives)
            fr = torch.from_numpy(flow).requires_grad_(True)
            ex = O.iwe_dense(torch.from_numpy(ev_raw), fr, (h, w), pad=(pad, pad), direction=direction)
            cr = ex[1:-1, 1:-1] if omit else ex
            pr = ebos.EventPlan.build(G(ev_raw), (h, w), direction, True, tile=(th, tw))
            fgr = G(flow).float().requires_grad_(True)
            vr = pr.contrast_dense(fgr, "image_variance", omit, pad=(pad, pad), halo=halo, splits=splits)
            assert abs(vr.item() - torch.var(cr).item()) <= 1e-5 * abs(torch.var(cr).item()) + 1e-9, tag
            vr.backward()
            torch.var(cr).backward()
            if float(fr.grad.norm()) > 0:
                unfiltered.append((float((fgr.grad.cpu().double() - fr.grad).norm()) / float(fr.grad.norm()),
                                   len(ev_raw) - len(ev), len(ev_raw), amp))
        n = len(ev)
        tev = torch.from_numpy(ev)
        ft = torch.from_numpy(flow).requires_grad_(True)
        expect = O.iwe_dense(tev, ft, (h, w), pad=(pad, pad), direction=direction)
        crop = expect[1:-1, 1:-1] if omit else expect
        v_ref = torch.var(crop) if crop.numel() > 1 else None
        # every other case on a LEAN plan (emit="compact": two-level counting sort, no SoA / perm); cases that go on to use per-event
        # weights (case % 3 == 0) need the full build, fractional sources (kind 4) fall back to it by themselves
        emit = "compact" if (case % 2 == 1 and case % 3 != 0) else "full"
        plan = ebos.EventPlan.build(G(ev), (h, w), direction, True, tile=(th, tw), emit=emit)
        assert plan.compact == (kind != 4) and plan.lean == (emit == "compact" and kind != 4), tag
        fg = G(flow).float().requires_grad_(True)
        got = plan.iwe_dense(fg, pad=(pad, pad), halo=halo, splits=splits)
        # (relative to the image, but not below 5 % of ONE event's unit mass: with a couple of events pushed almost entirely
        # out of the image, what is left inside is a tap of weight ~1e-3 whose f32 error is that of a full-weight tap)
        scale = max(float(expect.detach().norm()), 0.05)
        assert float((got.detach().cpu().double() - expect.detach()).norm()) / scale < 1e-4, tag
        if v_ref is not None and float(v_ref.detach()) > 0:
            v = plan.contrast_dense(fg, "image_variance", omit, pad=(pad, pad), halo=halo, splits=splits)
            assert abs(v.item() - v_ref.item()) <= 1e-5 * abs(v_ref.item()) + 1e-9, tag
            v.backward()
            v_ref.backward()
            gn = float(ft.grad.detach().norm())
            if gn > 0 and amp > 0:  # at zero flow every event sits on the kink of the bilinear vote
                assert float((fg.grad.cpu().double() - ft.grad).norm()) / gn < 1e-3, tag
        # per-event weights (weights and their gradient in input order).  Negative weights: the slices that hold one take the f64 LDS
        # accumulators; every other weighted case: weights >= 0 with exact zeros among them -- the fixed-point loops in units of the
        # slice's max |w| (integer source pixels: the weighted lean loop on the compact format)
        if case % 3 == 0:
            wts = rs.uniform(-1.0, 2.0, n)
            if case % 6 == 0:
                wts = np.abs(wts) * (rs.uniform(size=n) > 0.1)
            wt = torch.from_numpy(wts).requires_grad_(True)
            ft2 = torch.from_numpy(flow).requires_grad_(True)
            exp_w = O.iwe_dense(tev, ft2, (h, w), pad=(pad, pad), direction=direction, weight=wt)
            probe = torch.from_numpy(rs.normal(size=tuple(exp_w.shape)))
            (exp_w * probe).sum().backward()
            wg = G(wts).float().requires_grad_(True)
            fg2 = G(flow).float().requires_grad_(True)
            got_w = plan.iwe_dense(fg2, pad=(pad, pad), weight=wg, halo=halo, splits=splits)
            assert float((got_w.detach().cpu().double() - exp_w.detach()).norm()) / max(float(exp_w.detach().norm()), 0.05) < 1e-4, tag
            (got_w * G(probe.numpy()).float()).sum().backward()
            assert float((wg.grad.cpu().double() - wt.grad).norm()) / max(float(wt.grad.norm()), 1e-12) < 1e-3, tag
            if amp > 0 and float(ft2.grad.norm()) > 0:
                assert float((fg2.grad.cpu().double() - ft2.grad).norm()) / float(ft2.grad.norm()) < 1e-3, tag
        # the 2-DoF model through the same tile-private kernels (UNIFORM variant)
        theta = rs.uniform(-amp - 1, amp + 1, 2)
        exp2 = O.iwe_2dof(tev, torch.from_numpy(theta), (h, w), pad=(pad, pad), direction=direction)
        got2 = plan.iwe_2dof(G(theta[None]).float(), pad=(pad, pad), halo=halo, splits=splits)[0]
        assert float((got2.cpu().double() - exp2).norm()) / max(float(exp2.norm()), 0.05) < 1e-4, tag
    if unfiltered:
        errs = sorted(e for e, *_ in unfiltered)
        worst = max(unfiltered)
        print(f"[fuzz] un-filtered streams ({len(errs)} cases, events near a kink kept): flow-gradient rel-L2 median "
              f"{errs[len(errs) // 2]:.2e}, max {worst[0]:.2e} ({worst[1]} of {worst[2]} events near a kink, amp {worst[3]}); "
              "filtered bar: 1e-3")
        assert worst[0] < 0.2  # (one-sided derivatives at kinks; a broken kernel is O(1))


def test_fuzz_plugin_surface_against_oracle(ebos):
    """Seeded fuzz of the drop-in layer (Warp.warp_event + EventImageConverter.create_iwe) over what a caller can vary:
    numpy / torch (CPU and GPU) inputs, f32 / f64, batched / un-batched, every direction, normalize_t, dense and 2-DoF
    models, image methods, scalar / per-event weights, padding, blur.  Warped events BIT-EXACT on equal dtype (f64: same
    op order), images rel-L2 <= 1e-12 in f64 and < 1e-5 in f32; output type, dtype and device follow the input."""
    rs = np.random.RandomState(int(os.environ.get("EBOS_FUZZ_SEED", 77)))
    for case in range(48):
        h, w = int(rs.randint(8, 40)), int(rs.randint(8, 50))
        n = int(rs.choice([2, 3, 50, 400]))
        batched = bool(rs.randint(2))
        b = int(rs.randint(1, 4)) if batched else 1
        dtype = [np.float64, np.float32][rs.randint(2)]
        kind = ["numpy", "torch_cpu", "torch_gpu"][rs.randint(3)]
        model = ["dense-flow", "2d-translation"][rs.randint(2)] if not batched else "dense-flow"  # 2-DoF is un-batched only
        direction = ["first", "middle", "last", 0.7, "before", "after"][rs.randint(6)]
        norm_t = bool(rs.randint(2))
        pad = int(rs.choice([0, 2]))
        evs = np.stack([np.stack([rs.uniform(0, h - 1e-3, n), rs.uniform(0, w - 1e-3, n), np.sort(rs.uniform(1.0, 1.4, n)),
                                  rs.randint(0, 2, n).astype(float)], 1) for _ in range(b)]).astype(dtype)
        flow = rs.uniform(-4, 4, (b, 2, h, w)).astype(dtype)
        theta = rs.uniform(-4, 4, 2).astype(dtype)
        if not batched:
            evs, flow = evs[0], flow[0]
        tag = f"case {case}: {h}x{w} n {n} b {b if batched else None} {dtype.__name__} {kind} {model} {direction} norm {norm_t} pad {pad}"

        def give(a):
            if kind == "numpy":
                return a
            t = torch.from_numpy(a)
            return t.cuda() if kind == "torch_gpu" else t

        def back(v):
            return v if isinstance(v, np.ndarray) else v.detach().cpu().numpy()

        warper = ebos.Warp((h, w), normalize_t=norm_t)
        motion = flow if model == "dense-flow" else theta
        warped, feat = warper.warp_event(give(evs), give(motion), model, direction=direction)
        assert set(feat) == {"iwe", "iwe_var", "iwe_grad", "time_image", "time_image_var"} or isinstance(feat, dict), tag
        assert isinstance(warped, np.ndarray) == (kind == "numpy"), tag
        if kind == "torch_gpu":
            assert warped.is_cuda, tag
        if model == "dense-flow":
            if kind == "numpy":
                expect = O.warp_dense_numpy(evs, flow, direction, norm_t)
            else:
                expect = O.warp_dense_torch(torch.from_numpy(evs), torch.from_numpy(flow), direction, norm_t).numpy()
        else:
            expect = back(O.warp_2dof(evs if kind == "numpy" else torch.from_numpy(evs), theta if kind == "numpy" else torch.from_numpy(theta),
                                      direction, norm_t))
        got = back(warped)
        assert got.dtype == dtype and got.shape == expect.shape, (tag, got.shape, expect.shape)
        assert np.array_equal(got, expect), (tag, float(np.abs(got - expect).max()))  # BIT-EXACT
        # images of the warped events
        ic = ebos.EventImageConverter((h, w), outer_padding=pad)
        method = ["bilinear_vote", "count", "polarity"][rs.randint(3)]
        sigma = int(rs.choice([0, 0, 1]))
        per_event = bool(rs.randint(2)) and method != "count"
        if kind != "numpy":
            if method == "polarity":
                method = "bilinear_vote"  # the tensor branch has bilinear_vote and count (src/event_image_converter.py:383-397)
        wt = rs.uniform(0.5, 1.5, got.shape[:-1]).astype(dtype) if per_event else 1.0
        if kind == "numpy":
            img = ic.create_image_from_events_numpy(got, method, weight=wt, sigma=sigma)
            exp_img = O.create_image_numpy(expect, (h + 2 * pad, w + 2 * pad), (pad, pad), method, wt, sigma)
        else:
            wtt = give(wt) if per_event else 1.0
            img = ic.create_image_from_events_tensor(give(got), method, weight=wtt, sigma=sigma)
            te = torch.from_numpy(expect)
            if method == "count":
                exp_img = O.count_events_torch(te, (h + 2 * pad, w + 2 * pad), (pad, pad))
            else:
                exp_img = O.bilinear_vote_torch(te, (h + 2 * pad, w + 2 * pad), (pad, pad), torch.from_numpy(wt) if per_event else 1.0)
            if sigma > 0:
                exp_img = O.gaussian_blur3_torch(exp_img, float(sigma))
            exp_img = exp_img.numpy()
        gi = back(img)
        assert gi.shape == exp_img.shape, (tag, method, gi.shape, exp_img.shape)
        # f64 inputs: same arithmetic, atomics only reorder the sum.  f32 inputs: the reference forms the four weights in
        # f32 (numpy: then sums them into an f64 image); the kernels form them in the image's precision
        tol = 1e-12 if dtype == np.float64 else 1e-5
        assert rel(gi, exp_img) <= tol, (tag, method, sigma, rel(gi, exp_img))


@pytest.mark.parametrize("omit,pad", [(False, 0), (True, 2)])
def test_variance_finalize_as_side_job_of_the_regulariser_pass(ebos, omit, pad):
    """ebos_iwe_dense_slab_f32(want_variance = 2) leaves (sum, sum of squares) partials in the workspace
    (ebos_iwe_slab_partials locates them); ebos_flow_regularisers_f32 reduces them as a side job.  Same summation order
    as the finalize kernel: variance and (mean, M) BIT-EXACT."""
    import ctypes as C

    from event_based_bos_amd import event_plan as EP

    h, w = 96, 128
    ev = O.synth_events(40000, h, w, seed=61)
    flow = G(O.synth_dense_flow(h, w, seed=62, max_val=5.0)).float()
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    lib = ebos.load_library()
    _, var1, mom1 = EP._launch_iwe_dense_slab(plan, flow, None, (pad, pad), 32, 1, True, omit)
    ws = EP._workspace(plan, (pad, pad), 32, 1)
    iwe = torch.empty((h + 2 * pad, w + 2 * pad), device=dev())
    EP.check(lib.ebos_iwe_dense_slab_f32(EP.ptr(plan.x), EP.ptr(plan.y), EP.ptr(plan.dt), None, *plan._compact_ptrs(),
                                         EP.ptr(plan.key_offsets), plan.n, EP.ptr(flow), h, w, plan.tile[0], plan.tile[1], 32, 1,
                                         pad, pad, EP.ptr(ws), ws.numel(), EP.ptr(iwe), 2, int(omit), None, None, None,
                                         EP.stream_ptr()), "slab")
    off, n_parts, n_px = C.c_size_t(), C.c_int64(), C.c_int64()
    EP.check(lib.ebos_iwe_slab_partials(h, w, plan.tile[0], plan.tile[1], 32, 1, pad, pad, int(omit), C.byref(off),
                                        C.byref(n_parts), C.byref(n_px)), "partials")
    lo = 1 if omit else 0
    assert n_px.value == (h + 2 * pad - 2 * lo) * (w + 2 * pad - 2 * lo) and n_parts.value >= 1
    d = torch.empty_like(flow)
    parts = torch.zeros(lib.ebos_flow_regularisers_partials(), dtype=torch.float64, device=dev())
    var2 = torch.zeros(1, device=dev())
    mom2 = torch.zeros((1, 2), dtype=torch.float64, device=dev())
    EP.check(lib.ebos_flow_regularisers_f32(EP.ptr(flow), h, w, 0.5, 0.25, EP.ptr(d), EP.ptr(parts), ws.data_ptr() + off.value,
                                            n_parts.value, n_px.value, EP.ptr(var2), EP.ptr(mom2), EP.stream_ptr()), "reg")
    assert torch.equal(var2, var1) and torch.equal(mom2, mom1)


def test_many_dense_hypotheses_on_streams(ebos):
    """variance_dense_many (K independent dense flows dealt to 3 HIP streams) == K single evaluations, bit for bit."""
    h, w = 96, 128
    ev = O.synth_events(50000, h, w, seed=71)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    flows = torch.stack([G(O.synth_dense_flow(h, w, seed=80 + k, max_val=3.0 + 4 * k)).float() for k in range(7)])
    many = plan.variance_dense_many(flows)
    many2 = plan.variance_dense_many(flows, omit_boundary=True, n_streams=2)
    for k in range(7):
        assert many[k].item() == plan.contrast_dense(flows[k]).item()
        assert many2[k].item() == plan.contrast_dense(flows[k], omit_boundary=True).item()
    with pytest.raises(ValueError):
        plan.variance_dense_many(flows[:, :, :-1])


def test_variance_and_grad_without_autograd(ebos):
    """variance_and_grad_dense == contrast_dense(flow) + backward(), same kernels: value bit-exact, gradient to 1e-6
    (the f64 LDS adds of the backward kernel are not ordered)."""
    h, w = 96, 128
    ev = O.synth_events(50000, h, w, seed=91)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    for omit, pad in ((False, 0), (True, 2)):
        flow = G(O.synth_dense_flow(h, w, seed=92, max_val=6.0)).float()
        f = flow.clone().requires_grad_(True)
        v = plan.contrast_dense(f, "image_variance", omit, pad=(pad, pad))
        v.backward()
        var, grad = plan.variance_and_grad_dense(flow, omit, pad=(pad, pad))
        assert var.item() == v.item() and not grad.requires_grad
        assert rel(grad.cpu().numpy(), f.grad.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("case", [
    # (H, W, tile, halo, patch, slide, n, blob, splits)
    (135, 240, (45, 80), 32, (24, 32), (24, 32), 60000, False, 1),
    (135, 240, (45, 80), 32, (30, 40), (15, 20), 60000, False, 1),      # overlapping patches
    (100, 150, (45, 80), 32, (17, 23), (17, 23), 30000, False, 1),      # odd sizes, tiles cut by the border
    (96, 128, (32, 64), 32, (16, 16), (16, 16), 30000, False, 1),
    (96, 128, (32, 32), 16, (24, 32), (8, 8), 30000, False, 1),         # many cells per tile
    (135, 240, (45, 80), 32, (24, 32), (24, 32), 120000, True, 0),      # adaptive work items (clustered events)
    (135, 240, (45, 80), 32, (24, 32), (24, 32), 2, False, 1),          # nearly empty
])
@pytest.mark.parametrize("terms", ["var", "var+norm", "var+reg", "gm"])
def test_grid_sampling_kernels_match_upsample_then_dense(ebos, case, terms):
    """ebos_iwe_patch_slab_f32 / ebos_iwe_patch_tiled_bwd_f32 / ebos_patch_grad_combine_adam_f32 (the event kernels evaluate
    the patch grid -> dense map per tile in LDS) against (a) the materialised route upsample -> dense kernels -> adjoint
    (IWE and loss rel 1e-6, gradient rel-L2 1e-5) and (b) the fp64 oracle, autograd through upsample_patch_flow + iwe_dense +
    cost (IWE rel-L2 < 1e-4, loss < 1e-5, d loss / d theta rel-L2 < 1e-3)."""
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    H, W, tile, halo, patch, slide, n, blob, splits = case
    lib = ebos.load_library()
    if not lib.ebos_patch_fused_supported(tile[0], tile[1], halo, slide[0], slide[1]):
        pytest.skip("configuration outside ebos_patch_fused_supported")
    rs = np.random.RandomState(123 + H + n)
    if blob:
        r = np.clip(np.rint(rs.normal(H / 2, 6, n)), 0, H - 1)
        c = np.clip(np.rint(rs.normal(W / 3, 9, n)), 0, W - 1)
    else:
        r, c = rs.randint(0, H, n), rs.randint(0, W, n)
    ev = np.stack([r, c, np.sort(rs.uniform(1.0, 1.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
    gh, gw = len(np.arange(0, H - patch[0] + slide[0], slide[0])), len(np.arange(0, W - patch[1] + slide[1], slide[1]))
    theta = rs.uniform(-12, 12, (2, gh, gw))
    w_var, w_gm = (0.0, 1.5) if terms == "gm" else (2.0, 0.0)
    # the regularisers are evaluated inside the backward kernel from the tile's flow (no dense field, no regulariser launch)
    w_norm, w_tv = {"var+reg": (0.02, 0.03), "var+norm": (0.05, 0.0)}.get(terms, (0.0, 0.0))
    ev = _off_the_kinks_patch(ev, theta, (H, W), patch, slide)
    plan = ebos.EventPlan.build(G(ev), (H, W), "first", True, tile=tile)
    if splits == 0:
        assert plan.resolve_splits(None) == 0, "the clustered window should have split a tile"
    out = {}
    for grid in (True, False):
        loop = FusedPatchLoop(plan, patch, slide, G(theta).float(), w_var, w_norm, w_tv, halo=halo, capacity=1, splits=splits,
                              w_gradient_magnitude=w_gm, sample_grid=grid)
        assert loop.sample_grid == grid and (loop.d_dense is None) == grid
        assert loop.fuse_norm == (grid and terms in ("var+norm", "var+reg")) and (loop.dense is None) == grid
        loss, grad = loop.value_and_grad(G(theta).float())
        out[grid] = (loop.iwe.cpu().double().numpy(), float(loss), grad.cpu().double().numpy())
    assert rel(out[True][0], out[False][0]) < 1e-6
    assert abs(out[True][1] - out[False][1]) <= 1e-6 * abs(out[False][1])
    assert rel(out[True][2], out[False][2]) < 1e-5
    # the oracle
    tt = torch.from_numpy(theta).requires_grad_(True)
    dense = O.upsample_patch_flow(tt, (H, W), patch, slide)
    iwe = O.iwe_dense(torch.from_numpy(ev), dense, (H, W), direction="first")
    sob = O.sobel3(iwe) / 8.0
    loss = -(w_gm * torch.mean(sob[0] ** 2 + sob[1] ** 2) if w_gm else w_var * torch.var(iwe))
    if w_norm or w_tv:
        loss = loss + w_norm * O.flow_norm(dense) + w_tv * O.image_gradient_tv(dense, torch.ones((H, W), dtype=torch.float64))
    loss.backward()
    assert rel(out[True][0], iwe.detach().numpy()) < 1e-4
    assert abs(out[True][1] - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-9
    if n > 100:
        assert rel(out[True][2], tt.grad.numpy()) < 1e-3


def test_grid_sampling_refuses_unsupported_configurations(ebos):
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    lib = ebos.load_library()
    assert lib.ebos_patch_fused_supported(45, 80, 32, 24, 32) == 1
    assert lib.ebos_patch_fused_supported(45, 80, 32, 2, 2) == 0      # too many cells per tile
    assert lib.ebos_patch_fused_supported(64, 64, 32, 24, 32) == 0    # no LDS left for the tile's flow
    assert lib.ebos_patch_fused_supported(45, 80, 16, 24, 32) == 1    # the small-displacement configuration
    assert lib.ebos_patch_fused_supported(45, 80, 24, 24, 32) == 0    # not a built configuration
    ev = O.synth_events(5000, 128, 128, seed=3)
    plan = ebos.EventPlan.build(G(ev), (128, 128), "first", True, tile=(64, 64))
    with pytest.raises(ValueError):
        FusedPatchLoop(plan, (32, 32), (32, 32), torch.zeros((2, 4, 4)), 1.0, sample_grid=True)
    loop = FusedPatchLoop(plan, (32, 32), (32, 32), torch.zeros((2, 4, 4)), 1.0)   # default: falls back to the dense route
    assert loop.sample_grid is False
    loop.run(1)


def test_fuzz_grid_sampling_route(ebos):
    """Seeded fuzz of the grid-sampling event kernels over what interacts there: image size (tiles cut by the border), tile
    configuration, patch size and sliding window (overlapping patches, odd sizes, many / few cells per tile, grids smaller than
    a tile), event clustering (adaptive work items), flow magnitude (beyond the halo), omit_boundary, padding of the image,
    theta_mask, objective terms.  Against the materialised route: IWE / loss 1e-6, gradient rel-L2 1e-5 (both f32, same
    expressions); against the fp64 oracle: IWE rel-L2 < 1e-4, loss < 1e-5, d loss / d theta rel-L2 < 1e-3."""
    from event_based_bos_amd import _hip
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    lib = ebos.load_library()
    configs = [c for c in _hip.slab_configs()]
    rs = np.random.RandomState(int(os.environ.get("EBOS_FUZZ_SEED", 4242)))
    done = 0
    for case in range(60):
        th, tw, halo = configs[rs.randint(len(configs))]
        H, W = int(rs.randint(24, 200)), int(rs.randint(24, 260))
        patch = (int(rs.randint(4, 40)), int(rs.randint(4, 48)))
        slide = (int(rs.randint(max(2, patch[0] // 3), patch[0] + 1)), int(rs.randint(max(2, patch[1] // 3), patch[1] + 1)))
        if patch[0] > H or patch[1] > W or not lib.ebos_patch_fused_supported(th, tw, halo, slide[0], slide[1]):
            continue
        n = int(rs.choice([2, 300, 20000, 80000]))
        kind = rs.randint(3)
        if kind == 0:
            r, c = rs.randint(0, H, n), rs.randint(0, W, n)
        elif kind == 1:   # blob: splits tiles of the adaptive plan
            r = np.clip(np.rint(rs.normal(H / 2, 5, n)), 0, H - 1)
            c = np.clip(np.rint(rs.normal(W / 3, 7, n)), 0, W - 1)
        else:             # borders
            r, c = rs.choice([0, 1, H - 2, H - 1], n), rs.randint(0, W, n)
        ev = np.stack([r, c, np.sort(rs.uniform(2.0, 2.4, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
        gh, gw = len(np.arange(0, H - patch[0] + slide[0], slide[0])), len(np.arange(0, W - patch[1] + slide[1], slide[1]))
        amp = float(rs.choice([0.5, 6.0, 25.0, 60.0]))
        theta = rs.uniform(-amp, amp, (2, gh, gw))
        terms = ["var", "var+norm", "var+reg", "gm"][rs.randint(4)]
        w_var, w_gm = (0.0, 1.5) if terms == "gm" else (2.0, 0.0)
        w_norm, w_tv = {"var+reg": (0.02, 0.03), "var+norm": (0.05, 0.0)}.get(terms, (0.0, 0.0))
        omit = bool(rs.randint(2))
        pad = int(rs.choice([0, 0, 3]))
        splits = int(rs.choice([0, 1])) if kind == 1 and n >= 20000 else 1
        mask = (rs.uniform(size=(gh, gw)) > 0.3).astype(np.float32) if rs.randint(3) == 0 else None
        if case % 3 == 2 and lib.ebos_patch_fused_supported(th, tw, 32, slide[0], slide[1]):
            halo = "auto"  # run-time windows per tile (the grid-sampling kernels take the bound from the cells a tile interpolates)
        tag = f"case {case}: {H}x{W} tile {th}x{tw} halo {halo} patch {patch} slide {slide} grid {gh}x{gw} n {n} kind {kind} amp {amp} {terms} omit {omit} pad {pad} splits {splits} mask {mask is not None}"
        ev = _off_the_kinks_patch(ev, theta, (H, W), patch, slide)
        plan = ebos.EventPlan.build(G(ev), (H, W), "first", True, tile=(th, tw))
        out = {}
        for grid in (True, False):
            loop = FusedPatchLoop(plan, patch, slide, G(theta).float(), w_var, w_norm, w_tv, omit_boundary=omit, pad=pad, halo=halo,
                                  capacity=1, splits=splits, w_gradient_magnitude=w_gm, sample_grid=grid,
                                  theta_mask=None if mask is None else G(mask))
            loss, grad = loop.value_and_grad(G(theta).float())
            out[grid] = (loop.iwe.cpu().double().numpy(), float(loss), grad.cpu().double().numpy())
        # the two routes evaluate the same f32 expressions, but not bit for bit (FMA contraction differs between the kernels):
        # a flow of 60 px carries 4e-6 px per ulp
        scale = max(np.linalg.norm(out[False][0]), 1e-12)
        assert np.linalg.norm(out[True][0] - out[False][0]) / scale < 2e-5, tag
        assert abs(out[True][1] - out[False][1]) <= 2e-5 * abs(out[False][1]) + 1e-12, tag
        gn = np.linalg.norm(out[False][2])
        if gn > 0 and n > 100:
            assert np.linalg.norm(out[True][2] - out[False][2]) / gn < 1e-4, tag
        tt = torch.from_numpy(theta).requires_grad_(True)
        dense = O.upsample_patch_flow(tt, (H, W), patch, slide)
        iwe = O.iwe_dense(torch.from_numpy(ev), dense, (H, W), pad=(pad, pad), direction="first")
        crop = iwe[1:-1, 1:-1] if omit else iwe
        sob = O.sobel3(iwe) / 8.0
        mag = sob[0] ** 2 + sob[1] ** 2
        loss = -(w_gm * torch.mean(mag[1:-1, 1:-1] if omit else mag) if w_gm else w_var * torch.var(crop))
        if w_norm or w_tv:
            loss = loss + w_norm * O.flow_norm(dense) + w_tv * O.image_gradient_tv(dense, torch.ones((H, W), dtype=torch.float64))
        loss.backward()
        want = tt.grad.numpy() * (1.0 if mask is None else mask.astype(np.float64))
        assert np.linalg.norm(out[True][0] - iwe.detach().numpy()) / max(float(iwe.detach().norm()), 1e-12) < 1e-4, tag
        # 1e-5 is the bar of the +-30 px regime; beyond it (and with a handful of events) the f32 displacement error shows
        assert abs(out[True][1] - loss.item()) <= (1e-5 if amp <= 30.0 else 5e-5) * abs(loss.item()) + 1e-9, tag
        if n > 100 and np.linalg.norm(want) > 0:
            assert np.linalg.norm(out[True][2] - want) / np.linalg.norm(want) < 1e-3, tag
        done += 1
    assert done >= 25, done


@pytest.mark.parametrize("shape,omit", [((37, 53), False), ((37, 53), True), ((96, 130), False), ((16, 64), True), ((3, 5), False)])
def test_fused_gradient_magnitude_pass_equals_the_two_pass_kernels(ebos, shape, omit):
    """``ebos_gradient_magnitude_fused_f32`` (one LDS-tiled pass: value + gradient image, f32 stencils) against the two stand-alone
    fp64 kernels it replaces in the objective and the CPU oracle's autograd: value and gradient image to f32 round-off -- on sizes
    that are no multiple of the 16 x 64 tile, with and without the boundary ring, down to 3 x 5."""
    from event_based_bos_amd import _hip

    lib = _hip.require_gpu()
    h, w = shape
    img = torch.from_numpy(np.random.RandomState(5).gamma(2.0, 3.0, (h, w))).float().cuda()
    up = torch.tensor([-1.75], dtype=torch.float32, device="cuda")
    n = int(lib.ebos_gradient_magnitude_fused_partials(h, w))
    out, d_img = torch.empty(1, device="cuda"), torch.empty_like(img)
    partials = torch.empty(n, dtype=torch.float64, device="cuda")
    _hip.check(lib.ebos_gradient_magnitude_fused_f32(img.data_ptr(), h, w, int(omit), up.data_ptr(), out.data_ptr(), d_img.data_ptr(),
                                                     partials.data_ptr(), n, _hip.stream_ptr()), "fused")
    ref_val = ebos.ops.gradient_magnitude(img, omit)
    x = img.clone().requires_grad_(True)
    (ebos.ops.gradient_magnitude(x, omit) * up[0]).backward()
    assert abs(out.item() - ref_val.item()) <= 1e-6 * abs(ref_val.item())
    assert rel(d_img.cpu().numpy(), x.grad.cpu().numpy()) < 2e-6 and (d_img - x.grad).abs().max().item() < 1e-5 * x.grad.abs().max().item()
    # against the CPU oracle's autograd (fp64)
    xo = img.double().cpu().requires_grad_(True)
    (O.gradient_magnitude(xo, omit, direction="maximize") * float(up[0])).backward()
    assert rel(d_img.cpu().numpy(), xo.grad.numpy()) < 2e-6
    # value only (d_image NULL): the same value from the lighter form of the pass (an objective evaluation needs no gradient image)
    out2 = torch.empty(1, device="cuda")
    _hip.check(lib.ebos_gradient_magnitude_fused_f32(img.data_ptr(), h, w, int(omit), up.data_ptr(), out2.data_ptr(), None,
                                                     partials.data_ptr(), n, _hip.stream_ptr()), "fused value")
    assert abs(out2.item() - out.item()) <= 1e-7 * abs(out.item())
    with pytest.raises(RuntimeError):
        _hip.check(lib.ebos_gradient_magnitude_fused_f32(img.data_ptr(), h, w, int(omit), None, out.data_ptr(), d_img.data_ptr(),
                                                         partials.data_ptr(), n - 1 if n > 1 else 0, _hip.stream_ptr()), "fused")


def test_gradient_magnitude_objective_is_one_native_call_with_an_eager_backward(ebos):
    """``plan.contrast_dense(flow, "gradient_magnitude")`` (BASELINE configs[2]): value and flow gradient by
    ``ebos_gradient_magnitude_dense_job_f32`` -- same numbers as the image -> ``ops.gradient_magnitude`` -> autograd route, an
    ``_EagerLoss`` whose ``backward()`` needs no engine, and the cost class of the reference idiom takes the same short cut."""
    from event_based_bos_amd.event_plan import _EagerLoss

    h, w, n = 96, 128, 30_000
    ev = O.synth_events(n, h, w, seed=15)
    fl = G(O.synth_dense_flow(h, w, seed=16, max_val=6.0), torch.float32)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto")
    f_ref = fl.clone().requires_grad_(True)
    ref = ebos.ops.gradient_magnitude(plan.iwe_dense(f_ref), False)
    (-ref).backward()
    f = fl.clone().requires_grad_(True)
    loss = -plan.contrast_dense(f, "gradient_magnitude")
    assert type(loss) is _EagerLoss
    loss.backward()
    assert abs(loss.item() + ref.item()) <= 1e-6 * abs(ref.item())
    assert rel(f.grad.cpu().numpy(), f_ref.grad.cpu().numpy()) < 1e-6
    for omit in (False, True):     # value only (no gradient wanted): the Sobel pass finalizes itself
        with torch.no_grad():
            v = plan.contrast_dense(fl, "gradient_magnitude", omit)
        assert abs(v.item() - ebos.ops.gradient_magnitude(plan.iwe_dense(fl), omit).item()) <= 1e-6 * abs(v.item())
    # fp64 CPU oracle
    fo = torch.from_numpy(O.synth_dense_flow(h, w, seed=16, max_val=6.0)).float().double().requires_grad_(True)
    lo = O.gradient_magnitude(O.iwe_dense(torch.from_numpy(ev), fo, (h, w)))
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-5 * abs(lo.item())
    # the reference idiom: warp_event -> create_iwe -> gradient_magnitude cost -> backward, fused at the cost step
    wp, ic = ebos.Warp((h, w), normalize_t=True), ebos.EventImageConverter((h, w))
    cost = ebos.costs.functions["gradient_magnitude"]()
    f2 = fl.clone().requires_grad_(True)
    warped, _ = wp.warp_event(G(ev, torch.float32), f2, "dense-flow", "first")
    l2 = cost.calculate({"iwe": ic.create_iwe(warped, "bilinear_vote", sigma=0), "omit_boundary": False})
    l2.backward()
    assert abs(l2.item() - loss.item()) <= 1e-6 * abs(loss.item())
    print("idiom gradient vs plan gradient", rel(f2.grad.cpu().numpy(), f.grad.cpu().numpy()), ebos.fusion.stats)
    assert rel(f2.grad.cpu().numpy(), f.grad.cpu().numpy()) < 2e-5   # (the idiom's plan has its own tile: another order of f32 sums)


def test_deferred_results_read_after_an_optimizer_step_hold_the_old_flow(ebos, monkeypatch):
    """The default (EBOS_FUSE_API=lazy) defers the idiom's warped events and image; ``warp_event`` copies the flow on the device, so
    a result first read AFTER ``optimizer.step()`` equals what the reference's eager call returned at call time
    (src/warp.py:330-342: values of the flow as it was) -- bit for bit against this package's eager path, to f32 round-off against
    the fp64 oracle -- for several iterations whose unread results are all kept alive (the snapshot ring is reused)."""
    monkeypatch.delenv("EBOS_FUSE_API", raising=False)
    h, w, n = 90, 120, 40_000
    ev_np = O.synth_events(n, h, w, seed=31)
    ev = G(ev_np, torch.float32)
    wp, ic = ebos.Warp((h, w), normalize_t=True), ebos.EventImageConverter((h, w))
    cost = ebos.costs.functions["image_variance"]()
    fl = G(O.synth_dense_flow(h, w, seed=32, max_val=5.0), torch.float32).requires_grad_(True)
    opt = torch.optim.SGD([fl], lr=50.0)
    kept = []
    for it in range(5):
        opt.zero_grad()
        flow_then = fl.detach().clone()
        warped, _ = wp.warp_event(ev, fl, "dense-flow", "first")
        iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
        loss = cost.calculate({"iwe": iwe, "omit_boundary": False})
        loss.backward()
        assert type(warped) is ebos.fusion.LazyWarped and type(iwe) is ebos.fusion.LazyIwe and not warped.computed and not iwe.computed
        opt.step()                                   # the flow changes in place; nothing has read the results yet
        assert not torch.equal(fl.detach(), flow_then)
        kept.append((warped, iwe, flow_then))
    monkeypatch.setenv("EBOS_FUSE_API", "off")
    for warped, iwe, flow_then in kept:              # late reads, oldest first
        w_ref, _ = wp.warp_event(ev, flow_then, "dense-flow", "first")
        i_ref = ic.create_iwe(w_ref, "bilinear_vote", sigma=0)
        assert torch.equal(warped, w_ref)
        assert rel(iwe.cpu().numpy(), i_ref.cpu().numpy()) < 1e-5
        w_o = O.warp_dense_torch(torch.from_numpy(ev_np).float().double(), flow_then.double().cpu(), "first", normalize_t=True)
        assert np.abs(warped.cpu().numpy()[:, :2] - w_o.numpy()[:, :2]).max() < 1e-3


@pytest.mark.parametrize("spike", [1e2, 4e2, 1e3, 1e4])
def test_backward_fixed_point_unit_under_an_upstream_outlier(ebos, spike):
    """The backward scatter's fixed-point unit follows max |upstream| over the tile's staged window (21 bits per event relative to
    max |dt| x 2 max |upstream|): ONE outlier in a tile -- a hot IWE pixel under the variance, a Sobel edge under the gradient
    magnitude -- coarsens every other event's contribution in that tile.  Measured against the fp64 oracle on an upstream
    image with a single value ``spike`` x the typical one.  The quantisation is round-to-nearest per event, so a pixel's error grows
    like sqrt(events) x half a unit; with a spike of 1e4 that was still 1.9 % of the neighbouring pixels' gradients.  The kernel now
    compares max |staged value| with 512 x the window's mean magnitude and gives such a tile the f64 accumulators (bwd_fx_unit,
    kFxOutlier): below the threshold (1e2, 4e2) >= 12 bits per event remain.  Bars: the whole gradient to 1e-3; the
    pixels of the outlier's own tile that the outlier does NOT reach to 0.2 % of the typical gradient magnitude there."""
    h, w, n = 96, 128, 60_000
    rs = np.random.RandomState(23)
    ev = O.synth_events(n, h, w, seed=91)
    fl = O.synth_dense_flow(h, w, seed=92, max_val=3.0)
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32))
    up_np = rs.uniform(0.5, 1.5, (h, w)) * rs.choice([-1.0, 1.0], (h, w))
    up_np[48, 48] = spike                                   # inside tile (1, 1) = rows 32..63, columns 32..63
    fg = G(fl, torch.float32).requires_grad_(True)
    plan.iwe_dense(fg, halo=32).backward(gradient=G(up_np, torch.float32))
    f64 = torch.from_numpy(fl).requires_grad_(True)
    O.iwe_dense(torch.from_numpy(ev), f64, (h, w)).backward(gradient=torch.from_numpy(up_np))
    got, want = fg.grad.cpu().numpy(), f64.grad.numpy()
    assert rel(got, want) < 1e-3
    # the outlier's tile, away from the source pixels whose events can reach the outlier (|flow| <= 3 px, + the two taps)
    tile = np.zeros((h, w), bool)
    tile[32:64, 32:64] = True
    tile[42:55, 42:55] = False
    err = np.abs(got - want)[:, tile]
    typical = np.abs(want[:, tile]).mean()
    print(f"spike {spike:g}: gradient rel-L2 {rel(got, want):.2e}; in the outlier's tile, away from it: max error {err.max():.3e}, "
          f"mean {err.mean():.3e} against a typical |gradient| of {typical:.3e} ({err.max() / typical:.2e} / {err.mean() / typical:.2e})")
    assert err.max() < 2e-3 * typical


@pytest.mark.parametrize("shape,omit,sigma", [((37, 53), False, 3.0), ((37, 53), True, 1.0), ((96, 130), False, 1.0), ((16, 64), True, 3.0),
                                               ((2, 2), False, 3.0), ((3, 5), True, 0.7), ((260, 346), False, 3.0)])
def test_blur3_variance_adjoint_pass_vs_oracle(ebos, shape, omit, sigma):
    """``ebos_blur3_variance_adjoint_f32`` (iwe.blur_sigma > 0 in the solver loop, src/event_image_converter.py:399-404): the
    partials give mean and variance of the oracle's blurred image (fp64) and z is the oracle's B^T (m . B x) -- autograd of
    0.5 sum (m . blur(x))^2 -- on sizes that are no multiple of the 16 x 64 tile, with and without the boundary ring, down to 2 x 2;
    the position weight the backward kernel folds the mean in with is B^T m, i.e. autograd of sum(m . blur(x))."""
    from event_based_bos_amd import _hip
    from event_based_bos_amd.solver.fused_loop import blur_taps

    lib = _hip.require_gpu()
    h, w = shape
    img = torch.from_numpy(np.random.RandomState(7).gamma(2.0, 3.0, (h, w))).float().cuda()
    k0, k1 = blur_taps(sigma)
    n = int(lib.ebos_blur3_variance_partials(h, w))
    z = torch.empty_like(img)
    partials = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    _hip.check(lib.ebos_blur3_variance_adjoint_f32(img.data_ptr(), h, w, int(omit), k0, k1, z.data_ptr(), partials.data_ptr(), n,
                                                   _hip.stream_ptr()), "blur3")
    xo = img.double().cpu().requires_grad_(True)
    y = O.gaussian_blur3_torch(xo, sigma)
    ym = y[1:-1, 1:-1] if omit else y
    if ym.numel() == 0:
        assert partials.abs().sum().item() == 0.0
        return
    s, ss = partials.sum(0).cpu().numpy()
    assert abs(s - ym.sum().item()) <= 2e-6 * abs(ym.sum().item())
    assert abs(ss - (ym ** 2).sum().item()) <= 2e-6 * (ym ** 2).sum().item()
    (0.5 * (ym ** 2).sum()).backward()
    assert rel(z.cpu().numpy(), xo.grad.numpy()) < 2e-6
    with pytest.raises(RuntimeError):
        _hip.check(lib.ebos_blur3_variance_adjoint_f32(img.data_ptr(), h, w, int(omit), k0, k1, z.data_ptr(), partials.data_ptr(), n - 1,
                                                       _hip.stream_ptr()), "blur3")
    with pytest.raises(RuntimeError):  # torch refuses to reflect-pad an axis of one sample
        _hip.check(lib.ebos_blur3_variance_adjoint_f32(img.data_ptr(), 1, h * w, int(omit), k0, k1, z.data_ptr(), partials.data_ptr(), n,
                                                       _hip.stream_ptr()), "blur3")


@pytest.mark.parametrize("model", ["patch", "2dof", "dense"])
def test_windows_with_empty_tiles_vs_oracle(ebos, model):
    """A static background: every event in the top-left quarter of the sensor, three quarters of the tiles EMPTY (the short cuts of
    DESIGN 4.5 #91: a slab of zeros under the smallest window in the forward kernel; no upstream window, no mean in the backward
    kernel -- whose epilogue still owes the regularisers' adjoint on the empty tiles' pixels).  Loss and gradient of one step of the
    four-launch patch-flow loop (both regularisers on) and of the 2-DoF loop, and value + flow gradient of the dense call, against the
    fp64 oracle's autograd."""
    from event_based_bos_amd.solver.fused_loop import Fused2dofLoop, FusedPatchLoop

    h, w, n = 192, 256, 60_000
    ev = O.synth_events(n, h // 2, w // 2, seed=31)            # rows < 96, columns < 128 of a 192 x 256 sensor
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile=(32, 32), emit="compact")
    tiles = plan.key_offsets[::32 * 32].diff()
    assert int((tiles == 0).sum()) >= 30 and int((tiles > 0).sum()) >= 12
    if model == "patch":
        patch = (24, 32)
        gh, gw = ebos.solver.patch_grid_shape((h, w), patch, patch)
        th0 = np.random.RandomState(32).uniform(-3.0, 3.0, (2, gh, gw))
        loop = FusedPatchLoop(plan, patch, patch, G(th0, torch.float32), 1.0, 0.002, 0.01, False, 0, "auto", lr=0.1, capacity=4)
        losses = loop.run(1, resident=False)
        to = torch.from_numpy(th0).float().double().requires_grad_(True)
        dense = O.upsample_patch_flow(to, (h, w), patch, patch)
        lo = O.image_variance(O.iwe_dense(torch.from_numpy(ev), dense, (h, w)), False) + 0.002 * O.flow_norm(dense) \
            + 0.01 * O.image_gradient_tv(dense, torch.ones((h, w), dtype=torch.float64))
        lo.backward()
        assert abs(losses[0].item() - lo.item()) <= 1e-5 * abs(lo.item())
        assert rel(loop.d_theta.cpu().numpy(), to.grad.numpy()) < 1e-3
    elif model == "2dof":
        th0 = np.array([1.5, -2.25])
        loop = Fused2dofLoop(plan, torch.tensor(th0, dtype=torch.float32), 1.0, False, 0, "auto", lr=0.05, capacity=4)
        losses = loop.run(1, resident=False)
        to = torch.from_numpy(th0).float().double().requires_grad_(True)
        dense = torch.stack([-to[0] * torch.ones((h, w), dtype=torch.float64), -to[1] * torch.ones((h, w), dtype=torch.float64)])
        lo = O.image_variance(O.iwe_dense(torch.from_numpy(ev), dense, (h, w)), False)
        lo.backward()
        assert abs(losses[0].item() - lo.item()) <= 1e-5 * abs(lo.item())
        assert rel(loop.d_theta.cpu().numpy().reshape(-1), to.grad.numpy()) < 1e-3
    else:
        fl = O.synth_dense_flow(h, w, seed=33, max_val=5.0)
        for halo in ("auto", 32):
            f = G(fl, torch.float32).requires_grad_(True)
            v = plan.contrast_dense(f, halo=halo)
            v.backward()
            fo = torch.from_numpy(fl).float().double().requires_grad_(True)
            vo = -O.image_variance(O.iwe_dense(torch.from_numpy(ev), fo, (h, w)), False)   # (the oracle's cost is -variance)
            vo.backward()
            assert abs(v.item() - vo.item()) <= 1e-5 * abs(vo.item())
            assert rel(f.grad.cpu().numpy(), fo.grad.numpy()) < 1e-3
            assert float(f.grad[:, h // 2 + 8:, :].abs().max()) == 0.0   # no event, no gradient: the empty tiles' rows


@pytest.mark.parametrize("omit,pad,sigma,tv", [(False, 0, 1.0, 0.0), (True, 0, 3.0, 0.01), (False, 3, 3.0, 0.0)])
def test_blurred_patch_loop_first_step_vs_oracle_autograd(ebos, omit, pad, sigma, tv):
    """One iteration of the native patch-flow loop with iwe.blur_sigma > 0 (combine -> blur image pass -> GRID backward with the
    position-weighted mean term): loss and d loss / d theta against the fp64 oracle's autograd through upsample -> warp -> vote ->
    gaussian_blur3 -> var (+ regularisers), with the boundary ring, image padding and both flow regularisers."""
    from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

    h, w, n = 96, 128, 40_000
    patch = (24, 32)
    ev = O.synth_events(n, h, w, seed=21)
    gh, gw = ebos.solver.patch_grid_shape((h, w), patch, patch)
    th0 = np.random.RandomState(22).uniform(-4.0, 4.0, (2, gh, gw))
    plan = ebos.EventPlan.build(G(ev), (h, w), "first", True, tile="auto", emit="compact")
    loop = FusedPatchLoop(plan, patch, patch, G(th0, torch.float32), 1.0, 0.002, tv, omit, pad, "auto", lr=0.1, capacity=4, blur_sigma=sigma)
    losses = loop.run(1, resident=False)
    to = torch.from_numpy(th0).float().double().requires_grad_(True)
    dense = O.upsample_patch_flow(to, (h, w), patch, patch)
    iwe = O.gaussian_blur3_torch(O.iwe_dense(torch.from_numpy(ev), dense, (h, w), (pad, pad)), sigma)
    lo = O.image_variance(iwe, omit) + 0.002 * O.flow_norm(dense)
    if tv:
        lo = lo + tv * O.image_gradient_tv(dense, torch.ones((h, w), dtype=torch.float64))
    lo.backward()
    assert abs(losses[0].item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert rel(loop.d_theta.cpu().numpy(), to.grad.numpy()) < 1e-3
    # ... and the same step as ONE resident launch (round 6: image padding inside the resident kernel too), against the same oracle
    res = FusedPatchLoop(plan, patch, patch, G(th0, torch.float32), 1.0, 0.002, tv, omit, pad, "auto", lr=0.1, capacity=4, blur_sigma=sigma)
    assert res.resident_supported(), ebos.load_library().ebos_last_error()
    l_res = res.run(1, resident=True)
    assert res.last_run_mode == "resident"
    assert abs(l_res[0].item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert rel(res.d_theta.cpu().numpy(), to.grad.numpy()) < 1e-3
