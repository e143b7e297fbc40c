#!/usr/bin/env python3
"""Soak of the resident solver kernels: many launches on several geometries and of every kernel variant (patch grid: variance, blurred
variance, gradient magnitude, fractional source coordinates; 2-DoF: plain, blurred on fractional coordinates).  Every launch must
complete (status 0); the variants whose arithmetic is order-free (integer-pixel patch-grid loops: fixed point throughout) must repeat
their losses and flows BIT FOR BIT -- the hand-offs are races if anything is wrong with them: a stale value shows up as a different
trajectory --, the others (f64 sums drawn from a dynamic chunk queue, f64 LDS atomics) to 1e-4.
    python tools/soak_resident.py [--launches 200] [--iters 60] [--variants plain blur gm frac 2dof 2dof_blur_frac]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import Fused2dofLoop, FusedPatchLoop

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--variants", nargs="*", default=["plain", "blur", "gm", "frac", "2dof", "2dof_blur_frac"])
ap.add_argument("--pad", type=int, default=0, help="outer_padding of the image (round 6: inside the resident kernels too)")
a = ap.parse_args()
cases = [((720, 1280), 400_000, (24, 32)), ((260, 346), 100_000, (20, 20)), ((720, 640), 300_000, (24, 32)), ((96, 128), 20_000, (24, 32))]
t0 = time.time()
for variant in a.variants:
    frac = "frac" in variant
    exact = variant in ("plain", "blur", "gm")
    for (H, W), n, patch in cases:
        rs = np.random.RandomState(7)
        ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
        if frac:
            ev[:, :2] = np.clip(ev[:, :2] + rs.randint(0, 64, (n, 2)) / 64.0, 0, [H - 1, W - 1])
        plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto", emit="full" if frac else "compact")
        gh, gw = ebos.solver.patch_grid_shape((H, W), patch, patch)
        theta0 = torch.from_numpy(rs.uniform(-2, 2, (2, gh, gw))).float()

        def make():
            if variant.startswith("2dof"):
                return Fused2dofLoop(plan, torch.tensor([1.5, -2.5]), 1.0, False, a.pad, "auto", lr=0.05, capacity=a.iters,
                                     blur_sigma=3.0 if "blur" in variant else 0.0)
            return FusedPatchLoop(plan, patch, patch, theta0, 0.0 if variant == "gm" else 1.0, 0.001, 0.01, False, a.pad, halo="auto", lr=0.02 if variant != "plain" else 0.1,
                                  capacity=a.iters, w_gradient_magnitude=1.0 if variant == "gm" else 0.0, blur_sigma=1.0 if variant == "blur" else 0.0)
        ref = None
        streams = [torch.cuda.Stream() for _ in range(3)]
        for k in range(a.launches):
            with torch.cuda.stream(streams[k % 3]):
                loop = make()
                losses = loop.run(a.iters, resident=True)
                assert loop.last_run_mode == "resident" and loop.resident_status == 0, (variant, (H, W), k, loop.resident_status, loop.last_run_mode)
                got = (losses.cpu().numpy().copy(), loop.theta.cpu().numpy().copy())
            if ref is None:
                ref = got
            elif exact:
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), f"{variant} {H}x{W}: launch {k} differs from launch 0"
            else:
                np.testing.assert_allclose(got[0], ref[0], rtol=1e-3, err_msg=f"{variant} {H}x{W}: launch {k}")
        print(f"{variant:15s} {H}x{W} tile {plan.tile}: {a.launches} launches x {a.iters} iterations, all status 0, "
              f"{'all bit-identical' if exact else 'losses within 1e-3 of the first launch'}", flush=True)
print(f"soak done in {time.time() - t0:.1f} s")
