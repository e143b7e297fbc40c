#!/usr/bin/env python3
"""One solver iteration of the patch-flow loop on integer-pixel and on fractional (undistorted) source coordinates, 2 M and 100 k events
at 1280x720: what ran (grid-sampling / dense route, resident / launches) and its time -- DESIGN 4.4 #59.

    python tools/bench_frac_patch.py
"""
import time, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop
H, W = 720, 1280
for n in (2_000_000, 100_000):
    for frac in (False, True):
        rs = np.random.RandomState(0)
        ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
        if frac:
            ev[:, :2] = np.clip(ev[:, :2] + rs.randint(0, 64, (n, 2)) / 64.0, 0, [H - 1, W - 1])
        plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto")
        patch = (24, 32)
        gh, gw = ebos.solver.patch_grid_shape((H, W), patch, patch)
        sl = FusedPatchLoop(plan, patch, patch, torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, halo="auto", lr=0.05, capacity=300)
        sl.run(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sl.run(200)
        torch.cuda.synchronize()
        print(n, "frac" if frac else "int", "sample_grid", sl.sample_grid, "mode", sl.last_run_mode, round((time.perf_counter() - t0) / 200 * 1e6, 1), "us/iteration", flush=True)
