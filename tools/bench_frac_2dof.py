#!/usr/bin/env python3
"""One iteration of the resident 2-DoF loop on integer-pixel and on fractional (undistorted) source coordinates, with and without the
3-tap blur, at BASELINE configs[0]'s size and at 1280x720 -- what the loop of the reference's configs/hot_plate1.yaml (2-DoF, blur 3) pays
for the fractions.

    python tools/bench_frac_2dof.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import Fused2dofLoop
for (H, W), n in (((260, 346), 100_000), ((720, 1280), 2_000_000)):
    for frac in (False, True):
        rs = np.random.RandomState(0)
        ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
        if frac:
            ev[:, :2] = np.clip(ev[:, :2] + rs.randint(0, 64, (n, 2)) / 64.0, 0, [H - 1, W - 1])
        tile = tuple(int(v) for v in os.environ["TILE"].split()) if os.environ.get("TILE") else "auto"   # (TILE="45 80": timing builds)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=tile)
        for sigma in (0.0, 3.0):
            out = {}
            for res in (True, False):
                sl = Fused2dofLoop(plan, torch.tensor([1.0, -0.5]), 1.0, False, 0, "auto", lr=0.02, capacity=300, blur_sigma=sigma)
                sl.run(10, resident=res); torch.cuda.synchronize(); t0 = time.perf_counter(); sl.run(200, resident=res); torch.cuda.synchronize()
                out[sl.last_run_mode] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
            print(f"{H}x{W} {n} events, {'fractional' if frac else 'integer'} coordinates, blur {sigma}: {out} us per iteration", flush=True)
