#!/usr/bin/env python3
"""Soak of the resident solver kernel: many launches on several geometries, every one must complete (status 0) and repeat its
losses BIT FOR BIT (the hand-offs are races if anything is wrong with them: a stale value shows up as a different trajectory).
    python tools/soak_resident.py [--launches 200] [--iters 60]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--iters", type=int, default=60)
a = ap.parse_args()
cases = [((720, 1280), 400_000, (24, 32)), ((260, 346), 100_000, (20, 20)), ((720, 640), 300_000, (24, 32)), ((96, 128), 20_000, (24, 32))]
t0 = time.time()
for (H, W), n, patch in cases:
    rs = np.random.RandomState(7)
    ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
    plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto", emit="compact")
    gh, gw = ebos.solver.patch_grid_shape((H, W), patch, patch)
    theta0 = torch.from_numpy(rs.uniform(-2, 2, (2, gh, gw))).float()
    ref = None
    streams = [torch.cuda.Stream() for _ in range(3)]
    for k in range(a.launches):
        with torch.cuda.stream(streams[k % 3]):
            loop = FusedPatchLoop(plan, patch, patch, theta0, 1.0, 0.001, 0.01, halo="auto", lr=0.1, capacity=a.iters)
            losses = loop.run(a.iters, resident=True)
            assert loop.last_run_mode == "resident" and loop.resident_status == 0, (k, loop.resident_status)
            got = (losses.cpu().numpy().copy(), loop.theta.cpu().numpy().copy())
        if ref is None:
            ref = got
        else:
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), f"{H}x{W}: launch {k} differs from launch 0"
    print(f"{H}x{W} tile {plan.tile}: {a.launches} launches x {a.iters} iterations, all status 0, all bit-identical", flush=True)
print(f"soak done in {time.time() - t0:.1f} s")
