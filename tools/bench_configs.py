#!/usr/bin/env python3
"""Per-GPU timings of BASELINE.json configs 2-5 (the non-headline configurations; bench.py measures config 2's
forward objective).  One process = one GPU's shard; with torchrun every rank times its own shard (no collective).

    python tools/bench_configs.py [--configs 2 3 4 5] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos  # noqa: E402
from event_based_bos_amd.sharding import shard_units, world  # noqa: E402

H, W = 720, 1280


def synth(n, seed):
    rs = np.random.RandomState(seed)
    ev = np.stack([rs.randint(0, H, n), rs.randint(0, W, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1)
    return ev.astype(np.float64)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 3, 4, 5])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--world", type=int, default=8, help="GPUs the full job is sharded over (this process times one shard)")
    args = ap.parse_args()
    rank, size = world()
    size = max(size, 1)
    logical_world = size if size > 1 else args.world
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    out = {}

    if 2 in args.configs or 3 in args.configs:
        ev = synth(10_000_000, 0)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True)
        flow = torch.from_numpy(np.random.RandomState(1).uniform(-30, 30, (2, H, W))).float().to(dev).requires_grad_(True)
        for cfg, cost in ((2, "image_variance"), (3, "gradient_magnitude")):
            if cfg not in args.configs:
                continue

            def fwd_bwd():
                flow.grad = None
                (-plan.contrast_dense(flow, cost)).backward()

            with torch.no_grad():
                t_f = timed(lambda: plan.contrast_dense(flow.detach(), cost), args.reps * 4)
            t_fb = timed(fwd_bwd, args.reps * 4)
            out[f"config{cfg}"] = {"workload": f"10M events, 1280x720 dense flow, {cost}, 1 GPU", "fwd_ms": t_f * 1e3,
                                   "fwd_bwd_ms": t_fb * 1e3, "fwd_Mev_s": 10 / t_f, "fwd_bwd_Mev_s": 10 / t_fb,
                                   "note": "through the Python autograd wrappers (host overhead included)"}
        del plan

    if 4 in args.configs:
        windows = shard_units(64, logical_world, rank if size > 1 else 0)
        plans, grids = [], []
        t0 = time.perf_counter()
        for wi in windows:
            plans.append(ebos.EventPlan.build(torch.from_numpy(synth(2_000_000, wi)).to(dev), (H, W), "first", True))
            grids.append(torch.from_numpy(np.random.RandomState(100 + wi).uniform(-30, 30, (2, 30, 40))).float().to(dev)
                         .requires_grad_(True))
        torch.cuda.synchronize()
        t_plan = time.perf_counter() - t0

        def sweep(backward):
            for p, g in zip(plans, grids):
                dense = ebos.ops.upsample_patch_flow(g, (24, 32), (24, 32), (H, W))
                loss = -p.contrast_dense(dense, "image_variance")
                if backward:
                    g.grad = None
                    loss.backward()

        with torch.no_grad():
            t_f = timed(lambda: sweep(False), args.reps)
        t_fb = timed(lambda: sweep(True), args.reps)
        # the solver's own form of the same work: one Adam iteration of the fixed kernel pipeline per window
        from event_based_bos_amd.solver.fused_loop import FusedPatchLoop
        loops = [FusedPatchLoop(p, (24, 32), (24, 32), g.detach(), 1.0, lr=0.05, capacity=58) for p, g in zip(plans, grids)]
        for lp in loops:
            lp.run(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for lp in loops:
            lp.run(50)
        torch.cuda.synchronize()
        t_iter = (time.perf_counter() - t0) / (50 * len(loops))
        nev = 2_000_000 * len(windows)
        out["config4"] = {"solver_iteration_us_per_window": t_iter * 1e6,
                          "solver_iteration_note": "fwd + bwd + Adam on the patch grid, ebos_cmax_patch_solve_f32 (no autograd)",
                          "workload": f"{len(windows)} of 64 windows x 2M events (shard of 1/{logical_world}), 40x30 patch-flow grid",
                          "windows_on_this_gpu": len(windows), "fwd_ms_per_window": t_f / len(windows) * 1e3,
                          "fwd_bwd_ms_per_window": t_fb / len(windows) * 1e3, "fwd_Mev_s": nev / t_f / 1e6,
                          "fwd_bwd_Mev_s": nev / t_fb / 1e6, "plan_build_s_total": t_plan,
                          "note": "fwd / fwd_bwd through the Python autograd wrappers (host overhead included)"}
        del plans

    if 5 in args.configs:
        ev = synth(50_000_000, 0)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True)
        del ev
        gx, gy = np.arange(-30, 30, 60 / 32), np.arange(-30, 30, 60 / 16)  # 32 x 16 grid (optuna 'uniform' sampler)
        grid = np.stack(np.meshgrid(gx, gy, indexing="ij"), -1).reshape(-1, 2)
        mine = shard_units(512, logical_world, rank if size > 1 else 0, "block")
        th = torch.from_numpy(grid[mine]).float().to(dev)
        res = {}

        def sweep():
            res["v"] = plan.variance_2dof(th, chunk=8)

        t = timed(sweep, max(1, args.reps // 2))
        best = int(torch.argmax(res["v"]).item())
        out["config5"] = {"workload": f"{len(mine)} of 512 2-DoF hypotheses (shard of 1/{logical_world}) x 50M events, 1280x720",
                          "sweep_ms": t * 1e3, "event_warps_per_s": 50e6 * len(mine) / t,
                          "ms_per_hypothesis": t / len(mine) * 1e3, "best_theta_in_shard": grid[mine][best].tolist()}

    print(json.dumps({"rank": rank, "results": out}))


if __name__ == "__main__":
    main()
