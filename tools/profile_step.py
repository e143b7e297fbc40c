#!/usr/bin/env python3
"""Profiling target for rocprofv3 (kernel trace or PMC): the event kernels of the three bench workloads, a few launches each.

  --mode dense     the tile-private objective, forward + backward, on the BASELINE configs[1] window (10 M events, dense flow
                   U(-30, 30)): built halo AND run-time windows (EBOS_HALO_AUTO) -- and the latter also at U(-4, 4)
  --mode grid      BASELINE configs[3]: 16 windows x 2 M events, 30x40 patch grids, one batched launch per pass (+ the single-window
                   forward / backward / Adam kernels of a solver iteration)
  --mode uniform   BASELINE configs[4]: 2-DoF hypotheses over 50 M events (--events-uniform), 8 of them
  --mode all       the three in turn; workloads.json (sizes) is written to --out-dir for tools/make_pmc_json.py"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import H, W, synth_window  # noqa: E402

import event_based_bos_amd as ebos  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="dense", choices=("dense", "general", "grid", "uniform", "all"))
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--events-uniform", type=int, default=50_000_000)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--tile", type=int, nargs=2, default=[45, 80])
    ap.add_argument("--halo", type=int, default=32)
    ap.add_argument("--splits", type=int, default=1)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--out-dir", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    sizes = {}
    if args.mode in ("dense", "all"):
        ev, flow_np = synth_window(args.events, 0)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile))
        for halo, amp in ((args.halo, 1.0), ("auto", 1.0)):
            flow = (torch.from_numpy(flow_np).float().to(dev) * amp).requires_grad_(not args.fwd_only)
            for _ in range(args.iters):
                loss = -plan.contrast_dense(flow, "image_variance", halo=halo, splits=args.splits)
                if not args.fwd_only:
                    loss.backward()
                    flow.grad = None
        # BASELINE configs[2]: the gradient-magnitude objective on the same plan (Sobel forward + adjoint fused into one pass)
        flow = torch.from_numpy(flow_np).float().to(dev).requires_grad_(not args.fwd_only)
        for _ in range(args.iters):
            loss_gm = -plan.contrast_dense(flow, "gradient_magnitude", halo=args.halo, splits=args.splits)
            if not args.fwd_only:
                loss_gm.backward()
                flow.grad = None
        torch.cuda.synchronize()
        print("dense contrast", -loss.item(), "gradient magnitude", -loss_gm.item())
        sizes["dense"] = {"events": plan.n}
        del plan
    if args.mode in ("general", "all"):
        # the general event formats (bench.py --fractional / --weighted): source coordinates on a 1/64 px grid (12 B/event, the warp
        # looks the flow up at the truncated pixel, src/warp.py:334) and per-event weights (f64 LDS accumulation)
        n_g = min(args.events, 10_000_000)
        ev, flow_np = synth_window(n_g, 0)
        flow = torch.from_numpy(flow_np).float().to(dev)
        plan_w = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile))
        wts = torch.from_numpy(np.random.RandomState(777).uniform(0.5, 1.5, n_g).astype(np.float32)).to(dev)
        rs = np.random.RandomState(4242)
        ev[:, 0] += rs.randint(0, 64, n_g) / 64.0
        ev[:, 1] += rs.randint(0, 64, n_g) / 64.0
        plan_f = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile))
        for _ in range(args.iters):
            iwe_f = plan_f.iwe_dense(flow, halo=args.halo)
            iwe_w = plan_w.iwe_dense(flow, weight=wts, halo=args.halo)
        torch.cuda.synchronize()
        print("general formats: IWE sums", float(iwe_f.sum()), float(iwe_w.sum()))
        sizes["general"] = {"events": n_g}
        del plan_f, plan_w, iwe_f, iwe_w
    if args.mode in ("grid", "all"):
        from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

        n, nw = 2_000_000, 16
        gh, gw = ebos.solver.patch_grid_shape((H, W), (24, 32), (24, 32))
        plans, grids = [], []
        for k in range(nw):
            ev, _ = synth_window(n, k, flow=False)
            plans.append(ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile), emit="compact"))
            grids.append(torch.from_numpy(np.random.RandomState(100 + k).uniform(-30, 30, (2, gh, gw))).float().to(dev))
        for halo in (args.halo, "auto"):
            batch = ebos.SlabBatch(plans, grids, patch=((24, 32), (24, 32)), halo=halo, splits=1)
            for _ in range(args.iters):
                batch.run()
            loop = FusedPatchLoop(plans[0], (24, 32), (24, 32), grids[0], 1.0, 0.001, 0.0, halo=halo, lr=0.1, capacity=2 * args.iters + 2)
            loop.run(args.iters, resident=False)   # four launches per iteration
            loop.run(args.iters)                   # the resident kernel: ONE launch of args.iters iterations
            print("patch-grid loop, halo", halo, "second run:", loop.last_run_mode)
        # ... the other contrasts of the resident patch-grid loop: blurred variance (iwe.blur_sigma 1), gradient magnitude
        for kw in ({"blur_sigma": 1.0}, {"w_gradient_magnitude": 1.0}):
            w_var = 0.0 if "w_gradient_magnitude" in kw else 1.0
            lc = FusedPatchLoop(plans[0], (24, 32), (24, 32), torch.zeros_like(grids[0]), w_var, 0.001, 0.0, halo="auto", lr=0.02,
                                capacity=2 * args.iters + 2, **kw)
            lc.run(args.iters)
            print("patch-grid loop", kw, "ran as", lc.last_run_mode)
            del lc
        # ... and the 2-DoF Adam loop (configs/hot_plate1.yaml:47): four launches per iteration, then ONE resident launch
        from event_based_bos_amd.solver.fused_loop import Fused2dofLoop

        l2 = Fused2dofLoop(plans[0], torch.tensor([1.0, -0.5]), 1.0, halo="auto", lr=0.05, capacity=2 * args.iters + 2)
        l2.run(args.iters, resident=False)
        l2.run(args.iters)
        print("2-DoF loop, second run:", l2.last_run_mode)
        del l2
        torch.cuda.synchronize()
        print("grid variances", batch.variances[:2].tolist())
        sizes["grid"] = {"events": n, "windows_per_launch": nw, "resident_iterations_per_launch": args.iters}
        del plans, batch, loop
    if args.mode in ("uniform", "all"):
        ev, _ = synth_window(args.events_uniform, 0, flow=False)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile), emit="compact")
        th = torch.tensor([[-30.0, -30.0], [3.75, -7.5], [11.25, 15.0], [26.25, 22.5], [0.0, 0.0], [-1.5, 2.0], [7.0, -29.0], [-18.0, 4.0]], device=dev)
        for halo in (args.halo, "auto"):
            for _ in range(max(1, args.iters // 3)):
                v = plan.variance_2dof(th, halo=halo, n_streams=1)
        torch.cuda.synchronize()
        print("uniform variances", v[:2].tolist())
        sizes["uniform"] = {"events": plan.n, "windows_per_launch": int(th.shape[0])}  # (the persistent pass: one launch per chunk of hypotheses)
    if args.out_dir:
        os.makedirs(args.out_dir, exist_ok=True)
        json.dump(sizes, open(os.path.join(args.out_dir, "workloads.json"), "w"))


if __name__ == "__main__":
    main()
