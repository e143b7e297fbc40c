#!/usr/bin/env python3
"""Profiling target: the fused contrast-maximisation loop (solver/fused_loop.py) on one synthetic window.
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/profile_solver.py --events 2000000"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import H, W, synth_window  # noqa: E402

import event_based_bos_amd as ebos  # noqa: E402
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--patch", type=int, nargs=2, default=[24, 32])
ap.add_argument("--dense", action="store_true", help="materialised route (upsample -> dense kernels -> adjoint) instead of grid sampling")
ap.add_argument("--flow-norm", type=float, default=0.001)
ap.add_argument("--image-gradient", type=float, default=0.0)
ap.add_argument("--halo", type=lambda v: v if v == "auto" else int(v), default=32, help="a built halo, or auto (run-time windows per tile)")
ap.add_argument("--tile", type=int, nargs=2, default=None, help="source tile (default: choose_tile for the halo)")
ap.add_argument("--mode", choices=["auto", "resident", "pipeline"], default="auto", help="one resident launch / four launches per iteration")
ap.add_argument("--gm", type=float, default=0.0, help="weight of the gradient_magnitude contrast (replaces the variance: its weight becomes 0)")
ap.add_argument("--blur", type=float, default=0.0, help="iwe.blur_sigma")
ap.add_argument("--size", type=int, nargs=2, default=None, help="image size (default 720 1280)")
ap.add_argument("--flow-max", type=float, default=0.0, help="initial patch flows U(-m, m) (0: zeros)")
ap.add_argument("--lr", type=float, default=0.1, help="Adam's learning rate (0: the flow stays where it starts -- timing builds)")
ap.add_argument("--blob", type=float, default=0.0, help="events concentrated in a Gaussian blob of this sigma (columns, px) in the middle of the frame "
                                                         "(a schlieren object in front of a static background) instead of uniform")
a = ap.parse_args()
if a.size or a.blob > 0:
    import numpy as np
    H, W = a.size or (H, W)
    rs = np.random.RandomState(3)
    if a.blob > 0:
        r = np.clip(np.rint(rs.normal(H / 2, a.blob * H / W, a.events)), 0, H - 1)
        c = np.clip(np.rint(rs.normal(W / 2, a.blob, a.events)), 0, W - 1)
    else:
        r, c = rs.randint(0, H, a.events), rs.randint(0, W, a.events)
    ev = np.stack([r, c, np.sort(rs.uniform(0, 0.5, a.events)), rs.randint(0, 2, a.events)], 1).astype(np.float64)
else:
    ev, _ = synth_window(a.events, 0)
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=tuple(a.tile) if a.tile else ebos.event_plan.choose_tile((H, W), 32 if a.halo == "auto" else a.halo))
gh, gw = ebos.solver.patch_grid_shape((H, W), a.patch, a.patch)
theta0 = torch.zeros((2, gh, gw)) if a.flow_max == 0 else (torch.rand((2, gh, gw), generator=torch.Generator().manual_seed(1)) * 2 - 1) * a.flow_max
res = {"auto": None, "resident": True, "pipeline": False}[a.mode]
loop = FusedPatchLoop(plan, a.patch, a.patch, theta0, 0.0 if a.gm else 1.0, a.flow_norm, a.image_gradient, halo=a.halo, lr=a.lr, capacity=a.iters + 3,
                      sample_grid=False if a.dense else None, w_gradient_magnitude=a.gm, blur_sigma=a.blur)
loop.run(3, resident=res)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = loop.run(a.iters, resident=res)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{H}x{W} mode {loop.last_run_mode} (status {loop.resident_status}), tile {plan.tile} halo {a.halo}, sample_grid {loop.sample_grid}, flow_norm {a.flow_norm}, image_gradient {a.image_gradient}: {a.events} events, {a.iters} iterations: {dt / a.iters * 1e6:.1f} us/iteration; loss {losses[0].item():.5f} -> {losses[-1].item():.5f}")
