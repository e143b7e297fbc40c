#!/usr/bin/env python3
"""The four-launch solver loop under clustered events: the plan's adaptive work items (splits 0) against one work item per tile
(splits 1) and what ``EventPlan.resolve_loop_splits`` picks (splits None) -- DESIGN 4.4 #62.

    python tools/bench_skew_splits.py
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop
from bench_skew_solver import window, H, W
patch = (24, 32)
gh, gw = ebos.solver.patch_grid_shape((H, W), patch, patch)
for n in (2_000_000, 10_000_000):
    for sigma in (None, 400, 200, 100):
        rs = np.random.RandomState(0)
        plan = ebos.EventPlan.build(torch.from_numpy(window(n, sigma, rs)).cuda(), (H, W), "first", True, tile="auto", emit="compact")
        tiles = plan.key_offsets[::plan.tile[0] * plan.tile[1]].diff().float()
        row = {"sigma": sigma, "ratio": round(float(tiles.max() / tiles.mean()), 1), "parts_used": plan.__dict__.get("_parts_used")}
        for splits in (None, 0, 1):
            sl = FusedPatchLoop(plan, patch, patch, torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, halo="auto", lr=0.1, capacity=300, splits=splits)
            sl.run(10, resident=False); torch.cuda.synchronize(); t0 = time.perf_counter(); sl.run(200, resident=False); torch.cuda.synchronize()
            row[f"splits={splits}->{sl.splits}"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
        print(json.dumps(row), flush=True)
