#!/usr/bin/env python3
"""Does the reference's own geometry fill the chip?  One solver iteration (FusedPatchLoop: accumulate, combine, backward, cell
combine + Adam) on
  (a) 2 M events over the full 720x1280 frame,
  (b) the same number of events inside the ROI of configs/hot_plate1.yaml (columns 320:960 -> 720x640, planned on the full
      frame as ContrastMaximization does: the events keep their sensor coordinates),
  (c) 100 k events at 346x260 (BASELINE configs[0]),
with the work items each plan launches and how many of them hold events.   python tools/bench_roi.py [--halo auto|32|16]"""
import argparse, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

ap = argparse.ArgumentParser()
ap.add_argument("--halo", type=lambda v: v if v == "auto" else int(v), default=32)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--tile", type=int, nargs=2, default=None)
a = ap.parse_args()


def events(n, h, w, c0, c1, seed):
    rs = np.random.RandomState(seed)
    return np.stack([rs.randint(0, h, n), rs.randint(c0, c1, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)


def run(name, h, w, n, c0, c1, patch):
    ev = events(n, h, w, c0, c1, 3)
    tile = tuple(a.tile) if a.tile else ebos.event_plan.choose_tile((h, w), 32 if a.halo == "auto" else a.halo)
    plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (h, w), "first", True, tile=tile, emit="compact")
    th, tw = plan.tile
    tiles = -(-h // th) * -(-w // tw)
    ko = plan.key_offsets.cpu().numpy()
    per_tile = ko[th * tw::th * tw][:tiles] - ko[:-1:th * tw][:tiles]
    splits = plan.resolve_splits(None)
    pt = plan.part_table.cpu().numpy()
    items = int(pt[tiles]) if splits == 0 else tiles  # (ebos_plan_parts: entry [tiles] = work items in use)
    gh, gw = ebos.solver.patch_grid_shape((h, w), patch, patch)
    loop = FusedPatchLoop(plan, patch, patch, torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, halo=a.halo, lr=0.1, capacity=a.iters + 8)
    loop.run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(a.iters)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / a.iters * 1e6
    rec = {"case": name, "image": [h, w], "events": n, "tile": list(plan.tile), "halo": a.halo, "tiles": tiles,
           "tiles_with_events": int((per_tile > 0).sum()), "work_items_mode": "adaptive" if splits == 0 else f"{splits} per tile",
           "work_items_launched": tiles * 2 if splits == 0 else tiles * splits, "work_items_in_use": items,
           "grid_sampling": bool(loop.sample_grid), "us_per_iteration": round(us, 1)}
    print(json.dumps(rec))
    return rec


ra = run("(a) full frame", 720, 1280, 2_000_000, 0, 1280, (24, 32))
rb = run("(b) hot_plate1 ROI, columns 320:960", 720, 1280, 2_000_000, 320, 960, (24, 32))
rc = run("(c) 346x260", 260, 346, 100_000, 0, 346, (20, 20))
print(json.dumps({"roi_vs_full": round(rb["us_per_iteration"] / ra["us_per_iteration"], 3)}))
