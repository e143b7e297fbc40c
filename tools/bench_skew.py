#!/usr/bin/env python3
"""Objective evaluation time (fwd, fwd+bwd) against the spatial distribution of the events: uniform (the BASELINE
recipe) vs events concentrated in a Gaussian blob, as a schlieren object in front of a static background produces.

    python tools/bench_skew.py [--events 10000000] [--sigma 400 200 100 50]
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

H, W = 720, 1280


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--sigma", type=float, nargs="*", default=[400, 200, 100, 50])
    ap.add_argument("--splits", type=int, nargs="*", default=[1])
    a = ap.parse_args()
    n = a.events
    rs = np.random.RandomState(0)
    flow = torch.from_numpy(rs.uniform(-30, 30, (2, H, W))).float().cuda()
    rows = []
    for sigma in [None] + list(a.sigma):
        if sigma is None:
            r, c = rs.randint(0, H, n), rs.randint(0, W, n)
        else:
            r = np.clip(np.rint(rs.normal(H / 2, sigma * H / W, n)), 0, H - 1)
            c = np.clip(np.rint(rs.normal(W / 2, sigma, n)), 0, W - 1)
        ev = np.stack([r, c, np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
        plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto")
        tiles = plan.key_offsets[::plan.tile[0] * plan.tile[1]].diff().float()
        for splits in a.splits:
            f = flow.clone().requires_grad_(True)

            def fwd():
                with torch.no_grad():
                    return plan.contrast_dense(flow, splits=splits)

            def fwd_bwd():
                f.grad = None
                plan.contrast_dense(f, splits=splits).backward()

            rows.append({"sigma_px": sigma, "splits": splits, "max_tile_share": round(float(tiles.max() / n), 4),
                         "fwd_us": round(timed(fwd), 1), "fwd_bwd_us": round(timed(fwd_bwd), 1)})
            print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
