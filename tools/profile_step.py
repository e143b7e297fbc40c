#!/usr/bin/env python3
"""Profiling target: N iterations of the tile-private objective (slab forward + variance, tiled backward)
on the BASELINE configs[1] window.  Run under rocprofv3 (kernel trace or PMC)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import H, W, synth_window  # noqa: E402

import event_based_bos_amd as ebos  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tile", type=int, nargs=2, default=[45, 80])
    ap.add_argument("--halo", type=int, default=32)
    ap.add_argument("--splits", type=int, default=1)
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ev, flow_np = synth_window(args.events, 0)
    plan = ebos.EventPlan.build(torch.from_numpy(ev).to(dev), (H, W), "first", True, tile=tuple(args.tile))
    flow = torch.from_numpy(flow_np).float().to(dev).requires_grad_(not args.fwd_only)
    for _ in range(args.iters):
        loss = -plan.contrast_dense(flow, "image_variance", halo=args.halo, splits=args.splits)
        if not args.fwd_only:
            loss.backward()
            flow.grad = None
    torch.cuda.synchronize()
    print("contrast", -loss.item())


if __name__ == "__main__":
    main()
