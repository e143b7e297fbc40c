#!/usr/bin/env python3
"""Profiling target: N plan builds of the BASELINE configs[1] window (run under rocprofv3 --kernel-trace --stats).
    python tools/profile_plan.py [--events N] [--iters K] [--emit compact|full]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import H, W, synth_window  # noqa: E402

import event_based_bos_amd as ebos  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--emit", default="compact")
a = ap.parse_args()
ev = torch.from_numpy(synth_window(a.events, 0, flow=False)[0]).cuda()
for _ in range(a.iters):
    plan = ebos.EventPlan.build(ev, (H, W), "first", True, tile="auto", emit=a.emit)
torch.cuda.synchronize()
print("plan", plan.n, plan.lean)
