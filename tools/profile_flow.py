#!/usr/bin/env python3
"""Profiling target (rocprofv3 --pmc / --kernel-trace): the dense forward objective at several flow amplitudes, 10 M events --
how the accumulate loop's LDS behaviour changes when displacements shrink.   python tools/profile_flow.py --flow-max 2"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from bench import H, W, synth_window
import event_based_bos_amd as ebos
ap = argparse.ArgumentParser()
ap.add_argument("--flow-max", type=float, default=30.0)
ap.add_argument("--events", type=int, default=10_000_000)
ap.add_argument("--halo", type=lambda v: v if v == "auto" else int(v), default=32)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--backward", action="store_true")
a = ap.parse_args()
ev, fl = synth_window(a.events, 0, flow_max=a.flow_max)
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=(45, 80), emit="compact")
flow = torch.from_numpy(fl).float().cuda()
for _ in range(a.iters):
    if a.backward:
        v, _ = plan.variance_and_grad_dense(flow, halo=a.halo)
    else:
        v = plan.contrast_dense(flow, halo=a.halo)
torch.cuda.synchronize()
print("flow_max", a.flow_max, "contrast", v.item())
