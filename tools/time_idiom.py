#!/usr/bin/env python3
"""Host time of each stage of the reference idiom (warp_event -> create_iwe -> cost.calculate -> backward), per iteration.
    python tools/time_idiom.py [--events N] [--iters K] [--profile]"""
import argparse, cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos  # noqa: E402
from bench import H, W, synth_window  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ev_np, fl_np = synth_window(a.events, 0)
ev = torch.from_numpy(ev_np).float().to(dev)
flow = torch.from_numpy(fl_np).float().to(dev).requires_grad_(True)
wp, ic = ebos.Warp((H, W), normalize_t=True), ebos.EventImageConverter((H, W))
cost = ebos.costs.functions["image_variance"]()
acc = [0.0] * 5
pc = time.perf_counter


def it(timed):
    t0 = pc()
    flow.grad = None
    t1 = pc()
    warped, _ = wp.warp_event(ev, flow, "dense-flow", "first")
    t2 = pc()
    iwe = ic.create_iwe(warped, "bilinear_vote", sigma=0)
    t3 = pc()
    loss = cost.calculate({"iwe": iwe, "omit_boundary": False})
    t4 = pc()
    loss.backward()
    t5 = pc()
    if timed:
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += d


for _ in range(50):
    it(False)
torch.cuda.synchronize()
T0 = pc()
for _ in range(a.iters):
    it(True)
host = pc() - T0
torch.cuda.synchronize()
total = pc() - T0
names = ("flow.grad = None", "warp_event", "create_iwe", "cost.calculate", "loss.backward()")
print(f"{a.events} events: {total / a.iters * 1e6:.1f} us per iteration (host enqueue {host / a.iters * 1e6:.1f})")
for n, v in zip(names, acc):
    print(f"  {n:18s} {v / a.iters * 1e6:6.1f} us")
if a.profile:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(500):
        it(False)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
