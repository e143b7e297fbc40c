#!/usr/bin/env python3
"""Wall time per Adam iteration of the contrast-maximisation solver (patch flow 30x40 -> 1280x720, N events):
the fixed kernel pipeline (solver/fused_loop.py) vs the autograd loop, each as HIP-graph replay and eagerly."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from bench import H, W, synth_window

ap = argparse.ArgumentParser(); ap.add_argument("--events", type=int, default=10_000_000); ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
ev, _ = synth_window(a.events, 0)
evg = torch.from_numpy(ev).cuda()
for fused, graph in ((True, True), (True, False), (False, True), (False, False)):
    cfg = {"motion_model": "dense-flow", "cost_with_weight": {"image_variance": 1.0, "flow_norm": 0.001},
           "patch": {"size": [24, 32], "sliding_window": [24, 32]},
           "optimizer": {"method": "Adam", "n_iter": a.iters, "parameters": {"lr": 0.1}, "graph": graph, "fused": fused}}
    s = ebos.solver.collections["contrast_maximization"]((H, W), (H, W), solver_config=cfg)
    s.estimate(evg)  # first call in the process: code objects, allocator, Adam state
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.estimate(evg)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"fused={s.fused} graph={graph} graphed={s.graphed}: {dt*1e3:.1f} ms for {a.iters} iterations incl. plan build -> {dt/a.iters*1e6:.1f} us/iter; "
          f"loss {s.history[0]:.5f} -> {s.history[-1]:.5f}")
