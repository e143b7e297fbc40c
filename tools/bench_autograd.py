#!/usr/bin/env python3
"""Host + device cost of the objective THROUGH THE PYTHON API (config 2: 10 M events, 1280x720, variance), per forward +
backward iteration, next to what the same kernels cost when enqueued by direct C-ABI calls (bench.py: fwd_bwd_ms):

    a  plan.contrast_dense(flow).backward()                       (the fused-objective API)
    b  plan.variance_and_grad_dense(flow)                         (no autograd graph)
    c  warp_event -> create_iwe -> image_variance -> backward     (the reference's idiom, lazily fused)

    python tools/bench_autograd.py [--events N] [--iters K] [--profile]
"""
import argparse
import cProfile
import json
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos  # noqa: E402
from bench import H, W, synth_window  # noqa: E402


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t_host = time.perf_counter() - t0  # host time to ENQUEUE the iterations
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6, t_host / iters * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ev_np, fl_np = synth_window(args.events, 0)
    ev = torch.from_numpy(ev_np).float().to(dev)
    plan = ebos.EventPlan.build(ev, (H, W), "first", True, tile="auto")
    flow = torch.from_numpy(fl_np).float().to(dev).requires_grad_(True)
    wp, ic = ebos.Warp((H, W), normalize_t=True), ebos.EventImageConverter((H, W))
    cost = ebos.costs.functions["image_variance"]()

    def a():
        flow.grad = None
        (-plan.contrast_dense(flow, "image_variance")).backward()

    def b():
        plan.variance_and_grad_dense(flow)

    def c():
        flow.grad = None
        warped, _ = wp.warp_event(ev, flow, "dense-flow", "first")
        cost.calculate({"iwe": ic.create_iwe(warped, "bilinear_vote", sigma=0), "omit_boundary": False}).backward()

    out = {"events": args.events}
    for name, fn in (("contrast_dense_backward", a), ("variance_and_grad_dense", b), ("reference_idiom_fused", c)):
        total, host = timed(fn, args.iters)
        out[name] = {"us_per_iteration": round(total, 1), "host_enqueue_us": round(host, 1)}
    # the autograd engine runs GPU nodes on a worker thread: two thread hand-offs per backward(); a single-threaded engine
    # (torch.autograd.set_multithreading_enabled(False), what a solver loop can set around itself) avoids them
    with torch.autograd.set_multithreading_enabled(False):
        for name, fn in (("contrast_dense_backward", a), ("reference_idiom_fused", c)):
            total, host = timed(fn, args.iters)
            out[name + "_single_threaded_engine"] = {"us_per_iteration": round(total, 1), "host_enqueue_us": round(host, 1)}
    # opt-in EBOS_FUSE_API=lazy: the idiom's warped events are only computed if something reads them (fusion.LazyWarped)
    os.environ["EBOS_FUSE_API"] = "lazy"
    total, host = timed(c, args.iters)
    out["reference_idiom_lazy_warp"] = {"us_per_iteration": round(total, 1), "host_enqueue_us": round(host, 1)}
    os.environ.pop("EBOS_FUSE_API")
    print(json.dumps(out))
    if args.profile:
        for name, fn in (("contrast_dense_backward", a), ("reference_idiom_fused", c)):
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(200):
                fn()
            pr.disable()
            torch.cuda.synchronize()
            print("=====", name)
            pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
