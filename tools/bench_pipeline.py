#!/usr/bin/env python3
"""Windows per second through the solver for a recording-like sequence (BASELINE config 4 shape: windows of 2 M
events at 1280x720, patch flow 30x40): per-window ``solver.estimate`` on host-resident float64 windows (the reference
driver's protocol) vs ``WindowPipeline`` (raw-column ingest under compute, n windows at once).

    python tools/bench_pipeline.py [--windows 8] [--events 2000000] [--iters 200]
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


def main():
    global H, W
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--events", type=int, default=2_000_000)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nc", type=int, nargs="+", default=[1, 2, 3, 4], help="windows in flight to try")
    ap.add_argument("--halo", default="auto", help="solver halo: auto (run-time windows, the solver's default) or a built halo (32, 16)")
    ap.add_argument("--repeat", type=int, default=1, help="timed runs per setting (the minimum is reported)")
    ap.add_argument("--size", type=int, nargs=2, default=[720, 1280], help="image size (e.g. 720 640: hot_plate1's ROI, 128 tiles)")
    ap.add_argument("--tile", type=int, nargs=2, default=None, help="source tile of the window plans (default: the one that fills the GPU with one window)")
    ap.add_argument("--blur", type=float, default=0.0, help="iwe.blur_sigma")
    ap.add_argument("--two-dof", action="store_true", help="the 2-DoF Adam loop (configs/hot_plate1.yaml:47,70 of the reference) instead of the patch flow")
    a = ap.parse_args()
    a.halo = a.halo if a.halo == "auto" else int(a.halo)
    H, W = a.size
    n = a.windows * a.events
    rs = np.random.RandomState(0)
    store = ebos.data_loader.RawEventStore({"x": rs.randint(0, W, n).astype(np.int16), "y": rs.randint(0, H, n).astype(np.int16),
                                            "t": np.sort(rs.randint(0, 8300 * a.windows, n)).astype(np.int32) + 10_000_000,
                                            "p": rs.randint(0, 2, n).astype(bool)})
    windows = [(k * a.events, (k + 1) * a.events) for k in range(a.windows)]
    cfg = {"motion_model": "dense-flow", "warp_direction": "first", "cost_with_weight": {"image_variance": 1.0, "flow_norm": 0.001},
           "patch": {"size": [24, 32], "sliding_window": [24, 32]}, "halo": a.halo, "tile": a.tile, "iwe": {"method": "bilinear_vote", "blur_sigma": a.blur},
           "optimizer": {"method": "Adam", "n_iter": a.iters, "parameters": {"lr": 0.1}}}
    if a.two_dof:
        cfg.update(motion_model="2d-translation", parameters=["trans_x", "trans_y"], cost_with_weight={"image_variance": 1.0},
                   optimizer={"method": "Adam", "n_iter": a.iters, "parameters": {"lr": 0.05}})
    solver = ebos.solver.collections["contrast_maximization"]((H, W), (H, W), solver_config=cfg)
    res = {"motion_model": cfg["motion_model"], "blur_sigma": a.blur, "windows": a.windows, "events_per_window": a.events, "iterations": a.iters, "halo": a.halo, "size": [H, W], "tile": list(solver.plan_tile())}
    host_windows = [store.load_event(*wnd) for wnd in windows[:2]]
    solver.estimate(host_windows[0])  # warm the process
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ev in host_windows:
        solver.estimate(ev)
    torch.cuda.synchronize()
    res["per_window_estimate_ms"] = (time.perf_counter() - t0) / len(host_windows) * 1e3
    for nc, resident in [(nc, r) for r in (True, False) for nc in a.nc]:
        pipe = ebos.solver.WindowPipeline(solver, n_concurrent=nc, resident=resident)
        pipe.run(store, windows)  # warm: streams, allocator pools, pinned staging
        torch.cuda.synchronize()
        times = []
        for _ in range(a.repeat):
            t0 = time.perf_counter()
            pipe.run(store, windows)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / a.windows * 1e3)
        tag = f"pipeline_{nc}" + ("_resident" if resident else "_four_launches")
        res[tag + "_ms_per_window"] = min(times)
        res[tag + "_fallbacks"] = len(pipe.resident_fallbacks)
        if a.repeat > 1:
            res[tag + "_all"] = [round(t, 2) for t in times]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
