#!/usr/bin/env python3
"""Wall-clock of EventPlan.build / build_raw per window (host + device, synchronised), after warm-up.

    python tools/bench_plan_build.py [--events 10000000] [--height 720 --width 1280]
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


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    a = ap.parse_args()
    H, W, n = a.height, a.width, a.events
    rs = np.random.RandomState(0)
    col = rs.randint(0, W, n).astype(np.int16)
    row = rs.randint(0, H, n).astype(np.int16)
    t = np.sort(rs.randint(10_000_000, 10_500_000, n)).astype(np.int32)
    pol = rs.randint(0, 2, n).astype(np.uint8)
    ev64 = np.stack([row, col, t / 1e6, pol], 1)
    g64 = torch.from_numpy(ev64).cuda()
    g32 = g64.float()
    raw = [torch.from_numpy(v).cuda() for v in (col, row, t, pol)]
    res = {"events": n, "image": [H, W]}
    res["build_f64_ms"] = timed(lambda: ebos.EventPlan.build(g64, (H, W), "first", True, tile="auto"))
    res["build_f32_ms"] = timed(lambda: ebos.EventPlan.build(g32, (H, W), "first", True, tile="auto"))
    res["build_raw_ms"] = timed(lambda: ebos.EventPlan.build_raw(*raw, (H, W), "first", True, tile="auto"))
    # lean build (emit="compact": ebos_plan_lean -- compact events + offsets only)
    res["lean_f64_ms"] = timed(lambda: ebos.EventPlan.build(g64, (H, W), "first", True, tile="auto", emit="compact"))
    res["lean_f32_ms"] = timed(lambda: ebos.EventPlan.build(g32, (H, W), "first", True, tile="auto", emit="compact"))
    res["lean_raw_ms"] = timed(lambda: ebos.EventPlan.build_raw(*raw, (H, W), "first", True, tile="auto", emit="compact"))
    res["lean_raw_deferred_ms"] = timed(lambda: ebos.EventPlan.build_raw(*raw, (H, W), "first", True, tile="auto", emit="compact",
                                                                         deferred=True))
    res["soa_only_raw_ms"] = timed(lambda: ebos.EventPlan.build_raw(*raw, (H, W), "first", True, tile=None))
    pin = [torch.from_numpy(v).pin_memory() for v in (col, row, t, pol)]
    res["h2d_raw_ms"] = timed(lambda: [v.to("cuda", non_blocking=True) for v in pin])
    pin64 = torch.from_numpy(ev64).pin_memory()
    res["h2d_f64_ms"] = timed(lambda: pin64.to("cuda", non_blocking=True))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
