#!/usr/bin/env python3
"""One solver iteration (forward + backward + Adam step, the loop of src/solver/generative_max_likelihood.py:306-341) against the
spatial distribution of the events: uniform (the BASELINE recipe) vs events concentrated in a Gaussian blob, as a schlieren
object in front of a static background produces (the windows of bos_event.py:144-220 are recordings, not uniform noise).
Per distribution: the four-launch pipeline (adaptive work items split crowded tiles) and the ONE-launch resident loop.

    python tools/bench_skew_solver.py [--events 2000000 10000000] [--sigma 400 200 100 50] [--out profiles/x.json]
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
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop  # noqa: E402

H, W = 720, 1280


def window(n, sigma, rs):
    if sigma is None:
        r, c = rs.randint(0, H, n), rs.randint(0, W, n)
    else:  # a Gaussian blob around the image centre; samples outside the sensor are drawn again (clipping them would pile
        # thousands of events onto single border pixels: a hot-pixel benchmark, not a schlieren object)
        r, c = np.empty(0), np.empty(0)
        while len(r) < n:
            rr, cc = np.rint(rs.normal(H / 2, sigma * H / W, n)), np.rint(rs.normal(W / 2, sigma, n))
            ok = (rr >= 0) & (rr < H) & (cc >= 0) & (cc < W)
            r, c = np.concatenate([r, rr[ok]]), np.concatenate([c, cc[ok]])
        r, c = r[:n], c[:n]
    return np.stack([r, c, np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, nargs="*", default=[2_000_000, 10_000_000])
    ap.add_argument("--sigma", type=float, nargs="*", default=[400, 200, 100, 50])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--modes", nargs="*", default=["pipeline", "resident", "resident_forced"])
    ap.add_argument("--no-uniform", action="store_true", help="skip the uniform window (for a kernel trace of one distribution)")
    ap.add_argument("--size", type=int, nargs=2, default=None, help="sensor H W (default 720 1280)")
    ap.add_argument("--patch", type=int, nargs=2, default=[24, 32])
    ap.add_argument("--tile", type=int, nargs=2, default=None, help="source tile of the plan (default: choose_tile -- 45 x 80 at 1280 x 720)")
    a = ap.parse_args()
    global H, W
    if a.size:
        H, W = a.size
    lib = ebos._hip.require_gpu()
    patch = tuple(a.patch)
    gh, gw = ebos.solver.patch_grid_shape((H, W), patch, patch)
    rows = []
    for n in a.events:
        for sigma in ([] if a.no_uniform else [None]) + list(a.sigma):
            rs = np.random.RandomState(0)
            plan = ebos.EventPlan.build(torch.from_numpy(window(n, sigma, rs)).cuda(), (H, W), "first", True, tile=tuple(a.tile) if a.tile else "auto", emit="compact")
            tiles = plan.key_offsets[::plan.tile[0] * plan.tile[1]].diff().float()
            row = {"events": n, "sigma_px": sigma, "tile": list(plan.tile), "fullest_tile_over_average": round(float(tiles.max() / tiles.mean()), 1)}
            last = {}
            for mode, res, env in (("pipeline", False, None), ("resident", True, None), ("resident_forced", True, "0")):
                if mode not in a.modes:
                    continue
                if env is None:
                    os.environ.pop("EBOS_RESIDENT_MAX_IMBALANCE", None)
                else:
                    os.environ["EBOS_RESIDENT_MAX_IMBALANCE"] = env
                sl = FusedPatchLoop(plan, patch, patch, torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, halo="auto", lr=0.1,
                                    capacity=a.iters + 20)
                if res and not sl.resident_supported():
                    row[mode] = {"unsupported": (lib.ebos_last_error() or b"").decode() or
                                 "crowded window: refused by the host-side check (crowded_for_resident), run() takes the four launches"}
                    continue
                sl.run(10, resident=None if res else False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                losses = sl.run(a.iters, resident=None if res else False)
                torch.cuda.synchronize()
                row[mode] = {"us_per_iteration": round((time.perf_counter() - t0) / a.iters * 1e6, 1), "ran_as": sl.last_run_mode,
                             "status": int(sl.resident_status)}
                last[mode] = float(losses[-1])
                del sl
            os.environ.pop("EBOS_RESIDENT_MAX_IMBALANCE", None)
            row["last_loss"] = last
            rows.append(row)
            print(json.dumps(row), flush=True)
            del plan
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"tool": "tools/bench_skew_solver.py", "image": [H, W], "patch": list(patch), "iterations_timed": a.iters,
                       "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
