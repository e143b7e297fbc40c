#!/usr/bin/env python3
"""Soak of the lean plan build (ebos_plan_lean) against the full build: random sensors, tiles, event counts (up to --max-events) and
distributions -- uniform, hot pixels (one or many, up to 60 000 events each), blobs (wide and narrow: bins far beyond the bin sort's
staging area), equal timestamps, events outside the image, raw columns with 32- and 64-bit ticks.  Per trial: two lean builds hold
identical arrays (runs larger than the whole staging area excepted: same events, order unspecified), every pixel's run is in ascending
dt, key_offsets and the events are those of the full build.  The committed fuzz (tests/test_ingest.py) is this loop with 24 trials.

    python tools/soak_plan_lean.py [--trials 200] [--seed 1] [--max-events 6000000]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-events", type=int, default=6_000_000)
    a = ap.parse_args()
    rs = np.random.RandomState(a.seed)
    sizes = [3, 900, 40_000, 300_000, 1_200_000, 3_000_000, a.max_events]
    bad = 0
    for trial in range(a.trials):
        H, W = int(rs.randint(33, 800)), int(rs.randint(33, 1300))
        n = int(rs.choice(sizes))
        kind = int(rs.randint(0, 6))
        r, c = rs.randint(0, H, n), rs.randint(0, W, n)
        giant = 0
        if kind == 1 and n > 10:      # hot pixels
            for _ in range(int(rs.randint(1, 5))):
                k = int(min(n // 4, rs.choice([700, 5000, 15000, 30000, 60000])))
                i0 = int(rs.randint(0, n - k + 1))
                r[i0:i0 + k], c[i0:i0 + k] = rs.randint(0, H), rs.randint(0, W)
                giant = max(giant, k)
        elif kind in (2, 3):          # a blob (3: a narrow one)
            s = 12 if kind == 2 else 40
            r = np.clip(np.rint(rs.normal(H / 2, H / s, n)), 0, H - 1).astype(np.int64)
            c = np.clip(np.rint(rs.normal(W / 2, W / s, n)), 0, W - 1).astype(np.int64)
        elif kind == 4:               # events outside the image
            r[: n // 10] = -3
            c[n // 2:: 17] = W + 5
        t = np.sort(rs.randint(0, 500_000 if kind != 5 else 50, n))   # (5: long runs of equal timestamps)
        pol = rs.randint(0, 2, n)
        tile = [(32, 32), (45, 80), (32, 64), (64, 64), "auto"][int(rs.randint(0, 5))]
        raw_mode = int(rs.randint(0, 3))
        if raw_mode == 0:
            ev = torch.from_numpy(np.stack([r, c, t / 1e6, pol], 1).astype(np.float64)).cuda()
            lean = lambda: ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="compact")
            full = lambda: ebos.EventPlan.build(ev, (H, W), "first", True, tile=tile, emit="full")
        else:
            tt = t.astype(np.int32) if raw_mode == 1 else t.astype(np.int64) + 2 ** 34
            cols = [torch.from_numpy(v).cuda() for v in (c.astype(np.int16), r.astype(np.int16), tt, pol.astype(np.uint8))]
            lean = lambda: ebos.EventPlan.build_raw(*cols, (H, W), "first", True, tile=tile, emit="compact")
            full = lambda: ebos.EventPlan.build_raw(*cols, (H, W), "first", True, tile=tile, emit="full")
        p1, p2, pf = lean(), lean(), full()
        tag = (trial, H, W, n, kind, tile, raw_mode)
        try:
            assert p1.lean and p1.n == pf.n and p1.counts() == pf.counts(), "counts"
            used = 4 * int(p1.grp_offsets[-1])
            th, tw = p1.tile
            ko = p1.key_offsets.cpu().numpy().astype(np.int64)
            assert np.array_equal(ko, pf.key_offsets.cpu().numpy()), "key_offsets"
            grp = p1.grp_offsets.cpu().numpy().astype(np.int64)
            cdt, cdt2, fdt = p1.cdt.cpu().numpy(), p2.cdt.cpu().numpy(), pf.cdt.cpu().numpy()
            cpx, cpx2, fpx = p1.cpix.cpu().numpy(), p2.cpix.cpu().numpy(), pf.cpix.cpu().numpy()
            assert np.array_equal(cpx[:used], cpx2[:used]), "cpix of two builds"
            for t_ in range(len(grp) - 1):
                offs = ko[t_ * th * tw:(t_ + 1) * th * tw + 1] - ko[t_ * th * tw]
                lo, hi = 4 * grp[t_], 4 * grp[t_] + offs[-1]
                if hi == lo:
                    continue
                runs = np.repeat(np.arange(th * tw), np.diff(offs))
                want = fdt[lo:hi][np.lexsort((fdt[lo:hi], runs))]
                assert np.array_equal(cpx[lo:hi], fpx[lo:hi]), "cpix"
                if np.diff(offs).max() > 20_000:   # a run that may exceed the staging area: same events, any order
                    keep = (np.diff(offs) <= 20_000)[runs]
                    assert np.array_equal(cdt[lo:hi][keep], want[keep]) and np.array_equal(cdt2[lo:hi][keep], want[keep]), "cdt"
                    assert np.array_equal(cdt[lo:hi][np.lexsort((cdt[lo:hi], runs))], want), "cdt (giant run)"
                else:
                    assert np.array_equal(cdt[lo:hi], want) and np.array_equal(cdt2[lo:hi], want), "cdt"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", tag, e, flush=True)
        if trial % 20 == 19:
            print(f"{trial + 1} trials, {bad} mismatches", flush=True)
    print(f"done: {a.trials} trials, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
