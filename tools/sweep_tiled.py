#!/usr/bin/env python3
"""Kernel-level sweep of the fused forward/backward kernels on one GPU (tile, halo, splits):
interleaved rounds in one process, median + min of HIP-event times (cdna guide rule 24)."""
import argparse
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import H, W, synth_window  # noqa: E402

import event_based_bos_amd as ebos  # noqa: E402
from event_based_bos_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--configs", type=str, default="64,64,32,1;64,64,32,2;32,64,32,1;32,64,32,2;32,64,48,1")
    ap.add_argument("--bwd", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _hip.require_gpu()
    ev, flow_np = synth_window(args.events, 0)
    ev_gpu = torch.from_numpy(ev).to(dev)
    flow = torch.from_numpy(flow_np).float().to(dev)
    iwe = torch.zeros((H, W), dtype=torch.float32, device=dev)
    P = lambda t: t.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    cfgs = [tuple(int(v) for v in c.split(",")) for c in args.configs.split(";")]
    plans = {}
    for th, tw, halo, sp in cfgs:
        if (th, tw) not in plans:
            plans[(th, tw)] = ebos.EventPlan.build(ev_gpu, (H, W), "first", True, tile=(th, tw))
    plain = ebos.EventPlan.build(ev_gpu, (H, W), "first", True, tile=None)
    times = {c: [] for c in cfgs}
    times["global_atomics_unsorted"] = []
    times["global_atomics_sorted"] = []
    ref = None

    def timed(fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iwe.zero_()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)

    for r in range(args.rounds + 1):
        for c in cfgs:
            th, tw, halo, sp = c
            pl = plans[(th, tw)]
            t = timed(lambda: _hip.check(lib.ebos_iwe_dense_tiled_f32(P(pl.x), P(pl.y), P(pl.dt), None, P(pl.key_offsets), pl.n,
                                                                      P(flow), H, W, th, tw, halo, sp, 0, 0, P(iwe), stream), "tiled"))
            if r:
                times[c].append(t)
            if ref is None:
                ref = iwe.clone()
            else:
                err = (torch.linalg.norm(iwe - ref) / torch.linalg.norm(ref)).item()
                assert err < 1e-5, (c, err)
        for name, pl in (("global_atomics_unsorted", plain), ("global_atomics_sorted", plans[cfgs[0][:2]])):
            t = timed(lambda: _hip.check(lib.ebos_iwe_dense_f32(P(pl.x), P(pl.y), P(pl.dt), None, pl.n, P(flow), H, W, W, 0, 0,
                                                               P(iwe), stream), "plain"))
            if r:
                times[name].append(t)
    # tile-private slab pipeline (accumulate + combine [+ variance])
    from event_based_bos_amd.event_plan import _launch_iwe_dense_slab, _launch_dense_bwd, _slab_ok
    slab_times = {}
    for c in cfgs:
        th, tw, halo, sp = c
        pl = plans[(th, tw)]
        if not _slab_ok(pl, halo):
            continue
        for want_var, compact in ((False, True), (True, True), (False, False)):
            saved_pix = pl.cpix
            if not compact:
                pl.cpix = None
            ts = []
            for r in range(args.rounds + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out_iwe, var, _ = _launch_iwe_dense_slab(pl, flow, None, (0, 0), halo, sp, want_var, False)
                b.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(a.elapsed_time(b))
            err = (torch.linalg.norm(out_iwe - ref) / torch.linalg.norm(ref)).item()
            assert err < 1e-5, (c, err)
            pl.cpix = saved_pix
            slab_times[(("slab+var" if want_var else "slab") + ("" if compact else "-xy12B"), c)] = ts
    times.update(slab_times)
    n = args.events
    for k, v in times.items():
        med, mn = statistics.median(v), min(v)
        print(f"{str(k):32s} median {med*1e3:9.1f} us  min {mn*1e3:9.1f} us  {n/med/1e6:9.1f} Gev/s  "
              f"algo {(12*n+12*H*W)/med/1e6:8.1f} GB/s")
    if args.bwd:
        g = torch.randn((H, W), dtype=torch.float32, device=dev)
        d_flow = torch.zeros((2, H, W), dtype=torch.float32, device=dev)
        for c in cfgs:
            th, tw, halo, sp = c
            pl = plans[(th, tw)]
            if sp != 1 or not _slab_ok(pl, halo):
                continue
            ts = []
            for r in range(args.rounds + 1):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                df, _ = _launch_dense_bwd(pl, flow, None, (0, 0), g, None, 0, False, halo)
                b.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(a.elapsed_time(b))
            print(f"{'bwd_tiled ' + str(c):32s} median {statistics.median(ts)*1e3:9.1f} us  min {min(ts)*1e3:9.1f} us")
        for name, pl, srt in (("bwd_sorted", plans[cfgs[0][:2]], 1), ("bwd_unsorted", plain, 0)):
            ts = []
            for r in range(args.rounds + 1):
                d_flow.zero_()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _hip.check(lib.ebos_iwe_dense_bwd_f32(P(pl.x), P(pl.y), P(pl.dt), None, pl.n, P(flow), H, W, W, 0, 0, P(g), None, 0,
                                                      srt, P(d_flow), None, stream), "bwd")
                b.record()
                torch.cuda.synchronize()
                if r:
                    ts.append(a.elapsed_time(b))
            print(f"{name:32s} median {statistics.median(ts)*1e3:9.1f} us  min {min(ts)*1e3:9.1f} us")


if __name__ == "__main__":
    main()
