#!/usr/bin/env python3
"""Phase breakdown of iwe_slab_accumulate_kernel from in-kernel stamps (diagnostic build:
EBOS_EXTRA_FLAGS=-DEBOS_STAMPS python -m event_based_bos_amd.build --force).  s_memrealtime ticks at 100 MHz."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd import _hip
from bench import H, W, synth_window

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=10_000_000)
ap.add_argument("--grid", action="store_true", help="the grid-sampling kernel of BASELINE configs[3] (30x40 patch grid) instead of a dense field")
args = ap.parse_args()
lib = _hip.require_gpu()
raw = ctypes.CDLL(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH))
ev, fl = synth_window(args.events, 0)
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto", emit="compact")
flow = torch.from_numpy(fl).float().cuda()
if args.grid:
    th, tw = plan.tile
    gh, gw = ebos.solver.patch_grid_shape((H, W), (24, 32), (24, 32))
    g = torch.from_numpy(np.random.RandomState(1).uniform(-30, 30, (2, gh, gw))).float().cuda()
    nws = int(lib.ebos_iwe_slab_workspace_bytes(H, W, th, tw, 32, 1, 0, 0))
    ws = torch.zeros(nws, dtype=torch.uint8, device="cuda")
    iwe = torch.empty((H, W), dtype=torch.float32, device="cuda")
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    mom = torch.empty((1, 2), dtype=torch.float64, device="cuda")
    P = lambda t: None if t is None else t.data_ptr()
    for _ in range(5):
        _hip.check(lib.ebos_iwe_patch_slab_f32(*plan._compact_ptrs(), P(plan.key_offsets), plan.n, P(g), gh, gw, 24, 32, 24, 32, H, W, th, tw,
                                               32, 1, 0, 0, P(ws), nws, P(iwe), 1, 0, P(out), P(mom), P(plan.part_table),
                                               torch.cuda.current_stream().cuda_stream), "ebos_iwe_patch_slab")
else:
    for _ in range(5):
        iwe = plan.iwe_dense(flow)
torch.cuda.synchronize()
n = 256
buf = (ctypes.c_ulonglong * (n * 8))()
raw.ebos_debug_read_stamps(buf, n * 8)
st = np.array(buf[:], dtype=np.float64).reshape(n, 8)[:, :5] * 10.0  # ns
t0 = st[:, 0].min()
names = ["clear LDS + barrier", "main loop (own lane-0 wave)", "wait for the other waves (barrier)", "decode + slab store + checksum"]
print(f"tile {plan.tile}, {n} workgroups; kernel span (first start -> last end): {(st[:,4].max()-t0)/1e3:.2f} us")
print(f"workgroup start skew: median {np.median(st[:,0]-t0)/1e3:.2f} us, max {(st[:,0].max()-t0)/1e3:.2f} us")
for k, nm in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print(f"{nm:42s} median {np.median(d)/1e3:6.2f} us   min {d.min()/1e3:6.2f}   max {d.max()/1e3:6.2f}")
d = st[:, 4] - st[:, 0]
print(f"{'workgroup total':42s} median {np.median(d)/1e3:6.2f} us   min {d.min()/1e3:6.2f}   max {d.max()/1e3:6.2f}")
