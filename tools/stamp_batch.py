#!/usr/bin/env python3
"""Phases of the PERSISTENT batched accumulate pass (ebos_iwe_slab_batch_f32) from in-kernel stamps -- diagnostic build:
EBOS_EXTRA_FLAGS=-DEBOS_STAMPS python -m event_based_bos_amd.build --force.  One launch of --windows windows of --events events
(30x40 patch grids, U(-flow-max, flow-max)); per workgroup: whole kernel / windows = time per window, and the phases of the LAST
window (the stamps are overwritten window by window).  s_memrealtime ticks at 100 MHz."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd import _hip
from bench import H, W, synth_window

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=2_000_000)
ap.add_argument("--windows", type=int, default=16)
ap.add_argument("--flow-max", type=float, default=30.0)
ap.add_argument("--halo", type=lambda v: v if v == "auto" else int(v), default=32)
ap.add_argument("--dense", action="store_true")
a = ap.parse_args()
lib = _hip.require_gpu()
raw = ctypes.CDLL(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH))
plans, flows = [], []
gh, gw = ebos.solver.patch_grid_shape((H, W), (24, 32), (24, 32))
for k in range(a.windows):
    ev, fl = synth_window(a.events, k, flow=a.dense, flow_max=a.flow_max)
    plans.append(ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto", emit="compact"))
    flows.append(torch.from_numpy(fl if a.dense else np.random.RandomState(100 + k).uniform(-a.flow_max, a.flow_max, (2, gh, gw))).float().cuda())
batch = ebos.SlabBatch(plans, flows, patch=None if a.dense else ((24, 32), (24, 32)), halo=a.halo, splits=1)
for _ in range(4):
    batch.run()
torch.cuda.synchronize()
n = 256
buf = (ctypes.c_ulonglong * (n * 8))()
raw.ebos_debug_read_stamps(buf, n * 8)
st = np.array(buf[:], dtype=np.float64).reshape(n, 8) * 10.0  # ns
span = st[:, 6] - st[:, 5]
print(f"{a.windows} windows x {a.events} events, {'dense' if a.dense else 'patch grid'} flow +-{a.flow_max:g}, halo {a.halo}")
print(f"persistent workgroup: kernel {np.median(span)/1e3:.2f} us median (max {span.max()/1e3:.2f}) = {np.median(span)/1e3/a.windows:.2f} us per window")
names = ["set-up (tile range, flow, bound) + barrier", "event loop (own lane-0 wave)", "wait for the other waves (barrier)", "decode + zero + slab store + checksum"]
for k, nm in enumerate(names):
    d = st[:, k + 1] - st[:, k]
    print(f"  last window: {nm:46s} median {np.median(d)/1e3:6.2f} us   min {d.min()/1e3:6.2f}   max {d.max()/1e3:6.2f}")
