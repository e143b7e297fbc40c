#!/usr/bin/env python3
"""Phase breakdown of ONE iteration of the resident solver kernel (cmax_resident.hip) from in-kernel stamps.
Diagnostic build of that one file, linked beside the product library:
    tools/build_stamps_lib.sh        ->  event_based_bos_amd/lib/libebos_stamps.so
    EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_stamps.so python tools/stamp_resident.py [--events N] [--size H W]
wall_clock64 ticks at 100 MHz (10 ns)."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd import _hip
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=2_000_000)
ap.add_argument("--size", type=int, nargs=2, default=[720, 1280])
ap.add_argument("--patch", type=int, nargs=2, default=[24, 32])
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--flow-max", type=float, default=0.0)
ap.add_argument("--tile", type=int, nargs=2, default=None, help="source tile (the diagnostic twin instruments the 45 x 80 kernels only)")
ap.add_argument("--blur", type=float, default=0.0, help="iwe.blur_sigma of the objective (0: none)")
a = ap.parse_args()
lib = _hip.require_gpu()
raw = ctypes.CDLL(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH))
H, W = a.size
rs = np.random.RandomState(3)
ev = np.stack([rs.randint(0, H, a.events), rs.randint(0, W, a.events), np.sort(rs.uniform(0, 0.5, a.events)), rs.randint(0, 2, a.events)], 1).astype(np.float64)
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile=tuple(a.tile) if a.tile else "auto", emit="compact")
gh, gw = ebos.solver.patch_grid_shape((H, W), a.patch, a.patch)
theta0 = torch.zeros((2, gh, gw)) if a.flow_max == 0 else (torch.rand((2, gh, gw), generator=torch.Generator().manual_seed(1)) * 2 - 1) * a.flow_max
loop = FusedPatchLoop(plan, a.patch, a.patch, theta0, 1.0, 0.001, 0.0, lr=0.02 if a.blur else 0.1, capacity=a.iters + 8, halo="auto", blur_sigma=a.blur)
loop.run(a.iters, resident=True)
torch.cuda.synchronize()
assert loop.last_run_mode == "resident", loop.resident_status
th, tw = plan.tile
n = -(-H // th) * -(-W // tw)
buf = (ctypes.c_ulonglong * (n * 32))()
raw.ebos_debug_read_stamps_resident(buf, n * 32)
st = np.array(buf[:], dtype=np.float64).reshape(n, 32) * 10.0  # ns
phases = [("F0 cells -> window bound, tile flow", 0, 1), ("F1 event loop + decode (+ own sum) + slab stores issued", 1, 2),
          ("   drain slab stores + barrier + record {sum, window}", 2, 3),
          ("S1 own part of the gather; poll ALL records (the all-to-all)", 3, 4), ("   reduce -> mean", 4, 5),
          ("G  gather the upstream window + affine map -> LDS", 5, 6), ("   wave sums, clear, tile flow (apron), prefetch, barrier", 6, 7),
          ("   (publish path, if any) + fixed-point unit", 7, 9), ("B1 sweep", 9, 10),
          ("B2 regulariser + tile adjoint (stores issued)", 10, 11), ("   barrier", 11, 12),
          ("S3 + A: poll the partial values, Adam (+ LDS clear, loss bookkeeping)", 12, 13), ("   (end of the iteration)", 13, 14)]
print(f"{H}x{W}, {a.events} events, tile {plan.tile}, {n} workgroups, blur_sigma {a.blur}; last of {a.iters} iterations")
tot = st[:, 14] - st[:, 0]
for nm, i0, i1 in phases:
    d = st[:, i1] - st[:, i0]
    print(f"  {nm:66s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}")
print(f"  {'iteration (F0 -> A)':66s} median {np.median(tot) / 1e3:6.2f} us   min {tot.min() / 1e3:6.2f}   max {tot.max() / 1e3:6.2f}")
print(f"  start skew of the iteration across workgroups: {(st[:, 0].max() - st[:, 0].min()) / 1e3:.2f} us;  arrival skew at S1: "
      f"{(st[:, 3].max() - st[:, 3].min()) / 1e3:.2f} us")
bb = (ctypes.c_ulonglong * (n * 8))()
raw.ebos_debug_read_stamps_resident_bwd(bb, n * 8)
sb = np.array(bb[:], dtype=np.float64).reshape(n, 8) * 10.0
for nm, d in (("B1: main sweep (lane-0 wave)", sb[:, 3] - st[:, 9]), ("B1: wait for the other waves", st[:, 10] - sb[:, 3]),
              ("B2: row / column weights", sb[:, 1] - st[:, 10]), ("B2: pixels: decode + regulariser + planar store", sb[:, 2] - sb[:, 1]),
              ("B2: wave sum of the regulariser value", sb[:, 7] - sb[:, 2]), ("B2: barrier", sb[:, 5] - sb[:, 7]), ("B2: row sums, column sums, stores", sb[:, 6] - sb[:, 5])):
    print(f"  {nm:62s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}")
