#!/usr/bin/env python3
"""Phase breakdown of the lean plan build's two large kernels (lean_stage_kernel, lean_bin_sort_kernel) from in-kernel stamps.
Diagnostic twin of the library: tools/build_lean_stamps_lib.sh (-DEBOS_LEAN_STAMPS) -> lib/libebos_lean_stamps.so, selected with
EBOS_HIP_LIBRARY.  wall_clock64 ticks at 100 MHz."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd import _hip

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=10_000_000)
ap.add_argument("--source", default="f64", choices=("f64", "f32", "raw"))
a = ap.parse_args()
H, W, n = 720, 1280, a.events
rs = np.random.RandomState(0)
col = rs.randint(0, W, n).astype(np.int16); row = rs.randint(0, H, n).astype(np.int16)
t = np.sort(rs.randint(10_000_000, 10_500_000, n)).astype(np.int32); pol = rs.randint(0, 2, n).astype(np.uint8)
if a.source == "raw":
    raw = [torch.from_numpy(v).cuda() for v in (col, row, t, pol)]
    build = lambda: ebos.EventPlan.build_raw(*raw, (H, W), "first", True, tile="auto", emit="compact")
else:
    ev = torch.from_numpy(np.stack([row, col, t / 1e6, pol], 1)).cuda()
    if a.source == "f32":
        ev = ev.float()
    build = lambda: ebos.EventPlan.build(ev, (H, W), "first", True, tile="auto", emit="compact")
for _ in range(4):
    plan = build()
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH))
buf = (ctypes.c_ulonglong * (2 * 8192 * 8))()
lib.ebos_debug_read_lean_stamps(buf, 2 * 8192 * 8)
st = np.array(buf[:], dtype=np.float64).reshape(2, 8192, 8) * 10.0  # ns
for which, name, phases in ((0, "lean_stage_kernel", ["clear hist", "load events + bin", "LDS rank atomics", "scan bins + table row", "scatter into LDS", "stream out + partial"]),
                            (1, "lean_bin_sort_kernel", ["clear", "gather 1 (histogram; a bin that fits: + pixels, dt)", "scan + key_offsets", "arrival order -> pixel order (LDS)", "canonical order + write (whole)"])):
    s = st[which]
    live = s[:, 0] > 0
    s = s[live]
    np_ = len(phases)
    t0 = s[:, 0].min()
    print(f"{name}: {len(s)} workgroups, span {(s[:, np_ - 1 + (0 if which == 0 else 0)].max() - t0) / 1e3:.1f} us; start skew median {np.median(s[:,0]-t0)/1e3:.1f} max {(s[:,0].max()-t0)/1e3:.1f} us")
    for k in range(1, np_ if which == 0 else np_):
        d = s[:, k] - s[:, k - 1]
        print(f"  {phases[k] if which == 0 else phases[k]:38s} median {np.median(d)/1e3:6.2f} us  min {d.min()/1e3:6.2f}  max {d.max()/1e3:6.2f}")
    d = s[:, (np_ - 1)] - s[:, 0]
    print(f"  {'workgroup total':38s} median {np.median(d)/1e3:6.2f} us  min {d.min()/1e3:6.2f}  max {d.max()/1e3:6.2f}")
    if which == 1 and s[:, 5].max() > 0:   # inside the canonical-order step (its last chunk): hot-run detection | 16-lane sorts | ranks + pixels | barrier
        for nm, a, b in (("  . (overfull bin: gather 2, the chunk's pixels) to hot-run detection", 3, 5), ("  . pixel stores + hot-run networks", 5, 6), ("  . 16-lane sorts, ranks of long runs", 6, 7), ("  . closing barrier (drains the stores)", 7, 4)):
            d = s[:, b] - s[:, a]
            print(f"  {nm:38s} median {np.median(d)/1e3:6.2f} us  min {d.min()/1e3:6.2f}  max {d.max()/1e3:6.2f}")
