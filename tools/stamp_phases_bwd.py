#!/usr/bin/env python3
"""Phase breakdown of the GRID backward kernel (iwe_dense_tiled_bwd_kernel<..., GRID>) and the GRID forward kernel inside the
solver loop (run it with EBOS_RESIDENT=0: the resident kernel has its own tool), from in-kernel stamps of the diagnostic twin of the library:
    EBOS_STAMPS_MORE="" bash tools/build_stamps_lib.sh;  EBOS_HIP_LIBRARY=$PWD/event_based_bos_amd/lib/libebos_stamps.so EBOS_RESIDENT=0 python tools/stamp_phases_bwd.py
(--setup: the twin built with EBOS_STAMPS_MORE=-DEBOS_STAMPS_SETUP; the default twin's -DEBOS_STAMPS_EPI overwrites slots 1 and 2).
wall_clock64 ticks at 100 MHz."""
import argparse, ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd import _hip
from event_based_bos_amd.solver.fused_loop import FusedPatchLoop
from bench import H, W, synth_window

ap = argparse.ArgumentParser()
ap.add_argument("--events", type=int, default=2_000_000)
ap.add_argument("--dense", action="store_true", help="the dense-field backward kernel (plan.variance_and_grad_dense) instead of the solver loop's")
ap.add_argument("--setup", action="store_true", help="library built with -DEBOS_STAMPS_SETUP: the sub-steps of the backward kernel's set-up")
ap.add_argument("--sigma", type=float, default=None, help="events in a Gaussian blob of this sigma [px] (tools/bench_skew_solver.py's window) instead of uniform")
ap.add_argument("--halo", type=lambda v: v if v == "auto" else int(v), default=32)
a = ap.parse_args()
lib = _hip.require_gpu()
raw = ctypes.CDLL(os.environ.get("EBOS_HIP_LIBRARY", _hip.LIB_PATH))
ev, _ = synth_window(a.events, 0)
if a.sigma is not None:
    from bench_skew_solver import window
    ev = window(a.events, a.sigma, np.random.RandomState(0))
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto")
if a.dense:
    flow = torch.from_numpy(synth_window(16, 0)[1]).float().cuda()
    for _ in range(5):
        plan.variance_and_grad_dense(flow, halo=a.halo)
    torch.cuda.synchronize()
    n = 256
    buf = (ctypes.c_ulonglong * (n * 8))()
    raw.ebos_debug_read_stamps_bwd(buf, n * 8)
    st = np.array(buf[:], dtype=np.float64).reshape(n, 8)[:, [0, 1, 2, 3, 4, 5]] * 10.0  # ns
    t0 = st[:, 0].min()
    print(f"dense backward, {a.events} events: kernel span {(st[:, 5].max() - t0) / 1e3:.2f} us; start skew max {(st[:, 0].max() - t0) / 1e3:.2f} us")
    for i, nm in enumerate(["loads issued, moments reduced, window known", "clear + upstream tile -> LDS + barrier", "main loop (lane-0 wave)",
                            "wait for other waves", "d_flow tile store"]):
        d = st[:, i + 1] - st[:, i]
        print(f"  {nm:40s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}")
    sys.exit(0)
gh, gw = ebos.solver.patch_grid_shape((H, W), (24, 32), (24, 32))
loop = FusedPatchLoop(plan, (24, 32), (24, 32), torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, lr=0.1, capacity=64, halo=a.halo)
loop.run(40)
torch.cuda.synchronize()
n = 1024
if a.setup:
    buf = (ctypes.c_ulonglong * (n * 8))()
    raw.ebos_debug_read_stamps_bwd(buf, n * 8)
    st = np.array(buf[:], dtype=np.float64).reshape(n, 8) * 10.0
    order = [0, 2, 3, 4, 5, 6]
    names = ["loads issued (staging, events, partials)", "interpolation tables + barrier + cell load issued", "window bound posted",
             "partials reduced (two barriers), mean published", "window known, (re)staging issued"]
    print(f"backward set-up, {a.events} events (slot 0 -> 1 of the usual table)")
    for (i0, i1), nm in zip(zip(order[:-1], order[1:]), names):
        d = st[:, i1] - st[:, i0]
        print(f"  {nm:52s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}")
    sys.exit(0)
for name, fn, names in (("backward", raw.ebos_debug_read_stamps_bwd,
                         ["variance partials reduce", "clear + upstream staging + tile flow", "main loop (lane-0 wave)", "wait for other waves",
                          "addend / flow_norm / weights", "tile adjoint + store"]),
                        ("forward", raw.ebos_debug_read_stamps,
                         ["clear + tile flow", "main loop (lane-0 wave)", "wait for other waves", "decode + slab store"])):
    buf = (ctypes.c_ulonglong * (n * 8))()
    fn(buf, n * 8)
    st = np.array(buf[:], dtype=np.float64).reshape(n, 8) * 10.0  # ns
    k = len(names)
    st = st[(st[:, 0] > 0) & (st[:, k] > 0)]   # (work items that ran)
    t0 = st[:, 0].min()
    print(f"{name}: kernel span (first start -> last end) {(st[:, k].max() - t0) / 1e3:.2f} us; start skew median {np.median(st[:, 0] - t0) / 1e3:.2f} max {(st[:, 0].max() - t0) / 1e3:.2f} us")
    for i, nm in enumerate(names):
        d = st[:, i + 1] - st[:, i]
        print(f"  {nm:40s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}")
    d = st[:, k] - st[:, 0]
    print(f"  {'workgroup total':40s} median {np.median(d) / 1e3:6.2f} us   min {d.min() / 1e3:6.2f}   max {d.max() / 1e3:6.2f}   ({len(st)} work items)")
    w = int(np.argmax(d))
    print("  the slowest work item: " + ", ".join(f"{nm.split('(')[0].strip()} {(st[w, i + 1] - st[w, i]) / 1e3:.2f}" for i, nm in enumerate(names)))
