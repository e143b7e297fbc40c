#!/usr/bin/env python3
"""Time of the reference's own two-call idiom (warp_event -> create_iwe -> cost -> backward) on GPU tensors,
10 M float32 events at 1280x720: lazily fused (default) vs the literal unfused kernels (EBOS_FUSE_API=off)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from bench import H, W, synth_window

ev_np, fl_np = synth_window(10_000_000, 0)
ev = torch.from_numpy(ev_np).float().cuda()
wp, ic = ebos.Warp((H, W), normalize_t=True), ebos.EventImageConverter((H, W))
cost = ebos.costs.functions["image_variance"]()
for mode in ("off", "f32"):
    os.environ["EBOS_FUSE_API"] = mode
    fl = torch.from_numpy(fl_np).float().cuda().requires_grad_(True)
    def it():
        fl.grad = None
        warped, _ = wp.warp_event(ev, fl, "dense-flow", "first")
        loss = cost.calculate({"iwe": ic.create_iwe(warped, "bilinear_vote", sigma=0), "omit_boundary": False})
        loss.backward()
        return loss
    for _ in range(3): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): l = it()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"EBOS_FUSE_API={mode}: {dt*1e3:.3f} ms per fwd+bwd iteration = {10/dt/1e3:.1f} Gev/s  (loss {l.item():.6f})")
