"""Inputs for gradient comparisons between an f32 path and the fp64 oracle: keep the events off the kinks of the vote."""
import numpy as np
import torch

from oracle import ebos_oracle as O


def off_the_kinks(ev, flow, direction, amp, margin=5e-4):
    """Drop the events whose f64-warped coordinate lies within ``margin`` px of an integer.  The bilinear vote is
    piecewise linear in the warped coordinate: AT an integer its value is continuous but its gradient jumps (and the
    inside-the-image test switches), so there an f32 warp (~1e-5 px of rounding at +-90 px) and the f64 oracle
    legitimately report the two different one-sided gradients -- one such event in 60000 is 2e-3..1e-2 of the flow
    gradient of these small images.  Everything else about the case (clustering, borders, out-of-image) is untouched;
    ~0.2 % of the events go.  Iterated because dropping the first / last event moves the reference time."""
    if amp == 0.0:
        return ev   # zero flow: every event sits on a kink, the gradient is not compared
    for _ in range(16):
        warped = O.warp_dense_torch(torch.from_numpy(ev), torch.from_numpy(flow), direction, True).numpy().reshape(-1, 4)
        near = (np.abs(warped[:, :2] - np.rint(warped[:, :2])) < margin).any(1)
        # t == t_ref leaves the source coordinate untouched (an integer for the compact kinds): those events stay, their
        # displacement is exactly zero in f32 and in f64 alike
        near &= warped[:, 2] != 0.0
        if not near.any() or len(ev) - int(near.sum()) < 2:
            return ev
        ev = ev[~near]
    return ev


def off_the_kinks_patch(ev, theta, size, patch, slide, margin=5e-4):
    """off_the_kinks for a patch-grid flow (the fp64 dense field of the oracle's upsample)."""
    dense = O.upsample_patch_flow(torch.from_numpy(theta), size, patch, slide).numpy()
    return off_the_kinks(ev, dense, "first", 1.0, margin)
