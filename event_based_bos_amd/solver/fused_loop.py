"""The Adam loop of the patch-flow contrast maximisation as a fixed pipeline of HIP kernels (no autograd graph).

One iteration of the loop in src/solver/generative_max_likelihood.py:306-341 (zero_grad -> objective -> backward ->
optimizer.step) for the objective  ``-w * contrast(IWE(dense(theta))) + w_n * flow_norm(dense) + w_g * image_gradient(dense)``
(contrast = image variance, or the gradient magnitude with two more cost kernels)
is, on the MATERIALISED route, five C-ABI calls / eight kernels on one stream, all on buffers allocated once per window:

    ebos_upsample_patch_flow_f32      theta [2, gh, gw] -> dense [2, H, W]
    ebos_iwe_dense_slab_f32           dense -> IWE + variance partials           (2 kernels; 3 without regularisers)
    ebos_flow_regularisers_f32        dense -> regulariser value partials + gradient image; variance, (mean, M)
    ebos_iwe_dense_tiled_bwd_f32      -> d loss / d dense  (variance gradient folded in, regulariser gradient added)
    ebos_upsample_patch_flow_bwd_adam_f32  -> d loss / d theta (2 kernels); the second one applies the Adam step of every
                                      grid element where its gradient appears and records loss[it]

Grid-sampling route (``sample_grid``, the default whenever ``ebos_patch_fused_supported``): the event kernels take theta
itself and evaluate the grid -> dense map per source tile in LDS, so the iteration shrinks to

    ebos_iwe_patch_slab_f32           theta -> IWE + variance partials                         (2 kernels)
    ebos_iwe_patch_tiled_bwd_f32      reduces the variance partials, folds the variance gradient in, evaluates the flow regularisers from the
                                      tile's flow, and leaves partial CELL gradients per tile    (1 kernel)
    ebos_patch_grad_combine_adam_f32  -> d loss / d theta, Adam step, loss[it]                  (1 kernel)

(4 launches, 48.5 us at 2 M events against 64.9 us; image_gradient too is evaluated there, on a 2 px apron of the tile's flow:
55 us against 98 us.)

Expressed through autograd the same iteration is ~35 launches (capturable Adam alone is a dozen) and runs at ~235 us
even as a replayed HIP graph; this pipeline is bounded by its event kernels.  Anything outside this objective family
(other costs, blurred IWE, scipy optimisers) takes the general autograd path of ``ContrastMaximization``.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch

from .. import _hip
from .._hip import check, ptr, stream_ptr
from ..event_plan import EventPlan, _slab_ok, _workspace

FLOW_TERMS = ("flow_norm", "image_gradient")
RESIDENT_BLUR = True   # the resident kernel blurs its gathered window in LDS (EBOS_RESIDENT_BLUR=0: the blurred loop stays four + one launches)


def blur_taps(sigma: float) -> Tuple[float, float]:
    """(k0, k1) of the 3-tap blur of the tensor branch of create_iwe: torchvision gaussian_blur(kernel_size=3, sigma),
    taps exp(-x^2 / 2 sigma^2) at x = -1, 0, 1 normalised to sum 1 (src/event_image_converter.py:399-404)."""
    import math

    e = math.exp(-0.5 / (float(sigma) ** 2))
    return e / (1.0 + 2.0 * e), 1.0 / (1.0 + 2.0 * e)


def crowded_for_resident(plan: EventPlan) -> bool:
    """The resident kernels' own refusal rule (RES_IMBALANCED, csrc/cmax_resident_core.h: one workgroup per tile waits for the fullest
    tile -- ~0.5 ns per event of it on top of the uniform iteration) applied on the host where the plan knows its fullest tile: such a window goes to the launches without a refused launch and
    a status read-back first.  ``EBOS_RESIDENT_MAX_IMBALANCE`` (the kernel's override) switches the host check off."""
    fullest = plan.__dict__.get("_fullest_tile")
    if fullest is None or plan.tile is None or "EBOS_RESIDENT_MAX_IMBALANCE" in os.environ:
        return False
    H, W = plan.image_size
    n_tiles = ((H + plan.tile[0] - 1) // plan.tile[0]) * ((W + plan.tile[1] - 1) // plan.tile[1])
    # measured on both routes (profiles/r06t_crowding_rule.json): at 1280 x 720 the cross-over sits at ~68 k events on the fullest tile
    # of a 1 M-event window, ~76 k at 2 M, ~100 k at 5 M; at 346 x 260 (slower launches) at ~85 k whatever the window
    return fullest > (60_000 + 0.008 * plan.n if n_tiles >= 128 else 85_000)


def blur_supported(plan: EventPlan, halo, sliding_window, contrast_terms: Dict[str, float]) -> bool:
    """iwe.blur_sigma > 0 inside the fixed pipeline: the variance contrast (the blur's image pass feeds the backward kernel of either
    route -- grid sampling, or the dense flow field of windows with fractional source coordinates; csrc/blur3.h), images of at
    least 2 x 2 pixels (torch's reflect padding)."""
    return set(contrast_terms) == {"image_variance"} and min(plan.image_size) >= 2


def objective_supported(contrast_terms: Dict[str, float], flow_terms: Dict[str, float], blur_sigma: float, plan: EventPlan,
                        halo, sliding_window=None) -> bool:
    """The objective family of the fixed kernel pipeline: one contrast term + the two flow regularisers; the 3-tap blur of the
    IWE (iwe.blur_sigma) with the variance contrast on the grid-sampling route (``sliding_window`` given)."""
    if not (len(contrast_terms) == 1 and set(contrast_terms) <= {"image_variance", "gradient_magnitude"}
            and set(flow_terms) <= set(FLOW_TERMS) and halo is not None and _slab_ok(plan, halo)):
        return False
    return not blur_sigma or (sliding_window is not None and blur_supported(plan, halo, sliding_window, contrast_terms))


def supported(contrast_terms: Dict[str, float], flow_terms: Dict[str, float], blur_sigma: float, method: str,
              plan: EventPlan, halo, sliding_window=None) -> bool:
    return method == "Adam" and objective_supported(contrast_terms, flow_terms, blur_sigma, plan, halo, sliding_window)


class ResidentStateTorn(RuntimeError):
    """A resident launch ended with two different verdicts among its workgroups (a wait past its cap AND a hand-over): theta and the
    optimiser state are partly written.  The loop object is unusable; callers rebuild it from the state they started with and run
    the four launches (``ContrastMaximization`` and ``WindowPipeline`` both do)."""


class FusedPatchLoop(object):
    def __init__(self, plan: EventPlan, patch_size: Tuple[int, int], sliding_window: Tuple[int, int], theta0: torch.Tensor,
                 w_variance: float, w_flow_norm: float = 0.0, w_image_gradient: float = 0.0, omit_boundary: bool = False,
                 pad: int = 0, halo: int = 32, lr: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8, capacity: int = 1024,
                 splits: Optional[int] = None, w_gradient_magnitude: float = 0.0, theta_mask: Optional[torch.Tensor] = None,
                 sample_grid: Optional[bool] = None, blur_sigma: float = 0.0):
        self.lib = _hip.require_gpu()
        self.plan, self.patch, self.slide = plan, tuple(int(v) for v in patch_size), tuple(int(v) for v in sliding_window)
        self.w_var, self.w_norm, self.w_tv = float(w_variance), float(w_flow_norm), float(w_image_gradient)
        self.w_gm = float(w_gradient_magnitude)
        if (self.w_var != 0.0) == (self.w_gm != 0.0):
            raise ValueError("exactly one contrast weight (variance or gradient magnitude) must be non-zero")
        from ..event_plan import _norm_halo

        self.omit, self.pad, self.halo = bool(omit_boundary), (int(pad), int(pad)), int(_norm_halo(plan, halo))  # ("auto": run-time windows)
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = plan.device
        H, W = plan.image_size
        f32 = dict(dtype=torch.float32, device=dev)
        self.theta = theta0.detach().to(**f32).contiguous().clone()
        _, self.gh, self.gw = self.theta.shape
        # theta_mask [gh, gw] (optional): 0 = patch not estimated (event thresholding, src/solver/patch_eklt.py:118-126);
        # its gradient is zeroed, so the patch keeps its initial flow
        self.theta_mask = None if theta_mask is None else theta_mask.detach().to(**f32).reshape(self.gh, self.gw).contiguous()
        self.d_theta = torch.empty_like(self.theta)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.theta), torch.zeros_like(self.theta)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)  # device mirror of self.t
        self.t = 0  # Adam steps applied so far (host side: the bias corrections of a step are kernel arguments)
        self.has_reg = self.w_norm != 0.0 or self.w_tv != 0.0
        # sample_grid (default: whenever supported): the event kernels evaluate the patch grid -> dense map per tile
        # themselves (ebos_iwe_patch_*): no upsample / adjoint launches, no [2, H, W] flow and gradient fields
        # (a window of fractional source coordinates -- undistorted events -- takes them on the compact layout WITH the fractions per
        # slot: general event loops, ebos_iwe_patch_slab_frac_f32 / _tiled_bwd_frac_f32; EBOS_FRAC_GRID=0: its dense route, A/B)
        frac_ok = plan.frac_compact is not None and os.environ.get("EBOS_FRAC_GRID", "1") != "0"
        can = bool((plan.compact or frac_ok) and self.lib.ebos_patch_fused_supported(plan.tile[0], plan.tile[1], self.halo, self.slide[0],
                                                                                    self.slide[1]))
        if sample_grid and not can:
            raise ValueError(f"sample_grid: tile {plan.tile} / halo {self.halo} / sliding window {self.slide} is outside "
                             "ebos_patch_fused_supported (or the plan is not compact)")
        if sample_grid is None and os.environ.get("EBOS_SAMPLE_GRID", "1") == "0":  # A/B switch for measurements
            can = False
        self.sample_grid = can if sample_grid is None else bool(sample_grid)
        self.native_grid = self.sample_grid  # (kept for callers of the round's first form, where the two differed for fractional plans)
        # iwe.blur_sigma > 0: the contrast of the 3-tap blurred image (ebos_blur3_variance_adjoint_f32 between the combine and the
        # backward pass; the backward kernel folds the variance gradient in as a z + c wgt, csrc/blur3.h)
        self.blur_sigma = float(blur_sigma or 0.0)
        self.blur = blur_taps(self.blur_sigma) if self.blur_sigma > 0 else (0.0, 0.0)
        if self.blur_sigma > 0 and self.w_gm:
            raise ValueError("blur_sigma > 0: the fixed pipeline takes the blurred image with the variance contrast only "
                             "(fused_loop.blur_supported)")
        # with grid sampling the backward kernel evaluates the flow regularisers per tile, from the flow it holds in LDS
        self.fuse_norm = self.sample_grid and (self.w_tv != 0.0 or self.w_norm != 0.0)
        self.has_reg = self.has_reg and not self.fuse_norm  # from here on: "the regulariser LAUNCH is needed"
        self.dense = torch.empty((2, H, W), **f32) if (self.has_reg or not self.sample_grid) else None
        self.d_dense = None if self.sample_grid else torch.empty((2, H, W), **f32)
        self.n_reg = int(self.lib.ebos_flow_regularisers_partials()) if self.has_reg else 0
        self.d_reg = torch.empty((2, H, W), **f32) if self.has_reg else None
        self.iwe = torch.empty((H + 2 * self.pad[0], W + 2 * self.pad[1]), **f32)
        self.blur_image = torch.empty_like(self.iwe) if self.blur_sigma > 0 else None
        self.variance = torch.empty(1, **f32)
        self.moments = torch.empty((1, 2), dtype=torch.float64, device=dev)
        self.upstream = torch.full((1,), -(self.w_gm or self.w_var), **f32)  # loss = -w * contrast
        self.d_iwe = torch.empty_like(self.iwe) if self.w_gm else None
        # (the Sobel pass's value partials: ebos_gradient_magnitude_fused_f32)
        self.cost_scratch = (torch.empty(int(self.lib.ebos_cmax_cost_scratch_bytes(H + 2 * self.pad[0], W + 2 * self.pad[1])),
                                         dtype=torch.uint8, device=dev) if (self.w_gm or self.blur_sigma > 0) else None)
        self.losses = torch.zeros(max(int(capacity), 1), **f32)
        self.splits = plan.resolve_loop_splits(splits)  # 0 = the plan's adaptive work items
        self.scratch_up = (None if self.sample_grid else
                           torch.empty(int(self.lib.ebos_upsample_bwd_scratch_bytes(self.gh, W)) // 4, **f32))
        self.grad_partials = (torch.empty(int(self.lib.ebos_patch_grad_partials_bytes(H, W, plan.tile[0], plan.tile[1],
                                                                                     int(self.splits == 0))) // 4, **f32)
                              if self.sample_grid else None)
        if self.fuse_norm:  # one value partial per work item of the backward kernel
            self.n_reg = self.grad_partials.numel() // 512
        self.reg_partials = torch.zeros(max(self.n_reg, 1), dtype=torch.float64, device=dev)
        self.ws = _workspace(plan, self.pad, self.halo, self.splits)
        self.graphed = False  # kept for callers that report it: this loop is never graph-replayed
        self._mailbox = None          # flags / records / status word of the resident launch (allocated on first use)
        self.resident_status = 0      # status of the last resident launch (0 = completed)
        self.resident_iterations = 0  # iterations it completed
        self.last_run_mode = "pipeline"
        self._resident_refused = False
        import ctypes as C
        off, n_parts, n_px = C.c_size_t(), C.c_int64(), C.c_int64()
        check(self.lib.ebos_iwe_slab_partials(H, W, plan.tile[0], plan.tile[1], self.halo, self.splits, self.pad[0], self.pad[1],
                                              int(self.omit), C.byref(off), C.byref(n_parts), C.byref(n_px)),
              "ebos_iwe_slab_partials")
        self._var_partials = (off.value, n_parts.value, n_px.value)

    def _forward_backward(self, lib, plan, s) -> None:
        """upsample -> IWE + contrast -> regularisers -> d loss / d dense (shared by iteration() and value_and_grad())."""
        H, W = plan.image_size
        gh, gw, (ph, pw), (sh, sw) = self.gh, self.gw, self.patch, self.slide
        grid = self.sample_grid
        if self.dense is not None:  # (with sample_grid only the regulariser pass reads the dense field)
            check(lib.ebos_upsample_patch_flow_f32(ptr(self.theta), gh, gw, ph, pw, sh, sw, H, W, ptr(self.dense), s),
                  "ebos_upsample_patch_flow")
        use_gm = self.w_gm != 0.0
        h, w = self.iwe.shape
        # variance: with a regulariser launch that kernel reduces the combine pass's partials; with grid sampling the backward
        # kernel does; otherwise a finalize launch
        want_var = 0 if use_gm else (2 if (self.has_reg or grid) else 1)
        off, n_parts, n_px = self._var_partials
        if grid:
            check(lib.ebos_iwe_patch_slab_frac_f32(*self._grid_ptrs(), ptr(plan.key_offsets), plan.n, ptr(self.theta), gh, gw, ph, pw,
                                                   sh, sw, H, W, plan.tile[0], plan.tile[1], self.halo, self.splits, self.pad[0],
                                                   self.pad[1], ptr(self.ws), self.ws.numel(), ptr(self.iwe), want_var, int(self.omit),
                                                   ptr(self.variance), ptr(self.moments), ptr(plan.part_table), s), "ebos_iwe_patch_slab")
        else:
            check(lib.ebos_iwe_dense_slab_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), None, *plan._compact_ptrs(),
                                              ptr(plan.key_offsets), plan.n, ptr(self.dense), H, W, plan.tile[0], plan.tile[1],
                                              self.halo, self.splits, self.pad[0], self.pad[1], ptr(self.ws), self.ws.numel(),
                                              ptr(self.iwe), want_var, int(self.omit),
                                              ptr(self.variance), ptr(self.moments), ptr(plan.part_table), s), "ebos_iwe_dense_slab")
        if use_gm:  # contrast = mean squared Sobel gradient; its gradient image is the upstream of the backward kernel
            check(lib.ebos_gradient_magnitude_fused_f32(ptr(self.iwe), h, w, int(self.omit), ptr(self.upstream), ptr(self.variance),
                                                        ptr(self.d_iwe), ptr(self.cost_scratch), self.cost_scratch.numel() // 8, s),
                  "ebos_gradient_magnitude_fused")
        if self.has_reg:  # ... which also reduces the variance partials of the combine pass (no finalize launch)
            check(lib.ebos_flow_regularisers_f32(ptr(self.dense), H, W, self.w_norm, self.w_tv, ptr(self.d_reg),
                                                 ptr(self.reg_partials), None if use_gm else self.ws.data_ptr() + off, n_parts, n_px,
                                                 ptr(self.variance), ptr(self.moments), s), "ebos_flow_regularisers")
        if grid:  # -> partial cell gradients per tile; _grid_gradient() sums them (and applies Adam)
            check(lib.ebos_iwe_patch_tiled_bwd_frac_f32(*self._grid_ptrs(), ptr(plan.key_offsets), plan.n, ptr(self.theta), gh, gw, ph, pw,
                                                        sh, sw, H, W, plan.tile[0], plan.tile[1], self.halo, self.pad[0], self.pad[1],
                                                        ptr(self.d_iwe if use_gm else self.iwe), 0 if use_gm else int(self.omit),
                                                        ptr(self.moments) if (self.has_reg and not use_gm) else None,
                                                        None if use_gm else ptr(self.upstream),
                                                        ptr(self.d_reg), ptr(self.grad_partials), self.grad_partials.numel() * 4,
                                                        ptr(plan.part_table) if self.splits == 0 else None,
                                                        self.w_norm if self.fuse_norm else 0.0, self.w_tv if self.fuse_norm else 0.0,
                                                        ptr(self.reg_partials),
                                                        None if (use_gm or self.has_reg) else self.ws.data_ptr() + off, n_parts, n_px,
                                                        ptr(self.variance), ptr(self.moments), 0.0, 0.0, s), "ebos_iwe_patch_tiled_bwd")
            return
        check(lib.ebos_iwe_dense_tiled_bwd_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), None, *plan._compact_ptrs(),
                                               ptr(plan.key_offsets), plan.n, ptr(self.dense), H, W, plan.tile[0], plan.tile[1],
                                               self.halo, self.pad[0], self.pad[1], ptr(self.d_iwe if use_gm else self.iwe), None,
                                               0 if use_gm else int(self.omit), ptr(self.d_dense), None,
                                               None if use_gm else ptr(self.moments), None if use_gm else ptr(self.upstream),
                                               ptr(self.d_reg), ptr(self.ws), self.ws.numel(),
                                               ptr(plan.part_table) if self.splits == 0 else None, s),
              "ebos_iwe_dense_tiled_bwd")

    def iteration(self) -> None:
        if self.blur_sigma > 0:
            raise NotImplementedError("blur_sigma > 0: run(native=True) -- the blurred contrast is part of the natively enqueued loop")
        lib, plan, s = self.lib, self.plan, stream_ptr()
        H, W = plan.image_size
        gh, gw, (ph, pw), (sh, sw) = self.gh, self.gw, self.patch, self.slide
        self._forward_backward(lib, plan, s)
        self.t += 1
        if self.sample_grid:
            check(lib.ebos_patch_grad_combine_adam_f32(ptr(self.grad_partials), ptr(plan.part_table) if self.splits == 0 else None,
                                                       plan.tile[0], plan.tile[1], gh, gw, ph, pw, sh, sw, H, W, ptr(self.d_theta),
                                                       ptr(self.theta), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.lr, self.betas[0],
                                                       self.betas[1], self.eps, self.t, ptr(self.step), ptr(self.variance),
                                                       -(self.w_gm or self.w_var), ptr(self.reg_partials), self.n_reg, ptr(self.losses),
                                                       self.losses.numel(), ptr(self.theta_mask), s), "ebos_patch_grad_combine_adam")
            return
        check(lib.ebos_upsample_patch_flow_bwd_adam_f32(ptr(self.d_dense), gh, gw, ph, pw, sh, sw, H, W, ptr(self.scratch_up),
                                                        ptr(self.d_theta), ptr(self.theta), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                                        self.lr, self.betas[0], self.betas[1], self.eps, self.t, ptr(self.step),
                                                        ptr(self.variance), -(self.w_gm or self.w_var), ptr(self.reg_partials), self.n_reg,
                                                        ptr(self.losses), self.losses.numel(), ptr(self.theta_mask), s),
              "ebos_upsample_patch_flow_bwd_adam")

    def value_and_grad(self, theta: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(loss [0-d], d loss / d theta [2, gh, gw]) at ``theta`` through the same kernels, without the Adam update --
        for optimisers that live on the host (scipy)."""
        lib, plan, s = self.lib, self.plan, stream_ptr()
        H, W = plan.image_size
        gh, gw, (ph, pw), (sh, sw) = self.gh, self.gw, self.patch, self.slide
        if self.blur_sigma > 0:
            raise NotImplementedError("blur_sigma > 0: value_and_grad is not part of the blurred pipeline (use the autograd objective)")
        with _hip.on_device(plan.device):
            self.theta.copy_(theta.detach().to(self.theta))
            self._forward_backward(lib, plan, s)
            if self.sample_grid:  # plain gradient: theta = NULL, no optimiser step
                check(lib.ebos_patch_grad_combine_adam_f32(ptr(self.grad_partials), ptr(plan.part_table) if self.splits == 0 else None,
                                                           plan.tile[0], plan.tile[1], gh, gw, ph, pw, sh, sw, H, W, ptr(self.d_theta),
                                                           None, None, None, 0.0, 0.0, 0.0, 0.0, 0, None, None, 0.0, None, 0, None, 0,
                                                           None, s), "ebos_patch_grad_combine_adam")
            else:
                check(lib.ebos_upsample_patch_flow_bwd_f32(ptr(self.d_dense), gh, gw, ph, pw, sh, sw, H, W, ptr(self.scratch_up),
                                                           ptr(self.d_theta), s), "ebos_upsample_patch_flow_bwd")
            loss = -(self.w_gm or self.w_var) * self.variance[0]
            if self.n_reg:
                loss = loss + self.reg_partials.sum().to(torch.float32)
        return loss, (self.d_theta.clone() if self.theta_mask is None else self.d_theta * self.theta_mask)

    def _resident_problem(self) -> "_hip.CmaxPatchProblem":
        """``problem()`` for the resident launch: a BUILT halo of 32 becomes "run-time windows of at most 32 px" where the plan knows
        its |dt| bound -- with every window the full one the resident kernel publishes and re-reads the image in every iteration
        and scatters in f64 (45.7 us per iteration at 2 M events: no faster than the four launches; 28.9 with run-time windows).
        Values are the same; the gradient's fixed-point unit follows the staged window, so against the four-launch pipeline at the
        built halo the trajectory agrees to rounding (~1e-6 per iteration), not bit for bit -- with ``halo="auto"`` it is identical."""
        q = self.problem()
        if self.halo >= 0 and self.plan.dt_bound is not None:
            q.halo = int(self.lib.ebos_halo_auto(int(self.halo), float(self.plan.dt_bound)))
        frac = self.plan.frac_compact
        if frac is not None:  # a window of undistorted events: the compact layout with the fractions per slot (FRAC kernels)
            q.grp_offsets, q.cpix, q.cdt, q.cfx, q.cfy = (ptr(t) for t in frac)
        return q

    def problem(self) -> "_hip.CmaxPatchProblem":
        """The loop's buffers as the ``ebos_cmax_patch_problem`` struct of the C ABI."""
        plan = self.plan
        H, W = plan.image_size
        gp, cp, cd = plan._compact_ptrs()
        q = _hip.CmaxPatchProblem()
        q.xs, q.ys, q.dts, q.grp_offsets, q.cpix, q.cdt = ptr(plan.x), ptr(plan.y), ptr(plan.dt), gp, cp, cd
        q.key_offsets, q.n = ptr(plan.key_offsets), plan.n
        q.H, q.W, q.tile_h, q.tile_w, q.halo = H, W, plan.tile[0], plan.tile[1], self.halo
        q.pad_h, q.pad_w, q.omit_boundary = self.pad[0], self.pad[1], int(self.omit)
        q.splits, q.part_table = self.splits, ptr(plan.part_table)
        q.gh, q.gw, (q.patch_h, q.patch_w), (q.slide_h, q.slide_w) = self.gh, self.gw, self.patch, self.slide
        q.w_variance, q.w_flow_norm, q.w_image_gradient = self.w_var, self.w_norm, self.w_tv
        q.w_gradient_magnitude, q.d_iwe, q.cost_scratch = self.w_gm, ptr(self.d_iwe), ptr(self.cost_scratch)
        q.cost_scratch_bytes = self.cost_scratch.numel() if self.cost_scratch is not None else 0
        q.lr, q.beta1, q.beta2, q.eps = self.lr, self.betas[0], self.betas[1], self.eps
        q.theta, q.d_theta, q.exp_avg, q.exp_avg_sq = ptr(self.theta), ptr(self.d_theta), ptr(self.exp_avg), ptr(self.exp_avg_sq)
        q.step, q.dense, q.d_dense, q.d_reg = ptr(self.step), ptr(self.dense), ptr(self.d_dense), ptr(self.d_reg)
        q.steps_done = self.t
        q.iwe, q.variance, q.moments, q.upstream = ptr(self.iwe), ptr(self.variance), ptr(self.moments), ptr(self.upstream)
        q.reg_partials, q.upsample_scratch = ptr(self.reg_partials), ptr(self.scratch_up)
        q.workspace, q.workspace_bytes = ptr(self.ws), self.ws.numel()
        q.losses, q.losses_cap = ptr(self.losses), self.losses.numel()
        q.theta_mask = ptr(self.theta_mask)
        q.grad_partials = ptr(self.grad_partials)
        q.grad_partials_bytes = self.grad_partials.numel() * 4 if self.grad_partials is not None else 0
        q.blur_k0, q.blur_k1, q.blur_image = self.blur[0], self.blur[1], ptr(self.blur_image)
        if self.sample_grid and not plan.compact:  # fractional source coordinates: the compact layout with the fractions per slot
            q.grp_offsets, q.cpix, q.cdt, q.cfx, q.cfy = self._grid_ptrs()
        return q

    def _grid_ptrs(self):
        """(grp_offsets, cpix, cdt, cfx, cfy) of the grid-sampling entries: the compact plan, or -- fractional source coordinates -- the
        compact layout with the fractions per slot."""
        plan = self.plan
        if plan.compact:
            return plan._compact_ptrs() + (None, None)
        return tuple(ptr(t) for t in plan.frac_compact)

    def resident_supported(self) -> bool:
        """Can ``run`` take the ONE-launch resident kernel (ebos_cmax_patch_solve_resident_f32)?  Grid-sampling route, either
        contrast (the blurred image with the variance only), image padding below half a tile, a tile / halo with a resident kernel, few enough tiles to
        be co-resident.  (The resident kernel runs one workgroup per tile whatever the plan's work-item table says: against a
        pipeline that split crowded tiles it agrees to rounding, not bit for bit.)"""
        frac = self.plan.frac_compact is not None   # fractional source coordinates: the four launches run the dense route, the resident
        # launch the compact layout with the fractions per slot (62 -> 31 us per iteration at 2 M events)
        if not (self.sample_grid or frac) or self.splits not in (0, 1):
            return False
        if self.pad != (0, 0) and os.environ.get("EBOS_RESIDENT_PAD", "1") == "0":   # (the library checks that the padding fits the windows)
            return False
        if frac and os.environ.get("EBOS_RESIDENT_FRAC", "1") == "0":
            return False
        if self.w_gm and os.environ.get("EBOS_RESIDENT_GM", "1") == "0":
            return False
        if crowded_for_resident(self.plan):
            return False
        if self.blur_sigma > 0 and not (RESIDENT_BLUR and os.environ.get("EBOS_RESIDENT_BLUR", "1") != "0"):
            return False
        import ctypes

        return bool(self.lib.ebos_cmax_resident_supported(ctypes.byref(self._resident_problem())))

    def enqueue_resident(self, n_iter: int, spin_timeout_s: float = 2.0) -> torch.Tensor:
        """Enqueue ``n_iter`` iterations as one resident launch on the current stream WITHOUT waiting for it; returns the launch's
        status word as a 1-element int32 tensor (a stream-ordered copy).  0 = completed.  Otherwise the low 8 bits say why it ended
        early -- 1 a wait passed its cap, 2 a tap left the largest LDS window (bits 8 and up: the iteration k it happened in), 3
        geometry, 4 a crowded tile -- and theta / exp_avg / exp_avg_sq / step / losses are unchanged, EXCEPT after a spill in
        iteration k >= 1: the launch then handed over the state of its k completed iterations (``ebos_cmax_resident_iterations``).
        ``self.t`` is advanced by ``n_iter`` here regardless: a caller that sees a non-zero word must not continue this loop
        object (solver.WindowPipeline solves such a window again from its start); everybody else: ``run``, which books what
        really happened."""
        import ctypes

        if self._mailbox is None:
            H, W = self.plan.image_size
            nb = int(self.lib.ebos_cmax_resident_mailbox_bytes(H, W, self.plan.tile[0], self.plan.tile[1]))
            self._mailbox = torch.zeros(nb, dtype=torch.uint8, device=self.plan.device)
        check(self.lib.ebos_cmax_patch_solve_resident_f32(ctypes.byref(self._resident_problem()), int(n_iter), ptr(self._mailbox),
                                                          self._mailbox.numel(), float(spin_timeout_s), stream_ptr()),
              "ebos_cmax_patch_solve_resident")
        self.t += int(n_iter)
        self.last_run_mode = "resident"
        return self._mailbox[:4].view(torch.int32).clone()

    def run_resident(self, n_iter: int, spin_timeout_s: float = 2.0) -> int:
        """``n_iter`` iterations as one resident launch; returns its status after synchronising: 0, or a negative code when the
        launch ended early (-101 a wait passed the cap, -102 a tap left the largest LDS window -- or, with the blurred contrast, the
        windows outgrew the blur's LDS region --, -103 geometry, -104 one tile far more crowded than the average one: the
        pipeline's work items split such tiles).  theta and the optimiser state are then UNCHANGED -- except after -102 in iteration
        k >= 1: ``self.resident_iterations`` = k iterations were completed and handed over (state, losses, step counter are those of
        k iterations) -- and the caller runs the four-launch pipeline for the rest (``run`` does)."""
        t, mode = self.t, self.last_run_mode
        self.enqueue_resident(n_iter, spin_timeout_s)
        self.t, self.last_run_mode = t, mode   # (``run`` books the iterations once it has seen the status)
        status = int(self.lib.ebos_cmax_resident_status(ptr(self._mailbox), stream_ptr()))
        # iterations the launch completed: all of them, none -- or, after a spill in iteration k >= 1, the k before it (the state IS
        # that of k iterations then: the launch hands over instead of discarding its work)
        self.resident_iterations = int(self.lib.ebos_cmax_resident_iterations(ptr(self._mailbox), stream_ptr()))
        if self.resident_iterations < 0:
            raise ResidentStateTorn("resident launch ended with two different verdicts among its workgroups (a wait past its cap AND a "
                                    "hand-over): theta and the optimiser state are partly written -- rebuild the loop from a saved state")
        return status

    def run(self, n_iter: int, native: bool = True, resident: Optional[bool] = None) -> torch.Tensor:
        """``n_iter`` more iterations; returns their losses [n_iter] (device).
        ``resident`` (default: whenever ``resident_supported()``; ``EBOS_RESIDENT=0`` turns the default off): the whole loop as
        ONE resident launch; a launch that ends early (status < 0: see ``run_resident``) leaves the state untouched and the
        four-launch pipeline below runs instead.  The resident call synchronises the stream once, to read its status.
        ``native`` (default): one C call enqueues the whole loop (ebos_cmax_patch_solve_f32); otherwise one Python call
        per kernel group.  (A HIP-graph replay of the iteration was measured slower than plain launches on ROCm 7.2 --
        172 vs 111 us at 2 M events -- and cannot carry the step number, which is a kernel argument.)"""
        n_iter = n_total = int(n_iter)
        if self.t + n_iter > self.losses.numel():
            raise ValueError(f"capacity {self.losses.numel()} < {self.t} steps done + {n_iter}")
        t0 = self.t
        if resident is None:
            # (a launch that ended with -104 -- one tile far more crowded than the average: the pipeline splits such tiles -- or with
            # -102 -- displacements beyond the largest LDS window: the pipeline's spill path -- is not tried again on this window)
            resident = native and os.environ.get("EBOS_RESIDENT", "1") != "0" and not self._resident_refused and self.resident_supported()
        elif resident and not self.resident_supported():
            raise ValueError("resident=True: " + (self.lib.ebos_last_error() or b"").decode())
        self.last_run_mode = "pipeline"
        with _hip.on_device(self.plan.device):
            if resident and n_iter > 0:
                self.resident_status = self.run_resident(n_iter)
                # (a wait past its cap -- the grid could not become co-resident, e.g. another process on the device --, flows beyond
                # the windows, a crowded tile: not tried again on this window)
                self._resident_refused = self.resident_status in (-101, -102, -104)
                if self.resident_status == 0:
                    self.t += n_iter
                    self.last_run_mode = "resident"
                    return self.losses[t0:t0 + n_iter]
                if self.resident_status == -102 and self.resident_iterations > 0:
                    # a flow left the LDS windows in iteration k: the first k iterations are done (state, losses, step counter);
                    # the four launches below continue with the rest
                    self.t += self.resident_iterations
                    n_iter -= self.resident_iterations
                    self.last_run_mode = "resident+pipeline"
            if not native and self.blur_sigma > 0:
                raise NotImplementedError("blur_sigma > 0 needs native=True")
            if native:
                import ctypes

                check(self.lib.ebos_cmax_patch_solve_f32(ctypes.byref(self.problem()), n_iter, stream_ptr()),
                      "ebos_cmax_patch_solve")
                self.t += n_iter
            else:
                for _ in range(n_iter):
                    self.iteration()
        return self.losses[t0:t0 + n_total]


class Fused2dofLoop(object):
    """The Adam loop of the 2-DoF motion model ("2d-translation" / "rigid-optical-flow": x' = x + dt theta, src/warp.py:364-383)
    on loss(theta) = -w var([blur3] IWE(theta)), enqueued natively (ebos_cmax_2dof_solve_f32): four launches per iteration (five
    with iwe.blur_sigma > 0), no host synchronisation -- the loop shape of src/solver/generative_max_likelihood.py:306-341 that
    configs/hot_plate1.yaml:47,65,70 selects (Adam, n_iter 600, blur_sigma 3)."""

    def __init__(self, plan: EventPlan, theta0: torch.Tensor, w_variance: float = 1.0, omit_boundary: bool = False, pad: int = 0,
                 halo="auto", lr: float = 0.05, betas=(0.9, 0.999), eps: float = 1e-8, capacity: int = 1024,
                 splits: Optional[int] = None, blur_sigma: float = 0.0):
        from ..event_plan import _norm_halo

        self.lib = _hip.require_gpu()
        if not plan.binned or (not plan.compact and plan.x is None):
            raise ValueError("Fused2dofLoop needs a binned plan with the compact events or the (x, y, dt) arrays")
        self.plan = plan
        self.w_var, self.omit, self.pad = float(w_variance), bool(omit_boundary), (int(pad), int(pad))
        self.halo = int(_norm_halo(plan, halo))
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        dev = plan.device
        H, W = plan.image_size
        f32 = dict(dtype=torch.float32, device=dev)
        self.theta = theta0.detach().to(**f32).reshape(2).contiguous().clone()
        self.d_theta = torch.zeros_like(self.theta)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.theta), torch.zeros_like(self.theta)
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.t = 0
        self.blur_sigma = float(blur_sigma or 0.0)
        self.blur = blur_taps(self.blur_sigma) if self.blur_sigma > 0 else (0.0, 0.0)
        h, w = H + 2 * self.pad[0], W + 2 * self.pad[1]
        if self.blur_sigma > 0 and min(h, w) < 2:
            raise ValueError("blur_sigma > 0 needs an image of at least 2 x 2 pixels (reflect padding)")
        self.iwe = torch.empty((h, w), **f32)
        self.blur_image = torch.empty_like(self.iwe) if self.blur_sigma > 0 else None
        self.cost_scratch = (torch.empty(16 * int(self.lib.ebos_blur3_variance_partials(h, w)), dtype=torch.uint8, device=dev)
                             if self.blur_sigma > 0 else None)
        self.variance = torch.zeros(1, **f32)
        self.moments = torch.zeros((1, 2), dtype=torch.float64, device=dev)
        self.upstream = torch.full((1,), -self.w_var, **f32)  # loss = -w * contrast
        self.losses = torch.zeros(max(int(capacity), 1), **f32)
        self.splits = plan.resolve_loop_splits(splits)
        self.ws = _workspace(plan, self.pad, self.halo, self.splits)
        self.last_run_mode = "pipeline"
        self._mailbox = None
        self.resident_status = 0
        self.resident_iterations = 0
        self._resident_refused = False

    def _resident_problem(self) -> "_hip.Cmax2dofProblem":
        """``problem()`` for the resident launch: a BUILT halo becomes "run-time windows of at most that many pixels" where the plan
        knows its |dt| bound (as ``FusedPatchLoop._resident_problem``)."""
        q = self.problem()
        if self.halo >= 0 and self.plan.dt_bound is not None:
            q.halo = int(self.lib.ebos_halo_auto(int(self.halo), float(self.plan.dt_bound)))
        fr = self.plan.frac_compact
        if not self.plan.compact and fr is not None:  # fractional source coordinates: the compact layout with the fractions
            q.grp_offsets, q.cpix, q.cdt, q.cfx, q.cfy = (ptr(t) for t in fr)
        return q

    def resident_supported(self) -> bool:
        """Can ``run`` take the ONE-launch resident kernel (ebos_cmax_2dof_solve_resident_f32)?  Compact plan -- of integer source
        pixels, or with the fractions of undistorted events (``EventPlan.frac_compact``) --, image padding below half a tile, a tile /
        halo with a resident kernel."""
        import ctypes

        if not (self.plan.compact or self.plan.frac_compact is not None) or self.splits not in (0, 1):
            return False
        if self.pad != (0, 0) and os.environ.get("EBOS_RESIDENT_PAD", "1") == "0":   # (the library checks that the padding fits the windows)
            return False
        if self.blur_sigma > 0 and os.environ.get("EBOS_RESIDENT_BLUR", "1") == "0":
            return False
        if crowded_for_resident(self.plan):
            return False
        return bool(self.lib.ebos_cmax_2dof_resident_supported(ctypes.byref(self._resident_problem())))

    def enqueue_resident(self, n_iter: int, spin_timeout_s: float = 2.0) -> torch.Tensor:
        """Enqueue ``n_iter`` iterations as one resident launch on the current stream WITHOUT waiting for it; returns the launch's
        status word as a 1-element int32 tensor (a stream-ordered copy; 0 = completed -- as ``FusedPatchLoop.enqueue_resident``).
        ``self.t`` is advanced by ``n_iter`` regardless: a caller that sees a non-zero word solves the window again
        (solver.WindowPipeline does); everybody else: ``run``."""
        import ctypes

        if self._mailbox is None:
            H, W = self.plan.image_size
            nb = int(self.lib.ebos_cmax_resident_mailbox_bytes(H, W, self.plan.tile[0], self.plan.tile[1]))
            self._mailbox = torch.zeros(nb, dtype=torch.uint8, device=self.plan.device)
        check(self.lib.ebos_cmax_2dof_solve_resident_f32(ctypes.byref(self._resident_problem()), int(n_iter), ptr(self._mailbox),
                                                         self._mailbox.numel(), float(spin_timeout_s), stream_ptr()),
              "ebos_cmax_2dof_solve_resident")
        self.t += int(n_iter)
        self.last_run_mode = "resident"
        return self._mailbox[:4].view(torch.int32).clone()

    def run_resident(self, n_iter: int, spin_timeout_s: float = 2.0) -> int:
        """``n_iter`` iterations as one resident launch; returns its status after synchronising (0, or -101 ... -104 as
        ``FusedPatchLoop.run_resident``; after -102 ``resident_iterations`` of them are done and handed over)."""
        t, mode = self.t, self.last_run_mode
        self.enqueue_resident(n_iter, spin_timeout_s)
        self.t, self.last_run_mode = t, mode   # (``run`` books the iterations once it has seen the status)
        status = int(self.lib.ebos_cmax_resident_status(ptr(self._mailbox), stream_ptr()))
        self.resident_iterations = int(self.lib.ebos_cmax_resident_iterations(ptr(self._mailbox), stream_ptr()))
        if self.resident_iterations < 0:
            raise ResidentStateTorn("resident launch ended with two different verdicts among its workgroups: theta and the optimiser "
                               "state are partly written -- rebuild the loop from a saved state")
        return status

    def problem(self) -> "_hip.Cmax2dofProblem":
        plan = self.plan
        H, W = plan.image_size
        gp, cp, cd = plan._compact_ptrs()
        q = _hip.Cmax2dofProblem()
        q.xs, q.ys, q.dts = ptr(plan.x), ptr(plan.y), ptr(plan.dt)
        q.grp_offsets, q.cpix, q.cdt, q.key_offsets, q.n = gp, cp, cd, ptr(plan.key_offsets), plan.n
        if plan.frac_compact is not None and os.environ.get("EBOS_FRAC_GRID", "1") != "0":
            # fractional source coordinates: the compact layout with the fractions per slot (the launches and the resident kernel then
            # run the same arithmetic; EBOS_FRAC_GRID=0: the launches on the (x, y, dt) arrays)
            q.grp_offsets, q.cpix, q.cdt, q.cfx, q.cfy = (ptr(t) for t in plan.frac_compact)
        q.H, q.W, q.tile_h, q.tile_w, q.halo = H, W, plan.tile[0], plan.tile[1], self.halo
        q.pad_h, q.pad_w, q.omit_boundary = self.pad[0], self.pad[1], int(self.omit)
        q.splits, q.part_table = self.splits, ptr(plan.part_table)
        q.w_variance = self.w_var
        q.blur_k0, q.blur_k1 = self.blur
        q.lr, q.beta1, q.beta2, q.eps = self.lr, self.betas[0], self.betas[1], self.eps
        q.theta, q.d_theta, q.exp_avg, q.exp_avg_sq, q.step = (ptr(self.theta), ptr(self.d_theta), ptr(self.exp_avg),
                                                               ptr(self.exp_avg_sq), ptr(self.step))
        q.steps_done = self.t
        q.iwe, q.blur_image, q.variance, q.moments, q.upstream = (ptr(self.iwe), ptr(self.blur_image), ptr(self.variance),
                                                                  ptr(self.moments), ptr(self.upstream))
        q.cost_scratch = ptr(self.cost_scratch)
        q.cost_scratch_bytes = self.cost_scratch.numel() if self.cost_scratch is not None else 0
        q.workspace, q.workspace_bytes = ptr(self.ws), self.ws.numel()
        q.losses, q.losses_cap = ptr(self.losses), self.losses.numel()
        return q

    def run(self, n_iter: int, resident: Optional[bool] = None) -> torch.Tensor:
        """``n_iter`` more Adam iterations; returns their losses [n_iter] (device).  ``resident`` (default: whenever
        ``resident_supported()``; ``EBOS_RESIDENT=0`` turns the default off): the whole loop as ONE resident launch (one
        synchronisation, to read its status); a launch that ends early leaves the state untouched -- or, after a hand-over in
        iteration k, that of k iterations -- and the four (five) launches per iteration below run the rest."""
        import ctypes

        n_iter = n_total = int(n_iter)
        if self.t + n_iter > self.losses.numel():
            raise ValueError(f"capacity {self.losses.numel()} < {self.t} steps done + {n_iter}")
        t0 = self.t
        if resident is None:
            resident = os.environ.get("EBOS_RESIDENT", "1") != "0" and not self._resident_refused and self.resident_supported()
        elif resident and not self.resident_supported():
            raise ValueError("resident=True: " + (self.lib.ebos_last_error() or b"").decode())
        self.last_run_mode = "pipeline"
        with _hip.on_device(self.plan.device):
            if resident and n_iter > 0:
                self.resident_status = self.run_resident(n_iter)
                self._resident_refused = self.resident_status in (-101, -102, -104)
                if self.resident_status == 0:
                    self.t += n_iter
                    self.last_run_mode = "resident"
                    return self.losses[t0:t0 + n_iter]
                if self.resident_status == -102 and self.resident_iterations > 0:
                    self.t += self.resident_iterations
                    n_iter -= self.resident_iterations
                    self.last_run_mode = "resident+pipeline"
            check(self.lib.ebos_cmax_2dof_solve_f32(ctypes.byref(self.problem()), n_iter, stream_ptr()), "ebos_cmax_2dof_solve")
        self.t += n_iter
        return self.losses[t0:t0 + n_total]
