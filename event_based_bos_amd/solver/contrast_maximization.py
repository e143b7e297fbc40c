"""``ContrastMaximization`` -- the solver the reference's release references but does not ship
(configs/README.md:65 points at a missing src/solver/contrast_maximization.py).

objective(theta) = - contrast(IWE(warp(events, motion(theta)))) [+ weighted regularisers on the flow]

    motion_model "dense-flow"      theta = patch-flow grid [2, gh, gw] (``patch.size`` / ``patch.sliding_window``),
                                   upsampled to a dense flow (src/solver/patch_eklt.py:173-204), optimised with Adam
                                   (loop shape of src/solver/generative_max_likelihood.py:306-341)
    motion_model "2d-translation"  theta = (trans_x, trans_y); exhaustive grid sweep over ``parameters`` min/max
                                   (the optuna "uniform"/grid sampler, :238-255), optionally refined with Adam

Everything per event runs in the fused tile-private HIP pipeline on an ``EventPlan`` built once per window.
YAML keys read (same names as configs/hot_plate1.yaml:46-80 of the reference): warp_direction, motion_model,
parameters, cost, cost_with_weight, outer_padding, iwe.{method, blur_sigma}, patch.{size, sliding_window, pyramid.{coarsest, finest}},
optimizer.{method (Adam | CG | BFGS | L-BFGS-B | TNC | SLSQP | grid | random | TPE | optuna + sampler), sampler, seed, n_iter, parameters.lr,
options, graph, fused,
refine_iters}.
"""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import costs, ops
from .._staging import to_gpu
from ..event_image_converter import EventImageConverter
from ..event_plan import EventPlan
from . import fused_loop
from .base import SolverBase

logger = logging.getLogger(__name__)

CONTRAST_COSTS = ("image_variance", "gradient_magnitude")
SCIPY_METHODS = ("CG", "BFGS", "L-BFGS-B", "TNC", "SLSQP")  # first-order methods of scipy.optimize.minimize


def patch_grid_shape(image_size, patch_size, sliding_window):
    """Number of patch centres per axis (src/solver/patch_eklt.py:85-89)."""
    gh = len(np.arange(0, image_size[0] - patch_size[0] + sliding_window[0], sliding_window[0]))
    gw = len(np.arange(0, image_size[1] - patch_size[1] + sliding_window[1], sliding_window[1]))
    return gh, gw


class ContrastMaximizationMixin(object):
    """The CMax logic, independent of WHICH ``SolverBase`` it is composed over (``make_solver_class``): the build's
    own (``solver/base.py``) or the reference's (``src/solver/base.py:54-378``, whose visualisation / flow-error methods the
    driver calls, bos_event.py:202-219).  Needs from the base's constructor only ``slv_config`` and ``orig_image_shape``
    (both bases set them, reference :72-83); everything else it uses is set up by ``_cmax_setup``."""

    def _cmax_setup(self) -> None:
        cfg = self.slv_config = dict(self.slv_config or {})
        self.orig_image_shape = tuple(int(v) for v in self.orig_image_shape)
        # attributes the build's SolverBase provides and the reference's does not (it calls the padding ``padding``, :74-76)
        if not hasattr(self, "pad"):
            self.pad = int(cfg.get("outer_padding", 0))
        if not hasattr(self, "warp_direction"):
            self.warp_direction = cfg.get("warp_direction", "first")
        if not hasattr(self, "motion_model"):  # (the reference's visualize_one_batch_warp reads self.motion_model, :183)
            self.motion_model = cfg.get("motion_model", "dense-flow")
        if not hasattr(self, "previous_best"):
            self.previous_best = None
        self.cost_with_weight: Dict[str, float] = dict(cfg.get("cost_with_weight") or {cfg.get("cost", "image_variance"): 1.0})
        self.contrast_terms = {k: w for k, w in self.cost_with_weight.items() if k in CONTRAST_COSTS}
        self.flow_terms = {k: w for k, w in self.cost_with_weight.items() if k not in CONTRAST_COSTS}
        if not self.contrast_terms:
            raise ValueError(f"cost_with_weight needs at least one contrast cost of {CONTRAST_COSTS}")
        unknown = [k for k in self.flow_terms if k not in costs.functions]
        if unknown:
            raise KeyError(f"unknown cost(s) {unknown}; registered: {sorted(costs.functions)}")
        self.flow_cost = (costs.HybridCost("minimize", self.flow_terms) if self.flow_terms else None)
        self.omit_boundary = bool(cfg.get("omit_boundary", False))
        iwe_cfg = cfg.get("iwe") or {}
        if iwe_cfg.get("method", "bilinear_vote") != "bilinear_vote":
            raise NotImplementedError("the contrast objective is defined on method='bilinear_vote'")
        # iwe.blur_sigma > 0: the contrast is taken on create_iwe(events, sigma=blur_sigma) of the tensor branch
        # (3-tap blur, src/event_image_converter.py:399-404).  It removes the spurious optimum at zero flow that
        # integer sensor coordinates create (un-warped events hit single pixels, any sub-pixel warp spreads them).
        self.blur_sigma = float(iwe_cfg.get("blur_sigma", 0) or 0)
        pcfg = cfg.get("patch") or {}
        self.patch_size = tuple(pcfg.get("size", (24, 32)))
        self.sliding_window = tuple(pcfg.get("sliding_window", self.patch_size))
        self.pyramid = pcfg.get("pyramid") or None
        # patch.do_event_thresholding / patch.event_thres (src/solver/patch_eklt.py:62-67): a patch is estimated only
        # when more than event_thres events fall inside it (:118-126); the others keep zero flow
        self.do_event_thresholding = bool(pcfg.get("do_event_thresholding", False))
        self.event_thres = int(pcfg.get("event_thres", 0) or 0)
        if self.pyramid and not (int(self.pyramid["coarsest"]) >= int(self.pyramid["finest"]) >= 1):
            raise ValueError("patch.pyramid needs coarsest >= finest >= 1")
        ocfg = cfg.get("optimizer") or {}
        self.opt_method = ocfg.get("method", "Adam")
        self.n_iter = int(ocfg.get("n_iter", 100))
        self.lr = float((ocfg.get("parameters") or {}).get("lr", 0.05))
        self.scipy_options = dict(ocfg.get("options") or {})
        self.refine_iters = int(ocfg.get("refine_iters", 0))  # 2-DoF models: Adam steps after the grid sweep
        # optimizer.sampler (configs/hot_plate1.yaml:69, src/solver/generative_max_likelihood.py:215-226): the outer loop of
        # method "optuna" -- grid / uniform, random, TPE; optimizer.seed (this build's) makes random / TPE draws reproducible
        self.sampler = ocfg.get("sampler", "grid")
        self.sampler_seed = ocfg.get("seed", None)
        self.param_ranges = cfg.get("parameters") or {}  # {name: {min, max}} or the reference's list of names (_param_range)
        # solver.halo: "auto" (default) = run-time LDS windows per tile (event_plan.resolve_halo: each work item sizes its window
        # from a bound on its own displacements; BOS flows are a few pixels), bounded by the largest built halo; an integer
        # = that built halo for every tile.  Either way displacements beyond a window are handled exactly (spill path).
        self.halo = "auto" if cfg.get("halo", "auto") == "auto" else int(cfg.get("halo"))
        # `tile` (optional, [rows, cols]): the source tile of the window plans instead of the one that fills the GPU best with ONE
        # window -- a recording's windows are independent, and larger tiles let more of them run side by side (WindowPipeline)
        self.tile = tuple(int(v) for v in cfg["tile"]) if cfg.get("tile") else None
        # optimizer.graph: capture one whole iteration (upsample -> fused objective -> backward -> Adam update) into a
        # HIP graph and replay it.  Measured on MI355X / ROCm 7.2 (tools/bench_solver.py, 2 M events at 1280x720):
        # 0.195 ms per replayed iteration against 0.68 ms for the eager loop (interpreter + autograd overhead around
        # ~0.1 ms of GPU work); capture + instantiation cost 1-8 ms, i.e. ~10 iterations -- on by default, and any
        # capture failure falls back to the eager loop.
        self.use_graph = bool(ocfg.get("graph", True))
        self.graphed = False
        # optimizer.fused (default on): objectives of the family -w var(IWE) + w_n flow_norm + w_g image_gradient run as
        # a fixed pipeline of HIP kernels (solver/fused_loop.py) instead of through autograd
        self.fused_loop = bool(ocfg.get("fused", True))
        self.fused = False
        # optimizer.resident (default: whenever the geometry allows it): the fused loop as ONE resident launch
        # (ebos_cmax_patch_solve_resident_f32: same trajectory bit for bit, 42.8 -> 31.5 us per iteration at 2 M events); False keeps
        # the four launches per iteration.  ``loop_mode`` = what the last fused loop actually ran ("resident" / "pipeline").
        self.resident = ocfg.get("resident", None)
        self.loop_mode: Optional[str] = None
        self.loop_modes: List[str] = []   # ... of every pyramid scale of the last estimate, coarse to fine
        self.history: List[float] = []

    # ------------------------------------------------------------------ objective pieces
    def _contrast(self, plan: EventPlan, flow: torch.Tensor) -> torch.Tensor:
        total = 0.0
        if self.blur_sigma > 0:
            iwe = plan.iwe_dense(flow, pad=(self.pad, self.pad), halo=self.halo)
            iwe = EventImageConverter._gaussian_blur3(iwe, self.blur_sigma)
            for name, wgt in self.contrast_terms.items():
                fn = ops.image_variance if name == "image_variance" else ops.gradient_magnitude
                total = total + wgt * fn(iwe, self.omit_boundary)
            return total
        for name, wgt in self.contrast_terms.items():
            total = total + wgt * plan.contrast_dense(flow, name, self.omit_boundary, pad=(self.pad, self.pad), halo=self.halo)
        return total

    def objective(self, plan: EventPlan, flow: torch.Tensor) -> torch.Tensor:
        """Loss to MINIMISE: -contrast + regularisers (weights from cost_with_weight)."""
        loss = -self._contrast(plan, flow)
        if self.flow_cost is not None:
            ones = torch.ones(flow.shape[-2:], dtype=flow.dtype, device=flow.device)
            loss = loss + self.flow_cost.calculate({"flow": flow, "omit_boundary": self.omit_boundary, "weights": ones})
        return loss

    # ------------------------------------------------------------------ estimate
    def estimate(self, events, *args, **kwargs) -> np.ndarray:
        """events [n, 4] (x=row, y=col, t, p) -> flow [2, H, W] (numpy), like the reference's solvers."""
        ev = to_gpu(events)
        # (the objective uses unit weights: the lean build -- compact events + offsets only -- is all it reads; windows with
        # fractional, i.e. undistorted, coordinates fall back to the full build inside)
        # (a stream of sub-pixel rectified events is fractional window after window: once a window fell back, the lean
        # attempt -- its kernels and its read-back, ~0.3 ms of a 100 k-event window's build -- is skipped until a full build finds
        # integer coordinates again; either build is valid for either kind of window)
        plan = EventPlan.build(ev, self.orig_image_shape, self.warp_direction, True, tile=self.plan_tile(),
                               emit="full" if getattr(self, "_fractional_stream", False) else "compact")
        self._fractional_stream = plan.frac_compact is not None
        self.history = []
        # The optimisation loops below call loss.backward() thousands of times on graphs of one or two nodes: with the autograd
        # engine's device thread each call pays two thread hand-offs (~35 us of a ~100 us iteration, tools/bench_autograd.py);
        # single-threaded, backward runs on the calling thread.  Restored on exit.
        with torch.autograd.set_multithreading_enabled(False):
            if self.motion_model == "dense-flow":
                flow = self._estimate_patch_flow(plan)
            elif self.motion_model in ("2d-translation", "rigid-optical-flow"):
                theta = self._estimate_translation(plan)
                H, W = self.orig_image_shape
                flow = (-theta).reshape(2, 1, 1).expand(2, H, W)  # dense flow equivalent of theta (src/warp.py:186-187)
            else:
                raise NotImplementedError(f"motion_model {self.motion_model!r}")
        return flow.detach().cpu().numpy().astype(np.float64)

    def _warm_start(self):
        """Warm start handed over by ``set_previous_frame_best_estimation`` -- the build's base stores it as
        ``previous_best``, the reference's as ``previous_frame_best_estimation`` (src/solver/base.py:355-361)."""
        prev = getattr(self, "previous_best", None)
        return prev if prev is not None else getattr(self, "previous_frame_best_estimation", None)

    def plan_tile(self):
        """Source tile of the window plans: the configuration built for this solver's ``halo`` that fills the GPU best
        (``halo: 16`` -- windows whose displacements stay within ~16 px -- selects the smaller LDS windows)."""
        from ..event_plan import choose_tile

        if self.tile is not None:
            return self.tile
        return choose_tile(self.orig_image_shape, 32 if self.halo == "auto" else self.halo)

    def pyramid_scales(self):
        """[(patch_size, sliding_window, n_iter)] coarse to fine.  Without ``patch.pyramid`` one scale (``patch.size``);
        with ``patch.pyramid: {coarsest: c, finest: f}`` the square patches c, c/2, ... of
        src/solver/patch_eklt_pyramid2.py:55-83 (scales 1 .. int(log2(c / f)) + 1, slide = patch) and its
        iteration split ``n_iter // (finest_scale - scale + 1)`` (:260)."""
        if not self.pyramid:
            return [(self.patch_size, self.sliding_window, self.n_iter)]
        c, f = int(self.pyramid["coarsest"]), int(self.pyramid["finest"])
        finest_scale = int(np.log2(c / f)) + 2
        out = []
        for i in range(1, finest_scale):
            size = (c // (2 ** (i - 1)),) * 2
            out.append((size, size, max(1, self.n_iter // (finest_scale - i + 1))))
        return out

    def patch_mask(self, plan: EventPlan, patch_size, sliding_window) -> Optional[torch.Tensor]:
        """[gh, gw] float32 on the device: 1 where the patch holds more than ``event_thres`` events, else 0 -- or None
        when thresholding is off.  The reference crops the whole event array once per patch
        (src/solver/patch_eklt.py:118-126); here the counts are box sums over the plan's per-pixel histogram
        (``EventPlan.patch_event_counts``), without a host read-back."""
        if not self.do_event_thresholding:
            return None
        return (plan.patch_event_counts(patch_size, sliding_window) > self.event_thres).to(torch.float32)

    def _estimate_patch_flow(self, plan: EventPlan) -> torch.Tensor:
        H, W = self.orig_image_shape
        self.history, self.patch_flow_per_scale, self.loop_modes = [], [], []
        theta = None
        for patch_size, sliding_window, n_iter in self.pyramid_scales():
            gh, gw = patch_grid_shape((H, W), patch_size, sliding_window)
            if theta is not None:  # initialise from the coarser scale: resize(x0, patch_image_size), pyramid2.py:253-255
                init = torch.nn.functional.interpolate(theta[None], size=(gh, gw), mode="bilinear", align_corners=False)[0]
            elif self._warm_start() is not None:
                init = to_gpu(self._warm_start(), device=plan.device, dtype=torch.float32).reshape(2, gh, gw)
            else:
                init = torch.zeros((2, gh, gw), dtype=torch.float32, device=plan.device)
            mask = self.patch_mask(plan, patch_size, sliding_window)
            if mask is not None:
                init = init * mask
            theta = self._optimise_patch_grid(plan, init.clone(), patch_size, sliding_window, n_iter, mask)
            self.patch_flow_per_scale.append(theta)
        self.patch_flow = theta
        self.patch_size_used, self.sliding_window_used = patch_size, sliding_window
        return ops.upsample_patch_flow(theta, patch_size, sliding_window, (H, W))

    def _optimise_patch_grid(self, plan: EventPlan, theta: torch.Tensor, patch_size, sliding_window, n_iter: int,
                             mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        H, W = self.orig_image_shape
        theta = theta.requires_grad_(True)

        def evaluate():  # (mask: the gradient of a patch that is not estimated is zero, so it keeps its masked start)
            dense = ops.upsample_patch_flow(theta if mask is None else theta * mask, patch_size, sliding_window, (H, W))
            return self.objective(plan, dense)

        if self.fused_loop and fused_loop.supported(self.contrast_terms, self.flow_terms, self.blur_sigma, self.opt_method,
                                                    plan, self.halo, sliding_window):
            theta_start = theta.detach().clone()   # (a torn resident launch leaves the loop's state partly written: start over from here)

            def make_loop():
                return fused_loop.FusedPatchLoop(plan, patch_size, sliding_window, theta_start.clone(), self.contrast_terms.get("image_variance", 0.0),
                                                 self.flow_terms.get("flow_norm", 0.0), self.flow_terms.get("image_gradient", 0.0),
                                                 self.omit_boundary, self.pad, self.halo, self.lr, capacity=n_iter,
                                                 w_gradient_magnitude=self.contrast_terms.get("gradient_magnitude", 0.0), theta_mask=mask,
                                                 blur_sigma=self.blur_sigma)

            loop = make_loop()
            try:
                losses = loop.run(n_iter, resident=None if self.resident is None else bool(self.resident) and loop.resident_supported())
            except fused_loop.ResidentStateTorn as e:   # as WindowPipeline does: re-solve the window from its start with the four launches
                logger.warning("%s; re-solving the window with the four launches", e)
                loop = make_loop()
                losses = loop.run(n_iter, resident=False)
            self.graphed, self.fused, self.loop_mode = loop.graphed, True, loop.last_run_mode
            self.loop_modes.append(loop.last_run_mode)  # (per pyramid scale, coarse to fine)
            self.history += losses.cpu().tolist()   # (one conversion: 600 float() calls cost 0.1 ms of a 12 ms window)
            return loop.theta
        self.fused = False
        if self.opt_method in SCIPY_METHODS:
            if self.fused_loop and fused_loop.objective_supported(self.contrast_terms, self.flow_terms, self.blur_sigma, plan,
                                                                  self.halo):
                loop = fused_loop.FusedPatchLoop(plan, patch_size, sliding_window, theta, self.contrast_terms.get("image_variance", 0.0),
                                                 self.flow_terms.get("flow_norm", 0.0), self.flow_terms.get("image_gradient", 0.0),
                                                 self.omit_boundary, self.pad, self.halo, self.lr, capacity=1,
                                                 w_gradient_magnitude=self.contrast_terms.get("gradient_magnitude", 0.0), theta_mask=mask)
                self.fused = True
                return self._run_scipy(None, theta, n_iter, value_and_grad=loop.value_and_grad)
            return self._run_scipy(evaluate, theta, n_iter)
        if self.opt_method != "Adam":
            raise NotImplementedError(f"optimizer.method {self.opt_method!r}: Adam or one of {SCIPY_METHODS}")

        def iteration(opt):
            opt.zero_grad(set_to_none=True)
            loss = evaluate()
            loss.backward()
            opt.step()
            return loss.detach()

        self.graphed = False
        losses = torch.zeros(max(n_iter, 1), dtype=torch.float32, device=plan.device)
        done = 0
        if self.use_graph and n_iter > 4:
            done = self._run_graphed(iteration, theta, losses, n_iter)
        if not self.graphed:
            opt = torch.optim.Adam([theta], lr=self.lr)
            for it in range(done, n_iter):
                losses[it] = iteration(opt)
        self.history += losses[:n_iter].cpu().tolist()
        return theta.detach()

    def _run_scipy(self, evaluate, theta: torch.Tensor, n_iter: int, value_and_grad=None) -> torch.Tensor:
        """Gradient-based scipy optimisers on the GPU objective, the role of the reference's
        ``scipy_autograd.minimize`` (src/solver/scipy_autograd/scipy_minimize.py:6-125): scipy drives float64 numpy
        parameters on the host; every function/gradient evaluation is one fused forward + backward on the device.
        Line searches need a differentiable start: with integer sensor coordinates the all-zero flow sits on the kink
        of the bilinear vote (every event exactly on a pixel centre), so warm-start these methods
        (``set_previous_frame_best_estimation``) or use Adam, which does not care."""
        import scipy.optimize as sopt

        shape = tuple(theta.shape)

        def fun(x):
            if value_and_grad is not None:  # fixed kernel pipeline, no autograd graph
                loss, grad = value_and_grad(torch.from_numpy(x.reshape(shape)))
                self.history.append(float(loss))
                return self.history[-1], grad.double().cpu().numpy().reshape(-1)
            with torch.no_grad():
                theta.copy_(torch.from_numpy(x.reshape(shape)).to(theta))
            theta.grad = None
            loss = evaluate()
            loss.backward()
            self.history.append(float(loss.detach()))
            return self.history[-1], theta.grad.detach().double().cpu().numpy().reshape(-1)

        x0 = theta.detach().double().cpu().numpy().reshape(-1)
        res = sopt.minimize(fun, x0, jac=True, method=self.opt_method, options={"maxiter": int(n_iter), **self.scipy_options})
        if not res.success:
            logger.warning(f"Unsuccessful optimization step! ({res.message})")
        self.scipy_result = res
        return torch.from_numpy(res.x.reshape(shape)).to(theta).detach()

    def _run_graphed(self, iteration, theta: torch.Tensor, losses: torch.Tensor, n_iter: int) -> int:
        """Adam loop as a replayed HIP graph.  The first iterations run eagerly on a side stream (they also create
        the plan workspace and Adam state), one iteration is captured, the rest are replays.  Returns the number of
        iterations performed; falls back to the eager loop (self.graphed = False) if capture is not possible."""
        warm = 3
        start = theta.detach().clone()
        try:
            opt = torch.optim.Adam([theta], lr=self.lr, capturable=True)
            side = torch.cuda.Stream(device=theta.device)
            side.wait_stream(torch.cuda.current_stream(theta.device))
            with torch.cuda.stream(side):
                for it in range(warm):
                    losses[it] = iteration(opt)
            torch.cuda.current_stream(theta.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            opt.zero_grad(set_to_none=True)
            # capture_begin/capture_end directly: the torch.cuda.graph() context adds gc.collect() + empty_cache()
            side.wait_stream(torch.cuda.current_stream(theta.device))
            with torch.cuda.stream(side):
                graph.capture_begin()
                try:
                    static_loss = iteration(opt)
                finally:
                    graph.capture_end()
            torch.cuda.current_stream(theta.device).wait_stream(side)
            for it in range(warm, n_iter):  # the capture itself executed nothing: iteration `warm` is the first replay
                graph.replay()
                losses[it] = static_loss
            self.graphed = True
            return n_iter
        except Exception as err:  # capture unsupported for some op: restart the whole loop eagerly
            logger.warning(f"HIP graph capture of the solver iteration failed ({err!r}); running eagerly")
            with torch.no_grad():
                theta.copy_(start)
            theta.grad = None
            self.graphed = False
            return 0

    def _param_range(self, key: str) -> Dict[str, float]:
        """{min, max} of a motion parameter: ``parameters: {trans_x: {min, max}}`` (this solver's YAMLs) or -- when
        ``parameters`` is the reference's LIST of parameter names (configs/hot_plate1.yaml:48-50) -- the sampler ranges of
        ``optimizer.parameters`` (:71-80), default +-30 px."""
        rng = self.param_ranges.get(key) if isinstance(self.param_ranges, dict) else None
        if rng is None:
            rng = ((self.slv_config.get("optimizer") or {}).get("parameters") or {}).get(key)
        return rng if isinstance(rng, dict) and "min" in rng and "max" in rng else {"min": -30.0, "max": 30.0}

    def _translation_loss(self, plan: EventPlan, theta: torch.Tensor) -> torch.Tensor:
        """-contrast of the IWE under the translation ``theta`` [2] (autograd through the 2-DoF tile-private kernels)."""
        iwe = plan.iwe_2dof(theta[None], pad=(self.pad, self.pad), halo=self.halo)[0]
        if self.blur_sigma > 0:
            iwe = EventImageConverter._gaussian_blur3(iwe, self.blur_sigma)
        total = 0.0
        for name, wgt in self.contrast_terms.items():
            fn = ops.image_variance if name == "image_variance" else ops.gradient_magnitude
            total = total + wgt * fn(iwe, self.omit_boundary)
        return -total

    def _translation_loop_fused(self, plan: EventPlan) -> bool:
        """Can the 2-DoF Adam loop run natively (``fused_loop.Fused2dofLoop``: ebos_cmax_2dof_solve_f32)?  The variance contrast
        (optionally of the 3-tap blurred image) on a binned plan -- compact, or the (x, y, dt) format of fractional (undistorted)
        source coordinates -- with a tile-private kernel configuration."""
        from ..event_plan import _slab_ok

        return (self.fused_loop and set(self.contrast_terms) == {"image_variance"} and not self.flow_terms
                and (plan.compact or plan.x is not None) and self.halo is not None and _slab_ok(plan, self.halo) and (not self.blur_sigma or min(plan.image_size) >= 2))

    def _adam_translation(self, plan: EventPlan, theta0: torch.Tensor, n_iter: int) -> torch.Tensor:
        """``n_iter`` Adam steps on (trans_x, trans_y) from ``theta0``: natively when the objective allows it (one C call
        enqueues the whole loop, no host synchronisation per iteration), else through autograd."""
        if self._translation_loop_fused(plan):
            def make_loop():
                return fused_loop.Fused2dofLoop(plan, theta0.detach().clone(), self.contrast_terms["image_variance"], self.omit_boundary, self.pad,
                                                self.halo, self.lr, capacity=max(n_iter, 1), blur_sigma=self.blur_sigma)

            loop = make_loop()
            try:
                losses = loop.run(n_iter, resident=None if self.resident is None else bool(self.resident) and loop.resident_supported())
            except fused_loop.ResidentStateTorn as e:   # theta0 is untouched (the loop works on its own copy): start over, four launches
                logger.warning("%s; re-solving the window with the four launches", e)
                loop = make_loop()
                losses = loop.run(n_iter, resident=False)
            self.fused, self.loop_mode = True, loop.last_run_mode
            self.history += losses.cpu().tolist()   # (one conversion: 600 float() calls cost 0.1 ms of a 12 ms window)
            return loop.theta
        self.fused = False
        theta = theta0.clone().requires_grad_(True)
        opt = torch.optim.Adam([theta], lr=self.lr)
        for _ in range(n_iter):
            opt.zero_grad(set_to_none=True)
            loss = self._translation_loss(plan, theta)
            loss.backward()
            opt.step()
            self.history.append(float(loss.detach()))
        return theta.detach()

    def _tpe_translation(self, contrast, rx, ry, batch: int = 32, n_candidates: int = 24, gamma: float = 0.25):
        """``sampler: TPE`` -- a tree-structured Parzen estimator in the shape optuna runs it (src/solver/generative_max_likelihood.py:
        216-219: ``n_startup_trials = max(10, n_iter // 10)`` uniform draws first), batched for the GPU: each round fits the two Parzen
        densities (Gaussian kernels on the best ``gamma`` quantile of the trials so far / on the rest, per parameter, bandwidth from
        the neighbour spacing, plus the uniform prior), draws ``n_candidates`` points per slot from the good density, keeps the one with
        the largest l(x) / g(x), and evaluates ``batch`` such picks as ONE sweep.  optuna itself is not installed here: the algorithm
        follows Bergstra et al. 2011 as optuna documents it, trial for trial it is not optuna's stream (parity unpinned)."""
        rs = np.random.RandomState(self.sampler_seed)
        lo, hi = np.array([rx["min"], ry["min"]], float), np.array([rx["max"], ry["max"]], float)
        n_total = max(1, self.n_iter)
        n_start = min(n_total, max(10, n_total // 10))
        xs = rs.uniform(lo, hi, (n_start, 2))
        vals = contrast(torch.from_numpy(xs.astype(np.float32))).double().cpu().numpy()

        def log_parzen(pts, x):  # [m, 2] kernels (+ the prior) -> log density at x [k, 2], product over the two parameters
            out = np.zeros(len(x))
            for d in range(2):
                mu = np.sort(pts[:, d])
                ext = np.concatenate([[lo[d]], mu, [hi[d]]])
                sig = np.clip(np.maximum(ext[1:-1] - ext[:-2], ext[2:] - ext[1:-1]), (hi[d] - lo[d]) / min(100.0, 1.0 + len(mu)), hi[d] - lo[d])
                mu = np.concatenate([mu, [0.5 * (lo[d] + hi[d])]])
                sig = np.concatenate([sig, [hi[d] - lo[d]]])   # the prior: one wide kernel in the middle of the range
                z = (x[:, d, None] - mu[None]) / sig[None]
                out += np.log(np.mean(np.exp(-0.5 * z * z) / (sig[None] * np.sqrt(2 * np.pi)), axis=1) + 1e-300)
            return out

        while len(xs) < n_total:
            k = min(batch, n_total - len(xs))
            order = np.argsort(-vals)                      # maximise the contrast
            n_good = max(1, int(np.ceil(gamma * len(xs))))
            good, bad = xs[order[:n_good]], xs[order[n_good:]] if len(xs) > n_good else xs
            picks = np.empty((k, 2))
            for j in range(k):
                comp = rs.randint(0, len(good) + 1, n_candidates)            # (index len(good) = the prior)
                spread = np.maximum((hi - lo) / max(4.0, np.sqrt(len(good) + 1.0)), 1e-6)
                centre = np.where((comp == len(good))[:, None], 0.5 * (lo + hi), good[np.minimum(comp, len(good) - 1)])
                width = np.where((comp == len(good))[:, None], hi - lo, spread)
                cand = np.clip(centre + rs.standard_normal((n_candidates, 2)) * width, lo, hi)
                picks[j] = cand[np.argmax(log_parzen(good, cand) - log_parzen(bad, cand))]
            v = contrast(torch.from_numpy(picks.astype(np.float32))).double().cpu().numpy()
            xs, vals = np.concatenate([xs, picks]), np.concatenate([vals, v])
        return torch.from_numpy(xs.astype(np.float32)), torch.from_numpy(vals.astype(np.float32))

    def _estimate_translation(self, plan: EventPlan) -> torch.Tensor:
        """``optimizer.method: grid`` -- exhaustive sweep over the parameter ranges (the optuna grid sampler of
        src/solver/generative_max_likelihood.py:238-255); ``random`` / ``TPE`` (or ``method: optuna`` + ``optimizer.sampler``, :215-226)
        -- n_iter uniform draws / a batched Parzen-estimator search inside the same ranges; each optionally refined by
        ``refine_iters`` Adam steps;
        ``Adam`` -- n_iter Adam steps on (trans_x, trans_y) from the warm start (or zero), the loop shape of :306-341."""
        if self.opt_method == "Adam":
            start = self._warm_start()
            theta0 = (torch.zeros(2, dtype=torch.float32, device=plan.device) if start is None else
                      to_gpu(start, device=plan.device, dtype=torch.float32).reshape(2))
            return self._adam_translation(plan, theta0, self.n_iter)
        # optimizer.method: "optuna" + optimizer.sampler (src/solver/generative_max_likelihood.py:215-226) or the sampler's name as the
        # method: grid / uniform (= sweep), random, TPE
        sampler = self.opt_method
        if sampler == "optuna":
            sampler = self.sampler
        sampler = {"sweep": "grid", "uniform": "grid", "tpe": "TPE"}.get(sampler, sampler)
        if sampler not in ("grid", "random", "TPE"):
            raise NotImplementedError(f"optimizer.method {self.opt_method!r} (sampler {self.sampler!r}) for a 2-DoF motion model: "
                                      "Adam, grid / uniform, random or TPE")
        rx, ry = self._param_range("trans_x"), self._param_range("trans_y")

        def contrast(thetas: torch.Tensor) -> torch.Tensor:  # one batched launch sequence for all hypotheses of a round
            return plan.variance_2dof(thetas.to(plan.device), self.omit_boundary, pad=(self.pad, self.pad), halo=self.halo)

        if sampler == "random":
            # optuna.samplers.RandomSampler (:220-221): n_iter independent uniform draws inside parameters.{min, max}; all of them
            # are independent hypotheses, so they are ONE batched sweep (optimizer.seed makes the draws reproducible)
            rs = np.random.RandomState(self.sampler_seed)
            draws = rs.uniform([rx["min"], ry["min"]], [rx["max"], ry["max"]], (max(1, self.n_iter), 2))
            grid = torch.from_numpy(draws.astype(np.float32))
            var = contrast(grid)
        elif sampler == "TPE":
            grid, var = self._tpe_translation(contrast, rx, ry)
        else:
            n = max(2, int(round(np.sqrt(max(self.n_iter, 4)))))
            gx = torch.arange(n, dtype=torch.float32) * ((rx["max"] - rx["min"]) / n) + rx["min"]  # np.arange(min, max, step), :238-255
            gy = torch.arange(n, dtype=torch.float32) * ((ry["max"] - ry["min"]) / n) + ry["min"]
            grid = torch.stack(torch.meshgrid(gx, gy, indexing="ij"), -1).reshape(-1, 2)
            var = contrast(grid)
        grid = grid.to(plan.device)
        self.history = [float(-v) for v in var.cpu()]
        self.sweep_grid, self.sweep_contrast = grid, var
        theta = grid[int(torch.argmax(var).item())]
        if self.refine_iters > 0:  # optimizer.refine_iters: Adam from the best grid point (gradient of the 2-DoF kernels)
            theta = self._adam_translation(plan, theta, self.refine_iters)
        return theta


def make_solver_class(base, name: str = "ContrastMaximization"):
    """``ContrastMaximization`` composed over ``base`` (a ``SolverBase``): constructor signature of the plugin surface
    (src/solver/base.py:64-71), ``estimate`` and the objective from the mixin, everything else -- ``preprocess``, the
    imagers / warpers, and with the reference's base its ``visualize_*`` / ``calculate_flow_error`` /
    ``save_flow_error_as_text`` -- from ``base``."""

    def __init__(self, orig_image_shape, crop_image_shape, calibration_parameter=None, solver_config=None, visualize_module=None):
        base.__init__(self, orig_image_shape, crop_image_shape, {} if calibration_parameter is None else calibration_parameter,
                      {} if solver_config is None else solver_config, visualize_module)
        self._cmax_setup()

    return type(name, (ContrastMaximizationMixin, base), {"__init__": __init__, "__doc__": ContrastMaximizationMixin.__doc__,
                                                          "__module__": __name__})


ContrastMaximization = make_solver_class(SolverBase)


def register_into(solver_module, names=("contrast_maximization", "cmax")):
    """Add the CMax solver to ANOTHER solver registry -- the reference's ``src.solver`` -- built over THAT module's
    ``SolverBase``, so that ``bos_event.py`` drives it unchanged: ``solver.collections[config["solver"]["method"]](...)``
    (bos_event.py:330-337), ``solv.preprocess`` / ``solv.estimate`` (:190-194) and the visualisation / error methods
    (:202-219) all resolve.  Returns the class."""
    cls = make_solver_class(solver_module.SolverBase)
    for n in names:
        solver_module.collections[n] = cls
    return cls
