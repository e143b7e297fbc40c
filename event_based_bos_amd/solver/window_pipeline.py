"""Windows of one recording through the fused solver loop, several at a time and with ingest under compute.

The reference's driver walks a recording window by window (bos_event.py:144-220): load the events of the window,
``solver.estimate(events)``, next window.  The windows are independent, so on one GPU

  * ``n_concurrent`` windows are solved at once, each on its own HIP stream (``ebos_cmax_patch_solve_many_f32``
    enqueues their iterations alternately): an event kernel occupies every CU with one workgroup, and the small
    kernels of another window's iteration (slab combine, regularisers, upsample, Adam) run in the wave slots it
    leaves free;
  * the next group of windows is ingested -- raw sensor columns (9 B/event) from pinned host memory, expanded and
    binned into an ``EventPlan`` on the device -- on a separate stream while the current group iterates.  The solve of
    a group is one asynchronous native call, so the host is free for that as soon as the launches are enqueued.

Measured on MI355X (tools/bench_pipeline.py, 8 windows x 2 M events at 1280x720, 600 iterations): 45.7 ms per window
with per-window ``estimate`` on host float64 windows, 43.9 ms with one window at a time (ingest hidden), 35.8 ms with
two, 26.8 ms with three at once -- and 51.7 ms with four: ROCm multiplexes streams onto 4 hardware queues per process, so
three solver streams + the ingest stream is the most that runs truly concurrently there (the package asks for 16 queues at
import since round 5; on a small sensor -- 346 x 260 -- eight windows run side by side as resident launches on 45 x 80 tiles:
2.3 ms per 600-iteration window, the default there).

Only the objective family of ``fused_loop`` is pipelined; ``run`` raises for any other solver configuration (use
``solver.estimate`` per window then).  Across ranks, windows are dealt out with ``sharding.shard_units``.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _hip, ops
from .._hip import check, stream_ptr
from .._staging import to_gpu
from ..data_loader import RawEventStore
from ..event_plan import EventPlan
from . import fused_loop
from .contrast_maximization import ContrastMaximization, patch_grid_shape


# The pipelines of a process share their streams.  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default),
# in creation order: two streams that land on the same queue run their kernels one after the other -- resident launches of two
# windows that fit the device side by side then take turns (346 x 260, 100 k events, two windows in flight: 10.5 ms per window
# instead of 5.6, for every SECOND pipeline object of a process: its fresh streams aliased the first one's).  Streams created once
# keep the mapping the first pipeline got; where more windows in flight are wanted than queues exist, GPU_MAX_HW_QUEUES=8 in the
# environment of the process (before its first HIP call) gives every stream a queue of its own.
_STREAM_POOL = {}


def _pooled_streams(device: torch.device, n: int):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    pool = _STREAM_POOL.setdefault(key, {"ingest": None, "solve": []})
    with _hip.on_device(device):
        if pool["ingest"] is None:
            # high priority: the few short ingest kernels must not queue behind thousands of solver launches (the plan
            # build ends in a host read-back, and the host is what enqueues the next group)
            pool["ingest"] = torch.cuda.Stream(device=device, priority=-1)
        while len(pool["solve"]) < n:
            pool["solve"].append(torch.cuda.Stream(device=device))
    return pool["ingest"], pool["solve"][:n]


RESIDENT_TILES = ((32, 32), (32, 64), (45, 80))   # the tiles with resident kernels (cmax_resident_<tile>.hip; halo 32 / run-time windows)


class WindowPipeline(object):
    def __init__(self, solver: ContrastMaximization, n_concurrent: Optional[int] = None, device="cuda", resident: Optional[bool] = None):
        # the patch-flow solver, or the 2-DoF Adam loop of the reference's shipped YAML (configs/hot_plate1.yaml:47,70)
        self.two_dof = solver.motion_model in ("2d-translation", "rigid-optical-flow")
        if solver.motion_model != "dense-flow" and not (self.two_dof and solver.opt_method == "Adam"):
            raise NotImplementedError("WindowPipeline drives the patch-flow (dense-flow) solver and the 2-DoF Adam loop")
        # n_concurrent None: three windows in flight -- eight where the resident loops of eight windows fit the device side by side
        # on the largest resident tile (a small sensor: 346 x 260 takes 2.3 ms per window with eight in flight, 3.2 - 3.5 with four;
        # at 1280 x 720, where one window fills the device, eight in flight cost 6.3 ms against 5.8 with three)
        auto_n = n_concurrent is None
        self.solver, self.n_concurrent = solver, 3 if auto_n else max(1, int(n_concurrent))
        # resident (default: the solver's optimizer.resident, else on unless EBOS_RESIDENT=0): each window's loop as one resident
        # launch where the geometry allows it; False = four launches per iteration, the windows of a group interleaved on streams
        if resident is None:
            resident = getattr(solver, "resident", None)
        self.resident = bool(resident) if resident is not None else os.environ.get("EBOS_RESIDENT", "1") != "0"
        self.resident_fallbacks: List[int] = []
        self.device = torch.device(device)
        self.lib = _hip.require_gpu()
        self.histories: List[List[float]] = []
        # the streams live as long as the process (_pooled_streams): torch's caching allocator pools blocks per stream, so fresh
        # streams per run would turn every buffer of every window into a new hipMalloc -- and fresh streams per pipeline may alias
        # hardware queues
        with _hip.on_device(self.device):
            # Resident launches run side by side only while all their workgroups fit the device at once (cmax_resident.hip): a small
            # sensor's window (99 tiles at 346 x 260) leaves room for a second one, and a group of three would run 2 + 1 -- the group
            # size is rounded up to a multiple of the windows that fit
            self.tile = tuple(solver.plan_tile())
            if self.resident:
                H, W = solver.orig_image_shape
                n_cu = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
                wgs = lambda t: -(-H // t[0]) * -(-W // t[1])   # noqa: E731  (one workgroup per source tile, one workgroup per CU)
                # An iteration of a resident loop on a small sensor is latency (two grid-wide exchanges), not work: fewer, larger
                # tiles per window cost a window alone a little and let more windows run side by side.  Unless the solver names its
                # tile, the pipeline takes the resident tile with the most workgroups for which the requested windows all fit
                # (346 x 260: two windows -> 32 x 32, 99 workgroups each; four -> 32 x 64, 54; eight -> 45 x 80, 30).
                if solver.tile is None and solver.halo in ("auto", 32):
                    slides = [] if self.two_dof else [sl for _, sl, _ in solver.pyramid_scales()]
                    usable = [t for t in RESIDENT_TILES   # (every scale's sliding window on the grid-sampling route with this tile)
                              if all(self.lib.ebos_patch_fused_supported(t[0], t[1], 32, int(sl[0]), int(sl[1])) for sl in slides)]
                    if auto_n and _hip.hw_queues() >= 10 and any(wgs(t) * 8 <= n_cu for t in usable):
                        self.n_concurrent = 8
                    # (each window in flight needs a hardware queue of its own beside the ingest and the default stream's)
                    want = max(1, min(self.n_concurrent, _hip.hw_queues() - 2))
                    fits = [t for t in usable if wgs(t) * want <= n_cu]
                    if fits:
                        self.tile = max(fits, key=wgs)
                    elif wgs(self.tile) > n_cu // 2:   # (not even two of the default tile: keep it, one window at a time)
                        pass
                    elif usable:
                        self.tile = min(usable, key=wgs)
                fit = max(1, n_cu // wgs(self.tile))
                # (whole rounds of the windows that fit side by side; leaving an eighth of the CUs to the ingest stream's plan builds --
                # seven 30-workgroup windows instead of eight -- was measured slower: 2.94 against 2.69 ms per window)
                self.n_concurrent = -(-self.n_concurrent // fit) * fit
            self.ingest_stream, self.streams = _pooled_streams(self.device, self.n_concurrent)

    # ------------------------------------------------------------------ stages
    def _ingest(self, store: RawEventStore, window: Tuple[int, int]) -> EventPlan:
        s = self.solver
        plan = store.plan(window[0], window[1], s.orig_image_shape, s.warp_direction, True, tile=self.tile, device=self.device,
                          deferred=True, emit="compact")  # no host read-back: the host never waits for the GPU until the end;
        # lean build: the fused loop reads only the compact events and offsets (0.09 ms instead of 0.4 per 2 M-event window)
        if self.two_dof:
            if not s._translation_loop_fused(plan):
                raise NotImplementedError("this 2-DoF configuration is outside the native loop (variance contrast, optionally blurred, no "
                                          "regulariser): call solver.estimate(store.load_event(i0, i1)) per window instead")
            return plan
        if not fused_loop.supported(s.contrast_terms, s.flow_terms, s.blur_sigma, s.opt_method, plan, s.halo, s.sliding_window):
            raise NotImplementedError("this solver configuration is outside the fused objective family: "
                                      "call solver.estimate(store.load_event(i0, i1)) per window instead")
        return plan

    def _solve_group(self, plans: Sequence[EventPlan], streams: Sequence[torch.cuda.Stream],
                     resident: Optional[bool] = None) -> List[dict]:
        """Enqueue the whole coarse-to-fine solve of every plan of the group; nothing here waits for the GPU.
        Only the patch flow and the losses of a window outlive this call: its images, workspace and plan go back to
        the caching allocator (stream-ordered reuse by the next group -- a fresh hipMalloc would wait for the GPU to
        drain and serialise the pipeline)."""
        s = self.solver
        H, W = s.orig_image_shape
        thetas = [None] * len(plans)
        losses = [[] for _ in plans]
        statuses = [[] for _ in plans]
        modes = [[] for _ in plans]
        resident = self.resident if resident is None else resident
        if self.two_dof:
            return self._solve_group_2dof(plans, streams, resident)
        for patch_size, sliding_window, n_iter in s.pyramid_scales():
            gh, gw = patch_grid_shape((H, W), patch_size, sliding_window)
            loops = []
            for w, plan in enumerate(plans):
                with torch.cuda.stream(streams[w]):
                    if thetas[w] is None:
                        init = torch.zeros((2, gh, gw), dtype=torch.float32, device=self.device)
                    else:
                        init = torch.nn.functional.interpolate(thetas[w][None], size=(gh, gw), mode="bilinear",
                                                               align_corners=False)[0]
                    mask = s.patch_mask(plan, patch_size, sliding_window)
                    if mask is not None:
                        init = init * mask
                    loops.append(fused_loop.FusedPatchLoop(
                        plan, patch_size, sliding_window, init, s.contrast_terms.get("image_variance", 0.0),
                        s.flow_terms.get("flow_norm", 0.0), s.flow_terms.get("image_gradient", 0.0), s.omit_boundary, s.pad,
                        s.halo, s.lr, capacity=n_iter, w_gradient_magnitude=s.contrast_terms.get("gradient_magnitude", 0.0),
                        theta_mask=mask, blur_sigma=s.blur_sigma))
            if resident and all(lp.resident_supported() for lp in loops):
                # every window's loop as ONE resident launch (30 us per iteration at 2 M events against 43 as four launches).  A
                # resident workgroup owns its CU: launches of different streams run side by side only while all their workgroups fit
                # the device together -- cmax_resident.hip orders them itself -- and the ingest stream's short kernels slip in
                # between.  Nothing waits here: the status words are looked at when the results are collected (``run``)
                for w, lp in enumerate(loops):
                    with torch.cuda.stream(streams[w]):
                        statuses[w].append(lp.enqueue_resident(n_iter))
                    modes[w].append("resident")
            else:
                for w in range(len(loops)):
                    modes[w].append("pipeline")
                problems = (_hip.CmaxPatchProblem * len(loops))(*[lp.problem() for lp in loops])
                handles = (ctypes.c_void_p * len(loops))(*[st.cuda_stream for st in streams[:len(loops)]])
                with _hip.on_device(self.device):
                    check(self.lib.ebos_cmax_patch_solve_many_f32(problems, handles, len(loops), int(n_iter)),
                          "ebos_cmax_patch_solve_many")
            for w, lp in enumerate(loops):
                with torch.cuda.stream(streams[w]):
                    thetas[w] = lp.theta.clone()
                    losses[w].append(lp.losses[:n_iter].clone())
            last = (patch_size, sliding_window)
        for w, plan in enumerate(plans):  # the plan was built on the ingest stream and read on streams[w]
            for t in (plan.x, plan.y, plan.dt, plan.p, plan.key_offsets, plan.perm, plan.grp_offsets, plan.cpix, plan.cdt,
                      plan.part_table):
                if t is not None:
                    t.record_stream(streams[w])
        return [dict(theta=thetas[w], losses=losses[w], patch=last, counts=plans[w].__dict__.get("_counts"), status=statuses[w],
                     modes=modes[w]) for w in range(len(plans))]

    def _solve_group_2dof(self, plans: Sequence[EventPlan], streams: Sequence[torch.cuda.Stream], resident: bool) -> List[dict]:
        """The 2-DoF Adam loop (``Fused2dofLoop``) of every plan of the group, each on its stream: one resident launch per window, or
        the natively enqueued four (five) launches per iteration.  Nothing waits for the GPU."""
        s = self.solver
        out = []
        # the start ``estimate`` honours too: the driver's set_previous_frame_best_estimation (src/solver/base.py:355-361), else zero
        start = s._warm_start()
        theta0 = (torch.zeros(2, dtype=torch.float32, device=self.device) if start is None else
                  to_gpu(start, device=self.device, dtype=torch.float32).reshape(2))
        for w, plan in enumerate(plans):
            with torch.cuda.stream(streams[w]):
                loop = fused_loop.Fused2dofLoop(plan, theta0.clone(), s.contrast_terms["image_variance"], s.omit_boundary, s.pad, s.halo, s.lr,
                                                capacity=max(s.n_iter, 1), blur_sigma=s.blur_sigma)
                status, mode = [], "pipeline"
                if resident and loop.resident_supported():
                    status, mode = [loop.enqueue_resident(s.n_iter)], "resident"
                else:
                    with _hip.on_device(self.device):
                        check(self.lib.ebos_cmax_2dof_solve_f32(ctypes.byref(loop.problem()), int(s.n_iter), stream_ptr()), "ebos_cmax_2dof_solve")
                out.append(dict(theta=loop.theta.clone(), losses=[loop.losses[:s.n_iter].clone()], patch=None,
                                counts=plan.__dict__.get("_counts"), status=status, modes=[mode]))
            for t in (plan.x, plan.y, plan.dt, plan.p, plan.key_offsets, plan.perm, plan.grp_offsets, plan.cpix, plan.cdt, plan.part_table):
                if t is not None:
                    t.record_stream(streams[w])
        return out

    # ------------------------------------------------------------------ driver
    def run(self, store: RawEventStore, windows: Sequence[Tuple[int, int]]) -> List[np.ndarray]:
        """Dense flow [2, H, W] (float64 numpy, like ``estimate``) of every (start_index, end_index) window."""
        dev = self.device
        with _hip.on_device(dev):
            ingest, streams = self.ingest_stream, self.streams
            groups = [list(windows[i:i + self.n_concurrent]) for i in range(0, len(windows), self.n_concurrent)]
            pending: List[dict] = []

            def ingest_group(group):   # one event per window: a window's stream waits for ITS plan, not for the group's last one
                plans, ready = [], []
                with torch.cuda.stream(ingest):
                    for wnd in group:
                        plans.append(self._ingest(store, wnd))
                        ev = torch.cuda.Event()
                        ev.record(ingest)
                        ready.append(ev)
                return plans, ready

            nxt = ingest_group(groups[0]) if groups else None
            resident = self.resident
            for g in range(len(groups)):
                plans, ready = nxt
                for st, ev in zip(streams, ready):
                    st.wait_event(ev)
                # A recording whose windows the resident kernel refuses (crowded tiles: status -104, or flows beyond its windows) would
                # otherwise pay for every window twice -- once here and once, one by one, in the re-solve below: as soon as a launch
                # of an earlier group is KNOWN to have ended early, the rest of the run takes the four launches
                if resident and g == 1:
                    # (once per run, one blocking read-back: the first group's verdicts decide for the recording -- the host is
                    # otherwise so far ahead that no verdict would arrive in time.  The status words are copies enqueued on the
                    # windows' side streams: those are waited for first -- read from the current stream, the words would be
                    # whatever the allocator left in them)
                    first = [sw for r in pending for sw in r["status"]]
                    if first:
                        for st in streams:
                            st.synchronize()
                        if bool((torch.cat(first) != 0).any().item()):
                            resident = False
                solved = self._solve_group(plans, streams, resident=resident)   # asynchronous: returns once enqueued
                for r, wnd in zip(solved, groups[g]):
                    r["window"] = wnd
                pending += solved
                nxt = ingest_group(groups[g + 1]) if g + 1 < len(groups) else None  # ... so this overlaps with it
            for st in streams:
                st.synchronize()
            # a resident launch that ended early (a displacement beyond the largest LDS window, a wait past its cap) left its
            # window's patch flow where it was: such a window is solved again, as four launches per iteration
            words = [sw for r in pending for sw in r["status"]]
            owner = [k for k, r in enumerate(pending) for _ in r["status"]]
            bad = torch.cat(words).cpu().numpy() != 0 if words else np.zeros(0, dtype=bool)   # (one read-back for all launches)
            self.resident_fallbacks = sorted({owner[i] for i in np.flatnonzero(bad)})
            for i in range(0, len(self.resident_fallbacks), len(streams)):   # (in groups, like the first pass)
                ks = self.resident_fallbacks[i:i + len(streams)]
                plans, ready = ingest_group([pending[k]["window"] for k in ks])
                for st, ev in zip(streams, ready):
                    st.wait_event(ev)
                for k, redo in zip(ks, self._solve_group(plans, streams, resident=False)):
                    pending[k].update(redo)
            if self.resident_fallbacks:
                for st in streams:
                    st.synchronize()
            H, W = self.solver.orig_image_shape
            # (ONE read-back per kind of result for the whole run: per window, a copy and its synchronisation cost 0.1 - 0.3 ms of host
            # time behind the last kernel -- a fifth of what a 346 x 260 window takes with eight windows in flight)
            parts = [part for r in pending for part in r["losses"]]
            flat = torch.cat(parts).cpu().numpy().astype(np.float64) if parts else np.zeros(0)
            self.histories, o = [], 0
            for r in pending:
                n_r = sum(int(part.numel()) for part in r["losses"])
                self.histories.append(flat[o:o + n_r].tolist())
                o += n_r
            self.patch_flows = [r["theta"] for r in pending]
            self.window_modes = [list(r["modes"]) for r in pending]   # per window and pyramid scale: how its final solve ran
            cnt = [r["counts"][:1] for r in pending if r["counts"] is not None]
            cnt = iter(torch.cat(cnt).cpu().tolist() if cnt else [])
            self.dropped_events = [int(next(cnt)) if r["counts"] is not None else 0 for r in pending]
            if self.two_dof:   # dense flow equivalent of theta: -theta everywhere (src/warp.py:186-187), as ``estimate`` returns it
                th = (-torch.stack([r["theta"] for r in pending])).cpu().numpy().astype(np.float64) if pending else np.zeros((0, 2))
                return [np.broadcast_to(t.reshape(2, 1, 1), (2, H, W)).copy() for t in th]
            if not pending:
                return []
            dense = torch.stack([ops.upsample_patch_flow(r["theta"], r["patch"][0], r["patch"][1], (H, W)) for r in pending])
            return list(dense.cpu().numpy().astype(np.float64))
