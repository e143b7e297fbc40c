#!/usr/bin/env python3
"""Headline benchmark: Mevents/s through the fused warp + IWE (+ variance contrast) pass at 1280x720.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one window of 10 M synthetic events (recipe of
src/utils/event_utils.py:40-47, seed = rank) at 1280x720, dense per-pixel flow U(-30, 30)
(src/utils/flow_utils.py:29), normalised time, reference time "first", variance contrast.
One "step" = one evaluation of the contrast objective on the resident window (ebos_iwe_dense_slab_f32):
    fused warp + bilinear splat into tile-private LDS images -> slab combine (writes the IWE) -> variance.
The window is resident in HBM in its plan form (SoA f32, binned by source tile: built once per window
and reused by every solver iteration; its one-off build time is reported as plan_build_ms and is NOT
in the timed region).  With N > 1 every rank owns an independent window (weak scaling, no collective
in the data path); timing is barrier + synchronize bracketed, max over ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 720, 1280
N_EVENTS = 10_000_000
FLOW_MAX = 30.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def synth_window(n, seed):
    rs = np.random.RandomState(seed)
    x = rs.randint(0, H, n)
    y = rs.randint(0, W, n)
    t = np.sort(rs.uniform(0.0, 0.5, n))
    p = rs.randint(0, 2, n)
    ev = np.stack([x, y, t, p], axis=1).astype(np.float64)
    flow = np.random.RandomState(1000 + seed).uniform(-FLOW_MAX, FLOW_MAX, (2, H, W))
    return ev, flow


def cpu_baseline(ev, flow, sample):
    """The oracle's op-for-op torch-CPU restatement of the reference path (kind 'port'), timed on this
    host's cores on a bounded sample of the same window."""
    from oracle import ebos_oracle as O

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # torch's intra-op pool does not scale to hundreds of threads on this memory-bound op chain:
    # calibrate on a 2 M-event sample and keep the fastest thread count.
    e_cal = torch.from_numpy(ev[:min(sample, 2_000_000)])  # large enough to leave the caches, like the timed run
    f_cal = torch.from_numpy(flow)
    best, threads = None, 1
    for cand in [c for c in (8, 16, 32, 64, 128, 256) if c <= avail] or [avail]:
        torch.set_num_threads(cand)
        O.iwe_dense(e_cal, f_cal, (H, W))
        t0 = time.perf_counter()
        O.iwe_dense(e_cal, f_cal, (H, W))
        dt_ = time.perf_counter() - t0
        if best is None or dt_ < best:
            best, threads = dt_, cand
    torch.set_num_threads(threads)
    out = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        e = torch.from_numpy(ev[:sample]).to(dt)
        f = torch.from_numpy(flow).to(dt)
        times = []
        for rep in range(6):
            t0 = time.perf_counter()
            iwe = O.iwe_dense(e, f, (H, W))
            O.image_variance(iwe)
            times.append(time.perf_counter() - t0)
            if sum(times) > 40.0 and rep >= 2:  # a slow host: stay within the bench's few minutes
                break
        out[name] = sample / statistics.median(times[1:]) / 1e6
        reps = len(times) - 1
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(out["f64"], 3), "unit": "Mevents/s", "cores": threads, "host_cpus": avail, "cpu_model": model,
            "kind": "port",
            "sample": (f"{'the whole window' if sample >= len(ev) else 'first ' + str(sample) + ' events of the window'} "
                       f"({min(sample, len(ev))} events), fwd warp+IWE+variance, torch-CPU fp64 (reference default dtype), "
                       f"median of {reps} after 1 warm-up; fp32 on the same sample: {out['f32']:.2f} Mevents/s"),
            "value_f32": round(out["f32"], 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--events", type=int, default=N_EVENTS)
    ap.add_argument("--tile", type=int, nargs=2, default=[0, 0], help="source tile (0 0 = choose_tile: 45x80 at 1280x720)")
    ap.add_argument("--halo", type=int, default=32)
    ap.add_argument("--splits", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compact", action="store_true", help="read the 12 B/event (x, y, dt) plan instead of 6 B/event")
    ap.add_argument("--cpu-sample", type=int, default=N_EVENTS, help="events of the window the CPU baseline is timed on")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun also for a world of 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)

    import event_based_bos_amd as ebos
    from event_based_bos_amd import _hip

    lib = _hip.require_gpu()
    n = args.events
    ev, flow_np = synth_window(n, seed=rank)
    ev_gpu = torch.from_numpy(ev).to(dev)
    flow = torch.from_numpy(flow_np).float().to(dev)
    if args.tile[0] <= 0:
        args.tile = list(ebos.event_plan.choose_tile((H, W), args.halo))
    plan_first_ms = plan_build_ms = 0.0
    for attempt in range(2):  # the first build also pays one-off allocator / code-object costs: report the second
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan = ebos.EventPlan.build(ev_gpu, (H, W), "first", True, tile=tuple(args.tile))
        torch.cuda.synchronize()
        plan_build_ms = (time.perf_counter() - t0) * 1e3
        plan_first_ms = plan_first_ms or plan_build_ms
    del ev_gpu

    import ctypes

    nws = int(lib.ebos_iwe_slab_workspace_bytes(H, W, args.tile[0], args.tile[1], args.halo, args.splits, 0, 0))
    ws = torch.zeros(nws, dtype=torch.uint8, device=dev)  # zero-filled once (spill section stays zero)
    out = torch.empty(1, dtype=torch.float32, device=dev)
    moments = torch.empty((1, 2), dtype=torch.float64, device=dev)
    iwe = torch.empty((H, W), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()

    cptrs = plan._compact_ptrs() if (plan.compact and not args.no_compact) else (None, None, None)
    bytes_per_event = 6.0 if cptrs[0] else 12.0

    def step():
        # one objective evaluation: tile accumulate -> slab combine (writes the IWE) -> variance
        _hip.check(lib.ebos_iwe_dense_slab_f32(P(plan.x), P(plan.y), P(plan.dt), None, *cptrs, P(plan.key_offsets), plan.n, P(flow),
                                               H, W, args.tile[0], args.tile[1], args.halo, args.splits, 0, 0, P(ws), nws,
                                               P(iwe), 1, 0, P(out), P(moments), P(plan.part_table), stream), "ebos_iwe_dense_slab")

    def sync_all():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    _hip.check(lib.ebos_profile_start(args.steps), "ebos_profile_start")  # HIP events around the dominant kernel
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    buf = (ctypes.c_float * args.steps)()
    nrec = lib.ebos_profile_stop(buf, args.steps)
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = statistics.mean(buf[i] for i in range(nrec)) if nrec else float("nan")
    contrast = float(out.item())

    # extra (outside the timed region): forward + backward of the objective, direct C-ABI calls
    #   slab forward + variance -> tile-private backward (variance gradient folded in from the moments)
    upstream = torch.full((1,), -1.0, dtype=torch.float32, device=dev)  # loss = -variance
    d_flow = torch.empty((2, H, W), dtype=torch.float32, device=dev)

    def step_fwd_bwd():
        step()
        _hip.check(lib.ebos_iwe_dense_tiled_bwd_f32(P(plan.x), P(plan.y), P(plan.dt), None, *cptrs, P(plan.key_offsets), plan.n,
                                                    P(flow), H, W, args.tile[0], args.tile[1], args.halo, 0, 0, P(iwe), None,
                                                    0, P(d_flow), None, P(moments), P(upstream), None, P(ws), nws,
                                                    P(plan.part_table) if args.splits == 0 else None, stream), "ebos_iwe_dense_tiled_bwd")

    for _ in range(3):
        step_fwd_bwd()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    reps = max(5, args.steps // 2)
    for _ in range(reps):
        step_fwd_bwd()
    torch.cuda.synchronize()
    fwdbwd_ms = (time.perf_counter() - t1) / reps * 1e3

    # Informative only (NOT `value`): independent evaluations -- hypotheses of a sweep, windows of a recording -- in flight on
    # three HIP streams, each with its own workspace and outputs: the small combine / finalize kernels of one evaluation
    # run in the wave slots the one-workgroup-per-CU accumulate kernel of another leaves free.
    lanes = []
    for _ in range(3):
        lanes.append((torch.cuda.Stream(device=dev), torch.zeros(nws, dtype=torch.uint8, device=dev), torch.empty_like(iwe),
                      torch.empty_like(out), torch.empty_like(moments)))

    def overlapped(count):
        for k in range(count):
            st, ws_k, iwe_k, out_k, mom_k = lanes[k % 3]
            _hip.check(lib.ebos_iwe_dense_slab_f32(P(plan.x), P(plan.y), P(plan.dt), None, *cptrs, P(plan.key_offsets), plan.n,
                                                   P(flow), H, W, args.tile[0], args.tile[1], args.halo, args.splits, 0, 0,
                                                   P(ws_k), nws, P(iwe_k), 1, 0, P(out_k), P(mom_k), P(plan.part_table),
                                                   st.cuda_stream), "ebos_iwe_dense_slab")

    torch.cuda.synchronize()
    overlapped(6)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    overlapped(3 * max(10, args.steps))
    torch.cuda.synchronize()
    overlapped_ms = (time.perf_counter() - t2) / (3 * max(10, args.steps)) * 1e3

    # Informative only: one Adam iteration of the patch-flow solver on the same window (BASELINE configs[3] shape: 30x40 patch grid
    # -> 1280x720 flow, image_variance + flow_norm), the whole loop enqueued by one C call (ebos_cmax_patch_solve_f32)
    solver_extra = None
    if rank == 0:
        try:
            from event_based_bos_amd.solver.fused_loop import FusedPatchLoop

            gh, gw = ebos.solver.patch_grid_shape((H, W), (24, 32), (24, 32))
            sl = FusedPatchLoop(plan, (24, 32), (24, 32), torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, lr=0.1, capacity=260)
            sl.run(10)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            sl.run(200)
            torch.cuda.synchronize()
            solver_extra = {"us_per_iteration": round((time.perf_counter() - t4) / 200 * 1e6, 1), "events": plan.n,
                            "patch_grid": [gh, gw], "objective": "image_variance + 0.001 flow_norm, Adam",
                            "event_kernels_sample_the_patch_grid": bool(sl.sample_grid),
                            "note": "informative, not `value`: forward + backward + Adam step per iteration"}
            del sl
        except Exception as err:  # the headline measurement must not depend on the solver layer
            solver_extra = {"error": repr(err)}

    # next to the nominal peak, a bandwidth this box actually delivers -- a device-to-device copy of 1 GiB
    # (read + write bytes counted), best of 5
    a_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b_buf = torch.empty_like(a_buf)
    copy_gbs = 0.0
    for _ in range(6):
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        b_buf.copy_(a_buf)
        torch.cuda.synchronize()
        copy_gbs = max(copy_gbs, 2.0 * a_buf.numel() * 4 / (time.perf_counter() - t3) / 1e9)
    del a_buf, b_buf

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed / 1e6
        # 12 B/event (x, y, dt; p unused) + flow read (8 B/px) + IWE write (4 B/px)
        algo_bytes = 12.0 * plan.n + 12.0 * H * W
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        format_bytes = bytes_per_event * plan.n + 12.0 * H * W  # what the plan format actually stores per event
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("iwe_slab_accumulate_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mevents/sec warped+IWE at 1280x720", "value": round(value, 2), "unit": "Mevents/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 10M synthetic events, 1280x720 dense per-pixel flow U(-30,30), "
                                   "variance cost, fwd objective (tile accumulate + slab combine + variance)",
                       "events_per_gpu": n, "events_in_plan": plan.n, "height": H, "width": W,
                       "layout": ("compact SoA (u16 tile-local pixel + f32 dt, 6 B/event)" if cptrs[0] else "SoA f32 (x,y,dt), 12 B/event")
                                 + f", binned by source tile {args.tile[0]}x{args.tile[1]}, halo {args.halo}, splits {args.splits}", "parallelism": f"windows sharded, {world} rank(s), no collective"},
            "roofline": {"bound": "hbm", "kernel": "iwe_slab_accumulate_kernel", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": algo_bytes,
                         "measured_copy_GBps": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 4),
                         "plan_format_bytes": format_bytes,
                         "plan_format_GBps": round(format_bytes / (kernel_ms * 1e-3) / 1e9, 1)},
            "plan_build_ms": round(plan_build_ms, 2), "plan_build_first_call_ms": round(plan_first_ms, 2), "fwd_bwd_ms": round(fwdbwd_ms, 4),
            "fwd_bwd_mevents_per_s": round(n / fwdbwd_ms / 1e3, 2), "contrast": contrast,
            "independent_evaluations_on_3_streams": {"ms_per_evaluation": round(overlapped_ms, 4),
                                                     "mevents_per_s": round(n / overlapped_ms / 1e3, 2),
                                                     "note": "informative, not `value`: 3 evaluations in flight, own workspaces"},
        }
        if solver_extra is not None:
            line["solver_iteration"] = solver_extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ev, flow_np, min(args.cpu_sample, n))
            line["speedup_vs_cpu_port_f64"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
