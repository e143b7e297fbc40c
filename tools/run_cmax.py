#!/usr/bin/env python3
"""End-to-end run of the contrast-maximisation path with the driver protocol of the reference's bos_event.py
(parse config -> propagate_config -> build solver from the registry -> preprocess(events) -> estimate(events) -> report),
on a synthetic window (the recorded CCS sequences are not distributable).

    python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml           # this build's own YAML (patch-flow solver)
    python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json    # the REFERENCE's configs/hot_plate1.yaml,
                                                                                  # key for key (parsed data, make_golden.py --config)
    python tools/run_cmax.py --config_file tests/golden/config_hot_plate1.json --height 260 --width 346   # BASELINE configs[0] size

With the reference's file: 720x1280 and the region of interest rows 0:720, cols 320:960 as declared (:5-6, :23-26), motion model
2d-translation, Adam, n_iter 600, blur_sigma 3 (:46-70) are taken from it; ``solver.method`` (the release ships no CMax solver,
SURVEY F5) and the cost (its costs need frames) are the two overrides, ``--method`` / ``--cost``.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos  # noqa: E402


def synthetic_window(cfg):
    """Points translating with a smooth, spatially varying displacement (a schlieren-like bump)."""
    d = cfg["data"]
    h, w, n = int(d["height"]), int(d["width"]), int(d["n_events"])
    rs = np.random.RandomState(int(d.get("seed", 0)))
    per_point = 40
    pts = np.stack([rs.uniform(0, h - 1, n // per_point), rs.uniform(0, w - 1, n // per_point)], 1)
    cy, cx, sig, amp = h / 2, w / 2, min(h, w) / 4, float(d.get("max_displacement", 6.0))
    g = amp * np.exp(-((pts[:, 0] - cy) ** 2 + (pts[:, 1] - cx) ** 2) / (2 * sig ** 2))
    base = np.array(d.get("base_displacement", [3.0, -2.0]))  # the whole background texture shifts, plus the bump
    flow = np.stack([base[0] + g, base[1] - 0.5 * g], 1)  # true displacement over the window at each point
    t = rs.uniform(0, 1, (len(pts), per_point))
    x = (pts[:, None, 0] + t * flow[:, None, 0]).reshape(-1)
    y = (pts[:, None, 1] + t * flow[:, None, 1]).reshape(-1)
    # raw sensor events sit on integer pixels -- what the reference's loaders hand out (src/data_loader/ccs.py:57-66 reads the int16
    # columns; `data.warp` there warps the FRAMES with a homography, ccs.py:85-86,152-153; src/utils/event_utils.py:242-266 truncates
    # undistorted coordinates to int32).  --fractional keeps the sub-pixel coordinates (events rectified with a sub-pixel map).
    # (On integer pixels the variance has a local optimum at zero flow -- an event that sits on a pixel centre is not smeared --: the
    # 2-DoF Adam loop of the reference's YAML, started at zero, stays there on this synthetic window; the run then times the loop.)
    if not d.get("fractional_events", False):
        x, y = np.rint(x), np.rint(y)
    ev = np.stack([x, y, 10.0 + 0.0083 * t.reshape(-1), rs.randint(0, 2, x.size)], 1)
    ev = ev[(ev[:, 0] >= 0) & (ev[:, 0] < h) & (ev[:, 1] >= 0) & (ev[:, 1] < w)]
    return ev[np.argsort(ev[:, 2], kind="stable")], (h, w)


def load_config(path):
    if path.endswith(".json"):  # fixture made from the reference's YAML by tests/golden/make_golden.py --config
        return json.load(open(path))["input"]
    return yaml.safe_load(open(path))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_file", default=os.path.join(ROOT, "configs", "cmax_hot_plate1.yaml"))
    ap.add_argument("--method", default="contrast_maximization", help="overrides solver.method when that solver is not registered")
    ap.add_argument("--cost", default="image_variance", help="contrast cost used when cost_with_weight names none")
    ap.add_argument("--height", type=int, default=None, help="override data.height (and the ROI rows) -- e.g. 260")
    ap.add_argument("--width", type=int, default=None, help="override data.width (and the ROI columns) -- e.g. 346")
    ap.add_argument("--n-events", type=int, default=100_000)
    ap.add_argument("--n-iter", type=int, default=None, help="override solver.optimizer.n_iter")
    ap.add_argument("--fractional", action="store_true", help="keep the synthetic events' sub-pixel coordinates (events rectified with a sub-pixel map) "
                                                            "instead of rounding them to the sensor's integer pixels")
    args = ap.parse_args()
    cfg = load_config(args.config_file)
    d, cp = cfg["data"], cfg.setdefault("common_params", {})
    if args.height is not None or args.width is not None:  # the small-fixture size: whole frame as region of interest
        d["height"], d["width"] = int(args.height or d["height"]), int(args.width or d["width"])
        cp.update(xmin=0, xmax=d["height"], ymin=0, ymax=d["width"])
    for k, v in (("xmin", 0), ("xmax", d["height"]), ("ymin", 0), ("ymax", d["width"])):
        cp.setdefault(k, v)
    d.setdefault("n_events", args.n_events)
    if args.fractional:
        d["fractional_events"] = True
    ebos.utils.propagate_config(cfg)                                   # src/utils/config_utils.py:42-88
    scfg = cfg["solver"]
    overrides = {}
    if scfg.get("method") not in ebos.solver.collections:
        overrides["solver.method"] = [scfg.get("method"), args.method]
        scfg["method"] = args.method
    cww = scfg.get("cost_with_weight") or {}
    if not any(k in ("image_variance", "gradient_magnitude") for k in cww):
        overrides["solver.cost_with_weight"] = [cww, {args.cost: 1.0}]
        scfg["cost_with_weight"] = {args.cost: 1.0}
    if args.n_iter is not None:
        scfg.setdefault("optimizer", {})["n_iter"] = args.n_iter
    events, shape = synthetic_window(cfg)
    crop_shape = (d["crop_height"], d["crop_width"])
    solver = ebos.solver.collections[scfg["method"]](shape, crop_shape, calibration_parameter=None, solver_config=scfg,
                                                     visualize_module=None)
    n_in = len(events)
    t0 = time.perf_counter()
    events, period = solver.preprocess(events)                         # bos_event.py:190
    flow = solver.estimate(events)                                     # bos_event.py:192-194
    dt = time.perf_counter() - t0
    # ... and once more, warm (the first call loads the kernels' code objects and sizes the allocator's pools): what a window of a
    # recording costs after the first one -- plan build + the whole optimiser loop + the flow's way back to the host
    first_history = list(solver.history)
    t1 = time.perf_counter()
    flow2 = solver.estimate(events)
    dt_warm = time.perf_counter() - t1
    assert np.array_equal(flow, flow2) or np.allclose(flow, flow2, atol=1e-3), "the second estimate of the same window differs"
    solver.history = first_history
    iwe0 = solver.orig_imager.create_iwe(events, "bilinear_vote", sigma=0)
    warped, _ = solver.orig_warper.warp_event(events, flow, "dense-flow", solver.warp_direction)
    iwe1 = solver.orig_imager.create_iwe(warped, "bilinear_vote", sigma=0)
    roi = (slice(cp["xmin"], cp["xmax"]), slice(cp["ymin"], cp["ymax"]))
    print(json.dumps({"config_file": os.path.relpath(args.config_file, ROOT), "overrides": overrides, "events_in": n_in,
                      "events": int(len(events)), "image": list(shape), "crop": list(crop_shape), "roi": [cp[k] for k in ("xmin", "xmax", "ymin", "ymax")],
                      "motion_model": solver.motion_model, "optimizer": solver.opt_method, "blur_sigma": solver.blur_sigma,
                      "time_period_s": period, "solver_s": round(dt, 3), "solver_warm_s": round(dt_warm, 5),
                      "us_per_iteration_warm": round(dt_warm / max(len(solver.history), 1) * 1e6, 2),
                      "fused": bool(getattr(solver, "fused", False)), "loop_mode": getattr(solver, "loop_mode", None),
                      "iterations": len(solver.history), "loss_first": solver.history[0], "loss_last": solver.history[-1],
                      "variance_unwarped": float(iwe0.var(ddof=1)), "variance_warped": float(iwe1.var(ddof=1)),
                      "flow_mean_in_roi": [float(flow[0][roi].mean()), float(flow[1][roi].mean())],
                      "true_base_displacement": list(d.get("base_displacement", [3.0, -2.0])),
                      "flow_abs_max": float(np.abs(flow).max())}))


if __name__ == "__main__":
    main()
