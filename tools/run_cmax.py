#!/usr/bin/env python3
"""End-to-end run of the contrast-maximisation path with the driver protocol of the reference's bos_event.py
(parse YAML -> build solver from the registry -> preprocess(events) -> estimate(events) -> report), on a synthetic
window described by the YAML's ``data`` section (the recorded CCS sequences are not distributable).

    python tools/run_cmax.py --config_file configs/cmax_hot_plate1.yaml
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
    x = np.rint(pts[:, None, 0] + t * flow[:, None, 0]).reshape(-1)
    y = np.rint(pts[:, None, 1] + t * flow[:, None, 1]).reshape(-1)
    ev = np.stack([x, y, 10.0 + 0.0083 * t.reshape(-1), rs.randint(0, 2, x.size)], 1)
    ev = ev[(ev[:, 0] >= 0) & (ev[:, 0] < h) & (ev[:, 1] >= 0) & (ev[:, 1] < w)]
    return ev[np.argsort(ev[:, 2], kind="stable")], (h, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_file", default=os.path.join(ROOT, "configs", "cmax_hot_plate1.yaml"))
    args = ap.parse_args()
    cfg = yaml.safe_load(open(args.config_file))
    cp = cfg.get("common_params", {})
    cfg["solver"].setdefault("filter", {})["parameters"] = {k: cp[k] for k in ("xmin", "xmax", "ymin", "ymax") if k in cp}
    events, shape = synthetic_window(cfg)
    solver = ebos.solver.collections[cfg["solver"]["method"]](shape, shape, calibration_parameter=None,
                                                              solver_config=cfg["solver"], visualize_module=None)
    t0 = time.perf_counter()
    events, period = solver.preprocess(events)
    flow = solver.estimate(events)
    dt = time.perf_counter() - t0
    iwe0 = solver.orig_imager.create_iwe(events, "bilinear_vote", sigma=0)
    warped, _ = solver.orig_warper.warp_event(events, flow, "dense-flow", solver.warp_direction)
    iwe1 = solver.orig_imager.create_iwe(warped, "bilinear_vote", sigma=0)
    print(json.dumps({"events": int(len(events)), "image": list(shape), "time_period_s": period, "solver_s": round(dt, 3),
                      "iterations": len(solver.history), "loss_first": solver.history[0], "loss_last": solver.history[-1],
                      "variance_unwarped": float(iwe0.var(ddof=1)), "variance_warped": float(iwe1.var(ddof=1)),
                      "flow_abs_max": float(np.abs(flow).max())}))


if __name__ == "__main__":
    main()
