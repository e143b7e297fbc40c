#!/usr/bin/env python3
"""Host-side profile (cProfile) and warm timings of ``ContrastMaximization.estimate`` for the two shipped YAMLs (the reference's
configs/hot_plate1.yaml as parsed data at 346x260, and configs/cmax_hot_plate1.yaml): where a window's time goes outside the kernels.

    python tools/profile_estimate.py
"""
import cProfile, pstats, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import event_based_bos_amd as ebos
import run_cmax as R
for cfgf, hw in ((os.path.join(ROOT, "tests/golden/config_hot_plate1.json"), (260, 346)), (os.path.join(ROOT, "configs/cmax_hot_plate1.yaml"), None)):
    cfg = R.load_config(cfgf)
    d, cp = cfg["data"], cfg.setdefault("common_params", {})
    if hw:
        d["height"], d["width"] = hw
        cp.update(xmin=0, xmax=hw[0], ymin=0, ymax=hw[1])
    for k, v in (("xmin", 0), ("xmax", d["height"]), ("ymin", 0), ("ymax", d["width"])):
        cp.setdefault(k, v)
    d.setdefault("n_events", 100_000)
    ebos.utils.propagate_config(cfg)
    scfg = cfg["solver"]
    if scfg.get("method") not in ebos.solver.collections:
        scfg["method"] = "contrast_maximization"
    cww = scfg.get("cost_with_weight") or {}
    if not any(k in ("image_variance", "gradient_magnitude") for k in cww):
        scfg["cost_with_weight"] = {"image_variance": 1.0}
    events, shape = R.synthetic_window(cfg)
    solver = ebos.solver.collections[scfg["method"]](shape, (d["crop_height"], d["crop_width"]), calibration_parameter=None, solver_config=scfg, visualize_module=None)
    events, period = solver.preprocess(events)
    for _ in range(3):
        solver.estimate(events)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); solver.estimate(events); ts.append(time.perf_counter() - t0)
    print(os.path.basename(cfgf), "estimate warm ms:", [round(t * 1e3, 2) for t in ts], "loop_mode", solver.loop_mode, "iters", len(solver.history))
    pr = cProfile.Profile(); pr.enable(); solver.estimate(events); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
