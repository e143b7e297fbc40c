import sys, yaml, copy, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import event_based_bos_amd as ebos
from run_cmax import synthetic_window
base = yaml.safe_load(open("/root/repo/configs/cmax_hot_plate1.yaml"))
for bd in ([3.0, -2.0], [0.0, 0.0]):
  for blur in (0, 1):
    for lr in (0.08, 0.3):
        cfg = copy.deepcopy(base)
        cfg["data"]["base_displacement"] = bd
        cfg["solver"]["iwe"]["blur_sigma"] = blur
        cfg["solver"]["optimizer"]["parameters"]["lr"] = lr
        ev, shape = synthetic_window(cfg)
        s = ebos.solver.collections["cmax"](shape, shape, solver_config=cfg["solver"])
        flow = s.estimate(ev)
        w = ebos.Warp(shape, normalize_t=True); ic = ebos.EventImageConverter(shape)
        evg = torch.from_numpy(ev).cuda()
        i0 = ic.create_iwe(evg, method="bilinear_vote", sigma=0)
        wd, _ = w.warp_event(evg, torch.from_numpy(flow).cuda(), "dense-flow", direction="first")
        i1 = ic.create_iwe(wd, method="bilinear_vote", sigma=0)
        print(bd, blur, lr, "loss", round(s.history[0], 3), "->", round(s.history[-1], 3), "var", round(float(i0.var()), 3), "->", round(float(i1.var()), 3),
              "flow med", np.round(np.median(flow, axis=(1, 2)), 2), "max", np.round(np.abs(flow).max(), 2), flush=True)
