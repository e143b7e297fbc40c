#!/usr/bin/env python3
"""One solver iteration with outer_padding, four launches vs the resident launch (2 M events at 1280 x 720; 100 k at 346 x 260)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import event_based_bos_amd as ebos
from event_based_bos_amd.solver.fused_loop import Fused2dofLoop, FusedPatchLoop

out = []
for (h, w, n) in ((720, 1280, 2_000_000), (260, 346, 100_000)):
    rs = np.random.RandomState(0)
    ev = np.stack([rs.randint(0, h, n), rs.randint(0, w, n), np.sort(rs.uniform(0, 0.5, n)), rs.randint(0, 2, n)], 1).astype(np.float64)
    plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (h, w), "first", True, tile="auto", emit="compact")
    gh, gw = ebos.solver.patch_grid_shape((h, w), (24, 32), (24, 32))
    for pad in (0, 2):
        for name, make in (("patch", lambda: FusedPatchLoop(plan, (24, 32), (24, 32), torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, False, pad, "auto", lr=0.05, capacity=300)),
                           ("patch_blur1", lambda: FusedPatchLoop(plan, (24, 32), (24, 32), torch.zeros((2, gh, gw)), 1.0, 0.001, 0.0, False, pad, "auto", lr=0.05, capacity=300, blur_sigma=1.0)),
                           ("2dof_blur3", lambda: Fused2dofLoop(plan, torch.tensor([1.0, -0.5]), 1.0, False, pad, "auto", lr=0.01, capacity=300, blur_sigma=3.0))):
            row = {"image": [h, w], "events": n, "pad": pad, "loop": name}
            for mode, res in (("four_launches", False), ("resident", True)):
                loop = make()
                loop.run(20, resident=res)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop.run(200, resident=res)
                torch.cuda.synchronize()
                row[mode] = {"us_per_iteration": round((time.perf_counter() - t0) / 200 * 1e6, 1), "ran_as": loop.last_run_mode}
            out.append(row)
            print(json.dumps(row), flush=True)
