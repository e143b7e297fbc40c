import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import event_based_bos_amd as ebos
from bench import H, W, synth_window
n = 50_000_000
ev, _ = synth_window(n, 0)
plan = ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto", emit="compact") if False else ebos.EventPlan.build(torch.from_numpy(ev).cuda(), (H, W), "first", True, tile="auto")
rs = np.random.RandomState(5)
th = torch.from_numpy(rs.uniform(-30, 30, (512, 2))).float().cuda()
for chunk, ns in [(16, 3), (32, 3), (64, 3), (16, 2), (32, 2), (8, 3), (16, 4), (32, 4), (128, 2), (64, 2), (171, 3), (256, 2)]:
    for _ in range(2): plan.variance_2dof(th, chunk=chunk, halo="auto", n_streams=ns)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): v = plan.variance_2dof(th, chunk=chunk, halo="auto", n_streams=ns)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"chunk {chunk:3d} streams {ns}: {dt*1e3:.2f} ms", flush=True)
