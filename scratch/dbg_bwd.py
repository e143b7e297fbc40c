import sys, numpy as np, torch
sys.path.insert(0, ".")
import event_based_bos_amd as ebos
from oracle import ebos_oracle as O
def rel(a,b): return float(np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30))
G=lambda a,dt=None: torch.as_tensor(a).to("cuda", dtype=dt) if dt else torch.as_tensor(a).cuda()
h,w,n=130,173,60000
ev=O.synth_events(n,h,w,seed=3); fl=O.synth_dense_flow(h,w,seed=4,max_val=12.0)
f=torch.from_numpy(fl).clone().requires_grad_(True)
iwe=O.iwe_dense(torch.from_numpy(ev), f, (h,w)); L=O.image_variance(iwe); L.backward(); ref=f.grad.numpy()
for tile,halo in [((64,64),32),((32,64),32),((32,32),8),((64,64),None)]:
    plan=ebos.EventPlan.build(G(ev),(h,w),"first",True,tile=tile)
    flow=G(fl,torch.float32).requires_grad_(True)
    iw=plan.iwe_dense(flow,halo=halo); loss=-ebos.ops.image_variance(iw); loss.backward()
    g=flow.grad.cpu().numpy()
    d=np.abs(g-ref); k=np.unravel_index(d.argmax(), d.shape)
    print(tile,halo,"rel",rel(g,ref),"max abs err",d.max(),"at",k,"ref",ref[k],"got",g[k], "dt_bound", plan.dt_bound, flush=True)
    f2=G(fl,torch.float32).requires_grad_(True)
    l2=-plan.contrast_dense(f2,"image_variance",halo=halo); l2.backward()
    print("   contrast_dense rel", rel(f2.grad.cpu().numpy(), ref))
plan=ebos.EventPlan.build(G(ev),(h,w),"first",True,tile=(64,64))
flow=G(fl,torch.float32).requires_grad_(True)
iw=plan.iwe_dense(flow,halo=32); loss=-ebos.ops.image_variance(iw); loss.backward()
g=flow.grad.cpu().numpy(); d=np.abs(g-ref)
bad=np.argwhere(d>1e-5)
print("n bad", len(bad)); print(bad[:40].tolist())
print([ (tuple(k), float(g[tuple(k)]), float(ref[tuple(k)])) for k in bad[:12]])
