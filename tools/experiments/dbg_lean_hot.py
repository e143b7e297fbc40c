import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
import event_based_bos_amd as ebos
h, w, n = 720, 1280, 1_000_000
rs = np.random.RandomState(3)
r, c = rs.randint(0, h, n).astype(np.float64), rs.randint(0, w, n).astype(np.float64)
r[:3000], c[:3000] = 100, 200
r[3000:15000], c[3000:15000] = 300, 700
ev = np.stack([r, c, rs.uniform(0, 0.5, n), rs.randint(0, 2, n)], 1)
ev = torch.from_numpy(ev[np.argsort(ev[:, 2], kind="stable")]).cuda()
plans = [ebos.EventPlan.build(ev, (h, w), "first", True, tile="auto", emit="compact") for _ in range(3)]
used = 4 * int(plans[0].grp_offsets[-1])
a, b = plans[0].cdt[:used].view(torch.int32).cpu().numpy(), plans[1].cdt[:used].view(torch.int32).cpu().numpy()
d = np.nonzero(a != b)[0]
print("tile", plans[0].tile, "differ", len(d), d[:10], d[-10:] if len(d) else None)
th, tw = plans[0].tile
ko, grp = plans[0].key_offsets.cpu().numpy(), plans[0].grp_offsets.cpu().numpy().astype(np.int64)
if len(d):
    t = np.searchsorted(4 * grp, d[0], side="right") - 1
    off = d[0] - 4 * grp[t]
    offs = ko[t * th * tw:(t + 1) * th * tw + 1] - ko[t * th * tw]
    px = np.searchsorted(offs, off, side="right") - 1
    print("tile", t, "pixel", px, divmod(px, tw), "run", offs[px], offs[px + 1], "tile events", offs[-1])
    t2 = np.searchsorted(4 * grp, d[-1], side="right") - 1
    off2 = d[-1] - 4 * grp[t2]
    px2 = np.searchsorted(offs, off2, side="right") - 1 if t2 == t else -1
    print("last: tile", t2, "pixel", px2)
    seg = plans[0].cdt[4*grp[t]+offs[px]:4*grp[t]+offs[px+1]].cpu().numpy()
    print("sorted?", np.all(np.diff(seg) >= 0), len(seg))
    seg = plans[1].cdt[4*grp[t]+offs[px]:4*grp[t]+offs[px+1]].cpu().numpy()
    print("sorted?", np.all(np.diff(seg) >= 0), len(seg))
