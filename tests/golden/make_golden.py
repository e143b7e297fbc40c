#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, imported read-only with
PYTHONDONTWRITEBYTECODE=1).  The fixtures are data: seeded inputs + the reference's
outputs.  They pin oracle/ebos_oracle.py (tests/test_oracle_golden.py), which in turn is
the checker for the HIP path on the GPU box (where /root/reference does not exist).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Reference entry points exercised (paths relative to /root/reference):
  src/warp.py:193-383  src/event_image_converter.py:51-73,288-301,332-620
  src/utils/stat_utils.py:48-139 (SobelTorch)  src/costs/*.py  src/solver/patch_eklt.py:70-95
"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True


class _AnyModule(types.ModuleType):
    """Empty stand-in for an absent third-party package: any attribute is a dummy class."""

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        cls = type(item, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, item, cls)
        return cls


def _stub(name, **attrs):
    m = _AnyModule(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:  # make `from pkg import sub` resolve to the stubbed submodule
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


def import_reference():
    """Import the reference's hot-path modules; third-party packages that are absent from
    the image and unused on this path are stubbed as empty modules."""
    sys.path.insert(0, REF)
    from src.warp import Warp  # noqa
    from src.event_image_converter import EventImageConverter  # noqa

    _stub("cv2")
    spec = importlib.util.spec_from_file_location("ref_stat_utils", f"{REF}/src/utils/stat_utils.py")
    stat = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(stat)

    # src.costs pulls src.utils / src.visualizer -> stub the absent third-party imports.
    for n in ["openpiv", "openpiv.windef", "openpiv.tools", "openpiv.pyprocess", "openpiv.validation",
              "openpiv.filters", "openpiv.scaling", "openpiv.preprocess", "openpiv.smoothn",
              "ffmpeg", "skimage", "skimage.util", "skimage.metrics", "skimage.transform",
              "h5py", "hdf5plugin", "pivpy", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "plotly", "plotly.graph_objects",
              "plotly.express", "plotly.subplots", "optuna.samplers", "optuna.distributions",
              "optuna.storages", "optuna.study", "optuna.trial", "torch_scatter"]:
        _stub(n)
    op = _stub("optuna")
    for sub in ("storages", "samplers", "distributions", "study", "trial"):
        setattr(op, sub, sys.modules["optuna." + sub])
    sys.modules["optuna.storages"].InMemoryStorage = type("InMemoryStorage", (), {})
    sys.modules["optuna.storages"].BaseStorage = type("BaseStorage", (), {})
    sys.modules["optuna.distributions"].BaseDistribution = type("BaseDistribution", (), {})
    op.storages = sys.modules["optuna.storages"]
    op.samplers = sys.modules["optuna.samplers"]
    op.distributions = sys.modules["optuna.distributions"]
    costs = None
    try:
        import src.costs as costs  # noqa
    except Exception as e:  # pragma: no cover - informational
        print("WARNING: src.costs not importable here:", repr(e))
    return Warp, EventImageConverter, stat.SobelTorch, costs


def synth_events(n, h, w, seed=0, tmin=0.0, tmax=0.5):
    # recipe of src/utils/event_utils.py:40-47 with an explicit legacy RandomState
    rs = np.random.RandomState(seed)
    x = rs.randint(0, h, n)
    y = rs.randint(0, w, n)
    t = np.sort(rs.uniform(tmin, tmax, n))
    p = rs.randint(0, 2, n)
    return np.stack([x, y, t, p], axis=1).astype(np.float64)


def synth_flow(h, w, seed=1, fmax=30.0):
    return np.random.RandomState(seed).uniform(-fmax, fmax, (2, h, w))  # src/utils/flow_utils.py:29


def make_patches():
    """golden_patches.npz: the reference's patch grid (src/solver/patch_eklt.py:70-95), FlowPatch bounds
    (src/types/flow_patch.py:33-47) and per-patch ``len(crop_event(...))`` (src/utils/event_utils.py:109-129, the loop of
    src/solver/patch_eklt.py:118-124) on seeded events -- pins types.patch_bounds / utils.crop_event /
    EventPlan.patch_event_counts."""
    import_reference()
    import src.utils as rutils
    from src.solver.patch_eklt import PatchEklt

    out = {}
    cases = [((60, 80), (10, 16), (10, 16)), ((60, 80), (15, 21), (7, 9)), ((50, 70), (16, 16), (8, 8)),
             ((45, 64), (9, 13), (9, 13)), ((33, 47), (8, 8), (5, 3))]
    for k, (size, patch, slide) in enumerate(cases):
        patches, shape = PatchEklt.prepare_patch(None, size, patch, slide)
        ev = synth_events(4000, size[0], size[1], seed=300 + k)
        ev[::7, 0] += 0.5   # fractional (undistorted) coordinates too
        ev[::5, 1] += 0.25
        counts = np.array([len(rutils.crop_event(ev, patches[i].x_min, patches[i].x_max, patches[i].y_min, patches[i].y_max))
                           for i in range(len(patches))]).reshape(shape)
        bounds = np.array([[patches[i].x_min, patches[i].x_max, patches[i].y_min, patches[i].y_max] for i in range(len(patches))])
        tag = f"p{k}"
        out[tag + "_cfg"] = np.array([*size, *patch, *slide])
        out[tag + "_events"] = ev
        out[tag + "_grid_shape"] = np.array(shape)
        out[tag + "_bounds"] = bounds.reshape(shape + (4,))
        out[tag + "_counts"] = counts
        i0 = len(patches) // 2
        out[tag + "_crop_mid"] = rutils.crop_event(ev, patches[i0].x_min, patches[i0].x_max, patches[i0].y_min, patches[i0].y_max)
        out[tag + "_crop_mid_torch"] = rutils.crop_event(torch.from_numpy(ev), patches[i0].x_min, patches[i0].x_max,
                                                         patches[i0].y_min, patches[i0].y_max).numpy()
    np.savez_compressed(os.path.join(HERE, "golden_patches.npz"), **out)
    print("golden_patches.npz:", len(out), "arrays")


def install_torchvision_shim():
    """torchvision is absent from the image (and there is no network to install it): the two torchvision functions the
    reference calls on this path are SHIMMED with torch primitives, so that the reference's OWN code around them (pad
    arithmetic, resize target size, centre crop -- src/solver/patch_eklt.py:173-204; the [None, None] / squeeze handling and
    kernel_size of src/event_image_converter.py:394-405) runs and is pinned.  What stays un-pinned is the inside of the two
    shimmed functions themselves; the fixtures carry ``shimmed = 1`` to say so.

      resize(img, size, interpolation=BILINEAR)  ->  F.interpolate(img[None], size, mode="bilinear", align_corners=False)[0]
          (what torchvision's tensor path calls; its antialias flag only acts when DOWN-sampling, the path up-samples)
      gaussian_blur(img, kernel_size=3, sigma)   ->  taps exp(-x^2 / (2 sigma^2)) at x = -1, 0, 1, normalised, outer product,
          reflect padding 1, depth-wise conv2d (torchvision's published _get_gaussian_kernel1d / gaussian_blur)"""
    import torch.nn.functional as F

    def resize(img, size, interpolation=None, max_size=None, antialias=None):
        assert img.dim() == 3 and (interpolation is None or str(interpolation).lower().endswith("bilinear"))
        return F.interpolate(img[None], size=list(size), mode="bilinear", align_corners=False)[0]

    def gaussian_blur(img, kernel_size, sigma=None):
        ks = [kernel_size, kernel_size] if isinstance(kernel_size, int) else list(kernel_size)
        sg = [float(sigma), float(sigma)] if not isinstance(sigma, (list, tuple)) else [float(v) for v in sigma]
        ker = []
        for k, sd in zip(ks, sg):
            half = (k - 1) * 0.5
            x = torch.linspace(-half, half, steps=k, dtype=img.dtype)
            pdf = torch.exp(-0.5 * (x / sd) ** 2)
            ker.append(pdf / pdf.sum())
        k2 = ker[1][:, None] * ker[0][None, :]   # (y taps) x (x taps)
        lead = img.shape[:-2]
        x = img.reshape(-1, 1, *img.shape[-2:])
        x = F.pad(x, (ks[0] // 2, ks[0] // 2, ks[1] // 2, ks[1] // 2), mode="reflect")
        return F.conv2d(x, k2[None, None]).reshape(*lead, *img.shape[-2:])

    class InterpolationMode(object):
        BILINEAR = "bilinear"

    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    fn.resize, fn.gaussian_blur, fn.InterpolationMode = resize, gaussian_blur, InterpolationMode
    tr.functional, tr.InterpolationMode = fn, InterpolationMode
    tv.transforms = tr
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn})
    return fn


def make_upsample():
    """golden_upsample.npz (SHIMMED, see install_torchvision_shim): the reference's
    ``PatchEklt.interpolate_dense_flow_from_patch_tensor`` (src/solver/patch_eklt.py:173-204) on seeded patch grids --
    BASELINE's (30, 40) -> 720x1280, an overlapping-window case, non-divisible sizes, a 1x1 grid -- and
    ``EventImageConverter.create_image_from_events_tensor(..., sigma in {1, 3})`` (src/event_image_converter.py:372-405),
    un-batched and batched."""
    fn = install_torchvision_shim()
    import_reference()
    for n in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):  # import_reference re-stubs them
        pass
    install_torchvision_shim()
    import src.event_image_converter as ref_eic
    from src.solver import patch_eklt as ref_pe

    ref_pe.transforms = sys.modules["torchvision.transforms"]          # the names the reference module bound at import
    ref_eic.gaussian_blur = fn.gaussian_blur                           # (its own import was swallowed: ImportError, :12-17)
    out = {"shimmed": np.array(1)}
    cases = [((720, 1280), (24, 32), (24, 32)),    # BASELINE configs[3]: grid (30, 40)
             ((128, 160), (32, 32), (16, 16)),     # overlapping windows: pad 2
             ((100, 150), (24, 32), (24, 32)),     # sizes that are no multiple of the sliding window
             ((90, 121), (30, 40), (15, 27)),      # everything odd
             ((64, 64), (64, 64), (64, 64)),       # a 1 x 1 grid
             ((60, 80), (10, 16), (5, 4))]         # patch / 2 // slide > 1: pad 2 x 3
    for k, (size, patch, slide) in enumerate(cases):
        _, shape = ref_pe.PatchEklt.prepare_patch(None, size, patch, slide)
        ns = types.SimpleNamespace(patch_size=patch, sliding_window=slide, patch_image_size=tuple(shape), n_pixel_downsample=1,
                                   orig_image_shape=size)
        grid = np.random.RandomState(500 + k).uniform(-30, 30, (2,) + tuple(shape))
        dense = ref_pe.PatchEklt.interpolate_dense_flow_from_patch_tensor(ns, torch.from_numpy(grid.reshape(-1))).numpy()
        tag = f"u{k}"
        out[tag + "_cfg"] = np.array([*size, *patch, *slide])
        out[tag + "_grid"] = grid
        out[tag + "_shape"] = np.array(dense.shape)
        if dense.size <= 100_000:
            out[tag + "_dense"] = dense
        else:   # the 1280x720 field is 14.7 MB in fp64: keep strided samples, border lines and sums (a fixture is small data)
            out[tag + "_dense_stride"] = dense[:, ::7, ::11].copy()
            out[tag + "_dense_rows"] = dense[:, [0, 1, 23, 24, 359, 695, 696, 718, 719], :].copy()
            out[tag + "_dense_cols"] = dense[:, :, [0, 1, 31, 32, 640, 1247, 1248, 1278, 1279]].copy()
            out[tag + "_dense_rowsum"] = dense.sum(2)
            out[tag + "_dense_colsum"] = dense.sum(1)
        print(tag, size, patch, slide, "grid", shape, "->", dense.shape)
    # tensor blur: un-batched and batched, sigma 1 and 3 (kernel_size stays 3: src/event_image_converter.py:404)
    H, W, N = 24, 32, 2000
    e2 = synth_events(N, H, W, seed=10)
    e2[:, 0] += np.random.RandomState(11).uniform(0, 0.999, N) * (np.arange(N) % 3 == 0)
    eb = np.stack([e2, synth_events(N, H, W, seed=20)])
    out["b_events"], out["b_events_batched"] = e2, eb
    for pad in (0, 2):
        ic = ref_eic.EventImageConverter((H, W), outer_padding=pad)
        for sigma in (1, 3):
            out[f"b_p{pad}_s{sigma}"] = ic.create_image_from_events_tensor(torch.from_numpy(e2), "bilinear_vote", sigma=sigma).numpy()
            out[f"b_p{pad}_s{sigma}_batched"] = ic.create_image_from_events_tensor(torch.from_numpy(eb), "bilinear_vote",
                                                                                  sigma=sigma).numpy()
            out[f"b_p{pad}_s{sigma}_f32"] = ic.create_image_from_events_tensor(torch.from_numpy(e2).float(), "bilinear_vote",
                                                                              sigma=sigma).numpy()
    out["b_create_iwe_default"] = ref_eic.EventImageConverter((H, W)).create_iwe(torch.from_numpy(e2)).numpy()  # sigma = 1 (:55)
    np.savez_compressed(os.path.join(HERE, "golden_upsample.npz"), **out)
    print("golden_upsample.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "golden_upsample.npz")) // 1024, "KiB")


def public_surface(cls):
    """{method name: str(inspect.signature)} of the public callables a class defines or inherits (object's own excluded),
    plus ``__init__``."""
    import inspect

    out = {}
    for name, member in inspect.getmembers(cls):
        if name != "__init__" and name.startswith("_"):
            continue
        if callable(member) and not isinstance(member, type):
            try:
                sig = inspect.signature(member)
            except (TypeError, ValueError):
                continue
            out[name] = {"signature": str(sig),
                         "params": [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
                                    for p in sig.parameters.values()]}
    return out


def make_signatures():
    """signatures.json: the plugin surface of the reference as data -- ``inspect.signature`` of every public method of
    ``Warp``, ``EventImageConverter``, ``CostBase``, ``HybridCost``, every registered cost and ``SolverBase``, plus the
    registries' keys and class attributes (``name``, ``required_keys``).  tests/test_surface.py compares the build's classes
    with it (and checks that the solver composed over a base with exactly these methods exposes all of them)."""
    import json

    Warp, EIC, _, costs = import_reference()
    import src.solver as rsolver

    sig = {"Warp": public_surface(Warp), "EventImageConverter": public_surface(EIC), "CostBase": public_surface(costs.CostBase),
           "HybridCost": public_surface(costs.HybridCost), "SolverBase": public_surface(rsolver.SolverBase),
           "costs.functions": {k: {"methods": public_surface(v), "name": v.name, "required_keys": list(v.required_keys)}
                               for k, v in sorted(costs.functions.items())},
           "solver.collections": sorted(rsolver.collections),
           "driver_calls_on_solver": ["preprocess", "estimate", "visualize_original_sequential", "visualize_flows",
                                      "visualize_pred_sequential", "visualize_gt_sequential", "calculate_flow_error",
                                      "save_flow_error_as_text"],   # bos_event.py:190-219
           "driver_reads_on_solver": ["sequential_video_list", "evaluation_text_list", "visualizer", "orig_imager"]}
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(sig, f, indent=1, sort_keys=True)
    print("signatures.json:", {k: len(v) for k, v in sig.items()})


def make_config():
    """config_hot_plate1.json: the reference's configs/hot_plate1.yaml as PARSED data (``input``) and what the reference's
    own ``propagate_config`` (src/utils/config_utils.py:42-88) makes of it (``propagated``) -- BASELINE configs[0] declares this
    file as its input; pins ``event_based_bos_amd.utils.propagate_config`` and feeds tools/run_cmax.py on the GPU box."""
    import copy
    import json

    import yaml

    import_reference()
    spec = importlib.util.spec_from_file_location("ref_config_utils", f"{REF}/src/utils/config_utils.py")
    sys.modules.setdefault("src.utils.misc", types.ModuleType("src.utils.misc")).fetch_runtime_information = lambda: {}
    cu = importlib.util.module_from_spec(spec)
    cu.__package__ = "src.utils"
    spec.loader.exec_module(cu)
    with open(f"{REF}/configs/hot_plate1.yaml") as f:
        cfg = yaml.safe_load(f)
    out = {"source": "configs/hot_plate1.yaml of the reference, parsed with yaml.safe_load", "input": copy.deepcopy(cfg)}
    cu.propagate_config(cfg)
    out["propagated"] = cfg
    with open(os.path.join(HERE, "config_hot_plate1.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("config_hot_plate1.json: solver keys", sorted(cfg["solver"]))


def main():
    if "--patches" in sys.argv:   # the other fixtures stay byte-identical
        return make_patches()
    if "--config" in sys.argv:
        return make_config()
    if "--signatures" in sys.argv:
        return make_signatures()
    if "--upsample" in sys.argv:
        return make_upsample()
    Warp, EIC, SobelTorch, costs = import_reference()
    out = {}

    # ---- G1: micro known-answer vectors ------------------------------------------------
    ev = np.array([[1, 2], [3, 4], [3.5, 4.5], [-0.5, 0.5], [0.9999995, 1.0], [10, 10]], dtype=np.float64)
    ev = np.concatenate([ev, np.zeros((6, 2))], axis=1)
    im = EIC((4, 5))
    out["g1_events"] = ev
    out["g1_vote_torch"] = im.bilinear_vote_tensor(torch.from_numpy(ev)).numpy()
    out["g1_vote_numpy"] = im.bilinear_vote_numpy(ev)
    out["g1_count_numpy"] = im.count_event_numpy(ev)
    wts = np.arange(6, dtype=np.float64)
    out["g1_vote_weighted_torch"] = im.bilinear_vote_tensor(torch.from_numpy(ev), weight=torch.from_numpy(wts)).numpy()
    out["g1_vote_weighted_numpy"] = im.bilinear_vote_numpy(ev, weight=wts)
    im2 = EIC((4, 5), outer_padding=2)
    out["g1_vote_pad2_torch"] = im2.bilinear_vote_tensor(torch.from_numpy(ev)).numpy()
    out["g1_vote_pad2_numpy"] = im2.bilinear_vote_numpy(ev)
    out["g1_mask_torch"] = im.create_eventmask(torch.from_numpy(ev)).numpy()

    wv = np.array([[1, 1, 0, 1], [2, 3, .5, 0], [3, 4, 1, 1]], dtype=np.float64)
    fl = np.ones((2, 4, 5)) * np.array([1.0, 2.0])[:, None, None]
    wp = Warp((4, 5))
    out["g1_warp_events"] = wv
    out["g1_warp_flow"] = fl
    for d in ["first", "middle", "last", "before", "after"]:
        out[f"g1_warp_dense_{d}_numpy"] = wp.warp_event(wv, fl, "dense-flow", d)[0]
        out[f"g1_warp_dense_{d}_torch"] = wp.warp_event(torch.from_numpy(wv), torch.from_numpy(fl), "dense-flow", d)[0].numpy()
    out["g1_warp_dense_f025_numpy"] = wp.warp_event(wv, fl, "dense-flow", 0.25)[0]
    out["g1_warp_2dof_first_numpy"] = wp.warp_event(wv, np.array([1.0, 2.0]), "2d-translation", "first")[0]
    out["g1_warp_2dof_middle_torch"] = wp.warp_event(torch.from_numpy(wv), torch.tensor([1.0, 2.0], dtype=torch.float64),
                                                     "rigid-optical-flow", "middle")[0].numpy()
    wfrac = np.array([[1.9, 2.9, 0.0, 1], [0.2, 0.7, 1.0, 0]], dtype=np.float64)
    flr = synth_flow(4, 5, seed=3, fmax=2.0)
    out["g1_warp_frac_events"] = wfrac
    out["g1_warp_frac_flow"] = flr
    out["g1_warp_frac_torch"] = wp.warp_event(torch.from_numpy(wfrac), torch.from_numpy(flr), "dense-flow", "first")[0].numpy()
    out["g1_warp_frac_numpy"] = wp.warp_event(wfrac, flr, "dense-flow", "first")[0]
    wpn = Warp((4, 5), normalize_t=True)
    wv_s = wv.copy()
    wv_s[:, 2] *= 0.02
    out["g1_warp_norm_scaled_numpy"] = wpn.warp_event(wv_s, fl, "dense-flow", "first")[0]
    out["g1_flow_from_motion"] = wp.get_flow_from_motion(np.array([1.0, 2.0]), "2d-translation")

    # ---- G2: small full-array cases (24x32, 2000 events) ---------------------------------
    H, W, N = 24, 32, 2000
    e2 = synth_events(N, H, W, seed=10)
    e2[:, 0] += np.random.RandomState(11).uniform(0, 0.999, N) * (np.arange(N) % 3 == 0)  # some fractional sources
    e2[:, 1] += np.random.RandomState(12).uniform(0, 0.999, N) * (np.arange(N) % 5 == 0)
    f2 = synth_flow(H, W, seed=13, fmax=5.0)
    out["g2_events"] = e2
    out["g2_flow"] = f2
    te, tf = torch.from_numpy(e2), torch.from_numpy(f2)
    for norm in (False, True):
        wq = Warp((H, W), normalize_t=norm)
        for d in ["first", "middle", "last", 0.25, "before", "after"]:
            tag = f"g2_warp_dense_n{int(norm)}_{d}"
            out[tag + "_numpy"] = wq.warp_event(e2, f2, "dense-flow", d)[0]
            out[tag + "_torch"] = wq.warp_event(te, tf, "dense-flow", d)[0].numpy()
        out[f"g2_warp_2dof_n{int(norm)}_middle_numpy"] = wq.warp_event(e2, np.array([3.0, -2.0]), "2d-translation", "middle")[0]
        out[f"g2_warp_2dof_n{int(norm)}_first_torch"] = wq.warp_event(te, torch.tensor([3.0, -2.0], dtype=torch.float64),
                                                                      "2d-translation", "first")[0].numpy()
    # float32 torch path (the dtype the HIP kernels run in)
    wq = Warp((H, W), normalize_t=True)
    out["g2_warp_dense_n1_first_torch_f32"] = wq.warp_event(te.float(), tf.float(), "dense-flow", "first")[0].numpy()
    # batched
    eb = np.stack([e2, synth_events(N, H, W, seed=20)], 0)
    fb = np.stack([f2, synth_flow(H, W, seed=21, fmax=5.0)], 0)
    out["g2_events_b"] = eb
    out["g2_flow_b"] = fb
    out["g2_warp_dense_b_n1_middle_numpy"] = wq.warp_event(eb, fb, "dense-flow", "middle")[0]
    out["g2_warp_dense_b_n1_middle_torch"] = wq.warp_event(torch.from_numpy(eb), torch.from_numpy(fb), "dense-flow", "middle")[0].numpy()

    warped = wq.warp_event(e2, f2, "dense-flow", "first")[0]
    tw = torch.from_numpy(warped)
    wgt = np.random.RandomState(14).uniform(0.0, 2.0, N)
    for pad in (0, 2):
        ic = EIC((H, W), outer_padding=pad)
        out[f"g2_iwe_p{pad}_numpy"] = ic.bilinear_vote_numpy(warped)
        out[f"g2_iwe_p{pad}_torch"] = ic.bilinear_vote_tensor(tw).numpy()
        out[f"g2_iwe_p{pad}_w_numpy"] = ic.bilinear_vote_numpy(warped, weight=wgt)
        out[f"g2_iwe_p{pad}_w_torch"] = ic.bilinear_vote_tensor(tw, weight=torch.from_numpy(wgt)).numpy()
        out[f"g2_iwe_p{pad}_w05_torch"] = ic.bilinear_vote_tensor(tw, weight=0.5).numpy()
        out[f"g2_count_p{pad}_numpy"] = ic.count_event_numpy(warped)
        out[f"g2_polarity_p{pad}_numpy"] = ic.create_iwe(warped, method="polarity", sigma=0)
        out[f"g2_mask_p{pad}_numpy"] = ic.create_eventmask(warped)
    ic = EIC((H, W))
    out["g2_weight"] = wgt
    out["g2_iwe_f32_torch"] = ic.bilinear_vote_tensor(tw.float()).numpy()
    for s in (1, 3):
        out[f"g2_iwe_sigma{s}_numpy"] = ic.create_iwe(warped, method="bilinear_vote", sigma=s)
    out["g2_iwe_default_numpy"] = ic.create_iwe(warped)  # default sigma = 1 (numpy)
    out["g2_iwe_default_torch"] = ic.create_iwe(tw, sigma=0).numpy()
    wb = wq.warp_event(eb, fb, "dense-flow", "middle")[0]
    out["g2_iwe_b_numpy"] = ic.bilinear_vote_numpy(wb)
    out["g2_iwe_b_torch"] = ic.bilinear_vote_tensor(torch.from_numpy(wb)).numpy()

    # derived images (A12): weighted / averaged splats, numpy (sigma 0 and 1) and torch (sigma 0) branches
    val = np.random.RandomState(16).uniform(0.5, 1.5, N)
    for name, fn in (("iwa", ic.create_iwa), ("iwd", ic.create_iwd), ("iwt", ic.create_iwt),
                     ("timeimage", ic.create_timeimage), ("prob", ic.create_probability_iwe)):
        out[f"g2_{name}_s0_numpy"] = fn(warped, val, sigma=0)
        out[f"g2_{name}_s1_numpy"] = fn(warped, val, sigma=1)
        out[f"g2_{name}_s0_torch"] = fn(tw, torch.from_numpy(val), sigma=0).numpy()
    out["g2_derived_values"] = val

    # contrast costs + autograd gradients (fp64), built from the reference's primitives
    sob = SobelTorch(ksize=3, in_channels=1, precision="64")

    def contrast(events_t, motion_t, model, cost, omit, imager, warper, weight=1.0):
        w_, _ = warper.warp_event(events_t, motion_t, model, "first")
        iwe = imager.bilinear_vote_tensor(w_, weight=weight)
        if cost == "var":
            x = iwe[1:-1, 1:-1] if omit else iwe
            return -torch.var(x), iwe
        g = sob(iwe[None, None]) / 8.0
        m = g[0, 0] ** 2 + g[0, 1] ** 2
        m = m[1:-1, 1:-1] if omit else m
        return -torch.mean(m), iwe

    out["g2_sobel"] = sob(torch.from_numpy(out["g2_iwe_p0_torch"])[None, None])[0].detach().numpy()
    for cost in ("var", "gm"):
        for omit in (False, True):
            fl_t = tf.clone().requires_grad_(True)
            wt_t = torch.from_numpy(wgt).clone().requires_grad_(True)
            L, iwe = contrast(te, fl_t, "dense-flow", cost, omit, ic, wq, weight=wt_t)
            L.backward()
            tag = f"g2_{cost}_omit{int(omit)}"
            out[tag + "_loss"] = np.float64(L.item())
            out[tag + "_dflow"] = fl_t.grad.numpy()
            out[tag + "_dweight"] = wt_t.grad.numpy()
            th = torch.tensor([3.0, -2.0], dtype=torch.float64, requires_grad=True)
            L2, _ = contrast(te, th, "2d-translation", cost, omit, ic, wq)
            L2.backward()
            out[tag + "_2dof_loss"] = np.float64(L2.item())
            out[tag + "_2dof_dtheta"] = th.grad.numpy()
    # d(loss)/d(iwe) for the two costs (used to pin the cost kernels in isolation)
    for cost in ("var", "gm"):
        for omit in (False, True):
            x = torch.from_numpy(out["g2_iwe_p0_torch"]).clone().requires_grad_(True)
            if cost == "var":
                L = -torch.var(x[1:-1, 1:-1] if omit else x)
            else:
                g = sob(x[None, None]) / 8.0
                m = g[0, 0] ** 2 + g[0, 1] ** 2
                L = -torch.mean(m[1:-1, 1:-1] if omit else m)
            L.backward()
            out[f"g2_{cost}_omit{int(omit)}_diwe"] = x.grad.numpy()

    # shipped costs (A15) + hybrid through the reference's registry, if importable here
    if costs is not None:
        out["g2_costs_registry"] = np.array(sorted(costs.functions.keys()))
        fn = costs.functions["flow_norm"](direction="minimize")
        out["g2_cost_flow_norm"] = np.float64(fn.calculate({"flow": tf}).item())
        out["g2_cost_flow_norm_numpy"] = np.float64(fn.calculate({"flow": f2}))
        ig = costs.functions["image_gradient"](direction="minimize")
        wmap = torch.from_numpy(np.random.RandomState(15).uniform(0.5, 1.5, (H, W)))
        out["g2_cost_weights"] = wmap.numpy()
        out["g2_cost_image_gradient"] = np.float64(
            ig.calculate({"flow": tf, "omit_boundary": False, "weights": wmap}).item())
        dn = costs.functions["diff_norm"](direction="minimize")
        pred = torch.from_numpy(out["g2_iwe_p0_torch"])
        meas = torch.from_numpy(out["g2_iwe_p0_w_torch"])
        out["g2_cost_diff_norm"] = np.float64(
            dn.calculate({"prediction": pred, "measurement": meas, "weights": None}).item())
        hy = costs.HybridCost("minimize", {"flow_norm": 0.5, "image_gradient": "inv"}, store_history=True)
        v = hy.calculate({"flow": tf, "omit_boundary": False, "weights": wmap})
        out["g2_cost_hybrid"] = np.float64(v.item())
        out["g2_cost_hybrid_history_keys"] = np.array(sorted(hy.get_history().keys()))

    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **out)
    print("golden_small.npz:", len(out), "arrays")

    # ---- G3: mid-size summaries ---------------------------------
    mid = {}
    for (h, w, n, fmax) in [(260, 346, 100_000, 5.0), (720, 1280, 1_000_000, 30.0)]:
        tag = f"g3_{h}x{w}_{n}"
        e = synth_events(n, h, w, seed=0)
        f = synth_flow(h, w, seed=1, fmax=fmax)
        te, tf = torch.from_numpy(e), torch.from_numpy(f)
        wq, ic = Warp((h, w), normalize_t=True), EIC((h, w))
        stats = {}
        for cost in ("var", "gm"):
            fl_t = tf.clone().requires_grad_(True)
            L, iwe = contrast(te, fl_t, "dense-flow", cost, False, ic, wq)
            L.backward()
            stats[cost] = (L.item(), fl_t.grad)
        iwe = iwe.detach()
        mid[tag + "_iwe_sum"] = np.float64(iwe.sum().item())
        mid[tag + "_iwe_max"] = np.float64(iwe.max().item())
        mid[tag + "_iwe_l2"] = np.float64(torch.linalg.norm(iwe).item())
        mid[tag + "_iwe_center"] = np.float64(iwe[h // 2, w // 2].item())
        mid[tag + "_iwe_stride"] = iwe[::13, ::17].numpy().copy()
        mid[tag + "_iwe_rowsum"] = iwe.sum(1).numpy()
        mid[tag + "_iwe_colsum"] = iwe.sum(0).numpy()
        for cost in ("var", "gm"):
            L, g = stats[cost]
            mid[tag + f"_{cost}_loss"] = np.float64(L)
            mid[tag + f"_{cost}_dflow_l2"] = np.float64(torch.linalg.norm(g).item())
            mid[tag + f"_{cost}_dflow_stride"] = g[:, ::13, ::17].numpy().copy()
        th = torch.tensor([3.0, -2.0], dtype=torch.float64, requires_grad=True)
        L2, iwe2 = contrast(te, th, "2d-translation", "var", False, ic, wq)
        L2.backward()
        mid[tag + "_2dof_iwe_sum"] = np.float64(iwe2.sum().item())
        mid[tag + "_2dof_var_loss"] = np.float64(L2.item())
        mid[tag + "_2dof_var_dtheta"] = th.grad.numpy()
        pol = ic.create_iwe(e, method="polarity", sigma=0)
        mid[tag + "_polarity_sums"] = np.array([pol[0].sum(), pol[1].sum()])
        # float32 reference run (what the reference computes when fed fp32 tensors)
        w32, _ = wq.warp_event(te.float(), tf.float(), "dense-flow", "first")
        i32 = ic.bilinear_vote_tensor(w32)
        mid[tag + "_iwe_f32_rel_l2_vs_f64"] = np.float64(
            (torch.linalg.norm(i32.double() - iwe) / torch.linalg.norm(iwe)).item())
        print(tag, {k: (float(v) if np.ndim(v) == 0 else np.shape(v)) for k, v in mid.items() if k.startswith(tag)})
    np.savez_compressed(os.path.join(HERE, "golden_mid.npz"), **mid)
    print("golden_mid.npz:", len(mid), "arrays")


if __name__ == "__main__":
    main()
