"""CPU oracle for the contrast-maximisation hot path (warp -> IWE -> contrast cost).

TEST INFRASTRUCTURE ONLY.  This module is a CPU restatement (numpy + torch-CPU) of the
algorithm of tub-rip/event_based_bos for the one hot path this repository accelerates.
It is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline``
leg of ``bench.py`` -- and by nothing else.  The product package ``event_based_bos_amd``
never imports it: the product path is HIP-only and fails loudly without its extension.

Parity status: PINNED.  Every function below is checked against golden vectors captured
from the reference itself (imported from /root/reference in the build container by
``tests/golden/make_golden.py``; fixtures committed under ``tests/golden/``) by
``tests/test_oracle_golden.py``.  Two pieces call torchvision, which is absent from the
image: the torch 3-tap blur and the patch->dense upsample.  For them the REFERENCE'S OWN
functions were run with only ``torchvision...resize`` / ``gaussian_blur`` shimmed by torch
primitives (``make_golden.py --upsample`` -> ``golden_upsample.npz``, labelled shimmed):
the arithmetic around those two calls is pinned, their insides are not ("shimmed" parity,
short of full) -- see ``gaussian_blur3_torch`` and ``upsample_patch_flow``.  The raw-column
loader restatement (``events_from_raw_columns``) is pinned by ``golden_loader.npz``: the
reference's loader run with the real h5py (``tests/golden/make_golden_loader.py``).

Conventions (reference src/data_loader/ccs.py:293-296, src/warp.py:334):
  event = (x, y, t, p);  x = ROW (height) coordinate, y = COLUMN (width) coordinate;
  images are [H, W] row-major; flow is [2, H, W], channel 0 = row direction.

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

EPS_TORCH = 1e-6  # src/event_image_converter.py:586  (tensor branch)
EPS_NUMPY = 1e-8  # src/event_image_converter.py:528  (numpy branch)

_DIRECTION_FRACTION = {  # src/warp.py:248-259
    "first": 0.0,
    "middle": 0.5,
    "last": 1.0,
    "before": -1.0,
    "after": 2.0,
}


# --------------------------------------------------------------------------------------
# A1/A2  reference time and dt                                       src/warp.py:230-288
# --------------------------------------------------------------------------------------
def _tmin_tmax(t):
    if isinstance(t, np.ndarray):
        return t.min(axis=-1), t.max(axis=-1)
    return torch.min(t, -1).values, torch.max(t, -1).values


def reference_time(events, direction: Union[str, float] = "first"):
    """Reference time of a warp.  src/warp.py:230-262.

    'first' -> min t and 'last' -> max t are returned as-is; every other mode (a python
    ``float`` f, or middle/before/after/random) is ``tmin + (tmax - tmin) * f``.
    Anything else (ints, numpy scalars, unknown strings) raises ValueError like the
    reference's fall-through (src/warp.py:245,260-262).
    """
    t = events[..., 2]
    tmin, tmax = _tmin_tmax(t)
    if type(direction) is float:
        return tmin + (tmax - tmin) * direction
    if direction == "first":
        return tmin
    if direction == "last":
        return tmax
    if direction == "random":
        return tmin + (tmax - tmin) * float(np.random.uniform(low=0.0, high=1.0))
    if isinstance(direction, str) and direction in _DIRECTION_FRACTION:
        return tmin + (tmax - tmin) * _DIRECTION_FRACTION[direction]
    raise ValueError(f"direction argument should be first, middle, last. Or float. {direction}")


def delta_t(events, ref_time, normalize_t: bool, time_period=None):
    """dt = t - t_ref, optionally divided by the window period.  src/warp.py:264-288."""
    dt = events[..., 2] - ref_time
    if normalize_t:
        if time_period is None:
            lo, hi = _tmin_tmax(dt)
            time_period = hi - lo
        dt = dt / time_period[..., None]
    return dt


# --------------------------------------------------------------------------------------
# A3  dense-flow warp                                                src/warp.py:292-342
# --------------------------------------------------------------------------------------
def warp_dense_numpy(events: np.ndarray, flow: np.ndarray, direction="first",
                     normalize_t: bool = False) -> np.ndarray:
    """numpy branch, src/warp.py:314-329.  events [(b,) n,4], flow [(b,) 2,H,W]."""
    ref = reference_time(events, direction)
    if events.ndim == 3:
        ref = ref[..., None]
    dt = delta_t(events, ref, normalize_t)
    ev, fl, d = events, flow, dt
    if ev.ndim == 2:
        ev, fl, d = ev[None], fl[None], d[None]
    out = ev.copy()
    rows = ev[..., 0].astype(np.int32)  # truncation toward zero, src/warp.py:319-320
    cols = ev[..., 1].astype(np.int32)
    for b in range(ev.shape[0]):
        out[b, :, 0] = ev[b, :, 0] - d[b] * fl[b, 0, rows[b], cols[b]]
        out[b, :, 1] = ev[b, :, 1] - d[b] * fl[b, 1, rows[b], cols[b]]
    out[..., 2] = d
    return out.squeeze()


def warp_dense_torch(events: torch.Tensor, flow: torch.Tensor, direction="first",
                     normalize_t: bool = False, row_stride: Optional[int] = None) -> torch.Tensor:
    """torch branch, src/warp.py:330-342 (same op sequence: clone, gather x2, mul, sub).

    ``row_stride`` is ``Warp.image_size[1]``; defaults to the flow's W.
    """
    ref = reference_time(events, direction)
    if events.dim() == 3:
        ref = ref[..., None]
    dt = delta_t(events, ref, normalize_t)
    ev, fl, d = events, flow, dt
    if ev.dim() == 2:
        ev, fl, d = ev[None], fl[None], d[None]
    stride = fl.shape[-1] if row_stride is None else row_stride
    out = ev.clone()
    flat = fl.reshape(fl.shape[0], 2, -1)
    lin = ev[..., 0].long() * stride + ev[..., 1].long()
    out[..., 0] = ev[..., 0] - d * torch.gather(flat[:, 0], 1, lin)
    out[..., 1] = ev[..., 1] - d * torch.gather(flat[:, 1], 1, lin)
    out[..., 2] = d
    return out.squeeze()


# --------------------------------------------------------------------------------------
# A4  2-DoF translation warp                                         src/warp.py:344-383
# --------------------------------------------------------------------------------------
def warp_2dof(events, theta, direction="first", normalize_t: bool = False, time_period=None):
    """x' = x + dt*theta0, y' = y + dt*theta1 (note the + sign).  Un-batched only."""
    ref = reference_time(events, direction)
    ev = events[None, :] if events.ndim == 1 else events
    dt = delta_t(ev, ref, normalize_t, time_period)
    xs = ev[:, 0] + dt * theta[0]
    ys = ev[:, 1] + dt * theta[1]
    if isinstance(ev, np.ndarray):
        return np.vstack([xs, ys, dt, ev[:, 3]]).T
    return torch.vstack([xs, ys, dt, ev[:, 3]]).T


# --------------------------------------------------------------------------------------
# A7/A8/A10  bilinear splat and count images          src/event_image_converter.py:407-620
# --------------------------------------------------------------------------------------
def padded_size(image_size: Tuple[int, int], outer_padding) -> Tuple[Tuple[int, int], Tuple[int, int]]:
    """src/event_image_converter.py:29-34."""
    if isinstance(outer_padding, (int, float)):
        pad = (int(outer_padding), int(outer_padding))
    else:
        pad = tuple(outer_padding)
    return tuple(int(s + 2 * p) for s, p in zip(image_size, pad)), pad


def _taps_numpy(xy: np.ndarray, h: int, w: int, ph: int, pw: int, eps: float):
    base = np.floor(xy + eps)
    frac = xy - base
    col = base[..., 1] + pw
    row = base[..., 0] + ph
    lin = np.concatenate([col + row * w, col + (row + 1) * w,
                          (col + 1) + row * w, (col + 1) + (row + 1) * w], axis=-1)
    ok_c0 = (0 <= col) & (col < w)
    ok_c1 = (0 <= col + 1) & (col + 1 < w)
    ok_r0 = (0 <= row) & (row < h)
    ok_r1 = (0 <= row + 1) & (row + 1 < h)
    mask = np.concatenate([ok_c0 & ok_r0, ok_c0 & ok_r1, ok_c1 & ok_r0, ok_c1 & ok_r1], axis=-1)
    return frac, lin, mask


def bilinear_vote_numpy(events: np.ndarray, image_size: Tuple[int, int], pad=(0, 0),
                        weight: Union[float, np.ndarray] = 1.0, eps: float = EPS_NUMPY) -> np.ndarray:
    """numpy branch, src/event_image_converter.py:503-560.  ``image_size`` is the PADDED
    size.  float64 image always; masked taps add 0 at pixel 0.  ``np.bincount`` performs
    the same in-order sequential accumulation as the reference's ``np.add.at``."""
    if isinstance(weight, np.ndarray):
        assert weight.shape == events.shape[:-1]
    ev = events[None] if events.ndim == 2 else events
    h, w = image_size
    ph, pw = pad
    frac, lin, mask = _taps_numpy(ev[..., :2], h, w, ph, pw, eps)
    f0, f1 = frac[..., 0], frac[..., 1]
    vals = np.concatenate([(1 - f0) * (1 - f1) * weight, f0 * (1 - f1) * weight,
                           (1 - f0) * f1 * weight, f0 * f1 * weight], axis=-1)
    lin = (lin * mask).astype(np.int64)
    vals = vals * mask
    img = np.zeros((ev.shape[0], h * w), dtype=np.float64)
    for b in range(ev.shape[0]):
        img[b] = np.bincount(lin[b], weights=vals[b], minlength=h * w)
    return img.reshape((ev.shape[0], h, w)).squeeze()


def count_events_numpy(events: np.ndarray, image_size: Tuple[int, int], pad=(0, 0),
                       eps: float = EPS_NUMPY) -> np.ndarray:
    """src/event_image_converter.py:407-453: +1 on every in-bounds neighbour (no weights)."""
    ev = events[None] if events.ndim == 2 else events
    h, w = image_size
    ph, pw = pad
    _, lin, mask = _taps_numpy(ev[..., :2], h, w, ph, pw, eps)
    lin = (lin * mask).astype(np.int64)
    img = np.zeros((ev.shape[0], h * w), dtype=np.float64)
    for b in range(ev.shape[0]):
        img[b] = np.bincount(lin[b], weights=mask[b].astype(np.float64), minlength=h * w)
    return img.reshape((ev.shape[0], h, w)).squeeze()


def bilinear_vote_torch(events: torch.Tensor, image_size: Tuple[int, int], pad=(0, 0),
                        weight: Union[float, torch.Tensor] = 1.0, eps: float = EPS_TORCH) -> torch.Tensor:
    """tensor branch, src/event_image_converter.py:562-620; same temporaries (floor, frac,
    4 index planes, 4 masks, 4 weight planes, cat x3, mask-mul x2, scatter_add_) so that
    timing it is a fair stand-in for the reference's CPU path."""
    if isinstance(weight, torch.Tensor):
        assert weight.shape == events.shape[:-1]
    ev = events[None] if events.dim() == 2 else events
    h, w = image_size
    ph, pw = pad
    nb = ev.shape[0]
    img = ev.new_zeros((nb, h * w))
    base = torch.floor(ev[..., :2] + eps)
    frac = ev[..., :2] - base
    base = base.long()
    col = base[..., 1] + pw
    row = base[..., 0] + ph
    lin = torch.cat([col + row * w, col + (row + 1) * w,
                     (col + 1) + row * w, (col + 1) + (row + 1) * w], dim=-1)
    mask = torch.cat([(0 <= col) * (col < w) * (0 <= row) * (row < h),
                      (0 <= col) * (col < w) * (0 <= row + 1) * (row + 1 < h),
                      (0 <= col + 1) * (col + 1 < w) * (0 <= row) * (row < h),
                      (0 <= col + 1) * (col + 1 < w) * (0 <= row + 1) * (row + 1 < h)], dim=-1)
    f0, f1 = frac[..., 0], frac[..., 1]
    vals = torch.cat([(1 - f0) * (1 - f1) * weight, f0 * (1 - f1) * weight,
                      (1 - f0) * f1 * weight, f0 * f1 * weight], dim=-1)
    lin = (lin * mask).long()
    vals = vals * mask
    img.scatter_add_(1, lin, vals)
    return img.reshape((nb, h, w)).squeeze()


def count_events_torch(events: torch.Tensor, image_size: Tuple[int, int], pad=(0, 0)) -> torch.Tensor:
    """The reference's tensor counter (src/event_image_converter.py:455-501) raises a dtype
    RuntimeError (int64 ``vals`` into a float image); the build implements the
    numpy semantics for tensors, so the oracle does too (eps of the tensor branch)."""
    out = count_events_numpy(events.detach().cpu().double().numpy(), image_size, pad, eps=EPS_TORCH)
    return torch.from_numpy(np.asarray(out)).to(events.dtype)


def polarity_numpy(events: np.ndarray, image_size, pad=(0, 0), weight=1.0) -> np.ndarray:
    """src/event_image_converter.py:355-363: stack([vote(p>0), vote(p<=0)], axis=-3)."""
    pos = events[..., 3] > 0
    if isinstance(weight, np.ndarray):
        a = bilinear_vote_numpy(events[pos], image_size, pad, weight[pos])
        b = bilinear_vote_numpy(events[~pos], image_size, pad, weight[~pos])
    else:
        a = bilinear_vote_numpy(events[pos], image_size, pad, weight)
        b = bilinear_vote_numpy(events[~pos], image_size, pad, weight)
    return np.stack([a, b], axis=-3)


def create_image_numpy(events: np.ndarray, image_size, pad=(0, 0), method="bilinear_vote",
                       weight=1.0, sigma=1) -> np.ndarray:
    """src/event_image_converter.py:332-370 (scipy gaussian over ALL axes when sigma>0)."""
    from scipy.ndimage import gaussian_filter

    if method == "count":
        img = count_events_numpy(events, image_size, pad)
    elif method == "bilinear_vote":
        img = bilinear_vote_numpy(events, image_size, pad, weight)
    elif method == "polarity":
        img = polarity_numpy(events, image_size, pad, weight)
    else:
        raise NotImplementedError(f"{method = } is not supported.")
    if sigma > 0:
        img = gaussian_filter(img, sigma)
    return img


def gaussian_blur3_torch(img: torch.Tensor, sigma: float) -> torch.Tensor:
    """torchvision ``gaussian_blur(img, kernel_size=3, sigma)`` restated.  torchvision is absent from the image, so the
    reference's call site (src/event_image_converter.py:394-405) was run with that one function SHIMMED
    (tests/golden/make_golden.py --upsample -> golden_upsample.npz, tests/test_oracle_golden.py): the surrounding reshapes,
    kernel_size = 3 and the squeeze are pinned, the inside of gaussian_blur is this restatement on both sides ("shimmed",
    not full parity).  Published algorithm (torchvision 0.13
    transforms/functional_tensor.py ``_get_gaussian_kernel1d``/``gaussian_blur``): taps
    ``exp(-0.5 (x/sigma)^2)`` at x = -1,0,1, normalised to sum 1; separable; reflect pad 1.
    Call site: src/event_image_converter.py:399-404.  img [..., H, W]."""
    xs = torch.tensor([-1.0, 0.0, 1.0], dtype=img.dtype)
    k = torch.exp(-0.5 * (xs / sigma) ** 2)
    k = k / k.sum()
    k2 = (k[:, None] * k[None, :])[None, None]
    lead = img.shape[:-2]
    x = img.reshape(-1, 1, *img.shape[-2:])
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    return F.conv2d(x, k2).reshape(*lead, *img.shape[-2:])


def averaged_image(events, values, base, image_size, pad=(0, 0), sigma=1):
    """create_iwa (base 1) / create_iwd (base 0) / create_iwt (base 2), src/event_image_converter.py:75-234:
    (splat weighted by values - base) / (unit splat + 1e-2) + base; numpy branch blurs with scipy's
    gaussian_filter, torch branch returns [1, 1, H, W] (blur needs torchvision: sigma = 0 only here)."""
    if isinstance(events, np.ndarray):
        from scipy.ndimage import gaussian_filter

        num = create_image_numpy(events, image_size, pad, weight=values - base, sigma=0)
        den = create_image_numpy(events, image_size, pad, sigma=0)
        out = np.divide(num, den + 1e-2) + base
        return gaussian_filter(out, sigma) if sigma > 0 else out
    num = bilinear_vote_torch(events, image_size, pad, values - base)
    den = bilinear_vote_torch(events, image_size, pad)
    out = torch.divide(num, den + 1e-2) + base
    out = out[None, None] if out.dim() == 2 else out[:, None]
    if sigma > 0:
        out = gaussian_blur3_torch(out, sigma)
    return out


def weighted_image(events, values, image_size, pad=(0, 0), sigma=1):
    """create_timeimage / create_probability_iwe, src/event_image_converter.py:239-286: a splat weighted per event."""
    if isinstance(events, np.ndarray):
        return create_image_numpy(events, image_size, pad, weight=values, sigma=sigma)
    out = bilinear_vote_torch(events, image_size, pad, values)
    return gaussian_blur3_torch(out, sigma) if sigma > 0 else out


def event_mask(events, image_size, pad=(0, 0)):
    """src/event_image_converter.py:288-301."""
    if isinstance(events, np.ndarray):
        return (0 != bilinear_vote_numpy(events, image_size, pad))[..., None, :, :]
    return (0 != bilinear_vote_torch(events, image_size, pad))[..., None, :, :]


# --------------------------------------------------------------------------------------
# A14  contrast costs (absent from the release; defined by SURVEY.md A14 on the
#      reference's own primitives torch.var and SobelTorch src/utils/stat_utils.py:48-139)
# --------------------------------------------------------------------------------------
_SOBEL_GX = [[-1.0, -2.0, -1.0], [0.0, 0.0, 0.0], [1.0, 2.0, 1.0]]  # row direction
_SOBEL_GY = [[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]  # column direction


def sobel3(img: torch.Tensor) -> torch.Tensor:
    """SobelTorch(ksize=3, in_channels=1) forward (src/utils/stat_utils.py:69-92,136-139):
    cross-correlation with Gx (row derivative) and Gy (column derivative), replicate pad.
    img [H,W] -> [2,H,W] (un-normalised; callers divide by 8)."""
    x = F.pad(img[None, None], (1, 1, 1, 1), mode="replicate")
    k = torch.tensor([_SOBEL_GX, _SOBEL_GY], dtype=img.dtype)[:, None]
    return F.conv2d(x, k)[0]


def _crop(iwe: torch.Tensor, omit_boundary: bool) -> torch.Tensor:
    return iwe[..., 1:-1, 1:-1] if omit_boundary else iwe


def _signed(value, direction: str):
    # sign convention of the shipped costs (src/costs/diff_norm.py:54-57): "minimize"
    # returns the quantity to be minimised, i.e. NEGATIVE contrast; otherwise contrast.
    return -value if direction == "minimize" else value


def image_variance(iwe: torch.Tensor, omit_boundary: bool = False, direction: str = "minimize"):
    """L = -/+ torch.var(iwe) (unbiased); key usage src/solver/base.py:337-339."""
    return _signed(torch.var(_crop(iwe, omit_boundary)), direction)


def gradient_magnitude(iwe: torch.Tensor, omit_boundary: bool = False, direction: str = "minimize"):
    """L = -/+ mean(gx^2 + gy^2) with (gx,gy) = Sobel3(iwe)/8, SURVEY A14."""
    g = sobel3(iwe) / 8.0
    mag = g[0] ** 2 + g[1] ** 2
    return _signed(torch.mean(_crop(mag, omit_boundary)), direction)


# --------------------------------------------------------------------------------------
# A15  shipped image-domain costs (kept as torch ops in the product as well)
# --------------------------------------------------------------------------------------
def flow_norm(flow: torch.Tensor):
    """src/costs/flow_norm.py:45-56."""
    return torch.linalg.norm(flow, dim=0).mean()


def image_gradient_tv(flow: torch.Tensor, weights):
    """src/costs/image_gradient.py:60-75."""
    gx = torch.gradient(flow, dim=1)[0] * weights
    gy = torch.gradient(flow, dim=2)[0] * weights
    return torch.mean(torch.abs(gx) + torch.abs(gy))


def diff_norm(prediction: torch.Tensor, measurement: torch.Tensor):
    """src/costs/diff_norm.py:47-57: matrix 1-norm (max abs column sum) for 2-D input."""
    return torch.linalg.norm(prediction - measurement, ord=1)


# --------------------------------------------------------------------------------------
# A16  patch grid -> dense flow                          src/solver/patch_eklt.py:70-95,173-204
# --------------------------------------------------------------------------------------
def patch_grid_shape(image_size, patch_size, sliding_window) -> Tuple[int, int]:
    """Number of patch centres per axis, src/solver/patch_eklt.py:85-89."""
    nh = len(np.arange(0, image_size[0] - patch_size[0] + sliding_window[0], sliding_window[0]))
    nw = len(np.arange(0, image_size[1] - patch_size[1] + sliding_window[1], sliding_window[1]))
    return nh, nw


def upsample_patch_flow(patch_flow: torch.Tensor, image_size, patch_size, sliding_window) -> torch.Tensor:
    """src/solver/patch_eklt.py:173-204 with torchvision ``resize(bilinear)`` restated as
    ``F.interpolate(mode='bilinear', align_corners=False)`` (what torchvision 0.13 calls for
    tensors; antialias only acts when down-sampling).  Pinned against the reference's own function run with ONLY ``resize``
    shimmed to F.interpolate (torchvision is absent; tests/golden/make_golden.py --upsample -> golden_upsample.npz, six
    geometries incl. overlapping windows and non-divisible sizes, agreement 1e-13): the pad / target-size / centre-crop
    arithmetic is the reference's, the inside of ``resize`` is "shimmed".  Also pinned analytically (constant field,
    linear ramp interior)."""
    gh, gw = patch_flow.shape[-2:]
    pad_h = int(patch_size[0] / 2 // sliding_window[0]) + 1
    pad_w = int(patch_size[1] / 2 // sliding_window[1]) + 1
    x = F.pad(patch_flow.reshape(1, 2, gh, gw), (pad_w, pad_w, pad_h, pad_h), mode="replicate")
    size = [x.shape[2] * sliding_window[0], x.shape[3] * sliding_window[1]]
    dense = F.interpolate(x, size=size, mode="bilinear", align_corners=False)[0]
    ch, cw = dense.shape[1] // 2, dense.shape[2] // 2
    r0 = ch - image_size[0] // 2
    c0 = cw - image_size[1] // 2
    return dense[..., r0:r0 + image_size[0], c0:c0 + image_size[1]]


# --------------------------------------------------------------------------------------
# Composite objective -- used by tests and the cpu_baseline leg of bench.py
# --------------------------------------------------------------------------------------
def iwe_dense(events: torch.Tensor, flow: torch.Tensor, image_size, pad=(0, 0),
              direction="first", normalize_t=True, weight=1.0) -> torch.Tensor:
    warped = warp_dense_torch(events, flow, direction, normalize_t)
    psize = (image_size[0] + 2 * pad[0], image_size[1] + 2 * pad[1])
    return bilinear_vote_torch(warped, psize, pad, weight)


def iwe_2dof(events: torch.Tensor, theta: torch.Tensor, image_size, pad=(0, 0),
             direction="first", normalize_t=True, weight=1.0) -> torch.Tensor:
    warped = warp_2dof(events, theta, direction, normalize_t)
    psize = (image_size[0] + 2 * pad[0], image_size[1] + 2 * pad[1])
    return bilinear_vote_torch(warped, psize, pad, weight)


# --------------------------------------------------------------------------------------
# Synthetic inputs (legacy RandomState: identical streams on every numpy version)
# --------------------------------------------------------------------------------------
def events_from_raw_columns(x, y, t, p, start_index: int, end_index: int) -> np.ndarray:
    """CcsDataLoader.load_event_from_hdf, src/data_loader/ccs.py:275-297, on the raw columns of
    h5py_loader (:57-66: x int16 = column, y int16 = row, t int32 microseconds, p bool).
    Pinned (round 6) by tests/golden/golden_loader.npz: the reference's loader itself, run with the real h5py of the container's
    Anaconda interpreter on a synthetic recording (tests/golden/make_golden_loader.py; tests/test_loader_golden.py)."""
    n_events = end_index - start_index
    events = np.zeros((n_events, 4), dtype=np.float64)
    if len(x) <= start_index:
        raise IndexError
    events[:, 0] = y[start_index:end_index]
    events[:, 1] = x[start_index:end_index]
    events[:, 2] = t[start_index:end_index] / 1e6  # from micro sec to sec
    events[:, 3] = p[start_index:end_index]
    return events


def synth_raw_columns(n: int, height: int, width: int, seed: int = 0, t0_us: int = 10_000_000, span_us: int = 8300):
    """Synthetic raw_events columns with the dtypes of src/data_loader/ccs.py:63-66."""
    rs = np.random.RandomState(seed)
    x = rs.randint(0, width, n).astype(np.int16)
    y = rs.randint(0, height, n).astype(np.int16)
    t = np.sort(rs.randint(t0_us, t0_us + span_us, n)).astype(np.int32)
    p = rs.randint(0, 2, n).astype(bool)
    return x, y, t, p


def synth_events(n: int, height: int, width: int, seed: int = 0, tmin=0.0, tmax=0.5) -> np.ndarray:
    """Distribution of src/utils/event_utils.py:40-47 with an explicit seed."""
    rs = np.random.RandomState(seed)
    x = rs.randint(0, height, n)
    y = rs.randint(0, width, n)
    t = np.sort(rs.uniform(tmin, tmax, n))
    p = rs.randint(0, 2, n)
    return np.stack([x, y, t, p], axis=1).astype(np.float64)


def synth_dense_flow(height: int, width: int, seed: int = 1, max_val: float = 30.0) -> np.ndarray:
    """src/utils/flow_utils.py:29 with an explicit seed."""
    return np.random.RandomState(seed).uniform(-max_val, max_val, (2, height, width))


def rel_l2(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
