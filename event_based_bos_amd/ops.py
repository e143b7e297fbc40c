"""Tensor-level operators over libebos_hip.so, with autograd.

Every function here takes CUDA (ROCm) tensors, launches hand-written gfx950 kernels through the
C ABI on torch's current stream, and returns CUDA tensors.  torch is used for memory, streams and
the autograd graph only.  Shapes are always the canonical batched ones ([b, n, 4] events,
[b, 2, H, W] flow, [K, h, w] images); the reference's numpy/torch, batched/un-batched polymorphism
is handled one level up (warp.py, event_image_converter.py).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _hip
from ._hip import check, ptr, stream_ptr, suffix

_INT32 = torch.int32


def _cuda_contig(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _hip.HipUnavailableError(f"{name} must live on the GPU (got {t.device}); there is no CPU path")
    return t if t.is_contiguous() else t.contiguous()


def _same(a: torch.Tensor, b: torch.Tensor, what: str) -> None:
    if a.dtype != b.dtype or a.device != b.device:
        raise TypeError(f"{what}: dtype/device mismatch ({a.dtype},{a.device}) vs ({b.dtype},{b.device})")


# ----------------------------------------------------------------------------------------------
# A2  time range
# ----------------------------------------------------------------------------------------------
def time_range(events: torch.Tensor) -> torch.Tensor:
    """events [b, n, 4] -> [b, 2] (min t, max t) per batch row.  src/warp.py:245-253."""
    lib = _hip.require_gpu()
    events = _cuda_contig(events, "events")
    b, n, _ = events.shape
    out = torch.empty((b, 2), dtype=events.dtype, device=events.device)
    with _hip.on_device(events.device):
        fn = getattr(lib, "ebos_time_range_" + suffix(events.dtype))
        check(fn(ptr(events), b, n, ptr(out), stream_ptr()), "ebos_time_range")
    return out


# ----------------------------------------------------------------------------------------------
# A3  dense-flow warp
# ----------------------------------------------------------------------------------------------
class _WarpDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, events, flow, ref_mode, ref_fraction, normalize_t, row_stride, oob, timebase, tmm):
        lib = _hip.require_gpu()
        events = _cuda_contig(events, "events")
        flow = _cuda_contig(flow, "flow")
        _same(events, flow, "warp_dense(events, flow)")
        b, n, _ = events.shape
        H, W = flow.shape[-2:]
        if timebase is not None:  # explicit (t_ref, period) per batch row
            tmm = _cuda_contig(timebase.to(events.dtype).reshape(b, 2), "timebase")
            ref_mode = _hip.REF_TIMEBASE
        elif tmm is None:  # (a caller that warps the same window again and again passes the range it already has)
            tmm = time_range(events)
        out = torch.empty_like(events)
        with _hip.on_device(events.device):
            fn = getattr(lib, "ebos_warp_dense_" + suffix(events.dtype))
            check(fn(ptr(events), ptr(flow), ptr(tmm), ref_mode, ref_fraction, int(normalize_t), b, n, H, W,
                     row_stride, ptr(out), ptr(oob), stream_ptr()), "ebos_warp_dense")
        ctx.save_for_backward(events, tmm)
        ctx.meta = (ref_mode, ref_fraction, int(normalize_t), row_stride, tuple(flow.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        events, tmm = ctx.saved_tensors
        ref_mode, ref_fraction, normalize_t, row_stride, fshape = ctx.meta
        d_flow = None
        if ctx.needs_input_grad[1]:
            lib = _hip.require_gpu()
            g = _cuda_contig(g, "grad")
            b, n, _ = events.shape
            d_flow = torch.zeros(fshape, dtype=events.dtype, device=events.device)
            with _hip.on_device(events.device):
                fn = getattr(lib, "ebos_warp_dense_bwd_" + suffix(events.dtype))
                check(fn(ptr(events), ptr(tmm), ref_mode, ref_fraction, normalize_t, ptr(g), b, n, fshape[-2],
                         fshape[-1], row_stride, ptr(d_flow), stream_ptr()), "ebos_warp_dense_bwd")
        return None, d_flow, None, None, None, None, None, None, None


def warp_dense(events: torch.Tensor, flow: torch.Tensor, ref_mode: int, ref_fraction: float, normalize_t: bool,
               row_stride: Optional[int] = None, oob: Optional[torch.Tensor] = None,
               timebase: Optional[torch.Tensor] = None, tmm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """events [b, n, 4], flow [b, 2, H, W] -> warped [b, n, 4].  src/warp.py:292-342.
    ``timebase`` [b, 2] = (t_ref, period) overrides the direction-derived reference time; ``tmm`` [b, 2] = the (min t,
    max t) of ``time_range(events)`` if the caller already has it."""
    if events.dim() != 3 or events.shape[-1] != 4 or flow.dim() != 4 or flow.shape[1] != 2:
        raise ValueError(f"warp_dense expects events [b,n,4] and flow [b,2,H,W], got {tuple(events.shape)}, {tuple(flow.shape)}")
    if events.shape[0] != flow.shape[0]:
        raise ValueError("warp_dense: batch sizes of events and flow differ")
    stride = int(flow.shape[-1] if row_stride is None else row_stride)
    return _WarpDense.apply(events, flow, int(ref_mode), float(ref_fraction), bool(normalize_t), stride, oob, timebase, tmm)


# ----------------------------------------------------------------------------------------------
# A4  2-DoF warp
# ----------------------------------------------------------------------------------------------
class _Warp2Dof(torch.autograd.Function):
    @staticmethod
    def forward(ctx, events, theta, ref_mode, ref_fraction, normalize_t, time_period, timebase):
        lib = _hip.require_gpu()
        events = _cuda_contig(events, "events")
        theta = _cuda_contig(theta, "theta")
        _same(events, theta, "warp_2dof(events, theta)")
        n = events.shape[0]
        if timebase is not None:
            tmm = _cuda_contig(timebase.to(events.dtype).reshape(1, 2), "timebase")
            ref_mode = _hip.REF_TIMEBASE
        else:
            tmm = time_range(events[None])
        out = torch.empty_like(events)
        if time_period is not None:
            time_period = _cuda_contig(time_period.to(events.dtype).reshape(1), "time_period")
        with _hip.on_device(events.device):
            fn = getattr(lib, "ebos_warp_2dof_" + suffix(events.dtype))
            check(fn(ptr(events), ptr(theta), ptr(tmm), ref_mode, ref_fraction, int(normalize_t), ptr(time_period), n,
                     ptr(out), stream_ptr()), "ebos_warp_2dof")
        ctx.save_for_backward(events, tmm, time_period if time_period is not None else torch.empty(0))
        ctx.meta = (ref_mode, ref_fraction, int(normalize_t), time_period is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        events, tmm, period = ctx.saved_tensors
        ref_mode, ref_fraction, normalize_t, has_period = ctx.meta
        d_theta = None
        if ctx.needs_input_grad[1]:
            lib = _hip.require_gpu()
            g = _cuda_contig(g, "grad")
            d_theta = torch.zeros(2, dtype=events.dtype, device=events.device)
            with _hip.on_device(events.device):
                fn = getattr(lib, "ebos_warp_2dof_bwd_" + suffix(events.dtype))
                check(fn(ptr(events), ptr(tmm), ref_mode, ref_fraction, normalize_t, ptr(period) if has_period else None,
                         ptr(g), events.shape[0], ptr(d_theta), stream_ptr()), "ebos_warp_2dof_bwd")
        return None, d_theta, None, None, None, None, None


def warp_2dof(events: torch.Tensor, theta: torch.Tensor, ref_mode: int, ref_fraction: float, normalize_t: bool,
              time_period: Optional[torch.Tensor] = None, timebase: Optional[torch.Tensor] = None) -> torch.Tensor:
    """events [n, 4], theta [2] -> warped [n, 4].  src/warp.py:344-383."""
    if events.dim() != 2 or events.shape[-1] != 4:
        raise ValueError(f"warp_2dof expects un-batched events [n,4], got {tuple(events.shape)}")
    return _Warp2Dof.apply(events, theta.reshape(-1)[:2], int(ref_mode), float(ref_fraction), bool(normalize_t),
                           time_period, timebase)


# ----------------------------------------------------------------------------------------------
# A7/A8/A10  event -> image accumulation
# ----------------------------------------------------------------------------------------------
class _Splat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, events, weight, weight_scalar, mode, eps, h, w, pad_h, pad_w):
        lib = _hip.require_gpu()
        events = _cuda_contig(events, "events")
        b, n, _ = events.shape
        if weight is not None:
            weight = _cuda_contig(weight.to(events.dtype), "weight")
        shape = (b, 2, h, w) if mode == _hip.SPLAT_POLARITY else (b, h, w)
        image = torch.zeros(shape, dtype=events.dtype, device=events.device)
        with _hip.on_device(events.device):
            fn = getattr(lib, "ebos_splat_" + suffix(events.dtype))
            check(fn(ptr(events), ptr(weight), float(weight_scalar), mode, float(eps), b, n, h, w, pad_h, pad_w,
                     ptr(image), stream_ptr()), "ebos_splat")
        ctx.save_for_backward(events, weight if weight is not None else torch.empty(0))
        ctx.meta = (weight is not None, float(weight_scalar), mode, float(eps), h, w, pad_h, pad_w)
        return image

    @staticmethod
    def backward(ctx, g):
        events, weight = ctx.saved_tensors
        has_w, ws, mode, eps, h, w, pad_h, pad_w = ctx.meta
        need_e, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and has_w
        d_events = d_weight = None
        if need_e or need_w:
            if mode != _hip.SPLAT_BILINEAR:
                raise NotImplementedError("only method='bilinear_vote' is differentiable")
            lib = _hip.require_gpu()
            g = _cuda_contig(g, "grad")
            b, n, _ = events.shape
            if need_e:
                d_events = torch.empty_like(events)
            if need_w:
                d_weight = torch.empty((b, n), dtype=events.dtype, device=events.device)
            with _hip.on_device(events.device):
                fn = getattr(lib, "ebos_splat_bwd_" + suffix(events.dtype))
                check(fn(ptr(events), ptr(weight) if has_w else None, ws, eps, ptr(g), b, n, h, w, pad_h, pad_w,
                         ptr(d_events), ptr(d_weight), stream_ptr()), "ebos_splat_bwd")
        return d_events, d_weight, None, None, None, None, None, None, None


def splat(events: torch.Tensor, image_size: Tuple[int, int], pad: Tuple[int, int] = (0, 0),
          weight=1.0, mode: int = _hip.SPLAT_BILINEAR, eps: float = 1e-6) -> torch.Tensor:
    """events [b, n, 4] -> image [b, h, w] ([b, 2, h, w] for polarity); ``image_size`` is the
    PADDED size.  src/event_image_converter.py:407-620."""
    if events.dim() != 3 or events.shape[-1] != 4:
        raise ValueError(f"splat expects events [b,n,4], got {tuple(events.shape)}")
    h, w = int(image_size[0]), int(image_size[1])
    if isinstance(weight, torch.Tensor) and weight.dim() > 0:
        wt, ws = weight.reshape(events.shape[0], events.shape[1]), 1.0
    elif isinstance(weight, torch.Tensor):
        # 0-d tensor weight: keep it differentiable by scaling the unit-weight image
        return _Splat.apply(events, None, 1.0, mode, eps, h, w, int(pad[0]), int(pad[1])) * weight.to(events.device)
    else:
        wt, ws = None, float(weight)
    return _Splat.apply(events, wt, ws, int(mode), float(eps), h, w, int(pad[0]), int(pad[1]))


# ----------------------------------------------------------------------------------------------
# A14  contrast costs on images
# ----------------------------------------------------------------------------------------------
def _cost_scratch(lib, K: int, device) -> torch.Tensor:
    return torch.empty(int(lib.ebos_cost_scratch_bytes(K)), dtype=torch.uint8, device=device)


class _ImageVariance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, omit):
        lib = _hip.require_gpu()
        images = _cuda_contig(images, "iwe")
        K, h, w = images.shape
        out = torch.empty(K, dtype=images.dtype, device=images.device)
        moments = torch.empty((K, 2), dtype=torch.float64, device=images.device)
        scratch = _cost_scratch(lib, K, images.device)
        with _hip.on_device(images.device):
            fn = getattr(lib, "ebos_image_variance_" + suffix(images.dtype))
            check(fn(ptr(images), K, h, w, int(omit), ptr(out), ptr(moments), ptr(scratch), scratch.numel(),
                     stream_ptr()), "ebos_image_variance")
        ctx.save_for_backward(images, moments)
        ctx.omit = int(omit)
        return out

    @staticmethod
    def backward(ctx, g):
        images, moments = ctx.saved_tensors
        lib = _hip.require_gpu()
        K, h, w = images.shape
        g = _cuda_contig(g.to(images.dtype), "grad")
        d = torch.empty_like(images)
        with _hip.on_device(images.device):
            fn = getattr(lib, "ebos_image_variance_grad_" + suffix(images.dtype))
            check(fn(ptr(images), K, h, w, ctx.omit, ptr(moments), ptr(g), ptr(d), stream_ptr()),
                  "ebos_image_variance_grad")
        return d, None


class _GradientMagnitude(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, omit):
        lib = _hip.require_gpu()
        images = _cuda_contig(images, "iwe")
        K, h, w = images.shape
        out = torch.empty(K, dtype=images.dtype, device=images.device)
        scratch = _cost_scratch(lib, K, images.device)
        with _hip.on_device(images.device):
            fn = getattr(lib, "ebos_gradient_magnitude_" + suffix(images.dtype))
            check(fn(ptr(images), K, h, w, int(omit), ptr(out), ptr(scratch), scratch.numel(), stream_ptr()),
                  "ebos_gradient_magnitude")
        ctx.save_for_backward(images)
        ctx.omit = int(omit)
        return out

    @staticmethod
    def backward(ctx, g):
        (images,) = ctx.saved_tensors
        lib = _hip.require_gpu()
        K, h, w = images.shape
        g = _cuda_contig(g.to(images.dtype), "grad")
        d = torch.empty_like(images)
        with _hip.on_device(images.device):
            fn = getattr(lib, "ebos_gradient_magnitude_grad_" + suffix(images.dtype))
            check(fn(ptr(images), K, h, w, ctx.omit, ptr(g), ptr(d), stream_ptr()), "ebos_gradient_magnitude_grad")
        return d, None


def _as_stack(images: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    if images.dim() == 2:
        return images[None], True
    if images.dim() == 3:
        return images, False
    raise ValueError(f"expected an image [h,w] or a stack [K,h,w], got {tuple(images.shape)}")


def image_variance(images: torch.Tensor, omit_boundary: bool = False) -> torch.Tensor:
    """Unbiased variance of each image ([h,w] -> 0-d, [K,h,w] -> [K])."""
    st, single = _as_stack(images)
    out = _ImageVariance.apply(st, bool(omit_boundary))
    return out[0] if single else out


def gradient_magnitude(images: torch.Tensor, omit_boundary: bool = False) -> torch.Tensor:
    """mean(gx^2 + gy^2) with Sobel3/8, replicate padding ([h,w] -> 0-d, [K,h,w] -> [K])."""
    st, single = _as_stack(images)
    out = _GradientMagnitude.apply(st, bool(omit_boundary))
    return out[0] if single else out


# ----------------------------------------------------------------------------------------------
# A16  patch grid -> dense flow
# ----------------------------------------------------------------------------------------------
class _UpsamplePatchFlow(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, patch, slide, image_size):
        lib = _hip.require_gpu()
        grid = _cuda_contig(grid.float(), "patch flow")
        _, gh, gw = grid.shape
        H, W = image_size
        dense = torch.empty((2, H, W), dtype=torch.float32, device=grid.device)
        with _hip.on_device(grid.device):
            check(lib.ebos_upsample_patch_flow_f32(ptr(grid), gh, gw, patch[0], patch[1], slide[0], slide[1], H, W,
                                                   ptr(dense), stream_ptr()), "ebos_upsample_patch_flow")
        ctx.meta = (gh, gw, patch, slide, H, W)
        return dense

    @staticmethod
    def backward(ctx, g):
        lib = _hip.require_gpu()
        gh, gw, patch, slide, H, W = ctx.meta
        g = _cuda_contig(g.float(), "grad")
        d = torch.empty((2, gh, gw), dtype=torch.float32, device=g.device)
        scratch = torch.empty(int(lib.ebos_upsample_bwd_scratch_bytes(gh, W)) // 4, dtype=torch.float32, device=g.device)
        with _hip.on_device(g.device):
            check(lib.ebos_upsample_patch_flow_bwd_f32(ptr(g), gh, gw, patch[0], patch[1], slide[0], slide[1], H, W,
                                                       ptr(scratch), ptr(d), stream_ptr()), "ebos_upsample_patch_flow_bwd")
        return d, None, None, None


def upsample_patch_flow(grid: torch.Tensor, patch_size, sliding_window, image_size) -> torch.Tensor:
    """[2, gh, gw] patch flow -> [2, H, W] dense flow (f32).  src/solver/patch_eklt.py:173-204."""
    if grid.dim() != 3 or grid.shape[0] != 2:
        raise ValueError(f"patch flow must be [2, gh, gw], got {tuple(grid.shape)}")
    dt = grid.dtype
    out = _UpsamplePatchFlow.apply(grid, (int(patch_size[0]), int(patch_size[1])),
                                   (int(sliding_window[0]), int(sliding_window[1])),
                                   (int(image_size[0]), int(image_size[1])))
    return out if dt == torch.float32 else out.to(dt)


# ----------------------------------------------------------------------------------------------
# K11  Gaussian blur passes
# ----------------------------------------------------------------------------------------------
class _Gauss1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, axis, taps, boundary):
        ctx.save_for_backward(taps)
        ctx.axis, ctx.boundary = axis, boundary
        return _gauss1d_launch(x, axis, taps, boundary, "ebos_gauss1d_")

    @staticmethod
    def backward(ctx, g):
        (taps,) = ctx.saved_tensors
        return _gauss1d_launch(_cuda_contig(g, "grad"), ctx.axis, taps, ctx.boundary, "ebos_gauss1d_bwd_"), None, None, None


def _gauss1d_launch(x: torch.Tensor, axis: int, taps: torch.Tensor, boundary: int, entry: str) -> torch.Tensor:
    lib = _hip.require_gpu()
    L = x.shape[axis]
    outer = 1
    for s in x.shape[:axis]:
        outer *= s
    inner = 1
    for s in x.shape[axis + 1:]:
        inner *= s
    out = torch.empty_like(x)
    if x.numel() == 0:
        return out
    with _hip.on_device(x.device):
        fn = getattr(lib, entry + suffix(x.dtype))
        check(fn(ptr(x), ptr(out), outer, L, inner, ptr(taps), (taps.numel() - 1) // 2, boundary, stream_ptr()),
              entry.rstrip("_"))
    return out


def gauss1d(x: torch.Tensor, axis: int, taps: torch.Tensor, boundary: int) -> torch.Tensor:
    """One separable blur pass along ``axis``; taps = [2r+1] float64 (any device).  Differentiable in x."""
    _hip.require_gpu()
    x = _cuda_contig(x, "image")
    taps = taps.detach().to(device=x.device, dtype=torch.float64).contiguous()
    return _Gauss1d.apply(x, axis % x.dim(), taps, int(boundary))
