"""Lazy fusion of the reference's two-call idiom

    warped, _ = warper.warp_event(events, flow, "dense-flow", direction)
    iwe = imager.create_iwe(warped, "bilinear_vote", sigma=0)

so that UNCHANGED solver code written against the reference's API runs on the fused tile-private kernels.

``Warp.warp_event`` still materialises and returns the warped events (they are a public result), but tags the
returned tensor with its provenance (source events, flow, reference-time mode).  When that very tensor -- unmodified,
checked through its version counter -- is handed to ``EventImageConverter`` with unit weight, the image is produced by
``EventPlan.iwe_dense`` (one pass over 6-12 B/event, autograd to the flow through the tile-private backward) instead
of a 4-atomics-per-event splat of the materialised coordinates.  The plan (SoA conversion + counting sort) is cached
per ``events`` tensor OBJECT (a weak reference, checked on every hit) and its version counter, so a solver loop pays
for it once per window.  An address is not an identity: a per-window loop frees ``events`` and the caching allocator
hands the same address (same shape, version 0) to a later window, so the entry dies with its tensor
(``weakref.finalize``) and a hit requires ``entry.ref() is events``.  The flow's version counter is recorded too: a
flow updated in place between ``warp_event`` and ``create_iwe`` (``optimizer.step()``, ``clamp_``) no longer matches
the materialised coordinates, and the image is then splatted from those, like the reference.

Policy (env ``EBOS_FUSE_API``): ``f32`` (default) fuses float32 inputs only -- the fused path computes in f32 and a
float64 caller is given the float64 kernels it asked for; ``all`` also fuses float64 inputs (result cast back);
``off`` disables it; ``lazy`` (opt-in) is ``f32`` plus a warped-events result that is only COMPUTED if something other than
``create_iwe`` looks at it (``LazyWarped``): a solver loop that warps, images, evaluates and steps never pays for the 16 bytes per
event of coordinates it does not read -- at 10 M events they were 220 of the idiom's 296 us.  What the caller gives up: warped
events that are first looked at AFTER the flow was updated in place can no longer be computed from the flow they belonged to, and
raise instead.

The third call of the idiom, ``cost.calculate({"iwe": iwe, ...})`` with the variance contrast, is fused too (``fused_variance``):
on the untouched image of ``fused_iwe`` it runs the objective's one native call and returns a result whose ``backward()`` does not
enter the autograd engine (``event_plan._EagerLoss``).
"""
from __future__ import annotations

import os
import weakref
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.utils import _pytree

from .event_plan import EventPlan

MAX_CACHED_PLANS = 4
stats = {"plan_builds": 0, "plan_hits": 0, "fused_images": 0}
_plans: "OrderedDict[tuple, tuple]" = OrderedDict()  # key -> (weakref to the events tensor, plan)


def policy() -> str:
    return os.environ.get("EBOS_FUSE_API", "f32").lower()


@dataclass
class Provenance:
    events: torch.Tensor          # the caller's [n, 4] tensor
    events_version: int
    flow: torch.Tensor            # [2, H, W] as passed (keeps its autograd history)
    ref_mode: int
    ref_fraction: float
    direction: object
    normalize_t: bool
    image_size: Tuple[int, int]
    warped_version: int = 0
    flow_version: int = 0


def eligible(events: torch.Tensor, flow: torch.Tensor, image_size) -> bool:
    mode = policy()
    if mode == "off" or not events.is_cuda or events.dim() != 2 or flow.dim() != 3:
        return False
    if events.dtype == torch.float64 and mode != "all":
        return False
    return tuple(flow.shape[-2:]) == (int(image_size[0]), int(image_size[1]))


def lazy_eligible(events: torch.Tensor, flow: torch.Tensor, image_size) -> bool:
    """``EBOS_FUSE_API=lazy``: float32 GPU events [n, 4] and flow [2, H, W] of one device, the shapes the fused image needs."""
    return policy() == "lazy" and type(events) is torch.Tensor and events.is_cuda and events.dim() == 2 and \
        events.dtype == torch.float32 and flow.dtype == torch.float32 and flow.device == events.device and \
        tuple(flow.shape) == (2, int(image_size[0]), int(image_size[1]))


class LazyWarped(torch.Tensor):
    """The warped events of ``Warp.warp_event`` under ``EBOS_FUSE_API=lazy``: shape, dtype and device are there at once, the
    coordinates are computed the first time anything reads them (every ``torch`` function and tensor method except the metadata
    queries goes through ``__torch_function__`` and runs on the computed tensor, autograd history included).
    ``EventImageConverter`` with unit weight does not read them: it builds the image from (events, flow) directly (``fused_iwe``)."""

    @staticmethod
    def make(events: torch.Tensor, flow: torch.Tensor, prov: "Provenance", compute) -> "LazyWarped":
        # (a stride-0 view of one element carries shape, dtype and device; nothing reads its 4 bytes)
        t = torch.Tensor._make_subclass(LazyWarped, torch.empty(1, dtype=events.dtype, device=events.device).expand(events.shape))
        t._ebos_lazy = [compute, None]  # thunk, computed tensor
        prov.flow_version = flow._version
        t._ebos_provenance = prov
        return t

    def _real(self) -> torch.Tensor:
        st = self._ebos_lazy
        if st[1] is None:
            prov = self._ebos_provenance
            if prov.flow._version != prov.flow_version or prov.events._version != prov.events_version:
                raise RuntimeError("EBOS_FUSE_API=lazy: these warped events are read for the first time after their flow (or their "
                                   "events) were modified in place; they can no longer be computed.  Read them before the update, "
                                   "or run with EBOS_FUSE_API=f32 (warped events computed at the call)")
            st[1] = st[0]()
            st[0] = None
        return st[1]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _LAZY_META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        args, kwargs = _pytree.tree_map_only(LazyWarped, lambda a: a._real(), (tuple(args), kwargs))
        return func(*args, **kwargs)


_LAZY_META = {torch.Tensor.shape.__get__, torch.Tensor.dtype.__get__, torch.Tensor.device.__get__, torch.Tensor.is_cuda.__get__,
              torch.Tensor.ndim.__get__, torch.Tensor.layout.__get__, torch.Tensor.dim, torch.Tensor.size, torch.Tensor.numel,
              torch.Tensor.is_floating_point, torch.Tensor.is_complex, torch.Tensor.__len__, torch.Tensor.nelement,
              torch.Tensor.element_size}


def tag(warped: torch.Tensor, prov: Provenance) -> torch.Tensor:
    prov.warped_version = warped._version
    prov.flow_version = prov.flow._version
    warped._ebos_provenance = prov
    return warped


def provenance_of(warped) -> Optional[Provenance]:
    if type(warped) is LazyWarped:
        if warped._ebos_lazy[1] is not None:  # it has been computed (someone read it): the computed tensor speaks for itself
            return provenance_of(warped._ebos_lazy[1])
        prov = warped._ebos_provenance     # never read, so never modified: only its sources can have changed
        if prov.events._version != prov.events_version or prov.flow._version != prov.flow_version:
            return None
        return prov
    prov = getattr(warped, "_ebos_provenance", None)
    if prov is None or warped._version != prov.warped_version or prov.events._version != prov.events_version:
        return None  # the warped events (or their source) were modified in place since the warp
    if prov.flow._version != prov.flow_version:
        return None  # the flow was updated in place since the warp: `warped` holds the OLD flow's coordinates
    return prov


def _evict(key) -> None:
    _plans.pop(key, None)


def plan_for(prov: Provenance) -> EventPlan:
    ev = prov.events
    key = (id(ev), ev.data_ptr(), tuple(ev.shape), ev.dtype, ev._version, ev.device.index, prov.ref_mode, prov.ref_fraction,
           prov.normalize_t, tuple(prov.image_size))
    entry = _plans.get(key)
    if entry is not None and entry[0]() is ev:  # the very tensor object the plan was built from, still alive
        _plans.move_to_end(key)
        stats["plan_hits"] += 1
        return entry[1]
    direction = {0: "first", 1: "last"}.get(prov.ref_mode, float(prov.ref_fraction))
    # (only unit-weight images are fused: the lean build -- compact events and offsets, nothing else -- is all they read)
    plan = EventPlan.build(ev.detach(), prov.image_size, direction, prov.normalize_t, tile="auto", emit="compact")
    _plans[key] = (weakref.ref(ev), plan)
    weakref.finalize(ev, _evict, key)  # the entry (and its device memory) goes when the caller drops the window
    stats["plan_builds"] += 1
    while len(_plans) > MAX_CACHED_PLANS:
        _plans.popitem(last=False)
    return plan


def fused_iwe(warped: torch.Tensor, padded_image_size, pad) -> Optional[torch.Tensor]:
    """IWE of provenance-tagged warped events through the fused kernels, or None if not applicable."""
    prov = provenance_of(warped)
    if prov is None:
        return None
    H, W = prov.image_size
    if tuple(padded_image_size) != (H + 2 * int(pad[0]), W + 2 * int(pad[1])):
        return None
    plan = plan_for(prov)
    if plan.n_dropped:  # the reference raises for out-of-range sources; leave that to the unfused path
        return None
    pad = (int(pad[0]), int(pad[1]))
    iwe = plan.iwe_dense(prov.flow, pad=pad)
    stats["fused_images"] += 1
    if iwe.dtype != warped.dtype:
        return iwe.to(warped.dtype)
    iwe._ebos_iwe = IweProvenance(plan, prov.flow, pad, iwe._version, prov.flow._version)
    return iwe


@dataclass
class IweProvenance:
    plan: EventPlan
    flow: torch.Tensor
    pad: Tuple[int, int]
    iwe_version: int
    flow_version: int


def carry_iwe_tag(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """``view`` is ``src`` reshaped without copying (the converter's ``squeeze``): it IS the image, keep the tag."""
    tag_ = getattr(src, "_ebos_iwe", None)
    if tag_ is not None and view is not src and view.data_ptr() == src.data_ptr() and view.numel() == src.numel():
        view._ebos_iwe = tag_
    return view


def fused_variance(iwe: torch.Tensor, omit_boundary: bool, cost: str = "image_variance") -> Optional[torch.Tensor]:
    """Third step of the idiom, ``cost.calculate({"iwe": iwe, ...})`` with a contrast cost (variance, or ``cost="gradient_magnitude"``) on an image that came out of
    ``fused_iwe`` and has not been touched since: value AND flow gradient by the objective's one native call
    (``EventPlan.contrast_dense`` -> ``_EagerLoss``: ``backward()`` on the cost, its negation or a weighted multiple stores the
    gradient without the autograd engine; combined with other terms it becomes an ordinary graph node).  The image's own autograd
    node is simply not used.  None when the short cut does not apply: the image (or the flow) was modified in place, someone
    asked for the image's gradient (``retain_grad`` / hooks), the flow is not a plain float32 leaf."""
    tag_ = getattr(iwe, "_ebos_iwe", None)
    if tag_ is None or iwe.dim() != 2 or iwe._version != tag_.iwe_version or tag_.flow._version != tag_.flow_version:
        return None
    if iwe.retains_grad or iwe._backward_hooks:
        return None
    from .event_plan import DEFAULT_HALO, _eager_ok, _norm_halo
    if not _eager_ok(tag_.plan, tag_.flow, _norm_halo(tag_.plan, DEFAULT_HALO)):
        return None
    stats["fused_costs"] = stats.get("fused_costs", 0) + 1
    return tag_.plan.contrast_dense(tag_.flow, cost, bool(omit_boundary), pad=tag_.pad)


def clear_cache() -> None:
    _plans.clear()
