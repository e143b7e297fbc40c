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

Policy (env ``EBOS_FUSE_API``): ``lazy`` (default) fuses float32 GPU inputs and makes the idiom's intermediate results
DEFERRED: ``warp_event`` returns a ``LazyWarped`` and ``create_iwe`` on it a ``LazyIwe`` -- tensors that know their shape, dtype and
device at once and compute their values the first time anything reads them.  A solver loop that warps, images, evaluates a contrast
cost and steps reads neither: the cost step runs the objective's one native call on (events, flow) and the 16 bytes per event of
warped coordinates (220 of the idiom's 296 us at 10 M events) and the separate image pass are never paid for.  A late read stays
exact: ``warp_event`` keeps a device copy of the flow as it is at the call (7.4 MB, ~3 us), so a result first read AFTER
``optimizer.step()`` updated the flow in place holds the OLD flow's coordinates, as the reference's eager tensor would
(src/warp.py:330-342); that late value is detached from the flow (a gradient into a leaf that has since been overwritten is not
reproduced).  Only events modified in place before the first read cannot be honoured (they are not copied: 160 MB) and raise.
``f32``: float32 inputs fused, warped events computed at the call (round 3's default); ``all`` also fuses float64 inputs (result
cast back); ``off`` disables the fusion.

The third call of the idiom, ``cost.calculate({"iwe": iwe, ...})`` with a contrast cost, is fused too (``fused_variance``):
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
    return os.environ.get("EBOS_FUSE_API", "lazy").lower()


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
    flow_snapshot: Optional[torch.Tensor] = None   # lazy results: the flow as it was at the warp_event call (device copy)
    flow_slot: Optional[list] = None               # ... and the ring slot that holds it (FlowSnapshots)


def eligible(events: torch.Tensor, flow: torch.Tensor, image_size) -> bool:
    mode = policy()
    if mode == "off" or not events.is_cuda or events.dim() != 2 or flow.dim() != 3:
        return False
    if events.dtype == torch.float64 and mode != "all":
        return False
    return tuple(flow.shape[-2:]) == (int(image_size[0]), int(image_size[1]))


def lazy_eligible(events: torch.Tensor, flow: torch.Tensor, image_size) -> bool:
    """``EBOS_FUSE_API=lazy`` (the default): float32 GPU events [n, 4] and flow [2, H, W] of one device, the shapes the fused image needs."""
    return policy() == "lazy" and type(events) is torch.Tensor and events.is_cuda and events.dim() == 2 and \
        events.dtype == torch.float32 and flow.dtype == torch.float32 and flow.device == events.device and \
        tuple(flow.shape) == (2, int(image_size[0]), int(image_size[1]))


class _Deferred(torch.Tensor):
    """A tensor that knows its shape, dtype and device at once and computes its values the first time anything reads them: every
    ``torch`` function and tensor method except the metadata queries goes through ``__torch_function__`` and runs on the computed
    tensor (autograd history included)."""

    _SHELLS: dict = {}

    @staticmethod
    def _shell(cls, shape, dtype, device):
        # (a stride-0 view of one element carries shape, dtype and device; nothing reads its 4 bytes -- so every shell of one
        # (shape, dtype, device) can alias ONE such view: a solver loop makes two shells per iteration)
        key = (tuple(shape), dtype, device)
        base = _Deferred._SHELLS.get(key)
        if base is None:
            if len(_Deferred._SHELLS) > 64:
                _Deferred._SHELLS.clear()
            base = _Deferred._SHELLS[key] = torch.empty(1, dtype=dtype, device=device).expand(shape)
        return torch.Tensor._make_subclass(cls, base)

    def _real(self) -> torch.Tensor:
        st = self._ebos_lazy
        if st[1] is None:
            st[1] = st[0]()
            st[0] = None
        return st[1]

    @property
    def computed(self) -> bool:
        return self._ebos_lazy[1] is not None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _LAZY_META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        args, kwargs = _pytree.tree_map_only(_Deferred, lambda a: a._real(), (tuple(args), kwargs))
        return func(*args, **kwargs)


def flow_for_read(prov: "Provenance") -> torch.Tensor:
    """The flow a deferred result computes its values from: the caller's tensor while it is what it was at the ``warp_event`` call
    (autograd history kept), else the copy made at that call -- the reference's eager tensors hold the OLD flow's values too."""
    if prov.events._version != prov.events_version:
        raise RuntimeError("these warped events (or their image) are read for the first time after their EVENTS were modified in "
                           "place; they can no longer be computed (the flow is copied at the call, 160 MB of events are not).  "
                           "Read them before the update, or run with EBOS_FUSE_API=f32 (warped events computed at the call)")
    if prov.flow._version == prov.flow_version:
        return prov.flow
    if prov.flow_snapshot is None:
        raise RuntimeError("no copy of the flow was kept for this deferred result")
    return prov.flow_snapshot


class LazyWarped(_Deferred):
    """The warped events of ``Warp.warp_event`` under ``EBOS_FUSE_API=lazy``: computed the first time anything reads them.
    ``EventImageConverter`` with unit weight does not read them: it builds the image from (events, flow) directly (``fused_iwe``)."""

    @staticmethod
    def make(events: torch.Tensor, flow: torch.Tensor, prov: "Provenance", compute) -> "LazyWarped":
        """``compute(flow_tensor)`` -> the warped events of ``events`` under that flow."""
        t = _Deferred._shell(LazyWarped, events.shape, events.dtype, events.device)
        prov.flow_version = flow._version
        t._ebos_provenance = prov
        t._ebos_lazy = [lambda: compute(flow_for_read(prov)), None]  # thunk, computed tensor
        return t


class LazyIwe(_Deferred):
    """The image ``EventImageConverter`` makes of an unread ``LazyWarped`` with unit weight: computed (one fused pass over the
    events, ``EventPlan.iwe_dense``) the first time anything reads it; a contrast cost does not (``fused_variance``)."""

    @staticmethod
    def make(plan: EventPlan, prov: "Provenance", pad, shape) -> "LazyIwe":
        t = _Deferred._shell(LazyIwe, shape, prov.flow.dtype, prov.flow.device)
        t._ebos_iwe_lazy = (plan, prov, pad)

        def compute():
            fl = flow_for_read(prov)
            iwe = plan.iwe_dense(fl, pad=pad)
            stats["fused_images"] += 1
            iwe._ebos_iwe = IweProvenance(plan, fl, pad, iwe._version, fl._version)
            return iwe

        t._ebos_lazy = [compute, None]
        if prov.flow_slot is not None:
            FlowSnapshots.own(prov.flow_slot, t)
        return t


class FlowSnapshots(object):
    """Device copies of the flow for deferred results, a small ring per (device, shape): a slot is reused when its previous owner --
    the ``LazyWarped`` of an earlier call -- is gone or has been read; an owner that is still alive and unread (a caller keeping
    unread results of several calls) computes its values from the slot first."""

    def __init__(self, slots: int = 3):
        self.slots, self.rings = slots, {}

    def take(self, flow: torch.Tensor):
        key = (flow.device, tuple(flow.shape))
        ring = self.rings.get(key)
        if ring is None:
            ring = self.rings[key] = [[[None, []] for _ in range(self.slots)], 0]
        slots, cursor = ring
        slot = slots[cursor % self.slots]
        ring[1] = cursor + 1
        for ref in slot[1] or ():
            owner = ref()
            if owner is not None and not owner.computed:
                owner._real()  # (rare) its flow copy is about to be overwritten
        slot[1] = []
        if slot[0] is None:
            slot[0] = torch.empty_like(flow, requires_grad=False)
        with torch.no_grad():
            slot[0].copy_(flow)
        return slot

    @staticmethod
    def own(slot, lazy: "_Deferred") -> None:
        """``lazy`` (a LazyWarped, or the LazyIwe made of one) computes from this slot's copy if it is read late."""
        slot[1].append(weakref.ref(lazy))


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
        if warped.computed:  # someone read it: the computed tensor speaks for itself
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


def lazy_iwe_of(warped, padded_image_size, pad) -> Optional[torch.Tensor]:
    """``create_iwe`` on the unread ``LazyWarped`` of the idiom, unit weight, no blur: the deferred image, with nothing of the generic
    path's container handling in front of it (the loop is bound by host time: DESIGN 4.3 #42).  None: take the generic path."""
    st = warped._ebos_lazy
    if st[1] is not None:
        return None
    prov = warped._ebos_provenance
    if prov.events._version != prov.events_version or prov.flow._version != prov.flow_version:
        return None
    H, W = prov.image_size
    ph, pw = int(pad[0]), int(pad[1])
    if tuple(padded_image_size) != (H + 2 * ph, W + 2 * pw) or H + 2 * ph == 1 or W + 2 * pw == 1:
        return None  # (the converter squeezes its result: an image with a unit dimension takes the generic path)
    plan = plan_for(prov)
    if plan.n_dropped:
        return None
    return LazyIwe.make(plan, prov, (ph, pw), (H + 2 * ph, W + 2 * pw))


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
    if type(warped) is LazyWarped and not warped.computed and prov.flow.dtype == torch.float32 and warped.dtype == torch.float32:
        return LazyIwe.make(plan, prov, pad, (H + 2 * pad[0], W + 2 * pad[1]))  # computed if (and when) something reads it
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


def squeezed(img: torch.Tensor) -> torch.Tensor:
    """``img.squeeze()`` as the converter applies it to its result (src/event_image_converter.py:405,620), keeping the fusion tag;
    a deferred image whose shape has no unit dimension is returned as it is (squeezing it would compute it)."""
    if type(img) is LazyIwe and not img.computed and all(int(d) != 1 for d in img.shape):
        return img
    return carry_iwe_tag(img, img.squeeze())


def carry_iwe_tag(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """``view`` is ``src`` reshaped without copying (the converter's ``squeeze``): it IS the image, keep the tag."""
    tag_ = getattr(src, "_ebos_iwe", None)
    if tag_ is not None and view is not src and view.data_ptr() == src.data_ptr() and view.numel() == src.numel():
        view._ebos_iwe = tag_
    return view


def fused_variance(iwe: torch.Tensor, omit_boundary: bool, cost: str = "image_variance", sign: float = 1.0) -> Optional[torch.Tensor]:
    """Third step of the idiom, ``cost.calculate({"iwe": iwe, ...})`` with a contrast cost (variance, or ``cost="gradient_magnitude"``) on an image that came out of
    ``fused_iwe`` and has not been touched since: value AND flow gradient by the objective's one native call
    (``EventPlan.contrast_dense`` -> ``_EagerLoss``: ``backward()`` on the cost, its negation or a weighted multiple stores the
    gradient without the autograd engine; combined with other terms it becomes an ordinary graph node).  The image's own autograd
    node is simply not used.  None when the short cut does not apply: the image (or the flow) was modified in place, someone
    asked for the image's gradient (``retain_grad`` / hooks), the flow is not a plain float32 leaf.  ``sign``: the cost's direction
    (-1 for "minimize"), applied inside the kernels -- the result is ``sign * contrast``."""
    if type(iwe) is LazyIwe and not iwe.computed:
        # the deferred image of the idiom, never read: the objective's one native call on (events, flow) -- unless the flow has been
        # updated in place since warp_event (then the image of the OLD flow is computed from the copy, like any other read)
        plan, prov, pad = iwe._ebos_iwe_lazy
        if prov.flow._version != prov.flow_version or prov.events._version != prov.events_version:
            return None
        from .event_plan import DEFAULT_HALO, _eager_ok, _norm_halo
        if not _eager_ok(plan, prov.flow, _norm_halo(plan, DEFAULT_HALO)):
            return None
        stats["fused_costs"] = stats.get("fused_costs", 0) + 1
        return plan.contrast_dense(prov.flow, cost, bool(omit_boundary), pad=pad, sign=sign, _eager_checked=True)
    tag_ = getattr(iwe, "_ebos_iwe", None)
    if tag_ is None or iwe.dim() != 2 or iwe._version != tag_.iwe_version or tag_.flow._version != tag_.flow_version:
        return None
    if iwe.retains_grad or iwe._backward_hooks:
        return None
    from .event_plan import DEFAULT_HALO, _eager_ok, _norm_halo
    if not _eager_ok(tag_.plan, tag_.flow, _norm_halo(tag_.plan, DEFAULT_HALO)):
        return None
    stats["fused_costs"] = stats.get("fused_costs", 0) + 1
    return tag_.plan.contrast_dense(tag_.flow, cost, bool(omit_boundary), pad=tag_.pad, sign=sign)


def clear_cache() -> None:
    _plans.clear()
