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
``off`` disables it.
"""
from __future__ import annotations

import os
import weakref
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

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


def tag(warped: torch.Tensor, prov: Provenance) -> torch.Tensor:
    prov.warped_version = warped._version
    prov.flow_version = prov.flow._version
    warped._ebos_provenance = prov
    return warped


def provenance_of(warped) -> Optional[Provenance]:
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
    iwe = plan.iwe_dense(prov.flow, pad=(int(pad[0]), int(pad[1])))
    stats["fused_images"] += 1
    return iwe if iwe.dtype == warped.dtype else iwe.to(warped.dtype)


def clear_cache() -> None:
    _plans.clear()
