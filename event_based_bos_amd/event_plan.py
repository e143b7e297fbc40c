"""EventPlan: the device-resident, iteration-invariant form of one event window, and the fused
warp+IWE operators that run on it.

In a contrast-maximisation loop the events of a window never change -- only the
flow / motion hypothesis does.  The plan is therefore built once per window:

  AoS [n,4] (f32|f64)  --ebos_events_to_soa-->  SoA f32 (x, y, dt, p), dt evaluated in fp64
                       --ebos_bin_events---->  counting-sorted by source pixel, tile-major
                                               (+ key_offsets per source pixel, perm back to input order)

and every objective evaluation afterwards is one fused kernel over 12-16 B/event
(``iwe_dense`` / ``iwe_2dof``), its backward, and a cost kernel over the 3.7 MB image.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
from torch.utils import _pytree

from . import _hip, ops
from ._hip import check, ptr, stream_ptr

DEFAULT_TILE = (64, 64)
DEFAULT_HALO = 32


def choose_tile(image_size: Tuple[int, int], halo: int = DEFAULT_HALO, n_cu: int = 256) -> Tuple[int, int]:
    """Tile size (among the built tile-private configurations with this halo) that fills the CUs best:
    one workgroup owns one tile, workgroups run ~one per CU, so the cost of a pass is about
    ceil(tiles / CUs) * (events of one tile + fixed per-tile work on its LDS window).
    720 x 1280 -> (45, 80): exactly 16 x 16 = 256 tiles."""
    H, W = image_size
    best, best_cost = DEFAULT_TILE, None
    for th, tw, hl in _hip.slab_configs():
        if hl != halo:
            continue
        tiles = ((H + th - 1) // th) * ((W + tw - 1) // tw)
        rounds = (tiles + n_cu - 1) // n_cu
        cost = rounds * (th * tw + 0.15 * (th + 2 * hl) * (tw + 2 * hl))
        if best_cost is None or cost < best_cost:
            best, best_cost = (th, tw), cost
    return best


def parse_direction(direction: Union[str, float]) -> Tuple[int, float]:
    """Reference-time mode of src/warp.py:245-262 -> (ebos_reftime_mode, fraction)."""
    import numpy as np

    if type(direction) is float:
        return _hip.REF_FRACTION, direction
    if direction == "first":
        return _hip.REF_FIRST, 0.0
    if direction == "middle":
        return _hip.REF_FRACTION, 0.5
    if direction == "last":
        return _hip.REF_LAST, 1.0
    if direction == "random":
        return _hip.REF_FRACTION, float(np.random.uniform(low=0.0, high=1.0))
    if direction == "before":
        return _hip.REF_FRACTION, -1.0
    if direction == "after":
        return _hip.REF_FRACTION, 2.0
    raise ValueError(f"direction argument should be first, middle, last. Or float. {direction}")


def dt_bound_for(direction: Union[str, float], normalize_t: bool) -> Optional[float]:
    """max |dt| over the events of a window as the HOST knows it without looking at the events: with normalised time
    dt = (t - t_ref) / (t_max - t_min) lies in [-f, 1 - f] for a reference time at fraction f of the window
    (src/warp.py:245-253, 283-287).  None when the time stays in seconds (the bound is then the window's length: unknown here)."""
    if not normalize_t:
        return None
    _, frac = parse_direction(direction)
    return max(abs(frac), abs(1.0 - frac))


def resolve_halo(plan: "EventPlan", halo) -> int:
    """The ``halo`` argument of the tile-private operators -> the integer the C ABI takes.  An int is a built halo;
    ``"auto"`` asks for run-time windows per tile (``EBOS_HALO_AUTO``: each work item sizes its LDS window from a bound on ITS
    OWN displacements, |flow| over the tile x max |dt|), bounded by the largest halo built for the plan's tile; it needs a plan
    that knows its |dt| bound (normalised time) and falls back to that largest halo otherwise."""
    if halo != "auto":
        return int(halo)
    biggest = max((hl for th, tw, hl in _hip.slab_configs() if (th, tw) == tuple(plan.tile)), default=None)
    if biggest is None:
        raise ValueError(f"no tile-private kernels are built for tile {plan.tile}")
    if plan.dt_bound is None:
        return biggest
    return int(_hip.load_library().ebos_halo_auto(biggest, float(plan.dt_bound)))


def _norm_halo(plan: "EventPlan", halo):
    """None stays None (general kernels); "auto" becomes the ABI's EBOS_HALO_AUTO code on a binned plan."""
    if halo is None or halo != "auto":
        return halo
    return resolve_halo(plan, halo) if plan.binned else None


def _max_halo(halo: int) -> int:
    """the built halo behind a (possibly EBOS_HALO_AUTO-encoded) halo argument"""
    return halo if halo >= 0 else (-halo) & 255


_N_CU = {}


def _n_cu(device: torch.device) -> int:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _N_CU:
        _N_CU[idx] = torch.cuda.get_device_properties(idx).multi_processor_count
    return _N_CU[idx]


# per-work-item fixed work of the accumulate kernel (LDS clear, decode, slab store ~3.5 us) in events (~0.4 ns each)
PART_FIXED_EVENTS = 8192


def part_fixed_events(n_events: int, n_tiles: int, n_cu: int) -> int:
    """The fixed work of a work item in events, as ``ebos_plan_parts``' model prices it.  Where the tiles already fill the device, an
    extra work item is an extra ROUND on some CU, and a round has a floor (set-up, LDS clear, slab store) that does not shrink with
    the window: measured best on blobs of sigma 50 - 200 px at 1280 x 720 (tools/bench_skew_solver.py; four launches per iteration)
    were 65 k events at 1 M events, 32 k at 2 M, 8 - 16 k at 5 M, 8 k at 10 M -- i.e. ~6.5e10 / n (sigma 50 px at 2 M events: 88 ->
    76 us per iteration).  Few tiles on many CUs (a small sensor) keep the constant: their parts spread over idle CUs, no second round."""
    if 2 * n_tiles <= n_cu:
        return PART_FIXED_EVENTS
    return int(min(65536, max(PART_FIXED_EVENTS, 6.5e10 / max(int(n_events), 1))))


@dataclass
class EventPlan:
    """SoA f32 event window on one GPU (optionally binned by source tile)."""

    x: Optional[torch.Tensor]             # SoA f32 (None in a lean plan: emit="compact")
    y: Optional[torch.Tensor]
    dt: Optional[torch.Tensor]
    p: Optional[torch.Tensor]
    image_size: Tuple[int, int]           # (H, W) of the sensor / flow field
    n: int                                # events in the plan (out-of-image sources dropped when binned)
    n_input: int                          # events handed to build()
    tile: Optional[Tuple[int, int]] = None
    key_offsets: Optional[torch.Tensor] = None   # int32 [n_keys + 1]
    perm: Optional[torch.Tensor] = None          # int32 [n]: input index of each planned event
    n_dropped: int = 0
    grp_offsets: Optional[torch.Tensor] = None   # compact plan: int32 [tiles + 1] group offsets
    cpix: Optional[torch.Tensor] = None          # compact plan: int16 storage of u16 (row << 8 | col), padded groups
    cdt: Optional[torch.Tensor] = None           # compact plan: f32 dt, NaN in padding slots
    part_table: Optional[torch.Tensor] = None    # adaptive work items (ebos_plan_parts): int32 [5 tiles + 1]
    dt_bound: Optional[float] = None             # max |dt| over the events as known on the host (dt_bound_for), else None

    @property
    def binned(self) -> bool:
        return self.key_offsets is not None

    @property
    def device(self) -> torch.device:
        return (self.x if self.x is not None else self.key_offsets).device

    @property
    def lean(self) -> bool:
        """True for a plan that holds ONLY what the tile-private kernels read (compact events + offsets): no SoA arrays,
        no permutation -- per-event weights and the general (global-atomic) kernels are not available on it."""
        return self.x is None

    @property
    def compact(self) -> bool:
        """True when the tile-private kernels read the 6 B/event format (u16 pixel + f32 dt)."""
        return self.cpix is not None

    def clear_cache(self) -> None:
        """Free what the operators cached on the plan: workspaces, dense jobs and the sweep lanes of ``variance_2dof`` /
        ``variance_dense_many`` (streams, one workspace per hypothesis of a chunk, image buffers -- up to 1.7 GB at 1280x720)."""
        for k in ("_workspaces", "_jobs", "_sweep_lanes"):
            self.__dict__.pop(k, None)

    def resolve_splits(self, splits: Optional[int]) -> int:
        """``splits`` of the tile-private forward kernels: k >= 1 cuts every tile into k equal parts, 0 uses the plan's
        adaptive work items (heavy tiles cut into parts, ``ebos_plan_parts``).  ``None`` picks 0 when the plan has a
        split tile -- or cannot know, having been built without a host read-back -- and 1 otherwise (a plan whose tiles
        are even gains nothing from the 2x work-item grid)."""
        if splits is not None:
            return int(splits)
        if self.part_table is None:
            return 1
        used = self.__dict__.get("_parts_used")
        n_tiles = (self.part_table.numel() - 1) // 5
        return 1 if used is not None and used <= n_tiles else 0

    # The optimiser LOOPS (FusedPatchLoop / Fused2dofLoop: forward + backward + step per iteration) pay ~24 us per iteration for the
    # passes that sum the parts of the adaptive work items, and ~0.5 us per 1000 events their fullest tile holds beyond the average
    # one when every tile is ONE work item (2 M events at 1280x720, DESIGN 4.4 #62: fullest tile 2 x the average 66.9 us adaptive
    # against 48.8 with one item per tile, 6 x 70.9 against 63.4, 21 x 95 against 197): adaptive where that excess is large enough
    LOOP_ADAPTIVE_MIN_EXCESS = 48_000

    def resolve_loop_splits(self, splits: Optional[int]) -> int:
        """``resolve_splits`` for an optimiser loop: the adaptive work items only where the fullest tile holds at least
        ``LOOP_ADAPTIVE_MIN_EXCESS`` events more than the average tile."""
        s = self.resolve_splits(splits)
        fullest = self.__dict__.get("_fullest_tile")
        if splits is None and s == 0 and fullest is not None:
            n_tiles = (self.part_table.numel() - 1) // 5
            if fullest - self.n / max(n_tiles, 1) < self.LOOP_ADAPTIVE_MIN_EXCESS:
                return 1
        return s

    def _compact_ptrs(self):
        return (ptr(self.grp_offsets), ptr(self.cpix), ptr(self.cdt)) if self.compact else (None, None, None)

    @property
    def frac_compact(self):
        """(grp_offsets, cpix, cdt, cfx, cfy) of a plan whose source coordinates are fractional (undistorted events) -- the compact
        layout with the fractions per slot, read by the resident loops and the grid-sampling launches --, or None.  Built on the first
        access, on the current stream (``ebos_plan_compact_frac_f32``: every source pixel's events in a canonical order, so two builds
        of one window hold identical slots)."""
        d = self.__dict__
        if d.get("_frac") is None and d.get("_frac_pending"):
            lib = _hip.require_gpu()
            dev = self.x.device
            H, W = self.image_size
            th, tw = self.tile
            n_tiles = ((H + th - 1) // th) * ((W + tw - 1) // tw)
            cap = self.n + 3 * n_tiles + 8
            f_grp = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
            f_pix = torch.zeros(cap, dtype=torch.int16, device=dev)
            f_dt = torch.full((cap,), float("nan"), dtype=torch.float32, device=dev)
            f_x, f_y = torch.zeros(cap, dtype=torch.float32, device=dev), torch.zeros(cap, dtype=torch.float32, device=dev)
            with _hip.on_device(dev):
                check(lib.ebos_plan_compact_frac_f32(ptr(self.x), ptr(self.y), ptr(self.dt), ptr(self.key_offsets), self.n, H, W, th, tw,
                                                     ptr(f_grp), ptr(f_pix), ptr(f_dt), ptr(f_x), ptr(f_y), cap, stream_ptr()),
                      "ebos_plan_compact_frac")
            d["_frac"], d["_frac_pending"] = (f_grp, f_pix, f_dt, f_x, f_y), False
        return d.get("_frac")

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def build(events: torch.Tensor, image_size: Tuple[int, int], direction: Union[str, float] = "first",
              normalize_t: bool = True, tile: Optional[Tuple[int, int]] = DEFAULT_TILE, emit: str = "full",
              deferred: bool = False) -> "EventPlan":
        """events: [n, 4] (x=row, y=col, t, p), float32 or float64, on the GPU.

        ``tile=None`` keeps the input (time) order: only the general global-atomic kernels apply;
        ``tile="auto"`` picks the tile-private configuration that fills the GPU best (``choose_tile``).
        ``emit="compact"``: the LEAN build (``ebos_plan_lean``) -- only the compact events and offsets the tile-private
        kernels read, built by a two-level counting sort in a third of the time; such a plan takes no per-event weights
        (``plan.lean``).  Windows with fractional source coordinates fall back to the full build."""
        if emit not in ("full", "compact"):
            raise ValueError("emit must be 'full' or 'compact'")
        if isinstance(tile, str):
            if tile != "auto":
                raise ValueError("tile must be a (tile_h, tile_w) pair, None or 'auto'")
            tile = choose_tile(image_size)
        lib = _hip.require_gpu()
        if events.dim() != 2 or events.shape[-1] != 4:
            raise ValueError(f"EventPlan.build expects un-batched events [n,4], got {tuple(events.shape)}")
        if not events.is_cuda:
            raise _hip.HipUnavailableError("EventPlan.build: events must be on the GPU")
        events = events.contiguous()
        if emit == "compact" and deferred:
            # a lean plan is valid iff no source coordinate is fractional, and a deferred build never reads that count back: only
            # the raw int16 columns (build_raw) are integers by construction
            raise ValueError("EventPlan.build: emit='compact' with deferred=True needs integer source coordinates by construction: "
                             "use build_raw (sensor columns), or deferred=False")
        if emit == "compact" and tile is not None and events.shape[0] > 0:
            lean = _build_lean(0 if events.dtype == torch.float32 else 1, events, None, image_size, direction, normalize_t, tile,
                               1.0, deferred)
            if lean is not None:
                return lean
        H, W = int(image_size[0]), int(image_size[1])
        n = events.shape[0]
        ref_mode, frac = parse_direction(direction)
        dev = events.device
        tmm = ops.time_range(events[None])
        x, y, dt, p = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
        with _hip.on_device(dev):
            fn = getattr(lib, "ebos_events_to_soa_" + _hip.suffix(events.dtype))
            check(fn(ptr(events), ptr(tmm), ref_mode, frac, int(normalize_t), n, ptr(x), ptr(y), ptr(dt), ptr(p),
                     stream_ptr()), "ebos_events_to_soa")
        plan = EventPlan(x, y, dt, p, (H, W), n, n, dt_bound=dt_bound_for(direction, normalize_t))
        if tile is not None:
            plan = plan.bin(tile)
        return plan

    @staticmethod
    def build_raw(col: torch.Tensor, row: torch.Tensor, t: torch.Tensor, pol: torch.Tensor, image_size: Tuple[int, int],
                  direction: Union[str, float] = "first", normalize_t: bool = True,
                  tile: Optional[Tuple[int, int]] = DEFAULT_TILE, ticks_per_second: float = 1e6,
                  deferred: bool = False, emit: str = "full") -> "EventPlan":
        """Plan of a window given as raw sensor columns on the GPU -- ``raw_events/{x, y, t, p}`` of the CCS
        recordings: col int16 (sensor x), row int16 (sensor y), t int32/int64 ticks, pol bool/uint8
        (src/data_loader/ccs.py:57-66).  Same plan, bit for bit, as ``build`` on the float64 [n, 4] array the
        reference's loader makes of the window (:289-297), without materialising that array (32 B/event) anywhere."""
        if isinstance(tile, str):
            if tile != "auto":
                raise ValueError("tile must be a (tile_h, tile_w) pair, None or 'auto'")
            tile = choose_tile(image_size)
        lib = _hip.require_gpu()
        n = int(t.shape[0])
        for name, v, dts in (("col", col, (torch.int16,)), ("row", row, (torch.int16,)), ("t", t, (torch.int32, torch.int64)),
                             ("pol", pol, (torch.bool, torch.uint8))):
            if v.dim() != 1 or v.shape[0] != n or v.dtype not in dts:
                raise ValueError(f"EventPlan.build_raw: {name} must be a 1-D tensor of {n} elements, dtype in {dts}; "
                                 f"got {tuple(v.shape)} {v.dtype}")
            if not v.is_cuda:
                raise _hip.HipUnavailableError(f"EventPlan.build_raw: {name} must be on the GPU")
        if n == 0:
            raise IndexError("EventPlan.build_raw: empty window")  # the loader raises IndexError too (ccs.py:262-265)
        col, row, t = col.contiguous(), row.contiguous(), t.contiguous()
        pol = pol.contiguous().view(torch.uint8)
        if emit not in ("full", "compact"):
            raise ValueError("emit must be 'full' or 'compact'")
        if emit == "compact" and tile is not None:
            lean = _build_lean(2 if t.dtype == torch.int32 else 3, None, (col, row, t), image_size, direction, normalize_t, tile,
                               float(ticks_per_second), deferred)
            if lean is not None:
                return lean
        H, W = int(image_size[0]), int(image_size[1])
        ref_mode, frac = parse_direction(direction)
        dev = t.device
        ticks = torch.empty(2, dtype=torch.int64, device=dev)
        tmm = torch.empty(2, dtype=torch.float64, device=dev)
        x, y, dt, p = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
        with _hip.on_device(dev):
            check(lib.ebos_raw_time_range(ptr(t), t.element_size(), n, float(ticks_per_second), ptr(ticks), ptr(tmm),
                                          stream_ptr()), "ebos_raw_time_range")
            check(lib.ebos_raw_events_to_soa(ptr(col), ptr(row), ptr(t), t.element_size(), ptr(pol), float(ticks_per_second),
                                             ptr(tmm), ref_mode, frac, int(normalize_t), n, ptr(x), ptr(y), ptr(dt), ptr(p),
                                             stream_ptr()), "ebos_raw_events_to_soa")
        plan = EventPlan(x, y, dt, p, (H, W), n, n, dt_bound=dt_bound_for(direction, normalize_t))
        if tile is not None:
            plan = plan.bin(tile, deferred=deferred)  # int16 columns: integer coordinates by construction
        return plan

    def bin(self, tile: Tuple[int, int] = DEFAULT_TILE, deferred: bool = False) -> "EventPlan":
        """Counting-sort the plan by source pixel, tile-major (ebos_bin_events_f32).

        ``deferred=True`` skips the one host read-back of the build (how many events lie outside the image / have
        fractional coordinates), so that a whole window can be planned without the host ever waiting for the GPU
        (``solver.WindowPipeline``).  Only valid when every source coordinate is an integer -- raw sensor columns are --
        because the compact plan is then built unconditionally; ``n`` stays an upper bound (out-of-image events are in
        no tile range and are never read) and ``counts()`` reports the numbers later."""
        lib = _hip.require_gpu()
        H, W = self.image_size
        th, tw = int(tile[0]), int(tile[1])
        tiles_y, tiles_x = (H + th - 1) // th, (W + tw - 1) // tw
        n_keys = tiles_y * tiles_x * th * tw
        dev = self.device
        n = self.n
        n_pad = (n + 3) // 4 * 4 + 4  # the tile-private kernels read 4 events (16 B) per lane
        xs, ys, dts, ps = (torch.zeros(n_pad, dtype=torch.float32, device=dev) for _ in range(4))
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        key_offsets = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
        counts = torch.zeros(2, dtype=torch.int32, device=dev)  # [out-of-image sources, fractional sources]
        nbytes = int(lib.ebos_bin_scratch_bytes_events(n, H, W, th, tw))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _hip.on_device(dev):
            check(lib.ebos_bin_events_f32(ptr(self.x), ptr(self.y), ptr(self.dt), ptr(self.p), n, H, W, th, tw,
                                          ptr(xs), ptr(ys), ptr(dts), ptr(ps), ptr(perm), ptr(key_offsets), ptr(counts),
                                          counts.data_ptr() + 4, ptr(scratch), nbytes, stream_ptr()),
                  "ebos_bin_events")
        # (the work items depend on key_offsets only: enqueued in front of the read-back, which then carries their count as well)
        part_table = torch.empty(5 * tiles_y * tiles_x + 1, dtype=torch.int32, device=dev)
        with _hip.on_device(dev):
            check(lib.ebos_plan_parts(ptr(key_offsets), H, W, th, tw, _n_cu(dev), part_fixed_events(n, tiles_y * tiles_x, _n_cu(dev)),
                                      ptr(part_table), stream_ptr()), "ebos_plan_parts")
        used = fullest = None
        if deferred:
            dropped, fractional = 0, 0
        else:  # the ONE host read-back of the build: (outside the image, fractional sources, work items in use, fullest tile)
            dropped, fractional, used, fullest = _plan_facts(lib, key_offsets, H, W, th, tw, counts, part_table)
        kept = n - dropped
        src_perm = perm[:kept] if self.perm is None else self.perm[perm[:kept].long()]
        grp_offsets = cpix = cdt = None
        if fractional == 0 and th <= 256 and tw <= 256:
            # integer source coordinates (camera events): also build the compact 6 B/event plan
            n_tiles = tiles_y * tiles_x
            cap = kept + 3 * n_tiles + 8
            grp_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
            cpix = torch.zeros(cap, dtype=torch.int16, device=dev)
            cdt = torch.full((cap,), float("nan"), dtype=torch.float32, device=dev)
            with _hip.on_device(dev):
                check(lib.ebos_plan_compact_f32(ptr(xs), ptr(ys), ptr(dts), ptr(key_offsets), kept, H, W, th, tw,
                                                ptr(grp_offsets), ptr(cpix), ptr(cdt), cap, stream_ptr()), "ebos_plan_compact")
        # fractional source coordinates (undistorted events): the compact layout WITH the fractions per slot is built on the first
        # ``frac_compact`` access (the resident loops and the grid-sampling launches ask; API-parity operators and bench --fractional
        # never do and no longer pay the rank loop, the hot-pixel pass and 14 B/event: ADVICE r05)
        frac_pending = bool(fractional > 0 and th <= 256 and tw <= 256)
        out = EventPlan(xs[:kept], ys[:kept], dts[:kept], ps[:kept], self.image_size, kept, self.n_input,
                        (th, tw), key_offsets, src_perm, self.n_dropped + dropped, grp_offsets, cpix, cdt, part_table, self.dt_bound)
        out.__dict__["_frac"] = None          # (grp_offsets, cpix, cdt, cfx, cfy) once built
        out.__dict__["_frac_pending"] = frac_pending
        out.__dict__["_counts"], out.__dict__["_deferred"] = counts, bool(deferred)
        out.__dict__["_parts_used"], out.__dict__["_fullest_tile"] = used, fullest
        return out

    def counts(self) -> Tuple[int, int]:
        """(events outside the image, events with fractional source coordinates) seen by ``bin`` -- a host read-back;
        a ``deferred`` plan with fractional sources is invalid and raises here."""
        c = self.__dict__.get("_counts")
        if c is None:
            return self.n_dropped, 0
        dropped, fractional = (int(v) for v in c.tolist())
        if fractional and self.__dict__.get("_deferred"):
            raise ValueError(f"{fractional} events have fractional source coordinates: a deferred plan needs integer ones")
        return dropped, fractional

    # ------------------------------------------------------------------------------------------
    def pixel_event_counts(self) -> torch.Tensor:
        """Events per source pixel [H, W] int64 (device), read off the plan's counting sort: the histogram of
        ``(trunc(x), trunc(y))`` that ``crop_event`` with integer bounds counts in.  No kernel of its own, no host read."""
        if not self.binned:
            raise ValueError("pixel_event_counts needs a binned plan (EventPlan.build(..., tile=...))")
        H, W = self.image_size
        th, tw = self.tile
        ty, tx = -(-H // th), -(-W // tw)
        per_key = (self.key_offsets[1:] - self.key_offsets[:-1]).to(torch.int64)       # keys are tile-major
        return per_key.reshape(ty, tx, th, tw).permute(0, 2, 1, 3).reshape(ty * th, tx * tw)[:H, :W]

    def patch_event_counts(self, patch_size: Tuple[int, int], sliding_window: Tuple[int, int]) -> torch.Tensor:
        """``len(crop_event(events, p.x_min, p.x_max, p.y_min, p.y_max))`` for every patch p of the grid at once:
        [gh, gw] int64 on the device.  Replaces the per-patch loop over the whole event array of the reference's
        patch solvers (src/solver/patch_eklt.py:118-126, src/solver/patch_eklt_pyramid2.py:215-228: O(n_patch * n))
        by box sums over the per-pixel histogram the plan already holds.  Exact for events inside the image, which is
        all a binned plan keeps (``n_dropped`` counts the others)."""
        import numpy as np

        from .types import patch_bounds

        H, W = self.image_size
        x0, x1, y0, y1 = (torch.from_numpy(np.clip(b, 0, lim)).to(self.device)
                          for b, lim in zip(patch_bounds((H, W), patch_size, sliding_window), (H, H, W, W)))
        x1, y1 = torch.maximum(x1, x0), torch.maximum(y1, y0)
        integral = torch.zeros((H + 1, W + 1), dtype=torch.int64, device=self.device)
        integral[1:, 1:] = self.pixel_event_counts().cumsum(0).cumsum(1)
        return (integral[x1[:, None], y1[None, :]] - integral[x0[:, None], y1[None, :]]
                - integral[x1[:, None], y0[None, :]] + integral[x0[:, None], y0[None, :]])

    # ------------------------------------------------------------------------------------------
    def iwe_dense(self, flow: torch.Tensor, pad: Tuple[int, int] = (0, 0), weight: Optional[torch.Tensor] = None,
                  halo: Optional[int] = DEFAULT_HALO, splits: Optional[int] = None) -> torch.Tensor:
        """Fused dense-flow warp + bilinear IWE: flow [2, H, W] -> iwe [H + 2 pad_h, W + 2 pad_w].
        ``halo=None`` (or an un-binned plan) selects the general global-atomic kernel; ``halo="auto"``: run-time windows per
        tile (``resolve_halo``)."""
        return _FusedIweDense.apply(flow, weight, self, (int(pad[0]), int(pad[1])), _norm_halo(self, halo), self.resolve_splits(splits))

    def iwe_2dof(self, thetas: torch.Tensor, pad: Tuple[int, int] = (0, 0), weight: Optional[torch.Tensor] = None,
                 halo: Optional[int] = DEFAULT_HALO, splits: Optional[int] = None) -> torch.Tensor:
        """Fused 2-DoF warp + bilinear IWE for K hypotheses: thetas [K, 2] -> iwes [K, h, w].
        Binned plans use the tile-private pipeline (|dt * theta| beyond ``halo`` spills, still correct)."""
        return _FusedIwe2Dof.apply(thetas, weight, self, (int(pad[0]), int(pad[1])), _norm_halo(self, halo), self.resolve_splits(splits))

    def variance_2dof(self, thetas: torch.Tensor, omit_boundary: bool = False, pad: Tuple[int, int] = (0, 0),
                      halo: int = DEFAULT_HALO, splits: Optional[int] = None, chunk: int = 16,
                      n_streams: int = 3) -> torch.Tensor:
        """Variance contrast of K translation hypotheses (the solver's outer sweep): [K, 2] -> [K].
        No gradient; images go into reused buffers.  The hypotheses are independent: they are dealt in chunks to ``n_streams``
        HIP streams -- the combine / finalize kernels of one chunk run in the wave slots that the one-workgroup-per-CU accumulate
        kernel of another leaves free -- and on a compact plan a chunk's accumulate pass is ONE persistent launch in which every
        workgroup keeps its tile and walks the chunk's hypotheses (``ebos_iwe_2dof_slab_batch_f32``: one LDS clear per launch, the
        next hypothesis' first events requested while the current image is stored).  ``halo="auto"``: run-time windows per tile."""
        lib = _hip.require_gpu()
        halo = _norm_halo(self, halo)
        if not _slab_ok(self, halo):
            return ops.image_variance(self.iwe_2dof(thetas, pad, None, None), omit_boundary)
        th = thetas.detach().to(device=self.device, dtype=torch.float32).contiguous()
        K = th.shape[0]
        H, W = self.image_size
        h, w = H + 2 * pad[0], W + 2 * pad[1]
        splits = self.resolve_splits(splits)
        out = torch.empty(K, dtype=torch.float32, device=self.device)
        n_streams = max(1, int(n_streams))
        chunk = max(1, min(int(chunk), (K + n_streams - 1) // n_streams))  # (a short sweep does not allocate 16 workspaces per lane)
        n_streams = max(1, min(n_streams, (K + chunk - 1) // chunk))
        persistent = self.compact and w % 4 == 0 and pad[1] % 4 == 0  # (what the batched pass asks of its images; else one launch per hypothesis)
        key = ("sweep", int(halo), int(splits), int(pad[0]), int(pad[1]), h, w, chunk, persistent)
        lanes = self.__dict__.setdefault("_sweep_lanes", {}).get(key)
        # (the size is asked of the library: _workspace() would allocate and cache a zero-filled workspace just to be measured)
        nws = int(lib.ebos_iwe_slab_workspace_bytes(H, W, self.tile[0], self.tile[1], int(halo), int(splits), pad[0], pad[1]))
        nws_al = (nws + 255) // 256 * 256
        with _hip.on_device(self.device):
            if lanes is None or len(lanes) < n_streams:  # streams, workspaces and image buffers live with the plan
                # one workspace per hypothesis of a chunk (36 MB each at 1280x720, tile 45x80, halo 32: 1.7 GB for 3 lanes of 16);
                # EventPlan.clear_cache() frees them
                lanes = [(torch.cuda.Stream(device=self.device),
                          torch.zeros((chunk if persistent else 1) * nws_al, dtype=torch.uint8, device=self.device),
                          torch.empty((chunk, h, w), dtype=torch.float32, device=self.device)) for _ in range(n_streams)]
                self.__dict__["_sweep_lanes"][key] = lanes
            cur = torch.cuda.current_stream(self.device)
            for st, _, _ in lanes[:n_streams]:
                st.wait_stream(cur)  # thetas / the plan were produced on the caller's stream
            for j, k0 in enumerate(range(0, K, chunk)):
                kc = min(chunk, K - k0)
                st, ws, buf = lanes[j % n_streams]
                if persistent:
                    check(lib.ebos_iwe_2dof_slab_batch_f32(*self._compact_ptrs(), ptr(self.key_offsets), self.n, th.data_ptr() + 8 * k0, kc,
                                                           H, W, self.tile[0], self.tile[1], int(halo), int(splits), pad[0], pad[1],
                                                           ptr(ws), nws_al, ptr(buf), 1, int(omit_boundary), out.data_ptr() + 4 * k0,
                                                           None, ptr(self.part_table), st.cuda_stream, None), "ebos_iwe_2dof_slab_batch")
                else:
                    check(lib.ebos_iwe_2dof_slab_f32(ptr(self.x), ptr(self.y), ptr(self.dt), None, *self._compact_ptrs(),
                                                     ptr(self.key_offsets), self.n, th.data_ptr() + 8 * k0, kc, H, W,
                                                     self.tile[0], self.tile[1], int(halo), int(splits), pad[0], pad[1],
                                                     ptr(ws), ws.numel(), ptr(buf), 1, int(omit_boundary),
                                                     out.data_ptr() + 4 * k0, None, ptr(self.part_table), st.cuda_stream),
                          "ebos_iwe_2dof_slab")
            for st, _, _ in lanes[:n_streams]:
                cur.wait_stream(st)
            th.record_stream(cur)
        return out

    def variance_and_grad_dense(self, flow: torch.Tensor, omit_boundary: bool = False, pad: Tuple[int, int] = (0, 0),
                                halo: int = DEFAULT_HALO, splits: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(variance [1], d variance / d flow [2, H, W]) in two launches' worth of Python, without building an autograd
        graph -- for callers that drive their own optimiser (the autograd engine alone costs more per backward() than
        both event kernels take).  Same kernels and results as ``contrast_dense(flow).backward()``."""
        flow32 = _check_flow(self, flow.detach())
        splits = self.resolve_splits(splits)
        pad = (int(pad[0]), int(pad[1]))
        halo = _norm_halo(self, halo)
        if not _slab_ok(self, halo):
            f = flow32.clone().requires_grad_(True)
            v = self.contrast_dense(f, "image_variance", omit_boundary, pad, halo)
            v.backward()
            return v.detach().reshape(1), f.grad
        out, d_flow = _run_dense_job(_dense_job(self, pad, halo, splits, omit_boundary), flow32, True)
        return out[:1], d_flow

    def variance_dense_many(self, flows: torch.Tensor, omit_boundary: bool = False, pad: Tuple[int, int] = (0, 0),
                            halo: int = DEFAULT_HALO, splits: Optional[int] = None, n_streams: int = 3) -> torch.Tensor:
        """Variance contrast of K independent dense-flow hypotheses (the trial loop of a sampler-driven search,
        src/solver/generative_max_likelihood.py:229-236): flows [K, 2, H, W] -> [K].  No gradient.  The evaluations are
        dealt to ``n_streams`` HIP streams with a workspace and an image buffer each, like ``variance_2dof``."""
        lib = _hip.require_gpu()
        if flows.dim() != 4 or tuple(flows.shape[1:]) != (2,) + tuple(self.image_size):
            raise ValueError(f"flows must be [K, 2, {self.image_size[0]}, {self.image_size[1]}], got {tuple(flows.shape)}")
        halo = _norm_halo(self, halo)
        if not _slab_ok(self, halo):
            return torch.stack([self.contrast_dense(f, "image_variance", omit_boundary, pad, halo).detach() for f in flows])
        fl = flows.detach().to(device=self.device, dtype=torch.float32).contiguous()
        K = fl.shape[0]
        H, W = self.image_size
        h, w = H + 2 * pad[0], W + 2 * pad[1]
        splits = self.resolve_splits(splits)
        out = torch.empty(K, dtype=torch.float32, device=self.device)
        n_streams = max(1, min(int(n_streams), K))
        key = ("dense_many", int(halo), int(splits), int(pad[0]), int(pad[1]), h, w)
        lanes = self.__dict__.setdefault("_sweep_lanes", {}).get(key)
        with _hip.on_device(self.device):
            if lanes is None or len(lanes) < n_streams:
                lanes = [(torch.cuda.Stream(device=self.device),
                          torch.zeros(_workspace(self, pad, halo, splits).numel(), dtype=torch.uint8, device=self.device),
                          torch.empty((h, w), dtype=torch.float32, device=self.device)) for _ in range(n_streams)]
                self.__dict__["_sweep_lanes"][key] = lanes
            cur = torch.cuda.current_stream(self.device)
            for st, _, _ in lanes[:n_streams]:
                st.wait_stream(cur)
            for k in range(K):
                st, ws, buf = lanes[k % n_streams]
                check(lib.ebos_iwe_dense_slab_f32(ptr(self.x), ptr(self.y), ptr(self.dt), None, *self._compact_ptrs(),
                                                  ptr(self.key_offsets), self.n, fl.data_ptr() + 8 * H * W * k, H, W, self.tile[0],
                                                  self.tile[1], int(halo), int(splits), pad[0], pad[1], ptr(ws), ws.numel(),
                                                  ptr(buf), 1, int(omit_boundary), out.data_ptr() + 4 * k, None,
                                                  ptr(self.part_table), st.cuda_stream), "ebos_iwe_dense_slab")
            for st, _, _ in lanes[:n_streams]:
                cur.wait_stream(st)
            fl.record_stream(cur)
        return out

    def contrast_dense(self, flow: torch.Tensor, cost: str = "image_variance", omit_boundary: bool = False,
                       pad: Tuple[int, int] = (0, 0), halo: Optional[int] = DEFAULT_HALO,
                       splits: Optional[int] = None, sign: float = 1.0, _eager_checked: bool = False) -> torch.Tensor:
        """Contrast of the IWE under ``flow`` (0-d tensor, raw contrast: callers apply the sign -- or pass ``sign=-1.0``, the
        ``direction="minimize"`` of the cost plugins, and get ``sign * contrast`` with its gradient straight from the kernels).
        Same value and gradient as ``cost(iwe_dense(flow))``, but the variance gradient is folded
        into the backward event kernel (no d_iwe image).

        One stream at a time per plan: the evaluation runs on a job cached on the plan (image, moments and workspace are shared
        between calls), so calls on the same plan from two streams must be ordered by the caller -- independent evaluations
        go through ``variance_dense_many`` / ``variance_2dof``, which give every stream its own buffers.  First order only:
        the gradient is produced with the value, ``backward`` is once-differentiable (``create_graph=True`` raises)."""
        halo = _norm_halo(self, halo)
        if cost not in ("image_variance", "gradient_magnitude"):
            raise KeyError(f"unknown contrast cost {cost!r}")
        if cost == "image_variance" or _slab_ok(self, halo):
            pad2, splits = (int(pad[0]), int(pad[1])), self.resolve_splits(splits)
            if _eager_checked or _eager_ok(self, flow, halo):
                # value and gradient by the one native call, handed back as a tensor whose ``.backward()`` -- when it is called on
                # the result itself, the objective idiom -- stores the gradient without entering the autograd engine (the engine's
                # thread hand-off around a Python backward costs more than both event kernels); any other use of the result
                # attaches the ordinary autograd node first (_EagerLoss)
                # A RAW contrast (sign = 1) is about to be negated by its caller -- contrast is maximised by minimising its negative:
                # the kernels produce -d contrast / d flow, and ``(-loss).backward()`` finds its factor already applied; negating the
                # 7.4 MB gradient afterwards is a launch of its own (55 -> 50 us per forward + backward at 10 M events).  A caller
                # that does differentiate the raw contrast pays that launch instead (_EagerLoss applies what is left to apply).
                sign = float(sign)
                spec = -1.0 if sign == 1.0 else sign
                out, d_flow = _run_dense_job(_dense_job(self, pad2, halo, splits, bool(omit_boundary)), flow, True, cost, spec)
                return _EagerLoss.wrap(out[1] if sign != 1.0 else out[0], flow, d_flow, sign, applied=spec)
            v = _FusedVarianceDense.apply(flow, self, pad2, bool(omit_boundary), halo, splits, cost)
            return v if sign == 1.0 else v * sign
        v = ops.gradient_magnitude(self.iwe_dense(flow, pad=pad, halo=halo, splits=splits), omit_boundary)
        return v if sign == 1.0 else v * sign


# ----------------------------------------------------------------------------------------------
def _build_lean(source: int, events, raw, image_size, direction, normalize_t, tile, ticks_per_second, deferred):
    """``ebos_plan_lean``: compact plan straight from the window.  Returns None when the window cannot take it (fractional
    source coordinates, a geometry outside the LDS sort) -- the caller then runs the full build."""
    lib = _hip.require_gpu()
    H, W = int(image_size[0]), int(image_size[1])
    th, tw = int(tile[0]), int(tile[1])
    if th > 256 or tw > 256:
        return None
    ref_mode, frac = parse_direction(direction)
    if raw is None:
        dev, n = events.device, int(events.shape[0])
        col = row = t = None
    else:
        col, row, t = raw
        dev, n = t.device, int(t.shape[0])
    tiles_y, tiles_x = (H + th - 1) // th, (W + tw - 1) // tw
    n_tiles, n_keys = tiles_y * tiles_x, tiles_y * tiles_x * th * tw
    nbytes = int(lib.ebos_plan_lean_scratch_bytes(n, H, W, th, tw))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    cap = n + 3 * n_tiles + 8
    key_offsets = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    grp_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    cpix = torch.empty(cap, dtype=torch.int16, device=dev)
    cdt = torch.empty(cap, dtype=torch.float32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    part_table = torch.empty(5 * n_tiles + 1, dtype=torch.int32, device=dev)
    with _hip.on_device(dev):
        rc = lib.ebos_plan_lean(source, ptr(events), ptr(col), ptr(row), ptr(t), float(ticks_per_second), n, ref_mode, frac,
                                int(normalize_t), H, W, th, tw, ptr(key_offsets), ptr(grp_offsets), ptr(cpix), ptr(cdt), cap,
                                ptr(counts), None, ptr(scratch), nbytes, stream_ptr())
        if rc == -3:  # EBOS_ERR_UNSUPPORTED: geometry outside the LDS sort
            return None
        check(rc, "ebos_plan_lean")
        check(lib.ebos_plan_parts(ptr(key_offsets), H, W, th, tw, _n_cu(dev), part_fixed_events(n, n_tiles, _n_cu(dev)), ptr(part_table),
                                  stream_ptr()), "ebos_plan_parts")
    dropped, used, fullest = 0, None, None
    if not deferred:  # the one host read-back of the build: (outside the image, fractional sources, work items in use, fullest tile)
        dropped, fractional, used, fullest = _plan_facts(lib, key_offsets, H, W, th, tw, counts, part_table)
        if fractional:
            return None  # fractional (undistorted) source coordinates: the (x, y, dt) format of the full build
    plan = EventPlan(None, None, None, None, (H, W), n - dropped, n, (th, tw), key_offsets, None, dropped, grp_offsets, cpix, cdt,
                     part_table, dt_bound_for(direction, normalize_t))
    plan.__dict__["_counts"], plan.__dict__["_deferred"], plan.__dict__["_parts_used"] = counts, bool(deferred), used
    plan.__dict__["_fullest_tile"] = fullest
    return plan


def _plan_facts(lib, key_offsets: torch.Tensor, H: int, W: int, th: int, tw: int, counts: torch.Tensor, part_table: torch.Tensor):
    """(outside the image, fractional sources, work items in use, events of the fullest tile): ``ebos_plan_facts`` + one 16-byte copy."""
    facts = torch.empty(4, dtype=torch.int32, device=key_offsets.device)
    with _hip.on_device(key_offsets.device):
        check(lib.ebos_plan_facts(ptr(key_offsets), H, W, th, tw, ptr(counts), ptr(part_table), ptr(facts), stream_ptr()), "ebos_plan_facts")
    return tuple(int(v) for v in facts.tolist())


def _check_flow(plan: EventPlan, flow: torch.Tensor) -> torch.Tensor:
    H, W = plan.image_size
    if tuple(flow.shape) != (2, H, W):
        raise ValueError(f"flow must be [2, {H}, {W}], got {tuple(flow.shape)}")
    if flow.device != plan.device:
        raise _hip.HipUnavailableError(f"flow is on {flow.device}, the event plan on {plan.device}")
    return flow.to(torch.float32).contiguous()


_SLAB_CONFIGS = None


def _slab_ok(plan: EventPlan, halo) -> bool:
    global _SLAB_CONFIGS
    if not plan.binned or halo is None:
        return False
    if _SLAB_CONFIGS is None:
        _SLAB_CONFIGS = set(_hip.slab_configs())
    if halo == "auto":
        return any((th, tw) == tuple(plan.tile) for th, tw, _ in _SLAB_CONFIGS)
    return (plan.tile[0], plan.tile[1], _max_halo(int(halo))) in _SLAB_CONFIGS


def _refuse_deferred(plan: EventPlan, what: str) -> None:
    """A deferred plan keeps ``n`` as an upper bound and zero-fills the tail of its SoA arrays: only the tile-range
    (slab / tiled) kernels skip that tail, the general kernels would read it as events at (0, 0)."""
    if plan.__dict__.get("_deferred"):
        raise NotImplementedError(f"{what}: a plan built with deferred=True only runs on the tile-private kernels "
                                  "(a built tile / halo configuration); rebuild it with deferred=False for the general path")
    if plan.lean:
        raise NotImplementedError(f"{what}: a lean plan (emit='compact') holds no SoA events: it only runs on the tile-private "
                                  "kernels (a built tile / halo configuration) with unit weights; build it with emit='full'")


def _workspace(plan: EventPlan, pad, halo, splits) -> torch.Tensor:
    """Zero-filled once; the kernels keep the spill section zero between calls."""
    lib = _hip.require_gpu()
    key = (int(halo), int(splits), int(pad[0]), int(pad[1]))
    cache = plan.__dict__.setdefault("_workspaces", {})
    if key not in cache:
        H, W = plan.image_size
        nbytes = int(lib.ebos_iwe_slab_workspace_bytes(H, W, plan.tile[0], plan.tile[1], key[0], key[1], key[2], key[3]))
        cache[key] = torch.zeros(nbytes, dtype=torch.uint8, device=plan.device)
    return cache[key]


class SlabBatch(object):
    """Independent windows of one geometry evaluated by ``ebos_iwe_slab_batch_f32``: accumulate, combine and finalize each run as
    one launch over (work item, window), 16 windows at a time -- the time windows of bos_event.py:144-220 when they are thin
    (BASELINE configs[3]: 2 M events), where per-launch fixed work and the gaps between launches dominate.

        batch = SlabBatch(plans, flows, patch=((24, 32), (24, 32)))   # flows: patch grids [2, gh, gw]; patch=None: dense [2, H, W]
        batch.run()                      # -> variances [n] (device), batch.iwes [n, h, w]
    The array of window descriptors is filled once; ``run`` is ONE C call.  Results are bit-identical to per-window calls."""

    def __init__(self, plans, flows, patch=None, pad=(0, 0), halo: int = DEFAULT_HALO, splits: Optional[int] = None,
                 omit_boundary: bool = False):
        lib = _hip.require_gpu()
        if not plans:
            raise ValueError("SlabBatch: no windows")
        p0 = plans[0]
        H, W = p0.image_size
        self.plans, self.flows = list(plans), [f.contiguous().float() for f in flows]
        if halo == "auto":  # one |dt| bound for the whole batch: the largest of the plans' (a plan without one: the built halo)
            bounds = [pl.dt_bound for pl in plans]
            if any(b is None for b in bounds):
                halo = max(hl for th, tw, hl in _hip.slab_configs() if (th, tw) == tuple(p0.tile))
            else:
                halo = int(lib.ebos_halo_auto(max(hl for th, tw, hl in _hip.slab_configs() if (th, tw) == tuple(p0.tile)), float(max(bounds))))
        self.pad, self.halo, self.omit = (int(pad[0]), int(pad[1])), int(halo), bool(omit_boundary)
        self.splits = p0.resolve_splits(splits)
        for pl, fl in zip(self.plans, self.flows):
            if pl.image_size != p0.image_size or pl.tile != p0.tile or not pl.compact or pl.device != p0.device:
                raise ValueError("SlabBatch: windows must share image size, tile and device, and be compact (unit-weight) plans")
            if pl.resolve_splits(splits) != self.splits:
                raise ValueError("SlabBatch: windows must share the work-item mode (all adaptive or all with the same split count)")
        if patch is None:
            self.grid = (0, 0, 0, 0, 0, 0)
            want = (2, H, W)
        else:
            (ph, pw), (sh, sw) = patch
            gh, gw = self.flows[0].shape[-2:]
            self.grid = (int(gh), int(gw), int(ph), int(pw), int(sh), int(sw))
            want = (2, gh, gw)
        for fl in self.flows:
            if tuple(fl.shape) != want:
                raise ValueError(f"SlabBatch: flow of shape {tuple(fl.shape)}, expected {want}")
        n = len(self.plans)
        dev = p0.device
        self.nws = int(lib.ebos_iwe_slab_workspace_bytes(H, W, p0.tile[0], p0.tile[1], self.halo, self.splits, *self.pad))
        self.workspaces = torch.zeros((n, (self.nws + 255) // 256 * 256), dtype=torch.uint8, device=dev)
        self.iwes = torch.empty((n, H + 2 * self.pad[0], W + 2 * self.pad[1]), dtype=torch.float32, device=dev)
        self.variances = torch.empty(n, dtype=torch.float32, device=dev)
        self.moments = torch.empty((n, 2), dtype=torch.float64, device=dev)
        self.windows = (_hip.SlabWindow * n)()
        P = lambda t: None if t is None else t.data_ptr()
        for k, (pl, fl) in enumerate(zip(self.plans, self.flows)):
            w = self.windows[k]
            cp = pl._compact_ptrs()
            w.grp_offsets, w.cpix, w.cdt = cp[0], cp[1], cp[2]
            w.key_offsets, w.part_table, w.flow = P(pl.key_offsets), P(pl.part_table) if self.splits == 0 else None, P(fl)
            w.workspace, w.iwe = self.workspaces[k].data_ptr(), self.iwes[k].data_ptr()
            w.out_variance, w.moments = self.variances.data_ptr() + 4 * k, self.moments[k].data_ptr()
        self._lib, self._tile, self._size = lib, p0.tile, (H, W)

    def run(self, want_variance: bool = True, stream: Optional[int] = None, tail_stream: Optional[int] = None) -> torch.Tensor:
        """``tail_stream`` (a raw HIP stream of the caller's): the combine / finalize passes of each 16 windows run there, beside
        the accumulate pass of the next 16; the results are ordered on ``stream`` when the call returns either way."""
        H, W = self._size
        _hip.check(self._lib.ebos_iwe_slab_batch_f32(self.windows, len(self.plans), *self.grid, H, W, self._tile[0], self._tile[1],
                                                     self.halo, self.splits, self.pad[0], self.pad[1], self.workspaces.shape[1],
                                                     1 if want_variance else 0, 1 if self.omit else 0,
                                                     _hip.stream_ptr() if stream is None else stream, tail_stream),
                   "ebos_iwe_slab_batch")
        return self.variances


class _DenseJob(object):
    """``ebos_dense_job`` of one (plan, padding, halo, splits, omit_boundary) with the buffers it points at: filled once,
    after which an objective (+ gradient) evaluation is ONE C call with six arguments (``ebos_variance_dense_job_f32``)."""

    def __init__(self, plan: EventPlan, pad, halo, splits, omit):
        import ctypes as C

        H, W = plan.image_size
        self.ws = _workspace(plan, pad, halo, splits)
        self.iwe = torch.empty((H + 2 * pad[0], W + 2 * pad[1]), dtype=torch.float32, device=plan.device)
        self.moments = torch.empty((1, 2), dtype=torch.float64, device=plan.device)
        g, c, d = plan._compact_ptrs()
        self.struct = _hip.DenseJob(ptr(plan.x), ptr(plan.y), ptr(plan.dt), g, c, d, ptr(plan.key_offsets), plan.n, H, W,
                                    plan.tile[0], plan.tile[1], int(halo), int(splits), pad[0], pad[1], int(omit), ptr(self.ws),
                                    self.ws.numel(), ptr(plan.part_table), ptr(self.iwe), ptr(self.moments))
        self.ref = C.addressof(self.struct)
        self.index = plan.device.index
        self._gm = None

    def gm_buffers(self):
        """Scratch of the gradient-magnitude job: the gradient image and the Sobel pass's value partials (allocated on first use)."""
        if self._gm is None:
            lib = _hip.require_gpu()
            h, w = self.iwe.shape
            n = int(lib.ebos_gradient_magnitude_fused_partials(h, w))
            self._gm = (torch.empty_like(self.iwe), torch.empty(n, dtype=torch.float64, device=self.iwe.device), n)
        return self._gm


def _dense_job(plan: EventPlan, pad, halo, splits, omit) -> _DenseJob:
    key = ("job", int(pad[0]), int(pad[1]), int(halo), int(splits), bool(omit))
    cache = plan.__dict__.setdefault("_jobs", {})
    job = cache.get(key)
    if job is None:
        job = cache[key] = _DenseJob(plan, pad, halo, splits, omit)
    return job


_SIGNS = {}


def _sign_constant(device: torch.device, sign: float) -> torch.Tensor:
    """Device-resident f32 [1] holding ``sign`` (the ``upstream`` of a signed job), made once per (device, value)."""
    key = (device, float(sign))
    t = _SIGNS.get(key)
    if t is None:
        t = _SIGNS[key] = torch.full((1,), float(sign), dtype=torch.float32, device=device)
    return t


def _run_dense_job(job: _DenseJob, flow32: torch.Tensor, want_grad: bool, cost: str = "image_variance", sign: float = 1.0):
    """(out [2] = (contrast, sign * contrast), sign * d contrast / d flow [2, H, W] | None) in one native call: the variance (three
    launches) or the gradient magnitude (four: one Sobel pass yields the value partials and the gradient image,
    ``ebos_gradient_magnitude_dense_job_f32``).  ``sign`` (the direction of a cost) is applied INSIDE the kernels: no launch negates
    a scalar or scales the gradient field afterwards; without a gradient only out[0] is written."""
    lib = _hip.require_gpu()
    out = torch.empty(2, dtype=torch.float32, device=flow32.device)
    d_flow = torch.empty_like(flow32) if want_grad else None
    up = None if sign == 1.0 else _sign_constant(flow32.device, sign).data_ptr()
    scaled = out.data_ptr() + 4 if want_grad else None

    def call():
        if cost == "image_variance":
            return lib.ebos_variance_dense_job_signed_f32(job.ref, flow32.data_ptr(), out.data_ptr(), scaled, up, ptr(d_flow), stream_ptr())
        d_iwe, partials, n = job.gm_buffers()
        return lib.ebos_gradient_magnitude_dense_job_f32(job.ref, flow32.data_ptr(), out.data_ptr(), scaled, up, ptr(d_flow),
                                                         d_iwe.data_ptr(), partials.data_ptr(), n, stream_ptr())

    if _hip.current_device_index() == job.index:
        rc = call()
    else:
        with _hip.on_device(flow32.device):
            rc = call()
    check(rc, "ebos_variance_dense_job" if cost == "image_variance" else "ebos_gradient_magnitude_dense_job")
    return out, d_flow


def _launch_iwe_dense_slab(plan: EventPlan, flow32, weight, pad, halo, splits, want_variance=False, omit=False):
    """Tile-private forward: returns (iwe, variance [1] | None, moments [1, 2] | None)."""
    lib = _hip.require_gpu()
    H, W = plan.image_size
    ws = _workspace(plan, pad, halo, splits)
    iwe = torch.empty((H + 2 * pad[0], W + 2 * pad[1]), dtype=torch.float32, device=plan.device)
    out = torch.empty(1, dtype=torch.float32, device=plan.device) if want_variance else None
    moments = torch.empty((1, 2), dtype=torch.float64, device=plan.device) if want_variance else None
    with _hip.on_device(plan.device):
        check(lib.ebos_iwe_dense_slab_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(weight), *plan._compact_ptrs(),
                                          ptr(plan.key_offsets), plan.n,
                                          ptr(flow32), H, W, plan.tile[0], plan.tile[1], int(halo), int(splits), pad[0],
                                          pad[1], ptr(ws), ws.numel(), ptr(iwe), int(want_variance), int(omit), ptr(out),
                                          ptr(moments), ptr(plan.part_table), stream_ptr()), "ebos_iwe_dense_slab")
    return iwe, out, moments


def _launch_iwe_dense(plan: EventPlan, flow32: torch.Tensor, weight, pad, halo, splits) -> torch.Tensor:
    lib = _hip.require_gpu()
    H, W = plan.image_size
    if _slab_ok(plan, halo):
        return _launch_iwe_dense_slab(plan, flow32, weight, pad, halo, splits)[0]
    _refuse_deferred(plan, "iwe_dense")
    iwe = torch.zeros((H + 2 * pad[0], W + 2 * pad[1]), dtype=torch.float32, device=plan.device)
    with _hip.on_device(plan.device):
        if plan.binned and halo is not None:
            check(lib.ebos_iwe_dense_tiled_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(weight), ptr(plan.key_offsets),
                                               plan.n, ptr(flow32), H, W, plan.tile[0], plan.tile[1], int(halo), max(1, splits),
                                               pad[0], pad[1], ptr(iwe), stream_ptr()), "ebos_iwe_dense_tiled")
        else:
            check(lib.ebos_iwe_dense_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(weight), plan.n, ptr(flow32), H, W,
                                         W, pad[0], pad[1], ptr(iwe), stream_ptr()), "ebos_iwe_dense")
    return iwe


def _plan_weight(plan: EventPlan, weight: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Per-event weights are given in INPUT order; reorder them like the events."""
    if weight is None:
        return None
    if plan.__dict__.get("_deferred"):
        # a deferred plan does not know how many events it kept: the tail of `perm` is not defined
        raise NotImplementedError("per-event weights need a plan built with the host read-back (deferred=False)")
    if plan.lean:
        raise NotImplementedError("per-event weights need the SoA events and the permutation: build the plan with emit='full'")
    w = weight.to(device=plan.device, dtype=torch.float32).reshape(-1)
    if w.numel() != plan.n_input:
        raise ValueError(f"weight must have one entry per input event ({plan.n_input}), got {w.numel()}")
    w = w if plan.perm is None else w[plan.perm.long()]
    out = torch.zeros((plan.n + 3) // 4 * 4 + 4, dtype=torch.float32, device=plan.device)  # 16-byte loads: padded
    out[:plan.n] = w
    return out[:plan.n]


def _launch_dense_bwd(plan, flow32, weight_p, pad, g_image, affine, g_lo, want_dweight, halo=DEFAULT_HALO,
                      var_moments=None, upstream=None, splits=1):
    lib = _hip.require_gpu()
    H, W = plan.image_size
    d_w = torch.empty(plan.n, dtype=torch.float32, device=plan.device) if want_dweight else None
    if _slab_ok(plan, halo):  # tile-private backward: d_flow written with plain stores, no zero-fill
        d_flow = torch.empty((2, H, W), dtype=torch.float32, device=plan.device)
        adaptive = splits == 0 and plan.part_table is not None
        ws = _workspace(plan, pad, halo, 0) if adaptive else None
        with _hip.on_device(plan.device):
            check(lib.ebos_iwe_dense_tiled_bwd_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(weight_p), *plan._compact_ptrs(),
                                                   ptr(plan.key_offsets), plan.n, ptr(flow32), H, W, plan.tile[0], plan.tile[1], int(halo), pad[0],
                                                   pad[1], ptr(g_image), ptr(affine), g_lo, ptr(d_flow), ptr(d_w),
                                                   ptr(var_moments), ptr(upstream), None, ptr(ws), ws.numel() if adaptive else 0,
                                                   ptr(plan.part_table) if adaptive else None, stream_ptr()),
                  "ebos_iwe_dense_tiled_bwd")
        return d_flow, d_w
    _refuse_deferred(plan, "iwe_dense backward")
    if var_moments is not None:  # general kernels take the affine form
        affine = torch.empty(2, dtype=torch.float32, device=plan.device)
        with _hip.on_device(plan.device):
            check(lib.ebos_image_variance_affine_f32(ptr(var_moments), ptr(upstream), 1, ptr(affine), stream_ptr()),
                  "ebos_image_variance_affine")
    d_flow = torch.zeros((2, H, W), dtype=torch.float32, device=plan.device)
    with _hip.on_device(plan.device):
        check(lib.ebos_iwe_dense_bwd_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(weight_p), plan.n, ptr(flow32), H, W,
                                         W, pad[0], pad[1], ptr(g_image), ptr(affine), g_lo, int(plan.binned),
                                         ptr(d_flow), ptr(d_w), stream_ptr()), "ebos_iwe_dense_bwd")
    return d_flow, d_w


def _unpermute(plan: EventPlan, d_w: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(plan.n_input, dtype=d_w.dtype, device=d_w.device)
    if plan.perm is None:
        out.copy_(d_w)
    else:
        out[plan.perm.long()] = d_w
    return out


class _FusedIweDense(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, weight, plan, pad, halo, splits):
        flow32 = _check_flow(plan, flow)
        wp = _plan_weight(plan, weight)
        iwe = _launch_iwe_dense(plan, flow32, wp, pad, halo, splits)
        ctx.save_for_backward(flow32, wp if wp is not None else torch.empty(0))
        ctx.meta = (plan, pad, flow.dtype, weight.dtype if weight is not None else None, halo, splits)
        return iwe if flow.dtype == torch.float32 else iwe.to(flow.dtype)

    @staticmethod
    def backward(ctx, g):
        flow32, wp = ctx.saved_tensors
        plan, pad, fdt, wdt, halo, splits = ctx.meta
        wp = wp if wdt is not None else None
        need_w = wdt is not None and ctx.needs_input_grad[1]
        d_flow, d_w = _launch_dense_bwd(plan, flow32, wp, pad, g.to(torch.float32).contiguous(), None, 0, need_w, halo,
                                        splits=splits)
        d_weight = _unpermute(plan, d_w).to(wdt) if need_w else None
        return (d_flow.to(fdt) if ctx.needs_input_grad[0] else None), d_weight, None, None, None, None


def _eager_ok(plan: EventPlan, flow: torch.Tensor, halo) -> bool:
    """The objective idiom's short cut applies to a float32 leaf of the plan's own shape and device that wants its gradient, has no
    tensor hooks (they fire inside the engine), with grad mode on, on a tile-private configuration."""
    return (type(flow) is torch.Tensor or type(flow) is torch.nn.Parameter) and flow.requires_grad and flow.is_leaf and \
        torch.is_grad_enabled() and flow.dtype == torch.float32 and flow.is_contiguous() and flow.device == plan.device and \
        tuple(flow.shape) == (2,) + tuple(plan.image_size) and not flow._backward_hooks and \
        not getattr(flow, "_post_accumulate_grad_hooks", None) and _slab_ok(plan, halo)


class _AttachGrad(torch.autograd.Function):
    """The autograd node of an already computed (value, gradient) pair: what _EagerLoss turns into when it is used as anything
    but the direct target of ``backward()``.  ``scale``: the Python factor the value has been multiplied with since."""

    @staticmethod
    def forward(ctx, flow, value, d_flow, scale):
        ctx.save_for_backward(d_flow)
        ctx.scale = scale
        return value.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (d_flow,) = ctx.saved_tensors
        return d_flow * (g.to(torch.float32) * ctx.scale), None, None, None


class _ConsumedGrad(torch.autograd.Function):
    """What an _EagerLoss turns into once ``backward()`` has handed its gradient over: an ordinary tensor that still prints,
    copies, compares and enters arithmetic like any result whose graph has been freed -- only a SECOND differentiation raises,
    with the engine's own wording."""

    @staticmethod
    def forward(ctx, flow, value):
        return value.detach()

    @staticmethod
    def backward(ctx, g):
        raise RuntimeError("Trying to backward through the graph a second time: contrast_dense produced its gradient with the "
                           "value and backward() has already handed it to flow.grad. Evaluate the objective again (it is one "
                           "native call), or pass retain_graph=True to the first backward().")


class _EagerLoss(torch.Tensor):
    """0-d result of ``contrast_dense`` whose gradient w.r.t. the (leaf) flow exists already.

    ``loss.backward()`` on the object itself -- or on ``-loss`` / ``k * loss`` / ``loss / k`` with a Python number, the sign and
    weight a caller applies to a contrast -- accumulates that gradient into ``flow.grad`` exactly as the engine would (set when
    ``None``, added otherwise; scaled by ``gradient`` when one is given) without running the engine.  Every other use -- arithmetic
    with tensors, ``torch.*`` functions, ``grad_fn`` / ``requires_grad`` queries, ``backward`` with ``inputs`` / ``create_graph``
    -- first attaches the ordinary autograd node (_AttachGrad) and proceeds on that tensor, so results and graphs are those of
    ``_FusedVarianceDense``.  ``item()`` / ``detach()`` / ``float()`` / f-strings / ``print`` / comparisons read the value.
    ``torch.autograd.grad(loss, flow)`` dispatches the same way.  One entry point does not consult ``__torch_function__``: the
    FUNCTION ``torch.autograd.backward([loss])`` on the bare result raises ("does not require grad") instead of running -- use the
    method, or any expression of the result beyond a sign / Python weight (``loss + 0.0``).

    A result and its scaled descendants (``-loss``, ``k * loss``) share ONE gradient cell: the first ``backward()`` among them
    consumes it for all (as the engine frees the graph they would share), after which each of them is an ordinary tensor whose
    second differentiation raises (_ConsumedGrad) and whose every other use -- ``print``, ``cpu``, ``clone``, arithmetic for
    logging -- works."""

    _VALUE_ONLY = {"item", "detach", "__float__", "__format__", "tolist", "__bool__", "__int__", "dim", "size", "numel", "__len__",
                   "__repr__", "__str__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "lt", "le", "gt", "ge", "eq",
                   "ne", "isnan", "isfinite", "isinf"}

    @staticmethod
    def wrap(value: torch.Tensor, flow: torch.Tensor, d_flow: torch.Tensor, scale: float = 1.0, cell=None,
             applied: float = 1.0) -> "_EagerLoss":
        """``value`` = scale x contrast; ``d_flow`` = applied x d contrast / d flow (the kernels apply a cost's sign themselves)."""
        t = torch.Tensor._make_subclass(_EagerLoss, value)
        # flow, shared cell [gradient (None once handed over), the factor it already carries], attached tensor, factor of this object
        t._ebos = [flow, cell if cell is not None else [d_flow, float(applied)], None, scale]
        return t

    def _plain(self) -> torch.Tensor:
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    def _attached(self) -> torch.Tensor:
        st = self._ebos
        if st[2] is None:
            if st[1][0] is None:  # consumed (by this object or a scaled relative): an ordinary tensor with a freed graph
                st[2] = _ConsumedGrad.apply(st[0], self._plain())
            else:
                st[2] = _AttachGrad.apply(st[0], self._plain(), st[1][0], st[3] / st[1][1])
        return st[2]

    def _scaled(self, k) -> "_EagerLoss":
        st = self._ebos
        if st[2] is not None or st[1][0] is None:  # already a graph node (or consumed): the ordinary path
            return self._attached() * k
        return _EagerLoss.wrap(self._plain() * k, st[0], None, st[3] * float(k), cell=st[1])

    def __neg__(self):
        return self._scaled(-1.0)

    def __mul__(self, other):
        return self._scaled(other) if isinstance(other, (int, float)) and not isinstance(other, bool) else self._attached() * other

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self._scaled(1.0 / other) if isinstance(other, (int, float)) and not isinstance(other, bool) and other != 0 \
            else self._attached() / other

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        st = self._ebos
        flow, cell = st[0], st[1]
        if st[2] is not None or create_graph or inputs is not None or cell[0] is None or flow._backward_hooks or \
                getattr(flow, "_post_accumulate_grad_hooks", None):
            return self._attached().backward(gradient, retain_graph, create_graph, inputs)
        g, k = cell[0], st[3] / cell[1]  # (what is left to apply: nothing when the kernels applied this object's factor)
        with torch.no_grad():
            if gradient is not None:
                f = gradient.to(device=g.device, dtype=torch.float32) * k
            else:
                f = None if k == 1.0 else k
            if retain_graph:
                g = g * f if f is not None else g.clone()  # the cell keeps the unscaled gradient for the next backward()
            else:
                cell[0] = None  # consumed for this object AND its scaled relatives; the buffer now belongs to flow.grad
                if f is not None:
                    g.mul_(f)
            if flow.grad is None:
                flow.grad = g
            else:
                flow.grad.add_(g)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        value_only = getattr(func, "__name__", "") in cls._VALUE_ONLY
        conv = (lambda a: a._plain()) if value_only else (lambda a: a._attached())
        args, kwargs = _pytree.tree_map_only(_EagerLoss, conv, (tuple(args), kwargs))  # (also inside lists: torch.stack([loss, ...]))
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class _FusedVarianceDense(torch.autograd.Function):
    """var(IWE(flow)).  On the tile-private configurations the gradient is produced WITH the value (one native call:
    accumulate, combine, finalize, tile-private backward with unit upstream) whenever the flow requires grad, and
    ``backward`` only scales it -- the interpreter and the autograd engine, not the kernels, bound this path (272 us
    against 65 us of kernels per fwd + bwd at 10 M events before)."""

    @staticmethod
    def forward(ctx, flow, plan, pad, omit, halo, splits, cost="image_variance"):
        lib = _hip.require_gpu()
        fast = flow.dtype == torch.float32 and flow.is_contiguous() and flow.device == plan.device and \
            tuple(flow.shape) == (2,) + tuple(plan.image_size)
        flow32 = flow if fast else _check_flow(plan, flow)
        ctx.fdt = flow.dtype
        if _slab_ok(plan, halo):  # IWE + contrast (+ gradient) in one tile-private pipeline
            job = _dense_job(plan, pad, halo, splits, omit)
            out, d_flow = _run_dense_job(job, flow32, ctx.needs_input_grad[0], cost)
            ctx.eager = True
            if d_flow is not None:
                ctx.save_for_backward(d_flow)
            return out[0] if flow.dtype == torch.float32 else out[0].to(flow.dtype)
        ctx.eager = False
        iwe = _launch_iwe_dense(plan, flow32, None, pad, halo, splits)
        h, w = iwe.shape
        out = torch.empty(1, dtype=torch.float32, device=plan.device)
        moments = torch.empty((1, 2), dtype=torch.float64, device=plan.device)
        nbytes = int(lib.ebos_cost_scratch_bytes(1))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=plan.device)
        with _hip.on_device(plan.device):
            check(lib.ebos_image_variance_f32(ptr(iwe), 1, h, w, int(omit), ptr(out), ptr(moments), ptr(scratch), nbytes,
                                              stream_ptr()), "ebos_image_variance")
        ctx.save_for_backward(flow32, iwe, moments)
        ctx.meta = (plan, pad, int(omit), halo, splits)
        return out[0].to(flow.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable  # the saved gradient is a constant: create_graph must raise, not drop terms
    def backward(ctx, g):
        if ctx.eager:
            (d_flow,) = ctx.saved_tensors
            d = d_flow * g.to(torch.float32)
            return (d if ctx.fdt == torch.float32 else d.to(ctx.fdt)), None, None, None, None, None, None
        flow32, iwe, moments = ctx.saved_tensors
        plan, pad, omit, halo, splits = ctx.meta
        up = g.to(torch.float32).reshape(1).contiguous()
        d_flow, _ = _launch_dense_bwd(plan, flow32, None, pad, iwe, None, omit, False, halo, moments, up, splits=splits)
        return d_flow.to(ctx.fdt), None, None, None, None, None, None


class _FusedIwe2Dof(torch.autograd.Function):
    @staticmethod
    def forward(ctx, thetas, weight, plan, pad, halo, splits):
        lib = _hip.require_gpu()
        if thetas.dim() != 2 or thetas.shape[1] != 2:
            raise ValueError(f"thetas must be [K, 2], got {tuple(thetas.shape)}")
        th32 = thetas.to(device=plan.device, dtype=torch.float32).contiguous()
        wp = _plan_weight(plan, weight)
        K = th32.shape[0]
        H, W = plan.image_size
        h, w = H + 2 * pad[0], W + 2 * pad[1]
        with _hip.on_device(plan.device):
            if _slab_ok(plan, halo):
                ws = _workspace(plan, pad, halo, splits)
                iwes = torch.empty((K, h, w), dtype=torch.float32, device=plan.device)
                check(lib.ebos_iwe_2dof_slab_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(wp), *plan._compact_ptrs(),
                                                 ptr(plan.key_offsets), plan.n, ptr(th32), K, H, W, plan.tile[0],
                                                 plan.tile[1], int(halo), int(splits), pad[0], pad[1], ptr(ws), ws.numel(),
                                                 ptr(iwes), 0, 0, None, None, ptr(plan.part_table), stream_ptr()),
                      "ebos_iwe_2dof_slab")
            else:
                _refuse_deferred(plan, "iwe_2dof")
                iwes = torch.zeros((K, h, w), dtype=torch.float32, device=plan.device)
                check(lib.ebos_iwe_2dof_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(wp), plan.n, ptr(th32), K, h, w,
                                            pad[0], pad[1], ptr(iwes), stream_ptr()), "ebos_iwe_2dof")
        ctx.save_for_backward(th32, wp if wp is not None else torch.empty(0))
        ctx.meta = (plan, pad, thetas.dtype, thetas.device, weight is not None, halo, splits)
        return iwes if thetas.dtype == torch.float32 else iwes.to(thetas.dtype)

    @staticmethod
    def backward(ctx, g):
        lib = _hip.require_gpu()
        th32, wp = ctx.saved_tensors
        plan, pad, tdt, tdev, has_w, halo, splits = ctx.meta
        K = th32.shape[0]
        H, W = plan.image_size
        h, w = H + 2 * pad[0], W + 2 * pad[1]
        g32 = g.to(torch.float32).contiguous()
        d_th = torch.zeros((K, 2), dtype=torch.float32, device=plan.device)
        with _hip.on_device(plan.device):
            if _slab_ok(plan, halo):
                ws = _workspace(plan, pad, halo, splits)
                check(lib.ebos_iwe_2dof_tiled_bwd_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(wp) if has_w else None,
                                                      *plan._compact_ptrs(), ptr(plan.key_offsets), plan.n, ptr(th32), K, H, W,
                                                      plan.tile[0], plan.tile[1], int(halo), pad[0], pad[1], ptr(g32), None, 0,
                                                      ptr(d_th), ptr(ws), ws.numel(), stream_ptr()), "ebos_iwe_2dof_tiled_bwd")
                return d_th.to(device=tdev, dtype=tdt), None, None, None, None, None
            check(lib.ebos_iwe_2dof_bwd_f32(ptr(plan.x), ptr(plan.y), ptr(plan.dt), ptr(wp) if has_w else None, plan.n,
                                            ptr(th32), K, h, w, pad[0], pad[1], ptr(g32), None, 0, ptr(d_th),
                                            stream_ptr()), "ebos_iwe_2dof_bwd")
        return d_th.to(device=tdev, dtype=tdt), None, None, None, None, None
