/*
 * ebos_hip.h -- C ABI of libebos_hip.so: the MI355X (gfx950) implementation of the
 * contrast-maximisation inner loop of event-based BOS
 *        warp events -> bilinear-splat image of warped events (IWE) -> contrast cost (+ gradients)
 *
 * The reference (tub-rip/event_based_bos) is pure Python and has no FFI for this path; its
 * boundary is the plugin surface Warp / EventImageConverter / costs.  Every
 * entry point below names the reference code (file:line under /root/reference) that it replaces.
 * The Python host package event_based_bos_amd binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer unless it says "host";
 *  - nothing is allocated, freed or synchronised in here: the caller owns every buffer (torch's
 *    caching allocator in practice) and every call is asynchronous on `stream` (a hipStream_t
 *    passed as void*; NULL = the default stream).  All entry points are graph-capturable;
 *  - return value: EBOS_OK (0) or a negative ebos_status; ebos_last_error() gives the message
 *    (thread-local, valid until the next failing call on that thread);
 *  - event = (x, y, t, p) with x = ROW coordinate and y = COLUMN coordinate
 *    (src/data_loader/ccs.py:293-296); images are [H, W] row-major; flow is [2, H, W] with
 *    channel 0 = row displacement (src/warp.py:333-336);
 *  - suffix _f32 / _f64 = element type of events, flow and images.  The *_soa_* hot path is f32.
 *  - "accumulates": the kernel ADDS into the output; the caller zeroes it (hipMemsetAsync).
 */
#ifndef EBOS_HIP_H
#define EBOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBOS_ABI_VERSION 2

typedef void* ebos_stream_t; /* hipStream_t */

typedef enum ebos_status {
  EBOS_OK = 0,
  EBOS_ERR_INVALID_ARG = -1, /* null pointer, negative size, unsupported enum value       */
  EBOS_ERR_LAUNCH = -2,      /* hipLaunchKernel / hipMemsetAsync reported an error          */
  EBOS_ERR_UNSUPPORTED = -3, /* valid request this build has no kernel for                  */
  EBOS_ERR_SCRATCH = -4      /* caller-provided scratch buffer too small                    */
} ebos_status;

/* reference-time modes, src/warp.py:245-259 */
typedef enum ebos_reftime_mode {
  EBOS_REF_FIRST = 0,    /* t_ref = min t                                     (:248-249) */
  EBOS_REF_LAST = 1,     /* t_ref = max t                                     (:252-253) */
  EBOS_REF_FRACTION = 2, /* t_ref = tmin + (tmax - tmin) * fraction  (float/middle/before/after/random, :245-247) */
  EBOS_REF_TIMEBASE = 3  /* the `tminmax` argument holds (t_ref, period) itself instead of (min t, max t):
                            explicit reference_time / time_period of warp_event_from_optical_flow and
                            warp_event_2dof_xy (:292-293, 344-349) */
} ebos_reftime_mode;

/* image accumulation methods, src/event_image_converter.py:351-367 */
typedef enum ebos_splat_mode {
  EBOS_SPLAT_BILINEAR = 0, /* bilinear_vote_*  (:503-620)                                  */
  EBOS_SPLAT_COUNT = 1,    /* count_event_*    (:407-501): +1 on every in-bounds neighbour */
  EBOS_SPLAT_POLARITY = 2  /* "polarity" (:355-363): image is [b,2,h,w]; channel 0 <- p > 0 */
} ebos_splat_mode;

/* Optional timing of the dominant kernel (the tile accumulate pass of ebos_iwe_dense_slab_f32): between
 * ebos_profile_start(max_records) and ebos_profile_stop() every launch of that kernel carries a HIP event pair
 * stamped with the dispatch's own begin / end (hipExtLaunchKernelGGL) on the stream it runs on.  ebos_profile_stop synchronises those events, writes up to
 * `cap` durations in milliseconds to the HOST array `ms` and returns how many it wrote.  Not thread-safe;
 * off by default. */
int ebos_profile_start(int max_records);
int ebos_profile_stop(float* ms, int cap);
/* The same for another kernel of the path (one selector active at a time): */
typedef enum ebos_profile_kernel {
  EBOS_PROFILE_SLAB_ACCUMULATE = 0, /* iwe_slab_accumulate_kernel (what ebos_profile_start selects)                    */
  EBOS_PROFILE_TILED_BWD = 1,       /* iwe_dense_tiled_bwd_kernel of ebos_iwe_{dense,2dof,patch}_tiled_bwd_f32          */
  EBOS_PROFILE_SLAB_COMBINE = 2,    /* the slab combine pass (IWE assembly + variance) of ebos_iwe_*_slab_f32           */
  EBOS_PROFILE_GRADMAG_FUSED = 3    /* the Sobel pass of ebos_gradient_magnitude_fused_f32 (value partials + gradient)  */
} ebos_profile_kernel;
int ebos_profile_start_kernel(int which, int max_records);

int ebos_version(void);               /* EBOS_ABI_VERSION of the loaded library */
const char* ebos_last_error(void);    /* host string                            */
const char* ebos_build_info(void);    /* host string: arch, compiler, options   */

/* ------------------------------------------------------------------------------------------
 * A2  time range of an event batch.  Replaces nt_min/nt_max over events[..., 2]
 *     (src/warp.py:245-253, 283-287; src/types/__init__.py:21-47).
 *     events [b, n, 4] AoS; out tminmax [b, 2] = (min t, max t) per batch row.
 * ---------------------------------------------------------------------------------------- */
int ebos_time_range_f32(const float* events, int64_t b, int64_t n, float* tminmax, ebos_stream_t stream);
int ebos_time_range_f64(const double* events, int64_t b, int64_t n, double* tminmax, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A3  Warp.warp_event(..., "dense-flow")  (src/warp.py:193-228 -> :292-342, torch branch :330-342).
 *     dt = t - t_ref; if normalize_t: dt /= (max dt - min dt)            (:283-287)
 *     i  = trunc(x) * row_stride + trunc(y)                              (:334)
 *     x' = x - dt * flow[0][i];  y' = y - dt * flow[1][i];  t' = dt;  p' = p   (:335-337)
 *     Arithmetic is done in the element type with the reference's operation order and without
 *     FMA contraction, so the result is bit-identical to the reference on the same dtype.
 *     events/warped [b, n, 4]; flow [b, 2, H, W]; tminmax [b, 2] from ebos_time_range_*.
 *     row_stride = Warp.image_size[1].  An event whose source index falls outside [0, H*W)
 *     (torch.gather would raise) passes through un-displaced and increments *oob_count
 *     (device int32, nullable).
 * ---------------------------------------------------------------------------------------- */
int ebos_warp_dense_f32(const float* events, const float* flow, const float* tminmax, int ref_mode,
                        double ref_fraction, int normalize_t, int64_t b, int64_t n, int H, int W,
                        int row_stride, float* warped, int32_t* oob_count, ebos_stream_t stream);
int ebos_warp_dense_f64(const double* events, const double* flow, const double* tminmax, int ref_mode,
                        double ref_fraction, int normalize_t, int64_t b, int64_t n, int H, int W,
                        int row_stride, double* warped, int32_t* oob_count, ebos_stream_t stream);

/* autograd of A3 w.r.t. the flow: d_flow[c][i] += -dt * d_warped[..., c]
 * (accumulates; d_flow [b, 2, H, W]; d_warped [b, n, 4], columns 2,3 ignored). */
int ebos_warp_dense_bwd_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                            int normalize_t, const float* d_warped, int64_t b, int64_t n, int H, int W,
                            int row_stride, float* d_flow, ebos_stream_t stream);
int ebos_warp_dense_bwd_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                            int normalize_t, const double* d_warped, int64_t b, int64_t n, int H, int W,
                            int row_stride, double* d_flow, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A4  Warp.warp_event(..., "2d-translation" | "rigid-optical-flow")  (src/warp.py:344-383).
 *     x' = x + dt * theta[0]; y' = y + dt * theta[1]  (note the + sign, :368-375).  Un-batched.
 *     theta: device [2].  time_period: device [1] or NULL (then max dt - min dt, :285-286).
 *     bwd: d_theta[c] += sum_n dt * d_warped[n][c]  (accumulates, device [2]).
 * ---------------------------------------------------------------------------------------- */
int ebos_warp_2dof_f32(const float* events, const float* theta, const float* tminmax, int ref_mode,
                       double ref_fraction, int normalize_t, const float* time_period, int64_t n,
                       float* warped, ebos_stream_t stream);
int ebos_warp_2dof_f64(const double* events, const double* theta, const double* tminmax, int ref_mode,
                       double ref_fraction, int normalize_t, const double* time_period, int64_t n,
                       double* warped, ebos_stream_t stream);
int ebos_warp_2dof_bwd_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, const float* time_period, const float* d_warped, int64_t n,
                           float* d_theta, ebos_stream_t stream);
int ebos_warp_2dof_bwd_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, const double* time_period, const double* d_warped, int64_t n,
                           double* d_theta, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A7/A8/A10  EventImageConverter.bilinear_vote_tensor/_numpy, count_event_*, "polarity"
 *     (src/event_image_converter.py:407-620, 355-363).
 *     r0 = floor(x + eps), c0 = floor(y + eps); fr = x - r0, fc = y - c0;  R = r0 + pad_h, C = c0 + pad_w
 *     (R,C) += (1-fr)(1-fc) w; (R+1,C) += fr (1-fc) w; (R,C+1) += (1-fr) fc w; (R+1,C+1) += fr fc w
 *     each only when inside the [h, w] image (h, w = PADDED size, :34).  eps = 1e-6 for tensors
 *     (:586), 1e-8 for numpy arrays (:528).  Out-of-image taps are skipped (the reference adds 0
 *     to pixel 0, :617-618: identical for finite inputs).
 *     events [b, n, 4]; weight: device [b, n] or NULL (then weight_scalar, :576-577,611-614);
 *     image [b, h, w] ([b, 2, h, w] for EBOS_SPLAT_POLARITY).  Accumulates.
 * ---------------------------------------------------------------------------------------- */
int ebos_splat_f32(const float* events, const float* weight, double weight_scalar, int mode, double eps,
                   int64_t b, int64_t n, int h, int w, int pad_h, int pad_w, float* image,
                   ebos_stream_t stream);
int ebos_splat_f64(const double* events, const double* weight, double weight_scalar, int mode, double eps,
                   int64_t b, int64_t n, int h, int w, int pad_h, int pad_w, double* image,
                   ebos_stream_t stream);

/* autograd of the bilinear splat, G = d_image [b, h, w]:
 *   d_events[..., 0] = w [(1-fc)(G[R+1,C]-G[R,C]) + fc (G[R+1,C+1]-G[R,C+1])]
 *   d_events[..., 1] = w [(1-fr)(G[R,C+1]-G[R,C]) + fr (G[R+1,C+1]-G[R+1,C])]   (columns 2,3 = 0)
 *   d_weight[n]      = sum_taps tap_weight * G[tap]
 * d_events [b, n, 4] and d_weight [b, n] are overwritten; either may be NULL. */
int ebos_splat_bwd_f32(const float* events, const float* weight, double weight_scalar, double eps,
                       const float* d_image, int64_t b, int64_t n, int h, int w, int pad_h, int pad_w,
                       float* d_events, float* d_weight, ebos_stream_t stream);
int ebos_splat_bwd_f64(const double* events, const double* weight, double weight_scalar, double eps,
                       const double* d_image, int64_t b, int64_t n, int h, int w, int pad_h, int pad_w,
                       double* d_events, double* d_weight, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Event plan: the device-resident, iteration-invariant form of one event window used by the
 * fused hot path.  (x, y, t, p) of every event are constant across solver iterations and flow
 * hypotheses: only the motion changes, so this is built once per window.
 *
 * ebos_events_to_soa_*: AoS [n,4] -> SoA f32 (x, y, dt, p).  dt follows src/warp.py:264-288
 * but is evaluated in fp64 and rounded once to f32 (absolute timestamps of ~10 s do not
 * survive f32).
 * ---------------------------------------------------------------------------------------- */
int ebos_events_to_soa_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, int64_t n, float* x, float* y, float* dt, float* p,
                           ebos_stream_t stream);
int ebos_events_to_soa_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, int64_t n, float* x, float* y, float* dt, float* p,
                           ebos_stream_t stream);

/* Raw sensor columns -> SoA (event ingest).  The CCS recordings hold
 * raw_events/{x: int16 column, y: int16 row, t: int32 microseconds, p: bool}
 * (src/data_loader/ccs.py:57-66); the reference expands a window on the host to float64 [n, 4] =
 * (y, x, t / 1e6, p) (src/data_loader/ccs.py:289-297) before Warp sees it.  These two entry points
 * take the 9 B/event raw window as it is on the device and produce the same SoA plan as
 * ebos_events_to_soa_f64 on that float64 array (bit-identical: the same fp64 time arithmetic).
 *   t_bytes: 4 (int32) or 8 (int64) ticks; ticks_per_second: 1e6 for microseconds.
 *   ebos_raw_time_range: tminmax[2] (device, seconds) = (min t, max t) / ticks_per_second;
 *     scratch_ticks: device int64[2].  n >= 1.
 *   ebos_raw_events_to_soa: ref_mode / ref_fraction / normalize_t as ebos_events_to_soa_*. */
int ebos_raw_time_range(const void* t, int t_bytes, int64_t n, double ticks_per_second, int64_t* scratch_ticks,
                        double* tminmax, ebos_stream_t stream);
int ebos_raw_events_to_soa(const int16_t* col, const int16_t* row, const void* t, int t_bytes,
                           const uint8_t* pol, double ticks_per_second, const double* tminmax, int ref_mode,
                           double ref_fraction, int normalize_t, int64_t n, float* x, float* y, float* dt,
                           float* p, ebos_stream_t stream);

/* Source-tile binning (counting sort by source pixel, tile-major).  The image [H, W] is cut into
 * tiles of tile_h x tile_w pixels; key(event) = tile_id * tile_h*tile_w + pixel-in-tile of
 * (trunc(x), trunc(y)).  Events are reordered by key; events of one tile, and of one source
 * pixel, become contiguous (coalesced flow reads forward, segmented reduction backward).
 *   n_keys = tiles_y * tiles_x * tile_h * tile_w,  tiles_y = ceil(H / tile_h), tiles_x = ceil(W / tile_w)
 *   key_offsets [n_keys + 1] (int32, out): exclusive prefix sum of the per-key counts;
 *       tile t owns sorted positions [key_offsets[t * tile_h*tile_w], key_offsets[(t+1) * tile_h*tile_w])
 *   perm [n] (int32, out): original index of the event at each sorted position
 *   scratch: >= ebos_bin_scratch_bytes(n_keys) bytes
 *   oob_count (device int32, nullable): events whose source pixel is outside the image; they are
 *       dropped from the plan (torch.gather would raise for them, src/warp.py:334-336).
 *   frac_count (device int32, nullable): number of kept events with a fractional / negative source
 *       coordinate; the compact plan (ebos_plan_compact_f32) is valid iff it stays 0.
 * All SoA outputs must be 16-byte aligned and padded to a multiple of 4 elements (vector loads).
 * The order of events inside one source pixel is not deterministic (atomic cursor). */
size_t ebos_bin_scratch_bytes(int64_t n_keys);
/* scratch size that lets ebos_bin_events_f32 scatter one aligned 32-byte record per event (then unpacked to the SoA
 * arrays by a streaming pass) instead of five 4-byte stores to five arrays; with only ebos_bin_scratch_bytes(n_keys)
 * the five-store form runs.  Never smaller than ebos_bin_scratch_bytes(n_keys). */
size_t ebos_bin_scratch_bytes_events(int64_t n, int H, int W, int tile_h, int tile_w);
int ebos_bin_events_f32(const float* x, const float* y, const float* dt, const float* p, int64_t n,
                        int H, int W, int tile_h, int tile_w, float* xs, float* ys, float* dts, float* ps,
                        int32_t* perm, int32_t* key_offsets, int32_t* oob_count, int32_t* frac_count,
                        void* scratch, size_t scratch_bytes, ebos_stream_t stream);

/* Adaptive work items for the tile-private forward kernels.  With one workgroup per tile the pass lasts as long as
 * the fullest tile; real windows are far from uniform (a schlieren object in front of a static background).
 * ebos_plan_parts cuts heavy tiles into parts: parts(t) = max(1, ceil(load(t) / tau)), where tau balances the
 * longest work item against the average load of a CU:  tau + F = (N + items(tau) F) / n_cu  (n_cu = CUs of the
 * device, F = fixed_events = the per-item fixed work in events: 8192 for a 10 M-event window or few tiles on many CUs;
 * where the tiles fill the device an extra item is an extra round and ~6.5e10 / N was measured best -- 32 k at 2 M events,
 * at most 65536: EventPlan's part_fixed_events), within a budget of 2 x tiles work
 * items.  A uniform window keeps one part per tile.  part_table (int32, out, 5 tiles + 1 entries):
 *     [0, tiles]               part_off: first slab of each tile (part_off[tiles] = work items in use)
 *     [tiles + 1, 3 tiles]     item_tile: tile of each of the 2 x tiles work items, heaviest first (the dispatcher
 *                              then schedules longest-processing-time first), -1 = unused
 *     [3 tiles + 1, 5 tiles]   item_part: which part of that tile
 * Pass it with splits = 0 to ebos_iwe_dense_slab_f32 / ebos_iwe_2dof_slab_f32 (workspace sized with splits = 0). */
int ebos_plan_parts(const int32_t* key_offsets, int H, int W, int tile_h, int tile_w, int n_cu, int fixed_events,
                    int32_t* part_table, ebos_stream_t stream);

/* What the host wants to know about a freshly built plan, as four int32 side by side (one small launch, one 16-byte copy):
 *     facts[0] = counts[0]   events outside the image (dropped)      facts[2] = part_table[tiles]  work items in use
 *     facts[1] = counts[1]   events with fractional source pixels    facts[3] = events of the fullest source tile
 * counts = {oob_count, frac_count} of ebos_bin_events_f32 / the counts pair of ebos_plan_lean (NULL: 0), part_table of
 * ebos_plan_parts (NULL: 0).  The reference has no counterpart: its loader crops on the host (src/solver/patch_eklt.py:262-281);
 * the fullest tile decides between the resident solver kernel and the four launches (solver/fused_loop.py: crowded_for_resident). */
int ebos_plan_facts(const int32_t* key_offsets, int H, int W, int tile_h, int tile_w, const int32_t* counts,
                    const int32_t* part_table, int32_t* facts, ebos_stream_t stream);

/* Lean plan build: the compact plan (below) straight from the window -- AoS float32 / float64 [n, 4] = (x = row, y = col, t, p)
 * as the reference's loader hands it over (src/data_loader/ccs.py:289-297), or the raw sensor columns (:57-66) -- without
 * the SoA arrays and the permutation that only per-event weights and fractional source coordinates need.  The window is read
 * ONCE: every chunk of it is counting-sorted by (tile, row band) inside LDS and staged as a coalesced stream, one workgroup per
 * band then gathers its runs, sorts them by source pixel in LDS and puts every pixel's events in ascending dt (two builds of one
 * window hold identical arrays); no global atomics per event, no scattered writes.  Outputs exactly what ebos_bin_events_f32 +
 * ebos_plan_compact_f32 produce:
 *     key_offsets [n_keys + 1], grp_offsets [tiles + 1], cpix / cdt [capacity_slots >= n + 3 tiles + 4]
 * (the order of the events inside one source pixel is unspecified in both), plus
 *     counts [2] (device int32): events outside the image (dropped), kept events with a fractional / negative source
 *                coordinate -- the plan is valid iff counts[1] == 0;
 *     tminmax [2] (device double, nullable): (min t, max t) of the window in seconds.
 *   source: 0 = `events` float32 [n, 4], 1 = `events` float64 [n, 4], 2 / 3 = raw columns with int32 / int64 ticks.
 *   ref_mode: EBOS_REF_FIRST / LAST / FRACTION; dt = (t - t_ref) [/ (tmax - tmin) if normalize_t], evaluated in fp64.
 *   scratch: >= ebos_plan_lean_scratch_bytes(...) bytes (10 B per event + the chunks' bin tables).  1 <= n < 2^31. */
size_t ebos_plan_lean_scratch_bytes(int64_t n, int H, int W, int tile_h, int tile_w);
int ebos_plan_lean(int source, const void* events, const int16_t* col, const int16_t* row, const void* t,
                   double ticks_per_second, int64_t n, int ref_mode, double ref_fraction, int normalize_t, int H, int W,
                   int tile_h, int tile_w, int32_t* key_offsets, int32_t* grp_offsets, uint16_t* cpix, float* cdt,
                   int64_t capacity_slots, int32_t* counts, double* tminmax, void* scratch, size_t scratch_bytes,
                   ebos_stream_t stream);
/* Compact plan: the 6 B/event layout of the tile-private kernels, valid when every source coordinate is a
 * non-negative integer (frac_count == 0; camera events always are).  Per tile t the events occupy the groups
 * [grp_offsets[t], grp_offsets[t+1]) of 4 slots (16-byte vector loads, tiles start on a group boundary):
 *     cpix (u16) = (row_in_tile << 8) | col_in_tile          cdt (f32) = dt, NaN in padding slots
 * grp_offsets [tiles + 1] int32; cpix / cdt hold capacity_slots >= n + 3 tiles + 4 elements, 16-byte aligned.
 * Inputs are the binned arrays of ebos_bin_events_f32 (same tile size, tile_h, tile_w <= 256). */
int ebos_plan_compact_f32(const float* xs, const float* ys, const float* dts, const int32_t* key_offsets, int64_t n,
                          int H, int W, int tile_h, int tile_w, int32_t* grp_offsets, uint16_t* cpix, float* cdt,
                          int64_t capacity_slots, ebos_stream_t stream);
/* The compact plan of a window whose source coordinates are FRACTIONAL (events rectified with a sub-pixel map or already warped by an earlier stage;
 * the flow is looked up at the truncated coordinate, src/warp.py:334): cpix / cdt as above from floor(x),
 * floor(y), plus cfx / cfy [capacity_slots] f32 = x - floor(x), y - floor(y) per slot (0 in padding slots).  Same information as
 * the (x, y, dt) arrays it is made from; read by the resident 2-DoF launch (ebos_cmax_2dof_problem::cfx / cfy). */
int ebos_plan_compact_frac_f32(const float* xs, const float* ys, const float* dts, const int32_t* key_offsets, int64_t n,
                               int H, int W, int tile_h, int tile_w, int32_t* grp_offsets, uint16_t* cpix, float* cdt,
                               float* cfx, float* cfy, int64_t capacity_slots, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused hot path, dense flow: A3 + A7 in one pass, nothing materialised
 *   (src/warp.py:330-342 + src/event_image_converter.py:581-620), eps = 1e-6.
 *   x, y, dt, weight(nullable): SoA f32 [n];  flow [2, H, W];  iwe [h, w] with h = H + 2 pad_h ...
 *   Accumulates into iwe.
 * ebos_iwe_dense_f32        any event order; one global float atomic per tap.
 * ebos_iwe_dense_tiled_f32  binned events (ebos_bin_events_f32 with the same tile_h/tile_w):
 *   one workgroup per (tile, split) accumulates into an LDS image of the tile plus `halo` pixels
 *   on every side and flushes it once; taps beyond the halo fall back to global atomics, so the
 *   result is correct for any flow magnitude.  halo must be one of the built values
 *   (ebos_tiled_config lists them).  splits >= 1 workgroups share one tile's events.
 * ---------------------------------------------------------------------------------------- */
int ebos_iwe_dense_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                       const float* flow, int H, int W, int row_stride, int pad_h, int pad_w, float* iwe,
                       ebos_stream_t stream);
int ebos_iwe_dense_tiled_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                             const int32_t* key_offsets, int64_t n, const float* flow, int H, int W,
                             int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w,
                             float* iwe, ebos_stream_t stream);
/* number of supported (tile_h, tile_w, halo) triples; fills up to `cap` triples into out[3*i..] (host) */
int ebos_tiled_config(int* out, int cap);

/* backward of the fused dense path: G = a * g_image + c (g_image [h, w]; pass a = 1, c = 0 for a plain
 * upstream gradient; the variance cost folds its gradient 2 (IWE - mean)/(M-1) into (a, c) read from
 * the device array affine[2] so that no d_iwe image is materialised; affine may be NULL).
 *   d_flow[c][i] += -dt * dL/d(x',y')   with dL/dx', dL/dy' as in ebos_splat_bwd_*.
 * sorted != 0 promises that events sharing a source pixel are contiguous (binned plan): contributions
 * are pre-reduced across the wavefront with shuffles and one atomic per run is issued.
 * g_lo/g_hi: rows/cols [g_lo, h - g_lo) x [g_lo, w - g_lo) of g_image are valid, G = 0 elsewhere
 * (omit_boundary of the costs: g_lo = 1).  Accumulates into d_flow [2, H, W]. */
int ebos_iwe_dense_bwd_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                           const float* flow, int H, int W, int row_stride, int pad_h, int pad_w,
                           const float* g_image, const float* affine, int g_lo, int sorted, float* d_flow,
                           float* d_weight, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Tile-private pipeline (the fast path): same mathematics as ebos_iwe_dense_tiled_f32 /
 * ebos_iwe_dense_bwd_f32, organised so that no global float atomic and no zero-fill is left.
 *
 * ebos_iwe_dense_slab_f32   forward.  Each (tile, split) workgroup accumulates its tile + halo in LDS
 *   (f64) and writes it once as a plain f32 "slab"; a combine pass sums the slabs covering each pixel,
 *   OVERWRITES iwe [h, w] (no caller-side zeroing) and, if want_variance, also reduces the variance of
 *   the image (omit_boundary as in the costs) into out_variance [1] (f32, nullable) and
 *   moments [2] = (mean, M) (f64, nullable) -- the contrast cost of SURVEY.md A14 at no extra pass.
 *   want_variance = 2: only the (sum, sum of squares) partials are left in the workspace (ebos_iwe_slab_partials),
 *   no finalize launch; out_variance / moments are not touched.
 *   workspace: >= ebos_iwe_slab_workspace_bytes(...) bytes, ZERO-FILLED ONCE by the caller when it is
 *   allocated; the kernels keep its spill section (taps beyond the halo) zero between calls, and a word behind the partials
 *   holds the number of the last call whose accumulate pass wrote spill taps (the combine pass reads the spill section only
 *   for that call).  One workspace serves one stream at a time.
 *   Results are deterministic (fixed summation order) except for taps beyond the halo.
 *   splits >= 1: every tile is cut into `splits` equal parts (workgroups); splits = 0: the adaptive work items of
 *   part_table (ebos_plan_parts; NULL otherwise).
 * ebos_iwe_dense_tiled_bwd_f32   backward.  One workgroup per tile: upstream image tile in LDS,
 *   wavefront-segmented sums per source pixel, d_flow [2, H, W] OVERWRITTEN with plain stores
 *   (binned plans only; g_image/affine/g_lo/d_weight as in ebos_iwe_dense_bwd_f32; d_weight in plan order).
 *   var_moments [2] (f64: mean, M of ebos_iwe_dense_slab_f32) + upstream [1] (f32), both nullable together: the
 *   loss is upstream * var(g_image) with g_image = the IWE itself; its gradient 2 (IWE - mean) / (M - 1) is folded
 *   into the kernel (no d_iwe image, no affine launch).
 *   addend [2, H, W] (nullable): added to the result as it is stored -- the gradient of the flow regularisers
 *   (ebos_flow_regularisers_f32), so that d_flow is the gradient of the whole objective without another pass.
 *   part_table (nullable): the adaptive work items of ebos_plan_parts -- every part of a tile writes a partial d_flow
 *   tile into the slab section of `workspace` (the forward workspace sized with splits = 0; its slabs are dead once
 *   the IWE is combined) and a second small kernel sums the parts.  workspace may be NULL when part_table is.
 * grp_offsets / cpix / cdt (nullable trio): the compact plan of ebos_plan_compact_f32; when given and weight is
 *   NULL, xs/ys/dts are not read (6 B/event instead of 12).  All SoA arrays are read 4 events (16 bytes) per
 *   lane: 16-byte aligned, padded to a multiple of 4 elements.
 * (tile_h, tile_w, halo) must be one of ebos_slab_config() -- or halo = EBOS_HALO_AUTO(max_halo, q):
 *
 * RUN-TIME WINDOWS.  Every `halo` argument / field of this section also takes EBOS_HALO_AUTO(max_halo, q) (a negative number):
 *   (tile_h, tile_w, max_halo) is a built configuration -- it sizes the LDS, the slabs and the workspace -- and every work item
 *   chooses ITS OWN window (hr rows, hc columns of halo per side, hc a multiple of 4, both <= max_halo) inside the kernel, from a
 *   bound on its displacements: max |flow| over the tile's own pixels (dense flow), over the grid cells its pixels interpolate
 *   (patch grid), or |theta| (2-DoF), times q / 64 >= max |dt| over the plan's events (1.0 -> q = 64 with normalised time and
 *   reference time inside the window; ebos_halo_auto rounds a bound up for you).  LDS clear / decode, slab traffic, the combine
 *   pass's reads and the backward kernel's upstream tile shrink with the window; BOS displacements are a few pixels, the +-30 px
 *   of the sampler range (configs/hot_plate1.yaml:46-80) is the search bound, not the operating point.  Taps beyond a window go to
 *   the spill image exactly as with a built halo, so the choice moves time, never results: images are bit-identical to those of
 *   the built max_halo as long as nothing spills.  Compact plans with unit weights (the lean loops); other calls run max_halo.
 *   The forward pass records each tile's window behind the SpillEpoch word of its workspace for its combine pass.
 * ---------------------------------------------------------------------------------------- */
#define EBOS_HALO_AUTO(max_halo, dt_bound_q64) (-((int)(max_halo) + 256 * (int)(dt_bound_q64)))
int ebos_halo_auto(int max_halo, double dt_bound); /* EBOS_HALO_AUTO(max_halo, ceil(64 dt_bound)); max_halo itself if out of range */
int ebos_slab_config(int* out, int cap);
size_t ebos_iwe_slab_workspace_bytes(int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h,
                                     int pad_w);
/* want_variance = 2 in ebos_iwe_dense_slab_f32 / ebos_iwe_2dof_slab_f32 leaves the variance as (sum, sum of squares)
 * partials in the workspace and skips the finalize launch; this (host-only) call tells where they are: byte offset
 * inside the workspace, number of partial pairs, and the pixel count M the variance refers to. */
int ebos_iwe_slab_partials(int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w,
                           int omit_boundary, size_t* offset_bytes, int64_t* n_partials, int64_t* n_pixels);
int ebos_iwe_dense_slab_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                            const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                            const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                            int tile_w, int halo, int splits, int pad_h, int pad_w, void* workspace,
                            size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary,
                            float* out_variance, double* moments, const int32_t* part_table,
                            ebos_stream_t stream);
/* One call for the objective AND its gradient -- the drop-in autograd path (plan.contrast_dense(flow).backward(), i.e.
 * warp.py:330-342 + event_image_converter.py:581-620 + torch.var and what autograd derives) costs what its kernels
 * cost only if the host does not marshal ~60 arguments through two calls per iteration.  ebos_dense_job holds everything
 * that is constant per (plan, padding, halo, splits): the caller fills it ONCE (plain pointers / sizes, same meaning as the
 * arguments of ebos_iwe_dense_slab_f32 / ebos_iwe_dense_tiled_bwd_f32) and passes its address afterwards.
 *   iwe [h, w] and moments [2] are scratch outputs owned by the job (IWE and (mean, M) of the LAST evaluation).
 * ebos_variance_dense_job_f32: out_variance[0] = var(IWE(flow)) (omit_boundary as in the costs); if d_flow != NULL also
 *   d_flow [2, H, W] = upstream[0] * d var / d flow (upstream: device f32 [1]; NULL = 1).  Enqueues accumulate, combine and
 *   finalize on `stream` -- or, with d_flow, accumulate, combine and the tile-private backward, which reduces the variance
 *   partials of the combine pass itself and writes out_variance / moments (three launches, no finalize); no host
 *   synchronisation. */
typedef struct ebos_dense_job {
  const float *xs, *ys, *dts;          /* (x, y, dt) plan, nullable when the compact trio is given */
  const int32_t* grp_offsets;          /* compact plan (ebos_plan_compact_f32), nullable trio      */
  const uint16_t* cpix;
  const float* cdt;
  const int32_t* key_offsets;
  int64_t n;
  int H, W, tile_h, tile_w, halo, splits, pad_h, pad_w, omit_boundary;
  void* workspace;                     /* >= ebos_iwe_slab_workspace_bytes(...), zero-filled once   */
  size_t workspace_bytes;
  const int32_t* part_table;           /* adaptive work items (splits == 0), else nullable          */
  float* iwe;                          /* [H + 2 pad_h, W + 2 pad_w]                                */
  double* moments;                     /* [2]                                                       */
} ebos_dense_job;
int ebos_variance_dense_job_f32(const ebos_dense_job* job, const float* flow, float* out_variance, const float* upstream,
                                float* d_flow, ebos_stream_t stream);
/* The same for the gradient-magnitude contrast (BASELINE configs[2]; SURVEY.md A14: mean squared Sobel gradient of the IWE,
 * src/utils/stat_utils.py:69-92, 117-139): out_contrast[0] = gradient_magnitude(IWE(flow)); with d_flow != NULL also
 * d_flow [2, H, W] = upstream[0] * d contrast / d flow.  Enqueues accumulate, combine, ONE Sobel pass (value partials + gradient
 * image, ebos_gradient_magnitude_fused_f32) and the tile-private backward, whose first workgroup sums the value partials: four
 * launches, no finalize, no host synchronisation.  d_iwe [h, w] f32 and partials [ebos_gradient_magnitude_fused_partials(h, w)]
 * f64 are scratch of the caller's (job->moments is not used).  out_scaled (nullable): see ebos_variance_dense_job_signed_f32. */
int ebos_gradient_magnitude_dense_job_f32(const ebos_dense_job* job, const float* flow, float* out_contrast, float* out_scaled,
                                          const float* upstream, float* d_flow, float* d_iwe, double* partials, int64_t n_partials,
                                          ebos_stream_t stream);
/* ebos_variance_dense_job_f32 with one more output for a cost that has a DIRECTION (the cost plugins of src/costs: "minimize" negates the contrast):
 * out_scaled[0] = upstream[0] * variance (nullable; needs d_flow), written by the backward kernel's first workgroup beside
 * out_variance -- the signed loss and its gradient come out of the three launches, no kernel negates a scalar or scales 7.4 MB.
 * (ebos_gradient_magnitude_dense_job_f32 takes the same out_scaled.) */
int ebos_variance_dense_job_signed_f32(const ebos_dense_job* job, const float* flow, float* out_variance, float* out_scaled,
                                       const float* upstream, float* d_flow, ebos_stream_t stream);
int ebos_iwe_dense_tiled_bwd_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                                 const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                 const int32_t* key_offsets, int64_t n, const float* flow, int H, int W,
                                 int tile_h, int tile_w, int halo, int pad_h, int pad_w, const float* g_image,
                                 const float* affine, int g_lo, float* d_flow, float* d_weight,
                                 const double* var_moments, const float* upstream, const float* addend,
                                 void* workspace, size_t workspace_bytes, const int32_t* part_table,
                                 ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The same pipeline with the flow given as a PATCH GRID [2, gh, gw] (the parameters of the patch solvers,
 * src/solver/patch_eklt.py:173-204: dense = crop(resize(replicate_pad(grid)))): every workgroup evaluates the
 * grid -> dense map for its own source tile into LDS and the event loop takes the flow from there -- no dense
 * [2, H, W] field, no upsample launch, LDS reads instead of L2 gathers ("fused into the warp's flow
 * fetch").  Same results as ebos_upsample_patch_flow_f32 + ebos_iwe_dense_slab_f32 (one shared expression).
 *
 * ebos_patch_fused_supported   1 when (tile, halo) leaves LDS for the tile's flow and a tile + 2 px apron touches at most
 *   16 x 16 grid cells ((tile + 4) / slide + 3 <= 16 per axis); 0 otherwise (use the upsample + dense entry points).
 * ebos_iwe_patch_slab_f32      forward; arguments as ebos_iwe_dense_slab_f32 with (grid, gh, gw, patch, slide) in place
 *   of flow; compact plans with unit weights only.
 * ebos_iwe_patch_tiled_bwd_f32 backward; instead of d_flow every work item writes the adjoint of the grid -> dense map
 *   restricted to its tile: <= 16 x 16 partial cell gradients per flow component into grad_partials
 *   (ebos_patch_grad_partials_bytes; adaptive = part_table != NULL).  addend [2, H, W] (nullable) enters once per tile.
 *   w_flow_norm / w_image_gradient != 0: the flow regularisers w * mean |flow| (src/costs/flow_norm.py:45-56, pointwise) and
 *   w * mean(|d flow / d row| + |d flow / d col|) (src/costs/image_gradient.py:60-75, torch.gradient lines, unit weights) are
 *   evaluated on the flow the tile holds in LDS (with a 2 px apron): gradient added here, value as one f64 partial per work item in
 *   reg_partials [ebos_patch_grad_partials_bytes / 2048] (slots of unused work items are not written: zero the buffer
 *   once) -- no dense field and no ebos_flow_regularisers_f32 launch.
 *   var_partials (nullable; then var_moments must be NULL and upstream given): the (sum, sum of squares) partials that the
 *   forward call left with want_variance = 2 (ebos_iwe_slab_partials tells where).  Every workgroup reduces them itself
 *   (14 KB of L2 reads) and folds the variance gradient in; workgroup 0 also writes out_variance [1] / out_moments [2]
 *   (nullable) -- the finalize launch between forward and backward disappears.
 * ebos_patch_grad_combine_adam_f32   d_grid [2, gh, gw] := sum of the partials of the tiles touching each cell, times
 *   grad_mask (nullable); with theta != NULL also the Adam step and loss bookkeeping of
 *   ebos_upsample_patch_flow_bwd_adam_f32 (same arguments).  theta == NULL: plain gradient (optimiser arguments unused).
 * Together they replace upsample -> dense slab -> dense tiled bwd -> adjoint rows -> adjoint cols (+ Adam): five
 * launches become three and the [2, H, W] flow / gradient fields disappear.
 * ---------------------------------------------------------------------------------------- */
int ebos_patch_fused_supported(int tile_h, int tile_w, int halo, int slide_h, int slide_w);
/* ebos_iwe_slab_batch_f32  n_windows INDEPENDENT windows of one geometry (the time windows of bos_event.py:144-220, BASELINE
 *   configs[3]) per call: accumulate, combine and finalize each run as ONE launch over (work item, window) for up to 16 windows
 *   at a time -- thin windows are bound by per-launch fixed work, and at one workgroup per CU consecutive accumulate launches
 *   cannot overlap.  Per window: its compact plan (unit weights), its flow -- a dense field [2, H, W] when gh == gw == 0, else
 *   a patch grid [2, gh, gw] (arguments as ebos_iwe_patch_slab_f32) --, its OWN workspace (each >= workspace_bytes >=
 *   ebos_iwe_slab_workspace_bytes, zero-filled once), iwe [h, w] and variance outputs.  Needs w and pad_w multiples of 4.
 *   Results are bit-identical to n_windows calls of ebos_iwe_dense_slab_f32 / ebos_iwe_patch_slab_f32.  `windows` is a HOST
 *   array, read before the call returns.  tail_stream (nullable): a second stream of the caller's for the combine / finalize
 *   passes, which need no LDS and then run beside the accumulate pass of the next 16 windows; `stream` is made to wait for it
 *   before the call returns, so the results are ordered on `stream` either way. */
typedef struct ebos_slab_window {
  const int32_t* grp_offsets;  /* the window's compact plan (ebos_plan_lean / ebos_plan_events_*)                 */
  const uint16_t* cpix;
  const float* cdt;
  const int32_t* key_offsets;
  const int32_t* part_table;   /* adaptive work items (splits == 0), else nullable                               */
  const float* flow;           /* [2, H, W], or the patch grid [2, gh, gw]                                        */
  void* workspace;             /* this window's own, zero-filled once                                             */
  float* iwe;                  /* [H + 2 pad_h, W + 2 pad_w], overwritten                                         */
  float* out_variance;         /* [1], nullable                                                                   */
  double* moments;             /* [2] (mean, M), nullable                                                         */
} ebos_slab_window;
int ebos_iwe_slab_batch_f32(const ebos_slab_window* windows, int n_windows, int gh, int gw, int patch_h, int patch_w, int slide_h,
                            int slide_w, int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w,
                            size_t workspace_bytes, int want_variance, int omit_boundary, ebos_stream_t stream,
                            ebos_stream_t tail_stream);

int ebos_iwe_patch_slab_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                            const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw, int patch_h,
                            int patch_w, int slide_h, int slide_w, int H, int W, int tile_h, int tile_w, int halo,
                            int splits, int pad_h, int pad_w, void* workspace, size_t workspace_bytes, float* iwe,
                            int want_variance, int omit_boundary, float* out_variance, double* moments,
                            const int32_t* part_table, ebos_stream_t stream);
/* ... on a window of FRACTIONAL source coordinates (undistorted events): the arrays of ebos_plan_compact_frac_f32 -- the compact slots
 * with x - floor(x), y - floor(y) per slot.  The general event loop on the compact layout (cfx == cfy == NULL: ebos_iwe_patch_slab_f32). */
int ebos_iwe_patch_slab_frac_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const float* cfx, const float* cfy,
                                 const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w,
                                 int slide_h, int slide_w, int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h,
                                 int pad_w, void* workspace, size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary,
                                 float* out_variance, double* moments, const int32_t* part_table, ebos_stream_t stream);
size_t ebos_patch_grad_partials_bytes(int H, int W, int tile_h, int tile_w, int adaptive);
int ebos_iwe_patch_tiled_bwd_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                 const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw,
                                 int patch_h, int patch_w, int slide_h, int slide_w, int H, int W, int tile_h,
                                 int tile_w, int halo, int pad_h, int pad_w, const float* g_image,
                                 const float* affine, int g_lo, const double* var_moments, const float* upstream,
                                 const float* addend, float* grad_partials, size_t grad_partials_bytes,
                                 const int32_t* part_table, float w_flow_norm, float w_image_gradient,
                                 double* reg_partials,
                                 const double* var_partials, int64_t n_var_partials, int64_t n_var_pixels,
                                 float* out_variance, double* out_moments, ebos_stream_t stream);
/* The same backward for the contrast of the 3-tap BLURRED image (iwe.blur_sigma > 0, src/event_image_converter.py:399-404):
 * z_image / blur_partials are what ebos_blur3_variance_adjoint_f32 made of the IWE; the kernel reduces the partials (variance of the
 * blurred image -> out_variance / out_moments), and its upstream is a z + c wgt(pixel) with a = 2 upstream / (M - 1),
 * c = -a mean, wgt = B^T (valid region) evaluated from the pixel's position (csrc/blur3.h). */
int ebos_iwe_patch_tiled_bwd_blur_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                      const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw,
                                      int patch_h, int patch_w, int slide_h, int slide_w, int H, int W, int tile_h,
                                      int tile_w, int halo, int pad_h, int pad_w, const float* z_image, int g_lo,
                                      const float* upstream, float* grad_partials, size_t grad_partials_bytes,
                                      const int32_t* part_table, float w_flow_norm, float w_image_gradient,
                                      double* reg_partials, const double* blur_partials, int64_t n_blur_partials,
                                      int64_t n_var_pixels, float* out_variance, double* out_moments, float blur_k0,
                                      float blur_k1, ebos_stream_t stream);
/* The grid-sampling backward pass on a compact plan that carries the fractions of undistorted events (f64 sweep); blur_k0 == 0: the
 * plain contrast (g_image = the IWE, var_partials of the combine pass), else the blurred one (g_image = z_image, var_partials = the
 * blur pass's pairs) -- the arguments of the two entries above. */
int ebos_iwe_patch_tiled_bwd_frac_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const float* cfx, const float* cfy,
                                      const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w,
                                      int slide_h, int slide_w, int H, int W, int tile_h, int tile_w, int halo, int pad_h, int pad_w,
                                      const float* g_image, int g_lo, const double* var_moments, const float* upstream,
                                      const float* addend, float* grad_partials, size_t grad_partials_bytes, const int32_t* part_table,
                                      float w_flow_norm, float w_image_gradient, double* reg_partials, const double* var_partials,
                                      int64_t n_var_partials, int64_t n_var_pixels, float* out_variance, double* out_moments,
                                      float blur_k0, float blur_k1, ebos_stream_t stream);
/* ... and on the dense-flow route (windows of fractional source coordinates: the (x, y, dt) arrays; or compact ones): d loss / d flow
 * [2, H, W] of the blurred contrast, arguments as ebos_iwe_dense_tiled_bwd_f32 / ebos_iwe_patch_tiled_bwd_blur_f32. */
int ebos_iwe_dense_tiled_bwd_blur_f32(const float* xs, const float* ys, const float* dts, const int32_t* grp_offsets, const uint16_t* cpix,
                                      const float* cdt, const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                                      int tile_w, int halo, int pad_h, int pad_w, const float* z_image, int g_lo, const float* upstream,
                                      const float* addend, float* d_flow, void* workspace, size_t workspace_bytes,
                                      const int32_t* part_table, const double* blur_partials, int64_t n_blur_partials,
                                      int64_t n_var_pixels, float* out_variance, double* out_moments, float blur_k0, float blur_k1,
                                      ebos_stream_t stream);
int ebos_patch_grad_combine_adam_f32(const float* grad_partials, const int32_t* part_table, int tile_h, int tile_w,
                                     int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w, int H, int W,
                                     float* d_grid, float* theta, float* exp_avg, float* exp_avg_sq, double lr,
                                     double beta1, double beta2, double eps, int t, int* step, const float* contrast,
                                     float contrast_scale, const double* reg_partials, int n_reg, float* losses,
                                     int losses_cap, const float* grad_mask, ebos_stream_t stream);

/* 2-DoF hypotheses on the tile-private pipeline (BASELINE config 5): thetas [K, 2] (device), x' = x + dt theta
 * (src/warp.py:364-383); iwes [K, h, w] are OVERWRITTEN; out_variance [K] / moments [K, 2] as above.  The K
 * hypotheses run back to back on the stream and share one workspace (same size as for the dense flow). */
int ebos_iwe_2dof_slab_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                           const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                           const int32_t* key_offsets, int64_t n, const float* thetas, int K,
                           int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w,
                           void* workspace, size_t workspace_bytes, float* iwes, int want_variance,
                           int omit_boundary, float* out_variance, double* moments, const int32_t* part_table,
                           ebos_stream_t stream);

/* The same K hypotheses with the accumulate pass PERSISTENT over them (the sweep of src/solver/generative_max_likelihood.py:229-255,
 * BASELINE config 5): one launch per 16 hypotheses in which every workgroup keeps its tile and walks the hypotheses -- one LDS
 * clear per launch, the decode pass zeroes what it reads, the next hypothesis' first chunks are requested while the current image
 * is decoded and stored -- followed by one combine and one finalize launch over (pixel block | 1, hypothesis).  Compact plans with
 * unit weights.  workspaces: K consecutive workspaces of workspace_bytes (>= ebos_iwe_slab_workspace_bytes, a multiple of 256) each,
 * zero-filled once; tail_stream (nullable) as in ebos_iwe_slab_batch_f32.  Results are bit-identical to ebos_iwe_2dof_slab_f32. */
int ebos_iwe_2dof_slab_batch_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                 const int32_t* key_offsets, int64_t n, const float* thetas, int K, int H, int W,
                                 int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w, void* workspaces,
                                 size_t workspace_bytes, float* iwes, int want_variance, int omit_boundary,
                                 float* out_variance, double* moments, const int32_t* part_table, ebos_stream_t stream,
                                 ebos_stream_t tail_stream);

/* backward of ebos_iwe_2dof_slab_f32: d_thetas[k] = sum_n dt * dL/d(x', y') for upstream images g_images [K, h, w]
 * (affine [K, 2] / g_lo as in ebos_iwe_dense_bwd_f32); d_thetas [K, 2] is OVERWRITTEN.  workspace: the plan's forward
 * workspace (its slab section is reused for the per-tile partial sums). */
int ebos_iwe_2dof_tiled_bwd_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                                const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                const int32_t* key_offsets, int64_t n, const float* thetas,
                                int K, int H, int W, int tile_h, int tile_w, int halo, int pad_h, int pad_w,
                                const float* g_images, const float* affine, int g_lo, float* d_thetas,
                                void* workspace, size_t workspace_bytes, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused hot path, 2-DoF hypotheses (solver outer loop, SURVEY.md 3.4 / BASELINE config 5):
 *   A4 + A7 for K translations theta[k] = (theta0, theta1) in one pass over the events
 *   (src/warp.py:364-383 + src/event_image_converter.py:581-620).  iwes [K, h, w] accumulates.
 *   bwd: d_theta[k][c] += sum_n dt * dL/d(x',y') for upstream images g [K, h, w].
 * ---------------------------------------------------------------------------------------- */
int ebos_iwe_2dof_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                      const float* thetas, int K, int h, int w, int pad_h, int pad_w, float* iwes,
                      ebos_stream_t stream);
int ebos_iwe_2dof_bwd_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                          const float* thetas, int K, int h, int w, int pad_h, int pad_w,
                          const float* g_images, const float* affine, int g_lo, float* d_thetas,
                          ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A14  contrast costs on an image (absent from the release; defined in SURVEY.md A14 on the
 *      reference's primitives torch.var and SobelTorch, src/utils/stat_utils.py:48-139;
 *      dict keys "iwe"/"omit_boundary" per src/solver/base.py:337-339).
 *   images [K, h, w]; omit_boundary crops one pixel on every side (iwe[..., 1:-1, 1:-1]).
 *   variance:            out[k] = unbiased variance;  moments[k] = {mean, M} (f64, nullable)
 *   gradient magnitude:  out[k] = mean(gx^2 + gy^2), (gx, gy) = Sobel3(replicate pad) / 8
 *   *_grad: d_images[k] = upstream[k] * d(out[k]) / d(image[k])  (overwrites; upstream device [K], f32)
 *   Sums are carried in f64.  scratch: >= ebos_cost_scratch_bytes(K) bytes.
 * ---------------------------------------------------------------------------------------- */
size_t ebos_cost_scratch_bytes(int K);
int ebos_image_variance_f32(const float* images, int K, int h, int w, int omit_boundary, float* out,
                            double* moments, void* scratch, size_t scratch_bytes, ebos_stream_t stream);
int ebos_image_variance_grad_f32(const float* images, int K, int h, int w, int omit_boundary,
                                 const double* moments, const float* upstream, float* d_images,
                                 ebos_stream_t stream);
/* (a, c) per image such that d(out)/d(image) * upstream = a * image + c inside the valid region:
 * feeds ebos_iwe_*_bwd_f32's `affine` without materialising d_images.  affine [K, 2] f32. */
int ebos_image_variance_affine_f32(const double* moments, const float* upstream, int K, float* affine,
                                   ebos_stream_t stream);
int ebos_gradient_magnitude_f32(const float* images, int K, int h, int w, int omit_boundary, float* out,
                                void* scratch, size_t scratch_bytes, ebos_stream_t stream);
int ebos_gradient_magnitude_grad_f32(const float* images, int K, int h, int w, int omit_boundary,
                                     const float* upstream, float* d_images, ebos_stream_t stream);
int ebos_image_variance_f64(const double* images, int K, int h, int w, int omit_boundary, double* out,
                            double* moments, void* scratch, size_t scratch_bytes, ebos_stream_t stream);
int ebos_image_variance_grad_f64(const double* images, int K, int h, int w, int omit_boundary,
                                 const double* moments, const double* upstream, double* d_images,
                                 ebos_stream_t stream);
int ebos_gradient_magnitude_f64(const double* images, int K, int h, int w, int omit_boundary, double* out,
                                void* scratch, size_t scratch_bytes, ebos_stream_t stream);
int ebos_gradient_magnitude_grad_f64(const double* images, int K, int h, int w, int omit_boundary,
                                     const double* upstream, double* d_images, ebos_stream_t stream);
/* Value AND gradient image of the gradient-magnitude contrast in ONE pass over an LDS-tiled image (the two entry points above read
 * the image twice, the second one 81 times per pixel): out[0] = mean(gx^2 + gy^2) with (gx, gy) = Sobel 3x3 / 8, replicate padding
 * (src/utils/stat_utils.py:69-92, 117-139), d_image [h, w] = upstream[0] * d out / d image (upstream: device f32 [1], NULL = 1) --
 * the bits of ebos_gradient_magnitude_grad_f32.  partials: device f64 [ebos_gradient_magnitude_fused_partials(h, w)], one value
 * partial per workgroup; out == NULL: no finalize launch (the caller sums the partials: ebos_gradient_magnitude_dense_job_f32);
 * d_image == NULL (with out != NULL): the value only -- no gather of the stencils, no gradient image written. */
int64_t ebos_gradient_magnitude_fused_partials(int h, int w);
int ebos_gradient_magnitude_fused_f32(const float* image, int h, int w, int omit_boundary, const float* upstream, float* out,
                                      float* d_image, double* partials, int64_t n_partials, ebos_stream_t stream);

/* Variance of the 3-tap blurred image y = B x -- torchvision gaussian_blur(kernel_size = 3): taps (k0, k1, k0) =
 * exp(-x^2 / 2 sigma^2) at x = -1, 0, 1 normalised to sum 1, reflect padding without edge repeat,
 * src/event_image_converter.py:399-404 -- prepared for the solver loop in ONE pass over the image: partials
 * [ebos_blur3_variance_partials(h, w)][2] f64 = (sum, sum of squares) of the valid blurred pixels per 16 x 64 tile (the layout of
 * the slab combine pass's partials), and z_image [h, w] = B^T (m . y), the part of d var(y) / d x that is linear in x
 * (m = the valid region, omit_boundary).  h, w >= 2 (torch refuses to reflect-pad an axis of one sample). */
int64_t ebos_blur3_variance_partials(int h, int w);
/* cost_scratch of ebos_cmax_patch_problem / ebos_cmax_2dof_problem for an image of h x w pixels (padding included): enough for the
 * value partials of the fused Sobel pass (w_gradient_magnitude != 0) and for the blur's partial pairs (blur_k0 != 0).  ABI 2:
 * ebos_cost_scratch_bytes(1), which ABI 1 documented for this buffer, is too small for the Sobel partials of images beyond ~0.5 MP. */
size_t ebos_cmax_cost_scratch_bytes(int h, int w);
int ebos_blur3_variance_adjoint_f32(const float* image, int h, int w, int omit_boundary, float k0, float k1, float* z_image,
                                    double* partials, int64_t n_partials, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A16  patch grid -> dense flow  (src/solver/patch_eklt.py:173-204): replicate-pad the grid by
 *      pad = int(patch/2 // slide) + 1, bilinear resize (align_corners = False) by the sliding
 *      window, centre-crop to [H, W].  grid [2, gh, gw] -> dense [2, H, W] (overwrites).
 *      bwd: the adjoint, as two separable passes through a [2, gh, W] scratch.
 * ---------------------------------------------------------------------------------------- */
int ebos_upsample_patch_flow_f32(const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h,
                                 int slide_w, int H, int W, float* dense, ebos_stream_t stream);
/* adjoint: d_grid [2, gh, gw] is OVERWRITTEN.  scratch: device, ebos_upsample_bwd_scratch_bytes(gh, W). */
size_t ebos_upsample_bwd_scratch_bytes(int gh, int W);
int ebos_upsample_patch_flow_bwd_f32(const float* d_dense, int gh, int gw, int patch_h, int patch_w,
                                     int slide_h, int slide_w, int H, int W, float* scratch, float* d_grid,
                                     ebos_stream_t stream);
/* The same adjoint with the Adam step of the patch grid applied where each gradient element is produced (Adam is
 * element-wise): theta / exp_avg / exp_avg_sq [2, gh, gw] are updated in place with the gradient of step t (t >= 1,
 * counted by the caller; bias corrections in double on the host like torch.optim.Adam), d_grid still receives the
 * gradient, step[0] := t, and losses[t - 1] := contrast_scale * contrast[0] + sum(reg_partials) (the loss of the
 * parameters before the update; contrast / reg_partials / losses nullable).  Replaces ebos_upsample_patch_flow_bwd_f32 +
 * ebos_cmax_adam_step_f32 in the solver loop: one launch less per iteration.
 * grad_mask [gh, gw] (nullable) multiplies the gradient of both flow components before the step: 0 for the patches that
 * are not estimated -- the event thresholding of src/solver/patch_eklt.py:118-126 (`len(crop_event(...)) > event_thres`),
 * whose patch flow stays where it is (Adam with a zero gradient and zero moments does not move). */
int ebos_upsample_patch_flow_bwd_adam_f32(const float* d_dense, int gh, int gw, int patch_h, int patch_w,
                                          int slide_h, int slide_w, int H, int W, float* scratch, float* d_grid,
                                          float* theta, float* exp_avg, float* exp_avg_sq, double lr, double beta1,
                                          double beta2, double eps, int t, int* step, const float* contrast,
                                          float contrast_scale, const double* reg_partials, int n_reg, float* losses,
                                          int losses_cap, const float* grad_mask, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A9/K11  optional Gaussian blur of an event image (sigma > 0), one separable pass along one axis
 *     of a tensor viewed as [outer, L, inner]; taps: device [2 radius + 1] doubles (host-computed).
 *     boundary 0 = scipy 'reflect' (d c b a | a b c d): gaussian_filter(image, sigma) of the numpy
 *                  branch, one pass per array axis  (src/event_image_converter.py:368-369);
 *     boundary 1 = torch 'reflect' (d c b | a b c d): torchvision gaussian_blur(kernel_size=3) of the
 *                  tensor branch, last two axes     (src/event_image_converter.py:399-404).
 *     out must not alias in.
 * ---------------------------------------------------------------------------------------- */
int ebos_gauss1d_f32(const float* in, float* out, int64_t outer, int64_t L, int64_t inner, const double* taps,
                     int radius, int boundary, ebos_stream_t stream);
int ebos_gauss1d_f64(const double* in, double* out, int64_t outer, int64_t L, int64_t inner, const double* taps,
                     int radius, int boundary, ebos_stream_t stream);
/* Adjoint of one pass: g_in = (d loss / d in) from g_out = (d loss / d out).  What torch autograd
 * derives for gaussian_blur in the tensor branch (src/event_image_converter.py:399-404); gather
 * form, deterministic.  g_in must not alias g_out. */
int ebos_gauss1d_bwd_f32(const float* g_out, float* g_in, int64_t outer, int64_t L, int64_t inner,
                         const double* taps, int radius, int boundary, ebos_stream_t stream);
int ebos_gauss1d_bwd_f64(const double* g_out, double* g_in, int64_t outer, int64_t L, int64_t inner,
                         const double* taps, int radius, int boundary, ebos_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * One contrast-maximisation iteration without autograd glue.  The loop of
 * src/solver/generative_max_likelihood.py:306-341 (zero_grad -> objective -> backward -> Adam step)
 * becomes 5 calls / 8 kernels on one stream (7 without regularisers + 1 finalize):
 *     ebos_upsample_patch_flow_f32 -> ebos_iwe_dense_slab_f32 (want_variance) -> ebos_flow_regularisers_f32
 *     -> ebos_iwe_dense_tiled_bwd_f32 (var_moments, upstream = -weight, addend = regulariser gradient)
 *     -> ebos_upsample_patch_flow_bwd_adam_f32   (ebos_cmax_adam_step_f32 is the stand-alone form of the step)
 *
 * ebos_flow_regularisers_f32: value and gradient of
 *       w_flow_norm * mean_px |flow|_2                                   (src/costs/flow_norm.py:45-56)
 *     + w_image_gradient * mean(|d flow/d row| + |d flow/d col|)         (src/costs/image_gradient.py:60-75,
 *                                                                         torch.gradient, unit weights)
 *   flow [2, H, W]; d_flow [2, H, W] is OVERWRITTEN with the gradient; partials: device doubles,
 *   ebos_flow_regularisers_partials() of them, whose sum is the value (ebos_cmax_adam_step sums them).
 *   var_partials (nullable) .. moments: a side job for one of its workgroups -- reduce the variance partials that
 *   ebos_iwe_dense_slab_f32(want_variance = 2) left in its workspace (ebos_iwe_slab_partials locates them) into
 *   out_variance / moments, instead of a finalize launch of its own.
 * ebos_cmax_adam_step_f32: torch.optim.Adam (amsgrad off, no weight decay) on theta[n] given grad[n],
 *   with state exp_avg[n], exp_avg_sq[n] and a device step counter step[1] (all zero-initialised by the
 *   caller); records losses[step] = contrast_scale * contrast[0] + sum(reg_partials) for the parameters
 *   BEFORE the update (contrast / reg_partials / losses nullable) and increments step.  One workgroup.
 * ---------------------------------------------------------------------------------------- */
int ebos_flow_regularisers_partials(void);
int ebos_flow_regularisers_f32(const float* flow, int H, int W, float w_flow_norm, float w_image_gradient,
                               float* d_flow, double* partials, const double* var_partials, int64_t n_var_partials,
                               int64_t n_var_pixels, float* out_variance, double* moments, ebos_stream_t stream);
int ebos_cmax_adam_step_f32(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int n, double lr,
                            double beta1, double beta2, double eps, int* step, const float* contrast,
                            float contrast_scale, const double* reg_partials, int n_reg, float* losses,
                            int losses_cap, ebos_stream_t stream);

/* The whole loop natively: n_iter iterations of the six calls above, enqueued back to back on `stream` by one
 * C call (no interpreter between launches; asynchronous like everything else).  All buffers are the caller's:
 *   plan:     xs/ys/dts (nullable when the compact trio is given), grp_offsets/cpix/cdt, key_offsets, n, H, W, tile,
 *             halo, pad, omit_boundary, splits, part_table -- as ebos_iwe_dense_slab_f32 / ebos_iwe_dense_tiled_bwd_f32
 *   grid:     theta/d_theta/exp_avg/exp_avg_sq [2, gh, gw], step [1] int32, patch and sliding window
 *   images:   dense/d_dense [2, H, W], d_reg [2, H, W] (nullable iff both regulariser weights are 0),
 *             iwe [H + 2 pad_h, W + 2 pad_w], variance [1] f32, moments [2] f64, upstream [1] f32 = -w_variance
 *   scratch:  reg_partials [max(ebos_flow_regularisers_partials(), work items = ebos_patch_grad_partials_bytes / 2048)] f64,
 *             zero-filled once; upsample_scratch
 *             (ebos_upsample_bwd_scratch_bytes), workspace (ebos_iwe_slab_workspace_bytes, zero-filled once)
 *   losses:   [losses_cap] f32, entry `step` written per iteration (nullable)
 *   theta_mask: [gh, gw] f32, 0 = patch not estimated (nullable = all patches)                                 */
typedef struct ebos_cmax_patch_problem {
  const float *xs, *ys, *dts;
  const int32_t* grp_offsets;
  const uint16_t* cpix;
  const float* cdt;
  const int32_t* key_offsets;
  int64_t n;
  int H, W, tile_h, tile_w, halo, pad_h, pad_w, omit_boundary;
  int splits;                  /* as ebos_iwe_dense_slab_f32; 0 = adaptive work items of part_table */
  const int32_t* part_table;   /* nullable unless splits == 0 */
  int gh, gw, patch_h, patch_w, slide_h, slide_w;
  float w_variance, w_flow_norm, w_image_gradient;
  float w_gradient_magnitude;  /* contrast = gradient magnitude instead of variance: exactly one of the two
                                  contrast weights is non-zero; then `variance` receives the contrast value, `upstream` holds
                                  -w_gradient_magnitude, and d_iwe / cost_scratch below are needed */
  double lr, beta1, beta2, eps;
  float *theta, *d_theta, *exp_avg, *exp_avg_sq;
  int* step;
  int steps_done;              /* Adam steps already applied to theta (the first iteration of a solve is step steps_done + 1) */
  float *dense, *d_dense, *d_reg, *iwe, *variance;
  float* d_iwe;                /* [H + 2 pad_h, W + 2 pad_w]; nullable unless w_gradient_magnitude != 0 */
  void* cost_scratch;          /* >= ebos_cmax_cost_scratch_bytes(H + 2 pad_h, W + 2 pad_w) (the fused Sobel pass's value partials / the
                                  blur's partial pairs); nullable unless w_gradient_magnitude != 0 or blur_k0 != 0 */
  size_t cost_scratch_bytes;
  double* moments;
  const float* upstream;
  double* reg_partials;
  float* upsample_scratch;
  void* workspace;
  size_t workspace_bytes;
  float* losses;
  int losses_cap;
  const float* theta_mask;     /* [gh, gw], nullable: grad_mask of ebos_upsample_patch_flow_bwd_adam_f32 */
  float* grad_partials;        /* non-NULL: the event kernels sample the patch grid themselves (ebos_iwe_patch_*; needs
                                  ebos_patch_fused_supported and the compact plan); `dense` is then only written when a flow
                                  regulariser is on, d_dense / upsample_scratch are not used */
  size_t grad_partials_bytes;  /* ebos_patch_grad_partials_bytes(H, W, tile_h, tile_w, splits == 0) */
  /* ABI 2: iwe.blur_sigma > 0 -- the variance is taken on the 3-tap blurred IWE (src/event_image_converter.py:399-404).
   * blur_k0 = 0: no blur.  Otherwise the taps (blur_k0, blur_k1, blur_k0), blur_image [H + 2 pad_h, W + 2 pad_w] and cost_scratch
   * >= 16 * ebos_blur3_variance_partials(H + 2 pad_h, W + 2 pad_w) bytes; the variance contrast only. */
  float blur_k0, blur_k1;
  float* blur_image;
  /* the resident launch on a window of FRACTIONAL source coordinates (sub-pixel rectified or pre-warped events):
   * with cfx / cfy non-NULL, grp_offsets / cpix / cdt / cfx / cfy are the arrays of ebos_plan_compact_frac_f32.  The resident launch
   * reads them, and so does the four-launch loop when grad_partials is given (grid-sampling route on the fractions: general event
   * loops); with grad_partials NULL the launches run the dense route on xs / ys / dts. */
  const float *cfx, *cfy;
} ebos_cmax_patch_problem;
int ebos_cmax_patch_solve_f32(const ebos_cmax_patch_problem* problem, int n_iter, ebos_stream_t stream);
/* Several independent windows at once (windows are the unit that shards): problem w runs on
 * streams[w]; launches are enqueued iteration-major so that the windows' kernels interleave on the GPU. */
int ebos_cmax_patch_solve_many_f32(const ebos_cmax_patch_problem* problems, const ebos_stream_t* streams,
                                   int n_problems, int n_iter);

/* The Adam loop of the 2-DoF motion model natively (motion_model "2d-translation" with optimizer Adam, n_iter 600 and
 * iwe.blur_sigma 3 is what configs/hot_plate1.yaml:47,65,70 selects; loop: src/solver/generative_max_likelihood.py:306-341):
 *     loss(theta) = -w_variance * var([blur3] IWE(x + dt theta)),   theta = (trans_x, trans_y)   (src/warp.py:364-383)
 * n_iter iterations enqueued by one C call, four launches each (five with the blur), no host synchronisation.
 *   plan:    xs / ys / dts and / or the compact trio (grp_offsets / cpix / cdt), key_offsets, tile, halo, pad, omit_boundary,
 *            splits, part_table as ebos_iwe_2dof_slab_f32
 *   state:   theta / d_theta / exp_avg / exp_avg_sq [2] f32, step [1] int32; steps_done = Adam steps already applied
 *   images:  iwe [H + 2 pad_h, W + 2 pad_w]; blur_image of the same shape and cost_scratch >= 16 * ebos_blur3_variance_partials
 *            bytes when blur_k0 != 0 (taps blur_k0, blur_k1, blur_k0)
 *   scalars: variance [1] f32 and moments [2] f64 receive the contrast of the last iteration; upstream [1] f32 = -w_variance
 *   losses:  [losses_cap] f32, entry `step` written per iteration with the loss BEFORE the update (nullable) */
typedef struct ebos_cmax_2dof_problem {
  const float *xs, *ys, *dts;  /* the plan's (x, y, dt) arrays: needed when the compact trio is NULL (fractional -- undistorted --
                                  source coordinates), else nullable */
  const int32_t* grp_offsets;
  const uint16_t* cpix;
  const float* cdt;
  const float *cfx, *cfy;      /* compact plan of FRACTIONAL source coordinates (ebos_plan_compact_frac_f32): x - floor(x), y - floor(y)
                                  per slot, laid out like cdt; both or neither (else EBOS_ERR_INVALID_ARG; ..._resident_supported says 0); NULL = integer
                                  source pixels.  Read by BOTH forms of the loop: ebos_cmax_2dof_solve_f32's launches take the compact
                                  trio + cfx / cfy when they are given and xs / ys / dts only when the trio is NULL */
  const int32_t* key_offsets;
  int64_t n;
  int H, W, tile_h, tile_w, halo, pad_h, pad_w, omit_boundary;
  int splits;
  const int32_t* part_table;
  float w_variance;            /* > 0; the device word `upstream` below holds -w_variance (the four launches read the device word, the
                                  resident launch the host value) */
  float blur_k0, blur_k1;
  double lr, beta1, beta2, eps;
  float *theta, *d_theta, *exp_avg, *exp_avg_sq;
  int* step;
  int steps_done;
  float *iwe, *blur_image, *variance;
  double* moments;
  const float* upstream;
  void* cost_scratch;
  size_t cost_scratch_bytes;
  void* workspace;
  size_t workspace_bytes;
  float* losses;
  int losses_cap;
} ebos_cmax_2dof_problem;
int ebos_cmax_2dof_solve_f32(const ebos_cmax_2dof_problem* problem, int n_iter, ebos_stream_t stream);


/* ---- the same loop as ONE resident launch (csrc/cmax_resident.hip) -------------------------------------------------------------
 * Replaces, for one window, the n_iter x 4 launches that ebos_cmax_patch_solve_f32 enqueues for the loop of
 * src/solver/generative_max_likelihood.py:306-341 (600 iterations over the SAME events, configs/hot_plate1.yaml:70): one
 * 1024-thread workgroup per source tile stays resident for all iterations, keeps its tile's geometry, grid cells and Adam state
 * in registers / LDS, and exchanges only slabs, (sum, sum of squares) records and partial cell gradients with its neighbours
 * through `workspace`, `grad_partials` and `mailbox`.  Same `problem` struct, same results (IWE bit-identical; losses to ~1e-7).
 *
 *   ebos_cmax_resident_supported   1 when `problem` can run resident: a compact plan -- of the grid-sampling route (grad_partials), or
 *                                  with the fractions of undistorted events (cfx / cfy) --, either contrast (the blurred image with
 *                                  the variance only), splits 0 or 1, image padding below half a tile and at most the halo (round 6: the windows
 *                                  then reach at least the padding ring), tile / halo with a resident kernel ((45, 80, 32),
 *                                  (32, 32, 32), (32, 64, 32)), cell blocks of <= 256 elements per tile, cells whose supports span
 *                                  <= 16 tiles per axis; 0 otherwise (reason: ebos_last_error)
 *   mailbox                        device memory of ebos_cmax_resident_mailbox_bytes(...): flags, records, the status word;
 *                                  cleared by every call
 *   spin_timeout_s                 cap of every in-kernel wait (seconds; e.g. 2.0).  The grid must be co-resident -- the call
 *                                  checks tiles <= CUs x occupancy and orders resident launches of one device behind each other --
 *                                  but if a wait still passes the cap (another process's resident grid interleaved with this
 *                                  one), or a tap leaves the largest LDS window (the spill path of the four-launch pipeline),
 *                                  the launch ENDS instead of hanging and leaves theta / exp_avg / exp_avg_sq / step untouched --
 *                                  except after a spill in iteration k >= 1, when it hands over the state of its k completed
 *                                  iterations (ebos_cmax_resident_iterations = k; continue with ebos_cmax_patch_solve_f32 and
 *                                  steps_done + k).  The first verdict of a launch stands; should its workgroups have left on
 *                                  two different ones, ebos_cmax_resident_iterations returns -1: the state is partly written.
 *   ebos_cmax_resident_status      synchronises `stream`, returns EBOS_OK or a negative code (-101 spin cap, -102 spill,
 *                                  -103 geometry, -104 a crowded window: on a sensor of >= 128 tiles the fullest tile holds more
 *                                  than 60 k + 0.8 % of the window's events, on smaller ones more than 85 k -- the kernel's own
 *                                  verdict in its first iteration; EBOS_RESIDENT_MAX_IMBALANCE = r > 0: more than r x the average
 *                                  tile's events (and >= 32 k), or >= 120 k and more than r / 4 x; 0 = never); on a negative code run
 *                                  ebos_cmax_patch_solve_f32 with the same problem.  -102 is also how the blurred and the
 *                                  gradient-magnitude loops hand over when the windows outgrow their LDS regions (~12 px).  */
size_t ebos_cmax_resident_mailbox_bytes(int H, int W, int tile_h, int tile_w);
int ebos_cmax_resident_supported(const ebos_cmax_patch_problem* problem);
int ebos_cmax_patch_solve_resident_f32(const ebos_cmax_patch_problem* problem, int n_iter, void* mailbox, size_t mailbox_bytes,
                                       double spin_timeout_s, ebos_stream_t stream);
/* The 2-DoF Adam loop (ebos_cmax_2dof_solve_f32) as ONE resident launch: same problem struct, same mailbox / spin cap / status
 * protocol as ebos_cmax_patch_solve_resident_f32 (ebos_cmax_resident_status / _iterations read its mailbox too).  Compact plans,
 * splits <= 1, image padding below half a tile, the tiles (45, 80) / (32, 32) / (32, 64) with halo 32.  Every workgroup sums all tiles' partial pairs of
 * d loss / d theta itself and steps the two parameters redundantly; with blur_k0 != 0 the gathered window is blurred in LDS
 * (windows up to ~12 px; larger displacements hand over with -102 like a spill). */
int ebos_cmax_2dof_resident_supported(const ebos_cmax_2dof_problem* problem);
int ebos_cmax_2dof_solve_resident_f32(const ebos_cmax_2dof_problem* problem, int n_iter, void* mailbox, size_t mailbox_bytes,
                                      double spin_timeout_s, ebos_stream_t stream);
int ebos_cmax_resident_status(const void* mailbox, ebos_stream_t stream);
/* The iterations the launch completed (synchronises `stream`): n_iter after status 0; after -102 (a tap left the largest LDS
 * window in iteration k >= 1) the k iterations before it -- theta, the optimiser state, the step counter and the losses are then
 * those of k iterations and ebos_cmax_patch_solve_f32 continues with the remaining n_iter - k; 0 after any other negative status
 * (nothing changed). */
int ebos_cmax_resident_iterations(const void* mailbox, ebos_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EBOS_HIP_H */
