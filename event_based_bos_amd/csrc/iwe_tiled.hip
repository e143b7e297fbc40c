// iwe_tiled.hip -- tile-private fused warp + IWE pipeline for gfx950 (the fast path of the hot loop).
//
// Events are binned by source tile (event_plan.hip / plan_lean.hip).  One 1024-thread workgroup owns one work item (a tile, or a
// part of a heavy tile):
//
//   forward   iwe_slab_accumulate_kernel   compact events (u16 tile-local pixel + f32 dt, 16-byte loads, software-pipelined: group k+2
//                                          is loading and the flow of group k+1 is being gathered while group k's taps are
//                                          accumulated) -> warp in registers -> paired fixed point, two ds_add_u64 per event, into an
//                                          LDS image of the tile + HALO px per side (verified by a checksum; an exact f64 redo if a
//                                          field overflowed; per-event weights: ds_add_f64) -> the LDS image is written ONCE as a
//                                          coalesced f32 "slab", write-through (no global atomics, no memset)
//             iwe_slab_combine4_kernel     per pixel: sum the <= 9 slabs that cover it (+ the spill image of beyond-halo taps when
//                                          this call wrote any), write the IWE, and reduce the variance moments (sum, sum of
//                                          squares; f64) of its 4 x 256 pixels -> partials
//             moments_finalize_kernel      one workgroup: partials -> (mean, M, variance); deterministic
//             *_batch_kernel               the same three over (work item, window) for up to 16 independent windows
//   backward  iwe_dense_tiled_bwd_kernel   upstream image tile (+halo) staged in LDS with all its loads in flight, four LDS gathers per
//                                          event, the events of one source pixel summed in registers per lane (fixed point: one
//                                          ds_add_u64 per run; ds_add_f64 pairs with per-event weights) into a [2][TH][TW] LDS
//                                          tile, then d_flow of the tile is written once, 16 bytes per lane,
//                                          write-through -- every flow pixel belongs to exactly one tile, so no global atomics and
//                                          no zero-fill.
//   GRID variants of both: the flow argument is a patch grid [2, gh, gw], evaluated per tile into LDS (patch_grid.h).
//
// Why 64-bit LDS atomics: ds_add_f32 is ~8-15x slower than ds_add_u64 / ds_add_f64 on gfx950 (tools/ubench_lds.hip,
// profiles/r02_lds_cost_and_phases.txt).  Why slabs: a global float atomic costs ~50 ns per 256-B wave instruction per CU at the
// memory side and same-address atomics serialise (~88/us), whereas stores stream at HBM rate.  What bounds the loops, what was
// tried and what it was worth: DESIGN.md 4.1.
//
// reference semantics: src/warp.py:330-342 + src/event_image_converter.py:581-620 (forward);
// their autograd w.r.t. the flow and the per-event weight (backward).
// The device code lives in iwe_tile_core.h (shared with cmax_resident.hip); this file holds the launchers and the C ABI.
#include "iwe_tiled_launch.h"

namespace ebos {

// SpillEpoch: every forward call gets a number of its own (never 0: a zero-filled workspace matches no call).  Host-side state
// only; a replayed HIP graph repeats its number, which can only make a combine pass read an all-zero spill image it could skip.
unsigned next_spill_epoch() {
  static std::atomic<unsigned> counter{0};
  unsigned e = counter.fetch_add(1, std::memory_order_relaxed) + 1;
  if (e == 0) e = counter.fetch_add(1, std::memory_order_relaxed) + 1;
  return e;
}

namespace {

// stream-ordering events of the batched entry (fork to / join from its tail stream): a small ring of timing-free events, created
// on first use.  Re-recording an event does not disturb the waits already enqueued on its previous record.
// (an event that could not be created is nullptr: order_after then reports the failure)
inline hipEvent_t next_order_event() {
  constexpr int kRing = 64, kMaxDevices = 64;
  struct Ring {
    hipEvent_t ev[kRing];
    std::atomic<unsigned> created{0}, cursor{0};
    std::atomic_flag lock = ATOMIC_FLAG_INIT;
  };
  static Ring rings[kMaxDevices];  // an event belongs to the device it was created on: one ring per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  Ring& r = rings[dev];
  if (r.created.load(std::memory_order_acquire) == 0) {
    while (r.lock.test_and_set(std::memory_order_acquire)) {}
    if (r.created.load(std::memory_order_relaxed) == 0) {
      for (int i = 0; i < kRing; ++i)
        if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) r.ev[i] = nullptr;
      r.created.store(1, std::memory_order_release);
    }
    r.lock.clear(std::memory_order_release);
  }
  return r.ev[r.cursor.fetch_add(1, std::memory_order_relaxed) % kRing];
}

}  // namespace

// make `later` wait for everything enqueued on `earlier` so far
int order_after(hipStream_t later, hipStream_t earlier, const char* what) {
  hipEvent_t ev = next_order_event();
  if (ev == nullptr || hipEventRecord(ev, earlier) != hipSuccess || hipStreamWaitEvent(later, ev, 0) != hipSuccess) {
    set_error("%s: cannot order the tail stream (%s)", what, hipGetErrorString(hipGetLastError()));
    return EBOS_ERR_LAUNCH;
  }
  return EBOS_OK;
}

const SlabOps* slab_ops_64x64x32();
const SlabOps* slab_ops_45x80x32();
const SlabOps* slab_ops_32x64x32();
const SlabOps* slab_ops_32x32x32();
const SlabOps* slab_ops_64x64x16();
const SlabOps* slab_ops_45x80x16();
const SlabOps* slab_ops_32x32x16();
const SlabOps* slab_ops_32x32x8();
const SlabOps* slab_ops(int th, int tw, int halo) {
  static const SlabOps* const all[] = {slab_ops_64x64x32(), slab_ops_45x80x32(), slab_ops_32x64x32(), slab_ops_32x32x32(), slab_ops_64x64x16(), slab_ops_45x80x16(), slab_ops_32x32x16(), slab_ops_32x32x8()};
  for (const SlabOps* o : all)
    if (o->th == th && o->tw == tw && o->halo == halo) return o;
  return nullptr;
}

namespace {

bool slab_config_ok(int th, int tw, int halo) {
  for (int i = 0; i < kNumSlabConfigs; ++i)
    if (kSlabConfigs[i].th == th && kSlabConfigs[i].tw == tw && kSlabConfigs[i].halo == halo) return true;
  return false;
}

}  // namespace
}  // namespace ebos

#define EBOS_SLAB_DISPATCH(CALL)                                             \
  if (tile_h == 64 && tile_w == 64 && halo == 32) { rc = CALL(64, 64, 32); } \
  else if (tile_h == 45 && tile_w == 80 && halo == 32) { rc = CALL(45, 80, 32); } \
  else if (tile_h == 32 && tile_w == 64 && halo == 32) { rc = CALL(32, 64, 32); } \
  else if (tile_h == 32 && tile_w == 32 && halo == 32) { rc = CALL(32, 32, 32); } \
  else if (tile_h == 64 && tile_w == 64 && halo == 16) { rc = CALL(64, 64, 16); } \
  else if (tile_h == 45 && tile_w == 80 && halo == 16) { rc = CALL(45, 80, 16); } \
  else if (tile_h == 32 && tile_w == 32 && halo == 16) { rc = CALL(32, 32, 16); } \
  else if (tile_h == 32 && tile_w == 32 && halo == 8) { rc = CALL(32, 32, 8); }

extern "C" {

int ebos_slab_config(int* out, int cap) {
  using namespace ebos;
  for (int i = 0; i < kNumSlabConfigs && i < cap && out != nullptr; ++i) {
    out[3 * i] = kSlabConfigs[i].th;
    out[3 * i + 1] = kSlabConfigs[i].tw;
    out[3 * i + 2] = kSlabConfigs[i].halo;
  }
  return kNumSlabConfigs;
}

int ebos_halo_auto(int max_halo, double dt_bound) {
  if (max_halo <= 0 || max_halo > 255 || !(dt_bound >= 0.0) || dt_bound > 4096.0) return max_halo;  // (no auto: the built halo itself)
  int q = (int)(dt_bound * 64.0);
  if ((double)q < dt_bound * 64.0) ++q;  // rounded UP: the bound stays a bound
  return EBOS_HALO_AUTO(max_halo, q < 1 ? 1 : q);
}

int ebos_iwe_slab_partials(int H, int W, int tile_h, int tile_w, int halo_arg, int splits, int pad_h, int pad_w, int omit_boundary,
                           size_t* offset_bytes, int64_t* n_partials, int64_t* n_pixels) {
  using namespace ebos;
  const int halo = decode_halo(halo_arg).halo;
  EBOS_REQUIRE(offset_bytes && n_partials && n_pixels, "ebos_iwe_slab_partials: NULL output");
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && halo >= 0 && splits >= 0 && pad_h >= 0 && pad_w >= 0,
               "ebos_iwe_slab_partials: bad sizes");
  const SlabLayout L = slab_layout(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  *offset_bytes = L.off_partials;
  // the grid of the combine pass that ebos_iwe_*_slab_f32 launches for this geometry (one partial pair per workgroup)
  if (L.w % 4 == 0 && pad_w % 4 == 0) *n_partials = (int64_t)((L.w / 4 + 63) / 64) * ((L.h + kCombineRows - 1) / kCombineRows);
  else *n_partials = (int64_t)((L.w + kCombineBlock - 1) / kCombineBlock) * L.h;
  const int lo = omit_boundary ? 1 : 0;
  *n_pixels = (int64_t)(L.h - 2 * lo > 0 ? L.h - 2 * lo : 0) * (L.w - 2 * lo > 0 ? L.w - 2 * lo : 0);
  return EBOS_OK;
}

size_t ebos_iwe_slab_workspace_bytes(int H, int W, int tile_h, int tile_w, int halo_arg, int splits, int pad_h, int pad_w) {
  using namespace ebos;
  const int halo = decode_halo(halo_arg).halo;
  if (H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0 || halo < 0 || splits < 0 || pad_h < 0 || pad_w < 0) return 0;
  return slab_layout(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w).total;
}

static int iwe_slab_entry(const ebos::GridSrc* grid_src, const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* grp_offsets,
                            const uint16_t* cpix, const float* cdt,
                            const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                            int tile_w, int halo_arg, int splits, int pad_h, int pad_w, void* workspace,
                            size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary, float* out_variance,
                            double* moments, const int32_t* part_table, ebos_stream_t stream, const float* cfx = nullptr,
                            const float* cfy = nullptr) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(flow && iwe && key_offsets && workspace, "ebos_iwe_dense_slab: NULL flow/iwe/key_offsets/workspace");
  EBOS_REQUIRE((cfx == nullptr) == (cfy == nullptr) && (cfx == nullptr || (grid_src != nullptr && grp_offsets && cpix && cdt)),
               "ebos_iwe_patch_slab_frac: cfx / cfy come together, with the compact arrays, on the grid-sampling entry");
  EBOS_REQUIRE(((xs && ys && dts) || (grp_offsets && cpix && cdt)) || n == 0, "ebos_iwe_dense_slab: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && splits >= 0 && splits <= 64,
               "ebos_iwe_dense_slab: bad sizes (splits=%d)", splits);
  EBOS_REQUIRE(splits != 0 || part_table, "ebos_iwe_dense_slab: splits = 0 (adaptive work items) needs the plan's part_table");
  EBOS_REQUIRE(want_variance != 1 || out_variance || moments, "ebos_iwe_dense_slab: variance requested without an output");
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_dense_slab: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  if (workspace_bytes < need) {
    set_error("ebos_iwe_dense_slab: workspace too small (%zu < %zu)", workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int n_tiles_ = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  const EvPtrs evp{xs, ys, dts, weight, grp_offsets, cpix, cdt, part_table, part_table ? part_table + n_tiles_ + 1 : nullptr,
                   part_table ? part_table + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr, cfx, cfy};
  static const int acc_mode = [] {  // EBOS_SLAB_ACC=f64 forces the f64 accumulator (debug / A-B runs)
    const char* e = getenv("EBOS_SLAB_ACC");
    return (e && e[0] == 'f') ? (int)ACC_F64 : (int)ACC_FX;
  }();
  int rc = EBOS_ERR_UNSUPPORTED;
  rc = slab_ops(tile_h, tile_w, halo)->fwd(evp, key_offsets, flow, false, H, W, splits, pad_h, pad_w, ws, iwe, want_variance, omit_boundary,
                                           out_variance, moments, acc_mode, s, grid_src, ha);
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_dense_slab");
  return EBOS_OK;
}

int ebos_iwe_dense_slab_f32(const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* grp_offsets,
                            const uint16_t* cpix, const float* cdt,
                            const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                            int tile_w, int halo, int splits, int pad_h, int pad_w, void* workspace,
                            size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary, float* out_variance,
                            double* moments, const int32_t* part_table, ebos_stream_t stream) {
  return iwe_slab_entry(nullptr, xs, ys, dts, weight, grp_offsets, cpix, cdt, key_offsets, n, flow, H, W, tile_h, tile_w, halo, splits,
                        pad_h, pad_w, workspace, workspace_bytes, iwe, want_variance, omit_boundary, out_variance, moments, part_table,
                        stream);
}

int ebos_patch_fused_supported(int tile_h, int tile_w, int halo_arg, int slide_h, int slide_w) {
  using namespace ebos;
  const int halo = decode_halo(halo_arg).halo;
  int rc = 0;
#define EBOS_CALL(TH, TW, HL) \
  ((grid_fwd_fits<TH, TW, HL>() && grid_bwd_fits<TH, TW, HL>() && (TH + 2 * kBwdApron) / slide_h + 3 <= kGridCells && \
    (TW + 2 * kBwdApron) / slide_w + 3 <= kGridCells)                                                                   \
       ? 1                                                                                                              \
       : 0)
  if (slide_h <= 0 || slide_w <= 0) return 0;
  EBOS_SLAB_DISPATCH(EBOS_CALL)
#undef EBOS_CALL
  return rc;
}

int ebos_iwe_patch_slab_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets,
                            int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w, int H,
                            int W, int tile_h, int tile_w, int halo, int splits, int pad_h, int pad_w, void* workspace,
                            size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary, float* out_variance,
                            double* moments, const int32_t* part_table, ebos_stream_t stream) {
  return ebos_iwe_patch_slab_frac_f32(grp_offsets, cpix, cdt, nullptr, nullptr, key_offsets, n, grid, gh, gw, patch_h, patch_w, slide_h,
                                      slide_w, H, W, tile_h, tile_w, halo, splits, pad_h, pad_w, workspace, workspace_bytes, iwe,
                                      want_variance, omit_boundary, out_variance, moments, part_table, stream);
}

int ebos_iwe_patch_slab_frac_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const float* cfx, const float* cfy,
                                 const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w,
                                 int slide_h, int slide_w, int H, int W, int tile_h, int tile_w, int halo, int splits, int pad_h,
                                 int pad_w, void* workspace, size_t workspace_bytes, float* iwe, int want_variance, int omit_boundary,
                                 float* out_variance, double* moments, const int32_t* part_table, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(grid && gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0,
               "ebos_iwe_patch_slab: bad patch grid (%dx%d, patch %dx%d, slide %dx%d)", gh, gw, patch_h, patch_w, slide_h, slide_w);
  EBOS_REQUIRE(grp_offsets && cpix && cdt, "ebos_iwe_patch_slab: needs the compact plan (grp_offsets / cpix / cdt)");
  if (!ebos_patch_fused_supported(tile_h, tile_w, halo, slide_h, slide_w)) {
    set_error("ebos_iwe_patch_slab: tile %dx%d halo %d with sliding window %dx%d is outside ebos_patch_fused_supported", tile_h, tile_w,
              halo, slide_h, slide_w);
    return EBOS_ERR_UNSUPPORTED;
  }
  const GridSrc gs{make_axis(gh, patch_h, slide_h, H), make_axis(gw, patch_w, slide_w, W)};
  return iwe_slab_entry(&gs, nullptr, nullptr, nullptr, nullptr, grp_offsets, cpix, cdt, key_offsets, n, grid, H, W, tile_h, tile_w, halo,
                        splits, pad_h, pad_w, workspace, workspace_bytes, iwe, want_variance, omit_boundary, out_variance, moments,
                        part_table, stream, cfx, cfy);
}

int ebos_iwe_slab_batch_f32(const ebos_slab_window* windows, int n_windows, int gh, int gw, int patch_h, int patch_w, int slide_h,
                            int slide_w, int H, int W, int tile_h, int tile_w, int halo_arg, int splits, int pad_h, int pad_w,
                            size_t workspace_bytes, int want_variance, int omit_boundary, ebos_stream_t stream,
                            ebos_stream_t tail_stream) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(windows && n_windows >= 0, "ebos_iwe_slab_batch: NULL windows");
  EBOS_REQUIRE(H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && splits >= 0 && splits <= 64, "ebos_iwe_slab_batch: bad sizes (splits=%d)",
               splits);
  const bool patch = gh > 0 || gw > 0;
  if (patch) {
    EBOS_REQUIRE(gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0,
                 "ebos_iwe_slab_batch: bad patch grid (%dx%d, patch %dx%d, slide %dx%d)", gh, gw, patch_h, patch_w, slide_h, slide_w);
    if (!ebos_patch_fused_supported(tile_h, tile_w, halo, slide_h, slide_w)) {
      set_error("ebos_iwe_slab_batch: tile %dx%d halo %d with sliding window %dx%d is outside ebos_patch_fused_supported", tile_h, tile_w,
                halo, slide_h, slide_w);
      return EBOS_ERR_UNSUPPORTED;
    }
  }
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_slab_batch: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  if (workspace_bytes < need) {
    set_error("ebos_iwe_slab_batch: workspaces too small (%zu < %zu)", workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  const SlabLayout L = slab_layout(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  const int n_tiles_ = L.tiles_y * L.tiles_x;
  GridSrc gs{};
  if (patch) gs = GridSrc{make_axis(gh, patch_h, slide_h, H), make_axis(gw, patch_w, slide_w, W)};
  hipStream_t s = as_stream(stream);
  hipStream_t s_tail = tail_stream ? as_stream(tail_stream) : s;
  for (int first = 0; first < n_windows; first += kMaxBatch) {
    const int n = n_windows - first < kMaxBatch ? n_windows - first : kMaxBatch;
    FwdBatch b{};
    for (int k = 0; k < n; ++k) {
      const ebos_slab_window& q = windows[first + k];
      EBOS_REQUIRE(q.grp_offsets && q.cpix && q.cdt && q.key_offsets && q.flow && q.workspace && q.iwe,
                   "ebos_iwe_slab_batch: window %d: NULL plan / flow / workspace / iwe (compact plans with unit weights only)", first + k);
      EBOS_REQUIRE(splits != 0 || q.part_table, "ebos_iwe_slab_batch: window %d: splits = 0 needs the plan's part_table", first + k);
      EBOS_REQUIRE(want_variance != 1 || q.out_variance || q.moments, "ebos_iwe_slab_batch: window %d: variance without an output",
                   first + k);
      char* ws = reinterpret_cast<char*>(q.workspace);
      const int32_t* pt = q.part_table;
      FwdWindow& w = b.w[k];
      w.ev = EvPtrs{nullptr, nullptr, nullptr, nullptr, q.grp_offsets, q.cpix, q.cdt, pt, pt ? pt + n_tiles_ + 1 : nullptr,
                    pt ? pt + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr};
      w.key_offsets = q.key_offsets;
      w.flow = q.flow;
      w.slabs = reinterpret_cast<float*>(ws);
      w.spill = reinterpret_cast<float*>(ws + L.off_spill);
      w.partials = reinterpret_cast<double*>(ws + L.off_partials);
      w.spill_epoch = reinterpret_cast<unsigned*>(ws + L.off_epoch);
      w.halo_tab = reinterpret_cast<unsigned*>(ws + L.off_halo);
      w.iwe = q.iwe;
      w.out_var = q.out_variance;
      w.moments = q.moments;
    }
    int rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->fwd_batch(b, n, H, W, splits, pad_h, pad_w, want_variance, omit_boundary, s, s_tail,
                                                   patch ? &gs : nullptr, ha, false);
    if (rc != EBOS_OK) return rc;
  }
  if (s_tail != s && n_windows > 0)  // join: work enqueued on `stream` after this call sees every window's results
    if (int rc = order_after(s, s_tail, "ebos_iwe_slab_batch")) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_slab_batch");
  return EBOS_OK;
}

int ebos_iwe_2dof_slab_f32(const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* grp_offsets,
                            const uint16_t* cpix, const float* cdt,
                           const int32_t* key_offsets, int64_t n, const float* thetas, int K, int H, int W, int tile_h,
                           int tile_w, int halo_arg, int splits, int pad_h, int pad_w, void* workspace, size_t workspace_bytes,
                           float* iwes, int want_variance, int omit_boundary, float* out_variance, double* moments,
                           const int32_t* part_table, ebos_stream_t stream) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(thetas && iwes && key_offsets && workspace, "ebos_iwe_2dof_slab: NULL thetas/iwes/key_offsets/workspace");
  EBOS_REQUIRE(((xs && ys && dts) || (grp_offsets && cpix && cdt)) || n == 0, "ebos_iwe_2dof_slab: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && K >= 1 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && splits >= 0 && splits <= 64,
               "ebos_iwe_2dof_slab: bad sizes (K=%d splits=%d)", K, splits);
  EBOS_REQUIRE(splits != 0 || part_table, "ebos_iwe_2dof_slab: splits = 0 (adaptive work items) needs the plan's part_table");
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_2dof_slab: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  if (workspace_bytes < need) {
    set_error("ebos_iwe_2dof_slab: workspace too small (%zu < %zu)", workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  const int n_tiles_ = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  const EvPtrs evp{xs, ys, dts, weight, grp_offsets, cpix, cdt, part_table, part_table ? part_table + n_tiles_ + 1 : nullptr,
                   part_table ? part_table + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr};
  const int64_t hw = (int64_t)(H + 2 * pad_h) * (W + 2 * pad_w);
  for (int k = 0; k < K; ++k) {  // hypotheses reuse the workspace in stream order
    int rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->fwd(evp, key_offsets, thetas + 2 * k, true, H, W, splits, pad_h, pad_w, ws, iwes + k * hw,
                                             want_variance, omit_boundary, out_variance ? out_variance + k : nullptr,
                                             moments ? moments + 2 * k : nullptr, (int)ACC_FX, s, nullptr, ha);
    if (rc != EBOS_OK) return rc;
  }
  EBOS_CHECK_LAUNCH("ebos_iwe_2dof_slab");
  return EBOS_OK;
}

int ebos_iwe_2dof_slab_batch_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets,
                                 int64_t n, const float* thetas, int K, int H, int W, int tile_h, int tile_w, int halo_arg, int splits,
                                 int pad_h, int pad_w, void* workspaces, size_t workspace_bytes, float* iwes, int want_variance,
                                 int omit_boundary, float* out_variance, double* moments, const int32_t* part_table,
                                 ebos_stream_t stream, ebos_stream_t tail_stream) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(thetas && iwes && key_offsets && workspaces && grp_offsets && cpix && cdt,
               "ebos_iwe_2dof_slab_batch: NULL thetas / iwes / workspaces / compact plan");
  EBOS_REQUIRE(n >= 0 && K >= 1 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && splits >= 0 && splits <= 64,
               "ebos_iwe_2dof_slab_batch: bad sizes (K=%d splits=%d)", K, splits);
  EBOS_REQUIRE(splits != 0 || part_table, "ebos_iwe_2dof_slab_batch: splits = 0 (adaptive work items) needs the plan's part_table");
  EBOS_REQUIRE(want_variance != 1 || out_variance || moments, "ebos_iwe_2dof_slab_batch: variance requested without an output");
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_2dof_slab_batch: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  if (workspace_bytes < need || workspace_bytes % 256 != 0) {
    set_error("ebos_iwe_2dof_slab_batch: every hypothesis needs its own workspace of %zu bytes (a multiple of 256), got %zu", need,
              workspace_bytes);
    return EBOS_ERR_SCRATCH;
  }
  const SlabLayout L = slab_layout(H, W, tile_h, tile_w, halo, splits, pad_h, pad_w);
  const int n_tiles_ = L.tiles_y * L.tiles_x;
  const int64_t hw = (int64_t)L.h * L.w;
  hipStream_t s = as_stream(stream);
  hipStream_t s_tail = tail_stream ? as_stream(tail_stream) : s;
  const EvPtrs evp{nullptr, nullptr, nullptr, nullptr, grp_offsets, cpix, cdt, part_table, part_table ? part_table + n_tiles_ + 1 : nullptr,
                   part_table ? part_table + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr};
  for (int first = 0; first < K; first += kMaxBatch) {
    const int nb = K - first < kMaxBatch ? K - first : kMaxBatch;
    FwdBatch b{};
    for (int k = 0; k < nb; ++k) {
      char* ws = reinterpret_cast<char*>(workspaces) + (size_t)(first + k) * workspace_bytes;
      FwdWindow& w = b.w[k];
      w.ev = evp;
      w.key_offsets = key_offsets;
      w.flow = thetas + 2 * (first + k);
      w.slabs = reinterpret_cast<float*>(ws);
      w.spill = reinterpret_cast<float*>(ws + L.off_spill);
      w.partials = reinterpret_cast<double*>(ws + L.off_partials);
      w.spill_epoch = reinterpret_cast<unsigned*>(ws + L.off_epoch);
      w.halo_tab = reinterpret_cast<unsigned*>(ws + L.off_halo);
      w.iwe = iwes + (first + k) * hw;
      w.out_var = out_variance ? out_variance + (first + k) : nullptr;
      w.moments = moments ? moments + 2 * (first + k) : nullptr;
    }
    int rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->fwd_batch(b, nb, H, W, splits, pad_h, pad_w, want_variance, omit_boundary, s, s_tail, nullptr, ha, true);
    if (rc != EBOS_OK) return rc;
  }
  if (s_tail != s)  // join: work enqueued on `stream` after this call sees every hypothesis' results
    if (int rc = order_after(s, s_tail, "ebos_iwe_2dof_slab_batch")) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_2dof_slab_batch");
  return EBOS_OK;
}

int ebos_iwe_2dof_tiled_bwd_f32(const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* grp_offsets,
                            const uint16_t* cpix, const float* cdt,
                                const int32_t* key_offsets, int64_t n, const float* thetas, int K, int H, int W, int tile_h,
                                int tile_w, int halo_arg, int pad_h, int pad_w, const float* g_images, const float* affine,
                                int g_lo, float* d_thetas, void* workspace, size_t workspace_bytes, ebos_stream_t stream) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(thetas && g_images && d_thetas && key_offsets && workspace, "ebos_iwe_2dof_tiled_bwd: NULL argument");
  EBOS_REQUIRE(((xs && ys && dts) || (grp_offsets && cpix && cdt)) || n == 0, "ebos_iwe_2dof_tiled_bwd: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && K >= 1 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && g_lo >= 0, "ebos_iwe_2dof_tiled_bwd: bad sizes");
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_2dof_tiled_bwd: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const SlabLayout L = slab_layout(H, W, tile_h, tile_w, halo, 1, pad_h, pad_w);
  if (workspace_bytes < L.total) {
    set_error("ebos_iwe_2dof_tiled_bwd: workspace too small (%zu < %zu)", workspace_bytes, L.total);
    return EBOS_ERR_SCRATCH;
  }
  // the tile partials live in the slab section of the (forward) workspace: it is dead once the IWE is combined
  double* partials = reinterpret_cast<double*>(workspace);
  hipStream_t s = as_stream(stream);
  const EvPtrs evp{xs, ys, dts, weight, grp_offsets, cpix, cdt, nullptr, nullptr, nullptr};
  const int64_t hw = (int64_t)(H + 2 * pad_h) * (W + 2 * pad_w);
  for (int k = 0; k < K; ++k) {
    int rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->bwd(evp, key_offsets, thetas + 2 * k, true, H, W, pad_h, pad_w, g_images + k * hw,
                                             affine ? affine + 2 * k : nullptr, g_lo, d_thetas + 2 * k, nullptr, partials, nullptr, nullptr,
                                             nullptr, nullptr, s, nullptr, 0, 0.0f, 0.0f, nullptr, MomentsIn{}, ha, true);
    if (rc != EBOS_OK) return rc;
  }
  EBOS_CHECK_LAUNCH("ebos_iwe_2dof_tiled_bwd");
  return EBOS_OK;
}

// mj.partials != nullptr: the variance partials of the forward call (want_variance = 2) are reduced inside the kernel
static int dense_tiled_bwd_impl(const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* grp_offsets,
                                const uint16_t* cpix, const float* cdt, const int32_t* key_offsets, int64_t n, const float* flow, int H,
                                int W, int tile_h, int tile_w, int halo_arg, int pad_h, int pad_w, const float* g_image,
                                const float* affine, int g_lo, float* d_flow, float* d_weight, const double* var_moments,
                                const float* upstream, const float* addend, void* workspace, size_t workspace_bytes,
                                const int32_t* part_table, ebos_stream_t stream, const ebos::MomentsIn& mj) {
  using namespace ebos;
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(flow && g_image && d_flow && key_offsets, "ebos_iwe_dense_tiled_bwd: NULL flow/g_image/d_flow/key_offsets");
  EBOS_REQUIRE(((xs && ys && dts) || (grp_offsets && cpix && cdt)) || n == 0, "ebos_iwe_dense_tiled_bwd: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && g_lo >= 0, "ebos_iwe_dense_tiled_bwd: bad sizes");
  EBOS_REQUIRE(mj.partials != nullptr ? var_moments == nullptr : ((var_moments == nullptr) == (upstream == nullptr)),
               "ebos_iwe_dense_tiled_bwd: var_moments and upstream go together");
  float* part_out = nullptr;
  if (part_table != nullptr) {  // adaptive work items: partial tiles go through the slab section of the forward workspace
    const size_t need = ebos_iwe_slab_workspace_bytes(H, W, tile_h, tile_w, halo, 0, pad_h, pad_w);
    if (workspace == nullptr || workspace_bytes < need) {
      set_error("ebos_iwe_dense_tiled_bwd: part_table given but the workspace is missing or too small (%zu < %zu)",
                workspace_bytes, need);
      return EBOS_ERR_SCRATCH;
    }
    part_out = reinterpret_cast<float*>(workspace);
  }
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_iwe_dense_tiled_bwd: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  hipStream_t s = as_stream(stream);
  const int n_tiles_ = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  const EvPtrs evp{xs, ys, dts, weight, grp_offsets, cpix, cdt, part_table, part_table ? part_table + n_tiles_ + 1 : nullptr,
                   part_table ? part_table + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr};
  int rc = EBOS_ERR_UNSUPPORTED;
  rc = slab_ops(tile_h, tile_w, halo)->bwd(evp, key_offsets, flow, false, H, W, pad_h, pad_w, g_image, affine, g_lo, d_flow, d_weight, nullptr,
                                           var_moments, upstream, addend, part_out, s, nullptr, 0, 0.0f, 0.0f, nullptr, mj, ha, true);
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_dense_tiled_bwd");
  return EBOS_OK;
}

int ebos_iwe_dense_tiled_bwd_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                                 const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt,
                                 const int32_t* key_offsets, int64_t n, const float* flow, int H, int W,
                                 int tile_h,
                                 int tile_w, int halo, int pad_h, int pad_w, const float* g_image, const float* affine,
                                 int g_lo, float* d_flow, float* d_weight, const double* var_moments,
                                 const float* upstream, const float* addend, void* workspace, size_t workspace_bytes,
                                 const int32_t* part_table, ebos_stream_t stream) {
  return dense_tiled_bwd_impl(xs, ys, dts, weight, grp_offsets, cpix, cdt, key_offsets, n, flow, H, W, tile_h, tile_w, halo, pad_h, pad_w,
                              g_image, affine, g_lo, d_flow, d_weight, var_moments, upstream, addend, workspace, workspace_bytes,
                              part_table, stream, ebos::MomentsIn{});
}

int ebos_iwe_dense_tiled_bwd_blur_f32(const float* xs, const float* ys, const float* dts, const int32_t* grp_offsets, const uint16_t* cpix,
                                      const float* cdt, const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                                      int tile_w, int halo, int pad_h, int pad_w, const float* z_image, int g_lo, const float* upstream,
                                      const float* addend, float* d_flow, void* workspace, size_t workspace_bytes,
                                      const int32_t* part_table, const double* blur_partials, int64_t n_blur_partials,
                                      int64_t n_var_pixels, float* out_variance, double* out_moments, float blur_k0, float blur_k1,
                                      ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(blur_partials != nullptr && upstream != nullptr && n_blur_partials >= 1 && n_var_pixels >= 2 && blur_k0 > 0.0f && blur_k1 > 0.0f,
               "ebos_iwe_dense_tiled_bwd_blur: needs the partials of ebos_blur3_variance_adjoint_f32, upstream and positive taps");
  const MomentsIn mj{blur_partials, n_blur_partials, n_var_pixels, out_variance, out_moments, nullptr, 0, Blur3{blur_k0, blur_k1}};
  return dense_tiled_bwd_impl(xs, ys, dts, nullptr, grp_offsets, cpix, cdt, key_offsets, n, flow, H, W, tile_h, tile_w, halo, pad_h, pad_w,
                              z_image, nullptr, g_lo, d_flow, nullptr, nullptr, upstream, addend, workspace, workspace_bytes, part_table,
                              stream, mj);
}

int ebos_variance_dense_job_f32(const ebos_dense_job* job, const float* flow, float* out_variance, const float* upstream,
                                float* d_flow, ebos_stream_t stream) {
  return ebos_variance_dense_job_signed_f32(job, flow, out_variance, nullptr, upstream, d_flow, stream);
}

int ebos_variance_dense_job_signed_f32(const ebos_dense_job* job, const float* flow, float* out_variance, float* out_scaled,
                                       const float* upstream, float* d_flow, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(job && flow && out_variance, "ebos_variance_dense_job: NULL job / flow / out_variance");
  EBOS_REQUIRE(out_scaled == nullptr || d_flow != nullptr, "ebos_variance_dense_job: out_scaled comes with the gradient (d_flow)");
  EBOS_REQUIRE(job->iwe && job->moments && job->workspace, "ebos_variance_dense_job: the job needs iwe, moments and a workspace");
  // value + gradient: the forward call leaves the (sum, sum of squares) partials (want_variance = 2) and the backward kernel reduces
  // them itself -- no finalize launch in between; value only: the forward call finalizes
  int rc = ebos_iwe_dense_slab_f32(job->xs, job->ys, job->dts, nullptr, job->grp_offsets, job->cpix, job->cdt, job->key_offsets, job->n,
                                   flow, job->H, job->W, job->tile_h, job->tile_w, job->halo, job->splits, job->pad_h, job->pad_w,
                                   job->workspace, job->workspace_bytes, job->iwe, d_flow ? 2 : 1, job->omit_boundary, out_variance,
                                   job->moments, job->part_table, stream);
  if (rc != EBOS_OK || d_flow == nullptr) return rc;
  size_t poff = 0;
  int64_t nparts = 0, npix = 0;
  rc = ebos_iwe_slab_partials(job->H, job->W, job->tile_h, job->tile_w, job->halo, job->splits, job->pad_h, job->pad_w, job->omit_boundary,
                              &poff, &nparts, &npix);
  if (rc != EBOS_OK) return rc;
  EBOS_REQUIRE(npix >= 2, "ebos_variance_dense_job: the variance needs at least two pixels");
  const MomentsIn mj{reinterpret_cast<const double*>(reinterpret_cast<const char*>(job->workspace) + poff), nparts, npix, out_variance,
                     job->moments, out_scaled, 0};
  // (upstream == nullptr: the backward kernel takes 1.0f -- no device-resident constant, nothing per-device to cache)
  const bool adaptive = job->splits == 0 && job->part_table != nullptr;
  return dense_tiled_bwd_impl(job->xs, job->ys, job->dts, nullptr, job->grp_offsets, job->cpix, job->cdt, job->key_offsets, job->n, flow,
                              job->H, job->W, job->tile_h, job->tile_w, job->halo, job->pad_h, job->pad_w, job->iwe, nullptr,
                              job->omit_boundary ? 1 : 0, d_flow, nullptr, nullptr, upstream, nullptr, adaptive ? job->workspace : nullptr,
                              adaptive ? job->workspace_bytes : 0, adaptive ? job->part_table : nullptr, stream, mj);
}

int ebos_gradient_magnitude_dense_job_f32(const ebos_dense_job* job, const float* flow, float* out_contrast, float* out_scaled,
                                          const float* upstream, float* d_flow, float* d_iwe, double* partials, int64_t n_partials,
                                          ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(job && flow && out_contrast, "ebos_gradient_magnitude_dense_job: NULL job / flow / out_contrast");
  EBOS_REQUIRE(job->iwe && job->workspace && d_iwe && partials, "ebos_gradient_magnitude_dense_job: the job needs iwe, a workspace, d_iwe and partials");
  EBOS_REQUIRE(out_scaled == nullptr || d_flow != nullptr, "ebos_gradient_magnitude_dense_job: out_scaled comes with the gradient (d_flow)");
  const int h = job->H + 2 * job->pad_h, w = job->W + 2 * job->pad_w;
  // accumulate + combine (no variance), then ONE pass over the image for the Sobel value partials and the gradient image
  int rc = ebos_iwe_dense_slab_f32(job->xs, job->ys, job->dts, nullptr, job->grp_offsets, job->cpix, job->cdt, job->key_offsets, job->n,
                                   flow, job->H, job->W, job->tile_h, job->tile_w, job->halo, job->splits, job->pad_h, job->pad_w,
                                   job->workspace, job->workspace_bytes, job->iwe, 0, job->omit_boundary, nullptr, nullptr,
                                   job->part_table, stream);
  if (rc != EBOS_OK) return rc;
  // value only: the Sobel pass finalizes; value + gradient: the backward kernel's first workgroup sums the value partials
  // (a value-only call needs no gradient image: the Sobel pass then skips the gather of the nine stencils and 4 H W bytes of stores)
  rc = ebos_gradient_magnitude_fused_f32(job->iwe, h, w, job->omit_boundary, upstream, d_flow ? nullptr : out_contrast,
                                         d_flow ? d_iwe : nullptr, partials, n_partials, stream);
  if (rc != EBOS_OK || d_flow == nullptr) return rc;
  const int lo = job->omit_boundary ? 1 : 0;
  const int64_t npix = (int64_t)(h - 2 * lo > 0 ? h - 2 * lo : 0) * (w - 2 * lo > 0 ? w - 2 * lo : 0);
  EBOS_REQUIRE(npix >= 1, "ebos_gradient_magnitude_dense_job: empty image region");
  const MomentsIn mj{partials, ebos_gradient_magnitude_fused_partials(h, w), npix, out_contrast, nullptr, out_scaled, 1};
  const bool adaptive = job->splits == 0 && job->part_table != nullptr;
  return dense_tiled_bwd_impl(job->xs, job->ys, job->dts, nullptr, job->grp_offsets, job->cpix, job->cdt, job->key_offsets, job->n, flow,
                              job->H, job->W, job->tile_h, job->tile_w, job->halo, job->pad_h, job->pad_w, d_iwe, nullptr, 0, d_flow,
                              nullptr, nullptr, upstream /* (only scales out_scaled here: the Sobel pass applied it to d_iwe) */, nullptr,
                              adaptive ? job->workspace : nullptr,
                              adaptive ? job->workspace_bytes : 0, adaptive ? job->part_table : nullptr, stream, mj);
}

size_t ebos_patch_grad_partials_bytes(int H, int W, int tile_h, int tile_w, int adaptive) {
  using namespace ebos;
  if (H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0) return 0;
  const size_t items = (size_t)((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w) * (adaptive ? kAdaptiveItemsPerTile : 1);
  return items * 2 * kGridCells * kGridCells * sizeof(float);
}

// ---- the 2-DoF Adam loop natively (motion_model "2d-translation", configs/hot_plate1.yaml:47,65,70: Adam, 600 iterations, blur_sigma 3;
// the loop of src/solver/generative_max_likelihood.py:306-341 on loss = -w var([blur3] IWE(theta))) ---------------------------------
// One C call enqueues n_iter iterations of: accumulate (UNIFORM) + combine [+ the blur's image pass] -> backward (UNIFORM; every
// workgroup reduces the variance partials itself, workgroup 0 reports the variance) -> partial pairs summed, loss recorded, Adam
// step.  Four launches per iteration (five with the blur), nothing read back: the Python loop it replaces synchronised once per
// iteration (float(loss)) around ~35 launches.
static int cmax_2dof_check(const ebos_cmax_2dof_problem* q) {
  using namespace ebos;
  EBOS_REQUIRE(q != nullptr && q->steps_done >= 0, "ebos_cmax_2dof_solve: NULL problem or negative steps_done");
  EBOS_REQUIRE(q->key_offsets && ((q->grp_offsets && q->cpix && q->cdt) || (q->xs && q->ys && q->dts) || q->n == 0),
               "ebos_cmax_2dof_solve: NULL plan buffers");
  EBOS_REQUIRE((q->cfx == nullptr) == (q->cfy == nullptr), "ebos_cmax_2dof_solve: cfx and cfy come together or not at all");
  EBOS_REQUIRE(q->theta && q->d_theta && q->exp_avg && q->exp_avg_sq && q->step && q->iwe && q->variance && q->moments && q->upstream &&
                   q->workspace,
               "ebos_cmax_2dof_solve: NULL buffer");
  EBOS_REQUIRE(q->n >= 0 && q->H > 0 && q->W > 0 && q->pad_h >= 0 && q->pad_w >= 0 && q->splits >= 0 && q->splits <= 64,
               "ebos_cmax_2dof_solve: bad sizes");
  EBOS_REQUIRE(q->splits != 0 || q->part_table, "ebos_cmax_2dof_solve: splits = 0 (adaptive work items) needs the plan's part_table");
  EBOS_REQUIRE(q->w_variance != 0.0f, "ebos_cmax_2dof_solve: w_variance is 0 (the device word `upstream` holds -w_variance)");
  EBOS_REQUIRE(q->blur_k0 == 0.0f || (q->blur_k0 > 0.0f && q->blur_k1 > 0.0f && q->blur_image && q->cost_scratch),
               "ebos_cmax_2dof_solve: the blurred contrast needs positive taps, blur_image and cost_scratch");
  EBOS_REQUIRE(q->lr >= 0.0 && q->beta1 >= 0.0 && q->beta1 < 1.0 && q->beta2 >= 0.0 && q->beta2 < 1.0 && q->eps >= 0.0,
               "ebos_cmax_2dof_solve: bad hyper-parameters");
  return EBOS_OK;
}

int ebos_cmax_2dof_solve_f32(const ebos_cmax_2dof_problem* q, int n_iter, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(n_iter >= 0, "ebos_cmax_2dof_solve: negative n_iter");
  if (int rc = cmax_2dof_check(q)) return rc;
  const HaloArg ha = decode_halo(q->halo);
  const int halo = ha.halo, tile_h = q->tile_h, tile_w = q->tile_w;
  if (!slab_config_ok(tile_h, tile_w, halo)) {
    set_error("ebos_cmax_2dof_solve: no kernel built for tile %dx%d halo %d (see ebos_slab_config)", tile_h, tile_w, halo);
    return EBOS_ERR_UNSUPPORTED;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(q->H, q->W, tile_h, tile_w, halo, q->splits, q->pad_h, q->pad_w);
  if (q->workspace_bytes < need) {
    set_error("ebos_cmax_2dof_solve: workspace too small (%zu < %zu)", q->workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  const int h = q->H + 2 * q->pad_h, w = q->W + 2 * q->pad_w, lo = q->omit_boundary ? 1 : 0;
  const bool blur = q->blur_k0 != 0.0f;
  const int64_t n_blur = blur ? ebos_blur3_variance_partials(h, w) : 0;
  if (blur && q->cost_scratch_bytes < (size_t)n_blur * 16) {
    set_error("ebos_cmax_2dof_solve: cost_scratch too small (%zu < %zu)", q->cost_scratch_bytes, (size_t)n_blur * 16);
    return EBOS_ERR_SCRATCH;
  }
  size_t off = 0;
  int64_t n_parts = 0, n_px = 0;
  if (int rc = ebos_iwe_slab_partials(q->H, q->W, tile_h, tile_w, q->halo, q->splits, q->pad_h, q->pad_w, q->omit_boundary, &off, &n_parts,
                                      &n_px))
    return rc;
  EBOS_REQUIRE(n_px >= 2, "ebos_cmax_2dof_solve: the variance needs at least two valid pixels");
  hipStream_t s = as_stream(stream);
  char* ws = reinterpret_cast<char*>(q->workspace);
  const int n_tiles_ = ((q->H + tile_h - 1) / tile_h) * ((q->W + tile_w - 1) / tile_w);
  const int32_t* pt = q->part_table;
  // (cfx / cfy: the compact slots carry the fractions of undistorted events -- the general loops on the compact layout, the resident
  // FRAC kernels' arithmetic)
  const EvPtrs evf{q->xs, q->ys, q->dts, nullptr, q->grp_offsets, q->cpix, q->cdt, pt, pt ? pt + n_tiles_ + 1 : nullptr,
                   pt ? pt + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr, q->cfx, q->cfy};
  const EvPtrs evb{q->xs, q->ys, q->dts, nullptr, q->grp_offsets, q->cpix, q->cdt, nullptr, nullptr, nullptr, q->cfx, q->cfy};
  // the backward kernel's tile partials live in the slab section of the workspace (dead once the image is combined)
  double* tile_partials = reinterpret_cast<double*>(q->workspace);
  const double* var_partials = blur ? reinterpret_cast<const double*>(q->cost_scratch) : reinterpret_cast<const double*>(ws + off);
  const MomentsIn mj{var_partials, blur ? n_blur : n_parts, n_px, q->variance, q->moments, nullptr, 0, Blur3{q->blur_k0, q->blur_k1}};
  const float* g_image = blur ? q->blur_image : q->iwe;
  for (int it = 0; it < n_iter; ++it) {
    const int t = q->steps_done + it + 1;
    int rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->fwd(evf, q->key_offsets, q->theta, true, q->H, q->W, q->splits, q->pad_h, q->pad_w, ws, q->iwe,
                                             blur ? 0 : 2, q->omit_boundary, nullptr, nullptr, (int)ACC_FX, s, nullptr, ha);
    if (rc != EBOS_OK) return rc;
    if (blur) {
      rc = ebos_blur3_variance_adjoint_f32(q->iwe, h, w, q->omit_boundary, q->blur_k0, q->blur_k1, q->blur_image,
                                           reinterpret_cast<double*>(q->cost_scratch), n_blur, stream);
      if (rc != EBOS_OK) return rc;
    }
    rc = EBOS_ERR_UNSUPPORTED;
    rc = slab_ops(tile_h, tile_w, halo)->bwd(evb, q->key_offsets, q->theta, true, q->H, q->W, q->pad_h, q->pad_w, g_image, nullptr, lo,
                                             q->d_theta, nullptr, tile_partials, nullptr, q->upstream, nullptr, nullptr, s, nullptr, 0, 0.0f,
                                             0.0f, nullptr, mj, ha, false);
    if (rc != EBOS_OK) return rc;
    theta_adam_kernel<<<dim3(1), dim3(256), 0, s>>>(tile_partials, n_tiles_, q->d_theta, q->theta, q->exp_avg, q->exp_avg_sq, q->lr,
                                                    q->beta1, q->beta2, q->eps, t, q->step, q->variance, q->upstream, q->losses,
                                                    q->losses_cap);
  }
  EBOS_CHECK_LAUNCH("ebos_cmax_2dof_solve");
  return EBOS_OK;
}

static int patch_tiled_bwd_impl(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets,
                               int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w,
                               int H, int W, int tile_h, int tile_w, int halo_arg, int pad_h, int pad_w, const float* g_image,
                               const float* affine, int g_lo, const double* var_moments, const float* upstream,
                               const float* addend, float* grad_partials, size_t grad_partials_bytes, const int32_t* part_table,
                               float w_flow_norm, float w_image_gradient, double* reg_partials, const double* var_partials,
                               int64_t n_var_partials, int64_t n_var_pixels, float* out_variance, double* out_moments,
                               ebos::Blur3 blur, ebos_stream_t stream, const float* cfx = nullptr, const float* cfy = nullptr) {
  using namespace ebos;
  EBOS_REQUIRE((cfx == nullptr) == (cfy == nullptr), "ebos_iwe_patch_tiled_bwd_frac: cfx / cfy come together");
  const HaloArg ha = decode_halo(halo_arg);
  const int halo = ha.halo;
  EBOS_REQUIRE(var_partials == nullptr || (var_moments == nullptr && upstream != nullptr && n_var_partials >= 1 && n_var_pixels >= 2),
               "ebos_iwe_patch_tiled_bwd: var_partials needs upstream, no var_moments, and sane counts");
  EBOS_REQUIRE(blur.k0 == 0.0f || (var_partials != nullptr && affine == nullptr && blur.k0 > 0.0f && blur.k1 > 0.0f),
               "ebos_iwe_patch_tiled_bwd_blur: needs the partials of ebos_blur3_variance_adjoint_f32, no affine map, positive taps");
  const MomentsIn mj{var_partials, n_var_partials, n_var_pixels, out_variance, out_moments, nullptr, 0, blur};
  EBOS_REQUIRE((w_flow_norm == 0.0f && w_image_gradient == 0.0f) || reg_partials,
               "ebos_iwe_patch_tiled_bwd: regulariser weight given but reg_partials is NULL");
  EBOS_REQUIRE(w_image_gradient == 0.0f || (H >= 2 && W >= 2),
               "ebos_iwe_patch_tiled_bwd: image_gradient needs at least 2 samples per axis (torch.gradient)");
  const bool any_reg = w_flow_norm != 0.0f || w_image_gradient != 0.0f;
  EBOS_REQUIRE(grid && g_image && grad_partials && key_offsets && grp_offsets && cpix && cdt,
               "ebos_iwe_patch_tiled_bwd: NULL grid/g_image/grad_partials/plan buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && g_lo >= 0 && gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 &&
                   slide_h > 0 && slide_w > 0,
               "ebos_iwe_patch_tiled_bwd: bad sizes");
  EBOS_REQUIRE(var_partials != nullptr || (var_moments == nullptr) == (upstream == nullptr),
               "ebos_iwe_patch_tiled_bwd: var_moments and upstream go together");
  if (!slab_config_ok(tile_h, tile_w, halo) || !ebos_patch_fused_supported(tile_h, tile_w, halo, slide_h, slide_w)) {
    set_error("ebos_iwe_patch_tiled_bwd: tile %dx%d halo %d with sliding window %dx%d is outside ebos_patch_fused_supported", tile_h,
              tile_w, halo, slide_h, slide_w);
    return EBOS_ERR_UNSUPPORTED;
  }
  const int adaptive = part_table != nullptr;
  const size_t need = ebos_patch_grad_partials_bytes(H, W, tile_h, tile_w, adaptive);
  if (grad_partials_bytes < need) {
    set_error("ebos_iwe_patch_tiled_bwd: grad_partials too small (%zu < %zu)", grad_partials_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  const int n_tiles_ = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  const EvPtrs evp{nullptr, nullptr, nullptr, nullptr, grp_offsets, cpix, cdt, part_table, part_table ? part_table + n_tiles_ + 1 : nullptr,
                   part_table ? part_table + n_tiles_ + 1 + kAdaptiveItemsPerTile * n_tiles_ : nullptr, cfx, cfy};
  const GridSrc gs{make_axis(gh, patch_h, slide_h, H), make_axis(gw, patch_w, slide_w, W)};
  int rc = EBOS_ERR_UNSUPPORTED;
  rc = slab_ops(tile_h, tile_w, halo)->bwd(evp, key_offsets, grid, false, H, W, pad_h, pad_w, g_image, affine, g_lo, nullptr, nullptr, nullptr,
                                           var_moments, upstream, addend, grad_partials, s, &gs, adaptive,
                                           w_flow_norm / (float)((int64_t)H * W), w_image_gradient / (float)(2 * (int64_t)H * W),
                                           any_reg ? reg_partials : nullptr, mj, ha, true);
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_patch_tiled_bwd");
  return EBOS_OK;
}

int ebos_iwe_patch_tiled_bwd_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets,
                                 int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w,
                                 int H, int W, int tile_h, int tile_w, int halo_arg, int pad_h, int pad_w, const float* g_image,
                                 const float* affine, int g_lo, const double* var_moments, const float* upstream,
                                 const float* addend, float* grad_partials, size_t grad_partials_bytes, const int32_t* part_table,
                                 float w_flow_norm, float w_image_gradient, double* reg_partials, const double* var_partials,
                                 int64_t n_var_partials,
                                 int64_t n_var_pixels, float* out_variance, double* out_moments, ebos_stream_t stream) {
  return patch_tiled_bwd_impl(grp_offsets, cpix, cdt, key_offsets, n, grid, gh, gw, patch_h, patch_w, slide_h, slide_w, H, W, tile_h,
                              tile_w, halo_arg, pad_h, pad_w, g_image, affine, g_lo, var_moments, upstream, addend, grad_partials,
                              grad_partials_bytes, part_table, w_flow_norm, w_image_gradient, reg_partials, var_partials,
                              n_var_partials, n_var_pixels, out_variance, out_moments, ebos::Blur3{0.0f, 0.0f}, stream);
}

int ebos_iwe_patch_tiled_bwd_blur_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets,
                                      int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w,
                                      int H, int W, int tile_h, int tile_w, int halo_arg, int pad_h, int pad_w, const float* z_image,
                                      int g_lo, const float* upstream, float* grad_partials, size_t grad_partials_bytes,
                                      const int32_t* part_table, float w_flow_norm, float w_image_gradient, double* reg_partials,
                                      const double* blur_partials, int64_t n_blur_partials, int64_t n_var_pixels, float* out_variance,
                                      double* out_moments, float blur_k0, float blur_k1, ebos_stream_t stream) {
  return patch_tiled_bwd_impl(grp_offsets, cpix, cdt, key_offsets, n, grid, gh, gw, patch_h, patch_w, slide_h, slide_w, H, W, tile_h,
                              tile_w, halo_arg, pad_h, pad_w, z_image, nullptr, g_lo, nullptr, upstream, nullptr, grad_partials,
                              grad_partials_bytes, part_table, w_flow_norm, w_image_gradient, reg_partials, blur_partials,
                              n_blur_partials, n_var_pixels, out_variance, out_moments, ebos::Blur3{blur_k0, blur_k1}, stream);
}

/* the grid-sampling backward pass on a compact plan that carries the fractions of undistorted events (ebos_plan_compact_frac_f32);
 * blur_k0 == 0: the plain contrast (arguments as ebos_iwe_patch_tiled_bwd_f32 with var_partials), else the blurred one (as _blur_f32) */
int ebos_iwe_patch_tiled_bwd_frac_f32(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const float* cfx, const float* cfy,
                                      const int32_t* key_offsets, int64_t n, const float* grid, int gh, int gw, int patch_h, int patch_w,
                                      int slide_h, int slide_w, int H, int W, int tile_h, int tile_w, int halo_arg, int pad_h, int pad_w,
                                      const float* g_image, int g_lo, const double* var_moments, const float* upstream,
                                      const float* addend, float* grad_partials, size_t grad_partials_bytes, const int32_t* part_table,
                                      float w_flow_norm, float w_image_gradient, double* reg_partials, const double* var_partials,
                                      int64_t n_var_partials, int64_t n_var_pixels, float* out_variance, double* out_moments,
                                      float blur_k0, float blur_k1, ebos_stream_t stream) {
  return patch_tiled_bwd_impl(grp_offsets, cpix, cdt, key_offsets, n, grid, gh, gw, patch_h, patch_w, slide_h, slide_w, H, W, tile_h,
                              tile_w, halo_arg, pad_h, pad_w, g_image, nullptr, g_lo, var_moments, upstream, addend, grad_partials,
                              grad_partials_bytes, part_table, w_flow_norm, w_image_gradient, reg_partials, var_partials,
                              n_var_partials, n_var_pixels, out_variance, out_moments, ebos::Blur3{blur_k0, blur_k1}, stream, cfx, cfy);
}

}  // extern "C"
