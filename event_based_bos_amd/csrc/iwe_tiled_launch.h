// iwe_tiled_launch.h -- the launchers of the tile-private pipeline (accumulate -> combine [-> finalize], batched forward, backward) as
// templates over the tile configuration, and the table of them the C entry points of iwe_tiled.hip dispatch through.  One translation
// unit per configuration (iwe_tiled_<TH>x<TW>x<HALO>.hip) instantiates them: as one unit the eight configurations took three minutes
// to compile, the long pole of the build.
#pragma once
#include "iwe_tile_core.h"

namespace ebos {

// (shared host state: defined once, in iwe_tiled.hip)
unsigned next_spill_epoch();                                                   // a number of its own for every forward call (never 0)
int order_after(hipStream_t later, hipStream_t earlier, const char* what);    // `later` waits for everything enqueued on `earlier`

struct SlabOps {  // the launchers of one built configuration
  int th, tw, halo;
  int (*fwd)(const EvPtrs& ev, const int32_t* key_offsets, const float* flow, bool uniform, int H, int W, int splits, int pad_h, int pad_w,
             char* ws, float* iwe, int want_var, int omit, float* out_var, double* moments, int acc_mode, hipStream_t s,
             const GridSrc* grid_src, const HaloArg& ha);
  int (*fwd_batch)(const FwdBatch& b, int n, int H, int W, int splits, int pad_h, int pad_w, int want_var, int omit, hipStream_t s,
                   hipStream_t s_tail, const GridSrc* grid_src, const HaloArg& ha, bool uniform);
  int (*bwd)(const EvPtrs& ev, const int32_t* key_offsets, const float* flow, bool uniform, int H, int W, int pad_h, int pad_w,
             const float* g_image, const float* affine, int g_lo, float* d_flow, float* d_weight, double* partials,
             const double* var_moments, const float* upstream, const float* addend, float* part_out, hipStream_t s,
             const GridSrc* grid_src, int adaptive, float s_norm, float s_tv, double* reg_partials, MomentsIn mj, const HaloArg& ha,
             bool finalize_uniform);
};
// the built configurations' tables (kSlabConfigs, iwe_tile_core.h); nullptr: not built
const SlabOps* slab_ops(int th, int tw, int halo);

#ifdef EBOS_SLAB_OPS_UNIT  // a configuration's translation unit: the templates and EBOS_DEFINE_SLAB_OPS
namespace {

template <int TH, int TW, int HALO>
int launch_slab_fwd(const EvPtrs& ev, const int32_t* key_offsets, const float* flow, bool uniform, int H, int W, int splits,
                    int pad_h, int pad_w, char* ws, float* iwe, int want_var, int omit, float* out_var, double* moments,
                    int acc_mode, hipStream_t s, const GridSrc* grid_src = nullptr, const HaloArg& ha = HaloArg{HALO, false, 0.0f}) {
  size_t lds = (size_t)acc_cells<TH, TW, HALO, false>() * sizeof(double);  // + dummy region
  static_assert((size_t)acc_cells<TH, TW, HALO, true>() * sizeof(double) + 1024 <= 160 * 1024,
                "f64 tile + halo (at the run-time windows' pitch) must fit the 160 KiB LDS of a CDNA4 CU");
  const SlabLayout L = slab_layout(H, W, TH, TW, HALO, splits, pad_h, pad_w);
  if (L.off_spill >= ((size_t)1 << 32)) {  // the combine pass addresses the slab section with 32-bit byte offsets (sc1 buffer loads)
    set_error("ebos_iwe_*_slab: %zu bytes of slabs (image %dx%d, %d work items): the slab section must stay below 4 GiB", L.off_spill, H,
              W, L.nblk);
    return EBOS_ERR_UNSUPPORTED;
  }
  float* slabs = reinterpret_cast<float*>(ws);
  float* spill = reinterpret_cast<float*>(ws + L.off_spill);
  double* partials = reinterpret_cast<double*>(ws + L.off_partials);
  unsigned* spill_epoch = reinterpret_cast<unsigned*>(ws + L.off_epoch);
  unsigned* halo_tab = reinterpret_cast<unsigned*>(ws + L.off_halo);
  const unsigned epoch = next_spill_epoch();
  // unit weights -> verified fixed point (2 ds_add_u64 per event); per-event weights -> fixed point in units of the slice's max |w|,
  // exact f64 redo where a field wraps (overflow, negative weights)
  void (*ka)(EvPtrs, const int32_t*, const float*, int, int, int, int, int, int, float*, float*, GridSrc, unsigned*, unsigned, float,
             unsigned*);
  // (a compact plan holds integer source pixels; per-event weights ride along in plan order: load_weights4 -- 10 B / event instead
  // of the (x, y, dt) format's 16)
  const bool compact = ev.cpix != nullptr;
  // run-time windows: the lean loop only (compact plan, unit weights, fixed point); anything else runs the largest window
  const bool dyn = ha.dyn && compact && ev.w == nullptr && acc_mode == ACC_FX;
  if (dyn) lds = (size_t)acc_cells<TH, TW, HALO, true>() * sizeof(double);
  GridSrc gs{};
#define EBOS_PICK(HW, MD)                                                                                              \
  (uniform ? (compact ? iwe_slab_accumulate_kernel<TH, TW, HALO, HW, MD, FMT_COMPACT, true>                            \
                      : iwe_slab_accumulate_kernel<TH, TW, HALO, HW, MD, FMT_XY, true>)                                 \
           : (compact ? iwe_slab_accumulate_kernel<TH, TW, HALO, HW, MD, FMT_COMPACT, false>                           \
                      : iwe_slab_accumulate_kernel<TH, TW, HALO, HW, MD, FMT_XY, false>))
  // (per-event weights: fixed point too, in units of the slice's max |w| -- TileShared::wscale; EBOS_SLAB_ACC=f64 forces doubles)
  if (ev.w) ka = acc_mode == ACC_F64 ? EBOS_PICK(true, ACC_F64) : EBOS_PICK(true, ACC_FX);
  else if (acc_mode == ACC_F64) ka = EBOS_PICK(false, ACC_F64);
  else ka = EBOS_PICK(false, ACC_FX);
#undef EBOS_PICK
  if (dyn)
    ka = uniform ? iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, true, false, true>
                 : iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, false, true>;
  if (compact && ev.cfx != nullptr && grid_src == nullptr) {  // the compact slots carry the fractions of undistorted events: the general loop on them (the patch grid: below)
    if (!uniform || ev.w != nullptr || acc_mode != ACC_FX) {
      set_error("ebos_iwe_*_slab: fractions per compact slot (cfx / cfy) go with the 2-DoF model or the patch grid, unit weights");
      return EBOS_ERR_UNSUPPORTED;
    }
    ka = dyn ? iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, true, false, true, true>
             : iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, true, false, false, true>;
  }
  if (grid_src != nullptr) {  // `flow` is a patch grid, sampled per tile inside the kernel (compact unit-weight plans)
    if constexpr (grid_fwd_fits<TH, TW, HALO>()) {
      if (!compact || uniform) {
        set_error("ebos_iwe_patch_slab: needs the compact plan format and unit weights");
        return EBOS_ERR_UNSUPPORTED;
      }
      ka = dyn ? iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, true>
               : iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, false>;
      if (ev.cfx != nullptr)  // the compact slots carry the fractions of undistorted events: the general loop on them
        ka = dyn ? iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, true, true>
                 : iwe_slab_accumulate_kernel<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, false, true>;
      lds += grid_lds_extra<TH, TW, 0>();
      gs = *grid_src;
    } else {
      set_error("ebos_iwe_patch_slab: tile %dx%d halo %d leaves no LDS for the tile's flow (ebos_patch_fused_supported)", TH, TW, HALO);
      return EBOS_ERR_UNSUPPORTED;
    }
  }
  if (int rc = reserve_lds(ka, lds, "ebos_iwe_dense_slab")) return rc;
  hipEvent_t t0, t1;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_SLAB_ACCUMULATE))  // bench.py's roofline leg: events stamped with this dispatch's begin / end
    hipExtLaunchKernelGGL(ka, dim3((unsigned)L.nblk), dim3(kBlock), lds, s, t0, t1, 0, ev, key_offsets, flow, H, W, L.tiles_x,
                          splits, pad_h, pad_w, slabs, spill, gs, spill_epoch, epoch, ha.dt_bound, halo_tab);
  else
    ka<<<dim3((unsigned)L.nblk), dim3(kBlock), lds, s>>>(ev, key_offsets, flow, H, W, L.tiles_x, splits, pad_h, pad_w, slabs, spill, gs,
                                                         spill_epoch, epoch, ha.dt_bound, halo_tab);
  int64_t nparts;
  // the solver's patch-grid route sums the image exactly (kCombineExactSum): the resident form of its loop must find the same mean
  const int g_lo = (omit ? 1 : 0) | (grid_src != nullptr ? kCombineExactSum : 0);
  // want_var == 1: the combine pass's last workgroup reduces the partials itself (FinalizeIn: no finalize launch); its counters
  // have a section of the workspace (zero between calls)
  const int lo_ = omit ? 1 : 0;
  const long long m_valid = (long long)(L.h - 2 * lo_ > 0 ? L.h - 2 * lo_ : 0) * (L.w - 2 * lo_ > 0 ? L.w - 2 * lo_ : 0);
  const FinalizeIn fin{want_var == 1 ? reinterpret_cast<unsigned*>(ws + L.off_counters) : nullptr, out_var, moments, m_valid};
  if (L.w % 4 == 0 && pad_w % 4 == 0) {
    dim3 gb((L.w / 4 + 63) / 64, (L.h + kCombineRows - 1) / kCombineRows);
    nparts = (int64_t)gb.x * gb.y;
    auto kc = dyn ? iwe_slab_combine4_kernel<TH, TW, HALO, true> : iwe_slab_combine4_kernel<TH, TW, HALO, false>;
    if (profile_next_pair(&t0, &t1, EBOS_PROFILE_SLAB_COMBINE))
      hipExtLaunchKernelGGL(kc, gb, dim3(kCombineBlock), 0, s, t0, t1, 0, slabs, spill, L.tiles_y,
                            L.tiles_x, splits, H, W, pad_h, pad_w, iwe, g_lo, want_var ? partials : nullptr,
                            splits == 0 ? ev.part_off : nullptr, spill_epoch, epoch, halo_tab, fin);
    else
      kc<<<gb, dim3(kCombineBlock), 0, s>>>(slabs, spill, L.tiles_y, L.tiles_x, splits, H, W, pad_h, pad_w, iwe, g_lo,
                                            want_var ? partials : nullptr, splits == 0 ? ev.part_off : nullptr, spill_epoch, epoch,
                                            halo_tab, fin);
  } else {
    dim3 gb((L.w + kCombineBlock - 1) / kCombineBlock, L.h);
    nparts = (int64_t)gb.x * gb.y;
    iwe_slab_combine_kernel<TH, TW, HALO><<<gb, dim3(kCombineBlock), 0, s>>>(slabs, spill, L.tiles_y, L.tiles_x, splits, H, W,
                                                                            pad_h, pad_w, iwe, g_lo,
                                                                            want_var ? partials : nullptr,
                                                                            splits == 0 ? ev.part_off : nullptr, spill_epoch, epoch,
                                                                            dyn ? halo_tab : nullptr, fin);
  }
  (void)nparts;  // (want_var == 2: the caller reduces the partials itself, ebos_iwe_slab_partials; 1: the combine pass's last workgroup did)
  return EBOS_OK;
}

// n <= kMaxBatch windows of one geometry: the accumulate pass as ONE persistent launch (workgroup b = work item b of every window
// in turn), combine and finalize each as one launch over (pixel block | 1, window)
template <int TH, int TW, int HALO>
int launch_slab_fwd_batch(const FwdBatch& b, int n, int H, int W, int splits, int pad_h, int pad_w, int want_var, int omit,
                          hipStream_t s, hipStream_t s_tail, const GridSrc* grid_src, const HaloArg& ha, bool uniform = false) {
  size_t lds = (size_t)(ha.dyn ? acc_cells<TH, TW, HALO, true>() : acc_cells<TH, TW, HALO, false>()) * sizeof(double);
  const SlabLayout L = slab_layout(H, W, TH, TW, HALO, splits, pad_h, pad_w);
  if (L.off_spill >= ((size_t)1 << 32)) {  // the combine pass addresses the slab section with 32-bit byte offsets (sc1 buffer loads)
    set_error("ebos_iwe_*_slab: %zu bytes of slabs (image %dx%d, %d work items): the slab section must stay below 4 GiB", L.off_spill, H,
              W, L.nblk);
    return EBOS_ERR_UNSUPPORTED;
  }
  if (!(L.w % 4 == 0 && pad_w % 4 == 0)) {
    set_error("ebos_iwe_slab_batch: needs image and padding widths that are multiples of 4 (call the single-window entry)");
    return EBOS_ERR_UNSUPPORTED;
  }
  const unsigned epoch = next_spill_epoch();
  GridSrc gs{};
  hipEvent_t t0, t1;
  void (*ka)(FwdBatch, int, int, int, int, int, int, int, GridSrc, unsigned, float);
  if (grid_src != nullptr) {
    if constexpr (grid_fwd_fits<TH, TW, HALO>()) {
      ka = ha.dyn ? iwe_slab_accumulate_batch_kernel<TH, TW, HALO, true, true> : iwe_slab_accumulate_batch_kernel<TH, TW, HALO, true, false>;
      lds += grid_lds_extra<TH, TW, 0>();
      gs = *grid_src;
    } else {
      set_error("ebos_iwe_slab_batch: tile %dx%d halo %d leaves no LDS for the tile's flow (ebos_patch_fused_supported)", TH, TW, HALO);
      return EBOS_ERR_UNSUPPORTED;
    }
  } else if (uniform) {
    ka = ha.dyn ? iwe_slab_accumulate_batch_kernel<TH, TW, HALO, false, true, true> : iwe_slab_accumulate_batch_kernel<TH, TW, HALO, false, false, true>;
  } else {
    ka = ha.dyn ? iwe_slab_accumulate_batch_kernel<TH, TW, HALO, false, true> : iwe_slab_accumulate_batch_kernel<TH, TW, HALO, false, false>;
  }
  if (int rc = reserve_lds(ka, lds, "ebos_iwe_slab_batch")) return rc;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_SLAB_ACCUMULATE))
    hipExtLaunchKernelGGL(ka, dim3((unsigned)L.nblk), dim3(kBlock), lds, s, t0, t1, 0, b, n, H, W, L.tiles_x, splits, pad_h, pad_w, gs,
                          epoch, ha.dt_bound);
  else
    ka<<<dim3((unsigned)L.nblk), dim3(kBlock), lds, s>>>(b, n, H, W, L.tiles_x, splits, pad_h, pad_w, gs, epoch, ha.dt_bound);
  // the combine + finalize passes of this batch go to s_tail (when the caller gave one): they need no LDS and run beside the
  // accumulate pass of the NEXT batch, which the caller enqueues on s right behind this one
  if (s_tail != s)
    if (int rc = order_after(s_tail, s, "ebos_iwe_slab_batch")) return rc;
  const dim3 gb((L.w / 4 + 63) / 64, (L.h + kCombineRows - 1) / kCombineRows, (unsigned)n);
  auto kc = ha.dyn ? iwe_slab_combine4_batch_kernel<TH, TW, HALO, true> : iwe_slab_combine4_batch_kernel<TH, TW, HALO, false>;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_SLAB_COMBINE))
    hipExtLaunchKernelGGL(kc, gb, dim3(kCombineBlock), 0, s_tail, t0, t1, 0, b, L.tiles_y, L.tiles_x, splits, H, W, pad_h, pad_w,
                          omit ? 1 : 0, want_var, epoch);
  else
    kc<<<gb, dim3(kCombineBlock), 0, s_tail>>>(b, L.tiles_y, L.tiles_x, splits, H, W, pad_h, pad_w, omit ? 1 : 0, want_var, epoch);
  if (want_var == 1) {
    const int lo = omit ? 1 : 0;
    const int64_t m = (int64_t)(L.h - 2 * lo > 0 ? L.h - 2 * lo : 0) * (L.w - 2 * lo > 0 ? L.w - 2 * lo : 0);
    moments_finalize_batch_kernel<<<dim3((unsigned)n), dim3(256), 0, s_tail>>>(b, (int64_t)gb.x * gb.y, m);
  }
  return EBOS_OK;
}

// grid_src != nullptr: `flow` is the patch grid and `part_out` receives the per-item partial cell gradients
// ([items][2][kGridCells][kGridCells]); `adaptive` then selects the plan's work items
template <int TH, int TW, int HALO>
int launch_tiled_bwd(const EvPtrs& ev, const int32_t* key_offsets, const float* flow, bool uniform, int H, int W, int pad_h,
                     int pad_w, const float* g_image, const float* affine, int g_lo, float* d_flow, float* d_weight,
                     double* partials, const double* var_moments, const float* upstream, const float* addend, float* part_out,
                     hipStream_t s, const GridSrc* grid_src = nullptr, int adaptive = 0, float s_norm = 0.0f, float s_tv = 0.0f,
                     double* reg_partials = nullptr, MomentsIn mj = MomentsIn{}, const HaloArg& ha = HaloArg{HALO, false, 0.0f},
                     bool finalize_uniform = true) {
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  size_t lds = (size_t)2 * TH * TW * sizeof(double) + (size_t)LH * LW * sizeof(float);
  static_assert((size_t)2 * TH * TW * sizeof(double) + (size_t)LH * LW * sizeof(float) <= 160 * 1024,
                "backward tile must fit the 160 KiB LDS of a CDNA4 CU");
  const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
  if ((size_t)2 * H * W * sizeof(float) >= ((size_t)1 << 32)) {  // d_flow is written through a buffer descriptor, 32-bit byte offsets
    set_error("ebos_iwe_*_tiled_bwd: a %dx%d flow gradient does not fit 32-bit byte offsets", H, W);
    return EBOS_ERR_UNSUPPORTED;
  }
  const bool compact = ev.cpix != nullptr;  // (per-event weights ride along in plan order: load_weights4)
  const bool dyn = ha.dyn && compact && ev.w == nullptr;  // run-time windows: the lean loop only
  void (*kb)(EvPtrs, const int32_t*, const float*, int, int, int, int, int, const float*, const float*, int, float*, float*, double*,
             const double*, const float*, const float*, float*, GridSrc, int, float, float, double*, MomentsIn, float);
  if (grid_src != nullptr) {
    if constexpr (grid_bwd_fits<TH, TW, HALO>()) {
      if (!compact || uniform || part_out == nullptr) {
        set_error("ebos_iwe_patch_tiled_bwd: needs the compact plan format, unit weights and a partials buffer");
        return EBOS_ERR_UNSUPPORTED;
      }
      kb = dyn ? iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, false, true, true>
               : iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, false, true, false>;
      if (ev.cfx != nullptr)  // ... with the fractions of undistorted events: the f64 sweep
        kb = dyn ? iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, false, true, true, true>
                 : iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, false, true, false, true>;
      lds = grid_bwd_lds<TH, TW, HALO>();
      if (int rc = reserve_lds(kb, lds, "ebos_iwe_patch_tiled_bwd")) return rc;
      const unsigned grid = (unsigned)(tiles_y * tiles_x * (adaptive ? kAdaptiveItemsPerTile : 1));
      hipEvent_t t0, t1;
      if (profile_next_pair(&t0, &t1, EBOS_PROFILE_TILED_BWD))
        hipExtLaunchKernelGGL(kb, dim3(grid), dim3(kBlock), lds, s, t0, t1, 0, ev, key_offsets, flow, H, W, tiles_x, pad_h, pad_w, g_image,
                              affine, g_lo, (float*)nullptr, (float*)nullptr, (double*)nullptr, var_moments, upstream, addend, part_out,
                              *grid_src, adaptive, s_norm, s_tv, reg_partials, mj, ha.dt_bound);
      else
        kb<<<dim3(grid), dim3(kBlock), lds, s>>>(ev, key_offsets, flow, H, W, tiles_x, pad_h, pad_w, g_image, affine, g_lo, nullptr, nullptr,
                                                 nullptr, var_moments, upstream, addend, part_out, *grid_src, adaptive, s_norm, s_tv,
                                                 reg_partials, mj, ha.dt_bound);
      return EBOS_OK;
    } else {
      set_error("ebos_iwe_patch_tiled_bwd: tile %dx%d halo %d leaves no LDS for the tile's flow (ebos_patch_fused_supported)", TH, TW, HALO);
      return EBOS_ERR_UNSUPPORTED;
    }
  }
#define EBOS_PICK(HW)                                                                                      \
  (uniform ? (compact ? iwe_dense_tiled_bwd_kernel<TH, TW, HALO, HW, FMT_COMPACT, true>                    \
                      : iwe_dense_tiled_bwd_kernel<TH, TW, HALO, HW, FMT_XY, true>)                         \
           : (compact ? iwe_dense_tiled_bwd_kernel<TH, TW, HALO, HW, FMT_COMPACT, false>                   \
                      : iwe_dense_tiled_bwd_kernel<TH, TW, HALO, HW, FMT_XY, false>))
  if (ev.w) kb = EBOS_PICK(true);
  else kb = EBOS_PICK(false);
#undef EBOS_PICK
  if (dyn)
    kb = uniform ? iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, true, false, true>
                 : iwe_dense_tiled_bwd_kernel<TH, TW, HALO, false, FMT_COMPACT, false, false, true>;
  if (int rc = reserve_lds(kb, lds, "ebos_iwe_dense_tiled_bwd")) return rc;
  const unsigned grid = (unsigned)(tiles_y * tiles_x * (part_out ? kAdaptiveItemsPerTile : 1));
  hipEvent_t t0, t1;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_TILED_BWD))
    hipExtLaunchKernelGGL(kb, dim3(grid), dim3(kBlock), lds, s, t0, t1, 0, ev, key_offsets, flow, H, W, tiles_x, pad_h, pad_w, g_image, affine,
                          g_lo, d_flow, d_weight, partials, var_moments, upstream, part_out ? (const float*)nullptr : addend, part_out,
                          GridSrc{}, 0, 0.0f, 0.0f, (double*)nullptr, mj, ha.dt_bound);
  else
    kb<<<dim3(grid), dim3(kBlock), lds, s>>>(ev, key_offsets, flow, H, W, tiles_x, pad_h, pad_w, g_image, affine, g_lo, d_flow, d_weight,
                                             partials, var_moments, upstream, part_out ? nullptr : addend, part_out, GridSrc{}, 0, 0.0f,
                                             0.0f, nullptr, mj, ha.dt_bound);
  if (part_out)
    bwd_parts_combine_kernel<TH, TW, HALO><<<dim3((unsigned)(tiles_y * tiles_x)), dim3(256), 0, s>>>(part_out, ev.part_off, tiles_x, H, W,
                                                                                                  addend, d_flow);
  if (uniform && finalize_uniform) theta_grad_finalize_kernel<<<dim3(1), dim3(256), 0, s>>>(partials, tiles_y * tiles_x, d_flow);
  return EBOS_OK;
}

}  // namespace

// (a function-local table: a namespace-scope constant would also be emitted for the device, where host functions have no address)
#define EBOS_DEFINE_SLAB_OPS(TH, TW, HL)                                                                                   \
  const SlabOps* slab_ops_##TH##x##TW##x##HL() {                                                                           \
    static const SlabOps ops = {TH, TW, HL, &launch_slab_fwd<TH, TW, HL>, &launch_slab_fwd_batch<TH, TW, HL>,             \
                                &launch_tiled_bwd<TH, TW, HL>};                                                            \
    return &ops;                                                                                                           \
  }
#endif

}  // namespace ebos
