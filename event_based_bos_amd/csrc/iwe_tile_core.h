// iwe_tile_core.h -- device code of the tile-private fused warp + IWE pipeline (gfx950), shared by the translation units that build
// kernels from it: iwe_tiled.hip (one launch per phase: accumulate / combine / finalize / backward) and cmax_resident.hip (the
// whole contrast-maximisation inner loop as ONE resident launch).  Everything lives in an anonymous namespace: each translation
// unit gets its own copy, nothing is exported.  What each piece does and why: the header comment of iwe_tiled.hip, DESIGN.md 4.
#pragma once
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include <hip/hip_ext.h>

#include "common.h"
#include "patch_grid.h"
#include "blur3.h"

namespace ebos {
namespace {

constexpr float kEps = 1e-6f;  // src/event_image_converter.py:586
constexpr int kBlock = 1024;

// In-kernel phase stamps of the accumulate kernel: diagnostic builds only (python -m event_based_bos_amd.build with
// EBOS_EXTRA_FLAGS=-DEBOS_STAMPS); values leave through a buffer nothing else reads (cdna guide, In-kernel stamps).
#ifdef EBOS_STAMPS
__device__ unsigned long long g_stamps[4096 * 8];  // (one copy per translation unit that includes this header)
__device__ unsigned long long g_stamps_bwd[4096 * 8];
#define EBOS_STAMP(k)                                                                   \
  do {                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps[blockIdx.x * 8 + (k)] = wall_clock64(); \
  } while (0)
#ifndef EBOS_STAMPS_SETUP
#define EBOS_STAMP_BWD(k)                                                                   \
  do {                                                                                      \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps_bwd[blockIdx.x * 8 + (k)] = wall_clock64(); \
  } while (0)
#define EBOS_STAMP_BWD_S(k) \
  do {                      \
  } while (0)
#else  // -DEBOS_STAMPS_SETUP: slots 2 .. 7 take the sub-steps of the backward kernel's set-up instead (tools/stamp_phases_bwd.py --setup)
#define EBOS_STAMP_BWD(k)                                                                                          \
  do {                                                                                                             \
    if ((k) < 2 && threadIdx.x == 0 && blockIdx.x < 4096) g_stamps_bwd[blockIdx.x * 8 + (k)] = wall_clock64();    \
  } while (0)
#define EBOS_STAMP_BWD_S(k)                                                                                       \
  do {                                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_stamps_bwd[blockIdx.x * 8 + (k)] = wall_clock64();               \
  } while (0)
#endif
#else
#define EBOS_STAMP_BWD_S(k) \
  do {                      \
  } while (0)
#define EBOS_STAMP(k) \
  do {                \
  } while (0)
#define EBOS_STAMP_BWD(k) \
  do {                    \
  } while (0)
#endif

struct Taps {
  int R, C;      // top-left tap (un-padded image coordinates)
  float fr, fc;  // fractional offsets
  bool ok;       // finite
};
// source-pixel-relative warp arithmetic (see iwe_fused.hip): x' = rs + (fx + dx) keeps the f32 operand <~ 32
__device__ __forceinline__ Taps warped_taps(int rs, int cs, float lx, float ly) {
  const float r0 = floorf(lx + kEps), c0 = floorf(ly + kEps);
  Taps t;
  t.fr = lx - r0;
  t.fc = ly - c0;
  t.ok = (r0 > -1e9f) && (r0 < 1e9f) && (c0 > -1e9f) && (c0 < 1e9f);
  t.R = t.ok ? rs + (int)r0 : -(1 << 20);
  t.C = t.ok ? cs + (int)c0 : -(1 << 20);
  return t;
}

// Event formats of a binned plan.  Both are read 4 consecutive events per lane (16-byte loads).
//   FMT_XY       x f32, y f32, dt f32 (plan order)                         12 B/event   any source coordinate
//   FMT_COMPACT  cpix u16 = (row_in_tile << 8) | col_in_tile, cdt f32        6 B/event   integer source coordinates
//                (camera events); tiles padded to whole groups, padding slots carry dt = NaN (ebos_plan_compact_f32)
enum EvFormat { FMT_XY = 0, FMT_COMPACT = 1 };

struct EvPtrs {
  const float* xs;
  const float* ys;
  const float* dts;
  const float* w;
  const int32_t* grp_off;  // compact plan: [tiles + 1] group offsets
  const uint16_t* cpix;
  const float* cdt;
  // adaptive work items (splits == 0, ebos_plan_parts): heavy tiles are cut into several parts.  part_off [tiles + 1]
  // = first slab of each tile; work item i (= workgroup i, heaviest first) is part item_part[i] of tile item_tile[i],
  // item_tile[i] = -1 for an unused one
  const int32_t* part_off;
  const int32_t* item_tile;
  const int32_t* item_part;
  // compact plan with FRACTIONAL source coordinates (sub-pixel rectified or pre-warped events): the fractions
  // x - floor(x), y - floor(y) per slot, laid out like cdt; nullptr = integer source pixels.  Read by the general loops of the
  // compact format (FRAC kernels: the resident 2-DoF loop) -- the lean hot loops assume integer pixels.
  const float* cfx;
  const float* cfy;
};

struct Group {  // 4 consecutive events of one lane
  int rs[4], cs[4];   // source pixel (global)
  float fx[4], fy[4]; // fractional part of the source coordinate (0 in the compact format)
  float dt[4], w[4];  // w == 0 marks a dead slot (outside this workgroup's slice / padding)
};

struct TileRange {
  int ty, tx;
  int slab;                  // index of the slab this workgroup writes
  int part;                  // which part of its tile this workgroup is (0 when tiles are not split)
  int32_t beg, end;          // FMT_XY: this workgroup's slice of the tile's events (plan order)
  int32_t g_first, g_last;   // groups of 4 this workgroup reads (g_first > g_last: nothing to do)
  int32_t tile_groups;       // groups of the WHOLE tile (0: an empty tile -- the kernels' set-up has short cuts for it; set by tile_range only)
};

template <int FMT>
__device__ __forceinline__ TileRange tile_range(const int32_t* __restrict__ key_offsets, const EvPtrs& ev, int tile_px,
                                                int tiles_x, int splits) {
  TileRange r;
  int tile, part;
  if (splits == 0) {  // adaptive: this workgroup is one work item of the plan's part table
    tile = ev.item_tile[blockIdx.x];
    if (tile < 0) {  // unused item
      r.ty = r.tx = -1;
      r.slab = r.part = r.beg = r.end = r.g_first = r.tile_groups = 0;
      r.g_last = -1;
      return r;
    }
    part = ev.item_part[blockIdx.x];
    splits = ev.part_off[tile + 1] - ev.part_off[tile];
    r.slab = ev.part_off[tile] + part;
  } else {
    tile = blockIdx.x / splits;
    part = blockIdx.x - tile * splits;
    r.slab = blockIdx.x;
  }
  r.ty = tile / tiles_x;
  r.tx = tile - r.ty * tiles_x;
  r.part = part;
  if (FMT == FMT_COMPACT) {
    const int32_t g0 = ev.grp_off[tile], g1 = ev.grp_off[tile + 1];
    const int32_t chunk = (g1 - g0 + splits - 1) / splits;
    const int32_t gb = min(g1, g0 + part * chunk), ge = min(g1, gb + chunk);
    r.g_first = gb;
    r.g_last = ge - 1;
    r.tile_groups = g1 - g0;
    r.beg = key_offsets[tile * tile_px] + 4 * (gb - g0);  // plan index of the first slot
    r.end = key_offsets[(tile + 1) * tile_px];
  } else {
    const int32_t beg = key_offsets[tile * tile_px], end = key_offsets[(tile + 1) * tile_px];
    int32_t chunk = (end - beg + splits - 1) / splits;
    chunk = (chunk + kWave - 1) & ~(kWave - 1);
    r.beg = min(end, beg + part * chunk);
    r.end = min(end, r.beg + chunk);
    r.g_first = r.beg >> 2;
    r.g_last = r.beg < r.end ? (r.end - 1) >> 2 : r.g_first - 1;
    r.tile_groups = (end - beg + 3) >> 2;
  }
  return r;
}

// Per-event weights are stored in PLAN order (the caller permutes them like the events).  A group of the (x, y, dt) format is four
// consecutive plan indices 4 j .. 4 j + 3; a group of the compact format is four consecutive events of its TILE, whose slots are the
// tile's plan indices in order (padded to a multiple of four at the tile's end): group j of a work item that starts at group g_first /
// plan index beg holds beg + 4 (j - g_first) .. + 3 -- not 16-byte aligned in general (a dword-aligned 16-byte load; the weight array
// is padded by a group).
template <int FMT>
__device__ __forceinline__ int32_t group_plan_index(int32_t j, const TileRange& tr) {
  return FMT == FMT_COMPACT ? tr.beg + 4 * (j - tr.g_first) : 4 * j;
}
template <int FMT>
__device__ __forceinline__ float4 load_weights4(const float* __restrict__ w, int32_t j, const TileRange& tr) {
  float4 v;
  if (FMT == FMT_COMPACT) __builtin_memcpy(&v, w + group_plan_index<FMT>(j, tr), sizeof(float4));
  else v = reinterpret_cast<const float4*>(w)[j];
  return v;
}

// Branch-free: group indices past the slice are clamped (the loops never PROCESS such a group, they only prefetch
// it).  Predicated loads would become exec-masked branches and hipcc then waits vmcnt(0) for them, draining the
// prefetch that is supposed to stay in flight.
template <int FMT, bool HAS_W, int TH, int TW>
__device__ __forceinline__ void load_group(Group& g, int32_t grp, const TileRange& tr, const EvPtrs& p, int tile_r0,
                                           int tile_c0) {
  const int32_t j = max(min(grp, tr.g_last), tr.g_first);
  if (FMT == FMT_COMPACT) {
    const float4 D = reinterpret_cast<const float4*>(p.cdt)[j];
    const uint2 P = reinterpret_cast<const uint2*>(p.cpix)[j];
    float4 Wv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (HAS_W) Wv = load_weights4<FMT_COMPACT>(p.w, j, tr);
    const float dd[4] = {D.x, D.y, D.z, D.w}, ww[4] = {Wv.x, Wv.y, Wv.z, Wv.w};
    const unsigned pp[4] = {P.x & 0xffffu, P.x >> 16, P.y & 0xffffu, P.y >> 16};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool live = dd[e] == dd[e];  // padding slots carry NaN
      g.dt[e] = live ? dd[e] : 0.0f;
      g.w[e] = live ? ww[e] : 0.0f;
      g.rs[e] = tile_r0 + (int)(pp[e] >> 8);
      g.cs[e] = tile_c0 + (int)(pp[e] & 255u);
      g.fx[e] = 0.0f;
      g.fy[e] = 0.0f;
    }
    if (p.cfx != nullptr) {  // (uniform) fractional source coordinates
      const float4 FX = reinterpret_cast<const float4*>(p.cfx)[j], FY = reinterpret_cast<const float4*>(p.cfy)[j];
      g.fx[0] = FX.x, g.fx[1] = FX.y, g.fx[2] = FX.z, g.fx[3] = FX.w;
      g.fy[0] = FY.x, g.fy[1] = FY.y, g.fy[2] = FY.z, g.fy[3] = FY.w;
    }
  } else {
    const float4 D = reinterpret_cast<const float4*>(p.dts)[j];
    float4 Wv = make_float4(1.f, 1.f, 1.f, 1.f);
    if (HAS_W) Wv = reinterpret_cast<const float4*>(p.w)[j];
    const float4 X = reinterpret_cast<const float4*>(p.xs)[j], Y = reinterpret_cast<const float4*>(p.ys)[j];
    const float dd[4] = {D.x, D.y, D.z, D.w}, ww[4] = {Wv.x, Wv.y, Wv.z, Wv.w};
    const float xx[4] = {X.x, X.y, X.z, X.w}, yy[4] = {Y.x, Y.y, Y.z, Y.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int32_t i = 4 * j + e;
      const bool live = i >= tr.beg && i < tr.end;
      g.dt[e] = dd[e];
      g.w[e] = live ? ww[e] : 0.0f;
      const float x = live ? xx[e] : (float)tile_r0, y = live ? yy[e] : (float)tile_c0;
      g.rs[e] = (int)x;
      g.cs[e] = (int)y;
      g.fx[e] = x - (float)g.rs[e];
      g.fy[e] = y - (float)g.cs[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// forward A: accumulate one (tile, split) in LDS, write it as a slab
// ---------------------------------------------------------------------------------------------------
// Two LDS accumulation modes (same 8 bytes per cell):
//   F64  one double per cell, 4 ds_add_f64 per event.  Exact for any per-event weight.
//   FX   fixed point, scale 2^kFxShift.  The two horizontally adjacent taps of a row share ONE 64-bit
//        word as two signed 32-bit fields (hi = column c+1, lo = column c), so an event costs 2
//        ds_add_u64 (measured ~4.6 lanes/clk/CU vs ~2.8 for ds_add_f64, and half as many of them).
//        Plane A holds the pairs starting at even columns, plane B those starting at odd columns.
//        Integer adds commute: the tile sum is exact and bit-reproducible.  A field overflows only if
//        one cell collects more than 2^(32-kFxShift) = 4096 units of weight inside one workgroup; that
//        is DETECTED exactly -- every wrap lowers the 64-bit sum of the decoded fields by 2^32 - 1 (low
//        field, carry into the neighbour) or 2^32 (high field, carry out of the word), never raises it,
//        so sum(decoded fields) != sum(added) -- and the workgroup then redoes its slice in F64 mode.
//        Used for unit weights (the hot path).
constexpr int kFxShift = 20;
constexpr float kFxScale = (float)(1 << kFxShift);
constexpr double kFxInv = 1.0 / (double)(1 << kFxShift);

enum AccMode { ACC_F64 = 0, ACC_FX = 1 };

// ---- the hot loop: compact plan, unit weights, verified fixed point --------------------------------------------
// Written against an instruction budget (~45 VALU per event; the kernel is VALU-bound on MI355X):
//   * compact groups need no liveness logic: padding slots carry dt = NaN, which fails the `ok` test below;
//   * tile-local pixel is byte-packed (row << 8 | col): two bit-field extracts;
//   * flow gathers use an unsigned 32-bit element offset off a uniform base (hardware address form, no 64-bit math);
//   * taps outside the window are redirected to a dummy LDS region instead of being masked: the adds stay
//     branch-free and their (garbage) values are never read;
//   * fractions are clamped at 0, so all four fixed-point taps are non-negative and a word is simply (hi << 32) | lo;
//   * rounding by the magic-number trick (one FMA + one integer subtract per tap).
// GRID kernels: the flow is a patch grid [2, gh, gw] (src/solver/patch_eklt.py:173-204); every workgroup evaluates the
// grid -> dense map for its own source tile into LDS ([2][TH * TW] floats behind the accumulators) and the event loop
// fetches the flow from there -- no dense [2, H, W] field in memory, no upsample launch, LDS reads instead of L2 gathers.
struct GridSrc {
  Axis ay, ax;
};

// variance of the IWE as the (sum, sum of squares) partials the combine pass left (want_variance = 2): the GRID backward
// kernel reduces them itself -- every workgroup, redundantly, ~14 KB of L2 reads -- instead of waiting for a finalize launch
struct MomentsIn {
  const double* partials;  // nullptr: not used
  int64_t n_partials, n_pixels;
  float* out_var;          // written by workgroup 0 (nullable)
  double* moments;         // (mean, M), written by workgroup 0 (nullable)
  float* out_scaled;       // upstream[0] * value, written by workgroup 0 (nullable): the SIGNED loss of a cost with a direction,
                           // without a launch of its own to negate a scalar
  // mode 0: (sum, sum of squares) pairs of the combine pass -> variance, folded into the upstream as an affine map of the IWE;
  // mode 1: one value partial per workgroup of gradmag_fused_kernel -> out_var[0] = sum / n_pixels, by workgroup 0 only (the upstream
  //         image is the gradient image itself: nothing to fold; the finalize launch of the contrast value disappears)
  int mode;
  // blur.k0 != 0 (mode 0): the contrast is taken on the 3-tap blurred image y = B x (blur3.h).  The partials are then those of
  // blur3_variance_adjoint_kernel -- (sum, sum of squares) of the valid blurred pixels -- and g_image is its z = B^T (m . y): the
  // upstream is a z + c wgt with wgt = B^T m evaluated from the pixel's position, the valid region being part of z and wgt already
  Blur3 blur;
};

// AP = apron in pixels around the tile (0 forward; 2 backward: the image_gradient regulariser reads neighbours up to 2 px away).
// s_lerp [(TH + 2 AP) + (TW + 2 AP)]: row / column interpolation of the tile + apron (clamped to the image);
// s_cells [2][kGridCells][kGridCells]: the block of grid cells they touch; s_flow [2][TH + 2 AP][TW + 2 AP].
// Two halves with ONE global round trip (the cell block) between them, which the caller fills with its own set-up work (LDS
// clear, upstream-tile staging) -- sampling the grid straight from global memory cost 3-4 dependent L2 round trips per
// workgroup.  2 * kGridCells^2 <= kBlock: one cell value per thread.
struct TileGrid {
  int gi0, ni, gj0, nj;
  float cell;  // this thread's value of the cell block (threads >= 2 ni nj: unused)
};

template <int TH, int TW, int AP>
__device__ __forceinline__ TileGrid tile_grid_begin(const float* __restrict__ grid, const GridSrc& gs, int tr0, int tc0, int H, int W,
                                                    Lerp* s_lerp) {
  static_assert(2 * kGridCells * kGridCells <= kBlock, "one cell value per thread");
  constexpr int PH = TH + 2 * AP, PW = TW + 2 * AP;
  for (int i = threadIdx.x; i < PH + PW; i += kBlock)
    s_lerp[i] = i < PH ? lerp_at(gs.ay, min(max(tr0 + i - AP, 0), H - 1)) : lerp_at(gs.ax, min(max(tc0 + i - PH - AP, 0), W - 1));
  __syncthreads();
  TileGrid t;
  t.gi0 = s_lerp[0].i0, t.ni = s_lerp[PH - 1].i1 - t.gi0 + 1;          // (indices are monotone in the clamped coordinate)
  t.gj0 = s_lerp[PH].i0, t.nj = s_lerp[PH + PW - 1].i1 - t.gj0 + 1;
  const int idx = min((int)threadIdx.x, 2 * t.ni * t.nj - 1);  // clamped: the load is unconditional
  const int ch = idx / (t.ni * t.nj), rem = idx - ch * (t.ni * t.nj);
  const int i = rem / t.nj, j = rem - i * t.nj;
  t.cell = grid[((int64_t)ch * gs.ay.g + t.gi0 + i) * gs.ax.g + t.gj0 + j];
  return t;
}

// The dense flow of a PH x PW pixel block from the block of grid cells in LDS (s_rows [PH], s_cols [PW]: the pixels' row / column
// interpolation; s_cells [2][kGridCells][kGridCells] starting at cell (gi0, gj0)) -> s_flow [2][PH * PW].
// (Four consecutive columns per thread -- the row's Lerp and, where the four interpolate between the same two cells, the eight cell
// values read once: 15 LDS instructions per four pixels instead of 48 -- did not move the pass: 2.24 us for a 45 x 80 tile either way.)
template <int PH, int PW>
__device__ __forceinline__ void cells_to_flow(const Lerp* s_rows, const Lerp* s_cols, const float* s_cells, int gi0, int gj0,
                                              float* s_flow) {
  constexpr int kPlane = kGridCells * kGridCells;
  for (int i = threadIdx.x; i < PH * PW; i += kBlock) {
    const int rl = i / PW, cl = i - rl * PW;
    Lerp ly = s_rows[rl], lx = s_cols[cl];
    lx.i0 -= gj0;  // indices into the cell block
    lx.i1 -= gj0;
    const float* u0 = s_cells + (ly.i0 - gi0) * kGridCells;
    const float* u1 = s_cells + (ly.i1 - gi0) * kGridCells;
    s_flow[i] = grid_bilerp(u0, u1, ly, lx);
    s_flow[PH * PW + i] = grid_bilerp(u0 + kPlane, u1 + kPlane, ly, lx);
  }
}

// second half: cell block -> LDS, then the flow of tile + apron.  Ends with a barrier.
template <int TH, int TW, int AP>
__device__ __forceinline__ void tile_grid_finish(const TileGrid& t, float* s_flow, const Lerp* s_lerp, float* s_cells) {
  constexpr int PH = TH + 2 * AP, PW = TW + 2 * AP;
  if ((int)threadIdx.x < 2 * t.ni * t.nj) {
    const int ch = threadIdx.x / (t.ni * t.nj), rem = threadIdx.x - ch * (t.ni * t.nj);
    const int i = rem / t.nj, j = rem - i * t.nj;
    s_cells[(ch * kGridCells + i) * kGridCells + j] = t.cell;
  }
  __syncthreads();
  cells_to_flow<PH, PW>(s_lerp, s_lerp + PH, s_cells, t.gi0, t.gj0, s_flow);
  __syncthreads();
}

constexpr int kBwdApron = 2;  // pixels of flow the GRID backward kernel keeps around its tile

struct CGroup {
  unsigned pr[4], pc[4];  // tile-local source row / column
  float dt[4];
  float fx[4], fy[4];     // fractional part of the source coordinate (0 unless the plan carries fractions: EvPtrs::cfx)
};
template <int TH, int TW>
__device__ __forceinline__ void load_cgroup(CGroup& g, int32_t grp, const TileRange& tr, const EvPtrs& p) {
  const int32_t j = max(min(grp, tr.g_last), tr.g_first);
  const float4 D = reinterpret_cast<const float4*>(p.cdt)[j];
  const uint2 P = reinterpret_cast<const uint2*>(p.cpix)[j];
#pragma unroll
  for (int e = 0; e < 4; ++e) g.fx[e] = g.fy[e] = 0.0f;
  if (p.cfx != nullptr) {  // (uniform)
    const float4 FX = reinterpret_cast<const float4*>(p.cfx)[j], FY = reinterpret_cast<const float4*>(p.cfy)[j];
    g.fx[0] = FX.x, g.fx[1] = FX.y, g.fx[2] = FX.z, g.fx[3] = FX.w;
    g.fy[0] = FY.x, g.fy[1] = FY.y, g.fy[2] = FY.z, g.fy[3] = FY.w;
  }
  g.dt[0] = D.x; g.dt[1] = D.y; g.dt[2] = D.z; g.dt[3] = D.w;
  g.pr[0] = (P.x >> 8) & 255u; g.pc[0] = P.x & 255u;
  g.pr[1] = P.x >> 24;         g.pc[1] = (P.x >> 16) & 255u;
  g.pr[2] = (P.y >> 8) & 255u; g.pc[2] = P.y & 255u;
  g.pr[3] = P.y >> 24;         g.pc[3] = (P.y >> 16) & 255u;
}

// The LDS window of one work item: its source tile plus hr rows / hc columns of halo on each side.
//   static kernels (DYN = false): hr = hc = HALO, every expression below folds to the compile-time constant;
//   DYN kernels: HALO is the LARGEST window (it sizes the LDS and the slab stride) and (hr, hc) are chosen per tile AT RUN TIME
//   from a bound on the tile's own displacements, |flow| over the tile x max |dt| of the window (BOS displacements are a few
//   pixels; the +-30 px of the sampler range is the search bound, not the operating point).  Clear, decode, slab store,
//   combine reads and the backward kernel's upstream tile all scale with (TH + 2 hr)(TW + 2 hc); taps beyond the window
//   still go to the spill image, so the choice moves time, never results.  hc is a multiple of 4 (slab rows are written
//   and combined 4 cells at a time).
template <int TH, int TW, int HALO, bool DYN>
struct Win {
  int hr, hc;
  __device__ __forceinline__ int HR() const { return DYN ? hr : HALO; }
  __device__ __forceinline__ int HC() const { return DYN ? hc : HALO; }
  __device__ __forceinline__ int LH() const { return TH + 2 * HR(); }
  __device__ __forceinline__ int LW() const { return TW + 2 * HC(); }
  // Row pitch of the window IN LDS, in cells (= dwords of a paired-word plane row).  A run-time width that is a multiple of 32
  // dwords (80 + 2 x 8 = 96, 80 + 2 x 24 = 128) puts the two rows an event adds to -- and every vertically adjacent pair of
  // events -- on the same banks: the forward loop ran 26 us instead of 21 at 6 px flows (hc = 8).  The pitch is therefore the
  // width rounded up to 16 modulo 32, what the built windows happen to have (112, 144); slabs stay dense (LW floats per row).
  __device__ __forceinline__ int P() const { return DYN ? ((LW() - 16 + 31) / 32) * 32 + 16 : LW(); }
};
// cells of the LDS image of a kernel (accumulators + the dummy region that absorbs the adds of out-of-window lanes): the largest
// window, at ITS pitch when the kernel chooses windows at run time
template <int TH, int TW, int HALO, bool DYN>
constexpr int acc_cells() {
  constexpr int lw = TW + 2 * HALO, pt = DYN ? ((lw - 16 + 31) / 32) * 32 + 16 : lw;
  return (TH + 2 * HALO) * pt + pt / 2 + 2;
}
// halo that keeps every tap of a displacement of at most `disp` pixels inside the window: floor(x + eps) moves by <= ceil(disp)
// and the second tap sits one further
__device__ __forceinline__ int halo_for(float disp, int align, int cap) {
  const int need = (int)ceilf(fminf(disp, 4096.0f)) + 1;  // (NaN -> fminf gives 4096 -> the cap)
  return min(cap, (max(need, 1) + align - 1) / align * align);
}
// (hr, hc) as one table word: the accumulate pass tells the combine pass what it stored for a tile
__device__ __forceinline__ unsigned win_pack(int hr, int hc) { return (unsigned)hr | ((unsigned)hc << 8); }

// The forward loop's form of a group: the tile-local column comes as a BYTE offset inside a row of the LDS window,
// pcq = 4 * (col_in_tile + hc) -- the unit every address of that loop is computed in (flow gather offset, LDS word)
struct CGroupQ {
  unsigned pr[4], pcq[4];
  float dt[4];
};
// a group as it comes from memory (24 bytes): what the persistent batched kernel holds while it prefetches a window's first chunks
struct CRaw {
  float4 D;
  uint2 P;
};
__device__ __forceinline__ CRaw load_craw(int32_t grp, const TileRange& tr, const EvPtrs& p) {
  const int32_t j = max(min(grp, tr.g_last), tr.g_first);
  return CRaw{reinterpret_cast<const float4*>(p.cdt)[j], reinterpret_cast<const uint2*>(p.cpix)[j]};
}
// ... and as the backward kernel's fixed-point sweep wants it: the tile-local pixel index and the offset of the pixel's flow value are
// formed HERE, in the block that extracts the bit fields -- the 24-bit multiplier (full rate; v_mul_lo_u32 / v_mad_u64_u32 are
// quarter rate) is only selected where the compiler can prove the operand ranges, and what it knows about a value decoded in an
// earlier loop iteration does not reach the block that uses it.
typedef float v2f __attribute__((ext_vector_type(2)));
struct BGroup {
  unsigned pr[4], pc[4], pix[4], goff[4];
  float dt[4];
};
__device__ __forceinline__ void decode_bgroup(BGroup& g, const CRaw& raw, unsigned row_pitch, unsigned shift, unsigned base) {
  const float dd[4] = {raw.D.x, raw.D.y, raw.D.z, raw.D.w};
  const unsigned pp[4] = {raw.P.x & 0xffffu, raw.P.x >> 16, raw.P.y & 0xffffu, raw.P.y >> 16};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    g.dt[e] = dd[e];
    g.pr[e] = pp[e] >> 8;
    g.pc[e] = pp[e] & 255u;
    g.goff[e] = base + __umul24(g.pr[e], row_pitch) + (g.pc[e] << shift);  // (byte offset | element index of the pixel's flow value)
  }
}
template <int TW>
__device__ __forceinline__ void finish_bgroup(BGroup& g) {
#pragma unroll
  for (int e = 0; e < 4; ++e) g.pix[e] = g.pr[e] * (unsigned)TW + g.pc[e];
}
template <int TH, int TW>
__device__ __forceinline__ void load_bgroup(BGroup& g, int32_t grp, const TileRange& tr, const EvPtrs& p, unsigned row_pitch,
                                            unsigned shift, unsigned base) {
  decode_bgroup(g, load_craw(grp, tr, p), row_pitch, shift, base);
  finish_bgroup<TW>(g);
}

// What the fixed-point sweep of the backward kernel needs before its first step, requested while the kernel still sets up: the
// first two chunks of every wave (event loads: a round trip) and, dense field, the flow values of the first (gathers: a second,
// dependent round trip).  Issued with the upstream tile's staging loads they arrive under its LDS stores and the barrier; left to
// the sweep they were 2 - 3 us in front of its first multiply (in-kernel stamps).
struct BwdPre {
  BGroup A, B;
  float au[4], av[4];
};
struct BwdPreRaw {
  CRaw A, B;
};

__device__ __forceinline__ void decode_craw(CGroupQ& g, const CRaw& r, unsigned hc4);
__device__ __forceinline__ void load_cgroup_q(CGroupQ& g, int32_t grp, const TileRange& tr, const EvPtrs& p, unsigned hc4) {
  decode_craw(g, load_craw(grp, tr, p), hc4);
}
__device__ __forceinline__ void decode_craw(CGroupQ& g, const CRaw& r, unsigned hc4) {
  const float4 D = r.D;
  const uint2 P = r.P;
  g.dt[0] = D.x; g.dt[1] = D.y; g.dt[2] = D.z; g.dt[3] = D.w;
  g.pr[0] = (P.x >> 8) & 255u; g.pcq[0] = ((P.x & 255u) << 2) + hc4;
  g.pr[1] = P.x >> 24;         g.pcq[1] = (((P.x >> 16) & 255u) << 2) + hc4;
  g.pr[2] = (P.y >> 8) & 255u; g.pcq[2] = ((P.y & 255u) << 2) + hc4;
  g.pr[3] = P.y >> 24;         g.pcq[3] = (((P.y >> 16) & 255u) << 2) + hc4;
}

// Slabs travel from the accumulate pass to the combine pass WRITE-THROUGH (sc1 buffer stores): the 16 MB a launch stores do not sit
// dirty in the L2s until the end-of-kernel release writes them back in one burst (1.7 us of the accumulate pass).  Every load of
// them is an sc1 load, the reader's half of that hand-off form in the cdna guide (it costs nothing measurable).  Builtins, not
// inline asm: the compiler then owns the wait states around the stores' data registers.
#ifndef EBOS_PLAIN_SLABS
typedef unsigned slab_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxSc1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void slab_store4(__amdgpu_buffer_rsrc_t r, unsigned byte, float4 v) {
  const slab_u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte, 0, kAuxSc1);
}
__device__ __forceinline__ float4 slab_load4(__amdgpu_buffer_rsrc_t r, unsigned byte) {
  const slab_u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte, 0, kAuxSc1);
  return make_float4(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w));
}
__device__ __forceinline__ float slab_load1(__amdgpu_buffer_rsrc_t r, unsigned byte) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte, 0, kAuxSc1));
}
#endif

// Work distribution inside the workgroup is DYNAMIC: a wave processes one chunk of 64 groups (one group per lane) at a
// time and draws its next chunk from an LDS counter.  With a static stride the 16 waves finish far apart -- the SIMD
// arbiter favours older waves, so wave 0 was done after 9 us and then sat 7 us at the barrier (in-kernel stamps) --
// and the tail runs at low occupancy.
struct ChunkQueue {
  unsigned* next;  // LDS counter, initialised to 2 * waves (chunks 0 .. 2 * waves - 1 are pre-assigned)
  __device__ __forceinline__ int pull() const {
    unsigned c = 0;
    if ((threadIdx.x & (kWave - 1)) == 0) c = atomicAdd(next, 1u);
    return (int)__builtin_amdgcn_readfirstlane(c);
  }
};

// Ablation build (-DEBOS_ABL_MULTIK=K, results wrong on purpose): every event is warped and accumulated K times with K different
// translations into the SAME image -- what K hypotheses per event read would cost per hypothesis if the K images were free
// (tools/ab_multik.sh, profiles/r03_multi_hypothesis_ablation.txt).
#ifdef EBOS_ABL_MULTIK
#define EBOS_KLOOP for (int kk = 0; kk < EBOS_ABL_MULTIK; ++kk)
#define EBOS_KOFF(v, s) ((v) + (s) * (float)kk)
#else
#define EBOS_KLOOP
#define EBOS_KOFF(v, s) (v)
#endif

// lane l of chunk c takes group 64 c + l (plain mapping: one group of 4 events per lane, chunks of 64 groups)
__device__ __forceinline__ int32_t chunk_group(int32_t g_first, int c, int lane) { return g_first + c * kWave + lane; }

// MODE == ACC_F64: the same loop with four ds_add_f64 per event -- the exact redo of a slice whose fixed-point fields
// overflowed (hot pixels, or a flow that piles thousands of events onto one cell).
// PAIRS: lane l of a chunk of 128 groups takes groups 2 l and 2 l + 1, one per loop step, instead of group l of a chunk of 64.  With
// pixel-sorted events and ~11 events per source pixel the lanes of ONE wave instruction then hold events of different pixels, and
// their ds_add_u64 no longer meet on the same words: SQ_LDS_ADDR_CONFLICT per launch is 0.33 M at 30 px flows but 2.3 M at 6 px,
// 4.8 M at 2 px and 5.8 M at 0.5 px with the plain mapping -- the loop ran 18.5 / 22.0 / 27.7 / 30.4 us; BOS flows are the small
// ones (profiles/r03_small_flow_conflicts.txt).  A lane's two groups are adjacent in memory: its loads stay coalesced.
// HAS_W (plain mapping, fixed point): per-event weights ride along in plan order (load_weights4, 10 B / event).  An event's unit is
// then U = rint(w * wscale) <= 2^20 (wscale = 2^20 / max |w| of the slice, TileShared) instead of the constant 2^20, split between
// the four taps by the same exact scheme; the units put inside the window are summed per lane (the return value) -- three
// multiply-adds and a 64-bit add per event on top of the unit-weight loop.
template <int TH, int TW, int HALO, bool UNIFORM, int MODE = ACC_FX, bool GRID = false, bool DYN = false, bool PAIRS = false,
          bool HAS_W = false>
__device__ __forceinline__ unsigned long long accumulate_compact_fx(const TileRange& tr, double* s_acc, const EvPtrs& ev,
                                                          const float* __restrict__ flow, int H, int W, bool* any_spill,
                                                          const ChunkQueue& queue, const Win<TH, TW, HALO, DYN>& win,
                                                          const CRaw* pre = nullptr, float wscale = kFxScale) {
  static_assert(!HAS_W || (MODE == ACC_FX && !PAIRS && !DYN), "per-event weights: the plain fixed-point loop on the largest window");
  constexpr bool kMerge = PAIRS && MODE == ACC_FX;   // same-cell events of a lane merged before the add (see `deposit`)
  const int LH = win.LH(), LW = win.LW(), HR = win.HR(), HC = win.HC(), PT = win.P();  // (compile-time constants unless DYN)
  const unsigned kPlane = LH * PT / 2;  // words per plane
  const unsigned kDummy = LH * PT;      // first word of the dummy region (PT / 2 + 2 words)
  constexpr float kMagic = 12582912.0f;     // 1.5 * 2^23
  constexpr int kMagicBits = 0x4B400000;
  // GRID: `flow` is the tile's own [2][TH * TW] flow in LDS
  const float* __restrict__ flow1 = UNIFORM ? flow : flow + (GRID ? (int64_t)TH * TW : (int64_t)H * W);
  const float uni_u = UNIFORM ? -flow[0] : 0.0f, uni_v = UNIFORM ? -flow[1] : 0.0f;
  const unsigned base_lin = GRID ? 0u : (unsigned)(tr.ty * TH * W + tr.tx * TW);
  const unsigned uW = GRID ? (unsigned)TW : (unsigned)W;
  // Dense field in memory: the two gathers per event go through a buffer descriptor -- a 32-bit byte offset per lane
  // (one v_mad_u32_u24 + one shift) plus a scalar offset for the tile origin / the second component, instead of
  // v_mul_lo_u32 + 64-bit address arithmetic per load (the loop is VALU-throughput-bound: DESIGN 4.1 #18)
  constexpr bool kBuf = !UNIFORM && !GRID;
  // (pcq carries 4 * HC: the descriptor's base is moved back by as much -- a scalar offset must not go negative, the address
  // unit adds it as an unsigned 32-bit value -- and the range check gets the same allowance)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(flow)) - 4 * HC, 0, kBuf ? 2 * H * W * (int)sizeof(float) + 4 * HC : 0, 0x00020000);
  const int soff0 = (int)(base_lin * 4u), soff1 = soff0 + H * W * (int)sizeof(float);
  const unsigned uW4 = uW * 4u;
  const char* __restrict__ flow_b0 = reinterpret_cast<const char*>(flow) - 4 * HC;   // GRID: byte-addressed LDS reads
  const char* __restrict__ flow_b1 = reinterpret_cast<const char*>(flow1) - 4 * HC;
  const unsigned hc4 = 4u * (unsigned)HC, lw4 = 4u * (unsigned)PT;
  auto fetch = [&](unsigned pr, unsigned pcq, float& u, float& v) {
    if (UNIFORM) {
      u = uni_u, v = uni_v;
    } else {
      const unsigned off = __umul24(pr, uW4) + pcq;
      if (kBuf) {
        u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff0, 0));
        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff1, 0));
      } else {
        u = *reinterpret_cast<const float*>(flow_b0 + off);
        v = *reinterpret_cast<const float*>(flow_b1 + off);
      }
    }
  };
  bool spilled = false;
  const int32_t g_last = tr.g_last;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  constexpr int kWaves = kBlock / kWave;
  char* const s_bytes = reinterpret_cast<char*>(s_acc);
  unsigned long long added = 0ull;  // HAS_W: the units this lane put inside the window
  // the four events of one group into the LDS image (fu / fv: their flow, gathered one step earlier; wq: their weights)
  auto deposit = [&](const CGroupQ& cur, const float* fu, const float* fv, bool lane_live, const float4& wq4 = float4{}) {
    const float wq[4] = {wq4.x, wq4.y, wq4.z, wq4.w};
    unsigned m_byte = 0u, m00 = 0u, m01 = 0u, m10 = 0u, m11 = 0u;  // kMerge: the run of events that share a cell
#pragma unroll
    for (int e = 0; e < 4; ++e) EBOS_KLOOP {
      const float lx = -cur.dt[e] * EBOS_KOFF(fu[e], 0.37f), ly = -cur.dt[e] * EBOS_KOFF(fv[e], -0.21f);  // source coordinates are integers: x' = rs + lx
      const float r0 = floorf(lx + kEps), c0 = floorf(ly + kEps);
      const float fr = fmaxf(lx - r0, 0.0f), fc = fmaxf(ly - c0, 0.0f);
      // row of the LDS window, and 4 x its column (= byte offset of the column's f32 / half the offset of its pair word)
      const int rl = (int)cur.pr[e] + HR + (int)r0;
      const int cl4 = (int)cur.pcq[e] + ((int)c0 << 2);
      // false for NaN (padding slots carry dt = NaN), +-Inf and anything beyond 2^29: the column is tested in units of 4 (cl4), and
      // (int)c0 << 2 WRAPS for |c0| >= 2^29 (INT_MAX << 2 = -4, INT_MIN << 2 = 0): such an event would pass the window test and be
      // deposited at column pc - 1 or pc with full weight -- the reference masks it out
      const bool ok = lane_live && (fabsf(lx) + fabsf(ly) < 5.0e8f);
      const bool inside = ok && (unsigned)rl < (unsigned)(LH - 1) && (unsigned)cl4 < (unsigned)(4 * (LW - 1));
      spilled |= ok && !inside;
      if (MODE == ACC_F64) {  // one double per cell; out-of-window / padding lanes add +0.0 to cell (0, 0) (select, not
        const float a = 1.0f - fr, b = 1.0f - fc;  // multiply: their weights may be NaN)
        const int cl = cl4 >> 2;
        double* cell = s_acc + (inside ? rl * PT + cl : 0);
        atomic_add(cell, (double)(inside ? a * b : 0.0f));
        atomic_add(cell + 1, (double)(inside ? a * fc : 0.0f));
        atomic_add(cell + PT, (double)(inside ? fr * b : 0.0f));
        atomic_add(cell + PT + 1, (double)(inside ? fr * fc : 0.0f));
        continue;
      }
      // The event's unit of weight (2^20) is split exactly: first between the two rows, then each row between its two
      // columns -- every tap is a non-negative integer and the four sum to 2^20 whatever the rounding, so the checksum
      // of a wave is 2^20 x (events it put inside the window): a mask population count, no per-event arithmetic.
      const float b = 1.0f - fc;
      float unit = kFxScale;  // (a compile-time constant without weights)
      if (HAS_W) {            // U = rint(w * wscale): an integer-valued float <= 2^20 (w >= 0 and finite: the set-up's pre-pass saw to it)
        const float tu = __fmaf_rn(wq[e], wscale, kMagic);
        unit = tu - kMagic;
        added += inside ? (unsigned long long)(unsigned)(__float_as_int(tu) - kMagicBits) : 0ull;
      }
      const float t0 = __fmaf_rn(fr, -unit, unit + kMagic);  // magic + A0,  A0 = rint(U (1 - fr))   (U = 2^20 without weights)
      const float a0 = t0 - kMagic, a1 = unit - a0;           // A0, A1 = U - A0 as floats (exact)
      const float u0 = __fmaf_rn(a0, b, kMagic), u1 = __fmaf_rn(a1, b, kMagic);
      const float t1 = a1 + kMagic;
      const unsigned q00 = (unsigned)(__float_as_int(u0) - kMagicBits);
      const unsigned q01 = (unsigned)(__float_as_int(t0) - __float_as_int(u0));  // A0 - q00
      const unsigned q10 = (unsigned)(__float_as_int(u1) - kMagicBits);
      const unsigned q11 = (unsigned)(__float_as_int(t1) - __float_as_int(u1));  // A1 - q10
      // Byte address of the pair word: with t = rl LW + cl the word is (t >> 1) + (t & 1) kPlane, i.e. byte
      //   8 (t >> 1) + 8 kPlane (t & 1) = 4 t + (t & 1) (8 kPlane - 4),   and t & 1 = cl & 1 (LW is even)
      // -- two multiply-adds and a bit-field extract on values the loop has anyway (4 cl), instead of and / compare / select /
      // shift / add on t (four VALU instructions per event fewer, one of them a compare: DESIGN 4.1 #18).
      const unsigned t4 = __umul24((unsigned)rl, lw4) + (unsigned)cl4;  // (rl < 2^24 whenever the result is used)
      const unsigned byte = __umul24(((unsigned)cl4 >> 2) & 1u, 8u * kPlane - 4u) + t4;
      if constexpr (kMerge) {
        // A lane's consecutive events that share a cell leave as ONE pair of adds (their integer taps summed in registers: <= 4 x 2^20
        // per field).  Pixel-sorted events at BOS-sized flows mostly do -- the four events of a group sit on one or two source pixels
        // and move by a fraction of a pixel between their time stamps --, and every add saved is one fewer visit to a word that the
        // neighbouring lanes add to as well: config 2 at +-2 / +-0.5 px flows 28.5 / 29.7 -> 27.4 / 27.3 us per step, a solver
        // iteration on 2 M events crowded into a blob of sigma 100 / 50 px 70.6 / 76.4 -> 64.2 / 70.5 us (identical images: integer adds
        // commute).  Where nothing merges the run logic only costs -- at +-30 px flows 2.5 % of the built-halo loop, 13 % of the
        // run-time-window one (25.4 -> 29.1 us), on 2 M-event windows (2 events per pixel) 2.5 % --: it rides on PAIRS, i.e. on tiles
        // that hold >= 4 events per pixel behind a small window (tile_body chooses per work item at run time).
        const unsigned bsel = inside ? byte : 8u * kDummy;
        if (e > 0 && bsel != m_byte) {
          unsigned long long* wp = reinterpret_cast<unsigned long long*>(s_bytes + m_byte);
          atomicAdd(wp, ((unsigned long long)m01 << 32) | m00);
          atomicAdd(wp + PT / 2, ((unsigned long long)m11 << 32) | m10);
          m00 = m01 = m10 = m11 = 0u;
        }
        m_byte = bsel;
        m00 += q00, m01 += q01, m10 += q10, m11 += q11;
        if (e == 3) {
          unsigned long long* wp = reinterpret_cast<unsigned long long*>(s_bytes + m_byte);
          atomicAdd(wp, ((unsigned long long)m01 << 32) | m00);
          atomicAdd(wp + PT / 2, ((unsigned long long)m11 << 32) | m10);
        }
      } else {
        unsigned long long* w = reinterpret_cast<unsigned long long*>(s_bytes + (inside ? byte : 8u * kDummy));
        atomicAdd(w, ((unsigned long long)q01 << 32) | q00);
        atomicAdd(w + PT / 2, ((unsigned long long)q11 << 32) | q10);  // (next row of the same plane: + 4 PT bytes)
      }
    }
  };
  if (PAIRS) {
    // One loop iteration = the two groups of a lane's pair of macro chunk m, written out (no parity branch; the pipeline buffers
    // rotate once per TWO groups, which also halves the register moves per event): the pair of the next macro chunk was requested an
    // iteration ago; its first group is decoded after the first deposit, its second after the second group's gathers are on their way,
    // and then its registers take the loads of the macro chunk after it.  Same three stages as the plain loop: event loads two
    // groups ahead or more, flow gathers one group ahead.
    auto pair_group = [&](int m, int j) { return tr.g_first + m * (2 * kWave) + 2 * lane + j; };
    int m_cur = wave, m_next = wave + kWaves;  // (the plain loop's two pre-assigned chunks: macro chunks wave and wave + 16)
    CRaw raw0 = load_craw(pair_group(m_cur, 0), tr, ev), raw1 = load_craw(pair_group(m_cur, 1), tr, ev);
    CGroupQ ga, gb;  // groups 2 l and 2 l + 1 of the current macro chunk
    decode_craw(ga, raw0, hc4);
    decode_craw(gb, raw1, hc4);
    raw0 = load_craw(pair_group(m_next, 0), tr, ev);
    raw1 = load_craw(pair_group(m_next, 1), tr, ev);
    float fu[4], fv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fetch(ga.pr[e], ga.pcq[e], fu[e], fv[e]);
    while (tr.g_first + m_cur * (2 * kWave) <= g_last) {  // wave-uniform
      const int32_t g0 = pair_group(m_cur, 0);
      float gu[4], gv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) fetch(gb.pr[e], gb.pcq[e], gu[e], gv[e]);
      deposit(ga, fu, fv, g0 <= g_last);
      decode_craw(ga, raw0, hc4);  // group 2 l of the next macro chunk
#pragma unroll
      for (int e = 0; e < 4; ++e) fetch(ga.pr[e], ga.pcq[e], fu[e], fv[e]);
      CGroupQ gn;
      decode_craw(gn, raw1, hc4);  // group 2 l + 1 of the next macro chunk: its registers are free now
      const int m_new = queue.pull();
      raw0 = load_craw(pair_group(m_new, 0), tr, ev);
      raw1 = load_craw(pair_group(m_new, 1), tr, ev);
      deposit(gb, gu, gv, g0 + 1 <= g_last);
      gb = gn;
      m_cur = m_next;
      m_next = m_new;
    }
  } else {
  // chunk c covers groups [g_first + 64 c, g_first + 64 c + 64); this wave starts with chunks `wave` and `wave + 16`
  int c_cur = wave, c_nxt = wave + kWaves;
  CGroupQ cur, nxt;
  float4 w_cur = float4{}, w_nxt = float4{};
  auto weights_of = [&](int c) {  // (clamped like the group loads: a chunk past the slice is only ever prefetched)
    return load_weights4<FMT_COMPACT>(ev.w, max(min(chunk_group(tr.g_first, c, lane), tr.g_last), tr.g_first), tr);
  };
  if (pre != nullptr) {  // (persistent batched kernel: this window's first two chunks were requested during the previous window)
    decode_craw(cur, pre[0], hc4);
    decode_craw(nxt, pre[1], hc4);
  } else {
    load_cgroup_q(cur, chunk_group(tr.g_first, c_cur, lane), tr, ev, hc4);
    load_cgroup_q(nxt, chunk_group(tr.g_first, c_nxt, lane), tr, ev, hc4);
  }
  if (HAS_W) w_cur = weights_of(c_cur), w_nxt = weights_of(c_nxt);
  float fu[4], fv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) fetch(cur.pr[e], cur.pcq[e], fu[e], fv[e]);
  while (chunk_group(tr.g_first, c_cur, 0) <= g_last) {  // wave-uniform
    const int32_t grp = chunk_group(tr.g_first, c_cur, lane);
    float gu[4], gv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fetch(nxt.pr[e], nxt.pcq[e], gu[e], gv[e]);
    const int c_nn = queue.pull();
    CGroupQ nn;
    load_cgroup_q(nn, chunk_group(tr.g_first, c_nn, lane), tr, ev, hc4);
    float4 w_nn = float4{};
    if (HAS_W) w_nn = weights_of(c_nn);
    const bool lane_live = grp <= g_last;  // the last chunk of the slice may be partial
    deposit(cur, fu, fv, lane_live, w_cur);
    cur = nxt;
    nxt = nn;
    w_cur = w_nxt;
    w_nxt = w_nn;
    c_cur = c_nxt;
    c_nxt = c_nn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fu[e] = gu[e];
      fv[e] = gv[e];
    }
  }
  }
  if (any_spill) *any_spill = spilled;
  // Checksum: nothing is counted per event.  Every finite event whose taps are inside the window adds exactly 2^20 units, so the
  // kernel expects 2^20 x (events of the slice - events the spill pass took over); a non-finite event (NaN / Inf flow or time)
  // makes the sums disagree and the slice is redone by the exact f64 loop, which is correct for it as well.
  // (HAS_W: the units of the events inside the window, counted as they were added.)
  return added;
}

// PASS_MAIN: the lean hot loop -- every tap that lands inside the LDS window is accumulated, branch-free (dead or
//            out-of-window lanes add 0 to a dummy word); events whose taps leave the window only raise a flag.
// PASS_SPILL: the rare second sweep over the same slice that handles exactly those events with global atomics on
//            the spill image.  Keeping that code out of the hot loop halves its VALU count (it dragged ~90 64-bit
//            address operations and ~30 exec-mask branches per 4 events into a loop that is VALU-bound).
enum Pass { PASS_MAIN = 0, PASS_SPILL = 1 };

// UNIFORM: one translation theta for all events (2-DoF model, src/warp.py:364-383: x' = x + dt * theta, i.e. a
// dense flow of -theta everywhere) -- the two flow gathers disappear.
// FRAC: the compact plan carries fractional source coordinates (EvPtrs::cfx / cfy): every pass takes the general loop below, whose
// groups hold the fractions (load_group); the lean loop assumes integer pixels.
template <int TH, int TW, int HALO, bool HAS_W, int MODE, int PASS, int FMT, bool UNIFORM, bool GRID = false, bool DYN = false,
          bool PAIRS = false, bool FRAC = false>
__device__ __forceinline__ unsigned long long accumulate_slice(const TileRange& tr, double* s_acc, const EvPtrs& ev,
                                                     const float* __restrict__ flow, int H, int W, int pad_h, int pad_w,
                                                     float* spill, bool* any_spill, const ChunkQueue& queue,
                                                     const Win<TH, TW, HALO, DYN>& win, const CRaw* pre = nullptr,
                                                     float wscale = kFxScale) {
  const int LH = win.LH(), LW = win.LW(), PT = win.P();
  unsigned long long* s_fx = reinterpret_cast<unsigned long long*>(s_acc);
  const int h = H + 2 * pad_h, w = W + 2 * pad_w;
  const int tr0 = tr.ty * TH, tc0 = tr.tx * TW;
  const int oy = tr0 - win.HR(), ox = tc0 - win.HC();  // LDS cell (0,0) = un-padded pixel (oy, ox)
  const float* __restrict__ flow1 = UNIFORM ? flow : flow + (GRID ? (int64_t)TH * TW : (int64_t)H * W);
  const float uni_u = UNIFORM ? -flow[0] : 0.0f, uni_v = UNIFORM ? -flow[1] : 0.0f;  // flow == theta pair
  const int fl_r0 = GRID ? tr0 : 0, fl_c0 = GRID ? tc0 : 0, fl_w = GRID ? TW : W;  // GRID: tile-local flow in LDS
  unsigned long long added = 0;  // FX: integer total this thread put into LDS
  bool spilled = false;
  if (tr.g_first > tr.g_last) return 0;
  if (FMT == FMT_COMPACT && PASS == PASS_MAIN && !HAS_W && !FRAC)  // the lean hot loop (fixed point, or its exact f64 redo)
    return accumulate_compact_fx<TH, TW, HALO, UNIFORM, MODE, GRID, DYN, PAIRS>(tr, s_acc, ev, flow, H, W, any_spill, queue, win, pre);
  if constexpr (FMT == FMT_COMPACT && PASS == PASS_MAIN && HAS_W && !FRAC && MODE == ACC_FX && !DYN && !PAIRS)  // ... with per-event weights
    return accumulate_compact_fx<TH, TW, HALO, UNIFORM, ACC_FX, GRID, false, false, true>(tr, s_acc, ev, flow, H, W, any_spill, queue, win,
                                                                                       nullptr, wscale);
  // 3-stage software pipeline per lane:  16-byte SoA loads of group k+2 | flow gathers of group k+1 | LDS adds of
  // group k.  Everything is unconditional (clamped indices), so hipcc counts the queue and waits with vmcnt(N > 0).
  const int32_t g_last = tr.g_last;
  int32_t grp = tr.g_first + threadIdx.x;
  Group cur, nxt;
  load_group<FMT, HAS_W, TH, TW>(cur, grp, tr, ev, tr0, tc0);
  load_group<FMT, HAS_W, TH, TW>(nxt, grp + kBlock, tr, ev, tr0, tc0);
  float fu[4], fv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int lin = (cur.rs[e] - fl_r0) * fl_w + (cur.cs[e] - fl_c0);
    fu[e] = UNIFORM ? uni_u : flow[lin];
    fv[e] = UNIFORM ? uni_v : flow1[lin];
  }
  while (grp <= g_last) {
    float gu[4], gv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // gathers of the NEXT group (sorted events: broadcast / adjacent addresses)
      const int lin = (nxt.rs[e] - fl_r0) * fl_w + (nxt.cs[e] - fl_c0);
      gu[e] = UNIFORM ? uni_u : flow[lin];
      gv[e] = UNIFORM ? uni_v : flow1[lin];
    }
    Group nn;  // loads two groups ahead
    load_group<FMT, HAS_W, TH, TW>(nn, grp + 2 * kBlock, tr, ev, tr0, tc0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float wv = cur.w[e];  // 0 for dead slots
      const Taps f = warped_taps(cur.rs[e], cur.cs[e], cur.fx[e] - cur.dt[e] * fu[e], cur.fy[e] - cur.dt[e] * fv[e]);
      const int rl = f.R - oy, cl = f.C - ox;
      const bool inside = (unsigned)rl < (unsigned)(LH - 1) && (unsigned)cl < (unsigned)(LW - 1);  // f.ok false -> far outside
      // FX stores unsigned fields: the (at most 1e-6) negative fractions that floor(x + eps) can produce are clamped
      // in the fixed-point mode only (a relative change of <= 1e-6 of one event's weight)
      const float frq = MODE == ACC_FX ? fmaxf(f.fr, 0.0f) : f.fr, fcq = MODE == ACC_FX ? fmaxf(f.fc, 0.0f) : f.fc;
      const float a = 1.0f - frq, b = 1.0f - fcq;
      if (PASS == PASS_MAIN) {
        spilled |= (!inside) && (wv != 0.0f) && f.ok;
        const float ws = inside ? wv : 0.0f;
        if (MODE == ACC_FX) {
          // round-to-nearest by the magic-number trick: bits(x + 1.5 * 2^23) - bits(1.5 * 2^23) == rint(x) for
          // |x| < 2^22 -- one FMA + one integer subtract per tap instead of fma + floor + convert
          constexpr float kMagic = 12582912.0f;
          constexpr int kMagicBits = 0x4B400000;
          const float wsc = HAS_W ? wscale : kFxScale;  // (weights: normalised by the slice's max |w|, TileShared::wscale)
          const float as = a * (ws * wsc), fs = frq * (ws * wsc);
          const unsigned q00 = (unsigned)(__float_as_int(__fmaf_rn(as, b, kMagic)) - kMagicBits);
          const unsigned q10 = (unsigned)(__float_as_int(__fmaf_rn(fs, b, kMagic)) - kMagicBits);
          const unsigned q01 = (unsigned)(__float_as_int(__fmaf_rn(as, fcq, kMagic)) - kMagicBits);
          const unsigned q11 = (unsigned)(__float_as_int(__fmaf_rn(fs, fcq, kMagic)) - kMagicBits);
          const unsigned t = (unsigned)(rl * PT + cl);
          const int word = inside ? (int)((t >> 1) + (t & 1u) * (LH * PT / 2)) : LH * PT;  // LH*PT.. = dummy region
          atomicAdd(s_fx + word, ((unsigned long long)q01 << 32) | q00);
          atomicAdd(s_fx + word + PT / 2, ((unsigned long long)q11 << 32) | q10);
          added += q00 + q10 + q01 + q11;
        } else {
          const float as = a * ws, fs = frq * ws;
          const int cell = inside ? rl * PT + cl : LH * PT;
          const int dn = inside ? PT : 0;
          atomic_add(&s_acc[cell], (double)(as * b));
          atomic_add(&s_acc[cell + dn], (double)(fs * b));
          atomic_add(&s_acc[cell + 1], (double)(as * fcq));
          atomic_add(&s_acc[cell + dn + 1], (double)(fs * fcq));
        }
      } else if (!inside && wv != 0.0f && f.ok) {  // beyond the halo: spill image (zero-invariant scratch)
        if (FMT == FMT_COMPACT && !HAS_W && !FRAC) added += 1ull << kFxShift;  // (lean path: the checksum leaves these events out)
        const float w00 = a * b * wv, w10 = f.fr * b * wv, w01 = a * f.fc * wv, w11 = f.fr * f.fc * wv;
        const int R = f.R + pad_h, C = f.C + pad_w;
        const bool r0 = R >= 0 && R < h, r1 = R + 1 >= 0 && R + 1 < h;
        const bool c0 = C >= 0 && C < w, c1 = C + 1 >= 0 && C + 1 < w;
        const int gb = R * w + C;
        if (r0 && c0) atomic_add(&spill[gb], w00);
        if (r1 && c0) atomic_add(&spill[gb + w], w10);
        if (r0 && c1) atomic_add(&spill[gb + 1], w01);
        if (r1 && c1) atomic_add(&spill[gb + w + 1], w11);
      }
    }
    cur = nxt;
    nxt = nn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fu[e] = gu[e];
      fv[e] = gv[e];
    }
    grp += kBlock;
  }
  if (any_spill) *any_spill = spilled;
  return added;
}

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// signed fields of a paired word
__device__ __forceinline__ long long fx_lo(long long v) { return (long long)(int)(unsigned)(v & 0xffffffffll); }
__device__ __forceinline__ long long fx_hi(long long v) { return (v - fx_lo(v)) >> 32; }

// ---- run-time LDS windows (DYN kernels): a bound on the tile's displacements --------------------------------------------------
// max |u|, max |v| over the tile's own pixels of a dense flow [2, H, W]: unconditional clamped loads, all in flight at once (the
// same lines the loop's gathers are about to fetch: they warm the L2 for them).  NaN entries are ignored (their events are
// dropped by the loop), Inf gives Inf (-> the largest window).  Valid per thread; reduce with tile_bound_reduce.
template <int TH, int TW>
__device__ __forceinline__ void tile_flow_absmax(const float* __restrict__ flow, int H, int W, int tr0, int tc0, float& mu, float& mv) {
  constexpr int kIt = (TH * TW + kBlock - 1) / kBlock;
  float u[kIt], v[kIt];
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int i = min((int)threadIdx.x + k * kBlock, TH * TW - 1);
    const int rl = i / TW, cl = i - rl * TW;
    const int64_t o = (int64_t)min(tr0 + rl, H - 1) * W + min(tc0 + cl, W - 1);
    u[k] = flow[o];
    v[k] = flow[(int64_t)H * W + o];
  }
  mu = mv = 0.0f;
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    mu = fmaxf(mu, fabsf(u[k]));
    mv = fmaxf(mv, fabsf(v[k]));
  }
}
// per-wave maxima -> s_red [2 * waves]; the caller's next barrier publishes them and tile_bound_read folds them (every thread,
// broadcast reads: no barrier of its own)
__device__ __forceinline__ void tile_bound_post(float mu, float mv, float* s_red) {
  mu = wave_max_nonneg(mu);  // (|.| values; Inf stays Inf, NaN entries were skipped by fmaxf)
  mv = wave_max_nonneg(mv);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    s_red[threadIdx.x / kWave] = mu;
    s_red[kBlock / kWave + threadIdx.x / kWave] = mv;
  }
}
template <int TH, int TW, int HALO, bool DYN>
__device__ __forceinline__ Win<TH, TW, HALO, DYN> tile_bound_read(const float* s_red, float dt_bound) {
  Win<TH, TW, HALO, DYN> w{HALO, HALO};
  if (DYN) {
    float mu = 0.0f, mv = 0.0f;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; ++k) {
      mu = fmaxf(mu, s_red[k]);
      mv = fmaxf(mv, s_red[kBlock / kWave + k]);
    }
    w.hr = halo_for(mu * dt_bound, 1, HALO);
    w.hc = halo_for(mv * dt_bound, 4, HALO);
  }
  return w;
}

// cells 4 j .. 4 j + 3 of row r of a workgroup's LDS image as the decode pass of tile_body stores them to its slab: fixed-point planes
// A / B (row pitch PT cells, LH rows) or, after an exact redo (f64 = true), doubles -- the same conversions, the same bits
__device__ __forceinline__ float4 lds_image_cells4(const double* s_acc, int LH, int PT, int r, int j, bool f64) {
  if (f64) {
    const double* p = s_acc + r * PT + 4 * j;
    return make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
  }
  constexpr float kInv = (float)kFxInv;
  const uint2* pa = reinterpret_cast<const uint2*>(s_acc);
  const uint2* pb = pa + LH * PT / 2;
  const int wrow = r * (PT / 2);
  const uint2 a0 = pa[wrow + 2 * j], a1 = pa[wrow + 2 * j + 1];
  const uint2 b0 = pb[wrow + 2 * j], b1 = pb[wrow + 2 * j + 1];
  const unsigned bmh = j > 0 ? pb[wrow + 2 * j - 1].y : 0u;
  return make_float4(((float)a0.x + (float)bmh) * kInv, ((float)a0.y + (float)b0.x) * kInv, ((float)a1.x + (float)b0.y) * kInv,
                     ((float)a1.y + (float)b1.x) * kInv);
}

// workgroup state of the accumulate pass
struct TileShared {
  unsigned long long chk;  // sum over the workgroup of (units added - units decoded), modulo 2^64
  int flag[2];             // [0] fixed-point overflow, [1] some event left the LDS window
  unsigned next;           // chunk queue of the lean loop
  float bound[2 * kBlock / kWave];  // DYN: per-wave maxima of |u|, |v| over the tile
  // per-event weights in fixed point (HAS_W, ACC_FX): a weight enters as w / max |w| of the work item's slice in the 2^20 unit
  // (wscale = 2^20 / max |w|) and the decode pass gives the factor back (wdec = max |w| / 2^20): 2 ds_add_u64 per event instead of
  // 4 ds_add_f64.  A slice with a negative (or non-finite) weight runs in f64 from the start (accumulate_tile).
  float wscale, wdec;
};
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// what tile_body's decode pass shows every quad of floats it stores to the slab: nothing, or (resident solver kernel) the sum of
// the cells that lie inside the image's valid region -- the workgroup's share of sum(IWE), exact in a double (kCombineExactSum)
struct NoOwn {
  __device__ __forceinline__ void reset() {}
  __device__ __forceinline__ void operator()(int, int, const float4&) {}
};
struct OwnSum {
  int R0, C0;   // image row / column of the window's cell (0, 0)
  int lo, h, w; // valid region [lo, h - lo) x [lo, w - lo)
  double acc;
  __device__ __forceinline__ void reset() { acc = 0.0; }
  __device__ __forceinline__ void operator()(int r, int j, const float4& v) {
    const int R = R0 + r, C = C0 + 4 * j;
    if (R < lo || R >= h - lo) return;
    if (C >= lo && C + 3 < w - lo) {
      acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    } else {
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (C + k >= lo && C + k < w - lo) acc += (double)e[k];
    }
  }
};

// ... or (bk.k0 != 0) for the contrast of the 3-tap blurred image (blur3.h), the tile's share of sum(m . B x) = sum_p wgt(p) x(p), wgt = B^T m:
// interior pixels all carry the same weight (summed as above, scaled once at the end), the few rows / columns next to the image's
// border their own
struct OwnSumBlur {
  int R0, C0;
  int lo, h, w;
  Blur3 bk;
  double acc, acc_b;  // interior pixels (unit weight so far), border pixels (weighted)
  __device__ __forceinline__ void reset() { acc = 0.0, acc_b = 0.0; }
  __device__ __forceinline__ void operator()(int r, int j, const float4& v) {
    const int R = R0 + r, C = C0 + 4 * j;
    if (bk.k0 == 0.0f) {  // (uniform) no blur: OwnSum's exact sum over the valid region
      if (R < lo || R >= h - lo) return;
      if (C >= lo && C + 3 < w - lo) {
        acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      } else {
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (C + k >= lo && C + k < w - lo) acc += (double)e[k];
      }
      return;
    }
    if (R < 0 || R >= h) return;
    if (R >= lo + 2 && R < h - lo - 2 && C >= lo + 2 && C + 3 < w - lo - 2) {
      acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      return;
    }
    const float wr = blur3_axis_weight(R, h, lo, bk);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (C + k >= 0 && C + k < w) acc_b += (double)(wr * blur3_axis_weight(C + k, w, lo, bk)) * (double)e[k];
  }
  // the weight of an interior pixel: three valid outputs per axis read it
  __device__ __forceinline__ double total() const {
    if (bk.k0 == 0.0f) return acc;
    const float wi = blur3_axis_weight(lo + 2, max(h, 2 * lo + 5), lo, bk);
    return (double)(wi * wi) * acc + acc_b;
  }
};

// Everything of one work item after its set-up barrier: the event loop, the rare spill sweep, the decode pass that writes the slab
// and checks the fixed-point sums (f64 redo if a field wrapped).
//   DYN         the LDS window `win` was chosen at run time; halo_tab [tiles] tells the combine pass
//   ZERO        the decode pass zeroes every LDS word it reads: the NEXT window of a persistent workgroup starts on a clean image
//   pre         the slice's first two chunks, already loaded (persistent kernel), or nullptr
//   after_loop  called by every wave as it leaves the event loop, before the barrier: the persistent kernel requests the next
//               window's first chunks there, so that they arrive while this window's image is decoded and stored
//   own         sees every quad the decode pass stores (NoOwn / OwnSum)
//   FRAC        the compact plan carries fractional source coordinates (EvPtrs::cfx): the general loop instead of the lean one
template <int TH, int TW, int HALO, bool HAS_W, int MODE, int FMT, bool UNIFORM, bool GRID, bool DYN, bool ZERO, bool FRAC = false,
          typename Hook, typename Own>
__device__ __forceinline__ void tile_body(const TileRange& tr, const Win<TH, TW, HALO, DYN>& win, const float* flow, double* s_acc,
                                          TileShared& sh, const EvPtrs& ev, int H, int W, int tiles_x, int pad_h, int pad_w,
                                          float* __restrict__ slabs, float* spill, unsigned* __restrict__ spill_epoch, unsigned epoch,
                                          unsigned* __restrict__ halo_tab, const CRaw* pre, Hook&& after_loop, Own& own) {
  constexpr int kLHmax = TH + 2 * HALO, kLWmax = TW + 2 * HALO;
  constexpr int kCells = acc_cells<TH, TW, HALO, DYN>();
  const ChunkQueue queue{&sh.next};
  const int LH = win.LH(), LW = win.LW(), PT = win.P();  // (PT: row pitch in LDS; slabs are dense, LW floats per row)
  if (DYN && halo_tab != nullptr && threadIdx.x == 0) halo_tab[tr.ty * tiles_x + tr.tx] = win_pack(win.hr, win.hc);  // (every part of a tile: same word; nullptr: the resident kernel hands windows over itself)
  EBOS_STAMP(1);

  bool spilled = false;
  unsigned long long added;
  // run-time-window kernels: where a source pixel holds several events (>= 4 on average over the tile), neighbouring lanes are
  // given groups two apart (PAIRS), so that the lanes of one wave instruction hold other pixels' events (accumulate_compact_fx)
  constexpr bool kCanPair = DYN && FMT == FMT_COMPACT && !HAS_W && MODE == ACC_FX && !FRAC;
  // (... and where the window is small: at 30 px flows the plain mapping has few address conflicts to begin with and is 1 us faster)
  // (... or where the pixels are so full -- >= 24 events each: the 50 M-event window of the 2-DoF sweep holds 54 -- that even a 30 px
  // displacement leaves several events of a pixel on one word: 33.9 -> 33.1 ms per 512-hypothesis sweep)
  const int32_t n_ev = tr.end - tr.beg;
  if (kCanPair && n_ev >= 4 * TH * TW && ((win.HR() <= 16 && win.HC() <= 16) || n_ev >= 24 * TH * TW))
    added = accumulate_slice<TH, TW, HALO, HAS_W, MODE, PASS_MAIN, FMT, UNIFORM, GRID, DYN, kCanPair>(
        tr, s_acc, ev, flow, H, W, pad_h, pad_w, spill, &spilled, queue, win, nullptr);
  else
    added = accumulate_slice<TH, TW, HALO, HAS_W, MODE, PASS_MAIN, FMT, UNIFORM, GRID, DYN, false, FRAC>(
        tr, s_acc, ev, flow, H, W, pad_h, pad_w, spill, &spilled, queue, win, pre, HAS_W ? sh.wscale : kFxScale);
  after_loop();
  constexpr bool kLeanLoop = FMT == FMT_COMPACT && !HAS_W && !FRAC;  // accumulate_compact_fx: counts nothing per event
  if (kLeanLoop && threadIdx.x == 0 && tr.g_first <= tr.g_last)  // 2^20 units per event of the slice (padding slots excluded)
    added += (unsigned long long)(min(tr.end, tr.beg + 4 * (tr.g_last - tr.g_first + 1)) - tr.beg) << kFxShift;
  if (spilled) sh.flag[1] = 1;  // benign race: every writer stores 1
  EBOS_STAMP(2);
  __syncthreads();
  EBOS_STAMP(3);
  // SpillEpoch: a workgroup that puts anything into the spill image stamps the workspace with this call's number; the combine
  // pass reads the 3.7 MB spill image only if the stamp is this call's (it is all zero otherwise, and stays so)
  if (sh.flag[1] && spill_epoch != nullptr && threadIdx.x == 0) *spill_epoch = epoch;  // benign race: every writer stores the same value
  if (sh.flag[1] && spill != nullptr)  // rare: taps beyond the halo go to the spill image with global atomics (lean path: minus their units;
                                       // no spill image: the resident kernel ends its launch on a spill instead)
    added -= accumulate_slice<TH, TW, HALO, HAS_W, MODE, PASS_SPILL, FMT, UNIFORM, GRID, DYN, false, FRAC>(tr, s_acc, ev, flow, H, W, pad_h,
                                                                                                           pad_w, spill, nullptr, queue, win);

  // (the slab stride is the LARGEST window's: a run-time window fills the first LH * LW floats of its slab)
  float4* out = reinterpret_cast<float4*>(slabs + (int64_t)tr.slab * (kLHmax * kLWmax));
#ifndef EBOS_PLAIN_SLABS
  const __amdgpu_buffer_rsrc_t out_rsrc = slab_rsrc(reinterpret_cast<const float*>(out), (unsigned)(kLHmax * kLWmax * sizeof(float)));
#define EBOS_SLAB_STORE(i, v) slab_store4(out_rsrc, (unsigned)(i) * 16u, (v))
#else
#define EBOS_SLAB_STORE(i, v) out[i] = (v)
#endif
  bool f64_flush = (MODE == ACC_F64);
  if (MODE == ACC_FX) {
    // One pass: decode 4 consecutive cells (c0 % 4 == 0) of a row from planes A and B, write them to the slab
    // optimistically, and sum the decoded fields for the overflow check
    //   sum(decoded fields) == sum(added units), both in 64-bit arithmetic.  A low field that wraps loses 2^32 and
    //   carries 1 into its high neighbour (the sum drops by 2^32 - 1); a high field that wraps carries out of the word
    //   (the sum drops by 2^32) -- so the sums must be wider than 32 bits: modulo 2^32 the second case is invisible
    //   (it was, until a 40 000-event hot pixel showed it).  Both changes are negative: they cannot cancel.
    // Fields are unsigned 32-bit: lo = low dword (column c), hi = high dword (column c + 1).  The two fields that make
    // up one pixel (plane A + plane B) are added as floats: their integer sum could pass 2^32 although neither did.
    uint2* pa = reinterpret_cast<uint2*>(s_acc);
    uint2* pb = pa + LH * PT / 2;
    unsigned long long decoded = 0;
    const float kInv = HAS_W ? sh.wdec : (float)kFxInv;  // (a compile-time constant without weights)
    // cells 4j..4j+3 of row r <- A words 2j, 2j+1 and B words 2j-1, 2j, 2j+1 (pair (4j-1, 4j): its lo field belongs to the previous quad)
    auto decode_quad = [&](int r, int j, int slab_quad) {
      const int wrow = r * (PT / 2);
      const uint2 a0 = pa[wrow + 2 * j], a1 = pa[wrow + 2 * j + 1];
      const uint2 b0 = pb[wrow + 2 * j], b1 = pb[wrow + 2 * j + 1];
      const unsigned bmh = j > 0 ? pb[wrow + 2 * j - 1].y : 0u;
      if (ZERO) {
        const uint2 z = make_uint2(0u, 0u);
        pa[wrow + 2 * j] = z, pa[wrow + 2 * j + 1] = z;
        pb[wrow + 2 * j] = z, pb[wrow + 2 * j + 1] = z;
      }
      decoded += ((unsigned long long)a0.x + a0.y) + ((unsigned long long)a1.x + a1.y) + ((unsigned long long)b0.x + b0.y) +
                 ((unsigned long long)b1.x + b1.y);
      const float4 v4 = make_float4(((float)a0.x + (float)bmh) * kInv, ((float)a0.y + (float)b0.x) * kInv,
                                    ((float)a1.x + (float)b0.y) * kInv, ((float)a1.y + (float)b1.x) * kInv);
      EBOS_SLAB_STORE(slab_quad, v4);
      own(r, j, v4);
    };
    if (ZERO) {
      // Rows are dealt to WAVES (64 / q rows per wave and step, q = LW / 4 quads per row): a word that two neighbouring quads read
      // (the pair that straddles them) is read by two lanes of ONE wave -- in program order before either lane's zeroing store --
      // so the decode pass can zero what it reads without a second pass over the image.
      const int q = LW / 4, rpi = kWave / q;  // (q <= 64: LW <= 256)
      const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
      int sub = 0, j = lane;
      while (j >= q) j -= q, ++sub;  // (<= 3 steps: q >= 16 for the built tiles)
      const bool act = sub < rpi;
      for (int r0 = wave * rpi; r0 < LH; r0 += (kBlock / kWave) * rpi) {
        const int r = r0 + sub;
        if (act && r < LH) decode_quad(r, j, r * q + j);
      }
      // the dummy region took the (garbage) adds of out-of-window lanes; a later window may lay real cells over it
      for (int i = threadIdx.x; i < PT / 2 + 2; i += kBlock) reinterpret_cast<unsigned long long*>(s_acc)[LH * PT + i] = 0ull;
    } else {
      const int q = LW / 4;
      const float inv_q = 1.0f / (float)q;  // (run-time width: row = floor((i + 0.5) / q) exactly for i < 2^16)
      for (int i = threadIdx.x; i < LH * q; i += kBlock) {
        const int r = DYN ? (int)(((float)i + 0.5f) * inv_q) : i / q;
        decode_quad(r, i - r * q, i);
      }
    }
    // sum(added) == sum(decoded) over the workgroup  <=>  sum(added - decoded) == 0 modulo 2^64: one value per lane, one DPP wave
    // sum, one LDS atomic per wave, one barrier (two values, shuffles, a serial 32-term loop and two barriers before)
#ifdef EBOS_ABL_MULTIK
    const unsigned long long diff = 0ull * wave_sum(added - decoded);  // (ablation: K x the units, no redo)
#else
    const unsigned long long diff = wave_sum(added - decoded);
#endif
    if ((threadIdx.x & (kWave - 1)) == 0 && diff != 0ull) atomicAdd(&sh.chk, diff);
    __syncthreads();
    if (sh.chk != 0ull) {  // a field wrapped: redo this slice exactly in f64 and overwrite the slab (spill taps already issued)
      for (int i = threadIdx.x; i < kCells; i += kBlock) s_acc[i] = 0.0;
      if (threadIdx.x == 0) sh.next = 2 * (kBlock / kWave);  // the redo draws its chunks afresh
      __syncthreads();
      accumulate_slice<TH, TW, HALO, HAS_W, ACC_F64, PASS_MAIN, FMT, UNIFORM, GRID, DYN, false, FRAC>(tr, s_acc, ev, flow, H, W, pad_h, pad_w,
                                                                                                      spill, nullptr, queue, win);
      __syncthreads();
      f64_flush = true;
    }
  }
  if (f64_flush) {
    // slab = the LDS image as f32, 16 B per lane, fully coalesced plain stores
    const int q = LW / 4;
    const float inv_q = 1.0f / (float)q;
    own.reset();  // (what the optimistic fixed-point decode showed is void)
    for (int i = threadIdx.x; i < LH * q; i += kBlock) {
      const int r = DYN ? (int)(((float)i + 0.5f) * inv_q) : i / q;
      double* p = &s_acc[r * PT + 4 * (i - r * q)];
      const float4 v4 = make_float4((float)p[0], (float)p[1], (float)p[2], (float)p[3]);
      EBOS_SLAB_STORE(i, v4);
      own(r, i - r * q, v4);
      if (ZERO) p[0] = p[1] = p[2] = p[3] = 0.0;
    }
    if (ZERO)
      for (int i = threadIdx.x; i < PT / 2 + 2; i += kBlock) s_acc[LH * PT + i] = 0.0;
  }
  EBOS_STAMP(4);
}


// set-up of one work item: LDS clear (first window of a workgroup), tile range, GRID: the tile's flow into LDS, DYN: the window
// from a bound on the tile's displacements.  Returns false for an unused work item of an adaptive plan (nothing to do).
// FRAC: a compact plan that carries the fractions of undistorted events (EvPtrs::cfx / cfy): the general loop on the compact slots
template <int TH, int TW, int HALO, bool HAS_W, int MODE, int FMT, bool UNIFORM, bool GRID = false, bool DYN = false, bool ZERO = false,
          bool FRAC = false>
__device__ __forceinline__ void accumulate_tile(const EvPtrs& ev, const int32_t* __restrict__ key_offsets,
                                                const float* __restrict__ flow_arg, int H, int W, int tiles_x, int splits, int pad_h,
                                                int pad_w, float* __restrict__ slabs, float* spill, const GridSrc& gs,
                                                unsigned* __restrict__ spill_epoch, unsigned epoch, float dt_bound = 0.0f,
                                                unsigned* __restrict__ halo_tab = nullptr) {
  constexpr int kLWmax = TW + 2 * HALO;
  constexpr int kCells = acc_cells<TH, TW, HALO, DYN>();
  static_assert(kLWmax % 4 == 0, "slab rows are written 4 cells at a time");
  static_assert(!DYN || (FMT == FMT_COMPACT && !HAS_W), "run-time windows: the lean loop only");
  extern __shared__ double s_acc[];  // [LH][LW] doubles, or 2 planes of [LH][LW/2] paired words; + dummy
  __shared__ TileShared sh;
  EBOS_STAMP(0);
  static_assert(kCells % 2 == 0, "LDS image is cleared 16 bytes per lane");
  // (dense field: the clear comes first -- it needs nothing, and the dependent loads of tile_range fly over it; run-time
  // windows: the tile's flow values, which bound its displacements, are requested before the clear and awaited after it)
  constexpr bool kBoundFromFlow = DYN && !GRID && !UNIFORM;
  if (!GRID && !kBoundFromFlow)
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock)  // all-zero bits = 0 in both modes
      reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  const TileRange tr = tile_range<FMT>(key_offsets, ev, TH * TW, tiles_x, splits);
  if (tr.ty < 0) return;  // unused work item of an adaptive plan: its slab is never read
  // An EMPTY tile (the static background of a schlieren recording: most tiles of a crowded window) owes the combine pass a slab of
  // zeros under the smallest window and nothing else: no LDS image, no tile flow (2.2 us), no event loop, no checksum -- 5 us of a
  // work item that a second round of work items (one workgroup per CU) used to pay in full.
  if (tr.tile_groups == 0 && !ZERO) {
    const Win<TH, TW, HALO, DYN> win{DYN ? halo_for(0.0f, 1, HALO) : HALO, DYN ? halo_for(0.0f, 4, HALO) : HALO};
    if (DYN && halo_tab != nullptr && threadIdx.x == 0) halo_tab[tr.ty * tiles_x + tr.tx] = win_pack(win.hr, win.hc);
    float4* out0 = reinterpret_cast<float4*>(slabs + (int64_t)tr.slab * ((TH + 2 * HALO) * kLWmax));
    const int n4 = win.LH() * win.LW() / 4;
#ifndef EBOS_PLAIN_SLABS
    const __amdgpu_buffer_rsrc_t zr = slab_rsrc(reinterpret_cast<const float*>(out0), (unsigned)((TH + 2 * HALO) * kLWmax * sizeof(float)));
    for (int i = threadIdx.x; i < n4; i += kBlock) slab_store4(zr, (unsigned)i * 16u, make_float4(0.f, 0.f, 0.f, 0.f));
#else
    for (int i = threadIdx.x; i < n4; i += kBlock) out0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    return;
  }
  // the lean loop's first two chunks per wave, requested here: they arrive under the rest of the set-up and its barrier instead of
  // a round trip after it (the persistent batched kernel requests them a whole window ahead)
  constexpr bool kLeanPre = FMT == FMT_COMPACT && !HAS_W && MODE == ACC_FX && !FRAC;
  CRaw pre[2];
  if (kLeanPre) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    pre[0] = load_craw(tr.g_first + wave * kWave + lane, tr, ev);
    pre[1] = load_craw(tr.g_first + (wave + kBlock / kWave) * kWave + lane, tr, ev);
  }
  float mu = 0.0f, mv = 0.0f;
  if (kBoundFromFlow) {
    tile_flow_absmax<TH, TW>(flow_arg, H, W, tr.ty * TH, tr.tx * TW, mu, mv);
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  }

  const float* flow = flow_arg;
  float* s_flow = reinterpret_cast<float*>(s_acc + kCells);  // GRID: the tile's dense flow, behind the accumulators
  Lerp* s_lerp = reinterpret_cast<Lerp*>(s_flow + 2 * TH * TW);
  TileGrid tg{};
  if (GRID) {
    tg = tile_grid_begin<TH, TW, 0>(flow_arg, gs, tr.ty * TH, tr.tx * TW, H, W, s_lerp);  // (its cell load flies over the clear)
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  }
  if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    sh.next = 2 * (kBlock / kWave);
    sh.chk = 0ull;
    sh.wscale = kFxScale;
    sh.wdec = (float)kFxInv;
  }
  if (HAS_W && MODE == ACC_FX) {  // max |w| over this work item's events: the unit its weights are quantised in (TileShared)
    float am = 0.0f;
    bool odd = false;  // a negative, NaN or Inf weight: the unsigned fields cannot hold it -- the slice takes the exact f64 pass
    for (int32_t g = tr.g_first + threadIdx.x; g <= tr.g_last; g += kBlock) {
      const float4 Wv = load_weights4<FMT>(ev.w, g, tr);
      const float ww[4] = {Wv.x, Wv.y, Wv.z, Wv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int32_t i = group_plan_index<FMT>(g, tr) + e;
        if (i >= tr.beg && i < tr.end) {
          am = fmaxf(am, fabsf(ww[e]));
          odd |= !(ww[e] >= 0.0f && ww[e] < 3.0e38f);
        }
      }
    }
    am = wave_max_nonneg(am);
    const bool any_odd = __builtin_amdgcn_ballot_w64(odd) != 0ull;
    __shared__ float s_wmax[kBlock / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) s_wmax[threadIdx.x / kWave] = any_odd ? INFINITY : am;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = 0.0f;
      for (int k = 0; k < kBlock / kWave; ++k) m = fmaxf(m, s_wmax[k]);
      if (m > 0.0f && m < 3.0e38f) {
        sh.wscale = kFxScale / m;
        sh.wdec = m * (float)kFxInv;
      } else if (!(m < 3.0e38f)) {
        // (a single negative weight wraps nothing the checksum could see -- its field just reads as a huge positive number: the
        // slice's verdict is set here, and the work item runs in f64 from the start, below)
        sh.flag[0] = 1;
      }
    }
  }
  if (DYN) {  // a bound on this tile's displacements: |flow| over the tile (dense), the cells its pixels interpolate (GRID), theta
    if (UNIFORM) {
      mu = fabsf(flow_arg[0]), mv = fabsf(flow_arg[1]);
    } else if (GRID) {  // (bilinear interpolation never leaves the range of its cells; every thread holds a cell of the block)
      const int idx = min((int)threadIdx.x, 2 * tg.ni * tg.nj - 1);
      const bool second = idx >= tg.ni * tg.nj;
      mu = second ? 0.0f : fabsf(tg.cell), mv = second ? fabsf(tg.cell) : 0.0f;
    }
    tile_bound_post(mu, mv, sh.bound);
  }
  if (GRID) {
    tile_grid_finish<TH, TW, 0>(tg, s_flow, s_lerp, reinterpret_cast<float*>(s_lerp + TH + TW));
    flow = s_flow;
  }
  __syncthreads();
  const Win<TH, TW, HALO, DYN> win = tile_bound_read<TH, TW, HALO, DYN>(sh.bound, dt_bound);
  NoOwn no_own;
  if constexpr (HAS_W && MODE == ACC_FX) {
    if (sh.flag[0]) {  // (uniform) a negative / non-finite weight in this slice: doubles, as before weights had a fixed-point form
      __syncthreads();
      if (threadIdx.x == 0) sh.flag[0] = 0;
      __syncthreads();
      tile_body<TH, TW, HALO, HAS_W, ACC_F64, FMT, UNIFORM, GRID, DYN, ZERO>(tr, win, flow, s_acc, sh, ev, H, W, tiles_x, pad_h, pad_w, slabs,
                                                                            spill, spill_epoch, epoch, halo_tab, nullptr, NoHook{}, no_own);
      return;
    }
  }
  tile_body<TH, TW, HALO, HAS_W, MODE, FMT, UNIFORM, GRID, DYN, ZERO, FRAC>(tr, win, flow, s_acc, sh, ev, H, W, tiles_x, pad_h, pad_w, slabs,
                                                                           spill, spill_epoch, epoch, halo_tab, kLeanPre ? pre : nullptr,
                                                                           NoHook{}, no_own);
}

template <int TH, int TW, int HALO, bool HAS_W, int MODE, int FMT, bool UNIFORM, bool GRID = false, bool DYN = false, bool FRAC = false>
__global__ void __launch_bounds__(kBlock)
iwe_slab_accumulate_kernel(EvPtrs ev, const int32_t* __restrict__ key_offsets, const float* __restrict__ flow_arg, int H, int W,
                           int tiles_x, int splits, int pad_h, int pad_w, float* __restrict__ slabs, float* spill, GridSrc gs,
                           unsigned* __restrict__ spill_epoch, unsigned epoch, float dt_bound, unsigned* __restrict__ halo_tab) {
  accumulate_tile<TH, TW, HALO, HAS_W, MODE, FMT, UNIFORM, GRID, DYN, false, FRAC>(ev, key_offsets, flow_arg, H, W, tiles_x, splits, pad_h,
                                                                                  pad_w, slabs, spill, gs, spill_epoch, epoch, dt_bound,
                                                                                  halo_tab);
}

// ---- several independent windows of one geometry in ONE launch (ebos_iwe_slab_batch_f32) ---------------------------------
// Thin windows (BASELINE configs[3]: 2 M events) are bound by per-workgroup fixed work, not by their events: of 10.7 us a
// (tile, window) workgroup spent 1.6 clearing its LDS image, 2.2 evaluating the tile's flow, 2.4 in the event loop, 1.5 at the
// barrier and 3.0 decoding and storing its slab (in-kernel stamps).  The batched accumulate pass is therefore
// PERSISTENT: workgroup b takes work item b of EVERY window of the batch in turn.  The LDS image is cleared once; afterwards the
// decode pass zeroes each word as it reads it (ZERO), and the write-through slab stores of window k drain while the workgroup is
// already setting up and looping over window k + 1 (nothing waits for them until the kernel ends).
// The windows' pointers travel by value in the kernel arguments.
struct FwdWindow {
  EvPtrs ev;
  const int32_t* key_offsets;
  const float* flow;     // [2, H, W] or the patch grid [2, gh, gw]
  float* slabs;          // this window's workspace sections
  float* spill;
  unsigned* spill_epoch;
  unsigned* halo_tab;
  double* partials;
  float* iwe;
  float* out_var;        // nullable
  double* moments;       // nullable
};
constexpr int kMaxBatch = 16;
struct FwdBatch {
  FwdWindow w[kMaxBatch];
};
static_assert(sizeof(FwdBatch) <= 3072, "the batch travels in the kernel argument segment");

// UNIFORM: the "windows" are HYPOTHESES of one plan -- w.flow is a translation (theta0, theta1) of the 2-DoF model
// (src/warp.py:364-383) -- as in the sweep of src/solver/generative_max_likelihood.py:229-255: the workgroup keeps its tile and walks
// the hypotheses (ebos_iwe_2dof_slab_batch_f32).
template <int TH, int TW, int HALO, bool GRID, bool DYN, bool UNIFORM = false>
__global__ void __launch_bounds__(kBlock)
iwe_slab_accumulate_batch_kernel(FwdBatch b, int n, int H, int W, int tiles_x, int splits, int pad_h, int pad_w, GridSrc gs,
                                 unsigned epoch, float dt_bound) {
  static_assert(!(UNIFORM && GRID), "a translation is not a patch grid");
  // A software pipeline over the batch's windows.  What a window's work item needs before its event loop can start -- its
  // tile range (dependent scalar loads), GRID: its block of grid cells, its first two chunks of events -- used to be three exposed
  // memory round trips per window (3.2 us of set-up and ~1 us at the head of the loop, of 11.9 us: in-kernel stamps).  They are
  // now requested one window ahead: the tile range before the current window's loop, chunks and cells by every wave as it
  // leaves that loop, so that they arrive while the current image is decoded, zeroed and stored.
  constexpr int kCells = acc_cells<TH, TW, HALO, DYN>();
  extern __shared__ double s_acc[];
  __shared__ TileShared sh;
  float* s_flow = reinterpret_cast<float*>(s_acc + kCells);  // GRID: the tile's dense flow, behind the accumulators
  Lerp* s_lerp = reinterpret_cast<Lerp*>(s_flow + 2 * TH * TW);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  EBOS_STAMP(5);
  // the one clear of the LDS image: afterwards the decode pass zeroes what it reads
  for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  TileRange tr = tile_range<FMT_COMPACT>(b.w[0].key_offsets, b.w[0].ev, TH * TW, tiles_x, splits);
  CRaw pre[2];
  TileGrid tg{};
  int lerp_tile = -1;  // GRID: the tile whose row / column interpolation s_lerp holds
  auto request = [&](const FwdWindow& w, const TileRange& t) {  // a window's first two chunks (+ GRID, same tile: its cell block)
    pre[0] = load_craw(t.g_first + wave * kWave + lane, t, w.ev);
    pre[1] = load_craw(t.g_first + (wave + kBlock / kWave) * kWave + lane, t, w.ev);
    if (GRID && t.ty * tiles_x + t.tx == lerp_tile) {
      const int idx = min((int)threadIdx.x, 2 * tg.ni * tg.nj - 1);
      const int ch = idx / (tg.ni * tg.nj), rem = idx - ch * (tg.ni * tg.nj);
      const int i = rem / tg.nj, j = rem - i * tg.nj;
      tg.cell = w.flow[((int64_t)ch * gs.ay.g + tg.gi0 + i) * gs.ax.g + tg.gj0 + j];
    }
  };
  if (tr.ty >= 0) {
    if (GRID) {
      tg = tile_grid_begin<TH, TW, 0>(b.w[0].flow, gs, tr.ty * TH, tr.tx * TW, H, W, s_lerp);
      lerp_tile = tr.ty * tiles_x + tr.tx;
    }
    request(b.w[0], tr);
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const FwdWindow& w = b.w[k];
    const bool more = k + 1 < n;
    TileRange trn = tr;
    if (more) trn = tile_range<FMT_COMPACT>(b.w[k + 1].key_offsets, b.w[k + 1].ev, TH * TW, tiles_x, splits);  // (scalar loads: in flight)
    const bool next_live = more && trn.ty >= 0;
    const FwdWindow& wn = b.w[more ? k + 1 : k];
    EBOS_STAMP(0);
    if (tr.ty >= 0) {
      if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
      if (threadIdx.x == 0) {
        sh.next = 2 * (kBlock / kWave);
        sh.chk = 0ull;
      }
      const float* flow = w.flow;
      if (GRID && tr.ty * tiles_x + tr.tx != lerp_tile) {  // (adaptive plans: this window's item is another tile) no prefetch
        tg = tile_grid_begin<TH, TW, 0>(w.flow, gs, tr.ty * TH, tr.tx * TW, H, W, s_lerp);
        lerp_tile = tr.ty * tiles_x + tr.tx;
      }
      if (DYN) {
        float mu = 0.0f, mv = 0.0f;
        if (UNIFORM) {
          mu = fabsf(w.flow[0]), mv = fabsf(w.flow[1]);
        } else if (GRID) {
          const int idx = min((int)threadIdx.x, 2 * tg.ni * tg.nj - 1);
          const bool second = idx >= tg.ni * tg.nj;
          mu = second ? 0.0f : fabsf(tg.cell), mv = second ? fabsf(tg.cell) : 0.0f;
        } else {
          tile_flow_absmax<TH, TW>(w.flow, H, W, tr.ty * TH, tr.tx * TW, mu, mv);
        }
        tile_bound_post(mu, mv, sh.bound);
      }
      if (GRID) {
        tile_grid_finish<TH, TW, 0>(tg, s_flow, s_lerp, reinterpret_cast<float*>(s_lerp + TH + TW));
        flow = s_flow;
      }
      __syncthreads();
      const Win<TH, TW, HALO, DYN> win = tile_bound_read<TH, TW, HALO, DYN>(sh.bound, dt_bound);
      NoOwn no_own;
      tile_body<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, UNIFORM, GRID, DYN, true>(
          tr, win, flow, s_acc, sh, w.ev, H, W, tiles_x, pad_h, pad_w, w.slabs, w.spill, w.spill_epoch, epoch, w.halo_tab, pre,
          [&]() { if (next_live) request(wn, trn); }, no_own);
    } else if (next_live) {
      request(wn, trn);
    }
    __syncthreads();  // the flags / counters of this window are re-initialised by the next one
    tr = trn;
  }
  EBOS_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------
// forward B: combine slabs (+ spill) -> IWE, optional variance moments of the row segment
// ---------------------------------------------------------------------------------------------------
constexpr int kCombineBlock = 256;
// flag in the combine kernels' g_lo argument: the partial SUM of a workgroup adds every slab's contribution to a pixel as a double
// of its own instead of the pixel's f32 total.  The contributions are multiples of the fixed-point unit, so this sum is exact -- and
// therefore the same number however the image is cut up: the resident solver kernel (cmax_resident.hip) obtains it tile by tile from
// the LDS images, before any slab has travelled, and must arrive at the mean of this pass bit for bit.
constexpr int kCombineExactSum = 256;

// The variance without a finalize launch: every combine workgroup stores its partial pair write-through, drains and takes a number
// from a counter in the workspace; the LAST one reduces all pairs in moments_finalize_block's fixed order (same bits as the
// one-workgroup launch it replaces, which cost a kernel boundary and ~1 us of its own behind the combine pass) and leaves the
// counter at zero for the next call.  counter == nullptr: partials only (the caller reduces them).
constexpr int kFinalizeGroup = 32;
struct FinalizeIn {
  unsigned* counter;   // [32 * (1 + groups)] words: the top counter, then one per group of kFinalizeGroup workgroups, 128 bytes apart
  float* out_var;
  double* moments;
  long long n_pixels;
};
template <bool SC1 = false>
__device__ __forceinline__ void moments_finalize_block(const double* __restrict__ partials, int64_t nparts, int64_t m, float* out,
                                                       double* moments);
__device__ __forceinline__ void combine_store_partial(double* __restrict__ partials, int64_t b, int64_t nparts, double s, double ss,
                                                      const FinalizeIn& fin) {
  __shared__ int s_last_;
  if (fin.counter == nullptr) {
    if (threadIdx.x == 0) {
      partials[2 * b] = s;
      partials[2 * b + 1] = ss;
    }
    return;
  }
  if (threadIdx.x == 0) {
    typedef __attribute__((address_space(1))) double gf64_;
    typedef __attribute__((address_space(1))) unsigned gu32_;
    __hip_atomic_store((gf64_*)(partials + 2 * b), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (write-through)
    __hip_atomic_store((gf64_*)(partials + 2 * b + 1), ss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Two levels: same-address atomics serialise at the memory side (~88 per us: ~900 workgroups on ONE counter cost 10 us, the
    // whole pass ran 8 us longer than with a finalize launch).  Groups of kFinalizeGroup workgroups count on a counter of their
    // own -- 128 bytes apart, different channels --, the last of each group counts on the top counter.
    const int64_t g = b / kFinalizeGroup, n_groups = (nparts + kFinalizeGroup - 1) / kFinalizeGroup;
    const unsigned in_group = (unsigned)min((int64_t)kFinalizeGroup, nparts - g * kFinalizeGroup);
    gu32_* cg = (gu32_*)(fin.counter + 32 * (1 + g));
    bool last = __hip_atomic_fetch_add(cg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1u;
    if (last) {
      __hip_atomic_store(cg, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add((gu32_*)fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)n_groups - 1u;
    }
    s_last_ = last;
  }
  __syncthreads();
  if (!s_last_) return;
  if (threadIdx.x == 0) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)fin.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (coherent loads: the pairs were written by workgroups of other XCDs, and a line of this XCD's L2 that holds a neighbour's pair
  // from a write of its own would serve the others' stale)
  moments_finalize_block<true>(partials, nparts, fin.n_pixels, fin.out_var, fin.moments);
}

// halo_tab (DYN accumulate pass; nullptr otherwise): the window (hr, hc) each tile's slabs were stored with -- candidates are found
// with the largest window HALO, a candidate whose own window does not reach the pixel is skipped
template <int TH, int TW, int HALO>
__global__ void __launch_bounds__(kCombineBlock)
iwe_slab_combine_kernel(const float* __restrict__ slabs, float* spill, int tiles_y, int tiles_x, int splits, int H,
                        int W, int pad_h, int pad_w, float* __restrict__ iwe, int g_lo, double* __restrict__ partials,
                        const int32_t* __restrict__ part_off, const unsigned* __restrict__ spill_epoch, unsigned epoch,
                        const unsigned* __restrict__ halo_tab, FinalizeIn fin) {
  const bool spill_used = *spill_epoch == epoch;  // (uniform) some workgroup of THIS call's accumulate pass wrote spill taps
#ifndef EBOS_PLAIN_SLABS
  const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(slabs, 0xffffffffu);  // (offsets stay below the workspace size: < 4 GiB)
#endif
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  const int h = H + 2 * pad_h, w = W + 2 * pad_w;
  const int R = blockIdx.y, C = blockIdx.x * kCombineBlock + threadIdx.x;
  const int r = R - pad_h, c = C - pad_w;  // un-padded coordinates (may lie in the padding ring)
  const bool exact = (g_lo & kCombineExactSum) != 0;
  g_lo &= kCombineExactSum - 1;
  float v = 0.0f;
  double vd = 0.0;  // (exact: the contributions one by one)
  if (C < w) {
    // tiles whose LDS window [t*T - HALO, t*T + T + HALO) contains r (resp. c)
    int ty0 = (r - HALO - TH + 1 >= 0) ? (r - HALO - TH + 1 + TH - 1) / TH : 0;  // ceil((r - HALO - TH + 1) / TH) clamped at 0
    int ty1 = (r + HALO >= 0) ? (r + HALO) / TH : -1;
    if (ty1 > tiles_y - 1) ty1 = tiles_y - 1;
    int tx0 = (c - HALO - TW + 1 >= 0) ? (c - HALO - TW + 1 + TW - 1) / TW : 0;
    int tx1 = (c + HALO >= 0) ? (c + HALO) / TW : -1;
    if (tx1 > tiles_x - 1) tx1 = tiles_x - 1;
    for (int ty = ty0; ty <= ty1; ++ty) {
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int tile = ty * tiles_x + tx;
        int hr = HALO, hc = HALO;
        if (halo_tab != nullptr) {
          const unsigned t = halo_tab[tile];
          hr = (int)(t & 255u), hc = (int)(t >> 8);
        }
        const int rl = r - (ty * TH - hr), cl = c - (tx * TW - hc), lw = TW + 2 * hc;
        if ((unsigned)rl >= (unsigned)(TH + 2 * hr) || (unsigned)cl >= (unsigned)lw) continue;
        const int s0 = part_off ? part_off[tile] : tile * splits, np = part_off ? part_off[tile + 1] - s0 : splits;
#ifndef EBOS_PLAIN_SLABS
        const unsigned s_byte = ((unsigned)s0 * (unsigned)(LH * LW) + (unsigned)(rl * lw + cl)) * 4u;
        for (int p = 0; p < np; ++p) {
          const float t = slab_load1(all_slabs, s_byte + (unsigned)p * (unsigned)(LH * LW * 4));
          v += t;
          vd += (double)t;
        }
#else
        const float* s = slabs + (int64_t)s0 * (LH * LW) + rl * lw + cl;
        for (int p = 0; p < np; ++p) {
          v += s[(int64_t)p * (LH * LW)];
          vd += (double)s[(int64_t)p * (LH * LW)];
        }
#endif
      }
    }
    const int64_t gi = (int64_t)R * w + C;
    const float sp = spill_used ? spill[gi] : 0.0f;
    if (sp != 0.0f) {
      v += sp;
      vd += (double)sp;
      spill[gi] = 0.0f;  // keep the spill image zero between calls
    }
    iwe[gi] = v;
  }
  if (partials != nullptr) {
    const bool in = C < w && R >= g_lo && R < h - g_lo && C >= g_lo && C < w - g_lo;
    double s = in ? (exact ? vd : (double)v) : 0.0, ss = in ? (double)v * (double)v : 0.0;
    __shared__ double red[2 * kCombineBlock / kWave];
    block_sum2(s, ss, red);
    combine_store_partial(partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, (int64_t)gridDim.x * gridDim.y, s, ss, fin);
  }
}

// 4 pixels per thread (float4 loads/stores), 4 rows x 256 columns per workgroup -> ~900 workgroups and as
// many moment partials at 1280x720.  Needs w, pad_w, HALO, TW multiples of 4 (the scalar kernel covers the rest).
constexpr int kCombineRows = 4;

template <int TH, int TW, int HALO, bool DYN>
__device__ __forceinline__ void combine4_block(const float* __restrict__ slabs, float* spill, int tiles_y, int tiles_x, int splits, int H,
                                               int W, int pad_h, int pad_w, float* __restrict__ iwe, int g_lo,
                                               double* __restrict__ partials, const int32_t* __restrict__ part_off,
                                               const unsigned* __restrict__ spill_epoch, unsigned epoch,
                                               const unsigned* __restrict__ halo_tab, const FinalizeIn& fin = FinalizeIn{nullptr, nullptr, nullptr, 0}) {
  const bool spill_used = *spill_epoch == epoch;  // (uniform) some workgroup of THIS call's accumulate pass wrote spill taps
#ifndef EBOS_PLAIN_SLABS
  const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(slabs, 0xffffffffu);  // (offsets stay below the workspace size: < 4 GiB)
#endif
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  static_assert(HALO % 4 == 0 && TW % 4 == 0, "vector combine needs 4-aligned windows");
  const int h = H + 2 * pad_h, w = W + 2 * pad_w;
  const int tx_ = threadIdx.x & 63, ty_ = threadIdx.x >> 6;
  const int R = blockIdx.y * kCombineRows + ty_, C = (blockIdx.x * 64 + tx_) * 4;
  const int r = R - pad_h, c = C - pad_w;
  const bool exact = (g_lo & kCombineExactSum) != 0;  // (uniform) the solver's patch-grid route: see kCombineExactSum
  g_lo &= kCombineExactSum - 1;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  double vd[4] = {0.0, 0.0, 0.0, 0.0};
  const bool live = R < h && C < w;
  // DYN: the windows of the tiles this workgroup's 4 x 256 pixels can see, fetched ONCE into LDS (a per-candidate global load would
  // put a dependent L2 round trip in front of every slab load: +1.3 us on a 7.8 us pass)
  constexpr int kTabY = (kCombineRows + 2 * HALO + TH - 1) / TH + 2, kTabX = (256 + 2 * HALO + TW - 1) / TW + 2;  // (floor differences + 1)
  __shared__ unsigned s_tab[DYN ? kTabY * kTabX : 1];
  int tab_y0 = 0, tab_x0 = 0;
  if (DYN) {
    const int r_first = (int)blockIdx.y * kCombineRows - pad_h, c_first = (int)blockIdx.x * 256 - pad_w;
    tab_y0 = max((r_first - HALO - TH + 1 >= 0) ? (r_first - HALO) / TH : 0, 0);   // (a lower bound of every pixel's ty0 / tx0)
    tab_x0 = max((c_first - HALO - TW + 1 >= 0) ? (c_first - HALO) / TW : 0, 0);
    for (int i = threadIdx.x; i < kTabY * kTabX; i += kCombineBlock) {
      const int ty = min(tab_y0 + i / kTabX, tiles_y - 1), tx = min(tab_x0 + i % kTabX, tiles_x - 1);
      s_tab[i] = halo_tab[ty * tiles_x + tx];
    }
    __syncthreads();
  }
  if (live) {
    int ty0 = (r - HALO - TH + 1 >= 0) ? (r - HALO - TH + 1 + TH - 1) / TH : 0;
    int ty1 = (r + HALO >= 0) ? (r + HALO) / TH : -1;
    if (ty1 > tiles_y - 1) ty1 = tiles_y - 1;
    int tx0 = (c - HALO - TW + 1 >= 0) ? (c - HALO - TW + 1 + TW - 1) / TW : 0;
    int tx1 = (c + HALO >= 0) ? (c + HALO) / TW : -1;
    if (tx1 > tiles_x - 1) tx1 = tiles_x - 1;
    // The candidates: at most kNy x kNx tiles have a window that can reach this quad (the pixel's row lies in at most
    // floor((2 HALO + TH - 1) / TH) + 1 windows).  Written out as that many SLOTS whose first slab loads are all issued before the
    // first one is added: as nested loops over (ty, tx, part) the pass waited for each 16-byte load in turn -- up to four dependent
    // round trips per thread, most of its 6.4 us (the loads sat behind `s_waitcnt vmcnt(0)` one by one).  The additions keep the
    // loops' order (ty, tx, part), so the image has the same bits; parts beyond a tile's first (adaptive plans, splits > 1)
    // follow their slot's first in a rolled loop.
    constexpr int kNy = (2 * HALO + TH - 1) / TH + 1, kNx = (2 * HALO + TW - 1) / TW + 1;
    bool ok[kNy * kNx];
    unsigned byte0[kNy * kNx];
    int parts[kNy * kNx];
    float4 first[kNy * kNx];
    auto gather = [&](auto adaptive_tag) {  // (two copies: an adaptive plan's part offsets are loads of their own, issued first)
      constexpr bool kAdaptive = decltype(adaptive_tag)::value;
      int tile_of[kNy * kNx], cell[kNy * kNx], s0[kNy * kNx], s1[kNy * kNx];
#pragma unroll
      for (int a = 0; a < kNy; ++a) {
#pragma unroll
        for (int b = 0; b < kNx; ++b) {
          const int k = a * kNx + b, ty = ty0 + a, tx = tx0 + b;
          bool valid = ty <= ty1 && tx <= tx1;
          const int tyc = min(ty, tiles_y - 1), txc = min(tx, tiles_x - 1);  // (an unused slot still forms an address)
          tile_of[k] = tyc * tiles_x + txc;
          int hr = HALO, hc = HALO;
          if (DYN) {  // (hc is a multiple of 4 and so is c: a quad lies inside a window or outside it, never across its edge)
            const unsigned t = s_tab[min(tyc - tab_y0, kTabY - 1) * kTabX + min(txc - tab_x0, kTabX - 1)];
            hr = (int)(t & 255u), hc = (int)(t >> 8);
          }
          const int rl = r - (tyc * TH - hr), cl = c - (txc * TW - hc), lw = TW + 2 * hc;
          if (DYN) valid = valid && (unsigned)rl < (unsigned)(TH + 2 * hr) && (unsigned)cl < (unsigned)lw;
          ok[k] = valid;
          cell[k] = valid ? rl * lw + cl : 0;
          if (kAdaptive) {
            s0[k] = part_off[tile_of[k]];
            s1[k] = part_off[tile_of[k] + 1];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kNy * kNx; ++k) {
        const int first_slab = kAdaptive ? s0[k] : tile_of[k] * splits;
        parts[k] = kAdaptive ? s1[k] - s0[k] : splits;
        ok[k] = ok[k] && parts[k] > 0;
        byte0[k] = ok[k] ? ((unsigned)first_slab * (unsigned)(LH * LW) + (unsigned)cell[k]) * 4u : 0u;
#ifndef EBOS_PLAIN_SLABS
        first[k] = slab_load4(all_slabs, byte0[k]);
#else
        first[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slabs) + byte0[k]);
#endif
      }
    };
    if (part_off != nullptr) gather(std::true_type{});
    else gather(std::false_type{});
#pragma unroll
    for (int k = 0; k < kNy * kNx; ++k) {
      if (!ok[k]) continue;
      v.x += first[k].x;
      v.y += first[k].y;
      v.z += first[k].z;
      v.w += first[k].w;
      if (exact) vd[0] += (double)first[k].x, vd[1] += (double)first[k].y, vd[2] += (double)first[k].z, vd[3] += (double)first[k].w;
      // (a crowded tile of an adaptive plan has tens of parts: eight slab loads in flight per round trip -- one by one the pass took
      // 21.6 us on a window whose fullest tile holds 21 x the average, against 9 on a uniform one; the additions keep their order)
      constexpr int kPartBatch = 8;
      for (int p = 1; p < parts[k]; p += kPartBatch) {
        float4 t[kPartBatch];
#pragma unroll
        for (int j = 0; j < kPartBatch; ++j) {
          const unsigned q = (unsigned)min(p + j, parts[k] - 1);
#ifndef EBOS_PLAIN_SLABS
          t[j] = slab_load4(all_slabs, byte0[k] + q * (unsigned)(LH * LW * 4));
#else
          t[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(slabs) + byte0[k] + (size_t)q * (LH * LW * 4));
#endif
        }
#pragma unroll
        for (int j = 0; j < kPartBatch; ++j) {
          if (p + j >= parts[k]) break;
          v.x += t[j].x;
          v.y += t[j].y;
          v.z += t[j].z;
          v.w += t[j].w;
          if (exact) vd[0] += (double)t[j].x, vd[1] += (double)t[j].y, vd[2] += (double)t[j].z, vd[3] += (double)t[j].w;
        }
      }
    }
    const int64_t gi = (int64_t)R * w + C;
    const float4 s4 = spill_used ? *reinterpret_cast<const float4*>(spill + gi) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (s4.x != 0.f || s4.y != 0.f || s4.z != 0.f || s4.w != 0.f) {
      v.x += s4.x;
      v.y += s4.y;
      v.z += s4.z;
      v.w += s4.w;
      if (exact) vd[0] += (double)s4.x, vd[1] += (double)s4.y, vd[2] += (double)s4.z, vd[3] += (double)s4.w;
      *reinterpret_cast<float4*>(spill + gi) = make_float4(0.f, 0.f, 0.f, 0.f);  // keep the spill image zero
    }
    *reinterpret_cast<float4*>(iwe + gi) = v;
  }
  if (partials != nullptr) {
    double s = 0.0, ss = 0.0;
    if (live && R >= g_lo && R < h - g_lo) {
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (C + k >= g_lo && C + k < w - g_lo) {
          s += exact ? vd[k] : (double)e[k];
          ss += (double)e[k] * (double)e[k];
        }
    }
    __shared__ double red[2 * kCombineBlock / kWave];
    block_sum2(s, ss, red);
    // (the batched kernel's grid has windows in z: its partials are per window and it passes no counter)
    combine_store_partial(partials, (int64_t)blockIdx.y * gridDim.x + blockIdx.x, (int64_t)gridDim.x * gridDim.y, s, ss, fin);
  }
}

template <int TH, int TW, int HALO, bool DYN = false>
__global__ void __launch_bounds__(kCombineBlock)
iwe_slab_combine4_kernel(const float* __restrict__ slabs, float* spill, int tiles_y, int tiles_x, int splits, int H,
                         int W, int pad_h, int pad_w, float* __restrict__ iwe, int g_lo, double* __restrict__ partials,
                         const int32_t* __restrict__ part_off, const unsigned* __restrict__ spill_epoch, unsigned epoch,
                         const unsigned* __restrict__ halo_tab, FinalizeIn fin) {
  combine4_block<TH, TW, HALO, DYN>(slabs, spill, tiles_y, tiles_x, splits, H, W, pad_h, pad_w, iwe, g_lo, partials, part_off,
                                    spill_epoch, epoch, halo_tab, fin);
}

template <int TH, int TW, int HALO, bool DYN = false>
__global__ void __launch_bounds__(kCombineBlock)
iwe_slab_combine4_batch_kernel(FwdBatch b, int tiles_y, int tiles_x, int splits, int H, int W, int pad_h, int pad_w, int g_lo,
                               int want_var, unsigned epoch) {
  const FwdWindow& w = b.w[blockIdx.z];
  combine4_block<TH, TW, HALO, DYN>(w.slabs, w.spill, tiles_y, tiles_x, splits, H, W, pad_h, pad_w, w.iwe, g_lo,
                                    want_var ? w.partials : nullptr, splits == 0 ? w.ev.part_off : nullptr, w.spill_epoch, epoch,
                                    w.halo_tab);
}

// one workgroup: partials -> out (unbiased variance), moments (mean, M).  Fixed summation order.
template <bool SC1>
__device__ __forceinline__ void moments_finalize_block(const double* __restrict__ partials, int64_t nparts, int64_t m, float* out,
                                                       double* moments) {
  // four (sum, sum of squares) pairs per thread in flight at once: rolled, the loop waited for each 16-byte load in turn -- at the
  // ~900 partials of a 1280x720 image four round trips, most of what this one-workgroup kernel takes (4.6 us on average against
  // 2.4 at best in the rocprofv3 trace).  The order of the additions is the rolled loop's: same bits.
  double s = 0.0, ss = 0.0;
  const double2* __restrict__ pairs = reinterpret_cast<const double2*>(partials);
  for (int64_t base = threadIdx.x; base < nparts; base += 4 * (int64_t)blockDim.x) {
    double2 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = min(base + k * (int64_t)blockDim.x, nparts - 1);
      if (SC1) {
        typedef __attribute__((address_space(1))) double gf64_;
        v[k].x = __hip_atomic_load((gf64_*)(partials + 2 * i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[k].y = __hip_atomic_load((gf64_*)(partials + 2 * i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        v[k] = pairs[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (base + k * (int64_t)blockDim.x < nparts) {
        s += v[k].x;
        ss += v[k].y;
      }
  }
  __shared__ double red[8];
  block_sum2(s, ss, red);
  if (threadIdx.x == 0) {
    const double mean = m > 0 ? s / (double)m : 0.0;
    if (out) out[0] = (float)((ss - s * mean) / (double)(m - 1));
    if (moments) {
      moments[0] = mean;
      moments[1] = (double)m;
    }
  }
}
__global__ void __launch_bounds__(256)
moments_finalize_kernel(const double* __restrict__ partials, int64_t nparts, int64_t m, float* out, double* moments) {
  moments_finalize_block(partials, nparts, m, out, moments);
}
__global__ void __launch_bounds__(256) moments_finalize_batch_kernel(FwdBatch b, int64_t nparts, int64_t m) {
  const FwdWindow& w = b.w[blockIdx.x];
  moments_finalize_block(w.partials, nparts, m, w.out_var, w.moments);
}

// ---------------------------------------------------------------------------------------------------
// backward: d_flow tile by tile, no global atomics
// ---------------------------------------------------------------------------------------------------
struct GradImage {
  const float* g;
  float a, c;  // G = a * g + c inside the valid region, 0 outside
  int h, w, lo;
  // bk.k0 != 0: g is z = B^T (m . B x) of the blurred contrast (MomentsIn::blur) and G = a * g + c * wgt(R, C), wgt = B^T m with
  // m the valid region of ring `blo`; `lo` is then 0 (the region is part of z and wgt)
  Blur3 bk = {0.0f, 0.0f};
  int blo = 0;
  // the upstream value at (R, C) from the image's value there (the blurred form with explicit roundings: the resident solver
  // kernel evaluates the same expression on its gathered window)
  __device__ __forceinline__ float map(float src, int R, int C) const {
    if (bk.k0 != 0.0f) return __fmaf_rn(a, src, c * blur3_weight(R, C, h, w, blo, bk));
    return a * src + c;
  }
  __device__ __forceinline__ float at(int R, int C) const {  // padded coordinates
    if (R < lo || R >= h - lo || C < lo || C >= w - lo) return 0.0f;
    return map(g[(int64_t)R * w + C], R, C);
  }
  __device__ __forceinline__ void set_blur(const Blur3& b) {
    if (b.k0 != 0.0f) bk = b, blo = lo, lo = 0;
  }
};

// ---- lean backward sweep over a compact slice (unit weights) ---------------------------------------------------
// Same budget discipline as accumulate_compact_fx.  PASS_MAIN: events whose four taps lie inside the LDS window of the
// upstream image (others read a dummy cell and contribute 0, but raise the flag); PASS_SPILL: the rare second sweep
// for exactly those events, reading the upstream image from global memory.
// The scatter side.  A lane's four events are consecutive in the sorted plan and mostly share a source pixel:
// they are summed in registers per run, and a run adds to its pixel
//   MODE == ACC_F64   a pair of ds_add_f64 into s_d [2][TH * TW]  (16.6 - 24.3 issue units each: the kernel was LDS-bound,
//                     SQ_WAIT_INST_LDS 9.6 M of 57 M wave cycles, profiles/r02z_pmc_util.txt)
//   MODE == ACC_FX    ONE ds_add_u64 into s_w [TH * TW]: (d/du, d/dv) as two SIGNED 32-bit fixed-point fields of one word,
//                     word += (qv << 32) + qu in two's complement (a negative low field borrows from the high one and the
//                     decode gives it back).  The unit is chosen PER TILE: with n the largest event count of one of its source
//                     pixels (key_offsets) and C >= max |dt| x 2 max |upstream tile| the largest contribution of an event, an
//                     event gets p = min(28 - ceil(log2 n), 21) bits, fx_scale = 2^p / C rounded down to a power of two.  EXACT
//                     by construction, not by checksum (signed sums could cancel a wrap).  Main sweep: every EVENT is quantised
//                     by the multiply-add that forms its contribution (see the sweep below); |.| <= 2^p follows from the two
//                     bounds behind C, of which the upstream one holds by construction (C is taken from the staged tile) and the
//                     |dt| one is checked on the events (*bad otherwise).  Spill sweep (taps read from global memory, beyond the
//                     staged tile): run sums are converted and checked against fx_limit = 2^(p + 2) units (*bad otherwise).
//                     |pixel sum| <= n 2^(p + 2) <= 2^30: no field can leave its 32 bits.  A workgroup that raises *bad redoes
//                     its slice in ACC_F64; a tile with a hot pixel (n >= 1024: p < 18) takes ACC_F64 from the start.  Integer
//                     adds commute: the gradient is bit-reproducible, which the f64 atomics (order-dependent rounding) were not.
template <int TH, int TW, int HALO, bool UNIFORM, int PASS, bool GRID = false, bool DYN = false, int MODE = ACC_F64>
__device__ __forceinline__ void bwd_compact_slice(const TileRange& tr, double* s_d, const float* s_g, const EvPtrs& ev,
                                                  const float* __restrict__ flow, int H, int W, int pad_h, int pad_w,
                                                  const GradImage& G, double& tot_x, double& tot_y, bool* any_spill,
                                                  const ChunkQueue& queue, const Win<TH, TW, HALO, DYN>& win,
                                                  float fx_scale = 1.0f, float fx_limit = 0.0f, bool* bad = nullptr,
                                                  float dt_limit = 0.0f, const BwdPre* pre = nullptr) {
  unsigned long long* s_w = reinterpret_cast<unsigned long long*>(s_d);
  bool out_of_range = false;
  auto add_run = [&](unsigned pix, float ax, float ay) {
    if (MODE == ACC_FX) {
      const float sx = ax * fx_scale, sy = ay * fx_scale;
      out_of_range |= !(fmaxf(fabsf(sx), fabsf(sy)) < fx_limit);  // (also true for NaN)
      const int qx = (int)rintf(sx), qy = (int)rintf(sy);           // (v_rndne + v_cvt; a value past the limit converts to garbage: redone)
      atomicAdd(&s_w[pix], (unsigned long long)(((long long)qy << 32) + (long long)qx));
    } else {
      atomic_add(&s_d[pix], (double)ax);
      atomic_add(&s_d[TH * TW + pix], (double)ay);
    }
  };
  const int LH = win.LH(), LW = win.LW(), HR = win.HR(), HC = win.HC();  // (compile-time constants unless DYN)
  constexpr int PH = TH + 2 * kBwdApron, PW = TW + 2 * kBwdApron;  // GRID: the tile's flow (+ apron) in LDS
  const float* __restrict__ flow1 = UNIFORM ? flow : flow + (GRID ? (int64_t)PH * PW : (int64_t)H * W);
  const float uni_u = UNIFORM ? -flow[0] : 0.0f, uni_v = UNIFORM ? -flow[1] : 0.0f;
  const int tr0 = tr.ty * TH, tc0 = tr.tx * TW;
  const unsigned base_lin = GRID ? (unsigned)(kBwdApron * PW + kBwdApron) : (unsigned)(tr0 * W + tc0);
  const unsigned uW = GRID ? (unsigned)PW : (unsigned)W;
  // dense field in memory: gathers through a buffer descriptor with 32-bit offsets, as in accumulate_compact_fx
  constexpr bool kBuf = !UNIFORM && !GRID;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow), 0, kBuf ? 2 * H * W * (int)sizeof(float) : 0, 0x00020000);
  const int soff0 = (int)(base_lin * 4u), soff1 = soff0 + H * W * (int)sizeof(float);
  const unsigned uW4 = uW * 4u;
  auto fetch = [&](unsigned pr, unsigned pc, float& u, float& v) {
    if (UNIFORM) {
      u = uni_u, v = uni_v;
    } else if (kBuf) {
      const unsigned off = __umul24(pr, uW4) + (pc << 2);
      u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff0, 0));
      v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff1, 0));
    } else {
      const unsigned lin = base_lin + pr * uW + pc;
      u = flow[lin], v = flow1[lin];
    }
  };
  bool spilled = false;
  const int32_t g_last = tr.g_last;
  // dynamic chunks of 64 groups per wave, as in the forward loop: with a static stride the first wave was done 5.9 us
  // before the last one (in-kernel stamps)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  constexpr int kWaves = kBlock / kWave;
  if constexpr (MODE == ACC_FX && PASS == PASS_MAIN && !UNIFORM) {
    // The fixed-point main sweep: EVERY EVENT is quantised, by the multiply-add that forms its contribution -- fma(dt x scale,
    // dL/d(x', y'), 1.5 x 2^23) leaves round-to-nearest-even(dt x scale x d) in the low mantissa bits (|.| <= 2^21 units by the choice
    // of the unit: no range to test per value) -- and a lane's run of one source pixel is two int32 sums, packed into the (dv, du)
    // word only where the run ends.  Against converting run sums (two multiplies, two round + convert pairs, the range test and the
    // 64-bit pack at each of the 5 places a run can end, under divergence mostly executed): 239 -> 192 VALU instructions per
    // group of 4 events, none of them quarter rate (the LDS index is a 16-bit multiply, the pixel index comes with the group).  The three pipeline stages (group being processed / its successor, whose flow gathers fly / the one
    // whose event loads fly) are three named register sets used in rotation by a loop written out three times: no copies.
    // What the choice of the unit cannot promise -- |dt| within the caller's bound -- is checked on the events themselves
    // (*bad: the workgroup redoes its slice in f64); taps of the spill sweep (PASS_SPILL below) keep the per-run test.
    constexpr float kMagic = 12582912.0f;  // 1.5 x 2^23: bits 0x4B400000 + q for an integer |q| < 2^22
    const float nscale = -fx_scale;        // dL/dflow[src] += -dt * dL/d(x', y')
    float dt_max = 0.0f;
    auto flush = [&](unsigned pix, int qx, int qy) {
      atomicAdd(&s_w[pix], (unsigned long long)(((long long)qy << 32) + (long long)qx));
    };
    auto fetch_at = [&](unsigned off, float& u, float& v) {
      if (kBuf) {
        u = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff0, 0));
        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, soff1, 0));
      } else {
        u = flow[off], v = flow1[off];
      }
    };
    // (the upstream tile's LDS offset as an opaque scalar: folded as a constant it cost an add per tap pair -- ds_read2_b32 has 8-bit offsets)
    unsigned sg_off = (unsigned)(reinterpret_cast<const char*>(s_g) - reinterpret_cast<const char*>(s_w));
    asm volatile("" : "+s"(sg_off));
    const char* g_bytes = reinterpret_cast<const char*>(s_w) + sg_off;
    const unsigned g_pitch = kBuf ? uW4 : uW, g_shift = kBuf ? 2u : 0u, g_base = kBuf ? 0u : base_lin;
    auto step = [&](const BGroup& cur, const v2f (&f)[4], const BGroup& nxt, v2f (&fn)[4], BGroup& nn, int c_cur, int c_nn) {
      const bool lane_live = tr.g_first + c_cur * kWave + lane <= g_last;  // the last chunk may be partial
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u, v;
        fetch_at(nxt.goff[e], u, v);
        fn[e].x = u;
        fn[e].y = v;
      }
      load_bgroup<TH, TW>(nn, tr.g_first + c_nn * kWave + lane, tr, ev, g_pitch, g_shift, g_base);
      dt_max = fmaxf(fmaxf(dt_max, fabsf(cur.dt[0])), fabsf(cur.dt[1]));  // (a padding slot's NaN drops out of the max)
      dt_max = fmaxf(fmaxf(dt_max, fabsf(cur.dt[2])), fabsf(cur.dt[3]));
      unsigned run_pix = cur.pix[0];
      int qx = 0, qy = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float edt = cur.dt[e];
        const v2f l = f[e] * (-edt);  // (u, v) as a register pair: one packed multiply, no copies to form its operand
        const float lx = l.x, ly = l.y;
        const float r0 = floorf(lx + kEps), c0 = floorf(ly + kEps);
        // (clamped: a free output modifier, and a NaN -- padding slots carry dt = NaN, an Inf flow gives Inf - Inf -- becomes 0, so
        // that the masked event's zero factor below meets a finite gradient; the forward loop clamps its fractions at 0 as well)
        const float fr = fminf(fmaxf(lx - r0, 0.0f), 1.0f), fc = fminf(fmaxf(ly - c0, 0.0f), 1.0f);
        const int rl = (int)cur.pr[e] + HR + (int)r0, cl = (int)cur.pc[e] + HC + (int)c0;
        const bool ok = lane_live && (fabsf(lx) + fabsf(ly) < 5.0e8f);  // (the forward loop's test; false for NaN)
        const bool inside = ok && (unsigned)rl < (unsigned)(LH - 1) && (unsigned)cl < (unsigned)(LW - 1);
        spilled |= ok && !inside;
        // (16-bit multiply: full rate -- a plain 32-bit multiply-add is a quarter-rate instruction; an `inside` row fits by far)
        const float* p = reinterpret_cast<const float*>(
            g_bytes + 4u * (inside ? ((unsigned)rl & 0xffffu) * ((unsigned)LW & 0xffffu) + (unsigned)cl : 0u));  // cell 0..LW+1: always valid, finite
        const float g00 = p[0], g10 = p[LW], g01 = p[1], g11 = p[LW + 1];
        const float d0 = g10 - g00, d1 = g11 - g01, e0 = g01 - g00, e1 = g11 - g10;
        const float dx = d0 + fc * (d1 - d0);  // dL/dx' = (1 - fc) (g10 - g00) + fc (g11 - g01)
        const float dy = e0 + fr * (e1 - e0);  // dL/dy' = (1 - fr) (g01 - g00) + fr (g11 - g10)
        const float es = inside ? edt * nscale : 0.0f;
        const float tx = __builtin_fmaf(es, dx, kMagic), ty = __builtin_fmaf(es, dy, kMagic);
        if (e > 0 && cur.pix[e] != run_pix) {
          flush(run_pix, qx, qy);
          run_pix = cur.pix[e];
          qx = 0;
          qy = 0;
        }
        qx += (int)(__float_as_uint(tx) - 0x4B400000u);
        qy += (int)(__float_as_uint(ty) - 0x4B400000u);
      }
      flush(run_pix, qx, qy);
    };
    BGroup A, B, C;
    v2f fa[4], fb[4], fc3[4];
    int k0 = wave, k1 = wave + kWaves;
    if (pre) {  // (compile-time: the kernel passes its own object or nothing)
      A = pre->A;
      B = pre->B;
    } else {
      load_bgroup<TH, TW>(A, tr.g_first + k0 * kWave + lane, tr, ev, g_pitch, g_shift, g_base);
      load_bgroup<TH, TW>(B, tr.g_first + k1 * kWave + lane, tr, ev, g_pitch, g_shift, g_base);
    }
    if (pre && kBuf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) fa[e].x = pre->au[e], fa[e].y = pre->av[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u, v;
        fetch_at(A.goff[e], u, v);
        fa[e].x = u;
        fa[e].y = v;
      }
    }
    while (true) {  // (every exit test is wave-uniform)
      if (tr.g_first + k0 * kWave > g_last) break;
      const int k2 = queue.pull();
      step(A, fa, B, fb, C, k0, k2);
      if (tr.g_first + k1 * kWave > g_last) break;
      k0 = queue.pull();
      step(B, fb, C, fc3, A, k1, k0);
      if (tr.g_first + k2 * kWave > g_last) break;
      k1 = queue.pull();
      step(C, fc3, A, fa, B, k2, k1);
    }
    if (any_spill) *any_spill = spilled;
    if (bad) *bad = dt_max > dt_limit;
    return;
  }
  int c_cur = wave, c_nxt = wave + kWaves;
  CGroup cur, nxt;
  load_cgroup<TH, TW>(cur, tr.g_first + c_cur * kWave + lane, tr, ev);
  load_cgroup<TH, TW>(nxt, tr.g_first + c_nxt * kWave + lane, tr, ev);
  float fu[4], fv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) fetch(cur.pr[e], cur.pc[e], fu[e], fv[e]);
  while (tr.g_first + c_cur * kWave <= g_last) {  // wave-uniform
    const bool lane_live = tr.g_first + c_cur * kWave + lane <= g_last;  // the last chunk may be partial
    float gu[4], gv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) fetch(nxt.pr[e], nxt.pc[e], gu[e], gv[e]);
    const int c_nn = queue.pull();
    CGroup nn;
    load_cgroup<TH, TW>(nn, tr.g_first + c_nn * kWave + lane, tr, ev);
    // the lane's 4 events are consecutive in the sorted plan and mostly share one source pixel: sum per run
    unsigned run_pix = 0xffffffffu;
    float ax = 0.0f, ay = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float edt = cur.dt[e];
      const float lx = cur.fx[e] - edt * fu[e], ly = cur.fy[e] - edt * fv[e];  // (fractions: 0 for integer source pixels)
      const float r0 = floorf(lx + kEps), c0 = floorf(ly + kEps);
      const float fr = lx - r0, fc = ly - c0;
      const int rl = (int)cur.pr[e] + HR + (int)r0, cl = (int)cur.pc[e] + HC + (int)c0;
      // false for NaN (padding slots carry dt = NaN), +-Inf and anything beyond 2^29 -- the forward loop's test, so that value and
      // gradient agree on which events exist (an Inf displacement has NaN fractions: the spill sweep must not take it either)
      const bool ok = lane_live && (fabsf(lx) + fabsf(ly) < 5.0e8f);
      const bool inside = ok && (unsigned)rl < (unsigned)(LH - 1) && (unsigned)cl < (unsigned)(LW - 1);
      float g00, g10, g01, g11;
      bool use;
      if (PASS == PASS_MAIN) {
        spilled |= ok && !inside;
        use = inside;
        const float* p = &s_g[inside ? rl * LW + cl : 0];  // cell 0..LW+1 is always a valid address
        g00 = p[0];
        g10 = p[LW];
        g01 = p[1];
        g11 = p[LW + 1];
      } else {
        use = ok && !inside;
        const int R = tr0 - HR + rl + pad_h, C = tc0 - HC + cl + pad_w;
        g00 = use ? G.at(R, C) : 0.0f;
        g10 = use ? G.at(R + 1, C) : 0.0f;
        g01 = use ? G.at(R, C + 1) : 0.0f;
        g11 = use ? G.at(R + 1, C + 1) : 0.0f;
      }
      const float dx = (1.0f - fc) * (g10 - g00) + fc * (g11 - g01);  // dL/dx'
      const float dy = (1.0f - fr) * (g01 - g00) + fr * (g11 - g10);  // dL/dy'
      const float cx = use ? edt * dx : 0.0f, cy = use ? edt * dy : 0.0f;  // (select, not multiply: dx may be NaN)
      if (UNIFORM) {
        ax += cx;  // dL/dtheta += dt * dL/d(x', y')
        ay += cy;
      } else {
        const unsigned pix = cur.pr[e] * TW + cur.pc[e];
        if (pix != run_pix) {
          if (run_pix != 0xffffffffu) add_run(run_pix, ax, ay);
          run_pix = pix;
          ax = 0.0f;
          ay = 0.0f;
        }
        ax -= cx;  // dL/dflow[src] += -dt * dL/d(x', y')
        ay -= cy;
      }
    }
    if (UNIFORM) {
      tot_x += (double)ax;
      tot_y += (double)ay;
    } else if (run_pix != 0xffffffffu) {
      add_run(run_pix, ax, ay);
    }
    cur = nxt;
    nxt = nn;
    c_cur = c_nxt;
    c_nxt = c_nn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      fu[e] = gu[e];
      fv[e] = gv[e];
    }
  }
  if (any_spill) *any_spill = spilled;
  if (MODE == ACC_FX && bad) *bad = out_of_range;
}

// ---- pieces of the backward kernel that the resident solver kernel (cmax_resident.hip) runs too --------------------------------
// The unit of the fixed-point scatter (bwd_compact_slice, ACC_FX) from the per-wave maxima the set-up left in s_gmax
// [2 * waves]: |staged upstream tile| and events per source pixel.  fx = false: NaN / Inf upstream or a hot pixel -> f64.
struct FxUnit {
  bool fx;
  float scale, limit;
};
// s_gmax [3 * waves]: per-wave max |staged value|, max events per source pixel, sum |staged value| over the n_window staged values.
// kFxOutlier: ONE value 2^10 x the window's mean magnitude would leave every other event's contribution ~11 of its 21 bits (a
// spike of 1e4: 1.9 % error on the neighbouring pixels' gradients, test_backward_fixed_point_unit_under_an_upstream_outlier) --
// such a tile takes the f64 accumulators.
constexpr float kFxOutlier = 512.0f;
__device__ __forceinline__ FxUnit bwd_fx_unit(const float* s_gmax, float dt_bound, int n_window) {
  FxUnit u{true, 1.0f, 0.0f};
  float gmax = 0.0f, nmax = 1.0f, gsum = 0.0f;
#pragma unroll
  for (int k = 0; k < kBlock / kWave; ++k) {
    gmax = fmaxf(gmax, s_gmax[k]);
    nmax = fmaxf(nmax, s_gmax[kBlock / kWave + k]);
    gsum += s_gmax[2 * (kBlock / kWave) + k];
  }
  if (gmax * (float)n_window > kFxOutlier * gsum) {  // (max > 512 x mean; also an all-zero window with one value)
    u.fx = gmax == 0.0f;                              // (nothing staged at all: any unit is exact)
    if (!u.fx) return u;
  }
  int en;
  frexpf(nmax, &en);                    // nmax < 2^en
  const int pbits = min(28 - en, 21);   // bits of one event's contribution (21: the main sweep quantises per event through a
                                        // 2^23-based rounding constant, bwd_compact_slice)
  // an event contributes dt * (a convex combination of differences of neighbouring upstream values): |.| <= max |dt| x 2 gmax
  const float cmax = (dt_bound > 0.0f ? dt_bound : 1.0f) * 2.0f * gmax;
  if (!(cmax < 1.0e37f) || pbits < 18) {
    u.fx = false;  // NaN / Inf upstream, or a source pixel with >= 1024 events (a hot pixel: f64 keeps its tile's precision)
  } else {
    int e = 0;
    if (cmax > 0.0f) frexpf(cmax, &e);  // cmax = m 2^e, 0.5 <= m < 1
    u.scale = ldexpf(1.0f, max(min(pbits - e, 120), -120));  // cmax x scale <= 2^pbits
    u.limit = ldexpf(1.0f, pbits + 2);                       // a run of <= 4 events
  }
  return u;
}

// shared words of a workgroup's backward sweeps
struct BwdShared {
  int* spill;      // some event's taps left the LDS window of the upstream image
  int* bad;        // fixed-point scatter: a value left its range -> the slice is redone with f64 accumulators
  unsigned* next;  // chunk queue of the lean loop
};

// main sweep + (rare) spill sweep of a compact slice in one accumulation mode M; returns with every wave past its last add.
// before_spill: called by every thread ahead of the spill sweep, which reads the upstream image from GLOBAL memory (the resident
// kernel makes other workgroups' image tiles visible there).
template <int TH, int TW, int HALO, bool UNIFORM, bool GRID, bool DYN, int M, typename Hook>
__device__ __forceinline__ void bwd_sweeps_mode(const TileRange& tr, double* s_d, const float* s_g, const EvPtrs& ev,
                                                const float* __restrict__ flow, int H, int W, int pad_h, int pad_w, const GradImage& G,
                                                double& tot_x, double& tot_y, const ChunkQueue& queue,
                                                const Win<TH, TW, HALO, DYN>& win, const FxUnit& u, float dt_bound, const BwdPre& pre_v, bool has_pre,
                                                const BwdShared& sh, Hook&& before_spill) {
  const BwdPre* pre = has_pre ? &pre_v : nullptr;
  bool spilled = false, bad = false;
  // (`pre` reaches the sweep as the caller's object itself or as a literal nullptr, never through a select of pointers: that
  // kept the prefetched groups in scratch behind flat loads, and a flat access makes every counted vmcnt wait of the loop a
  // vmcnt(0) -- 13 us on the 23 us kernel)
  if (tr.g_first <= tr.g_last) {
    if constexpr (M == ACC_FX)
      bwd_compact_slice<TH, TW, HALO, UNIFORM, PASS_MAIN, GRID, DYN, M>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y,
                                                                        &spilled, queue, win, u.scale, u.limit, &bad,
                                                                        dt_bound > 0.0f ? dt_bound : 1.0f, pre);
    else
      bwd_compact_slice<TH, TW, HALO, UNIFORM, PASS_MAIN, GRID, DYN, M>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y,
                                                                        &spilled, queue, win, u.scale, u.limit, &bad,
                                                                        dt_bound > 0.0f ? dt_bound : 1.0f, nullptr);
  }
  if (spilled) *sh.spill = 1;
  if (bad) *sh.bad = 1;
  EBOS_STAMP_BWD(3);
  __syncthreads();
  EBOS_STAMP_BWD(4);
  if (*sh.spill) {  // rare second sweep; it draws its chunks afresh
    if (threadIdx.x == 0) *sh.next = 2 * (kBlock / kWave);
    before_spill();
    __syncthreads();
    bad = false;
    if (tr.g_first <= tr.g_last)
      bwd_compact_slice<TH, TW, HALO, UNIFORM, PASS_SPILL, GRID, DYN, M>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y,
                                                                         nullptr, queue, win, u.scale, u.limit, &bad);
    if (bad) *sh.bad = 1;
    __syncthreads();
  }
}

// the sweeps of a lean (compact, unit-weight) slice: fixed point when the unit allows it, redone in f64 if a value left its range.
// Returns whether the accumulators hold fixed-point words (true) or doubles (false).
template <int TH, int TW, int HALO, bool UNIFORM, bool GRID, bool DYN, typename Hook>
__device__ __forceinline__ bool bwd_lean_sweeps(const TileRange& tr, double* s_d, const float* s_g, const EvPtrs& ev,
                                                const float* __restrict__ flow, int H, int W, int pad_h, int pad_w, const GradImage& G,
                                                double& tot_x, double& tot_y, const ChunkQueue& queue,
                                                const Win<TH, TW, HALO, DYN>& win, const FxUnit& u, float dt_bound, const BwdPre& pre, bool has_pre,
                                                const BwdShared& sh, Hook&& before_spill) {
  if (u.fx) {
    bwd_sweeps_mode<TH, TW, HALO, UNIFORM, GRID, DYN, ACC_FX>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y, queue, win, u,
                                                              dt_bound, pre, has_pre, sh, before_spill);
    if (!*sh.bad) return true;
    // (uniform: read after the sweeps' last barrier) a value left the fixed-point range: exact redo in f64
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) s_d[i] = 0.0;  // (the words; the upper half was never touched)
    if (threadIdx.x == 0) *sh.next = 2 * (kBlock / kWave);
    __syncthreads();
  }
  bwd_sweeps_mode<TH, TW, HALO, UNIFORM, GRID, DYN, ACC_F64>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y, queue, win, u,
                                                             dt_bound, pre, has_pre, sh, before_spill);
  return false;
}

// the tile's gradient, pixel i, component ch: decoded from the fixed-point words or read from the f64 accumulators
template <int TH, int TW>
struct TileGrad {
  bool fx;
  float fx_inv;  // (a power of two: exact)
  const double* s_d;
  __device__ __forceinline__ float at(int i, int ch) const {
    if (fx) {
      // (both fields fit 32 signed bits: one v_cvt_f32_i32 each -- converting them as the 64-bit integers they are extracted as is a
      // ~15-instruction sequence per value, twice per pixel, in a pass bound by instruction issue)
      const long long w = (long long)reinterpret_cast<const unsigned long long*>(s_d)[i];
      const int lo = (int)(unsigned)(w & 0xffffffffll);
      const int hi = (int)((w - (long long)lo) >> 32);
      return (float)(ch ? hi : lo) * fx_inv;
    }
    return (float)s_d[ch * TH * TW + i];
  }
};

// one 4-byte store, plain or write-through (sc1: the value is handed to another workgroup of the SAME launch)
template <bool SC1, typename T>
__device__ __forceinline__ void store_scalar(T* p, T v) {
  if (SC1) __hip_atomic_store((__attribute__((address_space(1))) T*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store sc0 sc1
  else *p = v;
}

// Epilogue of the GRID backward pass: the flow regularisers on the tile's own flow (value partial -> *reg_out, gradient added per
// pixel), then the adjoint of the grid -> dense map restricted to this tile -> out [2][kGridCells][kGridCells] partial cell
// gradients.  Ends with the stores issued (not waited for).  SC1: write-through stores (resident kernel hand-off).
template <int TH, int TW, int HALO, bool SC1>
__device__ __forceinline__ void grid_tile_epilogue(const TileRange& tr, int tr0, int tc0, int H, int W, double* s_d, float* s_g,
                                                   const float* s_flow, const Lerp* s_lerp, const TileGrad<TH, TW>& grad,
                                                   const float* __restrict__ addend, float s_norm, float s_tv, double* reg_out,
                                                   float* out, unsigned sc1_tag = 0u) {
  // SC1 (resident solver kernel): `out` holds 8-byte granules {sc1_tag, value} -- the data is its own flag, consumers poll the
  // granules themselves (no separate flag store behind a drain, no dependent round trip for the values)
  auto put = [&](int idx, float v) {
    if constexpr (SC1) {
      typedef __attribute__((address_space(1))) unsigned long long gu64_t;
      __hip_atomic_store((gu64_t*)(reinterpret_cast<unsigned long long*>(out) + idx),
                         ((unsigned long long)sc1_tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      out[idx] = v;
    }
  };
  constexpr int AP = kBwdApron, PH = TH + 2 * AP, PW = TW + 2 * AP;
  const int64_t hw = (int64_t)H * W;
  // adjoint of the grid -> dense map on this tile, separable like the stand-alone adjoint: rows first
  //   S[ch][i][c] = sum_r wy(r, gi0 + i) * (d[ch][r][c] + addend),   then   P[ch][i][j] = sum_c wx(c, gj0 + j) * S[ch][i][c]
  const int rows = min(TH, H - tr0), cols = min(TW, W - tc0);
  const Lerp* s_ly = s_lerp + AP;        // row r of the tile
  const Lerp* s_lx = s_lerp + PH + AP;   // column c of the tile
  const int gi0 = s_ly[0].i0, ni = s_ly[rows - 1].i1 - gi0 + 1;   // the cells the TILE touches (the combine pass's indexing)
  const int gj0 = s_lx[0].i0, nj = s_lx[cols - 1].i1 - gj0 + 1;
  float* s_S = s_g;                          // the upstream tile is dead by now: [2][ni][TW] row sums,
  float* s_wy = s_g + 2 * kGridCells * TW;   // [ni][TH] row weights (0 beyond the image),
  float* s_wx = s_wy + kGridCells * TH;      // [nj][TW] column weights
  for (int idx = threadIdx.x; idx < ni * TH; idx += kBlock) {
    const int i = idx / TH, r = idx - i * TH;
    const Lerp l = s_ly[r];
    s_wy[idx] = r < rows ? (l.i0 == gi0 + i ? l.w0 : 0.0f) + (l.i1 == gi0 + i ? l.w1 : 0.0f) : 0.0f;
  }
  for (int idx = threadIdx.x; idx < nj * TW; idx += kBlock) {
    const int j = idx / TW, c = idx - j * TW;
    const Lerp l = s_lx[c];
    s_wx[idx] = c < cols ? (l.i0 == gj0 + j ? l.w0 : 0.0f) + (l.i1 == gj0 + j ? l.w1 : 0.0f) : 0.0f;
  }
  // This thread's pixels of the tile gradient: decoded from the accumulators (fixed-point words or doubles), plus what enters per
  // pixel (regulariser gradients), as PLANAR FLOATS s_f [2][TH * TW] for the row sums.
  //   kSeparate (where the dead upstream window has the room): s_f lies behind the sums' own buffers, every pixel is decoded,
  //     completed and stored in one go -- no values held in registers across a barrier, one barrier less (the resident solver kernel
  //     spilled them: 6.7 us for this half of the epilogue against 3.4 in the stand-alone backward kernel);
  //   otherwise in place of the accumulators, after a barrier: a pixel's float slots lie inside other pixels' words.
  constexpr int kPx = (TH * TW + kBlock - 1) / kBlock;
  constexpr bool kSeparate = (size_t)kGridCells * (3 * TW + TH) + (size_t)2 * TH * TW <= (size_t)(TH + 2 * HALO) * (TW + 2 * HALO);
  float* s_f = kSeparate ? s_wx + kGridCells * TW : reinterpret_cast<float*>(s_d);
  __shared__ double s_red_norm[kBlock / kWave];
  // one pixel's regulariser terms on the tile's own flow (in LDS, with a 2 px apron): value -> val, gradient -> (gu, gv).
  // flow_norm (src/costs/flow_norm.py:45-56: mean |flow|, s_norm = weight / (H W)) is pointwise; image_gradient
  // (src/costs/image_gradient.py:60-75: mean(|d/d row| + |d/d col|) over both components, s_tv = weight / (2 H W)) reads the
  // torch.gradient lines through the pixel, up to 2 px away -- the same device functions as the stand-alone regulariser kernel,
  // on LDS lines instead of global ones.
  auto regularise = [&](int rl, int cl, float& gu, float& gv, double& val) {
    const int o = (rl + AP) * PW + cl + AP;
    const float u = s_flow[o], v = s_flow[PH * PW + o];
    if (s_norm != 0.0f) {
      const float nrm = sqrtf(u * u + v * v);
      val += (double)(s_norm * nrm);
      if (nrm > 0.0f) {  // torch: the sub-gradient of the norm at 0 is 0
        const float inv = s_norm / nrm;
        gu += inv * u;
        gv += inv * v;
      }
    }
    if (s_tv != 0.0f) {
      // line bases such that base[i * stride] is sample i of the image column / row through this pixel
      const float* col_u = s_flow + (AP - tr0) * PW + cl + AP;
      const float* row_u = s_flow + (rl + AP) * PW + AP - tc0;
      const float* col_v = col_u + PH * PW;
      const float* row_v = row_u + PH * PW;
      const int r = tr0 + rl, c = tc0 + cl;
      val += (double)(s_tv * (fabsf(central(col_u, r, H, PW)) + fabsf(central(row_u, c, W, 1)) +
                              fabsf(central(col_v, r, H, PW)) + fabsf(central(row_v, c, W, 1))));
      gu += s_tv * (tv_adjoint(col_u, r, H, PW) + tv_adjoint(row_u, c, W, 1));
      gv += s_tv * (tv_adjoint(col_v, r, H, PW) + tv_adjoint(row_v, c, W, 1));
    }
  };
#ifdef EBOS_STAMPS_EPI
  EBOS_STAMP_BWD(1);
#endif
  if constexpr (kSeparate) {
    double val = 0.0;
    const bool first = tr.part == 0;  // the per-tile terms enter once per tile
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      const int idx = threadIdx.x + k * kBlock;
      if (idx < TH * TW) {
        const int rl = idx / TW, cl = idx - rl * TW;
#if defined(EBOS_ABL) && (EBOS_ABL & 128)
        float pu = 0.0f, pv = 0.0f;
        if (false) {
#else
        float pu = grad.at(idx, 0), pv = grad.at(idx, 1);
        if (first && rl < rows && cl < cols) {
#endif
          if (addend != nullptr) {
            const int64_t o = (int64_t)(tr0 + rl) * W + tc0 + cl;
            pu += addend[o];
            pv += addend[hw + o];
          }
          if (reg_out != nullptr) {
            float gu = 0.0f, gv = 0.0f;
            regularise(rl, cl, gu, gv, val);
            pu += gu;
            pv += gv;
          }
        }
        s_f[idx] = pu;
        s_f[TH * TW + idx] = pv;
      }
    }
#ifdef EBOS_STAMPS_EPI
    EBOS_STAMP_BWD(2);
#endif
    if (reg_out != nullptr) {
      val = wave_sum(val);  // per-wave partials, summed in wave order by one thread after the barrier below (deterministic)
      if ((threadIdx.x & (kWave - 1)) == 0) s_red_norm[threadIdx.x / kWave] = val;
    }
#ifdef EBOS_STAMPS_EPI
    EBOS_STAMP_BWD(7);
#endif
    __syncthreads();
  } else {
  float px_u[kPx], px_v[kPx];
#pragma unroll
  for (int k = 0; k < kPx; ++k) {
    const int idx = min((int)threadIdx.x + k * kBlock, TH * TW - 1);
    px_u[k] = grad.at(idx, 0);
    px_v[k] = grad.at(idx, 1);
  }
  if (addend != nullptr && tr.part == 0) {  // the regulariser gradient enters once per tile (coalesced row reads)
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      const int idx = threadIdx.x + k * kBlock;
      const int rl = idx / TW, cl = idx - rl * TW;
      if (idx < TH * TW && rl < rows && cl < cols) {
        const int64_t o = (int64_t)(tr0 + rl) * W + tc0 + cl;
        px_u[k] += addend[o];
        px_v[k] += addend[hw + o];
      }
    }
  }
  if (reg_out != nullptr) {
    double val = 0.0;
    if (tr.part == 0) {
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        const int idx = threadIdx.x + k * kBlock;
        const int rl = idx / TW, cl = idx - rl * TW;
        if (idx < TH * TW && rl < rows && cl < cols) {
          float gu = 0.0f, gv = 0.0f;
          regularise(rl, cl, gu, gv, val);
          px_u[k] += gu;
          px_v[k] += gv;
        }
      }
    }
    val = wave_sum(val);  // per-wave partials, summed in wave order by one thread after the barrier below (deterministic)
    if ((threadIdx.x & (kWave - 1)) == 0) s_red_norm[threadIdx.x / kWave] = val;
  }
  __syncthreads();  // every accumulator has been read
#pragma unroll
  for (int k = 0; k < kPx; ++k) {
    const int idx = threadIdx.x + k * kBlock;
    if (idx < TH * TW) {
      s_f[idx] = px_u[k];
      s_f[TH * TW + idx] = px_v[k];
    }
  }
  __syncthreads();
  }
  EBOS_STAMP_BWD(5);
  if (reg_out != nullptr && threadIdx.x == 0) {
    double val = 0.0;
    for (int k = 0; k < kBlock / kWave; ++k) val += s_red_norm[k];
    *reg_out = val;  // (resident kernel: a word of its own LDS)
  }
#if defined(EBOS_ABL) && (EBOS_ABL & 16)   // (timing build of the resident solver kernel: without the adjoint's sums)
  for (int o = threadIdx.x; o < 2 * ni * nj; o += kBlock) {
    const int ch = o / (ni * nj), rem = o - ch * (ni * nj);
    put((ch * kGridCells + rem / nj) * kGridCells + rem % nj, 0.0f);
  }
  return;
#endif
  for (int idx = threadIdx.x; idx < 2 * ni * TW; idx += kBlock) {
    const int ch = idx / (ni * TW), rem = idx - ch * (ni * TW);
    const int i = rem / TW, c = rem - i * TW;
    const float* d = s_f + ch * TH * TW + c;
    const float* wy = s_wy + i * TH;
    float acc = 0.0f;  // (bounding r to the rows that touch the cell -- about half -- was slower, with a search for the bounds and with
                       // their closed form (`support`) alike: +0.4 us per solver iteration, the loop no longer unrolls by 9; four
                       // columns per thread with 16-byte reads likewise: 2.2 -> 2.9 us for this half of the epilogue)
#pragma unroll 9
    for (int r = 0; r < TH; ++r) acc += wy[r] * d[r * TW];
    s_S[idx] = acc;
  }
  __syncthreads();
  // columns: one wavefront per output, lanes stride over the tile's columns
  const int lane = threadIdx.x & (kWave - 1);
  for (int o = threadIdx.x / kWave; o < 2 * ni * nj; o += kBlock / kWave) {
    const int ch = o / (ni * nj), rem = o - ch * (ni * nj);
    const int i = rem / nj, j = rem - i * nj;
    const float* S = s_S + (ch * ni + i) * TW;
    const float* wx = s_wx + j * TW;
    float acc = 0.0f;
    for (int c = lane; c < TW; c += kWave) acc += wx[c] * S[c];
    acc = wave_sum(acc);
    if (lane == 0) put((ch * kGridCells + i) * kGridCells + j, acc);
  }
  EBOS_STAMP_BWD(6);
}


// UNIFORM: 2-DoF model (flow == theta pair, x' = x + dt * theta): no flow gathers, and instead of a per-pixel
// d_flow tile every lane sums dt * dL/d(x', y'); the workgroup writes one partial pair, summed over tiles afterwards.
// GRID: `flow_arg` is the patch grid [2, gh, gw]; the tile's dense flow is evaluated into LDS, and instead of a d_flow tile
// the workgroup writes the adjoint of the grid -> dense map restricted to its tile: a block of <= kGridCells x kGridCells
// partial cell gradients per flow component (part_out [items][2][kGridCells][kGridCells]); patch_grad_combine_kernel
// (flow_upsample.hip) sums the tiles that touch a cell.  `adaptive`: work items of the plan's part table.
// DYN: the LDS window of the upstream image is chosen per tile at run time (Win), from the same bound as the forward pass's --
// the 63 KB per workgroup that a 32 px halo stages shrink to what the tile's displacements can reach.
// FRAC: a compact plan with the fractions of undistorted events: the sweep is the f64 one (the fixed-point sweep's groups hold integer pixels)
template <int TH, int TW, int HALO, bool HAS_W, int FMT, bool UNIFORM, bool GRID = false, bool DYN = false, bool FRAC = false>
__global__ void __launch_bounds__(kBlock)
iwe_dense_tiled_bwd_kernel(EvPtrs ev, const int32_t* __restrict__ key_offsets, const float* __restrict__ flow_arg, int H, int W,
                           int tiles_x, int pad_h, int pad_w, const float* __restrict__ g_image,
                           const float* __restrict__ affine, int g_lo, float* __restrict__ d_flow,
                           float* __restrict__ d_weight, double* __restrict__ partials,
                           const double* __restrict__ var_moments, const float* __restrict__ upstream,
                           const float* __restrict__ addend, float* __restrict__ part_out, GridSrc gs, int adaptive,
                           float s_norm, float s_tv, double* __restrict__ reg_partials, MomentsIn mj, float dt_bound) {
  constexpr int kLHmax = TH + 2 * HALO, kLWmax = TW + 2 * HALO;
  static_assert(!DYN || (FMT == FMT_COMPACT && !HAS_W), "run-time windows: the lean loop only");
  extern __shared__ double s_raw[];
  double* s_d = s_raw;                                             // [2][TH*TW] d_flow accumulators
  float* s_g = reinterpret_cast<float*>(s_raw + 2 * TH * TW);      // [LH][LW] upstream gradient tile
  constexpr int AP = kBwdApron, PH = TH + 2 * AP, PW = TW + 2 * AP;
  float* s_flow = s_g + kLHmax * kLWmax;                           // GRID: [2][PH][PW] flow of this tile + apron
  Lerp* s_lerp = reinterpret_cast<Lerp*>(s_flow + 2 * PH * PW);    // GRID: [PH + PW] row / column interpolation
  // part_out != nullptr (dense): adaptive work items -- this workgroup is one part of a tile and writes its partial d_flow
  // tile to slab tr.slab of part_out; bwd_parts_combine_kernel sums the parts
  // (the accumulator clear needs nothing: first, under the tile range's loads)
  for (int i = threadIdx.x; i < 2 * TH * TW; i += kBlock) s_d[i] = 0.0;
  const TileRange tr = tile_range<FMT>(key_offsets, ev, TH * TW, tiles_x, GRID ? (adaptive ? 0 : 1) : (part_out ? 0 : 1));
  __shared__ double s_mom[2];
  __shared__ int s_spill;  // some event's taps left the LDS window of the upstream image
  __shared__ int s_bad;    // fixed-point scatter: a run sum left its range -> the slice is redone with f64 accumulators
  __shared__ unsigned s_next;  // chunk queue of the lean loop
  __shared__ float s_bound[2 * kBlock / kWave];  // DYN: per-wave maxima of |u|, |v| over the tile
  __shared__ float s_gmax[3 * kBlock / kWave];   // per wave: max |upstream tile|, max events per source pixel, sum |upstream tile| (fixed-point unit)
  const ChunkQueue queue{&s_next};
  EBOS_STAMP_BWD(0);
  if (tr.ty < 0 && !(mj.partials != nullptr && blockIdx.x == 0)) return;  // unused work item (workgroup 0 still reports the variance)
  const bool var_mj = mj.partials != nullptr && mj.mode == 0;
  if (mj.partials != nullptr && mj.mode == 1) {  // (uniform) the contrast VALUE of the gradient-magnitude job: workgroup 0 sums its partials
    if (blockIdx.x == 0) {
      double sv = 0.0;
      for (int64_t i = threadIdx.x; i < mj.n_partials; i += kBlock) sv += mj.partials[i];
      __shared__ double red_v[kBlock / kWave];
      sv = block_sum(sv, red_v);
      if (threadIdx.x == 0) {
        const float v = (float)(sv / (double)mj.n_pixels);
        if (mj.out_var) mj.out_var[0] = v;
        if (mj.out_scaled) mj.out_scaled[0] = (upstream ? upstream[0] : 1.0f) * v;
      }
    }
    if (tr.ty < 0) return;
  }
  const float* flow = flow_arg;
  const int64_t hw = (int64_t)H * W;
  GradImage G;
  G.g = g_image;
  G.a = affine ? affine[0] : 1.0f;
  G.c = affine ? affine[1] : 0.0f;
  G.h = H + 2 * pad_h;
  G.w = W + 2 * pad_w;
  G.lo = g_lo;
  const int tr0 = max(tr.ty, 0) * TH, tc0 = max(tr.tx, 0) * TW;
  if (threadIdx.x == 0) {
    s_spill = 0;
    s_bad = 0;
    s_next = 2 * (kBlock / kWave);
  }
  // Fixed-point scatter (bwd_compact_slice, ACC_FX): its unit depends on the largest event count of a source pixel of the tile --
  // the per-pixel counts of the WHOLE tile (an upper bound for a part of it), read off the plan's key offsets while everything
  // else loads
  constexpr bool kFxScatter = (FMT == FMT_COMPACT) && !HAS_W && !UNIFORM && !FRAC;
  int nmax_t = 1;
  if (kFxScatter && tr.ty >= 0) {
    const int32_t* ko = key_offsets + (int64_t)(tr.ty * tiles_x + tr.tx) * (TH * TW);
#pragma unroll
    for (int k = 0; k < (TH * TW + kBlock - 1) / kBlock; ++k) {
      const int i = min((int)threadIdx.x + k * kBlock, TH * TW - 1);
      nmax_t = max(nmax_t, ko[i + 1] - ko[i]);
    }
  }
  float gmax_t = 0.0f, gsum_t = 0.0f;  // this thread's max / sum of |staged upstream value|
  // ---- set-up.  Every global read in flight at once: (1) the raw upstream tile into registers -- unconditional, clamped loads,
  // fully unrolled (a loop of bounds-checked loads is waited for one by one: 5.6 us of set-up per workgroup, in-kernel stamps; not
  // gated by "this item has events": that is known one round trip later than the tile's position) -- (2) GRID: the tile's block of
  // grid cells, (3) the variance partials the forward call left (want_variance = 2: every workgroup reduces them itself while its
  // staging loads fly, no finalize launch between forward and backward; workgroup 0 reports the variance).  Then the LDS work:
  // accumulator clear, affine map + store of the upstream tile, GRID: the tile's flow.
  // DYN: the window decides WHERE the upstream tile is, and its bound -- (2) / the tile's flow values / theta -- is one round trip
  // away.  The SMALLEST window (4 px: what a converged BOS flow needs) is therefore staged speculatively with everything else;
  // only a tile whose bound asks for more stages again, one round trip later.
  // Thread -> (row r_off + k x rows-per-pass, column c) of the window, rows-per-pass = kBlock / LW whole rows: a thread's column,
  // its clamp and its validity are found ONCE, its rows step by a constant, and its LDS slot is threadIdx + k x (rows-per-pass x LW).
  // (A flat index i = threadIdx + k x kBlock per element cost a division, two clamps, four validity compares and a 64-bit address
  // per element and pass: ~32 instructions x 16 elements of the ~750 a thread spends outside the event loop at 10 M events.)
  constexpr int kRowsMin = kBlock / kLWmax;
  constexpr int kStage = (kLHmax + kRowsMin - 1) / kRowsMin;
  constexpr int kSpecHalo = HALO < 4 ? HALO : 4;
  constexpr int kSpecStage = DYN ? ((TH + 2 * kSpecHalo) + kBlock / (TW + 2 * kSpecHalo) - 1) / (kBlock / (TW + 2 * kSpecHalo)) : 1;
  static_assert(kRowsMin >= 1 && kSpecStage <= kStage, "a window row fits a pass");
  float raw[kStage], raw_spec[kSpecStage];  // (two register sets: the real window's loads must not wait for the speculative ones)
  Win<TH, TW, HALO, DYN> win{HALO, HALO};
  bool spec_hit = false;
  auto stage_loads = [&](float* dst, int n_stage) {
    const int LW = win.LW(), LH = win.LH(), rpp = kBlock / LW;
    const int r_off = DYN ? (int)(((float)threadIdx.x + 0.5f) * (1.0f / (float)LW)) : (int)threadIdx.x / LW, c = (int)threadIdx.x - r_off * LW;
    const int C = min(max(tc0 - win.HC() + c + pad_w, 0), G.w - 1), R0 = tr0 - win.HR() + pad_h + r_off;
    const float* col = g_image + C;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      if (k >= n_stage || k * rpp >= LH) break;  // (uniform)
      const int R = min(max(R0 + k * rpp, 0), G.h - 1);   // (rows past the window and the idle threads of a pass: clamped, unused)
      dst[k] = col[(int64_t)R * G.w];
    }
  };
  auto stage_store = [&](const float* src, int n_stage) {  // affine map of the upstream image (the variance gradient), zero outside the valid region
    const int LW = win.LW(), LH = win.LH(), rpp = kBlock / LW;
    const int r_off = DYN ? (int)(((float)threadIdx.x + 0.5f) * (1.0f / (float)LW)) : (int)threadIdx.x / LW, c = (int)threadIdx.x - r_off * LW;
    const int C = tc0 - win.HC() + c + pad_w, R0 = tr0 - win.HR() + pad_h + r_off;
    const bool col_ok = r_off < rpp, col_valid = C >= G.lo && C < G.w - G.lo;
    float* dstc = s_g + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      if (k >= n_stage || k * rpp >= LH) break;
      const int R = R0 + k * rpp;
      const bool valid = col_valid && R >= G.lo && R < G.h - G.lo;
      const float gv = valid ? G.map(src[k], R, C) : 0.0f;
      if (col_ok && r_off + k * rpp < LH) {
        dstc[k * rpp * LW] = gv;
        gmax_t = fmaxf(gmax_t, gv == gv ? fabsf(gv) : INFINITY);  // (a NaN counts as Inf: such a tile takes the f64 path)
        gsum_t += fabsf(gv);
      }
    }
  };
  // An EMPTY tile (no events in the whole tile: the static background of a recording) sweeps nothing: it needs neither the upstream
  // window nor the mean -- only what its epilogue writes (zeros, GRID: the regularisers' adjoint on its pixels).  Its work item then
  // costs ~7 us instead of 13, which is what the second round of a crowded window's work items consists of.
  const bool tile_empty = tr.ty >= 0 && tr.tile_groups == 0;
  if (tile_empty) {
    // (nothing staged)
  } else if (DYN) {
    win = Win<TH, TW, HALO, DYN>{kSpecHalo, kSpecHalo};
    stage_loads(raw_spec, kSpecStage);
  } else {
    stage_loads(raw, kStage);
  }
  // (4) the fixed-point sweep's first two chunks of events per wave (BwdPre): decoded, and the first one's flow gathered, further down
  constexpr bool kPre = kFxScatter;
  const bool has_events = tr.ty >= 0 && tr.g_first <= tr.g_last;
  BwdPreRaw pre_raw;
  BwdPre pre;
  if (kPre && has_events) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    pre_raw.A = load_craw(tr.g_first + wave * kWave + lane, tr, ev);
    pre_raw.B = load_craw(tr.g_first + (wave + kBlock / kWave) * kWave + lane, tr, ev);
  }
  double sm = 0.0, sq = 0.0;
  const bool reduce_mj = var_mj && (!tile_empty || blockIdx.x == 0);   // (workgroup 0 reports the variance whatever its tile holds)
  if (reduce_mj) {
    for (int64_t i = threadIdx.x; i < mj.n_partials; i += kBlock) {
      sm += mj.partials[2 * i];
      sq += mj.partials[2 * i + 1];
    }
  }
  EBOS_STAMP_BWD_S(2);
  TileGrid tg{};
  if (GRID) tg = tile_grid_begin<TH, TW, AP>(flow_arg, gs, tr0, tc0, H, W, s_lerp);  // (the partials' loads fly over its barrier)
  EBOS_STAMP_BWD_S(3);
  if (DYN) {
    float mu, mv;
    if (UNIFORM) {
      mu = fabsf(flow_arg[0]), mv = fabsf(flow_arg[1]);
    } else if (GRID) {  // (the cell block covers tile + apron: a superset of what the tile's own pixels interpolate)
      const int idx = min((int)threadIdx.x, 2 * tg.ni * tg.nj - 1);
      const bool second = idx >= tg.ni * tg.nj;
      mu = second ? 0.0f : fabsf(tg.cell), mv = second ? fabsf(tg.cell) : 0.0f;
    } else {
      tile_flow_absmax<TH, TW>(flow_arg, H, W, tr0, tc0, mu, mv);
    }
    tile_bound_post(mu, mv, s_bound);
  }
  EBOS_STAMP_BWD_S(4);
  if (reduce_mj) {
    __shared__ double red_m[2 * kBlock / kWave];
    block_sum2(sm, sq, red_m);
    if (threadIdx.x == 0) {
      const double mean = mj.n_pixels > 0 ? sm / (double)mj.n_pixels : 0.0;
      s_mom[0] = mean;
      if (blockIdx.x == 0) {
        if (mj.out_var) mj.out_var[0] = (float)((sq - sm * mean) / (double)(mj.n_pixels - 1));
        if (mj.out_scaled) mj.out_scaled[0] = (upstream ? upstream[0] : 1.0f) * (float)((sq - sm * mean) / (double)(mj.n_pixels - 1));
        if (mj.moments) {
          mj.moments[0] = mean;
          mj.moments[1] = (double)mj.n_pixels;
        }
      }
    }
    __syncthreads();
    if (tr.ty < 0) return;  // (workgroup 0 of an adaptive plan may be an unused item: it only reports the variance)
    // d var / d IWE = 2 (IWE - mean) / (M - 1) as an affine map of the IWE, from the partials reduced above
    const double a = 2.0 * (upstream ? (double)upstream[0] : 1.0) / ((double)mj.n_pixels - 1.0);  // (null upstream: 1)
    G.a = (float)a;
    G.c = (float)(-a * s_mom[0]);
    G.set_blur(mj.blur);
  } else {
    if (DYN) __syncthreads();  // publishes s_bound (the reduction above has barriers of its own)
    if (var_moments != nullptr && !var_mj) {
      // g_image is the IWE itself and the loss is upstream * var(IWE): d var / d IWE = 2 (IWE - mean) / (M - 1), folded
      // in as an affine map (no d_iwe image, no separate affine kernel)
      const double a = 2.0 * (double)upstream[0] / (var_moments[1] - 1.0);
      G.a = (float)a;
      G.c = (float)(-a * var_moments[0]);
    }
  }
  EBOS_STAMP_BWD_S(5);
  if (DYN) {
    const Win<TH, TW, HALO, DYN> need = tile_bound_read<TH, TW, HALO, DYN>(s_bound, dt_bound);
    spec_hit = need.hr <= win.hr && need.hc <= win.hc;
    if (!spec_hit) {  // (uniform) the speculative window is too small: stage the real one
      win = need;
      if (!tile_empty) stage_loads(raw, kStage);
    }
  }
  EBOS_STAMP_BWD_S(6);
  if (kPre && has_events) {
    const unsigned pitch = GRID ? (unsigned)PW : 4u * (unsigned)W, shift = GRID ? 0u : 2u, base = GRID ? (unsigned)(AP * PW + AP) : 0u;
    decode_bgroup(pre.A, pre_raw.A, pitch, shift, base);
    decode_bgroup(pre.B, pre_raw.B, pitch, shift, base);
    finish_bgroup<TW>(pre.A);
    finish_bgroup<TW>(pre.B);
    if (!GRID) {  // (GRID: the tile's flow is in LDS after the barrier below -- the sweep fetches it there)
      const __amdgpu_buffer_rsrc_t rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(flow_arg), 0, 2 * H * W * (int)sizeof(float), 0x00020000);
      const int soff0 = (tr0 * W + tc0) * (int)sizeof(float), soff1 = soff0 + H * W * (int)sizeof(float);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pre.au[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)pre.A.goff[e], soff0, 0));
        pre.av[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)pre.A.goff[e], soff1, 0));
      }
    }
  }
  EBOS_STAMP_BWD(1);
  if (!tile_empty && (GRID || tr.g_first <= tr.g_last)) {
    if (DYN && spec_hit) stage_store(raw_spec, kSpecStage);
    else stage_store(raw, kStage);
  }
  if (GRID) {
    tile_grid_finish<TH, TW, AP>(tg, s_flow, s_lerp, reinterpret_cast<float*>(s_lerp + PH + PW));
    flow = s_flow;
  }
  if (kFxScatter) {
    gmax_t = wave_max_nonneg(gmax_t);
    gsum_t = wave_sum(gsum_t);
    const float nm = wave_max_nonneg((float)nmax_t);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      s_gmax[threadIdx.x / kWave] = gmax_t;
      s_gmax[kBlock / kWave + threadIdx.x / kWave] = nm;
      s_gmax[2 * (kBlock / kWave) + threadIdx.x / kWave] = gsum_t;
    }
  }
  __syncthreads();
  EBOS_STAMP_BWD(2);
  const int LH = win.LH(), LW = win.LW();
  const int oy = tr0 - win.HR(), ox = tc0 - win.HC();
  bool fx = kFxScatter;
  float fx_scale = 1.0f, fx_limit = 0.0f;
  if (fx) {
    const FxUnit unit = bwd_fx_unit(s_gmax, dt_bound, win.LH() * win.LW());
    fx = unit.fx, fx_scale = unit.scale, fx_limit = unit.limit;
  }

  double tot_x = 0.0, tot_y = 0.0;  // UNIFORM: this lane's sum of dt * dL/d(x', y')
  constexpr bool kLean = (FMT == FMT_COMPACT) && !HAS_W;
  if (kLean) {
    const BwdShared bsh{&s_spill, &s_bad, &s_next};
    const FxUnit unit{fx, fx_scale, fx_limit};
    if constexpr (kPre)
      fx = bwd_lean_sweeps<TH, TW, HALO, UNIFORM, GRID, DYN>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y, queue, win, unit,
                                                             dt_bound, pre, true, bsh, NoHook{});
    else
      fx = bwd_lean_sweeps<TH, TW, HALO, UNIFORM, GRID, DYN>(tr, s_d, s_g, ev, flow, H, W, pad_h, pad_w, G, tot_x, tot_y, queue, win, unit,
                                                             dt_bound, pre, false, bsh, NoHook{});
  } else if (tr.g_first <= tr.g_last) {
    const float* __restrict__ flow1 = UNIFORM ? flow : flow + hw;
    const float uni_u = UNIFORM ? -flow[0] : 0.0f, uni_v = UNIFORM ? -flow[1] : 0.0f;
    const int32_t g_last = tr.g_last;
    int32_t grp = tr.g_first + threadIdx.x;
    Group cur, nxt;
    load_group<FMT, HAS_W, TH, TW>(cur, grp, tr, ev, tr0, tc0);
    load_group<FMT, HAS_W, TH, TW>(nxt, grp + kBlock, tr, ev, tr0, tc0);
    float fu[4], fv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lin = cur.rs[e] * W + cur.cs[e];
      fu[e] = UNIFORM ? uni_u : flow[lin];
      fv[e] = UNIFORM ? uni_v : flow1[lin];
    }
    while (grp <= g_last) {
      float gu[4], gv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lin = nxt.rs[e] * W + nxt.cs[e];
        gu[e] = UNIFORM ? uni_u : flow[lin];
        gv[e] = UNIFORM ? uni_v : flow1[lin];
      }
      Group nn;
      load_group<FMT, HAS_W, TH, TW>(nn, grp + 2 * kBlock, tr, ev, tr0, tc0);
      // The lane's 4 events are consecutive in the sorted plan, so they mostly share one source pixel: sum
      // them in registers and issue one pair of LDS adds per run instead of a wave-wide shuffle reduction.
      int run_pix = -1;
      float ax = 0.0f, ay = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float wv = cur.w[e];
        // plan index of this slot: FMT_XY groups are aligned to the plan, compact groups to the tile
        const int32_t i = FMT == FMT_COMPACT ? tr.beg + 4 * (grp - tr.g_first) + e : 4 * grp + e;
        // (a real weight may be 0 -- its d_weight is still wanted: only unit-weight compact groups mark their padding by w == 0)
        const bool live = (FMT == FMT_COMPACT && !HAS_W) ? (wv != 0.0f) : (i >= tr.beg && i < tr.end);
        if (!live) continue;
        const float edt = cur.dt[e];
        const Taps f = warped_taps(cur.rs[e], cur.cs[e], cur.fx[e] - edt * fu[e], cur.fy[e] - edt * fv[e]);
        const int rl = f.R - oy, cl = f.C - ox;
        float g00, g10, g01, g11;
        if (f.ok && rl >= 0 && rl < LH - 1 && cl >= 0 && cl < LW - 1) {
          const float* p = &s_g[rl * LW + cl];
          g00 = p[0];
          g10 = p[LW];
          g01 = p[1];
          g11 = p[LW + 1];
        } else {
          const int R = f.R + pad_h, C = f.C + pad_w;
          g00 = f.ok ? G.at(R, C) : 0.0f;
          g10 = f.ok ? G.at(R + 1, C) : 0.0f;
          g01 = f.ok ? G.at(R, C + 1) : 0.0f;
          g11 = f.ok ? G.at(R + 1, C + 1) : 0.0f;
        }
        const float a = 1.0f - f.fr, b = 1.0f - f.fc;
        const float wl = HAS_W ? wv : 1.0f;
        const float dx = wl * (b * (g10 - g00) + f.fc * (g11 - g01));  // dL/dx'
        const float dy = wl * (a * (g01 - g00) + f.fr * (g11 - g10));  // dL/dy'
        if (d_weight) d_weight[i] = a * b * g00 + f.fr * b * g10 + a * f.fc * g01 + f.fr * f.fc * g11;
        if (UNIFORM) {
          ax += edt * dx;  // dL/dtheta0 += dt * dL/dx'
          ay += edt * dy;
          continue;
        }
        const int pix = (cur.rs[e] - tr0) * TW + (cur.cs[e] - tc0);
        if (pix != run_pix) {
          if (run_pix >= 0) {
            atomic_add(&s_d[run_pix], (double)ax);
            atomic_add(&s_d[TH * TW + run_pix], (double)ay);
          }
          run_pix = pix;
          ax = 0.0f;
          ay = 0.0f;
        }
        ax -= edt * dx;  // dL/dflow0[src] += -dt * dL/dx'
        ay -= edt * dy;
      }
      if (UNIFORM) {
        tot_x += (double)ax;
        tot_y += (double)ay;
      } else if (run_pix >= 0) {
        atomic_add(&s_d[run_pix], (double)ax);
        atomic_add(&s_d[TH * TW + run_pix], (double)ay);
      }
      cur = nxt;
      nxt = nn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        fu[e] = gu[e];
        fv[e] = gv[e];
      }
      grp += kBlock;
    }
  }
  __syncthreads();

  if (UNIFORM) {
    __shared__ double red[2 * kBlock / kWave];
    block_sum2(tot_x, tot_y, red);
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = tot_x;
      partials[2 * blockIdx.x + 1] = tot_y;
    }
    return;
  }
  const TileGrad<TH, TW> grad{fx, 1.0f / fx_scale, s_d};
  auto grad_at = [&](int i, int ch) -> float { return grad.at(i, ch); };
  if (GRID) {
    grid_tile_epilogue<TH, TW, HALO, false>(tr, tr0, tc0, H, W, s_d, s_g, s_flow, s_lerp, grad, addend, s_norm, s_tv,
                                            reg_partials ? reg_partials + blockIdx.x : nullptr,
                                            part_out + (int64_t)tr.slab * (2 * kGridCells * kGridCells));
    return;
  }
  if (part_out != nullptr) {  // partial tile [2][TH * TW] of this part, plain stores
    float* out = part_out + (int64_t)tr.slab * (kLHmax * kLWmax);
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) {
      out[i] = grad_at(i, 0);
      out[TH * TW + i] = grad_at(i, 1);
    }
    EBOS_STAMP_BWD(5);
    return;
  }
  // every flow pixel belongs to exactly one tile: coalesced stores, zeros where no event lives
#ifndef EBOS_PLAIN_SLABS
  if ((W & 3) == 0 && TW % 4 == 0) {
    // 16 bytes per lane, write-through: the 7.4 MB of a 1280x720 gradient do not wait for the end-of-kernel write-back
    const __amdgpu_buffer_rsrc_t dr = slab_rsrc(d_flow, 0xffffffffu);
    for (int i = threadIdx.x; i < TH * (TW / 4); i += kBlock) {
      const int rl = i / (TW / 4), cl = (i - rl * (TW / 4)) * 4;
      const int r = tr0 + rl, c = tc0 + cl;
      if (r < H && c < W) {
        const int64_t o = (int64_t)r * W + c;
        const int p0 = rl * TW + cl;
        float4 gx = make_float4(grad_at(p0, 0), grad_at(p0 + 1, 0), grad_at(p0 + 2, 0), grad_at(p0 + 3, 0));
        float4 gy = make_float4(grad_at(p0, 1), grad_at(p0 + 1, 1), grad_at(p0 + 2, 1), grad_at(p0 + 3, 1));
        if (addend) {
          const float4 ax = *reinterpret_cast<const float4*>(addend + o), ay = *reinterpret_cast<const float4*>(addend + hw + o);
          gx.x += ax.x, gx.y += ax.y, gx.z += ax.z, gx.w += ax.w;
          gy.x += ay.x, gy.y += ay.y, gy.z += ay.z, gy.w += ay.w;
        }
        slab_store4(dr, (unsigned)(o * 4), gx);
        slab_store4(dr, (unsigned)((hw + o) * 4), gy);
      }
    }
    EBOS_STAMP_BWD(5);
    return;
  }
#endif
  for (int i = threadIdx.x; i < TH * TW; i += kBlock) {
    const int rl = i / TW, cl = i - rl * TW;
    const int r = tr0 + rl, c = tc0 + cl;
    if (r < H && c < W) {
      const int64_t o = (int64_t)r * W + c;
      // addend: gradient of the flow regularisers, summed here instead of in a separate pass over [2, H, W]
      d_flow[o] = grad_at(i, 0) + (addend ? addend[o] : 0.0f);
      d_flow[hw + o] = grad_at(i, 1) + (addend ? addend[hw + o] : 0.0f);
    }
  }
  EBOS_STAMP_BWD(5);
}

__global__ void __launch_bounds__(256) theta_grad_finalize_kernel(const double* __restrict__ partials, int ntiles, float* d_theta) {
  double sx = 0.0, sy = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += blockDim.x) {
    sx += partials[2 * i];
    sy += partials[2 * i + 1];
  }
  __shared__ double red[8];
  block_sum2(sx, sy, red);
  if (threadIdx.x == 0) {
    d_theta[0] = (float)sx;
    d_theta[1] = (float)sy;
  }
}

// The 2-DoF Adam loop's last kernel (ebos_cmax_2dof_solve_f32): d loss / d theta from the tiles' partial pairs (the order of
// theta_grad_finalize_kernel), the loss of the iteration -- upstream * variance, for the parameters BEFORE the update, as torch's
// loop records it -- and torch.optim.Adam's step on the two parameters.  One workgroup.
__global__ void __launch_bounds__(256)
theta_adam_kernel(const double* __restrict__ partials, int ntiles, float* __restrict__ d_theta, float* __restrict__ theta,
                  float* __restrict__ m, float* __restrict__ v, double lr, double beta1, double beta2, double eps, int t,
                  int* __restrict__ step, const float* __restrict__ variance, const float* __restrict__ upstream,
                  float* __restrict__ losses, int losses_cap) {
  double sx = 0.0, sy = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += blockDim.x) {
    sx += partials[2 * i];
    sy += partials[2 * i + 1];
  }
  __shared__ double red[8];
  block_sum2(sx, sy, red);
  if (threadIdx.x == 0) {
    const AdamCoef coef = adam_coef(lr, beta1, beta2, t);
    const float g[2] = {(float)sx, (float)sy};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float mi = m[k], vi = v[k], th = theta[k];
      adam_update(g[k], mi, vi, th, coef.step_size, coef.bc2_sqrt, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
      m[k] = mi, v[k] = vi, theta[k] = th, d_theta[k] = g[k];
    }
    if (losses != nullptr && t - 1 < losses_cap) losses[t - 1] = upstream[0] * variance[0];
    step[0] = t;
  }
}

// adaptive backward, second step: d_flow of a tile = sum of its parts' partial tiles (+ the regulariser gradient)
template <int TH, int TW, int HALO>
__global__ void __launch_bounds__(256)
bwd_parts_combine_kernel(const float* __restrict__ parts, const int32_t* __restrict__ part_off, int tiles_x, int H, int W,
                         const float* __restrict__ addend, float* __restrict__ d_flow) {
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  static_assert(2 * TH * TW <= LH * LW, "a partial d_flow tile fits one slab");
  const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int s0 = part_off[tile], np = part_off[tile + 1] - s0;
  const int64_t hw = (int64_t)H * W;
  for (int i = threadIdx.x; i < TH * TW; i += blockDim.x) {
    const int rl = i / TW, cl = i - rl * TW;
    const int r = ty * TH + rl, c = tx * TW + cl;
    if (r >= H || c >= W) continue;
    float gx = 0.0f, gy = 0.0f;
    for (int p = 0; p < np; ++p) {
      const float* q = parts + (int64_t)(s0 + p) * (LH * LW);
      gx += q[i];
      gy += q[TH * TW + i];
    }
    const int64_t o = (int64_t)r * W + c;
    d_flow[o] = gx + (addend ? addend[o] : 0.0f);
    d_flow[hw + o] = gy + (addend ? addend[hw + o] : 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct SlabConfig {
  int th, tw, halo;
};
// f64 tile + halo must fit 160 KiB (forward); backward needs 16 TH TW + 4 LH LW bytes
// {45, 80, 32} cuts 720 x 1280 into exactly 16 x 16 = 256 tiles: one workgroup per CU of an MI355X
// smaller halos for windows whose displacements are small (BOS flows are typically a few pixels): slabs, LDS clear / decode and
// the backward kernel's upstream tile all shrink with (TH + 2 halo)(TW + 2 halo); taps beyond the halo stay correct (spill)
constexpr SlabConfig kSlabConfigs[] = {{64, 64, 32}, {45, 80, 32}, {32, 64, 32}, {32, 32, 32}, {64, 64, 16},
                                       {45, 80, 16}, {32, 32, 16}, {32, 32, 8}};
constexpr int kNumSlabConfigs = sizeof(kSlabConfigs) / sizeof(kSlabConfigs[0]);

struct SlabLayout {
  int tiles_y, tiles_x, nblk, h, w, combine_blocks;
  size_t slab_cells;   // per workgroup
  size_t off_spill, off_partials, off_epoch, off_halo, off_counters, total;
};

// `halo` arguments of the C ABI: h >= 0 is a built halo; EBOS_HALO_AUTO(max_halo, q) = -(max_halo + 256 q) asks for run-time
// windows per tile, at most max_halo (a built halo), with |dt| <= q / 64 for every event of the plan (ebos_hip.h)
struct HaloArg {
  int halo;        // the built configuration (the largest window)
  bool dyn;
  float dt_bound;
};
inline HaloArg decode_halo(int halo) {
  if (halo >= 0) return HaloArg{halo, false, 0.0f};
  const int a = -halo;
  return HaloArg{a & 255, true, (float)(a >> 8) / 64.0f};
}

constexpr int kAdaptiveItemsPerTile = 2;  // work items of an adaptive plan = 2 x tiles (ebos_plan_parts)

inline SlabLayout slab_layout(int H, int W, int th, int tw, int halo, int splits, int pad_h, int pad_w) {
  SlabLayout L;
  L.tiles_y = (H + th - 1) / th;
  L.tiles_x = (W + tw - 1) / tw;
  L.nblk = L.tiles_y * L.tiles_x * (splits == 0 ? kAdaptiveItemsPerTile : splits);  // splits == 0: adaptive work items
  L.h = H + 2 * pad_h;
  L.w = W + 2 * pad_w;
  L.slab_cells = (size_t)(th + 2 * halo) * (tw + 2 * halo);
  L.combine_blocks = ((L.w + kCombineBlock - 1) / kCombineBlock) * L.h;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.off_spill = align((size_t)L.nblk * L.slab_cells * sizeof(float));
  L.off_partials = L.off_spill + align((size_t)L.h * L.w * sizeof(float));
  L.off_epoch = L.off_partials + align((size_t)L.combine_blocks * 2 * sizeof(double));  // SpillEpoch word
  // counters of the combine pass's last-workgroup reduction (FinalizeIn): 128 bytes each, one per kFinalizeGroup workgroups + the top one
  L.off_counters = L.off_epoch + 256;
  L.off_halo = L.off_counters + align((size_t)(2 + (L.combine_blocks + kFinalizeGroup - 1) / kFinalizeGroup) * 128);  // [tiles] window (hr, hc) of each tile's slabs (run-time windows): the LAST section
  L.total = L.off_halo + align((size_t)L.tiles_y * L.tiles_x * sizeof(unsigned));
  return L;
}

// what the GRID kernels add to the LDS of their dense twins: the tile's flow, its row / column interpolation and the cell block
template <int TH, int TW, int AP>
constexpr size_t grid_lds_extra() {
  return (size_t)2 * (TH + 2 * AP) * (TW + 2 * AP) * sizeof(float) + (size_t)(TH + TW + 4 * AP) * sizeof(Lerp) +
         (size_t)2 * kGridCells * kGridCells * sizeof(float);
}

// LDS of the GRID accumulate kernel: accumulators + the tile's own flow; some tile configurations do not fit
template <int TH, int TW, int HALO>
constexpr bool grid_fwd_fits() {
  return (size_t)acc_cells<TH, TW, HALO, true>() * sizeof(double) + grid_lds_extra<TH, TW, 0>() + 1024 <= 160 * 1024;
}

template <int TH, int TW, int HALO>
constexpr size_t grid_bwd_lds() {
  return (size_t)2 * TH * TW * sizeof(double) + (size_t)(TH + 2 * HALO) * (TW + 2 * HALO) * sizeof(float) + grid_lds_extra<TH, TW, kBwdApron>();
}
template <int TH, int TW, int HALO>
constexpr bool grid_bwd_fits() {
  return grid_bwd_lds<TH, TW, HALO>() + 1024 <= 160 * 1024 &&
         (size_t)kGridCells * (3 * TW + TH) <= (size_t)(TH + 2 * HALO) * (TW + 2 * HALO);  // row sums + weights reuse the upstream tile
}

template <typename K>
int reserve_lds(K kern, size_t lds, const char* what) {
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    set_error("%s: cannot reserve %zu B of LDS", what, lds);
    return EBOS_ERR_LAUNCH;
  }
  return EBOS_OK;
}

}  // namespace
}  // namespace ebos
