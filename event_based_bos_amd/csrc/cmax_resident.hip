// cmax_resident.hip -- the contrast-maximisation inner loop as ONE resident launch (gfx950).
//
// The loop it runs is the reference's optimiser loop over ONE event window -- zero_grad -> objective -> backward -> Adam step,
// src/solver/generative_max_likelihood.py:306-341, 600 iterations in configs/hot_plate1.yaml:70 -- for the patch-flow objective
//     loss(theta) = -w * var(IWE(dense(theta))) + w_n * flow_norm(dense) + w_g * image_gradient(dense)
// (dense = patch grid -> per-pixel flow, src/solver/patch_eklt.py:173-204).  ebos_cmax_patch_solve_f32 enqueues it as four launches
// per iteration (accumulate, combine, backward, cell combine + Adam: solver_kernels.hip); at 2 M events those take 43 us of which
// 4.7 us are event loops -- the rest is what every launch re-derives (tile ranges, interpolation tables, LDS clears, grid cells,
// the variance partials of ~900 combine workgroups read back by 256 backward workgroups) and the launches' fill and drain.
//
// Here one 1024-thread workgroup per source tile (grid <= CUs, one per CU, co-residency checked on the host) stays resident for
// all n_iter iterations and keeps, across iterations:
//   registers  its tile range, its element of the block of grid cells the tile touches (theta, exp_avg, exp_avg_sq: every
//              workgroup steps the cells of its own block itself, redundantly and bit-identically -- no broadcast of theta),
//              which tiles' partial gradients each of its cells sums
//   LDS        the row / column interpolation tables of the tile (+ 2 px apron), the block of cells
// and exchanges per iteration, through global memory, only
//   S1  its LDS image as a slab (write-through) -> flag1 {epoch, window}: the EIGHT NEIGHBOURS are waited for, then every workgroup
//       sums the slabs over ITS OWN tile's pixels (the combine pass, restricted to what it owns; same order of additions: the image
//       has the bits of the four-launch pipeline) and stores its tile of the IWE
//   S2  (sum, sum of squares) of its tile + last iteration's regulariser partial as tagged 8-byte granules: the one all-to-all
//       of the iteration (mean of the IWE, loss bookkeeping); then the upstream window (tile + halo) is staged from the neighbours'
//       image tiles
//   S3  its <= 16 x 16 partial cell gradients -> flag3: the tiles whose partials its cells sum are waited for (<= 5 x 5), then Adam.
// Hand-off form (cdna guide, Guideline 16 / MI355X_MICROARCH visibility table, first row): every handed-off byte is an sc1
// (write-through) store, every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane
// stores the flag (sc1); consumers poll with sc1 loads and read the payload with sc1 loads only -- no fences, no atomics.
// Every spin is bounded: a wave that waits longer than the caller's cap (or sees the status word set) raises the status word
// and the whole grid leaves; theta and the optimiser state are written back only by a launch that completed, so the host can
// fall back to the four-launch pipeline from unchanged state (ebos_cmax_resident_status).  Taps beyond the LDS window (the
// spill path of the four-launch pipeline, global atomics) end the launch the same way: correct for any flow, fast for BOS-sized ones.
#include <algorithm>

#include "iwe_tile_core.h"

namespace ebos {
namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) float gf32;

__device__ __forceinline__ unsigned long long ld_sc1(const unsigned long long* p) {
  return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load((gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store((gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every storing wave, before the barrier behind which the flag is stored (inline asm: invisible to the pass that drops waits)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

enum ResidentStatus : unsigned {
  RES_OK = 0,
  RES_TIMEOUT = 1,   // a wait passed the caller's cap (a workgroup not resident, another resident launch interleaved, ...)
  RES_SPILL = 2,     // a tap left the largest LDS window: the four-launch pipeline handles such flows
  RES_GEOMETRY = 3,  // a cell sums more tiles than the kernel holds slots for (the host check should have refused)
};

// A whole wave polls: lane-wise predicate, true when every lane's holds.  Bounded: every 32 polls the status word and the clock
// (100 MHz) are looked at; false = the launch is over (status set by this wave or seen set).
template <typename Pred>
__device__ __forceinline__ bool wave_wait(Pred&& ready, unsigned* status, unsigned long long cap_ticks) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    if (__all(ready())) return true;
    if ((++spins & 31u) == 0u) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      const unsigned st = ld_sc1(status);
      if (st != RES_OK) return false;
      if (now - t0 > cap_ticks) {
        if ((threadIdx.x & (kWave - 1)) == 0) st_sc1(status, (unsigned)RES_TIMEOUT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// a double as two tagged 8-byte granules {tag, 32 bits}: the data is the flag (cdna guide, R2)
__device__ __forceinline__ void put_granules(unsigned long long* g, unsigned tag, double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  st_sc1(g, ((unsigned long long)tag << 32) | (b & 0xffffffffull));
  st_sc1(g + 1, ((unsigned long long)tag << 32) | (b >> 32));
}

constexpr int kRecGranules = 8;  // a record: (sum, sum of squares, regulariser partial of the previous iteration) = 6 granules, 64-byte stride
constexpr int kSpan = 4;         // candidate tiles per axis whose partial cell gradients a cell sums (patch_grad_combine_kernel's)

struct ResidentArgs {
  EvPtrs ev;
  const int32_t* key_offsets;
  int H, W, tiles_y, tiles_x;
  GridSrc gs;
  float *theta, *d_theta, *exp_avg, *exp_avg_sq;
  const float* theta_mask;
  int* step;
  float *iwe, *slabs, *cell_partials;
  unsigned* status;
  unsigned long long *flag1, *flag3, *rec2, *done;   // mailbox sections (zeroed before every launch)
  float* losses;
  int losses_cap, t0, n_iter;
  double lr, beta1, beta2, eps;
  float w_contrast, s_norm, s_tv;
  int omit;
  float dt_bound;
  float* variance;
  double* moments;
  unsigned long long cap_ticks;
};

// LDS of the kernel: the forward view (accumulators + the tile's flow) and the backward view (d_flow accumulators + upstream window +
// the tile's flow with its apron) overlay each other; the interpolation tables and the cell block follow and persist
template <int TH, int TW, int HALO>
constexpr size_t resident_union_bytes() {
  constexpr size_t fwd = (size_t)acc_cells<TH, TW, HALO, true>() * sizeof(double) + (size_t)2 * TH * TW * sizeof(float);
  constexpr size_t bwd = (size_t)2 * TH * TW * sizeof(double) + (size_t)(TH + 2 * HALO) * (TW + 2 * HALO) * sizeof(float) +
                         (size_t)2 * (TH + 2 * kBwdApron) * (TW + 2 * kBwdApron) * sizeof(float);
  return ((fwd > bwd ? fwd : bwd) + 15) & ~(size_t)15;
}
template <int TH, int TW, int HALO>
constexpr size_t resident_lds_bytes() {
  return resident_union_bytes<TH, TW, HALO>() + (size_t)(TH + TW + 4 * kBwdApron) * sizeof(Lerp) +
         (size_t)2 * kGridCells * kGridCells * sizeof(float);
}
template <int TH, int TW, int HALO>
constexpr bool resident_fits() {
  return resident_lds_bytes<TH, TW, HALO>() + 2048 <= 160 * 1024 && HALO <= TH && HALO <= TW && TW % 4 == 0 &&
         grid_bwd_fits<TH, TW, HALO>();
}

// the tile's dense flow (+ AP px apron) from the cell block in LDS: tile_grid_finish's second half on resident tables
template <int TH, int TW, int AP>
__device__ __forceinline__ void tile_flow_from_cells(const Lerp* s_rows, const Lerp* s_cols, const float* s_cells, int gi0, int gj0,
                                                     float* s_flow) {
  constexpr int PH = TH + 2 * AP, PW = TW + 2 * AP;
  for (int i = threadIdx.x; i < PH * PW; i += kBlock) {
    const int rl = i / PW, cl = i - rl * PW;
    Lerp ly = s_rows[rl], lx = s_cols[cl];
    lx.i0 -= gj0;
    lx.i1 -= gj0;
    const float* u0 = s_cells + (ly.i0 - gi0) * kGridCells;
    const float* u1 = s_cells + (ly.i1 - gi0) * kGridCells;
    s_flow[i] = grid_bilerp(u0, u1, ly, lx);
    s_flow[PH * PW + i] = grid_bilerp(u0 + kGridCells * kGridCells, u1 + kGridCells * kGridCells, ly, lx);
  }
}

template <int TH, int TW, int HALO>
__global__ void __launch_bounds__(kBlock) cmax_resident_kernel(ResidentArgs a) {
  constexpr int kLHmax = TH + 2 * HALO, kLWmax = TW + 2 * HALO;
  constexpr int kCells = acc_cells<TH, TW, HALO, true>();
  constexpr int AP = kBwdApron, PH = TH + 2 * AP, PW = TW + 2 * AP;
  constexpr int kWaves = kBlock / kWave;
  static_assert(HALO <= TH && HALO <= TW, "only the eight neighbours' windows reach a tile");
  static_assert(kCells % 2 == 0 && TW % 4 == 0, "16-byte LDS clears and slab quads");
  extern __shared__ __attribute__((aligned(16))) double s_raw[];
  double* s_acc = s_raw;                                               // forward: LDS image of the tile's window
  float* s_flow_f = reinterpret_cast<float*>(s_acc + kCells);           //          [2][TH * TW] flow of the tile
  double* s_d = s_raw;                                                 // backward: [2][TH * TW] d_flow accumulators
  float* s_g = reinterpret_cast<float*>(s_raw + 2 * TH * TW);           //           [LH][LW] upstream window
  float* s_flow_b = s_g + kLHmax * kLWmax;                              //           [2][PH][PW] flow of tile + apron
  Lerp* s_lerp = reinterpret_cast<Lerp*>(reinterpret_cast<char*>(s_raw) + resident_union_bytes<TH, TW, HALO>());  // [PH + PW]
  float* s_cells = reinterpret_cast<float*>(s_lerp + PH + PW);         // [2][kGridCells][kGridCells]
  __shared__ TileShared sh;
  __shared__ int s_spill, s_bad, s_ok;
  __shared__ unsigned s_next;
  __shared__ float s_gmax[2 * kWaves];
  __shared__ unsigned s_win[9];
  __shared__ double s_mom[4];    // mean, variance, sum of the regulariser partials of the previous iteration
  __shared__ double s_reg;       // this tile's regulariser value partial
  __shared__ float s_adam[2];    // step size and sqrt(bias correction 2) of the iteration's Adam step
  __shared__ double s_red[3 * kWaves];

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int n_tiles = a.tiles_y * a.tiles_x;
  const int tile = blockIdx.x, ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
  const int tr0 = ty * TH, tc0 = tx * TW, H = a.H, W = a.W;
  const TileRange tr = tile_range<FMT_COMPACT>(a.key_offsets, a.ev, TH * TW, a.tiles_x, 1);
  const Axis ay = a.gs.ay, ax = a.gs.ax;

  // ---- once: interpolation tables of tile + apron, the block of cells they touch, this thread's element of it -------------------
  for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  for (int i = threadIdx.x; i < PH + PW; i += kBlock)
    s_lerp[i] = i < PH ? lerp_at(ay, min(max(tr0 + i - AP, 0), H - 1)) : lerp_at(ax, min(max(tc0 + i - PH - AP, 0), W - 1));
  // events per source pixel (the backward scatter's fixed-point unit): of the plan, not of the iteration
  {
    int nmax_t = 1;
    const int32_t* ko = a.key_offsets + (int64_t)tile * (TH * TW);
#pragma unroll
    for (int k = 0; k < (TH * TW + kBlock - 1) / kBlock; ++k) {
      const int i = min((int)threadIdx.x + k * kBlock, TH * TW - 1);
      nmax_t = max(nmax_t, ko[i + 1] - ko[i]);
    }
    const float nm = wave_max_nonneg((float)nmax_t);
    if (lane == 0) s_gmax[kWaves + wave] = nm;
  }
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  const int gi0 = s_lerp[0].i0, ni = s_lerp[PH - 1].i1 - gi0 + 1;
  const int gj0 = s_lerp[PH].i0, nj = s_lerp[PH + PW - 1].i1 - gj0 + 1;
  const bool has = (int)threadIdx.x < 2 * ni * nj;  // this thread holds element (ch, gi0 + ci, gj0 + cj) of the cell block
  const int e_ = has ? (int)threadIdx.x : 0;
  const int ch = e_ / (ni * nj), ci = (e_ - ch * (ni * nj)) / nj, cj = e_ - ch * (ni * nj) - ci * nj;
  const int gi = gi0 + ci, gj = gj0 + cj;
  const int64_t gidx = ((int64_t)ch * ay.g + gi) * ax.g + gj;
  float th_e = 0.0f, m_e = 0.0f, v_e = 0.0f, g_e = 0.0f, mask_e = 1.0f;
  if (has) {
    th_e = a.theta[gidx], m_e = a.exp_avg[gidx], v_e = a.exp_avg_sq[gidx];
    if (a.theta_mask != nullptr) mask_e = a.theta_mask[(int64_t)gi * ax.g + gj];
  }
  // which tiles' partial cell gradients this cell sums (the arithmetic of patch_grad_combine_kernel, flow_upsample.hip): <= kSpan
  // candidate tiles per axis from the cell's conservative pixel support; a candidate counts if its own cell block holds the cell
  int cand_ty0 = 0, cand_tx0 = 0;
  unsigned cand_y = 0, cand_x = 0;  // per candidate k: bit 4 k + 3 = valid, bits 4 k .. 4 k + 2 ... (index of the cell in that tile's block: 4 bits)
  unsigned cand_yv = 0, cand_xv = 0;
  bool owner = false;
  {
    int r_lo, r_hi, c_lo, c_hi;
    support(ay, gi, H, &r_lo, &r_hi);
    support(ax, gj, W, &c_lo, &c_hi);
    cand_ty0 = r_lo / TH, cand_tx0 = c_lo / TW;
    const int ty_n = r_lo < r_hi ? (r_hi - 1) / TH - cand_ty0 + 1 : 0, tx_n = c_lo < c_hi ? (c_hi - 1) / TW - cand_tx0 + 1 : 0;
    if (has && (ty_n > kSpan || tx_n > kSpan) && lane == 0) st_sc1(a.status, (unsigned)RES_GEOMETRY);
    int first_ty = -1, first_tx = -1;
#pragma unroll
    for (int k = 0; k < kSpan; ++k) {
      const int cty = min(cand_ty0 + k, a.tiles_y - 1), ctx = min(cand_tx0 + k, a.tiles_x - 1);
      const int bi0 = lerp_at(ay, cty * TH).i0, bi1 = lerp_at(ay, min(cty * TH + TH, H) - 1).i1;
      const int bj0 = lerp_at(ax, ctx * TW).i0, bj1 = lerp_at(ax, min(ctx * TW + TW, W) - 1).i1;
      const bool oky = k < ty_n && gi >= bi0 && gi <= bi1, okx = k < tx_n && gj >= bj0 && gj <= bj1;
      cand_y |= (unsigned)(oky ? gi - bi0 : 0) << (4 * k);
      cand_x |= (unsigned)(okx ? gj - bj0 : 0) << (4 * k);
      cand_yv |= (unsigned)oky << k;
      cand_xv |= (unsigned)okx << k;
      if (oky && first_ty < 0) first_ty = cty;
      if (okx && first_tx < 0) first_tx = ctx;
    }
    owner = has && first_ty == ty && first_tx == tx;  // the first tile that holds a cell writes it back at the end
  }
  // the tiles whose partials any cell of this block sums: a rectangle of tiles (<= 64, host-checked), waited for at S3
  int rect_ty0, rect_tx0, rect_ny, rect_nx;
  {
    int lo, hi, dummy;
    support(ay, gi0, H, &lo, &dummy);
    support(ay, gi0 + ni - 1, H, &dummy, &hi);
    rect_ty0 = lo / TH;
    rect_ny = lo < hi ? min((hi - 1) / TH, a.tiles_y - 1) - rect_ty0 + 1 : 0;
    support(ax, gj0, W, &lo, &dummy);
    support(ax, gj0 + nj - 1, W, &dummy, &hi);
    rect_tx0 = lo / TW;
    rect_nx = lo < hi ? min((hi - 1) / TW, a.tiles_x - 1) - rect_tx0 + 1 : 0;
    if (rect_ny * rect_nx > kWave && threadIdx.x == 0) st_sc1(a.status, (unsigned)RES_GEOMETRY);
  }
  const int lo_px = a.omit ? 1 : 0;
  const double n_px = (double)max(H - 2 * lo_px, 0) * (double)max(W - 2 * lo_px, 0);
  const bool vec_store = (W & 3) == 0;
  const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(a.slabs, 0xffffffffu);
  const __amdgpu_buffer_rsrc_t iwe_rsrc = slab_rsrc(a.iwe, 0xffffffffu);
  double reg_prev = 0.0;   // thread 0: this tile's regulariser partial of the previous iteration
  float var_prev = 0.0f;   // workgroup 0, thread 0: the variance of the previous iteration (its loss is recorded one iteration late)
  bool done_ok = true;

  for (int it = 0; it < a.n_iter; ++it) {
    const unsigned ep = (unsigned)it + 1u;
    // ---- F0: cells -> LDS, the tile's window from a bound on its displacements, the tile's flow ------------------------------
    if (has) s_cells[(ch * kGridCells + ci) * kGridCells + cj] = th_e;
    tile_bound_post(has && ch == 0 ? fabsf(th_e) : 0.0f, has && ch == 1 ? fabsf(th_e) : 0.0f, sh.bound);
    if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
      sh.next = 2 * kWaves;
      sh.chk = 0ull;
      s_spill = 0;
      s_bad = 0;
      s_next = 2 * kWaves;
      // Adam's bias corrections of step t, as torch computes them (host double in the four-launch pipeline: make_adam_job)
      const AdamCoef coef = adam_coef(a.lr, a.beta1, a.beta2, a.t0 + it + 1);
      s_adam[0] = coef.step_size;
      s_adam[1] = coef.bc2_sqrt;
    }
    __syncthreads();
    const Win<TH, TW, HALO, true> win = tile_bound_read<TH, TW, HALO, true>(sh.bound, a.dt_bound);
    tile_flow_from_cells<TH, TW, 0>(s_lerp + AP, s_lerp + PH + AP, s_cells, gi0, gj0, s_flow_f);
    __syncthreads();
    // ---- F1: events -> LDS image -> slab (write-through) ------------------------------------------------------------------------
    tile_body<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, true, false>(tr, win, s_flow_f, s_acc, sh, a.ev, H, W, a.tiles_x, 0, 0,
                                                                                  a.slabs, nullptr, nullptr, 0u, nullptr, nullptr, NoHook{});
    drain_stores();
    __syncthreads();
    if (sh.flag[1]) {  // (uniform) a tap left the largest window: the four-launch pipeline's spill path handles that flow
      if (threadIdx.x == 0) st_sc1(a.status, (unsigned)RES_SPILL);
      done_ok = false;
      break;
    }
    if (threadIdx.x == 0) st_sc1(a.flag1 + tile, ((unsigned long long)ep << 32) | win_pack(win.hr, win.hc));
    // ---- S1: the eight neighbours' slabs (and their windows) --------------------------------------------------------------------
    if (wave == 0) {
      const int nty = ty + lane / 3 - 1, ntx = tx + lane % 3 - 1;
      const bool nb = lane < 9 && nty >= 0 && nty < a.tiles_y && ntx >= 0 && ntx < a.tiles_x;
      unsigned wv = 0xffffffffu;
      const bool ok = wave_wait([&]() {
        if (!nb) return true;
        const unsigned long long f = ld_sc1(a.flag1 + nty * a.tiles_x + ntx);
        wv = (unsigned)f;
        return (unsigned)(f >> 32) >= ep;
      }, a.status, a.cap_ticks);
      if (lane < 9) s_win[lane] = nb ? wv : 0xffffffffu;
      if (lane == 0 && !ok) s_ok = 0;
    }
    __syncthreads();
    if (!s_ok) { done_ok = false; break; }
    // ---- G: this tile's pixels of the IWE = sum of the slabs whose windows reach them, in the combine pass's order ---------------
    double sm = 0.0, sq = 0.0;
    for (int q = threadIdx.x; q < TH * (TW / 4); q += kBlock) {
      const int rl = q / (TW / 4), cl = (q - rl * (TW / 4)) * 4;
      const int r = tr0 + rl, c = tc0 + cl;
      if (r >= H || c >= W) continue;
      float4 part[9];
      bool okk[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const unsigned w = s_win[k];
        const int nty = ty + k / 3 - 1, ntx = tx + k % 3 - 1;
        const int hr = (int)(w & 255u), hc = (int)((w >> 8) & 255u);
        const int rr = r - (nty * TH - hr), cc = c - (ntx * TW - hc), lw = TW + 2 * hc;
        okk[k] = w != 0xffffffffu && (unsigned)rr < (unsigned)(TH + 2 * hr) && (unsigned)cc < (unsigned)lw;
        const unsigned byte = okk[k] ? ((unsigned)(nty * a.tiles_x + ntx) * (unsigned)(kLHmax * kLWmax) + (unsigned)(rr * lw + cc)) * 4u : 0u;
        part[k] = slab_load4(all_slabs, byte);
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (okk[k]) v.x += part[k].x, v.y += part[k].y, v.z += part[k].z, v.w += part[k].w;
      const float e4[4] = {v.x, v.y, v.z, v.w};
      const int64_t gi_px = (int64_t)r * W + c;
      if (vec_store) {
        slab_store4(iwe_rsrc, (unsigned)(gi_px * 4), v);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < W) st_sc1(a.iwe + gi_px + k, e4[k]);
      }
      if (r >= lo_px && r < H - lo_px) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k >= lo_px && c + k < W - lo_px) {
            sm += (double)e4[k];
            sq += (double)e4[k] * (double)e4[k];
          }
      }
    }
    drain_stores();
    block_sum2(sm, sq, s_red);  // (its barriers stand behind every wave's drain)
    if (threadIdx.x == 0) {
      unsigned long long* rec = a.rec2 + ((size_t)(it & 1) * n_tiles + tile) * kRecGranules;
      put_granules(rec, ep, sm);
      put_granules(rec + 2, ep, sq);
      put_granules(rec + 4, ep, reg_prev);
    }
    // ---- S2: the one all-to-all: every tile's (sum, sum of squares, regulariser partial of the previous iteration) ---------------
    double as = 0.0, aq = 0.0, ar = 0.0;
    if (wave * kWave < n_tiles) {
      const int k = wave * kWave + lane;
      const unsigned long long* rec = a.rec2 + ((size_t)(it & 1) * n_tiles + min(k, n_tiles - 1)) * kRecGranules;
      unsigned long long g[6];
      const bool ok = wave_wait([&]() {
        bool all = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          g[j] = ld_sc1(rec + j);
          all = all && (unsigned)(g[j] >> 32) == ep;
        }
        return all;
      }, a.status, a.cap_ticks);
      if (ok && k < n_tiles) {
        as = __builtin_bit_cast(double, (g[0] & 0xffffffffull) | (g[1] << 32));
        aq = __builtin_bit_cast(double, (g[2] & 0xffffffffull) | (g[3] << 32));
        ar = __builtin_bit_cast(double, (g[4] & 0xffffffffull) | (g[5] << 32));
      }
      if (lane == 0 && !ok) s_ok = 0;
    }
    {  // (three sums at once; block_sum2's order per sum)
      as = wave_sum(as), aq = wave_sum(aq), ar = wave_sum(ar);
      if (lane == 0) s_red[wave] = as, s_red[kWaves + wave] = aq, s_red[2 * kWaves + wave] = ar;
      __syncthreads();
      if (threadIdx.x == 0) {
        double S = 0.0, Q = 0.0, R = 0.0;
        for (int k = 0; k < kWaves; ++k) S += s_red[k], Q += s_red[kWaves + k], R += s_red[2 * kWaves + k];
        const double mean = n_px > 0.0 ? S / n_px : 0.0;
        s_mom[0] = mean;
        s_mom[1] = (Q - S * mean) / (n_px - 1.0);
        s_mom[2] = R;
      }
      __syncthreads();
    }
    if (!s_ok) { done_ok = false; break; }
    const double mean = s_mom[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // bookkeeping: the loss of the PREVIOUS iteration is complete now
      const float var_f = (float)s_mom[1];
      if (it > 0 && a.losses != nullptr && a.t0 + it - 1 < a.losses_cap)
        a.losses[a.t0 + it - 1] = (float)(-(double)a.w_contrast * (double)var_prev + s_mom[2]);
      var_prev = var_f;
      a.variance[0] = var_f;
      a.moments[0] = mean;
      a.moments[1] = n_px;
    }
    // ---- B0: upstream window (tile + halo) of d loss / d IWE = 2 (-w) (IWE - mean) / (M - 1), from the tiles' images ------------
    // (the window of the UPSTREAM image: the four-launch backward kernel stages at least its speculative 4 px window, and the
    // fixed-point unit of the scatter follows max |staged value| -- same window, same unit, same bits)
    constexpr int kSpecHalo = HALO < 4 ? HALO : 4;
    const Win<TH, TW, HALO, true> wb = (win.hr <= kSpecHalo && win.hc <= kSpecHalo) ? Win<TH, TW, HALO, true>{kSpecHalo, kSpecHalo} : win;
    const int LW = wb.LW(), n_win = wb.LH() * LW, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
    GradImage G;
    G.g = a.iwe;
    const double ga = 2.0 * (-(double)a.w_contrast) / (n_px - 1.0);
    G.a = (float)ga;
    G.c = (float)(-ga * mean);
    G.h = H, G.w = W, G.lo = lo_px;
    constexpr int kStage = (kLHmax * kLWmax + kBlock - 1) / kBlock;
    float raw[kStage];
    const float inv_lw = 1.0f / (float)LW;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      if (k * kBlock >= n_win) break;  // (uniform)
      const int i = min((int)threadIdx.x + k * kBlock, n_win - 1);
      const int rl = (int)(((float)i + 0.5f) * inv_lw), cl = i - rl * LW;
      const int R = min(max(oy + rl, 0), H - 1), C = min(max(ox + cl, 0), W - 1);
      raw[k] = ld_sc1(a.iwe + (int64_t)R * W + C);
    }
    for (int i = threadIdx.x; i < TH * TW; i += kBlock) reinterpret_cast<double2*>(s_d)[i] = make_double2(0.0, 0.0);  // [2][TH * TW]
    tile_flow_from_cells<TH, TW, AP>(s_lerp, s_lerp + PH, s_cells, gi0, gj0, s_flow_b);
    float gmax_t = 0.0f;
#pragma unroll
    for (int k = 0; k < kStage; ++k) {
      if (k * kBlock >= n_win) break;
      const int i = threadIdx.x + k * kBlock;
      const int rl = (int)(((float)i + 0.5f) * inv_lw), cl = i - rl * LW;
      const int R = oy + rl, C = ox + cl;
      const bool valid = R >= G.lo && R < G.h - G.lo && C >= G.lo && C < G.w - G.lo;
      const float gv = valid ? G.a * raw[k] + G.c : 0.0f;
      if (i < n_win) {
        s_g[i] = gv;
        gmax_t = fmaxf(gmax_t, gv == gv ? fabsf(gv) : INFINITY);
      }
    }
    gmax_t = wave_max_nonneg(gmax_t);
    if (lane == 0) s_gmax[wave] = gmax_t;
    __syncthreads();
    // ---- B1: the sweep: d loss / d flow of the tile's pixels, in LDS -----------------------------------------------------------------
    const FxUnit unit = bwd_fx_unit(s_gmax, a.dt_bound);
    double tot_x = 0.0, tot_y = 0.0;
    const BwdShared bsh{&s_spill, &s_bad, &s_next};
    const bool fx = bwd_lean_sweeps<TH, TW, HALO, false, true, true>(tr, s_d, s_g, a.ev, s_flow_b, H, W, 0, 0, G, tot_x, tot_y,
                                                                    ChunkQueue{&s_next}, wb, unit, a.dt_bound, BwdPre{}, false, bsh, NoHook{});
    __syncthreads();
    // ---- B2: regularisers on the tile's flow, adjoint of grid -> dense on the tile -> partial cell gradients (write-through) ------
    const TileGrad<TH, TW> grad{fx, 1.0f / unit.scale, s_d};
    const bool any_reg = a.s_norm != 0.0f || a.s_tv != 0.0f;
    grid_tile_epilogue<TH, TW, HALO, true>(tr, tr0, tc0, H, W, s_d, s_g, s_flow_b, s_lerp, grad, nullptr, a.s_norm, a.s_tv,
                                           any_reg ? &s_reg : nullptr, a.cell_partials + (int64_t)tile * (2 * kGridCells * kGridCells));
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      st_sc1(a.flag3 + tile, (unsigned long long)ep);
      reg_prev = any_reg ? s_reg : 0.0;
    }
    // ---- S3: the partials of the tiles this block's cells sum; meanwhile the other waves clear the LDS image for the next pass ---
    if (wave == 0) {
      const int k = lane, ry = k / max(rect_nx, 1), rx = k - ry * max(rect_nx, 1);
      const bool act = k < rect_ny * rect_nx;
      const unsigned long long* f = a.flag3 + (rect_ty0 + (act ? ry : 0)) * a.tiles_x + rect_tx0 + (act ? rx : 0);
      const bool ok = wave_wait([&]() { return !act || ld_sc1(f) >= (unsigned long long)ep; }, a.status, a.cap_ticks);
      if (lane == 0 && !ok) s_ok = 0;
    } else {
      for (int i = threadIdx.x - kWave; i < kCells / 2; i += kBlock - kWave) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    if (!s_ok) { done_ok = false; break; }
    // ---- A: d loss / d theta of this thread's cell element = sum of the partials of the tiles that hold it; Adam ------------------
    if (has) {
      float pv[kSpan][kSpan];
#pragma unroll
      for (int p = 0; p < kSpan; ++p)
#pragma unroll
        for (int q = 0; q < kSpan; ++q) {
          const int cty = min(cand_ty0 + p, a.tiles_y - 1), ctx = min(cand_tx0 + q, a.tiles_x - 1);
          const int li = (int)((cand_y >> (4 * p)) & 15u), lj = (int)((cand_x >> (4 * q)) & 15u);
          pv[p][q] = ld_sc1(a.cell_partials + (((int64_t)(cty * a.tiles_x + ctx) * 2 + ch) * kGridCells + li) * kGridCells + lj);
        }
      float g = 0.0f;
#pragma unroll
      for (int p = 0; p < kSpan; ++p)
#pragma unroll
        for (int q = 0; q < kSpan; ++q) g += (((cand_yv >> p) & 1u) && ((cand_xv >> q) & 1u)) ? pv[p][q] : 0.0f;
      if (a.theta_mask != nullptr) g *= mask_e;
      g_e = g;
      adam_update(g, m_e, v_e, th_e, s_adam[0], s_adam[1], (float)a.beta2, (float)(1.0 - a.beta1), (float)(1.0 - a.beta2), (float)a.eps);
    }
  }
  if (!done_ok) return;  // (uniform) nothing of the optimiser state was written: the host falls back from unchanged state

  // ---- the state goes back: every cell element by the first tile that holds it ------------------------------------------------------
  if (owner) {
    a.theta[gidx] = th_e;
    a.exp_avg[gidx] = m_e;
    a.exp_avg_sq[gidx] = v_e;
    a.d_theta[gidx] = g_e;
  }
  if (a.n_iter <= 0) return;
  // the last iteration's loss: its regulariser partials travel through the `done` granules; workgroup 0 gathers them
  if (threadIdx.x == 0) put_granules(a.done + (size_t)tile * 2, (unsigned)a.n_iter, reg_prev);
  if (blockIdx.x != 0) return;
  double ar = 0.0;
  if (wave * kWave < n_tiles) {
    const int k = wave * kWave + lane;
    const unsigned long long* rec = a.done + (size_t)min(k, n_tiles - 1) * 2;
    unsigned long long g0 = 0, g1 = 0;
    const bool ok = wave_wait([&]() {
      g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1);
      return (unsigned)(g0 >> 32) == (unsigned)a.n_iter && (unsigned)(g1 >> 32) == (unsigned)a.n_iter;
    }, a.status, a.cap_ticks);
    if (ok && k < n_tiles) ar = __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
  }
  ar = block_sum(ar, s_red);
  if (threadIdx.x == 0) {
    const int t_last = a.t0 + a.n_iter - 1;
    if (a.losses != nullptr && t_last < a.losses_cap) a.losses[t_last] = (float)(-(double)a.w_contrast * (double)var_prev + ar);
    a.step[0] = a.t0 + a.n_iter;
  }
}

struct MailboxLayout {
  size_t off_status, off_flag1, off_flag3, off_rec2, off_done, total;
};
inline MailboxLayout mailbox_layout(int n_tiles) {
  MailboxLayout m;
  m.off_status = 0;
  m.off_flag1 = 256;
  m.off_flag3 = m.off_flag1 + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
  m.off_rec2 = m.off_flag3 + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
  m.off_done = m.off_rec2 + (size_t)2 * n_tiles * kRecGranules * 8;
  m.total = m.off_done + (((size_t)n_tiles * 16 + 255) & ~(size_t)255);
  return m;
}

// resident launches of different streams must not interleave their workgroups (two half-resident grids would wait for each other
// until their caps): each one waits for the previous one's end on its device
inline int order_resident_launches(hipStream_t s, bool after_launch) {
  constexpr int kMaxDevices = 64;
  static hipEvent_t last[kMaxDevices] = {};
  static std::atomic_flag lock = ATOMIC_FLAG_INIT;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return EBOS_ERR_LAUNCH;
  while (lock.test_and_set(std::memory_order_acquire)) {}
  int rc = EBOS_OK;
  if (!after_launch) {
    if (last[dev] != nullptr && hipStreamWaitEvent(s, last[dev], 0) != hipSuccess) rc = EBOS_ERR_LAUNCH;
  } else {
    if (last[dev] == nullptr && hipEventCreateWithFlags(&last[dev], hipEventDisableTiming) != hipSuccess) last[dev] = nullptr;
    if (last[dev] == nullptr || hipEventRecord(last[dev], s) != hipSuccess) rc = EBOS_ERR_LAUNCH;
  }
  lock.clear(std::memory_order_release);
  return rc;
}

template <int TH, int TW, int HALO>
int launch_resident(const ResidentArgs& a, void* mailbox, size_t mailbox_total, hipStream_t s) {
  if constexpr (resident_fits<TH, TW, HALO>()) {
    auto k = cmax_resident_kernel<TH, TW, HALO>;
    constexpr size_t lds = resident_lds_bytes<TH, TW, HALO>();
    if (int rc = reserve_lds(k, lds, "ebos_cmax_patch_solve_resident")) return rc;
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kBlock, lds) != hipSuccess) {
      set_error("ebos_cmax_patch_solve_resident: cannot query the device's occupancy");
      return EBOS_ERR_LAUNCH;
    }
    const int n_tiles = a.tiles_y * a.tiles_x;
    if (per_cu < 1 || n_tiles > n_cu * per_cu) {
      set_error("ebos_cmax_patch_solve_resident: %d workgroups cannot be co-resident (%d CUs x %d)", n_tiles, n_cu, per_cu);
      return EBOS_ERR_UNSUPPORTED;
    }
    if (int rc = order_resident_launches(s, false)) return rc;
    // (cells no tile's block holds keep a zero gradient, as the four-launch pipeline reports them)
    if (hipMemsetAsync(a.d_theta, 0, (size_t)2 * a.gs.ay.g * a.gs.ax.g * sizeof(float), s) != hipSuccess ||
        hipMemsetAsync(mailbox, 0, mailbox_total, s) != hipSuccess) {
      set_error("ebos_cmax_patch_solve_resident: cannot clear the mailbox");
      return EBOS_ERR_LAUNCH;
    }
    k<<<dim3((unsigned)n_tiles), dim3(kBlock), lds, s>>>(a);
    return order_resident_launches(s, true);
  } else {
    set_error("ebos_cmax_patch_solve_resident: no resident kernel for tile %dx%d halo %d", TH, TW, HALO);
    return EBOS_ERR_UNSUPPORTED;
  }
}

// the geometry / objective a resident launch takes; reason in ebos_last_error otherwise
bool resident_problem_ok(const ebos_cmax_patch_problem* q) {
  const int halo = decode_halo(q->halo).halo;
  if (q->grad_partials == nullptr || !q->grp_offsets || !q->cpix || !q->cdt) {
    set_error("resident solve: needs the grid-sampling route (grad_partials) on a compact plan");
    return false;
  }
  if (q->splits != 1 || q->pad_h != 0 || q->pad_w != 0) {
    set_error("resident solve: one work item per tile and no image padding (splits = %d, pad %dx%d)", q->splits, q->pad_h, q->pad_w);
    return false;
  }
  if (q->w_gradient_magnitude != 0.0f || q->w_variance == 0.0f) {
    set_error("resident solve: the variance contrast only");
    return false;
  }
  if (!ebos_patch_fused_supported(q->tile_h, q->tile_w, q->halo, q->slide_h, q->slide_w)) {
    set_error("resident solve: tile %dx%d halo %d / sliding window %dx%d is outside ebos_patch_fused_supported", q->tile_h, q->tile_w,
              halo, q->slide_h, q->slide_w);
    return false;
  }
  const bool built = (q->tile_h == 45 && q->tile_w == 80 && halo == 32) || (q->tile_h == 32 && q->tile_w == 32 && halo == 32);
  if (!built) {
    set_error("resident solve: no resident kernel built for tile %dx%d halo %d", q->tile_h, q->tile_w, halo);
    return false;
  }
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w;
  if (tiles_y * tiles_x > kBlock) {
    set_error("resident solve: %d tiles (one record per thread: <= %d)", tiles_y * tiles_x, kBlock);
    return false;
  }
  // every cell sums at most kSpan x kSpan tiles, every tile's block waits for at most 64 tiles
  const Axis ay = make_axis(q->gh, q->patch_h, q->slide_h, q->H), ax = make_axis(q->gw, q->patch_w, q->slide_w, q->W);
  if (ay.off < 0 || ax.off < 0) {
    set_error("resident solve: image larger than the resized grid");
    return false;
  }
  int span_y = 0, span_x = 0;
  for (int gi = 0; gi < q->gh; ++gi) {
    int lo, hi;
    support(ay, gi, q->H, &lo, &hi);
    if (lo < hi) span_y = std::max(span_y, (hi - 1) / q->tile_h - lo / q->tile_h + 1);
  }
  for (int gj = 0; gj < q->gw; ++gj) {
    int lo, hi;
    support(ax, gj, q->W, &lo, &hi);
    if (lo < hi) span_x = std::max(span_x, (hi - 1) / q->tile_w - lo / q->tile_w + 1);
  }
  if (span_y > kSpan || span_x > kSpan) {
    set_error("resident solve: a grid cell's support spans %dx%d tiles (<= %d per axis)", span_y, span_x, kSpan);
    return false;
  }
  // ... and the cells of a tile's block (tile + apron) sum at most 64 tiles in all (one lane of the waiting wave each)
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx) {
      const int r0 = std::max(ty * q->tile_h - kBwdApron, 0), r1 = std::min(ty * q->tile_h + q->tile_h + kBwdApron, q->H) - 1;
      const int c0 = std::max(tx * q->tile_w - kBwdApron, 0), c1 = std::min(tx * q->tile_w + q->tile_w + kBwdApron, q->W) - 1;
      int lo, hi, d;
      support(ay, lerp_at(ay, r0).i0, q->H, &lo, &d);
      support(ay, lerp_at(ay, std::max(r1, r0)).i1, q->H, &d, &hi);
      const int ny = lo < hi ? std::min((hi - 1) / q->tile_h, tiles_y - 1) - lo / q->tile_h + 1 : 0;
      support(ax, lerp_at(ax, c0).i0, q->W, &lo, &d);
      support(ax, lerp_at(ax, std::max(c1, c0)).i1, q->W, &d, &hi);
      const int nx = lo < hi ? std::min((hi - 1) / q->tile_w, tiles_x - 1) - lo / q->tile_w + 1 : 0;
      if (ny * nx > kWave) {
        set_error("resident solve: the cell block of tile (%d, %d) sums %dx%d tiles (<= %d in all)", ty, tx, ny, nx, kWave);
        return false;
      }
    }
  return true;
}

}  // namespace
}  // namespace ebos

extern "C" {

size_t ebos_cmax_resident_mailbox_bytes(int H, int W, int tile_h, int tile_w) {
  if (H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0) return 0;
  return ebos::mailbox_layout(((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w)).total;
}

int ebos_cmax_resident_supported(const ebos_cmax_patch_problem* q) {
  using namespace ebos;
  if (q == nullptr) return 0;
  return resident_problem_ok(q) ? 1 : 0;
}

int ebos_cmax_patch_solve_resident_f32(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, size_t mailbox_bytes,
                                       double spin_timeout_s, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(q != nullptr && n_iter >= 0 && q->steps_done >= 0, "ebos_cmax_patch_solve_resident: NULL problem or negative counts");
  EBOS_REQUIRE(q->theta && q->d_theta && q->exp_avg && q->exp_avg_sq && q->step && q->iwe && q->variance && q->moments && q->workspace &&
                   q->key_offsets,
               "ebos_cmax_patch_solve_resident: NULL buffer");
  EBOS_REQUIRE(mailbox != nullptr && spin_timeout_s > 0.0, "ebos_cmax_patch_solve_resident: NULL mailbox or no spin cap");
  if (!resident_problem_ok(q)) return EBOS_ERR_UNSUPPORTED;
  const HaloArg ha = decode_halo(q->halo);
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w, n_tiles = tiles_y * tiles_x;
  const MailboxLayout m = mailbox_layout(n_tiles);
  if (mailbox_bytes < m.total) {
    set_error("ebos_cmax_patch_solve_resident: mailbox too small (%zu < %zu)", mailbox_bytes, m.total);
    return EBOS_ERR_SCRATCH;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(q->H, q->W, q->tile_h, q->tile_w, ha.halo, 1, 0, 0);
  if (q->workspace_bytes < need) {
    set_error("ebos_cmax_patch_solve_resident: workspace too small (%zu < %zu)", q->workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  if (q->grad_partials_bytes < ebos_patch_grad_partials_bytes(q->H, q->W, q->tile_h, q->tile_w, 0)) {
    set_error("ebos_cmax_patch_solve_resident: grad_partials too small");
    return EBOS_ERR_SCRATCH;
  }
  char* mb = reinterpret_cast<char*>(mailbox);
  ResidentArgs a{};
  a.ev = EvPtrs{nullptr, nullptr, nullptr, nullptr, q->grp_offsets, q->cpix, q->cdt, nullptr, nullptr, nullptr};
  a.key_offsets = q->key_offsets;
  a.H = q->H, a.W = q->W, a.tiles_y = tiles_y, a.tiles_x = tiles_x;
  a.gs = GridSrc{make_axis(q->gh, q->patch_h, q->slide_h, q->H), make_axis(q->gw, q->patch_w, q->slide_w, q->W)};
  a.theta = q->theta, a.d_theta = q->d_theta, a.exp_avg = q->exp_avg, a.exp_avg_sq = q->exp_avg_sq;
  a.theta_mask = q->theta_mask;
  a.step = q->step;
  a.iwe = q->iwe;
  a.slabs = reinterpret_cast<float*>(q->workspace);
  a.cell_partials = q->grad_partials;
  a.status = reinterpret_cast<unsigned*>(mb + m.off_status);
  a.flag1 = reinterpret_cast<unsigned long long*>(mb + m.off_flag1);
  a.flag3 = reinterpret_cast<unsigned long long*>(mb + m.off_flag3);
  a.rec2 = reinterpret_cast<unsigned long long*>(mb + m.off_rec2);
  a.done = reinterpret_cast<unsigned long long*>(mb + m.off_done);
  a.losses = q->losses, a.losses_cap = q->losses_cap, a.t0 = q->steps_done, a.n_iter = n_iter;
  a.lr = q->lr, a.beta1 = q->beta1, a.beta2 = q->beta2, a.eps = q->eps;
  a.w_contrast = q->w_variance;
  a.s_norm = q->w_flow_norm / (float)((int64_t)q->H * q->W);
  a.s_tv = q->w_image_gradient / (float)(2 * (int64_t)q->H * q->W);
  a.omit = q->omit_boundary ? 1 : 0;
  // a built halo (no run-time windows asked for): an infinite |dt| bound makes every tile take the largest window
  a.dt_bound = ha.dyn ? ha.dt_bound : INFINITY;
  a.variance = q->variance;
  a.moments = q->moments;
  const double ticks = spin_timeout_s * 1.0e8;  // wall_clock64: 100 MHz
  a.cap_ticks = ticks < 1.0e3 ? 1000ull : (ticks > 9.0e18 ? 9000000000000000000ull : (unsigned long long)ticks);
  hipStream_t s = as_stream(stream);
  int rc = EBOS_ERR_UNSUPPORTED;
  if (q->tile_h == 45 && q->tile_w == 80 && ha.halo == 32) rc = launch_resident<45, 80, 32>(a, mailbox, m.total, s);
  else if (q->tile_h == 32 && q->tile_w == 32 && ha.halo == 32) rc = launch_resident<32, 32, 32>(a, mailbox, m.total, s);
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_cmax_patch_solve_resident");
  return EBOS_OK;
}

int ebos_cmax_resident_status(const void* mailbox, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(mailbox != nullptr, "ebos_cmax_resident_status: NULL mailbox");
  unsigned st = 0;
  hipStream_t s = as_stream(stream);
  if (hipMemcpyAsync(&st, mailbox, sizeof(st), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("ebos_cmax_resident_status: cannot read the status word (%s)", hipGetErrorString(hipGetLastError()));
    return EBOS_ERR_LAUNCH;
  }
  if (st == RES_OK) return EBOS_OK;
  set_error("resident solve ended early: %s -- theta and the optimiser state are unchanged; run ebos_cmax_patch_solve_f32",
            st == RES_TIMEOUT ? "a wait passed the spin cap (the grid was not co-resident)"
                              : (st == RES_SPILL ? "a tap left the largest LDS window" : "unsupported cell geometry"));
  return -(100 + (int)st);
}

}  // extern "C"
