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

// In-kernel phase stamps (diagnostic builds: EBOS_EXTRA_FLAGS=-DEBOS_STAMPS): workgroup b, phase k of the LAST iteration ->
// g_rstamps[b * 32 + k] (100 MHz clock); read with ebos_debug_read_stamps_resident, tools/stamp_resident.py
#ifdef EBOS_STAMPS
__device__ unsigned long long g_rstamps[1024 * 32];
#define EBOS_RSTAMP(k)                                                                                   \
  do {                                                                                                   \
    if (threadIdx.x == 0 && it == n_iter - 1) g_rstamps[blockIdx.x * 32 + (k)] = wall_clock64();         \
  } while (0)
#else
#define EBOS_RSTAMP(k) \
  do {                 \
  } while (0)
#endif

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
constexpr int kResElems = 128;   // elements (2 components x cells) of a tile's cell block the resident kernel holds state for

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
  unsigned long long *flag1, *flag3, *flagi, *rec2, *done;   // mailbox sections (zeroed before every launch)
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

// Register pressure decides this kernel's speed between its phases: kept live across the event loops (each of which wants ~100
// VGPRs and ~100 SGPRs for itself), the 80 dwords of arguments and the tile's geometry were spilled -- 1200 lane moves and 380
// scratch accesses per iteration, every phase 1.5 - 2 x the time of its stand-alone kernel (first version: 53.6 us per iteration
// at 2 M events against 42.7 us for the four launches).  So nothing uniform is carried: every phase reads the arguments it needs
// afresh from the kernel-argument segment (scalar loads behind an opaque asm: the compiler can neither hoist them out of the
// iteration loop nor merge them across phases) and the geometry from a small LDS block.
typedef const __attribute__((address_space(4))) ResidentArgs KArgs;
__device__ __forceinline__ KArgs& fresh_args() {
  auto p = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(KArgs*)p;
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct Persist {  // one workgroup's iteration-invariant geometry
  int g_first, g_last, beg, end;                   // its slice of the plan (TileRange)
  int gi0, ni, gj0, nj;                            // the block of grid cells tile + apron touch
  int rect_ty0, rect_tx0, rect_ny, rect_nx;        // the tiles whose partial cell gradients that block's cells sum
};

template <int TH, int TW, int HALO>
__global__ void __launch_bounds__(kBlock) cmax_resident_kernel(ResidentArgs a_unused) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the stub: the body copies structs out of the constant address space)
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
  float* s_cells = reinterpret_cast<float*>(s_lerp + PH + PW);         // [2][kGridCells][kGridCells]: theta of the block
  __shared__ TileShared sh;
  __shared__ Persist P;
  __shared__ int s_spill, s_bad, s_ok;
  __shared__ unsigned s_next;
  __shared__ float s_gmax[3 * kWaves];
  __shared__ unsigned s_win[9];
  __shared__ int s_wmax[2];     // largest window (rows, columns) of the grid in this iteration
  __shared__ double s_mom[4];    // mean, variance, sum of the regulariser partials of the previous iteration
  __shared__ double s_reg[2];    // this tile's regulariser value partial: of this iteration, of the previous one
  __shared__ float s_adam[3];    // step size and sqrt(bias correction 2) of the iteration's Adam step; variance of the previous iteration
  __shared__ double s_red[3 * kWaves];

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int tile = blockIdx.x;

  // ---- once: interpolation tables of tile + apron, the block of cells they touch, this thread's element of it -------------------
  // per element of the cell block (thread e < 2 ni nj), in LDS rather than in registers that would be live across every phase:
  __shared__ float s_m[kResElems], s_v[kResElems], s_gl[kResElems];  // Adam's exp_avg / exp_avg_sq, the last gradient
  __shared__ unsigned s_cand_a[kResElems], s_cand_b[kResElems];      // which tiles' partials the element's cell sums (below)
  int n_iter;
  {
    KArgs& a = fresh_args();
    n_iter = a.n_iter;
    const int H = a.H, W = a.W, tiles_x = a.tiles_x, tiles_y = a.tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x, tr0 = ty * TH, tc0 = tx * TW;
    const Axis ay = a.gs.ay, ax = a.gs.ax;
    const EvPtrs ev = a.ev;
    const int32_t* key_offsets = a.key_offsets;
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
    for (int i = threadIdx.x; i < PH + PW; i += kBlock)
      s_lerp[i] = i < PH ? lerp_at(ay, min(max(tr0 + i - AP, 0), H - 1)) : lerp_at(ax, min(max(tc0 + i - PH - AP, 0), W - 1));
    {  // events per source pixel (the backward scatter's fixed-point unit): of the plan, not of the iteration
      int nmax_t = 1;
      const int32_t* ko = key_offsets + (int64_t)tile * (TH * TW);
#pragma unroll
      for (int k = 0; k < (TH * TW + kBlock - 1) / kBlock; ++k) {
        const int i = min((int)threadIdx.x + k * kBlock, TH * TW - 1);
        nmax_t = max(nmax_t, ko[i + 1] - ko[i]);
      }
      const float nm = wave_max_nonneg((float)nmax_t);
      if (lane == 0) s_gmax[kWaves + wave] = nm;
    }
    __syncthreads();
    const int gi0 = s_lerp[0].i0, ni = s_lerp[PH - 1].i1 - gi0 + 1;
    const int gj0 = s_lerp[PH].i0, nj = s_lerp[PH + PW - 1].i1 - gj0 + 1;
    const bool has = (int)threadIdx.x < 2 * ni * nj;  // this thread steps element (ch, gi0 + ci, gj0 + cj) of the cell block
    if (2 * ni * nj > kResElems && threadIdx.x == 0) st_sc1(a.status, (unsigned)RES_GEOMETRY);
    const int e_ = has ? (int)threadIdx.x : 0;
    const int ch = e_ / (ni * nj), ci = (e_ - ch * (ni * nj)) / nj, cj = e_ - ch * (ni * nj) - ci * nj;
    const int gi = gi0 + ci, gj = gj0 + cj;
    const int64_t gidx = ((int64_t)ch * ay.g + gi) * ax.g + gj;
    if (has) {
      s_cells[(ch * kGridCells + ci) * kGridCells + cj] = a.theta[gidx];
      s_m[e_ % kResElems] = a.exp_avg[gidx], s_v[e_ % kResElems] = a.exp_avg_sq[gidx], s_gl[e_ % kResElems] = 0.0f;
    }
    // which tiles' partial cell gradients this cell sums (the arithmetic of patch_grad_combine_kernel, flow_upsample.hip): <= kSpan
    // candidate tiles per axis from the cell's conservative pixel support; a candidate counts if its own cell block holds the cell.
    // cand_a = first candidate tile per axis (2 x 8 bits) | validity masks (2 x 4 bits); cand_b = the cell's index in each
    // candidate's block, 4 bits each (rows: bits 0..15, columns: 16..31)
    {
      unsigned cand_a = 0, cand_b = 0;
      int r_lo, r_hi, c_lo, c_hi;
      support(ay, gi, H, &r_lo, &r_hi);
      support(ax, gj, W, &c_lo, &c_hi);
      const int cty0 = r_lo / TH, ctx0 = c_lo / TW;
      const int ty_n = r_lo < r_hi ? (r_hi - 1) / TH - cty0 + 1 : 0, tx_n = c_lo < c_hi ? (c_hi - 1) / TW - ctx0 + 1 : 0;
      if (has && (ty_n > kSpan || tx_n > kSpan || cty0 > 255 || ctx0 > 255)) st_sc1(a.status, (unsigned)RES_GEOMETRY);
      int first_ty = -1, first_tx = -1;
      unsigned yv = 0, xv = 0;
#pragma unroll
      for (int k = 0; k < kSpan; ++k) {
        const int cty = min(cty0 + k, tiles_y - 1), ctx = min(ctx0 + k, tiles_x - 1);
        const int bi0 = lerp_at(ay, cty * TH).i0, bi1 = lerp_at(ay, min(cty * TH + TH, H) - 1).i1;
        const int bj0 = lerp_at(ax, ctx * TW).i0, bj1 = lerp_at(ax, min(ctx * TW + TW, W) - 1).i1;
        const bool oky = k < ty_n && gi >= bi0 && gi <= bi1, okx = k < tx_n && gj >= bj0 && gj <= bj1;
        cand_b |= (unsigned)(oky ? gi - bi0 : 0) << (4 * k);
        cand_b |= (unsigned)(okx ? gj - bj0 : 0) << (16 + 4 * k);
        yv |= (unsigned)oky << k;
        xv |= (unsigned)okx << k;
        if (oky && first_ty < 0) first_ty = cty;
        if (okx && first_tx < 0) first_tx = ctx;
      }
      // (bit 24: this tile is the first that holds the cell and writes it back at the end)
      cand_a = (unsigned)(cty0 & 255) | ((unsigned)(ctx0 & 255) << 8) | (yv << 16) | (xv << 20) |
               ((unsigned)(first_ty == ty && first_tx == tx) << 24);
      if (has) s_cand_a[e_ % kResElems] = cand_a, s_cand_b[e_ % kResElems] = cand_b;
    }
    if (threadIdx.x == 0) {
      const TileRange tr = tile_range<FMT_COMPACT>(key_offsets, ev, TH * TW, tiles_x, 1);
      P.g_first = tr.g_first, P.g_last = tr.g_last, P.beg = tr.beg, P.end = tr.end;
      P.gi0 = gi0, P.ni = ni, P.gj0 = gj0, P.nj = nj;
      // the tiles whose partials any cell of this block sums: a rectangle of tiles (<= 64, host-checked), waited for at S3
      int lo, hi, dummy;
      support(ay, gi0, H, &lo, &dummy);
      support(ay, gi0 + ni - 1, H, &dummy, &hi);
      P.rect_ty0 = lo / TH;
      P.rect_ny = lo < hi ? min((hi - 1) / TH, tiles_y - 1) - P.rect_ty0 + 1 : 0;
      support(ax, gj0, W, &lo, &dummy);
      support(ax, gj0 + nj - 1, W, &dummy, &hi);
      P.rect_tx0 = lo / TW;
      P.rect_nx = lo < hi ? min((hi - 1) / TW, tiles_x - 1) - P.rect_tx0 + 1 : 0;
      if (P.rect_ny * P.rect_nx > kWave) st_sc1(a.status, (unsigned)RES_GEOMETRY);
      s_ok = 1;
      s_reg[0] = s_reg[1] = 0.0;
      s_adam[2] = 0.0f;
    }
    __syncthreads();
  }
  bool done_ok = true;

  for (int it = 0; it < n_iter; ++it) {
    const unsigned ep = (unsigned)it + 1u;
    EBOS_RSTAMP(0);
    Win<TH, TW, HALO, true> win{HALO, HALO};
    // ---- F0 + F1: the tile's window from a bound on its displacements, the tile's flow; events -> LDS image -> slab ----------------
    {
      KArgs& a = fresh_args();
      const int tiles_x = a.tiles_x;
      TileRange tr;
      tr.ty = tile / tiles_x, tr.tx = tile - tr.ty * tiles_x, tr.slab = tile, tr.part = 0;
      tr.g_first = rfl(P.g_first), tr.g_last = rfl(P.g_last), tr.beg = rfl(P.beg), tr.end = rfl(P.end);
      const EvPtrs ev = a.ev;
      // the event loop's first two chunks per wave, requested now: they arrive under the window bound and the tile's flow instead of
      // a round trip in front of the loop's first deposit (up to 2 M events per window these ARE the tile's events)
      CRaw pre[2];
      pre[0] = load_craw(tr.g_first + wave * kWave + lane, tr, ev);
      pre[1] = load_craw(tr.g_first + (wave + kWaves) * kWave + lane, tr, ev);
      const int gi0 = rfl(P.gi0), gj0 = rfl(P.gj0), ninj = rfl(P.ni) * rfl(P.nj);
      {
        const bool h2 = (int)threadIdx.x < 2 * ninj;
        const int e_ = h2 ? (int)threadIdx.x : 0, ch = e_ / ninj, rem = e_ - ch * ninj, nj = rfl(P.nj), ci = rem / nj, cj = rem - ci * nj;
        const float th = h2 ? s_cells[(ch * kGridCells + ci) * kGridCells + cj] : 0.0f;
        tile_bound_post(h2 && ch == 0 ? fabsf(th) : 0.0f, h2 && ch == 1 ? fabsf(th) : 0.0f, sh.bound);
      }
      if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
      if (threadIdx.x == 0) {
        sh.next = 2 * kWaves;
        sh.chk = 0ull;
        s_spill = 0;
        s_bad = 0;
        s_next = 2 * kWaves;
        s_wmax[0] = s_wmax[1] = 0;
      }
      tile_flow_from_cells<TH, TW, 0>(s_lerp + AP, s_lerp + PH + AP, s_cells, gi0, gj0, s_flow_f);
      __syncthreads();
      win = tile_bound_read<TH, TW, HALO, true>(sh.bound, a.dt_bound);
      EBOS_RSTAMP(1);
      tile_body<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, true, true, false>(tr, win, s_flow_f, s_acc, sh, ev, a.H, a.W, tiles_x, 0, 0,
                                                                                    a.slabs, nullptr, nullptr, 0u, nullptr, pre, NoHook{});
      EBOS_RSTAMP(2);
      drain_stores();
      __syncthreads();
    }
    if (sh.flag[1]) {  // (uniform) a tap left the largest window: the four-launch pipeline's spill path handles that flow
      if (threadIdx.x == 0) st_sc1(fresh_args().status, (unsigned)RES_SPILL);
      done_ok = false;
      break;
    }
    // ---- S1: the eight neighbours' slabs (and their windows) --------------------------------------------------------------------
    constexpr int kQuads = (kLHmax * kLWmax / 4 + kBlock - 1) / kBlock;
    constexpr int kSpecHalo = HALO < 4 ? HALO : 4;
    // (the window of the UPSTREAM image: the four-launch backward kernel stages at least its speculative 4 px window, and the
    // fixed-point unit of the scatter follows max |staged value| -- same window, same unit, same bits)
    const Win<TH, TW, HALO, true> wb = (win.hr <= kSpecHalo && win.hc <= kSpecHalo) ? Win<TH, TW, HALO, true>{kSpecHalo, kSpecHalo} : win;
    float4 own[kQuads];  // this workgroup's own contribution to the quads of its upstream window, decoded from its LDS image
    {
      KArgs& a = fresh_args();
      const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, ty = tile / tiles_x, tx = tile - ty * tiles_x;
      if (threadIdx.x == 0) st_sc1(a.flag1 + tile, ((unsigned long long)ep << 32) | win_pack(win.hr, win.hc));
      EBOS_RSTAMP(3);
      {  // while the neighbours' flags travel: the own part of the gather below (needs nothing of theirs)
        const int H = a.H, W = a.W, tr0 = ty * TH, tc0 = tx * TW;
        const int qw = wb.LW() / 4, n_q = wb.LH() * qw, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
        const float inv_qw = 1.0f / (float)qw;
        const bool lds_f64 = sh.chk != 0ull;             // (tile_body redid its slice exactly: the LDS image holds doubles)
        const int own_lh = win.LH(), own_pt = win.P(), row0 = tr0 - win.HR(), col0 = tc0 - win.HC(), own_lw = win.LW();
#pragma unroll
        for (int kq = 0; kq < kQuads; ++kq) {
          own[kq] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kq * kBlock >= n_q) continue;  // (uniform)
          const int i = threadIdx.x + kq * kBlock;
          const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
          const int r = oy + rl, c = ox + 4 * cq, rr = r - row0, cc = c - col0;
          const bool ok = i < n_q && r >= 0 && r < H && c >= 0 && c < W && (unsigned)rr < (unsigned)own_lh && (unsigned)cc < (unsigned)own_lw;
          const float4 v = lds_image_cells4(s_acc, own_lh, own_pt, ok ? rr : 0, ok ? cc >> 2 : 0, lds_f64);
          if (ok) own[kq] = v;
        }
      }
      if (wave == 0) {
        const int nty = ty + lane / 3 - 1, ntx = tx + lane % 3 - 1;
        const bool nb = lane < 9 && nty >= 0 && nty < tiles_y && ntx >= 0 && ntx < tiles_x;
        unsigned long long* f1 = a.flag1 + (nb ? nty * tiles_x + ntx : tile);
        unsigned wv = 0xffffffffu;
        const bool ok = wave_wait([&]() {
          const unsigned long long f = ld_sc1(f1);
          wv = (unsigned)f;
          return !nb || (unsigned)(f >> 32) >= ep;
        }, a.status, a.cap_ticks);
        if (lane < 9) s_win[lane] = nb ? wv : 0xffffffffu;
        if (lane == 0 && !ok) s_ok = 0;
      } else if (threadIdx.x == kWave) {
        // (an idle wave: Adam's bias corrections of this iteration's step, as torch computes them -- adam_coef, patch_grid.h)
        const AdamCoef coef = adam_coef(a.lr, a.beta1, a.beta2, a.t0 + it + 1);
        s_adam[0] = coef.step_size;
        s_adam[1] = coef.bc2_sqrt;
      }
      __syncthreads();
    }
    if (!s_ok) { done_ok = false; break; }
    EBOS_RSTAMP(4);
    // ---- G: the UPSTREAM WINDOW of the IWE (this tile + the halo the backward sweep reads) = per pixel the sum of the slabs whose
    // windows reach it, in the combine pass's order (same bits as the four-launch image).  This workgroup's own contribution is
    // decoded from its LDS image (what it stored to its slab, without the round trip); the neighbours' come from their slabs.  A halo
    // pixel is complete with the 3 x 3 tiles around THIS tile as long as no tile two away reaches it: hr + hr' < TH, hc + hc' < TW
    // for any two windows -- checked for the whole grid after the all-to-all (every record carries its window); BOS-sized flows
    // pass, and nothing of the image then travels through memory.  Otherwise (checked below) the tiles publish their images and the
    // halo is staged from those, as the four-launch backward kernel does.
    {
      float4 wq[kQuads];
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x, tr0 = ty * TH, tc0 = tx * TW;
      const int lo_px = a.omit ? 1 : 0;
      const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(a.slabs, 0xffffffffu);
      const int qw = wb.LW() / 4, n_q = wb.LH() * qw, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
      const float inv_qw = 1.0f / (float)qw;
      double sm = 0.0, sq = 0.0;
#pragma unroll
      for (int kq = 0; kq < kQuads; ++kq) {
        wq[kq] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kq * kBlock >= n_q) continue;  // (uniform: a small window has fewer quads than the largest one's kQuads per thread)
        const int i = threadIdx.x + kq * kBlock;
        const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
        const int r = oy + rl, c = ox + 4 * cq;
        const bool live = i < n_q && r >= 0 && r < H && c >= 0 && c < W;
        float4 part[9];
        bool okk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          // (what depends on the candidate only is uniform: scalar registers and the scalar unit -- these phases are bound by vector
          // instruction issue at four waves per SIMD, not by memory)
          const unsigned w = (unsigned)rfl((int)s_win[k]);
          const int nty = ty + k / 3 - 1, ntx = tx + k % 3 - 1;
          const int hr = (int)(w & 255u), hc = (int)((w >> 8) & 255u);
          const int row0 = nty * TH - hr, col0 = ntx * TW - hc, lw = TW + 2 * hc, lh = TH + 2 * hr;
          const unsigned slab0 = (unsigned)(nty * tiles_x + ntx) * (unsigned)(kLHmax * kLWmax);
          const int rr = r - row0, cc = c - col0;
          okk[k] = live && w != 0xffffffffu && (unsigned)rr < (unsigned)lh && (unsigned)cc < (unsigned)lw;
          part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k == 4) {  // this workgroup's own image: decoded from LDS above
            part[k] = own[kq];
          } else if (w != 0xffffffffu && __builtin_amdgcn_ballot_w64(okk[k]) != 0ull) {
            // (a neighbour's window reaches only the rim of this window: most waves hold no quad of it and skip its load)
            const unsigned byte = okk[k] ? (slab0 + (unsigned)(rr * lw + cc)) * 4u : 0u;
            part[k] = slab_load4(all_slabs, byte);
          }
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 9; ++k)
          if (okk[k]) v.x += part[k].x, v.y += part[k].y, v.z += part[k].z, v.w += part[k].w;
        wq[kq] = v;
        // the variance moments: this tile's own pixels (quads lie inside a tile as a whole or outside it)
        if (live && r >= tr0 && r < tr0 + TH && c >= tc0 && c < tc0 + TW && r >= lo_px && r < H - lo_px) {
          const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c + k >= lo_px && c + k < W - lo_px) {
              sm += (double)e4[k];
              sq += (double)e4[k] * (double)e4[k];
            }
        }
      }
      EBOS_RSTAMP(5);
      block_sum2(sm, sq, s_red);  // (behind its barriers every thread has read what it needs of the LDS image: the backward view may overwrite it)
      if (threadIdx.x == 0) {
        unsigned long long* rec = a.rec2 + ((size_t)(it & 1) * (a.tiles_y * tiles_x) + tile) * kRecGranules;
        put_granules(rec, ep, sm);
        put_granules(rec + 2, ep, sq);
        put_granules(rec + 4, ep, s_reg[1]);
        st_sc1(rec + 6, ((unsigned long long)ep << 32) | win_pack(wb.hr, wb.hc));
      }
      // While the records travel: the raw window goes to its place in LDS (held in registers across the all-to-all it was spilled:
      // +4 us), the d_flow accumulators are cleared and the tile's flow with its apron is evaluated -- none of it needs the mean.
#pragma unroll
      for (int kq = 0; kq < kQuads; ++kq) {
        const int i = threadIdx.x + kq * kBlock;
        if (i < n_q) reinterpret_cast<float4*>(s_g)[i] = wq[kq];
      }
      for (int i = threadIdx.x; i < TH * TW; i += kBlock) reinterpret_cast<double2*>(s_d)[i] = make_double2(0.0, 0.0);  // [2][TH * TW]
      tile_flow_from_cells<TH, TW, AP>(s_lerp, s_lerp + PH, s_cells, rfl(P.gi0), rfl(P.gj0), s_flow_b);
    }
    // the backward sweep's first two chunks per wave, requested before the all-to-all and decoded behind it
    BwdPreRaw pre_raw;
    {
      KArgs& a = fresh_args();
      TileRange tr;
      tr.ty = tr.tx = 0, tr.slab = tile, tr.part = 0;
      tr.g_first = rfl(P.g_first), tr.g_last = rfl(P.g_last), tr.beg = rfl(P.beg), tr.end = rfl(P.end);
      const EvPtrs ev = a.ev;
      pre_raw.A = load_craw(tr.g_first + wave * kWave + lane, tr, ev);
      pre_raw.B = load_craw(tr.g_first + (wave + kWaves) * kWave + lane, tr, ev);
    }
    EBOS_RSTAMP(6);
    // ---- S2: the one all-to-all: every tile's (sum, sum of squares, regulariser partial of the previous iteration, window) ---------
    double mean;
    bool halo_complete;
    {
      KArgs& a = fresh_args();
      const int n_tiles = a.tiles_y * a.tiles_x;
      const int lo_px = a.omit ? 1 : 0;
      const double n_px = (double)max(a.H - 2 * lo_px, 0) * (double)max(a.W - 2 * lo_px, 0);
      double as = 0.0, aq = 0.0, ar = 0.0;
      if (wave * kWave < n_tiles) {
        const int k = wave * kWave + lane;
        const unsigned long long* rec = a.rec2 + ((size_t)(it & 1) * n_tiles + min(k, n_tiles - 1)) * kRecGranules;
        unsigned long long g[7];
        const bool ok = wave_wait([&]() {
          bool all = true;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
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
        const float mh = wave_max_nonneg(ok ? (float)((unsigned)g[6] & 255u) : 255.0f);
        const float mw = wave_max_nonneg(ok ? (float)(((unsigned)g[6] >> 8) & 255u) : 255.0f);
        if (lane == 0) {
          atomicMax(&s_wmax[0], (int)mh);
          atomicMax(&s_wmax[1], (int)mw);
          if (!ok) s_ok = 0;
        }
      }
      EBOS_RSTAMP(7);
      as = wave_sum(as), aq = wave_sum(aq), ar = wave_sum(ar);
      if (lane == 0) s_red[wave] = as, s_red[kWaves + wave] = aq, s_red[2 * kWaves + wave] = ar;
      __syncthreads();
      if (threadIdx.x == 0) {
        double S = 0.0, Q = 0.0, R = 0.0;
        for (int k = 0; k < kWaves; ++k) S += s_red[k], Q += s_red[kWaves + k], R += s_red[2 * kWaves + k];
        const double mn = n_px > 0.0 ? S / n_px : 0.0;
        s_mom[0] = mn;
        s_mom[1] = (Q - S * mn) / (n_px - 1.0);
        s_mom[2] = R;
      }
      __syncthreads();
      mean = s_mom[0];
      halo_complete = 2 * s_wmax[0] < TH && 2 * s_wmax[1] < TW;  // (uniform over the GRID: every workgroup saw every window)
      if (blockIdx.x == 0 && threadIdx.x == 0) {  // bookkeeping: the loss of the PREVIOUS iteration is complete now
        const float var_f = (float)s_mom[1];
        const int t_prev = a.t0 + it - 1;
        if (it > 0 && a.losses != nullptr && t_prev < a.losses_cap)
          a.losses[t_prev] = (float)(-(double)a.w_contrast * (double)s_adam[2] + s_mom[2]);
        s_adam[2] = var_f;
        a.variance[0] = var_f;
        a.moments[0] = mean;
        a.moments[1] = n_px;
      }
    }
    if (!s_ok) { done_ok = false; break; }
    EBOS_RSTAMP(8);
    // ---- the image leaves the kernel in its last iteration -- or now, when tiles two apart reach into each other's halos: every
    // tile publishes its pixels and the halo is read back from the neighbours' (the four-launch pipeline's staging)
    if (!halo_complete || it == n_iter - 1) {
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, tiles_x = a.tiles_x, tiles_y = a.tiles_y, ty = tile / tiles_x, tx = tile - ty * tiles_x, tr0 = ty * TH, tc0 = tx * TW;
      const int qw = wb.LW() / 4, n_q = wb.LH() * qw, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
      const float inv_qw = 1.0f / (float)qw;
      float* iwe = a.iwe;
#pragma unroll
      for (int kq = 0; kq < kQuads; ++kq) {
        const int i = threadIdx.x + kq * kBlock;
        const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
        const int r = oy + rl, c = ox + 4 * cq;
        if (i < n_q && r >= tr0 && r < min(tr0 + TH, H) && c >= tc0 && c < tc0 + TW) {
          const float4 v = reinterpret_cast<const float4*>(s_g)[i];
          const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c + k < W) st_sc1(iwe + (int64_t)r * W + c + k, e4[k]);
        }
      }
      if (!halo_complete) {
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) st_sc1(a.flagi + tile, (unsigned long long)ep);
        if (wave == 0) {
          const int nty = ty + lane / 3 - 1, ntx = tx + lane % 3 - 1;
          const bool nb = lane < 9 && nty >= 0 && nty < tiles_y && ntx >= 0 && ntx < tiles_x;
          const unsigned long long* f = a.flagi + (nb ? nty * tiles_x + ntx : tile);
          const bool ok = wave_wait([&]() { return !nb || ld_sc1(f) >= (unsigned long long)ep; }, a.status, a.cap_ticks);
          if (lane == 0 && !ok) s_ok = 0;
        }
        __syncthreads();
        if (s_ok) {
          float4 wq[kQuads];
#pragma unroll
          for (int kq = 0; kq < kQuads; ++kq) {
            const int i = min((int)threadIdx.x + kq * kBlock, n_q - 1);
            const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
            const int R = min(max(oy + rl, 0), H - 1), c = ox + 4 * cq;
            const float* row = iwe + (int64_t)R * W;
            wq[kq] = make_float4(ld_sc1(row + min(max(c, 0), W - 1)), ld_sc1(row + min(max(c + 1, 0), W - 1)),
                                 ld_sc1(row + min(max(c + 2, 0), W - 1)), ld_sc1(row + min(max(c + 3, 0), W - 1)));
          }
#pragma unroll
          for (int kq = 0; kq < kQuads; ++kq) {
            const int i = threadIdx.x + kq * kBlock;
            if (i < n_q) reinterpret_cast<float4*>(s_g)[i] = wq[kq];  // (every thread rewrites the quads it read above: no barrier needed)
          }
        }
      }
    }
    if (!s_ok) { done_ok = false; break; }
    // ---- B0 + B1: upstream window of d loss / d IWE = 2 (-w) (IWE - mean) / (M - 1) -> LDS; the sweep ------------------------------------
    FxUnit unit;
    bool fx;
    {
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x, tr0 = ty * TH, tc0 = tx * TW;
      const int lo_px = a.omit ? 1 : 0;
      const double n_px = (double)max(H - 2 * lo_px, 0) * (double)max(W - 2 * lo_px, 0);
      const int qw = wb.LW() / 4, n_q = wb.LH() * qw, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
      const float inv_qw = 1.0f / (float)qw;
      GradImage G;
      G.g = a.iwe;
      const double ga = 2.0 * (-(double)a.w_contrast) / (n_px - 1.0);
      G.a = (float)ga;
      G.c = (float)(-ga * mean);
      G.h = H, G.w = W, G.lo = lo_px;
      float gmax_t = 0.0f, gsum_t = 0.0f;  // (max and sum of |staged value|: the scatter's fixed-point unit, bwd_fx_unit)
      auto affine = [&](int R, int C, float v, bool in_window) {
        const bool valid = R >= G.lo && R < G.h - G.lo && C >= G.lo && C < G.w - G.lo;
        const float gv = valid ? G.a * v + G.c : 0.0f;
        if (in_window) {
          gmax_t = fmaxf(gmax_t, gv == gv ? fabsf(gv) : INFINITY);
          gsum_t += fabsf(gv);
        }
        return gv;
      };
      EBOS_RSTAMP(18);
      // the affine map of the image (the variance gradient), in place: every thread maps the quads it parked
      // (an interior tile's window lies inside the valid region as a whole: no per-pixel tests -- vector instruction issue, not
      // memory, bounds these passes)
      const bool all_valid = oy >= G.lo && oy + wb.LH() <= G.h - G.lo && ox >= G.lo && ox + wb.LW() <= G.w - G.lo;
#pragma unroll
      for (int kq = 0; kq < kQuads; ++kq) {
        const int i = threadIdx.x + kq * kBlock;
        const bool in = i < n_q;
        if (kq * kBlock >= n_q) continue;  // (uniform)
        const float4 v = reinterpret_cast<const float4*>(s_g)[in ? i : 0];
        float4 gq;
        if (all_valid) {
          gq = make_float4(G.a * v.x + G.c, G.a * v.y + G.c, G.a * v.z + G.c, G.a * v.w + G.c);
          if (in) {
            const float m4 = fmaxf(fmaxf(fabsf(gq.x), fabsf(gq.y)), fmaxf(fabsf(gq.z), fabsf(gq.w)));
            gmax_t = fmaxf(gmax_t, (gq.x + gq.y + gq.z + gq.w) == (gq.x + gq.y + gq.z + gq.w) ? m4 : INFINITY);  // (a NaN anywhere: Inf)
            gsum_t += (fabsf(gq.x) + fabsf(gq.y)) + (fabsf(gq.z) + fabsf(gq.w));
          }
        } else {
          const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
          const int R = oy + rl, C = ox + 4 * cq;
          gq = make_float4(affine(R, C, v.x, in), affine(R, C + 1, v.y, in), affine(R, C + 2, v.z, in), affine(R, C + 3, v.w, in));
        }
        if (in) reinterpret_cast<float4*>(s_g)[i] = gq;
      }
      EBOS_RSTAMP(19);
      gmax_t = wave_max_nonneg(gmax_t);
      gsum_t = wave_sum(gsum_t);
      if (lane == 0) s_gmax[wave] = gmax_t, s_gmax[2 * kWaves + wave] = gsum_t;
      __syncthreads();
      EBOS_RSTAMP(9);
      unit = bwd_fx_unit(s_gmax, a.dt_bound, n_q * 4);
      double tot_x = 0.0, tot_y = 0.0;
      const BwdShared bsh{&s_spill, &s_bad, &s_next};
      TileRange tr;
      tr.ty = ty, tr.tx = tx, tr.slab = tile, tr.part = 0;
      tr.g_first = rfl(P.g_first), tr.g_last = rfl(P.g_last), tr.beg = rfl(P.beg), tr.end = rfl(P.end);
      const EvPtrs ev = a.ev;
      BwdPre pre;
      decode_bgroup(pre.A, pre_raw.A, (unsigned)PW, 0u, (unsigned)(AP * PW + AP));  // (the tile's flow in LDS: element indices, pitch PW)
      decode_bgroup(pre.B, pre_raw.B, (unsigned)PW, 0u, (unsigned)(AP * PW + AP));
      finish_bgroup<TW>(pre.A);
      finish_bgroup<TW>(pre.B);
      fx = bwd_lean_sweeps<TH, TW, HALO, false, true, true>(tr, s_d, s_g, ev, s_flow_b, H, W, 0, 0, G, tot_x, tot_y, ChunkQueue{&s_next}, wb,
                                                           unit, a.dt_bound, pre, true, bsh, NoHook{});
      __syncthreads();
    }
    EBOS_RSTAMP(10);
    // ---- B2: regularisers on the tile's flow, adjoint of grid -> dense on the tile -> partial cell gradients (write-through) ------
    {
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
      TileRange tr;
      tr.ty = ty, tr.tx = tx, tr.slab = tile, tr.part = 0;
      tr.g_first = tr.g_last = tr.beg = tr.end = 0;
      const TileGrad<TH, TW> grad{fx, 1.0f / unit.scale, s_d};
      const float s_norm = a.s_norm, s_tv = a.s_tv;
      const bool any_reg = s_norm != 0.0f || s_tv != 0.0f;
      grid_tile_epilogue<TH, TW, HALO, true>(tr, ty * TH, tx * TW, H, W, s_d, s_g, s_flow_b, s_lerp, grad, nullptr, s_norm, s_tv,
                                             any_reg ? &s_reg[0] : nullptr, a.cell_partials + (int64_t)tile * (2 * kGridCells * kGridCells));
      EBOS_RSTAMP(11);
      drain_stores();
      __syncthreads();
      if (threadIdx.x == 0) {
        st_sc1(a.flag3 + tile, (unsigned long long)ep);
        s_reg[1] = any_reg ? s_reg[0] : 0.0;
      }
    }
    EBOS_RSTAMP(12);
    // ---- S3: the partials of the tiles this block's cells sum; meanwhile the other waves clear the LDS image for the next pass ---
    {
      KArgs& a = fresh_args();
      if (wave == 0) {
        const int rect_nx = max(rfl(P.rect_nx), 1), n_rect = rfl(P.rect_ny) * rfl(P.rect_nx);
        const int ry = lane / rect_nx, rx = lane - ry * rect_nx;
        const bool act = lane < n_rect;
        const unsigned long long* f = a.flag3 + (rfl(P.rect_ty0) + (act ? ry : 0)) * a.tiles_x + rfl(P.rect_tx0) + (act ? rx : 0);
        const bool ok = wave_wait([&]() { return !act || ld_sc1(f) >= (unsigned long long)ep; }, a.status, a.cap_ticks);
        if (lane == 0 && !ok) s_ok = 0;
      } else {
        for (int i = threadIdx.x - kWave; i < kCells / 2; i += kBlock - kWave) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
      }
      __syncthreads();
    }
    if (!s_ok) { done_ok = false; break; }
    EBOS_RSTAMP(13);
    // ---- A: d loss / d theta of this thread's cell element = sum of the partials of the tiles that hold it; Adam ------------------
    if ((int)threadIdx.x < 2 * rfl(P.ni) * rfl(P.nj)) {
      KArgs& a = fresh_args();
      const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;
      const int ninj = rfl(P.ni) * rfl(P.nj), nj = rfl(P.nj);
      const unsigned cand_a = s_cand_a[threadIdx.x % kResElems], cand_b = s_cand_b[threadIdx.x % kResElems];
      float m_e = s_m[threadIdx.x % kResElems], v_e = s_v[threadIdx.x % kResElems];
      const int ch = (int)threadIdx.x / ninj, rem = (int)threadIdx.x - ch * ninj, ci = rem / nj, cj = rem - ci * nj;
      const float* cp = a.cell_partials;
      const int cty0 = (int)(cand_a & 255u), ctx0 = (int)((cand_a >> 8) & 255u);
      float mask = 1.0f;
      if (a.theta_mask != nullptr) mask = a.theta_mask[(int64_t)(rfl(P.gi0) + ci) * a.gs.ax.g + rfl(P.gj0) + cj];
      float pv[kSpan][kSpan];
#pragma unroll
      for (int p = 0; p < kSpan; ++p)
#pragma unroll
        for (int q = 0; q < kSpan; ++q) {
          const int cty = min(cty0 + p, tiles_y - 1), ctx = min(ctx0 + q, tiles_x - 1);
          const int li = (int)((cand_b >> (4 * p)) & 15u), lj = (int)((cand_b >> (16 + 4 * q)) & 15u);
          pv[p][q] = ld_sc1(cp + (((int64_t)(cty * tiles_x + ctx) * 2 + ch) * kGridCells + li) * kGridCells + lj);
        }
      float g = 0.0f;
#pragma unroll
      for (int p = 0; p < kSpan; ++p)
#pragma unroll
        for (int q = 0; q < kSpan; ++q) g += (((cand_a >> (16 + p)) & 1u) && ((cand_a >> (20 + q)) & 1u)) ? pv[p][q] : 0.0f;
      if (a.theta_mask != nullptr) g *= mask;
      float th = s_cells[(ch * kGridCells + ci) * kGridCells + cj];
      adam_update(g, m_e, v_e, th, s_adam[0], s_adam[1], (float)a.beta2, (float)(1.0 - a.beta1), (float)(1.0 - a.beta2), (float)a.eps);
      s_cells[(ch * kGridCells + ci) * kGridCells + cj] = th;
      s_m[threadIdx.x % kResElems] = m_e, s_v[threadIdx.x % kResElems] = v_e, s_gl[threadIdx.x % kResElems] = g;
    }
    EBOS_RSTAMP(14);
    __syncthreads();  // the new theta block is in LDS
  }
  if (!done_ok) return;  // (uniform) nothing of the optimiser state was written: the host falls back from unchanged state

  // ---- the state goes back: every cell element by the first tile that holds it ------------------------------------------------------
  KArgs& a = fresh_args();
  if ((int)threadIdx.x < 2 * rfl(P.ni) * rfl(P.nj) && ((s_cand_a[threadIdx.x % kResElems] >> 24) & 1u)) {
    const int ninj = rfl(P.ni) * rfl(P.nj), nj = rfl(P.nj);
    const int ch = (int)threadIdx.x / ninj, rem = (int)threadIdx.x - ch * ninj, ci = rem / nj, cj = rem - ci * nj;
    const int64_t gidx = ((int64_t)ch * a.gs.ay.g + rfl(P.gi0) + ci) * a.gs.ax.g + rfl(P.gj0) + cj;
    a.theta[gidx] = s_cells[(ch * kGridCells + ci) * kGridCells + cj];
    a.exp_avg[gidx] = s_m[threadIdx.x % kResElems];
    a.exp_avg_sq[gidx] = s_v[threadIdx.x % kResElems];
    a.d_theta[gidx] = s_gl[threadIdx.x % kResElems];
  }
  if (n_iter <= 0) return;
  // the last iteration's loss: its regulariser partials travel through the `done` granules; workgroup 0 gathers them
  const int n_tiles = a.tiles_y * a.tiles_x;
  if (threadIdx.x == 0) put_granules(a.done + (size_t)tile * 2, (unsigned)n_iter, s_reg[1]);
  if (blockIdx.x != 0) return;
  double ar = 0.0;
  if (wave * kWave < n_tiles) {
    const int k = wave * kWave + lane;
    const unsigned long long* rec = a.done + (size_t)min(k, n_tiles - 1) * 2;
    unsigned long long g0 = 0, g1 = 0;
    const bool ok = wave_wait([&]() {
      g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1);
      return (unsigned)(g0 >> 32) == (unsigned)n_iter && (unsigned)(g1 >> 32) == (unsigned)n_iter;
    }, a.status, a.cap_ticks);
    if (ok && k < n_tiles) ar = __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
  }
  ar = block_sum(ar, s_red);
  if (threadIdx.x == 0) {
    const int t_last = a.t0 + n_iter - 1;
    if (a.losses != nullptr && t_last < a.losses_cap) a.losses[t_last] = (float)(-(double)a.w_contrast * (double)s_adam[2] + ar);
    a.step[0] = a.t0 + n_iter;
  }
#endif
}

struct MailboxLayout {
  size_t off_status, off_flag1, off_flag3, off_flagi, off_rec2, off_done, total;
};
inline MailboxLayout mailbox_layout(int n_tiles) {
  MailboxLayout m;
  m.off_status = 0;
  m.off_flag1 = 256;
  m.off_flag3 = m.off_flag1 + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
  m.off_flagi = m.off_flag3 + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
  m.off_rec2 = m.off_flagi + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
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
  // splits: 1, or 0 = "adaptive work items" (a table the four-launch pipeline splits crowded tiles by: the resident kernel always
  // runs one workgroup per tile and does not read it -- same objective, slab sums in another order where a tile was split)
  if (q->splits > 1 || q->splits < 0 || q->pad_h != 0 || q->pad_w != 0) {
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
  if (((q->tile_h + 2 * kBwdApron) / q->slide_h + 3) * ((q->tile_w + 2 * kBwdApron) / q->slide_w + 3) * 2 > kResElems) {
    set_error("resident solve: sliding window %dx%d: a tile's block of grid cells has more than %d elements", q->slide_h, q->slide_w, kResElems);
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

#ifdef EBOS_STAMPS
int ebos_debug_read_stamps_resident(unsigned long long* host, int count) {  // diagnostic builds only
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_rstamps), sizeof(unsigned long long) * count);
}
int ebos_debug_read_stamps_resident_bwd(unsigned long long* host, int count) {  // (this file's copy of the shared backward stamps)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_stamps_bwd), sizeof(unsigned long long) * count);
}
#endif

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
  a.flagi = reinterpret_cast<unsigned long long*>(mb + m.off_flagi);
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
