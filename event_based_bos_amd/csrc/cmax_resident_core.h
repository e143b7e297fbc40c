// cmax_resident_core.h -- the contrast-maximisation inner loop as ONE resident launch (gfx950): the kernel template and its launcher.
// Built per tile shape in a translation unit of its own (cmax_resident_45x80.hip, _32x32.hip, _32x64.hip: the kernel takes minutes to
// compile, the units compile side by side); cmax_resident.hip holds the C entry points and the per-device launch order.
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
// and exchanges per iteration, through global memory, in two hand-offs
//   S1  its LDS image as a slab (write-through) + a record {epoch, its share of sum(IWE), its window}: the ALL-TO-ALL of the
//       iteration.  The share is summed from the tile's own LDS image in the decode pass -- exact in a double, so the mean is the
//       four-launch pipeline's bit for bit (kCombineExactSum) -- and is known before any halo has travelled: behind the wait every
//       workgroup gathers its upstream window (tile + halo: own part from LDS, the neighbours' from their slabs, in the combine
//       pass's order of additions), maps it to d loss / d IWE and stages it in ONE pass
//   S3  its <= 16 x 16 partial cell gradients as tagged granules {epoch, value}: every thread that steps a cell element polls the
//       <= 4 x 4 values its cell sums (the data is its own flag: no flag store behind a drain, no second round trip), then Adam.
// The sum of squares of the image (the loss VALUE only) travels in a second record that workgroup 0 alone reads, one iteration
// later, in the shadow of its S3 wait.
// Hand-off form (cdna guide, Guideline 16 / MI355X_MICROARCH visibility table, first row): every handed-off byte is an sc1
// (write-through) store, every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier behind which ONE lane
// stores the flag (sc1); consumers poll with sc1 loads and read the payload with sc1 loads only -- no fences, no atomics.
// Every spin is bounded: a wave that waits longer than the caller's cap (or sees the status word set) raises the status word
// and the whole grid leaves; theta and the optimiser state are written back only by a launch that completed, so the host can
// fall back to the four-launch pipeline from unchanged state (ebos_cmax_resident_status).  Taps beyond the LDS window (the
// spill path of the four-launch pipeline, global atomics) end the launch the same way: correct for any flow, fast for BOS-sized ones.
//
// outer_padding (src/event_image_converter.py:29-34; round 6): the image is [H + 2 pad_h, W + 2 pad_w] while tiles, flow and grid live
// on the H x W source pixels.  Everything the kernel does in IMAGE coordinates (the own-part decode, the gather, the contrasts' passes,
// publish, the valid region) runs on the padded image with the tile origins shifted by the padding; every window is widened to at
// least the padding ring, so that the border tile that OWNS a ring pixel (publishes it, counts its square) has gathered whatever any
// neighbour deposited there -- a neighbour's larger run-time window was what kept padded problems on the four launches before.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "iwe_tile_core.h"
#include "handoff.h"
#include "sobel3.h"

// Timing builds (tools/ablate_resident.sh): EBOS_ABL is a mask of pieces of the iteration to leave out -- results are WRONG on
// purpose; what a piece costs where it stands is the difference to the whole.  0 in the product.
#ifndef EBOS_ABL
#define EBOS_ABL 0
#endif

namespace ebos {
namespace {

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
  RES_IMBALANCED = 4,  // one tile holds far more events than the average one: this kernel runs ONE workgroup per tile, the four-launch
                       // pipeline splits crowded tiles over several (adaptive work items) -- measured 184 against 95 us per iteration
                       // with 2 M events in a Gaussian blob of sigma 100 px
};

// The launch's verdict: the FIRST one stands (with plain stores a spill could overwrite a timeout, and the workgroups
// that read one verdict left without the write-back the others, reading the other, performed).
__device__ __forceinline__ void raise_status(unsigned* status, unsigned code) {
  typedef __attribute__((address_space(1))) unsigned gu32s;
  unsigned expected = 0u;  // RES_OK
  __hip_atomic_compare_exchange_strong((gu32s*)status, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A whole wave polls: lane-wise predicate, true when every lane's holds.  Bounded: every 32 polls the status word and the clock
// (100 MHz) are looked at; false = the launch is over (status set by this wave or seen set).
// A spill (status = RES_SPILL | iteration << 8) ends only the wait for THAT iteration's records (`spill_it`: the iteration a wait
// belongs to if it is the all-to-all's, -1 otherwise): every other wait's data still arrives -- the workgroups all finish the
// iterations before the spill and stop in front of the same all-to-all, from where the launch hands over a consistent state.
template <typename Pred>
__device__ __forceinline__ bool wave_wait(Pred&& ready, unsigned* status, unsigned long long cap_ticks, int spill_it = -1) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    if (__all(ready())) return true;
    if ((++spins & 31u) == 0u) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      const unsigned st = ld_sc1(status);
      if ((st & 255u) == RES_SPILL) {
        if (spill_it >= 0 && (int)(st >> 8) == spill_it) return false;
      } else if (st != RES_OK) return false;
      if (now - t0 > cap_ticks) {
        if ((threadIdx.x & (kWave - 1)) == 0) raise_status(status, (unsigned)RES_TIMEOUT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

constexpr int kRec1Granules = 4; // S1's record: (the tile's share of sum(IWE)) = 2 granules + its window, 32-byte stride
constexpr int kRecGranules = 8;  // the loss record: (sum of squares, regulariser partial of the previous iteration) = 4 granules, 64-byte stride
constexpr int kSpan = 4;         // candidate tiles per axis whose partial cell gradients a cell sums (patch_grad_combine_kernel's)
constexpr int kStoredCands = 128;
constexpr int kResElems = 256;   // elements (2 components x cells) of a tile's cell block the resident kernel holds state for
                                 // (tile 45 x 80 with 8 x 8 patches -- the finest scale of the reference's pyramid, patch_eklt_pyramid2.py:55-83 -- has 234)

struct ResidentArgs {
  EvPtrs ev;
  const int32_t* key_offsets;
  int H, W, tiles_y, tiles_x;
  int pad_h, pad_w;      // outer_padding of the image (src/event_image_converter.py:29-34): the IWE is [H + 2 pad_h, W + 2 pad_w]
  GridSrc gs;
  float *theta, *d_theta, *exp_avg, *exp_avg_sq;
  const float* theta_mask;
  int* step;
  float *iwe, *slabs, *cell_partials;
  unsigned* status;
  unsigned long long *rec1, *part3, *flagi, *rec2, *done;    // mailbox sections (zeroed before every launch)
  float* losses;
  int losses_cap, t0, n_iter;
  double lr, beta1, beta2, eps;
  float w_contrast, s_norm, s_tv;
  int omit;
  float dt_bound;
  float* variance;
  double* moments;
  unsigned long long cap_ticks;
  float max_imbalance;   // > 0: leave with RES_IMBALANCED when (events of the fullest tile) > max_imbalance x (events of the average tile);
                         // < 0: the default rule (fullest tile > 60 k + 0.8 % of the window on >= 128 tiles, > 85 k on smaller sensors); 0: never
  Blur3 blur;            // k0 != 0: the contrast of the 3-tap blurred image (iwe.blur_sigma > 0, blur3.h)
  int gm;                // the gradient-magnitude contrast (sobel3.h) instead of the variance; patch-flow problems only
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
  cells_to_flow<TH + 2 * AP, TW + 2 * AP>(s_rows, s_cols, s_cells, gi0, gj0, s_flow);
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

// Workgroup 0, one whole wave: iteration j's loss record (sum of squares, regulariser partials of iteration j - 1) is complete ->
// variance(j) = (Q - S mean) / (M - 1) with (S, mean) kept from j's all-to-all; loss(j - 1) = -w variance(j - 1) + regularisers(j - 1)
// is written, variance(j) kept in s_adam[2] for the next call.  (The order of the additions of Q differs from the four-launch
// pipeline's -- ~900 combine workgroups there -- in the last bits of a double that is then rounded to the f32 the loss is built from.)
template <typename Args>
__device__ __forceinline__ void book_loss(const Args& a, int j, int lane, const double* s_hist, float* s_adam) {
  const int n_tiles = a.tiles_y * a.tiles_x;
  double Q = 0.0, R = 0.0;
  for (int k = lane; k < n_tiles; k += kWave) {
    const unsigned long long* rec = a.rec2 + ((size_t)(j & 1) * n_tiles + k) * kRecGranules;
    const unsigned long long g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1), g2 = ld_sc1(rec + 2), g3 = ld_sc1(rec + 3);
    Q += __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
    R += __builtin_bit_cast(double, (g2 & 0xffffffffull) | (g3 << 32));
  }
  Q = wave_sum(Q);
  R = wave_sum(R);
  if (lane == 0) {
    const int lo_px = a.omit ? 1 : 0;
    const double n_px = (double)max(a.H + 2 * a.pad_h - 2 * lo_px, 0) * (double)max(a.W + 2 * a.pad_w - 2 * lo_px, 0);
    const double S = s_hist[(j & 1) * 2], mn = s_hist[(j & 1) * 2 + 1];
    // (gradient magnitude: Q is the sum of the squared Sobel pairs, the contrast their mean -- gradmag_fused_finalize_kernel)
    const float var_f = a.gm ? (float)(Q / n_px) : (float)((Q - S * mn) / (n_px - 1.0));
    const int t_prev = a.t0 + j - 1;
    if (j >= 1 && a.losses != nullptr && t_prev < a.losses_cap) a.losses[t_prev] = (float)(-(double)a.w_contrast * (double)s_adam[2] + R);
    s_adam[2] = var_f;
  }
}

// Which tiles' partial cell gradients cell (gi, gj) sums -- the arithmetic of patch_grad_combine_kernel (flow_upsample.hip): <= kSpan
// candidate tiles per axis from the cell's conservative pixel support; a candidate counts if its own cell block holds the cell.
//   cand_a = first candidate tile per axis (2 x 8 bits) | validity masks (2 x 4 bits) | bit 24: tile (ty, tx) is the FIRST that
//            holds the cell (it writes the cell back at the end) | bit 25: more than kSpan candidates on an axis (a wide cell)
//   cand_b = the cell's index in each candidate's block, 4 bits each (rows: bits 0..15, columns: 16..31)
// Recomputed where it is needed (once per iteration, by the <= 256 threads that step a cell element: a few hundred instructions)
// instead of kept: two more words per element in LDS were what held the cell block at 128 elements.
template <int TH, int TW>
__device__ __forceinline__ void cell_candidates(const Axis& ay, const Axis& ax, int gi, int gj, int H, int W, int tiles_y, int tiles_x,
                                                int ty, int tx, unsigned& cand_a, unsigned& cand_b) {
  int r_lo, r_hi, c_lo, c_hi;
  support(ay, gi, H, &r_lo, &r_hi);
  support(ax, gj, W, &c_lo, &c_hi);
  const int cty0 = r_lo / TH, ctx0 = c_lo / TW;
  const int ty_n = r_lo < r_hi ? (r_hi - 1) / TH - cty0 + 1 : 0, tx_n = c_lo < c_hi ? (c_hi - 1) / TW - ctx0 + 1 : 0;
  const bool too_many = ty_n > kSpan || tx_n > kSpan;
  int first_ty = -1, first_tx = -1;
  unsigned yv = 0, xv = 0;
  cand_b = 0;
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
  if (too_many) {  // a cell at the grid's edge of a coarse scale (its support takes in the replicate padding): plain loops, as
                   // patch_grad_combine_kernel's second branch -- S3 sums such a cell's tiles one by one, in the same order
    first_ty = first_tx = -1;
    for (int cty = cty0; cty < cty0 + ty_n && first_ty < 0; ++cty)
      if (gi >= lerp_at(ay, cty * TH).i0 && gi <= lerp_at(ay, min(cty * TH + TH, H) - 1).i1) first_ty = cty;
    for (int ctx = ctx0; ctx < ctx0 + tx_n && first_tx < 0; ++ctx)
      if (gj >= lerp_at(ax, ctx * TW).i0 && gj <= lerp_at(ax, min(ctx * TW + TW, W) - 1).i1) first_tx = ctx;
  }
  cand_a = (unsigned)(cty0 & 255) | ((unsigned)(ctx0 & 255) << 8) | (yv << 16) | (xv << 20) |
           ((unsigned)(first_ty == ty && first_tx == tx) << 24) | ((unsigned)too_many << 25);
}

struct Persist {  // one workgroup's iteration-invariant geometry
  int g_first, g_last, beg, end;                   // its slice of the plan (TileRange)
  int gi0, ni, gj0, nj;                            // the block of grid cells tile + apron touch
  int rect_ty0, rect_tx0, rect_ny, rect_nx;        // the tiles whose partial cell gradients that block's cells sum
};

// UNI: the 2-DoF motion model (theta = (trans_x, trans_y), x' = x + dt theta, src/warp.py:364-383) instead of the patch grid: no
// interpolation tables, no cell block; the tiles' partial pairs of d loss / d theta travel as one record per tile, every workgroup
// sums ALL of them (in the order of the four-launch loop's last kernel) and steps the two parameters itself.
// FRAC: the compact plan carries fractional source coordinates (events rectified with a sub-pixel map or already warped by an earlier stage):
// the event loops are the general ones of the compact format, whose groups hold the fractions; the
// patch-grid kernel then sweeps backward into f64 accumulators (the fixed-point sweep's groups hold integer pixels).
// CONTRAST: which contrast the loop maximises -- a template parameter, not an argument: as run-time branches the blur's and the Sobel
// passes' code cost the plain variance loop 1.3 us per iteration (register pressure in the event loop and the gather: 28.0 -> 29.3 us
// at 2 M events) although it never ran.
enum ResidentContrast : int { RC_VARIANCE = 0, RC_BLURRED_VARIANCE = 1, RC_GRADIENT_MAGNITUDE = 2 };
template <int TH, int TW, int HALO, bool UNI, bool FRAC = false, int CONTRAST = RC_VARIANCE>
__global__ void __launch_bounds__(kBlock) cmax_resident_kernel(ResidentArgs a_unused) {
  static_assert(!(UNI && CONTRAST == RC_GRADIENT_MAGNITUDE), "the 2-DoF problem takes the variance contrast (plain or blurred)");
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
  __shared__ int s_imb[2];      // first iteration: events of the fullest tile, of all tiles (units of 64)
  __shared__ double s_mom[4];    // [0] the mean of the IWE of this iteration
  __shared__ double s_hist[4];   // workgroup 0: (sum, mean) of the IWE of the last two iterations, by parity (the loss bookkeeping)
  __shared__ double s_reg[2];    // this tile's regulariser value partial: of this iteration, of the previous one
  __shared__ float s_adam[3];    // step size and sqrt(bias correction 2) of the iteration's Adam step; variance of the previous iteration
  __shared__ double s_red[3 * kWaves];

  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int tile = blockIdx.x;

  // ---- once: interpolation tables of tile + apron, the block of cells they touch, this thread's element of it -------------------
  // per element of the cell block (thread e < 2 ni nj), in LDS rather than in registers that would be live across every phase:
  __shared__ float s_m[kResElems], s_v[kResElems];                   // Adam's exp_avg / exp_avg_sq (the gradient goes to d_theta as it is formed)
  __shared__ float s_gl[2];                                          // UNI: the last gradient
  // which tiles' partials an element's cell sums (cell_candidates): kept for the first kStoredCands elements -- every cell block but
  // that of the finest pyramid scale at 45 x 80 tiles fits --, recomputed per iteration by the waves beyond (whole waves: no divergence)
  __shared__ unsigned s_cand_a[kStoredCands], s_cand_b[kStoredCands];
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
    if constexpr (!UNI)
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
    if constexpr (UNI) {  // theta = (trans_x, trans_y) and its Adam state: two elements, kept by every workgroup
      if (threadIdx.x < 2) {
        s_cells[threadIdx.x] = a.theta[threadIdx.x];
        s_m[threadIdx.x] = a.exp_avg[threadIdx.x], s_v[threadIdx.x] = a.exp_avg_sq[threadIdx.x], s_gl[threadIdx.x] = 0.0f;
      }
      if (threadIdx.x == 0) {
        const TileRange tr = tile_range<FMT_COMPACT>(key_offsets, ev, TH * TW, tiles_x, 1);
        P.g_first = tr.g_first, P.g_last = tr.g_last, P.beg = tr.beg, P.end = tr.end;
        P.gi0 = P.gj0 = 0, P.ni = P.nj = 1;
        P.rect_ty0 = P.rect_tx0 = P.rect_ny = P.rect_nx = 0;
        s_ok = 1;
        s_imb[0] = s_imb[1] = 0;
        s_reg[0] = s_reg[1] = 0.0;
        s_adam[2] = 0.0f;
      }
    } else {
    const int gi0 = s_lerp[0].i0, ni = s_lerp[PH - 1].i1 - gi0 + 1;
    const int gj0 = s_lerp[PH].i0, nj = s_lerp[PH + PW - 1].i1 - gj0 + 1;
    const bool has = (int)threadIdx.x < 2 * ni * nj;  // this thread steps element (ch, gi0 + ci, gj0 + cj) of the cell block
    if (2 * ni * nj > kResElems && threadIdx.x == 0) raise_status(a.status, (unsigned)RES_GEOMETRY);
    const int e_ = has ? (int)threadIdx.x : 0;
    const int ch = e_ / (ni * nj), ci = (e_ - ch * (ni * nj)) / nj, cj = e_ - ch * (ni * nj) - ci * nj;
    const int gi = gi0 + ci, gj = gj0 + cj;
    const int64_t gidx = ((int64_t)ch * ay.g + gi) * ax.g + gj;
    if (has) {
      s_cells[(ch * kGridCells + ci) * kGridCells + cj] = a.theta[gidx];
      s_m[e_ % kResElems] = a.exp_avg[gidx], s_v[e_ % kResElems] = a.exp_avg_sq[gidx];
    }
    if (has && e_ < kStoredCands) {
      unsigned cand_a, cand_b;
      cell_candidates<TH, TW>(ay, ax, gi, gj, H, W, tiles_y, tiles_x, ty, tx, cand_a, cand_b);
      s_cand_a[e_] = cand_a, s_cand_b[e_] = cand_b;
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
      if (P.rect_ny * P.rect_nx > kWave) raise_status(a.status, (unsigned)RES_GEOMETRY);
      s_ok = 1;
      s_imb[0] = s_imb[1] = 0;
      s_reg[0] = s_reg[1] = 0.0;
      s_adam[2] = 0.0f;
    }
    }
    __syncthreads();
  }
  bool done_ok = true;
  int it = 0;

  for (it = 0; it < n_iter; ++it) {
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
      if constexpr (UNI) {
        tile_bound_post(fabsf(s_cells[0]), fabsf(s_cells[1]), sh.bound);
      } else {
        const bool h2 = (int)threadIdx.x < 2 * ninj;
        const int e_ = h2 ? (int)threadIdx.x : 0, ch = e_ / ninj, rem = e_ - ch * ninj, nj = rfl(P.nj), ci = rem / nj, cj = rem - ci * nj;
        const float th = h2 ? s_cells[(ch * kGridCells + ci) * kGridCells + cj] : 0.0f;
        tile_bound_post(h2 && ch == 0 ? fabsf(th) : 0.0f, h2 && ch == 1 ? fabsf(th) : 0.0f, sh.bound);
      }
      if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
      if (threadIdx.x < 9) s_win[threadIdx.x] = 0xffffffffu;   // (a neighbour beyond the image's edge: no window)
      if (threadIdx.x == 0) {
        sh.next = 2 * kWaves;
        sh.chk = 0ull;
        s_spill = 0;
        s_bad = 0;
        s_next = 2 * kWaves;
        s_wmax[0] = s_wmax[1] = 0;
      }
      if constexpr (!UNI)
        if (!(EBOS_ABL & 1)) tile_flow_from_cells<TH, TW, 0>(s_lerp + AP, s_lerp + PH + AP, s_cells, gi0, gj0, s_flow_f);
      __syncthreads();
      win = tile_bound_read<TH, TW, HALO, true>(sh.bound, a.dt_bound);
      // A padded image: every window reaches at least the padding ring (rows pad_h, columns pad_w rounded to the windows' quads), so
      // that the border tile that OWNS a ring pixel -- publishes it, counts its square -- also gathers it, whatever window the
      // neighbour that deposits there chose (round 5 left padded problems to the four launches for this; host-checked: <= HALO)
      win.hr = max(win.hr, a.pad_h);
      win.hc = max(win.hc, (a.pad_w + 3) & ~3);
      EBOS_RSTAMP(1);
      // (own: what this tile's image holds inside the valid region -- its share of sum(IWE), exact; with the blur: of sum(m . B x),
      // position-weighted)
      auto body = [&](auto& own) {
        if (!(EBOS_ABL & 2048))
        tile_body<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, UNI, !UNI, true, false, FRAC>(tr, win, UNI ? s_cells : s_flow_f, s_acc, sh, ev, a.H, a.W,
                                                                                    tiles_x, a.pad_h, a.pad_w, a.slabs, nullptr, nullptr, 0u, nullptr, pre,
                                                                                    NoHook{}, own);
      };
      double os;
      if constexpr (CONTRAST == RC_BLURRED_VARIANCE) {
        OwnSumBlur own{tr.ty * TH - win.HR() + a.pad_h, tr.tx * TW - win.HC() + a.pad_w, a.omit ? 1 : 0, a.H + 2 * a.pad_h, a.W + 2 * a.pad_w, a.blur, 0.0, 0.0};
        body(own);
        os = own.total();
      } else {
        OwnSum own{tr.ty * TH - win.HR() + a.pad_h, tr.tx * TW - win.HC() + a.pad_w, a.omit ? 1 : 0, a.H + 2 * a.pad_h, a.W + 2 * a.pad_w, 0.0};
        body(own);
        os = own.acc;
      }
      EBOS_RSTAMP(2);
      os = wave_sum(os);
      if (lane == 0) s_red[wave] = os;
      drain_stores();
      __syncthreads();
    }
    if (sh.flag[1]) {  // (uniform) a tap left the largest window: the four-launch pipeline's spill path handles that flow
      if (threadIdx.x == 0) raise_status(fresh_args().status, (unsigned)RES_SPILL | ((unsigned)it << 8));
      done_ok = false;
      break;
    }
    // ---- S1: the all-to-all of the iteration: every tile's share of sum(IWE) and its window ------------------------------------
    // The mean of the IWE is all the variance gradient needs of the other tiles, and it does not need the assembled image: sum(IWE)
    // = the sum over tiles of what each tile's LDS image holds inside the valid region -- exact in a double, so the same number as
    // the four-launch pipeline's (kCombineExactSum).  It travels with the slab's flag, BEFORE any halo has been exchanged: the
    // upstream window below is then gathered, mapped and staged in one pass, and the sum of squares (the loss VALUE only) is left
    // to workgroup 0's bookkeeping one iteration later -- one all-to-all and one neighbour hand-off per iteration instead of three.
    constexpr int kQuads = (kLHmax * kLWmax / 4 + kBlock - 1) / kBlock;
    constexpr int kSpecHalo = HALO < 4 ? HALO : 4;
    // (the window of the UPSTREAM image: the four-launch backward kernel stages at least its speculative 4 px window, and the
    // fixed-point unit of the scatter follows max |staged value| -- same window, same unit, same bits)
    const Win<TH, TW, HALO, true> wb = (win.hr <= kSpecHalo && win.hc <= kSpecHalo) ? Win<TH, TW, HALO, true>{kSpecHalo, kSpecHalo} : win;
    // Blurred contrast (iwe.blur_sigma > 0, blur3.h): d loss / d IWE on the window needs the masked blurred image one pixel around it
    // and that the raw image two around it -- the GATHER window wx is the upstream window plus 2 rows / 4 columns (quads) per side.
    // Raw window (at s_g) and blurred window (behind it) share the LDS region of the largest upstream window: windows up to ~12 px at
    // 45 x 80; beyond that -- or where a tile two away reaches into the gather window -- the launch hands over to the pipeline below.
    // Gradient-magnitude contrast (sobel3.h): the same gather window -- Sobel pairs one pixel around the upstream window, the raw image
    // two around it.  gx lives where the d_flow accumulators will (cleared behind the passes instead of in front of them).
    constexpr bool blur_on = CONTRAST == RC_BLURRED_VARIANCE;
    constexpr bool gm_on = CONTRAST == RC_GRADIENT_MAGNITUDE;
    constexpr bool raw_on = blur_on || gm_on;           // the gather stages the RAW window; the contrast's passes follow
    const Win<TH, TW, HALO, true> wx = raw_on ? Win<TH, TW, HALO, true>{wb.hr + 2, wb.hc + 4} : wb;
    float4 own_q[kQuads];  // this workgroup's own contribution to the quads of its upstream window, decoded from its LDS image
    double mean;
    bool halo_complete;
    bool s_ok_local = true;
    {
      KArgs& a = fresh_args();
      const int tiles_x = a.tiles_x, tiles_y = a.tiles_y, ty = tile / tiles_x, tx = tile - ty * tiles_x, n_tiles = tiles_y * tiles_x;
      if (threadIdx.x == 0) {
        double S = 0.0;
        for (int k = 0; k < kWaves; ++k) S += s_red[k];
        unsigned long long* rec = a.rec1 + ((size_t)(it & 1) * n_tiles + tile) * kRec1Granules;
        put_granules(rec, ep, S);
        const unsigned cnt64 = (unsigned)min((rfl(P.end) - rfl(P.beg)) >> 6, 0xffff);   // the tile's events, in units of 64
        st_sc1(rec + 2, ((unsigned long long)ep << 32) | (unsigned long long)(win_pack(win.hr, win.hc) & 0xffffu) | ((unsigned long long)cnt64 << 16));
      }
      EBOS_RSTAMP(3);
      {  // while the records travel: the own part of the gather below (needs nothing of the others)
        // (IMAGE coordinates from here on: the padded image [H, W], the tile's origin shifted by the padding)
        const int H = a.H + 2 * a.pad_h, W = a.W + 2 * a.pad_w, tr0 = ty * TH + a.pad_h, tc0 = tx * TW + a.pad_w;
        const int qw = wx.LW() / 4, n_q = wx.LH() * qw, oy = tr0 - wx.HR(), ox = tc0 - wx.HC();
        const float inv_qw = 1.0f / (float)qw;
        const bool lds_f64 = sh.chk != 0ull;             // (tile_body redid its slice exactly: the LDS image holds doubles)
        const int own_lh = win.LH(), own_pt = win.P(), row0 = tr0 - win.HR(), col0 = tc0 - win.HC(), own_lw = win.LW();
#pragma unroll
        for (int kq = 0; kq < kQuads; ++kq) {
          own_q[kq] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kq * kBlock >= n_q) continue;  // (uniform)
          const int i = threadIdx.x + kq * kBlock;
          const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
          const int r = oy + rl, c = ox + 4 * cq, rr = r - row0, cc = c - col0;
          // (a quad may straddle the image's left / right edge when pad_w is no multiple of 4: its cells outside the image hold zeros)
          const bool ok = i < n_q && r >= 0 && r < H && c + 3 >= 0 && c < W && (unsigned)rr < (unsigned)own_lh && (unsigned)cc < (unsigned)own_lw;
          if (EBOS_ABL & 8) continue;
          const float4 v = lds_image_cells4(s_acc, own_lh, own_pt, ok ? rr : 0, ok ? cc >> 2 : 0, lds_f64);
          if (ok) own_q[kq] = v;
        }
      }
      double as = 0.0;
      if (wave * kWave < n_tiles) {
        const int k = wave * kWave + lane;
        const unsigned long long* rec = a.rec1 + ((size_t)(it & 1) * n_tiles + min(k, n_tiles - 1)) * kRec1Granules;
        unsigned long long g[3];
        const bool ok = wave_wait([&]() {
          bool all = true;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            g[j] = ld_sc1(rec + j);
            all = all && (unsigned)(g[j] >> 32) == ep;
          }
          return all || (EBOS_ABL & 256) != 0;
        }, a.status, a.cap_ticks, it);
        if (ok && k < n_tiles) {
          as = __builtin_bit_cast(double, (g[0] & 0xffffffffull) | (g[1] << 32));
          const int nty = k / tiles_x, dy = nty - ty, dx = k - nty * tiles_x - tx;
          if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) s_win[(dy + 1) * 3 + dx + 1] = (unsigned)g[2] & 0xffffu;  // a neighbour's window (or this tile's)
        }
        if (it == 0) {  // once: is one tile far more crowded than the average one?  (every workgroup sees every count: a uniform verdict)
          const float c64 = ok && k < n_tiles ? (float)(((unsigned)g[2] >> 16) & 0xffffu) : 0.0f;
          const float cmax = wave_max_nonneg(c64), csum = wave_sum(c64);
          if (lane == 0) {
            atomicMax(&s_imb[0], (int)cmax);
            atomicAdd(&s_imb[1], (int)csum);
          }
        }
        const float mh = wave_max_nonneg(ok ? (float)((unsigned)g[2] & 255u) : 255.0f);
        const float mw = wave_max_nonneg(ok ? (float)(((unsigned)g[2] >> 8) & 255u) : 255.0f);
        if (lane == 0) {
          atomicMax(&s_wmax[0], (int)mh);
          atomicMax(&s_wmax[1], (int)mw);
          if (!ok) s_ok = 0;
        }
      } else if (threadIdx.x == kBlock - kWave) {
        // (an idle wave: Adam's bias corrections of this iteration's step, as torch computes them -- adam_coef, patch_grid.h)
        const AdamCoef coef = adam_coef(a.lr, a.beta1, a.beta2, a.t0 + it + 1);
        s_adam[0] = coef.step_size;
        s_adam[1] = coef.bc2_sqrt;
      }
      EBOS_RSTAMP(4);
      as = wave_sum(as);
      if (lane == 0) s_red[kWaves + wave] = as;
      __syncthreads();
      if (threadIdx.x == 0) {
        const int lo_px = a.omit ? 1 : 0;
        const double n_px = (double)max(a.H + 2 * a.pad_h - 2 * lo_px, 0) * (double)max(a.W + 2 * a.pad_w - 2 * lo_px, 0);
        double S = 0.0;
        for (int k = 0; k < kWaves; ++k) S += s_red[kWaves + k];
        const double mn = n_px > 0.0 ? S / n_px : 0.0;
        s_mom[0] = mn;
        s_hist[(it & 1) * 2] = S;       // (workgroup 0's bookkeeping reads them one iteration later)
        s_hist[(it & 1) * 2 + 1] = mn;
      }
      __syncthreads();
      mean = s_mom[0];
      halo_complete = 2 * s_wmax[0] < TH && 2 * s_wmax[1] < TW;  // (uniform over the GRID: every workgroup saw every window)
      if constexpr (raw_on) {
        // (uniform over the grid, from the LARGEST window: every gather window lies within the 3 x 3 tiles around its own, no tile
        // two away reaches into it, and the raw + blurred windows fit the LDS region -- with the Sobel pairs: raw window + gy there,
        // gx in the d_flow accumulators' region)
        const int hrm = max(s_wmax[0], kSpecHalo), hcm = max(s_wmax[1], kSpecHalo);
        const int lh = TH + 2 * hrm, lw = TW + 2 * hcm;
        const bool fits = 2 * hrm + 2 <= TH && 2 * hcm + 4 <= TW &&
                          (lh + 4) * (lw + 8) + (lh + 2) * (lw + 8) <= kLHmax * kLWmax && (!gm_on || (lh + 2) * (lw + 8) <= 4 * TH * TW);
        if (!fits || !halo_complete) {
          if (threadIdx.x == 0) raise_status(a.status, (unsigned)RES_SPILL | ((unsigned)it << 8));
          s_ok_local = false;
        }
      }
      // (>= 32 k events on the fullest tile: below that nothing is slow.  Second rule, from profiles/r05m_skew_solver.json: a fullest
      // tile of >= 120 k events that is more than 3 x the average one -- 10 M events in a blob of sigma 200 px: 242 k, 6.2 x -- took
      // 315 us per iteration here against 116 with the pipeline's split tiles; at 2 M events the same blob, 48 k, still wins here)
      // Round 6, with both sides measured again (profiles/r06t_crowding_rule.json; the accumulate loop merges same-cell events now,
      // the launches skip empty tiles): on a sensor of >= 128 tiles (1280 x 720) the launches win once the fullest tile holds more than
      // 60 k + 0.8 % of the window's events -- 1 M events: ~68 k, 2 M: ~76 k, 5 M: ~100 k --; the two ratio rules refused a 1 M-event
      // window at 12 x (41 k events: 45.7 us here against 58.7) and kept a 2 M-event one at 10.7 x (84 k: 68.3 against 63.8).
      // On a small sensor (346 x 260: 99 tiles of 32 x 32) the launches are slower and the cross-over sits at ~85 k events on the
      // fullest tile whatever the window (100 k – 1 M events); the ratio rules had refused a 100 k-event window at 41 x (48.2 us here
      // against 57.7) and kept a 1 M-event one at 11.4 x (115 k events: 98.2 against 69.7).
      // max_imbalance > 0 (EBOS_RESIDENT_MAX_IMBALANCE): the ratio rules with that ratio.
      const float imb_ratio = (float)s_imb[0] * (float)n_tiles / fmaxf((float)s_imb[1], 1.0f);
      const bool by_ratio = s_imb[0] >= 512 && (imb_ratio > a.max_imbalance || (s_imb[0] >= 1875 && imb_ratio > 0.25f * a.max_imbalance));
      const bool by_count = n_tiles >= 128 ? (float)s_imb[0] > 60000.0f / 64.0f + 0.008f * (float)s_imb[1] : s_imb[0] > 85000 / 64;
      const bool crowded = a.max_imbalance < 0.0f ? by_count : by_ratio;
      if (it == 0 && a.max_imbalance != 0.0f && crowded) {
        if (threadIdx.x == 0) raise_status(a.status, (unsigned)RES_IMBALANCED);
        s_ok_local = false;
      }
    }
    if (!s_ok_local) { done_ok = false; break; }
    if (!s_ok) { done_ok = false; break; }
    EBOS_RSTAMP(5);
    // ---- G: the UPSTREAM WINDOW (this tile + the halo the backward sweep reads): per pixel the sum of the slabs whose windows reach
    // it, in the combine pass's order (same bits as the four-launch image), mapped to d loss / d IWE = 2 (-w) (IWE - mean) / (M - 1)
    // and stored to its place in LDS in the same pass.  This workgroup's own contribution is decoded from its LDS image (what it stored
    // to its slab, without the round trip); the neighbours' come from their slabs.  A halo pixel is complete with the 3 x 3 tiles
    // around THIS tile as long as no tile two away reaches it: hr + hr' < TH, hc + hc' < TW for any two windows -- known for the whole
    // grid from the records above; BOS-sized flows pass, and nothing of the image then travels through memory.  Otherwise the raw
    // window is stored, the tiles publish their images and the halo is staged from those, as the four-launch backward kernel does.
    const bool publish = !halo_complete || it == n_iter - 1;  // (the image leaves the kernel in its last iteration)
    BwdPreRaw pre_raw;  // the backward sweep's first two chunks per wave, requested here and decoded behind the barrier
    {
      KArgs& a = fresh_args();
      // (IMAGE coordinates: the padded image [H, W]; tile (ty, tx) starts at (tr0, tc0) in it, a neighbour's window at its tile's
      // origin + padding - its halo)
      const int pad_h = a.pad_h, pad_w = a.pad_w;
      const int H = a.H + 2 * pad_h, W = a.W + 2 * pad_w, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x,
                tr0 = ty * TH + pad_h, tc0 = tx * TW + pad_w;
      // this tile's OWN pixels (it publishes them and counts their squares): its tile, and for a tile on the border the padding ring
      // beside it (and, as before, nothing beyond the image for a last tile that is cut)
      const int or0 = ty == 0 ? 0 : tr0, or1 = ty == a.tiles_y - 1 ? H : tr0 + TH;
      const int oc0 = tx == 0 ? 0 : tc0, oc1 = tx == tiles_x - 1 ? W : tc0 + TW;
      const int lo_px = a.omit ? 1 : 0;
      const double n_px = (double)max(H - 2 * lo_px, 0) * (double)max(W - 2 * lo_px, 0);
      const double ga = 2.0 * (-(double)a.w_contrast) / (n_px - 1.0);
      const float Ga = (float)ga, Gc = (float)(-ga * mean);
      const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(a.slabs, 0xffffffffu);
      const int qw = wx.LW() / 4, n_q = wx.LH() * qw, oy = tr0 - wx.HR(), ox = tc0 - wx.HC();
      const float inv_qw = 1.0f / (float)qw;
      // (an interior tile's window lies inside the valid region as a whole: no per-pixel tests -- vector instruction issue, not
      // memory, bounds these passes)
      const bool all_valid = oy >= lo_px && oy + wx.LH() <= H - lo_px && ox >= lo_px && ox + wx.LW() <= W - lo_px;
      float* iwe = a.iwe;
      double sq = 0.0;
      float gmax_t = 0.0f, gsum_t = 0.0f;  // (max and sum of |staged value|: the scatter's fixed-point unit, bwd_fx_unit)
      // one quad of the window, assembled: the sum of squares of this tile's own pixels, the image itself when it leaves, the
      // affine map and the quad's place in LDS
      auto finish_quad = [&](int i, int r, int c, bool in, bool live, const float4& v) {
        // this tile's own pixels (quads lie inside a tile as a whole or outside it; in the padding ring beside a border tile a quad
        // may straddle the image's edge: per pixel there)
        if (live && r >= or0 && r < or1 && c + 3 >= oc0 && c < oc1) {
          const float e4[4] = {v.x, v.y, v.z, v.w};
          if (!raw_on && r >= lo_px && r < H - lo_px) {  // (blurred contrast: the sum of squares is the blurred image's, below)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c + k >= max(lo_px, oc0) && c + k < min(W - lo_px, oc1)) sq += (double)e4[k] * (double)e4[k];
          }
          if (publish) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c + k >= oc0 && c + k < oc1) st_sc1(iwe + (int64_t)r * W + c + k, e4[k]);
          }
        }
        float4 gq = v;
        if (raw_on) {
          if (!live) gq = make_float4(0.f, 0.f, 0.f, 0.f);  // (the RAW window is staged: zero outside the image)
        } else if (halo_complete) {
          if (all_valid) {
            gq = make_float4(Ga * v.x + Gc, Ga * v.y + Gc, Ga * v.z + Gc, Ga * v.w + Gc);
          } else {
            const float e4[4] = {v.x, v.y, v.z, v.w};
            float o4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const bool valid = r >= lo_px && r < H - lo_px && c + k >= lo_px && c + k < W - lo_px;
              o4[k] = valid ? Ga * e4[k] + Gc : 0.0f;
            }
            gq = make_float4(o4[0], o4[1], o4[2], o4[3]);
          }
          if (in) {
            const float m4 = fmaxf(fmaxf(fabsf(gq.x), fabsf(gq.y)), fmaxf(fabsf(gq.z), fabsf(gq.w)));
            gmax_t = fmaxf(gmax_t, (gq.x + gq.y + gq.z + gq.w) == (gq.x + gq.y + gq.z + gq.w) ? m4 : INFINITY);  // (a NaN anywhere: Inf)
            gsum_t += (fabsf(gq.x) + fabsf(gq.y)) + (fabsf(gq.z) + fabsf(gq.w));
          }
        }
        if (in) reinterpret_cast<float4*>(s_g)[i] = gq;   // (the LDS image is dead: every thread decoded its own_q before S1's barriers)
      };
      // what needs nothing of the other tiles: the d_flow accumulators are cleared, the tile's flow with its apron is evaluated, the
      // sweep's first chunks are requested -- placed between the request of the neighbours' slabs and their first use
      auto independent_work = [&]() {
        if (!(EBOS_ABL & 4) && !gm_on)  // (gradient magnitude: the Sobel pairs use the region first)
        for (int i = threadIdx.x; i < TH * TW; i += kBlock) reinterpret_cast<double2*>(s_d)[i] = make_double2(0.0, 0.0);  // [2][TH * TW]
        if constexpr (!UNI)
          if (!(EBOS_ABL & 2)) tile_flow_from_cells<TH, TW, AP>(s_lerp, s_lerp + PH, s_cells, rfl(P.gi0), rfl(P.gj0), s_flow_b);
        TileRange trp;
        trp.ty = trp.tx = 0, trp.slab = tile, trp.part = 0;
        trp.g_first = rfl(P.g_first), trp.g_last = rfl(P.g_last), trp.beg = rfl(P.beg), trp.end = rfl(P.end);
        const EvPtrs evp = a.ev;
        pre_raw.A = load_craw(trp.g_first + wave * kWave + lane, trp, evp);
        pre_raw.B = load_craw(trp.g_first + (wave + kWaves) * kWave + lane, trp, evp);
      };
      if (EBOS_ABL & 8192) {
        independent_work();   // (timing build: no gather, no affine map, nothing staged)
      } else if (halo_complete) {
        // No window is as large as half a tile: a pixel lies in the windows of at most 2 x 2 tiles -- the pair of tile rows
        // (ty - 1, ty) or (ty, ty + 1) by the half of the tile its row is in (or beyond), likewise for columns -- and this tile is one
        // of the four.  So a quad has at most THREE slab loads, all of a thread's are in flight before the first is used (as nine
        // candidates behind per-candidate branches, the second round of quads waited for the first: two memory round trips), and
        // the additions keep the combine pass's order (tile row, tile column).
        // Rounds of quads go in pairs: the loads of a pair are requested, the work that needs nothing of the other tiles runs while
        // they travel (first pair only), then the pair is assembled -- all rounds at once held 64 registers of loads across that work
        // and spilled.
        constexpr int kPair = 2;
#pragma unroll
        for (int k0 = 0; k0 < kQuads; k0 += kPair) {
          float4 ld[kPair][4];
          unsigned meta[kPair];  // bits 0..3: slot s contributes; bits 4..5: the slot that is this tile
#pragma unroll
          for (int kk = 0; kk < kPair; ++kk) {
            const int kq = k0 + kk;
            meta[kk] = 0u;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) ld[kk][sl] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kq >= kQuads || kq * kBlock >= n_q) continue;  // (uniform: a small window has fewer quads than the largest one's kQuads per thread)
            const int i = threadIdx.x + kq * kBlock;
            const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
            const int r = oy + rl, c = ox + 4 * cq;
            const bool live = i < n_q && r >= 0 && r < H && c + 3 >= 0 && c < W;
            const int ya = r < tr0 + (TH + 1) / 2 ? ty - 1 : ty, xa = c < tc0 + (TW + 1) / 2 ? tx - 1 : tx;
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
              const int nty = ya + (sl >> 1), ntx = xa + (sl & 1);
              const unsigned w = s_win[(nty - ty + 1) * 3 + (ntx - tx + 1)];   // (0xffffffff: no such tile)
              const int hr = (int)(w & 255u), hc = (int)((w >> 8) & 255u);
              const int rr = r - (nty * TH + pad_h - hr), cc = c - (ntx * TW + pad_w - hc), lw = TW + 2 * hc, lh = TH + 2 * hr;
              const bool is_own = nty == ty && ntx == tx;
              const bool ok = live && w != 0xffffffffu && (unsigned)rr < (unsigned)lh && (unsigned)cc < (unsigned)lw;
              meta[kk] |= (ok ? 1u : 0u) << sl;
              if (is_own) meta[kk] |= (unsigned)sl << 4;
              const bool need = ok && !is_own && !(EBOS_ABL & 64);
              if (__builtin_amdgcn_ballot_w64(need) != 0ull) {  // (most waves hold no quad a given neighbour reaches)
                const unsigned slab0 = (unsigned)(nty * tiles_x + ntx) * (unsigned)(kLHmax * kLWmax);
                ld[kk][sl] = slab_load4(all_slabs, need ? (slab0 + (unsigned)(rr * lw + cc)) * 4u : 0u);
              }
            }
          }
          if (k0 == 0) independent_work();
#pragma unroll
          for (int kk = 0; kk < kPair; ++kk) {
            const int kq = k0 + kk;
            if (kq >= kQuads || kq * kBlock >= n_q) continue;
            const int i = threadIdx.x + kq * kBlock;
            const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
            const int r = oy + rl, c = ox + 4 * cq;
            const bool in = i < n_q;
            const bool live = in && r >= 0 && r < H && c + 3 >= 0 && c < W;
            const unsigned own_slot = (meta[kk] >> 4) & 3u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
              const float4 part = own_slot == (unsigned)sl ? own_q[kq < kQuads ? kq : 0] : ld[kk][sl];
              if ((meta[kk] >> sl) & 1u) v.x += part.x, v.y += part.y, v.z += part.z, v.w += part.w;
            }
            finish_quad(i, r, c, in, live, v);
          }
        }
      } else {
#pragma unroll
      for (int kq = 0; kq < kQuads; ++kq) {
        if (kq * kBlock >= n_q) continue;  // (uniform)
        const int i = threadIdx.x + kq * kBlock;
        const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
        const int r = oy + rl, c = ox + 4 * cq;
        const bool in = i < n_q;
        const bool live = in && r >= 0 && r < H && c + 3 >= 0 && c < W;
        float4 part[9];
        bool okk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          // (what depends on the candidate only is uniform: scalar registers and the scalar unit)
          const unsigned w = (unsigned)rfl((int)s_win[k]);
          const int nty = ty + k / 3 - 1, ntx = tx + k % 3 - 1;
          const int hr = (int)(w & 255u), hc = (int)((w >> 8) & 255u);
          const int row0 = nty * TH + pad_h - hr, col0 = ntx * TW + pad_w - hc, lw = TW + 2 * hc, lh = TH + 2 * hr;
          const unsigned slab0 = (unsigned)(nty * tiles_x + ntx) * (unsigned)(kLHmax * kLWmax);
          const int rr = r - row0, cc = c - col0;
          okk[k] = live && w != 0xffffffffu && (unsigned)rr < (unsigned)lh && (unsigned)cc < (unsigned)lw;
          part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k == 4) {  // this workgroup's own image: decoded from LDS above
            part[k] = own_q[kq];
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
        finish_quad(i, r, c, in, live, v);
      }
      independent_work();
      }
      // pixels of the rectangle [r0, r0 + nr) x [c0, c0 + nc) OUTSIDE the box [lo_b, H - lo_b) x [lo_b, W - lo_b): top and bottom
      // strips whole, left and right strips between them; f(r, c) once per pixel, dealt to the threads
      // (ONE loop over both kinds of strip, the row of an item from a reciprocal: the few waves that hold border pixels run a long
      // chain of instructions while the others wait at the next barrier -- as two loops with integer divisions the chain was twice as long)
      auto for_border = [&](int r0, int nr, int c0, int nc, int lo_b, auto&& f) {
        const int rt = min(max(lo_b - r0, 0), nr), rb = min(max(r0 + nr - (H - lo_b), 0), nr - rt);   // rows above / below the box
        const int cl_ = min(max(lo_b - c0, 0), nc), cr_ = min(max(c0 + nc - (W - lo_b), 0), nc - cl_);  // columns left / right of it
        const int n_tb = (rt + rb) * nc, nm = nr - rt - rb, ns = cl_ + cr_;
        const float inv_nc = 1.0f / (float)max(nc, 1), inv_ns = 1.0f / (float)max(ns, 1);
        auto div = [](int i, int n, float inv) {  // i / n for 0 <= i < 2^22
          int k = (int)(((float)i + 0.5f) * inv);
          const int rem = i - k * n;
          k += rem >= n ? 1 : (rem < 0 ? -1 : 0);
          return k;
        };
        for (int i = threadIdx.x; i < n_tb + nm * ns; i += kBlock) {
          if (i < n_tb) {
            const int k = div(i, nc, inv_nc), c = c0 + i - k * nc;
            f(k < rt ? r0 + k : r0 + nr - rb + (k - rt), c);
          } else {
            const int j = i - n_tb, k = div(j, ns, inv_ns), q = j - k * ns;
            f(r0 + rt + k, q < cl_ ? c0 + q : c0 + nc - cr_ + (q - cl_));
          }
        }
      };
      // the same pixels in two loops -- what lies outside the image (many pixels, `outside(r, c)` is trivial) and the ring inside it (a
      // few hundred pixels at most: `ring(r, c)`, the general forms) -- when one loop would take more than one round: the ring's
      // pixels are then spread over the rounds of the large outside part and a wave runs the general form in several of them (3.5 us
      // per pass in a 45 x 80 border tile); a small window keeps the one loop (a second loop's set-up costs it 0.3 us)
      // (band_r / band_c: how far outside the image the pass's main loop can have left something to clear -- one pixel for a 3 x 3
      // stencil on a window that is zero outside the image; farther out nothing is visited: 0.6 us of a 45 x 80 border tile's
      // iteration went into clearing zeros)
      auto for_border_split = [&](int r0, int nr, int c0, int nc, int lo_b, int band_r, int band_c, auto&& outside, auto&& ring) {
        {
          const int r1 = min(r0 + nr, H + band_r), c1 = min(c0 + nc, W + band_c);
          r0 = max(r0, -band_r), c0 = max(c0, -band_c);
          nr = max(r1 - r0, 0), nc = max(c1 - c0, 0);
        }
        const int rt = min(max(lo_b - r0, 0), nr), rb = min(max(r0 + nr - (H - lo_b), 0), nr - rt);
        const int cl_ = min(max(lo_b - c0, 0), nc), cr_ = min(max(c0 + nc - (W - lo_b), 0), nc - cl_);
        if ((rt + rb) * nc + (nr - rt - rb) * (cl_ + cr_) <= kBlock) {
          for_border(r0, nr, c0, nc, lo_b, [&](int r, int c) {
            if (r >= 0 && r < H && c >= 0 && c < W) ring(r, c);
            else outside(r, c);
          });
          return;
        }
        if (!(EBOS_ABL & 524288)) for_border(r0, nr, c0, nc, 0, outside);
        const int rr0 = max(r0, 0), rr1 = min(r0 + nr, H), cc0 = max(c0, 0), cc1 = min(c0 + nc, W);
        if (rr1 > rr0 && cc1 > cc0) for_border(rr0, rr1 - rr0, cc0, cc1 - cc0, lo_b, ring);
      };
      if constexpr (blur_on) {
        // ---- blurred contrast: raw window (s_g, [wx]) -> masked blurred window (behind it: upstream window + 1 row / 4 columns per
        // side, i.e. the raw window's columns) -> upstream window a z + c wgt, z = B^T (m . B x) (s_g again, [wb]): the arithmetic of
        // the pipeline's image pass and backward staging (blur3.h, GradImage::map), on what this workgroup gathered.  The passes are
        // bound by instruction issue, and the iteration by its slowest tile: EVERY tile takes the interior form four pixels per
        // thread and step (shared column sums, 16-byte LDS accesses; next to the image's border it reads the window's zeros and
        // computes values nobody wants); a tile on the border then recomputes its few hundred pixels beside the border with the
        // general form, one per thread -- as a per-pixel choice inside the main loop every wave of a border tile ran both forms and
        // those 60 of 256 tiles took 17 us where the others took 9 (in-kernel stamps).
        const Blur3 bk = a.blur;
        const int xw = wx.LW(), xoy = tr0 - wx.HR(), xox = tc0 - wx.HC();
        const int bh = wb.LH() + 2, bw = xw, boy = xoy + 1, box = xox, bq = bw / 4;
        float* s_b = s_g + wx.LH() * xw;
        const int vlo = max(lo_px, 1);
        __syncthreads();
        {
          const bool inner = boy >= vlo && boy + bh <= H - vlo && box + 3 >= vlo && box + bw - 3 <= W - vlo;
          const float inv_q = 1.0f / (float)(bq - 2);
          const float4* A4 = reinterpret_cast<const float4*>(s_g);
          auto x_at = [&](int r, int c) { return s_g[(r - xoy) * xw + (c - xox)]; };
          if (!(EBOS_ABL & 16384))
          for (int i = threadIdx.x; i < bh * (bq - 2); i += kBlock) {   // (the window's outermost quads: one column each, below)
            const int rl = (int)(((float)i + 0.5f) * inv_q), cq = i - rl * (bq - 2) + 1;
            const float4* m = A4 + (rl + 1) * bq + cq;
            const float4 y4 = blur3_interior_quad(m[-bq - 1], m[-bq], m[-bq + 1], m[-1], m[0], m[1], m[bq - 1], m[bq], m[bq + 1], bk);
            reinterpret_cast<float4*>(s_b)[rl * bq + cq] = y4;
            const int r = boy + rl, c = box + 4 * cq;
            if (r >= or0 && r < or1 && c + 3 >= oc0 && c < oc1) {   // this tile's own pixels, the padding ring beside a border tile included (the border's: below)
              if (inner || (r >= vlo && r < H - vlo && c >= vlo && c + 3 < W - vlo)) {
                sq += ((double)y4.x * (double)y4.x + (double)y4.y * (double)y4.y) + ((double)y4.z * (double)y4.z + (double)y4.w * (double)y4.w);
              } else if (r >= vlo && r < H - vlo) {
                const float e4[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (c + k >= vlo && c + k < W - vlo) sq += (double)e4[k] * (double)e4[k];
              }
            }
          }
          for (int i = threadIdx.x; i < 2 * bh; i += kBlock) {   // the column left / right of the upstream window
            const int rl = i >> 1, cl = (i & 1) ? bw - 4 : 3;
            s_b[rl * bw + cl] = blur3_interior(x_at, boy + rl, box + cl, bk);
          }
          if (!inner && !(EBOS_ABL & 65536)) {  // (uniform) the pixels beside the image's border, and those outside the valid region
            __syncthreads();
            for_border_split(boy, bh, box + 3, bw - 6, vlo, 1, 1, [&](int r, int c) { s_b[(r - boy) * bw + (c - box)] = 0.0f; },
                             [&](int r, int c) {
                               const bool valid = r >= lo_px && r < H - lo_px && c >= lo_px && c < W - lo_px;
                               const float y = (valid && !(EBOS_ABL & 262144)) ? blur3_fwd_at_dense(x_at, r, c, H, W, bk) : 0.0f;  // (the raw window reaches >= 1 pixel further)
                               s_b[(r - boy) * bw + (c - box)] = y;
                               if (r >= or0 && r < or1 && c >= oc0 && c < oc1) sq += (double)y * (double)y;
                             });
          }
        }
        __syncthreads();
        {
          const int gw = wb.LW(), goy = tr0 - wb.HR(), gox = tc0 - wb.HC(), gq = gw / 4;
          const int ilo = lo_px + 2;
          const bool inner = goy >= ilo && goy + wb.LH() <= H - ilo && gox >= ilo && gox + gw <= W - ilo;
          float wi = 0.0f;  // (blur3_weight's interior value, its additions in its order)
          wi += bk.k0, wi += bk.k1, wi += bk.k0;
          const float cw_in = Gc * (wi * wi);
          const float inv_q = 1.0f / (float)gq;
          const float4* B4 = reinterpret_cast<const float4*>(s_b);
          if (!(EBOS_ABL & 32768))
          for (int i = threadIdx.x; i < wb.LH() * gq; i += kBlock) {
            const int rl = (int)(((float)i + 0.5f) * inv_q), cq = i - rl * gq;
            const float4* m = B4 + (rl + 1) * bq + cq + 1;
            const float4 z4 = blur3_interior_quad(m[-bq - 1], m[-bq], m[-bq + 1], m[-1], m[0], m[1], m[bq - 1], m[bq], m[bq + 1], bk);
            // (GradImage::map with the interior weight)
            const int r = goy + rl, c = gox + 4 * cq;
            // (a quad that lies outside the image as a whole stages zeros -- the constant term would fill it otherwise, and the
            // fix-up below then has only the quads across the image's left / right edge to clear)
            const bool in_img = inner || (r >= 0 && r < H && c + 3 >= 0 && c < W);
            const float4 g4 = in_img ? make_float4(__fmaf_rn(Ga, z4.x, cw_in), __fmaf_rn(Ga, z4.y, cw_in), __fmaf_rn(Ga, z4.z, cw_in), __fmaf_rn(Ga, z4.w, cw_in))
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(s_g)[i] = g4;
            if (inner || (r >= ilo && r < H - ilo && c >= ilo && c + 3 < W - ilo)) {
              const float m4 = fmaxf(fmaxf(fabsf(g4.x), fabsf(g4.y)), fmaxf(fabsf(g4.z), fabsf(g4.w)));
              gmax_t = fmaxf(gmax_t, (g4.x + g4.y + g4.z + g4.w) == (g4.x + g4.y + g4.z + g4.w) ? m4 : INFINITY);
              gsum_t += (fabsf(g4.x) + fabsf(g4.y)) + (fabsf(g4.z) + fabsf(g4.w));
            } else if (r >= ilo && r < H - ilo) {
              const float e4[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (c + k >= ilo && c + k < W - ilo) {
                  gmax_t = fmaxf(gmax_t, e4[k] == e4[k] ? fabsf(e4[k]) : INFINITY);
                  gsum_t += fabsf(e4[k]);
                }
            }
          }
          if (!inner && !(EBOS_ABL & 131072)) {  // (uniform) beside the border: folded coefficients, position-dependent weights; outside the image: 0
            GradImage Gm;  // (only map() is used: the staged value of a pixel from z and its position)
            Gm.g = nullptr, Gm.a = Ga, Gm.c = Gc, Gm.h = H, Gm.w = W, Gm.lo = lo_px;
            Gm.set_blur(bk);
            auto u_at = [&](int r, int c) { return s_b[(r - boy) * bw + (c - box)]; };
            __syncthreads();
            for_border_split(goy, wb.LH(), gox, gw, ilo, 0, 3, [&](int r, int c) { s_g[(r - goy) * gw + (c - gox)] = 0.0f; },   // (quads outside the image: cleared by the loop above)
                             [&](int r, int c) {   // the ring of ilo pixels inside the image
                               const float gv = Gm.map(blur3_adj_at_dense(u_at, r, c, H, W, bk), r, c);  // (s_b: one row / four columns further)
                               s_g[(r - goy) * gw + (c - gox)] = gv;
                               gmax_t = fmaxf(gmax_t, gv == gv ? fabsf(gv) : INFINITY);
                               gsum_t += fabsf(gv);
                             });
          }
        }
      }
      if constexpr (gm_on) {
        // ---- gradient-magnitude contrast: raw window (s_g, [wx]) -> Sobel pairs on upstream window + 1 row / 4 columns per side
        // (gx where the d_flow accumulators will be, gy behind the raw window) -> upstream window s . gather of the nine stencils around each pixel (s_g again,
        // [wb]): the arithmetic of the pipeline's image pass (gradmag_fused_kernel through sobel3.h) on what this workgroup gathered.
        // As with the blur, EVERY tile runs the plain quad form over its whole window; a tile at the image's border first fills the
        // ring one pixel outside the image with the border's pixels (replicate padding), afterwards clears the pairs of the stencils
        // outside the valid region and recomputes the image's outermost ring with the folded form, one pixel per thread.
        const int xw = wx.LW(), xh = wx.LH(), xoy = tr0 - wx.HR(), xox = tc0 - wx.HC();
        const int bh = wb.LH() + 2, bw = xw, boy = xoy + 1, box = xox, bq = bw / 4;
        float* s_gx = reinterpret_cast<float*>(s_d);   // (windows up to ~12 px at 45 x 80 and 32 x 32: as the blur's)
        float* s_gy = s_g + xh * xw;
        __syncthreads();
        if (xoy < 0 || xoy + xh > H || xox < 0 || xox + xw > W) {  // (uniform) the raw window leaves the image
          for (int i = threadIdx.x; i < 2 * (xw + xh); i += kBlock) {
            int r, c;
            if (i < 2 * xw) {
              r = (i & 1) ? H : -1, c = xox + (i >> 1);
            } else {
              const int k = i - 2 * xw;
              c = (k & 1) ? W : -1, r = xoy + (k >> 1);
            }
            if (r >= xoy && r < xoy + xh && c >= xox && c < xox + xw && r >= -1 && r <= H && c >= -1 && c <= W)
              s_g[(r - xoy) * xw + (c - xox)] = s_g[(min(max(r, 0), H - 1) - xoy) * xw + (min(max(c, 0), W - 1) - xox)];
          }
          __syncthreads();
        }
        {
          const bool inner = boy >= lo_px && boy + bh <= H - lo_px && box + 3 >= lo_px && box + bw - 3 <= W - lo_px;
          const float inv_q = 1.0f / (float)(bq - 2);
          const float4* A4 = reinterpret_cast<const float4*>(s_g);
          for (int i = threadIdx.x; i < bh * (bq - 2); i += kBlock) {   // (the window's outermost quads: one column each, below)
            const int rl = (int)(((float)i + 0.5f) * inv_q), cq = i - rl * (bq - 2) + 1;
            const float4* m = A4 + (rl + 1) * bq + cq;
            float4 x4, y4;
            sobel3_pair_quad(m[-bq - 1], m[-bq], m[-bq + 1], m[-1], m[0], m[1], m[bq - 1], m[bq], m[bq + 1], x4, y4);
            reinterpret_cast<float4*>(s_gx)[rl * bq + cq] = x4;
            reinterpret_cast<float4*>(s_gy)[rl * bq + cq] = y4;
            const int r = boy + rl, c = box + 4 * cq;
            if (r >= or0 && r < or1 && c + 3 >= oc0 && c < oc1) {   // this tile's own stencils (with the padding ring beside a border tile): its share of the contrast's value
              const float e4[4] = {sobel3_energy(x4.x, y4.x), sobel3_energy(x4.y, y4.y), sobel3_energy(x4.z, y4.z), sobel3_energy(x4.w, y4.w)};
              if (inner || (r >= lo_px && r < H - lo_px && c >= lo_px && c + 3 < W - lo_px)) {
                sq += ((double)e4[0] + (double)e4[1]) + ((double)e4[2] + (double)e4[3]);
              } else if (r >= lo_px && r < H - lo_px) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  if (c + k >= lo_px && c + k < W - lo_px) sq += (double)e4[k];
              }
            }
          }
          for (int i = threadIdx.x; i < 2 * bh; i += kBlock) {   // the column left / right of the upstream window
            const int rl = i >> 1, cl = (i & 1) ? bw - 4 : 3;
            const float* p = s_g + (rl + 1) * xw + cl;
            float vx, vy;
            sobel3_pair(p[-xw - 1], p[-xw], p[-xw + 1], p[-1], p[1], p[xw - 1], p[xw], p[xw + 1], vx, vy);
            s_gx[rl * bw + cl] = vx, s_gy[rl * bw + cl] = vy;
          }
          if (!inner) {  // (uniform) stencils outside the valid region count zero
            __syncthreads();
            for_border(boy, bh, box + 3, bw - 6, lo_px, [&](int r, int c) {
              s_gx[(r - boy) * bw + (c - box)] = 0.0f;
              s_gy[(r - boy) * bw + (c - box)] = 0.0f;
            });
          }
        }
        __syncthreads();
        {
          const int gw = wb.LW(), goy = tr0 - wb.HR(), gox = tc0 - wb.HC(), gq = gw / 4;
          const bool inner = goy >= 1 && goy + wb.LH() <= H - 1 && gox >= 1 && gox + gw <= W - 1;
          const float scale = sobel3_adjoint_scale(-(double)a.w_contrast, n_px);
          const float inv_q = 1.0f / (float)gq;
          const float4* X4 = reinterpret_cast<const float4*>(s_gx);
          const float4* Y4 = reinterpret_cast<const float4*>(s_gy);
          for (int i = threadIdx.x; i < wb.LH() * gq; i += kBlock) {
            const int rl = (int)(((float)i + 0.5f) * inv_q), cq = i - rl * gq;
            const float4* x = X4 + (rl + 1) * bq + cq + 1;
            const float4* y = Y4 + (rl + 1) * bq + cq + 1;
            const float4 acc = sobel3_adjoint_quad(x[-bq - 1], x[-bq], x[-bq + 1], x[bq - 1], x[bq], x[bq + 1], y[-bq - 1], y[-bq], y[-bq + 1],
                                                   y[-1], y[0], y[1], y[bq - 1], y[bq], y[bq + 1]);
            const float4 g4 = make_float4(scale * acc.x, scale * acc.y, scale * acc.z, scale * acc.w);
            reinterpret_cast<float4*>(s_g)[i] = g4;
            const int r = goy + rl, c = gox + 4 * cq;
            if (inner || (r >= 1 && r < H - 1 && c >= 1 && c + 3 < W - 1)) {
              const float m4 = fmaxf(fmaxf(fabsf(g4.x), fabsf(g4.y)), fmaxf(fabsf(g4.z), fabsf(g4.w)));
              gmax_t = fmaxf(gmax_t, (g4.x + g4.y + g4.z + g4.w) == (g4.x + g4.y + g4.z + g4.w) ? m4 : INFINITY);
              gsum_t += (fabsf(g4.x) + fabsf(g4.y)) + (fabsf(g4.z) + fabsf(g4.w));
            } else if (r >= 1 && r < H - 1) {
              const float e4[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (c + k >= 1 && c + k < W - 1) {
                  gmax_t = fmaxf(gmax_t, e4[k] == e4[k] ? fabsf(e4[k]) : INFINITY);
                  gsum_t += fabsf(e4[k]);
                }
            }
          }
          if (!inner) {  // (uniform) the image's outermost ring: folded taps; outside the image: 0
            auto gxy = [&](int qr, int qc, float& vx, float& vy) {  // (any position: the address is clamped into the pairs' window)
              const int o = min(max(qr - boy, 0), bh - 1) * bw + min(max(qc - box, 0), bw - 1);
              vx = s_gx[o], vy = s_gy[o];
            };
            __syncthreads();
            for_border_split(goy, wb.LH(), gox, gw, 1, 1, 1, [&](int r, int c) { s_g[(r - goy) * gw + (c - gox)] = 0.0f; },
                             [&](int r, int c) {   // the image's outermost ring
                               const float gv = scale * sobel3_adjoint_ring_dense(gxy, r, c, H, W, lo_px, H - lo_px, lo_px, W - lo_px);
                               s_g[(r - goy) * gw + (c - gox)] = gv;
                               gmax_t = fmaxf(gmax_t, gv == gv ? fabsf(gv) : INFINITY);
                               gsum_t += fabsf(gv);
                             });
          }
        }
        __syncthreads();  // (the pairs are dead: their region becomes the d_flow accumulators)
        for (int i = threadIdx.x; i < TH * TW; i += kBlock) reinterpret_cast<double2*>(s_d)[i] = make_double2(0.0, 0.0);
      }
      EBOS_RSTAMP(6);
      sq = wave_sum(sq);
      gmax_t = wave_max_nonneg(gmax_t);
      gsum_t = wave_sum(gsum_t);
      if (lane == 0) s_red[wave] = sq, s_gmax[wave] = gmax_t, s_gmax[2 * kWaves + wave] = gsum_t;
      if (publish) drain_stores();
    }
    __syncthreads();
    EBOS_RSTAMP(7);
    if (threadIdx.x == 0) {  // the record of the loss value: (sum of squares, regulariser partial of the previous iteration)
      KArgs& a = fresh_args();
      double Q = 0.0;
      for (int k = 0; k < kWaves; ++k) Q += s_red[k];
      unsigned long long* rec = a.rec2 + ((size_t)(it & 1) * (a.tiles_y * a.tiles_x) + tile) * kRecGranules;
      put_granules(rec, ep, Q);
      put_granules(rec + 2, ep, s_reg[1]);
    }
    // ---- tiles two apart reach into each other's halos: every tile has published its pixels (above); the halo is read back from the
    // neighbours' and mapped in a pass of its own (the four-launch pipeline's staging)
    if (!halo_complete) {
      KArgs& a = fresh_args();
      // (IMAGE coordinates, as in G)
      const int H = a.H + 2 * a.pad_h, W = a.W + 2 * a.pad_w, tiles_x = a.tiles_x, tiles_y = a.tiles_y, ty = tile / tiles_x, tx = tile - ty * tiles_x,
                tr0 = ty * TH + a.pad_h, tc0 = tx * TW + a.pad_w;
      const int lo_px = a.omit ? 1 : 0;
      const double n_px = (double)max(H - 2 * lo_px, 0) * (double)max(W - 2 * lo_px, 0);
      const double ga = 2.0 * (-(double)a.w_contrast) / (n_px - 1.0);
      const float Ga = (float)ga, Gc = (float)(-ga * mean);
      const int qw = wb.LW() / 4, n_q = wb.LH() * qw, oy = tr0 - wb.HR(), ox = tc0 - wb.HC();
      const float inv_qw = 1.0f / (float)qw;
      const float* iwe = a.iwe;
      if (threadIdx.x == 0) st_sc1(a.flagi + tile, (unsigned long long)ep);
      if (wave == 0) {
        const int nty = ty + lane / 3 - 1, ntx = tx + lane % 3 - 1;
        const bool nb = lane < 9 && nty >= 0 && nty < tiles_y && ntx >= 0 && ntx < tiles_x;
        const unsigned long long* f = a.flagi + (nb ? nty * tiles_x + ntx : tile);
        const bool ok = wave_wait([&]() { return !nb || ld_sc1(f) >= (unsigned long long)ep; }, a.status, a.cap_ticks);
        if (lane == 0 && !ok) s_ok = 0;
      }
      __syncthreads();
      float gmax_t = 0.0f, gsum_t = 0.0f;
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
          if (i >= n_q) continue;
          const int rl = (int)(((float)i + 0.5f) * inv_qw), cq = i - rl * qw;
          const int R = oy + rl, C = ox + 4 * cq;
          const float e4[4] = {wq[kq].x, wq[kq].y, wq[kq].z, wq[kq].w};
          float o4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool valid = R >= lo_px && R < H - lo_px && C + k >= lo_px && C + k < W - lo_px;
            o4[k] = valid ? Ga * e4[k] + Gc : 0.0f;
            gmax_t = fmaxf(gmax_t, o4[k] == o4[k] ? fabsf(o4[k]) : INFINITY);
            gsum_t += fabsf(o4[k]);
          }
          reinterpret_cast<float4*>(s_g)[i] = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
      }
      gmax_t = wave_max_nonneg(gmax_t);
      gsum_t = wave_sum(gsum_t);
      if (lane == 0) s_gmax[wave] = gmax_t, s_gmax[2 * kWaves + wave] = gsum_t;
      __syncthreads();
    }
    if (!s_ok) { done_ok = false; break; }
    // ---- B1: the sweep -------------------------------------------------------------------------------------------------------------
    FxUnit unit;
    bool fx;
    double tot_x = 0.0, tot_y = 0.0;  // UNI: this lane's sum of dt * d loss / d(x', y')
    {
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, pad_h = a.pad_h, pad_w = a.pad_w, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x;   // (the sweeps take SOURCE sizes + padding)
      const int lo_px = a.omit ? 1 : 0;
      const double n_px = (double)max(H + 2 * pad_h - 2 * lo_px, 0) * (double)max(W + 2 * pad_w - 2 * lo_px, 0);
      GradImage G;
      G.g = a.iwe;
      const double ga = 2.0 * (-(double)a.w_contrast) / (n_px - 1.0);
      G.a = (float)ga;
      G.c = (float)(-ga * mean);
      G.h = H + 2 * pad_h, G.w = W + 2 * pad_w, G.lo = lo_px;
      if constexpr (blur_on) G.set_blur(a.blur);
      else G.set_blur(Blur3{0.0f, 0.0f});
      EBOS_RSTAMP(9);
      unit = bwd_fx_unit(s_gmax, a.dt_bound, wb.LH() * wb.LW());
      const BwdShared bsh{&s_spill, &s_bad, &s_next};
      TileRange tr;
      tr.ty = ty, tr.tx = tx, tr.slab = tile, tr.part = 0;
      tr.g_first = rfl(P.g_first), tr.g_last = rfl(P.g_last), tr.beg = rfl(P.beg), tr.end = rfl(P.end);
      const EvPtrs ev = a.ev;
      BwdPre pre;
      fx = true;
      if constexpr (UNI) {
        // (no scatter: every lane sums dt * d loss / d(x', y') of its events -- the f64 sweep of the four-launch UNIFORM kernel)
        unit.fx = false, unit.scale = 1.0f, unit.limit = 0.0f;
        fx = bwd_lean_sweeps<TH, TW, HALO, true, false, true>(tr, s_d, s_g, ev, s_cells, H, W, pad_h, pad_w, G, tot_x, tot_y, ChunkQueue{&s_next}, wb,
                                                             unit, a.dt_bound, pre, false, bsh, NoHook{});
      } else if constexpr (FRAC) {
        unit.fx = false, unit.scale = 1.0f, unit.limit = 0.0f;  // (the general sweep: fractions per slot, f64 accumulators)
        fx = bwd_lean_sweeps<TH, TW, HALO, false, true, true>(tr, s_d, s_g, ev, s_flow_b, H, W, pad_h, pad_w, G, tot_x, tot_y, ChunkQueue{&s_next}, wb,
                                                             unit, a.dt_bound, pre, false, bsh, NoHook{});
      } else {
      decode_bgroup(pre.A, pre_raw.A, (unsigned)PW, 0u, (unsigned)(AP * PW + AP));  // (the tile's flow in LDS: element indices, pitch PW)
      decode_bgroup(pre.B, pre_raw.B, (unsigned)PW, 0u, (unsigned)(AP * PW + AP));
      finish_bgroup<TW>(pre.A);
      finish_bgroup<TW>(pre.B);
      if (!(EBOS_ABL & 32))
      fx = bwd_lean_sweeps<TH, TW, HALO, false, true, true>(tr, s_d, s_g, ev, s_flow_b, H, W, pad_h, pad_w, G, tot_x, tot_y, ChunkQueue{&s_next}, wb,
                                                           unit, a.dt_bound, pre, true, bsh, NoHook{});
      }
      __syncthreads();
    }
    EBOS_RSTAMP(10);
    // ---- B2: regularisers on the tile's flow, adjoint of grid -> dense on the tile -> partial cell gradients (write-through) ------
    if constexpr (UNI) {  // the tile's partial pair of d loss / d theta: one record {epoch, 2 x 2 granules}
      block_sum2(tot_x, tot_y, s_red);
      if (threadIdx.x == 0) {
        unsigned long long* rec = fresh_args().part3 + (size_t)tile * 4;
        put_granules(rec, ep, tot_x);
        put_granules(rec + 2, ep, tot_y);
      }
    } else {
      KArgs& a = fresh_args();
      const int H = a.H, W = a.W, tiles_x = a.tiles_x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
      TileRange tr;
      tr.ty = ty, tr.tx = tx, tr.slab = tile, tr.part = 0;
      tr.g_first = tr.g_last = tr.beg = tr.end = 0;
      const TileGrad<TH, TW> grad{fx, 1.0f / unit.scale, s_d};
      const float s_norm = a.s_norm, s_tv = a.s_tv;
      const bool any_reg = s_norm != 0.0f || s_tv != 0.0f;
      if (!(EBOS_ABL & 4096))
      // (the partial cell gradients leave as tagged granules {epoch, value}: no flag, no drain -- S3 polls the values themselves)
      grid_tile_epilogue<TH, TW, HALO, true>(tr, ty * TH, tx * TW, H, W, s_d, s_g, s_flow_b, s_lerp, grad, nullptr, s_norm, s_tv,
                                             any_reg ? &s_reg[0] : nullptr,
                                             reinterpret_cast<float*>(a.part3 + (size_t)tile * (2 * kGridCells * kGridCells)), ep);
      EBOS_RSTAMP(11);
      __syncthreads();  // (the epilogue's LDS buffers are dead: the image may be cleared over them)
      if (threadIdx.x == 0) s_reg[1] = any_reg ? s_reg[0] : 0.0;
    }
    EBOS_RSTAMP(12);
    // ---- S3 + A: every thread that steps a cell element polls the <= kSpan x kSpan partial values its cell sums -- tagged granules,
    // the data is its own flag -- and applies Adam; meanwhile the other waves clear the LDS image for the next pass (and workgroup 0
    // keeps the books of the loss) -------------------------------------------------------------------------------------------------
    if constexpr (UNI) {
      // every workgroup sums ALL tiles' partial pairs -- thread t < 256 the tiles t, t + 256, ... and then the four waves' sums, the
      // order of theta_adam_kernel (iwe_tile_core.h) -- and steps theta itself; the other waves clear the LDS image meanwhile
      KArgs& a = fresh_args();
      const int n_tiles = a.tiles_y * a.tiles_x;
      constexpr int kSumThreads = 256, kSumWaves = kSumThreads / kWave, kPer = kBlock / kSumThreads;
      if ((int)threadIdx.x < kSumThreads) {
        unsigned long long g[kPer][4];
        const bool ok = wave_wait([&]() {
          bool all = true;
#pragma unroll
          for (int k = 0; k < kPer; ++k) {
            const int t = (int)threadIdx.x + k * kSumThreads;
            const bool use = t < n_tiles;
            const unsigned long long* rec = a.part3 + (size_t)min(t, n_tiles - 1) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              g[k][j] = (unsigned long long)ep << 32;
              if (__builtin_amdgcn_ballot_w64(use) != 0ull) g[k][j] = ld_sc1(rec + j);
              all = all && (!use || (unsigned)(g[k][j] >> 32) == ep);
            }
          }
          return all;
        }, a.status, a.cap_ticks);
        if (lane == 0 && !ok) s_ok = 0;
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int k = 0; k < kPer; ++k)
          if ((int)threadIdx.x + k * kSumThreads < n_tiles) {
            sx += __builtin_bit_cast(double, (g[k][0] & 0xffffffffull) | (g[k][1] << 32));
            sy += __builtin_bit_cast(double, (g[k][2] & 0xffffffffull) | (g[k][3] << 32));
          }
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        if (lane == 0) s_red[wave] = sx, s_red[kSumWaves + wave] = sy;
      } else {
        if (blockIdx.x == 0 && wave == kWaves - 1 && it >= 1) book_loss(a, it - 1, lane, s_hist, s_adam);
        for (int i = threadIdx.x - kSumThreads; i < kCells / 2; i += kBlock - kSumThreads) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
      }
      __syncthreads();
      if (wave == 0) {
        const double gx = wave_sum(lane < kSumWaves ? s_red[lane] : 0.0), gy = wave_sum(lane < kSumWaves ? s_red[kSumWaves + lane] : 0.0);
        if (lane < 2) {
          const float g = lane == 0 ? (float)gx : (float)gy;
          float m_e = s_m[lane], v_e = s_v[lane], th = s_cells[lane];
          adam_update(g, m_e, v_e, th, s_adam[0], s_adam[1], (float)a.beta2, (float)(1.0 - a.beta1), (float)(1.0 - a.beta2), (float)a.eps);
          s_cells[lane] = th, s_m[lane] = m_e, s_v[lane] = v_e, s_gl[lane] = g;
        }
      }
    } else {
      KArgs& a = fresh_args();
      const int n_el = 2 * rfl(P.ni) * rfl(P.nj);
      if (wave * kWave < n_el) {
        const int tiles_x = a.tiles_x, tiles_y = a.tiles_y;
        const int ninj = rfl(P.ni) * rfl(P.nj), nj = rfl(P.nj);
        const bool has = (int)threadIdx.x < n_el;
        const int e_ = has ? (int)threadIdx.x : 0;
        float m_e = s_m[e_ % kResElems], v_e = s_v[e_ % kResElems];
        const int ch = e_ / ninj, rem = e_ - ch * ninj, ci = rem / nj, cj = rem - ci * nj;
        unsigned cand_a, cand_b;
        if (wave * kWave < kStoredCands) {  // (wave-uniform)
          cand_a = s_cand_a[e_ % kStoredCands], cand_b = s_cand_b[e_ % kStoredCands];
        } else {
          const Axis ay = a.gs.ay, ax = a.gs.ax;
          cell_candidates<TH, TW>(ay, ax, rfl(P.gi0) + ci, rfl(P.gj0) + cj, a.H, a.W, tiles_y, tiles_x, tile / tiles_x,
                                  tile - (tile / tiles_x) * tiles_x, cand_a, cand_b);
        }
        const unsigned long long* cp = a.part3;
        const int cty0 = (int)(cand_a & 255u), ctx0 = (int)((cand_a >> 8) & 255u);
        float mask = 1.0f;
        if (a.theta_mask != nullptr) mask = a.theta_mask[(int64_t)(rfl(P.gi0) + ci) * a.gs.ax.g + rfl(P.gj0) + cj];
        float pv[kSpan][kSpan];
        const bool wide = has && ((cand_a >> 25) & 1u);
        float wide_sum = 0.0f;
        bool ok = wave_wait([&]() {
          bool all = true;
#pragma unroll
          for (int p = 0; p < kSpan; ++p)
#pragma unroll
            for (int q = 0; q < kSpan; ++q) {
              const int cty = min(cty0 + p, tiles_y - 1), ctx = min(ctx0 + q, tiles_x - 1);
              const int li = (int)((cand_b >> (4 * p)) & 15u), lj = (int)((cand_b >> (16 + 4 * q)) & 15u);
              const bool use = has && !wide && ((cand_a >> (16 + p)) & 1u) && ((cand_a >> (20 + q)) & 1u);
              unsigned long long g = (unsigned long long)ep << 32;
              if (!(EBOS_ABL & 1024) && __builtin_amdgcn_ballot_w64(use) != 0ull)
                g = ld_sc1(cp + (((int64_t)(cty * tiles_x + ctx) * 2 + ch) * kGridCells + li) * kGridCells + lj);
              pv[p][q] = __uint_as_float((unsigned)g);
              all = all && (!use || (unsigned)(g >> 32) == ep || (EBOS_ABL & 512) != 0);
            }
          return all;
        }, a.status, a.cap_ticks);
        if (__builtin_amdgcn_ballot_w64(wide) != 0ull) {
          // wide cells (the edge cells of a coarse scale: their support takes in the replicate padding and spans more than kSpan
          // tiles): every tile of the support whose block holds the cell, one by one -- tile row, tile column: the order of
          // patch_grad_combine_kernel's plain loops
          ok = wave_wait([&]() {
            bool all = true;
            if (wide) {
              const Axis ay = a.gs.ay, ax = a.gs.ax;
              const int gi = rfl(P.gi0) + ci, gj = rfl(P.gj0) + cj;
              int r_lo, r_hi, c_lo, c_hi;
              support(ay, gi, a.H, &r_lo, &r_hi);
              support(ax, gj, a.W, &c_lo, &c_hi);
              float acc = 0.0f;
              if (r_lo < r_hi && c_lo < c_hi)
                for (int cty = r_lo / TH; cty <= (r_hi - 1) / TH; ++cty) {
                  const int bi0 = lerp_at(ay, cty * TH).i0, bi1 = lerp_at(ay, min(cty * TH + TH, a.H) - 1).i1;
                  if (gi < bi0 || gi > bi1) continue;
                  for (int ctx = c_lo / TW; ctx <= (c_hi - 1) / TW; ++ctx) {
                    const int bj0 = lerp_at(ax, ctx * TW).i0, bj1 = lerp_at(ax, min(ctx * TW + TW, a.W) - 1).i1;
                    if (gj < bj0 || gj > bj1) continue;
                    const unsigned long long g = ld_sc1(cp + (((int64_t)(cty * tiles_x + ctx) * 2 + ch) * kGridCells + (gi - bi0)) * kGridCells + (gj - bj0));
                    all = all && (unsigned)(g >> 32) == ep;
                    acc += __uint_as_float((unsigned)g);
                  }
                }
              wide_sum = acc;
            }
            return all;
          }, a.status, a.cap_ticks) && ok;
        }
        if (lane == 0 && !ok) s_ok = 0;
        if (has) {
          float g = 0.0f;
#pragma unroll
          for (int p = 0; p < kSpan; ++p)
#pragma unroll
            for (int q = 0; q < kSpan; ++q) g += (((cand_a >> (16 + p)) & 1u) && ((cand_a >> (20 + q)) & 1u)) ? pv[p][q] : 0.0f;
          if (wide) g = wide_sum;
          if (a.theta_mask != nullptr) g *= mask;
          float th = s_cells[(ch * kGridCells + ci) * kGridCells + cj];
          adam_update(g, m_e, v_e, th, s_adam[0], s_adam[1], (float)a.beta2, (float)(1.0 - a.beta1), (float)(1.0 - a.beta2), (float)a.eps);
          s_cells[(ch * kGridCells + ci) * kGridCells + cj] = th;
          s_m[e_ % kResElems] = m_e, s_v[e_ % kResElems] = v_e;
          // (d_theta: the last gradient, by the first tile that holds the cell -- stored as it is formed instead of kept)
          if ((cand_a >> 24) & 1u) a.d_theta[((int64_t)ch * a.gs.ay.g + rfl(P.gi0) + ci) * a.gs.ax.g + rfl(P.gj0) + cj] = g;
        }
      } else {
        // (workgroup 0, one wave: the loss of iteration it - 2 and the variance of it - 1 -- every workgroup has passed S1 of THIS
        // iteration, so the records of the previous one are complete)
        if (blockIdx.x == 0 && wave == kWaves - 1 && it >= 1) book_loss(a, it - 1, lane, s_hist, s_adam);
        const int first = ((n_el + kWave - 1) / kWave) * kWave;   // (threads of the waves above clear: a multiple of the wave size)
        for (int i = threadIdx.x - first; i < kCells / 2; i += kBlock - first) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
      }
    }
    EBOS_RSTAMP(13);
    EBOS_RSTAMP(14);
    __syncthreads();  // the new theta block is in LDS
    if (!s_ok) { done_ok = false; break; }
  }
  // A launch that ended early writes nothing of the optimiser state: the host falls back from unchanged state -- except after a
  // SPILL in iteration `it` >= 1: every workgroup then stands in front of that iteration's all-to-all with `it` completed
  // iterations behind it (nobody passes an all-to-all whose record is missing, and every other wait's data arrives), and the
  // launch hands over after those: state, losses and step counter of `it` iterations, the status says -102 and how many.
  if (!done_ok) {
    const unsigned st = ld_sc1(fresh_args().status);
    if ((st & 255u) != RES_SPILL || (int)(st >> 8) != it || it < 1) return;
    n_iter = it;
  }

  // ---- the state goes back: every cell element by the first tile that holds it ------------------------------------------------------
  KArgs& a = fresh_args();
  if constexpr (UNI) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
      a.theta[threadIdx.x] = s_cells[threadIdx.x];
      a.exp_avg[threadIdx.x] = s_m[threadIdx.x];
      a.exp_avg_sq[threadIdx.x] = s_v[threadIdx.x];
      a.d_theta[threadIdx.x] = s_gl[threadIdx.x];
    }
  } else if ((int)threadIdx.x < 2 * rfl(P.ni) * rfl(P.nj)) {
    const int ninj = rfl(P.ni) * rfl(P.nj), nj = rfl(P.nj);
    const int ch = (int)threadIdx.x / ninj, rem = (int)threadIdx.x - ch * ninj, ci = rem / nj, cj = rem - ci * nj;
    const int64_t gidx = ((int64_t)ch * a.gs.ay.g + rfl(P.gi0) + ci) * a.gs.ax.g + rfl(P.gj0) + cj;
    unsigned cand_a, cand_b;
    if ((int)threadIdx.x < kStoredCands) {
      cand_a = s_cand_a[threadIdx.x], cand_b = s_cand_b[threadIdx.x];
    } else {
      const Axis ay = a.gs.ay, ax = a.gs.ax;
      cell_candidates<TH, TW>(ay, ax, rfl(P.gi0) + ci, rfl(P.gj0) + cj, a.H, a.W, a.tiles_y, a.tiles_x, tile / a.tiles_x,
                              tile - (tile / a.tiles_x) * a.tiles_x, cand_a, cand_b);
    }
    if ((cand_a >> 24) & 1u) {
    a.theta[gidx] = s_cells[(ch * kGridCells + ci) * kGridCells + cj];
    a.exp_avg[gidx] = s_m[threadIdx.x % kResElems];
    a.exp_avg_sq[gidx] = s_v[threadIdx.x % kResElems];
    }
  }
  if (n_iter <= 0) return;
  // the last iteration's loss: its regulariser partials travel through the `done` granules; workgroup 0 gathers them
  const int n_tiles = a.tiles_y * a.tiles_x;
  if (threadIdx.x == 0) put_granules(a.done + (size_t)tile * 2, (unsigned)n_iter, s_reg[1]);
  if (blockIdx.x != 0) return;
  double ar = 0.0;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  if (wave * kWave < n_tiles) {
    const int k = wave * kWave + lane;
    const unsigned long long* rec = a.done + (size_t)min(k, n_tiles - 1) * 2;
    unsigned long long g0 = 0, g1 = 0;
    const bool ok = wave_wait([&]() {
      g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1);
      return (unsigned)(g0 >> 32) == (unsigned)n_iter && (unsigned)(g1 >> 32) == (unsigned)n_iter;
    }, a.status, a.cap_ticks);
    if (ok && k < n_tiles) ar = __builtin_bit_cast(double, (g0 & 0xffffffffull) | (g1 << 32));
    if (!ok && lane == 0) s_bad = 1;
  }
  ar = block_sum(ar, s_red);
  if (s_bad) {
    // Some workgroup never published its `done` granule for these n_iter iterations: it left on another verdict (a wait past its
    // cap while the others handed over after a spill) and did NOT write its cells back -- the state in memory is a mixture.  Say
    // so: the completed-iterations word reads -1 and the host refuses to continue from it (ebos_cmax_resident_iterations).
    if (threadIdx.x == 0) st_sc1(a.status + 1, 0xffffffffu);
    return;
  }
  // (every workgroup has left the loop: the last iteration's records are complete)
  if (wave == 1) book_loss(a, n_iter - 1, lane, s_hist, s_adam);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t_last = a.t0 + n_iter - 1;
    if (a.losses != nullptr && t_last < a.losses_cap) a.losses[t_last] = (float)(-(double)a.w_contrast * (double)s_adam[2] + ar);
    a.step[0] = a.t0 + n_iter;
    st_sc1(a.status + 1, (unsigned)n_iter);   // the iterations this launch completed (all of them, or those before a spill)
    const int lo_px = a.omit ? 1 : 0;
    a.variance[0] = s_adam[2];
    a.moments[0] = s_hist[((n_iter - 1) & 1) * 2 + 1];
    a.moments[1] = (double)max(a.H + 2 * a.pad_h - 2 * lo_px, 0) * (double)max(a.W + 2 * a.pad_w - 2 * lo_px, 0);
  }
#endif
}

struct MailboxLayout {
  size_t off_status, off_rec1, off_part3, off_flagi, off_rec2, off_done, total;
};
inline MailboxLayout mailbox_layout(int n_tiles) {
  MailboxLayout m;
  m.off_status = 0;
  m.off_rec1 = 256;
  m.off_part3 = m.off_rec1 + (((size_t)2 * n_tiles * kRec1Granules * 8 + 255) & ~(size_t)255);
  m.off_flagi = m.off_part3 + (size_t)n_tiles * (2 * kGridCells * kGridCells) * 8;   // a tile's partial cell gradients as granules
  m.off_rec2 = m.off_flagi + (((size_t)n_tiles * 8 + 255) & ~(size_t)255);
  m.off_done = m.off_rec2 + (size_t)2 * n_tiles * kRecGranules * 8;
  m.total = m.off_done + (((size_t)n_tiles * 16 + 255) & ~(size_t)255);
  return m;
}

}  // namespace

// Resident launches of different streams may only run side by side while ALL their workgroups fit the device at once (two
// half-resident grids would wait for each other until their caps): per device, the launches in flight and their grid sizes are kept;
// a new launch first waits for the oldest ones until it fits beside the rest (cmax_resident.hip; one table for all tile shapes).
int order_resident_launches(hipStream_t s, int workgroups, int n_cu, bool after_launch);

namespace {

template <int TH, int TW, int HALO, bool UNI, bool FRAC = false, int CONTRAST = RC_VARIANCE>
int launch_resident(const ResidentArgs& a, void* mailbox, size_t mailbox_total, hipStream_t s) {
  if constexpr (resident_fits<TH, TW, HALO>()) {
    auto k = cmax_resident_kernel<TH, TW, HALO, UNI, FRAC, CONTRAST>;
    constexpr size_t lds = resident_lds_bytes<TH, TW, HALO>();
    if (int rc = reserve_lds(k, lds, "ebos_cmax_solve_resident")) return rc;
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kBlock, lds) != hipSuccess) {
      set_error("resident solve: cannot query the device's occupancy");
      return EBOS_ERR_LAUNCH;
    }
    const int n_tiles = a.tiles_y * a.tiles_x;
    if (per_cu < 1 || n_tiles > n_cu * per_cu) {
      set_error("resident solve: %d workgroups cannot be co-resident (%d CUs x %d)", n_tiles, n_cu, per_cu);
      return EBOS_ERR_UNSUPPORTED;
    }
    if (int rc = order_resident_launches(s, n_tiles, n_cu * per_cu, false)) return rc;
    // (cells no tile's block holds keep a zero gradient, as the four-launch pipeline reports them)
    const size_t n_theta = UNI ? (size_t)2 : (size_t)2 * a.gs.ay.g * a.gs.ax.g;
    if (hipMemsetAsync(a.d_theta, 0, n_theta * sizeof(float), s) != hipSuccess || hipMemsetAsync(mailbox, 0, mailbox_total, s) != hipSuccess) {
      set_error("resident solve: cannot clear the mailbox");
      return EBOS_ERR_LAUNCH;
    }
    k<<<dim3((unsigned)n_tiles), dim3(kBlock), lds, s>>>(a);
    return order_resident_launches(s, n_tiles, n_cu * per_cu, true);
  } else {
    set_error("resident solve: no resident kernel for tile %dx%d halo %d", TH, TW, HALO);
    return EBOS_ERR_UNSUPPORTED;
  }
}

// the kernel's arguments from the C ABI's problem structs (either the patch-flow problem or the 2-DoF one)
inline void resident_common_args(ResidentArgs& a, int H, int W, int pad_h, int pad_w, int tile_h, int tile_w, void* mailbox, int n_iter,
                                 int steps_done, double spin_timeout_s, const HaloArg& ha) {
  const int tiles_y = (H + tile_h - 1) / tile_h, tiles_x = (W + tile_w - 1) / tile_w;
  const MailboxLayout m = mailbox_layout(tiles_y * tiles_x);
  char* mb = reinterpret_cast<char*>(mailbox);
  a.H = H, a.W = W, a.tiles_y = tiles_y, a.tiles_x = tiles_x;
  a.pad_h = pad_h, a.pad_w = pad_w;
  a.status = reinterpret_cast<unsigned*>(mb + m.off_status);
  a.rec1 = reinterpret_cast<unsigned long long*>(mb + m.off_rec1);
  a.part3 = reinterpret_cast<unsigned long long*>(mb + m.off_part3);
  a.flagi = reinterpret_cast<unsigned long long*>(mb + m.off_flagi);
  a.rec2 = reinterpret_cast<unsigned long long*>(mb + m.off_rec2);
  a.done = reinterpret_cast<unsigned long long*>(mb + m.off_done);
  a.t0 = steps_done, a.n_iter = n_iter;
  // a built halo (no run-time windows asked for): an infinite |dt| bound makes every tile take the largest window
  a.dt_bound = ha.dyn ? ha.dt_bound : INFINITY;
  const double ticks = spin_timeout_s * 1.0e8;  // wall_clock64: 100 MHz
  a.cap_ticks = ticks < 1.0e3 ? 1000ull : (ticks > 9.0e18 ? 9000000000000000000ull : (unsigned long long)ticks);
  // one workgroup per tile: a window whose fullest tile holds more than this many times the average tile's events is the
  // pipeline's (adaptive work items); EBOS_RESIDENT_MAX_IMBALANCE overrides (0: never refuse)
  a.max_imbalance = -1.0f;   // (< 0: the kernel's own rule; > 0: the ratio rules with this ratio; 0: never refuse)
  if (const char* e = getenv("EBOS_RESIDENT_MAX_IMBALANCE")) a.max_imbalance = (float)atof(e);
}

inline ResidentArgs resident_args(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, double spin_timeout_s) {
  const HaloArg ha = decode_halo(q->halo);
  ResidentArgs a{};
  resident_common_args(a, q->H, q->W, q->pad_h, q->pad_w, q->tile_h, q->tile_w, mailbox, n_iter, q->steps_done, spin_timeout_s, ha);
  a.ev = EvPtrs{nullptr, nullptr, nullptr, nullptr, q->grp_offsets, q->cpix, q->cdt, nullptr, nullptr, nullptr, q->cfx, q->cfy};
  a.key_offsets = q->key_offsets;
  a.gs = GridSrc{make_axis(q->gh, q->patch_h, q->slide_h, q->H), make_axis(q->gw, q->patch_w, q->slide_w, q->W)};
  a.theta = q->theta, a.d_theta = q->d_theta, a.exp_avg = q->exp_avg, a.exp_avg_sq = q->exp_avg_sq;
  a.theta_mask = q->theta_mask;
  a.step = q->step;
  a.iwe = q->iwe;
  a.slabs = reinterpret_cast<float*>(q->workspace);
  a.cell_partials = q->grad_partials;
  a.losses = q->losses, a.losses_cap = q->losses_cap;
  a.lr = q->lr, a.beta1 = q->beta1, a.beta2 = q->beta2, a.eps = q->eps;
  a.gm = q->w_gradient_magnitude != 0.0f ? 1 : 0;
  a.w_contrast = a.gm ? q->w_gradient_magnitude : q->w_variance;
  a.s_norm = q->w_flow_norm / (float)((int64_t)q->H * q->W);
  a.s_tv = q->w_image_gradient / (float)(2 * (int64_t)q->H * q->W);
  a.omit = q->omit_boundary ? 1 : 0;
  a.variance = q->variance;
  a.moments = q->moments;
  a.blur = Blur3{q->blur_k0, q->blur_k1};
  return a;
}

inline ResidentArgs resident_args(const ebos_cmax_2dof_problem* q, int n_iter, void* mailbox, double spin_timeout_s, float w_variance) {
  const HaloArg ha = decode_halo(q->halo);
  ResidentArgs a{};
  resident_common_args(a, q->H, q->W, q->pad_h, q->pad_w, q->tile_h, q->tile_w, mailbox, n_iter, q->steps_done, spin_timeout_s, ha);
  a.ev = EvPtrs{nullptr, nullptr, nullptr, nullptr, q->grp_offsets, q->cpix, q->cdt, nullptr, nullptr, nullptr, q->cfx, q->cfy};
  a.key_offsets = q->key_offsets;
  a.gs = GridSrc{};
  a.theta = q->theta, a.d_theta = q->d_theta, a.exp_avg = q->exp_avg, a.exp_avg_sq = q->exp_avg_sq;
  a.theta_mask = nullptr;
  a.step = q->step;
  a.iwe = q->iwe;
  a.slabs = reinterpret_cast<float*>(q->workspace);
  a.cell_partials = nullptr;
  a.losses = q->losses, a.losses_cap = q->losses_cap;
  a.lr = q->lr, a.beta1 = q->beta1, a.beta2 = q->beta2, a.eps = q->eps;
  a.w_contrast = w_variance;
  a.s_norm = 0.0f, a.s_tv = 0.0f;
  a.omit = q->omit_boundary ? 1 : 0;
  a.variance = q->variance;
  a.moments = q->moments;
  a.blur = Blur3{q->blur_k0, q->blur_k1};
  return a;
}

// Two translation units per tile shape define its launchers with these bodies: cmax_resident_<tile>.hip the patch-grid kernels (one per
// contrast), cmax_resident_<tile>_2dof.hip the 2-DoF ones (plain / blurred variance x integer / fractional source coordinates)
template <int TH, int TW, int HALO>
int resident_patch_launch(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, double spin_timeout_s, hipStream_t s) {
  const MailboxLayout m = mailbox_layout(((q->H + TH - 1) / TH) * ((q->W + TW - 1) / TW));
  const ResidentArgs a = resident_args(q, n_iter, mailbox, spin_timeout_s);
  if (q->cfx != nullptr) {  // fractional source coordinates
    if (a.gm) return launch_resident<TH, TW, HALO, false, true, RC_GRADIENT_MAGNITUDE>(a, mailbox, m.total, s);
    if (a.blur.k0 != 0.0f) return launch_resident<TH, TW, HALO, false, true, RC_BLURRED_VARIANCE>(a, mailbox, m.total, s);
    return launch_resident<TH, TW, HALO, false, true, RC_VARIANCE>(a, mailbox, m.total, s);
  }
  if (a.gm) return launch_resident<TH, TW, HALO, false, false, RC_GRADIENT_MAGNITUDE>(a, mailbox, m.total, s);
  if (a.blur.k0 != 0.0f) return launch_resident<TH, TW, HALO, false, false, RC_BLURRED_VARIANCE>(a, mailbox, m.total, s);
  return launch_resident<TH, TW, HALO, false, false, RC_VARIANCE>(a, mailbox, m.total, s);
}
template <int TH, int TW, int HALO>
int resident_2dof_launch(const ebos_cmax_2dof_problem* q, float w_variance, int n_iter, void* mailbox, double spin_timeout_s, hipStream_t s) {
  const MailboxLayout m = mailbox_layout(((q->H + TH - 1) / TH) * ((q->W + TW - 1) / TW));
  const ResidentArgs a = resident_args(q, n_iter, mailbox, spin_timeout_s, w_variance);
  const bool blur = a.blur.k0 != 0.0f;
  if (q->cfx != nullptr)  // fractional source coordinates
    return blur ? launch_resident<TH, TW, HALO, true, true, RC_BLURRED_VARIANCE>(a, mailbox, m.total, s)
                : launch_resident<TH, TW, HALO, true, true, RC_VARIANCE>(a, mailbox, m.total, s);
  return blur ? launch_resident<TH, TW, HALO, true, false, RC_BLURRED_VARIANCE>(a, mailbox, m.total, s)
              : launch_resident<TH, TW, HALO, true, false, RC_VARIANCE>(a, mailbox, m.total, s);
}

}  // namespace

// the per-tile-shape launchers (cmax_resident_<TH>x<TW>.hip, cmax_resident_<TH>x<TW>_2dof.hip)
#define EBOS_RESIDENT_LAUNCHERS(TILE)                                                                                                     \
  int resident_launch_##TILE(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, double spin_timeout_s, hipStream_t s);         \
  int resident_launch_2dof_##TILE(const ebos_cmax_2dof_problem* q, float w_variance, int n_iter, void* mailbox, double spin_timeout_s,   \
                                  hipStream_t s);
EBOS_RESIDENT_LAUNCHERS(45x80)
EBOS_RESIDENT_LAUNCHERS(32x32)
EBOS_RESIDENT_LAUNCHERS(32x64)
#undef EBOS_RESIDENT_LAUNCHERS

}  // namespace ebos
