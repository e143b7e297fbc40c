// plan_lean.hip -- the compact event plan straight from the window, nothing else ("lean" plan build).
//
// The tile-private kernels of the hot loop read only  cpix (u16 tile-local pixel) / cdt (f32 dt) / grp_offsets /
// key_offsets  (iwe_tiled.hip).  The general build (event_plan.hip: SoA conversion -> histogram atomics -> 32-byte record
// scatter -> unpack -> compact fill) also leaves SoA x / y / dt / p and the permutation behind, for per-event weights and
// fractional source coordinates, and pays ~6 GB of traffic and 10 M global atomics for 10 M events (1.3 ms).  When
// neither weights nor weight gradients are wanted, this file builds the same compact arrays with a two-level counting
// sort that never scatters at random into HBM:
//
//   K0 stage      ONE read of the window.  One workgroup per chunk of 8192 events: bin = (tile, row band of the tile); the chunk is
//                 counting-sorted by bin INSIDE LDS (the rank of an event in its (chunk, bin) run is what its LDS histogram atomic
//                 returns) and leaves as a coalesced stream: stage_px (u16 pixel-in-tile) + stage_t (the timestamp as it came: f32 /
//                 f64 / ticks -- 6 or 10 B/event), plus the chunk's row of the table [chunk][bin] = (offset in the chunk, count) and
//                 its partial (min t, max t, out-of-image, fractional).  No atomics to memory, no random writes.
//   K1 totals     column sums of the table -> events per bin; reduction of the chunks' time partials
//   K2 scans      over the bins (bin_base); over the tiles (grp_offsets, padded to groups of 4)
//   K3 bin sort   one workgroup per bin: GATHERS the bin's run out of every chunk (eight lanes per run: 12 - 64 contiguous bytes each,
//                 served by the L2 / Infinity Cache -- the staging stream of a 10 M-event window is 60 - 100 MB), dt in fp64
//                 (src/warp.py:264-288) now that the window's min / max are known, LDS histogram over the band's pixels ->
//                 key_offsets; counting sort staged in LDS -> cpix / cdt written coalesced at their final positions (+ NaN padding)
//
// Round 6: the window used to be read twice (count, then partition into per-bin segments with 8-byte scattered writes: 81 + 173 us
// of a 0.43 ms build of 10 M f64 events, 0.11 of the HBM roofline).  The timestamp travels unconverted because dt needs the window's
// min / max, which only the end of the one pass knows.
//
// Same layout contract as ebos_bin_events_f32 + ebos_plan_compact_f32 (include/ebos_hip.h): key_offsets, grp_offsets,
// cpix, cdt; the order of the events INSIDE one source pixel is unspecified there and here (integer accumulation makes
// the images independent of it).  Reference semantics: src/warp.py:230-288 (reference time, dt), :334 (source pixel =
// trunc), src/data_loader/ccs.py:289-297 (raw columns -> (row, col, t / 1e6, p)).
#include <algorithm>
#include <type_traits>

#include "common.h"

namespace ebos {
namespace {

// In-kernel phase stamps, diagnostic builds only (EBOS_EXTRA_FLAGS=-DEBOS_LEAN_STAMPS; tools/stamp_plan_lean.py)
#ifdef EBOS_LEAN_STAMPS
__device__ unsigned long long g_lean_stamps[2][8192 * 8];   // [stage | bin sort][workgroup][phase]
#define EBOS_LSTAMP(which, k)                                                                                     \
  do {                                                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_lean_stamps[which][blockIdx.x * 8 + (k)] = wall_clock64();       \
  } while (0)
#else
#define EBOS_LSTAMP(which, k) \
  do {                        \
  } while (0)
#endif

constexpr int kChunk = 8192;       // most events per workgroup of the staging pass (lean_layout picks the length)
constexpr int kLeanBlock = 1024;
constexpr int kMaxBins = 8192;     // LDS histogram of K0 / K2
constexpr int kSortBlock = 1024;   // threads of a bin-sort workgroup
// dynamic LDS of the bin sort: two int32 per pixel of the band + 12 B per staged event (arrival order, then sorted by pixel): one
// workgroup per CU.  (Measured: bins of half the size sorted by 512-thread workgroups, two per CU, are no faster -- 160 against 132 us
// for 10 M events: a bin's gather costs per RUN, not per event, and twice the bins are twice the runs.)
constexpr size_t kSortLds = 158 * 1024;

enum LeanSource { SRC_AOS_F32 = 0, SRC_AOS_F64 = 1, SRC_RAW32 = 2, SRC_RAW64 = 3 };

struct LeanIn {
  const void* events;   // AoS [n, 4] (x = row, y = col, t, p)
  const int16_t* col;   // raw columns
  const int16_t* row;
  const void* t;
  double ticks_per_second;
};

struct Ev {
  float x, y;  // row, column
  double t;    // seconds (AoS) or ticks (raw)
};

template <int SRC>
__device__ __forceinline__ Ev read_event(const LeanIn& in, int64_t i) {
  Ev e;
  if (SRC == SRC_AOS_F32) {
    const float4 v = reinterpret_cast<const float4*>(in.events)[i];
    e.x = v.x, e.y = v.y, e.t = (double)v.z;
  } else if (SRC == SRC_AOS_F64) {
    const double2 a = reinterpret_cast<const double2*>(in.events)[2 * i];
    const double tt = reinterpret_cast<const double*>(in.events)[4 * i + 2];
    e.x = (float)a.x, e.y = (float)a.y, e.t = tt;
  } else {
    e.x = (float)in.row[i];  // events[:, 0] = y (row), ccs.py:293
    e.y = (float)in.col[i];  // events[:, 1] = x (column), :294
    e.t = SRC == SRC_RAW32 ? (double)static_cast<const int32_t*>(in.t)[i] : (double)static_cast<const int64_t*>(in.t)[i];
  }
  return e;
}

struct LeanGeom {
  int H, W, th, tw, tiles_x, n_tiles, sub, n_bins;
  int chunk;             // events per chunk of the staging pass (<= kChunk) = stride of a chunk in the staged streams
  unsigned m_th, m_tw;   // floor(2^32 / d) + 1: floor(v / d) == umulhi(v, m) for v < 2^16, d <= 2^16 (0: plain division, larger images)
};
// three integer divisions by run-time values per event were most of the VALU work of the passes that bin events (~40 instructions
// each); v * d < 2^32 makes the multiply-high exact
__device__ __forceinline__ int div_small(int v, int d, unsigned m) { return m ? (int)__umulhi((unsigned)v, m) : v / d; }

// bin of an event (tile-major, then row band) and its pixel inside the tile; bin < 0: outside the image / not finite
__device__ __forceinline__ int event_bin(const LeanGeom& g, float x, float y, unsigned& pix, bool& fractional) {
  fractional = false;
  if (!(x > -1e9f && x < 1e9f && y > -1e9f && y < 1e9f)) return -1;
  const int r = (int)x, c = (int)y;  // truncation toward zero, src/warp.py:334
  if (r < 0 || r >= g.H || c < 0 || c >= g.W) return -1;
  fractional = (x != (float)r) || (y != (float)c) || x < 0.f || y < 0.f;
  const int ty = div_small(r, g.th, g.m_th), tx = div_small(c, g.tw, g.m_tw);
  const int rl = r - ty * g.th, cl = c - tx * g.tw;
  pix = ((unsigned)rl << 8) | (unsigned)cl;
  return (ty * g.tiles_x + tx) * g.sub + div_small(rl * g.sub, g.th, g.m_th);   // (rl sub < 256 * 256)
}
__device__ __forceinline__ int band_row0(const LeanGeom& g, int s) { return (s * g.th + g.sub - 1) / g.sub; }  // first row with (r sub) / th == s

// order-preserving map double -> uint64 (for atomicMin / atomicMax)
__device__ __forceinline__ unsigned long long ordered(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unordered(unsigned long long u) {
  const unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  return __longlong_as_double((long long)b);
}

// table entry of (chunk, bin): offset of the run inside the chunk's staged stream | count << 14   (both <= 8192)
constexpr int kTabShift = 14;
constexpr unsigned kTabMask = (1u << kTabShift) - 1u;

struct ChunkPartial {
  double lo, hi;   // min / max of t over ALL events of the chunk (like calculate_reftime, src/warp.py:245-253); +-inf when empty
  int bad, frac;
};

struct LeanScratch {
  unsigned* tab;                // [n_bins][n_chunks]  (a bin's entries contiguous: the bin sort reads them coalesced)
  int32_t* bin_base;            // [n_bins + 1]  totals (K1), then exclusive offsets (K2)
  unsigned long long* tm;       // [2] ordered min / max of t
  unsigned* ticket;             // [1] arrivals of lean_totals_kernel's workgroups (zeroed by K0)
  ChunkPartial* partial;        // [n_chunks]
  uint16_t* stage_px;           // [n_chunks * chunk]
  void* stage_t;                // [n_chunks * chunk] 4 bytes (f32 / int32 ticks) or 8 bytes (f64 / int64 ticks) each
};

template <int SRC> struct StageT { typedef unsigned type; };
template <> struct StageT<SRC_AOS_F64> { typedef unsigned long long type; };
template <> struct StageT<SRC_RAW64> { typedef unsigned long long type; };

// the timestamp as staged (bit-exact), and back to what read_event() hands out
template <int SRC>
__device__ __forceinline__ typename StageT<SRC>::type read_stamp(const LeanIn& in, int64_t i, float& x, float& y, double& t) {
  if (SRC == SRC_AOS_F32) {
    const float4 v = reinterpret_cast<const float4*>(in.events)[i];
    x = v.x, y = v.y, t = (double)v.z;
    return (typename StageT<SRC>::type)__float_as_uint(v.z);
  } else if (SRC == SRC_AOS_F64) {
    const double2 a = reinterpret_cast<const double2*>(in.events)[2 * i];
    const double tt = reinterpret_cast<const double*>(in.events)[4 * i + 2];
    x = (float)a.x, y = (float)a.y, t = tt;
    return (typename StageT<SRC>::type)__double_as_longlong(tt);
  } else if (SRC == SRC_RAW32) {
    x = (float)in.row[i], y = (float)in.col[i];  // events[:, 0] = y (row), events[:, 1] = x (column), ccs.py:293-294
    const int32_t k = static_cast<const int32_t*>(in.t)[i];
    t = (double)k;
    return (typename StageT<SRC>::type)(unsigned)k;
  } else {
    x = (float)in.row[i], y = (float)in.col[i];
    const int64_t k = static_cast<const int64_t*>(in.t)[i];
    t = (double)k;
    return (typename StageT<SRC>::type)k;
  }
}
__device__ __forceinline__ double stamp_seconds(int src, const void* stage_t, int64_t i, double ticks_per_second) {
  switch (src) {
    case SRC_AOS_F32: return (double)static_cast<const float*>(stage_t)[i];
    case SRC_AOS_F64: return static_cast<const double*>(stage_t)[i];
    case SRC_RAW32: return (double)static_cast<const int32_t*>(stage_t)[i] / ticks_per_second;
    default: return (double)static_cast<const int64_t*>(stage_t)[i] / ticks_per_second;
  }
}

// exclusive scan of one value per thread over the workgroup (kLeanBlock threads); total -> every thread
__device__ __forceinline__ int32_t block_exclusive_scan(int32_t v, int32_t* s_wave /* [kLeanBlock / kWave + 1] */, int32_t& total) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  int32_t inc = v;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int32_t o = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += o;
  }
  if (lane == kWave - 1) s_wave[wid] = inc;
  __syncthreads();
  int32_t wave_off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kLeanBlock / kWave; ++k) {
    const int32_t w = s_wave[k];
    wave_off += k < wid ? w : 0;
    tot += w;
  }
  total = tot;
  __syncthreads();
  return wave_off + inc - v;
}

// (eight waves per SIMD = TWO workgroups per CU: with 72 - 80 VGPRs a CU held one, and its load phase -- 60 % of a workgroup's time,
// tools/stamp_plan_lean.py -- overlapped with nothing: 1536 workgroups ran as six rounds of 256)
template <int SRC>
__global__ void __launch_bounds__(kLeanBlock, 8)
lean_stage_kernel(LeanIn in, int64_t n, LeanGeom g, LeanScratch sc) {
  typedef typename StageT<SRC>::type stamp_t;
  extern __shared__ int32_t s_mem[];
  int32_t* s_hist = s_mem;                                                          // [n_bins] counts, then exclusive offsets
  stamp_t* s_t = reinterpret_cast<stamp_t*>(s_mem + ((g.n_bins + 1) & ~1));         // [chunk]
  uint16_t* s_px = reinterpret_cast<uint16_t*>(s_t + g.chunk);                      // [chunk]
  __shared__ int32_t s_wave[kLeanBlock / kWave + 1];
  EBOS_LSTAMP(0, 0);
  for (int b = threadIdx.x; b < g.n_bins; b += kLeanBlock) s_hist[b] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *sc.ticket = 0u;   // (K1 counts its workgroups; it is a later launch)
  __syncthreads();
  constexpr int kPer = kChunk / kLeanBlock;
  const int64_t base = (int64_t)blockIdx.x * g.chunk;
  const int64_t chunk_end = min(n, base + g.chunk);
  double lo = 1.0e308 * 10.0, hi = -1.0e308 * 10.0;  // +-inf
  int bad = 0, frac = 0;
  int bin_of[kPer];      // the bin (< 2^13), later | rank in the (chunk, bin) run << 13 (< 2^14); -1: dropped
  unsigned pix[kPer];
  stamp_t stamp[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    // (four events' loads in flight per thread, not eight: with two workgroups per CU that is 260 KB per CU on its way, several times
    // what the latency asks for, and eight AoS events -- 48 registers for f64 -- spilled under the 64-register budget)
    if (k == kPer / 2) asm volatile("" ::: "memory");
    const int64_t i = base + k * kLeanBlock + threadIdx.x;
    bin_of[k] = -1;
    if (i < chunk_end) {
      float x, y;
      double t;
      stamp[k] = read_stamp<SRC>(in, i, x, y, t);
      lo = t < lo ? t : lo;
      hi = t > hi ? t : hi;
      bool fr;
      const int b = event_bin(g, x, y, pix[k], fr);
      bin_of[k] = b;
      bad += b < 0;
      frac += fr;
    }
  }
  EBOS_LSTAMP(0, 1);
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if (bin_of[k] >= 0) bin_of[k] |= atomicAdd(&s_hist[bin_of[k]], 1) << 13;   // the event's rank in its (chunk, bin) run
  __syncthreads();
  EBOS_LSTAMP(0, 2);
  // exclusive scan over the bins: every thread a stretch of consecutive bins
  const int per = (g.n_bins + kLeanBlock - 1) / kLeanBlock;   // <= kMaxBins / kLeanBlock = 8
  const int b0 = threadIdx.x * per;
  int32_t mine = 0;
  for (int j = 0; j < per; ++j) mine += b0 + j < g.n_bins ? s_hist[b0 + j] : 0;
  int32_t kept;
  int32_t run = block_exclusive_scan(mine, s_wave, kept);
  for (int j = 0; j < per; ++j)
    if (b0 + j < g.n_bins) {
      const int32_t c = s_hist[b0 + j];
      s_hist[b0 + j] = run;
      // (4-byte stores a table row apart: the bin sort reads a bin's 1536 entries as 96 lines instead of 1536, and it is the pass that
      // is bound by the number of its requests)
      sc.tab[(int64_t)(b0 + j) * gridDim.x + blockIdx.x] = (unsigned)run | ((unsigned)c << kTabShift);
      run += c;
    }
  __syncthreads();
  EBOS_LSTAMP(0, 3);
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if (bin_of[k] >= 0) {
      const int pos = s_hist[bin_of[k] & 8191] + (bin_of[k] >> 13);
      s_t[pos] = stamp[k];
      s_px[pos] = (uint16_t)pix[k];
    }
  __syncthreads();
  EBOS_LSTAMP(0, 4);
  // the sorted chunk leaves as two coalesced streams (pixels two per lane)
  stamp_t* out_t = static_cast<stamp_t*>(sc.stage_t) + base;
  for (int i = threadIdx.x; i < kept; i += kLeanBlock) out_t[i] = s_t[i];
  unsigned* out_px = reinterpret_cast<unsigned*>(sc.stage_px + base);
  const unsigned* s_px2 = reinterpret_cast<const unsigned*>(s_px);
  for (int i = threadIdx.x; i < (kept + 1) / 2; i += kLeanBlock) out_px[i] = s_px2[i];
  // the chunk's partial: min / max of t, counts
  __shared__ double s_lo[kLeanBlock / kWave], s_hi[kLeanBlock / kWave];
  __shared__ int s_bad[kLeanBlock / kWave], s_frac[kLeanBlock / kWave];
  lo = wave_min(lo);
  hi = wave_max(hi);
  bad = wave_sum(bad);
  frac = wave_sum(frac);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    s_lo[threadIdx.x / kWave] = lo;
    s_hi[threadIdx.x / kWave] = hi;
    s_bad[threadIdx.x / kWave] = bad;
    s_frac[threadIdx.x / kWave] = frac;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kLeanBlock / kWave; ++k) {
      lo = s_lo[k] < lo ? s_lo[k] : lo;
      hi = s_hi[k] > hi ? s_hi[k] : hi;
      bad += s_bad[k];
      frac += s_frac[k];
    }
    sc.partial[blockIdx.x] = ChunkPartial{lo, hi, bad, frac};
  }
  EBOS_LSTAMP(0, 5);
}

// one workgroup (1024 threads): bin totals -> bin_base (exclusive, + total); tile totals -> grp_offsets (groups of 4 slots,
// exclusive); key_offsets[n_keys] = events kept; tminmax (seconds) for the caller.  The totals are read with agent-scope loads: the
// caller is the LAST workgroup of lean_totals_kernel, and the other workgroups' atomics completed in other XCDs' memory paths.
__device__ __forceinline__ void lean_scan_bins_block(const LeanGeom& g, const LeanScratch& sc, int32_t* __restrict__ grp_offsets,
                                                     int32_t* __restrict__ key_offsets, int64_t n_keys, double ticks_per_second, int raw,
                                                     double* __restrict__ tminmax) {
  __shared__ int32_t s_wave[1024 / kWave];
  __shared__ int32_t s_carry;
  typedef __attribute__((address_space(1))) int32_t gi32_;
  auto total_of = [&](int b) { return __hip_atomic_load((gi32_*)(sc.bin_base + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  auto scan = [&](int count, auto value, auto store) {  // exclusive scan of value(i), store(i, prefix); returns nothing
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int start = 0; start < count; start += 1024) {
      const int i = start + threadIdx.x;
      const int32_t v = i < count ? value(i) : 0;
      int32_t inc = v;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int32_t o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
      }
      if (lane == kWave - 1) s_wave[wid] = inc;
      __syncthreads();
      int32_t wave_off = 0;
      for (int k = 0; k < wid; ++k) wave_off += s_wave[k];
      const int32_t carry = s_carry;
      if (i < count) store(i, carry + wave_off + inc - v);
      __syncthreads();
      if (threadIdx.x == 1023) s_carry = carry + wave_off + inc;
      __syncthreads();
    }
  };
  // tile totals first (they read the un-scanned bin totals)
  scan(g.n_tiles,
       [&](int t) {
         int32_t c = 0;
         for (int s = 0; s < g.sub; ++s) c += total_of(t * g.sub + s);
         return (c + 3) >> 2;
       },
       [&](int t, int32_t pre) { grp_offsets[t] = pre; });
  if (threadIdx.x == 0) grp_offsets[g.n_tiles] = s_carry;
  __syncthreads();
  scan(g.n_bins, [&](int b) { return total_of(b); }, [&](int b, int32_t pre) { sc.bin_base[b] = pre; });
  if (threadIdx.x == 0) {
    sc.bin_base[g.n_bins] = s_carry;
    key_offsets[n_keys] = s_carry;
    if (tminmax != nullptr) {
      const double lo = unordered(sc.tm[0]), hi = unordered(sc.tm[1]);
      tminmax[0] = raw ? lo / ticks_per_second : lo;  // t / 1e6, ccs.py:295
      tminmax[1] = raw ? hi / ticks_per_second : hi;
    }
  }
}

// K1: events per bin = sums of the table's rows, a wave per bin (sixteen bins per workgroup); workgroup 0 also folds the chunks' time
// partials and counts; the last workgroup to finish scans the totals
__global__ void __launch_bounds__(kLeanBlock)
lean_totals_kernel(LeanGeom g, int n_chunks, LeanScratch sc, int32_t* __restrict__ counts, int32_t* __restrict__ grp_offsets,
                   int32_t* __restrict__ key_offsets, int64_t n_keys, double ticks_per_second, int raw, double* __restrict__ tminmax) {
  __shared__ int s_is_last;
  {
    const int bin = blockIdx.x * (kLeanBlock / kWave) + (int)(threadIdx.x / kWave), lane = threadIdx.x & (kWave - 1);
    if (bin < g.n_bins) {
      const unsigned* __restrict__ row = sc.tab + (int64_t)bin * n_chunks;
      int32_t acc = 0;
      constexpr int kU = 4;   // (loads in flight per lane)
      for (int c = lane; c < n_chunks; c += kWave * kU) {
        unsigned v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = c + kWave * u < n_chunks ? row[c + kWave * u] : 0u;
#pragma unroll
        for (int u = 0; u < kU; ++u) acc += (int32_t)(v[u] >> kTabShift);
      }
      acc = wave_sum(acc);
      if (lane == 0) {
        typedef __attribute__((address_space(1))) int32_t gi32_;
        __hip_atomic_store((gi32_*)(sc.bin_base + bin), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through: the scanning workgroup may sit on another XCD)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (performed before this workgroup takes its number below)
      }
    }
  }
  if (blockIdx.x == 0) {
    double lo = 1.0e308 * 10.0, hi = -1.0e308 * 10.0;
    int bad = 0, frac = 0;
    for (int c = threadIdx.x; c < n_chunks; c += kLeanBlock) {
      const ChunkPartial p = sc.partial[c];
      lo = p.lo < lo ? p.lo : lo;
      hi = p.hi > hi ? p.hi : hi;
      bad += p.bad;
      frac += p.frac;
    }
    __shared__ double s_lo[kLeanBlock / kWave], s_hi[kLeanBlock / kWave];
    __shared__ int s_bad[kLeanBlock / kWave], s_frac[kLeanBlock / kWave];
    lo = wave_min(lo);
    hi = wave_max(hi);
    bad = wave_sum(bad);
    frac = wave_sum(frac);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      s_lo[threadIdx.x / kWave] = lo;
      s_hi[threadIdx.x / kWave] = hi;
      s_bad[threadIdx.x / kWave] = bad;
      s_frac[threadIdx.x / kWave] = frac;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < kLeanBlock / kWave; ++k) {
        lo = s_lo[k] < lo ? s_lo[k] : lo;
        hi = s_hi[k] > hi ? s_hi[k] : hi;
        bad += s_bad[k];
        frac += s_frac[k];
      }
      typedef __attribute__((address_space(1))) unsigned long long gu64_;
      // (write-through: the workgroup that scans reads them, and it may sit on another XCD)
      __hip_atomic_store((gu64_*)&sc.tm[0], lo <= hi ? ordered(lo) : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((gu64_*)&sc.tm[1], lo <= hi ? ordered(hi) : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      counts[0] = bad;
      counts[1] = frac;
    }
  }
  // the LAST workgroup to get here scans the totals (one launch and its gap less than a scan kernel of its own: every workgroup
  // drains its atomics, then takes a number)
  __syncthreads();
  if (threadIdx.x == 0) {
    typedef __attribute__((address_space(1))) unsigned gu32_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned n_wg = gridDim.x;
    s_is_last = __hip_atomic_fetch_add((gu32_*)sc.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_wg - 1u;
  }
  __syncthreads();
  if (!s_is_last) return;
  if (threadIdx.x == 0) {
    typedef __attribute__((address_space(1))) unsigned long long gu64_;
    // (tm through agent-scope loads in lean_scan_bins_block's reader below: re-publish into plain view of THIS workgroup)
    sc.tm[0] = __hip_atomic_load((gu64_*)&sc.tm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sc.tm[1] = __hip_atomic_load((gu64_*)&sc.tm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  lean_scan_bins_block(g, sc, grp_offsets, key_offsets, n_keys, ticks_per_second, raw, tminmax);
}

// what the bin sort needs to turn a staged timestamp into dt (fp64, exactly as events_to_soa_kernel / raw_to_soa_kernel, event_plan.hip)
struct LeanTime {
  int src, ref_mode, normalize_t;
  double ref_fraction, ticks_per_second;
};

// lane ^ STRIDE within a row of 16 lanes through the DPP path (VALU rate: a ds_bpermute per stage made the ten dependent stages of the
// 16-lane sorting network 1 100 cycles an iteration -- 8 us per workgroup -- on LDS-crossbar latency alone)
template <int STRIDE>
__device__ __forceinline__ int row_xor(int v) {
  static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8, "row_xor: 1, 2, 4 or 8");
  if (STRIDE == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1, 0, 3, 2]
  if (STRIDE == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2, 3, 0, 1]
  if (STRIDE == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
  const int m = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);            // row_half_mirror: lane ^ 7 ...
  return __builtin_amdgcn_update_dpp(0, m, 0x1B, 0xf, 0xf, false);                    // ... then quad_perm [3, 2, 1, 0]: lane ^ 3
}
template <int SIZE, int STRIDE>
__device__ __forceinline__ int bitonic16_step(int key, int sub16) {
  const int other = row_xor<STRIDE>(key);
  const bool up = (sub16 & SIZE) == 0;        // ascending block (SIZE 16: every lane)
  const bool low = (sub16 & STRIDE) == 0;     // this lane keeps the smaller of the pair when ascending
  const int mn = min(key, other), mx = max(key, other);
  return (low == up) ? mn : mx;
}
// the keys of lanes 0 .. n - 1 of every row of 16 lanes, ascending by lane, n = the smallest of 2 / 4 / 8 / 16 that holds `longest`
// (uniform over the wave: the longest run of its four rows); the lanes beyond hold a key that sorts behind every event.  A window of
// 2 M events has ~2 events per pixel: one or three of the ten stages.
__device__ __forceinline__ int bitonic16(int key, int sub16, int longest) {
  key = bitonic16_step<2, 1>(key, sub16);
  if (longest > 2) {
    key = bitonic16_step<4, 2>(key, sub16);
    key = bitonic16_step<4, 1>(key, sub16);
  }
  if (longest > 4) {
    key = bitonic16_step<8, 4>(key, sub16);
    key = bitonic16_step<8, 2>(key, sub16);
    key = bitonic16_step<8, 1>(key, sub16);
  }
  if (longest > 8) {
    key = bitonic16_step<16, 8>(key, sub16);
    key = bitonic16_step<16, 4>(key, sub16);
    key = bitonic16_step<16, 2>(key, sub16);
    key = bitonic16_step<16, 1>(key, sub16);
  }
  return key;
}

// a float's bits as an integer with the floats' order -- a TOTAL order: a NaN (a broken timestamp) still gets a rank of its own, where
// float comparisons would hand several events the same slot
__device__ __forceinline__ int sort_key(float d) {
  const int b = __float_as_int(d);
  return b ^ ((b >> 31) & 0x7fffffff);
}

// what a staged record holds: the pixel inside its tile and the timestamp's bits (4 or 8 of them)
struct LeanRec {
  unsigned pix;
  unsigned long long stamp;
};
// staged timestamp -> dt in fp64, exactly as events_to_soa_kernel / raw_to_soa_kernel (event_plan.hip): reference time and period from
// the window's min / max (K1 left them in sc.tm)
struct LeanClock {
  double ref, inv_period, tps;
  bool raw, wide;
  __device__ __forceinline__ LeanClock(const LeanScratch& sc, const LeanTime& tc) {
    raw = tc.src == SRC_RAW32 || tc.src == SRC_RAW64;
    wide = tc.src == SRC_AOS_F64 || tc.src == SRC_RAW64;   // (uniform) 8-byte stamps
    tps = tc.ticks_per_second;
    double tmin = unordered(sc.tm[0]), tmax = unordered(sc.tm[1]);
    if (raw) {
      tmin = tmin / tps;
      tmax = tmax / tps;
    } else if (tc.src == SRC_AOS_F32) {
      tmin = (double)(float)tmin;  // (exact: they are f32 values)
      tmax = (double)(float)tmax;
    }
    if (tc.ref_mode == EBOS_REF_FIRST) ref = tmin;
    else if (tc.ref_mode == EBOS_REF_LAST) ref = tmax;
    else ref = tmin + (tmax - tmin) * tc.ref_fraction;
    inv_period = tc.normalize_t ? 1.0 / (tmax - tmin) : 1.0;
  }
  __device__ __forceinline__ float dt(const LeanRec& r) const {   // (selects, no branches: sixteen of these stand unrolled in the gather)
    const double t8 = raw ? (double)(long long)r.stamp / tps : __longlong_as_double((long long)r.stamp);
    const double t4 = raw ? (double)(int32_t)(unsigned)r.stamp / tps : (double)__uint_as_float((unsigned)r.stamp);
    return (float)(((wide ? t8 : t4) - ref) * inv_period);
  }
};

// The events of one bin: one run per chunk of the staged streams, (offset, count) in the bin's row of the table.  A thread owns a
// chunk's entry; its wave walks the 64 runs eight at a time, EIGHT LANES PER RUN, sixteen elements of every run per pass -- all of them
// LOADED before the first one is used (sixteen loads in flight: one by one, each followed by the LDS atomic that consumes it, the
// gather was a chain of ~25 round trips per wave and the bin sort took 200 us for 10 M events).  One pass for the runs of a large
// uniform window (~6 events); a window of fewer, longer runs -- 2 M events are 512 chunks, ~15 events per run, a whole TILE's runs
// ~25 -- takes further passes.  load(index into the staged streams) -> payload; use(payload, arrival index): POS hands every element
// a slot of its own in [0, events of the bin) -- its wave reserves a stretch for its 64 runs (one LDS atomic per wave and BLOCK
// chunks), a run's elements follow each other inside it.
template <bool POS, int BLOCK, typename Load, typename Use>
__device__ __forceinline__ void lean_gather(const LeanScratch& sc, const LeanGeom& g, int bin, int n_chunks, int32_t* s_arrived, Load&& load,
                                            Use&& use) {
  typedef decltype(load((int64_t)0)) payload_t;
  const int lane = threadIdx.x & (kWave - 1);
  const unsigned* __restrict__ my_tab = sc.tab + (int64_t)bin * n_chunks;
  unsigned e_next = (int)threadIdx.x < n_chunks ? my_tab[threadIdx.x] : 0u;
  for (int blk = 0; blk < n_chunks; blk += BLOCK) {
    const unsigned e = e_next;
    const int c_next = blk + BLOCK + threadIdx.x;   // (the next BLOCK chunks' entries travel while these are walked)
    e_next = c_next < n_chunks ? my_tab[c_next] : 0u;
    const int my_cnt = (int)(e >> kTabShift), my_off = (int)(e & kTabMask);
    if (__ballot(my_cnt > 0) == 0ull) continue;   // (uniform per wave)
    int my_pos = 0;
    if (POS) {
      int inc = my_cnt;
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const int o = __shfl_up(inc, off, kWave);
        if (lane >= off) inc += o;
      }
      int wave_base = 0;
      if (lane == kWave - 1) wave_base = atomicAdd(s_arrived, inc);
      my_pos = __shfl(wave_base, kWave - 1, kWave) + inc - my_cnt;
    }
    const int64_t wave_chunk0 = blk + (threadIdx.x & ~(kWave - 1));
    const int i0 = lane & 7;
    for (int base = 0; __ballot(my_cnt > base) != 0ull; base += 16) {
      payload_t v[16];
      int cnt[8], pos[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = 8 * k + (lane >> 3);
        cnt[k] = __shfl(my_cnt, r, kWave) - base;
        pos[k] = (POS ? __shfl(my_pos, r, kWave) : 0) + base;
        // (an empty run -- also the lanes beyond the last chunk -- loads the first staged element: mapped, unused)
        const int off = __shfl(my_off, r, kWave);
        const int64_t src = cnt[k] > 0 ? (wave_chunk0 + r) * g.chunk + off + base : 0;
        v[2 * k] = load(src + min(i0, max(cnt[k] - 1, 0)));
        v[2 * k + 1] = load(src + min(i0 + 8, max(cnt[k] - 1, 0)));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (i0 < cnt[k]) use(v[2 * k], pos[k] + i0);
        if (i0 + 8 < cnt[k]) use(v[2 * k + 1], pos[k] + i0 + 8);
      }
    }
  }
}

// one workgroup per bin: counting sort of the bin's segment by pixel -> key_offsets of the band, cpix / cdt at their final slots
__global__ void __launch_bounds__(kSortBlock)
lean_bin_sort_kernel(LeanGeom g, LeanScratch sc, const int32_t* __restrict__ grp_offsets, int32_t* __restrict__ key_offsets,
                     uint16_t* __restrict__ cpix, float* __restrict__ cdt, int pix_cap, int sort_cap, int n_chunks, LeanTime tc) {
  // sort_cap: events of one bin staged in LDS, TWICE (6 B each: as they arrive from the gather, then sorted by pixel); a larger bin
  // (lean_layout leaves most bins of a window beyond ~10 k events per tile larger, on purpose) is gathered once for its histogram and
  // once per chunk of whole pixels, 2 x sort_cap events of the whole staging area at a time
  extern __shared__ int32_t s_raw[];
  int32_t* s_cnt = s_raw;                 // [pix_cap]  events per pixel of the band, then exclusive offsets
  int32_t* s_cur = s_raw + pix_cap;       // [pix_cap]  cursors
  float* a_dt = reinterpret_cast<float*>(s_raw + 2 * pix_cap);        // [sort_cap]  arrival order
  float* s_dt = a_dt + sort_cap;                                       // [sort_cap]  sorted by pixel (contiguous with a_dt)
  uint16_t* a_px = reinterpret_cast<uint16_t*>(s_dt + sort_cap);      // [sort_cap]
  uint16_t* s_px = a_px + sort_cap;                                    // [sort_cap]  (contiguous with a_px)
  __shared__ int32_t s_wave[kSortBlock / kWave];
  __shared__ int32_t s_arrived;
  // XCD-aware order: workgroups b and b + 8 share an XCD (its L2), and the runs of CONSECUTIVE bins lie next to each other in every
  // chunk of the staged streams -- a 128-byte line of staged pixels holds the runs of ~8 bins.  With bin = blockIdx the eight bins of
  // a line were gathered on eight different XCDs and every L2 fetched the line for 16 of its bytes (the pass took 180 us); so an XCD
  // takes a contiguous stretch of bins, its concurrent workgroups neighbouring ones.  (Placement is speed only, never correctness.)
  const int per_xcd = (g.n_bins + 7) / 8;
  const int bin = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (bin >= g.n_bins) return;
  const int tile = bin / g.sub, band = bin - tile * g.sub;
  const int r0 = band_row0(g, band), r1 = band_row0(g, band + 1);  // rows [r0, r1) of the tile
  const int n_pix = (r1 - r0) * g.tw;
  const int32_t seg0 = sc.bin_base[bin], seg1 = sc.bin_base[bin + 1], len = seg1 - seg0;
  const int64_t first_key = (int64_t)tile * g.th * g.tw + (int64_t)r0 * g.tw;
  const int32_t tile_first = sc.bin_base[tile * g.sub];            // events before this tile
  const int64_t out0 = (int64_t)grp_offsets[tile] * 4 + (seg0 - tile_first);  // final slot of the segment's first event
  EBOS_LSTAMP(1, 0);
  for (int i = threadIdx.x; i < n_pix; i += kSortBlock) s_cnt[i] = 0;
  if (threadIdx.x == 0) s_arrived = 0;
  __syncthreads();
  const bool staged = len <= sort_cap;
  const int lane = threadIdx.x & (kWave - 1);
  const LeanClock clock(sc, tc);
  typedef LeanRec Rec;
  auto dt_of = [&](const Rec& r) { return clock.dt(r); };
  const bool wide = clock.wide;
  auto load8 = [&](int64_t i) { return Rec{(unsigned)sc.stage_px[i], static_cast<const unsigned long long*>(sc.stage_t)[i]}; };
  auto load4 = [&](int64_t i) { return Rec{(unsigned)sc.stage_px[i], (unsigned long long)static_cast<const unsigned*>(sc.stage_t)[i]}; };
  auto for_each_staged = [&](auto with_pos, auto&& load, auto&& use) {
    lean_gather<decltype(with_pos)::value, kSortBlock>(sc, g, bin, n_chunks, &s_arrived, load, use);
  };
  if (staged) {
    // ONE gather: pixel and timestamp of every event of the bin, dt, into the arrival buffer; the pixel histogram beside it.  (As two
    // gathers -- pixels for the histogram, then pixels + timestamps for the placement -- the second one's table entries and loads were
    // another two dependent round trips per 1024 chunks: 14.5 of a workgroup's 38.6 us.)
    auto arrive = [&](const Rec& r, int at) {
      a_px[at] = (uint16_t)r.pix;
      a_dt[at] = dt_of(r);
      atomicAdd(&s_cnt[((int)(r.pix >> 8) - r0) * g.tw + (int)(r.pix & 255u)], 1);
    };
    if (wide) for_each_staged(std::true_type{}, load8, arrive);
    else for_each_staged(std::true_type{}, load4, arrive);
  } else {
    for_each_staged(std::false_type{}, [&](int64_t i) { return (unsigned)sc.stage_px[i]; },
                    [&](unsigned pix, int) { atomicAdd(&s_cnt[((int)(pix >> 8) - r0) * g.tw + (int)(pix & 255u)], 1); });
  }
  __syncthreads();
  EBOS_LSTAMP(1, 1);
  // exclusive scan of s_cnt [n_pix]: every thread a stretch of consecutive pixels, ONE scan over the workgroup (a band of a sparse
  // window is a whole tile, 3 600 pixels: as four rounds of a 1024-wide scan, three barriers each, the step took 4.3 us)
  const int wid = threadIdx.x / kWave;
  {
    const int per = (n_pix + kSortBlock - 1) / kSortBlock, p0 = threadIdx.x * per;
    int32_t mine = 0;
    for (int j = 0; j < per; ++j) mine += p0 + j < n_pix ? s_cnt[p0 + j] : 0;
    int32_t inc = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int32_t o = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += o;
    }
    if (lane == kWave - 1) s_wave[wid] = inc;
    __syncthreads();
    int32_t run = inc - mine;
    for (int k = 0; k < wid; ++k) run += s_wave[k];
    for (int j = 0; j < per; ++j)
      if (p0 + j < n_pix) {
        const int32_t c = s_cnt[p0 + j];
        s_cnt[p0 + j] = run;
        s_cur[p0 + j] = run;
        key_offsets[first_key + p0 + j] = seg0 + run;
        run += c;
      }
    __syncthreads();
  }
  EBOS_LSTAMP(1, 2);
  if (staged) {   // arrival order -> sorted by pixel, LDS to LDS
    for (int i = threadIdx.x; i < len; i += kSortBlock) {
      const unsigned pix = (unsigned)a_px[i];
      const int32_t pos = atomicAdd(&s_cur[((int)(pix >> 8) - r0) * g.tw + (int)(pix & 255u)], 1);
      s_px[pos] = (uint16_t)pix;
      s_dt[pos] = a_dt[i];
    }
  }
  // (an overfull bin -- a window far from uniform, or one of more than ~20 M events, whose bins are not cut finer than a (chunk, bin)
  // run of ~4 events allows -- is gathered AGAIN per chunk of whole pixels below, only that chunk's pixels kept: the runs come out of
  // the L2 / Infinity Cache, and nothing is scattered to memory and read back)
  __syncthreads();
  EBOS_LSTAMP(1, 3);
  // The cursors hand out a pixel's slots in the order the atomics arrive: two builds of one window differ in it, and a kernel that
  // sums a group's events in slot order before it accumulates exactly (the 2-DoF backward sweep) sees that in the last bit of its
  // gradient -- which an optimiser amplifies: two solves of one window drifted apart after a few dozen iterations.  So every pixel's
  // run leaves in ascending dt (equal dt: equal slots whatever their order): an event takes the slot of its RANK in its run, counted
  // over the run in LDS (O(run) reads per event); a hot pixel (a run beyond kLeanCanon) is first sorted in place by the whole
  // workgroup, a bitonic network.  A staged bin is ranked where it stands; an overfull one is gathered again in chunks of whole
  // pixels (a run that does not fit the staging area on its own keeps its order of arrival).
  if (!staged) {   // an overfull bin re-stages chunks of whole pixels in the WHOLE staging area (both buffers: 2 x sort_cap events)
    s_dt = a_dt;
    s_px = a_px;
    sort_cap *= 2;
  }
  constexpr int kLeanCanon = 1024, kHotList = 32;
  __shared__ int32_t s_hot[kHotList];
  __shared__ int32_t s_nhot, s_nlong;
  auto run_end = [&](int pi) { return pi + 1 < n_pix ? s_cnt[pi + 1] : len; };
  int p_lo = 0;
  while (p_lo < n_pix) {   // (uniform) chunks of whole pixels [p_lo, p_hi): events [c0, c1) of the segment; a staged bin is ONE chunk
    const int c0 = s_cnt[p_lo];
    int p_hi = n_pix;
    if (!staged) {   // the pixels whose runs end within sort_cap events of c0 (binary search on the exclusive offsets)
      int lo = p_lo, hi = n_pix;
      while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (run_end(mid) - c0 <= sort_cap) lo = mid + 1;
        else hi = mid;
      }
      p_hi = lo;
    }
    // (one run larger than the staging area on its own: straight to its slots of the segment, in the order it arrives)
    const bool direct = p_hi == p_lo;
    if (direct) p_hi = p_lo + 1;
    const int c1 = p_hi < n_pix ? s_cnt[p_hi] : len, m = c1 - c0;
    if (threadIdx.x == 0) s_nhot = 0, s_nlong = 0;
    if (!staged) {   // this chunk's pixels out of the bin's runs, each event to its pixel's next slot of the staging area
      __syncthreads();
      auto keep = [&](const Rec& r, int) {
        const int pi = ((int)(r.pix >> 8) - r0) * g.tw + (int)(r.pix & 255u);
        if (pi >= p_lo && pi < p_hi) {
          const int32_t pos = atomicAdd(&s_cur[pi], 1);
          if (direct) {
            cpix[out0 + pos] = (uint16_t)r.pix;
            cdt[out0 + pos] = dt_of(r);
          } else {
            s_px[pos - c0] = (uint16_t)r.pix;
            s_dt[pos - c0] = dt_of(r);
          }
        }
      };
      if (wide) for_each_staged(std::false_type{}, load8, keep);
      else for_each_staged(std::false_type{}, load4, keep);
      if (direct) {
        p_lo = p_hi;
        continue;
      }
    }
    __syncthreads();
    for (int pi = p_lo + threadIdx.x; pi < p_hi; pi += kSortBlock)
      if (run_end(pi) - s_cnt[pi] > kLeanCanon) {
        const int k = atomicAdd(&s_nhot, 1);
        if (k < kHotList) s_hot[k] = pi;
      }
    __syncthreads();
    EBOS_LSTAMP(1, 5);
    // every slot its pixel (a run's slots all carry it): stored FIRST -- the stores travel while the runs are put in order (behind the
    // sorts they were 5 us of a workgroup's 30: a store's way to memory, waited for with nothing else to do)
    for (int i = threadIdx.x; i < m; i += kSortBlock) cpix[out0 + c0 + i] = s_px[i];
    const int n_hot = min(s_nhot, kHotList);
    for (int h = 0; h < n_hot; ++h) {   // (uniform) a hot pixel: bitonic network with every comparator ascending -- the first stage of
      const int pi = s_hot[h];          // a merge pairs i with its mirror image in the block, the others i with i + stride --, so the
      const int rb = s_cnt[pi] - c0, L = run_end(pi) - c0 - rb;   // virtual +inf beyond the run's end never moves: any length, in place
      float* v = s_dt + rb;
      int n2 = 1;
      while (n2 < L) n2 <<= 1;
      for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int t = threadIdx.x; t < n2 / 2; t += kSortBlock) {
            const int blk = t / stride, off = t - blk * stride, lo = blk * 2 * stride + off;
            const int hi = stride == (size >> 1) ? blk * 2 * stride + (2 * stride - 1 - off) : lo + stride;
            if (hi < L) {
              const float x = v[lo], y = v[hi];
              if (sort_key(x) > sort_key(y)) v[lo] = y, v[hi] = x;
            }
          }
          __syncthreads();
        }
      }
    }
    EBOS_LSTAMP(1, 6);
    // Runs of up to 16 events -- nearly every pixel of a window that is not clustered -- are SORTED by sixteen lanes each, a bitonic
    // network over registers (ten compare-exchange stages across lanes, ~6 instructions each: ~4 per event).  Ranking every event
    // against its run cost ~14 instructions per comparison and a run's length squared of them: ~150 per event, 17 of a workgroup's
    // 35 us (VALU-bound: the latency of the LDS reads was not it -- four reads in flight changed nothing).  Equal keys are equal
    // values: the network's instability is invisible.
    // (uniform) a SPARSE chunk -- fewer than four events per pixel: the bands of a 2 M-event window are whole tiles of 3 600 pixels
    // with ~2 events each -- ranks its events one by one instead: the pass over the pixels costs per PIXEL (57 rounds of dependent LDS
    // reads for 7 800 events: 21 us), a rank costs the run's length
    const bool sparse = m < 4 * (p_hi - p_lo);
    if (sparse) {
      for (int i = threadIdx.x; i < m; i += kSortBlock) {
        const unsigned px = (unsigned)s_px[i];
        const int pi = ((int)(px >> 8) - r0) * g.tw + (int)(px & 255u);
        const int rb = s_cnt[pi] - c0, re = run_end(pi) - c0;
        const float d = s_dt[i];
        int slot = i;
        if (re - rb > 1 && re - rb <= kLeanCanon) {   // (a hot run stands sorted already: the network above)
          const int kd = sort_key(d);
          int rank = 0;
          for (int j = rb; j < re; ++j) {
            const int kj = sort_key(s_dt[j]);
            rank += (kj < kd || (kj == kd && j < i)) ? 1 : 0;
          }
          slot = rb + rank;
        }
        cdt[out0 + c0 + slot] = d;
      }
    } else {
      const int sub16 = lane & 15;
      for (int p0 = p_lo; p0 < p_hi; p0 += kSortBlock / 16) {   // (a row of 16 lanes per pixel; whole waves stay in the loop: DPP reads its neighbours)
        const int pi = p0 + (threadIdx.x >> 4);
        const bool live = pi < p_hi;
        const int rb = live ? s_cnt[pi] - c0 : 0, L = live ? run_end(pi) - c0 - rb : 0;
        const bool small = L <= 16;
        // (keys as integers with the floats' order; the padding sorts behind every event and is never written back)
        int key = small && sub16 < L ? sort_key(s_dt[rb + sub16]) : 0x7fffffff;
        // (the longest of the wave's four runs, uniform: ballots of the thresholds -- empty and single-event pixels need no network)
        const int longest = __ballot(small && L > 8) != 0ull ? 16 : (__ballot(small && L > 4) != 0ull ? 8 : (__ballot(small && L > 2) != 0ull ? 4 : (__ballot(small && L > 1) != 0ull ? 2 : 1)));
        if (longest > 1) key = bitonic16(key, sub16, longest);
        if (small && sub16 < L) cdt[out0 + c0 + rb + sub16] = __int_as_float(key ^ ((key >> 31) & 0x7fffffff));   // (sort_key is its own inverse)
        // a longer run (one pixel in sixty of a uniform window) goes on a list (the cursors' array is free by now)
        if (!small && sub16 == 0) s_cur[atomicAdd(&s_nlong, 1)] = pi;
      }
    }
    __syncthreads();
    // ... and the listed runs, 32 lanes each: every event to the slot of its rank in its run -- O(run) LDS reads per event --; a hot run
    // stands sorted already (the network above) and is copied out.  (In the list's order, which is the atomics': the runs are independent.)
    if (!sparse) {
      const int n_long = s_nlong, sub32 = lane & 31;
      for (int q = threadIdx.x >> 5; q < n_long; q += kSortBlock / 32) {
        const int pi = s_cur[q];
        const int rb = s_cnt[pi] - c0, L = run_end(pi) - c0 - rb;
        for (int e = sub32; e < L; e += 32) {
          const float d = s_dt[rb + e];
          int slot = e;
          if (L <= kLeanCanon) {
            const int kd = sort_key(d);
            int rank = 0;
            for (int j = 0; j < L; ++j) {
              const int kj = sort_key(s_dt[rb + j]);
              rank += (kj < kd || (kj == kd && j < e)) ? 1 : 0;
            }
            slot = rank;
          }
          cdt[out0 + c0 + rb + slot] = d;
        }
      }
    }
    EBOS_LSTAMP(1, 7);
    __syncthreads();
    p_lo = p_hi;
  }
  EBOS_LSTAMP(1, 4);
  if (band == g.sub - 1) {  // padding slots of the tile's last group: dt = NaN (no liveness logic in the hot kernels)
    const int64_t end = (int64_t)grp_offsets[tile + 1] * 4;
    for (int64_t o = out0 + len + threadIdx.x; o < end; o += kSortBlock) {
      cpix[o] = 0;
      cdt[o] = __builtin_nanf("");
    }
  }
}

struct LeanLayout {
  int n_chunks, chunk, sub, n_bins, pix_cap, sort_cap;
  size_t off_base, off_tm, off_partial, off_px, off_t, total;
};

inline LeanLayout lean_layout(int64_t n, int H, int W, int th, int tw) {
  LeanLayout L;
  const int n_tiles = ((H + th - 1) / th) * ((W + tw - 1) / tw);
  // chunks of the staging pass: up to kChunk events each, and as many of EQUAL length as keep every round of resident workgroups
  // (two per CU: kResident) full -- 10 M events are 1221 chunks of 8192, 2.4 rounds of which the last is 0.4 full; as 1536 chunks of
  // 6511 they are three full rounds of shorter workgroups.  Small windows: at least one event per thread, at most one round.
  constexpr int64_t kResident = 512;
  int64_t c = (n + kChunk - 1) / kChunk;
  if (c > kResident) c = (c + kResident - 1) / kResident * kResident;
  else c = std::min<int64_t>(kResident, std::max<int64_t>(1, (n + kLeanBlock - 1) / kLeanBlock));
  L.chunk = (int)(((std::max<int64_t>(n, 1) + c - 1) / c + 7) & ~(int64_t)7);
  L.n_chunks = (int)((std::max<int64_t>(n, 1) + L.chunk - 1) / L.chunk);
  // Row bands per tile: as FEW as leave the average bin about two chunks of the bin sort's staging area (sort_cap events are staged
  // and ranked in one go, 2 x sort_cap per chunk of an overfull bin's pixels); at most one band per row.  Fewer bins are longer
  // (chunk, bin) runs -- chunk / n_bins events each -- and a smaller table: what the stage kernel writes per chunk, the totals kernel
  // reads and the bin sort's gather walks.  Round 6 first cut the bins to FIT the staging (one gather): a 10 M-event window on 240
  // tiles got 1920 bins and runs of 4 events, a 50 M-event one 8192 bins and runs of ONE (as many table entries as events; its bin
  // sort alone took 2.25 ms).  With 240 bins (runs of 34) every bin is gathered 3 times (histogram + two chunks of its pixels, out
  // of the L2 / Infinity Cache) and the 10 M build is still 12 % shorter (0.350 -> 0.307 ms; 5 M 0.235 -> 0.204, 20 M 0.82 -> 0.72,
  // 50 M 3.25 -> 2.43 ms; tools/bench_plan_build.py, profiles/r06p_plan_bins_sweep.json).
  constexpr double kBinFill = 3.3;
  for (int sub = 1;; sub *= 2) {
    if (sub > th) sub = th;
    L.sub = sub;
    L.pix_cap = ((th + sub - 1) / sub + 1) * tw;
    const long long room = (long long)kSortLds - (long long)L.pix_cap * 8;
    L.sort_cap = room > 0 ? (int)(room / 12) & ~1 : 0;   // (two staging buffers: arrival order, sorted by pixel)
    if (sub == th || ((double)n / ((double)n_tiles * sub) <= kBinFill * L.sort_cap)) break;
  }
  L.n_bins = n_tiles * L.sub;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.off_base = align((size_t)L.n_bins * L.n_chunks * 4);
  L.off_tm = L.off_base + align((size_t)(L.n_bins + 1) * 4);
  L.off_partial = L.off_tm + 256;
  L.off_px = L.off_partial + align((size_t)L.n_chunks * sizeof(ChunkPartial));
  L.off_t = L.off_px + align((size_t)L.n_chunks * L.chunk * 2);
  L.total = L.off_t + align((size_t)L.n_chunks * L.chunk * 8);   // (sized for 8-byte timestamps: the query does not know the source)
  return L;
}

}  // namespace
}  // namespace ebos

extern "C" {

#ifdef EBOS_LEAN_STAMPS
int ebos_debug_read_lean_stamps(unsigned long long* host, int count) {  // diagnostic builds only
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ebos::g_lean_stamps), sizeof(unsigned long long) * count);
}
#endif

size_t ebos_plan_lean_scratch_bytes(int64_t n, int H, int W, int tile_h, int tile_w) {
  if (n < 0 || H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0) return 0;
  return ebos::lean_layout(n, H, W, tile_h, tile_w).total;
}

int ebos_plan_lean(int source, const void* events, const int16_t* col, const int16_t* row, const void* t, double ticks_per_second,
                   int64_t n, int ref_mode, double ref_fraction, int normalize_t, int H, int W, int tile_h, int tile_w,
                   int32_t* key_offsets, int32_t* grp_offsets, uint16_t* cpix, float* cdt, int64_t capacity_slots, int32_t* counts,
                   double* tminmax, void* scratch, size_t scratch_bytes, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(source >= 0 && source <= 3, "ebos_plan_lean: source must be 0 (f32 [n,4]), 1 (f64 [n,4]), 2 (raw, int32 t), 3 (raw, int64 t)");
  EBOS_REQUIRE(n >= 1 && n < (int64_t)1 << 31, "ebos_plan_lean: 1 <= n < 2^31 (got %lld)", (long long)n);
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && tile_h <= 256 && tile_w <= 256, "ebos_plan_lean: bad sizes (tiles up to 256 x 256)");
  EBOS_REQUIRE(ref_mode >= 0 && ref_mode <= 2, "ebos_plan_lean: ref_mode must be FIRST / LAST / FRACTION");
  EBOS_REQUIRE(source <= 1 ? events != nullptr : (col && row && t && ticks_per_second > 0.0), "ebos_plan_lean: NULL event buffer");
  EBOS_REQUIRE(key_offsets && grp_offsets && cpix && cdt && counts && scratch, "ebos_plan_lean: NULL output / scratch");
  const int tiles_y = (H + tile_h - 1) / tile_h, tiles_x = (W + tile_w - 1) / tile_w, n_tiles = tiles_y * tiles_x;
  const int64_t n_keys = (int64_t)n_tiles * tile_h * tile_w;
  const LeanLayout L = lean_layout(n, H, W, tile_h, tile_w);
  if (L.n_bins > kMaxBins || L.sort_cap < 64) {
    set_error("ebos_plan_lean: %d bins / %d pixels per band exceed what this build sorts in LDS", L.n_bins, L.pix_cap);
    return EBOS_ERR_UNSUPPORTED;
  }
  if (scratch_bytes < L.total) {
    set_error("ebos_plan_lean: scratch too small (%zu < %zu)", scratch_bytes, L.total);
    return EBOS_ERR_SCRATCH;
  }
  if (capacity_slots < n + 3 * (int64_t)n_tiles + 4) {
    set_error("ebos_plan_lean: capacity %lld < n + 3 tiles + 4 = %lld", (long long)capacity_slots, (long long)(n + 3 * (int64_t)n_tiles + 4));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  char* base = reinterpret_cast<char*>(scratch);
  LeanScratch sc{reinterpret_cast<unsigned*>(base), reinterpret_cast<int32_t*>(base + L.off_base),
                 reinterpret_cast<unsigned long long*>(base + L.off_tm), reinterpret_cast<unsigned*>(base + L.off_tm + 64),
                 reinterpret_cast<ChunkPartial*>(base + L.off_partial),
                 reinterpret_cast<uint16_t*>(base + L.off_px), base + L.off_t};
  const bool small = H <= 65536 && W <= 65536;
  const LeanGeom g{H, W, tile_h, tile_w, tiles_x, n_tiles, L.sub, L.n_bins, L.chunk,
                   small ? (unsigned)(0x100000000ull / (unsigned)tile_h) + 1u : 0u, small ? (unsigned)(0x100000000ull / (unsigned)tile_w) + 1u : 0u};
  const LeanIn in{events, col, row, t, ticks_per_second};
  const size_t stamp_bytes = (source == SRC_AOS_F64 || source == SRC_RAW64) ? 8 : 4;
  const size_t lds_stage = (size_t)((L.n_bins + 1) & ~1) * 4 + (size_t)L.chunk * (stamp_bytes + 2);
  const size_t lds_sort = (size_t)(2 * L.pix_cap) * 4 + (size_t)L.sort_cap * 12;
  // (every call, like reserve_lds() of the event kernels: the attribute is per device, and a process may drive several)
  bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(lean_bin_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kSortLds) == hipSuccess;
#define EBOS_LEAN_ATTR(SRC)                                                                                                  \
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(lean_stage_kernel<SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)lds_stage) == hipSuccess
#define EBOS_LEAN(SRC)                                                                                                       \
  do {                                                                                                                       \
    EBOS_LEAN_ATTR(SRC);                                                                                                     \
    if (ok) lean_stage_kernel<SRC><<<dim3(L.n_chunks), dim3(kLeanBlock), lds_stage, s>>>(in, n, g, sc);                      \
  } while (0)
  if (source == SRC_AOS_F32) EBOS_LEAN(SRC_AOS_F32);
  else if (source == SRC_AOS_F64) EBOS_LEAN(SRC_AOS_F64);
  else if (source == SRC_RAW32) EBOS_LEAN(SRC_RAW32);
  else EBOS_LEAN(SRC_RAW64);
#undef EBOS_LEAN
#undef EBOS_LEAN_ATTR
  if (!ok) {
    set_error("ebos_plan_lean: cannot reserve LDS for the staging pass / the bin sort");
    return EBOS_ERR_LAUNCH;
  }
  lean_totals_kernel<<<dim3((L.n_bins + kLeanBlock / kWave - 1) / (kLeanBlock / kWave)), dim3(kLeanBlock), 0, s>>>(g, L.n_chunks, sc, counts, grp_offsets, key_offsets, n_keys,
                                                                                     ticks_per_second, source >= SRC_RAW32, tminmax);
  const LeanTime tc{source, ref_mode, normalize_t, ref_fraction, ticks_per_second};
  lean_bin_sort_kernel<<<dim3(8 * ((L.n_bins + 7) / 8)), dim3(kSortBlock), lds_sort, s>>>(g, sc, grp_offsets, key_offsets, cpix, cdt, L.pix_cap, L.sort_cap,
                                                                          L.n_chunks, tc);
  EBOS_CHECK_LAUNCH("ebos_plan_lean");
  return EBOS_OK;
}

}  // extern "C"
