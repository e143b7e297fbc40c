// plan_lean.hip -- the compact event plan straight from the window, nothing else ("lean" plan build).
//
// The tile-private kernels of the hot loop read only  cpix (u16 tile-local pixel) / cdt (f32 dt) / grp_offsets /
// key_offsets  (iwe_tiled.hip).  The general build (event_plan.hip: SoA conversion -> histogram atomics -> 32-byte record
// scatter -> unpack -> compact fill) also leaves SoA x / y / dt / p and the permutation behind, for per-event weights and
// fractional source coordinates, and pays ~6 GB of traffic and 10 M global atomics for 10 M events (1.3 ms).  When
// neither weights nor weight gradients are wanted, this file builds the same compact arrays with a two-level counting
// sort that never scatters at random into HBM:
//
//   K0 count      one workgroup per chunk of 8192 events: bin = (tile, row band of the tile), LDS histogram -> chunk_hist
//                 [bin][chunk]; min / max of t; out-of-image and fractional-coordinate counts
//   K1 scans      per bin over the chunks; over the bins (bin_base); over the tiles (grp_offsets, padded to groups of 4)
//   K2 partition  same chunks again: dt in fp64 (src/warp.py:264-288), 8-byte record {pixel-in-tile, dt} written to its
//                 bin's segment -- per (chunk, bin) a contiguous run, so the partial lines complete inside the L2
//   K3 bin sort   one workgroup per bin: LDS histogram over the band's pixels -> key_offsets; counting sort of the segment
//                 staged in LDS -> cpix / cdt written coalesced at their final positions (+ NaN padding slots)
//
// Same layout contract as ebos_bin_events_f32 + ebos_plan_compact_f32 (include/ebos_hip.h): key_offsets, grp_offsets,
// cpix, cdt; the order of the events INSIDE one source pixel is unspecified there and here (integer accumulation makes
// the images independent of it).  Reference semantics: src/warp.py:230-288 (reference time, dt), :334 (source pixel =
// trunc), src/data_loader/ccs.py:289-297 (raw columns -> (row, col, t / 1e6, p)).
#include "common.h"

namespace ebos {
namespace {

constexpr int kChunk = 8192;       // events per workgroup of K0 / K2
constexpr int kLeanBlock = 1024;
constexpr int kMaxBins = 8192;     // LDS histogram of K0 / K2
constexpr size_t kSortLds = 158 * 1024;  // dynamic LDS of the bin sort: two int32 per pixel of the band + 6 B per staged event

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
};

// bin of an event (tile-major, then row band) and its pixel inside the tile; bin < 0: outside the image / not finite
__device__ __forceinline__ int event_bin(const LeanGeom& g, float x, float y, unsigned& pix, bool& fractional) {
  fractional = false;
  if (!(x > -1e9f && x < 1e9f && y > -1e9f && y < 1e9f)) return -1;
  const int r = (int)x, c = (int)y;  // truncation toward zero, src/warp.py:334
  if (r < 0 || r >= g.H || c < 0 || c >= g.W) return -1;
  fractional = (x != (float)r) || (y != (float)c) || x < 0.f || y < 0.f;
  const int ty = r / g.th, tx = c / g.tw;
  const int rl = r - ty * g.th, cl = c - tx * g.tw;
  pix = ((unsigned)rl << 8) | (unsigned)cl;
  return (ty * g.tiles_x + tx) * g.sub + (rl * g.sub) / g.th;
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

struct LeanScratch {
  int32_t* chunk_hist;          // [n_bins][n_chunks]  counts, then exclusive offsets inside the bin
  int32_t* bin_base;            // [n_bins + 1]
  unsigned long long* tm;       // [2] ordered min / max of t
  uint2* part;                  // [n] records {pix, dt bits}
};

template <int SRC>
__global__ void __launch_bounds__(kLeanBlock)
lean_count_kernel(LeanIn in, int64_t n, LeanGeom g, int n_chunks, LeanScratch sc, int32_t* __restrict__ counts) {
  extern __shared__ int32_t s_hist[];
  for (int b = threadIdx.x; b < g.n_bins; b += kLeanBlock) s_hist[b] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  double lo = 1.0e308 * 10.0, hi = -1.0e308 * 10.0;  // +-inf
  int bad = 0, frac = 0;
#pragma unroll
  for (int k = 0; k < kChunk / kLeanBlock; ++k) {
    const int64_t i = base + k * kLeanBlock + threadIdx.x;
    if (i < n) {
      const Ev e = read_event<SRC>(in, i);
      lo = e.t < lo ? e.t : lo;  // (all events, like calculate_reftime, src/warp.py:245-253)
      hi = e.t > hi ? e.t : hi;
      unsigned pix;
      bool fr;
      const int b = event_bin(g, e.x, e.y, pix, fr);
      if (b >= 0) atomicAdd(&s_hist[b], 1);
      else ++bad;
      frac += fr;
    }
  }
  // one pair of global atomics per WORKGROUP: same-address atomics serialise at ~88 per microsecond on this chip (one pair
  // per wavefront -- 312 000 of them for 10 M events -- made this kernel 0.47 ms instead of 0.05)
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
    if (lo <= hi) {
      atomicMin(&sc.tm[0], ordered(lo));
      atomicMax(&sc.tm[1], ordered(hi));
    }
    if (bad) atomicAdd(&counts[0], bad);
    if (frac) atomicAdd(&counts[1], frac);
  }
  for (int b = threadIdx.x; b < g.n_bins; b += kLeanBlock) sc.chunk_hist[(int64_t)b * n_chunks + blockIdx.x] = s_hist[b];
}

// exclusive scan over the chunks of one bin (in place); bin total -> bin_base[bin] (scanned by the next kernel)
__global__ void __launch_bounds__(256) lean_scan_chunks_kernel(int n_chunks, LeanScratch sc) {
  __shared__ int32_t s_wave[4];
  __shared__ int32_t s_carry;
  int32_t* h = sc.chunk_hist + (int64_t)blockIdx.x * n_chunks;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  for (int start = 0; start < n_chunks; start += 256) {
    const int c = start + threadIdx.x;
    const int32_t v = c < n_chunks ? h[c] : 0;
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
    if (c < n_chunks) h[c] = carry + wave_off + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = carry + wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) sc.bin_base[blockIdx.x] = s_carry;
}

// one workgroup: bin totals -> bin_base (exclusive, + total); tile totals -> grp_offsets (groups of 4 slots, exclusive);
// key_offsets[n_keys] = events kept; tminmax (seconds) for the caller
__global__ void __launch_bounds__(1024)
lean_scan_bins_kernel(LeanGeom g, LeanScratch sc, int32_t* __restrict__ grp_offsets, int32_t* __restrict__ key_offsets,
                      int64_t n_keys, double ticks_per_second, int raw, double* __restrict__ tminmax) {
  __shared__ int32_t s_wave[1024 / kWave];
  __shared__ int32_t s_carry;
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
         for (int s = 0; s < g.sub; ++s) c += sc.bin_base[t * g.sub + s];
         return (c + 3) >> 2;
       },
       [&](int t, int32_t pre) { grp_offsets[t] = pre; });
  if (threadIdx.x == 0) grp_offsets[g.n_tiles] = s_carry;
  __syncthreads();
  scan(g.n_bins, [&](int b) { return sc.bin_base[b]; }, [&](int b, int32_t pre) { sc.bin_base[b] = pre; });
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

template <int SRC>
__global__ void __launch_bounds__(kLeanBlock)
lean_partition_kernel(LeanIn in, int64_t n, LeanGeom g, int n_chunks, LeanScratch sc, int ref_mode, double ref_fraction,
                      int normalize_t) {
  extern __shared__ int32_t s_cur[];  // per bin: next free slot of this chunk's run inside the bin's segment
  for (int b = threadIdx.x; b < g.n_bins; b += kLeanBlock)
    s_cur[b] = sc.bin_base[b] + sc.chunk_hist[(int64_t)b * n_chunks + blockIdx.x];
  // reference time and period in fp64, exactly as events_to_soa_kernel / raw_to_soa_kernel (event_plan.hip)
  constexpr bool kRaw = SRC == SRC_RAW32 || SRC == SRC_RAW64;
  double tmin = unordered(sc.tm[0]), tmax = unordered(sc.tm[1]);
  if (kRaw) {
    tmin = tmin / in.ticks_per_second;
    tmax = tmax / in.ticks_per_second;
  } else if (SRC == SRC_AOS_F32) {
    tmin = (double)(float)tmin;  // (exact: they are f32 values)
    tmax = (double)(float)tmax;
  }
  double ref;
  if (ref_mode == EBOS_REF_FIRST) ref = tmin;
  else if (ref_mode == EBOS_REF_LAST) ref = tmax;
  else ref = tmin + (tmax - tmin) * ref_fraction;
  const double inv_period = normalize_t ? 1.0 / (tmax - tmin) : 1.0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kChunk;
#pragma unroll
  for (int k = 0; k < kChunk / kLeanBlock; ++k) {
    const int64_t i = base + k * kLeanBlock + threadIdx.x;
    if (i < n) {
      const Ev e = read_event<SRC>(in, i);
      unsigned pix;
      bool fr;
      const int b = event_bin(g, e.x, e.y, pix, fr);
      if (b >= 0) {
        const double ts = kRaw ? e.t / in.ticks_per_second : e.t;
        const float dt = (float)((ts - ref) * inv_period);
        const int32_t pos = atomicAdd(&s_cur[b], 1);
        sc.part[pos] = make_uint2(pix, (unsigned)__float_as_int(dt));
      }
    }
  }
}

// a float's bits as an integer with the floats' order -- a TOTAL order: a NaN (a broken timestamp) still gets a rank of its own, where
// float comparisons would hand several events the same slot
__device__ __forceinline__ int sort_key(float d) {
  const int b = __float_as_int(d);
  return b ^ ((b >> 31) & 0x7fffffff);
}

// one workgroup per bin: counting sort of the bin's segment by pixel -> key_offsets of the band, cpix / cdt at their final slots
__global__ void __launch_bounds__(kLeanBlock)
lean_bin_sort_kernel(LeanGeom g, LeanScratch sc, const int32_t* __restrict__ grp_offsets, int32_t* __restrict__ key_offsets,
                     uint16_t* __restrict__ cpix, float* __restrict__ cdt, int pix_cap, int sort_cap) {
  // sort_cap: events of one bin staged in LDS (6 B each); a larger bin scatters to global memory
  extern __shared__ int32_t s_raw[];
  int32_t* s_cnt = s_raw;                 // [pix_cap]  events per pixel of the band, then exclusive offsets
  int32_t* s_cur = s_raw + pix_cap;       // [pix_cap]  cursors
  float* s_dt = reinterpret_cast<float*>(s_raw + 2 * pix_cap);        // [sort_cap]
  uint16_t* s_px = reinterpret_cast<uint16_t*>(s_dt + sort_cap);      // [sort_cap]
  __shared__ int32_t s_wave[kLeanBlock / kWave];
  __shared__ int32_t s_carry;
  const int bin = blockIdx.x, tile = bin / g.sub, band = bin - tile * g.sub;
  const int r0 = band_row0(g, band), r1 = band_row0(g, band + 1);  // rows [r0, r1) of the tile
  const int n_pix = (r1 - r0) * g.tw;
  const int32_t seg0 = sc.bin_base[bin], seg1 = sc.bin_base[bin + 1], len = seg1 - seg0;
  const int64_t first_key = (int64_t)tile * g.th * g.tw + (int64_t)r0 * g.tw;
  const int32_t tile_first = sc.bin_base[tile * g.sub];            // events before this tile
  const int64_t out0 = (int64_t)grp_offsets[tile] * 4 + (seg0 - tile_first);  // final slot of the segment's first event
  for (int i = threadIdx.x; i < n_pix; i += kLeanBlock) s_cnt[i] = 0;
  __syncthreads();
  const uint2* part = sc.part + seg0;
  for (int i = threadIdx.x; i < len; i += kLeanBlock) {
    const unsigned pix = part[i].x;
    atomicAdd(&s_cnt[((int)(pix >> 8) - r0) * g.tw + (int)(pix & 255u)], 1);
  }
  __syncthreads();
  // exclusive scan of s_cnt [n_pix]
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int start = 0; start < n_pix; start += kLeanBlock) {
    const int i = start + threadIdx.x;
    const int32_t v = i < n_pix ? s_cnt[i] : 0;
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
    if (i < n_pix) {
      const int32_t ex = carry + wave_off + inc - v;
      s_cnt[i] = ex;
      s_cur[i] = ex;
      key_offsets[first_key + i] = seg0 + ex;
    }
    __syncthreads();
    if (threadIdx.x == kLeanBlock - 1) s_carry = carry + wave_off + inc;
    __syncthreads();
  }
  const bool staged = len <= sort_cap;
  for (int i = threadIdx.x; i < len; i += kLeanBlock) {
    const uint2 r = part[i];
    const int32_t pos = atomicAdd(&s_cur[((int)(r.x >> 8) - r0) * g.tw + (int)(r.x & 255u)], 1);
    if (staged) {
      s_px[pos] = (uint16_t)r.x;
      s_dt[pos] = __int_as_float((int)r.y);
    } else {  // an overfull bin (a window far from uniform): straight to its final slot
      cpix[out0 + pos] = (uint16_t)r.x;
      cdt[out0 + pos] = __int_as_float((int)r.y);
    }
  }
  __syncthreads();
  // The cursors hand out a pixel's slots in the order the atomics arrive: two builds of one window differ in it, and a kernel that
  // sums a group's events in slot order before it accumulates exactly (the 2-DoF backward sweep) sees that in the last bit of its
  // gradient -- which an optimiser amplifies: two solves of one window drifted apart after a few dozen iterations.  So every pixel's
  // run leaves in ascending dt (equal dt: equal slots whatever their order): an event takes the slot of its RANK in its run, counted
  // over the run in LDS (O(run) reads per event); a hot pixel (a run beyond kLeanCanon) is first sorted in place by the whole
  // workgroup, a bitonic network.  A staged bin is ranked where it stands; an overfull one comes back from memory in chunks of whole
  // pixels (a run that does not fit the staging area on its own keeps its order of arrival).
  constexpr int kLeanCanon = 1024, kHotList = 32;
  __shared__ int32_t s_hot[kHotList];
  __shared__ int32_t s_nhot;
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
      if (p_hi == p_lo) {   // one run larger than the staging area: left as it arrived
        p_lo += 1;
        continue;
      }
    }
    const int c1 = p_hi < n_pix ? s_cnt[p_hi] : len, m = c1 - c0;
    if (threadIdx.x == 0) s_nhot = 0;
    if (!staged) {
      __threadfence();
      __syncthreads();
      // (written by this workgroup a moment ago: not from this CU's L1; eight loads per thread in flight -- one at a time the ~20
      // round trips of a chunk were most of an overfull bin's time)
      constexpr int kU = 8;
      for (int i0 = threadIdx.x; i0 < m; i0 += kU * kLeanBlock) {
        unsigned v[kU];
        uint16_t q[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          const int i = min(i0 + u * kLeanBlock, m - 1);
          v[u] = __hip_atomic_load(reinterpret_cast<unsigned*>(cdt + out0 + c0 + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q[u] = __hip_atomic_load(cpix + out0 + c0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (i0 + u * kLeanBlock < m) s_dt[i0 + u * kLeanBlock] = __int_as_float((int)v[u]), s_px[i0 + u * kLeanBlock] = q[u];
      }
    }
    __syncthreads();
    for (int pi = p_lo + threadIdx.x; pi < p_hi; pi += kLeanBlock)
      if (run_end(pi) - s_cnt[pi] > kLeanCanon) {
        const int k = atomicAdd(&s_nhot, 1);
        if (k < kHotList) s_hot[k] = pi;
      }
    __syncthreads();
    const int n_hot = min(s_nhot, kHotList);
    for (int h = 0; h < n_hot; ++h) {   // (uniform) a hot pixel: bitonic network with every comparator ascending -- the first stage of
      const int pi = s_hot[h];          // a merge pairs i with its mirror image in the block, the others i with i + stride --, so the
      const int rb = s_cnt[pi] - c0, L = run_end(pi) - c0 - rb;   // virtual +inf beyond the run's end never moves: any length, in place
      float* v = s_dt + rb;
      int n2 = 1;
      while (n2 < L) n2 <<= 1;
      for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
          for (int t = threadIdx.x; t < n2 / 2; t += kLeanBlock) {
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
    // every event to the slot of its rank in its pixel's run (the hot runs stand sorted already, as do those of a list that overflowed
    // -- as they arrived)
    for (int i = threadIdx.x; i < m; i += kLeanBlock) {
      const unsigned px = (unsigned)s_px[i];
      const int pi = ((int)(px >> 8) - r0) * g.tw + (int)(px & 255u);
      const int rb = s_cnt[pi] - c0, re = run_end(pi) - c0;
      const float d = s_dt[i];
      const int kd = sort_key(d);
      int slot = i;
      if (re - rb > 1 && re - rb <= kLeanCanon) {
        int rank = 0;
        for (int j = rb; j < re; ++j) {
          const int kj = sort_key(s_dt[j]);
          rank += (kj < kd || (kj == kd && j < i)) ? 1 : 0;
        }
        slot = rb + rank;
      }
      if (staged) cpix[out0 + i] = (uint16_t)px;   // (a run's slots all carry its pixel)
      cdt[out0 + c0 + slot] = d;
    }
    __syncthreads();
    p_lo = p_hi;
  }
  if (band == g.sub - 1) {  // padding slots of the tile's last group: dt = NaN (no liveness logic in the hot kernels)
    const int64_t end = (int64_t)grp_offsets[tile + 1] * 4;
    for (int64_t o = out0 + len + threadIdx.x; o < end; o += kLeanBlock) {
      cpix[o] = 0;
      cdt[o] = __builtin_nanf("");
    }
  }
}

__global__ void lean_init_kernel(LeanScratch sc, int32_t* counts) {
  sc.tm[0] = ~0ull;
  sc.tm[1] = 0ull;
  counts[0] = 0;
  counts[1] = 0;
}

struct LeanLayout {
  int n_chunks, sub, n_bins, pix_cap, sort_cap;
  size_t off_base, off_tm, off_part, total;
};

inline LeanLayout lean_layout(int64_t n, int H, int W, int th, int tw) {
  LeanLayout L;
  const int n_tiles = ((H + th - 1) / th) * ((W + tw - 1) / tw);
  L.n_chunks = (int)((n + kChunk - 1) / kChunk);
  if (L.n_chunks < 1) L.n_chunks = 1;
  // row bands per tile: as few as keep the average bin inside the LDS staging of the bin sort (uniform windows then never
  // take the global-scatter branch); at most one band per row
  for (int sub = 1;; sub *= 2) {
    if (sub > th) sub = th;
    L.sub = sub;
    L.pix_cap = ((th + sub - 1) / sub + 1) * tw;
    const long long room = (long long)kSortLds - (long long)L.pix_cap * 8;
    L.sort_cap = room > 0 ? (int)(room / 6) & ~1 : 0;
    if (sub == th || ((double)n / ((double)n_tiles * sub) <= 0.8 * L.sort_cap)) break;
  }
  L.n_bins = n_tiles * L.sub;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.off_base = align((size_t)L.n_bins * L.n_chunks * 4);
  L.off_tm = L.off_base + align((size_t)(L.n_bins + 1) * 4);
  L.off_part = L.off_tm + 256;
  L.total = L.off_part + align((size_t)(n > 0 ? n : 1) * 8);
  return L;
}

}  // namespace
}  // namespace ebos

extern "C" {

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
  LeanScratch sc{reinterpret_cast<int32_t*>(base), reinterpret_cast<int32_t*>(base + L.off_base),
                 reinterpret_cast<unsigned long long*>(base + L.off_tm), reinterpret_cast<uint2*>(base + L.off_part)};
  const LeanGeom g{H, W, tile_h, tile_w, tiles_x, n_tiles, L.sub, L.n_bins};
  const LeanIn in{events, col, row, t, ticks_per_second};
  const size_t lds_bins = (size_t)L.n_bins * 4;
  const size_t lds_sort = (size_t)(2 * L.pix_cap) * 4 + (size_t)L.sort_cap * 6;
  // (every call, like reserve_lds() of the event kernels: the attribute is per device, and a process may drive several)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lean_bin_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kSortLds) != hipSuccess) {
    set_error("ebos_plan_lean: cannot reserve LDS for the bin sort");
    return EBOS_ERR_LAUNCH;
  }
  lean_init_kernel<<<dim3(1), dim3(1), 0, s>>>(sc, counts);
#define EBOS_LEAN(SRC)                                                                                                      \
  do {                                                                                                                     \
    lean_count_kernel<SRC><<<dim3(L.n_chunks), dim3(kLeanBlock), lds_bins, s>>>(in, n, g, L.n_chunks, sc, counts);         \
    lean_scan_chunks_kernel<<<dim3(L.n_bins), dim3(256), 0, s>>>(L.n_chunks, sc);                                          \
    lean_scan_bins_kernel<<<dim3(1), dim3(1024), 0, s>>>(g, sc, grp_offsets, key_offsets, n_keys, ticks_per_second,        \
                                                         SRC >= SRC_RAW32, tminmax);                                       \
    lean_partition_kernel<SRC><<<dim3(L.n_chunks), dim3(kLeanBlock), lds_bins, s>>>(in, n, g, L.n_chunks, sc, ref_mode,    \
                                                                                   ref_fraction, normalize_t);             \
  } while (0)
  if (source == SRC_AOS_F32) EBOS_LEAN(SRC_AOS_F32);
  else if (source == SRC_AOS_F64) EBOS_LEAN(SRC_AOS_F64);
  else if (source == SRC_RAW32) EBOS_LEAN(SRC_RAW32);
  else EBOS_LEAN(SRC_RAW64);
#undef EBOS_LEAN
  lean_bin_sort_kernel<<<dim3(L.n_bins), dim3(kLeanBlock), lds_sort, s>>>(g, sc, grp_offsets, key_offsets, cpix, cdt, L.pix_cap, L.sort_cap);
  EBOS_CHECK_LAUNCH("ebos_plan_lean");
  return EBOS_OK;
}

}  // extern "C"
