// event_plan.hip -- builds the device-resident, iteration-invariant form of one event window.
//
// (x, y, t, p) of every event are constant across solver iterations / flow hypotheses; only the
// motion changes.  So the window is converted ONCE to struct-of-arrays f32
// (x, y, dt, p) with dt evaluated in fp64 (src/warp.py:264-288 semantics), and counting-sorted by
// source pixel, tile-major, so that the fused kernels of iwe_fused.hip see
//   - coalesced SoA loads,
//   - all events of one image tile contiguous (LDS-privatised IWE tile per workgroup),
//   - all events of one source pixel contiguous (flow reads broadcast; the backward pass
//     pre-reduces d_flow across the wavefront and issues one atomic per pixel run).
#include "common.h"

namespace ebos {
namespace {

template <typename T>
__global__ void __launch_bounds__(256)
events_to_soa_kernel(const T* __restrict__ events, const T* __restrict__ tminmax, int ref_mode, double ref_fraction,
                     int normalize_t, int64_t n, float* __restrict__ x, float* __restrict__ y, float* __restrict__ dt,
                     float* __restrict__ p) {
  // reference time and period in fp64 (the f32 API path keeps the reference's own rounding;
  // the plan is allowed to be closer to the fp64 reference)
  const double tmin = (double)tminmax[0], tmax = (double)tminmax[1];
  double ref;
  if (ref_mode == EBOS_REF_FIRST) ref = tmin;
  else if (ref_mode == EBOS_REF_LAST) ref = tmax;
  else ref = tmin + (tmax - tmin) * ref_fraction;
  double inv_period = normalize_t ? 1.0 / (tmax - tmin) : 1.0;
  if (ref_mode == EBOS_REF_TIMEBASE) {  // (t_ref, period) given
    ref = tmin;
    inv_period = normalize_t ? 1.0 / tmax : 1.0;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T ex = events[4 * i], ey = events[4 * i + 1], et = events[4 * i + 2], ep = events[4 * i + 3];
    x[i] = (float)ex;
    y[i] = (float)ey;
    dt[i] = (float)(((double)et - ref) * inv_period);
    p[i] = (float)ep;
  }
}

// ---- raw sensor columns -> SoA (the event-ingest format) ----------------------------
// The CCS recordings store raw_events/{x: int16 column, y: int16 row, t: int32 microseconds, p: bool}
// (src/data_loader/ccs.py:57-66); the reference expands a window to float64 [n, 4] =
// (y, x, t / 1e6, p) on the host (:289-297).  Here the window stays in its 9 B/event raw form on the
// device and is expanded to the SoA plan directly, with the same fp64 time arithmetic.
template <typename TT>
__global__ void __launch_bounds__(256)
raw_time_range_kernel(const TT* __restrict__ t, int64_t n, long long* __restrict__ tminmax_ticks) {
  long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long v = (long long)t[i];
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    __hip_atomic_fetch_min(&tminmax_ticks[0], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(&tminmax_ticks[1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void raw_time_range_init_kernel(long long* tminmax_ticks) {
  tminmax_ticks[0] = 0x7fffffffffffffffLL;
  tminmax_ticks[1] = -0x7fffffffffffffffLL - 1;
}

__global__ void raw_time_range_seconds_kernel(const long long* tminmax_ticks, double ticks_per_second, double* tminmax) {
  tminmax[0] = (double)tminmax_ticks[0] / ticks_per_second;  // t / 1e6, src/data_loader/ccs.py:295
  tminmax[1] = (double)tminmax_ticks[1] / ticks_per_second;
}

template <typename TT>
__global__ void __launch_bounds__(256)
raw_to_soa_kernel(const int16_t* __restrict__ col, const int16_t* __restrict__ row, const TT* __restrict__ t,
                  const uint8_t* __restrict__ pol, double ticks_per_second, const double* __restrict__ tminmax,
                  int ref_mode, double ref_fraction, int normalize_t, int64_t n, float* __restrict__ x,
                  float* __restrict__ y, float* __restrict__ dt, float* __restrict__ p) {
  const double tmin = tminmax[0], tmax = tminmax[1];
  double ref;
  if (ref_mode == EBOS_REF_FIRST) ref = tmin;
  else if (ref_mode == EBOS_REF_LAST) ref = tmax;
  else ref = tmin + (tmax - tmin) * ref_fraction;
  double inv_period = normalize_t ? 1.0 / (tmax - tmin) : 1.0;
  if (ref_mode == EBOS_REF_TIMEBASE) {
    ref = tmin;
    inv_period = normalize_t ? 1.0 / tmax : 1.0;
  }
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = (float)row[i];  // events[:, 0] = y (row), :293
    y[i] = (float)col[i];  // events[:, 1] = x (column), :294
    dt[i] = (float)(((double)t[i] / ticks_per_second - ref) * inv_period);
    p[i] = pol[i] ? 1.0f : 0.0f;
  }
}

__device__ __forceinline__ int source_key(float x, float y, int H, int W, int tile_h, int tile_w, int tiles_x) {
  if (!(x > -1e9f && x < 1e9f && y > -1e9f && y < 1e9f)) return -1;
  const int r = (int)x, c = (int)y;  // truncation toward zero, src/warp.py:334
  if (r < 0 || r >= H || c < 0 || c >= W) return -1;  // (-1, 0) truncates to pixel 0 like .long()
  const int ty = r / tile_h, tx = c / tile_w;
  return (ty * tiles_x + tx) * (tile_h * tile_w) + (r - ty * tile_h) * tile_w + (c - tx * tile_w);
}

__global__ void __launch_bounds__(256)
bin_hist_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n, int H, int W, int tile_h,
                int tile_w, int tiles_x, int32_t* hist, int32_t* oob_count, int32_t* __restrict__ rank) {
  // rank (nullable): the value the histogram atomic returns is the event's rank inside its key -- kept, it saves the
  // scatter pass its own 10 M atomics on a cursor array
  int bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int key = source_key(x[i], y[i], H, W, tile_h, tile_w, tiles_x);
    if (key >= 0) {
      const int32_t r = atomicAdd(&hist[key], 1);
      if (rank != nullptr) rank[i] = r;
    } else {
      ++bad;
    }
  }
  if (oob_count != nullptr && bad) atomicAdd(oob_count, bad);
}

constexpr int kScanBlock = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanBlock * kScanItems;

// exclusive scan of one 4096-item tile in place; tile total -> block_sums[blockIdx.x]
__global__ void __launch_bounds__(kScanBlock) scan_tiles_kernel(int32_t* data, int64_t n, int32_t* block_sums) {
  __shared__ int32_t s_wave[kScanBlock / kWave];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t v[kScanItems];
  int32_t sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    sum += v[k];
  }
  // inclusive scan of per-thread sums across the wave, then across the 4 waves
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  int32_t inc = sum;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int32_t o = __shfl_up(inc, off, kWave);
    if (lane >= off) inc += o;
  }
  if (lane == kWave - 1) s_wave[wid] = inc;
  __syncthreads();
  int32_t wave_off = 0;
  for (int k = 0; k < wid; ++k) wave_off += s_wave[k];
  int32_t run = wave_off + inc - sum;  // exclusive prefix of this thread
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == kScanBlock - 1) block_sums[blockIdx.x] = run;
}

// single workgroup: exclusive scan of the tile totals, grand total -> *total_out
__global__ void __launch_bounds__(kScanBlock) scan_block_sums_kernel(int32_t* block_sums, int nblk, int32_t* total_out) {
  __shared__ int32_t s_wave[kScanBlock / kWave];
  __shared__ int32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  for (int start = 0; start < nblk; start += kScanBlock) {
    const int i = start + threadIdx.x;
    const int32_t v = (i < nblk) ? block_sums[i] : 0;
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
    if (i < nblk) block_sums[i] = carry + wave_off + inc - v;
    __syncthreads();
    if (threadIdx.x == kScanBlock - 1) s_carry = carry + wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = s_carry;
}

__global__ void __launch_bounds__(kScanBlock) scan_add_offsets_kernel(int32_t* data, int64_t n, const int32_t* block_sums) {
  const int32_t off = block_sums[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  for (int k = threadIdx.x; k < kScanTile; k += kScanBlock)
    if (base + k < n) data[base + k] += off;
}

__global__ void __launch_bounds__(256)
bin_scatter_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                   const float* __restrict__ p, int64_t n, int H, int W, int tile_h, int tile_w, int tiles_x,
                   const int32_t* __restrict__ key_offsets, int32_t* cursor, float* __restrict__ xs,
                   float* __restrict__ ys, float* __restrict__ dts, float* __restrict__ ps, int32_t* __restrict__ perm,
                   int32_t* frac_count) {
  int fractional = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float ex = x[i], ey = y[i];
    const int key = source_key(ex, ey, H, W, tile_h, tile_w, tiles_x);
    if (key < 0) continue;
    const int32_t pos = key_offsets[key] + atomicAdd(&cursor[key], 1);
    xs[pos] = ex;
    ys[pos] = ey;
    dts[pos] = dt[i];
    if (ps != nullptr) ps[pos] = p ? p[i] : 0.f;
    if (perm != nullptr) perm[pos] = (int32_t)i;
    fractional += (ex != (float)(int)ex) || (ey != (float)(int)ey) || ex < 0.f || ey < 0.f;
  }
  if (frac_count != nullptr && fractional) atomicAdd(frac_count, fractional);
}

// Scatter as ONE aligned 32-byte record per event instead of five 4-byte stores to five arrays.  A random 4-byte store
// costs a whole memory sector (read-modify-write under ECC): the five-array scatter of 10 M events took 1.26 ms, i.e.
// ~6 GB of HBM traffic for 200 MB of payload (a two-level sort through LDS that keeps the five small stores inside each
// tile's own range was no faster -- 32 tiles x 780 KB per XCD do not stay in a 4 MB L2 until their lines are complete).
// The records are then unpacked to the SoA arrays by a streaming pass.
struct alignas(32) EventRecord {
  float x, y, dt, p;
  int32_t idx;
  int32_t pad[3];
};

__global__ void __launch_bounds__(256)
bin_scatter_records_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                           const float* __restrict__ p, int64_t n, int H, int W, int tile_h, int tile_w, int tiles_x,
                           const int32_t* __restrict__ key_offsets, const int32_t* __restrict__ rank,
                           EventRecord* __restrict__ rec, int32_t* frac_count) {
  int fractional = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float ex = x[i], ey = y[i];
    const int key = source_key(ex, ey, H, W, tile_h, tile_w, tiles_x);
    if (key < 0) continue;
    const int32_t pos = key_offsets[key] + rank[i];
    EventRecord r;
    r.x = ex;
    r.y = ey;
    r.dt = dt[i];
    r.p = p ? p[i] : 0.f;
    r.idx = (int32_t)i;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    rec[pos] = r;  // two 16-byte stores into one 32-byte sector
    fractional += (ex != (float)(int)ex) || (ey != (float)(int)ey) || ex < 0.f || ey < 0.f;
  }
  if (frac_count != nullptr && fractional) atomicAdd(frac_count, fractional);
}

__global__ void __launch_bounds__(256)
unpack_records_kernel(const EventRecord* __restrict__ rec, const int32_t* __restrict__ total, float* __restrict__ xs,
                      float* __restrict__ ys, float* __restrict__ dts, float* __restrict__ ps, int32_t* __restrict__ perm) {
  const int64_t kept = total[0];  // = key_offsets[n_keys]
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < kept; i += (int64_t)gridDim.x * blockDim.x) {
    const EventRecord r = rec[i];
    xs[i] = r.x;
    ys[i] = r.y;
    dts[i] = r.dt;
    if (ps != nullptr) ps[i] = r.p;
    if (perm != nullptr) perm[i] = r.idx;
  }
}

// ---- compact plan: the 6 B/event layout read by the tile-private kernels --------------------------------------
// Per tile t the events occupy groups [grp_offsets[t], grp_offsets[t+1]) of 4 slots; a slot is
//   cpix (u16) = (row_in_tile << 8) | col_in_tile      cdt (f32) = dt, NaN in the padding slots of the last group
// so the kernels need no liveness logic at all (a NaN dt makes every tap fall outside the LDS window).
__global__ void __launch_bounds__(256)
compact_offsets_kernel(const int32_t* __restrict__ key_offsets, int tile_px, int n_tiles, int32_t* grp_offsets) {
  __shared__ int32_t s_wave[4];
  __shared__ int32_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  for (int start = 0; start < n_tiles; start += 256) {
    const int t = start + threadIdx.x;
    int32_t v = 0;
    if (t < n_tiles) v = (key_offsets[(int64_t)(t + 1) * tile_px] - key_offsets[(int64_t)t * tile_px] + 3) >> 2;
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
    if (t < n_tiles) grp_offsets[t] = carry + wave_off + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = carry + wave_off + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) grp_offsets[n_tiles] = s_carry;
}

// cfx / cfy (nullable): the fractional parts x - floor(x), y - floor(y) of the source coordinates per slot -- the compact plan of a
// window of undistorted events (ebos_plan_compact_frac_f32)
constexpr int kCanonMax = 64;
// a float's bits as an integer with the floats' order -- a total order (NaNs rank too: no two events take one slot)
__device__ __forceinline__ int sort_key(float d) {
  const int b = __float_as_int(d);
  return b ^ ((b >> 31) & 0x7fffffff);
}
__global__ void __launch_bounds__(256)
compact_fill_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ dts,
                    const int32_t* __restrict__ key_offsets, int tile_h, int tile_w, int tiles_x,
                    const int32_t* __restrict__ grp_offsets, uint16_t* __restrict__ cpix, float* __restrict__ cdt,
                    float* __restrict__ cfx = nullptr, float* __restrict__ cfy = nullptr) {
  const int t = blockIdx.x;
  const int tile_px = tile_h * tile_w;
  const int32_t beg = key_offsets[(int64_t)t * tile_px], end = key_offsets[(int64_t)(t + 1) * tile_px];
  const int64_t out0 = (int64_t)grp_offsets[t] * 4, out1 = (int64_t)grp_offsets[t + 1] * 4;
  const int r0 = (t / tiles_x) * tile_h, c0 = (t % tiles_x) * tile_w;
  for (int64_t o = out0 + blockIdx.y * blockDim.x + threadIdx.x; o < out1; o += (int64_t)gridDim.y * blockDim.x) {
    const int64_t src = beg + (o - out0);
    if (src < end) {
      const float x = xs[src], y = ys[src];
      const int r = (int)x - r0, c = (int)y - c0;
      if (cfx != nullptr) {  // (what load_group computes for the (x, y, dt) format: the same numbers)
        const float dt = dts[src], fx = x - (float)(int)x, fy = y - (float)(int)y;
        // The binning scatter leaves the events of one source pixel in the order its atomics arrived: two builds of one window
        // differ in it.  For integer source pixels that order is invisible (and the lean build's is canonical); with fractions per
        // slot the kernels' run sums see it in their last bits, and an optimiser loop amplifies those -- two solves of one window
        // drifted apart.  So an event takes the slot of its RANK in its pixel's run by (dt, fx, fy) (equal events: equal slots
        // whatever their order); a run beyond kCanonMax events -- a hot pixel -- is sorted by compact_canon_hot_kernel afterwards.
        const int64_t key = (int64_t)t * tile_px + r * tile_w + c;
        const int32_t kb = key_offsets[key], ke = key_offsets[key + 1];
        int64_t slot = o;
        if (ke - kb > 1 && ke - kb <= kCanonMax) {
          const int kdt = sort_key(dt), kfx = sort_key(fx), kfy = sort_key(fy);
          int rank = 0;
          for (int32_t j0 = kb; j0 < ke; j0 += 4) {  // (four events' loads in flight: the longest run's chain of round trips bounds the pass)
            float xj[4], yj[4], dj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int32_t j = min(j0 + u, ke - 1);
              xj[u] = xs[j], yj[u] = ys[j], dj[u] = dts[j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int32_t j = j0 + u;
              const float fxj = xj[u] - (float)(int)xj[u], fyj = yj[u] - (float)(int)yj[u];
              const int kd = sort_key(dj[u]), kx = sort_key(fxj), ky = sort_key(fyj);   // (a total order: NaNs rank too)
              const bool before = kd < kdt || (kd == kdt && (kx < kfx || (kx == kfx && (ky < kfy || (ky == kfy && j < src)))));
              rank += (j < ke && before) ? 1 : 0;
            }
          }
          slot = out0 + (kb - beg) + rank;
        }
        cpix[slot] = (uint16_t)((r << 8) | c);
        cdt[slot] = dt;
        cfx[slot] = fx;
        cfy[slot] = fy;
      } else {
        cpix[o] = (uint16_t)((r << 8) | c);
        cdt[o] = dts[src];
      }
    } else {
      cpix[o] = 0;
      cdt[o] = __builtin_nanf("");
      if (cfx != nullptr) cfx[o] = 0.0f, cfy[o] = 0.0f;
    }
  }
}

// ... and the runs beyond kCanonMax (hot pixels: one sensor pixel firing hundreds of times in a window) are sorted in place by the same
// key, one workgroup per tile, a bitonic network in LDS over up to kCanonHot events at a time.  Equal keys are equal slots, so the
// network's instability is invisible.  The hot runs of a tile are taken in the order of their pixel index -- found by a block-wide
// minimum per pass, as many passes as there are hot runs: no capped list, no dependence on the order atomics arrive in (a noisy sensor
// with more than 64 hot pixels in a tile used to lose the canonical order silently).  A run LONGER than kCanonHot is sorted
// by block-level odd-even merge-splitting: its chunks of kCanonHot / 2 events are sorted pairwise (chunk c with c + 1, the smaller half
// back to c), pairs of alternating parity, as many phases as there are chunks -- which leaves the run sorted, in place, without scratch.
// A tile without a hot run -- nearly every tile -- leaves after one pass over its key offsets.
constexpr int kCanonHot = 4096;
__global__ void __launch_bounds__(1024)
compact_canon_hot_kernel(const int32_t* __restrict__ key_offsets, int tile_px, const int32_t* __restrict__ grp_offsets,
                         float* __restrict__ cdt, float* __restrict__ cfx, float* __restrict__ cfy) {
  __shared__ float s_dt[kCanonHot], s_fx[kCanonHot], s_fy[kCanonHot];
  __shared__ int s_next;
  const int t = blockIdx.x;
  const int32_t* __restrict__ ko = key_offsets + (int64_t)t * tile_px;
  const int64_t out0 = (int64_t)grp_offsets[t] * 4;
  const int32_t beg = ko[0];
  // sorts cdt / cfx / cfy [o0, o0 + len), len <= kCanonHot, through LDS (block-uniform arguments)
  auto sort_span = [&](int64_t o0, int len) {
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {   // (padding sorts behind every event)
      s_dt[i] = i < len ? cdt[o0 + i] : __builtin_inff();
      s_fx[i] = i < len ? cfx[o0 + i] : 0.0f;
      s_fy[i] = i < len ? cfy[o0 + i] : 0.0f;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = threadIdx.x; i < n2 / 2; i += blockDim.x) {
          const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
          const bool ascending = (lo & size) == 0;
          const float da = s_dt[lo], xa = s_fx[lo], ya = s_fy[lo], db = s_dt[hi], xb = s_fx[hi], yb = s_fy[hi];
          const int qa = sort_key(da), qb = sort_key(db), kxa = sort_key(xa), kxb = sort_key(xb), kya = sort_key(ya), kyb = sort_key(yb);
          const bool a_after_b = qa > qb || (qa == qb && (kxa > kxb || (kxa == kxb && kya > kyb)));
          const bool b_after_a = qb > qa || (qa == qb && (kxb > kxa || (kxa == kxb && kyb > kya)));
          if (ascending ? a_after_b : b_after_a) {
            s_dt[lo] = db, s_fx[lo] = xb, s_fy[lo] = yb;
            s_dt[hi] = da, s_fx[hi] = xa, s_fy[hi] = ya;
          }
        }
        __syncthreads();
      }
    }
    // (an event whose dt is +inf or NaN ranks with / behind the padding: the len smallest keys are the events' unless such values
    // occur, and then they are equal to the padding's dt only for +inf -- whose fractions the padding's zeros would replace.  Events
    // carry finite dt = (t - t_ref) / period; the kernels treat NaN dt as an empty slot anyway.)
    for (int i = threadIdx.x; i < len; i += blockDim.x) cdt[o0 + i] = s_dt[i], cfx[o0 + i] = s_fx[i], cfy[o0 + i] = s_fy[i];
    __syncthreads();
  };
  int cursor = 0;
  while (true) {
    if (threadIdx.x == 0) s_next = 0x7fffffff;
    __syncthreads();
    for (int k = cursor + threadIdx.x; k < tile_px; k += blockDim.x) {
      if (ko[k + 1] - ko[k] > kCanonMax) {   // (a thread's candidates ascend: its first hot pixel is its smallest)
        atomicMin(&s_next, k);
        break;
      }
    }
    __syncthreads();
    const int k = s_next;
    __syncthreads();   // (everyone has read s_next before the next pass resets it)
    if (k == 0x7fffffff) break;
    cursor = k + 1;
    const int32_t kb = ko[k];
    const int len = ko[k + 1] - kb;
    const int64_t o0 = out0 + (kb - beg);
    if (len <= kCanonHot) {
      sort_span(o0, len);
    } else {
      constexpr int kChunk = kCanonHot / 2;
      const int m = (len + kChunk - 1) / kChunk;
      for (int phase = 0; phase < m; ++phase)
        for (int c = phase & 1; c + 1 < m; c += 2) sort_span(o0 + (int64_t)c * kChunk, min(2 * kChunk, len - c * kChunk));
    }
  }
}

// Adaptive work items: cut heavy tiles into parts of at most tau events, parts(t) = max(1, ceil(load(t) / tau)).
// A work item is one workgroup and a CU holds one of them at a time (LDS), so with F = the fixed work of an item
// (LDS clear, decode, slab store) expressed in events, a pass lasts about
//     max( tau + F ,  (N + items(tau) * F) / n_cu )        -- the longest item vs the average load of a CU.
// The first term grows with tau, the second falls: tau is the smallest value where the first reaches the second
// (and for which the parts fit the 2 x tiles budget) -- if that beats one part per tile by 20 % or more.  A uniform
// window keeps one part per tile, while a window whose events sit in a few tiles (a schlieren object in front of a
// static background) spreads those tiles over the otherwise idle CUs.  One workgroup; runs once per plan.
__global__ void __launch_bounds__(1024)
plan_parts_kernel(const int32_t* __restrict__ key_offsets, int n_tiles, int tile_px, int n_items, int n_cu, int fixed_events,
                  int32_t* __restrict__ part_table) {
  __shared__ long long red[1024 / kWave];
  __shared__ long long s_bcast;
  __shared__ int s_max;
  // the tiles' loads are read ~40 times (two searches, the ranking): kept in LDS (a window of more tiles reads them from memory --
  // with one 913 us launch per plan on a clustered 1280 x 720 window before, most of it the ranking's global loads)
  constexpr int kCached = 4096;
  __shared__ int s_load[kCached], s_parts[kCached];
  const bool cached = n_tiles <= kCached;
  auto load_global = [&](int t) { return key_offsets[(int64_t)(t + 1) * tile_px] - key_offsets[(int64_t)t * tile_px]; };
  if (cached) {
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) s_load[t] = load_global(t);
    __syncthreads();
  }
  auto load_of = [&](int t) { return cached ? s_load[t] : load_global(t); };
  auto items_at = [&](int tau) {  // block-uniform result
    long long cnt = 0;
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) cnt += max(1, (load_of(t) + tau - 1) / tau);
    cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) s_bcast = cnt;
    __syncthreads();
    cnt = s_bcast;
    __syncthreads();
    return cnt;
  };
  int lmax = 1;
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) lmax = max(lmax, load_of(t));
  if (threadIdx.x == 0) s_max = 1;
  __syncthreads();
  atomicMax(&s_max, lmax);
  __syncthreads();
  lmax = s_max;
  const long long total = (long long)key_offsets[(int64_t)n_tiles * tile_px] - key_offsets[0];
  int lo = 1, hi = lmax;
  // no tau can beat one part per tile by 20 % when even the average CU load does not (the common, even window):
  // skip the two searches (~35 workgroup-wide reductions)
  if ((long long)(lmax + fixed_events) * 80 * n_cu <= (total + (long long)n_tiles * fixed_events) * 100) lo = hi;
  // ... nor where the fixed work dominates even the fullest tile (a small window on few tiles: (lmax - tau) / (lmax + F) < 20 % for
  // every tau once lmax < F / 4) -- the searches took 26 us of a 100 k-event window's build
  if ((long long)lmax * 4 < fixed_events) lo = hi;
  while (lo < hi) {  // smallest tau whose parts fit the budget
    const int mid = lo + (hi - lo) / 2;
    if (items_at(mid) <= n_items) hi = mid;
    else lo = mid + 1;
  }
  hi = lmax;
  while (lo < hi) {  // smallest tau with  tau + F >= (N + items F) / n_cu
    const int mid = lo + (hi - lo) / 2;
    const long long items = items_at(mid);
    if ((long long)(mid + fixed_events) * n_cu >= total + items * fixed_events) hi = mid;
    else lo = mid + 1;
  }
  // Workgroups are dealt to the 8 XCDs round-robin and queue there, so a handful of extra items behind 256 equal
  // ones costs a whole extra round on some XCD (measured: 24.9 -> 35.2 us on a uniform window with two tiles halved).
  // Split only when the model promises a clear gain over one part per tile.
  if ((long long)(lo + fixed_events) * 100 > (long long)(lmax + fixed_events) * 80) lo = lmax;
  int32_t* part_off = part_table;
  int32_t* item_tile = part_table + n_tiles + 1;
  int32_t* item_part = item_tile + n_items;
  if (lo >= lmax) {  // one part per tile: the work items are the tiles, in order (no ranking needed)
    for (int t = threadIdx.x; t <= n_tiles; t += blockDim.x) part_off[t] = t;
    for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
      item_tile[i] = i < n_tiles ? i : -1;
      item_part[i] = 0;
    }
    return;
  }
  __shared__ int s_used;
  if (cached) {
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) s_parts[t] = max(1, (s_load[t] + lo - 1) / lo);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int off = 0;
    for (int t = 0; t < n_tiles; ++t) {
      part_off[t] = off;
      off += cached ? s_parts[t] : max(1, (load_of(t) + lo - 1) / lo);
    }
    part_off[n_tiles] = off;
    s_used = off;
  }
  __syncthreads();  // part_off is read below by other threads (global memory written by this workgroup: fence not needed
  __threadfence_block();  // across a barrier within one workgroup, kept explicit)
  // work items heaviest first (greedy longest-processing-time order of the dispatcher): rank by the events of a part, ties on the
  // slab index.  The parts of one tile carry the same load and consecutive slabs: they take consecutive ranks behind every part
  // of a heavier tile and of an equally heavy earlier one -- one pass over the tiles per tile.
  const int used = s_used;
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
    const int parts = cached ? s_parts[t] : part_off[t + 1] - part_off[t], load = load_of(t);
    const int mine = (load + parts - 1) / parts;
    int rank = 0;
    for (int u = 0; u < n_tiles; ++u) {
      const int pu = cached ? s_parts[u] : part_off[u + 1] - part_off[u], lu = (load_of(u) + pu - 1) / pu;
      if (lu > mine || (lu == mine && u < t)) rank += pu;
    }
    for (int k = 0; k < parts; ++k) {
      item_tile[rank + k] = t;
      item_part[rank + k] = k;
    }
  }
  for (int i = used + threadIdx.x; i < n_items; i += blockDim.x) {
    item_tile[i] = -1;
    item_part[i] = 0;
  }
}

// The build's one read-back as ONE small kernel (ebos_plan_facts): (outside the image, fractional sources, work items in use,
// events of the fullest tile) side by side -- gathered with torch ops they were four launches and a copy (~20 us of a 0.3 ms build)
__global__ void __launch_bounds__(256) plan_facts_kernel(const int32_t* __restrict__ key_offsets, int n_tiles, int tile_px,
                                                         const int32_t* __restrict__ counts, const int32_t* __restrict__ part_table,
                                                         int32_t* __restrict__ facts) {
  __shared__ int s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  int m = 0;
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x)
    m = max(m, key_offsets[(int64_t)(t + 1) * tile_px] - key_offsets[(int64_t)t * tile_px]);
  atomicMax(&s_max, m);
  __syncthreads();
  if (threadIdx.x == 0) {
    facts[0] = counts ? counts[0] : 0;
    facts[1] = counts ? counts[1] : 0;
    facts[2] = part_table ? part_table[n_tiles] : 0;
    facts[3] = s_max;
  }
}

template <typename T>
int events_to_soa_impl(const T* events, const T* tminmax, int ref_mode, double ref_fraction, int normalize_t,
                       int64_t n, float* x, float* y, float* dt, float* p, ebos_stream_t stream) {
  EBOS_REQUIRE(tminmax != nullptr, "ebos_events_to_soa: tminmax is NULL");
  EBOS_REQUIRE((events && x && y && dt && p) || n == 0, "ebos_events_to_soa: NULL buffer");
  EBOS_REQUIRE(ref_mode >= 0 && ref_mode <= 3 && n >= 0, "ebos_events_to_soa: bad ref_mode/n");
  if (n == 0) return EBOS_OK;
  events_to_soa_kernel<T><<<dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream)>>>(
      events, tminmax, ref_mode, ref_fraction, normalize_t, n, x, y, dt, p);
  EBOS_CHECK_LAUNCH("ebos_events_to_soa");
  return EBOS_OK;
}

inline int64_t scan_blocks(int64_t n_keys) { return (n_keys + kScanTile - 1) / kScanTile; }

}  // namespace
}  // namespace ebos

extern "C" {

int ebos_events_to_soa_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, int64_t n, float* x, float* y, float* dt, float* p,
                           ebos_stream_t stream) {
  return ebos::events_to_soa_impl<float>(events, tminmax, ref_mode, ref_fraction, normalize_t, n, x, y, dt, p, stream);
}
int ebos_events_to_soa_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, int64_t n, float* x, float* y, float* dt, float* p,
                           ebos_stream_t stream) {
  return ebos::events_to_soa_impl<double>(events, tminmax, ref_mode, ref_fraction, normalize_t, n, x, y, dt, p, stream);
}

int ebos_raw_time_range(const void* t, int t_bytes, int64_t n, double ticks_per_second, int64_t* scratch_ticks,
                        double* tminmax, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE((t && scratch_ticks && tminmax) && n >= 1, "ebos_raw_time_range: NULL buffer or empty window");
  EBOS_REQUIRE((t_bytes == 4 || t_bytes == 8) && ticks_per_second > 0.0, "ebos_raw_time_range: bad time format");
  hipStream_t s = as_stream(stream);
  long long* ticks = reinterpret_cast<long long*>(scratch_ticks);
  raw_time_range_init_kernel<<<dim3(1), dim3(1), 0, s>>>(ticks);
  const dim3 grid(stream_grid(n, 256, 1024));
  if (t_bytes == 4) raw_time_range_kernel<int32_t><<<grid, dim3(256), 0, s>>>(static_cast<const int32_t*>(t), n, ticks);
  else raw_time_range_kernel<int64_t><<<grid, dim3(256), 0, s>>>(static_cast<const int64_t*>(t), n, ticks);
  raw_time_range_seconds_kernel<<<dim3(1), dim3(1), 0, s>>>(ticks, ticks_per_second, tminmax);
  EBOS_CHECK_LAUNCH("ebos_raw_time_range");
  return EBOS_OK;
}

int ebos_raw_events_to_soa(const int16_t* col, const int16_t* row, const void* t, int t_bytes, const uint8_t* pol,
                           double ticks_per_second, const double* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, int64_t n, float* x, float* y, float* dt, float* p, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(tminmax != nullptr, "ebos_raw_events_to_soa: tminmax is NULL");
  EBOS_REQUIRE((col && row && t && pol && x && y && dt && p) || n == 0, "ebos_raw_events_to_soa: NULL buffer");
  EBOS_REQUIRE((t_bytes == 4 || t_bytes == 8) && ticks_per_second > 0.0, "ebos_raw_events_to_soa: bad time format");
  EBOS_REQUIRE(ref_mode >= 0 && ref_mode <= 3 && n >= 0, "ebos_raw_events_to_soa: bad ref_mode/n");
  if (n == 0) return EBOS_OK;
  const dim3 grid(stream_grid(n, 256));
  hipStream_t s = as_stream(stream);
  if (t_bytes == 4)
    raw_to_soa_kernel<int32_t><<<grid, dim3(256), 0, s>>>(col, row, static_cast<const int32_t*>(t), pol, ticks_per_second,
                                                        tminmax, ref_mode, ref_fraction, normalize_t, n, x, y, dt, p);
  else
    raw_to_soa_kernel<int64_t><<<grid, dim3(256), 0, s>>>(col, row, static_cast<const int64_t*>(t), pol, ticks_per_second,
                                                        tminmax, ref_mode, ref_fraction, normalize_t, n, x, y, dt, p);
  EBOS_CHECK_LAUNCH("ebos_raw_events_to_soa");
  return EBOS_OK;
}

int ebos_plan_parts(const int32_t* key_offsets, int H, int W, int tile_h, int tile_w, int n_cu, int fixed_events,
                    int32_t* part_table, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(key_offsets && part_table, "ebos_plan_parts: NULL buffer");
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && n_cu >= 1 && fixed_events >= 0, "ebos_plan_parts: bad sizes");
  const int n_tiles = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  plan_parts_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(key_offsets, n_tiles, tile_h * tile_w, 2 * n_tiles, n_cu,
                                                                  fixed_events, part_table);
  EBOS_CHECK_LAUNCH("ebos_plan_parts");
  return EBOS_OK;
}

int ebos_plan_facts(const int32_t* key_offsets, int H, int W, int tile_h, int tile_w, const int32_t* counts,
                    const int32_t* part_table, int32_t* facts, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(key_offsets && facts, "ebos_plan_facts: NULL buffer");
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0, "ebos_plan_facts: bad sizes");
  const int n_tiles = ((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w);
  plan_facts_kernel<<<dim3(1), dim3(256), 0, as_stream(stream)>>>(key_offsets, n_tiles, tile_h * tile_w, counts, part_table, facts);
  EBOS_CHECK_LAUNCH("ebos_plan_facts");
  return EBOS_OK;
}

size_t ebos_bin_scratch_bytes(int64_t n_keys) {
  if (n_keys < 0) return 0;
  // cursor[n_keys] + block_sums[scan_blocks], int32 each, 256-byte aligned sections
  const size_t a = ((size_t)n_keys * 4 + 255) & ~(size_t)255;
  const size_t b = ((size_t)ebos::scan_blocks(n_keys) * 4 + 255) & ~(size_t)255;
  return a + b + 256;
}

size_t ebos_bin_scratch_bytes_events(int64_t n, int H, int W, int tile_h, int tile_w) {
  if (n < 0 || H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0) return 0;
  const int64_t n_keys = (int64_t)((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w) * tile_h * tile_w;
  // + one 32-byte record and one 4-byte rank per event
  return ebos_bin_scratch_bytes(n_keys) + (size_t)(n > 0 ? n : 1) * (sizeof(ebos::EventRecord) + 4) + 512;
}

int ebos_bin_events_f32(const float* x, const float* y, const float* dt, const float* p, int64_t n, int H, int W,
                        int tile_h, int tile_w, float* xs, float* ys, float* dts, float* ps, int32_t* perm,
                        int32_t* key_offsets, int32_t* oob_count, int32_t* frac_count, void* scratch,
                        size_t scratch_bytes, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && n >= 0 && n < (int64_t)1 << 31,
               "ebos_bin_events: bad sizes n=%lld H=%d W=%d tile=%dx%d", (long long)n, H, W, tile_h, tile_w);
  EBOS_REQUIRE(key_offsets && scratch, "ebos_bin_events: NULL key_offsets/scratch");
  EBOS_REQUIRE((x && y && dt && xs && ys && dts) || n == 0, "ebos_bin_events: NULL event buffer");
  const int tiles_y = (H + tile_h - 1) / tile_h, tiles_x = (W + tile_w - 1) / tile_w;
  const int64_t n_keys = (int64_t)tiles_y * tiles_x * tile_h * tile_w;
  if (scratch_bytes < ebos_bin_scratch_bytes(n_keys)) {
    set_error("ebos_bin_events: scratch too small (%zu < %zu)", scratch_bytes, ebos_bin_scratch_bytes(n_keys));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  int32_t* cursor = reinterpret_cast<int32_t*>(scratch);
  const size_t a = ((size_t)n_keys * 4 + 255) & ~(size_t)255;
  int32_t* block_sums = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + a);
  const int nblk = (int)scan_blocks(n_keys);

  if (hipMemsetAsync(key_offsets, 0, (size_t)(n_keys + 1) * 4, s) != hipSuccess ||
      hipMemsetAsync(cursor, 0, (size_t)n_keys * 4, s) != hipSuccess) {
    set_error("ebos_bin_events: hipMemsetAsync failed");
    return EBOS_ERR_LAUNCH;
  }
  // with the larger scratch of ebos_bin_scratch_bytes_events: ranks from the histogram pass, 32-byte records scattered
  // without atomics, then unpacked to the SoA arrays
  const size_t rec_off = (ebos_bin_scratch_bytes(n_keys) + 255) & ~(size_t)255;
  const bool records = scratch_bytes >= ebos_bin_scratch_bytes_events(n, H, W, tile_h, tile_w);
  EventRecord* rec = reinterpret_cast<EventRecord*>(reinterpret_cast<char*>(scratch) + rec_off);
  int32_t* rank = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + rec_off +
                                             (((size_t)(n > 0 ? n : 1) * sizeof(EventRecord) + 255) & ~(size_t)255));
  if (n > 0)
    bin_hist_kernel<<<dim3(stream_grid(n, 256)), dim3(256), 0, s>>>(x, y, n, H, W, tile_h, tile_w, tiles_x, key_offsets,
                                                                   oob_count, records ? rank : nullptr);
  scan_tiles_kernel<<<dim3(nblk), dim3(kScanBlock), 0, s>>>(key_offsets, n_keys, block_sums);
  scan_block_sums_kernel<<<dim3(1), dim3(kScanBlock), 0, s>>>(block_sums, nblk, key_offsets + n_keys);
  scan_add_offsets_kernel<<<dim3(nblk), dim3(kScanBlock), 0, s>>>(key_offsets, n_keys, block_sums);
  if (n > 0 && records) {
    bin_scatter_records_kernel<<<dim3(stream_grid(n, 256)), dim3(256), 0, s>>>(x, y, dt, p, n, H, W, tile_h, tile_w, tiles_x,
                                                                              key_offsets, rank, rec, frac_count);
    unpack_records_kernel<<<dim3(stream_grid(n, 256)), dim3(256), 0, s>>>(rec, key_offsets + n_keys, xs, ys, dts, ps, perm);
  } else if (n > 0) {
    bin_scatter_kernel<<<dim3(stream_grid(n, 256)), dim3(256), 0, s>>>(x, y, dt, p, n, H, W, tile_h, tile_w, tiles_x,
                                                                      key_offsets, cursor, xs, ys, dts, ps, perm, frac_count);
  }
  EBOS_CHECK_LAUNCH("ebos_bin_events");
  return EBOS_OK;
}

int ebos_plan_compact_f32(const float* xs, const float* ys, const float* dts, const int32_t* key_offsets, int64_t n, int H,
                          int W, int tile_h, int tile_w, int32_t* grp_offsets, uint16_t* cpix, float* cdt,
                          int64_t capacity_slots, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(key_offsets && grp_offsets && cpix && cdt, "ebos_plan_compact: NULL buffer");
  EBOS_REQUIRE((xs && ys && dts) || n == 0, "ebos_plan_compact: NULL event buffer");
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && tile_h <= 256 && tile_w <= 256 && n >= 0,
               "ebos_plan_compact: bad sizes (tiles up to 256 x 256)");
  const int tiles_y = (H + tile_h - 1) / tile_h, tiles_x = (W + tile_w - 1) / tile_w;
  const int n_tiles = tiles_y * tiles_x;
  if (capacity_slots < n + 3 * (int64_t)n_tiles + 4) {
    set_error("ebos_plan_compact: capacity %lld < n + 3 tiles + 4 = %lld", (long long)capacity_slots,
              (long long)(n + 3 * (int64_t)n_tiles + 4));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  compact_offsets_kernel<<<dim3(1), dim3(256), 0, s>>>(key_offsets, tile_h * tile_w, n_tiles, grp_offsets);
  compact_fill_kernel<<<dim3(n_tiles), dim3(256), 0, s>>>(xs, ys, dts, key_offsets, tile_h, tile_w, tiles_x, grp_offsets, cpix,
                                                          cdt);
  EBOS_CHECK_LAUNCH("ebos_plan_compact");
  return EBOS_OK;
}

int ebos_plan_compact_frac_f32(const float* xs, const float* ys, const float* dts, const int32_t* key_offsets, int64_t n, int H,
                               int W, int tile_h, int tile_w, int32_t* grp_offsets, uint16_t* cpix, float* cdt, float* cfx,
                               float* cfy, int64_t capacity_slots, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(key_offsets && grp_offsets && cpix && cdt && cfx && cfy, "ebos_plan_compact_frac: NULL buffer");
  EBOS_REQUIRE((xs && ys && dts) || n == 0, "ebos_plan_compact_frac: NULL event buffer");
  EBOS_REQUIRE(H > 0 && W > 0 && tile_h > 0 && tile_w > 0 && tile_h <= 256 && tile_w <= 256 && n >= 0,
               "ebos_plan_compact_frac: bad sizes (tiles up to 256 x 256)");
  const int tiles_y = (H + tile_h - 1) / tile_h, tiles_x = (W + tile_w - 1) / tile_w;
  const int n_tiles = tiles_y * tiles_x;
  if (capacity_slots < n + 3 * (int64_t)n_tiles + 4) {
    set_error("ebos_plan_compact_frac: capacity %lld < n + 3 tiles + 4 = %lld", (long long)capacity_slots,
              (long long)(n + 3 * (int64_t)n_tiles + 4));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  compact_offsets_kernel<<<dim3(1), dim3(256), 0, s>>>(key_offsets, tile_h * tile_w, n_tiles, grp_offsets);
  // (the rank loop makes an event several loads: four workgroups per tile)
  compact_fill_kernel<<<dim3(n_tiles, 4), dim3(256), 0, s>>>(xs, ys, dts, key_offsets, tile_h, tile_w, tiles_x, grp_offsets, cpix,
                                                             cdt, cfx, cfy);
  compact_canon_hot_kernel<<<dim3(n_tiles), dim3(1024), 0, s>>>(key_offsets, tile_h * tile_w, grp_offsets, cdt, cfx, cfy);
  EBOS_CHECK_LAUNCH("ebos_plan_compact_frac");
  return EBOS_OK;
}

}  // extern "C"
