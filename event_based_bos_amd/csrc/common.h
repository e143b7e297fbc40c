// common.h -- shared helpers for the gfx950 kernels of libebos_hip.so (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "ebos_hip.h"

namespace ebos {

constexpr int kWave = 64;  // CDNA wavefront width

void set_error(const char* fmt, ...);
// false unless ebos_profile_start[_kernel]() is active for this launcher (`which`: ebos_profile_kernel)
bool profile_next_pair(hipEvent_t* start, hipEvent_t* stop, int which = 0);

#define EBOS_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::ebos::set_error(__VA_ARGS__);           \
      return EBOS_ERR_INVALID_ARG;              \
    }                                           \
  } while (0)

#define EBOS_CHECK_LAUNCH(what)                                                   \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ::ebos::set_error("%s: %s", what, hipGetErrorString(e__));                  \
      return EBOS_ERR_LAUNCH;                                                     \
    }                                                                             \
  } while (0)

inline hipStream_t as_stream(ebos_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// grid size for a grid-stride streaming kernel: enough workgroups to fill 256 CUs x 8, no more
inline int stream_grid(int64_t items, int block, int max_blocks = 256 * 8) {
  int64_t g = (items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return static_cast<int>(g);
}

// ---- wavefront reductions (64 lanes) -------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;  // valid in lane 0
}
// float / double sums through the DPP cross-lane path of the VALU (row_shr 1, 2, 4, 8, then row_bcast 15 and 31: the total ends up in
// lane 63 and is read back with v_readlane -> valid in EVERY lane) instead of six ds_bpermute round trips through the LDS pipe:
// a wave sum costs ~10 VALU instructions' worth of latency, not ~6 x 100 cycles.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_or_zero(unsigned long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(v & 0xffffffffull), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v += dpp_or_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_or_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_or_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_or_zero<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row of 16 holds the row's sum
  v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_or_zero<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return v;
}
template <>
__device__ __forceinline__ unsigned long long wave_sum<unsigned long long>(unsigned long long v) {  // modulo 2^64
  v = wave_sum_dpp(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
template <>
__device__ __forceinline__ float wave_sum<float>(float v) {
  v = wave_sum_dpp(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <>
__device__ __forceinline__ double wave_sum<double>(double v) {
  const long long b = __builtin_bit_cast(long long, wave_sum_dpp(v));
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// maximum of NON-NEGATIVE floats over the wave through the DPP path (lanes shifted in from outside a row read 0); valid in every lane
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = fmaxf(v, dpp_or_zero<0x111, 0xf>(v));  // row_shr:1
  v = fmaxf(v, dpp_or_zero<0x112, 0xf>(v));  // row_shr:2
  v = fmaxf(v, dpp_or_zero<0x114, 0xf>(v));  // row_shr:4
  v = fmaxf(v, dpp_or_zero<0x118, 0xf>(v));  // row_shr:8   -> lane 15 of every row of 16 holds the row's maximum
  v = fmaxf(v, dpp_or_zero<0x142, 0xa>(v));  // row_bcast:15 into rows 1 and 3
  v = fmaxf(v, dpp_or_zero<0x143, 0xc>(v));  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    T o = __shfl_down(v, off, kWave);
    v = o < v ? o : v;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    T o = __shfl_down(v, off, kWave);
    v = o > v ? o : v;
  }
  return v;
}

// block-wide sum through LDS; result valid in thread 0.  `red` holds >= blockDim.x / 64 items.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  v = wave_sum(v);
  if (lane == 0) red[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + kWave - 1) / kWave;
  T r = (threadIdx.x < nw) ? red[threadIdx.x] : T(0);
  if (wid == 0) r = wave_sum(r);
  __syncthreads();
  return r;
}

// two block-wide sums at once, in the order block_sum uses for each (bit-identical results): the two reductions interleave
// and the pair costs two barriers instead of four.  `red` holds >= 2 * blockDim.x / 64 items.  Results valid in thread 0.
template <typename T>
__device__ __forceinline__ void block_sum2(T& a, T& b, T* red) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  const int nw = (blockDim.x + kWave - 1) / kWave;
  a = wave_sum(a);
  b = wave_sum(b);
  if (lane == 0) {
    red[wid] = a;
    red[nw + wid] = b;
  }
  __syncthreads();
  T ra = (threadIdx.x < nw) ? red[threadIdx.x] : T(0), rb = (threadIdx.x < nw) ? red[nw + threadIdx.x] : T(0);
  if (wid == 0) {
    ra = wave_sum(ra);
    rb = wave_sum(rb);
  }
  __syncthreads();
  a = ra;
  b = rb;
}

// hardware float atomics (no CAS loop): global_atomic_add_f32 / _f64, ds_add_f32 / _f64
__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// reference time + period exactly as src/warp.py:245-253,283-287 evaluates them in element type T
template <typename T>
struct TimeBase {
  T ref;     // t_ref
  T period;  // max(dt) - min(dt) = (tmax - ref) - (tmin - ref), each rounded in T
};
template <typename T>
__device__ __forceinline__ TimeBase<T> time_base(const T* tminmax, int ref_mode, double ref_fraction) {
#pragma clang fp contract(off)
  const T tmin = tminmax[0], tmax = tminmax[1];
  TimeBase<T> tb;
  if (ref_mode == EBOS_REF_TIMEBASE) {  // caller supplies (t_ref, period)
    tb.ref = tmin;
    tb.period = tmax;
    return tb;
  }
  if (ref_mode == EBOS_REF_FIRST) {
    tb.ref = tmin;
  } else if (ref_mode == EBOS_REF_LAST) {
    tb.ref = tmax;
  } else {
    const T per = tmax - tmin;                       // src/warp.py:246
    tb.ref = tmin + per * static_cast<T>(ref_fraction);  // :247  (python float * tensor -> T)
  }
  tb.period = (tmax - tb.ref) - (tmin - tb.ref);     // :286 on dt = t - ref
  return tb;
}

// Bilinear footprint of one warped event in the padded image.
template <typename T>
struct Footprint {
  int R, C;     // top-left tap in padded image coordinates
  T fr, fc;     // fractional offsets (may be slightly negative: floor(x + eps))
  bool finite;  // false: NaN/Inf/astronomically large coordinate -> no tap is in the image
};
template <typename T>
__device__ __forceinline__ Footprint<T> footprint(T xw, T yw, T eps, int pad_h, int pad_w) {
  Footprint<T> f;
  const T r0 = floor(xw + eps);
  const T c0 = floor(yw + eps);
  f.fr = xw - r0;
  f.fc = yw - c0;
  const T lim = T(1 << 30);
  f.finite = (r0 > -lim) && (r0 < lim) && (c0 > -lim) && (c0 < lim);
  f.R = f.finite ? static_cast<int>(r0) + pad_h : -4;
  f.C = f.finite ? static_cast<int>(c0) + pad_w : -4;
  return f;
}

}  // namespace ebos
