// warp_kernels.hip -- API-parity (array-of-structs) warp kernels for gfx950.
//
// These kernels back Warp.warp_event for tensors that live on the GPU: one event = one
// [x, y, t, p] row = one 16-byte (f32) / 32-byte (f64) vector load per lane, fully coalesced.
// Arithmetic follows the reference's torch branch operation by operation in the element type
// (src/warp.py:245-253, 283-287, 330-342, 364-383 under /root/reference) with FMA contraction
// disabled, so outputs are bit-identical to the reference on the same dtype.
#include "common.h"

namespace ebos {
namespace {

template <typename T>
struct Vec4;
template <>
struct Vec4<float> {
  using type = float4;
};
template <>
struct Vec4<double> {
  using type = double4;
};

template <typename T>
__device__ __forceinline__ typename Vec4<T>::type load_event(const T* base, int64_t i) {
  return reinterpret_cast<const typename Vec4<T>::type*>(base)[i];
}
template <typename T>
__device__ __forceinline__ void store_event(T* base, int64_t i, typename Vec4<T>::type v) {
  reinterpret_cast<typename Vec4<T>::type*>(base)[i] = v;
}

template <typename T>
__device__ __forceinline__ T limit_hi();
template <>
__device__ __forceinline__ float limit_hi<float>() { return __builtin_huge_valf(); }
template <>
__device__ __forceinline__ double limit_hi<double>() { return __builtin_huge_val(); }

// ---- A2: per-batch-row (min t, max t) ------------------------------------------------------
template <typename T>
__global__ void time_range_init_kernel(T* tminmax, int64_t b) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < b) {
    tminmax[2 * i] = limit_hi<T>();
    tminmax[2 * i + 1] = -limit_hi<T>();
  }
}

template <typename T>
__global__ void __launch_bounds__(256) time_range_kernel(const T* __restrict__ events, int64_t n, T* tminmax) {
  const int64_t row = blockIdx.y;
  const T* ev = events + row * n * 4;
  T lo = limit_hi<T>(), hi = -limit_hi<T>();
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T t = ev[4 * i + 2];
    lo = t < lo ? t : lo;
    hi = t > hi ? t : hi;
  }
  __shared__ T s_lo[4], s_hi[4];
  lo = wave_min(lo);
  hi = wave_max(hi);
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  if (lane == 0) {
    s_lo[wid] = lo;
    s_hi[wid] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      lo = s_lo[k] < lo ? s_lo[k] : lo;
      hi = s_hi[k] > hi ? s_hi[k] : hi;
    }
    atomicMin(&tminmax[2 * row], lo);  // one atomic per workgroup
    atomicMax(&tminmax[2 * row + 1], hi);
  }
}

template <typename T>
__device__ __forceinline__ T event_dt(T t, const TimeBase<T>& tb, int normalize_t) {
#pragma clang fp contract(off)
  T dt = t - tb.ref;                    // src/warp.py:283
  if (normalize_t) dt = dt / tb.period;  // :284-287
  return dt;
}

// source-pixel linear index of src/warp.py:334 (trunc toward zero, flattened bounds as torch.gather)
template <typename T>
__device__ __forceinline__ bool source_index(T x, T y, int row_stride, int64_t hw, int64_t* lin) {
  const T lim = T(1e15);
  if (!(x > -lim && x < lim && y > -lim && y < lim)) return false;  // NaN / Inf
  const int64_t l = static_cast<int64_t>(x) * row_stride + static_cast<int64_t>(y);
  *lin = l;
  return l >= 0 && l < hw;
}

// ---- A3: dense-flow warp, forward ---------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
warp_dense_kernel(const T* __restrict__ events, const T* __restrict__ flow, const T* __restrict__ tminmax,
                  int ref_mode, double ref_fraction, int normalize_t, int64_t n, int H, int W, int row_stride,
                  T* __restrict__ warped, int32_t* oob_count) {
#pragma clang fp contract(off)
  const int64_t row = blockIdx.y;
  const int64_t hw = (int64_t)H * W;
  const T* ev = events + row * n * 4;
  const T* f0 = flow + row * 2 * hw;
  const T* f1 = f0 + hw;
  T* out = warped + row * n * 4;
  const TimeBase<T> tb = time_base(tminmax + 2 * row, ref_mode, ref_fraction);
  int bad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    auto e = load_event(ev, i);
    const T dt = event_dt(e.z, tb, normalize_t);
    int64_t lin;
    if (source_index(e.x, e.y, row_stride, hw, &lin)) {
      const T u = f0[lin], v = f1[lin];
      const T du = dt * u, dv = dt * v;  // separate roundings: mul, then sub (src/warp.py:335-336)
      e.x = e.x - du;
      e.y = e.y - dv;
    } else {
      ++bad;
    }
    e.z = dt;
    store_event(out, i, e);
  }
  if (oob_count != nullptr && bad) atomicAdd(oob_count, bad);
}

// ---- A3 backward: d_flow[c][src] += -dt * d_warped[c] --------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
warp_dense_bwd_kernel(const T* __restrict__ events, const T* __restrict__ tminmax, int ref_mode,
                      double ref_fraction, int normalize_t, const T* __restrict__ d_warped, int64_t n, int H, int W,
                      int row_stride, T* d_flow) {
  const int64_t row = blockIdx.y;
  const int64_t hw = (int64_t)H * W;
  const T* ev = events + row * n * 4;
  const T* dw = d_warped + row * n * 4;
  T* g0 = d_flow + row * 2 * hw;
  T* g1 = g0 + hw;
  const TimeBase<T> tb = time_base(tminmax + 2 * row, ref_mode, ref_fraction);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const auto e = load_event(ev, i);
    const auto g = load_event(dw, i);
    const T dt = event_dt(e.z, tb, normalize_t);
    int64_t lin;
    if (source_index(e.x, e.y, row_stride, hw, &lin)) {
      atomic_add(&g0[lin], -dt * g.x);
      atomic_add(&g1[lin], -dt * g.y);
    }
  }
}

// ---- A4: 2-DoF translation warp ------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
warp_2dof_kernel(const T* __restrict__ events, const T* __restrict__ theta, const T* __restrict__ tminmax,
                 int ref_mode, double ref_fraction, int normalize_t, const T* __restrict__ time_period, int64_t n,
                 T* __restrict__ warped) {
#pragma clang fp contract(off)
  TimeBase<T> tb = time_base(tminmax, ref_mode, ref_fraction);
  if (time_period != nullptr) tb.period = time_period[0];  // src/warp.py:285-287
  const T th0 = theta[0], th1 = theta[1];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    auto e = load_event(events, i);
    const T dt = event_dt(e.z, tb, normalize_t);
    const T dx = dt * th0, dy = dt * th1;  // src/warp.py:368-369
    e.x = e.x + dx;                        // :373-375 (plus sign)
    e.y = e.y + dy;
    e.z = dt;
    store_event(warped, i, e);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
warp_2dof_bwd_kernel(const T* __restrict__ events, const T* __restrict__ tminmax, int ref_mode, double ref_fraction,
                     int normalize_t, const T* __restrict__ time_period, const T* __restrict__ d_warped, int64_t n,
                     T* d_theta) {
  TimeBase<T> tb = time_base(tminmax, ref_mode, ref_fraction);
  if (time_period != nullptr) tb.period = time_period[0];
  T a0 = 0, a1 = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const T t = events[4 * i + 2];
    const auto g = load_event(d_warped, i);
    const T dt = event_dt(t, tb, normalize_t);
    a0 += dt * g.x;
    a1 += dt * g.y;
  }
  __shared__ T red[4];
  a0 = block_sum(a0, red);
  a1 = block_sum(a1, red);
  if (threadIdx.x == 0) {
    atomic_add(&d_theta[0], a0);
    atomic_add(&d_theta[1], a1);
  }
}

bool ref_mode_ok(int m) { return m >= EBOS_REF_FIRST && m <= EBOS_REF_TIMEBASE; }

template <typename T>
int time_range_impl(const T* events, int64_t b, int64_t n, T* tminmax, ebos_stream_t stream) {
  EBOS_REQUIRE(events != nullptr || n == 0, "ebos_time_range: events is NULL");
  EBOS_REQUIRE(tminmax != nullptr, "ebos_time_range: tminmax is NULL");
  EBOS_REQUIRE(b >= 1 && n >= 0 && b <= 65535, "ebos_time_range: bad sizes b=%lld n=%lld", (long long)b, (long long)n);
  hipStream_t s = as_stream(stream);
  time_range_init_kernel<T><<<dim3((unsigned)((b + 255) / 256)), dim3(256), 0, s>>>(tminmax, b);
  if (n > 0) {
    dim3 grid(stream_grid(n, 256, 1024), (unsigned)b);
    time_range_kernel<T><<<grid, dim3(256), 0, s>>>(events, n, tminmax);
  }
  EBOS_CHECK_LAUNCH("ebos_time_range");
  return EBOS_OK;
}

template <typename T>
int warp_dense_impl(const T* events, const T* flow, const T* tminmax, int ref_mode, double ref_fraction,
                    int normalize_t, int64_t b, int64_t n, int H, int W, int row_stride, T* warped,
                    int32_t* oob_count, ebos_stream_t stream) {
  EBOS_REQUIRE(flow && tminmax, "ebos_warp_dense: NULL flow/tminmax");
  EBOS_REQUIRE((events && warped) || n == 0, "ebos_warp_dense: NULL events/warped");
  EBOS_REQUIRE(ref_mode_ok(ref_mode), "ebos_warp_dense: bad ref_mode %d", ref_mode);
  EBOS_REQUIRE(b >= 1 && b <= 65535 && n >= 0 && H > 0 && W > 0 && row_stride > 0,
               "ebos_warp_dense: bad sizes b=%lld n=%lld H=%d W=%d stride=%d", (long long)b, (long long)n, H, W, row_stride);
  if (n == 0) return EBOS_OK;
  dim3 grid(stream_grid(n, 256), (unsigned)b);
  warp_dense_kernel<T><<<grid, dim3(256), 0, as_stream(stream)>>>(events, flow, tminmax, ref_mode, ref_fraction,
                                                                  normalize_t, n, H, W, row_stride, warped, oob_count);
  EBOS_CHECK_LAUNCH("ebos_warp_dense");
  return EBOS_OK;
}

template <typename T>
int warp_dense_bwd_impl(const T* events, const T* tminmax, int ref_mode, double ref_fraction, int normalize_t,
                        const T* d_warped, int64_t b, int64_t n, int H, int W, int row_stride, T* d_flow,
                        ebos_stream_t stream) {
  EBOS_REQUIRE(tminmax && d_flow, "ebos_warp_dense_bwd: NULL tminmax/d_flow");
  EBOS_REQUIRE((events && d_warped) || n == 0, "ebos_warp_dense_bwd: NULL events/d_warped");
  EBOS_REQUIRE(ref_mode_ok(ref_mode), "ebos_warp_dense_bwd: bad ref_mode %d", ref_mode);
  EBOS_REQUIRE(b >= 1 && b <= 65535 && n >= 0 && H > 0 && W > 0 && row_stride > 0, "ebos_warp_dense_bwd: bad sizes");
  if (n == 0) return EBOS_OK;
  dim3 grid(stream_grid(n, 256), (unsigned)b);
  warp_dense_bwd_kernel<T><<<grid, dim3(256), 0, as_stream(stream)>>>(events, tminmax, ref_mode, ref_fraction,
                                                                      normalize_t, d_warped, n, H, W, row_stride, d_flow);
  EBOS_CHECK_LAUNCH("ebos_warp_dense_bwd");
  return EBOS_OK;
}

template <typename T>
int warp_2dof_impl(const T* events, const T* theta, const T* tminmax, int ref_mode, double ref_fraction,
                   int normalize_t, const T* time_period, int64_t n, T* warped, ebos_stream_t stream) {
  EBOS_REQUIRE(theta && tminmax, "ebos_warp_2dof: NULL theta/tminmax");
  EBOS_REQUIRE((events && warped) || n == 0, "ebos_warp_2dof: NULL events/warped");
  EBOS_REQUIRE(ref_mode_ok(ref_mode) && n >= 0, "ebos_warp_2dof: bad ref_mode/n");
  if (n == 0) return EBOS_OK;
  warp_2dof_kernel<T><<<dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream)>>>(
      events, theta, tminmax, ref_mode, ref_fraction, normalize_t, time_period, n, warped);
  EBOS_CHECK_LAUNCH("ebos_warp_2dof");
  return EBOS_OK;
}

template <typename T>
int warp_2dof_bwd_impl(const T* events, const T* tminmax, int ref_mode, double ref_fraction, int normalize_t,
                       const T* time_period, const T* d_warped, int64_t n, T* d_theta, ebos_stream_t stream) {
  EBOS_REQUIRE(tminmax && d_theta, "ebos_warp_2dof_bwd: NULL tminmax/d_theta");
  EBOS_REQUIRE((events && d_warped) || n == 0, "ebos_warp_2dof_bwd: NULL events/d_warped");
  EBOS_REQUIRE(ref_mode_ok(ref_mode) && n >= 0, "ebos_warp_2dof_bwd: bad ref_mode/n");
  if (n == 0) return EBOS_OK;
  warp_2dof_bwd_kernel<T><<<dim3(stream_grid(n, 256, 1024)), dim3(256), 0, as_stream(stream)>>>(
      events, tminmax, ref_mode, ref_fraction, normalize_t, time_period, d_warped, n, d_theta);
  EBOS_CHECK_LAUNCH("ebos_warp_2dof_bwd");
  return EBOS_OK;
}

}  // namespace
}  // namespace ebos

extern "C" {

int ebos_time_range_f32(const float* events, int64_t b, int64_t n, float* tminmax, ebos_stream_t stream) {
  return ebos::time_range_impl<float>(events, b, n, tminmax, stream);
}
int ebos_time_range_f64(const double* events, int64_t b, int64_t n, double* tminmax, ebos_stream_t stream) {
  return ebos::time_range_impl<double>(events, b, n, tminmax, stream);
}

int ebos_warp_dense_f32(const float* events, const float* flow, const float* tminmax, int ref_mode,
                        double ref_fraction, int normalize_t, int64_t b, int64_t n, int H, int W, int row_stride,
                        float* warped, int32_t* oob_count, ebos_stream_t stream) {
  return ebos::warp_dense_impl<float>(events, flow, tminmax, ref_mode, ref_fraction, normalize_t, b, n, H, W,
                                      row_stride, warped, oob_count, stream);
}
int ebos_warp_dense_f64(const double* events, const double* flow, const double* tminmax, int ref_mode,
                        double ref_fraction, int normalize_t, int64_t b, int64_t n, int H, int W, int row_stride,
                        double* warped, int32_t* oob_count, ebos_stream_t stream) {
  return ebos::warp_dense_impl<double>(events, flow, tminmax, ref_mode, ref_fraction, normalize_t, b, n, H, W,
                                       row_stride, warped, oob_count, stream);
}
int ebos_warp_dense_bwd_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                            int normalize_t, const float* d_warped, int64_t b, int64_t n, int H, int W,
                            int row_stride, float* d_flow, ebos_stream_t stream) {
  return ebos::warp_dense_bwd_impl<float>(events, tminmax, ref_mode, ref_fraction, normalize_t, d_warped, b, n, H, W,
                                          row_stride, d_flow, stream);
}
int ebos_warp_dense_bwd_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                            int normalize_t, const double* d_warped, int64_t b, int64_t n, int H, int W,
                            int row_stride, double* d_flow, ebos_stream_t stream) {
  return ebos::warp_dense_bwd_impl<double>(events, tminmax, ref_mode, ref_fraction, normalize_t, d_warped, b, n, H, W,
                                           row_stride, d_flow, stream);
}

int ebos_warp_2dof_f32(const float* events, const float* theta, const float* tminmax, int ref_mode,
                       double ref_fraction, int normalize_t, const float* time_period, int64_t n, float* warped,
                       ebos_stream_t stream) {
  return ebos::warp_2dof_impl<float>(events, theta, tminmax, ref_mode, ref_fraction, normalize_t, time_period, n,
                                     warped, stream);
}
int ebos_warp_2dof_f64(const double* events, const double* theta, const double* tminmax, int ref_mode,
                       double ref_fraction, int normalize_t, const double* time_period, int64_t n, double* warped,
                       ebos_stream_t stream) {
  return ebos::warp_2dof_impl<double>(events, theta, tminmax, ref_mode, ref_fraction, normalize_t, time_period, n,
                                      warped, stream);
}
int ebos_warp_2dof_bwd_f32(const float* events, const float* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, const float* time_period, const float* d_warped, int64_t n,
                           float* d_theta, ebos_stream_t stream) {
  return ebos::warp_2dof_bwd_impl<float>(events, tminmax, ref_mode, ref_fraction, normalize_t, time_period, d_warped,
                                         n, d_theta, stream);
}
int ebos_warp_2dof_bwd_f64(const double* events, const double* tminmax, int ref_mode, double ref_fraction,
                           int normalize_t, const double* time_period, const double* d_warped, int64_t n,
                           double* d_theta, ebos_stream_t stream) {
  return ebos::warp_2dof_bwd_impl<double>(events, tminmax, ref_mode, ref_fraction, normalize_t, time_period, d_warped,
                                          n, d_theta, stream);
}

}  // extern "C"
