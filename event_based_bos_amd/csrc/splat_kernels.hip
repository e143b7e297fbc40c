// splat_kernels.hip -- API-parity event -> image accumulation for gfx950 (array-of-structs events).
//
// Backs EventImageConverter.bilinear_vote_tensor/_numpy, count_event_*, the "polarity" method and
// create_eventmask for GPU-resident (already warped) events: src/event_image_converter.py:407-620,
// 355-363 under /root/reference.  One lane = one event = one 16/32-byte vector load; the four taps
// are hardware float atomics on the image (global_atomic_add_f32 / _f64, executed at the memory
// side).  This is the general, order-free path; the LDS-tiled fused path lives in iwe_fused.hip.
#include "common.h"

namespace ebos {
namespace {

template <typename T>
struct Ev4;
template <>
struct Ev4<float> {
  using type = float4;
};
template <>
struct Ev4<double> {
  using type = double4;
};

template <typename T, int MODE>
__global__ void __launch_bounds__(256)
splat_kernel(const T* __restrict__ events, const T* __restrict__ weight, T weight_scalar, T eps, int64_t n, int h,
             int w, int pad_h, int pad_w, T* image) {
  const int64_t row = blockIdx.y;
  const auto* ev = reinterpret_cast<const typename Ev4<T>::type*>(events) + row * n;
  const T* wt = weight ? weight + row * n : nullptr;
  const int64_t hw = (int64_t)h * w;
  T* img = image + row * hw * (MODE == EBOS_SPLAT_POLARITY ? 2 : 1);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const auto e = ev[i];
    const Footprint<T> f = footprint<T>(e.x, e.y, eps, pad_h, pad_w);
    T* dst = img;
    if (MODE == EBOS_SPLAT_POLARITY) dst += (e.w > T(0)) ? 0 : hw;  // src/event_image_converter.py:356,363
    const bool r0 = f.R >= 0 && f.R < h, r1 = f.R + 1 >= 0 && f.R + 1 < h;
    const bool c0 = f.C >= 0 && f.C < w, c1 = f.C + 1 >= 0 && f.C + 1 < w;
    const int64_t base = (int64_t)f.R * w + f.C;
    if (MODE == EBOS_SPLAT_COUNT) {  // :448-452: +1 per in-bounds neighbour
      if (r0 && c0) atomic_add(&dst[base], T(1));
      if (r1 && c0) atomic_add(&dst[base + w], T(1));
      if (r0 && c1) atomic_add(&dst[base + 1], T(1));
      if (r1 && c1) atomic_add(&dst[base + w + 1], T(1));
    } else {
      const T wv = wt ? wt[i] : weight_scalar;
      const T a = T(1) - f.fr, b = T(1) - f.fc;  // :611-614
      if (r0 && c0) atomic_add(&dst[base], a * b * wv);
      if (r1 && c0) atomic_add(&dst[base + w], f.fr * b * wv);
      if (r0 && c1) atomic_add(&dst[base + 1], a * f.fc * wv);
      if (r1 && c1) atomic_add(&dst[base + w + 1], f.fr * f.fc * wv);
    }
  }
}

// gather of the upstream image gradient at the four taps
template <typename T>
__global__ void __launch_bounds__(256)
splat_bwd_kernel(const T* __restrict__ events, const T* __restrict__ weight, T weight_scalar, T eps,
                 const T* __restrict__ d_image, int64_t n, int h, int w, int pad_h, int pad_w, T* __restrict__ d_events,
                 T* __restrict__ d_weight) {
  const int64_t row = blockIdx.y;
  const auto* ev = reinterpret_cast<const typename Ev4<T>::type*>(events) + row * n;
  const T* wt = weight ? weight + row * n : nullptr;
  const T* g = d_image + row * (int64_t)h * w;
  auto* de = d_events ? reinterpret_cast<typename Ev4<T>::type*>(d_events) + row * n : nullptr;
  T* dwt = d_weight ? d_weight + row * n : nullptr;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const auto e = ev[i];
    const Footprint<T> f = footprint<T>(e.x, e.y, eps, pad_h, pad_w);
    const bool r0 = f.R >= 0 && f.R < h, r1 = f.R + 1 >= 0 && f.R + 1 < h;
    const bool c0 = f.C >= 0 && f.C < w, c1 = f.C + 1 >= 0 && f.C + 1 < w;
    const int64_t base = (int64_t)f.R * w + f.C;
    const T g00 = (r0 && c0) ? g[base] : T(0);
    const T g10 = (r1 && c0) ? g[base + w] : T(0);
    const T g01 = (r0 && c1) ? g[base + 1] : T(0);
    const T g11 = (r1 && c1) ? g[base + w + 1] : T(0);
    const T wv = wt ? wt[i] : weight_scalar;
    const T a = T(1) - f.fr, b = T(1) - f.fc;
    if (de) {
      typename Ev4<T>::type o;
      o.x = wv * (b * (g10 - g00) + f.fc * (g11 - g01));
      o.y = wv * (a * (g01 - g00) + f.fr * (g11 - g10));
      o.z = T(0);
      o.w = T(0);
      de[i] = o;
    }
    if (dwt) dwt[i] = a * b * g00 + f.fr * b * g10 + a * f.fc * g01 + f.fr * f.fc * g11;
  }
}

template <typename T>
int splat_impl(const T* events, const T* weight, double weight_scalar, int mode, double eps, int64_t b, int64_t n,
               int h, int w, int pad_h, int pad_w, T* image, ebos_stream_t stream) {
  EBOS_REQUIRE(image != nullptr, "ebos_splat: image is NULL");
  EBOS_REQUIRE(events != nullptr || n == 0, "ebos_splat: events is NULL");
  EBOS_REQUIRE(b >= 1 && b <= 65535 && n >= 0 && h > 0 && w > 0 && pad_h >= 0 && pad_w >= 0,
               "ebos_splat: bad sizes b=%lld n=%lld h=%d w=%d", (long long)b, (long long)n, h, w);
  if (n == 0) return EBOS_OK;
  dim3 grid(stream_grid(n, 256), (unsigned)b), block(256);
  hipStream_t s = as_stream(stream);
  const T ws = static_cast<T>(weight_scalar), e = static_cast<T>(eps);
  switch (mode) {
    case EBOS_SPLAT_BILINEAR:
      splat_kernel<T, EBOS_SPLAT_BILINEAR><<<grid, block, 0, s>>>(events, weight, ws, e, n, h, w, pad_h, pad_w, image);
      break;
    case EBOS_SPLAT_COUNT:
      splat_kernel<T, EBOS_SPLAT_COUNT><<<grid, block, 0, s>>>(events, weight, ws, e, n, h, w, pad_h, pad_w, image);
      break;
    case EBOS_SPLAT_POLARITY:
      splat_kernel<T, EBOS_SPLAT_POLARITY><<<grid, block, 0, s>>>(events, weight, ws, e, n, h, w, pad_h, pad_w, image);
      break;
    default:
      set_error("ebos_splat: unknown mode %d", mode);
      return EBOS_ERR_INVALID_ARG;
  }
  EBOS_CHECK_LAUNCH("ebos_splat");
  return EBOS_OK;
}

template <typename T>
int splat_bwd_impl(const T* events, const T* weight, double weight_scalar, double eps, const T* d_image, int64_t b,
                   int64_t n, int h, int w, int pad_h, int pad_w, T* d_events, T* d_weight, ebos_stream_t stream) {
  EBOS_REQUIRE(d_image != nullptr, "ebos_splat_bwd: d_image is NULL");
  EBOS_REQUIRE(events != nullptr || n == 0, "ebos_splat_bwd: events is NULL");
  EBOS_REQUIRE(b >= 1 && b <= 65535 && n >= 0 && h > 0 && w > 0 && pad_h >= 0 && pad_w >= 0, "ebos_splat_bwd: bad sizes");
  if (n == 0 || (d_events == nullptr && d_weight == nullptr)) return EBOS_OK;
  dim3 grid(stream_grid(n, 256), (unsigned)b), block(256);
  splat_bwd_kernel<T><<<grid, block, 0, as_stream(stream)>>>(events, weight, static_cast<T>(weight_scalar),
                                                             static_cast<T>(eps), d_image, n, h, w, pad_h, pad_w,
                                                             d_events, d_weight);
  EBOS_CHECK_LAUNCH("ebos_splat_bwd");
  return EBOS_OK;
}

}  // namespace
}  // namespace ebos

extern "C" {
int ebos_splat_f32(const float* events, const float* weight, double weight_scalar, int mode, double eps, int64_t b,
                   int64_t n, int h, int w, int pad_h, int pad_w, float* image, ebos_stream_t stream) {
  return ebos::splat_impl<float>(events, weight, weight_scalar, mode, eps, b, n, h, w, pad_h, pad_w, image, stream);
}
int ebos_splat_f64(const double* events, const double* weight, double weight_scalar, int mode, double eps, int64_t b,
                   int64_t n, int h, int w, int pad_h, int pad_w, double* image, ebos_stream_t stream) {
  return ebos::splat_impl<double>(events, weight, weight_scalar, mode, eps, b, n, h, w, pad_h, pad_w, image, stream);
}
int ebos_splat_bwd_f32(const float* events, const float* weight, double weight_scalar, double eps,
                       const float* d_image, int64_t b, int64_t n, int h, int w, int pad_h, int pad_w,
                       float* d_events, float* d_weight, ebos_stream_t stream) {
  return ebos::splat_bwd_impl<float>(events, weight, weight_scalar, eps, d_image, b, n, h, w, pad_h, pad_w, d_events,
                                     d_weight, stream);
}
int ebos_splat_bwd_f64(const double* events, const double* weight, double weight_scalar, double eps,
                       const double* d_image, int64_t b, int64_t n, int h, int w, int pad_h, int pad_w,
                       double* d_events, double* d_weight, ebos_stream_t stream) {
  return ebos::splat_bwd_impl<double>(events, weight, weight_scalar, eps, d_image, b, n, h, w, pad_h, pad_w, d_events,
                                      d_weight, stream);
}
}  // extern "C"
