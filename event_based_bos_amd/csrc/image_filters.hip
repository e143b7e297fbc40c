// image_filters.hip -- optional Gaussian blur of an event image and its adjoint (cold path, kept on the
// device so that no image ever round-trips through the host).
//
// One separable pass: 1-D correlation along one axis of a tensor viewed as [outer, L, inner].
//   EventImageConverter.create_image_from_events_numpy  -> scipy gaussian_filter(image, sigma):
//       taps exp(-x^2 / 2 sigma^2), radius int(4 sigma + 0.5), boundary 'reflect' (d c b a | a b c d),
//       one pass per axis of the array -- batch/channel axes included
//       (src/event_image_converter.py:368-369 under /root/reference)
//   EventImageConverter.create_image_from_events_tensor -> torchvision gaussian_blur(kernel_size=3):
//       3 taps, boundary 'reflect' in torch's sense (d c b | a b c d), last two axes (:399-404)
// The host computes the taps (2r+1 doubles) and passes them as a device array.
#include "common.h"

namespace ebos {
namespace {

__device__ __forceinline__ int64_t reflect_index(int64_t i, int64_t L, int boundary) {
  if (L == 1) return 0;
  if (boundary == 0) {  // scipy 'reflect': half-sample symmetric, edge value repeated
    while (i < 0 || i >= L) i = i < 0 ? -i - 1 : 2 * L - i - 1;
  } else {              // torch 'reflect': whole-sample symmetric, edge value not repeated
    while (i < 0 || i >= L) i = i < 0 ? -i : 2 * (L - 1) - i;
  }
  return i;
}

template <typename T>
__global__ void __launch_bounds__(256)
gauss1d_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t outer, int64_t L, int64_t inner,
               const double* __restrict__ taps, int radius, int boundary) {
  const int64_t total = outer * L * inner;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_i = i % inner, l = (i / inner) % L, o = i / (inner * L);
    const T* base = in + o * L * inner + in_i;
    double acc = 0.0;
    for (int k = -radius; k <= radius; ++k) acc += taps[k + radius] * (double)base[reflect_index(l + k, L, boundary) * inner];
    out[i] = (T)acc;
  }
}

// Adjoint of the pass above (d loss / d in from d loss / d out).  Gather form, so the sum order is fixed:
// in[j] is read by out[i] through every virtual index v = i + k that reflects onto j.  Those v form two
// arithmetic progressions of period P (scipy: P = 2L, v = j or -j-1 mod P; torch: P = 2(L-1), v = j or -j mod P).
template <typename T>
__global__ void __launch_bounds__(256)
gauss1d_bwd_kernel(const T* __restrict__ g_out, T* __restrict__ g_in, int64_t outer, int64_t L, int64_t inner,
                   const double* __restrict__ taps, int radius, int boundary) {
  const int64_t total = outer * L * inner;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in_i = idx % inner, j = (idx / inner) % L, o = idx / (inner * L);
    const T* base = g_out + o * L * inner + in_i;
    double acc = 0.0;
    if (L == 1) {
      for (int k = 0; k <= 2 * radius; ++k) acc += taps[k];
      g_in[idx] = (T)(acc * (double)base[0]);
      continue;
    }
    const int64_t P = boundary == 0 ? 2 * L : 2 * (L - 1);
    const int64_t mirror = boundary == 0 ? -j - 1 : -j;
    const bool twin = ((j - mirror) % P) != 0;  // torch: j = 0 and j = L-1 are their own mirror image
    const int64_t m_lo = -(radius / P) - 2, m_hi = (L - 1 + radius) / P + 2;
    for (int64_t m = m_lo; m <= m_hi; ++m) {
      for (int side = 0; side < (twin ? 2 : 1); ++side) {
        const int64_t v = (side == 0 ? j : mirror) + m * P;
        if (v < -(int64_t)radius || v > L - 1 + radius) continue;
        const int64_t i_lo = v - radius > 0 ? v - radius : 0, i_hi = v + radius < L - 1 ? v + radius : L - 1;
        for (int64_t i = i_lo; i <= i_hi; ++i) acc += taps[v - i + radius] * (double)base[i * inner];
      }
    }
    g_in[idx] = (T)acc;
  }
}

template <typename T>
int gauss1d_bwd_impl(const T* g_out, T* g_in, int64_t outer, int64_t L, int64_t inner, const double* taps, int radius,
                     int boundary, ebos_stream_t stream) {
  EBOS_REQUIRE(g_out && g_in && taps && g_out != g_in, "ebos_gauss1d_bwd: NULL or aliased buffers");
  EBOS_REQUIRE(outer >= 1 && L >= 1 && inner >= 1 && radius >= 0 && (boundary == 0 || boundary == 1),
               "ebos_gauss1d_bwd: bad sizes");
  gauss1d_bwd_kernel<T><<<dim3(stream_grid(outer * L * inner, 256, 4096)), dim3(256), 0, as_stream(stream)>>>(
      g_out, g_in, outer, L, inner, taps, radius, boundary);
  EBOS_CHECK_LAUNCH("ebos_gauss1d_bwd");
  return EBOS_OK;
}

template <typename T>
int gauss1d_impl(const T* in, T* out, int64_t outer, int64_t L, int64_t inner, const double* taps, int radius,
                 int boundary, ebos_stream_t stream) {
  EBOS_REQUIRE(in && out && taps && in != out, "ebos_gauss1d: NULL or aliased buffers");
  EBOS_REQUIRE(outer >= 1 && L >= 1 && inner >= 1 && radius >= 0 && (boundary == 0 || boundary == 1),
               "ebos_gauss1d: bad sizes");
  gauss1d_kernel<T><<<dim3(stream_grid(outer * L * inner, 256, 4096)), dim3(256), 0, as_stream(stream)>>>(
      in, out, outer, L, inner, taps, radius, boundary);
  EBOS_CHECK_LAUNCH("ebos_gauss1d");
  return EBOS_OK;
}

}  // namespace
}  // namespace ebos

extern "C" {
int ebos_gauss1d_f32(const float* in, float* out, int64_t outer, int64_t L, int64_t inner, const double* taps,
                     int radius, int boundary, ebos_stream_t stream) {
  return ebos::gauss1d_impl<float>(in, out, outer, L, inner, taps, radius, boundary, stream);
}
int ebos_gauss1d_f64(const double* in, double* out, int64_t outer, int64_t L, int64_t inner, const double* taps,
                     int radius, int boundary, ebos_stream_t stream) {
  return ebos::gauss1d_impl<double>(in, out, outer, L, inner, taps, radius, boundary, stream);
}
int ebos_gauss1d_bwd_f32(const float* g_out, float* g_in, int64_t outer, int64_t L, int64_t inner, const double* taps,
                         int radius, int boundary, ebos_stream_t stream) {
  return ebos::gauss1d_bwd_impl<float>(g_out, g_in, outer, L, inner, taps, radius, boundary, stream);
}
int ebos_gauss1d_bwd_f64(const double* g_out, double* g_in, int64_t outer, int64_t L, int64_t inner, const double* taps,
                         int radius, int boundary, ebos_stream_t stream) {
  return ebos::gauss1d_bwd_impl<double>(g_out, g_in, outer, L, inner, taps, radius, boundary, stream);
}
}
