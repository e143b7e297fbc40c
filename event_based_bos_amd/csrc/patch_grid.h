// patch_grid.h -- geometry of the patch-flow grid -> dense flow map (src/solver/patch_eklt.py:173-204 under
// /root/reference), shared by the stand-alone upsample kernels (flow_upsample.hip) and the tile-private event kernels
// that sample the grid themselves (iwe_tiled.hip).
#pragma once
#include "common.h"

namespace ebos {

// cells of the patch grid one source tile can touch per axis (tile / slide + 2, with a spare): the bound of the per-tile
// partial gradients the GRID backward kernel writes
constexpr int kGridCells = 16;

struct Axis {
  int g;      // grid cells on this axis
  int pad;    // replicate padding (cells)
  int slide;  // integer scale
  int off;    // first row/col of the crop in the un-cropped resize output
  int n_in;   // g + 2 pad
};

__host__ __device__ inline Axis make_axis(int g, int patch, int slide, int out) {
  Axis a;
  a.g = g;
  a.pad = (int)((patch / 2.0) / slide) + 1;  // int(patch / 2 // slide) + 1, src/solver/patch_eklt.py:183-184
  a.slide = slide;
  a.n_in = g + 2 * a.pad;
  const int n_full = a.n_in * slide;
  a.off = n_full / 2 - out / 2;              // :196-199
  return a;
}

struct Lerp {
  int i0, i1;  // grid indices (after un-padding + clamping)
  float w0, w1;
};

__host__ __device__ __forceinline__ Lerp lerp_at(const Axis& a, int r) {
  const int R = r + a.off;
  float src = ((float)R + 0.5f) / (float)a.slide - 0.5f;  // align_corners = False
  if (src < 0.0f) src = 0.0f;
  int p0 = (int)src;
  if (p0 > a.n_in - 1) p0 = a.n_in - 1;
  const int p1 = p0 < a.n_in - 1 ? p0 + 1 : p0;
  Lerp l;
  l.w1 = src - (float)p0;
  l.w0 = 1.0f - l.w1;
  int i0 = p0 - a.pad, i1 = p1 - a.pad;
  l.i0 = i0 < 0 ? 0 : (i0 > a.g - 1 ? a.g - 1 : i0);
  l.i1 = i1 < 0 ? 0 : (i1 > a.g - 1 ? a.g - 1 : i1);
  return l;
}

// value of the dense flow at a pixel whose row / column interpolation is (ly, lx); g0 / g1 = grid rows ly.i0 / ly.i1.
// ONE expression for every kernel that evaluates the map, so that they round alike.
__device__ __forceinline__ float grid_bilerp(const float* __restrict__ g0, const float* __restrict__ g1, const Lerp& ly,
                                             const Lerp& lx) {
  const float top = lx.w0 * g0[lx.i0] + lx.w1 * g0[lx.i1];
  const float bot = lx.w0 * g1[lx.i0] + lx.w1 * g1[lx.i1];
  return ly.w0 * top + ly.w1 * bot;
}

// conservative range [lo, hi) of output rows / columns whose interpolation can touch grid cell gi
__host__ __device__ __forceinline__ void support(const Axis& a, int gi, int n_out, int* lo, int* hi) {
  // padded indices that clamp onto this cell, +-1 cell of bilinear support, in output pixels
  const int p_lo = gi == 0 ? 0 : gi + a.pad, p_hi = gi == a.g - 1 ? a.n_in - 1 : gi + a.pad;
  int l = (p_lo - 1) * a.slide - a.off - 1, h = (p_hi + 2) * a.slide - a.off + 1;
  *lo = l < 0 ? 0 : l;
  *hi = h > n_out ? n_out : h;
}

__device__ __forceinline__ float weight_on(const Axis& a, int r, int gi) {
  const Lerp l = lerp_at(a, r);
  return (l.i0 == gi ? l.w0 : 0.0f) + (l.i1 == gi ? l.w1 : 0.0f);
}

// ---- image_gradient regulariser (src/costs/image_gradient.py:60-75): torch.gradient lines, shared by the stand-alone regulariser
// kernel (solver_kernels.hip) and the GRID backward kernel (iwe_tiled.hip), which evaluates it on the tile's own flow
__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// torch.gradient along one axis (spacing 1, edge_order 1) at index i of a line of n samples with stride st
__device__ __forceinline__ float central(const float* f, int i, int n, int64_t st) {
  if (i == 0) return f[st] - f[0];
  if (i == n - 1) return f[(int64_t)(n - 1) * st] - f[(int64_t)(n - 2) * st];
  return (f[(int64_t)(i + 1) * st] - f[(int64_t)(i - 1) * st]) * 0.5f;
}

// d/d f[i] of sum_k |central(f, k)|  (f: a line of n >= 2 samples)
__device__ __forceinline__ float tv_adjoint(const float* f, int i, int n, int64_t st) {
  float g = 0.0f;
  if (i >= 1) g += sgn(central(f, i - 1, n, st)) * (i - 1 == 0 ? 1.0f : 0.5f);          // k = i - 1 reads f[i] with +
  if (i + 1 <= n - 1) g -= sgn(central(f, i + 1, n, st)) * (i + 1 == n - 1 ? 1.0f : 0.5f);  // k = i + 1 reads f[i] with -
  if (i == 0) g -= sgn(central(f, 0, n, st));
  if (i == n - 1) g += sgn(central(f, n - 1, n, st));
  return g;
}

// ---- Adam (torch.optim.Adam, amsgrad off, no weight decay), shared by every kernel that steps the patch grid so that they
// round alike: the four-launch pipeline's cell combine pass (flow_upsample.hip) and the resident solver kernel
// (cmax_resident.hip) must follow the SAME trajectory bit for bit -- a last-bit difference in one step is amplified by the
// kinks of the piecewise-linear objective over a few hundred iterations.
struct AdamCoef {
  float step_size;  // lr / (1 - beta1^t)
  float bc2_sqrt;   // sqrt(1 - beta2^t)
};
// beta^t by repeated squaring: IEEE products only, so the host (make_adam_job) and the device (resident kernel) get the same bits
__host__ __device__ inline double pow_int(double b, int t) {
  double r = 1.0, p = b;
  while (t > 0) {
    if (t & 1) r *= p;
    p *= p;
    t >>= 1;
  }
  return r;
}
__host__ __device__ inline AdamCoef adam_coef(double lr, double beta1, double beta2, int t) {
  const double bc1 = 1.0 - pow_int(beta1, t), bc2 = 1.0 - pow_int(beta2, t);
  AdamCoef c;
  c.step_size = (float)(lr / bc1);
  c.bc2_sqrt = (float)sqrt(bc2);
  return c;
}
// one element's step; contraction off: every kernel executes exactly these multiplies and adds
__device__ __forceinline__ void adam_update(float g, float& m, float& v, float& th, float step_size, float bc2_sqrt, float beta2,
                                            float w1, float w2, float eps) {
#pragma clang fp contract(off)
  const float mi = m + w1 * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = v * beta2 + w2 * (g * g);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  const float denom = sqrtf(vi) / bc2_sqrt + eps;    // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
  m = mi;
  v = vi;
  th = th - step_size * (mi / denom);                // param.addcdiv_(exp_avg, denom, value = -step_size)
}

}  // namespace ebos
