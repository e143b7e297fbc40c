// blur3.h -- the 3-tap blur of the tensor branch of create_iwe and its adjoint, as device functions shared by the image kernel of the
// four-launch pipeline (cost_kernels.hip), the backward event kernels (iwe_tile_core.h) and the resident solver kernel
// (cmax_resident.hip): one arithmetic, so that the forms of the loop agree to the last bit wherever they see the same image.
//
// reference: EventImageConverter.create_image_from_events_tensor -> torchvision gaussian_blur(image, kernel_size=3, sigma)
// (src/event_image_converter.py:399-404): taps exp(-x^2 / 2 sigma^2) at x = -1, 0, 1 normalised to sum 1, separable, `reflect`
// padding in torch's sense (d c b | a b c d: the edge sample is not repeated).  With y = B x the contrast is taken on y:
//     loss = -w var(m . y)        (m: the valid region, omit_boundary)
//     d loss / d x = B^T u,  u = m . (a y + c),  a = 2 (-w) / (M - 1),  c = -a mean(y)
//                  = a z + c wgt,   z = B^T (m . y)  (linear in x, needs no mean),   wgt = B^T m  (a function of the position only).
// B is separable, B = B_rows (x) B_cols; along an axis of L samples
//     y(i) = k0 x(i - 1) + k1 x(i) + k0 x(i + 1),   x(-1) = x(1),  x(L) = x(L - 2)
// i.e. the coefficient of x(j) in y(i) is k1 for j == i, k0 for |j - i| == 1 -- doubled where the reflection folds the missing
// neighbour onto j (i == 0, j == 1 and i == L - 1, j == L - 2) -- and 0 otherwise.  L >= 2 (torch refuses to reflect-pad an axis of 1).
#pragma once
#include "common.h"

namespace ebos {

struct Blur3 {
  float k0, k1;  // taps (k0, k1, k0); k0 == 0: no blur
};

// coefficient of x(j) in y(i), |j - i| <= 1 assumed by the callers' loops; 0 where i or j lies outside [0, L)
__device__ __forceinline__ float blur3_coef(int i, int j, int L, const Blur3& b) {
  if (i < 0 || i >= L || j < 0 || j >= L) return 0.0f;
  if (i == j) return b.k1;
  const bool folded = (i == 0 && j == 1) || (i == L - 1 && j == L - 2);
  return folded ? 2.0f * b.k0 : b.k0;
}

// Away from the border every coefficient is a plain tap (forward: the pixel's 3 x 3 neighbourhood inside the image; adjoint: the
// pixel two away from the border): the general forms below term for term -- same roundings -- without their ~150 compares and
// selects per pixel.  A caller whose whole window is interior calls this directly (a uniform choice per tile: the passes over an LDS
// window are bound by instruction issue).
template <typename X>
__device__ __forceinline__ float blur3_interior(X&& x, int r, int c, const Blur3& b) {
#pragma clang fp contract(off)
  float y = 0.0f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    float t = 0.0f;
    t += b.k0 * x(r - 1, c + dc);
    t += b.k1 * x(r, c + dc);
    t += b.k0 * x(r + 1, c + dc);
    y += (dc == 0 ? b.k1 : b.k0) * t;
  }
  return y;
}

// ... four horizontally adjacent interior pixels at once: the three rows as quads (a = columns c - 4 .. c - 1, b = c .. c + 3,
// c = c + 4 .. c + 7; of a and c one column each is used).  Column sums are shared between neighbouring pixels; every output is
// blur3_interior's term for term.  All in named registers: arrays indexed through a pointer end up in scratch.
__device__ __forceinline__ float4 blur3_interior_quad(const float4& ua, const float4& ub, const float4& uc, const float4& ma,
                                                      const float4& mb, const float4& mc, const float4& da, const float4& db,
                                                      const float4& dc, const Blur3& b) {
#pragma clang fp contract(off)
  auto col = [&](float u, float m, float d) {
    float t = 0.0f;
    t += b.k0 * u;
    t += b.k1 * m;
    t += b.k0 * d;
    return t;
  };
  const float t0 = col(ua.w, ma.w, da.w), t1 = col(ub.x, mb.x, db.x), t2 = col(ub.y, mb.y, db.y), t3 = col(ub.z, mb.z, db.z),
              t4 = col(ub.w, mb.w, db.w), t5 = col(uc.x, mc.x, dc.x);
  auto out = [&](float l, float m, float r) {
    float y = 0.0f;
    y += b.k0 * l;
    y += b.k1 * m;
    y += b.k0 * r;
    return y;
  };
  return make_float4(out(t0, t1, t2), out(t1, t2, t3), out(t2, t3, t4), out(t3, t4, t5));
}

// y(r, c) from an accessor x(r, c) that is only asked for positions INSIDE the image
template <typename X>
__device__ __forceinline__ float blur3_fwd_at(X&& x, int r, int c, int h, int w, const Blur3& b) {
#pragma clang fp contract(off)
  if (r >= 1 && r < h - 1 && c >= 1 && c < w - 1) return blur3_interior(x, r, c, b);
  float y = 0.0f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    const float cc = blur3_coef(c, c + dc, w, b);
    float t = 0.0f;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
      const float cr = blur3_coef(r, r + dr, h, b);
      t += cr != 0.0f && cc != 0.0f ? cr * x(r + dr, c + dc) : 0.0f;
    }
    y += cc * t;
  }
  return y;
}

// z(r, c) = (B^T u)(r, c) from an accessor u(r, c) that is only asked for positions inside the image (u = the masked blurred image)
template <typename U>
__device__ __forceinline__ float blur3_adj_at(U&& u, int r, int c, int h, int w, const Blur3& b) {
#pragma clang fp contract(off)
  if (r >= 2 && r < h - 2 && c >= 2 && c < w - 2) return blur3_interior(u, r, c, b);  // (no folded coefficient reads this pixel)
  float z = 0.0f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    const float cc = blur3_coef(c + dc, c, w, b);
    float t = 0.0f;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
      const float cr = blur3_coef(r + dr, r, h, b);
      t += cr != 0.0f && cc != 0.0f ? cr * u(r + dr, c + dc) : 0.0f;
    }
    z += cc * t;
  }
  return z;
}

// The same two values from an accessor that may be asked for the WHOLE 3 x 3 neighbourhood (an LDS window with an apron: positions
// outside the image hold something finite or not -- they are read and discarded): all nine loads are in flight before the first
// product, where the forms above branch around every load they do not need.  Term for term the forms above: same bits.
template <typename X>
__device__ __forceinline__ float blur3_fwd_at_dense(X&& x, int r, int c, int h, int w, const Blur3& b) {
#pragma clang fp contract(off)
  float v[3][3];
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc)
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) v[dc + 1][dr + 1] = x(r + dr, c + dc);
  const float cr[3] = {blur3_coef(r, r - 1, h, b), blur3_coef(r, r, h, b), blur3_coef(r, r + 1, h, b)};
  float y = 0.0f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    const float cc = blur3_coef(c, c + dc, w, b);
    float t = 0.0f;
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) t += cr[dr] != 0.0f && cc != 0.0f ? cr[dr] * v[dc + 1][dr] : 0.0f;
    y += cc * t;
  }
  return y;
}
template <typename U>
__device__ __forceinline__ float blur3_adj_at_dense(U&& u, int r, int c, int h, int w, const Blur3& b) {
#pragma clang fp contract(off)
  float v[3][3];
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc)
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) v[dc + 1][dr + 1] = u(r + dr, c + dc);
  const float cr[3] = {blur3_coef(r - 1, r, h, b), blur3_coef(r, r, h, b), blur3_coef(r + 1, r, h, b)};
  float z = 0.0f;
#pragma unroll
  for (int dc = -1; dc <= 1; ++dc) {
    const float cc = blur3_coef(c + dc, c, w, b);
    float t = 0.0f;
#pragma unroll
    for (int dr = 0; dr < 3; ++dr) t += cr[dr] != 0.0f && cc != 0.0f ? cr[dr] * v[dc + 1][dr] : 0.0f;
    z += cc * t;
  }
  return z;
}

// (B^T m) along one axis: the sum of the coefficients with which x(j) enters the valid outputs lo <= i < L - lo
__device__ __forceinline__ float blur3_axis_weight(int j, int L, int lo, const Blur3& b) {
#pragma clang fp contract(off)
  float s = 0.0f;
#pragma unroll
  for (int d = -1; d <= 1; ++d) {
    const int i = j + d;
    s += (i >= lo && i < L - lo) ? blur3_coef(i, j, L, b) : 0.0f;
  }
  return s;
}
__device__ __forceinline__ float blur3_weight(int r, int c, int h, int w, int lo, const Blur3& b) {
#pragma clang fp contract(off)
  if (r >= lo + 2 && r < h - lo - 2 && c >= lo + 2 && c < w - lo - 2) {  // three valid outputs per axis read the pixel, none folded
    float wi = 0.0f;
    wi += b.k0;
    wi += b.k1;
    wi += b.k0;
    return wi * wi;
  }
  return blur3_axis_weight(r, h, lo, b) * blur3_axis_weight(c, w, lo, b);
}

}  // namespace ebos
