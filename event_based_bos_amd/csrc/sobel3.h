// sobel3.h -- the gradient-magnitude contrast's stencils as device functions shared by the image kernel of the four-launch pipeline
// (gradmag_fused_kernel, cost_kernels.hip) and the resident solver kernel (cmax_resident_core.h): one arithmetic, so that both
// forms of the loop stage the same upstream image to the last bit.
//
// reference: the `gradient_magnitude` cost -- mean(gx^2 + gy^2) over the valid region, (gx, gy) = SobelTorch(ksize = 3)(iwe) / 8 with
// replicate padding (src/costs/gradient_magnitude.py, src/utils/stat_utils.py: SobelTorch).
//   Gx = [[-1, -2, -1], [0, 0, 0], [1, 2, 1]] / 8  (row derivative),   Gy = [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]] / 8  (column derivative)
// With s = 2 upstream / M / 8 the adjoint at pixel p is s times the sum, over the stencils q = p - d that read p through their tap d, of
// gx(q) Gx[d] + gy(q) Gy[d] (stencils outside the valid region count zero); replicate padding folds the taps that aim outside the
// image onto its outermost ring, whose pixels therefore also collect the plain sums of their mirror positions.
// Every product below is by 0, +-1 or +-2 -- exact -- so the results do not depend on whether the compiler contracts them into
// fused multiply-adds; the ORDER of the additions is what the two users share.
#pragma once
#include "common.h"

namespace ebos {

// the Sobel pair of the 3 x 3 neighbourhood a[row][column] (already clamped to the image by the caller)
__device__ __forceinline__ void sobel3_pair(float a00, float a01, float a02, float a10, float a12, float a20, float a21, float a22,
                                            float& gx, float& gy) {
#pragma clang fp contract(off)
  gx = ((a20 + 2.0f * a21 + a22) - (a00 + 2.0f * a01 + a02)) * 0.125f;
  gy = ((a02 + 2.0f * a12 + a22) - (a00 + 2.0f * a10 + a20)) * 0.125f;
}

// ... of four horizontally adjacent pixels: rows above / at / below as quads (l = the four columns left of the pixels, m = theirs,
// r = the four right of them; of l and r one column each is used).  sobel3_pair's sums, shared between neighbours.
__device__ __forceinline__ void sobel3_pair_quad(const float4& ul, const float4& um, const float4& ur, const float4& ml, const float4& mm,
                                                 const float4& mr, const float4& dl, const float4& dm, const float4& dr, float4& gx,
                                                 float4& gy) {
#pragma clang fp contract(off)
  auto row = [](float l, float m, float r) { return l + 2.0f * m + r; };   // (a + 2 b) + c
  const float u0 = row(ul.w, um.x, um.y), u1 = row(um.x, um.y, um.z), u2 = row(um.y, um.z, um.w), u3 = row(um.z, um.w, ur.x);
  const float d0 = row(dl.w, dm.x, dm.y), d1 = row(dm.x, dm.y, dm.z), d2 = row(dm.y, dm.z, dm.w), d3 = row(dm.z, dm.w, dr.x);
  gx = make_float4((d0 - u0) * 0.125f, (d1 - u1) * 0.125f, (d2 - u2) * 0.125f, (d3 - u3) * 0.125f);
  // columns -1 .. 4 relative to the quad: (up + 2 mid) + down
  const float c0 = row(ul.w, ml.w, dl.w), c1 = row(um.x, mm.x, dm.x), c2 = row(um.y, mm.y, dm.y), c3 = row(um.z, mm.z, dm.z),
              c4 = row(um.w, mm.w, dm.w), c5 = row(ur.x, mr.x, dr.x);
  gy = make_float4((c2 - c0) * 0.125f, (c3 - c1) * 0.125f, (c4 - c2) * 0.125f, (c5 - c3) * 0.125f);
}

// the nine-stencil gather at a pixel inside the image's outermost ring: gxu / gxd = gx on the row above / below the pixel at columns
// -1, 0, +1; gy?l / gy?r = gy on the column left / right of it at rows -1, 0, +1
__device__ __forceinline__ float sobel3_adjoint_interior(float gxu_l, float gxu_m, float gxu_r, float gxd_l, float gxd_m, float gxd_r,
                                                         float gyu_l, float gym_l, float gyd_l, float gyu_r, float gym_r, float gyd_r) {
#pragma clang fp contract(off)
  return (gxu_l + 2.0f * gxu_m + gxu_r) - (gxd_l + 2.0f * gxd_m + gxd_r)   // tap dr = +1: the stencil on the row above; -1: below
       + (gyu_l + 2.0f * gym_l + gyd_l) - (gyu_r + 2.0f * gym_r + gyd_r);  // tap dc = +1: the stencil on the column to the left; -1: right
}

// ... of four adjacent pixels from quads of gx (rows above / below) and gy (rows above / at / below): l, m, r as in sobel3_pair_quad
__device__ __forceinline__ float4 sobel3_adjoint_quad(const float4& xul, const float4& xum, const float4& xur, const float4& xdl,
                                                      const float4& xdm, const float4& xdr, const float4& yul, const float4& yum,
                                                      const float4& yur, const float4& yml, const float4& ymm, const float4& ymr,
                                                      const float4& ydl, const float4& ydm, const float4& ydr) {
#pragma clang fp contract(off)
  auto row = [](float l, float m, float r) { return l + 2.0f * m + r; };
  const float u0 = row(xul.w, xum.x, xum.y), u1 = row(xum.x, xum.y, xum.z), u2 = row(xum.y, xum.z, xum.w), u3 = row(xum.z, xum.w, xur.x);
  const float d0 = row(xdl.w, xdm.x, xdm.y), d1 = row(xdm.x, xdm.y, xdm.z), d2 = row(xdm.y, xdm.z, xdm.w), d3 = row(xdm.z, xdm.w, xdr.x);
  const float c0 = row(yul.w, yml.w, ydl.w), c1 = row(yum.x, ymm.x, ydm.x), c2 = row(yum.y, ymm.y, ydm.y), c3 = row(yum.z, ymm.z, ydm.z),
              c4 = row(yum.w, ymm.w, ydm.w), c5 = row(yur.x, ymr.x, ydr.x);
  return make_float4(u0 - d0 + c0 - c2, u1 - d1 + c1 - c3, u2 - d2 + c2 - c4, u3 - d3 + c3 - c5);
}

// the outermost ring (pr == 0 || pr == h - 1 || pc == 0 || pc == w - 1): with p' = q + d the unclamped position a tap aims at, the
// gather -- over the stencils q, over their taps d with clamp(q + d) = p -- is the sum, over the positions p' that clamp onto p (p
// itself and its one or three mirror positions outside the image), of the plain nine-term form at p'.  gxy(qr, qc, vx, vy) yields the
// pair of a stencil INSIDE the valid region [r0, r1) x [c0, c1) (it is not asked for others).
template <typename GXY>
__device__ __forceinline__ float sobel3_adjoint_ring(GXY&& gxy, int pr, int pc, int h, int w, int r0, int r1, int c0, int c1) {
#pragma clang fp contract(off)
  auto plain = [&](int r, int c) {
    float f = 0.0f;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr)
#pragma unroll
      for (int dc = -1; dc <= 1; ++dc) {
        const int qr = r - dr, qc = c - dc;
        const bool ok = qr >= r0 && qr < r1 && qc >= c0 && qc < c1;  // (then q lies within one pixel of p)
        float vx = 0.0f, vy = 0.0f;
        if (ok) gxy(qr, qc, vx, vy);
        f += vx * ((float)dr * (dc == 0 ? 2.0f : 1.0f)) + vy * ((float)dc * (dr == 0 ? 2.0f : 1.0f));
      }
    return f;
  };
  float acc = 0.0f;
  for (int a2 = (pr == 0 ? -1 : 0); a2 <= (pr == h - 1 ? 1 : 0); ++a2)
    for (int b2 = (pc == 0 ? -1 : 0); b2 <= (pc == w - 1 ? 1 : 0); ++b2) acc += plain(pr + a2, pc + b2);
  return acc;
}

// The same value from an accessor that may be asked for ANY stencil position (an LDS window: it clamps the address itself) -- all
// nine pairs of a plain form are requested before the first product, where the form above branches around each pair it does not
// need.  A pair outside the valid region counts zero, as above: same bits.
template <typename GXY>
__device__ __forceinline__ float sobel3_adjoint_ring_dense(GXY&& gxy, int pr, int pc, int h, int w, int r0, int r1, int c0, int c1) {
#pragma clang fp contract(off)
  auto plain = [&](int r, int c) {
    float vx[9], vy[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gxy(r - (k / 3 - 1), c - (k % 3 - 1), vx[k], vy[k]);
    float f = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int dr = k / 3 - 1, dc = k % 3 - 1, qr = r - dr, qc = c - dc;
      const bool ok = qr >= r0 && qr < r1 && qc >= c0 && qc < c1;
      const float ux = ok ? vx[k] : 0.0f, uy = ok ? vy[k] : 0.0f;
      f += ux * ((float)dr * (dc == 0 ? 2.0f : 1.0f)) + uy * ((float)dc * (dr == 0 ? 2.0f : 1.0f));
    }
    return f;
  };
  float acc = 0.0f;
  for (int a2 = (pr == 0 ? -1 : 0); a2 <= (pr == h - 1 ? 1 : 0); ++a2)
    for (int b2 = (pc == 0 ? -1 : 0); b2 <= (pc == w - 1 ? 1 : 0); ++b2) acc += plain(pr + a2, pc + b2);
  return acc;
}

// one stencil's term of the contrast's value
__device__ __forceinline__ float sobel3_energy(float gx, float gy) {
#pragma clang fp contract(off)
  return gx * gx + gy * gy;
}

// s = 2 upstream / M / 8: the factor between the gather and d contrast / d image
__device__ __forceinline__ float sobel3_adjoint_scale(double upstream, double n_valid) { return (float)(2.0 * upstream / n_valid * 0.125); }

}  // namespace ebos
