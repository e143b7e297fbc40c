// cost_kernels.hip -- contrast costs on the image of warped events, and their image gradients.
//
// The release of the reference ships no contrast cost; A14 defines them on the
// reference's own primitives:
//   image_variance      L = var(iwe)  (torch.var, unbiased; keys per src/solver/base.py:337-339)
//   gradient_magnitude  L = mean(gx^2 + gy^2), (gx, gy) = SobelTorch(ksize=3)(iwe) / 8 with replicate
//                       padding (src/utils/stat_utils.py:69-92, 117-118, 136-139)
// `omit_boundary` crops one pixel per side (the [..., 1:-1, 1:-1] idiom of the reference's costs).
// Sign handling (minimize / maximize) stays in the Python cost classes; these kernels return the
// raw contrast.  One pass over a 3.7 MB image: HBM/L2-bound streaming reductions, partial sums in
// fp64, one f64 atomic per workgroup.
#include <hip/hip_ext.h>

#include "common.h"
#include "blur3.h"
#include "sobel3.h"

namespace ebos {
namespace {

constexpr int kCostBlock = 256;

struct Region {
  int r0, r1, c0, c1;  // [r0, r1) x [c0, c1)
  __host__ __device__ int64_t count() const { return (int64_t)(r1 - r0) * (c1 - c0); }
};
inline Region make_region(int h, int w, int omit) {
  Region g;
  g.r0 = omit ? 1 : 0;
  g.c0 = omit ? 1 : 0;
  g.r1 = omit ? h - 1 : h;
  g.c1 = omit ? w - 1 : w;
  if (g.r1 < g.r0) g.r1 = g.r0;
  if (g.c1 < g.c0) g.c1 = g.c0;
  return g;
}

// ---- variance ---------------------------------------------------------------------------------
// Reductions write one partial per workgroup and a one-workgroup finalize sums them in a fixed order:
// same-address global atomics serialise at the memory side (~88 per us on MI355X), which made a
// 1024-workgroup atomic reduction of a 3.7 MB image take 28 us.
constexpr int kCostGrid = 240;  // workgroups per image

template <typename T>
__global__ void __launch_bounds__(kCostBlock)
moments_kernel(const T* __restrict__ images, int h, int w, Region rg, double* partials /*[K][grid][2]*/) {
  const int k = blockIdx.y;
  const T* img = images + (int64_t)k * h * w;
  double s = 0.0, ss = 0.0;
  for (int r = rg.r0 + blockIdx.x; r < rg.r1; r += gridDim.x) {
    const T* row = img + (int64_t)r * w;
    for (int c = rg.c0 + threadIdx.x; c < rg.c1; c += kCostBlock) {
      const double v = (double)row[c];
      s += v;
      ss += v * v;
    }
  }
  __shared__ double red[kCostBlock / kWave];
  s = block_sum(s, red);
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    double* p = partials + ((int64_t)k * gridDim.x + blockIdx.x) * 2;
    p[0] = s;
    p[1] = ss;
  }
}

template <typename T>
__global__ void __launch_bounds__(kCostBlock)
variance_finalize_kernel(const double* __restrict__ partials, int nparts, int64_t m, T* out, double* moments) {
  const int k = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kCostBlock) {
    s += partials[((int64_t)k * nparts + i) * 2];
    ss += partials[((int64_t)k * nparts + i) * 2 + 1];
  }
  __shared__ double red[kCostBlock / kWave];
  s = block_sum(s, red);
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    const double mean = m > 0 ? s / (double)m : 0.0;
    const double var = (ss - s * mean) / (double)(m - 1);  // unbiased (torch.var default); m <= 1 -> nan/inf like torch
    out[k] = (T)var;
    if (moments) {
      moments[2 * k] = mean;
      moments[2 * k + 1] = (double)m;
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(kCostBlock)
variance_grad_kernel(const T* __restrict__ images, int h, int w, Region rg, const double* __restrict__ moments,
                     const T* __restrict__ upstream, T* __restrict__ d_images) {
  const int k = blockIdx.y;
  const int64_t hw = (int64_t)h * w;
  const T* img = images + k * hw;
  T* d = d_images + k * hw;
  const double mean = moments[2 * k], m = moments[2 * k + 1];
  const double scale = 2.0 * (double)upstream[k] / (m - 1.0);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / w), c = (int)(i % w);
    const bool in = r >= rg.r0 && r < rg.r1 && c >= rg.c0 && c < rg.c1;
    d[i] = in ? (T)(scale * ((double)img[i] - mean)) : T(0);
  }
}

__global__ void variance_affine_kernel(const double* moments, const float* upstream, int K, float* affine) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const double mean = moments[2 * k], m = moments[2 * k + 1];
  const double a = 2.0 * (double)upstream[k] / (m - 1.0);
  affine[2 * k] = (float)a;
  affine[2 * k + 1] = (float)(-a * mean);
}

// ---- gradient magnitude (Sobel 3x3 / 8, replicate padding) --------------------------------------
template <typename T>
__device__ __forceinline__ void sobel_at(const T* __restrict__ img, int h, int w, int r, int c, double* gx, double* gy) {
  const int rm = r > 0 ? r - 1 : 0, rp = r < h - 1 ? r + 1 : h - 1;
  const int cm = c > 0 ? c - 1 : 0, cp = c < w - 1 ? c + 1 : w - 1;
  const double a00 = img[(int64_t)rm * w + cm], a01 = img[(int64_t)rm * w + c], a02 = img[(int64_t)rm * w + cp];
  const double a10 = img[(int64_t)r * w + cm], a12 = img[(int64_t)r * w + cp];
  const double a20 = img[(int64_t)rp * w + cm], a21 = img[(int64_t)rp * w + c], a22 = img[(int64_t)rp * w + cp];
  // Gx = [[-1,-2,-1],[0,0,0],[1,2,1]] (row derivative); Gy = [[-1,0,1],[-2,0,2],[-1,0,1]] (column derivative)
  *gx = ((a20 + 2.0 * a21 + a22) - (a00 + 2.0 * a01 + a02)) * 0.125;
  *gy = ((a02 + 2.0 * a12 + a22) - (a00 + 2.0 * a10 + a20)) * 0.125;
}

template <typename T>
__global__ void __launch_bounds__(kCostBlock)
gradmag_kernel(const T* __restrict__ images, int h, int w, Region rg, double* partials /*[K][grid][2]*/) {
  const int k = blockIdx.y;
  const T* img = images + (int64_t)k * h * w;
  double s = 0.0;
  for (int r = rg.r0 + blockIdx.x; r < rg.r1; r += gridDim.x)
    for (int c = rg.c0 + threadIdx.x; c < rg.c1; c += kCostBlock) {
      double gx, gy;
      sobel_at(img, h, w, r, c, &gx, &gy);
      s += gx * gx + gy * gy;
    }
  __shared__ double red[kCostBlock / kWave];
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    double* p = partials + ((int64_t)k * gridDim.x + blockIdx.x) * 2;
    p[0] = s;
    p[1] = 0.0;
  }
}

template <typename T>
__global__ void __launch_bounds__(kCostBlock)
gradmag_finalize_kernel(const double* __restrict__ partials, int nparts, int64_t m, T* out) {
  const int k = blockIdx.x;
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kCostBlock) s += partials[((int64_t)k * nparts + i) * 2];
  __shared__ double red[kCostBlock / kWave];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[k] = (T)(s / (double)m);
}

// adjoint of (Sobel with replicate padding) o (square, mean): gather form.  For the input pixel p
// every output pixel q within one pixel of p is visited, and every tap d of q's 3x3 window whose
// CLAMPED position equals p contributes (gx(q) Gx[d] + gy(q) Gy[d]) / 8 * 2 / M.
template <typename T>
__global__ void __launch_bounds__(kCostBlock)
gradmag_grad_kernel(const T* __restrict__ images, int h, int w, Region rg, const T* __restrict__ upstream,
                    T* __restrict__ d_images) {
  const int k = blockIdx.y;
  const int64_t hw = (int64_t)h * w;
  const T* img = images + k * hw;
  T* d = d_images + k * hw;
  const double scale = 2.0 * (double)upstream[k] / (double)rg.count() * 0.125;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
    const int pr = (int)(i / w), pc = (int)(i % w);
    double acc = 0.0;
    for (int qr = pr - 1; qr <= pr + 1; ++qr) {
      if (qr < rg.r0 || qr >= rg.r1) continue;
      for (int qc = pc - 1; qc <= pc + 1; ++qc) {
        if (qc < rg.c0 || qc >= rg.c1) continue;
        double gx, gy;
        sobel_at(img, h, w, qr, qc, &gx, &gy);
        for (int dr = -1; dr <= 1; ++dr) {
          int tr = qr + dr;
          tr = tr < 0 ? 0 : (tr > h - 1 ? h - 1 : tr);
          if (tr != pr) continue;
          for (int dc = -1; dc <= 1; ++dc) {
            int tc = qc + dc;
            tc = tc < 0 ? 0 : (tc > w - 1 ? w - 1 : tc);
            if (tc != pc) continue;
            const double kx = (double)dr * (dc == 0 ? 2.0 : 1.0);  // Gx[dr][dc]
            const double ky = (double)dc * (dr == 0 ? 2.0 : 1.0);  // Gy[dr][dc]
            acc += gx * kx + gy * ky;
          }
        }
      }
    }
    d[i] = (T)(scale * acc);
  }
}

// ---- gradient magnitude, value partials AND d contrast / d image in ONE pass over an LDS-tiled image ----------------------------
// gradmag_kernel + gradmag_grad_kernel read the image twice, the second one 81 times per pixel (nine Sobel stencils, each
// re-evaluated for every pixel that a tap of theirs folds onto), all in fp64.  Here a 256-thread workgroup holds a 16 x 64 tile with
// a 2 px apron in LDS (replicate padding = clamped loads: the apron of a border tile holds the border's own pixels), evaluates the
// Sobel pair once per pixel of tile + 1 px (f32: sums of eight image values; zero where the stencil's centre lies outside the
// cost's region, so that it contributes to nothing), and every thread then sums the value partial of its four pixels (fp64) and
// gathers their adjoint: nine multiply-adds of neighbouring stencils for a pixel inside the image, the folded form of
// gradmag_grad_kernel (taps clamped onto the border) for the image's outermost ring only.  4 H W bytes read + 4 H W written:
// SURVEY 8(d)'s cost-kernel bytes, once.
constexpr int kGmTH = 16, kGmTW = 64;

// GRAD = false: the value partials only (d_img is not touched) -- an objective EVALUATION needs neither the gather of the nine stencils
// nor 4 H W bytes of gradient image (9.6 -> 6 us at 1280 x 720).
template <bool GRAD>
__global__ void __launch_bounds__(kCostBlock)
gradmag_fused_kernel(const float* __restrict__ img, int h, int w, Region rg, const float* __restrict__ upstream, float* __restrict__ d_img,
                     double* __restrict__ partials) {
  constexpr int IH = kGmTH + 4, IW = kGmTW + 4, SH = kGmTH + 2, SW = kGmTW + 2;
  __shared__ float s_img[IH * IW];
  __shared__ float s_gx[SH * SW], s_gy[SH * SW];
  const int tr0 = blockIdx.y * kGmTH, tc0 = blockIdx.x * kGmTW;
  // (every load of the tile in flight before the first LDS store: rolled, the loop waited for each load in turn -- six serial
  // round trips, 9.5 of the pass's 11 us)
  constexpr int kLoads = (IH * IW + kCostBlock - 1) / kCostBlock;
  float stage[kLoads];
#pragma unroll
  for (int k = 0; k < kLoads; ++k) {
    const int i = min((int)threadIdx.x + k * kCostBlock, IH * IW - 1);
    const int rl = i / IW, cl = i - rl * IW;
    const int r = min(max(tr0 - 2 + rl, 0), h - 1), c = min(max(tc0 - 2 + cl, 0), w - 1);
    stage[k] = img[(int64_t)r * w + c];
  }
#pragma unroll
  for (int k = 0; k < kLoads; ++k) {
    const int i = threadIdx.x + k * kCostBlock;
    if (i < IH * IW) s_img[i] = stage[k];
  }
  __syncthreads();
  if constexpr (!GRAD) {  // the tile's own stencils, summed where they are formed
    double val = 0.0;
#pragma unroll
    for (int k = 0; k < kGmTH * kGmTW / kCostBlock; ++k) {
      const int i = threadIdx.x + k * kCostBlock;
      const int rl = i / kGmTW, cl = i - rl * kGmTW;
      const int qr = tr0 + rl, qc = tc0 + cl;
      const float* p = s_img + (rl + 2) * IW + cl + 2;
      float vx, vy;
      sobel3_pair(p[-IW - 1], p[-IW], p[-IW + 1], p[-1], p[1], p[IW - 1], p[IW], p[IW + 1], vx, vy);
      const bool in = qr >= rg.r0 && qr < rg.r1 && qc >= rg.c0 && qc < rg.c1;   // (the region lies inside the image)
      val += in ? (double)sobel3_energy(vx, vy) : 0.0;
    }
    __shared__ double red_v[kCostBlock / kWave];
    val = block_sum(val, red_v);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = val;
    return;
  }
  for (int i = threadIdx.x; i < SH * SW; i += kCostBlock) {
    const int rl = i / SW, cl = i - rl * SW;
    const int qr = tr0 - 1 + rl, qc = tc0 - 1 + cl;  // the stencil's centre
    const float* p = s_img + (rl + 1) * IW + cl + 1;
    const bool in = qr >= rg.r0 && qr < rg.r1 && qc >= rg.c0 && qc < rg.c1;
    float vx, vy;  // (sobel3.h: the arithmetic shared with the resident solver kernel)
    sobel3_pair(p[-IW - 1], p[-IW], p[-IW + 1], p[-1], p[1], p[IW - 1], p[IW], p[IW + 1], vx, vy);
    s_gx[i] = in ? vx : 0.0f;
    s_gy[i] = in ? vy : 0.0f;
  }
  __syncthreads();
  const float scale = sobel3_adjoint_scale(upstream ? (double)upstream[0] : 1.0, (double)rg.count());
  double val = 0.0;
#pragma unroll
  for (int k = 0; k < kGmTH * kGmTW / kCostBlock; ++k) {
    const int i = threadIdx.x + k * kCostBlock;
    const int rl = i / kGmTW, cl = i - rl * kGmTW;
    const int pr = tr0 + rl, pc = tc0 + cl;
    if (pr >= h || pc >= w) continue;
    const float* gx = s_gx + (rl + 1) * SW + cl + 1;
    const float* gy = s_gy + (rl + 1) * SW + cl + 1;
    val += (double)sobel3_energy(gx[0], gy[0]);  // (zero outside the region)
    // stencil q = p - d reads p through its tap d:  sum_d gx(p - d) Gx[d] + gy(p - d) Gy[d],  Gx[dr][dc] = dr (2 - |dc|),
    // Gy[dr][dc] = dc (2 - |dr|)  (complete for a pixel inside the image; the outermost ring gets its folded taps below)
    const float acc = sobel3_adjoint_interior(gx[-SW - 1], gx[-SW], gx[-SW + 1], gx[SW - 1], gx[SW], gx[SW + 1], gy[-SW - 1], gy[-1],
                                              gy[SW - 1], gy[-SW + 1], gy[1], gy[SW + 1]);
    if (pr >= 1 && pr < h - 1 && pc >= 1 && pc < w - 1) d_img[(int64_t)pr * w + pc] = scale * acc;
  }
  // The image's outermost ring: taps of stencils at the border clamp onto it (replicate padding).  With p' = q + d the unclamped
  // position a tap aims at, the gather of gradmag_grad_kernel -- sum over q, over the taps d with clamp(q + d) = p -- is the sum over
  // the positions p' that clamp onto p (p itself and its one or three mirror positions outside the image) of the plain nine-term
  // form at p', stencils outside the region counting zero.  The ring's pixels of this tile are dealt to the threads one each, a
  // pass of their own: inside the loop above they made every wave of a border tile run both forms (6 of the pass's 12.7 us).
  if (tr0 == 0 || tr0 + kGmTH >= h || tc0 == 0 || tc0 + kGmTW >= w) {
    auto gxy = [&](int qr, int qc, float& vx, float& vy) {   // (a stencil inside the region lies within tile + 1 px)
      const int qi = (qr - tr0 + 1) * SW + qc - tc0 + 1;
      vx = s_gx[qi], vy = s_gy[qi];
    };
    // candidates: the tile's first / last row and first / last column (2 (TH + TW) slots; a slot counts if it lies on the ring)
    for (int i = threadIdx.x; i < 2 * (kGmTH + kGmTW); i += kCostBlock) {
      int rl, cl;
      if (i < kGmTW) rl = 0, cl = i;
      else if (i < 2 * kGmTW) rl = min(kGmTH, h - tr0) - 1, cl = i - kGmTW;
      else if (i < 2 * kGmTW + kGmTH) rl = i - 2 * kGmTW, cl = 0;
      else rl = i - 2 * kGmTW - kGmTH, cl = min(kGmTW, w - tc0) - 1;
      const int pr = tr0 + rl, pc = tc0 + cl;
      if (pr >= h || pc >= w) continue;
      const bool on_ring = pr == 0 || pr == h - 1 || pc == 0 || pc == w - 1;
      // (a corner pixel appears in a row slot and a column slot, a one-row / one-column tile's pixels twice: same value stored twice)
      if (!on_ring) continue;
      d_img[(int64_t)pr * w + pc] = scale * sobel3_adjoint_ring(gxy, pr, pc, h, w, rg.r0, rg.r1, rg.c0, rg.c1);
    }
  }
  __shared__ double red[kCostBlock / kWave];
  val = block_sum(val, red);
  if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = val;
}

// ---- variance of the 3-tap BLURRED image (iwe.blur_sigma > 0, src/event_image_converter.py:399-404), everything the solver loop needs
// of it in ONE pass: the (sum, sum of squares) partials of the valid blurred pixels, in the layout of the slab combine pass's
// partials (the backward event kernel reduces them the same way), and z = B^T (m . B x) -- the part of d var / d x that is linear in
// x; the backward kernel forms a z + c wgt from it once it knows the mean (blur3.h).  A 256-thread workgroup holds a 16 x 64 tile
// with a 2 px apron in LDS, the masked blurred image on tile + 1 px, and writes its tile of z.  4 H W bytes read + 4 H W written.
__global__ void __launch_bounds__(kCostBlock)
blur3_variance_adjoint_kernel(const float* __restrict__ img, int h, int w, Region rg, Blur3 bk, float* __restrict__ z_img,
                              double* __restrict__ partials) {
  constexpr int IH = kGmTH + 4, IW = kGmTW + 4, SH = kGmTH + 2, SW = kGmTW + 2;
  __shared__ float s_img[IH * IW];
  __shared__ float s_y[SH * SW];
  const int tr0 = blockIdx.y * kGmTH, tc0 = blockIdx.x * kGmTW;
  constexpr int kLoads = (IH * IW + kCostBlock - 1) / kCostBlock;
  float stage[kLoads];
#pragma unroll
  for (int k = 0; k < kLoads; ++k) {  // (every load in flight before the first LDS store; positions outside the image are never used)
    const int i = min((int)threadIdx.x + k * kCostBlock, IH * IW - 1);
    const int rl = i / IW, cl = i - rl * IW;
    const int r = min(max(tr0 - 2 + rl, 0), h - 1), c = min(max(tc0 - 2 + cl, 0), w - 1);
    stage[k] = img[(int64_t)r * w + c];
  }
#pragma unroll
  for (int k = 0; k < kLoads; ++k) {
    const int i = threadIdx.x + k * kCostBlock;
    if (i < IH * IW) s_img[i] = stage[k];
  }
  __syncthreads();
  auto x_at = [&](int r, int c) { return s_img[(r - tr0 + 2) * IW + (c - tc0 + 2)]; };
  double sm = 0.0, sq = 0.0;
  for (int i = threadIdx.x; i < SH * SW; i += kCostBlock) {
    const int rl = i / SW, cl = i - rl * SW;
    const int qr = tr0 - 1 + rl, qc = tc0 - 1 + cl;
    const bool in = qr >= rg.r0 && qr < rg.r1 && qc >= rg.c0 && qc < rg.c1;   // (the valid region lies inside the image)
    const float y = in ? blur3_fwd_at(x_at, qr, qc, h, w, bk) : 0.0f;
    s_y[i] = y;
    if (in && rl >= 1 && rl <= kGmTH && cl >= 1 && cl <= kGmTW) {  // the tile's own pixels
      sm += (double)y;
      sq += (double)y * (double)y;
    }
  }
  __syncthreads();
  auto u_at = [&](int r, int c) { return s_y[(r - tr0 + 1) * SW + (c - tc0 + 1)]; };
#pragma unroll
  for (int k = 0; k < kGmTH * kGmTW / kCostBlock; ++k) {
    const int i = threadIdx.x + k * kCostBlock;
    const int rl = i / kGmTW, cl = i - rl * kGmTW;
    const int pr = tr0 + rl, pc = tc0 + cl;
    if (pr < h && pc < w) z_img[(int64_t)pr * w + pc] = blur3_adj_at(u_at, pr, pc, h, w, bk);
  }
  __shared__ double red[2 * kCostBlock / kWave];
  block_sum2(sm, sq, red);
  if (threadIdx.x == 0) {
    const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    partials[2 * b] = sm;
    partials[2 * b + 1] = sq;
  }
}

__global__ void __launch_bounds__(kCostBlock) gradmag_fused_finalize_kernel(const double* __restrict__ partials, int nparts, int64_t m, float* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kCostBlock) s += partials[i];
  __shared__ double red[kCostBlock / kWave];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)m);
}

inline size_t cost_scratch(int K) { return (size_t)(K > 0 ? K : 0) * kCostGrid * 2 * sizeof(double) + 64; }

template <typename T>
int variance_impl(const T* images, int K, int h, int w, int omit, T* out, double* moments, void* scratch,
                  size_t scratch_bytes, ebos_stream_t stream) {
  EBOS_REQUIRE(images && out && scratch, "ebos_image_variance: NULL images/out/scratch");
  EBOS_REQUIRE(K >= 1 && K <= 65535 && h > 0 && w > 0, "ebos_image_variance: bad sizes K=%d h=%d w=%d", K, h, w);
  if (scratch_bytes < cost_scratch(K)) {
    set_error("ebos_image_variance: scratch too small (%zu < %zu)", scratch_bytes, cost_scratch(K));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  const Region rg = make_region(h, w, omit);
  double* partials = reinterpret_cast<double*>(scratch);
  const int64_t m = rg.count();
  moments_kernel<T><<<dim3(kCostGrid, K), dim3(kCostBlock), 0, s>>>(images, h, w, rg, partials);
  variance_finalize_kernel<T><<<dim3(K), dim3(kCostBlock), 0, s>>>(partials, kCostGrid, m, out, moments);
  EBOS_CHECK_LAUNCH("ebos_image_variance");
  return EBOS_OK;
}

template <typename T>
int variance_grad_impl(const T* images, int K, int h, int w, int omit, const double* moments, const T* upstream,
                       T* d_images, ebos_stream_t stream) {
  EBOS_REQUIRE(images && moments && upstream && d_images, "ebos_image_variance_grad: NULL argument");
  EBOS_REQUIRE(K >= 1 && K <= 65535 && h > 0 && w > 0, "ebos_image_variance_grad: bad sizes");
  const Region rg = make_region(h, w, omit);
  dim3 grid(stream_grid((int64_t)h * w, kCostBlock, 2048), K);
  variance_grad_kernel<T><<<grid, dim3(kCostBlock), 0, as_stream(stream)>>>(images, h, w, rg, moments, upstream, d_images);
  EBOS_CHECK_LAUNCH("ebos_image_variance_grad");
  return EBOS_OK;
}

template <typename T>
int gradmag_impl(const T* images, int K, int h, int w, int omit, T* out, void* scratch, size_t scratch_bytes,
                 ebos_stream_t stream) {
  EBOS_REQUIRE(images && out && scratch, "ebos_gradient_magnitude: NULL images/out/scratch");
  EBOS_REQUIRE(K >= 1 && K <= 65535 && h > 0 && w > 0, "ebos_gradient_magnitude: bad sizes");
  if (scratch_bytes < cost_scratch(K)) {
    set_error("ebos_gradient_magnitude: scratch too small (%zu < %zu)", scratch_bytes, cost_scratch(K));
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  const Region rg = make_region(h, w, omit);
  double* partials = reinterpret_cast<double*>(scratch);
  const int64_t m = rg.count();
  gradmag_kernel<T><<<dim3(kCostGrid, K), dim3(kCostBlock), 0, s>>>(images, h, w, rg, partials);
  gradmag_finalize_kernel<T><<<dim3(K), dim3(kCostBlock), 0, s>>>(partials, kCostGrid, m, out);
  EBOS_CHECK_LAUNCH("ebos_gradient_magnitude");
  return EBOS_OK;
}

template <typename T>
int gradmag_grad_impl(const T* images, int K, int h, int w, int omit, const T* upstream, T* d_images,
                      ebos_stream_t stream) {
  EBOS_REQUIRE(images && upstream && d_images, "ebos_gradient_magnitude_grad: NULL argument");
  EBOS_REQUIRE(K >= 1 && K <= 65535 && h > 0 && w > 0, "ebos_gradient_magnitude_grad: bad sizes");
  const Region rg = make_region(h, w, omit);
  dim3 grid(stream_grid((int64_t)h * w, kCostBlock, 4096), K);
  gradmag_grad_kernel<T><<<grid, dim3(kCostBlock), 0, as_stream(stream)>>>(images, h, w, rg, upstream, d_images);
  EBOS_CHECK_LAUNCH("ebos_gradient_magnitude_grad");
  return EBOS_OK;
}

}  // namespace
}  // namespace ebos

extern "C" {

size_t ebos_cost_scratch_bytes(int K) { return ebos::cost_scratch(K); }

int64_t ebos_gradient_magnitude_fused_partials(int h, int w) {
  if (h <= 0 || w <= 0) return 0;
  return (int64_t)((h + ebos::kGmTH - 1) / ebos::kGmTH) * ((w + ebos::kGmTW - 1) / ebos::kGmTW);
}

int ebos_gradient_magnitude_fused_f32(const float* image, int h, int w, int omit_boundary, const float* upstream, float* out,
                                      float* d_image, double* partials, int64_t n_partials, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(image && partials && image != d_image, "ebos_gradient_magnitude_fused: NULL image / partials, or d_image aliases image");
  EBOS_REQUIRE(d_image != nullptr || out != nullptr, "ebos_gradient_magnitude_fused: neither a value nor a gradient image asked for");
  EBOS_REQUIRE(h > 0 && w > 0, "ebos_gradient_magnitude_fused: bad sizes");
  const int64_t need = ebos_gradient_magnitude_fused_partials(h, w);
  if (n_partials < need) {
    set_error("ebos_gradient_magnitude_fused: %lld partials given, %lld needed", (long long)n_partials, (long long)need);
    return EBOS_ERR_SCRATCH;
  }
  const Region rg = make_region(h, w, omit_boundary);
  hipStream_t s = as_stream(stream);
  const dim3 grid((w + kGmTW - 1) / kGmTW, (h + kGmTH - 1) / kGmTH);
  hipEvent_t t0, t1;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_GRADMAG_FUSED))  // bench.py --config 3: events stamped with this dispatch's begin / end
    hipExtLaunchKernelGGL(d_image ? gradmag_fused_kernel<true> : gradmag_fused_kernel<false>, grid, dim3(kCostBlock), 0, s, t0, t1, 0, image, h, w,
                          rg, upstream, d_image, partials);
  else if (d_image != nullptr)
    gradmag_fused_kernel<true><<<grid, dim3(kCostBlock), 0, s>>>(image, h, w, rg, upstream, d_image, partials);
  else
    gradmag_fused_kernel<false><<<grid, dim3(kCostBlock), 0, s>>>(image, h, w, rg, upstream, d_image, partials);
  if (out != nullptr) gradmag_fused_finalize_kernel<<<dim3(1), dim3(kCostBlock), 0, s>>>(partials, (int)need, rg.count(), out);
  EBOS_CHECK_LAUNCH("ebos_gradient_magnitude_fused");
  return EBOS_OK;
}

int64_t ebos_blur3_variance_partials(int h, int w) { return ebos_gradient_magnitude_fused_partials(h, w); }

size_t ebos_cmax_cost_scratch_bytes(int h, int w) {
  const size_t tiles = (size_t)ebos_gradient_magnitude_fused_partials(h, w);
  const size_t a = ebos::cost_scratch(1), b = 16 * tiles + 64;  // (the blur's pairs: 16 B per tile; the Sobel pass's values: 8)
  return a > b ? a : b;
}

int ebos_blur3_variance_adjoint_f32(const float* image, int h, int w, int omit_boundary, float k0, float k1, float* z_image,
                                    double* partials, int64_t n_partials, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(image && z_image && partials && image != z_image, "ebos_blur3_variance_adjoint: NULL or aliased image / z_image / partials");
  EBOS_REQUIRE(h >= 2 && w >= 2, "ebos_blur3_variance_adjoint: reflect padding needs at least 2 samples per axis (h=%d w=%d)", h, w);
  EBOS_REQUIRE(k0 > 0.0f && k1 > 0.0f, "ebos_blur3_variance_adjoint: taps must be positive");
  const int64_t need = ebos_blur3_variance_partials(h, w);
  if (n_partials < need) {
    set_error("ebos_blur3_variance_adjoint: %lld partial pairs given, %lld needed", (long long)n_partials, (long long)need);
    return EBOS_ERR_SCRATCH;
  }
  const Region rg = make_region(h, w, omit_boundary);
  const dim3 grid((w + kGmTW - 1) / kGmTW, (h + kGmTH - 1) / kGmTH);
  blur3_variance_adjoint_kernel<<<grid, dim3(kCostBlock), 0, as_stream(stream)>>>(image, h, w, rg, Blur3{k0, k1}, z_image, partials);
  EBOS_CHECK_LAUNCH("ebos_blur3_variance_adjoint");
  return EBOS_OK;
}

int ebos_image_variance_f32(const float* images, int K, int h, int w, int omit_boundary, float* out, double* moments,
                            void* scratch, size_t scratch_bytes, ebos_stream_t stream) {
  return ebos::variance_impl<float>(images, K, h, w, omit_boundary, out, moments, scratch, scratch_bytes, stream);
}
int ebos_image_variance_f64(const double* images, int K, int h, int w, int omit_boundary, double* out,
                            double* moments, void* scratch, size_t scratch_bytes, ebos_stream_t stream) {
  return ebos::variance_impl<double>(images, K, h, w, omit_boundary, out, moments, scratch, scratch_bytes, stream);
}
int ebos_image_variance_grad_f32(const float* images, int K, int h, int w, int omit_boundary, const double* moments,
                                 const float* upstream, float* d_images, ebos_stream_t stream) {
  return ebos::variance_grad_impl<float>(images, K, h, w, omit_boundary, moments, upstream, d_images, stream);
}
int ebos_image_variance_grad_f64(const double* images, int K, int h, int w, int omit_boundary, const double* moments,
                                 const double* upstream, double* d_images, ebos_stream_t stream) {
  return ebos::variance_grad_impl<double>(images, K, h, w, omit_boundary, moments, upstream, d_images, stream);
}
int ebos_image_variance_affine_f32(const double* moments, const float* upstream, int K, float* affine,
                                   ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(moments && upstream && affine && K >= 1, "ebos_image_variance_affine: bad argument");
  variance_affine_kernel<<<dim3((K + 63) / 64), dim3(64), 0, as_stream(stream)>>>(moments, upstream, K, affine);
  EBOS_CHECK_LAUNCH("ebos_image_variance_affine");
  return EBOS_OK;
}
int ebos_gradient_magnitude_f32(const float* images, int K, int h, int w, int omit_boundary, float* out, void* scratch,
                                size_t scratch_bytes, ebos_stream_t stream) {
  return ebos::gradmag_impl<float>(images, K, h, w, omit_boundary, out, scratch, scratch_bytes, stream);
}
int ebos_gradient_magnitude_f64(const double* images, int K, int h, int w, int omit_boundary, double* out,
                                void* scratch, size_t scratch_bytes, ebos_stream_t stream) {
  return ebos::gradmag_impl<double>(images, K, h, w, omit_boundary, out, scratch, scratch_bytes, stream);
}
int ebos_gradient_magnitude_grad_f32(const float* images, int K, int h, int w, int omit_boundary,
                                     const float* upstream, float* d_images, ebos_stream_t stream) {
  return ebos::gradmag_grad_impl<float>(images, K, h, w, omit_boundary, upstream, d_images, stream);
}
int ebos_gradient_magnitude_grad_f64(const double* images, int K, int h, int w, int omit_boundary,
                                     const double* upstream, double* d_images, ebos_stream_t stream) {
  return ebos::gradmag_grad_impl<double>(images, K, h, w, omit_boundary, upstream, d_images, stream);
}

}  // extern "C"
