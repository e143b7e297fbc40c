// flow_upsample.hip -- patch-flow grid -> dense per-pixel flow (and its adjoint) for gfx950.
//
// Restates src/solver/patch_eklt.py:173-204 (under /root/reference) as ONE gather kernel instead
// of pad -> resize -> crop:  replicate-pad the [2, gh, gw] grid by pad = int(patch/2 // slide) + 1,
// bilinear resize by the sliding window with align_corners = False (what torchvision's tensor
// `resize` evaluates), centre-crop to [H, W].  Because the scale is the integer sliding window,
// a dense pixel (r, c) reads at most 2 x 2 grid cells:
//     src = (R + 0.5) / slide - 0.5, clamped at 0;  i0 = floor(src), i1 = min(i0 + 1, n - 1)
//     padded index i -> grid index clamp(i - pad, 0, g - 1)                      (replicate pad)
// with R = r + (n_full / 2 - H / 2) the row in the un-cropped resize output.
// The 2 x 30 x 40 grid of BASELINE config 4 lives in L1/L2; the kernel is a pure 7.4 MB store.
#include "common.h"

namespace ebos {
namespace {

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

__device__ __forceinline__ Lerp lerp_at(const Axis& a, int r) {
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

__global__ void __launch_bounds__(256)
upsample_kernel(const float* __restrict__ grid, Axis ay, Axis ax, int H, int W, float* __restrict__ dense) {
  const int ch = blockIdx.y;
  const float* g = grid + (int64_t)ch * ay.g * ax.g;
  float* out = dense + (int64_t)ch * H * W;
  const int64_t hw = (int64_t)H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i % W);
    const Lerp ly = lerp_at(ay, r), lx = lerp_at(ax, c);
    const float top = lx.w0 * g[ly.i0 * ax.g + lx.i0] + lx.w1 * g[ly.i0 * ax.g + lx.i1];
    const float bot = lx.w0 * g[ly.i1 * ax.g + lx.i0] + lx.w1 * g[ly.i1 * ax.g + lx.i1];
    out[i] = ly.w0 * top + ly.w1 * bot;
  }
}

// adjoint: one workgroup per grid cell gathers every dense pixel whose 2x2 footprint touches it
// (the weights are separable: w(r, i) * w(c, j)), reduces in registers/LDS, one plain add.
__global__ void __launch_bounds__(256)
upsample_bwd_kernel(const float* __restrict__ d_dense, Axis ay, Axis ax, int H, int W, float* d_grid) {
  const int cell = blockIdx.x, ch = blockIdx.y;
  const int gi = cell / ax.g, gj = cell % ax.g;
  const float* dd = d_dense + (int64_t)ch * H * W;
  // conservative pixel ranges: padded indices that clamp to this cell, +-1 cell of bilinear support
  const int pi_lo = gi == 0 ? 0 : gi + ay.pad, pi_hi = gi == ay.g - 1 ? ay.n_in - 1 : gi + ay.pad;
  const int pj_lo = gj == 0 ? 0 : gj + ax.pad, pj_hi = gj == ax.g - 1 ? ax.n_in - 1 : gj + ax.pad;
  int r_lo = (pi_lo - 1) * ay.slide - ay.off - 1, r_hi = (pi_hi + 2) * ay.slide - ay.off + 1;
  int c_lo = (pj_lo - 1) * ax.slide - ax.off - 1, c_hi = (pj_hi + 2) * ax.slide - ax.off + 1;
  r_lo = r_lo < 0 ? 0 : r_lo;
  c_lo = c_lo < 0 ? 0 : c_lo;
  r_hi = r_hi > H ? H : r_hi;
  c_hi = c_hi > W ? W : c_hi;
  const int nr = r_hi - r_lo, nc = c_hi - c_lo;
  float acc = 0.0f;
  if (nr > 0 && nc > 0) {
    for (int i = threadIdx.x; i < nr * nc; i += blockDim.x) {
      const int r = r_lo + i / nc, c = c_lo + i % nc;
      const Lerp ly = lerp_at(ay, r), lx = lerp_at(ax, c);
      const float wy = (ly.i0 == gi ? ly.w0 : 0.0f) + (ly.i1 == gi ? ly.w1 : 0.0f);
      const float wx = (lx.i0 == gj ? lx.w0 : 0.0f) + (lx.i1 == gj ? lx.w1 : 0.0f);
      if (wy != 0.0f && wx != 0.0f) acc += wy * wx * dd[(int64_t)r * W + c];
    }
  }
  __shared__ float red[4];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) d_grid[(int64_t)ch * ay.g * ax.g + cell] += acc;
}

}  // namespace
}  // namespace ebos

extern "C" {

int ebos_upsample_patch_flow_f32(const float* grid, int gh, int gw, int patch_h, int patch_w, int slide_h,
                                 int slide_w, int H, int W, float* dense, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(grid && dense, "ebos_upsample_patch_flow: NULL grid/dense");
  EBOS_REQUIRE(gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0 && H > 0 && W > 0,
               "ebos_upsample_patch_flow: bad sizes");
  const Axis ay = make_axis(gh, patch_h, slide_h, H), ax = make_axis(gw, patch_w, slide_w, W);
  EBOS_REQUIRE(ay.off >= 0 && ax.off >= 0 && ay.off + H <= ay.n_in * slide_h && ax.off + W <= ax.n_in * slide_w,
               "ebos_upsample_patch_flow: image %dx%d larger than the resized grid %dx%d", H, W, ay.n_in * slide_h,
               ax.n_in * slide_w);
  dim3 g(stream_grid((int64_t)H * W, 256, 2048), 2);
  upsample_kernel<<<g, dim3(256), 0, as_stream(stream)>>>(grid, ay, ax, H, W, dense);
  EBOS_CHECK_LAUNCH("ebos_upsample_patch_flow");
  return EBOS_OK;
}

int ebos_upsample_patch_flow_bwd_f32(const float* d_dense, int gh, int gw, int patch_h, int patch_w, int slide_h,
                                     int slide_w, int H, int W, float* d_grid, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(d_dense && d_grid, "ebos_upsample_patch_flow_bwd: NULL d_dense/d_grid");
  EBOS_REQUIRE(gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0 && H > 0 && W > 0,
               "ebos_upsample_patch_flow_bwd: bad sizes");
  const Axis ay = make_axis(gh, patch_h, slide_h, H), ax = make_axis(gw, patch_w, slide_w, W);
  EBOS_REQUIRE(ay.off >= 0 && ax.off >= 0, "ebos_upsample_patch_flow_bwd: image larger than the resized grid");
  dim3 g(gh * gw, 2);
  upsample_bwd_kernel<<<g, dim3(256), 0, as_stream(stream)>>>(d_dense, ay, ax, H, W, d_grid);
  EBOS_CHECK_LAUNCH("ebos_upsample_patch_flow_bwd");
  return EBOS_OK;
}

}  // extern "C"
