// iwe_fused.hip -- the hot path: fused per-event warp + bilinear-splat IWE (and its backward)
// for gfx950 / CDNA4.  Nothing per-event is materialised: an event is read once (SoA f32
// x, y, dt [, weight] = 12-16 B), warped in registers and its four taps are accumulated.
//
//   forward  = src/warp.py:330-342 (dense) | :364-383 (2-DoF)  +  src/event_image_converter.py:581-620
//   backward = autograd of the above w.r.t. flow / theta / weight
//
// Two forward organisations:
//   * iwe_dense_kernel        any event order, four global float atomics per event (memory-side
//                             atomics: ~20 G lane-atomics/s on MI355X -> the slow, general path);
//   * iwe_dense_tiled_kernel  events binned by source tile (event_plan.hip).  One workgroup owns one
//                             (tile, split): the IWE of the tile + HALO pixels per side lives in LDS
//                             (ds_add_f32), is accumulated there and flushed once with row-contiguous
//                             global atomics (256 B per wave instruction, the shape the memory-side
//                             atomic unit wants).  Taps beyond the halo go straight to global atomics,
//                             so any flow magnitude stays correct.
// All kernels evaluate the warp in source-pixel-relative coordinates (warped_taps below).
#include <type_traits>

#include "common.h"

namespace ebos {
namespace {

constexpr float kEps = 1e-6f;  // src/event_image_converter.py:586

// Warped footprint in SOURCE-PIXEL-RELATIVE coordinates.  x' = x + d with d = -dt * u.  Writing
// x = rs + fx (rs = trunc(x), fx exact) gives floor(x' + eps) = rs + floor(fx + d + eps): the float
// arithmetic only ever sees |fx + d| <~ 32, so its rounding error is ~2e-6 px instead of the ~1e-4 px
// of absolute 1280-px coordinates.  That matters because the bilinear splat's derivative jumps at
// integer coordinates: every event rounded across an integer flips its gradient contribution.
struct Taps {
  int R, C;      // top-left tap, padded image coordinates
  float fr, fc;  // fractional offsets
  bool ok;       // finite
};
__device__ __forceinline__ Taps warped_taps(float ex, float ey, float dx, float dy, int pad_h, int pad_w) {
  const int rs = (int)ex, cs = (int)ey;
  const float lx = (ex - (float)rs) + dx, ly = (ey - (float)cs) + dy;
  const float r0 = floorf(lx + kEps), c0 = floorf(ly + kEps);
  Taps t;
  t.fr = lx - r0;
  t.fc = ly - c0;
  t.ok = (r0 > -1e9f) && (r0 < 1e9f) && (c0 > -1e9f) && (c0 < 1e9f);
  t.R = t.ok ? rs + (int)r0 + pad_h : -4;
  t.C = t.ok ? cs + (int)c0 + pad_w : -4;
  return t;
}

// ---------------------------------------------------------------------------------------------
// general forward: global atomics
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
iwe_dense_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                 const float* __restrict__ weight, int64_t n, const float* __restrict__ flow, int H, int W,
                 int row_stride, int pad_h, int pad_w, float* iwe) {
  const int64_t hw = (int64_t)H * W;
  const int h = H + 2 * pad_h, w = W + 2 * pad_w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float ex = x[i], ey = y[i], edt = dt[i];
    if (!(ex > -1e9f && ex < 1e9f && ey > -1e9f && ey < 1e9f)) continue;
    const int64_t lin = (int64_t)(int)ex * row_stride + (int)ey;
    if (lin < 0 || lin >= hw) continue;  // torch.gather would raise (src/warp.py:334-336): dropped
    const Taps f = warped_taps(ex, ey, -edt * flow[lin], -edt * flow[hw + lin], pad_h, pad_w);
    const float wv = weight ? weight[i] : 1.0f;
    const bool r0 = f.R >= 0 && f.R < h, r1 = f.R + 1 >= 0 && f.R + 1 < h;
    const bool c0 = f.C >= 0 && f.C < w, c1 = f.C + 1 >= 0 && f.C + 1 < w;
    const int64_t base = (int64_t)f.R * w + f.C;
    const float a = 1.0f - f.fr, b = 1.0f - f.fc;
    if (r0 && c0) atomic_add(&iwe[base], a * b * wv);
    if (r1 && c0) atomic_add(&iwe[base + w], f.fr * b * wv);
    if (r0 && c1) atomic_add(&iwe[base + 1], a * f.fc * wv);
    if (r1 && c1) atomic_add(&iwe[base + w + 1], f.fr * f.fc * wv);
  }
}

// ---------------------------------------------------------------------------------------------
// tiled forward: LDS-privatised IWE tile per workgroup
// ---------------------------------------------------------------------------------------------
constexpr int kTiledBlock = 1024;

// ACC = accumulator type of the LDS tile.  Measured on MI355X (tools/ubench_lds_atomics.hip):
// ds_add_f32 sustains only ~0.33 lanes/clk/CU (200 Gop/s chip-wide, any bank pattern), ds_add_f64
// ~2.8 lanes/clk/CU (1.7 Top/s), ds_add_u32 ~6-8.  So the tile is accumulated in f64 whenever
// tile + halo fits the 160 KiB LDS in doubles; f32 remains only for the largest halos.
template <int TH, int TW, int HALO, typename ACC>
__global__ void __launch_bounds__(kTiledBlock)
iwe_dense_tiled_kernel(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ dts,
                       const float* __restrict__ weight, const int32_t* __restrict__ key_offsets,
                       const float* __restrict__ flow, int H, int W, int tiles_x, int splits, int pad_h, int pad_w,
                       float* iwe) {
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  extern __shared__ double s_raw[];  // [LH][LW] of ACC
  ACC* s_img = reinterpret_cast<ACC*>(s_raw);

  const int tile = blockIdx.x / splits, part = blockIdx.x - tile * splits;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int32_t beg = key_offsets[tile * (TH * TW)], end = key_offsets[(tile + 1) * (TH * TW)];
  if (beg == end) return;
  int32_t chunk = (end - beg + splits - 1) / splits;
  chunk = (chunk + kWave - 1) & ~(kWave - 1);
  const int32_t my_beg = beg + part * chunk;
  const int32_t my_end = min(end, my_beg + chunk);
  if (my_beg >= my_end) return;

  for (int i = threadIdx.x; i < LH * LW; i += kTiledBlock) s_img[i] = ACC(0);
  __syncthreads();

  const int64_t hw = (int64_t)H * W;
  const int h = H + 2 * pad_h, w = W + 2 * pad_w;
  // LDS cell (0,0) <-> un-padded image pixel (oy, ox); padded pixel (oy + pad_h, ox + pad_w)
  const int oy = ty * TH - HALO, ox = tx * TW - HALO;

  // Each thread keeps kUnroll events in flight: all coalesced SoA loads are issued first, then the
  // flow gathers, then the LDS atomics -- an in-order wave otherwise serialises
  // load -> gather -> atomic once per event and the kernel becomes latency-bound.
  constexpr int kUnroll = 8;
  for (int32_t base = my_beg + threadIdx.x; base < my_end; base += kTiledBlock * kUnroll) {
    float ex[kUnroll], ey[kUnroll], edt[kUnroll], wv[kUnroll], fu[kUnroll], fv[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int32_t i = base + k * kTiledBlock;
      const bool live = i < my_end;
      ex[k] = live ? xs[i] : -1.0f;  // -1 marks a dead slot (binned events have x >= 0 ... or trunc to 0)
      ey[k] = live ? ys[i] : 0.0f;
      edt[k] = live ? dts[i] : 0.0f;
      wv[k] = (live && weight) ? weight[i] : 1.0f;
      if (!live) wv[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int32_t i = base + k * kTiledBlock;
      const int64_t lin = (int64_t)(int)ex[k] * W + (int)ey[k];  // binned events have a valid source pixel
      const bool live = i < my_end;
      fu[k] = live ? flow[lin] : 0.0f;
      fv[k] = live ? flow[hw + lin] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int32_t i = base + k * kTiledBlock;
      if (i >= my_end) break;
      const Taps f = warped_taps(ex[k], ey[k], -edt[k] * fu[k], -edt[k] * fv[k], 0, 0);  // un-padded coordinates
      const float fr = f.fr, fc = f.fc;
      const float a = 1.0f - fr, b = 1.0f - fc;
      const float w00 = a * b * wv[k], w10 = fr * b * wv[k], w01 = a * fc * wv[k], w11 = fr * fc * wv[k];
      const int rl = f.R - oy, cl = f.C - ox;  // LDS cell of the top-left tap
      if (f.ok && rl >= 0 && rl < LH - 1 && cl >= 0 && cl < LW - 1) {
        ACC* p = &s_img[rl * LW + cl];
        atomic_add(p, (ACC)w00);
        atomic_add(p + LW, (ACC)w10);
        atomic_add(p + 1, (ACC)w01);
        atomic_add(p + LW + 1, (ACC)w11);
      } else if (f.ok) {
        // beyond the halo: straight to the image (rare when halo >= max |dt * flow| + 1)
        const int R = f.R + pad_h, C = f.C + pad_w;
        const bool rr0 = R >= 0 && R < h, rr1 = R + 1 >= 0 && R + 1 < h;
        const bool cc0 = C >= 0 && C < w, cc1 = C + 1 >= 0 && C + 1 < w;
        const int64_t gb = (int64_t)R * w + C;
        if (rr0 && cc0) atomic_add(&iwe[gb], w00);
        if (rr1 && cc0) atomic_add(&iwe[gb + w], w10);
        if (rr0 && cc1) atomic_add(&iwe[gb + 1], w01);
        if (rr1 && cc1) atomic_add(&iwe[gb + w + 1], w11);
      }
    }
  }
  __syncthreads();

  // flush: consecutive lanes -> consecutive columns of one image row
  const int gy0 = oy + pad_h, gx0 = ox + pad_w;
  for (int i = threadIdx.x; i < LH * LW; i += kTiledBlock) {
    const float v = (float)s_img[i];
    if (v == 0.0f) continue;
    const int rl = i / LW, cl = i - rl * LW;
    const int R = gy0 + rl, C = gx0 + cl;
    if (R >= 0 && R < h && C >= 0 && C < w) atomic_add(&iwe[(int64_t)R * w + C], v);
  }
}

struct TiledConfig {
  int th, tw, halo;
};
constexpr TiledConfig kTiledConfigs[] = {{64, 64, 32}, {32, 64, 32}, {32, 32, 32}, {64, 64, 16}, {32, 32, 16},
                                         {32, 32, 8},  {64, 64, 64}, {32, 64, 48}, {16, 64, 32}};
constexpr int kNumTiledConfigs = sizeof(kTiledConfigs) / sizeof(kTiledConfigs[0]);

template <int TH, int TW, int HALO>
struct TileAcc {  // f64 when it fits the LDS, else f32
  static constexpr bool kF64 = (size_t)(TH + 2 * HALO) * (TW + 2 * HALO) * sizeof(double) <= 160 * 1024;
  using type = typename std::conditional<kF64, double, float>::type;
};

template <int TH, int TW, int HALO>
int launch_tiled(const float* xs, const float* ys, const float* dts, const float* weight, const int32_t* key_offsets,
                 const float* flow, int H, int W, int splits, int pad_h, int pad_w, float* iwe, hipStream_t s) {
  constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
  using ACC = typename TileAcc<TH, TW, HALO>::type;
  constexpr size_t lds = (size_t)LH * LW * sizeof(ACC);
  static_assert(lds <= 160 * 1024, "tile + halo must fit the 160 KiB LDS of a CDNA4 CU");
  const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
  auto kern = iwe_dense_tiled_kernel<TH, TW, HALO, ACC>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      set_error("ebos_iwe_dense_tiled: cannot reserve %zu B of LDS", lds);
      return EBOS_ERR_LAUNCH;
    }
  }
  kern<<<dim3((unsigned)(tiles_y * tiles_x * splits)), dim3(kTiledBlock), lds, s>>>(
      xs, ys, dts, weight, key_offsets, flow, H, W, tiles_x, splits, pad_h, pad_w, iwe);
  return EBOS_OK;
}

// ---------------------------------------------------------------------------------------------
// backward, dense flow
// ---------------------------------------------------------------------------------------------
struct GradImage {
  const float* g;
  float a, c;  // G = a * g + c inside the valid region
  int h, w, lo;
  __device__ __forceinline__ float at(int R, int C) const {
    if (R < lo || R >= h - lo || C < lo || C >= w - lo) return 0.0f;
    return a * g[(int64_t)R * w + C] + c;
  }
};

template <bool SORTED>
__global__ void __launch_bounds__(256)
iwe_dense_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                     const float* __restrict__ weight, int64_t n, const float* __restrict__ flow, int H, int W,
                     int row_stride, int pad_h, int pad_w, const float* __restrict__ g_image,
                     const float* __restrict__ affine, int g_lo, float* d_flow, float* __restrict__ d_weight) {
  const int64_t hw = (int64_t)H * W;
  GradImage G;
  G.g = g_image;
  G.a = affine ? affine[0] : 1.0f;
  G.c = affine ? affine[1] : 0.0f;
  G.h = H + 2 * pad_h;
  G.w = W + 2 * pad_w;
  G.lo = g_lo;
  const int lane = threadIdx.x & (kWave - 1);
  // whole waves iterate together so that the shuffles below see all 64 lanes
  const int64_t n_round = (n + kWave - 1) / kWave * kWave;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_round; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t lin = -1;
    float gx = 0.0f, gy = 0.0f;
    if (i < n) {
      const float ex = x[i], ey = y[i], edt = dt[i];
      if (ex > -1e9f && ex < 1e9f && ey > -1e9f && ey < 1e9f) {
        lin = (int64_t)(int)ex * row_stride + (int)ey;
        if (lin < 0 || lin >= hw) lin = -1;
      }
      if (lin >= 0) {
        const Taps f = warped_taps(ex, ey, -edt * flow[lin], -edt * flow[hw + lin], pad_h, pad_w);
        const float g00 = G.at(f.R, f.C), g10 = G.at(f.R + 1, f.C);
        const float g01 = G.at(f.R, f.C + 1), g11 = G.at(f.R + 1, f.C + 1);
        const float wv = weight ? weight[i] : 1.0f;
        const float a = 1.0f - f.fr, b = 1.0f - f.fc;
        const float dx = wv * (b * (g10 - g00) + f.fc * (g11 - g01));  // dL/dx'
        const float dy = wv * (a * (g01 - g00) + f.fr * (g11 - g10));  // dL/dy'
        gx = -edt * dx;                                                // dL/dflow0[src]
        gy = -edt * dy;
        if (d_weight) d_weight[i] = a * b * g00 + f.fr * b * g10 + a * f.fc * g01 + f.fr * f.fc * g11;
      } else if (d_weight) {
        d_weight[i] = 0.0f;
      }
    }
    if (SORTED) {
      // events of one source pixel are contiguous: segmented wave reduction, one atomic per run
      const int64_t prev = __shfl_up(lin, 1, kWave);
      const bool head = (lane == 0) || (prev != lin);
      const unsigned long long heads = __ballot(head);
      // run id = number of heads at or below this lane
      const int run = __popcll(heads & (~0ull >> (63 - lane)));
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const float ox_ = __shfl_down(gx, off, kWave);
        const float oy_ = __shfl_down(gy, off, kWave);
        const int orun = __shfl_down(run, off, kWave);
        if (lane + off < kWave && orun == run) {
          gx += ox_;
          gy += oy_;
        }
      }
      if (head && lin >= 0) {
        atomic_add(&d_flow[lin], gx);
        atomic_add(&d_flow[hw + lin], gy);
      }
    } else if (lin >= 0) {
      atomic_add(&d_flow[lin], gx);
      atomic_add(&d_flow[hw + lin], gy);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 2-DoF hypotheses: K translations per pass over the events
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
iwe_2dof_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                const float* __restrict__ weight, int64_t n, const float* __restrict__ thetas, int K, int h, int w,
                int pad_h, int pad_w, float* iwes) {
  const int64_t hw = (int64_t)h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float ex = x[i], ey = y[i], edt = dt[i];
    const float wv = weight ? weight[i] : 1.0f;
    for (int k = blockIdx.y; k < K; k += gridDim.y) {
      // plus sign: src/warp.py:368-375
      const Taps f = warped_taps(ex, ey, edt * thetas[2 * k], edt * thetas[2 * k + 1], pad_h, pad_w);
      float* img = iwes + k * hw;
      const bool r0 = f.R >= 0 && f.R < h, r1 = f.R + 1 >= 0 && f.R + 1 < h;
      const bool c0 = f.C >= 0 && f.C < w, c1 = f.C + 1 >= 0 && f.C + 1 < w;
      const int64_t base = (int64_t)f.R * w + f.C;
      const float a = 1.0f - f.fr, b = 1.0f - f.fc;
      if (r0 && c0) atomic_add(&img[base], a * b * wv);
      if (r1 && c0) atomic_add(&img[base + w], f.fr * b * wv);
      if (r0 && c1) atomic_add(&img[base + 1], a * f.fc * wv);
      if (r1 && c1) atomic_add(&img[base + w + 1], f.fr * f.fc * wv);
    }
  }
}

__global__ void __launch_bounds__(256)
iwe_2dof_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dt,
                    const float* __restrict__ weight, int64_t n, const float* __restrict__ thetas, int K, int h, int w,
                    int pad_h, int pad_w, const float* __restrict__ g_images, const float* __restrict__ affine,
                    int g_lo, float* d_thetas) {
  __shared__ float red[4];
  const int k = blockIdx.y;
  GradImage G;
  G.g = g_images + (int64_t)k * h * w;
  G.a = affine ? affine[2 * k] : 1.0f;
  G.c = affine ? affine[2 * k + 1] : 0.0f;
  G.h = h;
  G.w = w;
  G.lo = g_lo;
  const float th0 = thetas[2 * k], th1 = thetas[2 * k + 1];
  float a0 = 0.0f, a1 = 0.0f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float ex = x[i], ey = y[i], edt = dt[i];
    const Taps f = warped_taps(ex, ey, edt * th0, edt * th1, pad_h, pad_w);
    const float g00 = G.at(f.R, f.C), g10 = G.at(f.R + 1, f.C);
    const float g01 = G.at(f.R, f.C + 1), g11 = G.at(f.R + 1, f.C + 1);
    const float wv = weight ? weight[i] : 1.0f;
    const float a = 1.0f - f.fr, b = 1.0f - f.fc;
    a0 += edt * wv * (b * (g10 - g00) + f.fc * (g11 - g01));
    a1 += edt * wv * (a * (g01 - g00) + f.fr * (g11 - g10));
  }
  a0 = block_sum(a0, red);
  a1 = block_sum(a1, red);
  if (threadIdx.x == 0) {
    atomic_add(&d_thetas[2 * k], a0);
    atomic_add(&d_thetas[2 * k + 1], a1);
  }
}

}  // namespace
}  // namespace ebos

extern "C" {

int ebos_iwe_dense_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                       const float* flow, int H, int W, int row_stride, int pad_h, int pad_w, float* iwe,
                       ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(flow && iwe, "ebos_iwe_dense: NULL flow/iwe");
  EBOS_REQUIRE((x && y && dt) || n == 0, "ebos_iwe_dense: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && row_stride > 0 && pad_h >= 0 && pad_w >= 0, "ebos_iwe_dense: bad sizes");
  if (n == 0) return EBOS_OK;
  iwe_dense_kernel<<<dim3(stream_grid(n, 256)), dim3(256), 0, as_stream(stream)>>>(x, y, dt, weight, n, flow, H, W,
                                                                                   row_stride, pad_h, pad_w, iwe);
  EBOS_CHECK_LAUNCH("ebos_iwe_dense");
  return EBOS_OK;
}

int ebos_tiled_config(int* out, int cap) {
  using namespace ebos;
  for (int i = 0; i < kNumTiledConfigs && i < cap && out != nullptr; ++i) {
    out[3 * i] = kTiledConfigs[i].th;
    out[3 * i + 1] = kTiledConfigs[i].tw;
    out[3 * i + 2] = kTiledConfigs[i].halo;
  }
  return kNumTiledConfigs;
}

int ebos_iwe_dense_tiled_f32(const float* xs, const float* ys, const float* dts, const float* weight,
                             const int32_t* key_offsets, int64_t n, const float* flow, int H, int W, int tile_h,
                             int tile_w, int halo, int splits, int pad_h, int pad_w, float* iwe,
                             ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(flow && iwe && key_offsets, "ebos_iwe_dense_tiled: NULL flow/iwe/key_offsets");
  EBOS_REQUIRE((xs && ys && dts) || n == 0, "ebos_iwe_dense_tiled: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0 && splits >= 1 && splits <= 64,
               "ebos_iwe_dense_tiled: bad sizes (splits=%d)", splits);
  if (n == 0) return EBOS_OK;
  hipStream_t s = as_stream(stream);
  int rc = EBOS_ERR_UNSUPPORTED;
#define EBOS_TILED_CASE(TH, TW, HL)                                                                             \
  if (tile_h == TH && tile_w == TW && halo == HL)                                                              \
    rc = launch_tiled<TH, TW, HL>(xs, ys, dts, weight, key_offsets, flow, H, W, splits, pad_h, pad_w, iwe, s);
  EBOS_TILED_CASE(64, 64, 32)
  EBOS_TILED_CASE(32, 64, 32)
  EBOS_TILED_CASE(32, 32, 32)
  EBOS_TILED_CASE(16, 64, 32)
  EBOS_TILED_CASE(64, 64, 16)
  EBOS_TILED_CASE(32, 32, 16)
  EBOS_TILED_CASE(32, 32, 8)
  EBOS_TILED_CASE(64, 64, 64)
  EBOS_TILED_CASE(32, 64, 48)
#undef EBOS_TILED_CASE
  if (rc == EBOS_ERR_UNSUPPORTED) {
    set_error("ebos_iwe_dense_tiled: no kernel built for tile %dx%d halo %d (see ebos_tiled_config)", tile_h, tile_w, halo);
    return rc;
  }
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_iwe_dense_tiled");
  return EBOS_OK;
}

int ebos_iwe_dense_bwd_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                           const float* flow, int H, int W, int row_stride, int pad_h, int pad_w,
                           const float* g_image, const float* affine, int g_lo, int sorted, float* d_flow,
                           float* d_weight, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(flow && g_image && d_flow, "ebos_iwe_dense_bwd: NULL flow/g_image/d_flow");
  EBOS_REQUIRE((x && y && dt) || n == 0, "ebos_iwe_dense_bwd: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && H > 0 && W > 0 && row_stride > 0 && pad_h >= 0 && pad_w >= 0 && g_lo >= 0,
               "ebos_iwe_dense_bwd: bad sizes");
  if (n == 0) return EBOS_OK;
  dim3 grid(stream_grid(n, 256)), block(256);
  hipStream_t s = as_stream(stream);
  if (sorted)
    iwe_dense_bwd_kernel<true><<<grid, block, 0, s>>>(x, y, dt, weight, n, flow, H, W, row_stride, pad_h, pad_w, g_image,
                                                      affine, g_lo, d_flow, d_weight);
  else
    iwe_dense_bwd_kernel<false><<<grid, block, 0, s>>>(x, y, dt, weight, n, flow, H, W, row_stride, pad_h, pad_w,
                                                       g_image, affine, g_lo, d_flow, d_weight);
  EBOS_CHECK_LAUNCH("ebos_iwe_dense_bwd");
  return EBOS_OK;
}

int ebos_iwe_2dof_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                      const float* thetas, int K, int h, int w, int pad_h, int pad_w, float* iwes,
                      ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(thetas && iwes, "ebos_iwe_2dof: NULL thetas/iwes");
  EBOS_REQUIRE((x && y && dt) || n == 0, "ebos_iwe_2dof: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && K >= 1 && h > 0 && w > 0 && pad_h >= 0 && pad_w >= 0, "ebos_iwe_2dof: bad sizes");
  if (n == 0) return EBOS_OK;
  const int gy = K < 8 ? K : 8;
  dim3 grid(stream_grid(n, 256, 256 * 8 / gy), gy);
  iwe_2dof_kernel<<<grid, dim3(256), 0, as_stream(stream)>>>(x, y, dt, weight, n, thetas, K, h, w, pad_h, pad_w, iwes);
  EBOS_CHECK_LAUNCH("ebos_iwe_2dof");
  return EBOS_OK;
}

int ebos_iwe_2dof_bwd_f32(const float* x, const float* y, const float* dt, const float* weight, int64_t n,
                          const float* thetas, int K, int h, int w, int pad_h, int pad_w, const float* g_images,
                          const float* affine, int g_lo, float* d_thetas, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(thetas && g_images && d_thetas, "ebos_iwe_2dof_bwd: NULL thetas/g_images/d_thetas");
  EBOS_REQUIRE((x && y && dt) || n == 0, "ebos_iwe_2dof_bwd: NULL event buffer");
  EBOS_REQUIRE(n >= 0 && K >= 1 && K <= 65535 && h > 0 && w > 0 && pad_h >= 0 && pad_w >= 0 && g_lo >= 0,
               "ebos_iwe_2dof_bwd: bad sizes");
  if (n == 0) return EBOS_OK;
  dim3 grid(stream_grid(n, 256, 512), K);
  iwe_2dof_bwd_kernel<<<grid, dim3(256), 0, as_stream(stream)>>>(x, y, dt, weight, n, thetas, K, h, w, pad_h, pad_w,
                                                                 g_images, affine, g_lo, d_thetas);
  EBOS_CHECK_LAUNCH("ebos_iwe_2dof_bwd");
  return EBOS_OK;
}

}  // extern "C"
