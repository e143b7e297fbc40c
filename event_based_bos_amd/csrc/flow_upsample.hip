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
#include <math.h>

#include "common.h"
#include "patch_grid.h"

namespace ebos {
namespace {

// forward: one thread = 4 consecutive columns (one 16-byte store per row) of kUpRows consecutive rows: the horizontal
// lerps are computed once per thread, the vertical one is uniform per row; the grid (a few KB) is read through L1.
constexpr int kUpRows = 4;

template <bool VEC4>
__global__ void __launch_bounds__(256)
upsample_kernel(const float* __restrict__ grid, Axis ay, Axis ax, int H, int W, float* __restrict__ dense) {
  const int ch = blockIdx.z, r0 = blockIdx.y * kUpRows;
  const int c0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c0 >= W) return;
  const float* g = grid + (int64_t)ch * ay.g * ax.g;
  Lerp lx[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) lx[k] = lerp_at(ax, c0 + k < W ? c0 + k : W - 1);
#pragma unroll
  for (int j = 0; j < kUpRows; ++j) {
    const int r = r0 + j;
    if (r >= H) break;
    float* out = dense + (int64_t)ch * H * W + (int64_t)r * W + c0;
    const Lerp ly = lerp_at(ay, r);
    const float* g0 = g + ly.i0 * ax.g;
    const float* g1 = g + ly.i1 * ax.g;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = grid_bilerp(g0, g1, ly, lx[k]);
    }
    if (VEC4) {
      *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + k < W) out[k] = v[k];
    }
  }
}

// adjoint, separable (the bilinear weight of pixel (r, c) on cell (i, j) is wy(r, i) * wx(c, j)):
//   pass 1  S[ch, i, c]      = sum_r wy(r, i) * d_dense[ch, r, c]     coalesced row reads
//   pass 2  d_grid[ch, i, j] = sum_c wx(c, j) * S[ch, i, c]
// Rows/columns outside a cell's conservative support are skipped; every pixel is read twice (cells i and i + 1).
// pass 1: workgroup = 64 columns x 4 row phases (wave k takes rows r_lo + k, r_lo + k + 4, ...), lane = column
__global__ void __launch_bounds__(256)
upsample_bwd_rows_kernel(const float* __restrict__ d_dense, Axis ay, int H, int W, float* __restrict__ S) {
  const int gi = blockIdx.y, ch = blockIdx.z;
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  int r_lo, r_hi;
  support(ay, gi, H, &r_lo, &r_hi);
  float acc = 0.0f;
  if (c < W) {
    const float* dd = d_dense + (int64_t)ch * H * W + c;
#pragma unroll 4
    for (int r = r_lo + phase; r < r_hi; r += 4) acc += weight_on(ay, r, gi) * dd[(int64_t)r * W];  // weight uniform per wave
  }
  __shared__ float part[4][64];
  part[phase][lane] = acc;
  __syncthreads();
  if (phase == 0 && c < W) S[((int64_t)ch * ay.g + gi) * W + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// Adam is element-wise: the wavefront that has just produced d loss / d theta[i] can apply the update of theta[i] on
// the spot -- the optimiser step of the solver loop costs no launch of its own (torch.optim.Adam semantics, amsgrad off,
// no weight decay; the bias corrections of step t are computed on the host in double, as torch does).  One wavefront
// also records the loss of the iteration.
struct AdamJob {
  float *theta, *exp_avg, *exp_avg_sq;  // theta == nullptr: plain adjoint
  float step_size, bc2_sqrt, beta2, w1, w2, eps;
  int t;
  int* step;
  const float* contrast;
  float contrast_scale;
  const double* reg_partials;
  int n_reg;
  float* losses;
  int losses_cap;
  const float* grad_mask;  // [gh, gw], nullable: patches whose flow is not estimated (too few events) keep their value
};

// loss of this iteration (parameters BEFORE the update) = contrast_scale * contrast + sum(regulariser partials); one wavefront
__device__ __forceinline__ void adam_record_loss(const AdamJob& job, int lane) {
  double reg = 0.0;
  for (int i = lane; i < job.n_reg; i += 64) reg += job.reg_partials[i];
  reg = wave_sum(reg);
  if (lane == 0) {
    if (job.losses != nullptr && job.t - 1 < job.losses_cap)
      job.losses[job.t - 1] = (float)((double)job.contrast_scale * (double)(job.contrast ? job.contrast[0] : 0.0f) + reg);
    job.step[0] = job.t;
  }
}

// The optimiser state of gradient element i, loaded BEFORE the gradient is known: the loads of (mask, exp_avg, exp_avg_sq, theta)
// then fly beside those of the gradient's terms instead of queueing behind the d_grid store (which may alias them as far as
// the compiler knows) -- one memory round trip per element instead of two in a kernel that is nothing but latency.
struct AdamState {
  float mask, m, v, th;
};
__device__ __forceinline__ AdamState adam_load(const AdamJob& job, int64_t i, int64_t cell) {
  AdamState a{1.0f, 0.0f, 0.0f, 0.0f};
  if (job.grad_mask != nullptr) a.mask = job.grad_mask[cell];
  if (job.theta != nullptr) a.m = job.exp_avg[i], a.v = job.exp_avg_sq[i], a.th = job.theta[i];
  return a;
}
// gradient element i (grid cell `cell` of one flow component): mask, store, Adam step
__device__ __forceinline__ void adam_apply(const AdamJob& job, int64_t i, const AdamState& a, float g, float* __restrict__ d_grid) {
  if (job.grad_mask != nullptr) g *= a.mask;
  d_grid[i] = g;
  if (job.theta != nullptr) {
    float m = a.m, v = a.v, th = a.th;
    adam_update(g, m, v, th, job.step_size, job.bc2_sqrt, job.beta2, job.w1, job.w2, job.eps);  // (patch_grid.h: one arithmetic for every kernel)
    job.exp_avg[i] = m;
    job.exp_avg_sq[i] = v;
    job.theta[i] = th;
  }
}
__device__ __forceinline__ void adam_apply(const AdamJob& job, int64_t i, int64_t cell, float g, float* __restrict__ d_grid) {
  adam_apply(job, i, adam_load(job, i, cell), g, d_grid);
}

// Partial cell gradients of the GRID backward kernel (iwe_tiled.hip) -> d_grid (+ Adam): one thread per grid element sums
// the <= 3 x 3 source tiles (and their work items) whose cell block contains it.
__global__ void __launch_bounds__(256)
patch_grad_combine_kernel(const float* __restrict__ partials, const int32_t* __restrict__ part_off, int th, int tw, int tiles_x,
                          int H, int W, Axis ay, Axis ax, float* __restrict__ d_grid, AdamJob job) {
  if (blockIdx.x == gridDim.x - 1) {  // one extra workgroup only records the loss: its serial partials read overlaps the rest
    if (job.theta != nullptr && threadIdx.x < 64) adam_record_loss(job, threadIdx.x);
    return;
  }
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 2 * ay.g * ax.g) return;
  const int ch = idx / (ay.g * ax.g), cell = idx - ch * (ay.g * ax.g);
  const int gi = cell / ax.g, gj = cell - gi * ax.g;
  const AdamState st = adam_load(job, idx, cell);  // (in flight beside the partials)
  // candidate tiles per axis (<= kSpan, the conservative pixel support of the cell) and the first cell of each one's block
  constexpr int kSpan = 4;
  int r_lo, r_hi, c_lo, c_hi;
  support(ay, gi, H, &r_lo, &r_hi);
  support(ax, gj, W, &c_lo, &c_hi);
  const int ty_a = r_lo / th, tx_a = c_lo / tw;
  const int ty_n = r_lo < r_hi ? (r_hi - 1) / th - ty_a + 1 : 0, tx_n = c_lo < c_hi ? (c_hi - 1) / tw - tx_a + 1 : 0;
  float acc = 0.0f;
  if (ty_n <= kSpan && tx_n <= kSpan && part_off == nullptr) {
    // all loads unconditional (clamped), validity as a 0 / 1 factor: the <= 16 reads are in flight together
    int li[kSpan], lj[kSpan], tyv[kSpan], txv[kSpan];
    bool oky[kSpan], okx[kSpan];
    const int tiles_y = (H + th - 1) / th;
#pragma unroll
    for (int k = 0; k < kSpan; ++k) {
      const int ty = min(ty_a + k, tiles_y - 1), tx = min(tx_a + k, tiles_x - 1);
      const int gi0 = lerp_at(ay, ty * th).i0, gi1 = lerp_at(ay, min(ty * th + th, H) - 1).i1;
      const int gj0 = lerp_at(ax, tx * tw).i0, gj1 = lerp_at(ax, min(tx * tw + tw, W) - 1).i1;
      oky[k] = k < ty_n && gi >= gi0 && gi <= gi1;
      okx[k] = k < tx_n && gj >= gj0 && gj <= gj1;
      li[k] = oky[k] ? gi - gi0 : 0;
      lj[k] = okx[k] ? gj - gj0 : 0;
      tyv[k] = ty;
      txv[k] = tx;
    }
    float v[kSpan][kSpan];
#pragma unroll
    for (int a = 0; a < kSpan; ++a)
#pragma unroll
      for (int b = 0; b < kSpan; ++b)
        v[a][b] = partials[(((int64_t)(tyv[a] * tiles_x + txv[b]) * 2 + ch) * kGridCells + li[a]) * kGridCells + lj[b]];
#pragma unroll
    for (int a = 0; a < kSpan; ++a)
#pragma unroll
      for (int b = 0; b < kSpan; ++b) acc += (oky[a] && okx[b]) ? v[a][b] : 0.0f;
  } else if (r_lo < r_hi && c_lo < c_hi) {  // adaptive work items, or a cell that spans many tiles: plain loops
    for (int ty = r_lo / th; ty <= (r_hi - 1) / th; ++ty) {
      const int gi0 = lerp_at(ay, ty * th).i0, gi1 = lerp_at(ay, min(ty * th + th, H) - 1).i1;
      if (gi < gi0 || gi > gi1) continue;
      for (int tx = c_lo / tw; tx <= (c_hi - 1) / tw; ++tx) {
        const int gj0 = lerp_at(ax, tx * tw).i0, gj1 = lerp_at(ax, min(tx * tw + tw, W) - 1).i1;
        if (gj < gj0 || gj > gj1) continue;
        const int tile = ty * tiles_x + tx;
        const int it0 = part_off ? part_off[tile] : tile, it1 = part_off ? part_off[tile + 1] : tile + 1;
        // (a crowded tile of an adaptive plan has tens of work items: eight loads in flight per round trip, added in order)
        constexpr int kItemBatch = 8;
        for (int it = it0; it < it1; it += kItemBatch) {
          float t[kItemBatch];
#pragma unroll
          for (int j = 0; j < kItemBatch; ++j)
            t[j] = partials[(((int64_t)min(it + j, it1 - 1) * 2 + ch) * kGridCells + (gi - gi0)) * kGridCells + (gj - gj0)];
#pragma unroll
          for (int j = 0; j < kItemBatch; ++j)
            if (it + j < it1) acc += t[j];
        }
      }
    }
  }
  adam_apply(job, idx, st, acc, d_grid);
}

// pass 2: one wavefront per grid cell, lanes stride over the cell's column support
__global__ void __launch_bounds__(256)
upsample_bwd_cols_kernel(const float* __restrict__ S, Axis ay, Axis ax, int W, float* __restrict__ d_grid, AdamJob job) {
  const int lane = threadIdx.x & 63;
  const int gj = blockIdx.x * 4 + (threadIdx.x >> 6), gi = blockIdx.y, ch = blockIdx.z;
  if (job.theta != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 64) adam_record_loss(job, lane);
  if (gj >= ax.g) return;
  int c_lo, c_hi;
  support(ax, gj, W, &c_lo, &c_hi);
  const float* s = S + ((int64_t)ch * ay.g + gi) * W;
  float acc = 0.0f;
  for (int c = c_lo + lane; c < c_hi; c += 64) acc += weight_on(ax, c, gj) * s[c];
  acc = wave_sum(acc);
  if (lane == 0) {
    const int64_t i = ((int64_t)ch * ay.g + gi) * ax.g + gj;
    adam_apply(job, i, (int64_t)gi * ax.g + gj, acc, d_grid);
  }
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
  dim3 g((W + 1023) / 1024, (H + kUpRows - 1) / kUpRows, 2);
  if (W % 4 == 0) upsample_kernel<true><<<g, dim3(256), 0, as_stream(stream)>>>(grid, ay, ax, H, W, dense);
  else upsample_kernel<false><<<g, dim3(256), 0, as_stream(stream)>>>(grid, ay, ax, H, W, dense);
  EBOS_CHECK_LAUNCH("ebos_upsample_patch_flow");
  return EBOS_OK;
}

size_t ebos_upsample_bwd_scratch_bytes(int gh, int W) {
  return gh > 0 && W > 0 ? (size_t)2 * gh * W * sizeof(float) : 0;
}

static int upsample_bwd_impl(const float* d_dense, int gh, int gw, int patch_h, int patch_w, int slide_h, int slide_w, int H,
                            int W, float* scratch, float* d_grid, const ebos::AdamJob& job, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(d_dense && d_grid && scratch, "ebos_upsample_patch_flow_bwd: NULL d_dense/d_grid/scratch");
  EBOS_REQUIRE(gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0 && H > 0 && W > 0,
               "ebos_upsample_patch_flow_bwd: bad sizes");
  const Axis ay = make_axis(gh, patch_h, slide_h, H), ax = make_axis(gw, patch_w, slide_w, W);
  EBOS_REQUIRE(ay.off >= 0 && ax.off >= 0, "ebos_upsample_patch_flow_bwd: image larger than the resized grid");
  hipStream_t s = as_stream(stream);
  upsample_bwd_rows_kernel<<<dim3((W + 63) / 64, gh, 2), dim3(256), 0, s>>>(d_dense, ay, H, W, scratch);
  upsample_bwd_cols_kernel<<<dim3((gw + 3) / 4, gh, 2), dim3(256), 0, s>>>(scratch, ay, ax, W, d_grid, job);
  EBOS_CHECK_LAUNCH("ebos_upsample_patch_flow_bwd");
  return EBOS_OK;
}

int ebos_upsample_patch_flow_bwd_f32(const float* d_dense, int gh, int gw, int patch_h, int patch_w, int slide_h,
                                     int slide_w, int H, int W, float* scratch, float* d_grid, ebos_stream_t stream) {
  return upsample_bwd_impl(d_dense, gh, gw, patch_h, patch_w, slide_h, slide_w, H, W, scratch, d_grid, ebos::AdamJob{}, stream);
}

static int make_adam_job(ebos::AdamJob* out, float* theta, float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2,
                         double eps, int t, int* step, const float* contrast, float contrast_scale, const double* reg_partials,
                         int n_reg, float* losses, int losses_cap, const float* grad_mask, const char* who) {
  using namespace ebos;
  EBOS_REQUIRE(theta && exp_avg && exp_avg_sq && step, "%s: NULL optimiser buffer", who);
  EBOS_REQUIRE(t >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
               "%s: bad hyper-parameters (t = %d)", who, t);
  EBOS_REQUIRE(n_reg >= 0 && (n_reg == 0 || reg_partials) && losses_cap >= 0, "%s: bad loss bookkeeping", who);
  const AdamCoef coef = adam_coef(lr, beta1, beta2, t);  // (bias corrections of step t, in double, as torch computes them)
  AdamJob job{};
  job.theta = theta;
  job.exp_avg = exp_avg;
  job.exp_avg_sq = exp_avg_sq;
  job.step_size = coef.step_size;
  job.bc2_sqrt = coef.bc2_sqrt;
  job.beta2 = (float)beta2;
  job.w1 = (float)(1.0 - beta1);
  job.w2 = (float)(1.0 - beta2);
  job.eps = (float)eps;
  job.t = t;
  job.step = step;
  job.contrast = contrast;
  job.contrast_scale = contrast_scale;
  job.reg_partials = reg_partials;
  job.n_reg = n_reg;
  job.losses = losses;
  job.losses_cap = losses_cap;
  job.grad_mask = grad_mask;
  *out = job;
  return EBOS_OK;
}

int ebos_upsample_patch_flow_bwd_adam_f32(const float* d_dense, int gh, int gw, int patch_h, int patch_w, int slide_h,
                                          int slide_w, int H, int W, float* scratch, float* d_grid, float* theta, float* exp_avg,
                                          float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int t, int* step,
                                          const float* contrast, float contrast_scale, const double* reg_partials, int n_reg,
                                          float* losses, int losses_cap, const float* grad_mask, ebos_stream_t stream) {
  ebos::AdamJob job{};
  if (int rc = make_adam_job(&job, theta, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, t, step, contrast, contrast_scale, reg_partials,
                             n_reg, losses, losses_cap, grad_mask, "ebos_upsample_patch_flow_bwd_adam"))
    return rc;
  return upsample_bwd_impl(d_dense, gh, gw, patch_h, patch_w, slide_h, slide_w, H, W, scratch, d_grid, job, stream);
}

int ebos_patch_grad_combine_adam_f32(const float* grad_partials, const int32_t* part_table, int tile_h, int tile_w, int gh, int gw,
                                     int patch_h, int patch_w, int slide_h, int slide_w, int H, int W, float* d_grid, float* theta,
                                     float* exp_avg, float* exp_avg_sq, double lr, double beta1, double beta2, double eps, int t,
                                     int* step, const float* contrast, float contrast_scale, const double* reg_partials, int n_reg,
                                     float* losses, int losses_cap, const float* grad_mask, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(grad_partials && d_grid, "ebos_patch_grad_combine_adam: NULL grad_partials/d_grid");
  EBOS_REQUIRE(gh > 0 && gw > 0 && patch_h > 0 && patch_w > 0 && slide_h > 0 && slide_w > 0 && H > 0 && W > 0 && tile_h > 0 && tile_w > 0,
               "ebos_patch_grad_combine_adam: bad sizes");
  AdamJob job{};
  job.grad_mask = grad_mask;
  if (theta != nullptr) {  // theta == NULL: plain gradient, no optimiser step
    if (int rc = make_adam_job(&job, theta, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, t, step, contrast, contrast_scale, reg_partials,
                               n_reg, losses, losses_cap, grad_mask, "ebos_patch_grad_combine_adam"))
      return rc;
  }
  const Axis ay = make_axis(gh, patch_h, slide_h, H), ax = make_axis(gw, patch_w, slide_w, W);
  EBOS_REQUIRE(ay.off >= 0 && ax.off >= 0, "ebos_patch_grad_combine_adam: image larger than the resized grid");
  const int tiles_x = (W + tile_w - 1) / tile_w;
  patch_grad_combine_kernel<<<dim3((2 * gh * gw + 255) / 256 + 1), dim3(256), 0, as_stream(stream)>>>(grad_partials, part_table, tile_h, tile_w,
                                                                                              tiles_x, H, W, ay, ax, d_grid, job);
  EBOS_CHECK_LAUNCH("ebos_patch_grad_combine_adam");
  return EBOS_OK;
}

}  // extern "C"
