// solver_kernels.hip -- the pieces of one contrast-maximisation iteration that are not the event kernels:
// the flow regularisers (value + gradient in one pass over the dense flow) and the Adam update of the patch grid.
//
// Why these exist: a solver iteration on a 2 M-event window is ~70 us of event-kernel time, but expressed through
// autograd it is ~35 kernel launches (a dozen for a capturable Adam, ten for `norm(flow, dim=0).mean()` and its
// backward, scalar glue) and every launch -- even as a HIP-graph node -- costs 3.5-7 us on this stack
// (profiles/r01g_solver_iteration.txt).  With these two kernels the iteration is 10 launches.
//
// reference semantics:
//   flow_norm       src/costs/flow_norm.py:45-56      mean over pixels of the per-pixel L2 norm of the flow
//   image_gradient  src/costs/image_gradient.py:60-75 mean(|d flow/d row| + |d flow/d col|), torch.gradient
//                                                      (central differences, one-sided at the borders), unit weights
//   Adam            torch.optim.Adam (defaults: amsgrad = False, weight_decay = 0), the optimiser of the loop in
//                   src/solver/generative_max_likelihood.py:306-341
#include "common.h"
#include "patch_grid.h"

namespace ebos {
namespace {

constexpr int kRegGrid = 1024;

// Optional side job of the regulariser pass: turn the (sum, sum of squares) partials that the slab combine pass left
// (ebos_iwe_dense_slab_f32 with want_variance = 2) into the variance and (mean, M) -- what moments_finalize_kernel does
// in a launch of its own (4.7 us at the launch floor, plus a gap).  Same summation order: same result.
struct MomentsJob {
  const double* partials;  // nullptr = no side job
  int64_t n_partials, n_pixels;
  float* out_var;
  double* moments;
};

struct RegGrad {
  float gu, gv;
  double val;
};

__device__ __forceinline__ RegGrad reg_at(const float* __restrict__ flow, int r, int c, int H, int W, int64_t hw, float u, float v,
                                          float s_norm, float s_tv) {
  RegGrad o{0.0f, 0.0f, 0.0};
  if (s_norm != 0.0f) {
    const float nrm = sqrtf(u * u + v * v);
    o.val += (double)(s_norm * nrm);
    if (nrm > 0.0f) {  // torch: the sub-gradient of the norm at 0 is 0
      const float inv = s_norm / nrm;
      o.gu += inv * u;
      o.gv += inv * v;
    }
  }
  if (s_tv != 0.0f) {
    const float* col_u = flow + c;               // the column through (r, c): stride W
    const float* col_v = flow + hw + c;
    const float* row_u = flow + (int64_t)r * W;  // the row through (r, c): stride 1
    const float* row_v = flow + hw + (int64_t)r * W;
    o.val += (double)(s_tv * (fabsf(central(col_u, r, H, W)) + fabsf(central(row_u, c, W, 1)) +
                              fabsf(central(col_v, r, H, W)) + fabsf(central(row_v, c, W, 1))));
    o.gu += s_tv * (tv_adjoint(col_u, r, H, W) + tv_adjoint(row_u, c, W, 1));
    o.gv += s_tv * (tv_adjoint(col_v, r, H, W) + tv_adjoint(row_v, c, W, 1));
  }
  return o;
}

// thread = 4 consecutive pixels of one row (16-byte loads / stores when W % 4 == 0); workgroups stride over the
// (row, column-block) work items; one f64 partial per workgroup.
template <bool VEC4>
__global__ void __launch_bounds__(256)
flow_regularisers_kernel(const float* __restrict__ flow, int H, int W, float w_norm, float w_tv, float* __restrict__ d_flow,
                         double* __restrict__ partials, MomentsJob mj) {
  const int64_t hw = (int64_t)H * W;
  const float s_norm = w_norm / (float)hw, s_tv = w_tv / (float)(2 * hw);
  const int col_blocks = (W + 1023) / 1024;
  double acc = 0.0;
  for (int item = blockIdx.x; item < H * col_blocks; item += gridDim.x) {
    const int r = item / col_blocks, c0 = ((item - r * col_blocks) * 256 + threadIdx.x) * 4;
    if (c0 >= W) continue;
    const int64_t o = (int64_t)r * W + c0;
    float u[4], v[4], gu[4], gv[4];
    if (VEC4) {
      const float4 u4 = *reinterpret_cast<const float4*>(flow + o), v4 = *reinterpret_cast<const float4*>(flow + hw + o);
      u[0] = u4.x, u[1] = u4.y, u[2] = u4.z, u[3] = u4.w;
      v[0] = v4.x, v[1] = v4.y, v[2] = v4.z, v[3] = v4.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u[k] = c0 + k < W ? flow[o + k] : 0.0f;
        v[k] = c0 + k < W ? flow[hw + o + k] : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gu[k] = gv[k] = 0.0f;
      if (c0 + k < W) {
        const RegGrad g = reg_at(flow, r, c0 + k, H, W, hw, u[k], v[k], s_norm, s_tv);
        gu[k] = g.gu, gv[k] = g.gv;
        acc += g.val;
      }
    }
    if (VEC4) {
      *reinterpret_cast<float4*>(d_flow + o) = make_float4(gu[0], gu[1], gu[2], gu[3]);
      *reinterpret_cast<float4*>(d_flow + hw + o) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + k < W) {
          d_flow[o + k] = gu[k];
          d_flow[hw + o + k] = gv[k];
        }
    }
  }
  __shared__ double red[256 / kWave];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
  if (mj.partials != nullptr && blockIdx.x == gridDim.x - 1) {  // side job of one workgroup: the variance of the IWE
    double s = 0.0, ss = 0.0;
    for (int64_t i = threadIdx.x; i < mj.n_partials; i += blockDim.x) {
      s += mj.partials[2 * i];
      ss += mj.partials[2 * i + 1];
    }
    __syncthreads();
    s = block_sum(s, red);
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) {
      const double mean = mj.n_pixels > 0 ? s / (double)mj.n_pixels : 0.0;
      if (mj.out_var) mj.out_var[0] = (float)((ss - s * mean) / (double)(mj.n_pixels - 1));
      if (mj.moments) {
        mj.moments[0] = mean;
        mj.moments[1] = (double)mj.n_pixels;
      }
    }
  }
}

// One workgroup.  loss[t] = contrast_scale * contrast + sum(reg_partials) is recorded for the parameters BEFORE the
// update (what the torch loop records), then Adam advances theta and the step counter.
__global__ void __launch_bounds__(1024)
cmax_adam_step_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                      int n, double lr, double beta1, double beta2, double eps, int* __restrict__ step,
                      const float* __restrict__ contrast, float contrast_scale, const double* __restrict__ reg_partials,
                      int n_reg, float* __restrict__ losses, int losses_cap) {
  const int t = step[0] + 1;
  const AdamCoef coef = adam_coef(lr, beta1, beta2, t);
  const float b2 = (float)beta2, w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), e = (float)eps;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float mi = m[i], vi = v[i], th = theta[i];
    adam_update(grad[i], mi, vi, th, coef.step_size, coef.bc2_sqrt, b2, w1, w2, e);  // (patch_grid.h: one arithmetic for every kernel)
    m[i] = mi;
    v[i] = vi;
    theta[i] = th;
  }
  double reg = 0.0;
  for (int i = threadIdx.x; i < n_reg; i += blockDim.x) reg += reg_partials[i];
  __shared__ double red[1024 / kWave];
  reg = block_sum(reg, red);
  __syncthreads();  // every thread has read step[0]
  if (threadIdx.x == 0) {
    if (losses != nullptr && t - 1 < losses_cap)
      losses[t - 1] = (float)((double)contrast_scale * (double)(contrast ? contrast[0] : 0.0f) + reg);
    step[0] = t;
  }
}

}  // namespace
}  // namespace ebos

extern "C" {

int ebos_flow_regularisers_partials(void) { return ebos::kRegGrid; }

int ebos_flow_regularisers_f32(const float* flow, int H, int W, float w_flow_norm, float w_image_gradient, float* d_flow,
                               double* partials, const double* var_partials, int64_t n_var_partials, int64_t n_var_pixels,
                               float* out_variance, double* moments, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(var_partials == nullptr || (n_var_partials >= 1 && n_var_pixels >= 2 && (out_variance || moments)),
               "ebos_flow_regularisers: bad variance side job");
  const MomentsJob mj{var_partials, n_var_partials, n_var_pixels, out_variance, moments};
  EBOS_REQUIRE(flow && d_flow && partials && flow != d_flow, "ebos_flow_regularisers: NULL or aliased buffers");
  EBOS_REQUIRE(H >= 1 && W >= 1, "ebos_flow_regularisers: bad sizes");
  EBOS_REQUIRE(w_image_gradient == 0.0f || (H >= 2 && W >= 2),
               "ebos_flow_regularisers: image_gradient needs at least 2 samples per axis (torch.gradient)");
  if (W % 4 == 0)
    flow_regularisers_kernel<true><<<dim3(kRegGrid), dim3(256), 0, as_stream(stream)>>>(flow, H, W, w_flow_norm, w_image_gradient,
                                                                                       d_flow, partials, mj);
  else
    flow_regularisers_kernel<false><<<dim3(kRegGrid), dim3(256), 0, as_stream(stream)>>>(flow, H, W, w_flow_norm,
                                                                                        w_image_gradient, d_flow, partials, mj);
  EBOS_CHECK_LAUNCH("ebos_flow_regularisers");
  return EBOS_OK;
}

int ebos_cmax_adam_step_f32(float* theta, const float* grad, float* exp_avg, float* exp_avg_sq, int n, double lr,
                            double beta1, double beta2, double eps, int* step, const float* contrast,
                            float contrast_scale, const double* reg_partials, int n_reg, float* losses, int losses_cap,
                            ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(theta && grad && exp_avg && exp_avg_sq && step, "ebos_cmax_adam_step: NULL buffer");
  EBOS_REQUIRE(n >= 1 && lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0,
               "ebos_cmax_adam_step: bad hyper-parameters");
  EBOS_REQUIRE(n_reg >= 0 && (n_reg == 0 || reg_partials) && losses_cap >= 0, "ebos_cmax_adam_step: bad loss bookkeeping");
  cmax_adam_step_kernel<<<dim3(1), dim3(1024), 0, as_stream(stream)>>>(theta, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                                                                      eps, step, contrast, contrast_scale, reg_partials,
                                                                      n_reg, losses, losses_cap);
  EBOS_CHECK_LAUNCH("ebos_cmax_adam_step");
  return EBOS_OK;
}

static int cmax_check_problem(const ebos_cmax_patch_problem* q) {
  using namespace ebos;
  EBOS_REQUIRE(q != nullptr && q->steps_done >= 0, "ebos_cmax_patch_solve: NULL problem or negative steps_done");
  EBOS_REQUIRE((q->cfx == nullptr) == (q->cfy == nullptr), "ebos_cmax_patch_solve: cfx / cfy come together");
  EBOS_REQUIRE(q->theta && q->d_theta && q->exp_avg && q->exp_avg_sq && q->step && q->iwe && q->variance && q->moments &&
                   q->upstream && q->workspace && q->reg_partials,
               "ebos_cmax_patch_solve: NULL buffer");
  const bool has_reg_ = q->w_flow_norm != 0.0f || q->w_image_gradient != 0.0f;
  if (q->grad_partials != nullptr) {  // the event kernels sample the patch grid: dense only feeds the regulariser pass
    EBOS_REQUIRE(q->grp_offsets && q->cpix && q->cdt, "ebos_cmax_patch_solve: grad_partials (grid-sampling kernels) needs the compact plan");
    EBOS_REQUIRE(ebos_patch_fused_supported(q->tile_h, q->tile_w, q->halo, q->slide_h, q->slide_w),
                 "ebos_cmax_patch_solve: grad_partials given but tile %dx%d halo %d / sliding window %dx%d is outside "
                 "ebos_patch_fused_supported", q->tile_h, q->tile_w, q->halo, q->slide_h, q->slide_w);
  } else {
    EBOS_REQUIRE(q->dense && q->d_dense && q->upsample_scratch, "ebos_cmax_patch_solve: NULL dense / d_dense / upsample_scratch");
    EBOS_REQUIRE(q->cfx == nullptr || (q->xs && q->ys && q->dts),
                 "ebos_cmax_patch_solve: a window of fractional source coordinates on the dense route needs xs / ys / dts");
  }
  // (grid sampling: the backward kernel evaluates the regularisers from the tile's flow, no d_reg image)
  EBOS_REQUIRE(!has_reg_ || q->d_reg || q->grad_partials != nullptr,
               "ebos_cmax_patch_solve: regulariser weights given but d_reg is NULL");
  EBOS_REQUIRE((q->w_variance != 0.0f) != (q->w_gradient_magnitude != 0.0f),
               "ebos_cmax_patch_solve: exactly one of w_variance / w_gradient_magnitude must be non-zero");
  EBOS_REQUIRE(q->w_gradient_magnitude == 0.0f || (q->d_iwe && q->cost_scratch),
               "ebos_cmax_patch_solve: the gradient-magnitude contrast needs d_iwe and cost_scratch");
  if (q->blur_k0 != 0.0f) {  // the variance of the 3-tap blurred image (iwe.blur_sigma > 0)
    EBOS_REQUIRE(q->blur_k0 > 0.0f && q->blur_k1 > 0.0f && q->blur_image && q->cost_scratch,
                 "ebos_cmax_patch_solve: the blurred contrast needs positive taps, blur_image and cost_scratch");
    if (q->w_gradient_magnitude != 0.0f) {
      set_error("ebos_cmax_patch_solve: the blurred image goes with the variance contrast only");
      return EBOS_ERR_UNSUPPORTED;
    }
    const size_t need = (size_t)16 * (size_t)ebos_blur3_variance_partials(q->H + 2 * q->pad_h, q->W + 2 * q->pad_w);
    if (q->cost_scratch_bytes < need) {
      set_error("ebos_cmax_patch_solve: cost_scratch too small for the blur's partials (%zu < %zu)", q->cost_scratch_bytes, need);
      return EBOS_ERR_SCRATCH;
    }
  }
  return EBOS_OK;
}

static int cmax_enqueue_iteration(const ebos_cmax_patch_problem* q, int t, ebos_stream_t stream) {
  using namespace ebos;
  const bool grid = q->grad_partials != nullptr;  // the event kernels evaluate the grid -> dense map per tile themselves
  // with grid sampling the backward kernel evaluates the flow regularisers from the tile's own flow (2 px apron in LDS)
  const bool fuse_norm = grid && (q->w_image_gradient != 0.0f || q->w_flow_norm != 0.0f);
  const bool has_reg = (q->w_flow_norm != 0.0f || q->w_image_gradient != 0.0f) && !fuse_norm;  // regulariser LAUNCH needed
  int rc = EBOS_OK;
  if (!grid || has_reg) {  // (the regulariser pass reads the dense field)
    rc = ebos_upsample_patch_flow_f32(q->theta, q->gh, q->gw, q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->dense,
                                      stream);
    if (rc) return rc;
  }
  const bool use_gm = q->w_gradient_magnitude != 0.0f;
  const bool blur = q->blur_k0 != 0.0f;  // (variance contrast: cmax_check_problem)
  // dense route: compact arrays that carry fractions (cfx / cfy) are the grid-sampling and the resident kernels' -- the dense-flow
  // kernels read such a window from xs / ys / dts
  const bool frac = q->cfx != nullptr;
  const int32_t* d_grp = frac ? nullptr : q->grp_offsets;
  const uint16_t* d_cpix = frac ? nullptr : q->cpix;
  const float* d_cdt = frac ? nullptr : q->cdt;
  const int h = q->H + 2 * q->pad_h, w = q->W + 2 * q->pad_w;
  const float contrast_weight = use_gm ? q->w_gradient_magnitude : q->w_variance;
  if (grid)  // (cfx / cfy: the compact slots carry the fractions of undistorted events; NULL: integer source pixels)
    rc = ebos_iwe_patch_slab_frac_f32(q->grp_offsets, q->cpix, q->cdt, q->cfx, q->cfy, q->key_offsets, q->n, q->theta, q->gh, q->gw, q->patch_h,
                                      q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->tile_h, q->tile_w, q->halo, q->splits, q->pad_h,
                                      q->pad_w, q->workspace, q->workspace_bytes, q->iwe, (use_gm || blur) ? 0 : 2, q->omit_boundary,
                                      q->variance, q->moments, q->part_table, stream);  // (variance: partials only; the regulariser or backward kernel reduces them)
  else
    rc = ebos_iwe_dense_slab_f32(q->xs, q->ys, q->dts, nullptr, d_grp, d_cpix, d_cdt, q->key_offsets, q->n, q->dense,
                                 q->H, q->W, q->tile_h, q->tile_w, q->halo, q->splits, q->pad_h, q->pad_w, q->workspace,
                                 q->workspace_bytes, q->iwe, (use_gm || blur) ? 0 : (has_reg ? 2 : 1), q->omit_boundary, q->variance,
                                 q->moments, q->part_table, stream);
  if (rc) return rc;
  if (use_gm) {  // contrast = mean squared Sobel gradient of the IWE; its gradient image feeds the backward event kernel
    // (one Sobel pass: value partials + gradient image; cost_scratch holds the partials)
    rc = ebos_gradient_magnitude_fused_f32(q->iwe, h, w, q->omit_boundary, q->upstream, q->variance, q->d_iwe,
                                           reinterpret_cast<double*>(q->cost_scratch), (int64_t)(q->cost_scratch_bytes / sizeof(double)), stream);
    if (rc) return rc;
  }
  size_t off = 0;
  int64_t n_parts = 0, n_px = 0;
  if (has_reg || (grid && !use_gm)) {
    rc = ebos_iwe_slab_partials(q->H, q->W, q->tile_h, q->tile_w, q->halo, q->splits, q->pad_h, q->pad_w, q->omit_boundary, &off,
                                &n_parts, &n_px);
    if (rc) return rc;
  }
  const double* var_partials = reinterpret_cast<const double*>(static_cast<const char*>(q->workspace) + off);
  if (has_reg) {  // the regulariser pass also reduces the variance moments the combine pass left (no finalize launch)
    rc = ebos_flow_regularisers_f32(q->dense, q->H, q->W, q->w_flow_norm, q->w_image_gradient, q->d_reg, q->reg_partials,
                                    (use_gm || blur) ? nullptr : var_partials, n_parts, n_px, q->variance, q->moments, stream);
    if (rc) return rc;
  }
  if (grid && blur) {
    // the blurred contrast: one pass over the image -> (sum, sum of squares) partials of the blurred pixels + z = B^T (m . B x);
    // the backward kernel reduces the partials and forms its upstream a z + c wgt itself (blur3.h)
    const int64_t n_blur = ebos_blur3_variance_partials(h, w);
    const int lo = q->omit_boundary ? 1 : 0;
    const int64_t n_valid = (int64_t)(h - 2 * lo > 0 ? h - 2 * lo : 0) * (w - 2 * lo > 0 ? w - 2 * lo : 0);
    rc = ebos_blur3_variance_adjoint_f32(q->iwe, h, w, q->omit_boundary, q->blur_k0, q->blur_k1, q->blur_image,
                                         reinterpret_cast<double*>(q->cost_scratch), n_blur, stream);
    if (rc) return rc;
    rc = ebos_iwe_patch_tiled_bwd_frac_f32(q->grp_offsets, q->cpix, q->cdt, q->cfx, q->cfy, q->key_offsets, q->n, q->theta, q->gh, q->gw,
                                           q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->tile_h, q->tile_w, q->halo,
                                           q->pad_h, q->pad_w, q->blur_image, lo, nullptr, q->upstream, nullptr, q->grad_partials,
                                           q->grad_partials_bytes, q->splits == 0 ? q->part_table : nullptr,
                                           fuse_norm ? q->w_flow_norm : 0.0f, fuse_norm ? q->w_image_gradient : 0.0f, q->reg_partials,
                                           reinterpret_cast<const double*>(q->cost_scratch), n_blur, n_valid, q->variance, q->moments,
                                           q->blur_k0, q->blur_k1, stream);
    if (rc) return rc;
    const int n_items = (int)(ebos_patch_grad_partials_bytes(q->H, q->W, q->tile_h, q->tile_w, q->splits == 0) / 2048);
    return ebos_patch_grad_combine_adam_f32(q->grad_partials, q->splits == 0 ? q->part_table : nullptr, q->tile_h, q->tile_w, q->gh,
                                            q->gw, q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->d_theta, q->theta,
                                            q->exp_avg, q->exp_avg_sq, q->lr, q->beta1, q->beta2, q->eps, t, q->step, q->variance,
                                            -contrast_weight, q->reg_partials, fuse_norm ? n_items : 0, q->losses, q->losses_cap,
                                            q->theta_mask, stream);
  }
  if (grid) {
    rc = ebos_iwe_patch_tiled_bwd_frac_f32(q->grp_offsets, q->cpix, q->cdt, q->cfx, q->cfy, q->key_offsets, q->n, q->theta, q->gh, q->gw,
                                           q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->tile_h, q->tile_w, q->halo,
                                           q->pad_h, q->pad_w, use_gm ? q->d_iwe : q->iwe, use_gm ? 0 : (q->omit_boundary ? 1 : 0),
                                           (use_gm || !has_reg) ? nullptr : q->moments, use_gm ? nullptr : q->upstream,
                                           has_reg ? q->d_reg : nullptr, q->grad_partials, q->grad_partials_bytes,
                                           q->splits == 0 ? q->part_table : nullptr, fuse_norm ? q->w_flow_norm : 0.0f,
                                           fuse_norm ? q->w_image_gradient : 0.0f, q->reg_partials,
                                           (use_gm || has_reg) ? nullptr : var_partials, n_parts, n_px, q->variance, q->moments, 0.0f, 0.0f,
                                           stream);
    if (rc) return rc;
    const int n_items = (int)(ebos_patch_grad_partials_bytes(q->H, q->W, q->tile_h, q->tile_w, q->splits == 0) / 2048);
    // partial cell gradients -> d_theta, the Adam step of every grid element and the loss of the iteration
    return ebos_patch_grad_combine_adam_f32(q->grad_partials, q->splits == 0 ? q->part_table : nullptr, q->tile_h, q->tile_w, q->gh,
                                            q->gw, q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W, q->d_theta, q->theta,
                                            q->exp_avg, q->exp_avg_sq, q->lr, q->beta1, q->beta2, q->eps, t, q->step, q->variance,
                                            -contrast_weight, q->reg_partials, fuse_norm ? n_items : (has_reg ? ebos::kRegGrid : 0),
                                            q->losses, q->losses_cap, q->theta_mask, stream);
  }
  if (blur) {  // dense route: the same image pass; the backward kernel reduces its partials, reports the variance and folds a z + c wgt
    const int64_t n_blur = ebos_blur3_variance_partials(h, w);
    const int lo = q->omit_boundary ? 1 : 0;
    const int64_t n_valid = (int64_t)(h - 2 * lo > 0 ? h - 2 * lo : 0) * (w - 2 * lo > 0 ? w - 2 * lo : 0);
    rc = ebos_blur3_variance_adjoint_f32(q->iwe, h, w, q->omit_boundary, q->blur_k0, q->blur_k1, q->blur_image,
                                         reinterpret_cast<double*>(q->cost_scratch), n_blur, stream);
    if (rc) return rc;
    rc = ebos_iwe_dense_tiled_bwd_blur_f32(q->xs, q->ys, q->dts, d_grp, d_cpix, d_cdt, q->key_offsets, q->n, q->dense, q->H, q->W,
                                           q->tile_h, q->tile_w, q->halo, q->pad_h, q->pad_w, q->blur_image, lo, q->upstream,
                                           has_reg ? q->d_reg : nullptr, q->d_dense, q->workspace, q->workspace_bytes,
                                           q->splits == 0 ? q->part_table : nullptr, reinterpret_cast<const double*>(q->cost_scratch), n_blur,
                                           n_valid, q->variance, q->moments, q->blur_k0, q->blur_k1, stream);
  } else
  rc = ebos_iwe_dense_tiled_bwd_f32(q->xs, q->ys, q->dts, nullptr, d_grp, d_cpix, d_cdt, q->key_offsets, q->n,
                                    q->dense, q->H, q->W, q->tile_h, q->tile_w, q->halo, q->pad_h, q->pad_w,
                                    use_gm ? q->d_iwe : q->iwe, nullptr, use_gm ? 0 : (q->omit_boundary ? 1 : 0), q->d_dense, nullptr,
                                    use_gm ? nullptr : q->moments, use_gm ? nullptr : q->upstream,
                                    has_reg ? q->d_reg : nullptr, q->workspace, q->workspace_bytes,
                                    q->splits == 0 ? q->part_table : nullptr, stream);
  if (rc) return rc;
  // adjoint of the upsample + the Adam step of every grid element where its gradient appears + the loss of the iteration
  return ebos_upsample_patch_flow_bwd_adam_f32(q->d_dense, q->gh, q->gw, q->patch_h, q->patch_w, q->slide_h, q->slide_w, q->H, q->W,
                                               q->upsample_scratch, q->d_theta, q->theta, q->exp_avg, q->exp_avg_sq, q->lr, q->beta1,
                                               q->beta2, q->eps, t, q->step, q->variance, -contrast_weight, q->reg_partials,
                                               has_reg ? ebos::kRegGrid : 0, q->losses, q->losses_cap, q->theta_mask, stream);
}

int ebos_cmax_patch_solve_f32(const ebos_cmax_patch_problem* q, int n_iter, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(n_iter >= 0, "ebos_cmax_patch_solve: negative n_iter");
  if (int rc = cmax_check_problem(q)) return rc;
  for (int it = 0; it < n_iter; ++it)
    if (int rc = cmax_enqueue_iteration(q, q->steps_done + it + 1, stream)) return rc;
  return EBOS_OK;
}

int ebos_cmax_patch_solve_many_f32(const ebos_cmax_patch_problem* problems, const ebos_stream_t* streams, int n_problems,
                                   int n_iter) {
  using namespace ebos;
  EBOS_REQUIRE(problems && streams && n_problems >= 1 && n_iter >= 0, "ebos_cmax_patch_solve_many: bad arguments");
  for (int w = 0; w < n_problems; ++w)
    if (int rc = cmax_check_problem(problems + w)) return rc;
  // iteration-major: the windows' kernels alternate in the launch order, so that the small kernels of one window
  // (combine, finalize, regularisers, upsample, Adam) find free wave slots next to the one-workgroup-per-CU event
  // kernels of another
  for (int it = 0; it < n_iter; ++it)
    for (int w = 0; w < n_problems; ++w)
      if (int rc = cmax_enqueue_iteration(problems + w, problems[w].steps_done + it + 1, streams[w])) return rc;
  return EBOS_OK;
}

}  // extern "C"
