// c_abi_window.cpp -- the drop-in boundary without Python or torch: a host program that talks to libebos_hip.so through
// include/ebos_hip.h only.  It takes one window in the raw sensor-column form of the CCS recordings
// (src/data_loader/ccs.py:57-66 under the reference), builds the device plan, evaluates the contrast objective
// (variance of the image of warped events, src/warp.py:330-342 + src/event_image_converter.py:581-620) and its
// gradient for a dense flow, and checks two things a caller can check without a reference:
//   * mass:      sum(IWE) == number of events whose four taps stay inside the image (zero flow: all of them)
//   * gradient:  a finite difference of the variance along the direction "towards zero flow" matches <d_flow, direction>
// and then hands the whole Adam loop of a patch-flow solver to the library (ebos_cmax_patch_solve_f32: 4 kernel launches per
// iteration, the event kernels sample the patch grid themselves), checks that the loss falls, and runs the same loop once more as ONE
// resident launch (ebos_cmax_patch_solve_resident_f32: mailbox, status, fallback), whose losses must be the four launches' bit for bit.
//
//   hipcc --offload-arch=gfx950 -std=c++17 -Iinclude examples/c_abi_window.cpp \
//         -Levent_based_bos_amd/lib -lebos_hip -Wl,-rpath,$PWD/event_based_bos_amd/lib -o examples/c_abi_window
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ebos_hip.h"

#define HIP_OK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 2;                                                                     \
    }                                                                               \
  } while (0)
#define EBOS_OK_(x)                                                                 \
  do {                                                                              \
    if ((x) != 0) {                                                                 \
      std::fprintf(stderr, "%s: %s\n", #x, ebos_last_error());                      \
      return 3;                                                                     \
    }                                                                               \
  } while (0)

template <typename T>
static T* dev_alloc(size_t n) {
  void* p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) std::abort();
  (void)hipMemset(p, 0, (n ? n : 1) * sizeof(T));
  return static_cast<T*>(p);
}

int main(int argc, char** argv) {
  const int H = 260, W = 346, TH = 32, TW = 32, HALO = 32;  // a DAVIS-sized sensor; tile / halo: see ebos_slab_config
  const int64_t n = argc > 1 ? std::atoll(argv[1]) : 200000;
  std::printf("libebos_hip ABI %d (%s)\n", ebos_version(), ebos_build_info());

  // ---- a synthetic window in raw sensor columns (x = column, y = row, t in microseconds, polarity)
  std::vector<int16_t> col(n), row(n);
  std::vector<int32_t> t(n);
  std::vector<uint8_t> pol(n);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (int64_t i = 0; i < n; ++i) {
    col[i] = (int16_t)(8 + rnd() % (W - 16));  // away from the border: with |flow| <= 3 every tap stays inside
    row[i] = (int16_t)(8 + rnd() % (H - 16));
    t[i] = 10000000 + (int32_t)(i * 8300 / n);
    pol[i] = (uint8_t)(rnd() & 1);
  }
  auto *d_col = dev_alloc<int16_t>(n), *d_row = dev_alloc<int16_t>(n);
  auto* d_t = dev_alloc<int32_t>(n);
  auto* d_pol = dev_alloc<uint8_t>(n);
  HIP_OK(hipMemcpy(d_col, col.data(), n * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_row, row.data(), n * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_t, t.data(), n * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_pol, pol.data(), n, hipMemcpyHostToDevice));

  // ---- plan: raw columns -> SoA (fp64 time arithmetic) -> binned by source tile -> compact 6 B/event form
  const int64_t npad = (n + 3) / 4 * 4 + 4;
  auto *x = dev_alloc<float>(npad), *y = dev_alloc<float>(npad), *dt = dev_alloc<float>(npad), *p = dev_alloc<float>(npad);
  auto* ticks = dev_alloc<int64_t>(2);
  auto* tmm = dev_alloc<double>(2);
  EBOS_OK_(ebos_raw_time_range(d_t, 4, n, 1e6, ticks, tmm, nullptr));
  EBOS_OK_(ebos_raw_events_to_soa(d_col, d_row, d_t, 4, d_pol, 1e6, tmm, EBOS_REF_FIRST, 0.0, 1, n, x, y, dt, p, nullptr));
  const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW, n_tiles = tiles_y * tiles_x;
  const int64_t n_keys = (int64_t)n_tiles * TH * TW;
  auto *xs = dev_alloc<float>(npad), *ys = dev_alloc<float>(npad), *dts = dev_alloc<float>(npad), *ps = dev_alloc<float>(npad);
  auto* perm = dev_alloc<int32_t>(n);
  auto* key_offsets = dev_alloc<int32_t>(n_keys + 1);
  auto* counts = dev_alloc<int32_t>(2);
  const size_t scratch_bytes = ebos_bin_scratch_bytes(n_keys);
  auto* scratch = dev_alloc<char>(scratch_bytes);
  EBOS_OK_(ebos_bin_events_f32(x, y, dt, p, n, H, W, TH, TW, xs, ys, dts, ps, perm, key_offsets, counts, counts + 1, scratch,
                               scratch_bytes, nullptr));
  int32_t h_counts[2];
  HIP_OK(hipMemcpy(h_counts, counts, 8, hipMemcpyDeviceToHost));
  if (h_counts[0] != 0 || h_counts[1] != 0) {
    std::fprintf(stderr, "unexpected: %d events outside the image, %d fractional\n", h_counts[0], h_counts[1]);
    return 4;
  }
  const int64_t cap = n + 3 * n_tiles + 8;
  auto* grp = dev_alloc<int32_t>(n_tiles + 1);
  auto* cpix = dev_alloc<uint16_t>(cap);
  auto* cdt = dev_alloc<float>(cap);
  std::vector<float> nan_fill(cap, std::nanf(""));
  HIP_OK(hipMemcpy(cdt, nan_fill.data(), cap * 4, hipMemcpyHostToDevice));  // padding slots carry dt = NaN
  EBOS_OK_(ebos_plan_compact_f32(xs, ys, dts, key_offsets, n, H, W, TH, TW, grp, cpix, cdt, cap, nullptr));

  // ---- objective and gradient: one forward call (3 kernels), one backward call (1 kernel)
  const size_t ws_bytes = ebos_iwe_slab_workspace_bytes(H, W, TH, TW, HALO, 1, 0, 0);
  auto* ws = dev_alloc<char>(ws_bytes);  // zero-filled once
  auto *flow = dev_alloc<float>(2 * H * W), *d_flow = dev_alloc<float>(2 * H * W), *iwe = dev_alloc<float>(H * W);
  auto* var = dev_alloc<float>(1);
  auto* moments = dev_alloc<double>(2);
  auto* upstream = dev_alloc<float>(1);
  const float one = 1.0f;
  HIP_OK(hipMemcpy(upstream, &one, 4, hipMemcpyHostToDevice));
  std::vector<float> h_flow(2 * H * W), h_dir(2 * H * W);
  for (int r = 0; r < H; ++r)
    for (int c = 0; c < W; ++c) {
      h_flow[r * W + c] = 2.0f * std::sin(0.02f * r) + 0.6f;            // rows
      h_flow[H * W + r * W + c] = -1.5f * std::cos(0.015f * c) + 0.35f;  // columns
    }
  // direction: towards zero flow.  The events do not move in this synthetic window, so shrinking the flow sharpens the
  // image -- a direction along which the variance has a large, well-defined slope
  for (size_t i = 0; i < h_dir.size(); ++i) h_dir[i] = -0.25f * h_flow[i];
  auto objective = [&](const std::vector<float>& f, float* out_var, bool backward) -> int {
    if (hipMemcpy(flow, f.data(), f.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    if (ebos_iwe_dense_slab_f32(nullptr, nullptr, nullptr, nullptr, grp, cpix, cdt, key_offsets, n, flow, H, W, TH, TW, HALO, 1, 0,
                                0, ws, ws_bytes, iwe, 1, 0, var, moments, nullptr, nullptr) != 0)
      return 3;
    if (backward && ebos_iwe_dense_tiled_bwd_f32(nullptr, nullptr, nullptr, nullptr, grp, cpix, cdt, key_offsets, n, flow, H, W, TH,
                                                 TW, HALO, 0, 0, iwe, nullptr, 0, d_flow, nullptr, moments, upstream, nullptr,
                                                 nullptr, 0, nullptr, nullptr) != 0)
      return 3;
    return hipMemcpy(out_var, var, 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
  };

  float v0 = 0.f;
  std::vector<float> zero(2 * H * W, 0.0f);
  if (int rc = objective(zero, &v0, false)) { std::fprintf(stderr, "objective failed: %s\n", ebos_last_error()); return rc; }
  std::vector<float> h_iwe(H * W);
  HIP_OK(hipMemcpy(h_iwe.data(), iwe, h_iwe.size() * 4, hipMemcpyDeviceToHost));
  double mass = 0.0;
  for (float v : h_iwe) mass += v;
  std::printf("zero flow:   variance %.6f, sum(IWE) %.1f of %lld events\n", v0, mass, (long long)n);
  if (std::fabs(mass - (double)n) > 1e-6 * (double)n) { std::fprintf(stderr, "mass is not conserved\n"); return 5; }

  float v = 0.f, vp = 0.f, vm = 0.f;
  if (int rc = objective(h_flow, &v, true)) { std::fprintf(stderr, "objective failed: %s\n", ebos_last_error()); return rc; }
  std::vector<float> h_grad(2 * H * W);
  HIP_OK(hipMemcpy(h_grad.data(), d_flow, h_grad.size() * 4, hipMemcpyDeviceToHost));
  double directional = 0.0;
  for (size_t i = 0; i < h_grad.size(); ++i) directional += (double)h_grad[i] * (double)h_dir[i];
  const float eps = 0.02f;
  std::vector<float> fp(h_flow), fm(h_flow);
  for (size_t i = 0; i < fp.size(); ++i) { fp[i] += eps * h_dir[i]; fm[i] -= eps * h_dir[i]; }
  if (objective(fp, &vp, false) || objective(fm, &vm, false)) return 3;
  const double fd = ((double)vp - (double)vm) / (2.0 * eps);
  std::printf("smooth flow: variance %.6f, d/d(direction): analytic %.6f, finite difference %.6f\n", v, directional, fd);
  if (std::fabs(fd - directional) > 0.1 * std::fabs(directional) + 1e-4) {  // the objective has kinks: FD is approximate
 std::fprintf(stderr, "gradient check failed\n"); return 6; }
  // ---- the Adam loop of a patch-flow solver, enqueued by ONE call: patch grid [2, gh, gw] -> flow by the patch -> dense map of
  // src/solver/patch_eklt.py:173-204, evaluated per source tile inside the event kernels (no dense flow field), loss =
  // -variance + 0.001 flow_norm.  The events of this window do not move, so from a grid of 1.5 px the loop has to walk the
  // flow back towards zero: the loss must fall.
  const int PH = 20, PW = 26, gh = (H + PH - 1) / PH, gw = (W + PW - 1) / PW, n_iter = 40;
  if (!ebos_patch_fused_supported(TH, TW, HALO, PH, PW)) {
    std::printf("(tile %dx%d + %d leaves no LDS for grid sampling: solver section skipped)\nOK\n", TH, TW, HALO);
    return 0;
  }
  const size_t n_grid = (size_t)2 * gh * gw;
  auto *theta = dev_alloc<float>(n_grid), *d_theta = dev_alloc<float>(n_grid), *m1 = dev_alloc<float>(n_grid), *m2 = dev_alloc<float>(n_grid);
  std::vector<float> h_theta(n_grid, 1.5f);
  HIP_OK(hipMemcpy(theta, h_theta.data(), n_grid * 4, hipMemcpyHostToDevice));
  const size_t gp_bytes = ebos_patch_grad_partials_bytes(H, W, TH, TW, 0);
  const size_t n_items = gp_bytes / 2048;  // one regulariser value partial per work item of the backward kernel
  const size_t n_reg = n_items > (size_t)ebos_flow_regularisers_partials() ? n_items : (size_t)ebos_flow_regularisers_partials();
  const float minus_one = -1.0f;
  HIP_OK(hipMemcpy(upstream, &minus_one, 4, hipMemcpyHostToDevice));  // loss = -w_variance * variance
  ebos_cmax_patch_problem q{};
  q.grp_offsets = grp; q.cpix = cpix; q.cdt = cdt; q.key_offsets = key_offsets; q.n = n;
  q.H = H; q.W = W; q.tile_h = TH; q.tile_w = TW; q.halo = HALO; q.splits = 1;
  q.gh = gh; q.gw = gw; q.patch_h = PH; q.patch_w = PW; q.slide_h = PH; q.slide_w = PW;
  q.w_variance = 1.0f; q.w_flow_norm = 0.001f;
  q.lr = 0.05; q.beta1 = 0.9; q.beta2 = 0.999; q.eps = 1e-8;
  q.theta = theta; q.d_theta = d_theta; q.exp_avg = m1; q.exp_avg_sq = m2; q.step = dev_alloc<int>(1);
  q.iwe = iwe; q.variance = var; q.moments = moments; q.upstream = upstream;
  q.reg_partials = dev_alloc<double>(n_reg);  // zero-filled once
  q.workspace = ws; q.workspace_bytes = ws_bytes;
  q.losses = dev_alloc<float>(n_iter); q.losses_cap = n_iter;
  q.grad_partials = reinterpret_cast<float*>(dev_alloc<char>(gp_bytes)); q.grad_partials_bytes = gp_bytes;
  if (ebos_cmax_patch_solve_f32(&q, n_iter, nullptr) != 0) { std::fprintf(stderr, "solve failed: %s\n", ebos_last_error()); return 7; }
  std::vector<float> h_losses(n_iter);
  HIP_OK(hipMemcpy(h_losses.data(), q.losses, n_iter * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_theta.data(), theta, n_grid * 4, hipMemcpyDeviceToHost));
  double mean_abs = 0.0;
  for (float t : h_theta) mean_abs += std::fabs(t) / (double)n_grid;
  std::printf("solver:      %d Adam iterations on a %dx%d patch grid, loss %.6f -> %.6f, mean |flow| 1.5 -> %.3f px\n", n_iter, gh, gw,
              h_losses[0], h_losses[n_iter - 1], mean_abs);
  for (float l : h_losses)
    if (!std::isfinite(l)) { std::fprintf(stderr, "non-finite loss\n"); return 8; }
  if (!(h_losses[n_iter - 1] < h_losses[0]) || !(mean_abs < 1.5)) { std::fprintf(stderr, "the solver did not improve the objective\n"); return 8; }

  // ---- the same loop as ONE resident launch (ebos_cmax_patch_solve_resident_f32): one workgroup per source tile stays on its CU for all
  // iterations.  Protocol: ask _supported, give it a mailbox, read the status afterwards; a negative status says why the launch ended
  // early -- theta and the optimiser state are then unchanged (or, after -102, those of the completed iterations the mailbox reports)
  // and the four launches above take over.  With run-time windows (EBOS_HALO_AUTO: normalised time, |dt| <= 1) in BOTH forms the
  // two run the same arithmetic in the same order: the losses must be the same bits.
  auto reset = [&]() -> int {
    std::vector<float> t0(n_grid, 1.5f), z(n_grid, 0.0f);
    const int zero_step = 0;
    if (hipMemcpy(theta, t0.data(), n_grid * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(m1, z.data(), n_grid * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m2, z.data(), n_grid * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(q.step, &zero_step, 4, hipMemcpyHostToDevice) != hipSuccess)
      return 2;
    return 0;
  };
  q.halo = ebos_halo_auto(HALO, 1.0);
  q.steps_done = 0;
  if (reset() || ebos_cmax_patch_solve_f32(&q, n_iter, nullptr) != 0) { std::fprintf(stderr, "solve (run-time windows) failed: %s\n", ebos_last_error()); return 7; }
  std::vector<float> l_four(n_iter), l_res(n_iter);
  HIP_OK(hipMemcpy(l_four.data(), q.losses, n_iter * 4, hipMemcpyDeviceToHost));
  if (!ebos_cmax_resident_supported(&q)) {
    std::printf("(resident launch not available for this geometry: %s)\nOK\n", ebos_last_error());
    return 0;
  }
  const size_t mb_bytes = ebos_cmax_resident_mailbox_bytes(H, W, TH, TW);
  auto* mailbox = dev_alloc<char>(mb_bytes);
  if (reset()) return 2;
  if (ebos_cmax_patch_solve_resident_f32(&q, n_iter, mailbox, mb_bytes, 2.0 /* s: cap of every in-kernel wait */, nullptr) != 0) {
    std::fprintf(stderr, "resident solve failed: %s\n", ebos_last_error());
    return 9;
  }
  const int status = ebos_cmax_resident_status(mailbox, nullptr), done = ebos_cmax_resident_iterations(mailbox, nullptr);
  if (status != 0) {  // not an error of the library: this window is the launches' (what a caller does: continue with ebos_cmax_patch_solve_f32)
    std::printf("resident launch ended early (status %d after %d iterations): %s\nOK\n", status, done, ebos_last_error());
    return 0;
  }
  HIP_OK(hipMemcpy(l_res.data(), q.losses, n_iter * 4, hipMemcpyDeviceToHost));
  int same = 0;
  for (int i = 0; i < n_iter; ++i) same += l_res[i] == l_four[i];
  std::printf("resident:    %d iterations as ONE launch, loss %.6f -> %.6f, %d of %d losses bit-identical to the four-launch loop\n", done,
              l_res[0], l_res[n_iter - 1], same, n_iter);
  if (done != n_iter || same != n_iter) { std::fprintf(stderr, "the resident loop differs from the four-launch loop\n"); return 10; }
  std::printf("OK\n");
  return 0;
}
