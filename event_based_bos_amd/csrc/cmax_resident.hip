// cmax_resident.hip -- the C entry points of the resident solver launches (kernels: cmax_resident_core.h, one translation unit per
// tile shape) and the per-device order of resident launches.
#include "cmax_resident_core.h"

#include <atomic>

namespace ebos {

// Resident launches of different streams may only run side by side while ALL their workgroups fit the device at once (two
// half-resident grids would wait for each other until their caps): per device, the launches in flight and their grid sizes are kept;
// a new launch first waits for the oldest ones until it fits beside the rest.  (Small sensors -- 99 tiles at 346 x 260 -- thus run
// two windows at a time; 256 tiles at 1280 x 720 one after the other.)
struct ResidentInFlight {
  hipEvent_t done;
  int workgroups;
};
int order_resident_launches(hipStream_t s, int workgroups, int n_cu, bool after_launch) {
  constexpr int kMaxDevices = 64, kMaxInFlight = 8;
  static ResidentInFlight fl[kMaxDevices][kMaxInFlight] = {};
  static int n_fl[kMaxDevices] = {};
  static std::atomic_flag lock = ATOMIC_FLAG_INIT;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return EBOS_ERR_LAUNCH;
  while (lock.test_and_set(std::memory_order_acquire)) {}
  int rc = EBOS_OK;
  ResidentInFlight* f = fl[dev];
  int& n = n_fl[dev];
  auto drop_oldest = [&]() {
    const hipEvent_t ev = f[0].done;   // (the event object is reused by the entry that takes the freed slot)
    for (int k = 1; k < n; ++k) f[k - 1] = f[k];
    --n;
    f[n].done = ev;
  };
  if (!after_launch) {
    while (n > 0 && hipEventQuery(f[0].done) == hipSuccess) drop_oldest();   // finished launches, oldest first
    (void)hipGetLastError();                                                  // (hipErrorNotReady is not an error here)
    int busy = 0;
    for (int k = 0; k < n; ++k) busy += f[k].workgroups;
    while (n > 0 && (busy + workgroups > n_cu || n == kMaxInFlight)) {
      if (hipStreamWaitEvent(s, f[0].done, 0) != hipSuccess) rc = EBOS_ERR_LAUNCH;
      busy -= f[0].workgroups;
      drop_oldest();
    }
  } else {
    if (n == kMaxInFlight) drop_oldest();  // (another thread's launches filled the table meanwhile)
    if (f[n].done == nullptr && hipEventCreateWithFlags(&f[n].done, hipEventDisableTiming) != hipSuccess) f[n].done = nullptr;
    if (f[n].done == nullptr || hipEventRecord(f[n].done, s) != hipSuccess) rc = EBOS_ERR_LAUNCH;
    else f[n].workgroups = workgroups, ++n;
  }
  lock.clear(std::memory_order_release);
  return rc;
}

namespace {

// a resident workgroup owns its CU (LDS): the grid must fit the device at once -- said by the `supported` queries already, so that a
// caller's fallback is taken before a launch is refused (720 x 640 on 32 x 32 tiles: 460 workgroups)
bool tiles_fit_device(int n_tiles, const char* what) {
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    set_error("%s: cannot query the device", what);
    return false;
  }
  if (n_tiles > n_cu) {
    set_error("%s: %d workgroups cannot be co-resident on %d CUs", what, n_tiles, n_cu);
    return false;
  }
  return true;
}

// outer_padding (src/event_image_converter.py:29-34) inside the resident kernels: every tile's window reaches at least the padding ring
// (so that the border tile that owns a ring pixel gathers everything deposited there), which has to stay a window the kernel's
// exchange of halos between direct neighbours covers: less than half a tile, at most the built halo
bool resident_padding_ok(int pad_h, int pad_w, int tile_h, int tile_w, int halo, const char* what) {
  const int pw4 = (pad_w + 3) & ~3;   // (windows are whole quads of columns)
  if (pad_h < 0 || pad_w < 0 || pad_h > halo || pw4 > halo || 2 * pad_h >= tile_h || 2 * pw4 >= tile_w) {
    set_error("%s: image padding %dx%d does not fit the windows of tile %dx%d halo %d (less than half a tile, at most the halo)", what,
              pad_h, pad_w, tile_h, tile_w, halo);
    return false;
  }
  return true;
}

// the geometry / objective a resident launch takes; reason in ebos_last_error otherwise
bool resident_problem_ok(const ebos_cmax_patch_problem* q) {
  const int halo = decode_halo(q->halo).halo;
  // (integer source pixels: the loop object of the grid-sampling route, whose four launches can take over; fractional ones: the
  // compact arrays with the fractions per slot -- that window's four-launch loop runs on the dense route)
  if (!q->grp_offsets || !q->cpix || !q->cdt || (q->grad_partials == nullptr && q->cfx == nullptr) || (q->cfx == nullptr) != (q->cfy == nullptr)) {
    set_error("resident solve: needs a compact plan -- of the grid-sampling route (grad_partials), or with the fractions of undistorted events (cfx / cfy)");
    return false;
  }
  // splits: 1, or 0 = "adaptive work items" (a table the four-launch pipeline splits crowded tiles by: the resident kernel always
  // runs one workgroup per tile and does not read it -- same objective, slab sums in another order where a tile was split)
  if (q->splits > 1 || q->splits < 0) {
    set_error("resident solve: one work item per tile (splits = %d)", q->splits);
    return false;
  }
  if (!resident_padding_ok(q->pad_h, q->pad_w, q->tile_h, q->tile_w, halo, "resident solve")) return false;
  if ((q->w_gradient_magnitude != 0.0f) == (q->w_variance != 0.0f)) {
    set_error("resident solve: exactly one of w_variance / w_gradient_magnitude must be non-zero");
    return false;
  }
  if (q->w_gradient_magnitude != 0.0f && q->blur_k0 != 0.0f) {
    set_error("resident solve: the blurred image goes with the variance contrast only");
    return false;
  }
  if (!ebos_patch_fused_supported(q->tile_h, q->tile_w, q->halo, q->slide_h, q->slide_w)) {
    set_error("resident solve: tile %dx%d halo %d / sliding window %dx%d is outside ebos_patch_fused_supported", q->tile_h, q->tile_w,
              halo, q->slide_h, q->slide_w);
    return false;
  }
  const bool built = halo == 32 && ((q->tile_h == 45 && q->tile_w == 80) || (q->tile_h == 32 && q->tile_w == 32) ||
                                    (q->tile_h == 32 && q->tile_w == 64));
  if (!built) {
    set_error("resident solve: no resident kernel built for tile %dx%d halo %d", q->tile_h, q->tile_w, halo);
    return false;
  }
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w;
  if (tiles_y * tiles_x > kBlock) {
    set_error("resident solve: %d tiles (one record per thread: <= %d)", tiles_y * tiles_x, kBlock);
    return false;
  }
  if (!tiles_fit_device(tiles_y * tiles_x, "resident solve")) return false;
  if (((q->tile_h + 2 * kBwdApron) / q->slide_h + 3) * ((q->tile_w + 2 * kBwdApron) / q->slide_w + 3) * 2 > kResElems) {
    set_error("resident solve: sliding window %dx%d: a tile's block of grid cells has more than %d elements", q->slide_h, q->slide_w, kResElems);
    return false;
  }
  // every cell sums at most kSpan x kSpan tiles, every tile's block waits for at most 64 tiles
  const Axis ay = make_axis(q->gh, q->patch_h, q->slide_h, q->H), ax = make_axis(q->gw, q->patch_w, q->slide_w, q->W);
  if (ay.off < 0 || ax.off < 0) {
    set_error("resident solve: image larger than the resized grid");
    return false;
  }
  int span_y = 0, span_x = 0;
  for (int gi = 0; gi < q->gh; ++gi) {
    int lo, hi;
    support(ay, gi, q->H, &lo, &hi);
    if (lo < hi) span_y = std::max(span_y, (hi - 1) / q->tile_h - lo / q->tile_h + 1);
  }
  for (int gj = 0; gj < q->gw; ++gj) {
    int lo, hi;
    support(ax, gj, q->W, &lo, &hi);
    if (lo < hi) span_x = std::max(span_x, (hi - 1) / q->tile_w - lo / q->tile_w + 1);
  }
  // (cells that span more than kSpan tiles on an axis -- the edge cells of a coarse scale -- are summed tile by tile in S3)
  if (span_y > 16 || span_x > 16) {
    set_error("resident solve: a grid cell's support spans %dx%d tiles (<= 16 per axis)", span_y, span_x);
    return false;
  }
  // ... and the cells of a tile's block (tile + apron) sum at most 64 tiles in all (one lane of the waiting wave each)
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx) {
      const int r0 = std::max(ty * q->tile_h - kBwdApron, 0), r1 = std::min(ty * q->tile_h + q->tile_h + kBwdApron, q->H) - 1;
      const int c0 = std::max(tx * q->tile_w - kBwdApron, 0), c1 = std::min(tx * q->tile_w + q->tile_w + kBwdApron, q->W) - 1;
      int lo, hi, d;
      support(ay, lerp_at(ay, r0).i0, q->H, &lo, &d);
      support(ay, lerp_at(ay, std::max(r1, r0)).i1, q->H, &d, &hi);
      const int ny = lo < hi ? std::min((hi - 1) / q->tile_h, tiles_y - 1) - lo / q->tile_h + 1 : 0;
      support(ax, lerp_at(ax, c0).i0, q->W, &lo, &d);
      support(ax, lerp_at(ax, std::max(c1, c0)).i1, q->W, &d, &hi);
      const int nx = lo < hi ? std::min((hi - 1) / q->tile_w, tiles_x - 1) - lo / q->tile_w + 1 : 0;
      if (ny * nx > kWave) {
        set_error("resident solve: the cell block of tile (%d, %d) sums %dx%d tiles (<= %d in all)", ty, tx, ny, nx, kWave);
        return false;
      }
    }
  return true;
}


// ... and the 2-DoF problem a resident launch takes
bool resident_2dof_ok(const ebos_cmax_2dof_problem* q) {
  const int halo = decode_halo(q->halo).halo;
  if (!q->grp_offsets || !q->cpix || !q->cdt) {
    set_error("resident 2-DoF solve: needs the compact plan (integer source pixels)");
    return false;
  }
  if ((q->cfx == nullptr) != (q->cfy == nullptr)) {  // (one without the other would launch the fractional kernel on a NULL array)
    set_error("resident 2-DoF solve: cfx and cfy come together (fractional source coordinates) or not at all");
    return false;
  }
  if (q->splits > 1 || q->splits < 0) {
    set_error("resident 2-DoF solve: one work item per tile (splits = %d)", q->splits);
    return false;
  }
  if (!resident_padding_ok(q->pad_h, q->pad_w, q->tile_h, q->tile_w, halo, "resident 2-DoF solve")) return false;
  if (q->w_variance == 0.0f) {
    set_error("resident 2-DoF solve: w_variance is 0");
    return false;
  }
  const bool built = halo == 32 && ((q->tile_h == 45 && q->tile_w == 80) || (q->tile_h == 32 && q->tile_w == 32) ||
                                    (q->tile_h == 32 && q->tile_w == 64));
  if (!built) {
    set_error("resident 2-DoF solve: no resident kernel built for tile %dx%d halo %d", q->tile_h, q->tile_w, halo);
    return false;
  }
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w;
  if (tiles_y * tiles_x > kBlock) {
    set_error("resident 2-DoF solve: %d tiles (one record per thread: <= %d)", tiles_y * tiles_x, kBlock);
    return false;
  }
  if (!tiles_fit_device(tiles_y * tiles_x, "resident 2-DoF solve")) return false;
  if (q->blur_k0 != 0.0f && (q->blur_k0 < 0.0f || q->blur_k1 <= 0.0f || q->H < 2 || q->W < 2)) {
    set_error("resident 2-DoF solve: bad blur taps / image smaller than 2 x 2");
    return false;
  }
  return true;
}

}  // namespace
}  // namespace ebos

extern "C" {


size_t ebos_cmax_resident_mailbox_bytes(int H, int W, int tile_h, int tile_w) {
  if (H <= 0 || W <= 0 || tile_h <= 0 || tile_w <= 0) return 0;
  return ebos::mailbox_layout(((H + tile_h - 1) / tile_h) * ((W + tile_w - 1) / tile_w)).total;
}

int ebos_cmax_resident_supported(const ebos_cmax_patch_problem* q) {
  using namespace ebos;
  if (q == nullptr) return 0;
  return resident_problem_ok(q) ? 1 : 0;
}

int ebos_cmax_patch_solve_resident_f32(const ebos_cmax_patch_problem* q, int n_iter, void* mailbox, size_t mailbox_bytes,
                                       double spin_timeout_s, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(q != nullptr && n_iter >= 0 && q->steps_done >= 0, "ebos_cmax_patch_solve_resident: NULL problem or negative counts");
  EBOS_REQUIRE(q->theta && q->d_theta && q->exp_avg && q->exp_avg_sq && q->step && q->iwe && q->variance && q->moments && q->workspace &&
                   q->key_offsets,
               "ebos_cmax_patch_solve_resident: NULL buffer");
  EBOS_REQUIRE(mailbox != nullptr && spin_timeout_s > 0.0, "ebos_cmax_patch_solve_resident: NULL mailbox or no spin cap");
  if (!resident_problem_ok(q)) return EBOS_ERR_UNSUPPORTED;
  const HaloArg ha = decode_halo(q->halo);
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w, n_tiles = tiles_y * tiles_x;
  const MailboxLayout m = mailbox_layout(n_tiles);
  if (mailbox_bytes < m.total) {
    set_error("ebos_cmax_patch_solve_resident: mailbox too small (%zu < %zu)", mailbox_bytes, m.total);
    return EBOS_ERR_SCRATCH;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(q->H, q->W, q->tile_h, q->tile_w, ha.halo, 1, 0, 0);
  if (q->workspace_bytes < need) {
    set_error("ebos_cmax_patch_solve_resident: workspace too small (%zu < %zu)", q->workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  if (q->grad_partials != nullptr && q->grad_partials_bytes < ebos_patch_grad_partials_bytes(q->H, q->W, q->tile_h, q->tile_w, 0)) {
    set_error("ebos_cmax_patch_solve_resident: grad_partials too small");
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  int rc = EBOS_ERR_UNSUPPORTED;
  if (q->tile_h == 45 && q->tile_w == 80 && ha.halo == 32) rc = resident_launch_45x80(q, n_iter, mailbox, spin_timeout_s, s);
  else if (q->tile_h == 32 && q->tile_w == 32 && ha.halo == 32) rc = resident_launch_32x32(q, n_iter, mailbox, spin_timeout_s, s);
  else if (q->tile_h == 32 && q->tile_w == 64 && ha.halo == 32) rc = resident_launch_32x64(q, n_iter, mailbox, spin_timeout_s, s);  // (720 x 640: hot_plate1's ROI)
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_cmax_patch_solve_resident");
  return EBOS_OK;
}

int ebos_cmax_2dof_resident_supported(const ebos_cmax_2dof_problem* q) {
  using namespace ebos;
  if (q == nullptr) return 0;
  return resident_2dof_ok(q) ? 1 : 0;
}

int ebos_cmax_2dof_solve_resident_f32(const ebos_cmax_2dof_problem* q, int n_iter, void* mailbox, size_t mailbox_bytes,
                                      double spin_timeout_s, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(q != nullptr && n_iter >= 0 && q->steps_done >= 0, "ebos_cmax_2dof_solve_resident: NULL problem or negative counts");
  EBOS_REQUIRE(q->theta && q->d_theta && q->exp_avg && q->exp_avg_sq && q->step && q->iwe && q->variance && q->moments && q->workspace &&
                   q->key_offsets,
               "ebos_cmax_2dof_solve_resident: NULL buffer");
  EBOS_REQUIRE(mailbox != nullptr && spin_timeout_s > 0.0, "ebos_cmax_2dof_solve_resident: NULL mailbox or no spin cap");
  if (!resident_2dof_ok(q)) return EBOS_ERR_UNSUPPORTED;
  const HaloArg ha = decode_halo(q->halo);
  const int tiles_y = (q->H + q->tile_h - 1) / q->tile_h, tiles_x = (q->W + q->tile_w - 1) / q->tile_w, n_tiles = tiles_y * tiles_x;
  const MailboxLayout m = mailbox_layout(n_tiles);
  if (mailbox_bytes < m.total) {
    set_error("ebos_cmax_2dof_solve_resident: mailbox too small (%zu < %zu)", mailbox_bytes, m.total);
    return EBOS_ERR_SCRATCH;
  }
  const size_t need = ebos_iwe_slab_workspace_bytes(q->H, q->W, q->tile_h, q->tile_w, ha.halo, 1, 0, 0);
  if (q->workspace_bytes < need) {
    set_error("ebos_cmax_2dof_solve_resident: workspace too small (%zu < %zu)", q->workspace_bytes, need);
    return EBOS_ERR_SCRATCH;
  }
  hipStream_t s = as_stream(stream);
  int rc = EBOS_ERR_UNSUPPORTED;
  if (q->tile_h == 45 && q->tile_w == 80 && ha.halo == 32) rc = resident_launch_2dof_45x80(q, q->w_variance, n_iter, mailbox, spin_timeout_s, s);
  else if (q->tile_h == 32 && q->tile_w == 32 && ha.halo == 32) rc = resident_launch_2dof_32x32(q, q->w_variance, n_iter, mailbox, spin_timeout_s, s);
  else if (q->tile_h == 32 && q->tile_w == 64 && ha.halo == 32) rc = resident_launch_2dof_32x64(q, q->w_variance, n_iter, mailbox, spin_timeout_s, s);
  if (rc != EBOS_OK) return rc;
  EBOS_CHECK_LAUNCH("ebos_cmax_2dof_solve_resident");
  return EBOS_OK;
}

int ebos_cmax_resident_status(const void* mailbox, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(mailbox != nullptr, "ebos_cmax_resident_status: NULL mailbox");
  unsigned st = 0;
  hipStream_t s = as_stream(stream);
  if (hipMemcpyAsync(&st, mailbox, sizeof(st), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("ebos_cmax_resident_status: cannot read the status word (%s)", hipGetErrorString(hipGetLastError()));
    return EBOS_ERR_LAUNCH;
  }
  st &= 255u;  // (a spill carries its iteration in the upper bits)
  if (st == RES_OK) return EBOS_OK;
  set_error("resident solve ended early: %s -- theta and the optimiser state are unchanged (after a spill: those of the iterations "
            "ebos_cmax_resident_iterations reports); run ebos_cmax_patch_solve_f32 for the rest",
            st == RES_TIMEOUT ? "a wait passed the spin cap (the grid was not co-resident)"
            : st == RES_SPILL ? "a tap left the largest LDS window"
            : st == RES_IMBALANCED ? "one tile holds far more events than the average one (the pipeline splits crowded tiles)"
                                   : "unsupported cell geometry");
  return -(100 + (int)st);
}

int ebos_cmax_resident_iterations(const void* mailbox, ebos_stream_t stream) {
  using namespace ebos;
  EBOS_REQUIRE(mailbox != nullptr, "ebos_cmax_resident_iterations: NULL mailbox");
  unsigned words[2] = {0, 0};
  hipStream_t s = as_stream(stream);
  if (hipMemcpyAsync(words, mailbox, sizeof(words), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    set_error("ebos_cmax_resident_iterations: cannot read the mailbox (%s)", hipGetErrorString(hipGetLastError()));
    return EBOS_ERR_LAUNCH;
  }
  return (int)words[1];
}

}  // extern "C"
