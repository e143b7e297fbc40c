// EXPERIMENT, NOT BUILT (round 5): the objective value of BASELINE configs[1] as ONE launch.  Measured on MI355X at
// 10 M events / 1280 x 720 (profiles/r05f_value_fused_one_launch_bench.json against r05f_value_three_launches_bench.json): kernel
// 27.2 us, step 30.7 us -- against 18.5 + 5.6 + ~1 us of kernels and a 25.7 us step for accumulate + combine + finalize.  The chain
// behind the LAST tile's accumulate pass is the same in both forms -- its slab reaches memory, the flag becomes visible, the
// neighbours load their halo, the sums of squares travel, somebody reduces them: five memory round trips of 1 - 2 us -- and a kernel
// boundary costs no more than the in-kernel hand-off that replaces it; the one-launch form adds the per-device ordering of resident
// launches (an event record and a stream wait per launch).  The resident SOLVER loop wins because it removes per-launch set-up
// that is redone every iteration (tables, cell blocks, ~900 partials read back by 256 workgroups), not because hand-offs are
// cheaper than launches.  Kept for the record; to build it again: add it to event_based_bos_amd/build.py SOURCES, restore the hook
// in launch_slab_fwd (iwe_tiled.hip) and a records section in slab_layout (git history of round 5).
// iwe_value_fused.hip -- one objective VALUE as ONE launch: warp + bilinear-splat IWE + variance of a dense-flow window
// (src/warp.py:330-342 + src/event_image_converter.py:581-620 + torch.var; BASELINE configs[1]: 10 M events at 1280 x 720).
//
// ebos_iwe_dense_slab_f32(want_variance = 1) is three launches -- accumulate (one workgroup per source tile: events -> LDS image ->
// slab), combine (slabs -> image + moment partials, ~900 workgroups), finalize (one workgroup) -- and at 10 M events the last two and
// the two kernel boundaries are 9 of the step's 26 us.  Here the accumulate pass's own workgroups finish the job:
//   P1  tile_body as in the accumulate kernel: events -> LDS image -> slab (write-through); the tile's share of sum(IWE) is taken
//       from its own LDS image in the decode pass (exact in a double: kCombineExactSum);
//   S1  one record per tile {tag, share, window, spilled?}: the all-to-all of the launch.  Behind it every slab is complete -- and so
//       is the spill image, should a tap have left the largest window;
//   P2  every workgroup assembles ITS OWN tile of the image: its part decoded from LDS, the <= 8 neighbours' from their slabs, in the
//       combine pass's order of additions (the image is the three-launch image bit for bit), stores it and sums its squares;
//   S2  a second record {tag, sum of squares} and one atomic per workgroup: the LAST workgroup to arrive reads the records, writes
//       variance and (mean, M), and leaves the counters ready for the next launch.
// Hand-off form: handoff.h.  The tag is a launch count kept in the workspace by the kernel itself (a replayed HIP graph repeats its
// arguments, not its tags).  The grid must be co-resident (one workgroup per tile, checked on the host against the occupancy; the
// launch joins the per-device order of resident launches, cmax_resident.hip); every spin is bounded: past the cap the launch
// writes NaN as the variance -- loud -- instead of hanging.
#include "iwe_tile_core.h"
#include "handoff.h"

namespace ebos {

int order_resident_launches(hipStream_t s, int workgroups, int n_cu, bool after_launch);  // cmax_resident.hip

namespace {

struct ValueArgs {
  EvPtrs ev;
  const int32_t* key_offsets;
  const float* flow;
  int H, W, tiles_y, tiles_x, omit;
  float* slabs;
  float* spill;
  float* iwe;
  unsigned long long *rec1, *rec2;   // [tiles][4], [tiles][2] granules
  unsigned* counters;                // [0] launches completed (the tag's base), [1] workgroups past S2 of the running launch
  float* out_var;
  double* moments;
  float dt_bound;
  unsigned long long cap_ticks;
};

constexpr int kVRec1 = 4, kVRec2 = 2;

// lane-wise predicate over a whole wave; bounded by the 100 MHz clock
template <typename Pred>
__device__ __forceinline__ bool value_wait(Pred&& ready, unsigned long long cap_ticks) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    if (__all(ready())) return true;
    if ((++spins & 31u) == 0u) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      if (now - t0 > cap_ticks) return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int TH, int TW, int HALO, bool DYN>
__global__ void __launch_bounds__(kBlock) iwe_value_fused_kernel(ValueArgs a) {
  constexpr int kLHmax = TH + 2 * HALO, kLWmax = TW + 2 * HALO;
  constexpr int kCells = acc_cells<TH, TW, HALO, DYN>();
  constexpr int kWaves = kBlock / kWave;
  static_assert(HALO <= TH && HALO <= TW, "only the eight neighbours' windows reach a tile");
  static_assert(TW % 4 == 0 && kCells % 2 == 0, "quads");
  extern __shared__ double s_acc[];
  __shared__ TileShared sh;
  __shared__ unsigned s_tag;
  __shared__ unsigned s_win[9];
  __shared__ int s_ok, s_any_spill, s_last;
  __shared__ double s_red[2 * kWaves];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int tile = blockIdx.x, tiles_x = a.tiles_x, n_tiles = a.tiles_y * a.tiles_x;
  const int H = a.H, W = a.W;

  // ---- P1: the accumulate kernel's work item (accumulate_tile, dense flow) ---------------------------------------------------------
  if (!DYN)
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
  if (threadIdx.x == 0) s_tag = ld_sc1(a.counters) + 1u;  // (the previous launch on this workspace has completed: stream order)
  const TileRange tr = tile_range<FMT_COMPACT>(a.key_offsets, a.ev, TH * TW, tiles_x, 1);
  CRaw pre[2];
  pre[0] = load_craw(tr.g_first + wave * kWave + lane, tr, a.ev);
  pre[1] = load_craw(tr.g_first + (wave + kWaves) * kWave + lane, tr, a.ev);
  if (DYN) {
    float mu, mv;
    tile_flow_absmax<TH, TW>(a.flow, H, W, tr.ty * TH, tr.tx * TW, mu, mv);
    for (int i = threadIdx.x; i < kCells / 2; i += kBlock) reinterpret_cast<double2*>(s_acc)[i] = make_double2(0.0, 0.0);
    tile_bound_post(mu, mv, sh.bound);
  }
  if (threadIdx.x < 2) sh.flag[threadIdx.x] = 0;
  if (threadIdx.x < 9) s_win[threadIdx.x] = 0xffffffffu;
  if (threadIdx.x == 0) {
    sh.next = 2 * kWaves;
    sh.chk = 0ull;
    s_ok = 1;
    s_any_spill = 0;
    s_last = 0;
  }
  __syncthreads();
  const Win<TH, TW, HALO, DYN> win = tile_bound_read<TH, TW, HALO, DYN>(sh.bound, a.dt_bound);
  const int lo_px = a.omit ? 1 : 0;
  const int tr0 = tr.ty * TH, tc0 = tr.tx * TW;
  OwnSum own{tr0 - win.HR(), tc0 - win.HC(), lo_px, H, W, 0.0};
  // (spill image given, no SpillEpoch word / window table: both travel in the records)
  tile_body<TH, TW, HALO, false, ACC_FX, FMT_COMPACT, false, false, DYN, false>(tr, win, a.flow, s_acc, sh, a.ev, H, W, tiles_x, 0, 0, a.slabs,
                                                                                a.spill, nullptr, 0u, nullptr, pre, NoHook{}, own);
  const double os = wave_sum(own.acc);
  if (lane == 0) s_red[wave] = os;
  drain_stores();   // slab stores and spill atomics of this wave have left
  __syncthreads();
  const unsigned tag = s_tag;
  // ---- S1 ---------------------------------------------------------------------------------------------------------------------------
  if (threadIdx.x == 0) {
    double S = 0.0;
    for (int k = 0; k < kWaves; ++k) S += s_red[k];
    unsigned long long* rec = a.rec1 + (size_t)tile * kVRec1;
    put_granules(rec, tag, S);
    st_sc1(rec + 2, ((unsigned long long)tag << 32) | (unsigned long long)(win_pack(win.HR(), win.HC()) & 0xffffu) |
                        ((unsigned long long)(sh.flag[1] ? 1u : 0u) << 16));
  }
  // while the records travel: this tile's own part of its pixels, decoded from the LDS image (what it stored to its slab)
  constexpr int kQ = (TH * (TW / 4) + kBlock - 1) / kBlock;
  float4 own_q[kQ];
  {
    const bool lds_f64 = sh.chk != 0ull;
#pragma unroll
    for (int kq = 0; kq < kQ; ++kq) {
      const int i = min((int)threadIdx.x + kq * kBlock, TH * (TW / 4) - 1);
      const int rl = i / (TW / 4), cq = i - rl * (TW / 4);
      own_q[kq] = lds_image_cells4(s_acc, win.LH(), win.P(), rl + win.HR(), cq + win.HC() / 4, lds_f64);
    }
  }
  double S_all = 0.0;
  if (wave * kWave < n_tiles) {
    const int k = wave * kWave + lane;
    const unsigned long long* rec = a.rec1 + (size_t)min(k, n_tiles - 1) * kVRec1;
    unsigned long long g[3];
    const bool ok = value_wait([&]() {
      bool all = true;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        g[j] = ld_sc1(rec + j);
        all = all && (unsigned)(g[j] >> 32) == tag;
      }
      return all;
    }, a.cap_ticks);
    if (ok && k < n_tiles) {
      S_all = granules_double(g[0], g[1]);
      const int nty = k / tiles_x, dy = nty - tr.ty, dx = k - nty * tiles_x - tr.tx;
      if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) s_win[(dy + 1) * 3 + dx + 1] = (unsigned)g[2] & 0xffffu;
      if (((unsigned)g[2] >> 16) & 1u) s_any_spill = 1;  // (benign race: every writer stores 1)
    }
    if (lane == 0 && !ok) s_ok = 0;
  }
  S_all = wave_sum(S_all);
  if (lane == 0) s_red[kWaves + wave] = S_all;
  __syncthreads();
  if (!s_ok) {  // a wait passed its cap (the grid was not co-resident): fail loudly, never hang
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.out_var) a.out_var[0] = __builtin_nanf("");
    return;
  }
  // ---- P2: this tile's pixels of the image, in the combine pass's order of additions (tile row, tile column) -----------------------
  double sq = 0.0;
  {
    const __amdgpu_buffer_rsrc_t all_slabs = slab_rsrc(a.slabs, 0xffffffffu);
    const bool spill_used = s_any_spill != 0;
#pragma unroll
    for (int kq = 0; kq < kQ; ++kq) {
      const int i = threadIdx.x + kq * kBlock;
      const bool in = i < TH * (TW / 4);
      const int ic = in ? i : 0, rl = ic / (TW / 4), cq = ic - rl * (TW / 4);
      const int r = tr0 + rl, c = tc0 + 4 * cq;
      const bool live = in && r < H && c < W;
      float4 part[9];
      bool okk[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)s_win[k]);
        const int nty = tr.ty + k / 3 - 1, ntx = tr.tx + k % 3 - 1;
        const int hr = (int)(w & 255u), hc = (int)((w >> 8) & 255u);
        const int rr = r - (nty * TH - hr), cc = c - (ntx * TW - hc), lw = TW + 2 * hc, lh = TH + 2 * hr;
        okk[k] = live && w != 0xffffffffu && (unsigned)rr < (unsigned)lh && (unsigned)cc < (unsigned)lw;
        part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k == 4) {
          part[k] = own_q[kq];
        } else if (w != 0xffffffffu && __builtin_amdgcn_ballot_w64(okk[k]) != 0ull) {
          const unsigned slab0 = (unsigned)(nty * tiles_x + ntx) * (unsigned)(kLHmax * kLWmax);
          part[k] = slab_load4(all_slabs, okk[k] ? (slab0 + (unsigned)(rr * lw + cc)) * 4u : 0u);
        }
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (okk[k]) v.x += part[k].x, v.y += part[k].y, v.z += part[k].z, v.w += part[k].w;
      if (!live) continue;
      float* px = a.iwe + (int64_t)r * W + c;
      float e4[4] = {v.x, v.y, v.z, v.w};
      if (spill_used) {  // rare: taps beyond the largest window went to the spill image with global atomics; keep it zero between calls
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < W) {
            float* sp = a.spill + (int64_t)r * W + c + k;
            const float sv = ld_sc1(sp);
            if (sv != 0.0f) {
              e4[k] += sv;
              *sp = 0.0f;
            }
          }
      }
      if ((W & 3) == 0) {
        *reinterpret_cast<float4*>(px) = make_float4(e4[0], e4[1], e4[2], e4[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k < W) px[k] = e4[k];
      }
      if (r >= lo_px && r < H - lo_px) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c + k >= lo_px && c + k < W - lo_px) sq += (double)e4[k] * (double)e4[k];
      }
    }
  }
  sq = wave_sum(sq);
  if (lane == 0) s_red[wave] = sq;
  __syncthreads();
  // ---- S2: the sum of squares, and who is last ----------------------------------------------------------------------------------------
  if (threadIdx.x == 0) {
    double Q = 0.0;
    for (int k = 0; k < kWaves; ++k) Q += s_red[k];
    put_granules(a.rec2 + (size_t)tile * kVRec2, tag, Q);
    drain_stores();
    const unsigned before = __hip_atomic_fetch_add(a.counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = before == (unsigned)n_tiles - 1u;
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: every record is there (each was stored and drained before its workgroup's atomic)
  double Q = 0.0;
  for (int k = threadIdx.x; k < n_tiles; k += kBlock) {
    const unsigned long long* rec = a.rec2 + (size_t)k * kVRec2;
    unsigned long long g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1);
    // (a record whose write-through has not landed yet -- the atomic overtook it -- is polled; bounded like every wait)
    unsigned spins = 0;
    while (((unsigned)(g0 >> 32) != tag || (unsigned)(g1 >> 32) != tag) && ++spins < (1u << 20)) g0 = ld_sc1(rec), g1 = ld_sc1(rec + 1);
    Q += ((unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag) ? granules_double(g0, g1) : __builtin_nan("");
  }
  Q = block_sum(Q, s_red);
  if (threadIdx.x == 0) {
    double S = 0.0;
    for (int k = 0; k < kWaves; ++k) S += s_red[kWaves + k];
    const double m = (double)max(H - 2 * lo_px, 0) * (double)max(W - 2 * lo_px, 0);
    const double mean = m > 0.0 ? S / m : 0.0;
    if (a.out_var) a.out_var[0] = (float)((Q - S * mean) / (m - 1.0));
    if (a.moments) {
      a.moments[0] = mean;
      a.moments[1] = m;
    }
    st_sc1(a.counters + 1, 0u);
    st_sc1(a.counters, tag);
  }
}

template <int TH, int TW, int HALO, bool DYN>
int launch_value_fused(const ValueArgs& a, hipStream_t s, int* capacity_cache) {
  auto k = iwe_value_fused_kernel<TH, TW, HALO, DYN>;
  constexpr size_t lds = (size_t)acc_cells<TH, TW, HALO, DYN>() * sizeof(double);
  if (int rc = reserve_lds(k, lds, "ebos_iwe_dense_slab (fused value)")) return rc;
  if (*capacity_cache == 0) {
    int dev = 0, n_cu = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, kBlock, lds) != hipSuccess) {
      set_error("ebos_iwe_dense_slab (fused value): cannot query the device's occupancy");
      return EBOS_ERR_LAUNCH;
    }
    *capacity_cache = n_cu * per_cu > 0 ? n_cu * per_cu : -1;
  }
  const int n_tiles = a.tiles_y * a.tiles_x;
  if (*capacity_cache < n_tiles) return EBOS_ERR_UNSUPPORTED;  // (the caller takes the three launches)
  if (int rc = order_resident_launches(s, n_tiles, *capacity_cache, false)) return rc;
  hipEvent_t t0, t1;
  if (profile_next_pair(&t0, &t1, EBOS_PROFILE_SLAB_ACCUMULATE))
    hipExtLaunchKernelGGL(k, dim3((unsigned)n_tiles), dim3(kBlock), lds, s, t0, t1, 0, a);
  else
    k<<<dim3((unsigned)n_tiles), dim3(kBlock), lds, s>>>(a);
  return order_resident_launches(s, n_tiles, *capacity_cache, true);
}

}  // namespace

static_assert(kVRec1 + kVRec2 == 6, "fused_value_section_bytes (iwe_tile_core.h) sizes the records");

// EBOS_ERR_UNSUPPORTED: not built for this configuration / the grid cannot be co-resident -- the caller enqueues the three launches
int value_fused_launch(const int32_t* grp_offsets, const uint16_t* cpix, const float* cdt, const int32_t* key_offsets, const float* flow,
                       int H, int W, int tile_h, int tile_w, int halo, bool dyn, float dt_bound, int omit, float* slabs, float* spill,
                       void* section, float* iwe, float* out_var, double* moments, hipStream_t s) {
  ValueArgs a{};
  a.ev = EvPtrs{nullptr, nullptr, nullptr, nullptr, grp_offsets, cpix, cdt, nullptr, nullptr, nullptr};
  a.key_offsets = key_offsets;
  a.flow = flow;
  a.H = H, a.W = W, a.tiles_y = (H + tile_h - 1) / tile_h, a.tiles_x = (W + tile_w - 1) / tile_w, a.omit = omit ? 1 : 0;
  a.slabs = slabs, a.spill = spill, a.iwe = iwe;
  char* sec = reinterpret_cast<char*>(section);
  a.counters = reinterpret_cast<unsigned*>(sec);
  a.rec1 = reinterpret_cast<unsigned long long*>(sec + 256);
  a.rec2 = a.rec1 + (size_t)a.tiles_y * a.tiles_x * kVRec1;
  a.out_var = out_var, a.moments = moments;
  a.dt_bound = dt_bound;
  a.cap_ticks = 200000000ull;  // 2 s of the 100 MHz clock
  static int cap[6] = {0, 0, 0, 0, 0, 0};
  if (halo != 32) return EBOS_ERR_UNSUPPORTED;
  if (tile_h == 45 && tile_w == 80) return dyn ? launch_value_fused<45, 80, 32, true>(a, s, &cap[0]) : launch_value_fused<45, 80, 32, false>(a, s, &cap[1]);
  if (tile_h == 32 && tile_w == 32) return dyn ? launch_value_fused<32, 32, 32, true>(a, s, &cap[2]) : launch_value_fused<32, 32, 32, false>(a, s, &cap[3]);
  if (tile_h == 32 && tile_w == 64) return dyn ? launch_value_fused<32, 64, 32, true>(a, s, &cap[4]) : launch_value_fused<32, 64, 32, false>(a, s, &cap[5]);
  return EBOS_ERR_UNSUPPORTED;
}

}  // namespace ebos

extern "C" int ebos_iwe_value_fused_supported(int H, int W, int tile_h, int tile_w, int halo) {
  using namespace ebos;
  if (H <= 0 || W <= 0) return 0;
  const char* e = getenv("EBOS_VALUE_FUSED");
  if (e != nullptr && e[0] == '0') return 0;
  const HaloArg ha = decode_halo(halo);
  return ha.halo == 32 && ((tile_h == 45 && tile_w == 80) || (tile_h == 32 && tile_w == 32) || (tile_h == 32 && tile_w == 64)) ? 1 : 0;
}
