// glrm_blocked.hip -- phase-aligned gather passes: the sweep family for problems whose opposing factor is far larger than the 4 MB
// L2 of an XCD and whose segments meet too few observations per LDS tile for the tiled sweeps (BASELINE config 4: 10M x 100k,
// rank 64, 100 observations per row).
//
// Why.  The one-kernel gather sweep (glrm_hip.hip) fetches one k-vector per observation from wherever the opposing factor lives:
// measured 6.5 TB/s from HBM (X, 5.1 GB) and 8.2 TB/s from the Infinity Cache (Y, 51 MB) -- the ceilings of random 512-byte reads at
// those levels (profiles/r02_ubench_gather.txt).  The same reads served by the L2 of the XCD run at ~30 TB/s.  The index lists are
// sorted, so a lane group that owns a segment walks the opposing factor front to back; if every group in flight starts at the same
// time and works at the same rate, all of them read the same ~2 MB window of the factor at any moment and that window stays in L2.
//
// How.  The pass structure of the LDS-tiled column sweep (glrm_tiled.hpp: pass over one super-tile, partials per (segment,
// super-tile), reduce in super-tile order, trial passes, decide) with the LDS tile taken out (tiled_pass<..., L2 = true>):
//   * a lane group (G lanes) owns one segment, its factor vector and gradient live in registers, its observations are consumed in
//     list order (the reference's summation order);
//   * one launch = one super-tile x one SLICE of the segments, the slice being what the chip holds at once (occupancy x CUs), so
//     all groups of a launch start together; launches are issued super-tile by super-tile;
//   * a super-tile is ~128 MB of the opposing factor (64 L2 windows): short enough that the groups have not drifted apart by the
//     end of it, long enough that the partial sums (k + 2 doubles per segment and super-tile) are a few percent of the traffic.
//     Its size is a function of (number of opposing vectors, kp) only, so the summation order does not depend on the sharding.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <utility>
#include <vector>

#include <chrono>
#include <thread>

#include "glrm_engine.hpp"
#include "glrm_tiled.hpp"

using namespace glrm;

// Waves per workgroup.  ONE: the finest grain the dispatcher can spread over the CUs and retire -- C4 Y half-step, same box, with the
// half-residency launch slices below: 138.5 (4 waves) / 134.9 (2) / 129.6 ms (1); at full-residency slices 150.7 / 146.3 / 144.4
// (profiles/r03_c4_blocked_knobs.txt).  Changes no sum (a lane group owns a segment whatever workgroup it sits in).
#ifndef GLRM_BLOCKED_NW
#define GLRM_BLOCKED_NW 1
#endif
constexpr int BNW = GLRM_BLOCKED_NW;                                       // waves per workgroup
constexpr int LOCK_CTR_WORDS = 8 * 32 + 32;                                // lockstep windows: one 128-byte line per XCD + one for the give-up count
constexpr int tile_rows_b(int kp) { return ((150 * 1024) / (kp * 8 + 16)) / 16 * 16; } // the LDS tile unit the super-tiles are counted in

int glrm_setup_blocked(glrm_handle* h) {
  h->blocked_row = h->blocked_col = 0;
  const int want = env_int("GLRM_HIP_BLOCKED", h->tiled_opt == 1 ? 0 : -1); // -1 auto, else bit0 rows, bit1 columns (glrm_options.tiled = 1: gather sweeps only)
  if (want == 0 || h->kp < 8) return GLRM_OK;
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, h->device));
  const int64_t cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const int T = tile_rows_b(h->kp);
  auto decide = [&](bool rows) -> bool {
    // Everything below is read from the WHOLE problem (m, n, k, glrm_signature, options), never from the shard, so that every
    // shard count runs the same family (the families differ in summation order).
    if (rows ? h->tiled_row : h->tiled_col) return false;          // the LDS-tiled sweep already owns this view
    if (rows && h->cached_want) return false;                      // the short rows run on the cached sweep (glrm_cached.hip)
    if (!(rows ? h->rows_sorted : h->cols_sorted)) return false;   // a group must meet the opposing factor front to back
    const int64_t nnz = rows ? h->sig.nnz_rows : h->sig.nnz_cols, nopp = rows ? h->n : h->m;
    if (nnz <= 0) return false;
    if (want > 0) return ((want >> (rows ? 0 : 1)) & 1) != 0;
    const double opp_bytes = (double)nopp * h->kp * 8;
    const int64_t nseg_glob = rows ? h->m : h->n;
    const double mean_len = (double)nnz / (double)nseg_glob;
    if (opp_bytes <= 32.0 * 1024 * 1024 || mean_len * (double)nseg_glob < 2e8) return false; // small factor: LDS tiles or plain L2 hits do better
    // observations that meet one vector of the opposing factor while the groups the chip holds (4 waves per SIMD) walk past it, per XCD
    const double resident = std::min<double>((double)nseg_glob, (double)cus * 16 * (64 / h->G));
    const double reuse = resident * mean_len / (double)nopp / 8.0;
    return reuse >= 2.0;
  };
  const bool br = decide(true), bc = decide(false);
  h->lockstep = bc && (h->G * 100 + h->R == 808 || h->G * 100 + h->R == 408) && env_int("GLRM_HIP_LOCKSTEP", 0) ? 1 : 0; // column view in lockstep windows (below); an environment switch, never the shard
  auto sups = [&](int64_t nopp, int& tps, int& nsup, bool lock) {
    const int64_t ntiles = (nopp + T - 1) / T;
    int64_t t = ((int64_t)128 * 1024 * 1024) / ((int64_t)T * h->kp * 8); // ~128 MB of the opposing factor per super-tile
    t = env_int("GLRM_HIP_BLOCKED_TPS", (int)(t < 1 ? 1 : t));
    if (lock) t = ntiles > 0 ? ntiles : 1; // one pass over the whole factor, no partial sums per super-tile
    tps = (int)t;
    nsup = (int)((ntiles + t - 1) / t);
  };
  if (br) {
    sups(h->n, h->tiles_per_sup_r, h->nsup_r, false);
    const int64_t ml1 = h->ml > 0 ? h->ml : 1;
    HIPCK(hipMalloc((void**)&h->part_r, (size_t)ml1 * h->nsup_r * (h->kp + 2) * 8));
    HIPCK(hipMalloc((void**)&h->gsum_r, (size_t)ml1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->trial_r, (size_t)ml1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->jold_r, (size_t)ml1 * 8));
    HIPCK(hipMalloc((void**)&h->active_r, (size_t)ml1 * 4));
    HIPCK(hipMalloc((void**)&h->ntrial_r, (size_t)ml1 * 4));
    h->blocked_row = 1;
  }
  if (bc) {
    sups(h->m, h->tiles_per_sup, h->nsup, h->lockstep != 0);
    const int64_t nl1 = h->nl > 0 ? h->nl : 1;
    HIPCK(hipMalloc((void**)&h->part, (size_t)nl1 * h->nsup * (h->kp + 2) * 8));
    HIPCK(hipMalloc((void**)&h->gsum, (size_t)nl1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->trialbuf, (size_t)nl1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->joldbuf, (size_t)nl1 * 8));
    HIPCK(hipMalloc((void**)&h->activebuf, (size_t)nl1 * 4));
    HIPCK(hipMalloc((void**)&h->ntrialbuf, (size_t)nl1 * 4));
    HIPCK(hipMemsetAsync(h->activebuf, 0, (size_t)nl1 * 4, h->stream)); // diverted columns are never touched by col_reduce: they must read "not searching"
    h->blocked_col = 1;
    // Skewed column lengths (power-law Omega; round 5).  A launch covers one super-tile x a slice of the columns, one lane group per column:
    // a column 100 x the mean keeps its group walking 100 x longer than the others of its launch, alone and latency bound (C4 recipe with
    // Zipf degrees: Y half-step 2.6 s against 0.13 s).  (i) The passes hand the columns out LONGEST FIRST, so the groups of a slice walk
    // lists of like length (and of like density: they also advance through the super-tile at the same rate).  (ii) Columns of at least
    // long_from = max(4 096, 2 x the whole problem's mean column length) observations leave the passes for the 8-wave gather sweep, which
    // spreads one column over 64 lane groups; it runs beside the passes on the side stream.  Both are functions of the column's own
    // length and of the whole problem's signature: shard-invariant.  Which slot a column sits in changes no sum; the diverted columns
    // add in the gather sweep's order (reported: glrm_sum_order.long_from).  Measured, C4 recipe with Zipf(0.5) degrees (995e6
    // observations, longest column 1.37e6 against a mean of 9 950; profiles/r05_c4_zipf_*): Y half-step 2 622 ms before, 207 ms with
    // long_from = 16 x mean, 168 / 154 / 151 ms with 8 / 4 / 2 x mean; the uniform recipe (no column reaches 2 x mean) is untouched.
    const int64_t mean_len = h->sig.nnz_cols / (h->n > 0 ? h->n : 1);
    h->blk_long_from = env_int("GLRM_HIP_BLOCKED_LONG_FROM", 0) > 0 ? env_int("GLRM_HIP_BLOCKED_LONG_FROM", 0)
                                                                     : std::max<int64_t>(4096, 2 * mean_len);
    if (h->lockstep) h->blk_long_from = 0;
    if (h->nl > 0 && h->blk_long_from > 0) {
      std::vector<int64_t> ptr((size_t)h->nl + 1);
      HIPCK(hipMemcpyAsync(ptr.data(), h->colptr, ((size_t)h->nl + 1) * 8, hipMemcpyDeviceToHost, h->stream));
      HIPCK(hipStreamSynchronize(h->stream));
      std::vector<int32_t> shortl, longl;
      for (int64_t s = 0; s < h->nl; ++s) (ptr[s + 1] - ptr[s] >= h->blk_long_from ? longl : shortl).push_back((int32_t)s);
      std::stable_sort(shortl.begin(), shortl.end(), [&](int32_t x, int32_t y) { return ptr[x + 1] - ptr[x] > ptr[y + 1] - ptr[y]; });
      h->blk_nshort_c = (int64_t)shortl.size();
      h->blk_nlong_c = (int64_t)longl.size();
      HIPCK(hipMalloc((void**)&h->blk_perm_c, std::max<size_t>(1, shortl.size()) * 4));
      HIPCK(hipMemcpyAsync(h->blk_perm_c, shortl.data(), shortl.size() * 4, hipMemcpyHostToDevice, h->stream));
      if (!longl.empty()) {
        HIPCK(hipMalloc((void**)&h->blk_long_c, longl.size() * 4));
        HIPCK(hipMemcpyAsync(h->blk_long_c, longl.data(), longl.size() * 4, hipMemcpyHostToDevice, h->stream));
        if (!h->side_stream) {
          HIPCK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
          HIPCK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
          HIPCK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
      }
      HIPCK(hipStreamSynchronize(h->stream)); // the lists are locals
    }
  }
  if ((br || bc) && !h->nactive) HIPCK(hipMalloc((void**)&h->nactive, 4));
  if (h->lockstep && !h->lock_ctr) HIPCK(hipMalloc((void**)&h->lock_ctr, LOCK_CTR_WORDS * 4));
  return GLRM_OK;
}

// segments one launch may cover: what the chip holds at once (every group of the launch then starts its walk together)
template <typename K>
static int64_t slice_capacity(K kernel, int device, int spb) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, BNW * 64, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
  hipDeviceProp_t prop;
  int cus = 256;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const int64_t cap = (int64_t)nb * cus * spb;
  // percent of the residency one launch covers.  Half: the next launch's workgroups fill the CUs the current one's stragglers have left
  // (launches of one stream overlap at their tails), which a slice of exactly the residency cannot do -- C4 Y half-step 143.2 -> 138.4 ms
  // with four-wave workgroups (25 / 33 / 66 / 100 / 200 %: 149.0 / 148.1 / 141.6 / 143.2 / 144.1), and with one-wave workgroups
  // 25 / 33 / 40 / 50 / 60 / 75 / 100 %: 139.7 / 137.0 / 135.6 / 131.0 / 133.0 / 136.9 / 144.4 (profiles/r03_c4_blocked_knobs.txt).
  // Changes no sum.
  const int pct = env_int("GLRM_HIP_BLOCKED_FILL", 50);
  return std::max<int64_t>(spb, cap * pct / 100 / spb * spb);
}

template <int G, int R, int LOSS, bool GRAD>
static int launch_blocked_inst(glrm_handle* h, TiledArgs a, bool rows) {
  constexpr int KP = G * R, T = tile_rows_b(KP), SPB = BNW * (64 / G);
  auto kernel = tiled_col_pass_kernel<G, R, BNW, T, LOSS, GRAD, true>;
  // per handle, view and pass kind: the row passes (LOSS_PER_OBS* instantiations of a heterogeneous model) and the column passes
  // (LOSS_SEGMENT*) of one handle are different kernels with their own occupancy; the CU count is the handle's device's, the fill
  // percentage the environment's at the handle's first sweep
  int64_t& cap_slot = h->blocked_cap[rows ? 0 : 1][GRAD ? 0 : 1];
  if (cap_slot == 0) cap_slot = slice_capacity(kernel, h->device, SPB);
  const int64_t cap = cap_slot;
  const int64_t nseg = a.npass > 0 ? a.npass : a.nseg;
  // equal slices: ceil(nseg / cap) launches per super-tile, all of the same size (a last slice of a few percent of the others is a launch
  // that cannot fill the chip)
  const int64_t nslices = (nseg + cap - 1) / cap;
  const int64_t per = nslices > 0 ? ((nseg + nslices - 1) / nslices + SPB - 1) / SPB * SPB : cap;
  // glrm_hip_step_y_arrival: the gradient pass of the column view walks the super-tiles in the order their rows of X arrive, each behind
  // the events of the blocks it touches (h->sup_order, glrm_run_blocked); the partial sums are per (segment, super-tile) and col_reduce
  // adds them in super-tile order, so the launch order changes no bit
  const bool ordered = GRAD && !rows && h->n_arrival > 0 && (int)h->sup_order.size() == a.nsup;
  const int64_t rows_per_sup = (int64_t)a.tiles_per_sup * T;
  auto launch_sup = [&](int sup) -> int { // one super-tile: behind the events of the blocks it touches (ordered), slice by slice
    if (ordered) {
      const int rcw = glrm_arrival_wait(h, sup * rows_per_sup, std::min<int64_t>((sup + 1) * rows_per_sup, a.n_other));
      if (rcw) return rcw;
    }
    for (int64_t s0 = 0; s0 < nseg; s0 += per) {
      a.sup_fixed = sup;
      a.seg_begin = s0;
      a.nseg_slice = std::min(per, nseg - s0);
      hipLaunchKernelGGL(kernel, dim3((unsigned)((a.nseg_slice + SPB - 1) / SPB)), dim3(BNW * 64), 0, h->stream, a);
    }
    return GLRM_OK;
  };
  std::vector<int> pend;
  if (ordered) pend = h->sup_order;
  else for (int si = 0; si < a.nsup; ++si) pend.push_back(si);
  if (ordered && !h->arrival_static && env_int("GLRM_HIP_ARRIVAL_DYNAMIC", 1)) {
    // TRUE arrival order.  The fixed order above is the order blocks arrive in when every peer keeps pace; behind a lagging peer the
    // in-order stream would stand in front of its block while super-tiles further down the list are ready.  So the host enqueues a
    // super-tile only once the events of ALL its blocks have fired (hipEventQuery), the ready ones in list order, and polls for the
    // rest: the stream never stands in a wait while work is ready.  An event that cannot be queried (or 5 s without any progress) hands
    // the remaining super-tiles -- and every later call on this handle -- to the in-stream waits below.  Which order the super-tiles run in changes no bit.
    std::vector<char> fired((size_t)h->n_arrival, 0);
    bool unknown = false;
    auto ready = [&](int sup) {
      const int64_t lo = sup * rows_per_sup, hi = std::min<int64_t>((sup + 1) * rows_per_sup, a.n_other);
      for (int b = 0; b < h->n_arrival; ++b) {
        const glrm_arrival& blk = h->arrival[b];
        if (fired[b] || !blk.event || blk.end <= lo || blk.begin >= hi) continue;
        const hipError_t q = hipEventQuery((hipEvent_t)blk.event);
        if (q == hipSuccess) { fired[b] = 1; continue; }
        (void)hipGetLastError();
        if (q != hipErrorNotReady) unknown = true;
        return false;
      }
      return true;
    };
    auto t_progress = std::chrono::steady_clock::now();
    // the host polls where the stream used to wait: with glrm_options.profile the time in which NO super-tile was ready is booked as exposed
    // exchange time (glrm_kernel_stats.ms_wait_y, ADVICE r5: in true arrival order the in-stream waits are ~0 and this time was counted nowhere).
    // An upper bound on what the device stood idle for: super-tiles enqueued earlier may still be running while the host finds nothing ready.
    double idle_s = 0.0;
    while (!pend.empty() && !unknown) {
      bool progressed = false;
      for (size_t i = 0; i < pend.size() && !unknown;) {
        if (!ready(pend[i])) { ++i; continue; }
        const int rcl = launch_sup(pend[i]);
        if (rcl) return rcl;
        pend.erase(pend.begin() + (long)i);
        progressed = true;
      }
      const auto now = std::chrono::steady_clock::now();
      if (progressed) t_progress = now;
      else if (std::chrono::duration<double>(now - t_progress).count() > 5.0) break;
      else {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        idle_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - now).count();
      }
    }
    if (h->profile) h->ms_wait += idle_s * 1e3;
    if (!pend.empty()) h->arrival_static = true; // gave up: this handle keeps to the in-stream waits (no 5 s stall per iteration)
  }
  for (int sup : pend) {
    const int rcl = launch_sup(sup);
    if (rcl) return rcl;
  }
  HIPCK(hipGetLastError());
  return GLRM_OK;
}


// ------------------------------------------------------------------------------------------------ lockstep windows (round 3)
// The launches above bound the DRIFT of the groups by a launch boundary every 128 MB of the opposing factor; inside a super-tile the groups
// of an XCD spread over far more than its 4 MB of L2 (TCC hit rate of the C4 column passes: 19 %, profiles/r03_c4_tcc_hit_rate.txt), and
// the passes run at what a miss stream reaches (8 TB/s).  Here the whole pass is ONE persistent kernel: a lane group keeps its column's
// gradient in registers from the first to the last row (no partial sums per super-tile at all: nsup = 1), walks the factor in windows of
// ~2 MB, and after every window the workgroups that share an XCD (block b runs on XCD b % 8) meet at a counter, so that the ~3 000 columns
// an XCD holds read each window out of ITS L2 while it is there (every row of a window is met ~3 times per XCD and residency round at C4).
// The meeting is a performance device only -- no data passes between workgroups, a late or lost arrival costs cache hits, never a result:
// thread 0 adds 1 to its XCD's counter and polls it at most la.spin times; a workgroup whose poll runs out stops meeting (and, not
// arriving any more, lets every other workgroup of its XCD run out once, in parallel): the kernel cannot hang.  Sums: a column's gradient
// terms are added in list order from its first row to its last, its losses in list order inside a window and the windows in order -- a
// function of (m, kp, window size) only.
struct LockArgs {
  unsigned int* ctr;
  int wt;      // LDS-tile units (tile_rows_b rows) per window
  int nwin;    // windows per pass
  int nrounds; // residency rounds: workgroup b takes segments ((r * gridDim.x + b) * SPB ...) in round r
  int spin;    // polls before a workgroup gives up meeting; 0: no meeting at all (sparse trial rounds)
};

template <int G, int R, int NW, int LOSS, bool GRAD>
__global__ void __launch_bounds__(NW * 64, 16 / NW) lockstep_col_pass_kernel(const TiledArgs a, const LockArgs la) {
  constexpr int KP = G * R, NGW = 64 / G, SPB = NW * NGW, PSTRIDE = KP + 2, T = tile_rows_b(KP);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % G, gi = lane / G;
  const unsigned xcd = blockIdx.x & 7u;
  const unsigned members = (gridDim.x - xcd + 7u) / 8u; // workgroups of this launch on the same XCD
  unsigned int* ctr = la.ctr + xcd * 32;
  bool meeting = la.spin > 0; // thread 0's
  unsigned step = 0;
  for (int r = 0; r < la.nrounds; ++r) {
    const int64_t slot = ((int64_t)r * gridDim.x + blockIdx.x) * SPB + wave * NGW + gi;
    bool have = slot < a.nseg;
    const int64_t seg = slot;
    if (!GRAD && have) have = a.active[seg] != 0;
    if (la.spin == 0 && !__syncthreads_or(have ? 1 : 0)) continue; // sparse round: nothing to evaluate in this column group
    const int64_t beg = have ? a.ptr[seg] : 0, end = have ? a.ptr[seg + 1] : 0;
    const int64_t gseg = a.own_offset + (have ? seg : 0);
    const double2* xp = reinterpret_cast<const double2*>(GRAD ? a.own + gseg * KP : a.trial + (have ? seg : 0) * (int64_t)KP);
    Vec<G, R> x, gacc; // the gradient is carried through the windows (tiled_pass<..., ACC = true>): list order from the first row to the last
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      x.v[i] = have ? xp[i * G + j] : make_double2(0.0, 0.0);
      gacc.v[i] = make_double2(0.0, 0.0);
    }
    LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
    if constexpr (loss_mode(LOSS) != 2) segloss = load_loss(a.losses, (a.loss_by_segment && have) ? gseg : 0);
    int64_t pos = beg;
    double Jacc = 0.0;
    for (int w = 0; w < la.nwin; ++w) {
      double J;
      tiled_pass<G, R, NW, T, LOSS, GRAD, true, 0, false, true>(a, nullptr, x, gacc, J, have, pos, end, w * la.wt, (w + 1) * la.wt, segloss, lane, j);
      Jacc += J;
      if (la.spin > 0) { // meet the other workgroups of the XCD at the end of the window
        ++step;
        if (NW > 1) __syncthreads();
        if (threadIdx.x == 0 && meeting) {
          __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned target = step * members;
          int polls = 0;
          while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            if (++polls > la.spin) {
              meeting = false;
              __hip_atomic_fetch_add(la.ctr + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // reported by the host
              break;
            }
            __builtin_amdgcn_s_sleep(4);
          }
        }
        if (NW > 1) __syncthreads();
      }
    }
    if (have) {
      double* p = a.part + (int64_t)seg * a.nsup * PSTRIDE; // nsup = 1
      if (GRAD) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) *reinterpret_cast<double2*>(p + i * 2 * G + 2 * j) = gacc.v[i];
      }
      if (j == 0) p[KP] = Jacc;
    }
  }
}

template <int G, int R, int LOSS, bool GRAD>
static int launch_lockstep_inst(glrm_handle* h, const TiledArgs& a, bool meet) {
  constexpr int KP = G * R, T = tile_rows_b(KP), NW = 4, SPB = NW * (64 / G);
  auto kernel = lockstep_col_pass_kernel<G, R, NW, LOSS, GRAD>;
  static std::atomic<int64_t> grid_cache{0};
  int64_t gridmax = grid_cache.load(std::memory_order_relaxed);
  if (gridmax == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, NW * 64, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    gridmax = std::max<int64_t>(8, (int64_t)nb * cus / 8 * 8); // every workgroup of the launch resident at once
    grid_cache.store(gridmax, std::memory_order_relaxed);
  }
  const int64_t need = (a.nseg + SPB - 1) / SPB;
  if (need <= 0) return GLRM_OK;
  LockArgs la{};
  la.nrounds = (int)((need + gridmax - 1) / gridmax);
  const int64_t grid = std::min<int64_t>(gridmax, ((need + la.nrounds - 1) / la.nrounds + 7) / 8 * 8); // equal rounds
  const int64_t ntiles = (a.n_other + T - 1) / T;
  const int wt_default = (int)std::max<int64_t>(1, ((int64_t)2 * 1024 * 1024) / ((int64_t)T * KP * 8)); // ~2 MB of the factor per window
  la.wt = std::max(1, env_int("GLRM_HIP_LOCKSTEP_WT", wt_default));
  la.nwin = (int)((ntiles + la.wt - 1) / la.wt);
  la.spin = meet ? std::max(1, env_int("GLRM_HIP_LOCKSTEP_SPIN", 4000)) : 0;
  la.ctr = h->lock_ctr;
  if (meet) HIPCK(hipMemsetAsync(h->lock_ctr, 0, LOCK_CTR_WORDS * 4, h->stream));
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NW * 64), 0, h->stream, a, la);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

template <int G, int R>
static int launch_lockstep_layout(glrm_handle* h, int loss, bool grad, const TiledArgs& a, bool meet) {
#define GLRM_LS(LOSSV) (grad ? launch_lockstep_inst<G, R, LOSSV, true>(h, a, meet) : launch_lockstep_inst<G, R, LOSSV, false>(h, a, meet))
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return GLRM_LS(0);
    case LOSS_SEGMENT: return GLRM_LS(1);
    case LOSS_SEGMENT_NOTRIG: return GLRM_LS(3);
    case LOSS_PER_OBS_NOTRIG: return GLRM_LS(4);
    default: return GLRM_LS(2);
  }
#undef GLRM_LS
}

static int launch_lockstep(glrm_handle* h, int loss, bool grad, const TiledArgs& a, bool meet) {
  switch (h->G * 100 + h->R) {
    case 408: return launch_lockstep_layout<4, 8>(h, loss, grad, a, meet);
    case 808: return launch_lockstep_layout<8, 8>(h, loss, grad, a, meet);
    default: return fail(GLRM_ERR_UNSUPPORTED, "no lockstep pass kernel for lane layout G=%d R=%d", h->G, h->R);
  }
}

template <int G, int R>
static int launch_blocked_layout(glrm_handle* h, int loss, bool grad, const TiledArgs& a, bool rows) {
#define GLRM_BL(LOSSV) (grad ? launch_blocked_inst<G, R, LOSSV, true>(h, a, rows) : launch_blocked_inst<G, R, LOSSV, false>(h, a, rows))
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return GLRM_BL(0);
    case LOSS_SEGMENT: return GLRM_BL(1);
    case LOSS_SEGMENT_NOTRIG: return GLRM_BL(3);
    case LOSS_PER_OBS_NOTRIG: return GLRM_BL(4);
    default: return GLRM_BL(2);
  }
#undef GLRM_BL
}

static int launch_blocked(glrm_handle* h, int loss, bool grad, const TiledArgs& a, bool rows) {
  switch (h->G * 100 + h->R) {
    case 402: return launch_blocked_layout<4, 2>(h, loss, grad, a, rows);
    case 404: return launch_blocked_layout<4, 4>(h, loss, grad, a, rows);
    case 408: return launch_blocked_layout<4, 8>(h, loss, grad, a, rows);
    case 808: return launch_blocked_layout<8, 8>(h, loss, grad, a, rows);
    case 1608: return launch_blocked_layout<16, 8>(h, loss, grad, a, rows);
    default: return fail(GLRM_ERR_UNSUPPORTED, "no phase-aligned pass kernel for lane layout G=%d R=%d", h->G, h->R);
  }
}

int glrm_run_blocked(glrm_handle* h, bool rows, int loss, int loss_by_segment, double min_stepsize, int eval_only) {
  TiledArgs a{};
  a.nseg = rows ? h->ml : h->nl;
  a.ptr = rows ? h->rowptr : h->colptr;
  a.idx = rows ? h->colidx : h->rowidx;
  a.vals = rows ? h->rowvals : h->colvals;
  a.own = rows ? h->X : h->Y;
  a.own_offset = rows ? h->rb : h->cb;
  a.other = rows ? h->Y : h->X;
  a.n_other = rows ? h->n : h->m;
  a.alpha = rows ? h->alpharow : h->alphacol;
  a.obj = rows ? nullptr : h->objcol;
  a.losses = h->losses;
  a.loss_by_segment = loss_by_segment;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.k = h->k;
  a.min_stepsize = min_stepsize;
  a.trials = rows ? h->trials_r : h->trials_c;
  a.accepts = rows ? h->accepts_r : h->accepts_c;
  a.eval_only = eval_only;
  a.fixed_alpha = eval_only ? 0.0 : h->fixed_alpha;
  a.nsup = rows ? h->nsup_r : h->nsup;
  a.tiles_per_sup = rows ? h->tiles_per_sup_r : h->tiles_per_sup;
  a.part = rows ? h->part_r : h->part;
  a.gsum = rows ? h->gsum_r : h->gsum;
  a.trial = rows ? h->trial_r : h->trialbuf;
  a.jold = rows ? h->jold_r : h->joldbuf;
  a.active = rows ? h->active_r : h->activebuf;
  a.ntrial = rows ? h->ntrial_r : h->ntrialbuf;
  a.nactive = h->nactive;
  if (!rows && h->blk_perm_c) { // length-sorted slots; the columns at or above long_from run on the gather sweep (run_sweep, glrm_hip.hip)
    a.segperm = h->blk_perm_c;
    a.npass = h->blk_nshort_c;
    a.long_from = h->blk_long_from;
    if (a.npass == 0) { // every local column is diverted: nothing for the passes (npass = 0 would mean "all")
      HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
      return GLRM_OK;
    }
  }
  if (rows && h->rng_e >= 0) { // glrm_hip_step_x_range: local rows [rng_b, rng_e)
    const int64_t s0 = h->rng_b;
    a.nseg = h->rng_e - s0;
    if (a.nseg <= 0) return GLRM_OK;
    a.ptr += s0; a.alpha += s0; a.own_offset += s0;
    if (!a.reg_single) a.regs += s0;
    a.trials += s0; a.accepts += s0;
    a.part += s0 * (int64_t)a.nsup * (h->kp + 2); a.gsum += s0 * (int64_t)h->kp; a.trial += s0 * (int64_t)h->kp;
    a.jold += s0; a.active += s0; a.ntrial += s0;
  }
  int rc;
  HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
  h->sup_order.clear();
  if (!rows && h->n_arrival > 0 && !h->lockstep && !eval_only) {
    // a super-tile can run once the LAST (in the host's order) of the blocks it touches is there: sort the super-tiles by that position
    const int64_t rows_per_sup = (int64_t)a.tiles_per_sup * tile_rows_b(h->kp);
    std::vector<std::pair<int, int>> key((size_t)a.nsup);
    for (int sup = 0; sup < a.nsup; ++sup) {
      const int64_t lo = sup * rows_per_sup, hi = std::min<int64_t>((sup + 1) * rows_per_sup, a.n_other);
      int last = 0;
      for (int b = 0; b < h->n_arrival; ++b)
        if (h->arrival[b].begin < hi && h->arrival[b].end > lo && h->arrival[b].event) last = std::max(last, b + 1);
      key[sup] = {last, sup};
    }
    std::sort(key.begin(), key.end());
    for (auto& kv : key) h->sup_order.push_back(kv.second);
  }
  const bool lock = !rows && h->lockstep; // the same kernel in every pass (one summation order); it meets at the windows while most columns take part
  if (lock) { if ((rc = launch_lockstep(h, loss, true, a, true))) return rc; }
  else if ((rc = launch_blocked(h, loss, true, a, rows))) return rc;   // gradient + loss partials, super-tile by super-tile
  glrm_launch_col_small(h->kp, 0, a, h->stream);                 // reduce in super-tile order, J_old, first trial point
  HIPCK(hipGetLastError());
  if (eval_only || a.fixed_alpha > 0.0) return GLRM_OK;
  constexpr int MAX_ROUNDS = 4096; // see glrm_run_tiled: a guard, never a silent cut
  for (int round = 0;; ++round) {
    if (round == MAX_ROUNDS) return fail(GLRM_ERR_INVALID, "line search still running after %d rounds (min_stepsize %g)", MAX_ROUNDS, min_stepsize);
    unsigned int nact = 0;
    HIPCK(hipMemcpyAsync(&nact, h->nactive, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (nact == 0) break;
    HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
    if (lock) { if ((rc = launch_lockstep(h, loss, false, a, (int64_t)nact * 4 >= a.nseg))) return rc; }
    else if ((rc = launch_blocked(h, loss, false, a, rows))) return rc; // loss partials at the trial points of the searching segments
    glrm_launch_col_small(h->kp, 1, a, h->stream);               // accept / shrink / give up, next trial point
    HIPCK(hipGetLastError());
  }
  return GLRM_OK;
}
