// glrm_lane.hip -- host side + instantiations of the lane-per-segment LDS-tiled passes (kernels and rationale: glrm_lane.hpp).
//
// A side (rows / columns) of a handle runs this family instead of the four-lane tiled kernels when glrm_setup_tiled chose the LDS tiles for it
// and (all functions of the WHOLE problem's signature and of (k, losses, options), never of the shard -- the two families add in different
// orders): padded rank 32; one loss descriptor per segment or per model, or -- rows of a model with a loss per column -- at most 256 distinct
// descriptors in the model; at most 2e9 (columns) / 6e9 (rows) observations in the view (the SELL copy is 12 B x its padding per observation on
// top of the lists; GLRM_HIP_LANE_MAX_NNZ_M / _ROWS_M).
// The half-step is the pass machinery of the tiled column sweep on BOTH sides: gradient pass -> col_reduce (J_old, first trial point, list of
// searching segments) -> rounds of (trial pass, col_decide); rows run it with ONE super-tile (nothing is re-added).  The first trial of a
// half-step walks the SELL layout over the full grid (idle segments masked); later rounds run whichever of the kernel's three forms fits
// the fraction of segments that still searches (glrm_run_lane): the same sums in the same order, so which form ran changes no bit.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "glrm_engine.hpp"
#include "glrm_lane.hpp"

using namespace glrm;

namespace {

constexpr int LANE_NW = 8; // 8 waves x 64 segments per workgroup, 256 VGPRs per lane
constexpr int lane_tile_rows(int kp) { return ((150 * 1024) / (kp * 8 + 16)) / 16 * 16; } // = tile_rows_c(kp, 1): the order unit of the lists

template <int LOSS, bool GRAD, int FORM, bool COMPACT = false>
int launch_lane_inst(const TiledArgs& a, const LaneArgs& la, int64_t nblocks, hipStream_t st) {
  constexpr int KP = 32, T = lane_tile_rows(KP);
  const int LDSB = T * KP * 8 + (loss_mode(LOSS) == 2 ? a.n_udesc * 32 : 0);
  auto k = lane_pass_kernel<KP, LANE_NW, T, LOSS, GRAD, FORM, COMPACT>;
  HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); // (per device: a process may drive several)
  const unsigned gx = (unsigned)((nblocks + LANE_NW - 1) / LANE_NW);
  if (gx == 0) return GLRM_OK;
  hipLaunchKernelGGL(k, dim3(gx, (unsigned)(a.nsup_launch > 0 ? a.nsup_launch : a.nsup)), dim3(LANE_NW * 64), LDSB, st, a, la);
  return GLRM_OK;
}

template <bool GRAD, int FORM>
int launch_lane_loss(int loss, const TiledArgs& a, const LaneArgs& la, int64_t nblocks, hipStream_t st) {
  if constexpr (FORM == 0) {
    if (la.off16) { // the compact form of the stream (sides without a descriptor id in the offset word)
      switch (loss) {
        case LOSS_QUAD_UNIFORM: return launch_lane_inst<0, GRAD, 0, true>(a, la, nblocks, st);
        case LOSS_SEGMENT: return launch_lane_inst<1, GRAD, 0, true>(a, la, nblocks, st);
        case LOSS_SEGMENT_NOTRIG: return launch_lane_inst<3, GRAD, 0, true>(a, la, nblocks, st);
        default: return fail(GLRM_ERR_UNSUPPORTED, "lane-per-segment passes, compact stream: no kernel for loss variant %d", loss);
      }
    }
  } else if constexpr (FORM == 2 && !GRAD) {
    if (la.off16) {
      if (!la.sval) return fail(GLRM_ERR_UNSUPPORTED, "lane-per-segment passes: the gathered form needs the step bases of the compact stream");
      switch (loss) {
        case LOSS_QUAD_UNIFORM: return launch_lane_inst<0, false, 2, true>(a, la, nblocks, st);
        case LOSS_SEGMENT: return launch_lane_inst<1, false, 2, true>(a, la, nblocks, st);
        case LOSS_SEGMENT_NOTRIG: return launch_lane_inst<3, false, 2, true>(a, la, nblocks, st);
        default: return fail(GLRM_ERR_UNSUPPORTED, "lane-per-segment passes, compact stream: no kernel for loss variant %d", loss);
      }
    }
  }
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return launch_lane_inst<0, GRAD, FORM>(a, la, nblocks, st);
    case LOSS_SEGMENT: return launch_lane_inst<1, GRAD, FORM>(a, la, nblocks, st);
    case LOSS_SEGMENT_NOTRIG: return launch_lane_inst<3, GRAD, FORM>(a, la, nblocks, st);
    case LOSS_PER_OBS: return launch_lane_inst<2, GRAD, FORM>(a, la, nblocks, st);
    case LOSS_PER_OBS_NOTRIG: return launch_lane_inst<4, GRAD, FORM>(a, la, nblocks, st);
    default: return fail(GLRM_ERR_UNSUPPORTED, "lane-per-segment passes: no kernel for loss variant %d", loss);
  }
}

// a wave per (segment, super-tile), no tile: the tail rounds (glrm_lane.hpp: lane_tail_kernel)
constexpr int LANE_TAIL_CAP = 2048; // terms it parks in LDS at a time
int launch_lane_tail(int loss, const TiledArgs& a, const int32_t* list, int64_t nact, hipStream_t st) {
  constexpr int KP = 32, T = lane_tile_rows(KP);
  nact *= a.nsup; // waves of the round
  const unsigned gx = (unsigned)((nact + 1) / 2);
  switch (loss) {
    case LOSS_QUAD_UNIFORM: hipLaunchKernelGGL((lane_tail_kernel<KP, 0, LANE_TAIL_CAP>), dim3(gx), dim3(128), 0, st, a, list, nact, T); break;
    case LOSS_SEGMENT: hipLaunchKernelGGL((lane_tail_kernel<KP, 1, LANE_TAIL_CAP>), dim3(gx), dim3(128), 0, st, a, list, nact, T); break;
    case LOSS_SEGMENT_NOTRIG: hipLaunchKernelGGL((lane_tail_kernel<KP, 3, LANE_TAIL_CAP>), dim3(gx), dim3(128), 0, st, a, list, nact, T); break;
    case LOSS_PER_OBS: hipLaunchKernelGGL((lane_tail_kernel<KP, 2, LANE_TAIL_CAP>), dim3(gx), dim3(128), 0, st, a, list, nact, T); break;
    case LOSS_PER_OBS_NOTRIG: hipLaunchKernelGGL((lane_tail_kernel<KP, 4, LANE_TAIL_CAP>), dim3(gx), dim3(128), 0, st, a, list, nact, T); break;
    default: return fail(GLRM_ERR_UNSUPPORTED, "lane-per-segment passes: no tail kernel for loss variant %d", loss);
  }
  return GLRM_OK;
}

// the small kernels of the pass machinery in the two-lane layout the family's sums are reported in
void launch_small(int which, const TiledArgs& a, hipStream_t st) {
  constexpr int G = 2, R = 16;
  const unsigned gx = (unsigned)((a.nseg + 4 * (64 / G) - 1) / (4 * (64 / G)));
  if (which == 0) hipLaunchKernelGGL((col_reduce_kernel<G, R>), dim3(gx), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((col_decide_kernel<G, R>), dim3(gx), dim3(256), 0, st, a);
}

} // namespace

// (a loss per observation -- rows of a model with a loss per column -- needs the one-byte descriptor ids: at most 256 distinct descriptors)
bool glrm_lane_loss_ok(const glrm_handle* h, int loss) {
  return loss == LOSS_QUAD_UNIFORM || loss == LOSS_SEGMENT || loss == LOSS_SEGMENT_NOTRIG ||
         ((loss == LOSS_PER_OBS || loss == LOSS_PER_OBS_NOTRIG) && h->rowdescid && h->n_udesc > 0 && h->n_udesc <= 256);
}

// at most 256 distinct loss descriptors in the model (what TiledArgs::descid can number)?  Host-side, from the descriptors alone.
bool glrm_lane_few_descriptors(const glrm_handle* h) {
  std::vector<glrm_loss> uniq;
  for (const glrm_loss& l : h->losses_h) {
    size_t u = 0;
    for (; u < uniq.size(); ++u)
      if (uniq[u].kind == l.kind && uniq[u].dim == l.dim && uniq[u].scale == l.scale && uniq[u].p0 == l.p0 && uniq[u].p1 == l.p1) break;
    if (u == uniq.size()) {
      if (uniq.size() == 256) return false;
      uniq.push_back(l);
    }
  }
  return true;
}

// what does not depend on the buffers: padded rank, tile configuration, losses, view size -- all of the WHOLE problem
bool glrm_lane_wants(const glrm_handle* h, bool rows) {
  const int want = env_int("GLRM_HIP_LANE", 3); // bit0 rows, bit1 columns
  if (!((want >> (rows ? 0 : 1)) & 1) || h->kp != 32 || !h->tile_cfg || h->tile_lw > 0 || h->multi || h->dense || h->sum_order_opt) return false;
  if (lane_tile_rows(h->kp) != h->order_unit) return false; // (loader-wave experiments order the lists by half tiles)
  // a loss per column: the row view meets a descriptor per observation -- through one-byte ids, i.e. at most 256 distinct descriptors.
  // Default since session r6_33 (GLRM_HIP_LANE_PER_OBS=0 keeps such rows on the four-lane kernels): with the trial rounds read out of the
  // SELL layout (glrm_run_lane) the C5 recipe's X half-step is 42.5 against 46.6 ms at 1M rows and 194 against 215 ms at its stated size.
  // (Session r6_19/20, rounds on the CSR form: 64.4 against 45.8 ms -- opt-in then.)
  if (rows && h->n_losses > 1 && (!env_int("GLRM_HIP_UDESC", 1) || !env_int("GLRM_HIP_LANE_PER_OBS", 1) || !glrm_lane_few_descriptors(h))) return false;
  // the SELL copy is 12 B x its padding per observation on top of the lists: a memory policy per view (rows: the copy REPLACES the kind-grouped
  // private row view of such models, and rows pad little; columns pad x 1.5-1.8).  C5 at its stated size: 5e9 observations per view
  // (columns beyond 2e9 observations: the COMPACT form of the stream -- 2-byte offsets, unpadded values, 11.5 instead of 21 B per observation at
  // x 1.76 padding; GLRM_HIP_LANE_COMPACT = 0 never, 1 every side it serves, 2 (default) views beyond 2e9 observations)
  const int64_t max_nnz = (int64_t)env_int(rows ? "GLRM_HIP_LANE_MAX_NNZ_ROWS_M" : "GLRM_HIP_LANE_MAX_NNZ_M", rows || env_int("GLRM_HIP_LANE_COMPACT", 2) ? 6000 : 2000) * 1000000ll;
  return (rows ? h->sig.nnz_rows : h->sig.nnz_cols) <= max_nnz;
}

// finalize, after glrm_setup_tiled has chosen the LDS tiles and built the slot permutations: decide the family per side and build its SELL layout
int glrm_setup_lane(glrm_handle* h) {
  h->lane[0] = h->lane[1] = 0;
  int64_t gseg_max = 0;
  const int T = lane_tile_rows(h->kp);
  hipStream_t st = h->stream;
  for (int side = 0; side < 2; ++side) {
    const bool rows = side == 0;
    if (!glrm_lane_wants(h, rows)) continue;
    if (rows ? !(h->tiled_row && !h->row_split && (h->tile_rounds & 1) && h->actlist && h->part_r) : !h->tiled_col) continue;
    if (rows && h->n_losses > 1 && !(h->rowdescid && h->n_udesc > 0)) continue;
    h->lane[side] = 1;
    const int64_t nseg = rows ? h->ml : h->nl;
    const int64_t nslots = rows ? nseg : (h->blk_nlong_c > 0 ? h->blk_nshort_c : nseg);
    const int64_t nother = rows ? h->n : h->m;
    const int ntiles = (int)((nother + T - 1) / T);
    const int64_t nwb = (nslots + 63) / 64;
    h->lane_nwb[side] = nwb;
    h->lane_ntiles[side] = ntiles;
    if (nwb == 0 || ntiles == 0) continue;
    const int64_t* ptr = rows ? h->rowptr : h->colptr;
    const int32_t* idx = rows ? h->colidx : h->rowidx;
    const double* vals = rows ? h->rowvals : h->colvals;
    const int32_t* perm = rows ? h->rowperm : h->colperm;
    const int64_t ncell = nwb * ntiles;
    if (ncell > (int64_t)1 << 30) { h->lane[side] = 0; continue; } // (cannot happen at shapes the LDS tiles are chosen for)
    // the compact form of the stream for this side?  A function of the whole problem (view size, losses, options), like the family itself
    const int cmode = env_int("GLRM_HIP_LANE_COMPACT", 2);
    const bool compact = !(rows && h->n_losses > 1) && (cmode == 1 || (cmode == 2 && (rows ? h->sig.nnz_rows : h->sig.nnz_cols) > 2000000000ll)) &&
                         64 * (rows ? h->sig.max_row_len : h->sig.max_col_len) < ((int64_t)1 << 31); // (a wave block's values are counted in 31 bits: LaneArgs::sval)
    int64_t *cnt = nullptr, *scan = nullptr, *vcnt = nullptr;
    void* tmp = nullptr;
    auto cleanup = [&](int rc) { (void)hipFree(cnt); (void)hipFree(scan); (void)hipFree(vcnt); (void)hipFree(tmp); return rc; };
    HIPCK(hipMalloc((void**)&cnt, (size_t)ncell * 8));
    if (hipMalloc((void**)&scan, (size_t)ncell * 8) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
    if (compact && hipMalloc((void**)&vcnt, (size_t)ncell * 8) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
    hipLaunchKernelGGL(lane_count_kernel, dim3((unsigned)nwb), dim3(64), 0, st, ptr, idx, perm, nslots, T, ntiles, cnt, vcnt);
    size_t bytes = 0;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt, scan, (int)ncell, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "scan (size query) failed"));
    if (hipMalloc(&tmp, bytes ? bytes : 1) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
    if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, cnt, scan, (int)ncell, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "scan failed"));
    if (hipMalloc((void**)&h->lane_bptr[side], (size_t)nwb * (ntiles + 1) * 8) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
    const int64_t nb = nwb * (ntiles + 1);
    hipLaunchKernelGGL(lane_bptr_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, scan, cnt, nwb, ntiles, h->lane_bptr[side]);
    int64_t last[2] = {0, 0};
    if (hipMemcpyAsync(&last[0], scan + ncell - 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&last[1], cnt + ncell - 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return cleanup(fail(GLRM_ERR_HIP, "lane layout: copy failed"));
    const int64_t steps = last[0] + last[1];
    h->lane_steps[side] = steps;
    const int64_t s1 = steps > 0 ? steps : 1;
    if (compact) {
      // values before every (wave block, tile): the same scan over the observation counts (the step scan's scratch serves again)
      if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, vcnt, scan, (int)ncell, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "scan failed"));
      if (hipMalloc((void**)&h->lane_vptr[side], (size_t)nwb * (ntiles + 1) * 8) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
      hipLaunchKernelGGL(lane_bptr_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, scan, vcnt, nwb, ntiles, h->lane_vptr[side]);
      const int64_t nv = std::max<int64_t>(1, rows ? h->nnz_r : h->nnz_c);
      const bool ok = hipMalloc((void**)&h->lane_off16[side], (size_t)s1 * 64 * 2) == hipSuccess && hipMalloc((void**)&h->lane_sval[side], (size_t)s1 * 4) == hipSuccess &&
                      hipMalloc((void**)&h->lane_val[side], (size_t)nv * 8) == hipSuccess;
      if (!ok) {
        // No room for the stream beside the lists: the side stays on the four-lane kernels.  (The only family choice that can depend on
        // the device rather than on the problem; glrm_hip_sum_order reports what runs, and the super-tile geometry is already fixed.)
        (void)hipGetLastError();
        (void)hipFree(h->lane_off16[side]); (void)hipFree(h->lane_sval[side]); (void)hipFree(h->lane_val[side]); (void)hipFree(h->lane_vptr[side]); (void)hipFree(h->lane_bptr[side]);
        h->lane_off16[side] = nullptr; h->lane_sval[side] = nullptr; h->lane_val[side] = nullptr; h->lane_vptr[side] = nullptr; h->lane_bptr[side] = nullptr;
        h->lane[side] = 0;
        h->lane_steps[side] = 0;
        fprintf(stderr, "[glrm lane] no device memory for the compact stream of the %s view (%.1f GB): the view stays on the four-lane LDS-tiled kernels\n", rows ? "row" : "column",
                ((double)s1 * 128 + (double)nv * 8) / 1e9);
        cleanup(0);
        continue;
      }
      hipLaunchKernelGGL(lane_fill_compact_kernel, dim3((unsigned)nwb), dim3(64), 0, st, ptr, idx, vals, perm, nslots, T, ntiles, h->lane_bptr[side], h->lane_vptr[side],
                         h->lane_off16[side], h->lane_sval[side], h->lane_val[side]);
    } else {
      if (hipMalloc((void**)&h->lane_off[side], (size_t)s1 * 64 * 4) != hipSuccess || hipMalloc((void**)&h->lane_val[side], (size_t)s1 * 64 * 8) != hipSuccess)
        return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the lane-per-segment stream of the %s view (%lld steps of 64 entries)", rows ? "row" : "column", (long long)steps));
      hipLaunchKernelGGL(lane_fill_kernel, dim3((unsigned)nwb), dim3(64), 0, st, ptr, idx, vals, (rows && h->n_losses > 1) ? h->rowdescid : nullptr, perm, nslots, T, ntiles,
                         h->kp * 8, h->lane_bptr[side], h->lane_off[side], h->lane_val[side]);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "lane layout: build failed"));
    cleanup(0);
    if (perm) { // FORM 2 of the rounds finds a segment's slot through the inverse of the slot permutation
      HIPCK(hipMalloc((void**)&h->lane_inv[side], (size_t)nseg * 4));
      HIPCK(hipMemsetAsync(h->lane_inv[side], 0xFF, (size_t)nseg * 4, st));
      hipLaunchKernelGGL(lane_inv_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, perm, nslots, h->lane_inv[side]);
      HIPCK(hipGetLastError());
    }
    if (nseg > gseg_max) gseg_max = nseg;
    if (env_int("GLRM_HIP_LANE_TRACE", 0))
      fprintf(stderr, "[glrm lane] %s view: %lld slots in %lld wave blocks x %d tiles, %lld steps of 64 = %.3f x the %lld observations%s\n", rows ? "row" : "column",
              (long long)nslots, (long long)nwb, ntiles, (long long)steps, (double)steps * 64.0 / (double)std::max<int64_t>(1, rows ? h->nnz_r : h->nnz_c),
              (long long)(rows ? h->nnz_r : h->nnz_c), compact ? " (compact stream: 2-byte offsets padded, values unpadded)" : "");
  }
  if (gseg_max > 0 && !h->lane_glist) { // the wave lists of the rounds (one set serves both sides: a half-step at a time)
    h->lane_gchunks = (gseg_max + LANE_CC - 1) / LANE_CC;
    h->lane_gcap = h->lane_gchunks * 16; // waves: a chunk holds at most 64 segments of a class
    HIPCK(hipMalloc((void**)&h->lane_gcnt, (size_t)h->lane_gchunks * 4));
    HIPCK(hipMalloc((void**)&h->lane_gbase, (size_t)h->lane_gchunks * 4));
    HIPCK(hipMalloc((void**)&h->lane_gtotal, 4));
    HIPCK(hipMalloc((void**)&h->lane_glist, (size_t)h->lane_gcap * 64 * 4));
  }
  return GLRM_OK;
}

// One half-step (or the evaluation pass of the column view) on the lane-per-segment passes.  `a` comes prepared by glrm_run_tiled: the side's
// views, factors, pass buffers, slot permutation and -- rows of a sub-range sweep -- everything offset to the range's first row.
int glrm_run_lane(glrm_handle* h, bool rows, int loss, const TiledArgs& a_in, double min_stepsize, int eval_only) {
  TiledArgs a = a_in;
  const int side = rows ? 0 : 1;
  hipStream_t st = h->stream;
  LaneArgs la{};
  la.bptr = h->lane_bptr[side];
  la.off = h->lane_off[side];
  la.val = h->lane_val[side];
  la.off16 = h->lane_off16[side];
  la.vptr = h->lane_vptr[side];
  la.sval = h->lane_sval[side];
  la.ntiles = h->lane_ntiles[side];
  la.nwb = h->lane_nwb[side];
  la.slot0 = 0;
  la.inv = h->lane_inv[side];
  la.glist = h->lane_glist;
  la.gwaves = 0;
  (void)min_stepsize;
  // does the SELL layout cover this launch?  It was built on the side's full slot space (slot permutation included); a row sub-range
  // (glrm_hip_step_x_range) is covered when the slots are the rows themselves: its wave blocks run with the lanes outside the range masked
  const bool range = rows && h->rng_e >= 0;
  const bool sell_ok = !range || h->rowperm == nullptr;
  if (range) la.slot0 = h->rng_b;
  const int64_t nslots = a.npass > 0 ? a.npass : a.nseg;
  const int64_t blk_lo = la.slot0 / 64, blk_hi = (la.slot0 + nslots + 63) / 64;
  int rc;
  const bool lists = h->actlist && (rows ? (h->tile_rounds & 1) != 0 : (h->tile_rounds & 2) != 0) && a.nseg <= h->actlist_cap;
  int32_t* list[2] = {lists ? h->actlist : nullptr, lists ? h->actlist + h->actlist_cap : nullptr};
  int cur = 0;
  HIPCK(hipMemsetAsync(h->nactive, 0, 4, st));
  a.actlist_out = list[cur];
  auto csr_args = [&](const TiledArgs& t) { // the CSR form numbers its slots from 0 over t.nseg (a compact list or a plain range)
    LaneArgs c = la;
    c.slot0 = 0;
    c.gwaves = 0;
    return c;
  };
  // gradient pass: under glrm_hip_step_y_arrival (columns) in runs of super-tiles, each behind the blocks of X it reads (announced order)
  auto grad = [&](int s0, int s1) {
    TiledArgs r = a;
    r.sup0 = s0;
    r.nsup_launch = s1 - s0;
    if (sell_ok) return launch_lane_loss<true, 0>(loss, r, la, blk_hi - blk_lo, st);
    r.npass = 0;
    return launch_lane_loss<true, 1>(loss, r, csr_args(r), (r.nseg + 63) / 64, st);
  };
  rc = rows ? grad(0, a.nsup) : glrm_for_sup_runs_in_arrival_order(h, a.nsup, (int64_t)a.tiles_per_sup * lane_tile_rows(h->kp), grad); // (rows read Y: complete)
  if (rc) return rc;
  launch_small(0, a, st);
  HIPCK(hipGetLastError());
  if (eval_only) return GLRM_OK;
  const TiledArgs full = a;
  const bool gather = env_int("GLRM_HIP_LANE_ROUNDS", 1) != 0 && sell_ok && h->lane_glist && la.bptr && h->lane_steps[side] > 0 &&
                      full.nseg <= h->lane_gchunks * (int64_t)LANE_CC;
  constexpr int MAX_ROUNDS = 4096; // see glrm_run_tiled
  for (int round = 0;; ++round) {
    if (round == MAX_ROUNDS) return fail(GLRM_ERR_INVALID, "line search still running after %d rounds (min_stepsize %g)", MAX_ROUNDS, min_stepsize);
    unsigned int nact = 0;
    TiledArgs t = full;
    TiledArgs d = full;
    HIPCK(hipMemcpyAsync(&nact, h->nactive, 4, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if (nact == 0) break;
    HIPCK(hipMemsetAsync(h->nactive, 0, 4, st));
    const int trace = env_int("GLRM_HIP_LANE_TRACE", 0);
    // Three forms of the trial pass, all adding the same terms in the same order (tests/test_gpu_families.py forces each of them on every
    // round).  FORM 0 walks the SELL layout over the full grid (idle segments masked): the cost of a whole pass whatever the fraction that
    // searches -- from 70 % up (sweep at C5's stated size, session r6_50: 55 / 70 / 85 % = 344.5 / 341.6 / 340.5 ms: flat).  FORM 2 reads the layout through gathered waves (glrm_lane.hpp): waves built chunk by chunk of 1 024
    // segments (or SLOTS, where the layout's slots are permuted: they were dealt out class by class) so that a wave's lanes share lines,
    // packed from the decide kernel's compact list and spread over the chip for the tail rounds.  On the compact stream a lane finds its
    // value through the rank in its offset word and the step's base.  The CSR form (one lane walks its own list: the address translation
    // of 512 lists far apart, 3-5 ms for any tail round) is what remains for sub-ranges the layout does not
    // cover.  C5 recipe, 1M rows, X half-step (profiles/r06_c5family_lane_rounds_trace.txt): full grid 9.85 ms; gathered 7.8 ms at 55 %,
    // 5.4 at 32 %, 3.1 at 12 %, 1.5-2.1 for the tails.
    const int64_t pct = (int64_t)nact * 100 / (full.nseg > 0 ? full.nseg : 1);
    // Tail rounds: a wave per (segment, super-tile) instead of a lane's serial walk through every tile (rows: 1.5-9 ms per round at C5's stated
    // size, eight to ten of them per X half-step) -- below GLRM_HIP_LANE_TAIL / GLRM_HIP_LANE_TAIL_COLS percent of the segments
    if (lists && (int64_t)nact * 100 < full.nseg * (int64_t)(full.nsup == 1 ? env_int("GLRM_HIP_LANE_TAIL", 5) : env_int("GLRM_HIP_LANE_TAIL_COLS", 3))) {
      if (trace >= 2) fprintf(stderr, "[glrm lane] %s round %d: %u of %lld segments search: a wave per segment\n", rows ? "row" : "column", round, nact, (long long)full.nseg);
      if ((rc = launch_lane_tail(loss, full, list[cur], (int64_t)nact, st))) return rc;
      d.actlist_in = list[cur];
      d.actlist_out = list[cur ^ 1];
      d.nact_in = nact;
      cur ^= 1;
      launch_small(1, d, st);
      HIPCK(hipGetLastError());
      continue;
    }
    bool packed = lists && (int64_t)nact * 100 < full.nseg * env_int("GLRM_HIP_LANE_GATHER_PACKED", 4);
    // (sides with permuted slots: chunk lists over the SLOTS, which were dealt out class by class -- make_segperm)
    const int32_t* gperm = la.inv ? full.segperm : nullptr;
    const bool chunks_ok = !la.inv || (h->lane_dealt[side] && gperm && env_int("GLRM_HIP_LANE_GATHER_SLOTS", 1));
    bool use_gather = gather && (chunks_ok || packed) && pct >= env_int("GLRM_HIP_LANE_GATHER_FROM", 0) && pct < env_int("GLRM_HIP_LANE_GATHER_TO", 70);
    int32_t gwaves = 0;
    if (use_gather) {
      const int off16 = (int)(full.own_offset & 15);
      if (packed) {
        const int spread = env_int("GLRM_HIP_LANE_GATHER_SPREAD", 32768); // searching segments per step of q: ~2 048 waves before a wave takes more per class
        int q = (int)(((int64_t)nact + spread - 1) / spread);
        q = q < 1 ? 1 : (q > 4 ? 4 : q);
        hipLaunchKernelGGL(lane_compact_list_kernel, dim3(1), dim3(1024), 0, st, list[cur], (int)nact, off16, q, (int)h->lane_gcap, h->lane_glist, h->lane_gtotal);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(&gwaves, h->lane_gtotal, 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        if (gwaves < 0) { // (one class holds nearly every entry: the waves would not fit the list)
          packed = false;
          if (!chunks_ok) use_gather = false;
        }
      }
      if (use_gather && !packed) {
        const int64_t nitems = gperm ? (full.npass > 0 ? full.npass : full.nseg) : full.nseg;
        const int64_t nchunks = (nitems + LANE_CC - 1) / LANE_CC;
        const unsigned cg = (unsigned)((nchunks + 3) / 4);
        hipLaunchKernelGGL(lane_compact_count_kernel, dim3(cg), dim3(256), 0, st, full.active, nitems, nchunks, gperm, h->lane_gcnt);
        hipLaunchKernelGGL(lane_compact_scan_kernel, dim3(1), dim3(1024), 0, st, h->lane_gcnt, nchunks, h->lane_gbase, h->lane_gtotal);
        hipLaunchKernelGGL(lane_compact_fill_kernel, dim3(cg), dim3(256), 0, st, full.active, nitems, nchunks, gperm ? 0 : off16, gperm, h->lane_gbase, h->lane_glist);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(&gwaves, h->lane_gtotal, 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
      }
    }
    if (use_gather) {
      la.gwaves = gwaves;
      if (trace >= 2) fprintf(stderr, "[glrm lane] %s round %d: %u of %lld segments search: SELL gathered%s, %d waves\n", rows ? "row" : "column", round, nact, (long long)full.nseg, packed ? " (packed)" : "", (int)gwaves);
      t.npass = 0;
      rc = launch_lane_loss<false, 2>(loss, t, la, (int64_t)gwaves, st);
      if (rc) return rc;
      if (lists) {
        d.actlist_in = list[cur];
        d.actlist_out = list[cur ^ 1];
        d.nact_in = nact;
        cur ^= 1;
      }
      launch_small(1, d, st);
      HIPCK(hipGetLastError());
      continue;
    }
    // (measured cross-over of the two older forms at about a sixth of the segments, session r6_20)
    const bool compact = lists && ((int64_t)nact * 100 < full.nseg * env_int("GLRM_HIP_LANE_CSR_BELOW", 16) || !sell_ok);
    if (trace >= 2) fprintf(stderr, "[glrm lane] %s round %d: %u of %lld segments search: %s\n", rows ? "row" : "column", round, nact, (long long)full.nseg, compact ? "CSR" : "SELL full grid");
    if (compact) {
      // (tried in session r6_45/46 and taken out again: this form over class-aware wave lists -- conflict-free tile reads -- measured the same at
      // full waves and worse spread over the chip; nor did batching its list reads move it.  A CSR round costs 2.8 x the full grid's time
      // per workgroup and tile at C5's stated size: 512 lanes walk 512 lists 1.2 MB apart, which is an address-translation pattern, not a
      // bandwidth or bank pattern.  Cross-over with the full grid stays at a sixth of the segments.)
      t.segperm = list[cur];
      t.nseg = nact;
      t.npass = 0;
      rc = launch_lane_loss<false, 1>(loss, t, csr_args(t), ((int64_t)nact + 63) / 64, st);
    } else if (sell_ok) {
      rc = launch_lane_loss<false, 0>(loss, t, la, blk_hi - blk_lo, st);
    } else {
      t.npass = 0;
      rc = launch_lane_loss<false, 1>(loss, t, csr_args(t), (t.nseg + 63) / 64, st);
    }
    if (rc) return rc;
    if (lists) {
      d.actlist_in = list[cur];
      d.actlist_out = list[cur ^ 1];
      d.nact_in = nact;
      cur ^= 1;
    }
    launch_small(1, d, st);
    HIPCK(hipGetLastError());
  }
  return GLRM_OK;
}
