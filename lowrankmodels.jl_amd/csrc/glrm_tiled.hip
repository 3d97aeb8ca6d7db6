// glrm_tiled.hip -- host side + instantiations of the LDS-tiled sweeps (kernels in glrm_tiled.hpp).
// Separate translation unit so that the two kernel families compile in parallel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <vector>

#include "glrm_engine.hpp"
#include "glrm_tiled.hpp"

using namespace glrm;

// ------------------------------------------------------------------ LDS-tiled sweeps: setup and launch

// opposing vectors per LDS tile (rows are padded by 16 B): cfg 0 = 64 KB tile, 8 waves, two workgroups per CU;
// cfg 1 = 150 KB tile, 16 waves, one workgroup per CU (fewer barriers, fewer factor re-reads)
constexpr int tile_rows_c(int kp, int cfg) { return ((cfg ? 150 * 1024 : 64 * 1024) / (kp * 8 + 16)) / 16 * 16; }
static int tile_rows(int kp, int cfg) { return tile_rows_c(kp, cfg); }
static bool tile_rot_rt(int G, int R) { return GLRM_TILE_ROT && (G == 4 || G == 8) && R == 8; }

// slot -> segment permutation of a tiled sweep: segments sorted by (loss kind of the column,) descending length, so that the 16
// lane groups of a wave meet one loss formula and lists of similar length.  nullptr when the natural order is already that
// (one loss kind and lengths within 25 % of each other: the synthetic BASELINE workloads).
// long_from > 0 (columns): segments of at least that many observations are left out of the slots (they run on the 8-wave gather sweep
// beside the passes, glrm_hip.hip: run_sweep) and listed in *longs; *nshort = slots handed out
static int make_segperm(glrm_handle* h, bool rows, int32_t** out, int64_t long_from = 0, std::vector<int32_t>* longs = nullptr, int64_t* nshort = nullptr) {
  *out = nullptr;
  const int64_t nseg = rows ? h->ml : h->nl;
  if (nshort) *nshort = nseg;
  if (nseg <= 1 || !env_int("GLRM_HIP_SEGPERM", 1)) return GLRM_OK;
  std::vector<int64_t> ptr((size_t)nseg + 1);
  HIPCK(hipMemcpyAsync(ptr.data(), rows ? h->rowptr : h->colptr, ((size_t)nseg + 1) * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  int64_t lmin = INT64_MAX, lmax = 0;
  for (int64_t s = 0; s < nseg; ++s) { const int64_t l = ptr[s + 1] - ptr[s]; lmin = l < lmin ? l : lmin; lmax = l > lmax ? l : lmax; }
  const bool kinds = !rows && h->n_losses > 1;
  const bool divert = long_from > 0 && longs && lmax >= long_from;
  if (!kinds && !divert && lmax * 4 <= lmin * 5) return GLRM_OK;
  std::vector<int32_t> perm;
  perm.reserve((size_t)nseg);
  for (int64_t s = 0; s < nseg; ++s) {
    if (divert && ptr[s + 1] - ptr[s] >= long_from) longs->push_back((int32_t)s);
    else perm.push_back((int32_t)s);
  }
  const int64_t nslots = (int64_t)perm.size();
  if (nshort) *nshort = nslots;
  const glrm_loss* lt = kinds ? h->losses_h.data() + h->cb : nullptr;
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) {
    if (kinds && lt[x].kind != lt[y].kind) return lt[x].kind < lt[y].kind;
    return ptr[x + 1] - ptr[x] > ptr[y + 1] - ptr[y];
  });
  const bool lane_side = (rows ? h->tiled_row : h->tiled_col) && glrm_lane_wants(h, rows) && env_int("GLRM_HIP_LANE_DEAL", 1);
  if (lane_side) {
    // The lane-per-segment passes (glrm_lane.hpp) read their tiles conflict free when the 16 lanes of an LDS cycle hold 16 different
    // classes (global id) & 15 -- true when slot s holds a segment of class s & 15.  In natural order that is given; a sorted list is dealt
    // out class by class (the 16 classes are equally frequent, so a wave still meets segments of one kind and similar length; a class
    // that runs dry is filled from the longest remaining queue: costs conflicts at the tail, never bits).  Measured before (session r6_35,
    // C5 recipe at 1M rows, columns sorted by kind and length): bank conflicts in 48 % of the column passes' LDS cycles.
    std::vector<int32_t> q[16];
    const int64_t g0 = rows ? h->rb : h->cb;
    h->lane_dealt[rows ? 0 : 1] = 1;
    for (int32_t sgm : perm) q[(g0 + sgm) & 15].push_back(sgm);
    size_t head[16] = {0};
    for (int64_t slot = 0; slot < nslots; ++slot) {
      int c = (int)(slot & 15);
      if (head[c] >= q[c].size()) {
        size_t best = 0;
        for (int d = 0; d < 16; ++d)
          if (q[d].size() - head[d] > best) { best = q[d].size() - head[d]; c = d; }
      }
      perm[(size_t)slot] = q[c][head[c]++];
    }
  } else if (!rows && tile_rot_rt(h->G, h->R)) {
    // the column passes read their tiles conflict-free (glrm_tiled.hpp: tile_rot) when slot s holds a segment of class (s & 7) >> 1,
    // class = ((global id) & 7) >> 1: deal the sorted list out class by class, two per block of eight slots -- the four classes are
    // equally frequent, so every wave still meets segments of one kind and similar length; a class that runs dry is filled from the
    // longest remaining queue (costs bank conflicts at the tail, never bits)
    std::vector<int32_t> q[4];
    for (int32_t sgm : perm) q[tile_rot_of(h->cb + sgm)].push_back(sgm);
    size_t head[4] = {0, 0, 0, 0};
    for (int64_t slot = 0; slot < nslots; ++slot) {
      int c = tile_rot_of(slot); // the class the slot's position wants
      if (head[c] >= q[c].size()) {
        size_t best = 0;
        for (int d = 0; d < 4; ++d)
          if (q[d].size() - head[d] > best) { best = q[d].size() - head[d]; c = d; }
      }
      perm[(size_t)slot] = q[c][head[c]++];
    }
  }
  HIPCK(hipMalloc((void**)out, (size_t)(nslots > 0 ? nslots : 1) * 4));
  HIPCK(hipMemcpyAsync(*out, perm.data(), (size_t)nslots * 4, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream)); // perm is a local
  return GLRM_OK;
}

// create phase: the tile configuration and whether this shard's lists are in tile order (glrm_signature::rows_unordered / cols_unordered)
int glrm_prepare_tiled(glrm_handle* h) {
  hipStream_t st = h->stream;
  h->tile_cfg = env_int("GLRM_HIP_TILE_CFG", 1);
  h->tile_cfg12 = h->tile_cfg == 2; // experiment: 12-wave heterogeneous row sweep
  // Loader waves of the double-buffered tiled sweeps (0 = single tile staged by every wave; glrm_tiled.hpp).  Round 2 ran the ROW sweep of
  // uniform QuadLoss models on two loader waves (C2 row sweep 8.31 -> 7.41 ms against load / ds_write staging).  Since round 3 the single
  // tile is staged by LDS-DMA from all waves (dma_tile_all) and wins on the four shapes that default was chosen on (tools/gpu_r3_30.sh,
  // profiles/r03_tile_dma_all_ab.txt: C2 row sweep 7.50 -> 7.12 ms, 1M x 50k 23.0 -> 20.4, 1M x 2k 1.51 -> 1.48, 300k x 3k 0.73 -> 0.69, same
  // objective bits): the default is 0 everywhere; GLRM_HIP_TILE_LW = 1 | 2 still selects the loader waves.
  h->tile_lw = env_int("GLRM_HIP_TILE_LW", 0);
  if (h->tile_lw < 0 || h->tile_lw > 2 || !((h->G == 4 || h->G == 8) && h->R == 8) || !h->tile_cfg) h->tile_lw = 0;
  h->tile_lw_sides = env_int("GLRM_HIP_TILE_LW_SIDES", 1); // bit0: row sweep, bit1: column passes
  h->tile_cfg = h->tile_cfg ? 1 : 0;
  h->tG = h->G;
  h->tR = h->R;
  // order granularity of the index lists: the entries of staged tile t must precede those of tile t+1; with loader waves the
  // staged unit is HALF a tile (a list ordered by half tile is ordered by tile as well)
  const int T0 = h->tile_lw > 0 ? tile_rows(h->kp, h->tile_cfg) / 2 : tile_rows(h->kp, h->tile_cfg);
  h->order_unit = T0;
  HIPCK(hipMalloc((void**)&h->dflag, 2 * sizeof(int)));
  HIPCK(hipMemsetAsync(h->dflag, 0, 2 * sizeof(int), st));
  if (h->ml > 0) hipLaunchKernelGGL(check_sorted_kernel, dim3((unsigned)h->ml), dim3(64), 0, st, h->rowptr, h->colidx, h->ml, T0, h->dflag);
  if (h->nl > 0) hipLaunchKernelGGL(check_sorted_kernel, dim3((unsigned)h->nl), dim3(256), 0, st, h->colptr, h->rowidx, h->nl, T0, h->dflag + 1);
  HIPCK(hipGetLastError());
  int flags[2] = {0, 0};
  HIPCK(hipMemcpyAsync(flags, h->dflag, sizeof flags, hipMemcpyDeviceToHost, st));
  HIPCK(hipStreamSynchronize(st));
  h->sig_local.rows_unordered = flags[0] != 0;
  h->sig_local.cols_unordered = flags[1] != 0;
  return GLRM_OK;
}

// finalize phase.  Everything that selects a family is read from h->sig -- the signature of the WHOLE problem -- and from (m, n, k,
// options): every shard of a sharded fit lands on the same family.
int glrm_setup_tiled(glrm_handle* h) {
  hipStream_t st = h->stream;
  int rc0 = GLRM_OK;
  const int T0 = h->order_unit;
  h->rows_sorted = !h->sig.rows_unordered;
  h->cols_sorted = !h->sig.cols_unordered;

  const int T = tile_rows(h->kp, h->tile_cfg);
  // expected observations of one segment inside one tile; the tiled sweeps pay off when a staged
  // vector is reused by several of the workgroup's segments
  const double per_tile_r = (double)h->sig.nnz_rows / (double)h->m * T / (double)h->n;
  const double per_tile_c = (double)h->sig.nnz_cols / (double)h->n * T / (double)h->m;
  // h->tiled_opt (glrm_options.tiled): 0 auto, 1 gather sweeps only, 2 tiled wherever the index lists are sorted.
  // GLRM_HIP_TILED (tuning): bit0 rows, bit1 columns; overrides the option.
  int want = h->tiled_opt == 1 ? 0 : (h->tiled_opt == 2 ? 3 : -1);
  want = env_int("GLRM_HIP_TILED", want);
  // Auto choice (measured on MI355X, tests/perf/bench_small.py): the tiled sweeps need enough workgroups to fill 256 CUs and
  // enough observations to amortise their per-tile barriers; below ~2e7 observations per view the gather sweeps win (100k x 5k
  // at 1e7 observations: 1.06 vs 1.23 ms per iteration; 300k x 3k at 4.5e7: 4.6 vs 3.1 ms).
  const int spb_auto = ((h->tile_cfg ? 16 : 8) - h->tile_lw) * (64 / h->tG);
  const bool big_r = h->sig.nnz_rows >= 20000000 && h->m >= (int64_t)512 * spb_auto;
  const bool big_c = h->sig.nnz_cols >= 20000000 && h->n >= 256;
  bool want_row = want < 0 ? (per_tile_r >= 4.0 && big_r) : (want & 1) != 0;
  bool want_col = want < 0 ? (per_tile_c >= 4.0 && big_c) : ((want >> 1) & 1) != 0;
  // Lists in arbitrary order (obs tuples pushed in sampling order): the engine's private copy is brought into tile order by a
  // stable segmented sort, so such inputs run the tiled sweeps too.  Only when the auto choice wants the tiled sweeps -- an
  // explicit tiled = 2 keeps its meaning "wherever the lists allow" (GLRM_HIP_TILE_SORT=0 disables, =2 sorts for tiled = 2 as well).
  const int tsort = env_int("GLRM_HIP_TILE_SORT", 1);
  const bool may_sort = tsort == 2 || (tsort == 1 && want < 0);
  // (a shard whose own lists are already in tile order gets them back unchanged -- the sort is stable -- so the decision may be
  // taken for the whole problem; a list too long for the segmented sort anywhere keeps every shard on the gather sweeps)
  const int64_t sort_limit = env_int("GLRM_HIP_TILE_SORT_BATCH", 0) > 0 ? env_int("GLRM_HIP_TILE_SORT_BATCH", 0) : 1500000000ll;
  if (want_row && !h->rows_sorted && may_sort && h->sig.max_row_len <= sort_limit) {
    if (h->sig_local.rows_unordered) {
      const int rc = glrm_tile_sort_view(st, h->rowptr, h->ml, h->nnz_r, T0, h->n, &h->colidx, &h->rowvals, h->own_rowview);
      if (rc) return rc;
      h->own_rowview = true;
    }
    h->rows_sorted = true;
  }
  if (want_col && !h->cols_sorted && may_sort && h->sig.max_col_len <= sort_limit) {
    if (h->sig_local.cols_unordered) {
      const int rc = glrm_tile_sort_view(st, h->colptr, h->nl, h->nnz_c, T0, h->m, &h->rowidx, &h->colvals, h->own_colview);
      if (rc) return rc;
      h->own_colview = true;
    }
    h->cols_sorted = true;
  }
  h->tiled_row = (h->rows_sorted && want_row) ? 1 : 0;
  h->tiled_col = (h->cols_sorted && want_col) ? 1 : 0;
  // (rows on the lane-per-segment passes -- glrm_lane.hpp -- keep the caller's order: every lane evaluates its own observation's loss, so
  // there is no wave-wide formula to align, and the grouped copy would cost 12 B per observation beside the SELL stream)
  const bool lane_rows = h->tiled_row && !env_int("GLRM_HIP_ROW_SPLIT", 0) && (env_int("GLRM_HIP_TILE_ROUNDS", 3) & 1) && !(h->tile_cfg12) && glrm_lane_wants(h, true);
  if (h->tiled_row && h->n_losses > 1 && h->nnz_r > 0 && env_int("GLRM_HIP_GROUP_KINDS", 1) && !lane_rows) {
    int32_t* oidx = nullptr;
    double* ovals = nullptr;
    HIPCK(hipMalloc((void**)&oidx, (size_t)h->nnz_r * 4));
    if (hipMalloc((void**)&ovals, (size_t)h->nnz_r * 8) != hipSuccess) { (void)hipFree(oidx); return fail(GLRM_ERR_OOM, "out of device memory"); }
    hipLaunchKernelGGL(group_rows_by_kind_kernel, dim3((unsigned)h->ml), dim3(64), 0, st, h->rowptr, h->colidx, h->rowvals, h->ml, T0, h->losses, oidx, ovals);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(st));
    if (h->own_rowview) {
      (void)hipFree(h->colidx);
      (void)hipFree(h->rowvals);
    }
    h->colidx = oidx;
    h->rowvals = ovals;
    h->own_rowview = true;
  }
  if (h->tiled_row && h->n_losses > 1 && h->nnz_r > 0 && env_int("GLRM_HIP_UDESC", 1)) {
    // distinct loss descriptors of the model; with at most 256 of them every entry of the row view carries a one-byte id and the
    // row sweep reads descriptors from an LDS table (TiledArgs::descid)
    std::vector<glrm_loss> uniq;
    std::vector<uint8_t> colid((size_t)h->n);
    bool ok = true;
    for (int64_t f = 0; f < h->n && ok; ++f) {
      const glrm_loss& l = h->losses_h[(size_t)f];
      size_t u = 0;
      for (; u < uniq.size(); ++u)
        if (uniq[u].kind == l.kind && uniq[u].dim == l.dim && uniq[u].scale == l.scale && uniq[u].p0 == l.p0 && uniq[u].p1 == l.p1) break;
      if (u == uniq.size()) {
        if (uniq.size() == 256) { ok = false; break; }
        uniq.push_back(l);
      }
      colid[(size_t)f] = (uint8_t)u;
    }
    if (ok) {
      uint8_t* dcolid = nullptr;
      HIPCK(hipMalloc((void**)&dcolid, (size_t)h->n));
      HIPCK(hipMalloc((void**)&h->udesc, uniq.size() * sizeof(glrm_loss)));
      HIPCK(hipMalloc((void**)&h->rowdescid, (size_t)h->nnz_r));
      HIPCK(hipMemcpyAsync(dcolid, colid.data(), (size_t)h->n, hipMemcpyHostToDevice, st));
      HIPCK(hipMemcpyAsync(h->udesc, uniq.data(), uniq.size() * sizeof(glrm_loss), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(entry_descid_kernel, dim3(4096), dim3(256), 0, st, h->colidx, h->nnz_r, dcolid, h->rowdescid);
      HIPCK(hipGetLastError());
      HIPCK(hipStreamSynchronize(st)); // colid / uniq are locals
      (void)hipFree(dcolid);
      h->n_udesc = (int)uniq.size();
    }
  }
  // super-tiles of ~32k rows: a function of (m, tile) only -- never of the shard layout -- so the partial-sum order
  // (and the result bits) do not depend on the GPU count, while long columns still spread over enough workgroups
  const int64_t ntiles = (h->m + T - 1) / T;
  {
    // enough (column group, super-tile) workgroups to fill the chip a few times over: the column groups come from the GLOBAL n
    // (256 columns per 16-wave workgroup at G=4), so the super-tile size -- hence the order of the partial sums -- is a function
    // of (m, n, tile) only, never of the shard layout
    // (the lane-per-segment form of the passes holds 512 columns per workgroup and stages its tiles of X at ~3 TB/s with nothing to overlap
    // them: more, shorter super-tiles even the workgroups out -- C2 Y half-step 7.6 -> 7.0 ms at twice the workgroups, session r6_09; the
    // family is a function of the whole problem's signature, so this still is)
    const bool lane_c = h->tiled_col && glrm_lane_wants(h, false);
    const int spb = lane_c ? 512 : (h->tile_cfg ? 16 : 8) * (64 / h->tG);
    const int64_t groups = (h->n + spb - 1) / spb;
    const int64_t want_wg = env_int("GLRM_HIP_COL_WORKGROUPS", lane_c ? 2048 : 1024);
    int64_t nsup_target = (want_wg + groups - 1) / groups;
    if (nsup_target < 1) nsup_target = 1;
    int64_t tps = ntiles / nsup_target;
    const int64_t tps_max = 32768 / T > 1 ? 32768 / T : 1; // at most ~32k rows per super-tile (long columns still spread out)
    if (tps > tps_max) tps = tps_max;
    if (tps < 1) tps = 1;
    h->tiles_per_sup = (int)tps;
  }
  h->nsup = (int)((ntiles + h->tiles_per_sup - 1) / h->tiles_per_sup);
  if (h->tiled_col) {
    const int64_t nl1 = h->nl > 0 ? h->nl : 1;
    HIPCK(hipMalloc((void**)&h->part, (size_t)nl1 * h->nsup * (h->kp + 2) * 8));
    HIPCK(hipMalloc((void**)&h->gsum, (size_t)nl1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->trialbuf, (size_t)nl1 * h->kp * 8));
    HIPCK(hipMalloc((void**)&h->joldbuf, (size_t)nl1 * 8));
    HIPCK(hipMalloc((void**)&h->activebuf, (size_t)nl1 * 4));
    HIPCK(hipMalloc((void**)&h->ntrialbuf, (size_t)nl1 * 4));
    HIPCK(hipMalloc((void**)&h->nactive, 4));
    HIPCK(hipMemsetAsync(h->activebuf, 0, (size_t)nl1 * 4, st)); // diverted columns are never touched by col_reduce: "not searching"
    // Skewed column lengths (round 5, like the phase-aligned passes: glrm_blocked.hip).  The columns are already handed out sorted by
    // length; a workgroup's 256 columns walk a tile in lockstep (one barrier per tile), so ONE column many times the others keeps its
    // workgroup on every tile for its own entries alone, and the few workgroups of the head of the sorted list are the makespan (C2
    // recipe with Zipf(0.5) degrees: Y half-step 22.1 ms against 9.0).  Columns of at least long_from = max(4 096, 4 x the whole problem's
    // mean column length) observations leave the passes for the 8-wave gather sweep on the side stream: a function of the column's own
    // length and the whole problem's signature (shard-invariant), reported in glrm_sum_order.long_from.  Measured at C2-Zipf (mean 50 266,
    // longest 879 189): Y half-step 27.5 / 19.0 / 16.2 / 17.1 ms at 1 / 2 / 4 / 8 x mean (profiles/r05_c2_zipf_long_from_sweep.txt): the
    // gather sweep pays 536 B per update where the tiles pay a fraction, so only the head of the distribution is worth diverting.
    const int64_t mean_len = h->sig.nnz_cols / (h->n > 0 ? h->n : 1);
    h->blk_long_from = env_int("GLRM_HIP_TILED_LONG_FROM", -1) >= 0 ? env_int("GLRM_HIP_TILED_LONG_FROM", 0) : std::max<int64_t>(4096, 4 * mean_len);
    std::vector<int32_t> longl;
    if ((rc0 = make_segperm(h, false, &h->colperm, h->blk_long_from, &longl, &h->blk_nshort_c))) return rc0;
    h->blk_nlong_c = (int64_t)longl.size();
    if (!longl.empty()) {
      HIPCK(hipMalloc((void**)&h->blk_long_c, longl.size() * 4));
      HIPCK(hipMemcpyAsync(h->blk_long_c, longl.data(), longl.size() * 4, hipMemcpyHostToDevice, st));
      if (!h->side_stream) {
        HIPCK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
      }
      HIPCK(hipStreamSynchronize(st)); // longl is a local
    }
  }
  if (h->tiled_row && (rc0 = make_segperm(h, true, &h->rowperm))) return rc0;
  // Line-search rounds over the still-searching segments only (glrm_tiled.hpp: TiledArgs::actlist_out).  Columns: the passes after the
  // first trial; rows: everything after the first trial, which stays fused with the gradient pass in one kernel (GLRM_HIP_TILE_ROUNDS:
  // bit0 rows, bit1 columns, default 3; 0 = the round-3 forms).  Two lists (the decide kernel reads one and writes the next).
  h->tile_rounds = env_int("GLRM_HIP_TILE_ROUNDS", 3);
  if (h->tile_cfg12) h->tile_rounds &= ~1; // the 12-wave experiment exists for the one-kernel row sweep only
  if ((h->tiled_row && (h->tile_rounds & 1)) || (h->tiled_col && (h->tile_rounds & 2))) {
    const int64_t cap = std::max<int64_t>(1, std::max(h->tiled_row ? h->ml : 0, h->tiled_col ? h->nl : 0));
    HIPCK(hipMalloc((void**)&h->actlist, (size_t)cap * 2 * sizeof(int32_t)));
    h->actlist_cap = cap;
  }
  // Row sweep in super-tile passes.  The one-kernel row sweep streams the WHOLE opposing factor through LDS per workgroup and pass;
  // workgroups that started at different times are at different tiles, so once Y no longer fits the 4 MB L2 of an XCD every tile
  // they stage comes from the Infinity Cache (~8 TB/s for the whole chip, profiles/r02_ubench_gather.txt) -- at 1M x 50k, k = 32
  // that is 100 GB per half-step and 2/3 of its time.  In pass form (the column machinery: pass over one super-tile per workgroup,
  // partials, reduce, trial passes, decide) the grid is ordered super-tile major, so the workgroups in flight all stage tiles of
  // the same L2-sized super-tile.  Super-tile = ~2.5 MB of Y, a function of (n, tile, kp) only.
  h->row_split = 0;
  if (h->tiled_row) {
    const int want_split = env_int("GLRM_HIP_ROW_SPLIT", 0); // measured: no gain (the staged tiles already hit L2 at 88 %); kept as an experiment switch
    if (!want_split && (h->tile_rounds & 1)) { // the pass buffers of the row rounds: ONE super-tile (nothing is re-added: the bits of the one-kernel sweep)
      const int64_t nt = (h->n + T - 1) / T;
      h->tiles_per_sup_r = (int)(nt > 0 ? nt : 1);
      h->nsup_r = 1;
      const int64_t ml1 = h->ml > 0 ? h->ml : 1;
      HIPCK(hipMalloc((void**)&h->part_r, (size_t)ml1 * (h->kp + 2) * 8));
      HIPCK(hipMalloc((void**)&h->gsum_r, (size_t)ml1 * h->kp * 8));
      HIPCK(hipMalloc((void**)&h->trial_r, (size_t)ml1 * h->kp * 8));
      HIPCK(hipMalloc((void**)&h->jold_r, (size_t)ml1 * 8));
      HIPCK(hipMalloc((void**)&h->active_r, (size_t)ml1 * 4));
      HIPCK(hipMalloc((void**)&h->ntrial_r, (size_t)ml1 * 4));
      if (!h->nactive) HIPCK(hipMalloc((void**)&h->nactive, 4));
    }
    if (want_split) {
      h->tile_rounds &= ~1;
      const int64_t nt = (h->n + T - 1) / T;
      int64_t tps = ((int64_t)5 * 512 * 1024) / ((int64_t)T * h->kp * 8);
      tps = env_int("GLRM_HIP_ROW_TPS", (int)(tps < 1 ? 1 : tps));
      h->tiles_per_sup_r = (int)tps;
      h->nsup_r = (int)((nt + tps - 1) / tps);
      const int64_t ml1 = h->ml > 0 ? h->ml : 1;
      HIPCK(hipMalloc((void**)&h->part_r, (size_t)ml1 * h->nsup_r * (h->kp + 2) * 8));
      HIPCK(hipMalloc((void**)&h->gsum_r, (size_t)ml1 * h->kp * 8));
      HIPCK(hipMalloc((void**)&h->trial_r, (size_t)ml1 * h->kp * 8));
      HIPCK(hipMalloc((void**)&h->jold_r, (size_t)ml1 * 8));
      HIPCK(hipMalloc((void**)&h->active_r, (size_t)ml1 * 4));
      HIPCK(hipMalloc((void**)&h->ntrial_r, (size_t)ml1 * 4));
      if (!h->nactive) HIPCK(hipMalloc((void**)&h->nactive, 4));
      h->row_split = 1;
    }
  }
  return glrm_setup_lane(h); // the lane-per-segment form of the passes where it applies (glrm_lane.hip)
}

template <typename K>
static int set_lds(K kernel, int bytes) {
  if (bytes > 65536) HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return GLRM_OK;
}

// kind: 0 = whole sweep (tiled_sweep_kernel), 1 = column pass 1, 2 = column trial pass, 3 = row sweep in rounds form (gradient pass +
// first trial, the rest handed to the rounds), 4 = trial pass of the row rounds.  LW > 0: double-buffered half tiles, the first LW waves
// of the workgroup are LDS-DMA loaders (glrm_tiled.hpp)
template <int G, int R, int NW, int TILE, int LOSS, int LW = 0>
static int launch_tiled_inst(int kind, const TiledArgs& a, hipStream_t st) {
  constexpr int SPB = (NW - LW) * (64 / G);
  const int lds = tile_lds_bytes<G, R, TILE, LW>() + (loss_mode(LOSS) == 2 && a.descid ? a.n_udesc * 32 : 0);
  const unsigned gx = (unsigned)(((a.npass > 0 ? a.npass : a.nseg) + SPB - 1) / SPB);
  int rc = GLRM_OK;
  if (kind == 0 && a.fixed_alpha > 0.0) {
    if ((rc = set_lds(tiled_sweep_kernel<G, R, NW, TILE, LOSS, true, LW>, lds))) return rc;
    hipLaunchKernelGGL((tiled_sweep_kernel<G, R, NW, TILE, LOSS, true, LW>), dim3(gx), dim3(NW * 64), lds, st, a);
  } else if (kind == 0) {
    if ((rc = set_lds(tiled_sweep_kernel<G, R, NW, TILE, LOSS, false, LW>, lds))) return rc;
    hipLaunchKernelGGL((tiled_sweep_kernel<G, R, NW, TILE, LOSS, false, LW>), dim3(gx), dim3(NW * 64), lds, st, a);
  } else if (kind == 3) {
    if ((rc = set_lds(tiled_sweep_kernel<G, R, NW, TILE, LOSS, false, LW, true>, lds))) return rc;
    hipLaunchKernelGGL((tiled_sweep_kernel<G, R, NW, TILE, LOSS, false, LW, true>), dim3(gx), dim3(NW * 64), lds, st, a);
  } else if (kind == 4) {
    if ((rc = set_lds(tiled_col_pass_kernel<G, R, NW, TILE, LOSS, false, false, LW, true>, lds))) return rc;
    hipLaunchKernelGGL((tiled_col_pass_kernel<G, R, NW, TILE, LOSS, false, false, LW, true>), dim3(gx, (unsigned)(a.nsup_launch > 0 ? a.nsup_launch : a.nsup)), dim3(NW * 64), lds, st, a);
  } else if (kind == 1) {
    if ((rc = set_lds(tiled_col_pass_kernel<G, R, NW, TILE, LOSS, true, false, LW>, lds))) return rc;
    hipLaunchKernelGGL((tiled_col_pass_kernel<G, R, NW, TILE, LOSS, true, false, LW>), dim3(gx, (unsigned)(a.nsup_launch > 0 ? a.nsup_launch : a.nsup)), dim3(NW * 64), lds, st, a);
  } else {
    if ((rc = set_lds(tiled_col_pass_kernel<G, R, NW, TILE, LOSS, false, false, LW>, lds))) return rc;
    hipLaunchKernelGGL((tiled_col_pass_kernel<G, R, NW, TILE, LOSS, false, false, LW>), dim3(gx, (unsigned)(a.nsup_launch > 0 ? a.nsup_launch : a.nsup)), dim3(NW * 64), lds, st, a);
  }
  return GLRM_OK;
}

template <int G, int R>
static int launch_tiled_layout(int cfg, int loss, int kind, const TiledArgs& a, hipStream_t st) {
  constexpr int KP = G * R, T0 = tile_rows_c(KP, 0), T1 = tile_rows_c(KP, 1);
  if constexpr ((G == 4 || G == 8) && R == 8) {
    // heterogeneous row sweep on 12 waves (3 per SIMD, 168 VGPRs: no spills) instead of 16 (128 VGPRs, ~45 spilled)
    if (cfg == 2 && kind == 0 && a.fixed_alpha <= 0.0 && (loss == LOSS_PER_OBS || loss == LOSS_PER_OBS_NOTRIG)) // (legacy one-kernel form only)
      return loss == LOSS_PER_OBS ? launch_tiled_inst<G, R, 12, T1, 2>(kind, a, st) : launch_tiled_inst<G, R, 12, T1, 4>(kind, a, st);
  }
  if (cfg == 2) cfg = 1;
  if constexpr ((G == 4 || G == 8) && R == 8) { // double-buffered half tiles with 1 / 2 loader waves (cfg 11 / 12)
#define GLRM_TLW(LOSSV) (cfg == 11 ? launch_tiled_inst<G, R, 16, T1, LOSSV, 1>(kind, a, st) : launch_tiled_inst<G, R, 16, T1, LOSSV, 2>(kind, a, st))
    if (cfg == 11 || cfg == 12) {
      switch (loss) {
        case LOSS_QUAD_UNIFORM: return GLRM_TLW(0);
        case LOSS_SEGMENT: return GLRM_TLW(1);
        case LOSS_SEGMENT_NOTRIG: return GLRM_TLW(3);
        case LOSS_PER_OBS_NOTRIG: return GLRM_TLW(4);
        default: return GLRM_TLW(2);
      }
    }
#undef GLRM_TLW
  }
  if (cfg >= 11) cfg = 1;
#define GLRM_TL(LOSSV)                                                                   \
  (cfg ? launch_tiled_inst<G, R, 16, T1, LOSSV>(kind, a, st) : launch_tiled_inst<G, R, 8, T0, LOSSV>(kind, a, st))
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return GLRM_TL(0);
    case LOSS_SEGMENT: return GLRM_TL(1);
    case LOSS_SEGMENT_NOTRIG: return GLRM_TL(3);
    case LOSS_PER_OBS_NOTRIG: return GLRM_TL(4);
    default: return GLRM_TL(2);
  }
#undef GLRM_TL
}

static int launch_tiled(glrm_handle* h, int loss, int kind, const TiledArgs& a) {
  const bool lw_here = h->tile_lw > 0 && h->tile_cfg && ((h->tile_lw_sides >> ((kind == 0 || kind == 3 || kind == 4) ? 0 : 1)) & 1);
  const int cfg = lw_here ? 10 + h->tile_lw : (h->tile_cfg12 && h->tile_cfg) ? 2 : h->tile_cfg;
  switch (h->tG * 100 + h->tR) {
    case 402: return launch_tiled_layout<4, 2>(cfg, loss, kind, a, h->stream);
    case 404: return launch_tiled_layout<4, 4>(cfg, loss, kind, a, h->stream);
    case 408: return launch_tiled_layout<4, 8>(cfg, loss, kind, a, h->stream);
    case 808: return launch_tiled_layout<8, 8>(cfg, loss, kind, a, h->stream);
    case 1608: return launch_tiled_layout<16, 8>(cfg, loss, kind, a, h->stream);
    default: return fail(GLRM_ERR_UNSUPPORTED, "no tiled kernel for lane layout G=%d R=%d", h->tG, h->tR);
  }
}

template <int G, int R>
static void launch_col_small(int which, const TiledArgs& a, hipStream_t st) {
  const unsigned gx = (unsigned)((a.nseg + 4 * (64 / G) - 1) / (4 * (64 / G)));
  if (which == 0) hipLaunchKernelGGL((col_reduce_kernel<G, R>), dim3(gx), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((col_decide_kernel<G, R>), dim3(gx), dim3(256), 0, st, a);
}

void glrm_launch_col_small(int kp, int which, const TiledArgs& a, hipStream_t st) {
  switch (kp) {
    case 8: launch_col_small<4, 2>(which, a, st); break;
    case 16: launch_col_small<4, 4>(which, a, st); break;
    case 32: launch_col_small<4, 8>(which, a, st); break;
    case 64: launch_col_small<8, 8>(which, a, st); break;
    default: launch_col_small<16, 8>(which, a, st); break;
  }
}

static void launch_col_small_any(glrm_handle* h, int which, const TiledArgs& a) { glrm_launch_col_small(h->kp, which, a, h->stream); }

// The tiled variants of run_sweep's launch.  Rows: one kernel.  Columns: pass 1 -> reduce -> rounds of
// (trial pass, decide) until no column is still searching (the count is read back once per round).
int glrm_run_tiled(glrm_handle* h, bool rows, int loss, int loss_by_segment, double min_stepsize, int eval_only) {
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
  a.descid = rows ? h->rowdescid : nullptr;
  a.stagger = env_int("GLRM_HIP_TILE_STAGGER", 0);
  a.udesc = h->udesc;
  a.n_udesc = h->n_udesc;
  if (rows && h->rng_e >= 0) { // glrm_hip_step_x_range
    const int64_t s0 = h->rng_b;
    a.nseg = h->rng_e - s0;
    if (a.nseg <= 0) return GLRM_OK;
    a.ptr += s0; a.alpha += s0; a.own_offset += s0;
    if (!a.reg_single) a.regs += s0;
    a.trials += s0; a.accepts += s0;
  }
  const bool row_rounds = rows && !h->row_split && (h->tile_rounds & 1) && !eval_only && a.fixed_alpha <= 0.0 && h->actlist;
  // lane-per-segment passes (glrm_lane.hpp): the ProxGradParams half-steps and the evaluation pass of the sides that run that family
  const bool lane_here = h->lane[rows ? 0 : 1] && glrm_lane_loss_ok(h, loss) && a.fixed_alpha <= 0.0 && (rows ? row_rounds : true);
  if (rows && !h->row_split && !row_rounds) {
    a.segperm = h->rng_e >= 0 ? nullptr : h->rowperm; // a sub-range sweep keeps the natural order
    return launch_tiled(h, loss, 0, a);
  }
  if (rows) { // pass buffers of the rows (row_split: super-tile passes, glrm_setup_tiled; rounds: one super-tile)
    const int64_t s0 = h->rng_e >= 0 ? h->rng_b : 0;
    a.nsup = h->nsup_r;
    a.tiles_per_sup = h->tiles_per_sup_r;
    a.part = h->part_r + s0 * (int64_t)h->nsup_r * (h->kp + 2); a.gsum = h->gsum_r + s0 * (int64_t)h->kp; a.trial = h->trial_r + s0 * (int64_t)h->kp;
    a.jold = h->jold_r + s0; a.active = h->active_r + s0; a.ntrial = h->ntrial_r + s0; a.nactive = h->nactive;
    a.segperm = h->rng_e >= 0 ? nullptr : h->rowperm;
  } else {
    a.nsup = h->nsup;
    a.tiles_per_sup = h->tiles_per_sup;
    a.part = h->part; a.gsum = h->gsum; a.trial = h->trialbuf; a.jold = h->joldbuf;
    a.active = h->activebuf; a.ntrial = h->ntrialbuf; a.nactive = h->nactive;
    a.segperm = h->colperm;
    if (h->blk_nlong_c > 0) { // the columns at or above long_from run on the gather sweep beside the passes (run_sweep, glrm_hip.hip)
      a.long_from = h->blk_long_from;
      a.npass = h->blk_nshort_c;
      if (a.npass == 0) { // every local column is diverted
        HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
        return GLRM_OK;
      }
    }
  }
  if (lane_here) return glrm_run_lane(h, rows, loss, a, min_stepsize, eval_only);
  int rc;
  const bool lists = h->actlist && (rows ? (h->tile_rounds & 1) != 0 : (h->tile_rounds & 2) != 0) && a.nseg <= h->actlist_cap;
  int32_t* list[2] = {lists ? h->actlist : nullptr, lists ? h->actlist + h->actlist_cap : nullptr};
  int cur = 0; // the list the kernels of this stage append to
  HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
  a.actlist_out = list[cur];
  if (row_rounds) {
    if ((rc = launch_tiled(h, loss, 3, a))) return rc; // gradient pass + first trial; rejected rows are listed
  } else {
    // gradient pass: under glrm_hip_step_y_arrival in runs of super-tiles, each behind the blocks of X it reads (announced order)
    const int T_ = tile_rows(h->kp, h->tile_cfg);
    if (rows) rc = launch_tiled(h, loss, 1, a); // (the row view reads Y, which is complete)
    else rc = glrm_for_sup_runs_in_arrival_order(h, a.nsup, (int64_t)a.tiles_per_sup * T_, [&](int s0, int s1) {
      TiledArgs r = a;
      r.sup0 = s0;
      r.nsup_launch = s1 - s0;
      return launch_tiled(h, loss, 1, r);
    });
    if (rc) return rc;
    launch_col_small_any(h, 0, a);
  }
  HIPCK(hipGetLastError());
  if (eval_only || a.fixed_alpha > 0.0) return GLRM_OK;
  const TiledArgs full = a;
  // A segment leaves the search when a trial is accepted or its step size is no longer above min_stepsize (`while alpha > min_stepsize`,
  // proxgrad.jl:136,180): at most log(alpha / min_stepsize) / log(1 / 0.7) rounds (13 from alpha = 1 and the default 0.01).  With
  // min_stepsize = 0 a search whose trials are all rejected never ends in the reference either: 0.7 x 4.9e-324 rounds back to 4.9e-324,
  // alpha never reaches 0 (~2 090 rounds from alpha = 1 to the smallest denormal, then forever).  The bound below is a guard against
  // exactly that loop, never a silent cut: running into it is an error where the reference would hang.
  constexpr int MAX_ROUNDS = 4096;
  for (int round = 0;; ++round) {
    if (round == MAX_ROUNDS) return fail(GLRM_ERR_INVALID, "line search still running after %d rounds (min_stepsize %g)", MAX_ROUNDS, min_stepsize);
    unsigned int nact = 0;
    HIPCK(hipMemcpyAsync(&nact, h->nactive, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (nact == 0) break;
    HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
    TiledArgs t = full;
    // the trial pass: over the listed segments when they are the minority (a full grid keeps the length / kind order of segperm, which
    // pays while nearly every segment still takes part: the first trial of the columns)
    const bool compact = lists && (row_rounds || (int64_t)nact * 4 < full.nseg * 3);
    if (compact) {
      t.segperm = list[cur];
      t.nseg = nact;
      t.npass = 0;
    }
    if ((rc = launch_tiled(h, loss, rows && row_rounds ? 4 : 2, t))) return rc;
    TiledArgs d = full;
    if (lists) {
      d.actlist_in = list[cur];
      d.nact_in = nact;
      d.actlist_out = list[cur ^ 1];
      cur ^= 1;
    }
    launch_col_small_any(h, 1, d);
    HIPCK(hipGetLastError());
  }
  return GLRM_OK;
}
