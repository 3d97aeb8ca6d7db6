// glrm_multi.hip -- host side of the general sweeps (glrm_multi.hpp): multi-dimensional losses, block
// regularizers and offsets.  Part of libglrm_hip.so.
#include "glrm_multi.hpp"

#include "glrm_engine.hpp"

using namespace glrm;

// Column spans of Y (get_yidxs, src/losses.jl:76-93) and the routing decision.  Called for every handle: the spans
// are also what a scalar problem uses (ystart[f] = f) if its regularizers are later replaced by wrapped ones.
int glrm_setup_multi(glrm_handle* h, const glrm_problem* p) {
  std::vector<int64_t> ys((size_t)p->n + 1);
  ys[0] = 0;
  h->dmax = 1;
  h->multi_kmask = 0;
  bool multi = false;
  for (int64_t f = 0; f < p->n; ++f) {
    const glrm_loss& l = p->n_losses == 1 ? p->losses[0] : p->losses[f];
    const int d = l.dim > 1 ? l.dim : 1;
    h->multi_kmask |= l.kind >= GLRM_LOSS_MULTINOMIAL ? (1 << l.kind) : MULTI_KM_SCALAR; // the kinds of the WHOLE model (never the shard's)
    if (d > 1) multi = true;
    if (d > h->dmax) h->dmax = d;
    ys[f + 1] = ys[f] + d;
  }
  h->d = ys[p->n];
  for (int64_t i = 0; i < p->n_rx; ++i) if (p->rx[i].wrap) multi = true;
  for (int64_t i = 0; i < p->n_ry; ++i) if (p->ry[i].wrap) multi = true;
  h->multi = multi;
  if (multi && h->kp > 64)
    return fail(GLRM_ERR_UNSUPPORTED, "multi-dimensional losses / wrapped regularizers need k <= 64 (got %d)", h->k);
  if (multi && p->dense_A)
    return fail(GLRM_ERR_UNSUPPORTED, "the dense hand-over covers scalar QuadLoss with unwrapped regularizers only");
  HIPCK(hipMalloc((void**)&h->ystart, ((size_t)p->n + 1) * 8));
  HIPCK(hipMemcpyAsync(h->ystart, ys.data(), ((size_t)p->n + 1) * 8, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream)); // ys is a local
  h->ys_cb = ys[p->col_begin];
  return GLRM_OK;
}

// A column longer than one chunk (8192 observations; GLRM_HIP_MULTI_CHUNK overrides, 0 = never) is swept by several
// workgroups.  The chunk is a constant, so the grouping of a column's partial sums depends on its own list only.
static int setup_split(glrm_handle* h) {
  if (h->mtrial || h->col_nsplit < 0) return GLRM_OK;
  const int64_t chunk = env_int("GLRM_HIP_MULTI_CHUNK", 8192);
  // split or not is decided from the longest column of the WHOLE problem (glrm_signature) and the global column count, so that every
  // shard of a sharded fit runs the same kernels
  const int64_t longest = h->sig.max_col_len;
  const size_t blk = (size_t)h->dmax * h->kp;
  const int64_t nsplit = chunk > 0 ? (longest + chunk - 1) / chunk : 1;
  if (nsplit <= 1 || nsplit > 65535 || (size_t)h->n * nsplit * blk * 8 > ((size_t)2 << 30)) { h->col_nsplit = -1; return GLRM_OK; }
  h->col_chunk = chunk;
  h->col_nsplit = (int)nsplit;
  HIPCK(hipMalloc((void**)&h->mtrial, (size_t)h->d * h->kp * 8));
  HIPCK(hipMalloc((void**)&h->mpart_loss, (size_t)h->nl * nsplit * 8));
  HIPCK(hipMalloc((void**)&h->mpart_G, (size_t)h->nl * nsplit * blk * 8));
  HIPCK(hipMalloc((void**)&h->mgtot, (size_t)h->nl * blk * 8));
  HIPCK(hipMalloc((void**)&h->mobjold, (size_t)h->nl * 8));
  HIPCK(hipMalloc((void**)&h->mactive, (size_t)h->nl * 4));
  HIPCK(hipMalloc((void**)&h->mnactive, 4));
  return GLRM_OK;
}

// GDC = 8 when no embedding is wider (gradient registers per lane: 8 instead of 32); TRIG: see LOSS_*_NOTRIG in glrm_engine.hpp
// kind-specialised instantiations (glrm_multi.hpp: MULTI_KM_*) exist for the small-dimension, no-PeriodicLoss variants
static int kind_class(const glrm_handle* h) { // 0 = all kinds, 1 = MultinomialLoss only, 2 = MultinomialLoss + scalar losses, 3 = ordinal
  if (!env_int("GLRM_HIP_MULTI_KINDS", 1)) return 0; // (BvSLoss / MultinomialOrdinalLoss: the columns of an ordinal data frame)
  if (h->multi_kmask == MULTI_KM_MNL) return 1;
  if ((h->multi_kmask & ~(MULTI_KM_MNL | MULTI_KM_SCALAR)) == 0 && (h->multi_kmask & MULTI_KM_MNL)) return 2;
  return h->multi_kmask != 0 && (h->multi_kmask & ~MULTI_KM_ORD) == 0 ? 3 : 0;
}

template <bool GRAD>
static void launch_colpass(bool small, bool trig, int kc, dim3 grid, size_t lds, hipStream_t st, const SplitArgs& sa) {
  constexpr int D = GLRM_MAX_EMBEDDING_DIM;
  if (small) {
    if (trig) hipLaunchKernelGGL((multi_colpass_kernel<GRAD, 8, true>), grid, dim3(512), lds, st, sa);
    else if (kc == 1) hipLaunchKernelGGL((multi_colpass_kernel<GRAD, 8, false, MULTI_KM_MNL>), grid, dim3(512), lds, st, sa);
    else if (kc == 2) hipLaunchKernelGGL((multi_colpass_kernel<GRAD, 8, false, MULTI_KM_MNL | MULTI_KM_SCALAR>), grid, dim3(512), lds, st, sa);
    else if (kc == 3) hipLaunchKernelGGL((multi_colpass_kernel<GRAD, 8, false, MULTI_KM_ORD>), grid, dim3(512), lds, st, sa);
    else hipLaunchKernelGGL((multi_colpass_kernel<GRAD, 8, false>), grid, dim3(512), lds, st, sa);
  } else {
    if (trig) hipLaunchKernelGGL((multi_colpass_kernel<GRAD, D, true>), grid, dim3(512), lds, st, sa);
    else hipLaunchKernelGGL((multi_colpass_kernel<GRAD, D, false>), grid, dim3(512), lds, st, sa);
  }
}

static int run_split_cols(glrm_handle* h, const MultiArgs& a) {
  SplitArgs sa{};
  sa.m = a;
  sa.nsplit = h->col_nsplit;
  sa.chunk = h->col_chunk;
  sa.trial = h->mtrial;
  sa.part_loss = h->mpart_loss; sa.part_G = h->mpart_G; sa.gtot = h->mgtot; sa.objold = h->mobjold;
  sa.active = h->mactive; sa.nactive = h->mnactive;
  const int S = h->kp + 1, sl = 64 >> a.lgP;
  const size_t lds_pass = ((size_t)2 * h->dmax * S + 16 + (size_t)8 * sl * (S + 64)) * 8;
  const size_t lds_dec = ((size_t)2 * h->dmax * S + 64 + 16) * 8;
  const dim3 grid((unsigned)a.nseg, (unsigned)sa.nsplit);
  const bool small = h->dmax <= 8; // gradient registers per lane: 8 instead of 32 -> 4 waves per SIMD instead of 2
  hipStream_t st = h->stream;
  sa.point = a.own;
  if (a.mode == 1) { // losses only
    sa.round = -1;
    launch_colpass<false>(small, h->has_trig, kind_class(h), grid, lds_pass, st, sa);
    hipLaunchKernelGGL(multi_coldecide_kernel, dim3((unsigned)a.nseg), dim3(512), lds_dec, st, sa);
    HIPCK(hipGetLastError());
    return GLRM_OK;
  }
  sa.round = 0;
  HIPCK(hipMemsetAsync(h->mnactive, 0, 4, st));
  launch_colpass<true>(small, h->has_trig, kind_class(h), grid, lds_pass, st, sa);
  hipLaunchKernelGGL(multi_coldecide_kernel, dim3((unsigned)a.nseg), dim3(512), lds_dec, st, sa);
  HIPCK(hipGetLastError());
  if (a.mode == 2) return GLRM_OK;
  sa.point = h->mtrial;
  for (int round = 1; round < 4096; ++round) { // a column makes at most ~13 + log(alpha growth) trials
    unsigned int nact = 0;
    HIPCK(hipMemcpyAsync(&nact, h->mnactive, 4, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if (nact == 0) break;
    sa.round = round;
    HIPCK(hipMemsetAsync(h->mnactive, 0, 4, st));
    launch_colpass<false>(small, h->has_trig, kind_class(h), grid, lds_pass, st, sa);
    hipLaunchKernelGGL(multi_coldecide_kernel, dim3((unsigned)a.nseg), dim3(512), lds_dec, st, sa);
    HIPCK(hipGetLastError());
  }
  return GLRM_OK;
}

int glrm_run_multi(glrm_handle* h, bool rows, double min_stepsize, int eval_only) {
  MultiArgs a{};
  a.nseg = rows ? h->ml : h->nl;
  a.ptr = rows ? h->rowptr : h->colptr;
  a.idx = rows ? h->colidx : h->rowidx;
  a.vals = rows ? h->rowvals : h->colvals;
  a.own = rows ? h->X : h->Y;
  a.own_offset = rows ? h->rb : h->cb;
  a.other = rows ? h->Y : h->X;
  a.ystart = h->ystart;
  a.losses = h->losses;
  a.loss_single = h->n_losses == 1;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.alpha = rows ? h->alpharow : h->alphacol;
  a.obj = rows ? nullptr : h->objcol;
  a.k = h->k; a.kp = h->kp; a.dmax = h->dmax;
  a.lgP = 2;
  while ((1 << a.lgP) < h->k || (1 << a.lgP) < h->dmax) ++a.lgP;
  a.mode = eval_only ? 1 : (h->fixed_alpha > 0.0 ? 2 : 0);
  a.fixed_alpha = h->fixed_alpha;
  a.min_stepsize = min_stepsize;
  a.trials = (eval_only || a.mode == 2) ? nullptr : (rows ? h->trials_r : h->trials_c);
  a.accepts = (eval_only || a.mode == 2) ? nullptr : (rows ? h->accepts_r : h->accepts_c);
  if (rows && h->rng_e >= 0) {
    const int64_t s0 = h->rng_b;
    a.nseg = h->rng_e - s0;
    a.ptr += s0; a.alpha += s0; a.own_offset += s0;
    if (!a.reg_single) a.regs += s0;
    if (a.trials) a.trials += s0;
    if (a.accepts) a.accepts += s0;
  }
  if (a.nseg <= 0) return GLRM_OK;
  if (rows) {
    const size_t lds = multi_lds_doubles(true, 1, h->kp, h->dmax, a.lgP) * 8;
    // every embedding dimension <= 8: the opposing block of an observation is held in registers (glrm_multi.hpp: multi_pass, RD)
    const bool regs = h->dmax <= 8 && env_int("GLRM_HIP_MULTI_REGS", 1);
    const int kc = kind_class(h);
    if (regs) {
      if (h->has_trig) hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, true, 8>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
      else if (kc == 1) hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, false, 8, MULTI_KM_MNL>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
      else if (kc == 2) hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, false, 8, MULTI_KM_MNL | MULTI_KM_SCALAR>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
      else if (kc == 3) hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, false, 8, MULTI_KM_ORD>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
      else hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, false, 8>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
    } else if (h->has_trig) hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, true>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
    else hipLaunchKernelGGL((multi_sweep_kernel<true, 1, GLRM_MAX_EMBEDDING_DIM, false>), dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
  } else {
    const int rc = setup_split(h); // decides once per handle whether the columns are long enough to split
    if (rc) return rc;
    if (h->col_nsplit > 1) return run_split_cols(h, a);
    const size_t lds = multi_lds_doubles(false, 8, h->kp, h->dmax, a.lgP) * 8;
    if (h->dmax <= 8) {
      const int kc = kind_class(h);
      if (h->has_trig) hipLaunchKernelGGL((multi_sweep_kernel<false, 8, 8, true, 8>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
      else if (kc == 1) hipLaunchKernelGGL((multi_sweep_kernel<false, 8, 8, false, 8, MULTI_KM_MNL>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
      else if (kc == 2) hipLaunchKernelGGL((multi_sweep_kernel<false, 8, 8, false, 8, MULTI_KM_MNL | MULTI_KM_SCALAR>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
      else if (kc == 3) hipLaunchKernelGGL((multi_sweep_kernel<false, 8, 8, false, 8, MULTI_KM_ORD>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
      else hipLaunchKernelGGL((multi_sweep_kernel<false, 8, 8, false, 8>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
    } else {
      if (h->has_trig) hipLaunchKernelGGL((multi_sweep_kernel<false, 8, GLRM_MAX_EMBEDDING_DIM, true>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
      else hipLaunchKernelGGL((multi_sweep_kernel<false, 8, GLRM_MAX_EMBEDDING_DIM, false>), dim3((unsigned)a.nseg), dim3(512), lds, h->stream, a);
    }
  }
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

int glrm_run_multi_penalty(glrm_handle* h, bool rows) {
  PenaltyArgs a{};
  a.nseg = rows ? h->ml : h->nl;
  if (a.nseg <= 0) return GLRM_OK;
  a.own = rows ? h->X : h->Y;
  a.own_offset = rows ? h->rb : h->cb;
  a.ystart = rows ? nullptr : h->ystart;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.k = h->k; a.kp = h->kp;
  a.out = rows ? h->objrow : h->objcol;
  const size_t lds = ((size_t)(rows ? 1 : h->dmax) * (h->kp + 1) + 16) * 8;
  hipLaunchKernelGGL(multi_penalty_kernel, dim3((unsigned)a.nseg), dim3(64), lds, h->stream, a);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}
