// glrm_dense.hip -- host side of the fully observed QuadLoss path on the fp64 matrix cores
// (kernels in glrm_dense.hpp; line-search bookkeeping kernels shared with glrm_tiled.hpp).
#include <hip/hip_runtime.h>

#include <cmath>

#include "glrm_engine.hpp"
#include "glrm_tiled.hpp"
#include "glrm_dense.hpp"

using namespace glrm;

static int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }

// super-tiles over the opposing dimension: a function of that dimension only (never of the shard layout),
// so the partial-sum order -- and therefore the result bits -- do not depend on the number of GPUs
static void pick_sup(int64_t n_other, int& nsup, int64_t& vps) {
  int64_t s = (n_other + 32767) / 32768;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  vps = round_up((n_other + s - 1) / s, DENSE_TN);
  nsup = (int)((n_other + vps - 1) / vps);
}

static int pack_view(glrm_handle* h, const double* dsrc, int64_t ldsrc, int colmajor, int transpose, int64_t seg0,
                     int64_t nseg, int64_t nother, double** dst, int64_t* lda) {
  const int64_t nseg_pad = round_up(nseg > 0 ? nseg : 1, 256); // whole 16-wave workgroups stay in bounds
  *lda = round_up(nother, 64);
  HIPCK(hipMalloc((void**)dst, (size_t)nseg_pad * (size_t)*lda * 8));
  const dim3 grid((unsigned)(*lda / 32), (unsigned)(nseg_pad / 32));
  hipLaunchKernelGGL(dense_pack_kernel, grid, dim3(256), 0, h->stream, dsrc, ldsrc, colmajor, transpose, seg0, nseg, nother, *dst, *lda);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

template <typename T>
static int alloc_arr(T** p, int64_t count) {
  HIPCK(hipMalloc((void**)p, (size_t)(count > 0 ? count : 1) * sizeof(T)));
  return GLRM_OK;
}

int glrm_setup_dense(glrm_handle* h, const glrm_problem* p) {
  if (p->rowptr || p->colidx || p->rowvals || p->colptr || p->rowidx || p->colvals)
    return fail(GLRM_ERR_INVALID, "dense_A and the observation lists are mutually exclusive");
  if (p->dense_reserved != 0) return fail(GLRM_ERR_INVALID, "glrm_problem.dense_reserved must be 0");
  if (!(p->n_losses == 1 && p->losses[0].kind == GLRM_LOSS_QUAD))
    return fail(GLRM_ERR_UNSUPPORTED, "the dense path needs one QuadLoss descriptor for all columns; pass observation lists otherwise");
  if (!(h->kp == 16 || h->kp == 32 || h->kp == 64))
    return fail(GLRM_ERR_UNSUPPORTED, "the dense path supports ranks 9..64; pass observation lists otherwise");
  const int64_t need_ld = p->dense_colmajor ? p->m : p->n;
  if (p->dense_ld < need_ld) return fail(GLRM_ERR_INVALID, "dense_ld is smaller than the matrix dimension");
  h->dense = true;
  h->dense_scale = p->losses[0].scale;
  h->nnz_r = h->ml * h->n;
  h->nnz_c = h->nl * h->m;
  // Bring the caller's matrix to the device if needed.  Whole problem on this handle: one upload, both packed views come from it.
  // A shard of a multi-GPU fit only needs its row block (for the X half-step) and its column block (for the Y half-step): those
  // two sub-matrices are uploaded one after the other (2D copies), never the whole matrix -- at C3 on 8 GPUs 10 + 10 GB per
  // device instead of 80 GB.  The NaN check (src/glrm.jl:63-71) covers the shard's row block; the row blocks partition A.
  const bool on_dev = (p->flags & GLRM_PROBLEM_DEVICE_ARRAYS) != 0;
  const bool whole = h->rb == 0 && h->re == h->m && h->cb == 0 && h->ce == h->n;
  const int cm = p->dense_colmajor ? 1 : 0;
  const int64_t ld = p->dense_ld;
  if (!on_dev) { // walk along the contiguous dimension of the caller's storage order (column-major: what the Julia shim passes)
    bool bad = false;
    int64_t bi = 0, bj = 0;
    if (cm) {
      for (int64_t j = 0; j < h->n && !bad; ++j) {
        const double* col = p->dense_A + j * ld;
        for (int64_t i = h->rb; i < h->re; ++i)
          if (std::isnan(col[i])) { bad = true; bi = i; bj = j; break; }
      }
    } else {
      for (int64_t i = h->rb; i < h->re && !bad; ++i) {
        const double* row = p->dense_A + i * ld;
        for (int64_t j = 0; j < h->n; ++j)
          if (std::isnan(row[j])) { bad = true; bi = i; bj = j; break; }
      }
    }
    if (bad) return fail(GLRM_ERR_NONFINITE, "Observed value in entry (%lld, %lld) is NaN.", (long long)bi, (long long)bj);
  }
  int rc = GLRM_OK;
  if (on_dev) {
    rc = pack_view(h, p->dense_A, ld, cm, 0, h->rb, h->ml, h->n, &h->Arow, &h->lda_r);
    if (!rc) rc = pack_view(h, p->dense_A, ld, cm, 1, h->cb, h->nl, h->m, &h->Acol, &h->lda_c);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(GLRM_ERR_HIP, "dense pack failed");
  } else if (whole) {
    double* tmp = nullptr;
    const size_t runs = cm ? (size_t)p->n : (size_t)p->m, run = cm ? (size_t)p->m : (size_t)p->n; // contiguous runs and their length
    HIPCK(hipMalloc((void**)&tmp, runs * run * 8));
    if (hipMemcpy2DAsync(tmp, run * 8, p->dense_A, (size_t)ld * 8, run * 8, runs, hipMemcpyHostToDevice, h->stream) != hipSuccess)
      rc = fail(GLRM_ERR_HIP, "upload of the dense matrix failed");
    if (!rc) rc = pack_view(h, tmp, (int64_t)run, cm, 0, 0, h->ml, h->n, &h->Arow, &h->lda_r);
    if (!rc) rc = pack_view(h, tmp, (int64_t)run, cm, 1, 0, h->nl, h->m, &h->Acol, &h->lda_c);
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(GLRM_ERR_HIP, "dense pack failed");
    (void)hipFree(tmp);
  } else {
    // (which, first segment, segments, extent of the other dimension) for the row block and the column block
    for (int which = 0; which < 2 && !rc; ++which) {
      const int64_t s0 = which ? h->cb : h->rb, ns = which ? h->nl : h->ml, other = which ? h->m : h->n;
      if (ns <= 0) { rc = pack_view(h, p->dense_A, ld, cm, which, 0, 0, other, which ? &h->Acol : &h->Arow, which ? &h->lda_c : &h->lda_r); continue; }
      // the block as a dense sub-matrix in the caller's own storage order: segments run along the caller's rows (which = 0) / columns
      const bool seg_is_run = (which == 0) != (cm != 0); // are whole segments contiguous runs of the source?
      const size_t width = (size_t)(seg_is_run ? other : ns) * 8, height = (size_t)(seg_is_run ? ns : other);
      const double* src = p->dense_A + (seg_is_run ? s0 * ld : s0);
      double* tmp = nullptr;
      HIPCK(hipMalloc((void**)&tmp, width * height));
      if (hipMemcpy2DAsync(tmp, width, src, (size_t)ld * 8, width, height, hipMemcpyHostToDevice, h->stream) != hipSuccess)
        rc = fail(GLRM_ERR_HIP, "upload of the dense block failed");
      // inside tmp the block starts at segment 0 and has leading dimension width / 8
      if (!rc) rc = pack_view(h, tmp, (int64_t)(width / 8), cm, which, 0, ns, other, which ? &h->Acol : &h->Arow, which ? &h->lda_c : &h->lda_r);
      if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(GLRM_ERR_HIP, "dense pack failed");
      (void)hipFree(tmp);
    }
  }
  if (rc) return rc;
  pick_sup(h->n, h->nsup_r, h->vps_r);
  pick_sup(h->m, h->nsup_c, h->vps_c);
  const int64_t ml1 = h->ml > 0 ? h->ml : 1, nl1 = h->nl > 0 ? h->nl : 1;
  const int ps = h->kp + 2;
  if ((rc = alloc_arr(&h->part_r, ml1 * h->nsup_r * ps))) return rc;
  if ((rc = alloc_arr(&h->gsum_r, ml1 * h->kp))) return rc;
  if ((rc = alloc_arr(&h->trial_r, ml1 * h->kp))) return rc;
  if ((rc = alloc_arr(&h->jold_r, ml1))) return rc;
  if ((rc = alloc_arr(&h->active_r, ml1))) return rc;
  if ((rc = alloc_arr(&h->ntrial_r, ml1))) return rc;
  if ((rc = alloc_arr(&h->part, nl1 * h->nsup_c * ps))) return rc;
  if ((rc = alloc_arr(&h->gsum, nl1 * h->kp))) return rc;
  if ((rc = alloc_arr(&h->trialbuf, nl1 * h->kp))) return rc;
  if ((rc = alloc_arr(&h->joldbuf, nl1))) return rc;
  if ((rc = alloc_arr(&h->activebuf, nl1))) return rc;
  if ((rc = alloc_arr(&h->ntrialbuf, nl1))) return rc;
  if ((rc = alloc_arr(&h->nactive, 1))) return rc;
  // glrm_options.quad_gram (GLRM_HIP_DENSE_GRAM overrides): line-search trials from the quadratic form, no pass over A per trial
  h->dense_gram = env_int("GLRM_HIP_DENSE_GRAM", h->opts.quad_gram ? 1 : 0) != 0;
  if (h->dense_gram) {
    if ((rc = alloc_arr(&h->gramH, (int64_t)h->kp * h->kp))) return rc;
    if ((rc = alloc_arr(&h->gram_part, (int64_t)GRAM_BLOCKS * h->kp * h->kp))) return rc;
    if ((rc = alloc_arr(&h->jloss_r, ml1))) return rc;
    if ((rc = alloc_arr(&h->jloss_c, nl1))) return rc;
  }
  return GLRM_OK;
}

template <int KP>
static void launch_gram_inst(int which, const TiledArgs& a, const double* other, int64_t n_other, double* part, double* H, double scale, hipStream_t st) {
  if (which == 0) {
    hipLaunchKernelGGL((dense_gram_partial_kernel<KP>), dim3(GRAM_BLOCKS), dim3(256), 0, st, other, n_other, part);
    hipLaunchKernelGGL((dense_gram_final_kernel<KP>), dim3((KP * KP + 255) / 256), dim3(256), 0, st, part, H);
  } else {
    hipLaunchKernelGGL((dense_gram_trial_kernel<KP>), dim3((unsigned)((a.nseg + 255) / 256)), dim3(256), 0, st, a, H, scale);
  }
}

// which = 0: H = other other' (kp x kp); which = 1: the trial objectives of the active segments from the quadratic form
static void launch_gram(int kp, int which, const TiledArgs& a, const double* other, int64_t n_other, double* part, double* H, double scale, hipStream_t st) {
  switch (kp) {
    case 16: launch_gram_inst<16>(which, a, other, n_other, part, H, scale, st); break;
    case 32: launch_gram_inst<32>(which, a, other, n_other, part, H, scale, st); break;
    default: launch_gram_inst<64>(which, a, other, n_other, part, H, scale, st); break;
  }
}

template <int KP, int NWD>
static void launch_dense_inst(bool grad, const DenseArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)((a.nseg + NWD * 16 - 1) / (NWD * 16)), (unsigned)a.nsup);
  if (grad) hipLaunchKernelGGL((dense_pass_kernel<KP, true, NWD>), grid, dim3(NWD * 64), 0, st, a);
  else hipLaunchKernelGGL((dense_pass_kernel<KP, false, NWD>), grid, dim3(NWD * 64), 0, st, a);
}

template <int KP>
static void launch_dense_pass(bool grad, const DenseArgs& a, hipStream_t st) {
  // 16-wave workgroups (256 segments share one staged tile) unless the problem is too small to fill the chip with them
  int nw = a.nseg * (int64_t)a.nsup >= 256 * 256 ? 16 : 4;
  nw = env_int("GLRM_HIP_DENSE_NW", nw) == 16 ? 16 : 4; // tuning override
  if (nw == 16) launch_dense_inst<KP, 16>(grad, a, st);
  else launch_dense_inst<KP, 4>(grad, a, st);
}

static void launch_dense_any(int kp, bool grad, const DenseArgs& a, hipStream_t st) {
  switch (kp) {
    case 16: launch_dense_pass<16>(grad, a, st); break;
    case 32: launch_dense_pass<32>(grad, a, st); break;
    default: launch_dense_pass<64>(grad, a, st); break;
  }
}

// One half-step on the dense path: pass 1 (residuals, objective, gradient on the matrix cores) -> per-segment
// reduce + first trial point -> rounds of (trial objective pass, decide) until no segment is still searching.
int glrm_run_dense(glrm_handle* h, bool rows, double min_stepsize, int eval_only) {
  const int64_t nseg = rows ? h->ml : h->nl;
  if (nseg <= 0) return GLRM_OK;
  DenseArgs d{};
  d.nseg = nseg;
  d.xsrc = rows ? h->X : h->Y;
  d.own_offset = rows ? h->rb : h->cb;
  d.other = rows ? h->Y : h->X;
  d.n_other = rows ? h->n : h->m;
  d.A = rows ? h->Arow : h->Acol;
  d.lda = rows ? h->lda_r : h->lda_c;
  d.scale = h->dense_scale;
  d.nsup = rows ? h->nsup_r : h->nsup_c;
  d.vec_per_sup = rows ? h->vps_r : h->vps_c;
  d.part = rows ? h->part_r : h->part;
  d.active = rows ? h->active_r : h->activebuf;
  TiledArgs a{};
  a.nseg = nseg;
  a.ptr = nullptr;
  a.dense_len = d.n_other;
  a.own = rows ? h->X : h->Y;
  a.own_offset = d.own_offset;
  a.alpha = rows ? h->alpharow : h->alphacol;
  a.obj = rows ? nullptr : h->objcol;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.k = h->k;
  a.min_stepsize = min_stepsize;
  a.trials = rows ? h->trials_r : h->trials_c;
  a.accepts = rows ? h->accepts_r : h->accepts_c;
  a.nsup = d.nsup;
  a.part = d.part;
  a.gsum = rows ? h->gsum_r : h->gsum;
  a.trial = rows ? h->trial_r : h->trialbuf;
  a.jold = rows ? h->jold_r : h->joldbuf;
  a.active = rows ? h->active_r : h->activebuf;
  a.ntrial = rows ? h->ntrial_r : h->ntrialbuf;
  a.nactive = h->nactive;
  a.eval_only = eval_only;
  a.fixed_alpha = eval_only ? 0.0 : h->fixed_alpha;
  const bool gram = h->dense_gram && !eval_only && a.fixed_alpha <= 0.0;
  a.jloss = gram ? (rows ? h->jloss_r : h->jloss_c) : nullptr;
  HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
  if (gram) launch_gram(h->kp, 0, a, d.other, d.n_other, h->gram_part, h->gramH, d.scale, h->stream);
  launch_dense_any(h->kp, true, d, h->stream);
  glrm_launch_col_small(h->kp, 0, a, h->stream);
  HIPCK(hipGetLastError());
  if (eval_only || a.fixed_alpha > 0.0) return GLRM_OK;
  DenseArgs t = d;
  t.xsrc = a.trial; // trial points are stored per local segment
  t.own_offset = 0;
  constexpr int MAX_ROUNDS = 4096; // see glrm_run_tiled: a guard against a loop that cannot end, never a silent cut of the search
  for (int round = 0;; ++round) {
    if (round == MAX_ROUNDS) return fail(GLRM_ERR_INVALID, "line search still running after %d rounds (min_stepsize %g)", MAX_ROUNDS, min_stepsize);
    unsigned int nact = 0;
    HIPCK(hipMemcpyAsync(&nact, h->nactive, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    if (nact == 0) break;
    HIPCK(hipMemsetAsync(h->nactive, 0, 4, h->stream));
    if (gram) launch_gram(h->kp, 1, a, d.other, d.n_other, h->gram_part, h->gramH, d.scale, h->stream);
    else launch_dense_any(h->kp, false, t, h->stream);
    glrm_launch_col_small(h->kp, 1, a, h->stream);
    HIPCK(hipGetLastError());
  }
  return GLRM_OK;
}
