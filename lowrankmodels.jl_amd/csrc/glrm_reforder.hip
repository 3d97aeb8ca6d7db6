// glrm_reforder.hip -- glrm_options.sum_order = 1 (GLRM_ORDER_REFERENCE): the half-steps of fit!(glrm, ProxGradParams) with every sum
// added in the REFERENCE's order (SURVEY.md Appendix A.3, section 8(b) `line_search_sum_order`).  A VALIDATION path, not a fast one.
//
// Why.  The accept test of the line search is a strict `<` between two long sums (src/algorithms/proxgrad.jl:143,187).  The engine's
// sweep families add the same fp64 terms in orders fixed by their lane layouts; on a trajectory that amplifies rounding (the NNMF recipe
// of BASELINE config 4: 1e-16 -> 8e-6 of the objective and 7e-3 of individual factor entries within 100 iterations)
// a checker that adds in the reference's order cannot tell an engine bug from summation order.  Rounds 3-4 closed that gap from the
// checker's side (the oracle adopts the engine's order and then equals it bit for bit).  This file closes it from the engine's side: the
// same per-entry arithmetic (glrm_device.hpp: the formulas every family evaluates) with
//   * u = <x_e, y_f>: one fma chain over the components 0 .. k-1                          (gemm! / x'Y per entry; the oracle's dotk)
//   * the gradient: g += L'(u, a) y_f per component in LIST order, one accumulator          (axpy!, proxgrad.jl:127,170)
//   * row_objective: err += L(u, a) in list order from 0.0, then += r(x)                     (src/evaluate_fit.jl:28-36)
//   * col_objective: 0.0 + loss + r(y) with the loss sum by Julia's pairwise reduce(+) -- sequential below 1024 terms, else split at
//     first + (last - first) >> 1 -- for DiffLoss / ClassificationLoss columns (map! then reduce(+), src/losses.jl:633-638) and
//     sequentially from 0 for OrdinalHingeLoss / PoissonLoss (:623-630)
//   * r(x) = scale * (sum of x_c^2 | sum of |x_c|) with separate multiply and add in component order (src/regularizers.jl:58,88)
//   * sum(obj_by_col) by the same pairwise rule (proxgrad.jl:205), on the host (glrm_hip_sum)
// ONE LANE owns a segment and walks it alone: no cross-lane step exists, so nothing depends on a lane layout.  Against the oracle in its
// default (reference) order the factors are then equal to the last bit for the losses both sides evaluate with the same instructions
// (everything but the exp / log / sin / cos based ones) -- tests/test_gpu_reforder.py, tests/test_gpu_jref.py.
//
// Cost: no parallelism inside a segment, 8k bytes gathered per observation and pass by a single lane, k-long dependent fma chains.  The
// 1e8-observation J_ref fixtures run in seconds per 100 iterations, which is what the mode is for.  ProxGradParams half-steps and the
// evaluation passes only (scalar losses, list problems, k <= 64); the sparse solver's fixed-step sweeps are not restated here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "glrm_device.hpp"
#include "glrm_engine.hpp"

using namespace glrm;

namespace {

struct RefArgs {
  int64_t nseg;          // local segments of this launch
  const int64_t* ptr;    // nseg + 1 (already offset to the first segment)
  const int32_t* idx;
  const double* vals;
  double* own;           // factor being updated (global array, ld = KP)
  int64_t own_offset;    // global id of local segment 0
  const double* other;   // opposing factor
  double* alpha;         // per local segment
  double* obj;           // per GLOBAL segment (columns: obj_by_col), nullable
  const glrm_loss* losses;
  int64_t n_losses;
  const glrm_reg* regs;
  int reg_single;
  int k;
  double min_stepsize;
  int32_t* trials;
  int32_t* accepts;
  int eval_only;         // columns: obj[seg] = loss sum at the current point, nothing else
};

// glrm_cpu_reg_evaluate, oracle/glrm_oracle.c (src/regularizers.jl:58,88,95,103-112,300-316): component order, multiply then add
template <int KP>
__device__ __forceinline__ double ref_reg_eval(const RegDesc& r, const double (&x)[KP], int k) {
  switch (r.kind) {
    case GLRM_REG_QUAD: {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < KP; ++c)
        if (c < k) s += x[c] * x[c];
      return r.scale * s;
    }
    case GLRM_REG_ONE: {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < KP; ++c)
        if (c < k) s += fabs(x[c]);
      return r.scale * s;
    }
    case GLRM_REG_NONNEG: {
      bool neg = false;
#pragma unroll
      for (int c = 0; c < KP; ++c) neg = neg || (c < k && x[c] < 0);
      return neg ? __builtin_inf() : 0.0;
    }
    case GLRM_REG_UNIT_ONE_SPARSE: {
      int ones = 0;
      bool other = false;
#pragma unroll
      for (int c = 0; c < KP; ++c)
        if (c < k && x[c] != 0) {
          if (x[c] == 1) ++ones; else other = true;
        }
      return (other || ones > 1) ? __builtin_inf() : 0.0;
    }
    default:
      return 0.0;
  }
}

// glrm_cpu_reg_prox (src/regularizers.jl:34,56,83-86,93,103,297)
template <int KP>
__device__ __forceinline__ void ref_reg_prox(const RegDesc& r, double (&u)[KP], int k, double alpha) {
  switch (r.kind) {
    case GLRM_REG_QUAD: {
      const double f = 1 / (1 + 2 * alpha * r.scale);
#pragma unroll
      for (int c = 0; c < KP; ++c) u[c] = f * u[c];
      break;
    }
    case GLRM_REG_ONE: {
      const double t = r.scale * alpha;
#pragma unroll
      for (int c = 0; c < KP; ++c) u[c] = fmax(u[c] - t, 0.0) + fmin(u[c] + t, 0.0);
      break;
    }
    case GLRM_REG_NONNEG: {
#pragma unroll
      for (int c = 0; c < KP; ++c) u[c] = u[c] > 0 ? u[c] : 0.0;
      break;
    }
    case GLRM_REG_UNIT_ONE_SPARSE: { // e_{argmax u}, first maximal index
      int idx = 0;
      double best = u[0];
#pragma unroll
      for (int c = 1; c < KP; ++c)
        if (c < k && u[c] > best) { best = u[c]; idx = c; }
#pragma unroll
      for (int c = 0; c < KP; ++c) u[c] = c == idx ? 1.0 : 0.0;
      break;
    }
    default:
      break;
  }
#pragma unroll
  for (int c = 0; c < KP; ++c)
    if (c >= k) u[c] = 0.0; // the padding stays exactly zero
}

// <x, y>: s = fma(x[c], y[c], s), c = 0 .. k-1 (the oracle's dotk); y is a row of the opposing factor in memory
template <int KP>
__device__ __forceinline__ double ref_dot(const double (&x)[KP], const double* __restrict__ y, int k) {
  double s = 0.0;
  const double2* y2 = reinterpret_cast<const double2*>(y);
#pragma unroll
  for (int c = 0; c < KP / 2; ++c) {
    const double2 v = y2[c];
    if (2 * c < k) s = fma(x[2 * c], v.x, s);
    if (2 * c + 1 < k) s = fma(x[2 * c + 1], v.y, s);
  }
  return s;
}

__device__ __forceinline__ bool ref_single_dim(int kind) { return !(kind == GLRM_LOSS_POISSON || kind == GLRM_LOSS_ORDINAL_HINGE); }

template <int KP>
__device__ __forceinline__ void load_vec(double (&x)[KP], const double* p) {
  const double2* p2 = reinterpret_cast<const double2*>(p);
#pragma unroll
  for (int c = 0; c < KP / 2; ++c) {
    const double2 v = p2[c];
    x[2 * c] = v.x;
    x[2 * c + 1] = v.y;
  }
}
template <int KP>
__device__ __forceinline__ void store_vec(double* p, const double (&x)[KP]) {
  double2* p2 = reinterpret_cast<double2*>(p);
#pragma unroll
  for (int c = 0; c < KP / 2; ++c) p2[c] = make_double2(x[2 * c], x[2 * c + 1]);
}

// ------------------------------------------------------------------------------------------------------------------------ rows
// src/algorithms/proxgrad.jl:118-156 for one row per lane
template <int KP, bool TRIG>
__global__ void __launch_bounds__(64) ref_row_kernel(const RefArgs a) {
  const int64_t el = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (el >= a.nseg) return;
  const int k = a.k;
  const int64_t b = a.ptr[el], e = a.ptr[el + 1];
  double* xp = a.own + (a.own_offset + el) * KP;
  double x[KP], g[KP], xn[KP];
  load_vec<KP>(x, xp);
#pragma unroll
  for (int c = 0; c < KP; ++c) g[c] = 0.0;
  // gradient (:122-132) and row_objective at x (:135) in one walk: each accumulator still adds its terms in list order
  double Jold = 0.0;
  for (int64_t t = b; t < e; ++t) {
    const int64_t f = a.idx[t];
    const double* y = a.other + f * KP;
    const double u = ref_dot<KP>(x, y, k);
    const LossDesc lo = load_loss(a.losses, a.n_losses == 1 ? 0 : f);
    double L, dL;
    loss_both<true, TRIG>(lo, u, a.vals[t], L, dL);
    Jold += L;
    const double2* y2 = reinterpret_cast<const double2*>(y);
#pragma unroll
    for (int c = 0; c < KP / 2; ++c) {
      const double2 v = y2[c];
      g[2 * c] = fma(dL, v.x, g[2 * c]);
      g[2 * c + 1] = fma(dL, v.y, g[2 * c + 1]);
    }
  }
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : el);
  Jold += ref_reg_eval<KP>(rd, x, k);
  const double l = (double)(e - b) + 1; // :134
  double alpha = a.alpha[el];
  int ntr = 0, nacc = 0;
  while (alpha > a.min_stepsize) { // :136
    const double stepsize = alpha / l;
#pragma unroll
    for (int c = 0; c < KP; ++c) xn[c] = fma(-stepsize, g[c], x[c]); // :140
    ref_reg_prox<KP>(rd, xn, k, stepsize);                            // :142
    double Jn = 0.0;
    for (int64_t t = b; t < e; ++t) {
      const int64_t f = a.idx[t];
      const double u = ref_dot<KP>(xn, a.other + f * KP, k);
      const LossDesc lo = load_loss(a.losses, a.n_losses == 1 ? 0 : f);
      double L, dL;
      loss_both<false, TRIG>(lo, u, a.vals[t], L, dL);
      Jn += L;
    }
    Jn += ref_reg_eval<KP>(rd, xn, k);
    ++ntr;
    if (Jn < Jold) { // :143
      store_vec<KP>(xp, xn);
      alpha *= 1.05;
      ++nacc;
      break;
    }
    alpha *= .7; // :147-153
    if (alpha < a.min_stepsize) {
      alpha = a.min_stepsize * 1.1;
      break;
    }
  }
  a.alpha[el] = alpha;
  if (a.trials) { a.trials[el] += ntr; a.accepts[el] += nacc; }
}

// ------------------------------------------------------------------------------------------------------------------------ columns
// The column's loss sum at y in the order of col_objective (src/evaluate_fit.jl:44-51): reduce(+, mapped) = Julia's pairwise sum for the
// single-dimensional losses, `out = 0; out += ...` for the others.  GRAD: the walk also accumulates the gradient (list order: the
// leaves of the pairwise tree are consecutive index ranges visited left to right).
template <int KP, bool TRIG, bool GRAD>
__device__ __forceinline__ double ref_col_loss(const RefArgs& a, const LossDesc& lo, int64_t b, int64_t e, const double (&y)[KP], double (&G)[KP]) {
  const int k = a.k;
  auto term = [&](int64_t t) -> double {
    const double* x = a.other + (int64_t)a.idx[t] * KP;
    const double u = ref_dot<KP>(y, x, k); // fma(x[c], y[c], s): the product commutes, the chain is the oracle's dotk(x_i, y)
    double L, dL;
    loss_both<GRAD, TRIG>(lo, u, a.vals[t], L, dL);
    if (GRAD) {
      const double2* x2 = reinterpret_cast<const double2*>(x);
#pragma unroll
      for (int c = 0; c < KP / 2; ++c) {
        const double2 v = x2[c];
        G[2 * c] = fma(dL, v.x, G[2 * c]);
        G[2 * c + 1] = fma(dL, v.y, G[2 * c + 1]);
      }
    }
    return L;
  };
  if (e <= b) return 0.0;
  if (!ref_single_dim(lo.kind)) {
    double out = 0.0;
    for (int64_t t = b; t < e; ++t) out += term(t);
    return out;
  }
  // Julia's mapreduce_impl (Base reduce.jl): [first, last] inclusive; fewer than 1024 + 1 elements: v[first] + v[first + 1], then += the
  // rest; else split at mid = first + (last - first) >> 1 and add the halves.  Iterative depth-first walk with an explicit stack.
  struct Frame { int64_t first, last; double v1; int stage; };
  Frame st[48];
  int sp = 0;
  st[0] = Frame{b, e - 1, 0.0, 0};
  double ret = 0.0;
  while (sp >= 0) {
    Frame& f = st[sp];
    if (f.stage == 0) {
      if (f.first == f.last) { ret = term(f.first); --sp; continue; }
      if (f.last - f.first < 1024) {
        double s = term(f.first);
        s += term(f.first + 1);
        for (int64_t i = f.first + 2; i <= f.last; ++i) s += term(i);
        ret = s;
        --sp;
        continue;
      }
      const int64_t mid = f.first + ((f.last - f.first) >> 1);
      f.stage = 1;
      st[sp + 1] = Frame{f.first, mid, 0.0, 0};
      ++sp;
    } else if (f.stage == 1) {
      f.v1 = ret;
      f.stage = 2;
      const int64_t mid = f.first + ((f.last - f.first) >> 1);
      st[sp + 1] = Frame{mid + 1, f.last, 0.0, 0};
      ++sp;
    } else {
      ret = f.v1 + ret;
      --sp;
    }
  }
  return ret;
}

// src/algorithms/proxgrad.jl:162-201 for one column per lane
template <int KP, bool TRIG>
__global__ void __launch_bounds__(64) ref_col_kernel(const RefArgs a) {
  const int64_t fl = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (fl >= a.nseg) return;
  const int k = a.k;
  const int64_t fg = a.own_offset + fl;
  const int64_t b = a.ptr[fl], e = a.ptr[fl + 1];
  double* yp = a.own + fg * KP;
  double y[KP], G[KP], yn[KP];
  load_vec<KP>(y, yp);
#pragma unroll
  for (int c = 0; c < KP; ++c) G[c] = 0.0;
  const LossDesc lo = load_loss(a.losses, a.n_losses == 1 ? 0 : fg);
  if (a.eval_only) { // glrm_cpu_col_losses: err += evaluate(...) sequentially (objective(), src/evaluate_fit.jl:13-17)
    double err = 0.0;
    for (int64_t t = b; t < e; ++t) {
      const double u = ref_dot<KP>(y, a.other + (int64_t)a.idx[t] * KP, k);
      double L, dL;
      loss_both<false, TRIG>(lo, u, a.vals[t], L, dL);
      err += L;
    }
    a.obj[fg] = err;
    return;
  }
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : fl);
  // gradient (:165-175) and the loss sum of col_objective at y (:178) in one walk
  double obj = 0.0;
  obj += ref_col_loss<KP, TRIG, true>(a, lo, b, e, y, G);
  obj += ref_reg_eval<KP>(rd, y, k);
  const double l = (double)(e - b) + 1; // :177
  double alpha = a.alpha[fl];
  int ntr = 0, nacc = 0;
  while (alpha > a.min_stepsize) { // :179
    const double stepsize = alpha / l;
#pragma unroll
    for (int c = 0; c < KP; ++c) yn[c] = fma(-stepsize, G[c], y[c]); // :183
    ref_reg_prox<KP>(rd, yn, k, stepsize);                            // :185
    double nobj = 0.0;
    nobj += ref_col_loss<KP, TRIG, false>(a, lo, b, e, yn, G);
    nobj += ref_reg_eval<KP>(rd, yn, k);
    ++ntr;
    if (nobj < obj) { // :187-191
      store_vec<KP>(yp, yn);
      alpha *= 1.05;
      obj = nobj;
      ++nacc;
      break;
    }
    alpha *= .7; // :192-199
    if (alpha < a.min_stepsize) {
      alpha = a.min_stepsize * 1.1;
      break;
    }
  }
  a.alpha[fl] = alpha;
  if (a.obj) a.obj[fg] = obj; // obj_by_col[f]
  if (a.trials) { a.trials[fl] += ntr; a.accepts[fl] += nacc; }
}

// objective() of the reference adds EVERY observation's loss into one accumulator, columns outer, list order inner
// (src/evaluate_fit.jl:12-17).  That sum is serial by definition; the terms are not: entry t of the column view gets its loss at the
// current factors (the same dot chain and loss formula as the passes above), the host adds them in order (glrm_reforder_objective).
template <int KP, bool TRIG>
__global__ void __launch_bounds__(256) ref_terms_kernel(const RefArgs a, int64_t t0, int64_t t1, double* __restrict__ out) {
  const int64_t t = t0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= t1) return;
  int64_t lo_ = 0, hi_ = a.nseg; // the column of entry t: last fl with ptr[fl] <= t
  while (hi_ - lo_ > 1) {
    const int64_t mid = lo_ + ((hi_ - lo_) >> 1);
    if (a.ptr[mid] <= t) lo_ = mid; else hi_ = mid;
  }
  const int64_t fg = a.own_offset + lo_;
  double y[KP];
  load_vec<KP>(y, a.own + fg * KP);
  const LossDesc lo = load_loss(a.losses, a.n_losses == 1 ? 0 : fg);
  const double u = ref_dot<KP>(y, a.other + (int64_t)a.idx[t] * KP, a.k);
  double L, dL;
  loss_both<false, TRIG>(lo, u, a.vals[t], L, dL);
  out[t - t0] = L;
}

template <int KP>
int launch_ref_terms(bool trig, const RefArgs& a, int64_t t0, int64_t t1, double* out, hipStream_t st) {
  const unsigned grid = (unsigned)((t1 - t0 + 255) / 256);
  if (trig) hipLaunchKernelGGL((ref_terms_kernel<KP, true>), dim3(grid), dim3(256), 0, st, a, t0, t1, out);
  else hipLaunchKernelGGL((ref_terms_kernel<KP, false>), dim3(grid), dim3(256), 0, st, a, t0, t1, out);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

template <int KP>
int launch_ref(bool rows, bool trig, const RefArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)((a.nseg + 63) / 64);
  if (rows) {
    if (trig) hipLaunchKernelGGL((ref_row_kernel<KP, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((ref_row_kernel<KP, false>), dim3(grid), dim3(64), 0, st, a);
  } else {
    if (trig) hipLaunchKernelGGL((ref_col_kernel<KP, true>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((ref_col_kernel<KP, false>), dim3(grid), dim3(64), 0, st, a);
  }
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

} // namespace

// glrm_options.sum_order = 1 is honoured by the scalar-loss list path up to rank 64
int glrm_setup_reforder(glrm_handle* h) {
  if (!h->sum_order_opt) return GLRM_OK;
  if (h->multi) return fail(GLRM_ERR_UNSUPPORTED, "glrm_options.sum_order = 1 (reference order) covers the scalar losses: this model has multi-dimensional losses or wrapped regularizers");
  if (h->dense) return fail(GLRM_ERR_UNSUPPORTED, "glrm_options.sum_order = 1 (reference order) covers list problems, not the dense hand-over");
  if (h->kp > 64) return fail(GLRM_ERR_UNSUPPORTED, "glrm_options.sum_order = 1 (reference order) is built for ranks up to 64");
  return GLRM_OK;
}

int glrm_run_reforder(glrm_handle* h, bool rows, double min_stepsize, int eval_only) {
  if (h->fixed_alpha > 0.0 && !eval_only)
    return fail(GLRM_ERR_UNSUPPORTED, "glrm_options.sum_order = 1 (reference order) restates the ProxGradParams half-steps only, not the fixed-step sweeps of SparseProxGradParams");
  if (rows && eval_only) return fail(GLRM_ERR_INVALID, "no evaluation pass over the row view");
  RefArgs a{};
  a.nseg = rows ? h->ml : h->nl;
  a.ptr = rows ? h->rowptr : h->colptr;
  a.idx = rows ? h->colidx : h->rowidx;
  a.vals = rows ? h->rowvals : h->colvals;
  a.own = rows ? h->X : h->Y;
  a.own_offset = rows ? h->rb : h->cb;
  a.other = rows ? h->Y : h->X;
  a.alpha = rows ? h->alpharow : h->alphacol;
  a.obj = rows ? nullptr : h->objcol;
  a.losses = h->losses;
  a.n_losses = h->n_losses;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.k = h->k;
  a.min_stepsize = min_stepsize;
  a.trials = eval_only ? nullptr : (rows ? h->trials_r : h->trials_c);
  a.accepts = rows ? h->accepts_r : h->accepts_c;
  a.eval_only = eval_only;
  if (rows && h->rng_e >= 0) { // glrm_hip_step_x_range: local rows [rng_b, rng_e)
    const int64_t s0 = h->rng_b;
    a.nseg = h->rng_e - s0;
    a.ptr += s0; a.alpha += s0; a.own_offset += s0;
    if (!a.reg_single) a.regs += s0;
    if (a.trials) a.trials += s0;
    a.accepts += s0;
  }
  if (a.nseg <= 0) return GLRM_OK;
  switch (h->kp) {
    case 8: return launch_ref<8>(rows, h->has_trig, a, h->stream);
    case 16: return launch_ref<16>(rows, h->has_trig, a, h->stream);
    case 32: return launch_ref<32>(rows, h->has_trig, a, h->stream);
    case 64: return launch_ref<64>(rows, h->has_trig, a, h->stream);
    default: return fail(GLRM_ERR_UNSUPPORTED, "no reference-order kernel for a padded rank of %d", h->kp);
  }
}

// sum(::Vector{Float64}) as Julia adds it (pairwise, blocks of 1024): the recorded objective sum(obj_by_col), proxgrad.jl:205
static double julia_pairwise_host(const double* v, int64_t first, int64_t last) {
  if (first == last) return v[first];
  if (last - first < 1024) {
    double s = v[first] + v[first + 1];
    for (int64_t i = first + 2; i <= last; ++i) s += v[i];
    return s;
  }
  const int64_t mid = first + ((last - first) >> 1);
  const double v1 = julia_pairwise_host(v, first, mid);
  const double v2 = julia_pairwise_host(v, mid + 1, last);
  return v1 + v2;
}

// objective(glrm, X, Y; include_regularization) as the reference adds it (src/evaluate_fit.jl:4-23, calc_penalty :91-104): ONE accumulator
// over all observations -- columns outer, list order inner -- then `err += penalty`, the penalty being ONE accumulator over rx(x_1) ..
// rx(x_m), ry(y_1) .. ry(y_n) (the oracle's full_objective).  Whole problems only (glrm_hip_fit's prologue, glrm_hip_objective): with it
// the recorded objective[0] of the reference-order mode equals the reference-order oracle's to the last bit.
int glrm_reforder_objective(glrm_handle* h, int include_reg, double* out) {
  double err = 0.0;
  if (h->nl > 0 && h->nnz_c > 0) {
    RefArgs a{};
    a.nseg = h->nl; a.ptr = h->colptr; a.idx = h->rowidx; a.vals = h->colvals;
    a.own = h->Y; a.own_offset = h->cb; a.other = h->X;
    a.losses = h->losses; a.n_losses = h->n_losses; a.k = h->k;
    const int64_t chunk = std::min<int64_t>(h->nnz_c, (int64_t)1 << 25); // 256 MB of terms at a time
    double* dterms = nullptr;
    HIPCK(hipMalloc((void**)&dterms, (size_t)chunk * 8));
    std::vector<double> host;
    try { host.resize((size_t)chunk); } catch (const std::exception&) { (void)hipFree(dterms); return fail(GLRM_ERR_OOM, "out of host memory"); }
    for (int64_t t0 = 0; t0 < h->nnz_c; t0 += chunk) {
      const int64_t t1 = std::min(h->nnz_c, t0 + chunk);
      int rc;
      switch (h->kp) {
        case 8: rc = launch_ref_terms<8>(h->has_trig, a, t0, t1, dterms, h->stream); break;
        case 16: rc = launch_ref_terms<16>(h->has_trig, a, t0, t1, dterms, h->stream); break;
        case 32: rc = launch_ref_terms<32>(h->has_trig, a, t0, t1, dterms, h->stream); break;
        case 64: rc = launch_ref_terms<64>(h->has_trig, a, t0, t1, dterms, h->stream); break;
        default: rc = fail(GLRM_ERR_UNSUPPORTED, "no reference-order kernel for a padded rank of %d", h->kp);
      }
      hipError_t e = rc ? hipSuccess : hipMemcpyAsync(host.data(), dterms, (size_t)(t1 - t0) * 8, hipMemcpyDeviceToHost, h->stream);
      if (!rc && e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (rc || e != hipSuccess) {
        (void)hipFree(dterms);
        return rc ? rc : fail(GLRM_ERR_HIP, "reference-order objective: %s", hipGetErrorString(e));
      }
      const double* v = host.data();
      for (int64_t i = 0, ne = t1 - t0; i < ne; ++i) err += v[i]; // err += evaluate(...), one accumulator (src/evaluate_fit.jl:15)
    }
    (void)hipFree(dterms);
  }
  if (include_reg) {
    double penalty = 0.0; // calc_penalty: rows then columns into one accumulator (src/evaluate_fit.jl:96-102)
    int rc;
    std::vector<double> host((size_t)std::max<int64_t>(h->m, h->n));
    if ((rc = glrm_hip_row_penalties(h))) return rc;
    HIPCK(hipMemcpyAsync(host.data(), h->objrow, (size_t)h->m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->m; ++i) penalty += host[(size_t)i];
    if ((rc = glrm_hip_col_penalties(h))) return rc;
    HIPCK(hipMemcpyAsync(host.data(), h->objcol, (size_t)h->n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    for (int64_t f = 0; f < h->n; ++f) penalty += host[(size_t)f];
    err += penalty;
  }
  *out = err;
  return GLRM_OK;
}

int glrm_reforder_sum(glrm_handle* h, const void* dvec, int64_t n, double* out) {
  if (n <= 0) { *out = 0.0; return GLRM_OK; }
  std::vector<double> host((size_t)n);
  HIPCK(hipMemcpyAsync(host.data(), dvec, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  *out = julia_pairwise_host(host.data(), 0, n - 1);
  return GLRM_OK;
}
