// glrm_impute.hpp -- impute(domain, loss, u) and error_metric(domain, loss, u, a) of src/impute_and_err.jl:24-130, shared by the
// HIP kernels (glrm_impute.hip).  u points to the d = embedding_dim values of (X'Y)[i, yidxs[j]] with element stride `us`.
#pragma once

#include "glrm_device.hpp"

namespace glrm {

__host__ __device__ inline double roundcutoff(double x, double a, double b) { // T(min(max(round(x),a),b)), round half to even
  const double r = rint(x);
  return fmin(fmax(r, a), b);
}

__device__ inline bool is_diff_loss(int kind) {
  return kind == GLRM_LOSS_QUAD || kind == GLRM_LOSS_L1 || kind == GLRM_LOSS_HUBER || kind == GLRM_LOSS_QUANTILE || kind == GLRM_LOSS_PERIODIC;
}

// evaluate(l, u::Vector, level) for the generic ordinal imputation (:98-100): only the losses of include/glrm_hip.h
__device__ inline double vloss_eval_strided(const LossDesc& l, const double* u, int us, int d, int a) {
  const double s = l.scale;
  if (l.kind == GLRM_LOSS_OVA || l.kind == GLRM_LOSS_BVS) {
    LossDesc b;
    b.kind = (int)l.p1; b.scale = l.p0; b.p0 = 1.0; b.p1 = 0.0;
    double loss = 0.0, L, dL;
    for (int j = 0; j < d; ++j) {
      const bool truth = l.kind == GLRM_LOSS_OVA ? a == j : a > j;
      loss_both<false>(b, u[j * us], truth ? 1.0 : 0.0, L, dL);
      loss += L;
    }
    return s * loss;
  }
  if (l.kind == GLRM_LOSS_MULTINOMIAL) {
    double mx = u[0];
    for (int j = 1; j < d; ++j) mx = u[j * us] > mx ? u[j * us] : mx;
    const double ua = u[a * us], M = mx - ua;
    double sumexp = 0.0;
    for (int j = 0; j < d; ++j) sumexp += exp(u[j * us] - ua - M);
    return s * (log(sumexp) + M);
  }
  return __builtin_nan("");
}

// impute(D, l, u); *bad is set for pairs the reference rejects
__device__ inline double impute_value(const glrm_domain& D, const LossDesc& l, int d, const double* u, int us, int* bad) {
  const int kind = l.kind;
  int dk = D.kind;
  double lo = D.lo, hi = D.hi;
  if (dk == GLRM_DOMAIN_COUNT) { dk = GLRM_DOMAIN_ORDINAL; lo = 0.0; }          // impute(OrdinalDomain(0, max_count), l, u) :124
  if (d > 1) {
    if (dk == GLRM_DOMAIN_CATEGORICAL && (kind == GLRM_LOSS_MULTINOMIAL || kind == GLRM_LOSS_OVA)) { // argmax(u) :106-107
      int best = 0;
      for (int j = 1; j < d; ++j) if (u[j * us] > u[best * us]) best = j;
      return (double)(best + 1);
    }
    if (dk == GLRM_DOMAIN_ORDINAL && kind == GLRM_LOSS_ORDISTIC) {                 // argmin(u.^2) :82
      int best = 0;
      for (int j = 1; j < d; ++j) if (u[j * us] * u[j * us] < u[best * us] * u[best * us]) best = j;
      return (double)(best + 1);
    }
    if (dk == GLRM_DOMAIN_ORDINAL && kind == GLRM_LOSS_MULTINOMIAL_ORDINAL) {     // :91-96: p = [1-eu[1], -diff(eu)..., eu[end]]
      const double TOL = 1e-3;
      double prev = u[0] < -TOL ? u[0] : -TOL;
      double eprev = exp(prev);
      double bestp = 1.0 - eprev;
      int best = 0;
      for (int j = 1; j < d; ++j) {
        const double cur = u[j * us] < prev - TOL ? u[j * us] : prev - TOL;
        const double ecur = exp(cur);
        const double p = -(ecur - eprev);
        if (p > bestp) { bestp = p; best = j; }
        prev = cur;
        eprev = ecur;
      }
      if (eprev > bestp) best = d;
      return (double)(best + 1);
    }
    if (dk == GLRM_DOMAIN_ORDINAL) {                                                // generic :98-100: argmin over the levels
      if ((kind == GLRM_LOSS_MULTINOMIAL && (lo < 1 || hi > d)) || !(kind == GLRM_LOSS_MULTINOMIAL || kind == GLRM_LOSS_OVA || kind == GLRM_LOSS_BVS)) {
        *bad = 1; // u[a] out of bounds / no evaluate method: the reference throws
        return 0.0;
      }
      int best = -1;
      double bl = 0.0;
      for (int lev = (int)lo; lev <= (int)hi; ++lev) {
        const double v = vloss_eval_strided(l, u, us, d, lev - 1);
        if (best < 0 || v < bl) { best = lev; bl = v; } // first minimum (NaN never wins after the first)
      }
      return (double)best;
    }
    *bad = 1;
    return 0.0;
  }
  const double u0 = u[0];
  switch (dk) {
    case GLRM_DOMAIN_REAL:
    case GLRM_DOMAIN_PERIODIC:                                                     // impute(RealDomain(), l, u) :113
      if (is_diff_loss(kind)) return u0;
      if (kind == GLRM_LOSS_POISSON) return exp(u0);
      if (kind == GLRM_LOSS_ORDINAL_HINGE) return roundcutoff(u0, l.p0, l.p1);
      if (kind == GLRM_LOSS_WEIGHTED_HINGE) return 1 / u0;
      *bad = 1;                                                                     // LogisticLoss: error(...) :42
      return 0.0;
    case GLRM_DOMAIN_BOOL: {
      if (kind == GLRM_LOSS_LOGISTIC || kind == GLRM_LOSS_WEIGHTED_HINGE) return u0 >= 0 ? 1.0 : 0.0; // :57
      double Lf, Lt, dL;                                                            // :60 evaluate(l,u,false) < evaluate(l,u,true) ? false : true
      loss_both<false>(l, u0, 0.0, Lf, dL);
      loss_both<false>(l, u0, 1.0, Lt, dL);
      return Lf < Lt ? 0.0 : 1.0;
    }
    case GLRM_DOMAIN_ORDINAL:
      if (is_diff_loss(kind) || kind == GLRM_LOSS_ORDINAL_HINGE) return roundcutoff(u0, lo, hi);  // :72,:74
      if (kind == GLRM_LOSS_POISSON) return roundcutoff(exp(u0), lo, hi);                          // :73
      if (kind == GLRM_LOSS_LOGISTIC) return u0 > 0 ? hi : lo;                                      // :75
      if (kind == GLRM_LOSS_WEIGHTED_HINGE) return roundcutoff(u0 > 0 ? ceil(1 / u0) : floor(1 / u0), lo, hi); // :76-80
      *bad = 1;
      return 0.0;
    default:
      *bad = 1;
      return 0.0;
  }
}

__host__ __device__ inline double pos_mod(double T, double x) { return x > 0 ? fmod(x, T) : fmod(x, T) + T; } // :116

// error_metric(D, l, u, a) given the imputed value
__device__ inline double entry_error(const glrm_domain& D, double a_imputed, double a) {
  switch (D.kind) {
    case GLRM_DOMAIN_BOOL:
    case GLRM_DOMAIN_CATEGORICAL: return a_imputed == a ? 0.0 : 1.0;              // misclassification :33
    case GLRM_DOMAIN_PERIODIC: { const double d = pos_mod(D.lo, a_imputed) - pos_mod(D.lo, a); return d * d; } // :117-121
    default: { const double d = a_imputed - a; return d * d; }                     // squared_error :32
  }
}

} // namespace glrm
