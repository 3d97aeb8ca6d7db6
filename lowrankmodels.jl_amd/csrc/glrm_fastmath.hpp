// glrm_fastmath.hpp -- fp64 exp / log1p / reciprocal for the loss formulas of the sweep kernels, written out so that
//   * LogisticLoss (src/losses.jl:298-311) costs ONE exponential per observation (fm_logistic below) instead of exp + log + exp + an
//     IEEE division (the ocml calls the compiler would inline);
//   * nothing is called: no table, no special-case branches -- straight-line FMA chains the scheduler can interleave with LDS reads.
// Accuracy (tools/check_fastmath.cpp, run on the host against libm over 4e6 points per function): see the numbers it prints
// (exp and log ~1e-16 relative; the logistic loss and derivative against the reference's literal double formula likewise).
// The CPU oracle keeps libm and the reference's literal formula; parity tests compare against it.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#define GLRM_FM __host__ __device__ __forceinline__
#else
#define GLRM_FM static inline
#endif

namespace glrm {

// Polynomial coefficients.  On the device they are read from constant memory with scalar loads at the point of use: as 64-bit literals the
// compiler parks all of them in VGPR pairs across the sweep loop (v_fmac_f64 wants its addend in the destination), ~50 VGPRs of a
// 128-VGPR budget -- measured: 84 more bytes of scratch per lane in the heterogeneous row kernel than with the ocml calls.
#if defined(__HIP_DEVICE_COMPILE__)
#define GLRM_FM_TABLE static __constant__
#else
#define GLRM_FM_TABLE static const
#endif
GLRM_FM_TABLE double fm_exp_c[12] = {1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
                                     2.7557319223985893e-06, 2.48015873015873e-05, 0.0001984126984126984, 0.001388888888888889,
                                     0.008333333333333333, 0.041666666666666664, 0.16666666666666666, 0.5};  // 1/13! ... 1/2!
GLRM_FM_TABLE double fm_atanh_c[10] = {1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0};

// 1 / d for finite d away from 0 and from the ends of the exponent range (here d is always in [1, 3]).
// v_rcp_f64 delivers ~26 good bits; two Newton steps square the error twice.
GLRM_FM double fm_rcp(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(d);
#else
  double y = (double)(1.0f / (float)d); // host stand-in with the same starting accuracy class (24 bits)
#endif
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  return y;
}

// exp(x).  Cody-Waite reduction x = n ln2 + r, |r| <= 0.3466, Taylor polynomial of degree 13 (remainder < 5e-18), scaling by 2^n with
// ldexp.  x is clamped to [-750, 750]: ldexp then delivers 0 / Inf by itself.  NaN in, NaN out.
GLRM_FM double fm_exp(double x) {
  const double xc = x < -750.0 ? -750.0 : (x > 750.0 ? 750.0 : x); // comparisons, not fmin / fmax: a NaN passes through
  const double n = rint(xc * 1.4426950408889634074);
  double r = fma(-n, 6.93147180369123816490e-01, xc);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = fm_exp_c[0];
#pragma unroll
  for (int i = 1; i < 12; ++i) p = fma(p, r, fm_exp_c[i]);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// log(w) for w in [1, Inf]: w = 2^e m with m in [sqrt(1/2), sqrt(2)), log(m) = 2 atanh((m - 1) / (m + 1)) with |(m - 1) / (m + 1)| <= 0.1716
// (eleven terms of the series), log(w) = e ln2 + log(m) with ln2 split in two.  Near w = 1 the result keeps its relative accuracy
// (e = 0 and m - 1 is exact).  log(Inf) = Inf, log(NaN) = NaN.
GLRM_FM double fm_log_ge1(double w) {
  int e;
  double m = frexp(w, &e);                     // m in [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;                          // [sqrt(1/2), sqrt(2))
  const double ed = (double)(lo ? e - 1 : e);
  const double f = m - 1.0;
  const double s = f * fm_rcp(2.0 + f);
  const double s2 = s * s;
  double p = fm_atanh_c[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) p = fma(p, s2, fm_atanh_c[i]);
  p = fma(p, s2 * s, s);                       // s + s^3 (1/3 + ...)
  const double r = fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, p + p));
  return w < __builtin_inf() ? r : w;          // Inf (frexp of Inf is unspecified) and NaN pass through
}

// log(x) for any x: the same reduction serves 0 < x < 1 (frexp handles denormals; the worst cancellation, e ln2 + log(m) at e = -1, costs
// under half a digit).  log(0) = -Inf, log(x < 0) = NaN, log(Inf) = Inf, log(NaN) = NaN, as IEEE / libm.
GLRM_FM double fm_log(double x) {
  const double r = fm_log_ge1(x);
  return x > 0.0 ? r : (x == 0.0 ? -__builtin_inf() : __builtin_nan(""));
}

// LogisticLoss (src/losses.jl:298-311) from ONE exponential, in the reference's own structure so that its rounding is reproduced:
// with z = (2a-1) u and E = exp(-z), w = 1 + E (rounded as in the reference),
//     evaluate = scale * log(1 + exp(-z))        = scale * log(w)            -- exactly 0 once E < 2^-53, Inf once exp overflows
//     grad     = -(2a-1) scale / (1 + exp(z))    = -(2a-1) scale * E / w     -- exp(z) = 1 / E; for E > 1 as 1 - 1 / w (no Inf * 0)
// (the line search compares sums of these with a strict `<`: a loss that is "more accurate than the reference" -- log1p(E), or a
// finite value where the reference overflows -- takes different decisions; found by tests/test_gpu_fuzz.py).
template <bool NEED_GRAD>
GLRM_FM void fm_logistic(double scale, double aa, double u, double& L, double& dL) {
  const double z = aa * u;
  const double E = fm_exp(-z);
  const double w = 1.0 + E;
  L = scale * fm_log_ge1(w);
  if (NEED_GRAD) {
    const bool fin = w < __builtin_inf();
    const double rw = fm_rcp(fin ? w : 1.0);   // the Newton steps of fm_rcp would turn 1 / Inf into NaN
    const double frac = E > 1.0 ? 1.0 - (fin ? rw : 0.0) : E * rw; // E / (1 + E) = 1 / (1 + exp(z)); NaN stays NaN
    dL = -aa * scale * frac;
  }
}

} // namespace glrm
