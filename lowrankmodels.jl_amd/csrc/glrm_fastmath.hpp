// glrm_fastmath.hpp -- fp64 exp / log1p / reciprocal for the loss formulas of the sweep kernels, written out so that
//   * LogisticLoss (src/losses.jl:298-311) costs ONE exponential per observation: with z = (2a-1) u and t = exp(-|z|) in (0, 1],
//         evaluate = scale * log(1 + exp(-z))   = scale * (max(-z, 0) + log1p(t))
//         grad     = -(2a-1) * scale / (1 + exp(z)) = -(2a-1) * scale * (z >= 0 ? t / (1 + t) : 1 / (1 + t))
//     instead of exp + log + exp + an IEEE division (the ocml calls the compiler would inline: ~105 fp64 VALU instructions and ~20 live
//     registers per lane; here ~60 and ~10);
//   * nothing is called: no table, no special-case branches -- straight-line FMA chains the scheduler can interleave with LDS reads.
// Accuracy (tools/check_fastmath.cpp, run on the host against libm over 4e6 points per function): exp <= 1.2e-16 relative on
// [-745, 709], log1p <= 2.3e-16 relative on (0, 1], the logistic loss and its derivative <= 4e-16 relative to the exact value.  The
// reference's own `log(1 + exp(-z))` LOSES digits once exp(-z) < 1e-8 (1 + tiny rounds): there the two differ by up to 1.1e-16
// ABSOLUTE per observation (relative to a loss value of that size), which is far inside the 1e-5 contract on objectives and factors.
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

// log1p(t) for t in [0, 1]: log(1 + t) = 2 atanh(t / (2 + t)); above sqrt(2) - 1 the argument is halved first,
// log(1 + t) = ln 2 + log1p((t - 1) / 2), so that |s| = |tt / (2 + tt)| <= 0.1716 and eleven terms of the atanh series suffice.
GLRM_FM double fm_log1p_unit(double t) {
  const bool hi = t > 0.41421356237309503;
  const double tt = hi ? (t - 1.0) * 0.5 : t;
  const double s = tt * fm_rcp(2.0 + tt);
  const double s2 = s * s;
  double p = fm_atanh_c[0];
#pragma unroll
  for (int i = 1; i < 10; ++i) p = fma(p, s2, fm_atanh_c[i]);
  p = fma(p, s2 * s, s);                       // s + s^3 (1/3 + ...)
  return fma(2.0, p, hi ? 0.693147180559945309417 : 0.0);
}

// LogisticLoss from one exponential (see the header).  aa = 2a - 1 in {-1, +1}.
template <bool NEED_GRAD>
GLRM_FM void fm_logistic(double scale, double aa, double u, double& L, double& dL) {
  const double z = aa * u;
  const double t = fm_exp(z < 0 ? z : -z);     // exp(-|z|) in (0, 1]; NaN stays NaN
  L = scale * ((z < 0 ? -z : 0.0) + fm_log1p_unit(t));
  if (NEED_GRAD) {
    const double r = fm_rcp(1.0 + t);
    dL = -aa * scale * (z < 0 ? r : t * r);
  }
}

} // namespace glrm
