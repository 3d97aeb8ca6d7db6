// glrm_device.hpp -- device-side operators of the GLRM sweep kernels (gfx950, wave64).
//
// Scalar losses and regularizers restate LowRankModels.jl's formulas
// (src/losses.jl:138-352, src/regularizers.jl:52-114,295-318) in fp64.
// The translation unit is built with -ffp-contract=off: every fused multiply-add below is an
// explicit fma(), so the per-entry arithmetic is the same expression tree as the CPU oracle's
// (oracle/glrm_oracle.c); only summation ORDER differs between the two.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/glrm_hip.h"
#include "glrm_fastmath.hpp"

namespace glrm {

// ---------------------------------------------------------------- cross-lane (DPP) helpers
// 64-bit values move as two 32-bit DPP movs; the controls used here only ever read lanes of
// the same G-lane group, and a group is always entirely active or entirely inactive.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false);
}

constexpr int DPP_XOR1 = 0xB1;         // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm:[2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i inside each 8 lanes
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15-i inside each 16 lanes

// Butterfly all-reduce over the G consecutive lanes of a group (G = 1, 2, 4, 8 or 16).  Every lane of
// the group ends with the bit-identical sum (each step adds two values that are already equal
// across the sub-group, and + commutes).
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (G >= 2) v += dpp_f64<DPP_XOR1>(v);
  if constexpr (G >= 4) v += dpp_f64<DPP_XOR2>(v);
  if constexpr (G >= 8) v += dpp_f64<DPP_HALF_MIRROR>(v);
  if constexpr (G >= 16) v += dpp_f64<DPP_MIRROR>(v);
  return v;
}

// value of lane u of the caller's group, in every lane of the group
template <int G>
__device__ __forceinline__ int group_bcast_i32(int v, int u, int lane) {
  if constexpr (G == 1) {
    return v;
  } else if constexpr (G == 2) {
    return u == 0 ? __builtin_amdgcn_mov_dpp(v, 0xA0, 0xF, 0xF, false)  // quad_perm:[0,0,2,2]
                  : __builtin_amdgcn_mov_dpp(v, 0xF5, 0xF, 0xF, false); // quad_perm:[1,1,3,3]
  } else if constexpr (G == 4) {
    switch (u) { // compile-time after unrolling
      case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, false);
      case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, false);
      case 2: return __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, false);
      default: return __builtin_amdgcn_mov_dpp(v, 0xFF, 0xF, 0xF, false);
    }
  } else {
    return __shfl(v, (lane & ~(G - 1)) + u, 64);
  }
}
template <int G>
__device__ __forceinline__ double group_bcast_f64(double v, int u, int lane) {
  const int lo = group_bcast_i32<G>(__double2loint(v), u, lane), hi = group_bcast_i32<G>(__double2hiint(v), u, lane);
  return __hiloint2double(hi, lo);
}

// All-reduce over the 64/G groups of a wave for values that are replicated inside each group
// (xor-butterfly on the group-index bits; ds_bpermute, only used once per pass).
template <int G>
__device__ __forceinline__ double across_groups_sum(double v) {
#pragma unroll
  for (int d = G; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---------------------------------------------------------------- losses
struct LossDesc {
  int kind;
  double scale, p0, p1;
};

__device__ __forceinline__ LossDesc load_loss(const glrm_loss* t, int64_t i) {
  LossDesc d;
  d.kind = t[i].kind;
  d.scale = t[i].scale;
  d.p0 = t[i].p0;
  d.p1 = t[i].p1;
  return d;
}

__device__ __forceinline__ double sign_jl(double d) { return d > 0 ? 1.0 : (d < 0 ? -1.0 : d); }

// evaluate(l,u,a) and grad(l,u,a) in one go (the gradient pass needs both).
// TRIG = false compiles the PeriodicLoss case out (NaN): see LOSS_*_NOTRIG in glrm_engine.hpp.
template <bool NEED_GRAD, bool TRIG = true>
__device__ __forceinline__ void loss_both(const LossDesc& l, double u, double a, double& L, double& dL) {
  const double s = l.scale;
  dL = 0.0;
  switch (l.kind) {
    case GLRM_LOSS_QUAD: { // src/losses.jl:144,146
      const double d = u - a;
      L = s * (d * d);
      if (NEED_GRAD) dL = 2 * d * s;
      break;
    }
    case GLRM_LOSS_L1: { // :158,160
      const double d = u - a;
      L = s * fabs(d);
      if (NEED_GRAD) dL = sign_jl(d) * s;
      break;
    }
    case GLRM_LOSS_HUBER: { // :173-177
      const double c = l.p0, d = u - a, ad = fabs(d);
      L = ad > c ? (ad - c + c * c) * s : (d * d) * s;
      if (NEED_GRAD) dL = ad > c ? sign_jl(d) * s : d * s;
      break;
    }
    case GLRM_LOSS_QUANTILE: { // :193-201
      const double q = l.p0, diff = a - u;
      L = diff > 0 ? s * q * diff : -s * (1 - q) * diff;
      if (NEED_GRAD) dL = diff > 0 ? -s * q : s * (1 - q);
      break;
    }
    case GLRM_LOSS_PERIODIC: { // :216,218
      if constexpr (TRIG) {
        const double T = l.p0, w = (a - u) * (2 * M_PI) / T;
        L = s * (1 - cos(w));
        if (NEED_GRAD) dL = -s * ((2 * M_PI) / T) * sin(w);
      } else {
        L = __builtin_nan("");
        dL = L;
      }
      break;
    }
    case GLRM_LOSS_POISSON: { // :237-241
#if defined(GLRM_POISSON_LIBM)
      const double eu = exp(u);
#else
      const double eu = fm_exp(u);
#endif
      L = s * (eu - a * u + (a == 0 ? 0.0 : a * (log(a) - 1)));
      if (NEED_GRAD) dL = s * (eu - a);
      break;
    }
    case GLRM_LOSS_ORDINAL_HINGE: { // :258-292.  The four branches of evaluate share one expression, n (n + 1) / 2 + (n + 1) f, and differ
      // in n and f only: the operands are selected, the expression is evaluated once -- the same operations on the same values as the
      // literal transcription (oracle/glrm_oracle.c), without four divergent code paths.
      const double mn = l.p0, mx = l.p1;
#if defined(GLRM_ORDINAL_BRANCHY)
      double n, loss;
      if (u > mx - 1) {
        n = fmin(floor(u), mx - 1) - a;
        loss = n * (n + 1) / 2 + (n + 1) * (u - mx + 1);
      } else if (u > a) {
        n = fmin(floor(u), mx) - a;
        loss = n * (n + 1) / 2 + (n + 1) * (u - floor(u));
      } else if (u > mn + 1) {
        n = a - fmax(ceil(u), mn + 1);
        loss = n * (n + 1) / 2 + (n + 1) * (ceil(u) - u);
      } else {
        n = a - fmax(ceil(u), mn + 1);
        loss = n * (n + 1) / 2 + (n + 1) * (mn + 1 - u);
      }
      L = s * loss;
      if (NEED_GRAD) {
        const double g = u > a ? fmin(ceil(u), mx) - a : -(a - fmax(floor(u), mn));
        dL = s * g;
      }
#else
      const double fl = floor(u), ce = ceil(u);
      const bool top = u > mx - 1, up = top || u > a, mid = u > mn + 1;
      const double n_up = fmin(fl, top ? mx - 1 : mx) - a, f_up = top ? u - mx + 1 : u - fl;
      const double n_dn = a - fmax(ce, mn + 1), f_dn = mid ? ce - u : mn + 1 - u;
      const double n = up ? n_up : n_dn, f = up ? f_up : f_dn;
      L = s * (n * (n + 1) / 2 + (n + 1) * f);
      if (NEED_GRAD) {
        const double g = u > a ? fmin(ce, mx) - a : -(a - fmax(fl, mn));
        dL = s * g;
      }
#endif
      break;
    }
    case GLRM_LOSS_LOGISTIC: { // :304,306 ; a is 1.0 (true) / 0.0 (false).  One exponential per observation: glrm_fastmath.hpp
#if defined(GLRM_LOGISTIC_LIBM)
      const double aa = 2 * a - 1;
      L = s * log(1 + exp(-aa * u));
      if (NEED_GRAD) dL = -aa * s / (1 + exp(aa * u));
#else
      fm_logistic<NEED_GRAD>(s, 2 * a - 1, u, L, dL);
#endif
      break;
    }
    case GLRM_LOSS_WEIGHTED_HINGE: { // :326-341
      const double r = l.p0, aa = 2 * a - 1;
      double loss = s * fmax(1 - aa * u, 0.0);
      double g = (aa * u >= 1 ? 0.0 : -aa * s);
      if (r != 1.0 && a == 1.0) {
        loss *= r;
        g *= r;
      }
      L = loss;
      if (NEED_GRAD) dL = g;
      break;
    }
    default:
      L = __builtin_nan("");
      dL = L;
  }
}

// ---------------------------------------------------------------- regularizers
struct RegDesc {
  int kind;
  double scale;
};

__device__ __forceinline__ RegDesc load_reg(const glrm_reg* t, int64_t i) {
  RegDesc d;
  d.kind = t[i].kind;
  d.scale = t[i].scale;
  return d;
}

// Lane layout of a k-vector inside a G-lane group (leading dimension KP = G*R, zero padded):
// lane j of the group holds R/2 double2 slices, slice i = components [i*2G + 2j, i*2G + 2j + 2).
// With that layout one 16-byte load per lane makes the G lanes of a group read 16*G
// contiguous bytes of the factor column.
template <int G, int R>
struct Vec {
  double2 v[R / 2];
};

template <int G, int R>
__device__ __forceinline__ int comp_index(int i, int j, int h) { return i * 2 * G + 2 * j + h; }

// prox(r, u, alpha) on the distributed vector (src/regularizers.jl:34,56,83-86,93,103,297);
// components >= k (padding) are forced to 0.
template <int G, int R>
__device__ __forceinline__ void reg_prox(const RegDesc& r, Vec<G, R>& u, double alpha, int j, int k) {
  switch (r.kind) {
    case GLRM_REG_QUAD: {
      const double f = 1 / (1 + 2 * alpha * r.scale);
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        u.v[i].x = f * u.v[i].x;
        u.v[i].y = f * u.v[i].y;
      }
      break;
    }
    case GLRM_REG_ONE: {
      const double t = r.scale * alpha;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        u.v[i].x = fmax(u.v[i].x - t, 0.0) + fmin(u.v[i].x + t, 0.0);
        u.v[i].y = fmax(u.v[i].y - t, 0.0) + fmin(u.v[i].y + t, 0.0);
      }
      break;
    }
    case GLRM_REG_NONNEG: {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        u.v[i].x = u.v[i].x > 0 ? u.v[i].x : 0.0;
        u.v[i].y = u.v[i].y > 0 ? u.v[i].y : 0.0;
      }
      break;
    }
    case GLRM_REG_UNIT_ONE_SPARSE: { // e_{argmax u}, first maximal index
      double best = -__builtin_inf();
      int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        const int c0 = comp_index<G, R>(i, j, 0);
        if (c0 < k && (u.v[i].x > best || bi == 0x7fffffff)) { best = u.v[i].x; bi = c0; }
        if (c0 + 1 < k && (u.v[i].y > best || bi == 0x7fffffff)) { best = u.v[i].y; bi = c0 + 1; }
      }
      auto merge = [&](double ob, int oi) {
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
      };
      if constexpr (G >= 2) merge(dpp_f64<DPP_XOR1>(best), dpp_i32<DPP_XOR1>(bi));
      if constexpr (G >= 4) merge(dpp_f64<DPP_XOR2>(best), dpp_i32<DPP_XOR2>(bi));
      if constexpr (G >= 8) merge(dpp_f64<DPP_HALF_MIRROR>(best), dpp_i32<DPP_HALF_MIRROR>(bi));
      if constexpr (G >= 16) merge(dpp_f64<DPP_MIRROR>(best), dpp_i32<DPP_MIRROR>(bi));
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        const int c0 = comp_index<G, R>(i, j, 0);
        u.v[i].x = (c0 == bi) ? 1.0 : 0.0;
        u.v[i].y = (c0 + 1 == bi) ? 1.0 : 0.0;
      }
      break;
    }
    default: // GLRM_REG_ZERO
      break;
  }
#pragma unroll
  for (int i = 0; i < R / 2; ++i) { // keep the padding exactly zero
    const int c0 = comp_index<G, R>(i, j, 0);
    if (c0 >= k) u.v[i].x = 0.0;
    if (c0 + 1 >= k) u.v[i].y = 0.0;
  }
}

// evaluate(r, x) on the distributed vector; every lane of the group returns the same value
// (src/regularizers.jl:58,88,95,103-112,300-316).
template <int G, int R>
__device__ __forceinline__ double reg_eval(const RegDesc& r, const Vec<G, R>& x, int j, int k) {
  switch (r.kind) {
    case GLRM_REG_QUAD: {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        s = fma(x.v[i].x, x.v[i].x, s);
        s = fma(x.v[i].y, x.v[i].y, s);
      }
      return r.scale * group_sum<G>(s);
    }
    case GLRM_REG_ONE: {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) s += fabs(x.v[i].x) + fabs(x.v[i].y);
      return r.scale * group_sum<G>(s);
    }
    case GLRM_REG_NONNEG: {
      double neg = 0.0;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) neg += (x.v[i].x < 0 ? 1.0 : 0.0) + (x.v[i].y < 0 ? 1.0 : 0.0);
      return group_sum<G>(neg) > 0 ? __builtin_inf() : 0.0;
    }
    case GLRM_REG_UNIT_ONE_SPARSE: {
      double code = 0.0; // ones + 4096 * (entries that are neither 0 nor 1)
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        const int c0 = comp_index<G, R>(i, j, 0);
        const double a = x.v[i].x, b = x.v[i].y;
        if (c0 < k) code += (a == 0 ? 0.0 : (a == 1 ? 1.0 : 4096.0));
        if (c0 + 1 < k) code += (b == 0 ? 0.0 : (b == 1 ? 1.0 : 4096.0));
      }
      code = group_sum<G>(code);
      return (code >= 4096.0 || code > 1.0) ? __builtin_inf() : 0.0;
    }
    default:
      return 0.0;
  }
}

} // namespace glrm
