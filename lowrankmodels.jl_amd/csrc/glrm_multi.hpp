// glrm_multi.hpp -- general sweeps: multi-dimensional losses, block regularizers, offsets.
//
// A column with a multi-dimensional loss (MultinomialLoss, OvALoss, BvSLoss, OrdisticLoss, MultinomialOrdinalLoss;
// src/losses.jl:360-620) owns d = embedding_dim columns of Y (get_yidxs, src/losses.jl:76-93); the per-observation
// gradient is a d-vector and the two half-steps become the gemm! branches of src/algorithms/proxgrad.jl:126-131 and
// :169-174.  The wrappers lastentry1 / lastentry_unpenalized (add_offset!, src/modify_glrm.jl:20-25) and the block
// regularizers OrdinalReg / MNLOrdinalReg (src/regularizers.jl:163-189,356-411) act on k-vectors / k x d blocks.
//
// This family is the engine's general path: one workgroup per segment (a row of X, or the k x d block of Y of one
// column), the segment's block, its trial point and its gradient live in LDS, the NW waves of the workgroup take
// the segment's observations round-robin, and the whole backtracking line search of the segment runs inside the
// kernel (proxgrad.jl:137-155 / :180-200).  The scalar fast paths (gather / LDS-tiled / dense MFMA sweeps) are
// untouched: a problem is routed here only when some loss has dim > 1 or some regularizer carries a wrap flag.
//
// Reductions have a fixed order (lane-sequential dots, wave partials added in wave order), so results do not
// depend on the launch configuration or on how segments are sharded.
#pragma once

#include "glrm_device.hpp"

#ifndef GLRM_MULTI_PF
#define GLRM_MULTI_PF 0 // software-pipeline depth of the row sweep with register-resident blocks (multi_pass); measured: 0 is fastest (the extra registers spill)
#endif

namespace glrm {

struct MultiArgs {
  int64_t nseg;
  const int64_t* ptr;    // segment -> observation range (local segment index)
  const int32_t* idx;    // rows: global column of the observation; columns: global row
  const double* vals;
  double* own;           // rows: X, columns: Y (global arrays, ld kp)
  int64_t own_offset;    // global index of local segment 0 (row number / column number)
  const double* other;   // rows: Y, columns: X
  const int64_t* ystart; // n+1: column f owns vectors [ystart[f], ystart[f+1]) of Y
  const glrm_loss* losses;
  int loss_single;
  const glrm_reg* regs;  // local segment index
  int reg_single;
  double* alpha;         // local
  double* obj;           // columns: objcol (global column index); rows: nullptr
  int k, kp, dmax;
  int lgP;               // log2 of the lanes per observation slot: 2^lgP >= max(k, dmax, 4)
  int mode;              // 0 = line-search step, 1 = losses only (columns), 2 = fixed step (SparseProxGradParams)
  double fixed_alpha, min_stepsize;
  int32_t* trials;
  int32_t* accepts;
};

struct PenaltyArgs {
  int64_t nseg;
  const double* own;
  int64_t own_offset;
  const int64_t* ystart; // nullptr for rows (one vector per segment)
  const glrm_reg* regs;
  int reg_single;
  int k, kp;
  double* out;           // global segment index
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum over all threads of the workgroup; every thread returns the same value.  Wave butterfly, then the wave
// partials in wave order.
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  if constexpr (NW == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += red[w];
  return t;
}

// ---------------------------------------------------------------- multi-dimensional losses (u lives in LDS)

__device__ __forceinline__ LossDesc bin_loss_of(const LossDesc& l) { // bin_loss of OvALoss / BvSLoss
  LossDesc b;
  b.kind = (int)l.p1;
  b.scale = l.p0;
  b.p0 = 1.0; // HingeLoss = WeightedHingeLoss with case_weight_ratio 1
  b.p1 = 0.0;
  return b;
}

__device__ __forceinline__ void enforce_mnl_ord_rules(double* u, int d) { // src/losses.jl:572-578
  const double TOL = 1e-3;
  u[0] = u[0] < -TOL ? u[0] : -TOL;
  for (int j = 1; j < d; ++j) u[j] = u[j] < u[j - 1] - TOL ? u[j] : u[j - 1] - TOL;
}

// evaluate(l, u::Vector, a) with a = level - 1
__device__ inline double vloss_eval(const LossDesc& l, const double* u, int d, int a) {
  const double s = l.scale;
  switch (l.kind) {
    case GLRM_LOSS_MULTINOMIAL: { // :377-388
      double mx = u[0];
      for (int j = 1; j < d; ++j) mx = u[j] > mx ? u[j] : mx;
      const double ua = u[a], M = mx - ua;
      double sumexp = 0.0;
      for (int j = 0; j < d; ++j) sumexp += exp(u[j] - ua - M);
      return s * (log(sumexp) + M);
    }
    case GLRM_LOSS_OVA: { // :424-430
      const LossDesc b = bin_loss_of(l);
      double loss = 0.0, L, dL;
      for (int j = 0; j < d; ++j) { loss_both<false, false>(b, u[j], a == j ? 1.0 : 0.0, L, dL); loss += L; }
      return s * loss;
    }
    case GLRM_LOSS_BVS: { // :461-467
      const LossDesc b = bin_loss_of(l);
      double loss = 0.0, L, dL;
      for (int j = 0; j < d; ++j) { loss_both<false, false>(b, u[j], a > j ? 1.0 : 0.0, L, dL); loss += L; }
      return s * loss;
    }
    case GLRM_LOSS_ORDISTIC: { // :499-505
      const double ua2 = u[a] * u[a];
      double M = -__builtin_inf(), invlik = 0.0;
      for (int j = 0; j < d; ++j) { const double q = ua2 - u[j] * u[j]; M = q > M ? q : M; }
      for (int j = 0; j < d; ++j) invlik += exp((ua2 - u[j] * u[j]) - M);
      return s * (M + log(invlik));
    }
    default: { // GLRM_LOSS_MULTINOMIAL_ORDINAL :581-590 (u already passed through enforce_mnl_ord_rules)
      if (a == 0) return -s * log(1.0 - exp(u[0]));
      if (a == d) return -s * u[a - 1];
      return -s * log(exp(u[a - 1]) - exp(u[a]));
    }
  }
}

// component j of grad(l, u::Vector, a)
__device__ inline double vloss_grad(const LossDesc& l, const double* u, int d, int a, int j) {
  const double s = l.scale;
  switch (l.kind) {
    case GLRM_LOSS_MULTINOMIAL: { // :390-406
      double mx = u[0];
      for (int jp = 1; jp < d; ++jp) mx = u[jp] > mx ? u[jp] : mx;
      const double uj = u[j], M = mx - uj;
      double sumexp = 0.0;
      for (int jp = 0; jp < d; ++jp) sumexp += exp(u[jp] - uj - M);
      double g = j == a ? -1.0 : 0.0;
      g += exp(-M) / sumexp;
      return s * g;
    }
    case GLRM_LOSS_OVA: {
      double L, dL;
      loss_both<true, false>(bin_loss_of(l), u[j], a == j ? 1.0 : 0.0, L, dL);
      return s * dL;
    }
    case GLRM_LOSS_BVS: {
      double L, dL;
      loss_both<true, false>(bin_loss_of(l), u[j], a > j ? 1.0 : 0.0, L, dL);
      return s * dL;
    }
    case GLRM_LOSS_ORDISTIC: { // :507-519
      const double uj = u[j], uj2 = uj * uj;
      double M = -__builtin_inf(), invlik = 0.0;
      for (int jp = 0; jp < d; ++jp) { const double q = uj2 - u[jp] * u[jp]; M = q > M ? q : M; }
      for (int jp = 0; jp < d; ++jp) invlik += exp((uj2 - u[jp] * u[jp]) - M);
      double g = j == a ? 2 * u[a] : 0.0;
      g -= 2 * uj * exp(-M) / invlik;
      return s * g;
    }
    default: { // GLRM_LOSS_MULTINOMIAL_ORDINAL :592-609
      double g = 0.0;
      if (a == 0) {
        if (j == 0) g = -exp(u[0]) / (1.0 - exp(u[0]));
      } else if (a == d) {
        if (j == a - 1) g = 1.0;
      } else {
        const double den = exp(u[a - 1]) - exp(u[a]);
        if (j == a) g = -exp(u[a]) / den;
        else if (j == a - 1) g = exp(u[a - 1]) / den;
      }
      return -s * g;
    }
  }
}

// ---------------------------------------------------------------- block regularizers (block in LDS, stride S)

// evaluate(r, block): every thread returns the same value.
template <int NW>
__device__ inline double block_reg_eval(const double* blk, int S, int k, int DO, const glrm_reg rg, double* red) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x;
  const int kr = rg.wrap ? k - 1 : k;                                                        // rows the base regularizer sees
  const int jc = (rg.wrap & (GLRM_WRAP_ORDINAL | GLRM_WRAP_MNL_ORDINAL)) ? 1 : DO;          // evaluate(r.r, a[1:end-1, 1])
  double bad = 0.0, v = 0.0;
  if (rg.wrap == GLRM_WRAP_LASTENTRY1)
    for (int j = tid; j < DO; j += NT) bad += blk[j * S + k - 1] != 1.0 ? 1.0 : 0.0;
  for (int i = tid; i < kr * jc; i += NT) {
    const int j = i / kr, c = i - j * kr;
    const double x = blk[j * S + c];
    switch (rg.kind) {
      case GLRM_REG_QUAD: v = fma(x, x, v); break;
      case GLRM_REG_ONE: v += fabs(x); break;
      case GLRM_REG_NONNEG: v += x < 0 ? 1.0 : 0.0; break;
      case GLRM_REG_UNIT_ONE_SPARSE: v += x == 0 ? 0.0 : (x == 1 ? 1.0 : 4096.0); break;
      default: break;
    }
  }
  bad = block_sum<NW>(bad, red);
  v = block_sum<NW>(v, red);
  if (bad > 0) return __builtin_inf();
  switch (rg.kind) {
    case GLRM_REG_QUAD:
    case GLRM_REG_ONE: return rg.scale * v;
    case GLRM_REG_NONNEG: return v > 0 ? __builtin_inf() : 0.0;
    case GLRM_REG_UNIT_ONE_SPARSE: return (v >= 4096.0 || v > 1.0) ? __builtin_inf() : 0.0;
    default: return 0.0;
  }
}

// prox of the base regularizer on rows [0, kr) of DO columns.  whole: UnitOneSparse picks one entry of the whole
// sub-block (column-major first maximum), otherwise one per column.
template <int NW>
__device__ inline void base_prox_region(double* blk, int S, int kr, int DO, const glrm_reg rg, double alpha, bool whole) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x;
  switch (rg.kind) {
    case GLRM_REG_QUAD: {
      const double f = 1 / (1 + 2 * alpha * rg.scale);
      for (int i = tid; i < kr * DO; i += NT) { const int j = i / kr, c = i - j * kr; blk[j * S + c] = f * blk[j * S + c]; }
      break;
    }
    case GLRM_REG_ONE: {
      const double t = rg.scale * alpha;
      for (int i = tid; i < kr * DO; i += NT) {
        const int j = i / kr, c = i - j * kr;
        const double x = blk[j * S + c];
        blk[j * S + c] = fmax(x - t, 0.0) + fmin(x + t, 0.0);
      }
      break;
    }
    case GLRM_REG_NONNEG:
      for (int i = tid; i < kr * DO; i += NT) { const int j = i / kr, c = i - j * kr; const double x = blk[j * S + c]; blk[j * S + c] = x > 0 ? x : 0.0; }
      break;
    case GLRM_REG_UNIT_ONE_SPARSE:
      if (tid == 0 && kr > 0) {
        if (whole) {
          int bj = 0, bc = 0;
          for (int j = 0; j < DO; ++j)
            for (int c = 0; c < kr; ++c)
              if (blk[j * S + c] > blk[bj * S + bc]) { bj = j; bc = c; }
          for (int j = 0; j < DO; ++j)
            for (int c = 0; c < kr; ++c) blk[j * S + c] = 0.0;
          blk[bj * S + bc] = 1.0;
        } else {
          for (int j = 0; j < DO; ++j) {
            int bc = 0;
            for (int c = 1; c < kr; ++c) if (blk[j * S + c] > blk[j * S + bc]) bc = c;
            for (int c = 0; c < kr; ++c) blk[j * S + c] = c == bc ? 1.0 : 0.0;
          }
        }
      }
      break;
    default: break;
  }
}

// prox!(r, block, alpha) (src/regularizers.jl:34-114,163-189,295-318,356-405); ends with a workgroup barrier.
template <int NW>
__device__ inline void block_prox(double* blk, int S, int k, int DO, const glrm_reg rg, double alpha, double* tmp) {
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x;
  const int kr = rg.wrap ? k - 1 : k;
  if (rg.wrap & (GLRM_WRAP_ORDINAL | GLRM_WRAP_MNL_ORDINAL)) {
    if (tid < kr) { // um = mean(u[1:end-1, :], dims=2)
      double acc = 0.0;
      for (int j = 0; j < DO; ++j) acc += blk[j * S + tid];
      tmp[tid] = acc / DO;
    }
    __syncthreads();
    base_prox_region<NW>(tmp, 0, kr, 1, rg, alpha, false);
    __syncthreads();
    for (int i = tid; i < kr * DO; i += NT) { const int j = i / kr, c = i - j * kr; blk[j * S + c] = tmp[c]; }
    if ((rg.wrap & GLRM_WRAP_MNL_ORDINAL) && tid == 0) { // decreasing, negative last row (not exactly the prox, :400-404)
      const double TOL = 1e-3;
      double* last = blk + (k - 1);
      last[0] = last[0] < -TOL ? last[0] : -TOL;
      for (int j = 1; j < DO; ++j) last[j * S] = last[j * S] < last[(j - 1) * S] - TOL ? last[j * S] : last[(j - 1) * S] - TOL;
    }
    __syncthreads();
    return;
  }
  base_prox_region<NW>(blk, S, kr, DO, rg, alpha, !(rg.wrap == GLRM_WRAP_LASTENTRY1 || DO == 1));
  if (rg.wrap == GLRM_WRAP_LASTENTRY1)
    for (int j = tid; j < DO; j += NT) blk[j * S + k - 1] = 1.0;
  __syncthreads();
}

// ---------------------------------------------------------------- slot reductions
// All-reduce over the P = 2^lg consecutive lanes of an observation slot (P >= 4).  DPP inside a 16-lane row, ds_bpermute
// across rows.  Every lane of the slot gets the same value; a slot is always entirely active or entirely inactive.
struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return a > b ? a : b; } };

template <class Op>
__device__ __forceinline__ double slot_reduce(double v, int lg, Op op) {
  v = op(v, dpp_f64<DPP_XOR1>(v));
  v = op(v, dpp_f64<DPP_XOR2>(v));
  if (lg >= 3) v = op(v, dpp_f64<DPP_HALF_MIRROR>(v));
  if (lg >= 4) v = op(v, dpp_f64<DPP_MIRROR>(v));
  if (lg >= 5) v = op(v, __shfl_xor(v, 16, 64));
  if (lg >= 6) v = op(v, __shfl_xor(v, 32, 64));
  return v;
}

// Inclusive prefix over the lanes 0 .. sub of a slot.  Slots of <= 16 lanes lie inside one DPP row (row_shr); wider slots span rows, which
// row_shr does not cross: they shuffle.  The lanes a shift would read from outside the slot keep their own value.
constexpr int DPP_SHR1 = 0x111, DPP_SHR2 = 0x112, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118; // row_shr:1 / 2 / 4 / 8
template <class Op>
__device__ __forceinline__ double slot_prefix(double v, int sub, int lg, Op op) {
  double t;
  if (lg <= 4) {
    t = dpp_f64<DPP_SHR1>(v); v = sub >= 1 ? op(v, t) : v;
    t = dpp_f64<DPP_SHR2>(v); v = sub >= 2 ? op(v, t) : v;
    if (lg >= 3) { t = dpp_f64<DPP_SHR4>(v); v = sub >= 4 ? op(v, t) : v; }
    if (lg >= 4) { t = dpp_f64<DPP_SHR8>(v); v = sub >= 8 ? op(v, t) : v; }
  } else {
    for (int o = 1; o < (1 << lg); o <<= 1) { t = __shfl_up(v, o, 64); v = sub >= o ? op(v, t) : v; }
  }
  return v;
}
struct OpMin { __device__ __forceinline__ double operator()(double a, double b) const { return b < a ? b : a; } };

// Lane-parallel evaluate / grad of one observation per slot.  Called by ALL lanes of the wave in uniform control flow (every
// cross-lane operation below executes with the full wave; slots differ only in data): dd = embedding dimension of the slot's
// observation (0 = idle slot), lane `sub` holds u_sub (sub < dd), u0 = u_0 in every lane (scalar losses).  Returns the loss
// (uniform over the slot) and leaves component `sub` of the gradient in cg.  Same formulas as src/losses.jl:377-406,424-446,
// 461-483,499-519,581-609 with the inner sums over the categories done as slot reductions:
//   Multinomial: sumexp_j of the reference is sum_j' exp(u_j' - max u) for every j, and exp(-M_j) = exp(u_j - max u);
//   Ordistic:    the same with v_j = -u_j^2;
//   MultinomialOrdinal: enforce_MNLOrdRules (:572-578) in closed form, u'_j = min(u_0, u_1 + TOL, ..., u_j + j TOL, -TOL) - j TOL.
//
// KM: the loss kinds the MODEL holds, as a bit mask (bit `kind` for the five multi-dimensional kinds, bit 0 = some scalar loss): the code
// of kinds the model does not have is compiled out.  The all-kinds row kernel is 11 000 instructions with every formula inlined three
// times (gradient pass, trial pass, fixed-step pass) and spills at the 128-VGPR cap that four waves per SIMD need; for an all-Multinomial
// model (KM = MULTI_KM_MNL) it is 6 100 instructions, 119 VGPRs, no scratch.  Instantiated: everything, MultinomialLoss only,
// MultinomialLoss + scalar losses (the categorical + real / boolean columns of a typical data frame), BvSLoss + MultinomialOrdinalLoss
// (an ordinal data frame).
constexpr int MULTI_KM_ALL = 0xFFFF, MULTI_KM_SCALAR = 1, MULTI_KM_MNL = 1 << GLRM_LOSS_MULTINOMIAL;
constexpr int MULTI_KM_ORD = (1 << GLRM_LOSS_BVS) | (1 << GLRM_LOSS_MULTINOMIAL_ORDINAL); // the ordinal columns: no exponential of a block at all
template <int KM>
constexpr bool mk_has(int kind) { return ((KM >> kind) & 1) != 0; }
template <bool GRAD, bool TRIG, int KM = MULTI_KM_ALL>
__device__ inline double obs_loss(const LossDesc& l, double u, double u0, double av, int dd, int sub, int lg, int slot_lane0, double* us,
                                  double& cg) {
  const double s = l.scale;
  const int kind = l.kind, P = 1 << lg;
  const bool in = sub < dd, vec = dd > 1;
  const int a = (int)av - 1;                           // level - 1: 0 .. dd (BvS / MultinomialOrdinal have dd + 1 levels)
  const int ash = a < 0 ? 0 : (a > P - 1 ? P - 1 : a); // in-range lane for the shuffles (idle / scalar slots too)
  // stage 1: slot maximum
  double mv = -__builtin_inf();
  if (mk_has<KM>(GLRM_LOSS_MULTINOMIAL) && in && vec && kind == GLRM_LOSS_MULTINOMIAL) mv = u;
  if (mk_has<KM>(GLRM_LOSS_ORDISTIC) && in && vec && kind == GLRM_LOSS_ORDISTIC) mv = -(u * u);
  const double mx = slot_reduce(mv, lg, OpMax());
  // stage 2: per-lane term, slot sum
  double term = 0.0, dLb = 0.0;
  if (in && vec) {
    if (mk_has<KM>(GLRM_LOSS_MULTINOMIAL) && kind == GLRM_LOSS_MULTINOMIAL) term = fm_exp(u - mx);          // arguments <= 0 (glrm_fastmath.hpp)
    else if (mk_has<KM>(GLRM_LOSS_ORDISTIC) && kind == GLRM_LOSS_ORDISTIC) term = fm_exp(-(u * u) - mx);
    else if ((mk_has<KM>(GLRM_LOSS_OVA) || mk_has<KM>(GLRM_LOSS_BVS)) && (kind == GLRM_LOSS_OVA || kind == GLRM_LOSS_BVS)) {
      const bool truth = kind == GLRM_LOSS_OVA ? a == sub : a > sub;
      loss_both<GRAD, false>(bin_loss_of(l), u, truth ? 1.0 : 0.0, term, dLb);
    }
  }
  const double se = slot_reduce(term, lg, OpSum());
  const double ua = __shfl(u, slot_lane0 + ash, 64);
  // MultinomialOrdinal: the thresholds u'_j = min(-TOL, min_{i <= j} (u_i + i TOL)) - j TOL as a prefix minimum over the slot's lanes (the
  // reference's running `wi < w ? wi : w` skips a NaN u_i: it enters as +Inf)
  const bool ord = mk_has<KM>(GLRM_LOSS_MULTINOMIAL_ORDINAL) && vec && kind == GLRM_LOSS_MULTINOMIAL_ORDINAL;
  double e_hi = 0.0, e_lo = 0.0, u_hi = 0.0;
  if (__any(ord)) {
    const double TOL = 1e-3;
#if defined(GLRM_MNLORD_LIBM) // A/B build: the thresholds through LDS, one read per lower threshold
    if (ord && in) us[sub] = u;
    wave_sync();
    double w = -TOL;
    if (ord && in) for (int i = 0; i <= sub; ++i) { const double wi = us[i] + i * TOL; w = wi < w ? wi : w; }
#else
    (void)us;
    const double wi = u + sub * TOL;
    const double pm = slot_prefix((ord && in && wi == wi) ? wi : __builtin_inf(), sub, lg, OpMin());
    const double w = (ord && in && pm < -TOL) ? pm : -TOL;
#endif
    const double up = w - sub * TOL;
    const double ea = ord ? fm_exp(up) : 0.0;
    int hi = a > 0 ? a - 1 : 0, lo = a < dd ? a : dd - 1; // lanes of u'_{a-1}, u'_a
    hi = hi < 0 ? 0 : (hi > P - 1 ? P - 1 : hi);
    lo = lo < 0 ? 0 : (lo > P - 1 ? P - 1 : lo);
    e_hi = __shfl(ea, slot_lane0 + hi, 64); // exp(u'_{a-1})
    e_lo = __shfl(ea, slot_lane0 + lo, 64); // exp(u'_a)
    u_hi = __shfl(up, slot_lane0 + hi, 64);
#if defined(GLRM_MNLORD_LIBM)
    wave_sync();
#endif
  }
  // stage 3: per-kind closing formulas (lane-local)
  cg = 0.0;
  double L = 0.0;
  if (mk_has<KM>(0) && dd == 1) {
    loss_both<GRAD, TRIG>(l, u0, av, L, cg);
  } else if (vec) {
    switch (kind) {
      case GLRM_LOSS_MULTINOMIAL: // se in [1, d]: the slot's largest term is exp(0)
        if (!mk_has<KM>(GLRM_LOSS_MULTINOMIAL)) break;
        L = s * (fm_log_ge1(se) + (mx - ua));
        if (GRAD && in) cg = s * ((sub == a ? -1.0 : 0.0) + term * fm_rcp(se));
        break;
      case GLRM_LOSS_OVA:
      case GLRM_LOSS_BVS:
        if (!(mk_has<KM>(GLRM_LOSS_OVA) || mk_has<KM>(GLRM_LOSS_BVS))) break;
        L = s * se;
        if (GRAD && in) cg = s * dLb;
        break;
      case GLRM_LOSS_ORDISTIC:
        if (!mk_has<KM>(GLRM_LOSS_ORDISTIC)) break;
        L = s * ((ua * ua + mx) + fm_log_ge1(se));
        if (GRAD && in) cg = s * ((sub == a ? 2 * u : 0.0) - 2 * u * term * fm_rcp(se));
        break;
      default: { // GLRM_LOSS_MULTINOMIAL_ORDINAL
        if (!mk_has<KM>(GLRM_LOSS_MULTINOMIAL_ORDINAL)) break;
#if defined(GLRM_MNLORD_LIBM) // A/B build: ocml log and a division per branch (the form before round 3)
        double g = 0.0;
        if (a == 0) {
          L = -s * log(1.0 - e_lo);
          if (sub == 0) g = -e_lo / (1.0 - e_lo);
        } else if (a == dd) {
          L = -s * u_hi;
          if (sub == a - 1) g = 1.0;
        } else {
          const double den = e_hi - e_lo;
          L = -s * log(den);
          if (sub == a) g = -e_lo / den;
          else if (sub == a - 1) g = e_hi / den;
        }
#else
        // src/losses.jl:581-609 with the operands selected per level and ONE logarithm, ONE division: level 1: -log(1 - e_1), the top
        // level: -u'_{d}, between: -log(e_{a-1} - e_a); the two lanes of the thresholds next to the level hold the only nonzero gradients
        const bool bot = a == 0, top = a == dd;
        const double den = bot ? 1.0 - e_lo : e_hi - e_lo;
        const bool at_lo = !top && sub == a, at_hi = !bot && sub == a - 1;
        L = top ? -s * u_hi : -s * fm_log(den);
        double g = 0.0;
        if (GRAD) {
          const double q = (at_lo ? -e_lo : e_hi) / den;
          g = top ? (at_hi ? 1.0 : 0.0) : (at_lo || at_hi ? q : 0.0);
        }
#endif
        if (GRAD && in) cg = -s * g;
      }
    }
  }
  return L;
}

// ---------------------------------------------------------------- one pass over the segment's observations
// Returns the loss sum at the block `own` (workgroup-uniform); GRAD additionally leaves the gradient in Gt.
//
// A wave works on SL = 64/P observations at a time ("slots" of P = 2^lgP lanes, P >= max(k, dmax, 4)).  Lane `sub` of a
// slot holds component `sub` of the own / opposing vector: u_j = <x, y_j> is a slot reduction of the products (u_j ends
// up in lane j), the loss and its d-vector gradient are evaluated lane-parallel (vloss_lanes), and lane `sub` accumulates
// component `sub` of the gradient.  Index / value loads run two wave-iterations ahead and (columns) the opposing row
// one iteration ahead.  Slot partials are combined in slot order, wave partials in wave order.
//
// RD > 0 (every embedding dimension of the model <= RD = 8): the d vectors an observation meets live in REGISTERS -- lane `sub` holds
// component `sub` of each -- instead of LDS.  Rows: the opposing block is read from memory straight into registers (no staging pass,
// no LDS round trip between the load and the dot products); columns: the own block is read out of LDS once per pass.  The components
// of the d-vector gradient come back by lane shuffles instead of through LDS.  Per observation that removes d LDS writes, 2 d LDS reads
// and three wave barriers; every product and every sum is formed in the same order as on the LDS path (same bits).
template <bool ROWS, int NW, bool GRAD, int GDC, bool TRIG, int RD = 0, int KM = MULTI_KM_ALL>
__device__ inline double multi_pass(const MultiArgs& a, int64_t b, int64_t e, const double* own, double* wbase, double* Gt, double* red,
                                    const LossDesc& lseg, int dseg) {
  constexpr int NT = NW * 64, GD = ROWS ? 1 : GDC; // GDC >= the largest embedding dimension of the problem
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lg = a.lgP, P = 1 << lg, SL = 64 >> lg, slot = lane >> lg, sub = lane & (P - 1);
  const int S = a.kp + 1, k = a.k, kp = a.kp;
  const int DO = ROWS ? 1 : dseg;
  const int dtcap = ROWS ? a.dmax : 1;
  double* oth = wbase + (size_t)slot * (dtcap * S + 64);
  double* us = oth + dtcap * S;
  double* cgs = us + 32;
  const bool comp = sub < k;
  const double xown = (ROWS && comp) ? own[sub] : 0.0;
  double yreg[RD > 0 ? RD : 1]; // RD > 0: component `sub` of the block's vectors
  if constexpr (RD > 0 && !ROWS) {
#pragma unroll
    for (int j = 0; j < RD; ++j) yreg[j] = (j < DO && comp) ? own[j * S + sub] : 0.0;
  }
  double lsum = 0.0;
  double G[GD];
#pragma unroll
  for (int j = 0; j < GD; ++j) G[j] = 0.0;
  const int64_t stride = (int64_t)NW * SL;
  int64_t t = b + (int64_t)wave * SL + slot;
  // software pipeline: index / value loads run three wave-iterations ahead, the opposing row (columns) two ahead
  int32_t id_n = 0, id_nn = 0, id_nnn = 0;
  double av_n = 0.0, av_nn = 0.0, av_nnn = 0.0, xr = 0.0, xr_n = 0.0;
  if (e > b) { // clamped, unconditional loads: the waits stay counted
    const int64_t t1 = t < e ? t : e - 1, t2 = t + stride < e ? t + stride : e - 1, t3 = t + 2 * stride < e ? t + 2 * stride : e - 1;
    id_n = a.idx[t1]; av_n = a.vals[t1];
    id_nn = a.idx[t2]; av_nn = a.vals[t2];
    id_nnn = a.idx[t3]; av_nnn = a.vals[t3];
    if constexpr (!ROWS) {
      xr = a.other[(int64_t)id_n * kp + (comp ? sub : 0)];
      xr_n = a.other[(int64_t)id_nn * kp + (comp ? sub : 0)];
    }
  }
  // Rows with register-resident blocks: the chain index -> (loss descriptor, first vector of the column) -> the column's vectors is
  // software-pipelined as well.  PF = 1: descriptor and ystart of the NEXT observation are requested while this one is worked on, so
  // the block loads at the top of an iteration wait for one memory round trip instead of two; PF = 2: they run two ahead and the block
  // itself one ahead (RD more doubles per lane).
  constexpr int PF = (ROWS && RD > 0) ? GLRM_MULTI_PF : 0;
  LossDesc l_n = lseg, l_nn = lseg;
  int d_n = 1, d_nn = 1;
  int64_t ys_n = 0, ys_nn = 0;
  double ynx[(PF >= 2) ? RD : 1];
  auto fetch_desc = [&](int32_t idc, LossDesc& lo, int& dc, int64_t& ysc) {
    const int64_t li = a.loss_single ? 0 : idc;
    lo = load_loss(a.losses, li);
    dc = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
    ysc = a.ystart[idc];
  };
  if constexpr (PF >= 1) {
    if (e > b) {
      fetch_desc(id_n, l_n, d_n, ys_n);
      if constexpr (PF >= 2) {
        fetch_desc(id_nn, l_nn, d_nn, ys_nn);
        const double* Yb = a.other + ys_n * kp;
#pragma unroll
        for (int j = 0; j < RD; ++j) ynx[j] = (j < d_n && comp) ? Yb[j * kp + sub] : 0.0;
      }
    }
  }
  for (int64_t t0 = b + (int64_t)wave * SL; t0 < e; t0 += stride) { // wave-uniform trip count
    const bool valid = t < e;
    const int32_t id = id_n;
    const double av = av_n;
    const double xc = ROWS ? xown : (comp ? xr : 0.0);
    id_n = id_nn; av_n = av_nn;
    id_nn = id_nnn; av_nn = av_nnn;
    xr = xr_n;
    {
      const int64_t t4 = t + 3 * stride < e ? t + 3 * stride : e - 1;
      id_nnn = a.idx[t4]; av_nnn = a.vals[t4];
    }
    if constexpr (!ROWS) xr_n = a.other[(int64_t)id_nn * kp + (comp ? sub : 0)]; // the row two observations ahead
    LossDesc l = lseg;
    int d = dseg;
    if constexpr (PF >= 1) { // (id_n is the NEXT observation from here on)
      l = l_n; d = d_n;
      if constexpr (PF >= 2) {
#pragma unroll
        for (int j = 0; j < RD; ++j) yreg[j] = ynx[j];
        l_n = l_nn; d_n = d_nn; ys_n = ys_nn;
        fetch_desc(id_nn, l_nn, d_nn, ys_nn);
        const double* Yn = a.other + ys_n * kp;
#pragma unroll
        for (int j = 0; j < RD; ++j) ynx[j] = (j < d_n && comp) ? Yn[j * kp + sub] : 0.0;
      } else {
        const double* Yb = a.other + ys_n * kp;
#pragma unroll
        for (int j = 0; j < RD; ++j) yreg[j] = (j < d && comp) ? Yb[j * kp + sub] : 0.0;
        fetch_desc(id_n, l_n, d_n, ys_n);
      }
    } else if constexpr (ROWS) {
      const int64_t li = a.loss_single ? 0 : id;
      l = load_loss(a.losses, li);
      d = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
      const double* Yb = a.other + a.ystart[id] * kp; // the d vectors of column id are contiguous (id: clamped into the list, always valid)
      if constexpr (RD > 0) {
#pragma unroll
        for (int j = 0; j < RD; ++j) yreg[j] = (j < d && comp) ? Yb[j * kp + sub] : 0.0;
      } else {
        if (valid) {
          for (int i = sub; i < d * kp; i += P) { const int j = i / kp, c = i - j * kp; oth[j * S + c] = Yb[i]; }
        }
        wave_sync();
      }
    }
    const double* blk = ROWS ? oth : own; // the d vectors this observation meets
    const int dd = valid ? d : 0;
    // u_j = <x, y_j>: one slot reduction per j, uniform trip count (the largest d among the wave's slots)
    double u = 0.0, u0 = 0.0;
    if constexpr (RD > 0) {
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        if (!__any(j < dd)) break;
        const double r = slot_reduce((comp && j < dd) ? xc * yreg[j] : 0.0, lg, OpSum());
        u = (sub == j && j < dd) ? r : u;
        u0 = j == 0 ? r : u0;
      }
    } else {
      for (int j = 0; __any(j < dd); ++j) {
        const double r = slot_reduce((comp && j < dd) ? xc * blk[j * S + sub] : 0.0, lg, OpSum());
        u = (sub == j && j < dd) ? r : u;
        u0 = j == 0 ? r : u0;
      }
    }
    double cg = 0.0;
    const double L = obs_loss<GRAD, TRIG, KM>(l, u, u0, av, dd, sub, lg, lane - sub, us, cg);
    lsum += L;
    if constexpr (GRAD && RD > 0) {
      // component j of the observation's gradient sits in lane j of the slot: fetch it by a shuffle (executed by the whole wave,
      // uniform trip count) and accumulate in the order j = 0, 1, ... of the LDS path.  A scalar loss has cg uniform over its slot.
      double g0 = G[0];
#pragma unroll
      for (int j = 0; j < RD; ++j) {
        if (!__any(j < dd)) break;
        const double cj = __shfl(cg, (lane - sub) + j, 64);
        if (valid && comp && j < d) {
          if constexpr (ROWS) g0 = fma(cj, yreg[j], g0); // g += Y_f * curgrad
          else G[j < GD ? j : 0] = fma(cj, xc, G[j < GD ? j : 0]); // G += x * curgrad'
        }
      }
      if constexpr (ROWS) G[0] = g0;
    } else if constexpr (GRAD) {
      const bool vec = valid && d > 1;
      if (__any(vec)) {
        if (vec && sub < d) cgs[sub] = cg;
        wave_sync();
      }
      if (valid && comp) {
        if constexpr (ROWS) { // g += Y_f * curgrad
          double g = G[0];
          if (d == 1) g = fma(cg, oth[sub], g);
          else for (int j = 0; j < d; ++j) g = fma(cgs[j], oth[j * S + sub], g);
          G[0] = g;
        } else {              // G += x * curgrad'
          if (d == 1) G[0] = fma(cg, xc, G[0]);
          else {
#pragma unroll
            for (int j = 0; j < GD; ++j) if (j < d) G[j] = fma(cgs[j], xc, G[j]);
          }
        }
      }
      wave_sync();
    } else if constexpr (ROWS && RD == 0) {
      wave_sync(); // oth is overwritten by the next staging
    }
    t += stride;
  }
  // slot partials -> wave partial (slot order); lsum is uniform inside a slot
  double wsum = 0.0;
  for (int sidx = 0; sidx < SL; ++sidx) wsum += __shfl(lsum, sidx << lg, 64);
  double total = wsum;
  if constexpr (NW > 1) {
    __syncthreads();
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    total = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += red[w];
  }
  if constexpr (GRAD) {
#pragma unroll
    for (int j = 0; j < GD; ++j) {
      if (j < DO) { // uniform
        double acc = 0.0;
        for (int sidx = 0; sidx < SL; ++sidx) acc += __shfl(G[j], (sidx << lg) + sub, 64);
        G[j] = acc;
      }
    }
    for (int i = tid; i < DO * S; i += NT) Gt[i] = 0.0;
    __syncthreads();
    for (int w = 0; w < NW; ++w) {
      if (wave == w && lane < k) { // lane < k <= P: slot 0
#pragma unroll
        for (int j = 0; j < GD; ++j) if (j < DO) Gt[j * S + lane] += G[j];
      }
      __syncthreads();
    }
  }
  return total;
}

// LDS carve-up (doubles): ownA | ownB | Gt (DOcap*S each) | tmp[64] | red[16] | per wave and slot: oth[DTcap*S] us[32] cgs[32]
__host__ __device__ inline size_t multi_lds_doubles(bool rows, int nw, int kp, int dmax, int lgP) {
  const size_t S = kp + 1, docap = rows ? 1 : dmax, dtcap = rows ? dmax : 1, sl = 64 >> lgP;
  return 3 * docap * S + 64 + 16 + (size_t)nw * sl * (dtcap * S + 64);
}

// TRIG = false: the scalar-loss columns of the model hold no PeriodicLoss (LOSS_*_NOTRIG in glrm_engine.hpp)
template <bool ROWS, int NW, int GDC, bool TRIG, int RD = 0, int KM = MULTI_KM_ALL>
__global__ void __launch_bounds__(NW * 64, NW == 1 ? 4 : 2) multi_sweep_kernel(const MultiArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int S = a.kp + 1, k = a.k, kp = a.kp;
  const int docap = ROWS ? 1 : a.dmax, dtcap = ROWS ? a.dmax : 1;
  double* ownA = sm;
  double* ownB = ownA + docap * S;
  double* Gt = ownB + docap * S;
  double* tmp = Gt + docap * S;
  double* red = tmp + 64;
  double* wbase = red + 16 + (size_t)wave * (64 >> a.lgP) * (dtcap * S + 64); // this wave's slots

  const int64_t s = blockIdx.x, gseg = a.own_offset + s;
  const int64_t b = a.ptr[s], e = a.ptr[s + 1];
  LossDesc lseg{};
  int dseg = 1;
  int64_t vec0 = gseg;
  if constexpr (!ROWS) {
    const int64_t li = a.loss_single ? 0 : gseg;
    lseg = load_loss(a.losses, li);
    dseg = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
    vec0 = a.ystart[gseg];
  }
  const int DO = ROWS ? 1 : dseg;
  double* ownp = a.own + vec0 * kp;
  for (int i = tid; i < DO * kp; i += NT) { const int j = i / kp, c = i - j * kp; ownA[j * S + c] = ownp[i]; }
  __syncthreads();
  const glrm_reg rg = a.regs[a.reg_single ? 0 : s];

  if (a.mode == 1) { // losses only
    const double tot = multi_pass<ROWS, NW, false, GDC, TRIG, RD, KM>(a, b, e, ownA, wbase, Gt, red, lseg, dseg);
    if (tid == 0 && a.obj) a.obj[gseg] = tot;
    return;
  }
  const double loss_old = multi_pass<ROWS, NW, true, GDC, TRIG, RD, KM>(a, b, e, ownA, wbase, Gt, red, lseg, dseg);
  const double l1 = (double)(e - b) + 1;
  if (a.mode == 2) { // sparse_proxgrad.jl:72-78 / :94-99: scale the gradient, add, prox -- no line search
    const double st = a.fixed_alpha / l1;
    for (int i = tid; i < DO * kp; i += NT) {
      const int j = i / kp, c = i - j * kp;
      if (c < k) { const double g = Gt[j * S + c] * (-st); ownA[j * S + c] = ownA[j * S + c] + g; }
    }
    __syncthreads();
    block_prox<NW>(ownA, S, k, DO, rg, st, tmp);
    for (int i = tid; i < DO * kp; i += NT) { const int j = i / kp, c = i - j * kp; ownp[i] = ownA[j * S + c]; }
    return;
  }
  double obj = loss_old + block_reg_eval<NW>(ownA, S, k, DO, rg, red);
  double alpha = a.alpha[s];
  int ntr = 0, nacc = 0;
  while (alpha > a.min_stepsize) { // proxgrad.jl:137-155 / :180-200
    const double stepsize = alpha / l1;
    for (int i = tid; i < DO * kp; i += NT) {
      const int j = i / kp, c = i - j * kp;
      ownB[j * S + c] = c < k ? fma(-stepsize, Gt[j * S + c], ownA[j * S + c]) : 0.0;
    }
    __syncthreads();
    block_prox<NW>(ownB, S, k, DO, rg, stepsize, tmp);
    const double nloss = multi_pass<ROWS, NW, false, GDC, TRIG, RD, KM>(a, b, e, ownB, wbase, Gt, red, lseg, dseg);
    const double nobj = nloss + block_reg_eval<NW>(ownB, S, k, DO, rg, red);
    ++ntr;
    if (nobj < obj) {
      for (int i = tid; i < DO * kp; i += NT) { const int j = i / kp, c = i - j * kp; ownp[i] = ownB[j * S + c]; }
      alpha *= 1.05;
      obj = nobj;
      nacc = 1;
      break;
    } else {
      alpha *= .7;
      if (alpha < a.min_stepsize) { alpha = a.min_stepsize * 1.1; break; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    a.alpha[s] = alpha;
    if (a.obj) a.obj[gseg] = obj;
    if (a.trials) a.trials[s] += ntr;
    if (a.accepts) a.accepts[s] += nacc;
  }
}

// ---------------------------------------------------------------- split column sweeps
// A tall model has few columns with very long observation lists (one workgroup per column leaves most of the chip idle), so
// the Y half-step is also available as rounds of two kernels: multi_colpass_kernel spreads each column's list over `nsplit`
// workgroups that write partial (loss, gradient) records, multi_coldecide_kernel adds the partials in split order and runs one
// step of the column's line-search state machine (proxgrad.jl:177-200).  The host repeats trial pass + decide until no column
// is active.  Same arithmetic per observation; only the grouping of the partial sums differs from the one-kernel sweep.
struct SplitArgs {
  MultiArgs m;
  int nsplit;
  int64_t chunk;        // observations per split: a constant, so the grouping of a column's partial sums depends on nothing
                        // but the column's own list (results do not depend on how columns are sharded)
  int round;            // 0 = after the gradient pass, >= 1 = after a trial pass, -1 = losses only
  const double* point;  // pass: the block to evaluate (own factor or trial buffer), indexed like m.own
  double* trial;        // trial points, indexed like m.own
  double* part_loss;    // [nseg][nsplit]
  double* part_G;       // [nseg][nsplit][dmax * kp]
  double* gtot;         // [nseg][dmax * kp]
  double* objold;       // [nseg]
  int32_t* active;      // [nseg]
  unsigned int* nactive;
};

template <bool GRAD, int GDC, bool TRIG, int KM = MULTI_KM_ALL>
__global__ void __launch_bounds__(512, (GRAD && GDC > 8) ? 2 : 4) multi_colpass_kernel(const SplitArgs sa) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int NW = 8, NT = NW * 64;
  const MultiArgs& a = sa.m;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int S = a.kp + 1, kp = a.kp;
  const int64_t s = blockIdx.x, gseg = a.own_offset + s;
  const int y = blockIdx.y;
  if (!GRAD && sa.round >= 1 && !sa.active[s]) return;
  double* own = sm;
  double* Gt = own + a.dmax * S;
  double* red = Gt + a.dmax * S;
  double* wbase = red + 16 + (size_t)wave * (64 >> a.lgP) * (S + 64);
  const int64_t li = a.loss_single ? 0 : gseg;
  const LossDesc lseg = load_loss(a.losses, li);
  const int dseg = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
  const int64_t vec0 = a.ystart[gseg];
  const double* src = sa.point + vec0 * kp;
  for (int i = tid; i < dseg * kp; i += NT) { const int j = i / kp, c = i - j * kp; own[j * S + c] = src[i]; }
  __syncthreads();
  const int64_t b0 = a.ptr[s], e0 = a.ptr[s + 1];
  int64_t b = b0 + (int64_t)y * sa.chunk, e = b + sa.chunk;
  b = b < e0 ? b : e0;
  e = e < e0 ? e : e0;
  const double tot = multi_pass<false, NW, GRAD, GDC, TRIG, (GDC <= 8 ? 8 : 0), KM>(a, b, e, own, wbase, Gt, red, lseg, dseg);
  if (tid == 0) sa.part_loss[s * sa.nsplit + y] = tot;
  if constexpr (GRAD) {
    double* pg = sa.part_G + ((size_t)s * sa.nsplit + y) * a.dmax * kp;
    for (int i = tid; i < dseg * kp; i += NT) { const int j = i / kp, c = i - j * kp; pg[i] = c < a.k ? Gt[j * S + c] : 0.0; }
  }
}

// One workgroup per column, same shape (8 waves) and the same block_prox / block_reg_eval instantiations as the one-kernel
// sweep: a column whose list fits one chunk gets bit-identical results on either path.
__global__ void __launch_bounds__(512) multi_coldecide_kernel(const SplitArgs sa) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const MultiArgs& a = sa.m;
  const int tid = threadIdx.x, S = a.kp + 1, k = a.k, kp = a.kp;
  const int64_t s = blockIdx.x, gseg = a.own_offset + s;
  if (sa.round >= 1 && !sa.active[s]) return;
  const int64_t li = a.loss_single ? 0 : gseg;
  const int DO = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
  const int64_t vec0 = a.ystart[gseg];
  double* blkA = sm;                 // own block
  double* blkB = blkA + a.dmax * S;  // trial block
  double* tmp = blkB + a.dmax * S;
  double* red = tmp + 64;
  double loss = 0.0;                 // partial losses in split order
  for (int y = 0; y < sa.nsplit; ++y) loss += sa.part_loss[s * sa.nsplit + y];
  if (sa.round < 0) { // losses only
    if (tid == 0 && a.obj) a.obj[gseg] = loss;
    return;
  }
  const glrm_reg rg = a.regs[a.reg_single ? 0 : s];
  double* ownp = a.own + vec0 * kp;
  double* trialp = sa.trial + vec0 * kp;
  double* gt = sa.gtot + (size_t)s * a.dmax * kp;
  const double l1 = (double)(a.ptr[s + 1] - a.ptr[s]) + 1;
  for (int i = tid; i < DO * kp; i += 512) { const int j = i / kp, c = i - j * kp; blkA[j * S + c] = ownp[i]; }
  double alpha = a.alpha[s], obj;
  bool finished = false;
  int ntr = 0, nacc = 0;
  if (sa.round == 0) {
    for (int i = tid; i < DO * kp; i += 512) { // gradient = partials in split order
      double g = 0.0;
      for (int y = 0; y < sa.nsplit; ++y) g += sa.part_G[((size_t)s * sa.nsplit + y) * a.dmax * kp + i];
      gt[i] = g;
    }
    __syncthreads();
    if (a.mode == 2) { // fixed step, no line search
      const double st = a.fixed_alpha / l1;
      for (int i = tid; i < DO * kp; i += 512) {
        const int j = i / kp, c = i - j * kp;
        if (c < k) { const double g = gt[i] * (-st); blkA[j * S + c] = blkA[j * S + c] + g; }
      }
      __syncthreads();
      block_prox<8>(blkA, S, k, DO, rg, st, tmp);
      for (int i = tid; i < DO * kp; i += 512) { const int j = i / kp, c = i - j * kp; ownp[i] = blkA[j * S + c]; }
      if (tid == 0) sa.active[s] = 0;
      return;
    }
    __syncthreads();
    obj = loss + block_reg_eval<8>(blkA, S, k, DO, rg, red);
    if (!(alpha > a.min_stepsize)) finished = true;
  } else {
    for (int i = tid; i < DO * kp; i += 512) { const int j = i / kp, c = i - j * kp; blkB[j * S + c] = trialp[i]; }
    __syncthreads();
    const double nobj = loss + block_reg_eval<8>(blkB, S, k, DO, rg, red);
    obj = sa.objold[s];
    ntr = 1;
    if (nobj < obj) {
      for (int i = tid; i < DO * kp; i += 512) ownp[i] = trialp[i];
      alpha *= 1.05;
      obj = nobj;
      nacc = 1;
      finished = true;
    } else {
      alpha *= .7;
      if (alpha < a.min_stepsize) { alpha = a.min_stepsize * 1.1; finished = true; }
      else if (!(alpha > a.min_stepsize)) finished = true; // `while alpha > min_stepsize` (proxgrad.jl:180)
    }
  }
  if (!finished) { // next trial point
    const double stepsize = alpha / l1;
    for (int i = tid; i < DO * kp; i += 512) {
      const int j = i / kp, c = i - j * kp;
      blkB[j * S + c] = c < k ? fma(-stepsize, gt[i], blkA[j * S + c]) : 0.0;
    }
    __syncthreads();
    block_prox<8>(blkB, S, k, DO, rg, stepsize, tmp);
    for (int i = tid; i < DO * kp; i += 512) { const int j = i / kp, c = i - j * kp; trialp[i] = blkB[j * S + c]; }
  }
  if (tid == 0) {
    a.alpha[s] = alpha;
    sa.objold[s] = obj;
    sa.active[s] = finished ? 0 : 1;
    if (!finished) atomicAdd(sa.nactive, 1u);
    if (finished && a.obj) a.obj[gseg] = obj;
    if (a.trials) a.trials[s] += ntr;
    if (a.accepts) a.accepts[s] += nacc;
  }
}

// calc_penalty pieces (src/evaluate_fit.jl:91-104) for wrapped / block regularizers: one wave per segment
__global__ void __launch_bounds__(64) multi_penalty_kernel(const PenaltyArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x, S = a.kp + 1, kp = a.kp;
  const int64_t s = blockIdx.x, gseg = a.own_offset + s;
  const int64_t v0 = a.ystart ? a.ystart[gseg] : gseg;
  const int DO = a.ystart ? (int)(a.ystart[gseg + 1] - v0) : 1;
  double* blk = sm;
  double* red = sm + (size_t)DO * S; // unused for one wave
  const double* src = a.own + v0 * kp;
  for (int i = tid; i < DO * kp; i += 64) { const int j = i / kp, c = i - j * kp; blk[j * S + c] = src[i]; }
  __syncthreads();
  const glrm_reg rg = a.regs[a.reg_single ? 0 : s];
  const double v = block_reg_eval<1>(blk, S, a.k, DO, rg, red);
  if (tid == 0) a.out[gseg] = v;
}

} // namespace glrm
