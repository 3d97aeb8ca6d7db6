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
      for (int j = 0; j < d; ++j) { loss_both<false>(b, u[j], a == j ? 1.0 : 0.0, L, dL); loss += L; }
      return s * loss;
    }
    case GLRM_LOSS_BVS: { // :461-467
      const LossDesc b = bin_loss_of(l);
      double loss = 0.0, L, dL;
      for (int j = 0; j < d; ++j) { loss_both<false>(b, u[j], a > j ? 1.0 : 0.0, L, dL); loss += L; }
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
      loss_both<true>(bin_loss_of(l), u[j], a == j ? 1.0 : 0.0, L, dL);
      return s * dL;
    }
    case GLRM_LOSS_BVS: {
      double L, dL;
      loss_both<true>(bin_loss_of(l), u[j], a > j ? 1.0 : 0.0, L, dL);
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

// ---------------------------------------------------------------- one pass over the segment's observations
// Returns the loss sum at the block `own` (workgroup-uniform); GRAD additionally leaves the gradient in Gt.
template <bool ROWS, int NW, bool GRAD>
__device__ inline double multi_pass(const MultiArgs& a, int64_t b, int64_t e, const double* own, double* oth, double* us, double* cgs,
                                    double* Gt, double* red, const LossDesc& lseg, int dseg) {
  constexpr int NT = NW * 64, GD = ROWS ? 1 : GLRM_MAX_EMBEDDING_DIM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.kp + 1, k = a.k, kp = a.kp;
  const int DO = ROWS ? 1 : dseg;
  double lsum = 0.0;
  double G[GD];
#pragma unroll
  for (int j = 0; j < GD; ++j) G[j] = 0.0;
  for (int64_t t = b + wave; t < e; t += NW) {
    const int32_t id = a.idx[t];
    const double av = a.vals[t];
    LossDesc l = lseg;
    int d = dseg;
    if constexpr (ROWS) {
      const int64_t li = a.loss_single ? 0 : id;
      l = load_loss(a.losses, li);
      d = a.losses[li].dim > 1 ? a.losses[li].dim : 1;
      const double* Yb = a.other + a.ystart[id] * kp; // the d vectors of column id are contiguous
      for (int i = lane; i < d * kp; i += 64) { const int j = i / kp, c = i - j * kp; oth[j * S + c] = Yb[i]; }
    } else {
      if (lane < kp) oth[lane] = a.other[(int64_t)id * kp + lane];
    }
    wave_sync();
    if (lane < d) { // u_j = <x, y_j>, sequential over the components
      const double* p = ROWS ? own : own + lane * S;
      const double* q = ROWS ? oth + lane * S : oth;
      double u = 0.0;
      for (int c = 0; c < k; ++c) u = fma(p[c], q[c], u);
      us[lane] = u;
    }
    wave_sync();
    if (l.kind == GLRM_LOSS_MULTINOMIAL_ORDINAL) {
      if (lane == 0) enforce_mnl_ord_rules(us, d);
      wave_sync();
    }
    const int ai = (int)av - 1;
    double L, dL = 0.0;
    if (d == 1) loss_both<GRAD>(l, us[0], av, L, dL);
    else L = vloss_eval(l, us, d, ai);
    lsum += L;
    if constexpr (GRAD) {
      if (d > 1) {
        if (lane < d) cgs[lane] = vloss_grad(l, us, d, ai, lane);
        wave_sync();
      }
      if (lane < k) {
        if constexpr (ROWS) { // g += Y_f * curgrad
          double g = G[0];
          if (d == 1) g = fma(dL, oth[lane], g);
          else for (int j = 0; j < d; ++j) g = fma(cgs[j], oth[j * S + lane], g);
          G[0] = g;
        } else {              // G += x * curgrad'
          const double xc = oth[lane];
          if (d == 1) G[0] = fma(dL, xc, G[0]);
          else {
#pragma unroll
            for (int j = 0; j < GD; ++j) if (j < d) G[j] = fma(cgs[j], xc, G[j]);
          }
        }
      }
    }
    wave_sync();
  }
  double total = lsum;
  if constexpr (NW > 1) {
    __syncthreads();
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    total = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += red[w];
  }
  if constexpr (GRAD) {
    for (int i = tid; i < DO * S; i += NT) Gt[i] = 0.0;
    __syncthreads();
    for (int w = 0; w < NW; ++w) {
      if (wave == w && lane < k) {
#pragma unroll
        for (int j = 0; j < GD; ++j) if (j < DO) Gt[j * S + lane] += G[j];
      }
      __syncthreads();
    }
  }
  return total;
}

// LDS carve-up (doubles): ownA | ownB | Gt (DOcap*S each) | tmp[64] | red[16] | per wave: oth[DTcap*S] us[32] cgs[32]
__host__ __device__ inline size_t multi_lds_doubles(bool rows, int nw, int kp, int dmax) {
  const size_t S = kp + 1, docap = rows ? 1 : dmax, dtcap = rows ? dmax : 1;
  return 3 * docap * S + 64 + 16 + (size_t)nw * (dtcap * S + 64);
}

template <bool ROWS, int NW>
__global__ void __launch_bounds__(NW * 64) multi_sweep_kernel(const MultiArgs a) {
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
  double* oth = red + 16 + (size_t)wave * (dtcap * S + 64);
  double* us = oth + dtcap * S;
  double* cgs = us + 32;

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
    const double tot = multi_pass<ROWS, NW, false>(a, b, e, ownA, oth, us, cgs, Gt, red, lseg, dseg);
    if (tid == 0 && a.obj) a.obj[gseg] = tot;
    return;
  }
  const double loss_old = multi_pass<ROWS, NW, true>(a, b, e, ownA, oth, us, cgs, Gt, red, lseg, dseg);
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
    const double nloss = multi_pass<ROWS, NW, false>(a, b, e, ownB, oth, us, cgs, Gt, red, lseg, dseg);
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
