// glrm_cached.hip -- row sweep with the row's opposing vectors fetched ONCE per half-step ("cached gather sweep").
//
// Where the opposing factor is far beyond L2 and a row is short (BASELINE config 4: 100 observations per row, rank 64, Y = 51 MB), the
// gather sweeps are bound by the k-vector gathers -- 512 B per observation and PASS, gradient pass and every line-search trial alike
// (the X half-step at C4: 2 x 1e9 x 512 B at the 8.2 TB/s the Infinity Cache delivers for such reads = 121 ms).  But all passes of
// a row's half-step read the SAME vectors: Y does not change during the X half-step.  So the row's vectors are fetched once and kept
// on chip for the gradient pass, the prox and every trial of the backtracking line search; the gather traffic of the half-step drops
// from (1 + trials) passes to one.  Two homes for the row:
//   registers (regcached_sweep_kernel, the default; rows of <= 13 trips of the lane layout): a 64-lane
//             wave may use 512 VGPRs per lane; two waves share a row, each holding every other trip's vectors, all loads of a wave
//             in flight together; the passes are straight-line code over registers.  C4 X half-step 120.6 -> 85.4 ms.
//   LDS       (cached_sweep_kernel; rows up to the LDS budget): one wave per row, the row gathered with LDS-DMA.  C4: 110.5 ms.
//
// Applies to the ROW view; index lists in any order.  Summation: a lane group takes its observations in ascending order, the groups of
// a wave are combined by the butterfly of the gather sweeps (glrm_hip.hip: sweep_pass), the waves of a row in wave order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "glrm_device.hpp"
#include "glrm_engine.hpp"

using namespace glrm;

namespace {

#ifndef CACHED_U
#define CACHED_U 4 // observations per lane group in flight per trip of the LDS variant (2: 111.9 ms, 4: 109.6 ms at C4)
#endif

struct CachedArgs {
  int64_t nseg;
  const int64_t* ptr;
  const int32_t* idx;
  const double* vals;
  double* own;
  int64_t own_offset;
  const double* other;
  double* alpha;
  const glrm_loss* losses;
  const glrm_reg* regs;
  int reg_single;
  int k;
  double fixed_alpha;
  double min_stepsize;
  int32_t* trials;
  int32_t* accepts;
  int cap; // LDS variant: vectors a wave's buffer holds (a multiple of the vectors one DMA instruction moves);
           // register variant: trips of 64 / G observations the longest row of this launch needs
  const int32_t* seglist; // nullable: the launch covers the local rows seglist[0..nseg) -- the rows short enough for the cached
                          // sweep when the shard also holds longer ones -- restricted to [seg_lo, seg_hi) (glrm_hip_step_x_range)
  int64_t seg_lo, seg_hi;
};

__device__ __forceinline__ int64_t cached_segment(const CachedArgs& a, int64_t slot) { // -1: nothing to do for this workgroup
  if (slot >= a.nseg) return -1;
  if (!a.seglist) return slot;
  const int64_t seg = a.seglist[slot];
  return (seg < a.seg_lo || seg >= a.seg_hi) ? -1 : seg;
}

// One pass over the row out of LDS: J = sum of losses at u = <xv, y_t>, and (GRAD) g = sum of dL * y_t.
template <int G, int R, int LOSS, bool GRAD>
__device__ __forceinline__ double cached_pass(const CachedArgs& a, const char* __restrict__ ybuf, const double* __restrict__ lval,
                                              const int32_t* __restrict__ lidx, const Vec<G, R>& xv, Vec<G, R>& g, int len, int gi, int j,
                                              const LossDesc& segloss) {
  constexpr int KPB = G * R * 8, NG = 64 / G, LM = loss_mode(LOSS), U = CACHED_U;
  constexpr bool TRIG = loss_trig(LOSS);
  double J = 0.0;
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  }
  for (int t0 = 0; t0 < len; t0 += NG * U) { // wave-uniform trip count; U observations per group in flight
    double2 y[U][R / 2];
    double av[U];
    int cc[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG + gi;
      valid[u] = t < len;
      const int tt = valid[u] ? t : len - 1;
      const char* yp = ybuf + tt * KPB + j * 16;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) y[u][i] = *reinterpret_cast<const double2*>(yp + i * (G * 16));
      av[u] = lval[tt];
      cc[u] = lidx[tt];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double dot = 0.0;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        dot = fma(xv.v[i].x, y[u][i].x, dot);
        dot = fma(xv.v[i].y, y[u][i].y, dot);
      }
      dot = group_sum<G>(dot);
      double L, dL;
      if constexpr (LOSS == LOSS_QUAD_UNIFORM) {
        const double d = dot - av[u];
        L = segloss.scale * (d * d);
        dL = 2 * d * segloss.scale;
      } else if constexpr (LM == LOSS_SEGMENT) {
        loss_both<GRAD, TRIG>(segloss, dot, av[u], L, dL);
      } else {
        const LossDesc lo = load_loss(a.losses, cc[u]);
        loss_both<GRAD, TRIG>(lo, dot, av[u], L, dL);
      }
      if (!valid[u]) {
        L = 0.0;
        dL = 0.0;
      }
      J += L;
      if (GRAD) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          g.v[i].x = fma(dL, y[u][i].x, g.v[i].x);
          g.v[i].y = fma(dL, y[u][i].y, g.v[i].y);
        }
      }
    }
  }
  J = across_groups_sum<G>(J);
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      g.v[i].x = across_groups_sum<G>(g.v[i].x);
      g.v[i].y = across_groups_sum<G>(g.v[i].y);
    }
  }
  return J;
}

// One wave (= one workgroup) per row.  Dynamic LDS: [cap vectors][cap values][cap indices].
template <int G, int R, int LOSS>
__global__ void __launch_bounds__(64) cached_sweep_kernel(const CachedArgs a) {
  constexpr int KP = G * R, KPB = KP * 8, CPV = KPB / 16, VPI = 64 / CPV; // 16-byte chunks per vector, vectors per DMA instruction
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  const int64_t seg = cached_segment(a, blockIdx.x);
  if (seg < 0) return;
  const int j = lane % G, gi = lane / G;
  const int64_t beg = a.ptr[seg];
  const int len = (int)(a.ptr[seg + 1] - beg);
  const int64_t gseg = a.own_offset + seg;
  double2* ownp = reinterpret_cast<double2*>(a.own + gseg * KP);
  char* ybuf = lds;
  double* lval = reinterpret_cast<double*>(lds + (size_t)a.cap * KPB);
  int32_t* lidx = reinterpret_cast<int32_t*>(lds + (size_t)a.cap * (KPB + 8));

  // the row's (index, value) list -> LDS
  for (int t = lane; t < len; t += 64) {
    lidx[t] = a.idx[beg + t];
    lval[t] = a.vals[beg + t];
  }
  __syncthreads(); // one wave: orders the LDS writes above before the reads below
  Vec<G, R> x, g;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) x.v[i] = ownp[i * G + j];
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : seg);
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != LOSS_PER_OBS) segloss = load_loss(a.losses, 0);
  const double alpha0 = a.alpha[seg];
  // gather the row's opposing vectors into LDS with LDS-DMA: lane l of instruction `it` moves chunk l % CPV of vector it * VPI + l / CPV.
  // No VGPRs, the whole row in flight at once.  Measured alternatives at C4 (X half-step; phase-aligned passes 120 ms): this 112 ms;
  // the gradient pass streamed behind the DMA with counted vmcnt waits 124 ms; 26 ordinary 16-byte loads per lane in flight, then
  // ds_write 129 ms.
  if (len > 0) {
    const int vsub = lane / CPV, chunk = lane % CPV;
    const char* obase = reinterpret_cast<const char*>(a.other) + chunk * 16;
    for (int it = 0; it * VPI < len; ++it) {
      int s = it * VPI + vsub;
      s = s < len ? s : len - 1; // the tail re-reads the last vector into an unused slot
      const char* src = obase + (int64_t)lidx[s] * KPB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(ybuf + it * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the gathered vectors have landed
  __syncthreads();

  // pass 1: gradient + objective at the current point (proxgrad.jl:122-135)
  double Jold = cached_pass<G, R, LOSS, true>(a, ybuf, lval, lidx, x, g, len, gi, j, segloss);
  if (a.fixed_alpha > 0.0) { // src/algorithms/sparse_proxgrad.jl:72-77: g *= -alpha/l; x += g; prox!(r, x, alpha/l)
    const double s = a.fixed_alpha / ((double)len + 1.0);
    Vec<G, R> xn;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      xn.v[i].x = x.v[i].x + g.v[i].x * (-s);
      xn.v[i].y = x.v[i].y + g.v[i].y * (-s);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    if (gi == 0) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
    }
    return;
  }
  Jold += reg_eval<G, R>(rd, x, j, a.k);

  // backtracking line search (proxgrad.jl:136-155); every trial reads the cached vectors
  double alpha = alpha0;
  const double l = (double)len + 1.0;
  int ntrials = 0;
  bool accepted = false;
  while (alpha > a.min_stepsize) {
    const double s = alpha / l;
    Vec<G, R> xn, dummy;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      xn.v[i].x = fma(-s, g.v[i].x, x.v[i].x);
      xn.v[i].y = fma(-s, g.v[i].y, x.v[i].y);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    double Jn = cached_pass<G, R, LOSS, false>(a, ybuf, lval, lidx, xn, dummy, len, gi, j, segloss);
    Jn += reg_eval<G, R>(rd, xn, j, a.k);
    ++ntrials;
    if (Jn < Jold) { // strict; false for NaN and for Inf < Inf
      x = xn;
      alpha *= 1.05;
      Jold = Jn;
      accepted = true;
      break;
    }
    alpha *= .7;
    if (alpha < a.min_stepsize) {
      alpha = a.min_stepsize * 1.1;
      break;
    }
  }
  if (accepted && gi == 0) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = x.v[i];
  }
  if (lane == 0) {
    a.alpha[seg] = alpha;
    if (a.trials) {
      a.trials[seg] += ntrials;
      a.accepts[seg] += accepted ? 1 : 0;
    }
  }
}

// ---- the row's vectors in REGISTERS ---------------------------------------------------------------------------------------------
// A 64-thread workgroup may use 512 VGPRs per lane: at k = 64 a row of up to MAXT * 8 observations is MAXT * 16 VGPRs per lane in the
// lane layout of the gather sweeps (lane group gi holds the vectors of observations gi, gi + NG, ...).  The row's vectors are loaded
// ONCE -- all MAXT * R / 2 16-byte loads of a lane in flight together -- and the gradient pass, the prox and every line-search trial run
// from registers: no LDS, four waves per CU, every trip of a pass independent of the others (the compiler interleaves them).
// (WAVES = 2: two waves share a row, wave w holds the observations (t * WAVES + w) * NG + gi; `gi0` = w * NG + gi and the stride NG * WAVES)
template <int G, int R, int LOSS, int MAXT, bool GRAD, int WAVES = 1>
__device__ __forceinline__ double reg_pass(const CachedArgs& a, const double2 (&y)[MAXT][R / 2], const double (&av)[MAXT], const int (&cc)[MAXT],
                                           const Vec<G, R>& xv, Vec<G, R>& g, int len, int gi, const LossDesc& segloss) {
  constexpr int NG = (64 / G) * WAVES, LM = loss_mode(LOSS);
  constexpr bool TRIG = loss_trig(LOSS);
  double J = 0.0;
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t * NG + (gi & ~(64 / G - 1)) < len) { // wave-uniform: gi = wave * (64 / G) + group
      double dot = 0.0;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        dot = fma(xv.v[i].x, y[t][i].x, dot);
        dot = fma(xv.v[i].y, y[t][i].y, dot);
      }
      dot = group_sum<G>(dot);
      double L, dL;
      if constexpr (LOSS == LOSS_QUAD_UNIFORM) {
        const double d = dot - av[t];
        L = segloss.scale * (d * d);
        dL = 2 * d * segloss.scale;
      } else if constexpr (LM == LOSS_SEGMENT) {
        loss_both<GRAD, TRIG>(segloss, dot, av[t], L, dL);
      } else {
        const LossDesc lo = load_loss(a.losses, cc[t]);
        loss_both<GRAD, TRIG>(lo, dot, av[t], L, dL);
      }
      if (!(t * NG + gi < len)) {
        L = 0.0;
        dL = 0.0;
      }
      J += L;
      if (GRAD) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          g.v[i].x = fma(dL, y[t][i].x, g.v[i].x);
          g.v[i].y = fma(dL, y[t][i].y, g.v[i].y);
        }
      }
    }
  }
  J = across_groups_sum<G>(J);
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      g.v[i].x = across_groups_sum<G>(g.v[i].x);
      g.v[i].y = across_groups_sum<G>(g.v[i].y);
    }
  }
  return J;
}

// Per-wave totals of a row shared by WAVES waves, combined through LDS in wave order: every wave ends with the same bits.
template <int G, int R, int WAVES, bool GRAD>
__device__ __forceinline__ double row_combine(double J, Vec<G, R>& g, double* red, int wave, int lane) {
  constexpr int KP = G * R, STRIDE = KP + 2;
  if constexpr (WAVES == 1) return J;
  const int j = lane % G;
  __syncthreads(); // previous readers of `red` are done
  if (lane < G) {
    if (GRAD) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) *reinterpret_cast<double2*>(&red[wave * STRIDE + i * 2 * G + 2 * j]) = g.v[i];
    }
    if (lane == 0) red[wave * STRIDE + KP] = J;
  }
  __syncthreads();
  double Js = 0.0;
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  }
  for (int w = 0; w < WAVES; ++w) {
    Js += red[w * STRIDE + KP];
    if (GRAD) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        const double2 p = *reinterpret_cast<const double2*>(&red[w * STRIDE + i * 2 * G + 2 * j]);
        g.v[i].x += p.x;
        g.v[i].y += p.y;
      }
    }
  }
  return Js;
}

// WAVES waves (= one workgroup) per row; wave w holds the observations (t * WAVES + w) * (64 / G) + group, t = 0 .. MAXT - 1.
template <int G, int R, int LOSS, int MAXT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) regcached_sweep_kernel(const CachedArgs a) {
  constexpr int KP = G * R, NG = (64 / G) * WAVES;
  __shared__ __attribute__((aligned(16))) double red[WAVES == 1 ? 2 : WAVES * (KP + 2)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t seg = cached_segment(a, blockIdx.x);
  if (seg < 0) return;
  const int j = lane % G, gi = wave * (64 / G) + lane / G;
  const int64_t beg = a.ptr[seg];
  const int len = (int)(a.ptr[seg + 1] - beg);
  const int64_t gseg = a.own_offset + seg;
  double2* ownp = reinterpret_cast<double2*>(a.own + gseg * KP);
  int cc[MAXT];
  double av[MAXT];
  double2 y[MAXT][R / 2];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) { // the group's entries (clamped: lanes past the end re-read the last entry and are masked)
    int tt = t * NG + gi;
    tt = tt < len ? tt : (len > 0 ? len - 1 : 0);
    cc[t] = len > 0 ? a.idx[beg + tt] : 0;
    av[t] = len > 0 ? a.vals[beg + tt] : 0.0;
  }
  Vec<G, R> x, g;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) x.v[i] = ownp[i * G + j];
  const double2* __restrict__ other2 = reinterpret_cast<const double2*>(a.other);
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t * NG + wave * (64 / G) < len) { // wave-uniform
      const double2* yp = other2 + (int64_t)cc[t] * (KP / 2) + j;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) y[t][i] = yp[i * G];
    } else {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) y[t][i] = make_double2(0.0, 0.0);
    }
  }
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : seg);
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != LOSS_PER_OBS) segloss = load_loss(a.losses, 0);

  double Jold = reg_pass<G, R, LOSS, MAXT, true, WAVES>(a, y, av, cc, x, g, len, gi, segloss);
  Jold = row_combine<G, R, WAVES, true>(Jold, g, red, wave, lane);
  if (a.fixed_alpha > 0.0) {
    const double s = a.fixed_alpha / ((double)len + 1.0);
    Vec<G, R> xn;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      xn.v[i].x = x.v[i].x + g.v[i].x * (-s);
      xn.v[i].y = x.v[i].y + g.v[i].y * (-s);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    if (gi == 0) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
    }
    return;
  }
  Jold += reg_eval<G, R>(rd, x, j, a.k);
  double alpha = a.alpha[seg];
  const double l = (double)len + 1.0;
  int ntrials = 0;
  bool accepted = false;
  while (alpha > a.min_stepsize) {
    const double s = alpha / l;
    Vec<G, R> xn, dummy;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      xn.v[i].x = fma(-s, g.v[i].x, x.v[i].x);
      xn.v[i].y = fma(-s, g.v[i].y, x.v[i].y);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    double Jn = reg_pass<G, R, LOSS, MAXT, false, WAVES>(a, y, av, cc, xn, dummy, len, gi, segloss);
    Jn = row_combine<G, R, WAVES, false>(Jn, dummy, red, wave, lane);
    Jn += reg_eval<G, R>(rd, xn, j, a.k);
    ++ntrials;
    if (Jn < Jold) {
      x = xn;
      alpha *= 1.05;
      Jold = Jn;
      accepted = true;
      break;
    }
    alpha *= .7;
    if (alpha < a.min_stepsize) {
      alpha = a.min_stepsize * 1.1;
      break;
    }
  }
  if (accepted && gi == 0) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = x.v[i];
  }
  if (lane == 0 && wave == 0) {
    a.alpha[seg] = alpha;
    if (a.trials) {
      a.trials[seg] += ntrials;
      a.accepts[seg] += accepted ? 1 : 0;
    }
  }
}

// ---- persistent form: a workgroup walks rows slot, slot + gridDim.x, ... and hands the NEXT row's (index, value) list over while it
// works on the current one.  In the one-row-per-workgroup kernel above a row's life is three dependent memory round trips -- row pointer
// -> list -> opposing vectors -- before the first FMA, with two waves per SIMD to hide them (244 VGPRs).  Here the pointers of the next
// row are scalar loads issued one row ahead, its list is requested right behind the current row's gathers (3 * PF dwords per lane, the
// only extra registers) and reaches the lanes through LDS after the current row's passes: the chain per row is ONE round trip, the
// gathers.  Which lane group adds which observation, and in which order, is unchanged: same bits as the kernel above.
template <int G, int R, int LOSS, int MAXT>
__global__ void __launch_bounds__(128, 2) regcached_persist_kernel(const CachedArgs a) {
  constexpr int WAVES = 2, KP = G * R, NG = (64 / G) * WAVES, MAXLEN = MAXT * NG, PF = (MAXLEN + 127) / 128;
  // ONE shared array (a second __shared__ object makes hipcc drain the load queue before every LDS read, cdna_hip_programming.md):
  // [combine buffer: WAVES * (KP + 2) doubles][values: PF * 128 doubles][indices: PF * 128 ints]
  constexpr int RED = WAVES * (KP + 2);
  __shared__ __attribute__((aligned(16))) double sh[RED + PF * 128 + PF * 64];
  double* red = sh;
  double* lvals = sh + RED;
  int* lidx = reinterpret_cast<int*>(sh + RED + PF * 128);
  int* lvals_dw = reinterpret_cast<int*>(lvals);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane % G, gi = wave * (64 / G) + lane / G;
  const RegDesc rd0 = load_reg(a.regs, 0);
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != LOSS_PER_OBS) segloss = load_loss(a.losses, 0);
  const double2* __restrict__ other2 = reinterpret_cast<const double2*>(a.other);

  // segment of a slot (-1: none / filtered out), its list
  auto seg_of = [&](int64_t slot) -> int64_t { return cached_segment(a, slot); };
  auto fetch_list = [&](int64_t beg, int len, int (&pi)[PF], int (&pv)[2 * PF]) { // this lane's share of the row's list, clamped
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      int e = q * 128 + tid;
      e = e < len ? e : (len > 0 ? len - 1 : 0);
#if defined(GLRM_CACHED_NT) // experiment: the streamed lists bypass the caches' retention (the opposing factor is what should stay in them)
      pi[q] = len > 0 ? __builtin_nontemporal_load(a.idx + beg + e) : 0;
      const double vd = len > 0 ? __builtin_nontemporal_load(a.vals + beg + e) : 0.0;
      const int2 v = make_int2(__double2loint(vd), __double2hiint(vd));
#else
      pi[q] = len > 0 ? a.idx[beg + e] : 0;
      const int2 v = len > 0 ? *reinterpret_cast<const int2*>(a.vals + beg + e) : make_int2(0, 0);
#endif
      pv[2 * q] = v.x; pv[2 * q + 1] = v.y;
    }
  };
  auto store_list = [&](const int (&pi)[PF], const int (&pv)[2 * PF]) {
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      lidx[q * 128 + tid] = pi[q];
      lvals_dw[2 * (q * 128 + tid)] = pv[2 * q];
      lvals_dw[2 * (q * 128 + tid) + 1] = pv[2 * q + 1];
    }
  };

  int64_t slot = blockIdx.x;
  int64_t seg = seg_of(slot), seg_n = seg_of(slot + gridDim.x);
  int64_t beg = seg >= 0 ? a.ptr[seg] : 0, beg_n = seg_n >= 0 ? a.ptr[seg_n] : 0;
  int len = seg >= 0 ? (int)(a.ptr[seg + 1] - beg) : 0, len_n = seg_n >= 0 ? (int)(a.ptr[seg_n + 1] - beg_n) : 0;
  {
    int pi[PF], pv[2 * PF];
    fetch_list(beg, len, pi, pv);
    store_list(pi, pv);
  }
  __syncthreads();
  for (; slot < a.nseg; slot += gridDim.x) { // block-uniform
    // the row after the next one: pointers only (scalar loads, consumed an iteration from now)
    const int64_t seg_nn = seg_of(slot + 2 * (int64_t)gridDim.x);
    const int64_t beg_nn = seg_nn >= 0 ? a.ptr[seg_nn] : 0;
    const int len_nn = seg_nn >= 0 ? (int)(a.ptr[seg_nn + 1] - beg_nn) : 0;
    int cc[MAXT];
    double av[MAXT];
    double2 y[MAXT][R / 2];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) { // the group's entries out of LDS (clamped: lanes past the end re-read the last entry and are masked)
      int tt = t * NG + gi;
      tt = tt < len ? tt : (len > 0 ? len - 1 : 0);
      cc[t] = lidx[tt];
      av[t] = lvals[tt];
    }
    const int64_t gseg = a.own_offset + (seg >= 0 ? seg : 0);
    double2* ownp = reinterpret_cast<double2*>(a.own + gseg * KP);
    Vec<G, R> x, g;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) x.v[i] = ownp[i * G + j];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t * NG + wave * (64 / G) < len) { // wave-uniform
        const double2* yp = other2 + (int64_t)cc[t] * (KP / 2) + j;
#pragma unroll
        for (int i = 0; i < R / 2; ++i) y[t][i] = yp[i * G];
      } else {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) y[t][i] = make_double2(0.0, 0.0);
      }
    }
    // the next row's list rides behind the gathers
    int pi[PF], pv[2 * PF];
    fetch_list(beg_n, len_n, pi, pv);
    const RegDesc rd = a.reg_single ? rd0 : load_reg(a.regs, seg >= 0 ? seg : 0);

    if (seg >= 0) {
      double Jold = reg_pass<G, R, LOSS, MAXT, true, WAVES>(a, y, av, cc, x, g, len, gi, segloss);
      Jold = row_combine<G, R, WAVES, true>(Jold, g, red, wave, lane);
      if (a.fixed_alpha > 0.0) {
        const double s = a.fixed_alpha / ((double)len + 1.0);
        Vec<G, R> xn;
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          xn.v[i].x = x.v[i].x + g.v[i].x * (-s);
          xn.v[i].y = x.v[i].y + g.v[i].y * (-s);
        }
        reg_prox<G, R>(rd, xn, s, j, a.k);
        if (gi == 0) {
#pragma unroll
          for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
        }
      } else {
        Jold += reg_eval<G, R>(rd, x, j, a.k);
        double alpha = a.alpha[seg];
        const double l = (double)len + 1.0;
        int ntrials = 0;
        bool accepted = false;
        while (alpha > a.min_stepsize) {
          const double s = alpha / l;
          Vec<G, R> xn, dummy;
#pragma unroll
          for (int i = 0; i < R / 2; ++i) {
            xn.v[i].x = fma(-s, g.v[i].x, x.v[i].x);
            xn.v[i].y = fma(-s, g.v[i].y, x.v[i].y);
          }
          reg_prox<G, R>(rd, xn, s, j, a.k);
          double Jn = reg_pass<G, R, LOSS, MAXT, false, WAVES>(a, y, av, cc, xn, dummy, len, gi, segloss);
          Jn = row_combine<G, R, WAVES, false>(Jn, dummy, red, wave, lane);
          Jn += reg_eval<G, R>(rd, xn, j, a.k);
          ++ntrials;
          if (Jn < Jold) {
            x = xn;
            alpha *= 1.05;
            Jold = Jn;
            accepted = true;
            break;
          }
          alpha *= .7;
          if (alpha < a.min_stepsize) {
            alpha = a.min_stepsize * 1.1;
            break;
          }
        }
        if (accepted && gi == 0) {
#pragma unroll
          for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = x.v[i];
        }
        if (tid == 0) {
          a.alpha[seg] = alpha;
          if (a.trials) {
            a.trials[seg] += ntrials;
            a.accepts[seg] += accepted ? 1 : 0;
          }
        }
      }
    }
    __syncthreads();      // everybody has read this row's list out of LDS (and is done with the combine buffer)
    store_list(pi, pv);   // hand the next row's list over
    __syncthreads();
    seg = seg_n; beg = beg_n; len = len_n;
    seg_n = seg_nn; beg_n = beg_nn; len_n = len_nn;
  }
}

template <int G, int R, int LOSS>
int launch_reg_inst(const CachedArgs& a, hipStream_t st, glrm_handle* h) { // a.cap = trips of one wave the longest row needs (64 / G observations each)
  // Two waves per row (each holds every other trip's vectors: half the registers, two waves per SIMD, so one wave's loads overlap the
  // other's arithmetic).  Measured at C4, X half-step: one wave per row 101.5 ms, two 85.4 ms, four 130.3 ms (phase-aligned passes 120.6).
  // ALWAYS two, also for rows one wave could hold: the wave count fixes the order of the sums, and it must not depend on the
  // longest row of the launch (MAXT only adds empty trips).  GLRM_HIP_CACHED_WAVES = 1 | 4 are the experiment switches.
  const int waves = env_int("GLRM_HIP_CACHED_WAVES", 2);
  if (waves == 2 && env_int("GLRM_HIP_CACHED_PERSIST", 1)) { // the persistent form of the two-wave kernel (same bits)
    // resident grid per handle (its device's CU count, its loss variant's occupancy, the fill percentage at its first sweep)
    const bool small = (a.cap + 1) / 2 <= 4;
    int& cache = h->cached_grid[small ? 1 : 0];
    int nb = cache;
    if (nb == 0) {
      int per_cu = 0;
      hipDeviceProp_t prop;
      const void* k = small ? (const void*)regcached_persist_kernel<G, R, LOSS, 4> : (const void*)regcached_persist_kernel<G, R, LOSS, 7>;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 128, 0) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; }
      int cus = 256;
      if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
      nb = (int)((int64_t)per_cu * cus * env_int("GLRM_HIP_CACHED_PERSIST_FILL", 100) / 100); // percent of the resident grid
      if (nb < 1) nb = 1;
      cache = nb;
    }
    const unsigned grid = (unsigned)std::min<int64_t>(a.nseg, nb);
    if (small) hipLaunchKernelGGL((regcached_persist_kernel<G, R, LOSS, 4>), dim3(grid), dim3(128), 0, st, a);
    else hipLaunchKernelGGL((regcached_persist_kernel<G, R, LOSS, 7>), dim3(grid), dim3(128), 0, st, a);
    return GLRM_OK;
  }
  if (waves == 4 && (a.cap + 3) / 4 <= 4) {
    hipLaunchKernelGGL((regcached_sweep_kernel<G, R, LOSS, 4, 4>), dim3((unsigned)a.nseg), dim3(256), 0, st, a);
  } else if (waves == 1) {
    if (a.cap <= 7) hipLaunchKernelGGL((regcached_sweep_kernel<G, R, LOSS, 7, 1>), dim3((unsigned)a.nseg), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((regcached_sweep_kernel<G, R, LOSS, 13, 1>), dim3((unsigned)a.nseg), dim3(64), 0, st, a);
  } else if ((a.cap + 1) / 2 <= 4) {
    hipLaunchKernelGGL((regcached_sweep_kernel<G, R, LOSS, 4, 2>), dim3((unsigned)a.nseg), dim3(128), 0, st, a);
  } else {
    hipLaunchKernelGGL((regcached_sweep_kernel<G, R, LOSS, 7, 2>), dim3((unsigned)a.nseg), dim3(128), 0, st, a);
  }
  return GLRM_OK;
}

template <int G, int R>
int launch_reg_layout(int loss, const CachedArgs& a, hipStream_t st, glrm_handle* h) {
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return launch_reg_inst<G, R, LOSS_QUAD_UNIFORM>(a, st, h);
    case LOSS_SEGMENT: return launch_reg_inst<G, R, LOSS_SEGMENT>(a, st, h);
    case LOSS_SEGMENT_NOTRIG: return launch_reg_inst<G, R, LOSS_SEGMENT_NOTRIG>(a, st, h);
    case LOSS_PER_OBS_NOTRIG: return launch_reg_inst<G, R, LOSS_PER_OBS_NOTRIG>(a, st, h);
    default: return launch_reg_inst<G, R, LOSS_PER_OBS>(a, st, h);
  }
}

template <int G, int R, int LOSS>
int launch_inst(const CachedArgs& a, hipStream_t st) {
  const int lds = a.cap * (G * R * 8 + 12);
  if (lds > 65536)
    HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(cached_sweep_kernel<G, R, LOSS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL((cached_sweep_kernel<G, R, LOSS>), dim3((unsigned)a.nseg), dim3(64), lds, st, a);
  return GLRM_OK;
}

template <int G, int R>
int launch_layout(int loss, const CachedArgs& a, hipStream_t st) {
  switch (loss) {
    case LOSS_QUAD_UNIFORM: return launch_inst<G, R, LOSS_QUAD_UNIFORM>(a, st);
    case LOSS_SEGMENT: return launch_inst<G, R, LOSS_SEGMENT>(a, st);
    case LOSS_SEGMENT_NOTRIG: return launch_inst<G, R, LOSS_SEGMENT_NOTRIG>(a, st);
    case LOSS_PER_OBS_NOTRIG: return launch_inst<G, R, LOSS_PER_OBS_NOTRIG>(a, st);
    default: return launch_inst<G, R, LOSS_PER_OBS>(a, st);
  }
}

} // namespace

// Does the WHOLE problem run its short rows on the cached sweep (glrm_handle::cached_want)?  Auto (GLRM_HIP_CACHED unset): the rows
// are not LDS-tiled, the opposing factor is beyond the sizes where the plain gathers already do well (> 32 MB) and the row view
// holds >= 1e8 observations -- all read from glrm_signature and (m, n, k), never from the shard.  WHICH rows it takes is a
// function of the row's own length (glrm_cached_maxlen): registers hold up to 13 trips of the lane layout (104 observations at
// k = 64, 208 at k <= 32); the LDS variant (GLRM_HIP_CACHED_REGS=0) up to its buffer.  Longer rows run the gather sweep with the
// waves their length asks for.  A row is therefore always summed in the same order, whatever shard holds it.
int glrm_setup_cached(glrm_handle* h) {
  h->cached_row = h->cached_want = 0;
  const int want = env_int("GLRM_HIP_CACHED", h->tiled_opt == 1 ? 0 : -1); // -1 auto, 0 off, 1 wherever the rows fit
  if (want == 0 || h->tiled_row || h->sig.nnz_rows <= 0) return GLRM_OK;
  if (!((h->G == 4 || h->G == 8) && h->R == 8)) return GLRM_OK;
  if (want < 0) {
    const double opp_bytes = (double)h->n * h->kp * 8;
    if (opp_bytes <= 32.0 * 1024 * 1024 || (double)h->sig.nnz_rows < 1e8) return GLRM_OK;
  }
  h->cached_want = 1;
  h->cached_row = env_int("GLRM_HIP_CACHED_REGS", 1) ? 2 : 1;
  return GLRM_OK;
}

// longest row the cached sweep takes
int64_t glrm_cached_maxlen(const glrm_handle* h) {
  if (h->cached_row == 2) return glrm_cached_reg_maxlen(h->G);
  const int vpi = 64 / (h->kp * 8 / 16);
  const int64_t budget = env_int("GLRM_HIP_CACHED", -1) > 0 ? 160 * 1024 : 53 * 1024; // auto: three waves per CU
  return budget / (h->kp * 8 + 12) / vpi * vpi;
}

// the launch parameter of the cached kernels for a longest row of `maxlen` observations
void glrm_cached_set_cap(glrm_handle* h, int64_t maxlen) {
  if (h->cached_row == 2) {
    const int ng = 64 / h->G;
    h->cached_cap = (int)((maxlen + ng - 1) / ng);
  } else {
    const int vpi = 64 / (h->kp * 8 / 16);
    h->cached_cap = (int)((maxlen + vpi - 1) / vpi * vpi);
    if (h->cached_cap < vpi) h->cached_cap = vpi;
  }
}

// seglist == nullptr: every local row (restricted to the range of glrm_hip_step_x_range, if one is set); otherwise the rows listed
int glrm_run_cached(glrm_handle* h, int loss, double min_stepsize, const int32_t* seglist, int64_t nlist, hipStream_t st) {
  CachedArgs a{};
  a.nseg = h->ml;
  a.ptr = h->rowptr;
  a.idx = h->colidx;
  a.vals = h->rowvals;
  a.own = h->X;
  a.own_offset = h->rb;
  a.other = h->Y;
  a.alpha = h->alpharow;
  a.losses = h->losses;
  a.regs = h->rx;
  a.reg_single = h->n_rx == 1;
  a.k = h->k;
  a.fixed_alpha = h->fixed_alpha;
  a.min_stepsize = min_stepsize;
  a.trials = h->trials_r;
  a.accepts = h->accepts_r;
  a.cap = h->cached_cap;
  a.seg_lo = 0;
  a.seg_hi = h->ml;
  if (seglist) {
    a.seglist = seglist;
    a.nseg = nlist;
    if (h->rng_e >= 0) { a.seg_lo = h->rng_b; a.seg_hi = h->rng_e; }
    if (a.nseg <= 0 || a.seg_hi <= a.seg_lo) return GLRM_OK;
  } else if (h->rng_e >= 0) { // glrm_hip_step_x_range: local rows [rng_b, rng_e)
    const int64_t s0 = h->rng_b;
    a.nseg = h->rng_e - s0;
    if (a.nseg <= 0) return GLRM_OK;
    a.ptr += s0; a.alpha += s0; a.own_offset += s0;
    if (!a.reg_single) a.regs += s0;
    a.trials += s0; a.accepts += s0;
  }
  int rc;
  if (h->cached_row == 2) rc = h->G == 4 ? launch_reg_layout<4, 8>(loss, a, st, h) : launch_reg_layout<8, 8>(loss, a, st, h);
  else rc = h->G == 4 ? launch_layout<4, 8>(loss, a, st) : launch_layout<8, 8>(loss, a, st);
  if (rc) return rc;
  HIPCK(hipGetLastError());
  return GLRM_OK;
}
