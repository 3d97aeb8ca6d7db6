// glrm_hip.hip -- libglrm_hip.so: the MI355X (gfx950 / CDNA4) engine behind
// LowRankModels.jl's fit!(glrm, ProxGradParams) (src/algorithms/proxgrad.jl:34-220).
//
// One HIP kernel per factor half-step ("sweep"):
//   X half-step = row sweep over CSR-by-row   (proxgrad.jl:118-156 + evaluate_fit.jl:24-38)
//   Y half-step = column sweep over CSC-by-col (proxgrad.jl:162-201 + evaluate_fit.jl:39-55)
// Both are the same kernel template: a *segment* (row or column) is owned by WAVES wavefronts;
// every observation of the segment is handled by a group of G lanes that reads the opposing
// factor's k-vector with one 16-byte load per lane (16*G contiguous bytes per group), reduces
// the dot product with DPP adds, evaluates loss and gradient, and accumulates the gradient in
// registers.  Gradient pass, backtracking line search (one more pass over the segment per trial)
// and the prox step are fused in the kernel; the accept/reject decision is wave/block uniform.
// Reduction order is fixed by the code => results are run-to-run deterministic and independent of
// how rows/columns are sharded over GPUs.
//
// No CPU fallback lives here; the CPU restatement (oracle/) is a separate, test-only library.

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "glrm_device.hpp"
#include "glrm_engine.hpp"

using namespace glrm;

// =============================================================================== kernels

struct SweepArgs {
  int64_t nseg;          // local segments
  const int64_t* ptr;    // nseg+1 offsets into idx/vals
  const int32_t* idx;    // index into the opposing factor (global id)
  const double* vals;    // A values
  double* own;           // factor being updated (global array, leading dimension KP)
  int64_t own_offset;    // global id of local segment 0
  const double* other;   // opposing factor (global array, leading dimension KP)
  double* alpha;         // per local segment step size
  double* obj;           // per GLOBAL segment objective (nullable)
  const glrm_loss* losses;
  int loss_by_segment;   // LOSS==1: 1 -> losses[own_offset+seg], 0 -> losses[0]
  const glrm_reg* regs;
  int reg_single;        // 1 -> regs[0], 0 -> regs[seg]
  int k;
  int eval_only;         // 1: obj[seg] = sum of losses (no regularizer), nothing else is written
  double fixed_alpha;    // > 0: one prox-gradient step with this global step size, no line search (SparseProxGradParams)
  double min_stepsize;
  int32_t* trials;       // per local segment accumulators (nullable)
  int32_t* accepts;
  const int32_t* seglist; // nullable: the launch covers the local segments seglist[0..nseg) instead of 0..nseg -- the segments of ONE
                          // wave class when the shard holds several (glrm_handle::seglist_r) -- restricted to [seg_lo, seg_hi)
  int64_t seg_lo, seg_hi; // (glrm_hip_step_x_range on such a shard; otherwise [0, local segments))
};


// One pass over the segment for one wave: J = sum of losses at u = <xv, other[idx]>, and (GRAD)
// g = sum of dL * other[idx].  Returns wave-level totals replicated in every lane.
// U observations per group are in flight per loop trip (U x R/2 16-byte loads per lane); a group
// always handles the observations t == gg (mod TG) in ascending order, so the result bits do not
// depend on U.
template <int G, int R, int WAVES, int LOSS, int U, bool GRAD>
__device__ __forceinline__ double sweep_pass(const SweepArgs& a, const Vec<G, R>& xv, Vec<G, R>& g, int64_t beg,
                                             int64_t len, int gg, int j, const LossDesc& segloss) {
  constexpr int KP = G * R, NG = 64 / G, TG = NG * WAVES;
  constexpr bool SCATTER = G == 4 && U == 4 && LOSS != LOSS_QUAD_UNIFORM;
  constexpr int LM = loss_mode(LOSS);
  constexpr bool TRIG = loss_trig(LOSS);
  double J = 0.0;
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  }
  const double2* __restrict__ other2 = reinterpret_cast<const double2*>(a.other);
  const int32_t* __restrict__ idx = a.idx + beg;
  const double* __restrict__ vals = a.vals + beg;
  // Software pipeline: the indices/values of trip t+1 are requested while trip t computes, so the
  // dependent chain per trip is only the factor gather.  The trip count is wave-uniform; lanes past
  // the end of the segment re-read its last entry and are masked by `valid`.
  int c[U];
  double av_next[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    c[u] = 0;
    av_next[u] = 0.0;
    if (len > 0) {
      int64_t tt = gg + (int64_t)u * TG;
      tt = tt < len ? tt : len - 1;
      c[u] = idx[tt];
      av_next[u] = vals[tt];
    }
  }
  for (int64_t t0 = 0; t0 < len; t0 += (int64_t)TG * U) {
    double2 y[U][R / 2];
    double av[U];
    int ccur[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      valid[u] = t0 + (int64_t)u * TG + gg < len;
      const double2* __restrict__ yp = other2 + (int64_t)c[u] * (KP / 2) + j;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) y[u][i] = yp[i * G];
      av[u] = av_next[u];
      ccur[u] = c[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t tn = t0 + (int64_t)(U + u) * TG + gg;
      tn = tn < len ? tn : len - 1;
      c[u] = idx[tn];
      av_next[u] = vals[tn];
    }
    if constexpr (SCATTER) {
      // Four observations per group and trip, one loss evaluation per LANE: the partial dot products are reduce-scattered in
      // two butterfly steps (the pairings of group_sum, hence its bits), lane u evaluates observation u, and the derivatives
      // come back by quad broadcasts.  The loop is wave-uniform, so every DPP source lane is active.
      double p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u] = 0.0;
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          p[u] = fma(xv.v[i].x, y[u][i].x, p[u]);
          p[u] = fma(xv.v[i].y, y[u][i].y, p[u]);
        }
      }
      const bool odd = (j & 1) != 0, hi2 = (j & 2) != 0;
      const double qa = (odd ? p[1] : p[0]) + dpp_f64<DPP_XOR1>(odd ? p[0] : p[1]);
      const double qb = (odd ? p[3] : p[2]) + dpp_f64<DPP_XOR1>(odd ? p[2] : p[3]);
      const double dot = (hi2 ? qb : qa) + dpp_f64<DPP_XOR2>(hi2 ? qa : qb); // observation u == j
      const double am = hi2 ? (odd ? av[3] : av[2]) : (odd ? av[1] : av[0]);
      const bool vm = hi2 ? (odd ? valid[3] : valid[2]) : (odd ? valid[1] : valid[0]);
      double L, dL;
      if constexpr (LM == LOSS_SEGMENT) {
        loss_both<GRAD, TRIG>(segloss, dot, am, L, dL);
      } else {
        const int cm = hi2 ? (odd ? ccur[3] : ccur[2]) : (odd ? ccur[1] : ccur[0]);
        const LossDesc lo = load_loss(a.losses, cm);
        loss_both<GRAD, TRIG>(lo, dot, am, L, dL);
      }
      if (!vm) {
        L = 0.0;
        dL = 0.0;
      }
      J += L; // lane-partial: summed over the group after the loop
      if (GRAD) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double d = group_bcast_f64<G>(dL, u, j);
#pragma unroll
          for (int i = 0; i < R / 2; ++i) {
            g.v[i].x = fma(d, y[u][i].x, g.v[i].x);
            g.v[i].y = fma(d, y[u][i].y, g.v[i].y);
          }
        }
      }
    } else {
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
        const LossDesc lo = load_loss(a.losses, ccur[u]);
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
  }
  if constexpr (SCATTER) J = group_sum<G>(J);
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

// Combine the per-wave totals of a multi-wave segment through LDS, in wave order, so that every
// thread of the block ends with the same bits.
template <int G, int R, int WAVES, bool GRAD>
__device__ __forceinline__ double block_combine(double J, Vec<G, R>& g, double* red, int wave, int lane) {
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

// EVAL = true is the one-pass objective evaluation (obj[seg] = sum of losses); it is a separate
// instantiation so that profiles list it apart from the two-pass half-step sweeps.
template <int G, int R, int WAVES, int LOSS, int U, bool EVAL>
__global__ void __launch_bounds__(WAVES == 1 ? 256 : WAVES * 64) sweep_kernel(const SweepArgs a) {
  constexpr int KP = G * R, NG = 64 / G;
  __shared__ __attribute__((aligned(16))) double red[WAVES == 1 ? 2 : WAVES * (KP + 2)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform -> SGPR
  const int64_t slot = WAVES == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (slot >= a.nseg) return; // wave-uniform (WAVES==1) or block-uniform
  int64_t seg = slot;
  if (a.seglist) {
    seg = (int64_t)a.seglist[slot];
    if (seg < a.seg_lo || seg >= a.seg_hi) return; // wave-uniform (WAVES==1) or block-uniform
  }
  const int j = lane % G, gi = lane / G;
  const int gg = (WAVES == 1 ? 0 : wave * NG) + gi;
  const int64_t beg = a.ptr[seg], len = a.ptr[seg + 1] - beg;
  const int64_t gseg = a.own_offset + seg;
  double2* ownp = reinterpret_cast<double2*>(a.own + gseg * KP);

  Vec<G, R> x, g;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) x.v[i] = ownp[i * G + j];
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : seg);
  LossDesc segloss;
  if constexpr (loss_mode(LOSS) != LOSS_PER_OBS) segloss = load_loss(a.losses, a.loss_by_segment ? gseg : 0);
  else segloss = LossDesc{0, 1.0, 0.0, 0.0};

  // pass 1: gradient + objective at the current point (proxgrad.jl:122-135 / :165-178)
  double Jold = sweep_pass<G, R, WAVES, LOSS, U, true>(a, x, g, beg, len, gg, j, segloss);
  Jold = block_combine<G, R, WAVES, true>(Jold, g, red, wave, lane);
  if constexpr (EVAL) {
    if (threadIdx.x == (WAVES == 1 ? wave * 64 : 0) && a.obj) a.obj[gseg] = Jold;
    return;
  }
  if (a.fixed_alpha > 0.0) { // src/algorithms/sparse_proxgrad.jl:72-77 / :94-99: g *= -alpha/l; x += g; prox!(r, x, alpha/l)
    const double s = a.fixed_alpha / ((double)len + 1.0);
    Vec<G, R> xn;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      xn.v[i].x = x.v[i].x + g.v[i].x * (-s);
      xn.v[i].y = x.v[i].y + g.v[i].y * (-s);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    if (wave == (WAVES == 1 ? wave : 0) && gi == 0) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
    }
    return;
  }
  Jold += reg_eval<G, R>(rd, x, j, a.k);

  // backtracking line search (proxgrad.jl:136-155 / :179-200); g is NOT recomputed between trials
  double alpha = a.alpha[seg];
  const double l = (double)len + 1.0;
  int ntrials = 0;
  bool accepted = false;
  while (alpha > a.min_stepsize) {
    const double s = alpha / l;
    Vec<G, R> xn, dummy;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) { // axpy!(-stepsize, g, newx)
      xn.v[i].x = fma(-s, g.v[i].x, x.v[i].x);
      xn.v[i].y = fma(-s, g.v[i].y, x.v[i].y);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k); // prox!(r, newx, stepsize)
    double Jn = sweep_pass<G, R, WAVES, LOSS, U, false>(a, xn, dummy, beg, len, gg, j, segloss);
    Jn = block_combine<G, R, WAVES, false>(Jn, dummy, red, wave, lane);
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

  if (accepted && wave == (WAVES == 1 ? wave : 0) && gi == 0) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = x.v[i];
  }
  if (lane == 0 && (WAVES == 1 || wave == 0)) {
    a.alpha[seg] = alpha;
    if (a.obj) a.obj[gseg] = Jold;
    if (a.trials) {
      a.trials[seg] += ntrials;
      a.accepts[seg] += accepted ? 1 : 0;
    }
  }
}

// evaluate(r, factor[:,seg]) for every local segment (calc_penalty, src/evaluate_fit.jl:91-104)
__global__ void penalty_kernel(const double* fac, int ld, int k, int64_t offset, int64_t nseg, const glrm_reg* regs,
                               int reg_single, double* out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const double* x = fac + (offset + s) * ld;
  const glrm_reg r = regs[reg_single ? 0 : s];
  double v = 0.0;
  switch (r.kind) {
    case GLRM_REG_QUAD: {
      double acc = 0.0;
      for (int c = 0; c < k; ++c) acc += x[c] * x[c];
      v = r.scale * acc;
      break;
    }
    case GLRM_REG_ONE: {
      double acc = 0.0;
      for (int c = 0; c < k; ++c) acc += fabs(x[c]);
      v = r.scale * acc;
      break;
    }
    case GLRM_REG_NONNEG:
      for (int c = 0; c < k; ++c)
        if (x[c] < 0) v = __builtin_inf();
      break;
    case GLRM_REG_UNIT_ONE_SPARSE: {
      int ones = 0, other = 0;
      for (int c = 0; c < k; ++c) {
        if (x[c] == 0) continue;
        if (x[c] == 1) ++ones; else ++other;
      }
      if (other > 0 || ones > 1) v = __builtin_inf();
      break;
    }
    default:
      break;
  }
  out[offset + s] = v;
}

// Fixed-shape two-stage sum: 256 blocks x 256 threads, strided partials, LDS tree, then one block.
// The shape never depends on the shard layout, so the recorded objective is G-invariant.
constexpr int SUM_BLOCKS = 256, SUM_THREADS = 256;

__device__ __forceinline__ double block_tree_sum(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = SUM_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  return sh[0];
}

__global__ void __launch_bounds__(SUM_THREADS) sum_stage1(const double* v, int64_t n, double* partials) {
  __shared__ double sh[SUM_THREADS];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * SUM_THREADS + threadIdx.x; i < n; i += (int64_t)SUM_BLOCKS * SUM_THREADS) acc += v[i];
  const double t = block_tree_sum(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ void __launch_bounds__(SUM_THREADS) sum_stage2(const double* partials, double* out) {
  __shared__ double sh[SUM_THREADS];
  const double t = block_tree_sum(partials[threadIdx.x], sh);
  if (threadIdx.x == 0) *out = t;
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void isum_kernel(const int32_t* v, int64_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += (unsigned long long)v[i];
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// Per-view statistics for glrm_signature and the class plan of the gather sweeps: out[0] = longest list, out[1..3] = segments per wave
// class (forced_cls > 0 pins the class), out[4] = segments the cached sweep takes (len <= cache_maxlen; cache_maxlen < 0: none; they are
// not counted in out[1..3]), out[5] = longest of those.
__global__ void seg_stats_kernel(const int64_t* ptr, int64_t nseg, int forced_cls, int64_t cache_maxlen, unsigned long long* out) {
  unsigned long long mx = 0, c1 = 0, c2 = 0, c3 = 0, cc = 0, mc = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t len = ptr[s + 1] - ptr[s];
    mx = (unsigned long long)len > mx ? (unsigned long long)len : mx;
    if (cache_maxlen >= 0 && len <= cache_maxlen) {
      ++cc;
      mc = (unsigned long long)len > mc ? (unsigned long long)len : mc;
    } else {
      const int cls = forced_cls > 0 ? forced_cls : glrm_wave_class(len);
      c1 += cls == 1; c2 += cls == 2; c3 += cls == 3;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    const unsigned long long o0 = __shfl_xor(mx, d, 64), o5 = __shfl_xor(mc, d, 64);
    mx = o0 > mx ? o0 : mx; mc = o5 > mc ? o5 : mc;
    c1 += __shfl_xor(c1, d, 64); c2 += __shfl_xor(c2, d, 64); c3 += __shfl_xor(c3, d, 64); cc += __shfl_xor(cc, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out + 0, mx); atomicMax(out + 5, mc);
    if (c1) atomicAdd(out + 1, c1);
    if (c2) atomicAdd(out + 2, c2);
    if (c3) atomicAdd(out + 3, c3);
    if (cc) atomicAdd(out + 4, cc);
  }
}

// =============================================================================== host side

thread_local char g_err[768];

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) ok = false;
    if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    int cur;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

extern "C" int glrm_hip_version(void) { return GLRM_HIP_ABI_VERSION; }
extern "C" const char* glrm_hip_last_error(void) { return g_err; }

// kp = G*R >= k.  Default layouts; GLRM_HIP_LANES_PER_OBS (tuning knob) selects another G for the same kp.
static int pick_layout(int k, int& G, int& R) {
  int kp;
  if (k <= 8) { kp = 8; G = 4; }
  else if (k <= 16) { kp = 16; G = 4; }
  else if (k <= 32) { kp = 32; G = 4; }
  else if (k <= 64) { kp = 64; G = 8; }
  else if (k <= 128) { kp = 128; G = 16; }
  else return -1;
  const int want = env_int("GLRM_HIP_LANES_PER_OBS", 0);
  if ((want == 4 || want == 8 || want == 16) && kp % want == 0) {
    const int r = kp / want;
    const bool have = (want == 4 && (r == 2 || r == 4 || r == 8)) || (want == 8 && (r == 4 || r == 8)) ||
                      (want == 16 && (r == 2 || r == 4 || r == 8));
    if (have) G = want;
  }
  R = kp / G;
  return 0;
}

template <typename T>
static int dev_copy_in(T** dst, const T* src, int64_t count, bool src_on_device, hipStream_t st) {
  *dst = nullptr;
  const size_t bytes = (size_t)(count > 0 ? count : 1) * sizeof(T);
  HIPCK(hipMalloc((void**)dst, bytes));
  if (count > 0)
    HIPCK(hipMemcpyAsync(*dst, src, (size_t)count * sizeof(T), src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
  return GLRM_OK;
}

static bool is_classification(int kind) { return kind == GLRM_LOSS_LOGISTIC || kind == GLRM_LOSS_WEIGHTED_HINGE; }

// Host-array validation: the checks of the GLRM constructor (src/glrm.jl:38-43,63-71) and of the
// Bool coercion (src/losses.jl:104-106), plus structural checks of the CSR / CSC arrays.
static int check_view(const char* name, int64_t nseg, const int64_t* ptr, const int32_t* idx, const double* vals,
                      int64_t bound, const glrm_problem* p, bool by_idx, int64_t seg_offset) {
  if (!ptr) return fail(GLRM_ERR_INVALID, "%s pointer array is NULL", name);
  if (ptr[0] != 0) return fail(GLRM_ERR_INVALID, "%s[0] must be 0", name);
  for (int64_t s = 0; s < nseg; ++s)
    if (ptr[s + 1] < ptr[s]) return fail(GLRM_ERR_INVALID, "%s is not monotone at %lld", name, (long long)s);
  if (ptr[nseg] > 0 && (!idx || !vals)) return fail(GLRM_ERR_INVALID, "%s index/value arrays are NULL", name);
  for (int64_t s = 0; s < nseg; ++s) {
    for (int64_t t = ptr[s]; t < ptr[s + 1]; ++t) {
      if (idx[t] < 0 || idx[t] >= bound)
        return fail(GLRM_ERR_INVALID, "%s: index %d out of range [0,%lld)", name, idx[t], (long long)bound);
      const int64_t e = by_idx ? seg_offset + s : idx[t], f = by_idx ? idx[t] : seg_offset + s;
      if (std::isnan(vals[t]))
        return fail(GLRM_ERR_NONFINITE, "Observed value in entry (%lld, %lld) is NaN.", (long long)e, (long long)f);
      const glrm_loss& l = p->n_losses == 1 ? p->losses[0] : p->losses[f];
      if (is_classification(l.kind) && !(vals[t] == 1.0 || vals[t] == 0.0))
        return fail(GLRM_ERR_NONFINITE, "entry in column %lld has label %g; a ClassificationLoss needs true(1)/false(0)",
                    (long long)f, vals[t]);
      if (l.kind >= GLRM_LOSS_MULTINOMIAL) { // levels 1..max index into u (BoundsError / InexactError in the reference)
        const int mx = (l.kind == GLRM_LOSS_BVS || l.kind == GLRM_LOSS_MULTINOMIAL_ORDINAL) ? l.dim + 1 : l.dim;
        if (!(vals[t] >= 1.0 && vals[t] <= (double)mx && vals[t] == std::floor(vals[t])))
          return fail(GLRM_ERR_NONFINITE, "entry (%lld, %lld) = %g is not a level in 1..%d of its categorical / ordinal loss",
                      (long long)e, (long long)f, vals[t], mx);
      }
    }
  }
  return GLRM_OK;
}

static bool wrap_ok(int w) {
  return w == 0 || w == GLRM_WRAP_LASTENTRY1 || w == GLRM_WRAP_LASTENTRY_UNPENALIZED || w == GLRM_WRAP_ORDINAL || w == GLRM_WRAP_MNL_ORDINAL;
}

static int check_desc(const glrm_problem* p) {
  const int64_t ml = p->row_end - p->row_begin, nl = p->col_end - p->col_begin;
  if (!p->losses || !(p->n_losses == 1 || p->n_losses == p->n))
    return fail(GLRM_ERR_INVALID, "There must be as many losses as there are columns in the data matrix (n_losses=%lld, n=%lld)",
                (long long)p->n_losses, (long long)p->n);
  if (!p->rx || !(p->n_rx == 1 || p->n_rx == ml))
    return fail(GLRM_ERR_INVALID, "There must be either one X regularizer or as many X regularizers as there are rows in the data matrix");
  if (!p->ry || !(p->n_ry == 1 || p->n_ry == nl))
    return fail(GLRM_ERR_INVALID, "There must be either one Y regularizer or as many Y regularizers as there are columns in the data matrix");
  for (int64_t i = 0; i < p->n_losses; ++i) {
    if (p->losses[i].kind < 0 || p->losses[i].kind >= GLRM_LOSS_KIND_COUNT)
      return fail(GLRM_ERR_UNSUPPORTED, "loss kind %d (column %lld) is not a supported loss", p->losses[i].kind, (long long)i);
    if (p->losses[i].kind < GLRM_LOSS_MULTINOMIAL ? !(p->losses[i].dim == 0 || p->losses[i].dim == 1)
                                                  : !(p->losses[i].dim >= 2 && p->losses[i].dim <= GLRM_MAX_EMBEDDING_DIM))
      return fail(GLRM_ERR_INVALID, "glrm_loss.dim = %d is not a valid embedding dimension for loss kind %d", p->losses[i].dim, p->losses[i].kind);
    if ((p->losses[i].kind == GLRM_LOSS_OVA || p->losses[i].kind == GLRM_LOSS_BVS) &&
        !(p->losses[i].p1 == GLRM_LOSS_LOGISTIC || p->losses[i].p1 == GLRM_LOSS_WEIGHTED_HINGE))
      return fail(GLRM_ERR_UNSUPPORTED, "bin_loss of OvALoss / BvSLoss must be LogisticLoss or HingeLoss");
  }
  for (int64_t i = 0; i < p->n_rx; ++i) {
    if (p->rx[i].kind < 0 || p->rx[i].kind >= GLRM_REG_KIND_COUNT) return fail(GLRM_ERR_UNSUPPORTED, "rx regularizer kind %d is not supported", p->rx[i].kind);
    if (!wrap_ok(p->rx[i].wrap)) return fail(GLRM_ERR_INVALID, "glrm_reg.wrap must be 0 or one GLRM_WRAP_* flag");
  }
  for (int64_t i = 0; i < p->n_ry; ++i) {
    if (p->ry[i].kind < 0 || p->ry[i].kind >= GLRM_REG_KIND_COUNT) return fail(GLRM_ERR_UNSUPPORTED, "ry regularizer kind %d is not supported", p->ry[i].kind);
    if (!wrap_ok(p->ry[i].wrap)) return fail(GLRM_ERR_INVALID, "glrm_reg.wrap must be 0 or one GLRM_WRAP_* flag");
  }
  return GLRM_OK;
}

extern "C" void glrm_hip_destroy(glrm_handle* h) {
  if (!h) return;
  DeviceGuard dg(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (!h->own_ptrs) h->rowptr = h->colptr = nullptr;        // borrowed arrays are the caller's
  if (!h->own_rowview) { h->colidx = nullptr; h->rowvals = nullptr; }
  if (!h->own_colview) { h->rowidx = nullptr; h->colvals = nullptr; }
  void* ptrs[] = {h->rowptr, h->colptr, h->colidx, h->rowidx, h->rowvals, h->colvals, h->losses, h->rx, h->ry,
                  h->alpharow, h->alphacol, h->oX, h->oY, h->oobjcol, h->oobjrow, h->partials, h->dscalar, h->dcount,
                  h->trials_r, h->accepts_r, h->trials_c, h->accepts_c, h->part, h->gsum, h->trialbuf, h->joldbuf,
                  h->activebuf, h->ntrialbuf, h->nactive, h->dflag, h->Arow, h->Acol, h->part_r, h->gsum_r, h->trial_r,
                  h->jold_r, h->active_r, h->ntrial_r, h->ystart, h->mtrial, h->mpart_loss, h->mpart_G, h->mgtot,
                  h->mobjold, h->mactive, h->mnactive, h->colperm, h->rowperm, h->seglist_r, h->seglist_c, h->rowdescid, h->udesc,
                  h->gramH, h->gram_part, h->jloss_r, h->jloss_c, h->lock_ctr, h->actlist, h->blk_perm_c, h->blk_long_c,
                  h->lane_bptr[0], h->lane_bptr[1], h->lane_off[0], h->lane_off[1], h->lane_val[0], h->lane_val[1],
                  h->lane_inv[0], h->lane_inv[1], h->lane_off16[0], h->lane_off16[1], h->lane_vptr[0], h->lane_vptr[1], h->lane_sval[0], h->lane_sval[1], h->lane_gcnt, h->lane_gbase, h->lane_gtotal, h->lane_glist};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->iter_exec) (void)hipGraphExecDestroy(h->iter_exec);
  if (h->iter_graph) (void)hipGraphDestroy(h->iter_graph);
  if (h->pinned_obj) (void)hipHostFree(h->pinned_obj);
  for (auto& e : h->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& e : h->pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int view_stats(glrm_handle* h, bool rows, int forced_cls, int64_t cache_maxlen, unsigned long long (&st)[6]) {
  for (auto& v : st) v = 0;
  const int64_t nseg = rows ? h->ml : h->nl;
  if (nseg <= 0) return GLRM_OK;
  unsigned long long* d = nullptr;
  HIPCK(hipMalloc((void**)&d, sizeof st));
  hipError_t e = hipMemsetAsync(d, 0, sizeof st, h->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(seg_stats_kernel, dim3(1024), dim3(256), 0, h->stream, rows ? h->rowptr : h->colptr, nseg, forced_cls, cache_maxlen, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(st, d, sizeof st, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(GLRM_ERR_HIP, "segment statistics failed: %s", hipGetErrorString(e));
  return GLRM_OK;
}

static int forced_class(int waves_opt) { return waves_opt == 1 ? 1 : (waves_opt == 4 ? 2 : (waves_opt == 8 ? 3 : 0)); }

// The class plan of a view that runs on the gather sweeps (and, for rows, the cached sweep): every segment is swept by the number of
// waves its OWN length asks for (glrm_wave_class; glrm_options.waves_* pins one count), rows short enough for the cached sweep by that
// one -- so a segment is summed in the same order whichever shard holds it.  One populated class: one launch over all local segments.
// Several (skewed lengths: a few very popular columns / very active rows next to many short ones): seglist = the segments class by
// class, one launch per class, the minority classes on a side stream beside the main launch.
static int build_class_plan(glrm_handle* h, bool rows) {
  int64_t* ncls = rows ? h->ncls_r : h->ncls_c;
  const int64_t nseg = rows ? h->ml : h->nl;
  const int forced = forced_class(rows ? h->opts.waves_row : h->opts.waves_col);
  const bool cached = rows && h->cached_row != 0;
  const int64_t cmax = cached ? glrm_cached_maxlen(h) : -1;
  unsigned long long st[6];
  int rc = view_stats(h, rows, forced, cmax, st);
  if (rc) return rc;
  ncls[0] = (int64_t)st[4]; ncls[1] = (int64_t)st[1]; ncls[2] = (int64_t)st[2]; ncls[3] = (int64_t)st[3];
  if (cached) {
    if (ncls[0] == 0) h->cached_row = 0; // this shard holds no row the cached sweep takes
    else glrm_cached_set_cap(h, (int64_t)st[5]);
  }
  int best = 1, populated = 0;
  for (int c = 0; c < 4; ++c) populated += ncls[c] > 0;
  for (int c = 2; c < 4; ++c) if (ncls[c] > ncls[best]) best = c;
  (rows ? h->waves_row : h->waves_col) = forced ? glrm_class_waves(forced) : glrm_class_waves(best);
  if (populated > 1 && !env_int("GLRM_HIP_SPLIT_LONG", 1)) { // experiment switch: ONE launch on the majority wave class (not shard-invariant)
    const int64_t all = ncls[0] + ncls[1] + ncls[2] + ncls[3];
    if (cached) h->cached_row = 0;
    for (int c = 0; c < 4; ++c) ncls[c] = 0;
    ncls[best] = all;
    populated = 1;
  }
  if (populated <= 1) return GLRM_OK;
  std::vector<int64_t> ptr((size_t)nseg + 1);
  HIPCK(hipMemcpyAsync(ptr.data(), rows ? h->rowptr : h->colptr, ((size_t)nseg + 1) * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  std::vector<int32_t> lst((size_t)nseg);
  int64_t pos[4] = {0, ncls[0], ncls[0] + ncls[1], ncls[0] + ncls[1] + ncls[2]};
  for (int64_t s = 0; s < nseg; ++s) {
    const int64_t len = ptr[s + 1] - ptr[s];
    const int c = (cmax >= 0 && len <= cmax) ? 0 : (forced ? forced : glrm_wave_class(len));
    lst[(size_t)pos[c]++] = (int32_t)s;
  }
  int32_t** dst = rows ? &h->seglist_r : &h->seglist_c;
  HIPCK(hipMalloc((void**)dst, (size_t)nseg * 4));
  HIPCK(hipMemcpyAsync(*dst, lst.data(), (size_t)nseg * 4, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream)); // lst is a local
  if (!h->side_stream) {
    HIPCK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
    HIPCK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  return GLRM_OK;
}

static int create_impl(glrm_handle* h, const glrm_problem* p, const glrm_options* o) {
  const bool on_dev = (p->flags & GLRM_PROBLEM_DEVICE_ARRAYS) != 0;
  h->m = p->m; h->n = p->n; h->k = p->k;
  h->rb = p->row_begin; h->re = p->row_end; h->cb = p->col_begin; h->ce = p->col_end;
  h->ml = h->re - h->rb; h->nl = h->ce - h->cb;
  pick_layout(p->k, h->G, h->R);
  h->kp = h->G * h->R;
  h->unroll_row = env_int("GLRM_HIP_UNROLL_ROW", 2) == 2 ? 2 : 1; // 2 observations per group in flight: -20 % on the L2-latency-bound row sweep
  h->unroll_col = env_int("GLRM_HIP_UNROLL_COL", 1) == 2 ? 2 : 1;
  h->unroll_long = env_int("GLRM_HIP_UNROLL_LONG", 8) >= 8 ? 8 : 1; // 8-wave segments: observations per lane group in flight (launch_sweep_loss)
  h->profile = o ? o->profile : 0;
  h->tiled_opt = o ? o->tiled : 0;
  h->sum_order_opt = o ? o->sum_order : 0;
  if (o && (o->reserved != 0 || o->reserved0 != 0)) return fail(GLRM_ERR_INVALID, "glrm_options.reserved0 / reserved must be 0");
  if (o && (o->sum_order < 0 || o->sum_order > 1)) return fail(GLRM_ERR_INVALID, "glrm_options.sum_order must be 0 (engine orders) or 1 (reference order)");
  if (o) h->opts = *o; else { h->opts = glrm_options{}; h->opts.device_id = -1; }
  h->losses_h.assign(p->losses, p->losses + p->n_losses);
  h->rx_h.assign(p->rx, p->rx + p->n_rx);
  h->ry_h.assign(p->ry, p->ry + p->n_ry);
  if (o && (o->stream || o->caller_stream)) {
    h->stream = (hipStream_t)o->stream; // may be NULL = the legacy default stream (caller_stream)
  } else {
    HIPCK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  hipStream_t st = h->stream;
  int rc;
  const bool borrow = on_dev && (p->flags & GLRM_PROBLEM_BORROW_DEVICE_ARRAYS) != 0;
  if (!p->dense_A && borrow) { // the caller's device arrays, read in place
    h->own_ptrs = h->own_rowview = h->own_colview = false;
    h->rowptr = const_cast<int64_t*>(p->rowptr); h->colptr = const_cast<int64_t*>(p->colptr);
    h->colidx = const_cast<int32_t*>(p->colidx); h->rowvals = const_cast<double*>(p->rowvals);
    h->rowidx = const_cast<int32_t*>(p->rowidx); h->colvals = const_cast<double*>(p->colvals);
    HIPCK(hipMemcpyAsync(&h->nnz_r, p->rowptr + h->ml, 8, hipMemcpyDeviceToHost, st));
    HIPCK(hipMemcpyAsync(&h->nnz_c, p->colptr + h->nl, 8, hipMemcpyDeviceToHost, st));
    HIPCK(hipStreamSynchronize(st));
    if ((h->nnz_r > 0 && (!h->colidx || !h->rowvals)) || (h->nnz_c > 0 && (!h->rowidx || !h->colvals)))
      return fail(GLRM_ERR_INVALID, "index / value arrays are NULL");
  } else if (!p->dense_A && (p->flags & GLRM_PROBLEM_ROWS_FROM_COLS)) { // the column view only; the row view is derived on the device
    if ((rc = dev_copy_in(&h->colptr, p->colptr, h->nl + 1, on_dev, st))) return rc;
    if (on_dev) {
      HIPCK(hipMemcpyAsync(&h->nnz_c, p->colptr + h->nl, 8, hipMemcpyDeviceToHost, st));
      HIPCK(hipStreamSynchronize(st));
    } else {
      h->nnz_c = p->colptr[h->nl];
    }
    if (h->nnz_c > 0 && (!p->rowidx || !p->colvals)) return fail(GLRM_ERR_INVALID, "index / value arrays are NULL");
    if ((rc = dev_copy_in(&h->rowidx, p->rowidx, h->nnz_c, on_dev, st))) return rc;
    if ((rc = dev_copy_in(&h->colvals, p->colvals, h->nnz_c, on_dev, st))) return rc;
    if ((rc = glrm_rows_from_cols(h))) return rc;
  } else if (!p->dense_A) {
    if ((rc = dev_copy_in(&h->rowptr, p->rowptr, h->ml + 1, on_dev, st))) return rc;
    if ((rc = dev_copy_in(&h->colptr, p->colptr, h->nl + 1, on_dev, st))) return rc;
    if (on_dev) {
      HIPCK(hipMemcpyAsync(&h->nnz_r, p->rowptr + h->ml, 8, hipMemcpyDeviceToHost, st));
      HIPCK(hipMemcpyAsync(&h->nnz_c, p->colptr + h->nl, 8, hipMemcpyDeviceToHost, st));
      HIPCK(hipStreamSynchronize(st));
    } else {
      h->nnz_r = p->rowptr[h->ml];
      h->nnz_c = p->colptr[h->nl];
    }
    if ((rc = dev_copy_in(&h->colidx, p->colidx, h->nnz_r, on_dev, st))) return rc;
    if ((rc = dev_copy_in(&h->rowvals, p->rowvals, h->nnz_r, on_dev, st))) return rc;
    if ((rc = dev_copy_in(&h->rowidx, p->rowidx, h->nnz_c, on_dev, st))) return rc;
    if ((rc = dev_copy_in(&h->colvals, p->colvals, h->nnz_c, on_dev, st))) return rc;
  }
  h->n_losses = p->n_losses; h->n_rx = p->n_rx; h->n_ry = p->n_ry;
  if ((rc = dev_copy_in(&h->losses, p->losses, p->n_losses, false, st))) return rc;
  if ((rc = dev_copy_in(&h->rx, p->rx, p->n_rx, false, st))) return rc;
  if ((rc = dev_copy_in(&h->ry, p->ry, p->n_ry, false, st))) return rc;
  h->loss_quad_uniform = p->n_losses == 1 && p->losses[0].kind == GLRM_LOSS_QUAD;
  h->has_trig = false;
  for (int64_t i = 0; i < p->n_losses; ++i) h->has_trig = h->has_trig || p->losses[i].kind == GLRM_LOSS_PERIODIC;
  const int64_t ml1 = h->ml > 0 ? h->ml : 1, nl1 = h->nl > 0 ? h->nl : 1;
  HIPCK(hipMalloc((void**)&h->alpharow, ml1 * 8));
  HIPCK(hipMalloc((void**)&h->alphacol, nl1 * 8));
  HIPCK(hipMalloc((void**)&h->partials, SUM_BLOCKS * 8));
  HIPCK(hipMalloc((void**)&h->dscalar, 8));
  HIPCK(hipMalloc((void**)&h->dcount, 8));
  HIPCK(hipMalloc((void**)&h->trials_r, ml1 * 4));
  HIPCK(hipMalloc((void**)&h->accepts_r, ml1 * 4));
  HIPCK(hipMalloc((void**)&h->trials_c, nl1 * 4));
  HIPCK(hipMalloc((void**)&h->accepts_c, nl1 * 4));
  HIPCK(hipMemsetAsync(h->trials_r, 0, ml1 * 4, st));
  HIPCK(hipMemsetAsync(h->accepts_r, 0, ml1 * 4, st));
  HIPCK(hipMemsetAsync(h->trials_c, 0, nl1 * 4, st));
  HIPCK(hipMemsetAsync(h->accepts_c, 0, nl1 * 4, st));
  int rc2 = glrm_setup_multi(h, p);
  if (rc2) return rc2;
  // this shard's contribution to the signature of the whole problem (include/glrm_hip.h: glrm_signature)
  h->sig_local = glrm_signature{};
  if (p->dense_A) {
    h->sig_local.nnz_rows = h->ml * h->n; h->sig_local.nnz_cols = h->nl * h->m;
    h->sig_local.max_row_len = h->ml > 0 ? h->n : 0; h->sig_local.max_col_len = h->nl > 0 ? h->m : 0;
    if ((rc2 = glrm_setup_dense(h, p))) return rc2;
  } else {
    unsigned long long st6[6];
    if ((rc = view_stats(h, true, 0, -1, st6))) return rc;
    h->sig_local.max_row_len = (int64_t)st6[0];
    if ((rc = view_stats(h, false, 0, -1, st6))) return rc;
    h->sig_local.max_col_len = (int64_t)st6[0];
    h->sig_local.nnz_rows = h->nnz_r; h->sig_local.nnz_cols = h->nnz_c;
    if (!h->multi && (rc2 = glrm_prepare_tiled(h))) return rc2;
  }
  HIPCK(hipStreamSynchronize(st)); // host descriptor / index arrays may be released by the caller now
  return GLRM_OK;
}

// Second half of create: kernel families and their buffers, chosen from the signature of the WHOLE problem.
static int finalize_impl(glrm_handle* h, const glrm_signature* whole) {
  h->sig = whole ? *whole : h->sig_local;
  if (h->sig.nnz_rows < h->sig_local.nnz_rows || h->sig.nnz_cols < h->sig_local.nnz_cols || h->sig.max_row_len < h->sig_local.max_row_len ||
      h->sig.max_col_len < h->sig_local.max_col_len || h->sig.rows_unordered < h->sig_local.rows_unordered ||
      h->sig.cols_unordered < h->sig_local.cols_unordered)
    return fail(GLRM_ERR_INVALID, "the signature of the whole problem cannot be smaller than this shard's (sum the counts, max the rest)");
  int rc;
  if ((rc = glrm_setup_reforder(h))) return rc; // glrm_options.sum_order = 1: refuses models the validation sweeps do not cover
  if (h->sum_order_opt) {
    // reference-order validation sweeps: one lane per segment in list order, no family to choose and no view to re-order
    h->tiled_row = h->tiled_col = h->blocked_row = h->blocked_col = h->cached_row = h->cached_want = 0;
    h->waves_row = h->waves_col = 1;
  } else if (!h->multi && !h->dense) {
    // a failure from here on leaves re-ordered private views and partial buffers behind: the handle can then only be destroyed
    h->finalize_failed = true;
    if (glrm_test_fail_finalize())  // csrc/glrm_testhooks.hip: always 0 in the product library; the test build injects a failure on request
      return fail(GLRM_ERR_OOM, "injected set-up failure (test build hook)");
    if ((rc = glrm_setup_tiled(h))) return rc;
    if ((rc = glrm_setup_cached(h))) return rc;
    if ((rc = glrm_setup_blocked(h))) return rc;
    if (!h->tiled_row && !h->blocked_row && (rc = build_class_plan(h, true))) return rc;
    if (!h->tiled_col && !h->blocked_col && (rc = build_class_plan(h, false))) return rc;
    h->finalize_failed = false;
  } else {
    h->waves_row = h->opts.waves_row ? h->opts.waves_row : 1;
    h->waves_col = h->opts.waves_col ? h->opts.waves_col : 4;
  }
  HIPCK(hipStreamSynchronize(h->stream));
  h->finalized = true;
  return GLRM_OK;
}

extern "C" int glrm_hip_create(glrm_handle** out, const glrm_problem* p, const glrm_options* o) {
  if (!out || !p) return fail(GLRM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (p->m <= 0 || p->n <= 0 || p->k <= 0) return fail(GLRM_ERR_INVALID, "m, n, k must be positive");
  if (p->m > INT32_MAX || p->n > INT32_MAX) return fail(GLRM_ERR_UNSUPPORTED, "m, n must fit int32 indices");
  int G, R;
  if (pick_layout(p->k, G, R)) return fail(GLRM_ERR_UNSUPPORTED, "rank k=%d is above the engine's limit of 128", p->k);
  if (p->row_begin < 0 || p->row_end > p->m || p->row_begin > p->row_end || p->col_begin < 0 || p->col_end > p->n ||
      p->col_begin > p->col_end)
    return fail(GLRM_ERR_INVALID, "shard ranges out of bounds");
  int rc = check_desc(p);
  if (rc) return rc;
  if ((p->flags & GLRM_PROBLEM_BORROW_DEVICE_ARRAYS) && (!(p->flags & GLRM_PROBLEM_DEVICE_ARRAYS) || p->dense_A))
    return fail(GLRM_ERR_INVALID, "GLRM_PROBLEM_BORROW_DEVICE_ARRAYS needs GLRM_PROBLEM_DEVICE_ARRAYS and observation lists (not dense_A)");
  const bool from_cols = (p->flags & GLRM_PROBLEM_ROWS_FROM_COLS) != 0;
  if (from_cols) {
    if (p->dense_A || (p->flags & GLRM_PROBLEM_BORROW_DEVICE_ARRAYS)) return fail(GLRM_ERR_INVALID, "GLRM_PROBLEM_ROWS_FROM_COLS: list problems, copied (not borrowed) arrays");
    if (p->rowptr || p->colidx || p->rowvals) return fail(GLRM_ERR_INVALID, "GLRM_PROBLEM_ROWS_FROM_COLS: rowptr / colidx / rowvals must be NULL (the engine derives the row view)");
    if (!(p->row_begin == 0 && p->row_end == p->m && p->col_begin == 0 && p->col_end == p->n))
      return fail(GLRM_ERR_INVALID, "GLRM_PROBLEM_ROWS_FROM_COLS needs the whole problem (a shard's rows meet every column)");
    if (!p->colptr) return fail(GLRM_ERR_INVALID, "colptr is NULL");
  }
  if (p->dense_A) {
    // validated in glrm_setup_dense
  } else if (!(p->flags & GLRM_PROBLEM_DEVICE_ARRAYS)) {
    rc = from_cols ? GLRM_OK : check_view("rowptr", p->row_end - p->row_begin, p->rowptr, p->colidx, p->rowvals, p->n, p, true, p->row_begin);
    if (rc) return rc;
    rc = check_view("colptr", p->col_end - p->col_begin, p->colptr, p->rowidx, p->colvals, p->m, p, false, p->col_begin);
    if (rc) return rc;
  } else if ((!from_cols && !p->rowptr) || !p->colptr) {
    return fail(GLRM_ERR_INVALID, "rowptr / colptr are NULL");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(GLRM_ERR_HIP, "no HIP device is visible (this engine has no CPU fallback)");
  int dev = o ? o->device_id : -1;
  if (dev < 0) HIPCK(hipGetDevice(&dev));
  if (dev >= ndev) return fail(GLRM_ERR_INVALID, "device_id %d out of range (%d devices)", dev, ndev);
  DeviceGuard dg(dev);
  if (!dg.ok) return fail(GLRM_ERR_HIP, "cannot select device %d", dev);
  glrm_handle* h = new (std::nothrow) glrm_handle();
  if (!h) return fail(GLRM_ERR_OOM, "out of host memory");
  h->device = dev;
  rc = create_impl(h, p, o);
  if (!rc && !(p->flags & GLRM_PROBLEM_DEFER_SETUP)) rc = finalize_impl(h, nullptr);
  if (rc) {
    char keep[sizeof g_err];
    memcpy(keep, g_err, sizeof keep);
    glrm_hip_destroy(h);
    memcpy(g_err, keep, sizeof keep);
    return rc;
  }
  *out = h;
  return GLRM_OK;
}

extern "C" int glrm_hip_signature(glrm_handle* h, glrm_signature* local) {
  if (!h || !local) return fail(GLRM_ERR_INVALID, "NULL argument");
  *local = h->sig_local;
  return GLRM_OK;
}

extern "C" int glrm_hip_finalize(glrm_handle* h, const glrm_signature* whole) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  if (h->finalized) return fail(GLRM_ERR_INVALID, "the handle is already set up (glrm_hip_finalize follows a GLRM_PROBLEM_DEFER_SETUP create, once)");
  if (h->finalize_failed)  // a set-up that failed half way (out of memory in a family's buffers) has re-ordered private views and partial buffers
    return fail(GLRM_ERR_INVALID, "an earlier glrm_hip_finalize on this handle failed: destroy it and create the shard again");
  DeviceGuard dg(h->device);
  return finalize_impl(h, whole);
}

#define GLRM_NEED_FINALIZED(h) \
  do { if (!(h)->finalized) return fail(GLRM_ERR_INVALID, "the handle was created with GLRM_PROBLEM_DEFER_SETUP: call glrm_hip_finalize first"); } while (0)

// ------------------------------------------------------------------ buffers and factors

static int ensure_owned(glrm_handle* h) {
  if (!h->X) {
    if (!h->oX) { HIPCK(hipMalloc((void**)&h->oX, (size_t)h->kp * h->m * 8)); HIPCK(hipMemsetAsync(h->oX, 0, (size_t)h->kp * h->m * 8, h->stream)); }
    h->X = h->oX;
  }
  if (!h->Y) {
    if (!h->oY) { HIPCK(hipMalloc((void**)&h->oY, (size_t)h->kp * h->d * 8)); HIPCK(hipMemsetAsync(h->oY, 0, (size_t)h->kp * h->d * 8, h->stream)); }
    h->Y = h->oY;
  }
  if (!h->objcol) {
    if (!h->oobjcol) { HIPCK(hipMalloc((void**)&h->oobjcol, (size_t)h->n * 8)); HIPCK(hipMemsetAsync(h->oobjcol, 0, (size_t)h->n * 8, h->stream)); }
    h->objcol = h->oobjcol;
  }
  if (!h->objrow) {
    if (!h->oobjrow) { HIPCK(hipMalloc((void**)&h->oobjrow, (size_t)h->m * 8)); HIPCK(hipMemsetAsync(h->oobjrow, 0, (size_t)h->m * 8, h->stream)); }
    h->objrow = h->oobjrow;
  }
  return GLRM_OK;
}

extern "C" int glrm_hip_factor_ld(glrm_handle* h) { return h ? h->kp : fail(GLRM_ERR_INVALID, "NULL handle"); }

extern "C" int glrm_hip_bind_buffers(glrm_handle* h, void* dX, void* dY, void* dObjCol, void* dObjRow) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  h->X = (double*)dX; h->Y = (double*)dY; h->objcol = (double*)dObjCol; h->objrow = (double*)dObjRow;
  if (h->iter_exec) { (void)hipGraphExecDestroy(h->iter_exec); h->iter_exec = nullptr; } // the captured launches hold the old pointers
  return ensure_owned(h);
}

extern "C" int glrm_hip_set_factors(glrm_handle* h, const double* X, const double* Y) {
  if (!h || !X || !Y) return fail(GLRM_ERR_INVALID, "NULL argument");
  DeviceGuard dg(h->device);
  int rc = ensure_owned(h);
  if (rc) return rc;
  const size_t kb = (size_t)h->k * 8, pb = (size_t)h->kp * 8;
  if (h->kp != h->k) {
    HIPCK(hipMemsetAsync(h->X, 0, pb * h->m, h->stream));
    HIPCK(hipMemsetAsync(h->Y, 0, pb * h->d, h->stream));
  }
  HIPCK(hipMemcpy2DAsync(h->X, pb, X, kb, kb, (size_t)h->m, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpy2DAsync(h->Y, pb, Y, kb, kb, (size_t)h->d, hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return GLRM_OK;
}

extern "C" int glrm_hip_get_factors(glrm_handle* h, double* X, double* Y) {
  if (!h || !X || !Y) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (!h->X || !h->Y) return fail(GLRM_ERR_INVALID, "no factors on the device yet");
  DeviceGuard dg(h->device);
  const size_t kb = (size_t)h->k * 8, pb = (size_t)h->kp * 8;
  HIPCK(hipMemcpy2DAsync(X, kb, h->X, pb, kb, (size_t)h->m, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipMemcpy2DAsync(Y, kb, h->Y, pb, kb, (size_t)h->d, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return GLRM_OK;
}

extern "C" int glrm_hip_reset_stepsizes(glrm_handle* h, double stepsize) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  if (h->ml > 0) hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((h->ml + 255) / 256)), dim3(256), 0, h->stream, h->alpharow, h->ml, stepsize);
  if (h->nl > 0) hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((h->nl + 255) / 256)), dim3(256), 0, h->stream, h->alphacol, h->nl, stepsize);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

extern "C" int glrm_hip_set_regularizers(glrm_handle* h, const glrm_reg* rx, int64_t n_rx, const glrm_reg* ry, int64_t n_ry) {
  if (!h || !rx || !ry) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (n_rx != h->n_rx || n_ry != h->n_ry)
    return fail(GLRM_ERR_INVALID, "regularizer counts must match the handle (rx %lld, ry %lld)", (long long)h->n_rx, (long long)h->n_ry);
  for (int64_t i = 0; i < n_rx; ++i)
    if (rx[i].kind < 0 || rx[i].kind >= GLRM_REG_KIND_COUNT || !wrap_ok(rx[i].wrap)) return fail(GLRM_ERR_UNSUPPORTED, "rx regularizer kind %d is not supported", rx[i].kind);
  for (int64_t i = 0; i < n_ry; ++i)
    if (ry[i].kind < 0 || ry[i].kind >= GLRM_REG_KIND_COUNT || !wrap_ok(ry[i].wrap)) return fail(GLRM_ERR_UNSUPPORTED, "ry regularizer kind %d is not supported", ry[i].kind);
  if (!h->multi) { // a handle created on the scalar fast paths moves to the general sweeps when a wrapper appears
    bool wrapped = false;
    for (int64_t i = 0; i < n_rx; ++i) wrapped |= rx[i].wrap != 0;
    for (int64_t i = 0; i < n_ry; ++i) wrapped |= ry[i].wrap != 0;
    if (wrapped) {
      if (h->dense || h->kp > 64) return fail(GLRM_ERR_UNSUPPORTED, "wrapped regularizers need a sparse-view handle with k <= 64");
      h->multi = true;
    }
  }
  DeviceGuard dg(h->device);
  h->rx_h.assign(rx, rx + n_rx);
  h->ry_h.assign(ry, ry + n_ry);
  HIPCK(hipMemcpyAsync(h->rx, rx, (size_t)n_rx * sizeof(glrm_reg), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemcpyAsync(h->ry, ry, (size_t)n_ry * sizeof(glrm_reg), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return GLRM_OK;
}

extern "C" int glrm_hip_synchronize(glrm_handle* h) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  HIPCK(hipStreamSynchronize(h->stream));
  return GLRM_OK;
}

// ------------------------------------------------------------------ sweep launch

template <int G, int R, int WAVES>
static void launch_sweep_loss(int loss, int unroll, const SweepArgs& a, hipStream_t st) {
  const unsigned grid = (unsigned)(WAVES == 1 ? (a.nseg + 3) / 4 : a.nseg);
  const dim3 block(WAVES == 1 ? 256 : WAVES * 64);
#define GLRM_LAUNCH(LOSSV, UV)                                                                                  \
  do {                                                                                                          \
    if (a.eval_only) hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSSV, 1, true>), dim3(grid), block, 0, st, a); \
    else hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSSV, UV, false>), dim3(grid), block, 0, st, a);          \
  } while (0)
  if constexpr (WAVES == 8) {
    // One 8-wave workgroup per very long segment (the diverted columns of a power-law view: 880 000 observations at the C2-Zipf recipe) is
    // bound by the latency of its factor gathers: with one observation per lane group in flight the longest column alone took 13.7 ms of a
    // 15.7 ms Y half-step (profiles/r06_c2_zipf_kernel_stats.csv).  Eight per group in flight (a group still adds its observations
    // t == gg (mod TG) in ascending order: the bits do not depend on U) -- session r6_26.
    if (unroll >= 8 && !a.eval_only) {
      switch (loss) {
        case LOSS_QUAD_UNIFORM: hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSS_QUAD_UNIFORM, 8, false>), dim3(grid), block, 0, st, a); break;
        case LOSS_SEGMENT: hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSS_SEGMENT, 8, false>), dim3(grid), block, 0, st, a); break;
        case LOSS_SEGMENT_NOTRIG: hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSS_SEGMENT_NOTRIG, 8, false>), dim3(grid), block, 0, st, a); break;
        case LOSS_PER_OBS_NOTRIG: hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSS_PER_OBS_NOTRIG, 8, false>), dim3(grid), block, 0, st, a); break;
        default: hipLaunchKernelGGL((sweep_kernel<G, R, WAVES, LOSS_PER_OBS, 8, false>), dim3(grid), block, 0, st, a); break;
      }
      return;
    }
  }
  switch (loss) {
    case LOSS_QUAD_UNIFORM:
      if (unroll == 2) GLRM_LAUNCH(LOSS_QUAD_UNIFORM, 2);
      else GLRM_LAUNCH(LOSS_QUAD_UNIFORM, 1);
      break;
    // G == 4: four observations per trip, one loss evaluation per lane (C5-family row sweep 169 -> 115 ms); the multi-wave
    // sweeps of long same-loss segments are bound by the factor gather and keep the leaner one-observation body
    case LOSS_SEGMENT: GLRM_LAUNCH(LOSS_SEGMENT, (G == 4 && WAVES == 1 ? 4 : 1)); break;
    case LOSS_SEGMENT_NOTRIG: GLRM_LAUNCH(LOSS_SEGMENT_NOTRIG, (G == 4 && WAVES == 1 ? 4 : 1)); break;
    case LOSS_PER_OBS_NOTRIG: GLRM_LAUNCH(LOSS_PER_OBS_NOTRIG, (G == 4 ? 4 : 1)); break;
    default: GLRM_LAUNCH(LOSS_PER_OBS, (G == 4 ? 4 : 1)); break;
  }
#undef GLRM_LAUNCH
}

template <int G, int R>
static void launch_sweep_waves(int waves, int loss, int unroll, const SweepArgs& a, hipStream_t st) {
  switch (waves) {
    case 1: launch_sweep_loss<G, R, 1>(loss, unroll, a, st); break;
    case 4: launch_sweep_loss<G, R, 4>(loss, unroll, a, st); break;
    default: launch_sweep_loss<G, R, 8>(loss, unroll, a, st); break;
  }
}

// (lanes per observation G, components per lane R) with G*R == kp
static void launch_sweep(int G, int R, int waves, int loss, int unroll, const SweepArgs& a, hipStream_t st) {
  switch (G * 100 + R) {
    case 402: launch_sweep_waves<4, 2>(waves, loss, unroll, a, st); break;
    case 404: launch_sweep_waves<4, 4>(waves, loss, unroll, a, st); break;
    case 408: launch_sweep_waves<4, 8>(waves, loss, unroll, a, st); break;
    case 804: launch_sweep_waves<8, 4>(waves, loss, unroll, a, st); break;
    case 1602: launch_sweep_waves<16, 2>(waves, loss, unroll, a, st); break;
    case 808: launch_sweep_waves<8, 8>(waves, loss, unroll, a, st); break;
    case 1604: launch_sweep_waves<16, 4>(waves, loss, unroll, a, st); break;
    default: launch_sweep_waves<16, 8>(waves, loss, unroll, a, st); break;
  }
}

static int drain_events(glrm_handle* h) {
  for (auto& e : h->pending) {
    HIPCK(hipEventSynchronize(e.b));
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, e.a, e.b));
    (e.which == 0 ? h->ms_x : e.which == 1 ? h->ms_y : h->ms_wait) += ms;
    h->pool.push_back(e);
  }
  h->pending.clear();
  return GLRM_OK;
}

// roctx ranges around every half-step (rocprofv3 --marker-trace shows "glrm step_x" / "glrm step_y" / "glrm col_losses" on the host
// timeline).  libroctx64 is looked up at run time so that the engine has no link-time dependency on the tracer; GLRM_HIP_ROCTX=0
// switches the ranges off.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    if (!env_int("GLRM_HIP_ROCTX", 1)) return;
    void* lib = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL);
    if (!lib) lib = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_LOCAL);
    if (!lib) return;
    push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
    pop = (int (*)())dlsym(lib, "roctxRangePop");
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
struct RoctxRange {
  static RoctxApi& api() { static RoctxApi a; return a; }
  explicit RoctxRange(const char* name) { if (api().push) api().push(name); }
  ~RoctxRange() { if (api().pop) api().pop(); }
};

// which: 0 = row sweep (X half-step), 1 = column sweep (Y half-step)
static int run_sweep(glrm_handle* h, int which, double min_stepsize, int eval_only) {
  RoctxRange range(eval_only ? "glrm col_losses" : (which == 0 ? "glrm step_x" : "glrm step_y"));
  GLRM_NEED_FINALIZED(h);
  int rc = ensure_owned(h);
  if (rc) return rc;
  SweepArgs a{};
  const bool rows = which == 0;
  a.nseg = rows ? h->ml : h->nl;
  if (a.nseg <= 0) return GLRM_OK;
  a.ptr = rows ? h->rowptr : h->colptr;
  a.idx = rows ? h->colidx : h->rowidx;
  a.vals = rows ? h->rowvals : h->colvals;
  a.own = rows ? h->X : h->Y;
  a.own_offset = rows ? h->rb : h->cb;
  a.other = rows ? h->Y : h->X;
  a.alpha = rows ? h->alpharow : h->alphacol;
  a.obj = rows ? nullptr : h->objcol;
  a.losses = h->losses;
  a.regs = rows ? h->rx : h->ry;
  a.reg_single = (rows ? h->n_rx : h->n_ry) == 1;
  a.k = h->k;
  a.eval_only = eval_only;
  a.fixed_alpha = eval_only ? 0.0 : h->fixed_alpha;
  a.min_stepsize = min_stepsize;
  a.trials = eval_only ? nullptr : (rows ? h->trials_r : h->trials_c);
  a.accepts = rows ? h->accepts_r : h->accepts_c;
  a.seg_lo = 0;
  a.seg_hi = a.nseg;
  if (rows && h->rng_e >= 0) { // glrm_hip_step_x_range: local rows [rng_b, rng_e)
    const int64_t s0 = h->rng_b;
    if (h->rng_e - s0 <= 0) return GLRM_OK;
    if (h->seglist_r) { // several classes: the class launches filter their lists by the range
      a.seg_lo = s0; a.seg_hi = h->rng_e;
    } else {
      a.nseg = h->rng_e - s0;
      a.seg_hi = a.nseg;
      a.ptr += s0; a.alpha += s0; a.own_offset += s0;
      if (!a.reg_single) a.regs += s0;
      if (a.trials) a.trials += s0;
      a.accepts += s0;
    }
  }
  int loss;
  if (h->loss_quad_uniform) { loss = LOSS_QUAD_UNIFORM; a.loss_by_segment = 0; }
  else if (h->n_losses == 1) { loss = LOSS_SEGMENT; a.loss_by_segment = 0; }
  else if (rows) { loss = LOSS_PER_OBS; a.loss_by_segment = 0; }
  else { loss = LOSS_SEGMENT; a.loss_by_segment = 1; }
  if (!h->has_trig && loss != LOSS_QUAD_UNIFORM) loss += 2; // LOSS_*_NOTRIG: kernels compiled without the PeriodicLoss case
  glrm_handle::Ev ev{};
  const bool timed = h->profile && !eval_only;
  if (timed) {
    if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
    else { HIPCK(hipEventCreate(&ev.a)); HIPCK(hipEventCreate(&ev.b)); }
    ev.which = which;
    HIPCK(hipEventRecord(ev.a, h->stream));
  }
  const bool tiled = rows ? (h->tiled_row && !eval_only) : h->tiled_col;
  if (h->sum_order_opt) {
    rc = glrm_run_reforder(h, rows, min_stepsize, eval_only);
    if (rc) return rc;
  } else if (h->multi) {
    rc = glrm_run_multi(h, rows, min_stepsize, eval_only);
    if (rc) return rc;
  } else if (h->dense) {
    rc = glrm_run_dense(h, rows, min_stepsize, eval_only);
    if (rc) return rc;
  } else if (tiled || (rows ? (h->blocked_row && !eval_only) : h->blocked_col)) {
    const bool divert = !rows && h->blk_nlong_c > 0; // the very long columns: 8-wave gather sweep on the side stream, beside the passes
    if (divert) {
      // the gather sweep reads all of X: behind the sweeps already queued (fork) and, while X is still arriving
      // (glrm_hip_step_y_arrival), behind every block -- without holding up the passes on the main stream
      HIPCK(hipEventRecord(h->ev_fork, h->stream));
      HIPCK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
      for (int b = 0; b < h->n_arrival; ++b)
        if (h->arrival[b].event) HIPCK(hipStreamWaitEvent(h->side_stream, (hipEvent_t)h->arrival[b].event, 0));
      SweepArgs b = a;
      b.seglist = h->blk_long_c;
      b.nseg = h->blk_nlong_c;
      launch_sweep(h->G, h->R, 8, loss, h->unroll_long, b, h->side_stream);
    }
    rc = tiled ? glrm_run_tiled(h, rows, loss, a.loss_by_segment, min_stepsize, eval_only)
               : glrm_run_blocked(h, rows, loss, a.loss_by_segment, min_stepsize, eval_only);
    if (divert) { // join before anything else (also before reporting an error: later work on the stream stays ordered)
      char keep[sizeof g_err];
      memcpy(keep, g_err, sizeof keep);
      (void)hipEventRecord(h->ev_join, h->side_stream);
      (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);
      memcpy(g_err, keep, sizeof keep);
    }
    if (rc) return rc;
  } else {
    // gather sweeps (and the cached row sweep), class by class -- see build_class_plan
    const int64_t* ncls = rows ? h->ncls_r : h->ncls_c;
    const int32_t* lst = rows ? h->seglist_r : h->seglist_c;
    const int unroll = rows ? h->unroll_row : h->unroll_col;
    if (!lst) { // one class holds every local segment
      int cls = 1;
      for (int c = 0; c < 4; ++c) if (ncls[c] > 0) cls = c;
      if (cls == 0 && !eval_only) {
        if ((rc = glrm_run_cached(h, loss, min_stepsize, nullptr, 0, h->stream))) return rc;
      } else {
        launch_sweep(h->G, h->R, cls == 0 ? 1 : glrm_class_waves(cls), loss, cls <= 1 ? unroll : (glrm_class_waves(cls) == 8 ? h->unroll_long : 1), a, h->stream);
      }
    } else {
      // fork: the minority classes on the side stream, beside the majority class on the main stream; join
      int main_cls = 0;
      for (int c = 1; c < 4; ++c) if (ncls[c] > ncls[main_cls]) main_cls = c;
      HIPCK(hipEventRecord(h->ev_fork, h->stream));
      HIPCK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
      int64_t off = 0;
      int rc_cls = GLRM_OK;
      for (int c = 0; c < 4 && !rc_cls; off += ncls[c], ++c) {
        if (ncls[c] <= 0) continue;
        hipStream_t st = c == main_cls ? h->stream : h->side_stream;
        if (c == 0 && !eval_only) {
          rc_cls = glrm_run_cached(h, loss, min_stepsize, lst + off, ncls[0], st);
        } else {
          SweepArgs b = a;
          b.seglist = lst + off;
          b.nseg = ncls[c];
          launch_sweep(h->G, h->R, c == 0 ? 1 : glrm_class_waves(c), loss, c <= 1 ? unroll : (glrm_class_waves(c) == 8 ? h->unroll_long : 1), b, st);
        }
      }
      if (rc_cls) { // join before reporting: later work on h->stream (and a stream capture) must stay ordered after what the side stream already holds
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        (void)hipEventRecord(h->ev_join, h->side_stream);
        (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);
        memcpy(g_err, keep, sizeof keep);
        return rc_cls;
      }
      HIPCK(hipEventRecord(h->ev_join, h->side_stream));
      HIPCK(hipStreamWaitEvent(h->stream, h->ev_join, 0));
    }
  }
  HIPCK(hipGetLastError());
  if (timed) {
    HIPCK(hipEventRecord(ev.b, h->stream));
    h->pending.push_back(ev);
    if (h->pending.size() >= 4096) { rc = drain_events(h); if (rc) return rc; }
  }
  if (!eval_only) (rows ? h->launches_x : h->launches_y) += 1;
  return GLRM_OK;
}

extern "C" int glrm_hip_step_x(glrm_handle* h, double min_stepsize) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  return run_sweep(h, 0, min_stepsize, 0);
}

extern "C" int glrm_hip_step_x_range(glrm_handle* h, int64_t seg_begin, int64_t seg_end, double min_stepsize) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  if (seg_begin < 0 || seg_end > h->ml || seg_begin > seg_end) return fail(GLRM_ERR_INVALID, "row range out of bounds");
  if (h->dense) return fail(GLRM_ERR_UNSUPPORTED, "step_x_range is not available on the dense path");
  DeviceGuard dg(h->device);
  h->rng_b = seg_begin;
  h->rng_e = seg_end;
  const int rc = run_sweep(h, 0, min_stepsize, 0);
  h->rng_b = 0;
  h->rng_e = -1;
  return rc;
}

extern "C" int glrm_hip_step_y(glrm_handle* h, double min_stepsize) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  return run_sweep(h, 1, min_stepsize, 0);
}

// ---- the Y half-step while X is still arriving (include/glrm_hip.h: glrm_hip_step_y_arrival) ----------------------------------------
int glrm_arrival_wait(glrm_handle* h, int64_t lo, int64_t hi) {
  for (int b = 0; b < h->n_arrival; ++b) {
    const glrm_arrival& blk = h->arrival[b];
    if (h->arrival_waited[b] || blk.end <= lo || blk.begin >= hi) continue;
    h->arrival_waited[b] = 1;
    if (!blk.event) continue;
    glrm_handle::Ev ev{};
    if (h->profile) { // what the launch stream spends in front of a block that is not there yet (ms_wait_y)
      if (!h->pool.empty()) { ev = h->pool.back(); h->pool.pop_back(); }
      else { HIPCK(hipEventCreate(&ev.a)); HIPCK(hipEventCreate(&ev.b)); }
      ev.which = 2;
      HIPCK(hipEventRecord(ev.a, h->stream));
    }
    HIPCK(hipStreamWaitEvent(h->stream, (hipEvent_t)blk.event, 0));
    if (h->profile) {
      HIPCK(hipEventRecord(ev.b, h->stream));
      h->pending.push_back(ev);
    }
  }
  return GLRM_OK;
}

int glrm_for_sup_runs_in_arrival_order(glrm_handle* h, int nsup, int64_t rows_per_sup, const std::function<int(int, int)>& launch) {
  if (!h->arrival || h->n_arrival <= 0) return launch(0, nsup);
  if (nsup <= 1) { // one super-tile reads every row: behind ALL announced blocks (before session r6_69 it was launched without any wait --
                   // problems of at most one tile of rows on several shards raced with the exchange: tests/perf/soak_lane_shards.py, seed 5004)
    const int rc = glrm_arrival_wait(h, 0, h->m);
    return rc ? rc : launch(0, nsup);
  }
  // need[s] = the LAST announced block super-tile s touches: the run of super-tiles that become complete with block b is launched behind b
  std::vector<int> need((size_t)nsup, -1);
  for (int s = 0; s < nsup; ++s) {
    const int64_t lo = (int64_t)s * rows_per_sup, hi = std::min<int64_t>(lo + rows_per_sup, h->m);
    for (int b = 0; b < h->n_arrival; ++b)
      if (h->arrival[b].begin < hi && h->arrival[b].end > lo && h->arrival[b].begin < h->arrival[b].end) need[s] = b;
  }
  for (int b = -1; b < h->n_arrival; ++b) { // (-1: super-tiles no block touches -- beyond m -- first)
    for (int s = 0; s < nsup;) {
      if (need[s] != b) { ++s; continue; }
      int e = s + 1;
      while (e < nsup && need[e] == b) ++e;
      const int64_t lo = (int64_t)s * rows_per_sup, hi = std::min<int64_t>((int64_t)e * rows_per_sup, h->m);
      int rc = glrm_arrival_wait(h, lo, hi);
      if (!rc) rc = launch(s, e);
      if (rc) return rc;
      s = e;
    }
  }
  return GLRM_OK;
}

extern "C" int glrm_hip_step_y_arrival(glrm_handle* h, double min_stepsize, const glrm_arrival* blocks, int32_t n_blocks) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  if (n_blocks < 0 || (n_blocks > 0 && !blocks)) return fail(GLRM_ERR_INVALID, "bad block list");
  if (n_blocks == 0) return glrm_hip_step_y(h, min_stepsize);
  { // the blocks must tile [0, m): every row of X is either there or announced by exactly one event
    std::vector<std::pair<int64_t, int64_t>> r;
    for (int b = 0; b < n_blocks; ++b) {
      if (blocks[b].begin < 0 || blocks[b].end > h->m || blocks[b].begin > blocks[b].end) return fail(GLRM_ERR_INVALID, "arrival block %d: rows [%lld, %lld) out of range", b, (long long)blocks[b].begin, (long long)blocks[b].end);
      if (blocks[b].begin < blocks[b].end) r.emplace_back(blocks[b].begin, blocks[b].end);
    }
    std::sort(r.begin(), r.end());
    int64_t at = 0;
    for (auto& x : r) {
      if (x.first != at) return fail(GLRM_ERR_INVALID, "arrival blocks must tile the rows [0, m) of X without gaps or overlaps (at row %lld)", (long long)at);
      at = x.second;
    }
    if (at != h->m) return fail(GLRM_ERR_INVALID, "arrival blocks must tile the rows [0, m) of X (they end at row %lld of %lld)", (long long)at, (long long)h->m);
  }
  GLRM_NEED_FINALIZED(h);
  DeviceGuard dg(h->device);
  h->arrival = blocks;
  h->n_arrival = n_blocks;
  h->arrival_waited.assign((size_t)n_blocks, 0);
  int rc = GLRM_OK;
  // the pass families of the column view can start on a part of X (the phase-aligned passes super-tile by super-tile in true arrival order,
  // the LDS-tiled / lane passes in runs of super-tiles in the announced order: round 6); everything else needs all of it
  const bool by_super_tile = ((h->blocked_col && !h->lockstep && !h->tiled_col) || h->tiled_col) && !h->multi && !h->dense && !h->sum_order_opt && env_int("GLRM_HIP_ARRIVAL", 1);
  if (!by_super_tile) rc = glrm_arrival_wait(h, 0, h->m);
  if (!rc) rc = run_sweep(h, 1, min_stepsize, 0);
  if (!rc) rc = glrm_arrival_wait(h, 0, h->m); // (a shard without columns launches nothing: later work on the stream still follows the arrivals)
  h->arrival = nullptr;
  h->n_arrival = 0;
  h->sup_order.clear();
  return rc;
}

// One prox-gradient step with a global step size and no line search (src/algorithms/sparse_proxgrad.jl:59-77, :81-99).
static int gradstep(glrm_handle* h, int which, double alpha) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  if (!(alpha > 0.0)) return fail(GLRM_ERR_INVALID, "the step size must be positive");
  DeviceGuard dg(h->device);
  h->fixed_alpha = alpha;
  const int rc = run_sweep(h, which, 0.0, 0);
  h->fixed_alpha = 0.0;
  return rc;
}
extern "C" int glrm_hip_gradstep_x(glrm_handle* h, double alpha) { return gradstep(h, 0, alpha); }
extern "C" int glrm_hip_gradstep_y(glrm_handle* h, double alpha) { return gradstep(h, 1, alpha); }

extern "C" int glrm_hip_col_losses(glrm_handle* h) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  return run_sweep(h, 1, 0.0, 1);
}

static int run_penalty(glrm_handle* h, bool rows) {
  int rc = ensure_owned(h);
  if (rc) return rc;
  const int64_t nseg = rows ? h->ml : h->nl;
  if (nseg <= 0) return GLRM_OK;
  if (h->multi) return glrm_run_multi_penalty(h, rows);
  hipLaunchKernelGGL(penalty_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, h->stream, rows ? h->X : h->Y, h->kp,
                     h->k, rows ? h->rb : h->cb, nseg, rows ? h->rx : h->ry, (rows ? h->n_rx : h->n_ry) == 1 ? 1 : 0,
                     rows ? h->objrow : h->objcol);
  HIPCK(hipGetLastError());
  return GLRM_OK;
}

extern "C" int glrm_hip_row_penalties(glrm_handle* h) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  return run_penalty(h, true);
}

extern "C" int glrm_hip_col_penalties(glrm_handle* h) {
  if (!h) return fail(GLRM_ERR_INVALID, "NULL handle");
  DeviceGuard dg(h->device);
  return run_penalty(h, false);
}

extern "C" int glrm_hip_sum(glrm_handle* h, const void* dvec, int64_t n, double* out) {
  if (!h || !out || (n > 0 && !dvec)) return fail(GLRM_ERR_INVALID, "NULL argument");
  DeviceGuard dg(h->device);
  if (h->sum_order_opt) return glrm_reforder_sum(h, dvec, n, out); // sum(::Vector{Float64}) as Julia adds it (proxgrad.jl:205)
  hipLaunchKernelGGL(sum_stage1, dim3(SUM_BLOCKS), dim3(SUM_THREADS), 0, h->stream, (const double*)dvec, n, h->partials);
  hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(SUM_THREADS), 0, h->stream, h->partials, h->dscalar);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(out, h->dscalar, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  return GLRM_OK;
}

static int count_sum(glrm_handle* h, const int32_t* v, int64_t n, int64_t* out) {
  *out = 0;
  if (n <= 0) return GLRM_OK;
  HIPCK(hipMemsetAsync(h->dcount, 0, 8, h->stream));
  hipLaunchKernelGGL(isum_kernel, dim3(256), dim3(256), 0, h->stream, v, n, h->dcount);
  HIPCK(hipGetLastError());
  unsigned long long r = 0;
  HIPCK(hipMemcpyAsync(&r, h->dcount, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCK(hipStreamSynchronize(h->stream));
  *out = (int64_t)r;
  return GLRM_OK;
}

extern "C" int glrm_hip_kernel_stats(glrm_handle* h, glrm_kernel_stats* out, int reset) {
  if (!h || !out) return fail(GLRM_ERR_INVALID, "NULL argument");
  DeviceGuard dg(h->device);
  int rc = drain_events(h);
  if (rc) return rc;
  memset(out, 0, sizeof *out);
  out->launches_x = h->launches_x; out->launches_y = h->launches_y;
  out->ms_x = h->ms_x; out->ms_y = h->ms_y; out->ms_wait_y = h->ms_wait;
  if ((rc = count_sum(h, h->trials_r, h->ml, &out->trials_x))) return rc;
  if ((rc = count_sum(h, h->accepts_r, h->ml, &out->accepts_x))) return rc;
  if ((rc = count_sum(h, h->trials_c, h->nl, &out->trials_y))) return rc;
  if ((rc = count_sum(h, h->accepts_c, h->nl, &out->accepts_y))) return rc;
  out->nnz_rows = h->nnz_r; out->nnz_cols = h->nnz_c;
  out->waves_row = h->waves_row; out->waves_col = h->waves_col; out->ld = h->kp;
  out->tiled = h->multi ? 8 : (h->tiled_row ? 1 : 0) | (h->tiled_col ? 2 : 0) | (h->dense ? 4 : 0) | (h->cached_row ? 64 : 0) | (h->blocked_row ? 16 : 0) |
               (h->blocked_col ? 32 : 0) | (h->sum_order_opt ? 128 : 0) | (h->lane[0] ? 256 : 0) | (h->lane[1] ? 512 : 0); // bit8 / bit9: lane-per-segment form of the LDS-tiled row / column passes; bit0 / bit1: LDS-tiled row / column sweep, bit4 / bit5: phase-aligned gather passes, bit7: reference order
  if (reset) {
    h->launches_x = h->launches_y = 0;
    h->ms_x = h->ms_y = h->ms_wait = 0;
    const int64_t ml1 = h->ml > 0 ? h->ml : 1, nl1 = h->nl > 0 ? h->nl : 1;
    HIPCK(hipMemsetAsync(h->trials_r, 0, ml1 * 4, h->stream));
    HIPCK(hipMemsetAsync(h->accepts_r, 0, ml1 * 4, h->stream));
    HIPCK(hipMemsetAsync(h->trials_c, 0, nl1 * 4, h->stream));
    HIPCK(hipMemsetAsync(h->accepts_c, 0, nl1 * 4, h->stream));
  }
  return GLRM_OK;
}

// The order in which this handle adds a segment's terms (include/glrm_hip.h: glrm_sum_order) -- a description of what the kernels of
// the family that run_sweep dispatches do, read from the same handle fields.  The CPU oracle adopts it (glrm_cpu_set_sum_order) and then
// lands on this engine's factors bit for bit: tests/test_gpu_sum_order.py.
extern "C" int glrm_hip_sum_order(glrm_handle* h, int32_t which, glrm_sum_order* out) {
  if (!h || !out || (which != 0 && which != 1)) return fail(GLRM_ERR_INVALID, "bad argument");
  GLRM_NEED_FINALIZED(h);
  const bool rows = which == 0;
  glrm_sum_order o{};
  o.cached_maxlen = -1;
  o.waves4_from = GLRM_WAVES4_FROM;
  o.waves8_from = GLRM_WAVES8_FROM;
  // the loss variant run_sweep instantiates: 0 = one QuadLoss, segment = one descriptor per segment, per-observation = rows of a heterogeneous model
  const bool quad = h->loss_quad_uniform, per_obs = !quad && h->n_losses > 1 && rows;
  const bool tiled = rows ? h->tiled_row != 0 : h->tiled_col != 0;
  const bool blocked = rows ? h->blocked_row != 0 : h->blocked_col != 0;
  if (!rows && (h->blocked_col || h->tiled_col) && !h->sum_order_opt && !h->multi && !h->dense) o.long_from = (int32_t)std::min<int64_t>(h->blk_long_from, INT32_MAX);
  if (h->sum_order_opt) {
    o.family = GLRM_ORDER_REFERENCE; // glrm_reforder.hip: one lane per segment, list order, one accumulator per sum
    o.lanes = 1; o.comps = h->kp;
  } else if (h->multi || h->dense || (blocked && !rows && h->lockstep)) {
    o.family = GLRM_ORDER_OTHER;
  } else if (tiled) {
    const int T = h->order_unit; // vectors per staged tile (half a tile with loader waves)
    const bool lw = h->tile_lw > 0 && h->tile_cfg && ((h->tile_lw_sides >> (rows && !h->row_split ? 0 : 1)) & 1);
    o.family = GLRM_ORDER_WINDOWED;
    o.lanes = h->tG; o.comps = h->tR;
    o.window = lw ? T : (h->tile_lw > 0 ? 2 * T : T);
    const int64_t tps = rows ? (h->row_split ? h->tiles_per_sup_r : 0) : h->tiles_per_sup; // (row rounds: one super-tile, nothing re-added = 0)
    o.windows_per_sup = lw ? 2 * tps : tps;
    o.batch = (!quad && (h->tG == 4 || h->tG == 8)) ? h->tG : 2;
    o.rotate = (!(rows && !h->row_split) && !lw && !quad && GLRM_TILE_ROT && (h->tG == 4 || h->tG == 8) && h->tR == 8) ? 1 : 0;
    // a private copy in another order: lists the engine tile-sorted, rows regrouped by loss kind inside the tile windows
    const bool sorted_here = rows ? h->sig_local.rows_unordered != 0 : h->sig_local.cols_unordered != 0;
    const bool grouped = rows && h->n_losses > 1 && h->nnz_r > 0 && env_int("GLRM_HIP_GROUP_KINDS", 1) && !h->lane[0]; // glrm_tiled.hpp: group_rows_by_kind_kernel (lane rows keep the caller's order)
    o.private_order = sorted_here ? 1 : grouped ? 2 : 0; // (2: stable grouping by ascending loss kind inside every window -- the oracle restates it)
    if (h->lane[rows ? 0 : 1]) { // lane-per-segment passes (glrm_lane.hpp): two fma chains over the even / odd chunks = the two-lane layout, rotated walk
      o.lanes = 2; o.comps = h->kp / 2;
      o.batch = 2;
      o.rotate = 2;
      o.window = T;
      o.windows_per_sup = rows ? 0 : h->tiles_per_sup;
    }
  } else if (blocked) {
    const int Tb = ((150 * 1024) / (h->kp * 8 + 16)) / 16 * 16; // glrm_blocked.hip: tile_rows_b
    o.family = GLRM_ORDER_WINDOWED;
    o.lanes = h->G; o.comps = h->R;
    o.window = (int64_t)Tb * (rows ? h->tiles_per_sup_r : h->tiles_per_sup); // one walk per super-tile: the super-tile is the window
    o.windows_per_sup = 1;
    o.batch = (!quad && (h->G == 4 || h->G == 8)) ? h->G : 2;
  } else {
    o.family = GLRM_ORDER_STRIDED;
    o.lanes = h->G; o.comps = h->R;
    const int forced = rows ? h->opts.waves_row : h->opts.waves_col;
    o.waves = (forced == 1 || forced == 4 || forced == 8) ? forced : 0;
    if (rows && h->cached_want) { // which rows the cached sweep takes is a function of the row's own length
      const int keep = h->cached_row;   // 0 on a shard that holds no such row: the variant is still the whole problem's
      if (!keep) h->cached_row = env_int("GLRM_HIP_CACHED_REGS", 1) ? 2 : 1;
      o.cached_maxlen = glrm_cached_maxlen(h);
      o.cached_waves = h->cached_row == 2 ? env_int("GLRM_HIP_CACHED_WAVES", 2) : 1;
      h->cached_row = keep;
    }
    // four observations per trip with one loss evaluation per lane (sweep_pass: SCATTER): G == 4 and not the uniform QuadLoss kernel;
    // the per-segment variant only on one-wave segments (launch_sweep_loss)
    o.batch = (!quad && h->G == 4) ? 4 : 1;
    o.batch_one_wave_only = (o.batch == 4 && !per_obs) ? 1 : 0;
  }
  *out = o;
  return GLRM_OK;
}

// ------------------------------------------------------------------ whole-fit API

static bool single_shard(const glrm_handle* h) { return h->rb == 0 && h->re == h->m && h->cb == 0 && h->ce == h->n; }

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// loss + rx + ry at the device-resident factors (objective(...), src/evaluate_fit.jl:4-23,91-104)
static int device_objective(glrm_handle* h, int include_reg, double* out) {
  int rc;
  double loss = 0, px = 0, py = 0;
  if (h->sum_order_opt) { // reference order: one accumulator over every observation, columns outer (src/evaluate_fit.jl:12-21)
    if ((rc = ensure_owned(h))) return rc;
    return glrm_reforder_objective(h, include_reg, out);
  }
  if ((rc = run_sweep(h, 1, 0.0, 1))) return rc;
  if ((rc = glrm_hip_sum(h, h->objcol, h->n, &loss))) return rc;
  if (include_reg) {
    if ((rc = run_penalty(h, true))) return rc;
    if ((rc = glrm_hip_sum(h, h->objrow, h->m, &px))) return rc;
    if ((rc = run_penalty(h, false))) return rc;
    if ((rc = glrm_hip_sum(h, h->objcol, h->n, &py))) return rc;
  }
  *out = loss + (px + py);
  return GLRM_OK;
}

extern "C" int glrm_hip_objective(glrm_handle* h, const double* X, const double* Y, int include_reg, double* out) {
  if (!h || !X || !Y || !out) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (!single_shard(h)) return fail(GLRM_ERR_INVALID, "glrm_hip_objective needs a single-shard handle");
  DeviceGuard dg(h->device);
  int rc = glrm_hip_set_factors(h, X, Y);
  if (rc) return rc;
  return device_objective(h, include_reg, out);
}

// One outer iteration as a hipGraph.  Eligible: gather sweeps (fixed launch sequence, no host round trips inside a half-step) on
// the handle's private stream (the legacy default stream cannot be captured), no per-launch event timing.
static bool graph_eligible(const glrm_handle* h) {
  return h->own_stream && !h->profile && !h->multi && !h->dense && !h->sum_order_opt && !h->tiled_row && !h->tiled_col && !h->blocked_row && !h->blocked_col && !h->cached_row &&
         env_int("GLRM_HIP_GRAPH", 1) != 0;
}

static int build_iteration_graph(glrm_handle* h, const glrm_params* prm) {
  if (h->iter_exec && h->graph_ix == prm->inner_iter_X && h->graph_iy == prm->inner_iter_Y && h->graph_min == prm->min_stepsize &&
      h->graph_step == prm->stepsize)
    return GLRM_OK;
  if (h->iter_exec) { (void)hipGraphExecDestroy(h->iter_exec); h->iter_exec = nullptr; }
  if (h->iter_graph) { (void)hipGraphDestroy(h->iter_graph); h->iter_graph = nullptr; }
  if (!h->pinned_obj) HIPCK(hipHostMalloc((void**)&h->pinned_obj, 8, hipHostMallocDefault));
  int rc = ensure_owned(h);
  if (rc) return rc;
  HIPCK(hipStreamSynchronize(h->stream));
  if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return GLRM_OK; } // no graph: plain launches
  bool ok = true;
  const int64_t lx = h->launches_x, ly = h->launches_y;
  if (prm->inner_iter_X > 1 || prm->inner_iter_Y > 1) ok = ok && glrm_hip_reset_stepsizes(h, prm->stepsize) == GLRM_OK;
  for (int64_t in = 0; ok && in < prm->inner_iter_X; ++in) ok = run_sweep(h, 0, prm->min_stepsize, 0) == GLRM_OK;
  for (int64_t in = 0; ok && in < prm->inner_iter_Y; ++in) ok = run_sweep(h, 1, prm->min_stepsize, 0) == GLRM_OK;
  if (ok) {
    hipLaunchKernelGGL(sum_stage1, dim3(SUM_BLOCKS), dim3(SUM_THREADS), 0, h->stream, (const double*)h->objcol, h->n, h->partials);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(SUM_THREADS), 0, h->stream, h->partials, h->dscalar);
    ok = hipMemcpyAsync(h->pinned_obj, h->dscalar, 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
  }
  h->launches_x = lx; h->launches_y = ly; // capturing is not launching
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(h->stream, &g);
  if (!ok || ec != hipSuccess || !g) { if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); return GLRM_OK; }
  if (hipGraphInstantiate(&h->iter_exec, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); h->iter_exec = nullptr; (void)hipGetLastError(); return GLRM_OK; }
  h->iter_graph = g;
  h->graph_ix = prm->inner_iter_X; h->graph_iy = prm->inner_iter_Y; h->graph_min = prm->min_stepsize; h->graph_step = prm->stepsize;
  return GLRM_OK;
}

extern "C" int glrm_hip_fit(glrm_handle* h, const glrm_params* prm, double* X, double* Y, double* objective,
                            double* seconds, int64_t cap, int64_t* n_recorded) {
  if (!h || !prm || !X || !Y || !objective || !seconds || !n_recorded) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (!single_shard(h)) return fail(GLRM_ERR_INVALID, "glrm_hip_fit needs a single-shard handle (use the step-level API per shard)");
  if (prm->max_iter < 0 || cap < prm->max_iter + 1) return fail(GLRM_ERR_INVALID, "objective/seconds capacity must be >= max_iter+1");
  if (prm->inner_iter_X < 1 || prm->inner_iter_Y < 1) return fail(GLRM_ERR_INVALID, "inner iteration counts must be >= 1");
  double ynorm = 0.0; // norm(Y)==0 guard, proxgrad.jl:45-48 (the reference would hit an UndefVarError)
  for (int64_t i = 0; i < (int64_t)h->k * h->d; ++i) ynorm += Y[i] * Y[i];
  if (ynorm == 0.0) return fail(GLRM_ERR_INVALID, "Y is all zeros (the reference cannot start from Y == 0)");
  DeviceGuard dg(h->device);
  int rc;
  if ((rc = glrm_hip_set_factors(h, X, Y))) return rc;                 // X = glrm.X; Y = glrm.Y (:43)
  if ((rc = glrm_hip_reset_stepsizes(h, prm->stepsize))) return rc;   // :69-70
  const double scaled_abs_tol = prm->abs_tol * (double)h->nnz_r;       // :72 (observed_features)
  int64_t nrec = 0;
  if ((rc = device_objective(h, 1, &objective[0]))) return rc;        // update_ch!(ch, 0, objective(...)) :76
  seconds[0] = 0.0;
  nrec = 1;
  double t = now_s();
  const bool use_graph = graph_eligible(h);
  if (use_graph && (rc = build_iteration_graph(h, prm))) return rc;
  for (int64_t i = 1; i <= prm->max_iter; ++i) {                       // :107
    double obj = 0.0;
    if (use_graph && h->iter_exec) { // the same launches, replayed from one hipGraph
      HIPCK(hipGraphLaunch(h->iter_exec, h->stream));
      HIPCK(hipStreamSynchronize(h->stream));
      obj = *h->pinned_obj;
      h->launches_x += prm->inner_iter_X; h->launches_y += prm->inner_iter_Y;
    } else {
      if (prm->inner_iter_X > 1 || prm->inner_iter_Y > 1)
        if ((rc = glrm_hip_reset_stepsizes(h, prm->stepsize))) return rc; // :112-115
      for (int64_t in = 0; in < prm->inner_iter_X; ++in)
        if ((rc = run_sweep(h, 0, prm->min_stepsize, 0))) return rc;    // :117-158
      for (int64_t in = 0; in < prm->inner_iter_Y; ++in)
        if ((rc = run_sweep(h, 1, prm->min_stepsize, 0))) return rc;    // :160-203
      if ((rc = glrm_hip_sum(h, h->objcol, h->n, &obj))) return rc;     // obj = sum(obj_by_col) :205
    }
    const double dt = now_s() - t;
    objective[nrec] = obj;
    seconds[nrec] = seconds[nrec - 1] + dt;                            // update_ch! (src/convergence.jl:22-26)
    ++nrec;
    t = now_s();
    const double dec = objective[nrec - 2] - obj;                      // :210
    if (i > 10 && (dec < scaled_abs_tol || dec / obj < prm->rel_tol)) break; // :211-213
  }
  if ((rc = glrm_hip_get_factors(h, X, Y))) return rc;
  *n_recorded = nrec;
  return GLRM_OK;
}

// fit!(glrm::GLRM, params::SparseProxGradParams), src/algorithms/sparse_proxgrad.jl:22-134: global step size, one
// gradient + prox step per factor and iteration, whole-iteration accept / revert on the full objective.
extern "C" int glrm_hip_fit_sparse(glrm_handle* h, const glrm_sparse_params* prm, double* X, double* Y, double* objective,
                                   double* seconds, int64_t cap, int64_t* n_recorded) {
  if (!h || !prm || !X || !Y || !objective || !seconds || !n_recorded) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (!single_shard(h)) return fail(GLRM_ERR_INVALID, "glrm_hip_fit_sparse needs a single-shard handle");
  if (prm->max_iter < 0 || cap < prm->max_iter + 2) return fail(GLRM_ERR_INVALID, "objective/seconds capacity must be >= max_iter+2");
  if (prm->inner_iter < 1) return fail(GLRM_ERR_INVALID, "inner_iter must be >= 1");
  double ynorm = 0.0; // norm(Y)==0 would be re-randomised by the reference (:41-43); not reproducible -> error
  for (int64_t i = 0; i < (int64_t)h->k * h->d; ++i) ynorm += Y[i] * Y[i];
  if (ynorm == 0.0) return fail(GLRM_ERR_INVALID, "Y is all zeros");
  DeviceGuard dg(h->device);
  int rc;
  if ((rc = glrm_hip_set_factors(h, X, Y))) return rc; // working copies X, Y (:33); glrm.X / glrm.Y = best so far
  const size_t xb = (size_t)h->kp * h->m * 8, yb = (size_t)h->kp * h->d * 8;
  double *bestX = nullptr, *bestY = nullptr;
  HIPCK(hipMalloc((void**)&bestX, xb));
  if (hipMalloc((void**)&bestY, yb) != hipSuccess) { (void)hipFree(bestX); return fail(GLRM_ERR_OOM, "out of device memory"); }
  auto cleanup = [&](int code) { (void)hipFree(bestX); (void)hipFree(bestY); return code; };
  if (hipMemcpyAsync(bestX, h->X, xb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(bestY, h->Y, yb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "device copy failed"));
  double alpha = prm->stepsize;                              // :46
  const double tol = prm->abs_tol * (double)h->nnz_r;        // :48
  int64_t nrec = 0;
  if ((rc = device_objective(h, 1, &objective[0]))) return cleanup(rc); // update_ch!(ch, 0, objective(glrm; sparse=true)) :52
  seconds[0] = 0.0;
  nrec = 1;
  double t = now_s();
  int64_t steps_in_a_row = 0;
  for (int64_t i = 1; i <= prm->max_iter; ++i) {
    for (int64_t in = 0; in < prm->inner_iter; ++in) { h->fixed_alpha = alpha; rc = run_sweep(h, 0, 0.0, 0); h->fixed_alpha = 0.0; if (rc) return cleanup(rc); }
    for (int64_t in = 0; in < prm->inner_iter; ++in) { h->fixed_alpha = alpha; rc = run_sweep(h, 1, 0.0, 0); h->fixed_alpha = 0.0; if (rc) return cleanup(rc); }
    double obj = 0.0;
    if ((rc = device_objective(h, 1, &obj))) return cleanup(rc); // :102
    if (obj < objective[nrec - 1]) {                              // :104-110
      const double dt = now_s() - t;
      objective[nrec] = obj;
      seconds[nrec] = seconds[nrec - 1] + dt;
      ++nrec;
      if (hipMemcpyAsync(bestX, h->X, xb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
          hipMemcpyAsync(bestY, h->Y, yb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
        return cleanup(fail(GLRM_ERR_HIP, "device copy failed"));
      alpha = alpha * 1.05;
      steps_in_a_row = steps_in_a_row + 1 > 1 ? steps_in_a_row + 1 : 1;
      t = now_s();
    } else {                                                      // :111-117
      const double div = -(double)steps_in_a_row > 1.5 ? -(double)steps_in_a_row : 1.5;
      alpha = alpha / div;
      if (hipMemcpyAsync(h->X, bestX, xb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
          hipMemcpyAsync(h->Y, bestY, yb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
        return cleanup(fail(GLRM_ERR_HIP, "device copy failed"));
      steps_in_a_row = steps_in_a_row - 1 < 0 ? steps_in_a_row - 1 : 0;
    }
    // :119  i>10 && (steps_in_a_row > 3 && ch.objective[end-1] - obj < tol) || alpha <= params.min_stepsize
    if ((i > 10 && (steps_in_a_row > 3 && nrec >= 2 && objective[nrec - 2] - obj < tol)) || alpha <= prm->min_stepsize) break;
  }
  // :126-127 the last objective is recorded once more with the remaining time
  objective[nrec] = objective[nrec - 1];
  seconds[nrec] = seconds[nrec - 1] + (now_s() - t);
  ++nrec;
  // glrm.X, glrm.Y are the best model found
  if (hipMemcpyAsync(h->X, bestX, xb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(h->Y, bestY, yb, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "device copy failed"));
  rc = glrm_hip_get_factors(h, X, Y);
  *n_recorded = nrec;
  return cleanup(rc);
}
