// glrm_lane.hpp -- LDS-tiled passes with ONE LANE per segment (round 6).
//
// The LDS-tiled sweeps of glrm_tiled.hpp give a segment (row or column) to a group of four lanes: every dot product ends in two DPP
// butterfly steps, the (index, value) batch is broadcast inside the group, validity is a chain of compares -- 2.7 (trial pass) to 4.0
// (gradient pass) VALU wave-instructions per observation and pass at C2 against 0.5 / 1.0 of fused multiply-adds, a quarter of every wave's
// cycles in issue stalls behind those dependent cross-lane steps (profiles/r06_c2_sq_counters.md).  Here a LANE owns the segment: x, g and the
// fetched y of a segment live in that lane's registers (k = 32: 64 + 64 + 64 VGPRs, two waves per SIMD), no cross-lane instruction exists in
// the pass, a workgroup of 8 waves holds 512 segments per staged tile instead of 256 (half the staging passes over the opposing factor).
//
// Round 2 priced this layout (tools/ubench_lanerow.hip: 10.8 ps per observation and pass against the product's ~8) and dropped it; that
// prototype read 64 unrelated PADDED rows per ds_read_b128 -- 51 TB/s of LDS against 118 conflict-free.  What makes it pay (tools/ubench_lane1.hip,
// profiles/r06_ubench_lane1.txt: 5.1 ps per observation for the trial pass, 6.9 for the gradient pass, against 8.4 / 9.5 of the product's column
// passes and 7.0 of its fused row sweep):
//   * conflict-free tile reads: rows are staged UNPADDED (256 B at k = 32: every row starts at bank 0) and lane l walks a row's sixteen
//     16-byte chunks in the order i ^ p, p = (global segment id) & 15.  The 16 lanes that share an LDS cycle of ds_read_b128
//     ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md) hold 16 different p when segments sit in the slots in order, so they read
//     16 different 4-bank groups whatever rows they read: SQ_LDS_BANK_CONFLICT = 0, 97 TB/s for the pattern alone against 9.9 TB/s in natural
//     order.  Register i of the lane holds chunk i ^ p of x, g and y alike; p is a function of the segment only, so no sum depends on the slot
//     a segment runs in (a slot order that breaks the pattern costs conflicts, never bits);
//   * the observation stream in a SELL layout built at finalize (glrm_lane.hip): per (wave block of 64 slots, tile) a run of STEPS, each step one
//     (byte offset of the staged vector inside the tile, value) pair per lane, padded to the longest list of the block in that tile (offset -1 =
//     idle lane).  Reads are coalesced (256 + 512 B per step and wave), trip counts are known, the LDS address is the stored offset plus a
//     per-lane constant -- no index arithmetic, no validity chain -- and the pairs run U steps ahead of their use (two waves per SIMD cannot
//     hide an HBM round trip per step).  Price: the padding (x 1.46 at 28 observations per segment and tile) in stream bytes and in issue slots.
//
// Summation order (what glrm_hip_sum_order reports for a side on this family; the oracle restates it): GLRM_ORDER_WINDOWED with lanes = 2,
// comps = kp / 2, batch = 2, rotate = 2 -- the dot product is TWO fma chains, over the even and the odd 16-byte chunks (the lane layout of two
// lanes), each in the order i ^ ((gseg >> 1) & 7) of its own chunks, added once; loss terms go to two partial sums by the entry's position
// inside its tile window modulo 2, added at the end of a super-tile; gradient terms are added in list order; the regularizer sums of
// col_reduce / col_decide run in the same two-lane layout (launched as <2, kp / 2>).
#pragma once

#include "glrm_tiled.hpp"

// Experiment switch (session r6_14, measured and dropped): request one dword of every 128-byte line of the NEXT tile while the current one is
// consumed, so that the LDS-DMA finds its lines in L2.  C2 13.0-13.2 ms against 12.5-12.6: vmcnt counts in order, so the first wait for stream
// data after these loads inherits their latency, and the extra registers push the gradient pass into scratch.
#ifndef GLRM_LANE_PREFETCH
#define GLRM_LANE_PREFETCH 0
#endif

namespace glrm {

struct LaneArgs {
  const int64_t* bptr; // [wave blocks][ntiles + 1]: first step of (wave block, tile); a step = 64 (offset, value) pairs, one per lane
  const int32_t* off;  // [steps][64] byte offset of the staged vector inside its tile (local index x kp x 8, below 2^20) | the id of the
                       // entry's loss descriptor << 20 (rows of a model with a loss per column; 0 otherwise); -1 = idle lane
  const double* val;   // [steps][64]
  int ntiles;          // tiles of the opposing factor (stride of bptr minus one)
  int64_t nwb;         // wave blocks the layout holds (a launch rounds its grid up to whole workgroups: blocks beyond have no steps)
  int64_t slot0;       // local id of TiledArgs' segment 0 in the slot space the layout was built on (row sub-ranges: glrm_hip_step_x_range)
};

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// GRAD: pass 1 (gradient + loss partials at a.own); else a trial pass (loss partials at a.trial for the still-searching segments).
// CSR: the (index, value) pairs come from the segment's own list (one lane walks it: uncoalesced) instead of the SELL layout -- the compact
// trial rounds over the few segments still searching and sub-range sweeps the layout does not cover; the same sums in the same order.
template <int KP, int NW, int TILE, int LOSS, bool GRAD, bool CSR>
__global__ void __launch_bounds__(NW * 64, 1) lane_pass_kernel(const TiledArgs a, const LaneArgs la) {
  static_assert(KP == 32, "one lane per segment: x, g and y of a segment in one lane's registers -- built for a padded rank of 32");
  constexpr int C = KP / 2;            // 16-byte chunks per vector
  constexpr int U = GRAD ? 2 : 4;      // steps in flight ahead of their use (even: a step's position parity inside its window is u & 1)
  constexpr int PSTRIDE = KP + 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wb = (CSR ? 0 : la.slot0 / 64) + (int64_t)blockIdx.x * NW + wave; // wave block in the layout's slot space
  const int64_t slot = wb * 64 + lane;
  const int64_t rel = CSR ? slot : slot - la.slot0;                                 // slot relative to TiledArgs' segment 0
  const int64_t nslots = CSR ? a.nseg : (a.npass > 0 ? a.npass : a.nseg);
  const int sup = a.sup0 + (int)blockIdx.y;
  bool have = rel >= 0 && rel < nslots;
  const int64_t seg = (have && a.segperm) ? (int64_t)a.segperm[rel] : (have ? rel : 0);
  if (!GRAD && have) have = a.active[seg] != 0;
  if (!GRAD && !__syncthreads_or(have ? 1 : 0)) return; // nothing left to evaluate in this block of segments
  const int64_t gseg = a.own_offset + seg;
  const int p = (int)(gseg & (C - 1)), pb = p * 16;
  const double2* xp = reinterpret_cast<const double2*>(GRAD ? a.own + gseg * KP : a.trial + seg * (int64_t)KP);
  double2 x[C], g[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    x[i] = have ? xp[i ^ p] : make_double2(0.0, 0.0);
    g[i] = make_double2(0.0, 0.0);
  }
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != 2) segloss = load_loss(a.losses, (a.loss_by_segment && have) ? gseg : 0);
  const int ntiles = (int)((a.n_other + TILE - 1) / TILE);
  const int tb = sup * a.tiles_per_sup;
  const int te = tb + a.tiles_per_sup < ntiles ? tb + a.tiles_per_sup : ntiles;
  double J0 = 0.0, J1 = 0.0;
  const double two_scale = 2 * segloss.scale;
  // one observation: y from the tile (chunk order i ^ p), two fma chains over the even / odd registers (= the even / odd chunks or the
  // other way round: the sum of the two commutes), loss and derivative, gradient in list order
  auto entry = [&](int off, double av, int par, int did) {
    const char* yp = lds + off;
    double2 y[C];
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = *reinterpret_cast<const double2*>(yp + ((i * 16) ^ pb));
    double uA = 0.0, uB = 0.0;
#pragma unroll
    for (int i = 0; i < C; i += 2) {
      uA = fma(x[i].x, y[i].x, uA);
      uA = fma(x[i].y, y[i].y, uA);
      uB = fma(x[i + 1].x, y[i + 1].x, uB);
      uB = fma(x[i + 1].y, y[i + 1].y, uB);
    }
    const double dot = uA + uB;
    double L, dL;
    if constexpr (LOSS == 0) { // one QuadLoss descriptor (src/losses.jl:144,146), the formula of the four-lane kernels
      const double dq = dot - av;
      L = segloss.scale * (dq * dq);
      dL = dq * two_scale; // == (2 * d) * scale bit for bit: doubling is exact
    } else if constexpr (loss_mode(LOSS) == 1) {
      loss_both<GRAD, loss_trig(LOSS)>(segloss, dot, av, L, dL);
    } else { // a loss per observation: the 32-byte descriptor from the LDS table behind the tile (glrm_loss layout: kind, dim, scale, p0, p1)
      const char* dp = lds + TILE * KP * 8 + did * 32;
      const int2 kd = *reinterpret_cast<const int2*>(dp);
      const double sc = *reinterpret_cast<const double*>(dp + 8);
      const double2 pp = *reinterpret_cast<const double2*>(dp + 16);
      loss_both<GRAD, loss_trig(LOSS)>(LossDesc{kd.x, sc, pp.x, pp.y}, dot, av, L, dL);
    }
    if (par) J1 += L; else J0 += L;
    if (GRAD) {
#pragma unroll
      for (int i = 0; i < C; ++i) {
        g[i].x = fma(dL, y[i].x, g[i].x);
        g[i].y = fma(dL, y[i].y, g[i].y);
      }
    }
  };
  if constexpr (loss_mode(LOSS) == 2) { // the model's distinct loss descriptors behind the tile (read after the first tile's barriers)
    const int words = a.n_udesc * 8;
    const int* src = reinterpret_cast<const int*>(a.udesc);
    int* dst = reinterpret_cast<int*>(lds + TILE * KP * 8);
    for (int w = threadIdx.x; w < words; w += NW * 64) dst[w] = src[w];
  }
  int64_t pos = 0, end = 0;
  if constexpr (CSR) {
    const int64_t beg = have ? a.ptr[seg] : 0;
    end = have ? a.ptr[seg + 1] : 0;
    pos = have ? lower_bound_idx<1>(a.idx, beg, end, (int64_t)tb * TILE) : 0;
  }
  const bool wb_ok = CSR || wb < la.nwb;
  const int64_t* bp = (CSR || !wb_ok) ? nullptr : la.bptr + wb * (int64_t)(la.ntiles + 1);
#if GLRM_LANE_PREFETCH
  int pf0 = 0, pf1 = 0, pf2 = 0;
#endif
  for (int t = tb; t < te; ++t) {
    const int64_t lo = (int64_t)t * TILE;
    const int64_t hi = lo + TILE < a.n_other ? lo + TILE : a.n_other;
    __syncthreads(); // everybody is done with the previous tile
    dma_tile_all<2, KP / 2, NW, true>(a.other, lo, hi, lds, wave, lane); // unpadded rows: a plain copy by LDS-DMA from all waves
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if GLRM_LANE_PREFETCH
    asm volatile("" ::"v"(pf0), "v"(pf1), "v"(pf2)); // the prefetch registers of the previous tile stay reserved until their loads have landed
#endif
    __syncthreads();
#if GLRM_LANE_PREFETCH
    // The staging above is latency: 143 KB per workgroup, all of it requested at once, then everybody waits (11.5 of 31 us per tile in the
    // column passes, whose factor does not live in L2).  While this tile is consumed, one dword of every 128-byte line of the NEXT tile is
    // requested and dropped: the lines are in the XCD's L2 when the LDS-DMA asks for them.  (vmcnt counts in order: the first wait for stream
    // data issued after these loads also waits for them -- once per tile, against the whole staging latency before.)
    if (t + 1 < te) {
      const int64_t nlo = hi, nhi = nlo + TILE < a.n_other ? nlo + TILE : a.n_other;
      const char* nsrc = reinterpret_cast<const char*>(a.other) + nlo * (KP * 8);
      const int nlines = (int)(nhi - nlo) * (KP * 8) / 128;
      const int l0 = (int)threadIdx.x, l1 = l0 + NW * 64, l2 = l1 + NW * 64;
      if (l0 < nlines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf0) : "v"(nsrc + (int64_t)l0 * 128) : "memory");
      if (l1 < nlines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf1) : "v"(nsrc + (int64_t)l1 * 128) : "memory");
      if (l2 < nlines) asm volatile("global_load_dword %0, %1, off" : "=v"(pf2) : "v"(nsrc + (int64_t)l2 * 128) : "memory");
    }
#endif
    if constexpr (CSR) {
      int e = 0;
      while (pos < end) {
        const int c = a.idx[pos];
        if (c >= (int)hi) break;
        int did = 0;
        if constexpr (loss_mode(LOSS) == 2) did = a.descid[pos];
        entry((c - (int)lo) * (KP * 8), a.vals[pos], e & 1, did);
        ++pos;
        ++e;
      }
    } else {
      int64_t s = wb_ok ? uniform_i64(bp[t]) : 0; // (the same value in every lane of the wave: keep the loop control scalar)
      const int64_t s1 = wb_ok ? uniform_i64(bp[t + 1]) : 0;
      if (s >= s1) continue;
      int32_t off[U], noff[U];
      double av[U], nav[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t q = s + u < s1 ? s + u : s1 - 1;
        off[u] = la.off[q * 64 + lane];
        av[u] = la.val[q * 64 + lane];
        if (s + u >= s1 || !have) off[u] = -1;
      }
      for (; s < s1; s += U) { // (s - first step of the tile) stays a multiple of U, U even: step u of a block sits at position parity u & 1
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t q = s + U + u < s1 ? s + U + u : s1 - 1;
          noff[u] = la.off[q * 64 + lane];
          nav[u] = la.val[q * 64 + lane];
          if (s + U + u >= s1 || !have) noff[u] = -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (off[u] >= 0) {
            if constexpr (loss_mode(LOSS) == 2) entry(off[u] & 0xFFFFF, av[u], u & 1, off[u] >> 20);
            else entry(off[u], av[u], u & 1, 0);
          }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          off[u] = noff[u];
          av[u] = nav[u];
        }
      }
    }
  }
  if (have) {
    double* pp = a.part + ((int64_t)seg * a.nsup + sup) * PSTRIDE;
    if (GRAD) {
#pragma unroll
      for (int i = 0; i < C; ++i) *reinterpret_cast<double2*>(pp + (i ^ p) * 2) = g[i];
    }
    pp[KP] = J0 + J1;
  }
}

// ---- the SELL layout (built once per side at finalize) -------------------------------------------------------------------------

// steps of (wave block, tile) = the longest run of the block's 64 segments inside the tile; one 64-thread workgroup per wave block
static __global__ void __launch_bounds__(64) lane_count_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const int32_t* __restrict__ perm,
                                                               int64_t nslots, int tile, int ntiles, int64_t* __restrict__ cnt) {
  const int lane = threadIdx.x;
  const int64_t wb = blockIdx.x, slot = wb * 64 + lane;
  const bool have = slot < nslots;
  const int64_t seg = have ? (perm ? (int64_t)perm[slot] : slot) : 0;
  int64_t pos = have ? ptr[seg] : 0;
  const int64_t end = have ? ptr[seg + 1] : 0;
  for (int t = 0; t < ntiles; ++t) {
    const int64_t nxt = lower_bound_idx<1>(idx, pos, end, (int64_t)(t + 1) * tile);
    int c = (int)(nxt - pos);
    pos = nxt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_xor(c, d, 64);
      c = o > c ? o : c;
    }
    if (lane == 0) cnt[wb * ntiles + t] = c;
  }
}

// scan[i] = exclusive prefix of cnt (flattened [wave block][tile]) -> bptr[wave block][tile], with the end of the block's last tile behind it
static __global__ void lane_bptr_kernel(const int64_t* __restrict__ scan, const int64_t* __restrict__ cnt, int64_t nwb, int ntiles, int64_t* __restrict__ bptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nwb * (ntiles + 1)) return;
  const int64_t wb = i / (ntiles + 1);
  const int t = (int)(i - wb * (ntiles + 1));
  bptr[i] = t < ntiles ? scan[wb * ntiles + t] : scan[wb * ntiles + ntiles - 1] + cnt[wb * ntiles + ntiles - 1];
}

static __global__ void __launch_bounds__(64) lane_fill_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ vals,
                                                              const uint8_t* __restrict__ descid, const int32_t* __restrict__ perm, int64_t nslots, int tile, int ntiles, int rowbytes,
                                                              const int64_t* __restrict__ bptr, int32_t* __restrict__ off, double* __restrict__ val) {
  const int lane = threadIdx.x;
  const int64_t wb = blockIdx.x, slot = wb * 64 + lane;
  const bool have = slot < nslots;
  const int64_t seg = have ? (perm ? (int64_t)perm[slot] : slot) : 0;
  int64_t pos = have ? ptr[seg] : 0;
  const int64_t end = have ? ptr[seg + 1] : 0;
  const int64_t* bp = bptr + wb * (int64_t)(ntiles + 1);
  for (int t = 0; t < ntiles; ++t) {
    const int64_t lo = (int64_t)t * tile, hi = lo + tile;
    int64_t s = bp[t];
    const int64_t s1 = bp[t + 1];
    for (; s < s1; ++s) {
      int32_t o = -1;
      double v = 0.0;
      if (pos < end) {
        const int c = idx[pos];
        if (c < hi) {
          o = (int32_t)(c - lo) * rowbytes | (descid ? (int32_t)descid[pos] << 20 : 0);
          v = vals[pos];
          ++pos;
        }
      }
      off[s * 64 + lane] = o;
      val[s * 64 + lane] = v;
    }
  }
}

} // namespace glrm
