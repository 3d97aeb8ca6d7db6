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
  // COMPACT form of the stream (views whose padded copy would not fit: C5's column view at its stated size): 2-byte offsets (tile-local
  // index of the staged vector, 0xFFFF = idle lane) in the padded [steps][64] arrangement, the VALUES unpadded -- a step's values are those
  // of its busy lanes in lane order, the steps of a wave block back to back; vptr[wave block][tile] = index of the tile's first value
  // offset word: bits 0-9 the tile-local index, bits 10-15 the lane's rank among the busy lanes of its step; sval[step] = values of the wave
  // block before the step (with vptr[wave block][0] the index of the step's first value): what a lane of a GATHERED wave (FORM 2) finds its
  // value by, where the full grid counts along
  const uint16_t* off16; // nullptr: the padded form above
  const int64_t* vptr;
  const int32_t* sval;
  int ntiles;          // tiles of the opposing factor (stride of bptr minus one)
  int64_t nwb;         // wave blocks the layout holds (a launch rounds its grid up to whole workgroups: blocks beyond have no steps)
  int64_t slot0;       // local id of TiledArgs' segment 0 in the slot space the layout was built on (row sub-ranges: glrm_hip_step_x_range)
  // FORM 2 (trial rounds over the still-searching segments, read out of the SELL layout): the segments class by class (lane_compact_* below)
  const int32_t* inv;    // local segment -> slot of the layout (nullptr: slot = segment + slot0)
  const int32_t* glist;  // [gwaves][64]: the local segment of every lane of the round (-1: idle), lane_compact_fill_kernel
  int64_t gwaves;
};

__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// GRAD: pass 1 (gradient + loss partials at a.own); else a trial pass (loss partials at a.trial for the still-searching segments).
// FORM 0: the SELL layout over the full grid of slots (idle segments masked).
// FORM 1 ("CSR"): the (index, value) pairs come from the segment's own list (one lane walks it: uncoalesced, nothing in flight ahead of its
// use) -- sub-range sweeps the layout does not cover; the fallback of the trial rounds.
// FORM 2 (session r6_25): the trial rounds over the segments STILL SEARCHING, read out of the SELL layout.  A wave takes 64 of them --
// lane l one of class (global id) & 15 == l & 15, so the rotated chunk walk stays bank-conflict free -- and every lane walks the steps of its
// OWN (wave block, tile) at its own slot of the layout: the segments of a wave come from neighbouring wave blocks (the class lists are in
// ascending order), so its loads still share lines (about 12 / fraction-still-searching lines per step against 128 of the CSR form) and
// run U steps ahead like FORM 0's.  All three forms add the same terms in the same order.
template <int KP, int NW, int TILE, int LOSS, bool GRAD, int FORM, bool COMPACT = false>
__global__ void __launch_bounds__(NW * 64, 1) lane_pass_kernel(const TiledArgs a, const LaneArgs la) {
  constexpr bool CSR = FORM == 1;
  static_assert(!COMPACT || (FORM != 1 && loss_mode(LOSS) != 2), "the compact stream: sides without a descriptor id in the offset word");
  constexpr bool CF0 = COMPACT && FORM == 0, CF2 = COMPACT && FORM == 2;
  static_assert(FORM == 0 || FORM == 1 || (FORM == 2 && !GRAD), "the gathered form runs trial rounds only");
  static_assert(KP == 32, "one lane per segment: x, g and y of a segment in one lane's registers -- built for a padded rank of 32");
  constexpr int C = KP / 2;            // 16-byte chunks per vector
  constexpr int U = GRAD ? 2 : 4;      // steps in flight ahead of their use (even: a step's position parity inside its window is u & 1)
  constexpr int PSTRIDE = KP + 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wb = (FORM != 0 ? 0 : la.slot0 / 64) + (int64_t)blockIdx.x * NW + wave; // wave block in the layout's slot space (FORM 2: wave of the round)
  const int64_t slot = wb * 64 + lane;
  const int64_t rel = FORM != 0 ? slot : slot - la.slot0;                                 // slot relative to TiledArgs' segment 0
  const int64_t nslots = CSR ? a.nseg : (a.npass > 0 ? a.npass : a.nseg);
  const int sup = a.sup0 + (int)blockIdx.y;
  bool have = rel >= 0 && rel < nslots;
  int64_t seg = 0;
  int64_t lwb = wb; // wave block and lane of the layout whose steps this lane walks
  int ll = lane;
  if constexpr (FORM == 2) {
    have = wb < la.gwaves;
    seg = have ? (int64_t)la.glist[wb * 64 + lane] : -1; // of class (global id) & 15 == lane & 15: lane_compact_fill_kernel
    have = seg >= 0;
    if (!have) seg = 0;
    const int64_t ls = la.inv ? (int64_t)la.inv[seg] : seg + la.slot0;
    have = have && ls >= 0 && (ls >> 6) < la.nwb;
    lwb = have ? ls >> 6 : 0;
    ll = (int)(ls & 63);
  } else {
    seg = (have && a.segperm) ? (int64_t)a.segperm[rel] : (have ? rel : 0);
  }
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
  // CSR form: the lane's list in batches of four entries (opposing index -- INT_MAX behind the end of the list --, value, descriptor id):
  // batch A is being consumed (cursor qk), batch B is in flight; a batch's four loads go out back to back, so a line of the list is
  // fetched once per batch instead of once per entry
  int qa[4], qb[4], qda[4], qdb[4], qk = 0;
  double qva[4], qvb[4];
  int64_t qpos = pos; // list position of A[0]
  auto qload = [&](int (&c)[4], double (&v)[4], int (&d)[4], int64_t p0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool more = CSR && p0 + j < end;
      c[j] = more ? a.idx[p0 + j] : 0x7FFFFFFF;
      v[j] = more ? a.vals[p0 + j] : 0.0;
      d[j] = 0;
      if constexpr (CSR && loss_mode(LOSS) == 2) d[j] = more ? (int)a.descid[p0 + j] : 0;
    }
  };
  qload(qa, qva, qda, qpos);
  qload(qb, qvb, qdb, qpos + 4);
  const bool wb_ok = CSR || (FORM == 2 ? have : wb < la.nwb);
  const int64_t* bp = (CSR || !wb_ok) ? nullptr : la.bptr + lwb * (int64_t)(la.ntiles + 1);
  // steps [s0, s1) of the current tile in the layout (FORM 0: the same in every lane of the wave -- loop control stays scalar; FORM 2: the
  // lane's own wave block); the end of the next tile is requested one tile ahead
  int64_t s0 = 0, s1 = 0;
  if constexpr (!CSR) {
    if (bp) {
      s0 = bp[tb];
      s1 = bp[tb + 1];
    }
    if constexpr (FORM == 0) {
      s0 = uniform_i64(s0);
      s1 = uniform_i64(s1);
    }
  }
  // COMPACT: the values of a wave block are unpadded, so a lane finds its value of a step at (values before the step) + (busy lanes below
  // it) -- a running wave-uniform count and a ballot; the offsets run one batch ahead of the values, and the first batch of a tile is
  // requested a whole tile ahead, so that no round trip to the stream stands between a tile's staging and its first step
  int64_t vrun = 0;
  int rawT[U];
  int64_t vblk = 0; // CF2: index of the lane's wave block's first value
  if constexpr (CF2) {
    if (bp) vblk = la.vptr[lwb * (int64_t)(la.ntiles + 1)];
  }
  if constexpr (CF0) {
    if (bp) vrun = uniform_i64(la.vptr[lwb * (int64_t)(la.ntiles + 1) + tb]);
    const int n0 = (int)(s1 - s0);
#pragma unroll
    for (int u = 0; u < U; ++u) rawT[u] = (bp && u < n0) ? (int)la.off16[(s0 + u) * 64 + lane] : 0xFFFF;
  }
  auto below = [&](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
#if GLRM_LANE_PREFETCH
  int pf0 = 0, pf1 = 0, pf2 = 0;
#endif
  for (int t = tb; t < te; ++t) {
    const int64_t lo = (int64_t)t * TILE;
    const int64_t hi = lo + TILE < a.n_other ? lo + TILE : a.n_other;
    int64_t s2 = s1;
    if constexpr (!CSR) {
      if (bp && t + 1 < te) s2 = bp[t + 2];
    }
    __syncthreads(); // everybody is done with the previous tile
    dma_tile_all<2, KP / 2, NW, true>(a.other, lo, hi, lds, wave, lane); // unpadded rows: a plain copy by LDS-DMA from all waves
    // the first U steps of the tile are requested behind its staging: they have landed when the tile has (before session r6_25 they were
    // requested after the barrier: one exposed round trip to the stream per tile)
    int32_t off[U], noff[U];
    double av[U], nav[U];
    int n = 0;
    int raw1[U]; // COMPACT: the offsets of the batch after the current one
    int sv1[U], rw0[U], sb0[U]; // CF2: the batch's step bases; the first batch of the tile
    if constexpr (CF2) {
      n = (int)(s1 - s0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool v = have && u < n, w = have && U + u < n;
        rw0[u] = v ? (int)la.off16[(s0 + u) * 64 + ll] : 0xFFFF;
        sb0[u] = v ? la.sval[s0 + u] : 0;
        raw1[u] = w ? (int)la.off16[(s0 + U + u) * 64 + ll] : 0xFFFF;
        sv1[u] = w ? la.sval[s0 + U + u] : 0;
      }
    } else if constexpr (CF0) {
      n = (int)(s1 - s0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rawT[u];
        const unsigned long long m = __ballot(r != 0xFFFF);
        const bool v = have && r != 0xFFFF;
        off[u] = v ? (r & 1023) * (KP * 8) : -1;
        av[u] = v ? la.val[vrun + below(m)] : 0.0;
        vrun += __popcll(m);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) raw1[u] = (bp && U + u < n) ? (int)la.off16[(s0 + U + u) * 64 + lane] : 0xFFFF;
    } else if constexpr (!CSR) {
      n = (int)(s1 - s0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool v = have && u < n;
        off[u] = v ? la.off[(s0 + u) * 64 + ll] : -1;
        av[u] = v ? la.val[(s0 + u) * 64 + ll] : 0.0;
      }
    }
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
    if constexpr (CF2) {
      // (the first batch's values wait for its offsets: one exposed round trip per tile -- these are rounds, not passes)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = rw0[u];
        const bool v = r != 0xFFFF;
        off[u] = v ? (r & 1023) * (KP * 8) : -1;
        av[u] = v ? la.val[vblk + sb0[u] + (r >> 10)] : 0.0;
      }
      for (int i = 0; __any(i < n) != 0; i += U) {
        int raw2[U], sv2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = raw1[u];
          const bool v = r != 0xFFFF;
          noff[u] = v ? (r & 1023) * (KP * 8) : -1;
          nav[u] = v ? la.val[vblk + sv1[u] + (r >> 10)] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool w = have && i + 2 * U + u < n;
          raw2[u] = w ? (int)la.off16[(s0 + i + 2 * U + u) * 64 + ll] : 0xFFFF;
          sv2[u] = w ? la.sval[s0 + i + 2 * U + u] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (off[u] >= 0) entry(off[u], av[u], u & 1, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          off[u] = noff[u];
          av[u] = nav[u];
          raw1[u] = raw2[u];
          sv1[u] = sv2[u];
        }
      }
      s0 = s1;
      s1 = s2;
    } else if constexpr (CF0) {
      s2 = uniform_i64(s2);
      const int nn = (int)(s2 - s1);
#pragma unroll
      for (int u = 0; u < U; ++u) rawT[u] = (bp && t + 1 < te && u < nn) ? (int)la.off16[(s1 + u) * 64 + lane] : 0xFFFF;
      for (int i = 0; i < n; i += U) {
        int raw2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { // batch i + U: its offsets have landed, its values are requested
          const int r = raw1[u];
          const unsigned long long m = __ballot(r != 0xFFFF);
          const bool v = have && r != 0xFFFF;
          noff[u] = v ? (r & 1023) * (KP * 8) : -1;
          nav[u] = v ? la.val[vrun + below(m)] : 0.0;
          vrun += __popcll(m);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) raw2[u] = (bp && i + 2 * U + u < n) ? (int)la.off16[(s0 + i + 2 * U + u) * 64 + lane] : 0xFFFF;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (off[u] >= 0) entry(off[u], av[u], u & 1, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          off[u] = noff[u];
          av[u] = nav[u];
          raw1[u] = raw2[u];
        }
      }
      s0 = s1;
      s1 = s2;
    } else if constexpr (CSR) {
      // (session r6_43/44; before: index, value and descriptor id were requested when the step needed them.  Measured the same: what this form
      // pays is the address translation of 512 lists far apart, see glrm_run_lane)
      int e = 0;
      for (;;) {
        const int hc = qk == 0 ? qa[0] : qk == 1 ? qa[1] : qk == 2 ? qa[2] : qa[3];
        const bool go = hc < (int)hi;
        if (__any(go) == 0) break;
        if (go) {
          const double hv = qk == 0 ? qva[0] : qk == 1 ? qva[1] : qk == 2 ? qva[2] : qva[3];
          const int hd = qk == 0 ? qda[0] : qk == 1 ? qda[1] : qk == 2 ? qda[2] : qda[3];
          entry((hc - (int)lo) * (KP * 8), hv, e & 1, hd);
          ++e;
          if (++qk == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              qa[j] = qb[j];
              qva[j] = qvb[j];
              qda[j] = qdb[j];
            }
            qload(qb, qvb, qdb, qpos + 8);
            qpos += 4;
            qk = 0;
          }
        }
      }
    } else {
      // (i stays a multiple of U, U even: step i + u of a segment's window sits at position parity u & 1)
      for (int i = 0; FORM == 0 ? i < n : __any(i < n) != 0; i += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool v = have && i + U + u < n;
          noff[u] = v ? la.off[(s0 + i + U + u) * 64 + ll] : -1;
          nav[u] = v ? la.val[(s0 + i + U + u) * 64 + ll] : 0.0;
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
      if constexpr (FORM == 0) s2 = uniform_i64(s2);
      s0 = s1;
      s1 = s2;
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

// ---- tail rounds: a wave per (segment, super-tile), no tile ---------------------------------------------------------------------------
// A trial round over a few segments still pays a lane's serial walk through every tile of the opposing factor (~1.5 ms for a row of C5
// whatever the number of rows, 8-10 such rounds per X half-step).  Here a WAVE takes one segment's entries inside one super-tile: lane l
// evaluates the entries l, l + 64, ... with the vectors read straight from memory -- the same two fma chains over the chunks i ^ p, the same
// loss formula, so the same term bit for bit -- and the terms are then added as the passes add them: two sequential sums over the entries at
// even / odd position inside their tile window, in list order (the terms are parked in LDS, compacted by parity, CAP at a time; lanes 0 and 1
// run the two chains), one partial per (segment, super-tile) like the passes'.
template <int KP, int LOSS, int CAP>
__global__ void __launch_bounds__(128) lane_tail_kernel(const TiledArgs a, const int32_t* __restrict__ list, int64_t nwork, int tile) {
  constexpr int C = KP / 2, PSTRIDE = KP + 2, OCAP = CAP / 2 + 64;
  __shared__ double ebuf[2][CAP], obuf[2][OCAP];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w = (int64_t)blockIdx.x * 2 + wv;
  if (w >= nwork) return; // (wave-uniform)
  const int64_t seg = list[w / a.nsup];
  const int sup = (int)(w % a.nsup);
  const int64_t gseg = a.own_offset + seg;
  const int p = (int)(gseg & (C - 1));
  const double2* xp = reinterpret_cast<const double2*>(a.trial + seg * (int64_t)KP);
  double2 x[C];
#pragma unroll
  for (int i = 0; i < C; ++i) x[i] = xp[i ^ p];
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != 2) segloss = load_loss(a.losses, a.loss_by_segment ? gseg : 0);
  int64_t beg = a.ptr[seg], end = a.ptr[seg + 1];
  if (a.nsup > 1) { // the segment's entries inside this super-tile
    const int64_t lo = (int64_t)sup * a.tiles_per_sup * tile, hi = lo + (int64_t)a.tiles_per_sup * tile;
    const int64_t b0 = lower_bound_idx<1>(a.idx, beg, end, lo);
    end = lower_bound_idx<1>(a.idx, b0, end, hi);
    beg = b0;
  }
  const int64_t len = end - beg;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane)), upto = below | (1ull << lane);
  int ne = 0, no = 0, prevtile = -1; // terms parked by parity; tile of the entry before the batch
  int64_t wfirst = 0;                // first entry of the window in progress
  double J = 0.0;                    // lane 0: the sum over even positions, lane 1: over odd ones
  auto flush = [&]() {
    __threadfence_block();
    if (lane == 0) for (int i = 0; i < ne; ++i) J += ebuf[wv][i];
    if (lane == 1) for (int i = 0; i < no; ++i) J += obuf[wv][i];
    __threadfence_block();
    ne = 0;
    no = 0;
  };
  for (int64_t base = 0; base < len; base += 64) {
    if (ne + 64 > CAP || no + 64 > OCAP) flush();
    const int64_t e = base + lane;
    const bool valid = e < len;
    const int c = valid ? a.idx[beg + e] : 0;
    const double av = valid ? a.vals[beg + e] : 0.0;
    int did = 0;
    if constexpr (loss_mode(LOSS) == 2) did = valid ? (int)a.descid[beg + e] : 0;
    const int tl = valid ? c / tile : 0x7FFFFFFF;
    int prev = __shfl_up(tl, 1, 64);
    if (lane == 0) prev = prevtile;
    const unsigned long long bm = __ballot(valid && tl != prev); // entries that open a window
    const unsigned long long mine = bm & upto;
    const int64_t wstart = mine ? base + 63 - __clzll((long long)mine) : wfirst;
    const int par = (int)((e - wstart) & 1);
    double L = 0.0;
    if (valid) {
      const double2* yp = reinterpret_cast<const double2*>(a.other + (int64_t)c * KP);
      double2 y[C];
#pragma unroll
      for (int i = 0; i < C; ++i) y[i] = yp[i ^ p];
      double uA = 0.0, uB = 0.0;
#pragma unroll
      for (int i = 0; i < C; i += 2) {
        uA = fma(x[i].x, y[i].x, uA);
        uA = fma(x[i].y, y[i].y, uA);
        uB = fma(x[i + 1].x, y[i + 1].x, uB);
        uB = fma(x[i + 1].y, y[i + 1].y, uB);
      }
      const double dot = uA + uB;
      double dL;
      if constexpr (LOSS == 0) {
        const double dq = dot - av;
        L = segloss.scale * (dq * dq);
      } else if constexpr (loss_mode(LOSS) == 1) {
        loss_both<false, loss_trig(LOSS)>(segloss, dot, av, L, dL);
      } else {
        loss_both<false, loss_trig(LOSS)>(load_loss(a.udesc, did), dot, av, L, dL);
      }
    }
    const unsigned long long em = __ballot(valid && par == 0), om = __ballot(valid && par == 1);
    if (valid) {
      if (par) obuf[wv][no + __popcll(om & below)] = L;
      else ebuf[wv][ne + __popcll(em & below)] = L;
    }
    ne += __popcll(em);
    no += __popcll(om);
    if (bm) wfirst = base + 63 - __clzll((long long)bm);
    const int64_t left = len - base - 1;
    prevtile = __shfl(tl, (int)(left < 63 ? left : 63), 64);
  }
  flush();
  const double J1 = __shfl(J, 1, 64);
  if (lane == 0) a.part[((int64_t)seg * a.nsup + sup) * PSTRIDE + KP] = J + J1;
}

// ---- the SELL layout (built once per side at finalize) -------------------------------------------------------------------------

// steps of (wave block, tile) = the longest run of the block's 64 segments inside the tile; one 64-thread workgroup per wave block
static __global__ void __launch_bounds__(64) lane_count_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const int32_t* __restrict__ perm,
                                                               int64_t nslots, int tile, int ntiles, int64_t* __restrict__ cnt, int64_t* __restrict__ vcnt) {
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
    if (vcnt) { // the compact form: observations of the (wave block, tile), not only its steps
      int sum = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d, 64);
      if (lane == 0) vcnt[wb * ntiles + t] = sum;
    }
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

// the compact form of the stream (LaneArgs::off16 / vptr): padded 2-byte offsets, unpadded values
static __global__ void __launch_bounds__(64) lane_fill_compact_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ vals,
                                                                      const int32_t* __restrict__ perm, int64_t nslots, int tile, int ntiles, const int64_t* __restrict__ bptr,
                                                                      const int64_t* __restrict__ vptr, uint16_t* __restrict__ off16, int32_t* __restrict__ sval,
                                                                      double* __restrict__ val) {
  const int lane = threadIdx.x;
  const int64_t wb = blockIdx.x, slot = wb * 64 + lane;
  const bool have = slot < nslots;
  const int64_t seg = have ? (perm ? (int64_t)perm[slot] : slot) : 0;
  int64_t pos = have ? ptr[seg] : 0;
  const int64_t end = have ? ptr[seg + 1] : 0;
  const int64_t* bp = bptr + wb * (int64_t)(ntiles + 1);
  const int64_t v0 = vptr[wb * (int64_t)(ntiles + 1)];
  int64_t vrun = v0;
  for (int t = 0; t < ntiles; ++t) {
    const int64_t lo = (int64_t)t * tile, hi = lo + tile;
    const int64_t s1 = bp[t + 1];
    for (int64_t s = bp[t]; s < s1; ++s) {
      bool has = false;
      int o = 0xFFFF;
      double v = 0.0;
      if (pos < end) {
        const int c = idx[pos];
        if (c < hi) {
          has = true;
          o = (int)(c - lo);
          v = vals[pos];
          ++pos;
        }
      }
      const unsigned long long m = __ballot(has);
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      off16[s * 64 + lane] = (uint16_t)(has ? (o | (rank << 10)) : 0xFFFF); // (tile-local index below 1 024: the tile holds 560 vectors)
      if (has) val[vrun + rank] = v;
      if (lane == 0) sval[s] = (int32_t)(vrun - v0);
      vrun += __popcll(m);
    }
  }
}

// ---- the still-searching segments, wave by wave (FORM 2 of the pass) ---------------------------------------------------------------
// glist[w][l] = the local segment lane l of wave w of the round works on (-1: idle), gtotal[0] = the waves of the round.  Lane l takes a
// segment of class (global id) & 15 == l & 15 (bank-conflict free tile reads), and the lanes of a wave come from ONE chunk of 1 024
// consecutive segments = 16 wave blocks of the layout, so that its loads share lines.  (First version, session r6_25: one ascending list per
// class, wave w = entries 4w .. 4w + 3 of every list -- the lists drift apart by the square root of their length, tens of wave blocks at
// 1M rows, and a wave's 64 lanes sat in ~16 different blocks: 25-40 ps per searching observation against 9.85 of the full grid.)  A chunk
// is one wavefront's work: lane (sub = lane >> 4, cs = lane & 15) owns the segments chunk x 1 024 + sub x 256 + j x 16 + cs, j = 0 .. 15 --
// one class, ascending; waves of the chunk = the longest class list / 4, rounded up (shorter classes leave idle lanes: binomial imbalance,
// ~20 % at half the segments searching).  Three small launches per round: waves per chunk, exclusive scan, fill.
constexpr int LANE_CC = 1024;

// (perm != nullptr: the chunks are chunks of SLOTS of a permuted layout -- dealt out class by class, so slot s holds a segment of class s & 15)
__device__ __forceinline__ unsigned lane_chunk_mask(const int32_t* __restrict__ active, int64_t nitems, int64_t chunk, int lane, const int32_t* __restrict__ perm) {
  const int64_t i0 = chunk * LANE_CC + (int64_t)(lane >> 4) * 256 + (lane & 15);
  unsigned m = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t i = i0 + j * 16;
    if (i < nitems && active[perm ? (int64_t)perm[i] : i] != 0) m |= 1u << j;
  }
  return m;
}

// waves the chunk needs: the longest of its 16 class lists / 4, rounded up (the same value in every lane)
__device__ __forceinline__ int lane_chunk_waves(int cnt) {
  int c = cnt + __shfl_xor(cnt, 16, 64);
  c += __shfl_xor(c, 32, 64); // the class total, in all four lanes of the class
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    const int o = __shfl_xor(c, d, 64);
    c = o > c ? o : c;
  }
  return (c + 3) / 4;
}

static __global__ void __launch_bounds__(256) lane_compact_count_kernel(const int32_t* __restrict__ active, int64_t nseg, int64_t nchunks, const int32_t* __restrict__ perm,
                                                                        int32_t* __restrict__ nwv) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  const int nw = lane_chunk_waves(__popc(lane_chunk_mask(active, nseg, chunk, lane, perm)));
  if (lane == 0) nwv[chunk] = nw;
}

// wbase = exclusive scan of nwv over the chunks, total[0] = the sum (one workgroup)
static __global__ void __launch_bounds__(1024) lane_compact_scan_kernel(const int32_t* __restrict__ nwv, int64_t nchunks, int32_t* __restrict__ wbase, int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < nchunks; c0 += 1024) {
    const int64_t ch = c0 + threadIdx.x;
    const int v = ch < nchunks ? nwv[ch] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (ch < nchunks) wbase[ch] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = carry;
}

static __global__ void __launch_bounds__(256) lane_compact_fill_kernel(const int32_t* __restrict__ active, int64_t nseg, int64_t nchunks, int off16, const int32_t* __restrict__ perm,
                                                                       const int32_t* __restrict__ wbase, int32_t* __restrict__ glist) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (chunk >= nchunks) return;
  const unsigned m = lane_chunk_mask(active, nseg, chunk, lane, perm);
  const int cnt = __popc(m);
  const int nw = lane_chunk_waves(cnt);
  const int64_t w0 = wbase[chunk];
  const int csk = ((lane & 15) - off16) & 15; // the owner lanes of the class this lane serves: (global id) & 15 == lane & 15
  for (int w = 0; w < nw; ++w) {
    int rem = 4 * w + (lane >> 4), seg = -1;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const int c = __shfl(cnt, sub * 16 + csk, 64);
      const unsigned mm = (unsigned)__shfl((int)m, sub * 16 + csk, 64);
      if (seg < 0 && rem >= 0) {
        if (rem < c) {
          int j = 0, r = rem;
          for (; j < 16; ++j)
            if ((mm >> j) & 1u) {
              if (r == 0) break;
              --r;
            }
          seg = (int)(chunk * LANE_CC) + sub * 256 + j * 16 + csk;
          if (perm) seg = perm[seg];
          rem = -1;
        } else {
          rem -= c;
        }
      }
    }
    glist[(w0 + w) * 64 + lane] = seg;
  }
}

// Few segments still searching (the tail rounds of a line search: a few percent down to a single row): a wave's 64 lanes sit in 64 different
// wave blocks of the layout whatever the order, so nothing is gained by keeping a chunk's segments together -- and the chunked lists above
// leave most lanes idle (794 searching rows of 1M: 546 waves).  Here the round's waves are packed from the compact list of the previous
// decide kernel: entry e of class c = (global id) & 15 with rank r among the list's entries of that class goes to wave r / q, lane
// (r % q) x 16 + c.  q (1 .. 4 entries per class and wave) spreads a short list over the chip: such a round is bound by the tile staging
// and the line requests of ITS workgroups, not by lanes -- 44 000 rows of 1M packed into full waves (90 workgroups, every lane its own
// line): 3.2 ms; the same rows in a third-filled waves on 244 workgroups: 1.7 ms (session r6_29).  One workgroup; ranks by ballots per
// class (list order: deterministic), idle lanes = -1.  total[0] = the waves, or minus that number when glist (cap waves) cannot hold them.
static __global__ void __launch_bounds__(1024) lane_compact_list_kernel(const int32_t* __restrict__ list, int nact, int off16, int q, int cap, int32_t* __restrict__ glist,
                                                                        int32_t* __restrict__ total) {
  __shared__ int wcnt[16][16]; // [wave][class]
  __shared__ int nwaves;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int mine = 0; // lane k < 16: entries of class k this wave has met
  for (int e0 = wave * 64; e0 < nact; e0 += 1024) {
    const int e = e0 + lane;
    const int cls = e < nact ? ((list[e] + off16) & 15) : -1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int n = __popcll(__ballot(cls == k));
      if (lane == k) mine += n;
    }
  }
  if (lane < 16) wcnt[wave][lane] = mine;
  __syncthreads();
  if (threadIdx.x < 16) { // exclusive prefix over the waves, per class; the class total decides the waves of the round
    int run = 0, mx = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = wcnt[w][threadIdx.x];
      wcnt[w][threadIdx.x] = run;
      run += c;
    }
    mx = (run + q - 1) / q;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int o = __shfl_xor(mx, d, 64);
      mx = o > mx ? o : mx;
    }
    if (threadIdx.x == 0) {
      nwaves = mx <= cap ? mx : -1;
      total[0] = mx <= cap ? mx : -mx;
    }
  }
  __syncthreads();
  if (nwaves < 0) return;
  for (int i = threadIdx.x; i < nwaves * 64; i += 1024) glist[i] = -1;
  __threadfence();
  __syncthreads();
  int base = lane < 16 ? wcnt[wave][lane] : 0; // lane k: rank of this wave's next entry of class k
  for (int e0 = wave * 64; e0 < nact; e0 += 1024) {
    const int e = e0 + lane;
    const int seg = e < nact ? list[e] : 0;
    const int cls = e < nact ? ((seg + off16) & 15) : -1;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned long long b = __ballot(cls == k);
      const int start = __shfl(base, k, 64);
      if (cls == k) {
        const int r = start + __popcll(b & below);
        glist[(int64_t)(r / q) * 64 + (r % q) * 16 + k] = seg;
      }
      if (lane == k) base += __popcll(b);
    }
  }
}

// inv[perm[slot]] = slot (inv preset to -1: segments outside the layout -- diverted long columns)
static __global__ void lane_inv_kernel(const int32_t* __restrict__ perm, int64_t nslots, int32_t* __restrict__ inv) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nslots) inv[perm[s]] = (int32_t)s;
}

} // namespace glrm
