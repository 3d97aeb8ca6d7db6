// glrm_tiled.hpp -- LDS-tiled sweeps: the opposing factor is staged through LDS tile by tile and every
// lane group owns ONE segment (row or column) whose observations it walks in list order.
//
// Why: in the gather sweeps (glrm_hip.hip) every observation fetches its own k-vector of the opposing
// factor from L2 / Infinity Cache / HBM (2 x 128-byte lines at k=32), which is what bounds them
// (profiles/r01_baseline_pmc_summary.md).  Here a workgroup of NW waves owns SPB = NW*64/G segments; it
// loops over tiles of TILE consecutive opposing vectors, copies the tile into LDS once (coalesced 16-byte
// loads) and every group consumes the observations of its segment that fall into the tile from LDS.  A
// factor vector is thus read from memory once per (workgroup, pass) instead of once per observation, and
// the per-observation traffic left is the 12-byte (index, value) stream.
//
// Requirements: the segment's index list is ordered by tile -- the entries of tile t come before those of tile t+1; a
// non-decreasing list (`findall` / sparse input, src/glrm.jl:46-48) is the common case.  Checked at create, otherwise the
// gather sweeps are used.  Duplicates are fine.
//
// Summation order: a group accumulates its segment's losses and gradient sequentially in the order of the engine's private
// list (the caller's list order unless the engine had to tile-sort or kind-group it) --
// the reference's own order (src/algorithms/proxgrad.jl:122-132, src/evaluate_fit.jl:28-32).  The column
// sweep additionally splits the rows into NSUP super-tiles whose partial sums are added in order; NSUP
// depends only on (m, TILE), never on the shard layout, so results stay independent of the GPU count.
#pragma once

#include "glrm_device.hpp"

namespace glrm {

struct TiledArgs {
  int64_t nseg;          // local segments
  const int64_t* ptr;    // nseg+1
  const int32_t* idx;    // non-decreasing inside every segment
  const double* vals;
  double* own;           // factor being updated (global array, ld = KP)
  int64_t own_offset;    // global id of local segment 0
  const double* other;   // opposing factor (global array, ld = KP)
  int64_t n_other;       // number of opposing vectors
  double* alpha;         // per local segment
  double* obj;           // per GLOBAL segment (nullable)
  const glrm_loss* losses;
  int loss_by_segment;
  const glrm_reg* regs;
  int reg_single;
  int k;
  double min_stepsize;
  int32_t* trials;
  int32_t* accepts;
  // column sweep (split over super-tiles)
  int nsup;              // number of super-tiles
  int tiles_per_sup;     // tiles per super-tile
  double* part;          // [nseg][nsup][KP+2]: partial gradient (KP), partial loss sum, pad
  double* gsum;          // [nseg][KP]   reduced gradient (kept for later trials)
  double* trial;         // [nseg][KP]   current trial point
  double* jold;          // [nseg]
  double* jloss;         // [nseg] nullable: col_reduce stores the loss sum at the current point (dense path, quad_gram trials)
  int32_t* active;       // [nseg] 1 while the segment's line search is still running
  int32_t* ntrial;       // [nseg] trials taken in this sweep
  unsigned int* nactive; // device counter of still-active segments
  int eval_only;         // col_reduce: obj = sum of losses, nothing else
  int64_t dense_len;     // ptr == nullptr (dense problem): every segment has this many observations
  double fixed_alpha;    // > 0: one prox-gradient step with this global step size, no line search
  // heterogeneous row sweep (a loss descriptor per observation): the DISTINCT descriptors of the model (at most 256) sit in LDS behind
  // the tile and every entry of the row view carries the id of its column's descriptor, so the step reads its descriptor from LDS
  // with the batch instead of waiting for a dependent global load of losses[column] (nullptr: table lookups)
  const uint8_t* descid;  // per entry of the view
  const glrm_loss* udesc; // n_udesc distinct descriptors
  int n_udesc;
  // phase-aligned gather passes (tiled_col_pass_kernel<..., L2 = true>): the launch covers segments [seg_begin, seg_begin + nseg_slice)
  // of super-tile sup_fixed
  int64_t seg_begin, nseg_slice;
  int sup_fixed;
  int nsup_launch;       // > 0: super-tiles this launch covers (grid y); 0 = all nsup
  int sup0;              // LDS-tiled / lane column passes: the launch covers super-tiles sup0 .. sup0 + gridDim.y - 1 (arrival order, round 6)
  const int32_t* segperm; // lane-group slot -> local segment (nullptr = identity).  Which segment a group works on changes no sum:
                          // columns are handed out sorted by (loss kind, length), rows by length, so that the 16 groups of a wave
                          // evaluate the same loss formula and finish their lists together.
  // Line-search rounds over the segments that are STILL SEARCHING only (round 4).  A workgroup stages every tile of the opposing factor
  // per pass whether one of its 256 segments needs the pass or all of them: with the search inside the kernel (rows) or a pass over
  // every workgroup that holds an active segment (columns), a few percent of rejected first trials made EVERY workgroup run another
  // pass (C2: 4 % of the rows, 1.4 % of the columns; C5: the tail of rows on their 3rd .. 6th trial).  Whoever leaves a segment searching
  // appends it to `actlist_out` (slot = the old value of *nactive); the next trial pass runs over that list (`segperm` = the list,
  // nseg = its length: ceil(count / 256) workgroups) and the decide kernel reads it as `actlist_in`.  Which slot a segment sits in
  // changes no sum (see segperm), so the bits are those of the uncompacted rounds.
  // Phase-aligned column passes on skewed data (round 5): the pass launches cover the first `npass` slots of `segperm` (the segments
  // below `long_from` observations, longest first, so that the groups of a launch slice walk lists of like length); segments of at
  // least `long_from` observations are swept by the 8-wave gather sweep beside the passes and are skipped by col_reduce / col_decide.
  int64_t npass;              // 0 = every segment
  int64_t long_from;          // 0 = no segment is diverted
  int stagger;                // > 0: workgroup i of the 32 an XCD holds at a time starts i x stagger x ~0.2 us late (tile_stagger below)
  int32_t* actlist_out;       // nullable
  const int32_t* actlist_in;  // col_decide_kernel: nullable; the segments to decide (nact_in of them) instead of all nseg
  int64_t nact_in;
};

// v from lane (lane ^ X) for X = 4 or 8 (ds_swizzle bit-mask mode: and 0x1F, or 0, xor X; no LDS memory touched)
template <int X>
__device__ __forceinline__ double swizzle_xor_f64(double v) {
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), (X << 10) | 0x1F);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), (X << 10) | 0x1F);
  return __hiloint2double(hi, lo);
}

// lane gets v from lane (u0 + (lane & 1)) of its group: the observation "owned" by the lane's parity
template <int G>
__device__ __forceinline__ double pair_bcast_f64(double v, int u0, int lane) {
  if constexpr (G == 4) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (u0 == 0) { // quad_perm:[0,1,0,1]
      lo = __builtin_amdgcn_mov_dpp(lo, 0x44, 0xF, 0xF, false);
      hi = __builtin_amdgcn_mov_dpp(hi, 0x44, 0xF, 0xF, false);
    } else { // quad_perm:[2,3,2,3]
      lo = __builtin_amdgcn_mov_dpp(lo, 0xEE, 0xF, 0xF, false);
      hi = __builtin_amdgcn_mov_dpp(hi, 0xEE, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
  } else {
    return __shfl(v, (lane & ~(G - 1)) + u0 + (lane & 1), 64);
  }
}

// ---- conflict-free tile reads in the column passes (G = 4 or 8 lanes, four 16-byte chunks per lane) ---------------------------
// ds_read_b128 serves a wave in four LDS cycles of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS) -- i.e. four lane groups (G = 4) per cycle, each reading 64 contiguous bytes of ITS row.  With rows
// padded by 16 B those four 16-bank windows start at random banks and overlap: 52 % of the LDS cycles of the C2 sweeps were conflict
// cycles, and random row reads reach 63 TB/s against 118 TB/s conflict-free (tools/ubench_lanerow.hip,
// profiles/r02_ubench_lanerow.txt).  With ROT the staged rows are NOT padded (every row starts at bank 0) and a lane group walks its
// row's chunks in the order i ^ p, p = ((global segment id) & 7) >> 1: the four groups of an LDS cycle sit at positions {0,3,5,6}
// or {1,2,4,7} of their half wave, whose p are all different when segments occupy the slots in order, so at every instruction they
// read four DIFFERENT bank quarters -- SQ_LDS_BANK_CONFLICT = 0, whatever rows they read.  Register i of a lane then holds chunk
// i ^ p of x, g and y alike (dot products and axpys pair equal registers; loads, stores and masks go by the chunk id).  p depends
// on the segment only, never on where it runs, so sums are formed in the same order under any shard layout or slot permutation (a
// slot order that breaks the pattern costs conflicts, not bits; make_segperm keeps the pattern).
// Measured (profiles/r02_rot_pmc_c2.md, r02_rot_ab.txt; same box A/B): conflicts 1.43e9 -> 0 and LDS-busy cycles halved in every tiled
// kernel, SQ_WAIT_INST_LDS / 20 -- but the sweeps are not LDS-bound.  Row sweeps: QuadLoss 8.56 vs 8.48 ms, heterogeneous 407 vs 387 ms
// (three more address XORs per observation, a few more spilled registers).  Column passes: QuadLoss 9.80 vs 9.67 ms, heterogeneous
// 47.96 vs 50.08 ms.  So only the column passes of models with non-quadratic losses (LOSS != 0) use it.
#ifndef GLRM_TILE_ROT
#define GLRM_TILE_ROT 1
#endif
template <int G, int R>
constexpr bool tile_rot() { return GLRM_TILE_ROT && (G == 4 || G == 8) && R == 8; }
__host__ __device__ __forceinline__ int tile_rot_of(int64_t gseg) { return ((int)gseg & 7) >> 1; }

template <int G, int R>
constexpr int tile_row_bytes() { return G * R * 8 + 16; } // LDS budget per staged row: +16 B pad, consecutive rows start 4 banks apart
template <int G, int R, bool ROT>
constexpr int tile_row_stride() { return ROT ? G * R * 8 : tile_row_bytes<G, R>(); } // ROT: unpadded rows inside the same budget

// Copy opposing vectors [lo, hi) into LDS (coalesced 16-byte loads, padded rows).
template <int G, int R, int NT, bool ROT = false>
__device__ __forceinline__ void stage_tile(const double* __restrict__ other, int64_t lo, int64_t hi, char* lds) {
  constexpr int KPB = G * R * 8, ROWB = tile_row_stride<G, R, ROT>();
  const char* src = reinterpret_cast<const char*>(other) + lo * KPB;
  const int total = (int)(hi - lo) * KPB;
  constexpr int U = 4; // 4 x 16-byte loads in flight per thread before the LDS writes
  for (int base = 0; base < total; base += NT * 16 * U) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { // unconditional loads (clamped), predicated writes
      int off = base + (u * NT + (int)threadIdx.x) * 16;
      off = off < total ? off : total - 16;
      v[u] = *reinterpret_cast<const double2*>(src + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int off = base + (u * NT + (int)threadIdx.x) * 16;
      if (off < total) {
        const int row = off / KPB, col = off - row * KPB;
        *reinterpret_cast<double2*>(lds + row * ROWB + col) = v[u];
      }
    }
  }
}

// ---- double-buffered tiles (LW > 0 loader waves) ------------------------------------------------------------------------
// Measured (tools/gpu_r2l.sh): staging is 20 % (C2) to 37 % (1M x 50k) of an LDS-tiled sweep and fully exposed -- every wave loads,
// waits, writes LDS and meets two barriers per tile before anyone computes.  With LW > 0 the first LW waves of the workgroup are
// LOADERS: they bring the next HALF tile into the other half of LDS with LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write
// pass) while the remaining waves consume the current one; one barrier per half tile ("next one landed, everybody done with this
// one").  Ordinary global loads of the compute waves cannot drain the DMA queue (hipcc waits vmcnt(0) at the first use of any
// load result while LDS-DMA is in flight) because the DMA is issued by other waves.
// The padded row layout is kept: the DMA destination is linear (base + lane x 16 B), so the SOURCE address of each 16-byte piece is
// computed from its position in the padded image (piece c of the image = row c / 17, piece c % 17 of that row for kp = 32; the 17th
// piece is the pad and re-reads the 16th).
template <int G, int R, int TILE, int LW>
constexpr int tile_buf_rows() { return LW > 0 ? TILE / 2 : TILE; }
template <int G, int R, int TILE, int LW>
constexpr int tile_buf_bytes() { // one staged buffer; a multiple of 1 KiB in loader mode (a DMA instruction writes a whole KiB)
  return LW > 0 ? (tile_buf_rows<G, R, TILE, LW>() * tile_row_bytes<G, R>() + 1023) / 1024 * 1024 : TILE * tile_row_bytes<G, R>();
}
template <int G, int R, int TILE, int LW>
constexpr int tile_lds_bytes() { return (LW > 0 ? 2 : 1) * tile_buf_bytes<G, R, TILE, LW>(); } // the descriptor table sits behind

template <int G, int R, int LW>
__device__ __forceinline__ void dma_tile(const double* __restrict__ other, int64_t lo, int64_t hi, char* buf, int wave, int lane) {
  constexpr int KPB = G * R * 8, ROWB = tile_row_bytes<G, R>(), CPR = ROWB / 16;
  const int rows = (int)(hi - lo), total = rows * CPR;
  const char* src0 = reinterpret_cast<const char*>(other) + lo * KPB;
  for (int base = wave * 64; base < total; base += LW * 64) { // one KiB of the padded image per instruction
    int c = base + lane;
    c = c < total ? c : total - 1;
    const int row = c / CPR;
    int cc = c - row * CPR;
    cc = cc < CPR - 1 ? cc : CPR - 2; // the pad piece
    const char* src = src0 + row * KPB + cc * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
  }
}

// Single-buffer staging by LDS-DMA from ALL waves (round 3; GLRM_TILE_DMA_ALL=0 builds the load / ds_write staging of stage_tile for A/B):
// every 16-byte piece of the tile image is one lane of a global_load_lds_dwordx4 -- all pieces of the tile in flight at once (nine
// instructions per wave for the 149 KB tile of k = 32), no VGPRs, no ds_write pass -- where stage_tile keeps four loads per thread in
// flight and needs three rounds of load -> wait -> write.  Lanes past the end of the image are masked off (an inactive lane writes
// nothing), so the buffer needs no KiB rounding.  Padded rows: piece c = row c / 17, piece c % 17 of it (the pad piece re-reads the
// 16th); ROT rows are unpadded and the image is a plain copy.  Measured on one box (profiles/r03_tile_dma_all_ab.txt): C5-family row sweep
// 63.2 -> 54.0 ms, C2 column passes 9.92 -> 9.00 ms, identical objectives.
#ifndef GLRM_TILE_DMA_ALL
#define GLRM_TILE_DMA_ALL 1
#endif
#ifndef GLRM_TILE_NT
#define GLRM_TILE_NT 0
#endif
#ifndef GLRM_TILE_CHUNK_MAJOR
#define GLRM_TILE_CHUNK_MAJOR 1
#endif
#ifndef GLRM_TILE_QUAD_FOUR
#define GLRM_TILE_QUAD_FOUR 0
#endif
template <int G, int R, int NW, bool ROT>
__device__ __forceinline__ void dma_tile_all(const double* __restrict__ other, int64_t lo, int64_t hi, char* buf, int wave, int lane) {
  constexpr int KPB = G * R * 8, ROWB = tile_row_stride<G, R, ROT>(), CPR = ROWB / 16;
  const int rows = (int)(hi - lo), total = rows * CPR;
  const char* src0 = reinterpret_cast<const char*>(other) + lo * KPB;
  for (int base = wave * 64; base < total; base += NW * 64) {
    const int c = base + lane;
    if (c < total) {
      const char* src;
      if constexpr (ROT) {
        src = src0 + c * 16;
      } else {
        const int row = c / CPR;
        int cc = c - row * CPR;
        cc = cc < CPR - 1 ? cc : CPR - 2; // the pad piece
        src = src0 + row * KPB + cc * 16;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + base * 16), 16, 0, 0);
    }
  }
}

// chunk held by register i of a lane whose first chunk sits at byte `ro` of the tile (ro carries the lane's 16-byte slot and, with
// ROT, its rotation in the chunk field, which register index i then flips: chunk i ^ rot)
template <bool ROT, int CB>
__device__ __forceinline__ double2 tile_chunk(const char* tile, int ro, int i) {
  return *reinterpret_cast<const double2*>(tile + (ROT ? (ro ^ (i * CB)) : ro + i * CB));
}

// One pass of every group of the workgroup over tiles [tile_begin, tile_end).
//   xv      the point at which the segment's losses are evaluated
//   g, J    outputs: gradient (GRAD) and loss sum of the segment's observations inside those tiles
//   pos     in/out: position in the segment's list (first entry with idx >= tile_begin*TILE on entry)
//   active  group-uniform; inactive groups only take part in staging and barriers
//
// L2 = true is the same walk WITHOUT the LDS tile: the group reads the opposing vectors of its observations straight from memory
// (one 16-byte load per lane and 2G components) and [tile_begin, tile_end) is one range that is walked in a single go, with no
// barrier.  It is what the phase-aligned pass kernels below run: when every group in flight walks its sorted list through the same
// super-tile at the same time, those reads hit the 4 MB L2 of the XCD (~30 TB/s) instead of the Infinity Cache / HBM (6.5-8 TB/s,
// profiles/r02_ubench_gather.txt).
template <int G, int R, int NW, int TILE, int LOSS, bool GRAD, bool L2 = false, int LW = 0, bool ROTK = false, bool ACC = false>
__device__ __forceinline__ void tiled_pass(const TiledArgs& a, char* lds, const Vec<G, R>& xv, Vec<G, R>& g, double& J,
                                           bool active, int64_t& pos, int64_t end, int tile_begin, int tile_end,
                                           const LossDesc& segloss, int lane, int j, int rot = 0) {
  constexpr bool ROT = ROTK && !L2 && LW == 0 && tile_rot<G, R>(); // register i holds chunk i ^ rot (see tile_rot)
  constexpr int ROWB = L2 ? G * R * 8 : tile_row_stride<G, R, ROT>(), NT = NW * 64;
  constexpr int CB = 2 * G * 8;                 // bytes per chunk row of the group
  const int jrot = j * 16 + (ROT ? rot * CB : 0);
  constexpr int TROWS = tile_buf_rows<G, R, TILE, LW>(), BUFB = tile_buf_bytes<G, R, TILE, LW>(); // staged rows / bytes per buffer
  if constexpr (LW > 0) { // tiles are counted in half tiles from here on
    tile_begin *= 2;
    tile_end *= 2;
  }
  // the whole batch of G observations per step (below).  GLRM_TILE_QUAD_FOUR: the uniform QuadLoss kernels too (G = 4 only).  The first
  // attempt (round 2, trial pass only, observation-major loops) lost to the two-observation step: one long dependent chain per four
  // observations.  With the chunk-major loops of round 4 the four chains are independent and interleaved.
  constexpr bool FOUR = (G == 4 || G == 8) && (LOSS != 0 || (GLRM_TILE_QUAD_FOUR && G == 4));
  const double two_scale = 2 * segloss.scale;
  J = 0.0;
  if (GRAD && !ACC) { // ACC: the caller's gradient is carried on (lockstep windows, glrm_blocked.hip)
#pragma unroll
    for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  }
  const int32_t* __restrict__ idx = a.idx;
  const double* __restrict__ vals = a.vals;
  // current batch: lane j of the group holds entry pos + j.  Batch loads are UNCONDITIONAL (address clamped
  // into the segment, validity applied when the entry is used) so that the compiler can keep the prefetch of
  // the next batch in flight with a counted s_waitcnt instead of draining the queue at every branch merge.
  const int64_t last = end > 0 ? end - 1 : 0; // idx/vals always hold at least one element
  constexpr bool UDESC = FOUR && loss_mode(LOSS) == 2 && !L2; // descriptor ids travel with the entries (see TiledArgs::descid)
  const uint8_t* __restrict__ descid = a.descid;
  const bool have_ids = UDESC && descid != nullptr;    // uniform
  const char* udesc_lds = lds + tile_lds_bytes<G, R, TILE, LW>(); // the kernel staged the distinct descriptors there
  // GLRM_TILE_NT: the (index, value, descriptor id) stream is read ONCE per pass -- 13 bytes per observation, 1.1 MB per tile step and XCD at
  // the C5 shape -- and with the default policy it pushes the staged tiles, which the other workgroups of the XCD are about to read, out
  // of the 4 MB L2; non-temporal loads keep it from being retained.
  auto load_entry = [&](int64_t p, int& c, double& av, int& did) {
    const int64_t q = p < last ? p : last;
#if GLRM_TILE_NT
    c = __builtin_nontemporal_load(idx + q);
    av = __builtin_nontemporal_load(vals + q);
#else
    c = idx[q];
    av = vals[q];
#endif
    did = 0;
    if constexpr (UDESC) {
#if GLRM_TILE_NT
      if (have_ids) did = __builtin_nontemporal_load(descid + q);
#else
      if (have_ids) did = descid[q];
#endif
    }
  };
  if constexpr (LW > 0) {
    if ((int)(threadIdx.x >> 6) < LW) { // loader wave: half tile t+1 lands in the other buffer while the compute waves consume t
      const int lwave = threadIdx.x >> 6;
      auto dma = [&](int t) {
        const int64_t lo = (int64_t)t * TROWS;
        const int64_t hi = lo + TROWS < a.n_other ? lo + TROWS : a.n_other;
        if (lo < hi) dma_tile<G, R, LW>(a.other, lo, hi, lds + ((t - tile_begin) & 1) * BUFB, lwave, lane);
      };
      if (tile_begin < tile_end) dma(tile_begin);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads(); // B0: the first half tile is in LDS
      for (int t = tile_begin; t < tile_end; ++t) {
        if (t + 1 < tile_end) dma(t + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // B(t+1): half tile t+1 has landed and every compute wave is done with half tile t
      }
      return;
    }
  }
  int cb, db;
  double ab;
  load_entry(pos + j, cb, ab, db);
  if (!(active && pos + j < end)) cb = 0x7fffffff;
  if constexpr (LW > 0) __syncthreads(); // B0
  for (int t = tile_begin; t < (L2 ? tile_begin + 1 : tile_end); ++t) {
    const int64_t lo = (int64_t)t * TROWS;
    const int64_t hi_ = L2 ? (int64_t)tile_end * TROWS : lo + TROWS;
    const int64_t hi = hi_ < a.n_other ? hi_ : a.n_other;
    const char* const mem = L2 ? reinterpret_cast<const char*>(a.other) : (LW > 0 ? lds + ((t - tile_begin) & 1) * BUFB : lds);
    if constexpr (!L2 && LW == 0) {
      __syncthreads(); // everybody is done with the previous tile
#if defined(GLRM_EXP_NOSTAGE) // timing experiment: stage only the first tile of the pass (results are wrong, the control flow stays finite)
      if (t == tile_begin) stage_tile<G, R, NT, ROT>(a.other, lo, hi, lds);
#elif GLRM_TILE_DMA_ALL
      dma_tile_all<G, R, NW, ROT>(a.other, lo, hi, lds, (int)(threadIdx.x >> 6), lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      stage_tile<G, R, NT, ROT>(a.other, lo, hi, lds);
#endif
      __syncthreads();
    }
#if defined(GLRM_EXP_NOCOMPUTE) // timing experiment: staging and barriers only
    bool done = true;
    if (t + 1 == tile_end && active) J = 1.0;
#else
    bool done = !active;
#endif
    while (!done) {
      int cn, dn; // prefetch the next batch while this one is consumed
      double an;
      load_entry(pos + G + j, cn, an, dn);
      const bool next_ok = pos + G + j < end;
      int nproc = 0;
      // Two observations of the batch per step.  Every lane forms its partial dot products for BOTH observations;
      // the first butterfly step is a reduce-scatter (even lanes keep observation u0, odd lanes u0+1), the remaining
      // steps reduce one value; loss and derivative are then evaluated once per lane for "its" observation and the
      // derivative of the other one comes back with one DPP move.  ~30 % (gradient pass) / ~45 % (trial pass) fewer
      // VALU instructions than one observation per step.
      const bool odd = (j & 1) != 0;
      if constexpr (FOUR) {
        // Losses with transcendental or branchy formulas (everything but the uniform QuadLoss model), G = 4 or 8: the whole
        // batch of G observations per step, one loss evaluation per LANE.  The partial dot products of all G observations are
        // reduce-scattered in log2(G) butterfly steps (same pairings as below, so the same bits), lane u evaluates observation
        // u, and the G derivatives come back by broadcasts.  The opposing vectors are re-read from LDS for the gradient update
        // instead of being kept live across the loss evaluation (which is what made this kernel spill).
        int c[G];
        bool ok[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          c[u] = group_bcast_i32<G>(cb, u, lane);
          ok[u] = (u == 0 || ok[u - 1]) && c[u] < (int)hi; // c == INT_MAX past the end of the segment
        }
        if (ok[0]) {
          const char* rp[G]; // L2: address of the lane's first chunk
          int ro[G];         // LDS tile: its byte offset in the tile
          double p[G];
          bool mine = false;
#pragma unroll
          for (int u = 0; u < G; ++u) {
            if constexpr (L2) rp[u] = mem + j * 16 + (int64_t)(ok[u] ? c[u] : c[0]) * ROWB;
            else ro[u] = ((ok[u] ? c[u] : c[0]) - (int)lo) * ROWB + jrot;
            mine = (j == u) ? ok[u] : mine;
            p[u] = 0.0;
          }
#if GLRM_TILE_CHUNK_MAJOR
          // chunk-major: the G reads of a chunk are issued together and feed G INDEPENDENT fma chains (round 4: the observation-major
          // order compiled to sixteen serialized LDS round trips per step -- one read in flight, `s_waitcnt lgkmcnt(0)` after each --
          // because the scheduler minimises registers in a kernel at its VGPR cap).  Every p[u] still adds its chunks in ascending order.
#pragma unroll
          for (int i = 0; i < R / 2; ++i) {
            double2 yv[G];
#pragma unroll
            for (int u = 0; u < G; ++u) yv[u] = L2 ? *reinterpret_cast<const double2*>(rp[u] + i * CB) : tile_chunk<ROT, CB>(mem, ro[u], i);
#pragma unroll
            for (int u = 0; u < G; ++u) {
              p[u] = fma(xv.v[i].x, yv[u].x, p[u]);
              p[u] = fma(xv.v[i].y, yv[u].y, p[u]);
            }
          }
#else
#pragma unroll
          for (int u = 0; u < G; ++u) {
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
              const double2 y = L2 ? *reinterpret_cast<const double2*>(rp[u] + i * CB) : tile_chunk<ROT, CB>(mem, ro[u], i);
              p[u] = fma(xv.v[i].x, y.x, p[u]);
              p[u] = fma(xv.v[i].y, y.y, p[u]);
            }
          }
#endif
          const bool hi2 = (j & 2) != 0;
          double q[G / 2]; // q[i]: observation 2i + odd, summed over lanes {j, j^1}
#pragma unroll
          for (int i = 0; i < G / 2; ++i) q[i] = (odd ? p[2 * i + 1] : p[2 * i]) + dpp_f64<DPP_XOR1>(odd ? p[2 * i] : p[2 * i + 1]);
          double dot;
          if constexpr (G == 4) {
            dot = (hi2 ? q[1] : q[0]) + dpp_f64<DPP_XOR2>(hi2 ? q[0] : q[1]); // observation j
          } else {
            const bool hi4 = (j & 4) != 0;
            const double r0 = (hi2 ? q[1] : q[0]) + dpp_f64<DPP_XOR2>(hi2 ? q[0] : q[1]); // observation (j & 3)
            const double r1 = (hi2 ? q[3] : q[2]) + dpp_f64<DPP_XOR2>(hi2 ? q[2] : q[3]); // observation 4 + (j & 3)
            dot = (hi4 ? r1 : r0) + swizzle_xor_f64<4>(hi4 ? r0 : r1);                     // observation j
          }
          double L, dL;
          if constexpr (LOSS == 0) { // one QuadLoss descriptor (src/losses.jl:144,146)
            const double dq = dot - ab;
            L = segloss.scale * (dq * dq);
            dL = dq * two_scale;
          } else if constexpr (loss_mode(LOSS) == 1) {
            loss_both<GRAD, loss_trig(LOSS)>(segloss, dot, ab, L, dL);
          } else {
            LossDesc lo_;
            if (have_ids) { // 32-byte descriptor from the LDS table (glrm_loss layout: kind, dim, scale, p0, p1)
              const char* dp = udesc_lds + db * 32;
              const int2 kd = *reinterpret_cast<const int2*>(dp);
              const double sc = *reinterpret_cast<const double*>(dp + 8);
              const double2 pp = *reinterpret_cast<const double2*>(dp + 16);
              lo_ = LossDesc{kd.x, sc, pp.x, pp.y};
            } else {
              lo_ = load_loss(a.losses, mine ? cb : c[0]);
            }
            loss_both<GRAD, loss_trig(LOSS)>(lo_, dot, ab, L, dL);
          }
          if (!mine) {
            L = 0.0;
            dL = 0.0;
          }
          J += L; // every observation once
          if (GRAD) {
            asm volatile("" ::: "memory"); // the reads below are real re-reads, not the values of the first ones kept in VGPRs
#if GLRM_TILE_CHUNK_MAJOR
            double dv[G]; // observations past the tile window carry a zero derivative
#pragma unroll
            for (int u = 0; u < G; ++u) dv[u] = group_bcast_f64<G>(dL, u, lane);
#pragma unroll
            for (int i = 0; i < R / 2; ++i) { // chunk-major like the dot products; every component still adds its terms in list order
              double2 yv[G];
#pragma unroll
              for (int u = 0; u < G; ++u) yv[u] = L2 ? *reinterpret_cast<const double2*>(rp[u] + i * CB) : tile_chunk<ROT, CB>(mem, ro[u], i);
#pragma unroll
              for (int u = 0; u < G; ++u) {
                g.v[i].x = fma(dv[u], yv[u].x, g.v[i].x);
                g.v[i].y = fma(dv[u], yv[u].y, g.v[i].y);
              }
            }
#else
#pragma unroll
            for (int u = 0; u < G; ++u) { // list order; observations past the tile window carry a zero derivative
              const double d = group_bcast_f64<G>(dL, u, lane);
#pragma unroll
              for (int i = 0; i < R / 2; ++i) {
                const double2 y = L2 ? *reinterpret_cast<const double2*>(rp[u] + i * CB) : tile_chunk<ROT, CB>(mem, ro[u], i);
                g.v[i].x = fma(d, y.x, g.v[i].x);
                g.v[i].y = fma(d, y.y, g.v[i].y);
              }
            }
#endif
          }
#pragma unroll
          for (int u = 0; u < G; ++u) nproc += ok[u] ? 1 : 0;
          if (!ok[G - 1]) done = true;
        } else {
          done = true;
        }
      } else {
#pragma unroll
      for (int u0 = 0; u0 < G; u0 += 2) {
        const int c0 = group_bcast_i32<G>(cb, u0, lane);
        const int c1 = group_bcast_i32<G>(cb, u0 + 1, lane);
        const bool ok0 = !done && c0 < (int)hi; // c == INT_MAX past the end of the segment
        const bool ok1 = ok0 && c1 < (int)hi;
        if (ok0) {
          double2 y0[R / 2], y1[R / 2];
          if constexpr (L2) {
            const char* rp0 = mem + (int64_t)c0 * ROWB + j * 16;
            const char* rp1 = mem + (int64_t)(ok1 ? c1 : c0) * ROWB + j * 16;
#pragma unroll
            for (int i = 0; i < R / 2; ++i) y0[i] = *reinterpret_cast<const double2*>(rp0 + i * CB);
#pragma unroll
            for (int i = 0; i < R / 2; ++i) y1[i] = *reinterpret_cast<const double2*>(rp1 + i * CB);
          } else {
            const int ro0 = (c0 - (int)lo) * ROWB + jrot, ro1 = ((ok1 ? c1 : c0) - (int)lo) * ROWB + jrot;
#pragma unroll
            for (int i = 0; i < R / 2; ++i) y0[i] = tile_chunk<ROT, CB>(mem, ro0, i);
#pragma unroll
            for (int i = 0; i < R / 2; ++i) y1[i] = tile_chunk<ROT, CB>(mem, ro1, i);
          }
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int i = 0; i < R / 2; ++i) {
            p0 = fma(xv.v[i].x, y0[i].x, p0);
            p0 = fma(xv.v[i].y, y0[i].y, p0);
          }
#pragma unroll
          for (int i = 0; i < R / 2; ++i) {
            p1 = fma(xv.v[i].x, y1[i].x, p1);
            p1 = fma(xv.v[i].y, y1[i].y, p1);
          }
          const double keep = odd ? p1 : p0, send = odd ? p0 : p1;
          double dot = keep + dpp_f64<DPP_XOR1>(send); // lane pair sum for observation u0 + odd
          // remaining butterfly steps must keep the lane parity (mirrors would mix the two observations)
          if constexpr (G >= 4) dot += dpp_f64<DPP_XOR2>(dot);
          if constexpr (G >= 8) dot += swizzle_xor_f64<4>(dot);
          if constexpr (G >= 16) dot += swizzle_xor_f64<8>(dot);
          double av = pair_bcast_f64<G>(ab, u0, lane);
          double L, dL;
          if constexpr (LOSS == 0) {
            const double d = dot - av;
            L = segloss.scale * (d * d);
            dL = d * two_scale; // == (2*d)*scale bit for bit: doubling is exact
          } else if constexpr (loss_mode(LOSS) == 1) {
            loss_both<GRAD, loss_trig(LOSS)>(segloss, dot, av, L, dL);
          } else {
            const LossDesc lo_ = load_loss(a.losses, odd ? (ok1 ? c1 : c0) : c0);
            loss_both<GRAD, loss_trig(LOSS)>(lo_, dot, av, L, dL);
          }
          if (odd && !ok1) {
            L = 0.0;
            dL = 0.0;
          }
          J += L; // every observation is accumulated by G/2 lanes; the caller rescales the group sum by 2/G (exact)
          if (GRAD) {
            // list order: u0 first, then u0+1.  Even lanes hold the derivative of u0, odd lanes that of u0+1: two quad
            // broadcasts ([0,0,2,2] / [1,1,3,3]) instead of an exchange plus two selects
            const double d0 = group_bcast_f64<2>(dL, 0, lane), d1 = group_bcast_f64<2>(dL, 1, lane);
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
              g.v[i].x = fma(d0, y0[i].x, g.v[i].x);
              g.v[i].y = fma(d0, y0[i].y, g.v[i].y);
            }
#pragma unroll
            for (int i = 0; i < R / 2; ++i) {
              g.v[i].x = fma(d1, y1[i].x, g.v[i].x);
              g.v[i].y = fma(d1, y1[i].y, g.v[i].y);
            }
          }
          nproc += ok1 ? 2 : 1;
          if (!ok1) done = true;
        } else {
          done = true;
        }
      }
      }
      pos += nproc;
      if (!done) { // the whole batch was consumed: continue with the prefetched one
        cb = next_ok ? cn : 0x7fffffff;
        ab = an;
        db = dn;
      } else if (nproc > 0) { // stopped inside the batch: re-anchor the batch at the new position
        load_entry(pos + j, cb, ab, db);
        if (!(pos + j < end)) cb = 0x7fffffff;
      }
    }
    if constexpr (LW > 0) __syncthreads(); // B(t+1), see the loader loop
  }
  J = group_sum<G>(J) * (FOUR ? 1.0 : 2.0 / G); // two per step: lanes hold parity-partial sums, each observation counted G/2 times
}

// Experiment (GLRM_HIP_TILE_STAGGER): the workgroups an XCD holds at a time all stage the SAME tile at the same moment -- they start
// together and do equal work -- and 32 simultaneous requests for a line that is not in L2 yet are 32 misses.  Started a fraction of a
// tile step apart, the first one misses and the others find the line in L2.  Workgroup b runs on XCD b % 8; its place among the 32 the
// XCD holds is (b / 8) % 32.  Changes no result.
__device__ __forceinline__ void tile_stagger(int stagger, unsigned linear_block) {
  if (stagger > 0) {
    const int n = (int)((linear_block >> 3) & 31u) * stagger;
    for (int t = 0; t < n; ++t) __builtin_amdgcn_s_sleep(8); // ~512 clocks each
  }
}

// distinct loss descriptors -> LDS behind the tile (read after the first tile barrier of tiled_pass); see TiledArgs::descid
template <int G, int R, int NW, int TILE, int LOSS, int LW = 0>
__device__ __forceinline__ void stage_udesc(const TiledArgs& a, char* lds) {
  if constexpr (loss_mode(LOSS) == 2) {
    if (a.descid) {
      const int words = a.n_udesc * 8; // 32 bytes each
      const int* src = reinterpret_cast<const int*>(a.udesc);
      int* dst = reinterpret_cast<int*>(lds + tile_lds_bytes<G, R, TILE, LW>());
      for (int w = threadIdx.x; w < words; w += NW * 64) dst[w] = src[w];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// X half-step (and any sweep whose segments are plentiful): whole sweep in one kernel.
// Workgroup = NW waves, SPB = NW*64/G segments; each pass streams the WHOLE opposing factor through LDS.
// FIXED = true is the SparseProxGradParams step (one gradient pass, x <- prox(x - (alpha/l) g), no line search); it is a
// separate instantiation so that the line-search kernel keeps its register budget.
// ROUNDS = true (round 4, the default): the kernel makes the gradient pass and the FIRST trial; a segment whose first trial is rejected
// leaves its state -- gradient, J_old, shrunk step size, next trial point -- in the pass buffers (gsum, jold, alpha, trial, ntrial,
// active) and its id in actlist_out, and the host runs the remaining trials as rounds of (tiled_col_pass_kernel<GRAD = false, ROWS> over
// the listed segments, col_decide_kernel): only workgroups made of still-searching rows stage tiles again.  Same sums, same bits.
template <int G, int R, int NW, int TILE, int LOSS, bool FIXED, int LW = 0, bool ROUNDS = false>
__global__ void __launch_bounds__(NW * 64, NW == 12 ? 3 : 4) tiled_sweep_kernel(const TiledArgs a) {
  constexpr int KP = G * R, NGW = 64 / G, SPB = (NW - LW) * NGW; // the first LW waves are loaders (double-buffered tiles)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % G, gi = lane / G;
  const int64_t slot = (int64_t)blockIdx.x * SPB + (wave - LW) * NGW + gi;
  const bool have = wave >= LW && slot < a.nseg;
  const int64_t seg = (have && a.segperm) ? (int64_t)a.segperm[slot] : slot; // segments of similar length share a wave (skewed data)
  const int64_t beg = have ? a.ptr[seg] : 0, end = have ? a.ptr[seg + 1] : 0;
  const int64_t gseg = a.own_offset + (have ? seg : 0);
  double2* ownp = reinterpret_cast<double2*>(a.own + gseg * KP);
  const int ntiles = (int)((a.n_other + TILE - 1) / TILE);
  tile_stagger(a.stagger, blockIdx.x);
  stage_udesc<G, R, NW, TILE, LOSS, LW>(a, lds);

  Vec<G, R> g, xn;
  const RegDesc rd = load_reg(a.regs, (a.reg_single || !have) ? 0 : seg);
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != 2) segloss = load_loss(a.losses, (a.loss_by_segment && have) ? gseg : 0);
  const double l = (double)(end - beg) + 1.0;

  double Jold;
  int64_t pos = beg;
  {
    Vec<G, R> x;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) x.v[i] = have ? ownp[i * G + j] : make_double2(0.0, 0.0);
    tiled_pass<G, R, NW, TILE, LOSS, true, false, LW>(a, lds, x, g, Jold, have, pos, end, 0, ntiles, segloss, lane, j);
    if constexpr (FIXED) { // src/algorithms/sparse_proxgrad.jl:72-78: g *= -alpha/l; x += g; prox!(r, x, alpha/l)
      const double s = a.fixed_alpha / l;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        xn.v[i].x = x.v[i].x + g.v[i].x * (-s);
        xn.v[i].y = x.v[i].y + g.v[i].y * (-s);
      }
      reg_prox<G, R>(rd, xn, s, j, a.k);
      if (have) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
      }
      return;
    }
    Jold += reg_eval<G, R>(rd, x, j, a.k);
  }

  // Backtracking line search.  The current point is NOT kept in registers across the trial passes: it is re-read
  // from the factor array when the next trial point is formed (32 B per lane per trial) and the accepted point is
  // written back at once, which frees R VGPRs per lane inside the pass.
  double alpha = have ? a.alpha[seg] : 0.0;
  int ntrials = 0;
  bool accepted = false;
  bool searching = have && alpha > a.min_stepsize;
  bool first = true;
  while ((!ROUNDS || first) && __syncthreads_or(searching ? 1 : 0)) {
    first = false;
    const double s = alpha / l;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      const double2 xi = have ? ownp[i * G + j] : make_double2(0.0, 0.0);
      xn.v[i].x = fma(-s, g.v[i].x, xi.x);
      xn.v[i].y = fma(-s, g.v[i].y, xi.y);
    }
    reg_prox<G, R>(rd, xn, s, j, a.k);
    double Jn;
    Vec<G, R> dummy;
    pos = beg;
    tiled_pass<G, R, NW, TILE, LOSS, false, false, LW>(a, lds, xn, dummy, Jn, searching, pos, end, 0, ntiles, segloss, lane, j);
    Jn += reg_eval<G, R>(rd, xn, j, a.k);
    if (searching) {
      ++ntrials;
      if (Jn < Jold) {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) ownp[i * G + j] = xn.v[i];
        alpha *= 1.05;
        Jold = Jn;
        accepted = true;
        searching = false;
      } else {
        alpha *= .7;
        if (alpha < a.min_stepsize) {
          alpha = a.min_stepsize * 1.1;
          searching = false;
        } else if (!(alpha > a.min_stepsize)) {
          searching = false; // `while alpha > min_stepsize` (proxgrad.jl:136): alpha == min_stepsize ends the search (min_stepsize = 0: alpha underflowed to 0)
        }
      }
    }
  }
  if constexpr (ROUNDS) {
    if (have) {
      if (searching) { // first trial rejected, step size still above the minimum: hand the segment over to the rounds
        const double s = alpha / l;
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          const double2 xi = ownp[i * G + j];
          xn.v[i].x = fma(-s, g.v[i].x, xi.x);
          xn.v[i].y = fma(-s, g.v[i].y, xi.y);
        }
        reg_prox<G, R>(rd, xn, s, j, a.k);
        double2* gp = reinterpret_cast<double2*>(a.gsum + seg * (int64_t)KP);
        double2* tp = reinterpret_cast<double2*>(a.trial + seg * (int64_t)KP);
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
          gp[i * G + j] = g.v[i];
          tp[i * G + j] = xn.v[i];
        }
        if (j == 0) {
          a.alpha[seg] = alpha;
          a.jold[seg] = Jold;
          a.ntrial[seg] = ntrials;
          a.active[seg] = 1;
          const unsigned slot_out = atomicAdd(a.nactive, 1u);
          a.actlist_out[slot_out] = (int32_t)seg;
        }
        return; // col_decide_kernel books trials / accepts when the segment's search ends
      }
      if (j == 0) a.active[seg] = 0;
    }
  }
  if (have && j == 0) {
    a.alpha[seg] = alpha;
    if (a.obj) a.obj[gseg] = Jold;
    if (a.trials) {
      a.trials[seg] += ntrials;
      a.accepts[seg] += accepted ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Y half-step when columns are few and long: rows are split into NSUP super-tiles, one workgroup per
// (column group, super-tile); partial sums are reduced in super-tile order by col_reduce_kernel /
// col_decide_kernel, and the line search runs as rounds of (pass, decide) launches.

template <int G>
__device__ __forceinline__ int64_t lower_bound_idx(const int32_t* idx, int64_t b, int64_t e, int64_t key) {
  while (b < e) {
    const int64_t mid = b + ((e - b) >> 1);
    if ((int64_t)idx[mid] < key) b = mid + 1; else e = mid;
  }
  return b;
}

// GRAD = true: pass 1 (gradient + loss partials at the current point a.own);
// GRAD = false: trial pass (loss partials at a.trial for the still-active segments).
//
// L2 = true: the phase-aligned gather pass (no LDS tile, tiled_pass<..., L2>).  One launch covers ONE super-tile (a.sup_fixed) and
// a slice [a.seg_begin, a.seg_begin + a.nseg_slice) of the segments that is at most what the chip holds at once, so every group of
// the launch starts its walk through the super-tile at the same moment; the host issues the launches super-tile by super-tile.
template <int G, int R, int NW, int TILE, int LOSS, bool GRAD, bool L2 = false, int LW = 0, bool ROWS = false>
__global__ void __launch_bounds__(NW * 64, L2 ? 1 : 4) tiled_col_pass_kernel(const TiledArgs a) {
  constexpr int KP = G * R, NGW = 64 / G, SPB = (NW - LW) * NGW, PSTRIDE = KP + 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % G, gi = lane / G;
  const int64_t slot = (L2 ? a.seg_begin : 0) + (int64_t)blockIdx.x * SPB + (wave - LW) * NGW + gi;
  const int sup = L2 ? a.sup_fixed : a.sup0 + (int)blockIdx.y;
  bool have = wave >= LW && slot < (L2 ? a.seg_begin + a.nseg_slice : (a.npass > 0 ? a.npass : a.nseg));
  const int64_t seg = (have && a.segperm) ? (int64_t)a.segperm[slot] : slot; // which column a group works on does not change any sum
  if (!GRAD && have) have = a.active[seg] != 0;
  if (!GRAD && !__syncthreads_or(have ? 1 : 0)) return; // nothing left to evaluate in this column group
  if constexpr (!L2) {
    tile_stagger(a.stagger, blockIdx.y * gridDim.x + blockIdx.x);
    stage_udesc<G, R, NW, TILE, LOSS, LW>(a, lds);
  }
  const int64_t beg = have ? a.ptr[seg] : 0, end = have ? a.ptr[seg + 1] : 0;
  const int64_t gseg = a.own_offset + (have ? seg : 0);
  const int ntiles = (int)((a.n_other + TILE - 1) / TILE);
  const int tb = sup * a.tiles_per_sup;
  const int te = tb + a.tiles_per_sup < ntiles ? tb + a.tiles_per_sup : ntiles;
  const double2* xp = reinterpret_cast<const double2*>(GRAD ? a.own + gseg * KP : a.trial + (have ? seg : 0) * (int64_t)KP);
  // ROWS: the trial rounds of the row view (the row sweep's own passes read padded tiles without rotation: the same bits here)
  constexpr bool ROT = !ROWS && !L2 && LW == 0 && LOSS != 0 && tile_rot<G, R>();
  const int rot = ROT ? tile_rot_of(gseg) : 0; // register i <-> chunk i ^ rot (conflict-free tile reads, see tile_rot)
  Vec<G, R> x, g;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) x.v[i] = have ? xp[(i ^ rot) * G + j] : make_double2(0.0, 0.0);
  LossDesc segloss = LossDesc{0, 1.0, 0.0, 0.0};
  if constexpr (loss_mode(LOSS) != 2) segloss = load_loss(a.losses, (a.loss_by_segment && have) ? gseg : 0);
  int64_t pos = have ? lower_bound_idx<G>(a.idx, beg, end, (int64_t)tb * TILE) : 0;
  double J;
  tiled_pass<G, R, NW, TILE, LOSS, GRAD, L2, LW, ROT>(a, lds, x, g, J, have, pos, end, tb, te, segloss, lane, j, rot);
  if (have) {
    double* p = a.part + ((int64_t)seg * a.nsup + sup) * PSTRIDE;
    if (GRAD) {
#pragma unroll
      for (int i = 0; i < R / 2; ++i) *reinterpret_cast<double2*>(p + (i ^ rot) * 2 * G + 2 * j) = g.v[i];
    }
    if (j == 0) p[KP] = J;
  }
}

// After pass 1: reduce the partials in super-tile order, J_old = loss + r(y), first trial point.
template <int G, int R>
__global__ void __launch_bounds__(256) col_reduce_kernel(const TiledArgs a) {
  constexpr int KP = G * R, NGW = 64 / G, PSTRIDE = KP + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % G, gi = lane / G;
  const int64_t seg = ((int64_t)blockIdx.x * 4 + wave) * NGW + gi;
  if (seg >= a.nseg) return; // group-uniform
  if (a.long_from > 0 && a.ptr && a.ptr[seg + 1] - a.ptr[seg] >= a.long_from) return; // swept by the gather sweep beside the passes
  const int64_t gseg = a.own_offset + seg;
  Vec<G, R> g, y, yn;
  double J = 0.0;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) g.v[i] = make_double2(0.0, 0.0);
  for (int s = 0; s < a.nsup; ++s) {
    const double* p = a.part + ((int64_t)seg * a.nsup + s) * PSTRIDE;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      const double2 v = *reinterpret_cast<const double2*>(p + i * 2 * G + 2 * j);
      g.v[i].x += v.x;
      g.v[i].y += v.y;
    }
    J += p[KP];
  }
  if (a.eval_only) {
    if (j == 0 && a.obj) a.obj[gseg] = J;
    return;
  }
  const double2* yp = reinterpret_cast<const double2*>(a.own + gseg * KP);
#pragma unroll
  for (int i = 0; i < R / 2; ++i) y.v[i] = yp[i * G + j];
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : seg);
  if (a.fixed_alpha > 0.0) { // SparseProxGradParams step on this segment, no trial rounds
    const double l0 = (double)(a.ptr ? a.ptr[seg + 1] - a.ptr[seg] : a.dense_len) + 1.0;
    const double s0 = a.fixed_alpha / l0;
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      yn.v[i].x = y.v[i].x + g.v[i].x * (-s0);
      yn.v[i].y = y.v[i].y + g.v[i].y * (-s0);
    }
    reg_prox<G, R>(rd, yn, s0, j, a.k);
    double2* yw = reinterpret_cast<double2*>(a.own + gseg * KP);
#pragma unroll
    for (int i = 0; i < R / 2; ++i) yw[i * G + j] = yn.v[i];
    if (j == 0) a.active[seg] = 0;
    return;
  }
  const double Jold = J + reg_eval<G, R>(rd, y, j, a.k);
  if (a.jloss && j == 0) a.jloss[seg] = J;
  const double alpha = a.alpha[seg];
  const double l = (double)(a.ptr ? a.ptr[seg + 1] - a.ptr[seg] : a.dense_len) + 1.0;
  const bool searching = alpha > a.min_stepsize;
  const double s = alpha / l;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) {
    yn.v[i].x = fma(-s, g.v[i].x, y.v[i].x);
    yn.v[i].y = fma(-s, g.v[i].y, y.v[i].y);
  }
  reg_prox<G, R>(rd, yn, s, j, a.k);
  double2* gp = reinterpret_cast<double2*>(a.gsum + seg * (int64_t)KP);
  double2* tp = reinterpret_cast<double2*>(a.trial + seg * (int64_t)KP);
#pragma unroll
  for (int i = 0; i < R / 2; ++i) {
    gp[i * G + j] = g.v[i];
    tp[i * G + j] = yn.v[i];
  }
  if (j == 0) {
    a.jold[seg] = Jold;
    a.active[seg] = searching ? 1 : 0;
    a.ntrial[seg] = 0;
    if (a.obj) a.obj[gseg] = Jold;
    if (searching) {
      const unsigned slot = atomicAdd(a.nactive, 1u);
      if (a.actlist_out) a.actlist_out[slot] = (int32_t)seg;
    }
  }
}

// After a trial pass: J' = sum of partial losses + r(y'); accept / shrink / give up; next trial point.
template <int G, int R>
__global__ void __launch_bounds__(256) col_decide_kernel(const TiledArgs a) {
  constexpr int KP = G * R, NGW = 64 / G, PSTRIDE = KP + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane % G, gi = lane / G;
  int64_t seg = ((int64_t)blockIdx.x * 4 + wave) * NGW + gi;
  if (a.actlist_in) { // the still-searching segments only
    if (seg >= a.nact_in) return;
    seg = a.actlist_in[seg];
  }
  if (seg >= a.nseg) return;
  if (a.active[seg] == 0) return; // group-uniform
  const int64_t gseg = a.own_offset + seg;
  double Jn = 0.0;
  for (int s = 0; s < a.nsup; ++s) Jn += a.part[((int64_t)seg * a.nsup + s) * PSTRIDE + KP];
  Vec<G, R> y, yn, g;
  double2* yp = reinterpret_cast<double2*>(a.own + gseg * KP);
  double2* tp = reinterpret_cast<double2*>(a.trial + seg * (int64_t)KP);
  const double2* gp = reinterpret_cast<const double2*>(a.gsum + seg * (int64_t)KP);
#pragma unroll
  for (int i = 0; i < R / 2; ++i) yn.v[i] = tp[i * G + j];
  const RegDesc rd = load_reg(a.regs, a.reg_single ? 0 : seg);
  Jn += reg_eval<G, R>(rd, yn, j, a.k);
  const double Jold = a.jold[seg];
  double alpha = a.alpha[seg];
  int still = 0, acc = 0;
  if (Jn < Jold) {
#pragma unroll
    for (int i = 0; i < R / 2; ++i) yp[i * G + j] = yn.v[i];
    alpha *= 1.05;
    acc = 1;
    if (j == 0 && a.obj) a.obj[gseg] = Jn;
  } else {
    alpha *= .7;
    if (alpha < a.min_stepsize) {
      alpha = a.min_stepsize * 1.1;
    } else if (alpha > a.min_stepsize) { // `while alpha > min_stepsize` (proxgrad.jl:136,180): alpha == min_stepsize ends the search without the 1.1 reset
      still = 1;
      const double l = (double)(a.ptr ? a.ptr[seg + 1] - a.ptr[seg] : a.dense_len) + 1.0;
      const double s = alpha / l;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        y.v[i] = yp[i * G + j];
        g.v[i] = gp[i * G + j];
        yn.v[i].x = fma(-s, g.v[i].x, y.v[i].x);
        yn.v[i].y = fma(-s, g.v[i].y, y.v[i].y);
      }
      reg_prox<G, R>(rd, yn, s, j, a.k);
#pragma unroll
      for (int i = 0; i < R / 2; ++i) tp[i * G + j] = yn.v[i];
    }
  }
  if (j == 0) {
    a.alpha[seg] = alpha;
    a.active[seg] = still;
    const int nt = a.ntrial[seg] + 1;
    a.ntrial[seg] = nt;
    if (still) {
      const unsigned slot = atomicAdd(a.nactive, 1u);
      if (a.actlist_out) a.actlist_out[slot] = (int32_t)seg;
    } else if (a.trials) {
      a.trials[seg] += nt;
      a.accepts[seg] += acc;
    }
  }
}

// 1 if some segment's list is not ordered by TILE: the tiled sweeps need the entries of tile t to precede those of tile t+1
// (order inside a tile is free: the tile sits in LDS); a non-decreasing index list is the common special case.
static __global__ void check_sorted_kernel(const int64_t* ptr, const int32_t* idx, int64_t nseg, int tile, int* unsorted) {
  const int64_t seg = (int64_t)blockIdx.x;
  if (seg >= nseg) return;
  const int64_t b = ptr[seg], e = ptr[seg + 1];
  int bad = 0;
  for (int64_t t = b + 1 + threadIdx.x; t < e; t += blockDim.x) bad |= idx[t] / tile < idx[t - 1] / tile;
  if (bad) *unsorted = 1;
}

// Heterogeneous models (a loss descriptor per column): inside every tile window of a row, group the entries by loss kind
// (stable).  The lane groups of a wave walk their windows in lockstep, so most steps then meet ONE loss formula instead of all
// of them.  Only the engine's private copy of the row view is reordered; the sums change by rounding only.
static __global__ void __launch_bounds__(64) group_rows_by_kind_kernel(const int64_t* ptr, const int32_t* idx, const double* vals, int64_t nseg, int tile,
                                                                        const glrm_loss* losses, int32_t* oidx, double* ovals) {
  const int64_t seg = (int64_t)blockIdx.x;
  if (seg >= nseg) return;
  const int64_t b = ptr[seg], e = ptr[seg + 1];
  for (int64_t t = b + threadIdx.x; t < e; t += 64) {
    const int c = idx[t], tl = c / tile, kind = losses[c].kind;
    int64_t wb = t, we = t + 1; // window of the entry's tile (windows hold a few dozen entries)
    while (wb > b && idx[wb - 1] / tile == tl) --wb;
    while (we < e && idx[we] / tile == tl) ++we;
    int64_t rank = 0;
    for (int64_t u = wb; u < we; ++u) {
      const int ku = losses[idx[u]].kind;
      rank += (ku < kind) || (ku == kind && u < t);
    }
    oidx[wb + rank] = c;
    ovals[wb + rank] = vals[t];
  }
}

// descid[t] = colid[idx[t]]: the id of the column's loss descriptor among the model's distinct descriptors
static __global__ void entry_descid_kernel(const int32_t* idx, int64_t nnz, const uint8_t* colid, uint8_t* descid) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) descid[t] = colid[idx[t]];
}

} // namespace glrm
