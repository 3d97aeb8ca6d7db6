// glrm_dense.hpp -- fully observed QuadLoss case on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// When every entry of A is observed and every column carries QuadLoss(scale), the half-step of a block of
// segments is two chained GEMMs (src/algorithms/proxgrad.jl:122-135 / :165-178 in matrix form):
//     R = X_b' Y - A_b            (b x n)      residuals
//     J_e = s * sum_c R[e,c]^2                 row objective (evaluate_fit.jl:24-38 restricted to Quad)
//     G_b = 2 s * R Y'            (b x k)      gradient
// XY is never materialised (the reference allocates it, proxgrad.jl:65-66): a wave owns 16 segments, walks the
// opposing factor in 16-vector tiles staged through LDS, forms the 16x16 residual tile with k/4 MFMAs, subtracts
// A (streamed once per pass, 32 contiguous bytes per lane), accumulates J, and feeds the residual tile straight
// back as the A-operand of the second GEMM (k/16 x 4 MFMAs) -- the D layout of the first product IS the A-operand
// layout of the second when the first computes R' (tile index <-> opposing vector, j <-> segment).
//
// MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane&15][kk = lane>>4], B[kk = lane>>4][j = lane&15],
// D[i = (lane>>4) + 4*reg][j = lane&15].
//   GEMM1: i = tile index of the opposing vector, j = segment e:  D1[i][e] = sum_kk Y[kk][col(i)] * X[kk][e]
//   GEMM2: i = segment e, kk = tile index inside slab r (i1 = kk + 4r), j = component:
//          D2[e][comp] += sum_kk R[e][col(kk + 4r)] * Y[comp][col(kk + 4r)]
// col(cq + 4r) = 4*cq + r, so the four residuals a lane holds are four CONTIGUOUS entries of its A row.
//
// Ceiling of the instruction (tools/ubench_mfma.hip, profiles/r02_ubench_mfma.txt): v_mfma_f64_16x16x4_f64 sustains 50 TFLOP/s on
// MI355X with register operands and nothing else going on (it issues every ~96 cycles at 2.38 GHz, not the 64 that the 78.6 TFLOP/s
// rating implies); the gradient pass below runs at 50-52.  v_mfma_f64_4x4x4_4b_f64 sustains 70-73 TFLOP/s but needs four times the
// operand words per flop: the same pass rebuilt on it (lane maps from tools/probe_mfma4.hip, operands in 16-byte LDS reads, conflict-
// free row padding) was parity-green and no faster (82.3 vs 78.0 ms per C3 iteration) -- measured and dropped, LABNOTES.md section 4.3.
//
// The line search (trial passes, accept / shrink) reuses col_reduce_kernel / col_decide_kernel of glrm_tiled.hpp.
#pragma once

#include "glrm_device.hpp"
#include "glrm_tiled.hpp"

namespace glrm {

using f64x4 = __attribute__((ext_vector_type(4))) double;

struct DenseArgs {
  int64_t nseg;         // local segments (rows of the packed A block)
  const double* xsrc;   // GRAD: the factor being updated (global array, ld KP, segment s at (own_offset+s)*KP);
                        // !GRAD: the trial points ([nseg][KP], own_offset = 0)
  int64_t own_offset;
  const double* other;  // opposing factor (global array, ld KP)
  int64_t n_other;      // opposing vectors
  const double* A;      // packed block of nseg_pad x lda entries (zero padded) in 16 x 16 tiles, see dense_tile_pos
  int64_t lda;
  double scale;         // QuadLoss scale
  int nsup;             // super-tiles over the opposing dimension
  int64_t vec_per_sup;  // opposing vectors per super-tile (multiple of TN)
  double* part;         // [nseg][nsup][KP+2]
  const int32_t* active; // !GRAD: skip waves without an active segment
};

constexpr int DENSE_TN = 64; // opposing vectors staged per barrier

template <int KP>
constexpr int dense_row_bytes() { return KP * 8 + 16; }

// NWD waves per workgroup (4 or 16; all share the staged tile: larger workgroups re-stage the opposing factor less
// often).  The lane's A entries are loaded per 16-vector tile right before use; requesting the whole staged tile's
// entries ahead of the barrier measured 4-7 % slower (profiles/r01_dense_c3_pmc_summary.md).
template <int KP, bool GRAD, int NWD>
__global__ void __launch_bounds__(NWD * 64) dense_pass_kernel(const DenseArgs a) {
  constexpr int ROWB = dense_row_bytes<KP>(), PSTRIDE = KP + 2, NQ = KP / 4, NCB = KP / 16;
  __shared__ __attribute__((aligned(16))) char lds2[2 * DENSE_TN * ROWB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = lane & 15, cq = lane >> 4;
  const int64_t seg0 = (int64_t)blockIdx.x * (NWD * 16) + wave * 16;
  const int sup = blockIdx.y;
  const int64_t seg = seg0 + e;
  const bool have = seg < a.nseg;
  bool wave_active = true;
  if constexpr (!GRAD) {
    const int act = have ? a.active[seg] : 0;
    wave_active = __any(act != 0);
  }
  // B operands of GEMM1: X[kk = 4q + cq][e]
  double xb[NQ];
  const double* xp = a.xsrc + (a.own_offset + (have ? seg : 0)) * KP;
#pragma unroll
  for (int q = 0; q < NQ; ++q) xb[q] = have ? xp[4 * q + cq] : 0.0;
  f64x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
  double J = 0.0;

  const int64_t v0 = (int64_t)sup * a.vec_per_sup;
  const int64_t v1 = v0 + a.vec_per_sup < a.n_other ? v0 + a.vec_per_sup : a.n_other;
  // A is packed in 16 x 16 tiles (dense_pack_kernel): the tile of segments [seg0, seg0+16) x opposing vectors [16 C, 16 C + 16) is one
  // contiguous 2 KB block whose first KB holds, at double2 index `lane`, the lane's entries 4 cq + {0,1} and whose second KB holds
  // 4 cq + {2,3} -- each of the wave's two loads per tile reads one fully coalesced KB, and a wave streams 8 KB of consecutive
  // addresses per stage (with row-major A every load touched 16 rows 80 KB apart, 64 useful bytes in each; together with the
  // double-buffered tiles: 84.8 -> 78.0 ms per C3 iteration, same box)
  const double2* atile = reinterpret_cast<const double2*>(a.A) + (seg0 / 16) * (a.lda / 16) * 128 + lane; // 128 double2 per tile
  const int i1 = lane & 15;               // GEMM1 A-operand: tile index i -> local vector 4*(i&3) + (i>>2)
  const int vloc1 = 4 * (i1 & 3) + (i1 >> 2);

  // The staged tiles are double-buffered: the next DENSE_TN opposing vectors are requested (into registers) before the current ones
  // are consumed and written to the other buffer afterwards -- one barrier per stage, the load latency hidden behind 4 x 16 MFMAs
  // per wave (single buffer, load + two barriers per stage: 84.8 vs 81.0 ms per C3 iteration, same box).
  constexpr int STAGE_B = DENSE_TN * KP * 8, BUF_B = DENSE_TN * ROWB, NP = (STAGE_B + NWD * 64 * 16 - 1) / (NWD * 64 * 16);
  double2 sv[NP];
  auto fetch = [&](int64_t t0) { // DENSE_TN opposing vectors from t0 on (zero beyond n_other)
    const char* src = reinterpret_cast<const char*>(a.other) + t0 * (KP * 8);
    const int64_t valid = (a.n_other - t0) * (KP * 8);
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int off = (pc * NWD * 64 + (int)threadIdx.x) * 16;
      sv[pc] = make_double2(0.0, 0.0);
      if (off < STAGE_B && off < valid) sv[pc] = *reinterpret_cast<const double2*>(src + off);
    }
  };
  auto put = [&](char* buf) { // padded rows
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) {
      const int off = (pc * NWD * 64 + (int)threadIdx.x) * 16;
      if (off < STAGE_B) {
        const int row = off / (KP * 8), col = off - row * (KP * 8);
        *reinterpret_cast<double2*>(buf + row * ROWB + col) = sv[pc];
      }
    }
  };
  if (v0 < v1) {
    fetch(v0);
    put(lds2);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t t0 = v0; t0 < v1; t0 += DENSE_TN, cur ^= 1) {
    const bool more = t0 + DENSE_TN < v1;
    if (more) fetch(t0 + DENSE_TN);
    const char* lds = lds2 + cur * BUF_B;
    if (wave_active) {
#pragma unroll
    for (int ct = 0; ct < DENSE_TN / 16; ++ct) {
      // the lane's four A entries: A[seg][t0 + ct*16 + 4*cq + (0..3)]
      const double2* ap = atile + ((t0 >> 4) + ct) * 128;
      const double2 a01 = ap[0], a23 = ap[64];
      // GEMM1: residual tile (transposed) = Y_tile' X
      f64x4 d = f64x4{0.0, 0.0, 0.0, 0.0};
      const char* y1 = lds + (ct * 16 + vloc1) * ROWB + cq * 8;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const double ya = *reinterpret_cast<const double*>(y1 + q * 32);
        d = __builtin_amdgcn_mfma_f64_16x16x4f64(ya, xb[q], d, 0, 0, 0);
      }
      const double r0 = d[0] - a01.x, r1 = d[1] - a01.y, r2 = d[2] - a23.x, r3 = d[3] - a23.y;
      J = fma(r0, r0, J);
      J = fma(r1, r1, J);
      J = fma(r2, r2, J);
      J = fma(r3, r3, J);
      if constexpr (GRAD) {
        // GEMM2: G += R Y_tile'; slab r uses local vectors 4*cq + r
        const char* y2 = lds + (ct * 16 + 4 * cq) * ROWB + (lane & 15) * 8;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(r0, *reinterpret_cast<const double*>(y2 + 0 * ROWB + cb * 128), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(r1, *reinterpret_cast<const double*>(y2 + 1 * ROWB + cb * 128), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(r2, *reinterpret_cast<const double*>(y2 + 2 * ROWB + cb * 128), acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(r3, *reinterpret_cast<const double*>(y2 + 3 * ROWB + cb * 128), acc[cb], 0, 0, 0);
        }
      }
    }
    }
    if (more) put(lds2 + (cur ^ 1) * BUF_B);
    __syncthreads(); // the other buffer is complete, everybody is done with this one
  }
  if (!wave_active) return;
  // J of segment e: add the four cq partials (lanes e, e+16, e+32, e+48)
  J += __shfl_xor(J, 16, 64);
  J += __shfl_xor(J, 32, 64);
  if (have && cq == 0) a.part[((int64_t)seg * a.nsup + sup) * PSTRIDE + KP] = a.scale * J;
  if constexpr (GRAD) {
    // D2[i = segment (cq + 4r)][j = component 16cb + e]
    const double two_s = 2 * a.scale;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t s2 = seg0 + cq + 4 * r;
        if (s2 < a.nseg) a.part[((int64_t)s2 * a.nsup + sup) * PSTRIDE + cb * 16 + e] = two_s * acc[cb][r];
      }
    }
  }
}

// ---- glrm_options.quad_gram (SURVEY.md 7.2 "K5"): trials without another pass over A -------------------------------------------
// The loss of a fully observed QuadLoss segment is a quadratic in its factor vector: with r_c = x.y_c - a_c and s = x' - x,
//   J(x') = scale sum_c (r_c + s.y_c)^2 = J(x) + g.s + scale s'Hs,   g = 2 scale sum_c r_c y_c,   H = sum_c y_c y_c' = Y Y',
// and H is the SAME k x k matrix for every segment of the half-step because every segment sees all opposing vectors.  The gradient
// pass already delivers J(x) and g; H costs one pass over the opposing FACTOR (not over A); every trial is then O(k^2) per segment.
// Deterministic: H is summed in GRAM_BLOCKS fixed chunks of the opposing vectors (chunking depends on their number only), each chunk
// sequentially, the chunk sums in order.
constexpr int GRAM_BLOCKS = 256;

template <int KP>
__global__ void __launch_bounds__(256) dense_gram_partial_kernel(const double* __restrict__ other, int64_t n_other, double* __restrict__ part) {
  constexpr int PAIRS = KP * KP, PER = (PAIRS + 255) / 256, TV = 32; // vectors staged per round
  __shared__ double v[TV][KP];
  const int64_t chunk = (n_other + GRAM_BLOCKS - 1) / GRAM_BLOCKS;
  const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n_other ? lo + chunk : n_other;
  double acc[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) acc[q] = 0.0;
  for (int64_t t0 = lo; t0 < hi; t0 += TV) {
    __syncthreads();
    for (int e = threadIdx.x; e < TV * KP; e += 256) {
      const int64_t vec = t0 + e / KP;
      v[e / KP][e % KP] = vec < hi ? other[vec * KP + e % KP] : 0.0;
    }
    __syncthreads();
    for (int t = 0; t < TV; ++t) {
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int pr = threadIdx.x + 256 * q;
        if (pr < PAIRS) acc[q] = fma(v[t][pr / KP], v[t][pr % KP], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int pr = threadIdx.x + 256 * q;
    if (pr < PAIRS) part[(int64_t)blockIdx.x * PAIRS + pr] = acc[q];
  }
}

template <int KP>
__global__ void __launch_bounds__(256) dense_gram_final_kernel(const double* __restrict__ part, double* __restrict__ H) {
  const int pr = blockIdx.x * 256 + threadIdx.x;
  if (pr >= KP * KP) return;
  double s = 0.0;
  for (int b = 0; b < GRAM_BLOCKS; ++b) s += part[(int64_t)b * KP * KP + pr];
  H[pr] = s;
}

// One thread per segment: s = trial - own, J' = jloss + g.s + scale s'Hs, delivered where the trial pass would have left its
// partial sums (super-tile 0 carries the value, the others zero) so that col_decide_kernel runs unchanged.
template <int KP>
__global__ void __launch_bounds__(256) dense_gram_trial_kernel(const TiledArgs a, const double* __restrict__ Hg, double scale) {
  __shared__ double H[KP * KP];
  for (int e = threadIdx.x; e < KP * KP; e += 256) H[e] = Hg[e];
  __syncthreads();
  const int64_t seg = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (seg >= a.nseg || a.active[seg] == 0) return;
  const double* own = a.own + (a.own_offset + seg) * KP;
  const double* tr = a.trial + seg * (int64_t)KP;
  const double* g = a.gsum + seg * (int64_t)KP;
  double s[KP];
  double gs = 0.0;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    s[i] = tr[i] - own[i];
    gs = fma(g[i], s[i], gs);
  }
  double q = 0.0;
#pragma unroll
  for (int i = 0; i < KP; ++i) { // rows of H from LDS (every lane reads the same word: broadcast)
    double t = 0.0;
#pragma unroll
    for (int jj = 0; jj < KP; ++jj) t = fma(H[i * KP + jj], s[jj], t);
    q = fma(s[i], t, q);
  }
  constexpr int PSTRIDE = KP + 2;
  double* p = a.part + (int64_t)seg * a.nsup * PSTRIDE + KP;
  p[0] = (a.jloss[seg] + gs) + scale * q;
  for (int sp = 1; sp < a.nsup; ++sp) p[(int64_t)sp * PSTRIDE] = 0.0;
}

// Where entry (segment s, opposing vector c) of a packed block lives: 16 x 16 tiles, tile (s / 16, c / 16) at ((s/16) (lda/16) + c/16)
// x 256 doubles; inside the tile the lane that needs the entry in dense_pass_kernel is lane = 16 (cc / 4) + e (e = s % 16, cc = c %
// 16) and the entry is double (cc % 2) of double2 `lane` of half (cc % 4) / 2.
__host__ __device__ __forceinline__ int64_t dense_tile_pos(int64_t s, int64_t c, int64_t lda) {
  const int e = (int)(s & 15), cc = (int)(c & 15);
  return ((s >> 4) * (lda >> 4) + (c >> 4)) * 256 + ((cc & 3) >> 1) * 128 + (((cc >> 2) << 4) + e) * 2 + (cc & 1);
}

// Pack a block of the caller's dense matrix into the padded tile layout the pass kernel streams (dense_tile_pos):
//   dst[pos(s, c)] = A(seg0 + s, c)  for the row view   (transpose = 0)
//   dst[pos(s, c)] = A(c, seg0 + s)  for the column view (transpose = 1)
// A(i,j) = src[i + j*ldsrc] if colmajor else src[i*ldsrc + j].  32x32 tiles through LDS keep both sides coalesced.
__global__ void __launch_bounds__(256) dense_pack_kernel(const double* src, int64_t ldsrc, int colmajor, int transpose,
                                                         int64_t seg0, int64_t nseg, int64_t nother, double* dst, int64_t lda) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  const int64_t s0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  // element (s, c) of dst comes from A(i, j) with (i, j) = transpose ? (c, seg0+s) : (seg0+s, c)
  // read side: make the fastest-varying source index follow tx
  const bool src_fast_is_c = transpose ? (colmajor != 0) : (colmajor == 0); // source contiguous along c?
  for (int yy = ty; yy < 32; yy += 8) {
    int64_t s, c;
    if (src_fast_is_c) { s = s0 + yy; c = c0 + tx; } else { s = s0 + tx; c = c0 + yy; }
    double v = 0.0;
    if (s < nseg && c < nother) {
      const int64_t i = transpose ? c : seg0 + s, j = transpose ? seg0 + s : c;
      v = colmajor ? src[i + j * ldsrc] : src[i * ldsrc + j];
    }
    if (src_fast_is_c) tile[yy][tx] = v; else tile[tx][yy] = v; // tile[s_local][c_local]
  }
  __syncthreads();
  for (int yy = ty; yy < 32; yy += 8) {
    const int64_t s = s0 + yy, c = c0 + tx;
    if (c < lda) dst[dense_tile_pos(s, c, lda)] = tile[yy][tx]; // dst rows are allocated up to a multiple of 256 >= nseg
  }
}

} // namespace glrm
