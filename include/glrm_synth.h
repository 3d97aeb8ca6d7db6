/*
 * glrm_synth.h -- counter-based synthetic GLRM workloads (measurement / test tooling,
 * NOT part of the drop-in boundary).  Implements SURVEY.md section 8(d): inputs are a pure
 * function of (seed, stream, i, j), so the CPU generator (oracle/synth.c) and the device
 * generator (lowrankmodels.jl_amd/csrc/glrm_synth.hip) produce the same Omega bit-for-bit
 * and the same values (integer hash -> uniform, no libm in the data path).
 *
 * Omega: row e observes exactly q columns, one per stratum of width S = n/q:
 *     col(e,t) = t*S + hash(seed,1,e,t) mod S,  t = 0..q-1        (sorted, unique)
 * which is the order `findall(!iszero, A)` produces per row (src/glrm.jl:46-48), and the
 * CSC view lists each column's rows ascending (same call, column-major).
 * Values (value_model): truth factors x*_e, y*_f have iid entries of variance 1/k
 *     0: uniform(-sqrt3,sqrt3)/sqrt(k)         1: uniform(0,1)/sqrt(k)  (non-negative, NNMF)
 * and a column of kind
 *     Quad          a = x*.y* + noise*uniform(-sqrt3,sqrt3)
 *     Logistic      a = 1[uniform < sigmoid(x*.y*)]            (stored 1.0 / 0.0)
 *     OrdinalHinge  a = clamp(round(3 + 1.5 x*.y*), 1, 5)
 * loss_mix 0: every column Quad; 1: column f has kind (Quad, Logistic, OrdinalHinge)[f mod 3].
 */
#ifndef GLRM_SYNTH_H
#define GLRM_SYNTH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define GLRM_HD __host__ __device__ inline
#else
#define GLRM_HD static inline
#endif

typedef struct glrm_synth_spec {
  int64_t m, n;
  int32_t k;
  int32_t q;           /* observations per row; q must divide n */
  uint64_t seed;
  int32_t value_model; /* 0 signed factors, 1 non-negative factors */
  int32_t loss_mix;    /* 0 all Quad, 1 Quad/Logistic/OrdinalHinge by f mod 3 */
  double noise;        /* std of the additive noise on Quad columns */
} glrm_synth_spec;

GLRM_HD uint64_t glrm_mix64(uint64_t z) { /* SplitMix64 finaliser */
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

GLRM_HD uint64_t glrm_hash4(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
  uint64_t h = glrm_mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull));
  h = glrm_mix64(h ^ i);
  h = glrm_mix64(h ^ (j * 0xA24BAED4963EE407ull));
  return h;
}

/* uniform in (0,1), 53 bits */
GLRM_HD double glrm_unif(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
/* uniform with mean 0, variance 1 */
GLRM_HD double glrm_unif_unit(uint64_t h) { return (2.0 * glrm_unif(h) - 1.0) * 1.7320508075688772; }

GLRM_HD int32_t glrm_synth_col(const glrm_synth_spec* s, int64_t e, int32_t t) {
  const int64_t S = s->n / s->q;
  return (int32_t)(t * S + (int64_t)(glrm_hash4(s->seed, 1, (uint64_t)e, (uint64_t)t) % (uint64_t)S));
}

GLRM_HD double glrm_synth_xstar(const glrm_synth_spec* s, int64_t e, int c, double inv_sqrt_k) {
  uint64_t h = glrm_hash4(s->seed, 2, (uint64_t)e, (uint64_t)c);
  return (s->value_model == 1 ? glrm_unif(h) : glrm_unif_unit(h)) * inv_sqrt_k;
}
GLRM_HD double glrm_synth_ystar(const glrm_synth_spec* s, int64_t f, int c, double inv_sqrt_k) {
  uint64_t h = glrm_hash4(s->seed, 3, (uint64_t)f, (uint64_t)c);
  return (s->value_model == 1 ? glrm_unif(h) : glrm_unif_unit(h)) * inv_sqrt_k;
}

/* column kind under loss_mix: 0 Quad, 1 Logistic, 2 OrdinalHinge(1,5) */
GLRM_HD int glrm_synth_colkind(const glrm_synth_spec* s, int64_t f) { return s->loss_mix ? (int)(f % 3) : 0; }

/* A[e,f] of the synthetic matrix (only meaningful on observed entries). */
GLRM_HD double glrm_synth_value(const glrm_synth_spec* s, int64_t e, int64_t f) {
  const double isk = 1.0 / sqrt((double)s->k);
  double d = 0.0;
  for (int c = 0; c < s->k; ++c) d = fma(glrm_synth_xstar(s, e, c, isk), glrm_synth_ystar(s, f, c, isk), d);
  const int kind = glrm_synth_colkind(s, f);
  if (kind == 0) return d + s->noise * glrm_unif_unit(glrm_hash4(s->seed, 4, (uint64_t)e, (uint64_t)f));
  if (kind == 1) return glrm_unif(glrm_hash4(s->seed, 5, (uint64_t)e, (uint64_t)f)) < 1.0 / (1.0 + exp(-d)) ? 1.0 : 0.0;
  double v = round(3.0 + 1.5 * d);
  return v < 1.0 ? 1.0 : (v > 5.0 ? 5.0 : v);
}

#ifdef __cplusplus
extern "C" {
#endif

/* CPU generator (oracle/synth.c). */
double glrm_synth_cpu_value(const glrm_synth_spec* s, int64_t e, int64_t f);
int glrm_synth_cpu_rows(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, int64_t* rowptr,
                        int32_t* colidx, double* vals);
int glrm_synth_cpu_col_counts(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, int64_t* colptr);
int glrm_synth_cpu_cols(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, const int64_t* colptr,
                        int32_t* rowidx, double* vals);
/* the same column view from the row view of rows [0, m) (stable counting sort, O(nnz)); needs every listed column inside [col_begin, col_end) */
int glrm_synth_cpu_cols_from_rows(int64_t m, int64_t col_begin, int64_t col_end, const int64_t* rowptr, const int32_t* colidx,
                                  const double* rowvals, int64_t* colptr, int32_t* rowidx, double* colvals);
/* X0 (k x m, leading dimension ld) and Y0 (k x n): iid N(0,1) by Box-Muller, streams 7 / 8
 * (the reference default is randn, src/glrm.jl:31). */
int glrm_synth_cpu_init(const glrm_synth_spec* s, uint64_t init_seed, int ld, double* X, double* Y);

/* Device generator (libglrm_synth.so); every pointer is a device pointer on the current device. */
int glrm_synth_hip_rows(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, int64_t* rowptr,
                        int32_t* colidx, double* vals, void* stream);
int glrm_synth_hip_col_counts(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, int64_t* colptr,
                              void* stream); /* writes counts into colptr[1..], colptr[0]=0; caller scans */
int glrm_synth_hip_cols(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, const int64_t* colptr,
                        int32_t* rowidx, double* vals, void* stream);
int glrm_synth_hip_init(const glrm_synth_spec* s, uint64_t init_seed, int ld, double* X, double* Y, void* stream);
/* dense block: rows [row_begin,row_end) x n, row-major with leading dimension ld; scratch_xs (rows*k) and scratch_ys (n*k) doubles */
int glrm_synth_hip_dense(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, double* A, int64_t ld,
                         double* scratch_xs, double* scratch_ys, void* stream);
const char* glrm_synth_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
