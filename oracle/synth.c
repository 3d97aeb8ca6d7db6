/*
 * synth.c -- CPU side of the counter-based synthetic workload generator
 * (include/glrm_synth.h).  Test / measurement tooling:
 * it feeds the oracle, the parity tests and bench.py's cpu_baseline sample.
 */
#include "../include/glrm_synth.h"

#include <math.h>
#include <stdlib.h>

double glrm_synth_cpu_value(const glrm_synth_spec* s, int64_t e, int64_t f) { return glrm_synth_value(s, e, f); }

static int spec_ok(const glrm_synth_spec* s) {
  return s && s->m > 0 && s->n > 0 && s->k > 0 && s->q > 0 && s->q <= s->n && s->n % s->q == 0;
}

int glrm_synth_cpu_rows(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, int64_t* rowptr,
                        int32_t* colidx, double* vals) {
  if (!spec_ok(s) || row_begin < 0 || row_end > s->m || row_begin > row_end) return -1;
  const int64_t q = s->q;
#pragma omp parallel for schedule(static)
  for (int64_t e = row_begin; e < row_end; ++e) {
    const int64_t base = (e - row_begin) * q;
    rowptr[e - row_begin] = base;
    for (int32_t t = 0; t < q; ++t) {
      int32_t c = glrm_synth_col(s, e, t);
      colidx[base + t] = c;
      vals[base + t] = glrm_synth_value(s, e, c);
    }
  }
  rowptr[row_end - row_begin] = (row_end - row_begin) * q;
  return 0;
}

/* column f = t*S + r holds the rows e with hash(seed,1,e,t) mod S == r, ascending */
int glrm_synth_cpu_col_counts(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, int64_t* colptr) {
  if (!spec_ok(s) || col_begin < 0 || col_end > s->n || col_begin > col_end) return -1;
  const int64_t S = s->n / s->q;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t f = col_begin; f < col_end; ++f) {
    const int32_t t = (int32_t)(f / S);
    int64_t cnt = 0;
    for (int64_t e = 0; e < s->m; ++e) cnt += (glrm_synth_col(s, e, t) == f);
    colptr[f - col_begin + 1] = cnt;
  }
  colptr[0] = 0;
  for (int64_t i = 0; i < col_end - col_begin; ++i) colptr[i + 1] += colptr[i];
  return 0;
}

int glrm_synth_cpu_cols(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, const int64_t* colptr,
                        int32_t* rowidx, double* vals) {
  if (!spec_ok(s) || col_begin < 0 || col_end > s->n || col_begin > col_end) return -1;
  const int64_t S = s->n / s->q;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t f = col_begin; f < col_end; ++f) {
    const int32_t t = (int32_t)(f / S);
    int64_t pos = colptr[f - col_begin];
    for (int64_t e = 0; e < s->m; ++e) {
      if (glrm_synth_col(s, e, t) == f) {
        rowidx[pos] = (int32_t)e;
        vals[pos] = glrm_synth_value(s, e, f);
        ++pos;
      }
    }
  }
  return 0;
}

/* The column view of rows [0, m) x columns [col_begin, col_end) from the ROW view of the same block, by a stable counting sort on the
 * column index: O(nnz) instead of the O(columns x m) hash scan above, and the same arrays (rows ascending inside a column because the
 * row view is walked in row order; a value depends on (e, f) only) -- tests/test_synth.py compares the two.  Bench tooling: the CPU
 * legs of bench.py build their samples with it.  colptr: col_end - col_begin + 1 entries. */
int glrm_synth_cpu_cols_from_rows(int64_t m, int64_t col_begin, int64_t col_end, const int64_t* rowptr, const int32_t* colidx,
                                  const double* rowvals, int64_t* colptr, int32_t* rowidx, double* colvals) {
  const int64_t nc = col_end - col_begin;
  if (m < 0 || nc < 0) return -1;
  for (int64_t i = 0; i <= nc; ++i) colptr[i] = 0;
  const int64_t nnz = rowptr[m];
  for (int64_t t = 0; t < nnz; ++t) {
    const int64_t f = colidx[t] - col_begin;
    if (f < 0 || f >= nc) return -1;
    ++colptr[f + 1];
  }
  for (int64_t i = 0; i < nc; ++i) colptr[i + 1] += colptr[i];
  int64_t* pos = (int64_t*)malloc((size_t)(nc > 0 ? nc : 1) * sizeof(int64_t));
  if (!pos) return -1;
  for (int64_t i = 0; i < nc; ++i) pos[i] = colptr[i];
  for (int64_t e = 0; e < m; ++e)
    for (int64_t t = rowptr[e]; t < rowptr[e + 1]; ++t) {
      const int64_t p = pos[colidx[t] - col_begin]++;
      rowidx[p] = (int32_t)e;
      colvals[p] = rowvals[t];
    }
  free(pos);
  return 0;
}

static double box_muller(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
  double u1 = glrm_unif(glrm_hash4(seed, stream, i, j));
  double u2 = glrm_unif(glrm_hash4(seed, stream + 100, i, j));
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

int glrm_synth_cpu_init(const glrm_synth_spec* s, uint64_t init_seed, int ld, double* X, double* Y) {
  if (!spec_ok(s) || ld < s->k) return -1;
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < s->m; ++e)
    for (int c = 0; c < ld; ++c) X[e * ld + c] = c < s->k ? box_muller(init_seed, 7, (uint64_t)e, (uint64_t)c) : 0.0;
#pragma omp parallel for schedule(static)
  for (int64_t f = 0; f < s->n; ++f)
    for (int c = 0; c < ld; ++c) Y[f * ld + c] = c < s->k ? box_muller(init_seed, 8, (uint64_t)f, (uint64_t)c) : 0.0;
  return 0;
}
