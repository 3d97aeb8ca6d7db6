// glrm_synth.hip -- libglrm_synth.so: device-side generator of the synthetic GLRM workloads of
// include/glrm_synth.h.  Measurement / test tooling: it lets bench.py and
// the full-size GPU tests build BASELINE-scale inputs (5e8 observations and more) directly in HBM
// in seconds instead of generating them on the host and pushing 12+ GB over PCIe.  It is not part
// of the drop-in boundary and the engine (libglrm_hip.so) does not depend on it.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/glrm_synth.h"

static thread_local char g_serr[256];
extern "C" const char* glrm_synth_hip_last_error(void) { return g_serr; }

#define SCK(expr)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      snprintf(g_serr, sizeof g_serr, "%s failed: %s", #expr, hipGetErrorString(e_));     \
      return -3;                                                                          \
    }                                                                                     \
  } while (0)

static bool spec_ok(const glrm_synth_spec* s) {
  if (!(s && s->m > 0 && s->n > 0 && s->k > 0 && s->q > 0 && s->q <= s->n && s->n % s->q == 0)) {
    snprintf(g_serr, sizeof g_serr, "invalid glrm_synth_spec (q must divide n)");
    return false;
  }
  return true;
}

// one thread per (row, t): CSR entries are exactly q per row
__global__ void synth_rows_kernel(glrm_synth_spec s, int64_t row_begin, int64_t nrows, int64_t* rowptr, int32_t* colidx,
                                  double* vals) {
  const int64_t total = nrows * s.q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t el = i / s.q;
    const int32_t t = (int32_t)(i - el * s.q);
    const int64_t e = row_begin + el;
    const int32_t c = glrm_synth_col(&s, e, t);
    colidx[i] = c;
    vals[i] = glrm_synth_value(&s, e, c);
    if (t == 0) rowptr[el] = el * s.q;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rowptr[nrows] = total;
}

// one block per column: count the rows e whose stratum-t pick is this column
__global__ void __launch_bounds__(256) synth_col_count_kernel(glrm_synth_spec s, int64_t col_begin, int64_t* colptr) {
  const int64_t f = col_begin + blockIdx.x;
  const int64_t S = s.n / s.q;
  const int32_t t = (int32_t)(f / S);
  unsigned long long cnt = 0;
  for (int64_t e = threadIdx.x; e < s.m; e += blockDim.x) cnt += (glrm_synth_col(&s, e, t) == f);
  __shared__ unsigned long long sh[256];
  sh[threadIdx.x] = cnt;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    colptr[blockIdx.x + 1] = (int64_t)sh[0];
    if (blockIdx.x == 0) colptr[0] = 0;
  }
}

// one block (256 threads = 4 waves) per column: ordered compaction of the matching rows
__global__ void __launch_bounds__(256) synth_cols_kernel(glrm_synth_spec s, int64_t col_begin, const int64_t* colptr,
                                                         int32_t* rowidx, double* vals) {
  const int64_t f = col_begin + blockIdx.x;
  const int64_t S = s.n / s.q;
  const int32_t t = (int32_t)(f / S);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int wcnt[4];
  int64_t pos = colptr[blockIdx.x];
  for (int64_t e0 = 0; e0 < s.m; e0 += 256) {
    const int64_t e = e0 + threadIdx.x;
    const bool hit = e < s.m && glrm_synth_col(&s, e, t) == f;
    const unsigned long long mask = __ballot(hit);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wave] = __popcll(mask);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) base += wcnt[w];
      tot += wcnt[w];
    }
    if (hit) {
      rowidx[pos + base + before] = (int32_t)e;
      vals[pos + base + before] = glrm_synth_value(&s, e, f);
    }
    pos += tot;
    __syncthreads();
  }
}

__device__ inline double box_muller(uint64_t seed, uint64_t stream, uint64_t i, uint64_t j) {
  const double u1 = glrm_unif(glrm_hash4(seed, stream, i, j));
  const double u2 = glrm_unif(glrm_hash4(seed, stream + 100, i, j));
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ void synth_init_kernel(uint64_t seed, uint64_t stream, int64_t count, int k, int ld, double* F) {
  const int64_t total = count * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = i / ld;
    const int c = (int)(i - s * ld);
    F[i] = c < k ? box_muller(seed, stream, (uint64_t)s, (uint64_t)c) : 0.0;
  }
}

// Dense matrix block (rows [row_begin,row_end) x all n columns, row-major, leading dimension ld): same values as
// glrm_synth_value, computed from tabulated truth factors (xs: rows x k, ys: n x k, filled by synth_factor_kernel).
__global__ void synth_factor_kernel(glrm_synth_spec s, int which, int64_t first, int64_t count, double* out) {
  const double isk = 1.0 / sqrt((double)s.k);
  const int64_t total = count * s.k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / s.k;
    const int c = (int)(i - r * s.k);
    out[i] = which == 0 ? glrm_synth_xstar(&s, first + r, c, isk) : glrm_synth_ystar(&s, first + r, c, isk);
  }
}

__global__ void __launch_bounds__(256) synth_dense_kernel(glrm_synth_spec s, int64_t row_begin, int64_t nrows, const double* xs,
                                                          const double* ys, double* A, int64_t ld) {
  // one block = 16 rows x 256 columns; thread = one column, loops the 16 rows
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 16;
  if (f >= s.n) return;
  const double* y = ys + f * s.k;
  for (int rr = 0; rr < 16 && r0 + rr < nrows; ++rr) {
    const int64_t e = row_begin + r0 + rr;
    const double* x = xs + (r0 + rr) * s.k;
    double d = 0.0;
    for (int c = 0; c < s.k; ++c) d = fma(x[c], y[c], d);
    A[(r0 + rr) * ld + f] = d + s.noise * glrm_unif_unit(glrm_hash4(s.seed, 4, (uint64_t)e, (uint64_t)f));
  }
}

extern "C" int glrm_synth_hip_dense(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, double* A, int64_t ld,
                                    double* scratch_xs, double* scratch_ys, void* stream) {
  if (!spec_ok(s) || row_begin < 0 || row_end > s->m || row_begin > row_end || ld < s->n || s->loss_mix) return -1;
  const int64_t nrows = row_end - row_begin;
  hipLaunchKernelGGL(synth_factor_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, *s, 0, row_begin, nrows, scratch_xs);
  hipLaunchKernelGGL(synth_factor_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, *s, 1, (int64_t)0, s->n, scratch_ys);
  const dim3 grid((unsigned)((s->n + 255) / 256), (unsigned)((nrows + 15) / 16));
  hipLaunchKernelGGL(synth_dense_kernel, grid, dim3(256), 0, (hipStream_t)stream, *s, row_begin, nrows, scratch_xs, scratch_ys, A, ld);
  SCK(hipGetLastError());
  return 0;
}

extern "C" int glrm_synth_hip_rows(const glrm_synth_spec* s, int64_t row_begin, int64_t row_end, int64_t* rowptr,
                                   int32_t* colidx, double* vals, void* stream) {
  if (!spec_ok(s) || row_begin < 0 || row_end > s->m || row_begin > row_end) return -1;
  const int64_t nrows = row_end - row_begin;
  hipLaunchKernelGGL(synth_rows_kernel, dim3(256 * 32), dim3(256), 0, (hipStream_t)stream, *s, row_begin, nrows, rowptr, colidx, vals);
  SCK(hipGetLastError());
  return 0;
}

extern "C" int glrm_synth_hip_col_counts(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, int64_t* colptr,
                                         void* stream) {
  if (!spec_ok(s) || col_begin < 0 || col_end > s->n || col_begin > col_end) return -1;
  if (col_end > col_begin)
    hipLaunchKernelGGL(synth_col_count_kernel, dim3((unsigned)(col_end - col_begin)), dim3(256), 0, (hipStream_t)stream, *s, col_begin, colptr);
  SCK(hipGetLastError());
  return 0;
}

extern "C" int glrm_synth_hip_cols(const glrm_synth_spec* s, int64_t col_begin, int64_t col_end, const int64_t* colptr,
                                   int32_t* rowidx, double* vals, void* stream) {
  if (!spec_ok(s) || col_begin < 0 || col_end > s->n || col_begin > col_end) return -1;
  if (col_end > col_begin)
    hipLaunchKernelGGL(synth_cols_kernel, dim3((unsigned)(col_end - col_begin)), dim3(256), 0, (hipStream_t)stream, *s, col_begin, colptr, rowidx, vals);
  SCK(hipGetLastError());
  return 0;
}

extern "C" int glrm_synth_hip_init(const glrm_synth_spec* s, uint64_t init_seed, int ld, double* X, double* Y, void* stream) {
  if (!spec_ok(s) || ld < s->k) return -1;
  hipLaunchKernelGGL(synth_init_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, init_seed, 7ull, s->m, s->k, ld, X);
  hipLaunchKernelGGL(synth_init_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, init_seed, 8ull, s->n, s->k, ld, Y);
  SCK(hipGetLastError());
  return 0;
}
