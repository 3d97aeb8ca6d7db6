// glrm_impute.hip -- glrm_hip_error_metric / glrm_hip_impute: post-fit evaluation (src/evaluate_fit.jl:107-168,
// src/impute_and_err.jl) on the resident column view.  One thread per entry: d dot products of length k into a per-thread LDS
// strip, then the domain's imputation rule.  Not a hot path: a single pass per call.
#include <cmath>
#include <vector>

#include "glrm_engine.hpp"
#include "glrm_impute.hpp"

using namespace glrm;

namespace {

constexpr int ET = 128;

struct EvalArgs {
  const int64_t* colptr;
  const int32_t* rowidx;
  const double* colvals;
  const glrm_loss* losses;
  int loss_single;
  const int64_t* ystart;
  const glrm_domain* domains;
  const double* X;
  const double* Y;
  int k, kp, dmax;
  int64_t m;
  int64_t chunk;
  int nsplit;
  double* part;   // [n][nsplit][2]: error sum, sum of A^2
  int* bad;
  double* Ahat;   // impute: m x n column-major
};

// u_j = <x_e, y_{ys+j}> for j < d into us[j * ET] (the thread's strip)
__device__ __forceinline__ void dots(const EvalArgs& a, int64_t e, int64_t ys, int d, double* us) {
  const double* x = a.X + e * a.kp;
  for (int j = 0; j < d; ++j) {
    const double* y = a.Y + (ys + j) * a.kp;
    double u = 0.0;
    for (int c = 0; c < a.k; ++c) u = fma(x[c], y[c], u);
    us[j * ET] = u;
  }
}

__global__ void __launch_bounds__(ET) error_metric_kernel(const EvalArgs a) {
  extern __shared__ double sm[];
  double* us = sm + threadIdx.x;
  __shared__ double r0[ET], r1[ET];
  const int64_t f = blockIdx.x;
  const int y = blockIdx.y;
  const glrm_loss lo = a.losses[a.loss_single ? 0 : f];
  const LossDesc l = load_loss(a.losses, a.loss_single ? 0 : f);
  const int d = lo.dim > 1 ? lo.dim : 1;
  const glrm_domain D = a.domains[f];
  const int64_t ys = a.ystart[f];
  int64_t b = a.colptr[f] + (int64_t)y * a.chunk, e = b + a.chunk;
  const int64_t e0 = a.colptr[f + 1];
  b = b < e0 ? b : e0;
  e = e < e0 ? e : e0;
  double err = 0.0, sq = 0.0;
  int bad = 0;
  for (int64_t t = b + threadIdx.x; t < e; t += ET) {
    const double av = a.colvals[t];
    dots(a, a.rowidx[t], ys, d, us);
    const double imp = impute_value(D, l, d, us, ET, &bad);
    err += entry_error(D, imp, av);
    sq = fma(av, av, sq);
  }
  if (bad) atomicOr(a.bad, 1);
  r0[threadIdx.x] = err;
  r1[threadIdx.x] = sq;
  __syncthreads();
  for (int w = ET / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { r0[threadIdx.x] += r0[threadIdx.x + w]; r1[threadIdx.x] += r1[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.part[((size_t)f * a.nsplit + y) * 2] = r0[0];
    a.part[((size_t)f * a.nsplit + y) * 2 + 1] = r1[0];
  }
}

// per column: chunks in order, optional standardization (src/evaluate_fit.jl:123-135)
__global__ void error_metric_cols_kernel(const EvalArgs a, int64_t n, int standardize, double* colerr) {
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n; f += (int64_t)gridDim.x * blockDim.x) {
    double err = 0.0, sq = 0.0;
    for (int y = 0; y < a.nsplit; ++y) { err += a.part[((size_t)f * a.nsplit + y) * 2]; sq += a.part[((size_t)f * a.nsplit + y) * 2 + 1]; }
    if (standardize) {
      const double column_mean = sq / (double)(a.colptr[f + 1] - a.colptr[f]); // 0/0 = NaN for an empty column, like the reference
      if (column_mean != 0) err = err / column_mean;
    }
    colerr[f] = err;
  }
}

__global__ void __launch_bounds__(ET) impute_kernel(const EvalArgs a) {
  extern __shared__ double sm[];
  double* us = sm + threadIdx.x;
  const int64_t nchunk = (a.m + ET - 1) / ET;             // 1-D grid: gridDim.y would overflow for m > 8.4e6
  const int64_t f = (int64_t)blockIdx.x / nchunk;
  const int64_t e = ((int64_t)blockIdx.x - f * nchunk) * ET + threadIdx.x;
  if (e >= a.m) return;
  const glrm_loss lo = a.losses[a.loss_single ? 0 : f];
  const LossDesc l = load_loss(a.losses, a.loss_single ? 0 : f);
  const int d = lo.dim > 1 ? lo.dim : 1;
  int bad = 0;
  dots(a, e, a.ystart[f], d, us);
  a.Ahat[e + f * a.m] = impute_value(a.domains[f], l, d, us, ET, &bad);
  if (bad) atomicOr(a.bad, 1);
}

int prepare(glrm_handle* h, const double* X, const double* Y, const glrm_domain* domains, EvalArgs& a, glrm_domain** ddom, int** dbad) {
  if (!(h->rb == 0 && h->re == h->m && h->cb == 0 && h->ce == h->n)) return fail(GLRM_ERR_INVALID, "needs a single-shard handle");
  if (h->dense) return fail(GLRM_ERR_UNSUPPORTED, "post-fit evaluation works on list handles (create the handle without dense_A)");
  if (!h->finalized) return fail(GLRM_ERR_INVALID, "the handle was created with GLRM_PROBLEM_DEFER_SETUP: call glrm_hip_finalize first");
  for (int64_t f = 0; f < h->n; ++f)
    if (domains[f].kind < 0 || domains[f].kind >= GLRM_DOMAIN_KIND_COUNT || domains[f].reserved != 0)
      return fail(GLRM_ERR_INVALID, "domain descriptor %lld is invalid", (long long)f);
  int rc = glrm_hip_set_factors(h, X, Y);
  if (rc) return rc;
  HIPCK(hipMalloc((void**)ddom, (size_t)h->n * sizeof(glrm_domain)));
  HIPCK(hipMalloc((void**)dbad, sizeof(int)));
  HIPCK(hipMemcpyAsync(*ddom, domains, (size_t)h->n * sizeof(glrm_domain), hipMemcpyHostToDevice, h->stream));
  HIPCK(hipMemsetAsync(*dbad, 0, sizeof(int), h->stream));
  a.colptr = h->colptr; a.rowidx = h->rowidx; a.colvals = h->colvals;
  a.losses = h->losses; a.loss_single = h->n_losses == 1 ? 1 : 0; a.ystart = h->ystart; a.domains = *ddom;
  a.X = h->X; a.Y = h->Y; a.k = h->k; a.kp = h->kp; a.dmax = h->dmax; a.m = h->m; a.bad = *dbad;
  return GLRM_OK;
}

} // namespace

extern "C" int glrm_hip_error_metric(glrm_handle* h, const double* X, const double* Y, const glrm_domain* domains, int32_t standardize, double* out) {
  if (!h || !X || !Y || !domains || !out) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (hipSetDevice(h->device) != hipSuccess) return fail(GLRM_ERR_HIP, "cannot select device %d", h->device);
  EvalArgs a{};
  glrm_domain* ddom = nullptr;
  int* dbad = nullptr;
  double *part = nullptr, *colerr = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)ddom, (void*)dbad, (void*)part, (void*)colerr}) if (p) (void)hipFree(p);
    return rc;
  };
  int rc = prepare(h, X, Y, domains, a, &ddom, &dbad);
  if (rc) return cleanup(rc);
  std::vector<int64_t> cp((size_t)h->n + 1);
  if (hipMemcpyAsync(cp.data(), h->colptr, ((size_t)h->n + 1) * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "copy failed"));
  int64_t longest = 0;
  for (int64_t f = 0; f < h->n; ++f) longest = cp[f + 1] - cp[f] > longest ? cp[f + 1] - cp[f] : longest;
  a.chunk = 65536;
  while ((longest + a.chunk - 1) / a.chunk > 65535) a.chunk *= 2; // gridDim.y
  a.nsplit = (int)((longest + a.chunk - 1) / a.chunk);
  if (a.nsplit < 1) a.nsplit = 1;
  if (hipMalloc((void**)&part, (size_t)h->n * a.nsplit * 16) != hipSuccess || hipMalloc((void**)&colerr, (size_t)h->n * 8) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory"));
  a.part = part;
  const size_t lds = (size_t)h->dmax * ET * 8;
  hipLaunchKernelGGL(error_metric_kernel, dim3((unsigned)h->n, (unsigned)a.nsplit), dim3(ET), lds, h->stream, a);
  hipLaunchKernelGGL(error_metric_cols_kernel, dim3(64), dim3(256), 0, h->stream, a, h->n, standardize, colerr);
  if (hipGetLastError() != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "error_metric kernels failed to launch"));
  if ((rc = glrm_hip_sum(h, colerr, h->n, out))) return cleanup(rc);
  int bad = 0;
  if (hipMemcpy(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "copy failed"));
  if (bad) return cleanup(fail(GLRM_ERR_UNSUPPORTED, "a column's (domain, loss) pair has no imputation rule in the reference (src/impute_and_err.jl)"));
  return cleanup(GLRM_OK);
}

extern "C" int glrm_hip_impute(glrm_handle* h, const double* X, const double* Y, const glrm_domain* domains, double* Ahat) {
  if (!h || !X || !Y || !domains || !Ahat) return fail(GLRM_ERR_INVALID, "NULL argument");
  if (hipSetDevice(h->device) != hipSuccess) return fail(GLRM_ERR_HIP, "cannot select device %d", h->device);
  EvalArgs a{};
  glrm_domain* ddom = nullptr;
  int* dbad = nullptr;
  double* dA = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)ddom, (void*)dbad, (void*)dA}) if (p) (void)hipFree(p);
    return rc;
  };
  int rc = prepare(h, X, Y, domains, a, &ddom, &dbad);
  if (rc) return cleanup(rc);
  if (hipMalloc((void**)&dA, (size_t)h->m * h->n * 8) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the m x n imputed matrix"));
  a.Ahat = dA;
  const size_t lds = (size_t)h->dmax * ET * 8;
  if ((h->m + ET - 1) / ET * h->n > 2147483647ll) return cleanup(fail(GLRM_ERR_UNSUPPORTED, "m x n too large for one impute launch"));
  hipLaunchKernelGGL(impute_kernel, dim3((unsigned)((h->m + ET - 1) / ET * h->n)), dim3(ET), lds, h->stream, a);
  if (hipGetLastError() != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "impute kernel failed to launch"));
  if (hipMemcpyAsync(Ahat, dA, (size_t)h->m * h->n * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "copy failed"));
  int bad = 0;
  if (hipMemcpy(&bad, dbad, sizeof bad, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "copy failed"));
  if (bad) return cleanup(fail(GLRM_ERR_UNSUPPORTED, "a column's (domain, loss) pair has no imputation rule in the reference (src/impute_and_err.jl)"));
  return cleanup(GLRM_OK);
}
