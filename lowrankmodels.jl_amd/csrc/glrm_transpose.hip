// glrm_transpose.hip -- GLRM_PROBLEM_ROWS_FROM_COLS: the row view (observed_features) derived ON THE DEVICE from the column view
// (observed_examples) for an Omega that is a sparse matrix's pattern.
//
// The reference builds both lists from `findall(!iszero, A)` for SparseMatrixCSC input (src/glrm.jl:46-48): CartesianIndices in
// column-major order pushed into per-row and per-column vectors (src/modify_glrm.jl:8-12) -- so observed_examples[j] is the column's
// rowval slice and observed_features[i] lists the SAME entries by ascending column.  A host that holds such a matrix hands over
// colptr / rowval / nzval only; the row view is one stable sort of the column-major stream by row id (entries of a row keep their column
// order), which the device does in a fraction of a second where a host needs a counting transpose of 1e9 entries (bench.py
// setup_s.create_from_host: 13.8 s in scipy) and a second 12 GB trip over PCIe.  Duplicates are kept (they appear in both views).
// Single-shard list problems with at most 1.5e9 observations (hipCUB counts items in int).
#include <hipcub/hipcub.hpp>

#include "glrm_engine.hpp"

namespace {

__global__ void iota_u32_kernel(uint32_t* p, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) p[t] = (uint32_t)t;
}

// rowptr[i] = first position of key i in the sorted key array (keys are row ids), rowptr[m] = nnz
__global__ void rowptr_from_sorted_kernel(const uint32_t* keys, int64_t nnz, int64_t m, int64_t* rowptr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = nnz; // first t with keys[t] >= i
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)keys[mid] < i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = lo;
  }
}

// entry t of the row view came from position pos[t] of the column view: its column = the segment of colptr that holds pos[t]
__global__ void gather_rows_kernel(const uint32_t* pos, int64_t nnz, const int64_t* colptr, int64_t n, const double* colvals, int32_t* colidx, double* rowvals) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = (int64_t)pos[t];
    int64_t lo = 0, hi = n; // last f with colptr[f] <= s
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (colptr[mid] <= s) lo = mid; else hi = mid;
    }
    colidx[t] = (int32_t)lo;
    rowvals[t] = colvals[s];
  }
}

__global__ void check_rowidx_kernel(const int32_t* rowidx, int64_t nnz, int64_t m, int* flag) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
    if (rowidx[t] < 0 || rowidx[t] >= m) *flag = 1;
}

} // namespace

// h holds the uploaded column view (colptr, rowidx, colvals, nnz_c) of the WHOLE problem; fills rowptr, colidx, rowvals, nnz_r.
int glrm_rows_from_cols(glrm_handle* h) {
  const int64_t nnz = h->nnz_c, m = h->m, n = h->n;
  hipStream_t st = h->stream;
  h->nnz_r = nnz;
  HIPCK(hipMalloc((void**)&h->rowptr, ((size_t)m + 1) * 8));
  HIPCK(hipMalloc((void**)&h->colidx, (size_t)(nnz > 0 ? nnz : 1) * 4));
  HIPCK(hipMalloc((void**)&h->rowvals, (size_t)(nnz > 0 ? nnz : 1) * 8));
  if (nnz == 0) {
    HIPCK(hipMemsetAsync(h->rowptr, 0, ((size_t)m + 1) * 8, st));
    return GLRM_OK;
  }
  if (nnz > 1500000000ll) return fail(GLRM_ERR_UNSUPPORTED, "GLRM_PROBLEM_ROWS_FROM_COLS: %lld observations (the device transpose takes at most 1.5e9: hand both views over)", (long long)nnz);
  uint32_t *k1 = nullptr, *p0 = nullptr, *p1 = nullptr;
  void* tmp = nullptr;
  int* flag = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)k1, (void*)p0, (void*)p1, tmp, (void*)flag}) if (p) (void)hipFree(p);
    return rc;
  };
  if (hipMalloc((void**)&k1, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&p0, (size_t)nnz * 4) != hipSuccess ||
      hipMalloc((void**)&p1, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&flag, 4) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (GLRM_PROBLEM_ROWS_FROM_COLS)"));
  if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "memset failed"));
  hipLaunchKernelGGL(check_rowidx_kernel, dim3(4096), dim3(256), 0, st, h->rowidx, nnz, m, flag);
  int bad = 0;
  if (hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "row index check failed"));
  if (bad) return cleanup(fail(GLRM_ERR_INVALID, "rowidx holds an index outside [0, m)"));
  hipLaunchKernelGGL(iota_u32_kernel, dim3(4096), dim3(256), 0, st, p0, nnz);
  int bits = 1;
  while (((int64_t)1 << bits) < m) ++bits;
  const uint32_t* k0 = reinterpret_cast<const uint32_t*>(h->rowidx); // non-negative int32: the same bits
  size_t bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k0, k1, p0, p1, (int)nnz, 0, bits, st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "radix sort (size query) failed"));
  if (hipMalloc(&tmp, bytes > 0 ? bytes : 1) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (sort scratch)"));
  if (hipcub::DeviceRadixSort::SortPairs(tmp, bytes, k0, k1, p0, p1, (int)nnz, 0, bits, st) != hipSuccess) // stable: a row's entries keep their column order
    return cleanup(fail(GLRM_ERR_HIP, "radix sort failed"));
  hipLaunchKernelGGL(rowptr_from_sorted_kernel, dim3(2048), dim3(256), 0, st, k1, nnz, m, h->rowptr);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(8192), dim3(256), 0, st, p1, nnz, h->colptr, n, h->colvals, h->colidx, h->rowvals);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "row view kernels failed"));
  return cleanup(GLRM_OK);
}
