// glrm_tilesort.hip -- bring an Omega view into TILE order on the device: a stable segmented radix sort of the engine's private
// copy by tile index (idx / tile).  The LDS-tiled sweeps need the entries of tile t before those of tile t+1; lists that arrive in
// arbitrary order (obs tuples pushed in sampling order, src/modify_glrm.jl:8-12) would otherwise fall back to the gather sweeps.
// Inside a tile the original list order is kept (stable sort), so sums change by rounding only.
#include <hipcub/hipcub.hpp>

#include "glrm_engine.hpp"

namespace {

__global__ void tile_keys_kernel(const int32_t* idx, int64_t nnz, int tile, uint32_t* keys, int64_t* pos) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
    keys[t] = (uint32_t)(idx[t] / tile);
    pos[t] = t;
  }
}

__global__ void apply_perm_kernel(const int64_t* perm, int64_t nnz, const int32_t* idx, const double* vals, int32_t* oidx, double* ovals) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = perm[t];
    oidx[t] = idx[s];
    ovals[t] = vals[s];
  }
}

} // namespace

// Reorders (*idx, *vals) in place (new allocations replace the old ones).  n_other = size of the index space (for the key width).
int glrm_tile_sort_view(hipStream_t st, const int64_t* ptr, int64_t nseg, int64_t nnz, int tile, int64_t n_other, int32_t** idx, double** vals) {
  if (nnz <= 0 || nseg <= 0) return GLRM_OK;
  if (nnz > 2000000000ll || nseg > 2000000000ll) return fail(GLRM_ERR_UNSUPPORTED, "view too large for the segmented sort"); // hipcub counts in int
  uint32_t *k0 = nullptr, *k1 = nullptr;
  int64_t *p0 = nullptr, *p1 = nullptr;
  int32_t* oidx = nullptr;
  double* ovals = nullptr;
  void* tmp = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)k0, (void*)k1, (void*)p0, (void*)p1, tmp}) if (p) (void)hipFree(p);
    if (rc) { if (oidx) (void)hipFree(oidx); if (ovals) (void)hipFree(ovals); }
    return rc;
  };
  if (hipMalloc((void**)&k0, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&k1, (size_t)nnz * 4) != hipSuccess ||
      hipMalloc((void**)&p0, (size_t)nnz * 8) != hipSuccess || hipMalloc((void**)&p1, (size_t)nnz * 8) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the tile sort"));
  hipLaunchKernelGGL(tile_keys_kernel, dim3(4096), dim3(256), 0, st, *idx, nnz, tile, k0, p0);
  int bits = 1;
  while (((int64_t)1 << bits) < (n_other + tile - 1) / tile) ++bits;
  size_t bytes = 0;
  if (hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, k0, k1, p0, p1, (int)nnz, (int)nseg, ptr, ptr + 1, 0, bits, st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "segmented sort (size query) failed"));
  if (hipMalloc(&tmp, bytes ? bytes : 1) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the tile sort"));
  if (hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, bytes, k0, k1, p0, p1, (int)nnz, (int)nseg, ptr, ptr + 1, 0, bits, st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "segmented sort failed"));
  if (hipMalloc((void**)&oidx, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&ovals, (size_t)nnz * 8) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the tile sort"));
  hipLaunchKernelGGL(apply_perm_kernel, dim3(4096), dim3(256), 0, st, p1, nnz, *idx, *vals, oidx, ovals);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "tile sort kernels failed"));
  (void)hipFree(*idx);
  (void)hipFree(*vals);
  *idx = oidx;
  *vals = ovals;
  return cleanup(GLRM_OK);
}
