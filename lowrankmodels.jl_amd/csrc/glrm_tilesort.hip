// glrm_tilesort.hip -- bring an Omega view into TILE order on the device: a stable segmented radix sort of the engine's private
// copy by tile index (idx / tile).  The LDS-tiled sweeps need the entries of tile t before those of tile t+1; lists that arrive in
// arbitrary order (obs tuples pushed in sampling order, src/modify_glrm.jl:8-12) would otherwise fall back to the gather sweeps.
// Inside a tile the original list order is kept (stable sort), so sums change by rounding only.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

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

struct MinusBase { // segment offsets relative to the batch's first entry
  int64_t base;
  __host__ __device__ int operator()(const int64_t& v) const { return (int)(v - base); }
};

// Reorders (*idx, *vals) in place (new allocations replace the old ones).  n_other = size of the index space (for the key width).
// hipCUB counts items and segments in int, so a view is sorted in batches of whole segments with at most `limit` entries each
// (1.5e9; GLRM_HIP_TILE_SORT_BATCH overrides, the tests use it to exercise the batching on small inputs); the scratch arrays are
// sized for one batch.  A single segment longer than the limit cannot be sorted this way (GLRM_ERR_UNSUPPORTED: gather sweeps).
// free_old = false: the old arrays are the caller's (GLRM_PROBLEM_BORROW_DEVICE_ARRAYS) and stay untouched.
int glrm_tile_sort_view(hipStream_t st, const int64_t* ptr, int64_t nseg, int64_t nnz, int tile, int64_t n_other, int32_t** idx, double** vals, bool free_old) {
  if (nnz <= 0 || nseg <= 0) return GLRM_OK;
  int64_t limit = env_int("GLRM_HIP_TILE_SORT_BATCH", 0);
  if (limit <= 0) limit = 1500000000ll;
  std::vector<int64_t> hp((size_t)nseg + 1);
  HIPCK(hipMemcpyAsync(hp.data(), ptr, ((size_t)nseg + 1) * 8, hipMemcpyDeviceToHost, st));
  HIPCK(hipStreamSynchronize(st));
  std::vector<int64_t> cuts{0}; // segment boundaries of the batches
  for (int64_t s = 0; s < nseg;) {
    int64_t e = s;
    while (e < nseg && hp[e + 1] - hp[s] <= limit && e - s < 2000000000ll) ++e;
    if (e == s) return fail(GLRM_ERR_UNSUPPORTED, "segment %lld holds %lld entries: too long for the segmented tile sort", (long long)s, (long long)(hp[s + 1] - hp[s]));
    cuts.push_back(e);
    s = e;
  }
  int64_t cap = 0;
  for (size_t b = 0; b + 1 < cuts.size(); ++b) cap = std::max(cap, hp[cuts[b + 1]] - hp[cuts[b]]);
  uint32_t *k0 = nullptr, *k1 = nullptr;
  int64_t *p0 = nullptr, *p1 = nullptr;
  int32_t* oidx = nullptr;
  double* ovals = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)k0, (void*)k1, (void*)p0, (void*)p1, tmp}) if (p) (void)hipFree(p);
    if (rc) { if (oidx) (void)hipFree(oidx); if (ovals) (void)hipFree(ovals); }
    return rc;
  };
  const size_t cap1 = (size_t)(cap > 0 ? cap : 1);
  if (hipMalloc((void**)&k0, cap1 * 4) != hipSuccess || hipMalloc((void**)&k1, cap1 * 4) != hipSuccess ||
      hipMalloc((void**)&p0, cap1 * 8) != hipSuccess || hipMalloc((void**)&p1, cap1 * 8) != hipSuccess ||
      hipMalloc((void**)&oidx, (size_t)nnz * 4) != hipSuccess || hipMalloc((void**)&ovals, (size_t)nnz * 8) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the tile sort"));
  int bits = 1;
  while (((int64_t)1 << bits) < (n_other + tile - 1) / tile) ++bits;
  for (size_t b = 0; b + 1 < cuts.size(); ++b) {
    const int64_t s0 = cuts[b], s1 = cuts[b + 1], base = hp[s0], cnt = hp[s1] - base;
    if (cnt <= 0) continue;
    hipLaunchKernelGGL(tile_keys_kernel, dim3(4096), dim3(256), 0, st, *idx + base, cnt, tile, k0, p0); // p0 = positions inside the batch
    hipcub::TransformInputIterator<int, MinusBase, const int64_t*> beg(ptr + s0, MinusBase{base}), end(ptr + s0 + 1, MinusBase{base});
    size_t bytes = 0;
    if (hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, k0, k1, p0, p1, (int)cnt, (int)(s1 - s0), beg, end, 0, bits, st) != hipSuccess)
      return cleanup(fail(GLRM_ERR_HIP, "segmented sort (size query) failed"));
    if (bytes > tmp_bytes) {
      if (tmp) (void)hipFree(tmp);
      tmp = nullptr;
      if (hipMalloc(&tmp, bytes) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the tile sort"));
      tmp_bytes = bytes;
    }
    if (hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, bytes, k0, k1, p0, p1, (int)cnt, (int)(s1 - s0), beg, end, 0, bits, st) != hipSuccess)
      return cleanup(fail(GLRM_ERR_HIP, "segmented sort failed"));
    hipLaunchKernelGGL(apply_perm_kernel, dim3(4096), dim3(256), 0, st, p1, cnt, *idx + base, *vals + base, oidx + base, ovals + base);
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "tile sort kernels failed"));
  if (free_old) {
    (void)hipFree(*idx);
    (void)hipFree(*vals);
  }
  *idx = oidx;
  *vals = ovals;
  return cleanup(GLRM_OK);
}
