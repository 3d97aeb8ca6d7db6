// glrm_subset.hip -- glrm_hip_subset: a child handle over a tagged subset of the parent's observations, compacted on the
// device (stream compaction of both Omega views).  Driver-level fusion for cross_validate / cv_by_iter / regularization_path
// (src/cross_validate.jl:9-53,141-240): the folds reuse the values and indices that are already resident in HBM.
#include "glrm_engine.hpp"

namespace {

constexpr int CT = 256, CI = 16, CB = CT * CI; // 4096 entries per workgroup

__device__ __forceinline__ bool keep_of(const uint8_t* tags, int64_t t, int match, int invert) {
  return ((int)tags[t] == match) != (invert != 0);
}

// kept entries per workgroup
__global__ void __launch_bounds__(CT) count_kernel(const uint8_t* tags, int64_t nnz, int match, int invert, int64_t* blockcount) {
  const int64_t base = (int64_t)blockIdx.x * CB;
  int c = 0;
  for (int i = 0; i < CI; ++i) {
    const int64_t t = base + (int64_t)i * CT + threadIdx.x;
    if (t < nnz) c += keep_of(tags, t, match, invert) ? 1 : 0;
  }
  __shared__ int sh[CT];
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int d = CT / 2; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) blockcount[blockIdx.x] = sh[0];
}

// exclusive scan of the workgroup counts (one workgroup; nb <= a few 100k)
__global__ void __launch_bounds__(1024) scan_blocks_kernel(int64_t* blockcount, int64_t nb, int64_t* total) {
  __shared__ int64_t sh[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += 1024) {
    const int64_t i = b0 + threadIdx.x;
    const int64_t v = i < nb ? blockcount[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan
      const int64_t add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) blockcount[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// position of every entry among the kept ones (exclusive), in list order: entry t of thread-chunk order is t itself
// (a thread owns CI CONSECUTIVE entries so the in-workgroup order is the list order)
__global__ void __launch_bounds__(CT) position_kernel(const uint8_t* tags, int64_t nnz, int match, int invert, const int64_t* blockbase,
                                                      int64_t* pos) {
  const int64_t base = (int64_t)blockIdx.x * CB + (int64_t)threadIdx.x * CI;
  int c = 0;
  for (int i = 0; i < CI; ++i) {
    const int64_t t = base + i;
    if (t < nnz) c += keep_of(tags, t, match, invert) ? 1 : 0;
  }
  __shared__ int sh[CT];
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int d = 1; d < CT; d <<= 1) {
    const int add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t p = blockbase[blockIdx.x] + sh[threadIdx.x] - c;
  for (int i = 0; i < CI; ++i) {
    const int64_t t = base + i;
    if (t < nnz) {
      pos[t] = p;
      p += keep_of(tags, t, match, invert) ? 1 : 0;
    }
  }
}

__global__ void scatter_kernel(const uint8_t* tags, int64_t nnz, int match, int invert, const int64_t* pos, const int32_t* idx,
                               const double* vals, int32_t* oidx, double* ovals) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
    if (keep_of(tags, t, match, invert)) {
      const int64_t q = pos[t];
      oidx[q] = idx[t];
      ovals[q] = vals[t];
    }
}

__global__ void newptr_kernel(const int64_t* ptr, int64_t nseg, int64_t nnz, const int64_t* pos, const int64_t* total, int64_t* optr) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= nseg; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = ptr[s];
    optr[s] = t < nnz ? pos[t] : *total;
  }
}

struct View {
  int64_t* ptr = nullptr;
  int32_t* idx = nullptr;
  double* vals = nullptr;
  ~View() {
    if (ptr) (void)hipFree(ptr);
    if (idx) (void)hipFree(idx);
    if (vals) (void)hipFree(vals);
  }
};

int compact_view(hipStream_t st, const int64_t* ptr, const int32_t* idx, const double* vals, int64_t nseg, int64_t nnz,
                 const uint8_t* host_tags, int match, int invert, View& out) {
  uint8_t* dtags = nullptr;
  int64_t *blockcount = nullptr, *pos = nullptr, *dtotal = nullptr;
  auto cleanup = [&](int rc) {
    if (dtags) (void)hipFree(dtags);
    if (blockcount) (void)hipFree(blockcount);
    if (pos) (void)hipFree(pos);
    if (dtotal) (void)hipFree(dtotal);
    return rc;
  };
  const int64_t nb = (nnz + CB - 1) / CB;
  int64_t total = 0;
  HIPCK(hipMalloc((void**)&out.ptr, (size_t)(nseg + 1) * 8));
  if (nnz > 0) {
    if (hipMalloc((void**)&dtags, (size_t)nnz) != hipSuccess || hipMalloc((void**)&blockcount, (size_t)nb * 8) != hipSuccess ||
        hipMalloc((void**)&pos, (size_t)nnz * 8) != hipSuccess || hipMalloc((void**)&dtotal, 8) != hipSuccess)
      return cleanup(fail(GLRM_ERR_OOM, "out of device memory while compacting %lld observations", (long long)nnz));
    if (hipMemcpyAsync(dtags, host_tags, (size_t)nnz, hipMemcpyHostToDevice, st) != hipSuccess)
      return cleanup(fail(GLRM_ERR_HIP, "tag upload failed"));
    hipLaunchKernelGGL(count_kernel, dim3((unsigned)nb), dim3(CT), 0, st, dtags, nnz, match, invert, blockcount);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, blockcount, nb, dtotal);
    hipLaunchKernelGGL(position_kernel, dim3((unsigned)nb), dim3(CT), 0, st, dtags, nnz, match, invert, blockcount, pos);
    if (hipMemcpyAsync(&total, dtotal, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return cleanup(fail(GLRM_ERR_HIP, "compaction failed: %s", hipGetErrorString(hipGetLastError())));
  }
  const size_t t1 = (size_t)(total > 0 ? total : 1);
  if (hipMalloc((void**)&out.idx, t1 * 4) != hipSuccess || hipMalloc((void**)&out.vals, t1 * 8) != hipSuccess)
    return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the subset (%lld observations)", (long long)total));
  if (nnz > 0) {
    hipLaunchKernelGGL(scatter_kernel, dim3(4096), dim3(256), 0, st, dtags, nnz, match, invert, pos, idx, vals, out.idx, out.vals);
    hipLaunchKernelGGL(newptr_kernel, dim3(1024), dim3(256), 0, st, ptr, nseg, nnz, pos, dtotal, out.ptr);
  } else {
    if (hipMemsetAsync(out.ptr, 0, (size_t)(nseg + 1) * 8, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "memset failed"));
  }
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "compaction kernels failed"));
  return cleanup(GLRM_OK);
}

} // namespace

extern "C" int glrm_hip_subset(glrm_handle* parent, const uint8_t* row_tags, const uint8_t* col_tags, int32_t match, int32_t invert,
                               glrm_handle** out) {
  if (!parent || !out) return fail(GLRM_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (parent->dense) return fail(GLRM_ERR_UNSUPPORTED, "glrm_hip_subset needs a list (not dense) parent handle");
  if (!parent->finalized) return fail(GLRM_ERR_INVALID, "the handle was created with GLRM_PROBLEM_DEFER_SETUP: call glrm_hip_finalize first");
  if ((parent->nnz_r > 0 && !row_tags) || (parent->nnz_c > 0 && !col_tags)) return fail(GLRM_ERR_INVALID, "NULL tag array");
  if (hipSetDevice(parent->device) != hipSuccess) return fail(GLRM_ERR_HIP, "cannot select device %d", parent->device);
  View rv, cv;
  int rc = compact_view(parent->stream, parent->rowptr, parent->colidx, parent->rowvals, parent->ml, parent->nnz_r, row_tags, match, invert, rv);
  if (rc) return rc;
  rc = compact_view(parent->stream, parent->colptr, parent->rowidx, parent->colvals, parent->nl, parent->nnz_c, col_tags, match, invert, cv);
  if (rc) return rc;
  glrm_problem p{};
  p.m = parent->m; p.n = parent->n; p.k = parent->k;
  p.flags = GLRM_PROBLEM_DEVICE_ARRAYS;
  p.row_begin = parent->rb; p.row_end = parent->re; p.col_begin = parent->cb; p.col_end = parent->ce;
  p.rowptr = rv.ptr; p.colidx = rv.idx; p.rowvals = rv.vals;
  p.colptr = cv.ptr; p.rowidx = cv.idx; p.colvals = cv.vals;
  p.losses = parent->losses_h.data(); p.n_losses = (int64_t)parent->losses_h.size();
  p.rx = parent->rx_h.data(); p.n_rx = (int64_t)parent->rx_h.size();
  p.ry = parent->ry_h.data(); p.n_ry = (int64_t)parent->ry_h.size();
  glrm_options o = parent->opts;
  o.device_id = parent->device;
  // The child of ONE SHARD of a sharded fit is a shard of the subset problem: like its parent it must choose its kernels from the
  // signature of the WHOLE (subset) problem, so it is created deferred and the host finalizes it with the combined signature of the
  // shards' children (glrm_hip_signature on each child, sum / max, glrm_hip_finalize) -- the protocol of the parents.  A single-shard
  // parent's child is set up here.
  const bool shard = !(parent->rb == 0 && parent->re == parent->m && parent->cb == 0 && parent->ce == parent->n);
  if (shard) p.flags |= GLRM_PROBLEM_DEFER_SETUP;
  return glrm_hip_create(out, &p, &o); // copies the compacted views; rv / cv are released on return
}
