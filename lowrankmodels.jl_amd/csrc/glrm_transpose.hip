// glrm_transpose.hip -- GLRM_PROBLEM_ROWS_FROM_COLS: the row view (observed_features) derived ON THE DEVICE from the column view
// (observed_examples) for an Omega that is a sparse matrix's pattern.
//
// The reference builds both lists from `findall(!iszero, A)` for SparseMatrixCSC input (src/glrm.jl:46-48): CartesianIndices in
// column-major order pushed into per-row and per-column vectors (src/modify_glrm.jl:8-12) -- so observed_examples[j] is the column's
// rowval slice and observed_features[i] lists the SAME entries by ascending column.  A host that holds such a matrix hands over
// colptr / rowval / nzval only; the row view is one stable sort of the column-major stream by row id (entries of a row keep their column
// order), which the device does in a fraction of a second where a host needs a counting transpose of 1e9 entries (bench.py
// setup_s.create_from_host: 13.8 s in scipy) and a second 12 GB trip over PCIe.  Duplicates are kept (they appear in both views).
// hipCUB counts items in int: an Omega of more than 1.5e9 observations (C5: 5e9) is sorted in row ranges of at most that many -- the
// entries of a range are picked out of the column-major stream in order (count, offsets, scatter: a stable selection) and sorted like
// the whole (GLRM_HIP_TRANSPOSE_CHUNK lowers the bound so that the tests walk the ranged path on small patterns).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "glrm_engine.hpp"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void iota_u32_kernel(uint32_t* p, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) p[t] = (uint32_t)t;
}

// rowptr[i + 1] += 1 per entry of row i (rowptr zeroed before); an inclusive scan turns the counts into the row pointers
__global__ void row_count_kernel(const int32_t* rowidx, int64_t nnz, unsigned long long* rowptr) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&rowptr[(int64_t)rowidx[t] + 1], 1ull);
}

// Stable selection of the entries with r0 <= row < r1 out of the column-major stream.  Block b owns the span [b * span, (b + 1) * span):
// pass 1 counts, the host turns the counts into offsets, pass 2 writes (row id, global position) in stream order.
constexpr int SEL_THREADS = 256;

__global__ void __launch_bounds__(SEL_THREADS) range_count_kernel(const int32_t* rowidx, int64_t nnz, int64_t span, int32_t r0, int32_t r1, unsigned long long* blockcnt) {
  const int64_t lo = (int64_t)blockIdx.x * span, hi = lo + span < nnz ? lo + span : nnz;
  unsigned long long c = 0;
  for (int64_t t = lo + threadIdx.x; t < hi; t += SEL_THREADS) c += (rowidx[t] >= r0 && rowidx[t] < r1) ? 1ull : 0ull;
  using Reduce = hipcub::BlockReduce<unsigned long long, SEL_THREADS>;
  __shared__ typename Reduce::TempStorage tmp;
  const unsigned long long total = Reduce(tmp).Sum(c);
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SEL_THREADS) range_scatter_kernel(const int32_t* rowidx, int64_t nnz, int64_t span, int32_t r0, int32_t r1,
                                                                    const unsigned long long* blockoff, uint32_t* keys, int64_t* gpos) {
  const int64_t lo = (int64_t)blockIdx.x * span, hi = lo + span < nnz ? lo + span : nnz;
  using Scan = hipcub::BlockScan<int, SEL_THREADS>;
  __shared__ typename Scan::TempStorage tmp;
  unsigned long long base = blockoff[blockIdx.x];
  for (int64_t t0 = lo; t0 < hi; t0 += SEL_THREADS) {
    const int64_t t = t0 + threadIdx.x;
    const int32_t r = t < hi ? rowidx[t] : -1;
    const int flag = (r >= r0 && r < r1) ? 1 : 0;
    int off = 0, total = 0;
    Scan(tmp).ExclusiveSum(flag, off, total);
    if (flag) { keys[base + (unsigned long long)off] = (uint32_t)r; gpos[base + (unsigned long long)off] = t; }
    base += (unsigned long long)total;
    __syncthreads(); // tmp is reused by the next tile
  }
}

// entry t of the row view came from position pos[t] of the column view: its column = the segment of colptr that holds pos[t]
// (gpos: the global position of selected entry pos[t] when a row range was sorted; NULL when the whole stream was)
__global__ void gather_rows_kernel(const uint32_t* pos, const int64_t* gpos, int64_t nnz, const int64_t* colptr, int64_t n, const double* colvals, int32_t* colidx,
                                   double* rowvals) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = gpos ? gpos[pos[t]] : (int64_t)pos[t];
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
  HIPCK(hipMemsetAsync(h->rowptr, 0, ((size_t)m + 1) * 8, st));
  if (nnz == 0) return GLRM_OK;
  if (m + 1 > (int64_t)INT32_MAX) return fail(GLRM_ERR_UNSUPPORTED, "GLRM_PROBLEM_ROWS_FROM_COLS: %lld rows", (long long)m);
  const int64_t cap0 = env_int("GLRM_HIP_TRANSPOSE_CHUNK", 1500000000);
  const int64_t cap = cap0 > 0 && cap0 <= 1500000000 ? cap0 : 1500000000;
  const bool trace = env_int("GLRM_HIP_TRANSPOSE_TRACE", 0) != 0; // step times on stderr (every step drains the stream)
  double t_last = now_s();
  auto lap = [&](const char* what) {
    if (!trace) return;
    (void)hipStreamSynchronize(st);
    const double t = now_s();
    fprintf(stderr, "[glrm transpose] %-28s %8.3f s\n", what, t - t_last);
    t_last = t;
  };
  uint32_t *k0 = nullptr, *k1 = nullptr, *p0 = nullptr, *p1 = nullptr;
  int64_t* gpos = nullptr;
  unsigned long long* blk = nullptr;
  void* tmp = nullptr;
  int* flag = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)k0, (void*)k1, (void*)p0, (void*)p1, (void*)gpos, (void*)blk, tmp, (void*)flag}) if (p) (void)hipFree(p);
    return rc;
  };
  // 1. indices in range; 2. row pointers = inclusive scan of the per-row counts
  if (hipMalloc((void**)&flag, 4) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (GLRM_PROBLEM_ROWS_FROM_COLS)"));
  if (hipMemsetAsync(flag, 0, 4, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "memset failed"));
  hipLaunchKernelGGL(check_rowidx_kernel, dim3(4096), dim3(256), 0, st, h->rowidx, nnz, m, flag);
  int bad = 0;
  if (hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "row index check failed"));
  if (bad) return cleanup(fail(GLRM_ERR_INVALID, "rowidx holds an index outside [0, m)"));
  lap("index check");
  hipLaunchKernelGGL(row_count_kernel, dim3(8192), dim3(256), 0, st, h->rowidx, nnz, reinterpret_cast<unsigned long long*>(h->rowptr));
  size_t bytes = 0;
  if (hipcub::DeviceScan::InclusiveSum(nullptr, bytes, h->rowptr, h->rowptr, (int)(m + 1), st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "scan (size query) failed"));
  if (hipMalloc(&tmp, bytes > 0 ? bytes : 1) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (scan scratch)"));
  if (hipcub::DeviceScan::InclusiveSum(tmp, bytes, h->rowptr, h->rowptr, (int)(m + 1), st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "scan failed"));
  std::vector<int64_t> rp;
  const bool whole = nnz <= cap;
  if (!whole) {
    rp.resize((size_t)m + 1);
    if (hipMemcpyAsync(rp.data(), h->rowptr, ((size_t)m + 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "row pointer read-back failed"));
  }
  if (hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "row pointers failed"));
  (void)hipFree(tmp);
  tmp = nullptr;
  lap("row counts + scan");
  // 3. row ranges of at most `cap` entries (one row at least); the whole stream when it fits
  std::vector<int64_t> cuts{0};
  int64_t maxcnt = whole ? nnz : 0;
  if (whole) cuts.push_back(m);
  while (cuts.back() < m) {
    const int64_t r0 = cuts.back();
    int64_t r1 = std::upper_bound(rp.begin() + r0, rp.end(), rp[(size_t)r0] + cap) - rp.begin() - 1; // last r with rp[r] - rp[r0] <= cap
    if (r1 <= r0) r1 = r0 + 1;
    if (r1 > m) r1 = m;
    cuts.push_back(r1);
    maxcnt = std::max(maxcnt, rp[(size_t)r1] - rp[(size_t)r0]);
  }
  if (maxcnt > (int64_t)INT32_MAX - 1) return cleanup(fail(GLRM_ERR_UNSUPPORTED, "GLRM_PROBLEM_ROWS_FROM_COLS: a row of %lld observations", (long long)maxcnt));
  const size_t cnt1 = (size_t)(maxcnt > 0 ? maxcnt : 1);
  const int64_t span = 1 << 20;
  const int64_t nblk = (nnz + span - 1) / span;
  bool oom = hipMalloc((void**)&k1, cnt1 * 4) != hipSuccess || hipMalloc((void**)&p0, cnt1 * 4) != hipSuccess || hipMalloc((void**)&p1, cnt1 * 4) != hipSuccess;
  if (!whole) oom = oom || hipMalloc((void**)&k0, cnt1 * 4) != hipSuccess || hipMalloc((void**)&gpos, cnt1 * 8) != hipSuccess || hipMalloc((void**)&blk, (size_t)nblk * 8) != hipSuccess;
  if (oom) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (GLRM_PROBLEM_ROWS_FROM_COLS)"));
  int bits = 1;
  while (((int64_t)1 << bits) < m) ++bits;
  bytes = 0;
  if (hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)k1, k1, p0, p1, (int)cnt1, 0, bits, st) != hipSuccess)
    return cleanup(fail(GLRM_ERR_HIP, "radix sort (size query) failed"));
  if (hipMalloc(&tmp, bytes > 0 ? bytes : 1) != hipSuccess) return cleanup(fail(GLRM_ERR_OOM, "out of device memory for the row view (sort scratch)"));
  std::vector<unsigned long long> cnts((size_t)nblk);
  lap("scratch allocation");
  for (size_t c = 0; c + 1 < cuts.size(); ++c) {
    const int64_t r0 = cuts[c], r1 = cuts[c + 1];
    const int64_t at = whole ? 0 : rp[(size_t)r0], cnt = whole ? nnz : rp[(size_t)r1] - rp[(size_t)r0];
    if (cnt == 0) continue;
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(h->rowidx); // non-negative int32: the same bits
    if (!whole) {
      hipLaunchKernelGGL(range_count_kernel, dim3((unsigned)nblk), dim3(SEL_THREADS), 0, st, h->rowidx, nnz, span, (int32_t)r0, (int32_t)r1, blk);
      if (hipMemcpyAsync(cnts.data(), blk, (size_t)nblk * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return cleanup(fail(GLRM_ERR_HIP, "row-range count failed"));
      unsigned long long run = 0;
      for (auto& v : cnts) { const unsigned long long x = v; v = run; run += x; }
      if ((int64_t)run != cnt) return cleanup(fail(GLRM_ERR_HIP, "row-range count disagrees with the row pointers (%llu vs %lld)", run, (long long)cnt));
      if (hipMemcpyAsync(blk, cnts.data(), (size_t)nblk * 8, hipMemcpyHostToDevice, st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "row-range offsets failed"));
      lap("range count");
      hipLaunchKernelGGL(range_scatter_kernel, dim3((unsigned)nblk), dim3(SEL_THREADS), 0, st, h->rowidx, nnz, span, (int32_t)r0, (int32_t)r1, blk, k0, gpos);
      keys = k0;
      lap("range scatter");
    }
    hipLaunchKernelGGL(iota_u32_kernel, dim3(4096), dim3(256), 0, st, p0, cnt);
    size_t b2 = bytes;
    if (hipcub::DeviceRadixSort::SortPairs(tmp, b2, keys, k1, p0, p1, (int)cnt, 0, bits, st) != hipSuccess) // stable: a row's entries keep their column order
      return cleanup(fail(GLRM_ERR_HIP, "radix sort failed"));
    lap("sort");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(8192), dim3(256), 0, st, p1, whole ? nullptr : gpos, cnt, h->colptr, n, h->colvals, h->colidx + at, h->rowvals + at);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return cleanup(fail(GLRM_ERR_HIP, "row view kernels failed")); // cnts is reused
    lap("gather");
  }
  return cleanup(GLRM_OK);
}
