// ubench_gather.hip -- what a k-vector gather sweep can reach on MI355X, by where the gathered table lives.
//
// Models the inner loop of csrc/glrm_hip.hip::sweep_pass: a group of G lanes fetches one ROW-byte vector (16 B per lane and load) at a
// pseudo-random row of a table and folds it into registers; U gathers are in flight per group.  The table window that a workgroup
// draws its rows from is what varies:
//   window = whole table (5 GB: HBM; 51 MB: Infinity Cache), or a per-XCD window of W bytes (block b draws from window b % 8), the
//   access pattern of an L2-blocked sweep.
// Prints GB/s of gathered bytes per configuration.   hipcc --offload-arch=gfx950 -O3 -o ubench_gather ubench_gather.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                                   \
  do {                                                                                          \
    hipError_t r_ = (e);                                                                        \
    if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } \
  } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// ROWB bytes per gathered vector, G = ROWB/64... lanes per vector with R/2 16-byte loads each: G * (R/2) * 16 = ROWB
template <int G, int R, int U>
__global__ void __launch_bounds__(256) gather_kernel(const double2* __restrict__ table, uint64_t rows_total, uint64_t win_rows, int per_xcd,
                                                     int trips, double* out) {
  const int lane = threadIdx.x & 63, j = lane % G;
  const uint64_t gid = ((uint64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const uint64_t base = per_xcd ? (uint64_t)(blockIdx.x % 8) * win_rows : 0;
  const uint64_t span = per_xcd ? win_rows : rows_total;
  double2 acc[R / 2];
#pragma unroll
  for (int i = 0; i < R / 2; ++i) acc[i] = make_double2(0.0, 0.0);
  uint64_t h = mix64(gid * 0x9E37ull + 12345);
  for (int t = 0; t < trips; ++t) {
    double2 y[U][R / 2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      h = mix64(h);
      const uint64_t row = base + h % span;
      const double2* p = table + row * (G * R / 2) + j;
#pragma unroll
      for (int i = 0; i < R / 2; ++i) y[u][i] = p[i * G];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < R / 2; ++i) {
        acc[i].x += y[u][i].x;
        acc[i].y += y[u][i].y;
      }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < R / 2; ++i) s += acc[i].x + acc[i].y;
  if (s == 1.2345e-300) out[0] = s;
}

template <int G, int R, int U>
static void run(const char* name, const double2* table, uint64_t rows_total, uint64_t win_bytes, int per_xcd, int blocks, int trips) {
  const uint64_t rowb = (uint64_t)G * R * 8;
  const uint64_t win_rows = win_bytes / rowb;
  double* out;
  CK(hipMalloc(&out, 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<G, R, U>), dim3(blocks), dim3(256), 0, 0, table, rows_total, win_rows, per_xcd, trips, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)blocks * 256 / G * trips * U * rowb;
    if (rep == 2) printf("%-44s G=%2d R=%d U=%d blocks=%6d  %8.3f ms  %9.1f GB/s\n", name, G, R, U, blocks, ms, bytes / ms / 1e6);
  }
  CK(hipFree(out));
}

int main(int argc, char** argv) {
  const uint64_t big = 5ull << 30;  // X at C4: 5.12 GB
  double2* table;
  CK(hipMalloc(&table, big));
  CK(hipMemset(table, 0, big));
  const int blocks = 256 * 8 * 4;  // 8 blocks per CU worth of work, x4 rounds
  // k = 64: 512-byte vectors, 8 lanes x 4 loads
  run<8, 8, 2>("HBM: 5 GB table, 512 B rows", table, big / 512, 0, 0, blocks, 64);
  run<8, 8, 4>("HBM: 5 GB table, 512 B rows", table, big / 512, 0, 0, blocks, 32);
  run<8, 8, 2>("Infinity Cache: 51 MB table, 512 B rows", table, (51ull << 20) / 512, 0, 0, blocks, 64);
  run<8, 8, 4>("Infinity Cache: 51 MB table, 512 B rows", table, (51ull << 20) / 512, 0, 0, blocks, 32);
  run<8, 8, 2>("Infinity Cache: 160 MB table, 512 B rows", table, (160ull << 20) / 512, 0, 0, blocks, 64);
  for (uint64_t w : {512ull << 10, 1ull << 20, 2ull << 20, 3ull << 20, 4ull << 20, 8ull << 20}) {
    char nm[96];
    snprintf(nm, sizeof nm, "L2: per-XCD window of %4llu KB, 512 B rows", (unsigned long long)(w >> 10));
    run<8, 8, 2>(nm, table, big / 512, w, 1, blocks, 64);
    run<8, 8, 4>(nm, table, big / 512, w, 1, blocks, 32);
  }
  // k = 32: 256-byte vectors, 4 lanes x 4 loads
  run<4, 8, 2>("HBM: 5 GB table, 256 B rows", table, big / 256, 0, 0, blocks, 64);
  run<4, 8, 2>("Infinity Cache: 51 MB table, 256 B rows", table, (51ull << 20) / 256, 0, 0, blocks, 64);
  run<4, 8, 2>("L2: per-XCD window of 2048 KB, 256 B rows", table, big / 256, 2ull << 20, 1, blocks, 64);
  CK(hipFree(table));
  return 0;
}
