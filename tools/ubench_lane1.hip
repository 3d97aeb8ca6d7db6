// ubench_lane1.hip -- round 6 prototype: ONE LANE per segment on the LDS tiles, conflict-free.
//
// Round 2 priced a row-per-lane tiled sweep (tools/ubench_lanerow.hip: 10.8 ps per observation and pass against the product's ~8) and dropped
// it -- but that prototype read 64 unrelated PADDED rows per ds_read_b128 (51 TB/s of LDS against 118 conflict-free, profiles/r02_ubench_lanerow.txt):
// it was bound by LDS bank conflicts, and the conflict-free chunk walk found later in the same round was never applied to it.  Here:
//   * rows of the tile are UNPADDED (256 B at k = 32: every row starts at bank 0) and lane l walks the row's sixteen 16-byte chunks in the order
//     i ^ (l & 15): the 16 lanes of an LDS cycle of ds_read_b128 ({0-3,12-15,20-27}, ...) have 16 different l & 15, so they read 16 different
//     4-bank groups whatever rows they read.  Register i of the lane holds chunk i ^ p of x, g and y alike;
//   * the tile is staged by LDS-DMA (global_load_lds_dwordx4) from all waves: a plain copy of the unpadded rows;
//   * the observation stream is the SELL layout of the round-2 prototype (per (wave, tile) a run of steps, one (LDS byte offset, value) per lane,
//     padded to the longest row of the wave in that tile), read coalesced with the next step prefetched.
// No cross-lane instruction exists in the pass.  Variants: waves per workgroup 8 / 12 / 16 (256 / 168 / 128 VGPRs), gradient kept / re-read.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/ubench_lane1 tools/ubench_lane1.hip ;  /tmp/ubench_lane1 [rows_log2=19]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                                   \
  do {                                                                                          \
    hipError_t r_ = (e);                                                                        \
    if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } \
  } while (0)

constexpr int K = 32;
constexpr int TILE = 576; // 576 x 256 B = 144 KB

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__host__ __device__ inline bool observed(uint64_t row, uint64_t col, uint32_t thresh) { return (uint32_t)(mix64(row * 1000003ull + col) >> 32) < thresh; }
__host__ __device__ inline double value_of(uint64_t row, uint64_t col) { return (double)(mix64(row * 7919ull + col * 104729ull + 17) >> 11) * (1.0 / 9007199254740992.0) - 0.5; }

__global__ void count_kernel(int m, int n, int ntiles, uint32_t thresh, int32_t* cnt) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)m * ntiles) return;
  const int row = (int)(id / ntiles), t = (int)(id % ntiles);
  int c = 0;
  for (int j = t * TILE; j < min(n, (t + 1) * TILE); ++j) c += observed(row, j, thresh);
  cnt[id] = c;
}

__global__ void fill_kernel(int m, int n, int ntiles, uint32_t thresh, const int64_t* sptr, int32_t* sidx, double* sval) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)m * ntiles) return;
  const int row = (int)(id / ntiles), t = (int)(id % ntiles);
  const int slice = row / 64, r = row % 64;
  int64_t s = sptr[(int64_t)slice * (ntiles + 1) + t];
  const int64_t s1 = sptr[(int64_t)slice * (ntiles + 1) + t + 1];
  for (int j = t * TILE; j < min(n, (t + 1) * TILE); ++j)
    if (observed(row, j, thresh)) {
      sidx[s * 64 + r] = (j - t * TILE) * K * 8; // byte offset of the staged vector
      sval[s * 64 + r] = value_of(row, j);
      ++s;
    }
  for (; s < s1; ++s) {
    sidx[s * 64 + r] = -1;
    sval[s * 64 + r] = 0.0;
  }
}

// GRAD: objective and gradient at x; else objective only.  REREAD: the gradient update reads y from LDS a second time instead of keeping it.
template <int NW, int MINB, bool GRAD, bool REREAD, int U>
__global__ void __launch_bounds__(NW * 64, MINB) lane_pass(const int32_t* __restrict__ sidx, const double* __restrict__ sval, const int64_t* __restrict__ sptr,
                                                           const double* __restrict__ X, const double* __restrict__ Y, double* __restrict__ Gout,
                                                           double* __restrict__ obj, int m, int n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int C = K / 2; // 16-byte chunks per vector
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane & 15;
  const int64_t slice = (int64_t)blockIdx.x * NW + wave;
  const int64_t row = slice * 64 + lane;
  double2 x[C], g[C];
#pragma unroll
  for (int i = 0; i < C; ++i) {
    x[i] = ((const double2*)(X + row * K))[i ^ p];
    g[i] = make_double2(0.0, 0.0);
  }
  double o = 0.0;
  const int64_t* sp = sptr + slice * (ntiles + 1);
  const int pb = p * 16;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    { // LDS-DMA: the tile image is a plain copy of rows [t TILE, ...)
      const int rows = min(TILE, n - t * TILE), total = rows * C;
      const char* src0 = (const char*)(Y + (int64_t)t * TILE * K);
      for (int base = wave * 64; base < total; base += NW * 64) {
        const int c = base + lane;
        if (c < total)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (size_t)c * 16),
                                           (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int64_t s = sp[t];
    const int64_t s1 = sp[t + 1];
    if (s >= s1) continue;
    // steps in blocks of U: the (offset, value) pairs of the NEXT block are in flight while this one is consumed (two waves per SIMD cannot
    // hide an HBM round trip per step: the one-step prefetch of the first version ran at ~1500 cycles per step)
    int32_t off[U], noff[U];
    double a[U], na[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t q = s + u < s1 ? s + u : s1 - 1;
      off[u] = sidx[q * 64 + lane];
      a[u] = sval[q * 64 + lane];
      if (s + u >= s1) off[u] = -1;
    }
    for (; s < s1; s += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t q = s + U + u < s1 ? s + U + u : s1 - 1;
        noff[u] = sidx[q * 64 + lane];
        na[u] = sval[q * 64 + lane];
        if (s + U + u >= s1) noff[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (off[u] >= 0) {
          const char* yp = lds + off[u];
          double2 y[C];
#pragma unroll
          for (int i = 0; i < C; ++i) y[i] = *(const double2*)(yp + ((i * 16) ^ pb));
          double u0 = 0.0, u1 = 0.0;
#pragma unroll
          for (int i = 0; i < C; ++i) {
            u0 = fma(x[i].x, y[i].x, u0);
            u1 = fma(x[i].y, y[i].y, u1);
          }
          const double res = (u0 + u1) - a[u];
          o = fma(res, res, o);
          if (GRAD) {
            const double d = 2.0 * res;
            if (REREAD) {
              asm volatile("" ::: "memory");
#pragma unroll
              for (int i = 0; i < C; ++i) y[i] = *(const double2*)(yp + ((i * 16) ^ pb));
            }
#pragma unroll
            for (int i = 0; i < C; ++i) {
              g[i].x = fma(d, y[i].x, g[i].x);
              g[i].y = fma(d, y[i].y, g[i].y);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { off[u] = noff[u]; a[u] = na[u]; }
    }
  }
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < C; ++i) ((double2*)(Gout + row * K))[i ^ p] = g[i];
  }
  obj[row] = o;
}

// v3: branch-free and software-pipelined across observations.  An idle lane (offset -1) reads row 0 and its terms are zeroed, so the body is
// straight-line code: the ds_reads of observation u + 1 are in flight while the fma chains of observation u run (trial pass: two copies of y in
// registers; gradient pass: y is streamed twice -- dot product, then axpy -- with the next observation's first stream issued before this one's
// axpy).  (offset, value) pairs run three blocks of U deep: the block in use, the next one (arrived), the one after (in flight).
template <int NW, bool GRAD, int U>
__global__ void __launch_bounds__(NW * 64, 1) lane_pass3(const int32_t* __restrict__ sidx, const double* __restrict__ sval, const int64_t* __restrict__ sptr,
                                                         const double* __restrict__ X, const double* __restrict__ Y, double* __restrict__ Gout,
                                                         double* __restrict__ obj, int m, int n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int C = K / 2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane & 15;
  const int64_t slice = (int64_t)blockIdx.x * NW + wave;
  const int64_t row = slice * 64 + lane;
  double2 x[C], g[C];
  int q[C]; // byte offset of the chunk register i reads: (i ^ p) * 16
#pragma unroll
  for (int i = 0; i < C; ++i) {
    x[i] = ((const double2*)(X + row * K))[i ^ p];
    g[i] = make_double2(0.0, 0.0);
    q[i] = (i ^ p) * 16;
  }
  double o = 0.0;
  const int64_t* sp = sptr + slice * (ntiles + 1);
  auto rd = [&](double2 (&y)[C], int off) {
#pragma unroll
    for (int i = 0; i < C; ++i) y[i] = *(const double2*)(lds + off + q[i]);
  };
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    {
      const int rows = min(TILE, n - t * TILE), total = rows * C;
      const char* src0 = (const char*)(Y + (int64_t)t * TILE * K);
      for (int base = wave * 64; base < total; base += NW * 64) {
        const int c = base + lane;
        if (c < total)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src0 + (size_t)c * 16),
                                           (__attribute__((address_space(3))) void*)(lds + base * 16), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int64_t s = sp[t];
    const int64_t s1 = sp[t + 1];
    if (s >= s1) continue;
    int32_t off[U], off1[U], off2[U];
    double a[U], a1[U], a2[U];
    auto ld = [&](int32_t (&of)[U], double (&av)[U], int64_t s0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t qq = s0 + u < s1 ? s0 + u : s1 - 1;
        of[u] = sidx[qq * 64 + lane];
        av[u] = sval[qq * 64 + lane];
        if (s0 + u >= s1) of[u] = -1;
      }
    };
    ld(off, a, s);
    ld(off1, a1, s + U);
    double2 ya[C], yb[C];
    rd(ya, off[0] < 0 ? 0 : off[0]);
    for (; s < s1; s += U) {
      ld(off2, a2, s + 2 * U);
#pragma unroll
      for (int u = 0; u < U; u += 2) {
        // observation u on ya, u + 1 on yb
        rd(yb, off[u + 1] < 0 ? 0 : off[u + 1]);
        {
          double u0 = 0.0, u1 = 0.0;
#pragma unroll
          for (int i = 0; i < C; ++i) { u0 = fma(x[i].x, ya[i].x, u0); u1 = fma(x[i].y, ya[i].y, u1); }
          double res = (u0 + u1) - a[u];
          res = off[u] < 0 ? 0.0 : res;
          o = fma(res, res, o);
          if (GRAD) {
            const double d = 2.0 * res;
            asm volatile("" ::: "memory");
            rd(ya, off[u] < 0 ? 0 : off[u]);
#pragma unroll
            for (int i = 0; i < C; ++i) { g[i].x = fma(d, ya[i].x, g[i].x); g[i].y = fma(d, ya[i].y, g[i].y); }
          }
        }
        const int nxt = u + 2 < U ? off[u + 2 < U ? u + 2 : 0] : off1[0];
        rd(ya, nxt < 0 ? 0 : nxt);
        {
          double u0 = 0.0, u1 = 0.0;
#pragma unroll
          for (int i = 0; i < C; ++i) { u0 = fma(x[i].x, yb[i].x, u0); u1 = fma(x[i].y, yb[i].y, u1); }
          double res = (u0 + u1) - a[u + 1];
          res = off[u + 1] < 0 ? 0.0 : res;
          o = fma(res, res, o);
          if (GRAD) {
            const double d = 2.0 * res;
            asm volatile("" ::: "memory");
            rd(yb, off[u + 1] < 0 ? 0 : off[u + 1]);
#pragma unroll
            for (int i = 0; i < C; ++i) { g[i].x = fma(d, yb[i].x, g[i].x); g[i].y = fma(d, yb[i].y, g[i].y); }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { off[u] = off1[u]; a[u] = a1[u]; off1[u] = off2[u]; a1[u] = a2[u]; }
    }
  }
  if (GRAD) {
#pragma unroll
    for (int i = 0; i < C; ++i) ((double2*)(Gout + row * K))[i ^ p] = g[i];
  }
  obj[row] = o;
}

template <int NW, int MINB, bool REREAD, int U>
static void run(int m, int n, int ntiles, uint32_t thresh, const int64_t* dsptr, const int32_t* sidx, const double* sval, const double* dX, const double* dY,
                const std::vector<double>& hX, const std::vector<double>& hY, int64_t nobs, int64_t steps) {
  double *dG, *dobj;
  CK(hipMalloc(&dG, (size_t)m * K * 8));
  CK(hipMalloc(&dobj, (size_t)m * 8));
  const size_t lds = (size_t)TILE * K * 8;
  auto kg = MINB == 3 ? lane_pass3<NW, true, U> : lane_pass<NW, MINB == 3 ? 1 : MINB, true, REREAD, U>;
  auto kt = MINB == 3 ? lane_pass3<NW, false, U> : lane_pass<NW, MINB == 3 ? 1 : MINB, false, REREAD, U>;
  CK(hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = (int)(m / 64 / NW);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best[2] = {1e30f, 1e30f};
  for (int rep = 0; rep < 4; ++rep)
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipEventRecord(e0));
      if (pass == 0) hipLaunchKernelGGL(kg, dim3(blocks), dim3(NW * 64), lds, 0, sidx, sval, dsptr, dX, dY, dG, dobj, m, n, ntiles);
      else hipLaunchKernelGGL(kt, dim3(blocks), dim3(NW * 64), lds, 0, sidx, sval, dsptr, dX, dY, dG, dobj, m, n, ntiles);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best[pass] = std::min(best[pass], ms);
    }
  CK(hipGetLastError());
  hipLaunchKernelGGL(kg, dim3(blocks), dim3(NW * 64), lds, 0, sidx, sval, dsptr, dX, dY, dG, dobj, m, n, ntiles);
  std::vector<double> hG((size_t)m * K), hobj(m);
  CK(hipMemcpy(hG.data(), dG, hG.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hobj.data(), dobj, hobj.size() * 8, hipMemcpyDeviceToHost));
  double worst = 0.0;
  for (int64_t row : {(int64_t)0, (int64_t)1, (int64_t)m / 2 + 3, (int64_t)m - 1}) {
    double o = 0.0, g[K] = {0};
    for (int j = 0; j < n; ++j)
      if (observed(row, j, thresh)) {
        double u = 0.0;
        for (int c = 0; c < K; ++c) u += hX[row * K + c] * hY[(size_t)j * K + c];
        const double res = u - value_of(row, j);
        o += res * res;
        for (int c = 0; c < K; ++c) g[c] += 2.0 * res * hY[(size_t)j * K + c];
      }
    worst = std::max(worst, std::fabs(o - hobj[row]) / std::fabs(o));
    for (int c = 0; c < K; ++c) worst = std::max(worst, std::fabs(g[c] - hG[row * K + c]) / (1e-9 + std::fabs(g[c])));
  }
  const double c2 = 5e8 / (double)nobs;
  printf("lane-per-row, conflict-free walk, U=%d waves=%2d (min blocks %d) rows/wg=%4d %s: grad pass %.3f ms, trial pass %.3f ms  (%.2f / %.2f ps per observation); SELL padding x%.2f; "
         "max rel err %.1e\n", U, NW, MINB, NW * 64, REREAD ? "y re-read for the gradient" : "y kept for the gradient   ", best[0], best[1], best[0] * 1e9 / nobs, best[1] * 1e9 / nobs,
         (double)steps * 64 / (double)nobs, worst);
  printf("      scaled to C2 (5e8 observations): grad %.2f ms + 1.04 trial passes %.2f ms = %.2f ms per X half-step (product, round 5: 7.0 ms)\n", best[0] * c2, 1.04 * best[1] * c2,
         (best[0] + 1.04 * best[1]) * c2);
  CK(hipFree(dG));
  CK(hipFree(dobj));
}

// what the access pattern alone reaches: every lane reads whole 256-byte rows (sixteen ds_read_b128) at pseudo-random rows of a resident tile;
// ROT: chunk order i ^ (lane & 15) on unpadded rows, else natural order (16-way conflicts on unpadded rows)
template <bool ROT, int MASKPCT>
__global__ void __launch_bounds__(512) lds_rows1(const double* __restrict__ Y, int trips, double* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int q = threadIdx.x; q < TILE * 16; q += 512) ((int4*)lds)[q] = ((const int4*)Y)[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, p = ROT ? (lane & 15) : 0;
  uint32_t h = (uint32_t)mix64((uint64_t)blockIdx.x * 512 + threadIdx.x);
  unsigned acc = 0;
  for (int t = 0; t < trips; ++t) {
    h = h * 1664525u + 1013904223u;
    const int r = (h >> 8) % TILE;
    if (MASKPCT > 0 && (int)((h >> 20) % 100) < MASKPCT) continue; // idle lanes (SELL padding)
    const char* yp = lds + r * 256;
    int4 y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) y[i] = *(const int4*)(yp + ((i ^ p) * 16));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= (unsigned)y[i].x ^ (unsigned)y[i].y ^ (unsigned)y[i].z ^ (unsigned)y[i].w;
  }
  if (acc == 0x12345678u) out[0] = 1.0;
}

template <bool ROT, int MASKPCT>
static void run_lds1(const double* dY, double* dout) {
  const size_t lds = (size_t)TILE * 256;
  CK(hipFuncSetAttribute((const void*)lds_rows1<ROT, MASKPCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int trips = 4000, blocks = 1024;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((lds_rows1<ROT, MASKPCT>), dim3(blocks), dim3(512), lds, 0, dY, trips, dout);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  CK(hipGetLastError());
  const double bytes = (double)blocks * 512 * (double)trips * 256 * (100 - MASKPCT) / 100.0;
  printf("LDS random row reads, ONE lane per 256-byte row, 8 waves per CU, %s, %d %% idle lanes: %6.1f TB/s of useful bytes (%.3f ms)\n",
         ROT ? "unpadded rows, chunk walk i ^ (lane & 15)" : "unpadded rows, natural chunk order          ", MASKPCT, bytes / (best * 1e-3) / 1e12, best);
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 393216; // divisible by 64 x 8, 64 x 12 and 64 x 16
  const int n = 10000, ntiles = (n + TILE - 1) / TILE;
  const uint32_t thresh = (uint32_t)(0.05 * 4294967296.0);
  int32_t* dcnt;
  const int64_t nt = (int64_t)m * ntiles;
  CK(hipMalloc(&dcnt, nt * 4));
  hipLaunchKernelGGL(count_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, m, n, ntiles, thresh, dcnt);
  std::vector<int32_t> cnt(nt);
  CK(hipMemcpy(cnt.data(), dcnt, nt * 4, hipMemcpyDeviceToHost));
  int64_t nobs = 0;
  for (int32_t c : cnt) nobs += c;
  std::vector<double> hX((size_t)m * K), hY((size_t)n * K);
  for (size_t i = 0; i < hX.size(); ++i) hX[i] = (double)(mix64(i + 99) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  for (size_t i = 0; i < hY.size(); ++i) hY[i] = (double)(mix64(i + 777777) >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  double *dX, *dY;
  CK(hipMalloc(&dX, hX.size() * 8));
  CK(hipMalloc(&dY, hY.size() * 8 + 4096));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dY, hY.data(), hY.size() * 8, hipMemcpyHostToDevice));
  printf("rows %d x columns %d, k = %d, %lld observations (%.1f per row), tile %d vectors (%d tiles)\n", m, n, K, (long long)nobs, (double)nobs / m, TILE, ntiles);
  {
    double* dout;
    CK(hipMalloc(&dout, 8));
    run_lds1<true, 0>(dY, dout);
    run_lds1<false, 0>(dY, dout);
    run_lds1<true, 30>(dY, dout);
  }
  const int64_t nslice = m / 64;
  std::vector<int64_t> sptr((size_t)nslice * (ntiles + 1));
  int64_t steps = 0;
  for (int64_t sl = 0; sl < nslice; ++sl)
    for (int t = 0; t <= ntiles; ++t) {
      sptr[sl * (ntiles + 1) + t] = steps;
      if (t == ntiles) break;
      int mx = 0;
      for (int r = 0; r < 64; ++r) mx = std::max(mx, cnt[(size_t)(sl * 64 + r) * ntiles + t]);
      steps += mx;
    }
  int64_t* dsptr;
  int32_t* sidx;
  double* sval;
  CK(hipMalloc(&dsptr, sptr.size() * 8));
  CK(hipMemcpy(dsptr, sptr.data(), sptr.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&sidx, (size_t)steps * 64 * 4));
  CK(hipMalloc(&sval, (size_t)steps * 64 * 8));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, m, n, ntiles, thresh, dsptr, sidx, sval);
  CK(hipDeviceSynchronize());
  run<8, 1, false, 4>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<8, 1, false, 2>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<8, 1, false, 1>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<12, 1, false, 4>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<12, 1, false, 2>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<16, 1, false, 2>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  run<16, 1, true, 2>(m, n, ntiles, thresh, dsptr, sidx, sval, dX, dY, hX, hY, nobs, steps);
  return 0;
}
