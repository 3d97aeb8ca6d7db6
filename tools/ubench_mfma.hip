// ubench_mfma.hip -- what v_mfma_f64_16x16x4_f64 sustains on this chip when nothing else happens (the ceiling the dense path's
// roofline.frac is read against): every wave runs NCH independent accumulation chains of MFMAs on register operands.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_mfma ubench_mfma.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(e)                                                                                   \
  do {                                                                                          \
    hipError_t r_ = (e);                                                                        \
    if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } \
  } while (0)

using f64x4 = __attribute__((ext_vector_type(4))) double;

// out[1], out[2]: shader-clock and constant-100-MHz-clock ticks of block 0's first wave over the loop (the clock the SIMDs really ran at)
template <int NCH>
__global__ void __launch_bounds__(256) mfma_kernel(int iters, double* out) {
  const long long c0 = clock64(), w0 = wall_clock64();
  f64x4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = f64x4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 1.2345e-300) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[1] = (double)(clock64() - c0);
    out[2] = (double)(wall_clock64() - w0);
  }
}

// v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks per instruction (512 flop), one f64 result per lane
template <int NCH>
__global__ void __launch_bounds__(256) mfma4_kernel(int iters, double* out) {
  const long long c0 = clock64(), w0 = wall_clock64();
  double acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = 0.0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c];
  if (s == 1.2345e-300) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[1] = (double)(clock64() - c0);
    out[2] = (double)(wall_clock64() - w0);
  }
}

static void run_mfma4(int waves_per_simd, double* out) {
  constexpr int NCH = 8;
  const int iters = 10000, blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma4_kernel<NCH>), dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  double h[3];
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  const double flop = (double)blocks * 4 * iters * NCH * 512.0;
  printf("v_mfma_f64_4x4x4_4b, %d chains per wave, %d wave(s) per SIMD: %.1f TFLOP/s (%.2f ms), shader clock %.0f MHz\n", NCH, waves_per_simd,
         flop / (best * 1e-3) / 1e12, best, h[1] / h[2] * 100.0);
}

// the same with fp64 vector FMAs: NCH independent chains per lane
template <int NCH>
__global__ void __launch_bounds__(256) fma_kernel(int iters, double* out) {
  const long long c0 = clock64(), w0 = wall_clock64();
  double acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = threadIdx.x * 1e-3 + c;
  const double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9 * threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = fma(acc[c], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += acc[c];
  if (s == 1.2345e-300) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[1] = (double)(clock64() - c0);
    out[2] = (double)(wall_clock64() - w0);
  }
}

static void run_fma(int waves_per_simd, double* out) {
  constexpr int NCH = 8;
  const int iters = 40000, blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((fma_kernel<NCH>), dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  double h[3];
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  const double flop = (double)blocks * 256 * iters * NCH * 2.0;
  printf("v_fma_f64, %d chains per lane, %d wave(s) per SIMD: %.1f TFLOP/s (%.2f ms), shader clock %.0f MHz\n", NCH, waves_per_simd,
         flop / (best * 1e-3) / 1e12, best, h[1] / h[2] * 100.0);
}

template <int NCH>
static void run(int waves_per_simd, double* out) {
  const int iters = 20000 / NCH, blocks = 256 * waves_per_simd; // 4 waves per block, one block per (CU, wave slot)
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((mfma_kernel<NCH>), dim3(blocks), dim3(256), 0, 0, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  double h[3];
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  const double flop = (double)blocks * 4 * iters * NCH * 2048.0;
  printf("v_mfma_f64_16x16x4, %d chain(s) per wave, %d wave(s) per SIMD: %.1f TFLOP/s (%.2f ms), shader clock %.0f MHz\n", NCH, waves_per_simd,
         flop / (best * 1e-3) / 1e12, best, h[1] / h[2] * 100.0);
}

int main() {
  double* out;
  CK(hipMalloc(&out, 24));
  run<1>(1, out);
  run<2>(1, out);
  run<4>(1, out);
  run<1>(2, out);
  run<1>(4, out);
  run<2>(4, out);
  run<4>(4, out);
  run<4>(8, out); // two rounds of blocks: a longer run (clocks under sustained load)
  run<4>(32, out);
  run_mfma4(1, out);
  run_mfma4(4, out);
  run_mfma4(16, out);
  run_fma(1, out);
  run_fma(4, out);
  run_fma(16, out);
  return 0;
}
