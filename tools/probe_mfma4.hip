// probe_mfma4.hip -- operand / result lane maps of v_mfma_f64_4x4x4_4b_f64, found by experiment: wave w = (la, lb) runs the instruction
// with A = 1 in lane la only and B = 1 in lane lb only and reports which lanes of D are non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
  const int la = blockIdx.x / 64, lb = blockIdx.x % 64, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[blockIdx.x * 64 + lane] = d != 0.0;
}
int main() {
  int* dout;
  hipMalloc(&dout, 4096 * 64 * 4);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dout);
  std::vector<int> h(4096 * 64);
  hipMemcpy(h.data(), dout, h.size() * 4, hipMemcpyDeviceToHost);
  // for every A lane: the B lanes it pairs with and the D lanes hit
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      int cnt = 0, first = -1;
      for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l]) { if (first < 0) first = l; ++cnt; }
      if (cnt) printf(" B%d->D%d%s", lb, first, cnt > 1 ? "+" : "");
    }
    printf("\n");
  }
  return 0;
}
