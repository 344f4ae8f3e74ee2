// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate with W waves per SIMD and A accumulators per wave,
// operands in registers (no memory).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
  f32x4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345f) out[0] = s;
}
template <int NACC>
void run(int threads, int iters) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * (double)iters * 16 * NACC * 2048.0;
  printf("waves/SIMD %d  acc %d : %.1f us  %.1f TFLOP/s\n", threads / 256, NACC, ms * 1e3, flops / (ms * 1e-3) / 1e12);
  hipFree(d);
}
int main() {
  for (int threads : {256, 512, 768, 1024}) { run<1>(threads, 2000); run<2>(threads, 1000); run<4>(threads, 500); }
  return 0;
}
