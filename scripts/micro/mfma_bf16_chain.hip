// Issue rate of v_mfma_f32_16x16x32_bf16 against the dependency distance of its accumulator chains (1 = every MFMA waits for the one
// before it, D = D independent accumulators in rotation) and the waves per SIMD.  Cycles per MFMA and SIMD.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_bf16_chain.hip -o /tmp/chain && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int D>
__global__ void chain(float* out, long long* cyc, int iters) {
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x + 2 * j)); }
  f32x4_t acc[D];
  for (int d = 0; d < D; ++d) acc[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 24; ++u) acc[u % D] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[u % D], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += acc[d][0] + acc[d][1] + acc[d][2] + acc[d][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int D>
static void run(int waves_per_simd) {
  float* out; long long* cyc; long long h;
  const int threads = 256 * waves_per_simd, iters = 2000;
  hipMalloc(&out, 4 * threads); hipMalloc(&cyc, 8);
  hipLaunchKernelGGL(chain<D>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(chain<D>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("  distance %d, %d wave(s) per SIMD: %.1f cycles per MFMA and SIMD\n", D, waves_per_simd, (double)h / (iters * 24.0 * waves_per_simd));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 3; ++w) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<6>(w); }
  return 0;
}
