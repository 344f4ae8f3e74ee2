// Microbenchmark of the conv2 forward inner loop in isolation: 12 waves per workgroup, 72 weight registers,
// A operands from an LDS patch with the kernel's addressing, 2 accumulators, 144 MFMAs per "band".
//   MODE 0: as the kernel (LDS A, 72 B regs)   1: A from a register (no LDS)   2: LDS A, single B register
//   3: LDS A with a conflict-free dense addressing (lane*4 bytes)
//   4 / 5 / 6: as 0 with an explicit operand ring 2 / 4 / 6 k-steps deep, order pinned with sched_group_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int RS = 65, PS = 9 * RS, PATCH = 32 * PS;
template <int MODE>
__global__ __launch_bounds__(768) void k(float* out, const float* w, int bands) {
  __shared__ float patch[2 * PATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave % 3, pg = wave / 3, rp = pg >> 1, ch = pg & 1, lr = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 2 * PATCH; i += 768) patch[i] = 0.001f * (i & 255);
  float wr[72];
#pragma unroll
  for (int ks = 0; ks < 72; ++ks) wr[ks] = w[(nt * 16 + lr) * 288 + ((ks & 7) * 4 + lq) * 9 + (ks >> 3)];
  __syncthreads();
  const int aoff = MODE == 3 ? lane : lq * PS + (4 * rp) * RS + 2 * (16 * ch + lr);
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float areg = (float)tid;
  float fill[4] = {areg, areg + 1.f, areg + 2.f, areg + 3.f};
  for (int b = 0; b < bands; ++b) {
    const float* ab = patch + (b & 1) * PATCH + aoff;
    if (MODE >= 4) {
      // explicit register ring: the operands of k-step ks + D are requested right after the MFMAs of k-step ks
      constexpr int D = MODE == 4 ? 2 : (MODE == 6 ? 6 : 4);
      constexpr int NV = MODE >= 7 ? 4 * (MODE - 6) : 0;          // modes 7 / 8 / 9 / 10: + 4 / 8 / 12 / 16 independent VALU FMAs per k-step
      float xa0[D], xa1[D];
      auto off = [](int ks) { const int tap = ks >> 3, ky = tap / 3, kx = tap % 3, cg = ks & 7; return cg * 4 * PS + ky * RS + kx; };
#pragma unroll
      for (int d = 0; d < D; ++d) { xa0[d] = ab[off(d)]; xa1[d] = ab[off(d) + 2 * RS]; }
#pragma unroll
      for (int ks = 0; ks < 72; ++ks) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0[ks % D], wr[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1[ks % D], wr[ks], acc1, 0, 0, 0);
        if (ks + D < 72) { xa0[ks % D] = ab[off(ks + D)]; xa1[ks % D] = ab[off(ks + D) + 2 * RS]; }
#pragma unroll
        for (int v = 0; v < NV; ++v) fill[v & 3] = __builtin_fmaf(fill[v & 3], 1.0001f, 1.f);
        if (NV) __builtin_amdgcn_sched_barrier(0);
        else {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
      const int tap = ks >> 3, ky = tap / 3, kx = tap % 3, cg = ks & 7;
      float x0, x1;
      if (MODE == 1) { x0 = areg; x1 = areg; }
      else if (MODE == 3) { x0 = ab[ks * 128]; x1 = ab[ks * 128 + 64]; }
      else { x0 = ab[cg * 4 * PS + ky * RS + kx]; x1 = ab[cg * 4 * PS + (2 + ky) * RS + kx]; }
      const float bw = MODE == 2 ? wr[0] : wr[ks];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, bw, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, bw, acc1, 0, 0, 0);
    }
  }
  const float s = fill[0] + fill[1] + fill[2] + fill[3] + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (s == 1.2345f) out[0] = s;
}
template <int MODE>
void run(const float* w, float* d, int bands) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(768), 0, 0, d, w, bands);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = 256.0 * 12 * (double)bands * 144 * 2048.0;
  printf("mode %d: %.1f us for %d bands -> %.1f TFLOP/s\n", MODE, ms * 1e3, bands, flops / (ms * 1e-3) / 1e12);
}
int main() {
  float *w, *d; (void)hipMalloc(&w, 48 * 288 * 4); (void)hipMalloc(&d, 4); (void)hipMemset(w, 0, 48 * 288 * 4);
  run<0>(w, d, 150); run<1>(w, d, 150); run<2>(w, d, 150); run<3>(w, d, 150); run<4>(w, d, 150); run<5>(w, d, 150); run<6>(w, d, 150); run<7>(w, d, 150); run<8>(w, d, 150); run<9>(w, d, 150); run<10>(w, d, 150);
  return 0;
}
