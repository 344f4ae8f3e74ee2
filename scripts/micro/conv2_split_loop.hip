// Probe (extras only): the conv2 forward inner loop on the bf16 matrix pipe with every fp32 operand split into bf16 pieces
// (scripts/micro/split_bf16_error.py has the error study: hi/mid/lo x 6 products is as exact as the fp32 MFMA).
// Same work split as the shipped kernel: a wave owns 16 output channels and two 16-position M-tiles of a band; K = 9 taps x 32 input
// channels = nine K = 32 blocks of v_mfma_f32_16x16x32_bf16 (lane l: A[row l & 15][k = 8 (l >> 4) + j], j < 8).  The weights'
// pieces live in registers (9 taps x 3 pieces x 4 VGPRs = 108), the activations' pieces in LDS as [row][col][piece][32 ci] bf16
// (one ds_read_b128 per piece, tap and tile).  FLOPs are counted as the fp32 convolution's (2 x 16 x 16 x 288 per tile and wave).
//   MODE 0: 6 products (hh, hm, mh, hl, lh, mm), A from LDS   1: the same with A from registers (no LDS)   2: 3 products (hh, hm, mh)
//   3: as 0 with the A reads of tap t + 1 issued behind the MFMAs of tap t (register ring)
// hipcc --offload-arch=gfx950 -O3 conv2_split_loop.hip -o conv2_split_loop && ./conv2_split_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifndef POSB
#define POSB 208                   // bytes per patch position: 3 pieces x 64 B + padding
#endif
constexpr int COLS = 65, ROWS = 9, NW = 12;
constexpr int PATCH_BYTES = ROWS * COLS * POSB;

__device__ __forceinline__ f32x4_t mfma_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(NW * 64) void k(float* out, const unsigned* w, int bands) {
  extern __shared__ __attribute__((aligned(16))) unsigned char patch[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = wave % 3, pg = wave / 3, rp = pg >> 1, ch = pg & 1, lr = lane & 15, lq = lane >> 4;
  for (int i = tid; i < PATCH_BYTES / 4; i += NW * 64) reinterpret_cast<unsigned*>(patch)[i] = 0x3c003c00u + (i & 255);
  bf16x8_t wr[9][3];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(w + ((((nt * 16 + lr) * 9 + t) * 3 + p) * 4 + lq) * 4);
      wr[t][p] = __builtin_bit_cast(bf16x8_t, v);
    }
  __syncthreads();
  // position of tile row 0 / 1, tap (ky, kx): patch row ky + 4 rp (+ 2 for the second tile), column 2 (16 ch + lr) + kx
  const int base = ((4 * rp) * COLS + 2 * (16 * ch + lr)) * POSB + lq * 16;
  auto aoff = [&](int tap, int row2, int piece) { return base + ((tap / 3 + row2) * COLS + tap % 3) * POSB + piece * 64; };
  auto aread = [&](int tap, int row2, int piece) {
    if (MODE == 1) { u32x4_t v = {0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + tap}; return __builtin_bit_cast(bf16x8_t, v); }
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(patch + aoff(tap, row2, piece)));
  };
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  constexpr int NP = MODE == 2 ? 2 : 3;             // pieces of A that are read
  for (int b = 0; b < bands; ++b) {
    bf16x8_t a0[2][3], a1[2][3];
    if (MODE == 3) {
#pragma unroll
      for (int p = 0; p < NP; ++p) { a0[0][p] = aread(0, 0, p); a1[0][p] = aread(0, 2, p); }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int cur = MODE == 3 ? t & 1 : 0;
      if (MODE != 3) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { a0[0][p] = aread(t, 0, p); a1[0][p] = aread(t, 2, p); }
      }
      // hh, hm, mh [, hl, lh, mm]: the small products first would be the accurate order; the pipe does not care
      acc0 = mfma_bf16(a0[cur][0], wr[t][0], acc0);  acc1 = mfma_bf16(a1[cur][0], wr[t][0], acc1);
      acc0 = mfma_bf16(a0[cur][0], wr[t][1], acc0);  acc1 = mfma_bf16(a1[cur][0], wr[t][1], acc1);
      acc0 = mfma_bf16(a0[cur][1], wr[t][0], acc0);  acc1 = mfma_bf16(a1[cur][1], wr[t][0], acc1);
      if (MODE == 3 && t + 1 < 9) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { a0[cur ^ 1][p] = aread(t + 1, 0, p); a1[cur ^ 1][p] = aread(t + 1, 2, p); }
      }
      if (MODE != 2) {
        acc0 = mfma_bf16(a0[cur][0], wr[t][2], acc0);  acc1 = mfma_bf16(a1[cur][0], wr[t][2], acc1);
        acc0 = mfma_bf16(a0[cur][2], wr[t][0], acc0);  acc1 = mfma_bf16(a1[cur][2], wr[t][0], acc1);
        acc0 = mfma_bf16(a0[cur][1], wr[t][1], acc0);  acc1 = mfma_bf16(a1[cur][1], wr[t][1], acc1);
      }
      if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (s == 1.2345f) out[0] = s;
}
template <int MODE>
void run(const unsigned* w, float* d, int bands) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, PATCH_BYTES);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(NW * 64), PATCH_BYTES, 0, d, w, bands);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = 256.0 * NW * (double)bands * 144 * 2048.0;          // the fp32 convolution's FLOPs (144 fp32 MFMAs of 2048 per band and wave)
  const int mf = (MODE == 2 ? 3 : 6) * 9 * 2;
  printf("mode %d (POSB %d): %.1f us for %d bands, %d bf16 MFMAs per band and wave -> %.1f fp32-equivalent TFLOP/s (%s)\n", MODE, POSB, ms * 1e3, bands, mf,
         flops / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}
int main() {
  unsigned* w; float* d; (void)hipMalloc(&w, 48 * 9 * 3 * 4 * 4 * 4); (void)hipMalloc(&d, 4); (void)hipMemset(w, 0x3c, 48 * 9 * 3 * 4 * 4 * 4);
  run<0>(w, d, 150); run<1>(w, d, 150); run<2>(w, d, 150); run<3>(w, d, 150);
  return 0;
}
