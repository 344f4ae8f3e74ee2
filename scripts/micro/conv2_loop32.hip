// Microbenchmark, round 5: the conv2 forward inner loop on v_mfma_f32_32x32x2_f32 next to the shipped
// v_mfma_f32_16x16x4_f32 loop of scripts/micro/conv2_loop.hip (mode 5 there = REF below: 12 waves, 3 per SIMD, operand ring).
// A band = 4 conv2 rows x 32 columns x 48 output channels x K = 288 = 2 * 128 * 48 * 288 FLOP, whatever the tiling.
//   REF  : 12 waves x 2 (16 x 16) tiles x 72 k-steps of 16x16x4, channel-innermost patch [row 9][col 65][ci 34], ds_read_b64 ring
//   M32  : 8 waves (2 per SIMD): waves 0-3 = one conv2 row (32 positions) x channels 0-31 on 32x32x2 (144 MFMAs, 144 weight
//          registers, ds_read2_b32 per two k-steps), waves 4-7 = channels 32-47 on 16x16x4 (2 rows x 2 column halves = 4 tiles of 16
//          positions, 72 k-steps: 288 MFMAs... see below), patch [row 9][col 65][ci 33]
//   M32x : all 8 waves on 32x32x2 with N = 64 (what the instruction gives when the channel count fits; FLOPs counted as executed)
// Each mode also runs with F independent VALU FMAs per 4096 FLOP of MFMA work (the shipped kernel carries ~1.3 VALU + 0.5 LDS per
// 16x16x4 MFMA = ~2.7 VALU per 4096 FLOP beside the loop's own reads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int RS = 65;

template <int F>
__device__ __forceinline__ void filler(float (&fill)[4]) {
#pragma unroll
  for (int v = 0; v < F; ++v) fill[v & 3] = __builtin_fmaf(fill[v & 3], 1.0001f, 1.f);
}

// ---- REF: the shipped structure --------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(768) void k_ref(float* out, const float* w, int bands) {
  constexpr int CS = 34, CROW = RS * CS, CPATCH = 9 * CROW;
  __shared__ __attribute__((aligned(16))) float patch[2 * CPATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % 3, pg = wave / 3, rp = pg >> 1, ch = pg & 1, lr = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 2 * CPATCH; i += 768) patch[i] = 0.001f * (i & 255);
  float wr[72];
#pragma unroll
  for (int ks = 0; ks < 72; ++ks) wr[ks] = w[(nt * 16 + lr) * 288 + (8 * ((ks & 7) >> 1) + 2 * lq + (ks & 1)) * 9 + (ks >> 3)];
  __syncthreads();
  const int aoff = ((4 * rp) * RS + 2 * (16 * ch + lr)) * CS + 2 * lq;
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float fill[4] = {(float)tid, tid + 1.f, tid + 2.f, tid + 3.f};
  for (int b = 0; b < bands; ++b) {
    const float* ab = patch + (b & 1) * CPATCH + aoff;
    constexpr int RDP = 2;
    auto aread = [&](int p, int row2) {
      const int tap = p >> 2, ky = tap / 3, kx = tap % 3, m = p & 3;
      return *reinterpret_cast<const f32x2_t*>(ab + (ky + row2) * CROW + kx * CS + 8 * m);
    };
    f32x2_t xa0[RDP], xa1[RDP];
#pragma unroll
    for (int d = 0; d < RDP; ++d) { xa0[d] = aread(d, 0); xa1[d] = aread(d, 2); }
#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0[(ks >> 1) % RDP][ks & 1], wr[ks], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1[(ks >> 1) % RDP][ks & 1], wr[ks], acc1, 0, 0, 0);
      if ((ks & 1) && (ks >> 1) + RDP < 36) { xa0[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 0); xa1[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 2); }
      filler<F>(fill);                                       // 2 MFMAs = 4096 FLOP per k-step
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float s = fill[0] + fill[1] + fill[2] + fill[3] + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (s == 1.2345f) out[0] = s;
}

// ---- M32: channels 0-31 on 32x32x2, 32-47 on 16x16x4, two waves per SIMD -----------------------------------------------------
// X64: every wave on 32x32x2 (N = 64 channels: 8 waves = 4 rows x 2 channel halves)
template <int F, bool X64>
__global__ __launch_bounds__(512) void k_m32(float* out, const float* w, int bands) {
  constexpr int CS = 33, CROW = RS * CS, CPATCH = 9 * CROW;
  __shared__ __attribute__((aligned(16))) float patch[2 * CPATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * CPATCH; i += 512) patch[i] = 0.001f * (i & 255);
  __syncthreads();
  float fill[4] = {(float)tid, tid + 1.f, tid + 2.f, tid + 3.f};
  float s = 0.f;
  const bool big = X64 || wave < 4;
  if (big) {
    // one conv2 row (32 positions) x 32 channels: lane = (position ox = lane % 32, k half kk = lane / 32); k-step s = (tap, u):
    // ci = 2 u + kk ... taken in pairs (ci, ci + 1) = 4 u' + 2 kk + {0, 1} so that ONE ds_read2_b32 feeds two k-steps
    const int ox = lane & 31, kk = lane >> 5, row = X64 ? (wave & 3) : wave, half = X64 ? (wave >> 2) : 0;
    float wr[144];
#pragma unroll
    for (int ks = 0; ks < 144; ++ks) wr[ks] = w[((32 * half + ox) % 48) * 288 + (4 * ((ks & 15) >> 1) + 2 * kk + (ks & 1)) * 9 + (ks >> 4)];
    const int aoff = (2 * row * RS + 2 * ox) * CS + 2 * kk;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int b = 0; b < bands; ++b) {
      const float* ab = patch + (b & 1) * CPATCH + aoff;
      constexpr int RDP = 3;
      auto aread = [&](int p) {       // pair p = (tap, u'): k-steps 2 p, 2 p + 1
        const int tap = p >> 3, ky = tap / 3, kx = tap % 3, u = p & 7;
        const float* q = ab + ky * CROW + kx * CS + 4 * u;
        return f32x2_t{q[0], q[1]};
      };
      f32x2_t xa[RDP];
#pragma unroll
      for (int d = 0; d < RDP; ++d) xa[d] = aread(d);
#pragma unroll
      for (int ks = 0; ks < 144; ++ks) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[(ks >> 1) % RDP][ks & 1], wr[ks], acc, 0, 0, 0);
        if ((ks & 1) && (ks >> 1) + RDP < 72) xa[(ks >> 1) % RDP] = aread((ks >> 1) + RDP);
        filler<F>(fill);                                     // 1 MFMA = 4096 FLOP per k-step
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
  } else {
    // channels 32-47: wave = one conv2 row, its two 16-column halves as two accumulator chains; k-step = (tap, m): ci = 4 m + lq ... in
    // pairs ci = 8 m' + 2 lq + {0, 1} as the shipped loop; CS = 33 makes the pair's read a ds_read2_b32
    const int lr = lane & 15, lq = lane >> 4, row = wave - 4;
    float wr[72];
#pragma unroll
    for (int ks = 0; ks < 72; ++ks) wr[ks] = w[(32 + lr) * 288 + (8 * ((ks & 7) >> 1) + 2 * lq + (ks & 1)) * 9 + (ks >> 3)];
    const int aoff = (2 * row * RS + 2 * lr) * CS + 2 * lq;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    for (int b = 0; b < bands; ++b) {
      const float* ab = patch + (b & 1) * CPATCH + aoff;
      constexpr int RDP = 2;
      auto aread = [&](int p, int colh) {
        const int tap = p >> 2, ky = tap / 3, kx = tap % 3, m = p & 3;
        const float* q = ab + ky * CROW + (kx + 32 * colh) * CS + 8 * m;
        return f32x2_t{q[0], q[1]};
      };
      f32x2_t xa0[RDP], xa1[RDP];
#pragma unroll
      for (int d = 0; d < RDP; ++d) { xa0[d] = aread(d, 0); xa1[d] = aread(d, 1); }
#pragma unroll
      for (int ks = 0; ks < 72; ++ks) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0[(ks >> 1) % RDP][ks & 1], wr[ks], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1[(ks >> 1) % RDP][ks & 1], wr[ks], acc1, 0, 0, 0);
        if ((ks & 1) && (ks >> 1) + RDP < 36) { xa0[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 0); xa1[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 1); }
        filler<F>(fill);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    s = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  }
  s += fill[0] + fill[1] + fill[2] + fill[3];
  if (s == 1.2345f) out[0] = s;
}

template <typename K>
void timeit(const char* name, K kern, int threads, double flops_per_band_wg, const float* w, float* d, int bands) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0, best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, w, bands);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-34s %8.1f us / %d bands -> %6.1f TFLOP/s (%.3f of 157.3)\n", name, best * 1e3, bands, 256.0 * bands * flops_per_band_wg / (best * 1e-3) / 1e12,
         256.0 * bands * flops_per_band_wg / (best * 1e-3) / 157.3e12);
}
int main() {
  float *w, *d; (void)hipMalloc(&w, 48 * 288 * 4); (void)hipMalloc(&d, 4); (void)hipMemset(w, 0, 48 * 288 * 4);
  const int bands = 150;
  const double band48 = 2.0 * 128 * 48 * 288, band64 = 2.0 * 128 * 64 * 288;
  for (int rep = 0; rep < 2; ++rep) {
    timeit("REF 16x16x4, 12 waves, F=0", k_ref<0>, 768, band48, w, d, bands);
    timeit("REF 16x16x4, 12 waves, F=2", k_ref<2>, 768, band48, w, d, bands);
    timeit("REF 16x16x4, 12 waves, F=4", k_ref<4>, 768, band48, w, d, bands);
    timeit("M32 32x32x2+16x16x4, 8 waves, F=0", k_m32<0, false>, 512, band48, w, d, bands);
    timeit("M32 32x32x2+16x16x4, 8 waves, F=2", k_m32<2, false>, 512, band48, w, d, bands);
    timeit("M32 32x32x2+16x16x4, 8 waves, F=4", k_m32<4, false>, 512, band48, w, d, bands);
    timeit("X64 32x32x2 only (N=64), F=0", k_m32<0, true>, 512, band64, w, d, bands);
    timeit("X64 32x32x2 only (N=64), F=2", k_m32<2, true>, 512, band64, w, d, bands);
    timeit("X64 32x32x2 only (N=64), F=4", k_m32<4, true>, 512, band64, w, d, bands);
  }
  return 0;
}
