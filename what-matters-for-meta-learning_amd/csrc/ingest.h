// Batch ingest: the byte -> float image conversion the reference's data loaders do on the host
// (dataset/shapenet_1d.py:189-190 `xs.astype(np.float32) / 255.0`, then utils/utils.py:26-30 channel-last -> channel-first),
// moved behind the PCIe copy so that a batch crosses the bus as uint8 (4x fewer bytes).
//
// HBM-bound byte work, no reuse: every source byte is read once and every destination float written once
// (algorithmic bytes per pixel-channel: 1 read + 4 written).  A lane owns a quad of 4 consecutive pixels; packed NHWC means
// quad q of the whole batch starts at byte 4*C*q, so a wave reads 256*C contiguous bytes and writes C runs of 1 KiB.
// The division is IEEE (hipcc's default correctly-rounded fp32 divide), hence bit-identical to numpy's float32 / 255.0.
#pragma once
#include "common.h"
#include "foreach.h"

namespace mlhot {
namespace ingest {

#ifndef MLHOT_HOSTSIM

constexpr int NT = 256;
constexpr int QPT = 4;            // independent quads in flight per lane

template <int C>
struct QuadBytes { uint32_t w[C]; };       // 4 pixels x C channels = 4*C bytes = C dwords

template <int C>
__device__ __forceinline__ uint8_t quad_byte(const QuadBytes<C>& q, int i) { return (uint8_t)(q.w[i >> 2] >> (8 * (i & 3))); }

template <int C>
__global__ __launch_bounds__(NT) void u8_nhwc_to_f32_nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                 long n_quads, int quads_per_img, int HW, float div) {
  const long stride = (long)gridDim.x * NT;
  const long q0 = (long)blockIdx.x * NT + threadIdx.x;
  QuadBytes<C> qb[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const long q = q0 + u * stride;
    if (q < n_quads) qb[u] = *reinterpret_cast<const QuadBytes<C>*>(src + q * (4 * C));
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const long q = q0 + u * stride;
    if (q >= n_quads) continue;
    const long img = q / quads_per_img;
    const int p = (int)(q - img * quads_per_img) * 4;
    float* o = dst + img * (long)C * HW + p;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float4 v;
      v.x = (float)quad_byte<C>(qb[u], 0 * C + c) / div;
      v.y = (float)quad_byte<C>(qb[u], 1 * C + c) / div;
      v.z = (float)quad_byte<C>(qb[u], 2 * C + c) / div;
      v.w = (float)quad_byte<C>(qb[u], 3 * C + c) / div;
      *reinterpret_cast<float4*>(o + (long)c * HW) = v;
    }
  }
}

#endif  // !MLHOT_HOSTSIM

// any H*W (not a multiple of 4) and any channel count: one index per destination float
struct IngestAny {
  const uint8_t* src; float* dst; int C, HW; float div;
  MLHOT_HD void operator()(size_t i) const {
    const size_t img = i / ((size_t)C * HW);
    const int r = (int)(i - img * (size_t)C * HW), c = r / HW, p = r - c * HW;
    dst[i] = (float)src[(img * HW + p) * C + c] / div;
  }
};

inline int run(const uint8_t* src, float* dst, long n_img, int H, int W, int C, float div, hipStream_t s) {
  const int HW = H * W;
  const long total = n_img * (long)C * HW;
  if (total == 0) return MLHOT_OK;
#ifndef MLHOT_HOSTSIM
  if ((HW & 3) == 0 && C >= 1 && C <= 4 && (reinterpret_cast<uintptr_t>(src) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const long n_quads = n_img * (long)(HW / 4);
    const int grid = (int)((n_quads + (long)NT * QPT - 1) / ((long)NT * QPT));
    ProfScope ps("ingest.u8", s);
    switch (C) {
      case 1: hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel<1>, dim3(grid), dim3(NT), 0, s, src, dst, n_quads, HW / 4, HW, div); break;
      case 2: hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel<2>, dim3(grid), dim3(NT), 0, s, src, dst, n_quads, HW / 4, HW, div); break;
      case 3: hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel<3>, dim3(grid), dim3(NT), 0, s, src, dst, n_quads, HW / 4, HW, div); break;
      default: hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel<4>, dim3(grid), dim3(NT), 0, s, src, dst, n_quads, HW / 4, HW, div); break;
    }
    return check_launch("ingest.u8");
  }
#endif
  return run_foreach(IngestAny{src, dst, C, HW, div}, (size_t)total, s, "ingest.u8.any");
}

// ---- host side: the inverse of the loaders' `astype(float32) / div` (dataset/shapenet_1d.py:189-190), checked element by element -----
// A reference-style loader hands out fp32 host images that ARE k / 255 for byte values k; such a batch can cross PCIe as bytes (a
// quarter of the traffic) and be expanded again by the ingest kernel - bit for bit the same fp32 values - but only if EVERY element
// round-trips.  One pass: k = round(x * div) clamped to a byte, dst = k, and the count of elements with (float)k / div != x.  Plain
// loops the host compiler vectorises; on x86 a second copy of the loop is compiled for AVX2 and picked at run time (the fp32 divide
// is the cost: 8 lanes instead of 4).  Callers cut a batch into pieces and run them on several host threads (mlhot/ingest.py).
template <int UNUSED = 0>
static inline long host_f32_to_u8_exact_body(const float* src, uint8_t* dst, long n, float div) {
  long bad = 0;
  for (long b0 = 0; b0 < n; b0 += 32768) {                  // 32-bit counters inside a block: the loop vectorises at the floats' width
    const int m = (int)(n - b0 < 32768 ? n - b0 : 32768);
    const float* __restrict__ sp = src + b0;
    uint8_t* __restrict__ dp = dst + b0;
    int bad32 = 0;
    for (int i = 0; i < m; ++i) {
      const float x = sp[i];
      float r = x * div + 0.5f;
      r = r > 0.f ? r : 0.f;                                 // a NaN lands on 0 here (the comparison is false) and fails the check below
      r = r < 255.f ? r : 255.f;
      const int k = (int)r;
      dp[i] = (uint8_t)k;
      bad32 += ((float)k / div != x) ? 1 : 0;
    }
    bad += bad32;
  }
  return bad;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) static long host_f32_to_u8_exact_avx2(const float* src, uint8_t* dst, long n, float div) {
  return host_f32_to_u8_exact_body<1>(src, dst, n, div);
}
#endif
inline long host_f32_to_u8_exact(const float* src, uint8_t* dst, long n, float div) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  if (__builtin_cpu_supports("avx2")) return host_f32_to_u8_exact_avx2(src, dst, n, div);
#endif
  return host_f32_to_u8_exact_body<0>(src, dst, n, div);
}

}  // namespace ingest
}  // namespace mlhot
