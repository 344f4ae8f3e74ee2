// Backward of the encoder's Linear(4096 -> 64) in ONE launch (nn.Linear of `encoder_w0`, e.g.
// ANPShapeNet1D.py:55): input gradient (masked by conv3's ReLU), weight gradient and bias gradient.
//
// The three generic launches it replaces (igemm data gradient, split-K weight gradient, slab reduce) all
// stream the same operands.  Here a workgroup owns 16 of the 4096 input features: it stages the row
// gradients dY [rows][64] and its 16-column slice of the activations a3 in LDS (240 rows at a time) and
//   * accumulates dW[:, 16 cols] = dY^T a3 over ALL rows in registers (no split-K, no slab),
//   * produces d a3[:, 16 cols] = (dY W[:, 16 cols]) masked by a3 > 0, with its W slice in 16 registers.
// LDS strides: dY rows 66 words (a lane pair (row, k) of the data-gradient A operand hits 32 different
// banks), a3 rows 16 words (the two row halves of a 32-lane group sit 16 banks apart).
// dim_w == 64 only (every shipped configuration); other widths keep the generic path.  GPU build only.
#pragma once
#include "common.h"
#include "problems.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace el {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int DW = 64, KIN = 4096, RCH = 240;        // rows per LDS chunk
constexpr int DYS = 66, AS = 16;
constexpr int NTH = 512;
constexpr int LDS_FLOATS = RCH * DYS + RCH * AS;      // 19,680 floats = 78.7 KB

__global__ __launch_bounds__(NTH) void enc_linear_bwd_kernel(const Rows2 dfeat, const float* __restrict__ wl, const float* __restrict__ a3,
                                                             float* __restrict__ dy3, float* __restrict__ dwl, float* __restrict__ dbl, int n) {
  __shared__ float lds[LDS_FLOATS];
  float* s_dy = lds;                  // [RCH][DYS]
  float* s_a3 = lds + RCH * DYS;      // [RCH][AS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int i0 = blockIdx.x * 16;
  const int jt = wave & 3, rh = wave >> 2;            // weight-gradient tile (16 outputs j) and row half of the chunk
  // W[:, i0 .. i0+15] as the data gradient's B operand: lane (k = j = 4ks + lq, n = i = lr)
  float wr[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) wr[ks] = wl[(size_t)(4 * ks + lq) * KIN + i0 + lr];
  f32x4_t wacc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // the chunk's operands: dY (float4 along the 64 columns: 8 per thread) and the a3 slice (2 per thread), rows >= nr zeroed.  The
  // loads of chunk c + 1 are issued before the MFMAs of chunk c (they used to follow them: an HBM round trip per chunk in the open)
  float4 vy[8], va[2];
  auto fetch = [&](int r0) {
    const int nr = n - r0 < RCH ? n - r0 : RCH;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * NTH, row = e >> 4, c4 = e & 15;
      vy[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < RCH * 16 && row < nr) vy[u] = *reinterpret_cast<const float4*>(dfeat.row(r0 + row) + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NTH, row = e >> 2, c4 = e & 3;
      va[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < RCH * 4 && row < nr) va[u] = *reinterpret_cast<const float4*>(a3 + (size_t)(r0 + row) * KIN + i0 + 4 * c4);
    }
  };
  if (n > 0) fetch(0);
  for (int r0 = 0; r0 < n; r0 += RCH) {
    const int nr = n - r0 < RCH ? n - r0 : RCH;
    __syncthreads();
    {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = tid + u * NTH, row = e >> 4, c4 = e & 15;
        if (e < RCH * 16) {
          float* d = s_dy + row * DYS + 4 * c4;
          d[0] = vy[u].x; d[1] = vy[u].y; d[2] = vy[u].z; d[3] = vy[u].w;
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + u * NTH, row = e >> 2, c4 = e & 3;
        if (e < RCH * 4) *reinterpret_cast<float4*>(s_a3 + row * AS + 4 * c4) = va[u];
      }
    }
    __syncthreads();
    if (r0 + RCH < n) fetch(r0 + RCH);
    // weight gradient: dW[j][i] += sum_row dY[row][j] a3[row][i]; A lane (m = j, k = row), B lane (k = row, n = i)
    {
      const float* ap = s_dy + (rh * (RCH / 2) + lq) * DYS + 16 * jt + lr;
      const float* bp = s_a3 + (rh * (RCH / 2) + lq) * AS + lr;
#pragma unroll 6
      for (int ks = 0; ks < RCH / 8; ++ks) wacc = mfma4(ap[4 * ks * DYS], bp[4 * ks * AS], wacc);
    }
    if (blockIdx.x == 0) {
      // bias gradient: column sums of dY (one workgroup is enough) - over ALL eight waves, rows part, part + 8, ...: as one
      // wave walking the chunk row by row (a rolled loop: LDS read, wait, add - 480 times for the 480-image batch) workgroup 0
      // finished ~10 us after the other 255 and set the kernel's duration
      const int col = tid & 63, part = tid >> 6;
      float s0 = 0.f, s1 = 0.f;
      for (int row = part; row < nr; row += 16) {
        s0 += s_dy[row * DYS + col];
        if (row + 8 < nr) s1 += s_dy[(row + 8) * DYS + col];
      }
      bsum += s0 + s1;
    }
    // data gradient: 16-row tiles over the waves; A lane (m = row, k = j), B = wr; masked by a3 > 0
    for (int mt = wave; mt * 16 < nr; mt += 8) {
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* ap = s_dy + (16 * mt + lr) * DYS + lq;
#pragma unroll
      for (int ks = 0; ks < 16; ks += 2) {
        acc0 = mfma4(ap[4 * ks], wr[ks], acc0);
        acc1 = mfma4(ap[4 * ks + 4], wr[ks + 1], acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * mt + 4 * lq + r;
        if (row < nr) dy3[(size_t)(r0 + row) * KIN + i0 + lr] = s_a3[row * AS + lr] > 0.f ? acc0[r] + acc1[r] : 0.f;
      }
    }
  }
  // fold the two row halves of the weight gradient through LDS and store dW[16jt + 4lq + r][i0 + lr]
  __syncthreads();
  if (rh == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[(jt * 4 + r) * 64 + lane] = wacc[r];
  }
  __syncthreads();
  if (rh == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dwl[(size_t)(16 * jt + 4 * lq + r) * KIN + i0 + lr] = wacc[r] + lds[(jt * 4 + r) * 64 + lane];
  }
  if (blockIdx.x == 0) {                             // fold the eight waves' partial column sums in a fixed order
    __syncthreads();
    lds[tid] = bsum;
    __syncthreads();
    if (tid < DW) {
      float v = lds[tid];
#pragma unroll
      for (int k = 1; k < 8; ++k) v += lds[k * 64 + tid];
      dbl[tid] = v;
    }
  }
}

}  // namespace el
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
