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
// Where its 12.3 us go (knock-outs, rocprofv3): 5.5 us with nothing but the loads, the barriers and the stores of zeros - launch ramp
// and the first (cold) round trip; the data gradient 4.1, the weight gradient 2.6.  Measured and dropped: the rows split over
// workgroup pairs with 32 features each (half of dY per workgroup: 99 KB fetched instead of 153, two partial dW rows folded by the
// deferred sum) - 12.2 us, the same; both feature tiles of a row tile sharing the A operand in four chains - the same again.  The
// kernel is latency and 64-byte store transactions (dy3 rows are 16 KB apart), not fetch volume or MFMA chains.
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
    if (blockIdx.x < DW && wave == 7) {
      // bias gradient: column sums of dY.  Every workgroup has the whole dY chunk in LDS anyway, so workgroup b < 64 sums column b
      // (wave 7, which has the fewest data-gradient tiles: rows lane, lane + 64, ...).  (As eight waves of workgroup 0 summing all 64
      // columns, that workgroup finished 2.2 us after the other 255 and set the kernel's duration: 14.5 -> 12.3 us without it.)
      float sb = 0.f;
      for (int row = lane; row < nr; row += 64) sb += s_dy[row * DYS + blockIdx.x];
      bsum += sb;
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
  if (blockIdx.x < DW && wave == 7) {                // fold the 64 lanes' partial sums in a fixed order
    float v = bsum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) dbl[blockIdx.x] = v;
  }
}

// Forward of the same layer: feat[n][64] = a3[n][4096] wl[64][4096]^T (+ bias in the slab fold).  A [480 x 64] result has 120 output
// tiles - too few for the chip - so K is split 16 ways: workgroup = (16 rows, 256 k), wave = 16 outputs, 480 workgroups for the
// 480-image batch; the 16 partial results [16][n][64] are folded (+ bias) by enc_linear_fold_kernel.  (The generic 64 x 64
// igemm needed 32 K splits to fill the chip: 3.9 MB of slabs, 9.0 + 4.8 us.)  K runs in chunks of 64, double-buffered in LDS: every
// thread brings 5 float4 per chunk, whole 256-byte row pieces per 16 lanes.  Operands are ds_read_b128: a lane (row lr, k-group lq) takes 4
// consecutive k at 16 j + 4 lq and MFMA (j, i) uses element i of every lane's float4 - the k order inside a block of 16 is permuted
// the same way for both operands, which a dot product does not see.  Row stride 72 words = 8 mod 16: a 16-lane group of a
// ds_read_b128 covers the 64 banks once.  Measured (rocprofv3): product 9.0 -> 7.3 us, fold 4.8 -> 4.7 us.  Both are latency, not
// work: the product reads a cold a3 (7.9 MB, ~1.1 TB/s over the kernel), the fold is one round trip.  On the way: the same float4s
// fed to the MFMAs straight from global memory (no LDS): 9.2 us, and 5.2 us with every load hitting ONE cache line - 64 loads of
// 16 rows x 64 bytes per wave are bound by the vector memory pipe's request rate, not by bytes, MFMAs or loads in flight (32 per wave
// made no difference, nor did walking the K slice from a different start per workgroup to spread the L2 channels).
constexpr int F_KS = 16, F_KLEN = KIN / F_KS, F_CH = 64, F_NCH = F_KLEN / F_CH, F_LD = 72;
constexpr int F_BUF = (16 + DW) * F_LD;                               // floats per chunk buffer: A rows 0..15, then the 64 wl rows
__global__ __launch_bounds__(256) void enc_linear_fwd_kernel(const float* __restrict__ a3, const float* __restrict__ wl, float* __restrict__ slab, int n) {
  __shared__ __attribute__((aligned(16))) float lds[2 * F_BUF];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int mt = (n + 15) / 16, mtile = blockIdx.x % mt, ks = blockIdx.x / mt;
  const int r0 = 16 * mtile;
  // staging items: (row of the 80-row chunk image, float4 of its 16): item e = tid + 256 u, u < 5; rows 0..15 = a3, 16..79 = wl
#define EL_ITEM(u)                                                                                                              \
  const int e##u = tid + 256 * u, row##u = e##u >> 4, c4##u = e##u & 15;                                                         \
  const float* src##u = (row##u < 16 ? a3 + (size_t)min(r0 + row##u, n - 1) * KIN : wl + (size_t)(row##u - 16) * KIN) + ks * F_KLEN + 4 * c4##u; \
  const int dst##u = row##u * F_LD + 4 * c4##u;
  EL_ITEM(0) EL_ITEM(1) EL_ITEM(2) EL_ITEM(3) EL_ITEM(4)
#undef EL_ITEM
  // ALL of the workgroup's 80 KB are requested up front (20 float4 per thread): a3 is cold (HBM / fabric latency ~2 us) and with
  // one chunk of look-ahead every chunk waited that latency out again (8 K splits x 8 chunks: 8.1 us); 8 K splits with all 40
  // float4 up front went to scratch memory (21 us)
  // twenty NAMED float4s, loads and stores written out: as arrays indexed in `#pragma unroll` loops (or passed to a lambda) hipcc
  // unrolled after it had already decided to keep them in scratch memory (21 us)
  static_assert(F_NCH == 4, "four chunk register sets");
  float4 v00, v01, v02, v03, v04, v10, v11, v12, v13, v14, v20, v21, v22, v23, v24, v30, v31, v32, v33, v34;
  v00 = *reinterpret_cast<const float4*>(src0 + 0 * F_CH);
  v01 = *reinterpret_cast<const float4*>(src1 + 0 * F_CH);
  v02 = *reinterpret_cast<const float4*>(src2 + 0 * F_CH);
  v03 = *reinterpret_cast<const float4*>(src3 + 0 * F_CH);
  v04 = *reinterpret_cast<const float4*>(src4 + 0 * F_CH);
  v10 = *reinterpret_cast<const float4*>(src0 + 1 * F_CH);
  v11 = *reinterpret_cast<const float4*>(src1 + 1 * F_CH);
  v12 = *reinterpret_cast<const float4*>(src2 + 1 * F_CH);
  v13 = *reinterpret_cast<const float4*>(src3 + 1 * F_CH);
  v14 = *reinterpret_cast<const float4*>(src4 + 1 * F_CH);
  v20 = *reinterpret_cast<const float4*>(src0 + 2 * F_CH);
  v21 = *reinterpret_cast<const float4*>(src1 + 2 * F_CH);
  v22 = *reinterpret_cast<const float4*>(src2 + 2 * F_CH);
  v23 = *reinterpret_cast<const float4*>(src3 + 2 * F_CH);
  v24 = *reinterpret_cast<const float4*>(src4 + 2 * F_CH);
  v30 = *reinterpret_cast<const float4*>(src0 + 3 * F_CH);
  v31 = *reinterpret_cast<const float4*>(src1 + 3 * F_CH);
  v32 = *reinterpret_cast<const float4*>(src2 + 3 * F_CH);
  v33 = *reinterpret_cast<const float4*>(src3 + 3 * F_CH);
  v34 = *reinterpret_cast<const float4*>(src4 + 3 * F_CH);
  __builtin_amdgcn_sched_barrier(0);
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int aoff = lr * F_LD + 4 * lq, boff = (16 + 16 * w + lr) * F_LD + 4 * lq;
#define EL_MFMAS(buf)                                                                                                     \
  _Pragma("unroll") for (int j = 0; j < F_CH / 16; ++j) {                                                                 \
    const float4 a = *reinterpret_cast<const float4*>(buf + aoff + 16 * j), b = *reinterpret_cast<const float4*>(buf + boff + 16 * j); \
    acc0 = mfma4(a.x, b.x, acc0);                                                                                          \
    acc1 = mfma4(a.y, b.y, acc1);                                                                                          \
    acc0 = mfma4(a.z, b.z, acc0);                                                                                          \
    acc1 = mfma4(a.w, b.w, acc1);                                                                                          \
  }
  {
    float* buf = lds + 0 * F_BUF;
    *reinterpret_cast<float4*>(buf + dst0) = v00;
    *reinterpret_cast<float4*>(buf + dst1) = v01;
    *reinterpret_cast<float4*>(buf + dst2) = v02;
    *reinterpret_cast<float4*>(buf + dst3) = v03;
    *reinterpret_cast<float4*>(buf + dst4) = v04;
    __syncthreads();                          // one barrier per chunk: the buffer written now was last read two chunks ago
    EL_MFMAS(buf)
  }
  {
    float* buf = lds + 1 * F_BUF;
    *reinterpret_cast<float4*>(buf + dst0) = v10;
    *reinterpret_cast<float4*>(buf + dst1) = v11;
    *reinterpret_cast<float4*>(buf + dst2) = v12;
    *reinterpret_cast<float4*>(buf + dst3) = v13;
    *reinterpret_cast<float4*>(buf + dst4) = v14;
    __syncthreads();                          // one barrier per chunk: the buffer written now was last read two chunks ago
    EL_MFMAS(buf)
  }
  {
    float* buf = lds + 0 * F_BUF;
    *reinterpret_cast<float4*>(buf + dst0) = v20;
    *reinterpret_cast<float4*>(buf + dst1) = v21;
    *reinterpret_cast<float4*>(buf + dst2) = v22;
    *reinterpret_cast<float4*>(buf + dst3) = v23;
    *reinterpret_cast<float4*>(buf + dst4) = v24;
    __syncthreads();                          // one barrier per chunk: the buffer written now was last read two chunks ago
    EL_MFMAS(buf)
  }
  {
    float* buf = lds + 1 * F_BUF;
    *reinterpret_cast<float4*>(buf + dst0) = v30;
    *reinterpret_cast<float4*>(buf + dst1) = v31;
    *reinterpret_cast<float4*>(buf + dst2) = v32;
    *reinterpret_cast<float4*>(buf + dst3) = v33;
    *reinterpret_cast<float4*>(buf + dst4) = v34;
    __syncthreads();                          // one barrier per chunk: the buffer written now was last read two chunks ago
    EL_MFMAS(buf)
  }
#undef EL_MFMAS
  float* out = slab + ((size_t)ks * n + r0) * DW + 16 * w + lr;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r0 + 4 * lq + r < n) out[(size_t)(4 * lq + r) * DW] = acc0[r] + acc1[r];
}

// feat[m][j] = bias[j] + sum over the F_KS partial results, in a fixed order; one output per thread, all 16 loads in flight
__global__ __launch_bounds__(256) void enc_linear_fold_kernel(const float* __restrict__ slab, const float* __restrict__ bias, Rows2 out, int n) {
  const int e = blockIdx.x * 256 + threadIdx.x, total = n * DW;
  if (e >= total) return;
  float p[F_KS];
#pragma unroll
  for (int z = 0; z < F_KS; ++z) p[z] = slab[(size_t)z * total + e];
  float t = bias[e & (DW - 1)];
#pragma unroll
  for (int z = 0; z < F_KS; ++z) t += p[z];
  out.row(e / DW)[e & (DW - 1)] = t;
}

}  // namespace el
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
