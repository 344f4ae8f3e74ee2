// NT-Xent (the functional-contrastive term of the FCL* models) as HIP kernels.
//
// Reference: trainer/losses.py:82-99 -> pytorch_metric_learning.losses.NTXentLoss(temperature = t) on [N, d] embeddings with labels
// that are always arange blocks: contrastive_loss: labels = [0..T-1, 0..T-1] (N = 2T), contrastive_loss_ANP: labels = i / Nq
// (N = T Nq).  Both are label(i) = (i / div) % mod, so the label structure travels as two ints and nothing is built on the host.
// The package is an un-vendored, un-versioned dependency that the image lacks; its published algorithm (restated and cited in
// oracle/ref_cpu.py::nt_xent): cosine similarity s = <z_i, z_j> / (|z_i| |z_j| t); every (anchor a, positive p != a of the same
// label) contributes  -log( e^{s_ap - m} / (e^{s_ap - m} + sum_{n: label(n) != label(a)} e^{s_an - m}) + tiny ),
// m = max(s_ap, max_n s_an) treated as a constant; mean over the pairs (anchors without negatives contribute nothing).
//
// Kernels (N <= 2048, d <= 256, d % 16 == 0; one workgroup = 16 anchors):
//   forward : S[16][N] = anchors x all rows on the matrix core (both operands read as float4 along d; the row norms come out of
//             the same loads), per anchor negmax / E = sum_neg e^{s - negmax} / the pairs' loss and W = sum_p q/(q+tiny) e^{negmax-m}/den;
//             per-workgroup partial sums, a one-workgroup finish kernel divides by the pair count.
//   backward: S again, H[i][j] = G[i][j] + G[j][i] from the saved per-row statistics (G = dL/ds), dzn = H zn / t on the matrix core
//             (z read as float4 along d through a column permutation), then the normalisation's backward.
#pragma once
#include <vector>
#include "common.h"
#include "favor.h"      // MLHOT_TRY
#include "../../include/mlhot.h"

namespace mlhot {
namespace ntx {

constexpr int MAXN = 2048, MAXD = 256;     // 16 x (N + 4) similarity rows in LDS: 156 KB of the CU's 160 at the limits (FCLANP's own config reaches N = 580)
struct Args {
  const float* z; int N, d, div, mod; float inv_t;
  float *rinv, *negmax, *E, *W;         // [N] each (saved for the backward)
  float* partial;                       // [ceil(N / 16)] per-workgroup loss sums
};
inline long long pair_count(int N, int div, int mod) {      // ordered positive pairs of anchors that have at least one negative
  std::vector<long long> cnt(mod > 0 ? mod : 1, 0);
  for (int i = 0; i < N; ++i) cnt[(i / div) % mod]++;
  long long p = 0;
  for (long long c : cnt) if (c < N) p += c * (c - 1);
  return p;
}

#ifndef MLHOT_HOSTSIM
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int label_of(int i, int div, int mod) { return (i / div) % mod; }

// S tile of anchors i0 .. i0+15 against every row, into LDS s_S[16][ldS]; rinv of every row into s_rinv[N16].
// s_A: [16][d + 4] LDS tile of the RAW anchor rows (zero rows beyond N).  256 threads.
__device__ __forceinline__ void similarity_tile(const float* __restrict__ z, int N, int d, float inv_t, int i0, float* s_A, float* s_S, int ldS,
                                                float* s_rinv, int tid) {
  const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4;
  const int ldA = d + 4, ntile = (N + 15) >> 4;
  for (int e = tid; e < 16 * (d >> 2); e += 256) {
    const int r = e / (d >> 2), c4 = e - r * (d >> 2);
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (i0 + r < N) v = *reinterpret_cast<const f32x4_t*>(z + (size_t)(i0 + r) * d + 4 * c4);
    *reinterpret_cast<f32x4_t*>(s_A + r * ldA + 4 * c4) = v;
  }
  __syncthreads();
  for (int jt = wave; jt < ntile; jt += 4) {
    const int j = jt * 16 + lr, jc = j < N ? j : N - 1;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    for (int kb = 0; kb < (d >> 4); ++kb) {
      const f32x4_t b = *reinterpret_cast<const f32x4_t*>(z + (size_t)jc * d + kb * 16 + 4 * lq);
      const f32x4_t a = *reinterpret_cast<const f32x4_t*>(s_A + lr * ldA + kb * 16 + 4 * lq);
      ss += b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
      if (kb & 1) { a1 = mfma4(a[0], b[0], a1); a1 = mfma4(a[1], b[1], a1); a1 = mfma4(a[2], b[2], a1); a1 = mfma4(a[3], b[3], a1); }
      else { a0 = mfma4(a[0], b[0], a0); a0 = mfma4(a[1], b[1], a0); a0 = mfma4(a[2], b[2], a0); a0 = mfma4(a[3], b[3], a0); }
    }
    ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);          // the 4 k groups of row j
    const float rj = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (lq == 0) s_rinv[jt * 16 + lr] = rj;
#pragma unroll
    for (int r = 0; r < 4; ++r) s_S[(4 * lq + r) * ldS + jt * 16 + lr] = (a0[r] + a1[r]) * rj;       // x rinv_i x 1/t below
  }
  __syncthreads();
  for (int e = tid; e < 16 * ntile * 16; e += 256) {
    const int r = e / (ntile * 16), j = e - r * (ntile * 16);
    const int i = i0 + r;
    s_S[r * ldS + j] *= (i < N ? s_rinv[i] : 0.f) * inv_t;
  }
  __syncthreads();
}

// per-anchor statistics: 16 threads per anchor (256 threads = 16 anchors)
struct RowStat { float negmax, E, W, lsum; bool has_neg; };
__device__ __forceinline__ RowStat row_stats(const float* s_row, int i, int N, int div, int mod, int part) {
  RowStat st{-INFINITY, 0.f, 0.f, 0.f, false};
  if (i >= N) return st;
  const int li = label_of(i, div, mod);
  float nm = -INFINITY;
  for (int j = part; j < N; j += 16)
    if (label_of(j, div, mod) != li) nm = fmaxf(nm, s_row[j]);
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) nm = fmaxf(nm, __shfl_xor(nm, off, 64));
  st.has_neg = nm > -INFINITY;
  if (!st.has_neg) return st;
  float e = 0.f;
  for (int j = part; j < N; j += 16)
    if (label_of(j, div, mod) != li) e += expf(s_row[j] - nm);
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) e += __shfl_xor(e, off, 64);
  float ls = 0.f, w = 0.f;
  const float tiny = 1.17549435e-38f;
  for (int j = part; j < N; j += 16)
    if (j != i && label_of(j, div, mod) == li) {
      const float s = s_row[j], m = fmaxf(s, nm);
      const float num = expf(s - m), den = e * expf(nm - m) + num, q = num / den;
      ls += -logf(q + tiny);
      w += q / (q + tiny) * expf(nm - m) / den;
    }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) { ls += __shfl_xor(ls, off, 64); w += __shfl_xor(w, off, 64); }
  st.negmax = nm; st.E = e; st.W = w; st.lsum = ls;
  return st;
}

__global__ __launch_bounds__(256) void ntx_fwd_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, i0 = blockIdx.x * 16;
  const int N16 = (a.N + 15) & ~15, ldS = N16 + 4;
  float* s_A = lds;                              // [16][d + 4]
  float* s_S = s_A + 16 * (a.d + 4);             // [16][ldS]
  float* s_rinv = s_S + 16 * ldS;                // [N16]
  float* s_red = s_rinv + N16;                   // [16]
  similarity_tile(a.z, a.N, a.d, a.inv_t, i0, s_A, s_S, ldS, s_rinv, tid);
  const int r = tid >> 4, part = tid & 15, i = i0 + r;
  const RowStat st = row_stats(s_S + r * ldS, i, a.N, a.div, a.mod, part);
  if (part == 0) {
    s_red[r] = st.has_neg ? st.lsum : 0.f;
    if (i < a.N) { a.negmax[i] = st.negmax; a.E[i] = st.E; a.W[i] = st.has_neg ? st.W : 0.f; }
  }
  if (blockIdx.x == 0) for (int j = tid; j < a.N; j += 256) a.rinv[j] = s_rinv[j];
  __syncthreads();
  if (tid == 0) {
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += s_red[k];
    a.partial[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(64) void ntx_finish_kernel(const float* __restrict__ partial, int n, float inv_pairs, float* __restrict__ loss) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += partial[k];       // fixed order
    loss[0] = s * inv_pairs;
  }
}

struct BwdArgs {
  Args f; const float* gscale;      // device scalar: upstream gradient of the loss
  float inv_pairs; float* dz;
};

__global__ __launch_bounds__(256) void ntx_bwd_kernel(const BwdArgs b) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Args& a = b.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lq = lane >> 4, i0 = blockIdx.x * 16;
  const int N16 = (a.N + 15) & ~15, ldS = N16 + 4, ldA = a.d + 4;
  float* s_A = lds;
  float* s_S = s_A + 16 * ldA;
  float* s_rinv = s_S + 16 * ldS;
  float* s_dot = s_rinv + N16;                   // [4 waves][16]
  similarity_tile(a.z, a.N, a.d, a.inv_t, i0, s_A, s_S, ldS, s_rinv, tid);
  // H'[i][j] = (G[i][j] + G[j][i]) rinv_j, in place of S (G = dL/ds per pair-sum, the 1/pairs factor rides in gscale)
  const float tiny = 1.17549435e-38f;
  for (int e = tid; e < 16 * N16; e += 256) {
    const int r = e / N16, j = e - r * N16, i = i0 + r;
    float h = 0.f;
    if (i < a.N && j < a.N && j != i) {
      const float s = s_S[r * ldS + j];
      const bool same = label_of(i, a.div, a.mod) == label_of(j, a.div, a.mod);
      const float nmi = a.negmax[i], nmj = a.negmax[j];
      if (same) {
        if (nmi > -INFINITY) { const float m = fmaxf(s, nmi), num = expf(s - m), den = a.E[i] * expf(nmi - m) + num, q = num / den; h += -q * (1.f - q) / (q + tiny); }
        if (nmj > -INFINITY) { const float m = fmaxf(s, nmj), num = expf(s - m), den = a.E[j] * expf(nmj - m) + num, q = num / den; h += -q * (1.f - q) / (q + tiny); }
      } else {
        h = expf(s - nmi) * a.W[i] + expf(s - nmj) * a.W[j];      // both have a negative (each other)
      }
      h *= s_rinv[j];
    }
    s_S[r * ldS + j] = h;
  }
  __syncthreads();
  // dzn[i][c] = inv_t sum_j H'[i][j] z[j][c]: wave w owns columns [64 w, 64 w + 64); MFMA te <-> column 64 w + 4 lr + te
  const float g = b.gscale[0] * b.inv_pairs * a.inv_t;
  f32x4_t acc[4];
#pragma unroll
  for (int te = 0; te < 4; ++te) acc[te] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int c0 = 64 * wave + 4 * lr;
  const bool colok = c0 < a.d;
  if (colok) {
    for (int jb = 0; jb < (N16 >> 2); ++jb) {
      const int j = 4 * jb + lq;
      f32x4_t zb = {0.f, 0.f, 0.f, 0.f};
      if (j < a.N) zb = *reinterpret_cast<const f32x4_t*>(a.z + (size_t)j * a.d + c0);
      const float av = s_S[lr * ldS + j];
#pragma unroll
      for (int te = 0; te < 4; ++te) acc[te] = mfma4(av, zb[te], acc[te]);
    }
  }
  // normalisation backward: dz_i = rinv_i (dzn_i - zn_i (zn_i . dzn_i)); this lane: rows 4 lq + r, columns c0 .. c0 + 3
  float dotp[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lq + r;
    float dsum = 0.f;
    if (colok) {
      const f32x4_t zi = *reinterpret_cast<const f32x4_t*>(s_A + row * ldA + c0);
#pragma unroll
      for (int te = 0; te < 4; ++te) dsum += acc[te][r] * zi[te];
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) dsum += __shfl_xor(dsum, off, 64);
    dotp[r] = dsum;
    if (lr == 0) s_dot[wave * 16 + row] = dsum;
  }
  __syncthreads();
  if (colok) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r, i = i0 + row;
      if (i < a.N) {
        const float ri = s_rinv[i];
        const float dot = (s_dot[row] + s_dot[16 + row] + s_dot[32 + row] + s_dot[48 + row]) * ri * ri;     // zn . dzn, zn = z rinv
        const f32x4_t zi = *reinterpret_cast<const f32x4_t*>(s_A + row * ldA + c0);
        f32x4_t o;
#pragma unroll
        for (int te = 0; te < 4; ++te) o[te] = g * ri * (acc[te][r] - zi[te] * dot);
        *reinterpret_cast<f32x4_t*>(b.dz + (size_t)i * a.d + c0) = o;
      }
    }
  }
  (void)dotp;
}

inline size_t lds_bytes(int N, int d) {
  const int N16 = (N + 15) & ~15;
  return sizeof(float) * (16 * (d + 4) + 16 * (N16 + 4) + N16 + 64);
}
#endif

}  // namespace ntx
}  // namespace mlhot
