// nn.Linear forward / data gradient / weight gradient for FEW ROWS (M <= 512: the task-side layers of the ResNet-family and MR
// models run on T x N = 120..480 rows, widths 256..2048) straight from global memory on the fp32 matrix core - no LDS staging,
// no split-K slab, one launch each.  The generic implicit-GEMM engine is built for large M: with a handful of 16-row tiles it
// left most of the chip idle and walked K = 2048 serially (15-25 us per layer; 42 such launches were 0.7 ms of a 2.7 ms c5 step).
//
//   * operands are read in MFMA lane order directly: lane (lr, lq) loads a float4 of 4 consecutive k (k0 + 4 lq .. + 3) of its
//     row; MFMA i of the four that consume it pairs element i of A with element i of B, i.e. the k index of lane group lq in
//     MFMA i is k0 + 4 lq + i - the same permutation on both operands, so the product is exact;
//   * the reduction dimension is split over the four waves of a workgroup and folded through 8 KB of LDS in a fixed order
//     (bitwise reproducible); a workgroup owns a 32 x 16 output tile, so even a [120 x 256] layer spreads over 64 workgroups;
//   * the weight gradient reduces over the rows (M <= 512): dword loads that are contiguous across the lanes of a tile row.
// v_mfma_f32_16x16x4_f32: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], C/D lane l reg r = C[4*(l>>4)+r][l&15].
#pragma once
#include "common.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace sk {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float4 ld4(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float dact(int act, float y) { return act == ACT_RELU ? (y > 0.f ? 1.f : 0.f) : (act == ACT_TANH ? 1.f - y * y : 1.f); }

// fold the four waves' partial tiles (2 tiles x 4 floats per lane) through LDS: returns the sum in waves' lanes of thread < 128
__device__ __forceinline__ bool fold4(f32x4_t (&acc)[2], float* red, int tid) {
  const int lane = tid & 63, w = tid >> 6;
#pragma unroll
  for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4_t*>(red + ((w * 2 + t) * 64 + lane) * 4) = acc[t];
  __syncthreads();
  if (tid >= 128) return false;
  const int t = tid >> 6;
  f32x4_t s = *reinterpret_cast<const f32x4_t*>(red + ((0 * 2 + t) * 64 + lane) * 4);
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(red + ((k * 2 + t) * 64 + lane) * 4);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  acc[0] = s;
  return true;
}

// y[M][N] = act(x[M][K] w[N][K]^T + b): grid (ceil(M / 32), ceil(N / 16)), 256 threads
// Two-source input (the reference's torch.cat in front of a layer, folded): columns [0, K1) of the layer's input come from x,
// columns [K1, K) from x2 (K1 % 4 == 0; x2 = nullptr / K1 >= K: one source).  The same for the data gradient's destination
// (dx | dx2) and the weight gradient's operand.
__device__ __forceinline__ void fwd_body(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw, const float* __restrict__ b,
                                         float* __restrict__ y, int ldy, int M, int K, int N, int act, float* red, int bx, int by,
                                         const float* __restrict__ x2 = nullptr, int ldx2 = 0, int K1 = 1 << 30) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int m0 = bx * 32, n0 = by * 16;
  const int ra = m0 + lr, rb = m0 + 16 + lr, rn = n0 + lr;
  const float* xa = x + (size_t)(ra < M ? ra : 0) * ldx + 4 * lq;
  const float* xb = x + (size_t)(rb < M ? rb : 0) * ldx + 4 * lq;
  const float* xa2 = x2 ? x2 + (size_t)(ra < M ? ra : 0) * ldx2 + 4 * lq - K1 : xa;      // second source, indexed by the layer's k as well
  const float* xb2 = x2 ? x2 + (size_t)(rb < M ? rb : 0) * ldx2 + 4 * lq - K1 : xb;
  const float* wp = w + (size_t)(rn < N ? rn : 0) * ldw + 4 * lq;
  f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  // four 16-deep k chunks per trip: 12 independent float4 loads are in flight before the first MFMA needs one
  for (int kb = 16 * wv; kb < K; kb += 256) {
    float4 a0[4], a1[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb + 64 * u;
      const bool kin = k0 + 4 * lq < K, first = k0 + 4 * lq < K1;
      a0[u] = ld4((first ? xa : xa2) + k0, kin && ra < M); a1[u] = ld4((first ? xb : xb2) + k0, kin && rb < M); bv[u] = ld4(wp + k0, kin && rn < N);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = mfma4(a0[u].x, bv[u].x, acc[0]); acc[1] = mfma4(a1[u].x, bv[u].x, acc[1]);
      acc[0] = mfma4(a0[u].y, bv[u].y, acc[0]); acc[1] = mfma4(a1[u].y, bv[u].y, acc[1]);
      acc[0] = mfma4(a0[u].z, bv[u].z, acc[0]); acc[1] = mfma4(a1[u].z, bv[u].z, acc[1]);
      acc[0] = mfma4(a0[u].w, bv[u].w, acc[0]); acc[1] = mfma4(a1[u].w, bv[u].w, acc[1]);
    }
  }
  if (!fold4(acc, red, tid)) return;
  const int t = tid >> 6, n = n0 + lr;
  if (n >= N) return;
  const float bn = b ? b[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 16 * t + 4 * lq + r;
    if (m < M) y[(size_t)m * ldy + n] = act_apply(act, acc[0][r] + bn);
  }
}
__global__ __launch_bounds__(256) void fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w, int ldw, const float* __restrict__ b,
                                                  float* __restrict__ y, int ldy, int M, int K, int N, int act) {
  __shared__ float red[4 * 2 * 64 * 4];
  fwd_body(x, ldx, w, ldw, b, y, ldy, M, K, N, act, red, blockIdx.x, blockIdx.y);
}

// dx[M][Kin] (+)= (dy * act'(y))[M][N] w[N][Kin]: out^T tile C[kin][m]; grid (ceil(M / 32), ceil(Kin / 16))
__device__ __forceinline__ void dgrad_body(const float* __restrict__ dy, int lddy, const float* __restrict__ yv, int ldy, int act,
                                           const float* __restrict__ w, int ldw, float* __restrict__ dx, int lddx, int accumulate,
                                           int M, int Kin, int N, float* red, int bx, int by,
                                           float* __restrict__ dx2 = nullptr, int lddx2 = 0, int K1 = 1 << 30) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int m0 = bx * 32, c0 = by * 16;
  const int ra = m0 + lr, rb = m0 + 16 + lr, col = c0 + lr;
  const size_t oa = (size_t)(ra < M ? ra : 0), ob = (size_t)(rb < M ? rb : 0);
  const float* wp = w + (col < Kin ? col : 0);
  f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int nb = 16 * wv; nb < N; nb += 128) {
    float4 g0[2], g1[2], y0[2], y1[2];
    float wq[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = nb + 64 * u + 4 * lq;
      const bool nin = n < N;
      g0[u] = ld4(dy + oa * lddy + n, nin && ra < M); g1[u] = ld4(dy + ob * lddy + n, nin && rb < M);
      if (act != ACT_NONE) { y0[u] = ld4(yv + oa * ldy + n, nin && ra < M); y1[u] = ld4(yv + ob * ldy + n, nin && rb < M); }
      // A[row = kin (lr)][k = n (lq)] = w[n + i][c0 + lr]
      const bool cin = nin && col < Kin;
#pragma unroll
      for (int i = 0; i < 4; ++i) wq[u][i] = cin ? wp[(size_t)(n + i) * ldw] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (act != ACT_NONE) {
        g0[u].x *= dact(act, y0[u].x); g0[u].y *= dact(act, y0[u].y); g0[u].z *= dact(act, y0[u].z); g0[u].w *= dact(act, y0[u].w);
        g1[u].x *= dact(act, y1[u].x); g1[u].y *= dact(act, y1[u].y); g1[u].z *= dact(act, y1[u].z); g1[u].w *= dact(act, y1[u].w);
      }
      acc[0] = mfma4(wq[u][0], g0[u].x, acc[0]); acc[1] = mfma4(wq[u][0], g1[u].x, acc[1]);
      acc[0] = mfma4(wq[u][1], g0[u].y, acc[0]); acc[1] = mfma4(wq[u][1], g1[u].y, acc[1]);
      acc[0] = mfma4(wq[u][2], g0[u].z, acc[0]); acc[1] = mfma4(wq[u][2], g1[u].z, acc[1]);
      acc[0] = mfma4(wq[u][3], g0[u].w, acc[0]); acc[1] = mfma4(wq[u][3], g1[u].w, acc[1]);
    }
  }
  if (!fold4(acc, red, tid)) return;
  const int t = tid >> 6, m = m0 + 16 * t + lr, c = c0 + 4 * lq;        // C[row = kin 4 lq + r][col = m lr]
  if (m >= M || c >= Kin) return;
  if (c >= K1 && dx2 == nullptr) return;                                  // the second source's gradient is not wanted
  float* d = c < K1 ? dx + (size_t)m * lddx + c : dx2 + (size_t)m * lddx2 + (c - K1);
  if (c < K1 && dx == nullptr) return;
  if (c + 3 < Kin && (c + 3 < K1 || c >= K1) && ((reinterpret_cast<uintptr_t>(d) & 15) == 0)) {
    float4 v = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    if (accumulate) { const float4 u = *reinterpret_cast<const float4*>(d); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    *reinterpret_cast<float4*>(d) = v;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c + r < Kin && ((c + r < K1) == (c < K1))) d[r] = accumulate ? d[r] + acc[0][r] : acc[0][r];      // K1 % 4 == 0: a group of 4 never straddles
  }
}

__global__ __launch_bounds__(256) void dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ yv, int ldy, int act,
                                                    const float* __restrict__ w, int ldw, float* __restrict__ dx, int lddx, int accumulate,
                                                    int M, int Kin, int N) {
  __shared__ float red[4 * 2 * 64 * 4];
  dgrad_body(dy, lddy, yv, ldy, act, w, ldw, dx, lddx, accumulate, M, Kin, N, red, blockIdx.x, blockIdx.y);
}

// dw[N][K] = (dy * act'(y))^T x ; db[N] = column sums.  Workgroup: 16 output rows (n) x 64 columns (k); the reduction over the M
// rows is split over the four waves (row groups of 4: g = wave, wave + 4, ...) and folded through LDS in a fixed order.
// grid (ceil(N / 16), ceil(K / 64))
__device__ __forceinline__ void wgrad_body(const float* __restrict__ dy, int lddy, const float* __restrict__ yv, int ldy, int act,
                                           const float* __restrict__ x, int ldx, float* __restrict__ dw, int lddw, float* __restrict__ db,
                                           int M, int K, int N, float* red, int bx, int by,
                                           const float* __restrict__ x2 = nullptr, int ldx2 = 0, int K1 = 1 << 30) {
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int n0 = bx * 16, k0 = by * 64;
  const int n = n0 + lr;
  const bool nin = n < N;
  f32x4_t acc[5] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const bool with_b = db != nullptr && by == 0;
  for (int mb = 4 * wv; mb < M; mb += 64) {           // four row groups (of 4 rows) per trip: all their loads in flight
    float g[4], yq[4], xv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = mb + 16 * u + lq;
      const bool min_ = m < M;
      g[u] = (min_ && nin) ? dy[(size_t)m * lddy + n] : 0.f;                   // A[row = n (lr)][k = m (lq)]
      yq[u] = (act != ACT_NONE && min_ && nin) ? yv[(size_t)m * ldy + n] : 0.f;
      const float* xr = x + (size_t)(min_ ? m : 0) * ldx + k0 + lr;             // B[k = m (lq)][col = k (lr)]
      const float* xr2 = x2 ? x2 + (size_t)(min_ ? m : 0) * ldx2 + k0 + lr - K1 : xr;
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[u][j] = (min_ && k0 + 16 * j + lr < K) ? (k0 + 16 * j + lr < K1 ? xr : xr2)[16 * j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float gg = act != ACT_NONE ? g[u] * dact(act, yq[u]) : g[u];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = mfma4(gg, xv[u][j], acc[j]);
      if (with_b) acc[4] = mfma4(gg, 1.f, acc[4]);
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) *reinterpret_cast<f32x4_t*>(red + ((wv * 5 + j) * 64 + lane) * 4) = acc[j];
  __syncthreads();
  // wave j folds tile j (wave 0 also the bias tile)
  for (int j = wv; j < (with_b ? 5 : 4); j += 4) {
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(red + ((k * 5 + j) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    if (j < 4) {
      const int kc = k0 + 16 * j + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + 4 * lq + r;
        if (nn < N && kc < K) dw[(size_t)nn * lddw + kc] = s[r];
      }
    } else if (lr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + 4 * lq + r < N) db[n0 + 4 * lq + r] = s[r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ yv, int ldy, int act,
                                                    const float* __restrict__ x, int ldx, float* __restrict__ dw, int lddw, float* __restrict__ db,
                                                    int M, int K, int N) {
  __shared__ float red[4 * 5 * 64 * 4];
  wgrad_body(dy, lddy, yv, ldy, act, x, ldx, dw, lddw, db, M, K, N, red, blockIdx.x, blockIdx.y);
}

// Both gradients of a layer in ONE launch (they share dy and are independent): blocks [0, nd) are the data gradient's
// (gdx x gdy grid), the rest the weight gradient's.  These layers are launch / ramp bound (6-11 us per launch for < 0.1 GFLOP).
struct BwdArgs {
  const float* dy; const float* yv; const float* w; const float* x; float* dx; float* dw; float* db;
  int lddy, ldy, act, ldx, lddx, accumulate, M, K, N, gdx, nd, gwx;
};
__global__ __launch_bounds__(256) void bwd_kernel(const BwdArgs a) {
  __shared__ float red[4 * 5 * 64 * 4];
  const int b = blockIdx.x;
  if (b < a.nd) dgrad_body(a.dy, a.lddy, a.yv, a.ldy, a.act, a.w, a.K, a.dx, a.lddx, a.accumulate, a.M, a.K, a.N, red, b % a.gdx, b / a.gdx);
  else wgrad_body(a.dy, a.lddy, a.yv, a.ldy, a.act, a.x, a.ldx, a.dw, a.K, a.db, a.M, a.K, a.N, red, (b - a.nd) % a.gwx, (b - a.nd) / a.gwx);
}

// float4 operand loads need 16-byte aligned rows
inline bool aligned4(const void* p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 4 == 0; }
constexpr int MAX_ROWS = 512;

inline int run_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N, int act, hipStream_t s, const char* what) {
  if (M <= 0) return MLHOT_OK;
  ProfScope ps(what, s);
  hipLaunchKernelGGL(fwd_kernel, dim3((M + 31) / 32, (N + 15) / 16), dim3(256), 0, s, x, ldx, w, K, b, y, ldy, M, K, N, act);
  return check_launch(what);
}
inline int run_dgrad(const float* dy, int lddy, const float* y, int ldy, int act, const float* w, float* dx, int lddx, int accumulate, int M, int Kin, int N,
                     hipStream_t s, const char* what) {
  if (M <= 0) return MLHOT_OK;
  ProfScope ps(what, s);
  hipLaunchKernelGGL(dgrad_kernel, dim3((M + 31) / 32, (Kin + 15) / 16), dim3(256), 0, s, dy, lddy, y, ldy, act, w, Kin, dx, lddx, accumulate, M, Kin, N);
  return check_launch(what);
}
inline int run_wgrad(const float* dy, int lddy, const float* y, int ldy, int act, const float* x, int ldx, float* dw, float* db, int M, int K, int N,
                     hipStream_t s, const char* what) {
  ProfScope ps(what, s);
  hipLaunchKernelGGL(wgrad_kernel, dim3((N + 15) / 16, (K + 63) / 64), dim3(256), 0, s, dy, lddy, y, ldy, act, x, ldx, dw, K, db, M, K, N);
  return check_launch(what);
}

inline int run_bwd(const float* dy, int lddy, const float* y, int ldy, int act, const float* w, const float* x, int ldx, float* dx, int lddx,
                   int accumulate, float* dw, float* db, int M, int K, int N, hipStream_t s, const char* what) {
  if (M <= 0) return MLHOT_OK;
  BwdArgs a{dy, y, w, x, dx, dw, db, lddy, ldy, act, ldx, lddx, accumulate, M, K, N, (M + 31) / 32, 0, (N + 15) / 16};
  a.nd = a.gdx * ((K + 15) / 16);
  ProfScope ps(what, s);
  hipLaunchKernelGGL(bwd_kernel, dim3(a.nd + a.gwx * ((K + 63) / 64)), dim3(256), 0, s, a);
  return check_launch(what);
}

}  // namespace sk
}  // namespace mlhot
#endif
