// Chains of few-row Linear layers in ONE launch per direction, and several independent few-row Linear layers in one launch.
//
// The ResNet-family models (ANP.py:44-52,80-97, models.py:139-145,182-184; ANPMRShapeNet3D.py:135-183) run 11 Linear layers on
// T x N = 120 rows per step: task_encoder 260 -> 256 -> 256 -> 256, the K / V / Q head stacks 256 -> 2048, _W 2048 -> 256,
// mu 256 -> 256, fc_mu 512 -> 256 -> 256 -> 4, each with a torch.cat in front of the first layer of its chain.  As one launch per
// layer and direction (linear_skinny.h) that is 22 launches of 6-12 us for < 0.1 GFLOP each: launch- and latency-bound (c5: 170 us
// of a 1.56 ms step at mfma_util 0.03-0.05).  Here:
//   * chain_fwd_kernel: a workgroup owns 16 ROWS and walks up to 4 layers with the activations in LDS (two [16][516] buffers);
//     its 8 waves split a layer's output columns (two 16-column tiles each for N = 256), the weights stream from L2 as float4
//     in MFMA lane order (linear_skinny.h's k permutation), the A operand is one ds_read_b128 per 4 MFMAs.  A layer may take
//     extra input columns from a second tensor ("side": the labels behind the context features, the decoder's image features in
//     front of the sampled latent) - the reference's torch.cat costs nothing.  Every layer's output is also written to global
//     memory (the backward's ReLU masks and weight-gradient operands).
//   * chain_dgrad_kernel: the same walk backwards: G_k = dY_k . act'(y_k) (kept for the weight gradients), dX_k = G_k W_k, its
//     "previous" columns feed layer k - 1 through LDS, its side columns go to the side's gradient.
//   * multi_wgrad_kernel / multi_fwd_kernel / multi_bwd_kernel: job tables over linear_skinny.h's bodies - the weight + bias
//     gradients of all layers of a chain in one launch; the K / V / Q head stacks' forwards in one launch, their six gradient
//     bodies in one launch.
// MEASURED (MI355X, c5: 120 rows, round 4): the chain kernels are NOT faster than one launch per layer at 256-wide layers - 48 us
// for the four-layer decoder head and 48 us for its data-gradient walk, against 4 x 7.7 us as four launches.  A workgroup that owns
// 16 rows pulls every layer's whole weight matrix (256-512 KB) through ONE CU (~70 GB/s: 3.7-7.5 us per layer) and issues all of
// the layer's 1024-2048 MFMAs on that CU's matrix pipe (3.4-6.8 us), one after the other; splitting the columns over several CUs
// instead needs a cross-CU exchange of the activations per layer (release + flag + acquire: 4-5 us, MI355X_MICROARCH.md price
// list "handoff-flag" / "splitk-seam") - the price of the launch boundary it would remove.  The models therefore keep one launch
// per layer (MLHOT_MLP_CHAIN=1 switches the chains on for A/B runs); what stays in the default path is the job-table launches
// (the three head stacks per direction).  The chains pay where the weights are small against the rows (not these models).
// v_mfma_f32_16x16x4_f32: A lane l = A[l&15][l>>4], B lane l =
// B[l>>4][l&15], C/D lane l reg r = C[4*(l>>4)+r][l&15].
#pragma once
#include "common.h"
#include "linear_skinny.h"
#include "../../include/mlhot.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace mc {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
using sk::mfma4;
constexpr int MAXL = MLHOT_CHAIN_MAX_LAYERS, KMAX = 512, NMAX = 256, LDX = KMAX + 4, LDG = NMAX + 4, NT = 512;

struct Layer {
  const float* w; const float* b; const float* side; float* y; const float* yin;   // yin: y as the backward reads it
  int K, N, act, side_w, side_ld, side_first, ldy;
  // backward only
  float* g; int ldg; float* dside; int dside_ld, dside_acc;
  MLHOT_HD int off_prev() const { return side_first ? side_w : 0; }      // first column of the previous layer's output inside this layer's input
  MLHOT_HD int off_side() const { return side_first ? 0 : K - side_w; }
  MLHOT_HD int prev_w() const { return K - side_w; }
};
struct Args {
  const float* x0; int ldx0, M, n;
  Layer L[MAXL];
  const float* dy; int lddy;            // backward: gradient of the last layer's output
  float* dx0; int lddx0, dx0_acc;       // backward: gradient of x0 (null: not wanted)
};

// rows m0 .. m0+15 of an [M][w] global tensor -> LDS columns [off, off + w) of a [16][LDX] buffer (zeros past M); w % 4 == 0
__device__ __forceinline__ void stage_cols(float* buf, int off, const float* __restrict__ src, int ld, int w, int m0, int M, int tid) {
  const int w4 = w >> 2;                      // <= 128: at most 4 float4 per thread, all requested before the first LDS store
  float4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + NT * j, r = e / w4, c = 4 * (e - r * w4);
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < 16 * w4 && m0 + r < M) v[j] = *reinterpret_cast<const float4*>(src + (size_t)(m0 + r) * ld + c);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = tid + NT * j, r = e / w4, c = 4 * (e - r * w4);
    if (e < 16 * w4) {
      float* d = buf + r * LDX + off + c;
      d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
    }
  }
}

__global__ __launch_bounds__(NT) void chain_fwd_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float buf[2][16 * LDX];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 16, M = a.M;
  {
    const Layer& l0 = a.L[0];
    stage_cols(buf[0], l0.off_prev(), a.x0, a.ldx0, l0.prev_w(), m0, M, tid);
    if (l0.side_w) stage_cols(buf[0], l0.off_side(), l0.side, l0.side_ld, l0.side_w, m0, M, tid);
  }
  __syncthreads();
  for (int k = 0; k < a.n; ++k) {
    const Layer& l = a.L[k];
    const float* in = buf[k & 1];
    float* out = buf[(k + 1) & 1];
    const bool last = k + 1 == a.n;
    // the next layer's side columns: its buffer is free from here on (its last readers were layer k - 1's MFMAs)
    if (!last && a.L[k + 1].side_w) stage_cols(out, a.L[k + 1].off_side(), a.L[k + 1].side, a.L[k + 1].side_ld, a.L[k + 1].side_w, m0, M, tid);
    const int K = l.K, N = l.N, ntile = (N + 15) >> 4;
    const int t0 = wv, t1 = wv + 8;
    const bool has0 = t0 < ntile, has1 = t1 < ntile;
    if (has0) {
      const int n0r = 16 * t0 + lr, n1r = 16 * t1 + lr;
      // no per-lane load masks: rows past N and k past K are clamped to valid addresses - their products are discarded by the
      // epilogue (n >= N) or multiplied by the zeroed A operand (k >= K)
      const float* w0 = l.w + (size_t)(n0r < N ? n0r : N - 1) * K;
      const float* w1 = l.w + (size_t)((has1 && n1r < N) ? n1r : N - 1) * K;
      const float* ap = in + lr * LDX + 4 * lq;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      // a pass = 256 k: ALL its 32 weight float4 (128 VGPRs) are requested before the first MFMA - one L2 round trip per pass
      // (as 64-k trips every trip waited out its own: 45 us for a four-layer chain)
      for (int kb = 0; kb < K; kb += 256) {
        float4 b0[16], b1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k0 = kb + 16 * u;
          if (k0 < K) {                                     // wave-uniform
            const int kk = k0 + 4 * lq < K ? k0 + 4 * lq : K - 4;
            b0[u] = *reinterpret_cast<const float4*>(w0 + kk);
            b1[u] = *reinterpret_cast<const float4*>(w1 + kk);
          }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k0 = kb + 16 * u;
          if (k0 < K) {                                     // wave-uniform
            const f32x4_t av = (k0 + 4 * lq < K) ? *reinterpret_cast<const f32x4_t*>(ap + k0) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            acc0 = mfma4(av[0], b0[u].x, acc0); acc1 = mfma4(av[0], b1[u].x, acc1);
            acc0 = mfma4(av[1], b0[u].y, acc0); acc1 = mfma4(av[1], b1[u].y, acc1);
            acc0 = mfma4(av[2], b0[u].z, acc0); acc1 = mfma4(av[2], b1[u].z, acc1);
            acc0 = mfma4(av[3], b0[u].w, acc0); acc1 = mfma4(av[3], b1[u].w, acc1);
          }
        }
      }
      const int poff = last ? 0 : a.L[k + 1].off_prev();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = (t ? n1r : n0r);
        if ((t && !has1) || n >= N) continue;
        const float bn = l.b ? l.b[n] : 0.f;
        const f32x4_t acc = t ? acc1 : acc0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          const float v = act_apply(l.act, acc[r] + bn);
          if (!last) out[row * LDX + poff + n] = v;
          if (m0 + row < M) l.y[(size_t)(m0 + row) * l.ldy + n] = v;
        }
      }
    }
    __syncthreads();
  }
}

// Backward walk.  LDS: gbuf [16][LDG] = G_k (A operand of dX_k = G_k W_k), dbuf [16][LDX] = dY of the layer below.
__global__ __launch_bounds__(NT) void chain_dgrad_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) float gbuf[16 * LDG];
  __shared__ __attribute__((aligned(16))) float dbuf[16 * LDX];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int m0 = blockIdx.x * 16, M = a.M;
  for (int k = a.n - 1; k >= 0; --k) {
    const Layer& l = a.L[k];
    const int K = l.K, N = l.N;
    // G_k = dY_k . act'(y_k): rows of this tile, all N columns; to LDS (zero past N up to a multiple of 16) and to global
    const int N16 = (N + 15) & ~15;
    {
      float dyv[8], yv[8];                    // <= 8 elements per thread; every global load requested before the first store
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = tid + NT * j, r = e / N16, c = e - r * N16;
        const bool ok = e < 16 * N16 && c < N && m0 + r < M;
        dyv[j] = 0.f; yv[j] = 0.f;
        if (ok) {
          dyv[j] = (k == a.n - 1) ? a.dy[(size_t)(m0 + r) * a.lddy + c] : dbuf[r * LDX + c];
          if (l.act != ACT_NONE) yv[j] = l.yin[(size_t)(m0 + r) * l.ldy + c];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = tid + NT * j, r = e / N16, c = e - r * N16;
        if (e < 16 * N16) {
          const bool ok = c < N && m0 + r < M;
          const float g = !ok ? 0.f : (l.act == ACT_NONE ? dyv[j] : dyv[j] * sk::dact(l.act, yv[j]));
          if (ok) l.g[(size_t)(m0 + r) * l.ldg + c] = g;
          gbuf[r * LDG + c] = g;
        }
      }
    }
    __syncthreads();
    const bool want_prev = k > 0 || a.dx0 != nullptr, want_side = l.side_w > 0 && l.dside != nullptr;
    if (!want_prev && !want_side) break;
    // dX[16][K] = G[16][N] W[N][K]: column tiles of K over the waves; B[kk = n][col] = W[n][col] (dword loads, contiguous over lr)
    const int ktile = (K + 15) >> 4;
    const int op = l.off_prev(), os = l.off_side(), pw = l.prev_w();
    // dX element (row, col): previous-layer columns -> dbuf (or dx0 for layer 0), side columns -> the side's gradient
    auto route = [&](int row, int col, float v) {
      if (col >= op && col < op + pw) {
        if (k > 0) dbuf[row * LDX + col - op] = v;
        else if (a.dx0 != nullptr && m0 + row < M) {
          float* d = a.dx0 + (size_t)(m0 + row) * a.lddx0 + col - op;
          *d = a.dx0_acc ? *d + v : v;
        }
      } else if (want_side && m0 + row < M) {
        float* d = l.dside + (size_t)(m0 + row) * l.dside_ld + col - os;
        *d = l.dside_acc ? *d + v : v;
      }
    };
    if (N & 15) {
      // a narrow layer (the model's output: 1 .. 4 columns, or any N that is not a multiple of 16): N multiply-adds per element
      for (int e = tid; e < 16 * K; e += NT) {
        const int row = e / K, col = e - row * K;
        float sacc = 0.f;
        for (int n = 0; n < N; ++n) sacc = fmaf(gbuf[row * LDG + n], l.w[(size_t)n * K + col], sacc);
        route(row, col, sacc);
      }
    } else {
      // a pass = two column tiles per wave (cp + wv, cp + wv + 8); ALL their weight words (2 x 16 chunks x 4 dwords = 128 VGPRs) are
      // requested before the first MFMA: one L2 round trip per pass (in 32-deep trips every trip waited out its own: 64 us per
      // chain).  Addressing: ONE 32-bit lane offset per tile (row 4 lq, the tile's column; columns past K clamped - their results
      // are never stored) on top of a wave-uniform row base per (chunk, i): 128 loads in flight on two address registers.
      for (int cp = 0; cp < ktile; cp += 16) {
        const int ct0 = cp + wv, ct1 = cp + wv + 8;
        if (ct0 >= ktile) break;                              // wave-uniform; no barrier inside the pass loop
        const int c0 = 16 * ct0 + lr, c1 = 16 * ct1 + lr;
        const bool in0 = c0 < K, in1 = ct1 < ktile && c1 < K;
        const unsigned lo0 = (unsigned)(4 * lq * K + (in0 ? c0 : K - 1)), lo1 = (unsigned)(4 * lq * K + (in1 ? c1 : K - 1));
        float wq0[16][4], wq1[16][4];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (16 * u < N) {                                   // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float* sb = l.w + (size_t)(16 * u + i) * K;
              wq0[u][i] = sb[lo0];
              wq1[u][i] = sb[lo1];
            }
          }
        }
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (16 * u < N) {                                   // wave-uniform
            const f32x4_t gv = *reinterpret_cast<const f32x4_t*>(gbuf + lr * LDG + 16 * u + 4 * lq);
            acc0 = mfma4(gv[0], wq0[u][0], acc0); acc1 = mfma4(gv[0], wq1[u][0], acc1);
            acc0 = mfma4(gv[1], wq0[u][1], acc0); acc1 = mfma4(gv[1], wq1[u][1], acc1);
            acc0 = mfma4(gv[2], wq0[u][2], acc0); acc1 = mfma4(gv[2], wq1[u][2], acc1);
            acc0 = mfma4(gv[3], wq0[u][3], acc0); acc1 = mfma4(gv[3], wq1[u][3], acc1);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (in0) route(4 * lq + r, c0, acc0[r]);
          if (in1) route(4 * lq + r, c1, acc1[r]);
        }
      }
    }
    __syncthreads();
  }
}

// ---- job tables over linear_skinny.h's bodies ------------------------------------------------------------------------
constexpr int MAXJ = 8;
struct WJob { const float* dy; const float* x; float* dw; float* db; int lddy, ldx, lddw, M, K, N, gx, first; };   // blocks [first, next first)
struct WJobs { WJob j[MAXJ]; int n, total; };
__global__ __launch_bounds__(256) void multi_wgrad_kernel(const WJobs t) {
  __shared__ float red[4 * 5 * 64 * 4];
  int ji = 0;
  while (ji + 1 < t.n && (int)blockIdx.x >= t.j[ji + 1].first) ++ji;
  const WJob& j = t.j[ji];
  const int b = (int)blockIdx.x - j.first;
  sk::wgrad_body(j.dy, j.lddy, nullptr, 0, ACT_NONE, j.x, j.ldx, j.dw, j.lddw, j.db, j.M, j.K, j.N, red, b % j.gx, b / j.gx);
}

struct LJob {       // one Linear layer: forward, or both gradients; x2 / dx2: the columns [K1, K) of a two-source input (torch.cat folded)
  const float* x; const float* w; const float* b; float* y; const float* dy; float* dx; float* dw; float* db;
  const float* x2; float* dx2;
  int ldx, ldy, lddy, lddx, M, K, N, act, dx_acc, gdx, nd, gwx, first, ldx2, lddx2, K1;
};
struct LJobs { LJob j[MAXJ]; int n, total; };
__global__ __launch_bounds__(256) void multi_fwd_kernel(const LJobs t) {
  __shared__ float red[4 * 2 * 64 * 4];
  int ji = 0;
  while (ji + 1 < t.n && (int)blockIdx.x >= t.j[ji + 1].first) ++ji;
  const LJob& j = t.j[ji];
  const int b = (int)blockIdx.x - j.first;
  sk::fwd_body(j.x, j.ldx, j.w, j.K, j.b, j.y, j.ldy, j.M, j.K, j.N, j.act, red, b % j.gdx, b / j.gdx, j.x2, j.ldx2, j.K1);
}
__global__ __launch_bounds__(256) void multi_bwd_kernel(const LJobs t) {
  __shared__ float red[4 * 5 * 64 * 4];
  int ji = 0;
  while (ji + 1 < t.n && (int)blockIdx.x >= t.j[ji + 1].first) ++ji;
  const LJob& j = t.j[ji];
  const int b = (int)blockIdx.x - j.first;
  if (b < j.nd) sk::dgrad_body(j.dy, j.lddy, j.y, j.ldy, j.act, j.w, j.K, j.dx, j.lddx, j.dx_acc, j.M, j.K, j.N, red, b % j.gdx, b / j.gdx, j.dx2, j.lddx2, j.K1);
  else sk::wgrad_body(j.dy, j.lddy, j.y, j.ldy, j.act, j.x, j.ldx, j.dw, j.K, j.db, j.M, j.K, j.N, red, (b - j.nd) % j.gwx, (b - j.nd) / j.gwx, j.x2, j.ldx2, j.K1);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int check_chain(const float* x0, int ldx0, int M, const mlhot_chain_layer* L, int n, const char* what) {
  if (!x0 || !L || n < 1 || n > MAXL || M < 0 || M > sk::MAX_ROWS || !a16(x0) || ldx0 % 4) { set_error("%s: bad argument", what); return MLHOT_ERR_ARG; }
  int prev = -1;
  for (int k = 0; k < n; ++k) {
    const mlhot_chain_layer& l = L[k];
    const int pw = l.K - l.side_w;
    if (!l.w || !l.y || l.K < 4 || l.K > KMAX || l.K % 4 || l.N < 1 || l.N > NMAX || l.act < 0 || l.act > 2 || l.side_w < 0 || l.side_w % 4 || pw < 4 ||
        pw % 4 || (l.side_w && (!l.side || !a16(l.side) || l.side_ld % 4)) || !a16(l.w) || l.ldy < l.N) {
      set_error("%s: layer %d: needs 4 <= K <= %d (K, side_w %% 4 == 0), N <= %d, 16-byte aligned operands", what, k, KMAX, NMAX);
      return MLHOT_ERR_ARG;
    }
    if (k > 0 && pw != prev) { set_error("%s: layer %d takes %d columns from layer %d, which has %d", what, k, pw, k - 1, prev); return MLHOT_ERR_ARG; }
    if (k == 0 && ldx0 < pw) { set_error("%s: x0 rows are shorter than layer 0's input", what); return MLHOT_ERR_ARG; }
    if (k + 1 < n && l.N % 4) { set_error("%s: layer %d: an inner layer needs N %% 4 == 0", what, k); return MLHOT_ERR_ARG; }
    prev = l.N;
  }
  return MLHOT_OK;
}

inline void fill_args(Args& a, const float* x0, int ldx0, int M, const mlhot_chain_layer* L, int n) {
  a = Args{};
  a.x0 = x0; a.ldx0 = ldx0; a.M = M; a.n = n;
  for (int k = 0; k < n; ++k) {
    Layer& d = a.L[k];
    d.w = L[k].w; d.b = L[k].b; d.side = L[k].side; d.y = L[k].y; d.yin = L[k].y;
    d.K = L[k].K; d.N = L[k].N; d.act = L[k].act; d.side_w = L[k].side_w; d.side_ld = L[k].side_ld; d.side_first = L[k].side_first; d.ldy = L[k].ldy;
  }
}

inline int chain_forward(const float* x0, int ldx0, int M, const mlhot_chain_layer* L, int n, hipStream_t s) {
  MLHOT_TRY(check_chain(x0, ldx0, M, L, n, "mlp_chain_fwd"));
  if (M == 0) return MLHOT_OK;
  Args a;
  fill_args(a, x0, ldx0, M, L, n);
  {
    ProfScope ps("chain.fwd", s);
    hipLaunchKernelGGL(chain_fwd_kernel, dim3((M + 15) / 16), dim3(NT), 0, s, a);
  }
  return check_launch("mlp_chain_fwd");
}

inline int chain_backward(const float* x0, int ldx0, int M, const mlhot_chain_layer* L, const mlhot_chain_grads* G, int n, const float* dy, int lddy,
                          float* dx0, int lddx0, int dx0_acc, hipStream_t s) {
  MLHOT_TRY(check_chain(x0, ldx0, M, L, n, "mlp_chain_bwd"));
  if (!G || !dy || lddy < L[n - 1].N) { set_error("mlp_chain_bwd: bad argument"); return MLHOT_ERR_ARG; }
  if (M == 0) return MLHOT_OK;
  Args a;
  fill_args(a, x0, ldx0, M, L, n);
  a.dy = dy; a.lddy = lddy; a.dx0 = dx0; a.lddx0 = lddx0; a.dx0_acc = dx0_acc;
  WJobs jobs{};
  auto add = [&](const float* g, int ldg, const float* x, int ldx, float* dw, int lddw, float* db, int K, int N) {
    WJob& j = jobs.j[jobs.n];
    j = WJob{g, x, dw, db, ldg, ldx, lddw, M, K, N, (N + 15) / 16, jobs.total};
    jobs.total += j.gx * ((K + 63) / 64);
    ++jobs.n;
  };
  for (int k = 0; k < n; ++k) {
    const mlhot_chain_grads& gk = G[k];
    if (!gk.g || !gk.dw || gk.ldg < L[k].N || gk.ldg % 4 || !a16(gk.g)) { set_error("mlp_chain_bwd: layer %d: needs g (ldg %% 4 == 0, >= N) and dw", k); return MLHOT_ERR_ARG; }
    Layer& d = a.L[k];
    d.g = gk.g; d.ldg = gk.ldg; d.dside = gk.dside; d.dside_ld = gk.dside_ld; d.dside_acc = gk.dside_accumulate;
    // weight gradient: the previous-layer columns and the side columns of dW are two jobs (two operand tensors)
    const float* xin = k == 0 ? x0 : L[k - 1].y;
    const int ldin = k == 0 ? ldx0 : L[k - 1].ldy;
    add(gk.g, gk.ldg, xin, ldin, gk.dw + d.off_prev(), d.K, gk.db, d.prev_w(), d.N);
    if (d.side_w) add(gk.g, gk.ldg, d.side, d.side_ld, gk.dw + d.off_side(), d.K, nullptr, d.side_w, d.N);
  }
  {
    ProfScope ps("chain.bwd.dgrad", s);
    hipLaunchKernelGGL(chain_dgrad_kernel, dim3((M + 15) / 16), dim3(NT), 0, s, a);
  }
  MLHOT_TRY(check_launch("mlp_chain_bwd (dgrad)"));
  {
    ProfScope ps("chain.bwd.wgrad", s);
    hipLaunchKernelGGL(multi_wgrad_kernel, dim3(jobs.total), dim3(256), 0, s, jobs);
  }
  return check_launch("mlp_chain_bwd (wgrad)");
}

inline int check_multi(const mlhot_linear_job* J, int n, bool bwd, const char* what) {
  if (!J || n < 1 || n > MAXJ) { set_error("%s: 1 .. %d jobs", what, MAXJ); return MLHOT_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    const mlhot_linear_job& j = J[i];
    bool ok = j.x && j.w && j.y && j.M >= 0 && j.M <= sk::MAX_ROWS && j.K >= 4 && j.K % 4 == 0 && j.N >= 1 && j.act >= 0 && j.act <= 2 &&
              sk::aligned4(j.x, j.ldx) && a16(j.w);
    if (j.x2) ok = ok && j.K2 >= 4 && j.K2 % 4 == 0 && j.K2 < j.K && sk::aligned4(j.x2, j.ldx2);
    if (bwd) ok = ok && j.dy && (j.dx || (j.x2 && j.dx2)) && j.dw && j.N % 4 == 0 && sk::aligned4(j.dy, j.lddy) && (j.act == ACT_NONE || sk::aligned4(j.y, j.ldy)) &&
                  (!j.dx || sk::aligned4(j.dx, j.lddx)) && (!j.dx2 || sk::aligned4(j.dx2, j.lddx2));
    if (!ok) { set_error("%s: job %d: few-row Linear jobs need M <= %d, K %% 4 == 0, 16-byte aligned rows%s", what, i, sk::MAX_ROWS, bwd ? ", N % 4 == 0, dx and dw" : ""); return MLHOT_ERR_ARG; }
  }
  return MLHOT_OK;
}
inline void fill_jobs(LJobs& t, const mlhot_linear_job* J, int n, bool bwd) {
  t = LJobs{};
  for (int i = 0; i < n; ++i) {
    const mlhot_linear_job& s = J[i];
    LJob& j = t.j[t.n];
    j = LJob{s.x, s.w, s.b, s.y, s.dy, s.dx, s.dw, s.db, s.x2, s.dx2, s.ldx, s.ldy, s.lddy, s.lddx, s.M, s.K, s.N, s.act, s.dx_accumulate, (s.M + 31) / 32, 0,
             (s.N + 15) / 16, t.total, s.ldx2, s.lddx2, s.x2 ? s.K - s.K2 : (1 << 30)};
    if (s.M == 0) continue;
    if (!bwd) t.total += j.gdx * ((s.N + 15) / 16);
    else { j.nd = j.gdx * ((s.K + 15) / 16); t.total += j.nd + j.gwx * ((s.K + 63) / 64); }
    ++t.n;
  }
}
inline int multi_forward(const mlhot_linear_job* J, int n, hipStream_t s) {
  MLHOT_TRY(check_multi(J, n, false, "linear_multi_fwd"));
  LJobs t;
  fill_jobs(t, J, n, false);
  if (t.total == 0) return MLHOT_OK;
  {
    ProfScope ps("linear_multi.fwd", s);
    hipLaunchKernelGGL(multi_fwd_kernel, dim3(t.total), dim3(256), 0, s, t);
  }
  return check_launch("linear_multi_fwd");
}
inline int multi_backward(const mlhot_linear_job* J, int n, hipStream_t s) {
  MLHOT_TRY(check_multi(J, n, true, "linear_multi_bwd"));
  LJobs t;
  fill_jobs(t, J, n, true);
  if (t.total == 0) return MLHOT_OK;
  {
    ProfScope ps("linear_multi.bwd", s);
    hipLaunchKernelGGL(multi_bwd_kernel, dim3(t.total), dim3(256), 0, s, t);
  }
  return check_launch("linear_multi_bwd");
}

}  // namespace mc
}  // namespace mlhot
#endif
