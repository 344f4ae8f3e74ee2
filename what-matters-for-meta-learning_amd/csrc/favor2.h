// FAVOR+ attention (networks/fast_attention.py:74-99,151-156) in TWO launches per direction, for any head width d (64 or 256)
// and up to 32 context / 32 target shots.  Same arithmetic as favor.h (which stays the checked route for larger shot counts):
//   F1  grid (task x head, feature chunk of 256): dd = (c x) P^T for the block's query AND key rows on the matrix core, operands
//       straight from global memory in MFMA lane order; partial row maxima (+ first arg-max) and the workgroup's key maximum.
//   F2  grid (task x head): row / batch-global stabilisers from the partials, E = ratio exp(dd - diag - stab) (kept for the
//       backward), S = F_q F_k^T (K = m features split over the four waves), D = rowsum(S), out = S V / D in the merged order.
//   B1  grid (task x head): w = <dO, O - c>, dS = (dO (V - c)^T - w) / D with c = the block's first value row, dV, G_q = (dS F_k) . E_q, G_k = (dS^T F_q) . E_k and their row sums.
//   B2  grid (task x head, 64-wide slice of d): dx = c (G - stabiliser corrections) P - rowsum c^2 x for query and key rows.
// Feature buffers use a row stride mp = m rounded up to 16 (float4 operand loads); features >= m are written as zeros.
// v_mfma_f32_16x16x4_f32: A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], C/D lane l reg r = C[4*(l>>4)+r][l&15]; a float4
// of 4 consecutive k per lane feeds 4 MFMAs whose lane-group lq covers k0 + 4 lq + i - the same permutation on both operands.
#pragma once
#include "common.h"
#include "favor.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
#ifdef MLHOT_TS
namespace tf { extern __device__ long long* g_ts_dev; }
#endif
namespace fv {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int F1_TPW = 3;         // 16-feature tiles per F1 wave
constexpr int FCH = 4 * 16 * F1_TPW;   // features per F1 workgroup (4 waves x 3 tiles of 16 = 192: m = 1419 -> 8 chunks x 64 (task, head) blocks =
                                       // 512 workgroups, two per CU; with 256 features it was 384 - half the CUs carried two, and the kernel ran at their pace)
constexpr int MAXN = 32;          // shots per side
constexpr int NSP = 8;            // B1 workgroups per (task, head): each takes an eighth of the feature tiles (512 workgroups, two per CU)
__device__ __forceinline__ float sum_splits(const float* p) {      // the NSP partial row sums of one row, in a fixed order
  return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}
static_assert(NSP == 8, "sum_splits");
constexpr int NPMAX = 32;         // partial row maxima per query row (4 per F1 feature chunk): m <= 2048

struct Ws {     // carved from the caller's workspace; the forward fills it, the backward reuses it
  float *eq, *ek;                 // dd, then E = ratio exp(...) in place: [rows_q][mp], [rows_k][mp]
  float *pm_v; int* pm_i;         // partial row maxima of the query rows [rows_q][npart]
  float* wg_v; int *wg_row, *wg_j;   // per F1 workgroup: maximum over its key rows / features, and where
  float *mx_q; int* arg_q;        // query row maximum / arg-max
  float* gmax; int* gpos;         // batch-global key maximum, {key row, feature}
  float *diag;                    // [rows_q + rows_k] (block order: a block's query rows, then its key rows)
  float *S, *D;                   // [T H][Nq][Nc], [T H][Nq]
  float *gq, *gk, *rs_q, *rs_k;   // backward: G; per-row sums of G, one partial per B1 split: [rows][NSP]
  float* gt_fix;                  // staged backward only (stab_xchg.h): the other ranks' share of the stabiliser gradient
  int mp, nch, npart;
  bool ok;
};
inline Ws carve(const FavorDims& f, void* ws, size_t bytes, size_t* need = nullptr) {
  Arena a(ws, bytes);
  Ws w{};
  w.mp = (f.m + 15) / 16 * 16; w.nch = (f.m + FCH - 1) / FCH; w.npart = w.nch * 4;
  const size_t rq = f.rows_q(), rk = f.rows_k(), th = (size_t)f.T * f.H;
  w.eq = a.take<float>(rq * w.mp); w.ek = a.take<float>(rk * w.mp);
  w.pm_v = a.take<float>(rq * w.npart); w.pm_i = a.take<int>(rq * w.npart);
  w.wg_v = a.take<float>(th * w.nch); w.wg_row = a.take<int>(th * w.nch); w.wg_j = a.take<int>(th * w.nch);
  w.mx_q = a.take<float>(rq); w.arg_q = a.take<int>(rq);
  w.gmax = a.take<float>(4); w.gpos = a.take<int>(4);
  w.diag = a.take<float>(rq + rk);
  w.S = a.take<float>(th * f.Nq * f.Nc); w.D = a.take<float>(th * f.Nq);
  w.gq = a.take<float>(rq * w.mp); w.gk = a.take<float>(rk * w.mp);
  w.rs_q = a.take<float>(rq * NSP); w.rs_k = a.take<float>(rk * NSP);
  w.gt_fix = a.take<float>(4);
  w.ok = a.ok;
  if (need) *need = a.off + 256;
  return w;
}
inline bool applies(const FavorDims& f) { return f.Nq <= MAXN && f.Nc <= MAXN && f.d % 16 == 0 && f.d <= 256 && f.m >= 16 && f.m <= FCH * NPMAX / 4; }

struct Args {
  FavorDims f; Ws w;
  const float *q, *k, *v, *proj;
  float* out; const float* dout;
  float *dq, *dk, *dv;
  float c, ratio, re;            // d^-1/4, m^-1/2, ratio * 1e-4
  const float* gt_fix;           // null unless the backward is staged
};

// row r of block (t, h): r < Nq -> query row, else key row; returns nullptr past the block
__device__ __forceinline__ const float* block_row(const Args& a, int t, int h, int r) {
  if (r < a.f.Nq) return a.q + ((size_t)(t * a.f.Nq + r) * a.f.H + h) * a.f.d;
  if (r < a.f.Nq + a.f.Nc) return a.k + ((size_t)(t * a.f.Nc + r - a.f.Nq) * a.f.H + h) * a.f.d;
  return nullptr;
}

// ---- F1 ---------------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void f1_kernel(const Args a) {
  __shared__ float sm_v[256]; __shared__ int sm_r[256], sm_j[256];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int th = blockIdx.x, t = th / a.f.H, h = th % a.f.H, chunk = blockIdx.y;
  const int d = a.f.d, m = a.f.m, Nq = a.f.Nq, R = Nq + a.f.Nc, mp = a.w.mp;
  const int f0 = chunk * FCH + wv * (16 * F1_TPW);
#ifdef MLHOT_TS
  const int ts_wg = blockIdx.y * gridDim.x + blockIdx.x;
#define F1_TS(k) do { if (tf::g_ts_dev && tid == 0 && ts_wg < 512) tf::g_ts_dev[1024 + 4 * ts_wg + (k)] = wall_clock64(); } while (0)
#else
#define F1_TS(k) do { } while (0)
#endif
  F1_TS(0);
  const float* xr[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) xr[i] = block_row(a, t, h, 16 * i + lr);
  const float* pr[F1_TPW];
#pragma unroll
  for (int j = 0; j < F1_TPW; ++j) { const int fj = f0 + 16 * j + lr; pr[j] = fj < m ? a.proj + (size_t)fj * d + 4 * lq : nullptr; }
  f32x4_t acc[RT][F1_TPW];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < F1_TPW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  constexpr int UF = RT <= 2 ? 4 : 2;           // 16-deep chunks per trip: all their operand loads are in flight together
  for (int c0 = 0; c0 < d; c0 += 16 * UF) {
    float4 av[UF][RT], bv[UF][F1_TPW];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int k0 = c0 + 16 * u;
      const bool kin = k0 < d;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        av[u][i] = (kin && xr[i]) ? *reinterpret_cast<const float4*>(xr[i] + k0 + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
        av[u][i].x *= a.c; av[u][i].y *= a.c; av[u][i].z *= a.c; av[u][i].w *= a.c;       // data_normalizer * data, then the product
      }
#pragma unroll
      for (int j = 0; j < F1_TPW; ++j) bv[u][j] = (kin && pr[j]) ? *reinterpret_cast<const float4*>(pr[j] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < UF; ++u)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < F1_TPW; ++j) {
          acc[i][j] = mfma4(av[u][i].x, bv[u][j].x, acc[i][j]); acc[i][j] = mfma4(av[u][i].y, bv[u][j].y, acc[i][j]);
          acc[i][j] = mfma4(av[u][i].z, bv[u][j].z, acc[i][j]); acc[i][j] = mfma4(av[u][i].w, bv[u][j].w, acc[i][j]);
        }
  }
  F1_TS(1);
  // store dd; partial maxima per row over this wave's 48 features (first position wins ties)
  float kbest = -INFINITY; int kbr = 0x7fffffff, kbj = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * i + 4 * lq + r;
      float best = -INFINITY; int bj = 0x7fffffff;
      const bool isq = row < Nq, valid = row < R;
      float* drow = nullptr;
      if (valid) drow = isq ? a.w.eq + ((size_t)(t * Nq + row) * a.f.H + h) * mp : a.w.ek + ((size_t)(t * a.f.Nc + row - Nq) * a.f.H + h) * mp;
#pragma unroll
      for (int j = 0; j < F1_TPW; ++j) {
        const int fj = f0 + 16 * j + lr;
        const float vv = acc[i][j][r];
        if (valid && fj < mp) drow[fj] = fj < m ? vv : 0.f;
        if (fj < m && vv > best) { best = vv; bj = fj; }
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        const float ov = __shfl_xor(best, off, 64); const int oj = __shfl_xor(bj, off, 64);
        if (ov > best || (ov == best && oj < bj)) { best = ov; bj = oj; }
      }
      if (valid && lr == 0) {
        if (isq) {
          const size_t grow = (size_t)(t * Nq + row) * a.f.H + h;
          a.w.pm_v[grow * a.w.npart + chunk * 4 + wv] = best; a.w.pm_i[grow * a.w.npart + chunk * 4 + wv] = bj;
        } else {
          const int grow = (t * a.f.Nc + row - Nq) * a.f.H + h;
          if (best > kbest || (best == kbest && (grow < kbr || (grow == kbr && bj < kbj)))) { kbest = best; kbr = grow; kbj = bj; }
        }
      }
    }
  F1_TS(2);
  sm_v[tid] = kbest; sm_r[tid] = kbr; sm_j[tid] = kbj;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      const float ov = sm_v[tid + s]; const int orow = sm_r[tid + s], oj = sm_j[tid + s];
      if (ov > sm_v[tid] || (ov == sm_v[tid] && (orow < sm_r[tid] || (orow == sm_r[tid] && oj < sm_j[tid])))) { sm_v[tid] = ov; sm_r[tid] = orow; sm_j[tid] = oj; }
    }
    __syncthreads();
  }
  if (tid == 0) { const int o = th * a.w.nch + chunk; a.w.wg_v[o] = sm_v[0]; a.w.wg_row[o] = sm_r[0]; a.w.wg_j[o] = sm_j[0]; }
  F1_TS(3);
}

// ---- F2 ---------------------------------------------------------------------------------------------------------
// 1024 threads: ONE workgroup per (task, head) - 64 of them on 256 CUs - so the block's 30 x 1424 exponentials and their dd / E
// traffic are spread over 16 waves instead of 4 (the grid cannot grow: S = F_q F_k^T needs every feature of the block).
constexpr int F2_NT = 1024, F2_NW = F2_NT / 64;
#ifdef MLHOT_TS
#define F2_TS(k) do { if (tf::g_ts_dev && threadIdx.x == 0 && blockIdx.x < 64) tf::g_ts_dev[3200 + 8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define F2_TS(k) do { } while (0)
#endif
__global__ __launch_bounds__(F2_NT) void f2_kernel(const Args a) {
  __shared__ float s_mx[MAXN], s_diag[2 * MAXN], s_S[MAXN * (MAXN + 1)], s_D[MAXN], red[F2_NW * 4 * 64 * 4];
  __shared__ float s_g; __shared__ float sm_v[F2_NW]; __shared__ int sm_r[F2_NW], sm_j[F2_NW];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int th = blockIdx.x, t = th / a.f.H, h = th % a.f.H;
  const int d = a.f.d, m = a.f.m, Nq = a.f.Nq, Nc = a.f.Nc, H = a.f.H, mp = a.w.mp;
  F2_TS(0);
  // query row maxima from the partials
  if (tid < Nq) {
    const size_t grow = (size_t)(t * Nq + tid) * H + h;
    // all partials requested before the first compare (as a loop with a run-time trip count the 24 value / index pairs were 24
    // serial L2 round trips: ~10 us of this kernel's 48)
    float pv[NPMAX]; int pj[NPMAX];
#pragma unroll
    for (int p = 0; p < NPMAX; ++p) {
      pv[p] = -INFINITY; pj[p] = 0x7fffffff;
      if (p < a.w.npart) { pv[p] = a.w.pm_v[grow * a.w.npart + p]; pj[p] = a.w.pm_i[grow * a.w.npart + p]; }
    }
    float best = -INFINITY; int bj = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < NPMAX; ++p)
      if (pv[p] > best || (pv[p] == best && pj[p] < bj)) { best = pv[p]; bj = pj[p]; }
    s_mx[tid] = best; a.w.mx_q[grow] = best; a.w.arg_q[grow] = bj;
  }
  // batch-global key maximum from the per-workgroup maxima
  {
    float best = -INFINITY; int br = 0x7fffffff, bj = 0x7fffffff;
    for (int i = tid; i < a.f.T * H * a.w.nch; i += F2_NT) {
      const float vv = a.w.wg_v[i]; const int r = a.w.wg_row[i], j = a.w.wg_j[i];
      if (vv > best || (vv == best && (r < br || (r == br && j < bj)))) { best = vv; br = r; bj = j; }
    }
    // lanes by shuffles, then the sixteen wave results (a ten-level tree through LDS was ten barriers of 1024 threads); the order
    // (value, then row, then feature) is total, so the result does not depend on the shape of the tree
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64); const int orow = __shfl_xor(br, off, 64), oj = __shfl_xor(bj, off, 64);
      if (ov > best || (ov == best && (orow < br || (orow == br && oj < bj)))) { best = ov; br = orow; bj = oj; }
    }
    if (lane == 0) { sm_v[wv] = best; sm_r[wv] = br; sm_j[wv] = bj; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
      for (int k = 1; k < F2_NW; ++k) {
        const float ov = sm_v[k]; const int orow = sm_r[k], oj = sm_j[k];
        if (ov > best || (ov == best && (orow < br || (orow == br && oj < bj)))) { best = ov; br = orow; bj = oj; }
      }
      s_g = best;
      if (th == 0) { a.w.gmax[0] = best; a.w.gpos[0] = br; a.w.gpos[1] = bj; }
    }
  }
  // diag = 0.5 c^2 |x|^2 of the block's rows: 4 threads per row (the first 256 threads)
  {
    const int r = tid >> 2, part = tid & 3;
    float s = 0.f;
    const float* xp = r < 2 * MAXN ? block_row(a, t, h, r) : nullptr;
    if (xp) {
      // d <= 256 (applies()): the row's 16 float4 of this thread requested together (unrolled by 4 it was four serial round trips)
      float4 u[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) { const int e = 4 * part + 16 * k; u[k] = e < d ? *reinterpret_cast<const float4*>(xp + e) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
      for (int k = 0; k < 16; ++k) s += (u[k].x * u[k].x + u[k].y * u[k].y) + (u[k].z * u[k].z + u[k].w * u[k].w);
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
    if (xp && part == 0) { s_diag[r] = 0.5f * a.c * a.c * s; a.w.diag[(size_t)th * (Nq + Nc) + r] = s_diag[r]; }
  }
  __syncthreads();
  F2_TS(1);
  // S = F_q F_k^T over the features: 16-feature chunks c = wv, wv + 4, ...
  const float gst = s_g;
  float* qrow[2]; float* krow[2]; float qsub[2], ksub[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rq = 16 * i + lr, rk = 16 * i + lr;
    qrow[i] = rq < Nq ? a.w.eq + ((size_t)(t * Nq + rq) * H + h) * mp : nullptr;
    krow[i] = rk < Nc ? a.w.ek + ((size_t)(t * Nc + rk) * H + h) * mp : nullptr;
    qsub[i] = rq < Nq ? s_diag[rq] + s_mx[rq] : 0.f;
    ksub[i] = rk < Nc ? s_diag[Nq + rk] + gst : 0.f;
  }
  f32x4_t acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  const int tq = (Nq + 15) / 16, tk = (Nc + 15) / 16;
  auto ldd = [&](float* row, int j) { return (row != nullptr && j < mp) ? *reinterpret_cast<const float4*>(row + j) : make_float4(0.f, 0.f, 0.f, 0.f); };
  auto feat = [&](float* row, int j, float sub, const float4 dd, float4& f) {        // dd -> E (stored back), returns F = E + ratio eps (0 beyond m)
    f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row == nullptr) return;
    float4 e;
    e.x = j < m ? a.ratio * expf(dd.x - sub) : 0.f; e.y = j + 1 < m ? a.ratio * expf(dd.y - sub) : 0.f;
    e.z = j + 2 < m ? a.ratio * expf(dd.z - sub) : 0.f; e.w = j + 3 < m ? a.ratio * expf(dd.w - sub) : 0.f;
    *reinterpret_cast<float4*>(row + j) = e;
    f = make_float4(j < m ? e.x + a.re : 0.f, j + 1 < m ? e.y + a.re : 0.f, j + 2 < m ? e.z + a.re : 0.f, j + 3 < m ? e.w + a.re : 0.f);
  };
  // two 16-feature chunks per trip; the dd of trip + 1 are requested before the exponentials of this trip (un-pipelined, each of the
  // 11 trips waited out an HBM round trip)
  float4 dq[2][2], dk[2][2];
  auto fetch = [&](int jb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = jb + 16 * F2_NW * u + 4 * lq;
#pragma unroll
      for (int i = 0; i < 2; ++i) { dq[u][i] = ldd(qrow[i], j); dk[u][i] = ldd(krow[i], j); }
    }
  };
  fetch(16 * wv);
  for (int jb = 16 * wv; jb < mp; jb += 32 * F2_NW) {
    float4 cq[2][2], ck[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) { cq[u][i] = dq[u][i]; ck[u][i] = dk[u][i]; }
    if (jb + 32 * F2_NW < mp) fetch(jb + 32 * F2_NW);
    float4 fa[2][2], fb[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = jb + 16 * F2_NW * u + 4 * lq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        feat(j < mp ? qrow[i] : nullptr, j, qsub[i], cq[u][i], fa[u][i]);
        feat(j < mp ? krow[i] : nullptr, j, ksub[i], ck[u][i], fb[u][i]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          if (i >= tq || jj >= tk) continue;
          acc[i][jj] = mfma4(fa[u][i].x, fb[u][jj].x, acc[i][jj]); acc[i][jj] = mfma4(fa[u][i].y, fb[u][jj].y, acc[i][jj]);
          acc[i][jj] = mfma4(fa[u][i].z, fb[u][jj].z, acc[i][jj]); acc[i][jj] = mfma4(fa[u][i].w, fb[u][jj].w, acc[i][jj]);
        }
  }
  F2_TS(2);
  // fold the sixteen waves: S[n][n'] (C layout: n = 16 i + 4 lq + r, n' = 16 jj + lr)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) *reinterpret_cast<f32x4_t*>(red + (((wv * 2 + i) * 2 + jj) * 64 + lane) * 4) = acc[i][jj];
  __syncthreads();
  if (wv < 4) {
    const int i = wv >> 1, jj = wv & 1;            // wave wv folds tile (i, jj), in a fixed order
    f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < F2_NW; ++k) {
      const f32x4_t vv = *reinterpret_cast<const f32x4_t*>(red + (((k * 2 + i) * 2 + jj) * 64 + lane) * 4);
      s[0] += vv[0]; s[1] += vv[1]; s[2] += vv[2]; s[3] += vv[3];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 16 * i + 4 * lq + r, np = 16 * jj + lr;
      if (n < Nq && np < Nc) { s_S[n * (MAXN + 1) + np] = s[r]; a.w.S[((size_t)th * Nq + n) * Nc + np] = s[r]; }
    }
  }
  __syncthreads();
  if (tid < Nq) {
    float s = 0.f;
    for (int np = 0; np < Nc; ++np) s += s_S[tid * (MAXN + 1) + np];
    s_D[tid] = s; a.w.D[(size_t)th * Nq + tid] = s;
  }
  __syncthreads();
  F2_TS(3);
  // out[t][n][e H + h] = sum_n' S[n][n'] v[(t, n', h)][e] / D[n] on the matrix core: wave = a 16-channel tile of e (d / 16 <= 16
  // tiles), A = S[n][n'] from LDS (lane: n = 16 i + lr, k = n' = 4 ks + lq), B = v[n'][e] (lane: k = n', column e = 16 wave + lr),
  // K = 32 key slots (zero beyond Nc).  (As scalar code - thread = (e, n half), 15 loads, 240 FMAs against LDS broadcasts, 16
  // stores - this phase was 7.6 of the kernel's 26 us.)
  for (int et = wv; et * 16 < d; et += F2_NW) {
    const int e = 16 * et + lr;
    float bv[MAXN / 4];
#pragma unroll
    for (int ks = 0; ks < MAXN / 4; ++ks) { const int np = 4 * ks + lq; bv[ks] = np < Nc ? a.v[((size_t)(t * Nc + np) * H + h) * d + e] : 0.f; }
    f32x4_t o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < MAXN / 4; ++ks) {
      const int np = 4 * ks + lq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float av = (16 * i + lr < Nq && np < Nc) ? s_S[(16 * i + lr) * (MAXN + 1) + np] : 0.f;
        o[i] = mfma4(av, bv[ks], o[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = 16 * i + 4 * lq + r;
        if (nn < Nq) a.out[(size_t)(t * Nq + nn) * ((size_t)d * H) + (size_t)e * H + h] = o[i][r] / s_D[nn];
      }
  }
  F2_TS(4);
}

// ---- B1 ---------------------------------------------------------------------------------------------------------
// grid (task x head, NSP): every split recomputes the small part (w, dS; split 0 also writes dV), then takes the feature tiles
// tau = 4 split + wave, + 16, ... of G_q / G_k and leaves its partial row sums.
__global__ __launch_bounds__(256) void b1_kernel(const Args a) {
  __shared__ float s_S[MAXN * (MAXN + 1)], s_dS[MAXN * (MAXN + 1)], s_D[MAXN], s_w[MAXN], s_rq[4][MAXN], s_rk[4][MAXN], red[4 * 4 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int th = blockIdx.x, t = th / a.f.H, h = th % a.f.H, sp = blockIdx.y;
  const int d = a.f.d, m = a.f.m, Nq = a.f.Nq, Nc = a.f.Nc, H = a.f.H, mp = a.w.mp;
  const int tq = (Nq + 15) / 16, tk = (Nc + 15) / 16;
  for (int i = tid; i < Nq * Nc; i += 256) s_S[(i / Nc) * (MAXN + 1) + i % Nc] = a.w.S[(size_t)th * Nq * Nc + i];
  if (tid < Nq) s_D[tid] = a.w.D[(size_t)th * Nq + tid];
  for (int i = tid; i < MAXN * (MAXN + 1); i += 256) s_dS[i] = 0.f;
  // dS[n][n'] = dO[n] . (v[n'] - O[n]) / D[n].  O[n] is a convex combination of the block's value rows, so when those share a
  // large common component (the task encoder's features do: |v| ~ 20 x the spread between keys) the two inner products dO . v and
  // dO . O agree in their leading digits and their fp32 rounding dominates the difference.  Both are therefore taken relative to
  // the block's first value row c = v[0]: dO . (v[n'] - c) and w[n] = dO . (O[n] - c) - the same dS, without the common mode.
  const float* vc = a.v + ((size_t)(t * Nc) * H + h) * d;
  // w[n] = sum_e dO[n][e] (O[n][e] - c[e]): 8 threads per row
  {
    const int n = tid >> 3, part = tid & 7;
    float s = 0.f;
    if (n < Nq) {
      const size_t ob = (size_t)(t * Nq + n) * ((size_t)d * H) + h;
#pragma unroll 4
      for (int e = part; e < d; e += 8) s += a.dout[ob + (size_t)e * H] * (a.out[ob + (size_t)e * H] - vc[e]);
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (n < Nq && part == 0) s_w[n] = s;
  }
  // T1[n][n'] = sum_e dO[n][e] (v[n'][e] - c[e]) on the matrix core: k-steps (4 channels e) ks = wv, wv + 4, ...; 8 per trip in flight
  {
    f32x4_t acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    const float* ap[2]; const float* bp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = 16 * i + lr;
      ap[i] = n < Nq ? a.dout + (size_t)(t * Nq + n) * ((size_t)d * H) + h : nullptr;        // A[n (lr)][e (lq)] = dO[n][e]
      bp[i] = n < Nc ? a.v + ((size_t)(t * Nc + n) * H + h) * d : nullptr;                    // B[e (lq)][n' (lr)] = v[n'][e]
    }
    for (int kb = wv; kb < d / 4; kb += 32) {
      float av[8][2], bv[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = 4 * (kb + 4 * u) + lq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          av[u][i] = (ap[i] && e < d) ? ap[i][(size_t)e * H] : 0.f;
          bv[u][i] = (bp[i] && e < d) ? bp[i][e] - vc[e] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (i < tq && jj < tk) acc[i][jj] = mfma4(av[u][i], bv[u][jj], acc[i][jj]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) *reinterpret_cast<f32x4_t*>(red + (((wv * 2 + i) * 2 + jj) * 64 + lane) * 4) = acc[i][jj];
  }
  __syncthreads();
  {
    const int i = wv >> 1, jj = wv & 1;            // wave wv folds tile (i, jj): dS = (T1 - w) / D
    f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4_t vv = *reinterpret_cast<const f32x4_t*>(red + (((k * 2 + i) * 2 + jj) * 64 + lane) * 4);
      sacc[0] += vv[0]; sacc[1] += vv[1]; sacc[2] += vv[2]; sacc[3] += vv[3];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 16 * i + 4 * lq + r, np = 16 * jj + lr;
      if (n < Nq && np < Nc) s_dS[n * (MAXN + 1) + np] = (sacc[r] - s_w[n]) / s_D[n];
    }
  }
  // dv[(t, n', h)][e] = sum_n S[n][n'] dO[n][e] / D[n]   (split 0 only)
  if (sp == 0) {
    for (int idx = tid; idx < d * 2; idx += 256) {
      const int e = idx % d, half = idx / d;
      if (16 * half >= Nc) continue;
      float o[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) o[k] = 0.f;
      for (int n = 0; n < Nq; ++n) {
        const float g = a.dout[(size_t)(t * Nq + n) * ((size_t)d * H) + (size_t)e * H + h] / s_D[n];
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = fmaf(s_S[n * (MAXN + 1) + 16 * half + k], g, o[k]);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int np = 16 * half + k;
        if (np < Nc) a.dv[((size_t)(t * Nc + np) * H + h) * d + e] = o[k];
      }
    }
  }
  __syncthreads();
  // G_q[n][j] = (sum_n' dS[n][n'] F_k[n'][j]) E_q[n][j];  G_k[n'][j] = (sum_n dS[n][n'] F_q[n][j]) E_k[n'][j]
  // A from LDS (dS or its transpose), B = F rows (lane lr = feature, lq = which of the 4 reduction rows of the k-step)
  float rq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, rk[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int tau = 4 * sp + wv; tau < mp / 16; tau += 4 * NSP) {
    const int j = 16 * tau + lr;
    const bool jin = j < m;
    float fk[8], fq[8], eqv[2][4], ekv[2][4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {            // reduction index (n' for G_q, n for G_k) = 4 ks + lq; every load of the tile in flight
      const int o = 4 * ks + lq;
      fk[ks] = (o < Nc && jin) ? a.w.ek[((size_t)(t * Nc + o) * H + h) * mp + j] + a.re : 0.f;
      fq[ks] = (o < Nq && jin) ? a.w.eq[((size_t)(t * Nq + o) * H + h) * mp + j] + a.re : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * lq + r;
        eqv[i][r] = (row < Nq && jin) ? a.w.eq[((size_t)(t * Nq + row) * H + h) * mp + j] : 0.f;
        ekv[i][r] = (row < Nc && jin) ? a.w.ek[((size_t)(t * Nc + row) * H + h) * mp + j] : 0.f;
      }
    f32x4_t gq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, gk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int o = 4 * ks + lq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < tq && 4 * ks < Nc) gq[i] = mfma4(s_dS[(16 * i + lr) * (MAXN + 1) + o], fk[ks], gq[i]);      // A[n (lr)][n' (lq)]
        if (i < tk && 4 * ks < Nq) gk[i] = mfma4(s_dS[o * (MAXN + 1) + 16 * i + lr], fq[ks], gk[i]);        // A[n' (lr)][n (lq)]
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + 4 * lq + r;
        if (row < Nq) { const float g = gq[i][r] * eqv[i][r]; a.w.gq[((size_t)(t * Nq + row) * H + h) * mp + j] = g; rq[i][r] += g; }
        if (row < Nc) { const float g = gk[i][r] * ekv[i][r]; a.w.gk[((size_t)(t * Nc + row) * H + h) * mp + j] = g; rk[i][r] += g; }
      }
  }
  // partial row sums of this split: across the 16 feature lanes, then the four waves
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float x0 = rq[i][r], x1 = rk[i][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { x0 += __shfl_xor(x0, off, 64); x1 += __shfl_xor(x1, off, 64); }
      if (lr == 0) { s_rq[wv][16 * i + 4 * lq + r] = x0; s_rk[wv][16 * i + 4 * lq + r] = x1; }
    }
  __syncthreads();
  if (tid < Nq) a.w.rs_q[((size_t)(t * Nq + tid) * H + h) * NSP + sp] = (s_rq[0][tid] + s_rq[1][tid]) + (s_rq[2][tid] + s_rq[3][tid]);
  if (tid >= 64 && tid - 64 < Nc) {
    const int n = tid - 64;
    a.w.rs_k[((size_t)(t * Nc + n) * H + h) * NSP + sp] = (s_rk[0][n] + s_rk[1][n]) + (s_rk[2][n] + s_rk[3][n]);
  }
}

// ---- B2 ---------------------------------------------------------------------------------------------------------
// dx[row][e] = c sum_j ddd[row][j] P[j][e] - rsum[row] c^2 x[row][e],  ddd = G - stabiliser corrections (rank-1 fix-ups in the epilogue).
// grid (task x head, d / 64): a workgroup owns 64 channels e (four N-tiles) for the block's rows; its four waves split the m
// features.  (With 16-channel slices every block's G rows - 171 KB - were re-read by 16 workgroups: 175 MB through L2 per launch for
// 11 MB of gradients, and the kernel ran at L2 bandwidth.)
constexpr int B2_NT = 4;
#ifndef B2_NW
#define B2_NW 8                   // waves per workgroup: the m features are split over them (with 4 - one wave per SIMD, 704 MFMAs and 400 loads each - 35.0 us at c5's shape, with 8: 32.8)
#endif
constexpr int B2_TH = 64 * B2_NW;
template <int RT>
__global__ __launch_bounds__(B2_TH) void b2_kernel(const Args a) {
  __shared__ float s_gt;
  __shared__ float sm[B2_TH];
  __shared__ float red[B2_NW * RT * B2_NT * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int th = blockIdx.x, t = th / a.f.H, h = th % a.f.H;
  const int d = a.f.d, m = a.f.m, Nq = a.f.Nq, Nc = a.f.Nc, H = a.f.H, mp = a.w.mp, R = Nq + Nc;
  const int e0 = blockIdx.y * 16 * B2_NT;
  // sum of G over every key row of the batch (the global stabiliser's gradient)
  {
    // (four independent partial sums: as one rolled loop its 15 trips were 15 serial L2 round trips in every one of the workgroups)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const int nrs = a.f.T * Nc * H * NSP;
    int i = tid;
    for (; i + 3 * B2_TH < nrs; i += 4 * B2_TH) { s0 += a.w.rs_k[i]; s1 += a.w.rs_k[i + B2_TH]; s2 += a.w.rs_k[i + 2 * B2_TH]; s3 += a.w.rs_k[i + 3 * B2_TH]; }
    for (; i < nrs; i += B2_TH) s0 += a.w.rs_k[i];
    const float s = (s0 + s1) + (s2 + s3);
    sm[tid] = s;
    __syncthreads();
    for (int k = B2_TH / 2; k > 0; k >>= 1) { if (tid < k) sm[tid] += sm[tid + k]; __syncthreads(); }
    if (tid == 0) s_gt = sm[0] + (a.gt_fix ? a.gt_fix[0] : 0.f);
  }
  const float* grow[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int r = 16 * i + lr;
    grow[i] = r < Nq ? a.w.gq + ((size_t)(t * Nq + r) * H + h) * mp : (r < R ? a.w.gk + ((size_t)(t * Nc + r - Nq) * H + h) * mp : nullptr);
  }
  f32x4_t acc[RT][B2_NT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int n = 0; n < B2_NT; ++n) acc[i][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float* pcol = a.proj + e0 + lr;
  constexpr int UF = 2;
  for (int jb = 16 * wv; jb < mp; jb += 16 * B2_NW * UF) {
    float4 g[UF][RT]; float p[UF][4][B2_NT];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int j = jb + 16 * B2_NW * u + 4 * lq;
#pragma unroll
      for (int i = 0; i < RT; ++i) g[u][i] = (grow[i] && j < mp) ? *reinterpret_cast<const float4*>(grow[i] + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int n = 0; n < B2_NT; ++n) p[u][k][n] = (j + k < m && e0 + 16 * n + lr < d) ? pcol[(size_t)(j + k) * d + 16 * n] : 0.f;        // B[k = j (lq)][n = e (lr)]
    }
#pragma unroll
    for (int u = 0; u < UF; ++u)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int n = 0; n < B2_NT; ++n) {
          acc[i][n] = mfma4(g[u][i].x, p[u][0][n], acc[i][n]); acc[i][n] = mfma4(g[u][i].y, p[u][1][n], acc[i][n]);
          acc[i][n] = mfma4(g[u][i].z, p[u][2][n], acc[i][n]); acc[i][n] = mfma4(g[u][i].w, p[u][3][n], acc[i][n]);
        }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int n = 0; n < B2_NT; ++n) *reinterpret_cast<f32x4_t*>(red + (((wv * RT + i) * B2_NT + n) * 64 + lane) * 4) = acc[i][n];
  __syncthreads();
  const int gr = a.w.gpos[0], gj = a.w.gpos[1];
  // tile (i, n) is folded and finished by wave (i * B2_NT + n) % B2_NW, in a fixed order
  for (int tile = wv; tile < RT * B2_NT; tile += B2_NW) {
    const int i = tile / B2_NT, n = tile % B2_NT;
    f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < B2_NW; ++k) {
      const f32x4_t vv = *reinterpret_cast<const f32x4_t*>(red + (((k * RT + i) * B2_NT + n) * 64 + lane) * 4);
      sacc[0] += vv[0]; sacc[1] += vv[1]; sacc[2] += vv[2]; sacc[3] += vv[3];
    }
    const int e = e0 + 16 * n + lr;
    if (e >= d) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * i + 4 * lq + r;
      if (row >= R) continue;
      float vv = sacc[r];
      if (row < Nq) {
        const size_t g = (size_t)(t * Nq + row) * H + h;
        const float rs = sum_splits(a.w.rs_q + g * NSP);
        vv -= rs * a.proj[(size_t)a.w.arg_q[g] * d + e];              // - [j == argmax] rowsum
        a.dq[g * d + e] = a.c * vv - rs * a.c * a.c * a.q[g * d + e];
      } else {
        const size_t g = (size_t)(t * Nc + row - Nq) * H + h;
        const float rs = sum_splits(a.w.rs_k + g * NSP);
        if ((int)g == gr) vv -= s_gt * a.proj[(size_t)gj * d + e];    // - [this is THE global arg-max element] total
        a.dk[g * d + e] = a.c * vv - rs * a.c * a.c * a.k[g * d + e];
      }
    }
  }
}

inline Args make_args(const FavorDims& f, const Ws& w, const float* q, const float* k, const float* v, const float* proj) {
  Args a{};
  a.f = f; a.w = w; a.q = q; a.k = k; a.v = v; a.proj = proj;
  a.c = powf((float)f.d, -0.25f); a.ratio = 1.0f / sqrtf((float)f.m); a.re = a.ratio * 1e-4f;
  return a;
}
inline int forward(const FavorDims& f, const float* q, const float* k, const float* v, const float* proj, float* out, void* ws, size_t ws_bytes,
                   hipStream_t s, const Stage& st = Stage{}) {
  const Ws w = carve(f, ws, ws_bytes);
  if (!w.ok) { set_error("favor_fwd: workspace too small"); return MLHOT_ERR_WORKSPACE; }
  Args a = make_args(f, w, q, k, v, proj);
  a.out = out;
  const int th = f.T * f.H, R = f.Nq + f.Nc;
  if (st.first()) {
    ProfScope ps("favor.f1", s);
    if (R <= 32) hipLaunchKernelGGL((f1_kernel<2>), dim3(th, w.nch), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((f1_kernel<4>), dim3(th, w.nch), dim3(256), 0, s, a);
  }
  MLHOT_TRY(check_launch("favor.f1"));
  // strict sharded parity (stab_xchg.h): F2 folds the per-workgroup candidates (wg_v, wg_row, wg_j) into the batch maximum
  if (st.stage == 0) return sx::max_publish(w.wg_v, th * w.nch, st.x, s);
  if (st.stage == 1) MLHOT_TRY(sx::max_apply(w.wg_v, w.wg_row, th * w.nch, st.x, s));
  {
    ProfScope ps("favor.f2", s);
    hipLaunchKernelGGL(f2_kernel, dim3(th), dim3(F2_NT), 0, s, a);
  }
  return check_launch("favor.f2");
}
inline int backward(const FavorDims& f, const float* q, const float* k, const float* v, const float* proj, const float* out, const float* dout,
                    float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, hipStream_t s, const Stage& st = Stage{}) {
  const Ws w = carve(f, ws, ws_bytes);
  if (!w.ok) { set_error("favor_bwd: workspace too small"); return MLHOT_ERR_WORKSPACE; }
  Args a = make_args(f, w, q, k, v, proj);
  a.out = const_cast<float*>(out); a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv;
  const int th = f.T * f.H, R = f.Nq + f.Nc;
  if (st.first()) {
    ProfScope ps("favor.b1", s);
    hipLaunchKernelGGL(b1_kernel, dim3(th, NSP), dim3(256), 0, s, a);
  }
  MLHOT_TRY(check_launch("favor.b1"));
  // strict sharded parity: B2 folds rs_k into the stabiliser's gradient; the other ranks' share arrives through gt_fix
  const int nrs = f.T * f.Nc * f.H * NSP;
  if (st.stage == 0) return sx::sum_publish(w.rs_k, nrs, st.x, s);
  if (st.stage == 1) { MLHOT_TRY(sx::sum_apply(w.rs_k, nrs, st.x, w.gt_fix, 0, s)); a.gt_fix = w.gt_fix; }
  {
    ProfScope ps("favor.b2", s);
    if (R <= 32) hipLaunchKernelGGL((b2_kernel<2>), dim3(th, (f.d + 16 * B2_NT - 1) / (16 * B2_NT)), dim3(B2_TH), 0, s, a);
    else hipLaunchKernelGGL((b2_kernel<4>), dim3(th, (f.d + 16 * B2_NT - 1) / (16 * B2_NT)), dim3(B2_TH), 0, s, a);
  }
  return check_launch("favor.b2");
}

}  // namespace fv
}  // namespace mlhot
#endif

namespace mlhot {
// Which FAVOR+ implementation runs: the two-launch kernels above when the shot counts fit (and "favor2" is on), else favor.h's.
extern int g_favor2;
inline bool favor2_on(const FavorDims& f) {
#ifndef MLHOT_HOSTSIM
  return g_favor2 && fv::applies(f);
#else
  (void)f; return false;
#endif
}
inline size_t favor_ws_need(const FavorDims& f) {
  size_t a = 0, b = 0;
  favor_carve(f, nullptr, 0, &a);
#ifndef MLHOT_HOSTSIM
  if (fv::applies(f)) fv::carve(f, nullptr, 0, &b);
#endif
  return a > b ? a : b;
}
inline int favor_fwd_any(const FavorDims& f, const float* q, const float* k, const float* v, const float* proj, float* out, void* ws,
                         size_t ws_bytes, hipStream_t s, const Stage& st = Stage{}) {
#ifndef MLHOT_HOSTSIM
  if (favor2_on(f)) return fv::forward(f, q, k, v, proj, out, ws, ws_bytes, s, st);
#endif
  return favor_forward(f, q, k, v, proj, out, ws, ws_bytes, s, st);
}
inline int favor_bwd_any(const FavorDims& f, const float* q, const float* k, const float* v, const float* proj, const float* out,
                         const float* dout, float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, hipStream_t s,
                         const Stage& st = Stage{}) {
#ifndef MLHOT_HOSTSIM
  if (favor2_on(f)) return fv::backward(f, q, k, v, proj, out, dout, dq, dk, dv, ws, ws_bytes, s, st);
#endif
  return favor_backward(f, q, k, v, out, dout, dq, dk, dv, ws, ws_bytes, s, st);
}
}  // namespace mlhot
