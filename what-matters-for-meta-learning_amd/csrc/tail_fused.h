// Fused "tail" of the vanilla ANP model: everything between the image encoder and the loss
// (transform_y, EncoderFC, the 8-head K/V/Q projections, FAVOR+ attention, _W, r_to_z, decoder0)
// in 3 forward + 3 backward launches instead of ~75 latency-bound ones.
//
// The tail is < 2 % of the model's FLOPs but a chain of ~25 dependent tiny GEMMs per direction;
// on MI355X it is bound by launch + load latency, not by any roof.  So the parallelisation is by
// INDEPENDENT UNIT, each unit walking its dependent chain inside one workgroup with activations
// resident in LDS and weights streamed from L2 as B operands:
//   phase A  (one 512-thread workgroup per task):          context-side MLP, K/V/Q projections, the
//                                                          task's share of the key-stabiliser max
//   phase B  (one 256-thread workgroup per (task, head)):  FAVOR+ feature maps, S = Q'K'^T, out = SV/D
//   phase C  (one workgroup per task):                     _W, r_to_z, decoder0 (-> mu)
// and mirrored for the backward, with per-task weight-gradient slabs summed by one reduce launch.
// Rows of a task (<= 16 context / <= 16 target shots) are exactly one MFMA M-tile.
//
// GPU build only; requires Nc <= 16, Nq <= 16, attention mode.  The generic path (np_vanilla.h)
// stays as the fallback for larger shot counts and as the A/B reference
// (mlhot_set_option("tail_fused", 0)).  Both paths fill the same `saved` buffers, so forward and
// backward of either flavour can be mixed (which is how each kernel here is tested in isolation).
#pragma once
#include "common.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace tf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int H = 8;

struct WB {            // weight given as `nb` row blocks [rows][K] (nb = 1 for a plain Linear)
  const float* w[H];
  const float* b[H];
  int rows;
};

// ---- Y[16 x N] = act(X[16 x K] W^T + b).  X in LDS (row stride ldx, finite everywhere), W from
// global: lane (n = lr, k-group lq) loads W[n][k0+4lq .. +3] as one float4 and X[lr][k0+4lq..+3] from
// LDS, feeding 4 MFMAs whose k order is permuted identically on both operands.  N-tiles round-robin
// over the NW waves.  Output goes to LDS (ys) and/or global (yg, first nrows rows).
template <int NW>
__device__ __forceinline__ void wg_linear(const float* xs, int ldx, int K, const WB& wb, int N, int act,
                                          float* ys, int ldy, float* yg, int ldg, int nrows, int wave, int lane) {
  const int lr = lane & 15, lq = lane >> 4;
  for (int nt = wave; nt * 16 < N; nt += NW) {
    const int n = nt * 16 + lr;
    const bool vn = n < N;
    // block of this N-tile (wave-uniform: block rows are multiples of 16 whenever nb > 1); selected
    // with an unrolled compare chain so the pointer table never becomes a runtime-indexed array
    const int blk = (nt * 16) / wb.rows, rr = vn ? n - blk * wb.rows : 0;
    const float* wsel = wb.w[0];
    const float* bsel = wb.b[0];
#pragma unroll
    for (int i = 1; i < H; ++i)
      if (blk == i) { wsel = wb.w[i]; bsel = wb.b[i]; }
    const float* wrow = wsel + (size_t)rr * K;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const bool vec = (K & 3) == 0;
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 16) {
      const int kk = k0 + 4 * lq;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vn) {
        if (vec) { if (kk < K) b = *reinterpret_cast<const float4*>(wrow + kk); }
        else {
          if (kk < K) b.x = wrow[kk];
          if (kk + 1 < K) b.y = wrow[kk + 1];
          if (kk + 2 < K) b.z = wrow[kk + 2];
          if (kk + 3 < K) b.w = wrow[kk + 3];
        }
      }
      const float* xp = xs + lr * ldx + kk;
      acc = mfma4(xp[0], b.x, acc);
      acc = mfma4(xp[1], b.y, acc);
      acc = mfma4(xp[2], b.z, acc);
      acc = mfma4(xp[3], b.w, acc);
    }
    float bias = 0.f;
    if (vn && bsel) bias = bsel[rr];
    if (vn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        const float v = act_apply(act, acc[r] + bias);
        if (ys) ys[row * ldy + n] = v;
        if (yg && row < nrows) yg[(size_t)row * ldg + n] = v;
      }
    }
  }
}

// zero a [16 x ld] LDS tile
__device__ __forceinline__ void lds_zero(float* p, int n, int tid, int nthreads) {
  for (int i = tid; i < n; i += nthreads) p[i] = 0.f;
}
// global [nrows x width] (row stride ldg) -> LDS [16 x ld] (rows >= nrows left as they are)
__device__ __forceinline__ void lds_load(float* dst, int ld, const float* src, int ldg, int nrows, int width, int tid, int nthreads) {
  for (int i = tid; i < nrows * width; i += nthreads) {
    const int r = i / width, c = i % width;
    dst[r * ld + c] = src[(size_t)r * ldg + c];
  }
}

struct TailDims {
  int T, Nc, Nq, label_dim, y_dim, dw, dz, h0, h1, dec_h, out_act, m;
};

struct TailParams {
  const float *ty_w, *ty_b, *er_w[3], *er_b[3], *r2z_w, *r2z_b, *dec_w[3], *dec_b[3];
  const float *wk_w[H], *wk_b[H], *wv_w[H], *wv_b[H], *wq_w[H], *wq_b[H], *wo_w, *wo_b, *proj;
};

__device__ __forceinline__ WB wb1(const float* w, const float* b, int rows) {
  WB x;
#pragma unroll
  for (int i = 0; i < H; ++i) { x.w[i] = w; x.b[i] = b; }
  x.rows = rows;
  return x;
}
__device__ __forceinline__ WB wb8(const float* const* w, const float* const* b, int rows) {
  WB x;
#pragma unroll
  for (int i = 0; i < H; ++i) { x.w[i] = w[i]; x.b[i] = b ? b[i] : nullptr; }
  x.rows = rows;
  return x;
}

// LDS row strides (floats): width rounded up to a multiple of 16, +4 (keeps float4 alignment and
// moves consecutive rows to different banks)
__host__ __device__ constexpr int ldpad(int w) { return (w + 15) / 16 * 16 + 4; }

// ==================================================================================================
// phase A forward, one workgroup (512 threads) per task:
//   cat_in[:, dw:] = transform_y(ctx_y); h0, h1 = EncoderFC hidden; rs; kh = W_k(x_ctx); vh = W_v(rs);
//   qh = W_q(x_qry); pc = c * P (task 0); per-task max / arg-max of ddk = kh_h pc^T over (row, head, j).
// ==================================================================================================
struct PhaseAArgs {
  TailDims d; TailParams p;
  const float* ctx_y;
  float *cat_in, *h0, *h1, *rs, *dec_in, *kh, *vh, *qh;   // saved activations (global)
  float* pc;                                              // [m][dw]  c * projection
  float* tmax; int* targ;                                 // per task: max of ddk, {row, col}
};

__global__ __launch_bounds__(512) void phaseA_fwd_kernel(const PhaseAArgs a) {
  extern __shared__ float lds[];
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lrs = ldpad(d.dw), Lkh = ldpad(H * d.dw), Lxq = ldpad(d.dw), Ly = ldpad(d.label_dim);
  float* s_cat = lds;                    // [16][Lcat]
  float* s_h0 = s_cat + 16 * Lcat;
  float* s_h1 = s_h0 + 16 * Lh0;
  float* s_rs = s_h1 + 16 * Lh1;
  float* s_kh = s_rs + 16 * Lrs;         // [16][Lkh]
  float* s_xq = s_kh + 16 * Lkh;
  float* s_y = s_xq + 16 * Lxq;
  float* s_red = s_y + 16 * Ly;          // [8 waves][2]
  const int total = 16 * (Lcat + Lh0 + Lh1 + Lrs + Lkh + Lxq + Ly) + 32;
  lds_zero(lds, total, tid, 512);
  __syncthreads();
  float* g_cat = a.cat_in + (size_t)t * d.Nc * ldc;
  float* g_dec = a.dec_in + (size_t)t * d.Nq * ldd;
  lds_load(s_cat, Lcat, g_cat, ldc, d.Nc, d.dw, tid, 512);                       // x_ctx (encoder output)
  lds_load(s_xq, Lxq, g_dec, ldd, d.Nq, d.dw, tid, 512);                         // x_qry
  lds_load(s_y, Ly, a.ctx_y + (size_t)t * d.Nc * d.label_dim, d.label_dim, d.Nc, d.label_dim, tid, 512);
  if (t == 0) {
    const float c = powf((float)d.dw, -0.25f);
    for (int i = tid; i < d.m * d.dw; i += 512) a.pc[i] = c * a.p.proj[i];
  }
  __syncthreads();
  // transform_y -> cat[:, dw:]
  wg_linear<8>(s_y, Ly, d.label_dim, wb1(a.p.ty_w, a.p.ty_b, d.dw / 4), d.dw / 4, ACT_NONE, s_cat + d.dw, Lcat, g_cat + d.dw, ldc, d.Nc, wave, lane);
  // Q projection only needs x_qry: issue it alongside
  wg_linear<8>(s_xq, Lxq, d.dw, wb8(a.p.wq_w, a.p.wq_b, d.dw), H * d.dw, ACT_NONE, nullptr, 0, a.qh + (size_t)t * d.Nq * H * d.dw, H * d.dw, d.Nq, wave, lane);
  // K projection needs x_ctx only
  wg_linear<8>(s_cat, Lcat, d.dw, wb8(a.p.wk_w, a.p.wk_b, d.dw), H * d.dw, ACT_NONE, s_kh, Lkh, a.kh + (size_t)t * d.Nc * H * d.dw, H * d.dw, d.Nc, wave, lane);
  __syncthreads();
  wg_linear<8>(s_cat, Lcat, ldc, wb1(a.p.er_w[0], a.p.er_b[0], d.h0), d.h0, ACT_RELU, s_h0, Lh0, a.h0 + (size_t)t * d.Nc * d.h0, d.h0, d.Nc, wave, lane);
  __syncthreads();
  wg_linear<8>(s_h0, Lh0, d.h0, wb1(a.p.er_w[1], a.p.er_b[1], d.h1), d.h1, ACT_RELU, s_h1, Lh1, a.h1 + (size_t)t * d.Nc * d.h1, d.h1, d.Nc, wave, lane);
  __syncthreads();
  wg_linear<8>(s_h1, Lh1, d.h1, wb1(a.p.er_w[2], a.p.er_b[2], d.dw), d.dw, ACT_NONE, s_rs, Lrs, a.rs + (size_t)t * d.Nc * d.dw, d.dw, d.Nc, wave, lane);
  __syncthreads();
  wg_linear<8>(s_rs, Lrs, d.dw, wb8(a.p.wv_w, a.p.wv_b, d.dw), H * d.dw, ACT_NONE, nullptr, 0, a.vh + (size_t)t * d.Nc * H * d.dw, H * d.dw, d.Nc, wave, lane);

  // key-stabiliser share: max over (row < Nc, head, feature j) of ddk = c * kh_h . P[j]   (fast_attention.py:97)
  // (c * P is recomputed from proj here so this phase does not depend on task 0's pc write)
  const int lr = lane & 15, lq = lane >> 4;
  const float c = powf((float)d.dw, -0.25f);
  float best = -INFINITY; int brow = 0, bcol = 0;
  const int ntile = (d.m + 15) / 16;
  for (int it = wave; it < ntile * H; it += 8) {
    const int jt = it % ntile, h = it / ntile;
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < d.dw; k0 += 16) {
      const int kk = k0 + 4 * lq;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vj) b = *reinterpret_cast<const float4*>(a.p.proj + (size_t)j * d.dw + kk);
      const float* xp = s_kh + lr * Lkh + h * d.dw + kk;
      acc = mfma4(xp[0], c * b.x, acc);
      acc = mfma4(xp[1], c * b.y, acc);
      acc = mfma4(xp[2], c * b.z, acc);
      acc = mfma4(xp[3], c * b.w, acc);
    }
    if (vj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        if (row < d.Nc) {
          const int grow = (t * d.Nc + row) * H + h;      // row index of the [T*Nc*H, m] view
          const float v = acc[r];
          if (v > best || (v == best && (grow < brow || (grow == brow && j < bcol)))) { best = v; brow = grow; bcol = j; }
        }
      }
    }
  }
  // reduce (max, first arg-max in (row, col) order) over the workgroup
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int orow = __shfl_xor(brow, off, 64), ocol = __shfl_xor(bcol, off, 64);
    if (ov > best || (ov == best && (orow < brow || (orow == brow && ocol < bcol)))) { best = ov; brow = orow; bcol = ocol; }
  }
  __syncthreads();
  int* s_redi = reinterpret_cast<int*>(s_red + 8);
  if (lane == 0) { s_red[wave] = best; s_redi[2 * wave] = brow; s_redi[2 * wave + 1] = bcol; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 8; ++w) {
      const float ov = s_red[w]; const int orow = s_redi[2 * w], ocol = s_redi[2 * w + 1];
      if (ov > best || (ov == best && (orow < brow || (orow == brow && ocol < bcol)))) { best = ov; brow = orow; bcol = ocol; }
    }
    a.tmax[t] = best; a.targ[2 * t] = brow; a.targ[2 * t + 1] = bcol;
  }
}

__host__ inline size_t phaseA_lds_bytes(const TailDims& d) {
  const int ldc = d.dw + d.dw / 4;
  return sizeof(float) * (16 * (ldpad(ldc) + ldpad(d.h0) + ldpad(d.h1) + ldpad(d.dw) + ldpad(H * d.dw) + ldpad(d.dw) + ldpad(d.label_dim)) + 32);
}

// ==================================================================================================
// phase B forward, one workgroup (256 threads) per (task, head): FAVOR+ in the S-form
//   dd = x pc^T;  E = ratio exp(dd - diag - stab)  (stab: row max for q, batch-global max for k);
//   S = (Eq + re)(Ek + re)^T masked to valid rows;  D = rowsum S;  out = S V / D.
// Fills the same workspace fields as the generic path (qf, kf, S, D, arg_q, gmax, gpos).
// ==================================================================================================
struct PhaseBArgs {
  TailDims d;
  const float *qh, *kh, *vh, *pc;          // [T*N][H*dw] rows, pc [m][dw]
  const float* tmax; const int* targ;      // per-task key max shares
  float *qf, *kf, *S, *D, *gmax; int *arg_q, *gpos;
  float* merged;                           // [T*Nq][dw*H], column e*H + h
};

__global__ __launch_bounds__(256) void phaseB_fwd_kernel(const PhaseBArgs a) {
  extern __shared__ float lds[];
  const TailDims& d = a.d;
  const int t = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int Lx = ldpad(d.dw), Lf = ldpad(d.m);
  float* s_q = lds;                    // [16][Lx]
  float* s_k = s_q + 16 * Lx;
  float* s_v = s_k + 16 * Lx;
  float* s_qf = s_v + 16 * Lx;         // [16][Lf]  dd -> E
  float* s_kf = s_qf + 16 * Lf;
  float* s_S = s_kf + 16 * Lf;         // [4 waves][16][17] partials, then final in wave 0's slot
  float* s_st = s_S + 4 * 16 * 17;     // diag_q[16], diag_k[16], max_q[16], D[16]
  int* s_arg = reinterpret_cast<int*>(s_st + 64);   // arg_q[16]
  const int total = 16 * (3 * Lx + 2 * Lf) + 4 * 16 * 17 + 64 + 16;
  lds_zero(lds, total, tid, 256);
  __syncthreads();
  const int HD = H * d.dw;
  lds_load(s_q, Lx, a.qh + (size_t)t * d.Nq * HD + h * d.dw, HD, d.Nq, d.dw, tid, 256);
  lds_load(s_k, Lx, a.kh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  lds_load(s_v, Lx, a.vh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  // batch-global key stabiliser (identical in every workgroup: first maximum in task order)
  float gm = a.tmax[0]; int gt = 0;
  for (int i = 1; i < d.T; ++i) if (a.tmax[i] > gm) { gm = a.tmax[i]; gt = i; }
  if (blockIdx.x == 0 && tid == 0) { a.gmax[0] = gm; a.gpos[0] = a.targ[2 * gt]; a.gpos[1] = a.targ[2 * gt + 1]; }
  __syncthreads();
  // dd tiles: q and k against pc (B operand streamed from L2)
  const int ntile = (d.m + 15) / 16;
  for (int it = wave; it < 2 * ntile; it += 4) {
    const int isk = it >= ntile, jt = isk ? it - ntile : it;
    const float* xs = isk ? s_k : s_q;
    float* fs = isk ? s_kf : s_qf;
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < d.dw; k0 += 16) {
      const int kk = k0 + 4 * lq;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vj) b = *reinterpret_cast<const float4*>(a.pc + (size_t)j * d.dw + kk);
      const float* xp = xs + lr * Lx + kk;
      acc = mfma4(xp[0], b.x, acc);
      acc = mfma4(xp[1], b.y, acc);
      acc = mfma4(xp[2], b.z, acc);
      acc = mfma4(xp[3], b.w, acc);
    }
    if (vj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) fs[(4 * lq + r) * Lf + j] = acc[r];
    }
  }
  // diag = c^2/2 |x|^2 : 32 rows (16 q + 16 k), 8 threads per row
  {
    const float half_c2 = 0.5f / sqrtf((float)d.dw);
    const int row = tid >> 3, part = tid & 7;
    const float* xr = (row < 16 ? s_q + row * Lx : s_k + (row - 16) * Lx);
    float s = 0.f;
    for (int e = part; e < d.dw; e += 8) s += xr[e] * xr[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) s_st[row] = s * half_c2;
  }
  __syncthreads();
  // query row max / first arg-max: 16 rows x 16 threads
  {
    const int row = tid >> 4, part = tid & 15;
    float best = -INFINITY; int arg = 0x7fffffff;
    for (int j = part; j < d.m; j += 16) { const float v = s_qf[row * Lf + j]; if (v > best) { best = v; arg = j; } }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const float ov = __shfl_xor(best, off, 64); const int oa = __shfl_xor(arg, off, 64);
      if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (part == 0) { s_st[32 + row] = best; s_arg[row] = arg; }
  }
  __syncthreads();
  // E features in place (padding columns j >= m stay exactly 0 -> they are skipped below via `re` masking)
  const float ratio = 1.0f / sqrtf((float)d.m), re = ratio * 1e-4f;
  for (int i = tid; i < 16 * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    s_qf[row * Lf + j] = ratio * expf(s_qf[row * Lf + j] - s_st[row] - s_st[32 + row]);
    s_kf[row * Lf + j] = ratio * expf(s_kf[row * Lf + j] - s_st[16 + row] - gm);
  }
  __syncthreads();
  // save E features and arg_q for the backward (rows of the [T*N*H, m] views)
  for (int i = tid; i < d.Nq * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    a.qf[((size_t)(t * d.Nq + row) * H + h) * d.m + j] = s_qf[row * Lf + j];
  }
  for (int i = tid; i < d.Nc * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    a.kf[((size_t)(t * d.Nc + row) * H + h) * d.m + j] = s_kf[row * Lf + j];
  }
  if (tid < d.Nq) a.arg_q[(t * d.Nq + tid) * H + h] = s_arg[tid];
  // S = (Eq + re)(Ek + re)^T : M = 16 q rows, N = 16 k rows, K = m split over the 4 waves
  {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = wave * 4; j0 < d.m; j0 += 16) {
      const int j = j0 + lq;
      const bool vj = j < d.m;
      const float av = vj ? s_qf[lr * Lf + j] + re : 0.f;
      const float bv = vj ? s_kf[lr * Lf + j] + re : 0.f;
      acc = mfma4(av, bv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_S[(wave * 16 + 4 * lq + r) * 17 + lr] = acc[r];
  }
  __syncthreads();
  {
    const int n = tid >> 4, np = tid & 15;
    float s = (s_S[n * 17 + np] + s_S[(16 + n) * 17 + np]) + (s_S[(32 + n) * 17 + np] + s_S[(48 + n) * 17 + np]);
    if (n >= d.Nq || np >= d.Nc) s = 0.f;
    __syncthreads();
    s_S[n * 17 + np] = s;
    if (n < d.Nq && np < d.Nc) a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + np] = s;
  }
  __syncthreads();
  if (tid < 16) {
    float s = 0.f;
    for (int np = 0; np < d.Nc; ++np) s += s_S[tid * 17 + np];
    s_st[48 + tid] = s;
    if (tid < d.Nq) a.D[((size_t)t * H + h) * d.Nq + tid] = s;
  }
  __syncthreads();
  // out[n][e] = sum_n' S[n][n'] v[n'][e] / D[n]: N-tiles of e over the waves, K = 16 k rows
  for (int et = wave; et * 16 < d.dw; et += 4) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int np = 4 * s4 + lq;
      acc = mfma4(s_S[lr * 17 + np], s_v[np * Lx + et * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, e = et * 16 + lr;
      if (n < d.Nq) a.merged[(size_t)(t * d.Nq + n) * (d.dw * H) + e * H + h] = acc[r] / s_st[48 + n];
    }
  }
}

__host__ inline size_t phaseB_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (3 * ldpad(d.dw) + 2 * ldpad(d.m)) + 4 * 16 * 17 + 64 + 16);
}

// ==================================================================================================
// phase C forward, one workgroup per task: rr = _W(merged); z = r_to_z(rr) -> dec_in[:, dw:];
// d1, d2 = decoder hidden; mu = act(decoder out).
// ==================================================================================================
struct PhaseCArgs {
  TailDims d; TailParams p;
  const float* merged;
  float *rr, *dec_in, *d1, *d2, *mu;
};

__global__ __launch_bounds__(512) void phaseC_fwd_kernel(const PhaseCArgs a) {
  extern __shared__ float lds[];
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = d.dw + d.dz, HD = H * d.dw;
  const int Lm = ldpad(HD), Lr = ldpad(d.dw), Ld = ldpad(ldd), Lh = ldpad(d.dec_h);
  float* s_m = lds;                 // [16][Lm]
  float* s_rr = s_m + 16 * Lm;
  float* s_dec = s_rr + 16 * Lr;
  float* s_d1 = s_dec + 16 * Ld;
  float* s_d2 = s_d1 + 16 * Lh;
  lds_zero(lds, 16 * (Lm + Lr + Ld + 2 * Lh), tid, 512);
  __syncthreads();
  float* g_dec = a.dec_in + (size_t)t * d.Nq * ldd;
  lds_load(s_m, Lm, a.merged + (size_t)t * d.Nq * HD, HD, d.Nq, HD, tid, 512);
  lds_load(s_dec, Ld, g_dec, ldd, d.Nq, d.dw, tid, 512);          // x_qry
  __syncthreads();
  wg_linear<8>(s_m, Lm, HD, wb1(a.p.wo_w, a.p.wo_b, d.dw), d.dw, ACT_NONE, s_rr, Lr, a.rr + (size_t)t * d.Nq * d.dw, d.dw, d.Nq, wave, lane);
  __syncthreads();
  wg_linear<8>(s_rr, Lr, d.dw, wb1(a.p.r2z_w, a.p.r2z_b, d.dz), d.dz, ACT_NONE, s_dec + d.dw, Ld, g_dec + d.dw, ldd, d.Nq, wave, lane);
  __syncthreads();
  wg_linear<8>(s_dec, Ld, ldd, wb1(a.p.dec_w[0], a.p.dec_b[0], d.dec_h), d.dec_h, ACT_RELU, s_d1, Lh, a.d1 + (size_t)t * d.Nq * d.dec_h, d.dec_h, d.Nq, wave, lane);
  __syncthreads();
  wg_linear<8>(s_d1, Lh, d.dec_h, wb1(a.p.dec_w[1], a.p.dec_b[1], d.dec_h), d.dec_h, ACT_RELU, s_d2, Lh, a.d2 + (size_t)t * d.Nq * d.dec_h, d.dec_h, d.Nq, wave, lane);
  __syncthreads();
  wg_linear<8>(s_d2, Lh, d.dec_h, wb1(a.p.dec_w[2], a.p.dec_b[2], d.y_dim), d.y_dim, d.out_act, nullptr, 0, a.mu + (size_t)t * d.Nq * d.y_dim, d.y_dim, d.Nq, wave, lane);
}

__host__ inline size_t phaseC_lds_bytes(const TailDims& d) {
  return sizeof(float) * 16 * (ldpad(H * d.dw) + ldpad(d.dw) + ldpad(d.dw + d.dz) + 2 * ldpad(d.dec_h));
}

}  // namespace tf
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
