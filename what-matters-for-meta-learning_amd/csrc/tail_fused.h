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

// stage timestamps for latency hunting (build with -DMLHOT_TS; see scripts/tail_ts.py).  Compiled out otherwise.
#ifdef MLHOT_TS
__device__ long long* g_ts_dev = nullptr;
#define MLHOT_TSTAMP(i) do { if (g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0) g_ts_dev[i] = wall_clock64(); } while (0)
// inside a layer function: cycle stamps of call number g_ts_dev[199] (slots 200 + 8 * call + i)
#define MLHOT_TSCALL_BEGIN() long long* tsc_ = nullptr; do { if (g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0) { \
    const long long c_ = g_ts_dev[199]; g_ts_dev[199] = c_ + 1; if (c_ < 36) tsc_ = g_ts_dev + 200 + 8 * c_; } } while (0)
#define MLHOT_TSC(i) do { if (tsc_) tsc_[i] = clock64(); } while (0)
#else
#define MLHOT_TSTAMP(i) do {} while (0)
#define MLHOT_TSCALL_BEGIN() do {} while (0)
#define MLHOT_TSC(i) do {} while (0)
#endif

// ---- address spaces ----------------------------------------------------------------------------
// The layer functions below are shared, NOT inlined (see wg_linear), so their pointer parameters
// carry explicit address spaces: with plain `float*` parameters every LDS and weight access in them
// compiled to FLAT instructions (slow path, and each wait drains both memory counters), which made
// every layer of the chain cost 6-10 us.  Typed, the same code is ds_read_b128 / global_load_dwordx4.
#define MLHOT_LDS __attribute__((address_space(3)))
#define MLHOT_GLB __attribute__((address_space(1)))
typedef MLHOT_LDS float* lptr;
typedef const MLHOT_LDS float* lcptr;
typedef MLHOT_GLB float* gptr;
typedef const MLHOT_GLB float* gcptr;
typedef MLHOT_LDS unsigned long long* lu64;
typedef const MLHOT_LDS f32x4_t* lc4ptr;
typedef const MLHOT_GLB f32x4_t* gc4ptr;
#define G(p) ((::mlhot::tf::gptr)(p))
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Pointer table: a kernel copies its parameter struct (nothing but pointers) into LDS once; a layer
// is handed table slots, so a head-blocked weight (8 separately allocated [dw][dw] blocks) is a
// ds_read_b64 of slot `blk` instead of a by-reference struct living in scratch memory.
template <class P>
__device__ __forceinline__ void ptab_fill(lu64 tab, const P& p, int tid) {
  constexpr int n = sizeof(P) / 8;
  unsigned long long v = 0;
#pragma unroll
  for (int i = 0; i < n; ++i) {
    unsigned long long e;
    __builtin_memcpy(&e, reinterpret_cast<const char*>(&p) + 8 * i, 8);
    if (tid == i) v = e;
  }
  if (tid < n) tab[tid] = v;
}
struct WB {            // weight given as row blocks [rows][K] (one block for a plain Linear): table slots
  lu64 w, b;
  int rows, hasb;
};
// PRM (the kernel's parameter struct type) and ptab (its LDS table) are in scope at every use
#define WB1(wm, bm, rows_) ::mlhot::tf::WB{ptab + offsetof(PRM, wm) / 8, ptab + offsetof(PRM, bm) / 8, (rows_), 1}
template <class P> __host__ __device__ constexpr int ptab_floats() { return (int)((sizeof(P) / 8 * 2 + 3) / 4 * 4); }
#define WB1N(wm, rows_) ::mlhot::tf::WB{ptab + offsetof(PRM, wm) / 8, ptab + offsetof(PRM, wm) / 8, (rows_), 0}

// a pointer every lane holds identically (a table slot, a kernel argument) as a scalar register pair,
// so accesses through it use the scalar-base + 32-bit-offset addressing form
template <class T>
__device__ __forceinline__ T uniptr(T p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ gcptr uniptr(unsigned long long v) { return uniptr(reinterpret_cast<gcptr>(v)); }
__device__ __forceinline__ f32x4_t lds_read4(lcptr p, bool vec) {
  if (vec) return *reinterpret_cast<lc4ptr>(p);
  return f32x4_t{p[0], p[1], p[2], p[3]};
}

// ---- Y[16 x N] = act(X[16 x K] W^T + b).  X in LDS (row stride ldx, finite everywhere), W from
// global: lane (n = lr, k-group lq) loads W[n][k0+4lq .. +3] as one float4 and X[lr][k0+4lq..+3] from
// LDS, feeding 4 MFMAs whose k order is permuted identically on both operands.
// The chain is latency-bound, so each wave works on up to 4 N-tiles AT ONCE and on 4 K-blocks per
// trip (up to 16 weight float4 in flight, sharing the A operands), and when there are fewer N-tiles
// than waves the K range is split over the idle waves and folded through `red` (LDS, NW*256 floats;
// nullptr forbids the split).  Contains barriers when it splits: call from all waves.
// NOT inlined on purpose: these kernels run each layer exactly once per workgroup, so with every
// layer's unrolled body inlined the kernel was 56 KB of straight-line code and was bound by cold
// instruction fetch; as one shared function the code stays hot.  Every scalar argument is made
// wave-uniform on entry so loop control and the null checks are scalar branches.
template <int NW>
__device__ __attribute__((noinline)) void wg_linear(lcptr xs, int ldx, int K, WB wb, int N, int act,
                                                    lptr ys, int ldy, gptr yg, int ldg, int nrows, lptr red, int wave, int lane) {
  MLHOT_TSCALL_BEGIN();
  MLHOT_TSC(0);
  ldx = uni(ldx); K = uni(K); N = uni(N); act = uni(act); ldy = uni(ldy); ldg = uni(ldg); nrows = uni(nrows); wave = uni(wave);
  const int rows = uni(wb.rows), hasb = uni(wb.hasb);
  const bool has_ys = uni(ys != nullptr), has_yg = uni(yg != nullptr), has_red = uni(red != nullptr);
  const bool avec = uni((((unsigned)(size_t)xs & 15u) == 0u) && (ldx & 3) == 0);
  gptr ygu = uniptr(yg);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = (N + 15) >> 4;
  // K split over idle waves: NW / ntile chunks, a power of two for NW = 8 (ntile 1..4 -> 8, 4, 2, 2)
  int csh = 0;
  if (has_red && ntile * 2 <= NW) csh = ntile == 1 ? 3 : ntile == 2 ? 2 : 1;
  const int nchunk = 1 << csh;
  const int kblocks = (K + 15) >> 4, per = (kblocks + nchunk - 1) >> csh;
  const bool vec = (K & 3) == 0;
  int tpw = (ntile + NW - 1) / NW;               // N-tiles a wave handles together (NW is a compile-time power of two)
  if (tpw > 4) tpw = 4;
  if (nchunk > 1) tpw = 1;
  // The padding columns of X (k >= K) are zero and columns n >= N of the result are never stored, so
  // out-of-range operand addresses are only CLAMPED to something finite - no selects, no branches.
  const int kmax = vec ? K - 4 : K - 1;
  for (int it0 = 0; it0 < ntile * nchunk; it0 += NW * tpw) {
    const int it = it0 + wave * tpw;
    const bool active = it < ntile * nchunk;
    int tile0 = it, chunk = 0;
    if (nchunk > 1) { while (tile0 >= ntile) { tile0 -= ntile; ++chunk; } }
    if (!active) { tile0 = 0; chunk = 0; }
    int blk = 0, boff = tile0 * 16;               // row block of tile0 and the tile's first row inside it (scalar, no division)
    while (boff >= rows) { boff -= rows; ++blk; }
    f32x4_t acc[4];
    gcptr wbase[4]; int woff[4]; float bias[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const bool tile_ok = active && q < tpw && tile0 + q < ntile;      // wave-uniform
      if (q > 0 && tile_ok) { boff += 16; if (boff >= rows) { boff -= rows; ++blk; } }
      const int n = (tile0 + q) * 16 + lr;
      const int rr = (tile_ok && n < N) ? boff + lr : 0;
      const int bsel = tile_ok ? blk : 0;
      wbase[q] = uniptr(wb.w[bsel]);
      woff[q] = rr * K + 4 * lq;
      bias[q] = hasb ? uniptr(wb.b[bsel])[rr] : 0.f;
    }
    const int kb0 = chunk * per, kb1 = active ? (kb0 + per < kblocks ? kb0 + per : kblocks) : kb0;
    lcptr xrow = xs + lr * ldx + 4 * lq;
    MLHOT_TSC(1);
    for (int kb = kb0; kb < kb1; kb += 4) {
      f32x4_t b[4][4], a4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kb + u < kb1) {
          const int k16 = (kb + u) * 16;
          a4[u] = lds_read4(xrow + k16, avec);
          const int kc = k16 + 4 * lq <= kmax ? k16 : kmax - 4 * lq;       // clamp this lane's 4 k's into the row
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q < tpw) {
              if (vec) b[u][q] = *reinterpret_cast<gc4ptr>(wbase[q] + woff[q] + kc);
              else {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int ko = woff[q] + k16 + e; b[u][q][e] = wbase[q][k16 + 4 * lq + e <= kmax ? ko : woff[q] - 4 * lq + kmax]; }
              }
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kb + u < kb1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (q < tpw) {
              acc[q] = mfma4(a4[u][0], b[u][q][0], acc[q]);
              acc[q] = mfma4(a4[u][1], b[u][q][1], acc[q]);
              acc[q] = mfma4(a4[u][2], b[u][q][2], acc[q]);
              acc[q] = mfma4(a4[u][3], b[u][q][3], acc[q]);
            }
          }
        }
      }
    }
    MLHOT_TSC(2);
    if (nchunk > 1) {                              // fold the K chunks (fixed order); tpw == 1 here
      if (active && chunk > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[0][r];
      }
      __syncthreads();
      if (active && chunk == 0) {
        for (int c = 1; c < nchunk; ++c) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[0][r] += red[((wave + c * ntile) * 4 + r) * 64 + lane];
        }
      }
      __syncthreads();
    }
    MLHOT_TSC(3);
    if (active && chunk == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= tpw || tile0 + q >= ntile) continue;
        const int n = (tile0 + q) * 16 + lr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[q][r] + bias[q];
        if (act == ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (act == ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
        }
        if (n < N) {
          if (has_ys) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[(4 * lq + r) * ldy + n] = v[r];
          }
          if (has_yg) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (4 * lq + r < nrows) ygu[(4 * lq + r) * ldg + n] = v[r];
          }
        }
      }
    }
    MLHOT_TSC(4);
  }
  MLHOT_TSC(5);
}

// zero a [16 x ld] LDS tile
__device__ __forceinline__ void lds_zero(lptr p, int n, int tid, int nthreads) {
  for (int i = tid; i < n; i += nthreads) p[i] = 0.f;
}
// global [nrows x width] (row stride ldg) -> LDS [16 x ld] (rows >= nrows left as they are)
template <class SrcPtr>
__device__ __forceinline__ void lds_load(lptr dst, int ld, SrcPtr src, int ldg, int nrows, int width, int tid, int nthreads) {
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

// LDS row strides (floats): width rounded up to a multiple of 16, +4 (keeps float4 alignment and
// moves consecutive rows to different banks)
__host__ __device__ constexpr int ldpad(int w) { return (w + 15) / 16 * 16 + 4; }

// ==================================================================================================
// phase A forward, one workgroup (512 threads) per task:
//   cat_in[:, dw:] = transform_y(ctx_y); h0, h1 = EncoderFC hidden; rs; kh = W_k(x_ctx); vh = W_v(rs);
//   qh = W_q(x_qry); pc = c * P (task 0); per-task max / arg-max of ddk = kh_h pc^T over (row, head, j).
// ==================================================================================================
struct PhaseAArgs {
  int dbg;              // timing experiments: early exits (results become wrong)
  TailDims d; TailParams p;
  const float* ctx_y;
  float *cat_in, *h0, *h1, *rs, *dec_in, *kh, *vh, *qh;   // saved activations (global)
  float* pc;                                              // [m][dw]  c * projection
  float* tmax; int* targ;                                 // per task: max of ddk, {row, col}
};

__global__ __launch_bounds__(512) void phaseA_fwd_kernel(const PhaseAArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lrs = ldpad(d.dw), Lkh = ldpad(H * d.dw), Lxq = ldpad(d.dw), Ly = ldpad(d.label_dim);
  lptr s_cat = L0;                    // [16][Lcat]
  lptr s_h0 = s_cat + 16 * Lcat;
  lptr s_h1 = s_h0 + 16 * Lh0;
  lptr s_rs = s_h1 + 16 * Lh1;
  lptr s_kh = s_rs + 16 * Lrs;         // [16][Lkh]
  lptr s_xq = s_kh + 16 * Lkh;
  lptr s_y = s_xq + 16 * Lxq;
  lptr s_red = s_y + 16 * Ly;          // [8 waves][256] K-split partials of wg_linear; later the max reduction
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  const int total = 16 * (Lcat + Lh0 + Lh1 + Lrs + Lkh + Lxq + Ly) + 8 * 256;
  MLHOT_TSTAMP(0);
  lds_zero(L0, total, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(1);
  gptr g_cat = G(a.cat_in) + (size_t)t * d.Nc * ldc;
  gptr g_dec = G(a.dec_in) + (size_t)t * d.Nq * ldd;
  lds_load(s_cat, Lcat, g_cat, ldc, d.Nc, d.dw, tid, 512);                       // x_ctx (encoder output)
  lds_load(s_xq, Lxq, g_dec, ldd, d.Nq, d.dw, tid, 512);                         // x_qry
  lds_load(s_y, Ly, a.ctx_y + (size_t)t * d.Nc * d.label_dim, d.label_dim, d.Nc, d.label_dim, tid, 512);
  {   // pc = c * P for the later phases: each task writes its slice (one workgroup doing all of it
      // was a 16 us serial tail on the whole kernel)
    const float c = powf((float)d.dw, -0.25f);
    const int n = d.m * d.dw, per = (n + d.T - 1) / d.T, lo = t * per, hi = lo + per < n ? lo + per : n;
    for (int i = lo + tid; i < hi; i += 512) a.pc[i] = c * a.p.proj[i];
  }
  __syncthreads();
  MLHOT_TSTAMP(2);
  if (a.dbg & 2) return;
  // transform_y -> cat[:, dw:]
  wg_linear<8>(s_y, Ly, d.label_dim, WB1(ty_w, ty_b, d.dw / 4), d.dw / 4, ACT_NONE, s_cat + d.dw, Lcat, g_cat + d.dw, ldc, d.Nc, nullptr, wave, lane);
  MLHOT_TSTAMP(3);
  // Q projection only needs x_qry: issue it alongside
  wg_linear<8>(s_xq, Lxq, d.dw, WB1(wq_w, wq_b, d.dw), H * d.dw, ACT_NONE, nullptr, 0, G(a.qh + (size_t)t * d.Nq * H * d.dw), H * d.dw, d.Nq, nullptr, wave, lane);
  MLHOT_TSTAMP(4);
  // K projection needs x_ctx only
  wg_linear<8>(s_cat, Lcat, d.dw, WB1(wk_w, wk_b, d.dw), H * d.dw, ACT_NONE, s_kh, Lkh, G(a.kh + (size_t)t * d.Nc * H * d.dw), H * d.dw, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(5);
  if (a.dbg & 4) return;
  wg_linear<8>(s_cat, Lcat, ldc, WB1(er_w[0], er_b[0], d.h0), d.h0, ACT_RELU, s_h0, Lh0, G(a.h0 + (size_t)t * d.Nc * d.h0), d.h0, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(6);
  wg_linear<8>(s_h0, Lh0, d.h0, WB1(er_w[1], er_b[1], d.h1), d.h1, ACT_RELU, s_h1, Lh1, G(a.h1 + (size_t)t * d.Nc * d.h1), d.h1, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(7);
  wg_linear<8>(s_h1, Lh1, d.h1, WB1(er_w[2], er_b[2], d.dw), d.dw, ACT_NONE, s_rs, Lrs, G(a.rs + (size_t)t * d.Nc * d.dw), d.dw, d.Nc, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(8);
  wg_linear<8>(s_rs, Lrs, d.dw, WB1(wv_w, wv_b, d.dw), H * d.dw, ACT_NONE, nullptr, 0, G(a.vh + (size_t)t * d.Nc * H * d.dw), H * d.dw, d.Nc, nullptr, wave, lane);
  MLHOT_TSTAMP(9);

  if (a.dbg & 8) return;
  // key-stabiliser share: max over (row < Nc, head, feature j) of ddk = c * kh_h . P[j]   (fast_attention.py:97)
  // (c * P is recomputed from proj here so this phase does not depend on task 0's pc write)
  const int lr = lane & 15, lq = lane >> 4;
  const float c = powf((float)d.dw, -0.25f);
  float best = -INFINITY; int brow = 0, bcol = 0;
  const int ntile = (d.m + 15) / 16;
  // feature tiles round-robin over the waves; a tile's slice of c*P is loaded ONCE (dw/16 float4 per
  // lane, all in flight together) and reused by all 8 heads (8 accumulators)
  for (int jt = wave; jt < ntile; jt += 8) {
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < d.dw; k0 += 16) {
      const int kk = k0 + 4 * lq;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vj) b = *reinterpret_cast<const float4*>(a.p.proj + (size_t)j * d.dw + kk);
      b.x *= c; b.y *= c; b.z *= c; b.w *= c;
#pragma unroll
      for (int h = 0; h < H; ++h) {
        lcptr xp = s_kh + lr * Lkh + h * d.dw + kk;
        acc[h] = mfma4(xp[0], b.x, acc[h]);
        acc[h] = mfma4(xp[1], b.y, acc[h]);
        acc[h] = mfma4(xp[2], b.z, acc[h]);
        acc[h] = mfma4(xp[3], b.w, acc[h]);
      }
    }
    if (vj) {
#pragma unroll
      for (int h = 0; h < H; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          if (row < d.Nc) {
            const int grow = (t * d.Nc + row) * H + h;      // row index of the [T*Nc*H, m] view
            const float v = acc[h][r];
            if (v > best || (v == best && (grow < brow || (grow == brow && j < bcol)))) { best = v; brow = grow; bcol = j; }
          }
        }
    }
  }
  MLHOT_TSTAMP(10);
  // reduce (max, first arg-max in (row, col) order) over the workgroup
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int orow = __shfl_xor(brow, off, 64), ocol = __shfl_xor(bcol, off, 64);
    if (ov > best || (ov == best && (orow < brow || (orow == brow && ocol < bcol)))) { best = ov; brow = orow; bcol = ocol; }
  }
  __syncthreads();
  MLHOT_LDS int* s_redi = reinterpret_cast<MLHOT_LDS int*>(s_red + 8);
  if (lane == 0) { s_red[wave] = best; s_redi[2 * wave] = brow; s_redi[2 * wave + 1] = bcol; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 8; ++w) {
      const float ov = s_red[w]; const int orow = s_redi[2 * w], ocol = s_redi[2 * w + 1];
      if (ov > best || (ov == best && (orow < brow || (orow == brow && ocol < bcol)))) { best = ov; brow = orow; bcol = ocol; }
    }
    a.tmax[t] = best; a.targ[2 * t] = brow; a.targ[2 * t + 1] = bcol;
  }
  MLHOT_TSTAMP(11);
}

__host__ inline size_t phaseA_lds_bytes(const TailDims& d) {
  const int ldc = d.dw + d.dw / 4;
  return sizeof(float) * (16 * (ldpad(ldc) + ldpad(d.h0) + ldpad(d.h1) + ldpad(d.dw) + ldpad(H * d.dw) + ldpad(d.dw) + ldpad(d.label_dim)) + 8 * 256 +
                          ptab_floats<TailParams>());
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
  MLHOT_TSTAMP(32);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int Lx = ldpad(d.dw), Lf = ldpad(d.m);
  lptr s_q = L0;                    // [16][Lx]
  lptr s_k = s_q + 16 * Lx;
  lptr s_v = s_k + 16 * Lx;
  lptr s_qf = s_v + 16 * Lx;         // [16][Lf]  dd -> E
  lptr s_kf = s_qf + 16 * Lf;
  lptr s_S = s_kf + 16 * Lf;         // [4 waves][16][17] partials, then final in wave 0's slot
  lptr s_st = s_S + 4 * 16 * 17;     // diag_q[16], diag_k[16], max_q[16], D[16]
  MLHOT_LDS int* s_arg = reinterpret_cast<MLHOT_LDS int*>(s_st + 64);   // arg_q[16]
  const int total = 16 * (3 * Lx + 2 * Lf) + 4 * 16 * 17 + 64 + 16;
  lds_zero(L0, total, tid, 256);
  __syncthreads();
  MLHOT_TSTAMP(33);
  const int HD = H * d.dw;
  lds_load(s_q, Lx, a.qh + (size_t)t * d.Nq * HD + h * d.dw, HD, d.Nq, d.dw, tid, 256);
  lds_load(s_k, Lx, a.kh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  lds_load(s_v, Lx, a.vh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  // batch-global key stabiliser (identical in every workgroup: first maximum in task order)
  float gm = a.tmax[0]; int gt = 0;
  for (int i = 1; i < d.T; ++i) if (a.tmax[i] > gm) { gm = a.tmax[i]; gt = i; }
  if (blockIdx.x == 0 && tid == 0) { a.gmax[0] = gm; a.gpos[0] = a.targ[2 * gt]; a.gpos[1] = a.targ[2 * gt + 1]; }
  __syncthreads();
  MLHOT_TSTAMP(34);
  // dd tiles: q and k against pc; a feature tile's pc slice is loaded once (all float4 in flight
  // together) and feeds both the query and the key accumulator
  const int ntile = (d.m + 15) / 16;
  for (int jt = wave; jt < ntile; jt += 4) {
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t accq = {0.f, 0.f, 0.f, 0.f}, acck = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k0 = 0; k0 < d.dw; k0 += 16) {
      const int kk = k0 + 4 * lq;
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vj) b = *reinterpret_cast<const float4*>(a.pc + (size_t)j * d.dw + kk);
      lcptr xq = s_q + lr * Lx + kk;
      lcptr xk = s_k + lr * Lx + kk;
      accq = mfma4(xq[0], b.x, accq); acck = mfma4(xk[0], b.x, acck);
      accq = mfma4(xq[1], b.y, accq); acck = mfma4(xk[1], b.y, acck);
      accq = mfma4(xq[2], b.z, accq); acck = mfma4(xk[2], b.z, acck);
      accq = mfma4(xq[3], b.w, accq); acck = mfma4(xk[3], b.w, acck);
    }
    if (vj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_qf[(4 * lq + r) * Lf + j] = accq[r]; s_kf[(4 * lq + r) * Lf + j] = acck[r]; }
    }
  }
  // diag = c^2/2 |x|^2 : 32 rows (16 q + 16 k), 8 threads per row
  {
    const float half_c2 = 0.5f / sqrtf((float)d.dw);
    const int row = tid >> 3, part = tid & 7;
    lcptr xr = (row < 16 ? s_q + row * Lx : s_k + (row - 16) * Lx);
    float s = 0.f;
    for (int e = part; e < d.dw; e += 8) s += xr[e] * xr[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) s_st[row] = s * half_c2;
  }
  __syncthreads();
  MLHOT_TSTAMP(35);
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
  MLHOT_TSTAMP(36);
  // E features in place (padding columns j >= m stay exactly 0 -> they are skipped below via `re` masking)
  const float ratio = 1.0f / sqrtf((float)d.m), re = ratio * 1e-4f;
  for (int i = tid; i < 16 * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    s_qf[row * Lf + j] = ratio * expf(s_qf[row * Lf + j] - s_st[row] - s_st[32 + row]);
    s_kf[row * Lf + j] = ratio * expf(s_kf[row * Lf + j] - s_st[16 + row] - gm);
  }
  __syncthreads();
  MLHOT_TSTAMP(37);
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
  MLHOT_TSTAMP(38);
  {
    const int n = tid >> 4, np = tid & 15;
    float s = (s_S[n * 17 + np] + s_S[(16 + n) * 17 + np]) + (s_S[(32 + n) * 17 + np] + s_S[(48 + n) * 17 + np]);
    if (n >= d.Nq || np >= d.Nc) s = 0.f;
    __syncthreads();
    s_S[n * 17 + np] = s;
    if (n < d.Nq && np < d.Nc) a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + np] = s;
  }
  __syncthreads();
  MLHOT_TSTAMP(39);
  if (tid < 16) {
    float s = 0.f;
    for (int np = 0; np < d.Nc; ++np) s += s_S[tid * 17 + np];
    s_st[48 + tid] = s;
    if (tid < d.Nq) a.D[((size_t)t * H + h) * d.Nq + tid] = s;
  }
  __syncthreads();
  MLHOT_TSTAMP(40);
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
  MLHOT_TSTAMP(41);
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
  MLHOT_TSTAMP(64);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = d.dw + d.dz, HD = H * d.dw;
  const int Lm = ldpad(HD), Lr = ldpad(d.dw), Ld = ldpad(ldd), Lh = ldpad(d.dec_h);
  lptr s_m = L0;                 // [16][Lm]
  lptr s_rr = s_m + 16 * Lm;
  lptr s_dec = s_rr + 16 * Lr;
  lptr s_d1 = s_dec + 16 * Ld;
  lptr s_d2 = s_d1 + 16 * Lh;
  lptr s_red = s_d2 + 16 * Lh;    // [8 waves][256] K-split partials of wg_linear
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 16 * (Lm + Lr + Ld + 2 * Lh), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(65);
  gptr g_dec = G(a.dec_in) + (size_t)t * d.Nq * ldd;
  lds_load(s_m, Lm, a.merged + (size_t)t * d.Nq * HD, HD, d.Nq, HD, tid, 512);
  lds_load(s_dec, Ld, g_dec, ldd, d.Nq, d.dw, tid, 512);          // x_qry
  __syncthreads();
  MLHOT_TSTAMP(66);
  wg_linear<8>(s_m, Lm, HD, WB1(wo_w, wo_b, d.dw), d.dw, ACT_NONE, s_rr, Lr, G(a.rr + (size_t)t * d.Nq * d.dw), d.dw, d.Nq, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(67);
  wg_linear<8>(s_rr, Lr, d.dw, WB1(r2z_w, r2z_b, d.dz), d.dz, ACT_NONE, s_dec + d.dw, Ld, g_dec + d.dw, ldd, d.Nq, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(68);
  wg_linear<8>(s_dec, Ld, ldd, WB1(dec_w[0], dec_b[0], d.dec_h), d.dec_h, ACT_RELU, s_d1, Lh, G(a.d1 + (size_t)t * d.Nq * d.dec_h), d.dec_h, d.Nq, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(69);
  wg_linear<8>(s_d1, Lh, d.dec_h, WB1(dec_w[1], dec_b[1], d.dec_h), d.dec_h, ACT_RELU, s_d2, Lh, G(a.d2 + (size_t)t * d.Nq * d.dec_h), d.dec_h, d.Nq, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(70);
  wg_linear<8>(s_d2, Lh, d.dec_h, WB1(dec_w[2], dec_b[2], d.y_dim), d.y_dim, d.out_act, nullptr, 0, G(a.mu + (size_t)t * d.Nq * d.y_dim), d.y_dim, d.Nq, s_red, wave, lane);
  MLHOT_TSTAMP(71);
}

__host__ inline size_t phaseC_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (ldpad(H * d.dw) + ldpad(d.dw) + ldpad(d.dw + d.dz) + 2 * ldpad(d.dec_h)) + 8 * 256 + ptab_floats<TailParams>());
}

// ==================================================================================================
// backward building blocks (16-row tiles in LDS, zero-padded to a multiple of 16 columns)
// ==================================================================================================

// dW[Nout][Kin] = dY^T X (sum over the 16 rows; padded rows of dY are zero), db = column sums of dY.
// Output tiles (16 j x 16 i) round-robin over the waves, 4 MFMAs each; results go to the task's slab.
template <int NW>
__device__ __attribute__((noinline)) void wg_wgrad(lcptr dys, int ldy, int Nout, lcptr xs, int ldx, int Kin,
                                                   gptr dw, gptr db, int wave, int lane, int tid) {
  ldy = uni(ldy); Nout = uni(Nout); ldx = uni(ldx); Kin = uni(Kin); wave = uni(wave);
  const bool has_db = uni(db != nullptr);
  const int lr = lane & 15, lq = lane >> 4;
  const int nj = (Nout + 15) >> 4, ni = (Kin + 15) >> 4;
  for (int it = wave; it < nj * ni; it += NW) {
    const int j0 = (it / ni) * 16, i0 = (it % ni) * 16;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int row = 4 * s4 + lq;
      acc = mfma4(dys[row * ldy + j0 + lr], xs[row * ldx + i0 + lr], acc);
    }
    const int i = i0 + lr;
    if (i < Kin) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + 4 * lq + r;
        if (j < Nout) dw[(size_t)j * Kin + i] = acc[r];
      }
    }
  }
  if (has_db) {
    for (int j = tid; j < Nout; j += NW * 64) {
      float sum = 0.f;
#pragma unroll
      for (int row = 0; row < 16; ++row) sum += dys[row * ldy + j];
      db[j] = sum;
    }
  }
}

// dX[16][Kin] = dY[16][Nout] W[Nout][Kin]  (W as row blocks of `rows` rows).  Work items are
// (16-column tile of i) x (chunk of the reduction index j): when Kin has fewer tiles than there are
// waves (the head projections: Kin = 64, Nout = 512) the j range is split over the idle waves and
// the partial tiles are folded through `red` (LDS, NW*256 floats).  Per 16-wide j block: A = one
// float4 of dY from LDS, B = 4 coalesced dwords of W; 4 j blocks (16 weight loads) are in flight per
// trip.  Result to LDS (dxs) and/or global (dxg: rows < nrows; accumulate adds to what is there).
// Contains barriers: call from all waves.
template <int NW>
__device__ __attribute__((noinline)) void wg_dgrad(lcptr dys, int ldy, int Nout, WB wb, int Kin,
                                                   lptr dxs, int ldxs, gptr dxg, int ldg, int nrows, bool accumulate,
                                                   lptr red, int wave, int lane) {
  ldy = uni(ldy); Nout = uni(Nout); Kin = uni(Kin); ldxs = uni(ldxs); ldg = uni(ldg); nrows = uni(nrows); wave = uni(wave);
  const int rows = uni(wb.rows);
  const bool has_dxs = uni(dxs != nullptr), has_dxg = uni(dxg != nullptr), acc_out = uni((int)accumulate);
  const bool avec = uni((((unsigned)(size_t)dys & 15u) == 0u) && (ldy & 3) == 0);
  gptr dxgu = uniptr(dxg);
  const int lr = lane & 15, lq = lane >> 4;
  const int ntile = (Kin + 15) >> 4;
  int csh = 0;                                              // NW / ntile chunks of the j range (a power of two for NW = 8)
  if (ntile * 2 <= NW) csh = ntile == 1 ? 3 : ntile == 2 ? 2 : 1;
  const int nchunk = 1 << csh;
  const int jblocks = (Nout + 15) >> 4, per = (jblocks + nchunk - 1) >> csh;
  // dY's padding columns (j >= Nout) are zero and columns i >= Kin of the result are never stored:
  // out-of-range weight addresses are only clamped
  for (int it0 = 0; it0 < ntile * nchunk; it0 += NW) {
    const int it = it0 + wave;
    const bool active = it < ntile * nchunk;
    int tile = it, chunk = 0;
    while (tile >= ntile) { tile -= ntile; ++chunk; }
    if (!active) { tile = 0; chunk = 0; }
    const int i = tile * 16 + lr;
    const bool vi = active && i < Kin;
    const int ic = i < Kin ? i : Kin - 1;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const int jb0 = chunk * per, jb1 = active ? (jb0 + per < jblocks ? jb0 + per : jblocks) : jb0;
    int blk = 0, boff = jb0 * 16;                           // row block of j block jb0 and its first row inside it
    while (boff >= rows) { boff -= rows; ++blk; }
    lcptr arow = dys + lr * ldy + 4 * lq;
    for (int jb = jb0; jb < jb1; jb += 4) {
      f32x4_t b[4], a4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jb + u < jb1) {
          const int j0 = (jb + u) * 16;
          gcptr wsel = uniptr(wb.w[blk]);
          a4[u] = lds_read4(arow + j0, avec);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int jl = j0 + 4 * lq + e < Nout ? boff + 4 * lq + e : 0;      // row inside the block
            b[u][e] = wsel[jl * Kin + ic];
          }
          boff += 16;
          if (boff >= rows) { boff -= rows; ++blk; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jb + u < jb1) {
          acc = mfma4(a4[u][0], b[u][0], acc);
          acc = mfma4(a4[u][1], b[u][1], acc);
          acc = mfma4(a4[u][2], b[u][2], acc);
          acc = mfma4(a4[u][3], b[u][3], acc);
        }
      }
    }
    if (nchunk > 1) {                                       // fold the j chunks (fixed order)
      if (active && chunk > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (active && chunk == 0) {
        for (int c = 1; c < nchunk; ++c) {
          const int ow = wave + c * ntile - it0;           // wave that holds chunk c of this tile
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += red[(ow * 4 + r) * 64 + lane];
        }
      }
      __syncthreads();
    }
    if (vi && chunk == 0) {
      if (has_dxs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dxs[(4 * lq + r) * ldxs + i] = acc[r];
      }
      if (has_dxg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          if (row < nrows) {
            gptr o = dxgu + row * ldg + i;
            *o = acc_out ? *o + acc[r] : acc[r];
          }
        }
      }
    }
  }
}

// g[row][c] *= act'(y[row][c]) on the valid columns
__device__ __forceinline__ void lds_actgrad(lptr g, int ldg_, lcptr y, int ldy, int width, int act, int tid, int nthreads) {
  for (int i = tid; i < 16 * width; i += nthreads) {
    const int r = i / width, c = i % width;
    g[r * ldg_ + c] *= act_grad_from_out(act, y[r * ldy + c]);
  }
}

// per-task gradient slab: offsets (floats) of every tail parameter, in reduce order
struct TailSlab {
  int ty_w, ty_b, er_w[3], er_b[3], r2z_w, r2z_b, dec_w[3], dec_b[3], wk_w, wk_b, wv_w, wv_b, wq_w, wq_b, wo_w, wo_b, total;
};
__host__ inline TailSlab tail_slab_layout(const TailDims& d) {
  TailSlab s; int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) / 4 * 4; return r; };
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  s.ty_w = take(d.dw / 4 * d.label_dim); s.ty_b = take(d.dw / 4);
  s.er_w[0] = take(d.h0 * ldc); s.er_b[0] = take(d.h0);
  s.er_w[1] = take(d.h1 * d.h0); s.er_b[1] = take(d.h1);
  s.er_w[2] = take(d.dw * d.h1); s.er_b[2] = take(d.dw);
  s.r2z_w = take(d.dz * d.dw); s.r2z_b = take(d.dz);
  s.dec_w[0] = take(d.dec_h * ldd); s.dec_b[0] = take(d.dec_h);
  s.dec_w[1] = take(d.dec_h * d.dec_h); s.dec_b[1] = take(d.dec_h);
  s.dec_w[2] = take(d.y_dim * d.dec_h); s.dec_b[2] = take(d.y_dim);
  s.wk_w = take(H * d.dw * d.dw); s.wk_b = take(H * d.dw);
  s.wv_w = take(H * d.dw * d.dw); s.wv_b = take(H * d.dw);
  s.wq_w = take(H * d.dw * d.dw); s.wq_b = take(H * d.dw);
  s.wo_w = take(d.dw * H * d.dw); s.wo_b = take(d.dw);
  s.total = o;
  return s;
}

// ==================================================================================================
// phase C backward, one workgroup per task: decoder0, r_to_z and _W backward.
//   in : dmu, saved mu / d2 / d1 / dec_in / rr / merged
//   out: d_dec_in[:, :dw] (decoder's share of d x_qry), d_merged, weight-gradient slab entries
// ==================================================================================================
struct PhaseCBwdArgs {
  TailDims d; TailParams p; TailSlab sl;
  const float *dmu, *mu, *d2, *d1, *dec_in, *rr, *merged;
  float *d_dec_in, *d_merged, *slab;
};

__global__ __launch_bounds__(512) void phaseC_bwd_kernel(const PhaseCBwdArgs a) {
  MLHOT_TSTAMP(96);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = d.dw + d.dz, HD = H * d.dw;
  const int Ly = ldpad(d.y_dim), Lh = ldpad(d.dec_h), Ld = ldpad(ldd), Lr = ldpad(d.dw), Lm = ldpad(HD);
  lptr s_g = L0;                  // [16][Ly]   dmu * act'(mu)
  lptr s_d2 = s_g + 16 * Ly;       // saved activations
  lptr s_d1 = s_d2 + 16 * Lh;
  lptr s_dec = s_d1 + 16 * Lh;
  lptr s_rr = s_dec + 16 * Ld;
  lptr s_m = s_rr + 16 * Lr;
  lptr s_dd2 = s_m + 16 * Lm;      // gradients
  lptr s_dd1 = s_dd2 + 16 * Lh;
  lptr s_ddec = s_dd1 + 16 * Lh;
  lptr s_drr = s_ddec + 16 * Ld;
  lptr s_red = s_drr + 16 * Lr;    // [8 waves][256] partial tiles of wg_dgrad
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 16 * (Ly + 4 * Lh + 2 * Ld + 2 * Lr + Lm), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(97);
  const size_t rq = (size_t)t * d.Nq;
  for (int i = tid; i < d.Nq * d.y_dim; i += 512) {
    const int r = i / d.y_dim, c = i % d.y_dim;
    s_g[r * Ly + c] = a.dmu[(rq + r) * d.y_dim + c] * act_grad_from_out(d.out_act, a.mu[(rq + r) * d.y_dim + c]);
  }
  lds_load(s_d2, Lh, a.d2 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h, tid, 512);
  lds_load(s_d1, Lh, a.d1 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h, tid, 512);
  lds_load(s_dec, Ld, a.dec_in + rq * ldd, ldd, d.Nq, ldd, tid, 512);
  lds_load(s_rr, Lr, a.rr + rq * d.dw, d.dw, d.Nq, d.dw, tid, 512);
  lds_load(s_m, Lm, a.merged + rq * HD, HD, d.Nq, HD, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(98);
  gptr sl = G(a.slab) + (size_t)t * a.sl.total;
  // decoder0.4
  wg_wgrad<8>(s_g, Ly, d.y_dim, s_d2, Lh, d.dec_h, sl + a.sl.dec_w[2], sl + a.sl.dec_b[2], wave, lane, tid);
  wg_dgrad<8>(s_g, Ly, d.y_dim, WB1N(dec_w[2], d.y_dim), d.dec_h, s_dd2, Lh, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(99);
  lds_actgrad(s_dd2, Lh, s_d2, Lh, d.dec_h, ACT_RELU, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(100);
  // decoder0.2
  wg_wgrad<8>(s_dd2, Lh, d.dec_h, s_d1, Lh, d.dec_h, sl + a.sl.dec_w[1], sl + a.sl.dec_b[1], wave, lane, tid);
  wg_dgrad<8>(s_dd2, Lh, d.dec_h, WB1N(dec_w[1], d.dec_h), d.dec_h, s_dd1, Lh, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(101);
  lds_actgrad(s_dd1, Lh, s_d1, Lh, d.dec_h, ACT_RELU, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(102);
  // decoder0.0: input gradient = [d x_qry | dz]
  wg_wgrad<8>(s_dd1, Lh, d.dec_h, s_dec, Ld, ldd, sl + a.sl.dec_w[0], sl + a.sl.dec_b[0], wave, lane, tid);
  wg_dgrad<8>(s_dd1, Lh, d.dec_h, WB1N(dec_w[0], d.dec_h), ldd, s_ddec, Ld, G(a.d_dec_in + rq * ldd), ldd, d.Nq, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(103);
  // r_to_z (dz = s_ddec[:, dw:])
  wg_wgrad<8>(s_ddec + d.dw, Ld, d.dz, s_rr, Lr, d.dw, sl + a.sl.r2z_w, sl + a.sl.r2z_b, wave, lane, tid);
  wg_dgrad<8>(s_ddec + d.dw, Ld, d.dz, WB1N(r2z_w, d.dz), d.dw, s_drr, Lr, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(104);
  // _W
  wg_wgrad<8>(s_drr, Lr, d.dw, s_m, Lm, HD, sl + a.sl.wo_w, sl + a.sl.wo_b, wave, lane, tid);
  wg_dgrad<8>(s_drr, Lr, d.dw, WB1N(wo_w, d.dw), HD, nullptr, 0, G(a.d_merged + rq * HD), HD, d.Nq, false, s_red, wave, lane);
  MLHOT_TSTAMP(105);
}
__host__ inline size_t phaseC_bwd_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (ldpad(d.y_dim) + 4 * ldpad(d.dec_h) + 2 * ldpad(d.dw + d.dz) + 2 * ldpad(d.dw) + ldpad(H * d.dw)) + 8 * 256 +
                          ptab_floats<TailParams>());
}

// ==================================================================================================
// phase B backward, one workgroup per (task, head): FAVOR+ backward (S-form, see favor.h).
//   out: dqh / dkh (without the global arg-max correction) / dvh, part_k[t*H+h] = sum of rsum_k
// ==================================================================================================
struct PhaseBBwdArgs {
  TailDims d;
  const float *qh, *kh, *vh, *pc, *qf, *kf, *S, *D, *merged, *d_merged; const int* arg_q;
  float *dqh, *dkh, *dvh, *part_k;
};

__global__ __launch_bounds__(256) void phaseB_bwd_kernel(const PhaseBBwdArgs a) {
  MLHOT_TSTAMP(128);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int Lx = ldpad(d.dw), Lf = ldpad(d.m), HD = H * d.dw;
  lptr s_q = L0;                    // [16][Lx]
  lptr s_k = s_q + 16 * Lx;
  lptr s_v = s_k + 16 * Lx;
  lptr s_do = s_v + 16 * Lx;         // dO
  lptr s_qf = s_do + 16 * Lx;        // [16][Lf] E features, later d(dd)
  lptr s_kf = s_qf + 16 * Lf;
  lptr s_gq = s_kf + 16 * Lf;        // G
  lptr s_gk = s_gq + 16 * Lf;
  lptr s_S = s_gk + 16 * Lf;         // [16][17]  S / D
  lptr s_dS = s_S + 16 * 17;         // [16][17]
  lptr s_st = s_dS + 16 * 17;        // wv[16], D[16], rsum_q[16], rsum_k[16]
  const int total = 16 * (4 * Lx + 4 * Lf) + 2 * 16 * 17 + 64;
  lds_zero(L0, total, tid, 256);
  __syncthreads();
  MLHOT_TSTAMP(129);
  lds_load(s_q, Lx, a.qh + (size_t)t * d.Nq * HD + h * d.dw, HD, d.Nq, d.dw, tid, 256);
  lds_load(s_k, Lx, a.kh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  lds_load(s_v, Lx, a.vh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw, tid, 256);
  for (int i = tid; i < d.Nq * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    s_qf[row * Lf + j] = a.qf[((size_t)(t * d.Nq + row) * H + h) * d.m + j];
  }
  for (int i = tid; i < d.Nc * d.m; i += 256) {
    const int row = i / d.m, j = i % d.m;
    s_kf[row * Lf + j] = a.kf[((size_t)(t * d.Nc + row) * H + h) * d.m + j];
  }
  // dO[n][e] = d_merged[(t,n)][e*H + h];  wv[n] = sum_e dO * O
  for (int i = tid; i < d.Nq * d.dw; i += 256) {
    const int n = i / d.dw, e = i % d.dw;
    s_do[n * Lx + e] = a.d_merged[(size_t)(t * d.Nq + n) * HD + e * H + h];
  }
  if (tid < d.Nq) s_st[16 + tid] = a.D[((size_t)t * H + h) * d.Nq + tid];
  __syncthreads();
  MLHOT_TSTAMP(130);
  {
    const int n = tid >> 4, part = tid & 15;
    float s = 0.f;
    if (n < d.Nq)
      for (int e = part; e < d.dw; e += 16) s += s_do[n * Lx + e] * a.merged[(size_t)(t * d.Nq + n) * HD + e * H + h];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
    if (part == 0) s_st[n] = s;
    // S / D
    float sd = 0.f;
    if (n < d.Nq && part < d.Nc) sd = a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + part] / s_st[16 + n];
    s_S[n * 17 + part] = sd;
  }
  __syncthreads();
  MLHOT_TSTAMP(131);
  // dS[n][n'] = (dO[n] . v[n'] - wv[n]) / D[n]   (wave 0), valid entries only
  if (wave == 0) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int e0 = 0; e0 < d.dw; e0 += 4) acc = mfma4(s_do[lr * Lx + e0 + lq], s_v[lr * Lx + e0 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, np = lr;
      s_dS[n * 17 + np] = (n < d.Nq && np < d.Nc) ? (acc[r] - s_st[n]) / s_st[16 + n] : 0.f;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(132);
  // dV[n'][e] = sum_n (S/D)[n][n'] dO[n][e]  -> dvh
  for (int et = wave; et * 16 < d.dw; et += 4) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int n = 4 * s4 + lq;
      acc = mfma4(s_S[n * 17 + lr], s_do[n * Lx + et * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int np = 4 * lq + r;
      if (np < d.Nc) a.dvh[(size_t)(t * d.Nc + np) * HD + h * d.dw + et * 16 + lr] = acc[r];
    }
  }
  // G = dF (.) E with dQ' = dS (Ek + re), dK' = dS^T (Eq + re): feature tiles over the waves
  const float ratio = 1.0f / sqrtf((float)d.m), re = ratio * 1e-4f;
  const int ntile = (d.m + 15) / 16;
  for (int it = wave; it < 2 * ntile; it += 4) {
    const int isk = it >= ntile, jt = isk ? it - ntile : it;
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int o = 4 * s4 + lq;                    // summed row index (n' for queries, n for keys)
      const float av = isk ? s_dS[o * 17 + lr] : s_dS[lr * 17 + o];
      const float bv = vj ? (isk ? s_qf[o * Lf + j] : s_kf[o * Lf + j]) + re : 0.f;
      acc = mfma4(av, bv, acc);
    }
    if (vj) {
      lptr g = isk ? s_gk : s_gq;
      lcptr f = isk ? s_kf : s_qf;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int row = 4 * lq + r; g[row * Lf + j] = acc[r] * f[row * Lf + j]; }
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(133);
  // row sums of G (32 rows x 8 threads), then d(dd): queries subtract the row sum at the arg-max
  {
    const int row = tid >> 3, part = tid & 7;
    lcptr g = row < 16 ? s_gq + row * Lf : s_gk + (row - 16) * Lf;
    float s = 0.f;
    for (int j = part; j < d.m; j += 8) s += g[j];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) s_st[32 + row] = s;
  }
  __syncthreads();
  MLHOT_TSTAMP(134);
  if (tid < d.Nq) s_gq[tid * Lf + a.arg_q[(t * d.Nq + tid) * H + h]] -= s_st[32 + tid];
  if (tid == 0) {
    float s = 0.f;
    for (int np = 0; np < d.Nc; ++np) s += s_st[48 + np];
    a.part_k[t * H + h] = s;
  }
  __syncthreads();
  MLHOT_TSTAMP(135);
  // dx[row][e] = sum_j d(dd)[row][j] pc[j][e] - rsum[row] c^2 x[row][e]: waves 0,1 -> q (e tiles 0..), 2,3 -> k
  {
    const float c2 = 1.0f / sqrtf((float)d.dw);
    const int net = d.dw / 16;
    for (int it = wave; it < 2 * net; it += 4) {
      const int isk = it >= net, et = isk ? it - net : it;
      lcptr g = isk ? s_gk : s_gq;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int j0 = 0; j0 < d.m; j0 += 4) {
        const int j = j0 + lq;
        const float bv = j < d.m ? a.pc[(size_t)j * d.dw + et * 16 + lr] : 0.f;
        acc = mfma4(g[lr * Lf + j], bv, acc);
      }
      lcptr xs = isk ? s_k : s_q;
      float* dst = isk ? a.dkh : a.dqh;
      const int nrows = isk ? d.Nc : d.Nq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r, e = et * 16 + lr;
        if (row < nrows)
          dst[(size_t)(t * nrows + row) * HD + h * d.dw + e] = acc[r] - s_st[32 + (isk ? 16 : 0) + row] * c2 * xs[row * Lx + e];
      }
    }
  }
  MLHOT_TSTAMP(136);
}
__host__ inline size_t phaseB_bwd_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (4 * ldpad(d.dw) + 4 * ldpad(d.m)) + 2 * 16 * 17 + 64);
}

// ==================================================================================================
// phase A backward, one workgroup per task: global key arg-max correction, W_q / W_v / W_k backward,
// EncoderFC backward, transform_y weight gradient.
//   out: d_dec_in[:, :dw] += d x_qry (attention share), d_cat_in, slab entries
// ==================================================================================================
struct PhaseABwdArgs {
  TailDims d; TailParams p; TailSlab sl;
  const float *ctx_y, *cat_in, *h0, *h1, *rs, *dec_in, *dqh, *dkh, *dvh, *pc, *part_k; const int* gpos;
  float *d_dec_in, *d_cat_in, *slab;
};

__global__ __launch_bounds__(512) void phaseA_bwd_kernel(const PhaseABwdArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz, HD = H * d.dw;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lw = ldpad(d.dw), Lhd = ldpad(HD), Ly = ldpad(d.label_dim);
  lptr s_cat = L0;                  // saved activations
  lptr s_h0 = s_cat + 16 * Lcat;
  lptr s_h1 = s_h0 + 16 * Lh0;
  lptr s_rs = s_h1 + 16 * Lh1;
  lptr s_xq = s_rs + 16 * Lw;
  lptr s_y = s_xq + 16 * Lw;
  lptr s_dq = s_y + 16 * Ly;         // [16][Lhd] head-space gradients (one buffer, reused q -> v -> k)
  lptr s_drs = s_dq + 16 * Lhd;      // gradients
  lptr s_dxc = s_drs + 16 * Lw;
  lptr s_dh1 = s_dxc + 16 * Lw;
  lptr s_dh0 = s_dh1 + 16 * Lh1;
  lptr s_dcat = s_dh0 + 16 * Lh0;
  lptr s_red = s_dcat + 16 * Lcat;   // [8 waves][256] partial tiles of wg_dgrad
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  MLHOT_TSTAMP(160);
  lds_zero(L0, 16 * (2 * Lcat + 2 * Lh0 + 2 * Lh1 + 4 * Lw + Ly + Lhd), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(161);
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  lds_load(s_cat, Lcat, a.cat_in + rc * ldc, ldc, d.Nc, ldc, tid, 512);
  lds_load(s_h0, Lh0, a.h0 + rc * d.h0, d.h0, d.Nc, d.h0, tid, 512);
  lds_load(s_h1, Lh1, a.h1 + rc * d.h1, d.h1, d.Nc, d.h1, tid, 512);
  lds_load(s_rs, Lw, a.rs + rc * d.dw, d.dw, d.Nc, d.dw, tid, 512);
  lds_load(s_xq, Lw, a.dec_in + rq * ldd, ldd, d.Nq, d.dw, tid, 512);
  lds_load(s_y, Ly, a.ctx_y + rc * d.label_dim, d.label_dim, d.Nc, d.label_dim, tid, 512);
  lds_load(s_dq, Lhd, a.dqh + rq * HD, HD, d.Nq, HD, tid, 512);
  gptr sl = G(a.slab) + (size_t)t * a.sl.total;
  __syncthreads();
  MLHOT_TSTAMP(162);
  // W_q: weight gradient and the attention share of d x_qry (accumulated onto the decoder's)
  wg_wgrad<8>(s_dq, Lhd, HD, s_xq, Lw, d.dw, sl + a.sl.wq_w, sl + a.sl.wq_b, wave, lane, tid);
  MLHOT_TSTAMP(163);
  wg_dgrad<8>(s_dq, Lhd, HD, WB1N(wq_w, d.dw), d.dw, nullptr, 0, G(a.d_dec_in + rq * ldd), ldd, d.Nq, true, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(164);
  // W_v
  lds_zero(s_dq, 16 * Lhd, tid, 512);
  __syncthreads();
  lds_load(s_dq, Lhd, a.dvh + rc * HD, HD, d.Nc, HD, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(165);
  wg_wgrad<8>(s_dq, Lhd, HD, s_rs, Lw, d.dw, sl + a.sl.wv_w, sl + a.sl.wv_b, wave, lane, tid);
  MLHOT_TSTAMP(166);
  wg_dgrad<8>(s_dq, Lhd, HD, WB1N(wv_w, d.dw), d.dw, s_drs, Lw, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(167);
  // W_k, with the batch-global key arg-max correction: that ONE element's d(dd) carries minus the
  // sum of G over every key row of the batch (fast_attention.py:97), i.e. dk[row] -= total * pc[col]
  lds_zero(s_dq, 16 * Lhd, tid, 512);
  __syncthreads();
  lds_load(s_dq, Lhd, a.dkh + rc * HD, HD, d.Nc, HD, tid, 512);
  __syncthreads();
  {
    const int grow = a.gpos[0], gcol = a.gpos[1];
    const int gt = grow / (d.Nc * H);
    if (gt == t && tid < d.dw) {
      float total = 0.f;
      for (int i = 0; i < d.T * H; ++i) total += a.part_k[i];
      const int n = (grow / H) % d.Nc, hh = grow % H;
      s_dq[n * Lhd + hh * d.dw + tid] -= total * a.pc[(size_t)gcol * d.dw + tid];
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(168);
  wg_wgrad<8>(s_dq, Lhd, HD, s_cat, Lcat, d.dw, sl + a.sl.wk_w, sl + a.sl.wk_b, wave, lane, tid);
  MLHOT_TSTAMP(169);
  wg_dgrad<8>(s_dq, Lhd, HD, WB1N(wk_w, d.dw), d.dw, s_dxc, Lw, nullptr, 0, 0, false, s_red, wave, lane);
  MLHOT_TSTAMP(170);
  // EncoderFC, last layer first
  wg_wgrad<8>(s_drs, Lw, d.dw, s_h1, Lh1, d.h1, sl + a.sl.er_w[2], sl + a.sl.er_b[2], wave, lane, tid);
  wg_dgrad<8>(s_drs, Lw, d.dw, WB1N(er_w[2], d.dw), d.h1, s_dh1, Lh1, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(171);
  lds_actgrad(s_dh1, Lh1, s_h1, Lh1, d.h1, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh1, Lh1, d.h1, s_h0, Lh0, d.h0, sl + a.sl.er_w[1], sl + a.sl.er_b[1], wave, lane, tid);
  wg_dgrad<8>(s_dh1, Lh1, d.h1, WB1N(er_w[1], d.h1), d.h0, s_dh0, Lh0, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(172);
  lds_actgrad(s_dh0, Lh0, s_h0, Lh0, d.h0, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh0, Lh0, d.h0, s_cat, Lcat, ldc, sl + a.sl.er_w[0], sl + a.sl.er_b[0], wave, lane, tid);
  wg_dgrad<8>(s_dh0, Lh0, d.h0, WB1N(er_w[0], d.h0), ldc, s_dcat, Lcat, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(173);
  // d_cat_in = EncoderFC input gradient (+ K-projection share on the x_ctx columns)
  for (int i = tid; i < d.Nc * ldc; i += 512) {
    const int r = i / ldc, c = i % ldc;
    a.d_cat_in[(rc + r) * ldc + c] = s_dcat[r * Lcat + c] + (c < d.dw ? s_dxc[r * Lw + c] : 0.f);
  }
  // transform_y: dW = d_cat[:, dw:]^T ctx_y, db
  wg_wgrad<8>(s_dcat + d.dw, Lcat, d.dw / 4, s_y, Ly, d.label_dim, sl + a.sl.ty_w, sl + a.sl.ty_b, wave, lane, tid);
  MLHOT_TSTAMP(174);
}
__host__ inline size_t phaseA_bwd_lds_bytes(const TailDims& d) {
  const int ldc = d.dw + d.dw / 4;
  return sizeof(float) * (16 * (2 * ldpad(ldc) + 2 * ldpad(d.h0) + 2 * ldpad(d.h1) + 4 * ldpad(d.dw) + ldpad(d.label_dim) + ldpad(H * d.dw)) + 8 * 256 +
                          ptab_floats<TailParams>());
}

// ---- sum the per-task slabs into the parameter gradients (fixed task order) -------------------------
constexpr int MAX_SEG = 72;
struct SlabReduce {
  float* dst[MAX_SEG]; int off[MAX_SEG]; int len[MAX_SEG];
  int nseg, T, total; const float* slab;
};
// One flat pass over the slab: a thread sums float4 `q` of all T task slabs (T loads in flight), finds
// the parameter segment the float4 belongs to (segments start and end on multiples of 4 floats, so a
// float4 never straddles two) and stores into that parameter's gradient.  grid = ceil(total / 1024).
__global__ __launch_bounds__(256) void slab_to_grads_kernel(const SlabReduce a) {
  __shared__ float* s_dst[MAX_SEG];
  __shared__ int s_off[MAX_SEG], s_len[MAX_SEG];
  {
    // compare chain instead of a.dst[tid]: a runtime-indexed by-value kernel argument would be copied to scratch
    float* dst = nullptr; int off = 0, len = 0;
#pragma unroll
    for (int i = 0; i < MAX_SEG; ++i)
      if (i == (int)threadIdx.x) { dst = a.dst[i]; off = a.off[i]; len = a.len[i]; }
    if (threadIdx.x < MAX_SEG) { s_dst[threadIdx.x] = dst; s_off[threadIdx.x] = off; s_len[threadIdx.x] = len; }
  }
  __syncthreads();
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= a.total) return;
  int seg = -1;
  for (int i = 0; i < a.nseg; ++i)
    if (e >= s_off[i] && e < s_off[i] + s_len[i]) seg = i;
  if (seg < 0) return;                              // alignment padding between two segments
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int t = 0; t < a.T; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(a.slab + (size_t)t * a.total + e);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* d = s_dst[seg] + (e - s_off[seg]);
  const int left = s_len[seg] - (e - s_off[seg]);   // a segment's length need not be a multiple of 4
  d[0] = s.x;
  if (left > 1) d[1] = s.y;
  if (left > 2) d[2] = s.z;
  if (left > 3) d[3] = s.w;
}

}  // namespace tf
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
