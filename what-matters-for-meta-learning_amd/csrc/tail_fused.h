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
// stays as the fallback for larger shot counts and as the A/B reference of a whole forward+backward
// (mlhot_set_option("tail_fused", 0)).  The fused backward consumes what the fused forward left behind
// (packed key arg-max, head-major copy of _W's weight), so one step runs either flavour end to end.
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
#define MLHOT_TSTAMP_AT(i, blk) do { if (g_ts_dev && (int)blockIdx.x == (blk) && threadIdx.x == 0) g_ts_dev[i] = wall_clock64(); } while (0)
// inside a layer function: cycle stamps of call number g_ts_dev[199] (slots 200 + 8 * call + i)
#define MLHOT_TSCALL_BEGIN() long long* tsc_ = nullptr; do { if (g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0) { \
    const long long c_ = g_ts_dev[199]; g_ts_dev[199] = c_ + 1; if (c_ < 36) tsc_ = g_ts_dev + 200 + 8 * c_; } } while (0)
#define MLHOT_TSC(i) do { if (tsc_) tsc_[i] = clock64(); } while (0)
#else
#define MLHOT_TSTAMP(i) do {} while (0)
#define MLHOT_TSTAMP_AT(i, blk) do {} while (0)
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
template <int NW, int TPW>
__device__ __attribute__((noinline)) void wg_linear_t(lcptr xs, int ldx, int K, WB wb, int N, int act,
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
  constexpr int tpw = TPW;                        // N-tiles a wave handles together; 16 / TPW k blocks of weights in flight per trip
  constexpr int KBT = 16 / TPW;
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
    f32x4_t acc[TPW];
    gcptr wbase[TPW]; int woff[TPW]; float bias[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
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
    for (int kb = kb0; kb < kb1; kb += KBT) {
      f32x4_t b[KBT][TPW];
#pragma unroll
      for (int u = 0; u < KBT; ++u) {
        if (kb + u < kb1) {
          const int k16 = (kb + u) * 16;
          const int kc = k16 + 4 * lq <= kmax ? k16 : kmax - 4 * lq;       // clamp this lane's 4 k's into the row
#pragma unroll
          for (int q = 0; q < TPW; ++q) {
            if (vec) b[u][q] = *reinterpret_cast<gc4ptr>(wbase[q] + woff[q] + kc);
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) { const int ko = woff[q] + k16 + e; b[u][q][e] = wbase[q][k16 + 4 * lq + e <= kmax ? ko : woff[q] - 4 * lq + kmax]; }
            }
          }
        }
      }
#pragma unroll
      for (int u0 = 0; u0 < KBT; u0 += 4) {
        f32x4_t a4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (u0 + v < KBT && kb + u0 + v < kb1) a4[v] = lds_read4(xrow + (kb + u0 + v) * 16, avec);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          if (u0 + v < KBT && kb + u0 + v < kb1) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
              acc[q] = mfma4(a4[v][0], b[u0 + v][q][0], acc[q]);
              acc[q] = mfma4(a4[v][1], b[u0 + v][q][1], acc[q]);
              acc[q] = mfma4(a4[v][2], b[u0 + v][q][2], acc[q]);
              acc[q] = mfma4(a4[v][3], b[u0 + v][q][3], acc[q]);
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
      for (int q = 0; q < TPW; ++q) {
        if (tile0 + q >= ntile) continue;
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

// dispatcher: N-tiles per wave (1 when the K range is split over idle waves, else up to 4), see wg_linear_t
template <int NW>
__device__ __forceinline__ void wg_linear(lcptr xs, int ldx, int K, WB wb, int N, int act,
                                          lptr ys, int ldy, gptr yg, int ldg, int nrows, lptr red, int wave, int lane) {
  const int ntile = (uni(N) + 15) >> 4;
  const bool split = uni(red != nullptr) && ntile * 2 <= NW;
  int tpw = (ntile + NW - 1) / NW;
  if (split || tpw < 2) wg_linear_t<NW, 1>(xs, ldx, K, wb, N, act, ys, ldy, yg, ldg, nrows, red, wave, lane);
  else if (tpw == 2) wg_linear_t<NW, 2>(xs, ldx, K, wb, N, act, ys, ldy, yg, ldg, nrows, red, wave, lane);
  else wg_linear_t<NW, 4>(xs, ldx, K, wb, N, act, ys, ldy, yg, ldg, nrows, red, wave, lane);
}

// zero a [16 x ld] LDS tile
__device__ __forceinline__ void lds_zero(lptr p, int n, int tid, int nthreads) {
  for (int i = tid; i < n; i += nthreads) p[i] = 0.f;
}
// global [nrows x width] (row stride ldg) -> LDS [16 x ld] (rows >= nrows left as they are)
// A wave walks whole rows (no division per element, coalesced), 4 rows x 4 column chunks = up to 16 loads in
// flight per lane before the first LDS store; out-of-range slots load a clamped address and are not stored.
// cstride: distance between consecutive columns in the source (1 = dense rows).
template <class SrcPtr>
__device__ __forceinline__ void lds_load(lptr dst, int ld, SrcPtr src, int ldg, int nrows, int width, int tid, int nthreads, int cstride = 1) {
  const int lane = tid & 63, nw = nthreads >> 6;
  for (int r0 = tid >> 6; r0 < nrows; r0 += 4 * nw)
    for (int c0 = lane; c0 < width; c0 += 256) {
      float v[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + k * nw, c = c0 + 64 * u;
          const bool ok = r < nrows && c < width;
          v[k][u] = src[ok ? (size_t)r * ldg + (size_t)c * cstride : 0];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = r0 + k * nw, c = c0 + 64 * u;
          if (r < nrows && c < width) dst[r * ld + c] = v[k][u];
        }
    }
}

// Several tiles at once: every job's loads are issued before the first LDS store, so a kernel prologue
// pays ONE memory round trip instead of one per tile.  The register window covers rows wave + k*nw (k < KR)
// and columns lane + 64u (u < KU) of every job; whatever a job has beyond it goes through lds_load.
struct LoadJob { lptr dst; int ld; const float* src; int ldg, nrows, width, cstride; };
__device__ __forceinline__ LoadJob load_job(lptr dst, int ld, const float* src, int ldg, int nrows, int width, int cstride = 1) {
  return LoadJob{dst, ld, src, ldg, nrows, width, cstride};
}
template <int NJ, int KR, int KU>
struct LoadBatch {
  float v[NJ][KR][KU];
  __device__ __forceinline__ void fetch(const LoadJob (&jobs)[NJ], int tid, int nw) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int r = w + k * nw, c = lane + 64 * u;
          const bool ok = r < jobs[j].nrows && c < jobs[j].width;
          v[j][k][u] = jobs[j].src[ok ? (size_t)r * jobs[j].ldg + (size_t)c * jobs[j].cstride : 0];
        }
  }
  __device__ __forceinline__ void stash(const LoadJob (&jobs)[NJ], int tid, int nw) {
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
      for (int k = 0; k < KR; ++k)
#pragma unroll
        for (int u = 0; u < KU; ++u) {
          const int r = w + k * nw, c = lane + 64 * u;
          if (r < jobs[j].nrows && c < jobs[j].width) jobs[j].dst[r * jobs[j].ld + c] = v[j][k][u];
        }
      if (jobs[j].nrows > KR * nw || jobs[j].width > 64 * KU) {          // beyond the window (large shapes): plain loops
        for (int r = w; r < jobs[j].nrows; r += nw)
          for (int c = lane; c < jobs[j].width; c += 64)
            if (r >= KR * nw || c >= 64 * KU) jobs[j].dst[r * jobs[j].ld + c] = jobs[j].src[(size_t)r * jobs[j].ldg + (size_t)c * jobs[j].cstride];
      }
    }
  }
};

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
// phase A forward, ONE launch with two kinds of 512-thread workgroups:
//   blocks [0, T)         one per task: cat_in[:, dw:] = transform_y(ctx_y); h0, h1 = EncoderFC hidden; rs;
//                         the task's slice of pc = c * P
//   blocks [T, T + T*H)   one per (task, head): kh = W_k,h(x_ctx) and this head's share of the batch-global
//                         key stabiliser, max / first arg-max of ddk = c * kh . P[j]  (fast_attention.py:97)
// The key-head blocks need nothing from the chain, so the 16-CU-wide dependent chain and the 128 blocks
// of projection / feature-map work run side by side (they used to be serial stages of the task block).
// The query and value projections moved into phase B.
// ==================================================================================================
struct PhaseAArgs {
  int dbg;              // timing experiments: early exits (results become wrong)
  TailDims d; TailParams p;
  const float* ctx_y;
  float *cat_in, *h0, *h1, *rs, *dec_in, *kh;             // saved activations (global)
  float* pc;                                              // [m][dw]  c * projection
  float* tmax; int* targ;                                 // per (task, head): max of ddk, packed (row * 4096 + col)
  float* wot;                                             // [H][dw][dw] head-major copy of _W's weight: wot[h][j][e] = Wo[j][e*H + h]
  // The encoder Linear's split-K partial results [xk][xn][dw] + its bias, when the encoder left its fold to this phase (tail_spec.h:
  // the image features arrive as partial sums, phase A's blocks fold the tiles they read anyway and T extra blocks fold the query
  // rows); nullptr: cat_in[:, :dw] / dec_in[:, :dw] already hold the features.
  const float* xslab; const float* xbias; int xk, xn;
};

// (value, packed position): larger value first, then the smaller position (row-major first occurrence)
__device__ __forceinline__ bool kmax_better(float v, int code, float bv, int bcode) { return v > bv || (v == bv && code < bcode); }

__device__ __forceinline__ void phaseA_keyhead(const PhaseAArgs& a, lptr L0, int th, int tid) {
  const TailDims& d = a.d;
  const int t = th / H, h = th - t * H, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const int ldc = d.dw + d.dw / 4, HD = H * d.dw, Lx = ldpad(d.dw);
  lptr s_x = L0;                      // [16][Lx] x_ctx
  lptr s_k = s_x + 16 * Lx;           // [16][Lx] kh
  lptr s_red = s_k + 16 * Lx;         // [8] max, [8] packed position
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 16);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 32 * Lx, tid, 512);
  if (t == 0) {
    // head-major copy of _W's weight for phase B (forward and backward): the head-merge interleaves the heads in the
    // column index (e*H + h), so a head's slice is a stride-H gather - done once here instead of in all T*H blocks
    for (int i = tid; i < d.dw * d.dw; i += 512) {
      const int j = i / d.dw, e = i - j * d.dw;
      a.wot[(size_t)h * d.dw * d.dw + i] = a.p.wo_w[(size_t)j * HD + e * H + h];
    }
  }
  __syncthreads();
  lds_load(s_x, Lx, a.cat_in + (size_t)t * d.Nc * ldc, ldc, d.Nc, d.dw, tid, 512);
  __syncthreads();
  // kh tile: waves 0 .. dw/16-1, one 16-column tile each; the others go straight to the feature tiles
  if (wave * 16 < d.dw) {
    gcptr wk = uniptr(ptab[offsetof(PRM, wk_w) / 8 + h]);
    gcptr bk = uniptr(ptab[offsetof(PRM, wk_b) / 8 + h]);
    const int n = wave * 16 + lr;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k64 = 0; k64 < d.dw; k64 += 64) {          // dw % 64 == 0
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k64 + 16 * u + 4 * lq;
        const f32x4_t b = *reinterpret_cast<gc4ptr>(wk + n * d.dw + kk);
        lcptr xp = s_x + lr * Lx + kk;
        acc = mfma4(xp[0], b[0], acc); acc = mfma4(xp[1], b[1], acc);
        acc = mfma4(xp[2], b[2], acc); acc = mfma4(xp[3], b[3], acc);
      }
    }
    const float bias = bk[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      if (row < d.Nc) {                 // rows >= Nc stay zero: they must not enter the max below
        s_k[row * Lx + n] = acc[r] + bias;
        a.kh[(size_t)(t * d.Nc + row) * HD + h * d.dw + n] = acc[r] + bias;
      }
    }
  }
  __syncthreads();
  const float c = powf((float)d.dw, -0.25f);
  float best = -INFINITY; int bcode = 0x7fffffff;
  const int ntile = (d.m + 15) / 16;
  for (int jt = wave; jt < ntile; jt += 8) {
    const int j = jt * 16 + lr;
    const int jc = j < d.m ? j : d.m - 1;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k64 = 0; k64 < d.dw; k64 += 64) {          // dw % 64 == 0: the constant inner trip count lets the 4 loads go out together
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k64 + 16 * u + 4 * lq;
        const float4 b = *reinterpret_cast<const float4*>(a.p.proj + (size_t)jc * d.dw + kk);
        lcptr xp = s_k + lr * Lx + kk;
        acc = mfma4(xp[0], b.x * c, acc); acc = mfma4(xp[1], b.y * c, acc);
        acc = mfma4(xp[2], b.z * c, acc); acc = mfma4(xp[3], b.w * c, acc);
      }
    }
    if (j < d.m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        if (row < d.Nc) {
          const int code = ((t * d.Nc + row) * H + h) * 4096 + j;      // row index of the [T*Nc*H, m] view, column
          if (kmax_better(acc[r], code, best, bcode)) { best = acc[r]; bcode = code; }
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oc = __shfl_xor(bcode, off, 64);
    if (kmax_better(ov, oc, best, bcode)) { best = ov; bcode = oc; }
  }
  MLHOT_LDS int* s_redi = reinterpret_cast<MLHOT_LDS int*>(s_red + 8);
  if (lane == 0) { s_red[wave] = best; s_redi[wave] = bcode; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 8; ++w)
      if (kmax_better(s_red[w], s_redi[w], best, bcode)) { best = s_red[w]; bcode = s_redi[w]; }
    a.tmax[th] = best; a.targ[th] = bcode;
  }
}

__global__ __launch_bounds__(512) void phaseA_fwd_kernel(const PhaseAArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= d.T) { phaseA_keyhead(a, L0, blockIdx.x - d.T, tid); return; }
  const int t = blockIdx.x;
  const int ldc = d.dw + d.dw / 4;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lrs = ldpad(d.dw), Ly = ldpad(d.label_dim);
  lptr s_cat = L0;                    // [16][Lcat]
  lptr s_h0 = s_cat + 16 * Lcat;
  lptr s_h1 = s_h0 + 16 * Lh0;
  lptr s_rs = s_h1 + 16 * Lh1;
  lptr s_y = s_rs + 16 * Lrs;
  lptr s_red = s_y + 16 * Ly;          // [8 waves][256] K-split partials of wg_linear
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  const int total = 16 * (Lcat + Lh0 + Lh1 + Lrs + Ly) + 8 * 256;
  MLHOT_TSTAMP(0);
  lds_zero(L0, total, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(1);
  gptr g_cat = G(a.cat_in) + (size_t)t * d.Nc * ldc;
  {
    const LoadJob xj[2] = {
        load_job(s_cat, Lcat, a.cat_in + (size_t)t * d.Nc * ldc, ldc, d.Nc, d.dw),                       // x_ctx (encoder output)
        load_job(s_y, Ly, a.ctx_y + (size_t)t * d.Nc * d.label_dim, d.label_dim, d.Nc, d.label_dim)};
    LoadBatch<2, 2, 1> xb;
    xb.fetch(xj, tid, 8);
    xb.stash(xj, tid, 8);
  }
  {   // pc = c * P for the later phases: each task writes its slice
    const float c = powf((float)d.dw, -0.25f);
    const int n = d.m * d.dw, per = (n + d.T - 1) / d.T, lo = t * per, hi = lo + per < n ? lo + per : n;
    for (int i = lo + tid; i < hi; i += 512) a.pc[i] = c * a.p.proj[i];
  }
  __syncthreads();
  MLHOT_TSTAMP(2);
  if (a.dbg & 2) return;
  // transform_y -> cat[:, dw:]
  wg_linear<8>(s_y, Ly, d.label_dim, WB1(ty_w, ty_b, d.dw / 4), d.dw / 4, ACT_NONE, s_cat + d.dw, Lcat, g_cat + d.dw, ldc, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(3);
  wg_linear<8>(s_cat, Lcat, ldc, WB1(er_w[0], er_b[0], d.h0), d.h0, ACT_RELU, s_h0, Lh0, G(a.h0 + (size_t)t * d.Nc * d.h0), d.h0, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(4);
  wg_linear<8>(s_h0, Lh0, d.h0, WB1(er_w[1], er_b[1], d.h1), d.h1, ACT_RELU, s_h1, Lh1, G(a.h1 + (size_t)t * d.Nc * d.h1), d.h1, d.Nc, nullptr, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(5);
  wg_linear<8>(s_h1, Lh1, d.h1, WB1(er_w[2], er_b[2], d.dw), d.dw, ACT_NONE, nullptr, 0, G(a.rs + (size_t)t * d.Nc * d.dw), d.dw, d.Nc, s_red, wave, lane);
  MLHOT_TSTAMP(6);
}

__host__ inline size_t phaseA_lds_bytes(const TailDims& d) {
  const int ldc = d.dw + d.dw / 4;
  const size_t chain = 16 * (ldpad(ldc) + ldpad(d.h0) + ldpad(d.h1) + ldpad(d.dw) + ldpad(d.label_dim)) + 8 * 256;
  const size_t keyhead = 32 * ldpad(d.dw) + 16;
  return sizeof(float) * ((chain > keyhead ? chain : keyhead) + ptab_floats<TailParams>());
}

// ==================================================================================================
// phase B forward, one workgroup (256 threads) per (task, head): FAVOR+ in the S-form
//   dd = x pc^T;  E = ratio exp(dd - diag - stab)  (stab: row max for q, batch-global max for k);
//   S = (Eq + re)(Ek + re)^T masked to valid rows;  D = rowsum S;  out = S V / D.
// Fills the same workspace fields as the generic path (qf, kf, S, D, arg_q, gmax, gpos).
// ==================================================================================================
struct PhaseBArgs {
  TailDims d; TailParams p;
  const float *dec_in, *rs;                // x_qry = dec_in[:, :dw] (row stride dw + dz), rs [T*Nc][dw]
  float *qh, *vh;                          // [T*N][H*dw] rows: written here (saved for the backward)
  const float *kh, *pc;                    // kh from phase A's key-head blocks, pc [m][dw]
  const float* tmax; const int* targ;      // per (task, head) key max shares
  float *qf, *kf, *S, *D, *gmax; int *arg_q, *gpos;
  float* merged;                           // [T*Nq][dw*H], column e*H + h
  float* rrp;                              // [T*H][Nq][dw]: this head's share of _W(merged) (summed by phase C)
  const float* wot;                        // [H][dw][dw] head-major _W weight (phase A)
};

__global__ __launch_bounds__(512) void phaseB_fwd_kernel(const PhaseBArgs a) {
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
  lptr s_S = s_kf + 16 * Lf;         // [8 waves][16][17] partials, then final in wave 0's slot
  lptr s_st = s_S + 8 * 16 * 17;     // diag_q[16], diag_k[16], max_q[16], D[16]
  MLHOT_LDS int* s_arg = reinterpret_cast<MLHOT_LDS int*>(s_st + 64);   // arg_q[16]
  lptr s_xq = s_st + 64 + 16;        // [16][Lx] x_qry
  lptr s_rs = s_xq + 16 * Lx;        // [16][Lx] rs
  const int total = 16 * (5 * Lx + 2 * Lf) + 8 * 16 * 17 + 64 + 16;
  lds_zero(L0, total, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(33);
  const int HD = H * d.dw;
  {
    const LoadJob xj[3] = {
        load_job(s_xq, Lx, a.dec_in + (size_t)t * d.Nq * (d.dw + d.dz), d.dw + d.dz, d.Nq, d.dw),
        load_job(s_rs, Lx, a.rs + (size_t)t * d.Nc * d.dw, d.dw, d.Nc, d.dw),
        load_job(s_k, Lx, a.kh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw)};
    LoadBatch<3, 2, 1> xb;
    xb.fetch(xj, tid, 8);
    xb.stash(xj, tid, 8);
  }
  // batch-global key stabiliser (identical in every workgroup): largest share, first position on ties
  float gm = -INFINITY; int gcode = 0x7fffffff;
  for (int i = tid; i < d.T * H; i += 512) {
    const float v = a.tmax[i]; const int cd = a.targ[i];
    if (kmax_better(v, cd, gm, gcode)) { gm = v; gcode = cd; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(gm, off, 64);
    const int oc = __shfl_xor(gcode, off, 64);
    if (kmax_better(ov, oc, gm, gcode)) { gm = ov; gcode = oc; }
  }
  if (lane == 0) { s_st[wave] = gm; s_arg[wave] = gcode; }
  __syncthreads();
  gm = s_st[0]; gcode = s_arg[0];
#pragma unroll
  for (int w = 1; w < 8; ++w)
    if (kmax_better(s_st[w], s_arg[w], gm, gcode)) { gm = s_st[w]; gcode = s_arg[w]; }
  if (blockIdx.x == 0 && tid == 0) { a.gmax[0] = gm; a.gpos[0] = gcode >> 12; a.gpos[1] = gcode & 4095; }
  __syncthreads();
  MLHOT_TSTAMP(34);
  // this head's query and value projections: qh = W_q,h(x_qry), vh = W_v,h(rs); one 16-column tile per wave and trip
  {
    const float *wq = a.p.wq_w[0], *bq = a.p.wq_b[0], *wv = a.p.wv_w[0], *bv = a.p.wv_b[0];
#pragma unroll
    for (int i = 1; i < H; ++i)
      if (h == i) { wq = a.p.wq_w[i]; bq = a.p.wq_b[i]; wv = a.p.wv_w[i]; bv = a.p.wv_b[i]; }
    const int ntq = d.dw / 16;
    for (int it = wave; it < 2 * ntq; it += 8) {        // items: (query | value) x 16-column tile
      const bool isv = it >= ntq;
      const int n = (isv ? it - ntq : it) * 16 + lr;
      const float* wsel = isv ? wv : wq;
      lcptr xs = isv ? s_rs : s_xq;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      for (int k64 = 0; k64 < d.dw; k64 += 64) {        // dw % 64 == 0
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kk = k64 + 32 * u + 4 * lq;
          const float4 b1 = *reinterpret_cast<const float4*>(wsel + (size_t)n * d.dw + kk);
          const float4 b2 = *reinterpret_cast<const float4*>(wsel + (size_t)n * d.dw + kk + 16);
          lcptr x1 = xs + lr * Lx + kk;
          acc0 = mfma4(x1[0], b1.x, acc0); acc0 = mfma4(x1[1], b1.y, acc0);
          acc0 = mfma4(x1[2], b1.z, acc0); acc0 = mfma4(x1[3], b1.w, acc0);
          acc1 = mfma4(x1[16], b2.x, acc1); acc1 = mfma4(x1[17], b2.y, acc1);
          acc1 = mfma4(x1[18], b2.z, acc1); acc1 = mfma4(x1[19], b2.w, acc1);
        }
      }
      const float bias = (isv ? bv : bq)[n];
      lptr ys = isv ? s_v : s_q;
      float* yg = isv ? a.vh : a.qh;
      const int nrows = isv ? d.Nc : d.Nq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        if (row < nrows) {
          ys[row * Lx + n] = acc0[r] + acc1[r] + bias;
          yg[(size_t)(t * nrows + row) * HD + h * d.dw + n] = acc0[r] + acc1[r] + bias;
        }
      }
    }
  }
  __syncthreads();
  // dd tiles: q and k against pc; a feature tile's pc slice is loaded once (all float4 in flight
  // together) and feeds both the query and the key accumulator
  const int ntile = (d.m + 15) / 16;
  for (int jt = wave; jt < ntile; jt += 8) {
    const int j = jt * 16 + lr;
    const bool vj = j < d.m;
    f32x4_t accq = {0.f, 0.f, 0.f, 0.f}, acck = {0.f, 0.f, 0.f, 0.f};
    for (int k64 = 0; k64 < d.dw; k64 += 64) {          // dw % 64 == 0
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int kk = k64 + 16 * u + 4 * lq;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vj) b = *reinterpret_cast<const float4*>(a.pc + (size_t)j * d.dw + kk);
        lcptr xq = s_q + lr * Lx + kk;
        lcptr xk = s_k + lr * Lx + kk;
        accq = mfma4(xq[0], b.x, accq); acck = mfma4(xk[0], b.x, acck);
        accq = mfma4(xq[1], b.y, accq); acck = mfma4(xk[1], b.y, acck);
        accq = mfma4(xq[2], b.z, accq); acck = mfma4(xk[2], b.z, acck);
        accq = mfma4(xq[3], b.w, accq); acck = mfma4(xk[3], b.w, acck);
      }
    }
    if (vj) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_qf[(4 * lq + r) * Lf + j] = accq[r]; s_kf[(4 * lq + r) * Lf + j] = acck[r]; }
    }
  }
  // diag = c^2/2 |x|^2 : 32 rows (16 q + 16 k), 16 threads per row
  {
    const float half_c2 = 0.5f / sqrtf((float)d.dw);
    const int row = tid >> 4, part = tid & 15;
    lcptr xr = (row < 16 ? s_q + row * Lx : s_k + (row - 16) * Lx);
    float s = 0.f;
    for (int e = part; e < d.dw; e += 16) s += xr[e] * xr[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if (part == 0) s_st[row] = s * half_c2;
  }
  __syncthreads();
  MLHOT_TSTAMP(35);
  // query row max / first arg-max: 16 rows x 32 threads
  {
    const int row = tid >> 5, part = tid & 31;
    float best = -INFINITY; int arg = 0x7fffffff;
    for (int j = part; j < d.m; j += 32) { const float v = s_qf[row * Lf + j]; if (v > best) { best = v; arg = j; } }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const float ov = __shfl_xor(best, off, 64); const int oa = __shfl_xor(arg, off, 64);
      if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (part == 0) { s_st[32 + row] = best; s_arg[row] = arg; }
  }
  __syncthreads();
  MLHOT_TSTAMP(36);
  // E features in place (padding columns j >= m stay exactly 0 -> they are skipped below via `re` masking)
  const float ratio = 1.0f / sqrtf((float)d.m), re = ratio * 1e-4f;
  // a wave owns rows wave, wave + 8: E in place, and the valid rows saved for the backward right away
  // (rows of the [T*N*H, m] views)
  for (int row = wave; row < 16; row += 8) {
    const float sq = s_st[row] + s_st[32 + row], sk = s_st[16 + row] + gm;
    float* gq = a.qf + ((size_t)(t * d.Nq + row) * H + h) * d.m;
    float* gk = a.kf + ((size_t)(t * d.Nc + row) * H + h) * d.m;
    for (int j = lane; j < d.m; j += 64) {
      const float eq = ratio * expf(s_qf[row * Lf + j] - sq), ek = ratio * expf(s_kf[row * Lf + j] - sk);
      s_qf[row * Lf + j] = eq;
      s_kf[row * Lf + j] = ek;
      if (row < d.Nq) gq[j] = eq;
      if (row < d.Nc) gk[j] = ek;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(37);
  if (tid < d.Nq) a.arg_q[(t * d.Nq + tid) * H + h] = s_arg[tid];
  // S = (Eq + re)(Ek + re)^T : M = 16 q rows, N = 16 k rows, K = m split over the 8 waves
  {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = wave * 4; j0 < d.m; j0 += 32) {
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
    const int n = (tid >> 4) & 15, np = tid & 15;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += s_S[(16 * w + n) * 17 + np];
    if (n >= d.Nq || np >= d.Nc) s = 0.f;
    __syncthreads();
    if (tid < 256) {
      s_S[n * 17 + np] = s;
      if (n < d.Nq && np < d.Nc) a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + np] = s;
    }
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
  for (int et = wave; et * 16 < d.dw; et += 8) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int np = 4 * s4 + lq;
      acc = mfma4(s_S[lr * 17 + np], s_v[np * Lx + et * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, e = et * 16 + lr;
      const float o = n < d.Nq ? acc[r] / s_st[48 + n] : 0.f;
      if (n < d.Nq) a.merged[(size_t)(t * d.Nq + n) * (d.dw * H) + e * H + h] = o;
      s_xq[n * Lx + e] = o;                          // x_qry is no longer needed: the tile now holds this head's output
    }
  }
  __syncthreads();
  // share of rr = _W(merged): rrp[n][j] = sum_e out[n][e] Wo[j][e*H + h], Wo_h from the head-major copy (float4 along e)
  {
    const float* woh = a.wot + (size_t)h * d.dw * d.dw;
    for (int jt = wave; jt * 16 < d.dw; jt += 8) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      for (int k64 = 0; k64 < d.dw; k64 += 64) {        // dw % 64 == 0
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = k64 + 16 * u + 4 * lq;
          const float4 b4 = *reinterpret_cast<const float4*>(woh + (size_t)(16 * jt + lr) * d.dw + kk);
          lcptr xp = s_xq + lr * Lx + kk;
          acc = mfma4(xp[0], b4.x, acc); acc = mfma4(xp[1], b4.y, acc);
          acc = mfma4(xp[2], b4.z, acc); acc = mfma4(xp[3], b4.w, acc);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 4 * lq + r;
        if (n < d.Nq) a.rrp[((size_t)(t * H + h) * d.Nq + n) * d.dw + 16 * jt + lr] = acc[r];
      }
    }
  }
  MLHOT_TSTAMP(41);
}

__host__ inline size_t phaseB_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (5 * ldpad(d.dw) + 2 * ldpad(d.m)) + 8 * 16 * 17 + 64 + 16);
}

// ==================================================================================================
// phase C forward, one workgroup per task: rr = _W(merged); z = r_to_z(rr) -> dec_in[:, dw:];
// d1, d2 = decoder hidden; mu = act(decoder out).
// ==================================================================================================
struct PhaseCArgs {
  TailDims d; TailParams p;
  const float* rrp;                        // [T*H][Nq][dw] per-head shares of _W(merged) from phase B
  float *rr, *dec_in, *d1, *d2, *mu;
};

__global__ __launch_bounds__(512) void phaseC_fwd_kernel(const PhaseCArgs a) {
  MLHOT_TSTAMP(64);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = d.dw + d.dz;
  const int Lr = ldpad(d.dw), Ld = ldpad(ldd), Lh = ldpad(d.dec_h);
  lptr s_rr = L0;                 // [16][Lr]
  lptr s_dec = s_rr + 16 * Lr;
  lptr s_d1 = s_dec + 16 * Ld;
  lptr s_d2 = s_d1 + 16 * Lh;
  lptr s_red = s_d2 + 16 * Lh;    // [8 waves][256] K-split partials of wg_linear
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 16 * (Lr + Ld + 2 * Lh), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(65);
  gptr g_dec = G(a.dec_in) + (size_t)t * d.Nq * ldd;
  {
    const LoadJob xj[1] = {load_job(s_dec, Ld, a.dec_in + (size_t)t * d.Nq * ldd, ldd, d.Nq, d.dw)};     // x_qry
    LoadBatch<1, 2, 1> xb;
    xb.fetch(xj, tid, 8);
    // rr = _W(merged) = bias + the 8 heads' shares (fixed order), all 8 loads of an element in flight together
    for (int i = tid; i < d.Nq * d.dw; i += 512) {
      const int n = i / d.dw, j = i - n * d.dw;
      float v[H];
#pragma unroll
      for (int h = 0; h < H; ++h) v[h] = a.rrp[((size_t)(t * H + h) * d.Nq + n) * d.dw + j];
      float sum = a.p.wo_b[j];
#pragma unroll
      for (int h = 0; h < H; ++h) sum += v[h];
      s_rr[n * Lr + j] = sum;
      a.rr[((size_t)t * d.Nq + n) * d.dw + j] = sum;
    }
    xb.stash(xj, tid, 8);
  }
  __syncthreads();
  MLHOT_TSTAMP(66);
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
  return sizeof(float) * (16 * (ldpad(d.dw) + ldpad(d.dw + d.dz) + 2 * ldpad(d.dec_h)) + 8 * 256 + ptab_floats<TailParams>());
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
  const int lane = tid & 63, nw = nthreads >> 6;
  for (int r = tid >> 6; r < 16; r += nw)
    for (int c = lane; c < width; c += 64) g[r * ldg_ + c] *= act_grad_from_out(act, y[r * ldy + c]);
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
//   out: d_dec_in[:, :dw] (decoder's share of d x_qry), d_rr (gradient of _W's output; _W's own backward runs per head in
//        phase B), weight-gradient slab entries
// ==================================================================================================
struct PhaseCBwdArgs {
  TailDims d; TailParams p; TailSlab sl;
  const float *dmu, *mu, *d2, *d1, *dec_in, *rr;
  float *d_dec_in, *d_rr, *slab;
  // tail_spec.h only: the loss whose gradient this launch takes itself (dmu: nullptr = nothing else, or an addend); kind < 0: none
  LossDesc loss;
};

__global__ __launch_bounds__(512) void phaseC_bwd_kernel(const PhaseCBwdArgs a) {
  MLHOT_TSTAMP(96);
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldd = d.dw + d.dz;
  const int Ly = ldpad(d.y_dim), Lh = ldpad(d.dec_h), Ld = ldpad(ldd), Lr = ldpad(d.dw);
  lptr s_g = L0;                  // [16][Ly]   dmu * act'(mu)
  lptr s_d2 = s_g + 16 * Ly;       // saved activations
  lptr s_d1 = s_d2 + 16 * Lh;
  lptr s_dec = s_d1 + 16 * Lh;
  lptr s_rr = s_dec + 16 * Ld;
  lptr s_dd2 = s_rr + 16 * Lr;     // gradients
  lptr s_dd1 = s_dd2 + 16 * Lh;
  lptr s_ddec = s_dd1 + 16 * Lh;
  lptr s_drr = s_ddec + 16 * Ld;
  lptr s_red = s_drr + 16 * Lr;    // [8 waves][256] partial tiles of wg_dgrad
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 16 * (Ly + 4 * Lh + 2 * Ld + 2 * Lr), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(97);
  const size_t rq = (size_t)t * d.Nq;
  for (int i = tid; i < d.Nq * d.y_dim; i += 512) {
    const int r = i / d.y_dim, c = i % d.y_dim;
    s_g[r * Ly + c] = a.dmu[(rq + r) * d.y_dim + c] * act_grad_from_out(d.out_act, a.mu[(rq + r) * d.y_dim + c]);
  }
  {
    const LoadJob xj[4] = {
        load_job(s_d2, Lh, a.d2 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h),
        load_job(s_d1, Lh, a.d1 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h),
        load_job(s_dec, Ld, a.dec_in + rq * ldd, ldd, d.Nq, ldd),
        load_job(s_rr, Lr, a.rr + rq * d.dw, d.dw, d.Nq, d.dw)};
    LoadBatch<4, 2, 2> xb;
    xb.fetch(xj, tid, 8);
    xb.stash(xj, tid, 8);
  }
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
  wg_dgrad<8>(s_ddec + d.dw, Ld, d.dz, WB1N(r2z_w, d.dz), d.dw, s_drr, Lr, G(a.d_rr + rq * d.dw), d.dw, d.Nq, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(104);
  // _W: only its bias gradient here (column sums of d rr); weight and input gradient run per head in phase B
  for (int j = tid; j < d.dw; j += 512) {
    float sum = 0.f;
#pragma unroll
    for (int row = 0; row < 16; ++row) sum += s_drr[row * Lr + j];
    sl[a.sl.wo_b + j] = sum;
  }
  MLHOT_TSTAMP(105);
}
__host__ inline size_t phaseC_bwd_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (ldpad(d.y_dim) + 4 * ldpad(d.dec_h) + 2 * ldpad(d.dw + d.dz) + 2 * ldpad(d.dw)) + 8 * 256 +
                          ptab_floats<TailParams>());
}

// ==================================================================================================
// phase B backward, one workgroup per (task, head): FAVOR+ backward (S-form, see favor.h).
// followed, still per head, by the backward of this head's three projections: dq / dk / dv never leave LDS.
//   out: slab entries of W_q,h / W_k,h / W_v,h (weight + bias gradients, dk WITHOUT the batch-global arg-max
//        correction, which phase A applies as a rank-1 fix-up), the head's shares pxq / pxc / prs
//        [T*H][N][dw] of d x_qry / d x_ctx / d rs, and part_k[t*H+h] = sum of rsum_k
// ==================================================================================================
struct PhaseBBwdArgs {
  TailDims d; TailParams p; TailSlab sl;
  const float *qh, *kh, *vh, *pc, *qf, *kf, *S, *D, *merged, *d_rr; const int* arg_q;
  const float *dec_in, *cat_in, *rs;       // x_qry = dec_in[:, :dw], x_ctx = cat_in[:, :dw]
  const float* wot;                        // [H][dw][dw] head-major _W weight (phase A of the forward)
  float *pxq, *pxc, *prs, *part_k, *slab;
};

__global__ __launch_bounds__(512) void phaseB_bwd_kernel(const PhaseBBwdArgs a) {
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
  lptr s_xq = s_st + 64;             // [16][Lx] projection inputs ...
  lptr s_xc = s_xq + 16 * Lx;
  lptr s_rs = s_xc + 16 * Lx;
  lptr s_dq = s_rs + 16 * Lx;        // ... and head-space gradients
  lptr s_dk = s_dq + 16 * Lx;
  lptr s_dv = s_dk + 16 * Lx;
  lptr s_o = s_dv + 16 * Lx;         // this head's attention output O (from merged) and d rr of the task (from phase C)
  lptr s_drr = s_o + 16 * Lx;
  const int total = 16 * (12 * Lx + 4 * Lf) + 2 * 16 * 17 + 64;
  lds_zero(L0, total, tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(129);
  {
    const LoadJob xj[8] = {
        load_job(s_q, Lx, a.qh + (size_t)t * d.Nq * HD + h * d.dw, HD, d.Nq, d.dw),
        load_job(s_k, Lx, a.kh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw),
        load_job(s_v, Lx, a.vh + (size_t)t * d.Nc * HD + h * d.dw, HD, d.Nc, d.dw),
        load_job(s_xq, Lx, a.dec_in + (size_t)t * d.Nq * (d.dw + d.dz), d.dw + d.dz, d.Nq, d.dw),
        load_job(s_xc, Lx, a.cat_in + (size_t)t * d.Nc * (d.dw + d.dw / 4), d.dw + d.dw / 4, d.Nc, d.dw),
        load_job(s_rs, Lx, a.rs + (size_t)t * d.Nc * d.dw, d.dw, d.Nc, d.dw),
        // O[n][e] = merged[(t,n)][e*H + h]
        load_job(s_o, Lx, a.merged + (size_t)t * d.Nq * HD + h, HD, d.Nq, d.dw, H),
        load_job(s_drr, Lx, a.d_rr + (size_t)t * d.Nq * d.dw, d.dw, d.Nq, d.dw)};
    const LoadJob ej[2] = {
        load_job(s_qf, Lf, a.qf + ((size_t)t * d.Nq * H + h) * d.m, H * d.m, d.Nq, d.m),
        load_job(s_kf, Lf, a.kf + ((size_t)t * d.Nc * H + h) * d.m, H * d.m, d.Nc, d.m)};
    LoadBatch<8, 2, 1> xb;
    LoadBatch<2, 2, 5> eb;
    xb.fetch(xj, tid, 8);
    eb.fetch(ej, tid, 8);
    xb.stash(xj, tid, 8);
    eb.stash(ej, tid, 8);
  }
  if (tid < d.Nq) s_st[16 + tid] = a.D[((size_t)t * H + h) * d.Nq + tid];
  __syncthreads();
  // _W's input gradient for this head: dO[n][e] = sum_j d rr[n][j] Wo[j][e*H + h] (head-major copy: coalesced along e);
  // rows >= Nq of d rr are zero, so those rows of dO are zero too
  {
    const float* woh = a.wot + (size_t)h * d.dw * d.dw;
    for (int et = wave; et * 16 < d.dw; et += 8) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      for (int j0 = 0; j0 < d.dw; j0 += 64) {
        float bw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + 16 * (u >> 2) + 4 * lq + (u & 3);
          bw[u] = woh[(size_t)(j < d.dw ? j : d.dw - 1) * d.dw + 16 * et + lr];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (j0 + 16 * ks < d.dw) {
            lcptr ap = s_drr + lr * Lx + j0 + 16 * ks + 4 * lq;
            acc = mfma4(ap[0], bw[4 * ks], acc); acc = mfma4(ap[1], bw[4 * ks + 1], acc);
            acc = mfma4(ap[2], bw[4 * ks + 2], acc); acc = mfma4(ap[3], bw[4 * ks + 3], acc);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) s_do[(4 * lq + r) * Lx + 16 * et + lr] = acc[r];
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(130);
  if (tid < 256) {
    const int n = tid >> 4, part = tid & 15;
    float s = 0.f;
    if (n < d.Nq)
      for (int e = part; e < d.dw; e += 16) s += s_do[n * Lx + e] * (s_o[n * Lx + e] - s_v[e]);     // relative to c = v[0], see below
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
  // dS[n][n'] = (dO[n] . (v[n'] - c) - wv[n]) / D[n], wv[n] = dO[n] . (O[n] - c), c = v[0]   (wave 0), valid entries only.  O[n] is
  // a convex combination of the value rows: without the common centre the two inner products share their leading digits whenever
  // the value rows have a large common component, and the difference is their fp32 rounding (favor2.h, B1)
  if (wave == 0) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int e0 = 0; e0 < d.dw; e0 += 4) acc = mfma4(s_do[lr * Lx + e0 + lq], s_v[lr * Lx + e0 + lq] - s_v[e0 + lq], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, np = lr;
      s_dS[n * 17 + np] = (n < d.Nq && np < d.Nc) ? (acc[r] - s_st[n]) / s_st[16 + n] : 0.f;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(132);
  // dV[n'][e] = sum_n (S/D)[n][n'] dO[n][e]  -> dvh
  for (int et = wave; et * 16 < d.dw; et += 8) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int n = 4 * s4 + lq;
      acc = mfma4(s_S[n * 17 + lr], s_do[n * Lx + et * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int np = 4 * lq + r;
      s_dv[np * Lx + et * 16 + lr] = np < d.Nc ? acc[r] : 0.f;
    }
  }
  // G = dF (.) E with dQ' = dS (Ek + re), dK' = dS^T (Eq + re): feature tiles over the waves
  const float ratio = 1.0f / sqrtf((float)d.m), re = ratio * 1e-4f;
  const int ntile = (d.m + 15) / 16;
  for (int it = wave; it < 2 * ntile; it += 8) {
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
  // row sums of G (32 rows x 16 threads), then d(dd): queries subtract the row sum at the arg-max
  {
    const int row = tid >> 4, part = tid & 15;
    lcptr g = row < 16 ? s_gq + row * Lf : s_gk + (row - 16) * Lf;
    float s = 0.f;
    for (int j = part; j < d.m; j += 16) s += g[j];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
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
  // dx[row][e] = sum_j d(dd)[row][j] pc[j][e] - rsum[row] c^2 x[row][e].  A wave owns an e tile for BOTH the
  // query and the key rows (they share the pc operand), 16 k-steps of pc loads in flight per trip.
  {
    const float c2 = 1.0f / sqrtf((float)d.dw);
    const int net = d.dw / 16;
    for (int it = wave; it < 2 * net; it += 8) {          // items: e tile x (query rows | key rows)
      const bool isk = it >= net;
      const int et = isk ? it - net : it;
      lcptr g = isk ? s_gk : s_gq;
      f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* pcol = a.pc + et * 16 + lr;
      for (int j0 = 0; j0 < d.m; j0 += 64) {
        float bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + 4 * u + lq;
          const float v = pcol[(size_t)(j < d.m ? j : d.m - 1) * d.dw];
          bv[u] = j < d.m ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          if (j0 + 4 * u < d.m) acc0 = mfma4(g[lr * Lf + j0 + 4 * u + lq], bv[u], acc0);
          if (j0 + 4 * u + 4 < d.m) acc1 = mfma4(g[lr * Lf + j0 + 4 * u + 4 + lq], bv[u + 1], acc1);
        }
      }
      lcptr xs = isk ? s_k : s_q;
      lptr dst = isk ? s_dk : s_dq;
      const int nrows = isk ? d.Nc : d.Nq, so = isk ? 48 : 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r, e = et * 16 + lr;
        dst[row * Lx + e] = row < nrows ? acc0[r] + acc1[r] - s_st[so + row] * c2 * xs[row * Lx + e] : 0.f;
      }
    }
  }
  __syncthreads();
  // ---- this head's W_q / W_k / W_v backward ------------------------------------------------------
  {
    const float *wq = a.p.wq_w[0], *wk = a.p.wk_w[0], *wv = a.p.wv_w[0];
#pragma unroll
    for (int i = 1; i < H; ++i)
      if (h == i) { wq = a.p.wq_w[i]; wk = a.p.wk_w[i]; wv = a.p.wv_w[i]; }
    float* sl = a.slab + (size_t)t * a.sl.total;
    const int nt = d.dw / 16, ww = d.dw * d.dw;
    // weight gradients dW[n][i] = sum_row dY[row][n] X[row][i]: 3 nt^2 tiles of 4 MFMAs over the waves
    for (int it = wave; it < 3 * nt * nt; it += 8) {
      const int pj = it / (nt * nt), rem = it - pj * nt * nt, j0 = (rem / nt) * 16, i0 = (rem % nt) * 16;
      lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
      lcptr x = pj == 0 ? s_xq : pj == 1 ? s_xc : s_rs;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = mfma4(dy[(4 * s4 + lq) * Lx + j0 + lr], x[(4 * s4 + lq) * Lx + i0 + lr], acc);
      float* dst = sl + (pj == 0 ? a.sl.wq_w : pj == 1 ? a.sl.wk_w : a.sl.wv_w) + h * ww;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(j0 + 4 * lq + r) * d.dw + i0 + lr] = acc[r];
    }
    // _W's weight gradient for this head's columns: dWo[j][e*H + h] = sum_n d rr[n][j] O[n][e]
    for (int it = wave; it < nt * nt; it += 8) {
      const int j0 = (it / nt) * 16, e0 = (it % nt) * 16;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = mfma4(s_drr[(4 * s4 + lq) * Lx + j0 + lr], s_o[(4 * s4 + lq) * Lx + e0 + lr], acc);
      float* dst = sl + a.sl.wo_w;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)(j0 + 4 * lq + r) * HD + (e0 + lr) * H + h] = acc[r];
    }
    // bias gradients: column sums
    for (int i = tid; i < 3 * d.dw; i += 512) {
      const int pj = i / d.dw, n = i - pj * d.dw;
      lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
      float sum = 0.f;
#pragma unroll
      for (int row = 0; row < 16; ++row) sum += dy[row * Lx + n];
      sl[(pj == 0 ? a.sl.wq_b : pj == 1 ? a.sl.wk_b : a.sl.wv_b) + h * d.dw + n] = sum;
    }
    // input-gradient shares P[16][dw] = dY[16][dw] W_h[dw][dw]: 3 nt tiles, 16 weight loads in flight per trip
    for (int it = wave; it < 3 * nt; it += 8) {
      const int pj = it / nt, i0 = (it - pj * nt) * 16;
      lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
      const float* wsel = (pj == 0 ? wq : pj == 1 ? wk : wv) + i0 + lr;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      for (int j0 = 0; j0 < d.dw; j0 += 64) {
        float bw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + 4 * (u >> 2) * 4 + 4 * lq + (u & 3);        // k-step u >> 2, element u & 3
          bw[u] = wsel[(size_t)(j < d.dw ? j : d.dw - 1) * d.dw];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (j0 + 16 * ks < d.dw) {
            lcptr ap = dy + lr * Lx + j0 + 16 * ks + 4 * lq;
            acc = mfma4(ap[0], bw[4 * ks], acc); acc = mfma4(ap[1], bw[4 * ks + 1], acc);
            acc = mfma4(ap[2], bw[4 * ks + 2], acc); acc = mfma4(ap[3], bw[4 * ks + 3], acc);
          }
        }
      }
      float* dst = pj == 0 ? a.pxq : pj == 1 ? a.pxc : a.prs;
      const int nrows = pj == 0 ? d.Nq : d.Nc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        if (row < nrows) dst[((size_t)(t * H + h) * nrows + row) * d.dw + i0 + lr] = acc[r];
      }
    }
  }
  MLHOT_TSTAMP(136);
}
__host__ inline size_t phaseB_bwd_lds_bytes(const TailDims& d) {
  return sizeof(float) * (16 * (12 * ldpad(d.dw) + 4 * ldpad(d.m)) + 2 * 16 * 17 + 64);
}

// ==================================================================================================
// phase A backward, one workgroup per task: sums the 8 heads' input-gradient shares from phase B,
// applies the batch-global key arg-max correction, EncoderFC backward, transform_y weight gradient.
//   out: d_dec_in[:, :dw] += d x_qry (attention share), d_cat_in, slab entries
// ==================================================================================================
struct PhaseABwdArgs {
  TailDims d; TailParams p; TailSlab sl;
  const float *ctx_y, *cat_in, *h0, *h1, *pxq, *pxc, *prs, *pc, *part_k; const int* gpos;
  float *d_dec_in, *d_cat_in, *slab;
};

__global__ __launch_bounds__(512) void phaseA_bwd_kernel(const PhaseABwdArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lw = ldpad(d.dw), Ly = ldpad(d.label_dim);
  lptr s_cat = L0;                  // saved activations
  lptr s_h0 = s_cat + 16 * Lcat;
  lptr s_h1 = s_h0 + 16 * Lh0;
  lptr s_y = s_h1 + 16 * Lh1;
  lptr s_drs = s_y + 16 * Ly;       // gradients
  lptr s_dxc = s_drs + 16 * Lw;
  lptr s_dh1 = s_dxc + 16 * Lw;
  lptr s_dh0 = s_dh1 + 16 * Lh1;
  lptr s_dcat = s_dh0 + 16 * Lh0;
  lptr s_red = s_dcat + 16 * Lcat;   // [8 waves][256] partial tiles of wg_dgrad; first the correction vector
  using PRM = TailParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  MLHOT_TSTAMP(160);
  lds_zero(L0, 16 * (2 * Lcat + 2 * Lh0 + 2 * Lh1 + 2 * Lw + Ly), tid, 512);
  __syncthreads();
  MLHOT_TSTAMP(161);
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  {
    const LoadJob xj[4] = {
        load_job(s_cat, Lcat, a.cat_in + rc * ldc, ldc, d.Nc, ldc),
        load_job(s_h0, Lh0, a.h0 + rc * d.h0, d.h0, d.Nc, d.h0),
        load_job(s_h1, Lh1, a.h1 + rc * d.h1, d.h1, d.Nc, d.h1),
        load_job(s_y, Ly, a.ctx_y + rc * d.label_dim, d.label_dim, d.Nc, d.label_dim)};
    LoadBatch<4, 2, 2> xb;
    xb.fetch(xj, tid, 8);
    xb.stash(xj, tid, 8);
  }
  // sum the heads' shares (fixed order): d rs, the K-projection share of d x_ctx, the attention share of d x_qry
  for (int i = tid; i < d.Nc * d.dw; i += 512) {
    const int row = i / d.dw, e = i - row * d.dw;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      s1 += a.prs[((size_t)(t * H + h) * d.Nc + row) * d.dw + e];
      s2 += a.pxc[((size_t)(t * H + h) * d.Nc + row) * d.dw + e];
    }
    s_drs[row * Lw + e] = s1;
    s_dxc[row * Lw + e] = s2;
  }
  for (int i = tid; i < d.Nq * d.dw; i += 512) {
    const int row = i / d.dw, e = i - row * d.dw;
    float s1 = 0.f;
#pragma unroll
    for (int h = 0; h < H; ++h) s1 += a.pxq[((size_t)(t * H + h) * d.Nq + row) * d.dw + e];
    a.d_dec_in[(rq + row) * ldd + e] += s1;
  }
  gptr sl = G(a.slab) + (size_t)t * a.sl.total;
  // The batch-global key arg-max: that ONE element's d(dd) carries minus the sum of G over every key row
  // of the batch (fast_attention.py:97), i.e. dk[row n, head hh] += delta, delta = -total * pc[col].  Phase B
  // ran W_k,hh's backward without it; it enters linearly, so it is added here as a rank-1 fix-up:
  //   dW_k,hh += delta (x) x_ctx[n],  db_k,hh += delta,  d x_ctx[n] += delta . W_k,hh
  const int grow = a.gpos[0], gcol = a.gpos[1];
  const bool fix = grow / (d.Nc * H) == t;
  const int fn = (grow / H) % d.Nc, fh = grow % H;
  {   // total = sum of part_k over every (task, head): lanes, then waves, in a fixed order
    float pv = 0.f;
    for (int i = tid; i < d.T * H; i += 512) pv += a.part_k[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pv += __shfl_xor(pv, off, 64);
    if (lane == 0) s_red[1024 + wave] = pv;
  }
  __syncthreads();
  if (fix && tid < d.dw) {
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) total += s_red[1024 + w];
    s_red[tid] = -total * a.pc[(size_t)gcol * d.dw + tid];
  }
  __syncthreads();
  MLHOT_TSTAMP(162);
  if (fix) {
    const float* wk = a.p.wk_w[0];
#pragma unroll
    for (int i = 1; i < H; ++i)
      if (fh == i) wk = a.p.wk_w[i];
    gptr gw = sl + a.sl.wk_w + fh * d.dw * d.dw;
    for (int i = tid; i < d.dw * d.dw; i += 512) {
      const int e = i / d.dw, c = i - e * d.dw;
      gw[i] += s_red[e] * s_cat[fn * Lcat + c];
    }
    if (tid < d.dw) {
      sl[a.sl.wk_b + fh * d.dw + tid] += s_red[tid];
      float acc = 0.f;
      for (int e = 0; e < d.dw; ++e) acc += s_red[e] * wk[(size_t)e * d.dw + tid];
      s_dxc[fn * Lw + tid] += acc;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(163);
  // EncoderFC, last layer first
  wg_wgrad<8>(s_drs, Lw, d.dw, s_h1, Lh1, d.h1, sl + a.sl.er_w[2], sl + a.sl.er_b[2], wave, lane, tid);
  wg_dgrad<8>(s_drs, Lw, d.dw, WB1N(er_w[2], d.dw), d.h1, s_dh1, Lh1, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(164);
  lds_actgrad(s_dh1, Lh1, s_h1, Lh1, d.h1, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh1, Lh1, d.h1, s_h0, Lh0, d.h0, sl + a.sl.er_w[1], sl + a.sl.er_b[1], wave, lane, tid);
  wg_dgrad<8>(s_dh1, Lh1, d.h1, WB1N(er_w[1], d.h1), d.h0, s_dh0, Lh0, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(165);
  lds_actgrad(s_dh0, Lh0, s_h0, Lh0, d.h0, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh0, Lh0, d.h0, s_cat, Lcat, ldc, sl + a.sl.er_w[0], sl + a.sl.er_b[0], wave, lane, tid);
  wg_dgrad<8>(s_dh0, Lh0, d.h0, WB1N(er_w[0], d.h0), ldc, s_dcat, Lcat, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(166);
  // d_cat_in = EncoderFC input gradient (+ K-projection share on the x_ctx columns)
  for (int i = tid; i < d.Nc * ldc; i += 512) {
    const int r = i / ldc, c = i % ldc;
    a.d_cat_in[(rc + r) * ldc + c] = s_dcat[r * Lcat + c] + (c < d.dw ? s_dxc[r * Lw + c] : 0.f);
  }
  // transform_y: dW = d_cat[:, dw:]^T ctx_y, db
  wg_wgrad<8>(s_dcat + d.dw, Lcat, d.dw / 4, s_y, Ly, d.label_dim, sl + a.sl.ty_w, sl + a.sl.ty_b, wave, lane, tid);
  MLHOT_TSTAMP(167);
}
__host__ inline size_t phaseA_bwd_lds_bytes(const TailDims& d) {
  const int ldc = d.dw + d.dw / 4;
  return sizeof(float) * (16 * (2 * ldpad(ldc) + 2 * ldpad(d.h0) + 2 * ldpad(d.h1) + 2 * ldpad(d.dw) + ldpad(d.label_dim)) + 8 * 256 +
                          ptab_floats<TailParams>());
}

// ---- sum the per-task slabs into the parameter gradients (fixed task order) -------------------------
constexpr int MAX_SEG = 72;
struct SlabReduce {
  float* dst[MAX_SEG]; int off[MAX_SEG]; int len[MAX_SEG];
  int nseg, T, total; const float* slab;
};
// One flat pass over the slab: a thread sums element `e` of all T task slabs (T coalesced loads in flight), finds the
// parameter segment it belongs to and stores into that parameter's gradient.  grid = ceil(total / 256).
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
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.total) return;
  float s0 = 0.f, s1 = 0.f;
  int t = 0;
  for (; t + 1 < a.T; t += 2) { s0 += a.slab[(size_t)t * a.total + e]; s1 += a.slab[(size_t)(t + 1) * a.total + e]; }
  if (t < a.T) s0 += a.slab[(size_t)t * a.total + e];
  int seg = -1;
  for (int i = 0; i < a.nseg; ++i)
    if (e >= s_off[i] && e < s_off[i] + s_len[i]) seg = i;
  if (seg >= 0) s_dst[seg][e - s_off[seg]] = s0 + s1;      // else: alignment padding between two segments
}

}  // namespace tf
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
