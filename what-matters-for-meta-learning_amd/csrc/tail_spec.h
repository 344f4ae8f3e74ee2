// The fused ANP tail (tail_fused.h) specialised for the dimensions every shipped vanilla-ANP config uses
// (cfg/train/ANP_*1D.yaml: dim_w = dim_r = dim_z = 64, n_hidden_units_r = [100, 100], decoder hidden 100, hence
// m = 266 random features), label / output widths 1..4 at run time.  Same six phases, same argument structs, same saved
// buffers and slab layout as tail_fused.h - a step may mix the two flavours phase by phase (mlhot_set_option("tail_spec", 0)
// keeps the generic kernels as the A/B reference) - but organised around what the stage timestamps of the generic kernels
// showed (scripts/tail_ts.py): a layer of a per-task chain cost ~3.5 us of which ~2 us was ONE exposed L2 round trip for its
// weights and ~0.5 us making run-time shapes wave-uniform.  Here
//   * every weight fragment a wave will ever need in the kernel is requested at kernel entry, into registers, together with
//     the activations (the weights do not depend on them): 60-120 VGPRs per lane, ONE round trip per kernel instead of one
//     per layer;
//   * every shape is a compile-time constant: tiles per wave, K-split and fold pattern are fixed, the k-loops unroll into
//     ds_read_b128 + 4 MFMAs per 16 k with two accumulator chains (a dependent v_mfma_f32_16x16x4_f32 chain issues every 40
//     cycles, two chains every 32);
//   * ReLU's backward rides in the data-gradient epilogue and the row sums ride in the tiles that have the data anyway, so a
//     layer is one barrier (two when its K range is split over idle waves).
#pragma once
#include "tail_fused.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace ts {
using namespace tf;

constexpr int DW = 64, DZ = 64, H0 = 100, H1 = 100, DH = 100, M = 266;
constexpr int LDC = DW + DW / 4, LDD = DW + DZ, HD = H * DW;
constexpr int NWV = 8;                                    // waves per workgroup (512 threads)
__host__ __device__ constexpr int pad16(int w) { return (w + 15) / 16 * 16; }
__host__ __device__ constexpr int lds_ld(int w) { return pad16(w) + 4; }

inline bool applies(const TailDims& d) {
  return d.dw == DW && d.dz == DZ && d.h0 == H0 && d.h1 == H1 && d.dec_h == DH && d.m == M && d.label_dim >= 1 && d.label_dim <= 4 &&
         d.y_dim >= 1 && d.y_dim <= 4 && d.Nc >= 1 && d.Nc <= 16 && d.Nq >= 1 && d.Nq <= 16;
}

// ---- Y[16 x N] = act(X[16 x K] W^T + b): weight fragments in registers ------------------------------------------------
// Work items = (16-column tile of N) x (chunk of the k blocks); item w belongs to wave w.  With fewer tiles than waves the k
// range is split (NT * NCH <= 8) and folded through LDS in a fixed order.  X lives in LDS with zero padding up to
// pad16(K) columns, so out-of-range weight addresses are only clamped to something finite.
// K, N compile time except KR / NR: the run-time extent (<= K / <= N) for the two layers whose width is the label / output size.
template <int K, int N>
struct Lin {
  static constexpr int KB = (K + 15) / 16, NT = (N + 15) / 16;
  static constexpr int NCH0 = NT >= 5 ? 1 : NT >= 3 ? 2 : NT == 2 ? 4 : 8;
  static constexpr int NCH = NCH0 < KB ? NCH0 : KB;
  static constexpr int PER = (KB + NCH - 1) / NCH;
  f32x4_t b[PER];
  float bias;

  __device__ __forceinline__ static bool active(int wave) { return wave < NT * NCH; }

  // kr / nr: run-time K / N of this call (= K / N for the fixed layers)
  __device__ __forceinline__ void load(const float* __restrict__ W, const float* __restrict__ B, int wave, int lane, int kr = K, int nr = N) {
    bias = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) b[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!active(wave)) return;
    const int lr = lane & 15, lq = lane >> 4;
    const int tile = wave % NT, chunk = wave / NT;
    const int n = tile * 16 + lr, nc = n < nr ? n : nr - 1;
    bias = B[nc];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int kb = chunk * PER + i;
      if constexpr (K % 4 == 0) {
        int k0 = kb * 16 + 4 * lq;
        k0 = k0 <= K - 4 ? k0 : K - 4;
        b[i] = *reinterpret_cast<const f32x4_t*>(W + nc * K + k0);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kb * 16 + 4 * lq + e;
          b[i][e] = W[nc * kr + (k < kr ? k : kr - 1)];
        }
      }
    }
  }

  // this wave's partial tile: rows 4 lq + r, column tile * 16 + lr
  __device__ __forceinline__ f32x4_t mma(lcptr x, int ldx, int wave, int lane) const {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (active(wave)) {
      const int lr = lane & 15, lq = lane >> 4;
      const int chunk = wave / NT;
      lcptr xr = x + lr * ldx + 4 * lq;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int kb = chunk * PER + i;
        if (kb < KB) {
          const f32x4_t a = *reinterpret_cast<lc4ptr>(xr + kb * 16);
          if (i & 1) {
            a1 = mfma4(a[0], b[i][0], a1); a1 = mfma4(a[1], b[i][1], a1); a1 = mfma4(a[2], b[i][2], a1); a1 = mfma4(a[3], b[i][3], a1);
          } else {
            a0 = mfma4(a[0], b[i][0], a0); a0 = mfma4(a[1], b[i][1], a0); a0 = mfma4(a[2], b[i][2], a0); a0 = mfma4(a[3], b[i][3], a0);
          }
        }
      }
    }
    return a0 + a1;
  }

  // fold the k chunks (fixed order), bias, activation, stores.  Call from ALL waves (one barrier inside when NCH > 1).
  // ys: LDS tile (row stride ldy) or nullptr; yg: global rows (< nrows, row stride ldg) or nullptr.
  __device__ __forceinline__ void finish(f32x4_t acc, int act, lptr red, lptr ys, int ldy, float* __restrict__ yg, int ldg, int nrows,
                                         int wave, int lane, int nr = N) const {
    const int lr = lane & 15, lq = lane >> 4;
    const int tile = wave % NT, chunk = wave / NT;
    if constexpr (NCH > 1) {
      if (active(wave) && chunk > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (active(wave) && chunk == 0) {
#pragma unroll
        for (int c = 1; c < NCH; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += red[((wave + c * NT) * 4 + r) * 64 + lane];
      }
    }
    if (active(wave) && chunk == 0) {
      const int n = tile * 16 + lr;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] + bias;
      if (act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (act == ACT_TANH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
      }
      if (ys != nullptr) {                            // the tile's padding columns (n >= nr) become exact zeros: the next layer's k padding
#pragma unroll
        for (int r = 0; r < 4; ++r) ys[(4 * lq + r) * ldy + n] = n < nr ? v[r] : 0.f;
      }
      if (yg != nullptr && n < nr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * lq + r < nrows) yg[(4 * lq + r) * ldg + n] = v[r];
      }
    }
  }
};

// one 16 x 64 activation tile (rows < nrows of a [rows][ldg] global matrix, 64 columns from column c0) as one float4 per
// thread of the first 256: issue (registers) / stash (LDS, row stride ld)
struct Tile64 {
  f32x4_t v;
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int ldg, int nrows, int tid) {
    const int row = (tid >> 4) & 15, c4 = tid & 15;
    v = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (tid < 256 && row < nrows) v = *reinterpret_cast<const f32x4_t*>(src + (size_t)row * ldg + 4 * c4);
  }
  __device__ __forceinline__ void stash(lptr dst, int ld, int tid) const {
    const int row = (tid >> 4) & 15, c4 = tid & 15;
    if (tid < 256) *reinterpret_cast<MLHOT_LDS f32x4_t*>(dst + row * ld + 4 * c4) = v;
  }
};

// The same tile when the encoder Linear left its split-K fold to the consumer (PhaseAArgs::xslab): feat[row][:] = bias + the XK
// partial results, summed in the order el::enc_linear_fold_kernel uses (bit-identical features whoever folds).  issue() requests
// all XK + 1 float4 of a thread; finish() - called where the tile is needed, behind the kernel's other up-front requests - adds
// them up (written as one function, hipcc put the sums, and their wait, in front of the weight fragments' loads: a second round trip).
// `r0`: the tile's first row in the encoder's [context | target] row order.
template <int NK>
struct FoldTile {
  f32x4_t part[NK], bias;
  bool on;
  __device__ __forceinline__ void issue(const float* __restrict__ slab, const float* __restrict__ b, int xn, int r0, int nrows, int tid) {
    const int row = (tid >> 4) & 15, c4 = tid & 15;
    on = tid < 256 && row < nrows;
    const float* src = slab + (size_t)(r0 + (on ? row : 0)) * DW + 4 * c4;
#pragma unroll
    for (int z = 0; z < NK; ++z) part[z] = *reinterpret_cast<const f32x4_t*>(src + (size_t)z * xn * DW);
    bias = *reinterpret_cast<const f32x4_t*>(b + 4 * c4);
  }
  __device__ __forceinline__ f32x4_t finish() const {
    f32x4_t t = bias;
#pragma unroll
    for (int z = 0; z < NK; ++z) t += part[z];
    return on ? t : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
};
constexpr int XK = 16;                                    // el::F_KS (np_vanilla.h checks it)

__device__ __forceinline__ void lds_zero4(lptr p, int nfloats, int tid) {      // nfloats % 4 == 0, p 16-byte aligned
  const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < nfloats / 4; i += NWV * 64) reinterpret_cast<MLHOT_LDS f32x4_t*>(p)[i] = z;
}

// ==================================================================================================
// phase A forward (see tail_fused.h): blocks [0, T) walk transform_y + EncoderFC of one task, blocks [T, T + T*H) make the
// key projection of one (task, head) and its share of the batch-global key stabiliser.
// ==================================================================================================
constexpr int A_LCAT = lds_ld(LDC), A_LH = lds_ld(H0), A_LY = 20, A_LX = lds_ld(DW);
constexpr int A_CHAIN_FLOATS = 16 * (A_LCAT + 2 * A_LH + A_LY) + NWV * 256;
constexpr int A_KEY_FLOATS = 32 * A_LX + 16;
__host__ inline size_t phaseA_lds_bytes() { return sizeof(float) * (A_CHAIN_FLOATS > A_KEY_FLOATS ? A_CHAIN_FLOATS : A_KEY_FLOATS); }

__device__ __forceinline__ void keyhead_block(const PhaseAArgs& a, lptr L0, int th, int tid) {
  const TailDims& d = a.d;
  const int t = th / H, h = th - t * H, lane = tid & 63, wave = uni(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  lptr s_x = L0;                       // [16][A_LX] x_ctx
  lptr s_k = s_x + 16 * A_LX;          // [16][A_LX] kh
  lptr s_red = s_k + 16 * A_LX;        // [8] max, [8] packed position
  MLHOT_TSTAMP_AT(16, d.T);
  // ---- every global read of the block, up front
  Tile64 xt;
  FoldTile<XK> xf;
  const bool folding = a.xslab != nullptr;              // kernel-uniform
  if (folding) xf.issue(a.xslab, a.xbias, a.xn, t * d.Nc, d.Nc, tid);
  else xt.fetch(a.cat_in + (size_t)t * d.Nc * LDC, LDC, d.Nc, tid);
  const float* wk = a.p.wk_w[0]; const float* bk = a.p.wk_b[0];
#pragma unroll
  for (int i = 1; i < H; ++i)
    if (h == i) { wk = a.p.wk_w[i]; bk = a.p.wk_b[i]; }
  f32x4_t wkf[4]; float kbias = 0.f;
  if (wave < 4) {
    kbias = bk[wave * 16 + lr];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) wkf[kb] = *reinterpret_cast<const f32x4_t*>(wk + (wave * 16 + lr) * DW + kb * 16 + 4 * lq);
  }
  constexpr int NTILE = (M + 15) / 16;             // 17 feature tiles: waves take tiles wave, wave + 8, wave + 16
  f32x4_t pf[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int jt = wave + 8 * i;
    if (jt < NTILE) {
      const int j = jt * 16 + lr, jc = j < M ? j : M - 1;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) pf[i][kb] = *reinterpret_cast<const f32x4_t*>(a.p.proj + (size_t)jc * DW + kb * 16 + 4 * lq);
    }
  }
  // head-major copy of _W's weight for phase B (forward and backward): wot[h][j][e] = Wo[j][e*H + h]; the T blocks of a head
  // copy 1/T of its [dw][dw] block each
  {
    const int per = (DW * DW + d.T - 1) / d.T, lo = t * per, hi = lo + per < DW * DW ? lo + per : DW * DW;
    for (int i = lo + tid; i < hi; i += NWV * 64) {
      const int j = i / DW, e = i - j * DW;
      a.wot[(size_t)h * DW * DW + i] = a.p.wo_w[(size_t)j * HD + e * H + h];
    }
  }
  __builtin_amdgcn_sched_barrier(0);               // every request above is out before the first wait
  if (folding) xt.v = xf.finish();
  xt.stash(s_x, A_LX, tid);                        // all 16 rows (zeros beyond Nc): nothing else of the tiles is ever read
  __syncthreads();
  MLHOT_TSTAMP_AT(17, d.T);
  if (wave < 4) {                                   // kh tile of wave: 16 columns
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const f32x4_t x = *reinterpret_cast<lc4ptr>(s_x + lr * A_LX + kb * 16 + 4 * lq);
      if (kb & 1) { a1 = mfma4(x[0], wkf[kb][0], a1); a1 = mfma4(x[1], wkf[kb][1], a1); a1 = mfma4(x[2], wkf[kb][2], a1); a1 = mfma4(x[3], wkf[kb][3], a1); }
      else { a0 = mfma4(x[0], wkf[kb][0], a0); a0 = mfma4(x[1], wkf[kb][1], a0); a0 = mfma4(x[2], wkf[kb][2], a0); a0 = mfma4(x[3], wkf[kb][3], a0); }
    }
    const int n = wave * 16 + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      const float v = a0[r] + a1[r] + kbias;
      s_k[row * A_LX + n] = row < d.Nc ? v : 0.f;       // rows >= Nc are zero: they must not enter the max below
      if (row < d.Nc) a.kh[(size_t)(t * d.Nc + row) * HD + h * DW + n] = v;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP_AT(18, d.T);
  const float c = powf((float)DW, -0.25f);
  float best = -INFINITY; int bcode = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int jt = wave + 8 * i;
    if (jt < NTILE) {
      const int j = jt * 16 + lr;
      f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4_t x = *reinterpret_cast<lc4ptr>(s_k + lr * A_LX + kb * 16 + 4 * lq);
        const f32x4_t b = pf[i][kb] * c;
        if (kb & 1) { a1 = mfma4(x[0], b[0], a1); a1 = mfma4(x[1], b[1], a1); a1 = mfma4(x[2], b[2], a1); a1 = mfma4(x[3], b[3], a1); }
        else { a0 = mfma4(x[0], b[0], a0); a0 = mfma4(x[1], b[1], a0); a0 = mfma4(x[2], b[2], a0); a0 = mfma4(x[3], b[3], a0); }
      }
      if (j < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          if (row < d.Nc) {
            const float v = a0[r] + a1[r];
            const int code = ((t * d.Nc + row) * H + h) * 4096 + j;      // row index of the [T*Nc*H, m] view, column
            if (kmax_better(v, code, best, bcode)) { best = v; bcode = code; }
          }
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
    for (int w = 1; w < NWV; ++w)
      if (kmax_better(s_red[w], s_redi[w], best, bcode)) { best = s_red[w]; bcode = s_redi[w]; }
    a.tmax[th] = best; a.targ[th] = bcode;
  }
  MLHOT_TSTAMP_AT(19, d.T);
}

__global__ __launch_bounds__(512) void phaseA_fwd_kernel(const PhaseAArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  if ((int)blockIdx.x >= d.T + d.T * H) {
    // blocks behind the key heads (only launched with a.xslab): fold the query rows of task tq into dec_in[:, :dw] - phase B's
    // query projection, phase C and the backward read them there
    const int tq = blockIdx.x - d.T - d.T * H;
    FoldTile<XK> xq;
    xq.issue(a.xslab, a.xbias, a.xn, d.T * d.Nc + tq * d.Nq, d.Nq, tid);
    const int row = (tid >> 4) & 15, c4 = tid & 15;
    if (xq.on) *reinterpret_cast<f32x4_t*>(a.dec_in + ((size_t)tq * d.Nq + row) * LDD + 4 * c4) = xq.finish();
    return;
  }
  if ((int)blockIdx.x >= d.T) { keyhead_block(a, L0, blockIdx.x - d.T, tid); return; }
  const int t = blockIdx.x;
  lptr s_cat = L0;                     // [16][A_LCAT]  [x_ctx | transform_y(ctx_y)]
  MLHOT_TSTAMP(0);
  lptr s_h0 = s_cat + 16 * A_LCAT;
  lptr s_h1 = s_h0 + 16 * A_LH;
  lptr s_y = s_h1 + 16 * A_LH;         // [16][A_LY] labels
  lptr s_red = s_y + 16 * A_LY;        // [8 waves][256] K-split partials
  // ---- every global read of the block, up front
  Tile64 xt;
  FoldTile<XK> xf;
  const bool folding = a.xslab != nullptr;              // kernel-uniform
  if (folding) xf.issue(a.xslab, a.xbias, a.xn, t * d.Nc, d.Nc, tid);
  else xt.fetch(a.cat_in + (size_t)t * d.Nc * LDC, LDC, d.Nc, tid);
  float yv = 0.f;
  if (tid < 256 && (tid >> 4) < d.Nc && (tid & 15) < d.label_dim) yv = a.ctx_y[((size_t)t * d.Nc + (tid >> 4)) * d.label_dim + (tid & 15)];
  Lin<4, DW / 4> l_ty;  Lin<LDC, H0> l_e0;  Lin<H0, H1> l_e1;  Lin<H1, DW> l_e2;
  // transform_y: K = label_dim (1..4, run time) rides in a K = 4 layer through the scalar path
  if (wave == 0) {
    const int lr = lane & 15, lq = lane >> 4;
    l_ty.bias = a.p.ty_b[lr];
    l_ty.b[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (lq == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) l_ty.b[0][e] = a.p.ty_w[lr * d.label_dim + (e < d.label_dim ? e : d.label_dim - 1)];
    }
  }
  l_e0.load(a.p.er_w[0], a.p.er_b[0], wave, lane);
  l_e1.load(a.p.er_w[1], a.p.er_b[1], wave, lane);
  l_e2.load(a.p.er_w[2], a.p.er_b[2], wave, lane);
  // pc = c * P for the later phases: each task writes its slice
  {
    const float c = powf((float)DW, -0.25f);
    constexpr int n = M * DW;
    const int per = ((n / 4 + d.T - 1) / d.T) * 4, lo = t * per, hi = lo + per < n ? lo + per : n;
    for (int i = lo + 4 * tid; i < hi; i += 4 * NWV * 64) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(a.p.proj + i);
      *reinterpret_cast<f32x4_t*>(a.pc + i) = v * c;
    }
  }
  MLHOT_TSTAMP(1);
  // no LDS zeroing: every tile is written in full (x rows via Tile64, labels as a 16 x 16 block, layer outputs incl. their
  // zero padding columns by Lin::finish) before it is read
  __builtin_amdgcn_sched_barrier(0);               // every request above is out before the first wait
  if (folding) xt.v = xf.finish();
  xt.stash(s_cat, A_LCAT, tid);
  if (tid < 256) s_y[(tid >> 4) * A_LY + (tid & 15)] = yv;
  float* g_cat = a.cat_in + (size_t)t * d.Nc * LDC;
  if (folding && tid < 256 && ((tid >> 4) & 15) < d.Nc)       // the folded context rows: phase B' / A' and the encoder backward read them in cat_in
    *reinterpret_cast<f32x4_t*>(g_cat + (size_t)((tid >> 4) & 15) * LDC + 4 * (tid & 15)) = xt.v;
  __syncthreads();
  MLHOT_TSTAMP(2);
  // transform_y -> cat[:, dw:]   (padding columns of s_y are zero, so the clamped weight columns contribute nothing)
  if (wave == 0) {
    const int lr = lane & 15, lq = lane >> 4;
    const f32x4_t x = *reinterpret_cast<lc4ptr>(s_y + lr * A_LY + 4 * lq);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma4(x[0], l_ty.b[0][0], acc); acc = mfma4(x[1], l_ty.b[0][1], acc); acc = mfma4(x[2], l_ty.b[0][2], acc); acc = mfma4(x[3], l_ty.b[0][3], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      const float v = acc[r] + l_ty.bias;
      s_cat[row * A_LCAT + DW + lr] = v;
      if (row < d.Nc) g_cat[row * LDC + DW + lr] = v;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(3);
  l_e0.finish(l_e0.mma(s_cat, A_LCAT, wave, lane), ACT_RELU, s_red, s_h0, A_LH, a.h0 + (size_t)t * d.Nc * H0, H0, d.Nc, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(4);
  l_e1.finish(l_e1.mma(s_h0, A_LH, wave, lane), ACT_RELU, s_red, s_h1, A_LH, a.h1 + (size_t)t * d.Nc * H1, H1, d.Nc, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(5);
  l_e2.finish(l_e2.mma(s_h1, A_LH, wave, lane), ACT_NONE, s_red, nullptr, 0, a.rs + (size_t)t * d.Nc * DW, DW, d.Nc, wave, lane);
  MLHOT_TSTAMP(6);
}

// ==================================================================================================
// phase B forward, one workgroup per (task, head): this head's query / value projections, FAVOR+ in the S-form, the head's
// share of _W(merged).  Same outputs as tf::phaseB_fwd_kernel.
// ==================================================================================================
constexpr int B_LX = lds_ld(DW), B_LF = lds_ld(M);
constexpr int B_FLOATS = 16 * (5 * B_LX + 2 * B_LF) + 8 * 16 * 17 + 16 * 17 + 64 + 16;
__host__ inline size_t phaseB_lds_bytes() { return sizeof(float) * B_FLOATS; }

__global__ __launch_bounds__(512) void phaseB_fwd_kernel(const PhaseBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  lptr s_q = L0;                      // [16][B_LX]
  lptr s_k = s_q + 16 * B_LX;
  lptr s_v = s_k + 16 * B_LX;
  lptr s_xq = s_v + 16 * B_LX;        // x_qry, later this head's attention output
  lptr s_rs = s_xq + 16 * B_LX;
  lptr s_qf = s_rs + 16 * B_LX;       // [16][B_LF]  dd -> E
  lptr s_kf = s_qf + 16 * B_LF;
  lptr s_Sp = s_kf + 16 * B_LF;       // [8 waves][16][17] partial S
  lptr s_S = s_Sp + 8 * 16 * 17;      // [16][17] S
  lptr s_st = s_S + 16 * 17;          // diag_q[16], diag_k[16], max_q[16], D[16]; [64]: the batch-global key maximum
  MLHOT_TSTAMP(32);
  // ---- every global read of the block, up front
  Tile64 xq, xr, xk;
  xq.fetch(a.dec_in + (size_t)t * d.Nq * LDD, LDD, d.Nq, tid);
  xr.fetch(a.rs + (size_t)t * d.Nc * DW, DW, d.Nc, tid);
  xk.fetch(a.kh + (size_t)t * d.Nc * HD + h * DW, HD, d.Nc, tid);
  float gm = -INFINITY; int gcode = 0x7fffffff;
  if (wave == 0) {
    for (int i = lane; i < d.T * H; i += 64) {
      const float v = a.tmax[i]; const int cd = a.targ[i];
      if (kmax_better(v, cd, gm, gcode)) { gm = v; gcode = cd; }
    }
  }
  const float *wq = a.p.wq_w[0], *bq = a.p.wq_b[0], *wv = a.p.wv_w[0], *bv = a.p.wv_b[0];
#pragma unroll
  for (int i = 1; i < H; ++i)
    if (h == i) { wq = a.p.wq_w[i]; bq = a.p.wq_b[i]; wv = a.p.wv_w[i]; bv = a.p.wv_b[i]; }
  const bool isv = wave >= 4;                       // waves 0-3: query tiles, 4-7: value tiles
  const int pn = (wave & 3) * 16 + lr;
  f32x4_t wpf[4], wof[4];
  const float pbias = (isv ? bv : bq)[pn];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) wpf[kb] = *reinterpret_cast<const f32x4_t*>((isv ? wv : wq) + pn * DW + kb * 16 + 4 * lq);
  if (wave < 4) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) wof[kb] = *reinterpret_cast<const f32x4_t*>(a.wot + (size_t)h * DW * DW + (16 * wave + lr) * DW + kb * 16 + 4 * lq);
  }
  constexpr int NTILE = (M + 15) / 16;
  f32x4_t pf[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int jt = wave + 8 * i;
    if (jt < NTILE) {
      const int j = jt * 16 + lr, jc = j < M ? j : M - 1;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) pf[i][kb] = *reinterpret_cast<const f32x4_t*>(a.pc + (size_t)jc * DW + kb * 16 + 4 * lq);
    }
  }
  MLHOT_TSTAMP(33);
  xq.stash(s_xq, B_LX, tid);
  xr.stash(s_rs, B_LX, tid);
  xk.stash(s_k, B_LX, tid);
  // batch-global key stabiliser (identical in every workgroup): largest share, first position on ties
  if (wave == 0) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(gm, off, 64);
      const int oc = __shfl_xor(gcode, off, 64);
      if (kmax_better(ov, oc, gm, gcode)) { gm = ov; gcode = oc; }
    }
    if (lane == 0) {
      s_st[64] = gm;
      if (blockIdx.x == 0) { a.gmax[0] = gm; a.gpos[0] = gcode >> 12; a.gpos[1] = gcode & 4095; }
    }
  }
  __syncthreads();
  gm = s_st[64];
  MLHOT_TSTAMP(34);
  // this head's query and value projections: qh = W_q,h(x_qry), vh = W_v,h(rs)
  {
    lcptr xs = isv ? s_rs : s_xq;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const f32x4_t x = *reinterpret_cast<lc4ptr>(xs + lr * B_LX + kb * 16 + 4 * lq);
      if (kb & 1) { a1 = mfma4(x[0], wpf[kb][0], a1); a1 = mfma4(x[1], wpf[kb][1], a1); a1 = mfma4(x[2], wpf[kb][2], a1); a1 = mfma4(x[3], wpf[kb][3], a1); }
      else { a0 = mfma4(x[0], wpf[kb][0], a0); a0 = mfma4(x[1], wpf[kb][1], a0); a0 = mfma4(x[2], wpf[kb][2], a0); a0 = mfma4(x[3], wpf[kb][3], a0); }
    }
    lptr ys = isv ? s_v : s_q;
    float* yg = isv ? a.vh : a.qh;
    const int nrows = isv ? d.Nc : d.Nq;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      const float v = a0[r] + a1[r] + pbias;
      ys[row * B_LX + pn] = row < nrows ? v : 0.f;      // rows beyond the shots are zeros (no LDS zeroing in this kernel)
      if (row < nrows) yg[(size_t)(t * nrows + row) * HD + h * DW + pn] = v;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(35);
  // dd tiles: q and k against pc (shared B operand, two independent accumulator chains)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int jt = wave + 8 * i;
    if (jt < NTILE) {
      const int j = jt * 16 + lr;
      f32x4_t accq = {0.f, 0.f, 0.f, 0.f}, acck = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const f32x4_t xqv = *reinterpret_cast<lc4ptr>(s_q + lr * B_LX + kb * 16 + 4 * lq);
        const f32x4_t xkv = *reinterpret_cast<lc4ptr>(s_k + lr * B_LX + kb * 16 + 4 * lq);
        const f32x4_t b = pf[i][kb];
        accq = mfma4(xqv[0], b[0], accq); acck = mfma4(xkv[0], b[0], acck);
        accq = mfma4(xqv[1], b[1], accq); acck = mfma4(xkv[1], b[1], acck);
        accq = mfma4(xqv[2], b[2], accq); acck = mfma4(xkv[2], b[2], acck);
        accq = mfma4(xqv[3], b[3], accq); acck = mfma4(xkv[3], b[3], acck);
      }
      if (j < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_qf[(4 * lq + r) * B_LF + j] = accq[r]; s_kf[(4 * lq + r) * B_LF + j] = acck[r]; }
      }
    }
  }
  // diag = c^2/2 |x|^2 : 32 rows (16 q + 16 k), 16 threads per row
  {
    const float half_c2 = 0.5f / sqrtf((float)DW);
    const int row = tid >> 4, part = tid & 15;
    lcptr xrow = (row < 16 ? s_q + row * B_LX : s_k + (row - 16) * B_LX);
    const f32x4_t x = *reinterpret_cast<lc4ptr>(xrow + 4 * part);
    float s = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    if (part == 0) s_st[row] = s * half_c2;
  }
  __syncthreads();
  MLHOT_TSTAMP(36);
  // query row max / first arg-max: 16 rows x 32 threads (tried: the maximum riding in the dd tiles' epilogue - the 32 cross-lane
  // steps per wave cost more than this pass and its barrier: 14.8 -> 16.6 us)
  {
    const int row = tid >> 5, part = tid & 31;
    float best = -INFINITY; int arg = 0x7fffffff;
    for (int j = part; j < M; j += 32) { const float v = s_qf[row * B_LF + j]; if (v > best) { best = v; arg = j; } }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const float ov = __shfl_xor(best, off, 64); const int oa = __shfl_xor(arg, off, 64);
      if (ov > best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (part == 0) {
      s_st[32 + row] = best;
      if (row < d.Nq) a.arg_q[(t * d.Nq + row) * H + h] = arg;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(37);
  // E features in place (padding columns j >= m stay exactly 0), valid rows saved for the backward right away
  const float ratio = 1.0f / sqrtf((float)M), re = ratio * 1e-4f;
  for (int row = wave; row < 16; row += NWV) {
    const float sq = s_st[row] + s_st[32 + row], sk = s_st[16 + row] + gm;
    float* gq = a.qf + ((size_t)(t * d.Nq + row) * H + h) * M;
    float* gk = a.kf + ((size_t)(t * d.Nc + row) * H + h) * M;
    for (int j = lane; j < M; j += 64) {
      const float eq = ratio * expf(s_qf[row * B_LF + j] - sq), ek = ratio * expf(s_kf[row * B_LF + j] - sk);
      s_qf[row * B_LF + j] = eq;
      s_kf[row * B_LF + j] = ek;
      if (row < d.Nq) gq[j] = eq;
      if (row < d.Nc) gk[j] = ek;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(38);
  // S = (Eq + re)(Ek + re)^T : M = 16 q rows, N = 16 k rows, K = m split over the 8 waves
  {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = wave * 4; j0 < M; j0 += 32) {
      const int j = j0 + lq;
      const bool vj = j < M;
      const float av = vj ? s_qf[lr * B_LF + j] + re : 0.f;
      const float bvv = vj ? s_kf[lr * B_LF + j] + re : 0.f;
      acc = mfma4(av, bvv, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_Sp[(wave * 16 + 4 * lq + r) * 17 + lr] = acc[r];
  }
  __syncthreads();
  MLHOT_TSTAMP(39);
  if (tid < 256) {                                  // fold the partials; D = row sums (16 consecutive lanes hold a row)
    const int n = tid >> 4, np = tid & 15;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += s_Sp[(16 * w + n) * 17 + np];
    if (n >= d.Nq || np >= d.Nc) s = 0.f;
    s_S[n * 17 + np] = s;
    if (n < d.Nq && np < d.Nc) a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + np] = s;
    float dsum = s;
    dsum += __shfl_xor(dsum, 1, 64); dsum += __shfl_xor(dsum, 2, 64); dsum += __shfl_xor(dsum, 4, 64); dsum += __shfl_xor(dsum, 8, 64);
    if (np == 0) {
      s_st[48 + n] = dsum;
      if (n < d.Nq) a.D[((size_t)t * H + h) * d.Nq + n] = dsum;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(40);
  // out[n][e] = sum_n' S[n][n'] v[n'][e] / D[n]: one e tile per wave 0-3, K = 16 k rows
  if (wave < 4) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int np = 4 * s4 + lq;
      acc = mfma4(s_S[lr * 17 + np], s_v[np * B_LX + wave * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, e = wave * 16 + lr;
      const float o = n < d.Nq ? acc[r] / s_st[48 + n] : 0.f;
      if (n < d.Nq) a.merged[(size_t)(t * d.Nq + n) * HD + e * H + h] = o;
      s_xq[n * B_LX + e] = o;                          // x_qry is no longer needed: the tile now holds this head's output
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(41);
  // share of rr = _W(merged): rrp[n][j] = sum_e out[n][e] Wo[j][e*H + h], Wo_h from the head-major copy
  if (wave < 4) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const f32x4_t x = *reinterpret_cast<lc4ptr>(s_xq + lr * B_LX + kb * 16 + 4 * lq);
      if (kb & 1) { a1 = mfma4(x[0], wof[kb][0], a1); a1 = mfma4(x[1], wof[kb][1], a1); a1 = mfma4(x[2], wof[kb][2], a1); a1 = mfma4(x[3], wof[kb][3], a1); }
      else { a0 = mfma4(x[0], wof[kb][0], a0); a0 = mfma4(x[1], wof[kb][1], a0); a0 = mfma4(x[2], wof[kb][2], a0); a0 = mfma4(x[3], wof[kb][3], a0); }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r;
      if (n < d.Nq) a.rrp[((size_t)(t * H + h) * d.Nq + n) * DW + 16 * wave + lr] = a0[r] + a1[r];
    }
  }
  MLHOT_TSTAMP(42);
}

// ==================================================================================================
// phase C forward, one workgroup per task: rr = _W(merged) (bias + the 8 heads' shares); z = r_to_z(rr) -> dec_in[:, dw:];
// d1, d2 = decoder hidden; mu = act(decoder out).
// ==================================================================================================
constexpr int C_LR = lds_ld(DW), C_LD = lds_ld(LDD), C_LH = lds_ld(DH);
constexpr int C_FLOATS = 16 * (C_LR + C_LD + 2 * C_LH) + NWV * 256;
__host__ inline size_t phaseC_lds_bytes() { return sizeof(float) * C_FLOATS; }

__global__ __launch_bounds__(512) void phaseC_fwd_kernel(const PhaseCArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  lptr s_rr = L0;                  // [16][C_LR]
  lptr s_dec = s_rr + 16 * C_LR;   // [16][C_LD]  [x_qry | z]
  lptr s_d1 = s_dec + 16 * C_LD;
  lptr s_d2 = s_d1 + 16 * C_LH;
  lptr s_red = s_d2 + 16 * C_LH;   // [8 waves][256] K-split partials
  MLHOT_TSTAMP(64);
  // ---- every global read of the block, up front
  Tile64 xq;
  xq.fetch(a.dec_in + (size_t)t * d.Nq * LDD, LDD, d.Nq, tid);
  // rr = _W(merged) = bias + the 8 heads' shares (fixed order): thread = (row tid / 32, columns 2 (tid % 32) .. + 1)
  float2 rv[H]; float2 rb = make_float2(0.f, 0.f);
  const int rrow = tid >> 5, rcol = 2 * (tid & 31);
  if (rrow < d.Nq) {
    rb = *reinterpret_cast<const float2*>(a.p.wo_b + rcol);
#pragma unroll
    for (int h = 0; h < H; ++h) rv[h] = *reinterpret_cast<const float2*>(a.rrp + ((size_t)(t * H + h) * d.Nq + rrow) * DW + rcol);
  }
  Lin<DW, DZ> l_z;  Lin<LDD, DH> l_d0;  Lin<DH, DH> l_d1;  Lin<DH, 4> l_d2;
  l_z.load(a.p.r2z_w, a.p.r2z_b, wave, lane);
  l_d0.load(a.p.dec_w[0], a.p.dec_b[0], wave, lane);
  l_d1.load(a.p.dec_w[1], a.p.dec_b[1], wave, lane);
  l_d2.load(a.p.dec_w[2], a.p.dec_b[2], wave, lane, DH, d.y_dim);
  MLHOT_TSTAMP(65);
  xq.stash(s_dec, C_LD, tid);
  {
    float2 sum = make_float2(0.f, 0.f);                // rows >= Nq: zeros (every tile is written in full, no LDS zeroing)
    if (rrow < d.Nq) {
      sum = rb;
#pragma unroll
      for (int h = 0; h < H; ++h) { sum.x += rv[h].x; sum.y += rv[h].y; }
      *reinterpret_cast<float2*>(a.rr + ((size_t)t * d.Nq + rrow) * DW + rcol) = sum;
    }
    s_rr[rrow * C_LR + rcol] = sum.x; s_rr[rrow * C_LR + rcol + 1] = sum.y;
  }
  __syncthreads();
  MLHOT_TSTAMP(66);
  float* g_dec = a.dec_in + (size_t)t * d.Nq * LDD;
  l_z.finish(l_z.mma(s_rr, C_LR, wave, lane), ACT_NONE, s_red, s_dec + DW, C_LD, g_dec + DW, LDD, d.Nq, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(67);
  l_d0.finish(l_d0.mma(s_dec, C_LD, wave, lane), ACT_RELU, s_red, s_d1, C_LH, a.d1 + (size_t)t * d.Nq * DH, DH, d.Nq, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(68);
  l_d1.finish(l_d1.mma(s_d1, C_LH, wave, lane), ACT_RELU, s_red, s_d2, C_LH, a.d2 + (size_t)t * d.Nq * DH, DH, d.Nq, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(69);
  l_d2.finish(l_d2.mma(s_d2, C_LH, wave, lane), d.out_act, s_red, nullptr, 0, a.mu + (size_t)t * d.Nq * d.y_dim, d.y_dim, d.Nq, wave, lane, d.y_dim);
  MLHOT_TSTAMP(70);
}

// ==================================================================================================
// backward building blocks
// ==================================================================================================

// dX[16 x KIN] = dY[16 x NOUT] W[NOUT x KIN], W fragments in registers (4 dwords per 16 rows of W, coalesced along KIN).
// Items = (16-column tile of KIN) x (chunk of the j blocks), item w on wave w; dY in LDS with zero padding up to
// pad16(NOUT) columns.  finish(): fold, optional ReLU mask from the saved activation tile (same layout as the result),
// zero padding columns up to pad16(KIN) in the LDS result, rows < nrows to global.
template <int NOUT, int KIN>
struct Dg {
  static constexpr int JB = (NOUT + 15) / 16, NI = (KIN + 15) / 16;
  static constexpr int NCH0 = NI >= 5 ? 1 : NI >= 3 ? 2 : NI == 2 ? 4 : 8;
  static constexpr int NCH = NCH0 < JB ? NCH0 : JB;
  static constexpr int PER = (JB + NCH - 1) / NCH;
  float b[PER][4];

  __device__ __forceinline__ static bool active(int wave) { return wave < NI * NCH; }

  __device__ __forceinline__ void load(const float* __restrict__ W, int wave, int lane, int nout = NOUT) {
#pragma unroll
    for (int p = 0; p < PER; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[p][e] = 0.f;
    if (!active(wave)) return;
    const int lr = lane & 15, lq = lane >> 4;
    const int tile = wave % NI, chunk = wave / NI;
    const int i = tile * 16 + lr, ic = i < KIN ? i : KIN - 1;
#pragma unroll
    for (int p = 0; p < PER; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = (chunk * PER + p) * 16 + 4 * lq + e;
        b[p][e] = W[(j < nout ? j : nout - 1) * KIN + ic];
      }
  }

  __device__ __forceinline__ f32x4_t mma(lcptr dys, int ldy, int wave, int lane) const {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    if (active(wave)) {
      const int lr = lane & 15, lq = lane >> 4;
      const int chunk = wave / NI;
      lcptr yr = dys + lr * ldy + 4 * lq;
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        const int jb = chunk * PER + p;
        if (jb < JB) {
          const f32x4_t a = *reinterpret_cast<lc4ptr>(yr + jb * 16);
          if (p & 1) { a1 = mfma4(a[0], b[p][0], a1); a1 = mfma4(a[1], b[p][1], a1); a1 = mfma4(a[2], b[p][2], a1); a1 = mfma4(a[3], b[p][3], a1); }
          else { a0 = mfma4(a[0], b[p][0], a0); a0 = mfma4(a[1], b[p][1], a0); a0 = mfma4(a[2], b[p][2], a0); a0 = mfma4(a[3], b[p][3], a0); }
        }
      }
    }
    return a0 + a1;
  }

  // Call from ALL waves (one barrier inside when NCH > 1).
  __device__ __forceinline__ void finish(f32x4_t acc, lptr red, lcptr relu_of, int ldm, lptr dxs, int ldxs, float* __restrict__ dxg, int ldg,
                                         int nrows, int wave, int lane) const {
    const int lr = lane & 15, lq = lane >> 4;
    const int tile = wave % NI, chunk = wave / NI;
    if constexpr (NCH > 1) {
      if (active(wave) && chunk > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
      }
      __syncthreads();
      if (active(wave) && chunk == 0) {
#pragma unroll
        for (int c = 1; c < NCH; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] += red[((wave + c * NI) * 4 + r) * 64 + lane];
      }
    }
    if (active(wave) && chunk == 0) {
      const int i = tile * 16 + lr;
      const bool vi = i < KIN;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = vi ? acc[r] : 0.f;
        if (relu_of != nullptr) v = relu_of[(4 * lq + r) * ldm + i] > 0.f ? v : 0.f;
        if (dxs != nullptr) dxs[(4 * lq + r) * ldxs + i] = v;
        if (dxg != nullptr && vi && 4 * lq + r < nrows) dxg[(4 * lq + r) * ldg + i] = v;
      }
    }
  }
};

// dW[NOUT x KIN] = dY^T X over the 16 rows (padded rows of dY are zero), db = column sums of dY; both to the task's slab.
// Output tiles (16 j x 16 i) round-robin over the waves, 4 MFMAs each, operands straight from LDS.
// GROUPS > 1: the tiles are dealt over GROUPS workgroups (this one is `grp`; `wave` = grp * NWV + the wave's index), the bias
// sums belong to group 0.
template <int NOUT, int KIN, int GROUPS = 1>
__device__ __forceinline__ void wgrad16(lcptr dys, int ldy, lcptr xs, int ldx, float* __restrict__ dw, float* __restrict__ db,
                                        int wave, int lane, int tid, int nout = NOUT, int kin = KIN) {
  constexpr int NJ = (NOUT + 15) / 16, NI = (KIN + 15) / 16, TRIPS = (NJ * NI + GROUPS * NWV - 1) / (GROUPS * NWV);
  const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int tr = 0; tr < TRIPS; ++tr) {
    const int it = wave + GROUPS * NWV * tr;
    if (it < NJ * NI) {
      const int jt = it / NI, j0 = jt * 16, i0 = (it - jt * NI) * 16;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int row = 4 * s4 + lq;
        acc = mfma4(dys[row * ldy + j0 + lr], xs[row * ldx + i0 + lr], acc);
      }
      const int i = i0 + lr;
      if (i < kin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + 4 * lq + r;
          if (j < nout) dw[j * kin + i] = acc[r];
        }
      }
    }
  }
  if (db != nullptr && tid < nout) {
    float sum = 0.f;
#pragma unroll
    for (int row = 0; row < 16; ++row) sum += dys[row * ldy + tid];
    db[tid] = sum;
  }
}

// rows < nrows of a [rows][W] global matrix (row stride ldg, W % 4 == 0, W <= 128) -> registers -> LDS tile [16][ld] with
// zeros in the rows >= nrows and in the padding columns up to pad16(W): one float4 per thread
template <int W>
struct TileW {
  static constexpr int C4 = pad16(W) / 4;            // float4 per row incl. padding
  f32x4_t v;
  __device__ __forceinline__ void fetch(const float* __restrict__ src, int ldg, int nrows, int tid) {
    const int row = tid / C4, c4 = tid - row * C4;
    v = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (row < nrows && 4 * c4 < W) v = *reinterpret_cast<const f32x4_t*>(src + (size_t)row * ldg + 4 * c4);
  }
  __device__ __forceinline__ void stash(lptr dst, int ld, int tid) const {
    const int row = tid / C4, c4 = tid - row * C4;
    if (row < 16) *reinterpret_cast<MLHOT_LDS f32x4_t*>(dst + row * ld + 4 * c4) = v;
  }
};

// ==================================================================================================
// phase C backward, one workgroup per task: decoder0, r_to_z backward and _W's bias gradient (see tf::phaseC_bwd_kernel).
// The data gradient of layer k+1 and the weight gradient of layer k run between the same two barriers.
// ==================================================================================================
constexpr int CB_LY = 20;
constexpr int CB_FLOATS = 16 * (CB_LY + 4 * C_LH + 2 * C_LD + 2 * C_LR) + NWV * 256;
__host__ inline size_t phaseC_bwd_lds_bytes() { return sizeof(float) * CB_FLOATS; }

// GR > 1 (round 5): GR workgroups per task.  The data-gradient chain is short and serial - all of them walk it - but the weight
// gradients riding in its barrier intervals (7 + 49 + 56 + 16 tiles of 4 MFMAs and their scattered stores) are dealt over the group's
// 8 GR waves; workgroup 0 of a group alone writes d_dec_in, d_rr and the bias sums.  Measured (one box): 1 -> 2 -> 4 workgroups:
// 14.9 -> 13.1 -> 12.6 us.
template <int GR>
__global__ __launch_bounds__(512) void phaseC_bwd_kernel(const PhaseCBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  if (a.loss.value != nullptr && (int)blockIdx.x == GR * d.T) {            // one workgroup more than the tasks need: the loss VALUE (ops_direct.h)
    loss_value_block(LossRed{a.loss.kind, d.y_dim, a.loss.gt_dim, d.T * d.Nq, a.mu, a.loss.gt, a.loss.value}, d.T * d.Nq, lds);
    return;
  }
  const int t = (int)blockIdx.x / GR, grp = (int)blockIdx.x % GR;          // GR workgroups per task
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), gwave = grp * NWV + wave;
  const bool first = grp == 0;
  lptr s_g = L0;                   // [16][CB_LY]   dmu * act'(mu)
  lptr s_d2 = s_g + 16 * CB_LY;     // saved activations
  lptr s_d1 = s_d2 + 16 * C_LH;
  lptr s_dec = s_d1 + 16 * C_LH;
  lptr s_rr = s_dec + 16 * C_LD;
  lptr s_dd2 = s_rr + 16 * C_LR;    // gradients
  lptr s_dd1 = s_dd2 + 16 * C_LH;
  lptr s_ddec = s_dd1 + 16 * C_LH;
  lptr s_drr = s_ddec + 16 * C_LD;
  lptr s_red = s_drr + 16 * C_LR;   // [8 waves][256] partial tiles
  const size_t rq = (size_t)t * d.Nq;
  // ---- every global read of the block, up front
  float gv = 0.f;
  MLHOT_TSTAMP(96);
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    if (r < d.Nq && c < d.y_dim) {
      float up = a.dmu != nullptr ? a.dmu[(rq + r) * d.y_dim + c] : 0.f;
      if (a.loss.kind >= 0) {
        // the loss's own gradient (trainer/losses.py:59-61 and its siblings) needs no reduction: every thread of a row derives the
        // row's y_dim <= 4 entries from mu and the labels and keeps its column - LossBwd's launch (4.6 us between the loss and this
        // kernel) is gone from the step's dependency chain
        float dl[8];
        loss_row_grad(a.loss.kind, d.y_dim, d.T * d.Nq, a.mu + (rq + r) * d.y_dim, a.loss.gt + (rq + r) * a.loss.gt_dim, a.loss.dloss[0], dl);
        float mine = dl[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) mine = c == j ? dl[j] : mine;
        up += mine;
      }
      gv = up * act_grad_from_out(d.out_act, a.mu[(rq + r) * d.y_dim + c]);
    }
  }
  TileW<DH> td2, td1; TileW<LDD> tdec; TileW<DW> trr;
  td2.fetch(a.d2 + rq * DH, DH, d.Nq, tid);
  td1.fetch(a.d1 + rq * DH, DH, d.Nq, tid);
  tdec.fetch(a.dec_in + rq * LDD, LDD, d.Nq, tid);
  trr.fetch(a.rr + rq * DW, DW, d.Nq, tid);
  Dg<4, DH> g2;  Dg<DH, DH> g1;  Dg<DH, LDD> g0;  Dg<DZ, DW> gz;
  g2.load(a.p.dec_w[2], wave, lane, d.y_dim);
  g1.load(a.p.dec_w[1], wave, lane);
  g0.load(a.p.dec_w[0], wave, lane);
  gz.load(a.p.r2z_w, wave, lane);
  if (tid < 256) s_g[(tid >> 4) * CB_LY + (tid & 15)] = gv;
  td2.stash(s_d2, C_LH, tid); td1.stash(s_d1, C_LH, tid); tdec.stash(s_dec, C_LD, tid); trr.stash(s_rr, C_LR, tid);
  __syncthreads();
  MLHOT_TSTAMP(97);
  float* sl = a.slab + (size_t)t * a.sl.total;
  // decoder0.4: d d2
  g2.finish(g2.mma(s_g, CB_LY, wave, lane), s_red, s_d2, C_LH, s_dd2, C_LH, nullptr, 0, 0, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(98);
  // decoder0.2: d d1  |  decoder0.4 weight gradient
  g1.finish(g1.mma(s_dd2, C_LH, wave, lane), s_red, s_d1, C_LH, s_dd1, C_LH, nullptr, 0, 0, wave, lane);
  wgrad16<4, DH, GR>(s_g, CB_LY, s_d2, C_LH, sl + a.sl.dec_w[2], first ? sl + a.sl.dec_b[2] : nullptr, gwave, lane, tid, d.y_dim, DH);
  __syncthreads();
  MLHOT_TSTAMP(99);
  // decoder0.0: input gradient = [d x_qry | dz]  |  decoder0.2 weight gradient
  g0.finish(g0.mma(s_dd1, C_LH, wave, lane), s_red, nullptr, 0, s_ddec, C_LD, first ? a.d_dec_in + rq * LDD : nullptr, LDD, d.Nq, wave, lane);
  wgrad16<DH, DH, GR>(s_dd2, C_LH, s_d1, C_LH, sl + a.sl.dec_w[1], first ? sl + a.sl.dec_b[1] : nullptr, gwave, lane, tid);
  __syncthreads();
  MLHOT_TSTAMP(100);
  // r_to_z (dz = s_ddec[:, dw:]): d rr  |  decoder0.0 weight gradient
  gz.finish(gz.mma(s_ddec + DW, C_LD, wave, lane), s_red, nullptr, 0, s_drr, C_LR, first ? a.d_rr + rq * DW : nullptr, DW, d.Nq, wave, lane);
  wgrad16<DH, LDD, GR>(s_dd1, C_LH, s_dec, C_LD, sl + a.sl.dec_w[0], first ? sl + a.sl.dec_b[0] : nullptr, gwave, lane, tid);
  __syncthreads();
  MLHOT_TSTAMP(101);
  wgrad16<DZ, DW, GR>(s_ddec + DW, C_LD, s_rr, C_LR, sl + a.sl.r2z_w, first ? sl + a.sl.r2z_b : nullptr, gwave, lane, tid);
  // _W: only its bias gradient here (column sums of d rr); weight and input gradient run per head in phase B
  if (first && tid >= 256 && tid < 256 + DW) {
    float sum = 0.f;
#pragma unroll
    for (int row = 0; row < 16; ++row) sum += s_drr[row * C_LR + tid - 256];
    sl[a.sl.wo_b + tid - 256] = sum;
  }
  MLHOT_TSTAMP(102);
}

// ==================================================================================================
// phase A backward, one workgroup per task (see tf::phaseA_bwd_kernel): sums the 8 heads' input-gradient shares from
// phase B, applies the batch-global key arg-max correction, EncoderFC backward, transform_y weight gradient.
// ==================================================================================================
constexpr int AB_FLOATS = 16 * (2 * A_LCAT + 4 * A_LH + 2 * A_LX + A_LY) + NWV * 256 + 64 + 16;
__host__ inline size_t phaseA_bwd_lds_bytes() { return sizeof(float) * AB_FLOATS; }

// GR > 1 (round 5): GR workgroups per task, as phase C': all sum the heads' shares and walk EncoderFC's data gradients, the weight
// gradients' 28 + 49 + 35 tiles are dealt over the group; workgroup 0 alone applies the arg-max fix-up (a read-modify-write of slab
// entries), adds into d_dec_in and writes d_cat_in.  1 -> 2 -> 4 workgroups: 18.2 -> 15.9 -> 15.2 us.
template <int GR>
__global__ __launch_bounds__(512) void phaseA_bwd_kernel(const PhaseABwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int t = (int)blockIdx.x / GR, grp = (int)blockIdx.x % GR;          // GR workgroups per task
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), gwave = grp * NWV + wave;
  const bool first = grp == 0;
  lptr s_cat = L0;                  // saved activations
  lptr s_h0 = s_cat + 16 * A_LCAT;
  lptr s_h1 = s_h0 + 16 * A_LH;
  lptr s_y = s_h1 + 16 * A_LH;
  lptr s_drs = s_y + 16 * A_LY;     // gradients
  lptr s_dxc = s_drs + 16 * A_LX;
  lptr s_dh1 = s_dxc + 16 * A_LX;
  lptr s_dh0 = s_dh1 + 16 * A_LH;
  lptr s_dcat = s_dh0 + 16 * A_LH;
  lptr s_red = s_dcat + 16 * A_LCAT; // [8 waves][256] partial tiles
  lptr s_fix = s_red + NWV * 256;    // [64] correction vector, [16] wave partials of the key row-sum total
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  // ---- every global read of the block, up front (the arg-max position first: the fix-up's loads depend on it)
  const int grow = a.gpos[0], gcol = a.gpos[1];
  TileW<LDC> tcat; TileW<H0> th0, th1;
  MLHOT_TSTAMP(160);
#ifdef MLHOT_TS
  if (g_ts_dev && threadIdx.x == 0 && blockIdx.x < 16) g_ts_dev[200 + blockIdx.x] = wall_clock64();
#endif
  tcat.fetch(a.cat_in + rc * LDC, LDC, d.Nc, tid);
  th0.fetch(a.h0 + rc * H0, H0, d.Nc, tid);
  th1.fetch(a.h1 + rc * H1, H1, d.Nc, tid);
  float yv = 0.f;
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    if (r < d.Nc && c < d.label_dim) yv = a.ctx_y[(rc + r) * d.label_dim + c];
  }
  // the heads' shares (fixed order): d rs, the K-projection share of d x_ctx, the attention share of d x_qry;
  // thread = (row tid / 32, columns 2 (tid % 32) .. + 1)
  const int srow = tid >> 5, scol = 2 * (tid & 31);
  float2 v1[H], v2[H], v3[H], dq0 = make_float2(0.f, 0.f);
  if (srow < d.Nc) {
#pragma unroll
    for (int h = 0; h < H; ++h) {
      v1[h] = *reinterpret_cast<const float2*>(a.prs + ((size_t)(t * H + h) * d.Nc + srow) * DW + scol);
      v2[h] = *reinterpret_cast<const float2*>(a.pxc + ((size_t)(t * H + h) * d.Nc + srow) * DW + scol);
    }
  }
  if (srow < d.Nq) {
    dq0 = *reinterpret_cast<const float2*>(a.d_dec_in + (rq + srow) * LDD + scol);
#pragma unroll
    for (int h = 0; h < H; ++h) v3[h] = *reinterpret_cast<const float2*>(a.pxq + ((size_t)(t * H + h) * d.Nq + srow) * DW + scol);
  }
  float pv = 0.f;
  for (int i = tid; i < d.T * H; i += NWV * 64) pv += a.part_k[i];
  Dg<DW, H1> g2;  Dg<H1, H0> g1;  Dg<H0, LDC> g0;
  g2.load(a.p.er_w[2], wave, lane);
  g1.load(a.p.er_w[1], wave, lane);
  g0.load(a.p.er_w[0], wave, lane);
  // The batch-global key arg-max (see tf::phaseA_bwd_kernel): a rank-1 fix-up in the ONE task that holds it.  Its operands
  // (the [dw][dw] slab block it updates, W_k of that head, the projection row) are requested here with everything else.
  float* sl = a.slab + (size_t)t * a.sl.total;
  const bool fix = first && grow / (d.Nc * H) == t;           // block-uniform
  const int fn = (grow / H) % d.Nc, fh = grow % H;
  float gwv[8], wkv[8], pcv = 0.f, bkv = 0.f;
  if (fix) {
    const float* wk = a.p.wk_w[0];
#pragma unroll
    for (int i = 1; i < H; ++i)
      if (fh == i) wk = a.p.wk_w[i];
    const float* gw = sl + a.sl.wk_w + fh * DW * DW;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gwv[k] = gw[tid + NWV * 64 * k];
      wkv[k] = wk[(8 * (tid >> 6) + k) * DW + (tid & 63)];      // thread (part = tid / 64, column tid % 64): rows 8 part .. + 7
    }
    if (tid < DW) { pcv = a.pc[(size_t)gcol * DW + tid]; bkv = sl[a.sl.wk_b + fh * DW + tid]; }
  }
  // ---- stash
  tcat.stash(s_cat, A_LCAT, tid); th0.stash(s_h0, A_LH, tid); th1.stash(s_h1, A_LH, tid);
  if (tid < 256) s_y[(tid >> 4) * A_LY + (tid & 15)] = yv;
  {
    float2 s1 = make_float2(0.f, 0.f), s2 = make_float2(0.f, 0.f);
    if (srow < d.Nc) {
#pragma unroll
      for (int h = 0; h < H; ++h) { s1.x += v1[h].x; s1.y += v1[h].y; s2.x += v2[h].x; s2.y += v2[h].y; }
    }
    s_drs[srow * A_LX + scol] = s1.x; s_drs[srow * A_LX + scol + 1] = s1.y;
    s_dxc[srow * A_LX + scol] = s2.x; s_dxc[srow * A_LX + scol + 1] = s2.y;
    if (first && srow < d.Nq) {
      float2 s3 = make_float2(0.f, 0.f);
#pragma unroll
      for (int h = 0; h < H; ++h) { s3.x += v3[h].x; s3.y += v3[h].y; }
      *reinterpret_cast<float2*>(a.d_dec_in + (rq + srow) * LDD + scol) = make_float2(dq0.x + s3.x, dq0.y + s3.y);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pv += __shfl_xor(pv, off, 64);
  if (lane == 0) s_fix[64 + wave] = pv;
  __syncthreads();
  MLHOT_TSTAMP(161);
  // the fix-up: delta = -total * pc[col]; dW_k,fh += delta (x) x_ctx[fn], db_k,fh += delta, d x_ctx[fn] += delta . W_k,fh
  if (fix) {
    if (tid < DW) {
      float total = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) total += s_fix[64 + w];
      const float dl = -total * pcv;
      s_fix[tid] = dl;
      sl[a.sl.wk_b + fh * DW + tid] = bkv + dl;
    }
    __syncthreads();
    float* gw = sl + a.sl.wk_w + fh * DW * DW;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + NWV * 64 * k, e = i >> 6, c = i & 63;
      gw[i] = gwv[k] + s_fix[e] * s_cat[fn * A_LCAT + c];
      part += s_fix[8 * (tid >> 6) + k] * wkv[k];
    }
    s_red[tid] = part;
    __syncthreads();
    if (tid < DW) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w) acc += s_red[w * 64 + tid];
      s_dxc[fn * A_LX + tid] += acc;
    }
    __syncthreads();
  }
  MLHOT_TSTAMP(162);
  // EncoderFC, last layer first: d h1
  g2.finish(g2.mma(s_drs, A_LX, wave, lane), s_red, s_h1, A_LH, s_dh1, A_LH, nullptr, 0, 0, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(163);
  g1.finish(g1.mma(s_dh1, A_LH, wave, lane), s_red, s_h0, A_LH, s_dh0, A_LH, nullptr, 0, 0, wave, lane);
  wgrad16<DW, H1, GR>(s_drs, A_LX, s_h1, A_LH, sl + a.sl.er_w[2], first ? sl + a.sl.er_b[2] : nullptr, gwave, lane, tid);
  __syncthreads();
  MLHOT_TSTAMP(164);
  g0.finish(g0.mma(s_dh0, A_LH, wave, lane), s_red, nullptr, 0, s_dcat, A_LCAT, nullptr, 0, 0, wave, lane);
  wgrad16<H1, H0, GR>(s_dh1, A_LH, s_h0, A_LH, sl + a.sl.er_w[1], first ? sl + a.sl.er_b[1] : nullptr, gwave, lane, tid);
  __syncthreads();
  MLHOT_TSTAMP(165);
  // d_cat_in = EncoderFC input gradient (+ K-projection share on the x_ctx columns)
  for (int i = first ? tid : d.Nc * LDC; i < d.Nc * LDC; i += NWV * 64) {
    const int r = i / LDC, c = i - r * LDC;
    a.d_cat_in[(rc + r) * LDC + c] = s_dcat[r * A_LCAT + c] + (c < DW ? s_dxc[r * A_LX + c] : 0.f);
  }
  wgrad16<H0, LDC, GR>(s_dh0, A_LH, s_cat, A_LCAT, sl + a.sl.er_w[0], first ? sl + a.sl.er_b[0] : nullptr, gwave, lane, tid);
  // transform_y: dW = d_cat[:, dw:]^T ctx_y, db
  if (first) wgrad16<DW / 4, 4>(s_dcat + DW, A_LCAT, s_y, A_LY, sl + a.sl.ty_w, sl + a.sl.ty_b, wave, lane, tid, DW / 4, d.label_dim);
  MLHOT_TSTAMP(166);
#ifdef MLHOT_TS
  if (g_ts_dev && threadIdx.x == 0 && blockIdx.x < 16) g_ts_dev[220 + blockIdx.x] = wall_clock64();
#endif
}

// ==================================================================================================
// phase B backward, one workgroup per (task, head): FAVOR+ backward (S-form) and this head's W_q / W_k / W_v / _W backward
// (see tf::phaseB_bwd_kernel: same inputs, same outputs).  Differences: every weight fragment (the head's slice of _W, the
// projection matrix for the feature-map gradient, the three projection weights) is in registers before the first barrier;
// dS and dV run side by side; the row sums of G come out of the G tiles' accumulators instead of a pass of their own.
// ==================================================================================================
constexpr int BB_FLOATS = 16 * (12 * B_LX + 4 * B_LF) + 2 * 16 * 17 + 64 + NWV * 32 + 2 * NWV * 256;
__host__ inline size_t phaseB_bwd_lds_bytes() { return sizeof(float) * BB_FLOATS; }

// SPLIT (round 5, VERDICT r4 item 1 i): TWO workgroups per (task, head) - the query side and the key / value side of the backward
// are independent once dS is known, so side 0 takes G_q, d x_qry's share, W_q's and _W's weight gradients, side 1 takes dV, G_k,
// the key row-sum share, d x_ctx's / d rs's shares, W_k's and W_v's gradients; both compute the small common part (dO, dS) and
// load both feature tiles.  256 workgroups (one per CU at T = 16) instead of 128, no exchange between the two.
template <bool SPLIT>
__global__ __launch_bounds__(512) void phaseB_bwd_kernel(const PhaseBBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const TailDims& d = a.d;
  const int bth = SPLIT ? (int)blockIdx.x >> 1 : (int)blockIdx.x, side = SPLIT ? (int)blockIdx.x & 1 : 0;
  const bool doq = !SPLIT || side == 0, dok = !SPLIT || side == 1;          // block-uniform
  const int t = bth / H, h = bth % H, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  lptr s_q = L0;                    // [16][B_LX]
  lptr s_k = s_q + 16 * B_LX;
  lptr s_v = s_k + 16 * B_LX;
  lptr s_do = s_v + 16 * B_LX;       // dO
  lptr s_xq = s_do + 16 * B_LX;      // projection inputs ...
  lptr s_xc = s_xq + 16 * B_LX;
  lptr s_rs = s_xc + 16 * B_LX;
  lptr s_dq = s_rs + 16 * B_LX;      // ... and head-space gradients
  lptr s_dk = s_dq + 16 * B_LX;
  lptr s_dv = s_dk + 16 * B_LX;
  lptr s_o = s_dv + 16 * B_LX;       // this head's attention output O (from merged) and d rr of the task (from phase C)
  lptr s_drr = s_o + 16 * B_LX;
  lptr s_qf = s_drr + 16 * B_LX;     // [16][B_LF] E features
  lptr s_kf = s_qf + 16 * B_LF;
  lptr s_gq = s_kf + 16 * B_LF;      // G, then d(dd)
  lptr s_gk = s_gq + 16 * B_LF;
  lptr s_S = s_gk + 16 * B_LF;       // [16][17]  S / D
  lptr s_dS = s_S + 16 * 17;         // [16][17]
  lptr s_st = s_dS + 16 * 17;        // wv[16], D[16], rsum_q[16], rsum_k[16]
  lptr s_part = s_st + 64;           // [8 waves][32] partial row sums of G
  lptr s_red = s_part + NWV * 32;    // 2 x [8 waves][256] partial tiles
  const size_t rq = (size_t)t * d.Nq, rc = (size_t)t * d.Nc;
  // ---- every global read of the block, up front
  Tile64 tq, tk, tv, txq, txc, trs, tdrr;
  MLHOT_TSTAMP(128);
  tq.v = tk.v = txq.v = txc.v = trs.v = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (doq) tq.fetch(a.qh + rq * HD + h * DW, HD, d.Nq, tid);
  if (dok) tk.fetch(a.kh + rc * HD + h * DW, HD, d.Nc, tid);
  tv.fetch(a.vh + rc * HD + h * DW, HD, d.Nc, tid);
  if (doq) txq.fetch(a.dec_in + rq * LDD, LDD, d.Nq, tid);
  if (dok) txc.fetch(a.cat_in + rc * LDC, LDC, d.Nc, tid);
  if (dok) trs.fetch(a.rs + rc * DW, DW, d.Nc, tid);
  tdrr.fetch(a.d_rr + rq * DW, DW, d.Nq, tid);
  float ov[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {                       // O[n][e] = merged[(t,n)][e*H + h]
    const int idx = tid + 512 * k, n = idx >> 6, e = idx & 63;
    ov[k] = n < d.Nq ? a.merged[(rq + n) * HD + e * H + h] : 0.f;
  }
  constexpr int F2 = M / 2;                           // 133 float2 per feature row
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t fq[5], fk[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int idx = tid + 512 * k, row = idx / F2, c2 = idx - row * F2;
    fq[k] = f32x2_t{0.f, 0.f}; fk[k] = f32x2_t{0.f, 0.f};
    if (row < d.Nq) fq[k] = *reinterpret_cast<const f32x2_t*>(a.qf + ((rq + row) * H + h) * M + 2 * c2);
    if (row < d.Nc) fk[k] = *reinterpret_cast<const f32x2_t*>(a.kf + ((rc + row) * H + h) * M + 2 * c2);
  }
  float sdv = 0.f, dval = 1.f; int argq = 0;
  if (tid < 256) {
    const int n = tid >> 4, np = tid & 15;
    if (n < d.Nq) dval = a.D[((size_t)t * H + h) * d.Nq + n];
    if (n < d.Nq && np < d.Nc) sdv = a.S[(((size_t)t * H + h) * d.Nq + n) * d.Nc + np];
  }
  if (doq && tid < d.Nq) argq = a.arg_q[(t * d.Nq + tid) * H + h];
  const float *wq = a.p.wq_w[0], *wk = a.p.wk_w[0], *wv = a.p.wv_w[0];
#pragma unroll
  for (int i = 1; i < H; ++i)
    if (h == i) { wq = a.p.wq_w[i]; wk = a.p.wk_w[i]; wv = a.p.wv_w[i]; }
  Dg<DW, DW> g_o;                                     // dO = d rr . Wo_h
  g_o.load(a.wot + (size_t)h * DW * DW, wave, lane);
  Dg<M, DW> g_p;                                      // dx = d(dd) . pc   (shared by the query and the key rows)
  g_p.load(a.pc, wave, lane);
  // input-gradient shares P = dY W_h: items (projection pj, 16-column tile): wave w takes item w, waves 0-3 also item 8 + w
  float wsh[2][4][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    // SPLIT: the query side's four items on its waves 0-3, the key / value side's eight on its waves 0-7, one each
    const int item = SPLIT ? (u == 0 ? (side == 0 ? (wave < 4 ? wave : 99) : 4 + wave) : 99) : wave + 8 * u;
    if (item < 12) {
      const int pj = item >> 2, i0 = (item & 3) * 16;
      const float* wsel = (pj == 0 ? wq : pj == 1 ? wk : wv) + i0 + lr;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int e = 0; e < 4; ++e) wsh[u][jb][e] = wsel[(jb * 16 + 4 * lq + e) * DW];
    }
  }
  // ---- stash
  tq.stash(s_q, B_LX, tid); tk.stash(s_k, B_LX, tid); tv.stash(s_v, B_LX, tid);
  txq.stash(s_xq, B_LX, tid); txc.stash(s_xc, B_LX, tid); trs.stash(s_rs, B_LX, tid); tdrr.stash(s_drr, B_LX, tid);
#pragma unroll
  for (int k = 0; k < 2; ++k) { const int idx = tid + 512 * k; s_o[(idx >> 6) * B_LX + (idx & 63)] = ov[k]; }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const int idx = tid + 512 * k, row = idx / F2, c2 = idx - row * F2;
    if (row < 16) {
      *reinterpret_cast<MLHOT_LDS f32x2_t*>(s_qf + row * B_LF + 2 * c2) = fq[k];
      *reinterpret_cast<MLHOT_LDS f32x2_t*>(s_kf + row * B_LF + 2 * c2) = fk[k];
    }
  }
  if (tid < 256) s_S[(tid >> 4) * 17 + (tid & 15)] = sdv / dval;          // S / D (zero outside the valid block)
  if (tid < 256 && (tid & 15) == 0) s_st[16 + (tid >> 4)] = dval;
  __syncthreads();
  MLHOT_TSTAMP(129);
  // _W's input gradient for this head: dO[n][e] = sum_j d rr[n][j] Wo[j][e*H + h]; rows >= Nq of d rr are zero
  g_o.finish(g_o.mma(s_drr, B_LX, wave, lane), s_red, nullptr, 0, s_do, B_LX, nullptr, 0, 0, wave, lane);
  __syncthreads();
  MLHOT_TSTAMP(130);
  // wv[n] = dO[n] . (O[n] - c), c = v[0]: the common centre of the two inner products of dS below  (16 threads per row)
  if (tid < 256) {
    const int n = tid >> 4, part = tid & 15;
    const f32x4_t x = *reinterpret_cast<lc4ptr>(s_do + n * B_LX + 4 * part), y = *reinterpret_cast<lc4ptr>(s_o + n * B_LX + 4 * part);
    const f32x4_t cz = *reinterpret_cast<lc4ptr>(s_v + 4 * part);
    float sum = x[0] * (y[0] - cz[0]) + x[1] * (y[1] - cz[1]) + x[2] * (y[2] - cz[2]) + x[3] * (y[3] - cz[3]);
    sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 8, 64);
    if (part == 0) s_st[n] = sum;
  }
  // dV[n'][e] = sum_n (S/D)[n][n'] dO[n][e]  (waves 4-7, one e tile each)
  if (wave >= 4 && dok) {
    const int et = wave - 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int n = 4 * s4 + lq;
      acc = mfma4(s_S[n * 17 + lr], s_do[n * B_LX + et * 16 + lr], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int np = 4 * lq + r;
      s_dv[np * B_LX + et * 16 + lr] = np < d.Nc ? acc[r] : 0.f;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(131);
  // dS[n][n'] = (dO[n] . (v[n'] - c) - wv[n]) / D[n]   (wave 0), valid entries only.  O[n] is a convex combination of the value rows:
  // taken against the common centre c = v[0] the two inner products no longer share the value rows' common component, whose fp32
  // rounding would otherwise dominate their difference (favor2.h, B1; measured on trained-scale features: 5e-4 -> 1e-5 of dk)
  if (wave == 0) {
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e0 = 0; e0 < DW; e0 += 8) {
      a0 = mfma4(s_do[lr * B_LX + e0 + lq], s_v[lr * B_LX + e0 + lq] - s_v[e0 + lq], a0);
      a1 = mfma4(s_do[lr * B_LX + e0 + 4 + lq], s_v[lr * B_LX + e0 + 4 + lq] - s_v[e0 + 4 + lq], a1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * lq + r, np = lr;
      s_dS[n * 17 + np] = (n < d.Nq && np < d.Nc) ? (a0[r] + a1[r] - s_st[n]) / s_st[16 + n] : 0.f;
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(132);
  // G = dF (.) E with dQ' = dS (Ek + re), dK' = dS^T (Eq + re): 2 x 17 feature tiles over the waves; the row sums of G
  // ride along (this lane: rows 4 lq + r of its tiles' column)
  const float ratio = 1.0f / sqrtf((float)M), re = ratio * 1e-4f;
  constexpr int NTILE = (M + 15) / 16;
  {
    float rsq[4] = {0.f, 0.f, 0.f, 0.f}, rsk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tr = 0; tr < (SPLIT ? 3 : 5); ++tr) {
      const int it = wave + 8 * tr;
      if (it < (SPLIT ? NTILE : 2 * NTILE)) {
        const bool isk = SPLIT ? side == 1 : it >= NTILE;
        const int jt = (!SPLIT && isk) ? it - NTILE : it;
        const int j = jt * 16 + lr;
        const bool vj = j < M;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int o = 4 * s4 + lq;                    // summed row index (n' for queries, n for keys)
          const float av = isk ? s_dS[o * 17 + lr] : s_dS[lr * 17 + o];
          const float bvv = vj ? (isk ? s_qf[o * B_LF + j] : s_kf[o * B_LF + j]) + re : 0.f;
          acc = mfma4(av, bvv, acc);
        }
        lptr g = isk ? s_gk : s_gq;
        lcptr f = isk ? s_kf : s_qf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          const float gv = vj ? acc[r] * f[row * B_LF + j] : 0.f;
          g[row * B_LF + j] = gv;                       // padding columns 266..271 become exact zeros
          if (isk) rsk[r] += gv; else rsq[r] += gv;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { rsq[r] += __shfl_xor(rsq[r], off, 64); rsk[r] += __shfl_xor(rsk[r], off, 64); }
      if (lr == 0) { s_part[wave * 32 + 4 * lq + r] = rsq[r]; s_part[wave * 32 + 16 + 4 * lq + r] = rsk[r]; }
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(133);
  // row sums (32 rows: 16 query + 16 key), then d(dd): queries subtract the row sum at the arg-max
  if (tid < 32) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) sum += s_part[w * 32 + tid];
    s_st[32 + tid] = sum;
    if (doq && tid < d.Nq) s_gq[tid * B_LF + argq] -= sum;
    float ks = (tid >= 16 && tid - 16 < d.Nc) ? sum : 0.f;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) ks += __shfl_xor(ks, off, 64);
    if (tid == 0 && dok) a.part_k[t * H + h] = ks;
  }
  __syncthreads();
  MLHOT_TSTAMP(134);
  // dx[row][e] = sum_j d(dd)[row][j] pc[j][e] - rsum[row] c^2 x[row][e], query and key rows on the same pc fragments;
  // the two halves of the j range are folded through LDS
  {
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t aq = doq ? g_p.mma(s_gq, B_LF, wave, lane) : zero4, ak = dok ? g_p.mma(s_gk, B_LF, wave, lane) : zero4;
    constexpr int NI = Dg<M, DW>::NI;                   // 4 tiles x 2 chunks
    const int tile = wave % NI, chunk = wave / NI;
    if (chunk > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { s_red[(wave * 4 + r) * 64 + lane] = aq[r]; s_red[NWV * 256 + (wave * 4 + r) * 64 + lane] = ak[r]; }
    }
    __syncthreads();
    if (chunk == 0) {
      const float c2 = 1.0f / sqrtf((float)DW);
      const int e = tile * 16 + lr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * lq + r;
        const float vq = aq[r] + s_red[((wave + NI) * 4 + r) * 64 + lane], vk = ak[r] + s_red[NWV * 256 + ((wave + NI) * 4 + r) * 64 + lane];
        if (doq) s_dq[row * B_LX + e] = row < d.Nq ? vq - s_st[32 + row] * c2 * s_q[row * B_LX + e] : 0.f;
        if (dok) s_dk[row * B_LX + e] = row < d.Nc ? vk - s_st[48 + row] * c2 * s_k[row * B_LX + e] : 0.f;
      }
    }
  }
  __syncthreads();
  MLHOT_TSTAMP(135);
  // ---- this head's W_q / W_k / W_v backward and _W's weight gradient -----------------------------------
  {
    float* sl = a.slab + (size_t)t * a.sl.total;
    constexpr int nt = DW / 16, ww = DW * DW;
    // input-gradient shares first (their weights have been in registers since the prologue)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int item = SPLIT ? (u == 0 ? (side == 0 ? (wave < 4 ? wave : 99) : 4 + wave) : 99) : wave + 8 * u;
      if (item < 12) {
        const int pj = item >> 2, i0 = (item & 3) * 16;
        lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const f32x4_t x = *reinterpret_cast<lc4ptr>(dy + lr * B_LX + jb * 16 + 4 * lq);
          if (jb & 1) { a1 = mfma4(x[0], wsh[u][jb][0], a1); a1 = mfma4(x[1], wsh[u][jb][1], a1); a1 = mfma4(x[2], wsh[u][jb][2], a1); a1 = mfma4(x[3], wsh[u][jb][3], a1); }
          else { a0 = mfma4(x[0], wsh[u][jb][0], a0); a0 = mfma4(x[1], wsh[u][jb][1], a0); a0 = mfma4(x[2], wsh[u][jb][2], a0); a0 = mfma4(x[3], wsh[u][jb][3], a0); }
        }
        float* dst = pj == 0 ? a.pxq : pj == 1 ? a.pxc : a.prs;
        const int nrows = pj == 0 ? d.Nq : d.Nc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 4 * lq + r;
          if (row < nrows) dst[((size_t)(t * H + h) * nrows + row) * DW + i0 + lr] = a0[r] + a1[r];
        }
      }
    }
    // weight gradients dW[n][i] = sum_row dY[row][n] X[row][i]: 3 nt^2 tiles of 4 MFMAs over the waves
#pragma unroll
    for (int tr = 0; tr < (SPLIT ? 4 : 3 * nt * nt / NWV); ++tr) {
      // SPLIT: the query side's 16 tiles (W_q) in two trips, the key / value side's 32 (W_k, W_v) in four
      if (SPLIT && side == 0 && tr >= 2) break;
      const int it = (SPLIT && side == 1 ? nt * nt : 0) + wave + NWV * tr;
      const int pj = it / (nt * nt), rem = it - pj * nt * nt, j0 = (rem / nt) * 16, i0 = (rem % nt) * 16;
      lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
      lcptr x = pj == 0 ? s_xq : pj == 1 ? s_xc : s_rs;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = mfma4(dy[(4 * s4 + lq) * B_LX + j0 + lr], x[(4 * s4 + lq) * B_LX + i0 + lr], acc);
      float* dst = sl + (pj == 0 ? a.sl.wq_w : pj == 1 ? a.sl.wk_w : a.sl.wv_w) + h * ww;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(j0 + 4 * lq + r) * DW + i0 + lr] = acc[r];
    }
    // _W's weight gradient for this head's columns: dWo[j][e*H + h] = sum_n d rr[n][j] O[n][e]
#pragma unroll
    for (int tr = 0; tr < nt * nt / NWV; ++tr) {
      if (!doq) break;                                  // SPLIT: _W's weight gradient rides with the (lighter) query side
      const int it = wave + NWV * tr;
      const int j0 = (it / nt) * 16, e0 = (it % nt) * 16;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = mfma4(s_drr[(4 * s4 + lq) * B_LX + j0 + lr], s_o[(4 * s4 + lq) * B_LX + e0 + lr], acc);
      float* dst = sl + a.sl.wo_w;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(size_t)(j0 + 4 * lq + r) * HD + (e0 + lr) * H + h] = acc[r];
    }
    // bias gradients: column sums
    if (tid < 3 * DW && (tid < DW ? doq : dok)) {
      const int pj = tid / DW, n = tid - pj * DW;
      lcptr dy = pj == 0 ? s_dq : pj == 1 ? s_dk : s_dv;
      float sum = 0.f;
#pragma unroll
      for (int row = 0; row < 16; ++row) sum += dy[row * B_LX + n];
      sl[(pj == 0 ? a.sl.wq_b : pj == 1 ? a.sl.wk_b : a.sl.wv_b) + h * DW + n] = sum;
    }
  }
  MLHOT_TSTAMP(136);
}

}  // namespace ts
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
