// conv3 of the vanilla image encoder (48 -> 64 channels, 3x3, stride 2, pad 1, 16x16 -> 8x8, + ReLU)
// as three weight-stationary MFMA kernels, the same scheme as conv2's (conv_tc.h):
// one persistent workgroup per CU keeps its share of the [64][432] weight matrix in registers, images
// stream through LDS, and the matrix core runs v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate).
//   forward : unit = half an image (4 output rows); double-buffered [48][9][24] input patch
//   wgrad   : unit = image; accumulators = the whole 432 x 64 gradient (18 tiles per wave), position
//             halves on two wave groups folded through LDS at the end, one slab per workgroup
//   dgrad   : unit = image; one wave per (parity class, 16-channel tile): a class only carries the
//             taps that reach it (1, 2, 2 or 4 of the 9), waves are placed so every SIMD gets ~equal work
// All LDS strides are chosen so the 32 lanes of a half-wave hit 32 different banks (comments inline).
// GPU build only.  Reference: networks/conv_embedding_model.py-style encoder used by
// CondNeuralProcess.py / ANP*.py (`encoder_w0`), third conv block.
#pragma once
#include "common.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace c3 {
#ifdef MLHOT_TS
#define C3_TS(slot) do { if (tf::g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0) tf::g_ts_dev[340 + (slot)] = clock64(); } while (0)
#else
#define C3_TS(slot) do { } while (0)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int KW_ = 432;
// conv3's [64][432] weight matrix -> LDS rows of 433 words, coalesced float4 loads: ALL of a thread's loads first, then the stores.
// (As one load -> store loop with a run-time trip count hipcc left it rolled, and each of the 14 iterations waited out an L2 round
// trip: 9 k of the forward's 54 k cycles per workgroup, stamps.)
template <int NTHREADS>
__device__ __forceinline__ void conv3w_stage(float* stage, const float* __restrict__ w, int tid) {
  constexpr int N4 = 64 * KW_ / 4, CNT = (N4 + NTHREADS - 1) / NTHREADS;
  float4 v[CNT];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int i = tid + j * NTHREADS;
    v[j] = i < N4 ? *reinterpret_cast<const float4*>(w + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int i = tid + j * NTHREADS;
    if (i < N4) {
      const int r = (4 * i) / KW_, c = (4 * i) - r * KW_;       // KW % 4 == 0: a float4 stays inside its row
      float* d = stage + r * (KW_ + 1) + c;
      d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
    }
  }
}
// The forward's variant: rows re-ordered [co][tap][ci 48 + 4] (tap stride TS_ = 52, row stride S_ = 472 words) so that the four
// weights of a lane's k-steps (tap, g, j = 0..3) - input channels 16 g + 4 lq + j - are ONE ds_read_b128: 27 gather reads per
// lane instead of 108 ds_read_b32 that were 2-way conflicted (a row of 433 words put lr and lr + 4 of the other lq on one bank),
// (tests/test_index_maps.py restates the bank counts).
template <int NTHREADS, int S_, int TS_>
__device__ __forceinline__ void conv3w_stage_tap_major(float* stage, const float* __restrict__ w, int tid) {
  // item = (row co, input channel ci): its 9 taps are 9 consecutive words of the [64][48][9] matrix (36 bytes - dword loads, the
  // lanes of a wave cover consecutive channels: whole cache lines), stored at ONE lane address + compile-time tap steps; the lanes'
  // words are consecutive channels: no bank conflicts inside a row.  (As float4 items the four words of an item were four
  // (ci, tap) pairs with a wrap in between: a compare, two selects and a multiply-add per store.)
  constexpr int NI = 64 * 48, CNT = (NI + NTHREADS - 1) / NTHREADS;
  float v[CNT][9];
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int i = tid + j * NTHREADS;
#pragma unroll
    for (int t = 0; t < 9; ++t) v[j][t] = i < NI ? w[9 * i + t] : 0.f;
  }
#ifdef MLHOT_TS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C3_TS(14);
#endif
#pragma unroll
  for (int j = 0; j < CNT; ++j) {
    const int i = tid + j * NTHREADS;
    if (i < NI) {
      const int r = i / 48, ci = i - 48 * r;
      float* d = stage + r * S_ + ci;                // word of W[co][ci][tap] = co * S_ + tap * TS_ + ci
#pragma unroll
      for (int t = 0; t < 9; ++t) d[t * TS_] = v[j][t];
    }
  }
}
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int CIN = 48, COUT = 64, KW = 432;
#ifndef C3_RD
#define C3_RD 4
#endif

// ---- forward ---------------------------------------------------------------------------------------
// 512 threads: wave = (nt = wave & 3: output channels 16nt..+15) x (mg = wave >> 2: output rows 2mg, 2mg+1
// of the unit's 4).  Lane (lr, lq) of an MFMA: A = input[ci][2oy + ky - 1][2ox + kx - 1] of position
// lr = (oy & 1) * 8 + ox, B = W[16nt + lr][ci][tap] from registers (108 per lane).
// Patch [r = iy - (8hf - 1)][c = ix + 1][ci 48 + 4]: the CHANNEL is innermost, and k-step (tap, g, j) takes ci = 16 g + 4 lq + j,
// so ONE ds_read_b128 at lane base + a compile-time offset brings the lane's A operands of four k-steps (as [ci] planes it was one
// ds_read_b32 per MFMA: 1.36 LDS instructions per MFMA on the port the fp32 MFMAs issue through).  Banks: a position is 52 words,
// the tile's columns are 2 positions = 104 = 40 (mod 64) words apart, its second row 2 x 24 x 52 = 0 (mod 64), lq adds 4: the 16
// lanes of every ds_read_b128 group cover the 64 banks once (checked below).
constexpr int F_RS = 24, F_CS = 52, F_ROW = F_RS * F_CS, F_PATCH = 9 * F_ROW;       // 11,232 floats = 44.9 KB per buffer
constexpr int F_NT = 512;
constexpr int F_WTS = 52, F_WLD = 472;              // weight staging [co][tap][ci 48 + 4], rows of 472 words (conv3w_stage_tap_major)
constexpr int F_LDS = 2 * F_PATCH > COUT * F_WLD ? 2 * F_PATCH : COUT * F_WLD;
constexpr bool f_b128_conflict_free() {
  // lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS): {0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32
  const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
  for (int half = 0; half < 2; ++half)
    for (int g = 0; g < 2; ++g) {
      unsigned long long seen = 0;
      for (int i = 0; i < 16; ++i) {
        const int lane = grp[g][i] + 32 * half, lr = lane & 15, lq = lane >> 4;
        const int word = ((2 * (lr >> 3)) * F_RS + 2 * (lr & 7)) * F_CS + 4 * lq;
        for (int q = 0; q < 4; ++q) {
          const unsigned long long bit = 1ull << ((word + q) & 63);
          if (seen & bit) return false;
          seen |= bit;
        }
      }
    }
  return true;
}
static_assert(f_b128_conflict_free(), "conv3 forward: operand reads have bank conflicts");

__global__ __launch_bounds__(F_NT) void conv3_fwd_kernel(const float* __restrict__ p2, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ a3, int n_img) {
  __shared__ __attribute__((aligned(16))) float patch2[F_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave & 3, mg = wave >> 2;
  const int lr = lane & 15, lq = lane >> 4;
  const int co = nt * 16 + lr;
  C3_TS(0);

  const int nunits = n_img * 2;
  // staging: 48 ci x 9 rows x 4 column quads of float4 per unit.  Thread = (quad c4 = tid % 4, channel ci = 8 (tid / 32 % 8) +
  // tid / 4 % 8, row phase tid / 256): its five items are consecutive rows 5 phase + j - one lane base and compile-time steps for
  // the requests and the transposing stores alike (threads with tid / 32 % 8 >= 6 idle: 48 = 6 x 8 channels).  A 32-lane half
  // holds 4 quads x 8 channels of ONE row: its stores (a quad = 4 positions = 208 = 16 mod 32 words, a channel 1 word) are 2-way
  // at worst.  (Row-major items - a channel's 9 rows are 576 contiguous bytes - put 8 rows, 24 x 52 = 0 mod 32 words apart, into
  // every half: 16-way stores, the kernel 2.4 us slower; items dealt as e = tid + 512 j cost two divisions per item and use.)
  const int s_c4 = tid & 3, s_ci = 8 * ((tid >> 5) & 7) + ((tid >> 2) & 7), s_r0 = 5 * (tid >> 8);
  const bool s_on = s_ci < CIN;
  float4 st[5];
  auto fetch = [&](int u) {
    const int img = u >> 1, hf = u & 1;
    const float* gl = p2 + (((size_t)img * CIN + (s_on ? s_ci : 0)) * 16 + 8 * hf - 1 + s_r0) * 16 + 4 * s_c4;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int r = s_r0 + j, iy = 8 * hf - 1 + r;
      st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s_on && r < 9 && iy >= 0) st[j] = *reinterpret_cast<const float4*>(gl + 16 * j);
    }
  };
  auto stash = [&](float* buf) {
    float* dl = buf + (s_r0 * F_RS + 1 + 4 * s_c4) * F_CS + s_ci;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (s_on && s_r0 + j < 9) {
        float* d = dl + j * F_ROW;
        d[0] = st[j].x; d[F_CS] = st[j].y; d[2 * F_CS] = st[j].z; d[3 * F_CS] = st[j].w;
      }
    }
  };
  int unit = blockIdx.x;
  if (unit < nunits) fetch(unit);        // the first unit's HBM round trip runs under the weight staging

  // The register-resident weights are a stride-9 gather of the [64][432] matrix: read straight from global every
  // wave load touches ~32 cache lines for 256 useful bytes.  Stage the matrix through LDS with coalesced loads, tap-major
  // (conv3w_stage_tap_major), then gather 27 x ds_read_b128 per lane.  Stamps of workgroup 0, cycles from entry, with the
  // [co][433] staging of conv3w_stage before: loads landed 6.0 k, stored + barrier 9.5 k (4-way conflicted ds_write_b32), gathered
  // 13.3 k (108 ds_read_b32, 2-way), exit 52.2 k; now 6.8 k / 9.2 k / 10.4 k, exit 49.9 k.
  conv3w_stage_tap_major<F_NT, F_WLD, F_WTS>(patch2, w, tid);
  __syncthreads();
  C3_TS(13);
  float wr[108];           // k-step ks = (tap, g, j) = 12 tap + 4 g + j: ci = 16 g + 4 lq + j
#pragma unroll
  for (int qd = 0; qd < 27; ++qd) {      // quad (tap, g): one ds_read_b128
    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(patch2 + co * F_WLD + (qd / 3) * F_WTS + 16 * (qd % 3) + 4 * lq);
    wr[4 * qd] = v[0]; wr[4 * qd + 1] = v[1]; wr[4 * qd + 2] = v[2]; wr[4 * qd + 3] = v[3];
  }
  const float bn = bias[co];
  __syncthreads();
  C3_TS(1);

  // halo column 0 of every patch row stays zero for good; nothing else of the two patches is ever read unwritten (columns 1..16
  // come from stash - zeros for the row above the image - and 17..23 / the four pad channels are never read)
  for (int i = tid; i < 2 * 9 * F_CS; i += F_NT) patch2[(i / (9 * F_CS)) * F_PATCH + ((i / F_CS) % 9) * F_ROW + i % F_CS] = 0.f;
  __syncthreads();
  C3_TS(15);
  if (unit < nunits) stash(patch2);
  C3_TS(16);
  if (unit + (int)gridDim.x < nunits) fetch(unit + gridDim.x);
  __syncthreads();
  const int aoff = ((2 * (2 * mg + (lr >> 3))) * F_RS + 2 * (lr & 7)) * F_CS + 4 * lq;
  int cur = 0;
  C3_TS(2);
#ifdef MLHOT_TS
  int ts_u = 0;
#endif
  for (; unit < nunits; unit += gridDim.x, cur ^= 1) {
    const float* ab = patch2 + cur * F_PATCH + aoff;
    const int next = unit + (int)gridDim.x;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // operand quad q = (tap, g) = k-steps 4 q .. 4 q + 3: one ds_read_b128, C3_RD quads ahead of its MFMAs (left alone hipcc reads
    // each operand right in front of its MFMA and waits lgkmcnt(0) on it)
    auto aread = [&](int q) { const int t = q / 3, g = q % 3; return *reinterpret_cast<const f32x4_t*>(ab + ((t / 3) * F_RS + t % 3) * F_CS + 16 * g); };
    constexpr int RDQ = 2;
    f32x4_t xa[RDQ];
#pragma unroll
    for (int d = 0; d < RDQ; ++d) xa[d] = aread(d);
#pragma unroll
    for (int ks = 0; ks < 108; ++ks) {
      if (ks == 54 && next < nunits) {             // the other buffer was released by the barrier that ended the previous unit
        stash(patch2 + (cur ^ 1) * F_PATCH);
        if (next + (int)gridDim.x < nunits) fetch(next + gridDim.x);
      }
      const float x = xa[(ks >> 2) % RDQ][ks & 3];
      if (ks & 1) acc1 = mfma4(x, wr[ks], acc1); else acc0 = mfma4(x, wr[ks], acc0);
      if ((ks & 3) == 3 && (ks >> 2) + RDQ < 27) xa[(ks >> 2) % RDQ] = aread((ks >> 2) + RDQ);
      if (ks % 2 == 1) __builtin_amdgcn_sched_barrier(0);
    }
    // lane holds positions 4lq..4lq+3 of the tile = row (lq >> 1), columns 4(lq & 1)..+3 of channel co
    const int img = unit >> 1, hf = unit & 1;
    const int oy = 4 * hf + 2 * mg + (lq >> 1);
    float4 o;
    o.x = fmaxf(acc0[0] + acc1[0] + bn, 0.f); o.y = fmaxf(acc0[1] + acc1[1] + bn, 0.f);
    o.z = fmaxf(acc0[2] + acc1[2] + bn, 0.f); o.w = fmaxf(acc0[3] + acc1[3] + bn, 0.f);
    *reinterpret_cast<float4*>(a3 + (((size_t)img * COUT + co) * 8 + oy) * 8 + 4 * (lq & 1)) = o;
#ifdef MLHOT_TS
    C3_TS(3 + 2 * ts_u);
#endif
    __syncthreads();
#ifdef MLHOT_TS
    C3_TS(4 + 2 * ts_u); ++ts_u;
#endif
  }
  C3_TS(12);
}

// ---- weight + bias gradient -----------------------------------------------------------------------
// 768 threads: wave = (job = wave % 6) x (ph = wave / 6: output rows 4ph..4ph+3); job = (tg = job % 3: taps
// ky = tg, kx = 0..2) x (np = job / 3: output-channel tiles 2np, 2np+1).  18 accumulator tiles per wave:
// dW[ci = 16cig + .][tap (tg, kx)] x [co tile].  One k-step = 4 positions (oy = 4ph + lq, ox = ks):
// A = patch[ci = 16cig + lr][2oy + ky][2ox + kx], B = dY^T[pos][co].
// Patch [ci][r = iy + 1][c = ix + 1]: row stride 24, plane stride 409 (odd); dY^T [pos][66] (8 * 66 = 16 mod 32).
constexpr int W_RS = 24, W_PS = 17 * W_RS + 1, W_PATCH = CIN * W_PS, W_DS = 66, W_DYT = 64 * W_DS;
constexpr int W_NT = 768;
constexpr int W_LDS = (W_PATCH + W_DYT) > 6 * 18 * 256 ? (W_PATCH + W_DYT) : 6 * 18 * 256;

// wg / nwg: this workgroup's index among the nwg that share the images (the kernel below: blockIdx / gridDim; the merged backward
// launch: its weight-gradient half)
__device__ __forceinline__ void conv3_wgrad_body(const float* __restrict__ p2, const float* __restrict__ dy3,
                                                 float* __restrict__ slab_w, float* __restrict__ slab_b, int n_img, float* lds, int wg, int nwg) {
  float* patch = lds;
  float* dyt = lds + W_PATCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int job = wave % 6, ph = wave / 6, tg = job % 3, np = job / 3;
  const int lr = lane & 15, lq = lane >> 4;

  f32x4_t acc[3][3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};

  for (int i = tid; i < W_LDS; i += W_NT) lds[i] = 0.f;          // halo row 0 / column 0 stay zero
  float4 sx[4], sd[2];
  auto fetch = [&](int img) {
#pragma unroll
    for (int j = 0; j < 4; ++j) sx[j] = *reinterpret_cast<const float4*>(p2 + (size_t)img * (CIN * 256) + 4 * (tid + j * W_NT));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * W_NT;
      sd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 1024) sd[j] = *reinterpret_cast<const float4*>(dy3 + (size_t)img * 4096 + 4 * e);
    }
  };
  int img = wg;
  if (img < n_img) fetch(img);
  const int aoff = lr * W_PS + (2 * (4 * ph + lq) + tg) * W_RS;
  const int boff = (8 * (4 * ph + lq)) * W_DS + 32 * np + lr;
  for (; img < n_img; img += nwg) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // image: float4 e -> ci = e / 64, iy = (e / 4) % 16, ix = 4 (e % 4)
      const int e = tid + j * W_NT;
      float* d = patch + (e >> 6) * W_PS + (((e >> 2) & 15) + 1) * W_RS + 1 + 4 * (e & 3);
      d[0] = sx[j].x; d[1] = sx[j].y; d[2] = sx[j].z; d[3] = sx[j].w;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                 // dY: float4 e -> co = e / 16, pos = 4 (e % 16) .. +3  (transposed into [pos][co])
      const int e = tid + j * W_NT;
      if (e < 1024) {
        float* d = dyt + (4 * (e & 15)) * W_DS + (e >> 4);
        d[0] = sd[j].x; d[W_DS] = sd[j].y; d[2 * W_DS] = sd[j].z; d[3 * W_DS] = sd[j].w;
        bsum[j] += (sd[j].x + sd[j].y) + (sd[j].z + sd[j].w);
      }
    }
    __syncthreads();
    if (img + nwg < n_img) fetch(img + nwg);
    // (double-buffering the 11 operands of a k-step across k-steps measured equal here: 35.7 vs 35.4 us)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float a[3][3], b[2];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int cig = 0; cig < 3; ++cig) a[kx][cig] = patch[aoff + cig * 16 * W_PS + 2 * ks + kx];
#pragma unroll
      for (int c = 0; c < 2; ++c) b[c] = dyt[boff + ks * W_DS + 16 * c];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int cig = 0; cig < 3; ++cig)
#pragma unroll
          for (int c = 0; c < 2; ++c) acc[kx][cig][c] = mfma4(a[kx][cig], b[c], acc[kx][cig][c]);
    }
  }
  // fold the two position halves through LDS, then one slab per workgroup
  __syncthreads();
  if (ph == 1) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int cig = 0; cig < 3; ++cig)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[((job * 18 + (kx * 3 + cig) * 2 + c) * 4 + r) * 64 + lane] = acc[kx][cig][c][r];
  }
  __syncthreads();
  if (ph == 0) {
    float* sw = slab_w + (size_t)wg * (COUT * KW + COUT);      // slab row = [weights | bias]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int cig = 0; cig < 3; ++cig)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          // accumulator order [tile = job * 18 + (kx * 3 + cig) * 2 + c][lane][r]: one coalesced 16-byte store per tile and lane
          // (as 4-byte scatters into [co][ci][tap] order these 72 stores per lane were a sixth of the kernel); the final fold
          // un-permutes (c2::SumParts kind 2)
          const int tile = job * 18 + (kx * 3 + cig) * 2 + c;
          float4 v;
          v.x = acc[kx][cig][c][0] + lds[(tile * 4 + 0) * 64 + lane];
          v.y = acc[kx][cig][c][1] + lds[(tile * 4 + 1) * 64 + lane];
          v.z = acc[kx][cig][c][2] + lds[(tile * 4 + 2) * 64 + lane];
          v.w = acc[kx][cig][c][3] + lds[(tile * 4 + 3) * 64 + lane];
          *reinterpret_cast<float4*>(sw + ((size_t)tile * 64 + lane) * 4) = v;
        }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {                   // 16 consecutive threads share an output channel
    float v = bsum[j];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int e = tid + j * W_NT;
    if ((lane & 15) == 0 && e < 1024) slab_b[(size_t)wg * (COUT * KW + COUT) + (e >> 4)] = v;
  }
}

__global__ __launch_bounds__(W_NT) void conv3_wgrad_kernel(const float* __restrict__ p2, const float* __restrict__ dy3,
                                                           float* __restrict__ slab_w, float* __restrict__ slab_b, int n_img) {
  __shared__ float lds[W_LDS];
  conv3_wgrad_body(p2, dy3, slab_w, slab_b, n_img, lds, blockIdx.x, gridDim.x);
}

// ---- data gradient ----------------------------------------------------------------------------------
// 768 threads, one wave per (parity class of the input position, 16-channel tile of ci).  With k = 3,
// s = 2, p = 1 an even input coordinate is reached by tap 1 only (from output y'), an odd one by tap 0
// (from y' + 1) and tap 2 (from y'): class (PY, PX) carries (1 + PY)(1 + PX) taps.
// M = 64 positions (y', x') of the class, N = 16 ci, K = 64 co x taps; B = W[co = 4ks + lq][ci][tap] in
// registers (16 per tap), A = dY[co][y' + doy][x' + dox] from the LDS patch [co][9][16] (row 8 / column 8
// are the zero halo; plane stride 168 = 8 mod 32: lq = 1 lands on banks 8..15 / 24..31).
constexpr int D_RS = 16, D_PS = 168, D_PATCH = COUT * D_PS;
constexpr int D_OS = 260;              // plane stride of the output tile [ci][256]: float4-aligned, 2-way at worst on the 4-byte writes
constexpr int D_NT = 768;
constexpr int D_WLD = KW + 1;                       // weight staging rows [co][433] (conv3w_stage): odd stride, conflict-free lane reads

template <int PY, int PX>
__device__ __forceinline__ void dgrad_class(const float* __restrict__ w, const float* __restrict__ dy3, float* __restrict__ dp2,
                                            float* patch2, int n_img, int nt, int tid, int lane, float4 (&sd)[2], int wg, int nwg) {
  constexpr int NTY = PY ? 2 : 1, NTX = PX ? 2 : 1, T = NTY * NTX;
  float* outt = patch2 + 2 * D_PATCH;
  const int lr = lane & 15, lq = lane >> 4;
  const int ci = 16 * nt + lr;
  float wr[T][16];
#pragma unroll
  for (int ty = 0; ty < NTY; ++ty)
#pragma unroll
    for (int tx = 0; tx < NTX; ++tx) {
      const int ky = PY ? (ty == 0 ? 0 : 2) : 1, kx = PX ? (tx == 0 ? 0 : 2) : 1;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) wr[ty * NTX + tx][ks] = patch2[(4 * ks + lq) * D_WLD + ci * 9 + ky * 3 + kx];     // staged by the kernel
    }
  __syncthreads();                                                   // every wave has its weights: the staging area becomes the patch
  for (int i = tid; i < 2 * D_PATCH; i += D_NT) patch2[i] = 0.f;      // halo row 8 / columns 8.. stay zero
  __syncthreads();
  auto fetch = [&](int img) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * D_NT;
      sd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 1024) sd[j] = *reinterpret_cast<const float4*>(dy3 + (size_t)img * 4096 + 4 * e);
    }
  };
  auto stash = [&](float* buf) {                   // float4 e -> co = e / 16, row = (e / 2) % 8, column 4 (e % 2)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * D_NT;
      if (e < 1024) *reinterpret_cast<float4*>(buf + (e >> 4) * D_PS + ((e >> 1) & 7) * D_RS + 4 * (e & 1)) = sd[j];
    }
  };
  int img = wg;
  if (img < n_img) stash(patch2);                    // fetched by the kernel, under the weight staging
  if (img + nwg < n_img) fetch(img + nwg);
  __syncthreads();
  const int aoff = lq * D_PS + (lr >> 3) * D_RS + (lr & 7);
  int cur = 0;
  for (; img < n_img; img += nwg, cur ^= 1) {
    const float* ab = patch2 + cur * D_PATCH + aoff;
    const int next = img + nwg;
    if (next < n_img) {
      stash(patch2 + (cur ^ 1) * D_PATCH);
      if (next + nwg < n_img) fetch(next + nwg);
    }
    // operand i of the image: M-tile mt = i / (16 T), tap t = (i / 16) % T, k-step ks = i % 16; a 4-deep register ring keeps the
    // reads ahead of their MFMAs (see conv3_fwd_kernel)
    auto aread = [&](int i) {
      const int mt = i / (16 * T), t = (i / 16) % T, ks = i % 16, ty = t / NTX, tx = t % NTX;
      const int doy = (PY && ty == 0) ? 1 : 0, dox = (PX && tx == 0) ? 1 : 0;
      return ab[(2 * mt + doy) * D_RS + dox + ks * 4 * D_PS];
    };
    constexpr int RD = 4, NOP = 4 * 16 * T;
    float xa[RD];
#pragma unroll
    for (int d = 0; d < RD; ++d) xa[d] = aread(d);
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NOP; ++i) {
      const int mt = i / (16 * T), t = (i / 16) % T, ks = i % 16;
      const float x = xa[i % RD];
      if (i + RD < NOP) xa[i % RD] = aread(i + RD);
      if (ks & 1) acc1 = mfma4(x, wr[t][ks], acc1); else acc0 = mfma4(x, wr[t][ks], acc0);
      if (i % 2 == 1) __builtin_amdgcn_sched_barrier(0);
      if (i % (16 * T) == 16 * T - 1) {
        // lane holds positions 4lq..+3 of the tile: y' = 2mt + (lq >> 1), x' = 4(lq & 1) + r, channel ci.  The four parity
        // classes interleave in x and y, so straight to global every store instruction is 64 four-byte writes to 64 cache
        // lines; the image's gradient is assembled in LDS ([ci][16][16], plane stride D_OS) and leaves as coalesced float4.
        const int y = 2 * (2 * mt + (lq >> 1)) + PY;
        float* o = outt + ci * D_OS + y * 16 + 8 * (lq & 1) + PX;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[2 * r] = acc0[r] + acc1[r];
        acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {                  // 48 x 256 floats = 3072 float4, 4 per thread
      const int e = tid + j * D_NT, c = e >> 6, q4 = e & 63;
      *reinterpret_cast<float4*>(dp2 + ((size_t)img * CIN + c) * 256 + 4 * q4) = *reinterpret_cast<const float4*>(outt + c * D_OS + 4 * q4);
    }
    __syncthreads();
  }
}

constexpr int D_LDS = 2 * D_PATCH + CIN * D_OS;
static_assert(D_LDS >= COUT * D_WLD, "weight staging area");
__device__ __forceinline__ void conv3_dgrad_body(const float* __restrict__ w, const float* __restrict__ dy3, float* __restrict__ dp2, int n_img,
                                                 float* patch2, int wg, int nwg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the [64][432] weight matrix through LDS with coalesced float4 loads (rows of 433 words); the waves' register slices are
  // stride-9 / stride-432 gathers of it - straight from global ~20 cache lines per load instruction, 16-64 of them per lane
  float4 sd[2];                                      // the first image's dY: its HBM round trip runs under the weight staging
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + j * D_NT;
    sd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < 1024 && wg < n_img) sd[j] = *reinterpret_cast<const float4*>(dy3 + (size_t)wg * 4096 + 4 * e);
  }
  // ((co, ci) items as in the forward - conflict-free stores instead of 4-way - made this kernel 1.2 us SLOWER: 36 dword loads per
  // thread whose lanes are 36 bytes apart cost the vector memory pipe more than the stores saved; it has no 108-read gather to win back)
  conv3w_stage<D_NT>(patch2, w, tid);
  __syncthreads();
  // per image a class costs 64 MFMAs per tap and tile: 256 / 128 / 128 / 64.  Waves w, w + 4, w + 8 share a SIMD:
  // SIMDs 0..2 get {256, 128, 64} (classes 11, 01, 00 of tile w), SIMD 3 gets the three 128s of class 10.
  const int s = wave & 3, g = wave >> 2;
  if (s == 3) dgrad_class<1, 0>(w, dy3, dp2, patch2, n_img, g, tid, lane, sd, wg, nwg);
  else if (g == 0) dgrad_class<1, 1>(w, dy3, dp2, patch2, n_img, s, tid, lane, sd, wg, nwg);
  else if (g == 1) dgrad_class<0, 1>(w, dy3, dp2, patch2, n_img, s, tid, lane, sd, wg, nwg);
  else dgrad_class<0, 0>(w, dy3, dp2, patch2, n_img, s, tid, lane, sd, wg, nwg);
}

__global__ __launch_bounds__(D_NT) void conv3_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dy3,
                                                           float* __restrict__ dp2, int n_img) {
  __shared__ float patch2[D_LDS];
  conv3_dgrad_body(w, dy3, dp2, n_img, patch2, blockIdx.x, gridDim.x);
}

// ---- weight AND data gradient in ONE launch (round 5) ----------------------------------------------------------------------------
// Both read the same dy3 and neither reads what the other writes.  As two launches of one persistent workgroup per CU each ran ~1.9
// images per workgroup behind its own launch ramp and weight / LDS prologue: 23.7 + 22.6 us for 2 x 10.8 us of matrix pipe.  Here the
// first `nw` workgroups run the weight gradient over ALL images and the others the data gradient: one ramp and one prologue per CU
// for ~3.75 images, and the weight gradient leaves `nw` slab rows instead of 256 (half the bytes for the final fold).
static_assert(W_NT == D_NT, "one block size");
constexpr int B_LDS = W_LDS > D_LDS ? W_LDS : D_LDS;
__global__ __launch_bounds__(W_NT) void conv3_bwd_kernel(const float* __restrict__ p2, const float* __restrict__ w, const float* __restrict__ dy3,
                                                         float* __restrict__ slab_w, float* __restrict__ slab_b, float* __restrict__ dp2,
                                                         int n_img, int nw) {
  __shared__ float lds[B_LDS];
  if ((int)blockIdx.x < nw) conv3_wgrad_body(p2, dy3, slab_w, slab_b, n_img, lds, blockIdx.x, nw);
  else conv3_dgrad_body(w, dy3, dp2, n_img, lds, (int)blockIdx.x - nw, (int)gridDim.x - nw);
}

}  // namespace c3
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
