// conv1 + conv2 + pool forward of the vanilla encoder with conv2 on the bf16 matrix pipe and SPLIT fp32 operands
// (mlhot_set_option("conv2_split", 1); OFF by default: bench.py's `value` stays on the exact-fp32 kernels of conv_tc.h, this
// variant is reported under `extras` with its arithmetic spelled out).
//
// Every fp32 operand is cut into three bf16 pieces by truncation, x = hi + mid + lo EXACTLY (3 x 8 mantissa bits: hi = x with the
// low 16 bits cleared, mid = the same of x - hi, lo = x - hi - mid).  A product x w is then the sum of piece products, each exact in
// fp32, accumulated in fp32 by v_mfma_f32_16x16x32_bf16; 6 of the 9 are kept (hh, hm, mh, hl, lh, mm: the dropped ones are below
// 2^-24 of the product).  scripts/micro/split_bf16_error.py: against float64 this is as exact as the fp32 MFMA it replaces
// (5.4e-7 of the largest output either way); scripts/micro/conv2_split_loop.hip: 2.2 x the fp32 inner loop.
//
// Shape: one persistent 768-thread workgroup per CU, bands of TWO conv2 output rows (16 per image): the a1 patch of a band is
// [5 rows][65 cols][3 pieces][32 ci] bf16 at 208 bytes per position (67.6 KB; two of them: the next band's conv1 is computed
// under this band's MFMAs - conv1 itself stays on the fp32 pipe, K = 9 + bias padded to 12, as in conv_tc.h, and leaves the same
// ReLU sign-bit records for the backward).  Wave = (16 output channels nt) x (row r2 of the band, column half ch): ONE 16 x 16
// output tile, nine K = 32 blocks (one per tap over the 32 input channels) x 6 products = 54 MFMAs; the weights' pieces live in
// 108 registers.  The two rows of a band sit in different waves, so ReLU'd tiles meet in an LDS scratch and the 2 x 2 max-pool
// (+ arg-max) runs one output per thread behind the band barrier.  Outputs (p2, arg-max, sign bits) have the layout of
// c2::conv12_fwd_pool_kernel: the backward kernels do not know which forward ran.
#pragma once
#include "conv_tc.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace c2s {

using c2::f32x4_t;
using c2::ImgSrc;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int NT = 768, CIN = 32, COUT = 48;
constexpr int PROWS = 5, PCOLS = 65, POSB = 208;             // patch position = 3 pieces x 64 B + 16 B padding (bank spread)
constexpr int PATCHB = PROWS * PCOLS * POSB;                  // 67,600 B
constexpr int POOL_LD = 33, POOLF = 2 * COUT * POOL_LD;       // pool scratch [row 2][co 48][col 32 + 1] floats
constexpr int LDS_BYTES = 2 * PATCHB + 2 * POOLF * 4;         // 160,544 B
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
static_assert(COUT * c2::W2_LD * 4 <= 2 * PATCHB, "weight staging area");

#ifdef C2S_TS      // scripts/micro/conv_split_bench.hip: cycle stamps of band C2S_TS (counted per workgroup) of workgroup 0, per wave
__device__ long long g_c2s_ts[12 * 16 + 4];
#define C2S_STAMP(i) do { if (blockIdx.x == 0 && lane == 0 && band_no == C2S_TS) g_c2s_ts[wave * 16 + (i)] = clock64(); } while (0)
#else
#define C2S_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ f32x4_t mfma_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// x = hi + mid + lo exactly; each piece has at most 8 significant bits, i.e. is a bf16 (returned as fp32 bit patterns whose low
// 16 bits are zero)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mid);
  lo = __float_as_uint(r2) & 0xffff0000u;
}

// A fragment of tap t (ky = t / 3, kx = t % 3), piece p, for the wave's 16 positions (pa: position (row 2 r2, column 2 (16 ch + lr)),
// channels 8 lq ..): one ds_read_b128, conflict-free (position stride 104 dwords over the 64 banks)
__device__ __forceinline__ bf16x8_t a_frag(const unsigned char* pa, int t, int p) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(pa + ((t / 3) * PCOLS + t % 3) * POSB + 64 * p));
}
// taps T0 .. T1 - 1 of the wave's tile.  Fragment registers: the hi piece double-buffered (fetched a tap ahead), mid and lo
// single: the tap's MFMAs run lo, mid, hi and each single piece is re-fetched for the next tap right behind its last use, four
// to five MFMAs before its next one (a full double buffer is 8 registers more than the kernel has).  Nothing of conv2 but the
// two accumulators is alive outside a call: the conv1 slice in between needs the registers.
template <int T0, int T1>
__device__ __forceinline__ void conv2_taps(const unsigned char* pa, const bf16x8_t (&wr)[9][3], f32x4_t& acc, f32x4_t& accs) {
  bf16x8_t a0[2], a1, a2;
  a2 = a_frag(pa, T0, 2);
  a1 = a_frag(pa, T0, 1);
  a0[T0 & 1] = a_frag(pa, T0, 0);
#pragma unroll
  for (int t = T0; t < T1; ++t) {
    const int c = t & 1;
    const bool more = t + 1 < T1;
    if (more) a0[c ^ 1] = a_frag(pa, t + 1, 0);
    __builtin_amdgcn_sched_barrier(0);        // or hipcc sinks the reads to their first use, behind an s_waitcnt lgkmcnt(0) per tap
    accs = mfma_bf16(a2, wr[t][0], accs);
    __builtin_amdgcn_sched_barrier(0);
    if (more) a2 = a_frag(pa, t + 1, 2);
    __builtin_amdgcn_sched_barrier(0);
    accs = mfma_bf16(a1, wr[t][0], accs);
    accs = mfma_bf16(a1, wr[t][1], accs);
    __builtin_amdgcn_sched_barrier(0);
    if (more) a1 = a_frag(pa, t + 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    acc = mfma_bf16(a0[c], wr[t][0], acc);
    accs = mfma_bf16(a0[c], wr[t][1], accs);
    accs = mfma_bf16(a0[c], wr[t][2], accs);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(NT) void conv12_fwd_split_kernel(const ImgSrc x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ p2, uint8_t* __restrict__ amax,
                                                               unsigned* __restrict__ m1, int n_img) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  unsigned char* patch0 = lds;
  float* pool0 = reinterpret_cast<float*>(lds + 2 * PATCHB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % 3, pt = wave / 3, r2 = pt >> 1, ch = pt & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int co = nt * 16 + lr;
  const int slot = wave >> 2;            // waves w, w + 4, w + 8 share a SIMD

  // ---- the wave's weight slice, split: wr[tap][piece] = 8 bf16 = W2[co][ci(k' = 8 lq + j)][tap], j < 8, in the patch's K order (c1_tile) --------------------------
  c2::conv2w_stage<NT>(reinterpret_cast<float*>(lds), w, tid);
  __syncthreads();
  bf16x8_t wr[9][3];
  {
    const float* stage = reinterpret_cast<const float*>(lds) + co * c2::W2_LD;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      unsigned pc[3][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) split3(stage[(16 * (j & 1) + 4 * lq + (j >> 1)) * 9 + t], pc[0][j], pc[1][j], pc[2][j]);   // k' = 8 lq + j <-> ci
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        u32x4_t v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (pc[p][2 * i] >> 16) | pc[p][2 * i + 1];
        wr[t][p] = __builtin_bit_cast(bf16x8_t, v);
      }
    }
  }
  const float bn = bias[co];
  __syncthreads();
  for (int i = tid; i < 2 * PATCHB / 16; i += NT) reinterpret_cast<u32x4_t*>(lds)[i] = u32x4_t{0u, 0u, 0u, 0u};   // column 0 (ix = -1) stays zero for good
  c2::Conv1W cw;
  c2::conv1w_load(cw, w1, b1, lr, lq);
  __syncthreads();

  // conv1 M-tiles of a band: tt = patch row (tt >> 2) x column group (tt & 3), 20 of them: wave w takes w and (w < 8) w + 12 -
  // both in column group cg = w & 3, three patch rows apart.  Pixel request of tap k = 4 ks + lq of position lr: byte
  // dl[ks] + 1024 iy1 of the image (iy1 = a1 row 4 b - 1 + r: wave-uniform), with the lane part dl[ks] hugely negative for
  // the padding taps (k > 8, the column left of the image); the row above the image (iy1 = 0, ky = 0) and a whole patch row
  // above it (iy1 = -1) come out negative by themselves, and a negative offset is the `masked` test.
  // Requests go out a band ahead (c1_load); the MFMAs and the ReLU / split / pack / store follow where the wave's slot says.
  const int cg = wave & 3;
  int dl[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int k = 4 * ks + lq, ky = k / 3, kx = k - 3 * ky, ix = 2 * (16 * cg + lr) + kx - 1;
    dl[ks] = (k < 9 && ix >= 0) ? 4 * ((ky - 1) * 128 + ix) : -(1 << 28);
  }
  const float k9 = lq == 1 ? 1.f : 0.f;                                   // the bias column k = 9 (ks = 2)
  const int st_lane = (1 + 16 * cg + 4 * lq) * POSB + 4 * lr;             // patch column = a1 column + 1
  auto c1_load = [&](int tile, int r, float (&v)[3]) {
    const int iy1 = __builtin_amdgcn_readfirstlane(4 * (tile & 15) - 1 + r);
    const char* xi = reinterpret_cast<const char*>(x.img(tile >> 4));
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) v[ks] = *reinterpret_cast<const float*>(xi + (unsigned)max(dl[ks] + 1024 * iy1, 0));
  };
  auto c1_tile = [&](int tile, int r, const float (&v)[3], unsigned char* dstpatch) {
    const int img = tile >> 4, iy1 = __builtin_amdgcn_readfirstlane(4 * (tile & 15) - 1 + r);
    f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float a = __builtin_fmaf(v[ks], dl[ks] + 1024 * iy1 >= 0 ? 1.f : 0.f, (ks == 2 && iy1 >= 0) ? k9 : 0.f);
      c0 = c2::mfma4(a, cw.b[ks][0], c0);
      c1 = c2::mfma4(a, cw.b[ks][1], c1);
    }
    // lane: channels lr (c0) and 16 + lr (c1) of columns 16 cg + 4 lq + reg.  The K order of a position interleaves the two
    // halves (k' = 2 (ci & 15) + (ci >> 4)), so a lane's two channels share a dword per piece.
    unsigned* mrec = m1 + c2::m1_record(img, iy1, cg, n_img);
    unsigned char* d = dstpatch + r * (PCOLS * POSB) + st_lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool p0 = c0[q] > 0.f, p1 = c1[q] > 0.f;
      const float v0 = p0 ? c0[q] : 0.f, v1 = p1 ? c1[q] : 0.f;          // (fmaxf costs a canonicalising v_max on top)
      // hi / mid / lo by truncation; v_perm_b32 packs the upper halves of the two channels, so only the subtrahends are masked
      const float h0 = __uint_as_float(__float_as_uint(v0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
      const float r0 = v0 - h0, r1 = v1 - h1;
      const float m0 = __uint_as_float(__float_as_uint(r0) & 0xffff0000u), mm1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
      const float l0 = r0 - m0, l1 = r1 - mm1;
      unsigned* dq = reinterpret_cast<unsigned*>(d + q * POSB);
      dq[0] = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
      dq[16] = __builtin_amdgcn_perm(__float_as_uint(mm1), __float_as_uint(m0), 0x07060302u);
      dq[32] = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
      const unsigned long long b0 = __builtin_amdgcn_ballot_w64(p0), bb1 = __builtin_amdgcn_ballot_w64(p1);
      *reinterpret_cast<uint4*>(mrec + 4 * q) = make_uint4((unsigned)b0, (unsigned)(b0 >> 32), (unsigned)bb1, (unsigned)(bb1 >> 32));
    }
  };
  const int row0 = wave >> 2;                  // patch rows of the wave's tiles: row0 and (w < 8) row0 + 3

  const int ntiles = n_img * 16;
  int tile = c2::first_tile<16>(blockIdx.x, gridDim.x);
  const int row1 = wave < 8 ? row0 + 3 : row0;
  float px[2][3], pn[2][3];                   // pixels of the band whose a1 this iteration computes | of the one after
  if (tile < ntiles) {
    c1_load(tile, row0, px[0]);
    c1_load(tile, row1, px[1]);
    c1_tile(tile, row0, px[0], patch0);
    if (wave < 8) c1_tile(tile, row0 + 3, px[1], patch0);
    c1_load(min(tile + (int)gridDim.x, ntiles - 1), row0, pn[0]);
    c1_load(min(tile + (int)gridDim.x, ntiles - 1), row1, pn[1]);
  }
  __syncthreads();
  // A fragment of tap (ky, kx), piece p: position (row 2 r2 + ky, column 2 (16 ch + lr) + kx), channels 8 lq .. + 7
  const int abase = ((2 * r2) * PCOLS + 2 * (16 * ch + lr)) * POSB + 16 * lq;
  int cur = 0;
#ifdef C2S_TS
  int band_no = -1;
  const long long ts_c0 = clock64(), ts_w0 = wall_clock64();
#endif
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
#ifdef C2S_TS
    ++band_no;
#endif
    // ONE place in the code where pixel requests are issued and one where they are awaited, a whole band apart (inside the
    // slot branches below hipcc's wait-count pass has to assume the worst path and puts s_waitcnt vmcnt(0) - the pool's own
    // stores included - in front of the conv1 MFMAs)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) px[j][ks] = pn[j][ks];
    {
      const int n2 = min(tile + 2 * (int)gridDim.x, ntiles - 1);
      c1_load(n2, row0, pn[0]);
      c1_load(n2, row1, pn[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(0);
    const unsigned char* pa = patch0 + cur * PATCHB + abase;
    unsigned char* nb = patch0 + (cur ^ 1) * PATCHB;
    const int next = tile + (int)gridDim.x;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, accs = {0.f, 0.f, 0.f, 0.f};      // hh | the five small products
    // the next band's a1 slice (conv1 on the fp32 pipe + ~200 VALU of ReLU / split / pack) into the other patch.  The three waves of a SIMD place it before, inside and behind their conv2 MFMAs: between
    // barriers they would otherwise all be in the same phase, and the matrix pipe would idle while they split.
    auto conv1_next = [&]() {
      if (next < ntiles) {
        c1_tile(next, row0, px[0], nb);
        if (wave < 8) c1_tile(next, row0 + 3, px[1], nb);
      }
    };
    if (slot == 0) conv1_next();
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(1);
    conv2_taps<0, 5>(pa, wr, acc, accs);
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(2);
    if (slot == 1) conv1_next();
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(3);
    conv2_taps<5, 9>(pa, wr, acc, accs);
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(4);
    if (slot == 2) conv1_next();
    __builtin_amdgcn_sched_barrier(0);
    C2S_STAMP(5);
    // this wave's ReLU'd tile -> pool scratch [r2][co][col]: lane (co = 16 nt + lr), columns 16 ch + 4 lq + reg
    float* pool = pool0 + cur * POOLF;
    {
      float* d = pool + (r2 * COUT + co) * POOL_LD + 16 * ch + 4 * lq;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = fmaxf(acc[q] + accs[q] + bn, 0.f);
    }
    C2S_STAMP(6);
    __syncthreads();
    C2S_STAMP(7);
    // 2 x 2 max-pool + arg-max: thread = (channel tid / 16, pooled column tid % 16); window order as the reference's pool kernel
    {
      const int pc = tid >> 4, px = tid & 15, img = tile >> 4, b = tile & 15;
      const float* s0 = pool + pc * POOL_LD + 2 * px;
      const float* s1 = s0 + COUT * POOL_LD;
      const float c0 = s0[0], c1 = s0[1], c2v = s1[0], c3 = s1[1];
      float best = c0; unsigned which = 0;
      if (c1 > best) { best = c1; which = 1; }
      if (c2v > best) { best = c2v; which = 2; }
      if (c3 > best) { best = c3; which = 3; }
      const size_t o = (((size_t)img * COUT + pc) * 16 + b) * 16 + px;
      p2[o] = best;
      amax[o] = (uint8_t)which;
    }
    C2S_STAMP(8);
  }
#ifdef C2S_TS
  if (blockIdx.x == 0 && tid == 0) { g_c2s_ts[192] = clock64() - ts_c0; g_c2s_ts[193] = wall_clock64() - ts_w0; g_c2s_ts[194] = band_no + 1; }
#endif
}

}  // namespace c2s
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
