// The vanilla encoder's conv1 + conv2 + pool block with conv2 on the bf16 matrix pipe and SPLIT fp32 operands: forward, data
// gradient (+ conv1's weight gradient) and weight gradient (mlhot_set_option("conv2_split", 1); OFF by default: bench.py's
// `value` stays on the exact-fp32 kernels of conv_tc.h, these variants are reported under `extras` with their arithmetic spelled out).
//
// Every fp32 operand is cut into three bf16 pieces, x = hi + mid + lo EXACTLY: hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid,
// each conversion round-to-nearest-even (v_cvt_pk_bf16_f32: two values per instruction, already packed).  The residual after two
// 8-bit pieces has at most 6 significant bits, so lo is exact.  A product x w is then the sum of piece products, each exact in fp32,
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16; 6 of the 9 are kept (hh, hm, mh, hl, lh, mm).  With nearest pieces |mid| <= 2^-9 |x|
// and |lo| <= 2^-18 |x|, so the dropped ml + lm + ll are below 2^-26 |x w| - under the 2^-25 half-ulp a single fp32 rounding of the
// product's sum costs - and of either sign (pieces cut by truncation, the round-3 form, dropped up to 2^-23, always towards zero).
// The five small products of a K block are summed in an accumulator of their own and join the hh sum once, at the end.
// tests/test_gpu_parity.py::test_split_precision_error_vs_fp32_mfma compares both arithmetics with float64 on adversarial operands;
// scripts/micro/split_bf16_error.py is the CPU model; scripts/micro/conv2_split_loop.hip: 2.2 x the fp32 inner loop.
//
// FORWARD.  One persistent 768-thread workgroup per CU, bands of TWO conv2 output rows (16 per image): the a1 patch of a band is
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

// (v0, v1) -> three dwords of packed bf16 pairs (low half = v0's piece): v = hi + mid + lo exactly, nearest pieces (see the header)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float v0, float v1) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{v0, v1}, bf16x2_t));     // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void split3_pk(float v0, float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
#ifdef C2S_TRUNC      // A/B only: the round-3 pieces, cut by truncation (2 ands + v_perm per piece pair instead of v_cvt_pk + shift + and)
  const float h0 = __uint_as_float(__float_as_uint(v0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
  const float q0 = v0 - h0, q1 = v1 - h1;
  const float m0 = __uint_as_float(__float_as_uint(q0) & 0xffff0000u), m1 = __uint_as_float(__float_as_uint(q1) & 0xffff0000u);
  hi = __builtin_amdgcn_perm(__float_as_uint(h1), __float_as_uint(h0), 0x07060302u);
  mid = __builtin_amdgcn_perm(__float_as_uint(m1), __float_as_uint(m0), 0x07060302u);
  lo = __builtin_amdgcn_perm(__float_as_uint(q1 - m1), __float_as_uint(q0 - m0), 0x07060302u);
  return;
#endif
  hi = pk_bf16(v0, v1);
  const float r0 = v0 - __uint_as_float(hi << 16), r1 = v1 - __uint_as_float(hi & 0xffff0000u);
  mid = pk_bf16(r0, r1);
  lo = pk_bf16(r0 - __uint_as_float(mid << 16), r1 - __uint_as_float(mid & 0xffff0000u));
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
      unsigned v[3][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)     // k' = 8 lq + j <-> ci = 16 (j & 1) + 4 lq + (j >> 1), j = 2 i | 2 i + 1
        split3_pk(stage[(4 * lq + i) * 9 + t], stage[(16 + 4 * lq + i) * 9 + t], v[0][i], v[1][i], v[2][i]);
#pragma unroll
      for (int p = 0; p < 3; ++p) wr[t][p] = __builtin_bit_cast(bf16x8_t, u32x4_t{v[p][0], v[p][1], v[p][2], v[p][3]});
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
      unsigned* dq = reinterpret_cast<unsigned*>(d + q * POSB);
      split3_pk(v0, v1, dq[0], dq[16], dq[32]);
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
  // 2 x 2 max-pool + arg-max of a finished band: thread = (channel tid / 16, pooled column tid % 16); window order as the reference's
  // pool kernel.  It runs BETWEEN the two MFMA halves of the band after - right behind the band's barrier it was ~1 k cycles of a
  // 9 k-cycle band in which every wave did the same dozen instructions and two stores and the matrix pipes idled (knock-outs: 40 us
  // of the kernel's 107 were neither conv1 nor MFMAs); the scratch is double-buffered, so the band after does not touch it.
  auto pool_store = [&](int t, int c) {
    const float* pool = pool0 + c * POOLF;
    const int pc = tid >> 4, px = tid & 15, img = t >> 4, b = t & 15;
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
  };
  int prev_tile = -1, prev_cur = 0;
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
    if (prev_tile >= 0) pool_store(prev_tile, prev_cur);       // the band before this one: its tiles met in the scratch at its barrier
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
    prev_tile = tile; prev_cur = cur;          // pooled in the middle of the next band (or behind the loop)
    C2S_STAMP(8);
  }
  if (prev_tile >= 0) pool_store(prev_tile, prev_cur);
#ifdef C2S_TS
  if (blockIdx.x == 0 && tid == 0) { g_c2s_ts[192] = clock64() - ts_c0; g_c2s_ts[193] = wall_clock64() - ts_w0; g_c2s_ts[194] = band_no + 1; }
#endif
}


// ==================================================================================================================================
// DATA GRADIENT of conv2 (+ conv1's ReLU mask + conv1's weight / bias gradient), the split twin of c2::conv12_dgrad_kernel: same
// inputs, same slab out, same band walk (one image x 8 a1 rows per band, one persistent 512-thread workgroup per CU), same hand-over
// of a finished row half to the conv1 weight-gradient MFMAs (c2::D12Pend; those stay on the fp32 pipe: K = positions, 8 per half).
//
// d a1[pos][ci] = sum over (tap, co) of dY[pos'(tap)][co] W2[co][ci][tap] runs over K = (tap, co) in blocks of 32 on
// v_mfma_f32_16x16x32_bf16: A = the un-pooled dY (16 conv2 columns ox = 16 xh + lr of one row, 8 consecutive co per lane), B = the
// wave's 16 input channels of W2 (pieces in registers).  The dY patch of a band is three piece planes [row 5][col 33][co 48] bf16,
// DENSE - 96 bytes per position - so that a K run continues from position ox into ox + 1: for an odd a1 column x = 2 ox + 1 the
// two taps of a tap row are kx = 2 at ox and kx = 0 at ox + 1, i.e. 96 consecutive co-slots = three K blocks whose fragments sit at
// lane base + 0 / 64 / 128 bytes.  The even column's single tap kx = 1 (48 co = 1.5 blocks) reads the first two of the SAME
// fragments against weights that are zero in the last two k-groups.  So a tap row costs 3 fragment reads x 3 pieces and 5 blocks x 6
// piece products; an even a1 row has one tap row (ky = 1), an odd one two (ky = 2 on dY row (y - 1) / 2, ky = 0 on the row below).
// Waves specialise by row parity - wave = (ci tile nt, parity PY, row group g): 60 / 120 weight registers, 30 / 60 MFMAs per half
// - and the two waves of a SIMD (w, w + 4) have different parities, so every SIMD issues the same 360 MFMAs per band.
// A 16-lane group of a ds_read_b128 covers the 64 banks once (position stride 24 dwords, k-group stride 4).
// ==================================================================================================================================
namespace dg {
constexpr int NT2 = c2::NT2;
constexpr int DCOLS = 33, DPOSB = 96, DROWB = DCOLS * DPOSB, DPLANEB = 5 * DROWB;       // 3,168 B per row, 15,840 B per piece plane
constexpr int DYPB = 3 * DPLANEB;                                                          // 47,520 B
constexpr int STRIPB = c2::STRIP_FLOATS * 4;                                               // 8,976 B
constexpr int BUFB = DYPB + STRIPB;                                                        // 56,496 B per band buffer
constexpr int LDS_BYTES = 2 * BUFB + 8 * 16 * 16 * 4;                                      // 121,184 B
static_assert(LDS_BYTES <= 160 * 1024 && BUFB % 16 == 0 && DPLANEB % 16 == 0, "LDS");
static_assert(COUT * c2::W2_LD * 4 <= 2 * BUFB, "weight staging area");

// pooled cells go in pairs of adjacent channels; a 32-lane half of a wave-item is 8 pairs x 4 px.  Its stores go to dword
// (row, 2 px + dx) * 24 + pair: bank (16 px + pair) mod 32 - 16 different banks, 2-way, which a ds_write_b32 absorbs (px fastest over
// 16 lanes put 8 lanes on a bank: half of this kernel's LDS cycles were conflict cycles); its loads are 16-byte runs of 8 channels.
__device__ __forceinline__ bf16x8_t dfrag(const unsigned char* base, int off) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + off));
}

// the six piece products of one K block for the odd-column accumulators (d: hh | ds: the five small ones) and - for the first
// two blocks of a tap row - the even-column ones (e | es) on the same fragments; the chains alternate
template <bool WITH_E>
__device__ __forceinline__ void block6(const bf16x8_t (&a)[3], const bf16x8_t (&wd)[3], const bf16x8_t (&we)[3],
                                       f32x4_t& e, f32x4_t& es, f32x4_t& d, f32x4_t& ds) {
  ds = mfma_bf16(a[2], wd[0], ds);
  if (WITH_E) es = mfma_bf16(a[2], we[0], es);
  ds = mfma_bf16(a[1], wd[0], ds);
  if (WITH_E) es = mfma_bf16(a[1], we[0], es);
  ds = mfma_bf16(a[1], wd[1], ds);
  if (WITH_E) es = mfma_bf16(a[1], we[1], es);
  d = mfma_bf16(a[0], wd[0], d);
  if (WITH_E) e = mfma_bf16(a[0], we[0], e);
  ds = mfma_bf16(a[0], wd[1], ds);
  if (WITH_E) es = mfma_bf16(a[0], we[1], es);
  ds = mfma_bf16(a[0], wd[2], ds);
  if (WITH_E) es = mfma_bf16(a[0], we[2], es);
}

// One row half (a1 row yl of the band, columns 32 xh ..): NR = 1 + PY tap rows of 3 blocks.  wd[3 NR][piece], we[2 NR][piece].
template <int PY, bool HAS_PREV>
__device__ __forceinline__ void half(const unsigned char* dyp, const float* strip, const bf16x8_t (&wd)[3 * (1 + PY)][3],
                                     const bf16x8_t (&we)[2 * (1 + PY)][3], const unsigned* __restrict__ mrow, f32x4_t& z, f32x4_t& z2,
                                     int yl, int xh, const c2::D12Lane& ln, const c2::D12Pend& prev, c2::D12Pend& out) {
  constexpr int NR = 1 + PY, NB = 3 * NR;
  const int lq = ln.lq;
  const unsigned* mrec = mrow + (2 * xh + (lq >> 1)) * c2::M1_REC + 2 * (ln.ci >> 4) + (lq & 1);
  uint4 mb;
  mb.x = mrec[0]; mb.y = mrec[4]; mb.z = mrec[8]; mb.w = mrec[12];
  float t0[4], t1[4];
  if (HAS_PREV) c2::d12_pend_taps(strip, prev, t0, t1);
  // dY row of the first tap row: even a1 row y: ky = 1 at y / 2; odd: ky = 2 at (y - 1) / 2, then ky = 0 one row further down
  const unsigned char* base = dyp + ((yl >> 1) * DCOLS + 16 * xh + ln.lr) * DPOSB + 16 * lq;
  f32x4_t e = {0.f, 0.f, 0.f, 0.f}, es = e, d = e, ds = e;
  bf16x8_t a[2][3];
  auto fetch = [&](int b, int slot) {
#pragma unroll
    for (int p = 0; p < 3; ++p) a[slot][p] = dfrag(base, (b / 3) * DROWB + 64 * (b % 3) + p * DPLANEB);
  };
  fetch(0, 0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) fetch(b + 1, (b + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (b % 3 < 2) block6<true>(a[b & 1], wd[b], we[2 * (b / 3) + b % 3], e, es, d, ds);
    else block6<false>(a[b & 1], wd[b], wd[b], e, es, d, ds);
    if (HAS_PREV) {          // the previous half's masks + conv1 weight-gradient MFMAs, spread over this half's blocks
      if (NB == 3) { c2::d12_pend_step(prev, b, t0[b], t1[b], ln, z, z2); if (b == 2) c2::d12_pend_step(prev, 3, t0[3], t1[3], ln, z, z2); }
      else if (b < 4) c2::d12_pend_step(prev, b, t0[b], t1[b], ln, z, z2);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) { out.e[r] = e[r] + es[r]; out.d[r] = d[r] + ds[r]; }
  out.mb = mb;
  out.tpoff = (2 * yl) * c2::SRS + ln.toff + 4 * (16 * xh + 4 * lq);
}

// the wave's weight fragments from the staged [48][289] matrix: block b of the odd-column run of tap row ky holds k-groups
// G = 4 (b % 3) + lq: G < 6 -> tap (ky, 2), co = 8 G ..; else tap (ky, 0), co = 8 (G - 6) ..; the even-column run (two blocks) holds
// tap (ky, 1), co = 8 G .. for G < 6 and zeros behind
__device__ __forceinline__ void wfrag(const float* stage, int ci, int tap, int co0, float sign_or_zero, bf16x8_t (&out)[3]) {
  unsigned v[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* sp = stage + (co0 + 2 * i) * c2::W2_LD + ci * 9 + tap;
    const float w0 = sp[0] * sign_or_zero, w1 = sp[c2::W2_LD] * sign_or_zero;
    split3_pk(w0, w1, v[0][i], v[1][i], v[2][i]);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) out[p] = __builtin_bit_cast(bf16x8_t, u32x4_t{v[p][0], v[p][1], v[p][2], v[p][3]});
}

template <int PY>
__device__ __forceinline__ void run(unsigned char* lds, const ImgSrc& x, const unsigned* __restrict__ m1, const float* __restrict__ dp2,
                                    const float* __restrict__ p2, const uint8_t* __restrict__ amax, float* __restrict__ slab1, int n_img,
                                    int tid, int lane, int wave) {
  constexpr int NR = 1 + PY;
  const int nt = wave & 1, g = (wave >> 1) & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int ci = 16 * nt + lr;
  float* red = reinterpret_cast<float*>(lds + 2 * BUFB);

  // v_mfma_f32_16x16x32_bf16 adds in four steps of eight products, and in every step FLOORS all nine addends - the accumulator
  // included - to 25 bits below the step's largest product before it sums them (scripts/micro/mfma_bf16_accum.hip): a small
  // accumulator that passes a step of huge products loses its low bits towards minus infinity even when those products cancel.
  // One such loss is no larger than the fp32 chain's rounding in the same spot, but it always points the same way, and the conv1
  // weight gradient sums d a1 over every position of the batch.  So half of the waves (row group g = 1) hold the NEGATED weights:
  // their accumulators carry -d a1 with the same downward losses, the sign comes back for free in the tap operand of the conv1
  // weight-gradient MFMAs (tmask / tconst), and over the two row groups of a band the losses cancel instead of adding up.  The
  // workgroup's bands alternate in sign on top of that (band loop).
  const float sg = g ? -1.f : 1.f;
  bf16x8_t wd[3 * NR][3], we[2 * NR][3];
  {
    const float* stage = reinterpret_cast<const float*>(lds);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int ky = PY ? (r == 0 ? 2 : 0) : 1;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int G = 4 * b + lq;
        wfrag(stage, ci, ky * 3 + (G < 6 ? 2 : 0), 8 * (G < 6 ? G : G - 6), sg, wd[3 * r + b]);
        if (b < 2) wfrag(stage, ci, ky * 3 + 1, 8 * (G < 6 ? G : 0), G < 6 ? sg : 0.f, we[2 * r + b]);
      }
    }
  }
  __syncthreads();        // the staging area becomes the band buffers
  const int toff = lr < 9 ? (lr / 3) * c2::SRS + lr % 3 : 0;
  c2::D12Lane ln{ci, lr, lq, toff, lr < 9 ? sg : 0.f, lr == 9 ? sg : 0.f, (unsigned)(ci & 15), (unsigned)(ci & 15) + 16u};      // tmask / tconst flip sign from band to band, see the band loop
  f32x4_t z = {0.f, 0.f, 0.f, 0.f}, z2 = z;

  for (int bsel = 0; bsel < 2; ++bsel) {
    unsigned char* b0 = lds + bsel * BUFB;
    float* stripb = reinterpret_cast<float*>(b0 + DYPB);
    for (int i = tid; i < 3 * 5 * (DPOSB / 4); i += NT2)       // halo column 32 of every row and piece
      reinterpret_cast<unsigned*>(b0 + (i / (5 * 24)) * DPLANEB + ((i / 24) % 5) * DROWB + 32 * DPOSB)[i % 24] = 0u;
    for (int i = tid; i < 17; i += NT2) stripb[i * c2::SRS] = 0.f;                                   // ix = -1 column
  }
  const int ntiles = n_img * 8;
  // pooled cells of a band in pairs of adjacent output channels (one dword of packed pieces): 24 pairs x 3 pooled rows x 16 px = 18
  // wave-items; wave w takes items w, w + 8 and (w < 2) w + 16.  Item W = (pooled row W / 6, px half (W / 3) & 1, pair group W % 3):
  // everything but the lane's (pair, px) inside the item is wave-uniform, and the lane part is folded into two offsets per item
  // ONCE - written with the item index derived from tid every band, the index arithmetic (divisions by 3 and 6, a 64-bit address per
  // load) cost this staging 3 - 4.8 k cycles per wave and band where its ~200 instructions should take under 1 k (stamps).
  float cdp[3][2], cp[3][2]; unsigned cam[3][2];
  float4 sv[2];
  unsigned cell_g[3], cell_l[3];                     // element offset inside an image's band slice | byte offset in the patch
  {
    const int pl = lane & 7, xl = (lane >> 3) & 7;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int W = wave + 8 * j, pr = 8 * (W % 3) + pl, px = 8 * ((W / 3) & 1) + xl, pyl = W / 6;
      cell_g[j] = (unsigned)(((2 * pr) * 16 + pyl) * 16 + px);
      cell_l[j] = (unsigned)(((2 * pyl) * DCOLS + 2 * px) * DPOSB + 4 * pr);
    }
  }
  const int srow0 = tid >> 5, sx4 = tid & 31;        // strip float4 (row, x4): tid, and row 16 for the first 32 threads
  auto cells_fetch = [&](int t) {
    const int img = t >> 3, band = t & 7;
    const float* dpb = dp2 + ((size_t)img * COUT * 256 + 32 * band);
    const float* pb = p2 + ((size_t)img * COUT * 256 + 32 * band);
    const uint8_t* ab = amax + ((size_t)img * COUT * 256 + 32 * band);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int W = wave + 8 * j;
      cdp[j][0] = cdp[j][1] = 0.f; cp[j][0] = cp[j][1] = 0.f; cam[j][0] = cam[j][1] = 0u;
      if (W < 18 && 2 * band + W / 6 < 16) {          // wave-uniform: a scalar branch
        cdp[j][0] = dpb[cell_g[j]]; cp[j][0] = pb[cell_g[j]]; cam[j][0] = ab[cell_g[j]];
        cdp[j][1] = dpb[cell_g[j] + 256u]; cp[j][1] = pb[cell_g[j] + 256u]; cam[j][1] = ab[cell_g[j] + 256u];
      }
    }
    const float* xs = x.img(img) + (16 * band - 1) * 128;       // image strip: 17 rows x 32 float4 from row 16 band - 1
    sv[0] = make_float4(0.f, 0.f, 0.f, 0.f); sv[1] = sv[0];
    if (band > 0 || srow0 > 0) sv[0] = *reinterpret_cast<const float4*>(xs + srow0 * 128 + 4 * sx4);
    if (wave == 0 && lane < 32) sv[1] = *reinterpret_cast<const float4*>(xs + 16 * 128 + 4 * sx4);
  };
  auto cells_store = [&](unsigned char* b0, bool negate) {          // negate: the band's sign (the workgroup's bands alternate, see `sg`)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int W = wave + 8 * j;
      if (W < 18) {
        unsigned pc[3];
        const float g0 = cp[j][0] > 0.f ? cdp[j][0] : 0.f, g1 = cp[j][1] > 0.f ? cdp[j][1] : 0.f;
        split3_pk(negate ? -g0 : g0, negate ? -g1 : g1, pc[0], pc[1], pc[2]);
        unsigned char* d = b0 + cell_l[j];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {          // window position (row w4 >> 1, column w4 & 1): the pair's pieces where its arg-max points
          if (w4 >= 2 && W / 6 == 2) continue;    // the halo pooled row gives only its upper row
          const unsigned m = (cam[j][0] == (unsigned)w4 ? 0xffffu : 0u) | (cam[j][1] == (unsigned)w4 ? 0xffff0000u : 0u);
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(d + (w4 >> 1) * DROWB + (w4 & 1) * DPOSB + p * DPLANEB) = pc[p] & m;
        }
      }
    }
    float* stripb = reinterpret_cast<float*>(b0 + DYPB);
    {
      float* d = stripb + srow0 * c2::SRS + 1 + 4 * sx4;
      d[0] = sv[0].x; d[1] = sv[0].y; d[2] = sv[0].z; d[3] = sv[0].w;
      if (wave == 0 && lane < 32) {
        float* d1 = stripb + 16 * c2::SRS + 1 + 4 * sx4;
        d1[0] = sv[1].x; d1[1] = sv[1].y; d1[2] = sv[1].z; d1[3] = sv[1].w;
      }
    }
  };
  int tile = c2::first_tile<8>(blockIdx.x, gridDim.x);
  __syncthreads();
  if (tile < ntiles) { cells_fetch(tile); cells_store(lds, false); }
  if (tile + (int)gridDim.x < ntiles) cells_fetch(tile + gridDim.x);
  __syncthreads();
  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const unsigned char* dyp = lds + cur * BUFB;
    const float* strip = reinterpret_cast<const float*>(dyp + DYPB);
    const int img = tile >> 3, band = tile & 7;
    // this wave's a1 rows of the band: PY + 4 g and PY + 4 g + 2, both column halves
    const int y0 = PY + 4 * g;
    const unsigned* mrow = m1 + ((size_t)img * 64 + 8 * band + y0) * 64;
#ifdef MLHOT_TS
    const bool tsb = tf::g_ts_dev && blockIdx.x == 0 && lane == 0 && (tile - c2::first_tile<8>(0, gridDim.x)) / (int)gridDim.x == 6;   // 7th band of workgroup 0
#define DGS_STAMP(i) do { if (tsb) tf::g_ts_dev[448 + wave * 6 + (i)] = clock64(); } while (0)
#else
#define DGS_STAMP(i) do { } while (0)
#endif
    DGS_STAMP(0);
    // The workgroup's bands alternate in sign as well: this band's dY went into the patch as +dY (cur = 0) or -dY (cur = 1), and the
    // sign returns in the tap operand.  The row groups' weight signs cancel the floors inside a band; over the 15 bands a workgroup
    // sums for the 480-image batch what is left of them still pointed one way (db1 against float64 at 480 images: 4.9 / 9.9 x the
    // fp32 kernel's error on the range / cancelling cases with the row-group signs alone).
    const c2::D12Lane& lb = ln;
    c2::D12Pend pa, pb;
    half<PY, false>(dyp, strip, wd, we, mrow, z, z2, y0, 0, lb, pb, pa);
    half<PY, true>(dyp, strip, wd, we, mrow, z, z2, y0, 1, lb, pa, pb);
    DGS_STAMP(1);
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) {
      cells_store(lds + (cur ^ 1) * BUFB, cur == 0);
      if (next + (int)gridDim.x < ntiles) cells_fetch(next + gridDim.x);
    }
    DGS_STAMP(2);
    half<PY, true>(dyp, strip, wd, we, mrow + 128, z, z2, y0 + 2, 0, lb, pb, pa);
    half<PY, true>(dyp, strip, wd, we, mrow + 128, z, z2, y0 + 2, 1, lb, pa, pb);
    c2::d12_flush(strip, pb, lb, z, z2);
    ln.tmask = -ln.tmask; ln.tconst = -ln.tconst;        // the next band's dY is in the patch with the other sign
    DGS_STAMP(3);
    __syncthreads();
    DGS_STAMP(4);
  }
  // conv1 gradient partials: wave (nt, PY, g) holds Z[ci = 16 nt + 4 lq + r][tap column lr] of its rows
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * lq + r) * 16 + lr] = z[r] + z2[r];
}
}  // namespace dg

__global__ __launch_bounds__(dg::NT2) void conv12_dgrad_split_kernel(const ImgSrc x, const unsigned* __restrict__ m1,
                                                                      const float* __restrict__ dp2, const float* __restrict__ p2,
                                                                      const uint8_t* __restrict__ amax, const float* __restrict__ w,
                                                                      float* __restrict__ slab1, int n_img) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[dg::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  c2::conv2w_stage<dg::NT2>(reinterpret_cast<float*>(lds), w, tid);
  __syncthreads();
  if (wave & 4) dg::run<1>(lds, x, m1, dp2, p2, amax, slab1, n_img, tid, lane, wave);
  else dg::run<0>(lds, x, m1, dp2, p2, amax, slab1, n_img, tid, lane, wave);
  __syncthreads();
  const float* red = reinterpret_cast<const float*>(lds + 2 * dg::BUFB);
  {
    const int c = tid >> 4, q = tid & 15, ntc = c >> 4, cl = c & 15;     // channel c, tap column q; waves with (wave & 1) == ntc hold it
    float sacc = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) sacc += red[((2 * g4 + ntc) * 16 + cl) * 16 + q];
    if (q < 10) slab1[(size_t)blockIdx.x * 320 + (q < 9 ? c * 9 + q : 288 + c)] = sacc;     // [block][32 x 9 weights | 32 biases]
  }
}


// ==================================================================================================================================
// WEIGHT (+ bias) GRADIENT of conv2, the split twin of c2::conv12_wgrad_kernel: same inputs, the same slab rows out (accumulator
// order, un-permuted by SumParts kind 1), a1 recomputed from the image by conv1 on the fp32 pipe.
//
// dW2[co][ci][tap] = sum over positions of dY[pos][co] a1[pos'(tap)][ci] has K = POSITIONS: a K block of v_mfma_f32_16x16x32_bf16
// is one conv2 output row (32 ox), and a lane's fragment is 8 consecutive ox.  Along ox the taps read a1 columns 2 ox + kx - 1,
// every second one, so the patch keeps the even and the odd a1 columns in separate planes (E entry i = column 2 i, O entry i =
// column 2 i + 1): kx = 1 reads E at ox, kx = 2 reads O at ox - contiguous 16-byte fragments - and kx = 0 reads column 2 ox - 1 =
// O at ox - 1, which would sit 2 bytes off: it runs over ox' = ox - 1 instead, on O at ox' against a SECOND copy of dY moved
// one position (dYs[ox'] = dY[ox' + 1], dYs[31] = 0; the term ox' = -1 is the zero padding).  conv1's accumulator layout gives a
// lane columns 4 c .. 4 c + 3 of a row: two E neighbours and two O neighbours - each pair is ONE v_cvt_pk split and one dword store
// per piece.
// Band = image x 2 conv2 rows (5 a1 rows): patch [ci 32][row 5][E | O][piece 3][32 entries] bf16 = 1,920 B per channel, stride
// 1,952 B, dY [plain | moved][piece 3][co 48][row 2][32 ox] bf16 with a channel stride of 160 B: 108.5 KB, single-buffered (stage - barrier -
// MFMAs - barrier; the next band's pixels and cells are requested before the MFMAs).  Wave = (M-tile triple mg = 3 of the 18 (tap,
// ci half) tiles, row ph) x all 3 co tiles: 9 accumulators, 54 MFMAs per band, every one on a different accumulator than the one
// before it.  All six piece products go into the ONE accumulator of their tile: it sums the whole batch share of the workgroup, so
// it is large beside any single product and the five small products have nothing to gain from an accumulator of their own.
// (Measured and dropped: the next band's conv1 tiles and dY cell computed into registers - 30 dwords of packed pieces - between this
// band's M-tiles and only STORED between the barriers: 102 -> 123 us.  The kernel is bound by vector issue - knock-outs put conv1 +
// split at 38 us, the bf16 MFMAs at 26, the patch stores at 13: they add up, they do not overlap - so moving work under the MFMAs moves
// nothing, and the fp32 MFMAs of conv1 between the bf16 ones cost both pipes.)
// ==================================================================================================================================
namespace wg {
constexpr int CISB = 1952, PATCHB = CIN * CISB;                       // 62,464 B
constexpr int COSB = 160, DYPIECEB = COUT * COSB, DYCOPYB = 3 * DYPIECEB;     // 7,680 B per piece, 23,040 B per copy
constexpr int LDS_BYTES = PATCHB + 2 * DYCOPYB;                       // 108,544 B
// ds_read_b128 serves 16 lanes per cycle - 8 channels of one k-group and the OTHER 8 channels of the next k-group
// (MI355X_MICROARCH.md, LDS) - so a fragment read is conflict-free when the channel stride is an odd multiple of 8 dwords (dY: 40) or,
// for the patch, 488 dwords with the 16-byte granule of a 64-byte row XORed with (channel >> 2) & 3: that XOR is what keeps conv1's
// dword stores (lane = channel x column quad) at 2 lanes per bank (tests/test_index_maps.py restates both counts)
static_assert(LDS_BYTES <= 160 * 1024 && CISB % 16 == 0 && (CISB / 4) % 16 == 8 && (COSB / 4) % 16 == 8, "LDS layout");
static_assert(9 * 9 * 4 * 64 * 4 + 2 * COUT * 4 <= LDS_BYTES, "epilogue fold area");
__device__ __forceinline__ int prow(int row, int plane, int piece) { return ((row * 2 + plane) * 3 + piece) * 64; }
}  // namespace wg

__global__ __launch_bounds__(NT) void conv12_wgrad_split_kernel(const ImgSrc x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                                 const float* __restrict__ dp2, const float* __restrict__ p2,
                                                                 const uint8_t* __restrict__ amax, float* __restrict__ slab_w,
                                                                 float* __restrict__ slab_b, int n_img) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[wg::LDS_BYTES];
  unsigned char* patch = lds;
  unsigned char* dyp = lds + wg::PATCHB;            // plain copy; the moved one behind it
  unsigned char* dys = dyp + wg::DYCOPYB;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mg = wave % 6, ph = wave / 6;
  const int lr = lane & 15, lq = lane >> 4;
  for (int i = tid; i < wg::LDS_BYTES / 16; i += NT) reinterpret_cast<u32x4_t*>(lds)[i] = u32x4_t{0u, 0u, 0u, 0u};   // padding, dYs[31]
  c2::Conv1W cw;
  c2::conv1w_load(cw, w1, b1, lr, lq);

  // conv1 M-tiles of a band as in the forward: tt = patch row (tt >> 2) x column group (tt & 3), wave w takes w and (w < 8) w + 12
  const int cg = wave & 3;
  int dl[3];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int k = 4 * ks + lq, ky = k / 3, kx = k - 3 * ky, ix = 2 * (16 * cg + lr) + kx - 1;
    dl[ks] = (k < 9 && ix >= 0) ? 4 * ((ky - 1) * 128 + ix) : -(1 << 28);
  }
  const float k9 = lq == 1 ? 1.f : 0.f;
  auto c1_load = [&](int tile, int r, float (&v)[3]) {
    const int iy1 = __builtin_amdgcn_readfirstlane(4 * (tile & 15) - 1 + r);
    const char* xi = reinterpret_cast<const char*>(x.img(tile >> 4));
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) v[ks] = *reinterpret_cast<const float*>(xi + (unsigned)max(dl[ks] + 1024 * iy1, 0));
  };
  // lane: channels lr (c0) and 16 + lr (c1), a1 columns 16 cg + 4 lq + q: E entries 8 cg + 2 lq + {0, 1} <- q = 0, 2; O <- q = 1, 3
  const int st_lane = 4 * lq, sw_lane = (lr >> 2) & 3;          // dword lq of granule cg ^ sw_lane
  auto c1_tile = [&](int tile, int r, const float (&v)[3]) {
    const int iy1 = __builtin_amdgcn_readfirstlane(4 * (tile & 15) - 1 + r);
    f32x4_t c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float a = __builtin_fmaf(v[ks], dl[ks] + 1024 * iy1 >= 0 ? 1.f : 0.f, (ks == 2 && iy1 >= 0) ? k9 : 0.f);
      c0 = c2::mfma4(a, cw.b[ks][0], c0);
      c1 = c2::mfma4(a, cw.b[ks][1], c1);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4_t c = h ? c1 : c0;
      float t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = __int_as_float(max(__float_as_int(c[q]), 0));      // ReLU on the bit pattern (conv_tc.h c1t_post)
      unsigned char* d = patch + (16 * h + lr) * wg::CISB + 16 * (cg ^ sw_lane) + st_lane;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        unsigned pc[3];
        split3_pk(t[pl], t[pl + 2], pc[0], pc[1], pc[2]);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(d + wg::prow(r, pl, p)) = pc[p];
      }
    }
  };
  const int row0 = wave >> 2, row1 = wave < 8 ? row0 + 3 : row0;

  // the band's pooled cells: one per thread (48 co x 16 px of ONE pooled row)
  const int cpx = tid & 15, cco = tid >> 4;
  float cdp = 0.f, cp = 0.f; unsigned cam = 0u;
  auto cell_fetch = [&](int t) {
    const size_t o = (((size_t)(t >> 4) * COUT + cco) * 16 + (t & 15)) * 16 + cpx;
    cdp = dp2[o]; cp = p2[o]; cam = amax[o];
  };
  float bsum = 0.f;
  auto cell_store = [&](bool negate) {
    const float g0 = cp > 0.f ? cdp : 0.f;
    bsum += g0;
    const float g = negate ? -g0 : g0;
    unsigned pc[3];
    split3_pk(g, g, pc[0], pc[1], pc[2]);
    // window position w4 = 2 row + column: the plain copy's dword px of a row is (ox = 2 px | 2 px + 1), the moved copy's (2 px + 1 | 2 px + 2)
    unsigned char* dp = dyp + cco * wg::COSB + 4 * cpx;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const unsigned m = (cam == (unsigned)(2 * dy) ? 0xffffu : 0u) | (cam == (unsigned)(2 * dy + 1) ? 0xffff0000u : 0u);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned v = pc[p] & m;
        unsigned char* a = dp + p * wg::DYPIECEB + 64 * dy;
        *reinterpret_cast<unsigned*>(a) = v;
        // moved dword px = (dY[2 px + 1], dY[2 px + 2]) = (this cell's upper half, the next cell's lower half): the neighbour's dword
        // comes over DPP (row_shl:1 inside the 16 lanes of a channel, zero behind the last one) and v_alignbit joins the halves.  (As
        // two ds_write_b16 into the neighbours' halves per dword, those twelve stores were 20 us of this kernel: 122 -> 102 us.)
        const unsigned vn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
        *reinterpret_cast<unsigned*>(a + wg::DYCOPYB) = __builtin_amdgcn_alignbit(vn, v, 16);
      }
    }
  };

  // TWO accumulator sets: the workgroup's bands alternate between +dY into `acc` and -dY into `accn`, and the result is acc - accn.
  // v_mfma_f32_16x16x32_bf16 floors its addends (the accumulator included) to 25 bits below the largest product of each 8-term step
  // (scripts/micro/mfma_bf16_accum.hip): every such loss points down, and a sum over the whole batch collects them; in the
  // difference of two sums that lose the same way they cancel (dW2 against float64, rms over the adversarial cases of
  // tests/split_cases.py: 1.0 - 1.7 x the fp32 kernel's error with one set, 0.9 - 1.1 x with two; same kernel time).
  f32x4_t acc[3][3], accn[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accn[i][j] = acc[i][j]; }
  // fragment bases of the wave's M-tiles mt = 3 mg + i = (tap, ci half): patch row 2 ph + ky, plane E for kx = 1, O otherwise
  int aoff[3]; bool moved[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int mt = 3 * mg + i, tap = mt >> 1, ky = tap / 3, kx = tap % 3;
    aoff[i] = (16 * (mt & 1) + lr) * wg::CISB + wg::prow(2 * ph + ky, kx == 1 ? 0 : 1, 0) + 16 * (lq ^ ((lr >> 2) & 3));
    moved[i] = kx == 0;
  }
  const int boff = lr * wg::COSB + 64 * ph + 16 * lq;

  const int ntiles = n_img * 16;
  int tile = c2::first_tile<16>(blockIdx.x, gridDim.x);
  float px[2][3];
  if (tile < ntiles) { c1_load(tile, row0, px[0]); c1_load(tile, row1, px[1]); cell_fetch(tile); }
  __syncthreads();                                   // the zero fill
  auto band = [&](f32x4_t (&ac)[3][3], bool negate) {
#ifdef MLHOT_TS
    const bool tsb = tf::g_ts_dev && blockIdx.x == 0 && lane == 0 && (tile - c2::first_tile<16>(0, gridDim.x)) / (int)gridDim.x == 6;
#define WGS_STAMP(i) do { if (tsb) tf::g_ts_dev[360 + wave * 5 + (i)] = clock64(); } while (0)
#else
#define WGS_STAMP(i) do { } while (0)
#endif
    WGS_STAMP(0);
    c1_tile(tile, row0, px[0]);
    if (wave < 8) c1_tile(tile, row0 + 3, px[1]);
    WGS_STAMP(1);
    cell_store(negate);
    WGS_STAMP(2);
    __syncthreads();
    WGS_STAMP(3);
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) { c1_load(next, row0, px[0]); c1_load(next, row1, px[1]); cell_fetch(next); }
    __builtin_amdgcn_sched_barrier(0);
    // B fragments (dY, 3 co tiles x 3 pieces) of the copy the M-tile needs, A fragments of the M-tile, 18 MFMAs; the tiles' kx
    // are wave-uniform, so `moved` is a scalar branch
    bf16x8_t bf[3][3];
    int have = -1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int want = moved[i] ? 1 : 0;
      if (want != have) {
        const unsigned char* bb = (want ? dys : dyp) + boff;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[j][p] = dg::dfrag(bb, 16 * j * wg::COSB + p * wg::DYPIECEB);
        have = want;
      }
      bf16x8_t af[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) af[p] = dg::dfrag(patch, aoff[i] + 64 * p);
      // products (a1 piece, dY piece): ll is dropped, lm and ml as well; the order keeps consecutive MFMAs on different accumulators
#pragma unroll
      for (int pp = 0; pp < 6; ++pp) {
        const int pa = pp == 0 ? 2 : (pp == 1 || pp == 2) ? 1 : 0;          // (l,h) (m,h) (m,m) (h,h) (h,m) (h,l)
        const int pb = pp == 0 ? 0 : pp == 1 ? 0 : pp == 2 ? 1 : pp - 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) ac[i][j] = mfma_bf16(af[pa], bf[j][pb], ac[i][j]);
      }
    }
    WGS_STAMP(4);
    __syncthreads();
  };
  while (tile < ntiles) {
    band(acc, false);
    tile += gridDim.x;
    if (tile >= ntiles) break;
    band(accn, true);
    tile += gridDim.x;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] -= accn[i][j];

  // fold the two rows' accumulators through LDS, one slab row per workgroup in accumulator order (c2::conv12_wgrad_kernel's epilogue)
  float* fl = reinterpret_cast<float*>(lds);
  if (ph == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) fl[((mg * 9 + i * 3 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (ph == 0) {
    float* sw = slab_w + (size_t)blockIdx.x * (COUT * CIN * 9 + COUT);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float4 v;
        v.x = acc[i][j][0] + fl[((mg * 9 + i * 3 + j) * 4 + 0) * 64 + lane];
        v.y = acc[i][j][1] + fl[((mg * 9 + i * 3 + j) * 4 + 1) * 64 + lane];
        v.z = acc[i][j][2] + fl[((mg * 9 + i * 3 + j) * 4 + 2) * 64 + lane];
        v.w = acc[i][j][3] + fl[((mg * 9 + i * 3 + j) * 4 + 3) * 64 + lane];
        *reinterpret_cast<float4*>(sw + ((size_t)((3 * mg + i) * 3 + j) * 64 + lane) * 4) = v;
      }
  }
  // bias gradient: a channel's 16 cells are 16 consecutive lanes
  float v = bsum;
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (cpx == 0) slab_b[(size_t)blockIdx.x * (COUT * CIN * 9 + COUT) + cco] = v;
}

}  // namespace c2s
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
