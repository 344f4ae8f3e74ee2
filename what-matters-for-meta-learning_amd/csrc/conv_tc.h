// conv2 of the vanilla encoder (32 -> 48 channels, 3x3 s2 p1, 64x64 -> 32x32; 81.5 % of the
// model's FLOPs) as three weight-stationary fp32-MFMA kernels: forward (+bias +ReLU +2x2 max-pool
// fused), data gradient (+pool/ReLU backward on the way in, +conv1-ReLU mask on the way out) and
// weight/bias gradient.  GPU build only; the generic igemm problems stay as the checked fallback
// for hostsim and as the A/B reference in the GPU tests.
//
// Shape of all three (MI355X-first, not a GEMM-library tiling):
//   * one persistent 768-thread workgroup per CU (12 waves = 3 per SIMD), looping over "bands":
//     one image x 4 conv2-output rows (= 8 conv1 rows);
//   * the STATIONARY operand lives in registers for the whole kernel (fwd / dgrad: the wave's
//     16-column slice of the weights, 72 / 108 VGPRs; wgrad: the 3x3 accumulator tiles), so the
//     MFMA loop reads ONE LDS dword per MFMA operand that changes;
//   * the band's activation patch is staged HBM -> registers -> LDS in channel planes whose
//     strides are chosen so that both 16-lane halves of every ds_read_b32 hit disjoint banks
//     (plane stride 585 = 9 mod 32 for stride-2 gathers, 176 = 16 mod 32 for unit-stride ones);
//     the next band's global loads are issued before the current band's MFMAs (T14 split);
//   * v_mfma_f32_16x16x4_f32 (exact fp32): A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15],
//     C/D lane l reg r = C[4*(l>>4)+r][l&15].
#pragma once
#include "common.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
#ifdef MLHOT_TS
namespace tf { extern __device__ long long* g_ts_dev; }
#endif
namespace c2 {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int CIN = 32, COUT = 48;
constexpr int NT = 768;
#ifndef C12_NT_STORE
#define C12_NT_STORE 0
#endif
#ifndef C12_M1_LANE0
#define C12_M1_LANE0 0
#endif
#ifndef C12_EARLY_PIXELS
#define C12_EARLY_PIXELS 0
#endif
// conv1-output patch of one band: [ci 32][row 9][col 65]; col c <-> ix = c - 1 (col 0 = zero pad),
// row r <-> iy = 8*band - 1 + r.
constexpr int RS = 65, ROWS = 9, PS = ROWS * RS;          // 585
constexpr int PATCH_FLOATS = CIN * PS;                     // 18720 floats = 74,880 B
// dY tile for wgrad: [pos 128][cout], row stride 50 (8 positions apart = 16 banks apart)
constexpr int DS = 50, DYT_FLOATS = 128 * DS;
// dY patch for dgrad: [row 5][col 33][co 48 + 2] (col 32 / row 4 = halo): the channel is the innermost index, so the twelve
// k-steps of a tap (co = 4 ks + lq) and the right-hand neighbour column sit within 100 words of ONE lane base - immediate
// offsets of a ds_read2_b32, no address arithmetic per step (as [co][row][col] planes of 176 words every step cost a v_add_u32 on
// the port the fp32 MFMAs issue through: 63 per band and wave).  Position stride 50 = 18 mod 32: the 16 columns of a 32-lane
// half land on the 16 even banks, lq = 1 on the odd ones.
constexpr int DRS = 33, DCS = 50, DYP_FLOATS = 5 * DRS * DCS;

__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// conv2's [48][288] weight matrix -> LDS rows of 289 words with coalesced float4 loads.  The kernels' register-resident slices
// are stride-9 / stride-288 gathers of it: read straight from global every wave load touches 16+ cache lines for 256 useful
// bytes (72 such loads per lane and 12 waves behind one texture addresser: ~4 us before the forward's first MFMA).
constexpr int W2_LD = CIN * 9 + 1;
template <int NTHREADS>
__device__ __forceinline__ void conv2w_stage(float* stage, const float* __restrict__ w, int tid) {
  // ALL of a thread's loads first, then the stores: as one load -> store loop (run-time trip count, not unrolled) every iteration
  // waited out an L2 round trip - the same loop over conv3's 110 KB was 9 k of that kernel's 54 k cycles
  constexpr int N4 = COUT * CIN * 9 / 4, CNT = (N4 + NTHREADS - 1) / NTHREADS;
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
      const int r = (4 * i) / (CIN * 9), c = 4 * i - r * (CIN * 9);        // 288 % 4 == 0: a float4 stays inside its row
      float* d = stage + r * W2_LD + c;
      d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
    }
  }
}

__device__ __forceinline__ void patch_zero_pad(float* patch, int tid) {
  for (int i = tid; i < CIN * ROWS; i += NT) patch[(i / ROWS) * PS + (i % ROWS) * RS] = 0.f;
}

// out[e] = sum_p slab[p][e]  (fixed order -> bitwise reproducible).  256 threads = 16 float4 column groups x
// 16 slab lanes; a lane walks its slabs 4 at a time (4 independent float4 loads in flight), then a
// fixed-order LDS fold of the 16 lanes.  grid = ceil(len / 64); len and the slab row stride must be multiples of 4.
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ slab, int nparts, int len, float* __restrict__ out, int stride) {
  __shared__ float4 sm[16][16];
  const int cx = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int e = (blockIdx.x * 16 + cx) * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  if (e < len) {
    int z = zl;
    for (; z + 48 < nparts; z += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(slab + (size_t)z * stride + e);
      const float4 v1 = *reinterpret_cast<const float4*>(slab + (size_t)(z + 16) * stride + e);
      const float4 v2 = *reinterpret_cast<const float4*>(slab + (size_t)(z + 32) * stride + e);
      const float4 v3 = *reinterpret_cast<const float4*>(slab + (size_t)(z + 48) * stride + e);
      add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; z < nparts; z += 16) add(s0, *reinterpret_cast<const float4*>(slab + (size_t)z * stride + e));
  }
  add(s0, s1); add(s2, s3); add(s0, s2);
  sm[zl][cx] = s0;
  __syncthreads();
  if (zl == 0 && e < len) {
    float4 t = sm[0][cx];
#pragma unroll
    for (int k = 1; k < 16; ++k) add(t, sm[k][cx]);
    *reinterpret_cast<float4*>(out + e) = t;
  }
}

// Up to 4 independent slab sums in ONE launch (the backward's deferred weight-gradient folds: conv3, conv2, conv1, the tail's
// per-task slabs): segment i owns blocks [first[i], first[i+1]) and is summed exactly as sum_parts_kernel would.
// kind 1: the slab holds conv2's weight gradient in the order the weight-gradient kernel's accumulators have it
// ([wave group mg][i][j][lane][r], coalesced float4 stores there); the fold writes element (mg, i, j, lane, r) to
// dW2[co = 16 j + lane % 16][ci = 16 (mt % 2) + 4 (lane / 16) + r][tap = mt / 2], mt = 3 mg + i.  kind 2: the same for conv3.
struct SumParts { const float* slab; float* out; int nparts, len, stride, kind; };
struct SumPartsMulti { SumParts seg[8]; int first[9]; int n; };
inline int sum_parts_blocks(int len) { return (len + 63) / 64; }
__global__ __launch_bounds__(256) void sum_parts_multi_kernel(const SumPartsMulti mp) {
  __shared__ float4 sm[16][16];
  int si = 0;
  while (si + 1 < mp.n && (int)blockIdx.x >= mp.first[si + 1]) ++si;
  const SumParts sp = mp.seg[si];
  const int cx = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const int e = (((int)blockIdx.x - mp.first[si]) * 16 + cx) * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  // (round 5: all 16 float4 of a lane requested before the first add instead of four trips of four - the same 0.5805 / 0.5814 /
  // 0.5815 ms per step as this form in a one-box A/B: the fold is bound by the rows' first touch, not by loads in flight)
  if (e < sp.len) {
    int z = zl;
    for (; z + 48 < sp.nparts; z += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(sp.slab + (size_t)z * sp.stride + e);
      const float4 v1 = *reinterpret_cast<const float4*>(sp.slab + (size_t)(z + 16) * sp.stride + e);
      const float4 v2 = *reinterpret_cast<const float4*>(sp.slab + (size_t)(z + 32) * sp.stride + e);
      const float4 v3 = *reinterpret_cast<const float4*>(sp.slab + (size_t)(z + 48) * sp.stride + e);
      add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; z < sp.nparts; z += 16) add(s0, *reinterpret_cast<const float4*>(sp.slab + (size_t)z * sp.stride + e));
  }
  add(s0, s1); add(s2, s3); add(s0, s2);
  sm[zl][cx] = s0;
  __syncthreads();
  if (zl == 0 && e < sp.len) {
    float4 t = sm[0][cx];
#pragma unroll
    for (int k = 1; k < 16; ++k) add(t, sm[k][cx]);
    if (sp.kind == 1) {
      const int lane = (e >> 2) & 63, tile = e >> 8, j = tile % 3, mt = tile / 3;      // tile = (3 mg + i) * 3 + j
      float* o = sp.out + ((size_t)(16 * j + (lane & 15)) * CIN + 16 * (mt & 1) + 4 * (lane >> 4)) * 9 + (mt >> 1);
      o[0] = t.x; o[9] = t.y; o[18] = t.z; o[27] = t.w;
    } else if (sp.kind == 2) {
      // conv3's weight gradient in ITS kernel's accumulator order (conv3_tc.h): tile = job * 18 + (kx * 3 + cig) * 2 + c,
      // job = (tg = job % 3: tap row, np = job / 3: output-channel tile pair)
      const int lane = (e >> 2) & 63, tile = e >> 8, job = tile / 18, rem = tile % 18, kx = rem / 6, cig = (rem >> 1) % 3, c = rem & 1;
      const int co = 16 * (2 * (job / 3) + c) + (lane & 15), ci = 16 * cig + 4 * (lane >> 4);
      float* o = sp.out + ((size_t)co * 48 + ci) * 9 + 3 * (job % 3) + kx;
      o[0] = t.x; o[9] = t.y; o[18] = t.z; o[27] = t.w;
    } else {
      *reinterpret_cast<float4*>(sp.out + e) = t;
    }
  }
}

// ==================================================================================================
// conv1-fused variants ("c12"): the conv1 output `a1` (512 KiB / image, the largest tensor of the
// model) is never written to HBM.  Each band's 32 x 9 x 64 slice of a1 is recomputed from the
// 19-row image strip (9.7 KB, L1/L2 resident) straight into the LDS patch as a small K = 12 GEMM
// on the matrix core; the forward keeps conv1's ReLU sign bits (16 KiB / image) for the backward,
// where the conv1 weight gradient is accumulated in registers right where d a1 is produced, so
// `d a1` (another 512 KiB / image) is never written either.  conv1 adds 6.8 % FLOPs and removes
// ~1.5 GB of HBM traffic per step at 480 images.
// ==================================================================================================

// wave-uniform issue priority 0..2 (s_setprio takes an immediate).  With every wave at priority 0 the OLDEST wave of a
// SIMD always wins the matrix pipe, finishes its band early and the youngest runs the tail alone; rotating the priority
// among the waves of a SIMD keeps them in step.
__device__ __forceinline__ void set_wave_prio(int pr) {
  if (pr == 0) __builtin_amdgcn_s_setprio(0);
  else if (pr == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(2);
}

// First band of workgroup g in the persistent conv12 kernels; a workgroup then walks tile += gridDim.x.  Tile = image * BPI + band
// (BPI bands per image).  With the full grid of 256 (one workgroup per CU, dispatched round-robin over the 8 XCDs: XCD = g % 8)
// the plain start tile = g sends band b of EVERY image to XCD b % 8, so the rows two bands share (the halo image rows, the halo
// pooled row of the data gradient, the other 3/4 of every 128-byte line of the byte-sized arg-max map) are fetched from HBM once
// per XCD that touches them: 157 MB for an algorithmic 92 MB in the data gradient (profiles/r03_final_pmc_traffic.json).  Here
// the 256 / BPI images of one sweep are dealt to the XCDs whole - all bands of an image run on the CUs of ONE XCD in the same
// sweep and meet in its L2; 256 tiles per sweep either way, so the walk itself is unchanged.
template <int BPI>
__device__ __forceinline__ int first_tile(int g, int grid) {
  if (grid != 256) return g;
  const int x = g & 7, h = g >> 3;                         // XCD, slot on it (0..31)
  return (8 * (h / BPI) + x) * BPI + h % BPI;              // image 8 (h / BPI) + x of the sweep, band h % BPI
}

struct ImgSrc {           // two-segment image batch (context | target), [n][1][128][128]
  const float* p0; int n0; const float* p1;
  __device__ __forceinline__ const float* img(int i) const { return i < n0 ? p0 + (size_t)i * 16384 : p1 + (size_t)(i - n0) * 16384; }
};

// ---- conv1 on the matrix core ----------------------------------------------------------------------
// conv1 of a band is a [576 positions] x [32 channels] x [K = 9 taps + bias] GEMM: 36 M-tiles of 16
// consecutive columns of one patch row, 3 per wave, 6 MFMAs each (K padded to 12).  The A operand
// (image pixels, lane = position x tap) comes straight from the L1/L2-resident strip, the B operand
// (w1 | b1) lives in 6 registers, and the accumulator lane layout (4 consecutive columns of one
// channel) writes into the [ci][row][col] patch directly.  ~12 % extra MFMA issue replaces ~300 VALU
// ops per thread and band and the scalar weight loads that drained the LDS prefetch queue.
struct Conv1W { float b[3][2]; };
struct Conv1A { float a[3][3]; };
__device__ __forceinline__ void conv1w_load(Conv1W& cw, const float* __restrict__ w1, const float* __restrict__ b1, int lr, int lq) {
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 4 * ks + lq, n = 16 * h + lr;
      cw.b[ks][h] = k < 9 ? w1[n * 9 + k] : (k == 9 ? b1[n] : 0.f);
    }
}
// A operand (k-step ks) of M-tile t (patch row t / 4, 16 columns from 16 * (t % 4)) of band `tile`: the image pixel under
// tap k = 4 ks + lq of position lr, 1.0 in the bias column k = 9, 0 in the padding.  A patch row above the image
// (band 0, row 0) gets an all-zero operand row, bias column included, so its a1 comes out as exactly 0 with no select
// behind the MFMAs.  The pixel is loaded unconditionally (index 0 when masked) and blended arithmetically
// (v * 1 + 0 | v * 0 + c): written as a select, hipcc sinks the load into an exec-masked branch behind an s_waitcnt vmcnt(0).
__device__ __forceinline__ float conv1a_pixel(const ImgSrc& x, int tile, int t, int ks, int lr, int lq) {
  const int img = tile >> 3, band = tile & 7;
  const float* xi = x.img(img);
  const int row = t >> 2, cg = t & 3;
  const int iy1 = 8 * band - 1 + row;
  const int k = 4 * ks + lq, ky = k / 3, kx = k - 3 * ky;
  const int iy = 2 * iy1 + ky - 1, ix = 2 * (16 * cg + lr) + kx - 1;
  const bool ok = k < 9 && iy1 >= 0 && iy >= 0 && ix >= 0;
  const float v = xi[ok ? iy * 128 + ix : 0];
  return __builtin_fmaf(v, ok ? 1.f : 0.f, (k == 9 && iy1 >= 0) ? 1.f : 0.f);
}
__device__ __forceinline__ void conv1a_fetch1(Conv1A& ca, int j, const ImgSrc& x, int tile, int wave, int lr, int lq) {
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) ca.a[j][ks] = conv1a_pixel(x, tile, wave + 12 * j, ks, lr, lq);
}
__device__ __forceinline__ void conv1a_fetch(Conv1A& ca, const ImgSrc& x, int tile, int wave, int lr, int lq) {
#pragma unroll
  for (int j = 0; j < 3; ++j) conv1a_fetch1(ca, j, x, tile, wave, lr, lq);
}

// The same operand with everything that does not change from band to band folded into three per-lane byte offsets
// (the forward's in-loop pixel requests: every VALU instruction beside fp32 MFMAs costs ~2.3 matrix-pipe cycles,
// scripts/micro/conv2_loop modes 7-10, so the ~20 instructions of conv1a_pixel's index arithmetic per request matter).
// Tile j of a wave is 3 patch rows below tile j - 1 in the same 16 columns, so request (j, ks) of band b sits at byte
// d[ks] + 4 (768 j + 2048 b) of the image; padding lanes (k > 9, the column left of the image) carry a large negative d,
// rows above the image come out negative by themselves, and a negative offset is the `masked` test.
struct Conv1Lane {
  int d[3];          // byte offset of tile 0's tap k = 4 ks + lq in band 0, or very negative
  float k9;          // 1.0 in the bias column's lanes (k = 9: ks = 2, lq = 1)
};
__device__ __forceinline__ void conv1lane_init(Conv1Lane& cl, int wave, int lr, int lq) {
  const int row0 = wave >> 2, cg = wave & 3;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int k = 4 * ks + lq, ky = k / 3, kx = k - 3 * ky;
    const int ix = 2 * (16 * cg + lr) + kx - 1;
    cl.d[ks] = (k < 9 && ix >= 0) ? 4 * ((2 * (row0 - 1) + ky - 1) * 128 + ix) : -(1 << 28);
  }
  cl.k9 = lq == 1 ? 1.f : 0.f;
}
__device__ __forceinline__ float conv1a_pixel_fast(const Conv1Lane& cl, const ImgSrc& x, int tile, int j, int ks, int row0) {
  const int band = tile & 7;
  const char* xi = reinterpret_cast<const char*>(x.img(tile >> 3));
  const int off = cl.d[ks] + 4 * (768 * j + 2048 * band);
  const float v = *reinterpret_cast<const float*>(xi + max(off, 0));
  const bool rowok = band > 0 || row0 + 3 * j > 0;              // wave-uniform: the patch row is inside the image
  return __builtin_fmaf(v, off >= 0 ? 1.f : 0.f, (ks == 2 && rowok) ? cl.k9 : 0.f);
}

// The same request cut in two for the forward's slot schedule: the load (2 VALU: offset, clamp) and - several k-steps later,
// when the value has had time to arrive - the blend.  As ONE function hipcc put the blend's v_fma right behind the load with an
// s_waitcnt vmcnt(0) in between: every one of the 9 pixel requests of a band stalled its wave for an L1 / L2 round trip inside
// the MFMA loop (the k-steps are fenced for the scheduler, so the consumer could not sink by itself).
__device__ __forceinline__ float conv1a_pixel_load(const Conv1Lane& cl, const ImgSrc& x, int tile, int j, int ks) {
  const char* xi = reinterpret_cast<const char*>(x.img(tile >> 3));
  const int off = cl.d[ks] + 4 * (768 * j + 2048 * (tile & 7));
  return *reinterpret_cast<const float*>(xi + max(off, 0));
}
__device__ __forceinline__ float conv1a_pixel_blend(float v, const Conv1Lane& cl, int tile, int j, int ks, int row0) {
  const int band = tile & 7;
  const int off = cl.d[ks] + 4 * (768 * j + 2048 * band);
  const bool rowok = band > 0 || row0 + 3 * j > 0;              // wave-uniform: the patch row is inside the image
  return __builtin_fmaf(v, off >= 0 ? 1.f : 0.f, (ks == 2 && rowok) ? cl.k9 : 0.f);
}

__device__ __forceinline__ void conv1a_fetch_fast(Conv1A& ca, const Conv1Lane& cl, const ImgSrc& x, int tile, int wave) {
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) ca.a[j][ks] = conv1a_pixel_fast(cl, x, tile, j, ks, wave >> 2);
}

// conv1's ReLU sign bits, as the forward's accumulator layout yields them (one v_cmp per accumulator register, no
// shuffling): record [img][a1 row][column group cg of 16] = 16 dwords; dword 4 r + 2 h + g, bit 16 e + c  <->  channel
// 16 h + c of column 16 cg + 4 (2 g + e) + r.  (v_cmp of register r of channel half h is a 64-bit lane mask whose bit
// 16 lq + lr is channel lr of column 4 lq + r: its two dwords are g = 0 / 1.)  The four dwords of a register go out as
// one 16-byte store of wave-uniform values from every lane - no exec masking, no cross-lane assembly (v_writelane needs
// inline asm, and any inline asm makes hipcc reserve a third of this kernel's 168 registers for AGPRs).  One junk
// record behind the last image takes the row above the image.
constexpr int M1_REC = 16;
__device__ __forceinline__ size_t m1_record(int img, int iy1, int cg, int n_img) {
  return iy1 >= 0 ? (((size_t)img * 64 + iy1) * 4 + cg) * M1_REC : (size_t)n_img * 4096;
}

// One conv1 M-tile in flight: 3 pixel loads -> 6 MFMAs -> ReLU + LDS store + sign bits, cut into slices that the
// forward places between its conv2 MFMAs (every slice is a handful of VALU / LDS instructions).
struct Conv1Tile {
  float a[3];
  f32x4_t c0, c1;
};
__device__ __forceinline__ void c1t_mfma(Conv1Tile& t, const Conv1W& cw, int i) {     // i = 0..5
  const int ks = i >> 1;
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  if (i & 1) t.c1 = mfma4(t.a[ks], cw.b[ks][1], ks == 0 ? zero : t.c1);
  else t.c0 = mfma4(t.a[ks], cw.b[ks][0], ks == 0 ? zero : t.c0);
}
// accumulator register r of both channel halves: ReLU, patch store, sign bits -> mrec (this tile's record; the halo row
// 0 of a band rewrites the record the band above writes, with the same bits)
template <bool MASK>
__device__ __forceinline__ void c1t_post(Conv1Tile& t, int r, float* d, unsigned* __restrict__ mrec) {
  if (!MASK) {      // no sign bits wanted: ReLU in ONE instruction, v_max_i32 on the bit pattern (a negative float is a negative
    // integer); compare + select is two, and fmaxf / v_med3_f32 come out as two as well (a canonicalising v_max in front)
    d[r] = __int_as_float(max(__float_as_int(t.c0[r]), 0));
    d[16 * PS + r] = __int_as_float(max(__float_as_int(t.c1[r]), 0));
    return;
  }
  const bool p0 = t.c0[r] > 0.f, p1 = t.c1[r] > 0.f;
  d[r] = p0 ? t.c0[r] : 0.f;
  d[16 * PS + r] = p1 ? t.c1[r] : 0.f;
  if (MASK) {
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(p0), b1 = __builtin_amdgcn_ballot_w64(p1);
    *reinterpret_cast<uint4*>(mrec + 4 * r) = make_uint4((unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32));
  }
}
// ---- the FORWARD's patch: [row 9][col 65][ci 32 + 2], the channel innermost ---------------------------------------------------
// (the weight-gradient kernel keeps the [ci][row][col] planes above: its A operand runs over ci along the lanes.)  The forward's A
// operand of k-step (tap, ci) is a1[ci][row + ky][2 x + kx]: with the channel innermost, every operand of a band sits at
// lane base + a compile-time byte offset below 64 KB - ds_read's immediate - where the [ci] planes of 585 words needed a fresh
// VGPR base per k-step (hipcc paired the two rows' reads into ds_read2_b32, 8-bit offsets: 40 v_add_u32 per band and wave on the
// port the fp32 MFMAs issue through).  One ds_read_b64 brings the operands of TWO consecutive k-steps (ci = 8 m + 2 lq + {0, 1}:
// the weights' registers are ordered to match).  Position stride 34 words: the 16 lanes of a tile sit 68 = 4 (mod 64) words apart
// and lq adds 2, so a 32-lane half of a ds_read_b64 covers all 64 banks once; conv1's stores (lane = channel, 4 lq columns
// = 136 = 8 mod 32 words apart) stay 2-way, which costs a ds_write_b32 nothing.
constexpr int CS = 34, CROW = RS * CS, CPATCH = ROWS * CROW;       // 19,890 floats = 79,560 B; two of them: 159,120 B
static_assert(2 * CPATCH * 4 <= 160 * 1024, "LDS");
__device__ __forceinline__ float* c1t_dst_cl(float* patch, int t, int lr, int lq) {
  return patch + ((t >> 2) * RS + 1 + 16 * (t & 3) + 4 * lq) * CS + lr;
}
template <bool MASK>
__device__ __forceinline__ void c1t_post_cl(Conv1Tile& t, int r, float* d, unsigned* __restrict__ mrec, bool lane0 = true) {
  const bool p0 = t.c0[r] > 0.f, p1 = t.c1[r] > 0.f;
  d[r * CS] = p0 ? t.c0[r] : 0.f;
  d[r * CS + 16] = p1 ? t.c1[r] : 0.f;
  if (MASK) {
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(p0), b1 = __builtin_amdgcn_ballot_w64(p1);
#if C12_M1_LANE0
    if (lane0)
#endif
    *reinterpret_cast<uint4*>(mrec + 4 * r) = make_uint4((unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32));
  }
}
__device__ __forceinline__ void patch_zero_pad_cl(float* patch, int tid) {      // column 0 (ix = -1) of every row
  for (int i = tid; i < ROWS * CS; i += NT) patch[(i / CS) * CROW + i % CS] = 0.f;
}

__device__ __forceinline__ float* c1t_dst(float* patch, int t, int lr, int lq) {
  return patch + lr * PS + (t >> 2) * RS + 1 + 16 * (t & 3) + 4 * lq;
}
__device__ __forceinline__ unsigned* c1t_rec(unsigned* __restrict__ m1, int tile, int tt, int n_img) {
  return m1 + m1_record(tile >> 3, 8 * (tile & 7) - 1 + (tt >> 2), tt & 3, n_img);
}
// M-tile j of this wave, start to end (prologue of the forward; the weight-gradient kernel)
template <bool MASK, bool CL = false>       // CL: into the forward's channel-innermost patch
__device__ __forceinline__ void conv1_tile(const Conv1A& ca, const Conv1W& cw, int j, float* patch, int tile, int wave, int lane,
                                           unsigned* __restrict__ m1, int n_img) {
  const int lr = lane & 15, lq = lane >> 4, tt = wave + 12 * j;
  Conv1Tile t;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) t.a[ks] = ca.a[j][ks];
#pragma unroll
  for (int i = 0; i < 6; ++i) c1t_mfma(t, cw, i);
  float* d = CL ? c1t_dst_cl(patch, tt, lr, lq) : c1t_dst(patch, tt, lr, lq);
  unsigned* mrec = MASK ? c1t_rec(m1, tile, tt, n_img) : nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (CL) c1t_post_cl<MASK>(t, r, d, mrec); else c1t_post<MASK>(t, r, d, mrec);
  }
}

// cache line li (< 76, clamped) of the 19-row image strip under band `tile`: rows 16 band - 3 .. 16 band + 15, 4 lines per row
__device__ __forceinline__ float strip_line(const ImgSrc& x, int tile, int li) {
  li = min(li, 75);
  const int row = min(max(16 * (tile & 7) - 3 + (li >> 2), 0), 127);
  return x.img(tile >> 3)[row * 128 + 32 * (li & 3)];
}

__global__ __launch_bounds__(NT) void conv12_fwd_pool_kernel(const ImgSrc x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ p2, uint8_t* __restrict__ amax,
                                                             unsigned* __restrict__ m1, int n_img, int dbg) {
  __shared__ __attribute__((aligned(16))) float patch2[2 * CPATCH];
  static_assert(COUT * W2_LD <= 2 * CPATCH, "weight staging area");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR arithmetic and scalar branches for everything derived from it
  const int nt = wave % 3, pg = wave / 3, rp = pg >> 1, ch = pg & 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int n = nt * 16 + lr;
  const int phase = __builtin_amdgcn_readfirstlane(wave >> 2);
#ifdef MLHOT_TS
  if (tf::g_ts_dev && blockIdx.x == 0 && tid == 0) tf::g_ts_dev[502] = clock64();
#endif

  conv2w_stage<NT>(patch2, w, tid);
  __syncthreads();
  float wr[72];
#pragma unroll
  for (int ks = 0; ks < 72; ++ks) wr[ks] = patch2[n * W2_LD + (8 * ((ks & 7) >> 1) + 2 * lq + (ks & 1)) * 9 + (ks >> 3)];      // k-step ks = (tap, pair m, j): ci = 8 m + 2 lq + j
  const float bn = bias[n];
  __syncthreads();
  Conv1W cw;
  conv1w_load(cw, w1, b1, lr, lq);
  Conv1Lane cl;
  conv1lane_init(cl, wave, lr, lq);

  patch_zero_pad_cl(patch2, tid);
  patch_zero_pad_cl(patch2 + CPATCH, tid);
  const int ntiles = n_img * 8;
  int tile = first_tile<8>(blockIdx.x, gridDim.x);
#ifdef MLHOT_TS
  long long ts_c0 = clock64(), ts_w0 = wall_clock64();
  if (tf::g_ts_dev && blockIdx.x == 0 && tid == 0) tf::g_ts_dev[503] = ts_c0;
#endif
  if (tile < ntiles) {
    Conv1A ca;
    conv1a_fetch(ca, x, tile, wave, lr, lq);
#pragma unroll
    for (int j = 0; j < 3; ++j) conv1_tile<true, true>(ca, cw, j, patch2, tile, wave, lane, m1, n_img);
  }
  __syncthreads();
  const int aoff = ((4 * rp) * RS + 2 * (16 * ch + lr)) * CS + 2 * lq;
  int cur = 0;
  // The pixels of a conv1 tile are asked for ~1.5 k cycles before its MFMAs: enough for an L1 / L2 hit, not for HBM (the
  // 8 bands of an image run on 8 workgroups at the same time, so the first touch of a strip is a miss for all of them:
  // measured: 162.3 us without, 158.9 us with this).  So two waves touch the 76 cache lines of the band after next (19 image rows
  // x 512 B) once the last pixel request of this band is out, 24 k-steps before the first request for that strip.
  float warm = 0.f;
  if (wave < 2 && tile + (int)gridDim.x < ntiles) warm = strip_line(x, tile + (int)gridDim.x, tid);
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const float* ab = patch2 + cur * CPATCH + aoff;
    float* nb = patch2 + (cur ^ 1) * CPATCH;
    // the next band's a1 slice (3 M-tiles per wave) is produced under this band's MFMAs.  The workgroup's last band
    // produces its own slice once more instead of branching around the slices (same values to the same sign-bit
    // records, an LDS buffer nobody reads): the band body stays one straight line.
    const int next = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;

#ifdef MLHOT_TS
    const bool tsb = tf::g_ts_dev && blockIdx.x == 0 && (tile - (int)blockIdx.x) / (int)gridDim.x == 6;   // 7th band of workgroup 0
    if (tsb && (lane == 0)) tf::g_ts_dev[400 + wave * 4 + 0] = clock64();
#endif
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // 72 k-steps of 2 MFMAs.  Their A operands run through a register ring RD k-steps deep (left alone, hipcc reads each
    // pair right before its MFMAs and waits lgkmcnt(0) on it: conv2 loop in isolation 131 -> 142 TF, scripts/micro/conv2_loop).
    // conv1 tile j of the next band rides along in fixed slots: pixels requested at k-steps 22j .. 22j+2, blended with their masks at
    // 22j+5 .. 22j+7 (conv1a_pixel_load / _blend), its 6 MFMAs one per k-step from 22j+8, ReLU / patch store / sign bits one accumulator register per k-step from 22j+15.  Every k-step is fenced for the scheduler, so each wave's VALU / LDS work sits in the shadow of its own MFMAs
    // and no wave ever leaves the matrix pipe for a long stretch.
    // operand pair p = (tap, m) = k-steps 2 p and 2 p + 1, for the tile's two conv2 rows (patch rows ky and ky + 2): one
    // ds_read_b64 each, offset = a compile-time constant; a ring RDP pairs (= 2 RDP k-steps) deep
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    constexpr int RDP = 2;
    auto aread = [&](int p, int row2) {
      const int tap = p >> 2, ky = tap / 3, kx = tap % 3, m = p & 3;
      return *reinterpret_cast<const f32x2_t*>(ab + (ky + row2) * CROW + kx * CS + 8 * m);
    };
    f32x2_t xa0[RDP], xa1[RDP];
#pragma unroll
    for (int d = 0; d < RDP; ++d) { xa0[d] = aread(d, 0); xa1[d] = aread(d, 2); }
    Conv1Tile ct;
    float bnw = bn;
#pragma unroll
    for (int ks = 0; ks < 72; ++ks) {
      // rotate the issue priority among the three waves of a SIMD (w, w + 4, w + 8): left at equal priority the oldest wave
      // wins the matrix pipe, finishes its band ~25 % early and idles at the barrier while the youngest runs the tail alone
      // (three turns per band measured best: 153 / 155 / 152 / 151 us for none / every 4 / 8 / 24 k-steps)
      if (ks % 24 == 0) set_wave_prio((phase + ks / 24) % 3);
      acc0 = mfma4(xa0[(ks >> 1) % RDP][ks & 1], wr[ks], acc0);
      acc1 = mfma4(xa1[(ks >> 1) % RDP][ks & 1], wr[ks], acc1);
      if ((ks & 1) && (ks >> 1) + RDP < 36) { xa0[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 0); xa1[(ks >> 1) % RDP] = aread((ks >> 1) + RDP, 2); }
      const int j = ks / 22, s = ks - 22 * j;          // j = 3 for ks >= 66: no slot
      if (j < 3) {
        const int tt = wave + 12 * j;
#if C12_EARLY_PIXELS
        // vmcnt retires in order: asked for at k-steps 22 j .. 22 j + 2, a tile's pixels queue BEHIND the previous tile's four sign-bit
        // record stores (k-steps 22 (j - 1) + 15 .. + 18), and the blend's wait for them is a wait for those stores' acknowledgements.
        // Tiles 1 and 2 therefore ask at 22 (j - 1) + 12 .. + 14 - in front of the stores, into operand registers the previous tile's
        // MFMAs have just released (a[0] after k-step 9, a[1] after 11, a[2] after 13) - and are 15 instead of 5 k-steps in flight.
        if (j == 0 && s < 3) ct.a[s] = conv1a_pixel_load(cl, x, next, 0, s);
        else if (j < 2 && s >= 12 && s < 15) ct.a[s - 12] = conv1a_pixel_load(cl, x, next, j + 1, s - 12);
#else
        if (s < 3) ct.a[s] = conv1a_pixel_load(cl, x, next, j, s);                           // raw pixel: in flight for 5 k-steps
#endif
        if (s >= 5 && s < 8) ct.a[s - 5] = conv1a_pixel_blend(ct.a[s - 5], cl, next, j, s - 5, wave >> 2);
        else if (s >= 8 && s < 14) c1t_mfma(ct, cw, s - 8);
#ifdef C12_KNOCK_M1      // timing experiment only (results WRONG for the backward): no sign-bit record stores
        else if (s >= 15 && s < 19) c1t_post_cl<false>(ct, s - 15, c1t_dst_cl(nb, tt, lr, lq), nullptr);
#else
        else if (s >= 15 && s < 19) c1t_post_cl<true>(ct, s - 15, c1t_dst_cl(nb, tt, lr, lq), c1t_rec(m1, next, tt, n_img), lane == 0);
#endif
      }
      // the warmed line has to be consumed somewhere or the load is dead code: it rides into the epilogue's bias as + 0 * pixel,
      // at the k-step where the wait for it costs nothing (inline asm would do, but see m1_record's note on AGPRs)
      if (ks == 7) bnw = __builtin_fmaf(warm, 0.f, bn);
      if (ks == 48 && wave < 2) warm = strip_line(x, next + (int)gridDim.x < ntiles ? next + (int)gridDim.x : next, tid);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef MLHOT_TS
    if (tsb && (lane == 0)) tf::g_ts_dev[400 + wave * 4 + 1] = clock64();
#endif
    // (Measured and dropped, round 4: this epilogue - bias, ReLU, 2 x 2 pool, two stores - kept as pending accumulators and run in
    // k-steps 66 .. 68 of the NEXT band, where the slot schedule has no conv1 work: 141.4 -> 141.2 us.  Vector instructions cost
    // the same issue slots under the MFMAs as behind them.)
    const int img = tile >> 3, band = tile & 7;
    float pv[2]; unsigned pa[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float c0 = fmaxf(acc0[2 * j] + bnw, 0.f), c1 = fmaxf(acc0[2 * j + 1] + bnw, 0.f);
      const float c2v = fmaxf(acc1[2 * j] + bnw, 0.f), c3 = fmaxf(acc1[2 * j + 1] + bnw, 0.f);
      float best = c0; unsigned which = 0;
      if (c1 > best) { best = c1; which = 1; }
      if (c2v > best) { best = c2v; which = 2; }
      if (c3 > best) { best = c3; which = 3; }
      pv[j] = best; pa[j] = which;
    }
    const size_t o = (((size_t)img * COUT + n) * 16 + 2 * band + rp) * 16 + 8 * ch + 2 * lq;
#if C12_NT_STORE
    // non-temporal: the pooled map and the arg-max bytes are read next by OTHER kernels (conv3 / the backward), on whatever XCD their
    // workgroups land - kept as dirty lines in this XCD's L2 they are written back at the kernel boundary, in front of conv3
    // (23.6 + 5.9 MB per c3 forward; MI355X_MICROARCH.md's "+ B / 6 TB/s" boundary term); streamed out they drain under the bands
    __builtin_nontemporal_store(f32x2_t{pv[0], pv[1]}, reinterpret_cast<f32x2_t*>(p2 + o));
    __builtin_nontemporal_store((unsigned short)(pa[0] | (pa[1] << 8)), reinterpret_cast<unsigned short*>(amax + o));
#else
    *reinterpret_cast<float2*>(p2 + o) = make_float2(pv[0], pv[1]);
    *reinterpret_cast<unsigned short*>(amax + o) = (unsigned short)(pa[0] | (pa[1] << 8));
#endif
#ifdef MLHOT_TS
    if (tsb && (lane == 0)) tf::g_ts_dev[400 + wave * 4 + 2] = clock64();
#endif
    __syncthreads();
#ifdef MLHOT_TS
    if (tsb && (lane == 0)) tf::g_ts_dev[400 + wave * 4 + 3] = clock64();
#endif
  }
#ifdef MLHOT_TS
  if (tf::g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0) {       // shader cycles and 100 MHz ticks of the main loop -> effective clock
    tf::g_ts_dev[500] = clock64() - ts_c0; tf::g_ts_dev[501] = wall_clock64() - ts_w0;
  }
#endif
}

// ---- weight + bias gradient of conv2 with the a1 patch recomputed from the image ------------------
// Pooled cell e (< 1536 = 48 co x 2 pooled rows x 16 px) of a band -> (px, pooled row, co), two per thread.  A 32-lane half of
// a wave is 8 px x 4 co: its four un-pooled stores into the dY tile ([pos][co], row stride DS = 50, a pooled px = 2 positions
// = 100 = 4 mod 32 words apart) land on 8 x 4 = 32 different banks.  (px fastest over 16, then the row, then co - the order
// of the global reads - put 4 lanes on every bank it touched: 2/3 of this kernel's bank-conflict cycles.)
__device__ __forceinline__ int wg_px(int e) { return (e & 7) + ((e >> 2) & 8); }
__device__ __forceinline__ int wg_pyl(int e) { return (e >> 6) & 1; }
__device__ __forceinline__ int wg_co(int e) { return ((e >> 3) & 3) + 4 * (e >> 7); }
__global__ __launch_bounds__(NT) void conv12_wgrad_kernel(const ImgSrc x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ dp2, const float* __restrict__ p2,
                                                          const uint8_t* __restrict__ amax, float* __restrict__ slab_w,
                                                          float* __restrict__ slab_b, int n_img, int dbg) {
  __shared__ float lds[PATCH_FLOATS + DYT_FLOATS];
  float* patch = lds;
  float* dyt = lds + PATCH_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR arithmetic and scalar branches for everything derived from it
  const int mg = wave % 6, ph = wave / 6;
  const int lr = lane & 15, lq = lane >> 4;
  const bool cact = !(dbg & 1);
  Conv1W cw;
  conv1w_load(cw, w1, b1, lr, lq);

  f32x4_t acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};

  patch_zero_pad(patch, tid);
  const int ntiles = n_img * 8;
  Conv1A ca;
  float cdp[2], cp[2]; unsigned cam[2];
  auto cells_fetch = [&](int t) {
    const int img = t >> 3, band = t & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * NT, px = wg_px(e), pyl = wg_pyl(e), co = wg_co(e);
      const size_t o = (((size_t)img * COUT + co) * 16 + 2 * band + pyl) * 16 + px;
      cdp[j] = dp2[o]; cp[j] = p2[o]; cam[j] = amax[o];
    }
  };
  int tile = first_tile<8>(blockIdx.x, gridDim.x);
  Conv1Lane cl;
  conv1lane_init(cl, wave, lr, lq);
  if (tile < ntiles) { if (cact) conv1a_fetch_fast(ca, cl, x, tile, wave); cells_fetch(tile); }
  int aoff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int mt = 3 * mg + i, tap = mt >> 1, ky = tap / 3, kx = tap % 3;
    aoff[i] = (16 * (mt & 1) + lr) * PS + ky * RS + kx + 2 * (8 * lq);
  }
  const int boff = (8 * lq) * DS + lr;

  for (; tile < ntiles; tile += gridDim.x) {
#ifdef MLHOT_TS
    const bool tsb = tf::g_ts_dev && blockIdx.x == 0 && lane == 0 && (tile - first_tile<8>(0, gridDim.x)) / (int)gridDim.x == 6;
#define WG_STAMP(i) do { if (tsb) tf::g_ts_dev[360 + wave * 5 + (i)] = clock64(); } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif
    WG_STAMP(0);
    __syncthreads();
    WG_STAMP(1);
    if (cact) {
#pragma unroll
      for (int j = 0; j < 3; ++j) conv1_tile<false>(ca, cw, j, patch, tile, wave, lane, nullptr, n_img);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * NT, px = wg_px(e), pyl = wg_pyl(e), co = wg_co(e);
      const float g = cp[j] > 0.f ? cdp[j] : 0.f;
      bsum[j] += g;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dyt[((2 * pyl + (q >> 1)) * 32 + 2 * px + (q & 1)) * DS + co] = (cam[j] == (unsigned)q) ? g : 0.f;
    }
    WG_STAMP(2);
    __syncthreads();
    WG_STAMP(3);
    if (tile + (int)gridDim.x < ntiles) { if (cact) conv1a_fetch_fast(ca, cl, x, tile + gridDim.x, wave); cells_fetch(tile + gridDim.x); }

    // 16 k-steps of 9 MFMAs (an explicit operand ring and the forward's priority rotation were measured here: 152.1 us as is,
    // 153.5 / 153.4 / 155.8 us with ring / rotation / both - this loop already issues one LDS read per three MFMAs)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int orow_l = s >> 3, oxb = s & 7;
      float a[3], b[3];
      const int arow = (2 * (2 * ph + orow_l)) * RS + 2 * oxb;
#pragma unroll
      for (int i = 0; i < 3; ++i) a[i] = patch[aoff[i] + arow];
      const int brow = ((2 * ph + orow_l) * 32 + oxb) * DS;
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j] = dyt[boff + brow + 16 * j];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
    }
    WG_STAMP(4);
  }

  // fold the two position halves through LDS (the patch is free now): one slab per workgroup
  __syncthreads();
  if (ph == 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[((mg * 9 + i * 3 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (ph == 0) {
    float* sw = slab_w + (size_t)blockIdx.x * (COUT * CIN * 9 + COUT);      // slab row = [weights | bias]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int mt = 3 * mg + i, tap = mt >> 1;
      (void)mt; (void)tap;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        // accumulator order, one coalesced 16-byte store per tile and lane; the final fold un-permutes (SumParts kind 1)
        float4 v;
        v.x = acc[i][j][0] + lds[((mg * 9 + i * 3 + j) * 4 + 0) * 64 + lane];
        v.y = acc[i][j][1] + lds[((mg * 9 + i * 3 + j) * 4 + 1) * 64 + lane];
        v.z = acc[i][j][2] + lds[((mg * 9 + i * 3 + j) * 4 + 2) * 64 + lane];
        v.w = acc[i][j][3] + lds[((mg * 9 + i * 3 + j) * 4 + 3) * 64 + lane];
        *reinterpret_cast<float4*>(sw + ((size_t)((3 * mg + i) * 3 + j) * 64 + lane) * 4) = v;
      }
    }
  }
  // bias gradient: a channel's 32 cells sit in 16 lanes (bits 0-2, 5) of two waves (the pooled row); fixed-order fold
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float v = bsum[j];
    v += __shfl_xor(v, 32, 64);
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int e = tid + j * NT;
    if ((lane & 39) == 0) lds[wg_pyl(e) * COUT + wg_co(e)] = v;
  }
  __syncthreads();
  if (tid < COUT) slab_b[(size_t)blockIdx.x * (COUT * CIN * 9 + COUT) + tid] = lds[tid] + lds[COUT + tid];
}

// (Measured and removed: the same kernel pipelined - units of HALF a band (2 conv2 rows, 5 a1 rows, one pooled dY row), patch + dY
// tile double buffered in 109 KB, the next unit's conv1 tiles and dY cell produced in slots between this unit's MFMAs, one barrier
// per unit.  Stamps of the kernel above had shown 4.1 k of a band's 20.5 k cycles in the staging phase with the matrix pipe all
// but idle.  Result: 136.3 -> 141.0 us.  The staging phase is not idle time that MFMAs could fill: its conv1 MFMAs, selects,
// LDS stores and loads issue through the same port as the main MFMAs (the issue-bound model of DESIGN.md section 4), so moving
// them under the MFMAs moves nothing - and half bands recompute the halo row twice and pay twice the barriers.)

// ---- data gradient of conv2 + conv1 ReLU mask + conv1 weight/bias gradient ------------------------
// 512 threads (8 waves, 2 per SIMD): wave w owns input channels 16*(w&1)..+15 and the band's a1 rows
// {2pg, 2pg+1} (pg = w>>1): 72 + 144 = 216 MFMAs per band for every wave.  After a row's two parity
// tiles the lane holds d a1 for 8 consecutive x of one channel: it masks them with the forward's
// ReLU bits and feeds the two accumulators STRAIGHT BACK into the matrix core as the A operand of
// conv1's weight gradient  Z[ci][tap] += sum_pos g[pos][ci] * T[pos][tap]  (an accumulator tile is a
// valid operand for a product that sums over its row index, cdna guide §3): 8 MFMAs + 8 LDS reads
// per row-half instead of ~150 VALU FMAs.  T comes from the image strip staged in LDS; tap column 9
// is all ones and yields the bias gradient.  The dY patch and the strip are double-buffered, so
// there is one barrier per band.
constexpr int NT2 = 512;
constexpr int SRS = 132, STRIP_FLOATS = 17 * SRS;     // image strip rows 16b-1 .. 16b+15, col c <-> ix = c - 1
constexpr int D12_BUF = DYP_FLOATS + STRIP_FLOATS;

// One "row half" = 8 consecutive x of one a1 row and channel per lane (two 16 x 16 parity tiles, accumulators e | d), then the
// conv1 weight-gradient MFMAs that consume them.  A wave's stream used to run half by half - operand reads, 36 / 72 conv MFMAs,
// wait for the accumulators, 16 selects, 8 tap reads, 8 MFMAs - and a wave alone on its SIMD filled only 55-60 % of the matrix
// pipe (stamps with the partner wave parked), the pair 70 %.  Now a half hands its accumulators on as PENDING: the masks and the
// 8 weight-gradient MFMAs of half h run between the MFMA steps of half h + 1, whose chains they do not touch; only the last half
// of a band is flushed on its own, in front of the band barrier.
// pooled cell e (< 2304 = 48 co x 3 pooled rows x 16 px) of a band: a 32-lane half is 8 px x 4 co, whose un-pooled stores into
// the [position][co] patch (a pooled px = 2 positions = 100 = 4 mod 32 words apart) land on 32 different banks
__device__ __forceinline__ int d12_px(int e) { return (e & 7) + ((e >> 2) & 8); }
__device__ __forceinline__ int d12_pyl(int e) { return (e >> 6) % 3; }
__device__ __forceinline__ int d12_co(int e) { return ((e >> 3) & 3) + 4 * ((e >> 6) / 3); }

struct D12Pend {
  f32x4_t e, d;      // d a1 before the ReLU mask: x = x0 + 2 r (e) and x0 + 2 r + 1 (d)
  uint4 mb;          // conv1 ReLU bits of those positions
  int tpoff;         // strip offset of the half's image taps (floats)
};
struct D12Lane { int ci, lr, lq, toff; float tmask, tconst; unsigned pos_lo, pos_hi; };      // tap column lr: image tap | ones (bias) | zero

// mask register r of a pending half and feed both accumulators to the matrix core: Z += g^T T.
// MFMA r of parity px has A = g[row 4lq+r][ci] (this lane's register) and needs
// B[k = lq][n = tap lr] = T[position x' = 16xh + 4lq + r][tap] = strip[(2yl+ky)][2x + kx], x = 2x' + px.
__device__ __forceinline__ void d12_pend_step(const D12Pend& p, int r, float t0, float t1, const D12Lane& ln, f32x4_t& z, f32x4_t& z2) {
  // bit (ci % 16) + 16 (r / 2) of the record dword, sign-extended to a 0 / ~0 word, ANDed onto the accumulator: v_bfe_i32 + v_and
  // (and / compare / select is one vector instruction more per value, 32 per band: 145.2 -> 142.1 us)
  const unsigned pos = r >= 2 ? ln.pos_hi : ln.pos_lo;
  const unsigned me = (r & 1) ? p.mb.z : p.mb.x, md = (r & 1) ? p.mb.w : p.mb.y;
  const float ge = __uint_as_float(__float_as_uint(p.e[r]) & (unsigned)__builtin_amdgcn_sbfe((int)me, pos, 1u));
  const float gd = __uint_as_float(__float_as_uint(p.d[r]) & (unsigned)__builtin_amdgcn_sbfe((int)md, pos, 1u));
  z = mfma4(ge, __builtin_fmaf(t0, ln.tmask, ln.tconst), z);
  z2 = mfma4(gd, __builtin_fmaf(t1, ln.tmask, ln.tconst), z2);
}
// (the taps are read unconditionally - tap columns lr > 8 have toff = 0 - and blended arithmetically: written as the select
// `lr < 9 ? tp[..] : c`, hipcc wraps EACH read in an exec-masked branch with an s_waitcnt lgkmcnt(0) of its own - eight serial
// LDS round trips, ~1 k cycles per half, during which the wave issues nothing)
__device__ __forceinline__ void d12_pend_taps(const float* strip, const D12Pend& p, float (&t0)[4], float (&t1)[4]) {
  const float* tp = strip + p.tpoff;
#pragma unroll
  for (int r = 0; r < 4; ++r) { t0[r] = tp[4 * r]; t1[r] = tp[4 * r + 2]; }
}
__device__ __forceinline__ void d12_flush(const float* strip, const D12Pend& p, const D12Lane& ln, f32x4_t& z, f32x4_t& z2) {
  float t0[4], t1[4];
  d12_pend_taps(strip, p, t0, t1);
#pragma unroll
  for (int r = 0; r < 4; ++r) d12_pend_step(p, r, t0[r], t1[r], ln, z, z2);
}

template <int PY, bool HAS_PREV>
__device__ __forceinline__ void dgrad12_half(const float* dyp, const float* strip, const float (&wr)[9][12], const unsigned* __restrict__ mrow,
                                             f32x4_t& z, f32x4_t& z2, int yl, int xh, const D12Lane& ln, const D12Pend& prev, D12Pend& out) {
  const int lr = ln.lr, lq = ln.lq, yp = yl >> 1;
  // conv1 ReLU bits of the 8 positions this lane will hold (x0 .. x0+7, x0 = 32xh + 8lq): column group 2xh + lq/2,
  // dwords 4 r' + 2 (ci / 16) + lq % 2 of its record, bit 16 e + ci % 16 for column offset 8 (lq % 2) + 4 e + r'
  const unsigned* mrec = mrow + (2 * xh + (lq >> 1)) * M1_REC + 2 * (ln.ci >> 4) + (lq & 1);
  uint4 mb;
  mb.x = mrec[0]; mb.y = mrec[4]; mb.z = mrec[8]; mb.w = mrec[12];
  float t0[4], t1[4];
  if (HAS_PREV) d12_pend_taps(strip, prev, t0, t1);
  f32x4_t& e = out.e; f32x4_t& d = out.d;          // accumulate where the next half will look for them
  e = f32x4_t{0.f, 0.f, 0.f, 0.f}; d = e;
  const float* base = dyp + (yp * DRS + 16 * xh + lr) * DCS + lq;
  // Step st = (tap row ty, k-step ks): TWO operand words (dY at the column and at its right neighbour, one ds_read2_b32) feed
  // THREE MFMAs - px = 1 / kx = 0 on the neighbour, px = 0 / kx = 1 and px = 1 / kx = 2 on the column itself - issued d, e, d:
  // the two accumulators alternate, so no MFMA waits out the 40-cycle latency of the one before it.  The operands run two steps
  // ahead in a ring; every step is fenced for the scheduler (unfenced, hipcc hoists all reads of a half to its top - 408
  // registers - or sinks each one to its use, behind an s_waitcnt lgkmcnt(0)).
  constexpr int NS = PY ? 24 : 12, RING = 3;
  float a0[RING], a1[RING];
  auto opnd = [&](int st, int slot) {
    const int ty = st / 12, ks = st % 12, doy = (PY && ty == 0) ? 1 : 0;
    a0[slot] = base[doy * DRS * DCS + 4 * ks];
    a1[slot] = base[doy * DRS * DCS + 4 * ks + DCS];
  };
#pragma unroll
  for (int st = 0; st < RING - 1; ++st) opnd(st, st);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + RING - 1 < NS) opnd(st + RING - 1, (st + RING - 1) % RING);
    __builtin_amdgcn_sched_barrier(0);
    const int ty = st / 12, ks = st % 12, ky = PY ? (ty == 0 ? 0 : 2) : 1, sl = st % RING;
    d = mfma4(a1[sl], wr[ky * 3 + 0][ks], d);
    e = mfma4(a0[sl], wr[ky * 3 + 1][ks], e);
    d = mfma4(a0[sl], wr[ky * 3 + 2][ks], d);
    if (HAS_PREV && st >= 2 && st < 6) d12_pend_step(prev, st - 2, t0[st - 2], t1[st - 2], ln, z, z2);
    __builtin_amdgcn_sched_barrier(0);
  }
  out.mb = mb;
  out.tpoff = (2 * yl) * SRS + ln.toff + 4 * (16 * xh + 4 * lq);      // + 4r + 2px
}

__global__ __launch_bounds__(NT2) void conv12_dgrad_kernel(const ImgSrc x, const unsigned* __restrict__ m1,
                                                           const float* __restrict__ dp2, const float* __restrict__ p2,
                                                           const uint8_t* __restrict__ amax, const float* __restrict__ w,
                                                           float* __restrict__ slab1, int n_img) {
  __shared__ float lds[2 * D12_BUF + 8 * 16 * 16];
  float* red = lds + 2 * D12_BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: SGPR arithmetic and scalar branches for everything derived from it
  const int nt = wave & 1, pg = wave >> 1;
  const int lr = lane & 15, lq = lane >> 4;
  const int ci = 16 * nt + lr;

#ifdef MLHOT_TS
  if (tf::g_ts_dev && blockIdx.x == 0 && tid == 0) tf::g_ts_dev[504] = clock64();
#endif
  // (staged through LDS like the forward's, these loads measured 1 us slower: along ci the gather is a 36-byte stride, ~5 lines per load)
  float wr[9][12];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) wr[tap][ks] = w[((size_t)(4 * ks + lq) * CIN + ci) * 9 + tap];
  // this lane's tap (as B-operand column lr of the conv1 weight gradient): strip offset ky*SRS + kx
  const int toff = lr < 9 ? (lr / 3) * SRS + lr % 3 : 0;
  const D12Lane ln{ci, lr, lq, toff, lr < 9 ? 1.f : 0.f, lr == 9 ? 1.f : 0.f, (unsigned)(ci & 15), (unsigned)(ci & 15) + 16u};
  f32x4_t z = {0.f, 0.f, 0.f, 0.f}, z2 = z;  // Z[ci = 16nt + 4lq + r][tap lr], summed over this wave's positions (even | odd x: two chains)

  for (int bsel = 0; bsel < 2; ++bsel) {
    float* dypb = lds + bsel * D12_BUF;
    float* stripb = dypb + DYP_FLOATS;
    for (int i = tid; i < 5 * DCS; i += NT2) dypb[((i / DCS) * DRS + 32) * DCS + i % DCS] = 0.f;   // halo column
    for (int i = tid; i < 17; i += NT2) stripb[i * SRS] = 0.f;                                   // ix = -1 column
  }
  const int ntiles = n_img * 8;
  float cdp[5], cp[5]; unsigned cam[5];
  float4 sv[2];
  // (Measured and dropped: cells dealt so that a thread's five requests differ by wave-uniform constants only - no division by
  // 48 / modulo / 64-bit address per cell, ~60 vector instructions per band less - with plain loads +0.5 %, with raw buffer
  // loads (scalar base + 32-bit lane offset, out-of-range lanes read 0, no branches, 16 registers less) +2 %: a buffer_load
  // costs more issue time beside fp32 MFMAs than the address arithmetic it saves.)
  auto cells_fetch = [&](int t) {       // 48 co x 3 pooled rows x 16 px = 2304 cells, <= 5 per thread
    const int img = t >> 3, band = t & 7;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = tid + j * NT2, px = d12_px(e), pyl = d12_pyl(e), co = d12_co(e);
      const int py = 2 * band + pyl;
      cdp[j] = 0.f; cp[j] = 0.f; cam[j] = 0;
      if (e < 2304 && py < 16) {
        const size_t o = (((size_t)img * COUT + co) * 16 + py) * 16 + px;
        cdp[j] = dp2[o]; cp[j] = p2[o]; cam[j] = amax[o];
      }
    }
    const float* xi = x.img(img);       // image strip: 17 rows x 32 float4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * NT2, row = e >> 5, x4 = e & 31;
      const int iy = 16 * band - 1 + row;
      sv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < 17 * 32 && iy >= 0) sv[j] = *reinterpret_cast<const float4*>(xi + iy * 128 + 4 * x4);
    }
  };
  auto cells_store = [&](float* dypb, float* stripb) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int e = tid + j * NT2, px = d12_px(e), pyl = d12_pyl(e), co = d12_co(e);
      if (e < 2304) {
        const float g = cp[j] > 0.f ? cdp[j] : 0.f;
        float* d = dypb + ((2 * pyl) * DRS + 2 * px) * DCS + co;
        d[0] = cam[j] == 0u ? g : 0.f;
        d[DCS] = cam[j] == 1u ? g : 0.f;
        if (pyl < 2) {
          d[DRS * DCS] = cam[j] == 2u ? g : 0.f;
          d[DRS * DCS + DCS] = cam[j] == 3u ? g : 0.f;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = tid + j * NT2, row = e >> 5, x4 = e & 31;
      if (e < 17 * 32) {
        float* d = stripb + row * SRS + 1 + 4 * x4;
        d[0] = sv[j].x; d[1] = sv[j].y; d[2] = sv[j].z; d[3] = sv[j].w;
      }
    }
  };
  int tile = first_tile<8>(blockIdx.x, gridDim.x);
  __syncthreads();
  if (tile < ntiles) { cells_fetch(tile); cells_store(lds, lds + DYP_FLOATS); }
  if (tile + (int)gridDim.x < ntiles) cells_fetch(tile + gridDim.x);
  __syncthreads();
#ifdef MLHOT_TS
  if (tf::g_ts_dev && blockIdx.x == 0 && tid == 0) tf::g_ts_dev[505] = clock64();
#endif
  int cur = 0;
  for (; tile < ntiles; tile += gridDim.x, cur ^= 1) {
    const float* dyp = lds + cur * D12_BUF;
    const float* strip = dyp + DYP_FLOATS;
    const int img = tile >> 3, band = tile & 7;
    const unsigned* mrow = m1 + ((size_t)img * 64 + 8 * band + 2 * pg) * 64;
#ifdef MLHOT_TS
    const bool tsb = tf::g_ts_dev && blockIdx.x == 0 && lane == 0 && (tile - first_tile<8>(0, gridDim.x)) / (int)gridDim.x == 6;   // 7th band of workgroup 0
#define D12_STAMP(i) do { if (tsb) tf::g_ts_dev[448 + wave * 6 + (i)] = clock64(); } while (0)
#else
#define D12_STAMP(i) do { } while (0)
#endif
    D12_STAMP(0);
    D12Pend pa, pb;
    dgrad12_half<0, false>(dyp, strip, wr, mrow, z, z2, 2 * pg, 0, ln, pb, pa);
    dgrad12_half<0, true>(dyp, strip, wr, mrow, z, z2, 2 * pg, 1, ln, pa, pb);
    D12_STAMP(1);
    // stage the next band into the idle buffers between the two rows; prefetch the one after it.  (The two waves of a SIMD
    // staging at different points of the band - one here, one in the middle of the second row - measured 1 % better and was
    // dropped for the pending-half pipeline; s_setprio 1 for the younger half of the waves: neutral.)
    const int next = tile + (int)gridDim.x;
    if (next < ntiles) {
      cells_store(lds + (cur ^ 1) * D12_BUF, lds + (cur ^ 1) * D12_BUF + DYP_FLOATS);
      if (next + (int)gridDim.x < ntiles) cells_fetch(next + gridDim.x);
    }
    D12_STAMP(2);
    dgrad12_half<1, true>(dyp, strip, wr, mrow + 64, z, z2, 2 * pg + 1, 0, ln, pb, pa);
    dgrad12_half<1, true>(dyp, strip, wr, mrow + 64, z, z2, 2 * pg + 1, 1, ln, pa, pb);
    d12_flush(strip, pb, ln, z, z2);
    D12_STAMP(3);
    __syncthreads();
    D12_STAMP(4);
  }
#ifdef MLHOT_TS
  if (tf::g_ts_dev && blockIdx.x == 0 && tid == 0) tf::g_ts_dev[506] = clock64();
#endif
  // conv1 gradient partials: fold the 4 row-group waves of each channel half, write [block][ci][16 tap columns]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * lq + r) * 16 + lr] = z[r] + z2[r];
  __syncthreads();
  if (tid < 512) {
    const int c = tid >> 4, q = tid & 15, ntc = c >> 4, cl = c & 15;     // channel c, tap column q
    float sacc = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) sacc += red[((2 * g4 + ntc) * 16 + cl) * 16 + q];
    if (q < 10) slab1[(size_t)blockIdx.x * 320 + (q < 9 ? c * 9 + q : 288 + c)] = sacc;     // [block][32 x 9 weights | 32 biases]
  }
}

// slab1 [nblocks][288 + 32] -> dw1 [32][9], db1 [32] (used when the two gradient tensors are not adjacent; otherwise the
// row is summed as one plain segment).  One workgroup per 20 outputs: 16 slab lanes x 20 columns...
// grid = 16 workgroups of 320 threads: thread = (slab lane z = tid / 20, output e = 20 * blockIdx + tid % 20); the 16
// lanes are folded through LDS in a fixed order (the single-block version walked all slabs serially: 18 us).
__global__ __launch_bounds__(320) void conv1_grads_kernel(const float* __restrict__ slab1, int nblocks, float* __restrict__ dw1, float* __restrict__ db1) {
  __shared__ float sm[16][20];
  const int tid = threadIdx.x, z = tid / 20, col = tid % 20, e = blockIdx.x * 20 + col;
  float s0 = 0.f, s1 = 0.f;
  int b = z;
  for (; b + 16 < nblocks; b += 32) { s0 += slab1[(size_t)b * 320 + e]; s1 += slab1[(size_t)(b + 16) * 320 + e]; }
  for (; b < nblocks; b += 16) s0 += slab1[(size_t)b * 320 + e];
  sm[z][col] = s0 + s1;
  __syncthreads();
  if (z == 0) {
    float v = sm[0][col];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += sm[k][col];
    if (e < 288) dw1[e] = v; else db1[e - 288] = v;
  }
}

}  // namespace c2
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
