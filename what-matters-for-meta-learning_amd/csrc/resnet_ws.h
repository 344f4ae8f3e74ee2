// Weight-stationary fp32-MFMA convolutions of the ResNet encoders / decoder (rows E2, D2 and the Bayes-by-backprop twin B1):
// the 5x5 s2 stem and the 64 -> 64 channel 3x3 block convolutions of ImageEncoder / NPDecoder (networks/models.py:63-192,
// networks/ResNet.py:58-74) and of the BBB encoder (networks/ANPMRShapeNet3D.py:40-90).  GPU build only; the run-time-shaped
// implicit-GEMM problems (conv_rt.h) stay as the checked fallback for other geometries and for the hostsim flavour.
//
// Shape (MI355X-first; the same scheme as conv_tc.h / conv3_tc.h, generalised over the map size):
//   * a 256-thread workgroup = 4 waves, wave nt owns output channels 16nt..16nt+15 and keeps its 16-column slice of the
//     [576][64] weight matrix in 144 registers for its whole life; TWO workgroups share a CU (<= 80 KiB of LDS and <= 256
//     registers each), so one stages its next band while the other one's waves have the matrix pipe;
//   * a workgroup walks "bands" of BPOS (16 / 32 / 64) output positions: the band's input patch [ci 64][rows][cols] is staged
//     HBM -> registers -> LDS once and every MFMA reads ONE LDS dword (its A operand); strides are chosen so that the two
//     16-lane halves of every ds_read_b32 hit disjoint banks (scripts/lds_layout_search.py, re-checked by static_assert);
//   * several "jobs" (the passes of a model step: context images, target images, decoder images, each with its own weight
//     set; or the 3x3 main and 3x3 skip convolution of a BBB block, which read the same input) ride in ONE launch;
//   * epilogues are fused: bias, ReLU, residual add + ReLU, the 1x1 stride-2 skip convolution of the plain ResNet block (it
//     reads exactly the centre-tap operand of the 3x3 stride-2 convolution next to it: 16 more registers, no extra LDS read),
//     and the ReLU mask of a data gradient.
//   * weights arrive "lane-native" ([wave][k-step][lane], made once per step by prep_kernel), so a wave's 144 registers load
//     with 144 coalesced 256-byte reads instead of a stride-9 gather.
// v_mfma_f32_16x16x4_f32 (exact fp32): A lane l = A[l&15][l>>4], B lane l = B[l>>4][l&15], C/D lane l reg r = C[4*(l>>4)+r][l&15].
#pragma once
#include "common.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
#ifdef MLHOT_TS
#ifndef RW_TS_HIN
#define RW_TS_HIN 16          // which conv3x3 geometry carries the band-timeline stamps (scripts/dev/trunk_ts.py)
#define RW_TS_S 1
#endif
namespace tf { extern __device__ long long* g_ts_dev; }
#define RW_TS(slot) do { if (tf::g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0 && G::HIN == RW_TS_HIN && G::S == RW_TS_S) tf::g_ts_dev[300 + (slot)] = clock64(); } while (0)
// every workgroup of the stamped geometry (round 6, scripts/dev/trunk_cu_timeline.py): 32 slots per workgroup from g_ts_dev[4096] - slot 0 the
// hardware id (XCC / SE / CU / SIMD of wave 0), 1 entry, 2 prologue done, then per band (up to 7) [staged, MFMAs done, stored], 31 exit;
// constant 100 MHz ticks (wall_clock64) so that workgroups on different CUs share a time base
#define RW_TSALL(slot) do { if (tf::g_ts_dev && threadIdx.x == 0 && G::HIN == RW_TS_HIN && G::S == RW_TS_S && !SKIP1 && blockIdx.x < 1024 && tf::g_ts_dev[4095] == 0) \
    tf::g_ts_dev[4096 + 32 * blockIdx.x + (slot)] = (slot) == 0 ? (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) : wall_clock64(); } while (0)
#else
#define RW_TS(slot) do {} while (0)
#define RW_TSALL(slot) do {} while (0)
#endif
namespace rw {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t mfma4(float a, float b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// A barrier that orders LDS traffic only: `s_waitcnt lgkmcnt(0)` + `s_barrier`.  __syncthreads() is also a workgroup-scope fence for
// GLOBAL memory, and with a band's output stores still in flight hipcc puts `s_waitcnt vmcnt(0)` in front of the barrier: every band
// then waited out its predecessor's store acknowledgements (and the first band the whole weight image) before it even ASKED for its
// patch (round 6, per-workgroup stamps of block 1's conv1: scripts/dev/trunk_cu_timeline.py).  The barriers of the band loop guard the
// LDS patch and nothing else - waves of a workgroup never exchange data through global memory inside a body (the fused small-map
// kernels do between bodies: they keep __syncthreads()).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A wave's 144 (or 16) registers of a lane-native weight image through ONE buffer descriptor: `buffer_load_dword v, v_lane, s[rsrc],
// s_group offen offset:256 j` - the lane part of the address is one VGPR, the k-step part an immediate (16 steps of 256 B) plus an SGPR
// (4 KB groups).  As `wp[ks * 64]` on a flat pointer every load carried its own 64-bit VGPR address (the 36 KB span does not fit
// global_load's 12-bit immediate): the address registers did not fit beside the 144 destinations, hipcc spilled, and the reload's
// `s_waitcnt vmcnt(0)` cut the sequence into several full L2 round trips - 5.2 us of prologue per workgroup (stamps, as above).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wimg_rsrc(const float* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float wimg_load(__amdgpu_buffer_rsrc_t rs, int lane_bytes, int idx) {      // element [idx][lane] of a [..][64] image
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane_bytes + (idx % 16) * 256, (idx / 16) * 4096, 0));
}

constexpr int CH = 64;                 // channels of every block convolution
constexpr int NKS = 144;               // k-steps of a 3x3 convolution over 64 channels (9 taps x 16 groups of 4 channels)
constexpr int WIMG = 4 * NKS * 64;     // floats of one lane-native 3x3 weight image
constexpr int WIMG1 = 4 * 16 * 64;     // ... of a 1x1 weight image
constexpr int MAX_JOBS = 6;
constexpr int WG_SLOTS = 512;          // two 256-thread workgroups per CU
#ifndef C3_RING
#define C3_RING 2
#endif
// Measured and dropped (round 3, c5, one gpurun call each; the option switches are gone again):
//   * fewer workgroups for launches with few bands (>= 2 / 3 / 4 / 6 bands each, to amortise the 147 KB weight image and the patch
//     zeroing of every workgroup): strictly slower (c5 trunk sum 1413 -> 1525 / 1615 / 1839 / 2218 us) - the bands are latency-
//     bound (staging 3.5-5 k cycles, 19.6 k of MFMAs, 21 k of epilogue under load) and more resident workgroups hide more of it;
//   * starting the CU's second workgroup (LDS base != 0, or block index >= 256) 4-16 k cycles late so that one stages while the
//     other computes: neutral to +3 % - the band timeline (scripts/dev/trunk_ts.py on a -DMLHOT_TS build) shows workgroup 0's
//     MFMA phase already running at the full single-wave rate (34 cycles per MFMA), i.e. the two are out of phase by themselves.

// ---- geometry of one 3x3, pad 1 convolution on square HIN x HIN maps -------------------------------------------
// BPOS output positions per band; an M-tile = 16 positions = TR x TC block of the output map (or several whole images when an
// image has fewer than 16 positions); patch = [ci][image-in-band][row][col], col 0 = ix -1, row 0 = iy (S*oy0 - 1).
// KIND 1 is the patch of the stride-2 DATA GRADIENT: HIN = size of the dy map, positions = the HIN x HIN grid of one parity
// class of input pixels, taps reach dy[a + {0,1}][b + {0,1}]: a stride-1 gather with a halo row / column at the bottom / right.
template <int HIN_, int S_, int BPOS_, int TC_, int RS_, int ISZ_, int PS_, int KIND_ = 0>
struct Geo {
  static constexpr int HIN = HIN_, S = S_, BPOS = BPOS_, TC = TC_, RS = RS_, ISZ = ISZ_, PS = PS_, KIND = KIND_;
  static constexpr int HO = HIN / S, WO = HO, PI = HO * WO;
  static constexpr bool MULTI = PI < BPOS;                   // a band covers NI whole images
  static constexpr int NI = MULTI ? BPOS / PI : 1;
  static constexpr int RB = MULTI ? HO : BPOS / WO;          // output rows of a band (per image)
  static constexpr int PR = KIND == 0 ? S * (RB - 1) + 3 : RB + 1;       // patch rows per image
  static constexpr int ROW0 = KIND == 0 ? -1 : 0, COL0 = KIND == 0 ? 1 : 0;   // patch row 0 <-> input row S*oy0 + ROW0; input column 0 sits at patch column COL0
  static constexpr int NACC = BPOS / 16;
  static constexpr int BANDS_PER_IMG = MULTI ? 1 : PI / BPOS;
  static constexpr int PATCH = CH * PS;
  static constexpr int SEGW = HIN >= 4 ? 4 : HIN;            // floats per staging item
  static constexpr int SEG = HIN / SEGW;
  static constexpr int ITEMS = CH * NI * PR * SEG, CNT = (ITEMS + 255) / 256;
  static_assert(RS >= HIN + (KIND == 0 ? (S == 1 ? 2 : 1) : 1) && ISZ >= PR * RS && PS >= NI * ISZ && (KIND == 0 || S == 1), "patch strides");
  static_assert(PATCH * 4 <= 80 * 1024, "two workgroups per CU");
  // (image-in-band, output row in band, output column) of row m of M-tile t
  static constexpr int tile_il(int t, int m) { return PI >= 16 ? (MULTI ? t / (PI / 16) : 0) : t * (16 / PI) + m / PI; }
  static constexpr int tile_oy(int t, int m) {
    if (PI >= 16) { const int tt = MULTI ? t % (PI / 16) : t; return (tt / (WO / TC)) * (16 / TC) + m / TC; }
    return (m % PI) / WO;
  }
  static constexpr int tile_ox(int t, int m) {
    if (PI >= 16) { const int tt = MULTI ? t % (PI / 16) : t; return (tt % (WO / TC)) * TC + m % TC; }
    return (m % PI) % WO;
  }
  static constexpr int posoff(int t, int m) { return tile_il(t, m) * ISZ + S * tile_oy(t, m) * RS + S * tile_ox(t, m); }
  // both 32-lane halves of the A-operand ds_read_b32 of every M-tile hit 32 distinct banks
  static constexpr bool conflict_free() {
    for (int t = 0; t < NACC; ++t)
      for (int half = 0; half < 4; half += 2) {
        unsigned seen = 0;
        for (int q = half; q < half + 2; ++q)
          for (int m = 0; m < 16; ++m) {
            const unsigned bit = 1u << ((q * PS + posoff(t, m)) & 31);
            if (seen & bit) return false;
            seen |= bit;
          }
      }
    return true;
  }
  static_assert(conflict_free(), "LDS layout has bank conflicts");
};
// scripts/lds_layout_search.py
typedef Geo<64, 2, 32, 16, 65, 195, 195> G64s2;
typedef Geo<32, 1, 64, 16, 34, 136, 144> G32s1;
typedef Geo<32, 2, 64, 16, 33, 297, 297> G32s2;
typedef Geo<16, 1, 64, 16, 18, 108, 112> G16s1;
typedef Geo<16, 2, 32, 4, 20, 180, 181> G16s2;    // half-image bands, as G8s1 (8 x 8 output maps: 360 whole-image bands balance badly over 256 CUs)
typedef Geo<8, 1, 32, 4, 12, 72, 80> G8s1;       // half-image bands: 240-image launches have 720 of them - three per CU, where 360 whole-image
                                                 // bands left 104 CUs with two and the rest with one (scripts/lds_layout_search.py search(8, 1, bpos_list=(32,)))
// The 4 x 4 / 2 x 2 maps (blocks 3 and 4 of a 64 x 64 trunk): ONE M-tile (16 positions) or two per band.  With 64 / 32 positions a
// 240-image launch was 60-120 bands on 256 CUs, each a serial chain of 288-576 MFMAs per wave behind its weight load: the 4 x 4
// stride-1 convolution and data gradient 20.8 / 20.0 -> 14.4 / 14.0 us, the 2 x 2 ones 15.3 / 15.3 -> 12.0 / 10.9 (c5 labels).
typedef Geo<8, 2, 16, 4, 12, 108, 109> G8s2;
typedef Geo<4, 1, 32, 4, 8, 48, 100> G4s1;       // (16 positions = one image per band: no further gain)
typedef Geo<4, 2, 16, 2, 6, 40, 161> G4s2;
typedef Geo<2, 1, 16, 2, 4, 24, 98> G2s1;
// stride-2 data gradient, by the size of the dy map
typedef Geo<32, 1, 32, 16, 33, 66, 80, 1> D32;
typedef Geo<16, 1, 32, 8, 24, 72, 80, 1> D16;
typedef Geo<8, 1, 32, 4, 12, 60, 80, 1> D8;
typedef Geo<4, 1, 16, 4, 8, 40, 44, 1> D4;       // one M-tile per band, as above: stride-2 data gradients of blocks 3 / 4 (two launches each)
typedef Geo<2, 1, 16, 2, 4, 24, 98, 1> D2;       // 17.0 / 17.0 / 17.9 / 18.0 -> 15.5 / 12.8 / 12.4 / 12.6 us

// ---- lane-native weight images ------------------------------------------------------------------------------------
// F image (forward operand order)  : [nt][ks][lane] = W[16nt + lr][4(ks % 16) + lq][tap = ks / 16]
// D image (data-gradient order)    : [nt][ks][lane] = W[4(ks % 16) + lq][16nt + lr][tap = ks / 16]
// 1x1 weights [64][64] the same with a single tap; stem weights [64][K] (K = C*25): [nt][ks][lane] = W[16nt + lr][4ks + lq] (0 beyond K).
struct PrepItem { const float* w; float* f; float* d; int kind; int K; };      // kind 0: 3x3, 1: 1x1, 2: stem
struct PrepItems { PrepItem it[56]; int n; };
__global__ __launch_bounds__(256) void prep_kernel(const PrepItems items) {
  const PrepItem it = items.it[blockIdx.y];
  const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
  const int nks = it.kind == 0 ? NKS : it.kind == 1 ? 16 : (it.K + 3) / 4;
  for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < 4 * nks; e += gridDim.x * 4) {
    const int nt = e / nks, ks = e % nks;
    if (it.kind == 2) {
      const int k = 4 * ks + lq;
      it.f[(size_t)e * 64 + lane] = k < it.K ? it.w[(size_t)(16 * nt + lr) * it.K + k] : 0.f;
      continue;
    }
    const int taps = it.kind == 0 ? 9 : 1, tap = ks / 16, c4 = 4 * (ks % 16) + lq, n = 16 * nt + lr;
    if (it.f) it.f[(size_t)e * 64 + lane] = it.w[((size_t)n * CH + c4) * taps + tap];
    if (it.d) it.d[(size_t)e * 64 + lane] = it.w[((size_t)c4 * CH + n) * taps + tap];
  }
}

// ---- forward-type 3x3 convolution ---------------------------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_BIAS_RES_RELU = 2, EPI_MASK = 3 };
struct FwdJob {
  const float* x;              // [n_img][64][HIN][HIN]
  const float* wimg;           // lane-native weights (F image; a D image with `flip` turns the kernel into the stride-1 data gradient)
  const float* b;              // bias [64] or null
  float* y;                    // [n_img][64][HO][HO]
  const float* aux;            // EPI_BIAS_RES_RELU: residual (same shape as y); EPI_MASK: activation whose sign masks the result
  const float* w1img; const float* b1; float* y1;     // fused 1x1 stride-2 convolution of the same input (SKIP1 kernels)
  int n_img, epi, flip;        // flip: use tap 8 - t (transposed convolution)
  int wg0, nwg;                // workgroups [wg0, wg0 + nwg) of the launch belong to this job
  int img_lo;                  // images below this index are treated as absent (zeros in, nothing out): the fused small-map kernels own image ranges
};
struct FwdJobs { FwdJob j[MAX_JOBS]; int n; };


// Stage one band's patch: HBM -> registers -> LDS in chunks of <= 10 vector loads per thread (a barrier in front: every wave is
// done reading the previous band; one behind: the patch is complete).  Rows outside the map and images beyond n_img become zeros.
template <class G>
__device__ __forceinline__ void stage_patch(float* patch, const float* __restrict__ x, int img0, int oy0, int n_img, int tid, int img_lo = 0) {
  typedef float stage_t __attribute__((ext_vector_type(G::SEGW)));
  constexpr int CH_ITEMS = 10;
  lds_barrier();
  if constexpr (G::SEG >= 4) {
    // Maps of 16 x 16 and up: thread = (row segment seg, channel cil of a group of 256 / SEG); its items run over (channel group,
    // image, patch row) - compile-time steps, so the requests and stores of a thread differ by wave-uniform / immediate offsets
    // only and whether a row exists is a scalar test.  (Dealt as item e = tid + 256 j every item cost two divisions by run-time-
    // irregular constants and a 64-bit address: ~15 vector instructions, 18 times per band in the 32 x 32 stride-2 geometry - on the
    // port the fp32 MFMAs issue through.)
    constexpr int TCI = 256 / G::SEG, NCH = CH / TCI, NROW = G::NI * G::PR, CNT2 = NCH * NROW;
    static_assert(TCI <= CH && CH % TCI == 0, "thread map");
    const int seg = tid & (G::SEG - 1), cil = tid / G::SEG;
    const float* gl = x + (size_t)cil * (G::HIN * G::HIN) + 4 * seg;                 // lane part of the address
    float* dl = patch + cil * G::PS + G::COL0 + 4 * seg;
#pragma unroll
    for (int j0 = 0; j0 < CNT2; j0 += CH_ITEMS) {
      stage_t st[CH_ITEMS];
#pragma unroll
      for (int jj = 0; jj < CH_ITEMS; ++jj) {
        const int j = j0 + jj;
        if (j >= CNT2) break;
        const int ch = j / NROW, il = (j % NROW) / G::PR, pr = j % G::PR;
        const int iy = G::S * oy0 + G::ROW0 + pr;                                     // wave-uniform
        // UNCONDITIONAL loads from a clamped (row, image): a row outside the map / an absent image is decided when its words are
        // stored (zeros).  As `v = 0; if (row exists) v = load` every item was a phi of a constant and a load: hipcc copied the
        // loaded registers right behind the request (`global_load_dwordx4 v[4:7]; s_waitcnt vmcnt(0); v_mov v45, v4`) - the second
        // of a band's 18 requests waited for the first two AND for everything older (the weight image), the rest went out a round
        // trip later (round 6, the kernel's ISA next to scripts/dev/trunk_cu_timeline.py's stamps).
        const int iyc = iy < 0 ? 0 : (iy >= G::HIN ? G::HIN - 1 : iy);
        int ic = img0 + il;
        ic = ic >= n_img ? n_img - 1 : ic;
        ic = ic < img_lo ? img_lo : ic;
        st[jj] = *reinterpret_cast<const stage_t*>(gl + ((size_t)ic * CH + ch * TCI) * (G::HIN * G::HIN) + iyc * G::HIN);
      }
#pragma unroll
      for (int jj = 0; jj < CH_ITEMS; ++jj) {
        const int j = j0 + jj;
        if (j >= CNT2) break;
        const int ch = j / NROW, il = (j % NROW) / G::PR, pr = j % G::PR;
        const int iy = G::S * oy0 + G::ROW0 + pr;
        float* d = dl + ch * TCI * G::PS + il * G::ISZ + pr * G::RS;
        if (iy >= 0 && iy < G::HIN && img0 + il < n_img && img0 + il >= img_lo) {     // wave-uniform: a scalar branch around four LDS stores
#pragma unroll
          for (int q = 0; q < 4; ++q) d[q] = st[jj][q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) d[q] = 0.f;
        }
      }
    }
    lds_barrier();
    return;
  }
#pragma unroll
  for (int j0 = 0; j0 < G::CNT; j0 += CH_ITEMS) {
    stage_t st[CH_ITEMS];
#pragma unroll
    for (int jj = 0; jj < CH_ITEMS; ++jj) {
      const int j = j0 + jj;
      if (j >= G::CNT) break;
      const int e = tid + j * 256;
      const int seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, il = (row / G::PR) % G::NI, ci = row / (G::PR * G::NI);
      const int iy = G::S * oy0 + G::ROW0 + pr;
      stage_t v;
#pragma unroll
      for (int q = 0; q < G::SEGW; ++q) v[q] = 0.f;
      if (e < G::ITEMS && iy >= 0 && iy < G::HIN && img0 + il < n_img && img0 + il >= img_lo)
        v = *reinterpret_cast<const stage_t*>(x + (((size_t)(img0 + il) * CH + ci) * G::HIN + iy) * G::HIN + G::SEGW * seg);
      st[jj] = v;
    }
#pragma unroll
    for (int jj = 0; jj < CH_ITEMS; ++jj) {
      const int j = j0 + jj;
      if (j >= G::CNT) break;
      const int e = tid + j * 256;
      const int seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, il = (row / G::PR) % G::NI, ci = row / (G::PR * G::NI);
      if (e < G::ITEMS) {
        float* d = patch + ci * G::PS + il * G::ISZ + pr * G::RS + G::COL0 + G::SEGW * seg;
#pragma unroll
        for (int q = 0; q < G::SEGW; ++q) d[q] = st[jj][q];
      }
    }
  }
  lds_barrier();
}

// The body of a forward-type launch: bands band0, band0 + band_step, ... < band_end of job `jb`, the patch in `patch` (G::PATCH floats
// of LDS).  conv3x3_kernel runs it over its share of a job's bands; the fused small-map kernels (tail34_*) call it once per
// convolution on the bands of the workgroup's own image group.
template <class G, bool SKIP1>
__device__ __forceinline__ void conv3x3_body(const FwdJob& jb, float* patch, int band0, int band_end, int band_step) {
  const int tid = threadIdx.x, lane = tid & 63, nt = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int co = 16 * nt + lr;

  RW_TS(0);
  RW_TSALL(0); RW_TSALL(1);
  const float bn = jb.b ? jb.b[co] : 0.f;            // in front of the weight image: vmcnt retires in order, and a use of the bias must not wait for 144 younger loads
  const bool has1 = SKIP1 && jb.w1img != nullptr;      // per job: a launch may mix blocks with and without the fused 1x1 skip
  const float bn1 = (has1 && jb.b1) ? jb.b1[co] : 0.f;

  static_assert(G::PATCH % 4 == 0, "patch zeroing");
  for (int i = tid; i < G::PATCH / 4; i += 256) reinterpret_cast<float4*>(patch)[i] = make_float4(0.f, 0.f, 0.f, 0.f);      // halo columns (and pad words) stay zero for good

  // A-operand base of every M-tile: channel lq of the lane's position (row lr of the tile)
  int aoff[G::NACC];
#pragma unroll
  for (int t = 0; t < G::NACC; ++t) {
    int o = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) o = (lr == m) ? G::posoff(t, m) : o;
    aoff[t] = lq * G::PS + o;
  }

  const int nbands_all = G::MULTI ? (jb.n_img + G::NI - 1) / G::NI : jb.n_img * G::BANDS_PER_IMG;
  const int nbands = band_end < nbands_all ? band_end : nbands_all;
  RW_TS(1);
  RW_TSALL(2);
  int ts_k = 0;
#ifdef MLHOT_TS
  int ts_b = 0;
#endif
  // The first band's patch is staged HERE, in front of the loop, and every later band's at the end of its predecessor's iteration.
  // With the staging at the top of the loop body the loop header carried the back edge's `s_waitcnt vmcnt(0)` (the epilogue's stores
  // have to have read their data registers before the staging reuses them) - and on the way IN that same instruction waited for the
  // whole weight image before the first patch request went out: two dependent round trips in every workgroup's prologue (round 6,
  // scripts/dev/trunk_cu_timeline.py: 5.2 us of prologue + 3.3 us of first staging with no wave of the CU on the matrix pipe).
  auto stage = [&](int band) {
    const int img0 = G::MULTI ? band * G::NI : band / G::BANDS_PER_IMG;
    const int oy0 = G::MULTI ? 0 : (band % G::BANDS_PER_IMG) * G::RB;
    stage_patch<G>(patch, jb.x, img0, oy0, jb.n_img, tid, jb.img_lo);
  };
  RW_TS(2);
  if (band0 < nbands) stage(band0);
  RW_TS(3);
  RW_TSALL(3);
  // The weight image is asked for BEHIND the first patch: vmcnt retires in order, so the patch (39 MB over the chip, the burst every
  // workgroup of the launch makes in the same microsecond) is waited for alone, and the 144 weight loads (L2-resident after the first
  // workgroups) then stream in under the first band's MFMAs - k-step ks needs wr[ks] only, hipcc counts the waits down
  // (`s_waitcnt vmcnt(143 - ks ...)`).  Asked for first, the whole image stood between every workgroup and its first MFMA: 8.5 us
  // from kernel entry in block 1's conv1 (scripts/dev/trunk_cu_timeline.py).  One load sequence for both tap orders: the transposed
  // convolution's 8 - tap is nine scalar selects on the descriptor offset, not a second copy of the loads behind a branch (two
  // definitions of every weight register would be 144 phis).
  float wr[NKS];
  {
    const __amdgpu_buffer_rsrc_t rs = wimg_rsrc(jb.wimg + (size_t)nt * NKS * 64, NKS * 64 * 4);
    int goff[9];
#pragma unroll
    for (int g = 0; g < 9; ++g) goff[g] = (jb.flip ? 8 - g : g) * 4096;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      wr[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * lane + (ks % 16) * 256, goff[ks / 16], 0));
  }
  float w1[SKIP1 ? 16 : 1];
  if (has1) {
    const __amdgpu_buffer_rsrc_t rs1 = wimg_rsrc(jb.w1img + (size_t)nt * 16 * 64, 16 * 64 * 4);
#pragma unroll
    for (int cg = 0; cg < 16; ++cg) w1[cg] = wimg_load(rs1, 4 * lane, cg);
  }
#pragma unroll 1
  for (int band = band0; band < nbands; band += band_step) {
    const int img0 = G::MULTI ? band * G::NI : band / G::BANDS_PER_IMG;
    const int oy0 = G::MULTI ? 0 : (band % G::BANDS_PER_IMG) * G::RB;

    // ---- the epilogue's addresses and its `aux` loads (residual / mask epilogues), in front of the MFMA loop ----------------------
    // Round 5, band timeline of the 16 x 16 conv2 (scripts/dev/trunk_ts.py): MFMAs done at 32.6 k cycles, aux loads issued at 34.2 k,
    // first tile stored at 49.5 k - the band sat 15 k cycles (as long as its MFMAs take) waiting for 16 floats per lane that could
    // have been requested before the first MFMA: every workgroup of the chip asks for its residual tile at about the same time.
    constexpr int VW = G::PI < 16 ? 4 : (G::TC >= 4 ? 4 : 2);
    constexpr int NV = 4 / VW;
    typedef float vw_t __attribute__((ext_vector_type(VW)));
    const bool need_aux = jb.epi == EPI_BIAS_RES_RELU || jb.epi == EPI_MASK;
    unsigned offs[G::NACC][NV];      // element offsets into y / aux / y1 (maps stay far below 2^32 elements; 64-bit offsets cost a register more each - the kernel sits at the 256-register limit)
    bool live[G::NACC][NV];
    vw_t ax[G::NACC][NV];
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) {
#pragma unroll
      for (int vi = 0; vi < NV; ++vi) {
        const int v = vi * VW;
        int il = 0, oy = 0, ox = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (lq == q) { il = G::tile_il(t, 4 * q + v); oy = G::tile_oy(t, 4 * q + v); ox = G::tile_ox(t, 4 * q + v); }
        const int img = img0 + il;
        live[t][vi] = img < jb.n_img && img >= jb.img_lo;
        offs[t][vi] = (((unsigned)(live[t][vi] ? img : 0) * CH + co) * G::HO + oy0 + oy) * G::WO + ox;
      }
    }
    // all of them in ONE job-uniform branch, unconditional on `live` (a dead tile reads image 0's words and never uses them), no
    // zero-initialisation: as `ax = 0; if (need_aux && live) ax = load` each was a phi of a constant and a load, and hipcc waited
    // `vmcnt(0)` behind the first request to copy it - a full round trip in front of every band's MFMAs (round 6, ISA)
    if (need_aux) {
#pragma unroll
      for (int t = 0; t < G::NACC; ++t)
#pragma unroll
        for (int vi = 0; vi < NV; ++vi) ax[t][vi] = *reinterpret_cast<const vw_t*>(jb.aux + offs[t][vi]);
    }

    // ---- 144 k-steps x NACC tiles ----
    f32x4_t acc[G::NACC], acc1[SKIP1 ? G::NACC : 1];
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (SKIP1) {
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) acc1[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    // A operands through a register ring RD k-steps deep: the reads of k-step ks + RD are issued behind the MFMAs of k-step ks.
    // Left to itself hipcc reads each operand right in front of its MFMA and waits lgkmcnt(0) on it - with 144 weight registers
    // there is no room for anything else, and the SKIP1 instantiations funnelled EVERY operand through one register
    // (`ds_read_b32 v16; s_waitcnt lgkmcnt(0); v_mfma ... v16` 576 times per band: the LDS round trip in front of every MFMA).
    constexpr int RD = C3_RING;
    auto aread = [&](int ks, int t) {
      const int tap = ks / 16, cg = ks % 16;
      return patch[aoff[t] + cg * 4 * G::PS + (tap / 3) * G::RS + tap % 3];
    };
    float xa[RD][G::NACC];
#pragma unroll
    for (int d = 0; d < RD; ++d)
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) xa[d][t] = aread(d, t);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      float a[G::NACC];
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) a[t] = xa[ks % RD][t];
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) acc[t] = mfma4(a[t], wr[ks], acc[t]);
      if (SKIP1 && ks / 16 == 4 && has1) {
#pragma unroll
        for (int t = 0; t < G::NACC; ++t) acc1[t] = mfma4(a[t], w1[ks % 16], acc1[t]);
      }
      if (ks + RD < NKS) {
#pragma unroll
        for (int t = 0; t < G::NACC; ++t) xa[ks % RD][t] = aread(ks + RD, t);
      }
      __builtin_amdgcn_sched_barrier(0);       // pins the order; unfenced, hipcc hoists the reads of the whole band to the top and spills
    }

    RW_TS(4 + 4 * ts_k);
#ifdef MLHOT_TS
    if (ts_b < 7) RW_TSALL(4 + 3 * ts_b);
#endif
    // ---- epilogue: lane holds rows 4lq..4lq+3 of every tile for channel co ----
    // The four rows are contiguous in the NCHW plane in runs of VW (a tile row has TC >= 2 columns; 2x2 maps: a whole plane).
    // Two passes: every tile's output offset and - for the residual / mask epilogues - its `aux` load are made IN FRONT of the
    // MFMA loop (round 5, below); here the arithmetic and the stores.  (One pass, tile by tile, hipcc waited for each tile's aux load
    // before it issued the next: the band timeline of the 16x16 conv2 showed 23 k cycles of epilogue behind 19.6 k of MFMAs.)
    RW_TS(16 + (ts_k > 0));
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) {
      if (t == 1) RW_TS(18 + (ts_k > 0));
#pragma unroll
      for (int vi = 0; vi < NV; ++vi) {
        if (!live[t][vi]) continue;
        const int v = vi * VW;
        const unsigned o = offs[t][vi];
        vw_t r;
#pragma unroll
        for (int q = 0; q < VW; ++q) r[q] = acc[t][v + q] + bn;
        if (need_aux) {
#pragma unroll
          for (int q = 0; q < VW; ++q) r[q] = jb.epi == EPI_MASK ? (ax[t][vi][q] > 0.f ? r[q] : 0.f) : r[q] + ax[t][vi][q];
        }
        if (jb.epi == EPI_BIAS_RELU || jb.epi == EPI_BIAS_RES_RELU) {
#pragma unroll
          for (int q = 0; q < VW; ++q) r[q] = fmaxf(r[q], 0.f);
        }
        *reinterpret_cast<vw_t*>(jb.y + o) = r;
        if (has1) {
          vw_t r1;
#pragma unroll
          for (int q = 0; q < VW; ++q) r1[q] = acc1[t][v + q] + bn1;
          *reinterpret_cast<vw_t*>(jb.y1 + o) = r1;
        }
      }
    }
    RW_TS(5 + 4 * ts_k);
#ifdef MLHOT_TS
    if (ts_b < 7) RW_TSALL(5 + 3 * ts_b);
    ++ts_b;
#endif
    ts_k = ts_k < 3 ? ts_k + 1 : 3;
    if (band + band_step < nbands) {
      RW_TS(2 + 4 * ts_k);
      stage(band + band_step);
      RW_TS(3 + 4 * ts_k);
#ifdef MLHOT_TS
      if (ts_b < 7) RW_TSALL(3 + 3 * ts_b);
#endif
    }
  }
  RW_TS(20);
  RW_TSALL(31);
}

template <class G, bool SKIP1>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const FwdJobs jobs) {
  __shared__ __attribute__((aligned(16))) float patch[G::PATCH];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const FwdJob& jb = jobs.j[ji];
  conv3x3_body<G, SKIP1>(jb, patch, (int)blockIdx.x - jb.wg0, 1 << 30, jb.nwg);
}

// Workgroup shares of the jobs of one launch: proportional to their band counts, at least one each, WG_SLOTS in all at most.
template <class G>
inline int plan_fwd(FwdJobs& jobs) {
  int nb[MAX_JOBS], total = 0;
  for (int i = 0; i < jobs.n; ++i) {
    nb[i] = G::MULTI ? (jobs.j[i].n_img + G::NI - 1) / G::NI : jobs.j[i].n_img * G::BANDS_PER_IMG;
    total += nb[i];
  }
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) {
    int share = total <= WG_SLOTS ? nb[i] : (int)((long)WG_SLOTS * nb[i] / total);
    if (share < 1) share = 1;
    if (share > nb[i]) share = nb[i];
    jobs.j[i].wg0 = wg; jobs.j[i].nwg = share;
    wg += share;
  }
  return wg;
}

template <class G, bool SKIP1>
inline int launch_conv3x3(FwdJobs& jobs, hipStream_t s, const char* what) {
  const int grid = plan_fwd<G>(jobs);
  if (grid <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL((conv3x3_kernel<G, SKIP1>), dim3(grid), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}

// -> MLHOT_ERR_UNSUPPORTED when no kernel is instantiated for (HIN, stride)
inline int conv3x3_dispatch(int HIN, int S, bool skip1, FwdJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
#define MLHOT_RW_CASE(H, ST, GEO)                                                                   \
  if (HIN == H && S == ST) return skip1 ? (ST == 2 ? launch_conv3x3<GEO, (ST == 2)>(jobs, s, what) : MLHOT_ERR_UNSUPPORTED) \
                                        : launch_conv3x3<GEO, false>(jobs, s, what);
  MLHOT_RW_CASE(64, 2, G64s2) MLHOT_RW_CASE(32, 1, G32s1) MLHOT_RW_CASE(32, 2, G32s2) MLHOT_RW_CASE(16, 1, G16s1)
  MLHOT_RW_CASE(16, 2, G16s2) MLHOT_RW_CASE(8, 1, G8s1) MLHOT_RW_CASE(8, 2, G8s2) MLHOT_RW_CASE(4, 1, G4s1)
  MLHOT_RW_CASE(4, 2, G4s2) MLHOT_RW_CASE(2, 1, G2s1)
#undef MLHOT_RW_CASE
  return MLHOT_ERR_UNSUPPORTED;
}
inline bool conv3x3_supported(int HIN, int S) {
  return (S == 2 && (HIN == 64 || HIN == 32 || HIN == 16 || HIN == 8 || HIN == 4)) || (S == 1 && (HIN == 32 || HIN == 16 || HIN == 8 || HIN == 4 || HIN == 2));
}

// ---- data gradient of a 3x3 stride-2 pad-1 convolution -------------------------------------------------------------
// Input pixel (2a + py, 2b + px) only sees the taps with ky = 1 (py = 0) or ky in {0, 2} (py = 1), same for kx: four parity
// classes with 1 / 2 / 2 / 4 taps, each a small stride-1 gather dy[a + {0,1}][b + {0,1}] on the HO x HO grid.  A wave keeps all
// nine taps of its 16 input channels (D image: 144 registers) and one accumulator per (class, M-tile): a band of BPOS grid
// positions costs 144 * BPOS/16 MFMAs, like the forward.  The epilogue interleaves the two column parities of a row into
// runs of 8 consecutive dx floats; optionally it adds into dx (second source of a residual join) and / or masks with the
// ReLU of the layer input.  SKIP1: the plain ResNet block's 1x1 stride-2 skip only reaches class (0, 0); its 16 k-steps (a
// second dy source g1, 16 more weight registers) ride in the same accumulators.
struct DgJob {
  const float* dy;             // [n_img][64][HO][HO] (already masked by its producer)
  const float* wimg;           // D image of the convolution's weights
  float* dx;                   // [n_img][64][2HO][2HO]
  const float* xact;           // when set: dx *= (xact > 0)   (the ReLU that produced the convolution's input)
  const float* g1; const float* w1img;      // SKIP1: dy and D image of the 1x1 stride-2 convolution on the same input
  int n_img, accumulate, wg0, nwg;
  int img_lo;                  // as FwdJob::img_lo
};
struct DgJobs { DgJob j[MAX_JOBS]; int n; };

// DUAL (512 threads): the block input's gradient has TWO 3x3 stride-2 sources - the skip convolution's (dy, wimg) and conv1's (g1,
// w1img) - that used to be two launches, the second adding onto the first one's dx (read - modify - write of the largest
// gradient map: 262 MB of HBM traffic in c5's block 1, `frac_wait_any` 0.61).  Waves 0-3 run the first source, waves 4-7 the
// second, each with its own weight registers and patch; the second half hands its accumulators over through LDS and the first
// half writes dx once, masked.
// mask rows in LDS: per band 2 RB dx rows x 2 HO columns = 128 floats per channel for the three geometries with whole-row bands
// (D8 / D16 / D32); channel stride 132 words = 4 (mod 64): the 16 lanes of an epilogue read (one channel each, one float4) cover the
// 64 banks once
constexpr int DG_XL_LD = 132, DG_XL_F4 = CH * 32;       // float4 items per band
template <class G> __host__ __device__ constexpr bool dg_xl() { return !G::MULTI && G::PI >= 16 && 2 * G::RB * 2 * G::HO == 128; }
template <class G> __host__ __device__ constexpr int dg_xl_floats() { return dg_xl<G>() ? CH * DG_XL_LD : 0; }

template <class G, bool SKIP1, bool DUAL = false>
__device__ __forceinline__ void dgrad2_body(const DgJob& jb, float* patch_all, int band0, int band_end, int band_step) {
  static_assert(G::KIND == 1 && (G::TC >= 4 || G::PI < 16), "epilogue needs 4 consecutive grid columns per lane (or whole 2x2 maps)");
  static_assert(!(SKIP1 && DUAL), "a pass has a 1x1 or a 3x3 skip");
  const int half = DUAL ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;
  float* patch = patch_all + (DUAL ? half * G::PATCH : 0);
  float* patch1 = patch + G::PATCH;
  const int tid = DUAL ? ((int)threadIdx.x & 255) : (int)threadIdx.x, lane = tid & 63, nt = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  const int ci = 16 * nt + lr;
  constexpr int NT = G::NACC, HX = 2 * G::HO;
  const float* src_dy = (DUAL && half) ? jb.g1 : jb.dy;
  // The ReLU mask's source rows (xact, the block's input under this band's dx rows) ride through LDS: requested together with the
  // band's dy patch, read by the epilogue.  (Round 5: as global loads IN the epilogue they cost the c5 step 35 us - knock-out,
  // profiles/r05_ab_dgrad2_mask_knockout.txt: block 1's dual launch 116.7 -> 95.3 us without them - the dual kernel has ONE
  // workgroup per CU and nothing else to run while 8 float4 per lane make their round trip.)
  constexpr bool XL = dg_xl<G>();
  constexpr int XNT = DUAL ? 512 : 256, XK = DG_XL_F4 / XNT;
  static_assert(XK == 4 || XK == 8, "mask staging items per thread");
  float* xl = patch_all + (DUAL ? 2 * G::PATCH + 4 * 16 * G::NACC * 64 : G::PATCH * (SKIP1 ? 2 : 1));

  const bool has1 = SKIP1 && jb.w1img != nullptr;      // per job
  for (int i = tid; i < G::PATCH * (SKIP1 ? 2 : 1); i += 256) patch[i] = 0.f;      // halo column / pad words stay zero

  int aoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int o = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) o = (lr == m) ? G::posoff(t, m) : o;
    aoff[t] = lq * G::PS + o;
  }
  const int nbands_all = G::MULTI ? (jb.n_img + G::NI - 1) / G::NI : jb.n_img * G::BANDS_PER_IMG;
  const int nbands = band_end < nbands_all ? band_end : nbands_all;
  const bool xl_on = XL && jb.xact != nullptr;                  // job-uniform
  // one band's staging: the mask rows' requests, the dy patch(es), the mask rows' LDS stores.  As in conv3x3_body (round 6) the FIRST
  // band is staged in front of the loop and in front of the weight image, every later band at the end of its predecessor's iteration.
  auto stage_all = [&](int band) __attribute__((always_inline)) {
    const int img0 = G::MULTI ? band * G::NI : band / G::BANDS_PER_IMG;
    const int a0 = G::MULTI ? 0 : (band % G::BANDS_PER_IMG) * G::RB;
    // eight NAMED float4s (an array indexed in an unrolled loop stays in scratch memory: hipcc decides before it unrolls - 144 bytes
    // per lane in the first version of this)
    float4 xr0, xr1, xr2, xr3, xr4, xr5, xr6, xr7;
    xr0 = xr1 = xr2 = xr3 = xr4 = xr5 = xr6 = xr7 = make_float4(0.f, 0.f, 0.f, 0.f);
#define DG_XE(k) ((int)threadIdx.x + XNT * (k))
#define DG_XLOAD(k) *reinterpret_cast<const float4*>(xg + (size_t)(DG_XE(k) >> 5) * (HX * HX) + 4 * (DG_XE(k) & 31))
#define DG_XSTORE(k, v) *reinterpret_cast<float4*>(xl + (DG_XE(k) >> 5) * DG_XL_LD + 4 * (DG_XE(k) & 31)) = v
    if (XL && xl_on) {                                            // the band's 128 mask floats per channel: contiguous rows 2 a0 .. 2 (a0 + RB) - 1
      const float* xg = jb.xact + ((size_t)img0 * CH * HX + 2 * a0) * HX;
      xr0 = DG_XLOAD(0); xr1 = DG_XLOAD(1); xr2 = DG_XLOAD(2); xr3 = DG_XLOAD(3);
      if (XK == 8) { xr4 = DG_XLOAD(4); xr5 = DG_XLOAD(5); xr6 = DG_XLOAD(6); xr7 = DG_XLOAD(7); }
    }
    stage_patch<G>(patch, src_dy, img0, a0, jb.n_img, tid, jb.img_lo);
    if (has1) stage_patch<G>(patch1, jb.g1, img0, a0, jb.n_img, tid, jb.img_lo);
    if (XL && xl_on) {                                            // (the previous band's epilogue is behind stage_patch's first barrier)
      DG_XSTORE(0, xr0); DG_XSTORE(1, xr1); DG_XSTORE(2, xr2); DG_XSTORE(3, xr3);
      if (XK == 8) { DG_XSTORE(4, xr4); DG_XSTORE(5, xr5); DG_XSTORE(6, xr6); DG_XSTORE(7, xr7); }
    }
#undef DG_XE
#undef DG_XLOAD
#undef DG_XSTORE
  };
  if (band0 < nbands) stage_all(band0);
  float wr[NKS];
  {
    const __amdgpu_buffer_rsrc_t rs = wimg_rsrc(((DUAL && half) ? jb.w1img : jb.wimg) + (size_t)nt * NKS * 64, NKS * 64 * 4);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) wr[ks] = wimg_load(rs, 4 * lane, ks);
  }
  float w1[SKIP1 ? 16 : 1];
  if (has1) {
    const __amdgpu_buffer_rsrc_t rs1 = wimg_rsrc(jb.w1img + (size_t)nt * 16 * 64, 16 * 64 * 4);
#pragma unroll
    for (int cg = 0; cg < 16; ++cg) w1[cg] = wimg_load(rs1, 4 * lane, cg);
  }
#pragma unroll 1
  for (int band = band0; band < nbands; band += band_step) {
    const int img0 = G::MULTI ? band * G::NI : band / G::BANDS_PER_IMG;
    const int a0 = G::MULTI ? 0 : (band % G::BANDS_PER_IMG) * G::RB;

    f32x4_t acc[4][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[c][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // (An operand ring - the reads of step or group g + 1 in front of the MFMAs of g, fenced - as in conv3x3_kernel and
    // wgrad_kernel made THIS kernel 8-13 % slower (78 -> 87 / 89 us on block 1): hipcc's own placement inside the 4-step fences
    // below already overlaps the reads of one tile with the MFMAs of the other.)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int py = c >> 1, px = c & 1;
#pragma unroll
      for (int ty = 0; ty < (py ? 2 : 1); ++ty)
#pragma unroll
        for (int tx = 0; tx < (px ? 2 : 1); ++tx) {
          const int ky = py ? (ty ? 2 : 0) : 1, kx = px ? (tx ? 2 : 0) : 1;
          const int roff = (py && ky == 0) ? 1 : 0, coff = (px && kx == 0) ? 1 : 0;          // dy row a + roff, column b + coff
#pragma unroll
          for (int cg = 0; cg < 16; ++cg) {
            const int off = cg * 4 * G::PS + roff * G::RS + coff;
            float a[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a[t] = patch[aoff[t] + off];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[c][t] = mfma4(a[t], wr[(ky * 3 + kx) * 16 + cg], acc[c][t]);
            if (cg % 4 == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if (has1) {
#pragma unroll
      for (int cg = 0; cg < 16; ++cg) {
        float a[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = patch1[aoff[t] + cg * 4 * G::PS];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[0][t] = mfma4(a[t], w1[cg], acc[0][t]);
        if (cg % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }

    if (DUAL) {       // the second source's accumulators -> LDS -> added by the first half, which alone runs the epilogue
      float* xw = patch_all + 2 * G::PATCH + (size_t)nt * (16 * NT) * 64 + lane;
      if (half == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[((c * NT + t) * 4 + r) * 64] = acc[c][t][r];
      }
      __syncthreads();
      if (half == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[c][t][r] += xw[((c * NT + t) * 4 + r) * 64];
      }
    }
    if (!DUAL && XL && xl_on) __syncthreads();                    // the mask rows are complete (DUAL: the exchange's barrier above)
    // ---- epilogue: rows 4lq..4lq+3 of tile t = 4 consecutive grid columns (or a whole 2x2 grid) of channel ci ----
    if (!DUAL || half == 0)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      int il = 0, a = 0, b = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (lq == q) { il = G::tile_il(t, 4 * q); a = G::tile_oy(t, 4 * q); b = G::tile_ox(t, 4 * q); }
      const int img = img0 + il;
      if (img >= jb.n_img || img < jb.img_lo) continue;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        if (G::PI >= 16) {
          // dx row 2(a0 + a) + py, columns 2b .. 2b + 7
          const size_t o = (((size_t)img * CH + ci) * HX + 2 * (a0 + a) + py) * HX + 2 * b;
          float r[8];
#pragma unroll
          for (int q = 0; q < 4; ++q) { r[2 * q] = acc[2 * py][t][q]; r[2 * q + 1] = acc[2 * py + 1][t][q]; }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 v = make_float4(r[4 * h], r[4 * h + 1], r[4 * h + 2], r[4 * h + 3]);
            if (jb.accumulate) { const float4 u = *reinterpret_cast<const float4*>(jb.dx + o + 4 * h); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            if (jb.xact) {
              const float4 u = XL ? *reinterpret_cast<const float4*>(xl + ci * DG_XL_LD + (2 * a + py) * HX + 2 * b + 4 * h)
                                  : *reinterpret_cast<const float4*>(jb.xact + o + 4 * h);
              v.x = u.x > 0.f ? v.x : 0.f; v.y = u.y > 0.f ? v.y : 0.f; v.z = u.z > 0.f ? v.z : 0.f; v.w = u.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(jb.dx + o + 4 * h) = v;
          }
        } else {
          // 2x2 grid -> 4x4 dx map: grid row ar in {0,1} -> dx row 2ar + py, all four columns
#pragma unroll
          for (int ar = 0; ar < 2; ++ar) {
            const size_t o = (((size_t)img * CH + ci) * HX + 2 * ar + py) * HX;
            float4 v = make_float4(acc[2 * py][t][2 * ar], acc[2 * py + 1][t][2 * ar], acc[2 * py][t][2 * ar + 1], acc[2 * py + 1][t][2 * ar + 1]);
            if (jb.accumulate) { const float4 u = *reinterpret_cast<const float4*>(jb.dx + o); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
            if (jb.xact) {
              const float4 u = *reinterpret_cast<const float4*>(jb.xact + o);
              v.x = u.x > 0.f ? v.x : 0.f; v.y = u.y > 0.f ? v.y : 0.f; v.z = u.z > 0.f ? v.z : 0.f; v.w = u.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(jb.dx + o) = v;
          }
        }
      }
    }
    if (band + band_step < nbands) stage_all(band + band_step);
  }
}

template <class G, bool SKIP1>
__global__ __launch_bounds__(256, 2) void dgrad2_kernel(const DgJobs jobs) {
  __shared__ __attribute__((aligned(16))) float patch[G::PATCH * (SKIP1 ? 2 : 1) + dg_xl_floats<G>()];
  static_assert((G::PATCH * (SKIP1 ? 2 : 1) + dg_xl_floats<G>()) * 4 <= 80 * 1024, "two workgroups per CU");
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const DgJob& jb = jobs.j[ji];
  dgrad2_body<G, SKIP1>(jb, patch, (int)blockIdx.x - jb.wg0, 1 << 30, jb.nwg);
}

template <class G>
inline int plan_dg(DgJobs& jobs) {
  int nb[MAX_JOBS], total = 0, wg = 0;
  for (int i = 0; i < jobs.n; ++i) {
    nb[i] = G::MULTI ? (jobs.j[i].n_img + G::NI - 1) / G::NI : jobs.j[i].n_img * G::BANDS_PER_IMG;
    total += nb[i];
  }
  for (int i = 0; i < jobs.n; ++i) {
    int share = total <= WG_SLOTS ? nb[i] : (int)((long)WG_SLOTS * nb[i] / total);
    if (share < 1) share = 1;
    if (share > nb[i]) share = nb[i];
    jobs.j[i].wg0 = wg; jobs.j[i].nwg = share; wg += share;
  }
  return wg;
}
template <class G, bool SKIP1>
inline int launch_dgrad2(DgJobs& jobs, hipStream_t s, const char* what) {
  const int grid = plan_dg<G>(jobs);
  if (grid <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL((dgrad2_kernel<G, SKIP1>), dim3(grid), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}
// HO: size of the dy map (the convolution's OUTPUT); dx is 2HO x 2HO
inline int dgrad2_dispatch(int HO, bool skip1, DgJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
#define MLHOT_RW_CASE(H, GEO) if (HO == H) return skip1 ? launch_dgrad2<GEO, true>(jobs, s, what) : launch_dgrad2<GEO, false>(jobs, s, what);
  MLHOT_RW_CASE(32, D32) MLHOT_RW_CASE(16, D16) MLHOT_RW_CASE(8, D8) MLHOT_RW_CASE(4, D4) MLHOT_RW_CASE(2, D2)
#undef MLHOT_RW_CASE
  return MLHOT_ERR_UNSUPPORTED;
}

template <class G>
__global__ __launch_bounds__(512) void dgrad2_dual_kernel(const DgJobs jobs) {
  __shared__ __attribute__((aligned(16))) float patch[2 * G::PATCH + 4 * 16 * G::NACC * 64 + dg_xl_floats<G>()];      // two dy patches + the accumulator exchange + the mask rows
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const DgJob& jb = jobs.j[ji];
  dgrad2_body<G, false, true>(jb, patch, (int)blockIdx.x - jb.wg0, 1 << 30, jb.nwg);
}
template <class G>
inline int launch_dgrad2_dual(DgJobs& jobs, hipStream_t s, const char* what) {
  int grid = plan_dg<G>(jobs);
  if (grid <= 0) return MLHOT_OK;
  if (grid > WG_SLOTS / 2) {            // one 512-thread workgroup per CU: plan against 256 slots
    int nb[MAX_JOBS], total = 0, wg = 0;
    for (int i = 0; i < jobs.n; ++i) { nb[i] = G::MULTI ? (jobs.j[i].n_img + G::NI - 1) / G::NI : jobs.j[i].n_img * G::BANDS_PER_IMG; total += nb[i]; }
    for (int i = 0; i < jobs.n; ++i) {
      int share = (int)((long)(WG_SLOTS / 2) * nb[i] / total);
      if (share < 1) share = 1;
      if (share > nb[i]) share = nb[i];
      jobs.j[i].wg0 = wg; jobs.j[i].nwg = share; wg += share;
    }
    grid = wg;
  }
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL((dgrad2_dual_kernel<G>), dim3(grid), dim3(512), 0, s, jobs);
  }
  return check_launch(what);
}
// both 3x3 stride-2 sources of a block input's gradient in one launch (DgJob: dy / wimg = the skip's, g1 / w1img = conv1's)
inline int dgrad2_dual_dispatch(int HO, DgJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
  if (HO == 16) return launch_dgrad2_dual<D16>(jobs, s, what);
  if (HO == 8) return launch_dgrad2_dual<D8>(jobs, s, what);
  if (HO == 32) return launch_dgrad2_dual<D32>(jobs, s, what);
  return MLHOT_ERR_UNSUPPORTED;
}
inline bool dgrad2_dual_supported(int HO) { return HO == 8 || HO == 16 || HO == 32; }

// ---- blocks 3 and 4 of a 64 x 64 trunk (8 x 8 -> 4 x 4 -> 2 x 2 maps) in ONE launch per direction ---------------------------
// These two blocks are < 1 % of a trunk's FLOPs and were 14 of its launches (c5: 195 us of a 1.52 ms step): every launch a
// weight-image load per workgroup, 144-288 dependent MFMAs per wave, an epilogue and a launch boundary - 9-24 us each.  A
// workgroup of the fused kernels owns FOUR images of one pass (their 2 x 2 maps fill one M-tile) and runs the same convolution
// bodies one after the other on those images' bands, each stage's output going through global memory (L2) to the next stage
// of the SAME workgroup: no cross-workgroup dependency, only __syncthreads() between stages (workgroup-scope release / acquire:
// the stores of a stage are visible to the CU's own later loads).  Unlike the few-row Linear chains (mlp_chain.h) nothing is
// lost: every per-layer launch reloaded a 147 KB weight image per workgroup as well.
//   forward  : conv1.b3 (+ 1x1 skip fused | 3x3 skip as a second body) -> conv2.b3 + skip + ReLU -> the same for block 4
//   backward : conv2^T.b4 (masked) -> stride-2 data gradients of block 4 (skip / conv1, join + mask) -> the same for block 3; the
//              masked gradients every stage leaves (DM4, G3, DM3, G2) are what the weight-gradient launches read afterwards.
struct T34Pass {
  const float* x;                       // y2 [n][64][8][8]: the output of block 2
  float *mid3, *y3, *mid4, *y4;         // forward: saved activations (outputs); backward: read
  float *idn3, *idn4;                   // forward scratch: skip-path outputs [n][64][4][4], [n][64][2][2]
  const float* wimg[6];                 // c1_3, c2_3, sk_3, c1_4, c2_4, sk_4: F images (forward) / D images (backward); sk: the 1x1 image when skip1
  const float* b[6];                    // forward: biases
  float *g4, *dm4, *g3, *dm3, *g2;      // backward: masked gradients wrt y4 (input), mid4, y3, mid3, y2 (outputs)
  int n_img, skip1, wg0, nwg;
};
struct T34Jobs { T34Pass p[MAX_JOBS]; int n; };
constexpr int cmax(int a, int b) { return a > b ? a : b; }

#ifndef T34_GI_F
#define T34_GI_F 2
#endif
#ifndef T34_GI_B
#define T34_GI_B 1
#endif
// images per workgroup (1, 2 or 4), forward and backward kernel separately (round 6; one constant before).  The 2 x 2 maps of FOUR images
// fill an M-tile, so with fewer the block-4 stages (and with one image block 3's stride-1 stages, two images per band) run their band
// with the other images masked out (img_lo / n_img of the job): wasted matrix rows, but a shorter dependent chain per workgroup - these
// launches are latency-bound.  Measured (c5, round 4): forward 64 / 49 / 53 us with 4 / 2 / 1 images per workgroup, backward 78 / 54 / 53.
constexpr int GIF = T34_GI_F, GIB = T34_GI_B;
static_assert((GIF == 1 || GIF == 2 || GIF == 4) && (GIB == 1 || GIB == 2 || GIB == 4), "image group");

__global__ __launch_bounds__(256, 2) void tail34_fwd_kernel(const T34Jobs jobs) {
  __shared__ __attribute__((aligned(16))) float patch[cmax(cmax(G8s2::PATCH, G4s1::PATCH), cmax(G4s2::PATCH, G2s1::PATCH))];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.p[ji + 1].wg0) ++ji;
  const T34Pass& P = jobs.p[ji];
  const int g = (int)blockIdx.x - P.wg0, i0 = GIF * g;       // image group: images i0 .. i0 + GIF - 1
  const int hi = i0 + GIF < P.n_img ? i0 + GIF : P.n_img;
  const bool s1 = P.skip1 != 0;
  // block 3, stage A: 8 x 8 -> 4 x 4, one image per band
  {
    const FwdJob a{P.x, P.wimg[0], P.b[0], P.mid3, nullptr, s1 ? P.wimg[2] : nullptr, s1 ? P.b[2] : nullptr, s1 ? P.idn3 : nullptr, hi, EPI_BIAS_RELU, 0, 0, 0, i0};
    if (s1) conv3x3_body<G8s2, true>(a, patch, i0, i0 + GIF, 1);
    else {
      conv3x3_body<G8s2, false>(a, patch, i0, i0 + GIF, 1);
      __syncthreads();
      const FwdJob k{P.x, P.wimg[2], P.b[2], P.idn3, nullptr, nullptr, nullptr, nullptr, hi, EPI_BIAS, 0, 0, 0, i0};
      conv3x3_body<G8s2, false>(k, patch, i0, i0 + GIF, 1);
    }
  }
  __syncthreads();
  {   // block 3, stage B: 4 x 4 stride 1, two images per band
    const FwdJob c{P.mid3, P.wimg[1], P.b[1], P.y3, P.idn3, nullptr, nullptr, nullptr, hi, EPI_BIAS_RES_RELU, 0, 0, 0, i0};
    conv3x3_body<G4s1, false>(c, patch, i0 / 2, (i0 + GIF + 1) / 2, 1);
  }
  __syncthreads();
  {   // block 4, stage A: 4 x 4 -> 2 x 2, four images per band
    const FwdJob a{P.y3, P.wimg[3], P.b[3], P.mid4, nullptr, s1 ? P.wimg[5] : nullptr, s1 ? P.b[5] : nullptr, s1 ? P.idn4 : nullptr, hi, EPI_BIAS_RELU, 0, 0, 0, i0};
    if (s1) conv3x3_body<G4s2, true>(a, patch, i0 / 4, i0 / 4 + 1, 1);
    else {
      conv3x3_body<G4s2, false>(a, patch, i0 / 4, i0 / 4 + 1, 1);
      __syncthreads();
      const FwdJob k{P.y3, P.wimg[5], P.b[5], P.idn4, nullptr, nullptr, nullptr, nullptr, hi, EPI_BIAS, 0, 0, 0, i0};
      conv3x3_body<G4s2, false>(k, patch, i0 / 4, i0 / 4 + 1, 1);
    }
  }
  __syncthreads();
  {   // block 4, stage B: 2 x 2 stride 1
    const FwdJob c{P.mid4, P.wimg[4], P.b[4], P.y4, P.idn4, nullptr, nullptr, nullptr, hi, EPI_BIAS_RES_RELU, 0, 0, 0, i0};
    conv3x3_body<G2s1, false>(c, patch, i0 / 4, i0 / 4 + 1, 1);
  }
}

__global__ __launch_bounds__(256, 2) void tail34_bwd_kernel(const T34Jobs jobs) {
  __shared__ __attribute__((aligned(16))) float patch[cmax(cmax(G2s1::PATCH, G4s1::PATCH), 2 * cmax(D2::PATCH, D4::PATCH))];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.p[ji + 1].wg0) ++ji;
  const T34Pass& P = jobs.p[ji];
  const int g = (int)blockIdx.x - P.wg0, i0 = GIB * g;
  const int hi = i0 + GIB < P.n_img ? i0 + GIB : P.n_img;
  const bool s1 = P.skip1 != 0;
  {   // block 4: d mid4 = conv2^T(g4) . (mid4 > 0)
    const FwdJob c{P.g4, P.wimg[4], nullptr, P.dm4, P.mid4, nullptr, nullptr, nullptr, hi, EPI_MASK, 1, 0, 0, i0};
    conv3x3_body<G2s1, false>(c, patch, i0 / 4, i0 / 4 + 1, 1);
  }
  __syncthreads();
  // block 4: gradient wrt y3 = (conv1^T(dm4) + skip^T(g4)) . (y3 > 0); dy maps 2 x 2, four images per band
  if (s1) {
    const DgJob a{P.dm4, P.wimg[3], P.g3, P.y3, P.g4, P.wimg[5], hi, 0, 0, 0, i0};
    dgrad2_body<D2, true>(a, patch, i0 / 4, i0 / 4 + 1, 1);
  } else {
    const DgJob a{P.g4, P.wimg[5], P.g3, nullptr, nullptr, nullptr, hi, 0, 0, 0, i0};
    dgrad2_body<D2, false>(a, patch, i0 / 4, i0 / 4 + 1, 1);
    __syncthreads();
    const DgJob b2{P.dm4, P.wimg[3], P.g3, P.y3, nullptr, nullptr, hi, 1, 0, 0, i0};
    dgrad2_body<D2, false>(b2, patch, i0 / 4, i0 / 4 + 1, 1);
  }
  __syncthreads();
  {   // block 3: d mid3 = conv2^T(g3) . (mid3 > 0); 4 x 4 maps, two images per band
    const FwdJob c{P.g3, P.wimg[1], nullptr, P.dm3, P.mid3, nullptr, nullptr, nullptr, hi, EPI_MASK, 1, 0, 0, i0};
    conv3x3_body<G4s1, false>(c, patch, i0 / 2, (i0 + GIB + 1) / 2, 1);
  }
  __syncthreads();
  // block 3: gradient wrt y2; dy maps 4 x 4, one image per band
  if (s1) {
    const DgJob a{P.dm3, P.wimg[0], P.g2, P.x, P.g3, P.wimg[2], hi, 0, 0, 0, i0};
    dgrad2_body<D4, true>(a, patch, i0, i0 + GIB, 1);
  } else {
    const DgJob a{P.g3, P.wimg[2], P.g2, nullptr, nullptr, nullptr, hi, 0, 0, 0, i0};
    dgrad2_body<D4, false>(a, patch, i0, i0 + GIB, 1);
    __syncthreads();
    const DgJob b2{P.dm3, P.wimg[0], P.g2, P.x, nullptr, nullptr, hi, 1, 0, 0, i0};
    dgrad2_body<D4, false>(b2, patch, i0, i0 + GIB, 1);
  }
}

inline int tail34_launch(T34Jobs& jobs, bool backward, hipStream_t s, const char* what) {
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) { jobs.p[i].wg0 = wg; { const int gi = backward ? GIB : GIF; jobs.p[i].nwg = (jobs.p[i].n_img + gi - 1) / gi; } wg += jobs.p[i].nwg; }
  if (wg == 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    if (backward) hipLaunchKernelGGL(tail34_bwd_kernel, dim3(wg), dim3(256), 0, s, jobs);
    else hipLaunchKernelGGL(tail34_fwd_kernel, dim3(wg), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}

// ---- weight gradient of a 3x3 (pad 1, stride 1 / 2) convolution ------------------------------------------------------
//   dW[co][ci][tap] = sum over positions  dy[co][pos] * x[ci][pos shifted by tap]:   M = co, N = (tap, ci), K = positions.
// Workgroup (q, z): input-channel tile q (16 channels, so it stages only a quarter of the x patch) and position split z; wave w
// = output-channel tile.  The nine accumulators of a wave (one per tap, 36 registers) live for the whole kernel; per k-step (4
// positions) a wave reads ONE dy operand and nine x operands from LDS.  Bands of 128 (small maps: 64) positions; the x slice uses the forward's
// patch geometry, dy is staged transposed ([pos][co], stride DS) - strides from scripts/lds_layout_search.py.  The k-step's four
// positions are Q apart (pb = blk*4Q + lq*Q + j), which is what makes both gathers bank-conflict free.  Results go out in
// MFMA-native order (one coalesced float4 per lane and tap) into slab row z; wsum_kernel folds the rows and un-permutes.
// The bias gradient rides along in the q = 0 workgroups as a tenth accumulator against an all-ones operand.
template <int HIN_, int S_, int Q_, int RS_, int ISZ_, int PS_, int DS_, int BPOS_ = 128>
struct GeoW {
  static constexpr int HIN = HIN_, S = S_, Q = Q_, RS = RS_, ISZ = ISZ_, PS = PS_, DS = DS_, BPOS = BPOS_, NKS_B = BPOS / 4;
  static constexpr int HO = HIN / S, WO = HO, PI = HO * WO;
  static constexpr bool MULTI = PI < BPOS;
  static constexpr int NI = MULTI ? BPOS / PI : 1, RB = MULTI ? HO : BPOS / WO, PR = S * (RB - 1) + 3;
  static constexpr int BANDS_PER_IMG = MULTI ? 1 : PI / BPOS;
  static constexpr int XS = 16 * PS, DYT = BPOS * DS, LDS = XS + DYT;
  static constexpr int SEGW = HIN >= 4 ? 4 : HIN, SEG = HIN / SEGW, ITEMS = 16 * NI * PR * SEG, CNT = (ITEMS + 255) / 256;
  static constexpr int DSEGW = PI >= 4 ? 4 : PI;                                  // dy staging: runs along the positions of one (image, co) plane
  static constexpr int DITEMS = 64 * BPOS / 4, DCNT = DITEMS / 256;              // float4 items (PI >= 4 always: 2x2 maps have 4 positions)
  static_assert(RS >= HIN + (S == 1 ? 2 : 1) && ISZ >= PR * RS && PS >= NI * ISZ && PI >= 4, "strides");
  static constexpr int pb(int ks, int lq) { return (ks / Q) * 4 * Q + lq * Q + ks % Q; }
  static constexpr int posoff(int p) { return (MULTI ? p / PI : 0) * ISZ + S * (((MULTI ? p % PI : p)) / WO) * RS + S * ((MULTI ? p % PI : p) % WO); }
  static constexpr bool conflict_free() {
    for (int ks = 0; ks < NKS_B; ++ks)
      for (int half = 0; half < 4; half += 2) {
        unsigned sb = 0, sa = 0;
        for (int q = half; q < half + 2; ++q)
          for (int m = 0; m < 16; ++m) {
            const unsigned bb = 1u << ((m * PS + posoff(pb(ks, q))) & 31), ba = 1u << ((pb(ks, q) * DS + m) & 31);
            if ((sb & bb) || (sa & ba)) return false;
            sb |= bb; sa |= ba;
          }
      }
    return true;
  }
  static_assert(conflict_free(), "LDS layout has bank conflicts");
  // operand offsets split into a per-lane base and a compile-time part: position pb(ks, lq) = pb(ks, 0) + lq * Q never crosses
  // a patch row differently from pb(0, lq), so posoff(pb(ks, lq)) = posoff(pb(ks, 0)) + [posoff(pb(0, lq)) - posoff(pb(0, 0))]
  static constexpr bool separable() {
    for (int ks = 0; ks < NKS_B; ++ks)
      for (int q = 0; q < 4; ++q)
        if (posoff(pb(ks, q)) - posoff(pb(ks, 0)) != posoff(pb(0, q)) - posoff(pb(0, 0)) || pb(ks, q) != pb(ks, 0) + q * Q) return false;
    return true;
  }
  static_assert(separable(), "k-step offsets are not lane base + constant");
};
typedef GeoW<64, 2, 8, 65, 585, 585, 66> W64s2;
typedef GeoW<32, 1, 16, 34, 204, 205, 65> W32s1;
typedef GeoW<32, 2, 8, 33, 561, 561, 66> W32s2;
typedef GeoW<16, 1, 1, 18, 180, 182, 80> W16s1;
typedef GeoW<16, 2, 16, 20, 340, 681, 65> W16s2;
typedef GeoW<8, 1, 1, 10, 100, 202, 80> W8s1;
// 64-position bands on the 4 x 4 / 2 x 2 output maps (blocks 3 / 4 of a 64 x 64 trunk): a 240-image launch has 150 bands instead of 75,
// so the planner reaches its 128 slab rows (512 workgroups, each half the chain) - 20.7 / 15.4 / 14.5 / 13.8 -> 19.4 / 13.7 / 13.5 / 12.4 us
typedef GeoW<8, 2, 16, 9, 81, 326, 65, 64> W8s2;
typedef GeoW<4, 1, 1, 6, 36, 146, 80, 64> W4s1;
typedef GeoW<4, 2, 4, 5, 25, 402, 68, 64> W4s2;
typedef GeoW<2, 1, 4, 4, 16, 257, 68, 64> W2s1;

constexpr int SLAB3 = 4 * 4 * 9 * 64 * 4;      // floats of one slab row of a 3x3 weight gradient (= 64 * 576), MFMA-native order
constexpr int SLAB1 = 4 * 4 * 64 * 4;          // ... of a 1x1 weight gradient
struct WgJob {
  const float* x; const float* dy;   // x [n_img][64][HIN][HIN] (the convolution's input), dy [n_img][64][HO][HO] (masked by its producer)
  float* slab; float* slab_b;        // slab rows [z][SLAB3 | SLAB1]; bias rows [z][64] (may be null)
  int n_img, z0, nz, wg0;            // this job owns slab rows z0 .. z0 + nz - 1 and workgroups wg0 .. wg0 + 4 nz - 1
};
struct WgJobs { WgJob j[MAX_JOBS]; int n; };

// body of one weight-gradient workgroup: `rel` = its index inside job `jb` (q = rel & 3: input-channel tile, z = rel >> 2: slab row)
template <class G, bool TAP1>          // TAP1: only the centre tap (the 1x1 stride-2 skip convolution: x[ci][2oy][2ox])
__device__ __forceinline__ void wgrad_body(const WgJob& jb, float* lds, int rel) {
  float* xs = lds;
  float* dyt = lds + G::XS;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  // (Measured and dropped: the four q of a slab row 8 workgroups apart - on ONE XCD, so that the dy tile they all stage meets in one
  // L2 instead of being fetched from HBM up to four times: block 1's launches unchanged (122 / 72 us: they are not HBM-bound), block
  // 2's 42 -> 47 and 25 -> 32 us, the c5 step 1.445 -> 1.456 ms.)
  const int q = rel & 3, z = rel >> 2;
  constexpr int NT = TAP1 ? 1 : 9;

  // per-lane operand bases; the k-step parts are compile-time immediates (GeoW::separable): as two 32-entry per-lane arrays they
  // held 64 registers for the whole kernel
  int lpa = 0, lpb = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lq == k) { lpa = k * G::Q * G::DS; lpb = G::posoff(G::pb(0, k)) - G::posoff(G::pb(0, 0)); }
  const int abase = lpa + 16 * w + lr;
  const int bbase = TAP1 ? lq * G::Q + lr * (G::BPOS + 1) : lpb + lr * G::PS;      // TAP1: compact [ci][position] tile
  f32x4_t acc[NT], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < G::LDS; i += 256) lds[i] = 0.f;

  const int nbands = G::MULTI ? (jb.n_img + G::NI - 1) / G::NI : jb.n_img * G::BANDS_PER_IMG;
  typedef float stage_t __attribute__((ext_vector_type(G::SEGW)));
  // the band's global loads are issued one band ahead: they are in flight under the previous band's MFMAs instead of in front of
  // this band's barrier (exposed, a workgroup spent about as long waiting for HBM as computing, and so did its CU partner)
  // regular staging map (one image per band, maps of 16 x 16 and up): 256 threads = SEG row segments x 16 channels x RP row phases
  constexpr bool REG_MAP = !TAP1 && !G::MULTI && G::SEG >= 4 && G::SEG <= 16;
  constexpr int RM_RP = REG_MAP ? 256 / (G::SEG * 16) : 1, RM_RPP = REG_MAP ? (G::PR + RM_RP - 1) / RM_RP : 1;
  const int rm_seg = tid & (G::SEG - 1), rm_ci = (tid / G::SEG) & 15, rm_row0 = (tid / (G::SEG * 16)) * RM_RPP;
  stage_t st[TAP1 ? 1 : (REG_MAP ? RM_RPP : G::CNT)];
  constexpr int S1N = 16 * G::BPOS / 256;      // TAP1: the compact [16 ci][BPOS positions] tile, items per thread
  float s1[TAP1 ? S1N : 1];
  float4 sd[G::DCNT];
  auto fetch = [&](int band) __attribute__((always_inline)) {
    const int img0 = G::MULTI ? band * G::NI : band / G::BANDS_PER_IMG;
    const int oy0 = G::MULTI ? 0 : (band % G::BANDS_PER_IMG) * G::RB;
    // ---- stage: x slice (channels 16q .. 16q + 15) and the transposed dy tile ----
    // TAP1 (1x1 stride-2 convolution) only ever reads x[ci][2 oy][2 ox]: a compact [16 ci][128 positions] gather, 8 loads per thread
    if (TAP1) {
#pragma unroll
      for (int j = 0; j < S1N; ++j) {
        const int e = tid + j * 256, p = e % G::BPOS, ci = e / G::BPOS;
        const int il = G::MULTI ? p / G::PI : 0, pin = G::MULTI ? p % G::PI : p;
        const int oy = oy0 + pin / G::WO, ox = pin % G::WO;
        s1[j] = img0 + il < jb.n_img ? jb.x[(((size_t)(img0 + il) * CH + 16 * q + ci) * G::HIN + 2 * oy) * G::HIN + 2 * ox] : 0.f;
      }
    } else if constexpr (REG_MAP) {
      // thread = (row segment, channel of the slice, row phase): its items are consecutive patch rows - one lane base, steps of
      // one map row (see stage_patch: the item = tid + 256 j deal cost two irregular divisions and a 64-bit address per item)
      const float* gl = jb.x + ((size_t)img0 * CH + 16 * q + rm_ci) * (G::HIN * G::HIN) + 4 * rm_seg;
#pragma unroll
      for (int j = 0; j < RM_RPP; ++j) {
        const int pr = rm_row0 + j, iy = G::S * oy0 - 1 + pr;
        stage_t v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = 0.f;
        if (pr < G::PR && iy >= 0 && iy < G::HIN) v = *reinterpret_cast<const stage_t*>(gl + iy * G::HIN);
        st[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < G::CNT; ++j) {
        const int e = tid + j * 256;
        const int seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, il = (row / G::PR) % G::NI, ci = row / (G::PR * G::NI);
        const int iy = G::S * oy0 - 1 + pr;
        stage_t v;
#pragma unroll
        for (int k = 0; k < G::SEGW; ++k) v[k] = 0.f;
        if (e < G::ITEMS && iy >= 0 && iy < G::HIN && img0 + il < jb.n_img)
          v = *reinterpret_cast<const stage_t*>(jb.x + (((size_t)(img0 + il) * CH + 16 * q + ci) * G::HIN + iy) * G::HIN + G::SEGW * seg);
        st[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < G::DCNT; ++j) {
      // item e: 4 consecutive band positions p4 .. p4 + 3 of output channel co (they lie in one image plane: PI % 4 == 0).
      // The lanes of a wave take 64 different channels of ONE position quad: transposed into [pos][co] (row stride DS) their
      // stores then hit 64 different banks.  With the positions along the lanes (coalesced 16-byte loads) the stores of a wave
      // landed on 32 / gcd-limited few banks - 4 DS is a multiple of 4, of 32 for DS = 80: every store 32-way serialised, and
      // the LDS pipe is shared with the other workgroup's operand reads; the loads now touch 64 lines of 16 bytes each, which the
      // texture path absorbs in the background.
      const int e = tid + j * 256, co = e & 63, p4 = 4 * (e >> 6);
      const int il = G::MULTI ? p4 / G::PI : 0, pin = G::MULTI ? p4 % G::PI : oy0 * G::WO + p4;
      sd[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (img0 + il < jb.n_img) sd[j] = *reinterpret_cast<const float4*>(jb.dy + ((size_t)(img0 + il) * CH + co) * G::PI + pin);
    }
  };
#ifdef MLHOT_TS
#define WG_TS(slot) do { if (tf::g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0 && !TAP1 && G::HIN == 32 && G::S == 2 && (slot) < 26) tf::g_ts_dev[420 + (slot)] = clock64(); } while (0)
  int ts_b = 0;
  if (tf::g_ts_dev && blockIdx.x == 0 && threadIdx.x == 0 && !TAP1 && G::HIN == 32 && G::S == 2) { tf::g_ts_dev[446] = nbands; tf::g_ts_dev[447] = jb.nz; }
#else
#define WG_TS(slot) do { } while (0)
#endif
  WG_TS(0);
#ifdef MLHOT_TS
  if (tf::g_ts_dev && threadIdx.x == 0 && !TAP1 && G::HIN == 32 && G::S == 2 && blockIdx.x < 1024) tf::g_ts_dev[1024 + 2 * blockIdx.x] = wall_clock64();
#endif
  if (z < nbands) fetch(z);
#pragma unroll 1
  for (int band = z; band < nbands; band += jb.nz) {
    WG_TS(1 + 4 * ts_b);
    __syncthreads();                   // the previous band's operands have been consumed
    WG_TS(2 + 4 * ts_b);
    if (TAP1) {
#pragma unroll
      for (int j = 0; j < S1N; ++j) { const int e = tid + j * 256; xs[(e / G::BPOS) * (G::BPOS + 1) + e % G::BPOS] = s1[j]; }
    } else if constexpr (REG_MAP) {
      float* dl = xs + rm_ci * G::PS + rm_row0 * G::RS + 1 + 4 * rm_seg;
#pragma unroll
      for (int j = 0; j < RM_RPP; ++j) {
        if (rm_row0 + j < G::PR) {
          float* d = dl + j * G::RS;
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = st[j][k];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < G::CNT; ++j) {
        const int e = tid + j * 256;
        const int seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, il = (row / G::PR) % G::NI, ci = row / (G::PR * G::NI);
        if (e < G::ITEMS) {
          float* d = xs + ci * G::PS + il * G::ISZ + pr * G::RS + 1 + G::SEGW * seg;
#pragma unroll
          for (int k = 0; k < G::SEGW; ++k) d[k] = st[j][k];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < G::DCNT; ++j) {
      const int e = tid + j * 256, co = e & 63, p4 = 4 * (e >> 6);
      float* d = dyt + p4 * G::DS + co;
      d[0] = sd[j].x; d[G::DS] = sd[j].y; d[2 * G::DS] = sd[j].z; d[3 * G::DS] = sd[j].w;
    }
    __syncthreads();
    WG_TS(3 + 4 * ts_b);
    if (band + jb.nz < nbands) fetch(band + jb.nz);

    // operands of k-step ks + 1 (one dY word, nine / one x words) are read before the MFMAs of k-step ks, every step fenced:
    // left alone, hipcc reads a step's ten operands right in front of its MFMAs and waits lgkmcnt(0) on them
    {
      float ar[2], br[2][NT];
      auto rd = [&](int ks, int slot) {
        ar[slot] = dyt[abase + G::pb(ks, 0) * G::DS];
        const float* bp = xs + bbase + (TAP1 ? G::pb(ks, 0) : G::posoff(G::pb(ks, 0)));
#pragma unroll
        for (int t = 0; t < NT; ++t) br[slot][t] = TAP1 ? bp[0] : bp[(t / 3) * G::RS + t % 3];
      };
      rd(0, 0);
#pragma unroll
      for (int ks = 0; ks < G::NKS_B; ++ks) {
        if (ks + 1 < G::NKS_B) rd(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const float a = ar[ks & 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma4(a, br[ks & 1][t], acc[t]);
        if (q == 0) accb = mfma4(a, 1.f, accb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef MLHOT_TS
    WG_TS(4 + 4 * ts_b); ++ts_b;
#endif
  }
  WG_TS(24);
  // ---- slab row z0 + z, MFMA-native: [q][w][tap][lane][4] ----
  float* row = jb.slab + (size_t)(jb.z0 + z) * (TAP1 ? SLAB1 : SLAB3);
#pragma unroll
  for (int t = 0; t < NT; ++t)
    *reinterpret_cast<float4*>(row + ((((size_t)q * 4 + w) * NT + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  if (q == 0 && jb.slab_b && lr == 0)
    *reinterpret_cast<float4*>(jb.slab_b + (size_t)(jb.z0 + z) * 64 + 16 * w + 4 * lq) = make_float4(accb[0], accb[1], accb[2], accb[3]);
  WG_TS(25);
#ifdef MLHOT_TS
  if (tf::g_ts_dev && threadIdx.x == 0 && !TAP1 && G::HIN == 32 && G::S == 2 && blockIdx.x < 1024) tf::g_ts_dev[1025 + 2 * blockIdx.x] = wall_clock64();
#endif
}

template <class G, bool TAP1>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgJobs jobs) {
  __shared__ float lds[G::LDS];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const WgJob& jb = jobs.j[ji];
  wgrad_body<G, TAP1>(jb, lds, (int)blockIdx.x - jb.wg0);
}

// The 3x3 weight gradients of blocks 3 and 4 of a 64 x 64 trunk (conv1 / 3x3 skip / conv2 of both blocks, every pass: up to 16 jobs
// of four geometries) in ONE launch: each was a launch of 10-15 us for a few hundred short workgroups (c5: 48 us in four launches).
constexpr int WG34_MAX = 16;
struct Wg34Jobs { WgJob j[WG34_MAX]; unsigned char geo[WG34_MAX]; int n; };      // geo: 0 = 8x8 s2, 1 = 4x4 s1, 2 = 4x4 s2, 3 = 2x2 s1
__global__ __launch_bounds__(256, 2) void wgrad34_kernel(const Wg34Jobs jobs) {
  __shared__ float lds[cmax(cmax(W8s2::LDS, W4s1::LDS), cmax(W4s2::LDS, W2s1::LDS))];
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const WgJob& jb = jobs.j[ji];
  const int rel = (int)blockIdx.x - jb.wg0;
  switch (jobs.geo[ji]) {
    case 0: wgrad_body<W8s2, false>(jb, lds, rel); break;
    case 1: wgrad_body<W4s1, false>(jb, lds, rel); break;
    case 2: wgrad_body<W4s2, false>(jb, lds, rel); break;
    default: wgrad_body<W2s1, false>(jb, lds, rel); break;
  }
}
inline int wgrad34_launch(Wg34Jobs& jobs, hipStream_t s, const char* what) {
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) { jobs.j[i].wg0 = wg; wg += 4 * jobs.j[i].nz; }
  if (wg <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL(wgrad34_kernel, dim3(wg), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}

// out = sum over slab rows, un-permuted from MFMA-native order to the parameter's own layout.
//   kind 0: 3x3 weight  dW[co][ci][tap]   from rows of SLAB3 floats ([q][w][tap][lane][r]: co = 16w + 4(lane>>4) + r, ci = 16q + (lane&15))
//   kind 1: 1x1 weight  dW[co][ci]        from rows of SLAB1 floats
//   kind 2: bias        db[co]            from rows of 64 floats (plain)
//   kind 3: stem weight dW[co][K] (+ db)  from rows of 4 * NJ * 256 floats ([w][j][lane][r]: co = 16w + 4(lane>>4) + r, n = 16j + (lane&15))
// 32 bytes per segment so that every fold of a backward (3 weight sets x 31 tensors for BASELINE c5) fits ONE launch's 4 KB of
// kernel arguments: the slab is an offset into the caller's scratch block.
struct WsumSeg { float* out; float* out_b; unsigned slab_off; int first; unsigned short nrows; unsigned char kind, K; };
constexpr int WSUM_MAX = 100;
struct WsumSegs { WsumSeg s[WSUM_MAX]; const float* base; int n; int blocks; };
static_assert(sizeof(WsumSegs) <= 4000, "kernel argument block");
__global__ __launch_bounds__(256) void wsum_kernel(const WsumSegs segs) {
  // 256 threads = 64 float4 columns x 4 row lanes: lane g sums rows g, g + 4, ... (4 loads in flight), fixed-order fold through LDS
  __shared__ float4 sm[4][64];
  int si = 0;
  while (si + 1 < segs.n && (int)blockIdx.x >= segs.s[si + 1].first) ++si;
  const WsumSeg sg = segs.s[si];
  const int nj = sg.kind == 3 ? (sg.K + 1 + 15) / 16 : 0;
  const int rowlen = sg.kind == 0 ? SLAB3 : sg.kind == 1 ? SLAB1 : sg.kind == 2 ? 64 : 4 * nj * 256;
  const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e4 = ((int)blockIdx.x - sg.first) * 64 + cx;        // float4 index inside a row
  const bool in = e4 * 4 < rowlen;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  auto add = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
  if (in) {
    const float* p = segs.base + sg.slab_off + (size_t)e4 * 4;
    int z = g;
    for (; z + 12 < sg.nrows; z += 16) {
      add(s0, *reinterpret_cast<const float4*>(p + (size_t)z * rowlen));
      add(s1, *reinterpret_cast<const float4*>(p + (size_t)(z + 4) * rowlen));
      add(s2, *reinterpret_cast<const float4*>(p + (size_t)(z + 8) * rowlen));
      add(s3, *reinterpret_cast<const float4*>(p + (size_t)(z + 12) * rowlen));
    }
    for (; z < sg.nrows; z += 4) add(s0, *reinterpret_cast<const float4*>(p + (size_t)z * rowlen));
  }
  add(s0, s1); add(s2, s3); add(s0, s2);
  sm[g][cx] = s0;
  __syncthreads();
  if (g != 0 || !in) return;
  add(s0, sm[1][cx]); add(s0, sm[2][cx]); add(s0, sm[3][cx]);
  const float v[4] = {s0.x, s0.y, s0.z, s0.w};
  if (sg.kind == 2) { *reinterpret_cast<float4*>(sg.out + e4 * 4) = s0; return; }
  const int lane = e4 & 63, lr = lane & 15, lq = lane >> 4;
  if (sg.kind == 3) {
    const int j = (e4 >> 6) % nj, w = (e4 >> 6) / nj, n = 16 * j + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = 16 * w + 4 * lq + r;
      if (n < sg.K) sg.out[(size_t)co * sg.K + n] = v[r];
      else if (n == sg.K && sg.out_b) sg.out_b[co] = v[r];
    }
    return;
  }
  const int nt = sg.kind == 0 ? 9 : 1;
  const int t = (e4 >> 6) % nt, w = ((e4 >> 6) / nt) & 3, q = (e4 >> 6) / (nt * 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) sg.out[((size_t)(16 * w + 4 * lq + r) * CH + 16 * q + lr) * nt + t] = v[r];
}
inline bool wsum_add(WsumSegs& segs, const float* slab, float* out, float* out_b, int nrows, int kind, int K = 0) {
  const int nj = kind == 3 ? (K + 1 + 15) / 16 : 0;
  const int rowlen = kind == 0 ? SLAB3 : kind == 1 ? SLAB1 : kind == 2 ? 64 : 4 * nj * 256;
  const size_t off = (size_t)(slab - segs.base);
  if (segs.n >= WSUM_MAX || nrows > 65535 || K > 255 || slab < segs.base || off > 0xffffffffu) return false;
  WsumSeg& sg = segs.s[segs.n++];
  sg = WsumSeg{out, out_b, (unsigned)off, segs.blocks, (unsigned short)nrows, (unsigned char)kind, (unsigned char)K};
  segs.blocks += (rowlen / 4 + 63) / 64;
  return true;
}

// ---- weight (+ bias) gradient of the plain ResNet block's 1x1 stride-2 skip convolution, every block of a step in ONE launch ----
// dW[co][ci] = sum over (image, oy, ox) of dy[co][oy][ox] * x[ci][2 oy][2 ox]: 0.3 GFLOP in all for BASELINE c5's decoder pass, so the
// four per-block launches of wgrad_kernel<G, true> (17 us each, latency bound) become one run-time-shaped kernel: a job per
// (block, pass), positions flattened over (image, oy, ox) and cut into chunks of 64; a workgroup (4 waves = 4 output-channel
// tiles, 4 input-channel tiles each) walks chunks z, z + nz, ...: both operands staged [channel][position] (row stride 66: the
// 16 channels x 2 positions of a half-wave land on 32 banks), slab rows in the same MFMA-native order as wgrad_kernel<G, true>.
struct Sk1Job {
  const float* x; const float* dy;     // x [n_img][64][2 HO][2 HO], dy [n_img][64][HO][HO]
  float* slab; float* slab_b;          // rows [z][SLAB1]; bias rows [z][64] (may be null)
  int n_img, ho_log2, z0, nz, wg0;
};
constexpr int SK1_MAX = 4 * MAX_JOBS, SK1_PS = 66;      // 4 blocks x the passes of a step
struct Sk1Jobs { Sk1Job j[SK1_MAX]; int n; };
__global__ __launch_bounds__(256) void skip1_wgrad_kernel(const Sk1Jobs jobs) {
  __shared__ float xs[CH * SK1_PS], ds[CH * SK1_PS];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const Sk1Job& jb = jobs.j[ji];
  const int z = (int)blockIdx.x - jb.wg0;
  const int lg = jb.ho_log2, HO = 1 << lg, HIN = 2 * HO, lpi = 2 * lg;          // positions per image = 1 << lpi
  const int total = jb.n_img << lpi, nchunks = (total + 63) >> 6;
  f32x4_t acc[4], accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const bool bias = jb.slab_b != nullptr;
#pragma unroll 1
  for (int c = z; c < nchunks; c += jb.nz) {
    float sx[16], sd[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = tid + j * 256, pos = e & 63, ch = e >> 6;
      const int p = c * 64 + pos, img = p >> lpi, pin = p & ((1 << lpi) - 1), oy = pin >> lg, ox = pin & (HO - 1);
      const bool ok = p < total;
      sd[j] = ok ? jb.dy[(((size_t)img * CH + ch) << lpi) + pin] : 0.f;
      sx[j] = ok ? jb.x[(((size_t)img * CH + ch) * HIN + 2 * oy) * HIN + 2 * ox] : 0.f;
    }
    __syncthreads();                   // the previous chunk's operands have been consumed
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = tid + j * 256, pos = e & 63, ch = e >> 6;
      ds[ch * SK1_PS + pos] = sd[j]; xs[ch * SK1_PS + pos] = sx[j];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a = ds[(16 * w + lr) * SK1_PS + 4 * ks + lq];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = mfma4(a, xs[(16 * q + lr) * SK1_PS + 4 * ks + lq], acc[q]);
      if (bias) accb = mfma4(a, 1.f, accb);
    }
  }
  float* row = jb.slab + (size_t)(jb.z0 + z) * SLAB1;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(row + (((size_t)q * 4 + w) * 64 + lane) * 4) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
  if (bias && lr == 0)
    *reinterpret_cast<float4*>(jb.slab_b + (size_t)(jb.z0 + z) * 64 + 16 * w + 4 * lq) = make_float4(accb[0], accb[1], accb[2], accb[3]);
}
inline int skip1_wgrad_launch(Sk1Jobs& jobs, hipStream_t s, const char* what) {
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) { jobs.j[i].wg0 = wg; wg += jobs.j[i].nz; }
  if (wg <= 0) return MLHOT_OK;
  {
    ProfScope pr(what, s);
    hipLaunchKernelGGL(skip1_wgrad_kernel, dim3(wg), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}

// Position splits of one job: enough workgroups to fill the chip twice over all jobs, at least 2 bands per workgroup when there are many
template <class G>
inline int wgrad_bands(int n_img) { return G::MULTI ? (n_img + G::NI - 1) / G::NI : n_img * G::BANDS_PER_IMG; }
template <class G, bool TAP1>
inline int launch_wgrad(WgJobs& jobs, hipStream_t s, const char* what) {
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) { jobs.j[i].wg0 = wg; wg += 4 * jobs.j[i].nz; }
  if (wg <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL((wgrad_kernel<G, TAP1>), dim3(wg), dim3(256), 0, s, jobs);
  }
  return check_launch(what);
}
// number of slab rows (position splits) a job of n_img images should use when `share` of the chip's 128 (x4 channel tiles) slots are its own
inline int wgrad_bands_rt(int HIN, int S, int n_img) {
  const int PI = (HIN / S) * (HIN / S);
  return PI >= 128 ? n_img * (PI / 128) : (n_img + 128 / PI - 1) / (128 / PI);
}
inline int wgrad_dispatch(int HIN, int S, bool tap1, WgJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
#define MLHOT_RW_CASE(H, ST, GEO) if (HIN == H && S == ST) return tap1 ? launch_wgrad<GEO, true>(jobs, s, what) : launch_wgrad<GEO, false>(jobs, s, what);
  MLHOT_RW_CASE(64, 2, W64s2) MLHOT_RW_CASE(32, 1, W32s1) MLHOT_RW_CASE(32, 2, W32s2) MLHOT_RW_CASE(16, 1, W16s1) MLHOT_RW_CASE(16, 2, W16s2)
  MLHOT_RW_CASE(8, 1, W8s1) MLHOT_RW_CASE(8, 2, W8s2) MLHOT_RW_CASE(4, 1, W4s1) MLHOT_RW_CASE(4, 2, W4s2) MLHOT_RW_CASE(2, 1, W2s1)
#undef MLHOT_RW_CASE
  return MLHOT_ERR_UNSUPPORTED;
}

// ---- stem: 5x5 stride-2 pad-2 convolution C -> 64 channels (+ bias + ReLU) -------------------------------------------
// K = 25 C (75 for the 3-channel ShapeNet3D images, 25 for the Distractor's grey images) padded to a multiple of 4; a band =
// 256 output positions (16 M-tiles: the accumulators take 64 registers, the weights 19 / 7); lane lq of k-step ks reads
// patch[ci][2oy + ky][2ox + kx] of k = 4ks + lq = (ci, ky, kx) through a per-lane table of 19 / 7 LDS offsets.
struct StemJob { const float* x; const float* wimg; const float* b; float* y; int n_img, wg0, nwg; };
struct StemJobs { StemJob j[MAX_JOBS]; int n; };

template <int C, int HIN>
struct StemGeo {
  static constexpr int K = 25 * C, NKS = (K + 3) / 4, HO = HIN / 2, WO = HO, PI = HO * WO;
  static constexpr int BPOS = 256, RB = BPOS / WO, NACC = 16, BANDS_PER_IMG = PI / BPOS;
  // col 0 = ix -2.  RS = 5 and PS = 25 (mod 32): consecutive k = (ci, ky, kx) then always sit 1 bank apart (kx + 1; the row wrap
  // RS - 4; the channel wrap PS - 4 RS - 4), so the forward's (2 lr + k) and the weight gradient's (k, +16 lq) gathers are conflict-free
  static constexpr int PR = 2 * (RB - 1) + 5, RS = HIN + 5, PS = PR * RS + (25 + 32 - (PR * RS) % 32) % 32, PATCH = C * PS;
  static_assert(RS % 32 == 5 && PS % 32 == 25, "stem strides");
  static constexpr int SEG = HIN / 4, ITEMS = C * PR * SEG, CNT = (ITEMS + 255) / 256;
  static_assert(WO >= 16 && BPOS % WO == 0 && PI % BPOS == 0, "stem geometry");
};

template <int C, int HIN>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemJobs jobs) {
  typedef StemGeo<C, HIN> G;
  __shared__ __attribute__((aligned(16))) float patch[G::PATCH];
  const int tid = threadIdx.x, lane = tid & 63, nt = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const StemJob& jb = jobs.j[ji];
  const int co = 16 * nt + lr;
  float wr[G::NKS];
  int koff[G::NKS];
#pragma unroll
  for (int ks = 0; ks < G::NKS; ++ks) {
    wr[ks] = jb.wimg[((size_t)nt * G::NKS + ks) * 64 + lane];
    const int k = 4 * ks + lq, ci = k / 25, t = k % 25;
    koff[ks] = 2 * lr + (k < G::K ? ci * G::PS + (t / 5) * G::RS + t % 5 : 0);
  }
  const float bn = jb.b ? jb.b[co] : 0.f;
  for (int i = tid; i < G::PATCH; i += 256) patch[i] = 0.f;
  const int nbands = jb.n_img * G::BANDS_PER_IMG;
#pragma unroll 1
  for (int band = (int)blockIdx.x - jb.wg0; band < nbands; band += jb.nwg) {
    const int img = band / G::BANDS_PER_IMG, oy0 = (band % G::BANDS_PER_IMG) * G::RB;
    float4 st[G::CNT];
#pragma unroll
    for (int j = 0; j < G::CNT; ++j) {
      const int e = tid + j * 256, seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, ci = row / G::PR;
      const int iy = 2 * oy0 - 2 + pr;
      st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < G::ITEMS && iy >= 0 && iy < HIN) st[j] = *reinterpret_cast<const float4*>(jb.x + (((size_t)img * C + ci) * HIN + iy) * HIN + 4 * seg);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < G::CNT; ++j) {
      const int e = tid + j * 256, seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, ci = row / G::PR;
      if (e < G::ITEMS) {
        float* d = patch + ci * G::PS + pr * G::RS + 2 + 4 * seg;
        d[0] = st[j].x; d[1] = st[j].y; d[2] = st[j].z; d[3] = st[j].w;
      }
    }
    __syncthreads();
    f32x4_t acc[G::NACC];
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::NKS; ++ks) {
      const float* base = patch + koff[ks];
#pragma unroll
      for (int t = 0; t < G::NACC; ++t) {
        const int row = t / (G::WO / 16), cb = t % (G::WO / 16);
        acc[t] = mfma4(base[2 * row * G::RS + 32 * cb], wr[ks], acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < G::NACC; ++t) {
      const int row = t / (G::WO / 16), cb = t % (G::WO / 16);
      const size_t o = (((size_t)img * CH + co) * G::HO + oy0 + row) * G::WO + 16 * cb + 4 * lq;
      *reinterpret_cast<float4*>(jb.y + o) = make_float4(fmaxf(acc[t][0] + bn, 0.f), fmaxf(acc[t][1] + bn, 0.f), fmaxf(acc[t][2] + bn, 0.f),
                                                         fmaxf(acc[t][3] + bn, 0.f));
    }
  }
}

inline bool stem_supported(int C, int HIN) { return (C == 3 && HIN == 64) || (C == 1 && HIN == 128); }
inline int stem_dispatch(int C, int HIN, StemJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
  const int per_img = (HIN / 2) * (HIN / 2) / 256;
  int nb[MAX_JOBS], total = 0, wg = 0;
  for (int i = 0; i < jobs.n; ++i) { nb[i] = jobs.j[i].n_img * per_img; total += nb[i]; }
  for (int i = 0; i < jobs.n; ++i) {
    int share = total <= WG_SLOTS ? nb[i] : (int)((long)WG_SLOTS * nb[i] / total);
    if (share < 1) share = 1;
    if (share > nb[i]) share = nb[i];
    jobs.j[i].wg0 = wg; jobs.j[i].nwg = share; wg += share;
  }
  if (wg <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    if (C == 3 && HIN == 64) hipLaunchKernelGGL((stem_kernel<3, 64>), dim3(wg), dim3(256), 0, s, jobs);
    else if (C == 1 && HIN == 128) hipLaunchKernelGGL((stem_kernel<1, 128>), dim3(wg), dim3(256), 0, s, jobs);
    else return MLHOT_ERR_UNSUPPORTED;
  }
  return check_launch(what);
}

// ---- stem weight gradient: dW[co][(ci, ky, kx)] and db[co] ------------------------------------------------------------
// M = co (wave w = 16 output channels), N = K + 1 columns (column K = the bias gradient, against an all-ones operand) in NJ
// tiles of 16, K-dim = positions: a band = the forward's 256 positions (same image patch in LDS); dy staged transposed
// ([pos][co], stride 66); a k-step's four positions are 8 columns apart (bank offsets 2*8 = 16 for x, 8*66 = 16 mod 32 for dy).
// Each workgroup owns slab row z (MFMA-native [w][j][lane][4]); wsum_kernel (kind 3) folds and un-permutes.
struct StemWgJob { const float* x; const float* dy; float* slab; int n_img, z0, nz, wg0; };
struct StemWgJobs { StemWgJob j[MAX_JOBS]; int n; };

template <int C, int HIN>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemWgJobs jobs) {
  typedef StemGeo<C, HIN> G;
  constexpr int NJ = (G::K + 1 + 15) / 16, DS = 66, HALF = G::BPOS / 2, NKS_H = HALF / 4;
  __shared__ float lds[G::PATCH + HALF * DS];       // dy is staged in two halves of 128 positions (two workgroups per CU)
  float* patch = lds;
  float* dyt = lds + G::PATCH;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lq = lane >> 4;
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.j[ji + 1].wg0) ++ji;
  const StemWgJob& jb = jobs.j[ji];
  const int z = (int)blockIdx.x - jb.wg0;
  // column n = 16j + lr of this lane: patch offset of (ci, ky, kx), or the ones / zero column
  int noff[NJ], nkind[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int n = 16 * j + lr, ci = n / 25, t = n % 25;
    nkind[j] = n < G::K ? 0 : (n == G::K ? 1 : 2);
    noff[j] = n < G::K ? ci * G::PS + (t / 5) * G::RS + t % 5 : 0;
  }
  f32x4_t acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < G::PATCH + HALF * DS; i += 256) lds[i] = 0.f;
  const int nbands = jb.n_img * G::BANDS_PER_IMG;
#pragma unroll 1
  for (int band = z; band < nbands; band += jb.nz) {
    const int img = band / G::BANDS_PER_IMG, oy0 = (band % G::BANDS_PER_IMG) * G::RB;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      float4 st[G::CNT], sd[8];
      if (half == 0) {
#pragma unroll
        for (int j = 0; j < G::CNT; ++j) {
          const int e = tid + j * 256, seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, ci = row / G::PR;
          const int iy = 2 * oy0 - 2 + pr;
          st[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < G::ITEMS && iy >= 0 && iy < HIN) st[j] = *reinterpret_cast<const float4*>(jb.x + (((size_t)img * C + ci) * HIN + iy) * HIN + 4 * seg);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {           // 64 co x 128 positions = 2048 float4; channels along the lanes (see wgrad_kernel: conflict-free transposition)
        const int e = tid + j * 256, co = e & 63, p4 = 4 * (e >> 6);
        sd[j] = *reinterpret_cast<const float4*>(jb.dy + ((size_t)img * CH + co) * G::PI + oy0 * G::WO + half * HALF + p4);
      }
      __syncthreads();
      if (half == 0) {
#pragma unroll
        for (int j = 0; j < G::CNT; ++j) {
          const int e = tid + j * 256, seg = e % G::SEG, row = e / G::SEG, pr = row % G::PR, ci = row / G::PR;
          if (e < G::ITEMS) {
            float* d = patch + ci * G::PS + pr * G::RS + 2 + 4 * seg;
            d[0] = st[j].x; d[1] = st[j].y; d[2] = st[j].z; d[3] = st[j].w;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = tid + j * 256, co = e & 63, p4 = 4 * (e >> 6);
        float* d = dyt + p4 * DS + co;
        d[0] = sd[j].x; d[DS] = sd[j].y; d[2 * DS] = sd[j].z; d[3 * DS] = sd[j].w;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < NKS_H; ++ks) {
        // positions of this k-step: a run of 32 positions (one or half an output row), lane group lq takes its columns 8lq + j8
        const int blk = ks / 8, j8 = ks % 8;
        const int ph = blk * 32 + 8 * lq + j8, pb = half * HALF + ph;
        const int oy = pb / G::WO, ox = pb % G::WO;
        const float a = dyt[ph * DS + 16 * w + lr];
        const float* bp = patch + 2 * oy * G::RS + 2 * ox;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          float b = bp[noff[j]];
          if (16 * j + 15 >= G::K) b = nkind[j] == 0 ? b : (nkind[j] == 1 ? 1.f : 0.f);
          acc[j] = mfma4(a, b, acc[j]);
        }
        if (ks % 4 == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float* row = jb.slab + (size_t)(jb.z0 + z) * (4 * NJ * 256);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    *reinterpret_cast<float4*>(row + (((size_t)w * NJ + j) * 64 + lane) * 4) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}
inline int stem_slab_row(int C) { return 4 * ((25 * C + 1 + 15) / 16) * 256; }
inline int stem_wgrad_dispatch(int C, int HIN, StemWgJobs& jobs, hipStream_t s, const char* what) {
  if (jobs.n <= 0) return MLHOT_OK;
  int wg = 0;
  for (int i = 0; i < jobs.n; ++i) { jobs.j[i].wg0 = wg; wg += jobs.j[i].nz; }
  if (wg <= 0) return MLHOT_OK;
  {
    ProfScope ps(what, s);
    if (C == 3 && HIN == 64) hipLaunchKernelGGL((stem_wgrad_kernel<3, 64>), dim3(wg), dim3(256), 0, s, jobs);
    else if (C == 1 && HIN == 128) hipLaunchKernelGGL((stem_wgrad_kernel<1, 128>), dim3(wg), dim3(256), 0, s, jobs);
    else return MLHOT_ERR_UNSUPPORTED;
  }
  return check_launch(what);
}

}  // namespace rw
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
