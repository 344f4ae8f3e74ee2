// Generic implicit-GEMM engine on the gfx950 fp32 matrix cores.
//
//   out(m, n) = sum_k  A(m, k) * B(k, n)            m < M, n < N, k < K
//
// A "problem" P is a small POD functor that maps logical GEMM indices to tensor
// elements (im2col gather, stride-2 parity classes, pooled-gradient routing, blocked
// head weights, fused bias / activation / ReLU-mask epilogues).  The kernel stages
// BM x BK / BK x BN tiles of A(.,.) / B(.,.) through LDS and contracts them with
// v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered fma chain).  Operand lane maps
// (cdna_hip_programming.md §3): A lane l holds A[row l&15][k l>>4], B lane l holds
// B[k l>>4][col l&15], C/D lane l reg r holds C[row 4*(l>>4)+r][col l&15].
//
// LDS images are laid out so both halves of a 32-lane ds_read_b32 group hit disjoint
// banks: [k][m] rows with a leading dimension == 16 (mod 32), or (for operands whose
// global image is k-contiguous) [m][k] rows of BK+1 floats.
//
// Split-K: grid.z slices the reduction; partial tiles go to a [split][M][N] slab and a
// second kernel sums the slab in a fixed order (bitwise reproducible, no atomics).
//
// Index hoisting: a gather problem may additionally describe how its indices factor (typedefs KEnt / RCtx / CCtx
// with kent(k), rctx(m), cctx(n), A2(rctx, kent, m, k), B2(kent, cctx, k, n)).  A thread's staging slots keep the
// same tile row / column for the whole kernel, so rctx / cctx are evaluated ONCE per slot, and the k-dependent
// part once per k of a tile by BK threads (through LDS), instead of 5-8 run-time integer divisions for every
// gathered element of every tile - the run-time-shaped convolutions were bound by that index arithmetic.
#pragma once
#include <type_traits>
#include "common.h"

namespace mlhot {

template <class P>
struct SlabReduceArgs {
  P p;
  const float* slab;
  int nsplit;
};

template <class P> struct IgemmBatch { P p[4]; int n; };      // up to 4 problems of one type for one launch (run_igemm_batch)

#ifndef MLHOT_HOSTSIM

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class P, class = void> struct igemm_hoists : std::false_type {};
template <class P> struct igemm_hoists<P, std::void_t<typename P::KEnt>> : std::true_type {};
struct IgemmNoCtx {};
template <class P, bool F> struct igemm_ctx { typedef IgemmNoCtx K; typedef IgemmNoCtx R; typedef IgemmNoCtx C; typedef float AR; typedef float BR; };
template <class P> struct igemm_ctx<P, true> {
  typedef typename P::KEnt K; typedef typename P::RCtx R; typedef typename P::CCtx C;
  typedef typename P::ARaw AR; typedef typename P::BRaw BR;    // what a gather leaves in registers until the tile is stashed
};

// a 16-byte k-entry out of LDS as ONE ds_read_b128 (field-wise reads get sunk into per-element branches by the compiler)
template <class K>
__device__ __forceinline__ K lds_ent(const K* p) {
  if constexpr (sizeof(K) == 16) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 v = *reinterpret_cast<const i32x4*>(p);
    K k;
    __builtin_memcpy(&k, &v, 16);
    return k;
  } else {
    return *p;
  }
}

constexpr int lds_ld16(int b) { return (b % 32 == 16) ? b : b + 16; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// One output tile of problem `p`: tile (blockIdx.x, blockIdx.y), k range [kz * k_chunk, (kz + 1) * k_chunk).
template <class P, int BM, int BN, int BK, int WM, int WN, bool SPLIT>
__device__ __forceinline__ void igemm_tile(const P& p, float* __restrict__ slab, int k_chunk, int kz) {
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(BM % (WM * 16) == 0 && BN % (WN * 16) == 0 && BK % 4 == 0, "tile shape");
  constexpr bool AK = P::A_ALONG_K, BKc = P::B_ALONG_K;
  constexpr int LDA = lds_ld16(BM), LDB = lds_ld16(BN), LDK = BK + 1;
  constexpr int A_SZ = AK ? BM * LDK : BK * LDA;
  constexpr int B_SZ = BKc ? BN * LDK : BK * LDB;
  __shared__ float As[A_SZ];
  __shared__ float Bs[B_SZ];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kb = kz * k_chunk;
  const int ke = (kb + k_chunk < p.K) ? kb + k_chunk : p.K;

  constexpr int EA = (BM * BK + NT - 1) / NT, EB = (BN * BK + NT - 1) / NT;
  constexpr bool FAST = igemm_hoists<P>::value;
  typedef igemm_ctx<P, FAST> CT;
  typename CT::AR ra[EA];
  typename CT::BR rb[EB];
  __shared__ typename CT::K ktab[2][FAST ? BK : 1];
  typename CT::R rctx[EA];
  typename CT::C cctx[EB];
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int e = tid + i * NT, mm = AK ? e / BK : e % BM;
      const int m = m0 + mm;
      rctx[i] = p.rctx(m < p.M ? m : 0);
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int e = tid + i * NT, nn = BKc ? e / BK : e % BN;
      const int n = n0 + nn;
      cctx[i] = p.cctx(n < p.N ? n : 0);
    }
  }
  int kt = 0;                       // which half of ktab the NEXT fetch reads
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  static_assert(EA <= 32 && EB <= 32, "tile-bounds masks are one word per operand");
  unsigned amask = 0, bmask = 0;
  auto fetch = [&](int k0) {
    amask = bmask = 0;
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int e = tid + i * NT;
      const int mm = AK ? e / BK : e % BM, kk = AK ? e % BK : e / BM;
      const int m = m0 + mm, k = k0 + kk;
      const bool ok = e < BM * BK && m < p.M && k < ke;
      if constexpr (FAST) ra[i] = p.A2(rctx[i], lds_ent(&ktab[kt][kk]), m, k, ok);
      else {      // in-range indices for every lane, the tile-bounds mask is applied when the tile is stashed: no branch
        ra[i] = p.A(m < p.M ? m : p.M - 1, k < ke ? k : ke - 1);    // around (and no wait behind) the individual load
        amask |= (unsigned)ok << i;
      }
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int e = tid + i * NT;
      const int nn = BKc ? e / BK : e % BN, kk = BKc ? e % BK : e / BN;
      const int n = n0 + nn, k = k0 + kk;
      const bool ok = e < BN * BK && n < p.N && k < ke;
      if constexpr (FAST) rb[i] = p.B2(lds_ent(&ktab[kt][kk]), cctx[i], k, n, ok);
      else {
        rb[i] = p.B(k < ke ? k : ke - 1, n < p.N ? n : p.N - 1);
        bmask |= (unsigned)ok << i;
      }
    }
  };
  // k-dependent index parts of the tile starting at k0 -> ktab[half] (BK threads; published by the next barrier)
  auto ktile = [&](int k0, int half) {
    if constexpr (FAST) {
      if (tid < BK) { const int k = k0 + tid; ktab[half][tid] = p.kent(k < ke ? k : (ke > 0 ? ke - 1 : 0)); }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < EA; ++i) {
      const int e = tid + i * NT;
      const int mm = AK ? e / BK : e % BM, kk = AK ? e % BK : e / BM;
      if constexpr (FAST) { if (e < BM * BK) As[AK ? mm * LDK + kk : kk * LDA + mm] = p.Afin(ra[i]); }
      else if (e < BM * BK) As[AK ? mm * LDK + kk : kk * LDA + mm] = (amask >> i) & 1u ? ra[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      const int e = tid + i * NT;
      const int nn = BKc ? e / BK : e % BN, kk = BKc ? e % BK : e / BN;
      if constexpr (FAST) { if (e < BN * BK) Bs[BKc ? nn * LDK + kk : kk * LDB + nn] = p.Bfin(rb[i]); }
      else if (e < BN * BK) Bs[BKc ? nn * LDK + kk : kk * LDB + nn] = (bmask >> i) & 1u ? rb[i] : 0.f;
    }
  };

  const int lr = lane & 15, lk = lane >> 4;
  if constexpr (FAST) {
    ktile(kb, 0);
    __syncthreads();
  }
  if (kb < ke) fetch(kb);
  for (int k0 = kb; k0 < ke; k0 += BK) {
    stash();
    if (k0 + BK < ke) ktile(k0 + BK, kt ^ 1);
    __syncthreads();
    kt ^= 1;
    if (k0 + BK < ke) fetch(k0 + BK);  // next tile's gathers fly under this tile's MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int kk = ks * 4 + lk;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mm = (wm * TM + i) * 16 + lr;
        a[i] = As[AK ? mm * LDK + kk : kk * LDA + mm];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nn = (wn * TN + j) * 16 + lr;
        b[j] = Bs[BKc ? nn * LDK + kk : kk * LDB + nn];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * TM + i) * 16 + lk * 4 + r;
        const int n = n0 + (wn * TN + j) * 16 + lr;
        if (m < p.M && n < p.N) {
          if (SPLIT)
            slab[((size_t)kz * p.M + m) * p.N + n] = acc[i][j][r];
          else
            p.store(m, n, acc[i][j][r]);
        }
      }
}

template <class P, int BM, int BN, int BK, int WM, int WN, bool SPLIT>
__global__ __launch_bounds__(WM* WN * 64) void igemm_kernel(const P p, float* __restrict__ slab, int k_chunk) {
  igemm_tile<P, BM, BN, BK, WM, WN, SPLIT>(p, slab, k_chunk, blockIdx.z);
}

// Up to 4 independent problems of one type in ONE launch (blockIdx.z picks the problem, no split-K): the stride-2 data
// gradients are four small GEMMs, one per input-position parity class.
template <class P, int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void igemm_batch_kernel(const IgemmBatch<P> b) {
  const P& p = b.p[blockIdx.z];
  if ((int)blockIdx.x * BM >= p.M || (int)blockIdx.y * BN >= p.N) return;      // the grid is sized for the largest member
  igemm_tile<P, BM, BN, BK, WM, WN, false>(p, nullptr, (p.K + BK - 1) / BK * BK > 0 ? (p.K + BK - 1) / BK * BK : BK, 0);
}

// 32 outputs x 8 split-lanes per workgroup: each lane sums every 8th partial (coalesced across the
// 32 outputs), then a fixed-order LDS tree folds the 8 lanes -> deterministic and short chains.
template <class P>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const SlabReduceArgs<P> a) {
  __shared__ float sm[8][32];
  const size_t total = (size_t)a.p.M * a.p.N;
  const int ex = threadIdx.x & 31, zl = threadIdx.x >> 5;
  for (size_t e0 = (size_t)blockIdx.x * 32; e0 < total; e0 += (size_t)gridDim.x * 32) {
    const size_t e = e0 + ex;
    float s = 0.f;
    if (e < total)
      for (int z = zl; z < a.nsplit; z += 8) s += a.slab[(size_t)z * total + e];
    sm[zl][ex] = s;
    __syncthreads();
    if (zl == 0 && e < total) {
      const float t = ((sm[0][ex] + sm[1][ex]) + (sm[2][ex] + sm[3][ex])) + ((sm[4][ex] + sm[5][ex]) + (sm[6][ex] + sm[7][ex]));
      a.p.store((int)(e / a.p.N), (int)(e % a.p.N), t);
    }
    __syncthreads();
  }
}

#endif  // !MLHOT_HOSTSIM

// Bytes of slab a split-K launch of problem (M, N) needs.
inline size_t igemm_slab_bytes(int M, int N, int nsplit) { return nsplit > 1 ? (size_t)nsplit * M * N * sizeof(float) : 0; }

// Launch (or, in the hostsim flavour, evaluate on the host) one implicit GEMM.
template <class P, int BM, int BN, int BK, int WM, int WN>
int run_igemm(const P& p, int nsplit, float* slab, hipStream_t stream, const char* what) {
  if (p.M <= 0 || p.N <= 0) return MLHOT_OK;
#ifdef MLHOT_HOSTSIM
  (void)nsplit; (void)slab; (void)stream; (void)what;
  for (int m = 0; m < p.M; ++m)
    for (int n = 0; n < p.N; ++n) {
      float s = 0.f;
      for (int k = 0; k < p.K; ++k) s = fmaf(p.A(m, k), p.B(k, n), s);
      p.store(m, n, s);
    }
  return MLHOT_OK;
#else
  if (nsplit < 1) nsplit = 1;
  int k_chunk = (p.K + nsplit - 1) / nsplit;
  k_chunk = (k_chunk + BK - 1) / BK * BK;
  if (k_chunk < BK) k_chunk = BK;
  nsplit = (p.K + k_chunk - 1) / k_chunk;
  if (nsplit < 1) nsplit = 1;
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, nsplit);
  dim3 block(WM * WN * 64);
  if (nsplit == 1) {
    ProfScope ps(what, stream);
    hipLaunchKernelGGL((igemm_kernel<P, BM, BN, BK, WM, WN, false>), grid, block, 0, stream, p, (float*)nullptr, k_chunk);
    return check_launch(what);
  }
  if (slab == nullptr) {
    set_error("%s: split-K launch without a slab", what);
    return MLHOT_ERR_WORKSPACE;
  }
  {
    ProfScope ps(what, stream);
    hipLaunchKernelGGL((igemm_kernel<P, BM, BN, BK, WM, WN, true>), grid, block, 0, stream, p, slab, k_chunk);
  }
  int rc = check_launch(what);
  if (rc) return rc;
  ProfScope ps2("slab_reduce", stream);
  SlabReduceArgs<P> a{p, slab, nsplit};
  size_t total = (size_t)p.M * p.N;
  int rb = (int)((total + 31) / 32);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL((slab_reduce_kernel<P>), dim3(rb), dim3(256), 0, stream, a);
  return check_launch(what);
#endif
}

// Launch the members of `b` (same problem type, no split-K) together; hostsim evaluates them one after the other.
template <class P, int BM, int BN, int BK, int WM, int WN>
int run_igemm_batch(const IgemmBatch<P>& b, hipStream_t stream, const char* what) {
#ifdef MLHOT_HOSTSIM
  for (int i = 0; i < b.n; ++i) { int rc = run_igemm<P, BM, BN, BK, WM, WN>(b.p[i], 1, nullptr, stream, what); if (rc) return rc; }
  return MLHOT_OK;
#else
  int gx = 0, gy = 0;
  for (int i = 0; i < b.n; ++i) {
    const int x = (b.p[i].M + BM - 1) / BM, y = (b.p[i].N + BN - 1) / BN;
    gx = x > gx ? x : gx; gy = y > gy ? y : gy;
  }
  if (b.n <= 0 || gx == 0 || gy == 0) return MLHOT_OK;
  ProfScope ps(what, stream);
  hipLaunchKernelGGL((igemm_batch_kernel<P, BM, BN, BK, WM, WN>), dim3(gx, gy, b.n), dim3(WM * WN * 64), 0, stream, b);
  return check_launch(what);
#endif
}

// Tile choice for plain GEMM-shaped problems without split-K: 64 x 64 tiles, or 16-row tiles when those would occupy fewer
// than 128 workgroups (the attention / MLP layers of the ResNet-family models have 120-960 rows: a handful of 64-row tiles
// on a 256-CU machine); k tiles are 32 deep.
template <class P>
int run_igemm_auto(const P& p, hipStream_t s, const char* what) {
  const long wgs64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  if (wgs64 < 128) return run_igemm<P, 16, 64, 32, 1, 4>(p, 1, nullptr, s, what);
  return run_igemm<P, 64, 64, 32, 2, 2>(p, 1, nullptr, s, what);
}

}  // namespace mlhot
