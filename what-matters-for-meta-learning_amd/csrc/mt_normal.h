// torch's CPU normal_() random stream, continued on the device.
//
// The reference draws every Bayes-by-backprop eps on the torch CPU generator (bbb/BBBConv.py:86-95: `torch.empty(size).normal_(0, 1)`)
// and copies it to the device.  For BASELINE config c5 that is 896 k normals per step = 1.85 ms of one EPYC core - as long as the
// whole GPU step.  These two kernels produce the SAME stream on the GPU, so the draw costs the host nothing and the device one
// CU for ~0.3 ms beside the step:
//   1. mt_fill_kernel (ONE workgroup; the MT19937 recurrence is sequential from block to block): regenerates the 624-word state
//      block by block - three dependent runs of 227 / 227 / 170 words, each a plain parallel map, the new block written beside
//      the old one so that a run needs one barrier - tempers the words and writes the 24-bit uniforms (y & 0xffffff) * 2^-24
//      in stream order; leaves (state, left, next) exactly where ATen's MT19937RNGEngine would be after the same number of calls;
//   2. mt_box_muller_kernel (parallel): ATen's normal_fill - per group of 16 uniforms: u1 = 1 - x[j], u2 = x[j + 8],
//      r = sqrt(-2 log u1), th = 2 pi u2 -> x[j] = r cos th, x[j + 8] = r sin th; a tensor whose size is not a multiple of
//      16 takes 16 MORE outputs for its last 16 elements.
// The uniforms and the engine state are bit-identical to torch's; the normals differ from ATen's (Sleef u10 log / sincos) by the
// last few ulp of the device's logf / sincosf.  Checked against torch itself in tests/ (1 M draws, odd sizes, state round trip).
#pragma once
#include "common.h"
#include "../../include/mlhot.h"

namespace mlhot {
namespace mt {

constexpr int N = 624, M = 397;
struct Seg { long long dst, size, src, first_group; };     // one normal_() call: out[dst .. dst + size), stream offset src

#ifndef MLHOT_HOSTSIM
__device__ __forceinline__ unsigned twist(unsigned u, unsigned v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned temper(unsigned y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  return y ^ (y >> 18);
}

// engine: state[624], left, next (uint32 each).  u[0 .. total): the next `total` uniforms of the stream.
__global__ __launch_bounds__(256) void mt_fill_kernel(unsigned* __restrict__ engine, float* __restrict__ u, long long total) {
  __shared__ unsigned sa[N], sb[N];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += 256) sa[i] = engine[i];
  int left = (int)engine[N], next = (int)engine[N + 1];
  unsigned* cur = sa;
  unsigned* nxt = sb;
  __syncthreads();
  long long done = 0;
  while (done < total) {
    if (left <= 1) {                                 // `--left == 0` on the next call: regenerate the block first
      if (tid < N - M) nxt[tid] = cur[tid + M] ^ twist(cur[tid], cur[tid + 1]);                       // words 0 .. 226
      __syncthreads();
      if (tid < N - M) nxt[tid + N - M] = nxt[tid] ^ twist(cur[tid + N - M], cur[tid + N - M + 1]);   // words 227 .. 453
      __syncthreads();
      if (tid < 2 * M - N - 1) nxt[tid + 2 * (N - M)] = nxt[tid + N - M] ^ twist(cur[tid + 2 * (N - M)], cur[tid + 2 * (N - M) + 1]);   // 454 .. 622
      if (tid == 255) nxt[N - 1] = nxt[M - 1] ^ twist(cur[N - 1], nxt[0]);                            // the wrap-around word
      __syncthreads();
      unsigned* t = cur; cur = nxt; nxt = t;
      left = N + 1; next = 0;
    }
    const long long room = total - done;
    const int take = room < (long long)(left - 1) ? (int)room : left - 1;
    for (int k = tid; k < take; k += 256) u[done + k] = (float)(temper(cur[next + k]) & 0xffffffu) * (1.0f / 16777216.0f);
    done += take; left -= take; next += take;
  }
  __syncthreads();
  for (int i = tid; i < N; i += 256) engine[i] = cur[i];
  if (tid == 0) { engine[N] = (unsigned)left; engine[N + 1] = (unsigned)next; }
}

// ---- the same stream from K sub-streams (jump-ahead; host side: mlhot/mt_jump.py) ----------------------------------------
// The recurrence is linear over GF(2): with g^(k) = t^(624 S k) mod phi (phi: MT19937's characteristic polynomial, degree 19937)
// the block the engine will hold S k regenerations from now is  B_{Sk}[n] = XOR over { i : g^(k)_i = 1 } of x[n + i],  x = the
// raw words from the current block on (x[0..623] = the current block).  Three launches instead of one workgroup's ~1 ms:
//   mt_window_kernel (1 workgroup): x[0 .. 33 * 624): the current block and its next 32 regenerations (19937 + 624 words needed);
//   mt_jump_kernel   (K - 1 sub-streams x 8 parts): part p XORs the windows of the polynomial's bits [2496 p, 2496 p + 2496)
//                    out of an LDS copy of x[2496 p .. 2496 p + 3120) - 624 x ~1250 word XORs per workgroup;
//   mt_chunk_kernel  (K workgroups): sub-stream k folds its 8 partial blocks (k = 0: the engine's own block), regenerates its S
//                    blocks, tempers and writes their uniforms at their stream positions; the workgroup that makes the LAST block
//                    leaves (state, left, next) where the sequential stream would.
// The low 31 bits of a jumped block's word 0 are not defined by the recurrence (they are not part of MT19937's state); the
// regeneration reads only that word's top bit and a jumped block itself is never output - it was output by the sub-stream before.
// Measured (c5, 896 k outputs per draw, MI355X): 64 sub-streams 0.99 -> 0.16-0.20 ms per draw (window 20 + jump 99 + chunks 21 + Box-
// Muller 16 us), 8 sub-streams 0.26 ms, 4 sub-streams 0.42 ms.  The draw of step k + 1 runs beside step k's kernels; the one-
// workgroup form, 0.99 ms alone, stretches to ~1.45 ms there (its CU is shared with the trunk kernels) and was the floor of the c5
// step: 1.527 ms per step with one workgroup, **1.474 ms with 2-4 sub-streams** (the default: 4), 1.52 with 8, 1.60 with 64 (their
// 56 / 504 jump workgroups cost the trunk kernels more than the shorter draw returns).
constexpr int DEG = 19937, JBLK = 33, JWIN = JBLK * N, JPARTS = 8, JPW = N / JPARTS, JSPAN = 32 * JPW + N;    // 20592 words; 78 words = 2496 bits per part; 3120
static_assert(JWIN >= DEG + N && JPARTS * JPW == N && (JPARTS - 1) * 32 * JPW + JSPAN == JWIN, "jump window");

__device__ __forceinline__ void mt_regen(unsigned*& cur, unsigned*& nxt, int tid) {      // one block regeneration; ends in a barrier
  if (tid < N - M) nxt[tid] = cur[tid + M] ^ twist(cur[tid], cur[tid + 1]);
  __syncthreads();
  if (tid < N - M) nxt[tid + N - M] = nxt[tid] ^ twist(cur[tid + N - M], cur[tid + N - M + 1]);
  __syncthreads();
  if (tid < 2 * M - N - 1) nxt[tid + 2 * (N - M)] = nxt[tid + N - M] ^ twist(cur[tid + 2 * (N - M)], cur[tid + 2 * (N - M) + 1]);
  if (tid == 255) nxt[N - 1] = nxt[M - 1] ^ twist(cur[N - 1], nxt[0]);
  __syncthreads();
  unsigned* t = cur; cur = nxt; nxt = t;
}

__global__ __launch_bounds__(256) void mt_window_kernel(const unsigned* __restrict__ engine, unsigned* __restrict__ x) {
  __shared__ unsigned sa[N], sb[N];
  const int tid = threadIdx.x;
  for (int i = tid; i < N; i += 256) { sa[i] = engine[i]; x[i] = sa[i]; }
  if (tid < 2) x[JWIN + tid] = engine[N + tid];       // (left, next) as they were: the chunk kernel's last workgroup rewrites the engine while others may still start
  unsigned* cur = sa;
  unsigned* nxt = sb;
  __syncthreads();
  for (int b = 1; b < JBLK; ++b) {
    mt_regen(cur, nxt, tid);
    for (int i = tid; i < N; i += 256) x[b * N + i] = cur[i];
  }
}

// grid (K - 1, JPARTS): partial[(k - 1) * JPARTS + part][624]
__global__ __launch_bounds__(256) void mt_jump_kernel(const unsigned* __restrict__ x, const unsigned* __restrict__ polys, unsigned* __restrict__ partial) {
  __shared__ unsigned xs[JSPAN];
  const int tid = threadIdx.x, k1 = blockIdx.x, part = blockIdx.y;
  for (int i = tid; i < JSPAN; i += 256) xs[i] = x[part * 32 * JPW + i];
  __syncthreads();
  const unsigned* g = polys + (size_t)k1 * N + part * JPW;
  const bool has2 = tid + 512 < N;
  unsigned a0 = 0u, a1 = 0u, a2 = 0u;
  // a loop over the SET bits (half the reads of a branch-free walk over all 32 positions, which measured 131 us against this
  // loop's 91-99 us per workgroup: the LDS pipe, not the per-iteration round trip, is what both wait for)
  for (int w = 0; w < JPW; ++w) {
    unsigned gw = __builtin_amdgcn_readfirstlane(g[w]);          // the same word for every thread: the bit loop is scalar control flow
    while (gw) {
      const int i = 32 * w + __builtin_ctz(gw);
      gw &= gw - 1;
      a0 ^= xs[i + tid];
      a1 ^= xs[i + tid + 256];
      if (has2) a2 ^= xs[i + tid + 512];
    }
  }
  unsigned* o = partial + ((size_t)k1 * JPARTS + part) * N;
  o[tid] = a0; o[tid + 256] = a1;
  if (has2) o[tid + 512] = a2;
}

// grid K: sub-stream k makes blocks k S + 1 .. (k + 1) S of the nb new blocks this draw needs (block j's words are outputs
// r + 624 (j - 1) ..., r = the usable rest of the engine's current block, written by sub-stream 0)
__global__ __launch_bounds__(256) void mt_chunk_kernel(unsigned* __restrict__ engine, const unsigned* __restrict__ x, const unsigned* __restrict__ partial,
                                                       float* __restrict__ u, long long total, int S) {
  __shared__ unsigned sa[N], sb[N];
  const int tid = threadIdx.x, k = blockIdx.x;
  const int left0 = (int)x[JWIN], next0 = (int)x[JWIN + 1];      // the window kernel's copy: `engine` is rewritten by the workgroup of the last block
  const int r = left0 > 1 ? left0 - 1 : 0;                       // usable words of the current block (mt_fill_kernel's `take`)
  const long long fresh = total - r;                             // outputs that come out of new blocks
  const long long nb = fresh > 0 ? (fresh + N - 1) / N : 0;
  for (int i = tid; i < N; i += 256) {
    unsigned v;
    if (k == 0) v = x[i];
    else {
      const unsigned* p = partial + (size_t)(k - 1) * JPARTS * N + i;
      v = p[0] ^ p[N] ^ p[2 * N] ^ p[3 * N] ^ p[4 * N] ^ p[5 * N] ^ p[6 * N] ^ p[7 * N];
    }
    sa[i] = v;
  }
  unsigned* cur = sa;
  unsigned* nxt = sb;
  __syncthreads();
  if (k == 0) {
    const int take = total < (long long)r ? (int)total : r;
    for (int i = tid; i < take; i += 256) u[i] = (float)(temper(cur[next0 + i]) & 0xffffffu) * (1.0f / 16777216.0f);
    if (nb == 0 && tid == 0) { engine[N] = (unsigned)(left0 - take); engine[N + 1] = (unsigned)(next0 + take); }    // the draw ends inside the current block
  }
  for (int b = 1; b <= S; ++b) {
    const long long j = (long long)k * S + b;                    // 1-based index of the new block
    if (j > nb) break;                                           // uniform over the workgroup
    mt_regen(cur, nxt, tid);
    const long long pos = (long long)r + (j - 1) * N;
    const long long room = total - pos;
    const int take = room < (long long)N ? (int)room : N;
    for (int i = tid; i < take; i += 256) u[pos + i] = (float)(temper(cur[i]) & 0xffffffu) * (1.0f / 16777216.0f);
    if (j == nb) {                                               // the stream's last block: hand the engine on
      __syncthreads();
      for (int i = tid; i < N; i += 256) engine[i] = cur[i];
      if (tid == 0) { engine[N] = (unsigned)(N + 1 - take); engine[N + 1] = (unsigned)take; }
    }
  }
}

// one thread per (group of 16, j < 8)
__global__ __launch_bounds__(256) void mt_box_muller_kernel(const float* __restrict__ u, float* __restrict__ out, const Seg* __restrict__ segs,
                                                            int nseg, long long total_groups) {
  const long long pair = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long g = pair >> 3;
  const int j = (int)(pair & 7);
  if (g >= total_groups) return;
  int s = 0;
  while (s + 1 < nseg && segs[s + 1].first_group <= g) ++s;
  const Seg sg = segs[s];
  const long long lg = g - sg.first_group, nbody = sg.size >> 4;
  const bool tail = lg >= nbody;                       // the 16 extra outputs behind a size that is not a multiple of 16
  const long long src = tail ? sg.src + sg.size : sg.src + 16 * lg;
  const long long rel = tail ? sg.size - 16 : 16 * lg;       // first element of the group inside the tensor
  const long long own = (sg.size & 15) ? sg.size - 16 : sg.size;   // body groups leave [own, size) to the tail group
  const float u1 = 1.0f - u[src + j], u2 = u[src + j + 8];
  const float r = sqrtf(-2.0f * logf(u1)), th = 6.28318530717958647692f * u2;
  float sn, cs;
  sincosf(th, &sn, &cs);
  if (tail || rel + j < own) out[sg.dst + rel + j] = r * cs;
  if (tail || rel + j + 8 < own) out[sg.dst + rel + j + 8] = r * sn;
}
#endif

// ---- host side: the engine moved forward by n calls, no output --------------------------------------------------------
// Lets K host threads draw K contiguous pieces of ONE torch CPU stream at the same time (networks/bbb/eps.py: a torch.Generator per
// piece, set to the state the sequential draw would have reached there).  The block regeneration is ATen's next_state() written as
// three dependence-free runs (words 0-226 read old words only, 227-453 read the first run's results, 454-622 the second's) so that
// the compiler vectorises them: ~0.2 ms per million outputs on one core.
inline unsigned host_twist(unsigned u, unsigned v) {
  return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);
}
inline void host_next_state(unsigned* p) {
  for (int i = 0; i < N - M; ++i) p[i] = p[i + M] ^ host_twist(p[i], p[i + 1]);                                  // 0 .. 226
  for (int i = N - M; i < 2 * (N - M); ++i) p[i] = p[i + M - N] ^ host_twist(p[i], p[i + 1]);                    // 227 .. 453
  for (int i = 2 * (N - M); i < N - 1; ++i) p[i] = p[i + M - N] ^ host_twist(p[i], p[i + 1]);                    // 454 .. 622
  p[N - 1] = p[M - 1] ^ host_twist(p[N - 1], p[0]);
}
// engine = state[624], left, next as ATen's MT19937RNGEngine holds them; one call there is `if (--left == 0) next_state(); y = state[next++]`
inline void host_advance(unsigned* engine, unsigned long long n) {
  long long left = (int)engine[N], next = (int)engine[N + 1];
  while (n > 0) {
    if (left > 1) {                                   // calls that stay inside the current block
      const unsigned long long k = n < (unsigned long long)(left - 1) ? n : (unsigned long long)(left - 1);
      left -= (long long)k; next += (long long)k; n -= k;
      continue;
    }
    host_next_state(engine);                          // this call regenerates, then reads word 0
    left = N; next = 1; --n;
  }
  engine[N] = (unsigned)left; engine[N + 1] = (unsigned)next;
}

}  // namespace mt
}  // namespace mlhot
