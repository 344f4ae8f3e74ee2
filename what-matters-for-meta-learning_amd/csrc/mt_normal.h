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

}  // namespace mt
}  // namespace mlhot
