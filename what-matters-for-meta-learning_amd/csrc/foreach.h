// Element-parallel and single-block-reduction launch helpers.
//
// Memory-bound glue (pool, aggregators, FAVOR+ feature maps, loss) is written as index
// functors `f(i)`; on the GPU they run under a grid-stride kernel sized for 256 CUs, in the
// hostsim flavour under a plain loop.  Reductions to a scalar run in ONE 1024-thread workgroup
// with a fixed-order LDS tree, so results are bitwise reproducible run to run.
#pragma once
#include "common.h"

namespace mlhot {

#ifndef MLHOT_HOSTSIM
template <class F>
__global__ __launch_bounds__(256) void foreach_kernel(const F f, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) f(i);
}

// 1024 threads, 4 independent strided accumulators per thread (4 loads in flight), then a fixed-order LDS tree
template <class R>
__global__ __launch_bounds__(1024) void reduce1_kernel(const R r, int n) {
  __shared__ typename R::T sm[1024];
  typename R::T a0 = r.identity(), a1 = r.identity(), a2 = r.identity(), a3 = r.identity();
  int i = threadIdx.x;
  for (; i + 3072 < n; i += 4096) {
    a0 = r.combine(a0, r.load(i)); a1 = r.combine(a1, r.load(i + 1024));
    a2 = r.combine(a2, r.load(i + 2048)); a3 = r.combine(a3, r.load(i + 3072));
  }
  for (; i < n; i += 1024) a0 = r.combine(a0, r.load(i));
  sm[threadIdx.x] = r.combine(r.combine(a0, a1), r.combine(a2, a3));
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = r.combine(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) r.finish(sm[0]);
}
#endif

// One 256-thread workgroup per "segment" g: acc = combine over i < n of r.load(g, i) in a fixed-order
// LDS tree, then r.finish(g, acc).  Used for per-channel batch-norm statistics (a segment = a channel).
#ifndef MLHOT_HOSTSIM
template <class R>
__global__ __launch_bounds__(256) void reduce_seg_kernel(const R r, int n) {
  __shared__ typename R::T sm[256];
  const int g = blockIdx.x;
  typename R::T acc = r.identity();
  for (int i = threadIdx.x; i < n; i += 256) acc = r.combine(acc, r.load(g, i));
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = r.combine(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) r.finish(g, sm[0]);
}
#endif
template <class R>
int run_reduce_seg(const R& r, int nseg, int n, hipStream_t stream, const char* what) {
#ifdef MLHOT_HOSTSIM
  (void)stream; (void)what;
  for (int g = 0; g < nseg; ++g) {
    typename R::T acc = r.identity();
    for (int i = 0; i < n; ++i) acc = r.combine(acc, r.load(g, i));
    r.finish(g, acc);
  }
  return MLHOT_OK;
#else
  if (nseg <= 0) return MLHOT_OK;
  ProfScope ps(what, stream);
  hipLaunchKernelGGL((reduce_seg_kernel<R>), dim3(nseg), dim3(256), 0, stream, r, n);
  return check_launch(what);
#endif
}

template <class F>
int run_foreach(const F& f, size_t n, hipStream_t stream, const char* what) {
  if (n == 0) return MLHOT_OK;
#ifdef MLHOT_HOSTSIM
  (void)stream; (void)what;
  for (size_t i = 0; i < n; ++i) f(i);
  return MLHOT_OK;
#else
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  ProfScope ps(what, stream);
  hipLaunchKernelGGL((foreach_kernel<F>), dim3((unsigned)blocks), dim3(256), 0, stream, f, n);
  return check_launch(what);
#endif
}

template <class R>
int run_reduce1(const R& r, int n, hipStream_t stream, const char* what) {
#ifdef MLHOT_HOSTSIM
  (void)stream; (void)what;
  typename R::T acc = r.identity();
  for (int i = 0; i < n; ++i) acc = r.combine(acc, r.load(i));
  r.finish(acc);
  return MLHOT_OK;
#else
  ProfScope ps(what, stream);
  hipLaunchKernelGGL((reduce1_kernel<R>), dim3(1), dim3(1024), 0, stream, r, n);
  return check_launch(what);
#endif
}

}  // namespace mlhot
