// Strict sharded parity of the FAVOR+ key stabiliser (networks/fast_attention.py:96-97: the key features are
// exp(dd - diag - torch.max(dd)) + eps with the maximum taken over EVERY key of the batch).  When the meta-batch is sharded
// over ranks each rank only sees its own tasks' keys.  The staged entry points of include/mlhot.h stop between the launch
// that produces the rank-local maximum and the launch that consumes it, and again in the backward between the launch that
// produces the rank-local sum of the stabiliser's gradient and the launch that routes it to the arg-max key.  The caller
// runs its collectives on the exchange block in between:
//   x[X_MAX]    out of forward stage 0: the rank's key maximum.  In to stage 1: the maximum over all ranks
//   x[X_OWNER]  in to forward stage 1: non-zero on the ONE rank that holds the arg-max (lowest rank on ties)
//   x[X_GSUM]   out of backward stage 0: the rank's sum of dL/d(stabiliser).  In to stage 1: the sum over all ranks
// Nothing here talks to a communication library: the kernels only publish a scalar and patch the partial results the
// consuming launch folds (a non-owner's candidates lose their position, so no local key receives the arg-max gradient).
#pragma once
#include "common.h"

namespace mlhot {

struct Stage {        // -1: the whole pass in one call (no exchange); 0 / 1: the halves around the caller's collective
  int stage = -1;
  float* x = nullptr;
  bool first() const { return stage != 1; }
  bool second() const { return stage != 0; }
  bool staged() const { return stage >= 0; }
};
inline int stage_check(const Stage& st, const char* what) {
  if (st.stage < -1 || st.stage > 1 || (st.staged() && !st.x)) { set_error("%s: stage must be 0 or 1 with an exchange block", what); return MLHOT_ERR_ARG; }
  return MLHOT_OK;
}

#ifndef MLHOT_HOSTSIM
namespace sx {

constexpr int NONE = 0x7fffffff;      // a position no key has (the consumers compare positions for equality / unpack a task from it)
enum { X_MAX = 0, X_OWNER = 1, X_GSUM = 2 };

__device__ __forceinline__ float block_sum_256(float s, float* sm, int tid) {      // fixed order: the apply kernel repeats it bit for bit
  sm[tid] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (tid < k) sm[tid] += sm[tid + k]; __syncthreads(); }
  return sm[0];
}

__global__ __launch_bounds__(256) void max_publish_kernel(const float* __restrict__ v, int n, float* __restrict__ x) {
  __shared__ float sm[256];
  const int tid = threadIdx.x;
  float best = -INFINITY;
  for (int i = tid; i < n; i += 256) best = fmaxf(best, v[i]);
  sm[tid] = best;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if (tid < k) sm[tid] = fmaxf(sm[tid], sm[tid + k]); __syncthreads(); }
  if (tid == 0) x[X_MAX] = sm[0];
}
// v / code: the candidates the consuming launch folds (value, position).  On a rank that does not hold the arg-max every
// candidate that ties with the batch maximum loses its position and candidate 0 carries the batch maximum.
__global__ __launch_bounds__(256) void max_apply_kernel(float* __restrict__ v, int* __restrict__ code, int n, const float* __restrict__ x) {
  const float g = x[X_MAX];
  if (x[X_OWNER] != 0.f) return;
  for (int i = threadIdx.x; i < n; i += 256)
    if (v[i] >= g) code[i] = NONE;
  __syncthreads();
  if (threadIdx.x == 0) { v[0] = g; code[0] = NONE; }
}
__global__ __launch_bounds__(256) void sum_publish_kernel(const float* __restrict__ v, int n, float* __restrict__ x) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum_256(s, sm, threadIdx.x);
  if (threadIdx.x == 0) x[X_GSUM] = s;
}
// dst[0] (+)= batch sum - rank sum: what the consuming launch has to add to its own fold of v
// (dst may be an element of v: every read of v precedes the fold's barriers)
__global__ __launch_bounds__(256) void sum_apply_kernel(const float* v, int n, const float* x, float* dst, int add) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = block_sum_256(s, sm, threadIdx.x);
  if (threadIdx.x == 0) dst[0] = (add ? dst[0] : 0.f) + (x[X_GSUM] - s);
}

inline int max_publish(const float* v, int n, float* x, hipStream_t s) {
  hipLaunchKernelGGL(max_publish_kernel, dim3(1), dim3(256), 0, s, v, n, x);
  return hipGetLastError() == hipSuccess ? MLHOT_OK : (set_error("stab.max_publish: launch failed"), MLHOT_ERR_LAUNCH);
}
inline int max_apply(float* v, int* code, int n, const float* x, hipStream_t s) {
  hipLaunchKernelGGL(max_apply_kernel, dim3(1), dim3(256), 0, s, v, code, n, x);
  return hipGetLastError() == hipSuccess ? MLHOT_OK : (set_error("stab.max_apply: launch failed"), MLHOT_ERR_LAUNCH);
}
inline int sum_publish(const float* v, int n, float* x, hipStream_t s) {
  hipLaunchKernelGGL(sum_publish_kernel, dim3(1), dim3(256), 0, s, v, n, x);
  return hipGetLastError() == hipSuccess ? MLHOT_OK : (set_error("stab.sum_publish: launch failed"), MLHOT_ERR_LAUNCH);
}
inline int sum_apply(const float* v, int n, const float* x, float* dst, int add, hipStream_t s) {
  hipLaunchKernelGGL(sum_apply_kernel, dim3(1), dim3(256), 0, s, v, n, x, dst, add);
  return hipGetLastError() == hipSuccess ? MLHOT_OK : (set_error("stab.sum_apply: launch failed"), MLHOT_ERR_LAUNCH);
}

}  // namespace sx
#endif  // !MLHOT_HOSTSIM
}  // namespace mlhot
