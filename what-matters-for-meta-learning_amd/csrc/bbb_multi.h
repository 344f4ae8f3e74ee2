// Bayes-by-backprop weight samples of MANY tensors in one launch (the 13 conv layers x (weight, bias) of the ShapeNet3D encoder
// are 26 tensors per pass; one launch pair instead of 52 sample + 52 KL-sum launches, one instead of 52 for the backward).
// Same element arithmetic as BbbSample / BbbSampleBwd (ops_direct.h; bbb/BBBConv.py:86-108).  A workgroup owns 1024
// consecutive elements of ONE tensor; the KL terms are summed per workgroup in a fixed-order LDS tree and the per-workgroup
// partials by a second single-workgroup launch, so the KL is reproducible run to run.
#pragma once
#include "common.h"
#include "foreach.h"
#include "ops_direct.h"
#include "favor.h"      // SumRed, MLHOT_TRY
#include "../../include/mlhot.h"

namespace mlhot {

constexpr int BBB_MAX = MLHOT_BBB_MAX_ITEMS;
constexpr int BBB_CHUNK = 1024;

struct FillZero { float* d; MLHOT_HD void operator()(size_t i) const { d[i] = 0.f; } };

struct BbbMulti {
  mlhot_bbb_item it[BBB_MAX];
  int first[BBB_MAX + 1];       // first workgroup of item i
  int n;
};

inline int bbb_multi_plan(const mlhot_bbb_item* items, int n_items, BbbMulti& m) {
  if (n_items < 0 || n_items > BBB_MAX) { set_error("bbb_sample_multi: at most %d tensors per call", BBB_MAX); return MLHOT_ERR_ARG; }
  m.n = n_items; m.first[0] = 0;
  for (int i = 0; i < n_items; ++i) {
    if (!items[i].mu || !items[i].rho || !items[i].eps) { set_error("bbb_sample_multi: null tensor"); return MLHOT_ERR_ARG; }
    m.it[i] = items[i];
    m.first[i + 1] = m.first[i] + (int)((items[i].n + BBB_CHUNK - 1) / BBB_CHUNK);
  }
  return MLHOT_OK;
}

#ifndef MLHOT_HOSTSIM
__global__ __launch_bounds__(256) void bbb_sample_multi_kernel(const BbbMulti m, float* __restrict__ partial) {
  __shared__ float sm[256];
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.first[i + 1]) ++i;
  const mlhot_bbb_item it = m.it[i];
  const size_t base = (size_t)((int)blockIdx.x - m.first[i]) * BBB_CHUNK;
  float acc = 0.f;
  // every operand of the workgroup's chunk requested before the first store: the item's pointers may alias as far as the compiler
  // knows, so with the stores in between the four trips of this loop were four serial round trips (14 us for 0.5 M elements)
  constexpr int NU = BBB_CHUNK / 256;
  float rho[NU], mu[NU], e1[NU], e2[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t e = base + u * 256 + threadIdx.x;
    const bool in = e < it.n;
    rho[u] = in ? it.rho[e] : 0.f; mu[u] = in ? it.mu[e] : 0.f; e1[u] = in ? it.eps[e] : 0.f;
    e2[u] = (in && it.eps2 != nullptr) ? it.eps2[e] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t e = base + u * 256 + threadIdx.x;
    if (e < it.n) {
      const float sg = log1pf(expf(rho[u]));
      it.w[e] = mu[u] + e1[u] * sg;
      if (it.eps2 != nullptr) it.w2[e] = mu[u] + e2[u] * sg;      // second, independent sample of the same posterior
      const float q = 0.1f / sg, z = mu[u] / sg;
      acc += 0.5f * (2.f * logf(sg / 0.1f) - 1.f + q * q + z * z);
    }
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void bbb_sample_multi_bwd_kernel(const BbbMulti m, const float* __restrict__ dkl) {
  int i = 0;
  while (i + 1 < m.n && (int)blockIdx.x >= m.first[i + 1]) ++i;
  const mlhot_bbb_item it = m.it[i];
  const size_t base = (size_t)((int)blockIdx.x - m.first[i]) * BBB_CHUNK;
  const float g = dkl[0];
  constexpr int NU = BBB_CHUNK / 256;           // all loads first (see the forward)
  float rho[NU], muv[NU], e1[NU], e2[NU], d1[NU], d2[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t e = base + u * 256 + threadIdx.x;
    const bool in = e < it.n;
    rho[u] = in ? it.rho[e] : 0.f; muv[u] = in ? it.mu[e] : 0.f; e1[u] = in ? it.eps[e] : 0.f;
    e2[u] = (in && it.eps2 != nullptr) ? it.eps2[e] : 0.f;
    d1[u] = (in && it.dw != nullptr) ? it.dw[e] : 0.f;              // a sample whose weight got no gradient (KL-only backward)
    d2[u] = (in && it.eps2 != nullptr && it.dw2 != nullptr) ? it.dw2[e] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const size_t e = base + u * 256 + threadIdx.x;
    if (e < it.n) {
      const float sg = log1pf(expf(rho[u])), inv = 1.f / sg, mu = muv[u];
      const float dwe = d1[u], dw2 = d2[u];
      it.dmu[e] = dwe + dw2 + g * mu * inv * inv;
      float dsig = dwe * e1[u] + g * (inv - 0.01f * inv * inv * inv - mu * mu * inv * inv * inv);
      if (it.eps2 != nullptr) dsig += dw2 * e2[u];
      it.drho[e] = dsig / (1.f + expf(-rho[u]));
    }
  }
}
#endif

inline int bbb_sample_multi_fwd(const mlhot_bbb_item* items, int n_items, float* partial, float* kl, hipStream_t s) {
  BbbMulti m;
  MLHOT_TRY(bbb_multi_plan(items, n_items, m));
  for (int i = 0; i < n_items; ++i) if (!items[i].w || (items[i].eps2 && !items[i].w2)) { set_error("bbb_sample_multi_fwd: null output"); return MLHOT_ERR_ARG; }
  const int blocks = m.first[m.n];
#ifdef MLHOT_HOSTSIM
  (void)s; (void)partial;
  float tot = 0.f;
  for (int i = 0; i < n_items; ++i) {
    const mlhot_bbb_item& it = items[i];
    for (size_t e = 0; e < it.n; ++e) {
      const float sg = log1pf(expf(it.rho[e]));
      it.w[e] = it.mu[e] + it.eps[e] * sg;
      if (it.eps2) it.w2[e] = it.mu[e] + it.eps2[e] * sg;
      const float q = 0.1f / sg, z = it.mu[e] / sg;
      tot += 0.5f * (2.f * logf(sg / 0.1f) - 1.f + q * q + z * z);
    }
  }
  kl[0] = tot;
  return MLHOT_OK;
#else
  if (blocks == 0) return run_foreach(FillZero{kl}, 1, s, "bbb.multi.zero");
  {
    ProfScope ps("bbb.multi.sample", s);
    hipLaunchKernelGGL(bbb_sample_multi_kernel, dim3(blocks), dim3(256), 0, s, m, partial);
  }
  MLHOT_TRY(check_launch("bbb.multi.sample"));
  return run_reduce1(SumRed{partial, kl}, blocks, s, "bbb.multi.kl");
#endif
}

inline int bbb_sample_multi_bwd(const mlhot_bbb_item* items, int n_items, const float* dkl, hipStream_t s) {
  BbbMulti m;
  MLHOT_TRY(bbb_multi_plan(items, n_items, m));
  for (int i = 0; i < n_items; ++i) if (!items[i].dmu || !items[i].drho) { set_error("bbb_sample_multi_bwd: null output"); return MLHOT_ERR_ARG; }
#ifdef MLHOT_HOSTSIM
  (void)s;
  for (int i = 0; i < n_items; ++i) {
    const mlhot_bbb_item& it = items[i];
    for (size_t e = 0; e < it.n; ++e) {
      const float sg = log1pf(expf(it.rho[e])), inv = 1.f / sg, mu = it.mu[e], g = dkl[0];
      const float dwe = it.dw ? it.dw[e] : 0.f, dw2 = (it.eps2 && it.dw2) ? it.dw2[e] : 0.f;
      it.dmu[e] = dwe + dw2 + g * mu * inv * inv;
      it.drho[e] = (dwe * it.eps[e] + (it.eps2 ? dw2 * it.eps2[e] : 0.f) + g * (inv - 0.01f * inv * inv * inv - mu * mu * inv * inv * inv)) / (1.f + expf(-it.rho[e]));
    }
  }
  return MLHOT_OK;
#else
  if (m.first[m.n] == 0) return MLHOT_OK;
  {
    ProfScope ps("bbb.multi.sample.bwd", s);
    hipLaunchKernelGGL(bbb_sample_multi_bwd_kernel, dim3(m.first[m.n]), dim3(256), 0, s, m, dkl);
  }
  return check_launch("bbb.multi.sample.bwd");
#endif
}

}  // namespace mlhot
