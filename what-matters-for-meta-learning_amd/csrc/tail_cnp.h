// Fused tail of the vanilla CNP models (CNPShapeNet1D / CNPVanillaPascal1D, mean or max
// aggregation): the whole task-side computation has no cross-task dependency, so it is ONE
// forward and ONE backward launch with one 512-thread workgroup per task (plus the slab reduce),
// built from the same LDS-resident layer functions as the ANP tail (tail_fused.h).
//   forward : transform_y, EncoderFC, mean/max over the shot axis, r_to_z, broadcast, decoder0
//   backward: the exact reverse, per-task weight-gradient slab
// GPU build only; Nc in 1..16, Nq <= 16, two hidden layers, dim_r <= 128.  Other shapes (and
// "baco") run the generic chain in np_vanilla.h.
#pragma once
#include "tail_fused.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace tf {

struct CnpDims { int T, Nc, Nq, label_dim, y_dim, dw, dr, dz, h0, h1, dec_h, out_act, agg; };
struct CnpParams { const float *ty_w, *ty_b, *er_w[3], *er_b[3], *r2z_w, *r2z_b, *dec_w[3], *dec_b[3]; };
struct CnpSlab { int ty_w, ty_b, er_w[3], er_b[3], r2z_w, r2z_b, dec_w[3], dec_b[3], total; };

__host__ inline CnpSlab cnp_slab_layout(const CnpDims& d) {
  CnpSlab s; int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) / 4 * 4; return r; };
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  s.ty_w = take(d.dw / 4 * d.label_dim); s.ty_b = take(d.dw / 4);
  s.er_w[0] = take(d.h0 * ldc); s.er_b[0] = take(d.h0);
  s.er_w[1] = take(d.h1 * d.h0); s.er_b[1] = take(d.h1);
  s.er_w[2] = take(d.dr * d.h1); s.er_b[2] = take(d.dr);
  s.r2z_w = take(d.dz * d.dr); s.r2z_b = take(d.dz);
  s.dec_w[0] = take(d.dec_h * ldd); s.dec_b[0] = take(d.dec_h);
  s.dec_w[1] = take(d.dec_h * d.dec_h); s.dec_b[1] = take(d.dec_h);
  s.dec_w[2] = take(d.y_dim * d.dec_h); s.dec_b[2] = take(d.y_dim);
  s.total = o;
  return s;
}

struct CnpFwdArgs {
  CnpDims d; CnpParams p;
  const float* ctx_y;
  float *cat_in, *h0, *h1, *rs, *r, *zt, *dec_in, *d1, *d2, *mu; int32_t* amax;
};

__global__ __launch_bounds__(512) void cnp_fwd_kernel(const CnpFwdArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const CnpDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lr = ldpad(d.dr), Lz = ldpad(d.dz), Ld = ldpad(ldd),
            Lh = ldpad(d.dec_h), Ly = ldpad(d.label_dim);
  lptr s_cat = L0;
  lptr s_h0 = s_cat + 16 * Lcat;
  lptr s_h1 = s_h0 + 16 * Lh0;
  lptr s_rs = s_h1 + 16 * Lh1;
  lptr s_r = s_rs + 16 * Lr;        // row 0 = aggregated r, rows 1.. = 0
  lptr s_zt = s_r + 16 * Lr;        // row 0 = r_to_z(r)
  lptr s_dec = s_zt + 16 * Lz;
  lptr s_d1 = s_dec + 16 * Ld;
  lptr s_d2 = s_d1 + 16 * Lh;
  lptr s_y = s_d2 + 16 * Lh;
  lptr s_red = s_y + 16 * Ly;
  using PRM = CnpParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  lds_zero(L0, 16 * (Lcat + Lh0 + Lh1 + 2 * Lr + Lz + Ld + 2 * Lh + Ly), tid, 512);
  __syncthreads();
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  gptr g_cat = G(a.cat_in) + rc * ldc;
  gptr g_dec = G(a.dec_in) + rq * ldd;
  lds_load(s_cat, Lcat, g_cat, ldc, d.Nc, d.dw, tid, 512);
  lds_load(s_dec, Ld, g_dec, ldd, d.Nq, d.dw, tid, 512);
  lds_load(s_y, Ly, a.ctx_y + rc * d.label_dim, d.label_dim, d.Nc, d.label_dim, tid, 512);
  __syncthreads();
  wg_linear<8>(s_y, Ly, d.label_dim, WB1(ty_w, ty_b, d.dw / 4), d.dw / 4, ACT_NONE, s_cat + d.dw, Lcat, g_cat + d.dw, ldc, d.Nc, nullptr, wave, lane);
  __syncthreads();
  wg_linear<8>(s_cat, Lcat, ldc, WB1(er_w[0], er_b[0], d.h0), d.h0, ACT_RELU, s_h0, Lh0, G(a.h0 + rc * d.h0), d.h0, d.Nc, nullptr, wave, lane);
  __syncthreads();
  wg_linear<8>(s_h0, Lh0, d.h0, WB1(er_w[1], er_b[1], d.h1), d.h1, ACT_RELU, s_h1, Lh1, G(a.h1 + rc * d.h1), d.h1, d.Nc, nullptr, wave, lane);
  __syncthreads();
  wg_linear<8>(s_h1, Lh1, d.h1, WB1(er_w[2], er_b[2], d.dr), d.dr, ACT_NONE, s_rs, Lr, G(a.rs + rc * d.dr), d.dr, d.Nc, nullptr, wave, lane);
  __syncthreads();
  // aggregate over the shot axis: one lane per feature, the shots of a task are the tile's rows
  for (int j = tid; j < d.dr; j += 512) {
    float v; int arg = 0;
    if (d.agg == 0) {
      float sacc = 0.f;
      for (int n = 0; n < d.Nc; ++n) sacc += s_rs[n * Lr + j];
      v = sacc / (float)d.Nc;
    } else {
      v = s_rs[j];
      for (int n = 1; n < d.Nc; ++n) { const float c = s_rs[n * Lr + j]; if (c > v) { v = c; arg = n; } }
    }
    s_r[j] = v;
    a.r[(size_t)t * d.dr + j] = v;
    a.amax[(size_t)t * d.dr + j] = arg;
  }
  __syncthreads();
  wg_linear<8>(s_r, Lr, d.dr, WB1(r2z_w, r2z_b, d.dz), d.dz, ACT_NONE, s_zt, Lz, G(a.zt + (size_t)t * d.dz), d.dz, 1, s_red, wave, lane);
  __syncthreads();
  for (int i = tid; i < d.Nq * d.dz; i += 512) {          // z broadcast over the target rows
    const int r = i / d.dz, j = i % d.dz;
    s_dec[r * Ld + d.dw + j] = s_zt[j];
    g_dec[(size_t)r * ldd + d.dw + j] = s_zt[j];
  }
  __syncthreads();
  wg_linear<8>(s_dec, Ld, ldd, WB1(dec_w[0], dec_b[0], d.dec_h), d.dec_h, ACT_RELU, s_d1, Lh, G(a.d1 + rq * d.dec_h), d.dec_h, d.Nq, nullptr, wave, lane);
  __syncthreads();
  wg_linear<8>(s_d1, Lh, d.dec_h, WB1(dec_w[1], dec_b[1], d.dec_h), d.dec_h, ACT_RELU, s_d2, Lh, G(a.d2 + rq * d.dec_h), d.dec_h, d.Nq, nullptr, wave, lane);
  __syncthreads();
  wg_linear<8>(s_d2, Lh, d.dec_h, WB1(dec_w[2], dec_b[2], d.y_dim), d.y_dim, d.out_act, nullptr, 0, G(a.mu + rq * d.y_dim), d.y_dim, d.Nq, s_red, wave, lane);
}
__host__ inline size_t cnp_fwd_lds_bytes(const CnpDims& d) {
  return sizeof(float) * (16 * (ldpad(d.dw + d.dw / 4) + ldpad(d.h0) + ldpad(d.h1) + 2 * ldpad(d.dr) + ldpad(d.dz) + ldpad(d.dw + d.dz) +
                                2 * ldpad(d.dec_h) + ldpad(d.label_dim)) + 8 * 256 + ptab_floats<CnpParams>());
}

struct CnpBwdArgs {
  CnpDims d; CnpParams p; CnpSlab sl;
  const float *ctx_y, *dmu, *mu, *d2, *d1, *dec_in, *r, *rs, *h1, *h0, *cat_in; const int32_t* amax;
  float *d_dec_in, *d_cat_in, *slab;
};

__global__ __launch_bounds__(512) void cnp_bwd_kernel(const CnpBwdArgs a) {
  extern __shared__ float lds[];
  lptr L0 = (lptr)lds;
  const CnpDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldc = d.dw + d.dw / 4, ldd = d.dw + d.dz;
  const int Lcat = ldpad(ldc), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lr = ldpad(d.dr), Lz = ldpad(d.dz), Ld = ldpad(ldd),
            Lh = ldpad(d.dec_h), Ly = ldpad(d.y_dim), Lyl = ldpad(d.label_dim);
  lptr p = L0;
  auto take = [&](int n) { lptr r = p; p += n; return r; };
  lptr s_g = take(16 * Ly);      lptr s_d2 = take(16 * Lh);    lptr s_d1 = take(16 * Lh);   lptr s_dec = take(16 * Ld);
  lptr s_dd2 = take(16 * Lh);    lptr s_dd1 = take(16 * Lh);   lptr s_ddec = take(16 * Ld);
  lptr s_r = take(16 * Lr);      lptr s_dzt = take(16 * Lz);   lptr s_dr = take(16 * Lr);
  lptr s_rs = take(16 * Lr);     lptr s_h1 = take(16 * Lh1);   lptr s_h0 = take(16 * Lh0);  lptr s_cat = take(16 * Lcat);
  lptr s_drs = take(16 * Lr);    lptr s_dh1 = take(16 * Lh1);  lptr s_dh0 = take(16 * Lh0); lptr s_dcat = take(16 * Lcat);
  lptr s_yl = take(16 * Lyl);    lptr s_red = take(8 * 256);
  lds_zero(L0, (int)(s_red - L0), tid, 512);
  using PRM = CnpParams;
  lu64 ptab = reinterpret_cast<lu64>(s_red + 8 * 256);
  ptab_fill(ptab, a.p, tid);
  __syncthreads();
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  for (int i = tid; i < d.Nq * d.y_dim; i += 512) {
    const int r = i / d.y_dim, c = i % d.y_dim;
    s_g[r * Ly + c] = a.dmu[(rq + r) * d.y_dim + c] * act_grad_from_out(d.out_act, a.mu[(rq + r) * d.y_dim + c]);
  }
  lds_load(s_d2, Lh, a.d2 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h, tid, 512);
  lds_load(s_d1, Lh, a.d1 + rq * d.dec_h, d.dec_h, d.Nq, d.dec_h, tid, 512);
  lds_load(s_dec, Ld, a.dec_in + rq * ldd, ldd, d.Nq, ldd, tid, 512);
  lds_load(s_r, Lr, a.r + (size_t)t * d.dr, d.dr, 1, d.dr, tid, 512);
  lds_load(s_rs, Lr, a.rs + rc * d.dr, d.dr, d.Nc, d.dr, tid, 512);
  lds_load(s_h1, Lh1, a.h1 + rc * d.h1, d.h1, d.Nc, d.h1, tid, 512);
  lds_load(s_h0, Lh0, a.h0 + rc * d.h0, d.h0, d.Nc, d.h0, tid, 512);
  lds_load(s_cat, Lcat, a.cat_in + rc * ldc, ldc, d.Nc, ldc, tid, 512);
  lds_load(s_yl, Lyl, a.ctx_y + rc * d.label_dim, d.label_dim, d.Nc, d.label_dim, tid, 512);
  __syncthreads();
  gptr sl = G(a.slab) + (size_t)t * a.sl.total;
  // decoder0
  wg_wgrad<8>(s_g, Ly, d.y_dim, s_d2, Lh, d.dec_h, sl + a.sl.dec_w[2], sl + a.sl.dec_b[2], wave, lane, tid);
  wg_dgrad<8>(s_g, Ly, d.y_dim, WB1N(dec_w[2], d.y_dim), d.dec_h, s_dd2, Lh, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  lds_actgrad(s_dd2, Lh, s_d2, Lh, d.dec_h, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dd2, Lh, d.dec_h, s_d1, Lh, d.dec_h, sl + a.sl.dec_w[1], sl + a.sl.dec_b[1], wave, lane, tid);
  wg_dgrad<8>(s_dd2, Lh, d.dec_h, WB1N(dec_w[1], d.dec_h), d.dec_h, s_dd1, Lh, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  lds_actgrad(s_dd1, Lh, s_d1, Lh, d.dec_h, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dd1, Lh, d.dec_h, s_dec, Ld, ldd, sl + a.sl.dec_w[0], sl + a.sl.dec_b[0], wave, lane, tid);
  wg_dgrad<8>(s_dd1, Lh, d.dec_h, WB1N(dec_w[0], d.dec_h), ldd, s_ddec, Ld, G(a.d_dec_in + rq * ldd), ldd, d.Nq, false, s_red, wave, lane);
  __syncthreads();
  // broadcast backward: dz_t = sum over the target rows
  for (int j = tid; j < d.dz; j += 512) {
    float sacc = 0.f;
    for (int r = 0; r < d.Nq; ++r) sacc += s_ddec[r * Ld + d.dw + j];
    s_dzt[j] = sacc;
  }
  __syncthreads();
  wg_wgrad<8>(s_dzt, Lz, d.dz, s_r, Lr, d.dr, sl + a.sl.r2z_w, sl + a.sl.r2z_b, wave, lane, tid);
  wg_dgrad<8>(s_dzt, Lz, d.dz, WB1N(r2z_w, d.dz), d.dr, s_dr, Lr, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  // aggregator backward: mean spreads dr / Nc over the shots, max routes it to the arg-max shot
  for (int i = tid; i < d.Nc * d.dr; i += 512) {
    const int n = i / d.dr, j = i % d.dr;
    const float g = s_dr[j];
    s_drs[n * Lr + j] = d.agg == 0 ? g / (float)d.Nc : (a.amax[(size_t)t * d.dr + j] == n ? g : 0.f);
  }
  __syncthreads();
  // EncoderFC, last layer first
  wg_wgrad<8>(s_drs, Lr, d.dr, s_h1, Lh1, d.h1, sl + a.sl.er_w[2], sl + a.sl.er_b[2], wave, lane, tid);
  wg_dgrad<8>(s_drs, Lr, d.dr, WB1N(er_w[2], d.dr), d.h1, s_dh1, Lh1, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  lds_actgrad(s_dh1, Lh1, s_h1, Lh1, d.h1, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh1, Lh1, d.h1, s_h0, Lh0, d.h0, sl + a.sl.er_w[1], sl + a.sl.er_b[1], wave, lane, tid);
  wg_dgrad<8>(s_dh1, Lh1, d.h1, WB1N(er_w[1], d.h1), d.h0, s_dh0, Lh0, nullptr, 0, 0, false, s_red, wave, lane);
  __syncthreads();
  lds_actgrad(s_dh0, Lh0, s_h0, Lh0, d.h0, ACT_RELU, tid, 512);
  __syncthreads();
  wg_wgrad<8>(s_dh0, Lh0, d.h0, s_cat, Lcat, ldc, sl + a.sl.er_w[0], sl + a.sl.er_b[0], wave, lane, tid);
  wg_dgrad<8>(s_dh0, Lh0, d.h0, WB1N(er_w[0], d.h0), ldc, s_dcat, Lcat, G(a.d_cat_in + rc * ldc), ldc, d.Nc, false, s_red, wave, lane);
  __syncthreads();
  wg_wgrad<8>(s_dcat + d.dw, Lcat, d.dw / 4, s_yl, Lyl, d.label_dim, sl + a.sl.ty_w, sl + a.sl.ty_b, wave, lane, tid);
}
__host__ inline size_t cnp_bwd_lds_bytes(const CnpDims& d) {
  const int Lcat = ldpad(d.dw + d.dw / 4), Lh0 = ldpad(d.h0), Lh1 = ldpad(d.h1), Lr = ldpad(d.dr), Lz = ldpad(d.dz), Ld = ldpad(d.dw + d.dz),
            Lh = ldpad(d.dec_h), Ly = ldpad(d.y_dim), Lyl = ldpad(d.label_dim);
  return sizeof(float) * (16 * (Ly + 4 * Lh + 2 * Ld + 4 * Lr + Lz + 2 * Lh1 + 2 * Lh0 + 2 * Lcat + Lyl) + 8 * 256 + ptab_floats<CnpParams>());
}

}  // namespace tf
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
