// The fused CNP tail (tail_cnp.h: CNPShapeNet1D / CNPVanillaPascal1D, mean or max aggregation) specialised for the dimensions every
// shipped vanilla-CNP config uses (cfg/train/CNP_*1D.yaml: dim_w = dim_z = 64, n_hidden_units_r = [100, 100], dim_r = 100, decoder
// hidden 100), label / output widths 1..4 at run time - built from tail_spec.h's blocks (weight fragments requested at kernel entry,
// compile-time shapes, ReLU's backward in the data-gradient epilogue).  Same argument structs, saved buffers and slab layout as
// tf::cnp_fwd_kernel / cnp_bwd_kernel, which stay the path of every other shape and the A/B reference (option tail_spec bits 1 / 8).
// Round 5: the run-time-shaped kernels were 34.9 + 53.0 us of BASELINE configs[1]'s 0.574 ms step on 16 workgroups.
//   forward : one workgroup per task: [fold of the encoder Linear's partial results,] transform_y, EncoderFC, mean / max over the shots,
//             r_to_z, broadcast, decoder0
//   backward: GR workgroups per task: all of them walk the data-gradient chain (decoder0^T, broadcast^T, r_to_z^T, the aggregator's
//             backward, EncoderFC^T), the weight-gradient tiles riding in its barrier intervals are dealt over their 8 GR waves;
//             workgroup 0 of a group writes d_dec_in, d_cat_in and the bias sums.  The loss's gradient may come as a descriptor.
#pragma once
#include "tail_spec.h"
#include "tail_cnp.h"

#ifndef MLHOT_HOSTSIM
namespace mlhot {
namespace ts {

constexpr int DR = 100;                        // dim_r of the CNP configs
constexpr int N_LR = lds_ld(DR);

inline bool cnp_applies(const CnpDims& d) {
  return d.dw == DW && d.dz == DZ && d.h0 == H0 && d.h1 == H1 && d.dec_h == DH && d.dr == DR && d.label_dim >= 1 && d.label_dim <= 4 &&
         d.y_dim >= 1 && d.y_dim <= 4 && d.Nc >= 1 && d.Nc <= 16 && d.Nq >= 1 && d.Nq <= 16 && (d.agg == 0 || d.agg == 1);
}

// the encoder Linear's split-K partial results, when the encoder left its fold to this kernel (as PhaseAArgs::xslab)
struct CnpXFold { const float* slab; const float* bias; int k, n; };

constexpr int CNF_FLOATS = 16 * (A_LCAT + 2 * A_LH + 2 * N_LR + C_LR + C_LD + 2 * C_LH + A_LY) + NWV * 256;
__host__ inline size_t cnp_fwd_lds_bytes() { return sizeof(float) * CNF_FLOATS; }

__global__ __launch_bounds__(512) void cnp_fwd_kernel(const CnpFwdArgs a, const CnpXFold xf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const CnpDims& d = a.d;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  lptr s_cat = L0;                      // [16][A_LCAT]  [x_ctx | transform_y(ctx_y)]
  lptr s_h0 = s_cat + 16 * A_LCAT;
  lptr s_h1 = s_h0 + 16 * A_LH;
  lptr s_rs = s_h1 + 16 * A_LH;          // [16][N_LR]
  lptr s_r = s_rs + 16 * N_LR;           // row 0 = the aggregated r, rows 1.. zero
  lptr s_zt = s_r + 16 * N_LR;           // [16][C_LR]: row 0 = r_to_z(r)
  lptr s_dec = s_zt + 16 * C_LR;         // [16][C_LD]  [x_qry | z]
  lptr s_d1 = s_dec + 16 * C_LD;
  lptr s_d2 = s_d1 + 16 * C_LH;
  lptr s_y = s_d2 + 16 * C_LH;           // [16][A_LY] labels
  lptr s_red = s_y + 16 * A_LY;          // [8 waves][256] K-split partials
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  // ---- every global read of the encoder half, up front
  Tile64 xt, xq;
  FoldTile<XK> fc, fq;
  const bool folding = xf.slab != nullptr;              // kernel-uniform
  if (folding) {
    fc.issue(xf.slab, xf.bias, xf.n, (int)rc, d.Nc, tid);
    fq.issue(xf.slab, xf.bias, xf.n, d.T * d.Nc + (int)rq, d.Nq, tid);
  } else {
    xt.fetch(a.cat_in + rc * LDC, LDC, d.Nc, tid);
    xq.fetch(a.dec_in + rq * LDD, LDD, d.Nq, tid);
  }
  float yv = 0.f;
  if (tid < 256 && (tid >> 4) < d.Nc && (tid & 15) < d.label_dim) yv = a.ctx_y[(rc + (tid >> 4)) * d.label_dim + (tid & 15)];
  Lin<4, DW / 4> l_ty;  Lin<LDC, H0> l_e0;  Lin<H0, H1> l_e1;  Lin<H1, DR> l_e2;
  if (wave == 0) {                       // transform_y: K = label_dim (1..4, run time) rides in a K = 4 layer through the scalar path
    const int lr = lane & 15, lq = lane >> 4;
    l_ty.bias = a.p.ty_b[lr];
    l_ty.b[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (lq == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) l_ty.b[0][e] = a.p.ty_w[lr * d.label_dim + (e < d.label_dim ? e : d.label_dim - 1)];
    }
  }
  l_e0.load(a.p.er_w[0], a.p.er_b[0], wave, lane);
  l_e1.load(a.p.er_w[1], a.p.er_b[1], wave, lane);
  l_e2.load(a.p.er_w[2], a.p.er_b[2], wave, lane);
  __builtin_amdgcn_sched_barrier(0);               // every request above is out before the first wait
  if (folding) { xt.v = fc.finish(); xq.v = fq.finish(); }
  xt.stash(s_cat, A_LCAT, tid);
  xq.stash(s_dec, C_LD, tid);
  if (tid < 256) s_y[(tid >> 4) * A_LY + (tid & 15)] = yv;
  lds_zero4(s_r, 16 * N_LR, tid);                  // rows 1.. of the one-row tiles stay zero
  lds_zero4(s_zt, 16 * C_LR, tid);
  float* g_cat = a.cat_in + rc * LDC;
  float* g_dec = a.dec_in + rq * LDD;
  if (folding && tid < 256) {                      // the folded features: the backward and the encoder's backward read them in cat_in / dec_in
    const int row = (tid >> 4) & 15, c4 = tid & 15;
    if (row < d.Nc) *reinterpret_cast<f32x4_t*>(g_cat + (size_t)row * LDC + 4 * c4) = xt.v;
    if (row < d.Nq) *reinterpret_cast<f32x4_t*>(g_dec + (size_t)row * LDD + 4 * c4) = xq.v;
  }
  __syncthreads();
  // the decoder half's fragments: requested now, needed four layers from here
  Lin<DR, DZ> l_z;  Lin<LDD, DH> l_d0;  Lin<DH, DH> l_d1;  Lin<DH, 4> l_d2;
  l_z.load(a.p.r2z_w, a.p.r2z_b, wave, lane);
  l_d0.load(a.p.dec_w[0], a.p.dec_b[0], wave, lane);
  l_d1.load(a.p.dec_w[1], a.p.dec_b[1], wave, lane);
  l_d2.load(a.p.dec_w[2], a.p.dec_b[2], wave, lane, DH, d.y_dim);
  // transform_y -> cat[:, dw:]
  if (wave == 0) {
    const int lr = lane & 15, lq = lane >> 4;
    const f32x4_t x = *reinterpret_cast<lc4ptr>(s_y + lr * A_LY + 4 * lq);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = mfma4(x[0], l_ty.b[0][0], acc); acc = mfma4(x[1], l_ty.b[0][1], acc); acc = mfma4(x[2], l_ty.b[0][2], acc); acc = mfma4(x[3], l_ty.b[0][3], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lq + r;
      const float v = acc[r] + l_ty.bias;
      s_cat[row * A_LCAT + DW + lr] = v;
      if (row < d.Nc) g_cat[row * LDC + DW + lr] = v;
    }
  }
  __syncthreads();
  l_e0.finish(l_e0.mma(s_cat, A_LCAT, wave, lane), ACT_RELU, s_red, s_h0, A_LH, a.h0 + rc * H0, H0, d.Nc, wave, lane);
  __syncthreads();
  l_e1.finish(l_e1.mma(s_h0, A_LH, wave, lane), ACT_RELU, s_red, s_h1, A_LH, a.h1 + rc * H1, H1, d.Nc, wave, lane);
  __syncthreads();
  l_e2.finish(l_e2.mma(s_h1, A_LH, wave, lane), ACT_NONE, s_red, s_rs, N_LR, a.rs + rc * DR, DR, d.Nc, wave, lane);
  __syncthreads();
  // aggregate over the shot axis: one thread per feature (CNPShapeNet1D.py:78-126: torch.mean / torch.max over dim 1)
  if (tid < DR) {
    float v; int arg = 0;
    if (d.agg == 0) {
      float sacc = 0.f;
      for (int n = 0; n < d.Nc; ++n) sacc += s_rs[n * N_LR + tid];
      v = sacc / (float)d.Nc;
    } else {
      v = s_rs[tid];
      for (int n = 1; n < d.Nc; ++n) { const float c = s_rs[n * N_LR + tid]; if (c > v) { v = c; arg = n; } }
    }
    s_r[tid] = v;
    a.r[(size_t)t * DR + tid] = v;
    a.amax[(size_t)t * DR + tid] = arg;
  }
  __syncthreads();
  l_z.finish(l_z.mma(s_r, N_LR, wave, lane), ACT_NONE, s_red, s_zt, C_LR, a.zt + (size_t)t * DZ, DZ, 1, wave, lane);
  __syncthreads();
  for (int i = tid; i < 16 * DZ; i += NWV * 64) {       // z broadcast over the target rows (zeros below them)
    const int r = i / DZ, j = i - r * DZ;
    const float v = r < d.Nq ? s_zt[j] : 0.f;
    s_dec[r * C_LD + DW + j] = v;
    if (r < d.Nq) g_dec[(size_t)r * LDD + DW + j] = v;
  }
  __syncthreads();
  l_d0.finish(l_d0.mma(s_dec, C_LD, wave, lane), ACT_RELU, s_red, s_d1, C_LH, a.d1 + rq * DH, DH, d.Nq, wave, lane);
  __syncthreads();
  l_d1.finish(l_d1.mma(s_d1, C_LH, wave, lane), ACT_RELU, s_red, s_d2, C_LH, a.d2 + rq * DH, DH, d.Nq, wave, lane);
  __syncthreads();
  l_d2.finish(l_d2.mma(s_d2, C_LH, wave, lane), d.out_act, s_red, nullptr, 0, a.mu + rq * d.y_dim, d.y_dim, d.Nq, wave, lane, d.y_dim);
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
constexpr int CNB_FLOATS = 16 * (2 * A_LY + 4 * C_LH + 2 * C_LD + 4 * N_LR + C_LR + 4 * A_LH + 2 * A_LCAT) + NWV * 256 + 128;
__host__ inline size_t cnp_bwd_lds_bytes() { return sizeof(float) * CNB_FLOATS; }

template <int GR>
__global__ __launch_bounds__(512) void cnp_bwd_kernel(const CnpBwdArgs a, const LossDesc loss) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  lptr L0 = (lptr)lds;
  const CnpDims& d = a.d;
  if (loss.value != nullptr && (int)blockIdx.x == GR * d.T) {              // one workgroup more than the tasks need: the loss VALUE (ops_direct.h)
    loss_value_block(LossRed{loss.kind, d.y_dim, loss.gt_dim, d.T * d.Nq, a.mu, loss.gt, loss.value}, d.T * d.Nq, lds);
    return;
  }
  const int t = (int)blockIdx.x / GR, grp = (int)blockIdx.x % GR;          // GR workgroups per task
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6), gwave = grp * NWV + wave;
  const bool first = grp == 0;
  lptr p = L0;
  auto take = [&](int n) { lptr r = p; p += n; return r; };
  lptr s_g = take(16 * A_LY);      lptr s_d2 = take(16 * C_LH);   lptr s_d1 = take(16 * C_LH);   lptr s_dec = take(16 * C_LD);
  lptr s_dd2 = take(16 * C_LH);    lptr s_dd1 = take(16 * C_LH);  lptr s_ddec = take(16 * C_LD);
  lptr s_r = take(16 * N_LR);      lptr s_dzt = take(16 * C_LR);  lptr s_dr = take(16 * N_LR);
  lptr s_h1 = take(16 * A_LH);     lptr s_h0 = take(16 * A_LH);   lptr s_cat = take(16 * A_LCAT);
  lptr s_drs = take(16 * N_LR);    lptr s_dh1 = take(16 * A_LH);  lptr s_dh0 = take(16 * A_LH);  lptr s_dcat = take(16 * A_LCAT);
  lptr s_y = take(16 * A_LY);      lptr s_red = take(NWV * 256);
  MLHOT_LDS int* s_am = reinterpret_cast<MLHOT_LDS int*>(take(128));
  (void)take(16 * N_LR);           // (one spare tile of the size estimate)
  const size_t rc = (size_t)t * d.Nc, rq = (size_t)t * d.Nq;
  // ---- every global read of the decoder half, up front
  float gv = 0.f;
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    if (r < d.Nq && c < d.y_dim) {
      float up = a.dmu != nullptr ? a.dmu[(rq + r) * d.y_dim + c] : 0.f;
      if (loss.kind >= 0) {              // the loss's own gradient, derived here (see ts::phaseC_bwd_kernel)
        float dl[8];
        loss_row_grad(loss.kind, d.y_dim, d.T * d.Nq, a.mu + (rq + r) * d.y_dim, loss.gt + (rq + r) * loss.gt_dim, loss.dloss[0], dl);
        float mine = dl[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) mine = c == j ? dl[j] : mine;
        up += mine;
      }
      gv = up * act_grad_from_out(d.out_act, a.mu[(rq + r) * d.y_dim + c]);
    }
  }
  TileW<DH> td2, td1; TileW<LDD> tdec; TileW<DR> tr;
  td2.fetch(a.d2 + rq * DH, DH, d.Nq, tid);
  td1.fetch(a.d1 + rq * DH, DH, d.Nq, tid);
  tdec.fetch(a.dec_in + rq * LDD, LDD, d.Nq, tid);
  tr.fetch(a.r + (size_t)t * DR, DR, 1, tid);
  const int am = tid < DR ? a.amax[(size_t)t * DR + tid] : 0;
  Dg<4, DH> g2;  Dg<DH, DH> g1;  Dg<DH, LDD> g0;  Dg<DZ, DR> gz;
  g2.load(a.p.dec_w[2], wave, lane, d.y_dim);
  g1.load(a.p.dec_w[1], wave, lane);
  g0.load(a.p.dec_w[0], wave, lane);
  gz.load(a.p.r2z_w, wave, lane);
  if (tid < 256) s_g[(tid >> 4) * A_LY + (tid & 15)] = gv;
  td2.stash(s_d2, C_LH, tid); td1.stash(s_d1, C_LH, tid); tdec.stash(s_dec, C_LD, tid); tr.stash(s_r, N_LR, tid);
  if (tid < 128) s_am[tid] = am;
  lds_zero4(s_dzt, 16 * C_LR, tid);                // its rows 1.. stay zero
  __syncthreads();
  // the encoder half's operands: requested now, needed five stages from here
  TileW<H0> th1, th0; TileW<LDC> tcat;
  th1.fetch(a.h1 + rc * H1, H1, d.Nc, tid);
  th0.fetch(a.h0 + rc * H0, H0, d.Nc, tid);
  tcat.fetch(a.cat_in + rc * LDC, LDC, d.Nc, tid);
  float yv = 0.f;
  if (tid < 256) {
    const int r = tid >> 4, c = tid & 15;
    if (r < d.Nc && c < d.label_dim) yv = a.ctx_y[(rc + r) * d.label_dim + c];
  }
  float* sl = a.slab + (size_t)t * a.sl.total;
  // decoder0.4: d d2
  g2.finish(g2.mma(s_g, A_LY, wave, lane), s_red, s_d2, C_LH, s_dd2, C_LH, nullptr, 0, 0, wave, lane);
  th1.stash(s_h1, A_LH, tid); th0.stash(s_h0, A_LH, tid); tcat.stash(s_cat, A_LCAT, tid);
  if (tid < 256) s_y[(tid >> 4) * A_LY + (tid & 15)] = yv;
  __syncthreads();
  Dg<DR, H1> e2;  Dg<H1, H0> e1;  Dg<H0, LDC> e0;
  e2.load(a.p.er_w[2], wave, lane);
  e1.load(a.p.er_w[1], wave, lane);
  e0.load(a.p.er_w[0], wave, lane);
  // decoder0.2: d d1  |  decoder0.4 weight gradient
  g1.finish(g1.mma(s_dd2, C_LH, wave, lane), s_red, s_d1, C_LH, s_dd1, C_LH, nullptr, 0, 0, wave, lane);
  wgrad16<4, DH, GR>(s_g, A_LY, s_d2, C_LH, sl + a.sl.dec_w[2], first ? sl + a.sl.dec_b[2] : nullptr, gwave, lane, tid, d.y_dim, DH);
  __syncthreads();
  // decoder0.0: input gradient = [d x_qry | dz]  |  decoder0.2 weight gradient
  g0.finish(g0.mma(s_dd1, C_LH, wave, lane), s_red, nullptr, 0, s_ddec, C_LD, first ? a.d_dec_in + rq * LDD : nullptr, LDD, d.Nq, wave, lane);
  wgrad16<DH, DH, GR>(s_dd2, C_LH, s_d1, C_LH, sl + a.sl.dec_w[1], first ? sl + a.sl.dec_b[1] : nullptr, gwave, lane, tid);
  __syncthreads();
  // broadcast backward: dz_t = column sums over the target rows (row 0 of its tile)  |  decoder0.0 weight gradient
  if (tid < DZ) {
    float sacc = 0.f;
    for (int r = 0; r < d.Nq; ++r) sacc += s_ddec[r * C_LD + DW + tid];
    s_dzt[tid] = sacc;
  }
  wgrad16<DH, LDD, GR>(s_dd1, C_LH, s_dec, C_LD, sl + a.sl.dec_w[0], first ? sl + a.sl.dec_b[0] : nullptr, gwave, lane, tid);
  __syncthreads();
  // r_to_z: d r (row 0)  |  its weight gradient (an outer product: only row 0 of both tiles is non-zero)
  gz.finish(gz.mma(s_dzt, C_LR, wave, lane), s_red, nullptr, 0, s_dr, N_LR, nullptr, 0, 0, wave, lane);
  wgrad16<DZ, DR, GR>(s_dzt, C_LR, s_r, N_LR, sl + a.sl.r2z_w, first ? sl + a.sl.r2z_b : nullptr, gwave, lane, tid);
  __syncthreads();
  // aggregator backward: mean spreads d r / Nc over the shots, max routes it to the arg-max shot; rows beyond the shots are zero
  for (int i = tid; i < 16 * DR; i += NWV * 64) {
    const int n = i / DR, j = i - n * DR;
    const float g = s_dr[j];
    s_drs[n * N_LR + j] = n < d.Nc ? (d.agg == 0 ? g / (float)d.Nc : (s_am[j] == n ? g : 0.f)) : 0.f;
  }
  for (int i = tid; i < 16 * (N_LR - DR); i += NWV * 64) s_drs[(i / (N_LR - DR)) * N_LR + DR + i % (N_LR - DR)] = 0.f;      // its k padding
  __syncthreads();
  // EncoderFC, last layer first: d h1
  e2.finish(e2.mma(s_drs, N_LR, wave, lane), s_red, s_h1, A_LH, s_dh1, A_LH, nullptr, 0, 0, wave, lane);
  __syncthreads();
  e1.finish(e1.mma(s_dh1, A_LH, wave, lane), s_red, s_h0, A_LH, s_dh0, A_LH, nullptr, 0, 0, wave, lane);
  wgrad16<DR, H1, GR>(s_drs, N_LR, s_h1, A_LH, sl + a.sl.er_w[2], first ? sl + a.sl.er_b[2] : nullptr, gwave, lane, tid);
  __syncthreads();
  e0.finish(e0.mma(s_dh0, A_LH, wave, lane), s_red, nullptr, 0, s_dcat, A_LCAT, first ? a.d_cat_in + rc * LDC : nullptr, LDC, d.Nc, wave, lane);
  wgrad16<H1, H0, GR>(s_dh1, A_LH, s_h0, A_LH, sl + a.sl.er_w[1], first ? sl + a.sl.er_b[1] : nullptr, gwave, lane, tid);
  __syncthreads();
  wgrad16<H0, LDC, GR>(s_dh0, A_LH, s_cat, A_LCAT, sl + a.sl.er_w[0], first ? sl + a.sl.er_b[0] : nullptr, gwave, lane, tid);
  // transform_y: dW = d_cat[:, dw:]^T ctx_y, db
  if (first) wgrad16<DW / 4, 4>(s_dcat + DW, A_LCAT, s_y, A_LY, sl + a.sl.ty_w, sl + a.sl.ty_b, wave, lane, tid, DW / 4, d.label_dim);
}

}  // namespace ts
}  // namespace mlhot
#endif  // !MLHOT_HOSTSIM
