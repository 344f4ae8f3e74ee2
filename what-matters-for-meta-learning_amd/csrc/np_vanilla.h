// Whole-model forward / backward of the vanilla CNP / ANP family (c1-c4):
// CNPVanillaPascal1D, CNPShapeNet1D, ANPVanillaPascal1D, ANPShapeNet1D.
// One C call enqueues the full kernel sequence; nothing returns to Python in between.
//
// Data flow (reference: CNPShapeNet1D.py:96-140, ANPShapeNet1D.py:93-157):
//   images --E1--> x_ctx (cols 0..dw of cat_in) | x_qry (cols 0..dw of dec_in)
//   cat_in = [x_ctx | transform_y(ctx_y)] --EncoderFC--> rs
//   CNP: r = agg(rs) -> z = r_to_z(r), broadcast over targets into dec_in[:, dw:]
//   ANP: K = W_k(x_ctx), V = W_v(rs), Q = W_q(x_qry) (8 heads in one GEMM each) -> FAVOR+
//        -> merged -> _W -> r_to_z -> dec_in[:, dw:]
//   mu = decoder0(dec_in)
#pragma once
#include "encoder.h"
#include "tail_fused.h"
#include "tail_spec.h"
#include "tail_cnp.h"
#include "cnp_spec.h"
#include "linear_skinny.h"
#include "favor2.h"

namespace mlhot {

struct NpBuf {   // forward activations kept for the backward ("saved")
  void* enc;
  float *cat_in, *h[MLHOT_MAX_HIDDEN], *rs, *dec_in, *d1, *d2;
  float *r, *sigma, *zt, *mu_l, *lv; int32_t* amax;
  float *kh, *vh, *qh, *merged, *rr, *wot; void* favor; size_t favor_bytes;
  bool ok; size_t bytes;
};

inline NpBuf np_saved_carve(const mlhot_np_dims& d, void* base, size_t cap) {
  Arena a(base, cap);
  NpBuf b{};
  const size_t Rc = (size_t)d.T * d.Nc, Rq = (size_t)d.T * d.Nq;
  const int n = (int)(Rc + Rq), dw = d.dim_w, H = MLHOT_HEADS;
  b.enc = a.take<char>(enc_saved_bytes(n));
  b.dec_in = a.take<float>(Rq * (dw + d.dim_z));
  b.d1 = a.take<float>(Rq * d.dec_hidden);
  b.d2 = a.take<float>(Rq * d.dec_hidden);
  if (d.Nc > 0) {
    b.cat_in = a.take<float>(Rc * (dw + dw / 4));
    for (int i = 0; i < d.n_hidden; ++i) b.h[i] = a.take<float>(Rc * d.hidden[i]);
    b.rs = a.take<float>(Rc * d.dim_r);
    if (d.agg_mode == MLHOT_AGG_ATTENTION) {
      b.kh = a.take<float>(Rc * H * dw); b.vh = a.take<float>(Rc * H * dw); b.qh = a.take<float>(Rq * H * dw);
      b.merged = a.take<float>(Rq * H * dw); b.rr = a.take<float>(Rq * dw);
      b.wot = a.take<float>((size_t)H * dw * dw);      // head-major copy of _W's weight (fused tail)
      FavorDims f{d.T, H, d.Nq, d.Nc, dw, d.m_feat};
      b.favor_bytes = favor_ws_need(f);
      b.favor = a.take<char>(b.favor_bytes);
    } else {
      b.r = a.take<float>((size_t)d.T * d.dim_r); b.zt = a.take<float>((size_t)d.T * d.dim_z);
      b.amax = a.take<int32_t>((size_t)d.T * d.dim_r); b.sigma = a.take<float>((size_t)d.T * d.dim_r);
      if (d.agg_mode == MLHOT_AGG_BACO) { b.mu_l = a.take<float>(Rc * d.dim_r); b.lv = a.take<float>(Rc * d.dim_r); }
    }
  }
  b.ok = a.ok; b.bytes = a.off + 256;
  return b;
}

struct NpScratch {
  void* enc; size_t enc_bytes;
  float *d_dec_in, *dd1, *dd2, *d_cat_in, *dh[MLHOT_MAX_HIDDEN], *d_rs;
  float *d_rr, *d_merged, *dqh, *dkh, *dvh, *dzt, *dr, *d_mu_l, *d_lv;
  float* tail_slab;   // fused tail: per-task weight-gradient partials
  float* dmu_tmp;     // [Rq][y_dim]: upstream gradient + the loss's own, where the first backward kernel cannot take the loss itself
  bool ok; size_t bytes;
};

inline NpScratch np_scratch_carve(const mlhot_np_dims& d, void* base, size_t cap) {
  Arena a(base, cap);
  NpScratch s{};
  const size_t Rc = (size_t)d.T * d.Nc, Rq = (size_t)d.T * d.Nq;
  const int n = (int)(Rc + Rq), dw = d.dim_w, H = MLHOT_HEADS;
  s.enc_bytes = enc_scratch_bytes(n, dw);
  s.enc = a.take<char>(s.enc_bytes);
  s.d_dec_in = a.take<float>(Rq * (dw + d.dim_z));
  s.dd1 = a.take<float>(Rq * d.dec_hidden); s.dd2 = a.take<float>(Rq * d.dec_hidden);
  s.dmu_tmp = a.take<float>(Rq * d.y_dim);
  if (d.Nc > 0) {
    s.d_cat_in = a.take<float>(Rc * (dw + dw / 4));
    for (int i = 0; i < d.n_hidden; ++i) s.dh[i] = a.take<float>(Rc * d.hidden[i]);
    s.d_rs = a.take<float>(Rc * d.dim_r);
    if (d.agg_mode == MLHOT_AGG_ATTENTION) {
      s.d_rr = a.take<float>(Rq * dw); s.d_merged = a.take<float>(Rq * H * dw);
      s.dqh = a.take<float>(Rq * H * dw); s.dkh = a.take<float>(Rc * H * dw); s.dvh = a.take<float>(Rc * H * dw);
#ifndef MLHOT_HOSTSIM
      if (d.n_hidden == 2) {
        const tf::TailDims td{d.T, d.Nc, d.Nq, d.label_dim, d.y_dim, dw, d.dim_z, d.hidden[0], d.hidden[1], d.dec_hidden, 0, d.m_feat};
        s.tail_slab = a.take<float>((size_t)d.T * tf::tail_slab_layout(td).total);
      }
#endif
    } else {
      s.dzt = a.take<float>((size_t)d.T * d.dim_z); s.dr = a.take<float>((size_t)d.T * d.dim_r);
      if (d.agg_mode == MLHOT_AGG_BACO) { s.d_mu_l = a.take<float>(Rc * d.dim_r); s.d_lv = a.take<float>(Rc * d.dim_r); }
#ifndef MLHOT_HOSTSIM
      if (d.n_hidden == 2 && d.agg_mode != MLHOT_AGG_BACO) {
        const tf::CnpDims cd{d.T, d.Nc, d.Nq, d.label_dim, d.y_dim, dw, d.dim_r, d.dim_z, d.hidden[0], d.hidden[1], d.dec_hidden, 0, d.agg_mode};
        s.tail_slab = a.take<float>((size_t)d.T * tf::cnp_slab_layout(cd).total);
      }
#endif
    }
  }
  s.ok = a.ok; s.bytes = a.off + 256;
  return s;
}

inline int np_check_dims(const mlhot_np_dims& d) {
  if (d.T <= 0 || d.Nq <= 0 || d.Nc < 0 || d.n_hidden < 1 || d.n_hidden > MLHOT_MAX_HIDDEN || d.dim_w % 4 ||
      d.agg_mode < 0 || d.agg_mode > 3 || d.y_dim < 1 || d.y_dim > 8) {
    set_error("np_vanilla: bad dims"); return MLHOT_ERR_ARG;
  }
  if (d.agg_mode == MLHOT_AGG_ATTENTION && (d.dim_r != d.dim_w || d.m_feat <= 0)) {
    set_error("np_vanilla: attention needs dim_r == dim_w and m_feat > 0"); return MLHOT_ERR_ARG;
  }
  return MLHOT_OK;
}

// ---- thin wrappers over the igemm problems -------------------------------------------------
inline WBlocks wb1(const float* w, const float* b, int rows) { WBlocks x{}; x.w[0] = w; x.b[0] = b; x.rows = rows; return x; }
inline WBlocks wb8(const float* const* w, const float* const* b, int rows) {
  WBlocks x{}; for (int i = 0; i < MLHOT_HEADS; ++i) { x.w[i] = w[i]; x.b[i] = b ? b[i] : nullptr; } x.rows = rows; return x;
}
inline WBlocksMut gb1(float* w, float* b, int rows) { WBlocksMut x{}; x.w[0] = w; x.b[0] = b; x.rows = rows; return x; }
inline WBlocksMut gb8(float* const* w, float* const* b, int rows) {
  WBlocksMut x{}; for (int i = 0; i < MLHOT_HEADS; ++i) { x.w[i] = w[i]; x.b[i] = b[i]; } x.rows = rows; return x;
}

// Few-row layers with a single weight block and 16-byte aligned rows take the direct-from-global kernels (linear_skinny.h).
inline int lin_fwd(const float* x, int ldx, const WBlocks& wb, float* y, int ldy, int M, int K, int N, int act,
                   hipStream_t s, const char* what) {
#ifndef MLHOT_HOSTSIM
  if (M <= sk::MAX_ROWS && wb.w[1] == nullptr && wb.rows >= N && K % 4 == 0 && sk::aligned4(x, ldx) && sk::aligned4(wb.w[0], K))
    return sk::run_fwd(x, ldx, wb.w[0], wb.b[0], y, ldy, M, K, N, act, s, what);
#endif
  LinearFwd p{M, N, K, x, ldx, wb, y, ldy, act};
  return run_igemm_auto(p, s, what);
}
// dx[M][Kin] (+)= (dy * act'(y)) W
inline int lin_dgrad(const float* dy, int lddy, const float* y, int ldy, int act, const WBlocks& wb,
                     float* dx, int lddx, int accumulate, int M, int Kin, int Nout, hipStream_t s, const char* what) {
#ifndef MLHOT_HOSTSIM
  if (M <= sk::MAX_ROWS && wb.w[1] == nullptr && wb.rows >= Nout && Nout % 4 == 0 && sk::aligned4(dy, lddy) && (act == ACT_NONE || sk::aligned4(y, ldy)))
    return sk::run_dgrad(dy, lddy, y, ldy, act, wb.w[0], dx, lddx, accumulate, M, Kin, Nout, s, what);
#endif
  LinearDgrad p{M, Kin, Nout, dy, lddy, y, ldy, act, wb, dx, lddx, accumulate};
  return run_igemm_auto(p, s, what);
}
inline int lin_wgrad(const float* dy, int lddy, const float* y, int ldy, int act, const float* x, int ldx,
                     const WBlocksMut& gb, int M, int Kin, int Nout, hipStream_t s, const char* what) {
#ifndef MLHOT_HOSTSIM
  if (M <= sk::MAX_ROWS && gb.w[1] == nullptr && gb.rows >= Nout)
    return sk::run_wgrad(dy, lddy, y, ldy, act, x, ldx, gb.w[0], gb.b[0], M, Kin, Nout, s, what);
#endif
  LinearWgrad p{Nout, Kin + 1, M, dy, lddy, y, ldy, act, x, ldx, gb};
  return run_igemm_auto(p, s, what);
}

#ifndef MLHOT_HOSTSIM
// ---- fused tail (csrc/tail_fused.h) ---------------------------------------------------------------
inline bool tail_fused_applies(const mlhot_np_dims& d) {
  return g_opt.tail_fused && d.agg_mode == MLHOT_AGG_ATTENTION && d.Nc >= 1 && d.Nc <= 16 && d.Nq <= 16 &&
         d.n_hidden == 2 && d.dim_w % 64 == 0 && d.dim_r == d.dim_w && d.m_feat <= 4096 &&
         (long long)d.T * d.Nc * MLHOT_HEADS < (1ll << 19);      // key arg-max positions are packed row * 4096 + col
}
inline tf::TailDims tail_dims(const mlhot_np_dims& d) {
  return tf::TailDims{d.T, d.Nc, d.Nq, d.label_dim, d.y_dim, d.dim_w, d.dim_z, d.hidden[0], d.hidden[1], d.dec_hidden,
                      d.out_tanh ? ACT_TANH : ACT_NONE, d.m_feat};
}
inline tf::TailParams tail_params(const mlhot_np_params& p) {
  tf::TailParams q;
  q.ty_w = p.ty_w; q.ty_b = p.ty_b;
  for (int i = 0; i < 3; ++i) { q.er_w[i] = p.er_w[i]; q.er_b[i] = p.er_b[i]; q.dec_w[i] = p.dec_w[i]; q.dec_b[i] = p.dec_b[i]; }
  q.r2z_w = p.r2z_w; q.r2z_b = p.r2z_b;
  for (int i = 0; i < MLHOT_HEADS; ++i) {
    q.wk_w[i] = p.wk_w[i]; q.wk_b[i] = p.wk_b[i]; q.wv_w[i] = p.wv_w[i]; q.wv_b[i] = p.wv_b[i];
    q.wq_w[i] = p.wq_w[i]; q.wq_b[i] = p.wq_b[i];
  }
  q.wo_w = p.wo_w; q.wo_b = p.wo_b; q.proj = p.proj;
  return q;
}
template <class K, class A>
inline int tail_launch(K kernel, int grid, int block, size_t lds, const A& args, hipStream_t s, const char* what) {
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    set_error("%s: %zu bytes of LDS refused", what, lds);
    return MLHOT_ERR_LAUNCH;
  }
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, s, args);
  }
  return check_launch(what);
}

// does phase A of this forward run the specialised kernel (which can fold the encoder Linear's partial results itself)?
inline bool tail_phaseA_folds(const mlhot_np_dims& d) {
  static_assert(ts::XK == el::F_KS, "phase A folds the encoder Linear's split-K factor");
  return tail_fused_applies(d) && ts::applies(tail_dims(d)) && (g_opt.tail_spec & 1) && (g_opt.tail_spec & 64) && g_opt.conv2_tc && d.dim_w == el::DW;
}
inline int tail_forward_fused(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_y, float* mu,
                              const NpBuf& b, const NpScratch& sc, hipStream_t s, const Stage& st = Stage{}, const EncXFold& xf = EncXFold{nullptr, nullptr, 0, 0}) {
  const tf::TailDims td = tail_dims(d);
  const tf::TailParams tp = tail_params(p);
  FavorDims f{d.T, MLHOT_HEADS, d.Nq, d.Nc, d.dim_w, d.m_feat};
  FavorWs w = favor_carve(f, b.favor, b.favor_bytes);
  if (!w.ok || !sc.d_merged) { set_error("tail_fused: workspace"); return MLHOT_ERR_WORKSPACE; }
  // the kernels specialised for the shipped dimensions (csrc/tail_spec.h) where they apply; the option is a bit mask over the
  // phases (1 / 2 / 4: forward A / B / C, 8 / 16 / 32: backward C / B / A; default 63 = all; per-phase A/B experiments)
  const int spec = ts::applies(td) ? g_opt.tail_spec : 0;
  tf::PhaseAArgs a{g_opt.dbg, td, tp, ctx_y, b.cat_in, b.h[0], b.h[1], b.rs, b.dec_in, b.kh, w.pc, w.max_k, w.arg_k, b.wot, xf.slab, xf.bias, xf.k, xf.n};
  if (xf.slab != nullptr && !(spec & 1)) { set_error("tail_fused: the encoder left its fold to a phase A that cannot do it"); return MLHOT_ERR_ARG; }
  if (st.first()) {
    if (spec & 1) MLHOT_TRY(tail_launch(ts::phaseA_fwd_kernel, d.T + d.T * MLHOT_HEADS + (xf.slab != nullptr ? d.T : 0), 512, ts::phaseA_lds_bytes(), a, s, "tail.A"));
    else MLHOT_TRY(tail_launch(tf::phaseA_fwd_kernel, d.T + d.T * MLHOT_HEADS, 512, tf::phaseA_lds_bytes(td), a, s, "tail.A"));
  }
  // strict sharded parity (stab_xchg.h): phase B folds the (task, head) shares (max_k, arg_k) into the batch-global key stabiliser
  if (st.stage == 0) return sx::max_publish(w.max_k, d.T * MLHOT_HEADS, st.x, s);
  if (st.stage == 1) MLHOT_TRY(sx::max_apply(w.max_k, w.arg_k, d.T * MLHOT_HEADS, st.x, s));
  tf::PhaseBArgs bb{td, tp, b.dec_in, b.rs, b.qh, b.vh, b.kh, w.pc, w.max_k, w.arg_k, w.qf, w.kf, w.S, w.D, w.gmax, w.arg_q, w.gpos, b.merged, sc.d_merged, b.wot};   // sc.d_merged: forward scratch for the heads' _W shares
  if (spec & 2) MLHOT_TRY(tail_launch(ts::phaseB_fwd_kernel, d.T * MLHOT_HEADS, 512, ts::phaseB_lds_bytes(), bb, s, "tail.B"));
  else MLHOT_TRY(tail_launch(tf::phaseB_fwd_kernel, d.T * MLHOT_HEADS, 512, tf::phaseB_lds_bytes(td), bb, s, "tail.B"));
  tf::PhaseCArgs c{td, tp, sc.d_merged, b.rr, b.dec_in, b.d1, b.d2, mu};
  if (spec & 4) MLHOT_TRY(tail_launch(ts::phaseC_fwd_kernel, d.T, 512, ts::phaseC_lds_bytes(), c, s, "tail.C"));
  else MLHOT_TRY(tail_launch(tf::phaseC_fwd_kernel, d.T, 512, tf::phaseC_lds_bytes(td), c, s, "tail.C"));
  return MLHOT_OK;
}

// ---- fused CNP tail (csrc/tail_cnp.h) ---------------------------------------------------------------
inline bool cnp_fused_applies(const mlhot_np_dims& d) {
  return g_opt.tail_fused && (d.agg_mode == MLHOT_AGG_MEAN || d.agg_mode == MLHOT_AGG_MAX) && d.Nc >= 1 && d.Nc <= 16 &&
         d.Nq <= 16 && d.n_hidden == 2 && d.dim_w % 16 == 0 && d.dim_r <= 128;
}
inline tf::CnpDims cnp_dims(const mlhot_np_dims& d) {
  return tf::CnpDims{d.T, d.Nc, d.Nq, d.label_dim, d.y_dim, d.dim_w, d.dim_r, d.dim_z, d.hidden[0], d.hidden[1], d.dec_hidden,
                     d.out_tanh ? ACT_TANH : ACT_NONE, d.agg_mode};
}
inline tf::CnpParams cnp_params(const mlhot_np_params& p) {
  tf::CnpParams q;
  q.ty_w = p.ty_w; q.ty_b = p.ty_b; q.r2z_w = p.r2z_w; q.r2z_b = p.r2z_b;
  for (int i = 0; i < 3; ++i) { q.er_w[i] = p.er_w[i]; q.er_b[i] = p.er_b[i]; q.dec_w[i] = p.dec_w[i]; q.dec_b[i] = p.dec_b[i]; }
  return q;
}
// the kernels specialised for the shipped CNP dimensions (csrc/cnp_spec.h) where they apply; option tail_spec bits as for the
// attention tail: 1 forward, 8 backward, 64 the forward folds the encoder Linear's partial results, 128 the backward takes the loss's
// gradient from a descriptor, 1024 / 4096 two / four workgroups per task in the backward
inline bool cnp_spec_applies(const mlhot_np_dims& d) { return cnp_fused_applies(d) && ts::cnp_applies(cnp_dims(d)); }
inline bool cnp_spec_folds(const mlhot_np_dims& d) {
  return cnp_spec_applies(d) && (g_opt.tail_spec & 1) && (g_opt.tail_spec & 64) && g_opt.conv2_tc && d.dim_w == el::DW;
}
template <class K, class... A>
inline int cnp_spec_launch(K kernel, int grid, size_t lds, hipStream_t s, const char* what, const A&... args) {
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    set_error("%s: %zu bytes of LDS refused", what, lds);
    return MLHOT_ERR_LAUNCH;
  }
  {
    ProfScope ps(what, s);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, s, args...);
  }
  return check_launch(what);
}
inline int cnp_forward_fused(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_y, float* mu, const NpBuf& b, hipStream_t s,
                             const EncXFold& xf = EncXFold{nullptr, nullptr, 0, 0}) {
  const tf::CnpDims cd = cnp_dims(d);
  tf::CnpFwdArgs a{cd, cnp_params(p), ctx_y, b.cat_in, b.h[0], b.h[1], b.rs, b.r, b.zt, b.dec_in, b.d1, b.d2, mu, b.amax};
  if (ts::cnp_applies(cd) && (g_opt.tail_spec & 1))
    return cnp_spec_launch(ts::cnp_fwd_kernel, d.T, ts::cnp_fwd_lds_bytes(), s, "tail.cnp", a, ts::CnpXFold{xf.slab, xf.bias, xf.k, xf.n});
  if (xf.slab != nullptr) { set_error("cnp_fused: the encoder left its fold to a tail kernel that cannot do it"); return MLHOT_ERR_ARG; }
  return tail_launch(tf::cnp_fwd_kernel, d.T, 512, tf::cnp_fwd_lds_bytes(cd), a, s, "tail.cnp");
}
// Per-task slabs -> parameter gradients.  When the caller laid the gradient tensors out as ONE flat buffer in slab
// order (mlhot_np_grads_flat_layout) the reduce is a single contiguous float4 sum; otherwise every element looks up
// its destination segment.
// `later`: where a contiguous sum may be parked instead of launched (the encoder backward that follows folds it into its own
// final reduce launch).
inline int tail_slab_reduce(tf::SlabReduce& r, hipStream_t s, PendingSum* later = nullptr) {
  bool flat = r.nseg > 0 && r.off[0] == 0 && (reinterpret_cast<uintptr_t>(r.dst[0]) & 15) == 0 && (r.total & 3) == 0;
  for (int i = 1; flat && i < r.nseg; ++i) flat = r.dst[i] == r.dst[0] + r.off[i];
  if (flat && later != nullptr) {
    *later = PendingSum{r.slab, r.dst[0], r.T, r.total, r.total};
    return MLHOT_OK;
  }
  {
    ProfScope ps("tail.bwd.reduce", s);
    if (flat) hipLaunchKernelGGL(c2::sum_parts_kernel, dim3((r.total + 63) / 64), dim3(256), 0, s, r.slab, r.T, r.total, r.dst[0], r.total);
    else hipLaunchKernelGGL(tf::slab_to_grads_kernel, dim3((r.total + 255) / 256), dim3(256), 0, s, r);
  }
  return check_launch("tail.bwd.reduce");
}

inline int cnp_backward_fused(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_y, const float* mu, const float* dmu,
                              const mlhot_np_grads& g, const NpBuf& b, const NpScratch& sc, hipStream_t s, PendingSum* later,
                              const LossDesc& loss = LossDesc{-1, nullptr, 0, nullptr, nullptr}) {
  const tf::CnpDims cd = cnp_dims(d);
  const tf::CnpSlab sl = tf::cnp_slab_layout(cd);
  if (!sc.tail_slab) { set_error("cnp_fused: workspace"); return MLHOT_ERR_WORKSPACE; }
  tf::CnpBwdArgs a{cd, cnp_params(p), sl, ctx_y, dmu, mu, b.d2, b.d1, b.dec_in, b.r, b.rs, b.h[1], b.h[0], b.cat_in, b.amax,
                   sc.d_dec_in, sc.d_cat_in, sc.tail_slab};
  if (ts::cnp_applies(cd) && (g_opt.tail_spec & 8)) {
    const int gr = (g_opt.tail_spec & 1024) ? ((g_opt.tail_spec & 4096) ? 4 : 2) : 1;
    const int lv = loss.value != nullptr ? 1 : 0;           // one workgroup more: the loss value
    if (gr == 4) MLHOT_TRY(cnp_spec_launch(ts::cnp_bwd_kernel<4>, 4 * d.T + lv, ts::cnp_bwd_lds_bytes(), s, "tail.bwd.cnp", a, loss));
    else if (gr == 2) MLHOT_TRY(cnp_spec_launch(ts::cnp_bwd_kernel<2>, 2 * d.T + lv, ts::cnp_bwd_lds_bytes(), s, "tail.bwd.cnp", a, loss));
    else MLHOT_TRY(cnp_spec_launch(ts::cnp_bwd_kernel<1>, d.T + lv, ts::cnp_bwd_lds_bytes(), s, "tail.bwd.cnp", a, loss));
  } else {
    if (loss.kind >= 0) { set_error("cnp_fused: a loss descriptor reached a backward kernel that cannot take it"); return MLHOT_ERR_ARG; }
    MLHOT_TRY(tail_launch(tf::cnp_bwd_kernel, d.T, 512, tf::cnp_bwd_lds_bytes(cd), a, s, "tail.bwd.cnp"));
  }
  tf::SlabReduce r{};
  int ns = 0, maxlen = 0;
  auto seg = [&](float* dst, int off, int len) { r.dst[ns] = dst; r.off[ns] = off; r.len[ns] = len; if (len > maxlen) maxlen = len; ++ns; };
  const int dw = d.dim_w, ldc = dw + dw / 4, ldd = dw + d.dim_z;
  seg(g.ty_w, sl.ty_w, dw / 4 * d.label_dim); seg(g.ty_b, sl.ty_b, dw / 4);
  seg(g.er_w[0], sl.er_w[0], cd.h0 * ldc); seg(g.er_b[0], sl.er_b[0], cd.h0);
  seg(g.er_w[1], sl.er_w[1], cd.h1 * cd.h0); seg(g.er_b[1], sl.er_b[1], cd.h1);
  seg(g.er_w[2], sl.er_w[2], cd.dr * cd.h1); seg(g.er_b[2], sl.er_b[2], cd.dr);
  seg(g.r2z_w, sl.r2z_w, d.dim_z * cd.dr); seg(g.r2z_b, sl.r2z_b, d.dim_z);
  seg(g.dec_w[0], sl.dec_w[0], cd.dec_h * ldd); seg(g.dec_b[0], sl.dec_b[0], cd.dec_h);
  seg(g.dec_w[1], sl.dec_w[1], cd.dec_h * cd.dec_h); seg(g.dec_b[1], sl.dec_b[1], cd.dec_h);
  seg(g.dec_w[2], sl.dec_w[2], d.y_dim * cd.dec_h); seg(g.dec_b[2], sl.dec_b[2], d.y_dim);
  r.nseg = ns; r.T = d.T; r.total = sl.total; r.slab = sc.tail_slab;
  (void)maxlen;
  return tail_slab_reduce(r, s, later);
}

inline int tail_backward_fused(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_y, const float* mu,
                               const float* dmu, const mlhot_np_grads& g, const NpBuf& b, const NpScratch& sc, hipStream_t s,
                               PendingSum* later, const Stage& st = Stage{}, const LossDesc& loss = LossDesc{-1, nullptr, 0, nullptr, nullptr}) {
  const tf::TailDims td = tail_dims(d);
  const tf::TailParams tp = tail_params(p);
  const tf::TailSlab sl = tf::tail_slab_layout(td);
  FavorDims f{d.T, MLHOT_HEADS, d.Nq, d.Nc, d.dim_w, d.m_feat};
  FavorWs w = favor_carve(f, b.favor, b.favor_bytes);
  if (!w.ok || !sc.tail_slab) { set_error("tail_fused: workspace"); return MLHOT_ERR_WORKSPACE; }
  float* part_k = w.rsum_k;   // [T*H]
  tf::PhaseCBwdArgs c{td, tp, sl, dmu, mu, b.d2, b.d1, b.dec_in, b.rr, sc.d_dec_in, sc.d_rr, sc.tail_slab, loss};
  const int spec = ts::applies(td) ? g_opt.tail_spec : 0;
  if (loss.kind >= 0 && !(spec & 8)) { set_error("tail_fused: a loss descriptor reached a phase C' that cannot take it"); return MLHOT_ERR_ARG; }
  if (st.first()) {
    const int lv = loss.value != nullptr ? 1 : 0;           // one workgroup more: the loss value
    if ((spec & 8) && (spec & 1024) && (spec & 4096)) MLHOT_TRY(tail_launch(ts::phaseC_bwd_kernel<4>, 4 * d.T + lv, 512, ts::phaseC_bwd_lds_bytes(), c, s, "tail.bwd.C"));
    else if ((spec & 8) && (spec & 1024)) MLHOT_TRY(tail_launch(ts::phaseC_bwd_kernel<2>, 2 * d.T + lv, 512, ts::phaseC_bwd_lds_bytes(), c, s, "tail.bwd.C"));
    else if (spec & 8) MLHOT_TRY(tail_launch(ts::phaseC_bwd_kernel<1>, d.T + lv, 512, ts::phaseC_bwd_lds_bytes(), c, s, "tail.bwd.C"));
    else MLHOT_TRY(tail_launch(tf::phaseC_bwd_kernel, d.T, 512, tf::phaseC_bwd_lds_bytes(td), c, s, "tail.bwd.C"));
  }
  // sc.dqh / dkh / dvh double as the heads' input-gradient shares [T*H][N][dw] (same sizes)
  tf::PhaseBBwdArgs bb{td, tp, sl, b.qh, b.kh, b.vh, w.pc, w.qf, w.kf, w.S, w.D, b.merged, sc.d_rr, w.arg_q,
                       b.dec_in, b.cat_in, b.rs, b.wot, sc.dqh, sc.dkh, sc.dvh, part_k, sc.tail_slab};
  if (st.first()) {
    if ((spec & 16) && (spec & 512)) MLHOT_TRY(tail_launch(ts::phaseB_bwd_kernel<true>, 2 * d.T * MLHOT_HEADS, 512, ts::phaseB_bwd_lds_bytes(), bb, s, "tail.bwd.B"));
    else if (spec & 16) MLHOT_TRY(tail_launch(ts::phaseB_bwd_kernel<false>, d.T * MLHOT_HEADS, 512, ts::phaseB_bwd_lds_bytes(), bb, s, "tail.bwd.B"));
    else MLHOT_TRY(tail_launch(tf::phaseB_bwd_kernel, d.T * MLHOT_HEADS, 512, tf::phaseB_bwd_lds_bytes(td), bb, s, "tail.bwd.B"));
  }
  // strict sharded parity: phase A folds part_k into the stabiliser's gradient and routes it to the arg-max key's task
  if (st.stage == 0) return sx::sum_publish(part_k, d.T * MLHOT_HEADS, st.x, s);
  if (st.stage == 1) MLHOT_TRY(sx::sum_apply(part_k, d.T * MLHOT_HEADS, st.x, part_k, 1, s));
  tf::PhaseABwdArgs a{td, tp, sl, ctx_y, b.cat_in, b.h[0], b.h[1], sc.dqh, sc.dkh, sc.dvh, w.pc, part_k, w.gpos,
                      sc.d_dec_in, sc.d_cat_in, sc.tail_slab};
  if ((spec & 32) && (spec & 2048) && (spec & 4096)) MLHOT_TRY(tail_launch(ts::phaseA_bwd_kernel<4>, 4 * d.T, 512, ts::phaseA_bwd_lds_bytes(), a, s, "tail.bwd.A"));
  else if ((spec & 32) && (spec & 2048)) MLHOT_TRY(tail_launch(ts::phaseA_bwd_kernel<2>, 2 * d.T, 512, ts::phaseA_bwd_lds_bytes(), a, s, "tail.bwd.A"));
  else if (spec & 32) MLHOT_TRY(tail_launch(ts::phaseA_bwd_kernel<1>, d.T, 512, ts::phaseA_bwd_lds_bytes(), a, s, "tail.bwd.A"));
  else MLHOT_TRY(tail_launch(tf::phaseA_bwd_kernel, d.T, 512, tf::phaseA_bwd_lds_bytes(td), a, s, "tail.bwd.A"));
  // per-task slabs -> parameter gradients
  tf::SlabReduce r{};
  int ns = 0, maxlen = 0;
  auto seg = [&](float* dst, int off, int len) { r.dst[ns] = dst; r.off[ns] = off; r.len[ns] = len; if (len > maxlen) maxlen = len; ++ns; };
  const int dw = d.dim_w, ldc = dw + dw / 4, ldd = dw + d.dim_z;
  seg(g.ty_w, sl.ty_w, dw / 4 * d.label_dim); seg(g.ty_b, sl.ty_b, dw / 4);
  seg(g.er_w[0], sl.er_w[0], td.h0 * ldc); seg(g.er_b[0], sl.er_b[0], td.h0);
  seg(g.er_w[1], sl.er_w[1], td.h1 * td.h0); seg(g.er_b[1], sl.er_b[1], td.h1);
  seg(g.er_w[2], sl.er_w[2], dw * td.h1); seg(g.er_b[2], sl.er_b[2], dw);
  seg(g.r2z_w, sl.r2z_w, d.dim_z * dw); seg(g.r2z_b, sl.r2z_b, d.dim_z);
  seg(g.dec_w[0], sl.dec_w[0], td.dec_h * ldd); seg(g.dec_b[0], sl.dec_b[0], td.dec_h);
  seg(g.dec_w[1], sl.dec_w[1], td.dec_h * td.dec_h); seg(g.dec_b[1], sl.dec_b[1], td.dec_h);
  seg(g.dec_w[2], sl.dec_w[2], d.y_dim * td.dec_h); seg(g.dec_b[2], sl.dec_b[2], d.y_dim);
  for (int i = 0; i < MLHOT_HEADS; ++i) {
    seg(g.wk_w[i], sl.wk_w + i * dw * dw, dw * dw); seg(g.wk_b[i], sl.wk_b + i * dw, dw);
    seg(g.wv_w[i], sl.wv_w + i * dw * dw, dw * dw); seg(g.wv_b[i], sl.wv_b + i * dw, dw);
    seg(g.wq_w[i], sl.wq_w + i * dw * dw, dw * dw); seg(g.wq_b[i], sl.wq_b + i * dw, dw);
  }
  seg(g.wo_w, sl.wo_w, dw * MLHOT_HEADS * dw); seg(g.wo_b, sl.wo_b, dw);
  r.nseg = ns; r.T = d.T; r.total = sl.total; r.slab = sc.tail_slab;
  (void)maxlen;
  return tail_slab_reduce(r, s, later);
}
#endif

// Layout of ONE flat gradient buffer: the fused tails' parameters at their slab offsets (so the per-task slabs reduce
// with a single contiguous sum), everything else packed behind, 4-float aligned.  `o` receives BYTE offsets in its
// pointer fields (fields of parameters the model does not have stay null); returns the buffer size in floats.
inline size_t np_grads_flat_layout(const mlhot_np_dims& d, mlhot_np_grads& o) {
  memset(&o, 0, sizeof(o));
  const int dw = d.dim_w, ldc = dw + dw / 4, ldd = dw + d.dim_z, H = MLHOT_HEADS;
  const bool attn = d.agg_mode == MLHOT_AGG_ATTENTION, baco = d.agg_mode == MLHOT_AGG_BACO;
  size_t next = 0;
  auto at = [](size_t off_floats) { return reinterpret_cast<float*>(off_floats * sizeof(float)); };
  auto take = [&](size_t n) { const size_t r = next; next += (n + 3) / 4 * 4; return at(r); };
  bool tail_done = false;
#ifndef MLHOT_HOSTSIM
  if (d.Nc > 0 && tail_fused_applies(d)) {
    const tf::TailSlab sl = tf::tail_slab_layout(tail_dims(d));
    o.ty_w = at(sl.ty_w); o.ty_b = at(sl.ty_b);
    for (int i = 0; i < 3; ++i) { o.er_w[i] = at(sl.er_w[i]); o.er_b[i] = at(sl.er_b[i]); o.dec_w[i] = at(sl.dec_w[i]); o.dec_b[i] = at(sl.dec_b[i]); }
    o.r2z_w = at(sl.r2z_w); o.r2z_b = at(sl.r2z_b);
    for (int h = 0; h < H; ++h) {
      o.wk_w[h] = at(sl.wk_w + (size_t)h * dw * dw); o.wk_b[h] = at(sl.wk_b + (size_t)h * dw);
      o.wv_w[h] = at(sl.wv_w + (size_t)h * dw * dw); o.wv_b[h] = at(sl.wv_b + (size_t)h * dw);
      o.wq_w[h] = at(sl.wq_w + (size_t)h * dw * dw); o.wq_b[h] = at(sl.wq_b + (size_t)h * dw);
    }
    o.wo_w = at(sl.wo_w); o.wo_b = at(sl.wo_b);
    next = sl.total; tail_done = true;
  } else if (d.Nc > 0 && cnp_fused_applies(d)) {
    const tf::CnpSlab sl = tf::cnp_slab_layout(cnp_dims(d));
    o.ty_w = at(sl.ty_w); o.ty_b = at(sl.ty_b);
    for (int i = 0; i < 3; ++i) { o.er_w[i] = at(sl.er_w[i]); o.er_b[i] = at(sl.er_b[i]); o.dec_w[i] = at(sl.dec_w[i]); o.dec_b[i] = at(sl.dec_b[i]); }
    o.r2z_w = at(sl.r2z_w); o.r2z_b = at(sl.r2z_b);
    next = sl.total; tail_done = true;
  }
#endif
  if (!tail_done) {
    o.ty_w = take((size_t)dw / 4 * d.label_dim); o.ty_b = take(dw / 4);
    int in = ldc;
    for (int i = 0; i < d.n_hidden; ++i) { o.er_w[i] = take((size_t)d.hidden[i] * in); o.er_b[i] = take(d.hidden[i]); in = d.hidden[i]; }
    o.er_w[d.n_hidden] = take((size_t)d.dim_r * in); o.er_b[d.n_hidden] = take(d.dim_r);
    o.r2z_w = take((size_t)d.dim_z * d.dim_r); o.r2z_b = take(d.dim_z);
    o.dec_w[0] = take((size_t)d.dec_hidden * ldd); o.dec_b[0] = take(d.dec_hidden);
    o.dec_w[1] = take((size_t)d.dec_hidden * d.dec_hidden); o.dec_b[1] = take(d.dec_hidden);
    o.dec_w[2] = take((size_t)d.y_dim * d.dec_hidden); o.dec_b[2] = take(d.y_dim);
    if (attn) {
      for (int h = 0; h < H; ++h) {
        o.wk_w[h] = take((size_t)dw * dw); o.wk_b[h] = take(dw); o.wv_w[h] = take((size_t)dw * dw); o.wv_b[h] = take(dw);
        o.wq_w[h] = take((size_t)dw * dw); o.wq_b[h] = take(dw);
      }
      o.wo_w = take((size_t)dw * H * dw); o.wo_b = take(dw);
    }
  }
  if (baco) {
    o.mu_w = take((size_t)d.dim_r * d.dim_r); o.mu_b = take(d.dim_r); o.var_w = take((size_t)d.dim_r * d.dim_r); o.var_b = take(d.dim_r);
  }
  o.enc.w1 = take(32 * 9); o.enc.b1 = take(32); o.enc.w2 = take(48 * 288); o.enc.b2 = take(48);
  o.enc.w3 = take(64 * 432); o.enc.b3 = take(64); o.enc.wl = take((size_t)dw * 4096); o.enc.bl = take(dw);
  return next;
}

inline int np_forward(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_x, const float* ctx_y,
                      const float* qry_x, float* mu, void* saved, void* scratch, size_t scratch_bytes, hipStream_t s,
                      const Stage& st = Stage{}) {
  MLHOT_TRY(np_check_dims(d));
  MLHOT_TRY(stage_check(st, "np_vanilla_fwd"));
  NpBuf b = np_saved_carve(d, saved, (size_t)-1 / 2);
  NpScratch sc = np_scratch_carve(d, scratch, scratch_bytes);
  if (!sc.ok) { set_error("np_vanilla_fwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes); return MLHOT_ERR_WORKSPACE; }
  const int Rc = d.T * d.Nc, Rq = d.T * d.Nq, dw = d.dim_w, H = MLHOT_HEADS;
  const int ldc = dw + dw / 4, ldd = dw + d.dim_z;

  // E1 on [context | target] images in one pass; rows land in cat_in / dec_in
#ifndef MLHOT_HOSTSIM
  const bool fused = tail_fused_applies(d);
#else
  const bool fused = false;
#endif
  if (st.staged() && !fused) {      // the staged pass is built on the fused attention tail's launch boundaries
    set_error("np_vanilla_fwd: staged passes need the fused attention tail (attention aggregation, Nc, Nq <= 16, option tail_fused)");
    return MLHOT_ERR_UNSUPPORTED;
  }
#ifndef MLHOT_HOSTSIM
  EncXFold xf{nullptr, nullptr, 0, 0};
  const bool cnp_folds = !fused && cnp_spec_folds(d);
  if (st.first()) MLHOT_TRY(enc_forward(ctx_x, Rc, qry_x, Rq, p.enc, dw, Rows2{b.cat_in, ldc, Rc, b.dec_in, ldd}, b.enc, sc.enc, sc.enc_bytes, s,
                                        (fused && tail_phaseA_folds(d)) || cnp_folds ? &xf : nullptr));
  if (fused) return tail_forward_fused(d, p, ctx_y, mu, b, sc, s, st, xf);
  if (cnp_fused_applies(d)) return cnp_forward_fused(d, p, ctx_y, mu, b, s, xf);
#else
  if (st.first()) MLHOT_TRY(enc_forward(ctx_x, Rc, qry_x, Rq, p.enc, dw, Rows2{b.cat_in, ldc, Rc, b.dec_in, ldd}, b.enc, sc.enc, sc.enc_bytes, s));
#endif

  if (d.Nc > 0) {
    MLHOT_TRY(lin_fwd(ctx_y, d.label_dim, wb1(p.ty_w, p.ty_b, dw / 4), b.cat_in + dw, ldc, Rc, d.label_dim, dw / 4, ACT_NONE, s, "np.transform_y"));
    const float* x = b.cat_in; int ldx = ldc, kin = ldc;
    for (int i = 0; i < d.n_hidden; ++i) {
      MLHOT_TRY(lin_fwd(x, ldx, wb1(p.er_w[i], p.er_b[i], d.hidden[i]), b.h[i], d.hidden[i], Rc, kin, d.hidden[i], ACT_RELU, s, "np.encoder_r"));
      x = b.h[i]; ldx = kin = d.hidden[i];
    }
    MLHOT_TRY(lin_fwd(x, ldx, wb1(p.er_w[d.n_hidden], p.er_b[d.n_hidden], d.dim_r), b.rs, d.dim_r, Rc, kin, d.dim_r, ACT_NONE, s, "np.encoder_r.out"));

    if (d.agg_mode == MLHOT_AGG_ATTENTION) {
      MLHOT_TRY(lin_fwd(b.cat_in, ldc, wb8(p.wk_w, p.wk_b, dw), b.kh, H * dw, Rc, dw, H * dw, ACT_NONE, s, "np.W_k"));
      MLHOT_TRY(lin_fwd(b.rs, d.dim_r, wb8(p.wv_w, p.wv_b, dw), b.vh, H * dw, Rc, dw, H * dw, ACT_NONE, s, "np.W_v"));
      MLHOT_TRY(lin_fwd(b.dec_in, ldd, wb8(p.wq_w, p.wq_b, dw), b.qh, H * dw, Rq, dw, H * dw, ACT_NONE, s, "np.W_q"));
      FavorDims f{d.T, H, d.Nq, d.Nc, dw, d.m_feat};
      MLHOT_TRY(favor_fwd_any(f, b.qh, b.kh, b.vh, p.proj, b.merged, b.favor, b.favor_bytes, s));
      MLHOT_TRY(lin_fwd(b.merged, H * dw, wb1(p.wo_w, p.wo_b, dw), b.rr, dw, Rq, H * dw, dw, ACT_NONE, s, "np.W"));
      MLHOT_TRY(lin_fwd(b.rr, dw, wb1(p.r2z_w, p.r2z_b, d.dim_z), b.dec_in + dw, ldd, Rq, d.dim_r, d.dim_z, ACT_NONE, s, "np.r_to_z"));
    } else {
      const float* src = b.rs;
      if (d.agg_mode == MLHOT_AGG_BACO) {
        MLHOT_TRY(lin_fwd(b.rs, d.dim_r, wb1(p.mu_w, p.mu_b, d.dim_r), b.mu_l, d.dim_r, Rc, d.dim_r, d.dim_r, ACT_NONE, s, "np.rs_to_mu"));
        MLHOT_TRY(lin_fwd(b.rs, d.dim_r, wb1(p.var_w, p.var_b, d.dim_r), b.lv, d.dim_r, Rc, d.dim_r, d.dim_r, ACT_NONE, s, "np.rs_to_var"));
        src = b.mu_l;
      }
      MLHOT_TRY(run_foreach(AggFwd{d.agg_mode, d.Nc, d.dim_r, src, b.lv, b.r, b.sigma, b.amax}, (size_t)d.T * d.dim_r, s, "np.agg"));
      MLHOT_TRY(lin_fwd(b.r, d.dim_r, wb1(p.r2z_w, p.r2z_b, d.dim_z), b.zt, d.dim_z, d.T, d.dim_r, d.dim_z, ACT_NONE, s, "np.r_to_z"));
      MLHOT_TRY(run_foreach(BcastRows{b.zt, d.dim_z, d.Nq, b.dec_in + dw, ldd}, (size_t)Rq * d.dim_z, s, "np.bcast_z"));
    }
  } else {
    MLHOT_TRY(run_foreach(Fill2D{b.dec_in + dw, ldd, d.dim_z, 0.f}, (size_t)Rq * d.dim_z, s, "np.zero_z"));
  }
  MLHOT_TRY(lin_fwd(b.dec_in, ldd, wb1(p.dec_w[0], p.dec_b[0], d.dec_hidden), b.d1, d.dec_hidden, Rq, ldd, d.dec_hidden, ACT_RELU, s, "np.decoder0.0"));
  MLHOT_TRY(lin_fwd(b.d1, d.dec_hidden, wb1(p.dec_w[1], p.dec_b[1], d.dec_hidden), b.d2, d.dec_hidden, Rq, d.dec_hidden, d.dec_hidden, ACT_RELU, s, "np.decoder0.2"));
  MLHOT_TRY(lin_fwd(b.d2, d.dec_hidden, wb1(p.dec_w[2], p.dec_b[2], d.y_dim), mu, d.y_dim, Rq, d.dec_hidden, d.y_dim,
                    d.out_tanh ? ACT_TANH : ACT_NONE, s, "np.decoder0.4"));
  return MLHOT_OK;
}

inline int np_backward(const mlhot_np_dims& d, const mlhot_np_params& p, const float* ctx_x, const float* ctx_y,
                       const float* qry_x, const float* mu, const float* dmu, const mlhot_np_grads& g,
                       const void* saved, void* scratch, size_t scratch_bytes, hipStream_t s, const Stage& st = Stage{},
                       const LossDesc* loss = nullptr) {
  MLHOT_TRY(np_check_dims(d));
  MLHOT_TRY(stage_check(st, "np_vanilla_bwd"));
  NpBuf b = np_saved_carve(d, (void*)saved, (size_t)-1 / 2);
  NpScratch sc = np_scratch_carve(d, scratch, scratch_bytes);
  if (!sc.ok) { set_error("np_vanilla_bwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes); return MLHOT_ERR_WORKSPACE; }
  const int Rc = d.T * d.Nc, Rq = d.T * d.Nq, dw = d.dim_w, H = MLHOT_HEADS, dh = d.dec_hidden;
  const int ldc = dw + dw / 4, ldd = dw + d.dim_z;
  const int out_act = d.out_tanh ? ACT_TANH : ACT_NONE;

  // The loss's gradient, when the caller left it to this call (mlhot_np_vanilla_bwd_loss): the specialised phase C' derives it in
  // its prologue; every other first kernel gets it materialised (dmu_tmp = dmu + d loss / d mu, one launch as mlhot_loss_bwd's).
  LossDesc in_kernel{-1, nullptr, 0, nullptr, nullptr};
  if (loss != nullptr) {
    if (st.staged()) { set_error("np_vanilla_bwd: the staged pass takes dmu, not a loss descriptor"); return MLHOT_ERR_UNSUPPORTED; }
#ifndef MLHOT_HOSTSIM
    if (tail_fused_applies(d) && ts::applies(tail_dims(d)) && (g_opt.tail_spec & 8) && (g_opt.tail_spec & 128)) in_kernel = *loss;
    if (!tail_fused_applies(d) && cnp_spec_applies(d) && (g_opt.tail_spec & 8) && (g_opt.tail_spec & 128)) in_kernel = *loss;
#endif
    if (in_kernel.kind < 0) {
      // (the loss VALUE, when the caller left that to this call as well: the launch mlhot_loss_fwd would have made)
      if (loss->value != nullptr) MLHOT_TRY(run_reduce1(LossRed{loss->kind, d.y_dim, loss->gt_dim, Rq, mu, loss->gt, loss->value}, Rq, s, "loss_fwd"));
      MLHOT_TRY(run_foreach(LossBwd{loss->kind, d.y_dim, loss->gt_dim, Rq, mu, loss->gt, loss->dloss, sc.dmu_tmp, dmu}, (size_t)Rq, s, "loss_bwd"));
      dmu = sc.dmu_tmp;
    }
  } else if (dmu == nullptr) { set_error("np_vanilla_bwd: null dmu"); return MLHOT_ERR_ARG; }
#ifndef MLHOT_HOSTSIM
  if (st.staged() && !tail_fused_applies(d)) {
    set_error("np_vanilla_bwd: staged passes need the fused attention tail (attention aggregation, Nc, Nq <= 16, option tail_fused)");
    return MLHOT_ERR_UNSUPPORTED;
  }
  if (tail_fused_applies(d) || cnp_fused_applies(d)) {
    PendingSum tail_sum{};       // the tail's per-task slabs: summed by the encoder backward's final reduce launch when contiguous
    if (tail_fused_applies(d)) MLHOT_TRY(tail_backward_fused(d, p, ctx_y, mu, dmu, g, b, sc, s, &tail_sum, st, in_kernel));
    else MLHOT_TRY(cnp_backward_fused(d, p, ctx_y, mu, dmu, g, b, sc, s, &tail_sum, in_kernel));
    if (st.stage == 0) return MLHOT_OK;
    return enc_backward(ctx_x, Rc, qry_x, Rq, p.enc, dw, Rows2{sc.d_cat_in, ldc, Rc, sc.d_dec_in, ldd}, b.enc, g.enc, sc.enc, sc.enc_bytes, s,
                        &tail_sum);
  }
#endif
  // decoder0
  MLHOT_TRY(lin_wgrad(dmu, d.y_dim, mu, d.y_dim, out_act, b.d2, dh, gb1(g.dec_w[2], g.dec_b[2], d.y_dim), Rq, dh, d.y_dim, s, "np.bwd.dec4.w"));
  MLHOT_TRY(lin_dgrad(dmu, d.y_dim, mu, d.y_dim, out_act, wb1(p.dec_w[2], nullptr, d.y_dim), sc.dd2, dh, 0, Rq, dh, d.y_dim, s, "np.bwd.dec4.x"));
  MLHOT_TRY(lin_wgrad(sc.dd2, dh, b.d2, dh, ACT_RELU, b.d1, dh, gb1(g.dec_w[1], g.dec_b[1], dh), Rq, dh, dh, s, "np.bwd.dec2.w"));
  MLHOT_TRY(lin_dgrad(sc.dd2, dh, b.d2, dh, ACT_RELU, wb1(p.dec_w[1], nullptr, dh), sc.dd1, dh, 0, Rq, dh, dh, s, "np.bwd.dec2.x"));
  MLHOT_TRY(lin_wgrad(sc.dd1, dh, b.d1, dh, ACT_RELU, b.dec_in, ldd, gb1(g.dec_w[0], g.dec_b[0], dh), Rq, ldd, dh, s, "np.bwd.dec0.w"));
  MLHOT_TRY(lin_dgrad(sc.dd1, dh, b.d1, dh, ACT_RELU, wb1(p.dec_w[0], nullptr, dh), sc.d_dec_in, ldd, 0, Rq, ldd, dh, s, "np.bwd.dec0.x"));

  if (d.Nc > 0) {
    const float* dz = sc.d_dec_in + dw;   // [Rq][dim_z], ld = ldd
    if (d.agg_mode == MLHOT_AGG_ATTENTION) {
      MLHOT_TRY(lin_wgrad(dz, ldd, nullptr, 0, ACT_NONE, b.rr, dw, gb1(g.r2z_w, g.r2z_b, d.dim_z), Rq, d.dim_r, d.dim_z, s, "np.bwd.r2z.w"));
      MLHOT_TRY(lin_dgrad(dz, ldd, nullptr, 0, ACT_NONE, wb1(p.r2z_w, nullptr, d.dim_z), sc.d_rr, dw, 0, Rq, d.dim_r, d.dim_z, s, "np.bwd.r2z.x"));
      MLHOT_TRY(lin_wgrad(sc.d_rr, dw, nullptr, 0, ACT_NONE, b.merged, H * dw, gb1(g.wo_w, g.wo_b, dw), Rq, H * dw, dw, s, "np.bwd.W.w"));
      MLHOT_TRY(lin_dgrad(sc.d_rr, dw, nullptr, 0, ACT_NONE, wb1(p.wo_w, nullptr, dw), sc.d_merged, H * dw, 0, Rq, H * dw, dw, s, "np.bwd.W.x"));
      FavorDims f{d.T, H, d.Nq, d.Nc, dw, d.m_feat};
      MLHOT_TRY(favor_bwd_any(f, b.qh, b.kh, b.vh, p.proj, b.merged, sc.d_merged, sc.dqh, sc.dkh, sc.dvh, b.favor, b.favor_bytes, s));
      // Q projection: x_qry gradient accumulates onto the decoder's
      MLHOT_TRY(lin_wgrad(sc.dqh, H * dw, nullptr, 0, ACT_NONE, b.dec_in, ldd, gb8(g.wq_w, g.wq_b, dw), Rq, dw, H * dw, s, "np.bwd.W_q.w"));
      MLHOT_TRY(lin_dgrad(sc.dqh, H * dw, nullptr, 0, ACT_NONE, wb8(p.wq_w, nullptr, dw), sc.d_dec_in, ldd, 1, Rq, dw, H * dw, s, "np.bwd.W_q.x"));
      MLHOT_TRY(lin_wgrad(sc.dvh, H * dw, nullptr, 0, ACT_NONE, b.rs, d.dim_r, gb8(g.wv_w, g.wv_b, dw), Rc, dw, H * dw, s, "np.bwd.W_v.w"));
      MLHOT_TRY(lin_dgrad(sc.dvh, H * dw, nullptr, 0, ACT_NONE, wb8(p.wv_w, nullptr, dw), sc.d_rs, d.dim_r, 0, Rc, dw, H * dw, s, "np.bwd.W_v.x"));
      MLHOT_TRY(lin_wgrad(sc.dkh, H * dw, nullptr, 0, ACT_NONE, b.cat_in, ldc, gb8(g.wk_w, g.wk_b, dw), Rc, dw, H * dw, s, "np.bwd.W_k.w"));
    } else {
      MLHOT_TRY(run_foreach(BcastRowsBwd{dz, ldd, d.dim_z, d.Nq, sc.dzt}, (size_t)d.T * d.dim_z, s, "np.bwd.bcast_z"));
      MLHOT_TRY(lin_wgrad(sc.dzt, d.dim_z, nullptr, 0, ACT_NONE, b.r, d.dim_r, gb1(g.r2z_w, g.r2z_b, d.dim_z), d.T, d.dim_r, d.dim_z, s, "np.bwd.r2z.w"));
      MLHOT_TRY(lin_dgrad(sc.dzt, d.dim_z, nullptr, 0, ACT_NONE, wb1(p.r2z_w, nullptr, d.dim_z), sc.dr, d.dim_r, 0, d.T, d.dim_r, d.dim_z, s, "np.bwd.r2z.x"));
      if (d.agg_mode == MLHOT_AGG_BACO) {
        MLHOT_TRY(run_foreach(AggBwd{d.agg_mode, d.Nc, d.dim_r, b.mu_l, b.lv, b.r, b.sigma, b.amax, sc.dr, sc.d_mu_l, sc.d_lv},
                              (size_t)d.T * d.dim_r, s, "np.bwd.agg"));
        MLHOT_TRY(lin_wgrad(sc.d_mu_l, d.dim_r, nullptr, 0, ACT_NONE, b.rs, d.dim_r, gb1(g.mu_w, g.mu_b, d.dim_r), Rc, d.dim_r, d.dim_r, s, "np.bwd.rs_to_mu.w"));
        MLHOT_TRY(lin_dgrad(sc.d_mu_l, d.dim_r, nullptr, 0, ACT_NONE, wb1(p.mu_w, nullptr, d.dim_r), sc.d_rs, d.dim_r, 0, Rc, d.dim_r, d.dim_r, s, "np.bwd.rs_to_mu.x"));
        MLHOT_TRY(lin_wgrad(sc.d_lv, d.dim_r, nullptr, 0, ACT_NONE, b.rs, d.dim_r, gb1(g.var_w, g.var_b, d.dim_r), Rc, d.dim_r, d.dim_r, s, "np.bwd.rs_to_var.w"));
        MLHOT_TRY(lin_dgrad(sc.d_lv, d.dim_r, nullptr, 0, ACT_NONE, wb1(p.var_w, nullptr, d.dim_r), sc.d_rs, d.dim_r, 1, Rc, d.dim_r, d.dim_r, s, "np.bwd.rs_to_var.x"));
      } else {
        MLHOT_TRY(run_foreach(AggBwd{d.agg_mode, d.Nc, d.dim_r, b.rs, nullptr, b.r, b.sigma, b.amax, sc.dr, sc.d_rs, nullptr},
                              (size_t)d.T * d.dim_r, s, "np.bwd.agg"));
      }
    }
    // EncoderFC, last layer first
    const float* dy = sc.d_rs; int lddy = d.dim_r, nout = d.dim_r, act = ACT_NONE; const float* yv = nullptr; int ldy = 0;
    for (int i = d.n_hidden; i >= 0; --i) {
      const float* x = i > 0 ? b.h[i - 1] : b.cat_in;
      const int kin = i > 0 ? d.hidden[i - 1] : ldc, ldx = kin;
      float* dx = i > 0 ? sc.dh[i - 1] : sc.d_cat_in;
      MLHOT_TRY(lin_wgrad(dy, lddy, yv, ldy, act, x, ldx, gb1(g.er_w[i], g.er_b[i], nout), Rc, kin, nout, s, "np.bwd.encoder_r.w"));
      MLHOT_TRY(lin_dgrad(dy, lddy, yv, ldy, act, wb1(p.er_w[i], nullptr, nout), dx, kin, 0, Rc, kin, nout, s, "np.bwd.encoder_r.x"));
      if (i > 0) { dy = dx; lddy = kin; nout = kin; act = ACT_RELU; yv = b.h[i - 1]; ldy = kin; }
    }
    if (d.agg_mode == MLHOT_AGG_ATTENTION)   // K projection: second consumer of x_ctx
      MLHOT_TRY(lin_dgrad(sc.dkh, H * dw, nullptr, 0, ACT_NONE, wb8(p.wk_w, nullptr, dw), sc.d_cat_in, ldc, 1, Rc, dw, H * dw, s, "np.bwd.W_k.x"));
    MLHOT_TRY(lin_wgrad(sc.d_cat_in + dw, ldc, nullptr, 0, ACT_NONE, ctx_y, d.label_dim, gb1(g.ty_w, g.ty_b, dw / 4), Rc, d.label_dim, dw / 4, s, "np.bwd.transform_y.w"));
  }
  MLHOT_TRY(enc_backward(ctx_x, Rc, qry_x, Rq, p.enc, dw, Rows2{sc.d_cat_in, ldc, Rc, sc.d_dec_in, ldd}, b.enc, g.enc, sc.enc, sc.enc_bytes, s));
  return MLHOT_OK;
}

}  // namespace mlhot
