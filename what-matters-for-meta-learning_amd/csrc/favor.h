// FAVOR+ (Performer) attention of the ANP models: forward and hand-derived backward.
//
// Reference arithmetic (networks/fast_attention.py:74-99, 151-156), per (task t, head h):
//   dd   = (c x) P^T,  c = d^-1/4                     [N, m]
//   diag = c^2/2 * |x|^2                               [N]
//   Q'   = m^-1/2 (exp(ddq - diagq - rowmax(ddq)) + 1e-4)
//   K'   = m^-1/2 (exp(ddk - diagk - MAX_over_whole_batch(ddk)) + 1e-4)
//   out  = (Q' (K'^T V)) / (Q' . sum_n K')
// The contraction is evaluated as S = Q' K'^T (Nq x Nc), out = S V / rowsum(S): identical
// algebra, and with Nc <= 30 << m it needs m/Nc times fewer FLOPs than the [m, d] context.
//
// Row layouts are token-major / head-minor: x[(t*N + n)*H + h][d], which is what the fused
// 8-head projection GEMM writes; out is written directly in the reference's merged order
// out[t][n][e*H + h] (ANPShapeNet1D.py:113-114).
#pragma once
#include "common.h"
#include "stab_xchg.h"
#include "foreach.h"
#include "igemm.h"
#include "problems.h"

namespace mlhot {

struct FavorDims {
  int T, H, Nq, Nc, d, m;
  size_t rows_q() const { return (size_t)T * Nq * H; }
  size_t rows_k() const { return (size_t)T * Nc * H; }
};

struct FavorWs {   // carved from the caller's workspace; forward fills, backward reuses
  float *pc, *qf, *kf, *diag_q, *diag_k, *max_q, *max_k, *gmax, *S, *D;
  int *arg_q, *arg_k, *gpos;
  float *dS, *wv, *Gq, *Gk, *rsum_q, *rsum_k, *gtotal;
  bool ok;
};

inline FavorWs favor_carve(const FavorDims& f, void* ws, size_t bytes, size_t* need = nullptr) {
  Arena a(ws, bytes);
  FavorWs w;
  const size_t rq = f.rows_q(), rk = f.rows_k(), sn = (size_t)f.T * f.H * f.Nq * f.Nc;
  w.pc = a.take<float>((size_t)f.m * f.d);
  w.qf = a.take<float>(rq * f.m);
  w.kf = a.take<float>(rk * f.m);
  w.diag_q = a.take<float>(rq); w.diag_k = a.take<float>(rk);
  w.max_q = a.take<float>(rq);  w.max_k = a.take<float>(rk);
  w.arg_q = a.take<int>(rq);    w.arg_k = a.take<int>(rk);
  w.gmax = a.take<float>(4);    w.gpos = a.take<int>(4);
  w.S = a.take<float>(sn);      w.D = a.take<float>((size_t)f.T * f.H * f.Nq);
  w.dS = a.take<float>(sn);     w.wv = a.take<float>((size_t)f.T * f.H * f.Nq);
  w.Gq = a.take<float>(rq * f.m);
  w.Gk = a.take<float>(rk * f.m);
  w.rsum_q = a.take<float>(rq); w.rsum_k = a.take<float>(rk);
  w.gtotal = a.take<float>(4);
  w.ok = a.ok;
  if (need) *need = a.off + 256;
  return w;
}

struct ScaleCopy { const float* s; float* d; float a; MLHOT_HD void operator()(size_t i) const { d[i] = a * s[i]; } };

// per row: diag = c^2/2 |x|^2, max_j dd[row][j] and its (first) arg-max
// Per-row statistics of the projected data, one workgroup per row (segmented reductions, fixed-order trees):
// diag = 0.5 c^2 |x|^2 (fast_attention.py:86-88) and the row maximum of data_dash with its FIRST position.
struct FavorRowDiag {
  typedef float T;
  const float* x; int d; float half_c2; float* diag;
  MLHOT_HD T identity() const { return 0.f; }
  MLHOT_HD T load(int row, int e) const { const float v = x[(size_t)row * d + e]; return v * v; }
  MLHOT_HD T combine(T a, T b) const { return a + b; }
  MLHOT_HD void finish(int row, T a) const { diag[row] = a * half_c2; }
};
struct ArgMaxPair { float v; int i; };
struct FavorRowMax {
  typedef ArgMaxPair T;
  const float* dd; int m; float* mx; int* arg;
  MLHOT_HD T identity() const { return T{-INFINITY, 0x7fffffff}; }
  MLHOT_HD T load(int row, int j) const { return T{dd[(size_t)row * m + j], j}; }
  MLHOT_HD T combine(T a, T b) const { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
  MLHOT_HD void finish(int row, T a) const { mx[row] = a.v; arg[row] = a.i; }
};
struct FavorGlobalMax {   // torch.max(data_dash) over every key row (fast_attention.py:97)
  typedef ArgMaxPair T;
  const float* mx; const int* arg; float* gmax; int* gpos;
  MLHOT_HD T identity() const { return T{-INFINITY, 0x7fffffff}; }
  MLHOT_HD T load(int i) const { return T{mx[i], i}; }
  MLHOT_HD T combine(T a, T b) const { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
  MLHOT_HD void finish(T a) const { gmax[0] = a.v; gpos[0] = a.i; gpos[1] = arg[a.i]; }
};

// dd -> E = ratio * exp(dd - diag - stab), in place.  The feature is F = E + ratio*eps; the
// buffers keep E and every consumer adds the constant, so the backward's exp-derivative
// (dF/d(arg) = E) is exact even where E underflows far below ratio*eps (no cancellation).
struct FavorFeat {
  float* f; const float* diag; const float* rowstab; const float* gstab; int m; float ratio;
  MLHOT_HD void operator()(size_t i) const {
    const size_t row = i / m;
    const float st = rowstab ? rowstab[row] : gstab[0];
    f[i] = ratio * expf(f[i] - diag[row] - st);
  }
};

// S[t][h][n][n'] = Q'[(t,n,h)] . K'[(t,n',h)]
struct FavorS {            // S[t,h,n,n'] = Q'[t,h,n,:] . K'[t,h,n',:]: one workgroup per entry, reduction over the m features
  typedef float T;
  FavorDims f; const float* qf; const float* kf; float re; float* S;
  MLHOT_HD T identity() const { return 0.f; }
  MLHOT_HD T load(int i, int j) const {
    const int np = i % f.Nc, n = (i / f.Nc) % f.Nq, h = (i / (f.Nc * f.Nq)) % f.H, t = i / (f.Nc * f.Nq * f.H);
    const float* a = qf + ((size_t)(t * f.Nq + n) * f.H + h) * f.m;
    const float* b = kf + ((size_t)(t * f.Nc + np) * f.H + h) * f.m;
    return (a[j] + re) * (b[j] + re);
  }
  MLHOT_HD T combine(T a, T b) const { return a + b; }
  MLHOT_HD void finish(int i, T a) const { S[i] = a; }
};
struct FavorD {
  int Nc; const float* S; float* D;
  MLHOT_HD void operator()(size_t i) const {
    float s = 0.f;
    for (int n = 0; n < Nc; ++n) s += S[i * Nc + n];
    D[i] = s;
  }
};
// out[t][n][e*H + h] = sum_n' S[t,h,n,n'] v[(t,n',h)][e] / D[t,h,n]
struct FavorOut {
  FavorDims f; const float* S; const float* D; const float* v; float* out;
  MLHOT_HD void operator()(size_t i) const {
    const int h = (int)(i % f.H), e = (int)((i / f.H) % f.d), n = (int)((i / ((size_t)f.H * f.d)) % f.Nq);
    const size_t t = i / ((size_t)f.H * f.d * f.Nq);
    const size_t sd = (t * f.H + h) * f.Nq + n;
    const float* s = S + sd * f.Nc;
    float acc = 0.f;
    for (int np = 0; np < f.Nc; ++np) acc = fmaf(s[np], v[((t * f.Nc + np) * f.H + h) * f.d + e], acc);
    out[i] = acc / D[sd];
  }
};

// ---- backward -------------------------------------------------------------------------------
// dS[t,h,n,n'] = dO[t,h,n,:] . (v[(t,n',h),:] - O[t,h,n,:]) / D[t,h,n]: the difference is formed per channel, BEFORE the sum - O[n] is
// a convex combination of the value rows, and dO . v - dO . O as two rounded sums loses the digits they share (favor2.h, B1)
struct FavorBwdDS {         // one workgroup per entry, reduction over the d channels
  typedef float T;
  FavorDims f; const float* dout; const float* v; const float* out; const float* D; float* dS;
  MLHOT_HD T identity() const { return 0.f; }
  MLHOT_HD T load(int i, int e) const {
    const int np = i % f.Nc, n = (i / f.Nc) % f.Nq, h = (i / (f.Nc * f.Nq)) % f.H, t = i / (f.Nc * f.Nq * f.H);
    const size_t ob = (size_t)(t * f.Nq + n) * ((size_t)f.d * f.H) + h;
    const float* vr = v + ((size_t)(t * f.Nc + np) * f.H + h) * f.d;
    return dout[ob + (size_t)e * f.H] * (vr[e] - out[ob + (size_t)e * f.H]);
  }
  MLHOT_HD T combine(T a, T b) const { return a + b; }
  MLHOT_HD void finish(int i, T a) const {
    const int n = (i / f.Nc) % f.Nq, h = (i / (f.Nc * f.Nq)) % f.H, t = i / (f.Nc * f.Nq * f.H);
    const size_t sd = (size_t)(t * f.H + h) * f.Nq + n;
    dS[i] = a / D[sd];
  }
};
// dv[(t,n',h)][e] = sum_n S[t,h,n,n'] dO[t,h,n,e] / D[t,h,n]
struct FavorBwdDV {
  FavorDims f; const float* S; const float* D; const float* dout; float* dv;
  MLHOT_HD void operator()(size_t i) const {
    const int e = (int)(i % f.d), h = (int)((i / f.d) % f.H), np = (int)((i / ((size_t)f.d * f.H)) % f.Nc);
    const size_t t = i / ((size_t)f.d * f.H * f.Nc);
    float acc = 0.f;
    for (int n = 0; n < f.Nq; ++n) {
      const size_t sd = (t * f.H + h) * f.Nq + n;
      acc = fmaf(S[sd * f.Nc + np] / D[sd], dout[(t * f.Nq + n) * ((size_t)f.d * f.H) + (size_t)e * f.H + h], acc);
    }
    dv[i] = acc;
  }
};
// G = dF (.) E with dQ' = dS K', dK' = dS^T Q'  (self_f / other_f hold E = F - ratio*eps)
struct FavorBwdG {
  FavorDims f; int is_query; const float* dS; const float* self_f; const float* other_f; float re; float* G;
  MLHOT_HD void operator()(size_t i) const {
    const int j = (int)(i % f.m); const size_t row = i / f.m;
    const int Nself = is_query ? f.Nq : f.Nc, Noth = is_query ? f.Nc : f.Nq;
    const int h = (int)(row % f.H), n = (int)((row / f.H) % Nself); const size_t t = row / ((size_t)f.H * Nself);
    float acc = 0.f;
    for (int o = 0; o < Noth; ++o) {
      const size_t si = is_query ? (((t * f.H + h) * f.Nq + n) * f.Nc + o) : (((t * f.H + h) * f.Nq + o) * f.Nc + n);
      acc = fmaf(dS[si], other_f[((t * Noth + o) * f.H + h) * f.m + j] + re, acc);
    }
    G[i] = acc * self_f[i];
  }
};
struct RowSum {            // one workgroup per row
  typedef float T;
  const float* G; int m; float* rs;
  MLHOT_HD T identity() const { return 0.f; }
  MLHOT_HD T load(int row, int j) const { return G[(size_t)row * m + j]; }
  MLHOT_HD T combine(T a, T b) const { return a + b; }
  MLHOT_HD void finish(int row, T a) const { rs[row] = a; }
};
struct SumRed {
  typedef float T;
  const float* v; float* out;
  MLHOT_HD float identity() const { return 0.f; }
  MLHOT_HD float load(int i) const { return v[i]; }
  MLHOT_HD float combine(float a, float b) const { return a + b; }
  MLHOT_HD void finish(float s) const { out[0] = s; }
};

#define MLHOT_TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

inline int favor_forward(const FavorDims& f, const float* q, const float* k, const float* v, const float* proj,
                         float* out, void* ws, size_t ws_bytes, hipStream_t s, const Stage& st = Stage{}) {
  FavorWs w = favor_carve(f, ws, ws_bytes);
  if (!w.ok) { set_error("favor_fwd: workspace too small"); return MLHOT_ERR_WORKSPACE; }
  const float c = powf((float)f.d, -0.25f), ratio = 1.0f / sqrtf((float)f.m), eps = 1e-4f;
  const size_t rq = f.rows_q(), rk = f.rows_k();
  if (st.first()) {
  MLHOT_TRY(run_foreach(ScaleCopy{proj, w.pc, c}, (size_t)f.m * f.d, s, "favor.scale_proj"));
  WBlocks pb{}; pb.w[0] = w.pc; pb.b[0] = nullptr; pb.rows = f.m;
  LinearFwd lq{(int)rq, f.m, f.d, q, f.d, pb, w.qf, f.m, ACT_NONE};
  LinearFwd lk{(int)rk, f.m, f.d, k, f.d, pb, w.kf, f.m, ACT_NONE};
  MLHOT_TRY(run_igemm_auto(lq, s, "favor.ddq"));
  MLHOT_TRY(run_igemm_auto(lk, s, "favor.ddk"));
  MLHOT_TRY(run_reduce_seg(FavorRowDiag{q, f.d, 0.5f * c * c, w.diag_q}, (int)rq, f.d, s, "favor.rowdiag_q"));
  MLHOT_TRY(run_reduce_seg(FavorRowDiag{k, f.d, 0.5f * c * c, w.diag_k}, (int)rk, f.d, s, "favor.rowdiag_k"));
  MLHOT_TRY(run_reduce_seg(FavorRowMax{w.qf, f.m, w.max_q, w.arg_q}, (int)rq, f.m, s, "favor.rowmax_q"));
  MLHOT_TRY(run_reduce_seg(FavorRowMax{w.kf, f.m, w.max_k, w.arg_k}, (int)rk, f.m, s, "favor.rowmax_k"));
  MLHOT_TRY(run_reduce1(FavorGlobalMax{w.max_k, w.arg_k, w.gmax, w.gpos}, (int)rk, s, "favor.gmax"));
  }
#ifndef MLHOT_HOSTSIM
  // strict sharded parity (stab_xchg.h): the rank's maximum goes out, the batch's comes back (and the position with it)
  if (st.stage == 0) return sx::max_publish(w.gmax, 1, st.x, s);
  if (st.stage == 1) MLHOT_TRY(sx::max_apply(w.gmax, w.gpos, 1, st.x, s));
#endif
  MLHOT_TRY(run_foreach(FavorFeat{w.qf, w.diag_q, w.max_q, nullptr, f.m, ratio}, rq * f.m, s, "favor.feat_q"));
  MLHOT_TRY(run_foreach(FavorFeat{w.kf, w.diag_k, nullptr, w.gmax, f.m, ratio}, rk * f.m, s, "favor.feat_k"));
  MLHOT_TRY(run_reduce_seg(FavorS{f, w.qf, w.kf, ratio * eps, w.S}, f.T * f.H * f.Nq * f.Nc, f.m, s, "favor.S"));
  MLHOT_TRY(run_foreach(FavorD{f.Nc, w.S, w.D}, (size_t)f.T * f.H * f.Nq, s, "favor.D"));
  MLHOT_TRY(run_foreach(FavorOut{f, w.S, w.D, v, out}, (size_t)f.T * f.Nq * f.d * f.H, s, "favor.out"));
  return MLHOT_OK;
}

inline int favor_backward(const FavorDims& f, const float* q, const float* k, const float* v, const float* out,
                          const float* dout, float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, hipStream_t s,
                          const Stage& st = Stage{}) {
  FavorWs w = favor_carve(f, ws, ws_bytes);
  if (!w.ok) { set_error("favor_bwd: workspace too small"); return MLHOT_ERR_WORKSPACE; }
  const float c = powf((float)f.d, -0.25f), ratio = 1.0f / sqrtf((float)f.m), eps = 1e-4f;
  const size_t rq = f.rows_q(), rk = f.rows_k(), thn = (size_t)f.T * f.H * f.Nq;
  if (st.first()) {
  MLHOT_TRY(run_reduce_seg(FavorBwdDS{f, dout, v, out, w.D, w.dS}, (int)(thn * f.Nc), f.d, s, "favor.bwd.dS"));
  MLHOT_TRY(run_foreach(FavorBwdDV{f, w.S, w.D, dout, dv}, rk * f.d, s, "favor.bwd.dv"));
  MLHOT_TRY(run_foreach(FavorBwdG{f, 1, w.dS, w.qf, w.kf, ratio * eps, w.Gq}, rq * f.m, s, "favor.bwd.Gq"));
  MLHOT_TRY(run_foreach(FavorBwdG{f, 0, w.dS, w.kf, w.qf, ratio * eps, w.Gk}, rk * f.m, s, "favor.bwd.Gk"));
  MLHOT_TRY(run_reduce_seg(RowSum{w.Gq, f.m, w.rsum_q}, (int)rq, f.m, s, "favor.bwd.rsum_q"));
  MLHOT_TRY(run_reduce_seg(RowSum{w.Gk, f.m, w.rsum_k}, (int)rk, f.m, s, "favor.bwd.rsum_k"));
  MLHOT_TRY(run_reduce1(SumRed{w.rsum_k, w.gtotal}, (int)rk, s, "favor.bwd.gtotal"));
  }
#ifndef MLHOT_HOSTSIM
  if (st.stage == 0) return sx::sum_publish(w.gtotal, 1, st.x, s);
  if (st.stage == 1) MLHOT_TRY(sx::sum_apply(w.gtotal, 1, st.x, w.gtotal, 1, s));
#endif
  FavorDx xq{(int)rq, f.d, f.m, w.Gq, w.rsum_q, w.arg_q, nullptr, nullptr, w.pc, q, c * c, dq};
  FavorDx xk{(int)rk, f.d, f.m, w.Gk, w.rsum_k, nullptr, w.gpos, w.gtotal, w.pc, k, c * c, dk};
  MLHOT_TRY(run_igemm_auto(xq, s, "favor.bwd.dq"));
  MLHOT_TRY(run_igemm_auto(xk, s, "favor.bwd.dk"));
  return MLHOT_OK;
}

}  // namespace mlhot
