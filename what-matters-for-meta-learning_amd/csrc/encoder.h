// E1: the vanilla image encoder (conv-conv-pool-conv-linear), forward and backward.
// Activations stay NCHW fp32 in HBM; what the backward needs is kept in `saved`:
//   a1  [n][32][64][64]  conv1 output (post-ReLU)          512 KiB / image
//   p2  [n][48][16][16]  pooled conv2 output (post-ReLU)    48 KiB / image
//   am2 [n][48][16][16]  pool arg-max (uint8)               12 KiB / image
//   a3  [n][4096]        conv3 output (post-ReLU, C-major flatten = nn.Flatten order)
// conv2's 192 KiB/image pre-pool map is scratch: ReLU and pool commute, so the pooled value
// and the arg-max are all the backward needs (DyPooled in problems.h).
#pragma once
#include "common.h"
#include "foreach.h"
#include "igemm.h"
#include "ops_direct.h"
#include "problems.h"
#include "favor.h"   // MLHOT_TRY
#include "conv_tc.h"
#include "conv_split.h"
#include "conv3_tc.h"
#include "enc_linear.h"
#include "../../include/mlhot.h"

namespace mlhot {

// Run-time switches (mlhot_set_option): which implementation of a hot-path row runs.  The
// generic igemm problems are always available as the A/B reference of the specialised kernels.
struct Options { int conv2_tc; int tail_fused; int materialize_a1; int dbg; int tail_spec; int conv2_split; int conv3_bwd_merged; };
extern Options g_opt;
constexpr int C2_GRID = 256;   // one persistent workgroup per CU

struct EncSaved {
  float* a1; float* p2; uint8_t* am2; float* a3; unsigned* m1; bool ok; size_t bytes;
};
inline EncSaved enc_saved_carve(int n, void* base, size_t cap) {
  Arena a(base, cap);
  EncSaved s;
  s.a1 = a.take<float>((size_t)n * 32 * 64 * 64);
  s.p2 = a.take<float>((size_t)n * 48 * 16 * 16);
  s.am2 = a.take<uint8_t>((size_t)n * 48 * 16 * 16);
  s.a3 = a.take<float>((size_t)n * 4096);
  s.m1 = a.take<unsigned>((size_t)n * 64 * 64 + 16);     // conv1 ReLU bits (16-dword records per 16 columns + one junk record, conv_tc.h m1_record), written by the fused forward
  s.ok = a.ok; s.bytes = a.off + 256;
  return s;
}
inline size_t enc_saved_bytes(int n) { return enc_saved_carve(n, nullptr, 0).bytes; }

// split-K factors (bounded so the slabs stay a few MB and every launch has >= ~256 workgroups)
inline int enc_lin_split(int n) { (void)n; return 32; }
inline int enc_linw_split(int n) { return n >= 64 ? 4 : 1; }
inline int conv3w_split(int n) { int s = n / 4; return s < 1 ? 1 : (s > 128 ? 128 : s); }
inline int conv2w_split(int n) { int s = n / 2; return s < 1 ? 1 : (s > 240 ? 240 : s); }
inline int conv1w_split(int n) { int s = n * 2; return s > 1024 ? 1024 : s; }

struct EncScratch {
  float* a2;      // fwd: [n][48][32][32]
  float* slab;    // split-K partials (fwd linear, bwd wgrads)
  float* dy3;     // bwd: [n][4096]
  float* dp2;     // bwd: [n][48][16][16]
  float* dy1;     // bwd: [n][32][64][64]
  bool ok; size_t bytes;
};
inline size_t enc_slab_floats(int n, int dim_w) {
  size_t m = (size_t)enc_lin_split(n) * n * dim_w;
  size_t v;
  v = (size_t)enc_linw_split(n) * dim_w * 4097; if (v > m) m = v;
  v = (size_t)conv3w_split(n) * 64 * 433;       if (v > m) m = v;
  v = (size_t)conv2w_split(n) * 48 * 289;       if (v > m) m = v;
  v = (size_t)conv1w_split(n) * 32 * 10;        if (v > m) m = v;
  // weight-stationary backward: the conv3, conv2 and conv1 partial slabs are all alive until the single deferred reduce
  v = (size_t)C2_GRID * (64 * 433 + 48 * 289 + 320) + (size_t)2 * C2_GRID * 48 * 288;     if (v > m) m = v;
  return m;
}
inline EncScratch enc_scratch_carve(int n, int dim_w, void* base, size_t cap) {
  Arena a(base, cap);
  EncScratch s;
  s.slab = a.take<float>(enc_slab_floats(n, dim_w));
  // forward and backward never run concurrently on one scratch: a2 aliases the backward buffers
  const size_t mark = a.off;
  s.a2 = a.take<float>((size_t)n * 48 * 32 * 32);
  const size_t fwd_end = a.off;
  a.off = mark;
  s.dy3 = a.take<float>((size_t)n * 4096);
  s.dp2 = a.take<float>((size_t)n * 48 * 16 * 16);
  s.dy1 = a.take<float>((size_t)n * 32 * 64 * 64);
  if (fwd_end > a.off) a.off = fwd_end;
  s.ok = a.ok; s.bytes = a.off + 256;
  return s;
}
inline size_t enc_scratch_bytes(int n, int dim_w) { return enc_scratch_carve(n, dim_w, nullptr, 0).bytes; }

#ifndef MLHOT_HOSTSIM
// conv1 + ReLU + conv2 + ReLU + 2x2 max-pool in one kernel (a1 is recomputed band by band in LDS and never stored): p2, the pool
// arg-max and conv1's ReLU sign bits land in `sv`.  Option conv2_split: bit 1 forward, bit 2 data gradient, bit 4 weight gradient
// run conv2 on the bf16 pipe over split operands (conv_split.h) - same inputs, same outputs' layout.
inline int conv12_forward(const c2::ImgSrc& xs, int n, const float* w1, const float* b1, const float* w2, const float* b2,
                          const EncSaved& sv, hipStream_t s) {
  if (g_opt.conv2_split & 1) {
    const int grid2 = n * 16 < C2_GRID ? n * 16 : C2_GRID;
    ProfScope ps("enc.conv12.split", s);
    hipLaunchKernelGGL(c2s::conv12_fwd_split_kernel, dim3(grid2), dim3(c2s::NT), 0, s, xs, w1, b1, w2, b2, sv.p2, sv.am2, sv.m1, n);
  } else {
    const int grid = n * 8 < C2_GRID ? n * 8 : C2_GRID;
    ProfScope ps("enc.conv12", s);
    hipLaunchKernelGGL(c2::conv12_fwd_pool_kernel, dim3(grid), dim3(c2::NT), 0, s, xs, w1, b1, w2, b2, sv.p2, sv.am2, sv.m1, n, g_opt.dbg);
  }
  return check_launch("enc.conv12");
}
// the block's two backward kernels: slab_w / slab_b rows of R2 = 48 * 288 + 48 floats per workgroup ([dW2 in accumulator order | db2]),
// slab_1 rows of 320 ([dW1 | db1]); the caller folds the `grid` rows
constexpr int C12_L2 = 48 * 288, C12_R2 = C12_L2 + 48;
inline int conv12_grid(int n) { return n * 8 < C2_GRID ? n * 8 : C2_GRID; }
template <class Between>          // between(): called behind the weight-gradient launch (the caller's slab fold beside the next kernel)
inline int conv12_backward(const c2::ImgSrc& xs, int n, const float* w1, const float* b1, const float* w2, const float* dp2,
                           const EncSaved& sv, float* slab_w, float* slab_b, float* slab_1, hipStream_t s, Between&& between) {
  const int grid = conv12_grid(n);
  if (g_opt.conv2_split & 4) {
    ProfScope ps("enc.bwd.conv12.wgrad.split", s);          // the same `grid` slab rows as the fp32 kernel (the caller folds that many)
    hipLaunchKernelGGL(c2s::conv12_wgrad_split_kernel, dim3(grid), dim3(c2s::NT), 0, s, xs, w1, b1, dp2, sv.p2, sv.am2, slab_w, slab_b, n);
  } else {
    ProfScope ps("enc.bwd.conv12.wgrad", s);
    hipLaunchKernelGGL(c2::conv12_wgrad_kernel, dim3(grid), dim3(c2::NT), 0, s, xs, w1, b1, dp2, sv.p2, sv.am2, slab_w, slab_b, n, g_opt.dbg);
  }
  MLHOT_TRY(check_launch("enc.bwd.conv12.wgrad"));
  MLHOT_TRY(between());
  if (g_opt.conv2_split & 2) {
    ProfScope ps("enc.bwd.conv12.dgrad.split", s);
    hipLaunchKernelGGL(c2s::conv12_dgrad_split_kernel, dim3(grid), dim3(c2s::dg::NT2), 0, s, xs, sv.m1, dp2, sv.p2, sv.am2, w2, slab_1, n);
  } else {
    ProfScope ps("enc.bwd.conv12.dgrad", s);
    hipLaunchKernelGGL(c2::conv12_dgrad_kernel, dim3(grid), dim3(c2::NT2), 0, s, xs, sv.m1, dp2, sv.p2, sv.am2, w2, slab_1, n);
  }
  return check_launch("enc.bwd.conv12.dgrad");
}
#endif

// `xfold`: the caller's first kernel folds the Linear's split-K partial results itself (the fused attention tail's phase A reads the
// feature tiles anyway: one launch and its ~4.7 us off the forward's critical path).  On return xfold->slab / bias / k / n describe
// the partial results [k][n][dim_w] - or slab == nullptr when this path wrote `feat` itself (generic Linear, other widths).
struct EncXFold { const float* slab; const float* bias; int k, n; };
inline int enc_forward(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params& p, int dim_w,
                       Rows2 feat, void* saved, void* scratch, size_t scratch_bytes, hipStream_t s, EncXFold* xfold = nullptr) {
  if (xfold != nullptr) *xfold = EncXFold{nullptr, nullptr, 0, 0};
  const int n = n0 + n1;
  if (n <= 0) return MLHOT_OK;
  EncSaved sv = enc_saved_carve(n, saved, (size_t)-1 / 2);
  EncScratch sc = enc_scratch_carve(n, dim_w, scratch, scratch_bytes);
  if (!sc.ok) { set_error("enc_vanilla_fwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes); return MLHOT_ERR_WORKSPACE; }
  const Src2 x{img0, n0, img1, (size_t)128 * 128};
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc) {
    // conv1 + conv2 + pool in one kernel: a1 is recomputed band by band in LDS and never stored
    if (g_opt.materialize_a1) MLHOT_TRY(run_foreach(Conv1Fwd<Src2>{x, p.w1, p.b1, sv.a1}, (size_t)n * 4096, s, "enc.conv1.debug"));
    MLHOT_TRY(conv12_forward(c2::ImgSrc{img0, n0, img1}, n, p.w1, p.b1, p.w2, p.b2, sv, s));
  } else
#endif
  {
    MLHOT_TRY(run_foreach(Conv1Fwd<Src2>{x, p.w1, p.b1, sv.a1}, (size_t)n * 4096, s, "enc.conv1"));
    typedef ConvFwd<32, 64, 64, 48, Src1> C2;
    C2 c2{n * 1024, 48, 288, Src1{sv.a1, (size_t)32 * 4096}, p.w2, p.b2, sc.a2};
    MLHOT_TRY((run_igemm<C2, 128, 48, 16, 4, 1>(c2, 1, nullptr, s, "enc.conv2")));
    MLHOT_TRY(run_foreach(Pool2Fwd{sc.a2, sv.p2, sv.am2, 32, 32}, (size_t)n * 48 * 256, s, "enc.pool"));
  }
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc) {
    const int grid = n * 2 < C2_GRID ? n * 2 : C2_GRID;
    {
      ProfScope ps("enc.conv3", s);
      hipLaunchKernelGGL(c3::conv3_fwd_kernel, dim3(grid), dim3(c3::F_NT), 0, s, sv.p2, p.w3, p.b3, sv.a3, n);
    }
    MLHOT_TRY(check_launch("enc.conv3"));
  } else
#endif
  {
    typedef ConvFwd<48, 16, 16, 64, Src1> C3;
    C3 c3{n * 64, 64, 432, Src1{sv.p2, (size_t)48 * 256}, p.w3, p.b3, sv.a3};
    MLHOT_TRY((run_igemm<C3, 64, 64, 16, 2, 2>(c3, 1, nullptr, s, "enc.conv3")));
  }
  EncLinFwd lf{n, dim_w, 4096, sv.a3, p.wl, p.bl, feat};
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc && dim_w == el::DW) {
    {
      ProfScope ps("enc.linear", s);
      hipLaunchKernelGGL(el::enc_linear_fwd_kernel, dim3(((n + 15) / 16) * el::F_KS), dim3(256), 0, s, sv.a3, p.wl, sc.slab, n);
    }
    MLHOT_TRY(check_launch("enc.linear"));
    if (xfold != nullptr) { *xfold = EncXFold{sc.slab, p.bl, el::F_KS, n}; return MLHOT_OK; }
    {
      ProfScope ps("slab_reduce", s);
      hipLaunchKernelGGL(el::enc_linear_fold_kernel, dim3((n * el::DW + 255) / 256), dim3(256), 0, s, sc.slab, p.bl, feat, n);
    }
    return check_launch("enc.linear.fold");
  }
#endif
  MLHOT_TRY((run_igemm<EncLinFwd, 64, 64, 16, 2, 2>(lf, enc_lin_split(n), sc.slab, s, "enc.linear")));
  return MLHOT_OK;
}

template <int PY, int PX>
inline int enc_conv3_dgrad(int n, const float* dy3, const float* w3, float* dp2, hipStream_t s) {
  typedef DyPlain<64, 8, 8> DY;
  typedef ConvDgrad<48, 16, 16, 64, PY, PX, DY> P;
  P p{n * 64, 48, P::NTY * P::NTX * 64, DY{dy3}, w3, nullptr, dp2};
  return run_igemm<P, 64, 48, 16, 4, 1>(p, 1, nullptr, s, "enc.bwd.conv3.dgrad");
}
template <int PY, int PX>
inline int enc_conv2_dgrad(int n, const DyPooled<48, 32, 32>& dy, const float* w2, const float* a1, float* dy1, hipStream_t s) {
  typedef ConvDgrad<32, 64, 64, 48, PY, PX, DyPooled<48, 32, 32>> P;
  P p{n * 1024, 32, P::NTY * P::NTX * 48, dy, w2, a1, dy1};
  return run_igemm<P, 128, 32, 16, 4, 1>(p, 1, nullptr, s, "enc.bwd.conv2.dgrad");
}

// A slab sum that has not been launched yet: out[e] = sum over parts p < nparts of slab[p * stride + e], e < len.
struct PendingSum { const float* slab; float* out; int nparts, len, stride; };

// `extra`: a slab sum the caller has pending (the fused tail's per-task gradient slabs); the weight-stationary path folds it
// into its own final reduce launch, every other path sums it first.
inline int enc_backward(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params& p, int dim_w,
                        Rows2 dfeat, const void* saved, const mlhot_enc_grads& g,
                        void* scratch, size_t scratch_bytes, hipStream_t s, const PendingSum* extra = nullptr) {
  const int n = n0 + n1;
#ifndef MLHOT_HOSTSIM
  // deferred slab sums of the weight-stationary path: ONE launch at the very end (4 kernels fewer on the step's critical path)
  c2::SumPartsMulti mp{};
  auto pend = [&](const float* slab, float* out, int nparts, int len, int stride, int kind = 0) {
    mp.seg[mp.n] = c2::SumParts{slab, out, nparts, len, stride, kind};
    mp.first[mp.n + 1] = mp.first[mp.n] + c2::sum_parts_blocks(len);
    ++mp.n;
  };
  auto flush_on = [&](hipStream_t st, const char* what) -> int {
    if (mp.n == 0) return MLHOT_OK;
    {
      ProfScope ps(what, st);
      hipLaunchKernelGGL(c2::sum_parts_multi_kernel, dim3(mp.first[mp.n]), dim3(256), 0, st, mp);
    }
    mp.n = 0;
    return check_launch(what);
  };
  auto flush = [&]() -> int { return flush_on(s, "slab_reduce"); };
  const bool defer = g_opt.conv2_tc && n > 0;
  // The folds do not sit on the step's critical path: with a side lane (common.h) each one is issued right behind its producer
  // and runs BESIDE the kernels that follow (the tail's slabs under the Linear backward, conv3's 28 MB under conv3's data gradient
  // and the conv12 weight gradient, conv2's 14 MB under the conv12 data gradient); the lane is joined at the end.  Fixed fold
  // order either way (bitwise the same result).  Without a lane: ONE deferred launch at the very end, as before.
  SideLane* lane = defer ? side_lane(s) : nullptr;
  bool forked = false;
  auto fold_aside = [&]() -> int {            // everything pended so far, on the lane, behind what `s` holds now
    if (lane == nullptr) return MLHOT_OK;     // stays pended for the final flush
    if (!lane->fork()) { set_error("enc_vanilla_bwd: side lane fork failed"); return MLHOT_ERR_LAUNCH; }
    forked = true;
    return flush_on(lane->side, "slab_reduce.side");
  };
  if (extra != nullptr && extra->slab != nullptr) {
    pend(extra->slab, extra->out, extra->nparts, extra->len, extra->stride);
    if (!defer) MLHOT_TRY(flush());
    else MLHOT_TRY(fold_aside());
  }
#else
  (void)extra;
#endif
  if (n <= 0) return MLHOT_OK;
  EncSaved sv = enc_saved_carve(n, (void*)saved, (size_t)-1 / 2);
  EncScratch sc = enc_scratch_carve(n, dim_w, scratch, scratch_bytes);
  if (!sc.ok) {
#ifndef MLHOT_HOSTSIM
    if (forked) (void)lane->join();
#endif
    set_error("enc_vanilla_bwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes);
    return MLHOT_ERR_WORKSPACE;
  }
  const Src2 x{img0, n0, img1, (size_t)128 * 128};

  // Linear(4096 -> dim_w): input gradient (masked by conv3's ReLU), weight + bias gradient
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc && dim_w == el::DW) {
    {
      ProfScope ps("enc.bwd.linear", s);
      hipLaunchKernelGGL(el::enc_linear_bwd_kernel, dim3(el::KIN / 16), dim3(el::NTH), 0, s, dfeat, p.wl, sv.a3, sc.dy3, g.wl, g.bl, n);
    }
    MLHOT_TRY(check_launch("enc.bwd.linear"));
  } else
#endif
  {
    EncLinDgrad ld{n, 4096, dim_w, dfeat, p.wl, sv.a3, sc.dy3};
    MLHOT_TRY((run_igemm<EncLinDgrad, 64, 64, 16, 2, 2>(ld, 1, nullptr, s, "enc.bwd.linear.dgrad")));
    EncLinWgrad lw{dim_w, 4097, n, dfeat, sv.a3, g.wl, g.bl};
    MLHOT_TRY((run_igemm<EncLinWgrad, 64, 64, 16, 2, 2>(lw, enc_linw_split(n), sc.slab, s, "enc.bwd.linear.wgrad")));
  }

  // conv3
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc) {
    const int grid = n < C2_GRID ? n : C2_GRID;
    // one slab row per workgroup = [dW (64*432) | db (64)]: when the caller's gradient tensors are adjacent in that order
    // (mlhot_np_grads_flat_layout) the weights and the bias reduce in ONE launch
    constexpr int L3 = 64 * 432, R3 = L3 + 64;
    float* slab_w = sc.slab;                       // conv3's slab region: alive until the deferred reduce
    float* slab_b = sc.slab + L3;
    if (g_opt.conv3_bwd_merged && grid == C2_GRID) {
      // weight and data gradient in ONE launch: the first nw workgroups the weight gradient, the rest the data gradient (conv3_tc.h)
      const int nw = g_opt.conv3_bwd_merged > 1 && g_opt.conv3_bwd_merged < C2_GRID ? g_opt.conv3_bwd_merged : C2_GRID / 2;
      {
        ProfScope ps("enc.bwd.conv3", s);
        hipLaunchKernelGGL(c3::conv3_bwd_kernel, dim3(C2_GRID), dim3(c3::W_NT), 0, s, sv.p2, p.w3, sc.dy3, slab_w, slab_b, sc.dp2, n, nw);
      }
      MLHOT_TRY(check_launch("enc.bwd.conv3"));
      pend(slab_w, g.w3, nw, L3, R3, 2);
      pend(slab_b, g.b3, nw, 64, R3);
      MLHOT_TRY(fold_aside());
    } else {
    {
      ProfScope ps("enc.bwd.conv3.wgrad", s);
      hipLaunchKernelGGL(c3::conv3_wgrad_kernel, dim3(grid), dim3(c3::W_NT), 0, s, sv.p2, sc.dy3, slab_w, slab_b, n);
    }
    MLHOT_TRY(check_launch("enc.bwd.conv3.wgrad"));
    // weights in accumulator order (coalesced stores in the kernel), un-permuted by the fold
    pend(slab_w, g.w3, grid, L3, R3, 2);
    pend(slab_b, g.b3, grid, 64, R3);
    MLHOT_TRY(fold_aside());
    {
      ProfScope ps("enc.bwd.conv3.dgrad", s);
      hipLaunchKernelGGL(c3::conv3_dgrad_kernel, dim3(grid), dim3(c3::D_NT), 0, s, p.w3, sc.dy3, sc.dp2, n);
    }
    MLHOT_TRY(check_launch("enc.bwd.conv3.dgrad"));
    }
  } else
#endif
  {
    typedef ConvWgrad<48, 16, 16, 64, DyPlain<64, 8, 8>, Src1> W3;
    W3 w3{64, 433, n * 64, DyPlain<64, 8, 8>{sc.dy3}, Src1{sv.p2, (size_t)48 * 256}, g.w3, g.b3};
    MLHOT_TRY((run_igemm<W3, 64, 64, 16, 2, 2>(w3, conv3w_split(n), sc.slab, s, "enc.bwd.conv3.wgrad")));
    MLHOT_TRY((enc_conv3_dgrad<0, 0>(n, sc.dy3, p.w3, sc.dp2, s)));
    MLHOT_TRY((enc_conv3_dgrad<0, 1>(n, sc.dy3, p.w3, sc.dp2, s)));
    MLHOT_TRY((enc_conv3_dgrad<1, 0>(n, sc.dy3, p.w3, sc.dp2, s)));
    MLHOT_TRY((enc_conv3_dgrad<1, 1>(n, sc.dy3, p.w3, sc.dp2, s)));
  }

  // conv2 (pool + ReLU backward are folded into the dY gather)
  const DyPooled<48, 32, 32> dy2{sc.dp2, sv.p2, sv.am2};
#ifndef MLHOT_HOSTSIM
  if (g_opt.conv2_tc) {
    const int grid = conv12_grid(n);
    constexpr int L2 = C12_L2, R2 = C12_R2;             // slab row = [dW2 | db2], see conv3 above
    float* slab_w = sc.slab + (size_t)C2_GRID * (64 * 433);      // behind conv3's region
    float* slab_b = slab_w + L2;
    float* slab_1 = slab_w + (size_t)C2_GRID * R2;
    MLHOT_TRY(conv12_backward(c2::ImgSrc{img0, n0, img1}, n, p.w1, p.b1, p.w2, sc.dp2, sv, slab_w, slab_b, slab_1, s, [&]() -> int {
      // the weights sit in the slab in accumulator order (coalesced stores in the kernel); the fold un-permutes them
      pend(slab_w, g.w2, grid, L2, R2, 1);
      pend(slab_b, g.b2, grid, 48, R2);
      return fold_aside();
    }));
    if (g.b1 == g.w1 + 288 && (reinterpret_cast<uintptr_t>(g.w1) & 15) == 0) {
      pend(slab_1, g.w1, grid, 320, 320);           // conv1's gradients came out of the dgrad kernel
    } else {
      ProfScope ps("slab_reduce", s);
      hipLaunchKernelGGL(c2::conv1_grads_kernel, dim3(16), dim3(320), 0, s, slab_1, grid, g.w1, g.b1);
    }
    MLHOT_TRY(check_launch("enc.bwd.conv1.grads"));
    MLHOT_TRY(flush());                        // conv1's 320 x grid slab (and everything, when there is no lane)
    if (forked && !lane->join()) { set_error("enc_vanilla_bwd: side lane join failed"); return MLHOT_ERR_LAUNCH; }
    return MLHOT_OK;
  } else
#endif
  {
    typedef ConvWgrad<32, 64, 64, 48, DyPooled<48, 32, 32>, Src1> W2;
    W2 w2{48, 289, n * 1024, dy2, Src1{sv.a1, (size_t)32 * 4096}, g.w2, g.b2};
    MLHOT_TRY((run_igemm<W2, 48, 64, 16, 1, 4>(w2, conv2w_split(n), sc.slab, s, "enc.bwd.conv2.wgrad")));
    MLHOT_TRY((enc_conv2_dgrad<0, 0>(n, dy2, p.w2, sv.a1, sc.dy1, s)));
    MLHOT_TRY((enc_conv2_dgrad<0, 1>(n, dy2, p.w2, sv.a1, sc.dy1, s)));
    MLHOT_TRY((enc_conv2_dgrad<1, 0>(n, dy2, p.w2, sv.a1, sc.dy1, s)));
    MLHOT_TRY((enc_conv2_dgrad<1, 1>(n, dy2, p.w2, sv.a1, sc.dy1, s)));
  }

  // conv1 (no input gradient: images are leaves)
  typedef ConvWgrad<1, 128, 128, 32, DyPlain<32, 64, 64>, Src2> W1;
  W1 w1{32, 10, n * 4096, DyPlain<32, 64, 64>{sc.dy1}, x, g.w1, g.b1};
  MLHOT_TRY((run_igemm<W1, 32, 16, 16, 2, 1>(w1, conv1w_split(n), sc.slab, s, "enc.bwd.conv1.wgrad")));
  return MLHOT_OK;
}

}  // namespace mlhot
