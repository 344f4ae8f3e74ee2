// libmlhot.so - C ABI (include/mlhot.h) over the gfx950 kernels.  Single translation unit:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC mlhot.hip -o libmlhot.so
#include <thread>
#include <stdarg.h>
#include <mutex>

#include "common.h"
#include "foreach.h"
#include "igemm.h"
#include "problems.h"
#include "ops_direct.h"
#include "favor.h"
#include "favor2.h"
#include "encoder.h"
#include "np_vanilla.h"
#include "conv_rt.h"
#include "ingest.h"
#include "bbb_multi.h"
#include "mt_normal.h"
#include "nt_xent.h"
#include "mlp_chain.h"
#include "resnet_trunk.h"
#include "../../include/mlhot.h"

namespace mlhot {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

Options g_opt = {1, 1, 0, 0, 7935, 0, 1};
int g_favor2 = 1;
#ifndef MLHOT_HOSTSIM
namespace rt { int g_trunk_fuse34 = 1, g_trunk_dual_dgrad = 1, g_trunk_wg_rows = 128; }
#endif


#ifndef MLHOT_HOSTSIM
// ---- side lanes (common.h) ---------------------------------------------------------------------------
int g_side_fold = 0;   // measured on c3: the folds beside the persistent one-workgroup-per-CU kernels cost +65 us per step (a fold workgroup and a conv workgroup do not fit one CU together, so the conv kernel waits for the fold), DESIGN.md section 4 round 4
static std::mutex g_lane_mu;
static SideLane g_lanes[8];
static int g_nlanes = 0;
SideLane* side_lane(hipStream_t main) {
  if (!g_side_fold) return nullptr;
  std::lock_guard<std::mutex> lk(g_lane_mu);
  for (int i = 0; i < g_nlanes; ++i) if (g_lanes[i].main == main) return &g_lanes[i];
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(main, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;   // no object creation inside a capture
  if (g_nlanes == 8) return nullptr;
  SideLane l{};
  l.main = main;
  if (hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
  if (hipEventCreateWithFlags(&l.ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&l.ev_join, hipEventDisableTiming) != hipSuccess) return nullptr;
  g_lanes[g_nlanes] = l;
  return &g_lanes[g_nlanes++];
}

// ---- per-launch event profiler -------------------------------------------------------------------
bool g_prof_on = false;
struct ProfRec { const char* what; hipEvent_t a, b; bool open; };
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;
// A failed event call drops that record (the profile then misses a launch; the launch itself is unaffected).
void prof_record(const char* what, hipStream_t s, bool begin) {
  if (g_prof_n >= g_prof_cap) return;
  if (begin) {
    g_prof[g_prof_n].what = what;
    g_prof[g_prof_n].open = hipEventRecord(g_prof[g_prof_n].a, s) == hipSuccess;
  } else if (g_prof[g_prof_n].open && hipEventRecord(g_prof[g_prof_n].b, s) == hipSuccess) {
    ++g_prof_n;
  }
}
#endif
}  // namespace mlhot

using namespace mlhot;

extern "C" {

int mlhot_version(void) { return MLHOT_ABI_VERSION; }
const char* mlhot_last_error(void) { return g_err; }

// ---- run-time options -------------------------------------------------------------------------
int mlhot_set_option(const char* name, int value) {
  if (!strcmp(name, "conv2_tc")) { g_opt.conv2_tc = value; return MLHOT_OK; }
  if (!strcmp(name, "conv2_split")) { g_opt.conv2_split = value; return MLHOT_OK; }
  if (!strcmp(name, "tail_fused")) { g_opt.tail_fused = value; return MLHOT_OK; }
  if (!strcmp(name, "tail_spec")) { g_opt.tail_spec = value; return MLHOT_OK; }     // fused tail: bit mask of the phases that run the kernels specialised for the shipped dimensions (csrc/tail_spec.h; bits 1..32 = the six phases, 64 = phase A also folds the encoder Linear's split-K partial results, 128 = phase C' takes the loss's gradient itself when handed a loss descriptor, 512 = phase B' as two workgroups per (task, head): query side | key / value side, 1024 / 2048 = phases C' / A' as several workgroups per task sharing the weight-gradient tiles (two; four with 4096); default 7935 = all) instead of the run-time-shaped ones
  if (!strcmp(name, "conv3_bwd_merged")) { g_opt.conv3_bwd_merged = value; return MLHOT_OK; }     // 0: two launches; 1: one launch, 128 + 128 workgroups; n > 1: n weight-gradient workgroups of 256
  if (!strcmp(name, "materialize_a1")) { g_opt.materialize_a1 = value; return MLHOT_OK; }
  if (!strcmp(name, "dbg")) { g_opt.dbg = value; return MLHOT_OK; }   // timing experiments only (results become wrong)
  if (!strcmp(name, "favor2")) { g_favor2 = value; return MLHOT_OK; }
#ifndef MLHOT_HOSTSIM
  if (!strcmp(name, "trunk_dual_dgrad")) { rt::g_trunk_dual_dgrad = value; return MLHOT_OK; }   // 1 (default): a 3x3-skip block's two stride-2 data gradients in one 512-thread launch (resnet_ws.h dgrad2_dual_kernel)
  if (!strcmp(name, "trunk_wg_rows")) { if (value < 16 || value > 128) { set_error("trunk_wg_rows: 16 .. 128"); return MLHOT_ERR_ARG; } rt::g_trunk_wg_rows = value; return MLHOT_OK; }   // slab rows (x 4 channel tiles = workgroups) a trunk weight-gradient launch is planned against
  if (!strcmp(name, "trunk_fuse34")) { rt::g_trunk_fuse34 = value; return MLHOT_OK; }   // 1 (default): blocks 3-4 of a 64 x 64 trunk as one launch per direction (resnet_ws.h tail34_*)
  if (!strcmp(name, "side_fold")) { g_side_fold = value; return MLHOT_OK; }   // 1 (default 0): slab folds of the encoder backward on the library's helper stream (common.h SideLane)
#endif  // FAVOR+: the two-launch kernels (csrc/favor2.h, default) or favor.h's chain
  set_error("mlhot_set_option: unknown option %s", name);
  return MLHOT_ERR_ARG;
}

// ---- profiler (bench only) ----------------------------------------------------------------------
int mlhot_prof_begin(int max_records) {
#ifndef MLHOT_HOSTSIM
  if (g_prof || max_records <= 0) { set_error("prof_begin: already recording, or bad capacity"); return MLHOT_ERR_ARG; }
  g_prof = new ProfRec[max_records];
  int made = 0;
  for (; made < max_records; ++made) {
    g_prof[made].open = false;
    if (hipEventCreate(&g_prof[made].a) != hipSuccess) break;
    if (hipEventCreate(&g_prof[made].b) != hipSuccess) { (void)hipEventDestroy(g_prof[made].a); break; }
  }
  g_prof_cap = made; g_prof_n = 0; g_prof_on = true;
#else
  (void)max_records;
#endif
  return MLHOT_OK;
}
// Stops recording, synchronises the events and returns the number of records; record i is
// (label, milliseconds).  Labels are static strings owned by the library.
int mlhot_prof_end(const char** labels, float* ms, int cap) {
  int n = 0;
#ifndef MLHOT_HOSTSIM
  g_prof_on = false;
  for (int i = 0; i < g_prof_n; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof[i].b) != hipSuccess || hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
    if (n < cap) { labels[n] = g_prof[i].what; ms[n] = t; ++n; }
  }
  for (int i = 0; i < g_prof_cap; ++i) { (void)hipEventDestroy(g_prof[i].a); (void)hipEventDestroy(g_prof[i].b); }
  delete[] g_prof; g_prof = nullptr; g_prof_cap = g_prof_n = 0;
#else
  (void)labels; (void)ms; (void)cap;
#endif
  return n;
}

// ---- E1 ---------------------------------------------------------------------------------------
size_t mlhot_enc_vanilla_saved_bytes(int n_img) { return enc_saved_bytes(n_img); }
size_t mlhot_enc_vanilla_scratch_bytes(int n_img, int dim_w) { return enc_scratch_bytes(n_img, dim_w); }

int mlhot_enc_vanilla_fwd(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params* p, int dim_w,
                          float* feat0, int ld0, float* feat1, int ld1, void* saved, void* scratch, size_t scratch_bytes,
                          void* stream) {
  if (!p || n0 < 0 || n1 < 0 || dim_w <= 0) { set_error("enc_vanilla_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return enc_forward(img0, n0, img1, n1, *p, dim_w, Rows2{feat0, ld0, n0, feat1, ld1}, saved, scratch, scratch_bytes,
                     (hipStream_t)stream);
}

int mlhot_enc_vanilla_bwd(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params* p, int dim_w,
                          const float* dfeat0, int ldd0, const float* dfeat1, int ldd1, const void* saved,
                          const mlhot_enc_grads* g, void* scratch, size_t scratch_bytes, void* stream) {
  if (!p || !g || n0 < 0 || n1 < 0 || dim_w <= 0) { set_error("enc_vanilla_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return enc_backward(img0, n0, img1, n1, *p, dim_w, Rows2{(float*)dfeat0, ldd0, n0, (float*)dfeat1, ldd1}, saved, *g,
                      scratch, scratch_bytes, (hipStream_t)stream);
}

// ---- E1's first block on its own: conv1 + ReLU + conv2 + ReLU + 2x2 max-pool (tests and micro-benchmarks of the kernels that
// carry 80 % of the vanilla models' FLOPs; `saved` has the layout of mlhot_enc_vanilla_saved_bytes) -------------------------------
size_t mlhot_conv12_scratch_bytes(int n_img) {
#ifndef MLHOT_HOSTSIM
  return ((size_t)conv12_grid(n_img < 1 ? 1 : n_img) * (C12_R2 + 320)) * sizeof(float) + 256;
#else
  (void)n_img; return 256;
#endif
}
int mlhot_conv12_fwd(const float* img, int n_img, const float* w1, const float* b1, const float* w2, const float* b2, void* saved,
                     void* stream) {
  if (n_img < 0 || !w1 || !b1 || !w2 || !b2 || !saved) { set_error("conv12_fwd: bad argument"); return MLHOT_ERR_ARG; }
  if (n_img == 0) return MLHOT_OK;
#ifndef MLHOT_HOSTSIM
  return conv12_forward(c2::ImgSrc{img, n_img, nullptr}, n_img, w1, b1, w2, b2, enc_saved_carve(n_img, saved, (size_t)-1 / 2),
                        (hipStream_t)stream);
#else
  (void)img; (void)stream; set_error("conv12_fwd: GPU build only"); return MLHOT_ERR_ARG;
#endif
}
int mlhot_conv12_bwd(const float* img, int n_img, const float* w1, const float* b1, const float* w2, const float* dp2,
                     const void* saved, float* dw1, float* db1, float* dw2, float* db2, void* scratch, size_t scratch_bytes,
                     void* stream) {
  if (n_img <= 0 || !w1 || !b1 || !w2 || !dp2 || !saved || !dw1 || !db1 || !dw2 || !db2) { set_error("conv12_bwd: bad argument"); return MLHOT_ERR_ARG; }
  if (scratch_bytes < mlhot_conv12_scratch_bytes(n_img)) { set_error("conv12_bwd: scratch too small"); return MLHOT_ERR_WORKSPACE; }
#ifndef MLHOT_HOSTSIM
  hipStream_t s = (hipStream_t)stream;
  const int grid = conv12_grid(n_img);
  float* slab_w = reinterpret_cast<float*>(scratch);
  float* slab_1 = slab_w + (size_t)grid * C12_R2;
  MLHOT_TRY(conv12_backward(c2::ImgSrc{img, n_img, nullptr}, n_img, w1, b1, w2, dp2, enc_saved_carve(n_img, (void*)saved, (size_t)-1 / 2),
                            slab_w, slab_w + C12_L2, slab_1, s, []() -> int { return MLHOT_OK; }));
  c2::SumPartsMulti mp{};
  mp.seg[0] = c2::SumParts{slab_w, dw2, grid, C12_L2, C12_R2, 1};
  mp.seg[1] = c2::SumParts{slab_w + C12_L2, db2, grid, 48, C12_R2, 0};
  mp.first[1] = c2::sum_parts_blocks(C12_L2);
  mp.first[2] = mp.first[1] + c2::sum_parts_blocks(48);
  mp.n = 2;
  hipLaunchKernelGGL(c2::sum_parts_multi_kernel, dim3(mp.first[2]), dim3(256), 0, s, mp);
  hipLaunchKernelGGL(c2::conv1_grads_kernel, dim3(16), dim3(320), 0, s, slab_1, grid, dw1, db1);
  return check_launch("conv12_bwd");
#else
  (void)img; (void)stream; set_error("conv12_bwd: GPU build only"); return MLHOT_ERR_ARG;
#endif
}

// ---- linear -----------------------------------------------------------------------------------
size_t mlhot_linear_bwd_scratch_bytes(int M, int K, int N) { (void)M; (void)K; (void)N; return 256; }

int mlhot_linear_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K, int N,
                     int act, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || act < 0 || act > 2) { set_error("linear_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return lin_fwd(x, ldx, wb1(w, b, N), y, ldy, M, K, N, act, (hipStream_t)stream, "linear_fwd");
}

int mlhot_linear_bwd(const float* x, int ldx, const float* w, const float* y, int ldy, const float* dy, int lddy, int M,
                     int K, int N, int act, float* dx, int lddx, int accumulate, float* dw, float* db, void* scratch,
                     size_t scratch_bytes, void* stream) {
  (void)scratch; (void)scratch_bytes;
  if (M < 0 || K <= 0 || N <= 0 || act < 0 || act > 2) { set_error("linear_bwd: bad argument"); return MLHOT_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
#ifndef MLHOT_HOSTSIM
  // few-row layers: both gradients in one launch (linear_skinny.h; the conditions are lin_dgrad's and lin_wgrad's)
  if (dw && dx && M <= sk::MAX_ROWS && N % 4 == 0 && sk::aligned4(dy, lddy) && (act == ACT_NONE || sk::aligned4(y, ldy)))
    return sk::run_bwd(dy, lddy, y, ldy, act, w, x, ldx, dx, lddx, accumulate, dw, db, M, K, N, s, "linear_bwd");
#endif
  if (dw) MLHOT_TRY(lin_wgrad(dy, lddy, y, ldy, act, x, ldx, gb1(dw, db, N), M, K, N, s, "linear_bwd.w"));
  if (dx) MLHOT_TRY(lin_dgrad(dy, lddy, y, ldy, act, wb1(w, nullptr, N), dx, lddx, accumulate, M, K, N, s, "linear_bwd.x"));
  return MLHOT_OK;
}

// ---- chains of few-row linears, independent few-row linears in one launch (csrc/mlp_chain.h) ---------------------
int mlhot_mlp_chain_fwd(const float* x0, int ldx0, int M, const mlhot_chain_layer* layers, int n_layers, void* stream) {
#ifndef MLHOT_HOSTSIM
  return mc::chain_forward(x0, ldx0, M, layers, n_layers, (hipStream_t)stream);
#else
  (void)x0; (void)ldx0; (void)M; (void)layers; (void)n_layers; (void)stream; set_error("mlp_chain: GPU build only"); return MLHOT_ERR_ARG;
#endif
}
int mlhot_mlp_chain_bwd(const float* x0, int ldx0, int M, const mlhot_chain_layer* layers, const mlhot_chain_grads* grads, int n_layers,
                        const float* dy, int lddy, float* dx0, int lddx0, int dx0_accumulate, void* stream) {
#ifndef MLHOT_HOSTSIM
  return mc::chain_backward(x0, ldx0, M, layers, grads, n_layers, dy, lddy, dx0, lddx0, dx0_accumulate, (hipStream_t)stream);
#else
  (void)x0; (void)ldx0; (void)M; (void)layers; (void)grads; (void)n_layers; (void)dy; (void)lddy; (void)dx0; (void)lddx0; (void)dx0_accumulate; (void)stream;
  set_error("mlp_chain: GPU build only"); return MLHOT_ERR_ARG;
#endif
}
int mlhot_linear_multi_fwd(const mlhot_linear_job* jobs, int n_jobs, void* stream) {
#ifndef MLHOT_HOSTSIM
  return mc::multi_forward(jobs, n_jobs, (hipStream_t)stream);
#else
  (void)jobs; (void)n_jobs; (void)stream; set_error("linear_multi: GPU build only"); return MLHOT_ERR_ARG;
#endif
}
int mlhot_linear_multi_bwd(const mlhot_linear_job* jobs, int n_jobs, void* stream) {
#ifndef MLHOT_HOSTSIM
  return mc::multi_backward(jobs, n_jobs, (hipStream_t)stream);
#else
  (void)jobs; (void)n_jobs; (void)stream; set_error("linear_multi: GPU build only"); return MLHOT_ERR_ARG;
#endif
}

// ---- aggregators ------------------------------------------------------------------------------
int mlhot_agg_fwd(int mode, const float* rs, const float* lv, int T, int Nc, int R, float* r, float* sigma_z,
                  int32_t* amax, void* stream) {
  if (mode < 0 || mode > 2 || T <= 0 || Nc <= 0 || R <= 0) { set_error("agg_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_foreach(AggFwd{mode, Nc, R, rs, lv, r, sigma_z, amax}, (size_t)T * R, (hipStream_t)stream, "agg_fwd");
}
int mlhot_agg_bwd(int mode, const float* rs, const float* lv, const float* r, const float* sigma_z, const int32_t* amax,
                  const float* dr, int T, int Nc, int R, float* drs, float* dlv, void* stream) {
  if (mode < 0 || mode > 2 || T <= 0 || Nc <= 0 || R <= 0) { set_error("agg_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_foreach(AggBwd{mode, Nc, R, rs, lv, r, sigma_z, amax, dr, drs, dlv}, (size_t)T * R, (hipStream_t)stream, "agg_bwd");
}

// ---- FAVOR+ -----------------------------------------------------------------------------------
size_t mlhot_favor_ws_bytes(int T, int H, int Nq, int Nc, int d, int m) {
  return favor_ws_need(FavorDims{T, H, Nq, Nc, d, m});
}
int mlhot_favor_fwd(const float* q, const float* k, const float* v, const float* proj, int T, int H, int Nq, int Nc,
                    int d, int m, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (T <= 0 || H <= 0 || Nq <= 0 || Nc <= 0 || d <= 0 || m <= 0) { set_error("favor_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return favor_fwd_any(FavorDims{T, H, Nq, Nc, d, m}, q, k, v, proj, out, ws, ws_bytes, (hipStream_t)stream);
}
int mlhot_favor_bwd(const float* q, const float* k, const float* v, const float* proj, int T, int H, int Nq, int Nc,
                    int d, int m, const float* out, const float* dout, float* dq, float* dk, float* dv, void* ws,
                    size_t ws_bytes, void* stream) {
  if (T <= 0 || H <= 0 || Nq <= 0 || Nc <= 0 || d <= 0 || m <= 0) { set_error("favor_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return favor_bwd_any(FavorDims{T, H, Nq, Nc, d, m}, q, k, v, proj, out, dout, dq, dk, dv, ws, ws_bytes, (hipStream_t)stream);
}
// Staged passes (strict sharded parity of the key stabiliser, csrc/stab_xchg.h): stage 0 runs up to the rank-local scalar and
// publishes it in xchg, stage 1 takes the batch-wide scalar from xchg and runs the rest.
static int staged_args(int stage, float* xchg, const char* what) {
  if ((stage != 0 && stage != 1) || !xchg) { set_error("%s: stage must be 0 or 1 and xchg a device block of 4 floats", what); return MLHOT_ERR_ARG; }
#ifdef MLHOT_HOSTSIM
  set_error("%s: GPU build only", what);
  return MLHOT_ERR_UNSUPPORTED;
#else
  return MLHOT_OK;
#endif
}
int mlhot_favor_fwd_staged(const float* q, const float* k, const float* v, const float* proj, int T, int H, int Nq, int Nc,
                           int d, int m, float* out, void* ws, size_t ws_bytes, int stage, float* xchg, void* stream) {
  if (T <= 0 || H <= 0 || Nq <= 0 || Nc <= 0 || d <= 0 || m <= 0) { set_error("favor_fwd_staged: bad argument"); return MLHOT_ERR_ARG; }
  MLHOT_TRY(staged_args(stage, xchg, "favor_fwd_staged"));
  return favor_fwd_any(FavorDims{T, H, Nq, Nc, d, m}, q, k, v, proj, out, ws, ws_bytes, (hipStream_t)stream, Stage{stage, xchg});
}
int mlhot_favor_bwd_staged(const float* q, const float* k, const float* v, const float* proj, int T, int H, int Nq, int Nc,
                           int d, int m, const float* out, const float* dout, float* dq, float* dk, float* dv, void* ws,
                           size_t ws_bytes, int stage, float* xchg, void* stream) {
  if (T <= 0 || H <= 0 || Nq <= 0 || Nc <= 0 || d <= 0 || m <= 0) { set_error("favor_bwd_staged: bad argument"); return MLHOT_ERR_ARG; }
  MLHOT_TRY(staged_args(stage, xchg, "favor_bwd_staged"));
  return favor_bwd_any(FavorDims{T, H, Nq, Nc, d, m}, q, k, v, proj, out, dout, dq, dk, dv, ws, ws_bytes, (hipStream_t)stream, Stage{stage, xchg});
}

// ---- losses -----------------------------------------------------------------------------------
int mlhot_loss_fwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, float* loss, void* stream) {
  if (kind < 0 || kind > 4 || rows <= 0 || y_dim <= 0 || y_dim > 8 || gt_dim < 1) { set_error("loss_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_reduce1(LossRed{kind, y_dim, gt_dim, rows, mu, gt, loss}, rows, (hipStream_t)stream, "loss_fwd");
}
int mlhot_loss_bwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, const float* dloss,
                   float* dmu, void* stream) {
  if (kind < 0 || kind > 4 || rows <= 0 || y_dim <= 0 || y_dim > 8 || gt_dim < 1) { set_error("loss_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_foreach(LossBwd{kind, y_dim, gt_dim, rows, mu, gt, dloss, dmu, nullptr}, (size_t)rows, (hipStream_t)stream, "loss_bwd");
}

int mlhot_loss_plus_fwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, const float* x, float alpha,
                        float* loss, float* total, void* stream) {
  if (kind < 0 || kind > 4 || rows <= 0 || y_dim <= 0 || y_dim > 8 || gt_dim < 1 || !x || !total) { set_error("loss_plus_fwd: bad argument"); return MLHOT_ERR_ARG; }
  LossPlusRed r;
  static_cast<LossRed&>(r) = LossRed{kind, y_dim, gt_dim, rows, mu, gt, loss};
  r.x = x; r.alpha = alpha; r.total = total;
  return run_reduce1(r, rows, (hipStream_t)stream, "loss_fwd");
}
int mlhot_loss_plus_bwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, const float* dtotal, float alpha,
                        float* dmu, float* dx, void* stream) {
  if (kind < 0 || kind > 4 || rows <= 0 || y_dim <= 0 || y_dim > 8 || gt_dim < 1 || !dtotal || !dmu) { set_error("loss_plus_bwd: bad argument"); return MLHOT_ERR_ARG; }
  LossPlusBwd f;
  static_cast<LossBwd&>(f) = LossBwd{kind, y_dim, gt_dim, rows, mu, gt, dtotal, dmu, nullptr};
  f.alpha = alpha; f.dx = dx;
  return run_foreach(f, (size_t)rows, (hipStream_t)stream, "loss_bwd");
}

// ---- E2 / D2 building blocks: run-time-shaped conv, residual join, 2x2 max-pool; B1: BBB sample -------
size_t mlhot_conv2d_bwd_scratch_bytes(int N, int Cin, int H, int W, int Cout, int k, int stride, int pad) {
  return conv_bwd_scratch_bytes(conv_shape(N, Cin, H, W, Cout, k, stride, pad));
}
int mlhot_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout, int k,
                     int stride, int pad, int relu, void* stream) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0 || H + 2 * pad < k || W + 2 * pad < k) {
    set_error("conv2d_fwd: bad argument"); return MLHOT_ERR_ARG;
  }
  return conv_rt_forward(conv_shape(N, Cin, H, W, Cout, k, stride, pad), x, w, b, y, relu, (hipStream_t)stream);
}
int mlhot_conv2d_bwd(const float* x, const float* w, const float* y, const float* dy, int N, int Cin, int H, int W, int Cout, int k,
                     int stride, int pad, int relu, float* dx, float* dw, float* db, void* scratch, size_t scratch_bytes, void* stream) {
  if (N <= 0 || Cin <= 0 || Cout <= 0 || k <= 0 || stride <= 0 || pad < 0) { set_error("conv2d_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return conv_rt_backward(conv_shape(N, Cin, H, W, Cout, k, stride, pad), x, w, relu ? y : nullptr, dy, dx, dw, db, scratch, scratch_bytes,
                          (hipStream_t)stream);
}
int mlhot_add_relu_fwd(const float* a, const float* b, float* y, size_t n, void* stream) {
  return run_foreach(AddRelu{a, b, y}, n, (hipStream_t)stream, "add_relu.fwd");
}
int mlhot_add_relu_bwd(const float* y, const float* dy, float* g, size_t n, void* stream) {
  return run_foreach(AddReluBwd{y, dy, g}, n, (hipStream_t)stream, "add_relu.bwd");
}
int mlhot_axpy(const float* a, const float* x, float alpha, float* y, size_t n, void* stream) {
  if (!x || !y) { set_error("axpy: null argument"); return MLHOT_ERR_ARG; }
  return run_foreach(Axpy{a, x, alpha, y}, n, (hipStream_t)stream, "axpy");
}
int mlhot_pool2_fwd(const float* x, float* y, uint8_t* amax, int planes, int H, int W, void* stream) {
  if (planes <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) { set_error("pool2_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_foreach(Pool2Fwd{x, y, amax, H, W}, (size_t)planes * (H / 2) * (W / 2), (hipStream_t)stream, "pool2.fwd");
}
int mlhot_pool2_bwd(const float* dy, const uint8_t* amax, float* dx, int planes, int H, int W, void* stream) {
  if (planes <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) { set_error("pool2_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return run_foreach(Pool2Bwd{dy, amax, dx, H, W}, (size_t)planes * H * W, (hipStream_t)stream, "pool2.bwd");
}
int mlhot_bbb_sample_fwd(const float* mu, const float* rho, const float* eps, float* w, float* klterm, float* kl, size_t n, void* stream) {
  MLHOT_TRY(run_foreach(BbbSample{mu, rho, eps, w, klterm}, n, (hipStream_t)stream, "bbb.sample"));
  return run_reduce1(SumRed{klterm, kl}, (int)n, (hipStream_t)stream, "bbb.kl");
}
int mlhot_bbb_sample_bwd(const float* mu, const float* rho, const float* eps, const float* dw, const float* dkl, float* dmu, float* drho,
                         size_t n, void* stream) {
  return run_foreach(BbbSampleBwd{mu, rho, eps, dw, dkl, dmu, drho}, n, (hipStream_t)stream, "bbb.sample.bwd");
}

size_t mlhot_bbb_sample_multi_scratch_floats(const mlhot_bbb_item* items, int n_items) {
  size_t blocks = 0;
  for (int i = 0; i < n_items; ++i) blocks += (items[i].n + BBB_CHUNK - 1) / BBB_CHUNK;
  return blocks > 0 ? blocks : 1;
}
int mlhot_bbb_sample_multi_fwd(const mlhot_bbb_item* items, int n_items, float* partial, float* kl, void* stream) {
  if (!items || !partial || !kl) { set_error("bbb_sample_multi_fwd: bad argument"); return MLHOT_ERR_ARG; }
  return bbb_sample_multi_fwd(items, n_items, partial, kl, (hipStream_t)stream);
}
int mlhot_bbb_sample_multi_bwd(const mlhot_bbb_item* items, int n_items, const float* dkl, void* stream) {
  if (!items || !dkl) { set_error("bbb_sample_multi_bwd: bad argument"); return MLHOT_ERR_ARG; }
  return bbb_sample_multi_bwd(items, n_items, dkl, (hipStream_t)stream);
}

// ---- NT-Xent (csrc/nt_xent.h) -------------------------------------------------------------------------------------------
size_t mlhot_nt_xent_ws_floats(int N) {
  const size_t n16 = (size_t)((N + 15) & ~15);
  return 4 * n16 + n16 / 16 + 16;
}
static int ntx_args(const float* z, int N, int d, int div, int mod, float t, float* ws, ntx::Args& a) {
  if (!z || !ws || N < 1 || N > ntx::MAXN || d < 16 || d > ntx::MAXD || d % 16 || div < 1 || mod < 1 || !(t > 0.f)) {
    set_error("nt_xent: needs 1 <= N <= %d rows, d <= %d with d %% 16 == 0, div, mod >= 1, t > 0", ntx::MAXN, ntx::MAXD);
    return MLHOT_ERR_ARG;
  }
  const int n16 = (N + 15) & ~15;
  a = ntx::Args{z, N, d, div, mod, 1.0f / t, ws, ws + n16, ws + 2 * n16, ws + 3 * n16, ws + 4 * n16};
  return MLHOT_OK;
}
int mlhot_nt_xent_fwd(const float* z, int N, int d, int div, int mod, float t, float* ws, float* loss, void* stream) {
  ntx::Args a;
  MLHOT_TRY(ntx_args(z, N, d, div, mod, t, ws, a));
  if (!loss) { set_error("nt_xent_fwd: null loss"); return MLHOT_ERR_ARG; }
#ifndef MLHOT_HOSTSIM
  hipStream_t s = (hipStream_t)stream;
  const long long pairs = ntx::pair_count(N, div, mod);
  const int nblk = (N + 15) / 16;
  MLHOT_TRY(tail_launch(ntx::ntx_fwd_kernel, nblk, 256, ntx::lds_bytes(N, d), a, s, "ntxent.fwd"));
  {
    ProfScope ps("ntxent.finish", s);
    hipLaunchKernelGGL(ntx::ntx_finish_kernel, dim3(1), dim3(64), 0, s, a.partial, nblk, pairs > 0 ? 1.0f / (float)pairs : 0.f, loss);
  }
  return check_launch("nt_xent_fwd (finish)");
#else
  (void)stream; set_error("nt_xent: GPU build only"); return MLHOT_ERR_ARG;
#endif
}
int mlhot_nt_xent_bwd(const float* z, int N, int d, int div, int mod, float t, const float* ws, const float* dloss, float* dz, void* stream) {
  ntx::Args a;
  MLHOT_TRY(ntx_args(z, N, d, div, mod, t, const_cast<float*>(ws), a));
  if (!dloss || !dz) { set_error("nt_xent_bwd: null argument"); return MLHOT_ERR_ARG; }
#ifndef MLHOT_HOSTSIM
  hipStream_t s = (hipStream_t)stream;
  const long long pairs = ntx::pair_count(N, div, mod);
  ntx::BwdArgs b{a, dloss, pairs > 0 ? 1.0f / (float)pairs : 0.f, dz};
  return tail_launch(ntx::ntx_bwd_kernel, (N + 15) / 16, 256, ntx::lds_bytes(N, d), b, s, "ntxent.bwd");
#else
  (void)stream; set_error("nt_xent: GPU build only"); return MLHOT_ERR_ARG;
#endif
}

// ---- host only: fp32 images that are k / div back to bytes, every element checked (csrc/ingest.h) -------------------------
int mlhot_host_f32_to_u8_exact(const float* src, uint8_t* dst, int64_t n, float div, int threads, int64_t* n_inexact) {
  if (!src || !dst || n < 0 || !(div > 0.f) || !n_inexact || threads < 1 || threads > 64) { set_error("host_f32_to_u8_exact: bad argument"); return MLHOT_ERR_ARG; }
  if (threads > 1 && n < (int64_t)threads * 65536) threads = (int)(n / 65536 > 0 ? n / 65536 : 1);      // a piece below 64 K elements is not worth a thread
  if (threads == 1) { *n_inexact = (int64_t)ingest::host_f32_to_u8_exact(src, dst, (long)n, div); return MLHOT_OK; }
  // native threads, started per call (~10 us each, the first ones already converting while the last start): a Python pool's submit /
  // result round trips cost 20 - 40 us per piece under the GIL, as much as the piece's work (measured on the GPU box, round 6)
  long bad[64] = {};
  std::thread th[64];
  const long per = ((long)n / threads + 63) / 64 * 64;
  auto piece = [&](int i) {
    const long lo = (long)i * per, hi = i == threads - 1 ? (long)n : (lo + per < (long)n ? lo + per : (long)n);
    bad[i] = lo < hi ? ingest::host_f32_to_u8_exact(src + lo, dst + lo, hi - lo, div) : 0;
  };
  for (int i = 1; i < threads; ++i) th[i] = std::thread(piece, i);
  piece(0);
  long total = bad[0];
  for (int i = 1; i < threads; ++i) { th[i].join(); total += bad[i]; }
  *n_inexact = (int64_t)total;
  return MLHOT_OK;
}

// ---- torch's CPU normal_() stream on the device (csrc/mt_normal.h) ----------------------------------------------------
int mlhot_mt19937_advance(uint32_t* engine, uint64_t n_outputs) {       // host only: no stream, no device
  if (!engine || (int)engine[mt::N] < 0 || (int)engine[mt::N] > mt::N || engine[mt::N + 1] > (uint32_t)mt::N) { set_error("mt19937_advance: bad engine"); return MLHOT_ERR_ARG; }
  mt::host_advance(engine, n_outputs);
  return MLHOT_OK;
}
size_t mlhot_mt19937_jump_ws_words(int n_sub) {
  return n_sub < 1 ? 0 : (size_t)33 * 624 + 8 + (size_t)(n_sub - 1) * 8 * 624;
}
int mlhot_mt19937_normal_par(uint32_t* engine, float* uniform_ws, float* out, const int64_t* segs, int nseg, int64_t total_outputs,
                             int64_t total_groups, const uint32_t* polys, int n_sub, int stride_blocks, uint32_t* jump_ws, void* stream) {
  if (!engine || !uniform_ws || !out || !segs || nseg <= 0 || total_outputs <= 0 || total_groups <= 0 || !polys || n_sub < 2 || n_sub > 1024 ||
      stride_blocks < 1 || !jump_ws) {
    set_error("mt19937_normal_par: bad argument");
    return MLHOT_ERR_ARG;
  }
  if ((int64_t)n_sub * stride_blocks * 624 < total_outputs) {      // every output of new blocks must belong to a sub-stream (conservative: ignores the current block's rest)
    set_error("mt19937_normal_par: %d sub-streams of %d blocks do not cover %lld outputs", n_sub, stride_blocks, (long long)total_outputs);
    return MLHOT_ERR_ARG;
  }
#ifndef MLHOT_HOSTSIM
  hipStream_t s = (hipStream_t)stream;
  uint32_t* x = jump_ws;
  uint32_t* partial = jump_ws + (size_t)mt::JWIN + 8;      // 8 words behind the window: the engine's (left, next) as the draw found them
  {
    ProfScope ps("eps.mt19937.window", s);
    hipLaunchKernelGGL(mt::mt_window_kernel, dim3(1), dim3(256), 0, s, engine, x);
  }
  MLHOT_TRY(check_launch("mt19937_normal_par (window)"));
  {
    ProfScope ps("eps.mt19937.jump", s);
    hipLaunchKernelGGL(mt::mt_jump_kernel, dim3(n_sub - 1, mt::JPARTS), dim3(256), 0, s, x, polys, partial);
  }
  MLHOT_TRY(check_launch("mt19937_normal_par (jump)"));
  {
    ProfScope ps("eps.mt19937.chunks", s);
    hipLaunchKernelGGL(mt::mt_chunk_kernel, dim3(n_sub), dim3(256), 0, s, engine, x, partial, uniform_ws, (long long)total_outputs, stride_blocks);
  }
  MLHOT_TRY(check_launch("mt19937_normal_par (chunks)"));
  {
    ProfScope ps("eps.box_muller", s);
    const long long pairs = (long long)total_groups * 8;
    hipLaunchKernelGGL(mt::mt_box_muller_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, uniform_ws, out,
                       reinterpret_cast<const mt::Seg*>(segs), nseg, (long long)total_groups);
  }
  return check_launch("mt19937_normal_par (transform)");
#else
  (void)stream; set_error("mt19937_normal_par: GPU build only"); return MLHOT_ERR_ARG;
#endif
}

int mlhot_mt19937_normal(uint32_t* engine, float* uniform_ws, float* out, const int64_t* segs, int nseg, int64_t total_outputs,
                         int64_t total_groups, void* stream) {
  if (!engine || !uniform_ws || !out || !segs || nseg <= 0 || total_outputs <= 0 || total_groups <= 0) {
    set_error("mt19937_normal: bad argument");
    return MLHOT_ERR_ARG;
  }
#ifndef MLHOT_HOSTSIM
  static_assert(sizeof(mt::Seg) == 4 * sizeof(int64_t), "segment record = 4 x int64");
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps("eps.mt19937", s);
    hipLaunchKernelGGL(mt::mt_fill_kernel, dim3(1), dim3(256), 0, s, engine, uniform_ws, (long long)total_outputs);
  }
  MLHOT_TRY(check_launch("mt19937_normal (fill)"));
  {
    ProfScope ps("eps.box_muller", s);
    const long long pairs = (long long)total_groups * 8;
    hipLaunchKernelGGL(mt::mt_box_muller_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, uniform_ws, out,
                       reinterpret_cast<const mt::Seg*>(segs), nseg, (long long)total_groups);
  }
  return check_launch("mt19937_normal (transform)");
#else
  (void)stream; set_error("mt19937_normal: GPU build only"); return MLHOT_ERR_ARG;
#endif
}

// ---- whole ResNet trunks (weight-stationary kernels, csrc/resnet_ws.h / resnet_trunk.h) --------------------------------
size_t mlhot_trunk_act_floats(int C, int H, int n_img, int k) {
#ifndef MLHOT_HOSTSIM
  if (k < 0 || k > 8 || n_img < 0) return 0;
  return rt::act_floats(rt::trunk_levels(C, H), n_img, k);
#else
  (void)C; (void)H; (void)n_img; (void)k; return 0;
#endif
}
size_t mlhot_trunk_scratch_bytes(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H, int backward) {
#ifndef MLHOT_HOSTSIM
  if (rt::trunk_check(passes, n_pass, wsets, n_wset, C, H)) return 0;
  return rt::trunk_carve(passes, n_pass, wsets, n_wset, rt::trunk_levels(C, H), backward != 0, nullptr, 0).bytes;
#else
  (void)passes; (void)n_pass; (void)wsets; (void)n_wset; (void)C; (void)H; (void)backward; return 0;
#endif
}
int mlhot_trunk_fwd(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H, void* scratch,
                    size_t scratch_bytes, void* stream) {
#ifndef MLHOT_HOSTSIM
  return rt::trunk_forward(passes, n_pass, wsets, n_wset, C, H, scratch, scratch_bytes, (hipStream_t)stream);
#else
  (void)passes; (void)n_pass; (void)wsets; (void)n_wset; (void)C; (void)H; (void)scratch; (void)scratch_bytes; (void)stream;
  set_error("resnet trunk: GPU build only"); return MLHOT_ERR_UNSUPPORTED;
#endif
}
int mlhot_trunk_bwd(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H, void* scratch,
                    size_t scratch_bytes, void* stream) {
#ifndef MLHOT_HOSTSIM
  return rt::trunk_backward(passes, n_pass, wsets, n_wset, C, H, scratch, scratch_bytes, (hipStream_t)stream);
#else
  (void)passes; (void)n_pass; (void)wsets; (void)n_wset; (void)C; (void)H; (void)scratch; (void)scratch_bytes; (void)stream;
  set_error("resnet trunk: GPU build only"); return MLHOT_ERR_UNSUPPORTED;
#endif
}

// ---- batch ingest: uint8 channel-last images -> fp32 channel-first, divided by `div` -----------------------
int mlhot_ingest_u8_nhwc(const uint8_t* src, float* dst, long n_img, int H, int W, int C, float div, void* stream) {
  if (n_img < 0 || H <= 0 || W <= 0 || C <= 0 || !(div > 0.f) || (n_img > 0 && (!src || !dst))) {
    set_error("ingest_u8_nhwc: bad argument");
    return MLHOT_ERR_ARG;
  }
  return ingest::run(src, dst, n_img, H, W, C, div, (hipStream_t)stream);
}

// ---- fused Adam over a flat parameter / gradient buffer ------------------------------------------------
int mlhot_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float grad_scale, int step, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) { set_error("adam_step: bad argument"); return MLHOT_ERR_ARG; }
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  return run_foreach(AdamStep{param, grad, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, grad_scale, (float)(lr / bc1),
                              (float)(1.0 / sqrt(bc2))}, n, (hipStream_t)stream, "adam.step");
}

int mlhot_adam_step_counter(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, float grad_scale, int* step_counter, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !step_counter) { set_error("adam_step_counter: bad argument"); return MLHOT_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  MLHOT_TRY(run_foreach(CounterInc{step_counter}, 1, s, "adam.count"));
  return run_foreach(AdamStepCounter{param, grad, exp_avg, exp_avg_sq, beta1, beta2, eps, weight_decay, grad_scale, lr, step_counter}, n, s,
                     "adam.step");
}

// ---- X1 building blocks: train-mode batch norm (+ReLU) over one task's shots, spatial mean ----------
int mlhot_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                      int N, int C, int HW, float* y, float* mean, float* var, void* stream) {
  if (N <= 0 || C <= 0 || HW <= 0) { set_error("bn_relu_fwd: bad argument"); return MLHOT_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  MLHOT_TRY(run_reduce_seg(BnStats{x, C, HW, mean, N * HW}, C, N * HW, s, "bn.mean"));
  MLHOT_TRY(run_reduce_seg(BnVar{x, C, HW, mean, var, run_mean, run_var, momentum, N * HW}, C, N * HW, s, "bn.var"));
  return run_foreach(BnApplyRelu{x, mean, var, gamma, beta, eps, C, HW, y}, (size_t)N * C * HW, s, "bn.apply");
}
int mlhot_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean, const float* var, float eps,
                      int N, int C, int HW, float* dx, float* dgamma, float* dbeta, void* stream) {
  if (N <= 0 || C <= 0 || HW <= 0) { set_error("bn_relu_bwd: bad argument"); return MLHOT_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  MLHOT_TRY(run_reduce_seg(BnBwdSums{x, y, dy, mean, var, eps, C, HW, dgamma, dbeta}, C, N * HW, s, "bn.bwd.sums"));
  return run_foreach(BnBwdApply{x, y, dy, mean, var, gamma, dgamma, dbeta, eps, C, HW, N * HW, dx}, (size_t)N * C * HW, s, "bn.bwd.apply");
}
int mlhot_spatial_mean_fwd(const float* x, float* y, int planes, int HW, void* stream) {
  return run_foreach(SpatialMean{x, HW, y}, (size_t)planes, (hipStream_t)stream, "spatial_mean.fwd");
}
int mlhot_spatial_mean_bwd(const float* dy, float* dx, int planes, int HW, void* stream) {
  return run_foreach(SpatialMeanBwd{dy, HW, dx}, (size_t)planes * HW, (hipStream_t)stream, "spatial_mean.bwd");
}

// ---- whole model ------------------------------------------------------------------------------
size_t mlhot_np_struct_bytes(int which) {
  return which == 0 ? sizeof(mlhot_np_dims) : which == 1 ? sizeof(mlhot_np_params) : which == 2 ? sizeof(mlhot_np_grads)
       : which == 3 ? sizeof(mlhot_chain_layer) : which == 4 ? sizeof(mlhot_chain_grads) : sizeof(mlhot_linear_job);
}
size_t mlhot_np_saved_bytes(const mlhot_np_dims* d) { return d ? np_saved_carve(*d, nullptr, 0).bytes : 0; }
size_t mlhot_np_scratch_bytes(const mlhot_np_dims* d) { return d ? np_scratch_carve(*d, nullptr, 0).bytes : 0; }
size_t mlhot_np_grads_flat_layout(const mlhot_np_dims* d, mlhot_np_grads* offsets) { return (d && offsets) ? np_grads_flat_layout(*d, *offsets) : 0; }

int mlhot_np_vanilla_fwd(const mlhot_np_dims* d, const mlhot_np_params* p, const float* ctx_x, const float* ctx_y,
                         const float* qry_x, float* mu, void* saved, void* scratch, size_t scratch_bytes, void* stream) {
  if (!d || !p) { set_error("np_vanilla_fwd: null dims/params"); return MLHOT_ERR_ARG; }
  return np_forward(*d, *p, ctx_x, ctx_y, qry_x, mu, saved, scratch, scratch_bytes, (hipStream_t)stream);
}
int mlhot_np_vanilla_bwd(const mlhot_np_dims* d, const mlhot_np_params* p, const float* ctx_x, const float* ctx_y,
                         const float* qry_x, const float* mu, const float* dmu, const mlhot_np_grads* g, const void* saved,
                         void* scratch, size_t scratch_bytes, void* stream) {
  if (!d || !p || !g) { set_error("np_vanilla_bwd: null dims/params/grads"); return MLHOT_ERR_ARG; }
  return np_backward(*d, *p, ctx_x, ctx_y, qry_x, mu, dmu, *g, saved, scratch, scratch_bytes, (hipStream_t)stream);
}
int mlhot_np_vanilla_bwd_loss(const mlhot_np_dims* d, const mlhot_np_params* p, const float* ctx_x, const float* ctx_y,
                              const float* qry_x, const float* mu, const float* dmu, const mlhot_loss_desc* loss, const mlhot_np_grads* g,
                              const void* saved, void* scratch, size_t scratch_bytes, void* stream) {
  if (!d || !p || !g || !loss) { set_error("np_vanilla_bwd_loss: null dims/params/grads/loss"); return MLHOT_ERR_ARG; }
  if (loss->kind < 0 || loss->kind > 4 || loss->kind == 3 || !loss->gt || !loss->dloss || loss->gt_dim < 1) {
    set_error("np_vanilla_bwd_loss: bad loss descriptor (kinds 0, 1, 2, 4 have a gradient)"); return MLHOT_ERR_ARG;
  }
  const LossDesc ld{loss->kind, loss->gt, loss->gt_dim, loss->dloss, loss->value};
  return np_backward(*d, *p, ctx_x, ctx_y, qry_x, mu, dmu, *g, saved, scratch, scratch_bytes, (hipStream_t)stream, Stage{}, &ld);
}
int mlhot_np_vanilla_fwd_staged(const mlhot_np_dims* d, const mlhot_np_params* p, const float* ctx_x, const float* ctx_y,
                                const float* qry_x, float* mu, void* saved, void* scratch, size_t scratch_bytes, int stage, float* xchg,
                                void* stream) {
  if (!d || !p) { set_error("np_vanilla_fwd_staged: null dims/params"); return MLHOT_ERR_ARG; }
  MLHOT_TRY(staged_args(stage, xchg, "np_vanilla_fwd_staged"));
  return np_forward(*d, *p, ctx_x, ctx_y, qry_x, mu, saved, scratch, scratch_bytes, (hipStream_t)stream, Stage{stage, xchg});
}
int mlhot_np_vanilla_bwd_staged(const mlhot_np_dims* d, const mlhot_np_params* p, const float* ctx_x, const float* ctx_y,
                                const float* qry_x, const float* mu, const float* dmu, const mlhot_np_grads* g, const void* saved,
                                void* scratch, size_t scratch_bytes, int stage, float* xchg, void* stream) {
  if (!d || !p || !g) { set_error("np_vanilla_bwd_staged: null dims/params/grads"); return MLHOT_ERR_ARG; }
  MLHOT_TRY(staged_args(stage, xchg, "np_vanilla_bwd_staged"));
  return np_backward(*d, *p, ctx_x, ctx_y, qry_x, mu, dmu, *g, saved, scratch, scratch_bytes, (hipStream_t)stream, Stage{stage, xchg});
}

}  // extern "C"

#ifdef MLHOT_TS
// latency-hunting builds only (-DMLHOT_TS): device buffer of int64 stage timestamps, see tail_fused.h
extern "C" int mlhot_dbg_tsbuf(void* dev_ptr) {
  long long* p = static_cast<long long*>(dev_ptr);
  return hipMemcpyToSymbol(HIP_SYMBOL(mlhot::tf::g_ts_dev), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif
