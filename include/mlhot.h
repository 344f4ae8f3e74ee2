/* mlhot.h - C ABI of libmlhot.so, the MI355X-native CNP/ANP meta-batch hot path.
 *
 * The reference (boschresearch/what-matters-for-meta-learning) is pure Python on torch and
 * has no FFI of its own; its plugin boundary is `networks/<Method>.py::<Method>(config)`
 * (train.py:41-45).  This header is the C boundary UNDER that plugin (SURVEY.md §8b, row
 * "C-ABI under the plugin"): one forward and one backward entry per hot-path row of
 * SURVEY.md §8(a), each citing the reference code it replaces.  The Python host mirror
 * (what-matters-for-meta-learning_amd/networks/*) binds these with ctypes.
 *
 * Contract for every entry:
 *   - returns 0 on success, a non-zero MLHOT_ERR_* code otherwise (mlhot_last_error() has text);
 *   - all pointers are DEVICE pointers to fp32 unless noted, caller-owned, 16-byte aligned;
 *   - no allocation, no host synchronisation, no default-stream use inside: work is enqueued
 *     on `stream` in order, so calls are hipGraph-capturable and re-entrant across streams;
 *   - workspaces are caller-provided; the *_bytes() helpers size them.
 *   - `stream` is a hipStream_t passed as void*.
 *
 * Threading (the reference's callers are ONE Python thread on the default stream).  The compute entries keep no
 * state between calls - everything lives in the caller's buffers - so concurrent calls from several host threads on different
 * streams and buffers are safe, and mlhot_last_error() is per thread.  The two DIAGNOSTIC facilities are process-global and are
 * not part of that guarantee: mlhot_set_option() flips implementation switches read by every later call of every thread (set
 * them before the first compute call, or from the one thread that issues the calls - the A/B tests do the latter), and the
 * mlhot_prof_begin/end() recorder may be driven by one thread at a time while no other thread is inside the library.
 */
#ifndef MLHOT_H
#define MLHOT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLHOT_ABI_VERSION 7

enum { MLHOT_ACT_NONE = 0, MLHOT_ACT_RELU = 1, MLHOT_ACT_TANH = 2 };
enum { MLHOT_AGG_MEAN = 0, MLHOT_AGG_MAX = 1, MLHOT_AGG_BACO = 2, MLHOT_AGG_ATTENTION = 3 };
/* trainer/losses.py:32-80 */
enum { MLHOT_LOSS_AZIMUTH = 0, MLHOT_LOSS_MSE = 1, MLHOT_LOSS_QUATERNION = 2, MLHOT_LOSS_DEGREE = 3, MLHOT_LOSS_DISTRACTOR = 4 };

int mlhot_version(void);
const char* mlhot_last_error(void);

/* Implementation switches for A/B tests (process-global, see Threading above): "conv2_tc" = 1 (default) runs the
 * weight-stationary conv2 kernels (csrc/conv_tc.h), 0 the generic implicit-GEMM problems; "tail_fused" = 1 (default) runs the
 * fused per-task tail kernels where they apply, "tail_spec" = bit mask of the six tail phases that use the kernels specialised
 * for the shipped dimensions (csrc/tail_spec.h; default 63 = all, 0 = the run-time-shaped csrc/tail_fused.h); "favor2" = 1
 * (default) the two-launch FAVOR+ kernels, 0 the operator chain; "materialize_a1" = 1 additionally stores the conv1 output
 * (debug / tests; the fused conv1+conv2 kernels never need it); "conv2_split" = 1 (default 0) runs the vanilla encoder's conv1 +
 * conv2 + pool forward with conv2 on the bf16 matrix pipe over exact hi / mid / lo splits of the fp32 operands (csrc/conv_split.h:
 * same outputs' layout, same parity tolerances, 1.5 x the fp32 kernel; the benchmark's headline stays on the fp32 kernels).      */
int mlhot_set_option(const char* name, int value);

/* ---- bench-only: per-launch HIP-event timing ------------------------------------------------
 * Between begin and end every kernel launch of the library is bracketed by two events recorded
 * on its launch stream.  end() synchronises them and returns (label, milliseconds) records.
 * Not re-entrant, not capturable; used only by bench.py's roofline leg.                        */
int mlhot_prof_begin(int max_records);
int mlhot_prof_end(const char** labels, float* ms, int cap);

/* ---- NT-Xent, the functional-contrastive term of the FCL* models ------------------
 * replaces trainer/losses.py:82-99 (LossFunc.contrastive_loss / contrastive_loss_ANP -> pytorch_metric_learning.losses.NTXentLoss
 * (temperature = t), an un-vendored dependency: algorithm restated in oracle/ref_cpu.py::nt_xent, value parity unpinned).  z: [N, d]
 * embeddings (N <= 2048, d <= 256, d % 16 == 0); the labels are always arange blocks, label(i) = (i / div) % mod
 * (contrastive_loss: div = 1, mod = T, N = 2T; contrastive_loss_ANP: div = Nq, mod = T, N = T Nq).  ws: mlhot_nt_xent_ws_floats(N)
 * floats kept between forward and backward.  loss / dloss: device scalars; dz: [N, d].                                          */
size_t mlhot_nt_xent_ws_floats(int N);
int mlhot_nt_xent_fwd(const float* z, int N, int d, int div, int mod, float t, float* ws, float* loss, void* stream);
int mlhot_nt_xent_bwd(const float* z, int N, int d, int div, int mod, float t, const float* ws, const float* dloss, float* dz, void* stream);

/* ---- B1 (eps stream): torch's CPU normal_() random stream continued on the device -------------------
 * replaces the host-side `torch.empty(size).normal_(0, 1)` + `.to(device)` of bbb/BBBConv.py:86-95 and BBBLinear.py:79-88 for callers
 * that hand the generator over (what-matters-for-meta-learning_amd/mlhot/rng.py): MT19937 (ATen/core/MT19937RNGEngine.h), 24-bit
 * float uniforms, ATen's 16-wide Box-Muller normal_fill.  engine: uint32[626] = state[624], left, next - the engine's fields,
 * advanced in place exactly as `total_outputs` calls would.  segs: nseg records of 4 int64 {dst offset in out, size (>= 16), stream
 * offset of its first output, index of its first group of 16} in call order; a size that is not a multiple of 16 consumes
 * size + 16 outputs and owns size / 16 + 1 groups.  uniform_ws: total_outputs floats of scratch.  Uniforms and engine state are
 * bit-identical to torch's, the normals equal up to the last ulps of logf / sincosf.                                        */
int mlhot_mt19937_normal(uint32_t* engine, float* uniform_ws, float* out, const int64_t* segs, int nseg, int64_t total_outputs,
                         int64_t total_groups, void* stream);
/* The same draw from n_sub parallel sub-streams of stride_blocks 624-word blocks each (MT19937 jump-ahead; the polynomials
 * polys[n_sub - 1][624] = t^(624 stride_blocks k) mod phi, k = 1 .. n_sub - 1, come from mlhot/mt_jump.py): identical uniforms, normals
 * and final engine state.  Needs n_sub * stride_blocks >= the number of new blocks of the draw; jump_ws: mlhot_mt19937_jump_ws_words(n_sub)
 * uint32 words of scratch.  ~0.1 ms for 0.9 M outputs where the one-workgroup form takes ~1 ms.                               */
size_t mlhot_mt19937_jump_ws_words(int n_sub);
int mlhot_mt19937_normal_par(uint32_t* engine, float* uniform_ws, float* out, const int64_t* segs, int nseg, int64_t total_outputs,
                             int64_t total_groups, const uint32_t* polys, int n_sub, int stride_blocks, uint32_t* jump_ws, void* stream);

/* Host only (no device, no stream): the engine (uint32[626] as above) moved forward by n_outputs calls without producing them - the
 * state `n_outputs` draws of torch's CPU generator would leave behind.  Lets K host threads draw K contiguous pieces of one
 * `normal_()` stream at once, each on a torch.Generator set to its piece's starting state (networks/bbb/eps.py, bit-identical to
 * the sequential draw of bbb/BBBConv.py:86-95).                                                                              */
int mlhot_mt19937_advance(uint32_t* engine, uint64_t n_outputs);

/* Host only (no device, no stream; ABI 6): the inverse of the loaders' host conversion `img.astype(float32) / 255.0`
 * (dataset/shapenet_1d.py:189-190, dataset/shapenet_3d.py, utils/utils.py:26-30), checked element by element: dst[i] = the byte k
 * nearest to src[i] * div, *n_inexact = the number of elements whose (float)k / div differs from src[i] in any bit.  When it is 0
 * the batch may cross PCIe as the n bytes of dst and mlhot_ingest_u8_nhwc (same div) reproduces src bit for bit on the device;
 * otherwise the caller ships the fp32 data as before.  Element order is kept (a channel-first fp32 batch gives channel-first bytes:
 * ingest it as [n_img * C, H, W, 1]).  `threads` (1 .. 64): the conversion runs on that many native host threads started by
 * the call (pieces of >= 64 K elements); thread-safe.                                                                         */
int mlhot_host_f32_to_u8_exact(const float* src, uint8_t* dst, int64_t n, float div, int threads, int64_t* n_inexact);

/* ---- E1: vanilla image encoder `encoder_w0` -------------------------------------------
 * replaces nn.Sequential(conv3x3s2+ReLU, conv3x3s2+ReLU, MaxPool2d(2), conv3x3s2+ReLU,
 * Flatten, Linear(4096,dim_w))   (networks/ANPShapeNet1D.py:46-56, CNPShapeNet1D.py:46-56,
 * CNPVanillaPascal1D.py:48-58, ANPVanillaPascal1D.py:50-60) for [n,1,128,128] images.
 * The image batch may come as two segments (context images, target images) that share the
 * weights; rows [0,n0) of the logical batch read img0 / write feat0, rows [n0,n0+n1) use
 * img1 / feat1 (n1 may be 0).  feat rows have leading dimension ld0 / ld1 (floats).        */
typedef struct mlhot_enc_params {
  const float *w1, *b1;   /* encoder_w0.0  [32,1,3,3],  [32] */
  const float *w2, *b2;   /* encoder_w0.2  [48,32,3,3], [48] */
  const float *w3, *b3;   /* encoder_w0.5  [64,48,3,3], [64] */
  const float *wl, *bl;   /* encoder_w0.8  [dim_w,4096], [dim_w] */
} mlhot_enc_params;
typedef struct mlhot_enc_grads {
  float *w1, *b1, *w2, *b2, *w3, *b3, *wl, *bl;
} mlhot_enc_grads;

size_t mlhot_enc_vanilla_saved_bytes(int n_img);
size_t mlhot_enc_vanilla_scratch_bytes(int n_img, int dim_w);
int mlhot_enc_vanilla_fwd(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params* p, int dim_w,
                          float* feat0, int ld0, float* feat1, int ld1,
                          void* saved, void* scratch, size_t scratch_bytes, void* stream);
int mlhot_enc_vanilla_bwd(const float* img0, int n0, const float* img1, int n1, const mlhot_enc_params* p, int dim_w,
                          const float* dfeat0, int ldd0, const float* dfeat1, int ldd1,
                          const void* saved, const mlhot_enc_grads* g, void* scratch, size_t scratch_bytes, void* stream);

/* E1's first block on its own - conv1 (1 -> 32, 3x3 s2 p1) + ReLU + conv2 (32 -> 48, 3x3 s2 p1) + ReLU + 2x2 max-pool of n
 * 128 x 128 images (networks/conv_embedding_model.py:18-31, the first five layers) - for tests and micro-benchmarks of the three
 * kernels that carry 80 % of the vanilla models' FLOPs.  `saved` has the layout and size of mlhot_enc_vanilla_saved_bytes(n): the
 * forward leaves p2 [n,48,16,16], the pool arg-max and conv1's ReLU sign bits there; the backward takes d p2 and returns the four
 * parameter gradients.  Option "conv2_split" (bit 1 forward, 2 data gradient, 4 weight gradient) selects the split-bf16 kernels. */
size_t mlhot_conv12_scratch_bytes(int n_img);
int mlhot_conv12_fwd(const float* img, int n_img, const float* w1, const float* b1, const float* w2, const float* b2,
                     void* saved, void* stream);
int mlhot_conv12_bwd(const float* img, int n_img, const float* w1, const float* b1, const float* w2, const float* dp2,
                     const void* saved, float* dw1, float* db1, float* dw2, float* db2, void* scratch, size_t scratch_bytes,
                     void* stream);

/* ---- M1 / D1 / A1 / A4: nn.Linear (+ReLU / tanh) ----------------------------------------
 * y[M,N] = act(x[M,K] w[N,K]^T + b)   (networks/models.py:27-60 EncoderFC, 195-203 AttnLinear;
 * ANPShapeNet1D.py:58-72 transform_y / r_to_z / decoder0).  b may be NULL.                  */
size_t mlhot_linear_bwd_scratch_bytes(int M, int K, int N);
int mlhot_linear_fwd(const float* x, int ldx, const float* w, const float* b, float* y, int ldy,
                     int M, int K, int N, int act, void* stream);
/* dx (+)= (dy*act'(y)) w ; dw = (dy*act'(y))^T x ; db = column sums.  dx/dw/db may be NULL. */
int mlhot_linear_bwd(const float* x, int ldx, const float* w, const float* y, int ldy, const float* dy, int lddy,
                     int M, int K, int N, int act, float* dx, int lddx, int accumulate, float* dw, float* db,
                     void* scratch, size_t scratch_bytes, void* stream);

/* ---- M1 / D2 as chains: up to 4 Linear(+ReLU / tanh) layers on few rows (M <= 512) in ONE launch --------------------------
 * The ResNet-family models' task-side MLPs (networks/ANP.py:44-52 task_encoder + mu, models.py:139-145,182-184 fc_mu behind
 * torch.cat([x, sample_features]); ANPMRShapeNet3D.py:135-139,204-216).  Layer k computes
 *     y_k = act_k([side_k | y_{k-1}] w_k^T + b_k)   (side_first = 1)    or    act_k([y_{k-1} | side_k] w_k^T + b_k)   (side_first = 0)
 * with y_{-1} = x0 and side_k an optional second input tensor of side_w columns (the reference's torch.cat, folded: the labels
 * behind the context features, the decoder's image features in front of the sampled latent).  K = the layer's total input width
 * (side_w + the previous layer's N; 4 <= K <= 512, K % 4 == 0, side_w % 4 == 0), N <= 256 (inner layers N % 4 == 0); every y_k is
 * written to the caller's buffer (ldy >= N): the backward reads them.  All operand rows 16-byte aligned.                       */
#define MLHOT_CHAIN_MAX_LAYERS 4
typedef struct {
  const float* w; const float* b;      /* [N][K], [N] or NULL */
  int K, N, act;
  const float* side; int side_w, side_ld, side_first;   /* side_w = 0: no side input */
  float* y; int ldy;                   /* [M][ldy]: this layer's output */
} mlhot_chain_layer;
typedef struct {
  float* dw; float* db;                /* [N][K], [N] or NULL: written */
  float* g; int ldg;                   /* workspace [M][ldg], ldg >= N, ldg % 4 == 0: dy_k * act'(y_k) */
  float* dside; int dside_ld, dside_accumulate;         /* gradient of the side columns, or NULL */
} mlhot_chain_grads;
int mlhot_mlp_chain_fwd(const float* x0, int ldx0, int M, const mlhot_chain_layer* layers, int n_layers, void* stream);
/* dy[M][lddy]: gradient of the last layer's output; dx0 (NULL: not wanted) (+)= the gradient of x0's columns.  Two launches:
 * the data-gradient walk (also fills every g) and ONE weight + bias gradient launch for all layers.                             */
int mlhot_mlp_chain_bwd(const float* x0, int ldx0, int M, const mlhot_chain_layer* layers, const mlhot_chain_grads* grads,
                        int n_layers, const float* dy, int lddy, float* dx0, int lddx0, int dx0_accumulate, void* stream);

/* Up to 8 INDEPENDENT few-row Linear layers in one launch (the K / V / Q head stacks of the attention, ANP.py:80-93: one
 * Linear(256 -> 8 x 256) each over the stacked head weights): forward y = act(x w^T + b); backward dx (+)= (dy act'(y)) w,
 * dw = (dy act'(y))^T x, db.  M <= 512, K % 4 == 0, rows 16-byte aligned; the backward needs N % 4 == 0 and all of dy, dx, dw.  */
typedef struct {
  const float* x; int ldx; const float* w; const float* b; float* y; int ldy; int M, K, N, act;
  const float* dy; int lddy; float* dx; int lddx, dx_accumulate; float* dw; float* db;      /* backward only */
  /* two-source input (the reference's torch.cat([x, x2], -1) in front of the layer, folded): the LAST K2 of the layer's K input
   * columns come from x2 (NULL: all K from x; K2 % 4 == 0); dx2 receives their gradient (NULL: not wanted; dx may then be NULL too) */
  const float* x2; int ldx2, K2; float* dx2; int lddx2;
} mlhot_linear_job;
int mlhot_linear_multi_fwd(const mlhot_linear_job* jobs, int n_jobs, void* stream);
int mlhot_linear_multi_bwd(const mlhot_linear_job* jobs, int n_jobs, void* stream);

/* ---- G1: per-task aggregation over the shot axis ------------------------------------------
 * mean / max / Bayesian ("baco") over dim 1 of rs[T,Nc,R]   (CNPShapeNet1D.py:78-126,
 * CondNeuralProcess.py:59-108).  baco: `rs` holds mu, `lv` the pre-softplus variance logits;
 * var = 1e-5 + softplus(lv), sigma_z = 1/(1+sum 1/var), r = sigma_z * sum(mu/var).
 * amax[T,R] (int32) receives the arg-max shot for mode MAX.                                  */
int mlhot_agg_fwd(int mode, const float* rs, const float* lv, int T, int Nc, int R,
                  float* r, float* sigma_z, int32_t* amax, void* stream);
int mlhot_agg_bwd(int mode, const float* rs, const float* lv, const float* r, const float* sigma_z, const int32_t* amax,
                  const float* dr, int T, int Nc, int R, float* drs, float* dlv, void* stream);

/* ---- A2 / A3: FAVOR+ softmax-kernel features + non-causal linear attention ----------------
 * replaces FastAttention.forward = linear_attention(softmax_kernel(q,True),
 * softmax_kernel(k,False), v)   (networks/fast_attention.py:74-99,151-156,187-205), including
 * the batch-global key stabiliser torch.max(data_dash) (fast_attention.py:97).
 * Layouts: q[T,Nq,H,d], k[T,Nc,H,d], v[T,Nc,H,d] (token-major, head-minor rows - what the head
 * projections write), proj[m,d].  out[T,Nq,d*H] is already in the reference's merged order
 * out[t,n,e*H+h] (ANPShapeNet1D.py:113-114 permute(0,2,3,1).view).                           */
size_t mlhot_favor_ws_bytes(int T, int H, int Nq, int Nc, int d, int m);
int mlhot_favor_fwd(const float* q, const float* k, const float* v, const float* proj,
                    int T, int H, int Nq, int Nc, int d, int m, float* out,
                    void* ws, size_t ws_bytes, void* stream);
/* ws must be the workspace the matching forward filled. */
int mlhot_favor_bwd(const float* q, const float* k, const float* v, const float* proj,
                    int T, int H, int Nq, int Nc, int d, int m, const float* out, const float* dout,
                    float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, void* stream);

/* ---- strict sharded parity of the key stabiliser -------------------------
 * fast_attention.py:96-97 takes torch.max over the keys of the WHOLE batch.  When a caller shards the meta-batch over
 * ranks, each rank's launch sequence can be run in two halves around the caller's own collectives on `xchg`
 * (device memory, 4 floats, caller-owned):
 *   forward  stage 0: everything up to the rank-local key maximum           -> xchg[0] = that maximum
 *            (caller: xchg[0] = MAX over ranks; xchg[1] = 1 on the lowest rank whose maximum equals it, else 0)
 *            stage 1: the rest of the forward, with xchg[0] as the stabiliser
 *   backward stage 0: everything up to the rank-local sum of dL/d(stabiliser) -> xchg[2] = that sum
 *            (caller: xchg[2] = SUM over ranks)
 *            stage 1: the rest; the rank with xchg[1] = 1 routes the batch-wide sum to its arg-max key, the others to none
 * The same saved / scratch / ws buffers must be passed to both stages; a world of one (xchg[1] = 1, xchg untouched) reproduces
 * the unstaged call.  No communication library is linked: the collective is the caller's (mlhot/dist.py: StabiliserExchange).  */
int mlhot_favor_fwd_staged(const float* q, const float* k, const float* v, const float* proj,
                           int T, int H, int Nq, int Nc, int d, int m, float* out,
                           void* ws, size_t ws_bytes, int stage, float* xchg, void* stream);
int mlhot_favor_bwd_staged(const float* q, const float* k, const float* v, const float* proj,
                           int T, int H, int Nq, int Nc, int d, int m, const float* out, const float* dout,
                           float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, int stage, float* xchg, void* stream);

/* ---- L1: losses --------------------------------------------------------------------------
 * LossFunc.calc_loss (trainer/losses.py:32-80).  mu[rows,y_dim], gt[rows,gt_dim];
 * loss / dloss are device scalars.                                                            */
int mlhot_loss_fwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, float* loss, void* stream);
int mlhot_loss_bwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim,
                   const float* dloss, float* dmu, void* stream);
/* ABI 7: the trainer's objective `losses = loss + kl * beta` (trainer/model_trainer.py:77-78) inside the loss's own launches - in a
 * replayed step every dependent launch costs ~4.7 us whatever it computes, and this pair of scalars was four of them.
 *   fwd: loss[0] (may be NULL) as mlhot_loss_fwd; total[0] = loss + alpha * x[0], product and sum rounded separately (mlhot_axpy's bits)
 *   bwd: dmu as mlhot_loss_bwd(dloss = dtotal); dx[0] (may be NULL) = alpha * dtotal[0]
 * x, total, dtotal, dx: device scalars.                                                                                               */
int mlhot_loss_plus_fwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, const float* x, float alpha,
                        float* loss, float* total, void* stream);
int mlhot_loss_plus_bwd(int kind, const float* mu, const float* gt, int rows, int y_dim, int gt_dim, const float* dtotal, float alpha,
                        float* dmu, float* dx, void* stream);

/* ---- E2 / D2: ResNet encoder building blocks --------------------------------------------------
 * nn.Conv2d (cross-correlation, zero padding, NCHW, weight [Cout,Cin,k,k], optional fused ReLU) for
 * the 5x5 s2 p2 stem, the 3x3 s2 / s1 p1 block convs and the 1x1 s2 (3x3 in the BBB twin) skip of
 * ImageEncoder / NPDecoder (networks/models.py:63-192, networks/ResNet.py:25-74).  Backward: `y` is
 * the forward output (used for the ReLU mask when relu=1); dx / dw / db may be NULL.               */
size_t mlhot_conv2d_bwd_scratch_bytes(int N, int Cin, int H, int W, int Cout, int k, int stride, int pad);
int mlhot_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout, int k,
                     int stride, int pad, int relu, void* stream);
int mlhot_conv2d_bwd(const float* x, const float* w, const float* y, const float* dy, int N, int Cin, int H, int W, int Cout, int k,
                     int stride, int pad, int relu, float* dx, float* dw, float* db, void* scratch, size_t scratch_bytes, void* stream);
/* residual join y = relu(a + b) (ResNet.py:69-72); backward g = dy * (y > 0) is the gradient of both inputs */
/* y[i] = a[i] + alpha * x[i] (a NULL: alpha * x[i]); product and sum rounded separately.  The trainer's `loss + kl * beta`
 * (trainer/model_trainer.py:77-78) as one launch per direction instead of two elementwise torch operators.                   */
int mlhot_axpy(const float* a, const float* x, float alpha, float* y, size_t n, void* stream);
int mlhot_add_relu_fwd(const float* a, const float* b, float* y, size_t n, void* stream);
int mlhot_add_relu_bwd(const float* y, const float* dy, float* g, size_t n, void* stream);
/* 2x2 max-pool over `planes` maps of H x W (nn.AdaptiveMaxPool2d((2,2)) on 4x4 maps, models.py:107-110) */
int mlhot_pool2_fwd(const float* x, float* y, uint8_t* amax, int planes, int H, int W, void* stream);
int mlhot_pool2_bwd(const float* dy, const uint8_t* amax, float* dx, int planes, int H, int W, void* stream);

/* ---- E2 / D2 / B1: whole ResNet trunks in one call -------------------------------------------------------------------
 * The 5x5 s2 stem (+ReLU) and the four BN-free BasicBlocks {conv3x3 s2 + ReLU; conv3x3 s1; skip conv (1x1 s2 in ImageEncoder /
 * NPDecoder, networks/ResNet.py:32-34,58-74; 3x3 s2 p1 in the Bayes-by-backprop twin, networks/ANPMRShapeNet3D.py:48-51,85);
 * add; ReLU} of networks/models.py:63-117,156-182 - for EVERY pass of a model step at once (context images, target images,
 * decoder images; each pass names its weight set, passes may share one).  Weight-stationary fp32-MFMA kernels (csrc/resnet_ws.h);
 * supported inputs: 3 x 64 x 64 (ShapeNet3D) and 1 x 128 x 128 (Distractor), MLHOT_ERR_UNSUPPORTED otherwise (callers then
 * compose mlhot_conv2d_* instead).
 *   w / b [13]: stem, then (conv1, conv2, skip) of blocks 1..4, in the reference's own layouts [Cout][Cin][k][k] / [Cout];
 *   act [9]   : saved activations, written by the forward, read by the backward: a0 = stem output, then (mid_i, y_i) of the
 *               blocks (post-ReLU); act[8] is the trunk's output map [n][64][H/32][H/32];
 *   dfeat     : backward: gradient wrt act[8];  dw / db: gradients (overwritten; summed over the passes that share the set).  */
#define MLHOT_TRUNK_MAX_PASS 6
#define MLHOT_TRUNK_MAX_WSET 4
typedef struct mlhot_trunk_wset { const float* w[13]; const float* b[13]; float* dw[13]; float* db[13]; int skip_k; } mlhot_trunk_wset;
typedef struct mlhot_trunk_pass { const float* img; int n_img; int wset; float* act[9]; const float* dfeat; } mlhot_trunk_pass;
size_t mlhot_trunk_act_floats(int C, int H, int n_img, int k);
size_t mlhot_trunk_scratch_bytes(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H, int backward);
int mlhot_trunk_fwd(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H,
                    void* scratch, size_t scratch_bytes, void* stream);
int mlhot_trunk_bwd(const mlhot_trunk_pass* passes, int n_pass, const mlhot_trunk_wset* wsets, int n_wset, int C, int H,
                    void* scratch, size_t scratch_bytes, void* stream);

/* ---- B1: Bayes-by-backprop weight sample + KL (bbb/BBBConv.py:86-108, bbb/BBBLinear.py:79-101) ----
 * w = mu + eps * log1p(exp(rho)); kl = sum 0.5*(2 log(sigma/0.1) - 1 + (0.1/sigma)^2 + (mu/sigma)^2).
 * eps is drawn by the caller on the torch CPU generator (parity with BBBConv.py:88). klterm: n floats. */
int mlhot_bbb_sample_fwd(const float* mu, const float* rho, const float* eps, float* w, float* klterm, float* kl, size_t n, void* stream);
int mlhot_bbb_sample_bwd(const float* mu, const float* rho, const float* eps, const float* dw, const float* dkl, float* dmu, float* drho,
                         size_t n, void* stream);

/* The same for up to MLHOT_BBB_MAX_ITEMS tensors in one launch pair (a Bayes-by-backprop encoder samples every layer's weight
 * and bias once per forward: 26 tensors for ANPMRShapeNet3D.py:40-90): w_i = mu_i + eps_i * softplus(rho_i) for every item,
 * kl = the sum of ALL items' KL terms.  `partial`: mlhot_bbb_sample_multi_scratch_floats() floats of scratch.  Backward: items
 * carry dw (may be NULL: no gradient reached that sample), dmu, drho (overwritten); dkl as above.
 * An item may carry a SECOND independent sample of the same posterior (eps2 -> w2; backward dw2): the model encodes the context
 * and the target images with two samples per step (ANPMRShapeNet3D.py:198-199); both come out of one launch, the KL (which does
 * not depend on eps) is computed once, and the backward adds both samples' contributions into dmu / drho in one pass. */
#define MLHOT_BBB_MAX_ITEMS 32
typedef struct {
  const float* mu; const float* rho; const float* eps;
  float* w;                       /* forward output */
  const float* dw; float* dmu; float* drho;   /* backward */
  size_t n;
  const float* eps2; float* w2; const float* dw2;   /* optional second sample (all NULL when unused) */
} mlhot_bbb_item;
size_t mlhot_bbb_sample_multi_scratch_floats(const mlhot_bbb_item* items, int n_items);
int mlhot_bbb_sample_multi_fwd(const mlhot_bbb_item* items, int n_items, float* partial, float* kl, void* stream);
int mlhot_bbb_sample_multi_bwd(const mlhot_bbb_item* items, int n_items, const float* dkl, void* stream);

/* ---- batch ingest: the host-side image conversion of the data loaders, on the device ----
 * dst[n][c][y][x] = (float)src[n][y][x][c] / div, i.e. dataset/shapenet_1d.py:189-190 (`xs.astype(np.float32) / 255.0`;
 * same in dataset/pascal_1d.py, shapenet_3d.py, distractor.py) followed by utils/utils.py:26-30
 * (convert_channel_last_np_to_tensor: permute(0,1,4,2,3).contiguous()).  Bit-identical to that host arithmetic (IEEE
 * fp32 divide).  src: n_img*H*W*C bytes packed channel-last (4-byte aligned for the fast path), dst: n_img*C*H*W floats. */
int mlhot_ingest_u8_nhwc(const uint8_t* src, float* dst, long n_img, int H, int W, int C, float div, void* stream);

/* ---- optimizer: torch.optim.Adam (train.py:52-56) as ONE launch over flat buffers -------------------
 * param / grad / exp_avg / exp_avg_sq: n floats each, laid out alike (e.g. mlhot_np_grads_flat_layout).
 * step >= 1 is the 1-based update count (bias correction); grad_scale multiplies the gradient first
 * (1/world after a sum all-reduce); weight_decay is torch's L2 form (added to the gradient); amsgrad off. */
int mlhot_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, float grad_scale, int step, void* stream);

/* The same update with the 1-based step count kept in device memory: *step_counter is incremented on the device first, then
 * used for the bias corrections.  Capture-safe - a replayed hipGraph of a whole training step advances the count by itself. */
int mlhot_adam_step_counter(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, float grad_scale, int* step_counter, void* stream);

/* ---- X1: ConvEmbeddingModel building blocks (networks/conv_embedding_model.py:99-184) -------------
 * Train-mode batch norm over the shots of ONE task, fused with the ReLU that follows it:
 * y = relu(gamma * (x - mean_c) / sqrt(var_c + eps) + beta) with per-channel batch statistics; like
 * F.batch_norm(training=True) it updates run_mean / run_var in place (momentum, unbiased variance).
 * mean / var [C] are returned for the backward.  x, y: [N, C, HW].                                 */
int mlhot_bn_relu_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float momentum, float eps,
                      int N, int C, int HW, float* y, float* mean, float* var, void* stream);
int mlhot_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean, const float* var, float eps,
                      int N, int C, int HW, float* dx, float* dgamma, float* dbeta, void* stream);
/* mean over the HW axis of `planes` maps (torch.mean(x.view(n, c, -1), dim=2), conv_embedding_model.py:119-123) */
int mlhot_spatial_mean_fwd(const float* x, float* y, int planes, int HW, void* stream);
int mlhot_spatial_mean_bwd(const float* dy, float* dx, int planes, int HW, void* stream);

/* ---- whole vanilla CNP/ANP model: forward + backward in one call each --------------------
 * replaces <Model>.forward for CNPVanillaPascal1D / CNPShapeNet1D / ANPVanillaPascal1D /
 * ANPShapeNet1D (CNPShapeNet1D.py:96-140, ANPShapeNet1D.py:93-157) and its autograd backward. */
#define MLHOT_MAX_HIDDEN 4
#define MLHOT_HEADS 8
typedef struct mlhot_np_dims {
  int T, Nc, Nq;              /* tasks, context shots, target shots (Nc may be 0)             */
  int label_dim, y_dim;       /* config.input_dim, config.output_dim                           */
  int dim_w, dim_r, dim_z;    /* 64, 64|100|256, 64                                            */
  int n_hidden;               /* len(n_hidden_units_r)                                         */
  int hidden[MLHOT_MAX_HIDDEN];
  int dec_hidden;             /* 100                                                           */
  int agg_mode;               /* MLHOT_AGG_*                                                   */
  int out_tanh;               /* 1 for the ShapeNet1D classes                                  */
  int m_feat;                 /* FAVOR+ features (attention only)                              */
} mlhot_np_dims;

typedef struct mlhot_np_params {
  mlhot_enc_params enc;
  const float *ty_w, *ty_b;                                   /* transform_y                  */
  const float *er_w[MLHOT_MAX_HIDDEN + 1], *er_b[MLHOT_MAX_HIDDEN + 1]; /* encoder_r.layers.{0,2,..} */
  const float *r2z_w, *r2z_b;                                 /* r_to_z                       */
  const float *dec_w[3], *dec_b[3];                           /* decoder0.{0,2,4}             */
  const float *mu_w, *mu_b, *var_w, *var_b;                   /* rs_to_mu / rs_to_var (baco)  */
  const float *wk_w[MLHOT_HEADS], *wk_b[MLHOT_HEADS];         /* _W_k.i.linear                */
  const float *wv_w[MLHOT_HEADS], *wv_b[MLHOT_HEADS];         /* _W_v.i.linear                */
  const float *wq_w[MLHOT_HEADS], *wq_b[MLHOT_HEADS];         /* _W_q.i.linear                */
  const float *wo_w, *wo_b;                                   /* _W.linear                    */
  const float *proj;                                          /* attn.projection_matrix       */
} mlhot_np_params;

typedef struct mlhot_np_grads {
  mlhot_enc_grads enc;
  float *ty_w, *ty_b;
  float *er_w[MLHOT_MAX_HIDDEN + 1], *er_b[MLHOT_MAX_HIDDEN + 1];
  float *r2z_w, *r2z_b;
  float *dec_w[3], *dec_b[3];
  float *mu_w, *mu_b, *var_w, *var_b;
  float *wk_w[MLHOT_HEADS], *wk_b[MLHOT_HEADS];
  float *wv_w[MLHOT_HEADS], *wv_b[MLHOT_HEADS];
  float *wq_w[MLHOT_HEADS], *wq_b[MLHOT_HEADS];
  float *wo_w, *wo_b;
} mlhot_np_grads;

size_t mlhot_np_struct_bytes(int which); /* 0 dims, 1 params, 2 grads, 3 chain_layer, 4 chain_grads, 5 linear_job: binding self-check */
size_t mlhot_np_saved_bytes(const mlhot_np_dims* d);
size_t mlhot_np_scratch_bytes(const mlhot_np_dims* d);
/* Recommended gradient layout: ONE flat fp32 buffer (what torch.optim / an all-reduce bucket want anyway,
 * train.py:52-56).  Fills the pointer fields of `offsets` with BYTE offsets into that buffer (null for parameters the
 * configuration does not have) and returns its size in floats.  mlhot_np_vanilla_bwd accepts any destination
 * pointers; when they follow this layout the per-task slab reduce of the fused tails is one contiguous sum. */
size_t mlhot_np_grads_flat_layout(const mlhot_np_dims* d, mlhot_np_grads* offsets);
/* ctx_x[T,Nc,1,128,128], ctx_y[T,Nc,label_dim], qry_x[T,Nq,1,128,128] -> mu[T,Nq,y_dim] */
int mlhot_np_vanilla_fwd(const mlhot_np_dims* d, const mlhot_np_params* p,
                         const float* ctx_x, const float* ctx_y, const float* qry_x, float* mu,
                         void* saved, void* scratch, size_t scratch_bytes, void* stream);
/* Every gradient buffer in g is OVERWRITTEN (parameters the forward did not use get zeros). */
int mlhot_np_vanilla_bwd(const mlhot_np_dims* d, const mlhot_np_params* p,
                         const float* ctx_x, const float* ctx_y, const float* qry_x,
                         const float* mu, const float* dmu, const mlhot_np_grads* g,
                         const void* saved, void* scratch, size_t scratch_bytes, void* stream);
/* The same backward taking the loss's gradient itself: trainer/model_trainer.py:77-79 computes `loss = calc_loss(mu, ., y)` and
 * calls loss.backward(); d loss / d mu of trainer/losses.py:32-80 needs no reduction, so the model's first backward kernel derives
 * it from (mu, gt) instead of reading a dmu that a separate launch wrote (the fused attention tail's specialised kernel does it in
 * its prologue; every other configuration materialises it first - same result either way):
 *   dmu_total = (dmu ? dmu : 0) + d loss(kind; mu, gt) / d mu * dloss[0]
 * kind: MLHOT_LOSS_* with a gradient (0 azimuth, 1 mse, 2 quaternion, 4 distractor); gt[T*Nq, gt_dim]; dloss: device scalar. */
typedef struct { int kind; const float* gt; int gt_dim; const float* dloss;
                 float* value;   /* ABI 5: NULL, or where this call also leaves the loss VALUE (what mlhot_loss_fwd(kind, mu, gt) writes, same
                                  * bits): one more workgroup of the first backward kernel instead of a launch between forward and backward */
} mlhot_loss_desc;
int mlhot_np_vanilla_bwd_loss(const mlhot_np_dims* d, const mlhot_np_params* p,
                              const float* ctx_x, const float* ctx_y, const float* qry_x,
                              const float* mu, const float* dmu, const mlhot_loss_desc* loss, const mlhot_np_grads* g,
                              const void* saved, void* scratch, size_t scratch_bytes, void* stream);
/* Staged variants (see "strict sharded parity" above).  Built on the fused attention tail's launch boundaries: attention
 * aggregation with Nc, Nq <= 16 (every shipped ANP configuration); MLHOT_ERR_UNSUPPORTED otherwise. */
int mlhot_np_vanilla_fwd_staged(const mlhot_np_dims* d, const mlhot_np_params* p,
                                const float* ctx_x, const float* ctx_y, const float* qry_x, float* mu,
                                void* saved, void* scratch, size_t scratch_bytes, int stage, float* xchg, void* stream);
int mlhot_np_vanilla_bwd_staged(const mlhot_np_dims* d, const mlhot_np_params* p,
                                const float* ctx_x, const float* ctx_y, const float* qry_x,
                                const float* mu, const float* dmu, const mlhot_np_grads* g,
                                const void* saved, void* scratch, size_t scratch_bytes, int stage, float* xchg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLHOT_H */
