// Element-parallel functors of the hot path (run by run_foreach / run_reduce1).
#pragma once
#include "common.h"
#include "problems.h"

namespace mlhot {

// ------------------------------------------------------------------------------------------
// conv1 of the vanilla encoder: 1 -> 32 channels, 3x3 s2 p1 + ReLU on 128x128 images
// (ANPShapeNet1D.py:47-48).  K = 9 is far too short for the matrix cores and the layer is
// bound by its 512 KiB/image output write, so one lane owns one output pixel: the 9 taps
// stay in registers, weights come through the scalar cache (uniform index), and the 32
// channel-plane stores are each a full 256-B line per wave.
// ------------------------------------------------------------------------------------------
template <class XS>
struct Conv1Fwd {
  XS x;             // [n][1][128][128]
  const float* w;   // [32][9]
  const float* b;   // [32]
  float* y;         // [n][32][64][64]
  MLHOT_HD void operator()(size_t i) const {
    const int ox = (int)(i & 63), oy = (int)((i >> 6) & 63), img = (int)(i >> 12);
    const float* xi = x.img(img);
    float t[9];
#ifndef MLHOT_HOSTSIM
#pragma unroll
#endif
    for (int ky = 0; ky < 3; ++ky)
#ifndef MLHOT_HOSTSIM
#pragma unroll
#endif
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
        t[ky * 3 + kx] = (iy >= 0 && ix >= 0) ? xi[iy * 128 + ix] : 0.f;   // iy, ix <= 127 always
      }
    float* yo = y + (size_t)img * 32 * 4096 + oy * 64 + ox;
    for (int co = 0; co < 32; ++co) {
      float s = 0.f;
#ifndef MLHOT_HOSTSIM
#pragma unroll
#endif
      for (int q = 0; q < 9; ++q) s = fmaf(t[q], w[co * 9 + q], s);
      s += b[co];
      yo[(size_t)co * 4096] = s > 0.f ? s : 0.f;
    }
  }
};

// MaxPool2d((2,2)) on a post-ReLU map [n][C][H][W] -> [n][C][H/2][W/2] + window arg-max
// (first maximum in scan order (0,0),(0,1),(1,0),(1,1), as ATen's max_pool2d).
struct Pool2Fwd {
  const float* a; float* p; uint8_t* amax; int H, W;   // H, W: input size
  MLHOT_HD void operator()(size_t i) const {
    const int wo = W / 2, ho = H / 2;
    const int px = (int)(i % wo), py = (int)((i / wo) % ho);
    const size_t plane = i / ((size_t)wo * ho);
    const float* s = a + plane * H * W + (size_t)(2 * py) * W + 2 * px;
    float best = s[0]; int which = 0;
    if (s[1] > best) { best = s[1]; which = 1; }
    if (s[W] > best) { best = s[W]; which = 2; }
    if (s[W + 1] > best) { best = s[W + 1]; which = 3; }
    p[i] = best; amax[i] = (uint8_t)which;
  }
};

// residual join of a BasicBlock: y = relu(a + b)  (ResNet.py:69-72) and its backward g = dy * (y > 0)
struct AddRelu { const float* a; const float* b; float* y; MLHOT_HD void operator()(size_t i) const { const float v = a[i] + b[i]; y[i] = v > 0.f ? v : 0.f; } };
// y = a + alpha * x  (a may be null: y = alpha * x), product and sum rounded separately like the two torch operators they replace
// (trainer/model_trainer.py:77-78: loss + kl * beta)
struct Axpy {
  const float* a; const float* x; float alpha; float* y;
  MLHOT_HD void operator()(size_t i) const {
#pragma clang fp contract(off)      // (the *_rn intrinsics alone do not keep hipcc from fusing the two: see LossPlusRed)
#ifdef MLHOT_HOSTSIM
    volatile float p = alpha * x[i];
#else
    const float p = alpha * x[i];
#endif
    y[i] = a ? a[i] + p : p;
  }
};
struct AddReluBwd { const float* y; const float* dy; float* g; MLHOT_HD void operator()(size_t i) const { g[i] = y[i] > 0.f ? dy[i] : 0.f; } };
// max-pool backward: route dp to the window arg-max (dx must be pre-zeroed by construction: every element written)
struct Pool2Bwd {
  const float* dp; const uint8_t* amax; float* dx; int H, W;   // H, W: input size
  MLHOT_HD void operator()(size_t i) const {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const size_t plane = i / ((size_t)W * H);
    const size_t o = plane * (H / 2) * (W / 2) + (size_t)(y >> 1) * (W / 2) + (x >> 1);
    dx[i] = amax[o] == (((y & 1) << 1) | (x & 1)) ? dp[o] : 0.f;
  }
};
// Bayes-by-backprop weight sample W = mu + eps * softplus(rho) and its KL term per element
// (bbb/BBBConv.py:86-108; calculate_kl is called as (mu_q=0, sig_q=0.1, mu_p=mu, sig_p=sigma)):
//   kl_i = 0.5 * (2 log(sigma/0.1) - 1 + (0.1/sigma)^2 + (mu/sigma)^2)
struct BbbSample {
  const float* mu; const float* rho; const float* eps; float* w; float* klterm;
  MLHOT_HD void operator()(size_t i) const {
    const float sg = log1pf(expf(rho[i]));
    w[i] = mu[i] + eps[i] * sg;
    const float q = 0.1f / sg, z = mu[i] / sg;
    klterm[i] = 0.5f * (2.f * logf(sg / 0.1f) - 1.f + q * q + z * z);
  }
};
struct BbbSampleBwd {   // dmu = dw + dkl * mu/sigma^2 ; drho = (dw*eps + dkl * (1/sigma - 0.01/sigma^3 - mu^2/sigma^3)) * sigmoid(rho)
  const float* mu; const float* rho; const float* eps; const float* dw; const float* dkl; float* dmu; float* drho;
  MLHOT_HD void operator()(size_t i) const {
    const float sg = log1pf(expf(rho[i])), g = dkl[0];
    const float inv = 1.f / sg;
    dmu[i] = dw[i] + g * mu[i] * inv * inv;
    const float dsig = dw[i] * eps[i] + g * (inv - 0.01f * inv * inv * inv - mu[i] * mu[i] * inv * inv * inv);
    drho[i] = dsig / (1.f + expf(-rho[i]));
  }
};

// ------------------------------------------------------------------------------------------
// Optimizer step over ONE flat buffer (train.py:52-56 builds torch.optim.Adam over ~70 small tensors;
// with the flat gradient layout the whole update is a single launch).  torch.optim.Adam semantics
// (amsgrad off, L2 weight decay added to the gradient): step t >= 1,
//   g = grad_scale * grad + wd * p;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// ------------------------------------------------------------------------------------------
struct AdamStep {
  float* p; const float* g; float* m; float* v;
  float b1, b2, eps, wd, grad_scale, step_size, inv_sqrt_bc2;
  MLHOT_HD void operator()(size_t i) const {
    const float gi = grad_scale * g[i] + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
};

// The same update with the step count living ON THE DEVICE (incremented by CounterInc right before): a captured hipGraph of
// a training step then advances Adam's bias correction on every replay.  The corrections are evaluated in double like the
// host-side variant, so both produce the same step sizes.
struct CounterInc { int* c; MLHOT_HD void operator()(size_t) const { c[0] += 1; } };
struct AdamStepCounter {
  float* p; const float* g; float* m; float* v;
  float b1, b2, eps, wd, grad_scale, lr; const int* step;
  MLHOT_HD void operator()(size_t i) const {
    const int t = step[0];
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, (double)t)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)t)));
    const float gi = grad_scale * g[i] + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
};

// ------------------------------------------------------------------------------------------
// X1: train-mode batch norm of ConvEmbeddingModel (conv_embedding_model.py:113-117): the batch is the
// shots of ONE task; statistics per channel over (n, h, w); F.batch_norm(training=True) also updates the
// running buffers in place with momentum 0.1 and the UNBIASED variance.
// ------------------------------------------------------------------------------------------
struct Pair2 { float a, b; };
struct BnStats {     // per channel: mean and biased variance (two-pass: sum, then centred squares via sum/sumsq in fp32 is avoided)
  typedef Pair2 T;
  const float* x; int C, HW; float* mean;    // pass 1: mean
  MLHOT_HD T identity() const { return T{0.f, 0.f}; }
  MLHOT_HD T load(int c, int i) const { const int n = i / HW, r = i % HW; return T{x[((size_t)n * C + c) * HW + r], 0.f}; }
  MLHOT_HD T combine(T u, T v) const { return T{u.a + v.a, 0.f}; }
  int count;
  MLHOT_HD void finish(int c, T s) const { mean[c] = s.a / (float)count; }
};
struct BnVar {       // pass 2: biased variance around the mean; updates the running buffers
  typedef Pair2 T;
  const float* x; int C, HW; const float* mean; float* var; float* run_mean; float* run_var; float momentum; int count;
  MLHOT_HD T identity() const { return T{0.f, 0.f}; }
  MLHOT_HD T load(int c, int i) const { const int n = i / HW, r = i % HW; const float d = x[((size_t)n * C + c) * HW + r] - mean[c]; return T{d * d, 0.f}; }
  MLHOT_HD T combine(T u, T v) const { return T{u.a + v.a, 0.f}; }
  MLHOT_HD void finish(int c, T s) const {
    var[c] = s.a / (float)count;
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (count > 1 ? s.a / (float)(count - 1) : var[c]);
    }
  }
};
// y = relu(gamma * (x - mean) / sqrt(var + eps) + beta)
struct BnApplyRelu {
  const float* x; const float* mean; const float* var; const float* gamma; const float* beta; float eps; int C, HW; float* y;
  MLHOT_HD void operator()(size_t i) const {
    const int c = (int)((i / HW) % C);
    const float v = gamma[c] * (x[i] - mean[c]) / sqrtf(var[c] + eps) + beta[c];
    y[i] = v > 0.f ? v : 0.f;
  }
};
// backward sums per channel: a = sum g, b = sum g * xhat, with g = dy * (y > 0)
struct BnBwdSums {
  typedef Pair2 T;
  const float* x; const float* y; const float* dy; const float* mean; const float* var; float eps; int C, HW;
  float* dgamma; float* dbeta;
  MLHOT_HD T identity() const { return T{0.f, 0.f}; }
  MLHOT_HD T load(int c, int i) const {
    const int n = i / HW, r = i % HW; const size_t o = ((size_t)n * C + c) * HW + r;
    const float g = y[o] > 0.f ? dy[o] : 0.f;
    return T{g, g * (x[o] - mean[c]) / sqrtf(var[c] + eps)};
  }
  MLHOT_HD T combine(T u, T v) const { return T{u.a + v.a, u.b + v.b}; }
  MLHOT_HD void finish(int c, T s) const { dbeta[c] = s.a; dgamma[c] = s.b; }
};
// dx = gamma / sqrt(var+eps) * (g - mean(g) - xhat * mean(g * xhat))
struct BnBwdApply {
  const float* x; const float* y; const float* dy; const float* mean; const float* var; const float* gamma;
  const float* dgamma; const float* dbeta; float eps; int C, HW, count; float* dx;
  MLHOT_HD void operator()(size_t i) const {
    const int c = (int)((i / HW) % C);
    const float inv = 1.f / sqrtf(var[c] + eps), xh = (x[i] - mean[c]) * inv;
    const float g = y[i] > 0.f ? dy[i] : 0.f;
    dx[i] = gamma[c] * inv * (g - dbeta[c] / (float)count - xh * dgamma[c] / (float)count);
  }
};
// spatial mean over HW: [n][C][HW] -> [n][C], and its backward
struct SpatialMean { const float* x; int HW; float* y; MLHOT_HD void operator()(size_t i) const { float s = 0.f; for (int r = 0; r < HW; ++r) s += x[i * HW + r]; y[i] = s / (float)HW; } };
struct SpatialMeanBwd { const float* dy; int HW; float* dx; MLHOT_HD void operator()(size_t i) const { dx[i] = dy[i / HW] / (float)HW; } };

struct FillF { float* p; float v; MLHOT_HD void operator()(size_t i) const { p[i] = v; } };

// strided 2-D fill: rows x cols window of a [rows][ld] matrix
struct Fill2D {
  float* p; int ld, cols; float v;
  MLHOT_HD void operator()(size_t i) const { p[(i / cols) * ld + (i % cols)] = v; }
};

// ------------------------------------------------------------------------------------------
// G1 aggregators over the shot axis.  One lane per (task, feature): lanes of a wave walk the
// contiguous feature axis, so every shot row is read as full lines and the reduction over the
// (<= 30) shots is a register loop.
// ------------------------------------------------------------------------------------------
MLHOT_HD float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }     // F.softplus, beta=1, threshold=20
MLHOT_HD float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

struct AggFwd {
  int mode, Nc, R;
  const float* rs; const float* lv; float* r; float* sigma; int32_t* amax;
  MLHOT_HD void operator()(size_t i) const {
    const int j = (int)(i % R); const size_t t = i / R;
    const float* src = rs + t * Nc * R + j;
    if (mode == 0) {
      float s = 0.f;
      for (int n = 0; n < Nc; ++n) s += src[(size_t)n * R];
      r[i] = s / (float)Nc;
    } else if (mode == 1) {
      float best = src[0]; int arg = 0;
      for (int n = 1; n < Nc; ++n) { const float v = src[(size_t)n * R]; if (v > best) { best = v; arg = n; } }
      r[i] = best; amax[i] = arg;
    } else {
      const float* lsrc = lv + t * Nc * R + j;
      float s1 = 1.f, s2 = 0.f;   // prior: mu_z = 0, sigma_z = 1  (CNPShapeNet1D.py:87-88)
      for (int n = 0; n < Nc; ++n) {
        const float iv = 1.f / (1e-5f + softplus_f(lsrc[(size_t)n * R]));
        s1 += iv; s2 += iv * src[(size_t)n * R];
      }
      const float sz = 1.f / s1;
      sigma[i] = sz; r[i] = sz * s2;
    }
  }
};

struct AggBwd {
  int mode, Nc, R;
  const float* rs; const float* lv; const float* r; const float* sigma; const int32_t* amax;
  const float* dr; float* drs; float* dlv;
  MLHOT_HD void operator()(size_t i) const {
    const int j = (int)(i % R); const size_t t = i / R;
    float* dst = drs + t * Nc * R + j;
    const float g = dr[i];
    if (mode == 0) {
      const float v = g / (float)Nc;
      for (int n = 0; n < Nc; ++n) dst[(size_t)n * R] = v;
    } else if (mode == 1) {
      const int arg = amax[i];
      for (int n = 0; n < Nc; ++n) dst[(size_t)n * R] = (n == arg) ? g : 0.f;
    } else {
      const float* src = rs + t * Nc * R + j;
      const float* lsrc = lv + t * Nc * R + j;
      float* ldst = dlv + t * Nc * R + j;
      const float sz = sigma[i], mz = r[i];
      for (int n = 0; n < Nc; ++n) {
        const float l = lsrc[(size_t)n * R];
        const float var = 1e-5f + softplus_f(l);
        const float mu = src[(size_t)n * R];
        dst[(size_t)n * R] = g * sz / var;
        // d mu_z / d var_n = -(mu_n - mu_z) * sigma_z / var_n^2 ; d var / d lv = sigmoid(lv)
        ldst[(size_t)n * R] = -g * (mu - mz) * sz / (var * var) * (l > 20.f ? 1.f : sigmoid_f(l));
      }
    }
  }
};

// z[t][dim_z] broadcast over the Nq target rows of the decoder input (…[:, None, :].repeat)
struct BcastRows {
  const float* z; int dz, Nq; float* dst; int ld;
  MLHOT_HD void operator()(size_t i) const {
    const int j = (int)(i % dz); const size_t row = i / dz;   // row = t*Nq + n
    dst[row * ld + j] = z[(row / Nq) * dz + j];
  }
};
// its backward: dz[t][j] = sum_n d dst[(t,n)][j]
struct BcastRowsBwd {
  const float* dsrc; int ld, dz, Nq; float* dzt;
  MLHOT_HD void operator()(size_t i) const {
    const int j = (int)(i % dz); const size_t t = i / dz;
    float s = 0.f;
    for (int n = 0; n < Nq; ++n) s += dsrc[(t * Nq + n) * ld + j];
    dzt[i] = s;
  }
};

// ------------------------------------------------------------------------------------------
// L1 losses (trainer/losses.py:32-80)
// ------------------------------------------------------------------------------------------
struct LossRed {
  typedef float T;
  int kind, y_dim, gt_dim, rows;
  const float* mu; const float* gt; float* out;
  MLHOT_HD float identity() const { return 0.f; }
  MLHOT_HD float combine(float a, float b) const { return a + b; }
  MLHOT_HD float load(int r) const {
    const float* m = mu + (size_t)r * y_dim; const float* g = gt + (size_t)r * gt_dim;
    if (kind == 0 || kind == 1) {                       // azimuth (first 2 labels) / plain MSE
      float s = 0.f;
      for (int j = 0; j < y_dim; ++j) { const float d = g[j] - m[j]; s += d * d; }
      return s;
    }
    if (kind == 2) {                                    // quaternion: min(L1(q - q^), L1(-q - q^)), q^ normalised
      float nn = 0.f;
      for (int j = 0; j < y_dim; ++j) nn += m[j] * m[j];
      nn = sqrtf(nn);
      float p = 0.f, q = 0.f;
      for (int j = 0; j < y_dim; ++j) { const float u = m[j] / nn; p += fabsf(g[j] - u); q += fabsf(-g[j] - u); }
      return p < q ? p : q;
    }
    if (kind == 3) {                                    // degree error (test-time, shapenet_1d)
      const float kRad2Deg = 57.29577951308232f;
      const float gd = g[gt_dim - 1] * kRad2Deg;
      float ang = acosf(m[0]);
      if (m[1] < 0.f) ang = 6.283185307179586f - ang;
      const float pd = ang * kRad2Deg;
      float e = fabsf(gd - pd);
      const float e2 = fabsf(gd + 360.f - pd), e3 = fabsf(gd - (pd + 360.f));
      if (e2 < e) e = e2;
      if (e3 < e) e = e3;
      return e;
    }
    float s = 0.f;                                      // distractor: L2 distance
    for (int j = 0; j < y_dim; ++j) { const float d = g[j] - m[j]; s += d * d; }
    return sqrtf(s);
  }
  MLHOT_HD void finish(float s) const { out[0] = s / (float)(kind == 1 ? rows * y_dim : rows); }
};
// loss + alpha * x[0] in the loss's own launch (trainer/model_trainer.py:77-78: `losses = loss + kl * beta`): the loss value as LossRed
// leaves it, then the product and the sum rounded separately - the bits of mlhot_loss_fwd followed by mlhot_axpy.
struct LossPlusRed : LossRed {
  const float* x; float alpha; float* total;
  MLHOT_HD void finish(float s) const {
#pragma clang fp contract(off)      // product and sum rounded separately: hipcc's __fadd_rn(l, __fmul_rn(..)) came out as ONE v_fma here (a last-bit difference to mlhot_axpy, caught by the parity test)
    const float l = s / (float)(kind == 1 ? rows * y_dim : rows);
    if (out != nullptr) out[0] = l;
#ifdef MLHOT_HOSTSIM
    volatile float pr = alpha * x[0];
#else
    const float pr = alpha * x[0];
#endif
    total[0] = l + pr;
  }
};

// d loss / d mu of ONE row (all y_dim <= 8 entries into d[]), scaled by the upstream scalar `up`
MLHOT_HD inline void loss_row_grad(int kind, int y_dim, int rows, const float* m, const float* g, float up, float* d) {
  if (kind == 0 || kind == 1) {
    const float s = up * 2.f / (float)(kind == 1 ? rows * y_dim : rows);
    for (int j = 0; j < y_dim; ++j) d[j] = s * (m[j] - g[j]);
  } else if (kind == 2) {
    float nn = 0.f;
    for (int j = 0; j < y_dim; ++j) nn += m[j] * m[j];
    nn = sqrtf(nn);
    float p = 0.f, q = 0.f;
    for (int j = 0; j < y_dim; ++j) { const float u = m[j] / nn; p += fabsf(g[j] - u); q += fabsf(-g[j] - u); }
    const float sgn = (p <= q) ? 1.f : -1.f;          // torch.minimum routes ties to ... measure zero
    // dL/du_j = -sign(sgn*g_j - u_j) / rows ;  u = m/|m|  =>  dm = (du - u (u.du)) / |m|
    float du[8]; float dot = 0.f;
    for (int j = 0; j < y_dim; ++j) {
      const float u = m[j] / nn, e = sgn * g[j] - u;
      du[j] = -(e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * up / (float)rows;
      dot += du[j] * u;
    }
    for (int j = 0; j < y_dim; ++j) d[j] = (du[j] - (m[j] / nn) * dot) / nn;
  } else if (kind == 4) {
    float s = 0.f;
    for (int j = 0; j < y_dim; ++j) { const float e = g[j] - m[j]; s += e * e; }
    s = sqrtf(s);
    for (int j = 0; j < y_dim; ++j) d[j] = up * (m[j] - g[j]) / (s * (float)rows);
  } else {
    for (int j = 0; j < y_dim; ++j) d[j] = 0.f;       // degree error is evaluation-only
  }
}

struct LossBwd {
  int kind, y_dim, gt_dim, rows;
  const float* mu; const float* gt; const float* dloss; float* dmu;
  const float* add;          // optional: a gradient mu received from somewhere else, added to the loss's (same shape as dmu)
  MLHOT_HD void operator()(size_t r) const {
    float d[8];
    loss_row_grad(kind, y_dim, rows, mu + r * y_dim, gt + r * gt_dim, dloss[0], d);
    for (int j = 0; j < y_dim; ++j) dmu[r * y_dim + j] = add != nullptr ? add[r * y_dim + j] + d[j] : d[j];
  }
};
// ... and the backward of `loss + alpha * x`: d mu as LossBwd, d x = alpha * d total by row 0's thread (mlhot_axpy(NULL, dtotal, alpha)'s bits)
struct LossPlusBwd : LossBwd {
  float alpha; float* dx;
  MLHOT_HD void operator()(size_t r) const {
    LossBwd::operator()(r);
    if (r == 0 && dx != nullptr) {
#ifdef MLHOT_HOSTSIM
      volatile float pr = alpha * dloss[0];
      dx[0] = pr;
#else
      dx[0] = __fmul_rn(alpha, dloss[0]);
#endif
    }
  }
};

// The loss whose gradient a model's backward takes itself (mlhot_np_vanilla_bwd_loss): kind < 0 = none
struct LossDesc { int kind; const float* gt; int gt_dim; const float* dloss; float* value; };

#ifndef MLHOT_HOSTSIM
// The loss VALUE by one extra workgroup of the model's first backward kernel (LossDesc.value != nullptr): nothing in the backward
// reads the value, but as a launch of its own the reduction sits between the forward's last kernel and the backward's first (4.9 us
// of c3's step + a kernel boundary; as a forked graph branch it cost the step +22 us).  reduce1_kernel<LossRed>'s arithmetic thread for
// thread - virtual thread v of 1024 sums rows v, v + 1024, ... into the same four accumulators, then the same LDS tree - so the
// value has the bits of mlhot_loss_fwd's.  sm: 1024 floats of LDS; any block size.
__device__ __forceinline__ void loss_value_block(const LossRed& r, int n, float* sm) {
  const int nt = (int)blockDim.x, tid = (int)threadIdx.x;
  for (int v = tid; v < 1024; v += nt) {
    float a0 = r.identity(), a1 = r.identity(), a2 = r.identity(), a3 = r.identity();
    int i = v;
    for (; i + 3072 < n; i += 4096) {
      a0 = r.combine(a0, r.load(i)); a1 = r.combine(a1, r.load(i + 1024));
      a2 = r.combine(a2, r.load(i + 2048)); a3 = r.combine(a3, r.load(i + 3072));
    }
    for (; i < n; i += 1024) a0 = r.combine(a0, r.load(i));
    sm[v] = r.combine(r.combine(a0, a1), r.combine(a2, a3));
  }
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    for (int v = tid; v < s; v += nt) sm[v] = r.combine(sm[v], sm[v + s]);
    __syncthreads();
  }
  if (tid == 0) r.finish(sm[0]);
}
#endif

}  // namespace mlhot
