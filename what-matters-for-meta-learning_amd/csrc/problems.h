// Implicit-GEMM problem functors for the CNP/ANP hot path (see igemm.h for the contract).
// All tensors are fp32, images/activations NCHW contiguous, conv weights [Cout][Cin][3][3],
// linear weights [out][in] - the reference's own state_dict layouts, so no repacking.
#pragma once
#include "common.h"

namespace mlhot {

// ------------------------------------------------------------------------------------------
// Image-batch sources: one contiguous batch, or two segments (context | target images).
// ------------------------------------------------------------------------------------------
struct Src1 {
  const float* p; size_t stride;
  MLHOT_HD const float* img(int i) const { return p + (size_t)i * stride; }
};
struct Src2 {
  const float* p0; int n0; const float* p1; size_t stride;
  MLHOT_HD const float* img(int i) const { return i < n0 ? p0 + (size_t)i * stride : p1 + (size_t)(i - n0) * stride; }
};

// ------------------------------------------------------------------------------------------
// dY sources for the conv backward problems
// ------------------------------------------------------------------------------------------

// Plain gradient tensor [N][C][H][W].
template <int C, int H, int W>
struct DyPlain {
  const float* dy;
  MLHOT_HD float at(int img, int c, int y, int x) const { return dy[(((size_t)img * C + c) * H + y) * W + x]; }
};

// Gradient w.r.t. the PRE-pool, PRE-ReLU conv output, routed on the fly from the pooled
// gradient: MaxPool2d((2,2)) sends dp[c][y/2][x/2] to the window's arg-max (first maximum in
// window scan order, like ATen), and ReLU kills it where the pooled value is not positive
// (max(relu(a)) = relu(max(a)), so one test of the pooled output covers both).
template <int C, int H, int W>  // H, W: PRE-pool size
struct DyPooled {
  const float* dp;       // [N][C][H/2][W/2]
  const float* pooled;   // [N][C][H/2][W/2]  forward output of relu+pool
  const uint8_t* amax;   // [N][C][H/2][W/2]  2*dy+dx of the arg-max
  MLHOT_HD float at(int img, int c, int y, int x) const {
    const size_t o = (((size_t)img * C + c) * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1);
    const int which = ((y & 1) << 1) | (x & 1);
    return (amax[o] == which && pooled[o] > 0.f) ? dp[o] : 0.f;
  }
};

// ------------------------------------------------------------------------------------------
// Convolution 3x3, stride 2, padding 1 (every conv of the vanilla encoder; A.1)
// ------------------------------------------------------------------------------------------

// forward:  M = Nimg*HO*WO output positions, N = COUT, K = CIN*9 (k = ci*9 + ky*3 + kx)
// epilogue: + bias, ReLU, store NCHW.
template <int CIN, int HIN, int WIN, int COUT, class XS>
struct ConvFwd {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = true;
  static constexpr int HO = HIN / 2, WO = WIN / 2;
  int M, N, K;
  XS x;             // [Nimg][CIN][HIN][WIN]
  const float* w;   // [COUT][CIN*9]
  const float* b;   // [COUT]
  float* y;         // [Nimg][COUT][HO][WO]  (post-ReLU)
  MLHOT_HD float A(int m, int k) const {
    const int img = m / (HO * WO), r = m % (HO * WO), oy = r / WO, ox = r % WO;
    const int ci = k / 9, t = k % 9, ky = t / 3, kx = t % 3;
    const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
    if (iy < 0 || iy >= HIN || ix < 0 || ix >= WIN) return 0.f;
    return x.img(img)[((size_t)ci * HIN + iy) * WIN + ix];
  }
  MLHOT_HD float B(int k, int n) const { return w[(size_t)n * K + k]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const int img = m / (HO * WO), r = m % (HO * WO);
    v += b[n];
    y[((size_t)img * COUT + n) * (HO * WO) + r] = v > 0.f ? v : 0.f;
  }
};

// data gradient for ONE parity class (PY, PX) of input positions y = 2y'+PY, x = 2x'+PX.
// With k=3, s=2, p=1 an even coordinate is reached only by tap 1 (from output y'), an odd one
// by tap 0 (from y'+1) and tap 2 (from y').  M = Nimg*HO*WO, N = CIN, K = taps*COUT.
// epilogue: multiply by the ReLU mask of the layer input (act > 0) and store.
template <int CIN, int HIN, int WIN, int COUT, int PY, int PX, class DY>
struct ConvDgrad {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = false;
  static constexpr int HO = HIN / 2, WO = WIN / 2;
  static constexpr int NTY = PY ? 2 : 1, NTX = PX ? 2 : 1;
  int M, N, K;
  DY dy;
  const float* w;     // [COUT][CIN][3][3]
  const float* act;   // [Nimg][CIN][HIN][WIN] post-ReLU layer input, or nullptr (no mask)
  float* dx;          // [Nimg][CIN][HIN][WIN]
  MLHOT_HD float A(int m, int k) const {
    const int img = m / (HO * WO), r = m % (HO * WO), yp = r / WO, xp = r % WO;
    const int co = k % COUT, tt = k / COUT, tx = tt % NTX, ty = tt / NTX;
    const int oy = (PY && ty == 0) ? yp + 1 : yp, ox = (PX && tx == 0) ? xp + 1 : xp;
    if (oy >= HO || ox >= WO) return 0.f;
    return dy.at(img, co, oy, ox);
  }
  MLHOT_HD float B(int k, int n) const {
    const int co = k % COUT, tt = k / COUT, tx = tt % NTX, ty = tt / NTX;
    const int ky = PY ? (ty == 0 ? 0 : 2) : 1, kx = PX ? (tx == 0 ? 0 : 2) : 1;
    return w[(((size_t)co * CIN + n) * 3 + ky) * 3 + kx];
  }
  MLHOT_HD void store(int m, int n, float v) const {
    const int img = m / (HO * WO), r = m % (HO * WO), yp = r / WO, xp = r % WO;
    const size_t o = (((size_t)img * CIN + n) * HIN + 2 * yp + PY) * WIN + 2 * xp + PX;
    dx[o] = (act == nullptr || act[o] > 0.f) ? v : 0.f;
  }
};

// weight + bias gradient:  out[co][n] = sum_pos dY[co][pos] * Xcol[pos][n], with one extra
// all-ones column n = CIN*9 that yields the bias gradient.  M = COUT, N = CIN*9+1,
// K = Nimg*HO*WO (split-K over positions).
template <int CIN, int HIN, int WIN, int COUT, class DY, class XS>
struct ConvWgrad {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = true;
  static constexpr int HO = HIN / 2, WO = WIN / 2;
  int M, N, K;
  DY dy;
  XS x;             // [Nimg][CIN][HIN][WIN]
  float* dw;        // [COUT][CIN*9]
  float* db;        // [COUT]
  MLHOT_HD float A(int m, int k) const {
    const int img = k / (HO * WO), r = k % (HO * WO);
    return dy.at(img, m, r / WO, r % WO);
  }
  MLHOT_HD float B(int k, int n) const {
    if (n == CIN * 9) return 1.f;
    const int img = k / (HO * WO), r = k % (HO * WO), oy = r / WO, ox = r % WO;
    const int ci = n / 9, t = n % 9, ky = t / 3, kx = t % 3;
    const int iy = 2 * oy + ky - 1, ix = 2 * ox + kx - 1;
    if (iy < 0 || iy >= HIN || ix < 0 || ix >= WIN) return 0.f;
    return x.img(img)[((size_t)ci * HIN + iy) * WIN + ix];
  }
  MLHOT_HD void store(int m, int n, float v) const {
    if (n == CIN * 9) db[m] = v;
    else dw[(size_t)m * (CIN * 9) + n] = v;
  }
};

// ------------------------------------------------------------------------------------------
// Run-time-shaped convolution (any k, stride, padding; NCHW) for the ResNet encoders of the
// ShapeNet3D / Distractor models: 5x5 s2 p2 stem, 3x3 s2 / s1 p1 block convs, 1x1 s2 (3x3 in the
// BBB twin) skip (networks/models.py:87-115, networks/ResNet.py:25-74).  Same three GEMM views
// as the templated problems above; shapes are ordinary members so one instantiation serves every
// layer (index arithmetic costs more - these layers are functional in round 1, not yet tuned).
// ------------------------------------------------------------------------------------------
struct ConvShape {
  int N, Cin, H, W, Cout, k, s, p, HO, WO;
};

struct ConvFwdRT {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = true;
  int M, N, K;
  ConvShape c;
  const float* x; const float* w; const float* b; float* y; int relu;
  MLHOT_HD float A(int m, int k) const {
    const int hw = c.HO * c.WO, img = m / hw, r = m % hw, oy = r / c.WO, ox = r % c.WO;
    const int kk = c.k * c.k, ci = k / kk, t = k % kk, ky = t / c.k, kx = t % c.k;
    const int iy = oy * c.s + ky - c.p, ix = ox * c.s + kx - c.p;
    if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) return 0.f;
    return x[(((size_t)img * c.Cin + ci) * c.H + iy) * c.W + ix];
  }
  MLHOT_HD float B(int k, int n) const { return w[(size_t)n * K + k]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const int hw = c.HO * c.WO, img = m / hw, r = m % hw;
    if (b) v += b[n];
    if (relu && v < 0.f) v = 0.f;
    y[((size_t)img * c.Cout + n) * hw + r] = v;
  }
  // hoisted index parts (igemm.h): same values as A / B above
  struct alignas(16) KEnt { int off, ky, kx, pad; };   // k -> (ci, ky, kx): ci*H*W + ky*W + kx (one 16-byte LDS read)
  struct RCtx { int base, iy0, ix0; };         // m -> (img, oy, ox): img*Cin*H*W + iy0*W + ix0
  struct CCtx { int woff; };
  MLHOT_HD KEnt kent(int k) const {
    const int kk = c.k * c.k, ci = k / kk, t = k % kk, ky = t / c.k, kx = t % c.k;
    return KEnt{(ci * c.H + ky) * c.W + kx, ky, kx, 0};
  }
  MLHOT_HD RCtx rctx(int m) const {
    const int hw = c.HO * c.WO, img = m / hw, r = m % hw, oy = r / c.WO, ox = r % c.WO;
    const int iy0 = oy * c.s - c.p, ix0 = ox * c.s - c.p;
    return RCtx{img * c.Cin * c.H * c.W + iy0 * c.W + ix0, iy0, ix0};
  }
  MLHOT_HD CCtx cctx(int n) const { return CCtx{n * K}; }
#ifndef MLHOT_HOSTSIM
  // A2 / B2 are branch-free AND select-free on purpose: a masked element (tile bounds `ok`, padding) loads the zero word of
  // g_zero_one instead, so the gathers of a tile issue back to back and nothing waits on them until the tile is stashed
  typedef float ARaw; typedef float BRaw;
  MLHOT_DEV float A2(const RCtx& r, const KEnt& e, int, int, bool ok) const {
    const int iy = r.iy0 + e.ky, ix = r.ix0 + e.kx;
    const bool in = ok & ((unsigned)iy < (unsigned)c.H) & ((unsigned)ix < (unsigned)c.W);
    return load_or_const(x, r.base + e.off, in);
  }
  MLHOT_DEV float B2(const KEnt&, const CCtx& cc, int k, int, bool ok) const {
    return load_or_const(w, cc.woff + k, ok);
  }
  MLHOT_DEV float Afin(float v) const { return v; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

// data gradient for the class of input positions y = s*y' + py, x = s*x' + px: only taps with
// ky = ky0 + s*ty (ky0 = (py + p) mod s) reach them, from output row oy = (y + p - ky) / s.
struct ConvDgradRT {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = false;
  int M, N, K;
  ConvShape c;
  int py, px, ky0, kx0, nty, ntx, ny, nx;     // class geometry (ny x nx positions per image)
  const float* dy; const float* yact;         // yact: forward output, for the ReLU mask (or nullptr)
  const float* w; float* dx;
  MLHOT_HD float A(int m, int k) const {
    const int img = m / (ny * nx), r = m % (ny * nx), yp = r / nx, xp = r % nx;
    const int co = k % c.Cout, tt = k / c.Cout, tx = tt % ntx, ty = tt / ntx;
    const int ky = ky0 + c.s * ty, kx = kx0 + c.s * tx;
    const int oy = (c.s * yp + py + c.p - ky) / c.s, ox = (c.s * xp + px + c.p - kx) / c.s;
    if (oy < 0 || oy >= c.HO || ox < 0 || ox >= c.WO) return 0.f;
    const size_t o = (((size_t)img * c.Cout + co) * c.HO + oy) * c.WO + ox;
    return (yact == nullptr || yact[o] > 0.f) ? dy[o] : 0.f;
  }
  MLHOT_HD float B(int k, int n) const {
    const int co = k % c.Cout, tt = k / c.Cout, tx = tt % ntx, ty = tt / ntx;
    return w[(((size_t)co * c.Cin + n) * c.k + ky0 + c.s * ty) * c.k + kx0 + c.s * tx];
  }
  MLHOT_HD void store(int m, int n, float v) const {
    const int img = m / (ny * nx), r = m % (ny * nx), yp = r / nx, xp = r % nx;
    dx[(((size_t)img * c.Cin + n) * c.H + c.s * yp + py) * c.W + c.s * xp + px] = v;
  }
  // hoisted index parts (igemm.h)
  struct alignas(16) KEnt { int co, ky, kx, woff; };   // k -> (co, ty, tx); woff = (co*Cin*k + ky)*k + kx
  struct RCtx { int ibase, yb, xb; };           // m -> (img, y', x'): img*Cout*HO*WO, s*y'+py+p, s*x'+px+p
  struct CCtx { int noff; };                    // n*k*k
  MLHOT_HD KEnt kent(int k) const {
    const int co = k % c.Cout, tt = k / c.Cout, tx = tt % ntx, ty = tt / ntx;
    const int ky = ky0 + c.s * ty, kx = kx0 + c.s * tx;
    return KEnt{co, ky, kx, (co * c.Cin * c.k + ky) * c.k + kx};
  }
  MLHOT_HD RCtx rctx(int m) const {
    const int img = m / (ny * nx), r = m % (ny * nx), yp = r / nx, xp = r % nx;
    return RCtx{img * c.Cout * c.HO * c.WO, c.s * yp + py + c.p, c.s * xp + px + c.p};
  }
  MLHOT_HD CCtx cctx(int n) const { return CCtx{n * c.k * c.k}; }
#ifndef MLHOT_HOSTSIM
  struct ARaw { float g, a; };                  // dY and the forward activation it is masked by (resolved at stash time)
  typedef float BRaw;
  MLHOT_DEV ARaw A2(const RCtx& r, const KEnt& e, int, int, bool ok) const {       // select-free, see ConvFwdRT
    const int ty_ = r.yb - e.ky, tx_ = r.xb - e.kx;           // multiples of s by construction of the class
    const int oy = c.s == 1 ? ty_ : (c.s == 2 ? ty_ >> 1 : ty_ / c.s), ox = c.s == 1 ? tx_ : (c.s == 2 ? tx_ >> 1 : tx_ / c.s);
    const bool in = ok & (ty_ >= 0) & (tx_ >= 0) & (oy < c.HO) & (ox < c.WO);
    const int o = r.ibase + (e.co * c.HO + oy) * c.WO + ox;
    ARaw v{load_or_const(dy, o, in), 1.f};
    if (yact != nullptr) v.a = load_or_const(yact, o, in);
    return v;
  }
  MLHOT_DEV float B2(const KEnt& e, const CCtx& cc, int, int, bool ok) const {
    return load_or_const(w, e.woff + cc.noff, ok);
  }
  MLHOT_DEV float Afin(const ARaw& v) const { return v.a > 0.f ? v.g : 0.f; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

struct ConvWgradRT {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = true;
  int M, N, K;            // M = Cout, N = Cin*k*k + 1, K = N*HO*WO
  ConvShape c;
  const float* dy; const float* yact; const float* x; float* dw; float* db;
  MLHOT_HD float A(int m, int k) const {
    const int hw = c.HO * c.WO, img = k / hw, r = k % hw;
    const size_t o = ((size_t)img * c.Cout + m) * hw + r;
    return (yact == nullptr || yact[o] > 0.f) ? dy[o] : 0.f;
  }
  MLHOT_HD float B(int k, int n) const {
    if (n == N - 1) return 1.f;
    const int hw = c.HO * c.WO, img = k / hw, r = k % hw, oy = r / c.WO, ox = r % c.WO;
    const int kk = c.k * c.k, ci = n / kk, t = n % kk, ky = t / c.k, kx = t % c.k;
    const int iy = oy * c.s + ky - c.p, ix = ox * c.s + kx - c.p;
    if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.W) return 0.f;
    return x[(((size_t)img * c.Cin + ci) * c.H + iy) * c.W + ix];
  }
  MLHOT_HD void store(int m, int n, float v) const {
    if (n == N - 1) { if (db) db[m] = v; }
    else dw[(size_t)m * (N - 1) + n] = v;
  }
  // hoisted index parts (igemm.h)
  struct alignas(16) KEnt { int abase, xbase, iy0, ix0; };  // k -> (img, oy, ox): img*Cout*hw + r, img*Cin*H*W + iy0*W + ix0
  struct RCtx { int moff; };                    // m*hw
  struct CCtx { int off, ky, kx, bias; };       // n -> (ci, ky, kx): ci*H*W + ky*W + kx
  MLHOT_HD KEnt kent(int k) const {
    const int hw = c.HO * c.WO, img = k / hw, r = k % hw, oy = r / c.WO, ox = r % c.WO;
    const int iy0 = oy * c.s - c.p, ix0 = ox * c.s - c.p;
    return KEnt{img * c.Cout * hw + r, img * c.Cin * c.H * c.W + iy0 * c.W + ix0, iy0, ix0};
  }
  MLHOT_HD RCtx rctx(int m) const { return RCtx{m * c.HO * c.WO}; }
  MLHOT_HD CCtx cctx(int n) const {
    if (n == N - 1) return CCtx{0, 0, 0, 1};
    const int kk = c.k * c.k, ci = n / kk, t = n % kk, ky = t / c.k, kx = t % c.k;
    return CCtx{(ci * c.H + ky) * c.W + kx, ky, kx, 0};
  }
#ifndef MLHOT_HOSTSIM
  struct ARaw { float g, a; };
  typedef float BRaw;
  MLHOT_DEV ARaw A2(const RCtx& r, const KEnt& e, int, int, bool ok) const {       // select-free, see ConvFwdRT
    const int o = e.abase + r.moff;
    ARaw v{load_or_const(dy, o, ok), 1.f};
    if (yact != nullptr) v.a = load_or_const(yact, o, ok);
    return v;
  }
  MLHOT_DEV float B2(const KEnt& e, const CCtx& cc, int, int, bool ok) const {
    const int iy = e.iy0 + cc.ky, ix = e.ix0 + cc.kx;
    const bool in = ok & (cc.bias == 0) & ((unsigned)iy < (unsigned)c.H) & ((unsigned)ix < (unsigned)c.W);
    return load_or_const(x, e.xbase + cc.off, in, (int)(ok & (cc.bias != 0)));   // the bias column reads the 1.0 word
  }
  MLHOT_DEV float Afin(const ARaw& v) const { return v.a > 0.f ? v.g : 0.f; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

// ------------------------------------------------------------------------------------------
// Linear layers:  Y = act(X W^T + b)   (nn.Linear, A.1).  The weight may be given as up to
// 8 row blocks (the 8 per-head AttnLinear matrices of _multihead_attention) so the head
// projections run as ONE GEMM without packing the parameters.
// ------------------------------------------------------------------------------------------

constexpr int MAX_WBLOCKS = 8;

struct WBlocks {          // logical weight [nb*rows][K] = concatenation of nb blocks [rows][K]
  const float* w[MAX_WBLOCKS];
  const float* b[MAX_WBLOCKS];   // may be nullptr
  int rows;                      // rows per block
};
struct WBlocksMut {
  float* w[MAX_WBLOCKS];
  float* b[MAX_WBLOCKS];
  int rows;
};

struct LinearFwd {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = true;
  int M, N, K;
  const float* x; int ldx;
  WBlocks wb;
  float* y; int ldy;
  int act;
  MLHOT_HD float A(int m, int k) const { return x[(size_t)m * ldx + k]; }
  MLHOT_HD float B(int k, int n) const { return wb.w[n / wb.rows][(size_t)(n % wb.rows) * K + k]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const float* bp = wb.b[n / wb.rows];
    if (bp) v += bp[n % wb.rows];
    y[(size_t)m * ldy + n] = act_apply(act, v);
  }
#ifndef MLHOT_HOSTSIM
  // hoisted index parts (igemm.h): the block lookup / division by `rows` happens once per tile column, not per element
  struct alignas(16) KEnt { int k, p0, p1, p2; };
  struct RCtx { int off; };
  struct CCtx { const float* row; };            // &W[n][0] of the block that holds output column n
  typedef float ARaw; typedef float BRaw;
  MLHOT_DEV KEnt kent(int k) const { return KEnt{k, 0, 0, 0}; }
  MLHOT_DEV RCtx rctx(int m) const { return RCtx{m * ldx}; }
  MLHOT_DEV CCtx cctx(int n) const { return CCtx{wb.w[n / wb.rows] + (size_t)(n % wb.rows) * K}; }
  MLHOT_DEV float A2(const RCtx& r, const KEnt& e, int, int, bool ok) const { return load_or_const(x, r.off + e.k, ok); }
  MLHOT_DEV float B2(const KEnt& e, const CCtx& cc, int, int, bool ok) const { return load_or_const(cc.row, e.k, ok); }
  MLHOT_DEV float Afin(float v) const { return v; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

// dX[M][Kin] (+)= (dY * act'(Y))[M][Nout] . W[Nout][Kin]
struct LinearDgrad {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = false;
  int M, N, K;               // GEMM dims: M rows, N = Kin, K = Nout
  const float* dy; int lddy;
  const float* y; int ldy;   // forward output (for the activation derivative); unused if act==NONE
  int act;
  WBlocks wb;                // blocks of [rows][Kin]
  float* dx; int lddx;
  int accumulate;
  MLHOT_HD float A(int m, int k) const {
    float g = dy[(size_t)m * lddy + k];
    if (act != ACT_NONE) g *= act_grad_from_out(act, y[(size_t)m * ldy + k]);
    return g;
  }
  MLHOT_HD float B(int k, int n) const { return wb.w[k / wb.rows][(size_t)(k % wb.rows) * N + n]; }
  MLHOT_HD void store(int m, int n, float v) const {
    float* o = dx + (size_t)m * lddx + n;
    *o = accumulate ? *o + v : v;
  }
#ifndef MLHOT_HOSTSIM
  struct alignas(16) KEnt { const float* wrow; int k, pad; };     // &W[k][0] of the block that holds row k
  struct RCtx { int offdy, offy; };
  struct CCtx { int n; };
  struct ARaw { float g, yv; };
  typedef float BRaw;
  MLHOT_DEV KEnt kent(int k) const { return KEnt{wb.w[k / wb.rows] + (size_t)(k % wb.rows) * N, k, 0}; }
  MLHOT_DEV RCtx rctx(int m) const { return RCtx{m * lddy, m * ldy}; }
  MLHOT_DEV CCtx cctx(int n) const { return CCtx{n}; }
  MLHOT_DEV ARaw A2(const RCtx& r, const KEnt& e, int, int, bool ok) const {
    ARaw v{load_or_const(dy, r.offdy + e.k, ok), 0.f};
    if (act != ACT_NONE) v.yv = load_or_const(y, r.offy + e.k, ok);
    return v;
  }
  MLHOT_DEV float B2(const KEnt& e, const CCtx& cc, int, int, bool ok) const { return load_or_const(e.wrow, cc.n, ok); }
  MLHOT_DEV float Afin(const ARaw& v) const { return act != ACT_NONE ? v.g * act_grad_from_out(act, v.yv) : v.g; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

// dW[Nout][Kin] = (dY*act')^T X, db = column sums (extra all-ones column n = Kin).
struct LinearWgrad {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = false;
  int M, N, K;               // GEMM dims: M = Nout, N = Kin+1, K = rows
  const float* dy; int lddy;
  const float* y; int ldy;
  int act;
  const float* x; int ldx;
  WBlocksMut gb;             // gradient blocks [rows][Kin] / bias [rows]
  MLHOT_HD float A(int m, int k) const {
    float g = dy[(size_t)k * lddy + m];
    if (act != ACT_NONE) g *= act_grad_from_out(act, y[(size_t)k * ldy + m]);
    return g;
  }
  MLHOT_HD float B(int k, int n) const { return n == N - 1 ? 1.f : x[(size_t)k * ldx + n]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const int blk = m / gb.rows, r = m % gb.rows;
    if (n == N - 1) { if (gb.b[blk]) gb.b[blk][r] = v; }
    else gb.w[blk][(size_t)r * (N - 1) + n] = v;
  }
#ifndef MLHOT_HOSTSIM
  struct alignas(16) KEnt { int offdy, offy, offx, pad; };        // row k of dY, Y and X
  struct RCtx { int m; };
  struct CCtx { int n, bias; };
  struct ARaw { float g, yv; };
  typedef float BRaw;
  MLHOT_DEV KEnt kent(int k) const { return KEnt{k * lddy, k * ldy, k * ldx, 0}; }
  MLHOT_DEV RCtx rctx(int m) const { return RCtx{m}; }
  MLHOT_DEV CCtx cctx(int n) const { return CCtx{n, n == N - 1}; }
  MLHOT_DEV ARaw A2(const RCtx& r, const KEnt& e, int, int, bool ok) const {
    ARaw v{load_or_const(dy, e.offdy + r.m, ok), 0.f};
    if (act != ACT_NONE) v.yv = load_or_const(y, e.offy + r.m, ok);
    return v;
  }
  MLHOT_DEV float B2(const KEnt& e, const CCtx& cc, int, int, bool ok) const {
    return load_or_const(x, e.offx + cc.n, ok & (cc.bias == 0), (int)(ok & (cc.bias != 0)));      // the bias column reads 1.0
  }
  MLHOT_DEV float Afin(const ARaw& v) const { return act != ACT_NONE ? v.g * act_grad_from_out(act, v.yv) : v.g; }
  MLHOT_DEV float Bfin(float v) const { return v; }
#endif
};

// ------------------------------------------------------------------------------------------
// The encoder's final Linear(4096 -> dim_w).  Its output rows are split in two segments
// (context rows land inside the encoder_r input, target rows inside the decoder input), so
// the "torch.cat" of the reference (ANPShapeNet1D.py:137,151) costs nothing.
// ------------------------------------------------------------------------------------------
struct Rows2 {            // rows [0,n0) -> (p0, ld0), rows [n0, ..) -> (p1, ld1)
  float* p0; int ld0; int n0; float* p1; int ld1;
  MLHOT_HD float* row(int r) const { return r < n0 ? p0 + (size_t)r * ld0 : p1 + (size_t)(r - n0) * ld1; }
};

struct EncLinFwd {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = true;
  int M, N, K;
  const float* a3;   // [M][K]
  const float* w;    // [N][K]
  const float* b;
  Rows2 out;
  MLHOT_HD float A(int m, int k) const { return a3[(size_t)m * K + k]; }
  MLHOT_HD float B(int k, int n) const { return w[(size_t)n * K + k]; }
  MLHOT_HD void store(int m, int n, float v) const { out.row(m)[n] = v + b[n]; }
};

// d a3[M][4096] = dfeat[M][dim_w] . w[dim_w][4096], masked by the conv3 ReLU (a3 > 0)
struct EncLinDgrad {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = false;
  int M, N, K;       // N = 4096, K = dim_w
  Rows2 dfeat;
  const float* w;    // [K][N]
  const float* a3;   // [M][N]
  float* dy3;        // [M][N]
  MLHOT_HD float A(int m, int k) const { return dfeat.row(m)[k]; }
  MLHOT_HD float B(int k, int n) const { return w[(size_t)k * N + n]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const size_t o = (size_t)m * N + n;
    dy3[o] = a3[o] > 0.f ? v : 0.f;
  }
};

struct EncLinWgrad {
  static constexpr bool A_ALONG_K = false, B_ALONG_K = false;
  int M, N, K;       // M = dim_w, N = 4096 + 1, K = rows
  Rows2 dfeat;
  const float* a3;   // [K][N-1]
  float* dw;         // [M][N-1]
  float* db;
  MLHOT_HD float A(int m, int k) const { return dfeat.row(k)[m]; }
  MLHOT_HD float B(int k, int n) const { return n == N - 1 ? 1.f : a3[(size_t)k * (N - 1) + n]; }
  MLHOT_HD void store(int m, int n, float v) const {
    if (n == N - 1) db[m] = v;
    else dw[(size_t)m * (N - 1) + n] = v;
  }
};

// ------------------------------------------------------------------------------------------
// FAVOR+ backward through the feature map: d x[row][e] = sum_j d(dd)[row][j] * (c P)[j][e]
//                                                        - rsum[row] * c^2 * x[row][e]
// d(dd) = G - [j == argmax] * (row sum of G)  for queries (fast_attention.py:92-94), and
// d(dd) = G - [this is THE global arg-max element] * (sum of G over everything) for keys (:97).
// ------------------------------------------------------------------------------------------
struct FavorDx {
  static constexpr bool A_ALONG_K = true, B_ALONG_K = false;
  int M, N, K;              // M = rows, N = d, K = m features
  const float* G;           // [M][K]
  const float* rsum;        // [M]
  const int* rowarg;        // [M]   (queries) or nullptr
  const int* gpos;          // {row, col} of the global arg-max (keys) or nullptr
  const float* gtotal;      // sum of all G (keys)
  const float* pc;          // [K][N]  c * projection
  const float* x;           // [M][N]
  float c2;                 // c^2
  float* dx;                // [M][N]
  MLHOT_HD float A(int m, int k) const {
    float g = G[(size_t)m * K + k];
    if (rowarg) { if (k == rowarg[m]) g -= rsum[m]; }
    else if (m == gpos[0] && k == gpos[1]) g -= gtotal[0];
    return g;
  }
  MLHOT_HD float B(int k, int n) const { return pc[(size_t)k * N + n]; }
  MLHOT_HD void store(int m, int n, float v) const {
    const size_t o = (size_t)m * N + n;
    dx[o] = v - rsum[m] * c2 * x[o];
  }
};

}  // namespace mlhot
