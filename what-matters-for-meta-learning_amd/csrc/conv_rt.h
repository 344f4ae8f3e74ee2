// Host launchers for the run-time-shaped convolution problems (ResNet encoders, row E2 / D2).
#pragma once
#include "common.h"
#include "foreach.h"
#include "igemm.h"
#include "problems.h"
#include "ops_direct.h"
#include "favor.h"   // MLHOT_TRY

namespace mlhot {

#ifndef MLHOT_CONV_BK
#define MLHOT_CONV_BK 32
#endif
// k-depth of one igemm iteration for the run-time-shaped convolutions (measured on the ShapeNet3D ResNet shapes: 32 and 64
// within 3 % of each other, both ahead of 16; 32 wastes less of conv1's K = 75 and needs 94 instead of 180 VGPRs).
constexpr int CONV_BK = MLHOT_CONV_BK;
#ifndef MLHOT_CONV_BM         // large layers: 128 x 64 tiles on 8 waves (50-53 TFLOP/s; 64- and 32-row tiles measured 45-47)
#define MLHOT_CONV_BM 128
#define MLHOT_CONV_WM 4
#define MLHOT_CONV_WN 2
#endif

inline ConvShape conv_shape(int N, int Cin, int H, int W, int Cout, int k, int s, int p) {
  ConvShape c{N, Cin, H, W, Cout, k, s, p, (H + 2 * p - k) / s + 1, (W + 2 * p - k) / s + 1};
  return c;
}
inline int conv_wgrad_split(const ConvShape& c) {
  const long pos = (long)c.N * c.HO * c.WO;
  long tiles = ((c.Cout + 63) / 64) * (long)((c.Cin * c.k * c.k + 1 + 63) / 64);
  long want = (512 + tiles - 1) / tiles;            // aim at >= 512 workgroups
  long maxs = pos / 64;                             // at least 4 k-tiles of 16 per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  return (int)want;
}
inline size_t conv_bwd_scratch_bytes(const ConvShape& c) {
  return (size_t)conv_wgrad_split(c) * c.Cout * (c.Cin * c.k * c.k + 1) * sizeof(float) + 256;
}

// The late ResNet layers have a few hundred to a few thousand GEMM rows: 64-row tiles would leave most of the 256 CUs idle,
// so below 128 workgroups the row tile shrinks to 16 (one MFMA tile per wave, 4 waves side by side along N).
template <class P>
inline int run_conv_igemm(const P& p, hipStream_t s, const char* what) {
  const long wgs64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  if (wgs64 < 128) return run_igemm<P, 16, 64, CONV_BK, 1, 4>(p, 1, nullptr, s, what);
  return run_igemm<P, MLHOT_CONV_BM, 64, CONV_BK, MLHOT_CONV_WM, MLHOT_CONV_WN>(p, 1, nullptr, s, what);
}

inline int conv_rt_forward(const ConvShape& c, const float* x, const float* w, const float* b, float* y, int relu, hipStream_t s) {
  ConvFwdRT p{c.N * c.HO * c.WO, c.Cout, c.Cin * c.k * c.k, c, x, w, b, y, relu};
  return run_conv_igemm(p, s, "conv2d.fwd");
}

inline int conv_rt_backward(const ConvShape& c, const float* x, const float* w, const float* yact, const float* dy,
                            float* dx, float* dw, float* db, void* scratch, size_t scratch_bytes, hipStream_t s) {
  if (dw) {
    if (scratch_bytes < conv_bwd_scratch_bytes(c)) { set_error("conv2d_bwd: scratch too small"); return MLHOT_ERR_WORKSPACE; }
    ConvWgradRT p{c.Cout, c.Cin * c.k * c.k + 1, c.N * c.HO * c.WO, c, dy, yact, x, dw, db};
    MLHOT_TRY((run_igemm<ConvWgradRT, 64, 64, CONV_BK, 2, 2>(p, conv_wgrad_split(c), (float*)scratch, s, "conv2d.wgrad")));
  }
  if (dx) {
    // one GEMM per parity class of input positions (s*s of them); up to 4 go out as ONE launch
    IgemmBatch<ConvDgradRT> batch{};
    long wgs64 = 0;
    auto flush = [&]() -> int {
      if (batch.n == 0) return MLHOT_OK;
      const int rc = wgs64 < 128 ? run_igemm_batch<ConvDgradRT, 16, 64, CONV_BK, 1, 4>(batch, s, "conv2d.dgrad")
                                 : run_igemm_batch<ConvDgradRT, MLHOT_CONV_BM, 64, CONV_BK, MLHOT_CONV_WM, MLHOT_CONV_WN>(batch, s, "conv2d.dgrad");
      batch.n = 0; wgs64 = 0;
      return rc;
    };
    for (int py = 0; py < c.s; ++py)
      for (int px = 0; px < c.s; ++px) {
        const int ky0 = (py + c.p) % c.s, kx0 = (px + c.p) % c.s;
        const int nty = ky0 < c.k ? (c.k - ky0 + c.s - 1) / c.s : 0, ntx = kx0 < c.k ? (c.k - kx0 + c.s - 1) / c.s : 0;
        const int ny = (c.H - py + c.s - 1) / c.s, nx = (c.W - px + c.s - 1) / c.s;
        if (ny <= 0 || nx <= 0) continue;
        ConvDgradRT p{c.N * ny * nx, c.Cin, nty * ntx * c.Cout, c, py, px, ky0, kx0, nty, ntx > 0 ? ntx : 1, ny, nx, dy, yact, w, dx};
        if (nty * ntx == 0) p.K = 0;   // no tap reaches this class: the kernel stores zeros
        batch.p[batch.n++] = p;
        wgs64 += (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        if (batch.n == 4) MLHOT_TRY(flush());
      }
    MLHOT_TRY(flush());
  }
  return MLHOT_OK;
}

}  // namespace mlhot
