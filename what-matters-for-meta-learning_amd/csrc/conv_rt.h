// Host launchers for the run-time-shaped convolution problems (ResNet encoders, row E2 / D2).
#pragma once
#include "common.h"
#include "foreach.h"
#include "igemm.h"
#include "problems.h"
#include "ops_direct.h"
#include "favor.h"   // MLHOT_TRY

namespace mlhot {

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

inline int conv_rt_forward(const ConvShape& c, const float* x, const float* w, const float* b, float* y, int relu, hipStream_t s) {
  ConvFwdRT p{c.N * c.HO * c.WO, c.Cout, c.Cin * c.k * c.k, c, x, w, b, y, relu};
  return run_igemm<ConvFwdRT, 64, 64, 16, 2, 2>(p, 1, nullptr, s, "conv2d.fwd");
}

inline int conv_rt_backward(const ConvShape& c, const float* x, const float* w, const float* yact, const float* dy,
                            float* dx, float* dw, float* db, void* scratch, size_t scratch_bytes, hipStream_t s) {
  if (dw) {
    if (scratch_bytes < conv_bwd_scratch_bytes(c)) { set_error("conv2d_bwd: scratch too small"); return MLHOT_ERR_WORKSPACE; }
    ConvWgradRT p{c.Cout, c.Cin * c.k * c.k + 1, c.N * c.HO * c.WO, c, dy, yact, x, dw, db};
    MLHOT_TRY((run_igemm<ConvWgradRT, 64, 64, 16, 2, 2>(p, conv_wgrad_split(c), (float*)scratch, s, "conv2d.wgrad")));
  }
  if (dx) {
    for (int py = 0; py < c.s; ++py)
      for (int px = 0; px < c.s; ++px) {
        const int ky0 = (py + c.p) % c.s, kx0 = (px + c.p) % c.s;
        const int nty = ky0 < c.k ? (c.k - ky0 + c.s - 1) / c.s : 0, ntx = kx0 < c.k ? (c.k - kx0 + c.s - 1) / c.s : 0;
        const int ny = (c.H - py + c.s - 1) / c.s, nx = (c.W - px + c.s - 1) / c.s;
        if (ny <= 0 || nx <= 0) continue;
        ConvDgradRT p{c.N * ny * nx, c.Cin, nty * ntx * c.Cout, c, py, px, ky0, kx0, nty, ntx > 0 ? ntx : 1, ny, nx, dy, yact, w, dx};
        if (nty * ntx == 0) p.K = 0;   // no tap reaches this class: the kernel stores zeros
        MLHOT_TRY((run_igemm<ConvDgradRT, 64, 64, 16, 2, 2>(p, 1, nullptr, s, "conv2d.dgrad")));
      }
  }
  return MLHOT_OK;
}

}  // namespace mlhot
