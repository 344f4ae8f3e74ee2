// How does v_mfma_f32_16x16x32_bf16 accumulate?  One wave, crafted operands: element (0, 0) of D = sum_k A[0][k] B[k][0] + C[0][0],
// with A's lane (row 0, k-group lq) holding k = 8 lq .. 8 lq + 7 (as every kernel of conv_split.h assumes; any consistent
// relabelling of k gives the same sum, so what the probe sees is the hardware's ORDER and internal rounding).
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_bf16_accum.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a /*[32]*/, const float* b /*[32]*/, float c, float* out, int fp32_path) {
  const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  if (lq == 0 && lr == 0) acc[0] = c;          // C[row 4 lq + r = 0][col lr = 0]
  if (!fp32_path) {
    bf16x8_t fa, fb;
    for (int j = 0; j < 8; ++j) {
      fa[j] = (__bf16)(lr == 0 ? a[8 * lq + j] : 0.f);      // A[row lr][k = 8 lq + j]
      fb[j] = (__bf16)(lr == 0 ? b[8 * lq + j] : 0.f);      // B[k][col lr]
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
  } else {
    for (int ks = 0; ks < 8; ++ks) {           // the fp32 kernels' chain: 8 x v_mfma_f32_16x16x4_f32, k = 4 ks + lq
      const float fa = lr == 0 ? a[4 * ks + lq] : 0.f, fb = lr == 0 ? b[4 * ks + lq] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
    }
  }
  if (lane == 0) out[0] = acc[0];
}

static float run(const std::vector<float>& a, const std::vector<float>& b, float c, int fp32_path) {
  float *da, *db, *dout, h;
  hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
  hipMemcpy(da, a.data(), 128, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 128, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout, fp32_path);
  hipMemcpy(&h, dout, 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  return h;
}

int main() {
  const float B24 = 16777216.f;      // 2^24 = 2^12 * 2^12
  auto zeros = [] { return std::vector<float>(32, 0.f); };
  printf("T1  +2^24 at k=p, 1.0 at k=q, -2^24 at k=r, C = 0: exact 1\n");
  const int trip[][3] = {{0, 1, 2}, {0, 2, 1}, {0, 8, 1}, {0, 1, 8}, {0, 8, 16}, {0, 16, 8}, {0, 9, 1}, {0, 31, 1}, {0, 1, 31}, {1, 0, 2}, {8, 0, 9}, {7, 0, 6}};
  for (auto& t : trip) {
    auto a = zeros(), b = zeros();
    a[t[0]] = 4096.f; b[t[0]] = 4096.f; a[t[1]] = 1.f; b[t[1]] = 1.f; a[t[2]] = -4096.f; b[t[2]] = 4096.f;
    printf("  p=%2d q=%2d r=%2d   bf16 mfma %g   fp32 chain %g\n", t[0], t[1], t[2], run(a, b, 0.f, 0), run(a, b, 0.f, 1));
  }
  printf("T2  C = 2^24 (ulp 2), ONE product v at k = 0: exact 2^24 + v\n");
  for (float v : {1.f, 3.f, 1.5f, 0.5f, 2.5f, -1.f, -3.f}) {
    auto a = zeros(), b = zeros();
    a[0] = v; b[0] = 1.f;
    printf("  v=%4g   bf16 mfma 2^24%+g   fp32 chain 2^24%+g\n", v, run(a, b, B24, 0) - B24, run(a, b, B24, 1) - B24);
  }
  printf("T3  C = 2^24, products 0.5 at n slots: exact 2^24 + n / 2\n");
  for (int n : {2, 4, 8, 16, 32}) {
    for (int stride : {1, 8}) {
      auto a = zeros(), b = zeros();
      for (int i = 0; i < n; ++i) { const int k = stride == 1 ? i : (i % 4) * 8 + i / 4; a[k] = 0.5f; b[k] = 1.f; }
      printf("  n=%2d (%s)   bf16 mfma 2^24%+g   fp32 chain 2^24%+g\n", n, stride == 1 ? "k = 0..n-1" : "spread over the lane groups",
             run(a, b, B24, 0) - B24, run(a, b, B24, 1) - B24);
    }
  }
  printf("T4  C = 1, product 2^-24 * m at k=0 (ulp(1) = 2^-23): exact 1 + m 2^-24\n");
  for (float m : {1.f, 2.f, 3.f, 1.5f}) {
    auto a = zeros(), b = zeros();
    a[0] = m; b[0] = ldexpf(1.f, -24);
    printf("  m=%4g   bf16 mfma 1%+g ulp   fp32 chain 1%+g ulp\n", m, (run(a, b, 1.f, 0) - 1.f) * 8388608.f, (run(a, b, 1.f, 1) - 1.f) * 8388608.f);
  }
  printf("T5  C = 0, +2^24 at k=p and -2^24 at k=r, 0.75 at EVERY other slot: exact 22.5\n");
  const int pairs[][2] = {{0, 1}, {0, 8}, {0, 16}, {3, 4}, {0, 31}};
  for (auto& t : pairs) {
    std::vector<float> a(32, 0.75f), b(32, 1.f);
    a[t[0]] = 4096.f; b[t[0]] = 4096.f; a[t[1]] = -4096.f; b[t[1]] = 4096.f;
    printf("  p=%2d r=%2d   bf16 mfma %g   fp32 chain %g\n", t[0], t[1], run(a, b, 0.f, 0), run(a, b, 0.f, 1));
  }
  printf("T6  small C beside a cancelling pair (+2^24 at k=0, -2^24 at k=1): exact C\n");
  for (float c : {1.75f, 1.25f, 1.5f, -1.75f, -1.25f, -1.5f, 0.75f, -0.75f, 3.75f, -3.75f}) {
    auto a = zeros(), b = zeros();
    a[0] = 4096.f; b[0] = 4096.f; a[1] = -4096.f; b[1] = 4096.f;
    printf("  C=%5g   bf16 mfma %g   fp32 chain %g\n", c, run(a, b, c, 0), run(a, b, c, 1));
  }
  printf("T7  small C beside ONE big product 2^24 at k=0: exact 2^24 + C (fp32 nearest: ulp 2)\n");
  for (float c : {1.f, 3.f, -1.f, -3.f, 1.5f, 2.5f, -2.5f}) {
    auto a = zeros(), b = zeros();
    a[0] = 4096.f; b[0] = 4096.f;
    printf("  C=%5g   bf16 mfma 2^24%+g   fp32 chain 2^24%+g\n", c, run(a, b, c, 0) - B24, run(a, b, c, 1) - B24);
  }
  printf("T8  C = 1.75, cancelling pair of magnitude 2^e in group 0 (k = 0, 1): what survives of C\n");
  for (int e : {20, 22, 23, 24, 25, 26, 28, 30}) {
    auto a = zeros(), b = zeros();
    a[0] = ldexpf(1.f, e / 2); b[0] = ldexpf(1.f, e - e / 2); a[1] = -a[0]; b[1] = b[0];
    printf("  e=%2d   bf16 mfma %g   (pair in group 2: %g)\n", e, run(a, b, 1.75f, 0), [&] { auto a2 = zeros(), b2 = zeros(); a2[16] = a[0]; b2[16] = b[0]; a2[17] = a[1]; b2[17] = b[1]; return run(a2, b2, 1.75f, 0); }());
  }
  return 0;
}
