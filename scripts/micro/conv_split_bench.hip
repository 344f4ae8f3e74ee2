// scripts/micro/conv_split_bench.hip: the split-precision conv12 forward alone (480 random images), timed with HIP events, plus -
// built with -DC2S_TS=<band> - the cycle stamps of that band of workgroup 0, per wave (see C2S_STAMP in csrc/conv_split.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=40000 -I what-matters-for-meta-learning_amd/csrc \
//         [-DC2S_TS=10] scripts/micro/conv_split_bench.hip -o /tmp/csb && /tmp/csb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "conv_split.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int n = 480;
  std::vector<float> img((size_t)n * 16384), w1(32 * 9), b1(32), w2(48 * 288), b2(48);
  srand(1);
  auto rnd = [] { return rand() / (float)RAND_MAX; };
  for (auto& v : img) v = rnd();
  for (auto& v : w1) v = 0.6f * (rnd() - 0.5f);
  for (auto& v : b1) v = 0.2f * (rnd() - 0.5f);
  for (auto& v : w2) v = 0.12f * (rnd() - 0.5f);
  for (auto& v : b2) v = 0.2f * (rnd() - 0.5f);
  float *dimg, *dw1, *db1, *dw2, *db2, *dp2; uint8_t* dam; unsigned* dm1;
  CK(hipMalloc(&dimg, img.size() * 4)); CK(hipMalloc(&dw1, w1.size() * 4)); CK(hipMalloc(&db1, 128)); CK(hipMalloc(&dw2, w2.size() * 4));
  CK(hipMalloc(&db2, 192)); CK(hipMalloc(&dp2, (size_t)n * 48 * 256 * 4)); CK(hipMalloc(&dam, (size_t)n * 48 * 256));
  CK(hipMalloc(&dm1, ((size_t)n * 4096 + 16) * 4));
  CK(hipMemcpy(dimg, img.data(), img.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db1, b1.data(), 128, hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db2, b2.data(), 192, hipMemcpyHostToDevice));
  mlhot::c2::ImgSrc x{dimg, n, nullptr};
  auto launch = [&] {
    hipLaunchKernelGGL(mlhot::c2s::conv12_fwd_split_kernel, dim3(256), dim3(mlhot::c2s::NT), 0, 0, x, dw1, db1, dw2, db2, dp2, dam, dm1, n);
  };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> p2((size_t)n * 48 * 256);
  CK(hipMemcpy(p2.data(), dp2, p2.size() * 4, hipMemcpyDeviceToHost));
  double cs = 0; for (float v : p2) cs += v;
  printf("conv12_fwd_split: %.1f us per launch (checksum %.6e)\n", 1e3 * ms / reps, cs);
#ifdef C2S_TS
  long long ts[196];
  CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(mlhot::c2s::g_c2s_ts), sizeof(ts)));
  printf("main loop of workgroup 0: %lld cycles, %lld bands, %.0f cycles per band, %.2f GHz\n", ts[192], ts[194], (double)ts[192] / ts[194], ts[192] / (ts[193] * 10.0));
  long long t0 = ts[0]; for (int w = 1; w < 12; ++w) if (ts[w * 16] < t0) t0 = ts[w * 16];
  printf("band %d, cycles from the first wave's start: top | site0 | taps0-4 | site1 | taps5-8 | site2 | scratch | barrier | pool\n", C2S_TS);
  for (int w = 0; w < 12; ++w) {
    printf("wave %2d (slot %d):", w, w >> 2);
    for (int i = 0; i < 9; ++i) printf(" %6lld", ts[w * 16 + i] - t0);
    printf("\n");
  }
#endif
  return 0;
}
