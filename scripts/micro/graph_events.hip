// Can HIP-event pairs live inside a captured graph and be read after every replay?  (hipEventRecordWithFlags + hipEventRecordExternal)
// Standalone against /opt/rocm (HIP 7.2): yes - 0.59 ms per replay for the spin kernel below.  Inside the PyTorch process (its
// bundled HIP runtime) the same call fails with "hipEventRecord add external event node failed" / hipErrorInvalidValue, and events
// captured with plain hipEventRecord cannot be queried after a replay - which is why bench.py's per-kernel roofline timing runs
// EAGER steps (each launch between two events) and reads ~6 % longer kernels than rocprofv3 sees under back-to-back graph replay.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/graph_events.hip -o /tmp/ge && /tmp/ge
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
__global__ void spin(float* p, int n) { float s = 0.f; for (int i = 0; i < n; ++i) s += __sinf(s + i); if (s == 123.f) p[0] = s; }
int main() {
  hipStream_t st; CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, -1));   // as torch creates its pool streams
  float* d; CK(hipMalloc(&d, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecordWithFlags(a, st, hipEventRecordExternal));
  hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 20000);
  CK(hipEventRecordWithFlags(b, st, hipEventRecordExternal));
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) {
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float ms = -1.f; CK(hipEventElapsedTime(&ms, a, b));
    printf("replay %d: %.4f ms\n", i, ms);
  }
  return 0;
}
