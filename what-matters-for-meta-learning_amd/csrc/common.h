// mlhot - MI355X-native CNP/ANP meta-batch hot path.  Common definitions.
//
// Two build flavours share every line of host orchestration and index arithmetic:
//   * the product:   hipcc --offload-arch=gfx950  -> libmlhot.so   (kernels run on the GPU)
//   * tests/hostsim: g++ -DMLHOT_HOSTSIM          -> libmlhot_hostsim.so
//     A TEST-ONLY artefact that replaces each kernel launch by a plain host loop over
//     the same index functors, so the im2col / parity-class / layout arithmetic can be
//     checked in the GPU-less build container.  The product never loads it.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdio.h>

#ifdef MLHOT_HOSTSIM
#define MLHOT_HD
#define MLHOT_DEV
typedef void* hipStream_t;
#else
#include <hip/hip_runtime.h>
#define MLHOT_HD __host__ __device__ __forceinline__
#define MLHOT_DEV __device__ __forceinline__
#endif

#ifndef MLHOT_HOSTSIM
// masked gathers point at these words instead of branching around the load or selecting after it (igemm.h, problems.h)
static __device__ const float g_zero_one[2] = {0.f, 1.f};
typedef const __attribute__((address_space(1))) float* gfptr;      // explicit global address space: global_load, not flat_load
// *(ok ? base + off : &g_zero_one[which]) as one global load, without a branch and without a select on the loaded value
__device__ __forceinline__ float load_or_const(const float* base, int off, bool ok, int which = 0) {
  gfptr p = ok ? (gfptr)base + off : (gfptr)g_zero_one + which;
  return *p;
}
#endif

#define MLHOT_OK 0
#define MLHOT_ERR_ARG 1
#define MLHOT_ERR_WORKSPACE 2
#define MLHOT_ERR_LAUNCH 3
#define MLHOT_ERR_UNSUPPORTED 4

namespace mlhot {

void set_error(const char* fmt, ...);

constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2;

MLHOT_HD float act_apply(int act, float v) {
  if (act == ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}
// derivative of the activation expressed through its OUTPUT y
MLHOT_HD float act_grad_from_out(int act, float y) {
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == ACT_TANH) return 1.f - y * y;
  return 1.f;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Bump allocator over a caller-owned workspace (no allocation inside the library).
struct Arena {
  char* base;
  size_t cap, off;
  bool ok;
  Arena(void* p, size_t bytes) : base((char*)p), cap(bytes), off(0), ok(true) {}
  template <class T>
  T* take(size_t n) {
    size_t o = align_up(off, 256);
    size_t need = n * sizeof(T);
    if (base == nullptr || o + need > cap) { ok = false; off = o + need; return nullptr; }
    off = o + need;
    return (T*)(base + o);
  }
};

// Optional per-launch HIP-event timing (bench.py's roofline leg).  Off by default; when on,
// every kernel launch is bracketed by two events recorded on the launch stream.
#ifndef MLHOT_HOSTSIM
void prof_record(const char* what, hipStream_t s, bool begin);
extern bool g_prof_on;
struct ProfScope {
  const char* what; hipStream_t s;
  ProfScope(const char* w, hipStream_t st) : what(w), s(st) { if (g_prof_on) prof_record(what, s, true); }
  ~ProfScope() { if (g_prof_on) prof_record(what, s, false); }
};
#else
struct ProfScope { ProfScope(const char*, hipStream_t) {} };
#endif

#ifndef MLHOT_HOSTSIM
// A helper stream of the library's own beside the caller's stream (defined in mlhot.hip).  fork(): the lane waits for
// everything enqueued on `main` so far; join(): `main` waits for the lane.  Both are event record + stream wait, so inside a
// hipGraph capture of `main` they become graph edges and the lane's launches parallel branches.  Used for work that is off the
// step's critical path (the weight-gradient slab folds of the encoder backward run beside the kernels that follow their
// producers).  One lane per caller stream (created on first use outside a capture; nullptr while `main` is capturing and no lane
// exists yet, or when the "side_fold" option is 0 - the callers then run the work in line).
struct SideLane {
  hipStream_t main, side; hipEvent_t ev_fork, ev_join;
  bool fork() { return hipEventRecord(ev_fork, main) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess; }
  bool join() { return hipEventRecord(ev_join, side) == hipSuccess && hipStreamWaitEvent(main, ev_join, 0) == hipSuccess; }
};
SideLane* side_lane(hipStream_t main);
#endif

#ifndef MLHOT_HOSTSIM
inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MLHOT_ERR_LAUNCH;
  }
  return MLHOT_OK;
}
#else
inline int check_launch(const char*) { return MLHOT_OK; }
#endif

}  // namespace mlhot
