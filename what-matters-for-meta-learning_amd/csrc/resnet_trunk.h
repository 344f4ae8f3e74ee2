// Host orchestration of the ResNet trunks (5x5 s2 stem + four BN-free BasicBlocks; networks/models.py:63-192,
// networks/ResNet.py:58-74, BBB twin networks/ANPMRShapeNet3D.py:40-90) on the weight-stationary kernels of resnet_ws.h.
// ONE call runs every pass of a model step (context images, target images, decoder images ...), each pass with its weight set:
//   forward  : prep (lane-native weight images) -> stem -> per block {conv1 3x3 s2 + ReLU [+ fused 1x1 s2 skip | + 3x3 s2 skip
//              as a second job of the same launch]} -> {conv2 3x3 s1 + skip + ReLU}                       (14 launches for c5)
//   backward : prep -> ReLU mask of the incoming gradient -> per block {conv2 data gradient (+ mask), conv2 weight gradient,
//              conv1 / skip data gradient (stride-2 classes, join + mask fused), conv1 / skip weight gradient} -> stem weight
//              gradient -> ONE fold of all weight-gradient slabs.  Passes that share a weight set (the deterministic encoder over
//              context and target images) write into the same slab rows' segment, so their gradients come out already summed.
// Everything lives in caller-owned buffers: `act` (saved activations per pass) and one scratch arena per call.
#pragma once
#include "common.h"
#include "foreach.h"
#include "favor.h"          // MLHOT_TRY
#include "resnet_ws.h"
#include "../../include/mlhot.h"

namespace mlhot {
#ifndef MLHOT_HOSTSIM
namespace rt {

constexpr int NCONV = 13;           // stem, then (conv1, conv2, skip) of the four blocks

struct Levels { int C, H, L[5]; };  // L[0] = stem output size, L[i] = output size of block i
inline Levels trunk_levels(int C, int H) { Levels v{C, H, {H / 2, H / 4, H / 8, H / 16, H / 32}}; return v; }
inline bool trunk_supported(int C, int H) { return rw::stem_supported(C, H); }      // (3, 64) and (1, 128): every block geometry below them is instantiated
inline size_t act_floats(const Levels& lv, int n, int k) {      // k: 0 = a0, 2i-1 = mid_i, 2i = y_i
  const int l = k == 0 ? lv.L[0] : lv.L[(k + 1) / 2];
  return (size_t)n * 64 * l * l;
}

struct MaskJob { const float* g; const float* y; float* out; size_t n; };
struct MaskJobs { MaskJob j[MLHOT_TRUNK_MAX_PASS]; int n; size_t first[MLHOT_TRUNK_MAX_PASS + 1]; };
__global__ __launch_bounds__(256) void mask_kernel(const MaskJobs jobs) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < jobs.first[jobs.n]; i += (size_t)gridDim.x * 256) {
    int k = 0;
    while (k + 1 < jobs.n && i >= jobs.first[k + 1]) ++k;
    const size_t e = i - jobs.first[k];
    jobs.j[k].out[e] = jobs.j[k].y[e] > 0.f ? jobs.j[k].g[e] : 0.f;
  }
}

struct TrunkScratch {
  float* wimg[MLHOT_TRUNK_MAX_WSET][NCONV];       // F images (forward) / D images (backward; stem: unused)
  float* idn[MLHOT_TRUNK_MAX_PASS];               // forward: skip-path output of the current block
  float* G[MLHOT_TRUNK_MAX_PASS][5];              // backward: masked gradient wrt a0 / y_1..y_4
  float* DM[MLHOT_TRUNK_MAX_PASS];                // backward: masked gradient wrt mid_i of the current block
  float* idn4[MLHOT_TRUNK_MAX_PASS];              // fused blocks 3-4, forward: block 4's skip-path output (idn = block 3's)
  float* DM34[MLHOT_TRUNK_MAX_PASS][2];           // fused blocks 3-4, backward: masked gradients wrt mid_3, mid_4 (both outlive the fused launch)
  float* slab[MLHOT_TRUNK_MAX_WSET][NCONV]; float* slab_b[MLHOT_TRUNK_MAX_WSET][NCONV]; int rows[MLHOT_TRUNK_MAX_WSET][NCONV];
  bool ok; size_t bytes;
};

// slab rows (position splits) of one weight-gradient job: the launch aims at ~128 rows (x 4 channel tiles = 512 workgroups)
// over all its jobs, a row covers at least one band
inline int wg_rows(int bands, int total_bands, int target = 128) {
  int nz = total_bands <= target ? bands : (int)((long)target * bands / total_bands);
  if (nz < 1) nz = 1;
  if (nz > bands) nz = bands;
  return nz;
}
inline int conv_hin(const Levels& lv, int conv) {        // input size of conv index 1..12: (c1, c2, sk) of block b = (conv - 1) / 3 + 1
  const int b = (conv - 1) / 3 + 1, r = (conv - 1) % 3;
  return r == 1 ? lv.L[b] : lv.L[b - 1];
}
inline int conv_stride(int conv) { return (conv - 1) % 3 == 1 ? 1 : 2; }
// slab-row target of a 3x3 / 1x1 weight-gradient job: every row is 147 KB written by the kernel and read again by the fold.  The
// 4 x 4 / 2 x 2 output maps get 64 rows (256 workgroups): measured 128 / 64 / 32 rows at c5's shape - weight gradient of block 3's
// conv1 19.2 / 17.9 / 22.9 us, conv2 13.0 / 13.0 / 15.4, fold 34.3 / 33.9 / 32.9
extern int g_trunk_wg_rows;      // slab rows per weight-gradient launch of the larger maps (option "trunk_wg_rows", default 128)
inline int wg_target(const Levels& lv, int conv) { return conv_hin(lv, conv) / conv_stride(conv) <= 4 ? 64 : g_trunk_wg_rows; }

// the band total a job's slab rows are planned against: the 1x1 skips of a step share a launch of their own (skip1_wgrad_kernel),
// so they split ITS workgroups among themselves, not those of a launch that also serves the other passes
inline int skip1_total(const mlhot_trunk_pass* ps, int n_pass, const mlhot_trunk_wset* ws, int p, int conv, const int* bands, int total) {
  if (conv % 3 == 2) return total;                                   // conv2: a launch of its own over all passes
  if (conv % 3 == 0 && ws[ps[p].wset].skip_k == 1) {
    int t = 0;
    for (int i = 0; i < n_pass; ++i) if (ws[ps[i].wset].skip_k == 1) t += bands[i];
    return t;
  }
  // conv1 of every pass and the 3x3 skips (same input, same geometry) share ONE launch: its ~512 workgroups are split over all
  // of those jobs.  (Planned per convolution, ShapeNet3D's block-1 launch came to 840 workgroups for 512 resident slots: a second
  // round 64 % full, 143 us where one round needs ~95 - per-workgroup start / end stamps, scripts/dev/trunk_wgrad_ts.py.)
  int t = total;
  for (int i = 0; i < n_pass; ++i) if (ws[ps[i].wset].skip_k != 1) t += bands[i];
  return t;
}

inline TrunkScratch trunk_carve(const mlhot_trunk_pass* ps, int n_pass, const mlhot_trunk_wset* ws, int n_wset, const Levels& lv, bool backward,
                                void* base, size_t cap) {
  Arena a(base, cap);
  TrunkScratch s{};
  for (int w = 0; w < n_wset; ++w) {
    s.wimg[w][0] = a.take<float>((size_t)4 * ((25 * lv.C + 3) / 4) * 64);
    for (int c = 1; c < NCONV; ++c) s.wimg[w][c] = a.take<float>((c % 3 == 0 && ws[w].skip_k == 1) ? rw::WIMG1 : rw::WIMG);
  }
  for (int p = 0; p < n_pass; ++p) {
    if (!backward) { s.idn[p] = a.take<float>(act_floats(lv, ps[p].n_img, 2)); s.idn4[p] = a.take<float>(act_floats(lv, ps[p].n_img, 8)); continue; }
    for (int l = 0; l < 5; ++l) s.G[p][l] = a.take<float>(act_floats(lv, ps[p].n_img, 2 * l));
    s.DM[p] = a.take<float>(act_floats(lv, ps[p].n_img, 1));
    s.DM34[p][0] = a.take<float>(act_floats(lv, ps[p].n_img, 5)); s.DM34[p][1] = a.take<float>(act_floats(lv, ps[p].n_img, 7));
  }
  if (backward) {
    // slab rows per (weight set, conv): the sum over the passes that use the set
    for (int c = 0; c < NCONV; ++c) {
      int bands[MLHOT_TRUNK_MAX_PASS], total = 0;
      for (int p = 0; p < n_pass; ++p) {
        bands[p] = c == 0 ? ps[p].n_img * (lv.L[0] * lv.L[0] / 256) : rw::wgrad_bands_rt(conv_hin(lv, c), conv_stride(c), ps[p].n_img);
        total += bands[p];
      }
      for (int w = 0; w < n_wset; ++w) s.rows[w][c] = 0;
      for (int p = 0; p < n_pass; ++p) s.rows[ps[p].wset][c] += c == 0 ? wg_rows(bands[p], total, 512) : wg_rows(bands[p], skip1_total(ps, n_pass, ws, p, c, bands, total), wg_target(lv, c));
      for (int w = 0; w < n_wset; ++w) {
        const size_t rowlen = c == 0 ? (size_t)rw::stem_slab_row(lv.C) : ((c % 3 == 0 && ws[w].skip_k == 1) ? rw::SLAB1 : rw::SLAB3);
        s.slab[w][c] = a.take<float>(rowlen * (s.rows[w][c] > 0 ? s.rows[w][c] : 1));
        s.slab_b[w][c] = a.take<float>((size_t)64 * (s.rows[w][c] > 0 ? s.rows[w][c] : 1));
      }
    }
  }
  s.ok = a.ok; s.bytes = a.off + 256;
  return s;
}

// Blocks 3 and 4 as one launch per direction (rw::tail34_*): 64 x 64 trunks (8 x 8 -> 4 x 4 -> 2 x 2 maps), "trunk_fuse34" option
extern int g_trunk_fuse34, g_trunk_dual_dgrad;
inline bool fuse34(const Levels& lv) { return g_trunk_fuse34 && lv.L[2] == 8; }

inline int trunk_check(const mlhot_trunk_pass* ps, int n_pass, const mlhot_trunk_wset* ws, int n_wset, int C, int H) {
  if (!ps || !ws || n_pass < 1 || n_pass > MLHOT_TRUNK_MAX_PASS || n_wset < 1 || n_wset > MLHOT_TRUNK_MAX_WSET) { set_error("resnet trunk: bad pass / weight-set count"); return MLHOT_ERR_ARG; }
  if (!trunk_supported(C, H)) { set_error("resnet trunk: no kernels for %d-channel %dx%d images", C, H, H); return MLHOT_ERR_UNSUPPORTED; }
  for (int p = 0; p < n_pass; ++p)
    if (ps[p].n_img < 1 || ps[p].wset < 0 || ps[p].wset >= n_wset || !ps[p].img) { set_error("resnet trunk: bad pass %d", p); return MLHOT_ERR_ARG; }
  for (int w = 0; w < n_wset; ++w) {
    if (ws[w].skip_k != 1 && ws[w].skip_k != 3) { set_error("resnet trunk: skip kernel must be 1 or 3"); return MLHOT_ERR_ARG; }
    bool used = false;
    for (int p = 0; p < n_pass; ++p) used = used || ps[p].wset == w;
    // a weight set without a pass would get no slab rows: its gradients would never be written (the caller would read whatever was in the buffers)
    if (!used) { set_error("resnet trunk: weight set %d is not used by any pass", w); return MLHOT_ERR_ARG; }
  }
  return MLHOT_OK;
}

inline int trunk_prep(const mlhot_trunk_wset* ws, int n_wset, const Levels& lv, const TrunkScratch& sc, bool d_images, hipStream_t s) {
  rw::PrepItems items{};
  for (int w = 0; w < n_wset; ++w) {
    if (!d_images) items.it[items.n++] = rw::PrepItem{ws[w].w[0], sc.wimg[w][0], nullptr, 2, 25 * lv.C};
    for (int c = 1; c < NCONV; ++c) {
      const int kind = (c % 3 == 0 && ws[w].skip_k == 1) ? 1 : 0;
      items.it[items.n++] = d_images ? rw::PrepItem{ws[w].w[c], nullptr, sc.wimg[w][c], kind, 0} : rw::PrepItem{ws[w].w[c], sc.wimg[w][c], nullptr, kind, 0};
    }
  }
  {
    ProfScope ps("trunk.prep", s);
    hipLaunchKernelGGL(rw::prep_kernel, dim3(144, items.n), dim3(256), 0, s, items);      // 4 x 144 = the 576 wave items of a 3x3 image: one trip per wave (with 16 blocks a wave made nine load -> store trips, each a round trip: 10.6 us)
  }
  return check_launch("trunk.prep");
}

// launch labels per block (mlhot_prof_*): a label is one geometry, so its time over its algorithmic FLOPs is a kernel's rate
#define MLHOT_TRUNK_LABELS(name) {name ".b0", name ".b1", name ".b2", name ".b3", name ".b4"}
static const char* const LBL_CONV1[5] = MLHOT_TRUNK_LABELS("trunk.conv1");
static const char* const LBL_CONV2[5] = MLHOT_TRUNK_LABELS("trunk.conv2");
static const char* const LBL_C2_DGRAD[5] = MLHOT_TRUNK_LABELS("trunk.bwd.conv2.dgrad");
static const char* const LBL_C2_WGRAD[5] = MLHOT_TRUNK_LABELS("trunk.bwd.conv2.wgrad");
static const char* const LBL_C1_DGRAD[5] = MLHOT_TRUNK_LABELS("trunk.bwd.conv1.dgrad");
static const char* const LBL_C1_DGRAD2[5] = MLHOT_TRUNK_LABELS("trunk.bwd.conv1.dgrad2");
static const char* const LBL_C1_WGRAD[5] = MLHOT_TRUNK_LABELS("trunk.bwd.conv1.wgrad");

inline int trunk_forward(const mlhot_trunk_pass* ps, int n_pass, const mlhot_trunk_wset* ws, int n_wset, int C, int H, void* scratch,
                         size_t scratch_bytes, hipStream_t s) {
  MLHOT_TRY(trunk_check(ps, n_pass, ws, n_wset, C, H));
  const Levels lv = trunk_levels(C, H);
  const TrunkScratch sc = trunk_carve(ps, n_pass, ws, n_wset, lv, false, scratch, scratch_bytes);
  if (!sc.ok) { set_error("resnet trunk fwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes); return MLHOT_ERR_WORKSPACE; }
  MLHOT_TRY(trunk_prep(ws, n_wset, lv, sc, false, s));
  {
    rw::StemJobs jobs{};
    for (int p = 0; p < n_pass; ++p) jobs.j[jobs.n++] = rw::StemJob{ps[p].img, sc.wimg[ps[p].wset][0], ws[ps[p].wset].b[0], ps[p].act[0], ps[p].n_img, 0, 0};
    MLHOT_TRY(rw::stem_dispatch(C, H, jobs, s, "trunk.stem"));
  }
  const bool fused = fuse34(lv);
  for (int b = 1; b <= (fused ? 2 : 4); ++b) {
    const int c1 = 3 * b - 2, c2 = c1 + 1, sk = c1 + 2;
    // stage A: conv1 (+ReLU) and the skip convolution, both on the block input - ONE launch for every pass: a 3x3 skip is a
    // second job on the same input, a 1x1 skip rides in its conv1 job (centre-tap operand)
    // On the 32x32 input the kernel variant with the fused 1x1 skip (16 more weight registers, a second epilogue) costs the
    // passes that do not need it more than a launch: there the two kinds of pass go out separately (158 us in one launch,
    // 49 + 97 us in two); from 16x16 down one launch is the faster way.
    const bool split_kinds = lv.L[b - 1] >= 32;
    for (int kind = 0; kind < (split_kinds ? 2 : 1); ++kind) {
      rw::FwdJobs jobs{};
      bool any1 = false;
      for (int p = 0; p < n_pass; ++p) {
        const mlhot_trunk_wset& w = ws[ps[p].wset];
        const float* x = ps[p].act[2 * b - 2];
        if (split_kinds && (w.skip_k == 1) != (kind == 0)) continue;
        if (jobs.n + (w.skip_k == 1 ? 1 : 2) > rw::MAX_JOBS) {
          MLHOT_TRY(rw::conv3x3_dispatch(lv.L[b - 1], 2, any1, jobs, s, LBL_CONV1[b]));
          jobs.n = 0; any1 = false;
        }
        if (w.skip_k == 1) {
          any1 = true;
          jobs.j[jobs.n++] = rw::FwdJob{x, sc.wimg[ps[p].wset][c1], w.b[c1], ps[p].act[2 * b - 1], nullptr, sc.wimg[ps[p].wset][sk], w.b[sk], sc.idn[p],
                                        ps[p].n_img, rw::EPI_BIAS_RELU, 0, 0, 0};
        } else {
          jobs.j[jobs.n++] = rw::FwdJob{x, sc.wimg[ps[p].wset][c1], w.b[c1], ps[p].act[2 * b - 1], nullptr, nullptr, nullptr, nullptr, ps[p].n_img,
                                        rw::EPI_BIAS_RELU, 0, 0, 0};
          jobs.j[jobs.n++] = rw::FwdJob{x, sc.wimg[ps[p].wset][sk], w.b[sk], sc.idn[p], nullptr, nullptr, nullptr, nullptr, ps[p].n_img, rw::EPI_BIAS, 0, 0, 0};
        }
      }
      if (jobs.n) MLHOT_TRY(rw::conv3x3_dispatch(lv.L[b - 1], 2, any1, jobs, s, LBL_CONV1[b]));
    }
    // stage B: conv2 + skip + ReLU
    rw::FwdJobs jobs{};
    for (int p = 0; p < n_pass; ++p)
      jobs.j[jobs.n++] = rw::FwdJob{ps[p].act[2 * b - 1], sc.wimg[ps[p].wset][c2], ws[ps[p].wset].b[c2], ps[p].act[2 * b], sc.idn[p], nullptr, nullptr, nullptr,
                                    ps[p].n_img, rw::EPI_BIAS_RES_RELU, 0, 0, 0};
    MLHOT_TRY(rw::conv3x3_dispatch(lv.L[b], 1, false, jobs, s, LBL_CONV2[b]));
  }
  if (fused) {
    rw::T34Jobs jobs{};
    for (int p = 0; p < n_pass; ++p) {
      const int wi = ps[p].wset;
      const mlhot_trunk_wset& w = ws[wi];
      rw::T34Pass& t = jobs.p[jobs.n++];
      t = rw::T34Pass{};
      t.x = ps[p].act[4]; t.mid3 = ps[p].act[5]; t.y3 = ps[p].act[6]; t.mid4 = ps[p].act[7]; t.y4 = ps[p].act[8];
      t.idn3 = sc.idn[p]; t.idn4 = sc.idn4[p];
      const int order[6] = {7, 8, 9, 10, 11, 12};        // c1_3, c2_3, sk_3, c1_4, c2_4, sk_4
      for (int i = 0; i < 6; ++i) { t.wimg[i] = sc.wimg[wi][order[i]]; t.b[i] = w.b[order[i]]; }
      t.n_img = ps[p].n_img; t.skip1 = w.skip_k == 1;
    }
    MLHOT_TRY(rw::tail34_launch(jobs, false, s, "trunk.fwd.b34"));
  }
  return MLHOT_OK;
}

inline int trunk_backward(const mlhot_trunk_pass* ps, int n_pass, const mlhot_trunk_wset* ws, int n_wset, int C, int H, void* scratch,
                          size_t scratch_bytes, hipStream_t s) {
  MLHOT_TRY(trunk_check(ps, n_pass, ws, n_wset, C, H));
  const Levels lv = trunk_levels(C, H);
  TrunkScratch sc = trunk_carve(ps, n_pass, ws, n_wset, lv, true, scratch, scratch_bytes);
  if (!sc.ok) { set_error("resnet trunk bwd: scratch too small (%zu < %zu)", scratch_bytes, sc.bytes); return MLHOT_ERR_WORKSPACE; }
  for (int p = 0; p < n_pass; ++p) if (!ps[p].dfeat) { set_error("resnet trunk bwd: pass %d has no output gradient", p); return MLHOT_ERR_ARG; }
  MLHOT_TRY(trunk_prep(ws, n_wset, lv, sc, true, s));
  {
    MaskJobs mj{};
    for (int p = 0; p < n_pass; ++p) {
      mj.j[p] = MaskJob{ps[p].dfeat, ps[p].act[8], sc.G[p][4], act_floats(lv, ps[p].n_img, 8)};
      mj.first[p + 1] = mj.first[p] + mj.j[p].n;
    }
    mj.n = n_pass;
    size_t blocks = (mj.first[n_pass] + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    ProfScope pr("trunk.bwd.mask", s);
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, mj);
    MLHOT_TRY(check_launch("trunk.bwd.mask"));
  }
  int next_row[MLHOT_TRUNK_MAX_WSET][NCONV] = {};
  // one weight-gradient launch over `which` convs' jobs (same geometry): returns through `jobs`
  auto wg_jobs = [&](int conv, bool tap1, bool want_tap1, rw::WgJobs& jobs, const float* const* xs, const float* const* dys) {
    int bands[MLHOT_TRUNK_MAX_PASS], total = 0;
    for (int p = 0; p < n_pass; ++p) { bands[p] = rw::wgrad_bands_rt(conv_hin(lv, conv), conv_stride(conv), ps[p].n_img); total += bands[p]; }
    for (int p = 0; p < n_pass; ++p) {
      const int w = ps[p].wset;
      if (conv % 3 == 0 && ((ws[w].skip_k == 1) != want_tap1)) continue;
      (void)tap1;
      const int nz = wg_rows(bands[p], skip1_total(ps, n_pass, ws, p, conv, bands, total), wg_target(lv, conv));
      jobs.j[jobs.n++] = rw::WgJob{xs[p], dys[p], sc.slab[w][conv], sc.slab_b[w][conv], ps[p].n_img, next_row[w][conv], nz, 0};
      next_row[w][conv] += nz;
    }
  };
  const bool fused = fuse34(lv);
  if (fused) {      // every data gradient of blocks 4 and 3 in one launch; their weight gradients follow in the loop below
    rw::T34Jobs jobs{};
    for (int p = 0; p < n_pass; ++p) {
      const int wi = ps[p].wset;
      rw::T34Pass& t = jobs.p[jobs.n++];
      t = rw::T34Pass{};
      t.x = ps[p].act[4]; t.mid3 = ps[p].act[5]; t.y3 = ps[p].act[6]; t.mid4 = ps[p].act[7]; t.y4 = ps[p].act[8];
      const int order[6] = {7, 8, 9, 10, 11, 12};
      for (int i = 0; i < 6; ++i) t.wimg[i] = sc.wimg[wi][order[i]];
      t.g4 = sc.G[p][4]; t.dm4 = sc.DM34[p][1]; t.g3 = sc.G[p][3]; t.dm3 = sc.DM34[p][0]; t.g2 = sc.G[p][2];
      t.n_img = ps[p].n_img; t.skip1 = ws[wi].skip_k == 1;
    }
    MLHOT_TRY(rw::tail34_launch(jobs, true, s, "trunk.bwd.b34.dgrad"));
  }
  rw::Sk1Jobs sk1{};
  rw::Wg34Jobs w34{};      // fused blocks 3-4: their 3x3 weight gradients as ONE launch when the job table holds them (conv1 + conv2 of every
  int n_skip3 = 0;         // pass + the 3x3 skips, for both blocks: 16 jobs for c5's three passes); per-block launches otherwise
  for (int p = 0; p < n_pass; ++p) n_skip3 += ws[ps[p].wset].skip_k == 3;
  const bool w34_ok = fused && 2 * (2 * n_pass + n_skip3) <= rw::WG34_MAX;
  for (int b = 4; b >= 1; --b) {
    const int c1 = 3 * b - 2, c2 = c1 + 1, sk = c1 + 2;
    const bool in_fused = fused && b >= 3;
    const float *xin[MLHOT_TRUNK_MAX_PASS], *mid[MLHOT_TRUNK_MAX_PASS], *g[MLHOT_TRUNK_MAX_PASS], *dm[MLHOT_TRUNK_MAX_PASS];
    for (int p = 0; p < n_pass; ++p) {
      xin[p] = ps[p].act[2 * b - 2]; mid[p] = ps[p].act[2 * b - 1]; g[p] = sc.G[p][b];
      dm[p] = in_fused ? sc.DM34[p][b - 3] : sc.DM[p];
    }
    if (!in_fused) {   // conv2 data gradient: d_mid = conv^T(g, W2) * (mid > 0)
      rw::FwdJobs jobs{};
      for (int p = 0; p < n_pass; ++p)
        jobs.j[jobs.n++] = rw::FwdJob{g[p], sc.wimg[ps[p].wset][c2], nullptr, sc.DM[p], mid[p], nullptr, nullptr, nullptr, ps[p].n_img, rw::EPI_MASK, 1, 0, 0};
      MLHOT_TRY(rw::conv3x3_dispatch(lv.L[b], 1, false, jobs, s, LBL_C2_DGRAD[b]));
    }
    {   // conv2 weight gradient
      rw::WgJobs jobs{};
      wg_jobs(c2, false, false, jobs, mid, g);
      if (in_fused && w34_ok) { for (int i = 0; i < jobs.n; ++i) { w34.j[w34.n] = jobs.j[i]; w34.geo[w34.n++] = b == 3 ? 1 : 3; } }      // one launch behind block 3 (rw::wgrad34_kernel)
      else MLHOT_TRY(rw::wgrad_dispatch(lv.L[b], 1, false, jobs, s, LBL_C2_WGRAD[b]));
    }
    // data gradient into the block input (not needed for images: block 1's input is the stem output, whose gradient feeds the stem's wgrad)
    if (!in_fused) {
      // launch 1: every first writer of dx - the 3x3 skips' data gradients and the 1x1-skip blocks' fused conv1 + skip gradient;
      // launch 2: the 3x3-skip blocks' conv1 gradient, added onto launch 1's result and masked
      rw::DgJobs ja{}, jb2{}, jd{};
      bool any1 = false;
      const bool dual = g_trunk_dual_dgrad && rw::dgrad2_dual_supported(lv.L[b]);
      for (int p = 0; p < n_pass; ++p) {
        const int w = ps[p].wset;
        if (ws[w].skip_k == 1) { any1 = true; ja.j[ja.n++] = rw::DgJob{dm[p], sc.wimg[w][c1], sc.G[p][b - 1], xin[p], g[p], sc.wimg[w][sk], ps[p].n_img, 0, 0, 0}; }
        else if (dual) {
          // 3x3 skip: both sources (skip^T(g), conv1^T(dm)) in ONE launch, dx written once with its mask (rw::dgrad2_dual_kernel)
          jd.j[jd.n++] = rw::DgJob{g[p], sc.wimg[w][sk], sc.G[p][b - 1], xin[p], dm[p], sc.wimg[w][c1], ps[p].n_img, 0, 0, 0};
        } else {
          ja.j[ja.n++] = rw::DgJob{g[p], sc.wimg[w][sk], sc.G[p][b - 1], nullptr, nullptr, nullptr, ps[p].n_img, 0, 0, 0};
          jb2.j[jb2.n++] = rw::DgJob{dm[p], sc.wimg[w][c1], sc.G[p][b - 1], xin[p], nullptr, nullptr, ps[p].n_img, 1, 0, 0};
        }
      }
      MLHOT_TRY(rw::dgrad2_dispatch(lv.L[b], any1, ja, s, LBL_C1_DGRAD[b]));      // (split by kind like the forward: no gain here)
      MLHOT_TRY(rw::dgrad2_dispatch(lv.L[b], false, jb2, s, LBL_C1_DGRAD2[b]));
      MLHOT_TRY(rw::dgrad2_dual_dispatch(lv.L[b], jd, s, LBL_C1_DGRAD2[b]));
    }
    {   // conv1 and 3x3-skip weight gradients (same input, same geometry: one launch); 1x1 skips on their own
      rw::WgJobs jobs{}, jobs1{};
      wg_jobs(c1, false, false, jobs, xin, dm);
      if (in_fused && w34_ok) {
        for (int i = 0; i < jobs.n; ++i) { w34.j[w34.n] = jobs.j[i]; w34.geo[w34.n++] = b == 3 ? 0 : 2; }
        jobs.n = 0;
        wg_jobs(sk, false, false, jobs, xin, g);
        for (int i = 0; i < jobs.n; ++i) { w34.j[w34.n] = jobs.j[i]; w34.geo[w34.n++] = b == 3 ? 0 : 2; }
        if (b == 3) MLHOT_TRY(rw::wgrad34_launch(w34, s, "trunk.bwd.b34.wgrad"));
      } else {
        if (jobs.n + n_pass > rw::MAX_JOBS) { MLHOT_TRY(rw::wgrad_dispatch(lv.L[b - 1], 2, false, jobs, s, LBL_C1_WGRAD[b])); jobs.n = 0; }
        wg_jobs(sk, false, false, jobs, xin, g);
        MLHOT_TRY(rw::wgrad_dispatch(lv.L[b - 1], 2, false, jobs, s, LBL_C1_WGRAD[b]));
      }
      // 1x1 skips: collected over the blocks, one launch behind the loop (same slab rows as a per-block launch would use)
      wg_jobs(sk, true, true, jobs1, xin, g);
      for (int i = 0; i < jobs1.n; ++i) {
        const rw::WgJob& j = jobs1.j[i];
        int lg = 0;
        while ((2 << lg) < lv.L[b - 1]) ++lg;                                  // output map HO = L[b - 1] / 2 = 1 << lg
        if (sk1.n >= rw::SK1_MAX) { set_error("resnet trunk bwd: too many 1x1 skip jobs"); return MLHOT_ERR_ARG; }
        sk1.j[sk1.n++] = rw::Sk1Job{j.x, j.dy, j.slab, j.slab_b, j.n_img, lg, j.z0, j.nz, 0};      // a row beyond the last chunk is written as zeros
      }
    }
  }
  MLHOT_TRY(rw::skip1_wgrad_launch(sk1, s, "trunk.bwd.skip1.wgrad"));
  {   // stem weight gradient
    rw::StemWgJobs jobs{};
    int bands[MLHOT_TRUNK_MAX_PASS], total = 0;
    for (int p = 0; p < n_pass; ++p) { bands[p] = ps[p].n_img * (lv.L[0] * lv.L[0] / 256); total += bands[p]; }
    for (int p = 0; p < n_pass; ++p) {
      const int w = ps[p].wset, nz = wg_rows(bands[p], total, 512);    // 512 workgroups in all
      jobs.j[jobs.n++] = rw::StemWgJob{ps[p].img, sc.G[p][0], sc.slab[w][0], ps[p].n_img, next_row[w][0], nz, 0};
      next_row[w][0] += nz;
    }
    MLHOT_TRY(rw::stem_wgrad_dispatch(C, H, jobs, s, "trunk.bwd.stem.wgrad"));
  }
  // fold the slabs: every convolution's weight and bias gradient of every weight set in one launch (a second one only if the
  // segment table overflows its kernel-argument block)
  rw::WsumSegs segs{};
  segs.base = static_cast<const float*>(scratch);
  auto flush = [&]() -> int {
    if (segs.blocks > 0) {
      ProfScope pr("trunk.bwd.wsum", s);
      hipLaunchKernelGGL(rw::wsum_kernel, dim3(segs.blocks), dim3(256), 0, s, segs);
      MLHOT_TRY(check_launch("trunk.bwd.wsum"));
    }
    segs.n = 0; segs.blocks = 0;
    return MLHOT_OK;
  };
  auto add = [&](const float* slab, float* out, float* out_b, int nrows, int kind, int K) -> int {
    if (segs.n >= rw::WSUM_MAX) MLHOT_TRY(flush());
    if (!rw::wsum_add(segs, slab, out, out_b, nrows, kind, K)) { set_error("resnet trunk bwd: slab does not fit the fold table"); return MLHOT_ERR_ARG; }
    return MLHOT_OK;
  };
  for (int w = 0; w < n_wset; ++w) {
    if (next_row[w][0] > 0 && ws[w].dw[0]) MLHOT_TRY(add(sc.slab[w][0], ws[w].dw[0], ws[w].db[0], next_row[w][0], 3, 25 * lv.C));
    for (int c = 1; c < NCONV; ++c) {
      if (next_row[w][c] <= 0 || !ws[w].dw[c]) continue;
      MLHOT_TRY(add(sc.slab[w][c], ws[w].dw[c], nullptr, next_row[w][c], (c % 3 == 0 && ws[w].skip_k == 1) ? 1 : 0, 0));
      if (ws[w].db[c]) MLHOT_TRY(add(sc.slab_b[w][c], ws[w].db[c], nullptr, next_row[w][c], 2, 0));
    }
  }
  MLHOT_TRY(flush());
  return MLHOT_OK;
}

}  // namespace rt
#endif  // !MLHOT_HOSTSIM
}  // namespace mlhot
