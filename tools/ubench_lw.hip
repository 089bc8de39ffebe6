// ubench_lw.hip — TQC's hidden-layer launches alone (5 nets x 512x512, B = 256), 16-row runs (k_lw_mid_run) vs
// 32-row runs (k_lw_mid_run2), fp32 and bf16, back to back (L2-warm).
//   hipcc -O3 --offload-arch=gfx950 -I oprl_amd/csrc tools/ubench_lw.hip -o tools/ubench_lw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../oprl_amd/csrc/layerwise.hip"

namespace oprl { bool mlp_slice_tp_shape_ok(const MlpArgs&, int) { return true; } }   // (slice_tp.hip is not part of this build)
using namespace oprl;

static float* dalloc(size_t n, float v) {
  float* p;
  (void)hipMalloc(&p, n * sizeof(float));
  std::vector<float> h(n, v);
  (void)hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

int main() {
  const int B = 256, W = 512, NETS = 5;
  MlpMultiArgs m{};
  for (int j = 0; j < NETS; ++j) {
    MlpArgs& a = m.a[j];
    a.B = B; a.do_fwd = 1; a.do_bwd = 1;
    a.net.n_layers = 4;
    a.net.dims[0] = 30; a.net.dims[1] = W; a.net.dims[2] = W; a.net.dims[3] = W; a.net.dims[4] = 25;
    for (int l = 0; l < 4; ++l) {
      a.net.pf[l] = dalloc((size_t)W * W, 0.01f);
      a.net.pb[l] = dalloc((size_t)W * W, 0.01f);
      a.net.b[l] = dalloc(W, 0.f);
      a.Xg[l] = dalloc((size_t)B * W, 0.5f);
      a.dYg[l] = dalloc((size_t)B * W, 0.5f);
    }
    a.x0 = dalloc((size_t)B * 24, 0.1f); a.k0 = 24;
    a.x1 = dalloc((size_t)B * 6, 0.1f); a.k1 = 6;
    a.ldx0 = 32;
  }
  (void)init_layerwise_attrs();
  hipStream_t st;
  (void)hipStreamCreate(&st);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const LwRun r = lw_run(B / 16, NETS, 256);
  const LwRun2 r2 = lw_run2(B, NETS, 256);
  const dim3 runs(8 * r.gpx * (B / 16)), runs2(8 * r2.ppx * r2.slices), blk(kThreads);
  printf("16-row runs: %d workgroups; 32-row runs: %d workgroups (rpn %d base %d rem %d)\n", runs.x, runs2.x, r2.rpn, r2.base, r2.rem);
  auto time_it = [&](const char* name, auto launch) {
    for (int w = 0; w < 20; ++w) launch();
    (void)hipEventRecord(e0, st);
    const int n = 300;
    for (int w = 0; w < n; ++w) launch();
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us per launch\n", name, ms * 1e3 / n);
  };
  time_it("run16 fwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run<0>), runs, blk, kLwRunLds, st, m, 2, r); });
  time_it("run32 fwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<0>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("run16 bwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run<1>), runs, blk, kLwRunLds, st, m, 2, r); });
  time_it("run32 bwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<1>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("run16 fwd+first f32", [&] { hipLaunchKernelGGL((k_lw_mid_run<2>), runs, blk, kLwRunLds, st, m, 1, r); });
  time_it("run32 fwd+first f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<2>), runs2, blk, kLwRun2Lds, st, m, 1, r2, 0); });
  time_it("run16 fwd bf16", [&] { hipLaunchKernelGGL((k_lw_mid_run<0, PrecBF16>), runs, blk, kLwRunLds, st, m, 2, r); });
  time_it("run32 fwd bf16", [&] { hipLaunchKernelGGL((k_lw_mid_run2<0, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("run32 fwd+first bf16", [&] { hipLaunchKernelGGL((k_lw_mid_run2<2, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, 1, r2, 0); });
  return 0;
}
