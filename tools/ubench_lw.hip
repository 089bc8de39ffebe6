// ubench_lw.hip — TQC's hidden-layer launches alone (5 nets x 512x512, B = 256): k_lw_mid_run2 in the three precisions,
// back to back (L2-warm); with -DLW_TRACE the mean stage times of its workgroups (stamps in layerwise.hip).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLW_TRACE -I oprl_amd/csrc tools/ubench_lw.hip -o tools/ubench_lw
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
  const LwRun2 r2 = lw_run2(B, NETS, 256);
  const dim3 runs2(8 * r2.ppx * r2.slices), blk(kThreads);
  printf("32-row runs: %d workgroups (rpn %d base %d rem %d)\n", runs2.x, r2.rpn, r2.base, r2.rem);
  auto time_it = [&](const char* name, auto launch) {
    for (int w = 0; w < 20; ++w) launch();
    (void)hipEventRecord(e0, st);
    const int n = 300;
    for (int w = 0; w < n; ++w) launch();
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f us per launch", name, ms * 1e3 / n);
#ifdef LW_TRACE
    static unsigned long long h[4096 * 8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lw_trace), sizeof h);
    double acc[7] = {0}; int cnt = 0; unsigned long long t0 = ~0ull, t1 = 0;
    for (unsigned b = 0; b < runs2.x && b < 4096; ++b) {
      if (h[b * 8 + 6] == 0 || h[b * 8] == 0) continue;
      ++cnt;
      for (int k = 1; k < 7; ++k) acc[k] += (double)(h[b * 8 + k] - h[b * 8 + k - 1]) * 0.01;
      if (h[b * 8] < t0) t0 = h[b * 8];
      if (h[b * 8 + 6] > t1) t1 = h[b * 8 + 6];
    }
    printf("   stages (us, mean of %d wgs): issue %.2f | rows staged %.2f | mfma %.2f | barrier %.2f | partials %.2f | sum+store %.2f | first-in..last-out %.2f",
           cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt, acc[4] / cnt, acc[5] / cnt, acc[6] / cnt, (double)(t1 - t0) * 0.01);
    (void)hipMemset(nullptr, 0, 0);
#endif
    printf("\n");
  };
  time_it("fwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<0>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("bwd f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<1>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("fwd+first f32", [&] { hipLaunchKernelGGL((k_lw_mid_run2<2>), runs2, blk, kLwRun2Lds, st, m, 1, r2, 0); });
  time_it("fwd x2", [&] { hipLaunchKernelGGL((k_lw_mid_run2<0, PrecX2>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("bwd x2", [&] { hipLaunchKernelGGL((k_lw_mid_run2<1, PrecX2>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  time_it("fwd+first x2", [&] { hipLaunchKernelGGL((k_lw_mid_run2<2, PrecX2>), runs2, blk, kLwRun2Lds, st, m, 1, r2, 0); });
  time_it("fwd bf16", [&] { hipLaunchKernelGGL((k_lw_mid_run2<0, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, 2, r2, 0); });
  return 0;
}
