// ubench_dw_wide.hip — k_dw_adam_wide alone on TQC's ten 512x512 layers (B = 256), whole and with parts
// switched off through its arguments, to see where its 42 us go.
//   hipcc -O3 --offload-arch=gfx950 -I oprl_amd/csrc tools/ubench_dw_wide.hip -o tools/ubench_dw_wide
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../oprl_amd/csrc/dw_wide.hip"

using namespace oprl;

static float* dalloc(size_t n, float v) {
  float* p;
  hipMalloc(&p, n * sizeof(float));
  std::vector<float> h(n, v);
  hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

int main() {
  const int N = 512, K = 512, B = 256, L = 10;
  DwItem it[L];
  for (int j = 0; j < L; ++j) {
    DwItem& I = it[j];
    I = DwItem{};
    I.X = dalloc((size_t)B * K, 0.01f); I.ldx = K; I.K = K;
    I.dY = dalloc((size_t)B * N, 0.001f); I.ldy = N; I.N = N;
    I.w = dalloc((size_t)N * K, 0.1f); I.w_t = dalloc((size_t)N * K, 0.1f);
    I.w_m = dalloc((size_t)N * K, 0.f); I.w_v = dalloc((size_t)N * K, 0.f); I.w_g = nullptr;
    I.b = dalloc(N, 0.f); I.b_t = dalloc(N, 0.f); I.b_m = dalloc(N, 0.f); I.b_v = dalloc(N, 0.f); I.b_g = nullptr;
    I.pf = dalloc((size_t)N * K, 0.f); I.pb = dalloc((size_t)N * K, 0.f); I.tpf = dalloc((size_t)N * K, 0.f);
    I.pf16 = I.pb16 = I.tpf16 = nullptr;
  }
  AdamScalars ad{};
  ad.lr = 3e-4f; ad.beta1 = 0.9f; ad.beta2 = 0.999f; ad.eps = 1e-8f;
  ad.omb1 = 0.1f; ad.omb2 = 0.001f; ad.omtau = 0.995f; ad.tau = 0.005f;
  ad.step_size_host = 3e-4f; ad.bc2_sqrt_host = 1.f; ad.do_polyak = 1; ad.do_adam = 1; ad.grad_scale = 1.f;
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, DwItem* items, int b, AdamScalars a) {
    for (int w = 0; w < 20; ++w) launch_dw_adam_wide(items, L, b, a, st, nullptr, 0);
    hipEventRecord(e0, st);
    const int n = 200;
    for (int w = 0; w < n; ++w) launch_dw_adam_wide(items, L, b, a, st, nullptr, 0);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us per launch\n", name, ms * 1e3 / n);
  };
  run("whole", it, B, ad);
  { DwItem v[L]; for (int j = 0; j < L; ++j) { v[j] = it[j]; v[j].pf = v[j].pb = v[j].tpf = nullptr; }
    run("no packs", v, B, ad); }
  { AdamScalars a = ad; a.do_polyak = 0;
    DwItem v[L]; for (int j = 0; j < L; ++j) { v[j] = it[j]; v[j].tpf = nullptr; }
    run("no Polyak (target tile, target pack)", v, B, a); }
  { AdamScalars a = ad; a.do_adam = 0; run("GEMM only (rows in, nothing out)", it, B, a); }
  run("no GEMM (B = 0): state in, state + packs out", it, 0, ad);
  { DwItem v[L]; for (int j = 0; j < L; ++j) { v[j] = it[j]; v[j].pf = v[j].pb = v[j].tpf = nullptr; }
    run("no GEMM, no packs", v, 0, ad); }
  { AdamScalars a = ad; a.do_adam = 0; run("nothing (B = 0, no Adam): launch + lookup", it, 0, a); }
  auto dump = [&]() {
#ifdef DWW_TRACE
  {
    std::vector<unsigned long long> t(1024 * 8);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_dww_trace), t.size() * 8);
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < 640; ++w) t0 = t[w * 8] < t0 ? t[w * 8] : t0;
    const char* names[5] = {"entry", "first rows in", "GEMM done", "Adam stores issued", "fp32 packs issued"};
    for (int k = 0; k < 5; ++k) {
      double mn = 1e9, mx = 0, av = 0;
      for (int w = 0; w < 640; ++w) { const double d = (double)(t[w * 8 + k] - t0) / 100.0; mn = d < mn ? d : mn; mx = d > mx ? d : mx; av += d / 640; }
      printf("  %-22s min %6.2f avg %6.2f max %6.2f us\n", names[k], mn, av, mx);
    }
    int late = 0;
    for (int w = 0; w < 640; ++w) late += (double)(t[w * 8] - t0) / 100.0 > 3.0 ? 1 : 0;
    printf("  workgroups entering after 3 us: %d\n", late);
    // per-workgroup durations
    double d01 = 0, d12 = 0, d23 = 0, d34 = 0;
    for (int w = 0; w < 640; ++w) { d01 += (t[w*8+1]-t[w*8]) / 64000.0; d12 += (t[w*8+2]-t[w*8+1]) / 64000.0; d23 += (t[w*8+3]-t[w*8+2]) / 64000.0; d34 += (t[w*8+4]-t[w*8+3]) / 64000.0; }
    printf("  per workgroup avg: entry->rows %.2f  rows->GEMM done %.2f  ->Adam stores %.2f  ->packs %.2f us\n", d01, d12, d23, d34);
  }
#endif
  };
  run("whole again", it, B, ad);
  dump();
  { AdamScalars a = ad; a.do_adam = 0; run("GEMM only again", it, B, a); }
  dump();
  return 0;
}
