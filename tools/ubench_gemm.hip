// Ablation microbenchmark of the slice engine's hidden-layer GEMM (16 rows x 256 x 256,
// fp32 MFMA, fragment-order packs, 16 waves): where do the cycles go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I oprl_amd/csrc tools/ubench_gemm.hip -o tools/ubench_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "engine.h"
using namespace oprl;

__global__ void k_touch(float* w, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) w[i] = w[i] * 0.5f + v;
}

// MODE 0 full; 1 no global loads (B from a register constant); 2 no LDS A reads; 3 no MFMA (adds);
// 4 full with the ring kept full ACROSS layers (last RING slots refilled with the next layer's head)
template <int MODE, int RING, int NACC>
__global__ __launch_bounds__(1024) void k_gemm(const float* __restrict__ packs, int n_layers, float* out, long long* cyc) {
  __shared__ __attribute__((aligned(16))) float X[2][kR * lds_ld(256)];
  constexpr int WL = lds_ld(256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * kR * WL; i += 1024) (&X[0][0])[i] = 0.001f * (i % 97);
  __syncthreads();
  const float* xrow0 = &X[0][0] + (lane & 15) * WL + 4 * (lane >> 4);
  long long t0 = __builtin_readcyclecounter();
  int cur = 0;
  f32x4 carry[RING];
#pragma unroll
  for (int d = 0; d < RING; ++d) carry[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < n_layers; ++l) {
    const float* pl = packs + (size_t)l * 65536 + (size_t)wave * 16 * 256 + lane * 4;
    const float* xrow = xrow0 + cur * kR * WL;
    f32x4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    static_assert(MODE != 4 || true, "");
    f32x4 b[RING];
    if (MODE != 4 || l == 0) {
#pragma unroll
      for (int d = 0; d < RING; ++d) b[d] = (MODE == 1) ? f32x4{1.f, 2.f, 3.f, 4.f} : ld4(pl + (size_t)d * 256);
    } else {
#pragma unroll
      for (int d = 0; d < RING; ++d) b[d] = carry[d];
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int sb = 0; sb < 16; sb += RING) {
#pragma unroll
      for (int d = 0; d < RING; ++d) {
        const int s = sb + d;
        f32x4 a4 = (MODE == 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : ld4(xrow + 16 * s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (MODE == 3) acc[t % NACC] += a4[t] * b[d][t];
          else acc[t % NACC] = mfma4(a4[t], b[d][t], acc[t % NACC]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 1 && s + RING < 16) b[d] = ld4(pl + (size_t)(s + RING) * 256);
        else if (MODE == 4 && l + 1 < n_layers) b[d] = ld4(pl + 65536 + (size_t)(s + RING - 16) * 256);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 4) {
#pragma unroll
      for (int d = 0; d < RING; ++d) carry[d] = b[d];
    }
    f32x4 r = acc[0];
#pragma unroll
    for (int a = 1; a < NACC; ++a) r += acc[a];
    float* Y = &X[cur ^ 1][0];
#pragma unroll
    for (int q = 0; q < 4; ++q) Y[((lane >> 4) * 4 + q) * WL + 16 * wave + (lane & 15)] = fmaxf(r[q], 0.f) * 1e-3f;
    cur ^= 1;
  }
  __syncthreads();
  long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (X[cur][tid % (kR * WL)] == 123.456f) out[0] = 1.f;
}

template <int MODE, int RING, int NACC>
void run(const char* name, float* packs, int n_layers, int n_wg, float* out, long long* cyc) {
  std::vector<long long> h(n_wg);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, packs, (long)n_layers * 65536, 1e-4f * rep);
    hipLaunchKernelGGL((k_gemm<MODE, RING, NACC>), dim3(n_wg), dim3(1024), 0, 0, packs, n_layers, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, sizeof(long long) * n_wg, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int i = 0; i < n_wg; ++i) mx = h[i] > mx ? h[i] : mx;
    if (mx < best) best = mx;
  }
  printf("%-40s ring=%2d nacc=%d wgs=%2d: %8.0f cycles per layer (%.2f us @2.4GHz)\n", name, RING, NACC, n_wg, best / n_layers, best / n_layers / 2400.0);
}

int main() {
  const int NL = 32;
  float *packs, *out; long long* cyc;
  hipMalloc(&packs, (size_t)NL * 65536 * 4); hipMalloc(&out, 64); hipMalloc(&cyc, 8 * 256);
  hipMemset(packs, 0, (size_t)NL * 65536 * 4);
  run<0, 8, 1>("full", packs, NL, 16, out, cyc);
  run<0, 8, 2>("full, 2 accumulators", packs, NL, 16, out, cyc);
  run<0, 8, 4>("full, 4 accumulators", packs, NL, 16, out, cyc);
  run<0, 16, 1>("full", packs, NL, 16, out, cyc);
  run<0, 4, 1>("full", packs, NL, 16, out, cyc);
  run<1, 8, 1>("no global loads", packs, NL, 16, out, cyc);
  run<1, 8, 4>("no global loads, 4 accumulators", packs, NL, 16, out, cyc);
  run<2, 8, 1>("no LDS A reads", packs, NL, 16, out, cyc);
  run<3, 8, 1>("no MFMA (VALU fma)", packs, NL, 16, out, cyc);
  run<4, 8, 1>("full, ring kept full across layers", packs, NL, 16, out, cyc);
  run<4, 4, 1>("full, ring kept full across layers", packs, NL, 16, out, cyc);
  run<0, 8, 1>("full, 1 workgroup", packs, NL, 1, out, cyc);
  return 0;
}
