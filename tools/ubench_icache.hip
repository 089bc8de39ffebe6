// Microbenchmark: what instruction fetch costs a latency-bound straight-line kernel.  Every wave runs L
// dependent-chain FMAs (4 chains) once, straight-line (fully unrolled); timed inside the kernel
// (s_memrealtime at entry / exit of wave 0 of every workgroup) for a COLD launch (after a kernel with other
// code of 4 x the size has run on every CU) and a WARM one (the same kernel launched again at once).
// Also the same dynamic instruction count as a rolled loop of 64 FMAs.  256 workgroups x 1024 threads.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_icache.hip -o tools/ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int L, int ID>
__global__ __launch_bounds__(1024) void k_straight(long long* out, float* sink, float a, float b) {
  const long long t0 = wall_clock64();
  float x0 = a + threadIdx.x, x1 = b, x2 = a * b, x3 = a - b;
#pragma unroll
  for (int i = 0; i < L / 4; ++i) {
    x0 = __builtin_fmaf(x0, a, b + (float)(ID + i));       // distinct literal per step: the code cannot be rolled back up
    x1 = __builtin_fmaf(x1, b, a);
    x2 = __builtin_fmaf(x2, a, x0);
    x3 = __builtin_fmaf(x3, b, x1);
  }
  const long long t1 = wall_clock64();
  if (x0 + x1 + x2 + x3 == 12345.f) sink[threadIdx.x] = x0;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int L>
__global__ __launch_bounds__(1024) void k_loop(long long* out, float* sink, float a, float b) {
  const long long t0 = wall_clock64();
  float x0 = a + threadIdx.x, x1 = b, x2 = a * b, x3 = a - b;
#pragma unroll 1
  for (int o = 0; o < L / 64; ++o) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x0 = __builtin_fmaf(x0, a, b + (float)i);
      x1 = __builtin_fmaf(x1, b, a);
      x2 = __builtin_fmaf(x2, a, x0);
      x3 = __builtin_fmaf(x3, b, x1);
    }
  }
  const long long t1 = wall_clock64();
  if (x0 + x1 + x2 + x3 == 12345.f) sink[threadIdx.x] = x0;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

static void report(const char* what, long long* out) {
  long long h[256];
  CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  double mean = 0, mx = 0;
  for (int i = 0; i < 256; ++i) { mean += h[i] / 100.0; if (h[i] / 100.0 > mx) mx = h[i] / 100.0; }
  printf("  %-28s mean %6.2f us  max %6.2f us\n", what, mean / 256, mx);
}

template <int L>
static void run(long long* out, float* sink) {
  printf("L = %d FMAs per wave (16 waves per CU):\n", L);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k_straight<4 * L, 7>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);   // evict
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_straight<L, 1>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    report("straight-line, cold", out);
    hipLaunchKernelGGL((k_straight<L, 1>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    report("straight-line, warm", out);
    hipLaunchKernelGGL((k_straight<4 * L, 7>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k_loop<L>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    report("loop of 64, cold", out);
    hipLaunchKernelGGL((k_loop<L>), dim3(256), dim3(1024), 0, 0, out, sink, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    report("loop of 64, warm", out);
  }
}

int main() {
  long long* out; float* sink;
  CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&sink, 4096));
  run<256>(out, sink);
  run<1024>(out, sink);
  run<4096>(out, sink);
  return 0;
}
