// Microbenchmark: what the kernel-argument fetch costs at the head of a short kernel, and whether
// kernarg preloading into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N, gfx940+) removes it.
// 256 workgroups x 1024 threads; every workgroup: stamp at entry -> one global load through a pointer
// argument (an L2-cold line: rewritten by the previous kernel) -> stamp -> store.  Build twice:
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_kernarg.hip -o tools/ubench_kernarg
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 tools/ubench_kernarg.hip -o tools/ubench_kernarg_pre
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Big { const float* p[24]; int n[16]; };   // a by-value struct like the library's argument blocks

__global__ __launch_bounds__(1024) void k_scalar(long long* out, float* data, int it) {
  const long long t0 = wall_clock64();
  const float v = data[blockIdx.x * 1024 + threadIdx.x];
  data[blockIdx.x * 1024 + threadIdx.x] = v + 1.f;          // the next launch finds the line rewritten
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t0; }
}

__global__ __launch_bounds__(1024) void k_struct(long long* out, const Big a, float* data) {
  const long long t0 = wall_clock64();
  const float v = data[blockIdx.x * 1024 + threadIdx.x] + a.p[blockIdx.x % 24][threadIdx.x] * (float)a.n[blockIdx.x & 15];
  data[blockIdx.x * 1024 + threadIdx.x] = v + 1.f;
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t0; }
}

int main() {
  long long* out; float* data; float* w;
  CK(hipMalloc(&out, 256 * 16)); CK(hipMalloc(&data, 256 * 1024 * 4)); CK(hipMalloc(&w, 24 * 1024 * 4));
  CK(hipMemset(data, 0, 256 * 1024 * 4)); CK(hipMemset(w, 0, 24 * 1024 * 4));
  Big b;
  for (int i = 0; i < 24; ++i) b.p[i] = w + i * 1024;
  for (int i = 0; i < 16; ++i) b.n[i] = i;
  const int N = 2000;
  for (int variant = 0; variant < 2; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto c0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (variant == 0) hipLaunchKernelGGL(k_scalar, dim3(256), dim3(1024), 0, 0, out, data, i);
        else hipLaunchKernelGGL(k_struct, dim3(256), dim3(1024), 0, 0, out, b, data);
      }
      CK(hipDeviceSynchronize());
      auto c1 = std::chrono::steady_clock::now();
      long long h[512];
      CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
      double mean = 0, mx = 0;
      for (int i = 0; i < 256; ++i) { mean += h[2 * i] / 100.0; if (h[2 * i] / 100.0 > mx) mx = h[2 * i] / 100.0; }
      printf("%s: %.2f us per launch back to back; in-kernel entry -> load returned + store issued: mean %.2f us, max %.2f us\n",
             variant == 0 ? "scalar args" : "struct arg ", std::chrono::duration<double, std::micro>(c1 - c0).count() / N,
             mean / 256, mx);
    }
  }
  return 0;
}
