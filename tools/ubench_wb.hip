// Microbenchmark: does the cache policy of a kernel's bulk stores change what the kernel BOUNDARY costs?
// On a multi-XCD part the end of a kernel writes the dirty lines of every XCD's L2 back to memory (the next
// kernel's workgroups may run on another XCD).  Kernel W stores `mb` MB from 256 workgroups (early in the
// kernel, then idles ~3 us like a latency-bound kernel), kernel R is trivial; the W,R pair is timed back to
// back.  Modes: plain stores | sc1 (agent-scope write-through) | nt | sc0 sc1 (system scope).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_wb.hip -o tools/ubench_wb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k_w(f32x4* buf, int n16, int idle) {
  const int idx = blockIdx.x * 1024 + threadIdx.x;
  const f32x4 v = {(float)idx, 1.f, 2.f, 3.f};
  for (int i = idx; i < n16; i += 256 * 1024) {
    f32x4* p = buf + i;
    if (MODE == 0) *p = v;
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < idle) __builtin_amdgcn_s_sleep(4);
}
__global__ __launch_bounds__(1024) void k_r(const f32x4* buf, float* out) {
  if (threadIdx.x == 0 && buf[blockIdx.x * 64][0] == -1.f) out[0] = 1.f;
}

template <int MODE>
static double run(f32x4* buf, float* out, int n16, int idle) {
  const int N = 2000;
  for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(k_w<MODE>, dim3(256), dim3(1024), 0, 0, buf, n16, idle); hipLaunchKernelGGL(k_r, dim3(256), dim3(1024), 0, 0, buf, out); }
  CK(hipDeviceSynchronize());
  auto c0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(k_w<MODE>, dim3(256), dim3(1024), 0, 0, buf, n16, idle); hipLaunchKernelGGL(k_r, dim3(256), dim3(1024), 0, 0, buf, out); }
  CK(hipDeviceSynchronize());
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count() / N;
}

int main() {
  f32x4* buf; float* out;
  CK(hipMalloc(&buf, 16 << 20)); CK(hipMalloc(&out, 64));
  for (int idle : {300, 0}) {           // 100 MHz ticks: 3 us of idling after the stores, or none
    for (double mb : {0.0, 1.0, 2.5, 8.0}) {
      const int n16 = (int)(mb * (1 << 20) / 16);
      printf("idle %d us, %4.1f MB stored per W launch: pair time plain %.2f | sc1 %.2f | nt %.2f | sc0sc1 %.2f us\n", idle / 100, mb,
             run<0>(buf, out, n16, idle), run<1>(buf, out, n16, idle), run<2>(buf, out, n16, idle), run<3>(buf, out, n16, idle));
    }
  }
  return 0;
}
