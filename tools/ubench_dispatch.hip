// ubench_dispatch.hip — when does the dispatcher place the NEXT workgroups of an over-subscribed grid?  One workgroup per
// compute unit (1024 threads, 133 KB of LDS); the first `n_long` workgroups run `t_long` us, the rest `t_short`; every
// workgroup records its start / end (100 MHz wall clock) and where it ran.  Question (r05-21): do the workgroups queued
// behind the first 256 start when the SHORT ones retire, or only when the long ones do?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_dispatch tools/ubench_dispatch.hip && tools/ubench_dispatch 109 11 8.4 640
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void k(long long* out, int n_long, long long ticks_long, long long ticks_short, int half_exit) {
  extern __shared__ float lds[];
  const long long t0 = wall_clock64();
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const long long want = id < n_long ? ticks_long : ticks_short;
  if (half_exit) {      // (the riding tiles' shape: a padded block of long workgroups — the padding exits at once — whose upper eight waves return)
    const int slots = (n_long + 63) / 64 * 64;
    if (id >= n_long && id < slots) return;
    if (id < n_long && threadIdx.x >= 512) return;
  }
  if (threadIdx.x == 0) lds[0] = 1.f;
  while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[id * 4 + 0] = t0;
    out[id * 4 + 1] = wall_clock64();
    out[id * 4 + 2] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;       // XCC_ID
    out[id * 4 + 3] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4); // HW_ID (wave / simd / cu ... low bits)
  }
}
int main(int argc, char** argv) {
  const int n_long = argc > 1 ? atoi(argv[1]) : 109;
  const double t_long = argc > 2 ? atof(argv[2]) : 11.0, t_short = argc > 3 ? atof(argv[3]) : 8.4;
  const int n = argc > 4 ? atoi(argv[4]) : 640;
  const int half_exit = argc > 5 ? atoi(argv[5]) : 0;
  long long* d;
  hipMalloc(&d, n * 4 * sizeof(long long));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 133 * 1024);
  std::vector<long long> h(n * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d, 0, n * 4 * sizeof(long long));
    hipLaunchKernelGGL(k, dim3(64, n / 64), dim3(1024), 133 * 1024, 0, d, n_long, (long long)(t_long * 100), (long long)(t_short * 100), half_exit);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, n * 4 * sizeof(long long), hipMemcpyDeviceToHost);
  long long t00 = h[0];
  for (int i = 0; i < n; ++i) t00 = std::min(t00, h[i * 4]);
  printf("n_long %d (%.1f us) then %d short (%.1f us), 64 x %d grid, half_exit %d\n", n_long, t_long, n - n_long, t_short, n / 64, half_exit);
  for (int lo = 0; lo < n; lo += 64) {
    std::vector<double> s;
    for (int i = lo; i < lo + 64 && i < n; ++i) s.push_back((h[i * 4] - t00) / 100.0);
    std::sort(s.begin(), s.end());
    printf("ids %3d..%3d start: min %6.2f  q1 %6.2f  median %6.2f  q3 %6.2f  max %6.2f\n", lo, lo + 63, s[0], s[s.size() / 4], s[s.size() / 2], s[3 * s.size() / 4], s.back());
  }
  long long t_end = 0;
  for (int i = 0; i < n; ++i) t_end = std::max(t_end, h[i * 4 + 1]);
  printf("last end %.2f us\n", (t_end - t00) / 100.0);
  return 0;
}
