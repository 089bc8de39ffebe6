// Microbenchmark: what an in-kernel acquire costs on a multi-XCD part — `buffer_inv sc1` (invalidate what this
// XCD's caches may hold stale of other XCDs' writes) and `buffer_wbl2 sc1` (write this XCD's dirty lines back),
// issued by every workgroup of a 256 x 1024 launch, and whether data another XCD wrote through (sc1 stores) is
// then seen by PLAIN loads.  One launch: phase 0 all workgroups read a buffer (their L2 now holds it), phase 1
// workgroup b rewrites its 4 KB with sc1 stores, grid barrier (atomic counter), phase 2: inv, plain loads of the
// neighbour XCD's 4 KB, check.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_inv.hip -o tools/ubench_inv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spin = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int MODE>   // 0: no acquire (expect stale reads), 1: buffer_inv sc1, 2: buffer_wbl2 sc1 + buffer_inv sc1
__global__ __launch_bounds__(1024) void k(float* buf, unsigned* counter, int iters, long long* out, int* bad) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int peer = (b + 1) % G;                 // the next block lives on the next XCD
  long long t_acq = 0;
  int nbad = 0;
  for (int it = 1; it <= iters; ++it) {
    float v = buf[(size_t)peer * 1024 + tid];   // cache the peer's current (old) data
    if (v == -1.f) nbad += 1000000;
    __hip_atomic_store(buf + (size_t)b * 1024 + tid, (float)(it * 7 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    grid_barrier(counter, (unsigned)it * G);
    const long long t0 = wall_clock64();
    if (MODE == 1) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)\n buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
    t_acq += wall_clock64() - t0;
    const float got = buf[(size_t)peer * 1024 + tid];      // PLAIN load
    if (got != (float)(it * 7 + peer)) ++nbad;
    grid_barrier(counter + 1, (unsigned)it * G);            // nobody rewrites before everybody has read
  }
  if (tid == 0) out[b] = t_acq;
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  const int G = 256, iters = 200;
  float* buf; unsigned* counter; long long* out; int* bad;
  CK(hipMalloc(&buf, sizeof(float) * G * 1024)); CK(hipMalloc(&counter, 8)); CK(hipMalloc(&out, G * 8)); CK(hipMalloc(&bad, 4));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(buf, 0, sizeof(float) * G * 1024)); CK(hipMemset(counter, 0, 8)); CK(hipMemset(bad, 0, 4));
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(1024), 0, 0, buf, counter, iters, out, bad);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(G), dim3(1024), 0, 0, buf, counter, iters, out, bad);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(G), dim3(1024), 0, 0, buf, counter, iters, out, bad);
    CK(hipDeviceSynchronize());
    long long h[256]; int hb;
    CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    double mean = 0, mx = 0;
    for (int i = 0; i < G; ++i) { const double u = h[i] / 100.0 / iters; mean += u; if (u > mx) mx = u; }
    printf("mode %d (%s): acquire %.2f us mean, %.2f us max per workgroup and iteration; stale or wrong plain reads: %d\n", mode,
           mode == 0 ? "none" : mode == 1 ? "buffer_inv sc1" : "buffer_wbl2 sc1 + buffer_inv sc1", mean / G, mx, hb);
  }
  return 0;
}
