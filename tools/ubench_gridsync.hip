// Microbenchmark: cost of an in-kernel grid barrier + cross-XCD data hand-off on MI355X,
// against the cost of a kernel boundary doing the same.  One workgroup per CU slot
// (co-resident), each iteration: every WG publishes 4 KB, grid barrier, every WG reads the
// 4 KB of WG (b + G/2) % G (another XCD), checks it.
//   mode 0: sc1 (agent-scope write-through) stores + sc1 loads, barrier = atomic counter
//   mode 1: normal stores + release fence, barrier, acquire fence + normal loads
//   mode 2: one kernel launch per iteration (no in-kernel barrier)
// hipcc --offload-arch=gfx950 -O3 tools/ubench_gridsync.hip -o /tmp/gs && /tmp/gs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kT = 1024;

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spin = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spin < (1 << 22))
      __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(kT) void k_loop(float* buf, unsigned* counter, int iters, int it0, int* bad) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int peer = (b + G / 2 + 1) % G;
  int nbad = 0;
  for (int it = it0; it < it0 + iters; ++it) {
    const float val = (float)(it * 7 + b);
    float* mine = buf + ((size_t)(it & 1) * G + b) * kT;
    if (MODE == 0) {
      __hip_atomic_store(mine + tid, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      mine[tid] = val;
      if (MODE == 1) __atomic_thread_fence(__ATOMIC_RELEASE);   // agent-scope release: L2 writeback
    }
    if (MODE != 2) grid_barrier(counter, (unsigned)(it + 1) * G);
    else break;
    const float* theirs = buf + ((size_t)(it & 1) * G + peer) * kT;
    float got;
    if (MODE == 0) got = __hip_atomic_load(theirs + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else { __atomic_thread_fence(__ATOMIC_ACQUIRE); got = theirs[tid]; }
    if (got != (float)(it * 7 + peer)) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(kT) void k_read(const float* buf, int it, int* bad) {
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int peer = (b + G / 2 + 1) % G;
  if (it < 0) return;
  const float got = buf[((size_t)(it & 1) * G + peer) * kT + tid];
  if (got != (float)(it * 7 + peer)) atomicAdd(bad, 1);
}

int main() {
  const int G = 192, iters = 2000;
  float* buf; unsigned* counter; int* bad;
  CK(hipMalloc(&buf, sizeof(float) * 2 * G * kT));
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(buf, 0, sizeof(float) * 2 * G * kT));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    if (mode == 0) hipLaunchKernelGGL(k_loop<0>, dim3(G), dim3(kT), 64 * 1024, 0, buf, counter, iters, 0, bad);
    else if (mode == 1) hipLaunchKernelGGL(k_loop<1>, dim3(G), dim3(kT), 64 * 1024, 0, buf, counter, iters, 0, bad);
    else for (int it = 0; it < iters; ++it) {
      hipLaunchKernelGGL(k_loop<2>, dim3(G), dim3(kT), 64 * 1024, 0, buf, counter, 1, it, bad);
      // the next launch's read of the previous data rides on the boundary
      hipLaunchKernelGGL(k_read, dim3(G), dim3(kT), 0, 0, buf, it, bad);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("mode %d: %.2f us per iteration%s, mismatches %d\n", mode, ms * 1e3 / iters,
           mode == 2 ? " (2 launches)" : "", hb);
  }
  return 0;
}
