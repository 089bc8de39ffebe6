// Microbenchmark: granule ping-pong between two workgroups of the SAME XCD (blocks 0 and 8) and of two XCDs (0 and 1):
// which (store, poll) forms are served by the XCD's own L2, and what a hop costs then.  Memory: hipMalloc (cached).
//   store forms: 0 agent-scope atomic store | 1 workgroup-scope atomic store | 2 workgroup-scope atomic exchange (RMW)
//   poll forms:  0 agent-scope atomic load  | 1 workgroup-scope fetch_or(0) (an RMW: executes in the L2)
//                2 agent-scope fetch_or(0)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_hop.hip -o tools/ubench_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ST, int PL>
__global__ __launch_bounds__(1024) void k_pp(unsigned long long* g, int peer, int rounds, long long* out) {
  const int b = blockIdx.x;
  if (b != 0 && b != peer) return;
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned long long* mine = g + (b == 0 ? 0 : 64) + lane;
  unsigned long long* theirs = g + (b == 0 ? 64 : 0) + lane;
  auto put = [&](unsigned long long v) {
    if (ST == 0) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ST == 1) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (ST == 2) (void)__hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto get = [&]() -> unsigned long long {
    if (PL == 0) return __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (PL == 1) return __hip_atomic_fetch_or(theirs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_fetch_or(theirs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const long long t0 = wall_clock64();
  int fails = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (b == 0) put((unsigned long long)r);
    unsigned long long x = 0;
    int spin = 0;
    do { x = get(); } while (x < (unsigned long long)r && ++spin < (1 << 14));
    if (x < (unsigned long long)r) { ++fails; break; }
    if (b != 0) put((unsigned long long)r);
  }
  const long long t1 = wall_clock64();
  if (lane == 0 && b == 0) { out[0] = t1 - t0; out[1] = fails; }
}

template <int ST, int PL>
void run(unsigned long long* g, long long* out, int peer, const char* mem) {
  const int rounds = 2000;
  CK(hipMemset(g, 0, 128 * 8)); CK(hipMemset(out, 0, 16));
  hipLaunchKernelGGL((k_pp<ST, PL>), dim3(64), dim3(1024), 64 * 1024, 0, g, peer, rounds, out);
  CK(hipDeviceSynchronize());
  long long h[2];
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  printf("%-8s peer %d store %d poll %d: round trip %.3f us%s\n", mem, peer, ST, PL, (double)h[0] / 100.0 / rounds, h[1] ? "  (TIMEOUT: never observed)" : "");
}

int main() {
  unsigned long long *g, *gu; long long* out;
  CK(hipMalloc(&g, 128 * 8)); CK(hipMalloc(&out, 16));
  CK(hipExtMallocWithFlags((void**)&gu, 128 * 8, hipDeviceMallocUncached));
  for (int peer : {8, 1}) {
    run<0, 0>(g, out, peer, "cached");
    run<0, 0>(gu, out, peer, "uncached");
    run<1, 1>(g, out, peer, "cached");
    run<2, 1>(g, out, peer, "cached");
    run<0, 1>(g, out, peer, "cached");
    run<1, 0>(g, out, peer, "cached");
    run<0, 2>(g, out, peer, "cached");
    run<2, 2>(g, out, peer, "cached");
    run<1, 1>(gu, out, peer, "uncached");
  }
  return 0;
}
