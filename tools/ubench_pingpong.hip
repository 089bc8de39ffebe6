// Microbenchmark: granule ping-pong latency between two workgroups, same XCD vs other XCD,
// agent-scope (sc1: served by memory) vs workgroup-scope (sc0: served by the XCD's L2)
// atomics.  Grid of 64 WGs x 1024 threads, 64 KB LDS each (one per CU); WG 0 plays with
// WG `peer`.  Prints each player's XCC_ID (s_getreg HW_REG_XCC_ID).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SCOPE>
__global__ __launch_bounds__(1024) void k_pp(unsigned long long* g, int peer, int rounds, long long* out, int* xcc) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
  if (b != 0 && b != peer) return;
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned long long* mine = g + (b == 0 ? 0 : 64) + lane;
  unsigned long long* theirs = g + (b == 0 ? 64 : 0) + lane;
  const long long t0 = wall_clock64();
  int fails = 0;
  for (int r = 1; r <= rounds; ++r) {
    if (b == 0) __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
    unsigned long long x = 0;
    int spin = 0;
    do { x = __hip_atomic_load(theirs, __ATOMIC_RELAXED, SCOPE); } while (x < (unsigned long long)r && ++spin < (1 << 16));
    if (x < (unsigned long long)r) { ++fails; break; }
    if (b != 0) __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED, SCOPE);
  }
  const long long t1 = wall_clock64();
  if (lane == 0 && b == 0) { out[0] = t1 - t0; out[1] = fails; }
}

int main() {
  unsigned long long* g; long long* out; int* xcc;
  CK(hipMalloc(&g, 128 * 8)); CK(hipMalloc(&out, 16)); CK(hipMalloc(&xcc, 64 * 4));
  const int rounds = 2000;
  for (int peer : {8, 1}) {
    for (int scope = 0; scope < 2; ++scope) {
      CK(hipMemset(g, 0, 128 * 8)); CK(hipMemset(out, 0, 16));
      if (scope == 0) hipLaunchKernelGGL(k_pp<__HIP_MEMORY_SCOPE_AGENT>, dim3(64), dim3(1024), 64 * 1024, 0, g, peer, rounds, out, xcc);
      else hipLaunchKernelGGL(k_pp<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(64), dim3(1024), 64 * 1024, 0, g, peer, rounds, out, xcc);
      CK(hipDeviceSynchronize());
      long long h[2]; int hx[64];
      CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost));
      printf("peer %d (xcc %d vs %d) scope %s: round trip %.3f us, timeouts %lld\n", peer, hx[0], hx[peer],
             scope == 0 ? "agent" : "workgroup", (double)h[0] / 100.0 / rounds, h[1]);
    }
  }
  int hx[64]; CK(hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost));
  printf("xcc of blocks 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
  printf("\n");
  return 0;
}
