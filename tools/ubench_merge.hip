// Microbenchmark for merging a producer kernel (k_dw_adam-like: rewrites "weights") and a
// consumer kernel (phase-like: reads them) into ONE launch with a completion counter:
//   - producers (blocks 0..P-1, 512 active threads) write 4 KB each with sc1 (write-through)
//     16-byte stores, drain (vmcnt 0), then bump an agent-scope counter;
//   - consumers (blocks P.., 1024 threads) first do 2 us of independent work, then wait for the
//     counter and read 96 KB of the producers' data with sc1 16-byte buffer loads.
// The consumers' L2 is polluted on purpose: every launch they ALSO re-read the region with plain
// loads at the very end, so the next launch finds stale lines in L2.
// Compared with the same work as two dependent launches.  Checks every value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int P = 152, Cn = 64, KB4 = 1024;   // floats per producer

__device__ __forceinline__ f32x4 ld4_sc1(const float* base, int elem_off) {
  // raw buffer load, cache policy sc0|sc1 (aux bit0 = sc0, bit4 = sc1 on gfx94x/gfx950)
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, elem_off * 4, 0, 1 | 16);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void st4_sc1(float* base, int elem_off, f32x4 v) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), r, elem_off * 4, 0, 1 | 16);
}

template <bool MERGED, bool COHERENT>
__global__ __launch_bounds__(1024) void k_step(float* w, unsigned* counter, unsigned target, int it, int role_sel, int* bad) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool producer = MERGED ? b < P : role_sel == 0;
  const int pb = b;
  if (producer) {
    if (tid >= 512) return;
    // "dW": ~3 us of dependent work, then the rewrite
    float acc = (float)it;
    for (int k = 0; k < 300; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 0.f);
    const float val = (float)(it * 131 + pb) + (acc - acc);
    f32x4 v = {val, val + 1.f, val + 2.f, val + 3.f};
    if (tid < KB4 / 4) {
      if (COHERENT) st4_sc1(w + (size_t)pb * KB4, tid * 4, v);
      else *reinterpret_cast<f32x4*>(w + (size_t)pb * KB4 + tid * 4) = v;
    }
    if (MERGED) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int cb = MERGED ? b - P : b;
  // independent prologue (~2 us)
  float acc = (float)cb;
  for (int k = 0; k < 200; ++k) acc = __builtin_fmaf(acc, 1.0000001f, 0.f);
  if (MERGED) {
    if (tid == 0) {
      int spin = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spin < (1 << 22))
        __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  // read 24 producers' 4 KB each = 96 KB: thread t reads float4 t%256 of producer (cb*7 + t/256 + 4j) % P
  int nbad = 0;
  for (int j = 0; j < 6; ++j) {
    const int p = (cb * 7 + (tid >> 8) + 4 * j) % P;
    const f32x4 v = COHERENT ? ld4_sc1(w + (size_t)p * KB4, (tid & 255) * 4)
                             : *reinterpret_cast<const f32x4*>(w + (size_t)p * KB4 + (tid & 255) * 4);
    const float e = (float)(it * 131 + p);
    if (v[0] != e || v[3] != e + 3.f) ++nbad;
  }
  if (acc == -1.f) nbad += 1000;
  // pollute this XCD's L2 with plain reads of everything (stale next launch)
  float s = 0.f;
  for (int k = tid; k < P * KB4; k += 1024 * 8) s += w[k];
  if (s == -12345.f) nbad += 1;
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  float* w; unsigned* counter; int* bad;
  CK(hipMalloc(&w, sizeof(float) * P * KB4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 1000;
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(counter, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(w, 0, sizeof(float) * P * KB4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int it = 1; it <= iters; ++it) {
      if (mode == 0) {          // two dependent launches, plain accesses
        hipLaunchKernelGGL((k_step<false, false>), dim3(P), dim3(1024), 64 * 1024, 0, w, counter, 0u, it, 0, bad);
        hipLaunchKernelGGL((k_step<false, false>), dim3(Cn), dim3(1024), 128 * 1024, 0, w, counter, 0u, it, 1, bad);
      } else if (mode == 1) {   // one launch, counter, sc1 stores + sc1 loads
        hipLaunchKernelGGL((k_step<true, true>), dim3(P + Cn), dim3(1024), 128 * 1024, 0, w, counter, (unsigned)(it * P), it, 0, bad);
      } else {                  // one launch, counter, PLAIN accesses (expected to read stale data)
        hipLaunchKernelGGL((k_step<true, false>), dim3(P + Cn), dim3(1024), 128 * 1024, 0, w, counter, (unsigned)(it * P), it, 0, bad);
      }
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    const char* names[3] = {"two launches, plain", "one launch + counter, sc1", "one launch + counter, plain"};
    printf("%-32s %.2f us per step, mismatches %d\n", names[mode], ms * 1e3 / iters, hb);
  }
  return 0;
}
