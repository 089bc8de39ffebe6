// Microbenchmark of the hand-over the whole-update kernels rely on: a PRODUCER workgroup writes a block, waits for its
// stores (s_waitcnt vmcnt(0)), raises a flag; a CONSUMER workgroup on another XCD polls the flag, invalidates its L1 and
// reads the block.  How often does the consumer see OLD data, by store width / cache policy / memory type, with and
// without other workgroups streaming memory in the background?  (r04-18 / -20: 16-byte write-through stores diverged
// bit-identical runs, 8-byte ones almost never.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_handoff.hip -o tools/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const volatile f32x4 __attribute__((address_space(1))) * gvp4;

// ST: 0 plain 16-byte store | 1 16-byte store, sc1 (raw buffer store builtin) | 2 two 8-byte agent-scope atomic stores
//     3 four 4-byte agent-scope atomic stores
// LD: 0 plain 16-byte load behind buffer_inv sc0 | 1 16-byte load, sc1 (raw buffer load builtin)
template <int ST, int LD>
__global__ __launch_bounds__(1024) void k(float* data, unsigned long long* flags, int pairs, int iters, unsigned* bad, float* bg, size_t bg_floats) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= 2 * pairs) {                                  // background: stream memory
    float acc = 0.f;
    const size_t n4 = bg_floats / 4, stride = (size_t)(gridDim.x - 2 * pairs) * 1024;
    for (int rep = 0; rep < iters / 64 + 1; ++rep)
      for (size_t i = (size_t)(b - 2 * pairs) * 1024 + tid; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(bg)[i];
        acc += v[0];
        v[1] += 1.f;
        reinterpret_cast<f32x4*>(bg)[i] = v;
      }
    if (acc == 12345.678f) bad[1] = 1;
    return;
  }
  const int pair = b >> 1, role = b & 1;                 // blocks 2p (producer, XCD 2p % 8) and 2p + 1 (consumer, next XCD)
  float* blk = data + (size_t)pair * 4096;               // 1024 threads x 4 floats
  unsigned long long* f_go = flags + pair * 16, *f_ack = flags + pair * 16 + 8;
  unsigned nbad = 0;
  for (int it = 1; it <= iters; ++it) {
    const float val = (float)it;
    if (role == 0) {
      const f32x4 v = f32x4{val, val, val, val};
      float* p = blk + tid * 4;
      if (ST == 0) *reinterpret_cast<f32x4*>(p) = v;
      if (ST == 1) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, tid * 16, 0, 16);
      }
      if (ST == 2) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 q = __builtin_bit_cast(u64x2, v);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (ST == 3)
        for (int t = 0; t < 4; ++t) __hip_atomic_store(p + t, v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(f_go, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spin = 0;
        while (__hip_atomic_load(f_ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)it && ++spin < (1 << 24)) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    } else {
      if (tid == 0) {
        int spin = 0;
        while (__hip_atomic_load(f_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)it && ++spin < (1 << 24)) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      f32x4 got;
      if (LD == 0 || LD >= 2) {
        if (LD == 0) asm volatile("buffer_inv sc0" ::: "memory");
        if (LD == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (LD == 3) asm volatile("buffer_inv sc0 sc1" ::: "memory");
        if (LD == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (LD == 5) { asm volatile("buffer_inv sc0\n s_waitcnt vmcnt(0)" ::: "memory"); }
        if (LD == 6) got = *(gvp4)(blk + tid * 4);                       // volatile: global_load_dwordx4 ... sc0 sc1
        else got = *reinterpret_cast<const f32x4*>(blk + tid * 4);
      } else {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 0x7fffffff, 0x00020000);
        got = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 0, 16));
      }
      nbad += (got[0] != val || got[1] != val || got[2] != val || got[3] != val) ? 1u : 0u;
      __syncthreads();
      if (tid == 0) __hip_atomic_store(f_ack, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (nbad) atomicAdd(bad, (nbad + 1023u) / 1024u);      // (in units of 1024 thread-hand-overs: no wrap)
}

template <int ST, int LD>
void run(bool uncached, int bg_blocks, int iters) {
  const int pairs = 64;
  float* data; unsigned long long* flags; unsigned* bad; float* bg;
  const size_t bgf = (size_t)64 << 20;
  if (uncached) { CK(hipExtMallocWithFlags((void**)&data, pairs * 4096 * 4, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void**)&flags, pairs * 16 * 8, hipDeviceMallocUncached)); }
  else { CK(hipMalloc(&data, pairs * 4096 * 4)); CK(hipMalloc(&flags, pairs * 16 * 8)); }
  CK(hipMalloc(&bad, 8)); CK(hipMalloc(&bg, bgf * 4));
  CK(hipMemset(data, 0, pairs * 4096 * 4)); CK(hipMemset(flags, 0, pairs * 16 * 8)); CK(hipMemset(bad, 0, 8)); CK(hipMemset(bg, 0, bgf * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k<ST, LD>), dim3(2 * pairs + bg_blocks), dim3(1024), 0, 0, data, flags, pairs, iters, bad, bg, bgf);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h[2]; CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
  static const char* sn[] = {"16 B plain", "16 B sc1", "2 x 8 B agent", "4 x 4 B agent"};
  static const char* ln[] = {"plain, inv sc0", "16 B sc1", "plain, inv sc1", "plain, inv sc0 sc1", "plain, acquire fence", "plain, inv sc0 + wait", "volatile (sc0 sc1)"};
  printf("%-9s store %-14s load %-22s background %3d: %9u K stale 16-byte reads in %d hand-overs x %d pairs x 1024 threads (%.2f us per hand-over)\n",
         uncached ? "uncached" : "cached", sn[ST], ln[LD], bg_blocks, h[0], iters, pairs, ms * 1e3 / iters);
  CK(hipFree(bg)); CK(hipFree(bad));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  for (int bgb : {0, 128}) {
    for (int unc = 1; unc >= 0; --unc) {
      run<2, 0>(unc, bgb, iters);
      run<2, 6>(unc, bgb, iters);
      run<0, 6>(unc, bgb, iters);
      run<1, 6>(unc, bgb, iters);
      run<2, 1>(unc, bgb, iters);
      run<1, 1>(unc, bgb, iters);
      run<0, 1>(unc, bgb, iters);
    }
  }
  return 0;
}
