// p2p.hip — one-shot all-reduce of the gradient arenas over xGMI peer windows.
//
// The data-parallel update exchanges two ~300 KB gradient arenas per step.  Through
// RCCL each is a ring over the eight GPUs, tens of microseconds of latency against a 42 us update.
// xGMI is a full mesh of point-to-point links, so here every rank owns a WINDOW in its own HBM
// (fine-grained memory, exported to the other processes as an IPC handle) with one slot per source
// rank; an exchange is
//     k_p2p_push    my arena -> slot[my rank] of EVERY rank's window (direct stores over xGMI),
//                   then — after the last workgroup's system-scope fence — one sequence-number
//                   flag per destination
//     k_p2p_reduce  wait for the `world` flags of my window, then out[i] = sum over ranks in rank
//                   order (the same order on every rank: replicas stay bit-identical)
// Two window halves alternate by exchange parity: a rank that has finished exchange e+1 has seen
// every peer's flag e+1, which a peer only raises after it has finished reading exchange e.
// The windows are fine-grained allocations: 16-byte stores drained by a system-scope release fence
// before the flags go up, a system-scope acquire after the flags were seen before the slots are read.
// Waits are bounded and poison the result with NaN instead of hanging.
// The RCCL path stays the default whenever the windows cannot be set up or the self-test
// (oprl_p2p_selftest) does not reproduce the expected sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/oprl_amd.h"
#include "p2p.h"
#include "tp3.h"

namespace oprl {

__host__ __device__ inline size_t p2p_flags_off(int world, size_t slot_floats) {
  return (size_t)2 * world * slot_floats * sizeof(float);
}
__host__ __device__ inline size_t p2p_window_bytes(int world, size_t slot_floats) {
  return p2p_flags_off(world, slot_floats) + (size_t)2 * world * kFlagStride * sizeof(unsigned long long);
}

struct P2pPushArgs {
  char* peer[kP2pMaxWorld];
  int world, rank, parity;
  size_t slot_floats, n;                     // n: 4-byte words to send
  const unsigned* src;
  unsigned* done;
  unsigned long long seq;
};

__global__ __launch_bounds__(kP2pThreads) void k_p2p_push(const P2pPushArgs a) {
  // 16-byte stores (the windows are fine-grained memory: coherent at system scope once the fence below
  // has drained them; 4-byte system-scope atomics per word cost 16 us per 300 KB exchange)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const size_t slot = ((size_t)a.parity * a.world + a.rank) * a.slot_floats;
  const size_t n4 = a.n >> 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const u32x4 v = reinterpret_cast<const u32x4*>(a.src)[i];
#pragma unroll
    for (int p = 0; p < kP2pMaxWorld; ++p)
      if (p < a.world) reinterpret_cast<u32x4*>(reinterpret_cast<unsigned*>(a.peer[p]) + slot)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const size_t i = (n4 << 2) + threadIdx.x;
    for (int p = 0; p < a.world; ++p) (reinterpret_cast<unsigned*>(a.peer[p]) + slot)[i] = a.src[i];
  }
  __threadfence_system();                    // this workgroup's words have left before it reports
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {              // last workgroup: everything is out, raise the flags
      __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      const size_t foff = p2p_flags_off(a.world, a.slot_floats);
      for (int p = 0; p < a.world; ++p) {
        unsigned long long* f = reinterpret_cast<unsigned long long*>(a.peer[p] + foff) +
                                ((size_t)a.parity * a.world + a.rank) * kFlagStride;
        __hip_atomic_store(f, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

struct P2pReduceArgs {
  const char* window;
  int world, parity, as_double;
  size_t slot_floats, n;                     // n: elements (floats, or doubles with as_double)
  void* dst;
  unsigned long long seq;
  unsigned* err;
};

__global__ __launch_bounds__(kP2pThreads) void k_p2p_reduce(const P2pReduceArgs a) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) s_ok = 1;
  __syncthreads();
  if ((int)threadIdx.x < a.world) {
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(a.window + p2p_flags_off(a.world, a.slot_floats)) +
                                  ((size_t)a.parity * a.world + threadIdx.x) * kFlagStride;
    bool ok = false;
    for (int spin = 0; spin < (1 << 22) && !ok; ++spin) {
      ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
      if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    if (!ok) { s_ok = 0; report_expired(a.err, (KERN_P2P << 8) | SITE_WINDOW); }   // bounded: a missing peer is reported and poisons the result instead of hanging
  }
  __syncthreads();
  __threadfence_system();
  const bool ok = s_ok != 0;
  const unsigned* base = reinterpret_cast<const unsigned*>(a.window) + (size_t)a.parity * a.world * a.slot_floats;
  if (!a.as_double) {
    // (after the acquire above plain 16-byte loads see the peers' stores)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const size_t n4 = a.n >> 2;
    const float nanv = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
      f32x4v s4 = f32x4v{0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < a.world; ++r)
        s4 += reinterpret_cast<const f32x4v*>(reinterpret_cast<const float*>(base) + (size_t)r * a.slot_floats)[i];
      reinterpret_cast<f32x4v*>(a.dst)[i] = ok ? s4 : f32x4v{nanv, nanv, nanv, nanv};
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
      const size_t i = (n4 << 2) + threadIdx.x;
      float s1 = 0.f;
      for (int r = 0; r < a.world; ++r) s1 += (reinterpret_cast<const float*>(base) + (size_t)r * a.slot_floats)[i];
      reinterpret_cast<float*>(a.dst)[i] = ok ? s1 : nanv;
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
    if (a.as_double) {
      double s = 0.0;
      for (int r = 0; r < a.world; ++r) {
        const unsigned* w = base + (size_t)r * a.slot_floats + 2 * i;
        const unsigned long long lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long hi = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s += __longlong_as_double((long long)(lo | (hi << 32)));
      }
      reinterpret_cast<double*>(a.dst)[i] = ok ? s : __builtin_nan("");
    } else {
      float s = 0.f;
      for (int r = 0; r < a.world; ++r)
        s += __uint_as_float(__hip_atomic_load(base + (size_t)r * a.slot_floats + i, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM));
      reinterpret_cast<float*>(a.dst)[i] = ok ? s : __builtin_nanf("");
    }
  }
}

// Both halves in ONE launch (float arenas): push to the PEERS' windows (this rank's own contribution is
// read straight from `buf`), raise the flags, wait for the peers' flags, sum in rank order.  All
// workgroups are co-resident (<= 64), so waiting inside the kernel cannot starve the workgroup that
// raises the flags.
struct P2pFusedArgs {
  char* peer[kP2pMaxWorld];
  const char* window;
  int world, rank, parity;
  size_t slot_floats, n;
  float* buf;
  unsigned* done;
  unsigned long long seq;
  unsigned* err;
};

__global__ __launch_bounds__(kP2pThreads) void k_p2p_all_reduce(const P2pFusedArgs a) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  __shared__ int s_ok;
  const size_t my_slot = ((size_t)a.parity * a.world + a.rank) * a.slot_floats;
  const size_t n4 = a.n >> 2;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, tstride = (size_t)gridDim.x * blockDim.x;
  const bool tail = blockIdx.x == 0 && threadIdx.x < (a.n & 3);
  const size_t ti = (n4 << 2) + threadIdx.x;
  if (threadIdx.x == 0) s_ok = 1;
  for (size_t i = t0; i < n4; i += tstride) {
    const f32x4v v = reinterpret_cast<const f32x4v*>(a.buf)[i];
#pragma unroll
    for (int p = 0; p < kP2pMaxWorld; ++p)
      if (p < a.world && p != a.rank) reinterpret_cast<f32x4v*>(reinterpret_cast<float*>(a.peer[p]) + my_slot)[i] = v;
  }
  if (tail)
    for (int p = 0; p < a.world; ++p)
      if (p != a.rank) (reinterpret_cast<float*>(a.peer[p]) + my_slot)[ti] = a.buf[ti];
  const size_t foff = p2p_flags_off(a.world, a.slot_floats);
  if (a.world > 1) {
    __threadfence_system();                  // this workgroup's words have left before it reports
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence_system();
        for (int p = 0; p < a.world; ++p)
          if (p != a.rank)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.peer[p] + foff) +
                                   ((size_t)a.parity * a.world + a.rank) * kFlagStride,
                               a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if ((int)threadIdx.x < a.world && (int)threadIdx.x != a.rank) {
      const unsigned long long* f = reinterpret_cast<const unsigned long long*>(a.window + foff) +
                                    ((size_t)a.parity * a.world + threadIdx.x) * kFlagStride;
      bool ok = false;
      for (int spin = 0; spin < (1 << 22) && !ok; ++spin) {
        ok = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
        if (!ok) __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) { s_ok = 0; report_expired(a.err, (KERN_P2P << 8) | SITE_WINDOW); }
    }
    __syncthreads();
    __threadfence_system();
  } else {
    __syncthreads();
  }
  const bool ok = s_ok != 0;
  const float nanv = __builtin_nanf("");
  const float* base = reinterpret_cast<const float*>(a.window) + (size_t)a.parity * a.world * a.slot_floats;
  for (size_t i = t0; i < n4; i += tstride) {
    f32x4v s4 = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < a.world; ++r)      // rank order on every rank: identical sums
      s4 += r == a.rank ? reinterpret_cast<const f32x4v*>(a.buf)[i]
                        : reinterpret_cast<const f32x4v*>(base + (size_t)r * a.slot_floats)[i];
    reinterpret_cast<f32x4v*>(a.buf)[i] = ok ? s4 : f32x4v{nanv, nanv, nanv, nanv};
  }
  if (tail) {
    float s1 = 0.f;
    for (int r = 0; r < a.world; ++r) s1 += r == a.rank ? a.buf[ti] : (base + (size_t)r * a.slot_floats)[ti];
    a.buf[ti] = ok ? s1 : nanv;
  }
}

// A handful of doubles (SAC's temperature gradient): two tagged 8-byte granules per value straight into
// the peers' slots, no fences, one workgroup.
struct P2pF64Args {
  char* peer[kP2pMaxWorld];
  const char* window;
  int world, rank, parity, n;
  size_t slot_floats;
  double* buf;
  unsigned long long seq;
  unsigned* err;
};

__global__ __launch_bounds__(kP2pThreads) void k_p2p_all_reduce_f64(const P2pF64Args a) {
  const int i = threadIdx.x;
  if (i >= a.n) return;
  const unsigned tag = (unsigned)a.seq;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(a.buf[i]);
  // the granules live in the spare words of the (parity, source) flag line: never shared with arena data
  const size_t foff = p2p_flags_off(a.world, a.slot_floats);
  for (int p = 0; p < a.world; ++p) {
    if (p == a.rank) continue;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.peer[p] + foff) +
                              ((size_t)a.parity * a.world + a.rank) * kFlagStride + 2 + 2 * i;
    __hip_atomic_store(dst, ((unsigned long long)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, ((unsigned long long)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  double s = 0.0;
  bool all_ok = true;
  for (int r = 0; r < a.world; ++r) {        // rank order
    if (r == a.rank) { s += a.buf[i]; continue; }
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.window + foff) +
                                    ((size_t)a.parity * a.world + r) * kFlagStride + 2 + 2 * i;
    unsigned long long lo = 0, hi = 0;
    bool ok = false;
    for (int spin = 0; spin < (1 << 22) && !ok; ++spin) {
      lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ok = (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    all_ok = all_ok && ok;
    s += __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
  }
  if (!all_ok) report_expired(a.err, (KERN_P2P << 8) | SITE_WINDOW);
  a.buf[i] = all_ok ? s : __builtin_nan("");
}

// ---- host -------------------------------------------------------------------------------------------
hipError_t p2p_create(P2pState& s, int rank, int world, size_t max_floats, size_t tile_region_bytes, void* handle_out) {
  if (world < 1 || world > kP2pMaxWorld || rank < 0 || rank >= world) return hipErrorInvalidValue;
  s.world = world; s.rank = rank;
  s.slot_floats = (max_floats + 1023) / 1024 * 1024;
  s.tile_off = (p2p_window_bytes(world, s.slot_floats) + 255) / 256 * 256;
  s.tile_bytes = tile_region_bytes;
  s.window_bytes = s.tile_off + tile_region_bytes;
  hipError_t e = hipExtMallocWithFlags((void**)&s.window, s.window_bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return e;
  e = hipMemset(s.window, 0, s.window_bytes);
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&s.done, sizeof(unsigned));
  if (e != hipSuccess) return e;
  e = hipMemset(s.done, 0, sizeof(unsigned));
  if (e != hipSuccess) return e;
  hipIpcMemHandle_t hd;
  e = hipIpcGetMemHandle(&hd, s.window);
  if (e != hipSuccess) return e;
  static_assert(sizeof(hipIpcMemHandle_t) <= OPRL_P2P_HANDLE_BYTES, "IPC handle does not fit");
  memset(handle_out, 0, OPRL_P2P_HANDLE_BYTES);
  memcpy(handle_out, &hd, sizeof hd);
  return hipDeviceSynchronize();
}

hipError_t p2p_connect(P2pState& s, const void* handles) {
  for (int r = 0; r < s.world; ++r) {
    if (r == s.rank) { s.peer[r] = s.window; continue; }
    hipIpcMemHandle_t hd;
    memcpy(&hd, (const char*)handles + (size_t)r * OPRL_P2P_HANDLE_BYTES, sizeof hd);
    hipError_t e = hipIpcOpenMemHandle((void**)&s.peer[r], hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return e;
  }
  s.connected = true;
  return hipSuccess;
}

void p2p_destroy(P2pState& s) {
  for (int r = 0; r < s.world; ++r)
    if (r != s.rank && s.peer[r] != nullptr) (void)hipIpcCloseMemHandle(s.peer[r]);
  if (s.window) (void)hipFree(s.window);
  if (s.done) (void)hipFree(s.done);
  s = P2pState();
}

// in place: buf[0..n) <- sum over ranks (n floats, or n doubles with as_double)
hipError_t p2p_all_reduce(P2pState& s, void* buf, size_t n, bool as_double, hipStream_t st) {
  const size_t words = as_double ? 2 * n : n;
  if (!s.connected || words == 0 || words > s.slot_floats) return hipErrorInvalidValue;
  (void)hipGetLastError();                   // (a stale error of an earlier call must not be read as ours)
  s.seq += 1;
  if (!as_double) {
    P2pFusedArgs fa;
    for (int r = 0; r < kP2pMaxWorld; ++r) fa.peer[r] = r < s.world ? s.peer[r] : nullptr;
    fa.window = s.window; fa.world = s.world; fa.rank = s.rank; fa.parity = (int)(s.seq & 1);
    fa.slot_floats = s.slot_floats; fa.n = n; fa.buf = (float*)buf; fa.done = s.done; fa.seq = s.seq; fa.err = s.err;
    const size_t want = ((n >> 2) + kP2pThreads - 1) / kP2pThreads;
    const int blocks = (int)(want < 1 ? 1 : (want < (size_t)kP2pBlocks ? want : (size_t)kP2pBlocks));
    hipLaunchKernelGGL(k_p2p_all_reduce, dim3(blocks), dim3(kP2pThreads), 0, st, fa);
    return hipGetLastError();
  }
  if (as_double && 2 + 2 * n <= kFlagStride) {
    P2pF64Args da;
    for (int r = 0; r < kP2pMaxWorld; ++r) da.peer[r] = r < s.world ? s.peer[r] : nullptr;
    da.window = s.window; da.world = s.world; da.rank = s.rank; da.parity = (int)(s.seq & 1); da.n = (int)n;
    da.slot_floats = s.slot_floats; da.buf = (double*)buf; da.seq = s.seq; da.err = s.err;
    hipLaunchKernelGGL(k_p2p_all_reduce_f64, dim3(1), dim3(kP2pThreads), 0, st, da);
    return hipGetLastError();
  }
  P2pPushArgs pa;
  for (int r = 0; r < kP2pMaxWorld; ++r) pa.peer[r] = r < s.world ? s.peer[r] : nullptr;
  pa.world = s.world; pa.rank = s.rank; pa.parity = (int)(s.seq & 1);
  pa.slot_floats = s.slot_floats; pa.n = words; pa.src = (const unsigned*)buf; pa.done = s.done; pa.seq = s.seq;
  const int blocks = (int)((words + kP2pThreads - 1) / kP2pThreads < kP2pBlocks ? (words + kP2pThreads - 1) / kP2pThreads : kP2pBlocks);
  hipLaunchKernelGGL(k_p2p_push, dim3(blocks), dim3(kP2pThreads), 0, st, pa);
  P2pReduceArgs ra;
  ra.window = s.window; ra.world = s.world; ra.parity = pa.parity; ra.as_double = as_double ? 1 : 0;
  ra.slot_floats = s.slot_floats; ra.n = n; ra.dst = buf; ra.seq = s.seq; ra.err = s.err;
  const int rblocks = (int)((n + kP2pThreads - 1) / kP2pThreads < kP2pBlocks ? (n + kP2pThreads - 1) / kP2pThreads : kP2pBlocks);
  hipLaunchKernelGGL(k_p2p_reduce, dim3(rblocks), dim3(kP2pThreads), 0, st, ra);
  return hipGetLastError();
}

}  // namespace oprl
