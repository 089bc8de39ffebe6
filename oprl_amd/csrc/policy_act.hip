// The per-env-step policy call of the trainer loop (reference trainers/base_trainer.py:38-74:
// actor.explore(state) once per environment step, right after the update of the step before).
//
// One observation through the actor, as a launch that is enqueued BEHIND the update in the same C call
// (oprl_learner_step_act) and talks to the host through host-mapped pinned memory: it reads the observation from
// there and writes the output row there as {ticket, value} granules — no H2D / D2H copies, no stream synchronise; the host
// spins on the granules' tickets (oprl_learner_act_wait).  A 1-row forward is a GEMV: it reads the row-major MASTER weights
// (fp32 FMAs, no MFMA tile to fill, no packs — which a PrecX2 learner does not even keep current in fp32), one
// workgroup, a wave per output neuron, 64 lanes across the inputs.
//
// The weights were written by the update's tile workgroups a moment ago: every load is a trip to memory (~2 us).  The
// three-layer form (the reference's policies: two hidden layers) therefore requests ALL weights of ALL layers before
// it touches the first — 84 to 100 registers per lane — and pays that trip once; other shapes go layer by layer.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace oprl {

__device__ __forceinline__ float act_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ void act_put(const PolicyActArgs& A, int n, float v) {
  // {ticket, value}: the value is its own flag, so no fence orders it against a separate ticket (a system-scope release
  // after the update's tiles would write back everything they left dirty in the L2s)
  __hip_atomic_store(A.out + n, ((unsigned long long)A.ticket_value << 32) | (unsigned long long)__float_as_uint(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dims: [K0 <= 64 J0] -> N0 <= 256 -> N1 <= 256 -> N2 <= 16.  Wave w owns output rows w, w + 16, ... of every layer.
template <int J0>
__global__ __launch_bounds__(1024) void k_policy_act3(const PolicyActArgs A) {
  __shared__ float xs[2][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K0 = A.dims[0], N0 = A.dims[1], N1 = A.dims[2], N2 = A.dims[3];
  float w0[16][J0], w1[16][4], w2[4];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n = wave + 16 * q;
#pragma unroll
    for (int j = 0; j < J0; ++j) {
      const int k = lane + 64 * j;
      w0[q][j] = (n < N0 && k < K0) ? A.w[0][(size_t)n * K0 + k] : 0.f;
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int n = wave + 16 * q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane + 64 * j;
      w1[q][j] = (n < N1 && k < N0) ? A.w[1][(size_t)n * N0 + k] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = lane + 64 * j;
    w2[j] = (wave < N2 && k < N1) ? A.w[2][(size_t)wave * N1 + k] : 0.f;
  }
  // the biases go through LDS (thread = neuron): 32 registers less per lane
  __shared__ float bs[2][256];
  if (tid < 256) bs[0][tid] = tid < N0 ? A.b[0][tid] : 0.f;
  else if (tid < 512) bs[1][tid - 256] = tid - 256 < N1 ? A.b[1][tid - 256] : 0.f;
  const float b2 = wave < N2 ? A.b[2][wave] : 0.f;
  if (tid < 256) xs[0][tid] = tid < K0 ? A.obs[tid] : 0.f;      // (host-mapped: one trip over the link, beside the weights')
  __syncthreads();
  {
    float x[J0];
#pragma unroll
    for (int j = 0; j < J0; ++j) x[j] = xs[0][(lane + 64 * j) & 255];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < J0; ++j) acc = fmaf(w0[q][j], x[j], acc);
      const int n = wave + 16 * q;
      acc = act_wave_sum(acc) + bs[0][n & 255];
      if (lane == 0 && n < 256) xs[1][n] = (n < N0 && acc > 0.f) ? acc : 0.f;
    }
  }
  __syncthreads();
  {
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = xs[1][lane + 64 * j];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = fmaf(w1[q][j], x[j], acc);
      const int n = wave + 16 * q;
      acc = act_wave_sum(acc) + bs[1][n & 255];
      if (lane == 0 && n < 256) xs[0][n] = (n < N1 && acc > 0.f) ? acc : 0.f;
    }
  }
  __syncthreads();
  if (wave < N2) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = fmaf(w2[j], xs[0][lane + 64 * j], acc);
    acc = act_wave_sum(acc) + b2;
    if (lane == 0) act_put(A, wave, acc);      // raw output row (the caller applies tanh / the Gaussian head)
  }
}

// any other shape (widths <= kPolicyActMaxWidth): layer by layer, four output rows of a wave in flight at a time
__global__ __launch_bounds__(1024) void k_policy_act(const PolicyActArgs A) {
  __shared__ float xs[2][kPolicyActMaxWidth];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < A.dims[0]; k += 1024) xs[0][k] = A.obs[k];
  __syncthreads();
  int cur = 0;
  for (int l = 0; l < A.n_layers; ++l) {
    const int K = A.dims[l], N = A.dims[l + 1];
    const float* W = A.w[l];
    const float* bias = A.b[l];
    const float* x = xs[cur];
    const bool last = l == A.n_layers - 1;
    for (int n0 = wave; n0 < N; n0 += 64) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 16 * q;
        if (n < N) {
          const float* row = W + (size_t)n * K;
          for (int k = lane; k < K; k += 64) acc[q] = fmaf(row[k], x[k], acc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = act_wave_sum(acc[q]);
        const int n = n0 + 16 * q;
        if (lane == 0 && n < N) {
          const float y = v + bias[n];
          if (last) act_put(A, n, y);
          else xs[cur ^ 1][n] = y > 0.f ? y : 0.f;
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

hipError_t launch_policy_act(const PolicyActArgs& a, hipStream_t st) {
  const bool three = a.n_layers == 3 && a.dims[0] <= 128 && a.dims[1] <= 256 && a.dims[2] <= 256 && a.dims[3] <= 16;
  if (three && a.dims[0] <= 64) hipLaunchKernelGGL(k_policy_act3<1>, dim3(1), dim3(1024), 0, st, a);
  else if (three) hipLaunchKernelGGL(k_policy_act3<2>, dim3(1), dim3(1024), 0, st, a);
  else hipLaunchKernelGGL(k_policy_act, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}

}  // namespace oprl
