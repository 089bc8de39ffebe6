// batch_rows.h — a 16-row slice of the minibatch into LDS, from caller-supplied rows or gathered from the replay
// with the draw of k_replay_gather (replay.hip), and the same slice written back out as plain rows: what the fused
// phase kernels start with (fused_ddpg.hip), their phase 2's prefetch row, and the riding workgroups that gather
// the NEXT update's rows during a layer-by-layer launch (layerwise.hip, PrefetchJob).
#pragma once
#include "kernels.h"
#include "philox.h"
#include "replay_index.h"

namespace oprl {

constexpr int kMaxEnds = 2048;

// rows [row0, row0+kR) of the minibatch -> xa = [s | a | 0], xb = [s' | 0], r, d (LDS)
__device__ __forceinline__ void load_batch(const BatchSrc& P, int row0, int B, int S, int A,
                                           float* xa, float* xb, float* rS, float* dS, int* meta,
                                           int* endsS) {
  const int tid = threadIdx.x;
  if (P.gather) {
    lds_zero(xa, 2 * kR * kX0Ld);   // xa and xb are adjacent
    const EndsLds ET = stage_ends(P.ends, P.n_eps, endsS, kMaxEnds, tid, kThreads);
    __syncthreads();
    if (tid < kR) {
      const int i = row0 + tid;
      int e = 0, t = 0;
      if (i < B) {
        const u32x4 rnd = philox4x32_10(
            u32x4{(uint32_t)P.counter, (uint32_t)(P.counter >> 32), (uint32_t)i, 0x5a17u},
            (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
        const long ind = (long)bounded_u32(rnd.x, (uint32_t)P.n_transitions);
        long start = 0;
        e = find_episode(P.ends, P.n_eps, ET, ind, &start);
        t = (int)(ind - start);
      }
      meta[tid] = e;
      meta[kR + tid] = t;
    }
    __syncthreads();
    for (int idx = tid; idx < kR * (2 * S + A + 2); idx += kThreads) {
      const int W = 2 * S + A + 2;
      const int row = idx / W, c = idx - row * W;
      if (row0 + row >= B) continue;
      const long e = meta[row], t = meta[kR + row];
      // branch-free: one load and one LDS store per element (an if/else per kind diverges inside
      // a wave and serialises load -> wait -> store)
      const float* src = P.states + (e * (P.L + 1) + t) * S + c;          // s | s' contiguous
      float* dst = xa + row * kX0Ld + c;
      if (c >= S) dst = xb + row * kX0Ld + (c - S);
      if (c >= 2 * S) { src = P.actions + (e * P.L + t) * A + (c - 2 * S); dst = xa + row * kX0Ld + S + (c - 2 * S); }
      if (c == 2 * S + A) { src = P.rewards + e * P.L + t; dst = rS + row; }
      if (c == 2 * S + A + 1) { src = P.dones + e * P.L + t; dst = dS + row; }
      *dst = *src;
    }
  } else {
    // staged rows: every element of the two padded tiles is written once, by one thread — value or zero — with no
    // barrier in between (a __syncthreads here would wait for every load the caller has in flight: the first pass's
    // weight fragments, which the lean passes request BEFORE they call this — r05-14)
    // (unconditional loads — a dummy address where the element is padding — all issued before the first LDS store)
    constexpr int kPer = (2 * kR * kX0Ld + kThreads - 1) / kThreads;
    float v[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int idx = min(tid + j * kThreads, 2 * kR * kX0Ld - 1);
      const int t = idx / (kR * kX0Ld), rem = idx - t * (kR * kX0Ld);     // tile 0: xa = [s | a | 0], 1: xb = [s' | 0]
      const int row = rem / kX0Ld, col = rem - row * kX0Ld, gr = row0 + row;
      const bool is_s = gr < B && col < S, is_a = gr < B && t == 0 && col >= S && col < S + A;
      const float* src = P.s;
      if (is_s) src = (t == 0 ? P.s : P.s2) + (size_t)gr * S + col;
      if (is_a) src = P.a + (size_t)gr * A + (col - S);
      const float x = *src;
      v[j] = (is_s || is_a) ? x : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int idx = tid + j * kThreads;
      if (idx < 2 * kR * kX0Ld) xa[idx] = v[j];
    }
    if (tid < kR) {
      const int gr = row0 + tid;
      rS[tid] = gr < B ? P.r[gr] : 0.f;
      dS[tid] = gr < B ? P.d[gr] : 0.f;
    }
  }
}

// The next update's rows, gathered by one workgroup per 16-row slice and left as plain rows in N.s / a / r / d / s2.
// smem: 2 * kR * kX0Ld + 96 + kMaxEnds floats.  (PrefetchJob: kernels.h)
// (WT: the rows are read by workgroups of the SAME launch behind a flag — k_ddpg_chain: written through)
template <bool WT = false>
__device__ __forceinline__ void prefetch_rows_src(const BatchSrc& N, int S, int Ad, int B, int slice, float* smem);
__device__ __forceinline__ void prefetch_rows_body(const PrefetchJob& J, int slice, float* smem) {
  prefetch_rows_src(J.next, J.S, J.A, J.B, slice, smem);
}
template <bool WT>
__device__ __forceinline__ void prefetch_rows_src(const BatchSrc& N, int S, int Ad, int B, int slice, float* smem) {
  const int tid = threadIdx.x, row0 = slice * kR;
  float* xa = smem;
  float* xb = xa + kR * kX0Ld;
  float* rS = xb + kR * kX0Ld;
  float* dS = rS + kR;
  int* meta = reinterpret_cast<int*>(dS + 2 * kR);
  int* endsS = reinterpret_cast<int*>(rS + 96);
  load_batch(N, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
  __syncthreads();
  if constexpr (WT) {
    store_rows_wt(xa, kX0Ld, const_cast<float*>(N.s), S, S, row0, B);
    store_rows_wt(xb, kX0Ld, const_cast<float*>(N.s2), S, S, row0, B);
    store_rows_wt(xa + S, kX0Ld, const_cast<float*>(N.a), Ad, Ad, row0, B);
    if (tid < kR && row0 + tid < B) {
      __hip_atomic_store(const_cast<float*>(N.r) + row0 + tid, rS[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(const_cast<float*>(N.d) + row0 + tid, dS[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  store_rows(xa, kX0Ld, const_cast<float*>(N.s), S, S, row0, B);
  store_rows(xb, kX0Ld, const_cast<float*>(N.s2), S, S, row0, B);
  for (int idx = tid; idx < kR * Ad; idx += kThreads) {
    const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
    if (gr < B) const_cast<float*>(N.a)[(size_t)gr * Ad + col] = xa[row * kX0Ld + S + col];
  }
  if (tid < kR && row0 + tid < B) {
    const_cast<float*>(N.r)[row0 + tid] = rS[tid];
    const_cast<float*>(N.d)[row0 + tid] = dS[tid];
  }
}

// The same rows without the LDS tile, for a workgroup of `nt` threads (a rider of the 512-thread dW launch): element
// by element from the replay to N.s / a / r / d / s2 — the copies load_batch + the stores above make, so the same rows
// bit for bit.  lds_i: kMaxEnds + 2 * kR ints.
__device__ __forceinline__ void prefetch_rows_direct(const PrefetchJob& J, int slice, int* lds_i, int nt) {
  const int tid = threadIdx.x, row0 = slice * kR, S = J.S, Ad = J.A, B = J.B;
  const BatchSrc& P = J.next;
  int* endsS = lds_i;
  int* meta = lds_i + kMaxEnds;
  const EndsLds ET = stage_ends(P.ends, P.n_eps, endsS, kMaxEnds, tid, nt);
  __syncthreads();
  if (tid < kR) {
    const int i = row0 + tid;
    int e = 0, t = 0;
    if (i < B) {
      const u32x4 rnd = philox4x32_10(u32x4{(uint32_t)P.counter, (uint32_t)(P.counter >> 32), (uint32_t)i, 0x5a17u},
                                      (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
      const long ind = (long)bounded_u32(rnd.x, (uint32_t)P.n_transitions);
      long start = 0;
      e = find_episode(P.ends, P.n_eps, ET, ind, &start);
      t = (int)(ind - start);
    }
    meta[tid] = e;
    meta[kR + tid] = t;
  }
  __syncthreads();
  const int W = 2 * S + Ad + 2;
  for (int idx = tid; idx < kR * W; idx += nt) {
    const int row = idx / W, c = idx - row * W, gr = row0 + row;
    if (gr >= B) continue;
    const long e = meta[row], t = meta[kR + row];
    const float* src = P.states + (e * (P.L + 1) + t) * S + c;          // s | s' contiguous
    float* dst = const_cast<float*>(P.s) + (size_t)gr * S + c;
    if (c >= S) dst = const_cast<float*>(P.s2) + (size_t)gr * S + (c - S);
    if (c >= 2 * S) { src = P.actions + (e * P.L + t) * Ad + (c - 2 * S); dst = const_cast<float*>(P.a) + (size_t)gr * Ad + (c - 2 * S); }
    if (c == 2 * S + Ad) { src = P.rewards + e * P.L + t; dst = const_cast<float*>(P.r) + gr; }
    if (c == 2 * S + Ad + 1) { src = P.dones + e * P.L + t; dst = const_cast<float*>(P.d) + gr; }
    *dst = *src;
  }
}

}  // namespace oprl
