// replay_index.h — flat transition index -> (episode, step) for the device-side samplers
// (k_replay_gather in replay.hip, load_batch in fused_ddpg.hip).
//
// Reference: EpisodicReplayBuffer._inds_to_episodic (buffers/episodic_buffer.py:114-121): the episode of flat
// index `ind` is the first slot whose cumulative end exceeds it (np.argmin over the >= mask; if none does,
// argmin of an all-True mask = slot 0).  A binary search over the table in global memory is ~log2(E)
// dependent round trips; the table is staged in LDS instead — whole when it fits (the reference's 1000
// episodes), otherwise as a COARSE table of every stride-th end, finished by <= log2(stride) + 1 global probes
// (a replay of 200-step episodes has 5000 of them: 5.4 us -> 1.x us for the sampling chain).
#pragma once
#include <hip/hip_runtime.h>

namespace oprl {

struct EndsLds { const int* lds; int stride, n_blocks; };

// all threads of the workgroup; a barrier must follow before find_episode
__device__ __forceinline__ EndsLds stage_ends(const int* __restrict__ ends, int n_eps, int* lds, int lds_cap,
                                              int tid, int n_threads) {
  EndsLds t;
  t.lds = lds;
  t.stride = (n_eps + lds_cap - 1) / lds_cap;
  if (t.stride < 1) t.stride = 1;
  t.n_blocks = (n_eps + t.stride - 1) / t.stride;
  for (int j = tid; j < t.n_blocks; j += n_threads) {
    const int last = min(n_eps, (j + 1) * t.stride) - 1;
    lds[j] = ends[last];
  }
  return t;
}

// first episode e with ends[e] > ind (0 if none); *start = ends[e - 1] (0 for e = 0)
__device__ __forceinline__ int find_episode(const int* __restrict__ ends, int n_eps, const EndsLds& T, long ind,
                                            long* start) {
  int lo = 0, hi = T.n_blocks;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((long)T.lds[mid] > ind) hi = mid; else lo = mid + 1;
  }
  int e;
  if (lo >= T.n_blocks) {
    e = 0;                                            // every end <= ind: the all-True argmin
  } else if (T.stride == 1) {
    e = lo;
  } else {
    int a = lo * T.stride, b = min(n_eps, a + T.stride) - 1;   // ends[b] > ind is known
    while (a < b) {
      const int mid = (a + b) >> 1;
      if ((long)ends[mid] > ind) b = mid; else a = mid + 1;
    }
    e = a;
  }
  if (e == 0) *start = 0;
  else *start = T.stride == 1 ? (long)T.lds[e - 1] : (long)ends[e - 1];
  return e;
}

}  // namespace oprl
