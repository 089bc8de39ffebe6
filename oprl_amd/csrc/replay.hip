// replay.hip — HBM-resident episodic replay: batched transition writes and the
// uniform sampler as a gather kernel.
//
// Reference: buffers/episodic_buffer.py
//   storage layout              :29-55   (states[E,L+1,S] actions[E,L,A] rewards/dones[E,L,1])
//   add_transition data movement:81-96
//   flat index -> (episode,step):114-121 (first episode whose cumulative end > index)
//   sample                      :123-133 (5 advanced-index gathers)
//
// Because states is [E, L+1, S], state (e,t) and next_state (e,t+1) are ADJACENT
// rows: one sample reads a single contiguous 2·S-float run, plus A + 1 + 1 floats.
// A workgroup stages 32 samples' rows in LDS and writes the five output arrays
// fully coalesced.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/oprl_amd.h"
#include "philox.h"
#include "replay_index.h"

namespace oprl {
void set_err(const char* fmt, ...);
void prof_begin(int kind, hipStream_t st);
void prof_end(hipStream_t st);
}
using oprl::set_err;

#define HIPC(x)                                                              \
  do {                                                                       \
    hipError_t _e = (x);                                                     \
    if (_e != hipSuccess) {                                                  \
      set_err("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
      return OPRL_ERR_HIP;                                                   \
    }                                                                        \
  } while (0)

namespace {

constexpr int kGatherThreads = 256;
constexpr int kSamplesPerWg = 16;   // more workgroups, one copy round per thread at walker dims
constexpr int kStageRows = 4096;
constexpr int kMaxEndsLds = 2048; // episode-end table entries staged in LDS by the gather  // transitions staged on the host between flushes

struct GatherArgs {
  const float *states, *actions, *rewards, *dones;
  const int* ends;  // cumulative episode ends, [n_eps]
  int n_eps, L, S, A, B;
  long n_transitions;
  const long long* idx;  // or null
  unsigned long long seed, counter;
  float *out_s, *out_a, *out_r, *out_d, *out_s2;
  int *out_ep, *out_step;
};

__global__ __launch_bounds__(kGatherThreads) void k_replay_gather(const GatherArgs G) {
  extern __shared__ float stage[];  // [kSamplesPerWg][2S + A + 2]
  __shared__ int s_ep[kSamplesPerWg], s_t[kSamplesPerWg];
  __shared__ int s_ends[kMaxEndsLds];
  const int tid = threadIdx.x;
  // the episode table goes to LDS in one coalesced pass (whole, or every stride-th end: replay_index.h); a
  // binary search over global memory is 10+ dependent round trips to L2/HBM (measured 7.4 us for this kernel)
  const oprl::EndsLds ET = oprl::stage_ends(G.ends, G.n_eps, s_ends, kMaxEndsLds, tid, kGatherThreads);
  __syncthreads();
  const int base = blockIdx.x * kSamplesPerWg;
  const int S = G.S, A = G.A, W = 2 * S + A + 2;
  if (tid < kSamplesPerWg) {
    const int i = base + tid;
    int e = 0, t = 0;
    if (i < G.B) {
      long ind;
      if (G.idx != nullptr) {
        ind = (long)G.idx[i];
      } else {
        const oprl::u32x4 r = oprl::philox4x32_10(
            oprl::u32x4{(uint32_t)G.counter, (uint32_t)(G.counter >> 32), (uint32_t)i, 0x5a17u},
            (uint32_t)G.seed, (uint32_t)(G.seed >> 32));
        ind = (long)oprl::bounded_u32(r.x, (uint32_t)G.n_transitions);
      }
      // first episode with ends[e] > ind  (np.argmin over the >= mask; all-True -> 0)
      long start = 0;
      e = oprl::find_episode(G.ends, G.n_eps, ET, ind, &start);
      t = (int)(ind - start);
      if (G.out_ep != nullptr) G.out_ep[i] = e;
      if (G.out_step != nullptr) G.out_step[i] = t;
    }
    s_ep[tid] = e;
    s_t[tid] = t;
  }
  __syncthreads();
  const int n_here = min(kSamplesPerWg, G.B - base);
  // Rows -> LDS.  Branch-free source selection and all of a thread's loads issued before its LDS
  // stores: an if/else per kind (s|s', a, r, d) diverges inside a wave and turns the copy into several
  // serialised load -> wait -> store round trips (the kernel is pure latency: 8.5 us at B = 256 before).
  constexpr int kU = 4;
  for (int i0 = 0; i0 < n_here * W; i0 += kU * kGatherThreads) {
    float v[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int idx = min(i0 + j * kGatherThreads + tid, n_here * W - 1);
      const int smp = idx / W, c = idx - smp * W;
      const long e = s_ep[smp], t = s_t[smp];
      const float* src = G.states + (e * (G.L + 1) + t) * S + c;                       // s | s' contiguous
      if (c >= 2 * S) src = G.actions + (e * G.L + t) * A + (c - 2 * S);
      if (c == 2 * S + A) src = G.rewards + e * G.L + t;
      if (c == 2 * S + A + 1) src = G.dones + e * G.L + t;
      v[j] = *src;
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int idx = i0 + j * kGatherThreads + tid;
      if (idx < n_here * W) stage[idx] = v[j];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < n_here * S; idx += kGatherThreads) {
    const int smp = idx / S, c = idx - smp * S;
    G.out_s[(size_t)(base + smp) * S + c] = stage[smp * W + c];
    G.out_s2[(size_t)(base + smp) * S + c] = stage[smp * W + S + c];
  }
  for (int idx = tid; idx < n_here * A; idx += kGatherThreads) {
    const int smp = idx / A, c = idx - smp * A;
    G.out_a[(size_t)(base + smp) * A + c] = stage[smp * W + 2 * S + c];
  }
  if (tid < n_here) {
    G.out_r[base + tid] = stage[tid * W + 2 * S + A];
    G.out_d[base + tid] = stage[tid * W + 2 * S + A + 1];
  }
}

// staged row: [ep, t] as two ints bit-cast into floats, then s[S], a[A], r, d
__global__ void k_replay_scatter(const float* rows, int n, int rowlen, float* states,
                                 float* actions, float* rewards, float* dones, int L, int S,
                                 int A) {
  for (int rix = blockIdx.x; rix < n; rix += gridDim.x) {
    const float* row = rows + (size_t)rix * rowlen;
    const long e = __float_as_int(row[0]), t = __float_as_int(row[1]);
    for (int c = threadIdx.x; c < S + A + 2; c += blockDim.x) {
      const float v = row[2 + c];
      if (c < S) states[(e * (L + 1) + t) * S + c] = v;
      else if (c < S + A) actions[(e * L + t) * A + (c - S)] = v;
      else if (c == S + A) rewards[e * L + t] = v;
      else dones[e * L + t] = v;
    }
  }
}

// The per-env-step form of the same (the trainer loop adds ONE transition between two updates): rows AND the changed
// tail of the episode-ends table straight from the pinned staging buffers (host-mapped: a few hundred bytes over the
// link inside the kernel) — one launch instead of two H2D copies and a scatter launch.
__global__ void k_replay_ingest(const float* rows, int n, int rowlen, float* states, float* actions, float* rewards,
                                float* dones, int L, int S, int A, const int* ends_src, int* ends_dst, int ends_first,
                                int ends_n) {
  for (int rix = blockIdx.x; rix < n; rix += gridDim.x) {
    const float* row = rows + (size_t)rix * rowlen;
    const long e = __float_as_int(row[0]), t = __float_as_int(row[1]);
    for (int c = threadIdx.x; c < S + A + 2; c += blockDim.x) {
      const float v = row[2 + c];
      if (c < S) states[(e * (L + 1) + t) * S + c] = v;
      else if (c < S + A) actions[(e * L + t) * A + (c - S)] = v;
      else if (c == S + A) rewards[e * L + t] = v;
      else dones[e * L + t] = v;
    }
  }
  for (int i = ends_first + (int)(blockIdx.x * blockDim.x + threadIdx.x); i < ends_n; i += (int)(gridDim.x * blockDim.x))
    ends_dst[i] = ends_src[i];
}
constexpr int kDirectRows = 16;       // staged rows / changed table entries up to which the ingest kernel reads the pinned buffers itself
constexpr int kDirectEnds = 2048;

}  // namespace

struct oprl_replay {
  int E, L, S, A;
  float *states, *actions, *rewards, *dones;
  int* ends_dev = nullptr;
  int n_eps = 0;
  long n_transitions = 0;
  // double-buffered pinned staging for add_transition rows and for ends uploads
  int rowlen = 0;
  float* stage_host[2] = {nullptr, nullptr};
  float* stage_dev[2] = {nullptr, nullptr};
  int* ends_host[2] = {nullptr, nullptr};
  hipEvent_t stage_ev[2], ends_ev[2];
  bool stage_busy[2] = {false, false}, ends_busy[2] = {false, false};
  int cur = 0, ends_cur = 0, n_staged = 0;
  // the device copies of the pinned buffers' addresses, and the ends-table upload oprl_replay_set_lens left for the
  // next flush: entries [first, n) of ends_host[ends_cur] differ from what the device holds (ends_last = its mirror)
  float* stage_map[2] = {nullptr, nullptr};
  int* ends_map[2] = {nullptr, nullptr};
  std::vector<int> ends_last;
  bool ends_pending = false;
  int ends_first = 0, ends_n = 0;
  // the stream of the caller's most recent flush / sample / block write / table upload: where a staging buffer
  // that fills up inside oprl_replay_write (which takes no stream) is flushed, so that the scatter stays ordered
  // with the caller's later gathers
};

extern "C" int oprl_replay_create(int32_t n_episodes, int32_t max_ep_len, int32_t state_dim,
                                  int32_t action_dim, float* states, float* actions,
                                  float* rewards, float* dones, oprl_replay** out) {
  if (!out || n_episodes < 1 || max_ep_len < 1 || state_dim < 1 || action_dim < 1 || !states ||
      !actions || !rewards || !dones) {
    set_err("oprl_replay_create: invalid argument");
    return OPRL_ERR_INVALID;
  }
  auto* h = new oprl_replay();
  h->E = n_episodes; h->L = max_ep_len; h->S = state_dim; h->A = action_dim;
  h->states = states; h->actions = actions; h->rewards = rewards; h->dones = dones;
  h->rowlen = 2 + state_dim + action_dim + 2;
  HIPC(hipMalloc(&h->ends_dev, sizeof(int) * n_episodes));
  for (int i = 0; i < 2; ++i) {
    HIPC(hipHostMalloc(&h->stage_host[i], sizeof(float) * h->rowlen * kStageRows));
    HIPC(hipMalloc(&h->stage_dev[i], sizeof(float) * h->rowlen * kStageRows));
    HIPC(hipHostMalloc(&h->ends_host[i], sizeof(int) * n_episodes));
    HIPC(hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&h->ends_ev[i], hipEventDisableTiming));
    if (hipHostGetDevicePointer((void**)&h->stage_map[i], h->stage_host[i], 0) != hipSuccess) h->stage_map[i] = nullptr;
    if (hipHostGetDevicePointer((void**)&h->ends_map[i], h->ends_host[i], 0) != hipSuccess) h->ends_map[i] = nullptr;
  }
  (void)hipGetLastError();
  *out = h;
  return OPRL_OK;
}

extern "C" int oprl_replay_destroy(oprl_replay* h) {
  if (!h) return OPRL_OK;
  (void)hipDeviceSynchronize();
  (void)hipFree(h->ends_dev);
  for (int i = 0; i < 2; ++i) {
    (void)hipHostFree(h->stage_host[i]);
    (void)hipFree(h->stage_dev[i]);
    (void)hipHostFree(h->ends_host[i]);
    (void)hipEventDestroy(h->stage_ev[i]);
    (void)hipEventDestroy(h->ends_ev[i]);
  }
  delete h;
  return OPRL_OK;
}

extern "C" int oprl_replay_flush(oprl_replay* h, void* stream) {
  if (!h) { set_err("null replay handle"); return OPRL_ERR_INVALID; }
  if (h->n_staged == 0 && !h->ends_pending) return OPRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const int c = h->cur, n = h->n_staged, ec = h->ends_cur;
  const int e_first = h->ends_pending ? h->ends_first : 0, e_n = h->ends_pending ? h->ends_n : 0;
  const bool direct = n <= kDirectRows && e_n - e_first <= kDirectEnds && h->stage_map[c] != nullptr && h->ends_map[ec] != nullptr;
  if (direct) {
    const int wgs = n > 0 ? n : 1;
    hipLaunchKernelGGL(k_replay_ingest, dim3(wgs), dim3(64), 0, st, h->stage_map[c], n, h->rowlen, h->states, h->actions,
                       h->rewards, h->dones, h->L, h->S, h->A, h->ends_map[ec], h->ends_dev, e_first, e_n);
    HIPC(hipGetLastError());
  } else {
    if (n > 0) {
      HIPC(hipMemcpyAsync(h->stage_dev[c], h->stage_host[c], sizeof(float) * h->rowlen * n, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_replay_scatter, dim3(n < 1024 ? n : 1024), dim3(64), 0, st,
                         h->stage_dev[c], n, h->rowlen, h->states, h->actions, h->rewards, h->dones,
                         h->L, h->S, h->A);
      HIPC(hipGetLastError());
    }
    if (e_n > e_first)
      HIPC(hipMemcpyAsync(h->ends_dev + e_first, h->ends_host[ec] + e_first, sizeof(int) * (e_n - e_first),
                          hipMemcpyHostToDevice, st));
  }
  if (n > 0) {
    HIPC(hipEventRecord(h->stage_ev[c], st));
    h->stage_busy[c] = true;
    h->cur ^= 1;
    h->n_staged = 0;
    if (h->stage_busy[h->cur]) {  // the other buffer must have drained before we refill it
      HIPC(hipEventSynchronize(h->stage_ev[h->cur]));
      h->stage_busy[h->cur] = false;
    }
  }
  if (h->ends_pending) {
    HIPC(hipEventRecord(h->ends_ev[ec], st));
    h->ends_busy[ec] = true;
    h->ends_cur ^= 1;
    h->ends_pending = false;
  }
  return OPRL_OK;
}

extern "C" int oprl_replay_write(oprl_replay* h, int32_t ep, int32_t t, const float* state_host,
                                 const float* action_host, float reward, float done) {
  if (!h || !state_host || !action_host) { set_err("oprl_replay_write: null argument"); return OPRL_ERR_INVALID; }
  if (ep < 0 || ep >= h->E || t < 0 || t >= h->L) {
    set_err("oprl_replay_write: slot (%d,%d) outside [%d,%d)", ep, t, h->E, h->L);
    return OPRL_ERR_INVALID;
  }
  if (h->n_staged == kStageRows) {
    // staging is full and this entry point has no stream argument: the scatter goes on the null stream and is
    // WAITED for — no stream handle of an earlier call is kept (it may have been destroyed, or not be the stream of the
    // next sample), and whatever stream that sample runs on finds the rows in HBM.  Once per kStageRows writes.
    int rc = oprl_replay_flush(h, nullptr);
    if (rc != OPRL_OK) return rc;
    HIPC(hipStreamSynchronize(nullptr));
  }
  float* row = h->stage_host[h->cur] + (size_t)h->n_staged * h->rowlen;
  memcpy(row, &ep, 4);
  memcpy(row + 1, &t, 4);
  memcpy(row + 2, state_host, sizeof(float) * h->S);
  memcpy(row + 2 + h->S, action_host, sizeof(float) * h->A);
  row[2 + h->S + h->A] = reward;
  row[3 + h->S + h->A] = done;
  ++h->n_staged;
  return OPRL_OK;
}

// The trainer loop's add_transition as ONE call: stage the row, take the episode table, and put both on their way
// (oprl_replay_write + oprl_replay_set_lens + oprl_replay_flush).
extern "C" int oprl_replay_write_flush(oprl_replay* h, int32_t ep, int32_t t, const float* state_host,
                                       const float* action_host, float reward, float done, const int32_t* ep_lens_host,
                                       int32_t episodes_counter, void* stream) {
  int rc = oprl_replay_write(h, ep, t, state_host, action_host, reward, done);
  if (rc != OPRL_OK) return rc;
  rc = oprl_replay_set_lens(h, ep_lens_host, episodes_counter, stream);
  if (rc != OPRL_OK) return rc;
  return oprl_replay_flush(h, stream);
}

// n consecutive steps [t0, t0 + n) of episode `ep` from host records [s (S) | a (A) | r | d | ...] that are
// `row_stride` floats apart — a whole episode (or a drained ring segment) in ONE call instead of n
// oprl_replay_write calls (the learner ranks of the distributed setup take in tens of thousands of
// transitions per second).  Staging that fills up is flushed on `stream`.
extern "C" int oprl_replay_write_block(oprl_replay* h, int32_t ep, int32_t t0, int32_t n, const float* rows_host,
                                       int32_t row_stride, void* stream) {
  if (!h || !rows_host || n < 0) { set_err("oprl_replay_write_block: invalid argument"); return OPRL_ERR_INVALID; }
  if (ep < 0 || ep >= h->E || t0 < 0 || t0 + n > h->L || row_stride < h->S + h->A + 2) {
    set_err("oprl_replay_write_block: steps [%d,%d) of episode %d outside [%d,%d) x [0,%d) or stride %d too small",
            t0, t0 + n, ep, 0, h->E, h->L, row_stride);
    return OPRL_ERR_INVALID;
  }
  for (int i = 0; i < n; ++i) {
    if (h->n_staged == kStageRows) {
      int rc = oprl_replay_flush(h, stream);
      if (rc != OPRL_OK) return rc;
    }
    float* row = h->stage_host[h->cur] + (size_t)h->n_staged * h->rowlen;
    const int t = t0 + i;
    memcpy(row, &ep, 4);
    memcpy(row + 1, &t, 4);
    memcpy(row + 2, rows_host + (size_t)i * row_stride, sizeof(float) * (h->S + h->A + 2));
    ++h->n_staged;
  }
  return OPRL_OK;
}

extern "C" int oprl_replay_set_lens(oprl_replay* h, const int32_t* ep_lens_host,
                                    int32_t episodes_counter, void* stream) {
  if (!h || !ep_lens_host || episodes_counter < 0 || episodes_counter > h->E) {
    set_err("oprl_replay_set_lens: invalid argument");
    return OPRL_ERR_INVALID;
  }
  (void)stream;      // (the table goes up with the next flush — every reader flushes first — on ITS stream)
  // validate the whole table BEFORE touching the mirrors: a bad entry midway must not leave ends_last ahead of what
  // the device holds (the next valid call would then see "no difference" and never upload those entries)
  for (int i = 0; i < episodes_counter; ++i)
    if (ep_lens_host[i] < 0 || ep_lens_host[i] > h->L) { set_err("ep_lens[%d]=%d out of range", i, ep_lens_host[i]); return OPRL_ERR_INVALID; }
  const int c = h->ends_cur;
  if (!h->ends_pending && h->ends_busy[c]) { HIPC(hipEventSynchronize(h->ends_ev[c])); h->ends_busy[c] = false; }
  long acc = 0;
  int first = -1;
  if ((int)h->ends_last.size() < episodes_counter) h->ends_last.resize(episodes_counter, -1);
  for (int i = 0; i < episodes_counter; ++i) {
    acc += ep_lens_host[i];
    h->ends_host[c][i] = (int)acc;
    if (h->ends_last[i] != (int)acc) {
      if (first < 0) first = i;
      h->ends_last[i] = (int)acc;
    }
  }
  if (first >= 0) {
    h->ends_first = h->ends_pending ? (first < h->ends_first ? first : h->ends_first) : first;
    h->ends_n = h->ends_pending ? (episodes_counter > h->ends_n ? episodes_counter : h->ends_n) : episodes_counter;
    h->ends_pending = true;
  }
  h->n_eps = episodes_counter;
  h->n_transitions = acc;
  return OPRL_OK;
}

extern "C" int oprl_replay_sample(oprl_replay* h, int32_t B, const int64_t* idx, uint64_t seed,
                                  uint64_t counter, float* out_s, float* out_a, float* out_r,
                                  float* out_d, float* out_s2, int32_t* out_ep, int32_t* out_step,
                                  void* stream) {
  if (!h || B < 1 || !out_s || !out_a || !out_r || !out_d || !out_s2) {
    set_err("oprl_replay_sample: invalid argument");
    return OPRL_ERR_INVALID;
  }
  if (h->n_transitions <= 0 || h->n_eps <= 0) {
    set_err("oprl_replay_sample: buffer is empty (np.random.randint(0, 0) raises in the reference)");
    return OPRL_ERR_STATE;
  }
  int rc = oprl_replay_flush(h, stream);
  if (rc != OPRL_OK) return rc;
  GatherArgs G;
  G.states = h->states; G.actions = h->actions; G.rewards = h->rewards; G.dones = h->dones;
  G.ends = h->ends_dev; G.n_eps = h->n_eps; G.L = h->L; G.S = h->S; G.A = h->A; G.B = B;
  G.n_transitions = h->n_transitions;
  G.idx = (const long long*)idx; G.seed = seed; G.counter = counter;
  G.out_s = out_s; G.out_a = out_a; G.out_r = out_r; G.out_d = out_d; G.out_s2 = out_s2;
  G.out_ep = out_ep; G.out_step = out_step;
  const int grid = (B + kSamplesPerWg - 1) / kSamplesPerWg;
  const size_t lds = sizeof(float) * kSamplesPerWg * (2 * h->S + h->A + 2);
  oprl::prof_begin(2, (hipStream_t)stream);
  hipLaunchKernelGGL(k_replay_gather, dim3(grid), dim3(kGatherThreads), lds, (hipStream_t)stream, G);
  oprl::prof_end((hipStream_t)stream);
  HIPC(hipGetLastError());
  return OPRL_OK;
}

// used by the learner's fused step_n
namespace oprl {
int replay_dims(const oprl_replay* h, int* S, int* A) { *S = h->S; *A = h->A; return 0; }
// raw view for kernels that gather in-place (fused_ddpg.hip)
int replay_view(const oprl_replay* h, const float** states, const float** actions,
                const float** rewards, const float** dones, const int** ends, int* n_eps, int* L,
                long* n_transitions) {
  *states = h->states; *actions = h->actions; *rewards = h->rewards; *dones = h->dones;
  *ends = h->ends_dev; *n_eps = h->n_eps; *L = h->L; *n_transitions = h->n_transitions;
  return 0;
}
}
