// learner.hip — host orchestration of update() for DDPG / TD3 / SAC / TQC and the
// C-ABI of include/oprl_amd.h.  Each update() is a short fixed sequence of
// k_mlp_slice / k_dw_adam launches on the caller's stream; no host sync inside.
//
// Order of operations follows the reference exactly (it matters: the actor
// loss sees the post-Adam critic, Polyak sees both updated nets):
//   DDPG  algos/ddpg.py:61-107      TD3  algos/td3.py:71-146
//   SAC   algos/sac.py:75-155       TQC  algos/tqc.py:116-189
#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/oprl_amd.h"
#include "kernels.h"
#include "p2p.h"

namespace oprl {

static thread_local std::string g_err;
void set_err(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---- HIP-event profiler (off by default; zero cost when off) ----------------
struct Prof {
  bool on = false;
  std::vector<hipEvent_t> ev;      // pairs
  std::vector<int> kind;
  long counts[OPRL_PROFILE_KINDS] = {0, 0, 0, 0, 0, 0};
  double ms[OPRL_PROFILE_KINDS] = {0, 0, 0, 0, 0, 0};
};
static Prof g_prof;
static const size_t kProfMaxPairs = 1 << 16;

void prof_fold() {
  if (g_prof.ev.empty()) return;
  (void)hipDeviceSynchronize();
  for (size_t i = 0; i + 1 < g_prof.ev.size(); i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]) == hipSuccess) {
      g_prof.counts[g_prof.kind[i / 2]] += 1;
      g_prof.ms[g_prof.kind[i / 2]] += t;
    }
    (void)hipEventDestroy(g_prof.ev[i]);
    (void)hipEventDestroy(g_prof.ev[i + 1]);
  }
  g_prof.ev.clear();
  g_prof.kind.clear();
}

void prof_begin(int kind, hipStream_t st) {
  if (!g_prof.on) return;
  if (g_prof.ev.size() >= 2 * kProfMaxPairs) prof_fold();
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
  g_prof.ev.push_back(a);
  g_prof.ev.push_back(b);
  g_prof.kind.push_back(kind);
  (void)hipEventRecord(a, st);
}

void prof_end(hipStream_t st) {
  if (!g_prof.on || g_prof.ev.empty()) return;
  (void)hipEventRecord(g_prof.ev.back(), st);
}

// (Learners that share a GPU: the fused phase kernels contain bounded cross-workgroup waits that rely on a launch's
// workgroups becoming resident together.  An event chain that serialised the phase launches of all learners of a process
// was measured in round 1 — 8 packed learners 45k -> 17.7k steps/s — and removed in round 3; what protects such runs is
// the clusters-of-four setting (oprl_learner_set_cluster, runners/train.py) and bench.py's verified multi_learner run.)
size_t mlp_slice_lds_bytes(int width, int n_layers);
hipError_t launch_mlp_slice(const MlpArgs& a, int width, hipStream_t st);
hipError_t launch_mlp_slice_multi(const MlpArgs* a, int n, int width, hipStream_t st);
bool mlp_layerwise_ok(const MlpArgs* a, int n, int width);
hipError_t launch_mlp_layerwise(const MlpArgs* a, int n, int width, int n_cus, hipStream_t st, int prec, const TqcJob* job,
                                const MlpArgs* rider, bool first_done, const MlpArgs* tail, int tail_n, int tail0, int tail_prec,
                                const PrefetchJob* prefetch, const LwPairBuf* pairs);
bool mlp_layerwise_fin_ok(const MlpArgs* a, int n, int width);
int mlp_layerwise_fin_fit(const MlpArgs* a, int n, int host_wgs, int n_cus);
hipError_t launch_slice_tp_with_fin(const MlpArgs& host, const MlpArgs* a, int n, int n_ride, int width, int n_cus, hipStream_t st,
                                    int prec);
hipError_t launch_mlp_layerwise_first(const MlpArgs* t, int t_n, int t0, int width, int n_cus, hipStream_t st, int prec);
bool mlp_layerwise_rider_ok(const MlpArgs* a, int n, const MlpArgs& rider, int n_cus);
hipError_t init_layerwise_attrs();
hipError_t launch_mlp_slice_tp(const MlpArgs& a, hipStream_t st);
hipError_t launch_mlp_slice_tp2(const MlpArgs& a0, const MlpArgs& a1, int n_cus, hipStream_t st);
hipError_t init_slice_tp_attrs();
bool mlp_slice_tp_shape_ok(const MlpArgs& a, int width);
hipError_t init_kernel_attrs();
hipError_t launch_dw_adam(const DwArgs& a, hipStream_t st);
hipError_t launch_repack(const RepackItem* items_dev, int n_items, int total_blocks, hipStream_t st);
hipError_t launch_adam_flat(float* th, float* m, float* v, float* tt, const float* g, long n,
                            const AdamScalars& ad, hipStream_t st);
hipError_t launch_polyak_flat(float* tt, const float* th, long n, double tau, hipStream_t st);
hipError_t launch_alpha_step(double* log_alpha, double* m, double* v, const float* logp, int B,
                             float target_entropy, double lr, double beta1, double beta2, double eps,
                             int step, double* grad_out, const double* grad_in, float grad_scale,
                             hipStream_t st);
hipError_t launch_reduce_partials(const float* partials, int n_slices, float* out, int out_off,
                                  float scale_loss, float scale_mean, hipStream_t st);
hipError_t launch_sum(const float* x, int n, float* out, int out_off, float scale, hipStream_t st);
hipError_t launch_tqc_target(const float* z, long net_stride, int ldz, int n_nets, int Q, int drop,
                             const float* r, const float* d, const float* logp,
                             const double* log_alpha, float gamma, int B, float* target,
                             hipStream_t st);
int replay_dims(const oprl_replay* h, int* S, int* A);
int replay_view(const oprl_replay* h, const float** states, const float** actions,
                const float** rewards, const float** dones, const int** ends, int* n_eps, int* L,
                long* n_transitions);
hipError_t launch_debug_normal(unsigned long long seed, unsigned long long ctr, int rows, int cols, float* out,
                               hipStream_t st);
hipError_t init_fused_attrs();
size_t fused_xbuf_granules_per_cluster(int nc);
hipError_t launch_ddpg_phase1(const DdpgArgs& a, hipStream_t st);
hipError_t launch_ddpg_phase1_dw(const DdpgArgs& a, const DwKArgs& d, hipStream_t st);
bool fused_ddpg_is_lean(const DdpgArgs& a);
hipError_t launch_ddpg_phase2(const DdpgArgs& a, hipStream_t st);
hipError_t launch_ddpg_phase2_dw(const DdpgArgs& a, const DwKArgs& d, hipStream_t st);
hipError_t launch_ddpg_chain(const DdpgArgs& a, const DwKArgs4& dc, const DwKArgs4& da, const ChainArgs& c, hipStream_t st);
hipError_t launch_ddpg_phase1_group(const DdpgArgs& a0, const DdpgArgs* batch_dev, int n, hipStream_t st);
hipError_t launch_ddpg_phase2_group(const DdpgArgs& a0, const DdpgArgs* batch_dev, int n, hipStream_t st);
int fill_dw_kargs(const DwArgs& a, DwKArgs* k, int tile_k = 32);
bool fused_x2_tiles();
hipError_t launch_dw_adam_group(const void* batch_dev, int ni, int n, int tiles, hipStream_t st);
int compact_dw_kargs(const DwKArgs& k, void* o, int ni);
size_t dw_group_block_bytes(int ni);

}  // namespace oprl

using namespace oprl;

// ---- minimal run-time binding of RCCL (NCCL API; enum values are the API's) ----
namespace {
struct NcclId { char internal[OPRL_COMM_ID_BYTES]; };
typedef int (*fn_get_unique_id)(NcclId*);
typedef int (*fn_comm_init_rank)(void**, int, NcclId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_comm_destroy)(void*);
typedef const char* (*fn_get_error_string)(int);
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_broadcast broadcast = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_get_error_string err_str = nullptr;
  void* comm = nullptr;
  int rank = 0, world = 1;
};

int rccl_bind(Rccl& r, const char* path) {
  if (r.lib) return OPRL_OK;
  r.lib = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) { set_err("dlopen(%s) failed: %s", path ? path : "librccl.so", dlerror()); return OPRL_ERR_INVALID; }
  r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
  r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
  r.broadcast = (fn_broadcast)dlsym(r.lib, "ncclBroadcast");
  r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
  r.err_str = (fn_get_error_string)dlsym(r.lib, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce) {
    set_err("%s does not export the NCCL API", path ? path : "librccl.so");
    return OPRL_ERR_INVALID;
  }
  return OPRL_OK;
}
}  // namespace

#define HIPC(x)                                                                        \
  do {                                                                                 \
    hipError_t _e = (x);                                                               \
    if (_e != hipSuccess) {                                                            \
      set_err("%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
      return OPRL_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)
#define RC(x)                      \
  do {                             \
    int _rc = (x);                 \
    if (_rc != OPRL_OK) return _rc; \
  } while (0)

namespace {

long net_param_count(const oprl_net& n) {
  long c = 0;
  for (int l = 0; l < n.n_layers; ++l) c += (long)n.dims[l + 1] * n.dims[l] + n.dims[l + 1];
  return c;
}

long w_off(const oprl_net& n, int l) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += (long)n.dims[j + 1] * n.dims[j] + n.dims[j + 1];
  return c;
}
long b_off(const oprl_net& n, int l) { return w_off(n, l) + (long)n.dims[l + 1] * n.dims[l]; }

int check_net(const oprl_net& n, const char* name, int* width) {
  if (n.n_layers < 2 || n.n_layers > OPRL_MAX_LAYERS) {
    set_err("%s: n_layers=%d unsupported (2..%d)", name, n.n_layers, OPRL_MAX_LAYERS);
    return OPRL_ERR_INVALID;
  }
  const int w = n.dims[1];
  if (w != 256 && w != 512) { set_err("%s: hidden width %d unsupported (256 or 512)", name, w); return OPRL_ERR_INVALID; }
  for (int l = 1; l < n.n_layers; ++l)
    if (n.dims[l] != w) { set_err("%s: hidden widths must be equal", name); return OPRL_ERR_INVALID; }
  if (n.dims[0] < 1 || n.dims[0] > 96) { set_err("%s: input dim %d unsupported (1..96)", name, n.dims[0]); return OPRL_ERR_INVALID; }
  if (n.dims[n.n_layers] < 1 || n.dims[n.n_layers] > kNarrowMax) {
    set_err("%s: output dim %d unsupported (1..%d)", name, n.dims[n.n_layers], kNarrowMax);
    return OPRL_ERR_INVALID;
  }
  if (!n.theta) { set_err("%s: theta is null", name); return OPRL_ERR_INVALID; }
  if (!n.pack) { set_err("%s: pack buffer is null (see oprl_net_pack_floats)", name); return OPRL_ERR_INVALID; }
  if (n.theta_target && !n.pack_target) { set_err("%s: pack_target is null", name); return OPRL_ERR_INVALID; }
  *width = w;
  return OPRL_OK;
}

// offsets (floats) of layer l's forward / backward pack inside a pack buffer
long pack_off_fwd(const oprl_net& n, int l) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += 2 * pack_floats(n.dims[j + 1], n.dims[j]);
  return c;
}
long pack_off_bwd(const oprl_net& n, int l) { return pack_off_fwd(n, l) + pack_floats(n.dims[l + 1], n.dims[l]); }
long net_pack_floats(const oprl_net& n) { return pack_off_fwd(n, n.n_layers); }

// the same for the bf16 packs (library-owned, oprl_learner::pack16 / pack16_t), in floats (16-byte fragments)
// (pl = fp16 / bf16 planes per block: 1 for the bf16 packs, 2 for the PrecX2 packs — hi | lo)
long pack16_off_fwd(const oprl_net& n, int l, int pl = 1) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += pl * (pack16_floats(n.dims[j + 1], n.dims[j]) + pack16_floats(n.dims[j], n.dims[j + 1]));
  return c;
}
long pack16_off_bwd(const oprl_net& n, int l, int pl = 1) { return pack16_off_fwd(n, l, pl) + pl * pack16_floats(n.dims[l + 1], n.dims[l]); }
long net_pack16_floats(const oprl_net& n, int pl = 1) { return pack16_off_fwd(n, n.n_layers, pl); }

Net net_view(const oprl_net& n, bool target) {
  Net v;
  memset(&v, 0, sizeof v);
  v.n_layers = n.n_layers;
  for (int l = 0; l <= n.n_layers; ++l) v.dims[l] = n.dims[l];
  const float* base = target ? n.theta_target : n.theta;
  const float* pk = target ? n.pack_target : n.pack;
  for (int l = 0; l < n.n_layers; ++l) {
    v.b[l] = base + b_off(n, l);
    v.pf[l] = pk + pack_off_fwd(n, l);
    v.pb[l] = pk + pack_off_bwd(n, l);
  }
  return v;
}

// per-net activation / gradient exchange buffers (HBM, sized for max_batch rows)
constexpr int kMaxCluster = 4;   // CUs per tensor-parallel slice cluster (csrc/tp3.h)

// a Net whose pf / pb point at the bf16 packs (for the PrecBF16 kernels only)
Net net_view16(const oprl_net& n, bool target, const float* pk16, int pl = 1) {
  Net v = net_view(n, target);
  for (int l = 0; l < n.n_layers; ++l) {
    v.pf[l] = pk16 + pack16_off_fwd(n, l, pl);
    v.pb[l] = pk16 + pack16_off_bwd(n, l, pl);
  }
  return v;
}

struct NetWs {
  float* X[kMaxLayers] = {nullptr, nullptr, nullptr, nullptr};
  float* dY[kMaxLayers] = {nullptr, nullptr, nullptr, nullptr};
  int ldx0 = 0, lddo = 0, width = 0;
  long dY0_stride = 0;   // dY[0] is kMaxCluster buffers this many floats apart (dz1 partials)
};

struct Pool {  // one hipMalloc, bump allocated
  char* base = nullptr;
  size_t cap = 0, used = 0;
  template <class T>
  T* take(size_t n) {
    used = (used + 255) & ~(size_t)255;
    T* p = reinterpret_cast<T*>(base + used);
    used += n * sizeof(T);
    return p;
  }
};

}  // namespace

struct oprl_learner {
  oprl_learner_config cfg;
  int S, A, Bmax, nc;
  int w_actor = 0, w_critic = 0;
  Pool pool;
  NetWs ws_actor, ws_critic[OPRL_MAX_CRITICS];
  std::vector<DwItem> items_host;  // [critic items..., actor items...] (travel in the kernel arguments)
  int n_items_critic = 0, n_items_actor = 0, tiles_critic = 0, tiles_actor = 0;
  // batch-sized scratch
  float *a2 = nullptr, *logp2 = nullptr, *qn = nullptr /*[nc][B][ldq]*/, *pi = nullptr,
        *raw = nullptr, *logp = nullptr, *da = nullptr /*[nc][B][A]*/, *qpi = nullptr /*[nc][B]*/,
        *target = nullptr, *ydbg = nullptr, *qdbg = nullptr;
  int ldq = 0;
  float *part_c = nullptr /*[nc][slices][4]*/, *part_a = nullptr, *scalars = nullptr;
  double* alpha_grad = nullptr;
  // step_n batch buffers
  float *bs = nullptr, *ba = nullptr, *br = nullptr, *bd = nullptr, *bs2 = nullptr;
  int64_t update_count = 0;
  int opt_step_critic = 0, opt_step_actor = 0, opt_step_alpha = 0;
  int last_B = 0;
  bool actor_updated_last = false;
  long long* trace = nullptr;
  int trace_slot = 0;
  Rccl rccl;
  long n_critic_params = 0, n_actor_params = 0;
  // side streams: independent per-net launches (twin / quantile critics) run concurrently
  hipStream_t side[OPRL_MAX_CRITICS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[OPRL_MAX_CRITICS] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool have_side = false;
  bool fused = false;          // DDPG / TD3 / SAC two-kernel path (csrc/fused_ddpg.hip) is built for this learner
  const float* noise1_pending = nullptr;   // update()'s injected actor-phase draws: SAC's role C runs in phase 1
  bool tp_generic_on = false;  // the generic per-net launches may run on clusters of 4 (csrc/slice_tp.hip)
  unsigned tp_tag = 0;         // launch-unique tag source of the cluster exchanges (fused and generic)
  // for_each_net over two nets: their cluster launches are collected and go out as one (k_mlp_slice_tp2)
  bool pair_collect = false;
  int pair_n = 0;
  MlpArgs pair_args[2];
  P2pState p2p;                // one-shot all-reduce windows (csrc/p2p.hip); used when p2p_ok
  bool p2p_ok = false, p2p_tested = false, p2p_inline = false;
  int p2p_max_tiles = 0;
  bool dp_inline = false;      // this data-parallel update exchanges inside the dW launches (k_dw_adam<true>)
  DwXchg dw_xchg;
  bool no_dp_inline = false;   // OPRL_AMD_NO_DP_INLINE: peer-window exchanges as separate launches (tests / A-B)
  bool no_twin_split = false;  // OPRL_AMD_NO_SIDE_BY_SIDE: role A runs both target critics back to back (tests / A-B)
  bool no_multi = false;
  PrefetchJob prefetch;        // step_n on the generic path (TQC): the next update's rows as riders of this update's k_lw_dact launch
  bool prefetch_pending = false, prefetch_done = false;
  bool no_gather_ride = false; // OPRL_AMD_NO_RIDE bit 8: a k_replay_gather launch per update (tests / A-B)
  float* batch_alt = nullptr;  // the second set of batch rows [Bmax x (2 S + A + 2)] the riders fill while an update reads the first
  MlpArgs fin_args[OPRL_MAX_CRITICS];   // TQC: the online critics' first-launch arguments of this update (critic_phase step 1) ...
  int fin_tail0 = -1;          // ... of which [fin_tail0, nc) did not fit beside the actor's forward: offered to the target pass's head launch (-1: none pending)
  bool fin16 = false;
  bool fin_done = false;       // TQC: the online critics' first hidden launch rode on the actor's forward on s' (critic_phase step 1); step 3 skips it
  bool no_fin_ride = false;    // OPRL_AMD_NO_RIDE bit 4: it stays the first launch of step 3 (tests / A-B)
  LwPairBuf lw_pairs = {nullptr, 0, 1u, 1 << 20, nullptr, 3, 0};   // k_lw_mid_pair: flags (own allocation), tags; OPRL_AMD_LW_PAIR: bit 0 forward, bit 1 backward pairs (default 3)
  float* lw_scratch = nullptr; // [critics][layers 1 .. L-1][Bmax x 512]: activations of forward-only layer-by-layer launches (the target pass) — not the nets' dW exchange buffers, which the early first launch has already filled
  MlpArgs rider;               // TQC: the actor's forward on s, prepared in critic_phase to ride on the critic step's head launch ...
  bool rider_pending = false;  // ... offered to the next for_each_net; taken: rider_done, and actor_phase skips its step 5
  bool rider_done = false;
  bool no_af_ride = false;     // OPRL_AMD_NO_RIDE bit 2: the forward stays a launch of actor_phase (tests / A-B)
  TqcJob tqc_job;              // TQC: the TD target as the tail of the target critics' head launch (kernels.h) ...
  bool tqc_job_pending = false; // ... offered to the next for_each_net; still set afterwards: k_tqc_target as a launch of its own
  bool no_tqc_ride = false;    // OPRL_AMD_NO_RIDE bit 1: always that launch (tests / A-B)
  unsigned long long* tqc_counter = nullptr;   // [slices at Bmax] arrival counters, zeroed once
  bool no_layerwise = false;   // OPRL_AMD_NO_LAYERWISE: wide nets stay on the single-CU slice kernel (tests / A-B)
  bool no_p2_pair = false;     // OPRL_AMD_NO_SIDE_BY_SIDE: SAC phase 2 runs the twin critics back to back (tests / A-B)
  bool multi_collect = false;  // for_each_net over > 2 single-CU nets: one k_mlp_slice_multi launch
  int multi_n = 0, multi_width = 0;
  MlpArgs multi_args[kMaxMulti];
  bool staged_ready = false;   // step_n: the staging batch holds the next update's rows (written by phase 2)
  unsigned long long* y_granules = nullptr;   // [Bmax] TD-target hand-off (fused DDPG)
  unsigned epoch = 0;          // monotonically increasing, never reset
  int ncl = 1;                 // CUs per slice cluster in the fused path (csrc/tp3.h)
  int n_cus = 256;
  int no_lean = 0;
  bool shared_chip = false;    // oprl_learner_set_cluster(< 8): this learner is one of several on the GPU
  int no_merge = 0;            // OPRL_AMD_FORM=plain: dW launches of their own
  int no_merge2 = 0;           // OPRL_AMD_FORM=p2 / plain: phase 2 runs the actor's backward itself, the actor's dW is a launch of its own
  // merged phase 2 (DdpgArgs::merged bit 1): du granules [Bm][kDuLd], the first layer's dz1 granules [16][Bm][16] and the
  // snapshot of the actor's output layer (Bm = min(max_batch, 256))
  // oprl_learner_step_act: host-mapped pinned block [obs 512 floats | out 512 granules {ticket, value}] and the ticket of the pending row
  float* act_pin = nullptr;
  float* act_map = nullptr;
  unsigned act_ticket = 0;
  bool act_pending = false;
  unsigned long long* du_granules = nullptr;
  unsigned long long* g1_granules = nullptr;
  float* w3_snap = nullptr;
  int no_wide = 0;             // OPRL_AMD_NO_WIDE: never run role A / phase 2's critic pass on clusters of eight
  int xnc = kMaxCluster;       // members an exchange area of xbuf is laid out for
  unsigned long long* xbuf = nullptr;
  size_t xbuf_granules = 0;
  // Largest cluster size for which ONE role's clusters (c x slices workgroups, one per CU) fit on the
  // chip.  Phase 1's grid may then exceed the CU count (B > 256): workgroups are dispatched in block
  // order — role A's clusters, then B's, then C's — role A waits for nobody, the members of a cluster
  // are dispatched together, and a B workgroup only ever waits for an A workgroup dispatched before it,
  // so later roles simply start as earlier workgroups retire.
  // Measured (profiles/r01g_batch_sweep.txt): worth it for the lean clusters of 4 (B = 512: 76 -> 51 us
  // per DDPG update); the generic passes on smaller clusters do better fully co-resident.
  int nc_cluster(int B) const {
    const int slices = (B + kR - 1) / kR;
    if (ncl == 4 && 4 * slices <= n_cus) return 4;
    const int roles = 2 + nc;
    int c = ncl;
    while (c > 1 && roles * c * slices > n_cus) c >>= 1;
    return c;
  }
  BatchSrc src;                // where the current update's minibatch comes from
  BatchSrc next_src;           // step_n: what phase 2 should gather for the next update
  int prefetch_next = 0;
  bool prefetch_p1 = false;    // step_n: phase 1 carries the next update's rows (two staging sets), not phase 2
  // key of the in-update noise streams (TD3 smoothing, SAC / TQC reparameterisation draws): the run
  // seed and, in a data-parallel job, the rank — every seed and every rank draws its own eps
  uint64_t noise_seed = 0;
  int noise_rank = 0;
  // OPRL_PREC_BF16: bf16 fragment packs of every net (online: forward + backward, target: forward),
  // derived state owned by the library and written by the dW + Adam epilogues / k_repack; index 0 = actor,
  // 1 + j = critic j
  // Bounded cross-workgroup waits (cluster all-reduce, TD-target hand-off, twin exchanges, gradient tile /
  // window exchanges) REPORT an expiry here besides poisoning their result with NaN: one word of
  // host-mapped memory, written by the device only on that error path (tp3.h report_expired), read by
  // the host at the start of every update / step_n / apply / read_scalars call — no copy, no sync.
  unsigned* err_host = nullptr;
  unsigned* err_dev = nullptr;
  int debug_expire = 0;        // test hook (oprl_learner_debug_expire): this wait site gives up at once
  bool bf16 = false;
  bool x2 = false;             // OPRL_PREC_X2: the lean fused kernels run PrecX2 (engine.h) from packs of two fp16 planes, kept in pack16 / pack16_t
  int planes = 1;              // fp16 / bf16 planes per block of those packs
  // PrecX2 learners: the fused updates do not write the fp32 packs (nothing of theirs reads them); whoever does —
  // the nets' own forward (oprl_mlp_forward / act / backward), a generic launch sequence — gets them rebuilt from the
  // master first (fresh32): [0] the critics' (online + target), [1] the actor's
  bool stale32[2] = {false, false};
  bool lazy_wide = false;      // this learner is in g_lazy and its wide layers' fp32 packs may be left stale (16-bit TQC)
  bool stale_wide = false;     // ... and are: only the critics' 512 x 512 layers' fp32 packs (the narrow layers' are current)
  float* uc_base = nullptr;    // the fp16 packs' uncached allocation (PrecX2 learners)
  bool uc_pool = false;        // the workspace pool is uncached memory as well
  // exact-fp32 DDPG learners (fchain): the fused kernels' fp32 fragment packs are library-owned UNCACHED mirrors of the
  // caller's pack arenas (same layout) — what k_ddpg_chain<PrecF32>'s tiles write, the next update's roles read without a
  // kernel boundary; the caller's packs are rebuilt from the masters when something outside reads them (fresh32).
  // fnet[0] = the actor, fnet[1] = the critic with pack / pack_target -> the mirrors (uc_base holds them)
  bool fchain = false;
  oprl_net fnet[2];
  // k_ddpg_chain (the whole update, several per launch): role C's / the critic tiles' flags, the critic's uncached bias copies
  unsigned long long* w_flags = nullptr;
  float* critic_b16 = nullptr;
  // k_ddpg_chain (several updates per launch): the tiles' FIN flags, the prefetch flags, the uncached bias copies of all
  // four nets ([0] actor, [1] actor target, [2] critic = critic_b16, [3] critic target) and the output layer's two buffers
  unsigned long long* chain_flags = nullptr;   // [ct_fin 192 | at_fin 192 | pf_done 64 | gu_flags 128 | partial q 1024]
  float* gu = nullptr;                         // [kDuLd][Bm][256] the actor's unit-seed dz1 rows (DwGate kind 3)
  float* chain_b16 = nullptr;                  // [4][kMaxLayers][256]
  float* w3buf1 = nullptr;                     // (w3buf[0] = w3_snap)
  int chain_u = 1;             // step_n: updates the next whole-update launch runs (k_ddpg_chain)
  bool chain_pf_last = false;  // ... and whether its last update stages the rows of the update after it
  const float* chain_set1[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // the other staging set (set 0 = the update's rows)
  int no_chain = 0;            // (always 0: every whole update goes through k_ddpg_chain)
  int chain_max = kChainMax;   // OPRL_AMD_CHAIN=n: at most n updates per launch
  int no_whole = 0;            // OPRL_AMD_FORM=two / p2 / plain: two launches per update (merged phase 1, merged phase 2)
  bool whole_done = false;     // this update's actor phase was part of the critic phase's launch
  float* pack16[OPRL_MAX_CRITICS + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float* pack16_t[OPRL_MAX_CRITICS + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // prebuilt device repack tables: [0] critics online, [1] critics online+target, [2] actor (+target)
  RepackItem* rp_dev[3] = {nullptr, nullptr, nullptr};
  int rp_n[3] = {0, 0, 0}, rp_blocks[3] = {0, 0, 0};
};

namespace {

// learners with lazily maintained fp32 packs, by pack pointer (oprl_mlp_* know a net, not its learner)
// Whole-update launches (k_ddpg_chain) take the whole chip for up to 32 updates.  Learners of one process that launch them
// from different streams (one host thread per learner: the multi-seed layout) take TURNS: a launch waits for the event
// behind the last whole-update launch of another stream — two such launches side by side would only hold each other's
// compute units with waiting workgroups (bounded waits would expire).  Nothing is recorded while the process has one
// learner (the headline path: no event, no barrier packet).
struct ChipTurn {
  std::mutex mu;
  hipEvent_t ev[16] = {};
  hipStream_t stream[16] = {};
  bool rec[16] = {};
};
ChipTurn g_turn;
std::atomic<int> g_live{0};

hipError_t chip_turn_begin(hipStream_t st, int* dev_out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  *dev_out = dev & 15;
  if (g_turn.rec[*dev_out] && g_turn.stream[*dev_out] != st) return hipStreamWaitEvent(st, g_turn.ev[*dev_out], 0);
  return hipSuccess;
}
void chip_turn_end(hipStream_t st, int dev) {
  if (g_live.load() <= 1) { g_turn.rec[dev] = false; return; }
  if (g_turn.ev[dev] == nullptr && hipEventCreateWithFlags(&g_turn.ev[dev], hipEventDisableTiming) != hipSuccess) { g_turn.ev[dev] = nullptr; return; }
  if (hipEventRecord(g_turn.ev[dev], st) == hipSuccess) { g_turn.rec[dev] = true; g_turn.stream[dev] = st; }
}

// the net the fused kernels of `h` see (fchain: the one with the mirrored packs)
const oprl_net& eff(const oprl_learner* h, const oprl_net& n) {
  if (h->fchain) {
    if (&n == &h->cfg.actor) return h->fnet[0];
    if (&n == &h->cfg.critics[0]) return h->fnet[1];
  }
  return n;
}

std::mutex g_lazy_mu;
std::vector<oprl_learner*> g_lazy;

// Uncached device memory (hipDeviceMallocUncached) is never handed back to the runtime: a block a destroyed learner
// owned waits here for the next PrecX2 learner.  Measured (tools/sac_probe2.py, r03 log): after hipFree of such a block,
// later ordinary allocations of the same process — another learner's workspace — lost flag granules in the fused
// kernels (bounded waits expired) until the process ended; with the blocks kept, 0 failures in the same churn.
struct UcBlock { void* p; size_t bytes; bool used; };
std::mutex g_uc_mu;
std::vector<UcBlock> g_uc;
hipError_t uc_alloc(void** out, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_uc_mu);
  for (UcBlock& b : g_uc)
    if (!b.used && b.bytes >= bytes && b.bytes <= 2 * bytes + (1u << 20)) { b.used = true; *out = b.p; return hipSuccess; }
  // a process that cycles learners of ever different shapes must not grow without bound: once more than 64 MB of
  // blocks lie idle, ANY idle block that is large enough serves (the smallest such), whatever its size
  size_t idle = 0;
  for (const UcBlock& b : g_uc) if (!b.used) idle += b.bytes;
  if (idle > ((size_t)64 << 20)) {
    UcBlock* best = nullptr;
    for (UcBlock& b : g_uc)
      if (!b.used && b.bytes >= bytes && (best == nullptr || b.bytes < best->bytes)) best = &b;
    if (best != nullptr) { best->used = true; *out = best->p; return hipSuccess; }
  }
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) return e;
  g_uc.push_back(UcBlock{p, bytes, true});
  *out = p;
  return hipSuccess;
}
void dev_free(void* p);
bool uc_release(void* p) {       // true: the block was one of these (and is NOT freed)
  std::lock_guard<std::mutex> lk(g_uc_mu);
  for (UcBlock& b : g_uc)
    if (b.p == p) { b.used = false; return true; }
  return false;
}
void dev_free(void* p) {
  if (p != nullptr && !uc_release(p)) (void)hipFree(p);
}

// (wide_too: also when only the wide layers' packs are stale — the launch about to run reads THOSE)
int fresh32_tables(oprl_learner* h, int which /* bit 0 critics, bit 1 actor */, hipStream_t st, bool wide_too = false) {
  if ((which & 1) && (h->stale32[0] || (wide_too && h->stale_wide))) {
    HIPC(launch_repack(h->rp_dev[1], h->rp_n[1], h->rp_blocks[1], st));
    h->stale32[0] = false;
    h->stale_wide = false;
  }
  if ((which & 2) && h->stale32[1]) {
    HIPC(launch_repack(h->rp_dev[2], h->rp_n[2], h->rp_blocks[2], st));
    h->stale32[1] = false;
  }
  return OPRL_OK;
}

// before a launch that reads `net`'s fp32 packs outside its learner's fused kernels
int fresh32(const oprl_net* net, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_lazy_mu);
  for (oprl_learner* h : g_lazy) {
    if (!h->stale32[0] && !h->stale32[1] && !h->stale_wide) continue;
    // (a target module of the Python host is a net of its own whose pack IS the learner's target pack)
    auto same = [&](const oprl_net& n) {
      return net->pack == n.pack || (n.pack_target != nullptr && (net->pack == n.pack_target || net->pack_target == n.pack_target));
    };
    if (same(h->cfg.actor)) return fresh32_tables(h, 2, st);
    for (int j = 0; j < h->nc; ++j)
      if (same(h->cfg.critics[j])) return fresh32_tables(h, 1, st, true);
  }
  return OPRL_OK;
}

void fill_items(const oprl_net& n, const NetWs& ws, std::vector<DwItem>& v, int* tiles, bool small_partial_tiles = false,
                float* pk16 = nullptr, float* pk16_t = nullptr, int pl = 1) {
  for (int l = 0; l < n.n_layers; ++l) {
    DwItem it;
    memset(&it, 0, sizeof it);
    it.K = n.dims[l];
    it.N = n.dims[l + 1];
    it.X = ws.X[l];
    it.ldx = (l == 0) ? ws.ldx0 : ws.width;
    it.dY = ws.dY[l];
    it.ldy = (l == n.n_layers - 1) ? ws.lddo : ws.width;
    const long wo = w_off(n, l), bo = b_off(n, l);
    it.w = n.theta + wo;                       it.b = n.theta + bo;
    it.w_t = n.theta_target ? n.theta_target + wo : nullptr;
    it.b_t = n.theta_target ? n.theta_target + bo : nullptr;
    it.w_m = n.adam_m ? n.adam_m + wo : nullptr; it.b_m = n.adam_m ? n.adam_m + bo : nullptr;
    it.w_v = n.adam_v ? n.adam_v + wo : nullptr; it.b_v = n.adam_v ? n.adam_v + bo : nullptr;
    it.w_g = n.grad ? n.grad + wo : nullptr;     it.b_g = n.grad ? n.grad + bo : nullptr;
    it.pf = n.pack + pack_off_fwd(n, l);
    it.pb = n.pack + pack_off_bwd(n, l);
    it.tpf = n.pack_target ? n.pack_target + pack_off_fwd(n, l) : nullptr;
    it.pf16 = pk16 ? pk16 + pack16_off_fwd(n, l, pl) : nullptr;
    it.pb16 = pk16 ? pk16 + pack16_off_bwd(n, l, pl) : nullptr;
    it.tpf16 = (pk16_t && n.theta_target) ? pk16_t + pack16_off_fwd(n, l, pl) : nullptr;
    it.x2 = pl == 2 ? 1 : 0;
    it.dY_part_stride = (l == 0 && n.n_layers > 1) ? ws.dY0_stride : 0;
    it.scaled = (l < n.n_layers - 1) ? 1 : 0;
    it.rs = ws.dY[n.n_layers - 1];
    it.rs_ld = ws.lddo;
    // (8-row tiles for this layer were measured: no difference — profiles/r01b_experiments.txt)
    it.tile_n = kDwTileN;
    const int tn = (it.N + it.tile_n - 1) / it.tile_n;
    it.tiles_k = (it.K + kDwTile - 1) / kDwTile;
    it.tile_begin = *tiles;
    *tiles += tn * it.tiles_k;
    it.tile_end = *tiles;
    v.push_back(it);
  }
}

size_t net_ws_floats(const oprl_net& n, int B) {
  size_t f = 0;
  f += (size_t)B * round_up(n.dims[0], 4) + 64;
  for (int l = 1; l < n.n_layers; ++l) f += (size_t)B * n.dims[1] + 64;
  for (int l = 0; l < n.n_layers - 1; ++l) f += ((size_t)B * n.dims[1] + 1088) * (l == 0 ? kMaxCluster : 1) + 64;
  f += (size_t)B * round_up(n.dims[n.n_layers], 4) + 64;
  return f + 64 * 8;
}

void alloc_net_ws(Pool& p, const oprl_net& n, int B, NetWs* ws) {
  ws->width = n.dims[1];
  ws->ldx0 = round_up(n.dims[0], 4);
  ws->lddo = round_up(n.dims[n.n_layers], 4);
  ws->X[0] = p.take<float>((size_t)B * ws->ldx0);
  for (int l = 1; l < n.n_layers; ++l) ws->X[l] = p.take<float>((size_t)B * ws->width);
  // the dz1 partial buffers are read together (k_dw_adam sums them on load): an odd multiple
  // of 4 KB + 256 B between them keeps the four loads of one element off the same HBM channel
  ws->dY0_stride = (long)B * ws->width + 1088;
  for (int l = 0; l < n.n_layers - 1; ++l)
    ws->dY[l] = p.take<float>(l == 0 ? (size_t)ws->dY0_stride * kMaxCluster : (size_t)B * ws->width);
  ws->dY[n.n_layers - 1] = p.take<float>((size_t)B * ws->lddo);
}

void set_adam(AdamScalars& ad, double lr, double beta1, double beta2, double eps, double tau) {
  ad.lr = (float)lr; ad.beta1 = (float)beta1; ad.beta2 = (float)beta2; ad.eps = (float)eps;
  ad.tau = (float)tau;
  ad.omb1 = (float)(1.0 - beta1);
  ad.omb2 = (float)(1.0 - beta2);
  ad.omtau = (float)(1.0 - tau);
  ad.lr_d = lr; ad.beta1_d = beta1; ad.beta2_d = beta2;
}

void set_step(AdamScalars& ad, int step) {
  ad.step_base = step; ad.step_dev = nullptr;
  ad.step_size_host = (float)(ad.lr_d / (1.0 - pow(ad.beta1_d, (double)step)));
  ad.bc2_sqrt_host = (float)sqrt(1.0 - pow(ad.beta2_d, (double)step));
}

AdamScalars adam_scalars(const oprl_learner* h, double lr, int step, bool polyak, float grad_scale) {
  AdamScalars ad;
  memset(&ad, 0, sizeof ad);
  const oprl_hparams& hp = h->cfg.hp;
  set_adam(ad, lr, hp.beta1, hp.beta2, hp.adam_eps, hp.tau);
  set_step(ad, step);
  ad.do_polyak = polyak ? 1 : 0;
  ad.do_adam = h->cfg.export_grads ? 0 : 1;
  ad.grad_scale = grad_scale;
  return ad;
}

// Does a slice launch of net `n` at batch B run on tensor-parallel clusters (slice_tp.hip)?  One
// predicate for the launch and for the dW kernel that has to sum the dz1 partials it leaves.
bool tp_generic(const oprl_learner* h, const oprl_net& n, int B) {
  if (!h->tp_generic_on || h->xbuf == nullptr) return false;
  if (n.n_layers != 3 || n.dims[1] != 256 || n.dims[2] != 256) return false;
  if (n.dims[0] > 96 || n.dims[3] > kNarrowMax) return false;
  if (((h->S & 15) + h->A - 1) / 16 >= 4) return false;          // input-gradient column span
  const int slices = (B + kR - 1) / kR;
  return slices * 4 <= h->n_cus;
}

MlpArgs base_args(oprl_learner* h, const oprl_net& n, bool target, int B) {
  MlpArgs a;
  memset(&a, 0, sizeof a);
  a.owner = h;
  if (tp_generic(h, n, B)) {
    a.tp_xbuf = h->xbuf;
    a.tp_tag_counter = &h->tp_tag;
    a.tp_xbuf_bytes = h->xbuf_granules * sizeof(unsigned long long);
  }
  if (h->trace != nullptr && h->trace_slot < OPRL_TRACE_SLOTS)
    a.trace = h->trace + (size_t)(h->trace_slot++) * 64 * kTraceStamps * 2;
  a.net = net_view(eff(h, n), target);
  a.err = h->err_dev;
  if (h->bf16 || h->x2) {     // (the 16-bit packs of the net's layers: bf16, or two fp16 planes per block)
    int idx = -1;                                      // 0 = actor, 1 + j = critic j
    if (&n == &h->cfg.actor) idx = 0;
    for (int j = 0; j < h->nc; ++j) if (&n == &h->cfg.critics[j]) idx = 1 + j;
    const float* pk16 = idx < 0 ? nullptr : (target ? h->pack16_t[idx] : h->pack16[idx]);
    for (int l = 0; pk16 != nullptr && l < n.n_layers; ++l) {
      a.pf16[l] = pk16 + pack16_off_fwd(n, l, h->planes);
      a.pb16[l] = target ? nullptr : pk16 + pack16_off_bwd(n, l, h->planes);
    }
  }
  a.B = B;
  a.action_dim = h->A;
  a.policy_noise = (float)h->cfg.hp.policy_noise;
  a.noise_clip = (float)h->cfg.hp.noise_clip;
  a.max_action = (float)h->cfg.hp.max_action;
  return a;
}

void with_store(MlpArgs& a, const NetWs& ws, bool x, bool dy) {
  for (int l = 0; l < kMaxLayers; ++l) {
    a.Xg[l] = x ? ws.X[l] : nullptr;
    a.dYg[l] = dy ? ws.dY[l] : nullptr;
  }
  a.ldx0 = ws.ldx0;
  a.lddo = ws.lddo;
  a.dY0_stride = ws.dY0_stride;
}

// launch-unique 26-bit tag for the cluster exchanges of one learner; on wrap-around every stale
// granule is retired
int next_tp_tag(unsigned* counter, unsigned long long* xbuf, size_t xbuf_bytes, hipStream_t st, unsigned* out) {
  *counter += 1;
  if ((*counter & 0x03FFFFFFu) == 0) {
    *counter += 1;
    HIPC(hipMemsetAsync(xbuf, 0, xbuf_bytes, st));
  }
  *out = *counter & 0x03FFFFFFu;
  return OPRL_OK;
}

int launch(const MlpArgs& a0, int width, hipStream_t st) {
  if (a0.tp_xbuf != nullptr && mlp_slice_tp_shape_ok(a0, width)) {
    MlpArgs a = a0;
    RC(next_tp_tag(a.tp_tag_counter, a.tp_xbuf, a.tp_xbuf_bytes, st, &a.tp_tag));
    oprl_learner* own = (oprl_learner*)a.owner;
    if (own != nullptr && own->pair_collect && own->pair_n < 2) {   // for_each_net over a pair: defer
      own->pair_args[own->pair_n++] = a;
      return OPRL_OK;
    }
    prof_begin(0, st);
    hipError_t e = launch_mlp_slice_tp(a, st);
    prof_end(st);
    HIPC(e);
    return OPRL_OK;
  }
  const MlpArgs& a = a0;
  {
    oprl_learner* own = (oprl_learner*)a.owner;
    if (own != nullptr && own->multi_collect && own->multi_n < kMaxMulti) {
      own->multi_width = width;
      own->multi_args[own->multi_n++] = a;
      return OPRL_OK;
    }
  }
  prof_begin(0, st);
  hipError_t e = launch_mlp_slice(a, width, st);
  prof_end(st);
  HIPC(e);
  return OPRL_OK;
}

hipError_t launch_dw_prof(const DwArgs& a, hipStream_t st) {
  prof_begin(1, st);
  hipError_t e = launch_dw_adam(a, st);
  prof_end(st);
  return e;
}

// Run launch_j(j, stream) for j in [0, n): net 0 on the caller's stream, the others on
// side streams forked from / joined back into it, so independent nets overlap on the GPU
// (each k_mlp_slice launch occupies only ceil(B/16) of the 256 CUs).
template <class F>
int for_each_net(oprl_learner* h, int n, hipStream_t st, F&& launch_j) {
  // measured: the event fork/join costs more than it saves for 2 nets (TD3 8.8k -> 7.7k/s),
  // pays for the 5 quantile critics of TQC (673 -> 1206/s)
  if (n == 2) {
    // twin nets on the same slices: their cluster launches (slice_tp.hip) go out as ONE launch
    h->pair_collect = true;
    h->pair_n = 0;
    int rc = launch_j(0, st);
    if (rc == OPRL_OK) rc = launch_j(1, st);
    h->pair_collect = false;
    RC(rc);
    if (h->pair_n == 2 && h->pair_args[0].B == h->pair_args[1].B) {
      prof_begin(0, st);
      hipError_t e = launch_mlp_slice_tp2(h->pair_args[0], h->pair_args[1], h->n_cus, st);
      prof_end(st);
      HIPC(e);
    } else {
      for (int k = 0; k < h->pair_n; ++k) {
        prof_begin(0, st);
        hipError_t e = launch_mlp_slice_tp(h->pair_args[k], st);
        prof_end(st);
        HIPC(e);
      }
    }
    h->pair_n = 0;
    return OPRL_OK;
  }
  if (n > 2 && n <= kMaxMulti && !h->no_multi) {
    // equal nets on the same slices (TQC's quantile critics): one launch, grid (slices, nets)
    h->multi_collect = true;
    h->multi_n = 0;
    int rc = OPRL_OK;
    for (int j = 0; j < n && rc == OPRL_OK; ++j) rc = launch_j(j, st);
    h->multi_collect = false;
    RC(rc);
    bool same = h->multi_n > 0;
    for (int k = 1; k < h->multi_n; ++k)
      same = same && h->multi_args[k].B == h->multi_args[0].B &&
             h->multi_args[k].net.n_layers == h->multi_args[0].net.n_layers;
    // wide nets go layer by layer over the whole chip (csrc/layerwise.hip); launches that keep no
    // activations (target nets, the actor phase's critics) borrow the nets' dW exchange buffers,
    // which nobody reads until the next storing launch overwrites them
    if (same && !h->no_layerwise && h->multi_width == 512 && h->multi_n <= h->nc) {
      for (int k = 0; k < h->multi_n; ++k) {
        MlpArgs& a = h->multi_args[k];
        const NetWs& ws = h->ws_critic[k];
        for (int l = 1; l < a.net.n_layers; ++l)
          if (a.Xg[l] == nullptr)
            a.Xg[l] = (!a.do_bwd && h->lw_scratch != nullptr)
                          ? h->lw_scratch + ((size_t)k * (kMaxLayers - 1) + (l - 1)) * (size_t)h->Bmax * 512
                          : ws.X[l];
        for (int l = 0; l + 1 < a.net.n_layers; ++l)
          if (a.dYg[l] == nullptr) a.dYg[l] = ws.dY[l];
      }
    }
    if (same && !h->no_layerwise && mlp_layerwise_ok(h->multi_args, h->multi_n, h->multi_width)) {
      // bf16 learners: the hidden layers (all but the first and the last) through their bf16 packs
      bool lw16 = h->bf16 || h->x2;
      for (int k = 0; k < h->multi_n; ++k) {
        const MlpArgs& a = h->multi_args[k];
        for (int l = 1; l + 1 < a.net.n_layers; ++l)
          lw16 = lw16 && a.pf16[l] != nullptr && (!a.do_bwd || a.pb16[l] != nullptr);
      }
      if (lw16)
        for (int k = 0; k < h->multi_n; ++k) {
          MlpArgs& a = h->multi_args[k];
          for (int l = 1; l + 1 < a.net.n_layers; ++l) { a.net.pf[l] = a.pf16[l]; if (a.pb16[l]) a.net.pb[l] = a.pb16[l]; }
        }
      prof_begin(0, st);
      // a pending TD-target job (critic_phase) rides on this launch's heads when it is the target critics' forward
      const TqcJob* job = nullptr;
      if (h->tqc_job_pending && !h->multi_args[0].do_bwd && h->multi_args[0].do_fwd && h->multi_n == h->tqc_job.n_nets &&
          h->multi_args[0].out == h->tqc_job.z) {
        job = &h->tqc_job;
        h->tqc_job_pending = false;
      }
      // a pending rider (critic_phase: the actor's forward on s) goes with the storing launch's heads
      const MlpArgs* rider = nullptr;
      if (h->rider_pending && h->multi_args[0].do_bwd && h->multi_args[0].do_fwd &&
          mlp_layerwise_rider_ok(h->multi_args, h->multi_n, h->rider, h->n_cus)) {
        rider = &h->rider;
        h->rider_pending = false;
        h->rider_done = true;
      }
      const bool first_done = h->fin_done && h->multi_args[0].do_bwd && h->multi_args[0].do_fwd && h->multi_args[0].Xg[0] != nullptr;
      if (first_done) h->fin_done = false;
      // the part of the online critics' early first launch that did not fit beside the actor's forward rides on the
      // target pass's heads (forward-only launch, 80 workgroups)
      const MlpArgs* tail = nullptr;
      int tail0 = 0;
      if (h->fin_tail0 >= 0 && !h->multi_args[0].do_bwd && h->multi_args[0].do_fwd) {
        const int slices = (h->multi_args[0].B + kR - 1) / kR;
        const int rest = h->nc - h->fin_tail0;
        if (mlp_layerwise_fin_fit(h->fin_args, h->nc, slices * h->multi_n, h->n_cus) >= rest) {   // all resident at once
          tail = h->fin_args; tail0 = h->fin_tail0;
          h->fin_tail0 = -1;
        }
      }
      // step_n: the next update's rows ride on the launch sequence that ends in k_lw_dact (the actor step's critics)
      const PrefetchJob* pf = nullptr;
      if (h->prefetch_pending && h->multi_args[0].do_bwd && h->multi_args[0].dact_cols > 0 && h->prefetch.B == h->multi_args[0].B) {
        pf = &h->prefetch;
        h->prefetch_pending = false;
        h->prefetch_done = true;
      }
      hipError_t e = launch_mlp_layerwise(h->multi_args, h->multi_n, h->multi_width, h->n_cus, st, lw16 ? (h->x2 ? 2 : 1) : 0, job, rider, first_done,
                                          tail, h->nc, tail0, h->fin16 ? (h->x2 ? 2 : 1) : 0, pf, &h->lw_pairs);
      // (a tag per pair launch; 2^32 launches on: every flag is retired before a tag can come round again)
      if (h->lw_pairs.next_tag + (unsigned)h->lw_pairs.used < h->lw_pairs.next_tag && h->lw_pairs.flags != nullptr)
        (void)hipMemsetAsync(h->lw_pairs.flags, 0, (size_t)h->lw_pairs.n_flags * sizeof(unsigned long long), st);
      h->lw_pairs.next_tag += (unsigned)h->lw_pairs.used;
      if (h->lw_pairs.next_tag == 0) h->lw_pairs.next_tag = 1;
      h->lw_pairs.used = 0;
      prof_end(st);
      HIPC(e);
    } else if (same) {
      // (these kernels read the fp32 packs: a 16-bit TQC learner's wide critics leave theirs stale)
      if (h->stale_wide && h->multi_width == h->w_critic) RC(fresh32_tables(h, 1, st, true));
      prof_begin(0, st);
      hipError_t e = launch_mlp_slice_multi(h->multi_args, h->multi_n, h->multi_width, st);
      prof_end(st);
      HIPC(e);
    } else {
      if (h->stale_wide && h->multi_width == h->w_critic) RC(fresh32_tables(h, 1, st, true));
      for (int k = 0; k < h->multi_n; ++k) {
        prof_begin(0, st);
        hipError_t e = launch_mlp_slice(h->multi_args[k], h->multi_width, st);
        prof_end(st);
        HIPC(e);
      }
    }
    h->multi_n = 0;
    return OPRL_OK;
  }
  if (n <= 2 || !h->have_side) {
    for (int j = 0; j < n; ++j) RC(launch_j(j, st));
    return OPRL_OK;
  }
  HIPC(hipEventRecord(h->ev_fork, st));
  for (int j = 1; j < n; ++j) HIPC(hipStreamWaitEvent(h->side[j], h->ev_fork, 0));
  for (int j = 0; j < n; ++j) {
    hipStream_t sj = j == 0 ? st : h->side[j];
    RC(launch_j(j, sj));
    if (j > 0) HIPC(hipEventRecord(h->ev_join[j], sj));
  }
  for (int j = 1; j < n; ++j) HIPC(hipStreamWaitEvent(st, h->ev_join[j], 0));
  return OPRL_OK;
}

const double* alpha_ptr(const oprl_learner* h) {
  const bool learned = h->cfg.algo == OPRL_TQC || (h->cfg.algo == OPRL_SAC && h->cfg.hp.tune_alpha);
  return learned ? h->cfg.log_alpha : nullptr;
}

// Philox key of noise stream `stream_id` (1: next-state draw / TD3 smoothing, 2: actor-step draw).
// Seed 0 on rank 0 is the bare stream constant; anything else is mixed in (splitmix64 finaliser).
unsigned long long noise_key(const oprl_learner* h, uint64_t stream_id) {
  uint64_t x = h->noise_seed ^ ((uint64_t)h->noise_rank * 0x9E3779B97F4A7C15ULL);
  if (x != 0) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
  }
  return 0x0b5e55edULL + stream_id + x;
}

void seed_rng(MlpArgs& a, const oprl_learner* h, const float* noise, uint64_t stream_id) {
  a.noise = noise;
  a.rng_seed = noise_key(h, stream_id);
  a.rng_ctr = (unsigned long long)h->update_count;
}

// ------------------------------------------------------------ fused DDPG / TD3
bool actor_due(const oprl_learner* h);

// grid rows of one update of k_ddpg_chain at batch B: 16 role rows + the tile-only rows of small batches (16 x 64 tiles)
int chain_rows(const oprl_learner* h, int B) {
  int tiles64[2] = {0, 0};
  for (size_t i = 0; i < h->items_host.size(); ++i)
    tiles64[(int)i < h->n_items_critic ? 0 : 1] += ((h->items_host[i].N + 15) / 16) * ((h->items_host[i].K + 63) / 64);
  const int sl = (B + kR - 1) / kR, mt = tiles64[0] > tiles64[1] ? tiles64[0] : tiles64[1];
  return 16 + (mt > 8 * sl ? (mt - 8 * sl + sl - 1) / sl : 0);
}

DdpgArgs ddpg_args(oprl_learner* h, int B) {
  const oprl_learner_config& c = h->cfg;
  DdpgArgs a;
  memset(&a, 0, sizeof a);
  a.actor = net_view(eff(h, c.actor), false);
  a.actor_t = net_view(eff(h, c.actor), true);
  a.critic = net_view(eff(h, c.critics[0]), false);
  a.critic_t = net_view(eff(h, c.critics[0]), true);
  a.n_critics = h->nc;
  a.do_actor = 1;
  if (h->nc == 2) {      // TD3 / SAC: twin critic
    a.critic2 = net_view(c.critics[1], false);
    a.critic2_t = net_view(c.critics[1], true);
    for (int l = 0; l < kMaxLayers; ++l) { a.c2X[l] = h->ws_critic[1].X[l]; a.c2dY[l] = h->ws_critic[1].dY[l]; }
    a.rng_seed = noise_key(h, 1);                         // the streams seed_rng() gives the generic path
    a.rng_ctr = (unsigned long long)h->update_count;
  }
  if (c.algo == OPRL_TD3) {   // target-policy smoothing (td3.py:83-93), delayed actor steps
    a.smooth = 1;
    a.policy_noise = (float)c.hp.policy_noise;
    a.noise_clip = (float)c.hp.noise_clip;
    a.max_action = (float)c.hp.max_action;
    a.do_actor = actor_due(h) ? 1 : 0;
  }
  if (c.algo == OPRL_SAC) {   // tanh-Gaussian actor, entropy term (sac.py:90-141)
    a.sac = 1;
    a.noise_pi = h->noise1_pending;
    a.rng_seed_pi = noise_key(h, 2);
    a.log_alpha = alpha_ptr(h);
    a.alpha_const = (float)c.hp.alpha_init;
    a.raw = h->raw;
    a.logp = h->logp;
    // both phase-2 clusters of a slice must be co-resident: cluster 0 waits for cluster 1's result
    a.p2_pair = (h->ncl == 4 && 2 * 4 * ((B + kR - 1) / kR) <= h->n_cus && !h->no_p2_pair) ? 1 : 0;
  }
  a.B = B; a.S = h->S; a.A = h->A;
  a.src = h->src;
  a.next = h->next_src;
  a.prefetch_next = 0;
  a.gamma = (float)c.hp.gamma;
  a.inv_B = 1.0f / (float)B;
  for (int l = 0; l < kMaxLayers; ++l) {
    a.cX[l] = h->ws_critic[0].X[l]; a.cdY[l] = h->ws_critic[0].dY[l];
    a.aX[l] = h->ws_actor.X[l];     a.adY[l] = h->ws_actor.dY[l];
  }
  a.cldx0 = h->ws_critic[0].ldx0; a.clddo = h->ws_critic[0].lddo;
  a.aldx0 = h->ws_actor.ldx0;     a.alddo = h->ws_actor.lddo;
  a.pi = h->pi;
  a.y_out = h->ydbg; a.q_out = h->qdbg;
  a.y_granules = h->y_granules; a.gran_stride = h->Bmax;
  a.gate_flags = h->y_granules + (size_t)3 * h->Bmax;   // 256 flag granules behind the TD / q granules
  a.merged = 0;
  a.epoch = h->epoch;
  a.trace = nullptr;
  a.nc = h->nc_cluster(B);
  a.no_lean = h->no_lean;
  // role A and the role-C cluster wait for each other: only with all four roles of a slice resident
  a.twin_split = (h->nc == 2 && a.nc == 4 && !h->no_lean && !h->no_twin_split &&
                  (2 + h->nc) * 4 * ((B + kR - 1) / kR) <= h->n_cus) ? 1 : 0;
  a.xbuf = h->xbuf;
  a.cdY0_stride = h->ws_critic[0].dY0_stride;
  a.adY0_stride = h->ws_actor.dY0_stride;
  a.partials_c = h->part_c; a.partials_a = h->part_a;
  a.err = h->err_dev;
  a.debug_expire = h->debug_expire;
  a.w3_src = c.actor.theta + w_off(c.actor, c.actor.n_layers - 1);
  if (h->x2 && fused_ddpg_is_lean(a)) {     // the PrecX2 instances: every net through its packs of two fp16 planes
    a.x2 = 1;
    a.actor = net_view16(c.actor, false, h->pack16[0], 2);
    if (c.actor.theta_target) a.actor_t = net_view16(c.actor, true, h->pack16_t[0], 2);
    a.critic = net_view16(c.critics[0], false, h->pack16[1], 2);
    a.critic_t = net_view16(c.critics[0], true, h->pack16_t[1], 2);
    if (h->nc == 2) {
      a.critic2 = net_view16(c.critics[1], false, h->pack16[2], 2);
      a.critic2_t = net_view16(c.critics[1], true, h->pack16_t[2], 2);
    }
  }
  if (h->bf16 && fused_ddpg_is_lean(a)) {   // the PrecBF16 instances of the (lean) phase kernels: every net through its bf16 packs
    a.bf16 = 1;
    a.actor = net_view16(c.actor, false, h->pack16[0]);
    if (c.actor.theta_target) a.actor_t = net_view16(c.actor, true, h->pack16_t[0]);
    a.critic = net_view16(c.critics[0], false, h->pack16[1]);
    a.critic_t = net_view16(c.critics[0], true, h->pack16_t[1]);
    if (h->nc == 2) {
      a.critic2 = net_view16(c.critics[1], false, h->pack16[2]);
      a.critic2_t = net_view16(c.critics[1], true, h->pack16_t[2]);
    }
  }
  // clusters of EIGHT for role A (DDPG: the roles then fill the chip exactly at B = 256) and for phase 2's
  // critic pass (DDPG / TD3), while the launch still fits the chip; exact-fp32 lean passes only
  // A property of the LEARNER (oprl_learner_set_cluster(h, 8) = the default / (h, 4) = never), not of the
  // moment: results differ in the last bits between cluster sizes (summation order of the exchanges).  A wide
  // launch wants the whole chip; FOUR such launches each cut in the middle of role A (64 members resident, 64
  // waiting for a CU) fill it with workgroups that spin for each other — measured with eight learners on eight
  // streams: every wait ran into its bound and was reported.  Learners that share a GPU with more than two
  // others (packed seeds on streams, one process per seed on one GPU) turn it off: set_cluster(h, 4) or
  // OPRL_AMD_NO_WIDE=1; up to three cannot dead-lock (the B roles always finish and free their CUs).
  a.xnc = h->xnc;
  a.wide = 0;
  if (h->xnc >= 8 && !h->no_wide && !a.sac && !a.bf16 && a.A <= 8 && fused_ddpg_is_lean(a)) {   // (narrow exchanges: <= 8 action columns)
    const int slices = (B + kR - 1) / kR;
    if (h->nc == 1 && (a.nc + 8 + a.nc) * slices <= h->n_cus) a.wide |= 1;
    if ((8 + 1) * slices <= h->n_cus) a.wide |= 2;
  }
  // merged launches (DDPG, lean passes, one 256-row chunk, this rank's own Adam step): the critic's dW tiles ride
  // on phase 1 — whose role A then stays on a cluster of four: 64 CUs must be free for tile workgroups from the start
  // (a gradient-exporting learner — data parallel over RCCL — merges only with the PrecX2 tiles, which know how to leave
  // dW in the gradient arena instead of running Adam: four launches per data-parallel update instead of six)
  const bool xport_ok = !h->cfg.export_grads || (a.x2 && fused_x2_tiles());
  // (dp_inline: the gradient exchange inside the dW tiles — the 16 x 32 tiles of k_dw_adam<true> as launches of their own,
  // or, PrecX2 learners, the 16 x 64 tiles of the merged / whole-update launches themselves: dw_tile_x2.h)
  const bool inline_x2 = h->dp_inline && a.x2 && fused_x2_tiles() && h->nc == 1;
  // exact-fp32 learners with mirrored packs: the whole update as k_ddpg_chain<PrecF32> when that form is possible at all
  // (there is no merged phase 2 with the fp32 tiles on its own: with the whole form out of reach the two bits below stay
  // what they were — merged phase 1 with role A on four, phase 2 and the actor's dW as launches)
  const bool whole_f32 = h->fchain && fused_x2_tiles() && !a.x2 && !a.bf16 && !h->no_whole && !h->no_merge && !h->no_merge2 && !h->shared_chip &&
                         !h->dp_inline && B <= 256 && fused_ddpg_is_lean(a) && (a.wide & 3) == 3 && h->chain_flags != nullptr &&
                         chain_rows(h, B) * ((B + kR - 1) / kR) <= h->n_cus;
  if (!h->no_merge && !h->shared_chip && h->nc == 1 && !a.sac && B <= 256 && xport_ok && (!h->dp_inline || inline_x2) && fused_ddpg_is_lean(a)) {
    a.merged |= 1;
    if (!(a.x2 && fused_x2_tiles()) && !whole_f32) a.wide &= ~1;     // (the 84 16 x 64 tiles of a PrecX2 learner get along with role A on eight)
  }
  // ... and the ACTOR's tiles on phase 2 (DDPG / TD3: the tanh head, action_dim <= kDuLd): the tiles form their dY from
  // du, the first layer's comes from one more backward step of the critic pass's members (csrc/fused_ddpg.hip).
  // PrecX2 learners only, the pass on clusters of eight: with the exact-fp32 tiles the merged form measured no faster
  // than the two launches (34.9 vs 34.7 us)
  if (!h->no_merge2 && !h->shared_chip && ((a.x2 && fused_x2_tiles()) || whole_f32) && h->du_granules != nullptr && !a.sac && B <= 256 && (!h->dp_inline || inline_x2) &&
      fused_ddpg_is_lean(a) && c.actor.theta_target != nullptr && (a.wide & 2) != 0) {
    a.merged |= 2;
    a.du_granules = h->du_granules;
    a.g1_granules = h->g1_granules;
    a.w3_snap = h->w3_snap;
  }
  // the whole update as ONE launch (k_ddpg_update): both merged forms, role A and the critic pass on eight, the 16 x 64
  // tiles, and everything the roles hand to each other in uncached memory
  if (!h->no_whole && (!h->cfg.export_grads || inline_x2) && ((a.x2 && fused_x2_tiles()) || whole_f32) && h->nc == 1 && (a.merged & 3) == 3 && (a.wide & 3) == 3 && h->uc_pool &&
      h->uc_base != nullptr && h->w_flags != nullptr && h->chain_flags != nullptr &&
      chain_rows(h, B) * ((B + kR - 1) / kR) <= h->n_cus) {      // (one update's workgroups wait for each other: all must fit the chip)
    a.whole = 1;
    a.w_flags = h->w_flags;
    a.ct_done = h->w_flags + 64;
    for (int l = 0; l < c.critics[0].n_layers; ++l) a.critic_b16[l] = h->critic_b16 + 256 * l;
  }
  if (!a.x2 && (a.merged & 2) != 0 && !a.whole) {     // (exact fp32: no merged phase 2 outside the whole form)
    a.merged &= ~2;
    a.wide &= ~1;
    a.du_granules = nullptr; a.g1_granules = nullptr; a.w3_snap = nullptr;
  }
  return a;
}

// DDPG runs fused for every batch size (generic tp3.h passes when the lean ones do not fit);
// TD3's and SAC's fused kernels exist in the lean form only, otherwise the generic launch sequence is used
bool use_fused(oprl_learner* h, int B) {
  if (!h->fused) return false;
  if (h->cfg.algo == OPRL_DDPG) return true;
  return fused_ddpg_is_lean(ddpg_args(h, B));
}

// The temperature step of this update as a job for the actor's dW launch (one more workgroup), when nothing
// but this rank's own Adam step is wanted of it; otherwise (gradient export, exchange inside the dW launch)
// the caller launches k_alpha_step itself.
bool alpha_rides(const oprl_learner* h) {
  return alpha_ptr(h) != nullptr && !h->cfg.export_grads && !h->dp_inline;
}
AlphaJob alpha_job(oprl_learner* h, int B) {
  const oprl_learner_config& c = h->cfg;
  h->opt_step_alpha += 1;
  AlphaJob j;
  j.log_alpha = c.log_alpha; j.m = c.log_alpha_m; j.v = c.log_alpha_v; j.logp = h->logp; j.B = B;
  j.target_entropy = (float)c.hp.target_entropy;
  j.lr = c.hp.lr_alpha; j.beta1 = c.hp.beta1; j.beta2 = c.hp.beta2; j.eps = c.hp.adam_eps;
  j.bc1 = 1.0 - std::pow(c.hp.beta1, (double)h->opt_step_alpha);
  j.bc2_sqrt = std::sqrt(1.0 - std::pow(c.hp.beta2, (double)h->opt_step_alpha));
  return j;
}

// (dw_build advances the optimiser's step count: call it once per launch, merged or not)
DwArgs dw_build(oprl_learner* h, bool critic, int B, bool polyak, bool with_alpha) {
  const oprl_learner_config& c = h->cfg;
  DwArgs dw;
  if (with_alpha) dw.alpha = alpha_job(h, B);
  if (critic) {
    h->opt_step_critic += 1;
    dw.items = h->items_host.data(); dw.n_items = h->n_items_critic; dw.total_tiles = h->tiles_critic;
    dw.ad = adam_scalars(h, c.hp.lr_critic, h->opt_step_critic, polyak, 1.0f);
  } else {
    h->opt_step_actor += 1;
    dw.items = h->items_host.data() + h->n_items_critic; dw.n_items = h->n_items_actor;
    dw.total_tiles = h->tiles_actor;
    dw.ad = adam_scalars(h, c.hp.lr_actor, h->opt_step_actor, polyak, 1.0f);
  }
  dw.B = B;
  const bool fused = use_fused(h, B);
  dw.n_part = fused ? h->nc_cluster(B) : 1;
  dw.trace = (fused && h->trace != nullptr) ? h->trace + (size_t)(critic ? 4 : 5) * 64 * kTraceStamps * 2 : nullptr;
  dw.apply_only = 0;
  // the lean phase 1 leaves unit-seed dz rows (tp4_scalar_fb): each critic's TD-error seed, dY of
  // its output layer, is applied per row (DwItem::rs)
  const bool lean = fused && fused_ddpg_is_lean(ddpg_args(h, B));
  dw.use_row_scale = (critic && lean) ? 1 : 0;
  dw.dy_tiled = (lean && dw.n_part > 1) ? 1 : 0;      // the lean passes leave tile-major dz1 partials
  if (h->x2 && lean && !c.export_grads) {             // the PrecX2 kernels read the fp16 packs only (fresh32)
    dw.skip32 = 1;
    h->stale32[critic ? 0 : 1] = true;
  }
  if (h->fchain) h->stale32[critic ? 0 : 1] = true;   // (the tiles write the mirrors: the caller's packs fall behind)
  return dw;
}

int dw_step(oprl_learner* h, bool critic, int B, bool polyak, hipStream_t st, bool with_alpha = false,
            const PrefetchJob* prefetch = nullptr) {
  DwArgs dw = dw_build(h, critic, B, polyak, with_alpha);
  dw.prefetch = prefetch;
  if (h->dp_inline) {
    // data-parallel on peer windows: this launch all-reduces its tiles itself and runs Adam on the mean
    P2pState& P = h->p2p;
    P.tile_seq += 1;
    DwXchg& X = h->dw_xchg;
    for (int r = 0; r < kDwXchgMaxWorld; ++r) X.peer[r] = r < P.world ? P.peer[r] + P.tile_off : nullptr;
    X.window = P.window + P.tile_off;
    X.world = P.world; X.rank = P.rank; X.parity = (int)(P.tile_seq & 1); X.max_tiles = h->p2p_max_tiles;
    X.seq = P.tile_seq;
    X.err = h->err_dev;
    dw.xchg = &X;
    dw.ad.do_adam = 1;
    dw.ad.grad_scale = 1.0f / (float)P.world;
  }
  HIPC(launch_dw_prof(dw, st));
  return OPRL_OK;
}

// the actor's forward on s with the activations kept for its backward (actor_phase step 5)
MlpArgs actor_forward_args(oprl_learner* h, const float* s, int B, const float* noise1) {
  const oprl_learner_config& c = h->cfg;
  const bool gauss = (c.algo == OPRL_SAC || c.algo == OPRL_TQC);
  MlpArgs f = base_args(h, c.actor, false, B);
  f.do_fwd = 1;
  f.x0 = s; f.k0 = h->S;
  with_store(f, h->ws_actor, true, false);
  f.out = h->pi; f.ldo = h->A;
  if (gauss) {
    f.out_act = ACT_GAUSS; f.raw_out = h->raw; f.ldraw = 2 * h->A; f.logp = h->logp;
    seed_rng(f, h, noise1, 2);
  } else {
    f.out_act = ACT_TANH;
  }
  return f;
}

// ------------------------------------------------------------ critic phase
int critic_phase(oprl_learner* h, const float* s, const float* a, const float* r, const float* d,
                 const float* s2, int B, const float* noise0, hipStream_t st) {
  const oprl_learner_config& c = h->cfg;
  if (use_fused(h, B)) {
    h->epoch += 1;
    if (h->epoch == 0 || h->epoch > 0xFFFFFFFFu - (unsigned)kChainMax) {   // the TD-target tag wrapped (or would inside a chain launch): retire every stale granule
      h->epoch = 1;
      HIPC(hipMemsetAsync(h->y_granules, 0, ((size_t)3 * h->Bmax + 256) * sizeof(unsigned long long), st));
      if (h->du_granules != nullptr) {
        HIPC(hipMemsetAsync(h->du_granules, 0, (size_t)(h->Bmax < 256 ? h->Bmax : 256) * kDuLd * sizeof(unsigned long long), st));
        HIPC(hipMemsetAsync(h->g1_granules, 0, (size_t)16 * (h->Bmax < 256 ? h->Bmax : 256) * 16 * sizeof(unsigned long long), st));
        HIPC(hipMemsetAsync(h->w_flags, 0, 256 * sizeof(unsigned long long), st));
        HIPC(hipMemsetAsync(h->chain_flags, 0, (192 + 192 + 64 + 128 + 1024) * sizeof(unsigned long long), st));
      }
    }
    DdpgArgs fa = ddpg_args(h, B);
    if (h->x2 && !fa.x2) RC(fresh32_tables(h, 3, st));     // (a batch the lean kernels do not take: fp32 packs)
    fa.noise = noise0;
    if (h->prefetch_p1 && h->prefetch_next) {
      fa.prefetch_p1 = 1;
      h->staged_ready = true;
    }
    RC(next_tp_tag(&h->tp_tag, h->xbuf, h->xbuf_granules * sizeof(unsigned long long), st, &fa.cluster_tag));
    if (h->trace != nullptr) fa.trace = h->trace;   // roles use slots 0 .. 1 + n_critics
    h->whole_done = false;
    if (fa.whole && B <= 256) {
      // the whole update as ONE launch (k_ddpg_update): phase 1's roles, the critic's tiles, role U + the actor's
      // tiles, the critic pass
      RC(next_tp_tag(&h->tp_tag, h->xbuf, h->xbuf_granules * sizeof(unsigned long long), st, &fa.cluster_tag2));
      if (h->trace != nullptr && c.algo == OPRL_DDPG) fa.trace2 = h->trace + (size_t)3 * 64 * kTraceStamps * 2;   // slot 3
      fa.prefetch_next = 0;
      const int slices = (B + kR - 1) / kR;
      DwKArgs kd;
      DwKArgs4 kc, ka;
      auto compact = [&](const DwKArgs& k, DwKArgs4* o) {
        for (int j = 0; j < kDwFusedItems; ++j) { o->tile_end[j] = k.tile_end[j]; o->items[j] = k.items[j]; }
        o->n_items = k.n_items; o->B = k.B; o->n_part = k.n_part; o->dy_tiled = k.dy_tiled; o->ad = k.ad; o->trace = k.trace;
        o->gate = k.gate;
        memset((void*)&o->xchg, 0, sizeof o->xchg);
      };
      {
        DwArgs dw = dw_build(h, true, B, true, false);
        if (fill_dw_kargs(dw, &kd, 64) < 0 || dw.n_items > kDwFusedItems) { set_err("whole update: bad critic dW table"); return OPRL_ERR_INVALID; }
        kd.gate.rows = fa.gate_flags; kd.gate.n_rows = 4 * slices;
        kd.gate.seed = fa.y_granules; kd.gate.n_seed = B;
        kd.gate.late_dY = h->ws_critic[0].dY[c.critics[0].n_layers - 1];
        kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
        kd.gate.err = h->err_dev; kd.gate.err_code = (1u << 8) | 7u;
        kd.gate.done = fa.ct_done;
        fa.n_ct = kd.tile_end[kDwMaxItems - 1];
        if (fa.n_ct > 192 - 64) { set_err("whole update: too many critic tiles"); return OPRL_ERR_INVALID; }
        compact(kd, &kc);
      }
      {
        DwArgs dw = dw_build(h, false, B, true, false);
        if (fill_dw_kargs(dw, &kd, 64) < 0 || dw.n_items != 3) { set_err("whole update: bad actor dW table"); return OPRL_ERR_INVALID; }
        kd.gate.seed = fa.du_granules; kd.gate.n_seed = B;
        kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
        kd.gate.err = h->err_dev; kd.gate.err_code = (2u << 8) | 7u;
        kd.gate.kind[0] = 2; kd.gate.kind[1] = 1; kd.gate.kind[2] = 0; kd.gate.kind[3] = 0;
        kd.gate.h2 = h->ws_actor.X[2];
        kd.gate.w3 = fa.w3_snap;
        kd.gate.g1 = fa.g1_granules;
        kd.gate.n_act = h->A;
        compact(kd, &ka);
      }
      // (every tile must find a role-B / role-C workgroup to be the continuation of: 8 per slice)
      const int max_tiles = kc.tile_end[kDwFusedItems - 1] > ka.tile_end[kDwFusedItems - 1] ? kc.tile_end[kDwFusedItems - 1] : ka.tile_end[kDwFusedItems - 1];
      const int t_rows = max_tiles > 8 * slices ? (max_tiles - 8 * slices + slices - 1) / slices : 0;
      // (one update's workgroups — 16 role rows per slice + the tile-only rows — wait for each other: all must fit the chip)
      const bool chain_fits = (16 + t_rows) * slices <= h->n_cus;
      const int U = (h->no_chain || h->chain_flags == nullptr || !chain_fits) ? 1 : h->chain_u;
      memset((void*)&kc.xchg, 0, sizeof kc.xchg);
      memset((void*)&ka.xchg, 0, sizeof ka.xchg);
      if (h->dp_inline) {
        // data parallel on peer windows: the tiles of this launch all-reduce their gradients themselves and run Adam on
        // the mean (dw_tile_x2.h): exchange sequence numbers seq0 + 2 u (critic) / + 1 (actor) for update u of the launch
        P2pState& P = h->p2p;
        if (max_tiles > h->p2p_max_tiles) { set_err("data-parallel whole update: the windows hold %d tiles, the launch has %d", h->p2p_max_tiles, max_tiles); return OPRL_ERR_STATE; }
        DwXchg X;
        memset((void*)&X, 0, sizeof X);
        for (int r = 0; r < kDwXchgMaxWorld; ++r) X.peer[r] = r < P.world ? P.peer[r] + P.tile_off : nullptr;
        X.window = P.window + P.tile_off;
        X.world = P.world; X.rank = P.rank; X.max_tiles = h->p2p_max_tiles;
        X.seq = P.tile_seq + 1; X.parity = 0;        // (parity = the exchange's sequence number & 1, formed in the tile)
        X.err = h->err_dev;
        P.tile_seq += 2ull * (unsigned long long)U;
        kc.xchg = X; ka.xchg = X;
        kc.ad.do_adam = 1; ka.ad.do_adam = 1;
        kc.ad.grad_scale = 1.0f / (float)P.world; ka.ad.grad_scale = 1.0f / (float)P.world;
        h->stale32[0] = true; h->stale32[1] = true;  // (the tiles write the fp16 packs only)
      }
      if (!h->no_chain && h->chain_flags != nullptr && chain_fits) {
        // SEVERAL updates as one launch (k_ddpg_chain): the tables above are update 0's; what changes per update — Adam's
        // bias-correction terms, epochs, exchange tags, the staging set — is in ChainArgs
        ChainArgs ca;
        memset((void*)&ca, 0, sizeof ca);
        ca.n_upd = U;
        ca.first_gather = fa.src.gather;
        ca.pf_last = h->chain_pf_last ? 1 : 0;
        ca.trace_u = U - 1;
        ca.rows = 16 + t_rows;
        ca.c_step[0] = kc.ad.step_size_host; ca.c_bc2[0] = kc.ad.bc2_sqrt_host;
        ca.a_step[0] = ka.ad.step_size_host; ca.a_bc2[0] = ka.ad.bc2_sqrt_host;
        for (int u = 1; u < U; ++u) {          // (dw_build advances the optimisers' step counts: once per update and net)
          const DwArgs dc = dw_build(h, true, B, true, false);
          const DwArgs da = dw_build(h, false, B, true, false);
          ca.c_step[u] = dc.ad.step_size_host; ca.c_bc2[u] = dc.ad.bc2_sqrt_host;
          ca.a_step[u] = da.ad.step_size_host; ca.a_bc2[u] = da.ad.bc2_sqrt_host;
        }
        ca.set0[0] = fa.src.s; ca.set0[1] = fa.src.a; ca.set0[2] = fa.src.r; ca.set0[3] = fa.src.d; ca.set0[4] = fa.src.s2;
        for (int i = 0; i < 5; ++i) ca.set1[i] = h->chain_set1[i];
        if ((U > 1 || ca.pf_last) && ca.set1[0] == nullptr) { set_err("chain launch: no second staging set"); return OPRL_ERR_STATE; }
        ca.ct_fin = h->chain_flags; ca.at_fin = h->chain_flags + 192; ca.pf_done = h->chain_flags + 384;
        for (int w = 0; w < 4; ++w)
          for (int l = 0; l < kMaxLayers; ++l) ca.b16[w][l] = h->chain_b16 + ((size_t)w * kMaxLayers + l) * 256;
        ca.w3buf[0] = h->w3_snap; ca.w3buf[1] = h->w3buf1;
        // the first hidden layer's dY from the unit-seed rows the pass's members leave before the pass (DwGate kind 3)
        ca.qp = h->chain_flags + 576;
        fa.gu = h->gu; fa.gu_flags = h->chain_flags + 448;
        ka.gate.kind[0] = 3; ka.gate.gu = h->gu; ka.gate.gu_flags = fa.gu_flags; ka.gate.n_gu_flags = 8 * slices;
        if (kc.tile_end[kDwFusedItems - 1] > 192 || ka.tile_end[kDwFusedItems - 1] > 192 || slices > 64) { set_err("chain launch: too many tiles"); return OPRL_ERR_INVALID; }
        // exchange tags: two per update (the roles', the critic pass's), consecutive: cluster_tag = the first
        {
          unsigned& ctr = h->tp_tag;
          if ((ctr & 0x03FFFFFFu) + 2u * (unsigned)U + 2u >= 0x03FFFFFFu) {      // (would wrap inside the launch: wrap now)
            ctr = (ctr | 0x03FFFFFFu) + 1u;
            HIPC(hipMemsetAsync(h->xbuf, 0, h->xbuf_granules * sizeof(unsigned long long), st));
          }
          fa.cluster_tag = (ctr + 1u) & 0x03FFFFFFu;
          ctr += 2u * (unsigned)U;
        }
        fa.prefetch_p1 = 0;
        fa.trace2 = fa.trace != nullptr ? h->trace + (size_t)3 * 64 * kTraceStamps * 2 : nullptr;
        h->epoch += (unsigned)(U - 1);
        prof_begin(4, st);
        hipError_t e;
        {
          std::lock_guard<std::mutex> lk(g_turn.mu);
          int dev = 0;
          e = chip_turn_begin(st, &dev);
          if (e == hipSuccess) e = launch_ddpg_chain(fa, kc, ka, ca, st);
          if (e == hipSuccess) chip_turn_end(st, dev);
        }
        prof_end(st);
        HIPC(e);
        h->whole_done = true;
        return OPRL_OK;
      }
      set_err("whole update: one update's workgroups do not fit this device (%d compute units)", h->n_cus);
      return OPRL_ERR_STATE;
    }
    if ((fa.merged & 1) != 0) {
      // phase 1 and the critic's dW + Adam tiles as ONE launch: the tiles wait for the roles' flag granules
      DwArgs dw = dw_build(h, true, B, true, false);
      DwKArgs kd;
      if (fill_dw_kargs(dw, &kd, (fa.x2 && fused_x2_tiles()) ? 64 : 32) < 0) { set_err("merged phase 1: bad dW table"); return OPRL_ERR_INVALID; }
      const int slices = (B + kR - 1) / kR;
      kd.gate.rows = fa.gate_flags; kd.gate.n_rows = 4 * slices;
      kd.gate.seed = fa.y_granules; kd.gate.n_seed = B;      // the seeds come as granules, one per row
      kd.gate.late_dY = h->ws_critic[0].dY[c.critics[0].n_layers - 1];
      kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
      kd.gate.err = h->err_dev; kd.gate.err_code = (1u << 8) | 7u;      // KERN_PHASE1, SITE_DW_GATE (csrc/tp3.h)
      prof_begin(4, st);
      hipError_t e = launch_ddpg_phase1_dw(fa, kd, st);
      prof_end(st);
      HIPC(e);
      return OPRL_OK;
    }
    prof_begin(4, st);
    hipError_t e = launch_ddpg_phase1(fa, st);
    prof_end(st);
    HIPC(e);
    // TD3 moves its targets only on actor steps (td3.py:135-146)
    return dw_step(h, true, B, c.algo == OPRL_TD3 ? actor_due(h) : true, st);
  }
  RC(fresh32_tables(h, 3, st));      // (a PrecX2 learner's generic launches read the fp32 packs)
  const int S = h->S, A = h->A, nc = h->nc;
  const int algo = c.algo;
  const int n_slices = (B + kR - 1) / kR;
  // 1. next action
  {
    const bool use_target_actor = (algo == OPRL_DDPG || algo == OPRL_TD3);
    MlpArgs f = base_args(h, c.actor, use_target_actor, B);
    f.do_fwd = 1;
    f.x0 = s2; f.k0 = S;
    f.out = h->a2; f.ldo = A;
    if (algo == OPRL_DDPG) f.out_act = ACT_TANH;
    else if (algo == OPRL_TD3) { f.out_act = ACT_TANH_SMOOTH; seed_rng(f, h, noise0, 1); }
    else { f.out_act = ACT_GAUSS; f.logp = h->logp2; seed_rng(f, h, noise0, 1); }
    // TQC: this launch is 64 workgroups on 256 CUs, and the online critics' first two layers on (s, a) — the
    // first launch of step 3 — depend on nothing of it: they ride behind it (k_slice_tp_fin, layerwise.hip)
    bool fin_ride = false, fin16 = false;
    MlpArgs* fin_args = h->fin_args;
    h->fin_done = false;
    h->fin_tail0 = -1;
    if (algo == OPRL_TQC && !h->no_fin_ride && !h->no_layerwise && !h->no_multi && nc > 2 && nc <= kMaxMulti &&
        h->lw_scratch != nullptr && h->w_critic == 512 && f.tp_xbuf != nullptr) {
      fin16 = h->bf16 || h->x2;
      for (int j = 0; j < nc; ++j) {
        MlpArgs g = base_args(h, c.critics[j], false, B);
        g.do_fwd = 1; g.do_bwd = 1;
        g.x0 = s; g.k0 = S; g.x1 = a; g.k1 = A;
        with_store(g, h->ws_critic[j], true, true);
        for (int l = 1; l + 1 < g.net.n_layers; ++l) fin16 = fin16 && g.pf16[l] != nullptr && g.pb16[l] != nullptr;
        fin_args[j] = g;
      }
      if (fin16)
        for (int j = 0; j < nc; ++j)
          for (int l = 1; l + 1 < fin_args[j].net.n_layers; ++l) fin_args[j].net.pf[l] = fin_args[j].pf16[l];
      fin_ride = mlp_layerwise_fin_ok(fin_args, nc, h->w_critic);
    }
    // as many nets as fit beside the forward's 4 x slices workgroups with everything resident at once (4 of TQC's 5)
    const int n_ride = fin_ride ? mlp_layerwise_fin_fit(fin_args, nc, 4 * n_slices, h->n_cus) : 0;
    if (fin_ride && n_ride >= 1) {
      MlpArgs ff = f;
      RC(next_tp_tag(ff.tp_tag_counter, ff.tp_xbuf, ff.tp_xbuf_bytes, st, &ff.tp_tag));
      prof_begin(0, st);
      hipError_t e = launch_slice_tp_with_fin(ff, fin_args, nc, n_ride, h->w_critic, h->n_cus, st, fin16 ? (h->x2 ? 2 : 1) : 0);
      prof_end(st);
      HIPC(e);
      h->fin_done = true;
      h->fin16 = fin16;
      if (n_ride < nc) h->fin_tail0 = n_ride;
    } else {
      RC(launch(f, h->w_actor, st));
    }
  }
  // 2. target critics on (s', a')   (independent: one stream each)
  if (algo == OPRL_TQC && h->tqc_counter != nullptr && !h->no_tqc_ride) {
    TqcJob& J = h->tqc_job;
    J.counter = h->tqc_counter;
    J.z = h->qn; J.net_stride = (long)h->Bmax * h->ldq; J.ldz = h->ldq;
    J.n_nets = nc; J.Q = c.hp.n_quantiles; J.drop = c.hp.top_quantiles_to_drop;
    J.r = r; J.d = d; J.logp = h->logp2; J.log_alpha = c.log_alpha; J.gamma = (float)c.hp.gamma;
    J.target = h->target;
    h->tqc_job_pending = true;
  }
  RC(for_each_net(h, nc, st, [&](int j, hipStream_t sj) {
    MlpArgs f = base_args(h, c.critics[j], true, B);
    f.do_fwd = 1;
    f.x0 = s2; f.k0 = S; f.x1 = h->a2; f.k1 = A;
    f.out = h->qn + (size_t)j * h->Bmax * h->ldq; f.ldo = h->ldq;
    return launch(f, h->w_critic, sj);
  }));
  if (h->fin_tail0 >= 0) {                           // (the rest of the early first launch found no head launch to ride on)
    prof_begin(0, st);
    hipError_t e = launch_mlp_layerwise_first(h->fin_args, nc, h->fin_tail0, h->w_critic, h->n_cus, st, h->fin16 ? (h->x2 ? 2 : 1) : 0);
    prof_end(st);
    HIPC(e);
    h->fin_tail0 = -1;
  }
  if (algo == OPRL_TQC && h->tqc_job_pending) {     // (the job did not ride: not the layer-by-layer path)
    h->tqc_job_pending = false;
    const int Q = c.hp.n_quantiles, drop = c.hp.top_quantiles_to_drop;
    HIPC(launch_tqc_target(h->qn, (long)h->Bmax * h->ldq, h->ldq, nc, Q, drop, r, d, h->logp2,
                           c.log_alpha, (float)c.hp.gamma, B, h->target, st));
  } else if (algo == OPRL_TQC && (h->tqc_counter == nullptr || h->no_tqc_ride)) {
    const int Q = c.hp.n_quantiles, drop = c.hp.top_quantiles_to_drop;
    HIPC(launch_tqc_target(h->qn, (long)h->Bmax * h->ldq, h->ldq, nc, Q, drop, r, d, h->logp2,
                           c.log_alpha, (float)c.hp.gamma, B, h->target, st));
  }
  // 3. online critics: forward + loss seed + backward   (independent: one stream each)
  // TQC: the actor step's forward on s depends on nothing of the critic step: offered as a rider of this launch's
  // heads (k_lw_head: 80 workgroups on 256 CUs), with the noise and the counter actor_phase would give it
  h->rider_pending = false;
  h->rider_done = false;
  if (algo == OPRL_TQC && !h->no_af_ride && !c.export_grads && actor_due(h)) {
    MlpArgs f = actor_forward_args(h, s, B, h->noise1_pending);
    if (f.tp_xbuf != nullptr) {
      RC(next_tp_tag(f.tp_tag_counter, f.tp_xbuf, f.tp_xbuf_bytes, st, &f.tp_tag));
      h->rider = f;
      h->rider_pending = true;
    }
  }
  RC(for_each_net(h, nc, st, [&](int j, hipStream_t sj) {
    MlpArgs f = base_args(h, c.critics[j], false, B);
    f.do_fwd = 1; f.do_bwd = 1;
    f.x0 = s; f.k0 = S; f.x1 = a; f.k1 = A;
    with_store(f, h->ws_critic[j], true, true);
    f.partials = h->part_c + (size_t)j * n_slices * 4;
    SeedArgs& sd = f.seed;
    if (algo == OPRL_TQC) {
      const int Q = c.hp.n_quantiles, M = nc * Q - c.hp.top_quantiles_to_drop;
      f.seed_mode = SEED_QHUBER;
      sd.p0 = h->target; sd.M = M; sd.Q = Q;
      sd.cval = 1.0f / ((float)B * (float)nc * (float)Q * (float)M);
    } else {
      f.seed_mode = SEED_MSE_TD;
      sd.p0 = h->qn;
      sd.p1 = nc > 1 ? h->qn + (size_t)h->Bmax * h->ldq : nullptr;
      sd.p2 = (algo == OPRL_SAC) ? h->logp2 : nullptr;
      sd.log_alpha = alpha_ptr(h); sd.alpha_const = (float)c.hp.alpha_init;
      sd.r = r; sd.d = d; sd.gamma = (float)c.hp.gamma;
      sd.cval = 1.0f / (float)B;
      if (j == 0) { sd.y_out = h->ydbg; sd.q_out = h->qdbg; }
    }
    return launch(f, h->w_critic, sj);
  }));
  h->rider_pending = false;          // (not taken: actor_phase launches the forward itself)
  if (h->fin_done) {                 // the early first launch was not picked up: step 3 did not run layer by layer
    h->fin_done = false;
    set_err("TQC critic step: the online critics' early first launch has no layer-by-layer continuation");
    return OPRL_ERR_STATE;
  }
  // 4. dW + Adam (+ Polyak where the reference does it every step)
  {
    const bool polyak = (algo == OPRL_TD3) ? (h->update_count % c.hp.policy_freq == 0) : true;
    h->opt_step_critic += 1;
    DwArgs dw;
    dw.items = h->items_host.data(); dw.n_items = h->n_items_critic; dw.total_tiles = h->tiles_critic;
    dw.B = B; dw.n_part = tp_generic(h, c.critics[0], B) ? 4 : 1; dw.use_row_scale = 0; dw.apply_only = 0;
    dw.dy_tiled = dw.n_part > 1 ? 1 : 0;               // k_mlp_slice_tp runs the tp4 passes
    dw.trace = h->trace != nullptr ? h->trace + (size_t)4 * 64 * kTraceStamps * 2 : nullptr;   // slot 4
    dw.ad = adam_scalars(h, c.hp.lr_critic, h->opt_step_critic, polyak, 1.0f);
    // TQC in a 16-bit mode: the critics' 512 x 512 layers run through the 16-bit packs (layerwise.hip); their fp32
    // packs — a third of the wide dW launch's stores — are left stale and rebuilt by whoever reads them (fresh32)
    if (algo == OPRL_TQC && (h->x2 || h->bf16) && h->lazy_wide && !c.export_grads && !h->no_layerwise) {
      dw.skip32_wide = 1;
      h->stale_wide = true;
    }
    HIPC(launch_dw_prof(dw, st));
  }
  return OPRL_OK;
}

// ------------------------------------------------------------- actor phase
int actor_phase(oprl_learner* h, const float* s, int B, const float* noise1, hipStream_t st) {
  const oprl_learner_config& c = h->cfg;
  if (h->whole_done) {          // (this update's actor step ran inside the critic phase's launch: k_ddpg_update)
    h->whole_done = false;
    return OPRL_OK;
  }
  if (use_fused(h, B)) {
    DdpgArgs fa = ddpg_args(h, B);
    if (h->x2 && !fa.x2) RC(fresh32_tables(h, 3, st));
    RC(next_tp_tag(&h->tp_tag, h->xbuf, h->xbuf_granules * sizeof(unsigned long long), st, &fa.cluster_tag));
    if (h->trace != nullptr && c.algo == OPRL_DDPG) fa.trace = h->trace + (size_t)3 * 64 * kTraceStamps * 2;   // slot 3
    fa.prefetch_next = h->prefetch_p1 ? 0 : h->prefetch_next;
    // Where phase 2's own workgroups already fill the chip (SAC at B = 1024: 4 x 64), its prefetch row is a round of
    // its own; the actor's dW launch, which follows and leaves 40 % of the chip idle, carries the row instead
    // (prefetch_rows_direct, the same rows).
    bool pf_on_dw = false;
    PrefetchJob pj;
    {
      const int slices = (B + kR - 1) / kR;
      const int rows = (fa.wide & 2) != 0 ? 8 : fa.nc * ((fa.sac && fa.p2_pair) ? 2 : 1);
      if (fa.prefetch_next && (rows + 1) * slices > h->n_cus && !h->dp_inline) {
        pf_on_dw = true;
        fa.prefetch_next = 0;
        memset((void*)&pj, 0, sizeof pj);
        pj.next = fa.next; pj.S = h->S; pj.A = h->A; pj.B = B; pj.z0 = -1;
      }
    }
    if (h->prefetch_next && !h->prefetch_p1) h->staged_ready = true;
    if ((fa.merged & 2) != 0 && !pf_on_dw) {
      // phase 2 and the actor's dW + Adam tiles as ONE launch: the tiles wait for the du (first layer: dz1) granules
      DwArgs dw = dw_build(h, false, B, true, false);
      DwKArgs kd;
      if (fill_dw_kargs(dw, &kd, (fa.x2 && fused_x2_tiles()) ? 64 : 32) < 0 || dw.n_items != 3) { set_err("merged phase 2: bad dW table"); return OPRL_ERR_INVALID; }
      kd.gate.seed = fa.du_granules; kd.gate.n_seed = B;
      kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
      kd.gate.err = h->err_dev; kd.gate.err_code = (2u << 8) | 7u;      // KERN_PHASE2, SITE_DW_GATE (csrc/tp3.h)
      kd.gate.kind[0] = 2; kd.gate.kind[1] = 1; kd.gate.kind[2] = 0; kd.gate.kind[3] = 0;   // items = the actor's layers 0, 1, 2
      kd.gate.h2 = h->ws_actor.X[2];
      kd.gate.w3 = fa.w3_snap;
      kd.gate.g1 = fa.g1_granules;
      kd.gate.n_act = h->A;
      prof_begin(5, st);
      hipError_t e = launch_ddpg_phase2_dw(fa, kd, st);
      prof_end(st);
      HIPC(e);
      return OPRL_OK;
    }
    prof_begin(5, st);
    hipError_t e = launch_ddpg_phase2(fa, st);
    prof_end(st);
    HIPC(e);
    const bool rides = alpha_rides(h);      // SAC temperature (sac.py:129-141), from role C's log pi
    RC(dw_step(h, false, B, c.actor.theta_target != nullptr, st, rides, pf_on_dw ? &pj : nullptr));
    if (alpha_ptr(h) != nullptr && !rides) {
      h->opt_step_alpha += 1;
      HIPC(launch_alpha_step(c.log_alpha, c.log_alpha_m, c.log_alpha_v, h->logp, B,
                             (float)c.hp.target_entropy, c.hp.lr_alpha, c.hp.beta1, c.hp.beta2,
                             c.hp.adam_eps, h->opt_step_alpha,
                             c.export_grads ? h->alpha_grad : nullptr, nullptr, 1.0f, st));
    }
    return OPRL_OK;
  }
  RC(fresh32_tables(h, 3, st));
  const int S = h->S, A = h->A, nc = h->nc;
  const int algo = c.algo;
  const bool gauss = (algo == OPRL_SAC || algo == OPRL_TQC);
  const int n_slices = (B + kR - 1) / kR;
  // 5. actor forward (activations kept for its backward) — unless it rode on the critic step's heads (critic_phase)
  if (h->rider_done) {
    h->rider_done = false;
  } else {
    const MlpArgs f = actor_forward_args(h, s, B, noise1);
    RC(launch(f, h->w_actor, st));
  }
  // 6./7. critics on (s, pi): gradient wrt the action columns
  const int n_q = (algo == OPRL_TD3 || algo == OPRL_DDPG) ? 1 : nc;   // TD3 uses Q1 only
  if (algo == OPRL_SAC) {
    RC(for_each_net(h, nc, st, [&](int j, hipStream_t sj) {   // both q's before either seed (min)
      MlpArgs f = base_args(h, c.critics[j], false, B);
      f.do_fwd = 1;
      f.x0 = s; f.k0 = S; f.x1 = h->pi; f.k1 = A;
      with_store(f, h->ws_critic[j], true, false);
      f.out = h->qpi + (size_t)j * h->Bmax; f.ldo = 1;
      return launch(f, h->w_critic, sj);
    }));
    RC(for_each_net(h, nc, st, [&](int j, hipStream_t sj) {
      MlpArgs f = base_args(h, c.critics[j], false, B);
      f.do_bwd = 1;
      with_store(f, h->ws_critic[j], true, false);
      f.seed_mode = SEED_MINQ;
      f.seed.p0 = h->qpi; f.seed.p1 = h->qpi + h->Bmax; f.seed.which = j;
      f.seed.cval = 1.0f / (float)B;
      f.dact_col0 = S; f.dact_cols = A; f.dact = h->da + (size_t)j * h->Bmax * A; f.lddact = A;
      if (j == 0) f.partials = h->part_a;
      return launch(f, h->w_critic, sj);
    }));
  } else {
    RC(for_each_net(h, n_q, st, [&](int j, hipStream_t sj) {
      MlpArgs f = base_args(h, c.critics[j], false, B);
      f.do_fwd = 1; f.do_bwd = 1;
      f.x0 = s; f.k0 = S; f.x1 = h->pi; f.k1 = A;
      f.seed_mode = SEED_CONST;
      f.seed.cval = (algo == OPRL_TQC) ? -1.0f / ((float)B * (float)nc * (float)c.hp.n_quantiles)
                                       : -1.0f / (float)B;
      f.dact_col0 = S; f.dact_cols = A; f.dact = h->da + (size_t)j * h->Bmax * A; f.lddact = A;
      if (j == 0) f.partials = h->part_a;
      return launch(f, h->w_critic, sj);
    }));
  }
  // 8. actor backward from the stored activations
  {
    MlpArgs f = base_args(h, c.actor, false, B);
    f.do_bwd = 1;
    with_store(f, h->ws_actor, true, true);
    SeedArgs& sd = f.seed;
    if (gauss) {
      f.seed_mode = SEED_GAUSS;
      sd.p0 = h->da; sd.ld0 = A; sd.n_da = n_q; sd.da_stride = (long)h->Bmax * A;
      sd.p1 = h->raw;
      seed_rng(f, h, noise1, 2);   // backward re-reads (or re-draws) the forward's eps
      sd.log_alpha = alpha_ptr(h); sd.alpha_const = (float)c.hp.alpha_init;
      sd.cval = 1.0f / (float)B;
    } else {
      f.seed_mode = SEED_TANH;
      sd.p0 = h->da; sd.ld0 = A; sd.p1 = h->pi;
    }
    RC(launch(f, h->w_actor, st));
  }
  // 9. dW + Adam (+ Polyak of the actor target for DDPG / TD3)
  {
    h->opt_step_actor += 1;
    DwArgs dw;
    dw.items = h->items_host.data() + h->n_items_critic; dw.n_items = h->n_items_actor;
    dw.total_tiles = h->tiles_actor; dw.B = B; dw.n_part = tp_generic(h, c.actor, B) ? 4 : 1; dw.use_row_scale = 0; dw.apply_only = 0;
    dw.dy_tiled = dw.n_part > 1 ? 1 : 0;
    dw.trace = h->trace != nullptr ? h->trace + (size_t)5 * 64 * kTraceStamps * 2 : nullptr;   // slot 5
    dw.ad = adam_scalars(h, c.hp.lr_actor, h->opt_step_actor, c.actor.theta_target != nullptr, 1.0f);
    if (alpha_rides(h)) dw.alpha = alpha_job(h, B);
    HIPC(launch_dw_prof(dw, st));
  }
  // 10. temperature (when it did not ride on the actor's dW launch)
  if (alpha_ptr(h) != nullptr && !alpha_rides(h)) {
    h->opt_step_alpha += 1;
    HIPC(launch_alpha_step(c.log_alpha, c.log_alpha_m, c.log_alpha_v, h->logp, B,
                           (float)c.hp.target_entropy, c.hp.lr_alpha, c.hp.beta1, c.hp.beta2,
                           c.hp.adam_eps, h->opt_step_alpha,
                           c.export_grads ? h->alpha_grad : nullptr, nullptr, 1.0f, st));
  }
  (void)n_slices;
  return OPRL_OK;
}

// master -> packs for a list of nets (which: bit0 online, bit1 target).  The item
// table goes through a small device scratch; synchronous on `st` only.
void build_repack_items(const oprl_net* const* nets, int n_nets, int which,
                        std::vector<RepackItem>& items, int* blocks_out,
                        float* const* pk16 = nullptr, float* const* pk16_t = nullptr, int pl = 1) {
  int blocks = 0;
  for (int i = 0; i < n_nets; ++i) {
    const oprl_net& n = *nets[i];
    for (int pass = 0; pass < 2; ++pass) {
      if (!(which & (1 << pass))) continue;
      const float* base = pass == 0 ? n.theta : n.theta_target;
      float* pk = pass == 0 ? n.pack : n.pack_target;
      if (!base || !pk) continue;
      for (int l = 0; l < n.n_layers; ++l) {
        RepackItem it;
        it.w = base + w_off(n, l);
        it.N = n.dims[l + 1]; it.K = n.dims[l];
        it.pf = pk + pack_off_fwd(n, l);
        it.pb = pass == 0 ? pk + pack_off_bwd(n, l) : nullptr;
        float* p16 = pass == 0 ? (pk16 ? pk16[i] : nullptr) : (pk16_t ? pk16_t[i] : nullptr);
        it.pf16 = p16 ? p16 + pack16_off_fwd(n, l, pl) : nullptr;
        it.pb16 = (p16 && pass == 0) ? p16 + pack16_off_bwd(n, l, pl) : nullptr;
        it.x2 = pl == 2 ? 1 : 0;
        it.blk_begin = blocks;
        blocks += (int)(((long)it.N * it.K + 255) / 256);
        it.blk_end = blocks;
        items.push_back(it);
      }
    }
  }
  *blocks_out = blocks;
}

int repack_nets(const oprl_net* const* nets, int n_nets, int which, hipStream_t st,
                float* const* pk16 = nullptr, float* const* pk16_t = nullptr, int pl = 1) {
  std::vector<RepackItem> items;
  int blocks = 0;
  build_repack_items(nets, n_nets, which, items, &blocks, pk16, pk16_t, pl);
  if (items.empty()) return OPRL_OK;
  RepackItem* dev = nullptr;
  HIPC(hipMalloc(&dev, sizeof(RepackItem) * items.size()));
  hipError_t e = hipMemcpyAsync(dev, items.data(), sizeof(RepackItem) * items.size(), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = launch_repack(dev, (int)items.size(), blocks, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(dev);
  HIPC(e);
  return OPRL_OK;
}

bool actor_due(const oprl_learner* h) {
  return h->cfg.algo != OPRL_TD3 || (h->update_count % h->cfg.hp.policy_freq == 0);
}

int check_batch(const oprl_learner* h, const void* s, const void* a, const void* r, const void* d,
                const void* s2, int B) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  if (!s || !a || !r || !d || !s2) { set_err("update: null batch pointer"); return OPRL_ERR_INVALID; }
  if (B < 1 || B > h->Bmax) { set_err("update: batch %d outside [1, max_batch=%d]", B, h->Bmax); return OPRL_ERR_INVALID; }
  return OPRL_OK;
}

// Has a kernel of this learner reported an expired wait?  (Sticky until oprl_learner_clear_error.)
int check_device_error(const oprl_learner* h) {
  if (h == nullptr || h->err_host == nullptr) return OPRL_OK;
  const unsigned code = *(volatile const unsigned*)h->err_host;
  if (code == 0) return OPRL_OK;
  static const char* kern[] = {"?", "k_ddpg_phase1", "k_ddpg_phase2", "k_mlp_slice_tp", "k_dw_adam<exchange>", "peer-window all-reduce", "k_lw_mid_pair"};
  static const char* site[] = {"?", "cluster all-reduce (a member of a 4-CU slice cluster never published its partial)",
                               "TD-target hand-off (role B never received y from role A)",
                               "twin-target exchange between role A and the role-C cluster",
                               "SAC phase 2 pair exchange (critic 2's cluster never delivered)",
                               "gradient tile exchange with another rank", "peer-window flag of another rank",
                               "gate of the dW tiles riding on a phase launch (a role never flagged its rows / seeds)",
                               "hand-over between the two hidden layers of one launch (a first-layer workgroup never flagged its rows)"};
  const unsigned k = (code >> 8) & 0xff, w = code & 0xff;
  if (w == 9) {     // SITE_X2_RANGE (csrc/tp3.h): not a wait
    set_err("device error 0x%x: an activation left the range of the split-fp16 mode (precision='x2': |x| < 4094 for observations and "
            "hidden activations, |w| < 256 for weights) in %s; the results of that update (and everything after it) are poisoned.  "
            "Normalise the observations or create the learner with precision='f32' (exact fp32, no such range).  Restore a "
            "checkpoint, then oprl_learner_clear_error().", code, k < 7 ? kern[k] : "?");
    return OPRL_ERR_STATE;
  }
  set_err("device error 0x%x: a bounded cross-workgroup wait expired in %s at the %s; the results of that "
          "update (and everything after it) are poisoned with NaN.  Typical causes: the launch's workgroups were "
          "not co-resident (another process or learner held the GPU's compute units for longer than the wait bound), "
          "or a data-parallel peer died.  Restore a checkpoint, then oprl_learner_clear_error() — after which this learner "
          "runs the launch forms for a shared GPU (set_cluster(4): no workgroup waits for another role of its own launch).",
          code, k < 7 ? kern[k] : "?", w < 9 ? site[w] : "?");
  return OPRL_ERR_STATE;
}

}  // namespace

// =========================================================================== C-ABI
extern "C" const char* oprl_last_error(void) { return g_err.c_str(); }

extern "C" int oprl_learner_check(oprl_learner* h) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  return check_device_error(h);
}

extern "C" int oprl_learner_clear_error(oprl_learner* h) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  if (h->err_host) {
    // A bounded wait that expired inside a launch whose workgroups wait for each other ACROSS roles (the whole-update
    // and merged launches: tiles for roles, the critic pass for tiles) means those workgroups were not running side by
    // side — the chip is shared (another process, an eval / actor process on the same GPU, a partitioned device).  The
    // learner then goes on, after the caller's restore, with the forms that cross a kernel boundary instead:
    // set_cluster(< 8) semantics, as for learners that share a GPU by design.
    const unsigned code = *(volatile unsigned*)h->err_host, w = code & 0xff;
    if (code != 0 && w != 9 /* SITE_X2_RANGE */ && w != 5 && w != 6 /* data-parallel peers */ && !h->debug_expire && !h->shared_chip) {
      h->shared_chip = true;
      h->no_whole = 1;
      h->no_wide = 1;
    }
    *(volatile unsigned*)h->err_host = 0;
  }
  return OPRL_OK;
}

extern "C" int oprl_debug_noise(oprl_learner* h, int32_t stream_id, uint64_t counter, int32_t rows, int32_t cols,
                                float* out_dev, void* stream) {
  if (!h || stream_id < 1 || stream_id > 2 || rows < 1 || cols < 1 || !out_dev) { set_err("oprl_debug_noise: invalid argument"); return OPRL_ERR_INVALID; }
  HIPC(launch_debug_normal(noise_key(h, (uint64_t)stream_id), counter, rows, cols, out_dev, (hipStream_t)stream));
  return OPRL_OK;
}

extern "C" int oprl_learner_debug_expire(oprl_learner* h, int32_t site) {
  if (!h || site < 0 || site > 8) { set_err("oprl_learner_debug_expire: invalid argument"); return OPRL_ERR_INVALID; }
  h->debug_expire = site;
  h->lw_pairs.spin = site == 8 ? 0 : (1 << 20);      // (8: the hand-over inside k_lw_mid_pair)
  return OPRL_OK;
}


#define NCCLC(h, x)                                                                   \
  do {                                                                                 \
    int _r = (x);                                                                      \
    if (_r != 0) {                                                                     \
      set_err("%s failed: %s", #x, (h)->rccl.err_str ? (h)->rccl.err_str(_r) : "nccl error"); \
      return OPRL_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)

extern "C" int oprl_comm_unique_id(const char* rccl_path, char id_out[OPRL_COMM_ID_BYTES]) {
  if (!id_out) { set_err("oprl_comm_unique_id: null output"); return OPRL_ERR_INVALID; }
  static Rccl r;
  RC(rccl_bind(r, rccl_path));
  NcclId id;
  const int rc = r.get_unique_id(&id);
  if (rc != 0) { set_err("ncclGetUniqueId failed (%d)", rc); return OPRL_ERR_HIP; }
  memcpy(id_out, id.internal, OPRL_COMM_ID_BYTES);
  return OPRL_OK;
}

namespace {
// the gradient arenas must be contiguous per group (critics back to back)
int dp_arena_sizes(oprl_learner* h) {
  long off = 0;
  for (int j = 0; j < h->nc; ++j) {
    if (h->cfg.critics[j].grad != h->cfg.critics[0].grad + off) { set_err("critic gradient arenas are not contiguous"); return OPRL_ERR_INVALID; }
    off += net_param_count(h->cfg.critics[j]);
  }
  h->n_critic_params = off;
  h->n_actor_params = net_param_count(h->cfg.actor);
  return OPRL_OK;
}
}  // namespace

extern "C" int oprl_comm_init(oprl_learner* h, const char* rccl_path, int32_t rank, int32_t world,
                              const char id[OPRL_COMM_ID_BYTES]) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) { set_err("oprl_comm_init: invalid argument"); return OPRL_ERR_INVALID; }
  if (!h->cfg.export_grads) { set_err("oprl_comm_init: learner was not created with export_grads"); return OPRL_ERR_STATE; }
  RC(dp_arena_sizes(h));
  RC(rccl_bind(h->rccl, rccl_path));
  NcclId nid;
  memcpy(nid.internal, id, OPRL_COMM_ID_BYTES);
  NCCLC(h, h->rccl.comm_init_rank(&h->rccl.comm, world, nid, rank));
  h->rccl.rank = rank;
  h->rccl.world = world;
  h->noise_rank = rank;          // every rank draws its own in-update noise
  return OPRL_OK;
}

// Every replica identical to rank `root`: parameters, targets, Adam moments (and the temperature with its
// moments) of all nets by ncclBroadcast, then the derived packs rebuilt.  Done once after oprl_comm_init
// ("parameters, targets and Adam state replicated, broadcast from rank 0 once").
extern "C" int oprl_comm_broadcast_params(oprl_learner* h, int32_t root, void* stream) {
  if (!h || !h->rccl.comm) { set_err("oprl_comm_broadcast_params: call oprl_comm_init first"); return OPRL_ERR_STATE; }
  if (!h->rccl.broadcast) { set_err("the RCCL library does not export ncclBroadcast"); return OPRL_ERR_INVALID; }
  if (root < 0 || root >= h->rccl.world) { set_err("oprl_comm_broadcast_params: bad root %d", root); return OPRL_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const oprl_learner_config& c = h->cfg;
  auto bc_net = [&](const oprl_net& n) -> int {
    const size_t cnt = (size_t)net_param_count(n);
    float* arenas[4] = {n.theta, n.theta_target, n.adam_m, n.adam_v};
    for (float* a : arenas)
      if (a != nullptr) NCCLC(h, h->rccl.broadcast(a, a, cnt, kNcclFloat32, root, h->rccl.comm, st));
    return OPRL_OK;
  };
  RC(bc_net(c.actor));
  for (int j = 0; j < h->nc; ++j) RC(bc_net(c.critics[j]));
  double* scalars[3] = {c.log_alpha, c.log_alpha_m, c.log_alpha_v};
  for (double* p : scalars)
    if (p != nullptr) NCCLC(h, h->rccl.broadcast(p, p, 1, kNcclFloat64, root, h->rccl.comm, st));
  return oprl_learner_sync_params(h, stream);
}

// ---- one-shot all-reduce over peer windows (csrc/p2p.hip) -------------------------------------------
extern "C" int oprl_p2p_create(oprl_learner* h, int32_t rank, int32_t world, char handle_out[OPRL_P2P_HANDLE_BYTES]) {
  if (!h || !handle_out) { set_err("oprl_p2p_create: invalid argument"); return OPRL_ERR_INVALID; }
  if (!h->cfg.export_grads) { set_err("oprl_p2p_create: learner was not created with export_grads"); return OPRL_ERR_STATE; }
  if (h->p2p.window != nullptr) { set_err("oprl_p2p_create: window already exists"); return OPRL_ERR_STATE; }
  RC(dp_arena_sizes(h));
  const size_t n = (size_t)std::max(h->n_critic_params, h->n_actor_params);
  // second region: the per-tile exchange of k_dw_adam<true> (fused learners; a few MB)
  h->p2p_max_tiles = std::max(h->tiles_critic, h->tiles_actor);
  size_t tile_bytes = h->fused ? dw_xchg_bytes(world, h->p2p_max_tiles) : 0;
  if (tile_bytes > ((size_t)256 << 20)) tile_bytes = 0;
  h->noise_rank = rank;
  h->p2p.err = h->err_dev;
  hipError_t e = p2p_create(h->p2p, rank, world, n, tile_bytes, handle_out);
  if (e != hipSuccess) {
    set_err("oprl_p2p_create: %s", hipGetErrorString(e));
    (void)hipGetLastError();
    p2p_destroy(h->p2p);
    return OPRL_ERR_HIP;
  }
  return OPRL_OK;
}

extern "C" int oprl_p2p_connect(oprl_learner* h, const char* handles) {
  if (!h || !handles || h->p2p.window == nullptr) { set_err("oprl_p2p_connect: call oprl_p2p_create first"); return OPRL_ERR_STATE; }
  hipError_t e = p2p_connect(h->p2p, handles);
  if (e != hipSuccess) { set_err("oprl_p2p_connect: %s", hipGetErrorString(e)); (void)hipGetLastError(); return OPRL_ERR_HIP; }
  return OPRL_OK;
}

// Every rank contributes (rank + 1) * (1 + i mod 7) at element i of its critic gradient arena; the
// windows are kept only if this rank's sum is exact everywhere.  (The ranks decide together: the
// host reduces the verdicts, oprl_amd/parallel.py.)
extern "C" int oprl_p2p_selftest(oprl_learner* h, void* stream) {
  if (!h || !h->p2p.connected) { set_err("oprl_p2p_selftest: windows are not connected"); return OPRL_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)h->n_critic_params;
  float* g = h->cfg.critics[0].grad;
  std::vector<float> host(n);
  bool all_ok = true;
  for (int round = 0; round < 3 && all_ok; ++round) {   // three rounds: both window halves and a reuse
    for (size_t i = 0; i < n; ++i) host[i] = (float)((h->p2p.rank + 1) * (1 + (int)((i + round) % 7)));
    HIPC(hipMemcpyAsync(g, host.data(), n * sizeof(float), hipMemcpyHostToDevice, st));
    HIPC(p2p_all_reduce(h->p2p, g, n, false, st));
    HIPC(hipMemcpyAsync(host.data(), g, n * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    const int tri = h->p2p.world * (h->p2p.world + 1) / 2;
    for (size_t i = 0; i < n && all_ok; ++i) all_ok = host[i] == (float)(tri * (1 + (int)((i + round) % 7)));
  }
  HIPC(hipMemsetAsync(g, 0, n * sizeof(float), st));
  if (const char* f = getenv("OPRL_AMD_P2P_SELFTEST_FAIL")) {   // tests: exercise the fall-back to RCCL
    if (atoi(f) != 0) all_ok = false;
  }
  h->p2p_tested = all_ok;
  if (!all_ok) { set_err("oprl_p2p_selftest: the exchanged sum is wrong; staying on RCCL"); return OPRL_ERR_STATE; }
  return OPRL_OK;
}

// The ranks agree on the host (every self-test passed) and then switch together.
extern "C" int oprl_p2p_enable(oprl_learner* h, int32_t on) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  if (on && !h->p2p_tested) { set_err("oprl_p2p_enable: the self-test has not passed on this rank"); return OPRL_ERR_STATE; }
  if (on < 0 || on > 2) { set_err("oprl_p2p_enable: level must be 0, 1 or 2"); return OPRL_ERR_INVALID; }
  h->p2p_ok = on != 0;
  h->p2p_inline = on == 2;     // 2: fused learners also exchange inside their dW launches (k_dw_adam<true>)
  return OPRL_OK;
}

namespace {
int dp_world(const oprl_learner* h) { return h->p2p_ok ? h->p2p.world : h->rccl.world; }
int dp_rank(const oprl_learner* h) { return h->p2p_ok ? h->p2p.rank : h->rccl.rank; }
// in-place sum over ranks of a float (or one-double) buffer: peer windows when they passed the self-test, else RCCL
int dp_all_reduce(oprl_learner* h, void* buf, size_t n, bool as_double, hipStream_t st) {
  // The one-shot exchange sends the whole arena to every peer: right for the latency-bound ~300 KB
  // arenas of the 256-wide nets, wrong for TQC's 11 MB critic arena, where a ring moves 2 x 7/8 of the
  // bytes instead of 7 x — those stay on RCCL when a communicator exists.
  const bool small = n * (as_double ? 8 : 4) <= ((size_t)1 << 20);
  if (h->p2p_ok && (small || !h->rccl.comm)) {
    HIPC(p2p_all_reduce(h->p2p, buf, n, as_double, st));
    return OPRL_OK;
  }
  NCCLC(h, h->rccl.all_reduce(buf, buf, n, as_double ? kNcclFloat64 : kNcclFloat32, kNcclSum, h->rccl.comm, st));
  return OPRL_OK;
}
}  // namespace

namespace {
int chain_loop(oprl_learner* h, int K, int B, float* (*set)[5], void* stream);
bool chain_ok(oprl_learner* h, int B);
}  // namespace

extern "C" int oprl_learner_dp_update(oprl_learner* h, const float* s, const float* a, const float* r,
                                      const float* d, const float* s2, int32_t B, const float* noise0,
                                      const float* noise1, void* stream) {
  if (!h || (!h->rccl.comm && !h->p2p_ok)) { set_err("oprl_learner_dp_update: call oprl_comm_init (or connect the peer windows) first"); return OPRL_ERR_STATE; }
  hipStream_t st = (hipStream_t)stream;
  const oprl_learner_config& c = h->cfg;
  const double scale = 1.0 / (double)dp_world(h);
  // Fused learners on peer windows: the two dW launches exchange their own tiles (k_dw_adam<true>) and
  // run Adam on the mean — no separate all-reduce or apply launches.
  if (h->p2p_ok && h->p2p_inline && !h->no_dp_inline && h->p2p.tile_bytes > 0 && use_fused(h, B)) {
    h->dp_inline = true;
    int rc = oprl_learner_update_phase(h, 0, s, a, r, d, s2, B, noise0, noise1, stream);
    if (rc == OPRL_OK) rc = oprl_learner_update_phase(h, 1, s, a, r, d, s2, B, noise0, noise1, stream);
    h->dp_inline = false;
    RC(rc);
    if (h->actor_updated_last && alpha_ptr(h) != nullptr) {   // the temperature: one double, exchanged on its own
      RC(dp_all_reduce(h, h->alpha_grad, 1, true, st));
      HIPC(launch_alpha_step(c.log_alpha, c.log_alpha_m, c.log_alpha_v, nullptr, 1, (float)c.hp.target_entropy,
                             c.hp.lr_alpha, c.hp.beta1, c.hp.beta2, c.hp.adam_eps, h->opt_step_alpha,
                             nullptr, h->alpha_grad, (float)scale, st));
    }
    return OPRL_OK;
  }
  RC(oprl_learner_update_phase(h, 0, s, a, r, d, s2, B, noise0, noise1, stream));
  RC(dp_all_reduce(h, c.critics[0].grad, (size_t)h->n_critic_params, false, st));
  RC(oprl_learner_apply(h, 0, scale, stream));
  RC(oprl_learner_update_phase(h, 1, s, a, r, d, s2, B, noise0, noise1, stream));
  if (h->actor_updated_last) {
    RC(dp_all_reduce(h, c.actor.grad, (size_t)h->n_actor_params, false, st));
    if (alpha_ptr(h) != nullptr) RC(dp_all_reduce(h, h->alpha_grad, 1, true, st));
    RC(oprl_learner_apply(h, 1, scale, stream));
  }
  return OPRL_OK;
}

extern "C" int oprl_learner_dp_step_n(oprl_learner* h, oprl_replay* replay, int32_t K, int32_t B,
                                      uint64_t seed, void* stream) {
  if (!h || !replay) { set_err("oprl_learner_dp_step_n: null handle"); return OPRL_ERR_INVALID; }
  if (!h->rccl.comm && !h->p2p_ok) { set_err("oprl_learner_dp_step_n: call oprl_comm_init (or connect the peer windows) first"); return OPRL_ERR_STATE; }
  int S = 0, A = 0;
  replay_dims(replay, &S, &A);
  if (S != h->S || A != h->A) { set_err("replay dims (%d,%d) != learner dims (%d,%d)", S, A, h->S, h->A); return OPRL_ERR_INVALID; }
  if (K < 0 || B < 1 || B > h->Bmax) { set_err("dp_step_n: bad K/B"); return OPRL_ERR_INVALID; }
  // every rank samples its own shard: the Philox key mixes the rank in
  const uint64_t rseed = seed * 0x9E3779B97F4A7C15ull + (uint64_t)dp_rank(h);
  if (use_fused(h, B)) {
    BatchSrc& sc = h->src;
    RC(oprl_replay_flush(replay, stream));
    long n_tr = 0;
    replay_view(replay, &sc.states, &sc.actions, &sc.rewards, &sc.dones, &sc.ends, &sc.n_eps, &sc.L, &n_tr);
    if (n_tr <= 0 || sc.n_eps <= 0) { set_err("dp_step_n: replay buffer is empty"); return OPRL_ERR_STATE; }
    sc.n_transitions = n_tr;
    sc.seed = rseed;
    sc.gather = 1;
    // as in oprl_learner_step_n: phase 2 of every update gathers the next update's rows
    h->next_src = sc;
    // The gradient exchange inside the tiles of the whole-update launch (peer windows, PrecX2 learners): the data-parallel
    // K-loop IS the single-GPU one — k_ddpg_chain, up to chain_max updates per launch, every tile all-reducing its
    // gradient with the other ranks' before Adam.  No all-reduce launches, no apply launches.
    if (h->p2p_ok && h->p2p_inline && !h->no_dp_inline && h->p2p.tile_bytes > 0) {
      h->dp_inline = true;
      if (chain_ok(h, B)) {
        const size_t Bm = (size_t)h->Bmax;
        float* alt = h->batch_alt;
        float* set[2][5] = {{h->bs, h->ba, h->br, h->bd, h->bs2},
                            {alt, alt + Bm * h->S, alt + Bm * (h->S + h->A), alt + Bm * (h->S + h->A + 1), alt + Bm * (h->S + h->A + 2)}};
        const int rc_chain = chain_loop(h, K, B, set, stream);
        h->dp_inline = false;
        return rc_chain;
      }
      h->dp_inline = false;
    }
    h->next_src.s = h->bs; h->next_src.a = h->ba; h->next_src.r = h->br; h->next_src.d = h->bd;
    h->next_src.s2 = h->bs2;
    int rc = OPRL_OK;
    for (int k = 0; k < K && rc == OPRL_OK; ++k) {
      sc.counter = (unsigned long long)h->update_count;
      h->next_src.counter = sc.counter + 1;
      h->prefetch_next = (k + 1 < K) ? 1 : 0;
      sc.gather = h->staged_ready ? 0 : 1;
      h->staged_ready = false;
      rc = oprl_learner_dp_update(h, h->bs, h->ba, h->br, h->bd, h->bs2, B, nullptr, nullptr, stream);
    }
    sc.gather = 0;
    h->prefetch_next = 0;
    h->staged_ready = false;
    return rc;
  }
  for (int k = 0; k < K; ++k) {
    RC(oprl_replay_sample(replay, B, nullptr, rseed, (uint64_t)h->update_count, h->bs, h->ba, h->br,
                          h->bd, h->bs2, nullptr, nullptr, stream));
    RC(oprl_learner_dp_update(h, h->bs, h->ba, h->br, h->bd, h->bs2, B, nullptr, nullptr, stream));
  }
  return OPRL_OK;
}

extern "C" int oprl_profile_enable(int32_t on) {
  if (!on) prof_fold();
  g_prof.on = on != 0;
  return OPRL_OK;
}

extern "C" int oprl_profile_read(int64_t* counts_host, double* ms_host, int32_t reset) {
  if (!counts_host || !ms_host) { set_err("oprl_profile_read: null argument"); return OPRL_ERR_INVALID; }
  prof_fold();
  for (int k = 0; k < OPRL_PROFILE_KINDS; ++k) {
    counts_host[k] = g_prof.counts[k];
    ms_host[k] = g_prof.ms[k];
    if (reset) { g_prof.counts[k] = 0; g_prof.ms[k] = 0; }
  }
  return OPRL_OK;
}
extern "C" int oprl_abi_version(void) { return OPRL_ABI_VERSION; }

extern "C" int oprl_learner_create(const oprl_learner_config* cfg, oprl_learner** out) {
  if (!cfg || !out) { set_err("oprl_learner_create: null argument"); return OPRL_ERR_INVALID; }
  if (cfg->abi_version != OPRL_ABI_VERSION) { set_err("ABI version mismatch: caller %d, library %d", cfg->abi_version, OPRL_ABI_VERSION); return OPRL_ERR_INVALID; }
  if (cfg->algo < OPRL_DDPG || cfg->algo > OPRL_TQC) { set_err("unknown algo %d", cfg->algo); return OPRL_ERR_INVALID; }
  if (cfg->precision != OPRL_PREC_F32 && cfg->precision != OPRL_PREC_BF16 && cfg->precision != OPRL_PREC_X2) { set_err("precision %d unknown", cfg->precision); return OPRL_ERR_INVALID; }
  const int nc_expect = cfg->algo == OPRL_DDPG ? 1 : (cfg->algo == OPRL_TQC ? cfg->n_critics : 2);
  if (cfg->n_critics != nc_expect || cfg->n_critics < 1 || cfg->n_critics > OPRL_MAX_CRITICS) {
    set_err("n_critics=%d invalid for algo %d", cfg->n_critics, cfg->algo);
    return OPRL_ERR_INVALID;
  }
  if (cfg->max_batch < 1 || cfg->state_dim < 1 || cfg->action_dim < 1) { set_err("bad dims"); return OPRL_ERR_INVALID; }
  auto* h = new oprl_learner();
  h->cfg = *cfg;
  h->S = cfg->state_dim; h->A = cfg->action_dim; h->Bmax = cfg->max_batch; h->nc = cfg->n_critics;
  h->bf16 = cfg->precision == OPRL_PREC_BF16;
  h->x2 = cfg->precision == OPRL_PREC_X2;
  h->planes = h->x2 ? 2 : 1;
  int rc = check_net(cfg->actor, "actor", &h->w_actor);
  for (int j = 0; rc == OPRL_OK && j < h->nc; ++j) {
    int w = 0;
    rc = check_net(cfg->critics[j], "critic", &w);
    if (rc == OPRL_OK && j > 0 && w != h->w_critic) { set_err("critics differ in width"); rc = OPRL_ERR_INVALID; }
    h->w_critic = w;
    if (rc == OPRL_OK && cfg->critics[j].dims[0] != h->S + h->A) { set_err("critic input dim != S+A"); rc = OPRL_ERR_INVALID; }
    if (rc == OPRL_OK && (!cfg->critics[j].theta_target || !cfg->critics[j].adam_m || !cfg->critics[j].adam_v)) {
      set_err("critic %d: theta_target/adam_m/adam_v required", j); rc = OPRL_ERR_INVALID;
    }
  }
  const bool gauss = cfg->algo == OPRL_SAC || cfg->algo == OPRL_TQC;
  if (rc == OPRL_OK && cfg->actor.dims[0] != h->S) { set_err("actor input dim != S"); rc = OPRL_ERR_INVALID; }
  if (rc == OPRL_OK && cfg->actor.dims[cfg->actor.n_layers] != (gauss ? 2 : 1) * h->A) { set_err("actor output dim mismatch"); rc = OPRL_ERR_INVALID; }
  if (rc == OPRL_OK && (!cfg->actor.adam_m || !cfg->actor.adam_v)) { set_err("actor adam state required"); rc = OPRL_ERR_INVALID; }
  if (rc == OPRL_OK && !gauss && !cfg->actor.theta_target) { set_err("actor target required for DDPG/TD3"); rc = OPRL_ERR_INVALID; }
  if (rc == OPRL_OK && cfg->algo == OPRL_TQC) {
    const int Q = cfg->hp.n_quantiles;
    if (Q < 1 || Q > kNarrowMax || h->nc * Q > 128 || cfg->hp.top_quantiles_to_drop < 0 ||
        cfg->hp.top_quantiles_to_drop >= h->nc * Q || cfg->critics[0].dims[cfg->critics[0].n_layers] != Q) {
      set_err("TQC quantile configuration unsupported"); rc = OPRL_ERR_INVALID;
    }
  }
  const bool learned_alpha = cfg->algo == OPRL_TQC || (cfg->algo == OPRL_SAC && cfg->hp.tune_alpha);
  if (rc == OPRL_OK && learned_alpha && (!cfg->log_alpha || !cfg->log_alpha_m || !cfg->log_alpha_v)) {
    set_err("log_alpha and its Adam state are required"); rc = OPRL_ERR_INVALID;
  }
  if (rc == OPRL_OK && cfg->export_grads && learned_alpha && !cfg->log_alpha_grad) {
    set_err("export_grads with a learned temperature needs log_alpha_grad"); rc = OPRL_ERR_INVALID;
  }
  if (rc == OPRL_OK && cfg->export_grads) {
    if (!cfg->actor.grad) { set_err("export_grads needs grad arenas"); rc = OPRL_ERR_INVALID; }
    for (int j = 0; j < h->nc; ++j) if (!cfg->critics[j].grad) { set_err("export_grads needs grad arenas"); rc = OPRL_ERR_INVALID; }
  }
  if (rc != OPRL_OK) { delete h; return rc; }

  hipError_t e = init_kernel_attrs();
  if (e == hipSuccess) e = init_fused_attrs();
  if (e == hipSuccess) e = init_slice_tp_attrs();
  if (e == hipSuccess) e = init_layerwise_attrs();
  if (e != hipSuccess) { set_err("hipFuncSetAttribute: %s", hipGetErrorString(e)); delete h; return OPRL_ERR_HIP; }
  memset(&h->src, 0, sizeof h->src);
  memset(&h->next_src, 0, sizeof h->next_src);
  if (h->nc > 2) {
    bool ok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int j = 1; ok && j < h->nc; ++j)
      ok = hipStreamCreateWithFlags(&h->side[j], hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&h->ev_join[j], hipEventDisableTiming) == hipSuccess;
    h->have_side = ok;
  }
  h->fused = (cfg->algo == OPRL_DDPG || cfg->algo == OPRL_TD3 || (cfg->algo == OPRL_SAC && h->nc == 2)) &&
             !cfg->no_fuse && h->w_actor == 256 && h->w_critic == 256 && cfg->actor.n_layers == 3 &&
             cfg->critics[0].n_layers == 3;

  const int B = h->Bmax, S = h->S, A = h->A, nc = h->nc;
  // scalar critics: q' is read with stride 1 by the TD seed; TQC: [B][ldq] quantile rows
  h->ldq = cfg->algo == OPRL_TQC ? round_up(cfg->critics[0].dims[cfg->critics[0].n_layers], 4) : 1;
  const int n_slices = (B + kR - 1) / kR;
  size_t floats = net_ws_floats(cfg->actor, B);
  for (int j = 0; j < nc; ++j) floats += net_ws_floats(cfg->critics[j], B);
  floats += (size_t)B * A + B + (size_t)nc * B * h->ldq + (size_t)B * A + (size_t)B * 2 * A + B +
            (size_t)nc * B * A + (size_t)nc * B + (size_t)B * 128 + 2 * (size_t)B;
  floats += (size_t)(nc + 1) * n_slices * 4 + 16;
  floats += (size_t)B * (2 * S + A + 2);
  floats += 64 * 32 + 6 * (size_t)B + 512;      // (granule arrays: y, q1, q2; 256 gate flags)
  const int Bm = B < 256 ? B : 256;             // merged phase 2 serves one 256-row chunk
  const bool merge2_bufs = h->fused && cfg->algo != OPRL_SAC && A <= kDuLd;
  if (merge2_bufs) floats += 2 * ((size_t)Bm * kDuLd + 64) + 2 * 256 + 128 + 2 * (size_t)16 * Bm * 16 + 64 + 16 * 256 + 4 * 64 + 2 * 256 + 64 + kMaxLayers * 256 + 64;
  if (merge2_bufs) floats += 2 * (192 + 192 + 64 + 128 + 1024) + 64 + 4 * kMaxLayers * 256 + 64 + 16 * 256 + 64 + (size_t)kDuLd * Bm * 256 + 64;      // (k_ddpg_chain)
  if (h->bf16 || h->x2) {
    floats += 2 * ((size_t)net_pack16_floats(cfg->actor, h->planes) + 64);
    for (int j = 0; j < nc; ++j) floats += 2 * ((size_t)net_pack16_floats(cfg->critics[j], h->planes) + 64);
  }
  const size_t bytes = floats * sizeof(float) + 8192 + sizeof(DwItem) * (size_t)(nc + 1) * kMaxLayers +
                       sizeof(RepackItem) * (size_t)(4 * nc + 4) * kMaxLayers;
  // PrecX2 learners: the workspace — activation exchange buffers, granules, staged rows — in UNCACHED device memory
  // (measured: no slower than cached, r03 log), so that a role of the whole-update launch reads what an earlier role
  // of the same launch wrote
  h->fchain = h->fused && !h->x2 && !h->bf16 && cfg->algo == OPRL_DDPG && nc == 1 && !cfg->export_grads && merge2_bufs &&
              cfg->actor.theta_target != nullptr && cfg->critics[0].theta_target != nullptr && cfg->actor.n_layers == 3 && cfg->critics[0].n_layers == 3;
  const int uc_pool = (h->x2 || h->fchain) ? 1 : 0;
  h->uc_pool = uc_pool != 0;
  if ((uc_pool ? uc_alloc((void**)&h->pool.base, bytes) : hipMalloc(&h->pool.base, bytes)) != hipSuccess) { set_err("hipMalloc(%zu) failed", bytes); delete h; return OPRL_ERR_NOMEM; }
  h->pool.cap = bytes;
  (void)hipMemset(h->pool.base, 0, bytes);
  {
    // the error word: host memory the device can write (only ever on the error path)
    void* eh = nullptr; void* ed = nullptr;
    if (hipHostMalloc(&eh, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&ed, eh, 0) != hipSuccess) {
      set_err("hipHostMalloc(error word) failed");
      if (eh) (void)hipHostFree(eh);
      dev_free(h->pool.base); delete h; return OPRL_ERR_NOMEM;
    }
    memset(eh, 0, 64);
    h->err_host = (unsigned*)eh;
    h->err_dev = (unsigned*)ed;
  }
  Pool& p = h->pool;
  alloc_net_ws(p, cfg->actor, B, &h->ws_actor);
  for (int j = 0; j < nc; ++j) alloc_net_ws(p, cfg->critics[j], B, &h->ws_critic[j]);
  h->a2 = p.take<float>((size_t)B * A);
  h->logp2 = p.take<float>(B);
  h->qn = p.take<float>((size_t)nc * B * h->ldq);
  h->pi = p.take<float>((size_t)B * A);
  h->raw = p.take<float>((size_t)B * 2 * A);
  h->logp = p.take<float>(B);
  h->da = p.take<float>((size_t)nc * B * A);
  h->qpi = p.take<float>((size_t)nc * B);
  h->target = p.take<float>((size_t)B * 128);
  h->ydbg = p.take<float>(B);
  h->qdbg = p.take<float>(B);
  h->part_c = p.take<float>((size_t)nc * n_slices * 4);
  h->part_a = p.take<float>((size_t)n_slices * 4);
  h->scalars = p.take<float>(16);
  h->alpha_grad = cfg->log_alpha_grad ? cfg->log_alpha_grad : p.take<double>(2);
  h->y_granules = p.take<unsigned long long>((size_t)3 * B + 256);
  if (merge2_bufs) {
    h->du_granules = p.take<unsigned long long>((size_t)Bm * kDuLd);
    h->g1_granules = p.take<unsigned long long>((size_t)16 * Bm * 16);
    h->w3_snap = p.take<float>(16 * 256);
    h->w_flags = p.take<unsigned long long>(256);
    h->chain_flags = p.take<unsigned long long>(192 + 192 + 64 + 128 + 1024);
    h->gu = p.take<float>((size_t)kDuLd * Bm * 256);
    h->chain_b16 = p.take<float>(4 * kMaxLayers * 256);
    h->critic_b16 = h->chain_b16 + 2 * kMaxLayers * 256;
    h->w3buf1 = p.take<float>(16 * 256);
  }
  h->bs = p.take<float>((size_t)B * S);
  h->ba = p.take<float>((size_t)B * A);
  h->br = p.take<float>(B);
  h->bd = p.take<float>(B);
  h->bs2 = p.take<float>((size_t)B * S);
  // the two-plane packs of a PrecX2 learner in UNCACHED device memory — every load and store goes to the fabric, so
  // that a workgroup reads what a workgroup on another XCD has just written without a kernel boundary in between
  // (measured: no slower than cached, r03 log)
  if (h->x2) {
    size_t fl = 2 * ((size_t)net_pack16_floats(cfg->actor, 2) + 64);
    for (int j = 0; j < nc; ++j) fl += 2 * ((size_t)net_pack16_floats(cfg->critics[j], 2) + 64);
    float* base = nullptr;
    if (uc_alloc((void**)&base, fl * sizeof(float)) != hipSuccess) {
      set_err("hipExtMallocWithFlags(uncached packs) failed"); dev_free(p.base); delete h; return OPRL_ERR_NOMEM;
    }
    (void)hipMemset(base, 0, fl * sizeof(float));
    h->uc_base = base;
    auto take = [&](size_t n) { float* q = base; base += (n + 63) & ~(size_t)63; return q; };
    h->pack16[0] = take((size_t)net_pack16_floats(cfg->actor, 2));
    h->pack16_t[0] = take((size_t)net_pack16_floats(cfg->actor, 2));
    for (int j = 0; j < nc; ++j) {
      h->pack16[1 + j] = take((size_t)net_pack16_floats(cfg->critics[j], 2));
      h->pack16_t[1 + j] = take((size_t)net_pack16_floats(cfg->critics[j], 2));
    }
  } else if (h->fchain) {
    // (the mirrors of the fp32 fragment packs: see oprl_learner::fchain)
    const oprl_net* src[2] = {&h->cfg.actor, &h->cfg.critics[0]};
    size_t fl = 0;
    for (int k = 0; k < 2; ++k) fl += 2 * (((size_t)oprl_net_pack_floats(src[k]) + 63) & ~(size_t)63);
    float* base = nullptr;
    if (uc_alloc((void**)&base, fl * sizeof(float)) != hipSuccess) {
      set_err("hipExtMallocWithFlags(uncached packs) failed"); dev_free(p.base); delete h; return OPRL_ERR_NOMEM;
    }
    (void)hipMemset(base, 0, fl * sizeof(float));
    h->uc_base = base;
    for (int k = 0; k < 2; ++k) {
      const size_t n = ((size_t)oprl_net_pack_floats(src[k]) + 63) & ~(size_t)63;
      h->fnet[k] = *src[k];
      h->fnet[k].pack = base; base += n;
      h->fnet[k].pack_target = base; base += n;
    }
  } else
  if (h->bf16 || h->x2) {   // (the pool is zeroed: pad positions of the packs stay zero for good)
    h->pack16[0] = p.take<float>((size_t)net_pack16_floats(cfg->actor, h->planes));
    h->pack16_t[0] = p.take<float>((size_t)net_pack16_floats(cfg->actor, h->planes));
    for (int j = 0; j < nc; ++j) {
      h->pack16[1 + j] = p.take<float>((size_t)net_pack16_floats(cfg->critics[j], h->planes));
      h->pack16_t[1 + j] = p.take<float>((size_t)net_pack16_floats(cfg->critics[j], h->planes));
    }
  }
  std::vector<DwItem> items;
  for (int j = 0; j < nc; ++j)
    fill_items(eff(h, h->cfg.critics[j]), h->ws_critic[j], items, &h->tiles_critic, h->fused, h->pack16[1 + j], h->pack16_t[1 + j], h->planes);
  h->n_items_critic = (int)items.size();
  if (h->critic_b16 != nullptr && nc == 1)
    for (int l = 0; l < h->n_items_critic; ++l) {
      items[l].b16 = h->critic_b16 + 256 * l;
      items[l].bt16 = h->chain_b16 + (3 * kMaxLayers + l) * 256;
    }
  fill_items(eff(h, h->cfg.actor), h->ws_actor, items, &h->tiles_actor, h->fused, h->pack16[0], h->pack16_t[0], h->planes);
  h->n_items_actor = (int)items.size() - h->n_items_critic;
  if (h->chain_b16 != nullptr && nc == 1)
    for (int l = 0; l < h->n_items_actor; ++l) {
      items[h->n_items_critic + l].b16 = h->chain_b16 + (0 * kMaxLayers + l) * 256;
      items[h->n_items_critic + l].bt16 = h->chain_b16 + (1 * kMaxLayers + l) * 256;
    }
  h->items_host = items;
  std::vector<RepackItem> rp[3];
  {
    const oprl_net* cn[OPRL_MAX_CRITICS];
    for (int j = 0; j < nc; ++j) cn[j] = &h->cfg.critics[j];
    const oprl_net* an[1] = {&h->cfg.actor};
    build_repack_items(cn, nc, 1, rp[0], &h->rp_blocks[0]);
    build_repack_items(cn, nc, 3, rp[1], &h->rp_blocks[1]);
    build_repack_items(an, 1, 3, rp[2], &h->rp_blocks[2]);
    for (int k = 0; k < 3; ++k) {
      h->rp_n[k] = (int)rp[k].size();
      h->rp_dev[k] = p.take<RepackItem>(rp[k].size());
    }
  }
  if (p.used > p.cap) { set_err("internal: workspace pool overflow (%zu > %zu)", p.used, p.cap); dev_free(p.base); delete h; return OPRL_ERR_NOMEM; }
  for (int k = 0; k < 3; ++k)
    if (!rp[k].empty()) (void)hipMemcpy(h->rp_dev[k], rp[k].data(), sizeof(RepackItem) * rp[k].size(), hipMemcpyHostToDevice);
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      h->n_cus = prop.multiProcessorCount;
    const char* env = getenv("OPRL_AMD_CLUSTER");
    h->ncl = env ? atoi(env) : kMaxCluster;
    if (h->ncl != 1 && h->ncl != 2 && h->ncl != 4) h->ncl = kMaxCluster;
    h->no_multi = false;
    // OPRL_AMD_NO_RIDE = bit mask of the riders / joined launches to switch off (tests: each is bit-identical to the
    // separate launches): 1 TD target on the target heads, 2 actor forward on the critic heads, 4 first hidden launch
    // behind the actor's forward, 8 next rows on k_lw_dact, 16 wide dW kernel (kernels.hip), 32 hidden-layer pairs
    const int no_ride = [] { const char* e = getenv("OPRL_AMD_NO_RIDE"); return e != nullptr ? atoi(e) : 0; }();
    const char* nlw = getenv("OPRL_AMD_NO_LAYERWISE");
    h->no_layerwise = (nlw != nullptr && atoi(nlw) != 0);
    h->no_gather_ride = (no_ride & 8) != 0;
    if (cfg->algo == OPRL_TQC || h->du_granules != nullptr) {
      const size_t n = (size_t)h->Bmax * (2 * (size_t)h->S + h->A + 2);
      // (PrecX2 learners: uncached, like the first staging set in the pool — inside k_ddpg_chain an update reads rows a
      // workgroup of the update before has gathered)
      if ((h->uc_pool ? uc_alloc((void**)&h->batch_alt, n * sizeof(float)) : hipMalloc(&h->batch_alt, n * sizeof(float))) != hipSuccess)
        h->batch_alt = nullptr;   // (then: a gather launch per update)
    }
    h->no_fin_ride = (no_ride & 4) != 0;
    if (cfg->algo == OPRL_TQC && h->w_critic == 512) {
      const size_t n = (size_t)nc * (kMaxLayers - 1) * (size_t)h->Bmax * 512;
      if (hipMalloc(&h->lw_scratch, n * sizeof(float)) != hipSuccess) h->lw_scratch = nullptr;   // (then: the nets' own buffers, no early launch)
      {
        const int pair_env = (no_ride & 32) != 0 ? 0 : 3;
        const int nf = kMaxMulti * ((h->Bmax + 31) / 32) * 32;
        void* fl = nullptr;
        if (pair_env != 0 && hipMalloc(&fl, (size_t)nf * sizeof(unsigned long long)) == hipSuccess) {
          (void)hipMemset(fl, 0, (size_t)nf * sizeof(unsigned long long));
          h->lw_pairs.flags = (unsigned long long*)fl; h->lw_pairs.n_flags = nf; h->lw_pairs.use = pair_env & 3;
          h->lw_pairs.err = h->err_dev;
        }
      }
    }
    h->no_af_ride = (no_ride & 2) != 0;
    h->no_tqc_ride = (no_ride & 1) != 0;
    if (cfg->algo == OPRL_TQC && nc * cfg->hp.n_quantiles <= 128) {
      const size_t slices = (size_t)(h->Bmax + kR - 1) / kR;
      if (hipMalloc(&h->tqc_counter, slices * sizeof(unsigned long long)) != hipSuccess) h->tqc_counter = nullptr;   // (then: the stand-alone launch)
      else (void)hipMemset(h->tqc_counter, 0, slices * sizeof(unsigned long long));
    }
    const char* ndi = getenv("OPRL_AMD_NO_DP_INLINE");
    h->no_dp_inline = (ndi != nullptr && atoi(ndi) != 0);
    // OPRL_AMD_NO_SIDE_BY_SIDE: TD3 / SAC twin nets back to back instead of on clusters that wait for each other
    const char* nsb = getenv("OPRL_AMD_NO_SIDE_BY_SIDE");
    h->no_twin_split = (nsb != nullptr && atoi(nsb) != 0);
    h->no_p2_pair = h->no_twin_split;
    const char* nl = getenv("OPRL_AMD_NO_LEAN");
    h->no_lean = (nl != nullptr && atoi(nl) != 0) ? 1 : 0;
    // OPRL_AMD_FORM: the launch structure of the fused DDPG / TD3 update — "chain" (default: the whole update, several
    // per launch), "two" (merged phase launches: phase 1 + the critic's tiles | phase 2 + the actor's), "p2" (phase 1
    // merged, phase 2 and the actor's dW as launches of their own), "plain" (phases and dW launches)
    h->no_merge = h->no_merge2 = h->no_whole = 0;
    if (const char* f = getenv("OPRL_AMD_FORM")) {
      if (!strcmp(f, "two")) h->no_whole = 1;
      else if (!strcmp(f, "p2")) { h->no_whole = 1; h->no_merge2 = 1; }
      else if (!strcmp(f, "plain")) { h->no_whole = 1; h->no_merge2 = 1; h->no_merge = 1; }
    }
    h->no_chain = 0;
    if (const char* cm = getenv("OPRL_AMD_CHAIN")) { const int v = atoi(cm); if (v >= 1 && v <= kChainMax) h->chain_max = v; }
    const char* nw = getenv("OPRL_AMD_NO_WIDE");
    h->no_wide = (nw != nullptr && atoi(nw) != 0) ? 1 : 0;
    // the generic per-net launches on clusters of 4 (slice_tp.hip): any net of the common shape
    // (decided per net by tp_generic(): TQC's 512-wide critics stay on k_mlp_slice, its actor moves)
    h->tp_generic_on = !h->no_lean && h->ncl == 4;
  }
  if (h->fused || h->tp_generic_on) {
    const size_t slices = (size_t)(h->Bmax + kR - 1) / kR;
    // (areas laid out for clusters of eight where wide clusters may run: DDPG / TD3, fp32, lean passes)
    h->xnc = (h->fused && !h->bf16 && !h->no_lean && !h->no_wide && h->ncl == 4 &&
              (cfg->algo == OPRL_DDPG || cfg->algo == OPRL_TD3)) ? 8 : kMaxCluster;
    h->xbuf_granules = (size_t)(2 + nc) * slices * fused_xbuf_granules_per_cluster(h->xnc);
    if (hipMalloc(&h->xbuf, h->xbuf_granules * sizeof(unsigned long long)) != hipSuccess) {
      set_err("hipMalloc(cluster exchange area, %zu MB) failed", (h->xbuf_granules * 8) >> 20);
      dev_free(p.base); delete h; return OPRL_ERR_NOMEM;
    }
    (void)hipMemset(h->xbuf, 0, h->xbuf_granules * sizeof(unsigned long long));
  }
  {
    const oprl_net* nets[OPRL_MAX_CRITICS + 1];
    for (int j = 0; j < nc; ++j) nets[j] = &h->cfg.critics[j];
    nets[nc] = &h->cfg.actor;
    float *p16[OPRL_MAX_CRITICS + 1], *p16t[OPRL_MAX_CRITICS + 1];
    for (int j = 0; j < nc; ++j) { p16[j] = h->pack16[1 + j]; p16t[j] = h->pack16_t[1 + j]; }
    p16[nc] = h->pack16[0]; p16t[nc] = h->pack16_t[0];
    int prc = repack_nets(nets, nc + 1, 3, nullptr, (h->bf16 || h->x2) ? p16 : nullptr, (h->bf16 || h->x2) ? p16t : nullptr, h->planes);
    if (prc == OPRL_OK && h->fchain) {
      const oprl_net* fn[2] = {&h->fnet[0], &h->fnet[1]};
      prc = repack_nets(fn, 2, 3, nullptr);
    }
    if (prc != OPRL_OK) { dev_free(p.base); delete h; return prc; }
  }
  if (h->x2 || h->fchain || (h->bf16 && cfg->algo == OPRL_TQC)) {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    g_lazy.push_back(h);
    h->lazy_wide = cfg->algo == OPRL_TQC;
  }
  if (g_live.fetch_add(1) >= 1) (void)hipDeviceSynchronize();      // (from here on whole-update launches take turns: ChipTurn)
  *out = h;
  return OPRL_OK;
}

extern "C" int oprl_learner_sync_params(oprl_learner* h, void* stream) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  const oprl_net* nets[OPRL_MAX_CRITICS + 1];
  for (int j = 0; j < h->nc; ++j) nets[j] = &h->cfg.critics[j];
  nets[h->nc] = &h->cfg.actor;
  float *p16[OPRL_MAX_CRITICS + 1], *p16t[OPRL_MAX_CRITICS + 1];
  for (int j = 0; j < h->nc; ++j) { p16[j] = h->pack16[1 + j]; p16t[j] = h->pack16_t[1 + j]; }
  p16[h->nc] = h->pack16[0]; p16t[h->nc] = h->pack16_t[0];
  h->stale32[0] = h->stale32[1] = false;     // (every pack is rebuilt from the master here)
  h->stale_wide = false;
  if (h->fchain) {
    const oprl_net* fn[2] = {&h->fnet[0], &h->fnet[1]};
    RC(repack_nets(fn, 2, 3, (hipStream_t)stream));
  }
  return repack_nets(nets, h->nc + 1, 3, (hipStream_t)stream, (h->bf16 || h->x2) ? p16 : nullptr, (h->bf16 || h->x2) ? p16t : nullptr, h->planes);
}

extern "C" int64_t oprl_net_pack_floats(const oprl_net* net) {
  if (!net || net->n_layers < 1 || net->n_layers > OPRL_MAX_LAYERS) return -1;
  return net_pack_floats(*net);
}

extern "C" int oprl_net_repack(const oprl_net* net, int32_t which, void* stream) {
  if (!net) { set_err("oprl_net_repack: null net"); return OPRL_ERR_INVALID; }
  int width = 0;
  RC(check_net(*net, "net", &width));
  const oprl_net* nets[1] = {net};
  return repack_nets(nets, 1, which, (hipStream_t)stream);
}

extern "C" int oprl_learner_destroy(oprl_learner* h) {
  if (!h) return OPRL_OK;
  {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    g_lazy.erase(std::remove(g_lazy.begin(), g_lazy.end(), h), g_lazy.end());
  }
  (void)hipDeviceSynchronize();
  g_live.fetch_sub(1);
  if (h->rccl.comm && h->rccl.comm_destroy) (void)h->rccl.comm_destroy(h->rccl.comm);
  if (h->xbuf) (void)hipFree(h->xbuf);
  if (h->tqc_counter) (void)hipFree(h->tqc_counter);
  if (h->lw_scratch) (void)hipFree(h->lw_scratch);
  if (h->lw_pairs.flags) (void)hipFree(h->lw_pairs.flags);
  dev_free(h->batch_alt);
  dev_free(h->uc_base);
  if (h->err_host) (void)hipHostFree(h->err_host);
  if (h->act_pin) (void)hipHostFree(h->act_pin);
  if (h->p2p.window) p2p_destroy(h->p2p);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (int j = 1; j < OPRL_MAX_CRITICS; ++j) {
    if (h->ev_join[j]) (void)hipEventDestroy(h->ev_join[j]);
    if (h->side[j]) (void)hipStreamDestroy(h->side[j]);
  }
  dev_free(h->pool.base);
  delete h;
  return OPRL_OK;
}

extern "C" int oprl_learner_update_phase(oprl_learner* h, int32_t phase, const float* s,
                                         const float* a, const float* r, const float* d,
                                         const float* s2, int32_t B, const float* noise0,
                                         const float* noise1, void* stream) {
  RC(check_batch(h, s, a, r, d, s2, B));
  RC(check_device_error(h));      // an expired wait of an earlier launch (asynchronous: whatever has run by now)
  hipStream_t st = (hipStream_t)stream;
  h->last_B = B;
  if (phase == 0) h->trace_slot = 0;
  if (!h->src.gather) { h->src.s = s; h->src.a = a; h->src.r = r; h->src.d = d; h->src.s2 = s2; }
  h->noise1_pending = noise1;
  if (phase == 0) return critic_phase(h, s, a, r, d, s2, B, noise0, st);
  if (phase == 1) {
    h->actor_updated_last = actor_due(h);
    int rc = OPRL_OK;
    if (h->actor_updated_last) rc = actor_phase(h, s, B, noise1, st);
    if (rc == OPRL_OK) h->update_count += 1;
    return rc;
  }
  set_err("phase must be 0 or 1");
  return OPRL_ERR_INVALID;
}

extern "C" int oprl_learner_update(oprl_learner* h, const float* s, const float* a, const float* r,
                                   const float* d, const float* s2, int32_t B, const float* noise0,
                                   const float* noise1, void* stream) {
  RC(oprl_learner_update_phase(h, 0, s, a, r, d, s2, B, noise0, noise1, stream));
  return oprl_learner_update_phase(h, 1, s, a, r, d, s2, B, noise0, noise1, stream);
}

extern "C" int oprl_learner_apply(oprl_learner* h, int32_t phase, double grad_scale, void* stream) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  if (!h->cfg.export_grads) { set_err("oprl_learner_apply: learner was not created with export_grads"); return OPRL_ERR_STATE; }
  RC(check_device_error(h));
  hipStream_t st = (hipStream_t)stream;
  const oprl_learner_config& c = h->cfg;
  if (phase == 0) {
    // update_count was not advanced yet for this update (phase 1 does that)
    const bool polyak = (c.algo == OPRL_TD3) ? (h->update_count % c.hp.policy_freq == 0) : true;
    // one launch: k_dw_adam's epilogue (Adam, Polyak, packs written in pack order) fed from the
    // all-reduced gradient arena instead of the GEMM — replaces k_adam_flat + k_repack
    DwArgs dw;
    dw.items = h->items_host.data(); dw.n_items = h->n_items_critic; dw.total_tiles = h->tiles_critic;
    dw.B = 0; dw.n_part = 1; dw.trace = nullptr; dw.use_row_scale = 0; dw.apply_only = 1;
    dw.ad = adam_scalars(h, c.hp.lr_critic, h->opt_step_critic, polyak, (float)grad_scale);
    dw.ad.do_adam = 1;
    HIPC(launch_dw_prof(dw, st));
    return OPRL_OK;
  }
  if (phase == 1) {
    if (!h->actor_updated_last) return OPRL_OK;
    const oprl_net& n = c.actor;
    DwArgs dw;
    dw.items = h->items_host.data() + h->n_items_critic; dw.n_items = h->n_items_actor;
    dw.total_tiles = h->tiles_actor;
    dw.B = 0; dw.n_part = 1; dw.trace = nullptr; dw.use_row_scale = 0; dw.apply_only = 1;
    dw.ad = adam_scalars(h, c.hp.lr_actor, h->opt_step_actor, n.theta_target != nullptr, (float)grad_scale);
    dw.ad.do_adam = 1;
    HIPC(launch_dw_prof(dw, st));
    if (alpha_ptr(h) != nullptr)
      HIPC(launch_alpha_step(c.log_alpha, c.log_alpha_m, c.log_alpha_v, nullptr, 1, (float)c.hp.target_entropy,
                             c.hp.lr_alpha, c.hp.beta1, c.hp.beta2, c.hp.adam_eps, h->opt_step_alpha,
                             nullptr, h->alpha_grad, (float)grad_scale, st));
    return OPRL_OK;
  }
  set_err("phase must be 0 or 1");
  return OPRL_ERR_INVALID;
}

namespace {
// step_n's K-loop as launches of up to chain_max updates each (k_ddpg_chain); h->src: the replay's view, seed set.
// set[2][5]: the two staging sets.  Also the data-parallel loop when the gradient exchange is inside the tiles.
int chain_loop(oprl_learner* h, int K, int B, float* (*set)[5], void* stream) {
  BatchSrc& sc = h->src;
  int cur = 0;
  int rc = OPRL_OK;
  h->staged_ready = false;
  for (int k = 0; k < K && rc == OPRL_OK;) {
    const int U = K - k < h->chain_max ? K - k : h->chain_max;
    float** b = set[cur];
    float** nb = set[cur ^ 1];
    for (int i = 0; i < 5; ++i) h->chain_set1[i] = nb[i];
    sc.counter = (unsigned long long)h->update_count;
    h->next_src.counter = sc.counter + 1;
    h->next_src.s = nb[0]; h->next_src.a = nb[1]; h->next_src.r = nb[2]; h->next_src.d = nb[3]; h->next_src.s2 = nb[4];
    h->prefetch_next = 0;
    sc.gather = h->staged_ready ? 0 : 1;
    sc.s = b[0]; sc.a = b[1]; sc.r = b[2]; sc.d = b[3]; sc.s2 = b[4];      // (set 0 of the launch, gathered or staged)
    h->chain_u = U;
    h->chain_pf_last = k + U < K;
    rc = oprl_learner_update(h, b[0], b[1], b[2], b[3], b[4], B, nullptr, nullptr, stream);
    h->chain_u = 1;
    if (rc == OPRL_OK) h->update_count += U - 1;        // (update() counted one)
    h->staged_ready = h->chain_pf_last;                 // the launch's last update staged the next rows: set (cur + U) & 1
    h->chain_pf_last = false;
    for (int i = 0; i < 5; ++i) h->chain_set1[i] = nullptr;
    cur = (cur + U) & 1;
    k += U;
  }
  sc.gather = 0;
  h->prefetch_next = 0;
  h->staged_ready = false;
  return rc;
}

// does step_n at this batch run as chain launches?
bool chain_ok(oprl_learner* h, int B) {
  return ddpg_args(h, B).whole && B <= 256 && h->batch_alt != nullptr;
}
}  // namespace

extern "C" int oprl_learner_step_n(oprl_learner* h, oprl_replay* replay, int32_t K, int32_t B,
                                   uint64_t seed, void* stream) {
  if (!h || !replay) { set_err("oprl_learner_step_n: null handle"); return OPRL_ERR_INVALID; }
  if (h->cfg.export_grads) { set_err("step_n is the single-GPU fused path; export_grads learners use update_phase/apply"); return OPRL_ERR_STATE; }
  int S = 0, A = 0;
  replay_dims(replay, &S, &A);
  if (S != h->S || A != h->A) { set_err("replay dims (%d,%d) != learner dims (%d,%d)", S, A, h->S, h->A); return OPRL_ERR_INVALID; }
  if (K < 0 || B < 1 || B > h->Bmax) { set_err("step_n: bad K/B"); return OPRL_ERR_INVALID; }
  if (use_fused(h, B)) {
    // the slice kernels gather their own rows (same Philox draw / index map as k_replay_gather)
    BatchSrc& sc = h->src;
    RC(oprl_replay_flush(replay, stream));
    long n_tr = 0;
    replay_view(replay, &sc.states, &sc.actions, &sc.rewards, &sc.dones, &sc.ends, &sc.n_eps, &sc.L, &n_tr);
    if (n_tr <= 0 || sc.n_eps <= 0) { set_err("step_n: replay buffer is empty"); return OPRL_ERR_STATE; }
    sc.n_transitions = n_tr;
    sc.seed = seed;
    sc.gather = 1;
    // The first update gathers in-kernel; every update's phase 2 also gathers the NEXT
    // update's rows into the staging batch (h->bs ..), which phase 1 then reads as plain rows.
    h->next_src = sc;
    // With the merged phase 2 (whose roles fill the chip) PHASE 1 carries the next update's rows — every update, TD3's
    // critic-only ones included — into the other of two staging sets, since its own roles are still reading theirs
    const size_t Bm = (size_t)h->Bmax;
    float* alt = h->batch_alt;
    float* set[2][5] = {{h->bs, h->ba, h->br, h->bd, h->bs2},
                        {alt, alt + Bm * h->S, alt + Bm * (h->S + h->A), alt + Bm * (h->S + h->A + 1), alt + Bm * (h->S + h->A + 2)}};
    // k_ddpg_chain: up to chain_max updates per launch, the rows of update u + 1 staged by update u inside the launch
    const bool chain = chain_ok(h, B);
    const DdpgArgs probe = ddpg_args(h, B);
    h->prefetch_p1 = !chain && alt != nullptr && (probe.merged & 2) != 0;
    int cur = 0;
    int rc = OPRL_OK;
    h->staged_ready = false;
    if (chain) return chain_loop(h, K, B, set, stream);
    for (int k = 0; k < K && rc == OPRL_OK; ++k) {
      float** b = set[cur];
      float** nb = set[h->prefetch_p1 ? cur ^ 1 : cur];
      h->next_src.s = nb[0]; h->next_src.a = nb[1]; h->next_src.r = nb[2]; h->next_src.d = nb[3]; h->next_src.s2 = nb[4];
      sc.counter = (unsigned long long)h->update_count;
      h->next_src.counter = sc.counter + 1;
      h->prefetch_next = (k + 1 < K) ? 1 : 0;
      // rows staged by the previous update (phase 2 does not run on TD3's critic-only steps), else the slice kernels
      // gather their own
      sc.gather = h->staged_ready ? 0 : 1;
      h->staged_ready = false;
      rc = oprl_learner_update(h, b[0], b[1], b[2], b[3], b[4], B, nullptr, nullptr, stream);
      if (h->prefetch_p1) cur ^= 1;
    }
    sc.gather = 0;
    h->prefetch_next = 0;
    h->prefetch_p1 = false;
    h->staged_ready = false;
    return rc;
  }
  if (h->cfg.algo == OPRL_TQC && h->batch_alt != nullptr && !h->no_gather_ride && K > 1) {
    // the rows of update k + 1 are gathered by riders of update k's k_lw_dact launch (same draw as k_replay_gather)
    // into the other of two sets of batch rows; only the first update's rows are a launch
    PrefetchJob base;
    memset((void*)&base, 0, sizeof base);
    RC(oprl_replay_flush(replay, stream));
    long n_tr = 0;
    replay_view(replay, &base.next.states, &base.next.actions, &base.next.rewards, &base.next.dones, &base.next.ends,
                &base.next.n_eps, &base.next.L, &n_tr);
    if (n_tr <= 0 || base.next.n_eps <= 0) { set_err("step_n: replay buffer is empty"); return OPRL_ERR_STATE; }
    base.next.n_transitions = n_tr;
    base.next.seed = seed;
    base.next.gather = 1;
    base.S = h->S; base.A = h->A; base.B = B; base.z0 = -1;
    const size_t Bm = (size_t)h->Bmax;
    float* alt = h->batch_alt;
    float* set[2][5] = {{h->bs, h->ba, h->br, h->bd, h->bs2},
                        {alt, alt + Bm * h->S, alt + Bm * (h->S + h->A), alt + Bm * (h->S + h->A + 1), alt + Bm * (h->S + h->A + 2)}};
    int cur = 0;
    bool staged = false;
    int rc = OPRL_OK;
    for (int k = 0; k < K && rc == OPRL_OK; ++k) {
      float** b = set[cur];
      if (!staged)
        rc = oprl_replay_sample(replay, B, nullptr, seed, (uint64_t)h->update_count, b[0], b[1], b[2], b[3], b[4], nullptr, nullptr, stream);
      staged = false;
      h->prefetch_done = false;
      h->prefetch_pending = false;
      if (rc == OPRL_OK && k + 1 < K) {
        float** nb = set[cur ^ 1];
        h->prefetch = base;
        h->prefetch.next.counter = (unsigned long long)h->update_count + 1;
        h->prefetch.next.s = nb[0]; h->prefetch.next.a = nb[1]; h->prefetch.next.r = nb[2]; h->prefetch.next.d = nb[3];
        h->prefetch.next.s2 = nb[4];
        h->prefetch_pending = true;
      }
      if (rc == OPRL_OK) rc = oprl_learner_update(h, b[0], b[1], b[2], b[3], b[4], B, nullptr, nullptr, stream);
      h->prefetch_pending = false;
      staged = h->prefetch_done;
      h->prefetch_done = false;
      cur ^= 1;
    }
    return rc;
  }
  for (int k = 0; k < K; ++k) {
    RC(oprl_replay_sample(replay, B, nullptr, seed, (uint64_t)h->update_count, h->bs, h->ba, h->br,
                          h->bd, h->bs2, nullptr, nullptr, stream));
    RC(oprl_learner_update(h, h->bs, h->ba, h->br, h->bd, h->bs2, B, nullptr, nullptr, stream));
  }
  return OPRL_OK;
}

// ===================================================================== the trainer loop's step
extern "C" int oprl_learner_step_act(oprl_learner* h, oprl_replay* replay, int32_t B, uint64_t seed,
                                     const float* obs_host, void* stream) {
  if (!h || !replay || !obs_host) { set_err("oprl_learner_step_act: null argument"); return OPRL_ERR_INVALID; }
  const oprl_net& n = h->cfg.actor;
  if (n.n_layers < 1 || n.n_layers > kMaxLayers) { set_err("oprl_learner_step_act: bad actor"); return OPRL_ERR_INVALID; }
  for (int l = 0; l <= n.n_layers; ++l)
    if (n.dims[l] > kPolicyActMaxWidth) { set_err("oprl_learner_step_act: layer width %d > %d", n.dims[l], kPolicyActMaxWidth); return OPRL_ERR_INVALID; }
  if (h->act_pin == nullptr) {
    HIPC(hipHostMalloc((void**)&h->act_pin, (3 * kPolicyActMaxWidth + 16) * sizeof(float), hipHostMallocMapped));
    HIPC(hipHostGetDevicePointer((void**)&h->act_map, h->act_pin, 0));
    memset(h->act_pin, 0, (3 * kPolicyActMaxWidth + 16) * sizeof(float));
  }
  RC(oprl_learner_step_n(h, replay, 1, B, seed, stream));
  memcpy(h->act_pin, obs_host, sizeof(float) * n.dims[0]);
  PolicyActArgs a;
  memset(&a, 0, sizeof a);
  a.n_layers = n.n_layers;
  for (int l = 0; l <= n.n_layers; ++l) a.dims[l] = n.dims[l];
  for (int l = 0; l < n.n_layers; ++l) { a.w[l] = n.theta + w_off(n, l); a.b[l] = n.theta + b_off(n, l); }
  a.obs = h->act_map;
  a.out = reinterpret_cast<unsigned long long*>(h->act_map + kPolicyActMaxWidth);
  h->act_ticket += 1;
  if (h->act_ticket == 0) h->act_ticket = 1;
  a.ticket_value = h->act_ticket;
  HIPC(launch_policy_act(a, (hipStream_t)stream));
  h->act_pending = true;
  return OPRL_OK;
}

extern "C" int oprl_learner_act_wait(oprl_learner* h, float* out_host, int32_t n_out, int64_t timeout_us) {
  if (!h || !out_host) { set_err("oprl_learner_act_wait: null argument"); return OPRL_ERR_INVALID; }
  if (!h->act_pending) { set_err("oprl_learner_act_wait: no row is pending (oprl_learner_step_act first)"); return OPRL_ERR_STATE; }
  const oprl_net& n = h->cfg.actor;
  if (n_out != n.dims[n.n_layers]) { set_err("oprl_learner_act_wait: n_out %d != the actor's %d outputs", n_out, n.dims[n.n_layers]); return OPRL_ERR_INVALID; }
  const unsigned long long* g = reinterpret_cast<const unsigned long long*>(h->act_pin + kPolicyActMaxWidth);
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  for (int i = 0; i < n_out; ++i) {
    unsigned long long x;
    while ((unsigned)((x = __atomic_load_n(g + i, __ATOMIC_ACQUIRE)) >> 32) != h->act_ticket) {
      if ((++spins & 1023) == 0 &&
          std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_us) {
        set_err("oprl_learner_act_wait: the policy row did not arrive within %lld us", (long long)timeout_us);
        return OPRL_ERR_STATE;
      }
      __builtin_ia32_pause();
    }
    const unsigned bits = (unsigned)x;
    memcpy(out_host + i, &bits, 4);
  }
  h->act_pending = false;
  return OPRL_OK;
}

// ===================================================================== packed learners
// The reference trains several seeds as several processes (runners/train.py:36-50).  One DDPG learner at B = 256
// is a chain of four latency-bound launches that keeps a fraction of the chip busy; a GROUP steps N independent
// learners (own weights, own replay keys) with FOUR launches per update for all of them: grid.z = learner, the
// argument blocks in device memory.  Group members run the single-CU-per-slice passes (cluster size 1): no
// workgroup of such a launch waits for a later one, so the N x 48 phase-1 workgroups may simply queue behind
// each other on the 256 CUs, and a learner's result does not depend on who else is in the launch.
struct oprl_group {
  std::vector<oprl_learner*> L;
  // The argument blocks of kGroupChunk updates — per update [N x DdpgArgs phase 1][N x DdpgArgs phase 2][N x DwKArgsN<ni_c>
  // critic(s)][N x DwKArgsG actor] — are built ahead on the host and go up in ONE copy per chunk (four copies per update
  // of the 5 KB blocks stood for 56 of 424 us per group update of 32 members).
  char* dev = nullptr;                         // [kGroupChunk][bytes]
  char* stage[2] = {nullptr, nullptr};         // pinned host staging (double buffered), the same layout
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};
  int cur = 0;
  size_t bytes = 0;                            // one update's blocks
  int span = 1;                                // XCDs a member's slices are dealt out to (generic passes: 1)
  int ni_c = kDwGroupItems;                    // layers per critic-step dW block (twin critics: kDwGroupItems2)
  int device = 0;                              // the device the group's buffers (and its members) live on
};
constexpr int kGroupChunk = 4;

static void group_free(oprl_group* g) {
  if (g->dev) (void)hipFree(g->dev);
  for (int i = 0; i < 2; ++i) {
    if (g->stage[i]) (void)hipHostFree(g->stage[i]);
    if (g->stage_ev[i]) (void)hipEventDestroy(g->stage_ev[i]);
  }
  delete g;
}

extern "C" int oprl_group_create(oprl_learner** learners, int32_t n, oprl_group** out) {
  if (!learners || !out || n < 1 || n > 64) { set_err("oprl_group_create: invalid argument"); return OPRL_ERR_INVALID; }
  const int algo0 = learners[0] ? learners[0]->cfg.algo : -1;
  for (int i = 0; i < n; ++i) {
    oprl_learner* h = learners[i];
    if (!h || (algo0 != OPRL_DDPG && algo0 != OPRL_TD3 && algo0 != OPRL_SAC) || h->cfg.algo != algo0 || !h->fused || h->cfg.export_grads ||
        h->bf16 != learners[0]->bf16 || h->x2 != learners[0]->x2 || h->S != learners[0]->S || h->A != learners[0]->A ||
        h->Bmax != learners[0]->Bmax || h->cfg.hp.policy_freq != learners[0]->cfg.hp.policy_freq ||
        (alpha_ptr(h) != nullptr) != (alpha_ptr(learners[0]) != nullptr)) {
      set_err("oprl_group_create: member %d is not a fused DDPG / TD3 / SAC learner of the group's algorithm, shape and precision", i);
      return OPRL_ERR_INVALID;
    }
  }
  // The members' launch form.  Exact fp32: the generic single-CU-per-slice passes (cluster size 1) — no workgroup of such
  // a launch waits for another, and 32 members measure 71k updates/s against 58k on clusters of four.  bf16 / x2: the
  // lean passes on clusters of four (the only form these precisions exist in).  OPRL_AMD_GROUP_NC=4: clusters of four
  // for exact fp32 as well.  TD3 / SAC members (fused in the lean form only): clusters of four in every precision.
  int group_nc = 4;
  {
    const int env_nc = 0;
    oprl_learner* h0 = learners[0];
    const int keep_ncl = h0->ncl;
    const bool keep_sc = h0->shared_chip;
    const int keep_nw = h0->no_wide;
    h0->ncl = 4; h0->shared_chip = true; h0->no_wide = 1;
    const bool lean = fused_ddpg_is_lean(ddpg_args(h0, h0->Bmax));
    h0->ncl = keep_ncl; h0->shared_chip = keep_sc; h0->no_wide = keep_nw;
    if (!lean || (algo0 == OPRL_DDPG && !h0->bf16 && !h0->x2 && env_nc != 4)) group_nc = 1;
    if (group_nc == 1 && (h0->bf16 || h0->x2 || algo0 != OPRL_DDPG)) {
      set_err("oprl_group_create: TD3 / SAC members and the bf16 / x2 modes need nets the lean passes take (256-wide hidden layers, narrow inputs)");
      return OPRL_ERR_INVALID;
    }
  }
  auto* g = new oprl_group();
  g->L.assign(learners, learners + n);
  (void)hipGetDevice(&g->device);
  g->ni_c = learners[0]->nc == 2 ? kDwGroupItems2 : kDwGroupItems;
  g->span = 1;        // (a member's slices on one XCD: 2 / 4 / 8 measured slower, r03-39)
  g->bytes = (size_t)n * (2 * sizeof(DdpgArgs) + dw_group_block_bytes(g->ni_c) + dw_group_block_bytes(kDwGroupItems));
  bool ok = hipMalloc((void**)&g->dev, g->bytes * kGroupChunk) == hipSuccess;
  for (int i = 0; i < 2 && ok; ++i) {
    ok = hipHostMalloc((void**)&g->stage[i], g->bytes * kGroupChunk) == hipSuccess &&
         hipEventCreateWithFlags(&g->stage_ev[i], hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) {      // (nothing is kept of a failed create: the partial allocations go, the members stay as they were)
    group_free(g);
    set_err("oprl_group_create: allocation failed");
    return OPRL_ERR_NOMEM;
  }
  // (a solo run for comparison: oprl_learner_set_cluster(h, 4) — the un-merged lean launches — or (h, 1))
  // (the twin critics' side-by-side forms want all of a slice's clusters resident at once: not in a queue of members)
  for (oprl_learner* h : g->L) { h->ncl = group_nc; h->shared_chip = true; h->no_wide = 1; h->no_twin_split = true; h->no_p2_pair = true; }
  *out = g;
  return OPRL_OK;
}

extern "C" int oprl_group_destroy(oprl_group* g) {
  if (!g) return OPRL_OK;
  // the group's OWN device, whatever the caller's current one is: launches that read the argument blocks may be in flight
  int cur = 0;
  const int dev = g->device;
  (void)hipGetDevice(&cur);
  if (cur != dev) (void)hipSetDevice(dev);
  (void)hipDeviceSynchronize();
  group_free(g);
  if (cur != dev) (void)hipSetDevice(cur);
  return OPRL_OK;
}

extern "C" int oprl_learner_set_cluster(oprl_learner* h, int32_t nc) {
  if (!h || (nc != 1 && nc != 2 && nc != 4 && nc != 8)) { set_err("oprl_learner_set_cluster: cluster size must be 1, 2, 4 or 8"); return OPRL_ERR_INVALID; }
  // 8 = clusters of four, and of eight where the fused kernels have them (the default); 4 = never eight
  h->ncl = nc == 8 ? 4 : nc;
  static const bool env_off = [] { const char* e = getenv("OPRL_AMD_NO_WIDE"); return e != nullptr && atoi(e) != 0; }();
  h->no_wide = (nc == 8 && !env_off) ? 0 : 1;
  // ... and a learner that shares the chip (anything but 8) keeps to the launch forms whose workgroups only wait within
  // their cluster: no tile workgroups riding on the phase launches (measured: 8 learners on 8 streams 47k -> 60k aggregate)
  h->shared_chip = nc != 8;
  return OPRL_OK;
}

// K updates of every member: per update one H2D copy of the N x 4 argument blocks and four launches.
extern "C" int oprl_group_step_n(oprl_group* g, oprl_replay* replay, int32_t K, int32_t B, const uint64_t* seeds,
                                 void* stream) {
  if (!g || !replay || !seeds || K < 0) { set_err("oprl_group_step_n: invalid argument"); return OPRL_ERR_INVALID; }
  const int n = (int)g->L.size();
  oprl_learner* h0 = g->L[0];
  if (B < 1 || B > h0->Bmax) { set_err("oprl_group_step_n: bad batch %d", B); return OPRL_ERR_INVALID; }
  int S = 0, A = 0;
  replay_dims(replay, &S, &A);
  if (S != h0->S || A != h0->A) { set_err("replay dims (%d,%d) != group dims (%d,%d)", S, A, h0->S, h0->A); return OPRL_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  RC(oprl_replay_flush(replay, stream));
  // everything that can be refused is checked BEFORE any member's counters move: the members advance together, so
  // being in phase now is being in phase for all K updates
  for (int l = 0; l < n; ++l) {
    RC(check_device_error(g->L[l]));
    if (actor_due(g->L[l]) != actor_due(h0) || g->L[l]->cfg.hp.policy_freq != h0->cfg.hp.policy_freq) {
      set_err("oprl_group_step_n: the members' delayed actor steps are out of phase (update counts differ modulo policy_freq)");
      return OPRL_ERR_STATE;
    }
  }
  // (what is left — an internal inconsistency of the launch tables — rolls the members' counters back to here)
  struct Snap { unsigned epoch, tp_tag; long long update_count; int oc, oa, oal; bool staged, aul, s0, s1; };
  std::vector<Snap> snap(n);
  auto take = [&]() {
    for (int l = 0; l < n; ++l) {
      const oprl_learner* h = g->L[l];
      snap[l] = Snap{h->epoch, h->tp_tag, (long long)h->update_count, (int)h->opt_step_critic, (int)h->opt_step_actor, (int)h->opt_step_alpha,
                     h->staged_ready, h->actor_updated_last, h->stale32[0], h->stale32[1]};
    }
  };
  auto roll_back = [&]() {
    for (int l = 0; l < n; ++l) {
      oprl_learner* h = g->L[l];
      const Snap& q = snap[l];
      h->epoch = q.epoch; h->tp_tag = q.tp_tag; h->update_count = q.update_count; h->opt_step_critic = q.oc; h->opt_step_actor = q.oa;
      h->opt_step_alpha = q.oal; h->staged_ready = q.staged; h->actor_updated_last = q.aul; h->stale32[0] = q.s0; h->stale32[1] = q.s1;
    }
  };
  for (int l = 0; l < n; ++l) {
    oprl_learner* h = g->L[l];
    BatchSrc& sc = h->src;
    long n_tr = 0;
    replay_view(replay, &sc.states, &sc.actions, &sc.rewards, &sc.dones, &sc.ends, &sc.n_eps, &sc.L, &n_tr);
    if (n_tr <= 0 || sc.n_eps <= 0) { set_err("oprl_group_step_n: replay buffer is empty"); return OPRL_ERR_STATE; }
    sc.n_transitions = n_tr;
    sc.seed = seeds[l];
    sc.gather = 1;
    sc.s = h->bs; sc.a = h->ba; sc.r = h->br; sc.d = h->bd; sc.s2 = h->bs2;
    h->next_src = sc;
    h->staged_ready = false;
    h->last_B = B;
  }
  static_assert(sizeof(DdpgArgs) % 8 == 0 && sizeof(DwKArgsG) % 8 == 0 && sizeof(DwKArgsG2) % 8 == 0, "the blocks of an update lie back to back");
  const size_t dc_bytes = dw_group_block_bytes(g->ni_c), da_bytes = dw_group_block_bytes(kDwGroupItems);
  for (oprl_learner* h : g->L) h->noise1_pending = nullptr;
  for (int k0 = 0; k0 < K; k0 += kGroupChunk) {
    const int m = K - k0 < kGroupChunk ? K - k0 : kGroupChunk;
    const int c = g->cur;
    if (g->stage_busy[c]) { HIPC(hipEventSynchronize(g->stage_ev[c])); g->stage_busy[c] = false; }
    take();                                   // nothing of this chunk has been launched until its blocks are complete
    int tiles_c = 0, tiles_a = 0;
    DdpgArgs first[kGroupChunk][2];           // member 0's blocks of each update (for the grids)
    bool due[kGroupChunk];                    // TD3: the actor steps every policy_freq updates — of ALL members at once
    for (int j = 0; j < m; ++j) {
      const int k = k0 + j;
      DdpgArgs* p1 = reinterpret_cast<DdpgArgs*>(g->stage[c] + (size_t)j * g->bytes);
      DdpgArgs* p2 = p1 + n;
      char* dc = reinterpret_cast<char*>(p2 + n);
      char* da = dc + (size_t)n * dc_bytes;
      due[j] = actor_due(g->L[0]);
      for (int l = 0; l < n; ++l) {
        oprl_learner* h = g->L[l];
        const oprl_learner_config& cf = h->cfg;
        if (actor_due(h) != due[j]) { roll_back(); set_err("oprl_group_step_n: the members' delayed actor steps are out of phase (update counts differ modulo policy_freq)"); return OPRL_ERR_STATE; }
        h->src.counter = (unsigned long long)h->update_count;
        h->next_src.counter = h->src.counter + 1;
        h->src.gather = h->staged_ready ? 0 : 1;
        h->staged_ready = false;
        const int prefetch = (k + 1 < K && due[j]) ? 1 : 0;     // (the row of phase 2's launch: actor steps only)
        h->epoch += 1;
        if (h->epoch == 0) { h->epoch = 1; HIPC(hipMemsetAsync(h->y_granules, 0, ((size_t)3 * h->Bmax + 256) * sizeof(unsigned long long), st)); }
        p1[l] = ddpg_args(h, B);
        p1[l].group_span = g->span;
        RC(next_tp_tag(&h->tp_tag, h->xbuf, h->xbuf_granules * sizeof(unsigned long long), st, &p1[l].cluster_tag));
        DwKArgs kd;
        // (as the un-merged launches of a solo learner; TD3 moves its targets on actor steps only)
        DwArgs dw = dw_build(h, true, B, cf.algo == OPRL_TD3 ? due[j] : true, false);
        const int tc = fill_dw_kargs(dw, &kd) < 0 ? -1 : compact_dw_kargs(kd, dc + (size_t)l * dc_bytes, g->ni_c);
        int ta = tiles_a;
        if (due[j]) {
          p2[l] = ddpg_args(h, B);
          p2[l].group_span = g->span;
          RC(next_tp_tag(&h->tp_tag, h->xbuf, h->xbuf_granules * sizeof(unsigned long long), st, &p2[l].cluster_tag));   // (a launch, a tag)
          p2[l].prefetch_next = prefetch;
          h->staged_ready = prefetch != 0;
          dw = dw_build(h, false, B, cf.actor.theta_target != nullptr, alpha_rides(h));    // (SAC: the temperature step rides)
          ta = fill_dw_kargs(dw, &kd) < 0 ? -1 : compact_dw_kargs(kd, da + (size_t)l * da_bytes, kDwGroupItems);
          if (l == 0 && tiles_a == 0) tiles_a = ta;
          if (ta < 0 || ta != tiles_a || p2[l].nc != p1[l].nc || p2[l].merged || p2[l].wide || p2[l].whole || p2[l].p2_pair) ta = -1;
        }
        if (l == 0 && j == 0) tiles_c = tc;
        if (tc < 0 || ta < 0 || tc != tiles_c || p1[l].nc != p1[0].nc || p1[l].merged || p1[l].wide || p1[l].whole || p1[l].twin_split) {
          roll_back();
          set_err("oprl_group_step_n: internal: bad launch arguments");
          return OPRL_ERR_INVALID;
        }
        h->actor_updated_last = due[j];
        h->update_count += 1;
      }
      first[j][0] = p1[0];
      if (due[j]) first[j][1] = p2[0];
    }
    HIPC(hipMemcpyAsync(g->dev, g->stage[c], g->bytes * m, hipMemcpyHostToDevice, st));
    HIPC(hipEventRecord(g->stage_ev[c], st));
    g->stage_busy[c] = true;
    g->cur ^= 1;
    for (int j = 0; j < m; ++j) {
      const DdpgArgs* p1 = reinterpret_cast<const DdpgArgs*>(g->dev + (size_t)j * g->bytes);
      const DdpgArgs* p2 = p1 + n;
      const char* dc = reinterpret_cast<const char*>(p2 + n);
      const char* da = dc + (size_t)n * dc_bytes;
      HIPC(launch_ddpg_phase1_group(first[j][0], p1, n, st));
      HIPC(launch_dw_adam_group(dc, g->ni_c, n, tiles_c, st));
      if (!due[j]) continue;
      HIPC(launch_ddpg_phase2_group(first[j][1], p2, n, st));
      HIPC(launch_dw_adam_group(da, kDwGroupItems, n, tiles_a, st));
    }
  }
  for (oprl_learner* h : g->L) { h->src.gather = 0; h->prefetch_next = 0; h->staged_ready = false; }
  return OPRL_OK;
}

extern "C" int oprl_learner_read_scalars(oprl_learner* h, float* out_host, int32_t n, void* stream) {
  if (!h || !out_host || n < 1) { set_err("oprl_learner_read_scalars: invalid argument"); return OPRL_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const int B = h->last_B > 0 ? h->last_B : 1;
  const int n_slices = (B + kR - 1) / kR;
  const oprl_learner_config& c = h->cfg;
  float loss_scale = 1.0f / (float)B;
  if (c.algo == OPRL_TQC) {
    const int Q = c.hp.n_quantiles, M = h->nc * Q - c.hp.top_quantiles_to_drop;
    loss_scale = 1.0f / ((float)B * (float)h->nc * (float)Q * (float)M);
  }
  // critic partials of all critics are contiguous: loss sums over critics (td1 + td2)
  HIPC(launch_reduce_partials(h->part_c, n_slices * h->nc, h->scalars, 0, loss_scale,
                              1.0f / ((float)B * (float)h->nc), st));
  HIPC(launch_reduce_partials(h->part_a, n_slices, h->scalars, 4, 0.f, -1.0f / (float)B, st));
  // critic 0 alone (the reference logs q1, not the twin mean) and the mean log-density of the actor step
  HIPC(launch_reduce_partials(h->part_c, n_slices, h->scalars, 8, loss_scale, 1.0f / (float)B, st));
  const bool gauss = c.algo == OPRL_SAC || c.algo == OPRL_TQC;
  if (gauss) HIPC(launch_sum(h->logp, B, h->scalars, 12, 1.0f / (float)B, st));
  float host[16] = {0};
  HIPC(hipMemcpyAsync(host, h->scalars, sizeof(float) * 13, hipMemcpyDeviceToHost, st));
  double la = 0.0;
  const double* lap = alpha_ptr(h);
  if (lap) HIPC(hipMemcpyAsync(&la, lap, sizeof(double), hipMemcpyDeviceToHost, st));
  HIPC(hipStreamSynchronize(st));
  RC(check_device_error(h));      // after the synchronisation: definitive for everything launched so far
  float res[6];
  res[0] = host[0];                 // critic loss
  res[1] = host[5];                 // actor loss (-mean q part)
  res[2] = host[1];                 // mean q
  res[3] = host[2];                 // mean TD target
  res[4] = lap ? (float)exp(la) : (float)c.hp.alpha_init;
  res[5] = (float)h->update_count;
  float res2[4];
  res2[0] = host[9];                                  // mean q of critic 0 (the reference's "q1")
  res2[1] = gauss ? host[12] : 0.f;                   // mean log pi(a|s) of the last actor step
  // SAC / TQC actor loss as the reference forms it: alpha * mean(log pi) - mean(min q)   (sac.py:124-126)
  res2[2] = gauss ? res[4] * res2[1] + res[1] : res[1];
  // temperature loss -log_alpha * (target_entropy + mean log pi)   (sac.py:133-135; with the CURRENT log_alpha)
  res2[3] = lap ? (float)(-la * (c.hp.target_entropy + (double)res2[1])) : 0.f;
  for (int i = 0; i < n && i < 6; ++i) out_host[i] = res[i];
  for (int i = 6; i < n && i < 10; ++i) out_host[i] = res2[i - 6];
  return OPRL_OK;
}

extern "C" int oprl_learner_set_trace(oprl_learner* h, int64_t* buf) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  h->trace = (long long*)buf;
  return OPRL_OK;
}

extern "C" int oprl_learner_update_count(oprl_learner* h, int64_t* out_host) {
  if (!h || !out_host) { set_err("null argument"); return OPRL_ERR_INVALID; }
  *out_host = h->update_count;
  return OPRL_OK;
}

extern "C" int oprl_learner_set_update_count(oprl_learner* h, int64_t count) {
  if (!h || count < 0) { set_err("invalid argument"); return OPRL_ERR_INVALID; }
  h->update_count = count;
  return OPRL_OK;
}

extern "C" int oprl_learner_set_seed(oprl_learner* h, uint64_t seed, int32_t rank) {
  if (!h || rank < 0) { set_err("oprl_learner_set_seed: invalid argument"); return OPRL_ERR_INVALID; }
  h->noise_seed = seed;
  h->noise_rank = rank;
  return OPRL_OK;
}

extern "C" int oprl_learner_get_counters(oprl_learner* h, int64_t out_host[OPRL_N_COUNTERS]) {
  if (!h || !out_host) { set_err("null argument"); return OPRL_ERR_INVALID; }
  out_host[0] = h->update_count;
  out_host[1] = h->opt_step_critic;
  out_host[2] = h->opt_step_actor;
  out_host[3] = h->opt_step_alpha;
  return OPRL_OK;
}

extern "C" int oprl_learner_set_counters(oprl_learner* h, const int64_t in_host[OPRL_N_COUNTERS]) {
  if (!h || !in_host) { set_err("null argument"); return OPRL_ERR_INVALID; }
  for (int k = 0; k < OPRL_N_COUNTERS; ++k)
    if (in_host[k] < 0 || in_host[k] > 0x7fffffffLL) { set_err("counter %d out of range", k); return OPRL_ERR_INVALID; }
  h->update_count = in_host[0];
  h->opt_step_critic = (int)in_host[1];
  h->opt_step_actor = (int)in_host[2];
  h->opt_step_alpha = (int)in_host[3];
  return OPRL_OK;
}

extern "C" int oprl_learner_debug_ptrs(oprl_learner* h, const float** q, const float** y) {
  if (!h) { set_err("null learner handle"); return OPRL_ERR_INVALID; }
  if (q) *q = h->qdbg;
  if (y) *y = h->ydbg;
  return OPRL_OK;
}

// (debug) device views of the workspace a fused DDPG update leaves behind: tools/race_hunt.py compares them between a
// chain learner and a one-update-per-launch learner.  which: 0..2 the actor's X rows, 3 pi, 4 unit-seed rows, 5 du granules
// (8 bytes each), 6..8 the critic's X rows, 9 / 10 the critic's dY rows (first hidden partials | second hidden), 11 the TD
// seed granules, 12 the batch rows of the last update's set (s), 13 the other set
extern "C" int oprl_learner_debug_view(oprl_learner* h, int32_t which, const void** ptr, int64_t* n_bytes) {
  if (!h || !ptr || !n_bytes) { set_err("oprl_learner_debug_view: invalid argument"); return OPRL_ERR_INVALID; }
  const size_t B = (size_t)h->Bmax;
  const void* p = nullptr;
  size_t n = 0;
  switch (which) {
    case 0: p = h->ws_actor.X[0]; n = B * h->ws_actor.ldx0 * 4; break;
    case 1: p = h->ws_actor.X[1]; n = B * h->ws_actor.width * 4; break;
    case 2: p = h->ws_actor.X[2]; n = B * h->ws_actor.width * 4; break;
    case 3: p = h->pi; n = B * h->A * 4; break;
    case 4: p = h->gu; n = h->gu ? (size_t)h->A * (B < 256 ? B : 256) * 256 * 4 : 0; break;
    case 5: p = h->du_granules; n = h->du_granules ? (B < 256 ? B : 256) * kDuLd * 8 : 0; break;
    case 6: p = h->ws_critic[0].X[0]; n = B * h->ws_critic[0].ldx0 * 4; break;
    case 7: p = h->ws_critic[0].X[1]; n = B * h->ws_critic[0].width * 4; break;
    case 8: p = h->ws_critic[0].X[2]; n = B * h->ws_critic[0].width * 4; break;
    case 9: p = h->ws_critic[0].dY[0]; n = B * h->ws_critic[0].width * 4; break;
    case 10: p = h->ws_critic[0].dY[1]; n = B * h->ws_critic[0].width * 4; break;
    case 11: p = h->y_granules; n = B * 8; break;
    case 12: p = h->bs; n = B * h->S * 4; break;
    case 13: p = h->batch_alt; n = h->batch_alt ? B * h->S * 4 : 0; break;
    default: set_err("oprl_learner_debug_view: no such view"); return OPRL_ERR_INVALID;
  }
  *ptr = p; *n_bytes = (int64_t)n;
  return OPRL_OK;
}

// ---------------------------------------------------------------- building blocks
namespace {
struct TmpBuf {  // small per-thread device scratch for the stand-alone MLP calls
  float* p = nullptr;
  size_t cap = 0;
  float* get(size_t floats) {
    if (floats > cap) {
      if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
      if (hipMalloc(&p, floats * sizeof(float)) != hipSuccess) return nullptr;
      cap = floats;
    }
    return p;
  }
};
thread_local TmpBuf g_tmp;
bool g_attrs_done = false;
}  // namespace

extern "C" int oprl_mlp_forward(const oprl_net* net, int32_t use_target, const float* x0, int32_t k0,
                                const float* x1, int32_t k1, int32_t B, int32_t out_act, float* out,
                                void* stream) {
  if (!net || !x0 || !out || B < 1) { set_err("oprl_mlp_forward: invalid argument"); return OPRL_ERR_INVALID; }
  int width = 0;
  RC(check_net(*net, "net", &width));
  if (use_target && !net->theta_target) { set_err("oprl_mlp_forward: no target arena"); return OPRL_ERR_INVALID; }
  if (k0 + (x1 ? k1 : 0) != net->dims[0]) { set_err("oprl_mlp_forward: k0+k1=%d != input dim %d", k0 + (x1 ? k1 : 0), net->dims[0]); return OPRL_ERR_INVALID; }
  if (out_act != ACT_NONE && out_act != ACT_TANH && out_act != ACT_GAUSS_MEAN) { set_err("oprl_mlp_forward: out_act %d unsupported here", out_act); return OPRL_ERR_INVALID; }
  if (!g_attrs_done) { HIPC(init_kernel_attrs()); g_attrs_done = true; }
  RC(fresh32(net, (hipStream_t)stream));
  MlpArgs a;
  memset(&a, 0, sizeof a);
  a.net = net_view(*net, use_target != 0);
  a.B = B; a.do_fwd = 1;
  a.x0 = x0; a.k0 = k0; a.x1 = x1; a.k1 = x1 ? k1 : 0;
  a.out_act = out_act;
  const int nout = net->dims[net->n_layers];
  a.action_dim = nout / 2;
  a.out = out; a.ldo = (out_act == ACT_GAUSS_MEAN) ? nout / 2 : nout;
  return launch(a, width, (hipStream_t)stream);
}

// One observation in HOST memory -> one output row in HOST memory: what a policy's explore() /
// exploit() does once per environment step (reference nn_models.py:138-150, 180-195: as_tensor ->
// forward -> .cpu()).  Pinned staging rows on both sides, one H2D copy, one slice launch, one D2H
// copy and a stream sync — four runtime calls instead of the dozen torch dispatches around
// oprl_mlp_forward (37 us for as_tensor alone).
namespace {
struct ActStage {
  std::mutex mu;
  float* host = nullptr;   // pinned: [0, 256) observation, [256, 512) output
  float* dev = nullptr;    // device: same layout
};
ActStage g_act;
}  // namespace

extern "C" int oprl_mlp_act(const oprl_net* net, const float* obs_host, int32_t k0, int32_t out_act,
                            float* out_host, int32_t n_out, void* stream) {
  if (!net || !obs_host || !out_host) { set_err("oprl_mlp_act: invalid argument"); return OPRL_ERR_INVALID; }
  int width = 0;
  RC(check_net(*net, "net", &width));
  const int nout = net->dims[net->n_layers];
  const int want = (out_act == ACT_GAUSS_MEAN) ? nout / 2 : nout;
  if (k0 != net->dims[0] || k0 > 256 || n_out != want || want > 256) {
    set_err("oprl_mlp_act: dims (%d in, %d out) do not match the net (%d in, %d out)", k0, n_out, net->dims[0], want);
    return OPRL_ERR_INVALID;
  }
  if (out_act != ACT_NONE && out_act != ACT_TANH && out_act != ACT_GAUSS_MEAN) { set_err("oprl_mlp_act: out_act %d unsupported here", out_act); return OPRL_ERR_INVALID; }
  if (!g_attrs_done) { HIPC(init_kernel_attrs()); g_attrs_done = true; }
  hipStream_t st = (hipStream_t)stream;
  RC(fresh32(net, st));
  std::lock_guard<std::mutex> lk(g_act.mu);
  if (g_act.host == nullptr) {
    HIPC(hipHostMalloc((void**)&g_act.host, 512 * sizeof(float), hipHostMallocDefault));
    HIPC(hipMalloc((void**)&g_act.dev, 512 * sizeof(float)));
  }
  memcpy(g_act.host, obs_host, sizeof(float) * k0);
  HIPC(hipMemcpyAsync(g_act.dev, g_act.host, sizeof(float) * k0, hipMemcpyHostToDevice, st));
  MlpArgs a;
  memset(&a, 0, sizeof a);
  a.net = net_view(*net, false);
  a.B = 1; a.do_fwd = 1;
  a.x0 = g_act.dev; a.k0 = k0;
  a.out_act = out_act;
  a.action_dim = nout / 2;
  a.out = g_act.dev + 256; a.ldo = want;
  RC(launch(a, width, st));
  HIPC(hipMemcpyAsync(g_act.host + 256, g_act.dev + 256, sizeof(float) * want, hipMemcpyDeviceToHost, st));
  HIPC(hipStreamSynchronize(st));
  memcpy(out_host, g_act.host + 256, sizeof(float) * want);
  return OPRL_OK;
}

extern "C" int oprl_mlp_backward(const oprl_net* net, const float* x0, int32_t k0, const float* x1,
                                 int32_t k1, int32_t B, const float* dout, float* dx, void* stream) {
  if (!net || !x0 || !dout || B < 1 || !net->grad) { set_err("oprl_mlp_backward: invalid argument (grad arena required)"); return OPRL_ERR_INVALID; }
  int width = 0;
  RC(check_net(*net, "net", &width));
  if (k0 + (x1 ? k1 : 0) != net->dims[0]) { set_err("oprl_mlp_backward: input dims mismatch"); return OPRL_ERR_INVALID; }
  if (!g_attrs_done) { HIPC(init_kernel_attrs()); g_attrs_done = true; }
  hipStream_t st = (hipStream_t)stream;
  RC(fresh32(net, st));
  NetWs ws;
  float* base = g_tmp.get(net_ws_floats(*net, B) + sizeof(DwItem) * kMaxLayers / sizeof(float) + 2048);
  if (!base) { set_err("oprl_mlp_backward: scratch allocation failed"); return OPRL_ERR_NOMEM; }
  Pool p; p.base = (char*)base; p.cap = (size_t)-1;
  alloc_net_ws(p, *net, B, &ws);
  std::vector<DwItem> items;
  int tiles = 0;
  fill_items(*net, ws, items, &tiles);
  MlpArgs a;
  memset(&a, 0, sizeof a);
  a.net = net_view(*net, false);
  a.B = B; a.do_fwd = 1; a.do_bwd = 1;
  a.x0 = x0; a.k0 = k0; a.x1 = x1; a.k1 = x1 ? k1 : 0;
  with_store(a, ws, true, true);
  a.seed_mode = SEED_PTR;
  const int nout = net->dims[net->n_layers];
  a.seed.p0 = dout; a.seed.ld0 = nout;
  if (dx) { a.dact_col0 = 0; a.dact_cols = net->dims[0]; a.dact = dx; a.lddact = net->dims[0]; }
  if (dx && net->dims[0] > kNarrowMax) { set_err("oprl_mlp_backward: dx supported for input dim <= %d", kNarrowMax); return OPRL_ERR_INVALID; }
  RC(launch(a, width, st));
  DwArgs dw;
  dw.items = items.data(); dw.n_items = (int)items.size(); dw.total_tiles = tiles; dw.B = B; dw.n_part = 1; dw.trace = nullptr; dw.use_row_scale = 0; dw.apply_only = 0; dw.apply_only = 0;
  memset(&dw.ad, 0, sizeof dw.ad);
  set_adam(dw.ad, 0.0, 0.9, 0.999, 1e-8, 0.0);
  set_step(dw.ad, 1); dw.ad.grad_scale = 1.0f; dw.ad.do_adam = 0;
  HIPC(launch_dw_prof(dw, st));
  return OPRL_OK;
}

extern "C" int oprl_adam_step(float* theta, float* m, float* v, const float* grad, int64_t n,
                              int32_t step, double lr, double beta1, double beta2, double eps,
                              double grad_scale, void* stream) {
  if (!theta || !m || !v || !grad || n < 1 || step < 1) { set_err("oprl_adam_step: invalid argument"); return OPRL_ERR_INVALID; }
  AdamScalars ad;
  memset(&ad, 0, sizeof ad);
  set_adam(ad, lr, beta1, beta2, eps, 0.0);
  set_step(ad, step); ad.do_adam = 1; ad.grad_scale = (float)grad_scale;
  HIPC(launch_adam_flat(theta, m, v, nullptr, grad, (long)n, ad, (hipStream_t)stream));
  return OPRL_OK;
}

extern "C" int oprl_polyak(float* target, const float* source, int64_t n, double tau, void* stream) {
  if (!target || !source || n < 1) { set_err("oprl_polyak: invalid argument"); return OPRL_ERR_INVALID; }
  HIPC(launch_polyak_flat(target, source, (long)n, tau, (hipStream_t)stream));
  return OPRL_OK;
}
