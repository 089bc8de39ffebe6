// learner_internal.h — what the host-side translation units share: the learner's state (struct oprl_learner), the
// launchers' declarations, small layout helpers, and the internals of learner.hip that learner_create.hip,
// learner_dp.hip, learner_group.hip and learner_misc.hip call (namespace oprl_host; defined in learner.hip).
#pragma once
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

void set_err(const char* fmt, ...);
// HIP-event profiler (off by default; zero cost when off): learner.hip
void prof_fold();
void prof_begin(int kind, hipStream_t st);
void prof_end(hipStream_t st);

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
                                const PrefetchJob* prefetch, const LwPairBuf* pairs, bool second_done = false, bool* second_rode = nullptr,
                                const MlpArgs* bwd_rider = nullptr, bool* bwd_rode = nullptr, const DwKArgs* bwd_tiles = nullptr,
                                int bwd_tile_wgs = 0, bool* tiles_rode = nullptr);
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
bool xcd_map_ok();
bool fused_tile64_all();
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
namespace oprl_host {
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

inline int rccl_bind(Rccl& r, const char* path) {
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
}  // namespace oprl_host

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

namespace oprl_host {

inline long net_param_count(const oprl_net& n) {
  long c = 0;
  for (int l = 0; l < n.n_layers; ++l) c += (long)n.dims[l + 1] * n.dims[l] + n.dims[l + 1];
  return c;
}

inline long w_off(const oprl_net& n, int l) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += (long)n.dims[j + 1] * n.dims[j] + n.dims[j + 1];
  return c;
}
inline long b_off(const oprl_net& n, int l) { return w_off(n, l) + (long)n.dims[l + 1] * n.dims[l]; }

inline int check_net(const oprl_net& n, const char* name, int* width) {
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
inline long pack_off_fwd(const oprl_net& n, int l) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += 2 * pack_floats(n.dims[j + 1], n.dims[j]);
  return c;
}
inline long pack_off_bwd(const oprl_net& n, int l) { return pack_off_fwd(n, l) + pack_floats(n.dims[l + 1], n.dims[l]); }
inline long net_pack_floats(const oprl_net& n) { return pack_off_fwd(n, n.n_layers); }

// the same for the bf16 packs (library-owned, oprl_learner::pack16 / pack16_t), in floats (16-byte fragments)
// (pl = fp16 / bf16 planes per block: 1 for the bf16 packs, 2 for the PrecX2 packs — hi | lo)
inline long pack16_off_fwd(const oprl_net& n, int l, int pl = 1) {
  long c = 0;
  for (int j = 0; j < l; ++j) c += pl * (pack16_floats(n.dims[j + 1], n.dims[j]) + pack16_floats(n.dims[j], n.dims[j + 1]));
  return c;
}
inline long pack16_off_bwd(const oprl_net& n, int l, int pl = 1) { return pack16_off_fwd(n, l, pl) + pl * pack16_floats(n.dims[l + 1], n.dims[l]); }
inline long net_pack16_floats(const oprl_net& n, int pl = 1) { return pack16_off_fwd(n, n.n_layers, pl); }

inline Net net_view(const oprl_net& n, bool target) {
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
inline Net net_view16(const oprl_net& n, bool target, const float* pk16, int pl = 1) {
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

}  // namespace oprl_host

using namespace oprl_host;

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
  bool no_bwd_tiles = false;   // OPRL_AMD_NO_RIDE bit 512: the actor's dW + Adam tiles as a launch of their own (r06-18)
  bool no_bwd_ride = false;    // OPRL_AMD_NO_RIDE bit 256: TQC's actor backward as a launch of its own instead of riders of k_lw_dact (r06-16)
  bool no_p1_rows = false;     // OPRL_AMD_NO_P1_ROWS: TD3's exact-fp32 / bf16 merged launches carry no next-rows row (tests / A-B; r06-15)
  bool fin_l2_done = false;    // ... and the second hidden layer's forward rode on the target pass's heads behind the tail (r06-12); step 3 skips it too
  bool fin_done = false;       // TQC: the online critics' first hidden launch rode on the actor's forward on s' (critic_phase step 1); step 3 skips it
  bool no_fin_ride = false;    // OPRL_AMD_NO_RIDE bit 4: it stays the first launch of step 3 (tests / A-B)
  LwPairBuf lw_pairs = {nullptr, 0, 1u, 1 << 20, nullptr, 3, 0};   // k_lw_mid_pair: flags (own allocation), tags; OPRL_AMD_LW_PAIR: bit 0 forward, bit 1 backward pairs (default 3)
  float* lw_scratch = nullptr; // [critics][layers 1 .. L-1][Bmax x 512]: activations of forward-only layer-by-layer launches (the target pass) — not the nets' dW exchange buffers, which the early first launch has already filled
  MlpArgs rider;               // TQC: the actor's forward on s, prepared in critic_phase to ride on the critic step's head launch ...
  bool rider_pending = false;  // ... offered to the next for_each_net; taken: rider_done, and actor_phase skips its step 5
  bool rider_done = false;
  MlpArgs bwd_rider;           // TQC: the actor's backward, prepared in actor_phase to ride on the k_lw_dact launch whose rows it consumes (r06-16) ...
  bool bwd_rider_pending = false;   // ... offered to the for_each_net with the action gradients; taken: bwd_rider_done, and step 8 is skipped
  bool bwd_rider_done = false;
  DwKArgs bwd_tiles;           // ... and its dW + Adam tiles (the actor's step 9) behind it (r06-18): offered with the rider; taken: bwd_tiles_done
  int bwd_tile_wgs = 0;
  bool bwd_tiles_pending = false, bwd_tiles_done = false;
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
  int no_rt2 = 0;              // OPRL_AMD_NO_RT2: phase 1's B roles stay on 16-row slices in the over-subscribed launches (tests / A-B)
  int no_wide = 0;             // OPRL_AMD_NO_WIDE: never run role A / phase 2's critic pass on clusters of eight
  int no_merge_twin = 0;       // OPRL_AMD_NO_RIDE bit 64: TD3's critics' tiles as a launch of their own
  bool xcd_local = false;      // XCD-local cluster exchanges (DdpgArgs::xcd_local): probed dispatcher, not OPRL_AMD_NO_XCD_LOCAL, cleared by an expired wait
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
  // bf16 DDPG learners (bchain, round 5): whole updates per launch as well — k_ddpg_chain<PrecBF16>: the passes on the bf16
  // packs (coherent loads), the tiles with the exact-fp32 product writing both pack sets, the actor's unit-seed rows in
  // exact fp32 from the fp32 W^T pack; the workspace in uncached memory like the other chain learners'
  bool bchain = false;
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

// ---- internals of learner.hip the other host units call
namespace oprl_host {
const double* alpha_ptr(const oprl_learner* h);
void dev_free(void* p);
bool actor_due(const oprl_learner* h);
hipError_t uc_alloc(void** out, size_t bytes);
size_t net_ws_floats(const oprl_net& n, int B);
int fresh32(const oprl_net* net, hipStream_t st);
DdpgArgs ddpg_args(oprl_learner* h, int B);
int chain_rows(const oprl_learner* h, int B);
void build_repack_items(const oprl_net* const* nets, int n_nets, int which,
                        std::vector<RepackItem>& items, int* blocks_out,
                        float* const* pk16 = nullptr, float* const* pk16_t = nullptr, int pl = 1);
void alloc_net_ws(Pool& p, const oprl_net& n, int B, NetWs* ws);
bool use_fused(oprl_learner* h, int B);
void set_step(AdamScalars& ad, int step);
void set_adam(AdamScalars& ad, double lr, double beta1, double beta2, double eps, double tau);
int next_tp_tag(unsigned* counter, unsigned long long* xbuf, size_t xbuf_bytes, hipStream_t st, unsigned* out);
const oprl_net& eff(const oprl_learner* h, const oprl_net& n);
int check_device_error(const oprl_learner* h);
void with_store(MlpArgs& a, const NetWs& ws, bool x, bool dy);
hipError_t launch_dw_prof(const DwArgs& a, hipStream_t st);
void fill_items(const oprl_net& n, const NetWs& ws, std::vector<DwItem>& v, int* tiles, bool small_partial_tiles = false,
                float* pk16 = nullptr, float* pk16_t = nullptr, int pl = 1);
DwArgs dw_build(oprl_learner* h, bool critic, int B, bool polyak, bool with_alpha);
bool alpha_rides(const oprl_learner* h);
int launch(const MlpArgs& a0, int width, hipStream_t st);
// learners with lazily maintained fp32 packs (fresh32)
extern std::mutex g_lazy_mu;
extern std::vector<oprl_learner*> g_lazy;
int repack_nets(const oprl_net* const* nets, int n_nets, int which, hipStream_t st,
                float* const* pk16 = nullptr, float* const* pk16_t = nullptr, int pl = 1);
// step_n's K-loop as launches of several updates each (k_ddpg_chain); also the inline data-parallel loop
int chain_loop(oprl_learner* h, int K, int B, float* (*set)[5], void* stream);
bool chain_ok(oprl_learner* h, int B);
}  // namespace oprl_host
using namespace oprl_host;

