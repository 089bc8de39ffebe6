// learner.hip — host orchestration of update() for DDPG / TD3 / SAC / TQC and the
// C-ABI of include/oprl_amd.h.  Each update() is a short fixed sequence of
// k_mlp_slice / k_dw_adam launches on the caller's stream; no host sync inside.
//
// Order of operations follows the reference exactly (it matters: the actor
// loss sees the post-Adam critic, Polyak sees both updated nets):
//   DDPG  algos/ddpg.py:61-107      TD3  algos/td3.py:71-146
//   SAC   algos/sac.py:75-155       TQC  algos/tqc.py:116-189
#include "learner_internal.h"

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

}  // namespace oprl

namespace oprl_host {

// learners with lazily maintained fp32 packs, by pack pointer (oprl_mlp_* know a net, not its learner)
// Whole-update launches (k_ddpg_chain) take the whole chip for up to 32 updates.  Learners of one process that launch them
// from DIFFERENT STREAMS (one host thread per learner: the multi-seed layout) take TURNS: a launch waits for the event
// behind the last whole-update launch of another stream — two such launches side by side would only hold each other's
// compute units with waiting workgroups (bounded waits would expire).  Nothing is recorded until a second stream shows
// up (a process whose learners share one stream — the headline path, bench.py's blocks — pays no event, no barrier
// packet); the first launch from a second stream drains the device on the host, once.
// Limits (deliberate): only the WHOLE-UPDATE launches take turns — another learner's merged / phase launches, or
// any other kernel of the process on a second stream, can still hold compute units while a chain launch waits for its
// own workgroups: the bounded wait then expires, the call reports it, and after clear_error the learner runs the
// shared-chip forms (tested: tests/test_gpu_errors.py); a learner that is KNOWN to share the GPU says so up front
// (set_cluster(4), OPRL_AMD_FORM=plain).  Streams are told apart by handle and devices by id & 15; the one
// hipDeviceSynchronize (the first launch from a second stream) runs under the mutex, once per process and device.
struct ChipTurn {
  std::mutex mu;
  hipEvent_t ev[16] = {};
  hipStream_t stream[16] = {};
  bool any[16] = {}, multi[16] = {}, rec[16] = {};
};
ChipTurn g_turn;

hipError_t chip_turn_begin(hipStream_t st, int* dev_out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 15;
  *dev_out = dev;
  if (!g_turn.any[dev] || g_turn.stream[dev] == st) return hipSuccess;
  if (!g_turn.multi[dev]) {                       // the second stream of this process: from now on events
    g_turn.multi[dev] = true;
    return hipDeviceSynchronize();                // (once; the other stream's handle may be gone by now)
  }
  return g_turn.rec[dev] ? hipStreamWaitEvent(st, g_turn.ev[dev], 0) : hipSuccess;
}
void chip_turn_end(hipStream_t st, int dev) {
  g_turn.any[dev] = true;
  g_turn.stream[dev] = st;
  g_turn.rec[dev] = false;
  if (!g_turn.multi[dev]) return;
  if (g_turn.ev[dev] == nullptr && hipEventCreateWithFlags(&g_turn.ev[dev], hipEventDisableTiming) != hipSuccess) { g_turn.ev[dev] = nullptr; return; }
  if (hipEventRecord(g_turn.ev[dev], st) == hipSuccess) g_turn.rec[dev] = true;
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

void fill_items(const oprl_net& n, const NetWs& ws, std::vector<DwItem>& v, int* tiles, bool small_partial_tiles,
                float* pk16, float* pk16_t, int pl) {
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
      const bool second_done = first_done && h->fin_l2_done;
      if (first_done) h->fin_l2_done = false;
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
                                          tail, h->nc, tail0, h->fin16 ? (h->x2 ? 2 : 1) : 0, pf, &h->lw_pairs, second_done,
                                          tail != nullptr ? &h->fin_l2_done : nullptr,
                                          (h->bwd_rider_pending && h->multi_args[0].do_bwd && h->multi_args[0].dact_cols > 0) ? &h->bwd_rider : nullptr,
                                          &h->bwd_rider_done, h->bwd_tiles_pending ? &h->bwd_tiles : nullptr, h->bwd_tile_wgs, &h->bwd_tiles_done);
      h->bwd_rider_pending = false;
      h->bwd_tiles_pending = false;
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
  return 16 + chain_tile_rows(mt, sl);
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
  a.seed2_granules = h->y_granules + (size_t)3 * h->Bmax + 256;
  a.merged = 0;
  a.epoch = h->epoch;
  a.trace = nullptr;
  a.nc = h->nc_cluster(B);
  a.no_lean = h->no_lean;
  a.xcd_local = h->xcd_local ? 1 : 0;
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
  if (h->xnc >= 8 && !h->no_wide && !a.sac && (!a.bf16 || h->bchain) && a.A <= 8 && fused_ddpg_is_lean(a)) {   // (narrow exchanges: <= 8 action columns)
    const int slices = (B + kR - 1) / kR;
    if (h->nc == 1 && (a.nc + 8 + a.nc) * slices <= h->n_cus) a.wide |= 1;
    if ((8 + 1) * slices <= h->n_cus) a.wide |= 2;
  }
  // merged launches (DDPG, lean passes, one 256-row chunk, this rank's own Adam step): the critic's dW tiles ride
  // on phase 1 — whose role A then stays on a cluster of four: 64 CUs must be free for tile workgroups from the start
  // (a gradient-exporting learner — data parallel over RCCL — merges only with the PrecX2 tiles, which know how to leave
  // dW in the gradient arena instead of running Adam: four launches per data-parallel update instead of six)
  // exact-fp32 learners with mirrored packs: the whole update as k_ddpg_chain<PrecF32> when that form is possible at all
  // (there is no merged phase 2 with the fp32 tiles on its own: with the whole form out of reach the two bits below stay
  // what they were — merged phase 1 with role A on four, phase 2 and the actor's dW as launches)
  // (a gradient-exporting learner takes the whole form only when its tiles exchange the gradients themselves: dp_inline)
  const bool dp_whole_ok = !h->cfg.export_grads || h->dp_inline;
  const bool whole_f32 = h->fchain && fused_x2_tiles() && !a.x2 && !a.bf16 && !h->no_whole && !h->no_merge && !h->no_merge2 && !h->shared_chip &&
                         dp_whole_ok && B <= 256 && fused_ddpg_is_lean(a) && (a.wide & 3) == 3 && h->chain_flags != nullptr && h->du_granules != nullptr &&
                         chain_rows(h, B) * ((B + kR - 1) / kR) <= h->n_cus;
  // ... and bf16 learners (k_ddpg_chain<PrecBF16>: learner_internal.h bchain)
  const bool whole_bf16 = h->bchain && fused_x2_tiles() && a.bf16 && !h->no_whole && !h->no_merge && !h->no_merge2 && !h->shared_chip &&
                          dp_whole_ok && B <= 256 && fused_ddpg_is_lean(a) && (a.wide & 3) == 3 && h->chain_flags != nullptr && h->du_granules != nullptr &&
                          chain_rows(h, B) * ((B + kR - 1) / kR) <= h->n_cus;
  // (dp_inline: the gradient exchange inside the dW tiles — the 16 x 32 tiles of k_dw_adam<true> as launches of their own,
  // or the 16 x 64 tiles of the merged / whole-update launches themselves: dw_tile_x2.h — PrecX2 learners in both forms,
  // exact-fp32 and bf16 learners in the whole-update form, round 6)
  const bool inline_tiles = h->dp_inline && fused_x2_tiles() && h->nc == 1 && (a.x2 || whole_f32 || whole_bf16);
  const bool xport_ok = !h->cfg.export_grads || (a.x2 && fused_x2_tiles()) || inline_tiles;
  // (TD3: both critics' tiles ride — roles A | B1 | B2 | C are the whole chip at B = 256, the 2 x 84 / 2 x 152 tiles take the
  // compute units the roles leave; this rank's own Adam step only)
  const bool merge_twin = h->nc == 2 && c.algo == OPRL_TD3 && !h->cfg.export_grads && !h->dp_inline && !h->no_merge_twin;
  if (h->bchain && !whole_bf16) a.wide = 0;      // (outside the whole form a bf16 learner's passes stay on clusters of four, as before)
  if (whole_bf16) a.actor_pb1_f32 = net_view(c.actor, false).pb[1];
  if (!h->no_merge && !h->shared_chip && (h->nc == 1 || merge_twin) && !a.sac && B <= 256 && xport_ok && (!h->dp_inline || inline_tiles) && fused_ddpg_is_lean(a)) {
    a.merged |= 1;
    if (!(a.x2 && fused_x2_tiles()) && !whole_f32 && !whole_bf16) a.wide &= ~1;     // (the 84 16 x 64 tiles of a PrecX2 learner get along with role A on eight)
  }
  // ... and the ACTOR's tiles on phase 2 (DDPG / TD3: the tanh head, action_dim <= kDuLd): the tiles form their dY from
  // du, the first layer's comes from one more backward step of the critic pass's members (csrc/fused_ddpg.hip).
  // PrecX2 learners only, the pass on clusters of eight: with the exact-fp32 tiles the merged form measured no faster
  // than the two launches (34.9 vs 34.7 us)
  if (!h->no_merge2 && !h->shared_chip && ((a.x2 && fused_x2_tiles()) || whole_f32 || whole_bf16) && h->du_granules != nullptr && !a.sac && B <= 256 && (!h->dp_inline || inline_tiles) &&
      fused_ddpg_is_lean(a) && c.actor.theta_target != nullptr && (a.wide & 2) != 0) {
    a.merged |= 2;
    a.du_granules = h->du_granules;
    a.g1_granules = h->g1_granules;
    a.w3_snap = h->w3_snap;
  }
  // the whole update as ONE launch (k_ddpg_update): both merged forms, role A and the critic pass on eight, the 16 x 64
  // tiles, and everything the roles hand to each other in uncached memory
  if (!h->no_whole && (!h->cfg.export_grads || inline_tiles) && ((a.x2 && fused_x2_tiles()) || whole_f32 || whole_bf16) && h->nc == 1 && (a.merged & 3) == 3 && (a.wide & 3) == 3 && h->uc_pool &&
      (h->uc_base != nullptr || h->bchain) && h->w_flags != nullptr && h->chain_flags != nullptr &&
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
  // Over-subscribed plain phase-1 launches (B >= 512: the roles' clusters need more compute units than the chip has, the
  // launch is several dispatch rounds): the B roles carry TWO row tiles per cluster — half the workgroups, one fetch of the
  // critic's fragments per 32 rows (tp4.h tp4_scalar_fb2; bit-identical to the one-tile form).  Slices a multiple of 16: a
  // cluster's members then sit on one XCD (fused_ddpg.hip role_b2).
  {
    const int slices = (B + kR - 1) / kR;
    // (twin critics only — TD3, SAC: their two B roles then share the first dispatch round; measured, r06-6: SAC humanoid
    // B = 1024 - 3.7 us per update in every mode, TD3 B = 512 - 4.7, SAC B = 512 - 5; a single critic's one B role on half the
    // workgroups only lengthens the wait of role A behind it: DDPG B = 512 + 1.8 us, B = 1024 + 2.4 .. 4)
    a.rt2 = (h->no_rt2 != 1 && h->nc == 2 && a.merged == 0 && (a.wide & 1) == 0 && a.nc == 4 && fused_ddpg_is_lean(a) && (slices & 15) == 0 &&
             (2 + h->nc) * 4 * slices > h->n_cus && !a.prefetch_p1) ? 1 : 0;
    // SAC there: role A's first pass — the online actor on s' — also carries role C's pass, the same actor on s (tp4_forward2;
    // bit-identical): role C's dispatch round is not launched (r06-13; OPRL_AMD_NO_RT2=2 keeps role C)
    // (where role A alone fills the chip — humanoid B = 1024: -3.1 us x2, -2.1 f32 / bf16 per update; at B = 512 role C ran
    // beside role A's second half and the merged pass is 3.5 us SLOWER: a tile's pass is bound by instruction issue, not by
    // its fragments, so the second tile costs a pass's 6.4 us, not the 2 - 3 us the B roles' shared fragments suggested)
    if (a.rt2 == 1 && a.sac && !a.twin_split && a.do_actor && h->no_rt2 == 0 && 4 * slices >= h->n_cus) a.rt2 = 2;
  }
  // (an exchanging rank outside PrecX2 has the in-tile exchange in the whole-update form only: whatever kept that form
  // away, its merged phase launch — 16 x 32 tiles that exchange nothing — must not run either)
  if (h->dp_inline && !a.x2 && !a.whole) a.merged = 0;
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
      HIPC(hipMemsetAsync(h->y_granules, 0, ((size_t)4 * h->Bmax + 256) * sizeof(unsigned long long), st));
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
      // (what this block advances — the optimisers' step counts, the exchange sequence numbers, the tags — is put back by
      // every error return below: after a failed call Adam's bias corrections and, data parallel, this rank's tile sequence
      // must not be ahead of the parameters / of the peers: ADVICE r4)
      const struct { int c, a; unsigned long long seq; unsigned tag, epoch; } snap{h->opt_step_critic, h->opt_step_actor, h->p2p.tile_seq, h->tp_tag, h->epoch};
      auto undo = [&]() { h->opt_step_critic = snap.c; h->opt_step_actor = snap.a; h->p2p.tile_seq = snap.seq; h->tp_tag = snap.tag; h->epoch = snap.epoch; };
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
        if (fill_dw_kargs(dw, &kd, 64) < 0 || dw.n_items > kDwFusedItems) { undo(); set_err("whole update: bad critic dW table"); return OPRL_ERR_INVALID; }
        kd.gate.rows = fa.gate_flags; kd.gate.n_rows = 4 * slices;
        kd.gate.seed = fa.y_granules; kd.gate.n_seed = B;
        kd.gate.late_dY = h->ws_critic[0].dY[c.critics[0].n_layers - 1];
        kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
        kd.gate.err = h->err_dev; kd.gate.err_code = (1u << 8) | 7u;
        kd.gate.done = fa.ct_done; kd.gate.what_if = h->debug_expire;
        fa.n_ct = kd.tile_end[kDwMaxItems - 1];
        if (fa.n_ct > 192 - 64) { undo(); set_err("whole update: too many critic tiles"); return OPRL_ERR_INVALID; }
        compact(kd, &kc);
      }
      {
        DwArgs dw = dw_build(h, false, B, true, false);
        if (fill_dw_kargs(dw, &kd, 64) < 0 || dw.n_items != 3) { undo(); set_err("whole update: bad actor dW table"); return OPRL_ERR_INVALID; }
        kd.gate.seed = fa.du_granules; kd.gate.n_seed = B;
        kd.gate.tag = h->epoch; kd.gate.spin = h->debug_expire == 7 ? 0 : (1 << 20);
        kd.gate.err = h->err_dev; kd.gate.err_code = (2u << 8) | 7u;
        kd.gate.kind[0] = 2; kd.gate.kind[1] = 1; kd.gate.kind[2] = 0; kd.gate.kind[3] = 0; kd.gate.what_if = h->debug_expire;
        kd.gate.h2 = h->ws_actor.X[2];
        kd.gate.w3 = fa.w3_snap;
        kd.gate.g1 = fa.g1_granules;
        kd.gate.n_act = h->A;
        compact(kd, &ka);
      }
      // (every tile must find a role-B / role-C workgroup to be the continuation of: 8 per slice)
      const int max_tiles = kc.tile_end[kDwFusedItems - 1] > ka.tile_end[kDwFusedItems - 1] ? kc.tile_end[kDwFusedItems - 1] : ka.tile_end[kDwFusedItems - 1];
      const int t_rows = chain_tile_rows(max_tiles, slices);
      // (one update's workgroups — 16 role rows per slice + the tile-only rows — wait for each other: all must fit the chip)
      const bool chain_fits = (16 + t_rows) * slices <= h->n_cus;
      const int U = (h->no_chain || h->chain_flags == nullptr || !chain_fits) ? 1 : h->chain_u;
      memset((void*)&kc.xchg, 0, sizeof kc.xchg);
      memset((void*)&ka.xchg, 0, sizeof ka.xchg);
      if (h->dp_inline) {
        // data parallel on peer windows: the tiles of this launch all-reduce their gradients themselves and run Adam on
        // the mean (dw_tile_x2.h): exchange sequence numbers seq0 + 2 u (critic) / + 1 (actor) for update u of the launch
        P2pState& P = h->p2p;
        if (max_tiles > h->p2p_max_tiles) { undo(); set_err("data-parallel whole update: the windows hold %d tiles, the launch has %d", h->p2p_max_tiles, max_tiles); return OPRL_ERR_STATE; }
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
        { static const int order = [] { const char* e = getenv("OPRL_AMD_CHAIN_ORDER"); return e != nullptr ? atoi(e) & 3 : 0; }(); ca.order = order; }
        ca.c_step[0] = kc.ad.step_size_host; ca.c_bc2[0] = kc.ad.bc2_sqrt_host;
        ca.a_step[0] = ka.ad.step_size_host; ca.a_bc2[0] = ka.ad.bc2_sqrt_host;
        for (int u = 1; u < U; ++u) {          // (the optimisers' step counts advance once per update and net, as dw_build does;
          h->opt_step_critic += 1;             // only Adam's bias-correction terms change from update to update: a call of
          h->opt_step_actor += 1;              // dw_build per update and net was 8 of the 17 us a step_n(20) call took to enqueue)
          AdamScalars sc_, sa_;
          memset(&sc_, 0, sizeof sc_); memset(&sa_, 0, sizeof sa_);
          set_adam(sc_, c.hp.lr_critic, c.hp.beta1, c.hp.beta2, c.hp.adam_eps, c.hp.tau); set_step(sc_, h->opt_step_critic);
          set_adam(sa_, c.hp.lr_actor, c.hp.beta1, c.hp.beta2, c.hp.adam_eps, c.hp.tau); set_step(sa_, h->opt_step_actor);
          ca.c_step[u] = sc_.step_size_host; ca.c_bc2[u] = sc_.bc2_sqrt_host;
          ca.a_step[u] = sa_.step_size_host; ca.a_bc2[u] = sa_.bc2_sqrt_host;
        }
        ca.set0[0] = fa.src.s; ca.set0[1] = fa.src.a; ca.set0[2] = fa.src.r; ca.set0[3] = fa.src.d; ca.set0[4] = fa.src.s2;
        for (int i = 0; i < 5; ++i) ca.set1[i] = h->chain_set1[i];
        if ((U > 1 || ca.pf_last) && ca.set1[0] == nullptr) { undo(); set_err("chain launch: no second staging set"); return OPRL_ERR_STATE; }
        ca.ct_fin = h->chain_flags; ca.at_fin = h->chain_flags + 192; ca.pf_done = h->chain_flags + 384;
        for (int w = 0; w < 4; ++w)
          for (int l = 0; l < kMaxLayers; ++l) ca.b16[w][l] = h->chain_b16 + ((size_t)w * kMaxLayers + l) * 256;
        ca.w3buf[0] = h->w3_snap; ca.w3buf[1] = h->w3buf1;
        // the first hidden layer's dY from the unit-seed rows the pass's members leave before the pass (DwGate kind 3)
        ca.qp = h->chain_flags + 576;
        fa.gu = h->gu; fa.gu_flags = h->chain_flags + 448;
        ka.gate.kind[0] = 3; ka.gate.gu = h->gu; ka.gate.gu_flags = fa.gu_flags; ka.gate.n_gu_flags = 8 * slices;
        if (kc.tile_end[kDwFusedItems - 1] > 192 || ka.tile_end[kDwFusedItems - 1] > 192 || slices > 64) { undo(); set_err("chain launch: too many tiles"); return OPRL_ERR_INVALID; }
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
        if (e != hipSuccess) undo();
        HIPC(e);
        h->whole_done = true;
        return OPRL_OK;
      }
      undo();
      set_err("whole update: one update's workgroups do not fit this device (%d compute units)", h->n_cus);
      return OPRL_ERR_STATE;
    }
    if ((fa.merged & 1) != 0) {
      // phase 1 and the critic's dW + Adam tiles as ONE launch: the tiles wait for the roles' flag granules
      // TD3 moves its targets only on actor steps (td3.py:135-146)
      DwArgs dw = dw_build(h, true, B, c.algo == OPRL_TD3 ? actor_due(h) : true, false);
      DwKArgs kd;
      if (fill_dw_kargs(dw, &kd, ((fa.x2 && fused_x2_tiles()) || (h->nc == 2 && fused_tile64_all())) ? 64 : 32) < 0) { set_err("merged phase 1: bad dW table"); return OPRL_ERR_INVALID; }
      const int slices = (B + kR - 1) / kR;
      kd.gate.rows = fa.gate_flags; kd.gate.n_rows = 4 * slices * h->nc;
      kd.gate.seed = fa.y_granules; kd.gate.n_seed = B;      // the seeds come as granules, one per row
      kd.gate.late_dY = h->ws_critic[0].dY[c.critics[0].n_layers - 1];
      if (h->nc == 2) {      // (twin critics: the second one's items, seeds and output-layer dY)
        kd.gate.item_split = c.critics[0].n_layers;
        kd.gate.seed2 = fa.seed2_granules;
        kd.gate.late_dY2 = h->ws_critic[1].dY[c.critics[1].n_layers - 1];
      }
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
    h->fin_l2_done = false;
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
    else if (h->trace != nullptr && h->nc == 2) fa.trace = h->trace + (size_t)6 * 64 * kTraceStamps * 2;      // twin critics: roles 0 .. 3 are phase 1's, slot 6
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
  const int n_q_all = (algo == OPRL_TD3 || algo == OPRL_DDPG) ? 1 : nc;   // TD3 uses Q1 only
  auto actor_backward_args = [&]() {      // step 8's launch
    MlpArgs f = base_args(h, c.actor, false, B);
    f.do_bwd = 1;
    with_store(f, h->ws_actor, true, true);
    SeedArgs& sd = f.seed;
    if (gauss) {
      f.seed_mode = SEED_GAUSS;
      sd.p0 = h->da; sd.ld0 = A; sd.n_da = n_q_all; sd.da_stride = (long)h->Bmax * A;
      sd.p1 = h->raw;
      seed_rng(f, h, noise1, 2);   // backward re-reads (or re-draws) the forward's eps
      sd.log_alpha = alpha_ptr(h); sd.alpha_const = (float)c.hp.alpha_init;
      sd.cval = 1.0f / (float)B;
    } else {
      f.seed_mode = SEED_TANH;
      sd.p0 = h->da; sd.ld0 = A; sd.p1 = h->pi;
    }
    return f;
  };
  DwArgs actor_dw;
  bool actor_dw_built = false;
  auto actor_dw_args = [&]() {            // step 9's launch (advances the actor's and — riding — the temperature's step counts: once per update)
    h->opt_step_actor += 1;
    DwArgs dw;
    dw.items = h->items_host.data() + h->n_items_critic; dw.n_items = h->n_items_actor;
    dw.total_tiles = h->tiles_actor; dw.B = B; dw.n_part = tp_generic(h, c.actor, B) ? 4 : 1; dw.use_row_scale = 0; dw.apply_only = 0;
    dw.dy_tiled = dw.n_part > 1 ? 1 : 0;
    dw.trace = h->trace != nullptr ? h->trace + (size_t)5 * 64 * kTraceStamps * 2 : nullptr;   // slot 5
    dw.ad = adam_scalars(h, c.hp.lr_actor, h->opt_step_actor, c.actor.theta_target != nullptr, 1.0f);
    if (alpha_rides(h)) dw.alpha = alpha_job(h, B);
    return dw;
  };
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
    // TQC: the actor's backward (step 8) consumes the action gradients the launch sequence below ends with (k_lw_dact):
    // offered as a rider of that launch (r06-16; the tag its launch would draw, now)
    h->bwd_rider_pending = false;
    h->bwd_rider_done = false;
    if (algo == OPRL_TQC && gauss && !c.export_grads && !h->no_bwd_ride) {
      MlpArgs f = actor_backward_args();
      if (f.tp_xbuf != nullptr && mlp_slice_tp_shape_ok(f, h->w_actor)) {
        RC(next_tp_tag(f.tp_tag_counter, f.tp_xbuf, f.tp_xbuf_bytes, st, &f.tp_tag));
        h->bwd_rider = f;
        h->bwd_rider_pending = true;
        // ... and step 9's tiles behind it (the 16 x 32 tiles of k_dw_adam, the temperature's step as the workgroup one past
        // them): built with THIS update's step count; if the launch does not take them, step 9 launches the same table
        h->bwd_tiles_pending = false;
        h->bwd_tiles_done = false;
        if (!h->no_bwd_tiles && alpha_rides(h) && h->n_items_actor <= kDwMaxItems) {
          actor_dw = actor_dw_args();
          actor_dw_built = true;
          const int total = fill_dw_kargs(actor_dw, &h->bwd_tiles, 32);
          if (total > 0) {
            h->bwd_tile_wgs = total + (actor_dw.alpha.log_alpha != nullptr ? 1 : 0);
            h->bwd_tiles_pending = true;
          }
        }
      }
    }
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
  // 8. actor backward from the stored activations — unless it rode on the k_lw_dact launch above
  h->bwd_rider_pending = false;
  if (h->bwd_rider_done) {
    h->bwd_rider_done = false;
  } else {
    RC(launch(actor_backward_args(), h->w_actor, st));
  }
  // 9. dW + Adam (+ Polyak of the actor target for DDPG / TD3) — unless the tiles rode behind the riding backward
  h->bwd_tiles_pending = false;
  if (h->bwd_tiles_done) {
    h->bwd_tiles_done = false;
  } else {
    if (!actor_dw_built) actor_dw = actor_dw_args();
    HIPC(launch_dw_prof(actor_dw, st));
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
                        float* const* pk16, float* const* pk16_t, int pl) {
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
                float* const* pk16, float* const* pk16_t, int pl) {
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
                               "hand-over inside a layer-by-layer launch (a producing workgroup — first hidden layer, riding tail, action-gradient rows — never flagged its rows)"};
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

}  // namespace oprl_host

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
    // (an expired wait of any kind: the cluster exchanges go back to agent-scope publishes as well — if a member sat on an
    // XCD the others did not expect, the demoted forms must not repeat it)
    if (code != 0 && w != 9 && !h->debug_expire) h->xcd_local = false;
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
  if (!h || site < 0 || (site > 8 && (site < 101 || site > 106 || site == 103))) { set_err("oprl_learner_debug_expire: invalid argument"); return OPRL_ERR_INVALID; }
  // (101 .. 106 are TIMING experiments — a cross-workgroup wait counts as satisfied, the update computes wrong parameters and
  // reports nothing: refused unless the process opted in, tools/what_if.py sets the variable)
  if (site > 8) {
    const char* e = getenv("OPRL_AMD_WHAT_IF");
    if (e == nullptr || atoi(e) == 0) { set_err("oprl_learner_debug_expire: sites 101..106 are timing experiments that corrupt the update; set OPRL_AMD_WHAT_IF=1 to allow them"); return OPRL_ERR_STATE; }
  }
  h->debug_expire = site;
  h->lw_pairs.spin = site == 8 ? 0 : (1 << 20);      // (8: the hand-over inside k_lw_mid_pair)
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
    if (h->fchain) h->stale32[0] = true;      // (the launch writes the mirrors: the caller's packs fall behind)
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
    if (h->fchain) h->stale32[1] = true;
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

namespace oprl_host {
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
}  // namespace oprl_host

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
    // (... and TD3's merged twin launches in every arithmetic, r06-15: without the merged phase 2 — exact fp32, bf16 — a
    // critic-only update had no launch that staged the next rows, and the update behind it gathered its own inside the roles'
    // first stage: 4.8 us against 2.9 before the first barrier of every role of every second launch)
    const bool td3_p1 = h->cfg.algo == OPRL_TD3 && (probe.merged & 1) != 0 && !h->no_p1_rows;
    h->prefetch_p1 = !chain && alt != nullptr && ((probe.merged & 2) != 0 || td3_p1);
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

