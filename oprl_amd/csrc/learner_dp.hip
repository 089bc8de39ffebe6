// learner_dp.hip — data parallel: the run-time binding of RCCL (oprl_comm_*), the peer windows (oprl_p2p_*; csrc/p2p.hip),
// and the data-parallel update / K-loop (oprl_learner_dp_update / dp_step_n).  Split from learner.hip (round 4).
#include "learner_internal.h"

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

