// learner_group.hip — packed learners (oprl_group_*): N independent learners stepped by one launch sequence.
// Split from learner.hip (round 4).
#include "learner_internal.h"

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
        if (h->epoch == 0) { h->epoch = 1; HIPC(hipMemsetAsync(h->y_granules, 0, ((size_t)4 * h->Bmax + 256) * sizeof(unsigned long long), st)); }
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

