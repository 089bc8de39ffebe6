// learner_create.hip — oprl_learner_create / destroy / sync_params and the pack helpers of the C-ABI: the workspace
// pool, the uncached areas, the tile and repack tables.  Split from learner.hip (round 4).
#include "learner_internal.h"

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
  floats += 64 * 32 + 8 * (size_t)B + 512;      // (granule arrays: y, q1, q2, the twin's seeds; 256 gate flags)
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
  // (gradient-exporting learners too — data-parallel ranks: on peer windows their update IS the whole-update launch, the
  // gradient exchange inside its tiles, in every arithmetic; over RCCL they run the phase / apply launches on the same
  // mirrors and the same uncached workspace)
  h->fchain = h->fused && !h->x2 && !h->bf16 && cfg->algo == OPRL_DDPG && nc == 1 && merge2_bufs &&
              cfg->actor.theta_target != nullptr && cfg->critics[0].theta_target != nullptr && cfg->actor.n_layers == 3 && cfg->critics[0].n_layers == 3;
  h->bchain = h->fused && h->bf16 && cfg->algo == OPRL_DDPG && nc == 1 && merge2_bufs &&
              cfg->actor.theta_target != nullptr && cfg->critics[0].theta_target != nullptr && cfg->actor.n_layers == 3 && cfg->critics[0].n_layers == 3 &&
              getenv("OPRL_AMD_NO_BF16_CHAIN") == nullptr;
  const int uc_pool = (h->x2 || h->fchain || h->bchain) ? 1 : 0;
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
  h->y_granules = p.take<unsigned long long>((size_t)4 * B + 256);      // [TD target / seeds | q1 | q2 | 256 gate flags | the twin critic's seeds]
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
    // behind the actor's forward, 8 next rows on k_lw_dact, 16 wide dW kernel (kernels.hip), 32 hidden-layer pairs,
    // (64: TD3's twin tiles, below) 128 the online critics' second hidden layer behind the tail on the target heads (r06-12),
    // 256 the actor's backward on the k_lw_dact launch (r06-16), 512 its dW + Adam tiles behind it (r06-18)
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
    h->no_bwd_ride = (no_ride & 256) != 0;
    h->no_bwd_tiles = (no_ride & 512) != 0;
    { const char* e = getenv("OPRL_AMD_NO_P1_ROWS"); h->no_p1_rows = e != nullptr && atoi(e) != 0; }
    if (cfg->algo == OPRL_TQC && h->w_critic == 512) {
      const size_t n = (size_t)nc * (kMaxLayers - 1) * (size_t)h->Bmax * 512;
      if (hipMalloc(&h->lw_scratch, n * sizeof(float)) != hipSuccess) h->lw_scratch = nullptr;   // (then: the nets' own buffers, no early launch)
      {
        const int pair_env = (no_ride & 32) != 0 ? 0 : (31 & ~((no_ride & 128) != 0 ? 4 : 0) & ~((no_ride & 256) != 0 ? 8 : 0) & ~((no_ride & 512) != 0 ? 16 : 0));
        const int nf = kMaxMulti * ((h->Bmax + 31) / 32) * 32;
        void* fl = nullptr;
        if (pair_env != 0 && hipMalloc(&fl, (size_t)nf * sizeof(unsigned long long)) == hipSuccess) {
          (void)hipMemset(fl, 0, (size_t)nf * sizeof(unsigned long long));
          h->lw_pairs.flags = (unsigned long long*)fl; h->lw_pairs.n_flags = nf; h->lw_pairs.use = pair_env & 31;
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
    h->no_merge_twin = (no_ride & 64) != 0 ? 1 : 0;
    { const char* e = getenv("OPRL_AMD_NO_RT2"); h->no_rt2 = e != nullptr ? atoi(e) : 0; if (h->no_rt2 < 0 || h->no_rt2 > 2) h->no_rt2 = 1; }   // 1: one row tile everywhere; 2: the B roles' two tiles, but SAC's role C stays a role of its own
    const char* nxl = getenv("OPRL_AMD_NO_XCD_LOCAL");
    h->xcd_local = h->fused && !(nxl != nullptr && atoi(nxl) != 0) && xcd_map_ok();
    // the generic per-net launches on clusters of 4 (slice_tp.hip): any net of the common shape
    // (decided per net by tp_generic(): TQC's 512-wide critics stay on k_mlp_slice, its actor moves)
    h->tp_generic_on = !h->no_lean && h->ncl == 4;
  }
  if (h->fused || h->tp_generic_on) {
    const size_t slices = (size_t)(h->Bmax + kR - 1) / kR;
    // (areas laid out for clusters of eight where wide clusters may run: DDPG / TD3, fp32, lean passes)
    h->xnc = (h->fused && (!h->bf16 || h->bchain) && !h->no_lean && !h->no_wide && h->ncl == 4 &&
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

