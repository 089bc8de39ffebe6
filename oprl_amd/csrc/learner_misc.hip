// learner_misc.hip — scalars, counters, traces and debug views of a learner, and the stand-alone building blocks of
// the C-ABI (oprl_mlp_forward / act / backward, oprl_adam_step, oprl_polyak).  Split from learner.hip (round 4).
#include "learner_internal.h"

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

// Which LAUNCH FORM would an update of batch size B take right now?  The selection (learner.hip ddpg_args / critic_phase:
// a dozen interacting conditions — precision, algorithm, batch size, cluster size, gradient export, the data-parallel
// exchange, a shared chip, the environment switches) as numbers a test can hold against a table
// (tests/test_gpu_forms.py), so that a mode falling off its fast form is a red test and not a line in a benchmark.
//   out[0]  fused: 1 the fused phase kernels (DDPG / TD3 / SAC), 0 the generic launch sequence (TQC, no_fuse, odd shapes)
//   out[1]  lean: 1 tp4.h's passes, 0 the generic tp3.h passes
//   out[2]  form: 4 whole updates per launch (k_ddpg_chain), 3 both merged launches (phase 1 + critic tiles | phase 2 +
//           actor tiles), 2 merged phase 1 only, 1 the plain phase launches + dW launches, 0 not fused
//   out[3]  updates per chain launch step_n may run (1 unless form 4)
//   out[4]  wide bits (1: role A on clusters of eight, 2: the critic pass)      out[5]  cluster size of the other roles
//   out[6]  twin_split    out[7]  p2_pair    out[8]  arithmetic of the kernels: 0 exact fp32, 1 bf16, 2 x2
//   out[9]  XCD-local cluster exchanges    out[10] the learner was demoted to the shared-chip forms
//   out[11] gradient-exporting learners: the form (as out[2]) of a data-parallel update whose exchange runs inside the dW
//           tiles (peer windows, level 2) — 4 = the rank runs the single-GPU whole-update launch; others: 0
extern "C" int oprl_learner_debug_form(oprl_learner* h, int32_t B, int32_t* out) {
  if (!h || !out || B < 1 || B > h->Bmax) { set_err("oprl_learner_debug_form: invalid argument"); return OPRL_ERR_INVALID; }
  for (int i = 0; i < 12; ++i) out[i] = 0;
  out[10] = h->shared_chip ? 1 : 0;
  out[8] = h->x2 ? 2 : (h->bf16 ? 1 : 0);
  if (!use_fused(h, B)) return OPRL_OK;
  const DdpgArgs a = ddpg_args(h, B);
  out[0] = 1;
  out[1] = fused_ddpg_is_lean(a) ? 1 : 0;
  const bool whole = a.whole && B <= 256;
  out[2] = whole ? 4 : ((a.merged & 3) == 3 ? 3 : ((a.merged & 1) ? 2 : 1));
  out[3] = 1;
  if (whole) {
    const int slices = (B + kR - 1) / kR;
    const bool fits = chain_rows(h, B) * slices <= h->n_cus;
    out[3] = (h->no_chain || h->chain_flags == nullptr || !fits) ? 1 : h->chain_max;
  }
  out[4] = a.wide; out[5] = a.nc; out[6] = a.twin_split; out[7] = a.p2_pair;
  out[8] = a.x2 ? 2 : (a.bf16 ? 1 : 0);
  out[9] = a.xcd_local;
  if (h->cfg.export_grads) {
    const bool was = h->dp_inline;
    h->dp_inline = true;
    const DdpgArgs d = ddpg_args(h, B);
    h->dp_inline = was;
    out[11] = (d.whole && B <= 256) ? 4 : ((d.merged & 3) == 3 ? 3 : ((d.merged & 1) ? 2 : 1));
  }
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

