/* oprl_amd.h — C-ABI of the MI355X-native off-policy learner (liboprl_amd.so).
 *
 * The reference (schatty/oprl) has no FFI: its "operator API" for this path is
 * two Python Protocols.  Each entry point below names the reference interface
 * it sits under (paths relative to /root/reference/src/oprl):
 *
 *   oprl_learner_*   AlgorithmProtocol.update()           algos/protocols.py:21-41
 *                    DDPG/TD3/SAC/TQC.update()            algos/ddpg.py:61, td3.py:71, sac.py:75, tqc.py:116
 *   oprl_replay_*    ReplayBufferProtocol                 buffers/protocols.py:6-26
 *                    EpisodicReplayBuffer.{add_transition,sample}  buffers/episodic_buffer.py:81-133
 *   oprl_mlp_*       MLP / Critic / policies forward      algos/nn_models.py:27-194
 *   oprl_adam_* / oprl_polyak   torch.optim.Adam.step / soft_update  algos/nn_functions.py:5-10
 *
 * Conventions: every function returns 0 on success or a negative oprl_status;
 * oprl_last_error() returns the text of the last failure on the calling
 * thread.  No exceptions cross the boundary.  All `float*`/`void*` data
 * pointers are DEVICE pointers (HBM) unless the name ends in `_host`.  `stream`
 * is a hipStream_t passed as void* (0 = default stream); all work is enqueued
 * asynchronously on it.  The library never frees caller memory; it allocates
 * only its own workspace in *_create and frees it in *_destroy.  A handle is
 * used from one host thread at a time; distinct handles are independent.
 */
#ifndef OPRL_AMD_H
#define OPRL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPRL_ABI_VERSION 1
#define OPRL_MAX_LAYERS 4   /* linear layers per MLP (TQC critic has 4) */
#define OPRL_MAX_CRITICS 5  /* TQC n_nets */

typedef enum oprl_status {
  OPRL_OK = 0,
  OPRL_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  OPRL_ERR_HIP = -2,       /* a HIP runtime call failed */
  OPRL_ERR_STATE = -3,     /* handle not ready (e.g. empty replay) */
  OPRL_ERR_NOMEM = -4
} oprl_status;

typedef enum oprl_algo { OPRL_DDPG = 0, OPRL_TD3 = 1, OPRL_SAC = 2, OPRL_TQC = 3 } oprl_algo;

/* Arithmetic mode of the MLP GEMMs.
 *   F32  = exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), the parity mode: Q-values and gradients within
 *          1e-4 of the reference (measured <= 2e-6).  A fused single-critic DDPG learner (own Adam step) runs whole
 *          updates, up to 32 per launch, in this mode too (k_ddpg_chain<PrecF32>): its kernels read and write
 *          library-owned UNCACHED mirrors of the packs below, and the caller's `pack` / `pack_target` are then
 *          maintained like an X2 learner's fp32 packs — rebuilt from the masters by every entry point that reads
 *          them (oprl_mlp_forward / backward / act, oprl_net_repack), not by the updates.
 *   BF16 = v_mfma_f32_16x16x32_bf16 with fp32 accumulation (16x the matrix rate, half the weight
 *          bytes): both GEMM operands of the forward and backward passes of the fused update kernels
 *          (DDPG / TD3 / SAC phase kernels, TQC's layer-wise critic kernels) are rounded to bf16 on the
 *          way into the matrix cores — the weights once per Adam step, into bf16 fragment packs the
 *          library owns.  Master weights, Adam moments, Polyak targets, activations in LDS / HBM, the
 *          heads, losses and the dW GEMM stay fp32.  Outside the fused kernels (oprl_mlp_forward / act /
 *          backward, generic fall-back launches for shapes the fused kernels do not cover) a BF16
 *          learner computes in fp32 from the fp32 packs.  Deviation from the reference: ~1e-3..1e-2
 *          relative on Q (bf16 inputs cannot meet the 1e-4 gate; tests/test_gpu_bf16.py states and
 *          measures the tolerance).
 *   X2   = every fp32 operand as the sum of two fp16 numbers (hi + lo), three v_mfma_f32_16x16x32_f16 per product
 *          (lo*hi + hi*lo + hi*hi) with fp32 accumulation: 22 mantissa bits per operand at 3/16 of the exact-fp32
 *          MFMA time.  Weights are kept as fp16 pairs of 2^8 w (packs of two planes the library owns), activations
 *          and gradient tiles enter scaled by a power of two (csrc/engine.h, PrecX2); accumulators, master weights,
 *          Adam and Polyak are fp32.  A PARITY mode: held to the reference's golden vectors and the CPU oracle at the
 *          F32 mode's gates (tests/test_gpu_x2.py); measured rms error of a 256-deep dot product 1.45e-7 relative (the
 *          fp32 MFMA chain: 2.8e-7).  Covers the fused DDPG / TD3 / SAC kernels — DDPG's whole updates, up to 32 per
 *          launch (k_ddpg_chain; oprl_learner_step_n) — and the hidden layers of TQC's 512-wide critics; everything else
 *          runs exact fp32.  RANGE: observations and hidden activations below 4094 in magnitude, weights below 256;
 *          leaving it is reported through the error word (oprl_learner_check / the next update return OPRL_ERR_STATE,
 *          "split-fp16 range"), never silent; F32 has no such range and is the default of the Python classes.  The fp32
 *          packs of such a learner are not kept current by its updates: every entry point that reads them
 *          (oprl_mlp_forward / backward / act, the generic launches) rebuilds them first. */
typedef enum oprl_precision { OPRL_PREC_F32 = 0, OPRL_PREC_BF16 = 1, OPRL_PREC_X2 = 2 } oprl_precision;

/* One MLP (ReLU hidden layers, identity output), parameters laid out exactly
 * like the reference module's state_dict: W0[out0,in0] row-major, b0, W1, b1...
 * (algos/nn_models.py:84-107).  theta is the trainable master copy; the other
 * arenas use the same layout and may be NULL when the net has no target / is
 * not trained. */
typedef struct oprl_net {
  int32_t n_layers;                     /* number of Linear layers, 2..OPRL_MAX_LAYERS */
  int32_t dims[OPRL_MAX_LAYERS + 1];    /* dims[0]=input ... dims[n_layers]=output */
  float* theta;
  float* theta_target;
  float* adam_m;
  float* adam_v;
  float* grad;                          /* written when grads are exported (DP / tests) */
  /* Fragment-order weight packs (device, caller-owned, oprl_net_pack_floats()
   * floats each, zero-initialised by the caller): the layout the MFMA kernels
   * stream (csrc/engine.h).  The library keeps them in step with theta /
   * theta_target whenever IT changes those (update, apply) — or, for the
   * learners named under oprl_precision, rebuilds them when one of its entry
   * points reads them; after an outside change of the master
   * (load_state_dict ...) call oprl_learner_sync_params() (every pack of the
   * learner, the library-owned ones included) or, for a net no learner of a
   * BF16 / X2 / chain kind owns, oprl_net_repack(). */
  float* pack;
  float* pack_target;                   /* NULL when there is no target */
} oprl_net;

/* Doubles on purpose: the reference's hyper-parameters are python floats and
 * torch forms 1-beta1, 1-beta2, 1-tau, lr/(1-beta1^t) in double before rounding
 * to fp32 (1.f - 0.999f differs from (float)(1 - 0.999) by 1.3e-5 relative). */
typedef struct oprl_hparams {
  double gamma, tau;
  double lr_actor, lr_critic, lr_alpha;
  double beta1, beta2, adam_eps;         /* torch defaults .9 / .999 / 1e-8 */
  double policy_noise, noise_clip, max_action;   /* TD3 (td3.py:98-103) */
  double alpha_init;                             /* SAC fixed alpha (sac.py:64) */
  double target_entropy;                         /* -action_dim */
  int32_t policy_freq;                           /* TD3 (td3.py:81) */
  int32_t tune_alpha;                            /* SAC (sac.py:65-70); TQC always 1 */
  int32_t n_quantiles, top_quantiles_to_drop;    /* TQC (tqc.py:71-73) */
} oprl_hparams;

typedef struct oprl_learner_config {
  int32_t abi_version;      /* OPRL_ABI_VERSION */
  int32_t algo;             /* oprl_algo */
  int32_t precision;        /* oprl_precision */
  int32_t state_dim, action_dim;
  int32_t max_batch;        /* workspace is sized for this many rows */
  int32_t n_critics;        /* 1 DDPG, 2 TD3/SAC, n_nets TQC */
  int32_t no_fuse;          /* 1: always use the generic per-net launch sequence (DDPG
                               otherwise runs the fused two-kernel path, csrc/fused_ddpg.hip) */
  int32_t export_grads;     /* 1: update() stops before Adam and leaves grads in
                               net.grad (data-parallel learner reduces them, then
                               calls oprl_learner_apply); 0: fused dW+Adam+Polyak */
  oprl_net actor;
  oprl_net critics[OPRL_MAX_CRITICS];
  double* log_alpha;        /* device scalar (float64 like the reference), or NULL */
  double* log_alpha_m;      /* its Adam state (device), or NULL */
  double* log_alpha_v;
  double* log_alpha_grad;   /* export_grads: d(alpha loss)/d(log_alpha) lands here */
  oprl_hparams hp;
} oprl_learner_config;

typedef struct oprl_learner oprl_learner;
typedef struct oprl_replay oprl_replay;

const char* oprl_last_error(void);
int oprl_abi_version(void);

/* ---- learner: AlgorithmProtocol.update() ------------------------------- */
int oprl_learner_create(const oprl_learner_config* cfg, oprl_learner** out);
int oprl_learner_destroy(oprl_learner* h);

/* One reference-semantics update() on a caller-supplied minibatch (row-major
 * fp32: s[B,S] a[B,A] r[B] d[B] s2[B,S]).  noise0/noise1 are the N(0,1) draws
 * the reference takes inside update() (TD3: randn_like(action), td3.py:98;
 * SAC/TQC: next-state draw then actor-step draw, nn_models.py:213), each
 * [B,A]; NULL = draw on device (Philox, keyed by seed and the update counter).*/
int oprl_learner_update(oprl_learner* h, const float* s, const float* a, const float* r,
                        const float* d, const float* s2, int32_t B,
                        const float* noise0, const float* noise1, void* stream);

/* Second half of an export_grads update: Adam (+Polyak) from net.grad after the
 * caller has all-reduced it.  phase 0 = critic(s), 1 = actor (+alpha). */
int oprl_learner_apply(oprl_learner* h, int32_t phase, double grad_scale, void* stream);
/* In export_grads mode update() is split so the reduction can sit between the
 * halves exactly where the reference's optimizer.step() sits:
 *   oprl_learner_update_phase(h, 0, batch...) -> critic grads in net.grad
 *   <all-reduce> ; oprl_learner_apply(h, 0, 1/world)
 *   oprl_learner_update_phase(h, 1, ...)      -> actor grads
 *   <all-reduce> ; oprl_learner_apply(h, 1, 1/world)                        */
int oprl_learner_update_phase(oprl_learner* h, int32_t phase, const float* s, const float* a,
                              const float* r, const float* d, const float* s2, int32_t B,
                              const float* noise0, const float* noise1, void* stream);

/* K back-to-back sample()+update() iterations with the minibatch drawn on
 * device from `replay` (the 4000-update loop of
 * distrib/policy_update_worker.py:65-68 as one call). */
int oprl_learner_step_n(oprl_learner* h, oprl_replay* replay, int32_t K, int32_t B,
                        uint64_t seed, void* stream);

/* The trainer loop's step (trainers/base_trainer.py:38-74: one update, then actor.explore(next_state) opens the next
 * environment step): oprl_learner_step_n with K = 1 and, enqueued behind it in the same call, the ACTOR's forward
 * of one observation obs_host[0 .. state_dim) with the weights that update leaves (a 1-row GEMV over the master
 * weights, csrc/policy_act.hip).  Nothing is waited for: the raw output row of the actor's last layer (no tanh, no
 * Gaussian head — what oprl_mlp_act returns with out_act = none) is collected with oprl_learner_act_wait, which
 * spins on a ticket in host-mapped memory (bounded: OPRL_ERR_STATE after `timeout_us`).  One pending row per learner. */
int oprl_learner_step_act(oprl_learner* h, oprl_replay* replay, int32_t B, uint64_t seed,
                          const float* obs_host, void* stream);
int oprl_learner_act_wait(oprl_learner* h, float* out_host, int32_t n_out, int64_t timeout_us);

/* ---- packed learners: the reference's --seeds fan-out (runners/train.py:24-50) on ONE GPU ----------------
 * A group steps N independent fused DDPG, TD3 or SAC learners of one algorithm, shape and precision (own weights, own sampler keys
 * seeds[i], one shared HBM replay) with FOUR launches per update for all of them (grid.z = learner; the
 * argument blocks of four updates travel to device memory in one copy).  OPRL_PREC_F32 members run on
 * single-CU slices (oprl_learner_set_cluster(h, 1) is applied to them), OPRL_PREC_X2 / OPRL_PREC_BF16 members on
 * clusters of four ((h, 4): the form those precisions exist in): a member's result is bit-identical to the same
 * learner stepped alone with that cluster size, whoever else is in the group.  TD3 / SAC members (fused in the lean
 * form only): clusters of four in every precision, the twin critics back to back (OPRL_AMD_NO_TWIN_SPLIT=1 /
 * OPRL_AMD_NO_P2_PAIR=1 give a solo learner the same form); TD3's delayed actor step is taken by all members at
 * once — members whose update counts differ modulo policy_freq are refused (OPRL_ERR_STATE); SAC's temperature
 * step rides on the actor's dW launch.  The members stay ordinary learners (update / step_n / checkpoints)
 * between group calls. */
typedef struct oprl_group oprl_group;
int oprl_group_create(oprl_learner** learners, int32_t n, oprl_group** out);
int oprl_group_destroy(oprl_group* g);
int oprl_group_step_n(oprl_group* g, oprl_replay* replay, int32_t K, int32_t B, const uint64_t* seeds, void* stream);
/* CUs per 16-row slice in the fused kernels.  8 (the default): tensor-parallel clusters of four, and of EIGHT
 * for the forward-only chains where that still fits the chip (DDPG's target chain, DDPG / TD3's critic pass of
 * the actor step; exact-fp32 mode, action_dim <= 8) — lowest latency for a learner that has the GPU (nearly) to
 * itself; such launches want every CU, so learners that share a GPU with more than two others (seeds packed on
 * streams, one process per seed on one GPU) use 4: clusters of four only (also OPRL_AMD_NO_WIDE=1), and the dW + Adam
 * tiles as launches of their own instead of workgroups riding on the phase launches.  2 or 1:
 * least CU time per update (what oprl_group uses).  Results differ in the last bits between settings (order of
 * the exchange sums); a setting is part of a run's configuration like the seed. */
int oprl_learner_set_cluster(oprl_learner* h, int32_t nc);
/* Diagnostics of the most recent update, read without forcing a sync inside
 * update(): out_host[0]=critic_loss, [1]=-mean q(s, pi) (DDPG / TD3 actor loss; min over twins for SAC),
 * [2]=mean q over all critics, [3]=mean target, [4]=alpha, [5]=update_step, and (n up to 10) [6]=mean q of
 * critic 0 (the reference's "q1"), [7]=mean log pi of the actor step, [8]=SAC / TQC actor loss
 * alpha * mean(log pi) - mean(min q), [9]=temperature loss.  Synchronises `stream`. */
int oprl_learner_read_scalars(oprl_learner* h, float* out_host, int32_t n, void* stream);
/* Rebuild every pack of the learner's nets from their masters (after the caller
 * changed parameters from outside, e.g. load_state_dict). */
int oprl_learner_sync_params(oprl_learner* h, void* stream);
int oprl_learner_update_count(oprl_learner* h, int64_t* out_host);
int oprl_learner_set_update_count(oprl_learner* h, int64_t count);
/* Checkpoint / exact resume (reference: base_trainer.py:113-120 saves only the policy).  The
 * learner's host-side state beyond the caller-owned arenas: {update_count, Adam step of the
 * critics, of the actor, of log_alpha}.  Restoring the arenas (theta, theta_target, m, v,
 * log_alpha + its m, v), these four counters and calling oprl_learner_sync_params resumes a
 * run bit for bit (tests/test_gpu_callers.py). */
#define OPRL_N_COUNTERS 4
int oprl_learner_get_counters(oprl_learner* h, int64_t out_host[OPRL_N_COUNTERS]);
int oprl_learner_set_counters(oprl_learner* h, const int64_t in_host[OPRL_N_COUNTERS]);
/* Device-side failures.  The fused update kernels and the data-parallel exchanges contain BOUNDED waits
 * between workgroups (4-CU slice clusters, the TD-target hand-off between roles, gradient tiles between
 * ranks).  An expired wait poisons its result with NaN — it cannot hang the GPU — and is REPORTED: the
 * kernel stores (kernel << 8 | wait site) into a host-visible error word of the learner.  The word is
 * checked, without any synchronisation, at the start of every oprl_learner_update / update_phase / apply /
 * step_n / dp_* call and after the synchronisation inside oprl_learner_read_scalars; once set those calls
 * return OPRL_ERR_STATE with the kernel and wait site in oprl_last_error() until it is cleared.
 * oprl_learner_check polls it explicitly (synchronise the stream first for a definitive answer);
 * oprl_learner_debug_expire (tests) makes one wait site (2 = TD-target hand-off, 1 = cluster all-reduce, 7 = gate of the dW tiles riding on phase 1's launch;
 * 0 = off) give up immediately in subsequent launches.  Sites 101, 102, 104, 105, 106 are TIMING experiments
 * (tools/what_if.py): one cross-workgroup wait of k_ddpg_chain counts as satisfied — role B's q for role A's tail,
 * the critic's tiles for the critic pass, role A's seeds / the pass's du / role B's rows for the tiles — so that
 * the change of the update's period shows what that hand-over contributes; such a learner computes wrong numbers. */
int oprl_learner_check(oprl_learner* h);
int oprl_learner_clear_error(oprl_learner* h);
int oprl_learner_debug_expire(oprl_learner* h, int32_t site);
/* Which launch form an update of batch size B would take right now (tests/test_gpu_forms.py holds the selection —
 * csrc/learner.hip ddpg_args / critic_phase — against a table).  out[12]: [0] 1 fused phase kernels / 0 the generic launch
 * sequence, [1] lean (tp4.h) passes, [2] form 4 whole updates per launch (k_ddpg_chain) / 3 both merged launches / 2 merged
 * phase 1 / 1 plain phase + dW launches / 0, [3] updates per chain launch, [4] wide bits (1 role A, 2 the critic pass on
 * clusters of eight), [5] cluster size of the other roles, [6] twin_split, [7] p2_pair, [8] arithmetic 0 exact fp32 /
 * 1 bf16 / 2 x2, [9] XCD-local cluster exchanges, [10] demoted to the shared-chip forms, [11] 0.  Reference: none (the
 * reference has one path, autograd). */
int oprl_learner_debug_form(oprl_learner* h, int32_t B, int32_t* out);
/* Key of the learner's device-side noise streams (TD3 target smoothing, the SAC / TQC
 * reparameterisation draws, drawn with Philox when update() gets no injected noise): `seed` is the
 * run seed (the reference seeds torch's generator in runners/train.py:14-21), `rank` the
 * data-parallel rank (oprl_comm_init / oprl_p2p_create set it too) so that every seed and every
 * rank draws its own eps.  Seed 0 on rank 0 is the default. */
int oprl_learner_set_seed(oprl_learner* h, uint64_t seed, int32_t rank);
/* Test / audit export of the device-side noise: out_dev[rows][cols] = the N(0,1) draws (Philox4x32-10 +
 * Box-Muller, csrc/philox.h) the update kernels of this learner take for noise stream `stream_id` (1: next-state
 * draw / TD3 target smoothing, 2: actor-step draw) at update counter `counter`, element (row, col) = (minibatch
 * row, action dimension), under the learner's current seed and rank (oprl_learner_set_seed). */
int oprl_debug_noise(oprl_learner* h, int32_t stream_id, uint64_t counter, int32_t rows, int32_t cols,
                     float* out_dev, void* stream);
/* device pointer to the per-row Q / TD-target of the last critic step ([B] each,
 * critic 0), for parity tests. */
int oprl_learner_debug_ptrs(oprl_learner* h, const float** q, const float** y);
/* (debug) device views of what a fused DDPG update leaves in the workspace; `which` as listed at the definition */
int oprl_learner_debug_view(oprl_learner* h, int32_t which, const void** ptr, int64_t* n_bytes);

/* ---- building blocks (nn_models.py forward; used by Module.__call__) ----- */
/* floats needed for ONE pack buffer of this net (dims only are read) */
int64_t oprl_net_pack_floats(const oprl_net* net);
/* rebuild pack from theta (which & 1) and/or pack_target from theta_target (which & 2) */
int oprl_net_repack(const oprl_net* net, int32_t which, void* stream);
/* out[B,dims[L]] = MLP([x0 | x1]) with x0[B,k0], x1[B,k1] (x1 may be NULL,
 * k0+k1 == dims[0]); out_act: 0 identity, 1 tanh.  use_target selects
 * theta_target. */
int oprl_mlp_forward(const oprl_net* net, int32_t use_target, const float* x0, int32_t k0,
                     const float* x1, int32_t k1, int32_t B, int32_t out_act, float* out,
                     void* stream);
/* The per-environment-step policy call (reference nn_models.py:138-150 explore / exploit and
 * :180-195: as_tensor(state) -> forward -> .cpu()): ONE observation obs_host[k0] in host memory ->
 * out_host[n_out] in host memory (n_out = dims[L], or dims[L]/2 with out_act 4 = tanh of the
 * Gaussian mean).  Synchronous on `stream`. */
int oprl_mlp_act(const oprl_net* net, const float* obs_host, int32_t k0, int32_t out_act,
                 float* out_host, int32_t n_out, void* stream);
/* Gradient of sum(out * dout) wrt every parameter (into net->grad) and, if
 * dx != NULL, wrt the concatenated input [B,dims[0]].  Test/debug entry that
 * exercises the same backward + dW kernels update() uses. */
int oprl_mlp_backward(const oprl_net* net, const float* x0, int32_t k0, const float* x1,
                      int32_t k1, int32_t B, const float* dout, float* dx, void* stream);
/* torch.optim.Adam.step over a flat arena of n floats; step = 1-based count. */
int oprl_adam_step(float* theta, float* m, float* v, const float* grad, int64_t n, int32_t step,
                   double lr, double beta1, double beta2, double eps, double grad_scale,
                   void* stream);
/* target <- (1-tau)*target + tau*source  (nn_functions.py:5-10) */
int oprl_polyak(float* target, const float* source, int64_t n, double tau, void* stream);

/* ---- data parallel over RCCL (xGMI) ---------------------------------------- */
/* One process per GPU.  The library binds RCCL at run time from `rccl_path`
 * (dlopen; e.g. torch's bundled librccl.so) — it does not link it.  Rank 0 obtains
 * a 128-byte unique id, the host distributes it (any side channel, e.g.
 * torch.distributed's store), every rank calls oprl_comm_init on a learner created
 * with export_grads = 1.  oprl_learner_dp_update / dp_step_n then run the
 * synchronous data-parallel update entirely in C: critic phase -> ncclAllReduce of
 * the critic gradient arena -> Adam with grads scaled by 1/world -> actor phase ->
 * ncclAllReduce of the actor gradient arena (and the temperature gradient) -> Adam.
 * New functionality (the reference has a single learner). */
#define OPRL_COMM_ID_BYTES 128
int oprl_comm_unique_id(const char* rccl_path, char id_out[OPRL_COMM_ID_BYTES]);
int oprl_comm_init(oprl_learner* h, const char* rccl_path, int32_t rank, int32_t world,
                   const char id[OPRL_COMM_ID_BYTES]);
/* Make every replica identical to rank `root` (ncclBroadcast of every net's theta / theta_target / Adam
 * moments and of the temperature with its moments; the packs are rebuilt).  Once, after oprl_comm_init. */
int oprl_comm_broadcast_params(oprl_learner* h, int32_t root, void* stream);
/* Optional: the two gradient exchanges as ONE-SHOT all-reduces over peer windows (csrc/p2p.hip)
 * instead of RCCL rings — every rank pushes its arena straight into a slot of every other rank's
 * window over the xGMI mesh and sums the slots in rank order.  oprl_p2p_create allocates this rank's
 * window and returns its IPC handle; the host gathers the `world` handles (any side channel) and
 * passes them, in rank order, to oprl_p2p_connect; oprl_p2p_selftest (all ranks together) exchanges
 * a known pattern and reports whether this rank's sums were exact; when every rank passed, every
 * rank calls oprl_p2p_enable(h, level): 1 = the two exchanges of dp_update / dp_step_n run as one
 * window kernel each; 2 = in addition the fused learners (DDPG / TD3 / SAC) all-reduce every gradient
 * tile INSIDE their dW + Adam launches (no separate exchange or apply launches at all; the caller should
 * check after a few updates that parameters are finite and replicas identical, and step down to 1
 * otherwise — bench.py does); 0 = back to RCCL.  Without RCCL
 * (oprl_comm_init never called) a connected learner runs dp_update / dp_step_n on the windows alone. */
#define OPRL_P2P_HANDLE_BYTES 64
int oprl_p2p_create(oprl_learner* h, int32_t rank, int32_t world, char handle_out[OPRL_P2P_HANDLE_BYTES]);
int oprl_p2p_connect(oprl_learner* h, const char* handles /* world x OPRL_P2P_HANDLE_BYTES */);
int oprl_p2p_selftest(oprl_learner* h, void* stream);
int oprl_p2p_enable(oprl_learner* h, int32_t on);
int oprl_learner_dp_update(oprl_learner* h, const float* s, const float* a, const float* r,
                           const float* d, const float* s2, int32_t B, const float* noise0,
                           const float* noise1, void* stream);
int oprl_learner_dp_step_n(oprl_learner* h, oprl_replay* replay, int32_t K, int32_t B,
                           uint64_t seed, void* stream);

/* ---- measurement --------------------------------------------------------- */
/* When enabled, every kernel launch of this library is bracketed by a pair of
 * hipEvents recorded on the launch stream.  oprl_profile_read synchronises the
 * device and returns, per kernel kind, the launch count and the summed
 * hipEventElapsedTime (ms).  Kinds: 0 k_mlp_slice, 1 k_dw_adam, 2 k_replay_gather,
 * 3 everything else, 4 k_ddpg_phase1 — and every launch that starts with phase 1's roles: the merged
 * k_ddpg_phase1_dw and k_ddpg_chain (whole updates, up to 32 per launch: one count per LAUNCH) —,
 * 5 k_ddpg_phase2 (and k_ddpg_phase2_dw).  bench.py's roofline uses this. */
#define OPRL_PROFILE_KINDS 6
int oprl_profile_enable(int32_t on);
int oprl_profile_read(int64_t* counts_host, double* ms_host, int32_t reset);

/* Debug tracing: when buf != NULL every k_mlp_slice launch of this learner writes
 * per-workgroup phase timestamps (shader cycle counter, 100 MHz realtime) to
 * buf[launch_slot][64 workgroups][OPRL_TRACE_STAMPS][2] (int64, device memory);
 * launch_slot restarts at 0 with each update().  NULL disables. */
#define OPRL_TRACE_STAMPS 24
#define OPRL_TRACE_SLOTS 24
int oprl_learner_set_trace(oprl_learner* h, int64_t* buf);

/* ---- replay: ReplayBufferProtocol --------------------------------------- */
/* Storage tensors are owned by the caller (torch) with the reference layout
 * (buffers/episodic_buffer.py:29-55): states[E,L+1,S] actions[E,L,A]
 * rewards[E,L,1] dones[E,L,1], fp32, HBM resident. */
int oprl_replay_create(int32_t n_episodes, int32_t max_ep_len, int32_t state_dim,
                       int32_t action_dim, float* states, float* actions, float* rewards,
                       float* dones, oprl_replay** out);
int oprl_replay_destroy(oprl_replay* h);
/* add_transition's data movement: stage one transition (host pointers) for slot
 * [ep, t]; staged rows reach HBM in one batched copy + scatter at the next
 * flush/sample.  Index bookkeeping stays with the caller (it is host logic). */
int oprl_replay_write(oprl_replay* h, int32_t ep, int32_t t, const float* state_host,
                      const float* action_host, float reward, float done);
/* oprl_replay_write, oprl_replay_set_lens and oprl_replay_flush as one call (the per-env-step caller:
 * trainers/base_trainer.py adds one transition between two updates; the row and the changed tail of the episode
 * table travel in one small launch that runs while the host goes on to the update call). */
int oprl_replay_write_flush(oprl_replay* h, int32_t ep, int32_t t, const float* state_host,
                            const float* action_host, float reward, float done, const int32_t* ep_lens_host,
                            int32_t episodes_counter, void* stream);
/* The same for n consecutive steps [t0, t0+n) of one episode from host records
 * [state (S) | action (A) | reward | done | ...], row_stride floats apart: add_episode / a drained actor
 * ring segment in one call.  Staging that fills up is flushed on `stream`. */
int oprl_replay_write_block(oprl_replay* h, int32_t ep, int32_t t0, int32_t n, const float* rows_host,
                            int32_t row_stride, void* stream);
int oprl_replay_flush(oprl_replay* h, void* stream);
/* Upload ep_lens[0:episodes_counter] (episodic_buffer.py:114-116); the device
 * keeps the cumulative ends for the flat-index -> (episode, step) map. */
int oprl_replay_set_lens(oprl_replay* h, const int32_t* ep_lens_host, int32_t episodes_counter,
                         void* stream);
/* sample(): gather B transitions.  idx (device int64[B], flat indices as
 * np.random.randint would give, episodic_buffer.py:124) or NULL to draw them
 * on device from (seed, counter).  Outputs: s[B,S] a[B,A] r[B,1] d[B,1] s2[B,S].
 * If out_ep/out_step are non-NULL the (episode, step) pairs are written too. */
int oprl_replay_sample(oprl_replay* h, int32_t B, const int64_t* idx, uint64_t seed,
                       uint64_t counter, float* out_s, float* out_a, float* out_r, float* out_d,
                       float* out_s2, int32_t* out_ep, int32_t* out_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPRL_AMD_H */
