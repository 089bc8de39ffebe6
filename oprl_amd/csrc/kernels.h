// kernels.h — argument blocks shared between the host orchestration (learner.hip)
// and the device kernels (kernels.hip).
#pragma once
#include "engine.h"

namespace oprl {

// ---- output activation of the slice kernel's forward half ------------------
enum OutAct : int {
  ACT_NONE = 0,
  ACT_TANH = 1,         // DeterministicPolicy.forward             nn_models.py:135-136
  ACT_TANH_SMOOTH = 2,  // TD3 target smoothing                    td3.py:98-103
  ACT_GAUSS = 3,        // GaussianActor train-mode forward        nn_models.py:168-178
  ACT_GAUSS_MEAN = 4,   // GaussianActor eval: tanh(mean)          nn_models.py:179-181
};

// ---- how the backward half obtains dLoss/d(out) ----------------------------
enum SeedMode : int {
  SEED_PTR = 0,       // read from global
  SEED_MSE_TD = 1,    // 2(q - y)/B with y = r + (1-d) gamma (min(qn1,qn2) - alpha logp')
  SEED_CONST = 2,     // constant (actor loss through a critic: -1/B, TQC: -1/(B N Q))
  SEED_MINQ = 3,      // SAC actor: -[q_j is the min]/B                 sac.py:126
  SEED_TANH = 4,      // du = da (1 - a^2)                               ddpg.py:104
  SEED_GAUSS = 5,     // tanh-Gaussian head backward
  SEED_QHUBER = 6,    // quantile-Huber dL/dz                            tqc.py:14-36
};

struct SeedArgs {
  const float* p0;   // PTR: dout | MSE_TD: qn1 | MINQ: q1 | TANH: da | GAUSS: da (net 0) | QHUBER: target[B,M]
  const float* p1;   //             MSE_TD: qn2 or null | MINQ: q2 | TANH: a | GAUSS: raw out [B,2A]
  const float* p2;   //             MSE_TD: logp' or null   (GAUSS eps comes from MlpArgs.noise / Philox)
  const float* r;    // MSE_TD
  const float* d;    // MSE_TD
  const double* log_alpha;  // device scalar or null (then alpha_const)
  float alpha_const;
  float gamma;
  float cval;        // CONST value; MINQ/GAUSS/MSE: 1/B ; QHUBER: 1/(B*N*Q*M)
  int which;         // MINQ: 0/1 = which critic this is
  int ld0;           // leading dim of p0 (PTR/TANH/GAUSS da)
  int n_da;          // GAUSS: number of stacked da buffers to sum (critics), stride da_stride
  long da_stride;
  int M;             // QHUBER: samples per row
  int Q;             // QHUBER: quantiles per net
  float* y_out;      // MSE_TD: TD target [B] (debug / diagnostics), may be null
  float* q_out;      // MSE_TD: current q [B], may be null
  // GAUSS, the backward riding on the launch that PRODUCES da (k_lw_dact, r06-16): flag (net n, slice) =
  // da_flags[n * da_fstride + slice] = {da_tag, *} once that net's rows of the slice are written through; null: da is an
  // earlier launch's
  const unsigned long long* da_flags;
  unsigned da_tag;
  int da_fstride, da_spin;
};

struct MlpArgs {
  Net net;
  int B;
  int do_fwd, do_bwd;
  // forward input [x0 | x1]
  const float* x0; int k0;
  const float* x1; int k1;
  int out_act;
  float* out; int ldo;            // activated output [B][ldo] (ACT_GAUSS*: action [B][A])
  float* raw_out; int ldraw;      // ACT_GAUSS: raw net output [B][2A] for the backward
  float* logp;                    // ACT_GAUSS: [B]
  const float* noise;             // ACT_TANH_SMOOTH / ACT_GAUSS: N(0,1) draws [B][A] (null: Philox)
  unsigned long long rng_seed, rng_ctr;
  float policy_noise, noise_clip, max_action;
  int action_dim;
  // activation exchange with the dW kernel / a later backward launch
  float* Xg[kMaxLayers]; int ldx0;   // Xg[0]: [B][ldx0] concatenated input; Xg[l>=1]: [B][WIDTH]
  // backward
  int seed_mode; SeedArgs seed;
  float* dYg[kMaxLayers]; int lddo;  // dYg[l<L-1]: [B][WIDTH]; dYg[L-1]: [B][lddo]
  int dact_col0, dact_cols; float* dact; int lddact;
  float* partials;                   // [gridDim.x][4] per-slice sums: loss, q, y, (spare)
  long long* trace;                  // debug: [gridDim.x][kTraceStamps][2] (shader clock, 100 MHz realtime)
  // tensor-parallel launch (csrc/slice_tp.hip): cluster exchange area and launch-unique tag, and
  // the distance between the members' dz1 partial buffers (dYg[0] + member * dY0_stride)
  unsigned long long* tp_xbuf; unsigned tp_tag; long dY0_stride;
  unsigned* err;                     // the learner's host-visible error word (expired waits), or null
  // a backward riding in front of ITS dW tiles (k_lw_dact, r06-18): everything the tiles read leaves written through and
  // member m of slice s raises done_flags[4 s + m] = {done_tag, *} behind it; null: the tiles are a later launch
  unsigned long long* done_flags; unsigned done_tag;
  // host-side only (ignored by the kernels): where launch() draws the tag from, the exchange
  // area's size for the wrap-around reset
  unsigned* tp_tag_counter; size_t tp_xbuf_bytes;
  void* owner;                       // the learner: lets launch() pair two nets of one for_each_net
  // host-side only: this net's bf16 fragment packs per layer (bf16 learners; null otherwise) — a launcher
  // whose kernels run PrecBF16 moves them into net.pf / net.pb
  const float* pf16[kMaxLayers]; const float* pb16[kMaxLayers];
};

// k_lw_mid_pair (csrc/layerwise.hip): the flag granules of the launches that run two hidden layers as one, the tag of
// the next such launch (the caller advances it by one per launch made: launch_mlp_layerwise reports how many in `used`)
struct LwPairBuf {
  unsigned long long* flags; int n_flags;
  unsigned next_tag; int spin; unsigned* err;
  int use;                             // bit 0: forward pairs, bit 1: backward pairs
  mutable int used;                    // launches made with next_tag, next_tag + 1, ...
};
constexpr int kMaxMulti = 5;           // nets per k_mlp_slice_multi launch (TQC: 5 quantile critics)
struct MlpMultiArgs { MlpArgs a[kMaxMulti]; };

// ---- dW + Adam + Polyak ------------------------------------------------------
struct DwItem {      // one Linear layer of one net
  const float* X; int ldx; int K;      // layer input  [B][ldx]
  const float* dY; int ldy; int N;     // grad wrt layer pre-activation output [B][ldy]
  float *w, *w_t, *w_m, *w_v, *w_g;    // [N][K] views into theta / theta_target / m / v / grad
  float *b, *b_t, *b_m, *b_v, *b_g;    // [N]
  float *pf, *pb, *tpf;                // fragment-order packs: W (fwd), W^T (bwd), target W (fwd)
  float *pf16, *pb16, *tpf16;          // the same three as bf16 packs (PrecBF16, engine.h), or null: fp32-only learner
  int x2;                              // 1: those three are PrecX2 packs (blocks of two fp16 planes, 2^8 w)
  float* b16;                          // copy of the bias in UNCACHED memory (or null): what a workgroup of the same launch reads (k_ddpg_update's critic pass)
  float* bt16;                         // ... of the TARGET net's bias (or null): k_ddpg_chain's later updates
  int tiles_k, tile_begin, tile_end;
  int tile_n;                          // n rows per tile: kDwTileN, or 8 for a layer that sums dz1 partials
  long dY_part_stride;                 // > 0: dY is the sum of DwArgs::n_part buffers this many floats apart
  int scaled;                          // 1: dY rows may be unit-seed (tp4_scalar_fb): with DwArgs::use_row_scale multiply row b by rs[b * rs_ld]
  const float* rs; int rs_ld;          // the net's per-row seed dLoss/dq = dY of its output layer
};

struct RepackItem {  // one Linear layer: master -> packs
  const float* w; int N, K;
  float *pf, *pb;    // pb may be null (target nets are never differentiated)
  float *pf16, *pb16; // bf16 packs (bf16 learners), or null
  int x2;             // 1: those are PrecX2 packs (two fp16 planes per block, 2^8 w)
  int blk_begin, blk_end;   // 256-element blocks of the N*K index space
};

struct AdamScalars {
  float lr, beta1, beta2, eps;
  // 1-beta1, 1-beta2, 1-tau formed in double on the host and then rounded, as
  // torch does with its python-float scalars (1.f - 0.999f is off by 1.3e-5 rel.)
  float omb1, omb2, omtau;
  double lr_d, beta1_d, beta2_d;       // the python-double hyper-parameters, for the bias corrections
  float step_size_host, bc2_sqrt_host; // lr/(1-b1^t), sqrt(1-b2^t) when the host knows t
  int step_base; const int* step_dev;  // Adam step = step_base + (step_dev ? *step_dev : 0)
  float tau; int do_polyak;
  int do_adam;                         // 0: only export grads
  float grad_scale;
};

// Per element, torch.optim.Adam single-tensor semantics and the Polyak target update:
//   m += (g-m)(1-b1);  v = b2 v + (1-b2) g g;  th -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps)
//   th_t = (1-tau) th_t + tau th                       (nn_functions.py:5-10)
// returns bit0: theta written (*th_new), bit1: target written (*tt_new)
// One element's Adam step and Polyak step with every rounding spelled out: `a * b + c` left to the compiler is fused into an
// fma one way in one instantiation of an epilogue and another way in the next (v beta2 + (1 - beta2) g g has two mul-adds to
// choose from) — the same dW tile came out one ulp apart from two kernels that inline the same epilogue (r05-21).  Every
// epilogue calls these.
__device__ __forceinline__ void adam_elem(float g, float& m, float& v, float& th, const AdamScalars& ad, float step_size, float bc2_sqrt) {
  m = __builtin_fmaf(g - m, ad.omb1, m);
  v = __builtin_fmaf(v, ad.beta2, (ad.omb2 * g) * g);
  const float denom = sqrtf(v) / bc2_sqrt + ad.eps;
  th = __builtin_fmaf(-step_size, m / denom, th);
}
__device__ __forceinline__ float polyak_elem(float tt, float th, const AdamScalars& ad) { return __builtin_fmaf(tt, ad.omtau, ad.tau * th); }

__device__ __forceinline__ int adam_polyak_elem(float g, float* th, float* m, float* v, float* tt,
                                                float* gout, const AdamScalars& ad,
                                                float step_size, float bc2_sqrt, float* th_new,
                                                float* tt_new) {
  g *= ad.grad_scale;
  if (gout != nullptr) *gout = g;
  if (!ad.do_adam) return 0;
  float mm = *m, vv = *v, t = *th;
  adam_elem(g, mm, vv, t, ad, step_size, bc2_sqrt);
  *m = mm;
  *v = vv;
  *th = t;
  *th_new = t;
  if (ad.do_polyak && tt != nullptr) {
    const float u = polyak_elem(*tt, t, ad);
    *tt = u;
    *tt_new = u;
    return 3;
  }
  return 1;
}


// Data-parallel learner, peer windows (csrc/p2p.hip): k_dw_adam<true> exchanges each 16x32 gradient tile
// (+ 16 bias sums) with the same tile of the other ranks between its GEMM and its Adam epilogue — the
// separate all-reduce and apply launches of the RCCL path disappear.  Tile region of a window:
// [parity][source rank, or `world` = the owner's summed tile][tile][kDwXchgTile] 8-byte {sequence, value}
// granules.
constexpr int kDwXchgTile = 1024 + 16;   // one 16 x 64 tile of dw_tile_x2.h + 16 bias sums (the 16 x 32 tiles of dw_body.h use the first 512 + their 16 at 512)
constexpr int kDwXchgMaxWorld = 8;
struct DwXchg {
  char* peer[kDwXchgMaxWorld];         // every rank's tile region as mapped here
  char* window;                        // this rank's tile region
  int world, rank, parity, max_tiles;
  unsigned long long seq;
  unsigned* err;                       // host-visible error word (expired tile wait), or null
};
__host__ __device__ inline size_t dw_xchg_bytes(int world, int max_tiles) {
  return (size_t)2 * (world + 1) * max_tiles * kDwXchgTile * sizeof(unsigned long long);
}

// The temperature's Adam step (float64, k_alpha_step) as ONE EXTRA workgroup of a k_dw_adam launch: SAC's and
// TQC's updates end with it, and as a launch of its own it cost a kernel boundary for 3 us of work.
struct AlphaJob {
  double* log_alpha = nullptr;         // null: no job
  double *m = nullptr, *v = nullptr;
  const float* logp = nullptr;         // [B] log pi of this update's actor sample
  int B = 0;
  float target_entropy = 0.f;
  double lr = 0, beta1 = 0, beta2 = 0, eps = 0, bc1 = 1, bc2_sqrt = 1;
};

// TQC's TD target (sort the target critics' n_nets * Q quantiles of a row, drop the top ones, TD backup:
// k_tqc_target) as the tail of the target critics' head launch (k_lw_head): the head workgroup that arrives LAST
// at its 16-row slice — a per-slice arrival counter, never reset: every launch adds n_nets to it — does the slice's
// rows, one wave per row.  The heads' outputs travel as agent-scope stores / loads (write-through: the nets'
// workgroups sit on different XCDs); as a launch of its own the target cost a kernel boundary for 2 us of work.
struct TqcJob {
  unsigned long long* counter = nullptr;   // [slices]; null: no job
  float* z = nullptr; long net_stride = 0; int ldz = 0;   // the heads' outputs [n_nets][B][ldz] (= their MlpArgs::out)
  int n_nets = 0, Q = 0, drop = 0;
  const float *r = nullptr, *d = nullptr, *logp = nullptr;
  const double* log_alpha = nullptr;
  float gamma = 0.f;
  float* target = nullptr;                 // [B][n_nets * Q - drop]
};

struct PrefetchJob;
struct DwArgs {                         // host-side description of one k_dw_adam launch
  const DwItem* items;                 // HOST array
  int n_items; int total_tiles; int B;
  int n_part;                          // members of the tensor-parallel cluster that wrote dz1 partials
  int dy_tiled = 0;                    // 1: those partial buffers are tile-major (written by the tp4 passes, Tp3Store::dY0_tile_rows = B)
  AdamScalars ad;
  long long* trace;                    // debug stamps (tools/trace_slice.py) or null
  int use_row_scale;                   // 1: the slice kernels left unit-seed dz rows (lean fused path)
  int skip32 = 0;                      // 1: do not write the fp32 packs (pf / pb / tpf): a PrecX2 learner's fused update, whose kernels read the fp16 packs only
  int skip32_wide = 0;                 // 1: ... of the WIDE layers only (TQC's 512 x 512 layers in a 16-bit mode: the hidden-layer launches read the 16-bit packs)
  int apply_only;                      // 1: no GEMM — the gradient is read from w_g / b_g (data-parallel apply after the all-reduce)
  const DwXchg* xchg = nullptr;       // data-parallel: all-reduce every gradient tile over the peer windows inside this launch
  AlphaJob alpha;                      // optional: the temperature step rides on this launch (one more workgroup)
  const PrefetchJob* prefetch = nullptr;   // optional: the next update's minibatch rows as riders of this launch (batch_rows.h prefetch_rows_direct)
};



// What the kernel receives: the layer table travels BY VALUE in the kernel arguments (read
// with scalar loads through the kernarg segment pointer), so a workgroup's first global
// round trip is already its X / dY rows.  With the table in device memory the kernel
// started with two dependent misses (probe, then the item) — 9.3 us per launch for ~3 MB
// of traffic.  grid = total tiles exactly (dispatch costs ~13 ns per workgroup, even one
// that exits at once); a workgroup finds its layer by a scalar scan of tile_end[].
// GATED tile workgroups (merged phase launches): flag granules {tag, *} they wait for (csrc/dw_body.h)
struct DwGate {
  const unsigned long long* rows = nullptr; int n_rows = 0;   // every producer of X / dY rows has finished
  const unsigned long long* seed = nullptr; int n_seed = 0;   // the per-row seeds (and `late_dY`) are out
  const float* late_dY = nullptr;      // dY buffer that is written with the seeds (the scalar critic's output layer)
  // TWIN critics on one merged launch (TD3): items >= item_split are the second critic's — its seeds, its output layer's dY
  const unsigned long long* seed2 = nullptr; const float* late_dY2 = nullptr; int item_split = 1 << 30;
  unsigned tag = 0; int spin = 0;
  int what_if = 0;                     // timing experiments (oprl_learner_debug_expire, sites 101 ..; tools/what_if.py): 104 the seeds, 105 du, 106 the rows' flags count as seen
  unsigned* err = nullptr; unsigned err_code = 0;
  // GATE == 2 (the ACTOR's tiles on phase 2's launch, csrc/fused_ddpg.hip): no dY rows exist when the tiles start —
  // the actor's backward is linear in the output seed du = dLoss/d(pre-tanh) [B x A], and the two big layers' dY are
  // formed in the tile from du (granules `seed`, kDuLd per row, published by the critic pass at its very end) and from
  // what does not depend on du:
  //   kind 0  output layer:  dY = du
  //   kind 1  second hidden: dY[b, n] = (h2[b, n] > 0) * sum_j du[b, j] W3[j, n]      (h2 rows, W3 snapshot `w3`)
  //   kind 2  first hidden:  dY[b, n] = g1[b, n], granules [16-column tile][B][16] {tag, value} — the critic pass's
  //           members go on, after du, with one tp4-style backward step through the second hidden layer (each member
  //           32 columns, its W2^T shard taken in BEFORE du) and publish the result element by element
  // k_ddpg_update (the whole update as one launch): a tile raises done[tile] = {tag, *} when its stores have been
  // acknowledged (the critic's tiles: the critic pass of the same launch waits for them); null: no flag
  unsigned long long* done = nullptr;
  int kind[4] = {0, 0, 0, 0};          // per item of the launch
  const float* h2 = nullptr;           // [B][256] the actor's second hidden activations (written by the launch before)
  const float* w3 = nullptr;           // the output layer [A][256] (row-major) as it was BEFORE this launch
  const unsigned long long* g1 = nullptr;
  int n_act = 0;
  //   kind 3  first hidden (k_ddpg_chain): dY[b, n] = sum_j du[b, j] G_j[b, n], G = unit-seed rows [A][B][256] the critic
  //           pass's members wrote BEFORE the pass (flags gu_flags[8 x slices] {tag, *}): the tile waits for du only
  const float* gu = nullptr;
  const unsigned long long* gu_flags = nullptr;
  int n_gu_flags = 0;
};
constexpr int kDuLd = 8;               // du granules per minibatch row (action_dim <= 8)

constexpr int kDwMaxItems = 20;        // 5 critics x 4 layers (TQC)
template <int NI>
struct DwKArgsN {
  static constexpr int kItems = NI;
  int tile_end[NI];                    // exclusive prefix ends, relative to this launch
  DwItem items[NI];
  int n_items, B, n_part, dy_tiled;
  AdamScalars ad;
  long long* trace;
  int use_row_scale;
  const float* one;                    // device word holding 1.0f (row scale of unscaled layers)
  int apply_only;
  DwXchg xchg;                         // k_dw_adam<true> only
  AlphaJob alpha;                      // workgroup `tile_end[n_items - 1]` (one past the tiles) runs it
  DwGate gate;                         // dw_adam_body<*, GATED> only
};
using DwKArgs = DwKArgsN<kDwMaxItems>;
// the packed learners' launches (oprl_group_step_n) read N of these from device memory, uploaded per update: a DDPG
// net's three layers need 1.4 KB of the 5 KB
constexpr int kDwGroupItems = 4;        // one net of up to four layers
using DwKArgsG = DwKArgsN<kDwGroupItems>;
constexpr int kDwGroupItems2 = 8;       // twin critics (TD3 / SAC members)
using DwKArgsG2 = DwKArgsN<kDwGroupItems2>;

// the same for the tiles of dw_tile_x2.h (PrecX2 learners: at most four layers per net), a quarter of the bytes: the
// whole-update launch carries two of them beside its DdpgArgs
constexpr int kDwFusedItems = 4;
struct DwKArgs4 {
  int tile_end[kDwFusedItems];
  DwItem items[kDwFusedItems];
  int n_items, B, n_part, dy_tiled;
  AdamScalars ad;
  long long* trace;
  DwGate gate;
  DwXchg xchg;                         // world > 1: the tile all-reduces its gradient with the other ranks' before Adam (dw_tile_x2.h)
};

struct BatchSrc {
  // mode 0: caller-supplied rows
  const float *s, *a, *r, *d, *s2;
  // mode 1: gather from the replay (reference: buffers/episodic_buffer.py:114-133)
  int gather;
  const float *states, *actions, *rewards, *dones;
  const int* ends;
  int n_eps, L;
  long n_transitions;
  unsigned long long seed, counter;
};

// The next update's minibatch rows gathered by riding workgroups of a layer-by-layer launch (batch_rows.h
// prefetch_rows_body, layerwise.hip k_lw_dact): one workgroup per 16-row slice
struct PrefetchJob {
  BatchSrc next;       // gather = 1; s .. s2: where the rows go
  int S, A, B;
  int z0;              // the riders are blockIdx.z == z0 of their host launch (< 0: no job)
};

struct DdpgArgs {      // the fused DDPG / TD3 update (csrc/fused_ddpg.hip)
  Net actor, actor_t, critic, critic_t;
  // TD3 (n_critics == 2): the twin critic, its exchange buffers, target-policy smoothing
  // (td3.py:83-93) and whether this update also steps the actor (policy_freq)
  Net critic2, critic2_t;
  int n_critics;                       // 1 (DDPG) or 2 (TD3)
  int do_actor;                        // 0: role C (actor forward) has nothing to do in this update
  int twin_split;                      // 1: target critic 2 runs in the role-C cluster, beside role A's target critic 1
  int smooth;                          // 1: a' = clip(tanh(mu) + clip(sigma * N(0,1), +-c), +-max_action)
  float policy_noise, noise_clip, max_action;
  const float* noise;                  // injected N(0,1) draws [B][A], or null: Philox(rng_seed, rng_ctr)
  unsigned long long rng_seed, rng_ctr;
  float* c2X[kMaxLayers];              // critic 2 layer inputs / pre-activation grads (ld as critic 1)
  float* c2dY[kMaxLayers];
  // SAC (sac == 1, twin critics, no target actor): both actor passes end in the tanh-Gaussian head.
  // Role A samples a' ~ pi(s') from the ONLINE actor (draws: noise / stream rng_seed) and subtracts
  // alpha log pi(a'|s') from the twin-min target (sac.py:90-104); role C samples pi(s) (draws:
  // noise_pi / stream rng_seed_pi) and leaves the raw head output and log pi for phase 2 and the
  // temperature step (sac.py:118-141)
  int sac;
  int p2_pair;                         // phase 2: the twin critics on two clusters per slice, side by side (both fit the chip)
  const float* noise_pi; unsigned long long rng_seed_pi;
  const double* log_alpha; float alpha_const;   // alpha = exp(*log_alpha) or the constant
  float* raw;                          // [B][2A] actor head output (mean | log_std) at s
  float* logp;                         // [B] log pi(pi(s) | s)
  int B, S, A;
  BatchSrc src;
  // step_n: phase 2 carries one extra row of workgroups that gathers the NEXT update's
  // minibatch (its Philox counter is known) into the staging rows next.s/a/r/d/s2 (writable
  // here), so the ends-table -> search -> random-row chain leaves phase 1's critical path
  BatchSrc next;
  int prefetch_next;
  // ... or PHASE 1 carries that row, as the last row of its grid, into the OTHER of two staging sets (step_n with the
  // merged phase 2: DdpgArgs::merged bit 1)
  int prefetch_p1;
  float gamma, inv_B;
  float* cX[kMaxLayers]; int cldx0;    // critic layer inputs ([s|a], h1, h2) for dW
  float* cdY[kMaxLayers]; int clddo;   // critic pre-activation grads
  float* aX[kMaxLayers]; int aldx0;    // actor layer inputs (s, h1, h2)
  float* adY[kMaxLayers]; int alddo;
  float* pi;                           // [B][A]
  float *y_out, *q_out;                // [B] diagnostics / parity
  unsigned long long* y_granules;      // [3][gran_stride] {epoch<<32 | float bits}: [0] TD target, role A -> B roles (generic passes);
                                       // [1 + j] q of online critic j, role B_j -> role A (lean passes: A publishes the seeds itself)
  int gran_stride;
  unsigned epoch;                      // monotonically increasing per update, never 0
  int bf16;                            // 1: the nets' pf / pb are bf16 packs and the lean passes run PrecBF16 (engine.h)
  int x2;                              // 1: ... are PrecX2 packs (two fp16 planes) and the lean passes run PrecX2
  int nc;                              // CUs per slice cluster (tensor-parallel, csrc/tp3.h): 1, 2 or 4
  int xnc;                             // members an exchange area of `xbuf` is laid out for (>= nc; 8 when wide clusters may run)
  int wide;                            // bit 0: phase 1's role A, bit 1: phase 2's critic pass on clusters of EIGHT (fp32 lean passes)
  // merged launches: the critic's dW + Adam tiles ride on phase 1's launch as extra workgroups (bit 0) — they
  // start while the roles run, take their Adam state in, and wait for flag granules: gate_flags[0 .. 4 slices)
  // = role B's members have written their X / dY rows through, gate_flags[64 + slice] = role A has written the
  // slice's seeds through; tagged with `epoch`
  // bit 1 (PrecX2 learners): the ACTOR's dW + Adam tiles ride on phase 2's launch: the critic pass publishes
  // du = da (1 - pi^2) as granules and the tiles of the two big layers form their dY from it (DwGate, GATE == 2); the
  // first layer's come from the pass's own backward step, granules `g1_granules`
  int merged;
  unsigned long long* gate_flags;
  const float* actor_pb1_f32;          // k_ddpg_chain<PrecBF16>: the online actor's second hidden layer as its fp32 W^T fragment pack — the unit-seed
                                       // rows of the actor's first layer are formed in exact fp32 (the 16-bit packs of a bf16 learner: 8 mantissa bits)
  unsigned long long* seed2_granules;  // [B] {epoch, seed} of the SECOND critic for the tiles of a merged launch (TD3); the first one's: y_granules[0 ..]
  unsigned long long* du_granules;     // [B][kDuLd] {epoch, du}
  unsigned long long* g1_granules;     // [16 tiles][B][16] {epoch, dz1 of the actor}
  float* gu;                           // k_ddpg_chain: [A][B][256] unit-seed dz1 rows of the actor (DwGate kind 3)
  unsigned long long* gu_flags;        // [8 x slices] {epoch, *}: a pass member's share of them is out
  // k_ddpg_update: phase 1's roles, the critic's tiles, phase 2's critic pass and the actor's tiles in ONE
  // launch.  What a role hands to a later one crosses no kernel boundary: it lives in uncached memory (the learner's
  // workspace and fp16 packs), is written before a flag {epoch, *} — w_flags[0, 64): role C's members, the critic's tiles
  // raise DwGate::done — and read after the flag with coherent loads (engine.h ld4c).
  int group_span;                      // group launches: XCDs a member's slices are dealt out to (1, 2, 4 or 8; fused_ddpg.hip k_ddpg_phase1_group)
  int whole;
  unsigned cluster_tag2;               // the critic pass's exchange tag (phase 2's cluster_tag in the two-launch form)
  long long* trace2;                   // ... and its trace slot
  unsigned long long* w_flags;
  unsigned long long* ct_done; int n_ct;   // the critic's tiles' completion flags
  const float* critic_b16[kMaxLayers]; // the critic's biases as its tiles of this launch leave them (uncached copies)
  float* w3_snap;                      // the actor's output layer [A][256] as it is before phase 2, copied by phase 1's role C (slice 0)
  const float* w3_src;                 // ... from here (the row-major master)
  int no_lean;                         // 1: never use the tp4.h specialisation (OPRL_AMD_NO_LEAN, tests)
  int rt2;                             // (2: ... and SAC's role A carries role C's pass — the same actor on s — as a second row tile: no role C rows)  1: phase 1's B roles carry TWO row tiles (32 rows) per cluster (k_ddpg_phase1_rt2: over-subscribed launches, B a multiple of 256)
  int xcd_local;                       // 1: a slice cluster whose members all see their expected XCD may publish its granules at workgroup scope
                                       // (the HOST's decision — the dispatcher was probed, OPRL_AMD_NO_XCD_LOCAL, no expired cluster wait so far —
                                       // and the same for every member: fused_ddpg.hip cluster_on_one_xcd)
  unsigned long long* xbuf;            // cluster exchange areas: [role][slice][kTpStages][nc][kTpBlk] granules
  unsigned cluster_tag;                // launch-unique
  long cdY0_stride, adY0_stride;       // floats between the members' dz1 partial buffers
  float *partials_c, *partials_a;      // [slices][4]
  long long* trace;
  unsigned* err;                       // host-visible error word: an expired bounded wait stores (kernel << 8 | site) there
  int debug_expire;                    // test hook: this wait site (WaitSite, tp3.h) gives up at once
};

// k_ddpg_chain (csrc/fused_ddpg.hip): what changes from update to update inside a launch that runs several
constexpr int kChainMax = 32;
struct ChainArgs {
  int n_upd;                           // updates in this launch (<= kChainMax)
  int first_gather;                    // 1: update 0 gathers its own rows from the replay (nothing staged them)
  int pf_last;                         // 1: the last update stages the next rows as well (another launch follows)
  int trace_u;                         // trace build: the update whose stages are stamped
  int rows;                            // grid rows per update: 16 (roles A 8 | B 4 | C 4) + the tile-only rows of small batches
  int order;                           // dispatch order of an update's role rows: 0 A | B | C, 1 B | A | C, 2 B | C | A, 3 A | C | B (fused_ddpg.hip chain_row)
  float c_step[kChainMax], c_bc2[kChainMax];   // Adam's lr / (1 - b1^t) and sqrt(1 - b2^t) of the critic's step, per update
  float a_step[kChainMax], a_bc2[kChainMax];   // ... of the actor's
  const float* set0[5];                // staging rows s, a, r, d, s2: update u reads set (u & 1), its prefetch fills the other
  const float* set1[5];
  unsigned long long* ct_fin;          // [critic tiles] {epoch, *}: every store of the tile acknowledged
  unsigned long long* at_fin;          // [actor tiles]
  unsigned long long* pf_done;         // [slices]: the rows of the update after `epoch` are staged
  const float* b16[4][kMaxLayers];     // uncached bias copies the tiles leave: 0 actor, 1 actor target, 2 critic, 3 critic target
  float* w3buf[2];                     // the actor's output layer [A][256] before update u: w3buf[u & 1]
  unsigned long long* qp;              // [slices][4 members][16 rows] {epoch, partial q}: role B -> role A (tp4.h QPart)
};

// k_ddpg_chain: tile-only grid rows of one update.  Roles B and C (8 workgroups per slice) go on as tile workgroups; `mt`
// tiles (the larger of the two nets' counts) need workgroups, and so do the `sl` gatherers of the next update's rows —
// workgroups WITHOUT a tile: a gatherer that also carries a tile starts it ~3 us late, and the critic pass of the update
// waits for the last critic tile (B = 128, the reference scripts' batch: 30.5 -> us per update before this rule, r06-4)
__host__ __device__ inline int chain_tile_rows(int mt, int sl) {
  const int need = mt + sl;
  return need > 8 * sl ? (need - 8 * sl + sl - 1) / sl : 0;
}

constexpr int kDwTile = 32;      // k (fan-in) extent of a dW tile
constexpr int kDwTileN = 16;     // n (fan-out) extent: 16 rows keep a workgroup's bytes at X 32 KB + dY 16 KB
constexpr int kDwThreads = 512;
constexpr int kDwWaves = 8;
constexpr int kTraceStamps = 24;
// in-kernel stage stamps (tools/trace_slice.py): compiled into liboprl_amd_trace.so only (-DOPRL_TRACE) — in the production
// library the stamp sites vanish (5 KB of cold code inside the hot paths of a 113 KB kernel)
#ifdef OPRL_TRACE
constexpr bool kTraceOn = true;
#else
constexpr bool kTraceOn = false;
#endif

// csrc/policy_act.hip: one observation through a net's row-major master weights; obs / out / ticket are host-mapped
constexpr int kPolicyActMaxWidth = 512;
struct PolicyActArgs {
  int n_layers;
  int dims[kMaxLayers + 1];
  const float* w[kMaxLayers];
  const float* b[kMaxLayers];
  const float* obs;
  unsigned long long* out;             // [n_out] {ticket, value}
  unsigned ticket_value;
};
hipError_t launch_policy_act(const PolicyActArgs& a, hipStream_t st);

}  // namespace oprl
