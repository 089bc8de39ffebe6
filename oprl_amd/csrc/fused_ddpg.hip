// fused_ddpg.hip — the DDPG update() as two slice kernels + two dW/Adam launches.
//
// The generic path (learner.hip) spends 6 k_mlp_slice launches + a gather per
// update; every launch pays dispatch, an input-load latency chain and an LDS
// re-zero, and the actor's forward — which does not depend on the critic step —
// sits on the critical path.  Here:
//
//   Every role below is a CLUSTER of nc = 4 workgroups sharing its 16-row slice
//   tensor-parallel (csrc/tp3.h; csrc/tp4.h = the instruction-lean specialisation for
//   nc = 4, width 256, used whenever the shapes allow): the 256 KB hidden layer is split by
//   columns, only [16 x N<=48] partial outputs are exchanged.
//
//   k_ddpg_phase1, grid (slices, 3*nc) — three concurrent roles per 16-row slice, each
//   gathering the same rows:
//     A  actor_target(s') -> critic_target(s', a') -> y = r + (1-d) gamma q'
//        -> y handed to role B as 8-byte {epoch, value} granules   (ddpg.py:94-95)
//     B  critic(s, a) forward (concurrently with A) -> wait for y -> 2(q-y)/B
//        -> critic backward                                        (ddpg.py:96-100)
//     C  actor(s) forward, activations and pi = tanh(.) to HBM     (first half of ddpg.py:104)
//   All 3*slices workgroups are co-resident (48 on 256 CUs at B=256); B's wait is a
//   bounded spin.
//   k_dw_adam(critic)          dW + Adam + Polyak     (ddpg.py:99-101, 72-77)
//   k_ddpg_phase2, grid (slices):
//            critic(s, pi) forward with the UPDATED critic -> -1/B -> critic
//            backward to the action columns -> du = da (1 - pi^2) -> actor
//            backward                                  (ddpg.py:103-107)
//   k_dw_adam(actor)           dW + Adam + Polyak     (ddpg.py:105-107, 79-84)
//
// TD3 runs the same two kernels with twin critics (roles A | B1 | B2 | C, target smoothing in A);
// SAC as well (template flag): both actor passes end in the tanh-Gaussian head, role A uses the
// online actor and subtracts alpha log pi, phase 2 runs both critics and routes the smaller q's
// action gradient through the head's backward (sac.py:90-141).  The temperature step stays a
// separate small launch (k_alpha_step).
//
// Arithmetic and summation order are those of the generic kernels (same
// engine.h routines, same seeds), so the two paths agree bit for bit
// (tests/test_gpu_fused.py).  The minibatch either comes from caller pointers
// (update()) or is gathered in-kernel from the HBM replay with the same Philox
// draw and index map as k_replay_gather (step_n).
#include <cstddef>
#include <mutex>
#include <type_traits>
#include "kernels.h"
#include "philox.h"
#include "replay_index.h"
#include "batch_rows.h"
#include "slice_head.h"
#include "tp4.h"
#include "dw_body.h"
#include "dw_tile_x2.h"

namespace oprl { constexpr bool kDwTileX2 = true; constexpr bool kMergedTile64 = true; }      // (kMergedTile64: every merged launch rides 16 x 64 tiles, whatever the arithmetic)    // PrecX2 learners: the 16 x 64 split-product tile of dw_tile_x2.h in the merged launches

namespace oprl {

// (kMaxEnds: batch_rows.h)

// one MLP pass of a cluster: the lean tp4 routines or the generic tp3 ones
// Do the members of this workgroup's slice cluster (same blockIdx.x, different blockIdx.y) share an XCD?  Workgroups go to
// XCD (linear id) % 8 and the grid is (slices, rows): with slices a multiple of eight a slice's workgroups all land on XCD
// blockIdx.x % 8.  Then the cluster's granules may be published at workgroup scope (tp3.h Tp::local: the stores stop in the
// XCD's L2, where the peers' agent-scope polls find them; A / B on one box: 25.94 against 26.10 us per update, r04-26).
// The decision must be the SAME for every member — one that published locally while a peer sits on another XCD would
// never be seen by it — so it is made in three layers: (1) the host asks for it (DdpgArgs::xcd_local) only after a PROBE
// launch of the same grid shape found every workgroup of a column on XCD column % 8 (xcd_map_ok below: the dispatcher
// deals round robin over eight XCDs — not on a partitioned / masked device), not under OPRL_AMD_NO_XCD_LOCAL=1 and not
// after an expired cluster wait of this learner; (2) the test depends on blockIdx.x and the hardware's XCC_ID only: members
// that do share an XCD all answer alike; (3) a workgroup on an unexpected XCD publishes at agent scope.
__device__ __forceinline__ bool cluster_on_one_xcd(const DdpgArgs& A) {
  const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;          // HW_REG_XCC_ID, bits 3:0
  return A.xcd_local != 0 && ((int)gridDim.x & 7) == 0 && xcc == ((int)blockIdx.x & 7);
}
// the probe: every workgroup of a (16, 16) grid reports its XCD
__global__ void k_xcc_probe(int* out) {
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;
}

// (`pre`: the caller's row staging, overlapped with the lean pass's fragment requests — tp4.h; generic passes: run first)
template <int WIDTH, bool LEAN, class P, int NM = 4, class ST, class PRE = NoStamp>
__device__ __forceinline__ void tp_fwd(const Net& net, const float* x0s, float* h1, float* h2, float* outS,
                                       float* scr, Tp& tp, const Tp3Store& st, int row0, int B, ST sf, PRE pre = PRE()) {
  if constexpr (LEAN) tp4_forward<P, NM>(net, x0s, h1, h2, outS, scr, tp, st, row0, B, sf, BiasOv(), pre);
  else { pre(); tp3_forward<WIDTH>(net, x0s, h1, h2, outS, scr, tp, st, row0, B, sf); }
}
template <int WIDTH, bool LEAN, class P, class ST>
__device__ __forceinline__ void tp_bwd(const Net& net, const float* doutS, float* h1, float* h2, float* scr,
                                       Tp& tp, const Tp3Store& st, int row0, int B, int dact_col0,
                                       int dact_cols, float* dactS, ST sf) {
  if constexpr (LEAN) tp4_backward<P>(net, doutS, h1, h2, scr, tp, st, row0, B, dact_col0, dact_cols, dactS, sf);
  else tp3_backward<WIDTH>(net, doutS, h1, h2, scr, tp, st, row0, B, dact_col0, dact_cols, dactS, sf);
}

template <int WIDTH>
struct FusedLds {   // floats
  static constexpr int WL = lds_ld(WIDTH);
  static constexpr int xa = 0;                       // [s | a] (phase 2: [s | pi])
  static constexpr int xb = xa + kR * kX0Ld;         // [s' | a']
  static constexpr int h = xb + kR * kX0Ld;          // 5 hidden buffers
  static constexpr int out = h + 5 * kR * WL;
  static constexpr int aux = out + kR * kOutLd;
  static constexpr int aux2 = aux + kR * kOutLd;     // SAC phase 2: the twin critic's action gradient
  static constexpr int scr = aux2 + kR * kOutLd;
  static constexpr int misc = scr + kWaves * kR * 16;   // r[16] d[16] y[16] ep[16] t[16] + ends
  static constexpr int total = misc + 96 + kMaxEnds;
};

// one tagged 8-byte granule: the value is its own flag (as the cluster exchanges of tp3.h / tp4.h)
__device__ __forceinline__ void granule_put(unsigned long long* g, unsigned tag, float v) {
  __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float granule_get(const unsigned long long* g, unsigned tag, unsigned* err = nullptr,
                                             unsigned code = 0) {
  for (int spin = 0; spin < kTpSpin; ++spin) {
    const unsigned long long x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(x >> 32) == tag) return __uint_as_float((unsigned)x);
    __builtin_amdgcn_s_sleep(1);
  }
  report_expired(err, code);   // bounded: a lost partner is reported to the host and shows up as NaN, not as a hang
  return __builtin_nanf("");
}

// SAC: tanh-Gaussian head over outS = [mean | log_std] (16 lanes per row, the first 256 threads, as
// slice_head.h does it): the action goes to aS[row * a_ld + col] (LDS) and/or pi_g, the row's log pi
// to lpS[row] (LDS) and/or logp_g, the raw head output to raw_g.  No barrier inside.
__device__ __forceinline__ void gauss_head(const float* outS, int row0, int B, int Ad, const float* noise,
                                           unsigned long long seed, unsigned long long ctr, float* aS, int a_ld,
                                           float* lpS, float* pi_g, float* raw_g, float* logp_g,
                                           unsigned long long* a_gran = nullptr, unsigned a_tag = 0) {
  // a_gran (optional): the action also goes out as granules [row][col] (zeros for rows beyond B)
  const int tid = threadIdx.x;
  if (tid >= kR * 16) return;
  const int row = tid >> 4, sub = tid & 15, gr = row0 + row;
  float lp = 0.f;
  if (gr >= B && a_gran != nullptr)
    for (int col = sub; col < Ad; col += 16) granule_put(a_gran + row * Ad + col, a_tag, 0.f);
  if (gr < B) {
    for (int col = sub; col < Ad; col += 16) {
      const float mu = outS[row * kOutLd + col];
      const float lsr = outS[row * kOutLd + Ad + col];
      const float e = noise != nullptr ? noise[(size_t)gr * Ad + col] : philox_normal(seed, ctr, (unsigned)gr, (unsigned)col);
      float a;
      lp += gauss_elem(mu, lsr, e, &a);
      if (a_gran != nullptr) granule_put(a_gran + row * Ad + col, a_tag, a);
      if (aS != nullptr) aS[row * a_ld + col] = a;
      if (pi_g != nullptr) pi_g[(size_t)gr * Ad + col] = a;
      if (raw_g != nullptr) {
        raw_g[(size_t)gr * 2 * Ad + col] = mu;
        raw_g[(size_t)gr * 2 * Ad + Ad + col] = lsr;
      }
    }
  }
  lp = row16_sum(lp);
  if (sub == 0 && gr < B) {
    if (lpS != nullptr) lpS[row] = lp;
    if (logp_g != nullptr) logp_g[gr] = lp;
  }
}

template <int WIDTH, bool LEAN, class P, class ST, class PRE>
__device__ __forceinline__ void role_b(const DdpgArgs& A, const Net& critic, float* const* cX, float* const* cdY,
                                       float* partials, bool diag, int j, float* smem, Tp& tp, ST& stamp, int slice, PRE& pre) {
  using LY = FusedLds<WIDTH>;
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  float* xa = smem + LY::xa;
  float* h1 = smem + LY::h;
  float* h2 = h1 + HB;
  float* outS = smem + LY::out;
  float* auxS = smem + LY::aux;
  float* scr = smem + LY::scr;
  float* yS = smem + LY::misc + 2 * kR;
  const int row0 = slice * kR, B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x;
  const bool lead = tp.c == 0;
  const bool wt = LEAN && (A.merged & 1) != 0;   // the dW tiles of this very launch read what this role stores
  const Tp3Store st{cX[1], cX[2], cdY[1], cdY[0], A.cdY0_stride, LEAN ? B : 0, wt};   // lean: tile-major dz1 partials
  if constexpr (LEAN) {
    // ... and, the critic being scalar-output, its whole backward with unit seed as well:
    // k_dw_adam applies 2(q - y)/B per row (tp4_scalar_fb), so after y arrives only that
    // vector is left to publish
    tp4_scalar_fb<P>(critic, xa, h1, h2, h1 + 2 * HB, outS, scr, tp, st, row0, B, 1.f, 0, 0, nullptr, stamp, nullptr, BiasOv(), QPart(), pre);
  } else {
    tp_fwd<WIDTH, LEAN, P>(critic, xa, h1, h2, outS, scr, tp, st, row0, B, stamp, pre);
  }
  if (lead) {
    if (wt) {
      for (int idx = tid; idx < kR * (S + Ad); idx += kThreads) {
        const int row = idx / (S + Ad), col = idx - row * (S + Ad), gr = row0 + row;
        if (gr < B) __hip_atomic_store(cX[0] + (size_t)gr * A.cldx0 + col, xa[row * kX0Ld + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      store_rows(xa, kX0Ld, cX[0], A.cldx0, S + Ad, row0, B);
    }
  }
  if constexpr (LEAN) {
    if (wt) {
      // every wave's rows are out (written through) before the member says so: one flag granule per member
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0)
        __hip_atomic_store(A.gate_flags + ((size_t)j * gridDim.x + slice) * 4 + tp.c, (unsigned long long)A.epoch << 32, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);   // (critic j's block of flags: twin critics on a merged launch)
    }
    // q goes to role A of this slice as granules — A, the last to finish, turns it into the per-row seed
    // 2 (q - y) / B and the diagnostics itself; this role is done (it used to wait here for y: one more hop
    // and a wake-up on the launch's critical path)
    if (!lead || tid >= 64) return;
    const int gr = row0 + tid;
    if (tid < kR && gr < B)
      __hip_atomic_store(A.y_granules + (size_t)(1 + j) * A.gran_stride + gr,
                         ((unsigned long long)A.epoch << 32) | (unsigned long long)__float_as_uint(outS[tid * kOutLd]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamp();   // q published
    return;
  }
  // wait for this slice's TD targets: lanes 0..15 of wave 0 poll their granule (relaxed,
  // L1-bypassing) with a sleep in between; the spin is BOUNDED — on give-up the target
  // becomes NaN, which the parity tests and the loss diagnostics expose, instead of a hang.
  if (tid < 64) {
    float y = 0.f;
    if (tid < kR && row0 + tid < B) {
      unsigned long long g = 0;
      bool ok = false;
      const int lim = A.debug_expire == (int)SITE_TD_TARGET ? 0 : (1 << 20);
      for (int spin = 0; spin < lim; ++spin) {
        g = __hip_atomic_load(A.y_granules + row0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (unsigned)(g >> 32) == A.epoch;
        if (ok) break;
        __builtin_amdgcn_s_sleep(8);
      }
      if (!ok) report_expired(A.err, (KERN_PHASE1 << 8) | SITE_TD_TARGET);
      y = ok ? __uint_as_float((unsigned)g) : __builtin_nanf("");
    }
    if (tid < kR) yS[tid] = y;
  }
  stamp();   // TD target received
  // ---- seed 2(q - y)/B, diagnostics (every member computes the same seed)
  lds_zero(auxS, kR * kOutLd);
  __syncthreads();
  float p_loss = 0.f, p_q = 0.f, p_y = 0.f;
  if (tid < kR) {
    const int gr = row0 + tid;
    if (gr < B) {
      const float q = outS[tid * kOutLd], y = yS[tid];
      auxS[tid * kOutLd] = 2.f * (q - y) * A.inv_B;
      if (lead && diag && A.y_out != nullptr) A.y_out[gr] = y;
      if (lead && diag && A.q_out != nullptr) A.q_out[gr] = q;
      p_loss = (q - y) * (q - y);
      p_q = q;
      p_y = y;
    }
  }
  if (partials != nullptr && lead) {
    __syncthreads();
    float v[3] = {p_loss, p_q, p_y};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) v[k] += __shfl_xor(v[k], m);
      if ((tid & 63) == 0) scr[(tid >> 6) * 4 + k] = v[k];
    }
    __syncthreads();
    if (tid < 3) {
      float sum = 0.f;
      for (int w = 0; w < kWaves; ++w) sum += scr[w * 4 + tid];
      partials[slice * 4 + tid] = sum;
    }
  }
  __syncthreads();
  if (lead) store_rows(auxS, kOutLd, cdY[2], A.clddo, 1, row0, B);
  stamp();
  if constexpr (!LEAN) tp_bwd<WIDTH, LEAN, P>(critic, auxS, h1, h2, scr, tp, st, row0, B, 0, 0, auxS, stamp);
}

// ---- role B on 32-row slices (tp4.h tp4_scalar_fb2): the over-subscribed phase-1 launches.  LDS of such a workgroup:
struct FusedLdsB2 {   // floats
  static constexpr int TX = kR * kX0Ld, TH = kR * lds_ld(256), TO = kR * kOutLd, TS = kWaves * 256;
  static constexpr int xa = 0;                     // [2][kR][kX0Ld] the two row tiles' [s | a]
  static constexpr int dump = xa + 2 * TX;         // what load_batch writes beside the second tile (its [s' | 0])
  static constexpr int h = dump + TX;              // h1 | h2 | g2, two tiles each
  static constexpr int out = h + 6 * TH;
  static constexpr int scr = out + 2 * TO;         // (the gather path's episode table lies over it, before the pass)
  static constexpr int misc = scr + 2 * TS;        // r[16] d[16] (unused by this role) + meta
  static constexpr int total = misc + 96;
};
static_assert(FusedLdsB2::total * 4 <= 160 * 1024, "role B on two row tiles fits the LDS");
static_assert(2 * FusedLdsB2::TS >= kMaxEnds, "the episode table fits over the partial tiles");

template <class P, class ST>
__device__ __forceinline__ void role_b2(const DdpgArgs& A, const Net& critic, float* const* cX, float* const* cdY, int j, float* smem,
                                        int member, int s32, ST& stamp) {
  using LY = FusedLdsB2;
  float* xa = smem + LY::xa;
  float* h1 = smem + LY::h;
  float* h2 = h1 + 2 * LY::TH;
  float* g2 = h2 + 2 * LY::TH;
  float* outS = smem + LY::out;
  float* scr = smem + LY::scr;
  float* rS = smem + LY::misc;
  int* meta = reinterpret_cast<int*>(rS + 2 * kR);
  int* endsS = reinterpret_cast<int*>(scr);
  const int row0 = s32 * 2 * kR, B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x;
  const size_t area = (size_t)kTpStages * A.xnc * kTpBlk;
  Tp tp{member, 4, A.xbuf + ((size_t)(1 + j) * gridDim.x + 2 * s32) * area, A.cluster_tag, 0,
        A.err, KERN_PHASE1 << 8, A.debug_expire == (int)SITE_CLUSTER ? 0 : kTpSpin};
  // (the launcher takes this form only where the four members of a cluster sit 8 k workgroups apart: on one XCD, and all
  // of them see the same blockIdx.x & 7 — cluster_on_one_xcd's rule)
  tp.local = cluster_on_one_xcd(A);
  const bool lead = member == 0;
  const Tp3Store st{cX[1], cX[2], cdY[1], cdY[0], A.cdY0_stride, B, false};
  auto rows = [&]() {
    if (A.src.gather) {
      // (the first update of a step_n call: the two tiles one after the other through load_batch, whose [s' | 0] half lands on
      // the next tile's place / the dump)
      for (int t = 0; t < 2; ++t) {
        load_batch(A.src, row0 + kR * t, B, S, Ad, xa + t * LY::TX, xa + (t + 1) * LY::TX, rS, rS + kR, meta, endsS);
        __syncthreads();
      }
    } else {
      // staged rows: every element of the two padded tiles written once — value or zero — loads first, no barrier
      constexpr int kPer = (2 * LY::TX + kThreads - 1) / kThreads;
      float v[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int idx = min(tid + q * kThreads, 2 * LY::TX - 1);
        const int row = idx / kX0Ld, col = idx - row * kX0Ld, gr = row0 + row;
        const bool is_s = gr < B && col < S, is_a = gr < B && col >= S && col < S + Ad;
        const float* src = A.src.s;
        if (is_s) src = A.src.s + (size_t)gr * S + col;
        if (is_a) src = A.src.a + (size_t)gr * Ad + (col - S);
        const float x = *src;
        v[q] = (is_s || is_a) ? x : 0.f;
      }
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int idx = tid + q * kThreads;
        if (idx < 2 * LY::TX) xa[idx] = v[q];
      }
    }
  };
  tp4_scalar_fb2<P>(critic, xa, h1, h2, g2, outS, scr, tp, area, st, row0, B, 1.f, stamp, rows);
  if (!lead) return;
#pragma unroll
  for (int t = 0; t < 2; ++t) store_rows(xa + t * LY::TX, kX0Ld, cX[0], A.cldx0, S + Ad, row0 + kR * t, B);
  // q -> role A of the two slices, as granules (role_b's hand-over)
  if (tid < 2 * kR) {
    const int gr = row0 + tid;
    if (gr < B)
      __hip_atomic_store(A.y_granules + (size_t)(1 + j) * A.gran_stride + gr,
                         ((unsigned long long)A.epoch << 32) | (unsigned long long)__float_as_uint(outS[(tid >> 4) * LY::TO + (tid & 15) * kOutLd]),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamp();   // q published
}

// (A separate template instance for the twin-critic algorithms, so that the single-critic kernel
// carries none of their code, was measured SLOWER for all three: phase 1 15.0 vs 14.3 us for DDPG,
// TD3 38.5 vs 36.0 us, SAC 54.5 vs 51.6 us per update — profiles/r01b_experiments.txt #29.)
// WIDE (DDPG, fp32 lean passes, a grid that still fits the chip): role A — two forward passes, the launch's
// critical chain — runs on clusters of EIGHT CUs (tp4_forward<P, 8>): half the 256 x 256 layer's bytes and
// MFMAs per member; roles B and C keep four.
// MERGED (DDPG, lean passes, B <= 256): the critic's dW + Adam tiles are extra grid rows of THIS launch
// (dw_adam_body<false, GATED>): dispatched after the roles' workgroups, they run where CUs are free or come free
// (role C's after 7 us, role B's after 8.5), take in their Adam state and — once role B's members have flagged
// their rows — X and dY, and wait for role A's seed flags: what is left after the launch's critical chain is one
// flag hop, 16 MFMAs and the Adam epilogue instead of a kernel boundary and a whole k_dw_adam launch.
__device__ __forceinline__ int dw_total(const DwKArgs& d) { return d.tile_end[kDwMaxItems - 1]; }
__device__ __forceinline__ int dw_total(const DwKArgs4& d) { return d.tile_end[kDwFusedItems - 1]; }

// (`by`: this workgroup's row of the roles' / tiles' block — blockIdx.y unless the caller runs the body inside a larger
// grid)
// Returns the critic tile this workgroup goes on with (merged launches; -1: none): the caller runs it — ONE inlined copy
// of the tile code per kernel instead of one per place a workgroup may turn into a tile (instruction cache, r03-14 / -25).
// MERGED kernels are single-critic, non-SAC by their launchers' rules: the twin paths are compiled out of them.
// (SINGLE: the caller's launcher admits one critic only — the merged kernels, the packed learners' group kernel)
template <int WIDTH, bool LEAN, bool SAC, class P, bool WIDE = false, bool MERGED = false, class KA = DwKArgs, bool SINGLE = MERGED, bool RT2 = false>
__device__ __forceinline__ int ddpg_phase1_body(const DdpgArgs& A, const KA* D = nullptr, int by_in = -1, int bx_in = -1) {   // (bx_in: the slice, where the caller deals the workgroups out itself — the group launches)
  constexpr bool TWIN = !SINGLE;            // twin critics / twin_split can occur at all
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int by = by_in < 0 ? (int)blockIdx.y : by_in;
  if constexpr (RT2) {
    // B roles on 32-row slices: critic j's clusters are grid rows 2 j, 2 j + 1 — member-major (member m of slice s32 is
    // workgroup m * slices / 2 + s32 of the pair of rows: the four members sit 8 k workgroups apart, on one XCD); roles A and
    // C follow on 16-row slices, as ever
    static_assert(LEAN && !MERGED && !WIDE, "two row tiles: the plain lean phase launch");
    const int yb2 = 2 * A.n_critics, half = (int)gridDim.x >> 1;
    if (by < yb2) {
      const int local = (by & 1) * (int)gridDim.x + (int)blockIdx.x;
      const int member = local / half, s32 = local - member * half;
      int n_stamp2 = 0;
      auto stamp2 = [&]() {
        if (kTraceOn && A.trace != nullptr && (threadIdx.x & 63) == 0 && (threadIdx.x == 0 || s32 == 0) && member == 0 && n_stamp2 < kTraceStamps) {
          const int slot = threadIdx.x == 0 ? s32 : 16 + ((int)threadIdx.x >> 6);
          long long* tr = A.trace + (((size_t)(1 + (by >> 1)) * 64 + slot) * kTraceStamps + n_stamp2) * 2;
          tr[0] = (long long)__builtin_readcyclecounter();
          tr[1] = (long long)wall_clock64();
        }
        ++n_stamp2;
      };
      stamp2();
      if ((by >> 1) == 1) role_b2<P>(A, A.critic2, A.c2X, A.c2dY, 1, smem, member, s32, stamp2);
      else role_b2<P>(A, A.critic, A.cX, A.cdY, 0, smem, member, s32, stamp2);
      return -1;
    }
    by += A.n_critics * A.nc - yb2;        // roles A and C: the row they have in the one-tile grid
  }
  // the last grid row of a step_n launch may be the PREFETCH row: the next update's rows (dispatched last: these
  // workgroups start as roles retire) into the other staging set — the same load_batch call as the roles', from A.next
  const bool pf_row = A.prefetch_p1 && blockIdx.y == gridDim.y - 1;
  if constexpr (MERGED) {
    const int rows = (2 + A.n_critics) * A.nc + ((LEAN && WIDE) ? 4 : 0);
    if (by >= rows && !pf_row) {
      const int tile = (by - rows) * (int)gridDim.x + (int)blockIdx.x;
      return tile < dw_total(*D) ? tile : -1;
    }
  }
  using LY = FusedLds<WIDTH>;
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  float* xa = smem + LY::xa;
  float* xb = smem + LY::xb;
  float* h1 = smem + LY::h;
  float* h2 = h1 + HB;
  float* outS = smem + LY::out;
  float* auxS = smem + LY::aux;
  float* scr = smem + LY::scr;
  float* rS = smem + LY::misc;
  float* dS = rS + kR;
  float* yS = dS + kR;
  int* meta = reinterpret_cast<int*>(yS + kR);
  int* endsS = reinterpret_cast<int*>(smem + LY::misc + 96);
  const int B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x;
  int slice = bx_in < 0 ? (int)blockIdx.x : bx_in;
  // roles: 0 = A target chain, 1 .. n_critics = B (one per online critic), last = C actor forward.
  // Order in the grid (= dispatch order when the grid over-subscribes the chip): whoever is waited for comes
  // first — generic passes A | B.. | C (the B roles wait for A's TD target), lean passes B.. | A | C (role A
  // waits for the B roles' q; twin_split's mutual A <-> C exchange needs co-residency either way)
  constexpr int NMA = (LEAN && WIDE) ? 8 : 4;          // members of role A's clusters (lean passes)
  const int role_c = 1 + A.n_critics;
  const int nA = (LEAN && WIDE) ? 8 : A.nc;
  int role, member;
  if constexpr (!LEAN) {
    role = by / A.nc;
    member = by % A.nc;
  } else {
    const int y = by, yb = A.n_critics * A.nc;
    if (y < yb) { role = 1 + y / A.nc; member = y % A.nc; }
    else if (y < yb + nA) {
      // (wide: slice-major like every role — all members of a slice on one XCD; member-major ids, eight
      // consecutive workgroups per cluster, measured 1.6 us slower per update.  The price: a launch cut in the
      // middle of this role leaves every slice with half its members resident and spinning, so wide clusters
      // are for learners that do not share the chip with several others — learner.hip's rule)
      role = 0; member = y - yb;
    }
    else { role = role_c; member = y - yb - nA; }
  }
  const int row0 = slice * kR;
  Tp tp{member, role == 0 ? nA : A.nc,
        A.xbuf + ((size_t)role * gridDim.x + slice) * kTpStages * A.xnc * kTpBlk, A.cluster_tag, 0,
        A.err, KERN_PHASE1 << 8, A.debug_expire == (int)SITE_CLUSTER ? 0 : kTpSpin};
  tp.local = bx_in < 0 && cluster_on_one_xcd(A);      // (the launch's own (slice, row) grid: r04-26)
  const bool lead = tp.c == 0;                 // member 0 does the un-sliced global stores
  int n_stamp = 0;
  auto stamp = [&]() {
    // wave 0 of every slice -> slot `slice`; the other 15 waves of slice 0 -> slots 16 + wave
    if (kTraceOn && A.trace != nullptr && (tid & 63) == 0 && (tid == 0 || slice == 0) && lead && n_stamp < kTraceStamps) {
      const int slot = tid == 0 ? slice : 16 + (tid >> 6);
      long long* tr = A.trace + (((size_t)role * 64 + slot) * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();   // entry
  // (both sources live in the kernel-argument segment, this block at its start: addressed through the segment pointer
  // they are read with scalar loads — `pf_row ? &A.next : &A.src` makes hipcc copy the block to scratch; the group
  // kernels, whose blocks sit in device memory, never carry a prefetch row)
  const BatchSrc* srcp = &A.src;
  if (pf_row) srcp = (const BatchSrc*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(DdpgArgs, next));
  // the slice's rows: staged by the first pass of the role, behind that pass's weight-fragment requests (lean passes:
  // the rows' round trip and the fragments' overlap — 1.5 us per workgroup at B = 1024, where a launch is four rounds of
  // workgroups: r05-14); the prefetch row and the generic passes stage them here
  bool rows_in = false;
  auto rows = [&]() {      // (A.src by name: through a pointer the block would be copied to scratch — see above)
    if (!rows_in) load_batch(A.src, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
    rows_in = true;
  };
  if constexpr (!LEAN) {
    load_batch(*srcp, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
    rows_in = true;
  } else if (pf_row) {
    load_batch(*srcp, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
  }
  if (pf_row) {
    __syncthreads();
    store_rows(xa, kX0Ld, const_cast<float*>(A.next.s), S, S, row0, B);
    store_rows(xb, kX0Ld, const_cast<float*>(A.next.s2), S, S, row0, B);
    for (int idx = tid; idx < kR * Ad; idx += kThreads) {
      const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
      if (gr < B) const_cast<float*>(A.next.a)[(size_t)gr * Ad + col] = xa[row * kX0Ld + S + col];
    }
    if (tid < kR && row0 + tid < B) {
      const_cast<float*>(A.next.r)[row0 + tid] = rS[tid];
      const_cast<float*>(A.next.d)[row0 + tid] = dS[tid];
    }
    return -1;
  }
  stamp();   // batch rows requested
  const Tp3Store nostore{nullptr, nullptr, nullptr, nullptr, 0};
  // twin_split (TD3 / SAC, all four roles co-resident): role A evaluates target critic 1 only; the
  // role-C cluster, after its actor pass, receives a' from role A, evaluates target critic 2 and
  // returns q2' — the two target passes run side by side instead of back to back.  Exchange slots:
  // the never-used last stage of the receiving role's cluster area.
  auto x_slot = [&](int to_role) {   // (formed only on the twin_split paths)
    return A.xbuf + (((size_t)to_role * gridDim.x + slice) * kTpStages + (kTpStages - 1)) * A.xnc * kTpBlk;
  };
  const unsigned x_tag = (A.cluster_tag << 6) | 62u;

  if (role == role_c) {
    if (A.do_actor) {   // (TD3: no actor step in every other update)
      // ---- role C: actor(s) forward.  Pack rows beyond S are zero, so [s | a] serves as input.
      const Tp3Store st{A.aX[1], A.aX[2], nullptr, nullptr, 0};
      tp_fwd<WIDTH, LEAN, P>(A.actor, xa, h1, h2, outS, scr, tp, st, row0, B, stamp, rows);
      if constexpr (SAC) {
        // pi(s) ~ tanh-Gaussian, its log-density (temperature step, actor seed) and the raw head output
        if (lead) {
          gauss_head(outS, row0, B, Ad, A.noise_pi, A.rng_seed_pi, A.rng_ctr, nullptr, 0, nullptr, A.pi, A.raw, A.logp);
          store_rows(xa, kX0Ld, A.aX[0], A.aldx0, S, row0, B);
        }
      } else if (lead) {
        for (int idx = tid; idx < kR * Ad; idx += kThreads) {
          const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
          if (gr < B) A.pi[(size_t)gr * Ad + col] = tanhf(outS[row * kOutLd + col]);
        }
        store_rows(xa, kX0Ld, A.aX[0], A.aldx0, S, row0, B);
        // merged phase 2: its tiles rewrite the actor's output layer while other tiles still want the OLD one
        // (dz2 = du W3): one copy of the layer (A x 256 floats of the row-major master) for them, taken here
        if ((A.merged & 2) != 0 && slice == 0)
          for (int idx = tid * 4; idx < Ad * kW4; idx += kThreads * 4)
            *reinterpret_cast<f32x4*>(A.w3_snap + idx) = ld4(A.w3_src + idx);
      }
      stamp();
    }
    if (!TWIN || !A.twin_split) return -1;
    // ---- ... then target critic 2 on (s', a') with role A's a'
    if (!A.do_actor) { rows(); __syncthreads(); }
    if (tid < kR * Ad) {
      const int row = tid / Ad, col = tid - row * Ad;
      xb[row * kX0Ld + S + col] = granule_get(x_slot(role_c) + tid, x_tag, A.err, (KERN_PHASE1 << 8) | SITE_TWIN_SPLIT);
    }
    tp_fwd<WIDTH, LEAN, P>(A.critic2_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, stamp);
    if (lead && tid < kR) granule_put(x_slot(0) + tid, x_tag, outS[tid * kOutLd]);
    stamp();
    return -1;
  }

  if (role == 0) {
    // ---- role A: a' = tanh(actor_target(s')) (TD3: + clipped noise), q' = critic_target(s', a')
    // (TD3: min over the twin targets), TD target                    (ddpg.py:94-95, td3.py:83-101)
    // (SAC: a' ~ pi(s') from the online actor, log pi(a'|s') kept per row     sac.py:90-97)
    bool ac_done = false;
    if constexpr (SAC && RT2) {
      if (A.rt2 == 2) {
        // SAC, over-subscribed launches (r06-13): a' ~ pi(s') and the actor step's pi(s) come from the SAME net — one pass on
        // two row tiles (tp4.h tp4_forward2: the fragments requested once), and role C — a dispatch round of its own behind
        // this role — is not launched.  Tile 1 (s): what role C does — pi, log pi, the raw head, X[0 .. 2] for the actor's dW;
        // it exchanges through role C's cluster area of the slice.
        static_assert(kR * lds_ld(WIDTH) >= kWaves * 256, "the second tile's partial tiles fit a hidden buffer");
        Tp4Two T;
        T.x0[0] = xb; T.x0[1] = xa;
        T.h1[0] = h1; T.h1[1] = h1 + 2 * HB;
        T.h2[0] = h2; T.h2[1] = h1 + 3 * HB;
        T.out[0] = outS; T.out[1] = auxS;
        T.scr[0] = scr; T.scr[1] = h1 + 4 * HB;
        T.st[0] = nostore;
        T.st[1] = Tp3Store{A.aX[1], A.aX[2], nullptr, nullptr, 0};
        const size_t area = (size_t)role_c * gridDim.x * kTpStages * A.xnc * kTpBlk;
        tp4_forward2<P>(A.actor, T, tp, area, row0, B, stamp, rows);
        if (lead) {
          gauss_head(auxS, row0, B, Ad, A.noise_pi, A.rng_seed_pi, A.rng_ctr, nullptr, 0, nullptr, A.pi, A.raw, A.logp);
          store_rows(xa, kX0Ld, A.aX[0], A.aldx0, S, row0, B);
        }
        ac_done = true;
      }
    }
    if (!ac_done) tp_fwd<WIDTH, LEAN, P, NMA>(SAC ? A.actor : A.actor_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, stamp, rows);
    const bool send_a2 = TWIN && A.twin_split && lead;
    unsigned long long* x_a2 = send_a2 ? x_slot(role_c) : nullptr;
    if constexpr (SAC)
      gauss_head(outS, row0, B, Ad, A.noise, A.rng_seed, A.rng_ctr, xb + S, kX0Ld, yS, nullptr, nullptr, nullptr,
                 x_a2, x_tag);
    for (int idx = tid; !SAC && idx < kR * Ad; idx += kThreads) {
      const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
      float v = 0.f;
      if (gr < B) {
        v = tanhf(outS[row * kOutLd + col]);
        if (A.smooth) {
          const float e = A.noise != nullptr ? A.noise[(size_t)gr * Ad + col]
                                             : philox_normal(A.rng_seed, A.rng_ctr, (unsigned)gr, (unsigned)col);
          const float n = fminf(fmaxf(e * A.policy_noise, -A.noise_clip), A.noise_clip);
          v = fminf(fmaxf(v + n, -A.max_action), A.max_action);
        }
      }
      if (send_a2) granule_put(x_a2 + idx, x_tag, v);
      xb[row * kX0Ld + S + col] = v;
    }
    // (the next GEMM's own barrier publishes xb)
    // lean: the B roles' q granules (long published by then) are requested from inside this pass — after its
    // layer-1 stage — so that the seed stage below does not start with a cold round trip
    unsigned long long gq[2] = {0ull, 0ull};
    int hook_n = 0;
    const unsigned long long* gq_src = A.y_granules + A.gran_stride + min(row0 + (tid & (kR - 1)), B - 1);
    auto hook = [&]() {
      stamp();
      if constexpr (LEAN) {
        if (++hook_n == 2) {
          gq[0] = __hip_atomic_load(gq_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (TWIN && A.n_critics == 2) gq[1] = __hip_atomic_load(gq_src + A.gran_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    };
    tp_fwd<WIDTH, LEAN, P, NMA>(A.critic_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, hook);
    float qn = (tid < kR) ? outS[tid * kOutLd] : 0.f;
    if (TWIN && A.twin_split) {
      if (lead && tid < kR) qn = fminf(qn, granule_get(x_slot(0) + tid, x_tag, A.err, (KERN_PHASE1 << 8) | SITE_TWIN_SPLIT));
    } else if (TWIN && A.n_critics == 2) {
      tp_fwd<WIDTH, LEAN, P, NMA>(A.critic2_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, stamp);
      if (tid < kR) qn = fminf(qn, outS[tid * kOutLd]);
    }
    if constexpr (SAC) {
      if (tid < kR) {
        const float alpha = A.log_alpha != nullptr ? (float)exp(*A.log_alpha) : A.alpha_const;
        qn -= alpha * yS[tid];
      }
    }
    if constexpr (!LEAN) {
      if (lead && tid < kR && row0 + tid < B) {
        const float y = rS[tid] + ((1.f - dS[tid]) * A.gamma) * qn;
        // hand-off to the B roles of this slice: ONE aligned 8-byte {epoch, value} granule per
        // row, written through (agent-scope relaxed atomic = sc1 store); the tag makes the
        // data its own flag, no fence needed (cdna guide, G16 R2).
        const unsigned long long g = ((unsigned long long)A.epoch << 32) | (unsigned long long)__float_as_uint(y);
        __hip_atomic_store(A.y_granules + row0 + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (lead && tid < 64) {
      // lean passes: the B roles ran the critics' whole backward with unit seed and sent their q long ago;
      // the seed vectors 2 (q_j - y) / B (k_dw_adam's per-row scale) and the diagnostics are written HERE,
      // by the role that finishes last — nobody waits for this role any more
      const int gr = row0 + tid;
      const bool row_ok = tid < kR && gr < B;
      const float y = row_ok ? rS[tid] + ((1.f - dS[tid]) * A.gamma) * qn : 0.f;
      const int lim = A.debug_expire == (int)SITE_TD_TARGET ? 0 : (1 << 20);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j < (TWIN ? A.n_critics : 1)) {
          float q = 0.f;
          if (row_ok) {
            unsigned long long g = gq[j];                  // requested during the pass above
            bool ok = lim > 0 && (unsigned)(g >> 32) == A.epoch;
            for (int spin = 0; spin < lim && !ok; ++spin) {
              g = __hip_atomic_load(A.y_granules + (size_t)(1 + j) * A.gran_stride + gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = (unsigned)(g >> 32) == A.epoch;
              if (!ok) __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) report_expired(A.err, (KERN_PHASE1 << 8) | SITE_TD_TARGET);
            q = ok ? __uint_as_float((unsigned)g) : __builtin_nanf("");
            float* const* dYj = j == 0 ? A.cdY : A.c2dY;
            const float seed = 2.f * (q - y) * A.inv_B;
            dYj[2][(size_t)gr * A.clddo] = seed;
            if ((A.merged & 1) != 0)   // ... and to the dW tiles of this very launch as a granule (the TD-target array is free in the lean form; the twin's: an array of its own)
              __hip_atomic_store((j == 0 ? A.y_granules : A.seed2_granules) + gr, ((unsigned long long)A.epoch << 32) | (unsigned long long)__float_as_uint(seed),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (j == 0 && A.y_out != nullptr) A.y_out[gr] = y;
            if (j == 0 && A.q_out != nullptr) A.q_out[gr] = q;
          }
          if (A.partials_c != nullptr) {
            float v[3] = {row_ok ? (q - y) * (q - y) : 0.f, row_ok ? q : 0.f, row_ok ? y : 0.f};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const float sum = row16_sum(v[k]);             // the slice's 16 rows sit in lanes 0..15
              if (tid == 0) A.partials_c[((size_t)j * gridDim.x + slice) * 4 + k] = sum;
            }
          }
        }
      }
    }
    stamp();
    return -1;
  }

  // ---- role B: q = critic_j(s, a) forward (runs while role A computes the target).  The twin
  // critic gets its own copy of the code (a runtime-selected Net would leave the kernel-argument
  // registers: profiles/r01b_experiments.txt #10).
  if constexpr (TWIN)
    if (role == 2) { role_b<WIDTH, LEAN, P>(A, A.critic2, A.c2X, A.c2dY, A.partials_c + (size_t)gridDim.x * 4, false, 1, smem, tp, stamp, slice, rows); return -1; }
  role_b<WIDTH, LEAN, P>(A, A.critic, A.cX, A.cdY, A.partials_c, true, 0, smem, tp, stamp, slice, rows);
  return -1;
}

// the tile a workgroup of a merged launch ended up with (the one inlined copy of the tile code per kernel); gate 1: a
// critic tile of phase 1's launch, 2: an actor tile of phase 2's
template <class P, class KA, bool T64 = false>
__device__ __forceinline__ void ddpg_tile(const KA* D, int tile, int gate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (tile < 0) return;
  if constexpr (kDwTileX2 && kMergedTile64 && T64) {
    // the 16 x 64 tile of dw_tile_x2.h, all 16 waves: the split product (PrecX2) or the exact-fp32 one (the other learners:
    // round 5 — TD3's twin critics: 2 x 84 tile workgroups behind the roles instead of 2 x 152; a single critic's 152 small
    // tiles stay the shorter chain: DDPG bf16 29.9 us with them, 30.9 with 84 exact-fp32 16 x 64 tiles)
    dw_tile_x2<KA, P>(*D, smem, tile, gate);
  } else if constexpr (P::kX2 && kDwTileX2) {
    dw_tile_x2<KA>(*D, smem, tile, gate);     // the 16 x 64 split-product tile (dw_tile_x2.h), all 16 waves
  } else if constexpr (std::is_same<KA, DwKArgs>::value) {
    if (threadIdx.x >= kDwThreads) return;    // a tile workgroup is the first 8 waves (the stand-alone kernel's shape and arithmetic)
    // (said again where the compiler sees it: without it the body's "one past the tiles" path — the temperature's Adam
    // step, with its 32 bytes of static LDS — stays in a kernel that asks for the whole LDS dynamically)
    if (tile >= dw_total(*D)) return;
    dw_adam_body<false, 1>(*D, smem, tile);
  }
}

template <int WIDTH, bool LEAN, bool SAC, class P = PrecF32, bool WIDE = false>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase1(const DdpgArgs A) { (void)ddpg_phase1_body<WIDTH, LEAN, SAC, P, WIDE>(A); }

// the plain lean phase launch with the B roles on 32-row slices (ddpg_phase1_body RT2; SINGLE = false: DDPG, TD3 and SAC alike)
template <bool SAC, class P>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase1_rt2(const DdpgArgs A) {
  (void)ddpg_phase1_body<256, true, SAC, P, false, false, DwKArgs, false, true>(A);
}

// phase 1 + the critic's dW tiles in one launch (both argument blocks by value; the tile workgroups index the
// second one through the kernel-argument segment: scalar loads, as k_dw_adam does)
constexpr size_t kMergedDwOffset = (sizeof(DdpgArgs) + alignof(DwKArgs) - 1) / alignof(DwKArgs) * alignof(DwKArgs);
// (WIDE: role A on clusters of eight — with the 84 16 x 64 tiles of a PrecX2 learner, which all find a compute unit when
// roles C and B retire, long before the seeds; the 152 tiles of dw_body.h needed 64 free from the start)
// (TWIN: TD3 — roles A | B1 | B2 | C with both critics' tiles riding; the single-critic instances carry none of the twin code)
template <class P, bool WIDE = false, bool TWIN = false>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase1_dw(const DdpgArgs A, const DwKArgs D) {
  const DwKArgs* Dp = (const DwKArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kMergedDwOffset);
  ddpg_tile<P, DwKArgs, TWIN>(Dp, ddpg_phase1_body<256, true, false, P, WIDE, true, DwKArgs, !TWIN>(A, Dp), 1);
}


// N independent learners in ONE launch (grid.z = learner): the argument blocks live in device memory (N x 1.7 KB
// does not fit the kernel-argument segment), every field read is a scalar load through one uniform pointer.
// (SINGLE: DDPG members — the twin-critic paths compiled out; TD3 / SAC members: the lean passes, SINGLE = false)
// Who is workgroup q of a group launch?  The hardware deals workgroups out to the eight XCDs round-robin by their
// linear index (q % 8), and a member's slices all stream the SAME weights: dealt out slice-fastest, every XCD's L2
// fetched every member's weights (8 x the weight bytes over the fabric — which is what bounded these launches:
// 375 MB per phase-1 launch of 32 members at ~3 TB/s = the 130 us it took).  Here member l lives on XCD l % 8:
// consecutive workgroups are the same position of eight consecutive members.  (n % 8 != 0: the plain order.)
template <int WIDTH, bool LEAN, bool SAC, class P = PrecF32, bool SINGLE = true>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase1_group(const DdpgArgs* __restrict__ batch) {
  const int S = (int)gridDim.x;
  if constexpr (LEAN) {
    // Lean passes: grid (slices, 3 or 4 roles x 4 members, n), learner-major (role-major measured 2-3 % slower there); inside
    // a member role-major, a cluster's four members consecutive (co-dispatched)
    const int R = (int)gridDim.y, n = (int)gridDim.z;
    int l = (int)blockIdx.z, slice = (int)blockIdx.x, by = (int)blockIdx.y;
    if ((n & 7) == 0) {
      const int q = (int)blockIdx.x + S * ((int)blockIdx.y + R * (int)blockIdx.z);
      const int p = q >> 3, inner = p % (S * R), c = inner >> 2;
      l = (q & 7) + 8 * (p / (S * R));
      slice = c % S;
      by = (c / S) * 4 + (inner & 3);
    }
    (void)ddpg_phase1_body<WIDTH, LEAN, SAC, P, false, false, DwKArgs, SINGLE>(batch[l], nullptr, by, slice);
  } else {
    // Generic passes (DDPG members in exact fp32): grid (slices, 1, 3 roles x n learners), ROLE-major — every learner's
    // role A, then every learner's role B ...: a role B workgroup finds its TD target written long ago instead of spinning
    // for it on a compute unit beside role A's (dispatch is in block order): 32 members 71.6k -> 75.2k updates/s
    static_assert(LEAN || SINGLE, "the generic passes: DDPG members only");
    const int n = (int)gridDim.z / 3, role = (int)blockIdx.z / n;
    int l = (int)blockIdx.z - role * n, slice = (int)blockIdx.x;
    const int G = batch[0].group_span;               // XCDs per member: its slices in G parts
    if (G < 8 && ((n * G) & 7) == 0 && S % G == 0) {
      const int r = (int)blockIdx.x + S * (int)blockIdx.z - role * S * n, Sp = S / G;
      const int v = (r & 7) + 8 * ((r >> 3) / Sp);   // (member, part)
      l = v / G;
      slice = (v - l * G) * Sp + (r >> 3) % Sp;
    }
    (void)ddpg_phase1_body<WIDTH, LEAN, SAC, P, false, false, DwKArgs, true>(batch[l], nullptr, role * (int)gridDim.y + (int)blockIdx.y, slice);
  }
}

// WIDE (DDPG / TD3, fp32 lean passes): the critic's forward + backward pass — two thirds of this kernel, on a
// launch that leaves most of the chip idle — runs on clusters of EIGHT CUs (tp4_scalar_fb<P, 8>: no dz1
// partials leave it); members 4..7 are done after it, members 0..3 go on to the actor's backward (clusters of 4,
// no exchange), so the actor's dW launch still sums four partial buffers.
template <int WIDTH, bool LEAN, bool SAC, class P, bool WIDE = false>
__device__ __forceinline__ void ddpg_phase2_body(const DdpgArgs& A, int bx_in = -1, int by_in = -1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LY = FusedLds<WIDTH>;
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  float* xa = smem + LY::xa;
  float* h1 = smem + LY::h;          // critic hidden (2 buffers), then actor hidden (2 buffers)
  float* h2 = h1 + HB;
  float* ha1 = h2 + HB;
  float* ha2 = ha1 + HB;
  float* outS = smem + LY::out;
  float* auxS = smem + LY::aux;
  float* scr = smem + LY::scr;
  float* piS = smem + LY::xb;        // [kR][kX0Ld] tile reused for pi
  const int B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x;
  const int slice = bx_in < 0 ? (int)blockIdx.x : bx_in;
  const int by = by_in < 0 ? (int)blockIdx.y : by_in;
  const int row0 = slice * kR;
  // SAC with p2_pair: two clusters per slice, one per online critic (g = 0 / 1), side by side
  static_assert(!(WIDE && SAC), "SAC's phase 2 runs its twin critics on two clusters of 4");
  constexpr int NMC = (LEAN && WIDE) ? 8 : 4;          // members of the critic pass's clusters (lean passes)
  const int ncl = (LEAN && WIDE) ? 8 : A.nc;           // rows of the grid per cluster
  const int n_clus = SAC ? 1 + A.p2_pair : 1;
  const int g = (SAC && by < n_clus * ncl) ? by / ncl : 0;
  Tp tp{by - g * ncl, ncl,
        A.xbuf + ((size_t)g * gridDim.x + slice) * kTpStages * A.xnc * kTpBlk, A.cluster_tag, 0,
        A.err, KERN_PHASE2 << 8, A.debug_expire == (int)SITE_CLUSTER ? 0 : kTpSpin};
  tp.local = bx_in < 0 && cluster_on_one_xcd(A);
  const bool lead = tp.c == 0;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (kTraceOn && A.trace != nullptr && (tid & 63) == 0 && (tid == 0 || slice == 0) && lead && g == 0 && n_stamp < kTraceStamps) {
      const int slot = tid == 0 ? slice : 16 + (tid >> 6);
      long long* tr = A.trace + ((size_t)slot * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  if (by == n_clus * ncl) {
    // ---- prefetch row: gather the next update's rows (same draw as load_batch will not have
    // to make) and leave them contiguous for phase 1 of the next step
    float* xb = smem + LY::xb;
    float* rS = smem + LY::misc;
    float* dS = rS + kR;
    int* meta = reinterpret_cast<int*>(dS + 2 * kR);
    int* endsS = reinterpret_cast<int*>(smem + LY::misc + 96);
    load_batch(A.next, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
    __syncthreads();
    store_rows(xa, kX0Ld, const_cast<float*>(A.next.s), S, S, row0, B);
    store_rows(xb, kX0Ld, const_cast<float*>(A.next.s2), S, S, row0, B);
    for (int idx = tid; idx < kR * Ad; idx += kThreads) {
      const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
      if (gr < B) const_cast<float*>(A.next.a)[(size_t)gr * Ad + col] = xa[row * kX0Ld + S + col];
    }
    if (tid < kR && row0 + tid < B) {
      const_cast<float*>(A.next.r)[row0 + tid] = rS[tid];
      const_cast<float*>(A.next.d)[row0 + tid] = dS[tid];
    }
    return;
  }
  stamp();
  const Tp3Store nostore{nullptr, nullptr, nullptr, nullptr, 0};
  if constexpr (LEAN) {
    // the prologue's pointers, fetched together HERE (one scalar wait, under the LDS zeroing): left to the
    // compiler each is loaded where it is first used — a scalar load and a wait between every two row requests
    const float* p0 = A.aX[0]; const float* p1 = A.aX[1]; const float* p2 = A.aX[2]; const float* p3 = A.pi;
    const int ld0 = A.aldx0;
    asm volatile("" :: "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(ld0));
  }
  // [s | pi] and the actor's forward activations (for its ReLU masks)
  lds_zero(xa, 2 * kR * kX0Ld);
  if constexpr (LEAN) lds_zero(auxS, kR * kOutLd);   // the gradient tile's padding columns stay zero
  __syncthreads();
  float g_mu = 0.f, g_ls = 0.f, g_e = 0.f;   // SAC: this thread's (row, action dim) head output and draw
  // (lean passes: the loads and LDS stores below run as the first pass's `pre` — behind its fragment requests, no
  // barrier of their own: tp4.h)
  auto prologue = [&]() {
  if constexpr (LEAN) {
    // every thread's loads first (one cold round trip), then its LDS stores: five helper calls in a
    // row are five load -> wait -> store sequences for the threads that take part in all of them
    constexpr int K4 = WIDTH / 4;                        // float4 per activation row
    const int r4 = tid / K4, c4 = (tid - r4 * K4) * 4;    // kR * K4 == kThreads for WIDTH 256
    const bool ok4 = row0 + r4 < B && g == 0 && tp.c < 4;  // (the twin critic's cluster, and members 4..7 of a wide one, need only [s | pi])
    f32x4 v1 = f32x4{0.f, 0.f, 0.f, 0.f}, v2 = v1;
    if (ok4) {
      v1 = ld4(A.aX[1] + (size_t)(row0 + r4) * WIDTH + c4);
      v2 = ld4(A.aX[2] + (size_t)(row0 + r4) * WIDTH + c4);
    }
    const int rs_ = tid / S, cs_ = tid - rs_ * S;          // state rows: kR * S <= 2 * kThreads (S <= 96)
    const bool oks = tid < kR * S && row0 + rs_ < B;
    const float vs = oks ? ldc(A.aX[0] + (size_t)(row0 + rs_) * A.aldx0 + cs_) : 0.f;
    const int tid2 = tid + kThreads;                       // elements kThreads .. kR * S of wide states
    const int rs2_ = tid2 / S, cs2_ = tid2 - rs2_ * S;
    const bool oks2 = tid2 < kR * S && row0 + rs2_ < B;
    const float vs2 = oks2 ? ldc(A.aX[0] + (size_t)(row0 + rs2_) * A.aldx0 + cs2_) : 0.f;
    const int rp_ = tid / Ad, cp_ = tid - rp_ * Ad;
    const bool okp = tid < kR * Ad && row0 + rp_ < B;
    const float vp = okp ? ldc(A.pi + (size_t)(row0 + rp_) * Ad + cp_) : 0.f;
    if constexpr (SAC) {
      if (okp && g == 0) {
        g_mu = A.raw[(size_t)(row0 + rp_) * 2 * Ad + cp_];
        g_ls = A.raw[(size_t)(row0 + rp_) * 2 * Ad + Ad + cp_];
        if (A.noise_pi != nullptr) g_e = A.noise_pi[(size_t)(row0 + rp_) * Ad + cp_];
      }
    }
    *reinterpret_cast<f32x4*>(ha1 + r4 * WL + c4) = v1;
    *reinterpret_cast<f32x4*>(ha2 + r4 * WL + c4) = v2;
    if (tid < kR * S) xa[rs_ * kX0Ld + cs_] = vs;
    if (tid2 < kR * S) xa[rs2_ * kX0Ld + cs2_] = vs2;
    if (tid < kR * Ad) { xa[rp_ * kX0Ld + S + cp_] = vp; piS[rp_ * kX0Ld + cp_] = vp; }
    if constexpr (SAC) {   // the same draw as role C's forward
      if (okp && g == 0 && A.noise_pi == nullptr) g_e = philox_normal(A.rng_seed_pi, A.rng_ctr, (unsigned)(row0 + rp_), (unsigned)cp_);
    }
  } else {
    load_rows(xa, kX0Ld, 0, A.aX[0], A.aldx0, S, row0, B);
    load_rows(xa, kX0Ld, S, A.pi, Ad, Ad, row0, B);
    load_rows(piS, kX0Ld, 0, A.pi, Ad, Ad, row0, B);
    load_rows4(ha1, WL, A.aX[1], WIDTH, WIDTH, row0, B);
    load_rows4(ha2, WL, A.aX[2], WIDTH, WIDTH, row0, B);
  }
  };
  if constexpr (!LEAN) prologue();
  stamp();
  // ---- q = critic(s, pi) with the updated critic, and its backward down to the action
  // columns: da -> auxS[:, 0:A].  The seed -1/B is a constant, so the lean path runs both
  // as one pass (tp4_scalar_fb) in which q — only logged — is off the critical path.
  if constexpr (SAC) {
    // SAC: both online critics at (s, pi), each forward + unit-seed (-1/B) backward to the action
    // columns; per row the smaller q routes its gradient (torch.min: ties split), then the
    // tanh-Gaussian head backward with the entropy term             (sac.py:118-127)
    float* d2S = smem + LY::aux2;
    float* qxS = smem + LY::misc;      // the q that is not in outS: q1 (back to back) or q2 (side by side)
    const float* q1p = qxS; int q1s = 1;
    const float* q2p = outS; int q2s = kOutLd;
    if (A.p2_pair) {
      // one granule per (row, action dim | q) from the lead member of cluster 1 to every member of
      // cluster 0, in the exchange area of the unused role slot 2, tagged with the launch's tag
      unsigned long long* xq = A.xbuf + ((size_t)2 * gridDim.x + slice) * kTpStages * A.xnc * kTpBlk + tid;
      const unsigned tag = (A.cluster_tag << 6) | 63u;
      const int xr = tid / (Ad + 1), xc = tid - xr * (Ad + 1);
      const bool xmine = tid < kR * (Ad + 1);
      if (g == 1) {
        tp4_scalar_fb<P>(A.critic2, xa, h1, h2, ha2 + HB, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, auxS, stamp, nullptr, BiasOv(), QPart(), prologue);
        if (lead && xmine) {
          const float v = xc < Ad ? auxS[xr * kOutLd + xc] : outS[xr * kOutLd];
          __hip_atomic_store(xq, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
      }
      tp4_scalar_fb<P>(A.critic, xa, h1, h2, ha2 + HB, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, auxS, stamp, nullptr, BiasOv(), QPart(), prologue);
      if (xmine) {
        unsigned long long x = 0;
        bool ok = false;
        for (int spin = 0; spin < kTpSpin; ++spin) {
          x = __hip_atomic_load(xq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (unsigned)(x >> 32) == tag;
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) report_expired(A.err, (KERN_PHASE2 << 8) | SITE_P2_PAIR);
        const float v = ok ? __uint_as_float((unsigned)x) : __builtin_nanf("");
        if (xc < Ad) d2S[xr * kOutLd + xc] = v; else qxS[xr] = v;
      }
      __syncthreads();
      q1p = outS; q1s = kOutLd;
      q2p = qxS; q2s = 1;
    } else {
      tp4_scalar_fb<P>(A.critic, xa, h1, h2, ha2 + HB, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, auxS, stamp, nullptr, BiasOv(), QPart(), prologue);
      if (tid < kR) qxS[tid] = outS[tid * kOutLd];
      tp4_scalar_fb<P>(A.critic2, xa, h1, h2, ha2 + HB, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, d2S, stamp);
    }
    if (A.partials_a != nullptr && lead && tid < 64) {
      float v = (tid < kR && row0 + tid < B) ? fminf(q1p[tid * q1s], q2p[tid * q2s]) : 0.f;
      v = row16_sum(v);
      if (tid == 0) {
        A.partials_a[slice * 4 + 0] = 0.f;
        A.partials_a[slice * 4 + 1] = v;
        A.partials_a[slice * 4 + 2] = 0.f;
      }
    }
    stamp();   // da ready
    const int r_ = tid / Ad, c_ = tid - r_ * Ad;
    const bool mine = tid < kR * Ad;
    float dmu = 0.f, dls = 0.f;
    if (mine && row0 + r_ < B) {
      const float q1 = q1p[r_ * q1s], q2 = q2p[r_ * q2s];
      const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
      const float da = w1 * auxS[r_ * kOutLd + c_] + (1.f - w1) * d2S[r_ * kOutLd + c_];
      const float alpha = A.log_alpha != nullptr ? (float)exp(*A.log_alpha) : A.alpha_const;
      gauss_elem_bwd(g_mu, g_ls, g_e, da, alpha * A.inv_B, &dmu, &dls);
    }
    if (mine) {   // in place over critic 1's action gradient; the tile's other columns are zero
      auxS[r_ * kOutLd + c_] = dmu;
      auxS[r_ * kOutLd + Ad + c_] = dls;
      if (lead && row0 + r_ < B) {
        A.adY[2][(size_t)(row0 + r_) * A.alddo + c_] = dmu;
        A.adY[2][(size_t)(row0 + r_) * A.alddo + Ad + c_] = dls;
      }
    }
    const Tp3Store sta{nullptr, nullptr, A.adY[1], A.adY[0], A.adY0_stride, LEAN ? B : 0};
    tp_bwd<WIDTH, LEAN, P>(A.actor, auxS, ha1, ha2, scr, tp, sta, row0, B, 0, 0, auxS, stamp);
    stamp();
    return;
  }
  if constexpr (LEAN) {
    float* qsum = nullptr;   // mean-q diagnostics: written by the q wave of the lead member
    if (A.partials_a != nullptr && lead) {
      qsum = A.partials_a + slice * 4 + 1;
      if (tid == 0) { A.partials_a[slice * 4 + 0] = 0.f; A.partials_a[slice * 4 + 2] = 0.f; }
    }
    tp4_scalar_fb<P, NMC>(A.critic, xa, h1, h2, ha2 + HB, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, auxS, stamp, qsum, BiasOv(), QPart(), prologue);
    if constexpr (WIDE) {
      if (tp.c >= 4) return;        // the actor's backward is a cluster of four
      tp.nc = 4;
    }
  } else {
    tp_fwd<WIDTH, LEAN, P>(A.critic, xa, h1, h2, outS, scr, tp, nostore, row0, B, stamp);
    stamp();
    lds_zero(auxS, kR * kOutLd);
    __syncthreads();
    if (tid < kR && row0 + tid < B) auxS[tid * kOutLd] = -A.inv_B;
  }
  if (!LEAN && A.partials_a != nullptr && lead) {
    float v = (tid < kR && row0 + tid < B) ? outS[tid * kOutLd] : 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
    if ((tid & 63) == 0) scr[(tid >> 6) * 4 + 1] = v;
    __syncthreads();
    if (tid == 0) {
      float sum = 0.f;
      for (int w = 0; w < kWaves; ++w) sum += scr[w * 4 + 1];
      A.partials_a[slice * 4 + 0] = 0.f;
      A.partials_a[slice * 4 + 1] = sum;
      A.partials_a[slice * 4 + 2] = 0.f;
    }
  }
  if constexpr (!LEAN)
    tp_bwd<WIDTH, LEAN, P>(A.critic, auxS, h1, h2, scr, tp, nostore, row0, B, S, Ad, auxS, stamp);
  stamp();   // da ready
  // ---- du = da (1 - pi^2), zero padded
  float du = 0.f;
  const int r_ = tid / Ad, c_ = tid - r_ * Ad;
  const bool mine = tid < kR * Ad;
  if (mine && row0 + r_ < B) {
    const float p = piS[r_ * kX0Ld + c_];
    du = auxS[r_ * kOutLd + c_] * (1.f - p * p);
  }
  if constexpr (LEAN) {
    // in place: the padding columns were zeroed in the prologue and the lean passes write only
    // the valid ones; the backward's first barrier publishes du, the dW input goes out from
    // registers
    if (mine) {
      auxS[r_ * kOutLd + c_] = du;
      if (lead && row0 + r_ < B) A.adY[2][(size_t)(row0 + r_) * A.alddo + c_] = du;
    }
  } else {
    __syncthreads();
    lds_zero(auxS, kR * kOutLd);
    __syncthreads();
    if (mine) auxS[r_ * kOutLd + c_] = du;
    __syncthreads();
    if (lead) store_rows(auxS, kOutLd, A.adY[2], A.alddo, Ad, row0, B);
  }
  // ---- actor backward over its stored activations
  const Tp3Store sta{nullptr, nullptr, A.adY[1], A.adY[0], A.adY0_stride, LEAN ? B : 0};
  tp_bwd<WIDTH, LEAN, P>(A.actor, auxS, ha1, ha2, scr, tp, sta, row0, B, 0, 0, auxS, stamp);
  stamp();
}

template <int WIDTH, bool LEAN, bool SAC, class P = PrecF32, bool WIDE = false>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase2(const DdpgArgs A) { ddpg_phase2_body<WIDTH, LEAN, SAC, P, WIDE>(A); }

template <int WIDTH, bool LEAN, bool SAC, class P = PrecF32>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase2_group(const DdpgArgs* __restrict__ batch) {
  // grid (slices, cluster size [+ 1: the prefetch row], n); member l on XCD l % 8 (k_ddpg_phase1_group)
  const int S = (int)gridDim.x, R = (int)gridDim.y, n = (int)gridDim.z;
  int l = (int)blockIdx.z, slice = (int)blockIdx.x, by = (int)blockIdx.y;
  if ((n & 7) == 0) {
    const int q = (int)blockIdx.x + S * ((int)blockIdx.y + R * (int)blockIdx.z);
    const int p = q >> 3, inner = p % (S * R);
    l = (q & 7) + 8 * (p / (S * R));
    if constexpr (LEAN) {           // a cluster's four members consecutive, the prefetch row's workgroups behind them
      if (inner < 4 * S) { slice = inner >> 2; by = inner & 3; }
      else { slice = inner - 4 * S; by = 4; }
    } else {                        // (cluster size 1: row 0 the pass, row 1 the prefetch row)
      slice = inner % S;
      by = inner / S;
      const int G = batch[0].group_span;
      if (G > 1 && G < 8 && S % G == 0) {      // a member's slices in G parts on G XCDs
        const int Sp = S / G, pr = p % (Sp * R);
        const int v = (q & 7) + 8 * (p / (Sp * R));
        l = v / G;
        slice = (v - l * G) * Sp + pr % Sp;
        by = pr / Sp;
      } else if (G >= 8) {
        l = (int)blockIdx.z; slice = (int)blockIdx.x; by = (int)blockIdx.y;
      }
    }
  }
  ddpg_phase2_body<WIDTH, LEAN, SAC, P>(batch[l], slice, by);
}


// ---------------------------------------------------------------------------------------------------------------
// Phase 2 with the ACTOR's dW + Adam tiles on the same launch (DdpgArgs::merged bit 1; PrecX2 learners, DDPG / TD3,
// lean passes on clusters of eight, action_dim <= kDuLd, B <= 256).
//
// The actor's backward is linear in its output seed du = da (1 - pi^2) [B x A], and everything else it needs — the
// actor's weights and its forward activations (role C of phase 1) — exists when this launch starts.  The critic pass
// (forward + constant-seed backward to the action columns, unchanged) ends by publishing du as {epoch, value}
// granules; the actor's dW tiles (dw_tile_x2.h, GATE 2), riding behind the pass like the critic's tiles ride on phase
// 1, have taken in their Adam state and X rows by then and form their dY from du: the output layer's is du itself, the
// second hidden layer's (h2 > 0) (du W3) — 84 of the 100 tiles' worth of dW.  The first hidden layer's dY is one more
// backward step,   g1[b, k] = (h1[b, k] > 0) sum_n g2[b, n] W2[n, k],   g2 = (h2 > 0) (du W3):
// the pass's eight members per slice go on with it (member c: columns 32 c .. 32 c + 31; its W2^T shard, the h2 / h1
// masks and W3 were taken in at the START of the pass — nothing it reads can be rewritten by a tile, which stores only
// after du), 3 MFMAs per wave, and publish g1 element by element as granules for the 16 first-layer tiles.  What used
// to follow the critic pass — the actor's backward (a whole pass), a kernel boundary and a k_dw_adam launch — is one
// granule hop, a few FMAs per element, the tile's MFMAs and the Adam epilogue.
// (An earlier form ran the du-independent part of that step with unit seeds on eight more workgroups per slice: 128
// compute units for 6 us per update that the whole-update launch needs for the pass itself — profiles/r03_experiments.txt.)
// Grid rows: [0, 8) critic pass | prefetch row (step_n, two-launch form) | tiles.
// ---------------------------------------------------------------------------------------------------------------
// one bounded wait for n flag granules {tag, *}: thread k polls flag k (n <= threads); the caller's barrier releases everybody
__device__ __forceinline__ void wait_flags(const unsigned long long* flags, int n, unsigned tag, unsigned* err, unsigned code) {
  const int k = (int)threadIdx.x;
  if (k < n) {
    bool ok = false;
    for (int spin = 0; spin < kTpSpin && !ok; ++spin) {
      ok = (unsigned)(__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(2);
    }
    if (!ok) report_expired(err, code);
  }
}

// (returns the actor tile this workgroup is, -1 for the pass's and the prefetch row's workgroups: ddpg_tile runs it)
// PF: the launch may carry the prefetch row (the two-launch form; k_ddpg_chain gathers the next rows itself)
// (what changes from update to update inside k_ddpg_chain; the one-update launches pass their argument block's values)
struct PassCtx {
  unsigned epoch;               // tag of this update's flags and granules
  unsigned tag;                 // the pass's cluster-exchange tag
  const float* w3;              // the actor's output layer [A][256] as it is before this update's actor step
  const float* cb[3];           // the critic's biases as this update's tiles leave them (uncached copies)
  long long* trace;
  const float* pb1_32 = nullptr;   // (bf16 learners) the actor's second hidden layer as its fp32 W^T pack: the unit-seed rows' B operand
};
// GE (k_ddpg_chain): the first hidden layer's dY leaves this pass as UNIT-SEED rows G_j = dz1 / d(du_j), j < A — the
// actor's backward is linear in du — computed and stored BEFORE the critic's new weights arrive (the pass's members sit
// idle, rows in, until the critic's tiles flag), so that every actor tile waits for du only and nothing follows du
// here.  Otherwise: the backward step after du, its result as granules (the two-launch form, whose pass starts at once).
template <class P, class KA = DwKArgs, bool PF = true, bool GE = false>
__device__ __forceinline__ int ddpg_phase2m_body(const DdpgArgs& A, const KA* D, int by_in, const PassCtx& cx) {
  static_assert((P::kX2 || GE) && kDwTileX2, "the merged phase 2: PrecX2 learners; with the unit-seed rows (k_ddpg_chain) every arithmetic");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LY = FusedLds<256>;
  constexpr int HB = kR * kWL4;
  constexpr int NMC = 8;
  const int B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x;
  const int slice = blockIdx.x, row0 = slice * kR;
  const int y = by_in < 0 ? (int)blockIdx.y : by_in;
  const int yP = NMC, yT = yP + ((PF && A.prefetch_next) ? 1 : 0);
  if (y >= yT) {   // a tile workgroup: all 16 waves (16 minibatch rows each)
    const int tile = (y - yT) * (int)gridDim.x + slice;
    if (tile >= dw_total(*D)) return -1;
    if (A.whole) {      // (never taken: k_ddpg_chain runs its tiles as continuations of the roles, not through here)
      wait_flags(A.w_flags, 4 * (int)gridDim.x, cx.epoch, A.err, (KERN_PHASE2 << 8) | SITE_DW_GATE);
      __syncthreads();
    }
    return tile;
  }
  float* xa = smem + LY::xa;
  if (PF && y == yP) {
    // ---- prefetch row: the next update's rows (as in ddpg_phase2_body)
    float* xb = smem + LY::xb;
    float* rS = smem + LY::misc;
    float* dS = rS + kR;
    int* meta = reinterpret_cast<int*>(dS + 2 * kR);
    int* endsS = reinterpret_cast<int*>(smem + LY::misc + 96);
    load_batch(A.next, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);
    __syncthreads();
    store_rows(xa, kX0Ld, const_cast<float*>(A.next.s), S, S, row0, B);
    store_rows(xb, kX0Ld, const_cast<float*>(A.next.s2), S, S, row0, B);
    for (int idx = tid; idx < kR * Ad; idx += kThreads) {
      const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
      if (gr < B) const_cast<float*>(A.next.a)[(size_t)gr * Ad + col] = xa[row * kX0Ld + S + col];
    }
    if (tid < kR && row0 + tid < B) {
      const_cast<float*>(A.next.r)[row0 + tid] = rS[tid];
      const_cast<float*>(A.next.d)[row0 + tid] = dS[tid];
    }
    return -1;
  }
  // ---- the critic pass: q = critic(s, pi) forward + constant-seed backward to the action columns -> du -> granules
  float* h1 = smem + LY::h;
  float* h2 = h1 + HB;
  float* g2 = h2 + HB;
  float* Wl = g2 + HB;                             // two spare hidden buffers: the member's W2^T shard [2 tiles][8 steps][hi 256 | lo 256]
  float* outS = smem + LY::out;
  float* auxS = smem + LY::aux;
  float* duS = smem + LY::aux2;                    // [kR] the rows' un-scaling factors of the backward step
  float* scr = smem + LY::scr;
  float* w3s = smem + LY::misc + 96;               // [kDuLd][256] the output layer's rows (the prefetch rows' ends table is not in use here)
  static_assert(2 * HB >= 2 * 8 * 512, "the shard fits the two spare hidden buffers");
  static_assert(kMaxEnds >= kDuLd * 256, "W3 fits the ends table's area");
  Tp tp{y, NMC, A.xbuf + (size_t)slice * kTpStages * A.xnc * kTpBlk, cx.tag, 0,
        A.err, KERN_PHASE2 << 8, A.debug_expire == (int)SITE_CLUSTER ? 0 : kTpSpin};
  tp.local = GE && cluster_on_one_xcd(A);
  const bool lead = tp.c == 0;
  const int c = tp.c;
  const int lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  long long* const trace = cx.trace;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (kTraceOn && trace != nullptr && (tid & 63) == 0 && (tid == 0 || slice == 0) && lead && n_stamp < kTraceStamps) {
      const int slot = tid == 0 ? slice : 16 + (tid >> 6);
      long long* tr = trace + ((size_t)slot * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();
  const Tp3Store nostore{nullptr, nullptr, nullptr, nullptr, 0};
  BiasOv cbo;
  if (A.whole) {
    // (k_ddpg_chain) role C of this launch wrote s, pi and the actor's activations — uncached memory: its members'
    // flags, then COHERENT loads of what they wrote (engine.h ld4c / ldc: a plain load may hit a stale line of this CU's
    // L1, and no cheap invalidate drops it); the critic's packs and biases follow below, behind the rows
    wait_flags(A.w_flags, 4 * (int)gridDim.x, cx.epoch, A.err, (KERN_PHASE2 << 8) | SITE_DW_GATE);
    __syncthreads();
  }
  stamp();   // role C's flags seen
  {
    const float* p0 = A.aX[0]; const float* p3 = A.pi; const int ld0 = A.aldx0;
    asm volatile("" :: "s"(p0), "s"(p3), "s"(ld0));
  }
  // ---- requests.  For the backward step at the end: the member's W2^T shard (32 KB, two b128 per thread), the
  // output layer's rows, this thread's four h2 elements and (threads < 512) its h1 element — the ReLU masks
  // (the member's W2^T shard: two 16-row k tiles x the whole contraction — 32 KB in the fp32 and x2 packs; a bf16 learner's
  // unit-seed rows are formed in exact fp32 from the fp32 pack its tiles keep current)
  const float* wsrc = ((P::kBf16 && cx.pb1_32 != nullptr) ? cx.pb1_32 : A.actor.pb[1]) + (size_t)c * 2 * 16 * 256;
  f32x4 wv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) wv[q] = ld4c(wsrc + ((size_t)q * kThreads + tid) * 4);      // (coherent loads throughout this prologue: engine.h ld4c)
  const int hr = tid >> 6, hc = (tid & 63) * 4;                  // g2 element quad: row hr, columns hc .. hc + 3
  f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (row0 + hr < B) hv = ld4c(A.aX[2] + (size_t)(row0 + hr) * kW4 + hc);
  f32x4 w3q = f32x4{0.f, 0.f, 0.f, 0.f};                         // W3[tid >> 6][4 (tid & 63) ..]: rows beyond A stay zero
  if ((tid >> 6) < Ad) w3q = ld4c(cx.w3 + (size_t)tid * 4);
  const int et = (tid >> 8) & 1, er = (tid >> 4) & 15, ec = tid & 15;   // g1 element (threads < 512): tile et, row er, column ec
  float m1 = 0.f;
  if (!GE && tid < 512 && row0 + er < B) m1 = ldc(A.aX[1] + (size_t)(row0 + er) * kW4 + 32 * c + 16 * et + ec);
  // (GE: the h1 elements — ReLU masks — of this lane's four unit-seed outputs, requested with everything else: a load
  // issued later would be waited for at the next barrier, a cold round trip inside the unit-seed stage)
  f32x4 gm1 = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (GE) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (row0 + 4 * kk + r < B) gm1[r] = ldc(A.aX[1] + (size_t)(row0 + 4 * kk + r) * kW4 + 32 * c + 16 * (wave & 1) + i);
  }
  // [s | pi]: loads first, then the zero fill and the stores
  const int rs_ = tid / S, cs_ = tid - rs_ * S;
  const bool oks = tid < kR * S && row0 + rs_ < B;
  const float vs = oks ? ldc(A.aX[0] + (size_t)(row0 + rs_) * A.aldx0 + cs_) : 0.f;
  const int tid2 = tid + kThreads;
  const int rs2_ = tid2 / S, cs2_ = tid2 - rs2_ * S;
  const bool oks2 = tid2 < kR * S && row0 + rs2_ < B;
  const float vs2 = oks2 ? ldc(A.aX[0] + (size_t)(row0 + rs2_) * A.aldx0 + cs2_) : 0.f;
  const int rp_ = tid / Ad, cp_ = tid - rp_ * Ad;
  const bool okp = tid < kR * Ad && row0 + rp_ < B;
  const float vp = okp ? ldc(A.pi + (size_t)(row0 + rp_) * Ad + cp_) : 0.f;
  lds_zero(xa, kR * kX0Ld);
  lds_zero(auxS, kR * kOutLd);
  __syncthreads();
  if (tid < kR * S) xa[rs_ * kX0Ld + cs_] = vs;
  if (tid2 < kR * S) xa[rs2_ * kX0Ld + cs2_] = vs2;
  if (tid < kR * Ad) xa[rp_ * kX0Ld + S + cp_] = vp;
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<f32x4*>(Wl + ((size_t)q * kThreads + tid) * 4) = wv[q];
  if (tid < kDuLd * 64) *reinterpret_cast<f32x4*>(w3s + (size_t)tid * 4) = w3q;
  // (the masks stay in one register through the pass: bits 0..3 the h2 quad, bit 4 the h1 element)
  const unsigned mbits = (hv[0] > 0.f ? 1u : 0u) | (hv[1] > 0.f ? 2u : 0u) | (hv[2] > 0.f ? 4u : 0u) | (hv[3] > 0.f ? 8u : 0u) |
                         (m1 > 0.f ? 16u : 0u);
  stamp();   // rows and shard staged in LDS
  if constexpr (GE && !P::kX2) {
    // ---- the same unit-seed rows in exact fp32: the A operand (h2 > 0) W3[j, :] formed per macro step from a 0 / 1
    // mask tile and the row (4 multiplies), 64 v_mfma_f32_16x16x4_f32 per (tile, seed) wave — 2.6 us of the matrix pipe,
    // inside the pass's wait for the critic's tiles
    float* mS = h1;
    *reinterpret_cast<f32x4*>(mS + hr * kWL4 + hc) = f32x4{(mbits & 1u) ? 1.f : 0.f, (mbits & 2u) ? 1.f : 0.f, (mbits & 4u) ? 1.f : 0.f, (mbits & 8u) ? 1.f : 0.f};
    __syncthreads();                                         // mask tile, W2^T shard, W3 rows in LDS
    const int t1 = wave & 1, j = wave >> 1;
    if (j < Ad) {
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll 4
      for (int sx = 0; sx < 16; ++sx) {
        const f32x4 b4 = ld4(Wl + ((size_t)t1 * 16 + sx) * 256 + lane * 4);
        const f32x4 a4 = ld4(mS + i * kWL4 + 16 * sx + 4 * kk) * ld4(w3s + j * 256 + 16 * sx + 4 * kk);
        if (sx & 1) { mac4(a4, b4, acc1); } else { mac4(a4, b4, acc0); }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + 4 * kk + r < B)
          st_ag(A.gu + ((size_t)j * B + row0 + 4 * kk + r) * kW4 + 32 * c + 16 * t1 + i, gm1[r] > 0.f ? acc0[r] + acc1[r] : 0.f);
    }
    stamp();   // unit-seed rows requested out
  }
  if constexpr (GE && P::kX2) {
    // ---- unit-seed backward through the second hidden layer, this member's 32 columns of the first one's dY:
    //     G_j[b, k] = (h1[b, k] > 0) sum_n (h2[b, n] > 0) W3[j, n] W2[n, k]          (dz1 = sum_j du_j G_j)
    // The A operand (h2 > 0) W3[j, :] is a MASKED copy of one row for all 16 minibatch rows: the row is split into its
    // two fp16 planes ONCE (wave j = row j: scaled by the power of two of its largest magnitude), the masks become
    // 0 / 0xffff halfs, and a macro step's operand is two ANDs — no per-step scaling, no per-step split (a first form
    // that split mask * W3 per step was bound by vector instructions: 4.3 us).  wave = (tile t1 of the member's two,
    // seed j) runs the whole 256-deep contraction itself: nothing is exchanged between waves, one barrier.  The rows
    // go to A.gu [A][B][256]; their flag is raised after the wait for the critic's tiles below.
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    _Float16* mH = reinterpret_cast<_Float16*>(h1);                   // [16 rows][8 steps][4 kk][8 halfs] masks
    _Float16* wH = reinterpret_cast<_Float16*>(h1 + 2048);            // [kDuLd][8][4][8] hi plane of s_j W3[j]
    _Float16* wL = reinterpret_cast<_Float16*>(h1 + 2048 + 1024);     // ... lo plane
    float* sJ = h1 + 2048 + 2048;                                      // [kDuLd] s_j
    static_assert(HB >= 2048 + 2048 + 16, "masks and the output layer's planes fit one hidden buffer");
    {
      const int l = tid & 63, ws_ = l >> 3, half_ = (l >> 2) & 1, kq_ = l & 3;
      const int off = ((ws_ * 4 + kq_) * 8) + 4 * half_;             // halfs, inside a (row | seed) block of 256
      union { f16x4 v; unsigned short b[4]; } mk;
#pragma unroll
      for (int t = 0; t < 4; ++t) mk.b[t] = ((mbits >> t) & 1u) ? 0xffffu : 0u;
      *reinterpret_cast<f16x4*>(mH + hr * 256 + off) = mk.v;          // thread = (row hr, columns hc .. hc + 3), hc = 4 l
      if ((tid >> 6) < kDuLd) {                                       // wave j: row j of the output layer (w3q; zero beyond A)
        float m = fmaxf(fmaxf(fabsf(w3q[0]), fabsf(w3q[1])), fmaxf(fabsf(w3q[2]), fabsf(w3q[3])));
        m = wave_max(m);
        const float sj = P::a_scale(m);
        f16x8 hi8, lo8;
        x2_split8(w3q * sj, f32x4{0.f, 0.f, 0.f, 0.f}, hi8, lo8);
        *reinterpret_cast<f16x4*>(wH + (tid >> 6) * 256 + off) = __builtin_shufflevector(hi8, hi8, 0, 1, 2, 3);
        *reinterpret_cast<f16x4*>(wL + (tid >> 6) * 256 + off) = __builtin_shufflevector(lo8, lo8, 0, 1, 2, 3);
        if (l == 0) sJ[tid >> 6] = sj;
      }
    }
    __syncthreads();                                         // masks, planes, W2^T shard in LDS
    const int t1 = wave & 1, j = wave >> 1;
    if (j < Ad) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int ws = 0; ws < 8; ++ws) {
        const float* bw = Wl + ((size_t)t1 * 8 + ws) * 512 + lane * 4;
        const f16x8 bh = __builtin_bit_cast(f16x8, ld4(bw)), bl = __builtin_bit_cast(f16x8, ld4(bw + 256));
        const u32x4v mk = *reinterpret_cast<const u32x4v*>(mH + i * 256 + (ws * 4 + kk) * 8);
        const u32x4v h4 = *reinterpret_cast<const u32x4v*>(wH + j * 256 + (ws * 4 + kk) * 8);
        const u32x4v l4 = *reinterpret_cast<const u32x4v*>(wL + j * 256 + (ws * 4 + kk) * 8);
        const f16x8 ah = __builtin_bit_cast(f16x8, h4 & mk), al = __builtin_bit_cast(f16x8, l4 & mk);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
      }
      const float un = PrecX2::kOut / sJ[j];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (row0 + 4 * kk + r < B)
          st_ag(A.gu + ((size_t)j * B + row0 + 4 * kk + r) * kW4 + 32 * c + 16 * t1 + i, gm1[r] > 0.f ? acc[r] * un : 0.f);
    }
    stamp();   // unit-seed rows requested out
  }
  if (A.whole) {
    // ... the critic's TILES of this launch wrote the critic's fp16 packs and biases (uncached memory): their flags
    // (one poller each), coherent loads after them (Coh<P>); the biases come from the tiles' uncached copies (the masters sit dirty in
    // another XCD's L2)
    if (tid < A.n_ct && A.debug_expire != 102) {      // (102, timing experiment: the critic's tiles count as done)
      bool ok = false;
      for (int spin = 0; spin < kTpSpin && !ok; ++spin) {
        ok = (unsigned)(__hip_atomic_load(A.ct_done + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == cx.epoch;
        if (!ok) __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) report_expired(A.err, (KERN_PHASE2 << 8) | SITE_DW_GATE);
    }
    __syncthreads();
    cbo.b0 = cx.cb[0]; cbo.b1 = cx.cb[1]; cbo.b2 = cx.cb[2];
  }
  if constexpr (GE) {
    // (every wave's unit-seed rows are out — uncached memory — before the member says so: the first-layer tiles read them after this)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(A.gu_flags + slice * NMC + c, (unsigned long long)cx.epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  stamp();
  float* qsum = nullptr;
  if (A.partials_a != nullptr && lead) {
    qsum = A.partials_a + slice * 4 + 1;
    if (tid == 0) { A.partials_a[slice * 4 + 0] = 0.f; A.partials_a[slice * 4 + 2] = 0.f; }
  }
  // (GE — k_ddpg_chain: the critic's packs and biases are this launch's tiles' — coherent loads, engine.h Coh)
  tp4_scalar_fb<typename std::conditional<GE, Coh<P>, P>::type, NMC>(A.critic, xa, h1, h2, g2, outS, scr, tp, nostore, row0, B, -A.inv_B, S, Ad, auxS, stamp, qsum, cbo);
  stamp();   // da ready
  // du = da (1 - pi^2): every member holds the same da; the lead member publishes (rows beyond B are never polled)
  if (lead && okp) {
    const float du = auxS[rp_ * kOutLd + cp_] * (1.f - vp * vp);
    granule_put(A.du_granules + (size_t)(row0 + rp_) * kDuLd + cp_, cx.epoch, du);
    A.adY[2][(size_t)(row0 + rp_) * A.alddo + cp_] = du;
  }
  if constexpr (GE) { stamp(); return -1; }      // (the first layer's dY left as unit-seed rows before the pass: nothing follows du)
  // ---- the backward step through the second hidden layer, this member's 32 columns of the first one's dY
  if constexpr (P::kX2 && !GE) {
    // g2[hr][hc ..] = (h2 > 0) sum_j du[hr][j] W3[j][hc ..] (wave = minibatch row hr; du once more from da and pi, both
    // in LDS since the pass: no exchange) -> h1 (the critic's buffers are free), each ROW scaled by the power of two
    // that brings its largest magnitude to [2^10, 2^11) — the fp16 split's range, as tp4_backward scales its tiles
    f32x4 gq = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool row_in = row0 + hr < B;
#pragma unroll
    for (int j = 0; j < kDuLd; ++j) {
      float d = 0.f;
      if (j < Ad && row_in) {
        const float pv = xa[hr * kX0Ld + S + j];
        d = auxS[hr * kOutLd + j] * (1.f - pv * pv);
      }
      gq += ld4(w3s + j * 256 + hc) * d;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) gq[t] = ((mbits >> t) & 1u) != 0u ? gq[t] : 0.f;
    float m = fmaxf(fmaxf(fabsf(gq[0]), fabsf(gq[1])), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
    m = wave_max(m);
    const float sr = P::a_scale(m);
    *reinterpret_cast<f32x4*>(h1 + hr * kWL4 + hc) = gq * sr;
    if (lane == 0) duS[hr] = PrecX2::kOut / sr;
  }
  __syncthreads();
  {
    // wave = (tile t1 of the member's two, macro step ws of eight): one split product, partial tile -> scr
    const int t1 = wave & 1, ws = wave >> 1;
    const float* hrow = h1 + i * kWL4 + 4 * kk;
    const float* bw = Wl + ((size_t)t1 * 8 + ws) * 512 + lane * 4;
    const FragX2 bf{ld4(bw), ld4(bw + 256)};
    const f32x4 un = ld4(duS + 4 * kk);                 // the un-scaling of this lane's four rows
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    PrecX2::mma3(ld4(hrow + 32 * ws), ld4(hrow + 32 * ws + 16), bf, acc);
    float* o = scr + wave * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[(4 * kk + r) * 16 + i] = acc[r] * un[r];
  }
  __syncthreads();
  if (tid < 512 && row0 + er < B) {
    float v = 0.f;
#pragma unroll
    for (int ws = 0; ws < 8; ++ws) v += scr[(2 * ws + et) * 256 + er * 16 + ec];
    v = (mbits & 16u) != 0u ? v : 0.f;
    granule_put(const_cast<unsigned long long*>(A.g1_granules) + ((size_t)(2 * c + et) * B + row0 + er) * 16 + ec, cx.epoch, v);
  }
  stamp();
  return -1;
}

template <class P>
__global__ __launch_bounds__(kThreads) void k_ddpg_phase2_dw(const DdpgArgs A, const DwKArgs D) {
  const DwKArgs* Dp = (const DwKArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kMergedDwOffset);
  const PassCtx cx{A.epoch, A.cluster_tag, A.w3_snap, {A.critic_b16[0], A.critic_b16[1], A.critic_b16[2]}, A.trace};
  ddpg_tile<P, DwKArgs>(Dp, ddpg_phase2m_body<P>(A, Dp, -1, cx), 2);
}

// ---------------------------------------------------------------------------------------------------------------
// The WHOLE update as one launch: k_ddpg_chain below (round 3's k_ddpg_update — one update per launch, tile workgroups as
// extra grid rows — was its first form and is gone: the chain launch with one update is the same thing with the tiles
// as continuations of the roles).  Offsets of the argument blocks inside the kernel-argument segment:
// ---------------------------------------------------------------------------------------------------------------
constexpr size_t kWholeDcOffset = (sizeof(DdpgArgs) + alignof(DwKArgs4) - 1) / alignof(DwKArgs4) * alignof(DwKArgs4);
constexpr size_t kWholeDaOffset = (kWholeDcOffset + sizeof(DwKArgs4) + alignof(DwKArgs4) - 1) / alignof(DwKArgs4) * alignof(DwKArgs4);

// ---------------------------------------------------------------------------------------------------------------
// SEVERAL updates as one launch (k_ddpg_chain; PrecX2 learners, DDPG, B <= 256, a chip that holds one update's
// workgroups at once): step_n's K-loop moves inside the launch.
//
// Every update is ONE block of 16 grid rows = (16 x slices) workgroups, dispatched in order A (8 rows) | B (4) | C (4),
// and every workgroup runs a CHAIN of stages for its update, handing results on through the flags and granules of
// k_ddpg_update:
//     role A (target chain -> seeds)        -> goes on as the critic pass's member (slice, member)
//     role B (critic forward + unit-seed backward), role C (actor forward)
//                                           -> the critic's tile -> the actor's tile (same index) [-> the next rows]
// No extra rows of tile / pass workgroups: nothing has to be dispatched inside an update.  The NEXT update's block is
// dispatched as compute units come free — role A's 128 workgroups are resident (rows in, waiting) several microseconds
// before the previous update's last tile raises its flag — so what an update boundary costs is one flag hop instead of
// end-of-kernel + dispatch + cold instruction caches + the rows' round trip.
// What crosses an update boundary without a kernel boundary:
//   * fp16 packs, biases (uncached copies b16 / bt16), the actor's output layer (w3buf[parity]): uncached memory, read
//     after the tiles' FIN flags (every store acknowledged), with coherent loads (engine.h ld4c / ldc / Coh<P>);
//   * masters and Adam moments: agent-scope loads / stores (dw_tile_x2.h), the same tile's next incarnation after FIN;
//   * the next rows: two staging sets by parity, gathered by 16 tile-less role-C workgroups, flag pf_done[slice].
// Update 0 of a launch reads what the launch before left (masters' biases, the row-major W3, staged or gathered rows).
// Waits (all on workgroups of the same or an EARLIER update, or — inside an update — on roles that are resident
// together: one update's 16 x slices workgroups must fit the chip, the launcher checks):
//   role A(u), C(u): every ct_fin / at_fin flag of update u - 1;  role B(u): ct_fin(u - 1);  all: pf_done[slice](u - 1).
// Write-after-read across updates is ordered by the same flags (role B(u + 1) rewrites the rows the critic's tiles of u
// read: after ct_fin(u); role C(u + 1) rewrites pi / the actor's activations: after at_fin(u); the prefetch of u rewrites
// the staging set the roles of u - 1 read: its workgroup was role C(u), i.e. after at_fin(u - 1); granules and exchange
// areas carry the update's tag).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chain_wait2(const unsigned long long* f0, int n0, const unsigned long long* f1, int n1,
                                            const unsigned long long* f2, int n2, unsigned tag, unsigned* err, unsigned code) {
  const int k = (int)threadIdx.x;
  const unsigned long long* f = k < n0 ? f0 + k : (k < n0 + n1 ? f1 + (k - n0) : (k < n0 + n1 + n2 ? f2 + (k - n0 - n1) : nullptr));
  if (f != nullptr) {
    bool ok = false;
    for (int spin = 0; spin < kTpSpin && !ok; ++spin) {
      ok = (unsigned)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) report_expired(err, code);
  }
  __syncthreads();
}

constexpr size_t kChainCOffset = (kWholeDaOffset + sizeof(DwKArgs4) + alignof(ChainArgs) - 1) / alignof(ChainArgs) * alignof(ChainArgs);
// which role row (A 0..7 | B 8..11 | C 12..15 | T 16..) the physical grid row y of an update is: the roles' DISPATCH
// order is a parameter of the launch (ChainArgs::order), everything else speaks of the logical row
__device__ __forceinline__ int chain_row(int y, int order) {
  if (y >= 16 || order == 0) return y;
  if (order == 1) return y < 4 ? y + 8 : (y < 12 ? y - 4 : y);          // B | A | C
  if (order == 2) return y < 8 ? y + 8 : y - 8;                         // B | C | A
  return y < 8 ? y : (y < 12 ? y + 4 : y - 4);                          // A | C | B
}

template <class P>
__global__ __launch_bounds__(kThreads) void k_ddpg_chain(const DdpgArgs A, const DwKArgs4 Dc, const DwKArgs4 Da, const ChainArgs C_) {
  static_assert(kDwTileX2, "the chain launch runs the 16 x 64 tiles of dw_tile_x2.h");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const DwKArgs4* Dcp = (const DwKArgs4*)(ka + kWholeDcOffset);
  const DwKArgs4* Dap = (const DwKArgs4*)(ka + kWholeDaOffset);
  // (read through the segment pointer: a run-time subscript into a by-value argument — c_step[u] — would make hipcc
  // pull the whole block into registers / scratch)
  const ChainArgs& C = *(const ChainArgs*)(ka + kChainCOffset);
  using LY = FusedLds<256>;
  constexpr int HB = kR * kWL4;
  const int slices = (int)gridDim.x, slice = (int)blockIdx.x;
  const int R = C.rows;                      // grid rows per update: A 8 | B 4 | C 4 | T (tile-only rows: small batches)
  const int u = (int)blockIdx.y / R, yy = chain_row((int)blockIdx.y - u * R, C.order);
  const int B = A.B, S = A.S, Ad = A.A, tid = threadIdx.x, row0 = slice * kR;
  const unsigned ep = A.epoch + (unsigned)u;
  const unsigned tag1 = A.cluster_tag + 2u * (unsigned)u, tag2 = tag1 + 1u;
  const int tc = Dcp->tile_end[kDwFusedItems - 1], ta = Dap->tile_end[kDwFusedItems - 1];
  const int par = u & 1;
  float* xa = smem + LY::xa;
  float* xb = smem + LY::xb;
  float* h1 = smem + LY::h;
  float* h2 = h1 + HB;
  float* outS = smem + LY::out;
  float* scr = smem + LY::scr;
  float* rS = smem + LY::misc;
  float* dS = rS + kR;
  float* yS = dS + kR;
  int* meta = reinterpret_cast<int*>(yS + kR);
  int* endsS = reinterpret_cast<int*>(smem + LY::misc + 96);
  // (role 3 = T: where roles B and C have fewer workgroups than a net has tiles — batches of fewer than 11 slices —
  // the remaining tiles are rows of workgroups of their own, dispatched behind role C)
  const int role = yy < 8 ? 0 : (yy < 12 ? 1 : (yy < 16 ? 2 : 3));
  const int member = yy < 8 ? yy : (yy < 16 ? (yy - 8) & 3 : 1);
  const bool lead = member == 0;
  const int spin = A.debug_expire == (int)SITE_CLUSTER ? 0 : kTpSpin;
  // in-kernel stage stamps (liboprl_amd_trace.so): update C.trace_u of the launch, the slots of round 3's k_ddpg_update
  const bool traced = kTraceOn && A.trace != nullptr && u == C.trace_u;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (traced && role < 3 && (tid & 63) == 0 && (tid == 0 || slice == 0) && lead && n_stamp < kTraceStamps) {
      const int slot = tid == 0 ? slice : 16 + (tid >> 6);
      long long* tr = A.trace + (((size_t)role * 64 + slot) * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();   // entry

  // ---- what the update before (of this launch) must have finished; then this slice's rows
  // (the rows first — they need the prefetch's flag only — then the flags of the tiles: a workgroup that is resident
  // before the update before has finished has its rows in LDS when the last tile flags)
  // (role A is resident long before the update before has finished: pf -> rows -> the tiles' flags; roles B and C are
  // dispatched when it is all but over: one look at everything they wait for)
  if (u > 0 && role == 0) chain_wait2(C.pf_done + slice, 1, nullptr, 0, nullptr, 0, ep - 1u, A.err, (KERN_PHASE1 << 8) | SITE_DW_GATE);
  if (u > 0 && role != 0) chain_wait2(C.ct_fin, tc, C.at_fin, role == 1 ? 0 : ta, C.pf_done + slice, role == 3 ? 0 : 1, ep - 1u, A.err, (KERN_PHASE1 << 8) | SITE_DW_GATE);
  if (role == 3) {
    // (a tile-only workgroup: no rows)
  } else if (u == 0 && C.first_gather) {
    load_batch(A.src, row0, B, S, Ad, xa, xb, rS, dS, meta, endsS);      // (gather = 1, this update's counter: the host's)
  } else {
    // staged rows of this update's parity (or, update 0 of an update() call, the caller's rows: set0)
    const float* ps = par ? C.set1[0] : C.set0[0]; const float* pa = par ? C.set1[1] : C.set0[1];
    const float* pr = par ? C.set1[2] : C.set0[2]; const float* pd = par ? C.set1[3] : C.set0[3];
    const float* ps2 = par ? C.set1[4] : C.set0[4];
    lds_zero(xa, 2 * kR * kX0Ld);   // xa and xb are adjacent
    __syncthreads();
    // (coherent loads: another workgroup of this launch staged these rows — engine.h ld4c / ldc)
    load_rows_c(xa, kX0Ld, 0, ps, S, S, row0, B);
    load_rows_c(xa, kX0Ld, S, pa, Ad, Ad, row0, B);
    load_rows_c(xb, kX0Ld, 0, ps2, S, S, row0, B);
    if (tid < kR) {
      const int gr = row0 + tid;
      rS[tid] = gr < B ? ldc(pr + gr) : 0.f;
      dS[tid] = gr < B ? ldc(pd + gr) : 0.f;
    }
  }
  stamp();   // batch rows requested
  if (u > 0 && role == 0) {
    __syncthreads();
    chain_wait2(C.ct_fin, tc, C.at_fin, ta, nullptr, 0, ep - 1u, A.err, (KERN_PHASE1 << 8) | SITE_DW_GATE);
  }
  stamp();   // the update before has finished
  const Tp3Store nostore{nullptr, nullptr, nullptr, nullptr, 0};
  // (later updates: the biases as the tiles of the update before left them — uncached copies; update 0: the masters)
  auto bias_of = [&](int which) {
    BiasOv bo;
    if (u > 0) { bo.b0 = C.b16[which][0]; bo.b1 = C.b16[which][1]; bo.b2 = C.b16[which][2]; }
    return bo;
  };

  if (role == 0) {
    // ---- role A: a' = tanh(actor_target(s')), q' = critic_target(s', a'), TD target, seeds       (ddpg.py:94-95)
    Tp tp{member, 8, A.xbuf + (size_t)slice * kTpStages * A.xnc * kTpBlk, tag1, 0, A.err, KERN_PHASE1 << 8, spin};
    tp.local = cluster_on_one_xcd(A);
    tp4_forward<Coh<P>, 8>(A.actor_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, stamp, bias_of(1));
    for (int idx = tid; idx < kR * Ad; idx += kThreads) {
      const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
      xb[row * kX0Ld + S + col] = gr < B ? tanhf(outS[row * kOutLd + col]) : 0.f;
    }
    // role B's q granules are requested from inside the second pass (after its layer-1 stage)
    // (role B's members leave their PARTIAL q — no exchange at the end of role B: summed here, in member order, with the
    // output bias: the cluster all-reduce's arithmetic)
    unsigned long long gq[4] = {0ull, 0ull, 0ull, 0ull};
    float qb = 0.f;
    int hook_n = 0;
    const unsigned long long* gq_src = C.qp + (size_t)slice * 64 + (tid & (kR - 1));
    auto hook = [&]() {
      stamp();
      if (++hook_n == 2) {
#pragma unroll
        for (int m = 0; m < 4; ++m) gq[m] = __hip_atomic_load(gq_src + m * kR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        qb = ldc(u > 0 ? C.b16[2][2] : A.critic.b[2]);
      }
    };
    tp4_forward<Coh<P>, 8>(A.critic_t, xb, h1, h2, outS, scr, tp, nostore, row0, B, hook, bias_of(3));
    if (lead && tid < 64) {
      const int gr = row0 + tid;
      const bool row_ok = tid < kR && gr < B;
      const float qn = row_ok ? outS[tid * kOutLd] : 0.f;
      const float y = row_ok ? rS[tid] + ((1.f - dS[tid]) * A.gamma) * qn : 0.f;
      const int lim = A.debug_expire == (int)SITE_TD_TARGET ? 0 : (1 << 20);
      float q = 0.f;
      if (row_ok) {
        bool ok = lim > 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) ok = ok && (unsigned)(gq[m] >> 32) == ep;
        if (A.debug_expire == 101) ok = true;      // (timing experiment, tools/what_if.py: role B's q counts as seen)
        for (int sp = 0; sp < lim && !ok; ++sp) {
#pragma unroll
          for (int m = 0; m < 4; ++m) gq[m] = __hip_atomic_load(gq_src + m * kR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = true;
#pragma unroll
          for (int m = 0; m < 4; ++m) ok = ok && (unsigned)(gq[m] >> 32) == ep;
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) report_expired(A.err, (KERN_PHASE1 << 8) | SITE_TD_TARGET);
        float qs = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) qs += __uint_as_float((unsigned)gq[m]);
        q = ok ? qs + qb : __builtin_nanf("");
        const float seed = 2.f * (q - y) * A.inv_B;
        A.cdY[2][(size_t)gr * A.clddo] = seed;
        __hip_atomic_store(A.y_granules + gr, ((unsigned long long)ep << 32) | (unsigned long long)__float_as_uint(seed),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (A.y_out != nullptr) A.y_out[gr] = y;
        if (A.q_out != nullptr) A.q_out[gr] = q;
      }
      if (A.partials_c != nullptr) {
        float v[3] = {row_ok ? (q - y) * (q - y) : 0.f, row_ok ? q : 0.f, row_ok ? y : 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float sum = row16_sum(v[k]);
          if (tid == 0) A.partials_c[(size_t)slice * 4 + k] = sum;
        }
      }
    }
    stamp();
    __syncthreads();       // (the pass below reuses this workgroup's LDS)
    // ---- ... and goes on as member `member` of this slice's critic pass (update and member formed again from the block
    // index: nothing of the role stays in a register through the pass)
    int by2 = (int)blockIdx.y;
    asm volatile("" : "+s"(by2));
    const int u2 = by2 / C.rows;
    const PassCtx cx{A.epoch + (unsigned)u2, A.cluster_tag + 2u * (unsigned)u2 + 1u, C.w3buf[u2 & 1], {C.b16[2][0], C.b16[2][1], C.b16[2][2]},
                     (kTraceOn && u2 == C.trace_u) ? A.trace2 : nullptr, A.actor_pb1_f32};
    (void)ddpg_phase2m_body<P, DwKArgs4, false, true>(A, Dap, chain_row(by2 - u2 * C.rows, C.order), cx);      // (GE: its pass runs on Coh<P>)
    return;
  }

  if (role == 1) {
    // ---- role B: q = critic(s, a) forward and its whole backward with unit seed (tp4_scalar_fb)      (ddpg.py:96-100)
    Tp tp{member, 4, A.xbuf + ((size_t)1 * slices + slice) * kTpStages * A.xnc * kTpBlk, tag1, 0, A.err, KERN_PHASE1 << 8, spin};
    tp.local = cluster_on_one_xcd(A);
    const Tp3Store st{A.cX[1], A.cX[2], A.cdY[1], A.cdY[0], A.cdY0_stride, B, true};
    if (lead) {        // (the input rows for the critic's first-layer tiles: out before the pass)
      __syncthreads();
      for (int idx = tid; idx < kR * (S + Ad); idx += kThreads) {
        const int row = idx / (S + Ad), col = idx - row * (S + Ad), gr = row0 + row;
        if (gr < B) __hip_atomic_store(A.cX[0] + (size_t)gr * A.cldx0 + col, xa[row * kX0Ld + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // q leaves as the members' partial sums (granules, no exchange): role A adds them up
    const QPart qp{C.qp + (size_t)slice * 64, ep};
    tp4_scalar_fb<Coh<P>>(A.critic, xa, h1, h2, h1 + 2 * HB, outS, scr, tp, st, row0, B, 1.f, 0, 0, nullptr, stamp, nullptr, bias_of(2), qp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(A.gate_flags + slice * 4 + member, (unsigned long long)ep << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stamp();   // rows flagged
  } else if (role == 2) {
    // ---- role C: actor(s) forward, pi = tanh(.) and the activations for the actor's tiles / the critic pass
    Tp tp{member, 4, A.xbuf + ((size_t)2 * slices + slice) * kTpStages * A.xnc * kTpBlk, tag1, 0, A.err, KERN_PHASE1 << 8, spin};
    tp.local = cluster_on_one_xcd(A);
    // (everything role C leaves is read by workgroups of this launch — the critic pass, the actor's tiles — behind its
    // flags: written through)
    const Tp3Store st{A.aX[1], A.aX[2], nullptr, nullptr, 0, 0, true};
    tp4_forward<Coh<P>>(A.actor, xa, h1, h2, outS, scr, tp, st, row0, B, stamp, bias_of(0));
    if (lead) {
      for (int idx = tid; idx < kR * Ad; idx += kThreads) {
        const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
        if (gr < B) st_ag(A.pi + (size_t)gr * Ad + col, tanhf(outS[row * kOutLd + col]));
      }
      store_rows_wt(xa, kX0Ld, A.aX[0], A.aldx0, S, row0, B);
      // update 0: the row-major output layer as the launch before left it -> w3buf[0] (later updates: the output
      // layer's tiles of the update before wrote w3buf[parity] themselves)
      if (u == 0 && slice == 0)
        for (int idx = tid * 4; idx < Ad * kW4; idx += kThreads * 4)
          st16_wt(C.w3buf[0], idx, ld4(A.w3_src + idx));
    }
    stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(A.w_flags + slice * 4 + member, (unsigned long long)ep << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- from here on the workgroup is (update, index among the 8 x slices workgroups of roles B and C) and nothing else:
  // both are formed AGAIN from the block index, laundered through an empty asm, so that no value of the role above stays
  // in a register through the tiles (these kernels sit at the scalar-register limit; what does not fit is spilled)
  int by2 = (int)blockIdx.y, bx2 = (int)blockIdx.x, nx2 = (int)gridDim.x;
  asm volatile("" : "+s"(by2), "+s"(bx2), "+s"(nx2));
  // (index among the tile workgroups: role B slice-major, role C behind it, then T's rows)
  const int u2 = by2 / C.rows, y2 = chain_row(by2 - u2 * C.rows, C.order);
  const int wg2 = y2 < 16 ? ((y2 >= 12) ? 4 * nx2 : 0) + bx2 * 4 + ((y2 - 8) & 3) : 8 * nx2 + (y2 - 16) * nx2 + bx2;
  const int n_bc = 8 * nx2;
  const unsigned ep2 = A.epoch + (unsigned)u2;

  // ---- the next update's rows: the last `slices` of the tile-capable workgroups (roles B and C, then T's rows) — which
  // carry no tile (chain_tile_rows leaves room for them)
  {
    const int p = n_bc + (C.rows - 16) * nx2 - 1 - wg2;
    if (p >= 0 && p < nx2 && (u2 + 1 < C.n_upd || C.pf_last)) {
      const int par2 = u2 & 1;
      BatchSrc nx = A.next;                 // (the replay's view; gather = 1)
      nx.s = par2 ? C.set0[0] : C.set1[0]; nx.a = par2 ? C.set0[1] : C.set1[1]; nx.r = par2 ? C.set0[2] : C.set1[2];
      nx.d = par2 ? C.set0[3] : C.set1[3]; nx.s2 = par2 ? C.set0[4] : C.set1[4];
      nx.counter = A.next.counter + (unsigned long long)u2;
      prefetch_rows_src<true>(nx, A.S, A.A, A.B, p, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(C.pf_done + p, (unsigned long long)ep2 << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // ---- the critic's tile (GATE 1: rows of role B, seeds of role A), then the actor's (GATE 2: role C's rows — this
  // launch's, uncached: its flags, then coherent loads — and du)
  // (two inlined copies of the tile code, each with its kind folded in: ONE copy inside a two-trip loop keeps the
  // thread-index arithmetic of both trips live across the tile and spills 14 vector registers)
  {
    const int tile = wg2 < Dcp->tile_end[kDwFusedItems - 1] ? wg2 : -1;
    if (tile >= 0) {
      DwX2Ovr ov;
      ov.chain = &C; ov.u = u2;
      dw_tile_x2<DwKArgs4, P>(*Dcp, smem, tile, 1, ov);
    }
  }
  {
    int by3 = (int)blockIdx.y, bx3 = (int)blockIdx.x, nx3 = (int)gridDim.x;
    asm volatile("" : "+s"(by3), "+s"(bx3), "+s"(nx3));
    const int u3 = by3 / C.rows, y3 = chain_row(by3 - u3 * C.rows, C.order);
    const int wg3 = y3 < 16 ? ((y3 >= 12) ? 4 * nx3 : 0) + bx3 * 4 + ((y3 - 8) & 3) : 8 * nx3 + (y3 - 16) * nx3 + bx3;
    const int tile = wg3 < Dap->tile_end[kDwFusedItems - 1] ? wg3 : -1;
    if (tile >= 0) {
      __syncthreads();
      wait_flags(A.w_flags, 4 * nx3, A.epoch + (unsigned)u3, A.err, (KERN_PHASE2 << 8) | SITE_DW_GATE);
      // (its own masters and moments: this tile's incarnation of the update before — implied by the seeds a workgroup
      // that was a critic tile has just consumed, said for the one that was not)
      if (u3 > 0 && threadIdx.x == kThreads - 1) {
        bool ok = false;
        for (int sp = 0; sp < kTpSpin && !ok; ++sp) {
          ok = (unsigned)(__hip_atomic_load(C.at_fin + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == A.epoch + (unsigned)u3 - 1u;
          if (!ok) __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) report_expired(A.err, (KERN_PHASE2 << 8) | SITE_DW_GATE);
      }
      __syncthreads();
      DwX2Ovr ov;
      ov.chain = &C; ov.u = u3;
      dw_tile_x2<DwKArgs4, P>(*Dap, smem, tile, 2, ov);
    }
  }
}

static_assert(FusedLds<256>::total >= kDwLdsFloats && FusedLds<256>::total >= DwLds<16>::floats && FusedLds<256>::total >= DwX2Lds::floats, "a tile workgroup fits the phase kernels' LDS");
bool fused_x2_tiles() { return kDwTileX2; }
bool fused_tile64_all() { return kDwTileX2 && kMergedTile64; }

size_t fused_ddpg_lds_bytes() { return sizeof(float) * FusedLds<256>::total; }
size_t fused_xbuf_granules_per_cluster(int nc) { return (size_t)kTpStages * nc * kTpBlk; }

// Does this device deal the workgroups of a (slices, rows) grid out round robin over eight XCDs — block (x, y) on XCD
// (x + slices y) % 8, hence a whole column on XCD x % 8 for slices = 16?  Probed once per device (three launches).
bool xcd_map_ok() {
  static std::mutex mu;
  static int known[16] = {0};            // 0: not probed, 1: yes, -1: no
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lk(mu);
  int& k = known[dev & 15];
  if (k != 0) return k > 0;
  k = -1;
  int* d = nullptr;
  if (hipMalloc(&d, 256 * sizeof(int)) != hipSuccess) return false;
  bool ok = true;
  int h[256];
  for (int rep = 0; rep < 3 && ok; ++rep) {
    ok = hipMemset(d, 0xff, 256 * sizeof(int)) == hipSuccess;
    if (ok) hipLaunchKernelGGL(k_xcc_probe, dim3(16, 16), dim3(64), 0, 0, d);
    ok = ok && hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) == hipSuccess;
    for (int b = 0; b < 256 && ok; ++b) ok = h[b] == (b & 7);
  }
  (void)hipFree(d);
  if (ok) k = 1;
  return ok;
}

hipError_t init_fused_attrs() {
  const void* ks[] = {reinterpret_cast<const void*>(&k_ddpg_phase1<256, false, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, false, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, false, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, true, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, false, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, true, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, false, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, true, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, false, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, true, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, false, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, false, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, false, PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, false, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, false, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecF32, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecBF16, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, false, PrecX2, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, true, PrecF32, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, true, PrecBF16, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_group<256, true, true, PrecX2, false>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, true, PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, true, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_group<256, true, true, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, false, PrecF32, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, false, PrecF32, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1<256, true, false, PrecX2, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2<256, true, false, PrecX2, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecX2, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecF32, false, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecBF16, false, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_dw<PrecX2, false, true>),
                      reinterpret_cast<const void*>(&k_ddpg_phase2_dw<PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_chain<PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_chain<PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_chain<PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<false, PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<false, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<false, PrecBF16>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<true, PrecF32>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<true, PrecX2>),
                      reinterpret_cast<const void*>(&k_ddpg_phase1_rt2<true, PrecBF16>)};
  for (const void* k : ks) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Group launches (N3: packed learners; DDPG).  `a0` = learner 0's arguments (for the grid), `batch_dev` = all of them in
// device memory.  Two forms:
//   nc = 1: the generic single-CU-per-slice passes (exact fp32) — a workgroup never waits for a later one of its launch;
//   nc = 4: the lean passes on clusters of four, in every precision.  A cluster's members are 16 workgroups apart in
//           dispatch order (rows of one learner's block) and land on ONE XCD (16 = 0 mod 8), whose dispatcher hands
//           out its workgroups in order: at most two clusters per XCD are partly resident at any time, every other
//           resident workgroup belongs to a complete cluster and retires — any number of learners may queue up.
bool fused_ddpg_is_lean(const DdpgArgs& a);
// (TD3 / SAC members: lean passes only, the twin critics' co-residency forms — twin_split, p2_pair — off)
template <class P>
static void launch_p1_group_lean(const DdpgArgs& a0, dim3 grid, const DdpgArgs* batch_dev, hipStream_t st) {
  const dim3 blk(kThreads);
  const size_t lds = fused_ddpg_lds_bytes();
  if (a0.sac) hipLaunchKernelGGL((k_ddpg_phase1_group<256, true, true, P, false>), grid, blk, lds, st, batch_dev);
  else if (a0.n_critics == 2) hipLaunchKernelGGL((k_ddpg_phase1_group<256, true, false, P, false>), grid, blk, lds, st, batch_dev);
  else hipLaunchKernelGGL((k_ddpg_phase1_group<256, true, false, P, true>), grid, blk, lds, st, batch_dev);
}
template <class P>
static void launch_p2_group_lean(const DdpgArgs& a0, dim3 grid, const DdpgArgs* batch_dev, hipStream_t st) {
  const dim3 blk(kThreads);
  const size_t lds = fused_ddpg_lds_bytes();
  if (a0.sac) hipLaunchKernelGGL((k_ddpg_phase2_group<256, true, true, P>), grid, blk, lds, st, batch_dev);
  else hipLaunchKernelGGL((k_ddpg_phase2_group<256, true, false, P>), grid, blk, lds, st, batch_dev);
}
hipError_t launch_ddpg_phase1_group(const DdpgArgs& a0, const DdpgArgs* batch_dev, int n, hipStream_t st) {
  if (a0.merged || a0.wide || a0.whole || a0.twin_split || a0.prefetch_p1 || (a0.sac && a0.n_critics != 2)) return hipErrorInvalidValue;
  dim3 grid((a0.B + kR - 1) / kR, (2 + a0.n_critics) * a0.nc, n);
  if (a0.nc == 4 && fused_ddpg_is_lean(a0)) {
    if (a0.x2) launch_p1_group_lean<PrecX2>(a0, grid, batch_dev, st);
    else if (a0.bf16) launch_p1_group_lean<PrecBF16>(a0, grid, batch_dev, st);
    else launch_p1_group_lean<PrecF32>(a0, grid, batch_dev, st);
    return hipGetLastError();
  }
  if (a0.nc != 1 || a0.bf16 || a0.x2 || a0.sac || a0.n_critics != 1) return hipErrorInvalidValue;
  grid = dim3((a0.B + kR - 1) / kR, 1, 3 * n);      // role-major (k_ddpg_phase1_group)
  hipLaunchKernelGGL((k_ddpg_phase1_group<256, false, false>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, batch_dev);
  return hipGetLastError();
}
hipError_t launch_ddpg_phase2_group(const DdpgArgs& a0, const DdpgArgs* batch_dev, int n, hipStream_t st) {
  if (a0.merged || a0.wide || a0.whole || a0.p2_pair || (a0.sac && a0.n_critics != 2)) return hipErrorInvalidValue;
  const dim3 grid((a0.B + kR - 1) / kR, a0.nc + (a0.prefetch_next ? 1 : 0), n);
  if (a0.nc == 4 && fused_ddpg_is_lean(a0)) {
    if (a0.x2) launch_p2_group_lean<PrecX2>(a0, grid, batch_dev, st);
    else if (a0.bf16) launch_p2_group_lean<PrecBF16>(a0, grid, batch_dev, st);
    else launch_p2_group_lean<PrecF32>(a0, grid, batch_dev, st);
    return hipGetLastError();
  }
  if (a0.nc != 1 || a0.bf16 || a0.x2 || a0.sac || a0.n_critics != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL((k_ddpg_phase2_group<256, false, false>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, batch_dev);
  return hipGetLastError();
}

// the lean tp4 passes serve clusters of 4 whose four nets fit tp4_shape_ok
static bool lean_ok(const DdpgArgs& a) { return fused_ddpg_is_lean(a); }
// (the twin-critic variant, TD3, exists in the lean form only: learner.hip falls back to the
// generic launch sequence when this returns false)
bool fused_ddpg_is_lean(const DdpgArgs& a) {
  return a.nc == 4 && !a.no_lean && tp4_shape_ok(256, a.S + a.A, 1) && tp4_shape_ok(256, a.S, a.sac ? 2 * a.A : a.A);
}

// phase 1 with the critic's dW + Adam tiles as extra grid rows (DdpgArgs::merged bit 0; `d` = fill_dw_kargs of
// that launch with its gate filled in)
hipError_t launch_ddpg_phase1_dw(const DdpgArgs& a, const DwKArgs& d, hipStream_t st) {
  const bool wide = (a.wide & 1) != 0;
  if (!lean_ok(a) || a.sac || a.n_critics < 1 || a.n_critics > 2 || (a.merged & 1) == 0 || (wide && (!a.x2 || a.xnc < 8 || a.n_critics != 1))) return hipErrorInvalidValue;
  const int slices = (a.B + kR - 1) / kR;
  const int tiles = d.tile_end[kDwMaxItems - 1];
  const dim3 grid(slices, (2 + a.n_critics) * a.nc + (wide ? 4 : 0) + (tiles + slices - 1) / slices + (a.prefetch_p1 ? 1 : 0));
  if (a.n_critics == 2) {
    if (a.seed2_granules == nullptr) return hipErrorInvalidValue;
    if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecX2, false, true>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
    else if (a.bf16) hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecBF16, false, true>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
    else hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecF32, false, true>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
    return hipGetLastError();
  }
  if (a.x2 && wide) hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecX2, true>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
  else if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecX2>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
  else if (a.bf16) hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecBF16>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
  else hipLaunchKernelGGL((k_ddpg_phase1_dw<PrecF32>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
  return hipGetLastError();
}

// phase 2 with the actor's dW + Adam tiles as extra grid rows (DdpgArgs::merged bit 1; `d` = fill_dw_kargs of the
// actor's launch with the phase-2 gate filled in)
hipError_t launch_ddpg_phase2_dw(const DdpgArgs& a, const DwKArgs& d, hipStream_t st) {
  if (!lean_ok(a) || a.sac || !a.x2 || !kDwTileX2 || (a.merged & 2) == 0 || (a.wide & 2) == 0 || a.xnc < 8 || a.A > kDuLd || a.B > 256)
    return hipErrorInvalidValue;
  const int slices = (a.B + kR - 1) / kR;
  const int tiles = d.tile_end[kDwMaxItems - 1];
  const dim3 grid(slices, 8 + (a.prefetch_next ? 1 : 0) + (tiles + slices - 1) / slices);
  hipLaunchKernelGGL((k_ddpg_phase2_dw<PrecX2>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, d);
  return hipGetLastError();
}

// n_upd updates as one launch (k_ddpg_chain): `a` / `dc` / `da` with both gates filled in (the gates' tags = the FIRST
// update's epoch), `c` = what changes per update
hipError_t launch_ddpg_chain(const DdpgArgs& a, const DwKArgs4& dc, const DwKArgs4& da, const ChainArgs& c, hipStream_t st) {
  if (!lean_ok(a) || a.sac || (a.bf16 && a.actor_pb1_f32 == nullptr) || !kDwTileX2 || a.n_critics != 1 || !a.whole || (a.merged & 3) != 3 || (a.wide & 3) != 3 ||
      a.xnc < 8 || a.A > kDuLd || a.B > 256 || a.prefetch_next || a.prefetch_p1 || c.n_upd < 1 || c.n_upd > kChainMax)
    return hipErrorInvalidValue;
  const int slices = (a.B + kR - 1) / kR;
  const int tc = dc.tile_end[kDwFusedItems - 1], ta = da.tile_end[kDwFusedItems - 1];
  if (tc + ta + 1 > kThreads) return hipErrorInvalidValue;       // (one poller per flag: chain_wait2)
  const int mt = tc > ta ? tc : ta;
  const int rows = 16 + chain_tile_rows(mt, slices);    // A 8 | B 4 | C 4 | T
  if (c.rows != rows) return hipErrorInvalidValue;
  const dim3 grid(slices, rows * c.n_upd);
  if (a.x2) hipLaunchKernelGGL((k_ddpg_chain<PrecX2>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, dc, da, c);
  else if (a.bf16) hipLaunchKernelGGL((k_ddpg_chain<PrecBF16>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, dc, da, c);
  else hipLaunchKernelGGL((k_ddpg_chain<PrecF32>), grid, dim3(kThreads), fused_ddpg_lds_bytes(), st, a, dc, da, c);
  return hipGetLastError();
}

hipError_t launch_ddpg_phase1(const DdpgArgs& a, hipStream_t st) {
  const int slices = (a.B + kR - 1) / kR;
  const dim3 blk(kThreads);
  const size_t lds = fused_ddpg_lds_bytes();
  const int pf = a.prefetch_p1 ? 1 : 0;
  if ((a.wide & 1) != 0) {      // role A on clusters of 8 (DDPG, lean passes; learner.hip decides)
    if (!lean_ok(a) || a.sac || a.bf16 || a.n_critics != 1 || a.xnc < 8) return hipErrorInvalidValue;
    const dim3 grid(slices, a.nc + 8 + a.nc + pf);
    if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1<256, true, false, PrecX2, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((k_ddpg_phase1<256, true, false, PrecF32, true>), grid, blk, lds, st, a);
    return hipGetLastError();
  }
  if (a.rt2) {      // the B roles on 32-row slices (learner.hip decides: lean, clusters of four, slices a multiple of 16)
    if (!lean_ok(a) || a.nc != 4 || (slices & 15) != 0 || pf || a.merged != 0) return hipErrorInvalidValue;
    if (a.rt2 == 2 && (!a.sac || a.twin_split || !a.do_actor)) return hipErrorInvalidValue;   // (role A carries role C's pass: SAC's one actor)
    const dim3 g2(slices, 2 * a.n_critics + (a.rt2 == 2 ? 1 : 2) * a.nc);
    const size_t l2 = std::max(lds, sizeof(float) * (size_t)FusedLdsB2::total);
    if (a.sac) {
      if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1_rt2<true, PrecX2>), g2, blk, l2, st, a);
      else if (a.bf16) hipLaunchKernelGGL((k_ddpg_phase1_rt2<true, PrecBF16>), g2, blk, l2, st, a);
      else hipLaunchKernelGGL((k_ddpg_phase1_rt2<true, PrecF32>), g2, blk, l2, st, a);
    } else {
      if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1_rt2<false, PrecX2>), g2, blk, l2, st, a);
      else if (a.bf16) hipLaunchKernelGGL((k_ddpg_phase1_rt2<false, PrecBF16>), g2, blk, l2, st, a);
      else hipLaunchKernelGGL((k_ddpg_phase1_rt2<false, PrecF32>), g2, blk, l2, st, a);
    }
    return hipGetLastError();
  }
  const dim3 grid(slices, (2 + a.n_critics) * a.nc + pf);
  if (a.bf16 || a.x2) {
    // the 16-bit MFMA modes exist for the lean passes only; the nets' pf / pb point at bf16 / two-plane fp16 packs
    if (!lean_ok(a)) return hipErrorInvalidValue;
    if (a.x2 && a.sac) hipLaunchKernelGGL((k_ddpg_phase1<256, true, true, PrecX2>), grid, blk, lds, st, a);
    else if (a.x2) hipLaunchKernelGGL((k_ddpg_phase1<256, true, false, PrecX2>), grid, blk, lds, st, a);
    else if (a.sac) hipLaunchKernelGGL((k_ddpg_phase1<256, true, true, PrecBF16>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((k_ddpg_phase1<256, true, false, PrecBF16>), grid, blk, lds, st, a);
  } else if (a.sac) {
    if (!lean_ok(a)) return hipErrorInvalidValue;   // SAC is fused in the lean form only (use_fused() checks)
    hipLaunchKernelGGL((k_ddpg_phase1<256, true, true>), grid, blk, lds, st, a);
  } else if (lean_ok(a)) {
    hipLaunchKernelGGL((k_ddpg_phase1<256, true, false>), grid, blk, lds, st, a);
  } else {
    hipLaunchKernelGGL((k_ddpg_phase1<256, false, false>), grid, blk, lds, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_ddpg_phase2(const DdpgArgs& a, hipStream_t st) {
  const int slices = (a.B + kR - 1) / kR;
  const dim3 blk(kThreads);
  const size_t lds = fused_ddpg_lds_bytes();
  // SAC side by side: one cluster per online critic; + the next-minibatch gather row
  if ((a.wide & 2) != 0) {      // the critic pass on clusters of 8 (DDPG / TD3, lean passes)
    if (!lean_ok(a) || a.sac || a.bf16 || a.xnc < 8) return hipErrorInvalidValue;
    const dim3 grid(slices, 8 + (a.prefetch_next ? 1 : 0));
    if (a.x2) hipLaunchKernelGGL((k_ddpg_phase2<256, true, false, PrecX2, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((k_ddpg_phase2<256, true, false, PrecF32, true>), grid, blk, lds, st, a);
    return hipGetLastError();
  }
  const dim3 grid(slices, a.nc * ((a.sac && a.p2_pair) ? 2 : 1) + (a.prefetch_next ? 1 : 0));
  if (a.bf16 || a.x2) {
    if (!lean_ok(a)) return hipErrorInvalidValue;
    if (a.x2 && a.sac) hipLaunchKernelGGL((k_ddpg_phase2<256, true, true, PrecX2>), grid, blk, lds, st, a);
    else if (a.x2) hipLaunchKernelGGL((k_ddpg_phase2<256, true, false, PrecX2>), grid, blk, lds, st, a);
    else if (a.sac) hipLaunchKernelGGL((k_ddpg_phase2<256, true, true, PrecBF16>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((k_ddpg_phase2<256, true, false, PrecBF16>), grid, blk, lds, st, a);
  } else if (a.sac) {
    if (!lean_ok(a)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_ddpg_phase2<256, true, true>), grid, blk, lds, st, a);
  } else if (lean_ok(a)) {
    hipLaunchKernelGGL((k_ddpg_phase2<256, true, false>), grid, blk, lds, st, a);
  } else {
    hipLaunchKernelGGL((k_ddpg_phase2<256, false, false>), grid, blk, lds, st, a);
  }
  return hipGetLastError();
}

}  // namespace oprl
