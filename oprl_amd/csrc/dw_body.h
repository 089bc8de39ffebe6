// dw_body.h — the body of k_dw_adam (dW = dY^T X on 16 x 32 tiles fused with Adam, Polyak and the pack
// rewrite) and the temperature's Adam step, as inline device code: k_dw_adam / k_dw_adam_group (kernels.hip)
// run it as launches of their own, the merged phase kernels (fused_ddpg.hip) run it in extra workgroups of a
// phase launch.
#pragma once
#include "kernels.h"
#include "tp3.h"

namespace oprl {

// log_alpha Adam step in float64 like the reference's 0-dim double tensor
// (sac.py:65-70,132-141; tqc.py:105,163,175-177): grad = -(H_target + mean logp).
// If grad_out != nullptr only the gradient is exported (data-parallel mode).
__device__ __forceinline__ double shfl_xor_f64(double x, int m) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __shfl_xor(lo, m);
  hi = __shfl_xor(hi, m);
  return __hiloint2double(hi, lo);
}

// bc1 = 1 - beta1^step and bc2_sqrt = sqrt(1 - beta2^step) arrive from the host (it knows the
// step; a device-side double pow() alone cost several microseconds of this scalar update).
// (the first 256 threads of the block sum logp — the same partition and order whether the block is the
// 256-thread k_alpha_step or the extra 512-thread workgroup of a k_dw_adam launch)
__device__ __forceinline__ void alpha_step_block(double* log_alpha, double* m, double* v, const float* logp, int B,
                                                 float target_entropy, double lr, double beta1, double beta2, double eps,
                                                 double bc1, double bc2_sqrt, double* grad_out, const double* grad_in,
                                                 float grad_scale) {
  __shared__ double red[4];
  double s = 0.0;
  if (grad_in == nullptr) {
    if (threadIdx.x < 256) {
      for (int idx = threadIdx.x; idx < B; idx += 256) s += (double)logp[idx];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) s += shfl_xor_f64(s, o);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  double g;
  if (grad_in != nullptr) {
    g = *grad_in * (double)grad_scale;
  } else {
    // reference: mean of fp32 logp in fp32, then promoted
    const float mean32 = (float)((((red[0] + red[1]) + red[2]) + red[3]) / (double)B);
    g = -((double)target_entropy + (double)mean32);
  }
  if (grad_out != nullptr) { *grad_out = g; return; }
  double mm = *m, vv = *v;
  mm = mm + (g - mm) * (1.0 - beta1);
  vv = vv * beta2 + (1.0 - beta2) * g * g;
  const double denom = sqrt(vv) / bc2_sqrt + eps;
  *log_alpha = *log_alpha - (lr / bc1) * (mm / denom);
  *m = mm;
  *v = vv;
}


// ---------------------------------------------------------------------------
// dW[n,k] = sum_b dY[b,n] X[b,k]  on a 32x32 tile per workgroup; the 4 waves
// split the minibatch; db = column sums of dY come for free from the A operand.
// Epilogue (per element, torch.optim.Adam single-tensor semantics):
//   m += (g-m)(1-b1);  v = b2 v + (1-b2) g g;  th -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps)
//   th_t = (1-tau) th_t + tau th                       (Polyak, nn_functions.py:5-10)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void adam_bias_corr(const AdamScalars& ad, float* step_size, float* bc2_sqrt) {
  if (ad.step_dev == nullptr) {   // host knows the step: corrections arrive precomputed (double math)
    *step_size = ad.step_size_host;
    *bc2_sqrt = ad.bc2_sqrt_host;
    return;
  }
  const int step = ad.step_base + *ad.step_dev;
  const double bc1 = 1.0 - pow(ad.beta1_d, (double)step);
  const double bc2 = 1.0 - pow(ad.beta2_d, (double)step);
  *step_size = (float)(ad.lr_d / bc1);
  *bc2_sqrt = (float)sqrt(bc2);
}

// XCHG (data-parallel learner on peer windows, csrc/p2p.hip): between the GEMM and the epilogue every
// workgroup all-reduces its gradient tile with the same tile of the other ranks — tagged granules into
// the peers' windows, a bounded per-element wait for the peers' granules, sum in rank order.  A separate template instance: the single-rank kernel is untouched.  All workgroups of the
// launch must be resident (they wait for their counterparts on the other GPUs): at most a few hundred.

// LDS of one tile workgroup (floats): the partial tiles, the wave-private row staging, the bias partials, two scalars
// (NW waves split the 256-row chunk: 8 waves x 32 rows in the stand-alone kernel, 16 x 16 in the actor's tiles on phase 2)
template <int NW> struct DwLds {
  static constexpr int RW = 256 / NW;                        // minibatch rows per wave and chunk
  static constexpr int part = 0;
  static constexpr int stA = part + NW * kDwTileN * (kDwTile + 4);
  static constexpr int stX = stA + NW * RW * kDwTileN;
  static constexpr int bpart = stX + NW * RW * (kDwTile + 4);
  static constexpr int sc = bpart + NW * kDwTileN;
  static constexpr int floats = sc + 4;
};
constexpr int kDwLdsFloats = DwLds<kDwWaves>::floats;            // 72.2 KB

// one bounded wait of a GATED tile workgroup for `n` flag granules {tag, *} (lanes of wave w poll, everybody is
// released by the barrier that follows in the caller)
__device__ __forceinline__ bool dw_gate_wait(const unsigned long long* flags, int n, unsigned tag, int spin_max) {
  bool ok = true;
  for (int k = (int)threadIdx.x; k < n; k += kDwThreads) {
    bool mine = false;
    for (int spin = 0; spin < spin_max && !mine; ++spin) {
      mine = (unsigned)(__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
      if (!mine) __builtin_amdgcn_s_sleep(2);
    }
    ok = ok && mine;
  }
  return ok;
}
// loads that see what another XCD has just written through (sc1): a GATED tile's rows come in through ld4_agent
// (engine.h: raw buffer loads with the cache policy in the instruction — loads the compiler counts), the 4-byte seeds as
// agent-scope atomic loads
__device__ __forceinline__ float ld1_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `lds`: kDwLdsFloats floats of LDS (a static array of the stand-alone kernels, the dynamic region of a merged
// phase launch).  GATED (merged phase launches, csrc/fused_ddpg.hip): the tile workgroup starts while the phase's
// roles are still running — layer lookup and the Adam-state requests go ahead — and waits for flag granules
// before it touches what they write: `gate_rows` (every producer of X / dY rows) before the row requests,
// `gate_seed` (the per-row seeds, and the output layer's dY) before those; rows and seeds are read with sc1 loads
// (the producers write them through: no kernel boundary lies in between).
// The updated 16 x 32 tile (tileW: online, tileT: target; staged in LDS, zero outside the matrix) into the fragment
// packs, in pack order as 16-byte stores: the fp32 packs (pf / pb / tpf, unless null: a PrecX2 learner's fused update)
// and the 16-bit packs (pf16 / pb16 / tpf16: bf16, or — I.x2 — blocks of two fp16 planes holding 2^8 w).
__device__ __forceinline__ void dw_write_packs(const DwItem& I, const float (*tileW)[kDwTile + 4], const float (*tileT)[kDwTile + 4],
                                               int tid, int tk, int n_base, int n_off, int ptile, int TNi, int NSk, int NSn,
                                               bool polyak) {
  if (tid < 384 && I.pf != nullptr) {
    const int which = tid >> 7, q = tid & 127;
    const int blk = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
    if (which == 1) {            // W^T pack: tiles over k, steps over n; we own n in [n_off, n_off + TNi)
      const int ktile = 2 * tk + blk;
      if (I.pb != nullptr && ktile < NSk && 4 * lk >= n_off && 4 * lk < n_off + TNi) {
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = tileW[4 * lk + t][16 * blk + li];
        *reinterpret_cast<f32x4*>(I.pb + (((size_t)ktile * NSn + ptile) * 64 + l) * 4) = v;
      }
    } else {                     // W pack (online, target): tiles over n, steps over k
      const int kstep = 2 * tk + blk;
      float* dst = which == 0 ? I.pf : (polyak ? I.tpf : nullptr);
      if (dst != nullptr && kstep < NSk && li >= n_off && li < n_off + TNi) {
        const float (*src)[kDwTile + 4] = which == 0 ? tileW : tileT;
        *reinterpret_cast<f32x4*>(dst + (((size_t)ptile * NSk + kstep) * 64 + l) * 4) =
            *reinterpret_cast<const f32x4*>(&src[li][16 * blk + 4 * lk]);
      }
    }
  }
  if (I.pf16 != nullptr) {
    // the 16-bit packs: a macro step = fp32 steps 2s, 2s + 1 side by side, so this 16 x 32 tile is ONE forward
    // fragment block (64 lanes x 16 B, online and target) and, for W^T, one HALF (8 B per lane) of a block for each
    // of its two 16-row k tiles — the other half belongs to the neighbouring n tile.
    // PrecX2 packs (I.x2): the same positions in blocks of TWO fp16 planes — hi = fp16(2^8 w), lo = fp16(2^8 w - hi),
    // the lo plane 256 floats behind the hi plane.
    const int NSk2 = cdiv(I.K, 32), NSn2 = cdiv(I.N, 32);
    const bool x2 = I.x2 != 0;
    const size_t BK16 = x2 ? 512 : 256;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    if (tid < 128) {             // W packs: which = online / target
      const int which = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
      float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
      if (dst != nullptr && li >= n_off && li < n_off + TNi) {
        const float (*src)[kDwTile + 4] = which == 0 ? tileW : tileT;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&src[li][4 * lk]), v1 = *reinterpret_cast<const f32x4*>(&src[li][16 + 4 * lk]);
        float* d = dst + ((size_t)ptile * NSk2 + tk) * BK16 + (size_t)l * 4;
        if (x2) {
          f16x8 hi, lo;
          x2_split8(v0 * PrecX2::kWScale, v1 * PrecX2::kWScale, hi, lo);
          *reinterpret_cast<f16x8*>(d) = hi;
          *reinterpret_cast<f16x8*>(d + 256) = lo;
        } else {
          *reinterpret_cast<bf16x8*>(d) = cvt_bf16x8(v0, v1);
        }
      }
    } else if (tid < 256 && I.pb16 != nullptr) {   // W^T pack: k tile 2 tk + blk, n step n_base / 32, half (n_base / 16) & 1
      const int q = tid - 128, blk = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
      const int ktile = 2 * tk + blk;
      if (16 * ktile < I.K && 4 * lk >= n_off && 4 * lk < n_off + TNi) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = tileW[4 * lk + t][16 * blk + li];
        float* d = I.pb16 + ((size_t)ktile * NSn2 + (n_base >> 5)) * BK16 + (size_t)l * 4 + 2 * ((n_base >> 4) & 1);
        if (x2) {
          const f32x4 vs = v * PrecX2::kWScale;
          const f16x4 hi = __builtin_convertvector(vs, f16x4);
          const f32x4 r = vs - __builtin_convertvector(hi, f32x4);
          *reinterpret_cast<f16x4*>(d) = hi;
          *reinterpret_cast<f16x4*>(d + 256) = __builtin_convertvector(r, f16x4);
        } else {
          *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(v, bf16x4);
        }
      }
    }
  }
}


// GATE: 0 = a launch of its own, 1 = the critic's tiles on phase 1's launch (above).  (The actor's tiles on phase 2's
// launch — GATE 2 of DwGate — exist as PrecX2 tiles only: csrc/dw_tile_x2.h.)
template <bool XCHG, int GATE = 0, int NW = kDwWaves, class KAT = DwKArgs>
__device__ __forceinline__ void dw_adam_body(const KAT& A, float* lds, int bx) {   // bx: the tile (bx of a stand-alone launch)
  constexpr bool GATED = GATE == 1;
  constexpr int TN = kDwTileN, TK = kDwTile, LD = TK + 4;
  using DL = DwLds<NW>;
  constexpr int RW = DL::RW;
  static_assert(GATE == 0 || GATE == 1, "see above");
  static_assert(NW == kDwWaves, "the chunk loop is written for 8 waves x 32 rows");
  float (*part)[TN][LD] = reinterpret_cast<float (*)[TN][LD]>(lds + DL::part);
  float (*bpart)[TN] = reinterpret_cast<float (*)[TN]>(lds + DL::bpart);
  float* sc = lds + DL::sc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's layer, straight from the kernel-argument segment (dynamic index into a
  // by-value array: through the segment pointer it is a scalar load, not a scratch copy)
  const KAT* KA = &A;
  // this workgroup's layer: the first four prefix ends in ONE scalar load (a loop with a load and a wait per
  // item cost two dependent round trips before the first row request); entries past the last item hold the
  // launch's total (fill_dw_kargs), so launches of up to four layers never look further, and the workgroup one
  // past the tiles — the temperature's Adam step riding on this launch (AlphaJob) — is found on the rare path
  const int te0 = KA->tile_end[0], te1 = KA->tile_end[1], te2 = KA->tile_end[2], te3 = KA->tile_end[3];
  // ... and the launch's scalar header with them, pinned into SGPRs here: left to the compiler every field is
  // fetched where it is first used — a scalar load and a wait each, four of them in a row before the first row
  // request and more in the epilogue
  const int hB = A.B, h_n_part = A.n_part, h_tiled = A.dy_tiled, h_row_scale = A.use_row_scale, h_apply = A.apply_only;
  long long* const h_trace = A.trace;
  const float* const h_one = A.one;
  const AdamScalars ad = A.ad;
  asm volatile("" :: "s"(hB), "s"(h_n_part), "s"(h_tiled), "s"(h_row_scale), "s"(h_apply), "s"(h_trace), "s"(h_one),
               "s"(ad.step_dev), "s"(ad.do_adam), "s"(ad.do_polyak), "s"(ad.omb1), "s"(ad.beta2), "s"(ad.omb2), "s"(ad.eps),
               "s"(ad.omtau), "s"(ad.tau), "s"(ad.grad_scale), "s"(ad.step_size_host), "s"(ad.bc2_sqrt_host));
  int item = (bx >= te0 ? 1 : 0) + (bx >= te1 ? 1 : 0) + (bx >= te2 ? 1 : 0) + (bx >= te3 ? 1 : 0);
  if (bx >= te3) {
    if (bx >= KA->tile_end[KAT::kItems - 1]) {
      const AlphaJob& J = A.alpha;
      if constexpr (GATED) {     // (riding beside the backward that READS the temperature: its step waits for that backward's flags)
        const bool ok = dw_gate_wait(A.gate.rows, A.gate.n_rows, A.gate.tag, A.gate.spin);
        if (!ok) report_expired(A.gate.err, A.gate.err_code);
        __syncthreads();
      }
      alpha_step_block(J.log_alpha, J.m, J.v, J.logp, J.B, J.target_entropy, J.lr, J.beta1, J.beta2, J.eps, J.bc1,
                       J.bc2_sqrt, nullptr, nullptr, 1.f);
      return;
    }
#pragma unroll
    for (int j = 4; j + 1 < KAT::kItems; ++j) item += bx >= KA->tile_end[j] ? 1 : 0;   // more than four layers (TQC)
  }
  const DwItem I = KA->items[item];
  const int lt = bx - (item > 0 ? KA->tile_end[item - 1] : 0);
  int n_stamp = 0;
  auto stamp = [&]() {
    const int wg = item * 16 + lt;   // the first 16 tiles of each item
    if (kTraceOn && h_trace != nullptr && tid == 0 && lt < 16 && wg < 64 && n_stamp < kTraceStamps) {
      long long* tr = h_trace + ((size_t)wg * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();
  if (tid == 64) adam_bias_corr(ad, &sc[0], &sc[1]);
  const int tn = lt / I.tiles_k, tk = lt - tn * I.tiles_k;
  const int TNi = I.tile_n;                               // 16, or 8 (partial-sum layers)
  const int n_base = tn * TNi, k_base = tk * TK;
  const int ptile = n_base >> 4, n_off = n_base & 15;     // 16-row pack tile and our offset in it
  const int i = lane & 15, c = lane >> 4;
  const int NSk = cdiv(I.K, 16), NSn = cdiv(I.N, 16);
  const int npart = I.dY_part_stride > 0 ? h_n_part : 1;
  const bool tiled = I.dY_part_stride > 0 && h_tiled != 0;   // tile-major partial buffers (tp4_store_dz1)
  const bool polyak = ad.do_polyak && I.w_t != nullptr;

  // this thread's element of the epilogue; its Adam state is requested NOW so the round
  // trip overlaps the GEMM
  const int nl = tid >> 5, kl = tid & 31;
  const int en = n_base + nl, ek = k_base + kl;
  const bool e_ok = nl < TNi && en < I.N && ek < I.K;
  const size_t eo = (size_t)en * I.K + ek;
  float p_th = 0.f, p_m = 0.f, p_v = 0.f, p_tt = 0.f;
  if (e_ok && ad.do_adam) {
    p_th = I.w[eo];
    p_m = I.w_m[eo];
    p_v = I.w_v[eo];
    if (polyak) p_tt = I.w_t[eo];
  }
  // ... and the bias element of the tiles that own one (tk == 0, thread = column): requested here as
  // well — fetched in the epilogue it was a cold round trip (0.6 us) at the very end of the layer-0
  // tiles, the ones every launch waits for
  const bool b_own = tk == 0 && tid < TNi && n_base + tid < I.N;
  const bool b_pol = ad.do_polyak && I.b_t != nullptr;
  float q_th = 0.f, q_m = 0.f, q_v = 0.f, q_tt = 0.f;
  if (b_own && ad.do_adam) {
    const int n = n_base + tid;
    q_th = I.b[n];
    q_m = I.b_m[n];
    q_v = I.b_v[n];
    if (b_pol) q_tt = I.b_t[n];
  }

  // ---- dW tile = sum_b dY[b, n]^T X[b, k].  Each wave owns 32 consecutive minibatch rows per
  // 256-row chunk.  Rows are fetched with 16-byte loads (16 rows of dY / 8 rows of X per
  // instruction; the 4-byte, 4-rows-per-instruction version was bound by the NUMBER of load
  // instructions: 48 per lane on a layer with four dz1 partials), partials are summed and the
  // per-row seed applied in registers, then the wave stages its rows in wave-private LDS and
  // reads them back in MFMA layout: lane (c, i) feeds dY[row 4u + c][n_base + i] as the A
  // operand and X[row 4u + c][k_base + 2i + {0,1}] as two B operands.
  float (*stA)[RW][TN] = reinterpret_cast<float (*)[RW][TN]>(lds + DL::stA);
  float (*stX)[RW][LD] = reinterpret_cast<float (*)[RW][LD]>(lds + DL::stX);
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float sA = 0.f;
  // per-row seed of unit-seed layers, or the constant 1 (stride 0): always a load, no branch
  // between the row requests
  const bool scaled = I.scaled != 0 && h_row_scale != 0;
  const float* rsp = scaled ? I.rs : h_one;
  const size_t rs_ld = scaled ? (size_t)I.rs_ld : 0;
  const int ar = lane >> 2, an = (lane & 3) * 4;      // dY: 16 rows x 4 lanes x float4
  const int xr = lane >> 3, xk = (lane & 7) * 4;      // X :  8 rows x 8 lanes x float4
  const bool an_ok = an < TNi && n_base + an < I.ldy;  // ldy, ldx are multiples of 4
  const bool xk_ok = k_base + xk < I.ldx;
  // GATED: dY of a layer whose rows ARE the seeds (the scalar critic's output layer, written by the role that
  // publishes the seeds) waits for the second gate like the seeds themselves
  const bool dy_late = GATED && (I.dY == A.gate.late_dY || (A.gate.late_dY2 != nullptr && I.dY == A.gate.late_dY2));
  const unsigned long long* const gseed = (GATED && item >= A.gate.item_split) ? A.gate.seed2 : A.gate.seed;   // (twin critics: the second one's seeds)
  if constexpr (GATED) {
    const bool ok = dw_gate_wait(A.gate.rows, A.gate.n_rows, A.gate.tag, A.gate.spin);
    if (!ok) report_expired(A.gate.err, A.gate.err_code);
    __syncthreads();
  }
  for (int chunk = 0; chunk * 256 < (h_apply ? 0 : hB); ++chunk) {
    const int base = chunk * 256 + 32 * wave;
    f32x4 va[2][4], vx[4];
    float rs[2];

    auto load_dy = [&](int h) {
      const int bb = base + ar + 16 * h;
#pragma unroll
      for (int m = 0; m < 4; ++m) va[h][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bb < hB && an_ok) {
        const size_t off = tiled ? ((size_t)((n_base + an) >> 4) * hB + bb) * 16 + ((n_base + an) & 15)
                                 : (size_t)bb * I.ldy + n_base + an;
        const float* src = I.dY + off;
        va[h][0] = GATED ? ld4_agent(I.dY, (unsigned)off) : ld4(src);
        // tensor-parallel slices leave the first layer's dz as n_part (<= 4) partial buffers
        // (csrc/tp3.h): all requested up front, summed below in member order
#pragma unroll
        for (int m = 1; m < 4; ++m)
          if (m < npart)
            va[h][m] = GATED ? ld4_agent(I.dY, (unsigned)(off + (size_t)m * I.dY_part_stride)) : ld4(src + (size_t)m * I.dY_part_stride);
      }
    };
    auto load_rs = [&](int h) {
      const int bb = base + ar + 16 * h;
      const float* q = rsp + (size_t)(bb < hB ? bb : 0) * rs_ld;
      rs[h] = (GATED && scaled) ? ld1_sc1(q) : *q;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (!GATED) load_rs(h);
      if (!dy_late) load_dy(h);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bb = base + xr + 8 * j;
      vx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bb < hB && xk_ok) {
        const size_t off = (size_t)bb * I.ldx + k_base + xk;
        vx[j] = GATED ? ld4_agent(I.X, (unsigned)off) : ld4(I.X + off);
      }
    }
    if constexpr (GATED) {
      // everything that does not depend on the seeds is under way; the seeds themselves arrive as {tag, value}
      // granules, one per minibatch row (the value is its own flag: the producer neither waits nor flags): every
      // lane polls the granules of its two rows.  They are the row scale of the unit-seed layers and the dY rows
      // of the output layer (`late_dY`, one column).
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bb = base + ar + 16 * h;
        float sd = 1.f;
        if (bb < hB && A.gate.n_seed > 0) {
          unsigned long long g = 0;
          bool ok = false;
          for (int spin = 0; spin < A.gate.spin && !ok; ++spin) {
            g = __hip_atomic_load(gseed + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (unsigned)(g >> 32) == A.gate.tag;
            if (!ok) __builtin_amdgcn_s_sleep(1);
          }
          if (!ok) report_expired(A.gate.err, A.gate.err_code);
          sd = ok ? __uint_as_float((unsigned)g) : __builtin_nanf("");
        }
        rs[h] = scaled ? sd : 1.f;
        if (dy_late) {
#pragma unroll
          for (int m = 0; m < 4; ++m) va[h][m] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (n_base + an == 0 && bb < hB) va[h][0][0] = sd;   // (rows past the minibatch stay zero)
        }
      }
    }
    stamp();   // rows requested
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = (((va[h][0] + va[h][1]) + va[h][2]) + va[h][3]) * rs[h];   // rs = 1 unless unit-seed rows
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = (n_base + an + t < I.N) ? v[t] : 0.f;
      *reinterpret_cast<f32x4*>(&stA[wave][ar + 16 * h][an]) = v;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = vx[j];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = (k_base + xk + t < I.K) ? v[t] : 0.f;
      *reinterpret_cast<f32x4*>(&stX[wave][xr + 8 * j][xk]) = v;
    }
    // wave-private staging: the wave's own LDS writes are ordered before its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float av = stA[wave][4 * u + c][i];
      const f32x2 xv = *reinterpret_cast<const f32x2*>(&stX[wave][4 * u + c][2 * i]);
      sA += av;
      acc[0] = mfma4(av, xv[0], acc[0]);
      acc[1] = mfma4(av, xv[1], acc[1]);
    }
    __builtin_amdgcn_wave_barrier();   // next chunk overwrites the staging rows
  }
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][c * 4 + r][2 * i + w] = acc[w][r];
  {
    float sb = sA;
    sb += __shfl_xor(sb, 16);
    sb += __shfl_xor(sb, 32);
    if (c == 0) bpart[wave][i] = sb;
  }
  stamp();   // MFMA done, partial tiles in LDS
  __syncthreads();
  stamp();
  const float step_size = sc[0], bc2_sqrt = sc[1];
  // ---- epilogue: one element per thread; the updated tile then goes through LDS so that the
  // packs are written IN PACK ORDER as 16-byte stores
  float g = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) g += part[w][nl][kl];
  float gb_x = 0.f;
  if constexpr (XCHG) {
    const DwXchg& X = A.xchg;
    float gbw = 0.f;
    if (tid < TN) {
#pragma unroll
      for (int w = 0; w < NW; ++w) gbw += bpart[w][tid];
    }
    // 8-byte {sequence, value} granules written through at system scope: the value is its own flag, no
    // fences (two system fences per workgroup cost 50 us per launch), the wait is per element.
    // Two hops per tile instead of an all-to-all: the tile's OWNER (tile % world) collects the other
    // ranks' partial tiles, sums them in rank order and sends the sum back — 2 x (world - 1) / world of
    // the arena leaves every GPU instead of (world - 1) x, and all replicas apply the very same sum.
    const unsigned tag = (unsigned)X.seq;
    const int owner = (int)(bx % (unsigned)X.world);
    auto slot = [&](char* base, int src_slot) {
      return reinterpret_cast<unsigned long long*>(base) +
             (((size_t)X.parity * (X.world + 1) + src_slot) * X.max_tiles + bx) * kDwXchgTile;
    };
    auto put = [&](unsigned long long* dst, float v, float vb) {
      __hip_atomic_store(dst + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tid < TN)
        __hip_atomic_store(dst + 512 + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(vb),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto get = [&](const unsigned long long* src, float* v, float* vb) {
      unsigned long long x = 0, xb = 0;
      bool ok = false;
      for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
        x = __hip_atomic_load(src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        xb = tid < TN ? __hip_atomic_load(src + 512 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : x;
        ok = (unsigned)(x >> 32) == tag && (unsigned)(xb >> 32) == tag;
        if (!ok) __builtin_amdgcn_s_sleep(2);
      }
      *v = __uint_as_float((unsigned)x);
      *vb = __uint_as_float((unsigned)xb);
      return ok;
    };
    float gs = 0.f, gbs = 0.f;
    bool all_ok = true;
    if (X.rank == owner) {
      for (int r = 0; r < X.world; ++r) {    // rank order
        float v = g, vb = gbw;
        if (r != X.rank) all_ok = get(slot(X.window, r), &v, &vb) && all_ok;
        gs += v;
        if (tid < TN) gbs += vb;
      }
      if (!all_ok) { gs = __builtin_nanf(""); gbs = gs; }   // (the poison travels to every replica)
      for (int p = 0; p < X.world; ++p)
        if (p != X.rank) put(slot(X.peer[p], X.world), gs, gbs);
    } else {
      put(slot(X.peer[owner], X.rank), g, gbw);
      all_ok = get(slot(X.window, X.world), &gs, &gbs);
    }
    if (!all_ok) {   // bounded wait: a lost rank is reported and poisons the tile instead of hanging
      report_expired(X.err, (KERN_DW_XCHG << 8) | SITE_DW_TILE);
      gs = __builtin_nanf(""); gbs = gs;
    }
    g = gs;
    gb_x = gbs;
  }
  if (h_apply) g = e_ok ? I.w_g[eo] : 0.f;   // the (all-reduced) gradient instead of this rank's GEMM
  g *= ad.grad_scale;
  float th_new = 0.f, tt_new = 0.f;
  if (e_ok) {
    if (I.w_g != nullptr && !h_apply) I.w_g[eo] = g;
    if (ad.do_adam) {
      float mm = p_m, vv = p_v, th = p_th;
      adam_elem(g, mm, vv, th, ad, step_size, bc2_sqrt);
      I.w_m[eo] = mm;
      I.w_v[eo] = vv;
      I.w[eo] = th;
      th_new = th;
      if (polyak) {
        tt_new = polyak_elem(p_tt, th, ad);
        I.w_t[eo] = tt_new;
      }
    }
  }
  if (ad.do_adam && (I.pf != nullptr || I.pf16 != nullptr)) {
    // keep the fragment-order packs in step with the master: stage the new 16x32 tile(s)
    // (zero outside the matrix, like the packs' padding), then 3 x 128 float4 jobs
    // (I.pf == nullptr: a PrecX2 learner's fused update — its kernels read the two-plane fp16 packs only, the fp32
    // packs are rebuilt from the master when somebody asks for them, learner.hip fresh32)
    float (*tileW)[LD] = part[0];
    float (*tileT)[LD] = part[1];
    __syncthreads();               // every thread has read its partial sums
    if (nl < TNi) {
      tileW[n_off + nl][kl] = th_new;
      tileT[n_off + nl][kl] = tt_new;
    }
    __syncthreads();
    dw_write_packs(I, tileW, tileT, tid, tk, n_base, n_off, ptile, TNi, NSk, NSn, polyak);
  }
  if (b_own) {
    const int n = n_base + tid;
    float gb = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) gb += bpart[w][tid];
    if constexpr (XCHG) gb = gb_x;
    if (h_apply) gb = I.b_g[n];
    gb *= ad.grad_scale;                 // (the arithmetic of adam_polyak_elem, on the prefetched state)
    if (I.b_g != nullptr && !h_apply) I.b_g[n] = gb;
    if (ad.do_adam) {
      float mm = q_m, vv = q_v, th = q_th;
      adam_elem(gb, mm, vv, th, ad, step_size, bc2_sqrt);
      I.b_m[n] = mm;
      I.b_v[n] = vv;
      I.b[n] = th;
      if (b_pol) I.b_t[n] = polyak_elem(q_tt, th, ad);
    }
  }
  stamp();   // stores issued
}


}  // namespace oprl
