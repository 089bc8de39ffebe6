// kernels.hip — gfx950 kernels of the off-policy learner hot path.
//
//   k_mlp_slice<WIDTH>   one workgroup per 16-row minibatch slice: input assembly
//                        -> whole-MLP forward -> policy-head / TD epilogue ->
//                        loss-gradient seed -> whole-MLP backward (engine.h)
//   k_dw_adam            dW = dY^T X on 16x32 tiles (fp32 MFMA) fused with
//                        torch-semantics Adam, Polyak target update, grad export
//   k_adam_flat / k_polyak_flat / k_alpha_step / k_reduce_partials   small fused
//                        elementwise pieces
//   k_tqc_target         row-wise bitonic sort of 125 quantiles + truncation + TD
//
// Reference op groups replaced: SURVEY.md §2.2 K2-K13.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "kernels.h"
#include "philox.h"
#include "slice_head.h"
#include "tp3.h"

namespace oprl {

template <int WIDTH>
__device__ __forceinline__ void mlp_slice_body(const MlpArgs& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LY = SliceLds<WIDTH>;
  constexpr int WL = lds_ld(WIDTH);
  const int L = A.net.n_layers, nh = L - 1;
  const int Nout = pick(A.net.dims, L);
  float* x0s = smem;
  float* hb = smem + LY::h_off;
  float* outS = smem + LY::out_off(nh);
  float* auxS = smem + LY::aux_off(nh);
  float* scr = smem + LY::scr_off(nh);
  const int row0 = blockIdx.x * kR;
  const int B = A.B;
  const int tid = threadIdx.x;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (A.trace != nullptr && tid == 0 && n_stamp < kTraceStamps) {
      long long* t = A.trace + ((size_t)blockIdx.x * kTraceStamps + n_stamp) * 2;
      t[0] = (long long)__builtin_readcyclecounter();
      t[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();  // 0: entry

  if (A.do_fwd) {
    lds_zero(x0s, kR * kX0Ld);
    __syncthreads();
    load_rows(x0s, kX0Ld, 0, A.x0, A.k0, A.k0, row0, B);
    if (A.x1 != nullptr) load_rows(x0s, kX0Ld, A.k0, A.x1, A.k1, A.k1, row0, B);
    stamp();  // 1: inputs in LDS
    mlp_forward_slice<WIDTH>(A.net, x0s, hb, outS, scr, A.Xg, A.Xg[1] != nullptr, row0, B, stamp);
    if (A.Xg[0] != nullptr) store_rows(x0s, kX0Ld, A.Xg[0], A.ldx0, A.net.dims[0], row0, B);
    stamp();  // after narrow output layer
    slice_head(A, outS, Nout, row0, true);
  } else if (A.do_bwd) {
    _Pragma("unroll") for (int l = 1; l < kMaxLayers; ++l)
      if (l < L) load_rows4(hb + (l - 1) * LY::hbuf, WL, A.Xg[l], WIDTH, WIDTH, row0, B);
  }
  stamp();  // head / reload done
  if (!A.do_bwd) return;

  slice_seed(A, outS, auxS, scr, Nout, L, row0, blockIdx.x, true);
  stamp();  // seed done
  mlp_backward_slice<WIDTH>(A.net, auxS, hb, scr, A.dYg, row0, B, A.dact_col0, A.dact_cols, auxS, stamp);
  if (A.dact_cols > 0 && A.dact != nullptr) store_rows(auxS, kOutLd, A.dact, A.lddact, A.dact_cols, row0, B);
  stamp();  // end
}

template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_mlp_slice(const MlpArgs A) { mlp_slice_body<WIDTH>(A); }

// Up to kMaxMulti independent nets of one learner (TQC's five quantile critics) in ONE launch,
// grid (slices, nets): side streams gave them only ~1.7x overlap (5 streams share 4 hardware
// queues, every fork/join is an event round trip).  The argument blocks travel by value; a
// workgroup addresses its own through the kernarg segment pointer (scalar loads, as k_dw_adam does).
template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_mlp_slice_multi(const MlpMultiArgs M) {
  const MlpMultiArgs* kp = (const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  mlp_slice_body<WIDTH>(kp->a[blockIdx.y]);
}

template __global__ void k_mlp_slice<256>(const MlpArgs);
template __global__ void k_mlp_slice<512>(const MlpArgs);
template __global__ void k_mlp_slice_multi<256>(const MlpMultiArgs);
template __global__ void k_mlp_slice_multi<512>(const MlpMultiArgs);

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
__device__ __forceinline__ void alpha_step_block(double* log_alpha, double* m, double* v, const float* logp, int B,
                                                 float target_entropy, double lr, double beta1, double beta2, double eps,
                                                 double bc1, double bc2_sqrt, double* grad_out, const double* grad_in,
                                                 float grad_scale);

template <bool XCHG>
__device__ __forceinline__ void dw_adam_body(const DwKArgs& A) {
  constexpr int TN = kDwTileN, TK = kDwTile, LD = TK + 4;
  __shared__ __attribute__((aligned(16))) float part[kDwWaves][TN][LD];
  __shared__ float bpart[kDwWaves][TN];
  __shared__ float sc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this workgroup's layer, straight from the kernel-argument segment (dynamic index into a
  // by-value array: through the segment pointer it is a scalar load, not a scratch copy)
  const DwKArgs* KA = &A;
  // this workgroup's layer: the first four prefix ends in ONE scalar load (a loop with a load and a wait per
  // item cost two dependent round trips before the first row request); entries past the last item hold the
  // launch's total (fill_dw_kargs), so launches of up to four layers never look further, and the workgroup one
  // past the tiles — the temperature's Adam step riding on this launch (AlphaJob) — is found on the rare path
  const int bx = (int)blockIdx.x;
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
    if (bx >= KA->tile_end[kDwMaxItems - 1]) {
      const AlphaJob& J = A.alpha;
      alpha_step_block(J.log_alpha, J.m, J.v, J.logp, J.B, J.target_entropy, J.lr, J.beta1, J.beta2, J.eps, J.bc1,
                       J.bc2_sqrt, nullptr, nullptr, 1.f);
      return;
    }
#pragma unroll
    for (int j = 4; j + 1 < kDwMaxItems; ++j) item += bx >= KA->tile_end[j] ? 1 : 0;   // more than four layers (TQC)
  }
  const DwItem I = KA->items[item];
  const int lt = (int)blockIdx.x - (item > 0 ? KA->tile_end[item - 1] : 0);
  int n_stamp = 0;
  auto stamp = [&]() {
    const int wg = item * 16 + lt;   // the first 16 tiles of each item
    if (h_trace != nullptr && tid == 0 && lt < 16 && wg < 64 && n_stamp < kTraceStamps) {
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
  __shared__ __attribute__((aligned(16))) float stA[kDwWaves][32][TN];
  __shared__ __attribute__((aligned(16))) float stX[kDwWaves][32][LD];
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
  for (int chunk = 0; chunk * 256 < (h_apply ? 0 : hB); ++chunk) {
    const int base = chunk * 256 + 32 * wave;
    f32x4 va[2][4], vx[4];
    float rs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int bb = base + ar + 16 * h;
      rs[h] = rsp[(size_t)(bb < hB ? bb : 0) * rs_ld];
#pragma unroll
      for (int m = 0; m < 4; ++m) va[h][m] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bb < hB && an_ok) {
        const float* src = tiled ? I.dY + ((size_t)((n_base + an) >> 4) * hB + bb) * 16 + ((n_base + an) & 15)
                                 : I.dY + (size_t)bb * I.ldy + n_base + an;
        va[h][0] = ld4(src);
        // tensor-parallel slices leave the first layer's dz as n_part (<= 4) partial buffers
        // (csrc/tp3.h): all requested up front, summed below in member order
#pragma unroll
        for (int m = 1; m < 4; ++m)
          if (m < npart) va[h][m] = ld4(src + (size_t)m * I.dY_part_stride);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int bb = base + xr + 8 * j;
      vx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (bb < hB && xk_ok) vx[j] = ld4(I.X + (size_t)bb * I.ldx + k_base + xk);
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
  for (int w = 0; w < kDwWaves; ++w) g += part[w][nl][kl];
  float gb_x = 0.f;
  if constexpr (XCHG) {
    const DwXchg& X = A.xchg;
    float gbw = 0.f;
    if (tid < TN) {
#pragma unroll
      for (int w = 0; w < kDwWaves; ++w) gbw += bpart[w][tid];
    }
    // 8-byte {sequence, value} granules written through at system scope: the value is its own flag, no
    // fences (two system fences per workgroup cost 50 us per launch), the wait is per element.
    // Two hops per tile instead of an all-to-all: the tile's OWNER (tile % world) collects the other
    // ranks' partial tiles, sums them in rank order and sends the sum back — 2 x (world - 1) / world of
    // the arena leaves every GPU instead of (world - 1) x, and all replicas apply the very same sum.
    const unsigned tag = (unsigned)X.seq;
    const int owner = (int)(blockIdx.x % (unsigned)X.world);
    auto slot = [&](char* base, int src_slot) {
      return reinterpret_cast<unsigned long long*>(base) +
             (((size_t)X.parity * (X.world + 1) + src_slot) * X.max_tiles + blockIdx.x) * kDwXchgTile;
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
      mm = mm + (g - mm) * ad.omb1;
      vv = vv * ad.beta2 + ad.omb2 * g * g;
      th = th - step_size * (mm / (sqrtf(vv) / bc2_sqrt + ad.eps));
      I.w_m[eo] = mm;
      I.w_v[eo] = vv;
      I.w[eo] = th;
      th_new = th;
      if (polyak) {
        tt_new = p_tt * ad.omtau + ad.tau * th;
        I.w_t[eo] = tt_new;
      }
    }
  }
  if (ad.do_adam && I.pf != nullptr) {
    // keep the fragment-order packs in step with the master: stage the new 16x32 tile(s)
    // (zero outside the matrix, like the packs' padding), then 3 x 128 float4 jobs
    float (*tileW)[LD] = part[0];
    float (*tileT)[LD] = part[1];
    __syncthreads();               // every thread has read its partial sums
    if (nl < TNi) {
      tileW[n_off + nl][kl] = th_new;
      tileT[n_off + nl][kl] = tt_new;
    }
    __syncthreads();
    if (tid < 384) {
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
          const float (*src)[LD] = which == 0 ? tileW : tileT;
          *reinterpret_cast<f32x4*>(dst + (((size_t)ptile * NSk + kstep) * 64 + l) * 4) =
              *reinterpret_cast<const f32x4*>(&src[li][16 * blk + 4 * lk]);
        }
      }
    }
  }
  if (ad.do_adam && I.pf16 != nullptr) {
    // ... and the bf16 packs (PrecBF16, engine.h): a bf16 macro step = fp32 steps 2s, 2s + 1 side by
    // side, so this 16 x 32 tile is ONE forward fragment block (64 lanes x 16 B, online and target) and,
    // for W^T, one HALF (8 B per lane) of a block for each of its two 16-row k tiles — the other half
    // belongs to the neighbouring n tile.  Same staged tiles, no further barrier.
    const float (*tileW)[LD] = part[0];
    const float (*tileT)[LD] = part[1];
    const int NSk2 = cdiv(I.K, 32), NSn2 = cdiv(I.N, 32);
    if (tid < 128) {             // W packs: which = online / target
      const int which = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
      float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
      if (dst != nullptr && li >= n_off && li < n_off + TNi) {
        const float (*src)[LD] = which == 0 ? tileW : tileT;
        const bf16x8 v = cvt_bf16x8(*reinterpret_cast<const f32x4*>(&src[li][4 * lk]),
                                    *reinterpret_cast<const f32x4*>(&src[li][16 + 4 * lk]));
        *reinterpret_cast<bf16x8*>(dst + (((size_t)ptile * NSk2 + tk) * 64 + l) * 4) = v;
      }
    } else if (tid < 256 && I.pb16 != nullptr) {   // W^T pack: k tile 2 tk + blk, n step n_base / 32, half (n_base / 16) & 1
      const int q = tid - 128, blk = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
      const int ktile = 2 * tk + blk;
      if (16 * ktile < I.K && 4 * lk >= n_off && 4 * lk < n_off + TNi) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = tileW[4 * lk + t][16 * blk + li];
        const bf16x4 h = __builtin_convertvector(v, bf16x4);
        *reinterpret_cast<bf16x4*>(I.pb16 + (((size_t)ktile * NSn2 + (n_base >> 5)) * 64 + l) * 4 + 2 * ((n_base >> 4) & 1)) = h;
      }
    }
  }
  if (b_own) {
    const int n = n_base + tid;
    float gb = 0.f;
#pragma unroll
    for (int w = 0; w < kDwWaves; ++w) gb += bpart[w][tid];
    if constexpr (XCHG) gb = gb_x;
    if (h_apply) gb = I.b_g[n];
    gb *= ad.grad_scale;                 // (the arithmetic of adam_polyak_elem, on the prefetched state)
    if (I.b_g != nullptr && !h_apply) I.b_g[n] = gb;
    if (ad.do_adam) {
      float mm = q_m, vv = q_v, th = q_th;
      mm = mm + (gb - mm) * ad.omb1;
      vv = vv * ad.beta2 + ad.omb2 * gb * gb;
      const float denom = sqrtf(vv) / bc2_sqrt + ad.eps;
      th = th - step_size * (mm / denom);
      I.b_m[n] = mm;
      I.b_v[n] = vv;
      I.b[n] = th;
      if (b_pol) I.b_t[n] = q_tt * ad.omtau + ad.tau * th;
    }
  }
  stamp();   // stores issued
}

template <bool XCHG>
__global__ __launch_bounds__(kDwThreads) void k_dw_adam(const DwKArgs A) {
  // (through the kernel-argument segment pointer: a dynamic index into the by-value table is then a scalar load)
  dw_adam_body<XCHG>(*(const DwKArgs*)__builtin_amdgcn_kernarg_segment_ptr());
}

// N learners' dW + Adam launches as one (grid.z = learner; argument blocks in device memory)
__global__ __launch_bounds__(kDwThreads) void k_dw_adam_group(const DwKArgs* __restrict__ batch) {
  const DwKArgs& A = batch[blockIdx.z];
  if ((int)blockIdx.x >= A.tile_end[kDwMaxItems - 1]) return;   // (entries past the last item hold the total)
  dw_adam_body<false>(A);
}

// flat Adam over an arena (data-parallel apply after the all-reduce; alpha-free)
__global__ void k_adam_flat(float* th, float* m, float* v, float* tt, const float* g, long n,
                            const AdamScalars ad) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) adam_bias_corr(ad, &sc[0], &sc[1]);
  __syncthreads();
  const float step_size = sc[0], bc2_sqrt = sc[1];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x)
  {
    float t0, t1;
    (void)adam_polyak_elem(g[idx], th + idx, m + idx, v + idx, tt ? tt + idx : nullptr, nullptr, ad,
                           step_size, bc2_sqrt, &t0, &t1);
  }
}

__global__ void k_polyak_flat(float* tt, const float* th, long n, float tau, float omtau) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x)
    tt[idx] = tt[idx] * omtau + tau * th[idx];
}

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

__global__ void k_alpha_step(double* log_alpha, double* m, double* v, const float* logp, int B,
                             float target_entropy, double lr, double beta1, double beta2, double eps,
                             double bc1, double bc2_sqrt, double* grad_out, const double* grad_in,
                             float grad_scale) {
  alpha_step_block(log_alpha, m, v, logp, B, target_entropy, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_out, grad_in,
                   grad_scale);
}

// out[0..3] = sum over slices of partials[.][0..3]  (then scaled by the caller)
__global__ void k_reduce_partials(const float* partials, int n_slices, float* out, int out_off,
                                  float scale_loss, float scale_mean) {
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int i = 0; i < n_slices; ++i) s += partials[i * 4 + threadIdx.x];
    out[out_off + threadIdx.x] = s * (threadIdx.x == 0 ? scale_loss : scale_mean);
  }
}

// out[out_off] = scale * sum of x[0..n)  (one workgroup; diagnostics only)
__global__ void k_sum(const float* x, int n, float* out, int out_off, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[out_off] = scale * (((red[0] + red[1]) + red[2]) + red[3]);
}

hipError_t launch_sum(const float* x, int n, float* out, int out_off, float scale, hipStream_t st) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, x, n, out, out_off, scale);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// TQC target (tqc.py:129-145): per row gather n_nets*Q quantiles of the target
// critics, ascending bitonic sort in LDS (one wave per row, 128-slot network),
// drop the top `drop`, target[b, s] = r + (1-d) gamma (z_sorted[s] - alpha logp').
// z layout: [n_nets][B][ldz];  target: [B][M], M = n_nets*Q - drop.
// ---------------------------------------------------------------------------
constexpr int kTqcWaves = 4;
__global__ __launch_bounds__(64 * kTqcWaves) void k_tqc_target(const float* z, long net_stride, int ldz,
                                                         int n_nets, int Q, int drop,
                                                         const float* r, const float* d,
                                                         const float* logp, const double* log_alpha,
                                                         float gamma, int B, float* target) {
  __shared__ float buf[kTqcWaves][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kTqcWaves + wave;
  const int total = n_nets * Q, M = total - drop;
  float* sb = buf[wave];
  if (row < B) {
    for (int e = lane; e < 128; e += 64) {
      float v = __builtin_huge_valf();
      if (e < total) {
        const int n = e / Q, q = e - n * Q;
        v = z[n * net_stride + (size_t)row * ldz + q];
      }
      sb[e] = v;
    }
  }
  __syncthreads();
  for (int k = 2; k <= 128; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (row < B) {
        const int e = ((lane & ~(j - 1)) << 1) | (lane & (j - 1));  // lower index of the pair
        const int p = e | j;
        const bool up = (e & k) == 0;
        const float a = sb[e], b = sb[p];
        if ((a > b) == up) { sb[e] = b; sb[p] = a; }
      }
      __syncthreads();
    }
  }
  if (row < B) {
    const float alpha = (float)exp(*log_alpha);
    const float al = alpha * logp[row];
    const float coef = (1.f - d[row]) * gamma;
    const float rr = r[row];
    for (int s = lane; s < M; s += 64) target[(size_t)row * M + s] = rr + coef * (sb[s] - al);
  }
}

// the device N(0,1) stream exactly as the update kernels draw it: out[row][col] = philox_normal(seed, ctr, row, col)
__global__ void k_debug_normal(unsigned long long seed, unsigned long long ctr, int rows, int cols, float* out) {
  const long n = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = philox_normal(seed, ctr, (unsigned)(e / cols), (unsigned)(e % cols));
}

hipError_t launch_debug_normal(unsigned long long seed, unsigned long long ctr, int rows, int cols, float* out,
                               hipStream_t st) {
  const long n = (long)rows * cols;
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_debug_normal, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, seed, ctr, rows, cols, out);
  return hipGetLastError();
}

// master (row-major [N][K]) -> fragment-order packs; one 256-thread block per 256
// consecutive elements of a layer.  Pad positions are never written (the pack
// buffers are zeroed once at allocation).
__global__ void k_repack(const RepackItem* items, int n_items) {
  int it = 0;
  while (it + 1 < n_items && (int)blockIdx.x >= items[it].blk_end) ++it;
  const RepackItem I = items[it];
  const long e = (long)(blockIdx.x - I.blk_begin) * 256 + threadIdx.x;
  if (e >= (long)I.N * I.K) return;
  const int n = (int)(e / I.K), k = (int)(e - (long)n * I.K);
  const float w = I.w[e];
  if (I.pf != nullptr) I.pf[pack_index(n, k, cdiv(I.K, 16))] = w;
  if (I.pb != nullptr) I.pb[pack_index(k, n, cdiv(I.N, 16))] = w;
  if (I.pf16 != nullptr) reinterpret_cast<__bf16*>(I.pf16)[pack16_index(n, k, cdiv(I.K, 32))] = (__bf16)w;
  if (I.pb16 != nullptr) reinterpret_cast<__bf16*>(I.pb16)[pack16_index(k, n, cdiv(I.N, 32))] = (__bf16)w;
}

// ---------------------------------------------------------------------------
// host-visible launchers
// ---------------------------------------------------------------------------
size_t mlp_slice_lds_bytes(int width, int n_layers) {
  const int nh = n_layers - 1;
  return sizeof(float) * (width == 256 ? SliceLds<256>::total(nh) : SliceLds<512>::total(nh));
}

hipError_t launch_mlp_slice(const MlpArgs& a, int width, hipStream_t st) {
  const int grid = (a.B + kR - 1) / kR;
  const size_t lds = mlp_slice_lds_bytes(width, a.net.n_layers);
  if (width == 256) {
    hipLaunchKernelGGL(k_mlp_slice<256>, dim3(grid), dim3(kThreads), lds, st, a);
  } else {
    hipLaunchKernelGGL(k_mlp_slice<512>, dim3(grid), dim3(kThreads), lds, st, a);
  }
  return hipGetLastError();
}

// n <= kMaxMulti launches of equal width, depth and batch as one
hipError_t launch_mlp_slice_multi(const MlpArgs* a, int n, int width, hipStream_t st) {
  if (n < 1 || n > kMaxMulti) return hipErrorInvalidValue;
  MlpMultiArgs m;
  for (int j = 0; j < n; ++j) m.a[j] = a[j];
  for (int j = n; j < kMaxMulti; ++j) m.a[j] = a[0];
  const dim3 grid((a[0].B + kR - 1) / kR, n);
  const size_t lds = mlp_slice_lds_bytes(width, a[0].net.n_layers);
  if (width == 256) {
    hipLaunchKernelGGL(k_mlp_slice_multi<256>, grid, dim3(kThreads), lds, st, m);
  } else {
    hipLaunchKernelGGL(k_mlp_slice_multi<512>, grid, dim3(kThreads), lds, st, m);
  }
  return hipGetLastError();
}

hipError_t init_kernel_attrs() {
  const void* ks[4] = {reinterpret_cast<const void*>(&k_mlp_slice<256>), reinterpret_cast<const void*>(&k_mlp_slice<512>),
                       reinterpret_cast<const void*>(&k_mlp_slice_multi<256>),
                       reinterpret_cast<const void*>(&k_mlp_slice_multi<512>)};
  for (const void* k : ks) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

__device__ float g_one = 1.f;

bool dw_wide_item_ok(const DwItem& it, const DwArgs& a);
hipError_t launch_dw_adam_wide(const DwItem* items, int n_items, int B, const AdamScalars& ad, hipStream_t st);

static const float* dw_one_dev() {
  static const float* one_dev = nullptr;
  if (one_dev == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_one)) != hipSuccess) return nullptr;
    one_dev = (const float*)p;
  }
  return one_dev;
}

// the kernel-side argument block of a (non-wide, single-rank) launch; returns the tile count or -1
int fill_dw_kargs(const DwArgs& a, DwKArgs* k) {
  if (a.n_items < 1 || a.n_items > kDwMaxItems || a.xchg != nullptr) return -1;
  int total = 0;
  for (int j = 0; j < a.n_items; ++j) {
    k->items[j] = a.items[j];
    total += a.items[j].tile_end - a.items[j].tile_begin;
    k->tile_end[j] = total;
  }
  for (int j = a.n_items; j < kDwMaxItems; ++j) { k->items[j] = a.items[0]; k->tile_end[j] = total; }
  k->n_items = a.n_items; k->B = a.B; k->n_part = a.n_part; k->dy_tiled = a.dy_tiled; k->ad = a.ad; k->trace = a.trace;
  k->use_row_scale = a.use_row_scale; k->one = dw_one_dev();
  k->apply_only = a.apply_only;
  k->alpha = AlphaJob{};
  memset(&k->xchg, 0, sizeof k->xchg);
  return k->one != nullptr ? total : -1;
}

hipError_t launch_dw_adam_group(const DwKArgs* batch_dev, int n, int tiles, hipStream_t st) {
  hipLaunchKernelGGL(k_dw_adam_group, dim3(tiles, 1, n), dim3(kDwThreads), 0, st, batch_dev);
  return hipGetLastError();
}

hipError_t launch_dw_adam(const DwArgs& a0, hipStream_t st) {
  if (a0.n_items < 1 || a0.n_items > kDwMaxItems) return hipErrorInvalidValue;
  // wide layers (TQC's 512x512) go to the 64x64-tile kernel (csrc/dw_wide.hip), the rest stay here
  static const bool no_wide = [] { const char* e = getenv("OPRL_AMD_NO_DW_WIDE"); return e != nullptr && atoi(e) != 0; }();
  DwItem rest[kDwMaxItems], wide[kDwMaxItems];
  int n_rest = 0, n_wide = 0;
  for (int j = 0; j < a0.n_items; ++j) {
    if (!no_wide && n_wide < 10 && dw_wide_item_ok(a0.items[j], a0)) wide[n_wide++] = a0.items[j];
    else rest[n_rest++] = a0.items[j];
  }
  if (n_wide > 0) {
    hipError_t e = launch_dw_adam_wide(wide, n_wide, a0.B, a0.ad, st);
    if (e != hipSuccess) return e;
    if (n_rest == 0) return a0.alpha.log_alpha != nullptr ? hipErrorInvalidValue : hipSuccess;   // (a job needs a tile launch to ride on)
  }
  DwArgs a = a0;
  a.items = rest;
  a.n_items = n_rest;
  static const float* one_dev = nullptr;
  if (one_dev == nullptr) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_one));
    if (e != hipSuccess) return e;
    one_dev = (const float*)p;
  }
  DwKArgs k;
  int total = 0;
  for (int j = 0; j < a.n_items; ++j) {
    k.items[j] = a.items[j];
    total += a.items[j].tile_end - a.items[j].tile_begin;
    k.tile_end[j] = total;
  }
  for (int j = a.n_items; j < kDwMaxItems; ++j) { k.items[j] = a.items[0]; k.tile_end[j] = total; }
  k.n_items = a.n_items; k.B = a.B; k.n_part = a.n_part; k.dy_tiled = a.dy_tiled; k.ad = a.ad; k.trace = a.trace;
  k.use_row_scale = a.use_row_scale; k.one = one_dev;
  k.apply_only = a.apply_only;
  k.alpha = a.alpha;
  memset(&k.xchg, 0, sizeof k.xchg);
  if (a.xchg != nullptr) {
    if (a.apply_only || n_wide > 0 || total > a.xchg->max_tiles || a.alpha.log_alpha != nullptr) return hipErrorInvalidValue;
    k.xchg = *a.xchg;
    hipLaunchKernelGGL(k_dw_adam<true>, dim3(total), dim3(kDwThreads), 0, st, k);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_dw_adam<false>, dim3(total + (a.alpha.log_alpha != nullptr ? 1 : 0)), dim3(kDwThreads), 0, st, k);
  return hipGetLastError();
}

hipError_t launch_repack(const RepackItem* items_dev, int n_items, int total_blocks, hipStream_t st) {
  if (total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_repack, dim3(total_blocks), dim3(256), 0, st, items_dev, n_items);
  return hipGetLastError();
}

hipError_t launch_adam_flat(float* th, float* m, float* v, float* tt, const float* g, long n,
                            const AdamScalars& ad, hipStream_t st) {
  const int grid = (int)((n + 4 * 256 - 1) / (4 * 256));
  hipLaunchKernelGGL(k_adam_flat, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, th, m, v, tt, g, n, ad);
  return hipGetLastError();
}

hipError_t launch_polyak_flat(float* tt, const float* th, long n, double tau, hipStream_t st) {
  const int grid = (int)((n + 4 * 256 - 1) / (4 * 256));
  hipLaunchKernelGGL(k_polyak_flat, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, tt, th, n, (float)tau,
                     (float)(1.0 - tau));
  return hipGetLastError();
}

hipError_t launch_alpha_step(double* log_alpha, double* m, double* v, const float* logp, int B,
                             float target_entropy, double lr, double beta1, double beta2, double eps,
                             int step, double* grad_out, const double* grad_in, float grad_scale,
                             hipStream_t st) {
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2_sqrt = std::sqrt(1.0 - std::pow(beta2, (double)step));
  hipLaunchKernelGGL(k_alpha_step, dim3(1), dim3(256), 0, st, log_alpha, m, v, logp, B,
                     target_entropy, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_out, grad_in, grad_scale);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const float* partials, int n_slices, float* out, int out_off,
                                  float scale_loss, float scale_mean, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(64), 0, st, partials, n_slices, out, out_off,
                     scale_loss, scale_mean);
  return hipGetLastError();
}

hipError_t launch_tqc_target(const float* z, long net_stride, int ldz, int n_nets, int Q, int drop,
                             const float* r, const float* d, const float* logp,
                             const double* log_alpha, float gamma, int B, float* target,
                             hipStream_t st) {
  hipLaunchKernelGGL(k_tqc_target, dim3((B + kTqcWaves - 1) / kTqcWaves), dim3(64 * kTqcWaves), 0, st, z,
                     net_stride, ldz, n_nets, Q, drop, r, d, logp, log_alpha, gamma, B, target);
  return hipGetLastError();
}

}  // namespace oprl
