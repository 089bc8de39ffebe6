// layerwise.hip — wide nets (TQC's 512-wide, 4-layer quantile critics) layer by layer.
//
// k_mlp_slice carries a 16-row slice through a whole MLP on ONE CU.  For a 512x512 layer that is
// 4096 fp32 MFMAs = 13.7 us at the CU's matrix peak, twice per forward, and five critics x 16
// slices occupy 80 of the 256 CUs (profiles/r01g_kernel_stats_tqc.csv: 50 / 122 / 97 us per
// launch).  The activations between layers are [B, 512] rows — 0.5 MB per net — so here the
// hand-over between layers is a kernel boundary (2.1 us) instead of a cluster exchange: every
// launch covers ONE layer of all nets with a workgroup per (slice, 128 output columns, net), 320
// workgroups for TQC at B = 256 — two per CU, all resident at once — each a [16 x K] x [K x 128]
// product through the same packs and the same gemm_packed() as the slice kernels (two waves per
// 16-column tile split the contraction).
// The hidden activations and pre-activation gradients travel through the very buffers k_dw_adam
// reads afterwards (NetWs::X / dY), so nothing is stored twice.
//
//   k_lw_in      [x0 | x1] -> h1 = relu(W0 x + b0)                        slices x W/128 x nets workgroups
//   k_lw_mid     forward:  h[l+1] = relu(W_l h[l] + b_l)                  (1-D, XCD-aware: lw_who())
//                backward: dz[l-1] = (dz[l] W_l) * [h[l] > 0]
//   k_lw_head    out = W_L h + b_L -> head -> loss seed -> dz[L-2]        grid (slices, 1, nets)
//   k_lw_dact    gradient wrt the action columns of the input             grid (slices, 1, nets)
//
// Heads and seeds are those of k_mlp_slice (slice_head.h).  Summation order differs from the
// single-CU kernel (the contraction is split over two waves), so the two agree to rounding, not
// bit for bit (tests/test_gpu_fused.py::test_tqc_layerwise_equals_slice_kernel).
// Reference ops replaced: the addmm / threshold_backward chains of algos/nn_models.py:84-107 under
// autograd, for tqc.py:128-177.
#include <cstdlib>
#include <cstring>
#include "kernels.h"
#include "slice_head.h"
#include "tp4.h"
#include "slice_tp_body.h"
#include "batch_rows.h"
#include "dw_body.h"

namespace oprl {

constexpr int kLwCols = 128;                 // output columns per workgroup: 8 tiles x 2 waves
constexpr int kLwTiles = kLwCols / 16;

template <int WIDTH>
struct LwLds {   // floats
  static constexpr int WL = lds_ld(WIDTH);
  static constexpr int x = 0;                              // [kR][WL] input rows of the layer
  static constexpr int x0 = x + kR * WL;                   // [kR][kX0Ld] net input (k_lw_in)
  static constexpr int out = x0 + kR * kX0Ld;
  static constexpr int aux = out + kR * kOutLd;
  static constexpr int scr = aux + kR * kOutLd;
  static constexpr int total = scr + kWaves * kR * 16;
};

// Wide launches are 1-D and XCD-aware.  Every launch starts with cold L2s (eight of them, one per
// XCD, filled over the fabric), workgroup i runs on XCD i % 8, and the bytes that matter are the
// 256 KB weight slab of a (net, column group) pair — shared by that pair's slices — and the
// [16 x K] input rows of a (net, slice).  A UNIT is a pair's slices, or 1/nsplit of them (nsplit is
// the smallest power of two that makes the unit count a multiple of 8, so the XCDs carry equal
// loads: TQC has 20 pairs -> 40 half-units, five per XCD); XCD x owns the consecutive units
// [x * upx, (x + 1) * upx) in net-major order, so a slab crosses the fabric once (twice) and an XCD
// reads the rows of at most two nets.
struct LwWho { int slice, cg, net; bool ok; };
struct LwGrid { int slices, nets, nsplit, spu, upx; };   // spu: slices per unit, upx: units per XCD
__device__ __forceinline__ LwWho lw_who(const LwGrid& G, int ncg) {
  const int units = G.nets * ncg * G.nsplit;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int ul = j / G.spu, u = xcd * G.upx + ul;
  const int p = u / G.nsplit, part = u - p * G.nsplit;
  LwWho w;
  w.slice = part * G.spu + (j - ul * G.spu);
  w.net = p / ncg;
  w.cg = p - w.net * ncg;
  w.ok = u < units && w.slice < G.slices;
  return w;
}
inline LwGrid lw_grid(int slices, int ncg, int nets) {
  LwGrid g;
  g.slices = slices; g.nets = nets; g.nsplit = 1;
  while (g.nsplit < 8 && (nets * ncg * g.nsplit) % 8 != 0 && g.nsplit * 2 <= slices) g.nsplit *= 2;
  g.spu = (slices + g.nsplit - 1) / g.nsplit;
  g.upx = (nets * ncg * g.nsplit + 7) / 8;
  return g;
}
inline int lw_blocks(const LwGrid& g) { return 8 * g.upx * g.spu; }

__host__ __device__ constexpr size_t lw_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ const MlpArgs& lw_args(int net) {
  const MlpMultiArgs* kp = (const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  return kp->a[net];
}

// this thread's two elements of a [kR x 128] result in gemm_packed()'s split-contraction epilogue:
// (row, col) and (row, col + 64)
__device__ __forceinline__ void lw_elem(int* row, int* col) {
  const int rt = threadIdx.x >> 8, rrem = threadIdx.x & 255;
  *row = rrem >> 4;
  *col = 16 * rt + (rrem & 15);
}

// (launch bounds: 8 waves per SIMD = two workgroups per CU, i.e. at most 64 VGPRs — with one
// workgroup per CU a TQC layer launch was 2.5 rounds of workgroups, 15.8 us)
template <int WIDTH>
__global__ __launch_bounds__(kThreads, 8) void k_lw_in(const MlpMultiArgs M, const LwGrid G) {
  __shared__ __attribute__((aligned(16))) float smem[kR * kX0Ld + kWaves * kR * 16];
  const LwWho w = lw_who(G, WIDTH / kLwCols);
  if (!w.ok) return;
  const MlpArgs& A = lw_args(w.net);
  float* x0s = smem;
  float* scr = smem + kR * kX0Ld;
  const int row0 = w.slice * kR, cg = w.cg, B = A.B;
  const int K0 = A.net.dims[0], NS0 = cdiv(K0, 16);
  lds_zero(x0s, kR * kX0Ld);
  __syncthreads();
  load_rows(x0s, kX0Ld, 0, A.x0, A.k0, A.k0, row0, B);
  if (A.x1 != nullptr) load_rows(x0s, kX0Ld, A.k0, A.x1, A.k1, A.k1, row0, B);
  float* Y = A.Xg[1];
  gemm_packed(x0s, kX0Ld, A.net.pf[0] + (size_t)cg * kLwTiles * NS0 * 256, kLwTiles, NS0, scr,
              A.net.b[0] + cg * kLwCols, kLwCols, [&](int row, int col, float v) {
                if (row0 + row < B) Y[(size_t)(row0 + row) * WIDTH + cg * kLwCols + col] = fmaxf(v, 0.f);
              });
  if (cg == 0 && A.Xg[0] != nullptr) store_rows(x0s, kX0Ld, A.Xg[0], A.ldx0, K0, row0, B);
}

// ---- hidden layers ------------------------------------------------------------------------------------
// (Two earlier cuts of this work — equal (slice, 128-column, net) workgroups, and 16-row runs of ten tiles across nets —
// were retired in round 3: profiles/r01*, r02_experiments.txt hold their measurements.)
// MODE 0: forward layer l, 1: backward through layer l, 2: forward layer 1 with the FIRST layer fused in —
// for a narrow net input (<= 32 columns: two macro steps) every workgroup recomputes the slice's
// h1 = relu(W0 [x0 | x1] + b0) tile(s) in LDS instead of reading them (8 MFMAs per wave; the run
// that starts a net also stores h1 and the input rows for k_dw_adam), which saves the k_lw_in launch
// P (engine.h): the precision of the hidden-layer GEMM — PrecBF16 reads layer l's pack as bf16 fragments
// (the launch's nets carry bf16 pointers for their hidden layers then; the folded first layer, the heads
// and k_lw_dact stay fp32: together 3 % of a TQC critic's FLOPs).

// ---- 32-row runs -------------------------------------------------------------------------------------
// A CU takes data in at ≈45 KB/us whatever the request order, and k_lw_mid_run's 16 rows x 10
// tiles are 32 KB of rows + 320 KB of weight fragments per workgroup: 8 us before the last MFMA can issue.
// The same ten 16x16 output tiles cut as 32 rows x 5 column tiles are 64 + 160 KB.  Here a workgroup owns
// 32 rows x a run of 5 or 6 column tiles of ONE net (32 tiles = 4 runs of 5 + 2 of 6: no run straddles two
// nets, 5 nets x 6 runs x 8 row slices = 240 workgroups for TQC at B = 256).  Wave w contracts K-eighth
// (w & 7) of every second tile of the run (parity w >> 3) for BOTH row tiles with the same B fragments
// (a fragment is requested by exactly one wave); the partial tiles meet in LDS — over the input rows, which
// are dead by then — and are summed in K order.
constexpr int kLw2Rows = 32;
constexpr int kLw2MaxRun = 6;
struct LwRun2 { int slices, nets, rpn, base, rem, ppx; };   // runs per net; tiles per run = base (+1 for the first rem runs); pairs per XCD

// (the body as a function of the workgroup index bx: the launch below, and the riding form k_slice_tp_fin)
// KM: the nets' arguments in the kernel-argument segment; net0: R's nets are KM->a[net0 ..]
// ROLE (k_lw_mid_pair: two consecutive layers in ONE launch): 1 = the first layer's workgroup — its result rows leave
// written through (sc1) and, once every wave's stores are acknowledged, flag X.flags[(net, slice, run)] = {tag, *} goes
// up; 2 = the second layer's — weight fragments, bias / masks first, then a bounded wait for the rpn flags of its
// (net, slice), then the rows with sc1 loads.  MAXRUN: the longest run of column tiles (the partial tiles' LDS).
struct LwPair { unsigned long long* flags; unsigned tag; int spin; unsigned* err; int wait_net0; };   // wait_net0: ROLE 2 waits for the flags of nets >= wait_net0 only (the others' rows come from an earlier launch)
#ifdef LW_TRACE            // tools/ubench_lw.hip: stage stamps of the hidden-layer workgroups (100 MHz wall clock)
__device__ unsigned long long g_lw_trace[4096 * 8];
#define LW_STAMP(k) do { if (threadIdx.x == 0 && bx < 4096) g_lw_trace[bx * 8 + (k)] = wall_clock64(); } while (0)
#else
#define LW_STAMP(k) do { } while (0)
#endif
constexpr int kLwFlagStride = 32;             // flags per (net, slice): rpn <= 32
template <int MODE, class P, int ROLE = 0, int MAXRUN = 6>
__device__ __forceinline__ void lw_mid_run2_body(const MlpMultiArgs* KM, int l, const LwRun2& R, int bx, int net0,
                                                 const LwPair* X = nullptr) {
  static_assert(!(MODE == 2 && ROLE == 2), "the first hidden layer has no producer in the launch");
  constexpr bool BWD = MODE == 1, FIN = MODE == 2;
  constexpr int NSE = 64 / P::KS;               // macro steps of a K-eighth (64 columns): 4 / 2
  constexpr int NSW = 512 / P::KS;
  constexpr int WIDTH = 512, WL = lds_ld(WIDTH), NTW = WIDTH / 16;
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* xs = dsm;                              // [32][WL] input rows of the layer
  float* xin = dsm + kLw2Rows * WL;             // FIN: [32][kX0Ld] net input rows
  float* scr = dsm;                             // after the MFMAs: [row tile][run tile][K-eighth][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kk = lane >> 4;
  // XCD x owns the (net, run) pairs [x * ppx, (x + 1) * ppx) with all their row slices: a run's weights
  // cross the fabric once and an XCD reads the rows of at most two nets
  const int xcd = bx & 7, j = bx >> 3;
  const int pl = j / R.slices, pair = xcd * R.ppx + pl, slice = j - pl * R.slices;
  if (pl >= R.ppx || pair >= R.nets * R.rpn) return;      // (pl >= ppx: a riding grid rounded up past the runs)
  const int net = pair / R.rpn, run = pair - net * R.rpn;
  const int t0 = run * R.base + min(run, R.rem), nt = R.base + (run < R.rem ? 1 : 0);
  const MlpArgs& A = KM->a[net0 + net];
  const int row0 = slice * kLw2Rows, B = A.B;
  const int ke = wave & 7, par = wave >> 3;
  LW_STAMP(0);

  // ---- requests, in the order of use: rows, (first-layer fragments,) B fragments, bias / masks
  f32x4 v[4];
  float x0v = 0.f;
  f32x4 w0f[2][2];
  float b0f[2];
  const int K0 = FIN ? A.net.dims[0] : 0, NS0 = FIN ? cdiv(K0, 16) : 0;   // <= 2 (host-checked)
  if constexpr (FIN) {
    const int row = tid >> 5, col = tid & 31, gr = row0 + row;   // one element of the [32 x 32] input tile each
    if (gr < B && col < K0)
      x0v = col < A.k0 ? A.x0[(size_t)gr * A.k0 + col] : A.x1[(size_t)gr * A.k1 + col - A.k0];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tile = wave + 16 * q;
      b0f[q] = A.net.b[0][16 * tile + i];
#pragma unroll
      for (int st = 0; st < 2; ++st)
        w0f[q][st] = st < NS0 ? ld4(A.net.pf[0] + (((size_t)tile * NS0 + st) * 64 + lane) * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else if constexpr (ROLE != 2) {
    const float* src = BWD ? A.dYg[l] : A.Xg[l];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int idx = tid + h * kThreads;                 // 32 rows x 128 float4
      const int row = idx >> 7, col = (idx & 127) * 4, gr = row0 + row;
      v[h] = gr < B ? ld4(src + (size_t)gr * WIDTH + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  typename P::Frag b[3][NSE];
  {
    const float* pk0 = (BWD ? A.net.pb[l] : A.net.pf[l]) + (size_t)NSE * ke * P::kBlk + lane * 4;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int tl = par + 2 * q;
#pragma unroll
      for (int s = 0; s < NSE; ++s) b[q][s] = P::zf();
      if (tl < nt) {
        const float* pk = pk0 + (size_t)(t0 + tl) * NSW * P::kBlk;
#pragma unroll
        for (int s = 0; s < NSE; ++s) b[q][s] = P::ldf(pk + s * P::kBlk);
      }
    }
  }
  // this thread's (up to three) output elements: out tile ot = row tile * kLw2MaxRun + run tile
  size_t e_off[3];
  float e_x[3];
  bool e_ok[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = tid + k * kThreads;
    const int ot = e >> 8, r = e & 255, rt = ot / MAXRUN, tl = ot - rt * MAXRUN;
    const int gr = row0 + rt * 16 + (r >> 4), col = (t0 + tl) * 16 + (r & 15);
    e_ok[k] = ot < 2 * MAXRUN && tl < nt && gr < B;
    e_off[k] = e_ok[k] ? (size_t)gr * WIDTH + col : 0;
    e_x[k] = 0.f;
    if (e_ok[k]) e_x[k] = BWD ? A.Xg[l][e_off[k]] : A.net.b[l][col];
  }
  __builtin_amdgcn_sched_barrier(0);
  LW_STAMP(1);

  if constexpr (FIN) {
    xin[(tid >> 5) * kX0Ld + (tid & 31)] = x0v;
    __syncthreads();
    // dW's copies: the net's input rows (the run that starts a net stores them, from the registers) and,
    // below, h1 — every run its own columns (one run storing all 64 KB was 1.5 us of the launch)
    if (run == 0 && A.Xg[0] != nullptr) {
      const int row = tid >> 5, col = tid & 31;
      if (row0 + row < B && col < K0) A.Xg[0][(size_t)(row0 + row) * A.ldx0 + col] = x0v;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float* xr = xin + (rt * 16 + i) * kX0Ld + 4 * kk;
      const f32x4 a0 = ld4(xr), a1 = ld4(xr + 16);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        mac4(a0, w0f[q][0], acc);
        mac4(a1, w0f[q][1], acc);
        float* o = xs + (rt * 16 + kk * 4) * WL + 16 * (wave + 16 * q) + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * WL] = fmaxf(acc[r] + b0f[q], 0.f);
      }
    }
    __syncthreads();                          // h1 rows complete
    {
      const int c4n = nt * 4, row = tid / c4n, col = t0 * 16 + (tid - row * c4n) * 4;
      if (row < kLw2Rows && row0 + row < B)
        *reinterpret_cast<f32x4*>(A.Xg[1] + (size_t)(row0 + row) * WIDTH + col) = ld4(xs + row * WL + col);
    }
  } else {
    if constexpr (ROLE == 2) {
      // the producers of this (net, slice): one flag per run of the layer before, lanes of wave 0 poll
      int* s_okp = reinterpret_cast<int*>(xin);              // (a word behind the rows: no static LDS — two workgroups per CU)
      if (wave == 0) {
        bool ok = true;
        if (lane < R.rpn && net0 + net >= X->wait_net0) {
          const unsigned long long* f = X->flags + ((size_t)(net0 + net) * R.slices + slice) * kLwFlagStride + lane;
          ok = false;
          for (int spin = 0; spin < X->spin && !ok; ++spin) {
            ok = (unsigned)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == X->tag;
            if (!ok) __builtin_amdgcn_s_sleep(2);
          }
        }
        const bool all = __all(ok);
        if (lane == 0) {
          *s_okp = all ? 1 : 0;
          if (!all) report_expired(X->err, (KERN_LW_PAIR << 8) | SITE_LW_PAIR);
        }
      }
      __syncthreads();
      const bool ok = *s_okp != 0;
      const float* src = BWD ? A.dYg[l] : A.Xg[l];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int idx = tid + h * kThreads;
        const int row = idx >> 7, col = (idx & 127) * 4, gr = row0 + row;
        v[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gr < B) {
          // an agent-scope load (sc1: past this XCD's L2) the compiler counts (engine.h ld4_agent) — an inline-asm load's
          // result registers were copied and reused before it had returned (the bf16 instance: a GPU memory fault)
          v[h] = ld4_agent(src, (unsigned)(gr * WIDTH + col));
        }
        if (!ok) v[h] = f32x4{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};   // a lost producer shows up as NaN
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int idx = tid + h * kThreads;
      *reinterpret_cast<f32x4*>(xs + (idx >> 7) * WL + (idx & 127) * 4) = v[h];
    }
    __syncthreads();
  }

  LW_STAMP(2);
  // ---- partial tiles: the wave's K-eighth of its tiles, both row tiles against the same B fragments
  f32x4 acc[3][2];
  if constexpr (P::kX2) {
    // PrecX2: the A operand goes through fp16 (hi + lo).  The wave's [32 rows x 64 columns] block of the input is
    // scaled by the power of two that brings ITS largest magnitude to [2^10, 2^11) — no exchange between waves: a
    // partial tile is un-scaled before it meets the others in LDS (forward activations and backward gradients alike)
    f32x4 xa[2][NSE][2];
    float m = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float* xr = xs + (rt * 16 + i) * WL + 64 * ke + 4 * kk;
#pragma unroll
      for (int s = 0; s < NSE; ++s) {
        xa[rt][s][0] = ld4(xr + 32 * s);
        xa[rt][s][1] = ld4(xr + 32 * s + 16);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 4; ++t) m = fmaxf(m, fabsf(xa[rt][s][h][t]));
      }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float s1 = P::a_scale(m), un = P::kOut / s1;
    // the A block meets up to three B tiles: split into its fp16 hi / lo planes ONCE (per tile it was 2/3 of the VALU
    // work of this phase, r03-42: 96 v_fma_mix + 96 v_cvt per wave for 36 MFMAs)
    f16x8 ah[2][NSE], al[2][NSE];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int s = 0; s < NSE; ++s) x2_split8(xa[rt][s][0] * s1, xa[rt][s][1] * s1, ah[rt][s], al[rt][s]);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[q][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (par + 2 * q < nt) {
#pragma unroll
          for (int s = 0; s < NSE; ++s) P::mma3_split(ah[rt][s], al[rt][s], b[q][s], acc[q][rt]);
          acc[q][rt] *= un;
        }
      }
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        acc[q][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (par + 2 * q < nt) {
          const float* xr = xs + (rt * 16 + i) * WL + 64 * ke + 4 * kk;
#pragma unroll
          for (int s = 0; s < NSE; ++s) P::mac(xr, s, b[q][s], acc[q][rt]);
        }
      }
  }
  LW_STAMP(3);
  __syncthreads();                            // every wave is done with the rows: the partial tiles go over them
  LW_STAMP(4);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int tl = par + 2 * q;
    if (tl < nt)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
        *reinterpret_cast<f32x4*>(scr + ((size_t)((rt * MAXRUN + tl) * 8 + ke) * 64 + lane) * 4) = acc[q][rt];
  }
  __syncthreads();
  LW_STAMP(5);

  // ---- K-ordered sum, bias + ReLU (forward) or ReLU mask (backward), rows out
  float* dst = BWD ? A.dYg[l - 1] : A.Xg[l + 1];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!e_ok[k]) continue;
    const int e = tid + k * kThreads;
    const int ot = e >> 8, r = e & 255, row = r >> 4, col = r & 15;
    const float* sp = scr + ((size_t)(ot * 8) * 64 + (row >> 2) * 16 + col) * 4 + (row & 3);
    float vs = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) vs += sp[p * 256];
    const float res = BWD ? (e_x[k] > 0.f ? vs : 0.f) : fmaxf(vs + e_x[k], 0.f);
    if constexpr (ROLE == 1) __hip_atomic_store(dst + e_off[k], res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written through
    else dst[e_off[k]] = res;
  }
  LW_STAMP(6);
  if constexpr (ROLE == 1) {
    // every wave's rows are out before the workgroup says so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(X->flags + ((size_t)(net0 + net) * R.slices + slice) * kLwFlagStride + run, (unsigned long long)X->tag << 32,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE, class P = PrecF32>
__global__ __launch_bounds__(kThreads) void k_lw_mid_run2(const MlpMultiArgs M, int l, const LwRun2 R, int net0) {
  lw_mid_run2_body<MODE, P>((const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr(), l, R, (int)blockIdx.x, net0);
}

// A k_mlp_slice_tp launch (`A`: slices x 4 workgroups, blocks [0, host)) carrying, as riding workgroups behind its
// own, the first hidden launch (first layer folded in, MODE 2) of the nets `M` — which must not depend on it.
// TQC's critic step opens with the actor's forward on s' on 64 of 256 CUs; the online critics' first two layers on
// (s, a) need nothing of it, and as a launch of their own they are the next 10-16 us of the chain.
// `host` is a multiple of 8, so a rider's bx & 7 is still its XCD.
// Two consecutive hidden-layer launches as ONE (forward: the first hidden layer with the net input folded in + the
// second; backward: through the second + through the first).  Blocks [0, nb) are the first layer's, [nb, 2 nb) the
// second layer's: dispatched behind their producers — a workgroup only ever waits for workgroups dispatched before it —
// they take in their weight fragments (what bounds these launches) while the first layer's last workgroups are still
// running, and wait for the flags of their (net, slice) only then.  One workgroup per compute unit (113-121 registers
// in the x2 mode, 96 KB of partial tiles): the second layer's workgroups start as the first layer's retire — what is
// saved is the launch boundary (drain, dispatch ramp, cold instruction fetch) and the fragments' round trip.
constexpr int kLwPairMaxRun = kLw2MaxRun;
constexpr size_t kLwPairLds = sizeof(float) * (2 * kLwPairMaxRun * 8 * 256);
static_assert(kLwPairLds >= sizeof(float) * (kLw2Rows * lds_ld(512) + kLw2Rows * kX0Ld + 4), "rows + net input fit under the partial tiles");
template <int MODE_A, int MODE_B, class P>
__global__ __launch_bounds__(kThreads) void k_lw_mid_pair(const MlpMultiArgs M, int la, int lb, const LwRun2 R, int nb, const LwPair X) {
  const MlpMultiArgs* KM = (const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  if ((int)blockIdx.x < nb) lw_mid_run2_body<MODE_A, P, 1, kLwPairMaxRun>(KM, la, R, (int)blockIdx.x, 0, &X);
  else lw_mid_run2_body<MODE_B, P, 2, kLwPairMaxRun>(KM, lb, R, (int)blockIdx.x - nb, 0, &X);
}

template <class P>
__global__ __launch_bounds__(kThreads) void k_slice_tp_fin(const MlpMultiArgs M, const MlpArgs A, const LwRun2 R, int host,
                                                           int slices) {
  const int b = (int)blockIdx.x;
  if (b < host) {
    if (b < slices * 4) slice_tp_body(A, b % slices, b / slices);
    return;
  }
  lw_mid_run2_body<2, P>((const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr(), 1, R, b - host, 0);
}

// The rest of such a first launch — the nets that did not fit beside k_slice_tp_fin's host — as riders of a head
// launch (k_lw_head, blockIdx.z >= z0): their own argument block, R.nets nets from net0 on.
// z1 >= 0 (r06-12): behind the tail, the SECOND hidden layer's forward of ALL of M's nets rides too (blockIdx.z >= z1, run
// shape R2) — the online critics' layer on (s, a) that was a launch of its own (k_lw_mid_run2<0>, 240 workgroups, 8.9 us)
// between the target pass's heads and the online heads.  The tail's workgroups then publish their rows as the first
// layer of a k_lw_mid_pair does (written through + a flag per run: X), and the riders of the nets >= net0 wait for those
// flags; the other nets' rows are an earlier launch's.  Same arithmetic as the stand-alone launch: bit-identical rows.
struct LwFinTail {
  MlpMultiArgs M;
  LwRun2 R;
  int net0, z0, prec;     // z0 < 0: nothing rides; prec: 0 fp32, 1 bf16, 2 split fp16 (the hidden layer's packs)
  LwRun2 R2;
  LwPair X;
  int z1;
};

inline LwRun2 lw_run2(int B, int nets, int n_cus) {
  LwRun2 r;
  r.slices = (B + kLw2Rows - 1) / kLw2Rows;
  r.nets = nets;
  int rpn = n_cus / (nets * r.slices);
  const int min_rpn = (32 + kLw2MaxRun - 1) / kLw2MaxRun;
  if (rpn < min_rpn) rpn = min_rpn;
  if (rpn > 32) rpn = 32;
  r.rpn = rpn;
  r.base = 32 / rpn;
  r.rem = 32 % rpn;
  r.ppx = (nets * rpn + 7) / 8;
  return r;
}
constexpr size_t kLwRun2Lds = sizeof(float) * (2 * kLw2MaxRun * 8 * 256);   // partial tiles 96 KB >= rows 65 KB + net input 13 KB
static_assert(kLwRun2Lds >= sizeof(float) * (kLw2Rows * lds_ld(512) + kLw2Rows * kX0Ld), "rows + net input fit under the partial tiles");

// TqcJob (kernels.h): the slice's TD targets by the last of the nets' head workgroups to arrive.  Called by all
// threads of a forward-only head workgroup after slice_head().
__device__ __forceinline__ void lw_tqc_target(const TqcJob& J, const MlpArgs& A, const float* outS, int Nout, int row0,
                                              int slice) {
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, B = A.B;
  for (int idx = tid; idx < kR * Nout; idx += kThreads) {
    const int row = idx / Nout, col = idx - row * Nout;
    if (row0 + row < B)
      __hip_atomic_store(A.out + (size_t)(row0 + row) * A.ldo + col, outS[row * kOutLd + col], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // written through before this workgroup counts as arrived
  __syncthreads();
  if (tid == 0) {
    const unsigned long long old = __hip_atomic_fetch_add(J.counter + slice, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ((old + 1ull) % (unsigned long long)J.n_nets) == 0ull ? 1 : 0;
  }
  __syncthreads();
  if (s_last == 0) return;
  const int row = row0 + wave;                          // one wave per row
  if (row >= B) return;
  // 128-slot ascending bitonic network in REGISTERS: lane l holds slots l and l + 64; a stage's partner is the other
  // register (j = 64) or lane l ^ j (a cross-lane move) — the same compare-exchange network as k_tqc_target's LDS
  // version, hence the same sorted row, in ~1/10 of its time (28 stages of LDS round trips were 6 us of this launch)
  const int total = J.n_nets * J.Q, Mt = total - J.drop;
  auto fetch = [&](int e) {
    float v = __builtin_huge_valf();
    if (e < total) {
      const int n = e / J.Q, q = e - n * J.Q;
      v = __hip_atomic_load(J.z + n * J.net_stride + (size_t)row * J.ldz + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
  };
  // (what the epilogue needs, requested with the samples: behind the sort they were one more round trip)
  const double la = *J.log_alpha;
  const float lp = J.logp[row], dd = J.d[row], rr = J.r[row];
  float v0 = fetch(lane), v1 = fetch(lane + 64);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 2; k <= 128; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j == 64) {                                    // (k = 128: ascending everywhere)
        const float lo = fminf(v0, v1), hi = fmaxf(v0, v1);
        v0 = lo; v1 = hi;
      } else {
        const float p0 = __shfl_xor(v0, j), p1 = __shfl_xor(v1, j);
        const bool low = (lane & j) == 0;               // this slot is the lower index of its pair
        const bool up0 = (lane & k) == 0, up1 = ((lane + 64) & k) == 0;
        v0 = (low == up0) ? fminf(v0, p0) : fmaxf(v0, p0);
        v1 = (low == up1) ? fminf(v1, p1) : fmaxf(v1, p1);
      }
    }
  }
  const float alpha = (float)exp(la);
  const float al = alpha * lp;
  const float coef = (1.f - dd) * J.gamma;
  if (lane < Mt) J.target[(size_t)row * Mt + lane] = rr + coef * (v0 - al);
  if (lane + 64 < Mt) J.target[(size_t)row * Mt + lane + 64] = rr + coef * (v1 - al);
}

// A slice's [16 x 512] rows of `src` into `hb` and the NARROW product out = rows · pack (NTo <= 2 output tiles over the 32
// steps of a 512-deep contraction) with EVERYTHING requested at entry: the rows, this wave's fragments (its steps of its
// tile: at most four), the bias element of the thread that finishes an element — then `more()` (the caller's further
// requests).  Through load_rows4 + gemm_packed the fragments were requested after the rows had landed and been staged:
// one more cold round trip on launches of 80 workgroups that are nothing but a latency chain (r06-10).  The products and
// their order are gemm_packed's narrow form (engine.h): epi(row, col, sum + bias) once per element, a barrier behind it.
template <class More, class Epi>
__device__ __forceinline__ void lw_rows_narrow_gemm(const float* __restrict__ src, const float* __restrict__ pack, int NTo, bool run,
                                                    const float* __restrict__ bias, int nbias, int row0, int B, float* hb,
                                                    float* scr, More&& more, Epi&& epi) {
  constexpr int WIDTH = 512, WL = lds_ld(WIDTH), NTW = WIDTH / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kk = lane >> 4;
  f32x4 rv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = tid + h * kThreads, row = idx >> 7, col = (idx & 127) * 4, gr = row0 + row;
    rv[h] = gr < B ? ld4(src + (size_t)gr * WIDTH + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int wpt = kWaves / NTo, tile_f = wave / wpt, part = wave - tile_f * wpt;
  const int per = cdiv(NTW, wpt), s0 = part * per, s1 = min(NTW, s0 + per);      // per = 2 / 4
  f32x4 fb[4];
  {
    const float* pl = pack + ((size_t)tile_f * NTW) * 256 + lane * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) fb[d] = (run && s0 + d < s1) ? ld4(pl + (size_t)(s0 + d) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool rmine = tid < kR * 16 * NTo;
  const int rt = tid / (kR * 16), rrem = tid - rt * (kR * 16), rrow = rrem >> 4, rcol = 16 * rt + (rrem & 15);
  const float rb = (run && bias != nullptr && rmine && rcol < nbias) ? bias[rcol] : 0.f;
  more();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int idx = tid + h * kThreads;
    *reinterpret_cast<f32x4*>(hb + (idx >> 7) * WL + (idx & 127) * 4) = rv[h];
  }
  __syncthreads();                                    // the rows are visible
  if (!run) return;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* xrow = hb + i * WL + 4 * kk;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int st = s0 + d;
    if (st < s1) {
      const f32x4 a4 = ld4(xrow + 16 * st);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = mfma4(a4[t], fb[d][t], acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) scr[(wave * kR + kk * 4 + r) * 16 + i] = acc[r];
  __syncthreads();
  if (rmine) {
    float v = 0.f;
    const float* sp = scr + ((rt * wpt) * kR + rrow) * 16 + (rrem & 15);
#pragma unroll 4
    for (int pq = 0; pq < wpt; ++pq) v += sp[pq * kR * 16];
    epi(rrow, rcol, v + rb);
  }
  __syncthreads();
}

// R (n_ride = 4): ANOTHER net's k_mlp_slice_tp launch on the same slices — its clusters of four ride as the
// workgroups blockIdx.z >= nets (slice_tp_body.h).  The heads are (slices x nets) workgroups, 80 of TQC's 256 CUs:
// the actor's forward on s, which the actor step needs only after the critic step, runs beside the critic step's
// heads instead of as a launch of its own (8.5 us).  The host checks that all workgroups are resident at once
// (the riders' cluster exchanges wait for each other).
constexpr size_t kLwHeadTailOffset =       // LwFinTail's place in k_lw_head's kernel-argument segment
    lw_align_up(lw_align_up(lw_align_up(lw_align_up(sizeof(MlpMultiArgs), alignof(TqcJob)) + sizeof(TqcJob), alignof(MlpArgs)) +
                                sizeof(MlpArgs), alignof(int)) + sizeof(int), alignof(LwFinTail));
template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_lw_head(const MlpMultiArgs M, const TqcJob J, const MlpArgs R, int nets,
                                                       const LwFinTail F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];        // max(LwLds<WIDTH>::total, SliceLds<256>::total(2)[, kLwRun2Lds]) floats
  if (F.z0 >= 0 && (int)blockIdx.z >= F.z0) {
    const LwFinTail* KF = (const LwFinTail*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kLwHeadTailOffset);
    if (F.z1 >= 0 && (int)blockIdx.z >= F.z1) {             // the second hidden layer's forward, all nets
      const int bx = ((int)blockIdx.z - F.z1) * (int)gridDim.x + (int)blockIdx.x;
      if (F.prec == 2) lw_mid_run2_body<0, PrecX2, 2>(&KF->M, 2, KF->R2, bx, 0, &KF->X);
      else if (F.prec == 1) lw_mid_run2_body<0, PrecBF16, 2>(&KF->M, 2, KF->R2, bx, 0, &KF->X);
      else lw_mid_run2_body<0, PrecF32, 2>(&KF->M, 2, KF->R2, bx, 0, &KF->X);
      return;
    }
    const int bx = ((int)blockIdx.z - F.z0) * (int)gridDim.x + (int)blockIdx.x;
    if (F.z1 >= 0) {                                        // the tail publishes its rows for those riders
      if (F.prec == 2) lw_mid_run2_body<2, PrecX2, 1>(&KF->M, 1, KF->R, bx, F.net0, &KF->X);
      else if (F.prec == 1) lw_mid_run2_body<2, PrecBF16, 1>(&KF->M, 1, KF->R, bx, F.net0, &KF->X);
      else lw_mid_run2_body<2, PrecF32, 1>(&KF->M, 1, KF->R, bx, F.net0, &KF->X);
      return;
    }
    if (F.prec == 2) lw_mid_run2_body<2, PrecX2>(&KF->M, 1, KF->R, bx, F.net0);
    else if (F.prec == 1) lw_mid_run2_body<2, PrecBF16>(&KF->M, 1, KF->R, bx, F.net0);
    else lw_mid_run2_body<2, PrecF32>(&KF->M, 1, KF->R, bx, F.net0);
    return;
  }
  if ((int)blockIdx.z >= nets) {
    slice_tp_body(R, (int)blockIdx.x, (int)blockIdx.z - nets);
    return;
  }
  using LY = LwLds<WIDTH>;
  constexpr int WL = LY::WL, NTW = WIDTH / 16;
  const MlpArgs& A = lw_args(blockIdx.z);
  float* hb = smem + LY::x;
  float* outS = smem + LY::out;
  float* auxS = smem + LY::aux;
  float* scr = smem + LY::scr;
  const int slice = blockIdx.x, row0 = slice * kR, B = A.B;
  const int L = A.net.n_layers, Nout = A.net.dims[L];
  const int NTo = cdiv(Nout, 16);
  if (NTo <= 2 && WIDTH == 512) {
    // ---- narrow heads (TQC: 25 quantiles): EVERYTHING the workgroup will need is requested at entry — the slice's rows, this
    // wave's fragments of the forward head and (`more`) of the backward step: its two column tiles, two steps each.  Through
    // gemm_packed the backward fragments were requested after the seeds.  Same products in the same order (engine.h).
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kk = lane >> 4;
    f32x4 bb[2][2];
    SeedPre sp;
    lw_rows_narrow_gemm(A.Xg[L - 1], A.net.pf[L - 1], NTo, A.do_fwd != 0, A.net.b[L - 1], Nout, row0, B, hb, scr,
                        [&]() {
                          sp = seed_pre_request(A, row0);       // (the quantile-Huber seed's target samples)
#pragma unroll
                          for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int st = 0; st < 2; ++st)
                              bb[q][st] = (A.do_bwd && st < NTo)
                                              ? ld4(A.net.pb[L - 1] + ((size_t)(wave + kWaves * q) * NTo + st) * 256 + lane * 4)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
                        },
                        [&](int row, int col, float v) { outS[row * kOutLd + col] = col < Nout ? v : 0.f; });
    if (A.do_fwd) slice_head(A, outS, Nout, row0, true);
    if (!A.do_bwd) {
      if (J.counter != nullptr) {
        __syncthreads();                                  // slice_head is done with outS / scr
        lw_tqc_target(J, A, outS, Nout, row0, slice);
      }
      return;
    }
    slice_seed(A, outS, auxS, scr, Nout, L, row0, slice, true, sp);
    __syncthreads();                                    // dout visible
    // dz[L-2] = (dout W_{L-1}) * [h > 0], in place over the activations (each element's mask is read by the lane that
    // overwrites it): wave w owns column tiles w and w + 16
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* xrow = auxS + i * kOutLd + 4 * kk;
#pragma unroll
      for (int st = 0; st < 2; ++st)
        if (st < NTo) {
          const f32x4 a4 = ld4(xrow + 16 * st);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc = mfma4(a4[t], bb[q][st][t], acc);
        }
      const int col = 16 * (wave + kWaves * q) + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* pp = hb + (kk * 4 + r) * WL + col;
        *pp = *pp > 0.f ? acc[r] : 0.f;
      }
    }
    __syncthreads();
    store_rows4(hb, WL, A.dYg[L - 2], WIDTH, WIDTH, row0, B);
    return;
  }
  load_rows4(hb, WL, A.Xg[L - 1], WIDTH, WIDTH, row0, B);
  if (A.do_fwd) {
    gemm_packed(hb, WL, A.net.pf[L - 1], cdiv(Nout, 16), NTW, scr, A.net.b[L - 1], Nout,
                [&](int row, int col, float v) { outS[row * kOutLd + col] = col < Nout ? v : 0.f; });
    slice_head(A, outS, Nout, row0, true);
  } else {
    __syncthreads();
  }
  if (!A.do_bwd) {
    if (J.counter != nullptr) {
      __syncthreads();                                  // slice_head is done with outS / scr
      lw_tqc_target(J, A, outS, Nout, row0, slice);
    }
    return;
  }
  slice_seed(A, outS, auxS, scr, Nout, L, row0, slice, true);
  // dz[L-2] = (dout W_{L-1}) * [h > 0], in place over the activations (each element's mask is
  // read by the lane that overwrites it)
  gemm_packed(auxS, kOutLd, A.net.pb[L - 1], NTW, cdiv(Nout, 16), scr, nullptr, 0, [&](int row, int col, float v) {
    float* p = hb + row * WL + col;
    *p = *p > 0.f ? v : 0.f;
  });
  __syncthreads();
  store_rows4(hb, WL, A.dYg[L - 2], WIDTH, WIDTH, row0, B);
}

// Pj (blockIdx.z == Pj.z0): one more row of workgroups gathers the NEXT update's minibatch rows from the replay
// (step_n: a k_replay_gather launch per update, 5.6 us + gap, as riders of a launch that fills a third of the chip)
// Rd (z_r >= 0, r06-16): the ACTOR's backward — the k_mlp_slice_tp launch that follows this one and consumes its rows (64
// workgroups, 8.4 us as a launch of its own) — rides as the workgroups blockIdx.z >= z_r (slice_tp_body.h): it takes in
// its fragments and stored activations while the action gradients are formed, and waits for the flags the dact workgroups
// raise behind their written-through rows (SeedArgs::da_flags).  Dispatched behind the dact (and prefetch) rows: a
// workgroup only ever waits for workgroups dispatched before it; the host checks that all are resident at once.
// D (z_t >= 0, r06-18): ... and behind the backward ITS dW + Adam tiles (the k_dw_adam launch that followed: 152 16 x 32
// tiles + the temperature's step, 7.4 us), gated as the critic's tiles on a phase launch are (dw_body.h GATE 1): they take in
// their Adam state while the backward runs and wait for its members' flags (MlpArgs::done_flags) before the rows.
struct LwDactRide { MlpArgs R; unsigned long long* flags; unsigned tag; int fstride, z_r, z_t, n_tile_wgs; };
constexpr size_t kLwDactPjOffset = lw_align_up(sizeof(MlpMultiArgs), alignof(PrefetchJob));
constexpr size_t kLwDactDwOffset = lw_align_up(lw_align_up(kLwDactPjOffset + sizeof(PrefetchJob), alignof(LwDactRide)) + sizeof(LwDactRide), alignof(DwKArgs));
static_assert(LwLds<512>::total >= 2 * kR * kX0Ld + 96 + kMaxEnds, "the prefetch riders' LDS fits in the launch's");
template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_lw_dact(const MlpMultiArgs M, const PrefetchJob Pj, const LwDactRide Rd, const DwKArgs D) {
  extern __shared__ __attribute__((aligned(16))) float smem[];      // max(LwLds<WIDTH>::total, SliceLds<256>::total(2)[, kDwLdsFloats]) floats
  if (Rd.z_t >= 0 && (int)blockIdx.z >= Rd.z_t) {
    const int tile = ((int)blockIdx.z - Rd.z_t) * (int)gridDim.x + (int)blockIdx.x;
    if (tile >= Rd.n_tile_wgs || threadIdx.x >= kDwThreads) return;     // a tile workgroup is the first 8 waves (the stand-alone kernel's shape and arithmetic)
    dw_adam_body<false, 1>(*(const DwKArgs*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kLwDactDwOffset), smem, tile);
    return;
  }
  if (Rd.z_r >= 0 && (int)blockIdx.z >= Rd.z_r) {
    slice_tp_body(Rd.R, (int)blockIdx.x, (int)blockIdx.z - Rd.z_r);
    return;
  }
  if (Pj.z0 >= 0 && (int)blockIdx.z >= Pj.z0) {
    prefetch_rows_body(*(const PrefetchJob*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kLwDactPjOffset),
                       (int)blockIdx.x, smem);
    return;
  }
  using LY = LwLds<WIDTH>;
  constexpr int WL = LY::WL, NTW = WIDTH / 16;
  const MlpArgs& A = lw_args(blockIdx.z);
  float* xs = smem + LY::x;
  float* dactS = smem + LY::aux;
  float* scr = smem + LY::scr;
  const int row0 = blockIdx.x * kR, B = A.B;
  const int c0 = A.dact_col0, nc = A.dact_cols;
  auto epi = [&](int row, int col, float v) {
    const int c = col - c0;
    if (c >= 0 && c < nc) dactS[row * kOutLd + c] = v;
  };
  if (WIDTH == 512 && cdiv(A.net.dims[0], 16) <= 2) {     // (a narrow net input: rows and fragments requested together)
    lw_rows_narrow_gemm(A.dYg[0], A.net.pb[0], cdiv(A.net.dims[0], 16), true, nullptr, 0, row0, B, xs, scr, [] {}, epi);
  } else {
    load_rows4(xs, WL, A.dYg[0], WIDTH, WIDTH, row0, B);
    gemm_packed(xs, WL, A.net.pb[0], cdiv(A.net.dims[0], 16), NTW, scr, nullptr, 0, epi);
  }
  if (Rd.z_r < 0) {
    store_rows(dactS, kOutLd, A.dact, A.lddact, nc, row0, B);
    return;
  }
  // the riding backward reads these rows: written through, acknowledged, then the (net, slice) flag
  store_rows_wt(dactS, kOutLd, A.dact, A.lddact, nc, row0, B);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(Rd.flags + (size_t)blockIdx.z * Rd.fstride + blockIdx.x, (unsigned long long)Rd.tag << 32, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Can these launches (one for_each_net round) run layer by layer?  Equal shapes and flags, wide
// hidden layers, every exchange buffer present.
bool mlp_layerwise_ok(const MlpArgs* a, int n, int width) {
  if (n < 1 || n > kMaxMulti || width != 512) return false;
  const MlpArgs& r = a[0];
  const int L = r.net.n_layers;
  if (L < 3) return false;
  for (int j = 0; j < n; ++j) {
    const MlpArgs& x = a[j];
    if (x.B != r.B || x.net.n_layers != L || x.do_fwd != r.do_fwd || x.do_bwd != r.do_bwd) return false;
    if ((x.dact_cols > 0) != (r.dact_cols > 0)) return false;
    if (x.net.dims[0] > 96 || x.net.dims[L] > kNarrowMax) return false;
    for (int l = 1; l < L; ++l)
      if (x.net.dims[l] != width || x.Xg[l] == nullptr) return false;
    if (x.do_bwd)
      for (int l = 0; l + 1 < L; ++l)
        if (x.dYg[l] == nullptr) return false;
    if (x.dact_cols > 0 && x.dact == nullptr) return false;
  }
  return true;
}

hipError_t init_layerwise_attrs() {
  {
    constexpr size_t head_f = LwLds<512>::total, ride_f = SliceLds<256>::total(2);
    size_t lds = sizeof(float) * (ride_f > head_f ? ride_f : head_f);
    if (kLwRun2Lds > lds) lds = kLwRun2Lds;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_head<512>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  {
    constexpr size_t dact_f = LwLds<512>::total, ride_f = SliceLds<256>::total(2), tile_f = kDwLdsFloats;
    constexpr size_t m1 = ride_f > dact_f ? ride_f : dact_f;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lw_dact<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(sizeof(float) * (tile_f > m1 ? tile_f : m1)));
    if (e != hipSuccess) return e;
  }
  for (const void* k : {reinterpret_cast<const void*>(&k_slice_tp_fin<PrecF32>), reinterpret_cast<const void*>(&k_slice_tp_fin<PrecBF16>),
                        reinterpret_cast<const void*>(&k_slice_tp_fin<PrecX2>)}) {
    constexpr size_t ride_f = SliceLds<256>::total(2);
    const size_t lds = sizeof(float) * ride_f > kLwRun2Lds ? sizeof(float) * ride_f : kLwRun2Lds;
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const void* k2[9] = {reinterpret_cast<const void*>(&k_lw_mid_run2<0>), reinterpret_cast<const void*>(&k_lw_mid_run2<1>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<2>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<0, PrecBF16>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<1, PrecBF16>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<2, PrecBF16>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<0, PrecX2>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<1, PrecX2>),
                       reinterpret_cast<const void*>(&k_lw_mid_run2<2, PrecX2>)};
  for (const void* k : k2) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLwRun2Lds);
    if (e != hipSuccess) return e;
  }
  const void* kp[6] = {reinterpret_cast<const void*>(&k_lw_mid_pair<2, 0, PrecF32>), reinterpret_cast<const void*>(&k_lw_mid_pair<1, 1, PrecF32>),
                       reinterpret_cast<const void*>(&k_lw_mid_pair<2, 0, PrecBF16>), reinterpret_cast<const void*>(&k_lw_mid_pair<1, 1, PrecBF16>),
                       reinterpret_cast<const void*>(&k_lw_mid_pair<2, 0, PrecX2>), reinterpret_cast<const void*>(&k_lw_mid_pair<1, 1, PrecX2>)};
  for (const void* k : kp) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLwPairLds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// prec 1 / 2: the nets' pf / pb of the HIDDEN layers (1 .. L-2) point at bf16 / split-fp16 packs (MlpArgs::pf16 / pb16 moved in
// by the caller) and the hidden-layer launches run PrecBF16; needs the balanced-run kernels.
bool mlp_slice_tp_shape_ok(const MlpArgs& a, int width);

// Can `rider` (a prepared k_mlp_slice_tp launch: tag drawn, exchange area set) ride on the head launch of these nets?
bool mlp_layerwise_rider_ok(const MlpArgs* a, int n, const MlpArgs& rider, int n_cus) {
  const int slices = (a[0].B + kR - 1) / kR;
  return rider.B == a[0].B && rider.tp_xbuf != nullptr && mlp_slice_tp_shape_ok(rider, 256) &&
         slices * (n + 4) <= (n_cus > 0 ? n_cus : 256);
}

// job: a TD-target job for the heads of a forward-only launch of all its nets (TqcJob, kernels.h), or null
// rider: see k_lw_head; or null

// Can the first hidden launch (first layer folded in) of these nets ride on another launch (k_slice_tp_fin)?
bool mlp_layerwise_fin_ok(const MlpArgs* a, int n, int width) {
  if (!mlp_layerwise_ok(a, n, width) || !a[0].do_fwd || a[0].net.n_layers < 3) return false;
  for (int j = 0; j < n; ++j)
    if (a[j].net.dims[0] > 32) return false;
  return true;
}

// How many of these n nets' first-launch workgroups fit beside `host_wgs` workgroups of a host launch, all resident at once
int mlp_layerwise_fin_fit(const MlpArgs* a, int n, int host_wgs, int n_cus) {
  const int cus = n_cus > 0 ? n_cus : 256;
  const LwRun2 r = lw_run2(a[0].B, n, cus);          // (the run shape of the launch of all n nets: kept for every subset)
  const int per_net = r.rpn * r.slices;
  int fit = per_net > 0 ? (cus - host_wgs) / per_net : 0;
  return fit < 0 ? 0 : (fit > n ? n : fit);
}

// the run shape of `n_all` nets' launch, applied to `n_sub` of them
static LwRun2 lw_run2_subset(int B, int n_all, int n_sub, int n_cus) {
  LwRun2 r = lw_run2(B, n_all, n_cus);
  r.nets = n_sub;
  r.ppx = (n_sub * r.rpn + 7) / 8;
  return r;
}

// `host`: a prepared k_mlp_slice_tp launch (tag drawn); the first hidden launch of nets a[0 .. n_ride) of the n rides behind it
hipError_t launch_slice_tp_with_fin(const MlpArgs& host, const MlpArgs* a, int n, int n_ride, int width, int n_cus, hipStream_t st,
                                    int prec) {
  if (!mlp_layerwise_fin_ok(a, n, width) || host.tp_xbuf == nullptr || !mlp_slice_tp_shape_ok(host, 256) || n_ride < 1 || n_ride > n)
    return hipErrorInvalidValue;
  MlpMultiArgs m;
  for (int j = 0; j < n; ++j) m.a[j] = a[j];
  for (int j = n; j < kMaxMulti; ++j) m.a[j] = a[0];
  const LwRun2 r2 = lw_run2_subset(a[0].B, n, n_ride, n_cus > 0 ? n_cus : 256);
  const int slices = (host.B + kR - 1) / kR, hostb = (slices * 4 + 7) / 8 * 8;
  constexpr size_t ride_f = SliceLds<256>::total(2);
  const size_t lds = sizeof(float) * ride_f > kLwRun2Lds ? sizeof(float) * ride_f : kLwRun2Lds;
  const dim3 grid(hostb + 8 * r2.ppx * r2.slices), blk(kThreads);
  if (prec == 2) hipLaunchKernelGGL((k_slice_tp_fin<PrecX2>), grid, blk, lds, st, m, host, r2, hostb, slices);
  else if (prec == 1) hipLaunchKernelGGL((k_slice_tp_fin<PrecBF16>), grid, blk, lds, st, m, host, r2, hostb, slices);
  else hipLaunchKernelGGL((k_slice_tp_fin<PrecF32>), grid, blk, lds, st, m, host, r2, hostb, slices);
  return hipGetLastError();
}

// The first hidden launch of nets t[t0 .. t_n) (of t_n nets that pass mlp_layerwise_fin_ok) as a launch of its own
hipError_t launch_mlp_layerwise_first(const MlpArgs* t, int t_n, int t0, int width, int n_cus, hipStream_t st, int prec) {
  if (!mlp_layerwise_fin_ok(t, t_n, width) || t0 < 0 || t0 >= t_n) return hipErrorInvalidValue;
  MlpMultiArgs m;
  for (int j = 0; j < t_n; ++j) m.a[j] = t[j];
  for (int j = t_n; j < kMaxMulti; ++j) m.a[j] = t[0];
  const LwRun2 r = lw_run2_subset(t[0].B, t_n, t_n - t0, n_cus > 0 ? n_cus : 256);
  const dim3 grid(8 * r.ppx * r.slices), blk(kThreads);
  if (prec == 2) hipLaunchKernelGGL((k_lw_mid_run2<2, PrecX2>), grid, blk, kLwRun2Lds, st, m, 1, r, t0);
  else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_run2<2, PrecBF16>), grid, blk, kLwRun2Lds, st, m, 1, r, t0);
  else hipLaunchKernelGGL((k_lw_mid_run2<2>), grid, blk, kLwRun2Lds, st, m, 1, r, t0);
  return hipGetLastError();
}

// prefetch: a PrefetchJob (batch_rows.h) for a launch with a k_lw_dact stage: the next update's rows, or null
// first_done: the first hidden launch of this forward already ran (launch_slice_tp_with_fin [+ a tail])
// tail / tail_n / tail0 / tail16: ANOTHER forward's nets tail[0 .. tail_n), of which the first hidden launch of
//   tail[tail0 ..] rides on this launch's heads (LwFinTail); null: nothing
hipError_t launch_mlp_layerwise(const MlpArgs* a, int n, int width, int n_cus, hipStream_t st, int prec, const TqcJob* job,
                                const MlpArgs* rider, bool first_done, const MlpArgs* tail, int tail_n, int tail0, int tail_prec,
                                const PrefetchJob* prefetch, const LwPairBuf* pairs, bool second_done, bool* second_rode,
                                const MlpArgs* bwd_rider, bool* bwd_rode, const DwKArgs* bwd_tiles, int bwd_tile_wgs, bool* tiles_rode) {
  if (second_rode != nullptr) *second_rode = false;
  if (bwd_rode != nullptr) *bwd_rode = false;
  if (tiles_rode != nullptr) *tiles_rode = false;
  if (!mlp_layerwise_ok(a, n, width)) return hipErrorInvalidValue;
  if (second_done && !first_done) return hipErrorInvalidValue;
  if (first_done && !mlp_layerwise_fin_ok(a, n, width)) return hipErrorInvalidValue;
  if (tail != nullptr && (!mlp_layerwise_fin_ok(tail, tail_n, width) || tail0 < 0 || tail0 >= tail_n || tail[0].B != a[0].B))
    return hipErrorInvalidValue;
  if (rider != nullptr && !mlp_layerwise_rider_ok(a, n, *rider, n_cus)) return hipErrorInvalidValue;
  if (job != nullptr && (a[0].do_bwd || !a[0].do_fwd || job->n_nets != n || job->n_nets * job->Q > 128 || job->counter == nullptr))
    return hipErrorInvalidValue;
  MlpMultiArgs m;
  for (int j = 0; j < n; ++j) m.a[j] = a[j];
  for (int j = n; j < kMaxMulti; ++j) m.a[j] = a[0];
  const int slices = (a[0].B + kR - 1) / kR, L = a[0].net.n_layers;
  const LwGrid g = lw_grid(slices, width / kLwCols, n);
  const dim3 wide(lw_blocks(g)), narrow(slices, 1, n), blk(kThreads);
  // hidden layers: 32-row runs of column tiles balanced over the CUs (k_lw_mid_run2)
  const LwRun2 r2 = lw_run2(a[0].B, n, n_cus > 0 ? n_cus : 256);
  const dim3 runs2(8 * r2.ppx * r2.slices);
  auto mid = [&](int mode, int l) {
    if (mode == 0) {
      if (prec == 2) hipLaunchKernelGGL((k_lw_mid_run2<0, PrecX2>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_run2<0, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else hipLaunchKernelGGL((k_lw_mid_run2<0>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
    }
    if (mode == 1) {
      if (prec == 2) hipLaunchKernelGGL((k_lw_mid_run2<1, PrecX2>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_run2<1, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else hipLaunchKernelGGL((k_lw_mid_run2<1>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
    }
    if (mode == 2) {
      if (prec == 2) hipLaunchKernelGGL((k_lw_mid_run2<2, PrecX2>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_run2<2, PrecBF16>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
      else hipLaunchKernelGGL((k_lw_mid_run2<2>), runs2, blk, kLwRun2Lds, st, m, l, r2, 0);
    }
  };
  // two hidden layers (TQC's critics): each direction's two launches as one (k_lw_mid_pair)
  const LwRun2 rp = lw_run2(a[0].B, n, n_cus > 0 ? n_cus : 256);
  const bool pair_ok = pairs != nullptr && pairs->flags != nullptr && L == 4 && rp.base + (rp.rem ? 1 : 0) <= kLwPairMaxRun &&
                       rp.rpn <= kLwFlagStride && n * rp.slices * kLwFlagStride <= pairs->n_flags;
  auto pair = [&](int dir, int la, int lb) {       // dir 0: forward (first layer folded in) + forward, 1: backward + backward
    LwPair X;
    X.flags = pairs->flags; X.tag = pairs->next_tag + (unsigned)pairs->used; X.spin = pairs->spin; X.err = pairs->err;
    X.wait_net0 = 0;
    pairs->used += 1;
    const int nb = 8 * rp.ppx * rp.slices;
    const dim3 grid(2 * nb);
    if (dir == 0) {
      if (prec == 2) hipLaunchKernelGGL((k_lw_mid_pair<2, 0, PrecX2>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
      else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_pair<2, 0, PrecBF16>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
      else hipLaunchKernelGGL((k_lw_mid_pair<2, 0, PrecF32>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
    } else {
      if (prec == 2) hipLaunchKernelGGL((k_lw_mid_pair<1, 1, PrecX2>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
      else if (prec == 1) hipLaunchKernelGGL((k_lw_mid_pair<1, 1, PrecBF16>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
      else hipLaunchKernelGGL((k_lw_mid_pair<1, 1, PrecF32>), grid, blk, kLwPairLds, st, m, la, lb, rp, nb, X);
    }
  };
  if (a[0].do_fwd) {
    // a narrow net input (two macro steps) is folded into the first hidden layer's launch
    bool fuse_in = L >= 3;
    for (int j = 0; j < n; ++j) fuse_in = fuse_in && a[j].net.dims[0] <= 32;
    if (!fuse_in) hipLaunchKernelGGL(k_lw_in<512>, wide, blk, 0, st, m, g);
    if (pair_ok && fuse_in && !first_done && (pairs->use & 1) != 0) {
      pair(0, 1, 2);
    } else {
      for (int l = 1; l + 1 < L; ++l) {
        if (l == 1 && first_done) continue;
        if (l == 2 && second_done) continue;
        mid(l == 1 && fuse_in ? 2 : 0, l);
      }
    }
  }
  {
    constexpr size_t head_f = LwLds<512>::total, ride_f = SliceLds<256>::total(2);
    size_t lds = sizeof(float) * (rider != nullptr && ride_f > head_f ? ride_f : head_f);
    int z = n + (rider != nullptr ? 4 : 0);
    static const LwFinTail none = [] { LwFinTail f; memset((void*)&f, 0, sizeof f); f.z0 = -1; f.z1 = -1; return f; }();
    LwFinTail ft_local;
    const LwFinTail* ft = &none;
    if (tail != nullptr) {
      for (int j = 0; j < tail_n; ++j) ft_local.M.a[j] = tail[j];
      for (int j = tail_n; j < kMaxMulti; ++j) ft_local.M.a[j] = tail[0];
      ft_local.R = lw_run2_subset(tail[0].B, tail_n, tail_n - tail0, n_cus > 0 ? n_cus : 256);
      ft_local.net0 = tail0; ft_local.z0 = z; ft_local.prec = tail_prec;
      z += (8 * ft_local.R.ppx * ft_local.R.slices + slices - 1) / slices;
      if (kLwRun2Lds > lds) lds = kLwRun2Lds;
      // ... and behind it the second hidden layer's forward of all of the tail's nets (a net of four layers; the flags of
      // the pair launches carry the tail's hand-over)
      ft_local.z1 = -1;
      memset((void*)&ft_local.X, 0, sizeof ft_local.X);
      ft_local.R2 = lw_run2(tail[0].B, tail_n, n_cus > 0 ? n_cus : 256);
      if (second_rode != nullptr && pairs != nullptr && pairs->flags != nullptr && (pairs->use & 4) != 0 && tail[0].net.n_layers == 4 &&
          ft_local.R2.base + (ft_local.R2.rem ? 1 : 0) <= kLw2MaxRun && ft_local.R2.rpn <= kLwFlagStride &&
          tail_n * ft_local.R2.slices * kLwFlagStride <= pairs->n_flags) {
        ft_local.X.flags = pairs->flags; ft_local.X.tag = pairs->next_tag + (unsigned)pairs->used;
        ft_local.X.spin = pairs->spin; ft_local.X.err = pairs->err; ft_local.X.wait_net0 = tail0;
        pairs->used += 1;
        ft_local.z1 = z;
        z += (8 * ft_local.R2.ppx * ft_local.R2.slices + slices - 1) / slices;
        *second_rode = true;
      }
      ft = &ft_local;
    }
    const dim3 heads(slices, 1, z);
    hipLaunchKernelGGL(k_lw_head<512>, heads, blk, lds, st, m, job != nullptr ? *job : TqcJob{},
                       rider != nullptr ? *rider : a[0], n, *ft);
  }
  if (a[0].do_bwd) {
    if (pair_ok && (pairs->use & 2) != 0) {
      pair(1, 2, 1);
    } else {
      for (int l = L - 2; l >= 1; --l) {
        mid(1, l);
      }
    }
    if (a[0].dact_cols > 0) {
      static const PrefetchJob no_job = [] { PrefetchJob j; memset((void*)&j, 0, sizeof j); j.z0 = -1; return j; }();
      PrefetchJob pj = no_job;
      if (prefetch != nullptr && prefetch->B == a[0].B) { pj = *prefetch; pj.z0 = n; }
      int z = n + (pj.z0 >= 0 ? 1 : 0);
      static const LwDactRide no_ride = [] { LwDactRide r; memset((void*)&r, 0, sizeof r); r.z_r = -1; r.z_t = -1; return r; }();
      LwDactRide rd = no_ride;
      static const DwKArgs no_tiles = [] { DwKArgs z; memset((void*)&z, 0, sizeof z); return z; }();
      DwKArgs dk = no_tiles;
      constexpr size_t dact_f = LwLds<512>::total, ride_f = SliceLds<256>::total(2);
      size_t lds = sizeof(float) * dact_f;
      // the consumer of these rows — the actor's backward (a prepared k_mlp_slice_tp launch: tag drawn) — rides behind them
      if (bwd_rider != nullptr && bwd_rode != nullptr && pairs != nullptr && pairs->flags != nullptr && (pairs->use & 8) != 0 &&
          bwd_rider->B == a[0].B && bwd_rider->tp_xbuf != nullptr && mlp_slice_tp_shape_ok(*bwd_rider, 256) && !bwd_rider->do_fwd &&
          bwd_rider->do_bwd && bwd_rider->seed_mode == SEED_GAUSS && bwd_rider->seed.n_da == n && slices * (z + 4) <= (n_cus > 0 ? n_cus : 256) &&
          n * slices <= pairs->n_flags) {
        rd.R = *bwd_rider;
        rd.flags = pairs->flags; rd.tag = pairs->next_tag + (unsigned)pairs->used; rd.fstride = slices; rd.z_r = z;
        pairs->used += 1;
        rd.R.seed.da_flags = rd.flags; rd.R.seed.da_tag = rd.tag; rd.R.seed.da_fstride = slices; rd.R.seed.da_spin = pairs->spin;
        z += 4;
        if (sizeof(float) * ride_f > lds) lds = sizeof(float) * ride_f;
        *bwd_rode = true;
        // ... and behind it its own dW + Adam tiles (+ the temperature's step), gated on the members' flags
        if (bwd_tiles != nullptr && tiles_rode != nullptr && bwd_tile_wgs > 0 && (pairs->use & 16) != 0 && bwd_tiles->B == a[0].B &&
            (n + 4) * slices <= pairs->n_flags) {
          dk = *bwd_tiles;
          unsigned long long* done = pairs->flags + (size_t)n * slices;
          rd.R.done_flags = done; rd.R.done_tag = rd.tag;
          dk.gate.rows = done; dk.gate.n_rows = 4 * slices;
          dk.gate.seed = nullptr; dk.gate.n_seed = 0;
          dk.gate.tag = rd.tag; dk.gate.spin = pairs->spin;
          dk.gate.err = pairs->err; dk.gate.err_code = (KERN_LW_PAIR << 8) | SITE_LW_PAIR;
          rd.z_t = z; rd.n_tile_wgs = bwd_tile_wgs;
          z += (bwd_tile_wgs + slices - 1) / slices;
          if (sizeof(float) * (size_t)kDwLdsFloats > lds) lds = sizeof(float) * (size_t)kDwLdsFloats;
          *tiles_rode = true;
        }
      }
      hipLaunchKernelGGL(k_lw_dact<512>, dim3(slices, 1, z), blk, lds, st, m, pj, rd, dk);
    } else if (prefetch != nullptr) {
      return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

}  // namespace oprl
