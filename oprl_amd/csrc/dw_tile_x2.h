// dw_tile_x2.h — the dW + Adam tile of a PrecX2 learner's MERGED phase launches (fused_ddpg.hip): 16 (n) x 64 (k)
// outputs per workgroup of 1024 threads, the 256-row contraction on the fp16 matrix rate with the split product of
// engine.h (PrecX2), built around what such a tile waits for.
//
// A gated tile (dw_body.h, GATE 1 / 2) has everything early but a few numbers per minibatch row — the critic's
// TD-error seed, the actor's du.  X therefore goes into LDS at once, transposed (minibatch index contiguous) and
// already split into its two fp16 planes; what follows the late numbers is short:
//     form dY (4 elements per lane) -> LDS, transposed          [barrier]
//     wave = (32-row group g, half of the k columns): A = its group's dY rows, scaled by a power of two fixed by the
//       GROUP's largest magnitude and split; two 16 x 16 output tiles x 3 MFMAs; partial tiles -> LDS   [barrier]
//     thread = element: 8 partials, Adam (state requested at entry), Polyak, stores; the new tile -> LDS  [barrier]
//     fp16 packs in pack order
// Twice the tile of dw_body.h (84 tiles per net instead of 152: every tile of a merged launch is resident long before
// its seeds arrive) at a quarter of its matrix time (the exact-fp32 MFMAs of a 16 x 32 tile are 0.43 us per CU).
//   GATE 1 (the critic's tiles on phase 1): dY = U[b, n] * seed[b], U = the unit-seed rows role B wrote (summed over
//       the cluster's partial buffers where there are any), or e_0 for the output layer.
//   GATE 2 (the actor's tiles on phase 2): dY from du, dw_body.h's three kinds.
// Scales: X goes in as 2^4 x (PrecX2::kFwdA); a group's dY as s dY, s = a_scale(its largest |dY|).
// One 256-row chunk (B <= 256); fp32 packs are not written (PrecX2 learners keep them lazily: learner.hip fresh32).
// (A first form of this file — all 16 waves as loaders, TWO waves computing a 16 x 32 tile over the whole contraction,
// Adam from their accumulators — was slower than dw_body.h's: one wave's 8 dependent steps of LDS reads, a 16-
// instruction split and three MFMAs are a 2.5 us latency chain with nothing to hide it behind.  r03 experiment log.)
#pragma once
#include "dw_body.h"

namespace oprl {

constexpr int kDwX2TileK = 64;

// max over the 64 lanes, in every lane: four DPP steps inside the rows of 16 (engine.h row16_sum's), then the four
// rows through scalar registers — six ds_bpermute round trips (~100 cycles each, serial) as __shfl_xor steps
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float ld_ag(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16- / 8-byte stores WRITTEN THROUGH (engine.h st16_agent): what a tile hands to readers of the SAME launch — packs, bias
// copies, the output layer's rows — is acknowledged by memory, not by this XCD's L2, before the tile's flag goes up
template <class V16>
__device__ __forceinline__ void st16_wt(float* base, size_t off, const V16 v) {      // base: wave-uniform
  static_assert(sizeof(V16) == 16, "16-byte vector");
  st16_agent_at(base, (unsigned)off, __builtin_bit_cast(f32x4, v));
}
template <class V8>
__device__ __forceinline__ void st8_wt(void* p, const V8 v) {
  static_assert(sizeof(V8) == 8, "8-byte vector");
  st8_agent(p, __builtin_bit_cast(unsigned long long, v));
}

struct DwX2Lds {   // floats
  // minibatch extent of a transposed row, padded such that the compute lanes' b128 reads (16 rows i, 16 bytes each)
  // fall into 16 different groups of four banks — fp32 rows 260 dwords apart, fp16 rows 264 halfs = 132 dwords
  static constexpr int LDF = 260, LDH = 264, LDT = kDwX2TileK + 4;
  static constexpr int dyt = 0;                          // [16 n][LDF] fp32; later the updated tile: online [16][LDT] | target
  static constexpr int xh = dyt + 16 * LDF;              // [64 k][LDH] fp16: hi plane of 2^4 X
  static constexpr int xl = xh + 64 * LDH / 2;           // ... lo plane
  static constexpr int part = xl + 64 * LDH / 2;         // [8 groups][16][LDT] partial tiles
  static constexpr int bpart = part + 8 * 16 * LDT;      // [8 groups x 4 row quarters][16] partial column sums of dY
  static constexpr int floats = bpart + 512;
};
static_assert(2 * 16 * DwX2Lds::LDT <= 16 * DwX2Lds::LDF, "the staged tiles fit the dY area");

// begin(): the tile's header, its Adam state and (GATE 2) every row that does not depend on this launch; finish(): the
// rest.  Two calls, so that a workgroup that has something else to do first (role U of phase 2, which goes on as a tile
// workgroup) can have its rows in flight meanwhile.
// (`gate` — 1: the critic's tiles, 2: the actor's — is a RUN-TIME member: k_ddpg_update runs both kinds and carries ONE copy
// of this code for them, 10 KB less in a kernel whose speed follows its instruction-cache footprint (r03-25 / -26); in
// the two-launch kernels the value is a constant at the only call site and the other kind's branches fold away)
// A tile of a launch that runs SEVERAL updates (k_ddpg_chain, fused_ddpg.hip): what the host bakes into the argument
// block of a one-update launch changes from update to update there — the gate's tag, Adam's bias-correction terms, the
// output-layer snapshot — and is read from the launch's ChainArgs where it is used (two registers carried through the
// tile instead of a dozen: these kernels sit at the scalar-register limit).  chain == null: a one-update launch.
struct DwX2Ovr {
  const ChainArgs* chain = nullptr;
  int u = 0;                             // the update's index in the launch
};

// P: PrecX2 (the split-fp16 products above) or PrecF32 — the SAME tile with exact-fp32 arithmetic (round 4: the
// exact-fp32 learner's whole-update launch): X transposed into ONE fp32 plane in LDS (the two fp16 planes' bytes), eight
// v_mfma_f32_16x16x4_f32 per 16 x 16 output tile and 32-row group, no scales, the fp32 fragment packs written from the
// staged tile (DwItem::pf / pb / tpf: for such a learner they point at the library's uncached mirrors of the caller's
// packs, learner.hip).
template <class KArgs = DwKArgs, class P = PrecX2>
struct DwX2Tile {
  static constexpr int TK = kDwX2TileK, LDF = DwX2Lds::LDF, LDH = DwX2Lds::LDH, LDT = DwX2Lds::LDT;
  DwX2Ovr ov;
  // (only what must survive between begin() and finish(): the layer's table entry, the tile's coordinates and flags are
  // formed again in finish() — scalar work — instead of being carried through whatever runs in between)
  const KArgs* KA;
  float* lds;
  int item, lt, n_stamp, gate;
  float p_th, p_m, p_v, p_tt, q_th, q_m, q_v, q_tt;
  f32x4 vx[2][2], hmask;

  __device__ __forceinline__ void stamp() {
    long long* const h_trace = KA->trace;
    const int wg = item * 16 + lt;
    if (kTraceOn && h_trace != nullptr && threadIdx.x == 0 && lt < 16 && wg < 64 && n_stamp < kTraceStamps) {
      long long* tr = h_trace + ((size_t)wg * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  }

  __device__ __forceinline__ void begin(const KArgs& A, float* lds_, int bx, int gate_) {
    gate = gate_;
    KA = &A;
    lds = lds_;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int te0 = KA->tile_end[0], te1 = KA->tile_end[1], te2 = KA->tile_end[2], te3 = KA->tile_end[3];
    const int hB = A.B;
    const AdamScalars& ad = A.ad;
    item = (bx >= te0 ? 1 : 0) + (bx >= te1 ? 1 : 0) + (bx >= te2 ? 1 : 0) + (bx >= te3 ? 1 : 0);
    if constexpr (sizeof(KA->tile_end) / sizeof(int) > 4) {          // (twin critics on one launch: up to eight layers)
#pragma unroll
      for (int j = 4; j < 8; ++j) item += bx >= KA->tile_end[j] ? 1 : 0;
    }
    const DwItem I = KA->items[item];
    const DwGate& G = KA->gate;
    lt = bx - (item > 0 ? KA->tile_end[item - 1] : 0);
    n_stamp = 0;
    stamp();
    const int tiles_k = cdiv(I.K, TK);
    const int tn = lt / tiles_k, tk = lt - tn * tiles_k;
    const int n_base = tn * kDwTileN, k_base = tk * TK;
    const bool polyak = ad.do_polyak && I.w_t != nullptr;
    // ---- this thread's element (nl, kl) of the epilogue and its Adam state, requested now
    const int nl = tid >> 6, kl = tid & 63;
    const int en = n_base + nl, ek = k_base + kl;
    const bool e_ok = en < I.N && ek < I.K;
    const size_t eo = (size_t)en * I.K + ek;
    p_th = p_m = p_v = p_tt = 0.f;
    // (masters and moments travel past the L2 — agent-scope loads here, agent-scope stores in finish(): inside a launch
    // that runs several updates a tile's next incarnation sits on another compute unit, possibly another XCD)
    if (e_ok) {
      p_th = ld_ag(I.w + eo); p_m = ld_ag(I.w_m + eo); p_v = ld_ag(I.w_v + eo);
      if (polyak) p_tt = ld_ag(I.w_t + eo);
    }
    const bool b_own = tk == 0 && tid < kDwTileN && n_base + tid < I.N;
    const bool b_pol = ad.do_polyak && I.b_t != nullptr;
    q_th = q_m = q_v = q_tt = 0.f;
    if (b_own) {
      const int n = n_base + tid;
      q_th = ld_ag(I.b + n); q_m = ld_ag(I.b_m + n); q_v = ld_ag(I.b_v + n);
      if (b_pol) q_tt = ld_ag(I.b_t + n);
    }
    // ---- loaders (wave w: minibatch rows 16 w .. 16 w + 15).  X: lane = (row pair p = lane >> 3, quad q = lane & 7) x
    // two column halves — two ADJACENT rows per lane, so that a transposed fp16 pair is one dword
    const int xq = (lane & 7) * 4, xb0 = 16 * wave + 2 * (lane >> 3);
    const int an = (lane & 3) * 4, bb = 16 * wave + (lane >> 2), ncol = n_base + an;
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) vx[h][0] = vx[h][1] = z4;
    hmask = z4;
    if (gate == 2) {
      const int kind = item == 0 ? G.kind[0] : (item == 1 ? G.kind[1] : (item == 2 ? G.kind[2] : G.kind[3]));
      // X is the launch before's: at once
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int xc = k_base + 32 * h + xq;
        if (xc < I.ldx) {
          if (xb0 < hB) vx[h][0] = ld4c(I.X + (size_t)xb0 * I.ldx + xc);      // (coherent loads: role C of this launch may be the writer)
          if (xb0 + 1 < hB) vx[h][1] = ld4c(I.X + (size_t)(xb0 + 1) * I.ldx + xc);
        }
      }
      if (kind == 1 && ncol < I.ldy && bb < hB) hmask = ld4c(G.h2 + (size_t)bb * I.ldy + ncol);
    }
  }

  __device__ __forceinline__ void finish() {
  const KArgs& A = *KA;
  float* dyt = lds + DwX2Lds::dyt;
  _Float16* xh = reinterpret_cast<_Float16*>(lds + DwX2Lds::xh);
  _Float16* xl = reinterpret_cast<_Float16*>(lds + DwX2Lds::xl);
  float* part = lds + DwX2Lds::part;
  float* bpart = lds + DwX2Lds::bpart;
  float* tileW = lds + DwX2Lds::dyt;
  float* tileT = tileW + 16 * LDT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hB = A.B, h_n_part = A.n_part, h_tiled = A.dy_tiled;
  const AdamScalars ad = A.ad;
  const DwGate& G = KA->gate;
  const DwItem I = KA->items[item];
  const int tiles_k = cdiv(I.K, TK);
  const int tn = lt / tiles_k, tk = lt - tn * tiles_k;
  const int n_base = tn * kDwTileN, k_base = tk * TK;
  const bool polyak = ad.do_polyak && I.w_t != nullptr;
  const bool e_ok = n_base + (tid >> 6) < I.N && k_base + (tid & 63) < I.K;
  const size_t eo = (size_t)(n_base + (tid >> 6)) * I.K + k_base + (tid & 63);
  const bool b_own = tk == 0 && tid < kDwTileN && n_base + tid < I.N;
  const bool b_pol = ad.do_polyak && I.b_t != nullptr;
  int kind = 0;
  if (gate == 2) kind = item == 0 ? G.kind[0] : (item == 1 ? G.kind[1] : (item == 2 ? G.kind[2] : G.kind[3]));
  const int ptile = n_base >> 4;
  const int i = lane & 15, kk = lane >> 4;
  // (the host knows the step in the merged launches; k_ddpg_chain: per update)
  const bool chained = ov.chain != nullptr;
  const float step_size = chained ? (gate == 1 ? ov.chain->c_step[ov.u] : ov.chain->a_step[ov.u]) : ad.step_size_host;
  const float bc2_sqrt = chained ? (gate == 1 ? ov.chain->c_bc2[ov.u] : ov.chain->a_bc2[ov.u]) : ad.bc2_sqrt_host;
  const unsigned gtag = G.tag + (unsigned)ov.u;
  const int nl = tid >> 6, kl = tid & 63;
  const int xp = lane >> 3, xq = (lane & 7) * 4;
  const int xb0 = 16 * wave + 2 * xp;
  // dY: lane = (row ar = lane >> 2, column quad an = 4 (lane & 3))
  const int ar = lane >> 2, an = (lane & 3) * 4;
  const int bb = 16 * wave + ar, ncol = n_base + an;
  const bool an_ok = ncol < I.ldy;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 va[kDuLd];
#pragma unroll
  for (int j = 0; j < kDuLd; ++j) va[j] = z4;
  if (gate == 2) {
    if (kind == 3) {
      // unit-seed rows G_j[bb][ncol .. ncol + 3] of the critic pass's members (uncached memory): their flags, an L1
      // then the rows (coherent loads) — all long before du
      {
        const int k = tid;
        if (k < G.n_gu_flags) {
          bool ok = false;
          for (int spin = 0; spin < G.spin && !ok; ++spin) {
            ok = (unsigned)(__hip_atomic_load(G.gu_flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == gtag;
            if (!ok) __builtin_amdgcn_s_sleep(2);
          }
          if (!ok) report_expired(G.err, G.err_code);
        }
        __syncthreads();
      }
      if (an_ok && bb < hB) {
#pragma unroll
        for (int j = 0; j < kDuLd; ++j)
          if (j < G.n_act) va[j] = ld4c(G.gu + ((size_t)j * hB + bb) * I.ldy + ncol);
      }
    }
    // the output layer's rows W3[j][ncol .. ncol + 3] (a few hundred bytes, shared by every tile of the layer: not
    // worth eight registers per lane through role U)
    if (kind == 1 && an_ok) {
#pragma unroll
      for (int j = 0; j < kDuLd; ++j)
        if (j < G.n_act) va[j] = ld4c((chained ? ov.chain->w3buf[ov.u & 1] : G.w3) + (size_t)j * I.ldy + ncol);
    }
  }
  // first attempts at what the tile waits for, requested with the rows
  // (kind 2: the lane's four dz1 granules of the critic pass's backward step, in g[0 .. 3])
  unsigned long long g[kDuLd];
  const unsigned long long* gsrc = nullptr;
  int n_g = 0;
  if (gate == 2) {
    const bool row_in = bb < hB;
    if (kind == 2) {
      gsrc = G.g1 + ((size_t)(ncol >> 4) * hB + (row_in ? bb : 0)) * 16 + (ncol & 15);
      n_g = (row_in && an_ok) ? 4 : 0;
    } else {
      gsrc = G.seed + (size_t)(row_in ? bb : 0) * kDuLd;
      n_g = row_in ? G.n_act : 0;
    }
#pragma unroll
    for (int j = 0; j < kDuLd; ++j)
      g[j] = j < n_g ? __hip_atomic_load(gsrc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)gtag << 32);
  }
  // (GATE 2 waits for granules only: what the critic pass reads of the actor's packs it has taken in BEFORE it publishes
  // du, and no tile stores before it has du)
  if (gate == 1) {
    const unsigned long long* myf = G.rows + (tid < G.n_rows ? tid : 0);
    bool ok = (unsigned)(__hip_atomic_load(myf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == gtag || G.what_if == 106;
    for (int spin = 0; spin < G.spin && !ok; ++spin) {
      __builtin_amdgcn_s_sleep(2);
      ok = (unsigned)(__hip_atomic_load(myf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == gtag;
    }
    if (!ok) report_expired(G.err, G.err_code);
  }
  f32x4 u = z4;          // GATE 1: the unit-seed dY of this lane
  unsigned long long seed0 = 0ull;
  if (gate == 1) {
    __syncthreads();     // role B's members have flagged their rows (written through): X and the unit-seed dY
    // The write-through rows are read past this XCD's L2 (sc1) with raw buffer loads (engine.h ld4_agent: loads the
    // compiler counts; round 3's first form — the same instruction as inline asm + an explicit wait — left the result
    // registers open to being copied before the data had arrived); from clamped addresses, unconditionally.
    const bool late = I.dY == G.late_dY || (G.late_dY2 != nullptr && I.dY == G.late_dY2);          // the output layer: dY IS the seed (one column): U = e_0
    const int npart = I.dY_part_stride > 0 ? h_n_part : 1;
    const bool tiled = I.dY_part_stride > 0 && h_tiled != 0;
    const int xr0 = xb0 < hB ? xb0 : hB - 1, xr1 = xb0 + 1 < hB ? xb0 + 1 : hB - 1;
    const bool c0 = k_base + xq < I.ldx, c1 = k_base + 32 + xq < I.ldx;
    const int xc0 = c0 ? k_base + xq : 0, xc1 = c1 ? k_base + 32 + xq : 0;
    const int ub = bb < hB ? bb : hB - 1, uc = (an_ok && !late) ? ncol : 0;
    const unsigned uoff = tiled ? (unsigned)(((uc >> 4) * hB + ub) * 16 + (uc & 15)) : (unsigned)(ub * I.ldy + uc);
    const unsigned ps = (unsigned)I.dY_part_stride;
    const f32x4 r00 = ld4_agent(I.X, (unsigned)(xr0 * I.ldx + xc0)), r01 = ld4_agent(I.X, (unsigned)(xr1 * I.ldx + xc0));
    const f32x4 r10 = ld4_agent(I.X, (unsigned)(xr0 * I.ldx + xc1)), r11 = ld4_agent(I.X, (unsigned)(xr1 * I.ldx + xc1));
    const f32x4 pa0 = ld4_agent(I.dY, uoff), pa1 = ld4_agent(I.dY, uoff + (npart > 1 ? ps : 0u)),
                pa2 = ld4_agent(I.dY, uoff + (npart > 2 ? 2u * ps : 0u)), pa3 = ld4_agent(I.dY, uoff + (npart > 3 ? 3u * ps : 0u));
    vx[0][0] = (c0 && xb0 < hB) ? r00 : z4;
    vx[0][1] = (c0 && xb0 + 1 < hB) ? r01 : z4;
    vx[1][0] = (c1 && xb0 < hB) ? r10 : z4;
    vx[1][1] = (c1 && xb0 + 1 < hB) ? r11 : z4;
    // (a first look at this lane's seed, requested WITH the rows: a tile that starts late — on a workgroup whose role ended
    // late — finds its seeds published already, and the look costs it no round trip of its own behind the staging; the
    // granule is its own flag: r05-18)
    if (bb < hB && G.n_seed > 0)
      seed0 = __hip_atomic_load((item >= G.item_split ? G.seed2 : G.seed) + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u = ((pa0 + (npart > 1 ? pa1 : z4)) + (npart > 2 ? pa2 : z4)) + (npart > 3 ? pa3 : z4);   // member order, as k_dw_adam sums them
    if (late) u = f32x4{ncol == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f};
    if (!(bb < hB && (an_ok || late))) u = z4;
  }
  float* xt = lds + DwX2Lds::xh;       // PrecF32: ONE fp32 plane [64 k][LDF] over the two fp16 planes' area
  static_assert(64 * DwX2Lds::LDF <= 64 * DwX2Lds::LDH, "the fp32 plane fits the two fp16 planes");
  if constexpr (!P::kX2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool k_in = k_base + 32 * h + xq + t < I.K;
        *reinterpret_cast<f32x2*>(xt + (size_t)(32 * h + xq + t) * LDF + xb0) = f32x2{k_in ? vx[h][0][t] : 0.f, k_in ? vx[h][1][t] : 0.f};
      }
    }
  }
  // X -> 2^4 X -> two fp16 planes, transposed: plane[k][b], the two rows of this lane side by side
#pragma unroll
  for (int h = 0; P::kX2 && h < 2; ++h) {
    f32x4 a = vx[h][0] * PrecX2::kFwdA, b = vx[h][1] * PrecX2::kFwdA;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool k_in = k_base + 32 * h + xq + t < I.K;
      a[t] = k_in ? a[t] : 0.f;
      b[t] = k_in ? b[t] : 0.f;
    }
    f16x8 hi, lo;
    x2_split8(a, b, hi, lo);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f16x2*>(xh + (size_t)(32 * h + xq + t) * LDH + xb0) = f16x2{hi[t], hi[4 + t]};
      *reinterpret_cast<f16x2*>(xl + (size_t)(32 * h + xq + t) * LDH + xb0) = f16x2{lo[t], lo[4 + t]};
    }
  }
  // ---- the late numbers: per-row seeds (GATE 1) / du (GATE 2), then this lane's four dY elements
  f32x4 v = z4;
  if (gate == 1) {
    float sd = 0.f;
    if (bb < hB && G.n_seed > 0) {
      unsigned long long x = seed0;
      bool ok = G.what_if == 104 || (G.spin > 0 && (unsigned)(x >> 32) == gtag);
      for (int spin = 0; spin < G.spin && !ok; ++spin) {
        x = __hip_atomic_load((item >= G.item_split ? G.seed2 : G.seed) + bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = (unsigned)(x >> 32) == gtag;
        if (!ok) __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) report_expired(G.err, G.err_code);
      sd = ok ? __uint_as_float((unsigned)x) : __builtin_nanf("");
    }
    stamp();   // seeds in
    v = u * sd;
  } else {
    float du[kDuLd];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kDuLd; ++j) ok = ok && (unsigned)(g[j] >> 32) == gtag;
    if (G.what_if == 105) ok = true;
    for (int spin = 0; spin < G.spin && !ok; ++spin) {
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < kDuLd; ++j)
        if (j < n_g) g[j] = __hip_atomic_load(gsrc + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int j = 0; j < kDuLd; ++j) ok = ok && (unsigned)(g[j] >> 32) == gtag;
    }
    if (!ok) report_expired(G.err, G.err_code);
#pragma unroll
    for (int j = 0; j < kDuLd; ++j)
      du[j] = j < n_g ? (ok ? __uint_as_float((unsigned)g[j]) : __builtin_nanf("")) : 0.f;
    stamp();   // rows and seeds in
    if (kind == 2) {
      v = f32x4{du[0], du[1], du[2], du[3]};
    } else if (kind == 1) {
#pragma unroll
      for (int j = 0; j < kDuLd; ++j) v += va[j] * du[j];
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = hmask[t] > 0.f ? v[t] : 0.f;
    } else if (kind == 3) {
#pragma unroll
      for (int j = 0; j < kDuLd; ++j) v += va[j] * du[j];      // (the masks are in the unit-seed rows)
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < kDuLd; ++j) x = (ncol + t == j) ? du[j] : x;
        v[t] = x;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) dyt[(an + t) * LDF + bb] = (ncol + t < I.N) ? v[t] : 0.f;
  __syncthreads();       // dY and both planes of X are in LDS

  // ---- wave = (32-row group g, half of the k columns): two output tiles over the group's rows
  if constexpr (!P::kX2) {
    // exact fp32: per output tile 2 x (one b128 of dY^T, one of X^T, four v_mfma_f32_16x16x4_f32) — the lane group kk of a
    // quad takes minibatch rows 4 kk + t of its 16-row half in both operands (any order inside a macro step, engine.h)
    const int grp = wave >> 1, half = wave & 1;
    const float* arow = dyt + (size_t)i * LDF + 32 * grp + 4 * kk;
    const f32x4 a0 = ld4(arow), a1 = ld4(arow + 16);
    if (half == 0) bpart[(grp * 4 + kk) * 16 + i] = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* brow = xt + (size_t)(32 * half + 16 * t + i) * LDF + 32 * grp + 4 * kk;
      const f32x4 b0 = ld4(brow), b1 = ld4(brow + 16);
      f32x4 acc = z4;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma4(a0[e], b0[e], acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma4(a1[e], b1[e], acc);
      float* o = part + ((size_t)grp * 16 + 4 * kk) * LDT + 32 * half + 16 * t + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * LDT] = acc[r];
    }
  } else {
    const int grp = wave >> 1, half = wave & 1;
    const float* arow = dyt + (size_t)i * LDF + 32 * grp + 8 * kk;
    const f32x4 a0 = ld4(arow), a1 = ld4(arow + 4);
    // the group's largest |dY| fixes its scale (both waves of the group find the same value)
    float m = fmaxf(fmaxf(fmaxf(fabsf(a0[0]), fabsf(a0[1])), fmaxf(fabsf(a0[2]), fabsf(a0[3]))),
                    fmaxf(fmaxf(fabsf(a1[0]), fabsf(a1[1])), fmaxf(fabsf(a1[2]), fabsf(a1[3]))));
    m = wave_max(m);
    const float sa = PrecX2::a_scale(m);
    // column n = i over this lane's eight rows; the four row quarters kk are summed by the bias's thread below
    if (half == 0) bpart[(grp * 4 + kk) * 16 + i] = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
    f16x8 ah, al;
    x2_split8(a0 * sa, a1 * sa, ah, al);
    const float un = 1.f / (sa * PrecX2::kFwdA);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const size_t brow = (size_t)(32 * half + 16 * t + i) * LDH + 32 * grp + 8 * kk;
      const f16x8 xhi = *reinterpret_cast<const f16x8*>(xh + brow), xlo = *reinterpret_cast<const f16x8*>(xl + brow);
      f32x4 acc = z4;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xhi, acc, 0, 0, 0);
      float* o = part + ((size_t)grp * 16 + 4 * kk) * LDT + 32 * half + 16 * t + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * LDT] = acc[r] * un;
    }
  }
  stamp();   // MFMAs done, partial tiles in LDS
  __syncthreads();

  // ---- thread = element: the eight groups' partials in group order, Adam (torch.optim.Adam single-tensor semantics)
  // and Polyak as dw_adam_body's epilogue; the updated tile goes through LDS for the packs
  // (a tile with a completion flag — k_ddpg_update's critic tiles, which the critic pass of the same launch waits for —
  // writes what that pass reads FIRST: the online packs and the bias copy, then the flag, then the masters, the moments
  // and the target's packs)
  const bool flagged = G.done != nullptr;
  const int bx_ = (item > 0 ? KA->tile_end[item - 1] : 0) + lt;
  float e_m = 0.f, e_v = 0.f;
  float gsum = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) gsum += part[((size_t)q * 16 + nl) * LDT + kl];
  // ---- data parallel over peer windows (csrc/p2p.hip; dw_body.h's XCHG for this tile): the tile's 1024 gradient
  // elements (+ the 16 bias sums of a k-block-0 tile) are all-reduced with the same tile of the other ranks BEFORE Adam —
  // 8-byte {sequence, value} granules written at system scope into the tile's slot of the OWNER's window (tile % world),
  // who sums the partial tiles in rank order and sends the sum back: every replica applies the very same bits, and a
  // data-parallel update is the single-GPU launch + these two hops per tile, no all-reduce or apply launches.
  float gb_x = 0.f;
  bool xchg_on = false;
  if (A.xchg.world > 1) {
    const DwXchg& X = A.xchg;
    xchg_on = true;
    float gbw = 0.f;
    if (b_own) {
#pragma unroll
      for (int q = 0; q < 32; ++q) gbw += bpart[q * 16 + tid];
    }
    const unsigned long long seq = X.seq + (chained ? 2ull * (unsigned long long)ov.u : 0ull) + (unsigned long long)(gate == 2 ? 1 : 0);
    const unsigned tag = (unsigned)seq;
    const int parity = (int)(seq & 1ull);
    const int owner = (int)((unsigned)bx_ % (unsigned)X.world);
    auto slot = [&](char* base, int src_slot) {
      return reinterpret_cast<unsigned long long*>(base) +
             (((size_t)parity * (X.world + 1) + src_slot) * X.max_tiles + bx_) * kDwXchgTile;
    };
    auto put = [&](unsigned long long* dst, float v, float vb) {
      __hip_atomic_store(dst + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (tid < kDwTileN)
        __hip_atomic_store(dst + 1024 + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(vb),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto get = [&](const unsigned long long* src, float* v, float* vb) {
      unsigned long long x = 0, xb = 0;
      bool ok = false;
      for (int spin = 0; spin < (1 << 20) && !ok; ++spin) {
        x = __hip_atomic_load(src + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        xb = tid < kDwTileN ? __hip_atomic_load(src + 1024 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : x;
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
        float v = gsum, vb = gbw;
        if (r != X.rank) all_ok = get(slot(X.window, r), &v, &vb) && all_ok;
        gs += v;
        gbs += vb;
      }
      if (!all_ok) { gs = __builtin_nanf(""); gbs = gs; }   // (the poison travels to every replica)
      for (int p = 0; p < X.world; ++p)
        if (p != X.rank) put(slot(X.peer[p], X.world), gs, gbs);
    } else {
      put(slot(X.peer[owner], X.rank), gsum, gbw);
      all_ok = get(slot(X.window, X.world), &gs, &gbs);
    }
    if (!all_ok) {   // bounded wait: a lost rank is reported and poisons the tile instead of hanging
      report_expired(X.err, (KERN_DW_XCHG << 8) | SITE_DW_TILE);
      gs = __builtin_nanf(""); gbs = gs;
    }
    gsum = gs;
    gb_x = gbs;
  }
  gsum *= ad.grad_scale;
  if (!ad.do_adam) {
    // a gradient-exporting (data-parallel) learner: the tile leaves dW / db in the gradient arena for the all-reduce;
    // Adam, Polyak and the packs are the apply launch's (k_dw_adam, apply_only)
    if (e_ok && I.w_g != nullptr) I.w_g[eo] = gsum;
    if (b_own && I.b_g != nullptr) {
      float gb = 0.f;
#pragma unroll
      for (int q = 0; q < 32; ++q) gb += bpart[q * 16 + tid];
      I.b_g[n_base + tid] = gb * ad.grad_scale;
    }
    return;
  }
  float th_new = 0.f, tt_new = 0.f;
  if (e_ok) {
    float mm = p_m, vv = p_v, th = p_th;
    adam_elem(gsum, mm, vv, th, ad, step_size, bc2_sqrt);
    th_new = th;
    if (polyak) tt_new = polyak_elem(p_tt, th, ad);
    e_m = mm; e_v = vv;
    if (!flagged) {
      st_ag(I.w_m + eo, mm);
      st_ag(I.w_v + eo, vv);
      st_ag(I.w + eo, th);
      if (polyak) st_ag(I.w_t + eo, tt_new);
    }
    // (the actor's output layer, row-major, for whoever reads it in the next update of this launch)
    if (chained && gate == 2 && kind == 0) st_ag(ov.chain->w3buf[(ov.u + 1) & 1] + eo, th);
  }
  tileW[nl * LDT + kl] = th_new;      // (the dY area: every wave is past its MFMAs)
  tileT[nl * LDT + kl] = tt_new;
  if (b_own) {
    const int n = n_base + tid;
    float gb = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) gb += bpart[q * 16 + tid];
    if (xchg_on) gb = gb_x;          // (the all-reduced bias sums)
    gb *= ad.grad_scale;
    float mm = q_m, vv = q_v, th = q_th;
    adam_elem(gb, mm, vv, th, ad, step_size, bc2_sqrt);
    if (I.b16 != nullptr) st_ag(I.b16 + n, th);      // (uncached copy for readers inside the same launch)
    st_ag(I.b_m + n, mm);
    st_ag(I.b_v + n, vv);
    st_ag(I.b + n, th);
    if (b_pol) {
      const float tb = polyak_elem(q_tt, th, ad);
      st_ag(I.b_t + n, tb);
      if (I.bt16 != nullptr) st_ag(I.bt16 + n, tb);
    }
  }
  __syncthreads();       // the updated tile is staged

  // ---- the two-plane fp16 packs (hi = fp16(2^8 w), lo = fp16(2^8 w - hi); the lo plane 256 floats behind the hi
  // plane of a block), in pack order: threads 0..255 the forward packs (online | target) x two 32-column macro
  // steps, threads 256..511 the W^T pack's four 16-row k tiles (this tile's n columns are one HALF of their block)
  if constexpr (!P::kX2) {
    // ---- the fp32 fragment packs (engine.h: pack[((tile NS + s) 64 + lane) 4 + t] = M[16 tile + (lane & 15)][16 s + 4 (lane >> 4) + t])
    // in pack order, 16-byte stores: threads 0..255 the forward pack's four 16-column macro steps of this tile's n tile,
    // 256..511 the target's, 512..767 the W^T pack (tiles over k: this tile's four; its 16 n are ONE macro step there)
    if (I.pf != nullptr) {
      const int NSk = cdiv(I.K, 16), NSn = cdiv(I.N, 16);
      const int grp3 = tid >> 8, q = tid & 255, j = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
      if (grp3 < 2) {
        float* dst = grp3 == 0 ? I.pf : (polyak ? I.tpf : nullptr);
        if (dst != nullptr && 4 * tk + j < NSk && !(flagged && grp3 == 1))
          st16_wt(dst, (((size_t)ptile * NSk + 4 * tk + j) * 64 + l) * 4, ld4((grp3 == 0 ? tileW : tileT) + li * LDT + 16 * j + 4 * lk));
      } else if (grp3 == 2 && I.pb != nullptr) {
        const int ktile = 4 * tk + j;
        if (16 * ktile < I.K) {
          f32x4 w4;
#pragma unroll
          for (int t = 0; t < 4; ++t) w4[t] = tileW[(4 * lk + t) * LDT + 16 * j + li];
          st16_wt(I.pb, (((size_t)ktile * NSn + ptile) * 64 + l) * 4, w4);
        }
      }
    }
  }
  if (!P::kX2 && I.pf16 != nullptr && !I.x2) {
    // ---- a bf16 learner's tile (exact-fp32 product above; merged launches): its bf16 packs as well — blocks of 64 lanes x
    // 8 bf16 in pack16_index order: threads 0..255 the forward packs (online | target) x the tile's two 32-column steps,
    // 256..511 the W^T pack's four 16-row k tiles (this tile's 16 n are one HALF of a 32-wide step there)
    const int NSk2 = cdiv(I.K, 32), NSn2 = cdiv(I.N, 32);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if (tid < 256) {
      const int which = tid >> 7, hb = (tid >> 6) & 1, l = tid & 63, li = l & 15, lk = l >> 4;
      float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
      if (dst != nullptr && 2 * tk + hb < NSk2) {
        const float* src = (which == 0 ? tileW : tileT) + li * LDT + 32 * hb;
        st16_wt(dst, (((size_t)ptile * NSk2 + 2 * tk + hb) * 64 + l) * 4, cvt_bf16x8(ld4(src + 4 * lk), ld4(src + 16 + 4 * lk)));
      }
    } else if (tid < 512 && I.pb16 != nullptr) {
      const int q = tid - 256, blk = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
      const int ktile = 4 * tk + blk;
      if (16 * ktile < I.K) {
        f32x4 w4;
#pragma unroll
        for (int t = 0; t < 4; ++t) w4[t] = tileW[(4 * lk + t) * LDT + 16 * blk + li];
        st8_wt(I.pb16 + (((size_t)ktile * NSn2 + (n_base >> 5)) * 64 + l) * 4 + 2 * ((n_base >> 4) & 1), __builtin_convertvector(w4, bf16x4));
      }
    }
  }
  if (P::kX2 && I.pf16 != nullptr) {
    const int NSk2 = cdiv(I.K, 32), NSn2 = cdiv(I.N, 32);
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    if (tid < 256) {
      const int which = tid >> 7, hb = (tid >> 6) & 1, l = tid & 63, li = l & 15, lk = l >> 4;
      float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
      if (dst != nullptr && 2 * tk + hb < NSk2 && !(flagged && which == 1)) {
        const float* src = (which == 0 ? tileW : tileT) + li * LDT + 32 * hb;
        f16x8 hi, lo;
        x2_split8(ld4(src + 4 * lk) * PrecX2::kWScale, ld4(src + 16 + 4 * lk) * PrecX2::kWScale, hi, lo);
        const size_t d = ((size_t)ptile * NSk2 + 2 * tk + hb) * 512 + (size_t)l * 4;
        st16_wt(dst, d, hi);
        st16_wt(dst, d + 256, lo);
      }
    } else if (tid < 512 && I.pb16 != nullptr) {
      const int q = tid - 256, blk = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
      const int ktile = 4 * tk + blk;
      if (16 * ktile < I.K) {
        f32x4 w4;
#pragma unroll
        for (int t = 0; t < 4; ++t) w4[t] = tileW[(4 * lk + t) * LDT + 16 * blk + li];
        const f32x4 vs = w4 * PrecX2::kWScale;
        const f16x4 hi = __builtin_convertvector(vs, f16x4);
        const f32x4 r = vs - __builtin_convertvector(hi, f32x4);
        float* d = I.pb16 + ((size_t)ktile * NSn2 + (n_base >> 5)) * 512 + (size_t)l * 4 + 2 * ((n_base >> 4) & 1);
        st8_wt(d, hi);
        st8_wt(d + 256, __builtin_convertvector(r, f16x4));
      }
    }
  }
  if (flagged) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(G.done + bx_, (unsigned long long)gtag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e_ok) {
      st_ag(I.w_m + eo, e_m);
      st_ag(I.w_v + eo, e_v);
      st_ag(I.w + eo, th_new);
      if (polyak) st_ag(I.w_t + eo, tt_new);
    }
    if constexpr (!P::kX2) {       // (the target's forward pack: after the flag)
      if (I.pf != nullptr && polyak && I.tpf != nullptr && tid >= 256 && tid < 512) {
        const int NSk = cdiv(I.K, 16);
        const int q = tid & 255, j = q >> 6, l = q & 63, li = l & 15, lk = l >> 4;
        if (4 * tk + j < NSk)
          st16_wt(I.tpf, (((size_t)ptile * NSk + 4 * tk + j) * 64 + l) * 4, ld4(tileT + li * LDT + 16 * j + 4 * lk));
      }
    }
    if (P::kX2 && I.pf16 != nullptr && polyak && I.tpf16 != nullptr && tid >= 128 && tid < 256) {
      const int NSk2 = cdiv(I.K, 32);
      const int hb = (tid >> 6) & 1, l = tid & 63, li = l & 15, lk = l >> 4;
      if (2 * tk + hb < NSk2) {
        const float* src = tileT + li * LDT + 32 * hb;
        f16x8 hi, lo;
        x2_split8(ld4(src + 4 * lk) * PrecX2::kWScale, ld4(src + 16 + 4 * lk) * PrecX2::kWScale, hi, lo);
        const size_t d = ((size_t)ptile * NSk2 + 2 * tk + hb) * 512 + (size_t)l * 4;
        st16_wt(I.tpf16, d, hi);
        st16_wt(I.tpf16, d + 256, lo);
      }
    }
  }
  stamp();   // stores issued
  if (chained) {
    // (a launch that runs several updates: the next update's roles and this tile's next incarnation read what this
    // tile has written — masters, moments, every pack — after THIS flag)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store((gate == 1 ? ov.chain->ct_fin : ov.chain->at_fin) + bx_, (unsigned long long)gtag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  }
};

template <class KArgs = DwKArgs, class P = PrecX2>
__device__ __forceinline__ void dw_tile_x2(const KArgs& A, float* lds, int bx, int gate, const DwX2Ovr& ov = DwX2Ovr()) {
  DwX2Tile<KArgs, P> T;
  T.ov = ov;
  T.begin(A, lds, bx, gate);
  T.finish();
}

}  // namespace oprl
