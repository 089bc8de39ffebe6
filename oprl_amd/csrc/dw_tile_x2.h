// dw_tile_x2.h — the dW + Adam tile of a PrecX2 learner's MERGED phase launches (fused_ddpg.hip), rebuilt around what
// such a tile waits for.
//
// A gated tile (dw_body.h, GATE 1 / 2) has everything early but a few numbers per minibatch row — the critic's TD-error
// seed, the actor's du — and what followed their arrival WAS the update's tail twice over: staging rows through
// wave-private LDS, 16 MFMAs per wave on partial tiles, a barrier that waited for the slowest of 8 / 16 waves, a
// 16-way partial-sum reduction, Adam, two more barriers around the pack staging: 2.9 us (phase 1) and 4.4 us (phase
// 2) of a 32 us update.  Here the 16 waves are LOADERS — they put the tile's operands into LDS TRANSPOSED (minibatch
// index contiguous), X already split into its two fp16 planes — and TWO waves compute: one 16 x 16 output tile each
// over the whole 256-row contraction with the split product of engine.h (8 macro steps of 32 rows, three
// v_mfma_f32_16x16x32_f16 each), the result in their registers: no partial tiles, no reduction, Adam straight from
// the accumulators with state those lanes requested at entry.  After the seeds: poll, 8 steps, Adam, stores, one
// barrier, packs.
//   GATE 1 (the critic's tiles on phase 1): dY = U[b, n] * seed[b], U = the unit-seed rows role B wrote (summed over
//       the cluster's partial buffers where there are any) or e_0 for the output layer; the compute waves poll the
//       256 seed granules themselves and scale their A operand on the way in.
//   GATE 2 (the actor's tiles on phase 2): dY is formed by the loaders from du (dw_body.h's three kinds), written
//       transposed, one more barrier.
// Scales: X goes in as 2^4 x (forward activations, PrecX2::kFwdA); the dY tile as s dY with s = a_scale of its
// largest magnitude (GATE 1: max|U| max|seed|; GATE 2: the loaders' max) — the accumulators come out as 16 s dW.
// One 256-row chunk (B <= 256), 16-row n tiles.
#pragma once
#include "dw_body.h"

namespace oprl {

struct DwX2Lds {   // floats
  // minibatch extent of a transposed row: 256 + padding such that the compute lanes' b128 reads (16 rows i, 16 bytes
  // each) fall into 16 different groups of four banks — fp32 rows 260 dwords apart, fp16 rows 264 halfs = 132 dwords
  static constexpr int LDF = 260, LDH = 264;
  static constexpr int dyt = 0;                          // [16 n][LDF] fp32
  static constexpr int xh = dyt + 16 * LDF;              // [32 k][LDH] fp16: hi plane of 2^4 X
  static constexpr int xl = xh + 32 * LDH / 2;           // ... lo plane
  static constexpr int seed = xl + 32 * LDH / 2;         // [2][256] the compute waves' seeds (GATE 1)
  static constexpr int tw = seed + 512;                  // [16][36] updated tile, online
  static constexpr int tt = tw + 16 * (kDwTile + 4);     // ... target
  static constexpr int misc = tt + 16 * (kDwTile + 4);   // [16] the loader waves' max|dY|
  static constexpr int floats = misc + 16;
};

template <int GATE>
__device__ __forceinline__ void dw_tile_x2(const DwKArgs& A, float* lds, int bx) {
  static_assert(GATE == 1 || GATE == 2, "the gated tiles of the merged phase launches");
  constexpr int TK = kDwTile, LD = TK + 4, LDF = DwX2Lds::LDF, LDH = DwX2Lds::LDH;
  float* dyt = lds + DwX2Lds::dyt;
  _Float16* xh = reinterpret_cast<_Float16*>(lds + DwX2Lds::xh);
  _Float16* xl = reinterpret_cast<_Float16*>(lds + DwX2Lds::xl);
  float (*tileW)[LD] = reinterpret_cast<float (*)[LD]>(lds + DwX2Lds::tw);
  float (*tileT)[LD] = reinterpret_cast<float (*)[LD]>(lds + DwX2Lds::tt);
  float* amaxw = lds + DwX2Lds::misc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const DwKArgs* KA = &A;
  const int te0 = KA->tile_end[0], te1 = KA->tile_end[1], te2 = KA->tile_end[2], te3 = KA->tile_end[3];
  const int hB = A.B, h_n_part = A.n_part, h_tiled = A.dy_tiled;
  long long* const h_trace = A.trace;
  const AdamScalars ad = A.ad;
  asm volatile("" :: "s"(hB), "s"(h_n_part), "s"(h_tiled), "s"(h_trace), "s"(ad.do_polyak), "s"(ad.omb1), "s"(ad.beta2),
               "s"(ad.omb2), "s"(ad.eps), "s"(ad.omtau), "s"(ad.tau), "s"(ad.grad_scale), "s"(ad.step_size_host),
               "s"(ad.bc2_sqrt_host));
  const int item = (bx >= te0 ? 1 : 0) + (bx >= te1 ? 1 : 0) + (bx >= te2 ? 1 : 0) + (bx >= te3 ? 1 : 0);   // (<= 4 layers)
  const DwItem I = KA->items[item];
  const DwGate& G = KA->gate;
  const int lt = bx - (item > 0 ? KA->tile_end[item - 1] : 0);
  int n_stamp = 0;
  auto stamp = [&]() {
    const int wg = item * 16 + lt;
    if (h_trace != nullptr && tid == 0 && lt < 16 && wg < 64 && n_stamp < kTraceStamps) {
      long long* tr = h_trace + ((size_t)wg * kTraceStamps + n_stamp) * 2;
      tr[0] = (long long)__builtin_readcyclecounter();
      tr[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();
#if defined(DW_X2_DEBUG) && DW_X2_DEBUG == 1
  if (GATE == 1) return;
#endif
  const int tn = lt / I.tiles_k, tk = lt - tn * I.tiles_k;
  const int n_base = tn * kDwTileN, k_base = tk * TK;
  const int ptile = n_base >> 4;
  const int i = lane & 15, kk = lane >> 4;
  const int NSk = cdiv(I.K, 16), NSn = cdiv(I.N, 16);
  const bool polyak = ad.do_polyak && I.w_t != nullptr;
  const float step_size = ad.step_size_host, bc2_sqrt = ad.bc2_sqrt_host;   // (the host knows the step in the merged launches)

  // ---- the two compute waves: lane (kk, i) of wave w owns dW[n_base + 4 kk + r][k_base + 16 w + i], r = 0..3; their
  // Adam state is requested once the loaders' own rows have left the registers (below: `request_state`)
  const bool cw = wave < 2;
  const int ek = k_base + 16 * wave + i;
  float p_th[4], p_m[4], p_v[4], p_tt[4];
  bool e_ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e_ok[r] = cw && n_base + 4 * kk + r < I.N && ek < I.K;
    p_th[r] = p_m[r] = p_v[r] = p_tt[r] = 0.f;
  }
  // the bias element: wave 0, lanes kk == 0 (column n = i), of the tiles that own one
  const bool b_own = tk == 0 && wave == 0 && kk == 0 && n_base + i < I.N;
  const bool b_pol = ad.do_polyak && I.b_t != nullptr;
  float q_th = 0.f, q_m = 0.f, q_v = 0.f, q_tt = 0.f;
  auto request_state = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e_ok[r]) {
        const size_t eo = (size_t)(n_base + 4 * kk + r) * I.K + ek;
        p_th[r] = I.w[eo]; p_m[r] = I.w_m[eo]; p_v[r] = I.w_v[eo];
        if (polyak) p_tt[r] = I.w_t[eo];
      }
    }
    if (b_own) {
      const int n = n_base + i;
      q_th = I.b[n]; q_m = I.b_m[n]; q_v = I.b_v[n];
      if (b_pol) q_tt = I.b_t[n];
    }
  };

  // ---- loaders.  X: lane = (row pair p = lane >> 3, quad q = lane & 7) of the wave's 16 rows — two ADJACENT rows, so
  // that a transposed fp16 pair is one dword
  const int xp = lane >> 3, xq = (lane & 7) * 4;
  const int xb0 = 16 * wave + 2 * xp;
  const bool xk_ok = k_base + xq < I.ldx;
  f32x4 vx0 = f32x4{0.f, 0.f, 0.f, 0.f}, vx1 = vx0;
  // dY ingredients: lane = (row ar = lane >> 2, column quad an = 4 (lane & 3))
  const int ar = lane >> 2, an = (lane & 3) * 4;
  const int bb = 16 * wave + ar, ncol = n_base + an;
  const bool an_ok = ncol < I.ldy;
  f32x4 va[kDuLd];
#pragma unroll
  for (int j = 0; j < kDuLd; ++j) va[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 hmask = f32x4{0.f, 0.f, 0.f, 0.f};
  int kind = 0;
  if constexpr (GATE == 2) {
    kind = item == 0 ? G.kind[0] : (item == 1 ? G.kind[1] : (item == 2 ? G.kind[2] : G.kind[3]));
    // X is the launch before's: at once
    if (xk_ok) {
      if (xb0 < hB) vx0 = ld4(I.X + (size_t)xb0 * I.ldx + k_base + xq);
      if (xb0 + 1 < hB) vx1 = ld4(I.X + (size_t)(xb0 + 1) * I.ldx + k_base + xq);
    }
    if (kind == 1 && an_ok) {
      if (bb < hB) hmask = ld4(G.h2 + (size_t)bb * I.ldy + ncol);
#pragma unroll
      for (int j = 0; j < kDuLd; ++j)
        if (j < G.n_act) va[j] = ld4(G.w3 + (size_t)j * I.ldy + ncol);
    }
  }
  // first attempts at what the tile waits for, requested with the rows
  unsigned long long g[kDuLd];
  if constexpr (GATE == 2) {
#pragma unroll
    for (int j = 0; j < kDuLd; ++j)
      g[j] = (j < G.n_act && bb < hB) ? __hip_atomic_load(G.seed + (size_t)bb * kDuLd + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : ((unsigned long long)G.tag << 32);
  }
  {
    const unsigned long long* fl = (GATE == 2 && kind != 2) ? G.read : G.rows;
    const int nfl = (GATE == 2 && kind != 2) ? G.n_read : G.n_rows;
    const unsigned long long* myf = fl + (tid < nfl ? tid : 0);
    bool ok = (unsigned)(__hip_atomic_load(myf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == G.tag;
    for (int spin = 0; spin < G.spin && !ok; ++spin) {
      __builtin_amdgcn_s_sleep(2);
      ok = (unsigned)(__hip_atomic_load(myf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == G.tag;
    }
    if (!ok) report_expired(G.err, G.err_code);
  }
  float umax = 0.f;
  if constexpr (GATE == 1) {
    __syncthreads();     // role B's members have flagged their rows (written through): X and the unit-seed dY
    // The write-through rows are read with inline-asm sc1 loads, which hipcc neither counts nor orders: they are
    // issued UNCONDITIONALLY, from clamped addresses, in one straight line up to the explicit wait — a load under a
    // branch gets its result register copied (and then reused) at the branch's end, before the data has arrived.
    const bool late = I.dY == G.late_dY;          // the output layer: dY IS the seed (one column): U = e_0
    const int npart = I.dY_part_stride > 0 ? h_n_part : 1;
    const bool tiled = I.dY_part_stride > 0 && h_tiled != 0;
    const int xr0 = xb0 < hB ? xb0 : hB - 1, xr1 = xb0 + 1 < hB ? xb0 + 1 : hB - 1;
    const int xc = xk_ok ? k_base + xq : 0;
    const int ub = bb < hB ? bb : hB - 1, uc = (an_ok && !late) ? ncol : 0;
    const float* usrc = tiled ? I.dY + ((size_t)(uc >> 4) * hB + ub) * 16 + (uc & 15) : I.dY + (size_t)ub * I.ldy + uc;
    const size_t ps = (size_t)I.dY_part_stride;
    const f32x4 rx0 = ld4_sc1(I.X + (size_t)xr0 * I.ldx + xc), rx1 = ld4_sc1(I.X + (size_t)xr1 * I.ldx + xc);
    const f32x4 pa0 = ld4_sc1(usrc), pa1 = ld4_sc1(usrc + (npart > 1 ? ps : 0)), pa2 = ld4_sc1(usrc + (npart > 2 ? 2 * ps : 0)),
                pa3 = ld4_sc1(usrc + (npart > 3 ? 3 * ps : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    vx0 = (xk_ok && xb0 < hB) ? rx0 : z4;
    vx1 = (xk_ok && xb0 + 1 < hB) ? rx1 : z4;
    f32x4 u = ((pa0 + (npart > 1 ? pa1 : z4)) + (npart > 2 ? pa2 : z4)) + (npart > 3 ? pa3 : z4);   // member order, as k_dw_adam sums them
    if (late) u = f32x4{ncol == 0 ? 1.f : 0.f, 0.f, 0.f, 0.f};
    if (!(bb < hB && (an_ok || late))) u = z4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u[t] = (ncol + t < I.N) ? u[t] : 0.f;
      dyt[(an + t) * LDF + bb] = u[t];
      umax = fmaxf(umax, fabsf(u[t]));
    }
  } else if (kind == 2) {
    __syncthreads();     // every member of role U has flagged its rows
    // (unconditional sc1 loads from clamped addresses, then the wait: see GATE 1)
    const bool ok_u = an_ok && bb < hB;
    const int ub = bb < hB ? bb : hB - 1, uc = an_ok ? ncol : 0;
    const float* src = G.U + (((size_t)(uc >> 4) * G.n_act) * hB + ub) * 16 + (uc & 15);
    f32x4 ru[kDuLd];
#pragma unroll
    for (int j = 0; j < kDuLd; ++j) ru[j] = ld4_sc1(src + (size_t)(j < G.n_act ? j : 0) * hB * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < kDuLd; ++j) va[j] = (ok_u && j < G.n_act) ? ru[j] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // X -> 2^4 X -> two fp16 planes, transposed: plane[k][b], the two rows of this lane side by side
  {
    if constexpr (GATE == 2) { /* plain loads: hipcc counts them */ }
    f32x4 a = vx0 * PrecX2::kFwdA, b = vx1 * PrecX2::kFwdA;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool k_in = k_base + xq + t < I.K;
      a[t] = k_in ? a[t] : 0.f;
      b[t] = k_in ? b[t] : 0.f;
    }
    f16x8 hi, lo;
    x2_split8(a, b, hi, lo);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f16x2*>(xh + (size_t)(xq + t) * LDH + xb0) = f16x2{hi[t], hi[4 + t]};
      *reinterpret_cast<f16x2*>(xl + (size_t)(xq + t) * LDH + xb0) = f16x2{lo[t], lo[4 + t]};
    }
  }
  request_state();
  if constexpr (GATE == 2) {
    // ---- du (granules), the combination, the transposed dY tile
    float du[kDuLd];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kDuLd; ++j) ok = ok && (unsigned)(g[j] >> 32) == G.tag;
    for (int spin = 0; spin < G.spin && !ok; ++spin) {
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < kDuLd; ++j)
        if (j < G.n_act) g[j] = __hip_atomic_load(G.seed + (size_t)bb * kDuLd + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int j = 0; j < kDuLd; ++j) ok = ok && (unsigned)(g[j] >> 32) == G.tag;
    }
    if (!ok) report_expired(G.err, G.err_code);
#pragma unroll
    for (int j = 0; j < kDuLd; ++j)
      du[j] = (bb < hB && j < G.n_act) ? (ok ? __uint_as_float((unsigned)g[j]) : __builtin_nanf("")) : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (role U's rows: inline-asm loads)
    stamp();   // rows and seeds in
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kind != 0) {
#pragma unroll
      for (int j = 0; j < kDuLd; ++j) v += va[j] * du[j];
      if (kind == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = hmask[t] > 0.f ? v[t] : 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float x = 0.f;
#pragma unroll
        for (int j = 0; j < kDuLd; ++j) x = (ncol + t == j) ? du[j] : x;
        v[t] = x;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      v[t] = (ncol + t < I.N) ? v[t] : 0.f;
      dyt[(an + t) * LDF + bb] = v[t];
      umax = fmaxf(umax, fabsf(v[t]));
    }
  }
  // the tile's largest |dY| (|U| for GATE 1): every loader wave leaves its own
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) umax = fmaxf(umax, __shfl_xor(umax, o));
  if (lane == 0) amaxw[wave] = umax;
  __syncthreads();       // the operands are in LDS
  if constexpr (GATE == 1) stamp();   // operands staged (the seeds are still out)
#if defined(DW_X2_DEBUG) && DW_X2_DEBUG == 2
  if (GATE == 1) return;
#endif
  if (!cw) {
    __syncthreads();     // (the pack staging's barrier below)
    dw_write_packs(I, tileW, tileT, tid, tk, n_base, 0, ptile, kDwTileN, NSk, NSn, polyak);
    return;
  }

  // ---- the two compute waves
  float dscale = 1.f;      // seeds' share of the A operand's magnitude (GATE 1)
  float* sd = lds + DwX2Lds::seed + 256 * wave;
  if constexpr (GATE == 1) {
    // the per-row seeds, {tag, value} granules straight from role A: every compute wave polls all 256 (4 per lane)
    float smax = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = 64 * q + lane;
      float sv = 0.f;
      if (b < hB && G.n_seed > 0) {
        unsigned long long x = 0;
        bool ok = false;
        for (int spin = 0; spin < G.spin && !ok; ++spin) {
          x = __hip_atomic_load(G.seed + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (unsigned)(x >> 32) == G.tag;
          if (!ok) __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) report_expired(G.err, G.err_code);
        sv = ok ? __uint_as_float((unsigned)x) : __builtin_nanf("");
      }
      sd[b] = sv;
      smax = fmaxf(smax, fabsf(sv));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) smax = fmaxf(smax, __shfl_xor(smax, o));
    dscale = smax;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    stamp();   // seeds in
  }
  float tmax = amaxw[lane & 15];
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  tmax *= (GATE == 1 ? dscale : 1.f);
  const float sa = PrecX2::a_scale(tmax);
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float bsum = 0.f;
  {
    const float* arow = dyt + (size_t)i * LDF + 8 * kk;
    const _Float16* bh = xh + (size_t)(16 * wave + i) * LDH + 8 * kk;
    const _Float16* bl = xl + (size_t)(16 * wave + i) * LDH + 8 * kk;
    auto step = [&](int s, f32x4& acc) {
      f32x4 a0 = ld4(arow + 32 * s), a1 = ld4(arow + 32 * s + 4);
      if constexpr (GATE == 1) {
        a0 *= ld4(sd + 32 * s + 8 * kk);
        a1 *= ld4(sd + 32 * s + 8 * kk + 4);
      }
      bsum += ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
      f16x8 ah, al;
      x2_split8(a0 * sa, a1 * sa, ah, al);
      const f16x8 xhi = *reinterpret_cast<const f16x8*>(bh + 32 * s), xlo = *reinterpret_cast<const f16x8*>(bl + 32 * s);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xlo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xhi, acc, 0, 0, 0);
    };
    // (two steps per trip, not eight unrolled: hipcc hoists every LDS read of an unrolled loop to its top and the
    // kernel — 128 VGPRs per lane at 1024 threads — spills)
#pragma unroll 1
    for (int s = 0; s < 8; s += 2) {
      step(s, acc0);
      step(s + 1, acc1);
    }
  }
  stamp();   // MFMAs done
#if defined(DW_X2_DEBUG) && DW_X2_DEBUG == 3
  if (GATE == 1) return;
#endif
  const float unscale = ad.grad_scale / (sa * PrecX2::kFwdA);
  // ---- Adam (torch.optim.Adam single-tensor semantics) and Polyak on the accumulators, as dw_adam_body's epilogue
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float gr = (acc0[r] + acc1[r]) * unscale;
    float th_new = 0.f, tt_new = 0.f;
    if (e_ok[r]) {
      const size_t eo = (size_t)(n_base + 4 * kk + r) * I.K + ek;
      float mm = p_m[r], vv = p_v[r], th = p_th[r];
      mm = mm + (gr - mm) * ad.omb1;
      vv = vv * ad.beta2 + ad.omb2 * gr * gr;
      th = th - step_size * (mm / (sqrtf(vv) / bc2_sqrt + ad.eps));
      I.w_m[eo] = mm;
      I.w_v[eo] = vv;
      I.w[eo] = th;
      th_new = th;
      if (polyak) {
        tt_new = p_tt[r] * ad.omtau + ad.tau * th;
        I.w_t[eo] = tt_new;
      }
    }
    tileW[4 * kk + r][16 * wave + i] = th_new;
    tileT[4 * kk + r][16 * wave + i] = tt_new;
  }
  // the bias gradient: column sums of dY — this lane's 64 rows, then over the four row groups kk
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  if (b_own) {
    const int n = n_base + i;
    const float gb = bsum * ad.grad_scale;
    float mm = q_m, vv = q_v, th = q_th;
    mm = mm + (gb - mm) * ad.omb1;
    vv = vv * ad.beta2 + ad.omb2 * gb * gb;
    th = th - step_size * (mm / (sqrtf(vv) / bc2_sqrt + ad.eps));
    I.b_m[n] = mm;
    I.b_v[n] = vv;
    I.b[n] = th;
    if (b_pol) I.b_t[n] = q_tt * ad.omtau + ad.tau * th;
  }
  __syncthreads();       // the updated tile is staged
  dw_write_packs(I, tileW, tileT, tid, tk, n_base, 0, ptile, kDwTileN, NSk, NSn, polyak);
  stamp();   // stores issued
}

}  // namespace oprl
