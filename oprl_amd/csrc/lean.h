// lean.h — WAVE-SPECIALISED slice passes (round 5): the critical passes of k_ddpg_chain with one job per wave.
//
// Why.  tp4.h's passes run every stage on all 16 waves of the workgroup: a forward pass is ~1300 static
// instructions PER WAVE around 15 MFMAs (address arithmetic, pointer selects, redundant operand splits, scratch
// reduces), four waves share a SIMD's issue port, and a pass takes ~4-5 us whatever the precision (r04-6: "not a
// memory stage: ~200 instructions per wave x 4 waves per SIMD").  The arithmetic of a pass is ~700 instructions for
// the WHOLE workgroup.  Here
//   * a stage runs on ONE QUARTET of waves (one wave per SIMD) — or on one wave — and a wave has ONE job per pass:
//     its weight fragments and biases are requested straight into ITS registers as early as the data exists
//     (the second pass's while the first one runs), all other waves sit in s_barrier;
//   * the products are formed TRANSPOSED, D^T = W-tile · X^T: the weight fragment is the MFMA's A operand (the packs'
//     lane order serves either side), the activations the B operand, and a lane ends up with FOUR CONSECUTIVE
//     FEATURES of one minibatch row — bias is one 16-byte load, the epilogue one LDS store;
//   * PrecX2: activations live in LDS as their two fp16 planes, split ONCE by the lane that produces them (8 vector
//     instructions per 16 x 16 tile) in the k-order of the packs inside every 32-column block — a consumer's operand
//     is two ds_read_b128, no conversion (tp4.h splits per consumer wave and macro step: 16 instructions each);
//   * exact fp32: row-major fp32 tiles as before (the operand is one ds_read_b128 per macro step).
// Decomposition, summation order of the exchanges (member order) and the exchange protocol (tp3.h granules) are those
// of tp4.h's clusters of eight; inside a member the 256-deep contraction of the hidden layer is summed as two halves.
#pragma once
#include "tp4.h"

namespace oprl {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kLdH = 272;                    // halfs between two rows of a 256-column fp16 plane (136 dwords = 8 mod 64)
constexpr int kLd0H = 112;                   // ... of a layer-0 input plane (<= 96 columns; 56 dwords = -8 mod 64)
constexpr int kLdLH = 48;                    // ... of a member-local plane (32 columns; 24 dwords)
constexpr int kLdLF = 40;                    // floats between two rows of a member-local fp32 tile (32 columns)
constexpr int kLeanX0Floats = 16 * kLd0H;    // floats of the layer-0 input planes (hi | lo)

// position (halfs) of column c inside its plane row: 32-column blocks in the k-order of the fp16 packs (pack16_index)
__host__ __device__ constexpr int lean_pos(int c) { return (c & ~31) | (((c & 15) >> 2) << 3) | (((c >> 4) & 1) << 2) | (c & 3); }

template <class T> struct LeanIsCoh { static constexpr bool v = false; };
template <class P> struct LeanIsCoh<Coh<P>> { static constexpr bool v = true; };
// four consecutive bias elements (16-byte aligned)
template <class PL> __device__ __forceinline__ f32x4 lean_ldb4(const float* p) {
  if constexpr (LeanIsCoh<PL>::v) return ld4c(p);
  else return ld4(p);
}

// ---------------------------------------------------------------------------------------------------------------
// Operand handling per arithmetic.  LDH / LDF: the row stride of the tile the call addresses.
// ---------------------------------------------------------------------------------------------------------------
template <bool X2> struct LeanAct;

template <> struct LeanAct<true> {
  struct Bop { f16x8 h, l; };
  // planes: hi at `buf`, lo `rows * LDH` halfs behind it (rows = 16)
  template <int LDH> __device__ static __forceinline__ Bop ldB(const float* buf, int i, int kk, int s) {
    const _Float16* hp = reinterpret_cast<const _Float16*>(buf) + i * LDH + 32 * s + 8 * kk;
    return Bop{*reinterpret_cast<const f16x8*>(hp), *reinterpret_cast<const f16x8*>(hp + 16 * LDH)};
  }
  __device__ static __forceinline__ void mma(f32x4& acc, const FragX2& w, const Bop& b) {
    const f16x8 wh = __builtin_bit_cast(f16x8, w.hi), wl = __builtin_bit_cast(f16x8, w.lo);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, b.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, b.h, acc, 0, 0, 0);
  }
  // features f0 .. f0 + 3 (f0 a multiple of 4) of row i <- v * sc, split
  template <int LDH> __device__ static __forceinline__ void st4(float* buf, int i, int f0, const f32x4 v, float sc) {
    const f32x4 x = v * sc;
    const f16x4 hi = __builtin_convertvector(x, f16x4);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 hp = __builtin_bit_cast(u32x2, hi);
    const f32x4 r = f32x4{x2_res_lo(x[0], hp[0]), x2_res_hi(x[1], hp[0]), x2_res_lo(x[2], hp[1]), x2_res_hi(x[3], hp[1])};
    const f16x4 lo = __builtin_convertvector(r, f16x4);
    _Float16* hpp = reinterpret_cast<_Float16*>(buf) + i * LDH + lean_pos(f0);
    *reinterpret_cast<f16x4*>(hpp) = hi;
    *reinterpret_cast<f16x4*>(hpp + 16 * LDH) = lo;
  }
  // the signs of features f0 .. f0 + 3 of row i as written by st4 (ReLU masks: v > 0)
  template <int LDH> __device__ static __forceinline__ f32x4 ld4pos(const float* buf, int i, int f0) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(buf) + i * LDH + lean_pos(f0));
    return __builtin_convertvector(h, f32x4);
  }
};

template <> struct LeanAct<false> {
  struct Bop { f32x4 v; };
  template <int LDF> __device__ static __forceinline__ Bop ldB(const float* buf, int i, int kk, int s) {
    return Bop{ld4(buf + i * LDF + 16 * s + 4 * kk)};
  }
  __device__ static __forceinline__ void mma(f32x4& acc, const f32x4 w, const Bop& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = mfma4(w[t], b.v[t], acc);
  }
  template <int LDF> __device__ static __forceinline__ void st4(float* buf, int i, int f0, const f32x4 v, float) {
    *reinterpret_cast<f32x4*>(buf + i * LDF + f0) = v;
  }
  template <int LDF> __device__ static __forceinline__ f32x4 ld4pos(const float* buf, int i, int f0) { return ld4(buf + i * LDF + f0); }
};

// strides of the three kinds of tiles by arithmetic
template <bool X2> struct LeanLd;
template <> struct LeanLd<true> { static constexpr int X0 = kLd0H, H = kLdH, L = kLdLH; };
template <> struct LeanLd<false> { static constexpr int X0 = kX0Ld, H = kWL4, L = kLdLF; };

// fp32 rows [16][kX0Ld] -> the layer-0 input planes (PrecX2), columns [c0, c0 + n) (n <= 96), by `nthr` threads from `t0`
__device__ __forceinline__ void lean_x0_planes(const float* x, float* planes, int c0, int n, int t, int nthr) {
  _Float16* hp = reinterpret_cast<_Float16*>(planes);
  for (int e = t; e < 16 * n; e += nthr) {
    const int row = e / n, col = c0 + (e - row * n);
    const float v = x[row * kX0Ld + col] * PrecX2::kFwdA;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    hp[row * kLd0H + lean_pos(col)] = h;
    hp[16 * kLd0H + row * kLd0H + lean_pos(col)] = l;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The cluster all-reduce of a NARROW block (<= 8 columns) held TRANSPOSED in accumulator layout: lane (kk, i) has
// features 4 kk + r of minibatch row i.  Valid features [f_first, f_first + ncols) travel as columns cc_first .. of the
// exchange block and land in out[row * kOutLd + cc_first + k] (+ b0 / b1: the bias of this lane's element slot 0 / 1,
// lean_elem_col).  One wave; wscr = 256 floats of LDS of its own.  Slots are feature-major (16 rows of a column
// contiguous).  poll = false: publish only (a member that does not need the sum).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lean_elem_col(int j, int ncols) {     // column (0 .. ncols) of this lane's element slot j, -1: none
  const int e = (int)(threadIdx.x & 63) + 64 * j;
  return e < 16 * ncols ? e >> 4 : -1;
}
template <int NM, class ST = NoStamp>
__device__ __forceinline__ void lean_allreduce(const f32x4 mine, int f_first, int ncols, int cc_first, float* wscr,
                                               float b0, float b1, float* out, const Tp& tp, bool poll = true, ST sf = ST()) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  const unsigned tag = (tp.tag << 6) | (unsigned)(tp.stage & 63);
#pragma unroll
  for (int r = 0; r < 4; ++r) wscr[(4 * kk + r) * 16 + i] = mine[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int n = 16 * ncols;                     // <= 128
  unsigned long long* base = tp.xbuf + (size_t)tp.stage * NM * kTpBlk;
  float val[2];
  const float bv[2] = {b0, b1};
  int off[2], oidx[2];
  bool have[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = lane + 64 * j;
    have[j] = e < n;
    const int fl = have[j] ? e >> 4 : 0, row = e & 15;
    val[j] = wscr[(f_first + fl) * 16 + row];
    off[j] = (cc_first + fl) * 16 + row;
    oidx[j] = row * kOutLd + cc_first + fl;
    if (have[j]) {
      const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(val[j]);
      if (tp.local) __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off[j], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off[j], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  sf();     // published
  if (!poll) return;
  if (have[0]) {
    auto issue = [&](unsigned long long (&x)[NM][2]) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          x[m][j] = (m == tp.c || !have[j]) ? ((unsigned long long)tag << 32)
                                            : __hip_atomic_load(base + (size_t)m * kTpBlk + off[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto good = [&](const unsigned long long (&x)[NM][2]) {
      bool ok = true;
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) ok = ok && (unsigned)(x[m][j] >> 32) == tag;
      return ok;
    };
    auto finish = [&](const unsigned long long (&x)[NM][2], bool ok) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < NM; ++m) sum += (m == tp.c) ? val[j] : __uint_as_float((unsigned)x[m][j]);
        if (have[j]) out[oidx[j]] = ok ? sum + bv[j] : __builtin_nanf("");
      }
    };
    unsigned long long xa[NM][2];
    bool ok = false;
    // (one poll set at a time, a sleep between two: polling harder — two sets in flight, no sleep — made every exchange
    // and every stage around it SLOWER, 26 -> 32 us per update: the polls queue in front of the very stores they wait for)
    for (int spin = 0; spin < tp.spin && !ok; ++spin) {
      issue(xa);
      ok = good(xa);
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    sf();   // all peers seen
    if (ok) finish(xa, true);
    if (!ok) {
      report_expired(tp.err, tp.err_code | SITE_CLUSTER);
      finish(xa, false);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Jobs.  `PL` = the precision policy, possibly Coh<> (engine.h); q = the wave's index inside its quartet.
// Every job is load() — requests only, as early as the weights exist — and run().
// ---------------------------------------------------------------------------------------------------------------
// Layer 0 (replicated): quartet wave q computes hidden tiles 4 q .. 4 q + 3 over the whole (short) contraction.
// S0P macro steps come from registers; inputs wider than that (humanoid) fetch the rest inside run().
template <class PL>
struct LeanL0 {
  static constexpr bool X2 = PL::kX2;
  static constexpr int S0P = X2 ? 1 : 2;             // 32 input columns
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  typename PL::Frag w[4][S0P];
  f32x4 b[4];
  const float* pf;
  int ns0;
  __device__ __forceinline__ void load(const float* pf0, const float* b0, int k0, int q) {
    const int lane = threadIdx.x & 63, kk = lane >> 4;
    ns0 = (k0 + PL::KS - 1) / PL::KS;
    pf = pf0 + (size_t)(4 * q) * ns0 * PL::kBlk + lane * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) PL::template ldfn<S0P>(w[j], pf + (size_t)j * ns0 * PL::kBlk, ns0);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = lean_ldb4<PL>(b0 + 16 * (4 * q + j) + 4 * kk);
  }
  // x0: the input tile (planes / fp32 rows); h1: the hidden tile it writes.  Returns false if a value left PrecX2's range.
  __device__ __forceinline__ bool run(const float* x0, float* h1, int q) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    constexpr float kO = PL::kOut / PL::kFwdA;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S0P; ++s) {
      if (s < ns0) {
        const typename ACT::Bop bx = ACT::template ldB<LD::X0>(x0, i, kk, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) ACT::mma(acc[j], w[j][s], bx);
      }
    }
    for (int s = S0P; s < ns0; ++s) {                 // (wide inputs: fragments fetched here)
      const typename ACT::Bop bx = ACT::template ldB<LD::X0>(x0, i, kk, s);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const typename PL::Frag f = PL::ldf(pf + ((size_t)j * ns0 + s) * PL::kBlk);
        ACT::mma(acc[j], f, bx);
      }
    }
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 pre = acc[j] * kO + b[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
      const f32x4 v = f32x4{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), fmaxf(pre[2], 0.f), fmaxf(pre[3], 0.f)};
      ACT::template st4<LD::H>(h1, i, 16 * (4 * q + j) + 4 * kk, v, PL::kFwdA);
    }
    return ok;
  }
};

// Layer 1, member c of EIGHT: the member's two hidden tiles (columns 32 c ..), quartet wave q = (tile q & 1, half q >> 1
// of the 256-deep contraction).  run_partial(): the wave's partial tile; the upper halves go through `scr` (2 x 256
// floats), [barrier], finish(): waves q < 2 add them, bias + ReLU, and write the member-LOCAL tile h2 (32 columns).
template <class PL>
struct LeanL1 {
  static constexpr bool X2 = PL::kX2;
  static constexpr int W = kW4 / PL::KS, NQ = W / 2;       // 8 / 16 macro steps; 4 / 8 per wave
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  typename PL::Frag w[NQ];
  f32x4 b;
  f32x4 acc;
  __device__ __forceinline__ void load(const float* pf1, const float* b1, int c, int q) {
    const int lane = threadIdx.x & 63, kk = lane >> 4;
    const int t = q & 1, kh = q >> 1;
    PL::template ldfn<NQ>(w, pf1 + ((size_t)(2 * c + t) * W + kh * NQ) * PL::kBlk + lane * 4);
    b = lean_ldb4<PL>(b1 + 32 * c + 16 * t + 4 * kk);
  }
  __device__ __forceinline__ void run_partial(const float* h1, float* scr, int q) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const int t = q & 1, kh = q >> 1;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      const typename ACT::Bop bx = ACT::template ldB<LD::H>(h1, i, kk, kh * NQ + s);
      if (s & 1) ACT::mma(a1, w[s], bx);
      else ACT::mma(a0, w[s], bx);
    }
    acc = a0 + a1;
    if (kh == 1) *reinterpret_cast<f32x4*>(scr + t * 256 + lane * 4) = acc;
  }
  // (waves q < 2 only) -> pre-activation values of features 16 t + 4 kk + r (member-local) of row i
  __device__ __forceinline__ f32x4 finish_pre(const float* scr, int q) {
    const int lane = threadIdx.x & 63;
    constexpr float kO = PL::kOut / PL::kFwdA;
    return (acc + ld4(scr + (q & 1) * 256 + lane * 4)) * kO + b;
  }
};

// Output layer (<= 8 outputs: one tile) over the member's 32 columns + the cluster all-reduce: ONE wave.
template <class PL, int NM>
struct LeanL2 {
  static constexpr bool X2 = PL::kX2;
  static constexpr int W = kW4 / PL::KS, M = 32 / PL::KS;     // 1 / 2 macro steps
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  typename PL::Frag w[M];
  float bv[2];
  __device__ __forceinline__ void load(const float* pf2, const float* b2, int n_out, int c) {
    const int lane = threadIdx.x & 63;
    PL::template ldfn<M>(w, pf2 + ((size_t)c * M) * PL::kBlk + lane * 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = lean_elem_col(j, n_out);
      bv[j] = k >= 0 ? PL::ldb(b2 + k) : 0.f;
    }
  }
  template <class ST = NoStamp>
  __device__ __forceinline__ void run(const float* h2, float* wscr, float* outS, int n_out, const Tp& tp, bool poll = true, ST sf = ST()) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    constexpr float kO = PL::kOut / PL::kFwdA;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < M; ++s) ACT::mma(acc, w[s], ACT::template ldB<LD::L>(h2, i, kk, s));
    lean_allreduce<NM>(acc * kO, 0, n_out, 0, wscr, bv[0], bv[1], outS, tp, poll, sf);
  }
};

// a pass's barrier (workgroup-wide; LDS traffic only — requests to memory stay in flight across it)
__device__ __forceinline__ void lean_bar() { __syncthreads(); }

// ---------------------------------------------------------------------------------------------------------------
// Scalar-output net (a critic), forward + constant-seed backward to the input columns [dcol0, dcol0 + dcols) (dcols <= 8,
// inside at most two 16-column tiles), on a cluster of EIGHT — tp4_scalar_fb<P, 8>'s arithmetic, one job per wave:
//   quartet 0: layer 0 -> h1                         [bar 1]
//   quartet 1: layer 1 partial                       [bar 2]   waves 4, 5: h2 and the unit-seed g2 (member-local)   [bar 3]
//   quartet 2: dz1 partial (4 tiles each) over the member's columns, masked in place over h1     [bar 4]
//              wave 0 (idle since layer 0): q = output layer + all-reduce (the lead member's q_sum_out; the others publish only)
//   quartet 3: input-column gradient, a quarter of the 256-deep contraction each -> scr          [bar 5]
//              waves 12, 13: tile sums + all-reduce -> dactS                                       [bar 6]
// The caller has x0 ready (visible after a barrier of its own or bar 0 below) and every wave calls this once.
// tp.stage advances by 2 (q: stage, gradient: stage + 1).
// ---------------------------------------------------------------------------------------------------------------
struct LeanBufs {
  float* x0;        // layer-0 input: PrecX2 planes [2][16][kLd0H] halfs / fp32 rows [16][kX0Ld]
  float* h1;        // [2][16][kLdH] halfs / [16][kWL4] floats
  float* h2;        // member-local: [2][16][kLdLH] halfs / [16][kLdLF] floats
  float* g2;        // the same (scalar_fb)
  float* scr;       // >= 4096 floats
};

template <class PL, class ST = NoStamp>
__device__ __forceinline__ void lean_scalar_fb(const Net& net, const LeanBufs& L, float* outS, Tp& tp, int row0, int B, float seed,
                                               int dcol0, int dcols, float* dactS, ST sf, float* q_sum_out, const BiasOv bo) {
  constexpr bool X2 = PL::kX2;
  constexpr int NM = 8;
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  constexpr int W = kW4 / PL::KS, M = 32 / PL::KS, Q4 = W / 4;
  constexpr int BK = PL::kBlk;
  const float* const nb0 = bo.b0 != nullptr ? bo.b0 : net.b[0];
  const float* const nb1 = bo.b1 != nullptr ? bo.b1 : net.b[1];
  const float* const nb2 = bo.b2 != nullptr ? bo.b2 : net.b[2];
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int quart = wave >> 2, q = wave & 3;
  const int c = tp.c;
  // PrecX2: the unit-seed gradient tiles go in scaled by 2^12 / |seed| (tp4_scalar_fb)
  float sb = 1.f;
  if constexpr (X2) sb = 4.f * PL::a_scale(fabsf(seed));
  const float ob = PL::kOut / sb;
  const int dt0 = dcol0 >> 4, dnt = ((dcol0 + dcols - 1) >> 4) - dt0 + 1;      // 1 or 2 tiles
  if (quart == 0) {
    LeanL0<PL> j0;
    j0.load(net.pf[0], nb0, net.dims[0], q);
    LeanL2<PL, NM> j2;
    if (q == 0) j2.load(net.pf[2], nb2, 1, c);
    lean_bar();                                   // bar 0: x0 visible
    const bool ok = j0.run(L.x0, L.h1, q);
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    sf();
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3: h2, g2 visible
    if (q == 0) {
      const bool lead = c == 0;
      j2.run(L.h2, L.scr + 2048, outS, 1, tp, lead && q_sum_out != nullptr);
      if (lead && q_sum_out != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float qv = (lane < kR && row0 + lane < B) ? outS[lane * kOutLd] : 0.f;
        const float qs = row16_sum(qv);
        if (lane == 0) *q_sum_out = qs;
      }
    }
    lean_bar();                                   // 4
    lean_bar();                                   // 5
  } else if (quart == 1) {
    LeanL1<PL> j1;
    j1.load(net.pf[1], nb1, c, q);
    // W3[32 c + 16 t + 4 kk + r], r < 4 (the critic's output layer through its W^T pack: one macro step, element 0 of lane
    // (kk' = 0, i' = 4 kk + r))
    f32x4 w3 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < 2) {
      const int NSo = (net.dims[3] + PL::KS - 1) / PL::KS;
      const float* p = net.pb[2] + (size_t)(2 * c + q) * NSo * BK + (4 * kk) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) w3[r] = PL::first(p + 4 * r);
    }
    lean_bar();                                   // 0
    lean_bar();                                   // 1: h1 visible
    j1.run_partial(L.h1, L.scr, q);
    sf();
    lean_bar();                                   // 2: upper halves visible
    if (q < 2) {
      const f32x4 pre = j1.finish_pre(L.scr, q);
      bool ok = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
      if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
      f32x4 v, g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaxf(pre[r], 0.f);
        g[r] = v[r] > 0.f ? seed * w3[r] : 0.f;
      }
      ACT::template st4<LD::L>(L.h2, i, 16 * q + 4 * kk, v, PL::kFwdA);
      ACT::template st4<LD::L>(L.g2, i, 16 * q + 4 * kk, g, sb);
    }
    sf();
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    lean_bar();                                   // 5
  } else if (quart == 2) {
    // dz1 partial: k tiles 4 q .. 4 q + 3 of W2^T, contraction over the member's 32 columns
    typename PL::Frag wz[4][M];
    {
      const float* p = net.pb[1] + ((size_t)(4 * q) * W + c * M) * BK + lane * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) PL::template ldfn<M>(wz[j], p + (size_t)j * W * BK);
    }
    lean_bar();                                   // 0
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    // (the masks of this lane's outputs: h1 is complete since bar 1)
    f32x4 mk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = ACT::template ld4pos<LD::H>(L.h1, i, 16 * (4 * q + j) + 4 * kk);
    lean_bar();                                   // 3: g2 visible
    {
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < M; ++s) {
        const typename ACT::Bop bx = ACT::template ldB<LD::L>(L.g2, i, kk, s);
#pragma unroll
        for (int j = 0; j < 4; ++j) ACT::mma(acc[j], wz[j][s], bx);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 d;
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = mk[j][r] > 0.f ? acc[j][r] * ob : 0.f;
        ACT::template st4<LD::H>(L.h1, i, 16 * (4 * q + j) + 4 * kk, d, sb);
      }
    }
    sf();
    lean_bar();                                   // 4: dz1 partial visible (over h1)
    lean_bar();                                   // 5
  } else {
    // input-column gradient: tiles dt0 (, dt0 + 1) of W0^T, quarter q of the 256-deep contraction
    typename PL::Frag wd[2][Q4];
    {
      const float* p = net.pb[0] + ((size_t)dt0 * W + q * Q4) * BK + lane * 4;
      PL::template ldfn<Q4>(wd[0], p);
      PL::template ldfn<Q4>(wd[1], p + (size_t)W * BK, dnt > 1 ? Q4 : 0);
    }
    lean_bar();                                   // 0
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    {
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int s = 0; s < Q4; ++s) {
        const typename ACT::Bop bx = ACT::template ldB<LD::H>(L.h1, i, kk, q * Q4 + s);
        ACT::mma(a0, wd[0][s], bx);
        if (dnt > 1) ACT::mma(a1, wd[1][s], bx);
      }
      *reinterpret_cast<f32x4*>(L.scr + (q * 2 + 0) * 256 + lane * 4) = a0;
      *reinterpret_cast<f32x4*>(L.scr + (q * 2 + 1) * 256 + lane * 4) = a1;
    }
    sf();
    lean_bar();                                   // 5: quarters visible
    if (q < dnt) {
      Tp tp2 = tp;
      tp2.stage = tp.stage + 1;
      f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) part += ld4(L.scr + (k * 2 + q) * 256 + lane * 4);
      const int t0 = 16 * (dt0 + q);
      const int f_first = dcol0 > t0 ? dcol0 - t0 : 0;
      const int cc_first = t0 + f_first - dcol0;
      const int ncols = min(16 - f_first, dcols - cc_first);
      lean_allreduce<NM>(part * ob, f_first, ncols, cc_first, L.scr + 2048 + 256 * (1 + q), 0.f, 0.f, dactS, tp2);
    }
  }
  tp.stage += 2;
  sf();
  lean_bar();                                     // 6: dactS (and outS) visible
}


// one WAVE waits for n (<= 192) flag granules {tag, *}; bounded; `wait` false: nothing to wait for
__device__ __forceinline__ void lean_wave_wait(const unsigned long long* flags, int n, unsigned tag, bool wait, unsigned* err, unsigned code) {
  if (!wait) return;
  const int lane = threadIdx.x & 63;
  bool ok = false;
  auto sample = [&](unsigned (&t)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = lane + 64 * k;
      t[k] = idx < n ? (unsigned)(__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) : tag;
    }
  };
  auto all_set = [&](const unsigned (&t)[3]) {
    const bool mine = t[0] == tag && t[1] == tag && t[2] == tag;
    return __builtin_amdgcn_ballot_w64(!mine) == 0ull;
  };
  unsigned ta[3];
  for (int spin = 0; spin < kTpSpin && !ok; ++spin) {
    sample(ta);
    ok = all_set(ta);
    if (!ok) __builtin_amdgcn_s_sleep(2);
  }
  if (!ok && lane == 0) report_expired(err, code);
}

// ---------------------------------------------------------------------------------------------------------------
// Role A of k_ddpg_chain — a' = tanh(actor_target(s')), q' = critic_target(s', a') — on a cluster of EIGHT, one job
// per wave.  The target critic's weights are final since the critic's tiles of the update before (flags ct): their
// requests go out BEFORE the actor's tiles (flags at) are waited for, into the registers of the waves that use them.
//   waves 0-3   actor_t layer 0            | wave 0: critic_t output layer + all-reduce
//   waves 4-7   actor_t layer 1
//   waves 8-11  critic_t layer 0           | wave 8: actor_t output layer + all-reduce, a' = tanh(.) -> the input tile
//   waves 12-15 critic_t layer 1           | wave 15: the flag waits
// xb: the fp32 rows [s' | .] (a' is written there too); PrecX2: L.x0 = the planes of [s' | 0] (a' is added).
// TAIL (wave 0 only): request(ctx) while the wave idles before its last job, finish(ctx, outS) right behind q' (outS[row *
// kOutLd] = q', this wave's own writes) — the TD target and the seeds, no workgroup barrier in front of them.
// tp.stage advances by 2.
// ---------------------------------------------------------------------------------------------------------------
struct LeanFlags {
  const unsigned long long* ct; int n_ct;
  const unsigned long long* at; int n_at;
  unsigned tag; bool wait;
  unsigned* err; unsigned code;
};
template <class PL, class TAIL, class ST, class TCTX>
__device__ __forceinline__ void lean_role_a(const Net& actor_t, const Net& critic_t, const BiasOv ba, const BiasOv bc, const LeanBufs& L,
                                            float* xb, float* outS, Tp& tp, int S, int Ad, int row0, int B, const LeanFlags F,
                                            ST sf, const TCTX& tctx) {
  constexpr bool X2 = PL::kX2;
  constexpr int NM = 8;
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  const float* const ab0 = ba.b0 != nullptr ? ba.b0 : actor_t.b[0];
  const float* const ab1 = ba.b1 != nullptr ? ba.b1 : actor_t.b[1];
  const float* const ab2 = ba.b2 != nullptr ? ba.b2 : actor_t.b[2];
  const float* const cb0 = bc.b0 != nullptr ? bc.b0 : critic_t.b[0];
  const float* const cb1 = bc.b1 != nullptr ? bc.b1 : critic_t.b[1];
  const float* const cb2 = bc.b2 != nullptr ? bc.b2 : critic_t.b[2];
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int quart = wave >> 2, q = wave & 3;
  const int c = tp.c;
  if (quart == 0) {
    LeanL2<PL, NM> j2;
    LeanL0<PL> j0;
    TAIL tail;                                    // (wave 0: what follows q' — its state lives in this wave's registers only)
    lean_bar();                                   // a: the critic's tiles of the update before are done
    if (q == 0) j2.load(critic_t.pf[2], cb2, 1, c);
    lean_bar();                                   // b: the actor's tiles as well; s' in place
    j0.load(actor_t.pf[0], ab0, actor_t.dims[0], q);
    const bool ok = j0.run(L.x0, L.h1, q);
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    sf();
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    lean_bar();                                   // 5
    if (q == 0) tail.request(tctx);
    lean_bar();                                   // 6
    lean_bar();                                   // 7: h2 of the second pass
    if (q == 0) {
      Tp tp2 = tp;
      tp2.stage = tp.stage + 1;
      j2.run(L.h2, L.scr + 2304, outS, 1, tp2);
      sf();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tail.finish(tctx, outS);                    // (q' is this wave's own: no barrier in front of the seeds)
    }
    sf();
  } else if (quart == 1) {
    LeanL1<PL> j1;
    lean_bar();                                   // a
    lean_bar();                                   // b
    j1.load(actor_t.pf[1], ab1, c, q);
    lean_bar();                                   // 1: h1
    j1.run_partial(L.h1, L.scr, q);
    sf();
    lean_bar();                                   // 2
    if (q < 2) {
      const f32x4 pre = j1.finish_pre(L.scr, q);
      bool ok = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
      if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
      ACT::template st4<LD::L>(L.h2, i, 16 * q + 4 * kk, f32x4{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), fmaxf(pre[2], 0.f), fmaxf(pre[3], 0.f)}, PL::kFwdA);
    }
    sf();
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    lean_bar();                                   // 5
    lean_bar();                                   // 6
    lean_bar();                                   // 7
  } else if (quart == 2) {
    LeanL0<PL> j0;
    LeanL2<PL, NM> j2;
    lean_bar();                                   // a
    j0.load(critic_t.pf[0], cb0, critic_t.dims[0], q);
    lean_bar();                                   // b
    if (q == 0) j2.load(actor_t.pf[2], ab2, Ad, c);
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3: h2 of the first pass
    if (q == 0) {
      sf();
      j2.run(L.h2, L.scr + 2048, outS, Ad, tp, true, sf);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // a' = tanh(.) -> the input tile's action columns (rows beyond the batch: zero)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = lane + 64 * j;
        if (e < 16 * Ad) {
          const int col = e >> 4, row = e & 15;
          const float a = row0 + row < B ? tanhf(outS[row * kOutLd + col]) : 0.f;
          xb[row * kX0Ld + S + col] = a;
          if constexpr (X2) {
            const float v = a * PL::kFwdA;
            const _Float16 h = (_Float16)v;
            _Float16* hp = reinterpret_cast<_Float16*>(L.x0) + row * kLd0H + lean_pos(S + col);
            hp[0] = h;
            hp[16 * kLd0H] = (_Float16)(v - (float)h);
          }
        }
      }
    }
    sf();
    lean_bar();                                   // 4: [s' | a'] complete
    const bool ok = j0.run(X2 ? L.x0 : xb, L.h1, q);
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    sf();
    lean_bar();                                   // 5
    lean_bar();                                   // 6
    lean_bar();                                   // 7
  } else {
    LeanL1<PL> j1;
    if (q == 3) lean_wave_wait(F.ct, F.n_ct, F.tag, F.wait, F.err, F.code);
    lean_bar();                                   // a
    j1.load(critic_t.pf[1], cb1, c, q);
    if (q == 3) lean_wave_wait(F.at, F.n_at, F.tag, F.wait, F.err, F.code);
    lean_bar();                                   // b
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    lean_bar();                                   // 5: h1 of the second pass
    j1.run_partial(L.h1, L.scr, q);
    sf();
    lean_bar();                                   // 6
    if (q < 2) {
      const f32x4 pre = j1.finish_pre(L.scr, q);
      bool ok = true;
#pragma unroll
      for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
      if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
      ACT::template st4<LD::L>(L.h2, i, 16 * q + 4 * kk, f32x4{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), fmaxf(pre[2], 0.f), fmaxf(pre[3], 0.f)}, PL::kFwdA);
    }
    sf();
    lean_bar();                                   // 7
  }
  tp.stage += 2;
  lean_bar();                                     // 8: q' in outS
  sf();
}

}  // namespace oprl
