// lean.h — WAVE-SPECIALISED slice passes (round 5): the critical passes of k_ddpg_chain with one job per wave.
//
// Why.  tp4.h's passes run every stage on all 16 waves of the workgroup: a forward pass is ~1300 static
// instructions PER WAVE around 15 MFMAs (address arithmetic, pointer selects, redundant operand splits, scratch
// reduces), four waves share a SIMD's issue port, and a pass takes ~4-5 us whatever the precision (r04-6: "not a
// memory stage: ~200 instructions per wave x 4 waves per SIMD").  The arithmetic of a pass is ~700 instructions for
// the WHOLE workgroup.  Here
//   * a wave has ONE job per stage, a few dozen instructions, and its weight fragments and biases are requested straight
//     into ITS registers as early as the data exists (the second pass's while the first one runs).  One wave issues one
//     vector instruction per ~5 cycles whatever its neighbours do (tools/ubench_icache2.hip: 4.8 cycles with 1, 4 or 16
//     waves per workgroup), so what a stage costs is the instruction count of its LONGEST wave: layer 0 and the dz1
//     stage are one 16 x 16 tile per wave on all sixteen, the member's two layer-1 tiles run on a quartet, the output
//     layer + cluster exchange on one wave; the others sit in s_barrier;
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
// The cluster all-reduce of a NARROW block (<= 8 valid columns) held TRANSPOSED in accumulator layout — lane (kk, i) has
// features 4 kk + r of minibatch row i — STRAIGHT FROM THE REGISTERS: no LDS transpose on either side of the exchange
// (the first form went through LDS twice: 0.56 us from the last MFMA to the publish, profiles/r05_experiments.txt).
// Valid features [f_first, f_first + ncols); the lane groups kk that hold one of them take part (at most three).  A
// member's block: [16 rows][12 slots] granules from `slot0` on; a lane's four granules are 32 contiguous bytes, read
// as two 16-byte loads per peer (each 8-byte half carries its own tag: a torn pair is two valid granules or a retry).
// Returns whether this lane holds sums; sum[r] = the members' sum (member order) of feature 4 kk + r.
// poll = false: publish only.
// ---------------------------------------------------------------------------------------------------------------
template <int NM, class ST = NoStamp>
__device__ __forceinline__ bool lean_ar(const f32x4 mine, int f_first, int ncols, int slot0, const Tp& tp, f32x4& sum,
                                        bool poll = true, ST sf = ST()) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  const unsigned tag = (tp.tag << 6) | (unsigned)(tp.stage & 63);
  const int g = kk - (f_first >> 2), ng = ((f_first + ncols - 1) >> 2) - (f_first >> 2) + 1;
  const bool in = g >= 0 && g < ng;
  // (every group's four slots are written — zeros beyond the valid columns — so that a reader's 16-byte halves always
  // find two tagged granules)
  const int nh = (ncols + (f_first & 3)) > 2 ? 2 : 1;      // 16-byte halves a lane needs (wave-uniform)
  unsigned long long* base = tp.xbuf + (size_t)tp.stage * NM * kTpBlk + slot0;
  const int off = i * 12 + 4 * (in ? g : 0);
  sum = mine;
  if (!in) return false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = 4 * kk + r;
    const float v = (f >= f_first && f < f_first + ncols) ? mine[r] : 0.f;
    const unsigned long long gv = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    if (r < 2 * nh) {
      if (tp.local) __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off + r, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off + r, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  sf();     // published
  if (!poll) return true;
  const float* fb = reinterpret_cast<const float*>(base);
  f32x4 x[NM][2];
  bool ok = false;
  for (int spin = 0; spin < tp.spin && !ok; ++spin) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (m != tp.c && h < nh) x[m][h] = ld4_agent(fb, (unsigned)(((size_t)m * kTpBlk + off + 2 * h) * 2));
    ok = true;
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (m != tp.c && h < nh) ok = ok && __float_as_uint(x[m][h][1]) == tag && __float_as_uint(x[m][h][3]) == tag;
    if (!ok) __builtin_amdgcn_s_sleep(1);
  }
  sf();     // all peers seen
  if (!ok) {
    report_expired(tp.err, tp.err_code | SITE_CLUSTER);
    const float nan = __builtin_nanf("");
    sum = f32x4{nan, nan, nan, nan};
    return true;
  }
  sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    if (m == tp.c) {
      sum += mine;
    } else {
      sum[0] += x[m][0][0];
      sum[1] += x[m][0][2];
      if (nh > 1) { sum[2] += x[m][1][0]; sum[3] += x[m][1][2]; }
    }
  }
  return true;
}

// tanh through the hardware's exp2 and reciprocal, 1 - 2 / (e^2x + 1): absolute error ~1.2e-7 (the library routine is ~40
// instructions with a branch; four of them per lane sat behind the first exchange of role A)
__device__ __forceinline__ float lean_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);      // e^(2x)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// ---------------------------------------------------------------------------------------------------------------
// Jobs.  `PL` = the precision policy, possibly Coh<> (engine.h).  Every job is load() — requests only, as early as
// the weights exist — and run().
// ---------------------------------------------------------------------------------------------------------------
// Layer 0 (replicated over the cluster's members): wave w computes hidden tile w over the whole (short) contraction.
// S0P macro steps come from registers; inputs wider than that (humanoid) fetch the rest inside run().
template <class PL>
struct LeanL0 {
  static constexpr bool X2 = PL::kX2;
  static constexpr int S0P = X2 ? 1 : 2;             // 32 input columns
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  typename PL::Frag w[S0P];
  f32x4 b;
  const float* pf;
  int ns0;
  __device__ __forceinline__ void load(const float* pf0, const float* b0, int k0, int wave) {
    const int lane = threadIdx.x & 63, kk = lane >> 4;
    ns0 = (k0 + PL::KS - 1) / PL::KS;
    pf = pf0 + (size_t)wave * ns0 * PL::kBlk + lane * 4;
    PL::template ldfn<S0P>(w, pf, ns0);
    b = lean_ldb4<PL>(b0 + 16 * wave + 4 * kk);
  }
  // x0: the input tile (planes / fp32 rows); h1: the hidden tile it writes.  Returns false if a value left PrecX2's range.
  __device__ __forceinline__ bool run(const float* x0, float* h1, int wave) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    constexpr float kO = PL::kOut / PL::kFwdA;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S0P; ++s)
      if (s < ns0) ACT::mma(acc, w[s], ACT::template ldB<LD::X0>(x0, i, kk, s));
    for (int s = S0P; s < ns0; ++s)                   // (wide inputs: fragments fetched here)
      ACT::mma(acc, PL::ldf(pf + (size_t)s * PL::kBlk), ACT::template ldB<LD::X0>(x0, i, kk, s));
    const f32x4 pre = acc * kO + b;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
    ACT::template st4<LD::H>(h1, i, 16 * wave + 4 * kk, f32x4{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), fmaxf(pre[2], 0.f), fmaxf(pre[3], 0.f)}, PL::kFwdA);
    return ok;
  }
};

// Layer 1, member c of EIGHT: the member's two hidden tiles (columns 32 c ..), quartet wave q = (tile q & 1, half q >> 1
// of the 256-deep contraction).  run_partial(): the wave's partial tile; the upper halves go through `scr` (2 x 256
// floats), [barrier], finish_pre(): waves q < 2 add them + bias -> the pre-activation values of the member-LOCAL tile.
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
  // ... ReLU'd and written to h2; false if a value left PrecX2's range
  __device__ __forceinline__ bool finish(const float* scr, float* h2, int q) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    const f32x4 pre = finish_pre(scr, q);
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) ok = ok && PL::range_ok(pre[r]);
    ACT::template st4<LD::L>(h2, i, 16 * q + 4 * kk, f32x4{fmaxf(pre[0], 0.f), fmaxf(pre[1], 0.f), fmaxf(pre[2], 0.f), fmaxf(pre[3], 0.f)}, PL::kFwdA);
    return ok;
  }
};

// Output layer (<= 8 outputs: one tile) over the member's 32 columns + the cluster all-reduce: ONE wave.  The sums
// (+ bias) come back in the wave's registers: lane (kk, i) = outputs 4 kk + r of row i (kk < 2).
template <class PL, int NM>
struct LeanL2 {
  static constexpr bool X2 = PL::kX2;
  static constexpr int W = kW4 / PL::KS, M = 32 / PL::KS;     // 1 / 2 macro steps
  using ACT = LeanAct<X2>;
  using LD = LeanLd<X2>;
  typename PL::Frag w[M];
  f32x4 b;
  __device__ __forceinline__ void load(const float* pf2, const float* b2, int n_out, int c) {
    const int lane = threadIdx.x & 63, kk = lane >> 4;
    PL::template ldfn<M>(w, pf2 + ((size_t)c * M) * PL::kBlk + lane * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = 4 * kk + r < n_out ? PL::ldb(b2 + 4 * kk + r) : 0.f;
  }
  template <class ST = NoStamp>
  __device__ __forceinline__ bool run(const float* h2, int n_out, const Tp& tp, f32x4& out, bool poll = true, ST sf = ST()) {
    const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
    constexpr float kO = PL::kOut / PL::kFwdA;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < M; ++s) ACT::mma(acc, w[s], ACT::template ldB<LD::L>(h2, i, kk, s));
    f32x4 sum;
    const bool in = lean_ar<NM>(acc * kO, 0, n_out, 0, tp, sum, poll, sf);
    out = sum + b;
    return in;
  }
};

// a pass's barrier (workgroup-wide; LDS traffic only — requests to memory stay in flight across it)
__device__ __forceinline__ void lean_bar() { __syncthreads(); }

struct LeanBufs {
  float* x0;        // layer-0 input: PrecX2 planes [2][16][kLd0H] halfs / fp32 rows [16][kX0Ld]
  float* h1;        // [2][16][kLdH] halfs / [16][kWL4] floats
  float* h2;        // member-local: [2][16][kLdLH] halfs / [16][kLdLF] floats
  float* g2;        // the same (scalar_fb)
  float* scr;       // >= 4096 floats
};

// ---------------------------------------------------------------------------------------------------------------
// Scalar-output net (a critic), forward + constant-seed backward to the input columns [dcol0, dcol0 + dcols) (dcols <= 8,
// inside at most two 16-column tiles), on a cluster of EIGHT — tp4_scalar_fb<P, 8>'s arithmetic, one job per wave and stage:
//   all waves : layer 0, tile = wave -> h1                                                      [bar 1]
//   waves 4-7 : layer 1 partial                       [bar 2]   waves 4, 5: h2 and the unit-seed g2 (member-local)   [bar 3]
//   all waves : dz1 partial, k tile = wave, over the member's columns, masked in place over h1  [bar 4]
//   waves 12-15: input-column gradient, a quarter of the 256-deep contraction each -> scr        [bar 5]
//               waves 12 (, 13): tile sums + all-reduce; the sums go to `done(sum, first column of the lane's four, row)`
//               from the registers of the lanes that hold valid columns — no barrier behind them (need_sum false: a
//               member that only contributes publishes and leaves)
//   wave 0 (behind bar 5): q = output layer + all-reduce (for the lead member's q_sum_out; the other members publish only)
// The caller has x0 ready (visible after bar 0 below at the latest) and every wave calls this once.
// tp.stage advances by 2 (q: stage, gradient: stage + 1).
// ---------------------------------------------------------------------------------------------------------------
template <class PL, class ST, class DONE>
__device__ __forceinline__ void lean_scalar_fb(const Net& net, const LeanBufs& L, Tp& tp, int row0, int B, float seed,
                                               int dcol0, int dcols, ST sf, float* q_sum_out, const BiasOv bo, bool need_sum, DONE done) {
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

  // A PROGRAM PER QUARTET (the same barrier sequence in each): what a quartet never holds costs it no registers — one
  // common instruction stream with wave-uniform ifs makes every job's fragments live at once (39 registers spilled).
  auto l0 = [&](LeanL0<PL>& j0) {
    const bool ok = j0.run(L.x0, L.h1, wave);
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    sf();
  };
  // dz1 partial, k tile = wave, masked in place over h1 (a lane rewrites the positions whose masks it holds)
  auto dz1 = [&](const typename PL::Frag (&wz)[M], const f32x4 mk) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < M; ++s) ACT::mma(acc, wz[s], ACT::template ldB<LD::L>(L.g2, i, kk, s));
    f32x4 d;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = mk[r] > 0.f ? acc[r] * ob : 0.f;
    ACT::template st4<LD::H>(L.h1, i, 16 * wave + 4 * kk, d, sb);
    sf();
  };
  auto ld_wz = [&](typename PL::Frag (&wz)[M]) {       // dz1: k tile `wave` of W2^T over the member's 32 columns
    PL::template ldfn<M>(wz, net.pb[1] + ((size_t)wave * W + c * M) * BK + lane * 4);
  };
  auto masks = [&]() { return ACT::template ld4pos<LD::H>(L.h1, i, 16 * wave + 4 * kk); };   // (h1 complete)
  if (quart == 1) {
    LeanL0<PL> j0;
    j0.load(net.pf[0], nb0, net.dims[0], wave);
    typename PL::Frag wz[M];
    ld_wz(wz);
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
    lean_bar();                                   // 0: x0 visible
    l0(j0);
    lean_bar();                                   // 1: h1
    j1.run_partial(L.h1, L.scr, q);
    const f32x4 mk = masks();
    sf();
    lean_bar();                                   // 2: upper halves
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
    lean_bar();                                   // 3: h2, g2
    dz1(wz, mk);
    lean_bar();                                   // 4: dz1 partial (over h1)
    lean_bar();                                   // 5
  } else if (quart == 3) {
    LeanL0<PL> j0;
    j0.load(net.pf[0], nb0, net.dims[0], wave);
    typename PL::Frag wz[M];
    ld_wz(wz);
    typename PL::Frag wd[2][Q4];                  // input-column gradient: tiles dt0 (, dt0 + 1) of W0^T, quarter q
    {
      const float* p = net.pb[0] + ((size_t)dt0 * W + q * Q4) * BK + lane * 4;
      PL::template ldfn<Q4>(wd[0], p);
      PL::template ldfn<Q4>(wd[1], p + (size_t)W * BK, dnt > 1 ? Q4 : 0);
    }
    lean_bar();                                   // 0
    l0(j0);
    lean_bar();                                   // 1
    const f32x4 mk = masks();
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    dz1(wz, mk);
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
    lean_bar();                                   // 5: quarters
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
      f32x4 sum;
      if (lean_ar<NM>(part * ob, f_first, ncols, 192 * q, tp2, sum, need_sum, sf) && need_sum)
        done(sum, t0 + 4 * kk - dcol0, i);        // (column of sum[0] relative to dcol0 — may be negative / beyond dcols: the callee masks)
    }
  } else {
    // quartets 0 and 2: layer 0 and dz1 only; wave 0 also the logged q
    LeanL0<PL> j0;
    j0.load(net.pf[0], nb0, net.dims[0], wave);
    typename PL::Frag wz[M];
    ld_wz(wz);
    LeanL2<PL, NM> j2;
    if (wave == 0) j2.load(net.pf[2], nb2, 1, c);
    lean_bar();                                   // 0
    l0(j0);
    lean_bar();                                   // 1
    const f32x4 mk = masks();
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    dz1(wz, mk);
    lean_bar();                                   // 4
    lean_bar();                                   // 5
    if (wave == 0) {
      // q: logged only — the lead member sums it, the others publish; beside the gradient's exchange, nothing waits for it
      const bool lead = c == 0;
      f32x4 qv;
      const bool in = j2.run(L.h2, 1, tp, qv, lead && q_sum_out != nullptr);
      if (lead && q_sum_out != nullptr) {
        const float qs = row16_sum((in && kk == 0 && row0 + i < B) ? qv[0] : 0.f);
        if (lane == 0) *q_sum_out = qs;
      }
    }
  }
  tp.stage += 2;
  sf();
}

// one WAVE waits for n (<= 192) flag granules {tag, *}; bounded; `wait` false: nothing to wait for
__device__ __forceinline__ void lean_wave_wait(const unsigned long long* flags, int n, unsigned tag, bool wait, unsigned* err, unsigned code) {
  if (!wait) return;
  const int lane = threadIdx.x & 63;
  bool ok = false;
  for (int spin = 0; spin < kTpSpin && !ok; ++spin) {
    bool mine = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int idx = lane + 64 * k;
      if (idx < n) mine = mine && (unsigned)(__hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == tag;
    }
    ok = __builtin_amdgcn_ballot_w64(!mine) == 0ull;
    if (!ok) __builtin_amdgcn_s_sleep(2);
  }
  if (!ok && lane == 0) report_expired(err, code);
}

// ---------------------------------------------------------------------------------------------------------------
// Role A of k_ddpg_chain — a' = tanh(actor_target(s')), q' = critic_target(s', a') — on a cluster of EIGHT.  The target
// critic's weights are final since the critic's tiles of the update before (flags ct): their requests go out BEFORE the
// actor's tiles (flags at) are waited for, into the registers of the waves that use them.
//   all waves   layer 0 of both passes, tile = wave
//   waves 4-7   actor_t layer 1                 | wave 8: actor_t output layer + all-reduce, a' = tanh(.) -> the input tile
//   waves 12-15 critic_t layer 1                | wave 0: critic_t output layer + all-reduce, then TAIL
//   wave 15     the flag waits
// xb: the fp32 rows [s' | .] (a' is written there too); PrecX2: L.x0 = the planes of [s' | 0] (a' is added).
// TAIL (wave 0): request(ctx) before the second pass's layer 1, finish(ctx, q') right behind the exchange — lane i < 16 holds
// row i's q' — the TD target and the seeds, no barrier in front of them.  tp.stage advances by 2.
// ---------------------------------------------------------------------------------------------------------------
struct LeanFlags {
  const unsigned long long* ct; int n_ct;
  const unsigned long long* at; int n_at;
  unsigned tag; bool wait;
  unsigned* err; unsigned code;
};
template <class PL, class TAIL, class ST, class TCTX>
__device__ __forceinline__ void lean_role_a(const Net& actor_t, const Net& critic_t, const BiasOv ba, const BiasOv bc, const LeanBufs& L,
                                            float* xb, Tp& tp, int S, int Ad, int row0, int B, const LeanFlags F,
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
  const float* const x0 = X2 ? L.x0 : xb;
  // A PROGRAM PER QUARTET, the same barrier sequence in each (lean_scalar_fb)
  auto l0 = [&](LeanL0<PL>& j) {
    const bool ok = j.run(x0, L.h1, wave);
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
  };
  auto l1fin = [&](LeanL1<PL>& j1) {
    if (q < 2) {
      if (__builtin_expect(!j1.finish(L.scr, L.h2, q), 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    }
    sf();
  };
  if (quart == 0) {
    // ---- layer 0 of both passes; wave 0: the target critic's output layer, its exchange and the tail
    LeanL0<PL> a0, c0;
    LeanL2<PL, NM> j2;
    TAIL tail;
    lean_bar();                                   // a: the critic's tiles of the update before are done
    c0.load(critic_t.pf[0], cb0, critic_t.dims[0], wave);
    if (q == 0) j2.load(critic_t.pf[2], cb2, 1, c);
    lean_bar();                                   // b: the actor's tiles as well; s' in place
    a0.load(actor_t.pf[0], ab0, actor_t.dims[0], wave);
    l0(a0);
    sf();
    lean_bar();                                   // 1: h1
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    lean_bar();                                   // 4: [s' | a'] complete
    l0(c0);
    if (q == 0) tail.request(tctx);
    sf();
    lean_bar();                                   // 5
    lean_bar();                                   // 6
    lean_bar();                                   // 7: h2 of the second pass
    if (q == 0) {
      Tp tp2 = tp;
      tp2.stage = tp.stage + 1;
      f32x4 o;
      const bool in = j2.run(L.h2, 1, tp2, o, tail.wants(tctx), sf);      // (the members that only contribute publish and leave)
      sf();
      tail.finish(tctx, in && kk == 0, i, o[0]);  // (q' is in this wave's registers: nothing in front of the seeds)
    }
  } else if (quart == 2) {
    // ---- layer 0 of both passes; wave 8: the target actor's output layer, its exchange, a' = tanh(.)
    LeanL0<PL> a0, c0;
    LeanL2<PL, NM> j2;
    lean_bar();                                   // a
    c0.load(critic_t.pf[0], cb0, critic_t.dims[0], wave);
    lean_bar();                                   // b
    a0.load(actor_t.pf[0], ab0, actor_t.dims[0], wave);
    if (q == 0) j2.load(actor_t.pf[2], ab2, Ad, c);
    l0(a0);
    sf();
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3: h2 of the first pass
    if (q == 0) {
      sf();
      f32x4 o;
      if (j2.run(L.h2, Ad, tp, o, true, sf)) {
        // a' = tanh(.) -> the input tile's action columns (rows beyond the batch: zero) — this lane's four consecutive
        // columns S + 4 kk ..
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * kk + r < Ad) {
            const float a = row0 + i < B ? lean_tanh(o[r]) : 0.f;
            xb[i * kX0Ld + S + 4 * kk + r] = a;
            if constexpr (X2) {
              const float v = a * PL::kFwdA;
              const _Float16 h = (_Float16)v;
              _Float16* hp = reinterpret_cast<_Float16*>(L.x0) + i * kLd0H + lean_pos(S + 4 * kk + r);
              hp[0] = h;
              hp[16 * kLd0H] = (_Float16)(v - (float)h);
            }
          }
      }
    }
    sf();
    lean_bar();                                   // 4
    l0(c0);
    sf();
    lean_bar();                                   // 5
    lean_bar();                                   // 6
    lean_bar();                                   // 7
  } else if (quart == 1) {
    // ---- layer 0 of both passes, the target actor's layer 1
    LeanL0<PL> a0, c0;
    LeanL1<PL> j1;
    lean_bar();                                   // a
    c0.load(critic_t.pf[0], cb0, critic_t.dims[0], wave);
    lean_bar();                                   // b
    a0.load(actor_t.pf[0], ab0, actor_t.dims[0], wave);
    j1.load(actor_t.pf[1], ab1, c, q);
    l0(a0);
    sf();
    lean_bar();                                   // 1
    j1.run_partial(L.h1, L.scr, q);
    sf();
    lean_bar();                                   // 2
    l1fin(j1);
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    l0(c0);
    sf();
    lean_bar();                                   // 5
    lean_bar();                                   // 6
    lean_bar();                                   // 7
  } else {
    // ---- layer 0 of both passes, the target critic's layer 1; wave 15: the flag waits
    LeanL0<PL> a0, c0;
    LeanL1<PL> j1;
    if (q == 3) lean_wave_wait(F.ct, F.n_ct, F.tag, F.wait, F.err, F.code);
    lean_bar();                                   // a
    c0.load(critic_t.pf[0], cb0, critic_t.dims[0], wave);
    j1.load(critic_t.pf[1], cb1, c, q);
    if (q == 3) lean_wave_wait(F.at, F.n_at, F.tag, F.wait, F.err, F.code);
    lean_bar();                                   // b
    a0.load(actor_t.pf[0], ab0, actor_t.dims[0], wave);
    l0(a0);
    sf();
    lean_bar();                                   // 1
    lean_bar();                                   // 2
    lean_bar();                                   // 3
    lean_bar();                                   // 4
    l0(c0);
    sf();
    lean_bar();                                   // 5: h1 of the second pass
    j1.run_partial(L.h1, L.scr, q);
    sf();
    lean_bar();                                   // 6
    l1fin(j1);
    lean_bar();                                   // 7
  }
  tp.stage += 2;
  sf();
}

}  // namespace oprl
