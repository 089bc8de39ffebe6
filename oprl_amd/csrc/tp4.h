// tp4.h — the tensor-parallel slice passes of tp3.h, specialised for the headline shape:
// hidden width 256, cluster of 4 CUs, layer-0 fan-in <= 64, output width <= 48.
//
// Why a second implementation.  Stamps inside tp3_forward (tools/trace_slice.py,
// profiles/r01d_stage_stamps.txt) show 1.5-1.8 us between entry and the first barrier of a
// pass and ~1 us of non-MFMA time in every later stage: with 16 waves per CU (4 per SIMD,
// a wave64 VALU op = 4 SIMD cycles) the kernel is bound by INSTRUCTION ISSUE — one generic
// tp3_forward is ~3000 static instructions (runtime integer divisions, shape branches,
// split-contraction scratch reduces, 9 barriers) around 28 executed MFMAs per wave.  Here
// every shape is a compile-time constant, so a pass is a few hundred instructions:
//   forward    [x0 visible] L0: wave w = tile w (<= 4 steps)            -> h1
//              [h1 visible] L1: waves 0..3 own one of the member's 4 tiles over the whole
//                               contraction (two accumulator chains, 64 MFMAs; no scratch
//                               reduce) while waves 4.. store h1 for the dW kernel   -> h2
//              [h2 visible] L2: wave 4+t = output tile t over the member's 64 columns, then
//                               the cluster all-reduce STRAIGHT FROM ITS REGISTERS
//                               (publish 4 granules per lane, poll the 3 peers)      -> out
//              [out visible]
//   backward   [dout visible] waves 0..3: dz2 tile (<= 3 steps), ReLU mask in place (h2)
//              [dz2 visible]  all waves: dz1-partial tile w (4 steps), mask in place (h1)
//              [dz1 visible]  stores for dW; optional input-column gradient: 4 waves per
//                             tile split the contraction, one b128 scratch exchange, then
//                             all-reduce from registers
// The weight fragments of ALL stages of a pass are requested at entry (<= 24 b128 per lane).
// Summation order: members in index order (as tp3.h), inside a member even/odd-step chains.
#pragma once
#include "tp3.h"

namespace oprl {

constexpr int kW4 = 256;
constexpr int kWL4 = lds_ld(kW4);
constexpr int kTpc4 = 4;                      // 16-column tiles per member
constexpr int kCols4 = kTpc4 * 16;            // hidden columns per member
constexpr int kMaxS0 = 6;                     // layer-0 fp32 macro steps: fan-in <= 96 (humanoid S + A = 88)

// Every pass is a template over the precision policy P (engine.h): the stage structure, barriers,
// exchanges and epilogues are the same, only the macro step changes — 16 contraction indices and
// four fp32 MFMAs (PrecF32, the parity mode) or 32 indices and one bf16 MFMA (PrecBF16).  Step
// counts of the fixed shapes:
template <class P> struct Tp4Steps {
  static constexpr int W = kW4 / P::KS;       // the whole 256-deep contraction: 16 / 8
  static constexpr int M = kCols4 / P::KS;    // a member's 64 columns: 4 / 2
  static constexpr int S0 = 16 * kMaxS0 / P::KS;   // widest layer-0 input: 6 / 3
  static constexpr int O = (kNarrowMax + P::KS - 1) / P::KS;   // widest output: 3 / 2
};

// The cluster size NM (members per slice) is a template parameter of the forward passes: 4 (the default; the
// backward and every dz1-partial producer) or 8 (forward / scalar-critic passes of workgroups that would
// otherwise idle: DDPG's role A and phase 2).  A member then owns TPM 16-column tiles of the hidden layer and
// takes in (and multiplies) 1 / NM of the 256 x 256 layer; layer 0 stays replicated.
template <class P, int NM> struct Tp4Shape {
  static_assert(NM == 4 || NM == 8, "cluster sizes of the lean passes");
  static constexpr int TPM = 16 / NM;                    // 16-column tiles per member: 4 / 2
  static constexpr int COLS = 16 * TPM;                  // hidden columns per member: 64 / 32
  static constexpr int KP = 16 / TPM;                    // layer 1: waves (contraction parts) per tile: 4 / 8
  static constexpr int NQ = Tp4Steps<P>::W / KP;         // macro steps of one part
  static constexpr int KW = kW4 / KP;                    // floats of one part: 64 / 32
  static constexpr int M = COLS / P::KS;                 // macro steps over the member's columns
  static_assert(NQ >= 1 && M >= 1, "bf16 macro steps are 32 wide: clusters of 8 need the fp32 policy");
};

// sum over N macro steps, even steps on one accumulator and odd ones on another (two independent MFMA chains)
// (`sc`: the power of two the A operand is multiplied by on its way in — PrecX2; the other policies ignore it)
template <class P, int N>
__device__ __forceinline__ f32x4 tp4_mac_steps(const float* xr, const typename P::Frag (&w)[N], float sc = P::kFwdA) {
  f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
  for (int s = 0; s < N; ++s) {
    if (s & 1) P::mac_s(xr, s, w[s], a1, sc);
    else P::mac_s(xr, s, w[s], a0, sc);
  }
  return a0 + a1;
}
// the largest magnitude of the [kR][ncols16 * 16] block of an LDS tile (leading dim kOutLd), in every lane of the wave
__device__ __forceinline__ float tp4_tile_amax(const float* T, int ncols16) {
  const int lane = threadIdx.x & 63;
  float m = 0.f;
  for (int e = lane; e < kR * 4 * ncols16; e += 64) {        // float4 units: 4 per 16 columns
    const int row = e / (4 * ncols16), q = e - row * (4 * ncols16);
    const f32x4 v = ld4(T + row * kOutLd + 4 * q);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}

// Biases from somewhere else than the net's master arena (k_ddpg_chain's later updates: the uncached copies the
// tiles of the update before left); null members = the net's own.
struct BiasOv { const float* b0 = nullptr; const float* b1 = nullptr; const float* b2 = nullptr; };
// tp4_scalar_fb, clusters of four: q leaves as the members' PARTIAL sums — granules out[member * kR + row] {tag, value},
// no exchange inside the pass — for a consumer that adds them up itself in member order (+ the output bias): role B of
// k_ddpg_chain, whose q only role A reads.  out == null: the cluster all-reduce.
struct QPart { unsigned long long* out = nullptr; unsigned tag = 0; };

__host__ __device__ inline bool tp4_shape_ok(int width, int fan_in, int n_out) {
  return width == kW4 && fan_in <= 16 * kMaxS0 && n_out <= kNarrowMax;
}

__device__ __forceinline__ void mac4(const f32x4 a, const f32x4 b, f32x4& acc) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = mfma4(a[t], b[t], acc);
}

// All-reduce of one 16x16 tile held in MFMA accumulator layout (lane (kk,i): rows 4kk..4kk+3,
// column i) over the NM members; `valid` lanes take part, `col` (< kNarrowMax, distinct per
// valid (tile, i)) names the lane's column.  Returns sum over members in member order.
// Granule slot of (row = 4kk + r, col): ((r*4 + kk) * kNarrowMax + col)  (< kTpBlk).
template <int NM = 4>
__device__ __forceinline__ f32x4 tp4_allreduce_regs(const f32x4 mine, int col, bool valid, const Tp& tp) {
  const int kk = (threadIdx.x & 63) >> 4;
  constexpr int kRs = 4 * kNarrowMax;   // granules between two r
  const unsigned tag = (tp.tag << 6) | (unsigned)(tp.stage & 63);
  unsigned long long* slot = tp.xbuf + (size_t)tp.stage * NM * kTpBlk + (valid ? kk * kNarrowMax + col : 0);
  f32x4 sum = mine;
  if (valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (tp.local)
        __hip_atomic_store(slot + (size_t)tp.c * kTpBlk + r * kRs,
                           ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine[r]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_store(slot + (size_t)tp.c * kTpBlk + r * kRs,
                           ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine[r]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // all peers' granules are requested together (one round trip per poll), then checked; the values are
    // summed straight from the granules in member order (this member's own term from its registers)
    bool ok = false;
    unsigned long long x[NM][4];
    for (int spin = 0; spin < tp.spin && !ok; ++spin) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          x[m][r] = (m == tp.c) ? ((unsigned long long)tag << 32)
                                : __hip_atomic_load(slot + (size_t)m * kTpBlk + r * kRs, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ok = ok && (unsigned)(x[m][r] >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (ok) {
      sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] += (m == tp.c) ? mine[r] : __uint_as_float((unsigned)x[m][r]);
    } else {
      report_expired(tp.err, tp.err_code | SITE_CLUSTER);
      const float nan = __builtin_nanf("");
      sum = f32x4{nan, nan, nan, nan};
    }
  }
  return sum;
}

// The same all-reduce for NARROW tiles (<= 8 valid columns: a critic's q, an action-sized output or input
// gradient) with one ELEMENT per lane slot instead of four rows per lane: the tile goes through 1 KB of
// wave-private LDS, lane L takes elements L and L + 64 of the 16 x ncols valid block — at most two granules per
// peer and lane, so that a cluster of eight needs no more registers than a cluster of four does above.  Valid
// columns = lanes i in [i_first, i_first + ncols); they are exchanged as columns cc_first .. of the granule block
// and written to out[row * kOutLd + cc_first + k] (+ bias[k], nullable).  Same slots as tp4_allreduce_regs.
// (tp4_narrow_elem: the k-th valid column of lane slot j — what the caller needs to fetch a per-column bias EARLY.)
__device__ __forceinline__ int tp4_narrow_elem(int j, int ncols) {
  const int e = (int)(threadIdx.x & 63) + 64 * j;
  return e < 16 * ncols ? e % ncols : -1;
}
template <int NM>
__device__ __forceinline__ void tp4_allreduce_narrow(const f32x4 mine, int i_first, int ncols, int cc_first,
                                                     float* wscr, float b0, float b1, float* out, const Tp& tp) {
  const int lane = threadIdx.x & 63, i = lane & 15, kk = lane >> 4;
  constexpr int kRs = 4 * kNarrowMax;
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
    const int row = have[j] ? e / ncols : 0, k = have[j] ? e - row * ncols : 0;
    val[j] = wscr[row * 16 + i_first + k];
    off[j] = ((row & 3) * 4 + (row >> 2)) * kNarrowMax + cc_first + k;     // (r * 4 + kk) * kNarrowMax + col
    oidx[j] = row * kOutLd + cc_first + k;
    if (have[j]) {
      if (tp.local)
        __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off[j],
                           ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(val[j]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else
        __hip_atomic_store(base + (size_t)tp.c * kTpBlk + off[j],
                           ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(val[j]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  (void)kRs;
  if (have[0]) {
    bool ok = false;
    unsigned long long x[NM][2];
    for (int spin = 0; spin < tp.spin && !ok; ++spin) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          x[m][j] = (m == tp.c || !have[j]) ? ((unsigned long long)tag << 32)
                                            : __hip_atomic_load(base + (size_t)m * kTpBlk + off[j], __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) ok = ok && (unsigned)(x[m][j] >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) report_expired(tp.err, tp.err_code | SITE_CLUSTER);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < NM; ++m) sum += (m == tp.c) ? val[j] : __uint_as_float((unsigned)x[m][j]);
      if (have[j]) out[oidx[j]] = ok ? sum + bv[j] : __builtin_nanf("");
    }
  }
}

// a 16-byte store of what k_dw_adam reads: plain, or written through (Tp3Store::wt)
__device__ __forceinline__ void tp4_st4(float* base, size_t off, const f32x4 v, bool wt) {
  if (wt) st16_agent_at(base, (unsigned)off, v);
  else *reinterpret_cast<f32x4*>(base + off) = v;
}

// This member's dz1 partial (the whole [16 x 256] tile in h1) -> its partial buffer, for k_dw_adam.
// Tile-major when st.dY0_tile_rows > 0: a wave writes one 16-column tile of the slice, 16 rows x 64 B = one
// contiguous KB.
__device__ __forceinline__ void tp4_store_dz1(const Tp3Store& st, int c, const float* h1, int row0, int B) {
  if (st.dY0 == nullptr) return;
  float* dst = st.dY0 + (size_t)c * st.dY0_stride;
  const int idx = threadIdx.x;                       // 16 rows x 64 float4
  if (st.dY0_tile_rows > 0) {
    const int t = idx >> 6, row = (idx >> 2) & 15, c4 = (idx & 3) * 4, gr = row0 + row;
    if (gr < B) tp4_st4(dst, ((size_t)t * st.dY0_tile_rows + gr) * 16 + c4, ld4(h1 + row * kWL4 + 16 * t + c4), st.wt);
  } else {
    const int row = idx >> 6, col = (idx & 63) * 4, gr = row0 + row;
    if (gr < B) tp4_st4(dst, (size_t)gr * kW4 + col, ld4(h1 + row * kWL4 + col), st.wt);
  }
}

// Layer 1 (the member's 256 x 64 shard) is contracted by ALL 16 waves: wave w = output tile (w & 3), quarter
// (w >> 2) of the contraction, the four partial tiles meeting in `scr` ([kWaves][256] floats) and summed in
// quarter order by one thread per element.  An earlier version gave whole tiles to waves 0..3: 16 fragments
// (64 VGPRs) per lane in four waves while twelve waves idled, and — the fragment requests sitting under a
// wave-id branch — hipcc's wait-count pass could not tell how many younger loads follow the layer-0
// fragments, so layer 0 waited for ALL of the pass's fragments (s_waitcnt vmcnt(1)).  With uniform,
// unconditional requests layer 0 starts as soon as ITS fragments are in.
// `pre` (optional): what the caller has to do before x0 is complete — staging the slice's rows — run AFTER the pass's
// fragment requests are out, so that the two round trips overlap instead of following each other (r05-14)
template <class P = PrecF32, int NM = 4, class ST = NoStamp, class PRE = NoStamp>
__device__ __forceinline__ void tp4_forward(const Net& net, const float* x0s, float* h1, float* h2,
                                            float* outS, float* scr, Tp& tp, const Tp3Store& st, int row0, int B,
                                            ST sf = ST(), const BiasOv bo = BiasOv(), PRE pre = PRE()) {
  using NS = Tp4Steps<P>;
  using SH = Tp4Shape<P, NM>;
  const float* const nb0 = bo.b0 != nullptr ? bo.b0 : net.b[0];
  const float* const nb1 = bo.b1 != nullptr ? bo.b1 : net.b[1];
  const float* const nb2 = bo.b2 != nullptr ? bo.b2 : net.b[2];
  // the net's pointers and dims, fetched from the kernel arguments TOGETHER (one scalar wait): left to the
  // compiler they come in where first used, a scalar load and a wait in front of every group of fragment requests
  asm volatile("" :: "s"(net.pf[0]), "s"(net.pf[1]), "s"(net.pf[2]), "s"(nb0), "s"(nb1), "s"(nb2),
               "s"(net.dims[0]), "s"(net.dims[3]));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int c = tp.c, c0 = c * SH::COLS;
  const int N = net.dims[3];
  const int NS0 = (net.dims[0] + P::KS - 1) / P::KS, NTo = (N + 15) >> 4;
  const int t2 = wave - 4;                    // output tile of waves 4..4+NTo-1
  const bool l2_wave = t2 >= 0 && t2 < NTo;

  constexpr int NQ = SH::NQ;                  // macro steps of one part of layer 1's contraction
  const int t1 = wave % SH::TPM, kq = wave / SH::TPM;   // layer 1: this wave's tile of the member and contraction part
  // the layer-1 element this thread finishes (threads < 256 TPM): tile rt, accumulator slot (rl, rr) = row 4 (rl >> 4) + rr, column rl & 15
  const int rt = (int)threadIdx.x >> 8, rl = ((int)threadIdx.x & 255) >> 2, rr = (int)threadIdx.x & 3;
  const bool r_mine = rt < SH::TPM;

  // ---- requests for the whole pass (uniform over the waves but for the few output-layer fragments)
  using F = typename P::Frag;
  constexpr int BK = P::kBlk;                 // floats between two (tile, step) blocks of a pack
  constexpr float kO = P::kOut / P::kFwdA;    // accumulator -> product (1 but for PrecX2)
  F w0[NS::S0], w1[NQ], w2[SH::M];
  {
    const float* p0 = net.pf[0] + (size_t)wave * NS0 * BK + lane * 4;
    P::template ldfn<NS::S0>(w0, p0, NS0);
  }
  const float bias0 = P::ldb(nb0 + 16 * wave + i);
  {
    const float* p1 = net.pf[1] + ((size_t)(c * SH::TPM + t1) * NS::W + kq * NQ) * BK + lane * 4;
    P::template ldfn<NQ>(w1, p1);
  }
  const float bias1 = P::ldb(nb1 + c0 + 16 * (r_mine ? rt : 0) + (rl & 15));
  float bias2 = 0.f, bias2e[2] = {0.f, 0.f};   // output bias: of this lane's column / of its narrow-exchange elements
#pragma unroll
  for (int s = 0; s < SH::M; ++s) w2[s] = P::zf();
  if (l2_wave) {
    const float* p2 = net.pf[2] + ((size_t)t2 * NS::W + c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(w2, p2);
    if (16 * t2 + i < N) bias2 = P::ldb(nb2 + 16 * t2 + i);
    if constexpr (NM == 8) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = tp4_narrow_elem(j, N);
        if (k >= 0) bias2e[j] = P::ldb(nb2 + k);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  pre();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();   // x0 visible

  // ---- L0: tile = wave
  {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr = x0s + i * kX0Ld + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::S0; ++s)
      if (s < NS0) P::mac_s(xr, s, w0[s], acc, P::kFwdA);
    float* o = h1 + (kk * 4) * kWL4 + 16 * wave + i;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre = acc[r] * kO + bias0;
      ok = ok && P::range_ok(pre);
      o[r * kWL4] = fmaxf(pre, 0.f);
    }
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
  }
  sf();
  __syncthreads();   // h1 visible

  // ---- L1: every wave one part of one tile's contraction -> scr; member 0's threads store h1 for the dW kernel
  *reinterpret_cast<f32x4*>(scr + ((size_t)wave * 64 + lane) * 4) =
      tp4_mac_steps<P, NQ>(h1 + i * kWL4 + SH::KW * kq + 4 * kk, w1);
  if (st.X1 != nullptr && c == 0) {
    const int row = (int)threadIdx.x >> 6, col = ((int)threadIdx.x & 63) * 4, gr = row0 + row;     // 16 rows x 64 float4
    if (gr < B) tp4_st4(st.X1, (size_t)gr * kW4 + col, ld4(h1 + row * kWL4 + col), st.wt);
  }
  sf();
  __syncthreads();   // partial tiles visible
  if (r_mine) {
    const float* sp = scr + ((size_t)rt * 64 + rl) * 4 + rr;      // wave (part q, tile rt) = TPM q + rt
    float v = sp[0];
#pragma unroll
    for (int q = 1; q < SH::KP; ++q) v += sp[q * SH::TPM * 256];
    const float pre = v * kO + bias1;
    if (__builtin_expect(!P::range_ok(pre), 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    h2[(4 * (rl >> 4) + rr) * kWL4 + c0 + 16 * rt + (rl & 15)] = fmaxf(pre, 0.f);
  }
  __syncthreads();   // the member's h2 columns visible

  // ---- L2 partial + all-reduce from registers on waves 4..; waves 8.. store the h2 columns
  if (l2_wave) {
    const f32x4 part = tp4_mac_steps<P, SH::M>(h2 + i * kWL4 + c0 + 4 * kk, w2) * kO;
    const int col = 16 * t2 + i;
    const bool valid = col < N;
    float* o = outS + (kk * 4) * kOutLd + col;
    if constexpr (NM == 8) {
      // clusters of eight serve outputs of at most 8 columns (one tile, this wave): padding columns zeroed
      // here, the valid ones written by the narrow exchange
      if (!valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * kOutLd] = 0.f;
      }
      tp4_allreduce_narrow<NM>(part, 0, N, 0, scr + wave * 256, bias2e[0], bias2e[1], outS, tp);
    } else {
      const f32x4 sum = tp4_allreduce_regs<NM>(part, col, valid, tp);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * kOutLd] = valid ? sum[r] + bias2 : 0.f;
    }
  } else if (st.X2 != nullptr && wave >= 8 && wave < 8 + SH::TPM) {
    constexpr int C4 = SH::COLS / 4;                 // float4 per row of the member's columns
    const int idx = (int)threadIdx.x - 512;          // 16 rows x C4 float4
    const int row = idx / C4, col = c0 + (idx - row * C4) * 4, gr = row0 + row;
    if (gr < B) tp4_st4(st.X2, (size_t)gr * kW4 + col, ld4(h2 + row * kWL4 + col), st.wt);
  }
  tp.stage += 1;
  sf();
  __syncthreads();   // out visible
  sf();
}

// `pre` (optional): what still has to happen before dout is complete — the loss-gradient seed of a backward that rides on
// the launch producing its input (slice_tp_body.h, r06-16) — run AFTER the pass's fragment requests are out
template <class P = PrecF32, class ST = NoStamp, class PRE = NoStamp>
__device__ __forceinline__ void tp4_backward(const Net& net, const float* doutS, float* h1, float* h2,
                                             float* scr, Tp& tp, const Tp3Store& st, int row0, int B,
                                             int dact_col0, int dact_cols, float* dactS, ST sf = ST(), PRE pre = PRE()) {
  using NS = Tp4Steps<P>;
  asm volatile("" :: "s"(net.pb[0]), "s"(net.pb[1]), "s"(net.pb[2]), "s"(net.dims[3]));   // (as in tp4_forward)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int c = tp.c, c0 = c * kCols4;
  const int No16 = (net.dims[3] + 15) >> 4;                  // valid 16-column groups of dout
  const int NSo = (net.dims[3] + P::KS - 1) / P::KS;
  const bool dact = dact_cols > 0;
  // input-column gradient: only the 16-column tiles that overlap [dact_col0, +dact_cols)
  // (at most 4: one wave per tile and contraction quarter)
  const int dt0 = dact_col0 >> 4, dnt = dact ? ((dact_col0 + dact_cols - 1) >> 4) - dt0 + 1 : 0;
  const int dt = wave >> 2, dpart = wave & 3;       // dact: tile (relative), contraction quarter
  const bool dact_wave = dact && dt < dnt;

  // ---- requests
  using F = typename P::Frag;
  constexpr int BK = P::kBlk;
  F wo[NS::O], wz[NS::M], wd[NS::M];
#pragma unroll
  for (int s = 0; s < NS::O; ++s) wo[s] = P::zf();
  if (wave < kTpc4) {
    const float* q2 = net.pb[2] + (size_t)(c * kTpc4 + wave) * NSo * BK + lane * 4;
    P::template ldfn<NS::O>(wo, q2, NSo);
  }
  {
    const float* q1 = net.pb[1] + ((size_t)wave * NS::W + c * NS::M) * BK + lane * 4;
    P::template ldfn<NS::M>(wz, q1);
  }
#pragma unroll
  for (int s = 0; s < NS::M; ++s) wd[s] = P::zf();
  if (dact_wave) {
    const float* q0 = net.pb[0] + ((size_t)(dt0 + dt) * NS::W + dpart * NS::M) * BK + lane * 4;
    P::template ldfn<NS::M>(wd, q0);
  }
  __builtin_amdgcn_sched_barrier(0);
  pre();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();   // dout visible
  // PrecX2: the gradient tiles go into the MFMAs scaled by a power of two fixed by the largest seed of the slice
  // (every wave looks for itself: twelve of the sixteen idle through the first stage anyway; all members and waves
  // find the same value).  dz2 = dout W3^T stays below |dout| sum|W3| and dz1 below 256 max|dz2| max|W2|: the later
  // stages get 2^-2 of the headroom back.
  float s1 = 1.f, s2 = 1.f;
  if constexpr (P::kX2) {
    s1 = P::a_scale(tp4_tile_amax(doutS, No16));
    s2 = 0.25f * s1;
  }
  const float o1 = P::kOut / s1, o2 = P::kOut / s2;

  // ---- dz2[:, mine] = (dout · W3^T)[:, mine] ⊙ (h2 > 0), in place (waves 0..3)
  if (wave < kTpc4) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* dr = doutS + i * kOutLd + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::O; ++s)
      if (s < NSo) P::mac_tail_s(dr, s, wo[s], acc, No16, s1);
    float* p = h2 + (kk * 4) * kWL4 + c0 + 16 * wave + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * kWL4] = p[r * kWL4] > 0.f ? acc[r] * o1 : 0.f;
  }
  sf();
  __syncthreads();   // dz2 visible

  // ---- dz1 partial: tile = wave, contraction over the member's columns; mask in place (h1)
  {
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
    const float* hr = h2 + i * kWL4 + c0 + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::M; s += 2) {
      P::mac_s(hr, s, wz[s], a0, s1);
      P::mac_s(hr, s + 1, wz[s + 1], a1, s1);
    }
    float* p = h1 + (kk * 4) * kWL4 + 16 * wave + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * kWL4] = p[r * kWL4] > 0.f ? (a0[r] + a1[r]) * o1 : 0.f;
  }
  if (st.dY1 != nullptr && wave >= 12) {
    const int idx = (int)threadIdx.x - 768;          // 16 rows x 16 float4 of dz2
    const int row = idx >> 4, col = c0 + (idx & 15) * 4, gr = row0 + row;
    if (gr < B) tp4_st4(st.dY1, (size_t)gr * kW4 + col, ld4(h2 + row * kWL4 + col), st.wt);
  }
  sf();
  __syncthreads();   // dz1 partial visible

  tp4_store_dz1(st, c, h1, row0, B);
  if (dact) {
    // partial gradient wrt the input columns: tile dt, contraction quarter dpart
    if (dact_wave) {
      f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
      const float* hr = h1 + i * kWL4 + 64 * dpart + 4 * kk;
#pragma unroll
      for (int s = 0; s < NS::M; s += 2) {
        P::mac_s(hr, s, wd[s], a0, s2);
        P::mac_s(hr, s + 1, wd[s + 1], a1, s2);
      }
      *reinterpret_cast<f32x4*>(scr + wave * 256 + lane * 4) = (a0 + a1) * o2;
    }
    sf();
    __syncthreads();
    if (wave < dnt) {        // wave t gathers tile t's four quarters, then the cluster all-reduce
      f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) part += ld4(scr + (wave * 4 + q) * 256 + lane * 4);
      const int cc = 16 * (dt0 + wave) + i - dact_col0;
      const bool valid = cc >= 0 && cc < dact_cols;
      const f32x4 sum = tp4_allreduce_regs(part, cc, valid, tp);
      if (valid) {
        float* o = dactS + (kk * 4) * kOutLd + cc;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * kOutLd] = sum[r];
      }
    }
    tp.stage += 1;
    sf();
    __syncthreads();
  }
  sf();
}

// ---------------------------------------------------------------------------------------
// Forward AND backward of a SCALAR-OUTPUT net (a critic) in one pass, seeded with a constant.
//
// For N = 1 the backward is linear in the per-row seed d[row] = dLoss/dq[row]:
//     dz2[row, j] = d[row] * W3[j] * (h2[row, j] > 0)          (elementwise, no GEMM)
//     dz1[row, :] = d[row] * ((W3 ⊙ mask2[row]) · W2) ⊙ mask1[row]
// so the pass runs with the constant `seed` in place of d:
//   - DDPG phase 2 (actor loss -mean Q, d = -1/B for every row): exact, and q itself — the
//     output layer + cluster all-reduce — is only logged, so it leaves the critical path;
//   - DDPG phase 1 role B (d = 2(q - y)/B): seed = 1 BEFORE the TD target y has arrived;
//     k_dw_adam multiplies the stored unit-seed dz rows by d[row] (DwItem::scaled).  The
//     product order differs from autograd's by one rounding per element (<= 1.2e-7 rel.).
// Stages: [x0] L0 -> h1   [h1] L1 -> h2 and g2 = unit dz2 (waves 0..3; the rest store h1)
//         [h2,g2] dz1 partial, mask in place (h1); stores of h2 / g2 columns
//         [dz1] dz1-partial store; input-column gradient quarters (if wanted); wave 12:
//               output layer + all-reduce from registers -> outS
//         [quarters] gather + all-reduce -> dactS            (if wanted)
// g2: one more [kR][kWL4] LDS buffer.  tp.stage advances by 1 (2 with dact).
// ---------------------------------------------------------------------------------------
template <class P = PrecF32, int NM = 4, class ST = NoStamp, class PRE = NoStamp>
__device__ __forceinline__ void tp4_scalar_fb(const Net& net, const float* x0s, float* h1, float* h2,
                                              float* g2, float* outS, float* scr, Tp& tp,
                                              const Tp3Store& st, int row0, int B, float seed,
                                              int dact_col0, int dact_cols, float* dactS, ST sf = ST(),
                                              float* q_sum_out = nullptr, const BiasOv bo = BiasOv(), const QPart qp = QPart(),
                                              PRE pre = PRE()) {
  const float* const nb0 = bo.b0 != nullptr ? bo.b0 : net.b[0];
  const float* const nb1 = bo.b1 != nullptr ? bo.b1 : net.b[1];
  const float* const nb2 = bo.b2 != nullptr ? bo.b2 : net.b[2];
  // q_sum_out (optional, one float): sum of q over the slice's valid rows, written by the wave
  // that finishes the q all-reduce (diagnostics without a barrier on the main path)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  using NS = Tp4Steps<P>;
  using SH = Tp4Shape<P, NM>;
  asm volatile("" :: "s"(net.pf[0]), "s"(net.pf[1]), "s"(net.pf[2]), "s"(net.pb[0]), "s"(net.pb[1]), "s"(net.pb[2]),
               "s"(nb0), "s"(nb1), "s"(nb2), "s"(net.dims[0]));   // (as in tp4_forward)
  const int c = tp.c, c0 = c * SH::COLS;
  const int NS0 = (net.dims[0] + P::KS - 1) / P::KS;   // layer-0 steps
  const bool dact = dact_cols > 0;
  const int dt0 = dact_col0 >> 4, dnt = dact ? ((dact_col0 + dact_cols - 1) >> 4) - dt0 + 1 : 0;
  const int dt = wave >> 2, dpart = wave & 3;
  const bool dact_wave = dact && dt < dnt;
  constexpr int kOutWave = 12;

  constexpr int NQ = SH::NQ;                  // layer 1 as in tp4_forward: wave = (tile, contraction part)
  constexpr int Q4 = NS::W / 4;               // macro steps of a QUARTER of a 256-deep contraction (the input-column gradient)
  const int t1 = wave % SH::TPM, kq = wave / SH::TPM;
  const int rt = (int)threadIdx.x >> 8, rl = ((int)threadIdx.x & 255) >> 2, rr = (int)threadIdx.x & 3;
  const bool r_mine = rt < SH::TPM;

  // ---- requests
  using F = typename P::Frag;
  constexpr int BK = P::kBlk;
  constexpr float kO = P::kOut / P::kFwdA;    // accumulator -> product, forward stages (1 but for PrecX2)
  // PrecX2: the unit-seed gradient tiles g2 = seed w3 (h2 > 0) and dz1 = g2 W2 go in scaled by 2^12 / |seed| (a
  // power of two): |w3| < 16, |dz1| <= |seed| max|w3| sum|W2 column| < 16 |seed| stay inside fp16's range
  float sb = 1.f;
  if constexpr (P::kX2) sb = 4.f * P::a_scale(fabsf(seed));
  const float ob = P::kOut / sb;
  F w0[NS::S0], w1[NQ], w2[SH::M], wz[SH::M], wd[Q4];
  {
    const float* p0 = net.pf[0] + (size_t)wave * NS0 * BK + lane * 4;
    P::template ldfn<NS::S0>(w0, p0, NS0);
  }
  const float bias0 = P::ldb(nb0 + 16 * wave + i);
  {
    const float* p1 = net.pf[1] + ((size_t)(c * SH::TPM + t1) * NS::W + kq * NQ) * BK + lane * 4;
    P::template ldfn<NQ>(w1, p1);
  }
  const float bias1 = P::ldb(nb1 + c0 + 16 * (r_mine ? rt : 0) + (rl & 15));
  const float w3 = P::first(net.pb[2] + (size_t)(c * SH::TPM + (r_mine ? rt : 0)) * BK + (rl & 15) * 4);   // W3[c0 + 16 rt + col]  (one step)
  float bias2 = 0.f;
#pragma unroll
  for (int s = 0; s < SH::M; ++s) w2[s] = P::zf();
  if (wave == kOutWave) {
    const float* p2 = net.pf[2] + ((size_t)c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(w2, p2);
    if (i == 0 || NM == 8) bias2 = P::ldb(nb2);
  }
  __builtin_amdgcn_sched_barrier(0);
  pre();             // (as in tp4_forward)
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();   // x0 visible

  // ---- L0
  {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr = x0s + i * kX0Ld + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::S0; ++s)
      if (s < NS0) P::mac_s(xr, s, w0[s], acc, P::kFwdA);
    float* o = h1 + (kk * 4) * kWL4 + 16 * wave + i;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre = acc[r] * kO + bias0;
      ok = ok && P::range_ok(pre);
      o[r * kWL4] = fmaxf(pre, 0.f);
    }
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
  }
  {
    const float* q1 = net.pb[1] + ((size_t)wave * NS::W + c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(wz, q1);
  }
  sf();
  __syncthreads();   // h1 visible

  // ---- L1 (every wave a quarter of one tile's contraction -> scr); member 0's threads store h1
  *reinterpret_cast<f32x4*>(scr + ((size_t)wave * 64 + lane) * 4) =
      tp4_mac_steps<P, NQ>(h1 + i * kWL4 + SH::KW * kq + 4 * kk, w1);
  if (st.X1 != nullptr && c == 0) {
    const int row = (int)threadIdx.x >> 6, col = ((int)threadIdx.x & 63) * 4, gr = row0 + row;
    if (gr < B) tp4_st4(st.X1, (size_t)gr * kW4 + col, ld4(h1 + row * kWL4 + col), st.wt);
  }
  sf();
  __syncthreads();   // partial tiles visible
  if (r_mine) {   // ... summed in part order, bias + ReLU, and the unit-seed dz2 of the same element
    const float* sp = scr + ((size_t)rt * 64 + rl) * 4 + rr;
    float sum = sp[0];
#pragma unroll
    for (int q = 1; q < SH::KP; ++q) sum += sp[q * SH::TPM * 256];
    const float pre = sum * kO + bias1;
    if (__builtin_expect(!P::range_ok(pre), 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    const float v = fmaxf(pre, 0.f);
    const int off = (4 * (rl >> 4) + rr) * kWL4 + c0 + 16 * rt + (rl & 15);
    h2[off] = v;
    g2[off] = v > 0.f ? seed * w3 : 0.f;
  }
#pragma unroll
  for (int s = 0; s < Q4; ++s) wd[s] = P::zf();
  if (dact_wave) {
    const float* q0 = net.pb[0] + ((size_t)(dt0 + dt) * NS::W + dpart * Q4) * BK + lane * 4;
    P::template ldfn<Q4>(wd, q0);
  }
  sf();
  __syncthreads();   // h2, g2 (the member's columns) visible

  // ---- dz1 partial (unit seed), mask in place over h1; then the h2 / g2 column stores
  {
    const f32x4 a = tp4_mac_steps<P, SH::M>(g2 + i * kWL4 + c0 + 4 * kk, wz, sb) * ob;
    float* p = h1 + (kk * 4) * kWL4 + 16 * wave + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * kWL4] = p[r * kWL4] > 0.f ? a[r] : 0.f;
  }
  if ((wave >= 4 && wave < 4 + SH::TPM) || (wave >= 8 && wave < 8 + SH::TPM)) {
    float* dstg = wave < 8 ? st.X2 : st.dY1;
    if (dstg != nullptr) {
      constexpr int C4 = SH::COLS / 4;                               // float4 per row of the member's columns
      const float* src = wave < 8 ? h2 : g2;
      const int idx = (int)threadIdx.x - (wave < 8 ? 256 : 512);   // 16 rows x C4 float4
      const int row = idx / C4, col = c0 + (idx - row * C4) * 4, gr = row0 + row;
      if (gr < B) tp4_st4(dstg, (size_t)gr * kW4 + col, ld4(src + row * kWL4 + col), st.wt);
    }
  }
  sf();
  __syncthreads();   // dz1 partial visible

  tp4_store_dz1(st, c, h1, row0, B);
  if (dact_wave)
    *reinterpret_cast<f32x4*>(scr + wave * 256 + lane * 4) = tp4_mac_steps<P, Q4>(h1 + i * kWL4 + 64 * dpart + 4 * kk, wd, sb) * ob;
  if (dact) {
    sf();
    __syncthreads();   // quarters visible
  }
  // q (wave 12) and the input-column gradient (waves < dnt) finish side by side: both are one
  // hop of the cluster exchange, neither waits for the other
  if (wave == kOutWave) {
    const f32x4 qpart = tp4_mac_steps<P, SH::M>(h2 + i * kWL4 + c0 + 4 * kk, w2) * kO;
    const bool valid = i == 0;
    float* o = outS + (kk * 4) * kOutLd + i;
    if constexpr (NM == 8) {
      if (!valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * kOutLd] = 0.f;
      }
      tp4_allreduce_narrow<NM>(qpart, 0, 1, 0, scr + wave * 256, bias2, 0.f, outS, tp);   // lanes 0..15: one row each
      if (q_sum_out != nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float qv = (lane < kR && row0 + lane < B) ? outS[lane * kOutLd] : 0.f;
        const float qs = row16_sum(qv);
        if (lane == 0) *q_sum_out = qs;
      }
    } else if (qp.out != nullptr) {
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __hip_atomic_store(qp.out + tp.c * kR + 4 * kk + r,
                             ((unsigned long long)qp.tag << 32) | (unsigned long long)__float_as_uint(qpart[r]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
    const f32x4 sum = tp4_allreduce_regs<NM>(qpart, i, valid, tp);
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r * kOutLd] = valid ? sum[r] + bias2 : 0.f;
    if (q_sum_out != nullptr) {
      float qs = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) qs += (valid && row0 + kk * 4 + r < B) ? sum[r] + bias2 : 0.f;
      qs += __shfl_xor(qs, 16);
      qs += __shfl_xor(qs, 32);
      if (lane == 0) *q_sum_out = qs;
    }
    }
  }
  if (dact) {
    if (wave < dnt) {
      Tp tp2 = tp;
      tp2.stage = tp.stage + 1;
      f32x4 part = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) part += ld4(scr + (wave * 4 + q) * 256 + lane * 4);
      const int cc = 16 * (dt0 + wave) + i - dact_col0;
      const bool valid = cc >= 0 && cc < dact_cols;
      if constexpr (NM == 8) {
        // this tile's share of the (at most 8) wanted columns: lanes [i_first, i_first + ncols)
        const int t0 = 16 * (dt0 + wave);
        const int i_first = dact_col0 > t0 ? dact_col0 - t0 : 0;
        const int cc_first = t0 + i_first - dact_col0;
        const int ncols = min(16 - i_first, dact_cols - cc_first);
        tp4_allreduce_narrow<NM>(part, i_first, ncols, cc_first, scr + (size_t)wave * 4 * 256, 0.f, 0.f, dactS, tp2);
      } else {
      const f32x4 sum = tp4_allreduce_regs<NM>(part, cc, valid, tp2);
      if (valid) {
        float* o = dactS + (kk * 4) * kOutLd + cc;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r * kOutLd] = sum[r];
      }
      }
    }
    tp.stage += 1;
  }
  tp.stage += 1;
  sf();
  __syncthreads();   // outS (and dactS) visible
  sf();
}

// ---------------------------------------------------------------------------------------
// tp4_scalar_fb for TWO row tiles (32 minibatch rows) per cluster — the over-subscribed launches (B >= 512: SAC at
// B = 1024 is four dispatch rounds of 256 workgroups, one per role, and every workgroup streams ~ 390 KB of weight
// fragments before its first MFMA).  The fragments of a pass are requested ONCE and meet both tiles: per stage two
// accumulator sets, two epilogues, one barrier.  Arithmetic per tile is that of tp4_scalar_fb<P, 4> (same fragments,
// same chains, same member order in the exchange): the results are bit-identical to two one-tile workgroups.
// Clusters of four, no input-column gradient, q through the cluster all-reduce (role B of the phase launches).
// LDS: x0s [2][kR][kX0Ld], h1 / h2 / g2 [2][kR][kWL4] each (tile t at + t * kR * ld), outS [2][kR][kOutLd],
// scr [2][kWaves][256].  Tile t exchanges through the area of slice 2 * slice32 + t: tp.xbuf + t * area.
// ---------------------------------------------------------------------------------------
// all-reduce of two 16 x 16 tiles at once (tp4_allreduce_regs's slots and order; both tiles' granules are published
// before the first poll: one hop, not two)
__device__ __forceinline__ void tp4_allreduce_regs_x2(const f32x4 (&mine)[2], int col, bool valid, const Tp& tp, size_t area,
                                                      f32x4 (&sum)[2]) {
  constexpr int NM = 4;
  const int kk = (threadIdx.x & 63) >> 4;
  constexpr int kRs = 4 * kNarrowMax;
  const unsigned tag = (tp.tag << 6) | (unsigned)(tp.stage & 63);
  sum[0] = mine[0]; sum[1] = mine[1];
  if (!valid) return;
  unsigned long long* slot[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    slot[t] = tp.xbuf + (size_t)t * area + (size_t)tp.stage * NM * kTpBlk + kk * kNarrowMax + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine[t][r]);
      if (tp.local) __hip_atomic_store(slot[t] + (size_t)tp.c * kTpBlk + r * kRs, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_store(slot[t] + (size_t)tp.c * kTpBlk + r * kRs, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // (both tiles' granules are out; the polls go tile by tile — the second tile's peers' granules have usually landed by the
  // time the first tile's have: one hop, and 32 registers of granules instead of 64)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bool ok = false;
    unsigned long long x[NM][4];
    for (int spin = 0; spin < tp.spin && !ok; ++spin) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          x[m][r] = (m == tp.c) ? ((unsigned long long)tag << 32)
                                : __hip_atomic_load(slot[t] + (size_t)m * kTpBlk + r * kRs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = true;
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) ok = ok && (unsigned)(x[m][r] >> 32) == tag;
      if (!ok) __builtin_amdgcn_s_sleep(1);
    }
    if (ok) {
      sum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[t][r] += (m == tp.c) ? mine[t][r] : __uint_as_float((unsigned)x[m][r]);
    } else {
      report_expired(tp.err, tp.err_code | SITE_CLUSTER);
      const float nan = __builtin_nanf("");
      sum[t] = f32x4{nan, nan, nan, nan};
    }
  }
}

template <class P = PrecF32, class ST = NoStamp, class PRE = NoStamp>
__device__ __forceinline__ void tp4_scalar_fb2(const Net& net, const float* x0s, float* h1, float* h2, float* g2, float* outS,
                                               float* scr, Tp& tp, size_t area, const Tp3Store& st, int row0, int B, float seed,
                                               ST sf = ST(), PRE pre = PRE()) {
  constexpr int NM = 4, RT = 2;
  constexpr int TX = kR * kX0Ld, TH = kR * kWL4, TO = kR * kOutLd, TS = kWaves * 256;      // floats between the two tiles' buffers
  const float* const nb0 = net.b[0];
  const float* const nb1 = net.b[1];
  const float* const nb2 = net.b[2];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  using NS = Tp4Steps<P>;
  using SH = Tp4Shape<P, NM>;
  asm volatile("" :: "s"(net.pf[0]), "s"(net.pf[1]), "s"(net.pf[2]), "s"(net.pb[1]), "s"(net.pb[2]), "s"(nb0), "s"(nb1), "s"(nb2),
               "s"(net.dims[0]));
  const int c = tp.c, c0 = c * SH::COLS;
  const int NS0 = (net.dims[0] + P::KS - 1) / P::KS;
  constexpr int kOutWave = 12;
  constexpr int NQ = SH::NQ;
  const int t1 = wave % SH::TPM, kq = wave / SH::TPM;
  const int rt = (int)threadIdx.x >> 8, rl = ((int)threadIdx.x & 255) >> 2, rr = (int)threadIdx.x & 3;

  // ---- requests: once for both row tiles
  using F = typename P::Frag;
  constexpr int BK = P::kBlk;
  constexpr float kO = P::kOut / P::kFwdA;
  float sb = 1.f;
  if constexpr (P::kX2) sb = 4.f * P::a_scale(fabsf(seed));
  const float ob = P::kOut / sb;
  F w0[NS::S0], w1[NQ], w2[SH::M], wz[SH::M];
  {
    const float* p0 = net.pf[0] + (size_t)wave * NS0 * BK + lane * 4;
    P::template ldfn<NS::S0>(w0, p0, NS0);
  }
  const float bias0 = P::ldb(nb0 + 16 * wave + i);
  {
    const float* p1 = net.pf[1] + ((size_t)(c * SH::TPM + t1) * NS::W + kq * NQ) * BK + lane * 4;
    P::template ldfn<NQ>(w1, p1);
  }
  const float bias1 = P::ldb(nb1 + c0 + 16 * rt + (rl & 15));
  const float w3 = P::first(net.pb[2] + (size_t)(c * SH::TPM + rt) * BK + (rl & 15) * 4);
  float bias2 = 0.f;
#pragma unroll
  for (int s = 0; s < SH::M; ++s) w2[s] = P::zf();
  if (wave == kOutWave) {
    const float* p2 = net.pf[2] + ((size_t)c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(w2, p2);
    if (i == 0) bias2 = P::ldb(nb2);
  }
  __builtin_amdgcn_sched_barrier(0);
  pre();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();   // x0 (both tiles) visible

  // ---- L0
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr = x0s + t * TX + i * kX0Ld + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::S0; ++s)
      if (s < NS0) P::mac_s(xr, s, w0[s], acc, P::kFwdA);
    float* o = h1 + t * TH + (kk * 4) * kWL4 + 16 * wave + i;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre_ = acc[r] * kO + bias0;
      ok = ok && P::range_ok(pre_);
      o[r * kWL4] = fmaxf(pre_, 0.f);
    }
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
  }
  {
    const float* q1 = net.pb[1] + ((size_t)wave * NS::W + c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(wz, q1);
  }
  sf();
  __syncthreads();   // h1 visible

  // ---- L1 partials -> scr (per tile); member 0 stores h1
#pragma unroll
  for (int t = 0; t < RT; ++t)
    *reinterpret_cast<f32x4*>(scr + t * TS + ((size_t)wave * 64 + lane) * 4) =
        tp4_mac_steps<P, NQ>(h1 + t * TH + i * kWL4 + SH::KW * kq + 4 * kk, w1);
  if (st.X1 != nullptr && c == 0) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      const int row = (int)threadIdx.x >> 6, col = ((int)threadIdx.x & 63) * 4, gr = row0 + 16 * t + row;
      if (gr < B) tp4_st4(st.X1, (size_t)gr * kW4 + col, ld4(h1 + t * TH + row * kWL4 + col), st.wt);
    }
  }
  sf();
  __syncthreads();   // partial tiles visible
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const float* sp = scr + t * TS + ((size_t)rt * 64 + rl) * 4 + rr;
    float sum = sp[0];
#pragma unroll
    for (int q = 1; q < SH::KP; ++q) sum += sp[q * SH::TPM * 256];
    const float pre_ = sum * kO + bias1;
    if (__builtin_expect(!P::range_ok(pre_), 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    const float v = fmaxf(pre_, 0.f);
    const int off = t * TH + (4 * (rl >> 4) + rr) * kWL4 + c0 + 16 * rt + (rl & 15);
    h2[off] = v;
    g2[off] = v > 0.f ? seed * w3 : 0.f;
  }
  sf();
  __syncthreads();   // h2, g2 (the member's columns) visible

  // ---- dz1 partial (unit seed), mask in place over h1; the h2 / g2 column stores
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const f32x4 a = tp4_mac_steps<P, SH::M>(g2 + t * TH + i * kWL4 + c0 + 4 * kk, wz, sb) * ob;
    float* p = h1 + t * TH + (kk * 4) * kWL4 + 16 * wave + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r * kWL4] = p[r * kWL4] > 0.f ? a[r] : 0.f;
  }
  if ((wave >= 4 && wave < 4 + SH::TPM) || (wave >= 8 && wave < 8 + SH::TPM)) {
    float* dstg = wave < 8 ? st.X2 : st.dY1;
    if (dstg != nullptr) {
      constexpr int C4 = SH::COLS / 4;
      const float* src = wave < 8 ? h2 : g2;
      const int idx = (int)threadIdx.x - (wave < 8 ? 256 : 512);
      const int row = idx / C4, col = c0 + (idx - row * C4) * 4;
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const int gr = row0 + 16 * t + row;
        if (gr < B) tp4_st4(dstg, (size_t)gr * kW4 + col, ld4(src + t * TH + row * kWL4 + col), st.wt);
      }
    }
  }
  sf();
  __syncthreads();   // dz1 partial visible

#pragma unroll
  for (int t = 0; t < RT; ++t) tp4_store_dz1(st, c, h1 + t * TH, row0 + 16 * t, B);
  if (wave == kOutWave) {
    f32x4 qpart[2], qsum[2];
#pragma unroll
    for (int t = 0; t < RT; ++t) qpart[t] = tp4_mac_steps<P, SH::M>(h2 + t * TH + i * kWL4 + c0 + 4 * kk, w2) * kO;
    const bool valid = i == 0;
    tp4_allreduce_regs_x2(qpart, i, valid, tp, area, qsum);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float* o = outS + t * TO + (kk * 4) * kOutLd + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * kOutLd] = valid ? qsum[t][r] + bias2 : 0.f;
    }
  }
  tp.stage += 1;
  sf();
  __syncthreads();   // outS visible
  sf();
}

// ---------------------------------------------------------------------------------------
// tp4_forward on TWO row tiles with DIFFERENT inputs through the same net (clusters of four): SAC's online actor on s'
// (the target chain's a') and on s (the actor step's pi(s)) in ONE pass of role A of the over-subscribed phase-1
// launches (r06-13) — role C, a dispatch round of 256 workgroups of its own behind role A (10 us at humanoid
// B = 1024), disappears; its pass costs the fragments' second use here.  Per tile the arithmetic is tp4_forward<P, 4>'s
// (same fragments, same chains, same member order in the exchange): bit-identical to two one-tile workgroups.
// Tile t: input T.x0[t], buffers T.h1[t] / T.h2[t] / T.out[t] / T.scr[t] ([kWaves][256] floats), stores T.st[t]; tile 0
// exchanges through tp.xbuf, tile 1 through tp.xbuf + area (the cluster area of the role it replaces).
struct Tp4Two {
  const float* x0[2];
  float* h1[2];
  float* h2[2];
  float* out[2];
  float* scr[2];
  Tp3Store st[2];
};
template <class P = PrecF32, class ST = NoStamp, class PRE = NoStamp>
__device__ __forceinline__ void tp4_forward2(const Net& net, const Tp4Two& T, Tp& tp, size_t area, int row0, int B,
                                             ST sf = ST(), PRE pre = PRE()) {
  constexpr int NM = 4, RT = 2;
  using NS = Tp4Steps<P>;
  using SH = Tp4Shape<P, NM>;
  const float* const nb0 = net.b[0];
  const float* const nb1 = net.b[1];
  const float* const nb2 = net.b[2];
  asm volatile("" :: "s"(net.pf[0]), "s"(net.pf[1]), "s"(net.pf[2]), "s"(nb0), "s"(nb1), "s"(nb2),
               "s"(net.dims[0]), "s"(net.dims[3]));
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kk = lane >> 4;
  const int c = tp.c, c0 = c * SH::COLS;
  const int N = net.dims[3];
  const int NS0 = (net.dims[0] + P::KS - 1) / P::KS, NTo = (N + 15) >> 4;
  const int t2 = wave - 4;
  const bool l2_wave = t2 >= 0 && t2 < NTo;
  constexpr int NQ = SH::NQ;
  const int t1 = wave % SH::TPM, kq = wave / SH::TPM;
  const int rt = (int)threadIdx.x >> 8, rl = ((int)threadIdx.x & 255) >> 2, rr = (int)threadIdx.x & 3;

  // ---- requests: once for both row tiles
  using F = typename P::Frag;
  constexpr int BK = P::kBlk;
  constexpr float kO = P::kOut / P::kFwdA;
  F w0[NS::S0], w1[NQ], w2[SH::M];
  {
    const float* p0 = net.pf[0] + (size_t)wave * NS0 * BK + lane * 4;
    P::template ldfn<NS::S0>(w0, p0, NS0);
  }
  const float bias0 = P::ldb(nb0 + 16 * wave + i);
  {
    const float* p1 = net.pf[1] + ((size_t)(c * SH::TPM + t1) * NS::W + kq * NQ) * BK + lane * 4;
    P::template ldfn<NQ>(w1, p1);
  }
  const float bias1 = P::ldb(nb1 + c0 + 16 * rt + (rl & 15));
  float bias2 = 0.f;
#pragma unroll
  for (int s = 0; s < SH::M; ++s) w2[s] = P::zf();
  if (l2_wave) {
    const float* p2 = net.pf[2] + ((size_t)t2 * NS::W + c * SH::M) * BK + lane * 4;
    P::template ldfn<SH::M>(w2, p2);
    if (16 * t2 + i < N) bias2 = P::ldb(nb2 + 16 * t2 + i);
  }
  __builtin_amdgcn_sched_barrier(0);
  pre();
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();   // x0 (both tiles) visible

  // ---- L0
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xr = T.x0[t] + i * kX0Ld + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS::S0; ++s)
      if (s < NS0) P::mac_s(xr, s, w0[s], acc, P::kFwdA);
    float* o = T.h1[t] + (kk * 4) * kWL4 + 16 * wave + i;
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pre_ = acc[r] * kO + bias0;
      ok = ok && P::range_ok(pre_);
      o[r * kWL4] = fmaxf(pre_, 0.f);
    }
    if (__builtin_expect(!ok, 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
  }
  sf();
  __syncthreads();   // h1 visible

  // ---- L1 partials -> scr (per tile); member 0 stores h1 of the tiles that keep it
#pragma unroll
  for (int t = 0; t < RT; ++t)
    *reinterpret_cast<f32x4*>(T.scr[t] + ((size_t)wave * 64 + lane) * 4) =
        tp4_mac_steps<P, NQ>(T.h1[t] + i * kWL4 + SH::KW * kq + 4 * kk, w1);
#pragma unroll
  for (int t = 0; t < RT; ++t)
    if (T.st[t].X1 != nullptr && c == 0) {
      const int row = (int)threadIdx.x >> 6, col = ((int)threadIdx.x & 63) * 4, gr = row0 + row;
      if (gr < B) tp4_st4(T.st[t].X1, (size_t)gr * kW4 + col, ld4(T.h1[t] + row * kWL4 + col), T.st[t].wt);
    }
  sf();
  __syncthreads();   // partial tiles visible
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const float* sp = T.scr[t] + ((size_t)rt * 64 + rl) * 4 + rr;
    float v = sp[0];
#pragma unroll
    for (int q = 1; q < SH::KP; ++q) v += sp[q * SH::TPM * 256];
    const float pre_ = v * kO + bias1;
    if (__builtin_expect(!P::range_ok(pre_), 0)) report_expired(tp.err, tp.err_code | SITE_X2_RANGE);
    T.h2[t][(4 * (rl >> 4) + rr) * kWL4 + c0 + 16 * rt + (rl & 15)] = fmaxf(pre_, 0.f);
  }
  __syncthreads();   // the member's h2 columns visible

  // ---- L2 partials + both tiles' all-reduce from registers on waves 4..; waves 8.. store the h2 columns
  if (l2_wave) {
    f32x4 part[2], sum[2];
#pragma unroll
    for (int t = 0; t < RT; ++t) part[t] = tp4_mac_steps<P, SH::M>(T.h2[t] + i * kWL4 + c0 + 4 * kk, w2) * kO;
    const int col = 16 * t2 + i;
    const bool valid = col < N;
    tp4_allreduce_regs_x2(part, col, valid, tp, area, sum);
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float* o = T.out[t] + (kk * 4) * kOutLd + col;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r * kOutLd] = valid ? sum[t][r] + bias2 : 0.f;
    }
  } else if (wave >= 8 && wave < 8 + SH::TPM) {
    constexpr int C4 = SH::COLS / 4;
    const int idx = (int)threadIdx.x - 512;
    const int row = idx / C4, col = c0 + (idx - row * C4) * 4, gr = row0 + row;
#pragma unroll
    for (int t = 0; t < RT; ++t)
      if (T.st[t].X2 != nullptr && gr < B) tp4_st4(T.st[t].X2, (size_t)gr * kW4 + col, ld4(T.h2[t] + row * kWL4 + col), T.st[t].wt);
  }
  tp.stage += 1;
  sf();
  __syncthreads();   // out (both tiles) visible
  sf();
}

}  // namespace oprl
