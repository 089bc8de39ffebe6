// engine.h — workgroup-level MLP building blocks for gfx950 (MI355X).
//
// Decomposition.  A workgroup (1024 threads = 16 wave64, four per SIMD) owns a
// SLICE of kR = 16 minibatch rows and carries it through whole MLPs: the slice's
// activations live in LDS, weights stream from L2/HBM straight into MFMA
// B-operand registers.  Arithmetic is exact fp32 on the matrix cores,
// v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain) — the mode that meets the 1e-4
// Q-value gate.
//
// Weight layout.  One CU streaming a weight matrix is bound by the shape of its
// loads, not by HBM (tools/ubench_stream.hip on MI355X, 256 KB rewritten by a
// previous kernel, one workgroup per CU; profiles/r01a_ubench_stream.txt):
//     16 rows x 64 B per wave-instruction (MFMA fragment of a row-major [out,in]
//     matrix):   13.8 B/clk/CU with 4 waves, 25 with 16 — and no faster on re-read
//     1 KB contiguous per wave-instruction:  20.5 with 4 waves, 48.8 with 16
// So every Linear layer is kept, besides the torch-visible row-major master, as
// FRAGMENT-ORDER PACKS in which the 64 lanes' b128 operands of one macro step
// are one contiguous KB:
//     pack[((tile*NS + s)*64 + lane)*4 + t] = M[16*tile + (lane&15)][16*s + 4*(lane>>4) + t]
// (zero beyond the matrix), with M = W for the forward (tile over out-features,
// contraction over in-features) and M = W^T for the backward dX = dY·W.  MFMA
// step t of macro step s contracts index 16s + 4(lane>>4) + t, so the A operand
// of four steps is ONE ds_read_b128 of the row-major LDS activation tile.  With
// packs the forward, the first layer (K not a multiple of 4), narrow outputs and
// both backward forms are the same routine; padding is free.  The packs are
// written by the dW+Adam kernel in its epilogue (and by k_repack when the master
// was changed from outside).
//
// Replaces (reference, torch ATen): addmm/mm/threshold_backward sequences of
// algos/nn_models.py:84-107 under autograd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oprl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kR = 16;          // minibatch rows per workgroup slice
constexpr int kThreads = 1024;  // 16 waves
constexpr int kWaves = 16;
constexpr int kMaxLayers = 4;
constexpr int kNarrowMax = 48;  // widest "narrow" output (humanoid 2A = 42)
constexpr int kRing = 8;        // macro steps of B fragments a wave keeps in flight (8 KB; 16 spills at 128 VGPRs)

__host__ __device__ constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ constexpr int cdiv(int x, int m) { return (x + m - 1) / m; }
// LDS leading dimension for a [kR][K] fp32 tile: K rounded to the 16-deep macro
// step plus 8 floats, i.e. ld ≡ 8 (mod 64) for K ≡ 0 (mod 64): the 16-lane
// groups of ds_read_b128 then touch all 64 banks exactly once (conflict-free).
__host__ __device__ constexpr int lds_ld(int k) { return round_up(k, 16) + 8; }
// floats in one fragment-order pack of an [rows x cols] matrix (tile over rows)
__host__ __device__ constexpr long pack_floats(int rows, int cols) {
  return (long)cdiv(rows, 16) * cdiv(cols, 16) * 256;
}
// position of M[r][c] inside its pack (NS = cdiv(cols,16))
__host__ __device__ inline long pack_index(int r, int c, int NS) {
  const int tile = r >> 4, i = r & 15, s = c >> 4, kk = (c & 15) >> 2, t = c & 3;
  return (((long)tile * NS + s) * 64 + (kk * 16 + i)) * 4 + t;
}

struct Net {  // device view of one MLP (by value in kernel args)
  int n_layers;
  int dims[kMaxLayers + 1];
  const float* b[kMaxLayers];   // biases (row-major master arena)
  const float* pf[kMaxLayers];  // forward packs  (M = W,   tiles over out-features)
  const float* pb[kMaxLayers];  // backward packs (M = W^T, tiles over in-features)
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// COHERENT loads: what another workgroup of the SAME launch wrote (behind its flag).  A plain load may hit this compute
// unit's L1 — a line an earlier workgroup on this unit, or this one, loaded before the writer wrote — and `buffer_inv sc0`
// does NOT drop such lines here (workgroup-scope invalidate: a no-op outside threadgroup-split mode); `buffer_inv sc1` does,
// at 14.7 us (it walks the L2 as well).  Loads with a scope above the workgroup miss the L1: tools/ubench_handoff.hip —
// plain loads behind `buffer_inv sc0`: every read stale; these: 0 of 1.3e10 (r04-23).  ldc: an agent-scope atomic load.
// ld4c: ONE 16-byte load at agent scope (buffer_load_dwordx4 ... sc1 through the raw-buffer builtin: a load the compiler
// tracks).  The buffer's base is the first active lane's pointer (v_readfirstlane), the other lanes' pointers go in as
// unsigned byte offsets from it: the callers' addresses do not decrease with the lane index (base + f(wave) + g(lane), g
// non-decreasing), which every use below satisfies.  (A volatile global load — sc0 sc1, system scope — is as coherent and
// needs no such promise, but made an update 4 us longer.)
__device__ __forceinline__ f32x4 ld4c(const float* p) {
  const unsigned long long pv = (unsigned long long)p;
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)pv), bhi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
  const unsigned long long bv = ((unsigned long long)bhi << 32) | (unsigned long long)blo;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bv), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(pv - bv), 0, 16 /* sc1 */));
}
__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---------------------------------------------------------------------------
// Precision policies of the lean passes (tp4.h), the layer-wise kernels and the dW GEMMs.
//
// A MACRO STEP is one 16-byte B fragment per lane (1 KB per wave, contiguous in the pack):
//   PrecF32   16 contraction indices: four v_mfma_f32_16x16x4_f32 (exact fp32, the parity mode)
//   PrecBF16  32 contraction indices: ONE v_mfma_f32_16x16x32_bf16 (fp32 accumulate), 16x the
//             matrix rate and half the weight bytes.  Its fragment is two consecutive fp32 macro
//             steps side by side, rounded to bf16 (RNE, v_cvt_pk_bf16_f32): lane (kk, i) holds
//                 M[16 tile + i][32 s + 4 kk + t]        t = 0..3   (fp32 step 2s)
//                 M[16 tile + i][32 s + 16 + 4 kk + t]   t = 0..3   (fp32 step 2s + 1)
//             so the A operand is the SAME two conflict-free ds_read_b128 of the fp32 LDS
//             activation tile the fp32 mode issues for steps 2s and 2s + 1, converted on the way
//             in (activations stay fp32 in LDS and HBM; master weights, Adam and Polyak stay fp32;
//             the bf16 packs are derived state written by the dW + Adam epilogue).  The MFMA pairs
//             element t of A's lane group with element t of B's: any k-order inside a step is
//             as good as any other as long as both operands use it.
// xr in mac(): X + (lane & 15) * ld + 4 * (lane >> 4), X a row-major fp32 LDS tile.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), returned in every lane of the row: four
// data-parallel-primitive adds (quad xor 1, quad xor 2, half mirror, row mirror) — VALU operand modifiers,
// a few cycles each — where __shfl_xor costs a ds_bpermute round trip (~100 cycles) per step.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__device__ __forceinline__ bf16x8 cvt_bf16x8(const f32x4 lo, const f32x4 hi) {
  const f32x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_convertvector(v, bf16x8);
}

// A 16-byte load that goes past this XCD's L2 (sc1: what a workgroup on another XCD has just written through) as a
// load the COMPILER sees: a raw buffer load with the cache policy in its aux operand.  (The same instruction as inline
// asm is invisible to hipcc's wait-count insertion and to its register allocator — result registers copied or reused
// before the explicit s_waitcnt: r03-37, tools/check_asm_loads.py.)
typedef unsigned u32x4_ld __attribute__((ext_vector_type(4)));
// `base`: the same in every lane (the buffer resource lives in scalar registers), `float_off` < 2^29: this lane's element.
__device__ __forceinline__ f32x4 ld4_agent(const float* base, unsigned float_off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4u, 0, 16 /* sc1 */));
}

// 16- / 8-byte stores WRITTEN THROUGH (agent scope: sc1) as stores the COMPILER sees — 64-bit atomic stores.  (The same
// thing as `global_store_dwordx4 ... sc1` in inline asm is invisible to hipcc's hazard recogniser: a vector-memory store of
// more than 64 bits reads its data registers late, and a following VALU write of those registers needs a wait state the
// compiler only inserts for stores it knows — the first dword of a 16-byte store came out wrong now and then, r04-18.)
__device__ __forceinline__ void st16_agent(void* p, const f32x4 v) {
  typedef unsigned long long u64x2_st __attribute__((ext_vector_type(2)));
  const u64x2_st q = __builtin_bit_cast(u64x2_st, v);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and as ONE 16-byte instruction where the destination is `base` (wave-uniform at run time: it goes through
// v_readfirstlane) + this lane's offset: a raw buffer store with the cache policy in its aux operand, the mirror of ld4c.
// With coherent loads on the reader's side every store form hands over correctly (tools/ubench_handoff.hip); this one is
// the fastest (1.96 us per hand-over against 2.37 for two 8-byte stores) and WRITE_SIZE counts it as 27 % less traffic.
__device__ __forceinline__ void st16_agent_at(float* base, unsigned float_off, const f32x4 v) {
  typedef unsigned u32x4_s __attribute__((ext_vector_type(4)));
  const unsigned long long bv = (unsigned long long)base;
  const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bv), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bv >> 32));
  base = reinterpret_cast<float*>(((unsigned long long)bhi << 32) | (unsigned long long)blo);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_s, v), r, float_off * 4u, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void st8_agent(void* p, unsigned long long v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct PrecF32 {
  static constexpr int KS = 16;
  __device__ static __forceinline__ bool range_ok(float) { return true; }   // (PrecX2's forward range check: nothing to check here)
  static constexpr bool kBf16 = false;
  static constexpr bool kX2 = false;
  // the policy-independent part of the interface (PrecX2 below is where it matters): a macro step's B fragment, the
  // floats between two (tile, step) blocks of a pack, the factor every accumulator is multiplied by
  typedef f32x4 Frag;
  static constexpr int kBlk = 256;
  static constexpr float kOut = 1.f;
  static constexpr float kFwdA = 1.f;
  __device__ static __forceinline__ Frag ldf(const float* p) { return ld4(p); }
  __device__ static __forceinline__ float ldb(const float* p) { return *p; }     // a bias element
  // N consecutive (tile, step) blocks of a pack from p on: w[s] = s < n ? ldf(p + s kBlk) : zf()
  template <int N> __device__ static __forceinline__ void ldfn(Frag (&w)[N], const float* p, int n = N) {
#pragma unroll
    for (int s = 0; s < N; ++s) w[s] = s < n ? ldf(p + s * kBlk) : zf();
  }
  __device__ static __forceinline__ Frag zf() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ static __forceinline__ void mac_s(const float* xr, int s, const Frag b, f32x4& acc, float) { mac(xr, s, b, acc); }
  __device__ static __forceinline__ void mac_tail_s(const float* xr, int s, const Frag b, f32x4& acc, int k16, float) { mac_tail(xr, s, b, acc, k16); }
  __device__ static __forceinline__ float a_scale(float) { return 1.f; }     // scale of the A operand for a tile whose largest |element| is given
  __device__ static __forceinline__ void mac(const float* xr, int s, const f32x4 b, f32x4& acc) {
    const f32x4 a = ld4(xr + 16 * s);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = mfma4(a[t], b[t], acc);
  }
  // k16 = valid fp32-size steps of the contraction: nothing to guard here
  __device__ static __forceinline__ void mac_tail(const float* xr, int s, const f32x4 b, f32x4& acc, int) {
    mac(xr, s, b, acc);
  }
  // the first matrix element of a fragment
  __device__ static __forceinline__ float first(const float* frag) { return frag[0]; }
};

struct PrecBF16 {
  static constexpr int KS = 32;
  __device__ static __forceinline__ bool range_ok(float) { return true; }   // (PrecX2's forward range check: nothing to check here)
  static constexpr bool kBf16 = true;
  static constexpr bool kX2 = false;
  typedef f32x4 Frag;
  static constexpr int kBlk = 256;
  static constexpr float kOut = 1.f;
  static constexpr float kFwdA = 1.f;
  __device__ static __forceinline__ Frag ldf(const float* p) { return ld4(p); }
  __device__ static __forceinline__ float ldb(const float* p) { return *p; }     // a bias element
  // N consecutive (tile, step) blocks of a pack from p on: w[s] = s < n ? ldf(p + s kBlk) : zf()
  template <int N> __device__ static __forceinline__ void ldfn(Frag (&w)[N], const float* p, int n = N) {
#pragma unroll
    for (int s = 0; s < N; ++s) w[s] = s < n ? ldf(p + s * kBlk) : zf();
  }
  __device__ static __forceinline__ Frag zf() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ static __forceinline__ void mac_s(const float* xr, int s, const Frag b, f32x4& acc, float) { mac(xr, s, b, acc); }
  __device__ static __forceinline__ void mac_tail_s(const float* xr, int s, const Frag b, f32x4& acc, int k16, float) { mac_tail(xr, s, b, acc, k16); }
  __device__ static __forceinline__ float a_scale(float) { return 1.f; }
  __device__ static __forceinline__ void mac(const float* xr, int s, const f32x4 b, f32x4& acc) {
    const bf16x8 a = cvt_bf16x8(ld4(xr + 32 * s), ld4(xr + 32 * s + 16));
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
  // the upper half of the last step may lie beyond the LDS tile's row (narrow [kR][kOutLd] tiles):
  // it is read only when fp32-size step 2s + 1 exists
  __device__ static __forceinline__ void mac_tail(const float* xr, int s, const f32x4 b, f32x4& acc, int k16) {
    const f32x4 lo = ld4(xr + 32 * s);
    const f32x4 hi = (2 * s + 1 < k16) ? ld4(xr + 32 * s + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cvt_bf16x8(lo, hi), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ float first(const float* frag) {
    return __uint_as_float((*reinterpret_cast<const unsigned*>(frag) & 0xffffu) << 16);
  }
};

// ---------------------------------------------------------------------------
// PrecX2 — fp32-class arithmetic at the fp16 matrix rate ("split" mode, the parity mode of round 3).
// Every operand is the sum of two fp16 numbers, x = hi + lo with hi = fp16(x) and lo = fp16(x - hi) (round to nearest:
// 22 significant bits), and a product is three MFMAs v_mfma_f32_16x16x32_f16 with fp32 accumulation,
//     lo_a hi_b  +  hi_a lo_b  +  hi_a hi_b            (the dropped lo lo term is 2^-22 of the product)
// — 48 cycles per 32 contraction indices where the exact-fp32 instruction takes 256, for a per-product error of
// ~2^-21 against fp32's 2^-24 (summation-order noise of the 256-deep contractions is of the same size).
//   B operand: fragment packs with TWO fp16 planes per (tile, macro step) block — [hi: 64 lanes x 8 halfs | lo: the
//     same] = 2 KB, the bytes of the fp32 pack — in the k-order of the bf16 packs (pack16_index), holding 2^8 w (the
//     weights sit around 2^-4: scaled, their lo parts are normal fp16 numbers); written by the dW + Adam epilogues
//     and k_repack from the fp32 master (RNE).
//   A operand: the SAME two ds_read_b128 of the fp32 LDS activation tile as the other modes, split on the way in:
//     4 v_cvt_pk_f16_f32 (hi), 8 v_fma_mix_f32 (x - hi, the fp16 operand converted by the instruction), 4 v_cvt_pk
//     (lo): 16 VALU operations per macro step.  Gradient tiles are scaled by a power of two first (mac_s): their
//     elements sit far below fp16's normal range (2^-14).  a_scale(m) = 2^(10 - exponent of m), m the largest
//     magnitude the tile can hold; below 2^-25 of that an element contributes its hi part only (fixed point).
//   Accumulators come out as 2^8 a_scale times the product: every epilogue multiplies by kOut / a_scale (exact).
// Range: forward activations go in as 2^4 x (kFwdA): |x| < 4094 (PrecX2::kActMax) — beyond that the hi plane is inf, the
// residual -inf and every product NaN, which a ReLU would quietly turn into 0.  So every stage of the lean passes that
// writes activations for a later stage checks what it writes (PrecX2::range_ok: NaN or >= kActMax -> the learner's error
// word, SITE_X2_RANGE: the host raises on the same update); raw observations are caught by the first layer's check
// (their NaN products), weights by the same route (|w| >= 256 overflows the 2^8 w packs).  Gradient tiles scale
// themselves (a_scale) and have no such limit.
// ---------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned x2u4 __attribute__((ext_vector_type(4)));
struct FragX2 { f32x4 hi, lo; };

// x - (float)half of a packed pair, the conversion folded into the FMA (v_fma_mix_f32: operand 0 read as fp16)
__device__ __forceinline__ float x2_res_lo(float x, unsigned hp) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
  return r;
}
__device__ __forceinline__ float x2_res_hi(float x, unsigned hp) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(x));
  return r;
}
__device__ __forceinline__ void x2_split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
  const f32x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  hi = __builtin_convertvector(v, f16x8);
  const x2u4 hp = __builtin_bit_cast(x2u4, hi);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[2 * i] = x2_res_lo(v[2 * i], hp[i]);
    r[2 * i + 1] = x2_res_hi(v[2 * i + 1], hp[i]);
  }
  lo = __builtin_convertvector(r, f16x8);
}

struct PrecX2 {
  static constexpr int KS = 32;
  static constexpr bool kBf16 = false;
  static constexpr bool kX2 = true;
  typedef FragX2 Frag;
  static constexpr int kBlk = 512;
  static constexpr float kWScale = 256.f;            // the packs hold 2^8 w
  static constexpr float kOut = 1.f / 256.f;
  // forward activations are scaled by 2^4 on the way in: full 22 bits for |x| >= 2^-7, an absolute floor of 2^-29
  // below, |x| < 4094
  static constexpr float kFwdA = 16.f;
  __device__ static __forceinline__ Frag ldf(const float* p) { return FragX2{ld4(p), ld4(p + 256)}; }
  __device__ static __forceinline__ float ldb(const float* p) { return *p; }     // a bias element
  // N consecutive (tile, step) blocks of a pack from p on: w[s] = s < n ? ldf(p + s kBlk) : zf()
  template <int N> __device__ static __forceinline__ void ldfn(Frag (&w)[N], const float* p, int n = N) {
#pragma unroll
    for (int s = 0; s < N; ++s) w[s] = s < n ? ldf(p + s * kBlk) : zf();
  }
  __device__ static __forceinline__ Frag zf() { return FragX2{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}; }
  __device__ static __forceinline__ void mma3(const f32x4 x0, const f32x4 x1, const Frag& b, f32x4& acc) {
    f16x8 ah, al;
    x2_split8(x0, x1, ah, al);
    const f16x8 bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.lo);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
  // ... the same with the A operand split ONCE by the caller (an A block that meets several B tiles: the split is 16
  // VALU operations, the three MFMAs are 96 cycles of the matrix pipe)
  __device__ static __forceinline__ void mma3_split(const f16x8 ah, const f16x8 al, const Frag& b, f32x4& acc) {
    const f16x8 bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.lo);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void mac(const float* xr, int s, const Frag& b, f32x4& acc) {
    mma3(ld4(xr + 32 * s), ld4(xr + 32 * s + 16), b, acc);
  }
  __device__ static __forceinline__ void mac_s(const float* xr, int s, const Frag& b, f32x4& acc, float sc) {
    mma3(ld4(xr + 32 * s) * sc, ld4(xr + 32 * s + 16) * sc, b, acc);
  }
  // (the upper half of the last step may lie beyond the LDS tile's row: read only when fp32-size step 2s + 1 exists)
  __device__ static __forceinline__ void mac_tail_s(const float* xr, int s, const Frag& b, f32x4& acc, int k16, float sc) {
    const f32x4 lo = ld4(xr + 32 * s);
    const f32x4 hi = (2 * s + 1 < k16) ? ld4(xr + 32 * s + 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    mma3(lo * sc, hi * sc, b, acc);
  }
  // 2^(10 - e) for m in [2^e, 2^(e+1)); 1 for m = 0 (an all-zero tile).  The scale's exponent field is clamped to 254
  // (2^127): a tile whose largest magnitude lies below 2^-117 would otherwise get inf (or a sign bit) for a scale, and
  // inf times the tile's zeros is NaN
  __device__ static __forceinline__ float a_scale(float m) {
    const unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
    const unsigned f = 127u + 10u + 127u - e;
    return e == 0u ? 1.f : __uint_as_float((f > 254u ? 254u : f) << 23);
  }
  // what a forward stage may hand to the next one: finite and below fp16's range after the 2^4 scale
  static constexpr float kActMax = 4094.f;
  __device__ static __forceinline__ bool range_ok(float pre) { return pre < kActMax; }     // (false for NaN)
  // the first matrix element of a fragment block (pointer at the lane's hi quad)
  __device__ static __forceinline__ float first(const float* frag) {
    const unsigned h = *reinterpret_cast<const unsigned*>(frag) & 0xffffu, l = *reinterpret_cast<const unsigned*>(frag + 256) & 0xffffu;
    return ((float)__builtin_bit_cast(_Float16, (unsigned short)h) + (float)__builtin_bit_cast(_Float16, (unsigned short)l)) * kOut;
  }
};

// Coh<P>: the policy P with COHERENT weight-fragment and bias loads — for the passes of a launch whose weights another
// workgroup of the same launch has just written (k_ddpg_chain: the tiles of the update before / of this update).
template <class P>
struct Coh : P {
  typedef typename P::Frag Frag;
  // one buffer resource per call (two v_readfirstlane), the steps as constant offsets from it
  template <int N> __device__ static __forceinline__ void ldfn(Frag (&w)[N], const float* p, int n = N) {
      const unsigned long long pv = (unsigned long long)p;
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)pv), bhi = __builtin_amdgcn_readfirstlane((unsigned)(pv >> 32));
    const unsigned long long bv = ((unsigned long long)bhi << 32) | (unsigned long long)blo;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(bv), 0, 0x7fffffff, 0x00020000);
    const unsigned vo = (unsigned)(pv - bv);
#pragma unroll
    for (int s = 0; s < N; ++s) {
      if (s < n) {
        if constexpr (P::kX2) {
          w[s].hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (unsigned)(s * P::kBlk * 4), 16));
          w[s].lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (unsigned)(s * P::kBlk * 4 + 1024), 16));
        } else {
          w[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, (unsigned)(s * P::kBlk * 4), 16));
        }
      } else {
        w[s] = P::zf();
      }
    }
  }
  __device__ static __forceinline__ Frag ldf(const float* p) { Frag w[1]; ldfn<1>(w, p); return w[0]; }
  __device__ static __forceinline__ float ldb(const float* p) { return ldc(p); }
  __device__ static __forceinline__ float first(const float* frag) {
    if constexpr (P::kX2) {
      const unsigned h = __hip_atomic_load(reinterpret_cast<const unsigned*>(frag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
      const unsigned l = __hip_atomic_load(reinterpret_cast<const unsigned*>(frag + 256), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
      return ((float)__builtin_bit_cast(_Float16, (unsigned short)h) + (float)__builtin_bit_cast(_Float16, (unsigned short)l)) * P::kOut;
    } else if constexpr (P::kBf16) {
      const unsigned h = __hip_atomic_load(reinterpret_cast<const unsigned*>(frag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return __uint_as_float((h & 0xffffu) << 16);
    } else {
      return ldc(frag);
    }
  }
};


// position (in bf16 elements) of M[r][c] inside its bf16 pack (NS = cdiv(cols, 32))
__host__ __device__ inline long pack16_index(int r, int c, int NS) {
  const int tile = r >> 4, i = r & 15, s = c >> 5, c5 = c & 31;
  const int half = c5 >> 4, kk = (c5 & 15) >> 2, t = (c5 & 3) + 4 * half;
  return (((long)tile * NS + s) * 64 + (kk * 16 + i)) * 8 + t;
}
// 16-byte units ("float4 slots") of one bf16 pack of an [rows x cols] matrix, in floats
__host__ __device__ constexpr long pack16_floats(int rows, int cols) {
  return (long)cdiv(rows, 16) * cdiv(cols, 32) * 256;
}

// Compile-time-indexed select from a small array that lives in kernel-argument
// memory: a runtime subscript would make hipcc copy the whole by-value argument
// block to scratch (observed: 440 B/lane).
template <int N, class T>
__device__ __forceinline__ T pick(const T (&arr)[N], int idx) {
  T v = arr[0];
#pragma unroll
  for (int i = 1; i < N; ++i)
    if (idx == i) v = arr[i];
  return v;
}

// ---------------------------------------------------------------------------
// One wave: acc += X[kR, 16*s0 .. 16*s1) · pack_tile, i.e. a 16x16 output tile
// over macro steps [s0, s1).  `ptile` points at the tile's step-0 block (steps
// are 256 floats apart).  All B fragments (up to kRing steps at a time) are
// issued before the first MFMA; hipcc's scheduler would otherwise sink each load
// next to its use (observed: one miss latency per macro step), hence the
// sched_barriers.
// ---------------------------------------------------------------------------
// `sync_first`: the workgroup barrier that makes X visible is taken AFTER the
// first ring of weight loads has been issued (weights do not depend on X), so
// the miss latency of a layer's first fragments overlaps the previous layer's
// tail instead of following it.  All waves of the workgroup must call with the
// same flag, exactly once per GEMM (callers pass it for a wave's first tile).
__device__ __forceinline__ void tile_mac(const float* __restrict__ Xs, int ldx,
                                         const float* __restrict__ ptile, int s0, int s1,
                                         f32x4& acc, bool sync_first) {
  const int lane = threadIdx.x & 63;
  const float* xrow = Xs + (lane & 15) * ldx + 4 * (lane >> 4);
  const float* pl = ptile + lane * 4;
  // software pipeline: kRing fragments in flight at all times — slot d is
  // re-requested (macro step s + kRing) right after step s's MFMAs are issued.
  // (A first version loaded a ring, computed it, then loaded the next: the
  // second ring's miss was fully exposed and the 256x256 layer took 6.2 us;
  // tools/ubench_gemm.hip: this structure 4.9 us, MFMA+LDS alone 3.75, stream alone 3.1.)
  f32x4 b[kRing];
#pragma unroll
  for (int d = 0; d < kRing; ++d)
    if (s0 + d < s1) b[d] = ld4(pl + (size_t)(s0 + d) * 256);
  __builtin_amdgcn_sched_barrier(0);
  if (sync_first) {
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int sb = s0; sb < s1; sb += kRing) {
#pragma unroll
    for (int d = 0; d < kRing; ++d) {
      const int s = sb + d;
      if (s < s1) {
        const f32x4 a4 = ld4(xrow + 16 * s);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma4(a4[t], b[d][t], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (s + kRing < s1) b[d] = ld4(pl + (size_t)(s + kRing) * 256);   // refill the slot just consumed
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// OUT[kR, 16*NT) = epilogue(X[kR, 16*NS) · pack)   — every GEMM of the slice.
//   NT >= kWaves ("wide", NT a multiple of kWaves): wave w owns tiles w, w+16, ..
//       over the whole contraction; epi(row, col, v) is applied from registers.
//   NT <  kWaves ("narrow", NT <= 8): the waves split the contraction of each
//       tile, partial tiles meet in `scratch` ([kWaves][kR][16] floats), then
//       epi runs once per element.  Contains the barriers it needs for that.
// X must be zero padded to 16*NS columns.  The routine takes the barrier that
// makes X visible itself (after issuing its first weight loads, see tile_mac);
// the caller syncs after (OUT complete) in the wide case, the narrow case ends
// with a barrier.  OUT must not alias X.
// ---------------------------------------------------------------------------
// `bias` (nullable, `nbias` valid entries) is added to the result before epi; it is
// loaded BEFORE the MFMA loop — a load issued in the epilogue sits behind the
// sched_barriers and costs a full exposed round trip per GEMM.
struct NoStamp { __device__ __forceinline__ void operator()() const {} };

template <class Epi, class DStamp = NoStamp>
__device__ __forceinline__ void gemm_packed(const float* __restrict__ Xs, int ldx,
                                            const float* __restrict__ pack, int NT, int NS,
                                            float* __restrict__ scratch,
                                            const float* __restrict__ bias, int nbias, Epi&& epi,
                                            DStamp dstamp = DStamp(), int tile_stride = 0) {
  // tile_stride: floats between two tiles' step-0 blocks; 0 = NS*256 (a whole pack).  A
  // caller contracting over a sub-range of a pack's steps passes the pack's full stride.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const size_t tstride = tile_stride > 0 ? (size_t)tile_stride : (size_t)NS * 256;
  if (NT >= kWaves) {
#pragma unroll 1
    for (int tile = wave; tile < NT; tile += kWaves) {
      const int col = 16 * tile + i;
      const float bv = (bias != nullptr && col < nbias) ? bias[col] : 0.f;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      tile_mac(Xs, ldx, pack + (size_t)tile * tstride, 0, NS, acc, tile == wave);
#pragma unroll
      for (int r = 0; r < 4; ++r) epi(kk * 4 + r, col, acc[r] + bv);
    }
  } else {
    const int wpt = kWaves / NT;            // waves per tile
    const int tile = wave / wpt, part = wave - tile * wpt;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const int per = cdiv(NS, wpt);
    const int s0 = part * per, s1 = min(NS, s0 + per);
    // the reduce phase's (row, col) of this thread, and its bias, known up front
    const int ridx = threadIdx.x;
    const bool rmine = ridx < kR * 16 * NT;
    const int rt = ridx / (kR * 16), rrem = ridx - rt * (kR * 16);
    const int rrow = rrem >> 4, rcol = 16 * rt + (rrem & 15);
    const float rb = (bias != nullptr && rmine && rcol < nbias) ? bias[rcol] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    dstamp();
    if (tile < NT && s0 < s1) {
      tile_mac(Xs, ldx, pack + (size_t)tile * tstride, s0, s1, acc, true);
    } else {
      __syncthreads();   // idle wave: still owes the X-visibility barrier
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[(wave * kR + kk * 4 + r) * 16 + i] = acc[r];
    dstamp();
    __syncthreads();
    dstamp();
    if (rmine) {
      float v = 0.f;
      const float* sp = scratch + ((rt * wpt) * kR + rrow) * 16 + (rrem & 15);
#pragma unroll 4
      for (int p = 0; p < wpt; ++p) v += sp[p * kR * 16];
      epi(rrow, rcol, v + rb);
    }
    // NT > 4 (wide first-layer inputs, e.g. humanoid S+A = 88 -> 6 tiles): remaining elements
    for (int idx = threadIdx.x + kThreads; idx < kR * 16 * NT; idx += kThreads) {
      const int t = idx / (kR * 16), rem = idx - t * (kR * 16);
      const int row = rem >> 4, col = 16 * t + (rem & 15);
      float v = (bias != nullptr && col < nbias) ? bias[col] : 0.f;
      for (int p = 0; p < wpt; ++p) v += scratch[((t * wpt + p) * kR + row) * 16 + (rem & 15)];
      epi(row, col, v);
    }
    dstamp();
    __syncthreads();
    dstamp();
  }
}

// ---------------------------------------------------------------------------
// LDS tile helpers
// ---------------------------------------------------------------------------
// zero an LDS region of n floats (n multiple of 4, 16-byte aligned)
__device__ __forceinline__ void lds_zero(float* p, int n) {
  for (int idx = threadIdx.x * 4; idx < n; idx += kThreads * 4)
    *reinterpret_cast<f32x4*>(p + idx) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// copy rows [row0, row0+kR) of a global [B, k] matrix (leading dim ldg) into
// columns [c0, c0+k) of an LDS tile; rows >= B read as zero.
__device__ __forceinline__ void load_rows(float* __restrict__ Xs, int ldx, int c0,
                                          const float* __restrict__ G, int ldg, int k, int row0,
                                          int B) {
  for (int idx = threadIdx.x; idx < kR * k; idx += kThreads) {
    const int row = idx / k, col = idx - row * k;
    const int gr = row0 + row;
    Xs[row * ldx + c0 + col] = gr < B ? G[(size_t)gr * ldg + col] : 0.f;
  }
}

// ... with coherent loads (rows another workgroup of the same launch staged)
__device__ __forceinline__ void load_rows_c(float* __restrict__ Xs, int ldx, int c0,
                                            const float* __restrict__ G, int ldg, int k, int row0,
                                            int B) {
  for (int idx = threadIdx.x; idx < kR * k; idx += kThreads) {
    const int row = idx / k, col = idx - row * k;
    const int gr = row0 + row;
    Xs[row * ldx + c0 + col] = gr < B ? ldc(G + (size_t)gr * ldg + col) : 0.f;
  }
}

// store columns [0,k) of an LDS tile to rows [row0, ...) of a global matrix
__device__ __forceinline__ void store_rows(const float* __restrict__ Xs, int ldx,
                                           float* __restrict__ G, int ldg, int k, int row0, int B) {
  for (int idx = threadIdx.x; idx < kR * k; idx += kThreads) {
    const int row = idx / k, col = idx - row * k;
    const int gr = row0 + row;
    if (gr < B) G[(size_t)gr * ldg + col] = Xs[row * ldx + col];
  }
}

// ... written through (agent-scope stores): rows a workgroup of the SAME launch reads behind a flag
__device__ __forceinline__ void store_rows_wt(const float* __restrict__ Xs, int ldx,
                                              float* __restrict__ G, int ldg, int k, int row0, int B) {
  for (int idx = threadIdx.x; idx < kR * k; idx += kThreads) {
    const int row = idx / k, col = idx - row * k;
    const int gr = row0 + row;
    if (gr < B) __hip_atomic_store(G + (size_t)gr * ldg + col, Xs[row * ldx + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// float4 variants for WIDTH-wide tiles (k multiple of 4, ldg multiple of 4)
__device__ __forceinline__ void store_rows4(const float* __restrict__ Xs, int ldx,
                                            float* __restrict__ G, int ldg, int k, int row0, int B) {
  const int k4 = k >> 2;
  for (int idx = threadIdx.x; idx < kR * k4; idx += kThreads) {
    const int row = idx / k4, col = (idx - row * k4) * 4;
    const int gr = row0 + row;
    if (gr < B) *reinterpret_cast<f32x4*>(G + (size_t)gr * ldg + col) = ld4(Xs + row * ldx + col);
  }
}

__device__ __forceinline__ void load_rows4(float* __restrict__ Xs, int ldx,
                                           const float* __restrict__ G, int ldg, int k, int row0,
                                           int B) {
  const int k4 = k >> 2;
  for (int idx = threadIdx.x; idx < kR * k4; idx += kThreads) {
    const int row = idx / k4, col = (idx - row * k4) * 4;
    const int gr = row0 + row;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (gr < B) v = ld4(G + (size_t)gr * ldg + col);
    *reinterpret_cast<f32x4*>(Xs + row * ldx + col) = v;
  }
}

// ---------------------------------------------------------------------------
// Whole-MLP forward / backward for one slice.
//
// LDS map (floats), WL = lds_ld(WIDTH), XL = kX0Ld:
//   x0   [kR][XL]          layer-0 input (zero padded)
//   h[l] [kR][WL]  l < n_layers-1   hidden activations (ReLU outputs), later
//                          overwritten in place by the gradient flowing into them
//   out  [kR][kOutLd]      network output
//   aux  [kR][kOutLd]      dLoss/d(out), later the input-column gradient
//   scr  [kWaves][kR][16]  split-contraction partial tiles
// ---------------------------------------------------------------------------
constexpr int kX0Ld = lds_ld(96);   // widest layer-0 input: humanoid S+A = 88
constexpr int kOutLd = kNarrowMax + 8;

template <int WIDTH>
struct SliceLds {
  static constexpr int WL = lds_ld(WIDTH);
  static constexpr int h_off = kR * kX0Ld;
  static constexpr int hbuf = kR * WL;
  __host__ __device__ static constexpr int out_off(int n_h) { return h_off + n_h * hbuf; }
  __host__ __device__ static constexpr int aux_off(int n_h) { return out_off(n_h) + kR * kOutLd; }
  __host__ __device__ static constexpr int scr_off(int n_h) { return aux_off(n_h) + kR * kOutLd; }
  __host__ __device__ static constexpr int total(int n_h) { return scr_off(n_h) + kWaves * kR * 16; }
};

// Forward.  x0 must be loaded (and zero padded) by the caller; NO barrier is
// needed after that (the first GEMM takes it).
// Hidden layer l's output goes to hbase + l*kR*WL.  If store_x the inputs of
// layers 1.. (the hidden activations) are also stored to Xg[l] ([B, WIDTH]).
// Result: outS[kR][kOutLd] columns [0, dims[L]); pad columns up to the next
// multiple of 16 are written as zero.
// (Prefetching the small GEMMs' fragments a stage early was measured not to pay:
// profiles/r01b_experiments.txt #2.)
template <int WIDTH, class Stamp>
__device__ __forceinline__ void mlp_forward_slice(const Net& net, const float* x0s, float* hbase,
                                                  float* outS, float* scr,
                                                  float* const (&Xg)[kMaxLayers], bool store_x,
                                                  int row0, int B, Stamp&& stamp) {
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  constexpr int NTW = WIDTH / 16;
  const int L = net.n_layers;
  const int N = pick(net.dims, L);
  const int NTo = cdiv(N, 16);
  const float* pf_out = pick(net.pf, L - 1);
  {
    const float* bias = net.b[0];
    float* Ys = hbase;
    gemm_packed(x0s, kX0Ld, net.pf[0], NTW, cdiv(net.dims[0], 16), scr, bias, WIDTH,
                [&](int row, int col, float v) { Ys[row * WL + col] = fmaxf(v, 0.f); });
  }
  stamp();
#pragma unroll
  for (int l = 1; l < kMaxLayers - 1; ++l) {
    if (l < L - 1) {
      const float* bias = net.b[l];
      float* Ys = hbase + l * HB;
      gemm_packed(hbase + (l - 1) * HB, WL, net.pf[l], NTW, NTW, scr, bias, WIDTH,
                  [&](int row, int col, float v) { Ys[row * WL + col] = fmaxf(v, 0.f); });
      stamp();
    }
  }
  {
    const float* bias = pick(net.b, L - 1);
    gemm_packed(hbase + (L - 2) * HB, WL, pf_out, NTo, NTW, scr, bias, N,
                [&](int row, int col, float v) { outS[row * kOutLd + col] = col < N ? v : 0.f; });
  }
  // (the narrow GEMM ended with a barrier: every hidden buffer is complete)
  if (store_x) {
#pragma unroll
    for (int l = 1; l < kMaxLayers; ++l)
      if (l < L) store_rows4(hbase + (l - 1) * HB, WL, Xg[l], WIDTH, WIDTH, row0, B);
  }
}

// Backward.  On entry doutS[kR][kOutLd] holds dLoss/d(out) with ZERO padding up
// to round_up(dims[L],16) columns (no barrier needed after writing it) and the
// hidden buffers hold the forward activations.  The gradient wrt hidden layer l's pre-activation output is
// written IN PLACE over hidden buffer l (each element's ReLU mask is read by the
// lane that overwrites it) and, if dYg[l] != nullptr, to dYg[l] ([B,WIDTH]) for
// the dW kernel; the caller stores dY[L-1] = dout itself.  If dact_cols > 0 the
// gradient wrt input columns [dact_col0, +dact_cols) lands in dactS[kR][kOutLd]
// (may alias doutS).
template <int WIDTH, class Stamp>
__device__ __forceinline__ void mlp_backward_slice(const Net& net, const float* doutS,
                                                   float* hbase, float* scr,
                                                   float* const (&dYg)[kMaxLayers], int row0,
                                                   int B, int dact_col0, int dact_cols,
                                                   float* dactS, Stamp&& stamp) {
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  constexpr int NTW = WIDTH / 16;
  const int L = net.n_layers;
  const float* dy = doutS;
  int ldy = kOutLd;
  int ns = cdiv(pick(net.dims, L), 16);
  const int K0 = net.dims[0];
  const int NT0 = cdiv(K0, 16);
#pragma unroll
  for (int l = kMaxLayers - 1; l >= 1; --l) {
    if (l <= L - 1) {
      float* dx = hbase + (l - 1) * HB;   // holds H (mask) now, dX afterwards
      gemm_packed(dy, ldy, net.pb[l], NTW, ns, scr, nullptr, 0, [&](int row, int col, float v) {
        float* p = dx + row * WL + col;
        *p = *p > 0.f ? v : 0.f;
      });
      if (dYg[l - 1] != nullptr) {
        __syncthreads();
        store_rows4(dx, WL, dYg[l - 1], WIDTH, WIDTH, row0, B);
      }
      stamp();
      dy = dx;
      ldy = WL;
      ns = NTW;
    }
  }
  if (dact_cols > 0) {
    // gradient wrt the layer-0 input over all (padded) input columns; keep the range
    gemm_packed(dy, WL, net.pb[0], NT0, NTW, scr, nullptr, 0, [&](int row, int col, float v) {
      const int c = col - dact_col0;
      if (c >= 0 && c < dact_cols) dactS[row * kOutLd + c] = v;
    });
  }
}

}  // namespace oprl
