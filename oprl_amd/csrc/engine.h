// engine.h — workgroup-level MLP building blocks for gfx950 (MI355X).
//
// Decomposition: a workgroup (256 threads = 4 wave64, one per SIMD) owns a
// SLICE of kR = 16 minibatch rows and carries it through whole MLPs; the
// activations of the slice live in LDS, weights stream from L2/HBM straight
// into MFMA B-operand registers (they are used once per workgroup, so an LDS
// round trip would be pure overhead).  Arithmetic is exact fp32 on the matrix
// cores: v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain), the mode that meets
// the 1e-4 Q-value gate.
//
// k-permutation trick: MFMA step t of a 16-deep macro step contracts index
// k0 + 4*(lane>>4) + t, so the A fragment of four steps is ONE ds_read_b128 of
// the row-major LDS tile and the B fragment is ONE global_load_dwordx4 of the
// row-major [out,in] torch weight — no transposed or packed weight copies.
// The backward (dX = dY·W) uses the same trick on the output index, which
// makes its weight reads 256 contiguous bytes per row.
//
// Replaces (reference, torch ATen): addmm/mm/threshold_backward sequences of
// algos/nn_models.py:84-107 under autograd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oprl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kR = 16;         // minibatch rows per workgroup slice
constexpr int kThreads = 256;  // 4 waves
constexpr int kWaves = 4;
constexpr int kMaxLayers = 4;
constexpr int kNarrowMax = 48;  // widest "narrow" output (humanoid 2A = 42)

__host__ __device__ constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
// LDS leading dimension for a [kR][K] fp32 tile: K rounded to the 16-deep macro
// step plus 8 floats, i.e. ld ≡ 8 (mod 64) for K ≡ 0 (mod 64): the 16-lane
// groups of ds_read_b128 then touch all 64 banks exactly once (conflict-free).
__host__ __device__ constexpr int lds_ld(int k) { return round_up(k, 16) + 8; }

struct Net {  // device view of one MLP (by value in kernel args)
  int n_layers;
  int dims[kMaxLayers + 1];
  const float* W[kMaxLayers];
  const float* b[kMaxLayers];
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---------------------------------------------------------------------------
// Memory-level parallelism.  A slice workgroup streams each weight matrix once,
// straight from L2/HBM into MFMA operands, and the weights were rewritten by the
// previous step's Adam kernel, so most of these loads miss the XCD's L2.  With
// one wave per SIMD nothing else hides that latency: every GEMM below therefore
// keeps a deep register ring of B fragments in flight (we have 512 VGPRs per
// lane at this occupancy) or, for the short loops, issues every load up front.
// Measured before this change: 19.7 us per slice launch against a ~4 us MFMA
// floor (profiles/r01_*).
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// First layer: Y[kR, WIDTH] = relu(X[kR, K] · W[WIDTH, K]^T + b), K <= 96 and
// not necessarily a multiple of 4 (rows of W are then not 16-byte aligned), so
// weights come in as predicated scalar loads; X is zero padded to 16 columns.
// Wave w owns output columns [w*WIDTH/4, (w+1)*WIDTH/4).
// ---------------------------------------------------------------------------
template <int WIDTH, int VEC>
__device__ __forceinline__ void gemm_fwd_first_v(const float* __restrict__ Xs, int ldx, int K,
                                                 const float* __restrict__ W,
                                                 const float* __restrict__ bias,
                                                 float* __restrict__ Ys, int ldy) {
  constexpr int TPW = WIDTH / 64;  // 16-wide tiles per wave
  constexpr int CH = 3;            // macro steps preloaded per chunk (K <= 96 -> <= 2 chunks)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int n0w = wave * (WIDTH / 4);
  f32x4 acc[TPW];
  const float* wrow[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    wrow[q] = W + (size_t)(n0w + 16 * q + i) * K;
  }
  const float* xrow = Xs + i * ldx + 4 * kk;
  const int nstep = round_up(K, 16) >> 4;
  for (int c0 = 0; c0 < nstep; c0 += CH) {
    f32x4 b[CH][TPW];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = 16 * (c0 + c) + 4 * kk;
#pragma unroll
      for (int q = 0; q < TPW; ++q) {
        if constexpr (VEC == 4) {         // K % 4 == 0: rows are 16-byte aligned
          b[c][q] = (k < K) ? ld4(wrow[q] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else if constexpr (VEC == 2) {  // K % 2 == 0: 8-byte aligned pairs
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 v = f32x2{0.f, 0.f};
            if (k + 2 * h < K) v = *reinterpret_cast<const f32x2*>(wrow[q] + k + 2 * h);
            b[c][q][2 * h] = v[0];
            b[c][q][2 * h + 1] = v[1];
          }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) b[c][q][t] = (k + t < K) ? wrow[q][k + t] : 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (c0 + c < nstep) {
        const f32x4 a4 = ld4(xrow + 16 * (c0 + c));
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int q = 0; q < TPW; ++q) acc[q] = mfma4(a4[t], b[c][q][t], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = n0w + 16 * q + i;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) Ys[(kk * 4 + r) * ldy + col] = fmaxf(acc[q][r] + bv, 0.f);
  }
}

template <int WIDTH>
__device__ __forceinline__ void gemm_fwd_first(const float* __restrict__ Xs, int ldx, int K,
                                               const float* __restrict__ W,
                                               const float* __restrict__ bias,
                                               float* __restrict__ Ys, int ldy) {
  // W's base is 16-byte aligned (arena offsets are multiples of 4 floats only when
  // every earlier tensor is; check the pointer, not just K)
  const bool a16 = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (K % 4 == 0);
  const bool a8 = ((reinterpret_cast<uintptr_t>(W) & 7) == 0) && (K % 2 == 0);
  if (a16) gemm_fwd_first_v<WIDTH, 4>(Xs, ldx, K, W, bias, Ys, ldy);
  else if (a8) gemm_fwd_first_v<WIDTH, 2>(Xs, ldx, K, W, bias, Ys, ldy);
  else gemm_fwd_first_v<WIDTH, 1>(Xs, ldx, K, W, bias, Ys, ldy);
}

// ---------------------------------------------------------------------------
// Hidden layer: Y[kR, WIDTH] = relu(X[kR, WIDTH] · W[WIDTH, WIDTH]^T + b).
// Register ring of D macro steps (D*TPW = 32 b128 fragments = 128 VGPRs) keeps
// 8 KB per wave in flight.
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ void gemm_fwd_hidden(const float* __restrict__ Xs,
                                                const float* __restrict__ W,
                                                const float* __restrict__ bias,
                                                float* __restrict__ Ys) {
  constexpr int TPW = WIDTH / 64;
  constexpr int NSTEP = WIDTH / 16;
  constexpr int D = 32 / TPW;
  constexpr int LD = lds_ld(WIDTH);
  static_assert(NSTEP % D == 0, "ring depth must divide the step count");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int n0w = wave * (WIDTH / 4);
  f32x4 acc[TPW];
  const float* wrow[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    wrow[q] = W + (size_t)(n0w + 16 * q + i) * WIDTH + 4 * kk;
  }
  const float* xrow = Xs + i * LD + 4 * kk;
  // hipcc's scheduler sinks loads next to their first use (observed: effective
  // prefetch distance of one macro step, every step paying a full ~1100-cycle
  // miss).  sched_barrier(0) pins "issue step s+D's loads, then run step s".
  f32x4 ring[D][TPW];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int q = 0; q < TPW; ++q) ring[d][q] = ld4(wrow[q] + 16 * d);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    constexpr int dummy = 0; (void)dummy;
    const int d = s % D;
    f32x4 b4[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) b4[q] = ring[d][q];
    if (s + D < NSTEP) {
#pragma unroll
      for (int q = 0; q < TPW; ++q) ring[d][q] = ld4(wrow[q] + 16 * (s + D));
    }
    const f32x4 a4 = ld4(xrow + 16 * s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < TPW; ++q) acc[q] = mfma4(a4[t], b4[q][t], acc[q]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int col = n0w + 16 * q + i;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) Ys[(kk * 4 + r) * LD + col] = fmaxf(acc[q][r] + bv, 0.f);
  }
}

// ---------------------------------------------------------------------------
// Y[kR, N] = X[kR, WIDTH] · W[N, WIDTH]^T + b,  N <= kNarrowMax  (narrow output)
// The 4 waves split the contraction (WIDTH/64 macro steps each, all B fragments
// loaded up front); partial tiles meet in `scratch` ([kWaves][kR][kNarrowMax]
// floats of LDS).  Contains its own barriers; Y is complete on return.
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ void gemm_fwd_narrow(const float* __restrict__ Xs, int ldx,
                                                const float* __restrict__ W,
                                                const float* __restrict__ bias, int N,
                                                float* __restrict__ scratch,
                                                float* __restrict__ Ys, int ldy) {
  constexpr int TMAX = kNarrowMax / 16;
  constexpr int NS = WIDTH / 64;  // macro steps per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kk = lane >> 4;
  const int T = (N + 15) >> 4;
  const int kbeg = wave * (WIDTH / 4);
  f32x4 acc[TMAX];
  f32x4 b[NS][TMAX];
#pragma unroll
  for (int q = 0; q < TMAX; ++q) {
    acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int n = 16 * q + i;
    const bool valid = n < N;
    const float* wrow = W + (size_t)(valid ? n : N - 1) * WIDTH + kbeg + 4 * kk;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      b[s][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (q < T && valid) b[s][q] = ld4(wrow + 16 * s);
    }
  }
  const float* xrow = Xs + i * ldx + kbeg + 4 * kk;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const f32x4 a4 = ld4(xrow + 16 * s);
#pragma unroll
    for (int q = 0; q < TMAX; ++q)
      if (q < T) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[q] = mfma4(a4[t], b[s][q][t], acc[q]);
      }
  }
  float* part = scratch + wave * (kR * kNarrowMax);
#pragma unroll
  for (int q = 0; q < TMAX; ++q)
    if (q < T)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(kk * 4 + r) * kNarrowMax + 16 * q + i] = acc[q][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < kR * N; idx += kThreads) {
    const int row = idx / N, col = idx - row * N;
    float v = bias[col];
#pragma unroll
    for (int w = 0; w < kWaves; ++w) v += scratch[w * (kR * kNarrowMax) + row * kNarrowMax + col];
    Ys[row * ldy + col] = v;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// dX[kR, WIDTH] = (dY[kR, Ncon] · W[Ncon, WIDTH]) ⊙ (H > 0)       (wide output)
// dY in LDS, zero padded to round_up(Ncon,16) columns (pad MUST be zero: rows of
// W past Ncon are clamped, not skipped).  H = the ReLU output this gradient
// flows into (mask), read from LDS.  Output column of tile t / lane j is
// c0 + 4*j + t, so each lane ends up with 4 consecutive columns (one b128).
// Register ring of D macro steps (16*G*D VGPRs).  Writes dXs (LDS, may alias Hs)
// and, when dXg != nullptr, rows < nrows of the global [.,WIDTH] buffer.
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ void gemm_bwd_wide(const float* __restrict__ dYs, int ldy, int Ncon,
                                              const float* __restrict__ W,
                                              const float* Hs, int ldh,  // may alias dXs
                                              float* dXs, int ldx,
                                              float* __restrict__ dXg, int nrows) {
  constexpr int G = WIDTH / 256;  // 64-column groups per wave
  constexpr int D = 8 / G;        // ring depth in macro steps
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, c = lane >> 4;
  const int c0w = wave * (WIDTH / 4);
  f32x4 acc[G][4];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* yrow = dYs + j * ldy + 4 * c;
  const float* wcol = W + c0w + 4 * j;
  const int nstep = round_up(Ncon, 16) >> 4;
  f32x4 ring[D][4][G];
  auto fetch = [&](int d, int step) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int n = 16 * step + 4 * c + s;
      n = n < Ncon ? n : Ncon - 1;
#pragma unroll
      for (int g = 0; g < G; ++g) ring[d][s][g] = ld4(wcol + (size_t)n * WIDTH + 64 * g);
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nstep) fetch(d, d);
  __builtin_amdgcn_sched_barrier(0);
  for (int s0 = 0; s0 < nstep; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int step = s0 + d;
      if (step < nstep) {
        f32x4 b4[4][G];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < G; ++g) b4[s][g] = ring[d][s][g];
        if (step + D < nstep) fetch(d, step + D);
        const f32x4 a4 = ld4(yrow + 16 * step);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[g][t] = mfma4(a4[s], b4[s][g][t], acc[g][t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = c * 4 + r, col = c0w + 64 * g + 4 * j;
      const f32x4 h = ld4(Hs + row * ldh + col);
      f32x4 d;
#pragma unroll
      for (int t = 0; t < 4; ++t) d[t] = h[t] > 0.f ? acc[g][t][r] : 0.f;
      *reinterpret_cast<f32x4*>(dXs + row * ldx + col) = d;
      if (dXg != nullptr && row < nrows) *reinterpret_cast<f32x4*>(dXg + (size_t)row * WIDTH + col) = d;
    }
}

// ---------------------------------------------------------------------------
// out[kR, ncols] = dY[kR, WIDTH] · W1[WIDTH, Kin][:, col0 : col0+ncols]
// (gradient wrt a column range of the first layer's input — the action columns
// of a critic).  ncols <= kNarrowMax.  Waves split the contraction; all weight
// elements are fetched (predicated scalar loads) before the MFMA chain.
// ---------------------------------------------------------------------------
template <int WIDTH>
__device__ __forceinline__ void gemm_bwd_narrow(const float* __restrict__ dYs, int ldy,
                                                const float* __restrict__ W1, int Kin, int col0,
                                                int ncols, float* __restrict__ scratch,
                                                float* __restrict__ outS, int ldo) {
  constexpr int TMAX = kNarrowMax / 16;
  constexpr int NS = WIDTH / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, c = lane >> 4;
  const int T = (ncols + 15) >> 4;
  const int nbeg = wave * (WIDTH / 4);
  f32x4 acc[TMAX];
  float b[NS][4][TMAX];
#pragma unroll
  for (int q = 0; q < TMAX; ++q) {
    acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = 16 * q + j;
    const bool ok = q < T && col < ncols;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        b[st][s][q] = ok ? W1[(size_t)(nbeg + 16 * st + 4 * c + s) * Kin + col0 + col] : 0.f;
  }
  const float* yrow = dYs + j * ldy + nbeg + 4 * c;
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const f32x4 a4 = ld4(yrow + 16 * st);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int q = 0; q < TMAX; ++q)
        if (q < T) acc[q] = mfma4(a4[s], b[st][s][q], acc[q]);
  }
  float* part = scratch + wave * (kR * kNarrowMax);
#pragma unroll
  for (int q = 0; q < TMAX; ++q)
    if (q < T)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(c * 4 + r) * kNarrowMax + 16 * q + j] = acc[q][r];
  __syncthreads();
  for (int idx = threadIdx.x; idx < kR * ncols; idx += kThreads) {
    const int row = idx / ncols, col = idx - row * ncols;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) v += scratch[w * (kR * kNarrowMax) + row * kNarrowMax + col];
    outS[row * ldo + col] = v;
  }
  __syncthreads();
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

// store columns [0,k) of an LDS tile to rows [row0, ...) of a global matrix
__device__ __forceinline__ void store_rows(const float* __restrict__ Xs, int ldx,
                                           float* __restrict__ G, int ldg, int k, int row0, int B) {
  for (int idx = threadIdx.x; idx < kR * k; idx += kThreads) {
    const int row = idx / k, col = idx - row * k;
    const int gr = row0 + row;
    if (gr < B) G[(size_t)gr * ldg + col] = Xs[row * ldx + col];
  }
}

// float4 variant for WIDTH-wide tiles (k multiple of 4, ldg multiple of 4)
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
//   h[l] [kR][WL]  l < n_layers-1   hidden activations (ReLU outputs)
//   out  [kR][kOutLd]      network output
//   aux  [kR][kOutLd]      dLoss/d(out), later the input-column gradient
//   scr  [kWaves][kR][kNarrowMax]
// ---------------------------------------------------------------------------
constexpr int kX0Ld = lds_ld(96);   // widest layer-0 input: humanoid S+A = 88
constexpr int kOutLd = kNarrowMax + 8;

template <int WIDTH>
struct SliceLds {
  static constexpr int WL = lds_ld(WIDTH);
  static constexpr int h_off = kR * kX0Ld;
  static constexpr int hbuf = kR * WL;
  // n_h hidden buffers follow h_off, then out, aux (dout / dact) and scratch
  __host__ __device__ static constexpr int out_off(int n_h) { return h_off + n_h * hbuf; }
  __host__ __device__ static constexpr int aux_off(int n_h) { return out_off(n_h) + kR * kOutLd; }
  __host__ __device__ static constexpr int scr_off(int n_h) { return aux_off(n_h) + kR * kOutLd; }
  __host__ __device__ static constexpr int total(int n_h) { return scr_off(n_h) + kWaves * kR * kNarrowMax; }
};

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

// Forward.  x0 must be loaded (and zero padded) by the caller, barrier included.
// Hidden layer l's output goes to hbase + l*kR*WL.  If store_x the inputs of
// layers 1.. (the hidden activations) are also stored to Xg[l] ([B, WIDTH]).
// Result: outS[kR][kOutLd] columns [0, dims[L]).
template <int WIDTH, class Stamp>
__device__ __forceinline__ void mlp_forward_slice(const Net& net, const float* x0s, float* hbase,
                                                  float* outS, float* scr,
                                                  float* const (&Xg)[kMaxLayers], bool store_x,
                                                  int row0, int B, Stamp&& stamp) {
  constexpr int WL = lds_ld(WIDTH);
  constexpr int HB = kR * WL;
  const int L = net.n_layers;
  gemm_fwd_first<WIDTH>(x0s, kX0Ld, net.dims[0], net.W[0], net.b[0], hbase, WL);
  __syncthreads();
  stamp();
#pragma unroll
  for (int l = 1; l < kMaxLayers - 1; ++l) {
    if (l < L - 1) {
      gemm_fwd_hidden<WIDTH>(hbase + (l - 1) * HB, net.W[l], net.b[l], hbase + l * HB);
      __syncthreads();
      stamp();
    }
  }
  if (store_x) {
#pragma unroll
    for (int l = 1; l < kMaxLayers; ++l)
      if (l < L) store_rows4(hbase + (l - 1) * HB, WL, Xg[l], WIDTH, WIDTH, row0, B);
  }
  gemm_fwd_narrow<WIDTH>(hbase + (L - 2) * HB, WL, pick(net.W, L - 1), pick(net.b, L - 1),
                         pick(net.dims, L), scr, outS, kOutLd);
}

// Backward.  On entry doutS[kR][kOutLd] holds dLoss/d(out) with ZERO padding up
// to round_up(dims[L],16) columns and the hidden buffers hold the forward
// activations.  The gradient wrt hidden layer l's pre-activation output is
// written IN PLACE over hidden buffer l (each element's mask is read by the lane
// that overwrites it) and, if dYg[l] != nullptr, to dYg[l] ([B,WIDTH]) for the dW
// kernel; the caller stores dY[L-1] = dout itself.  If dact_cols > 0 the
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
  const int L = net.n_layers;
  const int nrows = B - row0;
  const float* dy = doutS;
  int ldy = kOutLd;
  int ncon = pick(net.dims, L);
#pragma unroll
  for (int l = kMaxLayers - 1; l >= 1; --l) {
    if (l <= L - 1) {
      float* dx = hbase + (l - 1) * HB;
      gemm_bwd_wide<WIDTH>(dy, ldy, ncon, net.W[l], dx, WL, dx, WL,
                           dYg[l - 1] != nullptr ? dYg[l - 1] + (size_t)row0 * WIDTH : nullptr,
                           nrows);
      __syncthreads();
      stamp();
      dy = dx;
      ldy = WL;
      ncon = WIDTH;
    }
  }
  if (dact_cols > 0)
    gemm_bwd_narrow<WIDTH>(dy, WL, net.W[0], net.dims[0], dact_col0, dact_cols, scr, dactS, kOutLd);
}

}  // namespace oprl
