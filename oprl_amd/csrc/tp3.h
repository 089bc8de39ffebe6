// tp3.h — a 16-row slice of a 3-layer MLP (K0 -> W -> W -> N, the actor / critic
// shape of DDPG, TD3, SAC) spread over a CLUSTER of nc compute units, tensor-parallel:
//
//   layer 0  (K0 -> W, K0 <= 96)   replicated: every member computes all of h1 (cheap)
//   layer 1  (W -> W, the 256 KB one)   column-parallel: member c computes columns
//            [c*W/nc, (c+1)*W/nc) of h2 from 1/nc of the weights with 1/nc of the MFMAs
//   layer 2  (W -> N, N <= 48)     row-parallel: member c contracts over ITS columns of
//            h2, giving a partial [16 x N]; the partials are summed by a tiny all-reduce
//   backward: dz2 is needed only for the member's own columns (local); the layer-1
//            backward dz1 = (dz2 · W2) ⊙ relu' contracts over the member's columns and
//            yields a PARTIAL [16 x W] per member — these are never reduced in-kernel:
//            each member stores its partial and k_dw_adam sums the nc buffers while
//            loading its A operand; the gradient wrt input columns (a critic's action
//            columns) is again a tiny all-reduce.
//
// So a forward costs one exchange of <= 16x48 floats and a backward at most one, instead
// of all-gathering 16 KB of hidden activations per layer (measured: 3.5 us per such
// exchange, slower than not clustering at all — profiles/r01b_experiments.txt).
//
// Exchange = data-is-the-flag granules (CDNA guide G16 R2): each value travels as ONE
// aligned 8-byte {tag, float} written with a relaxed agent-scope atomic store and read
// with relaxed agent-scope atomic loads until the tag matches; no fences; a fresh slot
// per exchange of a launch and a launch-unique tag; bounded spins (give-up -> NaN).
// The sum runs in member order on every member, so all members hold identical bits.
#pragma once
#include "engine.h"

namespace oprl {

constexpr int kTpStages = 8;                 // exchanges per launch (phase 2 uses 2)
constexpr int kTpBlk = kR * kNarrowMax;      // granules per member per exchange
constexpr int kTpSpin = 1 << 20;

struct Tp {
  int c, nc;                    // member index, cluster size (1: no cluster)
  unsigned long long* xbuf;     // this cluster's area: [kTpStages][nc][kTpBlk] granules
  unsigned tag;                 // launch-unique (26 bits used)
  int stage;
  // an expired wait is REPORTED (and the value poisoned with NaN): err = the learner's host-visible error
  // word (or null), err_code = kernel id << 8; spin = the wait bound (kTpSpin; 0 from the test hook)
  unsigned* err = nullptr;
  unsigned err_code = 0;
  int spin = kTpSpin;
  // every member of the cluster runs on the SAME XCD (the launch made sure): granules may be published at workgroup scope —
  // they reach the XCD's L2, where the peers' agent-scope polls find them, without the trip to memory (r04-16 / -26)
  bool local = false;
};

// Wait sites (low byte of the error word) and kernels (second byte): oprl_learner_check() decodes them.
enum WaitSite : unsigned { SITE_CLUSTER = 1, SITE_TD_TARGET = 2, SITE_TWIN_SPLIT = 3, SITE_P2_PAIR = 4, SITE_DW_TILE = 5, SITE_WINDOW = 6, SITE_DW_GATE = 7, SITE_LW_PAIR = 8,
                           SITE_X2_RANGE = 9 /* not a wait: an activation left the split-fp16 range (engine.h PrecX2) */ };
enum WaitKernel : unsigned { KERN_PHASE1 = 1, KERN_PHASE2 = 2, KERN_SLICE_TP = 3, KERN_DW_XCHG = 4, KERN_P2P = 5, KERN_LW_PAIR = 6 };
// First report wins (the word is host-mapped memory: one system-scope store, only ever on the error path).
__device__ __forceinline__ void report_expired(unsigned* err, unsigned code) {
  if (err != nullptr && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u)
    __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// P[kR][ncols] (LDS, leading dim kOutLd) holds this member's partial, complete and
// visible.  OUT[row][col] = sum over members (in member order) + bias[col].  OUT may be P.
__device__ __forceinline__ void tp_allreduce(float* P, int ncols, const float* __restrict__ bias,
                                             float* OUT, Tp& tp) {
  const int n = kR * ncols;
  if (tp.nc == 1) {
    for (int e = threadIdx.x; e < n; e += kThreads) {
      const int row = e / ncols, col = e - row * ncols;
      OUT[row * kOutLd + col] = P[row * kOutLd + col] + (bias != nullptr ? bias[col] : 0.f);
    }
    __syncthreads();
    return;
  }
  const unsigned tag = (tp.tag << 6) | (unsigned)(tp.stage & 63);
  unsigned long long* slot = tp.xbuf + (size_t)tp.stage * tp.nc * kTpBlk;
  for (int e = threadIdx.x; e < n; e += kThreads) {
    const int row = e / ncols, col = e - row * ncols;
    const float mine = P[row * kOutLd + col];
    const float bv = bias != nullptr ? bias[col] : 0.f;
    if (tp.local)      // (every member on one XCD: the granule need not leave its L2 — Tp::local)
      __hip_atomic_store(slot + (size_t)tp.c * kTpBlk + e,
                         ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_store(slot + (size_t)tp.c * kTpBlk + e,
                         ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mine),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float sum = 0.f;
    for (int m = 0; m < tp.nc; ++m) {
      float v = mine;
      if (m != tp.c) {
        const unsigned long long* g = slot + (size_t)m * kTpBlk + e;
        unsigned long long x = 0;
        bool ok = false;
        for (int spin = 0; spin < tp.spin; ++spin) {
          x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (unsigned)(x >> 32) == tag;
          if (ok) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) report_expired(tp.err, tp.err_code | SITE_CLUSTER);
        v = ok ? __uint_as_float((unsigned)x) : __builtin_nanf("");
      }
      sum += v;
    }
    OUT[row * kOutLd + col] = sum + bv;
  }
  tp.stage += 1;
  __syncthreads();
}

// columns [c0, c0+ncol) of an LDS tile -> the same columns of rows [row0, ..) of a global
// [B, ldg] matrix (float4; ncol, c0, ldg multiples of 4)
__device__ __forceinline__ void store_cols4(const float* __restrict__ Xs, int ldx,
                                            float* __restrict__ G, int ldg, int c0, int ncol,
                                            int row0, int B) {
  const int n4 = ncol >> 2;
  for (int idx = threadIdx.x; idx < kR * n4; idx += kThreads) {
    const int row = idx / n4, col = c0 + (idx - row * n4) * 4;
    const int gr = row0 + row;
    if (gr < B) *reinterpret_cast<f32x4*>(G + (size_t)gr * ldg + col) = ld4(Xs + row * ldx + col);
  }
}

// Where the tensor-parallel slice leaves what k_dw_adam needs (all optional).
struct Tp3Store {
  float* X1;        // [B][W]   h1 (stored by member 0)
  float* X2;        // [B][W]   h2 (each member its columns)
  float* dY1;       // [B][W]   dz2 (each member its columns)
  float* dY0;       // [nc][B][W] dz1 partials (member c writes buffer c)
  long dY0_stride;  // floats between two members' partial buffers
  // > 0 (tp4 passes only): the partial buffers are TILE-MAJOR, [16-column tile][dY0_tile_rows rows][16] —
  // what a layer-0 tile of k_dw_adam reads from a member's buffer is then one contiguous run of full
  // cache lines instead of 64-byte pieces of 1 KB rows (DwArgs::dy_tiled).  0: row-major [B][W].
  int dY0_tile_rows = 0;
  // tp4 passes only: X / dY rows leave with write-through stores (sc1) — their consumer is a tile workgroup of
  // the SAME launch (merged phase kernels, csrc/fused_ddpg.hip), possibly on another XCD, with no kernel
  // boundary in between
  bool wt = false;
};

// Forward.  x0s: [kR][kX0Ld] input tile (zero padded; no barrier needed).  On return h1
// holds relu(layer 0) (full), h2 the member's columns of relu(layer 1) (other columns
// unspecified), outS[kR][kOutLd] columns [0,N) the network output, identical on all members.
template <int WIDTH, class ST = NoStamp>
__device__ __forceinline__ void tp3_forward(const Net& net, const float* x0s, float* h1, float* h2,
                                            float* outS, float* scr, Tp& tp, const Tp3Store& st,
                                            int row0, int B, ST sf = ST()) {
  constexpr int WL = lds_ld(WIDTH);
  constexpr int NTW = WIDTH / 16;
  const int tpc = NTW / tp.nc;          // tiles (= macro steps) per member
  const int c0 = tp.c * tpc * 16;
  const int N = net.dims[3];
  gemm_packed(x0s, kX0Ld, net.pf[0], NTW, cdiv(net.dims[0], 16), scr, net.b[0], WIDTH,
              [&](int row, int col, float v) { h1[row * WL + col] = fmaxf(v, 0.f); });
  // (the next GEMM takes the barrier that publishes h1)
  gemm_packed(h1, WL, net.pf[1] + (size_t)tp.c * tpc * NTW * 256, tpc, NTW, scr, net.b[1] + c0,
              tpc * 16, [&](int row, int col, float v) { h2[row * WL + c0 + col] = fmaxf(v, 0.f); });
  if (tpc >= kWaves) __syncthreads();   // wide path (nc == 1) does not end with a barrier
  if (st.X1 != nullptr && tp.c == 0) store_rows4(h1, WL, st.X1, WIDTH, WIDTH, row0, B);
  if (st.X2 != nullptr) store_cols4(h2, WL, st.X2, WIDTH, c0, tpc * 16, row0, B);
  // row-parallel output layer: contract over this member's columns only
  gemm_packed(h2 + c0, WL, net.pf[2] + (size_t)tp.c * tpc * 256, cdiv(N, 16), tpc, scr, nullptr, 0,
              [&](int row, int col, float v) { outS[row * kOutLd + col] = col < N ? v : 0.f; },
              NoStamp(), NTW * 256);
  tp_allreduce(outS, N, net.b[2], outS, tp);
  sf();
}

// Backward.  doutS[kR][kOutLd]: dLoss/d(out), zero padded to a multiple of 16 columns (no
// barrier needed).  h1 (full) and the member's columns of h2 hold the forward activations
// and are overwritten by the gradients.  If dact_cols > 0, dactS[kR][kOutLd] columns
// [0, dact_cols) receive the gradient wrt input columns [dact_col0, +dact_cols), reduced
// over the cluster (identical on all members).
template <int WIDTH, class ST = NoStamp>
__device__ __forceinline__ void tp3_backward(const Net& net, const float* doutS, float* h1,
                                             float* h2, float* scr, Tp& tp, const Tp3Store& st,
                                             int row0, int B, int dact_col0, int dact_cols,
                                             float* dactS, ST sf = ST()) {
  constexpr int WL = lds_ld(WIDTH);
  constexpr int NTW = WIDTH / 16;
  const int tpc = NTW / tp.nc;
  const int c0 = tp.c * tpc * 16;
  const int N = net.dims[3], NSo = cdiv(N, 16);
  // dz2[:, mine] = (dout · W3^T)[:, mine] ⊙ (h2 > 0), in place
  gemm_packed(doutS, kOutLd, net.pb[2] + (size_t)tp.c * tpc * NSo * 256, tpc, NSo, scr, nullptr, 0,
              [&](int row, int col, float v) {
                float* p = h2 + row * WL + c0 + col;
                *p = *p > 0.f ? v : 0.f;
              });
  if (tpc >= kWaves) __syncthreads();
  if (st.dY1 != nullptr) store_cols4(h2, WL, st.dY1, WIDTH, c0, tpc * 16, row0, B);
  // dz1 partial = (dz2[:, mine] · W2[mine, :]) ⊙ (h1 > 0): all W output columns, contraction
  // over this member's columns = macro steps [c*tpc, (c+1)*tpc) of the W2^T pack
  gemm_packed(h2 + c0, WL, net.pb[1] + (size_t)tp.c * tpc * 256, NTW, tpc, scr, nullptr, 0,
              [&](int row, int col, float v) {
                float* p = h1 + row * WL + col;
                *p = *p > 0.f ? v : 0.f;
              },
              NoStamp(), NTW * 256);
  __syncthreads();
  if (st.dY0 != nullptr)
    store_rows4(h1, WL, st.dY0 + (size_t)tp.c * st.dY0_stride, WIDTH, WIDTH, row0, B);
  if (dact_cols > 0) {
    // partial gradient wrt the input columns from this member's partial dz1, then all-reduce
    gemm_packed(h1, WL, net.pb[0], cdiv(net.dims[0], 16), NTW, scr, nullptr, 0,
                [&](int row, int col, float v) {
                  const int c = col - dact_col0;
                  if (c >= 0 && c < dact_cols) dactS[row * kOutLd + c] = v;
                });
    tp_allreduce(dactS, dact_cols, nullptr, dactS, tp);
  }
  sf();
}

}  // namespace oprl
