// dw_wide.hip — dW + Adam + Polyak for wide layers (TQC's 512x512 quantile-critic layers).
//
// k_dw_adam's 16(n) x 32(k) tiles are sized for the 256-wide nets of the headline path, where the
// launch is a latency chain and more, smaller workgroups win.  On a 512x512 layer every tile
// re-reads 48 KB of X / dY rows for 512 outputs: 24 MB per layer, 240 MB for TQC's ten wide layers
// (72 us per critic step, profiles/r01g_kernel_stats_tqc.csv).  Here a workgroup owns a 64 x 64
// tile (128 KB of rows for 4096 outputs, 8 MB per layer): eight waves, each two 16x16 MFMA tiles
// sharing the dY operand, the minibatch walked in 64-row chunks through LDS; the epilogue is the
// same torch-semantics Adam + Polyak as k_dw_adam, eight consecutive k per thread as float4
// accesses, the three fragment-order packs written block by block from the staged tile.
// Reference ops replaced: loss.backward()'s addmm-backward + optim.Adam.step + soft_update
// (tqc.py:150-177, nn_functions.py:5-10).
#include <cstring>
#include "kernels.h"
#include "dw_body.h"

namespace oprl {

constexpr int kWT = 64;                  // tile extent in n and in k
constexpr int kWLd = kWT + 16;           // staging rows: 80 floats, the four 16-lane groups of a read hit distinct banks
constexpr int kWChunk = 64;              // minibatch rows per staging round
constexpr int kWThreads = 512;
constexpr int kWMaxItems = 10;

struct DwWideArgs {
  int tile_end[kWMaxItems];
  DwItem items[kWMaxItems];
  int n_items, B;
  int total;                             // 64 x 64 tiles of this kernel's own work
  int grid_own;                          // 8 * ceil(total / 8) workgroups carry them (riding workgroups come after)
  AdamScalars ad;
};
static_assert(kWThreads == kDwThreads, "the riding k_dw_adam tiles run in this kernel's workgroups");
#ifdef DWW_SMALL_LDS      // tools/ubench_dw_wide.hip: the wide tiles alone at three workgroups per compute unit (r06-11)
constexpr int kWLdsFloats = 2 * kWChunk * kWLd;
#define DWW_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))
#else
constexpr int kWLdsFloats = 2 * kWChunk * kWLd > kDwLdsFloats ? 2 * kWChunk * kWLd : kDwLdsFloats;
#define DWW_ATTR
#endif

// `N`: the SAME update's narrow layers (TQC's critics: the 30 x 512 input layers and the 512 x 25 heads), whose
// 16 x 32 tiles of k_dw_adam (dw_body.h) ride as the workgroups past `total` — dispatched last, they fill the
// CUs the 640 wide tiles' second round leaves idle, instead of a launch of their own (9 us).  N.n_items == 0:
// nothing rides.
#ifdef DWW_TRACE
__device__ unsigned long long g_dww_trace[1024 * 8];
#endif
__global__ __launch_bounds__(kWThreads) DWW_ATTR void k_dw_adam_wide(const DwWideArgs A, const DwKArgs N) {
  __shared__ __attribute__((aligned(16))) float lds[kWLdsFloats];
#ifndef DWW_SMALL_LDS
  if ((int)blockIdx.x >= A.grid_own) {
    const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    dw_adam_body<false>(*(const DwKArgs*)(kp + ((sizeof(DwWideArgs) + alignof(DwKArgs) - 1) / alignof(DwKArgs)) * alignof(DwKArgs)),
                        lds, (int)blockIdx.x - A.grid_own);
    return;
  }
#endif
  float (*stA)[kWLd] = reinterpret_cast<float (*)[kWLd]>(lds);                      // dY rows [b][n]; later the new W tile
  float (*stX)[kWLd] = reinterpret_cast<float (*)[kWLd]>(lds + kWChunk * kWLd);     // X  rows [b][k]; later the new target tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const DwWideArgs* KA = (const DwWideArgs*)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef DWW_TRACE
#define DWW_STAMP(k) do { if (tid == 0) g_dww_trace[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DWW_STAMP(k) do { } while (0)
#endif
  DWW_STAMP(0);
  // Which tile: workgroup b runs on XCD b % 8, and a tile streams 64 KB of dY rows and 64 KB of X rows that its
  // row / column neighbours share.  With tiles dealt out in launch order every XCD read ALL of dY (5.2 MB for TQC's
  // ten layers, more than its 4 MB L2): 80+ MB of rows over the fabric per launch, as much as the Adam state and
  // the packs together, and the launch is bound by exactly that sum.  Here the tile list (layer-major, 4 x 4
  // super-blocks inside a layer) is cut into eight consecutive pieces, one per XCD: a piece is a whole layer plus
  // a super-block or two — 1.5 MB of rows, read once.
  const int per = (A.total + 7) >> 3;
  const int pos = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (pos >= A.total) return;
  int item = 0;
  for (int j = 0; j + 1 < A.n_items; ++j) item += pos >= KA->tile_end[j] ? 1 : 0;
  const DwItem I = KA->items[item];
  const int lt = pos - (item > 0 ? KA->tile_end[item - 1] : 0);
  const int tiles_k = I.K / kWT, tiles_n = I.N / kWT;
  int tn, tk;
  if (((tiles_k | tiles_n) & 3) == 0) {
    const int sb = lt >> 4, r = lt & 15, sbk = tiles_k >> 2, sbn = sb / sbk;
    tn = sbn * 4 + (r >> 2);
    tk = (sb - sbn * sbk) * 4 + (r & 3);
  } else {
    tn = lt / tiles_k;
    tk = lt - tn * tiles_k;
  }
  const int n_base = tn * kWT, k_base = tk * kWT;
  const int NSk = I.K >> 4, NSn = I.N >> 4;
  const bool polyak = A.ad.do_polyak && I.w_t != nullptr;
  const int B = A.B;

  // ---- this thread's 8 epilogue elements: row en, columns ek .. ek+7; Adam state requested now
  const int el_n = tid >> 3, el_k = (tid & 7) * 8;
  const size_t eo = (size_t)(n_base + el_n) * I.K + k_base + el_k;
  f32x4 p_th[2], p_m[2], p_v[2], p_tt[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) p_th[h] = p_m[h] = p_v[h] = p_tt[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  // requested behind the LAST chunk's rows (below): ahead of the first rows — loads return in order — the 64 KB
  // of state per workgroup, 42 MB from HBM over the chip, stood between the launch and its first MFMA; there
  // they arrive under the last chunks' MFMAs, when the memory pipe has nothing else to do
  auto request_state = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (A.ad.do_adam) {
        p_th[h] = ld4(I.w + eo + 4 * h);
        p_m[h] = ld4(I.w_m + eo + 4 * h);
        p_v[h] = ld4(I.w_v + eo + 4 * h);
        if (polyak) p_tt[h] = ld4(I.w_t + eo + 4 * h);
      }
  };

  // ---- dW tile = sum_b dY[b, n]^T X[b, k].  Wave w: n block w >> 1, k blocks 2 (w & 1) + {0, 1}.
  // MFMA step u of a chunk contracts rows 4u .. 4u+3: lane (c, i) feeds dY[4u + c][n] as A and
  // X[4u + c][k], X[4u + c][k + 16] as the two B operands.
  const int i = lane & 15, c = lane >> 4;
  const int nb = wave >> 1, kb = 2 * (wave & 1);
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  float sA = 0.f;
  // staging: 64 rows x 16 float4 per matrix = 2 float4 per thread per matrix
  const int sr = tid >> 4, sc4 = (tid & 15) * 4;
  f32x4 va[2], vx[2];
  auto request = [&](int chunk) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int bb = chunk * kWChunk + sr + 32 * h;
      va[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      vx[h] = va[h];
      if (bb < B) {
        va[h] = ld4(I.dY + (size_t)bb * I.ldy + n_base + sc4);
        vx[h] = ld4(I.X + (size_t)bb * I.ldx + k_base + sc4);
      }
    }
  };
  const int n_chunks = (B + kWChunk - 1) / kWChunk;
  request(0);
#ifndef DWW_LATE_STATE
  if (n_chunks <= 1) request_state();
#endif
  for (int chunk = 0; chunk < n_chunks; ++chunk) {
    __syncthreads();                      // the previous chunk's reads are done
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(&stA[sr + 32 * h][sc4]) = va[h];
      *reinterpret_cast<f32x4*>(&stX[sr + 32 * h][sc4]) = vx[h];
    }
    if (chunk + 1 < n_chunks) {
      request(chunk + 1);                           // in flight during this chunk's MFMAs
#ifndef DWW_LATE_STATE
      if (chunk + 2 == n_chunks) request_state();
#endif
    }
    __syncthreads();
    if (chunk == 0) DWW_STAMP(1);
#pragma unroll
    for (int u = 0; u < kWChunk / 4; ++u) {
      const float av = stA[4 * u + c][nb * 16 + i];
      const float x0 = stX[4 * u + c][kb * 16 + i];
      const float x1 = stX[4 * u + c][kb * 16 + 16 + i];
      sA += av;
#ifdef DWW_NO_MFMA           // tools/ubench_dw_wide.hip: what the launch costs without its matrix time (r06-11)
      acc[0][u & 3] += av * x0;
      acc[1][u & 3] += av * x1;
#else
      acc[0] = mfma4(av, x0, acc[0]);
      acc[1] = mfma4(av, x1, acc[1]);
#endif
    }
  }
#ifdef DWW_LATE_STATE
  request_state();
#endif
  __syncthreads();
  DWW_STAMP(2);
  // gradient tile -> stA[n][k] (acc row = n index 4c + r, column = k index i); db -> stX[0][n]
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) stA[nb * 16 + c * 4 + r][(kb + w) * 16 + i] = acc[w][r];
  {
    float sb = sA;
    sb += __shfl_xor(sb, 16);
    sb += __shfl_xor(sb, 32);
    if ((wave & 1) == 0 && c == 0) stX[0][nb * 16 + i] = sb;
  }
  __syncthreads();

  // ---- epilogue
  const float step_size = A.ad.step_size_host, bc2_sqrt = A.ad.bc2_sqrt_host;
  f32x4 th_new[2], tt_new[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 g = ld4(&stA[el_n][el_k + 4 * h]) * A.ad.grad_scale;
    th_new[h] = p_th[h];
    tt_new[h] = p_tt[h];
    if (I.w_g != nullptr) *reinterpret_cast<f32x4*>(I.w_g + eo + 4 * h) = g;
    if (A.ad.do_adam) {
      f32x4 mm, vv, th;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float me = p_m[h][t], ve = p_v[h][t], te = p_th[h][t];
        adam_elem(g[t], me, ve, te, A.ad, step_size, bc2_sqrt);
        mm[t] = me; vv[t] = ve; th[t] = te;
      }
      *reinterpret_cast<f32x4*>(I.w_m + eo + 4 * h) = mm;
      *reinterpret_cast<f32x4*>(I.w_v + eo + 4 * h) = vv;
      *reinterpret_cast<f32x4*>(I.w + eo + 4 * h) = th;
      th_new[h] = th;
      if (polyak) {
#pragma unroll
        for (int t = 0; t < 4; ++t) tt_new[h][t] = polyak_elem(p_tt[h][t], th[t], A.ad);
        *reinterpret_cast<f32x4*>(I.w_t + eo + 4 * h) = tt_new[h];
      }
    }
  }
  DWW_STAMP(3);
  // bias (one k tile per n range does it)
  float gb = 0.f;
  if (tk == 0 && tid < kWT) gb = stX[0][tid];
  __syncthreads();                        // gradient tile and db consumed
  if (tk == 0 && tid < kWT) {
    const int n = n_base + tid;
    float t0, t1;
    (void)adam_polyak_elem(gb, I.b + n, I.b_m ? I.b_m + n : nullptr, I.b_v ? I.b_v + n : nullptr,
                           I.b_t ? I.b_t + n : nullptr, I.b_g ? I.b_g + n : nullptr, A.ad, step_size, bc2_sqrt,
                           &t0, &t1);
  }
  if (!(A.ad.do_adam && (I.pf != nullptr || I.pf16 != nullptr))) return;
  // ---- packs: the new tile(s) staged as [n][k], then 16 blocks of 16x16 per pack in pack order
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    *reinterpret_cast<f32x4*>(&stA[el_n][el_k + 4 * h]) = th_new[h];
    *reinterpret_cast<f32x4*>(&stX[el_n][el_k + 4 * h]) = tt_new[h];
  }
  __syncthreads();
  const int l = tid & 63, li = l & 15, lk = l >> 4;
  for (int job = tid >> 6; I.pf != nullptr && job < 48; job += kWThreads / 64) {      // (pf null: the fp32 packs are not kept current)
    const int which = job >> 4, blk = job & 15, bn = blk >> 2, bk = blk & 3;   // block (n block, k block)
    if (which == 1) {            // W^T pack: tiles over k, steps over n
      if (I.pb == nullptr) continue;
      f32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) v[t] = stA[bn * 16 + 4 * lk + t][bk * 16 + li];
      *reinterpret_cast<f32x4*>(I.pb + (((size_t)((k_base >> 4) + bk) * NSn + (n_base >> 4) + bn) * 64 + l) * 4) = v;
    } else {                     // W pack (online, target): tiles over n, steps over k
      float* dst = which == 0 ? I.pf : (polyak ? I.tpf : nullptr);
      if (dst == nullptr) continue;
      float (*src)[kWLd] = which == 0 ? stA : stX;
      *reinterpret_cast<f32x4*>(dst + (((size_t)((n_base >> 4) + bn) * NSk + (k_base >> 4) + bk) * 64 + l) * 4) =
          *reinterpret_cast<const f32x4*>(&src[bn * 16 + li][bk * 16 + 4 * lk]);
    }
  }
  DWW_STAMP(4);
  if (I.pf16 == nullptr) return;
  const int NSk2 = I.K >> 5, NSn2 = I.N >> 5;
  if (I.x2) {
    // ---- split-fp16 packs (PrecX2, engine.h): the bf16 packs' blocks and element order, every block two planes of
    // 64 lanes x 8 fp16 — hi = fp16(2^8 w), lo = fp16(2^8 w - hi), the lo plane 256 floats behind the hi plane
    for (int job = tid >> 6; job < 24; job += kWThreads / 64) {
      const int which = job >> 3, blk = job & 7, b16 = blk >> 1, b32 = blk & 1;   // (16-row block, 32-column step)
      f32x4 lo4, hi4;
      float* d;
      if (which == 1) {            // W^T: rows = k (16-row tile b16), steps = n (32 wide, b32)
        if (I.pb16 == nullptr) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          lo4[t] = stA[b32 * 32 + 4 * lk + t][b16 * 16 + li];
          hi4[t] = stA[b32 * 32 + 16 + 4 * lk + t][b16 * 16 + li];
        }
        d = I.pb16 + ((size_t)((k_base >> 4) + b16) * NSn2 + (n_base >> 5) + b32) * 512 + (size_t)l * 4;
      } else {
        float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
        if (dst == nullptr) continue;
        float (*src)[kWLd] = which == 0 ? stA : stX;
        lo4 = *reinterpret_cast<const f32x4*>(&src[b16 * 16 + li][b32 * 32 + 4 * lk]);
        hi4 = *reinterpret_cast<const f32x4*>(&src[b16 * 16 + li][b32 * 32 + 16 + 4 * lk]);
        d = dst + ((size_t)((n_base >> 4) + b16) * NSk2 + (k_base >> 5) + b32) * 512 + (size_t)l * 4;
      }
      f16x8 ph, pl;
      x2_split8(lo4 * PrecX2::kWScale, hi4 * PrecX2::kWScale, ph, pl);
      *reinterpret_cast<f16x8*>(d) = ph;
      *reinterpret_cast<f16x8*>(d + 256) = pl;
    }
    return;
  }
  // ---- bf16 packs (PrecBF16, engine.h): 8 blocks of 16 x 32 per pack, a block = fp32 blocks 2b, 2b + 1 side by side
  for (int job = tid >> 6; job < 24; job += kWThreads / 64) {
    const int which = job >> 3, blk = job & 7, b16 = blk >> 1, b32 = blk & 1;   // (16-row block, 32-column step)
    if (which == 1) {            // W^T: rows = k (16-row tile b16), steps = n (32 wide, b32)
      if (I.pb16 == nullptr) continue;
      f32x4 lo, hi;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        lo[t] = stA[b32 * 32 + 4 * lk + t][b16 * 16 + li];
        hi[t] = stA[b32 * 32 + 16 + 4 * lk + t][b16 * 16 + li];
      }
      *reinterpret_cast<bf16x8*>(I.pb16 + (((size_t)((k_base >> 4) + b16) * NSn2 + (n_base >> 5) + b32) * 64 + l) * 4) =
          cvt_bf16x8(lo, hi);
    } else {
      float* dst = which == 0 ? I.pf16 : (polyak ? I.tpf16 : nullptr);
      if (dst == nullptr) continue;
      float (*src)[kWLd] = which == 0 ? stA : stX;
      *reinterpret_cast<bf16x8*>(dst + (((size_t)((n_base >> 4) + b16) * NSk2 + (k_base >> 5) + b32) * 64 + l) * 4) =
          cvt_bf16x8(*reinterpret_cast<const f32x4*>(&src[b16 * 16 + li][b32 * 32 + 4 * lk]),
                     *reinterpret_cast<const f32x4*>(&src[b16 * 16 + li][b32 * 32 + 16 + 4 * lk]));
    }
  }
}

// Is this layer one for the wide kernel?  (Full 64 x 64 tiles; no dz1 partial buffers, no per-row seed.)
bool dw_wide_item_ok(const DwItem& it, const DwArgs& a) {
  if (a.apply_only || a.B < 1) return false;
  if (it.N % kWT != 0 || it.K % kWT != 0 || it.N < 512 || it.K < 512) return false;   // (256-wide layers: k_dw_adam, one launch)
  if (it.dY_part_stride > 0 && a.n_part > 1) return false;
  if (it.scaled != 0 && a.use_row_scale != 0) return false;
  if (it.ldx % 4 != 0 || it.ldy % 4 != 0) return false;
  return true;
}

// ride / ride_blocks: a filled k_dw_adam argument block and its grid (tiles, + 1 with a temperature job), or null / 0
hipError_t launch_dw_adam_wide(const DwItem* items, int n_items, int B, const AdamScalars& ad, hipStream_t st,
                               const DwKArgs* ride, int ride_blocks) {
  if (n_items < 1 || n_items > kWMaxItems) return hipErrorInvalidValue;
  DwWideArgs k;
  int total = 0;
  for (int j = 0; j < n_items; ++j) {
    k.items[j] = items[j];
    total += (items[j].N / kWT) * (items[j].K / kWT);
    k.tile_end[j] = total;
  }
  for (int j = n_items; j < kWMaxItems; ++j) { k.items[j] = items[0]; k.tile_end[j] = total; }
  k.n_items = n_items; k.B = B; k.ad = ad; k.total = total;
  k.grid_own = 8 * ((total + 7) / 8);
  static const DwKArgs none = [] { DwKArgs z; std::memset((void*)&z, 0, sizeof z); return z; }();
  if (ride == nullptr) ride_blocks = 0;
  hipLaunchKernelGGL(k_dw_adam_wide, dim3(k.grid_own + ride_blocks), dim3(kWThreads), 0, st, k, ride != nullptr ? *ride : none);
  return hipGetLastError();
}

}  // namespace oprl
