// kernels.hip — gfx950 kernels of the off-policy learner hot path.
//
//   k_mlp_slice<WIDTH>   one workgroup per 16-row minibatch slice: input assembly
//                        -> whole-MLP forward -> policy-head / TD epilogue ->
//                        loss-gradient seed -> whole-MLP backward (engine.h)
//   k_dw_adam            dW = dY^T X on 16x32 tiles (fp32 MFMA) fused with
//                        torch-semantics Adam, Polyak target update, grad export
//   k_adam_flat / k_polyak_flat / k_alpha_step / k_reduce_partials   small fused
//                        elementwise pieces
//   k_tqc_target         row-wise bitonic sort of 125 quantiles + truncation + TD
//
// Reference op groups replaced: SURVEY.md §2.2 K2-K13.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "kernels.h"
#include "philox.h"
#include "slice_head.h"
#include "tp3.h"
#include "dw_body.h"
#include "batch_rows.h"

namespace oprl {

template <int WIDTH>
__device__ __forceinline__ void mlp_slice_body(const MlpArgs& A) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LY = SliceLds<WIDTH>;
  constexpr int WL = lds_ld(WIDTH);
  const int L = A.net.n_layers, nh = L - 1;
  const int Nout = pick(A.net.dims, L);
  float* x0s = smem;
  float* hb = smem + LY::h_off;
  float* outS = smem + LY::out_off(nh);
  float* auxS = smem + LY::aux_off(nh);
  float* scr = smem + LY::scr_off(nh);
  const int row0 = blockIdx.x * kR;
  const int B = A.B;
  const int tid = threadIdx.x;
  int n_stamp = 0;
  auto stamp = [&]() {
    if (kTraceOn && A.trace != nullptr && tid == 0 && n_stamp < kTraceStamps) {
      long long* t = A.trace + ((size_t)blockIdx.x * kTraceStamps + n_stamp) * 2;
      t[0] = (long long)__builtin_readcyclecounter();
      t[1] = (long long)wall_clock64();
    }
    ++n_stamp;
  };
  stamp();  // 0: entry

  if (A.do_fwd) {
    lds_zero(x0s, kR * kX0Ld);
    __syncthreads();
    load_rows(x0s, kX0Ld, 0, A.x0, A.k0, A.k0, row0, B);
    if (A.x1 != nullptr) load_rows(x0s, kX0Ld, A.k0, A.x1, A.k1, A.k1, row0, B);
    stamp();  // 1: inputs in LDS
    mlp_forward_slice<WIDTH>(A.net, x0s, hb, outS, scr, A.Xg, A.Xg[1] != nullptr, row0, B, stamp);
    if (A.Xg[0] != nullptr) store_rows(x0s, kX0Ld, A.Xg[0], A.ldx0, A.net.dims[0], row0, B);
    stamp();  // after narrow output layer
    slice_head(A, outS, Nout, row0, true);
  } else if (A.do_bwd) {
    _Pragma("unroll") for (int l = 1; l < kMaxLayers; ++l)
      if (l < L) load_rows4(hb + (l - 1) * LY::hbuf, WL, A.Xg[l], WIDTH, WIDTH, row0, B);
  }
  stamp();  // head / reload done
  if (!A.do_bwd) return;

  slice_seed(A, outS, auxS, scr, Nout, L, row0, blockIdx.x, true);
  stamp();  // seed done
  mlp_backward_slice<WIDTH>(A.net, auxS, hb, scr, A.dYg, row0, B, A.dact_col0, A.dact_cols, auxS, stamp);
  if (A.dact_cols > 0 && A.dact != nullptr) store_rows(auxS, kOutLd, A.dact, A.lddact, A.dact_cols, row0, B);
  stamp();  // end
}

template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_mlp_slice(const MlpArgs A) { mlp_slice_body<WIDTH>(A); }

// Up to kMaxMulti independent nets of one learner (TQC's five quantile critics) in ONE launch,
// grid (slices, nets): side streams gave them only ~1.7x overlap (5 streams share 4 hardware
// queues, every fork/join is an event round trip).  The argument blocks travel by value; a
// workgroup addresses its own through the kernarg segment pointer (scalar loads, as k_dw_adam does).
template <int WIDTH>
__global__ __launch_bounds__(kThreads) void k_mlp_slice_multi(const MlpMultiArgs M) {
  const MlpMultiArgs* kp = (const MlpMultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  mlp_slice_body<WIDTH>(kp->a[blockIdx.y]);
}

template __global__ void k_mlp_slice<256>(const MlpArgs);
template __global__ void k_mlp_slice<512>(const MlpArgs);
template __global__ void k_mlp_slice_multi<256>(const MlpMultiArgs);
template __global__ void k_mlp_slice_multi<512>(const MlpMultiArgs);

// Pj (blocks Pj.z0 ..): the next update's minibatch rows gathered from the replay by riding workgroups, one per 16-row
// slice (batch_rows.h) — where the phase-2 launch that otherwise carries them over-subscribes the chip (B = 1024)
constexpr size_t kDwPjOffset = (sizeof(DwKArgs) + alignof(PrefetchJob) - 1) / alignof(PrefetchJob) * alignof(PrefetchJob);
static_assert(kDwLdsFloats >= kMaxEnds + 2 * kR, "the riders' LDS fits in the launch's");
template <bool XCHG>
__global__ __launch_bounds__(kDwThreads) void k_dw_adam(const DwKArgs A, const PrefetchJob Pj) {
  // (through the kernel-argument segment pointer: a dynamic index into the by-value table is then a scalar load)
  __shared__ __attribute__((aligned(16))) float dw_lds[kDwLdsFloats];
  if (Pj.z0 >= 0 && (int)blockIdx.x >= Pj.z0) {
    prefetch_rows_direct(*(const PrefetchJob*)((const char*)__builtin_amdgcn_kernarg_segment_ptr() + kDwPjOffset),
                         (int)blockIdx.x - Pj.z0, reinterpret_cast<int*>(dw_lds), kDwThreads);
    return;
  }
  dw_adam_body<XCHG>(*(const DwKArgs*)__builtin_amdgcn_kernarg_segment_ptr(), dw_lds, (int)blockIdx.x);
}

// N learners' dW + Adam launches as one (grid.z = learner; argument blocks in device memory; NI = 4: one net of up
// to four layers (DDPG's critic, every actor), 8: twin critics).  grid.x = the tiles + 1: the last workgroup is the
// temperature step of a SAC member (AlphaJob), if it has one.
template <int NI>
__global__ __launch_bounds__(kDwThreads) void k_dw_adam_group(const DwKArgsN<NI>* __restrict__ batch) {
  // member l on XCD l % 8 (its tiles share their X / dY rows in that XCD's L2: k_ddpg_phase1_group, csrc/fused_ddpg.hip)
  int l = (int)blockIdx.z, bx = (int)blockIdx.x;
  if ((gridDim.z & 7) == 0) {
    const int q = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.z, p = q >> 3;
    l = (q & 7) + 8 * (p / (int)gridDim.x);
    bx = p % (int)gridDim.x;
  }
  const DwKArgsN<NI>& A = batch[l];
  const int total = A.tile_end[NI - 1];                          // (entries past the last item hold the total)
  if (bx > total || (bx == total && A.alpha.log_alpha == nullptr)) return;
  __shared__ __attribute__((aligned(16))) float dw_lds[kDwLdsFloats];
  dw_adam_body<false, 0, kDwWaves, DwKArgsN<NI>>(A, dw_lds, bx);
}

// flat Adam over an arena (data-parallel apply after the all-reduce; alpha-free)
__global__ void k_adam_flat(float* th, float* m, float* v, float* tt, const float* g, long n,
                            const AdamScalars ad) {
  __shared__ float sc[2];
  if (threadIdx.x == 0) adam_bias_corr(ad, &sc[0], &sc[1]);
  __syncthreads();
  const float step_size = sc[0], bc2_sqrt = sc[1];
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x)
  {
    float t0, t1;
    (void)adam_polyak_elem(g[idx], th + idx, m + idx, v + idx, tt ? tt + idx : nullptr, nullptr, ad,
                           step_size, bc2_sqrt, &t0, &t1);
  }
}

__global__ void k_polyak_flat(float* tt, const float* th, long n, float tau, float omtau) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (long)gridDim.x * blockDim.x)
    tt[idx] = __builtin_fmaf(tt[idx], omtau, tau * th[idx]);      // (polyak_elem's roundings, kernels.h)
}

__global__ void k_alpha_step(double* log_alpha, double* m, double* v, const float* logp, int B,
                             float target_entropy, double lr, double beta1, double beta2, double eps,
                             double bc1, double bc2_sqrt, double* grad_out, const double* grad_in,
                             float grad_scale) {
  alpha_step_block(log_alpha, m, v, logp, B, target_entropy, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_out, grad_in,
                   grad_scale);
}

// out[0..3] = sum over slices of partials[.][0..3]  (then scaled by the caller)
__global__ void k_reduce_partials(const float* partials, int n_slices, float* out, int out_off,
                                  float scale_loss, float scale_mean) {
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int i = 0; i < n_slices; ++i) s += partials[i * 4 + threadIdx.x];
    out[out_off + threadIdx.x] = s * (threadIdx.x == 0 ? scale_loss : scale_mean);
  }
}

// out[out_off] = scale * sum of x[0..n)  (one workgroup; diagnostics only)
__global__ void k_sum(const float* x, int n, float* out, int out_off, float scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[out_off] = scale * (((red[0] + red[1]) + red[2]) + red[3]);
}

hipError_t launch_sum(const float* x, int n, float* out, int out_off, float scale, hipStream_t st) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, st, x, n, out, out_off, scale);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// TQC target (tqc.py:129-145): per row gather n_nets*Q quantiles of the target
// critics, ascending bitonic sort in LDS (one wave per row, 128-slot network),
// drop the top `drop`, target[b, s] = r + (1-d) gamma (z_sorted[s] - alpha logp').
// z layout: [n_nets][B][ldz];  target: [B][M], M = n_nets*Q - drop.
// ---------------------------------------------------------------------------
constexpr int kTqcWaves = 4;
__global__ __launch_bounds__(64 * kTqcWaves) void k_tqc_target(const float* z, long net_stride, int ldz,
                                                         int n_nets, int Q, int drop,
                                                         const float* r, const float* d,
                                                         const float* logp, const double* log_alpha,
                                                         float gamma, int B, float* target) {
  __shared__ float buf[kTqcWaves][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * kTqcWaves + wave;
  const int total = n_nets * Q, M = total - drop;
  float* sb = buf[wave];
  if (row < B) {
    for (int e = lane; e < 128; e += 64) {
      float v = __builtin_huge_valf();
      if (e < total) {
        const int n = e / Q, q = e - n * Q;
        v = z[n * net_stride + (size_t)row * ldz + q];
      }
      sb[e] = v;
    }
  }
  __syncthreads();
  for (int k = 2; k <= 128; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (row < B) {
        const int e = ((lane & ~(j - 1)) << 1) | (lane & (j - 1));  // lower index of the pair
        const int p = e | j;
        const bool up = (e & k) == 0;
        const float a = sb[e], b = sb[p];
        if ((a > b) == up) { sb[e] = b; sb[p] = a; }
      }
      __syncthreads();
    }
  }
  if (row < B) {
    const float alpha = (float)exp(*log_alpha);
    const float al = alpha * logp[row];
    const float coef = (1.f - d[row]) * gamma;
    const float rr = r[row];
    for (int s = lane; s < M; s += 64) target[(size_t)row * M + s] = rr + coef * (sb[s] - al);
  }
}

// the device N(0,1) stream exactly as the update kernels draw it: out[row][col] = philox_normal(seed, ctr, row, col)
__global__ void k_debug_normal(unsigned long long seed, unsigned long long ctr, int rows, int cols, float* out) {
  const long n = (long)rows * cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    out[e] = philox_normal(seed, ctr, (unsigned)(e / cols), (unsigned)(e % cols));
}

hipError_t launch_debug_normal(unsigned long long seed, unsigned long long ctr, int rows, int cols, float* out,
                               hipStream_t st) {
  const long n = (long)rows * cols;
  const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_debug_normal, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, seed, ctr, rows, cols, out);
  return hipGetLastError();
}

// master (row-major [N][K]) -> fragment-order packs; one 256-thread block per 256
// consecutive elements of a layer.  Pad positions are never written (the pack
// buffers are zeroed once at allocation).
__global__ void k_repack(const RepackItem* items, int n_items) {
  int it = 0;
  while (it + 1 < n_items && (int)blockIdx.x >= items[it].blk_end) ++it;
  const RepackItem I = items[it];
  const long e = (long)(blockIdx.x - I.blk_begin) * 256 + threadIdx.x;
  if (e >= (long)I.N * I.K) return;
  const int n = (int)(e / I.K), k = (int)(e - (long)n * I.K);
  const float w = I.w[e];
  if (I.pf != nullptr) I.pf[pack_index(n, k, cdiv(I.K, 16))] = w;
  if (I.pb != nullptr) I.pb[pack_index(k, n, cdiv(I.N, 16))] = w;
  if (I.x2) {   // PrecX2 packs: blocks of two fp16 planes (hi | lo), 2^8 w
    const float ws = w * PrecX2::kWScale;
    const _Float16 hi = (_Float16)ws, lo = (_Float16)(ws - (float)hi);
    if (I.pf16 != nullptr) {
      const long e16 = pack16_index(n, k, cdiv(I.K, 32));
      _Float16* d = reinterpret_cast<_Float16*>(I.pf16) + (e16 >> 9) * 1024 + (e16 & 511);
      d[0] = hi; d[512] = lo;
    }
    if (I.pb16 != nullptr) {
      const long e16 = pack16_index(k, n, cdiv(I.N, 32));
      _Float16* d = reinterpret_cast<_Float16*>(I.pb16) + (e16 >> 9) * 1024 + (e16 & 511);
      d[0] = hi; d[512] = lo;
    }
    return;
  }
  if (I.pf16 != nullptr) reinterpret_cast<__bf16*>(I.pf16)[pack16_index(n, k, cdiv(I.K, 32))] = (__bf16)w;
  if (I.pb16 != nullptr) reinterpret_cast<__bf16*>(I.pb16)[pack16_index(k, n, cdiv(I.N, 32))] = (__bf16)w;
}

// ---------------------------------------------------------------------------
// host-visible launchers
// ---------------------------------------------------------------------------
size_t mlp_slice_lds_bytes(int width, int n_layers) {
  const int nh = n_layers - 1;
  return sizeof(float) * (width == 256 ? SliceLds<256>::total(nh) : SliceLds<512>::total(nh));
}

hipError_t launch_mlp_slice(const MlpArgs& a, int width, hipStream_t st) {
  const int grid = (a.B + kR - 1) / kR;
  const size_t lds = mlp_slice_lds_bytes(width, a.net.n_layers);
  if (width == 256) {
    hipLaunchKernelGGL(k_mlp_slice<256>, dim3(grid), dim3(kThreads), lds, st, a);
  } else {
    hipLaunchKernelGGL(k_mlp_slice<512>, dim3(grid), dim3(kThreads), lds, st, a);
  }
  return hipGetLastError();
}

// n <= kMaxMulti launches of equal width, depth and batch as one
hipError_t launch_mlp_slice_multi(const MlpArgs* a, int n, int width, hipStream_t st) {
  if (n < 1 || n > kMaxMulti) return hipErrorInvalidValue;
  MlpMultiArgs m;
  for (int j = 0; j < n; ++j) m.a[j] = a[j];
  for (int j = n; j < kMaxMulti; ++j) m.a[j] = a[0];
  const dim3 grid((a[0].B + kR - 1) / kR, n);
  const size_t lds = mlp_slice_lds_bytes(width, a[0].net.n_layers);
  if (width == 256) {
    hipLaunchKernelGGL(k_mlp_slice_multi<256>, grid, dim3(kThreads), lds, st, m);
  } else {
    hipLaunchKernelGGL(k_mlp_slice_multi<512>, grid, dim3(kThreads), lds, st, m);
  }
  return hipGetLastError();
}

hipError_t init_kernel_attrs() {
  const void* ks[4] = {reinterpret_cast<const void*>(&k_mlp_slice<256>), reinterpret_cast<const void*>(&k_mlp_slice<512>),
                       reinterpret_cast<const void*>(&k_mlp_slice_multi<256>),
                       reinterpret_cast<const void*>(&k_mlp_slice_multi<512>)};
  for (const void* k : ks) {
    hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

__device__ float g_one = 1.f;

bool dw_wide_item_ok(const DwItem& it, const DwArgs& a);
hipError_t launch_dw_adam_wide(const DwItem* items, int n_items, int B, const AdamScalars& ad, hipStream_t st,
                               const DwKArgs* ride, int ride_blocks);

static const float* dw_one_dev() {
  static const float* one_dev = nullptr;
  if (one_dev == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_one)) != hipSuccess) return nullptr;
    one_dev = (const float*)p;
  }
  return one_dev;
}

// the kernel-side argument block of a (non-wide, single-rank) launch; returns the tile count or -1
// tile_k: the k extent of a tile — kDwTile (k_dw_adam and the tiles of dw_body.h), or 64 (dw_tile_x2.h: the table is
// recomputed for 16 x 64 tiles)
int fill_dw_kargs(const DwArgs& a, DwKArgs* k, int tile_k) {
  if (a.n_items < 1 || a.n_items > kDwMaxItems || a.xchg != nullptr) return -1;
  int total = 0;
  for (int j = 0; j < a.n_items; ++j) {
    k->items[j] = a.items[j];
    if (tile_k == kDwTile) {
      total += a.items[j].tile_end - a.items[j].tile_begin;
    } else {
      k->items[j].tiles_k = (a.items[j].K + tile_k - 1) / tile_k;
      k->items[j].tile_begin = total;
      total += ((a.items[j].N + kDwTileN - 1) / kDwTileN) * k->items[j].tiles_k;
      k->items[j].tile_end = total;
    }
    k->tile_end[j] = total;
  }
  for (int j = a.n_items; j < kDwMaxItems; ++j) { k->items[j] = a.items[0]; k->tile_end[j] = total; }
  if (a.skip32)
    for (int j = 0; j < kDwMaxItems; ++j) { k->items[j].pf = nullptr; k->items[j].pb = nullptr; k->items[j].tpf = nullptr; }
  k->n_items = a.n_items; k->B = a.B; k->n_part = a.n_part; k->dy_tiled = a.dy_tiled; k->ad = a.ad; k->trace = a.trace;
  k->use_row_scale = a.use_row_scale; k->one = dw_one_dev();
  k->apply_only = a.apply_only;
  k->alpha = a.alpha;                  // (a temperature step riding on the launch: the workgroup one past the tiles)
  k->gate = DwGate{};
  memset(&k->xchg, 0, sizeof k->xchg);
  return k->one != nullptr ? total : -1;
}

// (the compact argument block of a packed learner's launch: the first NI layers of a full one)
template <int NI>
static int compact_dw_kargs_n(const DwKArgs& k, DwKArgsN<NI>* o) {
  if (k.n_items > NI) return -1;
  for (int j = 0; j < NI; ++j) { o->tile_end[j] = k.tile_end[j]; o->items[j] = k.items[j]; }
  o->n_items = k.n_items; o->B = k.B; o->n_part = k.n_part; o->dy_tiled = k.dy_tiled; o->ad = k.ad; o->trace = k.trace;
  o->use_row_scale = k.use_row_scale; o->one = k.one; o->apply_only = k.apply_only;
  o->xchg = k.xchg; o->alpha = k.alpha; o->gate = k.gate;
  return k.tile_end[kDwMaxItems - 1];
}
size_t dw_group_block_bytes(int ni) { return ni == kDwGroupItems ? sizeof(DwKArgsG) : sizeof(DwKArgsG2); }
int compact_dw_kargs(const DwKArgs& k, void* o, int ni) {
  if (ni == kDwGroupItems) return compact_dw_kargs_n(k, (DwKArgsG*)o);
  if (ni == kDwGroupItems2) return compact_dw_kargs_n(k, (DwKArgsG2*)o);
  return -1;
}
hipError_t launch_dw_adam_group(const void* batch_dev, int ni, int n, int tiles, hipStream_t st) {
  const dim3 grid(tiles + 1, 1, n);
  if (ni == kDwGroupItems) hipLaunchKernelGGL(k_dw_adam_group<kDwGroupItems>, grid, dim3(kDwThreads), 0, st, (const DwKArgsG*)batch_dev);
  else if (ni == kDwGroupItems2) hipLaunchKernelGGL(k_dw_adam_group<kDwGroupItems2>, grid, dim3(kDwThreads), 0, st, (const DwKArgsG2*)batch_dev);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_dw_adam(const DwArgs& a0, hipStream_t st) {
  if (a0.n_items < 1 || a0.n_items > kDwMaxItems) return hipErrorInvalidValue;
  static const PrefetchJob no_prefetch = [] { PrefetchJob j; memset((void*)&j, 0, sizeof j); j.z0 = -1; return j; }();
  // wide layers (TQC's 512x512) go to the 64x64-tile kernel (csrc/dw_wide.hip), the rest stay here
  static const bool no_wide = [] { const char* e = getenv("OPRL_AMD_NO_RIDE"); return e != nullptr && (atoi(e) & 16) != 0; }();   // (bit 16: learner.hip)
  DwItem rest[kDwMaxItems], wide[kDwMaxItems];
  int n_rest = 0, n_wide = 0;
  for (int j = 0; j < a0.n_items; ++j) {
    if (!no_wide && n_wide < 10 && dw_wide_item_ok(a0.items[j], a0)) {
      wide[n_wide] = a0.items[j];
      // (16-bit learners: nothing of the update reads these layers' fp32 packs — a third of the launch's stores;
      // whoever does read them later rebuilds them first: fresh32, learner.hip)
      if (a0.skip32_wide && wide[n_wide].pf16 != nullptr) { wide[n_wide].pf = nullptr; wide[n_wide].pb = nullptr; wide[n_wide].tpf = nullptr; }
      ++n_wide;
    } else {
      rest[n_rest++] = a0.items[j];
    }
  }
  // the narrow layers of the same update ride on the wide launch
  const bool ride = n_wide > 0 && n_rest > 0 && a0.xchg == nullptr;
  if (n_wide > 0 && !ride) {
    hipError_t e = launch_dw_adam_wide(wide, n_wide, a0.B, a0.ad, st, nullptr, 0);
    if (e != hipSuccess) return e;
    if (n_rest == 0) return a0.alpha.log_alpha != nullptr ? hipErrorInvalidValue : hipSuccess;   // (a job needs a tile launch to ride on)
  }
  DwArgs a = a0;
  a.items = rest;
  a.n_items = n_rest;
  static const float* one_dev = nullptr;
  if (one_dev == nullptr) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_one));
    if (e != hipSuccess) return e;
    one_dev = (const float*)p;
  }
  DwKArgs k;
  int total = 0;
  for (int j = 0; j < a.n_items; ++j) {
    k.items[j] = a.items[j];
    total += a.items[j].tile_end - a.items[j].tile_begin;
    k.tile_end[j] = total;
  }
  for (int j = a.n_items; j < kDwMaxItems; ++j) { k.items[j] = a.items[0]; k.tile_end[j] = total; }
  if (a.skip32)
    for (int j = 0; j < kDwMaxItems; ++j) { k.items[j].pf = nullptr; k.items[j].pb = nullptr; k.items[j].tpf = nullptr; }
  k.n_items = a.n_items; k.B = a.B; k.n_part = a.n_part; k.dy_tiled = a.dy_tiled; k.ad = a.ad; k.trace = a.trace;
  k.use_row_scale = a.use_row_scale; k.one = one_dev;
  k.apply_only = a.apply_only;
  k.alpha = a.alpha;
  k.gate = DwGate{};
  memset(&k.xchg, 0, sizeof k.xchg);
  if (a.xchg != nullptr) {
    if (a.apply_only || n_wide > 0 || total > a.xchg->max_tiles || a.alpha.log_alpha != nullptr) return hipErrorInvalidValue;
    k.xchg = *a.xchg;
    if (a.prefetch != nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dw_adam<true>, dim3(total), dim3(kDwThreads), 0, st, k, no_prefetch);
    return hipGetLastError();
  }
  const int blocks = total + (a.alpha.log_alpha != nullptr ? 1 : 0);
  if (ride) return a.prefetch != nullptr ? hipErrorInvalidValue : launch_dw_adam_wide(wide, n_wide, a0.B, a0.ad, st, &k, blocks);
  PrefetchJob pj = no_prefetch;
  int pf_blocks = 0;
  if (a.prefetch != nullptr) {
    pj = *a.prefetch;
    pj.z0 = blocks;
    pf_blocks = (pj.B + kR - 1) / kR;
  }
  hipLaunchKernelGGL(k_dw_adam<false>, dim3(blocks + pf_blocks), dim3(kDwThreads), 0, st, k, pj);
  return hipGetLastError();
}

hipError_t launch_repack(const RepackItem* items_dev, int n_items, int total_blocks, hipStream_t st) {
  if (total_blocks <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_repack, dim3(total_blocks), dim3(256), 0, st, items_dev, n_items);
  return hipGetLastError();
}

hipError_t launch_adam_flat(float* th, float* m, float* v, float* tt, const float* g, long n,
                            const AdamScalars& ad, hipStream_t st) {
  const int grid = (int)((n + 4 * 256 - 1) / (4 * 256));
  hipLaunchKernelGGL(k_adam_flat, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, th, m, v, tt, g, n, ad);
  return hipGetLastError();
}

hipError_t launch_polyak_flat(float* tt, const float* th, long n, double tau, hipStream_t st) {
  const int grid = (int)((n + 4 * 256 - 1) / (4 * 256));
  hipLaunchKernelGGL(k_polyak_flat, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, tt, th, n, (float)tau,
                     (float)(1.0 - tau));
  return hipGetLastError();
}

hipError_t launch_alpha_step(double* log_alpha, double* m, double* v, const float* logp, int B,
                             float target_entropy, double lr, double beta1, double beta2, double eps,
                             int step, double* grad_out, const double* grad_in, float grad_scale,
                             hipStream_t st) {
  const double bc1 = 1.0 - std::pow(beta1, (double)step), bc2_sqrt = std::sqrt(1.0 - std::pow(beta2, (double)step));
  hipLaunchKernelGGL(k_alpha_step, dim3(1), dim3(256), 0, st, log_alpha, m, v, logp, B,
                     target_entropy, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_out, grad_in, grad_scale);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const float* partials, int n_slices, float* out, int out_off,
                                  float scale_loss, float scale_mean, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(64), 0, st, partials, n_slices, out, out_off,
                     scale_loss, scale_mean);
  return hipGetLastError();
}

hipError_t launch_tqc_target(const float* z, long net_stride, int ldz, int n_nets, int Q, int drop,
                             const float* r, const float* d, const float* logp,
                             const double* log_alpha, float gamma, int B, float* target,
                             hipStream_t st) {
  hipLaunchKernelGGL(k_tqc_target, dim3((B + kTqcWaves - 1) / kTqcWaves), dim3(64 * kTqcWaves), 0, st, z,
                     net_stride, ldz, n_nets, Q, drop, r, d, logp, log_alpha, gamma, B, target);
  return hipGetLastError();
}

}  // namespace oprl
