// Microbenchmark: how fast can ONE workgroup (one CU) stream a 256 KB fp32 weight
// matrix that another kernel just rewrote?  Decides the slice-kernel design.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_touch(float* w, long n, float v) {  // "Adam": rewrite the weights
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) w[i] = w[i] * 0.5f + v;
}

// each WG streams `bytes` starting at w (+ wg_stride per WG), `passes` times; PATTERN 0: 1 KB
// contiguous per wave-instruction; 1: 16 rows x 64 B (MFMA B-fragment shape, row = 1 KB)
template <int PATTERN, int UNROLL>
__global__ void k_stream(const float* __restrict__ w, long wg_stride_f, long bytes, int passes, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float* base = w + (long)blockIdx.x * wg_stride_f;
  const long nf = bytes / 4;
  f32x4 acc = {0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {
    // wave `wave` handles chunks wave, wave+nw, ... of 1 KB (256 floats) each
    for (long c0 = wave; c0 * 256 < nf; c0 += (long)nw * UNROLL) {
      f32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        long c = c0 + (long)u * nw;
        long off;
        if (PATTERN == 0) off = c * 256 + lane * 4;
        else if (PATTERN == 2) {
          // what gemm_packed's wide path does: wave w streams ITS tile's 16 macro steps (16 KB, consecutive 1 KB blocks),
          // the waves of a workgroup are 16 KB apart; tiles w, w + nw, ... (c = wave + k * nw -> tile = c % nw + nw * (k / 16), step = k % 16)
          const int lg = 31 - __builtin_clz(nw);                     // (nw is a power of two: no 64-bit divisions in the address)
          const int ci = (int)c, k = ci >> lg, wv = ci & (nw - 1);
          off = (long)((((k >> 4) << lg) + wv) * 4096 + (k & 15) * 256 + lane * 4);
        }
        else if (PATTERN == 5 || PATTERN == 6) {
          // tile-per-wave, a wave's consecutive requests 4 KB (5) / 2 KB (6) apart inside its tile: steps 0, 4, 8, 12, 1, 5, ...
          const int lg = 31 - __builtin_clz(nw);
          const int ci = (int)c, k = ci >> lg, wv = ci & (nw - 1), kk = k & 15;
          const int st = PATTERN == 5 ? (kk & 3) * 4 + (kk >> 2) : (kk & 7) * 2 + (kk >> 3);
          off = (long)((((k >> 4) << lg) + wv) * 4096 + st * 256 + lane * 4);
        }
        else if (PATTERN == 3 || PATTERN == 4) {
          // tile-per-wave with a per-wave skew of the step order: wave w starts at step w (3) / 4 w (4) of its tile
          const int lg = 31 - __builtin_clz(nw);
          const int ci = (int)c, k = ci >> lg, wv = ci & (nw - 1);
          const int st = ((k & 15) + (PATTERN == 3 ? wv : 4 * wv)) & 15;
          off = (long)((((k >> 4) << lg) + wv) * 4096 + st * 256 + lane * 4);
        }
        else { long blk = c / 16, sub = c % 16; off = blk * 4096 + (lane & 15) * 256 + sub * 16 + (lane >> 4) * 4; }
        v[u] = (c * 256 < nf) ? *reinterpret_cast<const f32x4*>(base + off) : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (p == 0 && threadIdx.x == 0) cyc[blockIdx.x * 4 + 1] = __builtin_readcyclecounter() - t0;
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x * 4] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

template <int PATTERN, int UNROLL>
void run(const char* name, float* w, long total_f, int n_wg, int threads, long bytes, bool shared, int passes, float* out, long long* cyc) {
  std::vector<long long> h(n_wg * 4);
  double best = 1e30, best1 = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, w, total_f, 0.001f * rep);
    hipLaunchKernelGGL((k_stream<PATTERN, UNROLL>), dim3(n_wg), dim3(threads), 0, 0, w, shared ? 0 : bytes / 4, bytes, passes, out, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, sizeof(long long) * n_wg * 4, hipMemcpyDeviceToHost);
    long long mx = 0, mx1 = 0;
    for (int i = 0; i < n_wg; ++i) { if (h[i * 4] > mx) mx = h[i * 4]; if (h[i * 4 + 1] > mx1) mx1 = h[i * 4 + 1]; }
    if (mx < best) best = mx;
    if (mx1 < best1) best1 = mx1;
  }
  double later = passes > 1 ? (best - best1) / (passes - 1) : 0;
  printf("%-34s wgs=%3d thr=%4d %s KB=%4ld  pass1: %7.0f cyc = %5.1f B/clk/CU", name, n_wg, threads, shared ? "shared " : "private", bytes / 1024, best1, bytes / best1);
  if (passes > 1) printf("   re-read: %7.0f cyc = %5.1f B/clk/CU", later, bytes / later);
  printf("\n");
}

int main() {
  const long total_f = 64l << 20;  // 256 MB
  float *w, *out; long long* cyc;
  hipMalloc(&w, total_f * 4); hipMalloc(&out, 64); hipMalloc(&cyc, sizeof(long long) * 4096);
  hipMemset(w, 0, total_f * 4);
  const long KB256 = 256 * 1024;
  for (int thr : {256, 512, 1024}) {
    run<2, 8>("tile-per-wave u8 ", w, total_f, 16, thr, KB256, true, 3, out, cyc);
    run<2, 8>("tile-per-wave u8 ", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<5, 8>("tile-per-wave str4K", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<6, 8>("tile-per-wave str2K", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<3, 8>("tile-per-wave skew1", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<4, 8>("tile-per-wave skew4", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<2, 4>("tile-per-wave u4 ", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<2, 16>("tile-per-wave u16", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<0, 8>("contig  unroll8  ", w, total_f, 256, thr, KB256, true, 3, out, cyc);
    run<0, 8>("contig  unroll8  ", w, total_f, 16, thr, KB256, true, 3, out, cyc);
    run<1, 8>("frag16x64 unroll8", w, total_f, 16, thr, KB256, true, 3, out, cyc);
  }
  run<0, 16>("contig  unroll16 ", w, total_f, 16, 256, KB256, true, 3, out, cyc);
  run<0, 4>("contig  unroll4  ", w, total_f, 16, 1024, KB256, true, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 1, 256, KB256, true, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 1, 1024, KB256, true, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 16, 256, KB256, false, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 64, 256, KB256, true, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 256, 256, KB256, true, 3, out, cyc);
  run<0, 8>("contig  unroll8  ", w, total_f, 256, 1024, KB256, true, 3, out, cyc);
  run<0, 8>("contig 64KB      ", w, total_f, 64, 256, 64 * 1024, false, 3, out, cyc);
  run<0, 8>("contig 16KB      ", w, total_f, 256, 256, 16 * 1024, false, 3, out, cyc);
  return 0;
}
