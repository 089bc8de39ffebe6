// ubench_icache2 — how fast does ONE wave per CU run straight-line code of a given size, executed over and over?
// (round 5: are the chain kernel's stages instruction-FETCH bound?  k_ddpg_chain is 70-81 KB of code and every update
// walks through most of it once per workgroup.)  Kernel<N>: a loop whose body is N KB of independent v_fma (8 bytes
// each), run `iters` times by `waves` waves per workgroup, one workgroup per CU; prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int KB>
__global__ __launch_bounds__(1024) void body(float* out, int iters, long long* cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 1.0001f, c = 0.5f;
  long long t0 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == 1) t0 = __builtin_readcyclecounter();       // (the first pass warms whatever can be warmed)
#pragma unroll
    for (int k = 0; k < KB * 1024 / 64; ++k) {             // 8 v_fma_f32 (VOP3, 8 bytes) = 64 bytes
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KB>
void run(int waves, int iters, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 1024);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(body<KB>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  long long* h = (long long*)malloc(sizeof(long long) * blocks);
  hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; long long mx = 0;
  for (int b = 0; b < blocks; ++b) { mean += (double)h[b]; if (h[b] > mx) mx = h[b]; }
  mean /= blocks;
  const double n_inst = (double)(iters - 1) * KB * 1024 / 8;
  printf("code %4d KB  waves/WG %2d  blocks %3d: %.2f cycles per instruction (mean over blocks), %.2f worst block\n", KB, waves, blocks,
         mean / n_inst, (double)mx / n_inst);
  hipFree(out); hipFree(cyc); free(h);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  for (int waves : {1, 4, 16}) {
    run<8>(waves, 20, blocks); run<16>(waves, 20, blocks); run<32>(waves, 20, blocks); run<48>(waves, 20, blocks);
    run<64>(waves, 20, blocks); run<80>(waves, 20, blocks); run<96>(waves, 20, blocks); run<128>(waves, 20, blocks);
  }
  return 0;
}
