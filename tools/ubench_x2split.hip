#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float res_lo(float x, unsigned hpair) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x)); return r; }
__device__ __forceinline__ float res_hi(float x, unsigned hpair) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(x)); return r; }
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
  const f32x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  hi = __builtin_convertvector(v, f16x8);
  const u32x4 hp = __builtin_bit_cast(u32x4, hi);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[2 * i] = res_lo(v[2 * i], hp[i]); r[2 * i + 1] = res_hi(v[2 * i + 1], hp[i]); }
  lo = __builtin_convertvector(r, f16x8);
}
__global__ void k(const float* x, const float* w, float* out) {
  const int lane = threadIdx.x;
  f32x4 a = *(const f32x4*)(x + lane * 8), b = *(const f32x4*)(x + lane * 8 + 4);
  f16x8 hi, lo;
  split8(a, b, hi, lo);
  f16x8 whi = *(const f16x8*)(w + lane * 8), wlo = *(const f16x8*)(w + lane * 8 + 4);
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, whi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, wlo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, whi, acc, 0, 0, 0);
  *(f32x4*)(out + lane * 4) = acc;
}
// numeric self-test of the split product, incl. denormal lo parts
__global__ void ktest(const float* x, const float* w, float* out, int n) {
  // one wave: A rows = x vectors (16 rows x 32 k), B = w (32 k x 16 cols)
  const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
  f32x8 av, bv;
  for (int j = 0; j < 8; ++j) { av[j] = x[i * 32 + 8 * kk + j]; bv[j] = w[i * 32 + 8 * kk + j]; }
  f16x8 ah, al, bh, bl;
  f32x4 a0 = {av[0], av[1], av[2], av[3]}, a1 = {av[4], av[5], av[6], av[7]};
  f32x4 b0 = {bv[0], bv[1], bv[2], bv[3]}, b1 = {bv[4], bv[5], bv[6], bv[7]};
  split8(a0, a1, ah, al);
  split8(b0, b1, bh, bl);
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(4 * kk + r) * 16 + i] = acc[r];
}

// A [16 x 256] . B [256 x 16] tile three ways: exact-fp32 MFMA chain (the old parity mode), the split product with
// the production scales (A x 16, B x 256), and the split product with a fourth (lo lo) term
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__global__ void kdeep(const float* x, const float* w, float* out) {
  const int lane = threadIdx.x, i = lane & 15, kk = lane >> 4;
  f32x4 a32 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0}, a3b = {0, 0, 0, 0}, a4 = {0, 0, 0, 0};
  for (int s = 0; s < 64; ++s) {   // fp32: k = 4 s + kk
    a32 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i * 256 + 4 * s + kk], w[i * 256 + 4 * s + kk], a32, 0, 0, 0);
  }
  for (int s = 0; s < 8; ++s) {
    f32x4 a0, a1, b0, b1;
    for (int j = 0; j < 4; ++j) {
      a0[j] = 16.f * x[i * 256 + 32 * s + 8 * kk + j]; a1[j] = 16.f * x[i * 256 + 32 * s + 8 * kk + 4 + j];
      b0[j] = 256.f * w[i * 256 + 32 * s + 8 * kk + j]; b1[j] = 256.f * w[i * 256 + 32 * s + 8 * kk + 4 + j];
    }
    f16x8 ah, al, bh, bl;
    split8(a0, a1, ah, al);
    split8(b0, b1, bh, bl);
    f32x4& acc = (s & 1) ? a3b : a3;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, a4, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, a4, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, a4, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, a4, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) {
    out[(4 * kk + r) * 16 + i] = a32[r];
    out[256 + (4 * kk + r) * 16 + i] = (a3[r] + a3b[r]) * (1.f / 4096.f);
    out[512 + (4 * kk + r) * 16 + i] = a4[r] * (1.f / 4096.f);
  }
}
static void deep_test() {
  const int n = 16 * 256;
  static float hx[n], hw[n], ho[768];
  float *dx, *dw, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dw, n * 4); hipMalloc(&dout, 768 * 4);
  for (int trial = 0; trial < 3; ++trial) {
    srand(100 + trial);
    for (int k = 0; k < n; ++k) {
      float u = 0; for (int q = 0; q < 12; ++q) u += rand() / (float)RAND_MAX; u -= 6.f;   // ~N(0,1)
      hx[k] = trial == 1 ? (u > 0 ? u : 0.f) : u;                                         // trial 1: ReLU outputs
      hw[k] = 0.0625f * (rand() / (float)RAND_MAX * 2 - 1);
    }
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kdeep, dim3(1), dim3(64), 0, 0, dx, dw, dout);
    hipMemcpy(ho, dout, 768 * 4, hipMemcpyDeviceToHost);
    double e[3] = {0, 0, 0}, rr = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0;
      for (int k = 0; k < 256; ++k) ref += (double)hx[r * 256 + k] * hw[c * 256 + k];
      rr += ref * ref;
      for (int m = 0; m < 3; ++m) { const double d = ho[256 * m + r * 16 + c] - ref; e[m] += d * d; }
    }
    printf("K=256 trial %d: rms err / rms result: fp32 MFMA chain %.3g | split 3 terms %.3g | split 4 terms %.3g\n", trial,
           sqrt(e[0] / rr), sqrt(e[1] / rr), sqrt(e[2] / rr));
  }
}
int main() {
  const int n = 16 * 32;
  float hx[n], hw[n], ho[256];
  float *dx, *dw, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dw, n * 4); hipMalloc(&dout, 256 * 4);
  for (int scale_i = 0; scale_i < 6; ++scale_i) {
    const float sx = (float[]){1.f, 1e-2f, 1e-4f, 1e-6f, 256.f, 3e4f}[scale_i], sw = (float[]){0.06f, 0.06f, 0.06f, 0.06f, 15.f, 1.f}[scale_i];
    srand(1 + scale_i);
    for (int k = 0; k < n; ++k) { hx[k] = sx * (rand() / (float)RAND_MAX * 2 - 1); hw[k] = sw * (rand() / (float)RAND_MAX * 2 - 1); }
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(ktest, dim3(1), dim3(64), 0, 0, dx, dw, dout, n);
    hipMemcpy(ho, dout, 256 * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_abs = 0, ref_rms = 0;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) {
      double ref = 0, sabs = 0;
      for (int k = 0; k < 32; ++k) { ref += (double)hx[r * 32 + k] * hw[c * 32 + k]; sabs += fabs((double)hx[r * 32 + k] * hw[c * 32 + k]); }
      const double e = fabs(ho[r * 16 + c] - ref);
      if (e / sabs > worst) worst = e / sabs;
      ref_rms += ref * ref;
    }
    printf("x~%g w~%g: worst |err| / sum|terms| = %.3g (fp32 eps 6e-8)\n", sx, sw, worst);
  }
  deep_test();
  return 0;
}
