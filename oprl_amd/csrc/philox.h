// philox.h — counter-based Philox4x32-10 (Salmon et al., SC'11) for the
// device-side minibatch indices and the TD3 / tanh-Gaussian noise draws of the
// throughput path.  (Parity runs inject indices / noise instead: a GPU stream
// cannot reproduce numpy's MT19937 or torch's CPU generator — SURVEY.md §7.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oprl {

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * ctr.x, p1 = (uint64_t)M1 * ctr.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    ctr = u32x4{hi1 ^ ctr.y ^ k0, lo1, hi0 ^ ctr.w ^ k1, lo0};
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

// uniform integer in [0, n) from 32 random bits (multiply-shift)
__host__ __device__ inline uint32_t bounded_u32(uint32_t x, uint32_t n) {
  return (uint32_t)(((uint64_t)x * (uint64_t)n) >> 32);
}

// one N(0,1) draw addressed by (stream ctr, row, col): Box-Muller on two uniforms
__device__ inline float philox_normal(unsigned long long seed, unsigned long long ctr,
                                      uint32_t row, uint32_t col) {
  const u32x4 r = philox4x32_10(u32x4{(uint32_t)ctr, (uint32_t)(ctr >> 32), row, col},
                                (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
  const float u2 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace oprl
