// slice_head.h — what a slice kernel does between the passes of an MLP: the output head
// (tanh / TD3 smoothing / tanh-Gaussian sample + log pi) and the loss-gradient seed (MSE-TD with
// twin min and entropy term, constants, SAC min-Q routing, tanh / tanh-Gaussian backward,
// quantile-Huber), shared by the single-CU kernel (k_mlp_slice) and the tensor-parallel one
// (k_mlp_slice_tp, where every member of a cluster computes the same values and only the lead
// member writes to global memory).
// Reference: algos/nn_models.py:11-64 (heads), ddpg.py:94-104, td3.py:83-132, sac.py:90-141,
// tqc.py:30-60,128-177 (seeds).
#pragma once
#include "kernels.h"
#include "philox.h"
#include "tp3.h"

namespace oprl {

__device__ __forceinline__ float logsigmoidf(float x) {
  // min(0,x) - log1p(exp(-|x|))   (ATen log_sigmoid_forward)
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

__device__ __forceinline__ float alpha_of(const SeedArgs& s) {
  return s.log_alpha != nullptr ? (float)exp(*s.log_alpha) : s.alpha_const;
}

constexpr float kLogStdMin = -20.f, kLogStdMax = 2.f;   // nn_models.py:11
constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // log(sqrt(2*pi))
constexpr float kTwoLog2 = 1.38629436111989061883f;     // 2*log(2)

// N(0,1) draw for (row, col): injected array or counter-based Philox
__device__ __forceinline__ float noise_at(const MlpArgs& A, int gr, int col) {
  if (A.noise != nullptr) return A.noise[(size_t)gr * A.action_dim + col];
  return philox_normal(A.rng_seed, A.rng_ctr, (unsigned)gr, (unsigned)col);
}


// One action dimension of the tanh-Gaussian head (nn_models.py:168-178): reparameterised sample,
// squash, and this dimension's term of log pi.
__device__ __forceinline__ float gauss_elem(float mu, float lsr, float e, float* a_out) {
  const float ls = fminf(fmaxf(lsr, kLogStdMin), kLogStdMax);
  const float sd = expf(ls);
  const float u = mu + sd * e;
  *a_out = tanhf(u);
  const float diff = u - mu;
  return -(diff * diff) / (2.f * sd * sd) - logf(sd) - kHalfLog2Pi -
         (kTwoLog2 + logsigmoidf(2.f * u) + logsigmoidf(-2.f * u));
}

// ... and its backward: d(loss)/d(mean), d(loss)/d(log_std) from da = dLoss/da (through the critics)
// and dlp = dLoss/d(log pi) = alpha / B                                        (sac.py:118-127)
__device__ __forceinline__ void gauss_elem_bwd(float mu, float lsr, float e, float da, float dlp,
                                               float* dmu, float* dls) {
  const float ls = fminf(fmaxf(lsr, kLogStdMin), kLogStdMax);
  const float sd = expf(ls);
  const float a = tanhf(mu + sd * e);
  const float du = da * (1.f - a * a) + dlp * (2.f * a);
  const bool in = (lsr >= kLogStdMin) && (lsr <= kLogStdMax);
  *dmu = du;
  *dls = in ? (du * sd * e - dlp) : 0.f;
}

// Output head: reads outS[kR][kOutLd]; global results only from the lead member.
__device__ __forceinline__ void slice_head(const MlpArgs& A, const float* outS, int Nout, int row0, bool lead) {
  const int tid = threadIdx.x, B = A.B;
  // ---- output head ------------------------------------------------------
  if (A.out_act == ACT_GAUSS) {
    const int Ad = A.action_dim;
    const int row = (tid >> 4) & (kR - 1), sub = tid & 15, gr = row0 + row;
    const bool mine = tid < kR * 16;      // 16 lanes per row, first 256 threads
    float lp = 0.f;
    if (mine && gr < B) {
      for (int col = sub; col < Ad; col += 16) {
        const float mu = outS[row * kOutLd + col];
        const float lsr = outS[row * kOutLd + Ad + col];
        float a;
        lp += gauss_elem(mu, lsr, noise_at(A, gr, col), &a);
        if (lead && A.out != nullptr) A.out[(size_t)gr * A.ldo + col] = a;
        if (lead && A.raw_out != nullptr) {
          A.raw_out[(size_t)gr * A.ldraw + col] = mu;
          A.raw_out[(size_t)gr * A.ldraw + Ad + col] = lsr;
        }
      }
    }
    lp = row16_sum(lp);
    if (lead && mine && sub == 0 && gr < B && A.logp != nullptr) A.logp[gr] = lp;
  } else if (lead && A.out != nullptr) {
    const int ncol = (A.out_act == ACT_GAUSS_MEAN) ? A.action_dim : Nout;
    for (int idx = tid; idx < kR * ncol; idx += kThreads) {
      const int row = idx / ncol, col = idx - row * ncol, gr = row0 + row;
      if (gr >= B) continue;
      float v = outS[row * kOutLd + col];
      if (A.out_act == ACT_TANH || A.out_act == ACT_GAUSS_MEAN) {
        v = tanhf(v);
      } else if (A.out_act == ACT_TANH_SMOOTH) {
        float n = noise_at(A, gr, col) * A.policy_noise;
        n = fminf(fmaxf(n, -A.noise_clip), A.noise_clip);
        v = fminf(fmaxf(tanhf(v) + n, -A.max_action), A.max_action);
      }
      A.out[(size_t)gr * A.ldo + col] = v;
    }
  }
}

// Loss-gradient seed -> auxS (zero padded), per-slice diagnostics, dY of the output layer.
// Ends with a barrier-separated, fully written auxS (the backward pass syncs before reading).
// SeedPre: the quantile-Huber seed's target samples of this thread (elements tid and tid + kThreads of the slice's
// [kR][M] block), requested by the caller long before — k_lw_head asks for them at entry (seed_pre_request below).
struct SeedPre { float v[2] = {0.f, 0.f}; bool on = false; };
__device__ __forceinline__ SeedPre seed_pre_request(const MlpArgs& A, int row0) {
  SeedPre P;
  if (A.seed_mode != SEED_QHUBER || !A.do_bwd) return P;
  const SeedArgs& S = A.seed;
  const int M = S.M;
  if (kR * M > kWaves * kR * 16 || kR * M > 2 * kThreads) return P;
  P.on = true;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = (int)threadIdx.x + k * kThreads, row = idx / M, gr = row0 + row;
    P.v[k] = (idx < kR * M && gr < A.B) ? S.p0[(size_t)gr * M + (idx - row * M)] : 0.f;
  }
  return P;
}

__device__ __forceinline__ void slice_seed(const MlpArgs& A, const float* outS, float* auxS, float* scr,
                                           int Nout, int L, int row0, int slice, bool lead, const SeedPre pre = SeedPre()) {
  const int tid = threadIdx.x, B = A.B;
  // ---- loss-gradient seed -> auxS (zero padded) -----------------------------
  lds_zero(auxS, kR * kOutLd);
  __syncthreads();
  const SeedArgs& S = A.seed;
  float p_loss = 0.f, p_q = 0.f, p_y = 0.f;
  switch (A.seed_mode) {
    case SEED_PTR:
      for (int idx = tid; idx < kR * Nout; idx += kThreads) {
        const int row = idx / Nout, col = idx - row * Nout, gr = row0 + row;
        if (gr < B) auxS[row * kOutLd + col] = S.p0[(size_t)gr * S.ld0 + col];
      }
      break;
    case SEED_MSE_TD:
      if (tid < kR) {
        const int gr = row0 + tid;
        if (gr < B) {
          const float q = outS[tid * kOutLd];
          float qn = S.p0[gr];
          if (S.p1 != nullptr) qn = fminf(qn, S.p1[gr]);
          if (S.p2 != nullptr) qn -= alpha_of(S) * S.p2[gr];
          const float y = S.r[gr] + ((1.f - S.d[gr]) * S.gamma) * qn;
          auxS[tid * kOutLd] = 2.f * (q - y) * S.cval;
          if (lead && S.y_out != nullptr) S.y_out[gr] = y;
          if (lead && S.q_out != nullptr) S.q_out[gr] = q;
          p_loss = (q - y) * (q - y);
          p_q = q;
          p_y = y;
        }
      }
      break;
    case SEED_CONST:
      for (int idx = tid; idx < kR * Nout; idx += kThreads) {
        const int row = idx / Nout, col = idx - row * Nout;
        if (row0 + row < B) auxS[row * kOutLd + col] = S.cval;
      }
      if (A.do_fwd && tid < kR && row0 + tid < B) p_q = outS[tid * kOutLd];  // diagnostic: mean q
      break;
    case SEED_MINQ:
      if (tid < kR) {
        const int gr = row0 + tid;
        if (gr < B) {
          const float q1 = S.p0[gr], q2 = S.p1[gr];
          const float w1 = q1 < q2 ? 1.f : (q1 == q2 ? 0.5f : 0.f);
          auxS[tid * kOutLd] = -(S.which == 0 ? w1 : 1.f - w1) * S.cval;
          p_q = fminf(q1, q2);
        }
      }
      break;
    case SEED_TANH:
      for (int idx = tid; idx < kR * Nout; idx += kThreads) {
        const int row = idx / Nout, col = idx - row * Nout, gr = row0 + row;
        if (gr < B) {
          const float a = S.p1[(size_t)gr * Nout + col];
          auxS[row * kOutLd + col] = S.p0[(size_t)gr * S.ld0 + col] * (1.f - a * a);
        }
      }
      break;
    case SEED_GAUSS: {
      const int Ad = Nout >> 1;
      const float alpha = alpha_of(S);
      const float dlp = alpha * S.cval;
      bool da_ok = true;
      if (S.da_flags != nullptr) {
        // da comes from workgroups of THIS launch (the riding backward, k_lw_dact): one flag per net of this slice, polled
        // by the lanes of wave 0; the rows are then read past this XCD's L2 (they were written through)
        int* okp = reinterpret_cast<int*>(scr);
        if (tid < 64) {
          bool ok = true;
          if (tid < S.n_da) {
            const unsigned long long* f = S.da_flags + (size_t)tid * S.da_fstride + slice;
            ok = false;
            for (int spin = 0; spin < S.da_spin && !ok; ++spin) {
              ok = (unsigned)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) == S.da_tag;
              if (!ok) __builtin_amdgcn_s_sleep(2);
            }
          }
          const bool all = __all(ok);
          if (tid == 0) {
            *okp = all ? 1 : 0;
            if (!all) report_expired(A.err, (KERN_LW_PAIR << 8) | SITE_LW_PAIR);
          }
        }
        __syncthreads();
        da_ok = *okp != 0;
        __syncthreads();
      }
      for (int idx = tid; idx < kR * Ad; idx += kThreads) {
        const int row = idx / Ad, col = idx - row * Ad, gr = row0 + row;
        if (gr >= B) continue;
        const float mu = S.p1[(size_t)gr * Nout + col];
        const float lsr = S.p1[(size_t)gr * Nout + Ad + col];
        const float e = noise_at(A, gr, col);   // same injected / Philox draw as the forward
        float da = 0.f;
        if (S.da_flags != nullptr) {
          for (int n = 0; n < S.n_da; ++n)
            da += __hip_atomic_load(S.p0 + n * S.da_stride + (size_t)gr * S.ld0 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!da_ok) da = __builtin_nanf("");        // a lost producer shows up as NaN
        } else
        for (int n = 0; n < S.n_da; ++n) da += S.p0[n * S.da_stride + (size_t)gr * S.ld0 + col];
        float dmu, dls;
        gauss_elem_bwd(mu, lsr, e, da, dlp, &dmu, &dls);
        auxS[row * kOutLd + col] = dmu;
        auxS[row * kOutLd + Ad + col] = dls;
      }
    } break;
    case SEED_QHUBER: {
      const int Q = S.Q, M = S.M;
      // the slice's target samples [kR][M] through LDS (scr is free until the diagnostics below):
      // read from global inside the loop they are M dependent L1 round trips per thread — 20 of the
      // 29 us of the head launch of TQC's critic step
      const bool staged = kR * M <= kWaves * kR * 16;
      if (staged && pre.on) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (tid + k * kThreads < kR * M) scr[tid + k * kThreads] = pre.v[k];
        __syncthreads();
      } else if (staged) {
        for (int idx = tid; idx < kR * M; idx += kThreads) {
          const int row = idx / M, gr = row0 + row;
          scr[idx] = gr < B ? S.p0[(size_t)gr * M + (idx - row * M)] : 0.f;
        }
        __syncthreads();
      }
      // kR * Q elements (400 for TQC) x M samples each (123 at the defaults: 5 x 25 quantiles less top_quantiles_to_drop = 2): one thread per element kept 7 of the 16 waves busy for
      // M iterations — 5 us of the critic step's head launch.  Four lanes per element, a quarter of the samples each,
      // summed over the quad in a fixed order: every wave carries the same work.
      const int n_el = kR * Q, m4 = (M + 3) >> 2;
      for (int it = 0; it * kThreads < 4 * n_el; ++it) {   // (uniform trip count: the quad shuffles below need whole quads)
        const int idx = tid + it * kThreads, el = idx >> 2, part = idx & 3;
        const int row = el / Q, q = el - row * Q, gr = row0 + row;
        const bool ok = el < n_el && gr < B;
        const float z = ok ? outS[row * kOutLd + q] : 0.f;
        const float tau = ((float)q) / (float)Q + 0.5f / (float)Q;
        const float w_pos = fabsf(tau), w_neg = fabsf(tau - 1.f);
        const float* yrow = staged ? scr + row * M : S.p0 + (size_t)gr * M;
        float g = 0.f, ls = 0.f;
        const int s0 = part * m4, s1 = ok ? min(M, s0 + m4) : s0;
        for (int s = s0; s < s1; ++s) {
          const float dl = yrow[s] - z;
          const float w = dl < 0.f ? w_neg : w_pos;
          const float hub = fminf(fmaxf(dl, -1.f), 1.f);     // Huber's derivative: clamp(dl, -1, 1)
          g += w * hub;
          // Huber itself as hub * (dl - hub / 2): dl^2 / 2 inside, |dl| - 1/2 outside — the same values, bit for bit,
          // as the two-branch form (the halvings are exact), in three operations instead of seven
          ls += w * (hub * (dl - 0.5f * hub));
        }
        g += __shfl_xor(g, 1);
        ls += __shfl_xor(ls, 1);
        g += __shfl_xor(g, 2);
        ls += __shfl_xor(ls, 2);
        if (ok && part == 0) {
          auxS[row * kOutLd + q] = -g * S.cval;
          p_loss += ls;
        }
      }
    } break;
    default: break;
  }
  if (A.partials != nullptr && lead) {  // per-slice diagnostics (block reduce through scr)
    __syncthreads();
    float v[3] = {p_loss, p_q, p_y};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) v[k] += __shfl_xor(v[k], m);
      if ((tid & 63) == 0) scr[(tid >> 6) * 4 + k] = v[k];
    }
    __syncthreads();
    if (tid < 3) {
      float sum = 0.f;
      for (int w = 0; w < kWaves; ++w) sum += scr[w * 4 + tid];
      A.partials[slice * 4 + tid] = sum;
    }
  }
  __syncthreads();
  {
    float* dlast = pick(A.dYg, L - 1);
    if (lead && dlast != nullptr) {
      if (A.done_flags != nullptr) store_rows_wt(auxS, kOutLd, dlast, A.lddo, Nout, row0, B);   // (its reader is a tile of the same launch)
      else store_rows(auxS, kOutLd, dlast, A.lddo, Nout, row0, B);
    }
  }
}

}  // namespace oprl
