// slice_tp_body.h — the workgroup code of k_mlp_slice_tp (slice_tp.hip): member `member` of the 4-CU cluster
// that carries 16-row slice `slice` of one net's launch.  A header because the same workgroups also ride on other
// launches (layerwise.hip: TQC's actor forward beside the critics' head launch).  Uses the kernel's dynamic LDS from
// its start: SliceLds<256>::total(2) floats.
#pragma once
#include "slice_head.h"
#include "tp4.h"

namespace oprl {

__device__ __forceinline__ void slice_tp_body(const MlpArgs& A, int slice, int member) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using LY = SliceLds<256>;
  constexpr int WL = lds_ld(256);
  constexpr int L = 3;
  float* x0s = smem;
  float* h1 = smem + LY::h_off;
  float* h2 = h1 + LY::hbuf;
  float* outS = smem + LY::out_off(2);
  float* auxS = smem + LY::aux_off(2);
  float* scr = smem + LY::scr_off(2);
  const int row0 = slice * kR, B = A.B;
  Tp tp{member, 4, A.tp_xbuf + (size_t)slice * kTpStages * 4 * kTpBlk, A.tp_tag, 0, A.err, KERN_SLICE_TP << 8, kTpSpin};
  const bool lead = tp.c == 0;
  const int Nout = A.net.dims[3];

  if (A.do_fwd) {
    lds_zero(x0s, kR * kX0Ld);
    __syncthreads();
    load_rows(x0s, kX0Ld, 0, A.x0, A.k0, A.k0, row0, B);
    if (A.x1 != nullptr) load_rows(x0s, kX0Ld, A.k0, A.x1, A.k1, A.k1, row0, B);
    const Tp3Store st{A.Xg[1], A.Xg[2], nullptr, nullptr, 0};
    tp4_forward(A.net, x0s, h1, h2, outS, scr, tp, st, row0, B);
    if (lead && A.Xg[0] != nullptr) store_rows(x0s, kX0Ld, A.Xg[0], A.ldx0, A.net.dims[0], row0, B);
    slice_head(A, outS, Nout, row0, lead);
  } else if (A.do_bwd) {
    load_rows4(h1, WL, A.Xg[1], 256, 256, row0, B);
    load_rows4(h2, WL, A.Xg[2], 256, 256, row0, B);
  }
  if (!A.do_bwd) return;

  const Tp3Store sb{nullptr, nullptr, A.dYg[1], A.dYg[0], A.dY0_stride, B, A.done_flags != nullptr};   // tile-major dz1 partials (DwArgs::dy_tiled); written through for tiles of the same launch
  if (A.seed.da_flags != nullptr) {
    // riding on the launch that produces the seed's input: the backward's fragments are requested BEFORE the wait for it
    tp4_backward(A.net, auxS, h1, h2, scr, tp, sb, row0, B, A.dact_col0, A.dact_cols, auxS, NoStamp(),
                 [&]() { slice_seed(A, outS, auxS, scr, Nout, L, row0, slice, lead); });
  } else {
    slice_seed(A, outS, auxS, scr, Nout, L, row0, slice, lead);
    tp4_backward(A.net, auxS, h1, h2, scr, tp, sb, row0, B, A.dact_col0, A.dact_cols, auxS);
  }
  if (lead && A.dact_cols > 0 && A.dact != nullptr)
    store_rows(auxS, kOutLd, A.dact, A.lddact, A.dact_cols, row0, B);
  if (A.done_flags != nullptr) {       // every wave's rows are out before the member says so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_store(A.done_flags + 4 * slice + member, (unsigned long long)A.done_tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace oprl
