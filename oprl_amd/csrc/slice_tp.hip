// slice_tp.hip — k_mlp_slice on tensor-parallel clusters.
//
// The generic per-net launches (what SAC, TD3's fallback, the data-parallel phases of those and
// the stand-alone oprl_mlp_forward / backward use) carried one 16-row slice per CU through a whole
// MLP with the generic engine (engine.h: ~3000 instructions per pass around 28 executed MFMAs per
// wave — instruction-issue bound, profiles/r01d_stage_stamps.txt).  For the common shape — three
// layers, width 256, fan-in <= 96, <= 48 outputs — and ceil(B/16) * 4 <= CUs (B <= 1024 on
// MI355X) this kernel runs the same launch on clusters of 4 CUs per slice with the lean passes
// of tp4.h; inputs, head, loss-gradient seed and outputs are those of k_mlp_slice (slice_head.h),
// computed identically by every member, written by member 0.  The first layer's dz goes out as
// four partial buffers, summed by k_dw_adam on load (DwArgs::n_part = 4).
#include "slice_tp_body.h"

namespace oprl {

__global__ __launch_bounds__(kThreads) void k_mlp_slice_tp(const MlpArgs A) { slice_tp_body(A, blockIdx.x, blockIdx.y); }

// Two nets on the same slices in one launch (twin critics: target pair forward, online pair
// forward + backward): one boundary and one start-up instead of two.  Two by-value argument
// structs and two copies of the body — a runtime-selected struct would leave the kernel-argument
// registers.
__global__ __launch_bounds__(kThreads) void k_mlp_slice_tp2(const MlpArgs A0, const MlpArgs A1) {
  slice_tp_body(A0, blockIdx.x, blockIdx.y);
  __syncthreads();
  slice_tp_body(A1, blockIdx.x, blockIdx.y);
}

// The same pair side by side (grid.z = net) when both fit on the chip at once: 2 x slices x 4
// workgroups <= CUs (B <= 512 on MI355X).  Each net has its own exchange area.
__global__ __launch_bounds__(kThreads) void k_mlp_slice_tp2z(const MlpArgs A0, const MlpArgs A1) {
  if (blockIdx.z == 0) slice_tp_body(A0, blockIdx.x, blockIdx.y);
  else slice_tp_body(A1, blockIdx.x, blockIdx.y);
}

bool mlp_slice_tp_shape_ok(const MlpArgs& a, int width) {
  return a.net.n_layers == 3 && a.net.dims[1] == 256 && a.net.dims[2] == 256 &&
         tp4_shape_ok(width, a.net.dims[0], a.net.dims[3]) && a.dact_cols <= kNarrowMax &&
         (a.dact_cols <= 0 || ((a.dact_col0 + a.dact_cols - 1) >> 4) - (a.dact_col0 >> 4) < 4);
}

hipError_t init_slice_tp_attrs() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_slice_tp),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_slice_tp2),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_slice_tp2z),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t launch_mlp_slice_tp(const MlpArgs& a, hipStream_t st) {
  const int slices = (a.B + kR - 1) / kR;
  const size_t lds = sizeof(float) * SliceLds<256>::total(2);
  hipLaunchKernelGGL(k_mlp_slice_tp, dim3(slices, 4), dim3(kThreads), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_mlp_slice_tp2(const MlpArgs& a0, const MlpArgs& a1, int n_cus, hipStream_t st) {
  const int slices = (a0.B + kR - 1) / kR;
  const size_t lds = sizeof(float) * SliceLds<256>::total(2);
  if (2 * slices * 4 <= n_cus) {        // side by side; the second net's exchanges in their own area
    MlpArgs b1 = a1;
    b1.tp_xbuf = a1.tp_xbuf + (size_t)slices * kTpStages * 4 * kTpBlk;
    hipLaunchKernelGGL(k_mlp_slice_tp2z, dim3(slices, 4, 2), dim3(kThreads), lds, st, a0, b1);
  } else {
    hipLaunchKernelGGL(k_mlp_slice_tp2, dim3(slices, 4), dim3(kThreads), lds, st, a0, a1);
  }
  return hipGetLastError();
}

}  // namespace oprl
