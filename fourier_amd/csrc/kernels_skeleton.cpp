// kernels_skeleton.cpp -- the tile passes of the two-pass power-of-two plans with their arithmetic, table look-ups and LDS
// exchanges compiled out (FOURIER_ABLATE = 2): the SAME loads, the SAME stores, the same grid, tile order, cache policies,
// registers-per-thread budget and LDS reservation (hence the same two / one workgroups per CU) -- what the passes' memory
// shape streams when nothing else happens.  bench.py times these kernels (plan option "skeleton", experiments library only)
// on the workload's own buffers as `roofline.stream_ceiling_gbps`.  Results are meaningless.
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py); experiments library only.
#define FOURIER_EXPERIMENTS_TU 1  // an ablation: its templates live in the inline namespace `ablated` (kernels_common.h)
#define FOURIER_ABLATE 2
#include "engine_common.h"
#include "kernels_pass.h"
#include "tile_shapes.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

// a kernel name of its own: the product library's fft_pass_kernel<...> of the same shape is linked beside it
template <typename T, int L, int CG, int MODE>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_skeleton_kernel(PassArgs a) {
  FOURIER_DYN_SMEM(smem);
  pass_tile<T, L, CG, MODE, IO_PLAIN, PassPolicy<L, MODE, CG>::LD, PassPolicy<L, MODE, CG>::ST>(a, blockIdx.x, gridDim.x, smem, (int)threadIdx.x);
}

template <typename T, int L, int CG, int MODE> static KernelInfo make_skeleton_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_skeleton_kernel<T, L, CG, MODE>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::smem_bytes(MODE);
  return k;
}

KernelInfo get_skeleton_kernel(Real<TUReal>, int L, int mode) {
  typedef TUReal T;
  if (L == 1024 && mode == MODE_FIRST) return make_skeleton_info<T, 1024, FOURIER_CG_1024, MODE_FIRST>();
  if (L == 1024 && mode == MODE_LAST) return make_skeleton_info<T, 1024, FOURIER_CG_1024, MODE_LAST>();
  if (L == 2048 && mode == MODE_FIRST) return make_skeleton_info<T, 2048, FOURIER_CG_2048_FIRST, MODE_FIRST>();
  if (L == 2048 && mode == MODE_LAST) return make_skeleton_info<T, 2048, FOURIER_CG_2048, MODE_LAST>();
  return KernelInfo();
}

}  // namespace fourier_hip
