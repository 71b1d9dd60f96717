// kernels_tiled.cpp -- instantiates the mixed-length column-tile passes (kernels_tiled.h): one kernel per pass length
// L = 2^x * 3^y, 64 <= L <= 512.  Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_tiled.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, uint32_t L> static TiledKernel make_tiled() {
  using C = TiledCfg<T, L>;
  TiledKernel k;
  k.fn = &tiled_mixed_kernel_ct<T, L>;
  k.L = L; k.cols = C::COLS; k.threads = C::NT; k.smem = C::SMEM;
  return k;
}

TiledKernel get_tiled_kernel(Real<TUReal>, uint32_t L) {
  typedef TUReal T;
  switch (L) {
#define FOURIER_TILED(LL) case LL: return make_tiled<T, LL>();
    FOURIER_TILED(64) FOURIER_TILED(72) FOURIER_TILED(81) FOURIER_TILED(96) FOURIER_TILED(108) FOURIER_TILED(128)
    FOURIER_TILED(144) FOURIER_TILED(162) FOURIER_TILED(192) FOURIER_TILED(216) FOURIER_TILED(243) FOURIER_TILED(256)
    FOURIER_TILED(288) FOURIER_TILED(324) FOURIER_TILED(384) FOURIER_TILED(432) FOURIER_TILED(486) FOURIER_TILED(512)
#undef FOURIER_TILED
    default: return TiledKernel();
  }
}

}  // namespace fourier_hip
