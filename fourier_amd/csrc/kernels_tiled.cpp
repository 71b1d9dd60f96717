// kernels_tiled.cpp -- instantiates the mixed-length column-tile passes (kernels_tiled.h): one kernel per pass length L,
// 64 <= L <= 512.  Shard 0: L = 2^x * 3^y (the reference's own radices, autosort/mod.rs:20-21); shards 1 .. 3 (round 5): every
// other L whose prime factors stop at 7, so that lengths like 10^5 = 400 x 250, 44100 = 210 x 210, 48000, 96000 and 10^6 =
// 100 x 100 x 100 take two or three tile passes from plain fourier_create_* (they ran Bluestein unless the plan option "specialise"
// compiled these kernels at run time).  Compiled once per precision and shard: -DFOURIER_TU_REAL=float / double
// -DFOURIER_TILED_SHARD=i (fourier_amd/build.py, packaging/CMakeLists.txt).
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

#define FOURIER_TILED(LL) case LL: return make_tiled<T, LL>();
#if FOURIER_TILED_SHARD == 0
TiledKernel get_tiled_kernel_s0(Real<TUReal>, uint32_t L) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(64) FOURIER_TILED(72) FOURIER_TILED(81) FOURIER_TILED(96) FOURIER_TILED(108) FOURIER_TILED(128)
    FOURIER_TILED(144) FOURIER_TILED(162) FOURIER_TILED(192) FOURIER_TILED(216) FOURIER_TILED(243) FOURIER_TILED(256)
    FOURIER_TILED(288) FOURIER_TILED(324) FOURIER_TILED(384) FOURIER_TILED(432) FOURIER_TILED(486) FOURIER_TILED(512)
    default: return TiledKernel();
  }
}
// the registry entry of the plan layer: the shard that holds length L
TiledKernel get_tiled_kernel(Real<TUReal>, uint32_t L) {
  for (TiledKernel k : {get_tiled_kernel_s0(Real<TUReal>{}, L), get_tiled_kernel_s1(Real<TUReal>{}, L), get_tiled_kernel_s2(Real<TUReal>{}, L),
                        get_tiled_kernel_s3(Real<TUReal>{}, L)})
    if (k.fn) return k;
  return TiledKernel();
}
#elif FOURIER_TILED_SHARD == 1
TiledKernel get_tiled_kernel_s1(Real<TUReal>, uint32_t L) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(70) FOURIER_TILED(75) FOURIER_TILED(80) FOURIER_TILED(84) FOURIER_TILED(90) FOURIER_TILED(98) FOURIER_TILED(100)
    FOURIER_TILED(105) FOURIER_TILED(112) FOURIER_TILED(120) FOURIER_TILED(125) FOURIER_TILED(126) FOURIER_TILED(135) FOURIER_TILED(140)
    FOURIER_TILED(147) FOURIER_TILED(150) FOURIER_TILED(160) FOURIER_TILED(168) FOURIER_TILED(175) FOURIER_TILED(180) FOURIER_TILED(189)
    FOURIER_TILED(196) FOURIER_TILED(200) FOURIER_TILED(210) FOURIER_TILED(224) FOURIER_TILED(225)
    default: return TiledKernel();
  }
}
#elif FOURIER_TILED_SHARD == 2
TiledKernel get_tiled_kernel_s2(Real<TUReal>, uint32_t L) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(240) FOURIER_TILED(245) FOURIER_TILED(250) FOURIER_TILED(252) FOURIER_TILED(270) FOURIER_TILED(280) FOURIER_TILED(294)
    FOURIER_TILED(300) FOURIER_TILED(315) FOURIER_TILED(320) FOURIER_TILED(336) FOURIER_TILED(343) FOURIER_TILED(350) FOURIER_TILED(360)
    default: return TiledKernel();
  }
}
#else
TiledKernel get_tiled_kernel_s3(Real<TUReal>, uint32_t L) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(375) FOURIER_TILED(378) FOURIER_TILED(392) FOURIER_TILED(400) FOURIER_TILED(405) FOURIER_TILED(420) FOURIER_TILED(441)
    FOURIER_TILED(448) FOURIER_TILED(450) FOURIER_TILED(480) FOURIER_TILED(490) FOURIER_TILED(500) FOURIER_TILED(504)
    default: return TiledKernel();
  }
}
#endif
#undef FOURIER_TILED

}  // namespace fourier_hip
