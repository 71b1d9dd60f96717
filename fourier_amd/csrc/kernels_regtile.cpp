// kernels_regtile.cpp -- instantiates the register-resident column-tile passes of mixed length (kernels_regtile.h): one kernel per
// pass length L of kernels_tiled.cpp's menu that splits into two factors of at most 32 (every length there but 7^3, 5 * 7^2 and
// 10 * 7^2).  Compiled once per precision and shard: -DFOURIER_TU_REAL=float / double -DFOURIER_TILED_SHARD=i (fourier_amd/build.py,
// packaging/CMakeLists.txt).
#include "engine_common.h"
#include "kernels_regtile.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

// which: 0 = the plain pass, 1 / 2 / 3 = the Bluestein sweeps on a smooth M (chirp-in first pass, conv, chirp-out last pass)
template <typename T, uint32_t L> static TiledKernel make_regtile(int which) {
  if constexpr (reg_tile_shape(L, (uint32_t)sizeof(cpx<T>)).r1 != 0) {
    using C = RegTileCfg<T, L>;
    TiledKernel k;
    k.fn = which == 1 ? &tiled_reg_kernel<T, L, IO_BLU_IN> : which == 2 ? &tiled_reg_conv_kernel<T, L> : which == 3 ? &tiled_reg_kernel<T, L, IO_BLU_OUT>
                                                                                                                 : &tiled_reg_kernel<T, L, IO_PLAIN>;
    k.L = L; k.cols = C::COLS; k.threads = C::NT; k.smem = C::SMEM; k.r1 = C::R1; k.r2 = C::R2;
    return k;
  } else {
    return TiledKernel();
  }
}

// 513 ... 1024 points (64-byte row segments): the plain pass only -- two passes where three tile passes of at most 512 points were needed
template <typename T, uint32_t L> static TiledKernel make_regtile_long(int which) {
  if constexpr (reg_tile_shape(L, (uint32_t)sizeof(cpx<T>)).r1 != 0) {
    if (which != 0) return TiledKernel();
    using C = RegTileCfg<T, L>;
    TiledKernel k;
    k.fn = &tiled_reg_kernel<T, L, IO_PLAIN>;
    k.L = L; k.cols = C::COLS; k.threads = C::NT; k.smem = C::SMEM; k.r1 = C::R1; k.r2 = C::R2;
    return k;
  } else {
    (void)which;
    return TiledKernel();  // (f64: a stage of more than 32 points)
  }
}
#define FOURIER_TILED(LL) case LL: return make_regtile<T, LL>(which);
#define FOURIER_LONG(LL) case LL: return make_regtile_long<T, LL>(which);
#if FOURIER_TILED_SHARD == 0
TiledKernel get_regtile_kernel_s0(Real<TUReal>, uint32_t L, int which) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(64) FOURIER_TILED(72) FOURIER_TILED(81) FOURIER_TILED(96) FOURIER_TILED(108) FOURIER_TILED(128)
    FOURIER_TILED(144) FOURIER_TILED(162) FOURIER_TILED(192) FOURIER_TILED(216) FOURIER_TILED(243) FOURIER_TILED(256)
    FOURIER_TILED(288) FOURIER_TILED(324) FOURIER_TILED(384) FOURIER_TILED(432) FOURIER_TILED(486) FOURIER_TILED(512)
    FOURIER_LONG(525) FOURIER_LONG(540) FOURIER_LONG(560) FOURIER_LONG(567) FOURIER_LONG(576) FOURIER_LONG(588) FOURIER_LONG(600)
    default: return TiledKernel();
  }
}
TiledKernel get_regtile_kernel(Real<TUReal>, uint32_t L, int which) {
  for (TiledKernel k : {get_regtile_kernel_s0(Real<TUReal>{}, L, which), get_regtile_kernel_s1(Real<TUReal>{}, L, which),
                        get_regtile_kernel_s2(Real<TUReal>{}, L, which), get_regtile_kernel_s3(Real<TUReal>{}, L, which)})
    if (k.fn) return k;
  return TiledKernel();
}
#elif FOURIER_TILED_SHARD == 1
TiledKernel get_regtile_kernel_s1(Real<TUReal>, uint32_t L, int which) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(70) FOURIER_TILED(75) FOURIER_TILED(80) FOURIER_TILED(84) FOURIER_TILED(90) FOURIER_TILED(98) FOURIER_TILED(100)
    FOURIER_TILED(105) FOURIER_TILED(112) FOURIER_TILED(120) FOURIER_TILED(125) FOURIER_TILED(126) FOURIER_TILED(135) FOURIER_TILED(140)
    FOURIER_TILED(147) FOURIER_TILED(150) FOURIER_TILED(160) FOURIER_TILED(168) FOURIER_TILED(175) FOURIER_TILED(180) FOURIER_TILED(189)
    FOURIER_TILED(196) FOURIER_TILED(200) FOURIER_TILED(210) FOURIER_TILED(224) FOURIER_TILED(225)
    FOURIER_LONG(625) FOURIER_LONG(630) FOURIER_LONG(640) FOURIER_LONG(648) FOURIER_LONG(672) FOURIER_LONG(675) FOURIER_LONG(700)
    default: return TiledKernel();
  }
}
#elif FOURIER_TILED_SHARD == 2
TiledKernel get_regtile_kernel_s2(Real<TUReal>, uint32_t L, int which) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(240) FOURIER_TILED(245) FOURIER_TILED(250) FOURIER_TILED(252) FOURIER_TILED(270) FOURIER_TILED(280) FOURIER_TILED(294)
    FOURIER_TILED(300) FOURIER_TILED(315) FOURIER_TILED(320) FOURIER_TILED(336) FOURIER_TILED(343) FOURIER_TILED(350) FOURIER_TILED(360)
    FOURIER_LONG(720) FOURIER_LONG(729) FOURIER_LONG(750) FOURIER_LONG(756) FOURIER_LONG(768) FOURIER_LONG(784) FOURIER_LONG(800)
    default: return TiledKernel();
  }
}
#else
TiledKernel get_regtile_kernel_s3(Real<TUReal>, uint32_t L, int which) {
  typedef TUReal T;
  switch (L) {
    FOURIER_TILED(375) FOURIER_TILED(378) FOURIER_TILED(392) FOURIER_TILED(400) FOURIER_TILED(405) FOURIER_TILED(420) FOURIER_TILED(441)
    FOURIER_TILED(448) FOURIER_TILED(450) FOURIER_TILED(480) FOURIER_TILED(490) FOURIER_TILED(500) FOURIER_TILED(504)
    FOURIER_LONG(810) FOURIER_LONG(840) FOURIER_LONG(864) FOURIER_LONG(896) FOURIER_LONG(900) FOURIER_LONG(960) FOURIER_LONG(1024)
    FOURIER_LONG(875) FOURIER_LONG(945) FOURIER_LONG(972) FOURIER_LONG(980) FOURIER_LONG(1000)
    default: return TiledKernel();
  }
}
#endif
#undef FOURIER_TILED
#undef FOURIER_LONG

}  // namespace fourier_hip
