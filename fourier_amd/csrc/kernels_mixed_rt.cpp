// kernels_mixed_rt.cpp -- instantiates the runtime-parameterised LDS mixed-radix kernel (kernels_mixed.h): MAXP in {3, 7, 13} x five launch shapes.
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_mixed.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

// (points per thread, threads): 4 x 256 and 8 x 128 up to 1024 points per workgroup, 8 x 256 up to 2048, 8 x 512 up to 4096,
// 8 x 1024 up to 8192 (MixedEngine::pick_kernel).  f64 with a radix-13 butterfly does not fit 128 registers above 2048
// points (spills; 4095 f64: 20 % of the HBM peak against Bluestein's 25 %): not instantiated, the caller falls back.
MixKernelFn get_mixed_rt_kernel(Real<TUReal>, int maxp, int ppt, int nt) {
  typedef TUReal T;
#define FOURIER_MIX_RT(P, NT) (maxp == 13 ? &mixed_radix_kernel<T, 13, P, NT> : (maxp == 7 ? &mixed_radix_kernel<T, 7, P, NT> : &mixed_radix_kernel<T, 3, P, NT>))
#define FOURIER_MIX_RT_7(P, NT) (maxp == 7 ? &mixed_radix_kernel<T, 7, P, NT> : &mixed_radix_kernel<T, 3, P, NT>)
  if (ppt == 8 && nt == 128) return FOURIER_MIX_RT(8, 128);
  if (ppt == 8 && nt == 256) return FOURIER_MIX_RT(8, 256);
  if (maxp == 13) {
    // f64: 4 x 256 spills 36 B/lane, 8 x 1024 spills 132 B/lane, 8 x 512 holds two waves per SIMD (round 3's resource
    // table): none of the three is instantiated; 4 x 256 callers use 8 x 256, longer transforms take Bluestein
    if constexpr (sizeof(T) == 4) {
      if (ppt == 4 && nt == 256) return &mixed_radix_kernel<T, 13, 4, 256>;
      if (ppt == 8 && nt == 512) return &mixed_radix_kernel<T, 13, 8, 512>;
      if (ppt == 8 && nt == 1024) return &mixed_radix_kernel<T, 13, 8, 1024>;
    }
    return nullptr;
  }
  if (ppt == 4 && nt == 256) return FOURIER_MIX_RT_7(4, 256);
  if (ppt == 8 && nt == 512) return FOURIER_MIX_RT_7(8, 512);
  if (ppt == 8 && nt == 1024) return FOURIER_MIX_RT_7(8, 1024);
#undef FOURIER_MIX_RT
#undef FOURIER_MIX_RT_7
  return nullptr;
}

}  // namespace fourier_hip
