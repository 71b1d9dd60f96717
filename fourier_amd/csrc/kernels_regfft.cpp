// kernels_regfft.cpp -- instantiates the per-length register-stage transforms (kernels_regfft.h: regfft_kernel<T, R1, R2>,
// regfft3_kernel<T, R1, R2, R3>) for the lengths of regfft_shapes.h (generated: tools/gen_regfft_shapes.py; a length is listed in a
// precision where the A/B against the route it had says so).  Compiled FOURIER_REGFFT_SHARDS times per precision
// (-DFOURIER_TU_REAL=float / double -DFOURIER_REGFFT_SHARD=i): the rows are dealt round-robin over the shards (fourier_amd/build.py,
// packaging/CMakeLists.txt).
#include "engine_common.h"
#include "kernels_regfft.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, uint32_t R1, uint32_t R2, uint32_t R3> static ChirpzKernel make_regfft() {
  ChirpzKernel k;
  if constexpr (R3 == 0) {
    using C = ChirpzRegCfg<T, R1, R2>;
    k.fn = &regfft_kernel<T, R1, R2>;
    k.m = C::M; k.r1 = R1; k.r2 = R2; k.tpw = C::TPW; k.smem = C::SMEM;
  } else if constexpr (Chirpz3Cfg<T, R1, R2, R3>::SMEM <= (size_t)160 * 1024 && Chirpz3Cfg<T, R1, R2, R3>::NT <= 1024) {
    using C = Chirpz3Cfg<T, R1, R2, R3>;  // (a transform -- in f32 a pair -- with its padding within a compute unit's LDS, a stage within 1024 lanes)
    k.fn = &regfft3_kernel<T, R1, R2, R3>;
    k.m = C::M; k.r1 = R1; k.r2 = R2; k.r3 = R3; k.tpw = C::NV; k.threads = C::NT; k.smem = C::SMEM;
  }
  return k;
}

enum { REGFFT_COUNTER_BASE = __COUNTER__ };
#ifdef FOURIER_EMU  // the CPU emulation build keeps the lengths its test names (compile time)
#define FOURIER_REGFFT_BUILT(EMU) (EMU)
#else
#define FOURIER_REGFFT_BUILT(EMU) 1
#endif
#define FOURIER_REGFFT_ROW(NN, A, B, C, F32, F64, EMU) FOURIER_REGFFT_ROW_I(NN, A, B, C, F32, F64, EMU, (__COUNTER__ - REGFFT_COUNTER_BASE - 1))
#define FOURIER_REGFFT_ROW_I(NN, A, B, C, F32, F64, EMU, IDX)                                                          \
  case NN:                                                                                                             \
    if constexpr ((IDX) % FOURIER_REGFFT_SHARDS == FOURIER_REGFFT_SHARD && (sizeof(T) == 4 ? (F32) : (F64)) && FOURIER_REGFFT_BUILT(EMU)) \
      return make_regfft<T, A, B, C>();                                                                                \
    return ChirpzKernel();

template <typename T> static ChirpzKernel lookup(uint32_t n) {
  switch (n) {
#include "regfft_shapes.h"
    default: return ChirpzKernel();
  }
}
#undef FOURIER_REGFFT_ROW
#undef FOURIER_REGFFT_ROW_I

#define FOURIER_REGFFT_SHARD_FN_(I) get_regfft_kernel_s##I
#define FOURIER_REGFFT_SHARD_FN(I) FOURIER_REGFFT_SHARD_FN_(I)
ChirpzKernel FOURIER_REGFFT_SHARD_FN(FOURIER_REGFFT_SHARD)(Real<TUReal>, uint32_t n) { return lookup<TUReal>(n); }

#if FOURIER_REGFFT_SHARD == 0
ChirpzKernel get_regfft_kernel(Real<TUReal>, uint32_t n) {
#define FOURIER_REGFFT_TRY(I, T) if (ChirpzKernel k = get_regfft_kernel_s##I(Real<T>{}, n); k.fn) return k;
  FOURIER_REGFFT_SHARD_LIST(FOURIER_REGFFT_TRY, TUReal)
#undef FOURIER_REGFFT_TRY
  return ChirpzKernel();
}
#endif

}  // namespace fourier_hip
