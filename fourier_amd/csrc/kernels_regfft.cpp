// kernels_regfft.cpp -- instantiates the per-length register-stage transforms (kernels_regfft.h: regfft_kernel<T, R1, R2>,
// regfft3_kernel<T, R1, R2, R3>) for the lengths of regfft_shapes.h (generated: tools/gen_regfft_shapes.py; a length is listed in a
// precision where the A/B against the route it had says so).  Compiled FOURIER_REGFFT_SHARDS times per precision
// (-DFOURIER_TU_REAL=float / double -DFOURIER_REGFFT_SHARD=i): the rows are dealt round-robin over the shards (fourier_amd/build.py,
// packaging/CMakeLists.txt).
#include "engine_common.h"
#include "kernels_regfft.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, uint32_t R1, uint32_t R2, uint32_t R3, bool SPLIT, bool FACT, bool PAIR = true> static ChirpzKernel make_regfft() {
  ChirpzKernel k;
  if constexpr (R3 == 0) {
    using C = ChirpzRegCfg<T, R1, R2>;
    k.fn = &regfft_kernel<T, R1, R2>;
    k.m = C::M; k.r1 = R1; k.r2 = R2; k.tpw = C::TPW; k.smem = C::SMEM;
  } else if constexpr (Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>::SMEM <= (size_t)160 * 1024 && Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>::NT <= 1024) {
    using C = Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>;  // (a transform -- in f32 a pair -- with its padding within a compute unit's LDS, a stage within 1024 lanes)
    k.fn = &regfft3_kernel<T, R1, R2, R3, SPLIT, FACT, PAIR>;
    k.m = C::M; k.r1 = R1; k.r2 = R2; k.r3 = R3; k.tpw = C::NV; k.threads = C::NT; k.smem = C::SMEM; k.split = SPLIT; k.fact = FACT;
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
// a precision's flag: 0 = not listed; 1 = listed; three stages: 2 = split-plane exchanges, 3 = factored twiddle tables, 4 = both; f32: 5 = one
// transform per workgroup (unpaired), 6 = unpaired with factored tables.  A/B builds: 9 = 1 ... 4 built, 1 the default (variant = 1 ... 4 picks one);
// 10 + F (f32) = the listed F and the unpaired 5 / 6 built (variant = 5 / 6 picks them)
#define FOURIER_REGFFT_ROW_I(NN, A, B, C, F32, F64, EMU, IDX)                                                          \
  case NN:                                                                                                             \
    if constexpr ((IDX) % FOURIER_REGFFT_SHARDS == FOURIER_REGFFT_SHARD && (sizeof(T) == 4 ? (F32) : (F64)) != 0 && FOURIER_REGFFT_BUILT(EMU)) { \
      constexpr int F0 = sizeof(T) == 4 ? (F32) : (F64), F = F0 >= 10 ? F0 - 10 : F0;                                  \
      if constexpr (F0 == 9 && (C) != 0) {                                                                             \
        return variant == 4 ? make_regfft<T, A, B, C, true, true>() : variant == 3 ? make_regfft<T, A, B, C, false, true>()  \
             : variant == 2 ? make_regfft<T, A, B, C, true, false>() : make_regfft<T, A, B, C, false, false>();        \
      } else {                                                                                                         \
        if constexpr (F0 >= 10 && (C) != 0 && sizeof(T) == 4) {                                                        \
          if (variant == 5) return make_regfft<T, A, B, C, false, false, false>();                                     \
          if (variant == 6) return make_regfft<T, A, B, C, false, true, false>();                                      \
        }                                                                                                              \
        return make_regfft<T, A, B, C, (F == 2 || F == 4) && (C) != 0, (F == 3 || F == 4 || F == 6) && (C) != 0, !((F == 5 || F == 6) && (C) != 0 && sizeof(T) == 4)>(); \
      }                                                                                                                \
    }                                                                                                                  \
    return ChirpzKernel();

// a 2^a 3^b length: by default such lengths keep the reference's own schedule (bit-identical to the CPU restatement, plan.h); the register-stage
// kernel is handed out only to a caller that asks for it (variant >= 100: plan option "register_stages")
#define FOURIER_REGFFT_OPT_ROW(NN, A, B, C, F32, F64, EMU) FOURIER_REGFFT_OPT_ROW_I(NN, A, B, C, F32, F64, EMU, (__COUNTER__ - REGFFT_COUNTER_BASE - 1))
#define FOURIER_REGFFT_OPT_ROW_I(NN, A, B, C, F32, F64, EMU, IDX)                                                      \
  case NN:                                                                                                             \
    if constexpr ((IDX) % FOURIER_REGFFT_SHARDS == FOURIER_REGFFT_SHARD && (sizeof(T) == 4 ? (F32) : (F64)) != 0 && FOURIER_REGFFT_BUILT(EMU)) { \
      constexpr int F = sizeof(T) == 4 ? (F32) : (F64);                                                                \
      if (variant < 100) return ChirpzKernel();                                                                        \
      return make_regfft<T, A, B, C, (F == 2 || F == 4) && (C) != 0, (F == 3 || F == 4 || F == 6) && (C) != 0, !((F == 5 || F == 6) && (C) != 0 && sizeof(T) == 4)>(); \
    }                                                                                                                  \
    return ChirpzKernel();

template <typename T> static ChirpzKernel lookup(uint32_t n, int variant) {
  switch (n) {
#include "regfft_shapes.h"
    default: return ChirpzKernel();
  }
}
#undef FOURIER_REGFFT_ROW
#undef FOURIER_REGFFT_ROW_I
#undef FOURIER_REGFFT_OPT_ROW
#undef FOURIER_REGFFT_OPT_ROW_I

#define FOURIER_REGFFT_SHARD_FN_(I) get_regfft_kernel_s##I
#define FOURIER_REGFFT_SHARD_FN(I) FOURIER_REGFFT_SHARD_FN_(I)
ChirpzKernel FOURIER_REGFFT_SHARD_FN(FOURIER_REGFFT_SHARD)(Real<TUReal>, uint32_t n, int variant) { return lookup<TUReal>(n, variant); }

#if FOURIER_REGFFT_SHARD == 0
ChirpzKernel get_regfft_kernel(Real<TUReal>, uint32_t n, int variant) {
#define FOURIER_REGFFT_TRY(I, T) if (ChirpzKernel k = get_regfft_kernel_s##I(Real<T>{}, n, variant); k.fn) return k;
  FOURIER_REGFFT_SHARD_LIST(FOURIER_REGFFT_TRY, TUReal)
#undef FOURIER_REGFFT_TRY
  return ChirpzKernel();
}
#endif

}  // namespace fourier_hip
