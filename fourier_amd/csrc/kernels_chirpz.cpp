// kernels_chirpz.cpp -- instantiates the one-launch chirp-z kernels on a smooth M = R1 x R2 (kernels_chirpz.h): one kernel per M of the menu
// below (steps of 4 ... 25 %; R1 >= R2, both with prime factors up to 7, R1 <= 32 so that two to ten lane groups share a wave; f32 also 36 x 32,
// one lane group per wave: 36 x 36, 40 x 36 and 40 x 40 are level with the power-of-two kernels of M = 2048, profiles/r06_s45_chirpz_reg_ab.jsonl);
// and eight M = R1 x R2 x R3 of 1296 ... 3072 and 8820 / 9261 points (a workgroup per transform, three register stages each way: the lengths
// of a first menu of 23 that beat the power-of-two one-launch kernels -- 1728 ... 2048, 3375 ... 4096 against 2048 / 4096 and every M of 4500 ... 8000
// (stages of 20 points, one workgroup per compute unit) against 8192 are level or slower, profiles/r06_s46_chirpz_reg3_ab.jsonl).  Compiled once per precision and shard: -DFOURIER_TU_REAL=float / double
// -DFOURIER_TILED_SHARD=i (fourier_amd/build.py, packaging/CMakeLists.txt).
#include "engine_common.h"
#include "kernels_chirpz.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, uint32_t R1, uint32_t R2> static ChirpzKernel make_chirpz() {
  if constexpr (R1 > 32 && sizeof(T) == 8) {
    return ChirpzKernel();  // (f64: a stage of more than 32 points spills, profiles/r06_s42_1000_point_tiles_f64.jsonl)
  } else {
    using C = ChirpzRegCfg<T, R1, R2>;
    ChirpzKernel k;
    k.fn = &chirpz_reg_kernel<T, R1, R2>;
    k.m = C::M; k.r1 = R1; k.r2 = R2; k.tpw = C::TPW; k.smem = C::SMEM;
    return k;
  }
}
template <typename T, uint32_t R1, uint32_t R2, uint32_t R3> static ChirpzKernel make_chirpz3() {
  using C = Chirpz3Cfg<T, R1, R2, R3>;
  ChirpzKernel k;
  k.fn = &chirpz_reg3_kernel<T, R1, R2, R3>;
  k.m = C::M; k.r1 = R1; k.r2 = R2; k.r3 = R3; k.tpw = C::NV; k.threads = C::NT; k.smem = C::SMEM;
  return k;
}
#define FOURIER_CHIRPZ(A, B) case (A) * (B): return make_chirpz<T, A, B>();
#define FOURIER_CHIRPZ3(A, B, C) case (A) * (B) * (C): return make_chirpz3<T, A, B, C>();
#if FOURIER_TILED_SHARD == 0
ChirpzKernel get_chirpz_kernel_s0(Real<TUReal>, uint32_t m) {
  typedef TUReal T;
  switch (m) {
    FOURIER_CHIRPZ(6, 6) FOURIER_CHIRPZ(10, 10) FOURIER_CHIRPZ(14, 14) FOURIER_CHIRPZ(18, 16) FOURIER_CHIRPZ(20, 20) FOURIER_CHIRPZ(25, 21)
    FOURIER_CHIRPZ(27, 25) FOURIER_CHIRPZ(30, 28)
    FOURIER_CHIRPZ3(12, 12, 9) FOURIER_CHIRPZ3(16, 16, 10)
    default: return ChirpzKernel();
  }
}
ChirpzKernel get_chirpz_kernel(Real<TUReal>, uint32_t m) {
  for (ChirpzKernel k : {get_chirpz_kernel_s0(Real<TUReal>{}, m), get_chirpz_kernel_s1(Real<TUReal>{}, m), get_chirpz_kernel_s2(Real<TUReal>{}, m),
                         get_chirpz_kernel_s3(Real<TUReal>{}, m)})
    if (k.fn) return k;
  return ChirpzKernel();
}
#elif FOURIER_TILED_SHARD == 1
ChirpzKernel get_chirpz_kernel_s1(Real<TUReal>, uint32_t m) {
  typedef TUReal T;
  switch (m) {
    FOURIER_CHIRPZ(7, 7) FOURIER_CHIRPZ(12, 10) FOURIER_CHIRPZ(15, 15) FOURIER_CHIRPZ(18, 18) FOURIER_CHIRPZ(21, 21) FOURIER_CHIRPZ(24, 24)
    FOURIER_CHIRPZ(27, 27) FOURIER_CHIRPZ(30, 30)
    FOURIER_CHIRPZ3(12, 12, 10) FOURIER_CHIRPZ3(21, 21, 20)
    default: return ChirpzKernel();
  }
}
#elif FOURIER_TILED_SHARD == 2
ChirpzKernel get_chirpz_kernel_s2(Real<TUReal>, uint32_t m) {
  typedef TUReal T;
  switch (m) {
    FOURIER_CHIRPZ(8, 8) FOURIER_CHIRPZ(12, 12) FOURIER_CHIRPZ(16, 16) FOURIER_CHIRPZ(20, 18) FOURIER_CHIRPZ(24, 20) FOURIER_CHIRPZ(25, 25)
    FOURIER_CHIRPZ(28, 28) FOURIER_CHIRPZ(32, 30)
    FOURIER_CHIRPZ3(16, 10, 10) FOURIER_CHIRPZ3(16, 16, 12)
    default: return ChirpzKernel();
  }
}
#else
ChirpzKernel get_chirpz_kernel_s3(Real<TUReal>, uint32_t m) {
  typedef TUReal T;
  switch (m) {
    FOURIER_CHIRPZ(9, 9) FOURIER_CHIRPZ(14, 12) FOURIER_CHIRPZ(32, 32)
    FOURIER_CHIRPZ(36, 32)
    FOURIER_CHIRPZ3(16, 12, 12) FOURIER_CHIRPZ3(21, 21, 21)
    default: return ChirpzKernel();
  }
}
#endif
#undef FOURIER_CHIRPZ
#undef FOURIER_CHIRPZ3

}  // namespace fourier_hip
