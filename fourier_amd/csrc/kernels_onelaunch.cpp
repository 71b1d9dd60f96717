// kernels_onelaunch.cpp -- instantiates the one-launch plans (kernels_onelaunch.h): both passes of 2^11..2^15 and the whole chirp-z for M <= 2^15.
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_onelaunch.h"
#include "tile_shapes.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, int L1, int L2> static KernelInfo make_twolevel_info() {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  using CA = TileCfg<T, L1, L2 / VEC>;
  using CB = TileCfg<T, L2, L1 / VEC>;
  KernelInfo k;
  k.fn = &fft_twolevel_kernel<T, L1, L2>;
  k.L = L1; k.CG = L2 / VEC; k.NT = CA::NT; k.COLS = L2; k.R3 = 1;
  // the two in-tile exchanges and the transposes between them share one buffer (both role orders: the one-launch chirp-z
  // runs the L2 x L1 problem behind the L1 x L2 one)
  k.smem = std::max({CA::EXCH_BYTES, CB::EXCH_BYTES, TwolevelTr<T, L1, L2>::BYTES, TwolevelTr<T, L2, L1>::BYTES});
  return k;
}
// single-launch plans: 2^10 = 32x32 (f64), 2^11 = 64x32 (72 % of the HBM peak vs 57 % for the row kernel), 2^12 = 64x64,
// 2^13 = 128x64, 2^14 = 128x128, 2^15 = 256x128 (f32 only: the transform must fit one workgroup's
// registers, at most 1024 threads x 16 points x VEC)
bool get_twolevel_kernel(Real<TUReal>, int k, KernelInfo& info, int& l1, int& l2) {
  typedef TUReal T;
  switch (k) {
    case 10:  // f64 only: 75.2 against 72.4 % of the HBM peak for the whole-row kernel (f32: 60 against 74 %; profiles/r04_s29_*)
      if constexpr (sizeof(T) == 8) { info = make_twolevel_info<T, 32, 32>(); l1 = 32; l2 = 32; return true; }
      return false;
    case 11: info = make_twolevel_info<T, 64, 32>(); l1 = 64; l2 = 32; return true;
    case 12: info = make_twolevel_info<T, 64, 64>(); l1 = 64; l2 = 64; return true;
    case 13: info = make_twolevel_info<T, 128, 64>(); l1 = 128; l2 = 64; return true;
    case 14: info = make_twolevel_info<T, 128, 128>(); l1 = 128; l2 = 128; return true;
    case 15:
      if constexpr (sizeof(T) == 4) { info = make_twolevel_info<T, 256, 128>(); l1 = 256; l2 = 128; return true; }
      return false;
    default: return false;
  }
}

template <typename T, int L1, int L2> static KernelInfo make_blu_small_info() {
  KernelInfo k = make_twolevel_info<T, L1, L2>();
  k.fn = &bluestein_small_kernel<T, L1, L2>;
  return k;
}
template <typename T, int L, int CG> static KernelInfo make_blu_rows_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &bluestein_rows_kernel<T, L, CG>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::EXCH_BYTES;
  return k;
}
bool get_blu_small_kernel(Real<TUReal>, int k, KernelInfo& info) {
  typedef TUReal T;
  switch (k) {
    case 4: info = make_blu_rows_info<T, 16, 64>(); return true;   // same tile shapes as the row kernels
    case 5: info = make_blu_rows_info<T, 32, 32>(); return true;
    // M = 64 ... 1024: ONE WAVE per workgroup (64 / Q column groups) -- every barrier of the chain chirp -> FFT -> (.)w -> FFT -> chirp is then
    // inside a wave and the sixteen waves of a CU run sixteen independent chains instead of four lock-stepped groups of four: +8 ... 20 % over
    // the 256-thread workgroups of rounds 3 - 5 (f32 N = 439: 27.1 -> 32.5 % of the HBM peak, 331: 22.2 -> 27.0 %; f64 M = 1024 is best at four
    // column groups; profiles/r06_s5_chirpz_rows_tuning_ab.jsonl, r06_s6_chirpz_one_wave_ab.jsonl)
    case 6: info = make_blu_rows_info<T, 64, 16>(); return true;
    case 7: info = make_blu_rows_info<T, 128, 8>(); return true;
    case 8: info = make_blu_rows_info<T, 256, 4>(); return true;
    case 9: info = make_blu_rows_info<T, 512, 2>(); return true;
    case 10: info = make_blu_rows_info<T, 1024, (sizeof(T) == 4 ? 1 : 4)>(); return true;
    case 11: info = make_blu_small_info<T, 64, 32>(); return true;
    case 12: info = make_blu_small_info<T, 64, 64>(); return true;
    case 13: info = make_blu_small_info<T, 128, 64>(); return true;
    case 14: info = make_blu_small_info<T, 128, 128>(); return true;
    case 15:
      if constexpr (sizeof(T) == 4) { info = make_blu_small_info<T, 256, 128>(); return true; }
      return false;
    default: return false;
  }
}

}  // namespace fourier_hip
