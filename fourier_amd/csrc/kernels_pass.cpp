// kernels_pass.cpp -- instantiates the big-radix pass kernels (kernels_pass.h) and hands them to the plan layer.
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_pass.h"
#include "tile_shapes.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

template <typename T, int L, int CG, int MODE, int IO = IO_PLAIN> static KernelInfo make_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_pass_kernel<T, L, CG, MODE, IO>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::smem_bytes(MODE) + (IO == IO_BLU_IN ? C::TABV_BYTES : 0);
  return k;
}

KernelInfo get_kernel(Real<TUReal>, int L, int mode, int io) {
  typedef TUReal T;
  // first pass of length 4096 on 32-byte-wide tiles (128 KiB, two workgroups per CU): 2^22 = 4096 x 1024
  if (L == 4096 && mode == MODE_FIRST && io == IO_PLAIN) return make_info<T, 4096, FOURIER_CG_4096, MODE_FIRST>();
  if (L == 2048 && !dev_env("FOURIER_WIDE_2048")) {
    if (mode == MODE_FIRST)
      return io == IO_BLU_IN ? make_info<T, 2048, FOURIER_CG_2048_FIRST, MODE_FIRST, IO_BLU_IN>()
                             : make_info<T, 2048, FOURIER_CG_2048_FIRST, MODE_FIRST>();
    // half tiles for the last pass were measured 5-7 % SLOWER than the 16-column kernel (profiles/r02_s2_*_ab.jsonl:
    // 13.6-14.0 vs 13.1 ms per 1024 transforms of 2^22); kept behind FOURIER_SPLIT_2048=1 for experiments
    if (mode == MODE_LAST && dev_env("FOURIER_SPLIT_2048")) {
      const KernelInfo k = get_split_kernel(Real<T>{}, 2048, io);  // experiments library only
      if (k.fn) return k;
    }
  }
#define FK(LL, CGG) FK2(LL, CGG, CGG)
#define FK2(LL, CGG, CG_ROWS) /* CG_ROWS: the tile width of the whole-row kernel of this length */ \
  case LL:                                                                                       \
    switch (mode) {                                                                              \
      case MODE_FIRST:                                                                           \
        return io == IO_BLU_IN ? make_info<T, LL, CGG, MODE_FIRST, IO_BLU_IN>()                  \
                               : make_info<T, LL, CGG, MODE_FIRST>();                            \
      case MODE_MID: return make_info<T, LL, CGG, MODE_MID>();                                   \
      case MODE_LAST:                                                                            \
        return io == IO_BLU_OUT ? make_info<T, LL, CGG, MODE_LAST, IO_BLU_OUT>()                 \
                                : make_info<T, LL, CGG, MODE_LAST>();                            \
      default: return make_info<T, LL, CG_ROWS, MODE_ROWS>();                                    \
    }
#define FK_ROWS_ONLY(LL, CGG) \
  case LL: return make_info<T, LL, CGG, MODE_ROWS>();
  switch (L) {
    FK_ROWS_ONLY(16, 64)
    FK_ROWS_ONLY(32, 32)
    FK(64, 16)
    // whole rows of 128 points: 64 (f64: 32) transforms per 256-thread workgroup -- f32 64.5 -> 68.0 % of the HBM peak, f64 68.3 -> 70.2 %
    // against 32 / 16 per 128 threads (profiles/r04_s28_rows_tile_width_64_128_ab.jsonl; N = 64 gains nothing from either width)
    FK2(128, 16, FOURIER_CG_128_ROWS)
    FK(256, 16)
    FK(512, FOURIER_CG_512)
    // whole rows of 1024 points, f32: 8 transforms per 256-thread workgroup instead of 16 per 512 -- 61.4 -> 67.1 % of the HBM peak; N = 256 / 512
    // lose 2 - 6 % on narrower workgroups and stay (profiles/r06_s6_rows_width_ab.jsonl; f64 1024 is a 32 x 32 one-launch plan)
    FK2(1024, FOURIER_CG_1024, (sizeof(T) == 4 ? 4 : FOURIER_CG_1024))
    FK(2048, FOURIER_CG_2048)
    default: break;
  }
#undef FK
#undef FK2
#undef FK_ROWS_ONLY
  throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no kernel for pass length " + std::to_string(L));
}

// fft_conv_kernel: forward LAST + (.) w + inverse FIRST of a Bluestein plan, same tile shapes as the passes
template <typename T, int L, int CG> static KernelInfo make_conv_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_conv_kernel<T, L, CG>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::smem_bytes(MODE_FIRST);
  return k;
}
KernelInfo get_conv_kernel(Real<TUReal>, int L) {
  typedef TUReal T;
  switch (L) {
    case 64: return make_conv_info<T, 64, 16>();
    case 128: return make_conv_info<T, 128, 16>();
    case 256: return make_conv_info<T, 256, 16>();
    case 512: return make_conv_info<T, 512, FOURIER_CG_512>();
    case 1024: return make_conv_info<T, 1024, FOURIER_CG_1024>();
    case 2048: return make_conv_info<T, 2048, FOURIER_CG_2048>();
    default: break;
  }
  throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no conv kernel for pass length " + std::to_string(L));
}

}  // namespace fourier_hip
