// kernels_experiments.cpp -- instantiates the measured-slower designs (kernels_experiments.h); linked into lib/libfourier_experiments.so and the CPU emulation build only.
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_experiments.h"
#include "tile_shapes.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

// last pass of length L on half tiles: the register tile (and the thread count) of a length-L/2 pass
template <typename T, int L, int CG, int IO = IO_PLAIN> static KernelInfo make_split_info() {
  using C = TileCfg<T, L / 2, CG>;
  KernelInfo k;
  k.fn = &fft_last_split_kernel<T, L / 2, CG, IO>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3; k.split = 1;
  k.smem = C::smem_bytes(MODE_LAST);
  return k;
}

KernelInfo get_split_kernel(Real<TUReal>, int L, int io) {
  typedef TUReal T;
  if (L != 2048) return KernelInfo();
  return io == IO_BLU_OUT ? make_split_info<T, 2048, FOURIER_CG_1024, IO_BLU_OUT>() : make_split_info<T, 2048, FOURIER_CG_1024>();
}

template <typename T, int L1, int CG1, int L2, int CG2> static FusedInfo make_fused_info() {
  using CA = TileCfg<T, L1, CG1>;
  using CB = TileCfg<T, L2, CG2>;
  FusedInfo k;
  k.fn = &fft_l2fused_kernel<T, L1, CG1, L2, CG2>;
  k.L1 = L1; k.L2 = L2; k.NT = CA::NT; k.COLS_A = CA::COLS; k.COLS_B = CB::COLS;
  const size_t sa = CA::smem_bytes(MODE_FIRST), sb = CB::smem_bytes(MODE_LAST);
  k.smem = (((sa > sb ? sa : sb) + 15) & ~(size_t)15) + 16;  // + the broadcast slot
  return k;
}
// N * sizeof(complex) <= 2 MiB and two passes: f32 2^16 .. 2^18, f64 2^15 .. 2^17 (64 KiB tiles, 256 threads)
bool get_fused_kernel(Real<TUReal>, int k, FusedInfo& info) {
  typedef TUReal T;
  if constexpr (sizeof(T) == 4) {
    switch (k) {
      case 16: info = make_fused_info<T, 256, 16, 256, 16>(); return true;
      case 17: info = make_fused_info<T, 512, 8, 256, 16>(); return true;
      case 18: info = make_fused_info<T, 512, 8, 512, 8>(); return true;
      default: return false;
    }
  } else {
    switch (k) {
      case 15: info = make_fused_info<T, 256, 16, 128, 32>(); return true;
      case 16: info = make_fused_info<T, 256, 16, 256, 16>(); return true;
      case 17: info = make_fused_info<T, 512, 8, 256, 16>(); return true;
      default: return false;
    }
  }
}

template <typename T, int L, int CG, int IO> static KernelInfo make_prefetch_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_last_prefetch_kernel<T, L, CG, IO>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = PrefetchCfg<T, L, CG>::SMEM;
  return k;
}
KernelInfo get_prefetch_kernel(Real<TUReal>, int L, int io) {
  typedef TUReal T;
  if (io != IO_PLAIN && io != IO_BLU_OUT) return KernelInfo();
  switch (L) {
    case 1024: return io == IO_BLU_OUT ? make_prefetch_info<T, 1024, FOURIER_CG_1024, IO_BLU_OUT>() : make_prefetch_info<T, 1024, FOURIER_CG_1024, IO_PLAIN>();
    case 2048: return io == IO_BLU_OUT ? make_prefetch_info<T, 2048, FOURIER_CG_2048, IO_BLU_OUT>() : make_prefetch_info<T, 2048, FOURIER_CG_2048, IO_PLAIN>();
    default: return KernelInfo();
  }
}

}  // namespace fourier_hip
