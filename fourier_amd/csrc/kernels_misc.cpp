// kernels_misc.cpp -- instantiates the lane-per-transform kernels (kernels_small.h), the odd-radix and global-memory passes and the unfused Bluestein sweeps (kernels_misc.h).
// Compiled once per precision: -DFOURIER_TU_REAL=float / double (fourier_amd/build.py).
#include "engine_common.h"
#include "kernels_small.h"
#include "kernels_misc.h"

namespace fourier_hip {

typedef FOURIER_TU_REAL TUReal;

// N <= 16 (f32: 32): one lane per transform; N = 1 (and anything else) the one-thread-per-transform form
TinyKernel get_tiny_kernel(Real<TUReal>, size_t n) {
  typedef TUReal T;
  return n == 32 ? &tiny_shfl_kernel<T, (sizeof(T) == 4 ? 32 : 16)>
         : n == 16 ? &tiny_shfl_kernel<T, 16>
         : n == 8 ? &tiny_shfl_kernel<T, 8>
         : n == 4 ? &tiny_shfl_kernel<T, 4>
         : n == 2 ? &tiny_shfl_kernel<T, 2> : &tiny_dft_kernel<T>;
}

OddKernel get_odd_kernel(Real<TUReal>, int r) {
  typedef TUReal T;
  switch (r) {
    case 3: return &odd_last_kernel<T, 3>;
    case 9: return &odd_last_kernel<T, 9>;
    case 27: return &odd_last_kernel<T, 27>;
    default: return nullptr;
  }
}

GenKernel get_stockham_pass_kernel(Real<TUReal>, int r) {
  typedef TUReal T;
  switch (r) {
    case 2: return &stockham_pass_kernel<T, 2>;
    case 3: return &stockham_pass_kernel<T, 3>;
    case 4: return &stockham_pass_kernel<T, 4>;
    case 8: return &stockham_pass_kernel<T, 8>;
    case 9: return &stockham_pass_kernel<T, 9>;
    case 16: return &stockham_pass_kernel<T, 16>;
    case 27: return &stockham_pass_kernel<T, 27>;
    default: return nullptr;
  }
}

BluKernel get_blu_kernel(Real<TUReal>, int which) {
  typedef TUReal T;
  return which == 0 ? &blu_pre_kernel<T> : (which == 1 ? &blu_post_kernel<T> : &blu_mul_kernel<T>);
}

}  // namespace fourier_hip
