// kernels_small.h -- transforms of length 1..32: one lane per transform.
#pragma once
#include "kernels_common.h"

namespace fourier_hip {

// ---- transforms of length 2, 4, 8, 16: one lane per transform, coalesced I/O through wave shuffles ----
// A transform is U = N * sizeof(complex) / 16 consecutive 16-byte units.  The wave loads its 64 transforms as
// 64*U consecutive units (lane l takes units l, 64 + l, ...: whole 1 KiB lines per instruction); U adjacent
// lanes then hold one part each of U transforms, and a U x U transpose over those lanes (log2 U rounds of
// __shfl_xor with a register select) hands every lane one whole transform for the in-register butterfly.  The
// transpose is its own inverse, so the same routine restores the unit order for the coalesced store.
template <int U> __device__ __forceinline__ void transpose_units(int (&reg)[U][4], int lane) {
#pragma unroll
  for (int s = 1; s < U; s <<= 1) {
    const bool hi = (lane & s) != 0;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (j & s) continue;
#pragma unroll
      for (int d = 0; d < 4; ++d) {  // scalar selects: an array-element select would go through scratch memory
        const int lo_reg = reg[j][d], hi_reg = reg[j ^ s][d];
        const int recv = __shfl_xor(hi ? lo_reg : hi_reg, s);
        reg[j][d] = hi ? recv : lo_reg;
        reg[j ^ s][d] = hi ? hi_reg : recv;
      }
    }
  }
}
template <typename T, int N>
__global__ void __launch_bounds__(256) tiny_shfl_kernel(TinyArgs a) {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  constexpr int U = N / VEC;
  static_assert(U >= 1 && U <= 16, "tiny_shfl_kernel: 16..256-byte transforms");
  // streaming hints on the loads and stores up to 16 points: +1 ... 6 % on all eight (length, precision) cases, f32 32 loses 3 %
  // (profiles/r06_s19_tiny_policy_ab.jsonl)
  constexpr bool STREAM = N <= 16;
  const int lane = (int)threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const uint64_t u0 = wave * 64 * U, total_units = a.batch * (uint64_t)U;
  const cpx<T>* in = (const cpx<T>*)a.in;
  cpx<T>* out = (cpx<T>*)a.out;
  int reg[U][4];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint64_t u = u0 + (uint64_t)(64 * j + lane);
    Unit16<T> v{};
    if (u < total_units) v = load_unit_a8<T, STREAM>(in + u * VEC);
    __builtin_memcpy(reg[j], &v, 16);
  }
  transpose_units<U>(reg, lane);  // lane (g = lane / U, q = lane % U) owns transform (64 / U) * q + g of the wave
  cpx<T> x[N];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    Unit16<T> v;
    __builtin_memcpy(&v, reg[j], 16);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      x[j * VEC + c] = {v.a[2 * c], v.a[2 * c + 1]};
      if (a.swap_in) x[j * VEC + c] = {x[j * VEC + c].im, x[j * VEC + c].re};
    }
  }
  dft_r<T, N>(x);
  const T scale = (T)a.scale;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    Unit16<T> v;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      cpx<T> y = x[j * VEC + c];
      if (a.swap_out) y = {y.im, y.re};
      v.a[2 * c] = y.re * scale; v.a[2 * c + 1] = y.im * scale;
    }
    __builtin_memcpy(reg[j], &v, 16);
  }
  transpose_units<U>(reg, lane);
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint64_t u = u0 + (uint64_t)(64 * j + lane);
    Unit16<T> v;
    __builtin_memcpy(&v, reg[j], 16);
    if (u < total_units) store_unit_a8<T, STREAM>(out + u * VEC, v);
  }
}

// ---- transforms of length 1 (and the generic fallback form): one thread per transform ----
template <typename T>
__global__ void __launch_bounds__(256) tiny_dft_kernel(TinyArgs a) {
  const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= a.batch) return;
  const cpx<T>* in = (const cpx<T>*)a.in + b * a.n;
  cpx<T>* out = (cpx<T>*)a.out + b * a.n;
  cpx<T> x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    x[i] = (i < a.n) ? in[i] : cpx<T>{0, 0};
    if (a.swap_in) x[i] = {x[i].im, x[i].re};
  }
  if (a.n == 2) dft2(x); else if (a.n == 4) dft4(x); else if (a.n == 8) dft8(x);
  const T scale = (T)a.scale;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < a.n) {
      cpx<T> y = x[i];
      if (a.swap_out) y = {y.im, y.re};
      out[i] = {y.re * scale, y.im * scale};
    }
  }
}

}  // namespace fourier_hip
