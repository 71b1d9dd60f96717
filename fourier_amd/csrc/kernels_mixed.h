// kernels_mixed.h -- LDS-resident Stockham autosort on the reference's own schedule, tables and butterfly order
// (autosort/mod.rs:20-46,104-116,203-284; autosort/butterfly.rs:3-65), runtime-parameterised and per length.
#pragma once
#include "kernels_common.h"
#include "mixed_schedule.h"

namespace fourier_hip {

// ---- native Stockham autosort for small mixed-radix sizes N = 2^a * 3^b (b > 0), N <= 4096 ----
// This kernel is the reference's algorithm verbatim, one workgroup per group of transforms, all passes in
// LDS: radix schedule [4,8,4,3,2] (autosort/mod.rs:20-21,104-116), per-pass twiddle table
// [1, W^i, .., W^{(R-1)i}] (mod.rs:24-46), pass body out[j + R*s*i + s*k] = tw[i*R+k] * butterflyR(in[j + s*i + s*m*k'])_k
// (mod.rs:203-284), butterflies in the reference's operation order (autosort/butterfly.rs:3-65,
// vector/generic.rs:22-44) with FMA contraction off (Rust never fuses), then the scale pass (mod.rs:381-399).
// Same tables, same order, same roundings: results are bit-identical to the CPU restatement.
#ifndef FOURIER_EMU
#define FOURIER_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define FOURIER_NO_CONTRACT
#endif

template <typename T> __device__ __forceinline__ cpx<T> ref_mul(cpx<T> a, cpx<T> b) {
  FOURIER_NO_CONTRACT
  const T rr = a.re * b.re, ii = a.im * b.im, ri = a.re * b.im, ir = a.im * b.re;
  return {rr - ii, ri + ir};
}
template <typename T> __device__ __forceinline__ cpx<T> ref_add(cpx<T> a, cpx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ __forceinline__ cpx<T> ref_sub(cpx<T> a, cpx<T> b) { return {a.re - b.re, a.im - b.im}; }
// generic.rs:34-44
template <typename T> __device__ __forceinline__ cpx<T> ref_rotate(cpx<T> z, bool positive) {
  return positive ? cpx<T>{-z.im, z.re} : cpx<T>{z.im, -z.re};
}
template <typename T> __device__ __forceinline__ void ref_bf2(cpx<T>& a, cpx<T>& b) {  // butterfly.rs:3-5
  const cpx<T> s = ref_add(a, b), d = ref_sub(a, b);
  a = s; b = d;
}
template <typename T> __device__ __forceinline__ void ref_bf4(cpx<T>* x, bool fwd) {  // butterfly.rs:26-43
  cpx<T> a0 = x[0], a1 = x[2], a2 = x[1], a3 = x[3];
  ref_bf2(a0, a1);  // a[0], a[1]
  ref_bf2(a2, a3);  // a[2], a[3]
  a3 = ref_rotate(a3, fwd);
  ref_bf2(a0, a2);  // b[0], b[1]
  ref_bf2(a1, a3);  // b[2], b[3]
  x[0] = a0; x[1] = a3; x[2] = a2; x[3] = a1;  // [b0, b3, b1, b2]
}
template <typename T> __device__ __forceinline__ void ref_bf3(cpx<T>* x, cpx<T> t) {  // butterfly.rs:9-22
  const cpx<T> tc{t.re, -t.im};
  const cpx<T> y0 = ref_add(x[0], ref_add(x[1], x[2]));
  const cpx<T> y1 = ref_add(x[0], ref_add(ref_mul(x[1], t), ref_mul(x[2], tc)));
  const cpx<T> y2 = ref_add(x[0], ref_add(ref_mul(x[1], tc), ref_mul(x[2], t)));
  x[0] = y0; x[1] = y1; x[2] = y2;
}
template <typename T> __device__ __forceinline__ void ref_bf8(cpx<T>* x, bool fwd, cpx<T> t) {  // butterfly.rs:47-65
  const cpx<T> tneg{-t.re, t.im};
  cpx<T> a1[4] = {x[0], x[2], x[4], x[6]};
  cpx<T> b1[4] = {x[1], x[3], x[5], x[7]};
  ref_bf4(a1, fwd);
  ref_bf4(b1, fwd);
  b1[1] = ref_mul(b1[1], t);
  b1[2] = ref_rotate(b1[2], !fwd);
  b1[3] = ref_mul(b1[3], tneg);
#pragma unroll
  for (int k = 0; k < 4; ++k) ref_bf2(a1[k], b1[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) { x[k] = a1[k]; x[4 + k] = b1[k]; }
}

// ---- beyond the reference: butterflies of prime radix 5, 7, 11, 13 ----
// The reference sends every length with a prime factor above 3 to Bluestein (fourier/src/lib.rs:38-42).  Lengths whose
// prime factors stop at 13 run here instead, on the same Stockham pass (mod.rs:203-284) with the radix list continued
// [4, 8, 4, 3, 2, 5, 7, 11, 13]: one LDS-resident launch instead of two padded power-of-two transforms, and closer to the
// exact DFT than the chirp-z route (the results agree with the reference's within the Bluestein tolerance, they are not
// bit-identical -- there is no reference arithmetic for these radices to be identical to).
// DFT of prime length R by symmetry: with a_q = x_q + x_{R-q}, d_q = x_q - x_{R-q} (q = 1 .. (R-1)/2)
//   y_k, y_{R-k} = x_0 + sum_q cos(2 pi k q / R) a_q  -/+  i * sum_q sin(2 pi k q / R) d_q     (forward; inverse swaps the signs)
template <int R> struct PrimeTab { double c[R], s[R]; };   // cos / sin (2 pi j / R), j < R
template <int R> constexpr PrimeTab<R> prime_tab();
template <> constexpr PrimeTab<5> prime_tab<5>() {
  return {{1.0, 0.3090169943749474241, -0.8090169943749474241, -0.8090169943749474241, 0.3090169943749474241},
          {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917, -0.95105651629515357212}};
}
template <> constexpr PrimeTab<7> prime_tab<7>() {
  return {{1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624, -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053},
          {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048, -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871}};
}
template <> constexpr PrimeTab<11> prime_tab<11>() {
  return {{1.0, 0.84125353283118116886, 0.41541501300188642553, -0.14231483827328514044, -0.65486073394528506406, -0.95949297361449738989, -0.95949297361449738989, -0.65486073394528506406, -0.14231483827328514044, 0.41541501300188642553, 0.84125353283118116886},
          {0.0, 0.54064081745559758211, 0.90963199535451837141, 0.98982144188093273238, 0.75574957435425828377, 0.28173255684142969771, -0.28173255684142969771, -0.75574957435425828377, -0.98982144188093273238, -0.90963199535451837141, -0.54064081745559758211}};
}
template <> constexpr PrimeTab<13> prime_tab<13>() {
  return {{1.0, 0.8854560256532098959, 0.56806474673115580251, 0.12053668025532305335, -0.35460488704253562597, -0.74851074817110109863, -0.97094181742605202716, -0.97094181742605202716, -0.74851074817110109863, -0.35460488704253562597, 0.12053668025532305335, 0.56806474673115580251, 0.8854560256532098959},
          {0.0, 0.46472317204376854566, 0.82298386589365639458, 0.9927088740980539928, 0.93501624268541482344, 0.66312265824079520238, 0.23931566428755776715, -0.23931566428755776715, -0.66312265824079520238, -0.93501624268541482344, -0.9927088740980539928, -0.82298386589365639458, -0.46472317204376854566}};
}
template <typename T, int R> __device__ __forceinline__ void dft_prime(cpx<T>* x, bool fwd) {
  constexpr PrimeTab<R> tab = prime_tab<R>();
  constexpr int H = (R - 1) / 2;
  cpx<T> a[H], d[H];
  cpx<T> y0 = x[0];
#pragma unroll
  for (int q = 1; q <= H; ++q) {
    a[q - 1] = {x[q].re + x[R - q].re, x[q].im + x[R - q].im};
    d[q - 1] = {x[q].re - x[R - q].re, x[q].im - x[R - q].im};
    y0 = {y0.re + a[q - 1].re, y0.im + a[q - 1].im};
  }
  const T sg = fwd ? (T)1 : (T)-1;
  const cpx<T> x0 = x[0];
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    cpx<T> m = x0, n = {(T)0, (T)0};
#pragma unroll
    for (int q = 1; q <= H; ++q) {
      const T c = (T)tab.c[(k * q) % R], sn = (T)tab.s[(k * q) % R];
      m = {m.re + c * a[q - 1].re, m.im + c * a[q - 1].im};
      n = {n.re + sn * d[q - 1].re, n.im + sn * d[q - 1].im};
    }
    const cpx<T> r = {sg * n.im, -sg * n.re};  // -i*n forward, +i*n inverse
    x[k] = {m.re + r.re, m.im + r.im};
    x[R - k] = {m.re - r.re, m.im - r.im};
  }
  x[0] = y0;
}

template <typename T, int R> __device__ __forceinline__ void ref_butterfly(cpx<T>* x, bool fwd, cpx<T> w3, cpx<T> w8) {
  if constexpr (R == 2) ref_bf2(x[0], x[1]);
  else if constexpr (R == 3) ref_bf3(x, w3);
  else if constexpr (R == 4) ref_bf4(x, fwd);
  else if constexpr (R == 8) ref_bf8(x, fwd, w8);
  else dft_prime<T, R>(x, fwd);
}

// global memory <-> LDS in 16-byte units (two f32 points / one f64 point per lane and instruction; the rows of an odd-length f32
// batch are only 8-byte aligned, which global_load/store_dwordx4 tolerate).  MAXU: compile-time bound of the units of a workgroup.
// Up to eight units per thread are LOADED before the first of them is written: as a plain loop (`for u: lds[u] = g[u]`) hipcc
// emits load, s_waitcnt vmcnt(0), ds_write per iteration -- one exposed HBM latency per 16 bytes of a thread's share (three per
// workgroup at 768 points: +4 ... 22 % once batched, round 4).  Plain loads and stores: with streaming hints the lengths move by -7 ... +4 % (f64 96 / 100
// alone gain 4 - 14 %) with no rule to them, and the mixed-length tile passes lose up to 40 % (profiles/r06_s18_*, r06_s19_mixed_lds_*).
template <typename T, uint32_t NT, uint32_t MAXU, uint32_t CH = 8>  // CH: units of a thread in flight together
__device__ __forceinline__ void copy_in_units(cpx<T>* lds, const cpx<T>* g, uint32_t units) {
  constexpr uint32_t VEC = 16 / (uint32_t)sizeof(cpx<T>);
  constexpr uint32_t IT = (MAXU + NT - 1) / NT;
#pragma unroll
  for (uint32_t c0 = 0; c0 < IT; c0 += CH) {
    Unit16<T> w[CH];
#pragma unroll
    for (uint32_t q = 0; q < CH; ++q) {
      const uint32_t u = threadIdx.x + (c0 + q) * NT;
      if (c0 + q < IT && u < units) w[q] = load_unit_a8<T>(g + u * VEC);
    }
#pragma unroll
    for (uint32_t q = 0; q < CH; ++q) {
      const uint32_t u = threadIdx.x + (c0 + q) * NT;
      if (c0 + q < IT && u < units) *(Unit16<T>*)(lds + u * VEC) = w[q];
    }
  }
}
// ... and back, scaled (mod.rs:387-393: the unscaled codes skip the multiply, x * 1 is exact)
template <typename T, uint32_t NT, uint32_t MAXU, uint32_t CH = 8>
__device__ __forceinline__ void copy_out_units(cpx<T>* g, const cpx<T>* lds, uint32_t units, bool scaled, T scale) {
  constexpr uint32_t VEC = 16 / (uint32_t)sizeof(cpx<T>);
  constexpr uint32_t IT = (MAXU + NT - 1) / NT;
  // (the thread index behind a launder: the addresses below are then computed here, not shared with copy_in_units and carried -- or
  // spilled, under the register caps of the runtime-parameterised kernels -- across every pass)
  uint32_t tix = threadIdx.x;
  FOURIER_LAUNDER(tix);
#pragma unroll
  for (uint32_t c0 = 0; c0 < IT; c0 += CH) {
    Unit16<T> w[CH];
#pragma unroll
    for (uint32_t q = 0; q < CH; ++q) {
      const uint32_t u = tix + (c0 + q) * NT;
      if (c0 + q < IT && u < units) w[q] = *(const Unit16<T>*)(lds + u * VEC);
    }
#pragma unroll
    for (uint32_t q = 0; q < CH; ++q) {
      const uint32_t u = tix + (c0 + q) * NT;
      if (c0 + q < IT && u < units) {
        Unit16<T> v = w[q];
        if (scaled) {
#pragma unroll
          for (uint32_t c = 0; c < 2 * VEC; ++c) v.a[c] = v.a[c] * scale;
        }
        store_unit_a8<T>(g + u * VEC, v);
      }
    }
  }
}

// One pass of the runtime-parameterised kernel, IN PLACE on one LDS buffer: a thread computes up to ROUNDS butterflies,
// keeps their outputs in registers across a barrier and writes them back to the buffer it read from (the per-length
// kernels below do the same with every index a constant).  PPT = points per thread the instantiation is sized for.
template <typename T, int R, int PPT, int NT>
__device__ __forceinline__ void mixed_pass(cpx<T>* __restrict__ buf, const cpx<T>* __restrict__ tw, uint32_t n, uint32_t nb,
                                           uint32_t size, uint32_t stride, bool fwd, cpx<T> w3, cpx<T> w8) {
  constexpr int ROUNDS = (PPT + R - 1) / R;
  const uint32_t m = size / R, nbf = n / R, total = nb * nbf;
  // q / nbf and e / stride without integer division: operands stay below 2^16 (a workgroup holds <= 8192 points), so
  // the float quotient is off by at most one and a compare fixes it.  Every product below fits 24 bits: mul24 is a
  // full-rate instruction where the 32-bit multiply runs at a quarter (the first version spent 60+ v_mul_lo_u32 a pass);
  // the R addresses of a butterfly advance by addition.
  const float inv_nbf = fast_rcp((float)nbf), inv_stride = fast_rcp((float)stride);
  const uint32_t in_step = mul24(stride, m);
  cpx<T> y[ROUNDS][R];
  uint32_t off[ROUNDS];
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    const uint32_t q = threadIdx.x + (uint32_t)NT * rd;
    if (q < total) {
      uint32_t g = (uint32_t)((float)q * inv_nbf);
      g -= (mul24(g, nbf) > q); g += (mul24(g + 1, nbf) <= q);
      const uint32_t e = q - mul24(g, nbf);
      uint32_t i = (uint32_t)((float)e * inv_stride);
      i -= (mul24(i, stride) > e); i += (mul24(i + 1, stride) <= e);
      const uint32_t is = mul24(i, stride), j = e - is, base = mul24(g, n) + j;
      // The twiddles first: issued behind the butterfly, inside the reference's `size != R` branch (mod.rs:238,272) -- where
      // the compiler sinks them when the multiply is conditional -- their L2 latency adds to the LDS latency of every pass
      // instead of hiding under it.  So the multiply is unconditional: the last pass reads W^0 = (1, -0) from its table
      // section and multiplies by it, which returns every finite value unchanged.
      cpx<T> w[R];
      const cpx<T>* __restrict__ twi = tw + mul24(i, (uint32_t)R);
      constexpr bool EARLY = sizeof(T) == 4;  // f64: the early loads cost registers the 1024-thread kernels do not have
      if constexpr (EARLY) {
#pragma unroll
        for (int k = 1; k < R; ++k) w[k] = twi[k];
        FOURIER_SCHED_FENCE();
      }
      uint32_t idx = base + is;
#pragma unroll
      for (int k = 0; k < R; ++k) { y[rd][k] = buf[idx]; idx += in_step; }
      ref_butterfly<T, R>(y[rd], fwd, w3, w8);
      if constexpr (!EARLY) {
        FOURIER_SCHED_FENCE();
#pragma unroll
        for (int k = 1; k < R; ++k) w[k] = twi[k];
      }
#pragma unroll
      for (int k = 1; k < R; ++k) {
        if (!fwd) w[k].im = -w[k].im;  // inverse table = conj (twiddle.rs:14-18)
        y[rd][k] = ref_mul(y[rd][k], w[k]);
      }
      off[rd] = base + mul24(is, (uint32_t)R);
    }
  }
  __syncthreads();  // every input of the pass has been read
#pragma unroll
  for (int rd = 0; rd < ROUNDS; ++rd) {
    if (threadIdx.x + (uint32_t)NT * rd < total) {
      uint32_t idx = off[rd];
#pragma unroll
      for (int k = 0; k < R; ++k) { buf[idx] = y[rd][k]; idx += stride; }
    }
  }
  __syncthreads();
}

// The runtime-parameterised kernel: lengths with factors 5..13 that have no per-length kernel (and, in experiments builds,
// every length for A/B).  MAXP: the largest prime radix this instantiation carries (3: the reference's list; 7, 13: the
// continued list) -- the radix-13 butterfly's 26 live values would otherwise set the register allocation of every length.
// NT threads, PPT points per thread: group * n <= NT * PPT (128 x 8 / 256 x 4 / 256 x 8 up to 2048 points, 512 x 8 up to 4096, 1024 x 8 up to 8192).
// The second launch bound (waves per SIMD) is what makes hipcc economise: left at 128 threads and no bound it spends 119
// VGPRs on the f32 radix-7 instantiation, which halves the resident workgroups of a latency-bound kernel.
#define FOURIER_MIX_RT_WAVES(T, MAXP, PPT) ((sizeof(T) == 4 ? ((MAXP) <= 7 ? ((PPT) <= 4 ? 6 : ((PPT) <= 8 ? 5 : 4)) : 4) : ((MAXP) <= 7 ? ((PPT) <= 4 ? 4 : ((PPT) <= 8 ? 3 : 4)) : 2)))
template <typename T, int MAXP, int PPT, int NT>
__global__ void __launch_bounds__(NT, FOURIER_MIX_RT_WAVES(T, MAXP, PPT)) mixed_radix_kernel(MixArgs a) {
  FOURIER_DYN_SMEM(smem);
  cpx<T>* buf = (cpx<T>*)smem;
  const uint64_t b0 = (uint64_t)blockIdx.x * a.group;
  const uint32_t nb = (uint32_t)((a.batch - b0) < a.group ? (a.batch - b0) : a.group);
  const uint32_t total = nb * a.n;
  const cpx<T>* in = (const cpx<T>*)a.in + b0 * a.n;
  cpx<T>* out = (cpx<T>*)a.out + b0 * a.n;
  // global <-> LDS in 16-byte units (see mixed_radix_kernel_ct)
  constexpr uint32_t VEC = 16 / (2 * (uint32_t)sizeof(T));
  const uint32_t units = total / VEC;
  copy_in_units<T, NT, (uint32_t)NT * PPT / VEC, 4>(buf, in, units);  // (four in flight: the register caps of these kernels)
  if constexpr (VEC > 1) {
    if ((total % VEC) && threadIdx.x == 0) buf[total - 1] = in[total - 1];
  }
  __syncthreads();
  const bool fwd = a.forward != 0;
  cpx<T> w3{(T)a.w3re, (T)a.w3im}, w8{(T)a.w8re, (T)a.w8im};
  if (!fwd) { w3.im = -w3.im; w8.im = -w8.im; }
  const cpx<T>* tw = (const cpx<T>*)a.tw;
  uint32_t size = a.n, stride = 1;
  for (uint32_t ps = 0; ps < a.npass; ++ps) {
    const uint32_t R = a.radix[ps];
    if (R == 8) mixed_pass<T, 8, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
    else if (R == 4) mixed_pass<T, 4, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
    else if (R == 3) mixed_pass<T, 3, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
    else if (R == 2) mixed_pass<T, 2, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
    else if constexpr (MAXP >= 5) {
      if (R == 5) mixed_pass<T, 5, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
      else if (R == 7) mixed_pass<T, 7, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
      else if constexpr (MAXP >= 11) {
        if (R == 11) mixed_pass<T, 11, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
        else mixed_pass<T, 13, PPT, NT>(buf, tw, a.n, nb, size, stride, fwd, w3, w8);
      }
    }
    tw += size;  // each pass consumes `size` entries (mod.rs:357,377)
    size /= R;
    stride *= R;
  }
  const T scale = a.scaled ? (T)a.scale : (T)1;  // mod.rs:387-393 (the unscaled codes skip the multiply: x * 1 is exact)
  copy_out_units<T, NT, (uint32_t)NT * PPT / VEC, 4>(out, buf, units, a.scaled != 0, scale);
  if constexpr (VEC > 1) {
    if ((total % VEC) && threadIdx.x == 0) {
      cpx<T> y = buf[total - 1];
      if (a.scaled) y = {y.re * scale, y.im * scale};
      out[total - 1] = y;
    }
  }
}

// ---- the same kernel with the transform length fixed at compile time ----
// For the sizes the reference itself benchmarks (3^5, 3^6, 3^7, fft_bench.rs:153-159) and the common 3*2^k / 9*2^k
// lengths: schedule, sizes, strides and table offsets are constants, so the per-butterfly index arithmetic
// (two runtime divisions in mixed_pass) folds into multiply-shifts and the pass loop unrolls.  Same operations in
// the same order: still bit-identical to the CPU restatement.
// GROUP transforms per workgroup on NT threads, transform g at element g * LD of the buffer (defaults: the per-length kernels'
// own rules; the tile passes of kernels_tiled.h run COLS = GROUP columns at a padded leading dimension)
// LIN: the layout (mix_out_layout) of the data this pass reads
template <typename T, uint32_t N, uint32_t SIZE, uint32_t STRIDE, uint32_t TWOFF, bool FIRST_PASS, uint32_t GROUP = mix_group<T>(N),
          uint32_t NT_ = mix_threads<T>(N), uint32_t LD = N, bool GIO = false, uint32_t LIN = 0>
struct MixPassesCT {
  // entry (butterfly i, output k) of a pass's table of m butterflies: the reference's layout [i][k] (mod.rs:24-46).  (A transposed copy
  // staged in LDS where a workgroup's transforms share the tables was built and measured: no gain; profiles/r06_removed_ab_knobs.patch)
  static __device__ __forceinline__ constexpr uint32_t tw_at(uint32_t i, uint32_t k, uint32_t r, uint32_t m) { (void)m; return i * r + k; }
  static constexpr uint32_t R = mix_next_radix(N, SIZE, FIRST_PASS), M = SIZE / R, NT = NT_;
  static constexpr bool PAIR = ((R == 3 || R == 5) && SIZE >= R * R && (SIZE / R) % R == 0 && mix_pairs<T>(N, R));
  // two consecutive radix-R passes (R = 3, 5) on one LDS round trip: the R butterflies (i + M2*k2, j), k2 < R, of this pass
  // write exactly the inputs of the R butterflies (i, j + STRIDE*k), k < R, of the next one, so a thread that
  // loads those R*R points keeps them in registers in between -- same operations in the same order as two single
  // passes (mod.rs:203-284 twice), half the LDS traffic, barriers and index arithmetic
  static constexpr uint32_t SIZE2 = SIZE / R, M2 = SIZE2 / R;
  static constexpr uint32_t PTS = PAIR ? R * R : R;      // points one work item reads and writes
  static constexpr uint32_t NBF = N / PTS;               // work items per transform
  static constexpr uint32_t OUT_SIZE = PAIR ? SIZE2 / R : SIZE / R, OUT_STRIDE = STRIDE * PTS;
  static constexpr uint32_t OUT_TWOFF = PAIR ? TWOFF + SIZE + SIZE2 : TWOFF + SIZE;
  static constexpr bool LAST = (OUT_SIZE == 1);
  static constexpr uint32_t LOUT = mix_out_layout(N, STRIDE, PTS, LAST, sizeof(cpx<T>) == 8 ? 4u : 3u);  // the layout this pass writes
  // element e0 + off of a transform in the layout this pass reads; `off` a compile-time constant
  static __device__ __forceinline__ uint32_t wr_index(uint32_t q, uint32_t off, uint32_t k) {
    if (LOUT == 0) return off + STRIDE * k;
    return (GROUP == 1 ? 0 : (q / NBF) * LD) + mix_sw(LOUT, off + STRIDE * k);
  }
  // (pointer plus constant, not an index sum: the constant then folds into the instruction's offset field)
  static __device__ __forceinline__ const cpx<T>* rd_ptr(const cpx<T>* in, uint32_t e0, uint32_t off) {
    if (LIN == 0 || off % mix_sw_span(LIN) == 0) return (in + mix_sw(LIN, e0)) + off;  // the offset does not reach the map's bits
    return in + mix_sw(LIN, e0 + off);
  }

  // work item q: its PTS inputs, in the order finish() consumes them ([k2][k1] for a pass pair)
  static __device__ __forceinline__ void load(const cpx<T>* src, uint32_t q, cpx<T> (&x)[PTS]) {
    const uint32_t g = GROUP == 1 ? 0 : q / NBF, e = GROUP == 1 ? q : q % NBF, i = e / STRIDE, j = e % STRIDE;  // constants: multiply-shift
    const cpx<T>* in = src + g * LD;
    const uint32_t e0 = j + STRIDE * i;
    if constexpr (PAIR) {
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2)
#pragma unroll
        for (uint32_t k1 = 0; k1 < R; ++k1) {
          LDS_NOTE(rd_ptr(in, e0, STRIDE * (M2 * k2 + M * k1)), sizeof(cpx<T>), false, 100);
          x[k2 * R + k1] = *rd_ptr(in, e0, STRIDE * (M2 * k2 + M * k1));
        }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < R; ++k) {
        LDS_NOTE(rd_ptr(in, e0, STRIDE * M * k), sizeof(cpx<T>), false, 101);
        x[k] = *rd_ptr(in, e0, STRIDE * M * k);
      }
    }
  }
  // work item q: butterfly (+ twiddle) of its inputs, results in y[PTS] in the order of the output slots
  static __device__ __forceinline__ void finish(const cpx<T> (&xin)[PTS], const cpx<T>* tw, uint32_t q, bool fwd, cpx<T> w3, cpx<T> w8,
                                                cpx<T> (&y)[PTS], uint32_t& out_off) {
    const uint32_t g = GROUP == 1 ? 0 : q / NBF, e = GROUP == 1 ? q : q % NBF, i = e / STRIDE, j = e % STRIDE;
    const cpx<T>* __restrict__ t = tw + TWOFF;
    // plain output layout: the index in the buffer; else the index within the transform (wr_index() adds g * LD behind the map)
    out_off = (LOUT == 0 ? g * LD : 0) + j + PTS * STRIDE * i;
    if constexpr (PAIR) {
      const cpx<T>* __restrict__ t2 = tw + TWOFF + SIZE;
      cpx<T> x[R][R];
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2)
#pragma unroll
        for (uint32_t k1 = 0; k1 < R; ++k1) x[k2][k1] = xin[k2 * R + k1];
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2) {
        ref_butterfly<T, (int)R>(x[k2], fwd, w3, w8);
#pragma unroll
        for (uint32_t k = 1; k < R; ++k) {
          cpx<T> w = t[tw_at(i + M2 * k2, k, R, M)];
          if (!fwd) w.im = -w.im;
          x[k2][k] = ref_mul(x[k2][k], w);
        }
      }
#pragma unroll
      for (uint32_t k = 0; k < R; ++k) {
        cpx<T> z[R];
#pragma unroll
        for (uint32_t k2 = 0; k2 < R; ++k2) z[k2] = x[k2][k];
        ref_butterfly<T, (int)R>(z, fwd, w3, w8);
        if constexpr (SIZE2 != R) {
#pragma unroll
          for (uint32_t k2 = 1; k2 < R; ++k2) {
            cpx<T> w = t2[tw_at(i, k2, R, M2)];
            if (!fwd) w.im = -w.im;
            z[k2] = ref_mul(z[k2], w);
          }
        }
#pragma unroll
        for (uint32_t k2 = 0; k2 < R; ++k2) y[k + R * k2] = z[k2];  // output slot STRIDE * (k + R*k2)
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < R; ++k) y[k] = xin[k];
      ref_butterfly<T, (int)R>(y, fwd, w3, w8);
      if constexpr (SIZE != R) {  // mod.rs:238,272
#pragma unroll
        for (uint32_t k = 1; k < R; ++k) {
          cpx<T> w = t[tw_at(i, k, R, M)];
          if (!fwd) w.im = -w.im;
          y[k] = ref_mul(y[k], w);
        }
      }
    }
  }
  // work item q: load, butterfly (+ twiddle)
  static __device__ __forceinline__ void compute(const cpx<T>* src, const cpx<T>* tw, uint32_t q, bool fwd, cpx<T> w3, cpx<T> w8,
                                                 cpx<T> (&y)[PTS], uint32_t& out_off) {
    cpx<T> x[PTS];
    load(src, q, x);
    finish(x, tw, q, fwd, w3, w8, y, out_off);
  }

  using Next = MixPassesCT<T, N, OUT_SIZE, OUT_STRIDE, OUT_TWOFF, false, GROUP, NT_, LD, GIO, LOUT>;
  // entries of all tables from this pass on (the host uploads them back to back, mod.rs:24-46)
  static constexpr uint32_t table_end() {
    if constexpr (LAST) return OUT_TWOFF; else return Next::table_end();
  }
  // gin / gout / scale: GIO only (mix_gio) -- the first pass reads the transforms from global memory (same element order as the
  // LDS buffer), the last pass scales and writes them to global memory; returns nullptr then (nothing is left to copy out)
  static __device__ __forceinline__ const cpx<T>* run(const cpx<T>* src, cpx<T>* dst, const cpx<T>* tw, uint32_t nb, bool fwd,
                                                     cpx<T> w3, cpx<T> w8, const cpx<T>* gin = nullptr, cpx<T>* gout = nullptr,
                                                     T scale = (T)1, bool scaled = false) {
    if constexpr (mix_inplace<T>(N)) {
      constexpr bool FROM_GLOBAL = GIO && FIRST_PASS, TO_GLOBAL = GIO && LAST;
      constexpr uint32_t ROUNDS = (GROUP * NBF + NT - 1) / NT;
      cpx<T> y[ROUNDS][PTS];
      uint32_t off[ROUNDS];
      if constexpr (mix_loads_first<T>(N) && ROUNDS > 1) {  // the loads of every work item ahead of the first butterfly
        cpx<T> x[ROUNDS][PTS];
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
          const uint32_t q = threadIdx.x + NT * rd;
          if (q < nb * NBF) load(FROM_GLOBAL ? gin : src, q, x[rd]);
        }
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
          const uint32_t q = threadIdx.x + NT * rd;
          if (q < nb * NBF) finish(x[rd], tw, q, fwd, w3, w8, y[rd], off[rd]);
        }
      } else {
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
          const uint32_t q = threadIdx.x + NT * rd;
          if (q < nb * NBF) compute(FROM_GLOBAL ? gin : src, tw, q, fwd, w3, w8, y[rd], off[rd]);
        }
      }
      if constexpr (TO_GLOBAL) {
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
          const uint32_t q = threadIdx.x + NT * rd;
          if (q < nb * NBF) {
#pragma unroll
            for (uint32_t k = 0; k < PTS; ++k) {
              cpx<T> z = y[rd][k];
              if (scaled) z = {z.re * scale, z.im * scale};  // mod.rs:387-393
              gout[off[rd] + STRIDE * k] = z;  // (the last pass writes the plain layout)
            }
          }
        }
        return nullptr;
      } else {
        if constexpr (!FROM_GLOBAL) __syncthreads();  // every input of the pass has been read (nothing to wait for when they came from global memory)
        cpx<T>* buf = const_cast<cpx<T>*>(src);
#pragma unroll
        for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
          const uint32_t q = threadIdx.x + NT * rd;
          if (q < nb * NBF) {
#pragma unroll
            for (uint32_t k = 0; k < PTS; ++k) {
              const uint32_t idx = wr_index(q, off[rd], k);
              LDS_NOTE(buf + idx, sizeof(cpx<T>), true, 102);
              buf[idx] = y[rd][k];
            }
          }
        }
        __syncthreads();
        if constexpr (LAST) return src;
        else return Next::run(src, dst, tw, nb, fwd, w3, w8, gin, gout, scale, scaled);
      }
    } else {
      for (uint32_t q = threadIdx.x; q < nb * NBF; q += NT) {
        cpx<T> y[PTS];
        uint32_t off;
        compute(src, tw, q, fwd, w3, w8, y, off);
#pragma unroll
        for (uint32_t k = 0; k < PTS; ++k) dst[wr_index(q, off, k)] = y[k];
      }
      __syncthreads();
      if constexpr (LAST) return dst;
      else return Next::run(dst, const_cast<cpx<T>*>(src), tw, nb, fwd, w3, w8);
    }
  }
};
template <typename T, uint32_t N>
__global__ void __launch_bounds__(mix_threads<T>(N)) mixed_radix_kernel_ct(MixArgs a) {
  constexpr uint32_t NT = mix_threads<T>(N);
  FOURIER_DYN_SMEM(smem);
  constexpr uint32_t GROUP = mix_group<T>(N);
  cpx<T>* buf0 = (cpx<T>*)smem;
  cpx<T>* buf1 = buf0 + (size_t)GROUP * N;
  const uint64_t b0 = (uint64_t)blockIdx.x * GROUP;
  const uint32_t nb = (uint32_t)((a.batch - b0) < GROUP ? (a.batch - b0) : GROUP);
  const uint32_t total = nb * N;
  const cpx<T>* in = (const cpx<T>*)a.in + b0 * N;
  cpx<T>* out = (cpx<T>*)a.out + b0 * N;
  // global <-> LDS in 16-byte units (two f32 points / one f64 point per lane and instruction; the user rows of an
  // odd-length f32 batch are only 8-byte aligned, which global_load/store_dwordx4 tolerate), one odd point by itself --
  // unless the first pass reads global memory itself and the last pass writes it (mix_gio)
  constexpr uint32_t VEC = 16 / (2 * (uint32_t)sizeof(T));
  constexpr bool GIO = mix_gio<T>(N);
  const uint32_t units = total / VEC;
  if constexpr (GIO) {
  } else {
    copy_in_units<T, NT, (GROUP * N + VEC - 1) / VEC>(buf0, in, units);
    if constexpr (VEC > 1) {
      if ((total % VEC) && threadIdx.x == 0) buf0[total - 1] = in[total - 1];
    }
  }
  using Passes = MixPassesCT<T, N, N, 1, 0, true, GROUP, NT, N, GIO>;
  const cpx<T>* tw = (const cpx<T>*)a.tw;
  __syncthreads();
  const bool fwd = a.forward != 0;
  cpx<T> w3{(T)a.w3re, (T)a.w3im}, w8{(T)a.w8re, (T)a.w8im};
  if (!fwd) { w3.im = -w3.im; w8.im = -w8.im; }
  const T scale = a.scaled ? (T)a.scale : (T)1;  // mod.rs:387-393 (the unscaled codes skip the multiply: x * 1 is exact)
  const cpx<T>* res = Passes::run(buf0, buf1, tw, nb, fwd, w3, w8, in, out, scale, a.scaled != 0);
  if constexpr (GIO) {
    (void)res; (void)units;  // the last pass has written the output
  } else {
    copy_out_units<T, NT, (GROUP * N + VEC - 1) / VEC>(out, res, units, a.scaled != 0, scale);
    if constexpr (VEC > 1) {
      if ((total % VEC) && threadIdx.x == 0) {
        cpx<T> y = res[total - 1];
        if (a.scaled) y = {y.re * scale, y.im * scale};
        out[total - 1] = y;
      }
    }
  }
}

}  // namespace fourier_hip
