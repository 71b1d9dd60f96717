// kernels_misc.h -- odd-radix passes of the large 2^a*3^b plans, the one-radix-per-launch global pass, and the unfused
// Bluestein pointwise sweeps.
#pragma once
#include "kernels_common.h"

namespace fourier_hip {

// ---- final odd-radix Stockham pass for large N = 2^a * 3^b: R = 3^b in {3, 9, 27}, s = 2^a, m = 1 ----
// out[j + s*k] = DFT_R(in[j + s*k'])_k  (autosort/mod.rs:203-284 with size == R: no twiddle, :238).  The
// reference reaches radix 3 last as well (RADICES = [4,8,4,3,2], mod.rs:21).  One thread owns VEC adjacent
// columns j (one 16-byte unit per row) and all R rows: fully coalesced, in place allowed.
// radix-3 butterfly, forward: W3 = -1/2 - i*sqrt(3)/2 (the values of butterfly.rs:9-22, regrouped)
template <typename T> __device__ __forceinline__ void dft3(cpx<T>& a, cpx<T>& b, cpx<T>& c) {
  const T h = (T)0.86602540378443864676;
  const cpx<T> s = {b.re + c.re, b.im + c.im}, d = {b.re - c.re, b.im - c.im};
  const cpx<T> m = {a.re - (T)0.5 * s.re, a.im - (T)0.5 * s.im};
  const cpx<T> r = {h * d.im, -h * d.re};  // -i*h*d
  a = {a.re + s.re, a.im + s.im};
  b = {m.re + r.re, m.im + r.im};
  c = {m.re - r.re, m.im - r.im};
}
// natural-order DFT of R = 3^b points at x[0], x[STRIDE], ... using the table W_RT^e (RT = top-level radix)
template <typename T, int R, int RT, int STRIDE, typename Args>
__device__ __forceinline__ void dft_pow3(cpx<T>* x, const Args& a) {
  if constexpr (R == 3) {
    dft3(x[0], x[STRIDE], x[2 * STRIDE]);
  } else {
    constexpr int M = R / 3;
    // decimation in time: sub-transforms over n = 3*q + c (c = 0,1,2)
    cpx<T> e[3][M];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int q = 0; q < M; ++q) e[c][q] = x[(3 * q + c) * STRIDE];
      dft_pow3<T, M, RT, 1>(e[c], a);
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      cpx<T> u = e[0][k];
      cpx<T> v = cmul(e[1][k], cpx<T>{(T)a.wr[(RT / R) * k], (T)a.wi[(RT / R) * k]});
      cpx<T> w = cmul(e[2][k], cpx<T>{(T)a.wr[(RT / R) * 2 * k], (T)a.wi[(RT / R) * 2 * k]});
      dft3(u, v, w);
      x[k * STRIDE] = u; x[(k + M) * STRIDE] = v; x[(k + 2 * M) * STRIDE] = w;
    }
  }
}

template <typename T, int R>
__global__ void __launch_bounds__(256) odd_last_kernel(OddArgs a) {
  // One radix-R (R = 3, 9, 27) Stockham pass at stride s over the odd part of a 2^a*3^b plan (mod.rs:203-284 with the
  // reference's radix order, the odd radices after the powers of two):
  //   out[j + R*s*i + s*k] = W_size^{i*k} * DFT_R(in[j + s*i + s*m*k'])_k,  i < m, j < s.
  // m == 1 is the final pass (no twiddle; scaling / swap applied); m > 1 a middle pass.
  // One thread per (transform, i, 16-byte unit of j): s is a multiple of 4096, so a wave shares i and reads whole lines.
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  const uint64_t units = a.s / VEC;                       // 16-byte units per row
  const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t per = units * a.m;                       // threads per transform
  if (gid >= a.batch * per) return;
  const uint64_t b = gid / per, rem = gid - b * per;
  const uint64_t i = rem / units, u = rem - i * units;
  const bool last = (a.m == 1);
  const cpx<T>* in = (const cpx<T>*)a.in + b * a.n + u * VEC + a.s * i;
  cpx<T>* out = (cpx<T>*)a.out + b * a.n + u * VEC + (uint64_t)R * a.s * i;
  const uint64_t in_step = a.s * a.m;
  cpx<T> x[VEC][R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const Unit16<T> v = load_unit<T, false>(in + (uint64_t)k * in_step);
#pragma unroll
    for (int c = 0; c < VEC; ++c) x[c][k] = {v.a[2 * c], v.a[2 * c + 1]};
  }
#pragma unroll
  for (int c = 0; c < VEC; ++c) dft_pow3<T, R, R, 1>(x[c], a);
  const T scale = (T)a.scale;
  const cpx<T>* tw = (const cpx<T>*)a.tw;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    Unit16<T> v;
    cpx<T> w{(T)1, (T)0};
    if (!last && k > 0) w = tw[i * (uint64_t)k];          // wave-uniform address
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      cpx<T> y = x[c][k];
      if (last) {
        if (a.swap_out) y = {y.im, y.re};
        y = {y.re * scale, y.im * scale};
      } else if (k > 0) {
        y = cmul(y, w);
      }
      v.a[2 * c] = y.re; v.a[2 * c + 1] = y.im;
    }
    if (last) store_unit<T, true>(out + (uint64_t)k * a.s, v);
    else store_unit<T, false>(out + (uint64_t)k * a.s, v);
  }
}

// ---- one Stockham pass in global memory, any radix R in {2,3,4,8,9,16,27}, any stride: the 2^a*3^b lengths with a < 12
// that do not fit the LDS kernels (3^10, 2^8*3^5, ...).  The reference's pass verbatim (autosort/mod.rs:203-284):
//   out[j + R*s*i + s*k] = W_size^{i*k} * DFT_R(in[j + s*i + s*m*k'])_k,   i < m, j < s, size = R*m,
// one thread per butterfly e = j + s*i: for a fixed k' the reads in[e + s*m*k'] are contiguous over the threads whatever
// the stride; the writes are contiguous in runs of s.  Passes are scheduled odd radices first (27, 9, 3), then 16, 8, 4, 2
// (GenericEngine); every pass is one HBM round trip.
template <typename T, int R>
__global__ void __launch_bounds__(256) stockham_pass_kernel(GenArgs a) {
  const uint32_t per = a.s * a.m;
  const uint32_t b = blockIdx.x / a.blocks_per;                                     // wave-uniform: scalar division
  const uint32_t e0 = (blockIdx.x - b * a.blocks_per) * 256u, e = e0 + threadIdx.x;
  const bool valid = e < per;
  const uint32_t i = e / a.s, j = e - i * a.s;
  const cpx<T>* in = (const cpx<T>*)a.in + (uint64_t)b * a.n + e;
  cpx<T> x[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    x[k] = valid ? in[(uint64_t)per * k] : cpx<T>{0, 0};
    if (a.swap_in) x[k] = {x[k].im, x[k].re};
  }
  if constexpr (R == 3 || R == 9 || R == 27) dft_pow3<T, R, R, 1>(x, a);
  else dft_r<T, R>(x);
  // W_size^{i*k} from a two-level table (2 * sqrt(size) entries, cache-resident) instead of the reference's size-entry table
  // (mod.rs:24-46): one more complex multiply, no table the size of the transform in device memory
  const cpx<T>* lo = (const cpx<T>*)a.tw_lo;
  const cpx<T>* hi = (const cpx<T>*)a.tw_hi;
  const uint32_t lo_mask = (1u << a.lo_bits) - 1u;
  const T scale = (T)a.scale;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (lo && k > 0 && valid) {
      const uint64_t e = (uint64_t)i * (uint64_t)k;  // < size
      x[k] = cmul(x[k], cmul(lo[e & lo_mask], hi[e >> a.lo_bits]));
    }
    if (a.final_pass) {
      if (a.swap_out) x[k] = {x[k].im, x[k].re};
      x[k] = {x[k].re * scale, x[k].im * scale};
    }
  }
  if (a.s == 1) {
    // first pass: thread i owns out[R*i .. R*i + R), a lane stride of R elements -- every store instruction would touch 64
    // different lines.  The workgroup's 256 * R outputs are one contiguous run: stage them in LDS, store them linearly.
    FOURIER_DYN_SMEM(smem);
    cpx<T>* stage = (cpx<T>*)smem;
#pragma unroll
    for (int k = 0; k < R; ++k) stage[(uint32_t)R * threadIdx.x + (uint32_t)k] = x[k];
    __syncthreads();
    const uint32_t left = per - e0, count = (uint32_t)R * (left < 256u ? left : 256u);
    cpx<T>* out = (cpx<T>*)a.out + (uint64_t)b * a.n + (uint64_t)R * e0;
    for (uint32_t idx = threadIdx.x; idx < count; idx += 256u) out[idx] = stage[idx];
    return;
  }
  if (!valid) return;
  cpx<T>* out = (cpx<T>*)a.out + (uint64_t)b * a.n + j + (uint64_t)R * a.s * i;
#pragma unroll
  for (int k = 0; k < R; ++k) out[(uint64_t)a.s * k] = x[k];
}

// ---- Bluestein chirp-z pointwise steps (reference: fourier-algorithms/src/bluesteins.rs:229-258) ----
// work[b][i] *= w[i], i < m                                     (bluesteins.rs:236-239; unfused options only)
template <typename T>
__global__ void __launch_bounds__(256) blu_mul_kernel(BluArgs a) {
  cpx<T>* work = (cpx<T>*)a.out;
  const cpx<T>* wt = (const cpx<T>*)a.xtab;
  const uint64_t total = a.batch * a.m;
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256)
    work[idx] = cmul(work[idx], wt[idx % a.m]);
}
// work[b][i] = x[i] * in[b][i] for i < n, 0 for n <= i < m      (bluesteins.rs:229-234)
template <typename T>
__global__ void __launch_bounds__(256) blu_pre_kernel(BluArgs a) {
  const cpx<T>* in = (const cpx<T>*)a.in;
  cpx<T>* work = (cpx<T>*)a.out;
  const cpx<T>* xt = (const cpx<T>*)a.xtab;
  const uint64_t total = a.batch * a.m;
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint64_t b = idx / a.m, i = idx - b * a.m;
    cpx<T> y{0, 0};
    if (i < a.n) {
      cpx<T> v = in[b * a.n + i];
      if (a.swap) v = {v.im, v.re};
      y = cmul(xt[i], v);
    }
    work[idx] = y;
  }
}
// out[b][i] = work[b][i] * x[i] * scale for i < n                (bluesteins.rs:240-258)
template <typename T>
__global__ void __launch_bounds__(256) blu_post_kernel(BluArgs a) {
  const cpx<T>* work = (const cpx<T>*)a.in;
  cpx<T>* out = (cpx<T>*)a.out;
  const cpx<T>* xt = (const cpx<T>*)a.xtab;
  const T scale = (T)a.scale;
  const uint64_t total = a.batch * a.n;
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint64_t b = idx / a.n, i = idx - b * a.n;
    cpx<T> y = cmul(work[b * a.m + i], xt[i]);
    if (a.swap) y = {y.im, y.re};
    out[idx] = {y.re * scale, y.im * scale};
  }
}

}  // namespace fourier_hip
