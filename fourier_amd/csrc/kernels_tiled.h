// kernels_tiled.h -- big-radix Stockham passes of MIXED length: 2^a * 3^b (a < 12) beyond one compute unit's LDS.
//
// The reference runs such a length as one radix-2/3/4/8 pass after the other (autosort/mod.rs:104-116, 203-284); the
// global-pass route (stockham_pass_kernel) does the same with radices up to 27, one HBM round trip per radix -- four to six of
// them.  Here the length is split like the powers of two are (kernels_pass.h): N = L1 x L2 (x L3), every L_p = 2^x * 3^y
// <= 512, and ONE launch per factor computes
//     out[j + L*s*i + s*k] = W_size^{i*k} * DFT_L(in[j + s*i + s*m*k'])_k          (mod.rs:203-284 with R = L)
// on column tiles: a workgroup gathers COLS adjacent columns (128-byte row segments) of L rows into LDS, one column after
// the other, runs the L-point transforms of all columns there with the per-length LDS passes of kernels_mixed.h
// (MixPassesCT: the reference's radix schedule, tables and butterflies), multiplies by the inter-pass twiddle and writes the
// tile back -- transposed (every column's L outputs contiguous) in the first pass, as a column tile afterwards.  Two or
// three HBM round trips instead of one per radix.  Inverse transforms: swap . DFT . swap, as everywhere in this engine.
#pragma once
#include "kernels_mixed.h"

namespace fourier_hip {

template <typename T, uint32_t L> struct TiledCfg {  // the rules: tiled_shape (mixed_schedule.h), shared with the host
  static constexpr TileShape S = tiled_shape(L, (uint32_t)sizeof(cpx<T>));
  static constexpr uint32_t COLS = S.cols;  // 16 (f32) / 8 (f64) columns: 128-byte row segments
  static constexpr uint32_t LD = S.ld;      // odd leading dimension of a column in LDS
  static constexpr uint32_t POINTS = L * COLS;
  static constexpr uint32_t NT = S.threads;  // about eight points per thread
  static constexpr uint32_t KH = S.kh;       // inter-pass twiddle of a tile: W^{i*k} = TA[col][k / 16] * TB[col][k % 16]
  static constexpr size_t TAB_OFF = S.tab_off;
  static constexpr size_t SMEM = S.smem;
};

// One tile of one pass.  Columns: the first pass (s == 1) tiles the index i (m = n / L of them), later passes tile j (< s) at a
// fixed i; a ragged last tile of a row is handled by masking (lengths without a factor 16 have them).
template <typename T, uint32_t L>
__global__ void __launch_bounds__((TiledCfg<T, L>::NT)) tiled_mixed_kernel_ct(TiledArgs a) {
  using C = TiledCfg<T, L>;
  constexpr uint32_t COLS = C::COLS, LD = C::LD, NT = C::NT, KH = C::KH;
  FOURIER_DYN_SMEM(smem);
  cpx<T>* buf = (cpx<T>*)smem;
  cpx<T>* ta = (cpx<T>*)(smem + C::TAB_OFF);  // [COLS][KH]
  cpx<T>* tb = ta + COLS * KH;                // [COLS][16]
  const uint32_t tid = threadIdx.x;
  const bool first = (a.s == 1);
  // tile coordinates: block -> (transform b, i, first column c0)
  const uint32_t tiles_per_row = (uint32_t)a.tiles_per_row;      // ceil(columns per row / COLS)
  const uint32_t rows = first ? 1u : (uint32_t)a.m;              // values of i that have their own rows of tiles
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk);
  const uint32_t b = blk / (tiles_per_row * rows), rem = blk - b * (tiles_per_row * rows);
  const uint32_t i_row = rem / tiles_per_row, c0 = (rem - i_row * tiles_per_row) * COLS;
  const uint32_t ncols_total = first ? (uint32_t)a.m : (uint32_t)a.s;
  const uint32_t ncols = ncols_total - c0 < COLS ? ncols_total - c0 : COLS;
  const cpx<T>* __restrict__ in = (const cpx<T>*)a.in + (uint64_t)b * a.n;
  cpx<T>* __restrict__ out = (cpx<T>*)a.out + (uint64_t)b * a.n;
  // input element (column c, row k') = in[col0 + c + row_stride * k'], row_stride = s * m
  const uint64_t row_stride = a.s * a.m;
  const uint64_t col0 = first ? (uint64_t)c0 : (uint64_t)c0 + a.s * (uint64_t)i_row;

  // ---- every global load of the tile first, the two factors of each inter-pass twiddle table entry and then the data, so that one
  // memory latency covers both (the table loads used to complete -- they feed LDS writes -- before the gather was issued)
  // inter-pass twiddle tables of this tile (none in the last pass: size == L, mod.rs:238)
  const bool twiddled = a.m > 1;
  constexpr uint32_t TENT = COLS * (KH + 16), TITER = (TENT + NT - 1) / NT;
  cpx<T> tlo[TITER], thi[TITER];
  if (twiddled) {
    const cpx<T>* lo = (const cpx<T>*)a.tw_lo;
    const cpx<T>* hi = (const cpx<T>*)a.tw_hi;
    const uint32_t mask = (1u << a.lo_bits) - 1u;
#pragma unroll
    for (uint32_t it = 0; it < TITER; ++it) {
      const uint32_t e = tid + it * NT;
      if (e < TENT) {
        const uint32_t c = e / (KH + 16), q = e - c * (KH + 16);
        // (a masked column of a ragged last tile takes the last valid column's entries: its own index would reach past the
        // end of the tw_hi table -- ADVICE round 4; the values are never used)
        const uint64_t i = first ? (uint64_t)(c0 + c < ncols_total ? c0 + c : ncols_total - 1u) : (uint64_t)i_row;
        const uint64_t ex = i * (uint64_t)(q < KH ? 16u * q : q - KH);  // i * k < size
        tlo[it] = lo[ex & mask];
        thi[it] = hi[ex >> a.lo_bits];
      }
    }
  }

  // ---- gather: 128-byte row segments from global memory, column after column in LDS.  16-byte units (two f32 columns, one f64
  // column) -- a transform of odd length is only 8-byte aligned, which global_load_dwordx4 tolerates --, every load of a thread
  // issued before its first LDS write
  constexpr uint32_t VEC = 16 / (uint32_t)sizeof(cpx<T>), UPR = COLS / VEC;  // units per row segment
  constexpr uint32_t UNITS = L * UPR, ITER = (UNITS + NT - 1) / NT;
  {
    Unit16<T> v[ITER];
#pragma unroll
    for (uint32_t it = 0; it < ITER; ++it) {
      const uint32_t u = tid + it * NT, r = u / UPR, c = (u % UPR) * VEC;
      Unit16<T> w{};
      if (u < UNITS) {
        const cpx<T>* p = in + col0 + c + row_stride * (uint64_t)r;
        if (c + VEC <= ncols) w = load_unit_a8<T>(p);
        else if (c < ncols) { w.a[0] = p->re; w.a[1] = p->im; }  // ragged tile, f32: the last valid column by itself
      }
      v[it] = w;
    }
    if (twiddled) {
#pragma unroll
      for (uint32_t it = 0; it < TITER; ++it) {
        const uint32_t e = tid + it * NT;
        if (e < TENT) {
          const uint32_t c = e / (KH + 16), q = e - c * (KH + 16);
          const cpx<T> w = cmul(tlo[it], thi[it]);
          if (q < KH) ta[c * KH + q] = w; else tb[c * 16 + (q - KH)] = w;
        }
      }
    }
#pragma unroll
    for (uint32_t it = 0; it < ITER; ++it) {
      const uint32_t u = tid + it * NT, r = u / UPR, c = (u % UPR) * VEC;
      if (u < UNITS) {
#pragma unroll
        for (uint32_t e = 0; e < VEC; ++e) {
          cpx<T> z{v[it].a[2 * e], v[it].a[2 * e + 1]};
          if (a.swap_in) z = {z.im, z.re};
          buf[(c + e) * LD + r] = z;
        }
      }
    }
  }
  __syncthreads();

  // ---- the L-point transforms of the tile's columns: the reference's schedule in LDS, in place
  const cpx<T> w3{(T)a.w3re, (T)a.w3im}, w8{(T)a.w8re, (T)a.w8im};
  MixPassesCT<T, L, L, 1, 0, true, COLS, NT, LD>::run(buf, buf, (const cpx<T>*)a.tw, COLS, true, w3, w8);

  // ---- twiddle and store, again in 16-byte units with the LDS reads and twiddle products of a thread ahead of its stores
  const T scale = (T)a.scale;
  auto finish = [&](uint32_t c, uint32_t k, uint32_t tc) {  // element (column c, output k) of the tile; tc: table column
    cpx<T> y = buf[c * LD + k];
    if (twiddled) y = cmul(y, cmul(ta[tc * KH + (k >> 4)], tb[tc * 16 + (k & 15)]));
    else {  // last pass (mod.rs:238: no twiddle): the user-level scaling and the inverse's trailing swap
      if (a.swap_out) y = {y.im, y.re};
      y = {y.re * scale, y.im * scale};
    }
    return y;
  };
  if (first) {
    // out[L*i + k]: the tile's output is ONE contiguous run of ncols * L elements
    cpx<T>* o = out + (uint64_t)L * c0;
    const uint32_t total = ncols * L;
    constexpr uint32_t OUNITS = (L * COLS + VEC - 1) / VEC, OITER = (OUNITS + NT - 1) / NT;
    Unit16<T> v[OITER];
#pragma unroll
    for (uint32_t it = 0; it < OITER; ++it) {
      const uint32_t e0 = (tid + it * NT) * VEC;
#pragma unroll
      for (uint32_t e = 0; e < VEC; ++e) {
        const uint32_t idx = e0 + e, c = idx / L, k = idx - c * L;
        const cpx<T> y = idx < total ? finish(c, k, c) : cpx<T>{0, 0};
        v[it].a[2 * e] = y.re; v[it].a[2 * e + 1] = y.im;
      }
    }
#pragma unroll
    for (uint32_t it = 0; it < OITER; ++it) {
      const uint32_t e0 = (tid + it * NT) * VEC;
      if (e0 + VEC <= total) store_unit_a8<T>(o + e0, v[it]);
      else if (e0 < total) o[e0] = cpx<T>{v[it].a[0], v[it].a[1]};
    }
  } else {
    // out[j + L*s*i + s*k]: 128-byte row segments again, row k at stride s; one i for the whole tile (table column 0)
    cpx<T>* o = out + (uint64_t)c0 + (uint64_t)L * a.s * (uint64_t)i_row;
    Unit16<T> v[ITER];
#pragma unroll
    for (uint32_t it = 0; it < ITER; ++it) {
      const uint32_t u = tid + it * NT, k = u / UPR, c = (u % UPR) * VEC;
#pragma unroll
      for (uint32_t e = 0; e < VEC; ++e) {
        const cpx<T> y = (u < UNITS && c + e < ncols) ? finish(c + e, k, 0) : cpx<T>{0, 0};
        v[it].a[2 * e] = y.re; v[it].a[2 * e + 1] = y.im;
      }
    }
#pragma unroll
    for (uint32_t it = 0; it < ITER; ++it) {
      const uint32_t u = tid + it * NT, k = u / UPR, c = (u % UPR) * VEC;
      if (u >= UNITS) continue;
      cpx<T>* p = o + c + a.s * (uint64_t)k;
      if (c + VEC <= ncols) store_unit_a8<T>(p, v[it]);
      else if (c < ncols) *p = cpx<T>{v[it].a[0], v[it].a[1]};
    }
  }
}

}  // namespace fourier_hip
