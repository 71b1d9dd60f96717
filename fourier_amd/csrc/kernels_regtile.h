// kernels_regtile.h -- column-tile passes of MIXED length with the transform in REGISTERS (round 6).
//
// Same pass as kernels_tiled.h -- N = L1 x L2 (x L3), one launch per factor computes
//     out[j + L*s*i + s*k] = W_size^{i*k} * DFT_L(in[j + s*i + s*m*k'])_k          (autosort/mod.rs:203-284 with R = L)
// on tiles of COLS adjacent columns (128-byte row segments) -- but the L-point transform of a column is ONE Cooley-Tukey step
// L = R1 x R2 between two register-resident transforms (what kernels_pass.h does for the powers of two with 16 x 16 x R3):
//   stage A  thread (column c, j2 < R2) loads rows j2 + R2*j1 (j1 < R1) straight from global memory, transforms them (DFT_R1 in
//            registers, any R1 <= 32 whose prime factors stop at 13), multiplies by W_L^{j2*k1} and writes LDS;
//   stage B  thread (column c, k1 < R1) reads its R2 values, transforms them (DFT_R2), multiplies by the inter-pass twiddle
//            W_size^{i*(k1 + R1*k2)} and stores -- 128-byte row segments straight from registers in the later passes, through LDS
//            (every column's L outputs contiguous: mod.rs:203-284 at s = 1) in the first.
// kernels_tiled.h gathers the tile into LDS and runs the reference's small-radix schedule there (four or five LDS round trips with a
// barrier each, their index arithmetic per butterfly): 3.6 - 4.2 TB/s per pass against 5.9 for the power-of-two tiles, VALU and LDS
// instructions per byte 2.7 x / 3.2 x theirs (profiles/r06_s20_sq_tiled.json).  Here: one LDS round trip (two in the first pass), every
// index a compile-time constant.  Tolerance-only route like every tile pass of mixed length (include/fourier.h).
//
// Bluestein with a SMOOTH M = L1 x L2 (bluesteins.rs:110 asks for M >= 2N - 1 only; the reference takes the next power of two, up to
// 4N): the three sweeps of kernels_pass.h on these tiles -- tiled_reg_kernel<IO_BLU_IN> (first forward pass of length L1, reads the user
// array times the chirp, zero padded: bluesteins.rs:229-234), tiled_reg_conv_kernel (last forward pass of length L2, (.) w, first inverse
// pass of length L2 in one launch: bluesteins.rs:236-239) and tiled_reg_kernel<IO_BLU_OUT> (last inverse pass of length L1, times chirp
// and scale, the first N points into the user array: bluesteins.rs:240-258).
#pragma once
#include "kernels_mixed.h"

namespace fourier_hip {

// ---- unit roots at compile time: exp(-2 pi i e / R) = (c[e], -s[e]) ----
constexpr double ct_sin_poly(double x) {  // |x| <= pi / 4: Taylor to x^19, below 1e-17 there
  const double x2 = x * x;
  double r = -1.0 / 121645100408832000.0;  // -1 / 19!
  r = r * x2 + 1.0 / 355687428096000.0;
  r = r * x2 - 1.0 / 1307674368000.0;
  r = r * x2 + 1.0 / 6227020800.0;
  r = r * x2 - 1.0 / 39916800.0;
  r = r * x2 + 1.0 / 362880.0;
  r = r * x2 - 1.0 / 5040.0;
  r = r * x2 + 1.0 / 120.0;
  r = r * x2 - 1.0 / 6.0;
  return x + x * x2 * r;
}
constexpr double ct_cos_poly(double x) {
  const double x2 = x * x;
  double r = 1.0 / 6402373705728000.0;  // 1 / 18!
  r = r * x2 - 1.0 / 20922789888000.0;
  r = r * x2 + 1.0 / 87178291200.0;
  r = r * x2 - 1.0 / 479001600.0;
  r = r * x2 + 1.0 / 3628800.0;
  r = r * x2 - 1.0 / 40320.0;
  r = r * x2 + 1.0 / 720.0;
  r = r * x2 - 1.0 / 24.0;
  r = r * x2 + 0.5;
  return 1.0 - x2 * r;
}
template <int R> struct RootTab { double c[R], s[R]; };
template <int R> constexpr RootTab<R> root_tab() {
  RootTab<R> t{};
  for (int e = 0; e < R; ++e) {
    // angle = 2 pi e / R = quarter * pi/2 + (pi/2) * rem / R, the last term folded into [0, pi/4] by cos <-> sin
    const int quarter = (4 * e) / R, rem = 4 * e - quarter * R;
    const bool fold = 2 * rem > R;
    const double x = 1.5707963267948966192 * (double)(fold ? R - rem : rem) / (double)R;
    const double cx = fold ? ct_sin_poly(x) : ct_cos_poly(x), sx = fold ? ct_cos_poly(x) : ct_sin_poly(x);
    t.c[e] = quarter == 0 ? cx : quarter == 1 ? -sx : quarter == 2 ? -cx : sx;
    t.s[e] = quarter == 0 ? sx : quarter == 1 ? cx : quarter == 2 ? -sx : -cx;
  }
  return t;
}
// z * exp(-2 pi i e / R); e is a constant wherever this is called (fully unrolled loops): the branches fold
template <typename T, int R> __device__ __forceinline__ cpx<T> mul_root(cpx<T> z, int e) {
  constexpr RootTab<R> tab = root_tab<R>();
  e %= R;
  if (e == 0) return z;
  if (4 * e == R) return {z.im, -z.re};
  if (2 * e == R) return {-z.re, -z.im};
  if (4 * e == 3 * R) return {-z.im, z.re};
  const T c = (T)tab.c[e], s = (T)tab.s[e];
  return {z.re * c + z.im * s, z.im * c - z.re * s};
}

template <typename T> __device__ __forceinline__ void dft3(cpx<T>* x) {
  const T h = (T)0.86602540378443864676;  // sin(pi / 3)
  const cpx<T> t = {x[1].re + x[2].re, x[1].im + x[2].im}, d = {x[1].re - x[2].re, x[1].im - x[2].im};
  const cpx<T> m = {x[0].re - (T)0.5 * t.re, x[0].im - (T)0.5 * t.im};
  x[0] = {x[0].re + t.re, x[0].im + t.im};
  x[1] = {m.re + h * d.im, m.im - h * d.re};  // m - i h d
  x[2] = {m.re - h * d.im, m.im + h * d.re};  // m + i h d
}

// the leaf a composite length is split by first: the largest register butterfly that divides it
constexpr int reg_leaf(int r) {
  for (int p : {16, 8, 7, 5, 4, 3, 2, 13, 11})
    if (r % p == 0) return p;
  return r;
}
constexpr bool reg_is_leaf(int r) { return r == 1 || r == 2 || r == 3 || r == 4 || r == 5 || r == 7 || r == 8 || r == 11 || r == 13 || r == 16 || r == 32; }
// forward DFT of R points in registers, natural order in and out: leaves, or one Cooley-Tukey step R = P x Q around them --
// n = Q*n1 + n2, k = k1 + P*k2: X[k1 + P*k2] = sum_n2 W_Q^{n2*k2} W_R^{n2*k1} sum_n1 x[Q*n1 + n2] W_P^{n1*k1}
template <typename T, int R> __device__ __forceinline__ void dft_any(cpx<T>* x) {
  if constexpr (R == 1) {
  } else if constexpr (R == 2 || R == 4 || R == 8 || R == 16 || R == 32) {
    dft_r<T, R>(x);
  } else if constexpr (R == 3) {
    dft3(x);
  } else if constexpr (R == 5 || R == 7 || R == 11 || R == 13) {
    dft_prime<T, R>(x, true);
  } else {
    constexpr int P = reg_leaf(R), Q = R / P;
    static_assert(P < R, "dft_any: a prime factor above 13");
    cpx<T> a[Q][P];
#pragma unroll
    for (int n2 = 0; n2 < Q; ++n2) {
      cpx<T> t[P];
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) t[n1] = x[Q * n1 + n2];
      dft_any<T, P>(t);
#pragma unroll
      for (int k1 = 0; k1 < P; ++k1) a[n2][k1] = mul_root<T, R>(t[k1], n2 * k1);
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
      cpx<T> t[Q];
#pragma unroll
      for (int n2 = 0; n2 < Q; ++n2) t[n2] = a[n2][k1];
      dft_any<T, Q>(t);
#pragma unroll
      for (int k2 = 0; k2 < Q; ++k2) x[k1 + P * k2] = t[k2];
    }
  }
}

// Cache policy: plain loads and stores everywhere.  Streaming hints lose 1 ... 40 % (loads; stores 1 ... 12 %) -- most where row segments straddle
// 128-byte lines and neighbouring tiles meet in the L2 (44100 = 210 x 210: -34 % / -10 % / both -40 %; profiles/r06_s30_regtile_policy_ab.jsonl)
// one complex number (8 / 16 bytes, 8-byte aligned: an f64 transform of odd length inside a batch) as ONE global access
template <typename T> __device__ __forceinline__ void store_cpx(cpx<T>* p, cpx<T> z) {
  if constexpr (sizeof(T) == 8) {
    Unit16<T> u;
    u.a[0] = z.re; u.a[1] = z.im;
    store_unit_a8<T>(p, u);
  } else {
    *p = z;
  }
}

// Threads work on 16-byte UNITS of a row segment: one f64 column, two adjacent f32 columns (VEC) -- every global and LDS access of the
// stages is 16 bytes per lane (8-byte accesses for f32, one column per thread: 10 - 25 % slower, profiles/r06_s27_*first_version*).
template <typename T, uint32_t L> struct RegTileCfg {  // the rules: reg_tile_shape (mixed_schedule.h), shared with the host
  static constexpr RegTileShape S = reg_tile_shape(L, (uint32_t)sizeof(cpx<T>));
  static constexpr uint32_t R1 = S.r1, R2 = S.r2, COLS = S.cols, NT = S.threads, LDO = S.ldo;
  static constexpr uint32_t VEC = 16 / (uint32_t)sizeof(cpx<T>), CU = COLS / VEC;  // units per row segment
  static constexpr uint32_t XSU = R2 * CU;                                          // units between the k1 planes of the exchange buffer
  static constexpr size_t TAB_OFF = S.tab_off, SMEM = S.smem;
};
// (no register bound in the launch bounds: hipcc spreads these kernels over up to 256 registers, two workgroups per CU where the LDS would hold
// three -- bounds that aim at three or four spill in 79 / 120 of the 560 instantiations and lose up to 67 %, profiles/r06_s34_regtile_budget_ab.jsonl)

// the two halves of an inter-pass twiddle table build (W_size^{i * k} from the two-level tables, k = r for r < RA, RA * (r - RA) beyond): the
// global loads first, the products and LDS writes once the data loads are under way
template <typename T, uint32_t RA, uint32_t RB, uint32_t COLS, uint32_t NT> struct RegTileTabs {
  static constexpr uint32_t TITER = (COLS * (RA + RB) + NT - 1) / NT;
  cpx<T> tlo[TITER], thi[TITER];
  __device__ __forceinline__ void load(const TiledArgs& a, uint32_t tid, uint32_t tcols, bool per_column, uint32_t c0, uint32_t ncols_total, uint32_t i_row) {
    const cpx<T>* lo = (const cpx<T>*)a.tw_lo;
    const cpx<T>* hi = (const cpx<T>*)a.tw_hi;
    const uint32_t mask = (1u << a.lo_bits) - 1u;
#pragma unroll
    for (uint32_t it = 0; it < TITER; ++it) {
      const uint32_t e = tid + it * NT;
      if (e < tcols * (RA + RB)) {
        const uint32_t tc = e / (RA + RB), r = e - tc * (RA + RB);
        // (a masked column of a ragged last tile takes the last valid column's entries: its own index would reach past the tables)
        const uint64_t i = per_column ? (uint64_t)(c0 + tc < ncols_total ? c0 + tc : ncols_total - 1u) : (uint64_t)i_row;
        const uint64_t ex = i * (uint64_t)(r < RA ? r : RA * (r - RA));  // i * k < size
        tlo[it] = lo[ex & mask];
        thi[it] = hi[ex >> a.lo_bits];
      }
    }
  }
  __device__ __forceinline__ void store(uint32_t tid, uint32_t tcols, cpx<T>* tu, cpx<T>* tv) const {
#pragma unroll
    for (uint32_t it = 0; it < TITER; ++it) {
      const uint32_t e = tid + it * NT;
      if (e < tcols * (RA + RB)) {
        const uint32_t tc = e / (RA + RB), r = e - tc * (RA + RB);
        const cpx<T> v = cmul(tlo[it], thi[it]);
        if (r < RA) tu[tc * RA + r] = v; else tv[tc * RB + (r - RA)] = v;
      }
    }
  }
};
// a first pass's output: the tile's ncols * L elements, staged as [c][k] at c * LDO + k, are ONE contiguous run of global memory
template <typename T, uint32_t L, uint32_t COLS, uint32_t NT, uint32_t LDO>
__device__ __forceinline__ void reg_tile_copy_out(const cpx<T>* buf, cpx<T>* o, uint32_t ncols, uint32_t tid) {
  // one element per lane and instruction (f32: 8 bytes -- a wave still writes 512 contiguous bytes, and the LDS reads are at stride 1:
  // pairs of elements per lane read two 8-byte words at a 16-byte stride, a 2-way bank conflict, SQ_LDS_BANK_CONFLICT 41 % at L = 400)
  const uint32_t total = ncols * L;
  constexpr uint32_t OITER = (L * COLS + NT - 1) / NT;
  cpx<T> v[OITER];
#pragma unroll
  for (uint32_t it = 0; it < OITER; ++it) {
    const uint32_t idx = tid + it * NT, idc = idx < total ? idx : 0u, cc = idc / L, k = idc - cc * L;  // (clamped: no branch around the read)
    LDS_NOTE(buf + cc * LDO + k, sizeof(cpx<T>), false, 303);
    v[it] = buf[cc * LDO + k];
  }
#pragma unroll
  for (uint32_t it = 0; it < OITER; ++it) {
    const uint32_t idx = tid + it * NT;
    if (idx < total) store_cpx(o + idx, v[it]);
  }
}
// the pieces the three kernels share, on a thread's VEC columns
template <typename T, uint32_t L> struct RegTileOps {
  using C = RegTileCfg<T, L>;
  static constexpr uint32_t VEC = C::VEC, CU = C::CU, LDO = C::LDO;
  template <uint32_t R> static __device__ __forceinline__ void unpack(const Unit16<T>& u, cpx<T> (&x)[VEC][R], uint32_t r) {
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
  }
  template <uint32_t R> static __device__ __forceinline__ Unit16<T> pack(const cpx<T> (&x)[VEC][R], uint32_t r) {
    Unit16<T> u;
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) { u.a[2 * v] = x[v][r].re; u.a[2 * v + 1] = x[v][r].im; }
    return u;
  }
  // exchange: the R values of a thread into planes [k][row][unit] (rows j2 and j2 ^ 1 exchanged in the odd planes where RROW is even:
  // reg_tile_row), and a plane's RROW values back
  template <uint32_t R, uint32_t RROW> static __device__ __forceinline__ void xch_write(Unit16<T>* bufu, const cpx<T> (&x)[VEC][R], uint32_t row, uint32_t cu, uint32_t site) {
#pragma unroll
    for (uint32_t k = 0; k < R; ++k) {
      Unit16<T>* d = bufu + k * (RROW * CU) + reg_tile_row(RROW, k, row) * CU + cu;
      LDS_NOTE(d, 16, true, site);
      *d = pack<R>(x, k);
    }
  }
  template <uint32_t RROW> static __device__ __forceinline__ void xch_read(const Unit16<T>* bufu, cpx<T> (&y)[VEC][RROW], uint32_t plane, uint32_t cu, uint32_t site) {
#pragma unroll
    for (uint32_t j = 0; j < RROW; ++j) {
      const Unit16<T>* s = bufu + plane * (RROW * CU) + reg_tile_row(RROW, plane, j) * CU + cu;
      LDS_NOTE(s, 16, false, site);
      unpack<RROW>(*s, y, j);
    }
  }
  // a first pass's staging: output k = q + RQ * k2 of column c at c * LDO + k
  template <uint32_t R, uint32_t RQ> static __device__ __forceinline__ void stage_write(cpx<T>* buf, const cpx<T> (&y)[VEC][R], uint32_t q, uint32_t cu) {
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v)
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2) {
        cpx<T>* d = buf + (cu * VEC + v) * LDO + q + RQ * k2;
        LDS_NOTE(d, sizeof(cpx<T>), true, 302);
        *d = y[v][k2];
      }
  }
};

template <typename T, uint32_t L, int IO = IO_PLAIN>
__global__ void __launch_bounds__((RegTileCfg<T, L>::NT)) tiled_reg_kernel(TiledArgs a) {
  using C = RegTileCfg<T, L>;
  using O = RegTileOps<T, L>;
  constexpr uint32_t R1 = C::R1, R2 = C::R2, COLS = C::COLS, NT = C::NT, LDO = C::LDO, VEC = C::VEC, CU = C::CU;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>);
  static_assert(R1 * R2 == L && R1 >= R2, "tiled_reg_kernel: split");
  FOURIER_DYN_SMEM(smem);
  cpx<T>* buf = (cpx<T>*)smem;                // exchange planes (units), then -- first pass -- the staging [c][k] at c * LDO + k
  Unit16<T>* bufu = (Unit16<T>*)smem;
  cpx<T>* tu = (cpx<T>*)(smem + C::TAB_OFF);  // [COLS][R1]: W_size^{i * k1}
  cpx<T>* tv = tu + COLS * R1;                // [COLS][R2]: W_size^{i * R1 * k2}
  const uint32_t tid = threadIdx.x;
  const uint32_t cu = tid % CU, q = tid / CU;  // stage A: q = j2 (< R2); stage B: q = k1 (< R1)
  const uint32_t c = cu * VEC;                 // first column of the thread's unit
  const bool first = (a.s == 1);
  // tile coordinates: block -> (transform b, i, first column c0), as in tiled_mixed_kernel_ct
  const uint32_t tiles_per_row = (uint32_t)a.tiles_per_row;
  const uint32_t rows = first ? 1u : (uint32_t)a.m;
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk);
  const uint32_t b = blk / (tiles_per_row * rows), rem = blk - b * (tiles_per_row * rows);
  const uint32_t i_row = rem / tiles_per_row, c0 = (rem - i_row * tiles_per_row) * COLS;
  const uint32_t ncols_total = first ? (uint32_t)a.m : (uint32_t)a.s;
  const uint32_t ncols = ncols_total - c0 < COLS ? ncols_total - c0 : COLS;
  // global accesses through bounds-checked descriptors over ONE transform (the Bluestein end passes: over the blu_n points of the user
  // array -- the zero padding behind it and the outputs beyond it need no branch): a unit that reaches past a ragged tile's last column
  // loads what lies there (the next row; zero past the end of the transform) and is never stored
  const uint64_t in_len = IO == IO_BLU_IN ? a.blu_n : a.n, out_len = IO == IO_BLU_OUT ? a.blu_n : a.n;
  const cpx<T>* in_b = (const cpx<T>*)a.in + (uint64_t)b * in_len;
  cpx<T>* out_b = (cpx<T>*)a.out + (uint64_t)b * out_len;
  const BufRsrc rin = make_rsrc(in_b, (uint32_t)(in_len * EB));
  const uint64_t row_stride = a.s * a.m;
  const uint64_t col0 = first ? (uint64_t)c0 : (uint64_t)c0 + a.s * (uint64_t)i_row;
  const bool twiddled = a.m > 1;
  const bool full = c + VEC <= ncols, part = !full && c < ncols;  // part: f32, the last valid column of a ragged tile by itself

  // ---- stage A: rows j2 + R2*j1 of the unit, then every other global load of the thread ahead of its first use
  cpx<T> x[VEC][R1];
  if (q < R2) {
    const uint32_t voff = (uint32_t)((col0 + c + row_stride * (uint64_t)q) * EB), rowb = (uint32_t)(row_stride * (uint64_t)R2 * EB);
    // (chirp-in: 2N <= M + 1, so every row from L / 2 + 1 on is padding -- rows q + R2 * j1 with R2 * j1 beyond that are compile-time zeros,
    // never loaded and folded out of the first transform; the descriptor zeroes what is left of the padding)
    auto pad = [](uint32_t j1) { return IO == IO_BLU_IN && R2 * j1 >= L / 2 + 1; };
    Unit16<T> d[R1];
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1)
      if (!pad(j1)) d[j1] = buf_load_unit<T>(rin, voff + j1 * rowb);
    if constexpr (IO == IO_BLU_IN) {
      // work = x (.) in, zero padded (bluesteins.rs:229-234): element e = column + m * row of the user array and of the chirp table
      const BufRsrc rc = make_rsrc(a.blu_x, (uint32_t)(a.blu_n * EB));
      Unit16<T> ch[R1];
#pragma unroll
      for (uint32_t j1 = 0; j1 < R1; ++j1)
        if (!pad(j1)) ch[j1] = buf_load_unit<T>(rc, voff + j1 * rowb);
#pragma unroll
      for (uint32_t j1 = 0; j1 < R1; ++j1)
#pragma unroll
        for (uint32_t v = 0; v < VEC; ++v) {
          if (pad(j1)) { x[v][j1] = cpx<T>{(T)0, (T)0}; continue; }
          cpx<T> val{d[j1].a[2 * v], d[j1].a[2 * v + 1]};
          if (a.blu_swap) val = {val.im, val.re};
          x[v][j1] = cmul(cpx<T>{ch[j1].a[2 * v], ch[j1].a[2 * v + 1]}, val);
        }
    } else {
#pragma unroll
      for (uint32_t j1 = 0; j1 < R1; ++j1) O::template unpack<R1>(d[j1], x, j1);
    }
  }
  // the twiddle between the stages, W_L^{j2 * k1}, applied on the side with fewer values per thread (stage B: R2 <= R1), loaded with the
  // data (issued before the barrier instead, its latency is exposed in every tile: f64 -25 %, profiles/r06_s23_regtile_remap_ab.jsonl)
  cpx<T> w[R2];
  if (q < R1) {
    const cpx<T>* tw = (const cpx<T>*)a.tw + q * R2;  // [k1][j2]
#pragma unroll
    for (uint32_t j2 = 1; j2 < R2; ++j2) w[j2] = tw[j2];
  }
  RegTileTabs<T, R1, R2, COLS, NT> tabs;
  const uint32_t tcols = first ? COLS : 1u;  // the later passes have one i for the whole tile: table column 0
  if (twiddled) tabs.load(a, tid, tcols, first, c0, ncols_total, i_row);
  Unit16<T> chq[R2];  // chirp-out: the chirp at the thread's outputs, needed after stage B's transform (f64: loaded before the barrier, f32: with the data -- see the conv kernel)
  auto load_chq = [&]() {
    if constexpr (IO == IO_BLU_OUT) {
      if (q < R1) {
        const BufRsrc rc = make_rsrc(a.blu_x, (uint32_t)(a.blu_n * EB));
        const uint32_t voff = (uint32_t)(((uint64_t)c0 + c + a.s * (uint64_t)q) * EB), rowb = (uint32_t)(a.s * (uint64_t)R1 * EB);
#pragma unroll
        for (uint32_t k2 = 0; k2 < R2; ++k2)
          if (R1 * k2 < L / 2 + 1) chq[k2] = buf_load_unit<T>(rc, voff + k2 * rowb);  // (outputs k = q + R1 * k2 from L / 2 + 1 on lie beyond the user array)
      }
    }
  };
  if constexpr (sizeof(T) == 4) load_chq();
  if (q < R2) {
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) {
      if (a.swap_in) {
#pragma unroll
        for (uint32_t j1 = 0; j1 < R1; ++j1) x[v][j1] = {x[v][j1].im, x[v][j1].re};
      }
      dft_any<T, (int)R1>(x[v]);
    }
    O::template xch_write<R1, R2>(bufu, x, q, cu, 300);
  }
  if (twiddled) tabs.store(tid, tcols, tu, tv);
  if constexpr (sizeof(T) == 8) load_chq();
  __syncthreads();

  // ---- stage B: the R2 values of (unit, k1 = q); output k = k1 + R1*k2
  cpx<T> y[VEC][R2];
  if (q < R1) {
    O::template xch_read<R2>(bufu, y, q, cu, 301);
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) {
#pragma unroll
      for (uint32_t j2 = 1; j2 < R2; ++j2) y[v][j2] = cmul(y[v][j2], w[j2]);
      dft_any<T, (int)R2>(y[v]);
      if (twiddled) {
        const uint32_t tc = first ? c + v : 0u;
        const cpx<T> u = tu[tc * R1 + q];
#pragma unroll
        for (uint32_t k2 = 0; k2 < R2; ++k2) y[v][k2] = cmul(y[v][k2], k2 == 0 ? u : cmul(u, tv[tc * R2 + k2]));
      } else {  // last pass (mod.rs:238: no twiddle): the user-level scaling and the inverse's trailing swap
        const T scale = (T)a.scale;
#pragma unroll
        for (uint32_t k2 = 0; k2 < R2; ++k2) {
          if (a.swap_out) y[v][k2] = {y[v][k2].im, y[v][k2].re};
          y[v][k2] = {y[v][k2].re * scale, y[v][k2].im * scale};
        }
      }
    }
  }
  if (!first) {
    // out[j + L*s*i + s*k]: 128-byte row segments, row k at stride s.  Chirp-out (the last pass: i = 0): out = work (.) x (.) scale, the
    // first blu_n points only (bluesteins.rs:240-258) -- y was swapped and scaled above (the inverse inner transform's trailing swap),
    // the user-level inverse swaps once more; the descriptor drops what lies beyond the user array
    if (q < R1 && (full || part)) {
      const BufRsrc rout = make_rsrc(out_b, (uint32_t)(out_len * EB));
      const uint32_t voff = (uint32_t)(((uint64_t)c0 + c + (uint64_t)L * a.s * (uint64_t)i_row + a.s * (uint64_t)q) * EB);
      const uint32_t rowb = (uint32_t)(a.s * (uint64_t)R1 * EB);
#pragma unroll
      for (uint32_t k2 = 0; k2 < R2; ++k2) {
        if constexpr (IO == IO_BLU_OUT) {
          if (R1 * k2 >= L / 2 + 1) continue;  // 2N <= M + 1: beyond the user array, never stored
#pragma unroll
          for (uint32_t v = 0; v < VEC; ++v) {
            cpx<T> z = cmul(y[v][k2], cpx<T>{chq[k2].a[2 * v], chq[k2].a[2 * v + 1]});
            if (a.blu_swap) z = {z.im, z.re};
            y[v][k2] = z;
          }
        }
        if (full) buf_store_unit<T>(rout, voff + k2 * rowb, O::template pack<R2>(y, k2));
        else buf_store_elem<T>(rout, voff + k2 * rowb, y[0][k2]);
      }
    }
    return;
  }
  // first pass: out[L*i + k] -- the tile's output is ONE contiguous run of ncols * L elements; transposed through LDS
  __syncthreads();  // every thread has read its stage-B inputs
  if (q < R1) O::template stage_write<R2, R1>(buf, y, q, cu);
  __syncthreads();
  reg_tile_copy_out<T, L, COLS, NT, LDO>(buf, out_b + (uint64_t)L * c0, ncols, tid);
}

// Bluestein middle sweep on a smooth M = L1 x L: the LAST pass of the forward inner transform (length L, stride s = M / L, no twiddle:
// mod.rs:238), the pointwise product with w (bluesteins.rs:236-238) and the FIRST pass of the inverse inner transform (length L again, m = s:
// it reads element (column j, row k) exactly where the forward pass put it) in one launch -- the tile never leaves the CU in between.
// The second transform runs the split the other way round (its input index k = k1 + R1 * k2 IS what stage B's threads hold): stage A' =
// DFT_R2 on stage B's threads, exchange, stage B' = DFT_R1 on stage A's threads, output k'' = k1'' + R2 * k2''; the twiddle between its
// stages, W_L^{k1 * k1''}, is the first transform's table again.  Inverse = swap . DFT . swap: the swap after the product here, the
// trailing one in the chirp-out pass.
template <typename T, uint32_t L>
__global__ void __launch_bounds__((RegTileCfg<T, L>::NT)) tiled_reg_conv_kernel(TiledArgs a) {
  using C = RegTileCfg<T, L>;
  using O = RegTileOps<T, L>;
  constexpr uint32_t R1 = C::R1, R2 = C::R2, COLS = C::COLS, NT = C::NT, LDO = C::LDO, VEC = C::VEC, CU = C::CU;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>);
  FOURIER_DYN_SMEM(smem);
  cpx<T>* buf = (cpx<T>*)smem;
  Unit16<T>* bufu = (Unit16<T>*)smem;
  cpx<T>* tu = (cpx<T>*)(smem + C::TAB_OFF);  // [COLS][R2]: W_M^{i * k1''}
  cpx<T>* tv = tu + COLS * R2;                // [COLS][R1]: W_M^{i * R2 * k2''}
  const uint32_t tid = threadIdx.x;
  const uint32_t cu = tid % CU, q = tid / CU, c = cu * VEC;
  const uint32_t tiles_per_row = (uint32_t)a.tiles_per_row;  // ceil(s / COLS)
  const uint32_t blk = xcd_chunked(blockIdx.x, gridDim.x, a.xcd_chunk);
  const uint32_t b = blk / tiles_per_row, c0 = (blk - b * tiles_per_row) * COLS;
  const uint32_t ncols_total = (uint32_t)a.s;
  const uint32_t ncols = ncols_total - c0 < COLS ? ncols_total - c0 : COLS;
  const cpx<T>* in_b = (const cpx<T>*)a.in + (uint64_t)b * a.n;
  cpx<T>* out_b = (cpx<T>*)a.out + (uint64_t)b * a.n;
  const BufRsrc rin = make_rsrc(in_b, (uint32_t)(a.n * EB)), rw = make_rsrc(a.blu_w, (uint32_t)(a.n * EB));

  // ---- forward last pass, stage A: rows j2 + R2*j1 of the unit; then every other global load of the thread
  cpx<T> x[VEC][R1];
  if (q < R2) {
    const uint32_t voff = (uint32_t)(((uint64_t)c0 + c + a.s * (uint64_t)q) * EB), rowb = (uint32_t)(a.s * (uint64_t)R2 * EB);
    Unit16<T> d[R1];
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) d[j1] = buf_load_unit<T>(rin, voff + j1 * rowb);
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) O::template unpack<R1>(d[j1], x, j1);
  }
  cpx<T> w[R2];      // W_L^{k1 * j2}: the twiddle between the stages of both transforms
  Unit16<T> ww[R2];  // w[j + s * k] at the thread's outputs k = q + R1 * k2
  if (q < R1) {
    const cpx<T>* tw = (const cpx<T>*)a.tw + q * R2;
#pragma unroll
    for (uint32_t j2 = 1; j2 < R2; ++j2) w[j2] = tw[j2];
  }
  // w is needed after stage B's transform.  f64: loaded before the barrier, not with the data -- R2 units (4 registers each) fewer held through
  // stage A, the conv kernel of 324 points no longer spills into AGPRs (N = 75011: +4 -> +28 % over the power-of-two route); f32: with the data
  // (late: -3 ... 7 % at the long tiles) -- profiles/r06_s35_smooth_m_late_loads_ab.jsonl against r06_s32_*
  constexpr bool LATE = sizeof(T) == 8;
  auto load_w = [&]() {
    if (q < R1) {
      const uint32_t voff = (uint32_t)(((uint64_t)c0 + c + a.s * (uint64_t)q) * EB), rowb = (uint32_t)(a.s * (uint64_t)R1 * EB);
#pragma unroll
      for (uint32_t k2 = 0; k2 < R2; ++k2) ww[k2] = buf_load_unit<T>(rw, voff + k2 * rowb);
    }
  };
  if constexpr (!LATE) load_w();
  RegTileTabs<T, R2, R1, COLS, NT> tabs;  // the inverse first pass: i = c0 + column, output k'' = k1'' + R2 * k2''
  tabs.load(a, tid, COLS, true, c0, ncols_total, 0u);
  if (q < R2) {
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) dft_any<T, (int)R1>(x[v]);
    O::template xch_write<R1, R2>(bufu, x, q, cu, 310);
  }
  tabs.store(tid, COLS, tu, tv);
  if constexpr (LATE) load_w();
  __syncthreads();

  // ---- stage B, (.) w, swap; stage A' of the inverse first pass on the same threads
  cpx<T> y[VEC][R2];
  if (q < R1) {
    O::template xch_read<R2>(bufu, y, q, cu, 311);
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) {
#pragma unroll
      for (uint32_t j2 = 1; j2 < R2; ++j2) y[v][j2] = cmul(y[v][j2], w[j2]);
      dft_any<T, (int)R2>(y[v]);  // y[k2] = Y[q + R1 * k2]
#pragma unroll
      for (uint32_t k2 = 0; k2 < R2; ++k2) {
        const cpx<T> z = cmul(y[v][k2], cpx<T>{ww[k2].a[2 * v], ww[k2].a[2 * v + 1]});
        y[v][k2] = {z.im, z.re};
      }
      dft_any<T, (int)R2>(y[v]);  // over k2: y[k1''], k1'' < R2
#pragma unroll
      for (uint32_t k = 1; k < R2; ++k) y[v][k] = cmul(y[v][k], w[k]);  // W_L^{q * k1''}
    }
  }
  __syncthreads();  // every thread has read its stage-B inputs
  if (q < R1) O::template xch_write<R2, R1>(bufu, y, q, cu, 312);  // planes k1'' (< R2), rows k1 (< R1)
  __syncthreads();
  // ---- stage B' (unit, k1'' = q < R2): the R1 values over k1, DFT_R1, the inter-pass twiddle W_M^{i * (k1'' + R2 * k2'')}
  cpx<T> z[VEC][R1];
  if (q < R2) {
    O::template xch_read<R1>(bufu, z, q, cu, 313);
#pragma unroll
    for (uint32_t v = 0; v < VEC; ++v) {
      dft_any<T, (int)R1>(z[v]);
      const cpx<T> u = tu[(c + v) * R2 + q];
#pragma unroll
      for (uint32_t k2 = 0; k2 < R1; ++k2) z[v][k2] = cmul(z[v][k2], k2 == 0 ? u : cmul(u, tv[(c + v) * R1 + k2]));
    }
  }
  __syncthreads();
  if (q < R2) O::template stage_write<R1, R2>(buf, z, q, cu);
  __syncthreads();
  reg_tile_copy_out<T, L, COLS, NT, LDO>(buf, out_b + (uint64_t)L * c0, ncols, tid);
}

}  // namespace fourier_hip
