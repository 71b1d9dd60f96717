// kernels_onelaunch.h -- plans that keep a whole transform inside one workgroup: both Stockham passes of 2^11..2^15 in one
// launch, and the whole Bluestein chirp-z (bluesteins.rs:215-259) in one launch for M <= 2^15.
#pragma once
#include "kernels_pass.h"

FOURIER_KERNELS_BEGIN

// register rows of a one-launch chirp-z kernel that carry user data: 8 of 16 (2n <= M: the upper half of the work array is padding on
// the way in and beyond the user array on the way out; see bluestein_rows_kernel; all 16 until round 5: 7 - 19 % slower)
constexpr int BLU_ROWS = 8;

// ---- mid sizes N = L1 x L2 <= 2^15 (f32) / 2^14 (f64): BOTH Stockham passes in one launch ----
// One workgroup owns one whole transform in registers (N/16 points per ... 16 points x VEC per thread),
// so HBM sees it once in and once out instead of twice: pass A = column FFT of length L1 over the
// L1 x L2 matrix + twiddle W_N^{i*k1} (mod.rs:203-284 with R = L1, s = 1), an in-LDS transpose instead of
// the HBM round trip, pass B = column FFT of length L2 over the L2 x L1 matrix (R = L2, s = L1).
// L1, L2 in {64, 128, 256}: radix 16 x (L/16), two stages each.
template <typename T, int L, int CG>
__device__ __forceinline__ void two_stage_fft(RegTile<T, L, CG>& x, int th, int cg, unsigned char* smem, const cpx<T>* tw1,
                                              unsigned site) {
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, R2 = C::R2;
  static_assert(C::R3 == 1 && Q > 1, "two_stage_fft: 32 <= L <= 256");
#pragma unroll
  for (int v = 0; v < VEC; ++v) dft16(x[v]);
  FOURIER_SCHED_FENCE();
  stage_twiddle<T, VEC>(x, tw1 + th * 16);
  lds_exchange<T, L, CG, 0>(x, smem, cg, [=](int r) { return 16 * th + r; }, th, cg, site);
  FOURIER_SCHED_FENCE();
  constexpr int NB2 = 16 / R2;
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int u = 0; u < NB2; ++u) {
      cpx<T> t[R2];
#pragma unroll
      for (int k = 0; k < R2; ++k) t[k] = x[v][u + NB2 * k];
      dft_r<T, R2>(t);
#pragma unroll
      for (int k = 0; k < R2; ++k) x[v][u + NB2 * k] = t[k];
    }
}

// LDS layout of the in-workgroup transpose between the two passes of a one-launch plan: element (row i of the L2 x L1
// matrix, column k1) lives in unit (k1 / VEC) * (L2 + 1) + pi(i), pi(i) = i / VEC + (i % VEC) * (L2 / VEC) -- column-group
// major, one unit of padding per column group, rows de-interleaved by parity.  A writer's lanes walk the rows i = cg*VEC + v
// at a fixed k1 and v: adjacent units after pi (2-way on ds_write_b32 = free; row-major, or column-group-major without
// pi, put them on 8 of the 32 banks: 4-way, SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles of the 2^14 / 2^15 kernels in
// profiles/r03_s15_sq_breakdown.json).  A reader's lanes walk the column groups at a fixed row: stride L2 + 1 units, an
// odd number of 8-byte bank pairs, conflict-free for ds_read_b64 / b128.
template <int L2, int VEC> __device__ __forceinline__ constexpr int twolevel_tr_unit(int row, int colgroup) {
  return colgroup * (L2 + 1) + row / VEC + (row % VEC) * (L2 / VEC);
}
template <typename T, int L1, int L2> struct TwolevelTr {
  static constexpr int VEC = 16 / (2 * (int)sizeof(T));
  static constexpr bool SPLIT = TileCfg<T, L2, L1 / VEC>::SPLIT;
  static constexpr size_t BYTES = (size_t)(L1 / VEC) * (L2 + 1) * (SPLIT ? 8 : 16);
};

// loads of the inter-pass twiddle table in flight per thread (2-wave workgroups live on occupancy: stay under 128 VGPRs)
constexpr int twolevel_tw_batch(int nt) { return nt <= 128 ? 4 : 8; }
// Both passes of an N = L1 x L2 transform on register-resident data.  In: thread (th = tid / CG1,
// cg = tid % CG1) holds rows th + Q1*r of the L1 x L2 row-major matrix (element row*L2 + col), columns
// cg*VEC + v.  Out: thread (th2 = tid / CG2, cg2 = tid % CG2) holds X[k1 + L1*k2] for k2 = th2 + Q2*r,
// k1 = cg2*VEC + v -- i.e. exactly the input layout of an L2 x L1 problem, so the core can be chained.
template <typename T, int L1, int L2>
__device__ __forceinline__ void twolevel_core(cpx<T> (*x)[16], int tid, unsigned char* smem, const cpx<T>* tw1_a,
                                              const cpx<T>* tw1_b, const cpx<T>* tw_full, unsigned site) {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  constexpr int CG1 = L2 / VEC, CG2 = L1 / VEC, Q1 = L1 / 16, Q2 = L2 / 16;
  using CB = TileCfg<T, L2, CG2>;
  static_assert(Q1 * CG1 == Q2 * CG2, "same thread count in both phases");
  typedef cpx<T> Regs[VEC][16];
  Regs& xr = *reinterpret_cast<Regs*>(x);
  const int th = tid / CG1, cg = tid % CG1;
  two_stage_fft<T, L1, CG1>(xr, th, cg, smem, tw1_a, site);
  // register r now holds k1 = th + Q1*r of column i: inter-pass twiddle W_N^{i*k1}.  N <= 2^15, so the
  // full table (the reference's per-pass layout idea, mod.rs:24-46) is kept, stored [k1][i] so that a
  // thread reads it with the same coalesced 16-byte units as the data; it stays L2-resident.
  {
    const BufRsrc rt = make_rsrc(tw_full);
    const uint32_t voff = (uint32_t)((th * L2 + cg * VEC) * sizeof(cpx<T>));
    units_batched<T, twolevel_tw_batch(Q1 * CG1)>(
        [&](int r) { return buf_load_unit<T>(rt, voff, (uint32_t)((Q1 * r) * L2 * sizeof(cpx<T>))); },
        [&](int r, const Unit16<T>& u) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) xr[v][r] = cmul(xr[v][r], cpx<T>{u.a[2 * v], u.a[2 * v + 1]});
        });
  }
  // ---- transpose through LDS: element (i, k1) -> row i, column k1 of the L2 x L1 matrix
  int tb = tid;
  FOURIER_LAUNDER(tb);  // phase B's mapping is derived here, not at the top of the kernel (see tile_core)
  const int th2 = tb / CG2, cg2 = tb % CG2;
  {
    constexpr bool SPLIT = CB::SPLIT;
    __syncthreads();  // the reads of phase A's exchange are done
#pragma unroll
    for (int plane = 0; plane < (SPLIT ? 2 : 1); ++plane) {
      if (plane == 1) __syncthreads();
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int i = cg * VEC + v;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k1 = th + Q1 * r;
          const int unit = twolevel_tr_unit<L2, VEC>(i, k1 / VEC);
          if constexpr (SPLIT) {
            T* p = (T*)(smem + (size_t)unit * 8) + (k1 % VEC);
            LDS_NOTE(p, sizeof(T), true, site + 8 + plane);
            *p = plane ? xr[v][r].im : xr[v][r].re;
          } else {
            cpx<T>* p = (cpx<T>*)(smem + (size_t)unit * 16) + (k1 % VEC);
            LDS_NOTE(p, 2 * sizeof(T), true, site + 8);
            *p = xr[v][r];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int unit = twolevel_tr_unit<L2, VEC>(th2 + Q2 * r, cg2);
        if constexpr (SPLIT) {
          const Unit8<T>* p = (const Unit8<T>*)smem + unit;
          LDS_NOTE(p, 8, false, site + 10 + plane);
          const Unit8<T> u = *p;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            if (plane) xr[v][r].im = u.a[v]; else xr[v][r].re = u.a[v];
          }
        } else {
          const Unit16<T>* p = (const Unit16<T>*)smem + unit;
          LDS_NOTE(p, 16, false, site + 10);
          const Unit16<T> u = *p;
#pragma unroll
          for (int v = 0; v < VEC; ++v) xr[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
        }
      }
    }
    __syncthreads();
  }
  // ---- phase B: rows i = th2 + Q2*r of the L2 x L1 matrix, columns k1 = cg2*VEC + v
  two_stage_fft<T, L2, CG2>(xr, th2, cg2, smem, tw1_b, site + 12);
}

// ---- whole Bluestein chirp-z in ONE launch for M = L <= 1024: COLS transforms per workgroup ----
// Same chain as bluestein_small_kernel with the row form of the in-tile FFT (tile_core, MODE_ROWS: natural order in and
// out, so two calls chain without a re-layout): x(.)in -> FFT_M -> (.)w -> swap -> FFT_M -> swap
// -> (.)x(.)scale; lane-contiguous 8/16-byte accesses to the N-point user arrays.
template <typename T, int L, int CG>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) bluestein_rows_kernel(PassArgs a) {
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, COLS = C::COLS;
  FOURIER_DYN_SMEM(smem);
  const int tid = (int)threadIdx.x;
  const uint32_t n = (uint32_t)a.blu_n;
  const uint64_t g0 = (uint64_t)blockIdx.x * COLS;
  // One descriptor over this workgroup's transforms (a ragged last workgroup ends where the batch ends: transforms
  // beyond it load as zero and are not stored), one over the chirp; branch-free element accesses, RB rows in flight.
  // Every phase derives its lane offsets from a laundered copy of the thread index (see tile_core).
  const uint64_t left = a.total_cols - g0;
  const uint32_t ncols = (uint32_t)(left < (uint64_t)COLS ? left : (uint64_t)COLS);
  const uint32_t nbytes = ncols * n * (uint32_t)sizeof(cpx<T>);
  const BufRsrc ri = make_rsrc((const cpx<T>*)a.in + g0 * a.blu_n, nbytes), ro = make_rsrc((cpx<T>*)a.out + g0 * a.blu_n, nbytes);
  const BufRsrc rc = make_rsrc(a.blu_x, n * (uint32_t)sizeof(cpx<T>));
  // rows per batch: RB chirp values and RB * VEC data elements in flight per thread (f32: every load of a phase in one batch; +2 ... 4 %,
  // profiles/r06_s5_chirpz_rows_tuning_ab.jsonl)
  constexpr int RB = (sizeof(T) == 8 && L < 1024) ? 4 : 8;
  constexpr uint32_t ES = (uint32_t)sizeof(cpx<T>);
  // M = L >= 2n - 1 and L even give n <= L/2 (bluesteins.rs:110; the engine checks it): positions th + Q*r with r >= 8 are padding on
  // the way in and beyond the user array on the way out.  Registers 8 .. 15 are therefore CONSTANT zeros here -- the first radix-16
  // stage of the forward transform folds to half its additions at compile time, half the loads are never issued -- and are neither
  // multiplied nor stored at the end, which lets the compiler drop half of the inverse transform's last stage (round 5)
  cpx<T> x[VEC][16];
  {
    const int th = tid % Q, cg = tid / Q;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int r = BLU_ROWS; r < 16; ++r) x[v][r] = cpx<T>{(T)0, (T)0};
#pragma unroll
    for (int r0 = 0; r0 < BLU_ROWS; r0 += RB) {
      cpx<T> c[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) c[q] = buf_load_elem<T>(rc, (uint32_t)(th + Q * (r0 + q)) * ES);  // 0 beyond n
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          // a position at or beyond n is padding (bluesteins.rs:229-234).  Its address (col*n + pos) belongs to the NEXT
          // transform's row: the offset is pushed out of the descriptor's range instead, so the load returns zero and a
          // transform never sees its neighbour's data (0 * Inf / 0 * NaN from the zero chirp would poison the column)
          const uint32_t pos = (uint32_t)(th + Q * (r0 + q));
          x[v][r0 + q] = buf_load_elem<T>(ri, pos < n ? ((uint32_t)(cg * VEC + v) * n + pos) * ES : 0xfffffff0u);
        }
      FOURIER_SCHED_FENCE();
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int q = 0; q < RB; ++q) {  // bluesteins.rs:229-234; positions n .. L-1 are padding: data and chirp both load as zero there
          cpx<T> val = x[v][r0 + q];
          if (a.blu_swap) val = {val.im, val.re};
          x[v][r0 + q] = cmul(c[q], val);
        }
      FOURIER_SCHED_FENCE();
    }
    int th_ = th, cg_ = cg;
    tile_core<T, L, CG, MODE_ROWS>(x, th_, cg_, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  }
  {
    int t = tid;
    FOURIER_LAUNDER(t);
    const cpx<T>* __restrict__ wt = (const cpx<T>*)a.mul + t % Q;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
      cpx<T> w[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) w[q] = wt[Q * (r0 + q)];
      FOURIER_SCHED_FENCE();
#pragma unroll
      for (int q = 0; q < RB; ++q)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const cpx<T> y = cmul(x[v][r0 + q], w[q]);
          x[v][r0 + q] = {y.im, y.re};
        }
      FOURIER_SCHED_FENCE();
    }
  }
  if constexpr (Q > 1) __syncthreads();
  {
    int t = tid;
    FOURIER_LAUNDER(t);
    int th_ = t % Q, cg_ = t / Q;
    tile_core<T, L, CG, MODE_ROWS>(x, th_, cg_, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  }
  const T scale = (T)a.scale;
  int t = tid;
  FOURIER_LAUNDER(t);
  const int th = t % Q, cg = t / Q;
#pragma unroll
  for (int r0 = 0; r0 < BLU_ROWS; r0 += RB) {  // (registers 8 .. 15 hold outputs beyond the user array)
    cpx<T> c[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) c[q] = buf_load_elem<T>(rc, (uint32_t)(th + Q * (r0 + q)) * ES);
    FOURIER_SCHED_FENCE();
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const uint32_t pos = (uint32_t)(th + Q * (r0 + q));
        cpx<T> y{x[v][r0 + q].im, x[v][r0 + q].re};
        y = cmul(y, c[q]);
        if (a.blu_swap) y = {y.im, y.re};
        // positions beyond n would land in the next transform's row: push them out of the descriptor's range instead
        buf_store_elem<T>(ro, pos < n ? ((uint32_t)(cg * VEC + v) * n + pos) * ES : 0xfffffff0u, cpx<T>{y.re * scale, y.im * scale});
      }
    FOURIER_SCHED_FENCE();
  }
}

#define FOURIER_TWOLEVEL_NT(T, L1, L2) ((L1 / 16) * (L2 / (16 / (2 * (int)sizeof(T)))))

template <typename T, int L1, int L2>
__global__ void __launch_bounds__(FOURIER_TWOLEVEL_NT(T, L1, L2), FOURIER_MIN_WAVES(FOURIER_TWOLEVEL_NT(T, L1, L2)))
    fft_twolevel_kernel(PassArgs a) {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  constexpr int CG1 = L2 / VEC, CG2 = L1 / VEC, Q1 = L1 / 16, Q2 = L2 / 16, N = L1 * L2;
  FOURIER_DYN_SMEM(smem);
  const int tid = (int)threadIdx.x;
  uint64_t blk = blockIdx.x;
  if (a.nxcd > 1) {
    const uint64_t nwg = gridDim.x, nx = a.nxcd, xcd = blk % nx, q = nwg / nx, r = nwg % nx;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blk / nx;
  }
  // one descriptor per transform, one 32-bit lane offset, the row offsets are compile-time scalars
  cpx<T>* const obase = (cpx<T>*)a.out + blk * N;
  const BufRsrc ri = make_rsrc((const cpx<T>*)a.in + blk * N), ro = make_rsrc(obase);
  (void)ro;
  cpx<T> x[VEC][16];
  {
    const int th = tid / CG1, cg = tid % CG1;
    const uint32_t voff = (uint32_t)((th * L2 + cg * VEC) * sizeof(cpx<T>));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // (streaming hints on both sides also where one 1024-thread workgroup has a CU to itself: without them 2^15 f32 / 2^14 f64 lose 7 - 8 %,
      // the shorter plans 2 - 4 % -- unlike the 16-column last pass of length 2048; profiles/r06_s13_one_launch_policy_ab.jsonl)
      const Unit16<T> u = buf_load_unit<T, BUF_NT>(ri, voff, (uint32_t)((Q1 * r) * L2 * sizeof(cpx<T>)));
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
    }
  }
  if (a.swap_in) {
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[v][r] = {x[v][r].im, x[v][r].re};
  }
  twolevel_core<T, L1, L2>(x, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2, (const cpx<T>*)a.tw_lo, 0);
  // register r now holds k2 = th2 + Q2*r: X[k1 + L1*k2]
  int tb = tid;
  FOURIER_LAUNDER(tb);
  const int th2 = tb / CG2, cg2 = tb % CG2;
  const T scale = (T)a.scale;
  const uint32_t voff = (uint32_t)((th2 * L1 + cg2 * VEC) * sizeof(cpx<T>));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    Unit16<T> u;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      cpx<T> y = x[v][r];
      if (a.swap_out) y = {y.im, y.re};
      u.a[2 * v] = y.re * scale; u.a[2 * v + 1] = y.im * scale;
    }
    buf_store_unit<T, BUF_NT>(make_rsrc(obase + (Q2 * r) * L1), voff, u);
  }
}

// ---- whole Bluestein chirp-z (bluesteins.rs:215-259) in ONE launch for M = L1 x L2 <= 2^15 ----
// work = x (.) in (zero padded to M) -> FFT_M -> (.) w -> IFFT_M -> (.) x (.) scale, all on the
// register-resident M-point array of one workgroup: the forward two-level core, then the same core with the
// roles of L1 and L2 exchanged (its input layout is the other's output layout).  HBM sees the N-point user
// array once in and once out; the tables (x: N, w: M, twiddles) stay L2-resident.
// (workgroups of up to 256 threads -- M = 2048 ... 8192 -- under a 128-register cap: four waves per SIMD instead of three at 130 registers, a few
// dwords of scratch: f32 +6 ... 12 %, f64 level; profiles/r06_s7_chirpz_small_occupancy_ab.jsonl)
#define FOURIER_BLU_SMALL_MIN_WAVES(NT) ((NT) <= 256 ? 4 : FOURIER_MIN_WAVES(NT))
template <typename T, int L1, int L2>
__global__ void __launch_bounds__(FOURIER_TWOLEVEL_NT(T, L1, L2), FOURIER_BLU_SMALL_MIN_WAVES(FOURIER_TWOLEVEL_NT(T, L1, L2)))
    bluestein_small_kernel(PassArgs a) {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  constexpr int CG1 = L2 / VEC, CG2 = L1 / VEC, Q1 = L1 / 16, Q2 = L2 / 16;
  FOURIER_DYN_SMEM(smem);
  const int tid = (int)threadIdx.x;
  uint64_t blk = blockIdx.x;
  if (a.nxcd > 1) {
    const uint64_t nwg = gridDim.x, nx = a.nxcd, xcd = blk % nx, q = nwg / nx, r = nwg % nx;
    blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + blk / nx;
  }
  // bounds-checked descriptors over this transform's user arrays and the chirp: everything at or beyond blu_n loads
  // as zero (the padding, bluesteins.rs:229-234) and is not stored (bluesteins.rs:240-258); no branches, and the
  // user rows of an odd-length f32 batch are only 8-byte aligned, which buffer_load/store_dwordx4 tolerate
  const uint32_t nbytes = (uint32_t)(a.blu_n * sizeof(cpx<T>));
  const BufRsrc ri = make_rsrc((const cpx<T>*)a.in + blk * a.blu_n, nbytes), ro = make_rsrc((cpx<T>*)a.out + blk * a.blu_n, nbytes);
  const BufRsrc rc = make_rsrc(a.blu_x, nbytes);
  const int th = tid / CG1, cg = tid % CG1;
  const uint32_t voff = (uint32_t)((th * L2 + cg * VEC) * sizeof(cpx<T>));
  constexpr uint32_t ROWB = (uint32_t)(Q1 * L2 * sizeof(cpx<T>));  // register r holds index (th + Q1*r)*L2 + cg*VEC + v
  // (2n <= M: registers 8 .. 15 -- indices from M/2 on -- are padding on the way in and beyond the user array on the way out: constant
  // zeros that fold the first radix-16 stage, never loaded, never multiplied by the chirp, never stored; see bluestein_rows_kernel)
  cpx<T> x[VEC][16];
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int r = BLU_ROWS; r < 16; ++r) x[v][r] = cpx<T>{(T)0, (T)0};
#pragma unroll
  for (int r = 0; r < BLU_ROWS; ++r) {
    const Unit16<T> u = buf_load_unit<T>(ri, voff + (uint32_t)r * ROWB);
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
  }
  units_batched_rows<T, 8, BLU_ROWS>([&](int r) { return buf_load_unit<T>(rc, voff + (uint32_t)r * ROWB); },
                      [&](int r, const Unit16<T>& c) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) {
                          cpx<T> val = x[v][r];
                          if (a.blu_swap) val = {val.im, val.re};
                          x[v][r] = cmul(cpx<T>{c.a[2 * v], c.a[2 * v + 1]}, val);
                        }
                      });
  twolevel_core<T, L1, L2>(x, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2, (const cpx<T>*)a.tw_lo, 0);
  {  // (.) w (already FFT'd and scaled by 1/M on the host), then swap for the inverse transform (bluesteins.rs:236-239)
    int tb = tid;
    FOURIER_LAUNDER(tb);
    const BufRsrc rw = make_rsrc(a.mul);
    const uint32_t woff = (uint32_t)(((tb / CG2) * L1 + (tb % CG2) * VEC) * sizeof(cpx<T>));
    units_batched<T, 8>([&](int r) { return buf_load_unit<T>(rw, woff, (uint32_t)((Q2 * r) * L1 * sizeof(cpx<T>))); },
                        [&](int r, const Unit16<T>& u) {
#pragma unroll
                          for (int v = 0; v < VEC; ++v) {
                            const cpx<T> y = cmul(x[v][r], cpx<T>{u.a[2 * v], u.a[2 * v + 1]});
                            x[v][r] = {y.im, y.re};
                          }
                        });
  }
  __syncthreads();
  {
    int t2 = tid;
    FOURIER_LAUNDER(t2);  // the inverse's lane mappings are derived here, not carried through the forward transform
    twolevel_core<T, L2, L1>(x, t2, smem, (const cpx<T>*)a.tw2, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw_hi, 32);
  }
  // back in the original layout: register r holds index (th + Q1*r)*L2 + cg*VEC + v of the swapped inverse
  const T scale = (T)a.scale;
  {
    int tb = tid;
    FOURIER_LAUNDER(tb);
    const uint32_t soff = (uint32_t)(((tb / CG1) * L2 + (tb % CG1) * VEC) * sizeof(cpx<T>));
    units_batched_rows<T, 8, BLU_ROWS>([&](int r) { return buf_load_unit<T>(rc, soff + (uint32_t)r * ROWB); },
                        [&](int r, const Unit16<T>& c) {
                          Unit16<T> u;
#pragma unroll
                          for (int v = 0; v < VEC; ++v) {
                            cpx<T> y{x[v][r].im, x[v][r].re};
                            y = cmul(y, cpx<T>{c.a[2 * v], c.a[2 * v + 1]});
                            if (a.blu_swap) y = {y.im, y.re};
                            u.a[2 * v] = y.re * scale; u.a[2 * v + 1] = y.im * scale;
                          }
                          buf_store_unit<T>(ro, soff + (uint32_t)r * ROWB, u);
                        });
  }
}

FOURIER_KERNELS_END  // namespace fourier_hip
