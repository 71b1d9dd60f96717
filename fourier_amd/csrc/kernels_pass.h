// kernels_pass.h -- the big-radix Stockham pass over column tiles (FIRST / MID / LAST) or whole rows (ROWS), and the
// Bluestein middle kernel built from the same tile body.  Reference: fourier-algorithms/src/autosort/mod.rs:203-284 (one
// pass), bluesteins.rs:236-239 (the pointwise product between the two inner transforms).  See kernels_common.h.
#pragma once
#include "kernels_common.h"

FOURIER_KERNELS_BEGIN

// ---- tile configuration ----
template <typename T, int L, int CG> struct TileCfg {
  static constexpr int VEC = 16 / (2 * (int)sizeof(T));  // complex numbers per 16-byte unit
  static constexpr int COLS = CG * VEC;                  // columns per tile
  static constexpr int Q = L / 16;                       // threads per column
  static constexpr int NT = Q * CG;                      // threads per workgroup
  static constexpr int R2 = Q >= 16 ? 16 : Q;            // second-stage radix (1 = none)
  static constexpr int R3 = Q / R2;                      // third-stage radix (1 = none)
  // LDS exchange buffer: units indexed [pos][cg] plus a skew so that lanes walking `pos` at fixed
  // cg (the row-contiguous mapping) hit distinct banks.
  static constexpr int PADU = (Q == 1) ? 0 : ((CG >= 32) ? L : (L * CG) / 32);
  static constexpr int UNITS = (Q == 1) ? 0 : L * CG + PADU;
  static constexpr bool SPLIT = (size_t)UNITS * 16 > SPLIT_THRESHOLD;  // exchange re and im planes separately
  static constexpr size_t EXCH_BYTES = SPLIT ? (size_t)UNITS * 8 : (size_t)UNITS * 16;
  static constexpr size_t TABU_OFF = (EXCH_BYTES + 15) & ~(size_t)15;
  static constexpr size_t TABU_BYTES = (size_t)COLS * 16 * sizeof(cpx<T>);
  static constexpr size_t SMEM_FIRST = TABU_OFF + TABU_BYTES;
  static constexpr size_t TABV_BYTES = (size_t)COLS * 8 * sizeof(cpx<T>);  // chirp-in first pass: cross-term table behind tabU
  static constexpr size_t SMEM_MID = TABU_OFF + 16 * sizeof(cpx<T>);
  // MODE_ROWS where the Q lanes of a transform cover no more than 32 bytes of a line per access (f32 L = 64, f64 L = 32)
  // stages its global I/O through LDS, half a tile (COLS / 2 whole transforms) at a time, element (c, p) at
  // c * STAGE_LP + p (pass_tile).  The pad keeps the gather of a half-wave (th + Q*r at fixed r) on distinct banks.
  // Measured (r03_s28_rows_staged_io_ab.jsonl): f32 64 51 -> 62 % of the HBM peak, f64 32 58 -> 64 %; with 64-byte pieces
  // and wider the element form is as good or better (f32 128 64 / 63 %, f64 64 64 / 59 %, f64 128 72 / 59 %).
  static constexpr bool ROWS_STAGED = Q >= 2 && Q * 2 * (int)sizeof(T) <= 32;
  static constexpr int STAGE_LP = L + 4;
  static constexpr size_t STAGE_BYTES = ROWS_STAGED ? (size_t)(COLS / 2) * STAGE_LP * 2 * sizeof(T) : 0;
  static constexpr size_t SMEM_PLAIN = EXCH_BYTES > STAGE_BYTES ? EXCH_BYTES : STAGE_BYTES;
  static __host__ __device__ constexpr size_t smem_bytes(int mode) {
    return mode == MODE_FIRST ? SMEM_FIRST : (mode == MODE_MID ? SMEM_MID : SMEM_PLAIN);
  }
  // LAYOUT 0 ("skew"): conflict-free for lanes walking pos at fixed cg (row-contiguous mapping).
  // LAYOUT 1 ("xor"):  for the stage-1 exchange of the split-plane tiles, where a 16-lane ds_write_b64
  //   group holds 16/CG threads whose positions differ by 16: flip the unit index by the 16-block
  //   parity so those threads land in different bank quarters; reads (cg-fastest) stay contiguous.
  // LAYOUT 2 ("rows"): the whole-transform kernels (MODE_ROWS), whose lanes walk th = tid % Q -- a write instruction touches positions
  //   16*th + r, a read instruction positions th + Q*r, a wave spans 64 / Q column groups.  Under the skew layout that is 2.3x (L = 512, 1024)
  //   to 6.7x (L = 128) the conflict-free LDS cycles: SQ_LDS_BANK_CONFLICT = 57 % of the LDS cycles of the chirp-z kernel of M = 512, and the
  //   emulator's bank model agrees to three digits (round 6, profiles/r06_s3_sq_chirpz.json).  An XOR swizzle per (L, CG), found by
  //   tools/lds_rows_swizzle_search.py by exhaustive search over this family under the bank model of MI355X_MICROARCH.md: every row-mode
  //   exchange conflict-free, no padding units.
  struct RowSwz { int a, ma, b1, s1, b2, s2; };
  static constexpr bool HAS_ROW_SWZ = (L == 32 && CG == 32) || (L == 64 && CG == 16) || (L == 128 && (CG == 8 || CG == 16 || CG == 32)) ||
                                      (L == 256 && (CG == 4 || CG == 8 || CG == 16)) || (L == 512 && (CG == 2 || CG == 4 || CG == 8)) ||
                                      (L == 1024 && (CG == 1 || CG == 4 || CG == 8));
  static constexpr RowSwz row_swz() {
    return L == 32 ? RowSwz{0, 0, 0, 4, 1, 0} : L == 64 ? RowSwz{0, 0, 0, 2, 2, 0}
           : L == 128 ? (CG == 8 ? RowSwz{4, 1, 0, 0, 4, 0} : CG == 16 ? RowSwz{0, 0, 0, 1, 3, 0} : RowSwz{0, 0, 0, 2, 3, 0})
           : L == 256 ? (CG == 4 ? RowSwz{4, 3, 2, 0, 6, 0} : CG == 8 ? RowSwz{4, 1, 1, 0, 5, 0} : RowSwz{0, 0, 0, 0, 4, 0})
           : CG == 1 ? RowSwz{4, 15, 0, 0, 0, 0} : CG == 2 ? RowSwz{5, 7, 3, 0, 4, 0} : CG == 4 ? RowSwz{6, 3, 2, 0, 4, 0} : RowSwz{7, 1, 1, 0, 4, 0};  // 512, 1024
  }
  template <int LAYOUT> static __device__ __forceinline__ int unit_index(int pos, int cg) {
    if constexpr (LAYOUT == 2 && HAS_ROW_SWZ) {
      constexpr RowSwz z = row_swz();
      return (pos ^ ((pos >> z.a) & z.ma)) * CG + (cg ^ (((pos >> z.b1) << z.s1) & (CG - 1)) ^ (((pos >> z.b2) << z.s2) & (CG - 1)));
    } else if constexpr (LAYOUT == 1 && CG <= 8) return (pos * CG + cg) ^ (((pos >> 4) & (16 / CG - 1)) * CG);
    else return pos * CG + cg + ((CG >= 32) ? pos : ((pos * CG) >> 5));
  }
};

// Exchange through LDS: register r of this thread goes to position wpos(r) of column group cg_w;
// afterwards register r holds position th_r + Q*r of column group cg_r.
template <typename T, int L, int CG> using RegTile = cpx<T>[TileCfg<T, L, CG>::VEC][16];

template <typename T, int L, int CG, int LAYOUT, typename WPos>
__device__ __forceinline__ void lds_exchange(RegTile<T, L, CG>& x, unsigned char* smem, int cg_w, WPos wpos, int th_r,
                                             int cg_r, unsigned site) {
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q;
  if constexpr (C::SPLIT) {
    // two rounds of 8-byte units through a buffer of half the tile: the re plane, then the im plane, of the VEC columns of a unit
    Unit8<T>* lds = (Unit8<T>*)smem;
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
      if (plane == 1) __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Unit8<T> u;
#pragma unroll
        for (int v = 0; v < VEC; ++v) u.a[v] = plane ? x[v][r].im : x[v][r].re;
        Unit8<T>* p = lds + C::template unit_index<LAYOUT>(wpos(r), cg_w);
        LDS_NOTE(p, 8, true, site + plane);
        *p = u;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const Unit8<T>* p = lds + C::template unit_index<LAYOUT>(th_r + Q * r, cg_r);
        LDS_NOTE(p, 8, false, site + 2 + plane);
        const Unit8<T> u = *p;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          if (plane) x[v][r].im = u.a[v]; else x[v][r].re = u.a[v];
        }
      }
    }
  } else {
    Unit16<T>* lds = (Unit16<T>*)smem;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Unit16<T> u;
#pragma unroll
      for (int v = 0; v < VEC; ++v) { u.a[2 * v] = x[v][r].re; u.a[2 * v + 1] = x[v][r].im; }
      Unit16<T>* p = lds + C::template unit_index<LAYOUT>(wpos(r), cg_w);
      LDS_NOTE(p, 16, true, site);
      *p = u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const Unit16<T>* p = lds + C::template unit_index<LAYOUT>(th_r + Q * r, cg_r);
      LDS_NOTE(p, 16, false, site + 2);
      const Unit16<T> u = *p;
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
    }
  }
}

// Sixteen table units, one per register row of a tile, applied B at a time: the B loads of a batch are issued back to
// back, then consumed.  Left to itself hipcc (128-VGPR budget, 64 of them the tile) issues ONE load, waits for it,
// multiplies, and only then issues the next -- sixteen exposed L2 / HBM latencies per tile.
// (NR: the register rows that take part -- 16, or 8 where the upper half of a chirp-z work array is padding)
template <typename T, int B, int NR, typename Ld, typename Use>
__device__ __forceinline__ void units_batched_rows(const Ld& ld, const Use& use) {
#pragma unroll
  for (int r0 = 0; r0 < NR; r0 += B) {
    Unit16<T> u[B];
#pragma unroll
    for (int q = 0; q < B; ++q) u[q] = ld(r0 + q);
    FOURIER_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < B; ++q) use(r0 + q, u[q]);
    FOURIER_SCHED_FENCE();
  }
}
template <typename T, int B, typename Ld, typename Use>
__device__ __forceinline__ void units_batched(const Ld& ld, const Use& use) { units_batched_rows<T, B, 16>(ld, use); }

// x[v][k] *= t[k], k = 1..15 (t[0] = 1): the stage twiddles of one thread, sixteen consecutive table entries.  All of
// them (f64: half of them) are loaded in one batch; left alone hipcc picks a grouping that waits per load (r03; 0 and 4 per batch: slower)
template <typename T, int VEC>
__device__ __forceinline__ void stage_twiddle(cpx<T> (&x)[VEC][16], const cpx<T>* t) {
  constexpr int PER = 16 / (int)sizeof(cpx<T>), NU = 16 / PER, B = 8 < NU ? 8 : NU;
#pragma unroll
  for (int u0 = 0; u0 < NU; u0 += B) {
    Unit16<T> u[B];
#pragma unroll
    for (int q = 0; q < B; ++q) {
      if constexpr (FOURIER_ABLATE == 4) { for (int j = 0; j < 2 * PER; ++j) u[q].a[j] = (j & 1) ? (T)0.6 : (T)0.8; (void)t; }
      else u[q] = *(const Unit16<T>*)(t + (u0 + q) * PER);
    }
    FOURIER_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < B; ++q)
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int k = (u0 + q) * PER + j;
        if (k == 0) continue;
        const cpx<T> w{u[q].a[2 * j], u[q].a[2 * j + 1]};
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[v][k] = cmul(x[v][k], w);
      }
    FOURIER_SCHED_FENCE();
  }
}

template <typename T>
__device__ __forceinline__ cpx<T> two_level_twiddle(const PassArgs& a, uint64_t e) {
  const cpx<T>* lo = (const cpx<T>*)a.tw_lo;
  const cpx<T>* hi = (const cpx<T>*)a.tw_hi;
  const uint64_t el = e & ((1ull << a.lo_bits) - 1), eh = e >> a.lo_bits;
  return cmul(lo[el], hi[eh]);
}

// One big-radix Stockham pass over a tile of COLS columns (or COLS whole transforms in ROWS mode).
//   MODE_FIRST: s == 1. column-tile load, transposed (row-contiguous) store, twiddle W_size^{i*k}.
//   MODE_MID  : s >= COLS. column-tile load/store, twiddle W_size^{i*k} with i uniform per tile.
//   MODE_LAST : size == L. column-tile load/store, no twiddle; swap_out / scale on store.
//   MODE_ROWS : whole transforms of length L, contiguous rows; swap_out / scale on store.
// Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md).  Bijective remap of the block index so
// that each XCD's L2/TLB sees a compact working set; affects speed only.
//   mode 0: every XCD owns a contiguous range of tiles (= whole transforms): each 2 MiB page and each DRAM row is
//           touched by one XCD instead of all eight (+11..16% on the strided tile pattern, tools/membench.py --xcd)
//   mode 1: XCD x takes transforms x, x + 8, ... (eight XCDs on eight adjacent transforms; measured slower)
//   mode 2: XCD x owns the x-th eighth of the TILES of every transform: the slice of a per-transform table (Bluestein
//           chirp / transformed chirp, indexed like the data) that an XCD reads stays in its 4 MiB L2
//   mode 3: every XCD owns a contiguous range of whole transforms (as mode 0) but walks it band-major: an eighth of the
//           tile columns for ALL of its transforms, then the next eighth -- a per-transform table's band is re-read from
//           the L2 by transform after transform while no transform or page is shared between XCDs
// 32-bit arithmetic throughout (a grid has fewer than 2^31 blocks): the 64-bit form costs a few hundred scalar
// instructions per tile, which a persistent workgroup pays once per tile.
__device__ __forceinline__ uint32_t xcd_remap(const PassArgs& a, uint64_t blk64, uint64_t nwg64) {
  const uint32_t blk = (uint32_t)blk64, nwg = (uint32_t)nwg64, tiles = (uint32_t)a.tiles;
  if (a.nxcd <= 1) return blk;
  const uint32_t nx = a.nxcd, xcd = blk % nx, slot = blk / nx;
  if (a.xcd_interleave == 1 && tiles > 0 && nwg % (nx * tiles) == 0)
    return ((slot / tiles) * nx + xcd) * tiles + slot % tiles;
  if (a.xcd_interleave == 2 && tiles > 0 && tiles % nx == 0) {
    const uint32_t tpx = tiles / nx;
    return (slot / tpx) * tiles + xcd * tpx + slot % tpx;
  }
  if (a.xcd_interleave == 3 && tiles > 0 && tiles % 8 == 0 && nwg % (nx * tiles) == 0) {
    const uint32_t tpb = tiles / 8, t_per_xcd = nwg / (nx * tiles), per_band = t_per_xcd * tpb;
    const uint32_t band = slot / per_band, rem = slot % per_band;
    return (xcd * t_per_xcd + rem / tpb) * tiles + band * tpb + rem % tpb;
  }
  if (a.xcd_interleave == 4 && tiles > 0 && a.walk_band > 0 && tiles % a.walk_band == 0 && nwg % (nx * tiles) == 0) {
    // the general band walk: the XCD's contiguous range of whole transforms in groups of `walk_group`; inside a group band-major
    // (`walk_band` adjacent tiles of every transform of the group, then the next band).  Workgroups that are resident together
    // then work on walk_band * 128 bytes of every row of (resident tiles / walk_band) transforms instead of on whole rows of one.
    const uint32_t tpx = nwg / (nx * tiles), bw = a.walk_band;
    const uint32_t g = (a.walk_group == 0 || a.walk_group > tpx) ? tpx : a.walk_group;
    if (tpx % g == 0) {
      const uint32_t per_group = g * tiles, grp = slot / per_group, rem = slot % per_group;
      const uint32_t per_band = g * bw, band = rem / per_band, rem2 = rem % per_band;
      const uint32_t tr = (a.walk_tf & 1) ? rem2 % g : rem2 / bw, tl = (a.walk_tf & 1) ? rem2 / g : rem2 % bw;
      // (walk_tf bit 1, A/B: a band holds every (tiles / bw)-th tile instead of bw adjacent ones)
      return (xcd * tpx + grp * g + tr) * tiles + ((a.walk_tf & 2) ? band + (tiles / bw) * tl : band * bw + tl);
    }
  }
  const uint32_t q = nwg / nx, r = nwg % nx;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---- in-tile DFT of length L = 16 x R2 x R3 on a register tile (the body of every pass kernel) ----
// In: thread (th, cg) holds rows th + Q*r of columns cg*VEC + v.  Out: register r holds output index
// k = th + Q*r; for MODE_FIRST the last exchange also switches the thread mapping from cg-fastest ("A") to
// th-fastest ("B", th = tid % Q, cg = tid / Q) for the row-contiguous store.  Uses the exchange buffer at smem.
// The "B" mapping is derived from a laundered copy of tid where it is first needed, so that nothing that depends on
// it (the store addresses of the whole tile) is computed at the top of the kernel and carried through the butterflies.
template <typename T, int L, int CG, int MODE>
__device__ __forceinline__ void tile_core(RegTile<T, L, CG>& x, int& th, int& cg, const int tid,
                                          unsigned char* smem, const cpx<T>* tw1, const cpx<T>* tw2) {
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, R2 = C::R2, R3 = C::R3;
  constexpr bool IN_ROWS = (MODE == MODE_ROWS);
  // ---- stage 1: radix 16 over rows th + Q*k'  ->  positions 16*th + k, twiddle W_L^{th*k}
  constexpr bool DO_MATH = (FOURIER_ABLATE != 1 && FOURIER_ABLATE != 2);
  constexpr bool DO_EXCH = (FOURIER_ABLATE != 2);
  if constexpr (DO_MATH) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) dft16(x[v]);
  }

  if constexpr (Q > 1) {
    if constexpr (DO_MATH) {
      stage_twiddle<T, VEC>(x, tw1 + th * 16);
    }
    {
      constexpr bool remap = (MODE == MODE_FIRST) && (R3 == 1);
      int tb = tid;
      if constexpr (remap) FOURIER_LAUNDER(tb);
      const int th_r = remap ? tb % Q : th, cg_r = remap ? tb / Q : cg;
      const int th_w = th;
      // both sides cg-fastest and split planes -> xor layout; anything row-contiguous -> skew layout
      constexpr int LAY1 = IN_ROWS ? 2 : ((C::SPLIT && !(MODE == MODE_FIRST && R3 == 1)) ? 1 : 0);
      if constexpr (DO_EXCH)
        lds_exchange<T, L, CG, LAY1>(x, smem, cg, [=](int r) { return 16 * th_w + r; }, th_r, cg_r, 0);
      th = th_r; cg = cg_r;
    }

    // ---- stage 2: radix R2 on butterflies q = th + Q*u (register sets {u + NB2*k'})
    constexpr int NB2 = 16 / R2;
    if constexpr (DO_MATH)
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

    if constexpr (R3 > 1) {
      // here R2 == 16, one butterfly per thread: q = th, j = th & 15, i = th >> 4
      if constexpr (DO_MATH) {
        stage_twiddle<T, VEC>(x, tw2 + (th >> 4) * 16);
      }
      {
        constexpr bool remap = (MODE == MODE_FIRST);
        int tb = tid;
        if constexpr (remap) FOURIER_LAUNDER(tb);
        const int th_r = remap ? tb % Q : th, cg_r = remap ? tb / Q : cg;
        const int jw = th & 15, iw = th >> 4;
        __syncthreads();  // all reads of exchange 1 are done before the buffer is rewritten
        if constexpr (DO_EXCH)
        lds_exchange<T, L, CG, IN_ROWS ? 2 : 0>(x, smem, cg, [=](int r) { return jw + 16 * (16 * iw + r); }, th_r, cg_r, 4);
        th = th_r; cg = cg_r;
      }
      // ---- stage 3: radix R3 on register sets {u + NB3*k'}
      constexpr int NB3 = 16 / R3;
      if constexpr (DO_MATH)
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int u = 0; u < NB3; ++u) {
          cpx<T> t[R3];
#pragma unroll
          for (int k = 0; k < R3; ++k) t[k] = x[v][u + NB3 * k];
          dft_r<T, R3>(t);
#pragma unroll
          for (int k = 0; k < R3; ++k) x[v][u + NB3 * k] = t[k];
        }
    }
  }
}

// The body of a pass: one tile (block index `blk0` of `nblk`) of one big-radix Stockham pass.  LDPOL / STPOL = cache
// policy of the data loads / stores (POL_*): the stand-alone pass kernels stream (non-temporal), the XCD-fused kernel
// parks its intermediate in the L2 (plain stores, sc1 loads).
//
// SPLIT = 1 (MODE_LAST only): the pass has length 2L and TWO workgroups share one column tile.  Decimation in frequency:
// X[2k'+p] = DFT_L( (x[n] + (-1)^p x[n+L]) * W_2L^{p*n} )_k', so workgroup p (= block parity) loads all 2L rows, keeps
// the sums (p = 0) or the twiddled differences (p = 1) -- L points per column, the register tile of a length-L pass --
// and produces the even or odd output rows.  A 2048-point pass then runs as two 512-thread workgroups with a 128 KiB
// tile each (two per CU, load and compute phases overlap) instead of one 1024-thread workgroup whose 256 KiB tile
// fills the CU's registers; the tile is read twice, the second time from the XCD's L2 (the two workgroups are adjacent
// blocks of one XCD), written once.
// (x * y) mod n for x, y with x*y < 2^53, exactly: the f64 product and the fused remainder are exact, the quotient estimate
// is off by at most one
__device__ __forceinline__ uint32_t mulmod_n(uint32_t x, uint32_t y, double n, double inv_n) {
  const double prod = (double)x * (double)y;
  const double q = __builtin_floor(prod * inv_n);
  double r = __builtin_fma(-q, n, prod);
  r = r < 0.0 ? r + n : (r >= n ? r - n : r);
  return (uint32_t)r;
}
// W_n^e, e < n, from the two-level table (one complex multiply)
template <typename T> __device__ __forceinline__ cpx<T> root_n(const PassArgs& a, uint32_t e) {
  const cpx<T>* lo = (const cpx<T>*)a.tn_lo;
  const cpx<T>* hi = (const cpx<T>*)a.tn_hi;
  return cmul(lo[e & ((1u << a.tn_bits) - 1u)], hi[e >> a.tn_bits]);
}

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// `before_store` runs (on every thread) after the tile's arithmetic and before its first store: the XCD-fused kernel waits
// there for its window slot, so that a tile's HBM loads and butterflies are not held up by the readers of the slot's
// previous tenant.
template <typename T, int L, int CG, int MODE, int IO, int LDPOL, int STPOL, int SPLIT = 0, typename Hook = NoHook>
__device__ __forceinline__ void pass_tile(const PassArgs& a, uint64_t blk0, uint64_t nblk, unsigned char* smem, const int tid,
                                          const Hook& before_store = Hook()) {
  static_assert(IO == IO_PLAIN || (IO == IO_BLU_IN && MODE == MODE_FIRST) || (IO == IO_BLU_OUT && MODE == MODE_LAST),
                "Bluestein fusion: chirp-in on the first pass, chirp-out on the last pass");
  static_assert(SPLIT == 0 || (MODE == MODE_LAST && LDPOL != POL_SC1), "split tiles: last pass only");
  constexpr int KM = SPLIT ? 2 : 1;  // output rows (and the pass length) are KM times what the register tile holds
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, R2 = C::R2, R3 = C::R3, COLS = C::COLS;
  constexpr bool IN_ROWS = (MODE == MODE_ROWS);
  constexpr bool OUT_ROWS = (MODE == MODE_FIRST || MODE == MODE_ROWS);
  constexpr bool TWIDDLED = (MODE == MODE_FIRST || MODE == MODE_MID);
  constexpr bool FINAL = (MODE == MODE_LAST || MODE == MODE_ROWS);
  (void)R2; (void)R3;

  // cg-fastest mapping ("A") for column-tile I/O, th-fastest ("B") for row-contiguous I/O
  int th = IN_ROWS ? tid % Q : tid / CG;
  int cg = IN_ROWS ? tid / Q : tid % CG;

  // Workgroup b is dispatched to XCD b % 8 (observed, MI355X_MICROARCH.md).  Give every XCD its own
  // contiguous range of tiles (= whole transforms): each 2 MiB page and each DRAM row is then
  // touched by one XCD's L2/TLB instead of all eight (+11..16% on the strided tile pattern, measured
  // with tools/membench.py --xcd).  Bijective for any grid size; affects speed only.
  uint32_t blk = xcd_remap(a, blk0, nblk);
  const int par = SPLIT ? (int)(blk & 1) : 0;
  if constexpr (SPLIT) blk >>= 1;
  const cpx<T>* __restrict__ in = (const cpx<T>*)a.in;
  cpx<T>* __restrict__ out = (cpx<T>*)a.out;
  uint64_t b = 0, c0 = 0, g0 = 0;
  if constexpr (IN_ROWS) {
    g0 = (uint64_t)blk * COLS;
  } else {
    b = blk / (uint32_t)a.tiles;
    c0 = (uint64_t)(blk % (uint32_t)a.tiles) * COLS;
    g0 = b * a.cn + c0;
  }
  // Column-tile accesses (everything but the row-contiguous side of FIRST / ROWS) go through buffer descriptors: the
  // wave-uniform part of an address -- transform, tile, and the row r of the sixteen a thread owns -- sits in the
  // descriptor base (scalar registers, one 64-bit scalar add per row), the per-lane part is ONE 32-bit byte offset
  // for all sixteen rows, instead of sixteen 64-bit pointers in vector registers.  No size limit: a lane offset is below
  // n * sizeof(complex) / 16.
  constexpr int LDAUX = LDPOL == POL_NT ? BUF_NT : (LDPOL == POL_SC1 ? BUF_SC1 : BUF_PLAIN);
  constexpr int STAUX = STPOL == POL_NT ? BUF_NT : BUF_PLAIN;

  // ---- inter-pass twiddle table for this tile: tabU[col][r] = W_size^{i_col * Q * r}.  Filled BEHIND the tile's own loads:
  // its table loads feed LDS writes, and ahead of the tile's loads -- where it used to sit -- their L2
  // latency was served before the first HBM load of the tile was issued, once per tile
  auto fill_tabs = [&]() {
  if constexpr (TWIDDLED) {
    cpx<T>* tabU = (cpx<T>*)(smem + C::TABU_OFF);
    if constexpr (MODE == MODE_FIRST) {
      for (int idx = tid; idx < COLS * 16; idx += C::NT) {
        const uint64_t i = c0 + (uint64_t)(idx >> 4);
        tabU[idx] = two_level_twiddle<T>(a, i * (uint64_t)(Q * (idx & 15)));
      }
      if constexpr (IO == IO_BLU_IN) {
        // computed chirp, cross term of column b and register r: tabV[col][r] = W_n^{(cn*Q*b*r) mod n}, r < 8
        if (a.blu_p) {
          cpx<T>* tabV = tabU + COLS * 16;
          for (int idx = tid; idx < COLS * 8; idx += C::NT) {
            const uint32_t bcol = (uint32_t)c0 + (uint32_t)(idx >> 3);
            const uint32_t e = mulmod_n(mulmod_n(a.blu_cnq_mod, bcol, a.blu_nd, a.blu_inv_nd), (uint32_t)(idx & 7), a.blu_nd, a.blu_inv_nd);
            tabV[idx] = root_n<T>(a, e);
          }
        }
      }
    } else {
      if (tid < 16) tabU[tid] = two_level_twiddle<T>(a, (c0 >> a.s_shift) * (uint64_t)(Q * tid));
    }
  }
  };
  // ---- load: register r <- row th + Q*r
  cpx<T> x[VEC][16];
  constexpr bool STAGED = IN_ROWS && C::ROWS_STAGED;
  // staged rows: thread (th, cg) owns transform v*CG + cg of the tile (not cg*VEC + v), so that each half of the tile --
  // v = 0 / v = 1 in f32, cg below / above CG/2 in f64 -- is one contiguous run of COLS/2 transforms
  constexpr int HALF = COLS / 2, LP = C::STAGE_LP;
  uint32_t stage_valid = 0;  // elements of this tile that exist (the last tile of a batch may be ragged)
  if constexpr (STAGED) {
    const uint64_t left = a.total_cols > g0 ? a.total_cols - g0 : 0;
    stage_valid = (uint32_t)(left < (uint64_t)COLS ? left : (uint64_t)COLS) * (uint32_t)L;
    cpx<T>* stage = (cpx<T>*)smem;
    const cpx<T>* src = in + g0 * L;
    // every unit of BOTH halves is loaded before the first LDS write: one memory latency per tile instead of one per half (the
    // registers are free: x is filled from LDS afterwards)
    constexpr int SITER = (HALF * L / VEC + C::NT - 1) / C::NT;
    Unit16<T> sw[2][SITER];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int it = 0; it < SITER; ++it) {
        const int u = tid + it * C::NT;
        const uint32_t eh = (uint32_t)u * VEC, e = (uint32_t)(h * HALF * L) + eh;
        Unit16<T> w{};
        if (u < HALF * L / VEC && e < stage_valid) w = load_unit_a8<T>(src + e);
        sw[h][it] = w;
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int it = 0; it < SITER; ++it) {
        const int u = tid + it * C::NT;
        const uint32_t eh = (uint32_t)u * VEC;
        if (u < HALF * L / VEC) *(Unit16<T>*)(stage + (eh / L) * LP + (eh % L)) = sw[h][it];
      }
      __syncthreads();
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int col = v * CG + cg;
        if (col / HALF == h) {
          const cpx<T>* p = stage + (col % HALF) * LP + th;
#pragma unroll
          for (int r = 0; r < 16; ++r) x[v][r] = p[Q * r];
        }
      }
      __syncthreads();
    }
  } else if constexpr (IN_ROWS) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint64_t g = g0 + (uint64_t)(cg * VEC + v);
      const bool valid = g < a.total_cols;
      const cpx<T>* p = in + g * L + th;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[v][r] = valid ? p[Q * r] : cpx<T>{0, 0};
    }
  } else if constexpr (IO == IO_BLU_IN) {
    // work = x (.) in, zero padded (bluesteins.rs:229-234).  M >= 2N - 1 and M even give 2N <= M (bluesteins.rs:110; the
    // engine checks it), so rows L/2 .. L-1 of every column hold padding only: registers 8..15 are zero without a
    // load.  The others come through two bounds-checked descriptors (user array, chirp table; the user array is only
    // 8-byte aligned): everything at or beyond blu_n loads as zero, all sixteen loads are in flight together.
    const uint32_t nbytes = (uint32_t)(a.blu_n * sizeof(cpx<T>));
    const BufRsrc rd = make_rsrc(in + b * a.blu_n, nbytes), rc = make_rsrc(a.blu_x, nbytes);
    const uint32_t voff = (uint32_t)(((uint64_t)th * a.cn + c0 + (uint64_t)(cg * VEC)) * sizeof(cpx<T>));
    const uint32_t rowb = (uint32_t)((uint64_t)Q * a.cn * sizeof(cpx<T>));
    Unit16<T> d[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) d[r] = buf_load_unit<T, LDAUX>(rd, voff + (uint32_t)r * rowb);
    fill_tabs();
    if (a.blu_p) {
      // chirp computed, not read: x[k] = P[row] * (U[b] * W_n^{cn*b*th}) * tabV[col][r]   (see PassArgs)
      const cpx<T>* pt = (const cpx<T>*)a.blu_p + th;
      cpx<T> pr[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) pr[r] = pt[Q * r];
      const Unit16<T> uu = *(const Unit16<T>*)((const cpx<T>*)a.blu_u + c0 + (uint64_t)(cg * VEC));
      cpx<T> ub[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const uint32_t bcol = (uint32_t)c0 + (uint32_t)(cg * VEC + v);
        const uint32_t e = mulmod_n(mulmod_n(a.blu_cn_mod, bcol, a.blu_nd, a.blu_inv_nd), (uint32_t)th, a.blu_nd, a.blu_inv_nd);
        ub[v] = cmul(cpx<T>{uu.a[2 * v], uu.a[2 * v + 1]}, root_n<T>(a, e));
      }
      __syncthreads();  // tabV
      const cpx<T>* tabV = (const cpx<T>*)(smem + C::TABU_OFF) + COLS * 16;
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cpx<T> val{d[r].a[2 * v], d[r].a[2 * v + 1]};
          if (a.blu_swap) val = {val.im, val.re};
          const cpx<T> c = cmul(cmul(pr[r], tabV[(cg * VEC + v) * 8 + r]), ub[v]);
          x[v][r] = cmul(c, val);
          x[v][r + 8] = cpx<T>{0, 0};
        }
    } else {
      Unit16<T> c[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) c[r] = buf_load_unit<T>(rc, voff + (uint32_t)r * rowb);
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cpx<T> val{d[r].a[2 * v], d[r].a[2 * v + 1]};
          if (a.blu_swap) val = {val.im, val.re};
          x[v][r] = cmul(cpx<T>{c[r].a[2 * v], c[r].a[2 * v + 1]}, val);
          x[v][r + 8] = cpx<T>{0, 0};
        }
    }
  } else if constexpr (SPLIT) {
    // rows n and n + L of the 2L-row tile; plain loads: the sibling workgroup's copy of each line comes from the L2
    const cpx<T>* p = in + b * a.n + (uint64_t)th * a.cn + c0 + (uint64_t)(cg * VEC);
    const cpx<T>* wh = (const cpx<T>*)a.tw_half + th;  // W_2L^{th + Q*r} at [Q*r + th]
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {
      Unit16<T> u0[4], u1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u0[q] = load_unit<T, LDPOL == POL_NT>(p + (uint64_t)(Q * (r0 + q)) * a.cn);
        u1[q] = load_unit<T, LDPOL == POL_NT>(p + (uint64_t)(Q * (r0 + q) + L) * a.cn);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const cpx<T> w = wh[Q * (r0 + q)];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const cpx<T> lo{u0[q].a[2 * v], u0[q].a[2 * v + 1]}, hi{u1[q].a[2 * v], u1[q].a[2 * v + 1]};
          x[v][r0 + q] = par ? cmul(cpx<T>{lo.re - hi.re, lo.im - hi.im}, w) : cpx<T>{lo.re + hi.re, lo.im + hi.im};
        }
      }
      FOURIER_SCHED_FENCE();
    }
  } else {
    // (POL_SC1: L2-served loads of an intermediate another workgroup of this XCD has just written)
    const cpx<T>* p = in + b * a.n + c0;
    const uint32_t voff = (uint32_t)(((uint64_t)th * a.cn + (uint64_t)(cg * VEC)) * sizeof(cpx<T>));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const Unit16<T> u = buf_load_unit<T, LDAUX>(make_rsrc(p + (uint64_t)(Q * r) * a.cn), voff);
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
    }
    fill_tabs();
  }
  if (a.swap_in) {
#pragma unroll
    for (int v = 0; v < VEC; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) x[v][r] = {x[v][r].im, x[v][r].re};
  }

  // ---- in-tile DFT_L: register r <- row th + Q*r  ==>  register r holds output index k = th + Q*r
  tile_core<T, L, CG, MODE>(x, th, cg, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  // now register r holds output index k = th + Q*r of columns (cg*VEC + v)
  // ---- inter-pass twiddle W_size^{i*k} = W^{i*th} * tabU[col][r]
  if constexpr (TWIDDLED && (FOURIER_ABLATE == 0 || FOURIER_ABLATE >= 4)) {
    if constexpr (Q == 1) __syncthreads();  // tabU visibility when there was no exchange barrier
    const cpx<T>* tabU = (const cpx<T>*)(smem + C::TABU_OFF);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint64_t i = (MODE == MODE_FIRST) ? c0 + (uint64_t)(cg * VEC + v) : c0 >> a.s_shift;
      const cpx<T> base = FOURIER_ABLATE == 5 ? cpx<T>{(T)0.8, (T)(0.6 + 1e-9 * (double)i)} : two_level_twiddle<T>(a, i * (uint64_t)th);
      const cpx<T>* tu = (MODE == MODE_FIRST) ? tabU + (cg * VEC + v) * 16 : tabU;
#pragma unroll
      for (int r = 0; r < 16; ++r) x[v][r] = cmul(x[v][r], cmul(base, tu[r]));
    }
  }

  // ---- store
  before_store();
  const T scale = (T)a.scale;
  if constexpr (STAGED) {
    cpx<T>* stage = (cpx<T>*)smem;
    cpx<T>* dst = out + g0 * L;
    __syncthreads();  // the last exchange's readers are done with the buffer
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int col = v * CG + cg;
        if (col / HALF == h) {
          cpx<T>* p = stage + (col % HALF) * LP + th;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            cpx<T> y = x[v][r];
            if (a.swap_out) y = {y.im, y.re};
            p[Q * r] = {y.re * scale, y.im * scale};
          }
        }
      }
      __syncthreads();
      {  // (the LDS reads of a half ahead of its first store, as on the way in)
        constexpr int SITER = (HALF * L / VEC + C::NT - 1) / C::NT;
        Unit16<T> sw[SITER];
#pragma unroll
        for (int it = 0; it < SITER; ++it) {
          const int u = tid + it * C::NT;
          const uint32_t eh = (uint32_t)u * VEC;
          if (u < HALF * L / VEC) sw[it] = *(const Unit16<T>*)(stage + (eh / L) * LP + (eh % L));
        }
#pragma unroll
        for (int it = 0; it < SITER; ++it) {
          const int u = tid + it * C::NT;
          const uint32_t eh = (uint32_t)u * VEC, e = (uint32_t)(h * HALF * L) + eh;
          if (u < HALF * L / VEC && e < stage_valid) store_unit_a8<T>(dst + e, sw[it]);
        }
      }
      __syncthreads();
    }
  } else if constexpr (OUT_ROWS) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint64_t g = g0 + (uint64_t)(cg * VEC + v);
      if (MODE == MODE_ROWS && g >= a.total_cols) continue;
      cpx<T>* p = out + g * L + th;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cpx<T> y = x[v][r];
        if constexpr (FINAL) {
          if (a.swap_out) y = {y.im, y.re};
          y = {y.re * scale, y.im * scale};
        }
        store_elem<T, STPOL == POL_NT>(p + Q * r, y);
      }
    }
  } else if constexpr (IO == IO_BLU_OUT) {
    // out = work (.) x (.) scale, first blu_n points only (bluesteins.rs:240-258).  This is the last pass (c0 < s), and
    // 2N <= M puts the output rows of registers 8..15 (index >= M/2) beyond the user array: they are never stored.
    // Chirp loads and stores go through bounds-checked descriptors, so the ragged end needs no branch.
    const uint32_t nbytes = (uint32_t)(a.blu_n * sizeof(cpx<T>));
    const BufRsrc ro = make_rsrc(out + b * a.blu_n, nbytes), rc = make_rsrc(a.blu_x, nbytes);
    const uint32_t voff = (uint32_t)((c0 + (uint64_t)(cg * VEC) + a.s * (uint64_t)(KM * th + par)) * sizeof(cpx<T>));
    const uint32_t rowb = (uint32_t)(a.s * (uint64_t)(KM * Q) * sizeof(cpx<T>));
    Unit16<T> c[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) c[r] = buf_load_unit<T>(rc, voff + (uint32_t)r * rowb);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      Unit16<T> u;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        cpx<T> y = x[v][r];
        if (a.swap_out) y = {y.im, y.re};
        y = cmul(y, cpx<T>{c[r].a[2 * v], c[r].a[2 * v + 1]});
        if (a.blu_swap) y = {y.im, y.re};
        u.a[2 * v] = y.re * scale; u.a[2 * v + 1] = y.im * scale;
      }
      // no streaming hint: the user rows of an odd-length f32 batch are only 8-byte aligned, a wave's 128-byte row segment
      // then straddles two lines, and the L2 must be allowed to merge the halves (NT: +29 % bytes written, PMC, round 3)
      buf_store_unit<T, BUF_PLAIN>(ro, voff + (uint32_t)r * rowb, u);
    }
  } else {
    // output row of register r: j0 + s * (KM*L*i + KM*(th + Q*r) + par); uniform part in the descriptor base
    const uint64_t i = c0 >> a.s_shift, j0 = c0 & (a.s - 1);  // s is a power of two for every tile pass
    const uint64_t base = b * a.n + j0 + a.s * ((uint64_t)(KM * L) * i + (uint64_t)par);
    const uint32_t voff = (uint32_t)(((uint64_t)(cg * VEC) + a.s * (uint64_t)(KM * th)) * sizeof(cpx<T>));
    const uint64_t rows = a.s * (uint64_t)(KM * Q);  // elements between a thread's consecutive output rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      Unit16<T> u;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        cpx<T> y = x[v][r];
        if constexpr (FINAL) {
          if (a.swap_out) y = {y.im, y.re};
          y = {y.re * scale, y.im * scale};
        }
        u.a[2 * v] = y.re; u.a[2 * v + 1] = y.im;
      }
      buf_store_unit<T, STAUX>(make_rsrc(out + base + rows * (uint64_t)r), voff, u);
    }
  }
}

// cache policy of the stand-alone pass kernels (kernels_common.h has the measurements)
template <int L, int MODE, int CG = 8> struct PassPolicy {
  static constexpr bool FINAL = (MODE == MODE_LAST || MODE == MODE_ROWS);
  // the 16-column last passes of length 2048 (one 1024-thread workgroup per CU): NO streaming hint on either side.  Round 6, 18 fresh
  // allocations of the C5 chunk: 13.5 -> 12.1 - 12.25 ms on eleven of them, level on the other seven, never slower; 2^22 on shared buffers
  // 25.70 -> 24.51 ms (f64 25.71 -> 25.17), C4's chirp-out pass 3.00 -> 2.94, bit-identical (profiles/r06_s12_*).  The two-workgroups-per-CU
  // passes of length <= 1024 lose 0.2 - 1 ms per 4096 transforms without the hints (r06_s11_last_pass_store_policy.jsonl).
  static constexpr bool WIDE_LAST = (MODE == MODE_LAST && L >= 2048);
  // 64-byte-wide tiles (CG = 4): two workgroups share every 128-byte line, the second one must find it in the L2, so
  // no streaming hint (L = 2048 first pass: 6.5 vs 7.6 ms per 1024 transforms of 2^21, r01 session 11)
  static constexpr int LD = ((MODE != MODE_ROWS && CG < 8) || WIDE_LAST) ? POL_PLAIN : POL_NT;
  // MODE_ROWS: a wave's element stores only form whole lines for L >= 256; below that they rely on L2
  // write-combining and a non-temporal hint is a 2-6x loss (N = 16..64, r01 session 9)
  // (the 8-column first pass of length 2048 keeps its streaming stores: plain 2 - 4 % slower; its 16-column form on one workgroup per CU stays 8 - 25 %
  // slower with or without the hint on its loads; profiles/r06_s17_first_pass_2048_ab.jsonl)
  static constexpr int ST = WIDE_LAST ? POL_PLAIN : ((MODE == MODE_ROWS ? L >= 256 : (FINAL || L <= 1024 || CG < 8)) ? POL_NT : POL_PLAIN);
};

template <typename T, int L, int CG, int MODE, int IO = IO_PLAIN>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_pass_kernel(PassArgs a) {
  FOURIER_DYN_SMEM(smem);
  pass_tile<T, L, CG, MODE, IO, PassPolicy<L, MODE, CG>::LD, PassPolicy<L, MODE, CG>::ST>(a, blockIdx.x, gridDim.x, smem, (int)threadIdx.x);
}

// ---- Bluestein middle (bluesteins.rs:236-239): LAST pass of the forward inner FFT, (.) w, and FIRST pass of
// the inverse inner FFT in ONE launch.  The last forward pass (R = L, s = M/L) leaves X[j + (M/L)*k] of its
// column tile in registers; an inverse FFT whose first pass has the same length (R = L, s = 1, m = M/L) reads
// exactly those elements as its columns i = j, so the M-point spectrum never goes back to HBM: one read and
// one write of the work array instead of two of each.  The inverse is swap . DFT . swap (mod.rs:366-387):
// the leading swap happens here, the trailing one in the inverse plan's last pass.
template <typename T, int L, int CG>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_conv_kernel(PassArgs a) {
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, COLS = C::COLS;
  static_assert(Q > 1, "conv kernel: L >= 32");
  FOURIER_DYN_SMEM(smem);
  const int tid = (int)threadIdx.x;
  const uint32_t blk = xcd_remap(a, blockIdx.x, gridDim.x);
  const uint64_t b = blk / (uint32_t)a.tiles, c0 = (uint64_t)(blk % (uint32_t)a.tiles) * COLS;
  const cpx<T>* __restrict__ in = (const cpx<T>*)a.in + b * a.n;
  cpx<T>* __restrict__ out = (cpx<T>*)a.out + b * a.n;
  // Everything a phase derives from the thread index is derived from a laundered copy taken AT that phase: hipcc
  // otherwise computes the addresses of all phases at the top of the kernel and carries (or spills) them across the
  // two in-tile FFTs.
  const uint32_t rowb = (uint32_t)((uint64_t)Q * a.cn * sizeof(cpx<T>));  // byte distance of a thread's consecutive rows

  // inter-pass twiddle table of the inverse FFT's first pass: tabU[col][r] = W_M^{i_col * Q * r} -- filled behind the tile's loads
  // (see pass_tile)
  cpx<T>* tabU = (cpx<T>*)(smem + C::TABU_OFF);
  auto fill_tabs = [&]() {
    for (int idx = tid; idx < COLS * 16; idx += C::NT) {
      const uint64_t i = c0 + (uint64_t)(idx >> 4);
      tabU[idx] = two_level_twiddle<T>(a, i * (uint64_t)(Q * (idx & 15)));
    }
  };
  // forward LAST pass: rows th + Q*r (stride cn) of columns c0 + cg*VEC + v
  cpx<T> x[VEC][16];
  int th = tid / CG, cg = tid % CG;
  {
    const BufRsrc rs = make_rsrc(in);
    const uint32_t voff = (uint32_t)(((uint64_t)th * a.cn + c0 + (uint64_t)(cg * VEC)) * sizeof(cpx<T>));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // tiles narrower than a 128-byte line share every line with a sibling workgroup: no streaming hint then
      // (round 6: without the hint on these loads the kernel loses 3 - 7 %, without it on its stores 4 - 10 %; profiles/r06_s16_conv_policy_ab.jsonl)
      const Unit16<T> u = buf_load_unit<T, CG >= 8 ? BUF_NT : BUF_PLAIN>(rs, voff, (uint32_t)r * rowb);
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
    }
  }
  fill_tabs();
  tile_core<T, L, CG, MODE_LAST>(x, th, cg, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  // register r holds X[c + cn*(th + Q*r)]: (.) w (FFT'd chirp, 1/M folded in), then the inverse's leading swap
  {
    int t = tid;
    FOURIER_LAUNDER(t);
    const BufRsrc rw = make_rsrc(a.mul);
    const uint32_t voff = (uint32_t)(((uint64_t)(t / CG) * a.cn + c0 + (uint64_t)((t % CG) * VEC)) * sizeof(cpx<T>));
    // (eight loads of the w table in flight per thread: 4 and 16 measured slower, as did a streaming hint on them -- the XCD's slice of
    // the table is meant to stay in its L2)
    units_batched<T, 8>([&](int r) { return buf_load_unit<T, BUF_PLAIN>(rw, voff, (uint32_t)r * rowb); },
        [&](int r, const Unit16<T>& u) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const cpx<T> y = cmul(x[v][r], cpx<T>{u.a[2 * v], u.a[2 * v + 1]});
            x[v][r] = {y.im, y.re};
          }
        });
  }
  __syncthreads();  // every read of the last exchange is done before the buffer is rewritten
  // inverse FIRST pass on the same tile (columns i = c0 + ..., s = 1), thread mapping switches to th-fastest
  {
    int t = tid;
    FOURIER_LAUNDER(t);
    th = t / CG; cg = t % CG;
    tile_core<T, L, CG, MODE_FIRST>(x, th, cg, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const uint64_t i = c0 + (uint64_t)(cg * VEC + v);
    const cpx<T> base = two_level_twiddle<T>(a, i * (uint64_t)th);
    const cpx<T>* tu = tabU + (cg * VEC + v) * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) x[v][r] = cmul(x[v][r], cmul(base, tu[r]));
  }
  // transposed store: column i's L outputs are contiguous; streaming (with the XCD-sliced tile order of launch_conv: 4.5 vs 4.75 ms at
  // C4, 7.8 vs 8.3 ms at N = 65537, r02 session 3)
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    cpx<T>* p = out + (c0 + (uint64_t)(cg * VEC + v)) * L + th;
#pragma unroll
    for (int r = 0; r < 16; ++r) store_elem<T, true>(p + Q * r, x[v][r]);
  }
}

FOURIER_KERNELS_END  // namespace fourier_hip
