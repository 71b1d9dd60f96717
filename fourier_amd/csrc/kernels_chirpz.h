// kernels_chirpz.h -- the whole Bluestein chirp-z (bluesteins.rs:215-259) of a SHORT transform in one launch on a SMOOTH M = R1 x R2 with
// both M-point transforms in REGISTERS (round 6, sessions 43 - 45).
//
// bluesteins.rs:110 asks for M >= 2N - 1 only; the reference -- and kernels_onelaunch.h -- take the next power of two, up to 4N (N = 191:
// 512 where 400 = 20 x 20 does; 439: 1024 / 900; the reference's own prime and composite bench sets, fft_bench.rs:157-158).  Here one thread
// group of R1 lanes owns one transform (f32: TWO, see below), 64 / R1 groups share a wave, a workgroup is ONE wave (every barrier stays inside it):
//   stage A   lane q < R2 loads the user points q + R2*j1 (j1 < R1; those from M/2 + 1 on are padding: compile-time zeros), times the chirp,
//             DFT_R1 in registers (dft_any, kernels_regtile.h), writes the exchange buffer;
//   stage B   lane q = k1 < R1 reads its R2 values, twiddle W_M^{j2*k1}, DFT_R2: Y[k1 + R1*k2]; (.) w; swap; and stage A' of the
//             inverse transform on the same registers (DFT_R2 over k2, twiddle W_M^{k1*k1''}) -- the split the other way round, exactly
//             tiled_reg_conv_kernel's; second exchange;
//   stage B'  lane q = k1'' < R2: DFT_R1 over k1: output k1'' + R2*k2'', the positions the lane loaded: swap, times the chirp (the values of
//             stage A where they are cheap to keep), times the scale, the first N stored (bluesteins.rs:240-258).
// One LDS round trip per transform (two in all) where the power-of-two kernels take two or three per transform; HBM sees the user
// array once in and once out, every table (chirp: N, w: M, twiddle: M entries) is L1 / L2 resident.  Tolerance-only route like every
// Bluestein plan here (include/fourier.h).
//
// f32 runs TWO transforms per lane on packed arithmetic (Pk2: re and im of a point are register PAIRS over the two transforms, every butterfly
// a v_pk_add / v_pk_mul / v_pk_fma_f32): the scalar f32 kernels spent 78 - 86 % of their SIMD cycles issuing VALU work at 0.29 - 0.34 of the HBM
// peak (profiles/r06_s44_sq_chirpz_reg.json); the two transforms share every table load, every address and every 16-byte LDS access.
#pragma once
#include "kernels_regtile.h"

namespace fourier_hip {

// two f32 values, one per transform of the lane: arithmetic on both at once
typedef float v2f_t __attribute__((vector_size(8)));
struct Pk2 {
  v2f_t v;
  Pk2() = default;
  __device__ __forceinline__ explicit Pk2(double s) : v{(float)s, (float)s} {}
  __device__ __forceinline__ Pk2(v2f_t w) : v(w) {}
};
__device__ __forceinline__ Pk2 operator+(Pk2 a, Pk2 b) { return Pk2(a.v + b.v); }
__device__ __forceinline__ Pk2 operator-(Pk2 a, Pk2 b) { return Pk2(a.v - b.v); }
__device__ __forceinline__ Pk2 operator*(Pk2 a, Pk2 b) { return Pk2(a.v * b.v); }
__device__ __forceinline__ Pk2 operator-(Pk2 a) { return Pk2(-a.v); }
__device__ __forceinline__ Pk2 operator*(Pk2 a, float s) { return Pk2(a.v * v2f_t{s, s}); }
// lane value type P over memory type T: T itself (one transform per lane) or Pk2 over float (two)
template <typename P, typename T> struct LaneVal {
  static constexpr uint32_t NV = 1;
  static __device__ __forceinline__ P make(const T* s) { return s[0]; }
  static __device__ __forceinline__ T get(P p, uint32_t) { return p; }
};
template <> struct LaneVal<Pk2, float> {
  static constexpr uint32_t NV = 2;
  static __device__ __forceinline__ Pk2 make(const float* s) { return Pk2(v2f_t{s[0], s[1]}); }
  static __device__ __forceinline__ float get(Pk2 p, uint32_t v) { return p.v[v]; }
};
// a lane value times a table entry (one per lane, shared by the lane's transforms)
template <typename P, typename T> __device__ __forceinline__ cpx<P> cmul_tab(cpx<P> a, cpx<T> w) {
  return {a.re * w.re - a.im * w.im, a.re * w.im + a.im * w.re};
}
template <bool B, typename X, typename Y> struct ChirpzSelect { typedef X type; };
template <typename X, typename Y> struct ChirpzSelect<false, X, Y> { typedef Y type; };

template <typename T, uint32_t R1, uint32_t R2> struct ChirpzRegCfg {
  static_assert(R1 >= R2 && R1 <= 64, "chirpz_reg_kernel: split");
  // f32: two transforms per lane on packed arithmetic -- 1.2 ... 1.45 x one transform per lane on scalar arithmetic at every M (profiles/r06_s45_chirpz_reg_ab.jsonl)
  static constexpr bool VEC2 = sizeof(T) == 4;
  using P = typename ChirpzSelect<VEC2, Pk2, T>::type;
  static constexpr uint32_t NV = VEC2 ? 2u : 1u;
  static constexpr uint32_t M = R1 * R2, GPW = 64 / R1, TPW = GPW * NV;  // lane groups and transforms per wave
  // exchange planes: [k1 < R1][j2 < R2] then [k1'' < R2][k1 < R1], the inner pitch odd (a reader's lanes walk the planes)
  static constexpr uint32_t P1 = R2 | 1u, P2 = R1 | 1u;
  static constexpr uint32_t XE = R1 * P1 > R2 * P2 ? R1 * P1 : R2 * P2;  // elements per lane group
  static constexpr size_t SMEM = (size_t)GPW * XE * sizeof(cpx<P>);
  static constexpr uint32_t HALF = M / 2 + 1;  // 2N <= M + 1: user positions stop below this
  static constexpr uint32_t NROW = (HALF + R2 - 1) / R2;  // rows j1 (positions q + R2*j1) that can hold user data
  // waves per SIMD the register allocation aims at: what the LDS lets a compute unit hold (160 KiB, four SIMDs), at most CAP
  static constexpr uint32_t LDS_WAVES = (uint32_t)((160u * 1024u) / SMEM) / 4u;
  static constexpr uint32_t CAP = R1 <= 16 ? 4u : 2u;  // (no bound, or four everywhere: within 2 %, r06_s45)
  static constexpr uint32_t MINW = LDS_WAVES < 1u ? 1u : (LDS_WAVES < CAP ? LDS_WAVES : CAP);
};

template <typename T, uint32_t R1, uint32_t R2>
__global__ void __launch_bounds__(64, (ChirpzRegCfg<T, R1, R2>::MINW)) chirpz_reg_kernel(ChirpzArgs a) {
  using C = ChirpzRegCfg<T, R1, R2>;
  using P = typename C::P;
  using LV = LaneVal<P, T>;
  constexpr uint32_t GPW = C::GPW, TPW = C::TPW, NV = C::NV, P1 = C::P1, P2 = C::P2, XE = C::XE, NROW = C::NROW;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>), XB = (uint32_t)sizeof(cpx<P>), OOB = 0xfffffff0u;
  // the chirp values of stage A stay in registers for the output: f32 (two registers each; loaded again: -2 ... 8 %) and the short f64 stages
  // (+3 ... 8 % up to 14 x 12; 14 x 14 and longer lose up to 20 % to the 4 registers each) -- r06_s45
  constexpr bool KEEP = sizeof(T) == 4 || R1 * R2 <= 168;
  // table loads of a lane in flight together: 8 (4: -1 ... 6 %); 16 gains 3 - 6 % for the f32 stages of 20 ... 30 points and loses elsewhere
  // (f64 up to 30 %: 4 registers per entry, two batches live) -- r06_s45
  constexpr uint32_t TB = (sizeof(T) == 4 && R1 >= 20 && R1 * R2 <= 840) ? 16u : 8u;
  FOURIER_DYN_SMEM(smem);
  const uint32_t lane = threadIdx.x, c = lane / R1, q = lane - c * R1;
  const bool active = c < GPW;
  const uint64_t b0 = (uint64_t)blockIdx.x * TPW;
  const uint32_t n = (uint32_t)a.n;
  const uint32_t nb = a.batch - b0 < TPW ? (uint32_t)(a.batch - b0) : TPW;
  // one bounds-checked descriptor over the wave's transforms: a transform beyond the batch loads zeros and stores nothing; a position
  // beyond n would land in the next transform and is pushed out of range instead
  const BufRsrc rin = make_rsrc((const cpx<T>*)a.in + b0 * n, nb * n * EB), rout = make_rsrc((cpx<T>*)a.out + b0 * n, nb * n * EB);
  const BufRsrc rc = make_rsrc(a.chirp, n * EB);
  cpx<P>* xb = (cpx<P>*)smem + (active ? c : 0u) * XE;
  const cpx<T>* twt = (const cpx<T>*)a.tw + q;  // [j2 < R2][k1 < R1]: W_M^{j2 * k1}, the lane's column
  const cpx<T>* wt = (const cpx<T>*)a.w + q;    // FFT_M(conj chirp, mirrored) / M: entries q + R1 * k2
  const uint32_t tbase = c * NV * n;            // first element of the lane's first transform inside the descriptor

  // ---- stage A
  cpx<T> ch[NROW];
  if (active && q < R2) {
    cpx<P> x[R1];
    cpx<T> d[NV][NROW];
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) {
      const uint32_t pos = q + R2 * j1;
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) d[v][j1] = buf_load_elem<T>(rin, pos < n ? (tbase + v * n + pos) * EB : OOB);
    }
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) ch[j1] = buf_load_elem<T>(rc, (q + R2 * j1) * EB);  // zero beyond n
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) {
      if (j1 < NROW) {
        T re[NV], im[NV];
#pragma unroll
        for (uint32_t v = 0; v < NV; ++v) { re[v] = d[v][j1].re; im[v] = d[v][j1].im; }
        cpx<P> val{LV::make(re), LV::make(im)};
        if (a.swap) val = {val.im, val.re};
        x[j1] = cmul_tab(val, ch[j1]);  // work = x (.) in, zero padded (bluesteins.rs:229-234)
      } else {
        x[j1] = cpx<P>{(P)0, (P)0};
      }
    }
    dft_any<P, (int)R1>(x);
#pragma unroll
    for (uint32_t k1 = 0; k1 < R1; ++k1) {
      cpx<P>* p = xb + k1 * P1 + q;
      LDS_NOTE(p, XB, true, 320);
      *p = x[k1];
    }
  }
  // ---- stage B, (.) w, swap, stage A' of the inverse transform.  Three table products (the lane's R2 twiddles, its R2 entries of w, the
  // twiddles again), their loads in batches of TB, each batch issued one batch ahead of its use -- the first one before the barrier, the
  // first of a product before the transform in front of it: left to itself hipcc hoists every table load of the stage to its top (3 x R2
  // complex values live beside y: spills under any register bound that leaves room for more than two waves per SIMD); loaded batch by
  // batch behind scheduling fences every batch exposes its L2 latency (f64 20 x 20: 15 round trips per wave, profiles/r06_s44_*)
  constexpr uint32_t NBATCH = (R2 + TB - 1) / TB;
  cpx<T> tab[2][TB];
  auto load_batch = [&](uint32_t g) {  // batch g of the 3 * NBATCH: product g / NBATCH, rows (g % NBATCH) * TB ...
    const cpx<T>* t = (g / NBATCH) == 1u ? wt : twt;
    const uint32_t r0 = (g % NBATCH) * TB;
#pragma unroll
    for (uint32_t i = 0; i < TB; ++i)
      if (r0 + i < R2) tab[g & 1u][i] = t[(r0 + i) * R1];
  };
  if (active) load_batch(0);
  __syncthreads();
  cpx<P> y[R2];
  if (active) {
#pragma unroll
    for (uint32_t j2 = 0; j2 < R2; ++j2) {
      const cpx<P>* p = xb + q * P1 + j2;
      LDS_NOTE(p, XB, false, 321);
      y[j2] = *p;
    }
#pragma unroll
    for (uint32_t g = 0; g < 3 * NBATCH; ++g) {
      if (g + 1 < 3 * NBATCH) load_batch(g + 1);
      FOURIER_SCHED_FENCE();
      const uint32_t prod = g / NBATCH, r0 = (g % NBATCH) * TB;
#pragma unroll
      for (uint32_t i = 0; i < TB; ++i)
        if (r0 + i < R2) {
          const cpx<P> z = cmul_tab(y[r0 + i], tab[g & 1u][i]);  // (row 0 of the twiddle table is 1)
          y[r0 + i] = prod == 1u ? cpx<P>{z.im, z.re} : z;      // (.) w, swap for the inverse transform (bluesteins.rs:236-239)
        }
      FOURIER_SCHED_FENCE();
      if (g % NBATCH == NBATCH - 1 && prod < 2u) {
        dft_any<P, (int)R2>(y);  // first: y[k2] = Y[q + R1 * k2]; second, over k2: y[k1''], k1'' < R2
        FOURIER_SCHED_FENCE();
      }
    }
  }
  __syncthreads();  // every lane has read its stage-B inputs
  if (active) {
#pragma unroll
    for (uint32_t k = 0; k < R2; ++k) {
      cpx<P>* p = xb + k * P2 + q;
      LDS_NOTE(p, XB, true, 322);
      *p = y[k];
    }
  }
  __syncthreads();

  // ---- stage B': DFT_R1 over k1; output position q + R2 * k2''
  if (active && q < R2) {
    cpx<P> z[R1];
    if constexpr (!KEEP) {
#pragma unroll
      for (uint32_t j1 = 0; j1 < NROW; ++j1) ch[j1] = buf_load_elem<T>(rc, (q + R2 * j1) * EB);
    }
#pragma unroll
    for (uint32_t k1 = 0; k1 < R1; ++k1) {
      const cpx<P>* p = xb + q * P2 + k1;
      LDS_NOTE(p, XB, false, 323);
      z[k1] = *p;
    }
    dft_any<P, (int)R1>(z);
    const T scale = (T)a.scale;
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) {  // (positions from M/2 + 1 on lie beyond the user array)
      const uint32_t pos = q + R2 * j1;
      cpx<P> o{z[j1].im, z[j1].re};  // the inverse inner transform's trailing swap (1/M is folded into w)
      o = cmul_tab(o, ch[j1]);       // bluesteins.rs:240-258
      if (a.swap) o = {o.im, o.re};
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v)
        buf_store_elem<T>(rout, pos < n ? (tbase + v * n + pos) * EB : OOB, cpx<T>{LV::get(o.re, v) * scale, LV::get(o.im, v) * scale});
    }
  }
}


// ---- M = R1 x R2 x R3 (1152 ... 9261 points): one WORKGROUP per transform (f32: per pair of transforms), three register stages each way ----
// Forward: n = m + R2R3*j1 (m = j3 + R3*j2): DFT_R1 over j1 -> k1; W_M^{m*k1}; DFT_R2 over j2 -> k2; W_{R2R3}^{j3*k2}; DFT_R3 over j3 -> k3:
// Y[k1 + R1*k2 + R1R2*k3].  The inverse takes the split the other way round, as above: a = k1 + R1*k2, DFT_R3 over k3 -> c3 on the same lanes;
// W_M^{a*c3}; DFT_R2 over k2 -> c2; W_{R1R2}^{k1*c2}; DFT_R1 over k1 -> c1: output c3 + R3*c2 + R2R3*c1 -- the positions stage A loaded.
// Lanes: A / A' m (< R2R3), B k1*R3 + j3, C / C' a = k1 + R1*k2, B' c3*R1 + k1.  Four LDS exchanges [row][lane of the reader]: a reader's lanes are
// contiguous, a writer's lanes either run contiguously with the rows a multiple of 16 units + the run length apart, or walk an odd row pitch.
constexpr uint32_t chirpz3_pitch_runs(uint32_t lanes, uint32_t run) {  // >= lanes, = run (mod 16)
  uint32_t p = lanes;
  while (p % 16u != run % 16u) ++p;
  return p;
}
template <typename T, uint32_t R1, uint32_t R2, uint32_t R3, bool PAIR = true> struct Chirpz3Cfg {
  static constexpr bool VEC2 = sizeof(T) == 4 && PAIR;  // (PAIR = false: kernels_regfft.h, one f32 transform per workgroup)
  using P = typename ChirpzSelect<VEC2, Pk2, T>::type;
  static constexpr uint32_t NV = VEC2 ? 2u : 1u;
  static constexpr uint32_t M = R1 * R2 * R3, LA = R2 * R3, LB = R1 * R3, LC = R1 * R2;  // lanes of the stages
  static constexpr uint32_t LMAX = LA > LB ? (LA > LC ? LA : LC) : (LB > LC ? LB : LC);
  static constexpr uint32_t NT = (LMAX + 63u) / 64u * 64u;
  static constexpr uint32_t S1 = chirpz3_pitch_runs(LB, R3);  // exchange 1: rows j2, reader lane k1*R3 + j3, writer runs of R3
  static constexpr uint32_t S2 = LC | 1u;                      // exchange 2: rows j3, reader lane a, writer pitch walk
  static constexpr uint32_t S3 = chirpz3_pitch_runs(LB, R1);  // exchange 3: rows k2, reader lane c3*R1 + k1, writer runs of R1
  static constexpr uint32_t S4 = LA | 1u;                      // exchange 4: rows k1, reader lane m, writer pitch walk
  static constexpr uint32_t X12 = R2 * S1 > R3 * S2 ? R2 * S1 : R3 * S2, X34 = R2 * S3 > R1 * S4 ? R2 * S3 : R1 * S4;
  static constexpr uint32_t XE = X12 > X34 ? X12 : X34;
  static constexpr size_t SMEM = (size_t)XE * sizeof(cpx<P>);
  static constexpr uint32_t HALF = M / 2 + 1, NROW = (HALF + LA - 1) / LA;  // rows j1 (positions m + R2R3*j1) that can hold user data
  static constexpr uint32_t LDS_WG = (uint32_t)((160u * 1024u) / SMEM);
  static constexpr uint32_t WAVES = LDS_WG * (NT / 64u) / 4u;  // per SIMD, as far as the LDS goes
  static constexpr uint32_t MINW = WAVES < 1u ? 1u : (WAVES > 3u ? 3u : WAVES);
};
// y[r] *= tab[r * stride] for first <= r < R (swap: re <-> im afterwards), the loads in batches of TB, each issued one batch ahead of its use
template <typename P, typename T, uint32_t R, uint32_t TB> __device__ __forceinline__ void chirpz_table_product(cpx<P>* y, const cpx<T>* tab, uint32_t stride, bool swap) {
  constexpr uint32_t NB = (R + TB - 1) / TB;
  cpx<T> t[2][TB];
#pragma unroll
  for (uint32_t i = 0; i < TB; ++i)
    if (i < R) t[0][i] = tab[i * stride];
#pragma unroll
  for (uint32_t b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
#pragma unroll
      for (uint32_t i = 0; i < TB; ++i)
        if ((b + 1) * TB + i < R) t[(b + 1) & 1u][i] = tab[((b + 1) * TB + i) * stride];
    }
    FOURIER_SCHED_FENCE();
#pragma unroll
    for (uint32_t i = 0; i < TB; ++i)
      if (b * TB + i < R) {
        const cpx<P> z = cmul_tab(y[b * TB + i], t[b & 1u][i]);
        y[b * TB + i] = swap ? cpx<P>{z.im, z.re} : z;
      }
    FOURIER_SCHED_FENCE();
  }
}

template <typename T, uint32_t R1, uint32_t R2, uint32_t R3>
__global__ void __launch_bounds__((Chirpz3Cfg<T, R1, R2, R3>::NT), (Chirpz3Cfg<T, R1, R2, R3>::MINW)) chirpz_reg3_kernel(ChirpzArgs a) {
  using C = Chirpz3Cfg<T, R1, R2, R3>;
  using P = typename C::P;
  using LV = LaneVal<P, T>;
  constexpr uint32_t NV = C::NV, LA = C::LA, LB = C::LB, LC = C::LC, S1 = C::S1, S2 = C::S2, S3 = C::S3, S4 = C::S4, NROW = C::NROW;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>), XB = (uint32_t)sizeof(cpx<P>), OOB = 0xfffffff0u, TB = 8u;
  FOURIER_DYN_SMEM(smem);
  cpx<P>* xb = (cpx<P>*)smem;
  const uint32_t t = threadIdx.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * NV;
  const uint32_t n = (uint32_t)a.n;
  const uint32_t nb = a.batch - b0 < NV ? (uint32_t)(a.batch - b0) : NV;
  const BufRsrc rin = make_rsrc((const cpx<T>*)a.in + b0 * n, nb * n * EB), rout = make_rsrc((cpx<T>*)a.out + b0 * n, nb * n * EB);
  const BufRsrc rc = make_rsrc(a.chirp, n * EB);
  const cpx<T>* t1 = (const cpx<T>*)a.tw;     // [j2 < R2][lane k1*R3 + j3]: W_M^{(j3 + R3*j2) * k1}
  const cpx<T>* t2 = t1 + (size_t)R2 * LB;    // [k2 < R2][j3 < R3]: W_{R2R3}^{j3 * k2}
  const cpx<T>* t3 = t2 + (size_t)R2 * R3;    // [c3 < R3][a < R1R2]: W_M^{a * c3}
  const cpx<T>* t4 = t3 + (size_t)R3 * LC;    // [c2 < R2][k1 < R1]: W_{R1R2}^{k1 * c2}
  const cpx<T>* wt = (const cpx<T>*)a.w;

  // ---- stage A: lane m = t
  if (t < LA) {
    cpx<P> x[R1];
    cpx<T> d[NV][NROW], ch[NROW];
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) {
      const uint32_t pos = t + LA * j1;
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) d[v][j1] = buf_load_elem<T>(rin, pos < n ? (v * n + pos) * EB : OOB);
    }
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) ch[j1] = buf_load_elem<T>(rc, (t + LA * j1) * EB);  // zero beyond n
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) {
      if (j1 < NROW) {
        T re[NV], im[NV];
#pragma unroll
        for (uint32_t v = 0; v < NV; ++v) { re[v] = d[v][j1].re; im[v] = d[v][j1].im; }
        cpx<P> val{LV::make(re), LV::make(im)};
        if (a.swap) val = {val.im, val.re};
        x[j1] = cmul_tab(val, ch[j1]);  // work = x (.) in, zero padded (bluesteins.rs:229-234)
      } else {
        x[j1] = cpx<P>{(P)0, (P)0};
      }
    }
    dft_any<P, (int)R1>(x);
    const uint32_t j2 = t / R3, j3 = t - j2 * R3;
#pragma unroll
    for (uint32_t k1 = 0; k1 < R1; ++k1) {
      cpx<P>* p = xb + j2 * S1 + k1 * R3 + j3;
      LDS_NOTE(p, XB, true, 330);
      *p = x[k1];
    }
  }
  __syncthreads();
  // ---- stage B: lane k1*R3 + j3
  cpx<P> y[R2];
  if (t < LB) {
#pragma unroll
    for (uint32_t j2 = 0; j2 < R2; ++j2) {
      const cpx<P>* p = xb + j2 * S1 + t;
      LDS_NOTE(p, XB, false, 331);
      y[j2] = *p;
    }
    chirpz_table_product<P, T, R2, TB>(y, t1 + t, LB, false);
    dft_any<P, (int)R2>(y);
    FOURIER_SCHED_FENCE();
    chirpz_table_product<P, T, R2, TB>(y, t2 + t % R3, R3, false);
  }
  __syncthreads();  // exchange 1 is read
  if (t < LB) {
    const uint32_t k1 = t / R3, j3 = t - k1 * R3;
#pragma unroll
    for (uint32_t k2 = 0; k2 < R2; ++k2) {
      cpx<P>* p = xb + j3 * S2 + k1 + R1 * k2;
      LDS_NOTE(p, XB, true, 332);
      *p = y[k2];
    }
  }
  __syncthreads();
  // ---- stage C, (.) w, swap, stage C': lane a = k1 + R1*k2
  cpx<P> z[R3];
  if (t < LC) {
#pragma unroll
    for (uint32_t j3 = 0; j3 < R3; ++j3) {
      const cpx<P>* p = xb + j3 * S2 + t;
      LDS_NOTE(p, XB, false, 333);
      z[j3] = *p;
    }
    dft_any<P, (int)R3>(z);  // z[k3] = Y[a + R1R2 * k3]
    FOURIER_SCHED_FENCE();
    chirpz_table_product<P, T, R3, TB>(z, wt + t, LC, true);  // (.) w, swap for the inverse transform (bluesteins.rs:236-239)
    dft_any<P, (int)R3>(z);  // over k3: z[c3]
    FOURIER_SCHED_FENCE();
    chirpz_table_product<P, T, R3, TB>(z, t3 + t, LC, false);
  }
  __syncthreads();  // exchange 2 is read
  if (t < LC) {
    const uint32_t k2 = t / R1, k1 = t - k2 * R1;
#pragma unroll
    for (uint32_t c3 = 0; c3 < R3; ++c3) {
      cpx<P>* p = xb + k2 * S3 + c3 * R1 + k1;
      LDS_NOTE(p, XB, true, 334);
      *p = z[c3];
    }
  }
  __syncthreads();
  // ---- stage B': lane c3*R1 + k1
  if (t < LB) {
#pragma unroll
    for (uint32_t k2 = 0; k2 < R2; ++k2) {
      const cpx<P>* p = xb + k2 * S3 + t;
      LDS_NOTE(p, XB, false, 335);
      y[k2] = *p;
    }
    dft_any<P, (int)R2>(y);  // y[c2]
    FOURIER_SCHED_FENCE();
    chirpz_table_product<P, T, R2, TB>(y, t4 + t % R1, R1, false);
  }
  __syncthreads();  // exchange 3 is read
  if (t < LB) {
    const uint32_t c3 = t / R1, k1 = t - c3 * R1;
#pragma unroll
    for (uint32_t c2 = 0; c2 < R2; ++c2) {
      cpx<P>* p = xb + k1 * S4 + c3 + R3 * c2;
      LDS_NOTE(p, XB, true, 336);
      *p = y[c2];
    }
  }
  __syncthreads();
  // ---- stage A': lane m = c3 + R3*c2; output position m + R2R3 * c1
  if (t < LA) {
    cpx<P> v1[R1];
    cpx<T> ch[NROW];
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) ch[j1] = buf_load_elem<T>(rc, (t + LA * j1) * EB);
#pragma unroll
    for (uint32_t k1 = 0; k1 < R1; ++k1) {
      const cpx<P>* p = xb + k1 * S4 + t;
      LDS_NOTE(p, XB, false, 337);
      v1[k1] = *p;
    }
    dft_any<P, (int)R1>(v1);
    const T scale = (T)a.scale;
#pragma unroll
    for (uint32_t j1 = 0; j1 < NROW; ++j1) {  // (positions from M/2 + 1 on lie beyond the user array)
      const uint32_t pos = t + LA * j1;
      cpx<P> o{v1[j1].im, v1[j1].re};  // the inverse inner transform's trailing swap (1/M is folded into w)
      o = cmul_tab(o, ch[j1]);         // bluesteins.rs:240-258
      if (a.swap) o = {o.im, o.re};
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v)
        buf_store_elem<T>(rout, pos < n ? (v * n + pos) * EB : OOB, cpx<T>{LV::get(o.re, v) * scale, LV::get(o.im, v) * scale});
    }
  }
}

}  // namespace fourier_hip
