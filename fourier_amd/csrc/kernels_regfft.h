// kernels_regfft.h -- a length with factors 5 ... 13 as a DIRECT transform in one launch on the register stages of kernels_chirpz.h (round 6,
// sessions 48 - 49): N = R1 x R2 (x R3), one Cooley-Tukey step between register-resident transforms, ONE LDS exchange per step.
//
// The reference sends every length that is not 2^a 3^b to Bluestein (fourier/src/lib.rs:38-42: Autosort::new fails, autosort/mod.rs:24-46).
// Here such lengths had the LDS mixed-radix kernels (kernels_mixed.h): a per-length kernel for the common ones, the runtime-parameterised
// kernel for the rest (factors 7 / 11 / 13: 0.25 ... 0.35 of the HBM peak, one LDS round trip per SMALL radix, four to six per transform).
// regfft_kernel / regfft3_kernel are ahead-of-time kernels per length (regfft_shapes.h, kernels_regfft.cpp) that keep the lane layout, the
// packed f32 arithmetic (two transforms per lane) and the exchange layouts of the chirp-z kernels' forward half.  Tolerance-only route like
// every length beyond 2^a 3^b (include/fourier.h); 2^a 3^b keep the reference's own schedule (bit-identical to the CPU restatement).
#pragma once
#include "kernels_chirpz.h"

namespace fourier_hip {

// n = j2 + R2*j1: DFT_R1 over j1 -> k1; W_N^{j2*k1}; DFT_R2 over j2 -> k2: X[k1 + R1*k2] -- loads coalesced along j2, stores along k1, ONE exchange.
// Inverse = swap . DFT . swap at the load and the store; the five scalings on the store (fft.rs:4-16).
template <typename T, uint32_t R1, uint32_t R2>
__global__ void __launch_bounds__(64, (ChirpzRegCfg<T, R1, R2>::MINW)) regfft_kernel(ChirpzArgs a) {
  using C = ChirpzRegCfg<T, R1, R2>;
  using P = typename C::P;
  using LV = LaneVal<P, T>;
  constexpr uint32_t GPW = C::GPW, TPW = C::TPW, NV = C::NV, P1 = C::P1, N = R1 * R2;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>), XB = (uint32_t)sizeof(cpx<P>);
  FOURIER_DYN_SMEM(smem);
  const uint32_t lane = threadIdx.x, c = lane / R1, q = lane - c * R1;
  const bool active = c < GPW;
  const uint64_t b0 = (uint64_t)blockIdx.x * TPW;
  const uint32_t nb = a.batch - b0 < TPW ? (uint32_t)(a.batch - b0) : TPW;
  const BufRsrc rin = make_rsrc((const cpx<T>*)a.in + b0 * N, nb * N * EB), rout = make_rsrc((cpx<T>*)a.out + b0 * N, nb * N * EB);
  cpx<P>* xb = (cpx<P>*)smem + (active ? c : 0u) * (R1 * P1);
  const uint32_t tbase = c * NV * N;
  if (active && q < R2) {
    cpx<P> x[R1];
    cpx<T> d[NV][R1];
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1)
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) d[v][j1] = buf_load_elem<T>(rin, (tbase + v * N + q + R2 * j1) * EB);
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) {
      T re[NV], im[NV];
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) { re[v] = d[v][j1].re; im[v] = d[v][j1].im; }
      x[j1] = a.swap ? cpx<P>{LV::make(im), LV::make(re)} : cpx<P>{LV::make(re), LV::make(im)};
    }
    dft_any<P, (int)R1>(x);
#pragma unroll
    for (uint32_t k1 = 0; k1 < R1; ++k1) {
      cpx<P>* p = xb + k1 * P1 + q;
      LDS_NOTE(p, XB, true, 340);
      *p = x[k1];
    }
  }
  __syncthreads();
  if (active) {
    cpx<P> y[R2];
#pragma unroll
    for (uint32_t j2 = 0; j2 < R2; ++j2) {
      const cpx<P>* p = xb + q * P1 + j2;
      LDS_NOTE(p, XB, false, 341);
      y[j2] = *p;
    }
    chirpz_table_product<P, T, R2, 8u>(y, (const cpx<T>*)a.tw + q, R1, false);
    dft_any<P, (int)R2>(y);
    const T scale = (T)a.scale;
#pragma unroll
    for (uint32_t k2 = 0; k2 < R2; ++k2) {
      const cpx<P> o = a.swap ? cpx<P>{y[k2].im, y[k2].re} : y[k2];
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v)
        buf_store_elem<T>(rout, (tbase + v * N + q + R1 * k2) * EB, cpx<T>{LV::get(o.re, v) * scale, LV::get(o.im, v) * scale});
    }
  }
}

// N = R1 x R2 x R3: a workgroup per transform (f32: per pair), lanes and exchanges as the forward half of chirpz_reg3_kernel; X[a + R1R2*k3] leaves
// stage C's lanes a = k1 + R1*k2 coalesced.  SPLIT: the two exchanges carry the real parts, then the imaginary parts, through a buffer of HALF the
// size (two more barriers each): a workgroup holds N x 8 bytes of LDS instead of N x 16, so that a second one fits beside it -- one loads while
// the other computes and stores -- where the registers allow it too (154 - 168 of them: three waves per SIMD, i.e. workgroups of at most six
// waves); 1.2 - 1.33 x there, 2 - 5 % slower elsewhere: regfft_shapes.h names the variant per length (sessions 53, 55, 56).
// PAIR = false (f32): ONE transform per workgroup on scalar arithmetic -- half the registers and half the LDS of the packed pair, so that two
// workgroups of seven or eight waves share a compute unit (the packed arithmetic saves nothing here: the waves issue VALU work 13 % of their cycles).
template <typename T, uint32_t R1, uint32_t R2, uint32_t R3, bool SPLIT, bool PAIR = true> struct Regfft3Cfg : Chirpz3Cfg<T, R1, R2, R3, PAIR> {
  using B = Chirpz3Cfg<T, R1, R2, R3, PAIR>;
  using P = typename B::P;
  static constexpr uint32_t X12 = R2 * B::S1 > R3 * B::S2 ? R2 * B::S1 : R3 * B::S2;  // the forward exchanges only
  static constexpr size_t SMEM = (size_t)X12 * (SPLIT ? sizeof(P) : sizeof(cpx<P>));
  static constexpr uint32_t LDS_WG = (uint32_t)((160u * 1024u) / SMEM);
  static constexpr uint32_t WAVES = LDS_WG * (B::NT / 64u) / 4u;  // per SIMD, as far as the LDS goes
  // (unpaired f32: no bound -- the kernels take 70 ... 100 registers; any bound here is turned into one on whole workgroups and spills)
  static constexpr uint32_t MINW = !PAIR || WAVES < 1u ? 1u : (WAVES > 3u ? 3u : WAVES);
};
// one exchange: the writer lanes (t < LW) put their R values at widx(r), the reader lanes (t < LR) take theirs from ridx(r)
template <bool SPLIT, typename P, uint32_t RW, uint32_t RR, typename WI, typename RI>
__device__ __forceinline__ void regfft_exchange(void* smem, bool writer, bool reader, const cpx<P>* w, cpx<P>* r, WI widx, RI ridx, int note) {
  if constexpr (!SPLIT) {
    cpx<P>* xb = (cpx<P>*)smem;
    if (writer) {
#pragma unroll
      for (uint32_t i = 0; i < RW; ++i) {
        cpx<P>* p = xb + widx(i);
        LDS_NOTE(p, (uint32_t)sizeof(cpx<P>), true, note);
        *p = w[i];
      }
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (uint32_t i = 0; i < RR; ++i) {
        const cpx<P>* p = xb + ridx(i);
        LDS_NOTE(p, (uint32_t)sizeof(cpx<P>), false, note + 1);
        r[i] = *p;
      }
    }
  } else {
    P* xs = (P*)smem;
    if (writer) {
#pragma unroll
      for (uint32_t i = 0; i < RW; ++i) {
        P* p = xs + widx(i);
        LDS_NOTE(p, (uint32_t)sizeof(P), true, note);
        *p = w[i].re;
      }
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (uint32_t i = 0; i < RR; ++i) {
        const P* p = xs + ridx(i);
        LDS_NOTE(p, (uint32_t)sizeof(P), false, note + 1);
        r[i].re = *p;
      }
    }
    __syncthreads();
    if (writer) {
#pragma unroll
      for (uint32_t i = 0; i < RW; ++i) xs[widx(i)] = w[i].im;
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (uint32_t i = 0; i < RR; ++i) r[i].im = xs[ridx(i)];
    }
  }
}

// FACT: the twiddle between stages A and B, W_N^{(j3 + R3 j2) k1}, as W_{R1R2}^{j2 k1} (R1 R2 entries, the lanes of one k1 share an address) before
// DFT_R2 and W_N^{j3 k1} (one entry per lane) after it: tables of R1R2 + R1R3 + R2R3 entries that stay in the L1 instead of N + R2R3 entries
// streamed from the L2 beside the data (a third of a compute unit's read traffic at 8000 points); R2 more complex multiplies per lane.
template <typename T, uint32_t R1, uint32_t R2, uint32_t R3, bool SPLIT, bool FACT, bool PAIR = true>
__global__ void __launch_bounds__((Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>::NT), (Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>::MINW)) regfft3_kernel(ChirpzArgs a) {
  using C = Regfft3Cfg<T, R1, R2, R3, SPLIT, PAIR>;
  using P = typename C::P;
  using LV = LaneVal<P, T>;
  constexpr uint32_t NV = C::NV, LA = C::LA, LB = C::LB, LC = C::LC, S1 = C::S1, S2 = C::S2, N = C::M;
  constexpr uint32_t EB = (uint32_t)sizeof(cpx<T>), TB = 8u;
  FOURIER_DYN_SMEM(smem);
  const uint32_t t = threadIdx.x;
  const uint64_t b0 = (uint64_t)blockIdx.x * NV;
  const uint32_t nb = a.batch - b0 < NV ? (uint32_t)(a.batch - b0) : NV;
  const BufRsrc rin = make_rsrc((const cpx<T>*)a.in + b0 * N, nb * N * EB), rout = make_rsrc((cpx<T>*)a.out + b0 * N, nb * N * EB);
  // whole table: [j2 < R2][lane k1*R3 + j3]: W_N^{(j3 + R3*j2) * k1}; FACT: [j2 < R2][k1 < R1]: W_{R1R2}^{j2 * k1}, then [lane k1*R3 + j3]: W_N^{j3 * k1}
  const cpx<T>* t1 = (const cpx<T>*)a.tw;
  const cpx<T>* tb = t1 + (size_t)R1 * R2;
  const cpx<T>* t2 = FACT ? tb + LB : t1 + (size_t)R2 * LB;  // [k2 < R2][j3 < R3]: W_{R2R3}^{j3 * k2}
  cpx<P> x[R1], y[R2], z[R3];
  if (t < LA) {
    cpx<T> d[NV][R1];
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1)
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) d[v][j1] = buf_load_elem<T>(rin, (v * N + t + LA * j1) * EB);
#pragma unroll
    for (uint32_t j1 = 0; j1 < R1; ++j1) {
      T re[NV], im[NV];
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v) { re[v] = d[v][j1].re; im[v] = d[v][j1].im; }
      x[j1] = a.swap ? cpx<P>{LV::make(im), LV::make(re)} : cpx<P>{LV::make(re), LV::make(im)};
    }
    dft_any<P, (int)R1>(x);
  }
  {  // exchange 1: rows j2, the reader's lane k1*R3 + j3
    const uint32_t j2 = t / R3, j3 = t - j2 * R3;
    regfft_exchange<SPLIT, P, R1, R2>(smem, t < LA, t < LB, x, y, [&](uint32_t k1) { return j2 * S1 + k1 * R3 + j3; },
                                      [&](uint32_t r) { return r * S1 + t; }, 350);
  }
  if (t < LB) {
    if constexpr (FACT) {
      const cpx<T> base = tb[t];
      chirpz_table_product<P, T, R2, TB>(y, t1 + t / R3, R1, false);
      dft_any<P, (int)R2>(y);
      FOURIER_SCHED_FENCE();
      chirpz_table_product<P, T, R2, TB>(y, t2 + t % R3, R3, false);
#pragma unroll
      for (uint32_t k2 = 0; k2 < R2; ++k2) y[k2] = cmul_tab(y[k2], base);
    } else {
      chirpz_table_product<P, T, R2, TB>(y, t1 + t, LB, false);
      dft_any<P, (int)R2>(y);
      FOURIER_SCHED_FENCE();
      chirpz_table_product<P, T, R2, TB>(y, t2 + t % R3, R3, false);
    }
  }
  __syncthreads();  // exchange 1 is read
  {  // exchange 2: rows j3, the reader's lane k1 + R1*k2
    const uint32_t k1 = t / R3, j3 = t - k1 * R3;
    regfft_exchange<SPLIT, P, R2, R3>(smem, t < LB, t < LC, y, z, [&](uint32_t k2) { return j3 * S2 + k1 + R1 * k2; },
                                      [&](uint32_t r) { return r * S2 + t; }, 352);
  }
  if (t < LC) {
    dft_any<P, (int)R3>(z);
    const T scale = (T)a.scale;
#pragma unroll
    for (uint32_t k3 = 0; k3 < R3; ++k3) {
      const cpx<P> o = a.swap ? cpx<P>{z[k3].im, z[k3].re} : z[k3];
#pragma unroll
      for (uint32_t v = 0; v < NV; ++v)
        buf_store_elem<T>(rout, (v * N + t + LC * k3) * EB, cpx<T>{LV::get(o.re, v) * scale, LV::get(o.im, v) * scale});
    }
  }
}

}  // namespace fourier_hip
