// kernels_common.h -- gfx950 device code of the batched 1D c2c FFT engine: what every kernel family shares (16-byte
// units, buffer-descriptor accesses, the argument block of the pass kernels, the register butterflies).
//
// What it computes (reference: fourier-algorithms/src/autosort/mod.rs:203-284, one Stockham
// autosort pass `out[j + R*s*i + s*k] = W_size^{i*k} * DFT_R(in[j + s*i + s*m*k'])_k`) -- but with
// a *big* radix R = L in {16..2048} per HBM round trip instead of the reference's 2/3/4/8, so that
// N = 2^20 needs 2 sweeps of HBM instead of the reference's 7 (+2 copies).  Inside a pass the
// L-point DFT of every column is itself a Stockham autosort of radix 16 x R2 x R3 (mod.rs:20-21
// radix schedule idea, re-derived for a 64-wide wavefront): each thread keeps 16 points of VEC
// adjacent columns in registers, does the radix-16/8/4/2 butterflies there
// (autosort/butterfly.rs:3-65 equivalents), and exchanges through LDS between stages.
//
// Data layout: interleaved complex (re,im), AoS, exactly the reference's Complex<T>
// (fourier-ffi/include/fourier.h:10-11,23-24).  A "unit" is 16 bytes = VEC complex numbers of
// adjacent columns (VEC=2 for f32, 1 for f64): every global access of the column-tile modes is one
// 16-byte unit per lane, 128-byte segments per tile row.
//
// Inverse transforms use IDFT(x) = swap(DFT(swap(x))) with swap = exchange re<->im, applied at the
// first load / last store, so all twiddle tables are forward-only.
//
// Kernel families (one header each; DESIGN.md section 2 says which sizes take which):
//   kernels_pass.h: fft_pass_kernel, fft_conv_kernel      kernels_onelaunch.h: fft_twolevel_kernel, bluestein_*_kernel
//   kernels_small.h: tiny_*_kernel     kernels_mixed.h: mixed_radix_kernel[_ct]     kernels_misc.h: odd_last_kernel,
//   stockham_pass_kernel, blu_*_kernel     kernels_experiments.h: fft_l2fused_kernel, fft_last_split_kernel (experiments build)
//   fft_pass_kernel<T, L, CG, MODE, IO>   one big-radix pass over column tiles (FIRST / MID / LAST) or whole rows (ROWS)
//   fft_conv_kernel<T, L, CG>             Bluestein middle: last forward pass, (.) w, first inverse pass in one launch
//   fft_twolevel_kernel<T, L1, L2>        2^11..2^15: both passes inside one workgroup
//   bluestein_small_kernel / bluestein_rows_kernel   whole chirp-z in one launch for M <= 2^15
//   tiny_shfl_kernel<T, N>                N <= 16 (f32: 32): one lane per transform, wave-shuffle unit transpose
//   mixed_radix_kernel_ct<T, N>           2^a*3^b in LDS with the reference's schedule, one instantiation per length; also
//                                         every 2^a*3^b*5^c and 7^k (radices 5, 7: beyond the reference, which takes Bluestein)
//   mixed_radix_kernel<T, MAXP, PPT, NT>  the same passes, runtime-parameterised: the other lengths with factors 5..13 up to 8192 points
//   odd_last_kernel<T, R>                 radix-3/9/27 passes (twiddled middle ones and the final one) of the large 2^a*3^b sizes
//   stockham_pass_kernel<T, R>            one pass in global memory, any radix and stride: 2^a*3^b with a < 12 beyond the LDS limit
//   blu_pre_kernel / blu_post_kernel      unfused chirp sweeps (option bluestein_fusion = 0)
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif
#include "kernel_args.h"

#ifdef FOURIER_EMU
#define FOURIER_SCHED_FENCE()
#define FOURIER_WAIT_VMEM()
#define FOURIER_LAUNDER(v)
#define FOURIER_DYN_SMEM(name) unsigned char* name = hipemu::smem()
#define LDS_NOTE(p, bytes, w, site) hipemu::lds_note((p), (bytes), (w), (site))
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline uint32_t mul24(uint32_t a, uint32_t b) { return a * b; }
#else
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
// stops hipcc from hoisting a whole block of table loads above the arithmetic that consumes them
// (it otherwise keeps all 16 twiddle units live at once and spills under the 128-VGPR budget)
#define FOURIER_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// every global access this wave has issued (loads AND stores: gfx9 counts both on vmcnt) has completed at the L2
#define FOURIER_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// hides a per-lane value from the optimiser: inside a persistent loop it keeps everything derived from the value from
// being hoisted out of the loop (and spilled there) -- a few VALU instructions per iteration instead
#define FOURIER_LAUNDER(v) asm volatile("" : "+v"(v))
#ifdef FOURIER_RTC_LDS_BYTES
// a kernel specialised at run time (rtc.cpp) knows its LDS footprint when it is compiled: a static array, because a module
// function cannot be given more than 64 KiB of DYNAMIC LDS (hipFuncSetAttribute does not take a hipFunction_t)
#define FOURIER_DYN_SMEM(name) __shared__ __attribute__((aligned(16))) unsigned char name[FOURIER_RTC_LDS_BYTES]
#else
#define FOURIER_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
#define LDS_NOTE(p, bytes, w, site)
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence; callers correct the quotient with a compare
static __device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_mul_u32_u24: full rate (v_mul_lo_u32 runs at a quarter); both operands must be below 2^24
static __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#endif

// second __launch_bounds__ argument = min waves per SIMD: ask for two workgroups per CU
// (2*NT/64 waves over 4 SIMDs), which caps the kernel at 128 VGPRs for NT = 512.
#define FOURIER_MIN_WAVES(NT) ((NT) >= 1024 ? 4 : ((NT) >= 256 ? (NT) / 128 : 1))

// ABLATION builds (timing experiments, WRONG results: FOURIER_ABLATE = 1 no butterflies / twiddles, 2 = additionally no LDS exchange --
// pure load -> store --, 3 = no inter-pass twiddle only, 4 = stage twiddles from a constant, 5 = per-thread inter-pass factor from a
// constant) exist only as experiments translation units (-DFOURIER_EXPERIMENTS_TU: kernels_skeleton.cpp, the abl* builds of
// tools/build_variants.py); every other translation unit compiles the arithmetic in whatever its command line says.
#if !defined(FOURIER_EXPERIMENTS_TU) || !defined(FOURIER_ABLATE)
#undef FOURIER_ABLATE
#define FOURIER_ABLATE 0
#endif
// A translation unit that compiles these templates DIFFERENTLY -- an ablation -- gets them in an inline namespace of its own: the same
// template names then are different entities from those of the other translation units (no two definitions of one entity in a library
// that links both), while every unqualified use stays as it is.
#if FOURIER_ABLATE != 0
#define FOURIER_KERNELS_BEGIN namespace fourier_hip { inline namespace ablated {
#define FOURIER_KERNELS_END } }
#else
#define FOURIER_KERNELS_BEGIN namespace fourier_hip {
#define FOURIER_KERNELS_END }
#endif

FOURIER_KERNELS_BEGIN

template <typename T> struct alignas(16) Unit16 { T a[16 / sizeof(T)]; };  // VEC interleaved complex
template <typename T> struct alignas(8) Unit8 { T a[8 / sizeof(T)]; };     // one plane of VEC columns

// Settled by A/B on the GPU (the knobs they were are profiles/r06_removed_ab_knobs.patch):
//   cache policy: the data loads of every pass are non-temporal (each element is read once per pass; -10 % on the last pass of the
//   2^20 plan, r01 session 8), and so are the final stores (+1 %) and the intermediate stores of passes up to L = 1024 and of the
//   narrow-tile first passes of length 2048 / 4096 (+2 %, -3..-6 % on those passes; the one-workgroup-per-CU L = 2048 passes lose 8 %
//   with them) -- PassPolicy in kernels_pass.h;
//   SPLIT_THRESHOLD: exchange buffers above this many bytes are exchanged as two planes (re, im; f32 packed builds: two columns):
//   half the LDS per workgroup, twice the workgroups per CU (16 KiB measured best over 2^8..2^20, r01 sweep);
//   the shortest whole-transform kernels (f32 64, f64 32) move their data between global memory and registers through LDS
//   (16-byte units, whole lines per instruction) instead of element accesses that cover 32 bytes of a line per instruction.
constexpr size_t SPLIT_THRESHOLD = 16 * 1024;

// 16-byte global accesses (global_load_dwordx4 / global_store_dwordx4)
template <typename T, bool NT> __device__ __forceinline__ Unit16<T> load_unit(const void* p) {
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4u v;
  if constexpr (NT) v = __builtin_nontemporal_load((const v4u*)p);
  else v = *(const v4u*)p;
  Unit16<T> u;
  __builtin_memcpy(&u, &v, 16);
  return u;
#else
  return *(const Unit16<T>*)p;
#endif
}
template <typename T, bool NT> __device__ __forceinline__ void store_unit(void* p, const Unit16<T>& u) {
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4u v;
  __builtin_memcpy(&v, &u, 16);
  if constexpr (NT) __builtin_nontemporal_store(v, (v4u*)p);
  else *(v4u*)p = v;
#else
  *(Unit16<T>*)p = u;
#endif
}

// Cache policy of a pass's data accesses.  POL_SC1 (loads only) = `buffer_load_dwordx4 ... sc1`: bypasses this CU's L1 and
// is served by the XCD's L2 -- how a workgroup reads what ANOTHER workgroup of the same XCD stored a moment ago (the
// L2 is the coherence point of an XCD; a CU's L1 is never refreshed by other CUs' stores, MI355X_MICROARCH.md).
enum { POL_PLAIN = 0, POL_NT = 1, POL_SC1 = 2 };
// 16-byte accesses through a buffer descriptor built over a wave-uniform base pointer (buffer_load/store_dwordx4):
// the address is base + soff (SGPR) + voff (one 32-bit VGPR per lane) -- no 64-bit per-lane pointers -- and the
// hardware bounds-checks voff, dword by dword, against the descriptor's byte count: out-of-range dwords load as 0 and
// are not stored (soff is NOT part of the check).  The Bluestein end passes use exactly that for the zero padding
// behind the user array (bluesteins.rs:229-234) and for dropping the outputs beyond it (bluesteins.rs:240-258).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, uint32_t bytes = 0x7fffffffu) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
enum { BUF_PLAIN = 0, BUF_NT = 2, BUF_SC1 = 16 };  // aux bits of the gfx940+ buffer instructions
template <typename T, int AUX = BUF_PLAIN> __device__ __forceinline__ Unit16<T> buf_load_unit(BufRsrc r, uint32_t voff, uint32_t soff = 0) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX);
  Unit16<T> u;
  __builtin_memcpy(&u, &v, 16);
  return u;
}
// Stores take NO scalar offset, on purpose: `buffer_store_dwordx4 v[a:a+3], voff, rsrc, sN offen` followed a few
// instructions later by VALU writes to v[a:a+3] (hipcc reuses the data registers of consecutive stores, and inserts its
// wait state only for the soffset-less form) corrupted the stored data of lanes 12-15 of every 16 on gfx950 under load
// (2^14 / 2^15 one-launch plans, round 3; profiles/r03_s2_store_soffset_hazard.txt).  A row offset therefore goes into
// the descriptor base (scalar adds) or into voff.
template <typename T, int AUX = BUF_PLAIN> __device__ __forceinline__ void buf_store_unit(BufRsrc r, uint32_t voff, const Unit16<T>& u) {
  decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) v;
  __builtin_memcpy(&v, &u, 16);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, AUX);
}
// one complex element (8 / 16 bytes) through a descriptor, bounds-checked like the units
template <typename T> __device__ __forceinline__ cpx<T> buf_load_elem(BufRsrc r, uint32_t voff, uint32_t soff = 0) {
  cpx<T> y;
  if constexpr (sizeof(T) == 4) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    __builtin_memcpy(&y, &v, 8);
  } else {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    __builtin_memcpy(&y, &v, 16);
  }
  return y;
}
template <typename T, int AUX = BUF_PLAIN> __device__ __forceinline__ void buf_store_elem(BufRsrc r, uint32_t voff, const cpx<T>& y) {
  if constexpr (sizeof(T) == 4) {
    decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0)) v;
    __builtin_memcpy(&v, &y, 8);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, 0, AUX);
  } else {
    decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)) v;
    __builtin_memcpy(&v, &y, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, AUX);
  }
}
template <typename T> __device__ __forceinline__ Unit16<T> load_unit_sc1(BufRsrc r, uint32_t off) { return buf_load_unit<T, BUF_SC1>(r, off); }

// one complex element (8 / 16 bytes), optionally non-temporal
template <typename T, bool NT> __device__ __forceinline__ void store_elem(cpx<T>* p, const cpx<T>& y) {
#ifndef FOURIER_EMU
  if constexpr (NT) {
    typedef T v2 __attribute__((ext_vector_type(2)));
    v2 v = {y.re, y.im};
    __builtin_nontemporal_store(v, (v2*)p);
  } else {
    *p = y;
  }
#else
  *p = y;
#endif
}

// 16-byte accesses to arrays that are only 8-byte aligned (f32 user arrays of odd length inside a batch):
// global_load/store_dwordx4 need dword alignment only.
template <typename T, bool STREAM = false> __device__ __forceinline__ Unit16<T> load_unit_a8(const void* p) {  // STREAM: read-once data
  Unit16<T> u;
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  v4u v;
  if constexpr (STREAM) {
    typedef v4u v4u_a8 __attribute__((aligned(8)));
    v = __builtin_nontemporal_load((const v4u_a8*)p);
  } else {
    struct __attribute__((packed, aligned(8))) V { v4u v; };
    v = ((const V*)p)->v;
  }
  __builtin_memcpy(&u, &v, 16);
#else
  __builtin_memcpy(&u, p, 16);
#endif
  return u;
}
template <typename T, bool STREAM = false> __device__ __forceinline__ void store_unit_a8(void* p, const Unit16<T>& u) {
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  if constexpr (STREAM) {
    typedef v4u v4u_a8 __attribute__((aligned(8)));
    v4u v;
    __builtin_memcpy(&v, &u, 16);
    __builtin_nontemporal_store(v, (v4u_a8*)p);
  } else {
    struct __attribute__((packed, aligned(8))) V { v4u v; };
    V w;
    __builtin_memcpy(&w.v, &u, 16);
    *(V*)p = w;
  }
#else
  __builtin_memcpy(p, &u, 16);
#endif
}

// Tile index of workgroup blk of a grid (mixed-length tile passes).  Consecutive workgroups go to different XCDs (blockIdx % 8); with
// chunk = G > 0 every run of 8 * G workgroups is dealt out so that XCD x takes G NEIGHBOURING tiles: tiles next to each other in a row share
// the 128-byte lines their row segments straddle whenever a row stride is not a multiple of 128 bytes (44100 = 210 x 210: 1680 bytes) and
// then meet in one L2, close in time.  The last grid % (8 * G) workgroups keep their index.
__device__ __forceinline__ uint32_t xcd_chunked(uint32_t blk, uint32_t grid, uint32_t chunk) {
  const uint32_t span = 8u * chunk;
  if (chunk == 0u || blk >= grid - grid % span) return blk;
  const uint32_t r = blk % span;
  return blk - r + (r & 7u) * chunk + (r >> 3);
}
template <typename T> __device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T> __device__ __forceinline__ void bf2(cpx<T>& a, cpx<T>& b) {
  const cpx<T> t = a;
  a = {t.re + b.re, t.im + b.im};
  b = {t.re - b.re, t.im - b.im};
}
template <typename T> __device__ __forceinline__ cpx<T> mul_neg_i(cpx<T> z) { return {z.im, -z.re}; }

// ---- small forward DFTs, natural order in and out (W = exp(-2*pi*i/R)) ----
template <typename T> __device__ __forceinline__ void dft2(cpx<T>* x) { bf2(x[0], x[1]); }

template <typename T> __device__ __forceinline__ void dft4(cpx<T>& x0, cpx<T>& x1, cpx<T>& x2, cpx<T>& x3) {
  bf2(x0, x2);
  bf2(x1, x3);
  x3 = mul_neg_i(x3);
  bf2(x0, x1);  // x0 = X0, x1 = X2
  bf2(x2, x3);  // x2 = X1, x3 = X3
  const cpx<T> t = x1; x1 = x2; x2 = t;
}
template <typename T> __device__ __forceinline__ void dft4(cpx<T>* x) { dft4(x[0], x[1], x[2], x[3]); }

template <typename T> __device__ __forceinline__ void dft8(cpx<T>* x) {
  const T c = (T)0.70710678118654752440;
  dft4(x[0], x[2], x[4], x[6]);  // E0..E3 in x0,x2,x4,x6
  dft4(x[1], x[3], x[5], x[7]);  // O0..O3 in x1,x3,x5,x7
  x[3] = {c * (x[3].re + x[3].im), c * (x[3].im - x[3].re)};   // * W8^1
  x[5] = mul_neg_i(x[5]);                                       // * W8^2
  x[7] = {c * (x[7].im - x[7].re), -c * (x[7].re + x[7].im)};  // * W8^3
  bf2(x[0], x[1]);  // X0, X4
  bf2(x[2], x[3]);  // X1, X5
  bf2(x[4], x[5]);  // X2, X6
  bf2(x[6], x[7]);  // X3, X7
  const cpx<T> y1 = x[2], y2 = x[4], y3 = x[6], y4 = x[1], y5 = x[3], y6 = x[5];
  x[1] = y1; x[2] = y2; x[3] = y3; x[4] = y4; x[5] = y5; x[6] = y6;
}

template <typename T> __device__ __forceinline__ void dft16(cpx<T>* x) {
  const T c1 = (T)0.92387953251128675613;  // cos(pi/8)
  const T s1 = (T)0.38268343236508977173;  // sin(pi/8)
  const T c2 = (T)0.70710678118654752440;
  // n = a + 4b : DFT over b for each a; result kb stored at slot a + 4*kb
#pragma unroll
  for (int a = 0; a < 4; ++a) dft4(x[a], x[a + 4], x[a + 8], x[a + 12]);
  // twiddle W16^{a*kb}
  x[5] = cmul(x[5], cpx<T>{c1, -s1});                                   // a=1,kb=1: W^1
  x[9] = {c2 * (x[9].re + x[9].im), c2 * (x[9].im - x[9].re)};         // a=1,kb=2: W^2
  x[13] = cmul(x[13], cpx<T>{s1, -c1});                                 // a=1,kb=3: W^3
  x[6] = {c2 * (x[6].re + x[6].im), c2 * (x[6].im - x[6].re)};         // a=2,kb=1: W^2
  x[10] = mul_neg_i(x[10]);                                            // a=2,kb=2: W^4
  x[14] = {c2 * (x[14].im - x[14].re), -c2 * (x[14].re + x[14].im)};   // a=2,kb=3: W^6
  x[7] = cmul(x[7], cpx<T>{s1, -c1});                                   // a=3,kb=1: W^3
  x[11] = {c2 * (x[11].im - x[11].re), -c2 * (x[11].re + x[11].im)};   // a=3,kb=2: W^6
  x[15] = cmul(x[15], cpx<T>{-c1, s1});                                 // a=3,kb=3: W^9
  // DFT over a for each kb; result ka at slot ka + 4*kb holds X[kb + 4*ka]
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) dft4(x[4 * kb], x[4 * kb + 1], x[4 * kb + 2], x[4 * kb + 3]);
  // transpose 4x4 to natural order
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const cpx<T> t = x[a + 4 * b]; x[a + 4 * b] = x[b + 4 * a]; x[b + 4 * a] = t;
    }
}

// 32 points in registers: two 16-point DFTs over the even and odd inputs, then one radix-2 combine
template <typename T> __device__ __forceinline__ void dft32(cpx<T>* x) {
  cpx<T> e[16], o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { e[i] = x[2 * i]; o[i] = x[2 * i + 1]; }
  dft16(e);
  dft16(o);
  const cpx<T> w[16] = {{(T)1.00000000000000000000, (T)-0.00000000000000000000}, {(T)0.98078528040323043058, (T)-0.19509032201612824808}, {(T)0.92387953251128673848, (T)-0.38268343236508978178}, {(T)0.83146961230254523567, (T)-0.55557023301960217765}, {(T)0.70710678118654757274, (T)-0.70710678118654746172}, {(T)0.55557023301960228867, (T)-0.83146961230254523567}, {(T)0.38268343236508983729, (T)-0.92387953251128673848}, {(T)0.19509032201612833135, (T)-0.98078528040323043058}, {(T)0.00000000000000006123, (T)-1.00000000000000000000}, {(T)-0.19509032201612819257, (T)-0.98078528040323043058}, {(T)-0.38268343236508972627, (T)-0.92387953251128673848}, {(T)-0.55557023301960195560, (T)-0.83146961230254545772}, {(T)-0.70710678118654746172, (T)-0.70710678118654757274}, {(T)-0.83146961230254534669, (T)-0.55557023301960217765}, {(T)-0.92387953251128673848, (T)-0.38268343236508989280}, {(T)-0.98078528040323043058, (T)-0.19509032201612860891}};  // W32^k
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const cpx<T> t = (k == 0) ? o[0] : cmul(o[k], w[k]);
    x[k] = {e[k].re + t.re, e[k].im + t.im};
    x[k + 16] = {e[k].re - t.re, e[k].im - t.im};
  }
}

template <typename T, int R> __device__ __forceinline__ void dft_r(cpx<T>* x) {
  if constexpr (R == 2) dft2(x);
  else if constexpr (R == 4) dft4(x);
  else if constexpr (R == 8) dft8(x);
  else if constexpr (R == 16) dft16(x);
  else if constexpr (R == 32) dft32(x);
}


FOURIER_KERNELS_END  // namespace fourier_hip
