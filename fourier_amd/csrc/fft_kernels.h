// fft_kernels.h -- gfx950 device code of the batched 1D c2c FFT engine.
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
// Kernel families in this file (DESIGN.md section 2 says which sizes take which):
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
#include <stdint.h>

#ifdef FOURIER_EMU
#define FOURIER_SCHED_FENCE()
#define FOURIER_WAIT_VMEM()
#define FOURIER_LAUNDER(v)
#define FOURIER_DYN_SMEM(name) unsigned char* name = hipemu::smem()
#define LDS_NOTE(p, bytes, w, site) hipemu::lds_note((p), (bytes), (w), (site))
static inline float fast_rcp(float x) { return 1.0f / x; }
static inline uint32_t mul24(uint32_t a, uint32_t b) { return a * b; }
#else
#include <hip/hip_runtime.h>
// stops hipcc from hoisting a whole block of table loads above the arithmetic that consumes them
// (it otherwise keeps all 16 twiddle units live at once and spills under the 128-VGPR budget)
#define FOURIER_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// every global access this wave has issued (loads AND stores: gfx9 counts both on vmcnt) has completed at the L2
#define FOURIER_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// hides a per-lane value from the optimiser: inside a persistent loop it keeps everything derived from the value from
// being hoisted out of the loop (and spilled there) -- a few VALU instructions per iteration instead
#define FOURIER_LAUNDER(v) asm volatile("" : "+v"(v))
#define FOURIER_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define LDS_NOTE(p, bytes, w, site)
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence; callers correct the quotient with a compare
static __device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_mul_u32_u24: full rate (v_mul_lo_u32 runs at a quarter); both operands must be below 2^24
static __device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#endif

// second __launch_bounds__ argument = min waves per SIMD: ask for two workgroups per CU
// (2*NT/64 waves over 4 SIMDs), which caps the kernel at 128 VGPRs for NT = 512.
#ifndef FOURIER_MIN_WAVES
#define FOURIER_MIN_WAVES(NT) ((NT) >= 1024 ? 4 : ((NT) >= 256 ? (NT) / 128 : 1))
#endif

namespace fourier_hip {

template <typename T> struct cpx { T re, im; };
template <typename T> struct alignas(16) Unit16 { T a[16 / sizeof(T)]; };  // VEC interleaved complex
template <typename T> struct alignas(8) Unit8 { T a[8 / sizeof(T)]; };     // one plane of VEC columns

enum { MODE_FIRST = 0, MODE_MID = 1, MODE_LAST = 2, MODE_ROWS = 3 };
// Bluestein fusion (bluesteins.rs:229-258): IO_BLU_IN = the first pass of the forward inner FFT reads the
// USER array (length blu_n, zero padded to n) times the chirp x; IO_BLU_OUT = the last pass of the inverse
// inner FFT writes the first blu_n points times the chirp (and the user scaling) into the USER array.
enum { IO_PLAIN = 0, IO_BLU_IN = 1, IO_BLU_OUT = 2 };

// Build-time knobs (tools/build_variants.py A/B-tests them on the GPU):
//   FOURIER_NT_LOAD  = 2 (default): the data loads of every pass are non-temporal (each element is read once per
//   pass; -10% on the last pass of the 2^20 plan, r01 session 8); 1 = first pass only, 0 = none
//   FOURIER_NT_STORE = 2 (default): output stores are non-temporal -- the final pass's (+1%), and the intermediate
//   ones of passes up to L = 1024 (+2%; the one-workgroup-per-CU L = 2048 passes lose 8% with them); 1 = final only
//   FOURIER_ABLATE (timing experiments only, results are wrong): 1 = no butterflies / twiddles,
//   2 = additionally no LDS exchange (pure load -> store), 3 = no inter-pass twiddle only
#ifndef FOURIER_ABLATE
#define FOURIER_ABLATE 0
#endif
//   FOURIER_SPLIT_THRESHOLD: exchange buffers above this many bytes are exchanged as two planes (re, im):
//   half the LDS per workgroup, twice the workgroups per CU (16 KiB measured best over 2^8..2^20, r01 sweep)
//   FOURIER_ROWS_STAGED: the shortest whole-transform kernels (f32 64, f64 32) move their data between global memory and
//     registers through LDS (16-byte units, whole lines per instruction) instead of element accesses that cover
//     32 bytes of a line per instruction.
#ifndef FOURIER_ROWS_STAGED
#define FOURIER_ROWS_STAGED 1
#endif
#ifndef FOURIER_SPLIT_THRESHOLD
#define FOURIER_SPLIT_THRESHOLD (16 * 1024)
#endif
#ifndef FOURIER_NT_LOAD
#define FOURIER_NT_LOAD 2
#endif
#ifndef FOURIER_NT_STORE
#define FOURIER_NT_STORE 2
#endif

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
template <typename T> __device__ __forceinline__ Unit16<T> load_unit_a8(const void* p) {
  Unit16<T> u;
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(8))) V { v4u v; };
  const v4u v = ((const V*)p)->v;
  __builtin_memcpy(&u, &v, 16);
#else
  __builtin_memcpy(&u, p, 16);
#endif
  return u;
}
template <typename T> __device__ __forceinline__ void store_unit_a8(void* p, const Unit16<T>& u) {
#ifndef FOURIER_EMU
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  struct __attribute__((packed, aligned(8))) V { v4u v; };
  V w;
  __builtin_memcpy(&w.v, &u, 16);
  *(V*)p = w;
#else
  __builtin_memcpy(p, &u, 16);
#endif
}

// Kernel argument block (passed by value).
struct PassArgs {
  const void* in;
  void* out;
  const void* tw1;    // [Q][16]  W_L^{th*k}            (stage-1 twiddles)
  const void* tw2;    // [R3][16] W_Q^{i*k}             (stage-2 twiddles, only when R3 > 1)
  const void* tw_lo;  // W_size^{e},          e < 2^lo_bits     } two-level table of the
  const void* tw_hi;  // W_size^{h<<lo_bits}, h < size>>lo_bits } inter-pass twiddle W_size^{i*k}
  const void* tw_half;  // split tiles: W_2L^{n}, n < L (the radix-2 decimation-in-frequency twiddle in front of a length-L tile)
  const void* mul;    // Bluestein kernels (conv / one-launch): the transformed chirp w, indexed like the M-point spectrum
  uint64_t n;         // elements per transform (batch stride)
  uint64_t cn;        // columns of this pass = n / L
  uint64_t s;         // Stockham stride = product of the previous passes' lengths (a power of two for the tile passes)
  uint32_t s_shift;   // log2(s)
  uint64_t tiles;     // column tiles per transform = cn / COLS
  uint64_t total_cols;  // ROWS mode: number of transforms in this launch
  uint32_t lo_bits;
  uint32_t nxcd;      // >1: remap blockIdx so that each XCD (blockIdx % nxcd) walks a contiguous tile range
  uint32_t xcd_interleave;  // block -> tile mapping mode, see xcd_remap()
  const void* blu_x;  // Bluestein chirp table x[0..blu_n) (IO_BLU_IN / IO_BLU_OUT)
  // chirp-in pass WITHOUT the n-entry chirp table (a quarter of that pass's HBM-side traffic when read, PMC round 3):
  // index k = row*cn + b, so x[k] = W_2n^{k^2} = blu_p[row] * blu_u[b] * W_n^{cn*row*b}; the cross term splits like the
  // pass's own inter-pass twiddle into a per-thread factor and a per-tile LDS table, both from a two-level table of n-th
  // roots with EXACT integer exponents (f64 products below 2^53).  blu_p == nullptr selects the table read.
  const void* blu_p;     // [L/2]  W_2n^{(row*cn)^2 mod 2n}
  const void* blu_u;     // [cn]   W_2n^{b^2 mod 2n}
  const void* tn_lo;     // W_n^{e},            e < 2^tn_bits      } two-level table of n-th roots
  const void* tn_hi;     // W_n^{h << tn_bits}, h <= n >> tn_bits  }
  uint32_t tn_bits;
  uint32_t blu_cn_mod;   // cn mod n
  uint32_t blu_cnq_mod;  // (cn * Q) mod n
  double blu_nd, blu_inv_nd;  // n and 1/n as doubles
  uint64_t blu_n;     // user transform length (batch stride of the user-side buffer)
  int blu_swap;       // user-level inverse: swap re/im of the user data
  int swap_in, swap_out;
  double scale;       // applied on the final store (LAST / ROWS)
};

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
  static constexpr bool SPLIT = (size_t)UNITS * 16 > FOURIER_SPLIT_THRESHOLD;  // exchange re and im planes separately
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
  static constexpr bool ROWS_STAGED = (FOURIER_ROWS_STAGED != 0) && Q >= 2 && Q * 2 * (int)sizeof(T) <= 32;
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
  template <int LAYOUT> static __device__ __forceinline__ int unit_index(int pos, int cg) {
    if constexpr (LAYOUT == 1 && CG <= 8) return (pos * CG + cg) ^ (((pos >> 4) & (16 / CG - 1)) * CG);
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
template <typename T, int B, typename Ld, typename Use>
__device__ __forceinline__ void units_batched(const Ld& ld, const Use& use) {
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += B) {
    Unit16<T> u[B];
#pragma unroll
    for (int q = 0; q < B; ++q) u[q] = ld(r0 + q);
    FOURIER_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < B; ++q) use(r0 + q, u[q]);
    FOURIER_SCHED_FENCE();
  }
}

// x[v][k] *= t[k], k = 1..15 (t[0] = 1): the stage twiddles of one thread, sixteen consecutive table entries.  All of
// them (f64: half of them) are loaded in one batch -- under FOURIER_STAGE_TW_BATCHED; otherwise hipcc picks the grouping
#ifndef FOURIER_STAGE_TW_BATCH
#define FOURIER_STAGE_TW_BATCH 8
#endif
template <typename T, int VEC>
__device__ __forceinline__ void stage_twiddle(cpx<T> (&x)[VEC][16], const cpx<T>* t) {
#if FOURIER_STAGE_TW_BATCH > 0
  constexpr int PER = 16 / (int)sizeof(cpx<T>), NU = 16 / PER, B = FOURIER_STAGE_TW_BATCH < NU ? FOURIER_STAGE_TW_BATCH : NU;
#pragma unroll
  for (int u0 = 0; u0 < NU; u0 += B) {
    Unit16<T> u[B];
#pragma unroll
    for (int q = 0; q < B; ++q) u[q] = *(const Unit16<T>*)(t + (u0 + q) * PER);
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
#else
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const cpx<T> w = t[k];
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[v][k] = cmul(x[v][k], w);
  }
#endif
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
      constexpr int LAY1 = (C::SPLIT && !IN_ROWS && !(MODE == MODE_FIRST && R3 == 1)) ? 1 : 0;
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
        lds_exchange<T, L, CG, 0>(x, smem, cg, [=](int r) { return jw + 16 * (16 * iw + r); }, th_r, cg_r, 4);
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
#ifdef FOURIER_AB_NO_CHIRP  // timing experiment only (wrong results): the chirp is not read
template <typename T> __device__ __forceinline__ Unit16<T> ab_ones() { Unit16<T> u; for (int i = 0; i < (int)(16 / sizeof(T)); ++i) u.a[i] = (i & 1) ? (T)0 : (T)1; return u; }
#define FOURIER_AB_CHIRP_LOAD(rc, off) ab_ones<T>()
#else
#define FOURIER_AB_CHIRP_LOAD(rc, off) buf_load_unit<T>(rc, off)
#endif
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

#ifndef FOURIER_BLU_OUT_ST_NT
#define FOURIER_BLU_OUT_ST_NT 0
#endif
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

  // ---- inter-pass twiddle table for this tile: tabU[col][r] = W_size^{i_col * Q * r}
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
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      for (int u = tid; u < HALF * L / VEC; u += C::NT) {
        const uint32_t eh = (uint32_t)u * VEC, e = (uint32_t)(h * HALF * L) + eh;
        Unit16<T> w{};
        if (e < stage_valid) w = load_unit_a8<T>(src + e);
        *(Unit16<T>*)(stage + (eh / L) * LP + (eh % L)) = w;
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
      for (int r = 0; r < 8; ++r) c[r] = FOURIER_AB_CHIRP_LOAD(rc, voff + (uint32_t)r * rowb);
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
  if constexpr (TWIDDLED && FOURIER_ABLATE == 0) {
    if constexpr (Q == 1) __syncthreads();  // tabU visibility when there was no exchange barrier
    const cpx<T>* tabU = (const cpx<T>*)(smem + C::TABU_OFF);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint64_t i = (MODE == MODE_FIRST) ? c0 + (uint64_t)(cg * VEC + v) : c0 >> a.s_shift;
      const cpx<T> base = two_level_twiddle<T>(a, i * (uint64_t)th);
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
      for (int u = tid; u < HALF * L / VEC; u += C::NT) {
        const uint32_t eh = (uint32_t)u * VEC, e = (uint32_t)(h * HALF * L) + eh;
        if (e < stage_valid) store_unit_a8<T>(dst + e, *(const Unit16<T>*)(stage + (eh / L) * LP + (eh % L)));
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
    for (int r = 0; r < 8; ++r) c[r] = FOURIER_AB_CHIRP_LOAD(rc, voff + (uint32_t)r * rowb);
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
      buf_store_unit<T, FOURIER_BLU_OUT_ST_NT ? STAUX : BUF_PLAIN>(ro, voff + (uint32_t)r * rowb, u);
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

#ifndef FOURIER_NT_STORE_NARROW_2048
#define FOURIER_NT_STORE_NARROW_2048 1  // streaming intermediate stores also for the narrow-tile (two workgroups per CU) first passes of length 2048 / 4096: -3..-6 % on that pass (profiles/r03_s8_*.jsonl); 0 = plain
#endif
// default cache policy of the stand-alone pass kernels (see the FOURIER_NT_* notes at the top of this file)
template <int L, int MODE, int CG = 8> struct PassPolicy {
  static constexpr bool FINAL = (MODE == MODE_LAST || MODE == MODE_ROWS);
  // 64-byte-wide tiles (CG = 4): two workgroups share every 128-byte line, the second one must find it in the L2, so
  // no streaming hint (L = 2048 first pass: 6.5 vs 7.6 ms per 1024 transforms of 2^21, r01 session 11)
  static constexpr int LD = (MODE != MODE_ROWS && CG < 8) ? POL_PLAIN
                            : ((MODE == MODE_FIRST && FOURIER_NT_LOAD != 0) || FOURIER_NT_LOAD == 2) ? POL_NT : POL_PLAIN;
  // MODE_ROWS: a wave's element stores only form whole lines for L >= 256; below that they rely on L2
  // write-combining and a non-temporal hint is a 2-6x loss (N = 16..64, r01 session 9)
  static constexpr int ST = (MODE == MODE_ROWS ? (FOURIER_NT_STORE != 0 && L >= 256)
                             : (FINAL ? FOURIER_NT_STORE != 0 : (FOURIER_NT_STORE == 2 && (L <= 1024 || (FOURIER_NT_STORE_NARROW_2048 && CG < 8))))) ? POL_NT : POL_PLAIN;
};

// last pass of length 2L on half tiles (pass_tile, SPLIT = 1): grid = 2 x batch x tiles
#ifndef FOURIER_SPLIT_LD
#define FOURIER_SPLIT_LD POL_PLAIN  // the second reader of a line must find it in the L2: no streaming hint on the loads
#endif
template <typename T, int L, int CG, int IO = IO_PLAIN>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_last_split_kernel(PassArgs a) {
  FOURIER_DYN_SMEM(smem);
  pass_tile<T, L, CG, MODE_LAST, IO, FOURIER_SPLIT_LD, PassPolicy<L, MODE_LAST>::ST, 1>(a, blockIdx.x, gridDim.x, smem, (int)threadIdx.x);
}

template <typename T, int L, int CG, int MODE, int IO = IO_PLAIN>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_pass_kernel(PassArgs a) {
  FOURIER_DYN_SMEM(smem);
  pass_tile<T, L, CG, MODE, IO, PassPolicy<L, MODE, CG>::LD, PassPolicy<L, MODE, CG>::ST>(a, blockIdx.x, gridDim.x, smem, (int)threadIdx.x);
}


// ---- N = L1 x L2 with BOTH passes in one launch and the intermediate parked in the XCD's own L2 --------------------
// (2^16 .. 2^18 in f32, 2^15 .. 2^17 in f64: N * sizeof(complex) <= 2 MiB.)  The two-launch plan moves every point
// through HBM twice; here a transform is read from HBM once (pass A = the FIRST pass) and written once (pass B = the
// LAST pass), and the transposed intermediate between them lives in a small window that is written and read back by
// workgroups of ONE XCD, so it never leaves that XCD's 4 MiB L2 (measured with tools/membench.py --l2x: a window of
// <= 1 MiB per XCD that is written with plain stores and read back with sc1 loads costs nothing next to the HBM
// streams: 5.84 vs 5.85 TB/s; profiles/r02_membench.jsonl).
//
// Persistent workgroups, data-flow scheduling, no team barrier.  Every workgroup reads the id of the XCD it runs
// on (HW_REG_XCC_ID) and pulls work items from THAT XCD's queue, so all items of one transform are executed on one
// XCD whatever the dispatcher did (placement is observed, never assumed).  The queue of XCD x is the sequence, for
// step s = 0, 1, ...: the tiles of pass B of its local transform s - 1, then the tiles of pass A of local transform s
// (older work first: with depth = 1 pass A of s reuses the window pass B of s - 1 is reading).  The workgroup that draws (A, s, tile 0) claims the next global transform from one device-wide counter and
// publishes it (map[s]); XCDs therefore share the batch dynamically and any number of resident workgroups per XCD
// (even one) completes the job.  An item waits only for items drawn EARLIER from the same queue (its transform's
// claim; pass B: all tiles of pass A; pass A, just before its stores: the readers of the window slot's previous
// tenant, `depth` steps back), every drawn item is held by a running workgroup, hence no deadlock.  Waits are nevertheless bounded
// (spin_limit) and raise ctrl[1] instead of hanging the device.
// Visibility: producer = plain stores, every wave waits vmcnt(0) (the stores have reached the XCD's L2), workgroup
// barrier, then one relaxed agent-scope increment; consumer = one lane polls the counter (relaxed, sc1), workgroup
// barrier, then sc1 loads, which are served by that same L2.
struct FusedArgs {
  PassArgs a, b;       // pass A / pass B arguments; a.in, a.out, b.in, b.out are set per item
  const void* in;      // user input  (batch stride a.n)
  void* out;           // user output (may equal in: a transform is read completely before any of it is written)
  void* window;        // [16 XCC ids][depth][n] intermediates
  uint32_t* ctrl;      // control block, zeroed before every launch (layout below)
  uint32_t batch, depth, tiles_a, tiles_b, spin_limit;
};
// ctrl: [0] next global transform, [1] abort flag; queue of XCC id x at FUSED_CTRL_HDR + x * fused_ctrl_stride(batch):
//       [0] next item, [16 + j] map[j] (0 = unclaimed, 0xffffffff = batch exhausted, else transform + 1),
//       [16 + cap + j] done_a[j], [16 + 2*cap + j] done_b[j], cap = batch + 2
enum { FUSED_CTRL_HDR = 16, FUSED_XCC_IDS = 16 };
__host__ __device__ inline uint64_t fused_ctrl_stride(uint64_t batch) { return 16 + 3 * (batch + 2); }
__host__ __device__ inline uint64_t fused_ctrl_words(uint64_t batch) { return FUSED_CTRL_HDR + FUSED_XCC_IDS * fused_ctrl_stride(batch); }

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lane 0 only: wait until *p >= target (or *p != 0 when target == 0); false = gave up (abort flag raised)
__device__ __forceinline__ bool fused_wait(const uint32_t* p, uint32_t target, uint32_t* abort_flag, uint32_t limit, uint32_t* seen) {
  for (uint32_t spins = 0;; ++spins) {
    const uint32_t v = ld_relaxed(p);
    if (target ? v >= target : v != 0) { *seen = v; return true; }
    if (spins >= limit || ((spins & 63) == 63 && ld_relaxed(abort_flag))) {
      __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

#ifndef FOURIER_FUSED_MIN_WAVES
// three 256-thread workgroups per CU (<= 168 VGPRs): at four (<= 128) the two pass bodies spill 56-116 bytes per lane
// and every size measured slower (profiles/r02_s3_plan4096_conv_and_fused_ab.jsonl)
#define FOURIER_FUSED_MIN_WAVES 3
#endif
struct FusedWindowFree {
  const uint32_t* counter;  // done_b of the slot's previous tenant, or null when the slot has never been used
  uint32_t target;
  uint32_t* abort_flag;
  uint32_t limit;
  int tid;
  __device__ __forceinline__ void operator()() const {
    if (!counter) return;  // wave-uniform
    if (tid == 0) {
      uint32_t seen;
      (void)fused_wait(counter, target, abort_flag, limit, &seen);  // on give-up the abort flag is up: every later wait bails out
    }
    __syncthreads();
  }
};

template <typename T, int L1, int CG1, int L2, int CG2>
__global__ void __launch_bounds__((L1 / 16) * CG1, FOURIER_FUSED_MIN_WAVES) fft_l2fused_kernel(FusedArgs f) {
  using CA = TileCfg<T, L1, CG1>;
  using CB = TileCfg<T, L2, CG2>;
  static_assert(CA::NT == CB::NT, "both passes run on the same workgroup");
  constexpr size_t SMEM_A = CA::smem_bytes(MODE_FIRST), SMEM_B = CB::smem_bytes(MODE_LAST);
  constexpr size_t SLOT = ((SMEM_A > SMEM_B ? SMEM_A : SMEM_B) + 15) & ~(size_t)15;  // broadcast words behind the tiles' LDS
  FOURIER_DYN_SMEM(smem);
  volatile uint32_t* bc = (volatile uint32_t*)(smem + SLOT);
  const int tid = (int)threadIdx.x;
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (FUSED_XCC_IDS - 1);  // HW_REG_XCC_ID[3:0]
  uint32_t* const q = f.ctrl + FUSED_CTRL_HDR + (uint64_t)xcc * fused_ctrl_stride(f.batch);
  const uint64_t cap = (uint64_t)f.batch + 2;
  uint32_t* const map = q + 16;
  uint32_t* const done_a = map + cap;
  uint32_t* const done_b = done_a + cap;
  uint32_t* const abort_flag = f.ctrl + 1;
  const uint32_t per_step = f.tiles_a + f.tiles_b;
  cpx<T>* const win0 = (cpx<T>*)f.window + (uint64_t)xcc * f.depth * f.a.n;

  for (;;) {
    // ---- draw an item; lane 0 resolves its transform and waits for what the item depends on
    if (tid == 0) {
      const uint32_t item = __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t s = item / per_step, r = item % per_step;
      const bool is_a = r >= f.tiles_b;  // within a step: pass B of the previous transform first, then pass A of this one
      const uint32_t tile = is_a ? r - f.tiles_b : r;
      uint32_t g = 0xffffffffu, j = is_a ? s : s - 1;
      int act = 0;  // 0 skip, 1 run, 2 exit
      if (!is_a && s == 0) {
        act = 0;  // there is no transform -1
      } else if (j >= cap) {
        act = 2;
      } else {
        uint32_t v = 0;
        bool ok = true;
        if (is_a && tile == 0) {
          const uint32_t t = __hip_atomic_fetch_add(f.ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = t < f.batch ? t + 1 : 0xffffffffu;
          __hip_atomic_store(map + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          ok = fused_wait(map + j, 0, abort_flag, f.spin_limit, &v);
        }
        if (!ok) act = 2;
        else if (v == 0xffffffffu) act = is_a ? 0 : 2;  // the batch is exhausted: nothing after this pass-B item exists
        else {
          uint32_t seen;
          if (!is_a) ok = fused_wait(done_a + j, f.tiles_a, abort_flag, f.spin_limit, &seen);  // pass A waits later, see WindowFree
          act = ok ? 1 : 2;
          g = v - 1;
        }
      }
      bc[0] = (uint32_t)act; bc[1] = g; bc[2] = j; bc[3] = (is_a ? 0u : 0x80000000u) | tile;
    }
    __syncthreads();
    const uint32_t act = bc[0], g = bc[1], j = bc[2], kt = bc[3];
    __syncthreads();  // everyone has read the slot before lane 0 of the next iteration rewrites it
    if (act == 2) return;
    if (act == 0) continue;
    const bool is_a = (kt >> 31) == 0;
    const uint32_t tile = kt & 0x7fffffffu;
    cpx<T>* const win = win0 + (uint64_t)(j % f.depth) * f.a.n;
    int tid_i = tid;
    FOURIER_LAUNDER(tid_i);
    if (is_a) {
      PassArgs a = f.a;
      a.in = (const cpx<T>*)f.in + (uint64_t)g * f.a.n;
      a.out = win;
      // the window slot's previous tenant (local transform j - depth) must have been read completely -- checked only
      // now, with this tile's data already loaded and transformed in registers
      const FusedWindowFree hook{j >= f.depth ? done_b + (j - f.depth) : nullptr, f.tiles_b, abort_flag, f.spin_limit, tid};
      pass_tile<T, L1, CG1, MODE_FIRST, IO_PLAIN, POL_NT, POL_PLAIN, 0, FusedWindowFree>(a, tile, f.tiles_a, smem, tid_i, hook);
      FOURIER_WAIT_VMEM();  // this wave's window stores have reached the L2
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(done_a + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      PassArgs b = f.b;
      b.in = win;
      b.out = (cpx<T>*)f.out + (uint64_t)g * f.a.n;
      pass_tile<T, L2, CG2, MODE_LAST, IO_PLAIN, POL_SC1, POL_NT>(b, tile, f.tiles_b, smem, tid_i);
      __syncthreads();  // every wave holds its window data in registers by now (the tile's LDS exchanges waited for it)
      if (tid == 0) __hip_atomic_fetch_add(done_b + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifndef FOURIER_CONV_MIN_WAVES
#define FOURIER_CONV_MIN_WAVES(NT) FOURIER_MIN_WAVES(NT)
#endif
// A/B knobs of the conv kernel (tools/build_variants.py): non-temporal stores / non-temporal loads of the w table
#ifndef FOURIER_CONV_ST_NT
#define FOURIER_CONV_ST_NT 1  // with the XCD-sliced tile order of launch_conv: 4.5 vs 4.75 ms (C4), 7.8 vs 8.3 ms (N = 65537), r02 session 3
#endif
#ifndef FOURIER_CONV_W_NT
#define FOURIER_CONV_W_NT 0
#endif
#ifndef FOURIER_CONV_W_BATCH
#define FOURIER_CONV_W_BATCH 8  // loads of the w table in flight per thread
#endif
// ---- Bluestein middle (bluesteins.rs:236-239): LAST pass of the forward inner FFT, (.) w, and FIRST pass of
// the inverse inner FFT in ONE launch.  The last forward pass (R = L, s = M/L) leaves X[j + (M/L)*k] of its
// column tile in registers; an inverse FFT whose first pass has the same length (R = L, s = 1, m = M/L) reads
// exactly those elements as its columns i = j, so the M-point spectrum never goes back to HBM: one read and
// one write of the work array instead of two of each.  The inverse is swap . DFT . swap (mod.rs:366-387):
// the leading swap happens here, the trailing one in the inverse plan's last pass.
template <typename T, int L, int CG>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_CONV_MIN_WAVES((L / 16) * CG)) fft_conv_kernel(PassArgs a) {
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

  // inter-pass twiddle table of the inverse FFT's first pass: tabU[col][r] = W_M^{i_col * Q * r}
  cpx<T>* tabU = (cpx<T>*)(smem + C::TABU_OFF);
  for (int idx = tid; idx < COLS * 16; idx += C::NT) {
    const uint64_t i = c0 + (uint64_t)(idx >> 4);
    tabU[idx] = two_level_twiddle<T>(a, i * (uint64_t)(Q * (idx & 15)));
  }

  // forward LAST pass: rows th + Q*r (stride cn) of columns c0 + cg*VEC + v
  cpx<T> x[VEC][16];
  int th = tid / CG, cg = tid % CG;
  {
    const BufRsrc rs = make_rsrc(in);
    const uint32_t voff = (uint32_t)(((uint64_t)th * a.cn + c0 + (uint64_t)(cg * VEC)) * sizeof(cpx<T>));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // tiles narrower than a 128-byte line share every line with a sibling workgroup: no streaming hint then
      const Unit16<T> u = buf_load_unit<T, (FOURIER_NT_LOAD == 2 && CG >= 8) ? BUF_NT : BUF_PLAIN>(rs, voff, (uint32_t)r * rowb);
#pragma unroll
      for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
    }
  }
  tile_core<T, L, CG, MODE_LAST>(x, th, cg, tid, smem, (const cpx<T>*)a.tw1, (const cpx<T>*)a.tw2);
  // register r holds X[c + cn*(th + Q*r)]: (.) w (FFT'd chirp, 1/M folded in), then the inverse's leading swap
  {
    int t = tid;
    FOURIER_LAUNDER(t);
    const BufRsrc rw = make_rsrc(a.mul);
    const uint32_t voff = (uint32_t)(((uint64_t)(t / CG) * a.cn + c0 + (uint64_t)((t % CG) * VEC)) * sizeof(cpx<T>));
    units_batched<T, FOURIER_CONV_W_BATCH>(
        [&](int r) {
#ifdef FOURIER_AB_NO_W
          Unit16<T> u1; for (int i = 0; i < (int)(16 / sizeof(T)); ++i) u1.a[i] = (i & 1) ? (T)0 : (T)1; return u1;
#else
          return buf_load_unit<T, FOURIER_CONV_W_NT != 0 ? BUF_NT : BUF_PLAIN>(rw, voff, (uint32_t)r * rowb);
#endif
        },
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
  // transposed store: column i's L outputs are contiguous
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    cpx<T>* p = out + (c0 + (uint64_t)(cg * VEC + v)) * L + th;
#pragma unroll
    for (int r = 0; r < 16; ++r) store_elem<T, FOURIER_CONV_ST_NT != 0>(p + Q * r, x[v][r]);
  }
}

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

#ifndef FOURIER_TWOLEVEL_TW_BATCH
#define FOURIER_TWOLEVEL_TW_BATCH(NT) ((NT) <= 128 ? 4 : 8)  // loads of the inter-pass twiddle table in flight per thread (2-wave workgroups live on occupancy: stay under 128 VGPRs)
#endif
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
    units_batched<T, FOURIER_TWOLEVEL_TW_BATCH(Q1 * CG1)>(
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
  constexpr int RB = 4;  // rows per batch: RB chirp values and RB * VEC data elements in flight per thread
  constexpr uint32_t ES = (uint32_t)sizeof(cpx<T>);
  cpx<T> x[VEC][16];
  {
    const int th = tid % Q, cg = tid / Q;
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += RB) {
      cpx<T> c[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) c[q] = buf_load_elem<T>(rc, (uint32_t)(th + Q * (r0 + q)) * ES);  // 0 beyond n
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int q = 0; q < RB; ++q)
          x[v][r0 + q] = buf_load_elem<T>(ri, ((uint32_t)(cg * VEC + v) * n + (uint32_t)(th + Q * (r0 + q))) * ES);
      FOURIER_SCHED_FENCE();
#pragma unroll
      for (int v = 0; v < VEC; ++v)
#pragma unroll
        for (int q = 0; q < RB; ++q) {  // bluesteins.rs:229-234; positions n .. L-1 are padding: the chirp loaded there is zero
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
  for (int r0 = 0; r0 < 16; r0 += RB) {
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
      const Unit16<T> u = buf_load_unit<T, FOURIER_NT_LOAD != 0 ? BUF_NT : BUF_PLAIN>(ri, voff, (uint32_t)((Q1 * r) * L2 * sizeof(cpx<T>)));
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
#ifdef FOURIER_TWOLEVEL_STORE_SOFF  // A/B only: reproduces the corruption described at buf_store_unit
    __builtin_amdgcn_raw_buffer_store_b128(*(const decltype(__builtin_amdgcn_raw_buffer_load_b128(ro, 0, 0, 0))*)&u, ro, (int)voff,
                                           (int)((Q2 * r) * L1 * sizeof(cpx<T>)), FOURIER_NT_STORE != 0 ? BUF_NT : BUF_PLAIN);
#else
    buf_store_unit<T, FOURIER_NT_STORE != 0 ? BUF_NT : BUF_PLAIN>(make_rsrc(obase + (Q2 * r) * L1), voff, u);
#endif
  }
}

// ---- whole Bluestein chirp-z (bluesteins.rs:215-259) in ONE launch for M = L1 x L2 <= 2^15 ----
// work = x (.) in (zero padded to M) -> FFT_M -> (.) w -> IFFT_M -> (.) x (.) scale, all on the
// register-resident M-point array of one workgroup: the forward two-level core, then the same core with the
// roles of L1 and L2 exchanged (its input layout is the other's output layout).  HBM sees the N-point user
// array once in and once out; the tables (x: N, w: M, twiddles) stay L2-resident.
template <typename T, int L1, int L2>
__global__ void __launch_bounds__(FOURIER_TWOLEVEL_NT(T, L1, L2), FOURIER_MIN_WAVES(FOURIER_TWOLEVEL_NT(T, L1, L2)))
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
  cpx<T> x[VEC][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const Unit16<T> u = buf_load_unit<T>(ri, voff + (uint32_t)r * ROWB);
#pragma unroll
    for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
  }
  units_batched<T, 8>([&](int r) { return buf_load_unit<T>(rc, voff + (uint32_t)r * ROWB); },
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
    units_batched<T, 8>([&](int r) { return buf_load_unit<T>(rc, soff + (uint32_t)r * ROWB); },
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

struct TinyArgs {
  const void* in; void* out;
  uint64_t batch; int n; int swap_in, swap_out; double scale;
};

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
    if (u < total_units) v = load_unit_a8<T>(in + u * VEC);
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
    if (u < total_units) store_unit_a8<T>(out + u * VEC, v);
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

// ---- native Stockham autosort for small mixed-radix sizes N = 2^a * 3^b (b > 0), N <= 4096 ----
// This kernel is the reference's algorithm verbatim, one workgroup per group of transforms, all passes in
// LDS: radix schedule [4,8,4,3,2] (autosort/mod.rs:20-21,104-116), per-pass twiddle table
// [1, W^i, .., W^{(R-1)i}] (mod.rs:24-46), pass body out[j + R*s*i + s*k] = tw[i*R+k] * butterflyR(in[j + s*i + s*m*k'])_k
// (mod.rs:203-284), butterflies in the reference's operation order (autosort/butterfly.rs:3-65,
// vector/generic.rs:22-44) with FMA contraction off (Rust never fuses), then the scale pass (mod.rs:381-399).
// Same tables, same order, same roundings: results are bit-identical to the CPU restatement.
// The radix of the next pass of an n-point plan with `cur` points left to factor.  2^a*3^b: the reference's schedule,
// one radix 4 first when divisible, then greedily 8, 4, 3, 2 (autosort/mod.rs:20-21,104-116).  Lengths with a prime factor
// 5, 7, 11 or 13 are not the reference's to schedule (it sends them to Bluestein): odd radices first, largest first -- a
// stride-1 pass writes with a lane stride of R elements, which only an odd R spreads over all LDS banks -- then greedily
// 8, 4, 2, which never takes more passes than the reference's rule and one fewer when 2^a is a power of 8.
#ifndef FOURIER_MIX_PRIMES_FIRST
#define FOURIER_MIX_PRIMES_FIRST 1
#endif
constexpr bool mix_extended(uint32_t n) { return n % 5 == 0 || n % 7 == 0 || n % 11 == 0 || n % 13 == 0; }
constexpr uint32_t mix_next_radix(uint32_t n, uint32_t cur, bool first) {
  if (FOURIER_MIX_PRIMES_FIRST && mix_extended(n))
    return cur % 13 == 0 ? 13u : (cur % 11 == 0 ? 11u : (cur % 7 == 0 ? 7u : (cur % 5 == 0 ? 5u : (cur % 3 == 0 ? 3u :
           (cur % 8 == 0 ? 8u : (cur % 4 == 0 ? 4u : 2u))))));
  return (first && cur % 4 == 0) ? 4u : (cur % 8 == 0 ? 8u : (cur % 4 == 0 ? 4u : (cur % 3 == 0 ? 3u : (cur % 2 == 0 ? 2u :
         (cur % 5 == 0 ? 5u : (cur % 7 == 0 ? 7u : (cur % 11 == 0 ? 11u : 13u)))))));
}
struct MixArgs {
  const void* in; void* out; const void* tw;  // tw: forward table, Sum(size_cur) entries
  uint64_t batch;
  uint32_t n, group;      // transform length, transforms per workgroup
  uint32_t npass;         // passes, and the radix of each (mix_next_radix)
  uint8_t radix[20];
  int forward, scaled;
  double scale, w3re, w3im, w8re, w8im;  // compute_twiddle(1,3,true), compute_twiddle(1,8,true) as T values
};

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
#ifndef FOURIER_MIX_RT_WAVES
#define FOURIER_MIX_RT_WAVES(T, MAXP, PPT) ((sizeof(T) == 4 ? ((MAXP) <= 7 ? ((PPT) <= 4 ? 6 : ((PPT) <= 8 ? 5 : 4)) : 4) : ((MAXP) <= 7 ? ((PPT) <= 4 ? 4 : ((PPT) <= 8 ? 3 : 4)) : 2)))
#endif
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
  if constexpr (VEC == 1) {
    for (uint32_t idx = threadIdx.x; idx < total; idx += NT) buf[idx] = in[idx];
  } else {
    for (uint32_t u = threadIdx.x; u < units; u += NT) *(Unit16<T>*)(buf + u * VEC) = load_unit_a8<T>(in + u * VEC);
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
  if constexpr (VEC == 1) {
    for (uint32_t idx = threadIdx.x; idx < total; idx += NT) {
      cpx<T> y = buf[idx];
      if (a.scaled) y = {y.re * scale, y.im * scale};
      out[idx] = y;
    }
  } else {
    for (uint32_t u = threadIdx.x; u < units; u += NT) {
      Unit16<T> v = *(const Unit16<T>*)(buf + u * VEC);
      if (a.scaled) {
#pragma unroll
        for (uint32_t c = 0; c < 2 * VEC; ++c) v.a[c] = v.a[c] * scale;
      }
      store_unit_a8<T>(out + u * VEC, v);
    }
    if ((total % VEC) && threadIdx.x == 0) {
      cpx<T> y = buf[total - 1];
      if (a.scaled) y = {y.re * scale, y.im * scale};
      out[total - 1] = y;
    }
  }
}

// ---- final odd-radix Stockham pass for large N = 2^a * 3^b: R = 3^b in {3, 9, 27}, s = 2^a, m = 1 ----
// out[j + s*k] = DFT_R(in[j + s*k'])_k  (autosort/mod.rs:203-284 with size == R: no twiddle, :238).  The
// reference reaches radix 3 last as well (RADICES = [4,8,4,3,2], mod.rs:21).  One thread owns VEC adjacent
// columns j (one 16-byte unit per row) and all R rows: fully coalesced, in place allowed.
struct OddArgs {
  const void* in; void* out;
  uint64_t n, s, batch;       // transform length, Stockham stride of this pass, transforms
  uint64_t m;                 // size_cur / R: 1 for the last pass (stride = n / R), > 1 for a twiddled middle pass
  const void* tw;             // middle passes: W_size_cur^{e}, e < size_cur (size_cur = R * m)
  int swap_out;
  double scale;
  double wr[27], wi[27];      // W_R^e = exp(-2*pi*i*e/R), e < R (f64 on the host, cast on use)
};

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

// ---- the same kernel with the transform length fixed at compile time ----
// For the sizes the reference itself benchmarks (3^5, 3^6, 3^7, fft_bench.rs:153-159) and the common 3*2^k / 9*2^k
// lengths: schedule, sizes, strides and table offsets are constants, so the per-butterfly index arithmetic
// (two runtime divisions in mixed_pass) folds into multiply-shifts and the pass loop unrolls.  Same operations in
// the same order: still bit-identical to the CPU restatement.
// fused (3,3) pass pairs: for lengths with a factor 9; in f64 only from 1024 points on (below, the 18 extra VGPRs and
// the idle threads cost more than the saved LDS round trip: 729 f64 53 % without, 42 % with; 2187 30 % / 33 %)
#ifndef FOURIER_MIX_PAIR_MIN_N_F64
#define FOURIER_MIX_PAIR_MIN_N_F64 1024u
#endif
// fused (5,5) pairs (lengths beyond the reference's), 25 points per work item: built and measured, off -- too few work
// items per pass and 50+ live registers (f32 5000: 53 % of the HBM peak without, 33 % with; 10000: 48 / 33 %; f64 5000:
// 55 / 31 %; only 12500 / 15625 gain, 33 -> 35-36 %; r03_s22)
#ifndef FOURIER_MIX_PAIR5_MIN_N_F32
#define FOURIER_MIX_PAIR5_MIN_N_F32 0xffffffffu
#endif
#ifndef FOURIER_MIX_PAIR5_MIN_N_F64
#define FOURIER_MIX_PAIR5_MIN_N_F64 0xffffffffu
#endif
template <typename T> constexpr bool mix_pairs(uint32_t n, uint32_t r) {
  return r == 3 ? (n % 9 == 0 && (sizeof(T) == 4 || n >= FOURIER_MIX_PAIR_MIN_N_F64))
                : (r == 5 && n % 25 == 0 && n >= (sizeof(T) == 4 ? FOURIER_MIX_PAIR5_MIN_N_F32 : FOURIER_MIX_PAIR5_MIN_N_F64));
}
// transforms per workgroup: about 1024 points (16 KiB of LDS in f32).  More points per workgroup fill the 256
// threads better but lose more in resident workgroups than they gain (N=243 f32: 49 % at 1152 points, 40 % at
// 2304, 27 % at 4608; r01 session 13)
template <typename T> constexpr uint32_t mix_group(uint32_t n) { return 1024 / n ? 1024 / n : 1; }
// threads per workgroup: 256, and 1024 for one long transform per workgroup -- at 256 threads such a transform keeps 16+
// points per thread live across the in-place barrier and one 4-wave workgroup per CU cannot hide the LDS latency.  Same
// arithmetic, same bits.  2^a*3^b: above 4096 points (f32 9216: 29 -> 44 % of the HBM peak, 18432: 18 -> 34 %, f64 9216:
// 22 -> 35 %; below, f64 2187 loses 45 -> 34 %).  Lengths with factors 5..13: above 32 KiB per transform (f32 10000: 28 -> 48 %;
// f64 3125: 43 -> 57 %, 2500: 49 -> 57 %, but 2401: 45 -> 38 %; f32 from 16 KiB loses, 3125: 46 -> 28 %).  r03_s22.
#ifndef FOURIER_MIX_WIDE_MIN_BYTES
#define FOURIER_MIX_WIDE_MIN_BYTES 32768u
#endif
#ifndef FOURIER_MIX_WIDE_MIN_N
#define FOURIER_MIX_WIDE_MIN_N 4096u
#endif
// ... and 128 threads in f32 where a pass has, on average, no more than FOURIER_MIX_HALF_MAX_ITEMS work items (butterflies or
// butterfly pairs) per workgroup: these kernels are latency-bound chains of barrier-separated passes, most of 256 threads
// would idle, and half-size workgroups put twice as many chains on a CU (243: 50 -> 61 % of the HBM peak, 625: 43 -> 59 %,
// 729: 40 -> 54 %, 768: 46 -> 60 %; lengths with 200+ items per pass lose 3-12 points, every f64 length loses; 64 threads
// never beat 128; r03_s24_mixed_radix_threads_per_workgroup_ab.jsonl)
// (512 threads for the transforms between 16 KiB and the 1024-thread threshold: measured, no -- 2187 f32 56 -> 46 %, f64
// 1152 / 2000 53 / 55 -> 45 / 46 %, 4000 f32 46 -> 51 % the only gain; r03_s26_mixed_radix_mid_sizes_512_threads_ab.jsonl)
#ifndef FOURIER_MIX_MID_THREADS
#define FOURIER_MIX_MID_THREADS 256u
#define FOURIER_MIX_MID_MIN_BYTES 16384u
#endif
#ifndef FOURIER_MIX_HALF_MAX_ITEMS
#define FOURIER_MIX_HALF_MAX_ITEMS 190u
#endif
template <typename T> constexpr uint32_t mix_mean_items(uint32_t n) {
  uint32_t cur = n, passes = 0, items = 0;
  const uint32_t pts_total = mix_group<T>(n) * n;
  while (cur > 1) {
    const uint32_t r = mix_next_radix(n, cur, cur == n);
    if (cur % r) return 0xffffffffu;  // not a length these kernels factor
    const bool pair = (r == 3 || r == 5) && cur >= r * r && (cur / r) % r == 0 && mix_pairs<T>(n, r);
    const uint32_t pts = pair ? r * r : r;
    items += pts_total / pts;
    passes += 1;
    cur /= pts;
  }
  return passes ? items / passes : 0xffffffffu;
}
template <typename T> constexpr uint32_t mix_threads(uint32_t n) {
  return (mix_extended(n) ? n * 2u * (uint32_t)sizeof(T) > FOURIER_MIX_WIDE_MIN_BYTES : n > FOURIER_MIX_WIDE_MIN_N) ? 1024u
         : ((sizeof(T) == 4 && mix_mean_items<T>(n) <= FOURIER_MIX_HALF_MAX_ITEMS) ? 128u
         : (n * 2u * (uint32_t)sizeof(T) > FOURIER_MIX_MID_MIN_BYTES ? FOURIER_MIX_MID_THREADS : 256u));
}
// Every pass runs IN PLACE on one LDS buffer: a thread keeps the outputs of all its butterflies of a pass in
// registers across a barrier, then writes them back to the buffer it read from.  Same arithmetic as the ping-pong
// form; half the LDS, so twice the resident workgroups where LDS was the limit (N=6561 f32 16 -> 32 % of the HBM
// peak, 2304 37 -> 51 %, f64 1152 46 -> 61 %) and no loss elsewhere (A/B over the threshold,
// profiles/r01_s15_mixed_inplace_ab.txt).  FOURIER_MIX_INPLACE_BYTES > 0 restores ping-pong below that footprint.
#ifndef FOURIER_MIX_INPLACE_BYTES
#define FOURIER_MIX_INPLACE_BYTES 0u
#endif
template <typename T> constexpr bool mix_inplace(uint32_t n) {
  return FOURIER_MIX_INPLACE_BYTES == 0u || 2u * mix_group<T>(n) * n * 2u * sizeof(T) > FOURIER_MIX_INPLACE_BYTES;
}

template <typename T, uint32_t N, uint32_t SIZE, uint32_t STRIDE, uint32_t TWOFF, bool FIRST_PASS> struct MixPassesCT {
  static constexpr uint32_t R = mix_next_radix(N, SIZE, FIRST_PASS), M = SIZE / R, NT = mix_threads<T>(N);
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

  // work item q: load, butterfly (+ twiddle), results in y[PTS] in the order of the output slots
  static __device__ __forceinline__ void compute(const cpx<T>* src, const cpx<T>* tw, uint32_t q, bool fwd, cpx<T> w3, cpx<T> w8,
                                                 cpx<T> (&y)[PTS], uint32_t& out_off) {
    const uint32_t g = q / NBF, e = q % NBF, i = e / STRIDE, j = e % STRIDE;  // constants: multiply-shift
    const cpx<T>* in = src + g * N + j + STRIDE * i;
    const cpx<T>* __restrict__ t = tw + TWOFF;
    out_off = g * N + j + PTS * STRIDE * i;
    if constexpr (PAIR) {
      const cpx<T>* __restrict__ t2 = tw + TWOFF + SIZE;
      cpx<T> x[R][R];
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2)
#pragma unroll
        for (uint32_t k1 = 0; k1 < R; ++k1) {
          LDS_NOTE(in + STRIDE * (M2 * k2 + M * k1), sizeof(cpx<T>), false, 100);
          x[k2][k1] = in[STRIDE * (M2 * k2 + M * k1)];
        }
#pragma unroll
      for (uint32_t k2 = 0; k2 < R; ++k2) {
        ref_butterfly<T, (int)R>(x[k2], fwd, w3, w8);
#pragma unroll
        for (uint32_t k = 1; k < R; ++k) {
          cpx<T> w = t[(i + M2 * k2) * R + k];
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
            cpx<T> w = t2[i * R + k2];
            if (!fwd) w.im = -w.im;
            z[k2] = ref_mul(z[k2], w);
          }
        }
#pragma unroll
        for (uint32_t k2 = 0; k2 < R; ++k2) y[k + R * k2] = z[k2];  // output slot STRIDE * (k + R*k2)
      }
    } else {
#pragma unroll
      for (uint32_t k = 0; k < R; ++k) {
        LDS_NOTE(in + STRIDE * M * k, sizeof(cpx<T>), false, 101);
        y[k] = in[STRIDE * M * k];
      }
      ref_butterfly<T, (int)R>(y, fwd, w3, w8);
      if constexpr (SIZE != R) {  // mod.rs:238,272
#pragma unroll
        for (uint32_t k = 1; k < R; ++k) {
          cpx<T> w = t[i * R + k];
          if (!fwd) w.im = -w.im;
          y[k] = ref_mul(y[k], w);
        }
      }
    }
  }

  static __device__ __forceinline__ const cpx<T>* run(const cpx<T>* src, cpx<T>* dst, const cpx<T>* tw, uint32_t nb, bool fwd,
                                                     cpx<T> w3, cpx<T> w8) {
    if constexpr (mix_inplace<T>(N)) {
      constexpr uint32_t ROUNDS = (mix_group<T>(N) * NBF + NT - 1) / NT;
      cpx<T> y[ROUNDS][PTS];
      uint32_t off[ROUNDS];
#pragma unroll
      for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
        const uint32_t q = threadIdx.x + NT * rd;
        if (q < nb * NBF) compute(src, tw, q, fwd, w3, w8, y[rd], off[rd]);
      }
      __syncthreads();  // every input of the pass has been read
      cpx<T>* buf = const_cast<cpx<T>*>(src);
#pragma unroll
      for (uint32_t rd = 0; rd < ROUNDS; ++rd) {
        const uint32_t q = threadIdx.x + NT * rd;
        if (q < nb * NBF) {
#pragma unroll
          for (uint32_t k = 0; k < PTS; ++k) {
            LDS_NOTE(buf + off[rd] + STRIDE * k, sizeof(cpx<T>), true, 102);
            buf[off[rd] + STRIDE * k] = y[rd][k];
          }
        }
      }
      __syncthreads();
      if constexpr (LAST) return src;
      else return MixPassesCT<T, N, OUT_SIZE, OUT_STRIDE, OUT_TWOFF, false>::run(src, dst, tw, nb, fwd, w3, w8);
    } else {
      for (uint32_t q = threadIdx.x; q < nb * NBF; q += NT) {
        cpx<T> y[PTS];
        uint32_t off;
        compute(src, tw, q, fwd, w3, w8, y, off);
#pragma unroll
        for (uint32_t k = 0; k < PTS; ++k) dst[off + STRIDE * k] = y[k];
      }
      __syncthreads();
      if constexpr (LAST) return dst;
      else return MixPassesCT<T, N, OUT_SIZE, OUT_STRIDE, OUT_TWOFF, false>::run(dst, const_cast<cpx<T>*>(src), tw, nb, fwd, w3, w8);
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
  // odd-length f32 batch are only 8-byte aligned, which global_load/store_dwordx4 tolerate), one odd point by itself
  constexpr uint32_t VEC = 16 / (2 * (uint32_t)sizeof(T));
  const uint32_t units = total / VEC;
  if constexpr (VEC == 1) {
    for (uint32_t idx = threadIdx.x; idx < total; idx += NT) buf0[idx] = in[idx];
  } else {
    for (uint32_t u = threadIdx.x; u < units; u += NT) *(Unit16<T>*)(buf0 + u * VEC) = load_unit_a8<T>(in + u * VEC);
    if ((total % VEC) && threadIdx.x == 0) buf0[total - 1] = in[total - 1];
  }
  __syncthreads();
  const bool fwd = a.forward != 0;
  cpx<T> w3{(T)a.w3re, (T)a.w3im}, w8{(T)a.w8re, (T)a.w8im};
  if (!fwd) { w3.im = -w3.im; w8.im = -w8.im; }
  const cpx<T>* res = MixPassesCT<T, N, N, 1, 0, true>::run(buf0, buf1, (const cpx<T>*)a.tw, nb, fwd, w3, w8);
  const T scale = a.scaled ? (T)a.scale : (T)1;  // mod.rs:387-393 (the unscaled codes skip the multiply: x * 1 is exact)
  if constexpr (VEC == 1) {
    for (uint32_t idx = threadIdx.x; idx < total; idx += NT) {
      cpx<T> y = res[idx];
      if (a.scaled) y = {y.re * scale, y.im * scale};
      out[idx] = y;
    }
  } else {
    for (uint32_t u = threadIdx.x; u < units; u += NT) {
      Unit16<T> v = *(const Unit16<T>*)(res + u * VEC);
      if (a.scaled) {
#pragma unroll
        for (uint32_t c = 0; c < 2 * VEC; ++c) v.a[c] = v.a[c] * scale;
      }
      store_unit_a8<T>(out + u * VEC, v);
    }
    if ((total % VEC) && threadIdx.x == 0) {
      cpx<T> y = res[total - 1];
      if (a.scaled) y = {y.re * scale, y.im * scale};
      out[total - 1] = y;
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
    if (last) store_unit<T, FOURIER_NT_STORE != 0>(out + (uint64_t)k * a.s, v);
    else store_unit<T, false>(out + (uint64_t)k * a.s, v);
  }
}

// ---- one Stockham pass in global memory, any radix R in {2,3,4,8,9,16,27}, any stride: the 2^a*3^b lengths with a < 12
// that do not fit the LDS kernels (3^10, 2^8*3^5, ...).  The reference's pass verbatim (autosort/mod.rs:203-284):
//   out[j + R*s*i + s*k] = W_size^{i*k} * DFT_R(in[j + s*i + s*m*k'])_k,   i < m, j < s, size = R*m,
// one thread per butterfly e = j + s*i: for a fixed k' the reads in[e + s*m*k'] are contiguous over the threads whatever
// the stride; the writes are contiguous in runs of s.  Passes are scheduled odd radices first (27, 9, 3), then 16, 8, 4, 2
// (GenericEngine); every pass is one HBM round trip.
struct GenArgs {
  const void* in; void* out;
  const void* tw;             // W_size^{e}, e < size (null when m == 1: the last pass has no twiddle, mod.rs:238)
  uint64_t n;                 // transform length (batch stride)
  uint32_t s, m;              // stride, butterflies per stride group; s * m = n / R
  uint32_t blocks_per;        // workgroups per transform
  int swap_in, swap_out, final_pass;
  double scale;
  double wr[27], wi[27];      // W_R^e for the radix-3^b butterflies
};
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
  const cpx<T>* tw = (const cpx<T>*)a.tw;
  const T scale = (T)a.scale;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (tw && k > 0 && valid) x[k] = cmul(x[k], tw[(uint64_t)i * k]);
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
struct BluArgs {
  const void* in; void* out; const void* xtab;
  uint64_t n, m, batch; int swap; double scale;
};
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
