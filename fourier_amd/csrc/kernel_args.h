// kernel_args.h -- argument blocks of the kernels (passed by value) and the enums that name their modes: what the host
// translation unit needs to know about a kernel family without seeing its device code.
#pragma once
#ifdef __HIPCC_RTC__  // hipRTC (rtc.cpp): no system headers; its built-in runtime header is already in
typedef unsigned char uint8_t; typedef unsigned int uint32_t; typedef unsigned long long uint64_t; typedef int int32_t;
#else
#include <stdint.h>
#endif

namespace fourier_hip {

template <typename T> struct cpx { T re, im; };

enum { MODE_FIRST = 0, MODE_MID = 1, MODE_LAST = 2, MODE_ROWS = 3 };
// Bluestein fusion (bluesteins.rs:229-258): IO_BLU_IN = the first pass of the forward inner FFT reads the
// USER array (length blu_n, zero padded to n) times the chirp x; IO_BLU_OUT = the last pass of the inverse
// inner FFT writes the first blu_n points times the chirp (and the user scaling) into the USER array.
enum { IO_PLAIN = 0, IO_BLU_IN = 1, IO_BLU_OUT = 2 };


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
  uint32_t walk_band, walk_group, walk_tf;  // mode 4: tiles per band, transforms per group (0 = the XCD's whole range), transform-fastest
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


// ---- lane-per-transform kernels (kernels_small.h)
struct TinyArgs {
  const void* in; void* out;
  uint64_t batch; int n; int swap_in, swap_out; double scale;
};


// ---- LDS mixed-radix kernels (kernels_mixed.h)
struct MixArgs {
  const void* in; void* out; const void* tw;  // tw: forward table, Sum(size_cur) entries
  uint64_t batch;
  uint32_t n, group;      // transform length, transforms per workgroup
  uint32_t npass;         // passes, and the radix of each (mix_next_radix)
  uint8_t radix[20];
  int forward, scaled;
  double scale, w3re, w3im, w8re, w8im;  // compute_twiddle(1,3,true), compute_twiddle(1,8,true) as T values
};


// ---- odd-radix passes, global-memory passes, unfused Bluestein sweeps (kernels_misc.h)
struct OddArgs {
  const void* in; void* out;
  uint64_t n, s, batch;       // transform length, Stockham stride of this pass, transforms
  uint64_t m;                 // size_cur / R: 1 for the last pass (stride = n / R), > 1 for a twiddled middle pass
  const void* tw;             // middle passes: W_size_cur^{e}, e < size_cur (size_cur = R * m)
  int swap_out;
  double scale;
  double wr[27], wi[27];      // W_R^e = exp(-2*pi*i*e/R), e < R (f64 on the host, cast on use)
};


struct GenArgs {
  const void* in; void* out;
  const void* tw_lo;          // W_size^{e}, e < 2^lo_bits          } two-level table of the pass twiddle W_size^{i*k}, as in the
  const void* tw_hi;          // W_size^{h << lo_bits}               } tile passes (null when m == 1: the last pass has no twiddle,
  uint32_t lo_bits;           //                                       mod.rs:238)
  uint64_t n;                 // transform length (batch stride)
  uint32_t s, m;              // stride, butterflies per stride group; s * m = n / R
  uint32_t blocks_per;        // workgroups per transform
  int swap_in, swap_out, final_pass;
  double scale;
  double wr[27], wi[27];      // W_R^e for the radix-3^b butterflies
};

struct BluArgs {
  const void* in; void* out; const void* xtab;
  uint64_t n, m, batch; int swap; double scale;
};

// ---- big-radix passes of mixed length on column tiles (kernels_tiled.h)
struct TiledArgs {
  const void* in; void* out;
  const void* tw;             // tables of the in-tile length-L transform, the reference's per-pass layout (mod.rs:24-46)
  const void* tw_lo;          // two-level table of the inter-pass twiddle W_size^{i*k} (null in the last pass)
  const void* tw_hi;
  uint32_t lo_bits;
  uint64_t n, s, m;           // transform length (batch stride); Stockham stride; size / L (1 in the last pass)
  uint64_t tiles_per_row;     // ceil(columns / COLS): columns = m in the first pass (s == 1), s afterwards
  int swap_in, swap_out;
  uint32_t xcd_chunk;         // workgroup -> tile order (xcd_chunked, kernels_common.h): 0 = identity
  double scale;               // applied by the last pass
  double w3re, w3im, w8re, w8im;  // compute_twiddle(1, 3, true), compute_twiddle(1, 8, true) as T values (butterfly.rs:12,50)
  // Bluestein on smooth M (kernels_regtile.h): chirp table x (blu_n entries), w = FFT_M(conj chirp) / M (n entries), user length,
  // user-level inverse (swap at the user array)
  const void* blu_x; const void* blu_w;
  uint64_t blu_n;
  int blu_swap;
};

// ---- whole chirp-z of a short transform in one launch on a smooth M = R1 x R2, both transforms in registers (kernels_chirpz.h)
struct ChirpzArgs {
  const void* in; void* out;
  const void* chirp;          // x[0 .. n): exp(-i pi k^2 / n) (bluesteins.rs:51-61)
  const void* w;              // FFT_M(conj chirp, mirrored) / M (bluesteins.rs:18-48), M entries
  const void* tw;             // [j2 < R2][k1 < R1]: W_M^{j2 * k1}
  uint64_t n, batch;          // user transform length (batch stride), transforms
  int swap;                   // user-level inverse: swap re / im of the user data
  double scale;
};

// ---- XCD-fused one-launch plan (kernels_experiments.h)
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
constexpr uint64_t fused_ctrl_stride(uint64_t batch) { return 16 + 3 * (batch + 2); }
constexpr uint64_t fused_ctrl_words(uint64_t batch) { return FUSED_CTRL_HDR + FUSED_XCC_IDS * fused_ctrl_stride(batch); }


}  // namespace fourier_hip
