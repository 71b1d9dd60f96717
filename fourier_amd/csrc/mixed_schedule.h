// mixed_schedule.h -- the schedule of the LDS mixed-radix kernels as compile-time functions of the transform length: radix
// of each pass (autosort/mod.rs:20-21,104-116 for 2^a*3^b; continued for the factors 5..13), fused pass pairs, transforms
// and threads per workgroup.  Shared by the kernels (kernels_mixed.h) and by the host, which builds the twiddle tables and
// the launch shape from the same rules (engine_mixed.h).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace fourier_hip {

// The radix of the next pass of an n-point plan with `cur` points left to factor.  2^a*3^b: the reference's schedule,
// one radix 4 first when divisible, then greedily 8, 4, 3, 2 (autosort/mod.rs:20-21,104-116).  Lengths with a prime factor
// 5, 7, 11 or 13 are not the reference's to schedule (it sends them to Bluestein): odd radices first, largest first -- a
// stride-1 pass writes with a lane stride of R elements, which only an odd R spreads over all LDS banks -- then greedily
// 8, 4, 2, which never takes more passes than the reference's rule and one fewer when 2^a is a power of 8.
#define FOURIER_MIX_PRIMES_FIRST 1
constexpr bool mix_extended(uint32_t n) { return n % 5 == 0 || n % 7 == 0 || n % 11 == 0 || n % 13 == 0; }
constexpr uint32_t mix_next_radix(uint32_t n, uint32_t cur, bool first) {
  if (FOURIER_MIX_PRIMES_FIRST && mix_extended(n))
    return cur % 13 == 0 ? 13u : (cur % 11 == 0 ? 11u : (cur % 7 == 0 ? 7u : (cur % 5 == 0 ? 5u : (cur % 3 == 0 ? 3u :
           (cur % 8 == 0 ? 8u : (cur % 4 == 0 ? 4u : 2u))))));
  return (first && cur % 4 == 0) ? 4u : (cur % 8 == 0 ? 8u : (cur % 4 == 0 ? 4u : (cur % 3 == 0 ? 3u : (cur % 2 == 0 ? 2u :
         (cur % 5 == 0 ? 5u : (cur % 7 == 0 ? 7u : (cur % 11 == 0 ? 11u : 13u)))))));
}

// fused (3,3) pass pairs: for lengths with a factor 9; in f64 only from 1024 points on (below, the 18 extra VGPRs and
// the idle threads cost more than the saved LDS round trip: 729 f64 53 % without, 42 % with; 2187 30 % / 33 %)
#define FOURIER_MIX_PAIR_MIN_N_F64 1024u
// fused (5,5) pairs (lengths beyond the reference's), 25 points per work item: built and measured, off -- too few work
// items per pass and 50+ live registers (f32 5000: 53 % of the HBM peak without, 33 % with; 10000: 48 / 33 %; f64 5000:
// 55 / 31 %; only 12500 / 15625 gain, 33 -> 35-36 %; r03_s22)
#define FOURIER_MIX_PAIR5_MIN_N_F32 0xffffffffu
#define FOURIER_MIX_PAIR5_MIN_N_F64 0xffffffffu
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
#define FOURIER_MIX_WIDE_MIN_BYTES 32768u
#define FOURIER_MIX_WIDE_MIN_N 4096u
// ... and 128 threads in f32 where a pass has, on average, no more than FOURIER_MIX_HALF_MAX_ITEMS work items (butterflies or
// butterfly pairs) per workgroup: these kernels are latency-bound chains of barrier-separated passes, most of 256 threads
// would idle, and half-size workgroups put twice as many chains on a CU (243: 50 -> 61 % of the HBM peak, 625: 43 -> 59 %,
// 729: 40 -> 54 %, 768: 46 -> 60 %; lengths with 200+ items per pass lose 3-12 points, every f64 length loses; 64 threads
// never beat 128; r03_s24_mixed_radix_threads_per_workgroup_ab.jsonl)
// (512 threads for the transforms between 16 KiB and the 1024-thread threshold: measured, no -- 2187 f32 56 -> 46 %, f64
// 1152 / 2000 53 / 55 -> 45 / 46 %, 4000 f32 46 -> 51 % the only gain; r03_s26_mixed_radix_mid_sizes_512_threads_ab.jsonl)
#define FOURIER_MIX_MID_THREADS 256u
#define FOURIER_MIX_MID_MIN_BYTES 16384u
#define FOURIER_MIX_HALF_MAX_ITEMS 190u
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
#define FOURIER_MIX_INPLACE_BYTES 0u
// First pass straight from global memory, last pass straight to global memory (mix_gio): the per-length kernels otherwise copy the
// transforms into LDS, run every pass there and copy the result out -- 2 * passes + 2 LDS accesses per point and 2 * passes + 1
// barriers.  A work item of the first pass reads its points in[i + m * k] itself (for a fixed k the lanes of a wave -- consecutive
// butterflies i -- read consecutive elements), a work item of the last pass writes out[j + stride * k] itself (consecutive j):
// 2 * passes - 2 LDS accesses per point, 2 * passes - 3 barriers; same butterflies, tables and order, so the same bits.  Only
// where a wave's accesses stay contiguous: at least FOURIER_MIX_GIO_MIN_RUN consecutive butterflies in the first and in the last
// pass (short transforms pack several per workgroup and would read 3 of every 12 elements per instruction) -- and only where it
// was measured faster (profiles/r04_s13_mixed_radix_first_last_pass_global_io_ab.jsonl, bit-identical arms on shared buffers):
// 2^a*3^b from 10 KiB per transform on -- f32 1536 +7 %, 3072 +14 %, 9216 +8 %, 18432 +16 %, 19683 +10 %, 6561 +6 %; f64 729 +11 %,
// 2187 +27 %, 4374 +15 %, 9216 +22 % -- while 768 / 729 f32 lose 2-3 %; of the lengths with factors 5..13 only 5^5 and 5^6 gain
// (+11 % / +8 %; 1000 -6 %, 2401 -4 %, 5000 / 10000 +-1 %).
#define FOURIER_MIX_GIO_MIN_RUN 64u
#define FOURIER_MIX_GIO_MIN_BYTES 10240u
#define FOURIER_MIX_GIO_MIN_BYTES_F32 98304u
#define FOURIER_MIX_GIO_ALL 0  // 1: every length that satisfies the run-length condition (A/B)
template <typename T> constexpr bool mix_gio(uint32_t n) {
  if (FOURIER_MIX_GIO_MIN_RUN == 0u) return false;
  uint32_t cur = n, first_run = 0, last_run = 0, stride = 1;
  while (cur > 1) {
    const uint32_t r = mix_next_radix(n, cur, cur == n);
    if (cur % r) return false;
    const bool pair = (r == 3 || r == 5) && cur >= r * r && (cur / r) % r == 0 && mix_pairs<T>(n, r);
    const uint32_t pts = pair ? r * r : r;
    if (cur == n) first_run = n / pts;  // butterflies i = 0 .. n / pts - 1 at stride 1
    last_run = stride;                  // the last pass has m = 1: outputs j = 0 .. stride - 1 per k
    stride *= pts;
    cur /= pts;
  }
  // (re-measured once the copies into and out of LDS were batched, profiles/r04_s48_gio_rule_after_batched_copies_ab.jsonl: in f64 the
  // direct form still wins from 10 KiB on -- 729 +9 %, 3072 +12 %, 9216 +18 % --, in f32 only for the transforms that leave a CU no
  // second workgroup -- 13824 +11 %, 18432 +8 %, 15625 +4 % -- while 1536 ... 9216, 3125 and 19683 are 3 - 7 % faster through the copy)
  const uint32_t bytes = n * 2u * (uint32_t)sizeof(T);
  const bool measured_faster = mix_extended(n) ? n == 15625u
                               : (sizeof(T) == 8 ? bytes >= FOURIER_MIX_GIO_MIN_BYTES : (bytes >= FOURIER_MIX_GIO_MIN_BYTES_F32 && n != 19683u));
  return (measured_faster || FOURIER_MIX_GIO_ALL != 0) && first_run >= FOURIER_MIX_GIO_MIN_RUN && last_run >= FOURIER_MIX_GIO_MIN_RUN &&
         2u * mix_group<T>(n) * n * 2u * sizeof(T) > FOURIER_MIX_INPLACE_BYTES;
}
// LDS layout of the data between two passes (FOURIER_MIX_SWIZZLE).  A pass writes out[j + PTS*stride*i + stride*k]: with a
// power-of-two stride below 16 the sixteen lanes of a ds_write lane group -- consecutive (i, j) -- land on 4 (first pass, radix
// 4 at stride 1: element 4 i + k) or 4 (second pass, radix 8 at stride 4: j + 32 i + 4 k) of the 16 eight-byte bank pairs, a
// 4-way conflict that makes these two passes half of the kernel's LDS cycles (SQ_LDS_BANK_CONFLICT = 36 - 48 % of
// SQ_LDS_IDX_ACTIVE on 768 ... 18432 points, profiles/r04_s14b_sq_breakdown_mixed.json; the later passes, stride >= 32, write
// contiguously).  Such a pass therefore writes element e at e ^ (field << dst), field = the nbits address bits from bit `src` up
// that tell the lanes of a group apart but lie above the bank bits, moved onto the bank bits that are the same for all of
// them; the next pass reads through the same map.  The map permutes the elements of an aligned block of 16 (f64: 8), so contiguous
// accesses (every read, every later write) stay conflict-free and no LDS is added.  Encoded src << 8 | nbits << 4 | dst; 0 = the
// plain layout.  Only where the transform is a whole number of such blocks.
#define FOURIER_MIX_SWIZZLE 1
constexpr bool mix_is_pow2(uint32_t v) { return v != 0 && (v & (v - 1)) == 0; }
constexpr uint32_t mix_log2(uint32_t v) { return v <= 1 ? 0 : 1 + mix_log2(v >> 1); }
// group_bits: log2 of the lanes in a ds_write lane group = log2 of the bank slots an element can fall on -- 4 for 8-byte elements
// (ds_write_b64: 16 lanes, 16 bank pairs), 3 for 16-byte elements (ds_write_b128: 8 lanes, 8 slots of four banks)
constexpr uint32_t mix_out_layout(uint32_t n, uint32_t stride, uint32_t pts, bool last, uint32_t group_bits) {
  const uint32_t lanes = 1u << group_bits;
  if (!FOURIER_MIX_SWIZZLE || last || n % lanes != 0 || stride >= lanes || !mix_is_pow2(stride) || !mix_is_pow2(pts) || pts < 2) return 0;
  const uint32_t s = mix_log2(stride), b = s + mix_log2(pts);  // j: bits [0, s); k: bits [s, b); i: bits from b up
  // the group's group_bits - s bits of i: those below bit `group_bits` already select slots; the others move onto k's bits
  const uint32_t src = b >= group_bits ? b : group_bits, nbits = b >= group_bits ? group_bits - s : b - s, dst = s;
  return nbits == 0 ? 0 : (src << 8 | nbits << 4 | dst);
}
constexpr uint32_t mix_sw(uint32_t layout, uint32_t e) {
  return layout == 0 ? e : e ^ (((e >> (layout >> 8)) & ((1u << ((layout >> 4) & 15u)) - 1u)) << (layout & 15u));
}
// the bits of an element index the map looks at or changes: an offset that is a multiple of this leaves the map's XOR term alone
constexpr uint32_t mix_sw_span(uint32_t layout) { return layout == 0 ? 1u : 1u << ((layout >> 8) + ((layout >> 4) & 15u)); }
// In-place passes: the loads of ALL of a thread's work items ahead of its first butterfly (kernels_mixed.h: MixPassesCT::run).  A work
// item is a basic block of its own (q < nb * NBF) and hipcc otherwise waits for one item's loads -- global memory in the first pass of
// a mix_gio kernel -- before it issues the next item's.  Where it was measured faster (bit-identical arms,
// profiles/r04_s20_mixed_radix_loads_first_ab.jsonl): 2^a*3^b from 16 KiB per transform on -- f64 1536 +7 %, 3072 +9 %, 4608 +5 %,
// 9216 +8 %; f32 3072 +3 %, 6144 +5 %, 2304 / 13824 +2 % -- but not the f32 lengths that fill a CU's LDS (18432 -2 %); shorter
// transforms lose 1 - 2 %; of the lengths with factors 5..13 only 5^5 gains (+3 %; 5000 -3 %, 10000 f64 -2 %).  The pass's twiddles
// loaded with them as well: measured, no (r04_s21: 9216 f32 / f64 -22 % / -8 %, the others +-2 %).
#define FOURIER_MIX_LOADS_FIRST 1  // 0: never, 2: every length (A/B)
template <typename T> constexpr bool mix_loads_first(uint32_t n) {
  if (FOURIER_MIX_LOADS_FIRST != 1) return FOURIER_MIX_LOADS_FIRST != 0;
  if (mix_extended(n)) return false;  // (5^5 f32 gained 3 % while its first pass read global memory; through the batched copy: -1 %, r04_s49)
  return n * 2u * (uint32_t)sizeof(T) >= 16384u && (sizeof(T) == 8 || n <= 16384u);
}
template <typename T> constexpr bool mix_inplace(uint32_t n) {
  return FOURIER_MIX_INPLACE_BYTES == 0u || 2u * mix_group<T>(n) * n * 2u * sizeof(T) > FOURIER_MIX_INPLACE_BYTES;
}


// Launch shape of a column-tile pass of length L (kernels_tiled.h): ONE definition for the kernel (TiledCfg) and for the host,
// which launches a tile kernel compiled at run time from these numbers (engine_tiled.h; ADVICE round 4: the two copies of the rule
// could drift apart -- a block size or LDS size that differs from the kernel's compile-time one is silent corruption).
// elem = sizeof(complex<T>): 8 / 16.
struct TileShape {
  uint32_t cols;     // 128-byte row segments: 16 (f32) / 8 (f64) columns
  uint32_t ld;       // leading dimension of a column in LDS: odd, so that the lanes of a row segment fall on different banks
  uint32_t threads;  // about eight points per thread
  uint32_t kh;       // inter-pass twiddle of a tile: W^{i*k} = TA[col][k / 16] * TB[col][k % 16], kh = entries of TA per column
  uint32_t tab_off;  // byte offset of TA behind the tile
  uint32_t smem;     // bytes of LDS
};
constexpr TileShape tiled_shape(uint32_t L, uint32_t elem) {
  TileShape t{};
  t.cols = 128u / elem;
  t.ld = L | 1u;
  const uint32_t per8 = L * t.cols / 8u;
  t.threads = per8 <= 256u ? 256u : (per8 <= 512u ? 512u : 1024u);
  t.kh = (L + 15u) / 16u;
  t.tab_off = (t.cols * t.ld * elem + 15u) & ~15u;
  t.smem = t.tab_off + t.cols * (t.kh + 16u) * elem;
  return t;
}

// Launch shape of a register-resident column-tile pass of length L = R1 x R2 (kernels_regtile.h): ONE definition for the kernel
// (RegTileCfg) and for the host.  r1 == 0: the length has no split into two factors of at most 32 (7^3, 5 * 7^2, 10 * 7^2, ...): it
// stays on kernels_tiled.h.  The most balanced split, R1 >= R2: stage B (COLS x R1 threads, R2 stores each) keeps every thread busy.
struct RegTileShape {
  uint32_t r1, r2;
  uint32_t cols;     // 128-byte row segments: 16 (f32) / 8 (f64) columns
  uint32_t threads;  // (COLS / columns per 16-byte unit) x R1, rounded up to whole waves
  uint32_t xstride;  // elements between the k1 planes of the exchange buffer: R2 * COLS
  uint32_t ldo;      // leading dimension of a column in the first pass's output staging: odd
  uint32_t tab_off;  // byte offset of the inter-pass twiddle tables behind the buffer
  uint32_t smem;     // bytes of LDS
};
constexpr uint32_t REG_TILE_MAX_FACTOR = 32;
// row of (plane k1, j2) inside its plane of the exchange buffer
constexpr uint32_t reg_tile_row(uint32_t r2, uint32_t k1, uint32_t j2) { return (r2 % 2u == 0u) ? (j2 ^ (k1 & 1u)) : j2; }
constexpr RegTileShape reg_tile_shape(uint32_t L, uint32_t elem) {
  RegTileShape t{};
  uint32_t r2 = 0;
  // (f32 beyond 512 points: stages of up to 40 points -- 1000 = 40 x 25, so that 10^6 is two passes: 0.20 -> 0.29 of the HBM peak; f64 spills
  // into AGPRs at 40 points and stays on three passes: 0.19 against 0.21, profiles/r06_s42_1000_point_tiles*.jsonl)
  const uint32_t max_factor = (elem == 8u && L > 512u) ? 40u : REG_TILE_MAX_FACTOR;
  for (uint32_t b = 2; b * b <= L; ++b)
    if (L % b == 0 && L / b <= max_factor) r2 = b;
  // (125 = 25 x 5: 40 of 256 threads transform in stage A -- f64 15625 = 125 x 125 0.26 against 0.31 on the LDS kernel, r06_s25)
  if (r2 == 0 || L / r2 > 4u * r2) return t;
  t.r1 = L / r2; t.r2 = r2;
  // 128-byte row segments (256-byte ones: -26 ... +5 %, profiles/r06_s38_regtile_wide_ab.jsonl); 64-byte ones beyond 512 points: the tile stays
  // within 64 KiB of LDS, two workgroups per CU
  t.cols = (L > 512u ? 64u : 128u) / elem;
  t.threads = (t.cols / (16u / elem) * t.r1 + 63u) & ~63u;
  // stage B reads plane k1 = tid / COLS at j2 * COLS + c: the two (f64: four) planes a lane group of a ds_read touches must differ by an
  // odd number of 128-byte row segments -- R2 odd, or (R2 even) rows j2 and j2 ^ 1 exchanged in the odd planes (reg_tile_row); no padding
  t.xstride = t.r2 * t.cols;
  t.ldo = L | 1u;
  const uint32_t xch = t.r1 * t.xstride, stage = t.cols * t.ldo;
  t.tab_off = ((xch > stage ? xch : stage) * elem + 15u) & ~15u;
  t.smem = t.tab_off + t.cols * (t.r1 + t.r2) * elem;
  return t;
}

}  // namespace fourier_hip
