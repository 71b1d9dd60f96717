/* fourier.h -- C ABI of the MI355X-native FFT engine (libfourier.so).
 *
 * Part 1 is byte-compatible with the reference's C header, calebzulawski/fourier
 * `fourier-ffi/include/fourier.h:30-58` (implementation `fourier-ffi/src/lib.rs:14-106`): same
 * symbol names, argument meaning, transform codes and error behaviour, so existing C/C++ users and
 * the reference's own `Fft` trait (fourier-algorithms/src/fft.rs:40-82, through the Rust shim shown
 * in INTEGRATION.md) relink against this library unchanged.  The 8 legacy entry points take HOST
 * buffers of exactly `size` elements (no length argument, as in the reference) and are synchronous.
 *
 * Part 2 is the new surface the reversed boundary needs: batched execution
 * on device-resident interleaved buffers, stream-ordered, plus status queries.  The reference has
 * no batch API (fft.rs:48-61 is one slice per call); batching is what the GPU path is measured on.
 *
 * Handles are Send, not Sync -- like the reference's plans (RefCell scratch,
 * fourier-algorithms/src/autosort/mod.rs:54): one thread / one stream at a time per handle.
 */
#ifndef FOURIER_H_
#define FOURIER_H_

#ifdef __cplusplus
#include <complex>
#include <cstddef>
#include <memory>
#define FOURIER_COMPLEX_FLOAT_TYPE ::std::complex<float>
#define FOURIER_COMPLEX_DOUBLE_TYPE ::std::complex<double>
#define FOURIER_SIZE_TYPE ::std::size_t
#define FOURIER_STRUCT
namespace fourier {
namespace c {
extern "C" {
#else
#include <stddef.h>
#define FOURIER_COMPLEX_FLOAT_TYPE float _Complex
#define FOURIER_COMPLEX_DOUBLE_TYPE double _Complex
#define FOURIER_SIZE_TYPE size_t
#define FOURIER_STRUCT struct
#endif

/* ---------------- Part 1: legacy ABI (replaces fourier-ffi/include/fourier.h:30-58) ---------- */

/* Transform codes: fourier.h:30-36, fourier-ffi/src/lib.rs:3-12, fourier-algorithms/src/fft.rs:4-16 */
enum {
  FOURIER_TRANSFORM_FFT = 0,              /* forward, unscaled                */
  FOURIER_TRANSFORM_IFFT = 1,             /* inverse, scaled by 1/N           */
  FOURIER_TRANSFORM_UNSCALED_IFFT = 2,    /* inverse, unscaled                */
  FOURIER_TRANSFORM_SQRT_SCALED_FFT = 3,  /* forward, scaled by 1/sqrt(N)     */
  FOURIER_TRANSFORM_SQRT_SCALED_IFFT = 4, /* inverse, scaled by 1/sqrt(N)     */
};

struct fourier_fft_float;
struct fourier_fft_double;

/* replaces fourier.h:41-42 / lib.rs:15-20,62-67.  NULL on failure (the reference returns NULL when
 * plan creation panics, lib.rs:18-19).  size == 0 returns NULL (the reference hangs). */
struct fourier_fft_float *fourier_create_float(FOURIER_SIZE_TYPE);
struct fourier_fft_double *fourier_create_double(FOURIER_SIZE_TYPE);

/* replaces fourier.h:44-45 / lib.rs:22-29,69-76.  NULL is a no-op. */
void fourier_destroy_float(FOURIER_STRUCT fourier_fft_float *);
void fourier_destroy_double(FOURIER_STRUCT fourier_fft_double *);

/* replaces fourier.h:47-51 / lib.rs:31-43,78-90.  Host buffer of `size` elements, in place.
 * Unknown transform code: silent no-op, buffer untouched (lib.rs:10). */
void fourier_transform_in_place_float(const FOURIER_STRUCT fourier_fft_float *,
                                      FOURIER_COMPLEX_FLOAT_TYPE *, int);
void fourier_transform_in_place_double(
    const FOURIER_STRUCT fourier_fft_double *, FOURIER_COMPLEX_DOUBLE_TYPE *,
    int);

/* replaces fourier.h:53-58 / lib.rs:45-59,92-106.  Host buffers, out of place (in == out allowed). */
void fourier_transform_float(const FOURIER_STRUCT fourier_fft_float *,
                             const FOURIER_COMPLEX_FLOAT_TYPE *,
                             FOURIER_COMPLEX_FLOAT_TYPE *, int);
void fourier_transform_double(const FOURIER_STRUCT fourier_fft_double *,
                              const FOURIER_COMPLEX_DOUBLE_TYPE *,
                              FOURIER_COMPLEX_DOUBLE_TYPE *, int);

/* ---------------- Part 2: device-resident batched extension (new surface) -------------------- */

/* Status codes returned by the fourier_hip_* calls and by fourier_hip_last_status_*. */
enum {
  FOURIER_HIP_OK = 0,
  FOURIER_HIP_INVALID_ARGUMENT = 1, /* NULL handle/pointer, unknown transform code, bad option   */
  FOURIER_HIP_OUT_OF_MEMORY = 2,    /* device allocation failed                                  */
  FOURIER_HIP_RUNTIME_ERROR = 3,    /* a HIP call or kernel launch failed                        */
  FOURIER_HIP_UNSUPPORTED = 4,      /* size outside the engine's range                           */
};

/* Create a plan on a specific device (-1 = current device).  Same plan factory as `create_fft_f32/f64`
 * (fourier/src/lib.rs:31-60): Stockham autosort for 2^a * 3^b as in the reference (`Autosort::new`, autosort/mod.rs:104-134),
 * Bluestein chirp-z (fourier-algorithms/src/bluesteins.rs) otherwise -- with one extension: SOME lengths that the reference
 * sends to Bluestein run as direct Stockham passes here (closer to the exact DFT than the chirp-z route, within the same
 * tolerance).  The routes, in the order they are tried (fourier_amd/csrc/plan.h, Plan::Plan), with the string
 * `fourier_hip_describe_*` returns for each (followed by " f32" / " f64"):
 *   powers of two                                   "stockham <L1>[x<L2>[x<L3>]]", "stockham <L1>x<L2> one-launch", "stockham tiny(<n>)":
 *                                                   big-radix Stockham passes, one, two or three HBM round trips
 *   2^a * 3^b, a >= 12, N = L1 x L2 with both tile  "stockham mixed tiles <L1>x<L2>": two column-tile passes of mixed length
 *     lengths <= 576 (12288 ... 331776)
 *   2^a * 3^b, a >= 12, every other length          "stockham <L1>x...x<27|9|3>": the power-of-two passes over 2^a, then radix-27 / 9 / 3 passes
 *   a length of 14 ... 20480 points with a factor   "stockham registers <R1>x<R2>[x<R3>] one-launch" (round 6): the whole transform in one launch on two or
 *     5 ... 13 that fourier_amd/csrc/regfft_shapes.h  three register-resident stages of at most 40 points -- 736 lengths in f32, 763 in f64 (to 20475 points:
 *     lists in the precision                          above 10240 f32 one transform per workgroup, f64 split planes), each one at least 1.04 x faster
 *                                                   than the route below it had (5005: f32 0.24 -> 0.49 of the HBM peak, f64 0.14 (Bluestein)
 *                                                   -> 0.50; 1001: 0.42 -> 0.60, 0.27 -> 0.69; f32 15625: 0.33 -> 0.42)
 *   2^a * 3^b * 5^c * 7^d * 11^e * 13^f that fit    "stockham mixed-radix <r1>.<r2>...." (+ " specialised" for a kernel compiled at run time):
 *     one compute unit's LDS (<= 20480 points in    every 2^a * 3^b; every such length with factors 5 (and the instantiated ones with a
 *     f32, 10240 in f64) AND have a kernel          factor 7) has a per-length kernel; any other length of this family up to 8192 points runs
 *                                                   the runtime-parameterised kernel (f64 with a factor 11 or 13: up to 2048 points)
 *   2^a * 3^b, a < 12, beyond the LDS limit, with   "stockham mixed tiles <L1>x<L2>[x<L3>]": two or three column-tile passes of mixed length
 *     N = L1 x L2 (x L3), every L in 64 ... 1024
 *   2^a * 3^b, a < 12, without such a factorisation "stockham global-pass <r1>.<r2>....": one Stockham pass per radix in global memory (since round 6,
 *                                                   tile lengths up to 1024, no accepted length is left without one: the fallback)
 *   2^a * 3^b * 5^c * 7^d with c + d >= 1 beyond    "stockham mixed tiles <L1>x<L2>[x<L3>]" (round 5; 10^5 = 400x250, 44100 = 210x210,
 *     the LDS kernels, N = L1 x L2 (x L3), every      10^6 = 1000x1000 in f32, 100x100x100 in f64; 390625 = 625x625, 500000 = 800x625): column-tile passes whose
 *     L in 64 ... 1024 (28 lengths above 512,         lengths have prime factors up to 7
 *     f32: 33)
 *   prime factors up to 13, no route above, and     "stockham mixed-radix ... specialised" / "stockham mixed tiles ... specialised": kernels
 *     its run-time kernels in the code-object cache   compiled by an earlier "specialise" (below) -- see fourier_hip_set_default_option
 *   every other length                              "bluestein M=<M> inner <power-of-two plan>[ fused]": chirp-z over a power-of-two transform
 *                                                   (bluesteins.rs:110), or -- round 6, where that work array is at least 1.6 x (f64: 1.44 x) longer and is
 *                                                   swept three times (M > 2^15, f64 2^14) -- "bluestein M=<L1*L2> inner mixed tiles
 *                                                   <L1>x<L2>": the same chirp-z over a product of two tile lengths >= 2N - 1
 *                                                   -- or, for a short length (2N - 1 <= 1024; f32 where that saves a tenth of the power of two),
 *                                                   "bluestein M=<R1*R2> registers <R1>x<R2> one-launch": the whole chirp-z in one launch
 *                                                   over M = R1 x R2 >= 2N - 1 with both M-point transforms in registers; up to 2N - 1 = 9261
 *                                                   "... registers <R1>x<R2>x<R3> one-launch" where such an M (1296 ... 3072, 8820, 9261) is
 *                                                   at least 1.25 x shorter than the power of two
 *                                                   (plan option "bluestein_smooth_m" = 0 brings the power of two back)
 * Rely on `fourier_hip_describe_*`, not on this list, where the accuracy class (direct versus chirp-z) matters.  NULL on failure. */
struct fourier_fft_float *fourier_hip_create_float(FOURIER_SIZE_TYPE size, int device);
struct fourier_fft_double *fourier_hip_create_double(FOURIER_SIZE_TYPE size, int device);

/* `Fft::size()` (fft.rs:45). 0 for a NULL handle. */
FOURIER_SIZE_TYPE fourier_hip_size_float(const FOURIER_STRUCT fourier_fft_float *);
FOURIER_SIZE_TYPE fourier_hip_size_double(const FOURIER_STRUCT fourier_fft_double *);

/* Batched `Fft::transform` on DEVICE memory: `batch` contiguous transforms, transform b at element
 * offset b*size, interleaved complex.  d_in == d_out selects in-place (`transform_in_place`).
 * Enqueued on `stream` (a hipStream_t, NULL = default stream); the kernels are only enqueued, the call
 * does not wait for them.  Plans that need a plan-owned device buffer (in-place calls of the two-pass
 * plans, three-pass plans, the Bluestein work array) allocate it on the first call whose batch is larger
 * than any before -- hipMalloc / hipFree synchronise the device -- unless fourier_hip_reserve_* was
 * called for at least that batch first; after a reserve the call never allocates and can be captured
 * into a HIP graph.  Partial overlap of d_in and d_out is not allowed. */
int fourier_hip_transform_batch_float(const FOURIER_STRUCT fourier_fft_float *, const void *d_in,
                                      void *d_out, FOURIER_SIZE_TYPE batch, int transform,
                                      void *stream);
int fourier_hip_transform_batch_double(const FOURIER_STRUCT fourier_fft_double *, const void *d_in,
                                       void *d_out, FOURIER_SIZE_TYPE batch, int transform,
                                       void *stream);

/* Pre-size the plan-owned device buffers (scratch / Bluestein work array) for calls of up to `batch`
 * transforms, in place (in_place != 0) or out of place.  May synchronise the device; a later
 * fourier_hip_transform_batch_* with batch <= `batch` and the same placement does not allocate. */
int fourier_hip_reserve_float(const FOURIER_STRUCT fourier_fft_float *, FOURIER_SIZE_TYPE batch, int in_place);
int fourier_hip_reserve_double(const FOURIER_STRUCT fourier_fft_double *, FOURIER_SIZE_TYPE batch, int in_place);

/* Device index the plan lives on (its tables, scratch and kernels); -1 for a NULL handle.  Buffers
 * passed to fourier_hip_transform_batch_* must be resident on (or mapped into) that device. */
int fourier_hip_device_float(const FOURIER_STRUCT fourier_fft_float *);
int fourier_hip_device_double(const FOURIER_STRUCT fourier_fft_double *);

/* Blocks until everything queued on `stream` (a hipStream_t, NULL = the NULL stream) of the plan's device has
 * finished -- the wait that follows a stream-ordered fourier_hip_transform_batch_* for a caller that does not own a
 * HIP runtime of its own (the Rust shim, a ctypes binding). */
int fourier_hip_synchronize_float(const FOURIER_STRUCT fourier_fft_float *, void *stream);
int fourier_hip_synchronize_double(const FOURIER_STRUCT fourier_fft_double *, void *stream);

/* Batched `Fft::transform` on HOST memory -- what a caller of the reference holds (one slice per transform,
 * fourier-algorithms/src/fft.rs:48-61), `batch` of them contiguously.  The transforms are streamed through the
 * device in chunks (pinned staging; the host-to-device copy of one chunk, the kernels of the previous one and the
 * device-to-host copy of the one before overlap), so the rate is PCIe's, not one call's latency.  Synchronous:
 * `out` is complete on return.  in == out selects in-place; partial overlap is not allowed. */
int fourier_hip_transform_batch_host_float(const FOURIER_STRUCT fourier_fft_float *,
                                           const FOURIER_COMPLEX_FLOAT_TYPE *in, FOURIER_COMPLEX_FLOAT_TYPE *out,
                                           FOURIER_SIZE_TYPE batch, int transform);
int fourier_hip_transform_batch_host_double(const FOURIER_STRUCT fourier_fft_double *,
                                            const FOURIER_COMPLEX_DOUBLE_TYPE *in, FOURIER_COMPLEX_DOUBLE_TYPE *out,
                                            FOURIER_SIZE_TYPE batch, int transform);

/* Status of the LAST call that can fail made on this handle, and the text of a status.  The entry points that do
 * work -- the four legacy `void` transforms of Part 1, fourier_hip_transform_batch[_host]_*, _reserve_*,
 * _synchronize_*, _profile_* -- reset it to FOURIER_HIP_OK on entry and record their own failure, if any: a
 * successful call after a failed one reads FOURIER_HIP_OK again, and the legacy `void` entry points report through
 * this query only.  The pure queries (_size_*, _device_*, _describe_*, _model_bytes_*, _slot_names_*, _last_status_*
 * itself) and _set_option_* (which returns its own status) leave it untouched. */
int fourier_hip_last_status_float(const FOURIER_STRUCT fourier_fft_float *);
int fourier_hip_last_status_double(const FOURIER_STRUCT fourier_fft_double *);
const char *fourier_hip_status_string(int status);

/* Tunables (return FOURIER_HIP_OK or FOURIER_HIP_INVALID_ARGUMENT; "specialise" and "register_stages" also FOURIER_HIP_UNSUPPORTED).  A handle is Send, not
 * Sync, as in the reference (RefCell scratch, autosort/mod.rs:54): fourier_hip_set_option_* must not run concurrently with a
 * transform or another call on the SAME handle ("specialise" swaps the engine the plan executes with).
 *   "chunk_bytes"  bytes of one batch chunk pushed through all passes before the next chunk starts
 *                  (keeps the inter-pass intermediate inside the 256 MiB Infinity Cache); 0 = whole batch
 *   "scratch"      1 = always route the intermediate through the plan's reused scratch buffer,
 *                  0 = use the output buffer as intermediate when out of place (default)
 *   "xcd_swizzle"  1 (default) = XCD-aware workgroup->tile mapping (each XCD owns a contiguous run of transforms),
 *                  2 = XCDs interleaved over adjacent transforms, 3 = each XCD owns an eighth of every transform's
 *                  tiles, 4 = contiguous transforms per XCD walked band-major (the others measured slower or equal,
 *                  4 is "tile_walk" with bands of an eighth of a row), 0 = plain blockIdx order
 *   "bluestein_fusion" 1 (default where the inner FFT has >= 2 passes) = chirp steps fused into the inner passes
 *   "host_chunk_bytes" bytes of one chunk of fourier_hip_transform_batch_host_* (default 32 MiB, four in flight)
 *   "bluestein_conv"   1 (default with bluestein_fusion) = the forward inner FFT's last pass, the multiply by the
 *                  transformed chirp and the inverse inner FFT's first pass run as one launch
 *   "bluestein_smooth_m" 1 (default) = a Bluestein plan whose power-of-two work array would be swept three times takes M = L1 x L2, a
 *                  product of two tile lengths (64 ... 512, prime factors up to 7; the smallest that reaches 2N - 1, or one up to 4 % longer
 *                  whose lengths split more evenly into register stages), where that is at least 1.6 x shorter (f64: 1.44 x while the middle sweep's
 *                  tile stays within 336 points) (N = 16411: M = 32928 =
 *                  196 x 168 instead of 65536, f32 +39 %, f64 +53 %; N = 10007 f64: 20160 instead of 32768, +31 %); 0 = always the
 *                  reference's next power of two (bluesteins.rs:110); 2 = wherever such a product exists (measurements).  Rebuilds the plan's tables when the value changes (not while a
 *                  transform is in flight on the handle); INVALID_ARGUMENT on a plan that is not Bluestein.  Same tolerance class.
 *   "bluestein_chirp_compute" 1 = the fused chirp-in pass computes exp(-i*pi*k^2/N) (row table x column table x an
 *                  exact-exponent cross term) instead of reading the N-entry chirp table, a quarter of that pass's
 *                  memory traffic; default 1 where it was measured faster (first pass of length >= 1024 and a table
 *                  of >= 4 MiB, e.g. N = 999983), else 0.  Same tolerance class, not the same bits.
 *   "bluestein_reference_chirp" 1 = build the chirp tables from the reference's own expression, theta = k^2 * pi / N evaluated UNREDUCED in
 *                  f64 (bluesteins.rs:10,31,57), instead of from k^2 mod 2N reduced exactly: for a caller who wants the reference's f64
 *                  results rather than the exact DFT.  The reference's form costs it N * 1e-16 of angle -- 1.7e-10 of the result at
 *                  N = 999983 --; with the option the engine agrees with the reference's arithmetic to f64 rounding (1e-15), without it with
 *                  the exact DFT to 1e-15 (profiles/r05_s18_reference_chirp.jsonl).  Rebuilds two tables on the host, synchronises the device;
 *                  the chirp-in pass then reads its table ("bluestein_chirp_compute" off).  Default 0.  INVALID_ARGUMENT on a non-Bluestein plan.
 *   "specialise"   1 = compile this length's own LDS mixed-radix kernel with hipRTC, now (about a second, once per length,
 *                  device and process), and run it from the next call on.  For a length whose prime factors stop at 13, that
 *                  fits a compute unit's LDS and has no ahead-of-time per-length kernel -- by default it runs the
 *                  runtime-parameterised kernel (24-36 % of the HBM peak where per-length kernels reach 45-60 %) or, beyond
 *                  that kernel's reach, Bluestein.  OK and unchanged for a plan that already runs a per-length kernel;
 *                  FOURIER_HIP_UNSUPPORTED -- the plan keeps its route -- for any other length, where libhiprtc is not
 *                  installed, or where the compilation fails.  Beyond the LDS limit (up to 2^26 points) the option replaces a
 *                  Bluestein plan by two or three column-tile passes whose lengths (64 ... 1024 points) have prime factors up to 13
 *                  (143000 = 440 x 325, 5^8 = 625 x 625), compiled the same way, where such a factorisation exists.  Compiled code objects are kept in an
 *                  on-disk cache (below), so a length costs its second once per machine.  Compilation never happens implicitly unless
 *                  the library-wide default "specialise_at_create" is raised to 2 (fourier_hip_set_default_option).  Same tolerance
 *                  class as the default route, not the same bits.
 *   "tile_walk"    order in which an XCD walks the column tiles of its transforms: tiles per band | transforms per group << 8 | 1 << 19
 *                  for transform-fastest | 1 << 20 for strided bands (every (tiles / band)-th tile instead of adjacent ones; measured slower);
 *                  0 = tile-major (the default except f32 N = 2^20, which walks bands of eight tiles).  "tile_walk_last": the same encoding
 *                  for the LAST pass of a plain multi-pass plan alone (0 = as the other passes)
 *   "stream_pipeline" chunk | slots << 16 (| 1 << 24: both passes on ONE internal stream, a control): the two passes of a two-pass
 *                  power-of-two plan chunk by chunk over two internal streams, ordered by events only -- pass 0 of chunk k+1 beside pass 1 of
 *                  chunk k -- with the intermediate in a plan-owned ring of `slots` (>= 2) chunks of `chunk` transforms.  The same kernels on the same
 *                  data: the same bits.  Measured on MI355X: level with the two whole-batch launches for chunks of 128+ transforms, slower
 *                  below (launch and event latency); what it buys is MEMORY -- an in-place call then needs the ring (two chunks) instead of a
 *                  scratch of the whole batch.  The call stays stream-ordered on the caller's stream (forked into and joined from the internal
 *                  streams; capturable after fourier_hip_reserve_*).  0 = off (default).  INVALID_ARGUMENT on any other plan.
 *   "register_stages" 1 = a 2^a * 3^b length that runs the LDS kernel on the reference's own schedule (bit-identical to the reference's CPU
 *                  arithmetic as restated in oracle/) takes the register-stage kernel that fourier_amd/csrc/regfft_shapes.h lists for it ON REQUEST
 *                  instead ("stockham registers <R1>x<R2>[x<R3>] one-launch": the same values within rounding, not the same bits; 24 lengths in
 *                  f32, 34 in f64, each 1.04 ... 1.44 x faster -- 4608 f32 0.46 -> 0.56 of the HBM peak, 13122 f32 0.32 -> 0.41, 2592 f64 0.57 -> 0.72);
 *                  0 = back to the default.  FOURIER_HIP_UNSUPPORTED where no such kernel is listed (the plan is unchanged); OK and
 *                  unchanged on a plan that runs register stages by default.
 *   "l2_fused"     (lib/libfourier_experiments.so only; INVALID_ARGUMENT in the product library; so is
 *                  "last_pass_prefetch", the persistent prefetching last pass of DESIGN.md section 4) 1 = run both
 *                  passes of a two-pass plan in ONE launch with the intermediate parked in the XCD's L2 (persistent
 *                  workgroups, per-XCD work queues; f32 2^16..2^18, f64 2^15..2^17 only).  Same results bit for
 *                  bit; measured 30-45 % slower than the two-launch plan on MI355X.  Calls
 *                  under this option are synchronous: the kernel's bounded waits report a time-out through a flag
 *                  that is read back before the call returns (FOURIER_HIP_RUNTIME_ERROR).  "l2_fused_depth" (1..8
 *                  windows per XCD) and "l2_fused_grid" (persistent workgroups) tune it. */
int fourier_hip_set_option_float(FOURIER_STRUCT fourier_fft_float *, const char *key, long long value);
int fourier_hip_set_option_double(FOURIER_STRUCT fourier_fft_double *, const char *key, long long value);

/* Library-wide defaults for plans created AFTERWARDS (any thread; plans that exist keep what they have).  The keys:
 *   "specialise_at_create"  what `create` does for a length whose prime factors stop at 13 and that has no ahead-of-time route
 *                  (it would run the runtime-parameterised LDS kernel or Bluestein):
 *                    0  nothing: run-time kernels only through fourier_hip_set_option_*(h, "specialise", 1)
 *                    1  (default) load its specialised kernels where the on-disk code-object cache holds ALL of them (a few
 *                       milliseconds; nothing is ever compiled implicitly) -- a length specialised once is fast in every later process
 *                    2  ... and compile what the cache lacks (hipRTC, about a second per new length and machine, inside `create`)
 *                  so that a drop-in caller of fourier_create_float / create_fft_f32 reaches the specialised kernels with one call at
 *                  start-up, or with NO code change through the environment variable FOURIER_HIP_SPECIALISE=0|1|2 (read once, before the
 *                  first plan; the function overrides it).
 *   "register_stages_at_create"  0 (default) / 1: a 2^a * 3^b length with a register-stage kernel listed on request (plan option
 *                  "register_stages" above) takes it at create -- faster by 1.04 ... 1.44 x, the reference's values within rounding instead of
 *                  its bits; environment variable FOURIER_HIP_REGISTER_STAGES=1 (read once, before the first plan; the function overrides it).
 * Environment the library reads, all of it: FOURIER_HIP_VERBOSE (error text on stderr), FOURIER_HIP_SPECIALISE, FOURIER_HIP_REGISTER_STAGES (above) and the
 * location of the code-object cache: $FOURIER_HIP_CACHE_DIR, else $XDG_CACHE_HOME/fourier-hip, else $HOME/.cache/fourier-hip (an EMPTY
 * FOURIER_HIP_CACHE_DIR switches the disk cache off).  Cache files are keyed by device architecture, precision, kernel kind, length
 * and a hash of the embedded kernel sources, the compile options and the HIP runtime's version: a library or ROCm update never loads a stale
 * kernel.  A cache entry is executed on the device, so it is trusted only if its directory belongs to the calling user and nobody else may write it,
 * the entry itself is a regular file of that user that nobody else may write, opened without following a symbolic link, and it names the very
 * key (precision, kind, length, LDS bytes) and payload length it is read for; anything else of the user's under that name is discarded, anything of
 * another user's is ignored (files are created 0600, the directory 0700).  `fourier_warm_cache` (packaging/warm_cache.c, CMake target `warm_cache`)
 * and `python -m fourier_amd.warm_cache` fill the cache at install time: see INTEGRATION.md section 1.
 * Returns FOURIER_HIP_OK or FOURIER_HIP_INVALID_ARGUMENT (unknown key / value); _get_ returns the value or -1. */
int fourier_hip_set_default_option(const char *key, long long value);
long long fourier_hip_get_default_option(const char *key);

/* Human-readable plan description ("stockham 1024x1024 ..."), valid until the handle is destroyed. */
const char *fourier_hip_describe_float(const FOURIER_STRUCT fourier_fft_float *);
const char *fourier_hip_describe_double(const FOURIER_STRUCT fourier_fft_double *);

/* HBM bytes this plan reads+writes per transform across all its kernels (design traffic model). */
double fourier_hip_model_bytes_float(const FOURIER_STRUCT fourier_fft_float *);
double fourier_hip_model_bytes_double(const FOURIER_STRUCT fourier_fft_double *);

/* Measurement hook: runs ONE batched transform exactly like fourier_hip_transform_batch_*, with a
 * HIP event pair around every kernel launch on `stream`, waits for it, and returns per kernel slot
 * (launch order; names from fourier_hip_slot_names_*) the summed duration in ms and launch count. */
int fourier_hip_profile_float(const FOURIER_STRUCT fourier_fft_float *, const void *d_in, void *d_out,
                              FOURIER_SIZE_TYPE batch, int transform, void *stream, int nslots,
                              float *ms_sum, int *launches);
int fourier_hip_profile_double(const FOURIER_STRUCT fourier_fft_double *, const void *d_in, void *d_out,
                               FOURIER_SIZE_TYPE batch, int transform, void *stream, int nslots,
                               float *ms_sum, int *launches);
/* Comma-separated kernel slot names ("pass0,pass1" / "blu_pre,fwd_pass0,...,blu_post"). */
const char *fourier_hip_slot_names_float(const FOURIER_STRUCT fourier_fft_float *);
const char *fourier_hip_slot_names_double(const FOURIER_STRUCT fourier_fft_double *);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace c */

/* Header-only C++ RAII wrapper, same shape as the reference's (fourier.h:64-128). */
enum class transform {
  fft = ::fourier::c::FOURIER_TRANSFORM_FFT,
  ifft = ::fourier::c::FOURIER_TRANSFORM_IFFT,
  unscaled_ifft = ::fourier::c::FOURIER_TRANSFORM_UNSCALED_IFFT,
  sqrt_scaled_fft = ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_FFT,
  sqrt_scaled_ifft = ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_IFFT,
};

template <typename T> struct fft;

#define FOURIER_DEFINE_CXX_WRAPPER(T, SUFFIX)                                                      \
  template <> struct fft<T> {                                                                      \
    explicit fft(std::size_t size)                                                                 \
        : impl(::fourier::c::fourier_create_##SUFFIX(size), ::fourier::c::fourier_destroy_##SUFFIX) {} \
    fft() = delete;                                                                                \
    fft(const fft &) = delete;                                                                     \
    fft(fft &&) = default;                                                                         \
    fft &operator=(const fft &) = delete;                                                          \
    fft &operator=(fft &&) = default;                                                              \
    ~fft() = default;                                                                              \
    void transform_in_place(::std::complex<T> *x, transform t) const {                             \
      ::fourier::c::fourier_transform_in_place_##SUFFIX(impl.get(), x, static_cast<int>(t));       \
    }                                                                                              \
    void transform(const ::std::complex<T> *in, ::std::complex<T> *out, transform t) const {       \
      ::fourier::c::fourier_transform_##SUFFIX(impl.get(), in, out, static_cast<int>(t));          \
    }                                                                                              \
    /* device-resident batched execution (extension) */                                           \
    int transform_batch_device(const void *d_in, void *d_out, std::size_t batch,               \
                               ::fourier::transform t,                                            \
                               void *stream = nullptr) const {                                     \
      return ::fourier::c::fourier_hip_transform_batch_##SUFFIX(impl.get(), d_in, d_out, batch,    \
                                                                static_cast<int>(t), stream);      \
    }                                                                                              \
    /* many transforms in host memory, streamed through the device (extension) */                 \
    int transform_batch_host(const ::std::complex<T> *in, ::std::complex<T> *out, std::size_t batch, \
                             ::fourier::transform t) const {                                       \
      return ::fourier::c::fourier_hip_transform_batch_host_##SUFFIX(impl.get(), in, out, batch,   \
                                                                     static_cast<int>(t));         \
    }                                                                                              \
    explicit operator bool() const { return static_cast<bool>(impl); }                             \
                                                                                                   \
  private:                                                                                         \
    ::std::unique_ptr<::fourier::c::fourier_fft_##SUFFIX, void (*)(::fourier::c::fourier_fft_##SUFFIX *)> impl; \
  };
FOURIER_DEFINE_CXX_WRAPPER(float, float)
FOURIER_DEFINE_CXX_WRAPPER(double, double)
#undef FOURIER_DEFINE_CXX_WRAPPER

} /* namespace fourier */
#endif

#endif /* FOURIER_H_ */
