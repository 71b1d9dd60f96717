// engine.cpp -- host side of libfourier.so: plan factory, pass scheduling, C ABI.
//
// Mirrors the reference's plan layer one level up:
//   create_fft_f32/f64  (fourier/src/lib.rs:31-60)        -> Plan<T>::create      (Stockham, else Bluestein)
//   Autosort::new       (autosort/mod.rs:104-134)         -> Pow2Engine<T>        (big-radix pass schedule)
//   initialize_twiddles (autosort/mod.rs:24-46)           -> make_stage_tables / make_two_level (f64 trig, cast)
//   Bluesteins::new     (bluesteins.rs:109-130, :18-61)   -> Plan<T>::init_bluestein
//   apply_stages / apply (mod.rs:313-404, bluesteins.rs:215-259) -> Plan<T>::exec
//   fourier-ffi C ABI   (fourier-ffi/src/lib.rs:14-106)   -> extern "C" block at the end
// Compiled with hipcc for gfx950; the same file builds against tests/emu/hipemu.h (-DFOURIER_EMU)
// for CPU-side logic tests only.
#ifndef FOURIER_EMU
#include <hip/hip_runtime.h>
#endif

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fft_kernels.h"
#include "../../include/fourier.h"

namespace fourier_hip {

struct EngineError : std::runtime_error {
  int status;
  EngineError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

#define HIP_CHECK(expr)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      throw EngineError(e_ == hipErrorOutOfMemory ? ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY         \
                                                  : ::fourier::c::FOURIER_HIP_RUNTIME_ERROR,        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                         \
  } while (0)

#ifdef FOURIER_EMU
#define FOURIER_LAUNCH(fn, grid, block, smem, stream, arg) hipemu::launch(dim3((unsigned)(grid)), dim3((unsigned)(block)), (smem), (fn), (arg))
#else
#define FOURIER_LAUNCH(fn, grid, block, smem, stream, arg)                                \
  do {                                                                                    \
    (fn)<<<dim3((unsigned)(grid)), dim3((unsigned)(block)), (smem), (stream)>>>(arg);     \
    HIP_CHECK(hipGetLastError());                                                         \
  } while (0)
#endif

// Development switches (environment variables read at plan creation) exist only in builds with -DFOURIER_EXPERIMENTS:
// the emulator build of the CPU tests and lib/libfourier_experiments.so (A/B sessions, the GPU tests of the
// measured-slower designs).  The product library's plan selection never depends on the environment of the process
// that links it; FOURIER_HIP_VERBOSE (error text on stderr) is the one variable it reads.  The same flag compiles the
// experiment kernels (fft_l2fused_kernel, fft_last_split_kernel): DESIGN.md section 4 has their measurements.
#ifdef FOURIER_EXPERIMENTS
static inline const char* dev_env(const char* name) { return getenv(name); }
#else
static inline const char* dev_env(const char*) { return nullptr; }
#endif

// kernels that use more than 48 KiB of dynamic LDS must say so once
static void raise_smem_limit(const void* fn, size_t smem) {
  if (smem > 48 * 1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
}

// ---------------------------------------------------------------------------------------------
// device memory RAII
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void ensure(size_t n) {
    if (n <= bytes) return;
    release();
    HIP_CHECK(hipMalloc(&p, n));
    bytes = n;
  }
  template <typename V> void upload(const std::vector<V>& h) {
    ensure(h.size() * sizeof(V));
    if (!h.empty()) HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(V), hipMemcpyHostToDevice));
  }
};

// page-locked host staging buffer, mapped into the device address space (legacy host-buffer ABI)
struct PinnedBuf {
  void* h = nullptr;  // host address
  void* d = nullptr;  // the same memory as the device sees it
  size_t bytes = 0;
  PinnedBuf() {}
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  ~PinnedBuf() { release(); }
  void release() {
    if (h) (void)hipHostFree(h);
    h = d = nullptr;
    bytes = 0;
  }
  void ensure(size_t n) {
    if (n <= bytes) return;
    release();
    HIP_CHECK(hipHostMalloc(&h, n, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer(&d, h, 0));
    bytes = n;
  }
};

// ---------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg)
struct Profiler {
  hipStream_t stream;
  struct Span { int slot; hipEvent_t a, b; };
  std::vector<Span> spans;
  explicit Profiler(hipStream_t s) : stream(s) {}
  ~Profiler() { for (auto& sp : spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); } }
  void begin(int slot) {
    Span sp{slot, nullptr, nullptr};
    HIP_CHECK(hipEventCreate(&sp.a));
    HIP_CHECK(hipEventCreate(&sp.b));
    HIP_CHECK(hipEventRecord(sp.a, stream));
    spans.push_back(sp);
  }
  void end() { HIP_CHECK(hipEventRecord(spans.back().b, stream)); }
  void collect(int nslots, float* ms_sum, int* launches) {
    for (int i = 0; i < nslots; ++i) { ms_sum[i] = 0; launches[i] = 0; }
    for (auto& sp : spans) {
      HIP_CHECK(hipEventSynchronize(sp.b));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
      if (sp.slot >= 0 && sp.slot < nslots) { ms_sum[sp.slot] += ms; launches[sp.slot] += 1; }
    }
  }
};
#define PROF_BEGIN(prof, slot) do { if (prof) (prof)->begin(slot); } while (0)
#define PROF_END(prof) do { if (prof) (prof)->end(); } while (0)

// ---------------------------------------------------------------------------------------------
// kernel registry: one tile shape (CG) per pass length L
typedef void (*PassKernel)(PassArgs);
struct KernelInfo {
  PassKernel fn = nullptr;
  int L = 0, CG = 0, NT = 0, COLS = 0, R3 = 0;
  int split = 0;  // 1: two workgroups per tile (fft_last_split_kernel), grid = 2 x tiles
  size_t smem = 0;
};

template <typename T, int L, int CG, int MODE, int IO = IO_PLAIN> static KernelInfo make_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_pass_kernel<T, L, CG, MODE, IO>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::smem_bytes(MODE) + (IO == IO_BLU_IN ? C::TABV_BYTES : 0);
  return k;
}

#ifdef FOURIER_EXPERIMENTS
// last pass of length L on half tiles: the register tile (and the thread count) of a length-L/2 pass
template <typename T, int L, int CG, int IO = IO_PLAIN> static KernelInfo make_split_info() {
  using C = TileCfg<T, L / 2, CG>;
  KernelInfo k;
  k.fn = &fft_last_split_kernel<T, L / 2, CG, IO>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3; k.split = 1;
  k.smem = C::smem_bytes(MODE_LAST);
  return k;
}
#endif

// tile widths (column groups of 16 bytes) per pass length; overridable for A/B builds
#ifndef FOURIER_CG_512
#define FOURIER_CG_512 8
#endif
#ifndef FOURIER_CG_1024
#define FOURIER_CG_1024 8
#endif
#ifndef FOURIER_CG_2048
#define FOURIER_CG_2048 8
#endif
// L = 2048 holds a 256 KiB tile per workgroup at 16 columns -- one workgroup per CU, no overlap of its load and
// compute phases.  Default plans therefore run the FIRST pass on 64-byte-wide tiles (8 columns, 128 KiB, two
// workgroups per CU; the transposed store does not care about the tile width): 6.3-6.6 vs 7.2-7.8 ms per 1024
// transforms of 2^21 (profiles/r02_s2_l2048_and_xcd_fused_ab.jsonl).  FOURIER_WIDE_2048=1 in the environment at plan
// creation brings the 16-column first pass back (A/B).
#ifndef FOURIER_CG_2048_FIRST
#define FOURIER_CG_2048_FIRST 4
#endif

#ifndef FOURIER_CG_4096
#define FOURIER_CG_4096 2
#endif
template <typename T> static KernelInfo get_kernel(int L, int mode, int io = IO_PLAIN) {
  // first pass of length 4096 on 32-byte-wide tiles (128 KiB, two workgroups per CU): 2^22 = 4096 x 1024
  if (L == 4096 && mode == MODE_FIRST && io == IO_PLAIN) return make_info<T, 4096, FOURIER_CG_4096, MODE_FIRST>();
  if (L == 2048 && !dev_env("FOURIER_WIDE_2048")) {
    if (mode == MODE_FIRST)
      return io == IO_BLU_IN ? make_info<T, 2048, FOURIER_CG_2048_FIRST, MODE_FIRST, IO_BLU_IN>()
                             : make_info<T, 2048, FOURIER_CG_2048_FIRST, MODE_FIRST>();
    // half tiles for the last pass were measured 5-7 % SLOWER than the 16-column kernel (profiles/r02_s2_*_ab.jsonl:
    // 13.6-14.0 vs 13.1 ms per 1024 transforms of 2^22); kept behind FOURIER_SPLIT_2048=1 for experiments
#ifdef FOURIER_EXPERIMENTS
    if (mode == MODE_LAST && dev_env("FOURIER_SPLIT_2048"))
      return io == IO_BLU_OUT ? make_split_info<T, 2048, FOURIER_CG_1024, IO_BLU_OUT>() : make_split_info<T, 2048, FOURIER_CG_1024>();
#endif
  }
#define FK(LL, CGG)                                                                              \
  case LL:                                                                                       \
    switch (mode) {                                                                              \
      case MODE_FIRST:                                                                           \
        return io == IO_BLU_IN ? make_info<T, LL, CGG, MODE_FIRST, IO_BLU_IN>()                  \
                               : make_info<T, LL, CGG, MODE_FIRST>();                            \
      case MODE_MID: return make_info<T, LL, CGG, MODE_MID>();                                   \
      case MODE_LAST:                                                                            \
        return io == IO_BLU_OUT ? make_info<T, LL, CGG, MODE_LAST, IO_BLU_OUT>()                 \
                                : make_info<T, LL, CGG, MODE_LAST>();                            \
      default: return make_info<T, LL, CGG, MODE_ROWS>();                                        \
    }
#define FK_ROWS_ONLY(LL, CGG) \
  case LL: return make_info<T, LL, CGG, MODE_ROWS>();
  switch (L) {
    FK_ROWS_ONLY(16, 64)
    FK_ROWS_ONLY(32, 32)
    FK(64, 16)
    FK(128, 16)
    FK(256, 16)
    FK(512, FOURIER_CG_512)
    FK(1024, FOURIER_CG_1024)
    FK(2048, FOURIER_CG_2048)
    default: break;
  }
#undef FK
#undef FK_ROWS_ONLY
  throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no kernel for pass length " + std::to_string(L));
}

#ifndef FOURIER_CONV_CG_1024
#define FOURIER_CONV_CG_1024 FOURIER_CG_1024
#endif
// fft_conv_kernel: forward LAST + (.) w + inverse FIRST of a Bluestein plan, same tile shapes as the passes
template <typename T, int L, int CG> static KernelInfo make_conv_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &fft_conv_kernel<T, L, CG>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::smem_bytes(MODE_FIRST);
  return k;
}
template <typename T> static KernelInfo get_conv_kernel(int L) {
  switch (L) {
    case 64: return make_conv_info<T, 64, 16>();
    case 128: return make_conv_info<T, 128, 16>();
    case 256: return make_conv_info<T, 256, 16>();
    case 512: return make_conv_info<T, 512, FOURIER_CG_512>();
    case 1024: return make_conv_info<T, 1024, FOURIER_CONV_CG_1024>();
    case 2048: return make_conv_info<T, 2048, FOURIER_CG_2048>();
    default: break;
  }
  throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no conv kernel for pass length " + std::to_string(L));
}

enum { MODE_TWOLEVEL = 4 };  // host-side tag for fft_twolevel_kernel (both passes in one launch)

template <typename T, int L1, int L2> static KernelInfo make_twolevel_info() {
  constexpr int VEC = 16 / (2 * (int)sizeof(T));
  using CA = TileCfg<T, L1, L2 / VEC>;
  using CB = TileCfg<T, L2, L1 / VEC>;
  KernelInfo k;
  k.fn = &fft_twolevel_kernel<T, L1, L2>;
  k.L = L1; k.CG = L2 / VEC; k.NT = CA::NT; k.COLS = L2; k.R3 = 1;
  // the two in-tile exchanges and the transposes between them share one buffer (both role orders: the one-launch chirp-z
  // runs the L2 x L1 problem behind the L1 x L2 one)
  k.smem = std::max({CA::EXCH_BYTES, CB::EXCH_BYTES, TwolevelTr<T, L1, L2>::BYTES, TwolevelTr<T, L2, L1>::BYTES});
  return k;
}
// single-launch plans: 2^11 = 64x32 (72 % of the HBM peak vs 57 % for the row kernel), 2^12 = 64x64,
// 2^13 = 128x64, 2^14 = 128x128, 2^15 = 256x128 (f32 only: the transform must fit one workgroup's
// registers, at most 1024 threads x 16 points x VEC)
template <typename T> static bool get_twolevel_kernel(int k, KernelInfo& info, int& l1, int& l2) {
  switch (k) {
    case 11: info = make_twolevel_info<T, 64, 32>(); l1 = 64; l2 = 32; return true;
    case 12: info = make_twolevel_info<T, 64, 64>(); l1 = 64; l2 = 64; return true;
    case 13: info = make_twolevel_info<T, 128, 64>(); l1 = 128; l2 = 64; return true;
    case 14: info = make_twolevel_info<T, 128, 128>(); l1 = 128; l2 = 128; return true;
    case 15:
      if constexpr (sizeof(T) == 4) { info = make_twolevel_info<T, 256, 128>(); l1 = 256; l2 = 128; return true; }
      return false;
    default: return false;
  }
}

template <typename T, int L1, int L2> static KernelInfo make_blu_small_info() {
  KernelInfo k = make_twolevel_info<T, L1, L2>();
  k.fn = &bluestein_small_kernel<T, L1, L2>;
  return k;
}
template <typename T, int L, int CG> static KernelInfo make_blu_rows_info() {
  using C = TileCfg<T, L, CG>;
  KernelInfo k;
  k.fn = &bluestein_rows_kernel<T, L, CG>;
  k.L = L; k.CG = CG; k.NT = C::NT; k.COLS = C::COLS; k.R3 = C::R3;
  k.smem = C::EXCH_BYTES;
  return k;
}
template <typename T> static bool get_blu_small_kernel(int k, KernelInfo& info) {
  switch (k) {
    case 4: info = make_blu_rows_info<T, 16, 64>(); return true;   // same tile shapes as the row kernels
    case 5: info = make_blu_rows_info<T, 32, 32>(); return true;
    case 6: info = make_blu_rows_info<T, 64, 16>(); return true;
    case 7: info = make_blu_rows_info<T, 128, 16>(); return true;
    case 8: info = make_blu_rows_info<T, 256, 16>(); return true;
    case 9: info = make_blu_rows_info<T, 512, FOURIER_CG_512>(); return true;
    case 10: info = make_blu_rows_info<T, 1024, FOURIER_CG_1024>(); return true;
    case 11: info = make_blu_small_info<T, 64, 32>(); return true;
    case 12: info = make_blu_small_info<T, 64, 64>(); return true;
    case 13: info = make_blu_small_info<T, 128, 64>(); return true;
    case 14: info = make_blu_small_info<T, 128, 128>(); return true;
    case 15:
      if constexpr (sizeof(T) == 4) { info = make_blu_small_info<T, 256, 128>(); return true; }
      return false;
    default: return false;
  }
}

// fft_l2fused_kernel: both passes of an N = L1 x L2 plan in one launch, intermediate in the XCD's L2
typedef void (*FusedKernel)(FusedArgs);
struct FusedInfo {
  FusedKernel fn = nullptr;
  int L1 = 0, L2 = 0, NT = 0, COLS_A = 0, COLS_B = 0;
  size_t smem = 0;
};
template <typename T, int L1, int CG1, int L2, int CG2> static FusedInfo make_fused_info() {
  using CA = TileCfg<T, L1, CG1>;
  using CB = TileCfg<T, L2, CG2>;
  FusedInfo k;
  k.fn = &fft_l2fused_kernel<T, L1, CG1, L2, CG2>;
  k.L1 = L1; k.L2 = L2; k.NT = CA::NT; k.COLS_A = CA::COLS; k.COLS_B = CB::COLS;
  const size_t sa = CA::smem_bytes(MODE_FIRST), sb = CB::smem_bytes(MODE_LAST);
  k.smem = (((sa > sb ? sa : sb) + 15) & ~(size_t)15) + 16;  // + the broadcast slot
  return k;
}
// N * sizeof(complex) <= 2 MiB and two passes: f32 2^16 .. 2^18, f64 2^15 .. 2^17 (64 KiB tiles, 256 threads)
template <typename T> static bool get_fused_kernel(int k, FusedInfo& info) {
#ifdef FOURIER_EXPERIMENTS
  if constexpr (sizeof(T) == 4) {
    switch (k) {
      case 16: info = make_fused_info<T, 256, 16, 256, 16>(); return true;
      case 17: info = make_fused_info<T, 512, 8, 256, 16>(); return true;
      case 18: info = make_fused_info<T, 512, 8, 512, 8>(); return true;
      default: return false;
    }
  } else {
    switch (k) {
      case 15: info = make_fused_info<T, 256, 16, 128, 32>(); return true;
      case 16: info = make_fused_info<T, 256, 16, 256, 16>(); return true;
      case 17: info = make_fused_info<T, 512, 8, 256, 16>(); return true;
      default: return false;
    }
  }
#else
  (void)k; (void)info;
  return false;  // measured 30-45 % slower than the two-launch plan: not in the product build
#endif
}

enum { MODE_ODD_LAST = 5 };  // host-side tag for odd_last_kernel (final radix-3^b pass of a 2^a*3^b plan)
typedef void (*OddKernel)(OddArgs);
template <typename T> static OddKernel get_odd_kernel(int r) {
  switch (r) {
    case 3: return &odd_last_kernel<T, 3>;
    case 9: return &odd_last_kernel<T, 9>;
    case 27: return &odd_last_kernel<T, 27>;
    default: return nullptr;
  }
}

static inline int ilog2(uint64_t v) { int l = 0; while ((1ull << l) < v) ++l; return l; }
static inline bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }

// exp(-2*pi*i*e/size) in f64 (the reference evaluates twiddles in f64 and casts: twiddle.rs:7-19)
static inline void unit_root(uint64_t e, uint64_t size, double& re, double& im) {
  e %= size;
  const double frac = (double)e / (double)size;  // exact for power-of-two sizes; the quarter turns are exact below
  const double ang = 2.0 * M_PI * frac;
  re = std::cos(ang);
  im = -std::sin(ang);
  if (4 * e == size) { re = 0; im = -1; }
  else if (2 * e == size) { re = -1; im = 0; }
  else if (4 * e == 3 * size) { re = 0; im = 1; }
  else if (e == 0) { re = 1; im = 0; }
}

template <typename T> struct StageTables {
  DevBuf tw1, tw2;
};

template <typename T> static void make_stage_tables(int L, StageTables<T>& st) {
  const int Q = L / 16, R2 = Q >= 16 ? 16 : Q, R3 = Q / (R2 ? R2 : 1);
  std::vector<cpx<T>> t1((size_t)Q * 16);
  for (int th = 0; th < Q; ++th)
    for (int k = 0; k < 16; ++k) {
      double re, im;
      unit_root((uint64_t)th * k, (uint64_t)L, re, im);
      t1[(size_t)th * 16 + k] = {(T)re, (T)im};
    }
  st.tw1.upload(t1);
  if (Q > 1 && R3 > 1) {
    std::vector<cpx<T>> t2((size_t)R3 * 16);
    for (int i = 0; i < R3; ++i)
      for (int k = 0; k < 16; ++k) {
        double re, im;
        unit_root((uint64_t)i * k, (uint64_t)Q, re, im);
        t2[(size_t)i * 16 + k] = {(T)re, (T)im};
      }
    st.tw2.upload(t2);
  }
}

// ---------------------------------------------------------------------------------------------
// batched power-of-two FFT: schedule of big-radix Stockham passes
template <typename T> class Pow2Engine {
 public:
  struct Pass {
    int mode;
    KernelInfo k;
    KernelInfo k_blu;  // IO_BLU_IN variant of a FIRST pass / IO_BLU_OUT variant of a LAST pass (lazy)
    bool has_blu = false;
    StageTables<T>* st2 = nullptr;  // MODE_TWOLEVEL: stage tables of the second pass length
    OddKernel odd_fn = nullptr;     // MODE_ODD_LAST
    int odd_r = 0;
    uint64_t s, size, cn;
    uint32_t lo_bits = 0;
    DevBuf tw_lo, tw_hi, tw_half;  // tw_half: split last pass, W_L^{n} for n < L/2, laid out [Q*r + th]
    StageTables<T>* st = nullptr;
  };

  // Large mixed sizes N = 2^a * 3^b (12 <= a <= 30, 1 <= b <= 3): the 2^a part runs as big-radix passes
  // (FIRST, MID...), the 3^b part as one final odd-radix Stockham pass -- the reference's own order, radix 3
  // after the powers of two (RADICES = [4,8,4,3,2], autosort/mod.rs:21).
  static bool handles_mixed(size_t n) {
    const size_t total = n;
    size_t p3 = 1;
    while (n % 3 == 0) { n /= 3; p3 *= 3; }
    return p3 > 1 && is_pow2(n) && n >= 4096 && total <= ((size_t)1 << 30);
  }

  // mirror: the pass lengths in reverse order (the inverse inner FFT of a conv-fused Bluestein plan must start
  // with the length the forward one ends with)
  // plain: the plan is used as a whole transform (not as the inner FFT of a Bluestein plan, which needs a last pass and
  // a mirror image of every length it uses)
  explicit Pow2Engine(size_t n, bool mirror = false, bool plain = false) : n_(n) {
    size_t p3 = 1, p2 = n;
    while (p2 % 3 == 0) { p2 /= 3; p3 *= 3; }
    if (!is_pow2(p2) || (p3 > 1 && p2 < 4096))
      throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "StockhamEngine: size must be 2^a or 2^a*3^b (a >= 12)");
    const int k = ilog2(p2);
    std::vector<int> lens;
    KernelInfo tl;
    int tl1 = 0, tl2 = 0;
    if (p3 == 1 && !dev_env("FOURIER_NO_TWOLEVEL") && get_twolevel_kernel<T>(k, tl, tl1, tl2)) {
      // one launch, one HBM round trip: both passes inside a workgroup
      auto pass = std::unique_ptr<Pass>(new Pass());
      pass->mode = MODE_TWOLEVEL;
      pass->k = tl;
      pass->s = 1; pass->size = n; pass->cn = 1;
      for (int L : {tl1, tl2}) {
        if (stage_.find(L) == stage_.end()) {
          auto st = std::unique_ptr<StageTables<T>>(new StageTables<T>());
          make_stage_tables<T>(L, *st);
          stage_.emplace(L, std::move(st));
        }
      }
      pass->st = stage_[tl1].get();
      pass->st2 = stage_[tl2].get();
      {  // full inter-pass twiddle table W_N^{i*k1}, laid out [k1][i] (f64 trig, cast: twiddle.rs:7-19)
        std::vector<cpx<T>> tw((size_t)n);
        for (int k1 = 0; k1 < tl1; ++k1)
          for (int i = 0; i < tl2; ++i) {
            double re, im;
            unit_root((uint64_t)i * (uint64_t)k1, n, re, im);
            tw[(size_t)k1 * tl2 + i] = {(T)re, (T)im};
          }
        pass->tw_lo.upload(tw);
      }
      tl1_ = tl1; tl2_ = tl2;
      set_smem_attribute(pass->k);
      desc_override_ = std::to_string(tl1) + "x" + std::to_string(tl2) + " one-launch";
      passes_.push_back(std::move(pass));
      return;
    }
    if (k <= 3) {
      tiny_ = true;
    } else if (k <= 11) {
      lens = {k};
    } else if (k == 22 && plain && p3 == 1 && dev_env("FOURIER_PLAN_4096")) {
      // experiment: 4096 (first pass on 32-byte-wide tiles) x 1024 instead of 2048 x 2048.  f32: 27.3 vs 25.8-26.5 ms per
      // 1024 transforms; f64: 27.4 vs 28.9 ms per 512 but 7.9 vs 7.4 ms per 128 (profiles/r02_s3_*.jsonl,
      // r02_s4_sizes.jsonl) -- no consistent gain, so the default stays 2048 x 2048
      lens = {12, 10};
    } else if (k == 23 && plain && p3 == 1 && (sizeof(T) == 4 ? !dev_env("FOURIER_THREE_PASS_2P23") : dev_env("FOURIER_TWO_PASS_2P23") != nullptr)) {
      // 2^23 = 4096 x 2048: two HBM round trips (first pass of length 4096 on 32-byte-wide tiles, 16-column last pass of
      // length 2048) instead of three at 256 x 256 x 128.  f32: 27.4-30.0 vs 34.1-34.8 ms per 512 transforms (default);
      // f64: 30.8-35.2 vs 33.5-33.8 ms per 256, no consistent gain (opt-in) -- profiles/r02_s16_plan_2p23_ab.jsonl
      lens = {12, 11};
    } else if (k <= 22) {
      lens = {(k + 1) / 2, k / 2};
    } else if (k <= 30) {
      const int k1 = (k + 2) / 3, k2 = (k - k1 + 1) / 2, k3 = k - k1 - k2;
      lens = {k1, k2, k3};
    } else {
      throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "power-of-two sizes above 2^30 are not supported");
    }
    if (mirror) std::reverse(lens.begin(), lens.end());  // either order times the same (profiles/r01_s8_nt_and_pass_order.jsonl)
    uint64_t s = 1, size = n;
    for (size_t p = 0; p < lens.size(); ++p) {
      auto pass = std::unique_ptr<Pass>(new Pass());
      const int L = 1 << lens[p];
      pass->mode = lens.size() == 1 ? MODE_ROWS : (p == 0 ? MODE_FIRST : (p + 1 == lens.size() && p3 == 1 ? MODE_LAST : MODE_MID));
      pass->k = get_kernel<T>(L, pass->mode);
      pass->s = s; pass->size = size; pass->cn = n / L;
      if (pass->mode != MODE_ROWS) {
        const uint64_t extent = (pass->mode == MODE_FIRST) ? pass->cn : s;
        if (extent % (uint64_t)pass->k.COLS != 0)
          throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "tile does not divide pass extent");
        if (!is_pow2(s)) throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "tile pass behind an odd-radix pass");  // kernels shift by log2(s)
      }
      const int Lt = pass->k.split ? L / 2 : L;  // length of the in-tile FFT (a split pass runs a half-length tile)
      auto it = stage_.find(Lt);
      if (it == stage_.end()) {
        auto st = std::unique_ptr<StageTables<T>>(new StageTables<T>());
        make_stage_tables<T>(Lt, *st);
        it = stage_.emplace(Lt, std::move(st)).first;
      }
      pass->st = it->second.get();
      if (pass->k.split) {
        std::vector<cpx<T>> wh((size_t)Lt);
        for (int nn = 0; nn < Lt; ++nn) { double re, im; unit_root((uint64_t)nn, (uint64_t)L, re, im); wh[(size_t)nn] = {(T)re, (T)im}; }
        pass->tw_half.upload(wh);
      }
      if (pass->mode == MODE_FIRST || pass->mode == MODE_MID) make_two_level(*pass, size);
      set_smem_attribute(pass->k);
      passes_.push_back(std::move(pass));
      s *= (uint64_t)L;
      size /= (uint64_t)L;
    }
    // only a plan that is used as a whole transform may fuse its two passes: the inner engine of a Bluestein plan runs
    // its passes one by one with chirp / conv fusion (and `needs_scratch` must not be switched off under it)
    if (p3 == 1 && lens.size() == 2 && !mirror && plain) init_l2fused(k);
    // odd part 3^b as radix-27 passes plus one of radix 3 / 9 / 27: twiddled middle passes, then the final one
    // (the reference's order, radix 3 after the powers of two: RADICES = [4,8,4,3,2], autosort/mod.rs:21)
    while (p3 > 1) {
      const size_t r = p3 > 27 ? 27 : p3;
      auto pass = std::unique_ptr<Pass>(new Pass());
      pass->mode = MODE_ODD_LAST;
      pass->odd_r = (int)r;
      pass->odd_fn = get_odd_kernel<T>((int)r);
      pass->s = s; pass->size = size; pass->cn = size / r;  // cn = m of this pass
      if (size != r) {  // W_size^{e}, e < size (i*k < m*R)
        std::vector<cpx<T>> tw((size_t)size);
        for (size_t e = 0; e < (size_t)size; ++e) { double re, im; unit_root(e, size, re, im); tw[e] = {(T)re, (T)im}; }
        pass->tw_lo.upload(tw);
      }
      passes_.push_back(std::move(pass));
      s *= r; size /= r; p3 /= r;
    }
  }

  // two-level table of W_size^{e}: e = (e >> lo_bits) << lo_bits | (e & mask)
  static void make_two_level(Pass& pass, uint64_t size) {
    const int lb = (ilog2(size) + 1) / 2;
    pass.lo_bits = (uint32_t)lb;
    std::vector<cpx<T>> lo((size_t)1 << lb), hi((size_t)(size >> lb) + 1);  // +1: size need not be a power of two
    for (size_t e = 0; e < lo.size(); ++e) { double re, im; unit_root(e, size, re, im); lo[e] = {(T)re, (T)im}; }
    for (size_t h = 0; h < hi.size(); ++h) { double re, im; unit_root((uint64_t)h << lb, size, re, im); hi[h] = {(T)re, (T)im}; }
    pass.tw_lo.upload(lo);
    pass.tw_hi.upload(hi);
  }
  static void set_smem_attribute(const KernelInfo& k) { raise_smem_limit((const void*)k.fn, k.smem); }

  // Whole-Bluestein-in-one-launch (bluestein_small_kernel) is available when this (inner) plan is a
  // one-launch two-level plan: it additionally needs the inter-pass table of the role-swapped L2 x L1 problem.
  bool enable_bluestein_small() {
    if (tiny_ || passes_.size() != 1) return false;
    if (blu_small_.fn) return true;  // tables already uploaded (set_option may be called repeatedly)
    if (passes_[0]->mode == MODE_ROWS) {  // M <= 1024: row core twice, COLS transforms per workgroup
      if (!get_blu_small_kernel<T>(ilog2(n_), blu_small_)) return false;
      set_smem_attribute(blu_small_);
      return true;
    }
    if (passes_[0]->mode != MODE_TWOLEVEL) return false;
    if (!get_blu_small_kernel<T>(ilog2(n_), blu_small_)) return false;
    std::vector<cpx<T>> tw(n_);
    for (int k1 = 0; k1 < tl2_; ++k1)      // swapped roles: k1' < L2, i' < L1, layout [k1'][i']
      for (int i = 0; i < tl1_; ++i) {
        double re, im;
        unit_root((uint64_t)i * (uint64_t)k1, n_, re, im);
        tw[(size_t)k1 * tl1_ + i] = {(T)re, (T)im};
      }
    passes_[0]->tw_hi.upload(tw);
    set_smem_attribute(blu_small_);
    return true;
  }
  // in/out: USER arrays (batch stride n_user); xtab: chirp (n_user), wtab: FFT'd chirp / M (n_ entries)
  void run_bluestein_small(const cpx<T>* in, cpx<T>* out, size_t batch, const void* xtab, const void* wtab, uint64_t n_user,
                           bool inverse, double scale, hipStream_t stream, Profiler* prof, unsigned nxcd) const {
    if (batch == 0) return;
    const Pass& ps = *passes_[0];
    PassArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out;
    const bool rows = (ps.mode == MODE_ROWS);
    a.tw1 = ps.st->tw1.p; a.tw2 = rows ? ps.st->tw2.p : ps.st2->tw1.p;
    a.tw_lo = ps.tw_lo.p; a.tw_hi = ps.tw_hi.p;
    a.mul = wtab; a.blu_x = xtab; a.blu_n = n_user; a.blu_swap = inverse;
    a.n = n_; a.scale = scale; a.nxcd = nxcd; a.total_cols = batch;
    const uint64_t grid = rows ? (batch + blu_small_.COLS - 1) / blu_small_.COLS : batch;
    if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
    PROF_BEGIN(prof, 0);
    FOURIER_LAUNCH(blu_small_.fn, grid, blu_small_.NT, blu_small_.smem, stream, a);
    PROF_END(prof);
  }

  // ---- XCD-fused two-pass plan (fft_l2fused_kernel): opt-in via the plan option "l2_fused"
  void init_l2fused(int k) {
    FusedInfo fi;
    if (!get_fused_kernel<T>(k, fi)) return;
    if (passes_.size() != 2 || passes_[0]->k.L != fi.L1 || passes_[1]->k.L != fi.L2) return;
    if ((n_ / fi.L1) % (size_t)fi.COLS_A != 0 || (size_t)fi.L1 % (size_t)fi.COLS_B != 0) return;
    fused_ = fi;
    raise_smem_limit((const void*)fused_.fn, fused_.smem);
    int per_cu = 0, cus = 0, dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_.fn, fused_.NT, fused_.smem));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    fused_grid_ = (unsigned)std::max(1, per_cu) * (unsigned)std::max(1, cus);  // persistent: every workgroup resident
    const size_t bytes = n_ * sizeof(cpx<T>);
    fused_depth_ = bytes <= (512u << 10) ? 3 : 2;  // windows per XCD: <= 2 MiB of the 4 MiB L2 (profiles/r02_membench.jsonl, l2x)
    if (const char* e = dev_env("FOURIER_L2_FUSED")) fused_on_ = atoi(e) != 0;
  }
  bool has_l2fused() const { return fused_.fn != nullptr; }
  bool l2fused_enabled() const { return fused_on_ && fused_.fn; }
  void set_l2fused(bool on) { fused_on_ = on && fused_.fn; }
  bool set_l2fused_depth(unsigned d) {
    if (!fused_.fn || d < 1 || d > 8) return false;
    fused_depth_ = d;
    fused_window_.release();
    return true;
  }
  void set_l2fused_grid(unsigned g) { if (g) fused_grid_ = g; }
  static constexpr size_t FUSED_MAX_BATCH = 16384;  // transforms per launch (sizes the zeroed control block)
  // pre-size the window and control block (fourier_hip_reserve_*): launches then never allocate
  void reserve_l2fused(size_t batch) const {
    if (!l2fused_enabled()) return;
    fused_window_.ensure((size_t)FUSED_XCC_IDS * fused_depth_ * n_ * sizeof(cpx<T>));
    fused_ctrl_.ensure(fused_ctrl_words(std::min(batch, FUSED_MAX_BATCH)) * sizeof(uint32_t));
  }
  void run_l2fused(const cpx<T>* in, cpx<T>* out, size_t batch, bool inverse, double scale, hipStream_t stream, Profiler* prof,
                   int slot) const {
    reserve_l2fused(batch);
    const Pass& pa = *passes_[0];
    const Pass& pb = *passes_[1];
    FusedArgs f;
    std::memset(&f, 0, sizeof(f));
    f.a.tw1 = pa.st->tw1.p; f.a.tw2 = pa.st->tw2.p; f.a.tw_lo = pa.tw_lo.p; f.a.tw_hi = pa.tw_hi.p; f.a.lo_bits = pa.lo_bits;
    f.a.n = n_; f.a.cn = pa.cn; f.a.s = pa.s; f.a.s_shift = (uint32_t)ilog2(pa.s); f.a.tiles = pa.cn / fused_.COLS_A; f.a.swap_in = inverse; f.a.scale = 1.0;
    f.b.tw1 = pb.st->tw1.p; f.b.tw2 = pb.st->tw2.p;
    f.b.n = n_; f.b.cn = pb.cn; f.b.s = pb.s; f.b.s_shift = (uint32_t)ilog2(pb.s); f.b.tiles = pb.cn / fused_.COLS_B; f.b.swap_out = inverse; f.b.scale = scale;
    f.window = fused_window_.p;
    f.ctrl = (uint32_t*)fused_ctrl_.p;
    f.depth = fused_depth_;
    f.tiles_a = (uint32_t)f.a.tiles; f.tiles_b = (uint32_t)f.b.tiles;
    f.spin_limit = 1u << 21;
    for (size_t b0 = 0; b0 < batch; b0 += FUSED_MAX_BATCH) {
      const size_t nb = std::min(FUSED_MAX_BATCH, batch - b0);
      f.in = in + b0 * n_; f.out = out + b0 * n_; f.batch = (uint32_t)nb;
      HIP_CHECK(hipMemsetAsync(fused_ctrl_.p, 0, fused_ctrl_words(nb) * sizeof(uint32_t), stream));
      const uint64_t items = (uint64_t)nb * (f.tiles_a + f.tiles_b);
      const unsigned grid = (unsigned)std::min<uint64_t>(fused_grid_, items);
      PROF_BEGIN(prof, slot);
      FOURIER_LAUNCH(fused_.fn, grid, fused_.NT, fused_.smem, stream, f);
      PROF_END(prof);
      // The kernel bounds its inter-workgroup waits (spin_limit) and raises ctrl[1] when one gives up; every workgroup
      // then returns early and part of the output is unwritten.  That must not read as success: the flag comes back
      // before the call returns (this plan option is therefore synchronous) and turns into FOURIER_HIP_RUNTIME_ERROR.
      fused_flag_.ensure(sizeof(uint32_t));
      HIP_CHECK(hipMemcpyAsync(fused_flag_.h, (const uint32_t*)fused_ctrl_.p + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      if (*(const volatile uint32_t*)fused_flag_.h != 0)
        throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "l2_fused: an inter-workgroup wait timed out; the output is incomplete");
    }
  }

  static int kk_L(const Pass& ps) { return ps.k.L; }
  // geometry of the first pass (chirp-in tables): length and columns
  int first_len() const { return passes_.empty() ? 0 : passes_.front()->k.L; }
  uint64_t first_cn() const { return passes_.empty() ? 0 : passes_.front()->cn; }

  // Bluestein fusion is available when the plan has separate first and last passes.
  bool can_fuse_bluestein() const { return !tiny_ && passes_.size() >= 2; }
  void enable_bluestein_fusion() {
    if (!can_fuse_bluestein()) return;
    Pass& f = *passes_.front();
    Pass& l = *passes_.back();
    f.k_blu = get_kernel<T>(f.k.L, MODE_FIRST, IO_BLU_IN);
    l.k_blu = get_kernel<T>(l.k.L, MODE_LAST, IO_BLU_OUT);
    f.has_blu = l.has_blu = true;
    for (Pass* p : {&f, &l}) set_smem_attribute(p->k_blu);
  }

  size_t size() const { return n_; }
  size_t num_passes() const { return tiny_ ? 1 : passes_.size(); }
  // a split last pass cannot run in place: two workgroups read the whole column tile and each writes half of its rows
  bool last_is_split() const { return !passes_.empty() && passes_.back()->k.split != 0; }
  size_t hbm_round_trips() const { return l2fused_enabled() ? 1 : num_passes(); }
  bool needs_scratch(bool in_place) const {
    if (l2fused_enabled()) return false;  // a transform is read completely before any of it is written
    return passes_.size() >= 3 || (passes_.size() == 2 && (in_place || last_is_split()));
  }
  std::string describe() const {
    if (tiny_ || n_ == 16 || (n_ == 32 && sizeof(T) == 4)) return "tiny(" + std::to_string(n_) + ")";
    if (!desc_override_.empty()) return desc_override_;
    std::string d;
    if (l2fused_enabled()) return std::to_string(fused_.L1) + "x" + std::to_string(fused_.L2) + " one-launch xcd-l2";
    for (size_t p = 0; p < passes_.size(); ++p)
      d += (p ? "x" : "") + std::to_string(passes_[p]->mode == MODE_ODD_LAST ? passes_[p]->odd_r : passes_[p]->k.L);
    return d;
  }

  // Transform `batch` contiguous transforms.  in == out is allowed; scratch must hold batch*n
  // elements when needs_scratch(in == out) (or when force_scratch is set).
  // Optional Bluestein fusion: io == IO_BLU_IN: `in` is the USER array (batch stride blu_n); io == IO_BLU_OUT:
  // `out` is the USER array.  The other side and the scratch are plan-sized (batch stride n).
  struct BluIO {
    int io = IO_PLAIN;
    const void* xtab = nullptr;
    uint64_t n = 0;
    int swap = 0;
    // chirp-in pass computing the chirp (PassArgs::blu_p ...); null = read xtab
    const void* p_tab = nullptr;
    const void* u_tab = nullptr;
    const void* tn_lo = nullptr;
    const void* tn_hi = nullptr;
    uint32_t tn_bits = 0;
  };

  void run(const cpx<T>* in, cpx<T>* out, cpx<T>* scratch, size_t batch, bool inverse, double scale, const cpx<T>* mul,
           bool force_scratch, hipStream_t stream, Profiler* prof = nullptr, int slot0 = 0, unsigned nxcd = 8,
           BluIO blu = BluIO()) const {
    if (batch == 0) return;
    // N = 16 (and f32 N = 32) also run one lane per transform; their ROWS pass only serves Bluestein M = 16 / 32
    const bool lane_per_transform = tiny_ || n_ == 16 || (n_ == 32 && sizeof(T) == 4);
    if (lane_per_transform && blu.io == IO_PLAIN) {
      TinyArgs a{in, out, (uint64_t)batch, (int)n_, inverse, inverse, scale};
      PROF_BEGIN(prof, slot0);
      void (*fn)(TinyArgs) = n_ == 32 ? &tiny_shfl_kernel<T, (sizeof(T) == 4 ? 32 : 16)>
                             : n_ == 16 ? &tiny_shfl_kernel<T, 16>
                             : n_ == 8 ? &tiny_shfl_kernel<T, 8>
                             : n_ == 4 ? &tiny_shfl_kernel<T, 4>
                             : n_ == 2 ? &tiny_shfl_kernel<T, 2> : &tiny_dft_kernel<T>;
      FOURIER_LAUNCH(fn, (batch + 255) / 256, 256, 0, stream, a);
      PROF_END(prof);
      apply_mul(out, batch, mul, inverse, scale, stream);
      return;
    }
    if (l2fused_enabled() && blu.io == IO_PLAIN && !mul) {
      run_l2fused(in, out, batch, inverse, scale, stream, prof, slot0);
      return;
    }
    const size_t np = passes_.size();
    const bool in_place = ((const void*)in == (const void*)out);
    // Every pass but the last is out of place (its tile footprints differ between input and output); the
    // last one (LAST / ODD_LAST / ROWS / TWOLEVEL) may run in place.  Ping-pong between `out` and the
    // scratch so that the final result lands in `out` and `in` is never written.
    const cpx<T>* src[8] = {in};
    cpx<T>* dst[8] = {out};
    if (np > 8) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "too many passes");
    if (np >= 2) {
      const bool start_scratch = in_place || blu.io == IO_BLU_OUT || (force_scratch && np == 2) || (np == 2 && last_is_split());
      for (size_t p = 0; p + 1 < np; ++p) {
        const bool to_scratch = start_scratch ? (p % 2 == 0) : (p % 2 == 1);
        dst[p] = to_scratch ? scratch : out;
        if (blu.io == IO_BLU_OUT && !to_scratch) dst[p] = (cpx<T>*)in;  // user-side output is shorter than n
        src[p + 1] = dst[p];
      }
      dst[np - 1] = out;
    }
    for (size_t p = 0; p < np; ++p)
      launch_pass(p, src[p], dst[p], batch, inverse, scale, stream, prof, slot0 + (int)p, nxcd, blu);
    apply_mul(out, batch, mul, inverse, scale, stream);
  }

  // Pointwise multiplier on the M-point spectrum of a forward, unscaled transform (bluesteins.rs:236-239): its own sweep.
  // Only the unfused Bluestein options take it (bluestein_fusion = 0, bluestein_conv = 0); the default plans multiply
  // inside fft_conv_kernel / the one-launch kernels.  Untimed by profile(): those options exist for A/B and tests.
  void apply_mul(cpx<T>* out, size_t batch, const cpx<T>* mul, bool inverse, double scale, hipStream_t stream) const {
    if (!mul) return;
    if (inverse || scale != 1.0) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "pointwise multiplier: forward unscaled only");
    BluArgs m{nullptr, out, mul, (uint64_t)n_, (uint64_t)n_, (uint64_t)batch, 0, 1.0};
    const size_t blocks = (batch * n_ + 255) / 256;
    FOURIER_LAUNCH(&blu_mul_kernel<T>, std::min<size_t>(std::max<size_t>(blocks, 1), 256 * 32), 256, 0, stream, m);
  }

  // One pass of the schedule.  inverse / scale / mul take effect on the passes they belong to (leading swap on
  // pass 0, trailing swap + scale + pointwise multiplier on the last pass).
  void launch_pass(size_t p, const cpx<T>* src, cpx<T>* dst, size_t batch, bool inverse, double scale,
                   hipStream_t stream, Profiler* prof, int slot, unsigned nxcd, BluIO blu = BluIO()) const {
    const size_t np = passes_.size();
    {
      const Pass& ps = *passes_[p];
      if (ps.mode == MODE_ODD_LAST) {
        OddArgs o;
        std::memset(&o, 0, sizeof(o));
        const bool final_pass = (p + 1 == np);
        o.in = src; o.out = dst;
        o.n = n_; o.s = ps.s; o.batch = batch;
        o.m = ps.cn; o.tw = ps.tw_lo.p;
        o.swap_out = final_pass && inverse; o.scale = final_pass ? scale : 1.0;
        for (int e = 0; e < ps.odd_r; ++e) unit_root((uint64_t)e, (uint64_t)ps.odd_r, o.wr[e], o.wi[e]);
        constexpr int VEC = 16 / (2 * (int)sizeof(T));
        const uint64_t threads = (uint64_t)batch * (ps.s / VEC) * ps.cn;
        const uint64_t grid = (threads + 255) / 256;
        if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
        PROF_BEGIN(prof, slot);
        FOURIER_LAUNCH(ps.odd_fn, grid, 256, 0, stream, o);
        PROF_END(prof);
        return;
      }
      PassArgs a;
      std::memset(&a, 0, sizeof(a));
      a.in = src; a.out = dst;
      a.tw1 = ps.st->tw1.p; a.tw2 = ps.st->tw2.p;
      if (ps.mode == MODE_TWOLEVEL) a.tw2 = ps.st2->tw1.p;
      a.tw_lo = ps.tw_lo.p; a.tw_hi = ps.tw_hi.p; a.tw_half = ps.tw_half.p;
      a.n = n_; a.cn = ps.cn; a.s = ps.s; a.s_shift = (uint32_t)ilog2(ps.s);
      a.lo_bits = ps.lo_bits;
      a.nxcd = nxcd & 0xff;
      a.xcd_interleave = (nxcd >> 8) & 3;
      const bool blu_here = ps.has_blu && ((blu.io == IO_BLU_IN && p == 0) || (blu.io == IO_BLU_OUT && p + 1 == np));
      if (blu_here) {
        a.blu_x = blu.xtab; a.blu_n = blu.n; a.blu_swap = blu.swap;
        if (blu.io == IO_BLU_IN && blu.p_tab) {
          a.blu_p = blu.p_tab; a.blu_u = blu.u_tab; a.tn_lo = blu.tn_lo; a.tn_hi = blu.tn_hi; a.tn_bits = blu.tn_bits;
          a.blu_cn_mod = (uint32_t)(ps.cn % blu.n);
          a.blu_cnq_mod = (uint32_t)((ps.cn * (uint64_t)(kk_L(ps) / 16)) % blu.n);
          a.blu_nd = (double)blu.n; a.blu_inv_nd = 1.0 / (double)blu.n;
        }
      }
      const KernelInfo& kk = blu_here ? ps.k_blu : ps.k;
      a.swap_in = (p == 0) && inverse;
      a.swap_out = (p + 1 == np) && inverse;
      a.scale = (p + 1 == np) ? scale : 1.0;
      uint64_t grid;
      if (ps.mode == MODE_TWOLEVEL) {
        a.total_cols = batch;
        a.tiles = 1;
        grid = batch;  // one workgroup per transform
      } else if (ps.mode == MODE_ROWS) {
        a.total_cols = batch;
        a.tiles = 1;
        grid = (batch + ps.k.COLS - 1) / ps.k.COLS;
      } else {
        a.tiles = ps.cn / kk.COLS;
        grid = (uint64_t)batch * a.tiles * (kk.split ? 2 : 1);
      }
      if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
      PROF_BEGIN(prof, slot);
      FOURIER_LAUNCH(kk.fn, grid, kk.NT, kk.smem, stream, a);
      PROF_END(prof);
    }
  }

  // Bluestein middle: this plan's LAST pass + (.) wtab + the FIRST pass of an inverse plan that starts with the
  // same length, in one launch (fft_conv_kernel).  src and dst are M-point work arrays, dst != src.
  bool can_conv() const { return !tiny_ && passes_.size() >= 2 && passes_.back()->mode == MODE_LAST; }
  void enable_conv() {
    if (!can_conv()) return;
    conv_ = get_conv_kernel<T>(passes_.back()->k.L);
    set_smem_attribute(conv_);
    auto it = stage_.find(conv_.L);  // the last pass may run on half tiles with half-length stage tables
    if (it == stage_.end()) {
      auto st = std::unique_ptr<StageTables<T>>(new StageTables<T>());
      make_stage_tables<T>(conv_.L, *st);
      it = stage_.emplace(conv_.L, std::move(st)).first;
    }
    conv_st_ = it->second.get();
  }
  bool palindromic() const {
    for (size_t p = 0; p < passes_.size(); ++p)
      if (passes_[p]->k.L != passes_[passes_.size() - 1 - p]->k.L) return false;
    return true;
  }
  void launch_conv(const cpx<T>* src, cpx<T>* dst, size_t batch, const void* wtab, hipStream_t stream, Profiler* prof, int slot,
                   unsigned nxcd) const {
    const Pass& first = *passes_.front();
    const Pass& last = *passes_.back();
    PassArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = src; a.out = dst;
    a.tw1 = conv_st_->tw1.p; a.tw2 = conv_st_->tw2.p;
    a.tw_lo = first.tw_lo.p; a.tw_hi = first.tw_hi.p; a.lo_bits = first.lo_bits;  // W_M^e, the table of any first pass
    a.mul = wtab;
    a.n = n_; a.cn = last.cn; a.s = last.s; a.s_shift = (uint32_t)ilog2(last.s);
    a.tiles = last.cn / conv_.COLS;
    a.nxcd = nxcd & 0xff;
    a.xcd_interleave = (nxcd >> 8) & 3;
    // this kernel (only) reads a per-transform table indexed like the data, the transformed chirp: let every XCD own an
    // eighth of the TILES of every transform, so that its 1/8 of the table (2 MiB of 16 at M = 2^21) stays in its L2
    // (with streaming stores: conv 4.5 vs 4.75 ms per 512 at C4; the plain passes lose 10-15 % under this order)
    static const bool sliced = dev_env("FOURIER_CONV_XCD_PLAIN") == nullptr;  // development switch, read once
    if (sliced && a.nxcd == 8 && a.xcd_interleave == 0 && a.tiles % 8 == 0) a.xcd_interleave = 2;
    a.scale = 1.0;
    const uint64_t grid = (uint64_t)batch * a.tiles;
    if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
    PROF_BEGIN(prof, slot);
    FOURIER_LAUNCH(conv_.fn, grid, conv_.NT, conv_.smem, stream, a);
    PROF_END(prof);
  }

 private:
  size_t n_;
  bool tiny_ = false;
  int tl1_ = 0, tl2_ = 0;   // pass lengths of a one-launch (MODE_TWOLEVEL) plan
  KernelInfo blu_small_, conv_;
  StageTables<T>* conv_st_ = nullptr;
  FusedInfo fused_;
  bool fused_on_ = false;
  unsigned fused_grid_ = 0, fused_depth_ = 2;
  mutable DevBuf fused_window_, fused_ctrl_;
  mutable PinnedBuf fused_flag_;  // ctrl[1] (abort flag) of the last fused launch, read back before the call returns
  std::string desc_override_;
  std::vector<std::unique_ptr<Pass>> passes_;
  std::map<int, std::unique_ptr<StageTables<T>>> stage_;
};

// ---------------------------------------------------------------------------------------------
// small mixed-radix sizes in LDS: 2^a * 3^b (b > 0) on the reference's own schedule and tables; lengths with factors 5..13 on the same pass
template <typename T> class MixedEngine {
 public:
  // one LDS buffer of one transform must fit a workgroup (the passes run in place): N * sizeof(complex) <= 160 KiB, all of a
  // gfx950 CU's LDS (round 3; 144 KiB before: f64 N = 10000 is 156.25 KiB)
  static constexpr size_t MAX_LDS = 160 * 1024;
  static constexpr size_t MAX_N = MAX_LDS / sizeof(cpx<T>);  // 20480 (f32), 10240 (f64)
  // autosort/mod.rs:104-116: one radix-4 first when divisible, then greedily 8, 4, 3, 2 -- and, beyond the reference (which
  // sends such lengths to Bluestein, fourier/src/lib.rs:38-42), the same pass with prime radices 5, 7, 11, 13
  static bool factor(size_t size, std::vector<uint32_t>& radices) {
    radices.clear();
    if (size == 0 || size > MAX_N) return false;
    for (size_t cur = size; cur > 1;) {
      const uint32_t r = mix_next_radix((uint32_t)size, (uint32_t)cur, cur == size);  // the kernels' own schedule
      if (cur % r || radices.size() == sizeof(MixArgs{}.radix)) return false;
      radices.push_back(r);
      cur /= r;
    }
    return true;
  }
  struct Kernel { void (*fn)(MixArgs); uint32_t group; size_t nbuf; uint32_t threads; };
  // the per-length kernel where one is instantiated (every 2^a*3^b up to MAX_N, and the common lengths with factors
  // 5 / 7), else the runtime-parameterised kernel
  static Kernel pick_kernel(size_t n) {
    // runtime-parameterised: about 1024 points per workgroup up to 1024 points, then one transform per workgroup -- 256 threads
    // x 4 / 8 points up to 2048 points, 512 x 8 up to 4096, 1024 x 8 up to 8192 (1024 x 4 for 2049..4096 measured slower than
    // 256 x 16: 3125 f32 19 % against 24 %, r03_s22)
    Kernel k{nullptr, (uint32_t)std::max<size_t>(1, 1024 / n), 1, 256};
    const int maxp = (n % 11 == 0 || n % 13 == 0) ? 13 : ((n % 5 == 0 || n % 7 == 0) ? 7 : 3);
    const size_t pts = k.group * n;
#define FOURIER_MIX_RT(P, NT) (maxp == 13 ? &mixed_radix_kernel<T, 13, P, NT> : (maxp == 7 ? &mixed_radix_kernel<T, 7, P, NT> : &mixed_radix_kernel<T, 3, P, NT>))
    if (pts <= 1024 && mix_threads<T>((uint32_t)n) == 128) { k.fn = FOURIER_MIX_RT(8, 128); k.threads = 128; }  // few work items per pass
    else if (pts <= 1024) k.fn = FOURIER_MIX_RT(4, 256);
    else if (pts <= 2048) k.fn = FOURIER_MIX_RT(8, 256);
    else if (sizeof(T) == 8 && maxp == 13) {}  // f64 with a radix-13 butterfly does not fit 128 registers (spills; 4095 f64: 20 % against Bluestein's 25 %)
    else if (pts <= 4096) { k.fn = FOURIER_MIX_RT(8, 512); k.threads = 512; }
    else if (pts <= 8192) { k.fn = FOURIER_MIX_RT(8, 1024); k.threads = 1024; }
#undef FOURIER_MIX_RT
    if (dev_env("FOURIER_MIX_GENERIC") && k.fn) return k;
#define FOURIER_MIX_CT(NN)                                                      \
  case NN:                                                                      \
    if constexpr ((size_t)NN <= MAX_N) k = Kernel{&mixed_radix_kernel_ct<T, NN>, mix_group<T>(NN), mix_inplace<T>(NN) ? (size_t)1 : (size_t)2, mix_threads<T>(NN)}; \
    break;
    switch (n) {  // every 2^a * 3^b (b >= 1) the engine runs in LDS: 3 ... 19683 (f32) / 9216 (f64)
      FOURIER_MIX_CT(3) FOURIER_MIX_CT(6) FOURIER_MIX_CT(9) FOURIER_MIX_CT(12) FOURIER_MIX_CT(18) FOURIER_MIX_CT(24)
      FOURIER_MIX_CT(27) FOURIER_MIX_CT(36) FOURIER_MIX_CT(48) FOURIER_MIX_CT(54) FOURIER_MIX_CT(72) FOURIER_MIX_CT(81)
      FOURIER_MIX_CT(96) FOURIER_MIX_CT(108) FOURIER_MIX_CT(144) FOURIER_MIX_CT(162) FOURIER_MIX_CT(192) FOURIER_MIX_CT(216)
      FOURIER_MIX_CT(243) FOURIER_MIX_CT(288) FOURIER_MIX_CT(324) FOURIER_MIX_CT(384) FOURIER_MIX_CT(432) FOURIER_MIX_CT(486)
      FOURIER_MIX_CT(576) FOURIER_MIX_CT(648) FOURIER_MIX_CT(729) FOURIER_MIX_CT(768) FOURIER_MIX_CT(864) FOURIER_MIX_CT(972)
      FOURIER_MIX_CT(1152) FOURIER_MIX_CT(1296) FOURIER_MIX_CT(1458) FOURIER_MIX_CT(1536) FOURIER_MIX_CT(1728) FOURIER_MIX_CT(1944)
      FOURIER_MIX_CT(2187) FOURIER_MIX_CT(2304) FOURIER_MIX_CT(2592) FOURIER_MIX_CT(2916) FOURIER_MIX_CT(3072) FOURIER_MIX_CT(3456)
      FOURIER_MIX_CT(3888) FOURIER_MIX_CT(4374) FOURIER_MIX_CT(4608) FOURIER_MIX_CT(5184) FOURIER_MIX_CT(5832) FOURIER_MIX_CT(6144)
      FOURIER_MIX_CT(6561) FOURIER_MIX_CT(6912) FOURIER_MIX_CT(7776) FOURIER_MIX_CT(8748) FOURIER_MIX_CT(9216) FOURIER_MIX_CT(10368)
      FOURIER_MIX_CT(11664) FOURIER_MIX_CT(13122) FOURIER_MIX_CT(13824) FOURIER_MIX_CT(15552) FOURIER_MIX_CT(17496)
      FOURIER_MIX_CT(18432) FOURIER_MIX_CT(19683)
      // beyond the reference: every 2^a * 3^b * 5^c (c >= 1) up to MAX_N -- among them the reference's own benchmark lengths
      // 5^3 .. 5^5 (fft_bench.rs:156) -- the powers of 7 and a selection of lengths with a factor 7; other lengths with factors 7, 11, 13 take the runtime kernel
      FOURIER_MIX_CT(10) FOURIER_MIX_CT(25) FOURIER_MIX_CT(100) FOURIER_MIX_CT(125) FOURIER_MIX_CT(625) FOURIER_MIX_CT(1000)
      FOURIER_MIX_CT(3125) FOURIER_MIX_CT(5000) FOURIER_MIX_CT(8000) FOURIER_MIX_CT(10000) FOURIER_MIX_CT(15625) FOURIER_MIX_CT(49) FOURIER_MIX_CT(343) FOURIER_MIX_CT(16807)
#ifndef FOURIER_EMU  // the CPU emulation build keeps the subset above (compile time); its other lengths run the runtime kernel
      FOURIER_MIX_CT(5) FOURIER_MIX_CT(15) FOURIER_MIX_CT(20) FOURIER_MIX_CT(30) FOURIER_MIX_CT(40) FOURIER_MIX_CT(45)
      FOURIER_MIX_CT(50) FOURIER_MIX_CT(60) FOURIER_MIX_CT(75) FOURIER_MIX_CT(80) FOURIER_MIX_CT(90) FOURIER_MIX_CT(120)
      FOURIER_MIX_CT(135) FOURIER_MIX_CT(150) FOURIER_MIX_CT(160) FOURIER_MIX_CT(180) FOURIER_MIX_CT(200) FOURIER_MIX_CT(225)
      FOURIER_MIX_CT(240) FOURIER_MIX_CT(250) FOURIER_MIX_CT(270) FOURIER_MIX_CT(300) FOURIER_MIX_CT(320) FOURIER_MIX_CT(360)
      FOURIER_MIX_CT(375) FOURIER_MIX_CT(400) FOURIER_MIX_CT(405) FOURIER_MIX_CT(450) FOURIER_MIX_CT(480) FOURIER_MIX_CT(500)
      FOURIER_MIX_CT(540) FOURIER_MIX_CT(600) FOURIER_MIX_CT(640) FOURIER_MIX_CT(675) FOURIER_MIX_CT(720) FOURIER_MIX_CT(750)
      FOURIER_MIX_CT(800) FOURIER_MIX_CT(810) FOURIER_MIX_CT(900) FOURIER_MIX_CT(960) FOURIER_MIX_CT(1080) FOURIER_MIX_CT(1125)
      FOURIER_MIX_CT(1200) FOURIER_MIX_CT(1215) FOURIER_MIX_CT(1250) FOURIER_MIX_CT(1280) FOURIER_MIX_CT(1350)
      FOURIER_MIX_CT(1440) FOURIER_MIX_CT(1500) FOURIER_MIX_CT(1600) FOURIER_MIX_CT(1620) FOURIER_MIX_CT(1800)
      FOURIER_MIX_CT(1875) FOURIER_MIX_CT(1920) FOURIER_MIX_CT(2000) FOURIER_MIX_CT(2025) FOURIER_MIX_CT(2160)
      FOURIER_MIX_CT(2250) FOURIER_MIX_CT(2400) FOURIER_MIX_CT(2430) FOURIER_MIX_CT(2500) FOURIER_MIX_CT(2560)
      FOURIER_MIX_CT(2700) FOURIER_MIX_CT(2880) FOURIER_MIX_CT(3000) FOURIER_MIX_CT(3200) FOURIER_MIX_CT(3240)
      FOURIER_MIX_CT(3375) FOURIER_MIX_CT(3600) FOURIER_MIX_CT(3645) FOURIER_MIX_CT(3750) FOURIER_MIX_CT(3840)
      FOURIER_MIX_CT(4000) FOURIER_MIX_CT(4050) FOURIER_MIX_CT(4320) FOURIER_MIX_CT(4500) FOURIER_MIX_CT(4800)
      FOURIER_MIX_CT(4860) FOURIER_MIX_CT(5120) FOURIER_MIX_CT(5400) FOURIER_MIX_CT(5625) FOURIER_MIX_CT(5760)
      FOURIER_MIX_CT(6000) FOURIER_MIX_CT(6075) FOURIER_MIX_CT(6250) FOURIER_MIX_CT(6400) FOURIER_MIX_CT(6480)
      FOURIER_MIX_CT(6750) FOURIER_MIX_CT(7200) FOURIER_MIX_CT(7290) FOURIER_MIX_CT(7500) FOURIER_MIX_CT(7680)
      FOURIER_MIX_CT(8100) FOURIER_MIX_CT(8640) FOURIER_MIX_CT(9000) FOURIER_MIX_CT(9375) FOURIER_MIX_CT(9600)
      FOURIER_MIX_CT(9720) FOURIER_MIX_CT(10125) FOURIER_MIX_CT(10240) FOURIER_MIX_CT(10800) FOURIER_MIX_CT(10935)
      FOURIER_MIX_CT(11250) FOURIER_MIX_CT(11520) FOURIER_MIX_CT(12000) FOURIER_MIX_CT(12150) FOURIER_MIX_CT(12500)
      FOURIER_MIX_CT(12800) FOURIER_MIX_CT(12960) FOURIER_MIX_CT(13500) FOURIER_MIX_CT(14400) FOURIER_MIX_CT(14580)
      FOURIER_MIX_CT(15000) FOURIER_MIX_CT(15360) FOURIER_MIX_CT(16000) FOURIER_MIX_CT(16200) FOURIER_MIX_CT(16875)
      FOURIER_MIX_CT(17280) FOURIER_MIX_CT(18000) FOURIER_MIX_CT(18225) FOURIER_MIX_CT(18750) FOURIER_MIX_CT(19200)
      FOURIER_MIX_CT(19440) FOURIER_MIX_CT(20000) FOURIER_MIX_CT(20250) FOURIER_MIX_CT(20480) FOURIER_MIX_CT(2401)
      // a selection with a factor 7: 7 * 2^k, the highly composite 840 / 1260 / 1680 / 2520 / 5040 / 10080 and their kin
      FOURIER_MIX_CT(14) FOURIER_MIX_CT(21) FOURIER_MIX_CT(28) FOURIER_MIX_CT(35) FOURIER_MIX_CT(42) FOURIER_MIX_CT(56)
      FOURIER_MIX_CT(63) FOURIER_MIX_CT(70) FOURIER_MIX_CT(84) FOURIER_MIX_CT(105) FOURIER_MIX_CT(112) FOURIER_MIX_CT(126)
      FOURIER_MIX_CT(140) FOURIER_MIX_CT(168) FOURIER_MIX_CT(210) FOURIER_MIX_CT(224) FOURIER_MIX_CT(252) FOURIER_MIX_CT(280)
      FOURIER_MIX_CT(315) FOURIER_MIX_CT(336) FOURIER_MIX_CT(420) FOURIER_MIX_CT(448) FOURIER_MIX_CT(504) FOURIER_MIX_CT(560)
      FOURIER_MIX_CT(630) FOURIER_MIX_CT(672) FOURIER_MIX_CT(840) FOURIER_MIX_CT(896) FOURIER_MIX_CT(1008) FOURIER_MIX_CT(1120)
      FOURIER_MIX_CT(1260) FOURIER_MIX_CT(1344) FOURIER_MIX_CT(1680) FOURIER_MIX_CT(1792) FOURIER_MIX_CT(2016)
      FOURIER_MIX_CT(2240) FOURIER_MIX_CT(2520) FOURIER_MIX_CT(2688) FOURIER_MIX_CT(3360) FOURIER_MIX_CT(3584)
      FOURIER_MIX_CT(4480) FOURIER_MIX_CT(5040) FOURIER_MIX_CT(5376) FOURIER_MIX_CT(6720) FOURIER_MIX_CT(7168)
      FOURIER_MIX_CT(8960) FOURIER_MIX_CT(10080) FOURIER_MIX_CT(14336) FOURIER_MIX_CT(17920)
#endif
      default: break;
    }
#undef FOURIER_MIX_CT
    return k;
  }
  static bool handles(size_t n) {
    std::vector<uint32_t> c;
    const char* cap = dev_env("FOURIER_MIX_MAX_N");  // development switch: A/B against the Bluestein / odd-pass routes
    if (n > (cap ? std::min<size_t>(MAX_N, (size_t)atoll(cap)) : MAX_N) || is_pow2(n) || !factor(n, c)) return false;
    if (dev_env("FOURIER_MIX_REFERENCE_RADICES") && mix_extended((uint32_t)n)) return false;  // A/B against Bluestein
    return pick_kernel(n).fn != nullptr;  // no per-length kernel and beyond the runtime kernel's 8192 points: Bluestein
  }
  // twiddle.rs:7-19 verbatim: theta = (index*2) as f64 * PI / size as f64; (cos, -sin) cast to T.
  // cos and sin stay two separate libm calls, as in Rust (a merged sincos() differs in the last bit).
  __attribute__((noinline)) static double libm_cos(double t) { return std::cos(t); }
  __attribute__((noinline)) static double libm_sin(double t) { return std::sin(t); }
  static cpx<T> ref_twiddle(size_t index, size_t size) {
    const double theta = (double)(index * 2) * M_PI / (double)size;
    return {(T)libm_cos(theta), (T)(-libm_sin(theta))};
  }

  explicit MixedEngine(size_t n) : n_(n) {
    factor(n, radices_);
    std::vector<cpx<T>> tw;
    size_t cur = n;
    for (const size_t R : radices_) {  // mod.rs:24-46
      const size_t m = cur / R;
      for (size_t i = 0; i < m; ++i) {
        tw.push_back({(T)1, (T)0});
        for (size_t j = 1; j < R; ++j) tw.push_back(ref_twiddle(i * j, cur));
      }
      cur /= R;
    }
    if (tw.empty()) tw.push_back({(T)1, (T)0});
    tw_.upload(tw);
    // transforms per workgroup: about 1024 points (16 KiB of LDS in f32: several workgroups per CU; larger groups that
    // fill the 256 threads better lose more in occupancy than they gain, r01 session 9)
    const Kernel k = pick_kernel(n);
    fn_ = k.fn; group_ = k.group; nbuf_ = k.nbuf; threads_ = k.threads;
    smem_ = nbuf_ * (size_t)group_ * n * sizeof(cpx<T>);
    if (smem_ > MAX_LDS) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "mixed-radix length needs the per-length kernel");
    raise_smem_limit((const void*)fn_, smem_);
  }
  std::string describe() const {
    std::string d;
    for (const uint32_t r : radices_) d += (d.empty() ? "" : ".") + std::to_string(r);
    return d;
  }
  void run(const cpx<T>* in, cpx<T>* out, size_t batch, bool forward, bool scaled, double scale, hipStream_t stream,
           Profiler* prof) const {
    if (batch == 0) return;
    MixArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.tw = tw_.p; a.batch = batch; a.n = (uint32_t)n_; a.group = group_;
    a.npass = (uint32_t)radices_.size();
    for (size_t r = 0; r < radices_.size(); ++r) a.radix[r] = (uint8_t)radices_[r];
    a.forward = forward; a.scaled = scaled; a.scale = scale;
    const cpx<T> w3 = ref_twiddle(1, 3), w8 = ref_twiddle(1, 8);  // butterfly.rs:12,50
    a.w3re = w3.re; a.w3im = w3.im; a.w8re = w8.re; a.w8im = w8.im;
    const uint64_t grid = (batch + group_ - 1) / group_;
    if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
    PROF_BEGIN(prof, 0);
    FOURIER_LAUNCH(fn_, grid, threads_, smem_, stream, a);
    PROF_END(prof);
  }

 private:
  size_t n_;
  std::vector<uint32_t> radices_;
  uint32_t threads_ = 256;
  void (*fn_)(MixArgs) = nullptr;
  size_t nbuf_ = 1;  // LDS buffers of `group_` transforms: 1 = in-place passes (2 = ping-pong, FOURIER_MIX_INPLACE_BYTES builds)
  uint32_t group_ = 1;
  size_t smem_ = 0;
  DevBuf tw_;
};

// ---------------------------------------------------------------------------------------------
// 2^a * 3^b with a < 12 beyond the LDS kernels' reach (3^10, 2^8*3^5, ...): the reference's Stockham autosort pass by pass in
// global memory (autosort/mod.rs:203-284), radices 27 / 9 / 3 first, then 16 / 8 / 4 / 2; one HBM round trip per pass
// instead of Bluestein's five over a padded power of two.  Intermediates ping-pong between the two halves of the
// plan's scratch, the last pass writes the output (in place allowed).
template <typename T> class GenericEngine {
 public:
  static constexpr size_t MAX_N = (size_t)1 << 26;
  static bool handles(size_t n) {
    if (n < 2 || n > MAX_N || dev_env("FOURIER_NO_GENERIC_MIXED")) return false;
    size_t p = n;
    while (p % 3 == 0) p /= 3;
    return is_pow2(p) && p < 4096 && p != n;  // b >= 1, a < 12 (a >= 12 runs as tiled passes + odd passes)
  }
  explicit GenericEngine(size_t n) : n_(n) {
    size_t p3 = 1, p2 = n;
    while (p2 % 3 == 0) { p2 /= 3; p3 *= 3; }
    std::vector<int> radices;
    while (p3 > 1) { const int r = p3 % 27 == 0 ? 27 : (p3 % 9 == 0 ? 9 : 3); radices.push_back(r); p3 /= (size_t)r; }
    while (p2 > 1) { const int r = p2 % 16 == 0 ? 16 : (p2 % 8 == 0 ? 8 : (p2 % 4 == 0 ? 4 : 2)); radices.push_back(r); p2 /= (size_t)r; }
    size_t s = 1, size = n;
    for (int r : radices) {
      Pass ps;
      ps.r = r; ps.s = (uint32_t)s; ps.m = (uint32_t)(size / (size_t)r);
      ps.tw.reset(new DevBuf());
      if (ps.m > 1) {  // W_size^{e}, e < size (f64 trig, cast: twiddle.rs:7-19)
        std::vector<cpx<T>> tw(size);
        for (size_t e = 0; e < size; ++e) { double re, im; unit_root(e, size, re, im); tw[e] = {(T)re, (T)im}; }
        ps.tw->upload(tw);
      }
      switch (r) {
        case 2: ps.fn = &stockham_pass_kernel<T, 2>; break;
        case 3: ps.fn = &stockham_pass_kernel<T, 3>; break;
        case 4: ps.fn = &stockham_pass_kernel<T, 4>; break;
        case 8: ps.fn = &stockham_pass_kernel<T, 8>; break;
        case 9: ps.fn = &stockham_pass_kernel<T, 9>; break;
        case 16: ps.fn = &stockham_pass_kernel<T, 16>; break;
        default: ps.fn = &stockham_pass_kernel<T, 27>; break;
      }
      ps.smem = s == 1 ? (size_t)r * 256 * sizeof(cpx<T>) : 0;  // first pass: the workgroup's outputs are staged in LDS
      raise_smem_limit((const void*)ps.fn, ps.smem);
      passes_.push_back(std::move(ps));
      s *= (size_t)r; size /= (size_t)r;
    }
  }
  size_t num_passes() const { return passes_.size(); }
  std::string describe() const {
    std::string d;
    for (const Pass& p : passes_) d += (d.empty() ? "" : ".") + std::to_string(p.r);
    return d;
  }
  // scratch: 2 * batch * n elements (two halves), unused when there is a single pass
  void run(const cpx<T>* in, cpx<T>* out, cpx<T>* scratch, size_t batch, bool inverse, double scale, hipStream_t stream, Profiler* prof) const {
    if (batch == 0) return;
    const size_t np = passes_.size();
    cpx<T>* half[2] = {scratch, scratch + batch * n_};
    const cpx<T>* src = in;
    for (size_t p = 0; p < np; ++p) {
      const Pass& ps = passes_[p];
      cpx<T>* dst = (p + 1 == np) ? out : half[p & 1];
      GenArgs a;
      std::memset(&a, 0, sizeof(a));
      a.in = src; a.out = dst; a.tw = ps.m > 1 ? ps.tw->p : nullptr;
      a.n = n_; a.s = ps.s; a.m = ps.m;
      const uint64_t per = (uint64_t)ps.s * ps.m;
      a.blocks_per = (uint32_t)((per + 255) / 256);
      a.swap_in = (p == 0) && inverse; a.swap_out = (p + 1 == np) && inverse; a.final_pass = (p + 1 == np);
      a.scale = (p + 1 == np) ? scale : 1.0;
      for (int e = 0; e < ps.r && e < 27; ++e) unit_root((uint64_t)e, (uint64_t)ps.r, a.wr[e], a.wi[e]);
      const uint64_t grid = (uint64_t)a.blocks_per * batch;
      if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
      PROF_BEGIN(prof, (int)p);
      FOURIER_LAUNCH(ps.fn, grid, 256, ps.smem, stream, a);
      PROF_END(prof);
      src = dst;
    }
  }

 private:
  struct Pass { int r = 0; uint32_t s = 0, m = 0; std::unique_ptr<DevBuf> tw; void (*fn)(GenArgs) = nullptr; size_t smem = 0; };
  size_t n_;
  std::vector<Pass> passes_;
};

// ---------------------------------------------------------------------------------------------
// host f64 radix-2 FFT, used only at plan time for the Bluestein w table (bluesteins.rs:46-47)
static void host_fft(std::vector<double>& re, std::vector<double>& im) {
  const size_t m = re.size();
  for (size_t i = 1, j = 0; i < m; ++i) {
    size_t bit = m >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
  }
  std::vector<double> wr(m / 2 ? m / 2 : 1), wi(m / 2 ? m / 2 : 1);
  for (size_t k = 0; k < m / 2; ++k) unit_root(k, m, wr[k], wi[k]);
  for (size_t len = 2; len <= m; len <<= 1) {
    const size_t half = len / 2, step = m / len;
    for (size_t i = 0; i < m; i += len)
      for (size_t k = 0; k < half; ++k) {
        const double ur = wr[k * step], ui = wi[k * step];
        const double xr = re[i + k + half] * ur - im[i + k + half] * ui;
        const double xi = re[i + k + half] * ui + im[i + k + half] * ur;
        re[i + k + half] = re[i + k] - xr; im[i + k + half] = im[i + k] - xi;
        re[i + k] += xr; im[i + k] += xi;
      }
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T> class Plan {
 public:
  static constexpr size_t ELEM = sizeof(cpx<T>);

  Plan(size_t n, int device) : n_(n) {
    if (n == 0) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "size 0 is invalid");
    int count = 0;
    HIP_CHECK(hipGetDeviceCount(&count));
    if (count <= 0) throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "no HIP device");
    if (device < 0) HIP_CHECK(hipGetDevice(&device));
    if (device >= count) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "bad device index");
    device_ = device;
    DeviceGuard g(device_);
    if (is_pow2(n)) {
      eng_.reset(new Pow2Engine<T>(n, false, true));
    } else if (Pow2Engine<T>::handles_mixed(n)) {
      // big-radix passes over the 2^a part (a >= 12), then a radix-3^b pass: three HBM round trips at full tile
      // efficiency beat the one-workgroup-per-CU LDS kernel where both apply (3*2^12 f32: 23 % vs 14 %)
      eng_.reset(new Pow2Engine<T>(n));
    } else if (MixedEngine<T>::handles(n) && try_mixed(n)) {
    } else if (GenericEngine<T>::handles(n)) {
      gen_.reset(new GenericEngine<T>(n));
    } else {
      init_bluestein();
    }
    refresh_desc();
  }
  // the longest LDS plans ask for the whole 160 KiB of a CU: where the runtime refuses, the next route takes the length
  bool try_mixed(size_t n) {
    try { mix_.reset(new MixedEngine<T>(n)); return true; }
    catch (const EngineError& e) {
      if (e.status == ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY) throw;
      (void)hipGetLastError();
      mix_.reset();
      return false;
    }
  }
  void refresh_desc() {
    if (mix_) desc_ = "stockham mixed-radix " + mix_->describe();
    else if (gen_) desc_ = "stockham global-pass " + gen_->describe();
    else if (blu_) desc_ = "bluestein M=" + std::to_string(m_) + " inner " + eng_->describe() + (small_fused_ ? " fused" : "");
    else desc_ = "stockham " + eng_->describe();
    desc_ += sizeof(T) == 4 ? " f32" : " f64";
  }

  ~Plan() {
    if (legacy_stream_) {
      DeviceGuard g(device_);
      (void)hipStreamDestroy(legacy_stream_);
    }
  }
  Plan(const Plan&) = delete;
  Plan& operator=(const Plan&) = delete;

  size_t size() const { return n_; }
  int device() const { return device_; }
  const char* describe() const { return desc_.c_str(); }
  int last_status() const { return status_; }
  void set_status(int s) const { status_ = s; }

  // kernel "slots" in launch order, as reported by profile(): names for bench.py / rocprof matching
  std::string slot_names() const {
    std::string d;
    if (mix_) return "mixed_radix";
    if (gen_) { for (size_t p = 0; p < gen_->num_passes(); ++p) d += std::string(d.empty() ? "" : ",") + "pass" + std::to_string(p); return d; }
    auto passes = [&](const char* tag) {
      for (size_t p = 0; p < (blu_ ? eng_->num_passes() : eng_->hbm_round_trips()); ++p) d += std::string(d.empty() ? "" : ",") + tag + std::to_string(p);
    };
    if (!blu_) { passes("pass"); return d; }
    if (small_fused_) return "bluestein_one_launch";
    d = "blu_pre"; passes("fwd_pass"); passes("inv_pass"); d += ",blu_post";  // blu_pre/post stay empty when fused
    if (fused_ && conv_) {  // the last forward pass and the first inverse pass are one launch (inv_pass0 stays empty)
      const std::string from = "fwd_pass" + std::to_string(eng_->num_passes() - 1);
      d.replace(d.find(from), from.size(), "conv_pass");
    }
    return d;
  }

  double model_bytes() const {
    if (mix_) return 2.0 * n_ * ELEM;
    if (gen_) return 2.0 * n_ * ELEM * gen_->num_passes();
    if (!blu_) return 2.0 * n_ * ELEM * eng_->hbm_round_trips();
    // unfused: pre (n + table read, m write) + 2 inner FFTs + w table + post (n + table read, n write);
    // fused: the first / last inner pass read / write the n-point user array instead of an m-point sweep
    if (small_fused_) return (double)ELEM * 2.0 * n_;  // tables stay L2-resident
    const double chirp_reads = (chirp_compute_ ? 1.0 : 2.0) * n_;  // the n-entry chirp table: the chirp-out pass reads it, the chirp-in pass only without bluestein_chirp_compute
    if (fused_ && conv_) return (double)ELEM * (2.0 * m_ * (2.0 * eng_->num_passes() - 1.0) - 2.0 * (m_ - n_) + m_ + chirp_reads);
    if (fused_) return (double)ELEM * (2.0 * 2.0 * m_ * eng_->num_passes() - 2.0 * (m_ - n_) + m_ + chirp_reads);
    return (double)ELEM * ((2.0 * n_ + m_) + 2.0 * 2.0 * m_ * eng_->num_passes() + m_ + 3.0 * n_);
  }

  int set_option(const std::string& key, long long v) {
    DeviceGuard g(device_);  // bluestein_fusion may allocate tables: they must land on the plan's device
    if (key == "chunk_bytes" && v >= 0) { chunk_bytes_ = (size_t)v; return 0; }
    if (key == "scratch" && (v == 0 || v == 1)) { force_scratch_ = (v == 1); return 0; }
    if (key == "xcd_swizzle" && v >= 0 && v <= 4) { nxcd_ = v == 0 ? 1 : (8 | ((unsigned)(v - 1) << 8)); return 0; }
    if (key == "bluestein_fusion" && (v == 0 || v == 1)) {
      fused_ = (v == 1) && blu_ && eng_->can_fuse_bluestein();
      small_fused_ = (v == 1) && blu_ && eng_->enable_bluestein_small();
      return 0;
    }
    if (key == "bluestein_conv" && (v == 0 || v == 1)) { conv_ = (v == 1) && conv_ok_; return 0; }
    if (key == "bluestein_chirp_compute" && (v == 0 || v == 1)) { chirp_compute_ = (v == 1) && chirp_p_.p != nullptr; return 0; }
    if (key == "host_chunk_bytes" && v > 0) { host_chunk_bytes_ = (size_t)v; return 0; }
    // both passes in one launch with the intermediate in the XCD's L2 (2^16..2^18 f32, 2^15..2^17 f64); 0 where unavailable
    if (key == "l2_fused" && (v == 0 || v == 1)) {
      if (blu_ || !eng_ || (v == 1 && !eng_->has_l2fused())) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      eng_->set_l2fused(v == 1);
      refresh_desc();
      return 0;
    }
    if (key == "l2_fused_depth" && !blu_ && eng_ && eng_->set_l2fused_depth((unsigned)v)) return 0;
    if (key == "l2_fused_grid" && !blu_ && eng_ && eng_->has_l2fused() && v > 0) { eng_->set_l2fused_grid((unsigned)v); return 0; }
    return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
  }

  // Chunk size for a call of `batch` transforms and the plan-owned device buffers it needs (scratch of the in-place /
  // three-pass plans, the Bluestein work array).  exec() calls this on every call -- it allocates only when the batch
  // is larger than anything seen before -- and fourier_hip_reserve_* calls it ahead of time, so that a later
  // transform_batch of at most that batch never allocates (hipMalloc / hipFree synchronise the device) and can be
  // captured into a HIP graph.  Returns the number of transforms per chunk.
  size_t prepare(size_t batch, bool in_place) const {
    if (mix_ || batch == 0) return batch;
    if (gen_) {  // two scratch halves of one chunk each; chunked so that a launch stays below 2^31 workgroups
      size_t chunk = batch;
      if (chunk_bytes_) chunk = std::max<size_t>(1, std::min<size_t>(batch, chunk_bytes_ / (n_ * ELEM)));
      while (chunk > 1 && (double)chunk * (double)n_ / 256.0 > 2.0e9) chunk = (chunk + 1) / 2;
      for (;;) {
        try { scratch_.ensure(2 * chunk * n_ * ELEM); return chunk; }
        catch (const EngineError& e) {
          if (e.status != ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY || chunk <= 1) throw;
          (void)hipGetLastError();
          chunk = (chunk + 1) / 2;
        }
      }
    }
    const size_t per = (blu_ ? m_ : n_) * ELEM;
    size_t chunk = batch;
    if (chunk_bytes_) chunk = std::max<size_t>(1, std::min<size_t>(batch, chunk_bytes_ / per));
    // keep every launch's grid below 2^31 blocks
    while (chunk > 1 && (double)chunk * (double)(blu_ ? m_ : n_) / 16.0 > 2.0e9) chunk = (chunk + 1) / 2;
    // The plan's scratch (and the Bluestein work array) hold one chunk.  If the device cannot give that much -- an
    // in-place call on a batch that fills most of the HBM -- fall back to smaller chunks instead of failing: chunks
    // run back to back on the stream and the results are the same.
    auto reserve = [&](auto&& alloc) {
      for (;;) {
        try { alloc(chunk); return; }
        catch (const EngineError& e) {
          if (e.status != ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY || chunk <= 1) throw;
          (void)hipGetLastError();  // the allocation failure is handled here
          chunk = (chunk + 1) / 2;
        }
      }
    };
    if (!blu_) {
      if (eng_->l2fused_enabled()) { eng_->reserve_l2fused(chunk); return chunk; }
      const bool need = eng_->needs_scratch(in_place) || (force_scratch_ && eng_->num_passes() >= 2);
      if (need) reserve([&](size_t c) { scratch_.ensure(c * n_ * ELEM); });
      return chunk;
    }
    if (small_fused_) return batch;  // whole chirp-z in one launch: no work array
    reserve([&](size_t c) {
      work_.ensure(c * m_ * ELEM);
      if (eng_->needs_scratch(true) || fused_) scratch_.ensure(c * m_ * ELEM);
    });
    return chunk;
  }
  void reserve_for(size_t batch, bool in_place) const {
    DeviceGuard g(device_);
    (void)prepare(batch, in_place);
  }

  // Batched transform on device memory (the operator behind Fft::transform / transform_in_place).
  void exec(const void* d_in, void* d_out, size_t batch, int code, hipStream_t stream, Profiler* prof = nullptr) const {
    if (!d_in || !d_out) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "null buffer");
    if (code < 0 || code > 4) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "unknown transform code");
    if (batch == 0) return;
    DeviceGuard g(device_);
    // fft.rs:20-25 is_forward; autosort/mod.rs:381-385 scale computed in T
    const bool inverse = !(code == ::fourier::c::FOURIER_TRANSFORM_FFT || code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_FFT);
    double scale = 1.0;
    if (code == ::fourier::c::FOURIER_TRANSFORM_IFFT) scale = (double)((T)1 / (T)n_);
    else if (code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_FFT || code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_IFFT)
      scale = (double)((T)1 / std::sqrt((T)n_));
    const cpx<T>* in = (const cpx<T>*)d_in;
    cpx<T>* out = (cpx<T>*)d_out;
    const bool in_place = (d_in == d_out);
    if (mix_) {  // every pass stays in LDS: one launch, in place allowed (a workgroup reads its transforms first)
      const bool scaled = code == ::fourier::c::FOURIER_TRANSFORM_IFFT || code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_FFT ||
                          code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_IFFT;  // mod.rs:381-385
      mix_->run(in, out, batch, !inverse, scaled,
                scale, stream, prof);
      return;
    }
    const size_t chunk = prepare(batch, in_place);

    if (gen_) {
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        gen_->run(in + b0 * n_, out + b0 * n_, (cpx<T>*)scratch_.p, nb, inverse, scale, stream, prof);
      }
      return;
    }
    if (!blu_) {
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        eng_->run(in + b0 * n_, out + b0 * n_, (cpx<T>*)scratch_.p, nb, inverse, scale, nullptr, force_scratch_, stream, prof, 0, nxcd_);
      }
      return;
    }
    // Bluestein (bluesteins.rs:215-259): work = x.in (zero padded) ; FFT_M ; .w ; IFFT_M ; out = work.x.scale
    if (small_fused_) {  // M <= 2^15: the whole chirp-z in one launch, no work array
      eng_->run_bluestein_small(in, out, batch, xtab_.p, wtab_.p, n_, inverse, scale, stream, prof, nxcd_);
      return;
    }
    cpx<T>* work = (cpx<T>*)work_.p;
    for (size_t b0 = 0; b0 < batch; b0 += chunk) {
      const size_t nb = std::min(chunk, batch - b0);
      BluArgs pre{in + b0 * n_, work, xtab_.p, (uint64_t)n_, (uint64_t)m_, (uint64_t)nb, inverse, 1.0};
      const int np = (int)eng_->num_passes();
      if (fused_ && conv_) {
        // three sweeps instead of four: first forward pass (chirp-in fused), the conv kernel (last forward pass,
        // (.) w, first inverse pass), last inverse pass (chirp-out fused); intermediates ping-pong work/scratch
        typename Pow2Engine<T>::BluIO bin, bout;
        bin.io = IO_BLU_IN; bin.xtab = xtab_.p; bin.n = n_; bin.swap = inverse;
        if (chirp_compute_) { bin.p_tab = chirp_p_.p; bin.u_tab = chirp_u_.p; bin.tn_lo = tn_lo_.p; bin.tn_hi = tn_hi_.p; bin.tn_bits = tn_bits_; }
        bout.io = IO_BLU_OUT; bout.xtab = xtab_.p; bout.n = n_; bout.swap = inverse;
        const Pow2Engine<T>& inv = eng_inv_ ? *eng_inv_ : *eng_;
        cpx<T>* bufs[2] = {work, (cpx<T>*)scratch_.p};
        const cpx<T>* src = in + b0 * n_;
        int cur = 0;
        for (int p = 0; p + 1 < np; ++p) {
          eng_->launch_pass((size_t)p, src, bufs[cur], nb, false, 1.0, stream, prof, 1 + p, nxcd_, p == 0 ? bin : typename Pow2Engine<T>::BluIO());
          src = bufs[cur]; cur ^= 1;
        }
        eng_->launch_conv(src, bufs[cur], nb, wtab_.p, stream, prof, np, nxcd_);
        src = bufs[cur]; cur ^= 1;
        for (int p = 1; p < np; ++p) {
          const bool last = (p + 1 == np);
          cpx<T>* dst = last ? out + b0 * n_ : bufs[cur];
          inv.launch_pass((size_t)p, src, dst, nb, true, last ? scale : 1.0, stream, prof, 1 + np + p, nxcd_,
                          last ? bout : typename Pow2Engine<T>::BluIO());
          src = dst; cur ^= 1;
        }
        continue;
      }
      if (fused_) {
        // chirp multiply + zero pad fused into the forward inner FFT's first pass, chirp * scale fused into
        // the inverse inner FFT's last pass: no separate sweeps over the M-point work array
        typename Pow2Engine<T>::BluIO bin, bout;
        bin.io = IO_BLU_IN; bin.xtab = xtab_.p; bin.n = n_; bin.swap = inverse;
        if (chirp_compute_) { bin.p_tab = chirp_p_.p; bin.u_tab = chirp_u_.p; bin.tn_lo = tn_lo_.p; bin.tn_hi = tn_hi_.p; bin.tn_bits = tn_bits_; }
        bout.io = IO_BLU_OUT; bout.xtab = xtab_.p; bout.n = n_; bout.swap = inverse;
        eng_->run(in + b0 * n_, work, (cpx<T>*)scratch_.p, nb, false, 1.0, (const cpx<T>*)wtab_.p, false, stream, prof, 1,
                  nxcd_, bin);
        eng_->run(work, out + b0 * n_, (cpx<T>*)scratch_.p, nb, true, scale, nullptr, false, stream, prof, 1 + np, nxcd_, bout);
        continue;
      }
      PROF_BEGIN(prof, 0);
      FOURIER_LAUNCH(&blu_pre_kernel<T>, elementwise_grid(nb * m_), 256, 0, stream, pre);
      PROF_END(prof);
      eng_->run(work, work, (cpx<T>*)scratch_.p, nb, false, 1.0, (const cpx<T>*)wtab_.p, false, stream, prof, 1, nxcd_);
      eng_->run(work, work, (cpx<T>*)scratch_.p, nb, true, 1.0, nullptr, false, stream, prof, 1 + np, nxcd_);
      BluArgs post{work, out + b0 * n_, xtab_.p, (uint64_t)n_, (uint64_t)m_, (uint64_t)nb, inverse, scale};
      PROF_BEGIN(prof, 1 + 2 * np);
      FOURIER_LAUNCH(&blu_post_kernel<T>, elementwise_grid(nb * n_), 256, 0, stream, post);
      PROF_END(prof);
    }
  }

  // Wait for everything queued on `stream` of the plan's device (the blocking half of a stream-ordered batched call).
  void synchronize(hipStream_t stream) const {
    DeviceGuard g(device_);
    HIP_CHECK(hipStreamSynchronize(stream));
  }

  // Legacy host-buffer path (fourier-ffi/src/lib.rs:31-59): H2D, one transform, D2H, synchronous.
  void exec_host(const void* h_in, void* h_out, int code) const {
    if (!h_in || !h_out) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "null buffer");
    if (code < 0 || code > 4) return;  // unknown code: silent no-op (lib.rs:10)
    DeviceGuard g(device_);
    const size_t bytes = n_ * ELEM;
    pinned_.ensure(bytes);
    // the plan's own non-blocking stream: a legacy call never serialises against the NULL stream or any other
    // stream of the process (a relinked, threaded C/C++ program keeps its concurrency; one thread per handle)
    if (!legacy_stream_) HIP_CHECK(hipStreamCreateWithFlags(&legacy_stream_, hipStreamNonBlocking));
    const hipStream_t st = legacy_stream_;
    {
      const CopyJob in_job{pinned_.h, h_in, bytes};
      parallel_copy(&in_job, 1);  // one thread below 4 MiB, a few above (a 2^20-point transform is 8-16 MiB)
    }
    if (bytes <= ZERO_COPY_MAX) {
      // small transforms are latency-bound: the kernels read and write the mapped host buffer directly over
      // PCIe (every plan reads its input once and writes its output once) -- one launch chain, one sync
      exec(pinned_.d, pinned_.d, 1, code, st);
    } else {
      hostio_.ensure(bytes);
      HIP_CHECK(hipMemcpyAsync(hostio_.p, pinned_.h, bytes, hipMemcpyHostToDevice, st));
      exec(hostio_.p, hostio_.p, 1, code, st);
      HIP_CHECK(hipMemcpyAsync(pinned_.h, hostio_.p, bytes, hipMemcpyDeviceToHost, st));
    }
    // (polling hipStreamQuery before this blocking wait was measured: 27.0-27.4 vs 26.7 us per N = 4096 call -- the runtime's
    // own wait already spins; profiles/r03_s19_c1_spin_poll_ab.jsonl)
    HIP_CHECK(hipStreamSynchronize(st));
    const CopyJob out_job{h_out, pinned_.h, bytes};
    parallel_copy(&out_job, 1);
  }
  static constexpr size_t ZERO_COPY_MAX = 256 * 1024;

  // Batched transform on HOST memory (extension; the reference's callers hold host slices, fft.rs:48-61): `batch`
  // contiguous transforms are streamed through the device in chunks.  NSLOTS slots of pinned staging + device buffer;
  // the H2D copy of chunk i+1, the kernels of chunk i and the D2H copy of chunk i-1 run on three streams, and
  // the calling thread (helped by a few copy threads) moves pageable user memory in and out of the staging
  // buffers meanwhile.  Synchronous: returns when `h_out` is complete.  h_in == h_out is allowed.
  void exec_host_batch(const void* h_in, void* h_out, size_t batch, int code) const {
    if (!h_in || !h_out) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "null buffer");
    if (code < 0 || code > 4) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "unknown transform code");
    if (batch == 0) return;
    DeviceGuard g(device_);
    const size_t per = n_ * ELEM;
    const size_t chunk = std::max<size_t>(1, std::min<size_t>(batch, host_chunk_bytes_ / per));
    const size_t nchunks = (batch + chunk - 1) / chunk;
    pipe_.ensure(chunk * per);
    const char* src = (const char*)h_in;
    char* dst = (char*)h_out;
    auto chunk_bytes = [&](size_t i) { return std::min(chunk, batch - i * chunk) * per; };
    for (size_t i = 0; i < nchunks + NSLOTS; ++i) {
      const int s = (int)(i % NSLOTS);
      CopyJob jobs[2];
      int njobs = 0;
      if (i >= NSLOTS) {  // chunk i-NSLOTS used this slot: its result is in the staging buffer once its D2H has finished
        HIP_CHECK(hipEventSynchronize(pipe_.d2h_done[s]));
        jobs[njobs++] = {dst + (i - NSLOTS) * chunk * per, pipe_.pin_out[s].h, chunk_bytes(i - NSLOTS)};
      }
      if (i < nchunks) jobs[njobs++] = {pipe_.pin_in[s].h, src + i * chunk * per, chunk_bytes(i)};
      parallel_copy(jobs, njobs);  // result of chunk i-NSLOTS out of, input of chunk i into the staging buffers, together
      if (i < nchunks) {
        const size_t bytes = chunk_bytes(i);
        HIP_CHECK(hipMemcpyAsync(pipe_.dev[s].p, pipe_.pin_in[s].h, bytes, hipMemcpyHostToDevice, pipe_.s_h2d));
        HIP_CHECK(hipEventRecord(pipe_.h2d_done[s], pipe_.s_h2d));
        HIP_CHECK(hipStreamWaitEvent(pipe_.s_comp, pipe_.h2d_done[s], 0));
        exec(pipe_.dev[s].p, pipe_.dev[s].p, bytes / per, code, pipe_.s_comp);
        HIP_CHECK(hipEventRecord(pipe_.comp_done[s], pipe_.s_comp));
        HIP_CHECK(hipStreamWaitEvent(pipe_.s_d2h, pipe_.comp_done[s], 0));
        HIP_CHECK(hipMemcpyAsync(pipe_.pin_out[s].h, pipe_.dev[s].p, bytes, hipMemcpyDeviceToHost, pipe_.s_d2h));
        HIP_CHECK(hipEventRecord(pipe_.d2h_done[s], pipe_.s_d2h));
      }
    }
  }
  static constexpr size_t HOST_CHUNK_BYTES = (size_t)32 << 20;
  static constexpr size_t NSLOTS = 4;  // chunks in flight: copy-in, H2D, kernels, D2H + copy-out each take about one chunk time

 private:
  struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
      if (hipGetDevice(&prev) != hipSuccess) prev = -1;
      if (prev != dev) (void)hipSetDevice(dev);
      else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  };
  static unsigned elementwise_grid(size_t elems) {
    const size_t blocks = (elems + 255) / 256;
    return (unsigned)std::min<size_t>(std::max<size_t>(blocks, 1), 256 * 32);
  }

  void init_bluestein() {
    blu_ = true;
    if (n_ > ((size_t)1 << 26)) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "Bluestein sizes above 2^26 are not supported");
    m_ = 1;
    while (m_ < 2 * n_ - 1) m_ <<= 1;  // bluesteins.rs:110
    if (2 * n_ > m_) throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "Bluestein: M < 2N");  // the fused end passes rely on it
    // forward inner plan: the larger pass first (2048 x 1024 at M = 2^21), so the conv kernel runs at the SHORTER length
    // and the end passes at the longer one.  FOURIER_BLU_SHORT_FIRST=1 (experiment) swaps the roles: 1024 x 2048 forward,
    // end passes of length 1024, conv kernel at 2048.
    const bool short_first = dev_env("FOURIER_BLU_SHORT_FIRST") != nullptr;
    eng_.reset(new Pow2Engine<T>(m_, short_first));
    eng_->enable_bluestein_fusion();
    fused_ = eng_->can_fuse_bluestein();
    small_fused_ = eng_->enable_bluestein_small();
    if (fused_ && eng_->can_conv()) {
      // the inverse inner FFT must begin with the pass length the forward one ends with: the same plan when the
      // lengths read the same in both directions, otherwise its mirror image
      eng_->enable_conv();
      if (!eng_->palindromic()) {
        eng_inv_.reset(new Pow2Engine<T>(m_, !short_first));
        eng_inv_->enable_bluestein_fusion();
      }
      conv_ = conv_ok_ = true;
    }
    // chirp exp(-i*pi*k^2/n), angle reduced exactly with k^2 mod 2n (the reference leaves it
    // unreduced, bluesteins.rs:10,31,57; the reduction only removes f64 argument error)
    std::vector<double> cr(n_), ci(n_);
    const uint64_t two_n = 2 * (uint64_t)n_;
    for (size_t k = 0; k < n_; ++k) {
      const uint64_t r = (uint64_t)(((unsigned __int128)k * k) % two_n);
      const double ang = M_PI * (double)r / (double)n_;
      cr[k] = std::cos(ang); ci[k] = -std::sin(ang);
    }
    std::vector<cpx<T>> x(n_);
    for (size_t k = 0; k < n_; ++k) x[k] = {(T)cr[k], (T)ci[k]};  // x_fwd, bluesteins.rs:51-61
    xtab_.upload(x);
    if (fused_ && !small_fused_) {
      // Tables for the chirp-in pass that computes the chirp instead of reading x (a quarter of that pass's traffic):
      // index k = row*cn + b  =>  x[k] = W_2n^{(row*cn)^2} * W_2n^{b^2} * W_n^{cn*row*b}; exact exponents, f64 trig, cast.
      const uint64_t cn = eng_->first_cn(), rows = (uint64_t)eng_->first_len() / 2;
      std::vector<cpx<T>> pt(rows), ut(cn);
      for (uint64_t r = 0; r < rows; ++r) {
        const unsigned __int128 k = (unsigned __int128)r * cn;
        double re, im;
        unit_root((uint64_t)((k * k) % two_n), two_n, re, im);
        pt[r] = {(T)re, (T)im};
      }
      for (uint64_t b = 0; b < cn; ++b) {
        double re, im;
        unit_root((uint64_t)(((unsigned __int128)b * b) % two_n), two_n, re, im);
        ut[b] = {(T)re, (T)im};
      }
      chirp_p_.upload(pt);
      chirp_u_.upload(ut);
      tn_bits_ = (uint32_t)((ilog2(n_) + 1) / 2);
      std::vector<cpx<T>> lo((size_t)1 << tn_bits_), hi((size_t)(n_ >> tn_bits_) + 1);
      for (size_t e = 0; e < lo.size(); ++e) { double re, im; unit_root(e, n_, re, im); lo[e] = {(T)re, (T)im}; }
      for (size_t h = 0; h < hi.size(); ++h) { double re, im; unit_root((uint64_t)h << tn_bits_, n_, re, im); hi[h] = {(T)re, (T)im}; }
      tn_lo_.upload(lo);
      tn_hi_.upload(hi);
      // Default: only where it pays.  Measured (profiles/r03_s7_chirp_compute_ab.jsonl): C4 (N = 999983, first pass of length
      // 2048 on 8-column tiles) 2.76-2.85 vs 2.92-3.08 ms per 512, f64 2.61 vs 2.88; N = 40000 / 65537 (the 0.3-0.5 MB table
      // is L2-resident anyway) and N = 2200000 (first pass of length 256: 32-column tiles, eight times the per-tile table
      // work) are 5-17 % SLOWER.  So: a long first pass and a table beyond an XCD's L2.
      chirp_compute_ = eng_->first_len() >= 1024 && n_ * ELEM >= ((size_t)4 << 20);
    }
    // w = FFT_M(conj chirp, mirrored) (bluesteins.rs:18-48), evaluated in f64 on the host, with the
    // inner IFFT's 1/M (bluesteins.rs:239 -> mod.rs:383) folded in.
    std::vector<double> wr(m_, 0.0), wi(m_, 0.0);
    for (size_t k = 0; k < n_; ++k) {
      wr[k] = cr[k]; wi[k] = -ci[k];
      if (k) { wr[m_ - k] = cr[k]; wi[m_ - k] = -ci[k]; }
    }
    host_fft(wr, wi);
    std::vector<cpx<T>> w(m_);
    const double inv_m = 1.0 / (double)m_;
    for (size_t k = 0; k < m_; ++k) w[k] = {(T)(wr[k] * inv_m), (T)(wi[k] * inv_m)};
    wtab_.upload(w);
  }

  // pageable <-> pinned copies of a chunk, split over a few threads (one core moves ~10 GB/s, PCIe wants 50+ each way)
  struct CopyJob { void* dst; const void* src; size_t bytes; };
  static void parallel_copy(const CopyJob* jobs, int njobs) {
    std::vector<std::thread> th;
    for (int j = 0; j < njobs; ++j) {
      const CopyJob job = jobs[j];
      const size_t nt = std::max<size_t>(1, std::min<size_t>(COPY_THREADS, job.bytes / ((size_t)2 << 20)));
      // ceil(bytes / nt) rounded up to a page: nt * piece >= bytes for every byte count (floor division dropped
      // the last r < nt bytes of jobs of the form nt*4096*k + r)
      const size_t piece = (((job.bytes + nt - 1) / nt) + 4095) & ~(size_t)4095;
      for (size_t t = 0; t < nt; ++t) {
        const size_t off = t * piece;
        if (off >= job.bytes) break;
        const size_t len = std::min(piece, job.bytes - off);
        if (nt == 1 && njobs == 1) { std::memcpy(job.dst, job.src, len); return; }
        th.emplace_back([=] { std::memcpy((char*)job.dst + off, (const char*)job.src + off, len); });
      }
    }
    for (auto& t : th) t.join();
  }
  static constexpr size_t COPY_THREADS = 12;
  struct HostPipe {  // exec_host_batch: NSLOTS slots, three streams
    static constexpr int NS = 4;
    PinnedBuf pin_in[NS], pin_out[NS];
    DevBuf dev[NS];
    hipStream_t s_h2d = nullptr, s_comp = nullptr, s_d2h = nullptr;
    hipEvent_t h2d_done[NS] = {}, comp_done[NS] = {}, d2h_done[NS] = {};
    void ensure(size_t bytes) {
      if (!s_h2d) {
        HIP_CHECK(hipStreamCreateWithFlags(&s_h2d, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&s_d2h, hipStreamNonBlocking));
        for (int s = 0; s < NS; ++s) {
          HIP_CHECK(hipEventCreateWithFlags(&h2d_done[s], hipEventDisableTiming));
          HIP_CHECK(hipEventCreateWithFlags(&comp_done[s], hipEventDisableTiming));
          HIP_CHECK(hipEventCreateWithFlags(&d2h_done[s], hipEventDisableTiming));
        }
      }
      for (int s = 0; s < NS; ++s) { pin_in[s].ensure(bytes); pin_out[s].ensure(bytes); dev[s].ensure(bytes); }
    }
    ~HostPipe() {
      for (int s = 0; s < NS; ++s)
        for (hipEvent_t e : {h2d_done[s], comp_done[s], d2h_done[s]})
          if (e) (void)hipEventDestroy(e);
      for (hipStream_t st : {s_h2d, s_comp, s_d2h})
        if (st) (void)hipStreamDestroy(st);
    }
  };
  static_assert(NSLOTS == HostPipe::NS, "slot count");
  mutable HostPipe pipe_;

  size_t n_, m_ = 0;
  int device_ = 0;
  bool blu_ = false;
  std::unique_ptr<Pow2Engine<T>> eng_, eng_inv_;  // eng_inv_: mirrored inverse plan of a conv-fused Bluestein
  std::unique_ptr<MixedEngine<T>> mix_;
  std::unique_ptr<GenericEngine<T>> gen_;  // 2^a*3^b, a < 12, beyond the LDS kernels
  DevBuf xtab_, wtab_;
  DevBuf chirp_p_, chirp_u_, tn_lo_, tn_hi_;  // chirp-in pass computing the chirp (init_bluestein)
  uint32_t tn_bits_ = 0;
  bool chirp_compute_ = false;  // option "bluestein_chirp_compute"
  mutable DevBuf scratch_, work_, hostio_;
  mutable PinnedBuf pinned_;
  mutable hipStream_t legacy_stream_ = nullptr;  // legacy host-buffer calls (exec_host)
  size_t chunk_bytes_ = 0;
  size_t host_chunk_bytes_ = HOST_CHUNK_BYTES;  // exec_host_batch: bytes of one streamed chunk
  bool force_scratch_ = false;
  bool fused_ = false;  // Bluestein: chirp steps fused into the inner passes
  bool small_fused_ = false;  // Bluestein with M <= 2^15: everything in one launch
  bool conv_ = false, conv_ok_ = false;  // Bluestein: forward LAST + (.)w + inverse FIRST in one launch
  unsigned nxcd_ = 8;
  mutable int status_ = 0;
  std::string desc_;
};

template <typename T> static Plan<T>* create_plan(size_t n, int device) {
  try {
    return new Plan<T>(n, device);
  } catch (...) {
    return nullptr;  // never unwind into C (fourier-ffi/src/lib.rs:18-19)
  }
}

template <typename T, typename F> static int guarded(const Plan<T>* p, F&& f) {
  if (!p) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
  p->set_status(::fourier::c::FOURIER_HIP_OK);  // last_status = status of the LAST call on this handle
  try {
    f();
    return ::fourier::c::FOURIER_HIP_OK;
  } catch (const EngineError& e) {
    p->set_status(e.status);
    if (getenv("FOURIER_HIP_VERBOSE")) fprintf(stderr, "libfourier: %s\n", e.what());
    return e.status;
  } catch (const std::bad_alloc&) {
    p->set_status(::fourier::c::FOURIER_HIP_OUT_OF_MEMORY);
    return ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY;
  } catch (...) {
    p->set_status(::fourier::c::FOURIER_HIP_RUNTIME_ERROR);
    return ::fourier::c::FOURIER_HIP_RUNTIME_ERROR;
  }
}

}  // namespace fourier_hip

// ---------------------------------------------------------------------------------------------
// C ABI (declared in include/fourier.h)
using namespace fourier_hip;
namespace fc = ::fourier::c;

#define FOURIER_DEFINE_ABI(T, SUFFIX)                                                                            \
  extern "C" fc::fourier_fft_##SUFFIX* fourier_create_##SUFFIX(size_t size) {                                    \
    return (fc::fourier_fft_##SUFFIX*)create_plan<T>(size, -1);                                                  \
  }                                                                                                              \
  extern "C" fc::fourier_fft_##SUFFIX* fourier_hip_create_##SUFFIX(size_t size, int device) {                    \
    return (fc::fourier_fft_##SUFFIX*)create_plan<T>(size, device);                                              \
  }                                                                                                              \
  extern "C" void fourier_destroy_##SUFFIX(fc::fourier_fft_##SUFFIX* h) {                                        \
    try { delete (Plan<T>*)h; } catch (...) {}                                                                   \
  }                                                                                                              \
  extern "C" void fourier_transform_in_place_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, std::complex<T>* x, int code) { \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    (void)guarded<T>(p, [&] { p->exec_host(x, x, code); });                                                      \
  }                                                                                                              \
  extern "C" void fourier_transform_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, const std::complex<T>* in,       \
                                             std::complex<T>* out, int code) {                                   \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    (void)guarded<T>(p, [&] { p->exec_host(in, out, code); });                                                   \
  }                                                                                                              \
  extern "C" size_t fourier_hip_size_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                               \
    return h ? ((const Plan<T>*)h)->size() : 0;                                                                  \
  }                                                                                                              \
  extern "C" int fourier_hip_transform_batch_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, const void* d_in,       \
                                                      void* d_out, size_t batch, int code, void* stream) {       \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    return guarded<T>(p, [&] { p->exec(d_in, d_out, batch, code, (hipStream_t)stream); });                       \
  }                                                                                                              \
  extern "C" int fourier_hip_reserve_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, size_t batch, int in_place) {   \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    return guarded<T>(p, [&] { p->reserve_for(batch, in_place != 0); });                                         \
  }                                                                                                              \
  extern "C" int fourier_hip_device_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                                \
    return h ? ((const Plan<T>*)h)->device() : -1;                                                               \
  }                                                                                                              \
  extern "C" int fourier_hip_synchronize_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, void* stream) {            \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    return guarded<T>(p, [&] { p->synchronize((hipStream_t)stream); });                                          \
  }                                                                                                              \
  extern "C" int fourier_hip_transform_batch_host_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, const std::complex<T>* in, \
                                                           std::complex<T>* out, size_t batch, int code) {      \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    return guarded<T>(p, [&] { p->exec_host_batch(in, out, batch, code); });                                     \
  }                                                                                                              \
  extern "C" int fourier_hip_profile_##SUFFIX(const fc::fourier_fft_##SUFFIX* h, const void* d_in, void* d_out,  \
                                              size_t batch, int code, void* stream, int nslots, float* ms_sum,   \
                                              int* launches) {                                                   \
    const Plan<T>* p = (const Plan<T>*)h;                                                                        \
    if (!ms_sum || !launches || nslots <= 0) return fc::FOURIER_HIP_INVALID_ARGUMENT;                            \
    return guarded<T>(p, [&] {                                                                                   \
      Profiler prof((hipStream_t)stream);                                                                        \
      p->exec(d_in, d_out, batch, code, (hipStream_t)stream, &prof);                                             \
      prof.collect(nslots, ms_sum, launches);                                                                    \
    });                                                                                                          \
  }                                                                                                              \
  extern "C" const char* fourier_hip_slot_names_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                    \
    static thread_local std::string s;                                                                           \
    s = h ? ((const Plan<T>*)h)->slot_names() : "";                                                              \
    return s.c_str();                                                                                            \
  }                                                                                                              \
  extern "C" int fourier_hip_last_status_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                           \
    return h ? ((const Plan<T>*)h)->last_status() : fc::FOURIER_HIP_INVALID_ARGUMENT;                            \
  }                                                                                                              \
  extern "C" int fourier_hip_set_option_##SUFFIX(fc::fourier_fft_##SUFFIX* h, const char* key, long long v) {    \
    if (!h || !key) return fc::FOURIER_HIP_INVALID_ARGUMENT;                                                     \
    try { return ((Plan<T>*)h)->set_option(key, v); } catch (...) { return fc::FOURIER_HIP_INVALID_ARGUMENT; }   \
  }                                                                                                              \
  extern "C" const char* fourier_hip_describe_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                      \
    return h ? ((const Plan<T>*)h)->describe() : "";                                                             \
  }                                                                                                              \
  extern "C" double fourier_hip_model_bytes_##SUFFIX(const fc::fourier_fft_##SUFFIX* h) {                        \
    return h ? ((const Plan<T>*)h)->model_bytes() : 0.0;                                                         \
  }

FOURIER_DEFINE_ABI(float, float)
FOURIER_DEFINE_ABI(double, double)

extern "C" const char* fourier_hip_status_string(int status) {
  switch (status) {
    case fc::FOURIER_HIP_OK: return "ok";
    case fc::FOURIER_HIP_INVALID_ARGUMENT: return "invalid argument";
    case fc::FOURIER_HIP_OUT_OF_MEMORY: return "out of device memory";
    case fc::FOURIER_HIP_RUNTIME_ERROR: return "HIP runtime error";
    case fc::FOURIER_HIP_UNSUPPORTED: return "unsupported size";
    default: return "unknown status";
  }
}

#ifdef FOURIER_EMU
// test-only: LDS bank-conflict statistics gathered by the emulator
extern "C" void fourier_emu_lds_stats(uint64_t* instr, uint64_t* cycles, uint64_t* ideal, int reset) {
  auto& s = hipemu::lds_stats();
  *instr = s.instr; *cycles = s.cycles; *ideal = s.ideal;
  if (reset) { s.instr = 0; s.cycles = 0; s.ideal = 0; }
}
// test-only: number of device allocations so far (tests/test_engine_emu.py: reserve makes calls allocation-free)
extern "C" uint64_t fourier_emu_alloc_count() { return hipemu::alloc_count(); }
#endif
