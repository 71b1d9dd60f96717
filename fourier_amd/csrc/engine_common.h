// engine_common.h -- what the host-side translation units of libfourier.so share: error type, launch macro, device
// memory RAII, the per-kernel timing hook, and the kernel registry (each kernel family is instantiated in its own
// translation unit -- kernels_*.cpp, one object per precision, the per-length mixed-radix kernels in shards -- and handed
// to the plan layer through the get_*_kernel functions declared at the end of this file).
#pragma once
#ifndef FOURIER_EMU
#include <hip/hip_runtime.h>
#endif

#include <algorithm>
#include <atomic>
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


#include "kernel_args.h"
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


// Development switches (environment variables read at plan creation) exist only in lib/libfourier_experiments.so (A/B
// sessions, the GPU tests of the measured-slower designs) and in the emulator build of the CPU tests.  The product
// library's plan selection never depends on the environment of the process that links it; FOURIER_HIP_VERBOSE (error text
// on stderr) is the one variable it reads.  Decided at LINK time: both libraries are built from the same objects, the
// product links env_product.cpp (dev_env() returns nullptr, the experiment kernels' registry entries report "not
// available"), the experiments library links env_experiments.cpp and kernels_experiments.cpp (fft_l2fused_kernel,
// fft_last_split_kernel: DESIGN.md section 4 has their measurements).
const char* dev_env(const char* name);

// kernels that use more than 48 KiB of dynamic LDS must say so once
static inline void raise_smem_limit(const void* fn, size_t smem) {
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

enum { MODE_TWOLEVEL = 4 };  // host-side tag for fft_twolevel_kernel (both passes in one launch)
enum { MODE_ODD_LAST = 5 };  // host-side tag for odd_last_kernel (final radix-3^b pass of a 2^a*3^b plan)

// fft_l2fused_kernel: both passes of an N = L1 x L2 plan in one launch, intermediate in the XCD's L2
typedef void (*FusedKernel)(FusedArgs);
struct FusedInfo {
  FusedKernel fn = nullptr;
  int L1 = 0, L2 = 0, NT = 0, COLS_A = 0, COLS_B = 0;
  size_t smem = 0;
};

typedef void (*OddKernel)(OddArgs);
typedef void (*TinyKernel)(TinyArgs);
typedef void (*GenKernel)(GenArgs);
typedef void (*BluKernel)(BluArgs);
typedef void (*MixKernelFn)(MixArgs);
typedef void (*TiledKernelFn)(TiledArgs);
// a tile pass of mixed length L: columns per tile, threads, LDS bytes
// r1 != 0: the register-resident kernel of kernels_regtile.h (L = r1 x r2; its `tw` table is W_L^{j2 * k1}, [r1][r2])
struct TiledKernel { TiledKernelFn fn = nullptr; uint32_t L = 0, cols = 0, threads = 0; size_t smem = 0; uint32_t r1 = 0, r2 = 0; };
typedef void (*ChirpzKernelFn)(ChirpzArgs);
struct ChirpzKernel { ChirpzKernelFn fn = nullptr; uint32_t m = 0, r1 = 0, r2 = 0, r3 = 0, tpw = 0, threads = 64; size_t smem = 0; bool split = false, fact = false; };  // tpw: transforms per workgroup; split: re / im planes exchanged one after the other; fact: factored twiddle tables
// a mixed-radix LDS kernel with its launch shape: transforms per workgroup, LDS buffers of `group` transforms, threads
struct MixKernel { MixKernelFn fn; uint32_t group; size_t nbuf; uint32_t threads; };

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

// a kernel specialised at run time (rtc.cpp): launched through hipModuleLaunchKernel
struct RtcKernel { void* fn = nullptr; };
// mixed_radix_kernel_ct<float|double, n> compiled with hipRTC (cached per device, precision and length), its lds_bytes of LDS
// declared statically; false + the reason where hipRTC or the compilation is not available
// tile_pass: tiled_mixed_kernel_ct<T, n> (a column-tile pass of length n) instead of the whole-transform kernel
// allow_compile = false: only the process cache and the on-disk code-object cache are consulted (a few milliseconds)
bool rtc_mixed_kernel(bool f64, uint32_t n, size_t lds_bytes, RtcKernel& out, std::string& why, bool tile_pass = false, bool allow_compile = true);
// that kernel is in the process cache or has a file in the on-disk cache (no module is loaded)
bool rtc_cached(bool f64, uint32_t n, size_t lds_bytes, bool tile_pass);
// Library-wide policy for plans created from now on (fourier_hip_set_default_option "specialise_at_create", or the environment
// variable FOURIER_HIP_SPECIALISE read once): 0 = a plan never picks a run-time kernel by itself, 1 (default) = a plan created for a
// length with a specialised kernel in the on-disk cache loads it (no compilation ever happens implicitly), 2 = ... and compiles it
// where the cache has none (about a second per new length and machine)
int specialise_policy();
void set_specialise_policy(int v);
// ... and whether a 2^a 3^b length with a register-stage kernel listed on request takes it at create ("register_stages_at_create", or
// FOURIER_HIP_REGISTER_STAGES=1 read once): 0 (default) = the reference's schedule and its bits, 1 = the faster kernel, within rounding
int register_stages_default();
void set_register_stages_default(int v);

// ---------------------------------------------------------------------------------------------
// Kernel registry.  Real<T> selects the precision; every function is defined once per precision in the translation unit
// named beside it (compiled with -DFOURIER_TU_REAL=float / double).
template <typename T> struct Real {};
// number of translation units the per-length mixed-radix kernels are spread over (fourier_amd/build.py reads this line;
// the list below must name 0 .. FOURIER_MIX_SHARDS - 1)
#define FOURIER_MIX_SHARDS 8
#define FOURIER_MIX_SHARD_LIST(X, T) X(0, T) X(1, T) X(2, T) X(3, T) X(4, T) X(5, T) X(6, T) X(7, T)
#define FOURIER_DECLARE_MIX_SHARD(I, T) bool get_mixed_ct_kernel_s##I(Real<T>, size_t n, MixKernel& k);
// ... and the per-length register-stage kernels (kernels_regfft.cpp; fourier_amd/build.py reads this line too)
#define FOURIER_REGFFT_SHARDS 8
#define FOURIER_REGFFT_SHARD_LIST(X, T) X(0, T) X(1, T) X(2, T) X(3, T) X(4, T) X(5, T) X(6, T) X(7, T)
#define FOURIER_DECLARE_REGFFT_SHARD(I, T) ChirpzKernel get_regfft_kernel_s##I(Real<T>, uint32_t n, int variant);
#define FOURIER_DECLARE_REGISTRY(T)                                                                                    \
  /* kernels_pass.cpp: one tile shape (CG) per pass length L; conv = forward LAST + (.) w + inverse FIRST */           \
  KernelInfo get_kernel(Real<T>, int L, int mode, int io);                                                             \
  KernelInfo get_conv_kernel(Real<T>, int L);                                                                          \
  /* kernels_onelaunch.cpp: 2^11..2^15 in one launch; whole chirp-z in one launch for M = 2^k <= 2^15 */                \
  bool get_twolevel_kernel(Real<T>, int k, KernelInfo& info, int& l1, int& l2);                                        \
  bool get_blu_small_kernel(Real<T>, int k, KernelInfo& info);                                                         \
  /* kernels_misc.cpp */                                                                                               \
  TinyKernel get_tiny_kernel(Real<T>, size_t n);                                                                       \
  OddKernel get_odd_kernel(Real<T>, int r);                                                                            \
  GenKernel get_stockham_pass_kernel(Real<T>, int r);                                                                  \
  BluKernel get_blu_kernel(Real<T>, int which); /* 0 = pre, 1 = post, 2 = mul */                                       \
  /* kernels_mixed_rt.cpp: the runtime-parameterised LDS kernel (maxp in {3, 7, 13}), null where not instantiated */    \
  MixKernelFn get_mixed_rt_kernel(Real<T>, int maxp, int ppt, int nt);                                                 \
  /* kernels_mixed_ct.cpp, compiled FOURIER_MIX_SHARDS times per precision (-DFOURIER_MIX_SHARD=i): shard i of the  */  \
  /* per-length kernels; false when shard i holds no kernel for length n                                           */  \
  FOURIER_MIX_SHARD_LIST(FOURIER_DECLARE_MIX_SHARD, T)                                                                 \
  /* kernels_tiled.cpp (4 shards): column-tile pass of length L, 64 <= L <= 512, prime factors up to 7; fn == nullptr: none */ \
  TiledKernel get_tiled_kernel(Real<T>, uint32_t L);                                                                   \
  TiledKernel get_tiled_kernel_s0(Real<T>, uint32_t L); TiledKernel get_tiled_kernel_s1(Real<T>, uint32_t L);         \
  TiledKernel get_tiled_kernel_s2(Real<T>, uint32_t L); TiledKernel get_tiled_kernel_s3(Real<T>, uint32_t L);         \
  /* kernels_regtile.cpp (4 shards): the same pass with the transform in registers, L = R1 x R2, R1, R2 <= 32; fn == nullptr: none */ \
  /* which: 0 = the plain pass, 1 / 2 / 3 = Bluestein on a smooth M: chirp-in first pass, conv, chirp-out last pass */ \
  TiledKernel get_regtile_kernel(Real<T>, uint32_t L, int which = 0);                                                  \
  TiledKernel get_regtile_kernel_s0(Real<T>, uint32_t L, int which); TiledKernel get_regtile_kernel_s1(Real<T>, uint32_t L, int which); \
  TiledKernel get_regtile_kernel_s2(Real<T>, uint32_t L, int which); TiledKernel get_regtile_kernel_s3(Real<T>, uint32_t L, int which); \
  /* kernels_chirpz.cpp (4 shards): the whole chirp-z in one launch on a smooth M = R1 x R2 [x R3] (registers); fn == nullptr: none */ \
  ChirpzKernel get_chirpz_kernel(Real<T>, uint32_t m);                                                                 \
  ChirpzKernel get_chirpz_kernel_s0(Real<T>, uint32_t m); ChirpzKernel get_chirpz_kernel_s1(Real<T>, uint32_t m);     \
  ChirpzKernel get_chirpz_kernel_s2(Real<T>, uint32_t m); ChirpzKernel get_chirpz_kernel_s3(Real<T>, uint32_t m);     \
  /* kernels_regfft.cpp (FOURIER_REGFFT_SHARDS shards): a length with factors 5 ... 13 as a direct transform on the same register */ \
  /* stages, n = R1 x R2 [x R3], the lengths of regfft_shapes.h; fn == nullptr: none                                              */ \
  /* variant: 0 = as listed; 1 ... 4 = one of the exchange / table variants where an A/B build holds all four                    */ \
  ChirpzKernel get_regfft_kernel(Real<T>, uint32_t n, int variant = 0);                                               \
  FOURIER_REGFFT_SHARD_LIST(FOURIER_DECLARE_REGFFT_SHARD, T)                                                           \
  /* kernels_experiments.cpp (experiments library) or env_product.cpp (product: nothing available) */                  \
  bool get_fused_kernel(Real<T>, int k, FusedInfo& info);                                                              \
  KernelInfo get_split_kernel(Real<T>, int L, int io);                                                                 \
  /* persistent LAST pass that prefetches its next tile (fft_last_prefetch_kernel); fn == nullptr where not built */     \
  KernelInfo get_prefetch_kernel(Real<T>, int L, int io);                                                               \
  /* kernels_skeleton.cpp (experiments library): the tile passes of length 1024 / 2048 WITHOUT butterflies, twiddles and */ \
  /* LDS exchanges -- the load-tile / store-tile skeleton bench.py times as the streaming ceiling; product: fn == nullptr */ \
  KernelInfo get_skeleton_kernel(Real<T>, int L, int mode);
FOURIER_DECLARE_REGISTRY(float)
FOURIER_DECLARE_REGISTRY(double)
#undef FOURIER_DECLARE_REGISTRY

}  // namespace fourier_hip
