// engine_mixed.h -- small mixed-radix sizes in LDS: 2^a * 3^b (b > 0) on the reference's own schedule and tables
// (autosort/mod.rs:20-46,104-116); lengths with factors 5..13 on the same pass.
#pragma once
#include "engine_common.h"
#include "mixed_schedule.h"

namespace fourier_hip {

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
  typedef MixKernel Kernel;
  // the per-length kernel where one is instantiated (every 2^a*3^b up to MAX_N, and the common lengths with factors
  // 5 / 7), else the runtime-parameterised kernel
  static Kernel pick_kernel(size_t n) {
    // runtime-parameterised: about 1024 points per workgroup up to 1024 points, then one transform per workgroup -- 256 threads
    // x 4 / 8 points up to 2048 points, 512 x 8 up to 4096, 1024 x 8 up to 8192 (1024 x 4 for 2049..4096 measured slower than
    // 256 x 16: 3125 f32 19 % against 24 %, r03_s22)
    Kernel k{nullptr, (uint32_t)std::max<size_t>(1, 1024 / n), 1, 256};
    const int maxp = (n % 11 == 0 || n % 13 == 0) ? 13 : ((n % 5 == 0 || n % 7 == 0) ? 7 : 3);
    const size_t pts = k.group * n;
    const Real<T> real{};
    if (pts <= 1024 && mix_threads<T>((uint32_t)n) == 128) { k.fn = get_mixed_rt_kernel(real, maxp, 8, 128); k.threads = 128; }  // few work items per pass
    // f64 with a radix-13 butterfly spills at 4 points per thread x 256 (36 B/lane, round 3): the 8-point instantiation does not
    else if (pts <= 1024) k.fn = get_mixed_rt_kernel(real, maxp, (sizeof(T) == 8 && maxp == 13) ? 8 : 4, 256);
    else if (pts <= 2048) k.fn = get_mixed_rt_kernel(real, maxp, 8, 256);
    // (f64 with a radix-13 butterfly does not fit 128 registers beyond that: no instantiation, the length takes Bluestein)
    else if (pts <= 4096) { k.fn = get_mixed_rt_kernel(real, maxp, 8, 512); k.threads = 512; }
    else if (pts <= 8192) { k.fn = get_mixed_rt_kernel(real, maxp, 8, 1024); k.threads = 1024; }
    if (dev_env("FOURIER_MIX_GENERIC") && k.fn) return k;
    // the per-length kernel where one is instantiated (kernels_mixed_ct.cpp, dealt over FOURIER_MIX_SHARDS translation units)
    MixKernel ct{};
#define FOURIER_TRY_MIX_SHARD(I, TT) if (!ct.fn && get_mixed_ct_kernel_s##I(real, n, ct)) {}
    FOURIER_MIX_SHARD_LIST(FOURIER_TRY_MIX_SHARD, T)
#undef FOURIER_TRY_MIX_SHARD
    if (ct.fn) k = ct;
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

  // deferred_kernel: a length without any ahead-of-time kernel (beyond the runtime kernel's reach); the plan is usable only
  // after specialise() has succeeded (Plan::set_option "specialise" on a Bluestein plan)
  explicit MixedEngine(size_t n, bool deferred_kernel = false) : n_(n) {
    if (!factor(n, radices_)) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "length does not factor over 2, 3, 5, 7, 11, 13");
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
    tw_entries_ = tw.size();
    if (deferred_kernel) return;
    const Kernel k = pick_kernel(n);
    fn_ = k.fn; group_ = k.group; nbuf_ = k.nbuf; threads_ = k.threads;
    per_length_ = (k.group == mix_group<T>((uint32_t)n) && k.threads == mix_threads<T>((uint32_t)n) && fn_ != nullptr && !is_runtime_kernel(k));
    smem_ = nbuf_ * (size_t)group_ * n * sizeof(cpx<T>);
    if (smem_ > MAX_LDS) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "mixed-radix length needs the per-length kernel");
    raise_smem_limit((const void*)fn_, smem_);
  }
  std::string describe() const {
    std::string d;
    for (const uint32_t r : radices_) d += (d.empty() ? "" : ".") + std::to_string(r);
    if (rtc_.fn) d += " specialised";
    return d;
  }
  bool usable() const { return fn_ != nullptr || rtc_.fn != nullptr; }
  bool specialised() const { return rtc_.fn != nullptr; }
  // Plan option "specialise": compile this length's own kernel (mixed_radix_kernel_ct<T, n>, the form of the ahead-of-time
  // per-length kernels) with hipRTC and use it from now on.  Nothing to do where the plan already runs a per-length kernel.
  // Status: OK, or UNSUPPORTED (no hipRTC, or the compilation failed) -- the plan then keeps the kernel it had.
  // LDS bytes of the specialised kernel of length n (the launch shape rules of mixed_schedule.h)
  static size_t specialised_lds_bytes(uint32_t n) {
    return (mix_inplace<T>(n) ? 1 : 2) * (size_t)mix_group<T>(n) * n * sizeof(cpx<T>);
  }
  static bool specialised_kernel_cached(size_t n) { return n <= MAX_N && rtc_cached(sizeof(T) == 8, (uint32_t)n, specialised_lds_bytes((uint32_t)n), false); }
  int specialise(std::string* why = nullptr, bool allow_compile = true) {
    if (rtc_.fn || per_length_) return ::fourier::c::FOURIER_HIP_OK;
    const uint32_t n = (uint32_t)n_;
    const uint32_t group = mix_group<T>(n), threads = mix_threads<T>(n);
    const size_t nbuf = mix_inplace<T>(n) ? 1 : 2;
    const size_t smem = specialised_lds_bytes(n);
    std::string reason;
    RtcKernel k;
    if (smem > MAX_LDS) reason = "the length does not fit a compute unit's LDS";
    else if (rtc_mixed_kernel(sizeof(T) == 8, n, smem, k, reason, false, allow_compile)) {
      rtc_ = k; group_ = group; threads_ = threads; nbuf_ = nbuf;
      smem_ = 0;  // the specialised kernel declares its LDS statically
      return ::fourier::c::FOURIER_HIP_OK;
    }
    if (why) *why = reason;
    return ::fourier::c::FOURIER_HIP_UNSUPPORTED;
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
#ifndef FOURIER_EMU
    if (rtc_.fn) {  // the kernel specialised at run time: a module function
      void* params[] = {&a};
      HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)rtc_.fn, (unsigned)grid, 1, 1, threads_, 1, 1, (unsigned)smem_, stream, params, nullptr));
    } else
#endif
    {
      FOURIER_LAUNCH(fn_, grid, threads_, smem_, stream, a);
    }
    PROF_END(prof);
  }

 private:
  size_t n_;
  std::vector<uint32_t> radices_;
  uint32_t threads_ = 256;
  void (*fn_)(MixArgs) = nullptr;
  size_t nbuf_ = 1;  // LDS buffers of `group_` transforms: 1 = in-place passes (2 = ping-pong, FOURIER_MIX_INPLACE_BYTES builds)
  uint32_t group_ = 1;
  size_t smem_ = 0, tw_entries_ = 0;
  bool per_length_ = false;  // fn_ is this length's own ahead-of-time kernel
  RtcKernel rtc_;            // ... or its own kernel compiled at run time (specialise)
  DevBuf tw_;
  static bool is_runtime_kernel(const Kernel& k) {
    const Real<T> real{};
    for (int maxp : {3, 7, 13})
      for (int ppt : {4, 8})
        for (int nt : {128, 256, 512, 1024})
          if (k.fn && k.fn == get_mixed_rt_kernel(real, maxp, ppt, nt)) return true;
    return false;
  }
};

}  // namespace fourier_hip
