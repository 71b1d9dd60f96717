// plan.h -- the plan behind a handle: create_fft_f32/f64 (fourier/src/lib.rs:31-60) one level up -- Stockham where the length
// factors, else Bluestein (bluesteins.rs) -- with chunking, plan-owned buffers, the host-buffer paths and the error model.
#pragma once
#include "engine_pow2.h"
#include "engine_mixed.h"
#include "engine_generic.h"
#include "engine_tiled.h"

namespace fourier_hip {

// ---------------------------------------------------------------------------------------------
// host f64 radix-2 FFT, used only at plan time for the Bluestein w table (bluesteins.rs:46-47)
static inline void host_fft(std::vector<double>& re, std::vector<double>& im) {
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

// ... and for any length whose prime factors are small (Bluestein on a smooth M): one Stockham pass per prime factor,
// out[j + s*(p*i + k)] = W_cur^{i*k} sum_r W_p^{r*k} in[j + s*(i + (cur/p)*r)]   (mod.rs:203-284 with plain O(p^2) butterflies)
static inline void host_fft_any(std::vector<double>& re, std::vector<double>& im) {
  const size_t m = re.size();
  if (is_pow2(m)) { host_fft(re, im); return; }
  std::vector<double> ore(m), oim(m);
  size_t cur = m, s = 1;
  while (cur > 1) {
    size_t p = 2;
    while (cur % p) ++p;
    if (p > 16) throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "host_fft_any: a prime factor above 16");  // (tile lengths: up to 7)
    const size_t mm = cur / p;
    std::vector<double> wr(p * p), wi(p * p);
    for (size_t e = 0; e < p * p; ++e) unit_root(e % p, p, wr[e], wi[e]);  // W_p^{r*k} at [r * p + k] via (r*k) % p below
    for (size_t i = 0; i < mm; ++i) {
      std::vector<double> tr(p), ti(p);
      for (size_t k = 0; k < p; ++k) unit_root(i * k, cur, tr[k], ti[k]);
      for (size_t j = 0; j < s; ++j) {
        double xr[16], xi[16];
        for (size_t r = 0; r < p; ++r) { xr[r] = re[j + s * (i + mm * r)]; xi[r] = im[j + s * (i + mm * r)]; }
        for (size_t k = 0; k < p; ++k) {
          double yr = 0, yi = 0;
          for (size_t r = 0; r < p; ++r) {
            const size_t e = (r * k) % p;
            yr += xr[r] * wr[e] - xi[r] * wi[e]; yi += xr[r] * wi[e] + xi[r] * wr[e];
          }
          ore[j + s * (p * i + k)] = yr * tr[k] - yi * ti[k]; oim[j + s * (p * i + k)] = yr * ti[k] + yi * tr[k];
        }
      }
    }
    re.swap(ore); im.swap(oim);
    s *= p; cur = mm;
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
      // f32 2^20 = 1024 x 1024 (BASELINE configs[1]): every XCD walks its range of transforms in bands of eight adjacent tiles
      // (1 KiB of every row of eight transforms in flight per XCD instead of whole 8 KiB rows of one) -- 23.07 against 23.50 ms and
      // 23.05 against 23.70 ms per 4096 transforms on two boxes, bit-identical (profiles/r05_s2_*, r05_s3_c2_tile_walk_ab.jsonl);
      // f64, the L = 2048 plans and the Bluestein inner passes lose 0 - 8 % with it (r05_s3_*, r05_s4_*) and keep the tile order
      if (sizeof(T) == 4 && n == ((size_t)1 << 20) && !dev_env("FOURIER_NO_BAND_WALK")) nxcd_ = 8u | (4u << 8) | (8u << 12);
    } else if (tiled_before_pow2_tiles(n)) {
      // 2^a * 3^b with a >= 12 as TWO mixed-length tile passes instead of power-of-two tiles + odd passes (three or four round
      // trips).  Round 6, register tiles: 512 x 384 +25 %, 512 x 432 +26 ... 41 %, 576 x 576 +34 ... 50 %; 768 x 576 level, 768 x 768 -3 ... 5 %,
      // 1024 x 768 / 864 f32 -13 ... 17 % (profiles/r06_s40_long_tiles_ab.jsonl; the LDS tile passes of round 4 paid up to 384 x 384 only)
      tiled_.reset(new TiledMixedEngine<T>(n));
    } else if (Pow2Engine<T>::handles_mixed(n)) {
      // big-radix passes over the 2^a part (a >= 12), then a radix-3^b pass: three HBM round trips at full tile
      // efficiency beat the one-workgroup-per-CU LDS kernel where both apply (3*2^12 f32: 23 % vs 14 %)
      eng_.reset(new Pow2Engine<T>(n));
    } else if (regfft_route(n)) {
      // a length with factors 5 ... 13 that regfft_shapes.h lists in this precision: ONE launch with the transform on two or three register
      // stages (kernels_regfft.h) instead of the LDS mixed-radix kernel's round trip per small radix, two tile passes or Bluestein -- each
      // length at least 1.04 x faster than the route below it had (profiles/r06_s49 ... s51_regfft_*_ab.jsonl)
      regf_.reset(new BluRegEngine<T>(n, (uint32_t)n, true));
    } else if (MixedEngine<T>::handles(n) && try_mixed(n)) {
      // a length on the runtime-parameterised kernel takes its own kernel where the code-object cache has it (policy 2: compiles it)
      if (specialise_policy() >= 1) (void)mix_->specialise(nullptr, specialise_policy() >= 2);
      // ... and a 2^a 3^b length the register-stage kernel listed for it on request, where the library-wide default says so
      if (register_stages_default() == 1 && BluRegEngine<T>::has_direct_on_request(n)) {
        regf_.reset(new BluRegEngine<T>(n, (uint32_t)n, true, 100));
        regf_on_request_ = true;
      }
    } else if (TiledMixedEngine<T>::handles(n)) {
      tiled_.reset(new TiledMixedEngine<T>(n));  // two or three HBM round trips on column tiles of mixed length
    } else if (GenericEngine<T>::handles(n)) {
      gen_.reset(new GenericEngine<T>(n));       // what is left of 2^a*3^b, a < 12: one round trip per radix
    } else if (TiledMixedEngine<T>::handles_smooth(n) && !(n <= MixedEngine<T>::MAX_N && specialise_policy() >= 1 && specialised_route(n, specialise_policy() >= 2, nullptr))) {
      // factors 5 / 7 beyond the ahead-of-time LDS kernels: tile passes instead of Bluestein (round 5) -- unless the length fits a compute
      // unit's LDS and its own one-launch kernel is in the code-object cache (one HBM round trip instead of two)
      tiled_.reset(new TiledMixedEngine<T>(n));
    } else if (specialise_policy() >= 1 && specialised_route(n, specialise_policy() >= 2, nullptr)) {
      // prime factors up to 13 without an ahead-of-time route: the kernels of an earlier "specialise" from the on-disk cache
    } else {
      init_bluestein();
    }
    refresh_desc();
  }
  // A length the default routes send to Bluestein although its prime factors stop at 13: its own LDS kernel (<= MAX_N points) or
  // two / three column-tile passes, specialised at run time (rtc.cpp).  allow_compile = false: only where every kernel it needs is
  // in the code-object cache.  true: mix_ / tiled_ is set.
  bool specialised_route(size_t n, bool allow_compile, std::string* why) {
    std::string local;
    std::string& w = why ? *why : local;
    if (is_pow2(n) || n < 2) { w = "a power of two"; return false; }
    std::vector<uint32_t> radices;
    if (n <= MixedEngine<T>::MAX_N) {
      if (!MixedEngine<T>::factor(n, radices)) { w = "a prime factor above 13"; return false; }
      if (!allow_compile && !MixedEngine<T>::specialised_kernel_cached(n)) { w = "not in the code-object cache"; return false; }
      std::unique_ptr<MixedEngine<T>> m;
      try { m.reset(new MixedEngine<T>(n, true)); }
      catch (const EngineError& e) {
        if (e.status == ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY) throw;  // as try_mixed: out of memory is not "take the next route"
        (void)hipGetLastError(); w = e.what(); return false;
      }
      if (m->specialise(&w, allow_compile) != ::fourier::c::FOURIER_HIP_OK) return false;
      mix_ = std::move(m);
      return true;
    }
    if (n <= TiledMixedEngine<T>::MAX_N && !TiledMixedEngine<T>::factorise(n, true).empty()) {
      if (!allow_compile && !TiledMixedEngine<T>::specialised_kernels_cached(n)) { w = "not in the code-object cache"; return false; }
      try { tiled_.reset(new TiledMixedEngine<T>(n, true, allow_compile)); }
      catch (const EngineError& e) {
        tiled_.reset();
        if (e.status == ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY) throw;
        (void)hipGetLastError(); w = e.what(); return false;
      }
      return true;
    }
    w = "not a length whose prime factors stop at 13 with a kernel to specialise";
    return false;
  }
  // 2^a 3^b keep the LDS kernels on the reference's own schedule (bit-identical to the CPU restatement)
  static bool regfft_route(size_t n) {
    size_t p = n;
    while (p % 2 == 0) p /= 2;
    while (p % 3 == 0) p /= 3;
    return p != 1 && BluRegEngine<T>::has_direct(n);
  }
  static bool tiled_before_pow2_tiles(size_t n) {
    if (!Pow2Engine<T>::handles_mixed(n) || dev_env("FOURIER_POW2_TILES_FIRST") || !TiledMixedEngine<T>::handles(n, true)) return false;
    const std::vector<uint32_t> f = TiledMixedEngine<T>::factorise(n);
    return f.size() == 2 && f[0] <= 576 && f[1] <= 576;
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
    if (regf_) desc_ = "stockham " + regf_->describe();
    else if (mix_) desc_ = "stockham mixed-radix " + mix_->describe();
    else if (tiled_) desc_ = "stockham mixed tiles " + tiled_->describe();
    else if (gen_) desc_ = "stockham global-pass " + gen_->describe();
    else if (blur_) desc_ = "bluestein M=" + std::to_string(m_) + " " + blur_->describe();
    else if (blut_) desc_ = "bluestein M=" + std::to_string(m_) + " inner " + blut_->describe();
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
    if (regf_) return "registers_one_launch";
    if (mix_) return "mixed_radix";
    if (tiled_) { for (size_t p = 0; p < tiled_->num_passes(); ++p) d += std::string(d.empty() ? "" : ",") + "pass" + std::to_string(p); return d; }
    if (gen_) { for (size_t p = 0; p < gen_->num_passes(); ++p) d += std::string(d.empty() ? "" : ",") + "pass" + std::to_string(p); return d; }
    auto passes = [&](const char* tag) {
      for (size_t p = 0; p < (blu_ ? eng_->num_passes() : eng_->hbm_round_trips()); ++p) d += std::string(d.empty() ? "" : ",") + tag + std::to_string(p);
    };
    if (blur_) return "bluestein_one_launch";
    if (blut_) return "chirp_in_pass,conv_pass,chirp_out_pass";
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
    if (mix_ || regf_) return 2.0 * n_ * ELEM;
    if (tiled_) return 2.0 * n_ * ELEM * tiled_->num_passes();
    if (gen_) return 2.0 * n_ * ELEM * gen_->num_passes();
    if (!blu_) return 2.0 * n_ * ELEM * eng_->hbm_round_trips();
    // unfused: pre (n + table read, m write) + 2 inner FFTs + w table + post (n + table read, n write);
    // fused: the first / last inner pass read / write the n-point user array instead of an m-point sweep
    if (blut_) return (double)ELEM * (5.0 * m_ + 4.0 * n_);  // (n + n, m) + (m + m, m) + (m + n, n)
    if (small_fused_ || blur_) return (double)ELEM * 2.0 * n_;  // tables stay L2-resident
    const double chirp_reads = (chirp_compute_ ? 1.0 : 2.0) * n_;  // the n-entry chirp table: the chirp-out pass reads it, the chirp-in pass only without bluestein_chirp_compute
    if (fused_ && conv_) return (double)ELEM * (2.0 * m_ * (2.0 * eng_->num_passes() - 1.0) - 2.0 * (m_ - n_) + m_ + chirp_reads);
    // without the conv kernel the product with w is a sweep of its own over the M-point spectrum: 2 m on top of the w table
    if (fused_) return (double)ELEM * (2.0 * 2.0 * m_ * eng_->num_passes() - 2.0 * (m_ - n_) + 2.0 * m_ + m_ + chirp_reads);
    return (double)ELEM * ((2.0 * n_ + m_) + 2.0 * 2.0 * m_ * eng_->num_passes() + 2.0 * m_ + m_ + 3.0 * n_);
  }

  int set_option(const std::string& key, long long v) {
    DeviceGuard g(device_);  // bluestein_fusion may allocate tables: they must land on the plan's device
    if (key == "chunk_bytes" && v >= 0) { chunk_bytes_ = (size_t)v; return 0; }
    if (key == "scratch" && (v == 0 || v == 1)) { force_scratch_ = (v == 1); return 0; }
    if (key == "xcd_swizzle" && v >= 0 && v <= 4) { nxcd_ = v == 0 ? 1 : (8 | ((unsigned)(v - 1) << 8)); return 0; }
    // the general band walk of the tile passes (xcd_remap mode 4): v = tiles per band | transforms per group << 8 (0 = the XCD's
    // whole range) | transform-fastest << 19; v = 0 restores the default order
    if (key == "tile_walk_last" && v >= 0 && v < (1 << 21)) {  // the same encoding, for the last pass of a plain multi-pass plan alone (0 = as the others)
      const unsigned band = (unsigned)v & 0xff, group = ((unsigned)v >> 8) & 0x3ff, tf = ((unsigned)v >> 19) & 3;
      nxcd_last_ = band == 0 ? 0u : (8u | (4u << 8) | (band << 12) | (group << 20) | (tf << 30));
      return 0;
    }
    if (key == "tile_walk" && v >= 0 && v < (1 << 21)) {
      const unsigned band = (unsigned)v & 0xff, group = ((unsigned)v >> 8) & 0x3ff, tf = ((unsigned)v >> 19) & 3;  // bit 20: strided bands (A/B)
      nxcd_ = band == 0 ? 8u : (8u | (4u << 8) | (band << 12) | (group << 20) | (tf << 30));
      return 0;
    }
    // Bluestein's M: 1 (default) = the smallest product of two tile lengths where that saves a quarter of the power-of-two work array,
    // 0 = always the reference's next power of two (bluesteins.rs:110).  Rebuilds the plan's tables now; not while a transform is in flight.
    if (key == "bluestein_smooth_m" && v >= 0 && v <= 2) {  // (2: wherever a product of two tile lengths exists -- for measurements)
      if (!blu_) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      if ((int)v != smooth_m_mode_) rebuild_bluestein((int)v);
      return 0;
    }
    if (key == "bluestein_fusion" && (v == 0 || v == 1)) {
      if (blut_) return 0;  // the smooth-M route has no unfused form
      if (blur_) {          // nor has the register route: the unfused sweeps run on the reference's power-of-two M
        if (v == 1) return 0;
        rebuild_bluestein(0);
      }
      fused_ = (v == 1) && blu_ && eng_->can_fuse_bluestein();
      small_fused_ = (v == 1) && blu_ && eng_->enable_bluestein_small();
      return 0;
    }
    if (key == "bluestein_conv" && (v == 0 || v == 1)) { conv_ = (v == 1) && conv_ok_; return 0; }
    if (key == "bluestein_chirp_compute" && (v == 0 || v == 1)) {
      // (the computed chirp is built on exact exponents: under "bluestein_reference_chirp" it would disagree with the chirp-out table and w by the
      // reference's own angle error -- the request is remembered for when that option is switched off again, the pass keeps reading the table)
      if (reference_chirp_) { chirp_compute_saved_ = (v == 1) && chirp_p_.p != nullptr; return 0; }
      chirp_compute_ = (v == 1) && chirp_p_.p != nullptr;
      return 0;
    }
    if (key == "host_chunk_bytes" && v > 0) { host_chunk_bytes_ = (size_t)v; return 0; }
    // the two passes of a two-pass power-of-two plan software-pipelined over two internal streams with the intermediate in a small
    // ring (Pow2Engine::run_pipelined): v = transforms per chunk | ring slots << 16 | one-stream control << 24; 0 = off
    if (key == "stream_pipeline" && v >= 0 && v < (1 << 25)) {
      if (v != 0 && (blu_ || !eng_ || !eng_->can_pipeline())) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      pipe_chunk_ = (size_t)(v & 0xffff); pipe_slots_ = (size_t)((v >> 16) & 0xff); pipe_one_stream_ = ((v >> 24) & 1) != 0;
      if (pipe_chunk_ && pipe_slots_ < 2) pipe_slots_ = 2;
      return 0;
    }
    // the reference's unreduced chirp angle (build_chirp_tables): rebuilds the x and w tables now; the chirp-in pass then READS the
    // table (its computed chirp is built on exact exponents).  Not while a transform is in flight on this handle.
    if (key == "bluestein_reference_chirp" && (v == 0 || v == 1)) {
      if (!blu_) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      if ((v == 1) != reference_chirp_) {
        HIP_CHECK(hipDeviceSynchronize());  // the tables are replaced in place
        build_chirp_tables(v == 1);
        reference_chirp_ = (v == 1);
        if (reference_chirp_) { chirp_compute_saved_ = chirp_compute_; chirp_compute_ = false; }
        else chirp_compute_ = chirp_compute_saved_;
      }
      return 0;
    }
    // "register_stages" = 1: a 2^a 3^b length that runs the LDS kernel on the reference's own schedule (bit-identical to the CPU restatement)
    // takes the register-stage kernel regfft_shapes.h lists for it instead (kernels_regfft.h: within the tolerance, not the bits; 1.04 ... 1.6 x);
    // 0 brings the default back.  UNSUPPORTED where no such kernel is listed; OK unchanged on a plan that runs register stages by default.
    if (key == "register_stages" && (v == 0 || v == 1)) {
      if (v == 1) {
        if (regf_) return ::fourier::c::FOURIER_HIP_OK;
        if (!mix_ || !BluRegEngine<T>::has_direct_on_request(n_)) return ::fourier::c::FOURIER_HIP_UNSUPPORTED;
        try { regf_.reset(new BluRegEngine<T>(n_, (uint32_t)n_, true, 100)); } catch (const EngineError& e) { return e.status; }
        regf_on_request_ = true;
      } else if (regf_on_request_) {
        regf_.reset();
        regf_on_request_ = false;
      }
      refresh_desc();
      return ::fourier::c::FOURIER_HIP_OK;
    }
    // "specialise" = 1: compile this length's own LDS mixed-radix kernel with hipRTC (about a second, now) and run it from the
    // next call on -- for a length whose prime factors stop at 13, that fits a compute unit's LDS and has no ahead-of-time
    // per-length kernel (it runs the runtime-parameterised kernel, or Bluestein beyond that kernel's reach).  A plan that
    // already runs a per-length kernel returns OK unchanged; UNSUPPORTED where hipRTC is not available or the length is not of
    // that family: the plan keeps its route.
    if (key == "specialise" && v == 1) {
      std::string why;
      auto report = [&](int st) {
        if (st != ::fourier::c::FOURIER_HIP_OK && getenv("FOURIER_HIP_VERBOSE")) fprintf(stderr, "libfourier: specialise(%zu): %s\n", n_, why.c_str());
        return st;
      };
      if (regf_) return ::fourier::c::FOURIER_HIP_OK;  // already on a kernel of its own
      if (mix_) { const int st = mix_->specialise(&why); refresh_desc(); return report(st); }
      if (blu_) {  // exec() takes the new route from here on; the Bluestein tables stay allocated but unused
        if (specialised_route(n_, true, &why)) { refresh_desc(); return ::fourier::c::FOURIER_HIP_OK; }
        return report(::fourier::c::FOURIER_HIP_UNSUPPORTED);
      }
      if (tiled_) {
        // ahead-of-time tile passes: where the length fits a compute unit's LDS its own one-launch kernel replaces them (one HBM round trip
        // instead of two); otherwise there is nothing to specialise and the plan stays as it is (OK)
        if (n_ <= MixedEngine<T>::MAX_N && !tiled_->specialised()) {
          std::unique_ptr<TiledMixedEngine<T>> keep = std::move(tiled_);
          if (specialised_route(n_, true, &why) && mix_) { refresh_desc(); return ::fourier::c::FOURIER_HIP_OK; }
          tiled_ = std::move(keep);
          mix_.reset();
          return report(::fourier::c::FOURIER_HIP_UNSUPPORTED);
        }
        return ::fourier::c::FOURIER_HIP_OK;
      }
      why = "not a length whose prime factors stop at 13 with a kernel to specialise";
      return report(::fourier::c::FOURIER_HIP_UNSUPPORTED);
    }
    // LAST pass as persistent workgroups that prefetch their next tile (fft_last_prefetch_kernel; experiments library only,
    // measured slower): 1 where the kernel exists, INVALID_ARGUMENT where it does not
    if (key == "last_pass_prefetch" && (v == 0 || v == 1) && eng_) {
      if (v == 1 && !eng_->has_prefetch_last() && !(eng_inv_ && eng_inv_->has_prefetch_last())) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      eng_->set_prefetch_last(v == 1);
      if (eng_inv_) eng_inv_->set_prefetch_last(v == 1);
      return 0;
    }
    // the passes as their load / store skeleton (experiments library only; timing tool of bench.py's streaming ceiling, wrong results)
    if (key == "skeleton" && (v == 0 || v == 1) && eng_ && !blu_) {
      if (v == 1 && !eng_->has_skeleton()) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
      eng_->set_skeleton(v == 1);
      return 0;
    }
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
    if (mix_ || regf_ || batch == 0) return batch;
    if (tiled_) {  // one scratch of a chunk for in-place calls and three-pass plans
      size_t chunk = batch;
      if (chunk_bytes_) chunk = std::max<size_t>(1, std::min<size_t>(batch, chunk_bytes_ / (n_ * ELEM)));
      while (chunk > 1 && (double)chunk * (double)n_ / 8.0 > 2.0e9) chunk = (chunk + 1) / 2;
      if (!(tiled_->needs_scratch(in_place) || force_scratch_)) return chunk;
      for (;;) {
        try { scratch_.ensure(chunk * n_ * ELEM); return chunk; }
        catch (const EngineError& e) {
          if (e.status != ::fourier::c::FOURIER_HIP_OUT_OF_MEMORY || chunk <= 1) throw;
          (void)hipGetLastError();
          chunk = (chunk + 1) / 2;
        }
      }
    }
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
      if (pipe_chunk_) { eng_->reserve_pipeline(std::min(pipe_chunk_, batch), pipe_slots_); return batch; }
      if (eng_->l2fused_enabled()) { eng_->reserve_l2fused(chunk); return chunk; }
      const bool need = eng_->needs_scratch(in_place) || (force_scratch_ && eng_->num_passes() >= 2);
      if (need) reserve([&](size_t c) { scratch_.ensure(c * n_ * ELEM); });
      return chunk;
    }
    if (small_fused_ || blur_) return batch;  // whole chirp-z in one launch: no work array
    reserve([&](size_t c) {
      work_.ensure(c * m_ * ELEM);
      if (blut_ || eng_->needs_scratch(true) || fused_) scratch_.ensure(c * m_ * ELEM);
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
    if (regf_) {  // the whole transform in registers: one launch, in place allowed (a workgroup reads its transforms first)
      regf_->run(in, out, batch, nullptr, nullptr, inverse, scale, stream, prof);
      return;
    }
    if (mix_) {  // every pass stays in LDS: one launch, in place allowed (a workgroup reads its transforms first)
      const bool scaled = code == ::fourier::c::FOURIER_TRANSFORM_IFFT || code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_FFT ||
                          code == ::fourier::c::FOURIER_TRANSFORM_SQRT_SCALED_IFFT;  // mod.rs:381-385
      mix_->run(in, out, batch, !inverse, scaled,
                scale, stream, prof);
      return;
    }
    const size_t chunk = prepare(batch, in_place);

    if (tiled_) {
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        tiled_->run(in + b0 * n_, out + b0 * n_, (cpx<T>*)scratch_.p, nb, inverse, scale, stream, prof);
      }
      return;
    }
    if (gen_) {
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        gen_->run(in + b0 * n_, out + b0 * n_, (cpx<T>*)scratch_.p, nb, inverse, scale, stream, prof);
      }
      return;
    }
    if (!blu_) {
      if (pipe_chunk_ && (!prof || in_place)) {  // (a profiled out-of-place call times the kernels one by one on the caller's stream)
        eng_->run_pipelined(in, out, batch, pipe_chunk_, pipe_slots_, inverse, scale, stream, nxcd_, nxcd_last_, pipe_one_stream_);
        return;
      }
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        eng_->run(in + b0 * n_, out + b0 * n_, (cpx<T>*)scratch_.p, nb, inverse, scale, nullptr, force_scratch_, stream, prof, 0, nxcd_,
                  typename Pow2Engine<T>::BluIO(), nxcd_last_);
      }
      return;
    }
    // Bluestein (bluesteins.rs:215-259): work = x.in (zero padded) ; FFT_M ; .w ; IFFT_M ; out = work.x.scale
    if (small_fused_) {  // M <= 2^15: the whole chirp-z in one launch, no work array
      eng_->run_bluestein_small(in, out, batch, xtab_.p, wtab_.p, n_, inverse, scale, stream, prof, nxcd_);
      return;
    }
    if (blur_) {  // a short transform on a smooth M = R1 x R2: one launch, both transforms in registers (kernels_chirpz.h)
      blur_->run(in, out, batch, xtab_.p, wtab_.p, inverse, scale, stream, prof);
      return;
    }
    cpx<T>* work = (cpx<T>*)work_.p;
    if (blut_) {  // smooth M: three sweeps on register tiles (kernels_regtile.h)
      for (size_t b0 = 0; b0 < batch; b0 += chunk) {
        const size_t nb = std::min(chunk, batch - b0);
        blut_->run(in + b0 * n_, out + b0 * n_, work, (cpx<T>*)scratch_.p, nb, xtab_.p, wtab_.p, inverse, scale, stream, prof);
      }
      return;
    }
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
      FOURIER_LAUNCH(get_blu_kernel(Real<T>{}, 0), elementwise_grid(nb * m_), 256, 0, stream, pre);
      PROF_END(prof);
      eng_->run(work, work, (cpx<T>*)scratch_.p, nb, false, 1.0, (const cpx<T>*)wtab_.p, false, stream, prof, 1, nxcd_);
      eng_->run(work, work, (cpx<T>*)scratch_.p, nb, true, 1.0, nullptr, false, stream, prof, 1 + np, nxcd_);
      BluArgs post{work, out + b0 * n_, xtab_.p, (uint64_t)n_, (uint64_t)m_, (uint64_t)nb, inverse, scale};
      PROF_BEGIN(prof, 1 + 2 * np);
      FOURIER_LAUNCH(get_blu_kernel(Real<T>{}, 1), elementwise_grid(nb * n_), 256, 0, stream, post);
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

  // option "bluestein_smooth_m" (and "bluestein_fusion" = 0 on the register route): the Bluestein route chosen again under another rule
  void rebuild_bluestein(int mode) {
    HIP_CHECK(hipDeviceSynchronize());
    smooth_m_mode_ = mode;
    blur_.reset(); blut_.reset(); eng_.reset(); eng_inv_.reset();
    init_bluestein();
    if (reference_chirp_) { build_chirp_tables(true); chirp_compute_saved_ = chirp_compute_; chirp_compute_ = false; }
    refresh_desc();
  }
  void init_bluestein() {
    blu_ = true;
    if (n_ > ((size_t)1 << 26)) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "Bluestein sizes above 2^26 are not supported");
    m_ = 1;
    while (m_ < 2 * n_ - 1) m_ <<= 1;  // bluesteins.rs:110
    if (2 * n_ > m_) throw EngineError(::fourier::c::FOURIER_HIP_RUNTIME_ERROR, "Bluestein: M < 2N");  // the fused end passes rely on it
    fused_ = small_fused_ = conv_ = conv_ok_ = chirp_compute_ = false;
    // M need only reach 2N - 1 (bluesteins.rs:110 rounds up to a power of two: up to 4N).  Beyond the one-launch kernels (M <= 2^15, f64 2^14)
    // the work array is swept three times: a product of two register-tile lengths instead, where the power-of-two array is at least 1.6 x
    // longer -- f32 +3 ... 43 %, f64 +20 ... 51 % there -- and in f64 from 1.44 x on while the conv kernel stays within 336 points (+5 ... 11 %;
    // beyond: -8 %).  Below that the power-of-two tiles' higher rate wins (f32 1.49 x: -7 ... -2 %; 1.35 x: -5 ... -1 %; f64 1.35 x: 0 ... +3 %;
    // profiles/r06_s35_smooth_m_late_loads_*.jsonl)
    // A short transform (M up to 1024 points, f32 1152): the whole chirp-z in one launch on M = R1 x R2 with both transforms in registers
    // (kernels_chirpz.h).  f64: wherever such an M exists -- 1.15 ... 2.05 x the power-of-two one-launch kernels, 1.2 - 1.3 x at the SAME M (64, 256,
    // 1024).  f32 (two transforms per lane, packed arithmetic): where M is at least 1.1 x shorter (1.1 ... 1.55 x up to M = 784; 480 against 512
    // and the stages of 30 and 32 points against 1024: 0.88 ... 0.99), up to M = 64 regardless (1.2 x at 64 against 64), 1152 against 2048
    // (1.37 x) -- profiles/r06_s45_chirpz_reg_ab.jsonl
    if (smooth_m_mode_ != 0 && !dev_env("FOURIER_NO_SMOOTH_M")) {
      const uint32_t mr = BluRegEngine<T>::choose_m(n_);
      // Three stages (M = R1 x R2 x R3, 1296 ... 3072 and 8820 / 9261): where M is at least 1.25 x shorter -- f32 1.13 ... 1.6 x, f64 1.05 ... 1.55 x
      // the power-of-two one-launch kernels of M = 2048, 4096, 16384 (profiles/r06_s46_chirpz_reg3_ab.jsonl)
      const bool pays = mr > 1152 ? 5 * (uint64_t)mr <= 4 * (uint64_t)m_
                                  : (sizeof(T) == 8 || mr <= 64 || (mr <= 800 && 11 * (uint64_t)mr <= 10 * (uint64_t)m_) || 17 * (uint64_t)mr <= 10 * (uint64_t)m_);
      if (mr != 0 && (smooth_m_mode_ == 2 || pays)) {
        m_ = mr;
        blur_.reset(new BluRegEngine<T>(n_, mr));
        build_chirp_tables(false);
        return;
      }
    }
    if (smooth_m_mode_ != 0 && !dev_env("FOURIER_NO_SMOOTH_M") && m_ > ((size_t)1 << (sizeof(T) == 4 ? 15 : 14))) {
      uint32_t l1 = 0, l2 = 0;
      const uint64_t ms = BluTiledEngine<T>::choose_m(n_, l1, l2);
      const bool pays = smooth_m_mode_ == 2 || 8 * ms <= 5 * (uint64_t)m_ || (sizeof(T) == 8 && 36 * ms <= 25 * (uint64_t)m_ && l2 <= 336);
      if (ms != 0 && pays) {
        m_ = ms;
        blut_.reset(new BluTiledEngine<T>(n_, l1, l2));
        build_chirp_tables(false);
        return;
      }
    }
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
    build_chirp_tables(false);
    const uint64_t two_n = 2 * (uint64_t)n_;
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
  }
  // The chirp x[k] = exp(-i*pi*k^2/n) (bluesteins.rs:51-61) and w = FFT_M(conj chirp, mirrored) (bluesteins.rs:18-48), evaluated in f64 on
  // the host and cast (twiddle.rs:7-19 style), the inner IFFT's 1/M (bluesteins.rs:239 -> mod.rs:383) folded into w.
  // reference_angle = false (default): the angle is reduced exactly, k^2 mod 2n, before the trigonometry -- the reference leaves it
  // unreduced (bluesteins.rs:10,31,57: theta = k^2 * pi / n in f64, up to pi * n radians), which costs it about n * 1e-16 of
  // absolute angle error (2e-10 at n = 10^6: invisible in f32, the whole error budget in f64).  true (plan option
  // "bluestein_reference_chirp"): the reference's own expression, for a caller who wants the reference's f64 RESULTS rather than the
  // exact DFT: the engine then agrees with the CPU restatement to f64 rounding instead of to 1e-9 (tests: test_bluestein_reference_chirp).
  void build_chirp_tables(bool reference_angle) {
    std::vector<double> cr(n_), ci(n_);
    const uint64_t two_n = 2 * (uint64_t)n_;
    for (size_t k = 0; k < n_; ++k) {
      double ang;
      if (reference_angle) {
        const double index = (double)k * (double)k;  // (i as f64).powi(2), bluesteins.rs:31,57
        ang = index * M_PI / (double)n_;             // bluesteins.rs:10
      } else {
        const uint64_t r = (uint64_t)(((unsigned __int128)k * k) % two_n);
        ang = M_PI * (double)r / (double)n_;
      }
      cr[k] = std::cos(ang); ci[k] = -std::sin(ang);
    }
    std::vector<cpx<T>> x(n_);
    for (size_t k = 0; k < n_; ++k) x[k] = {(T)cr[k], (T)ci[k]};  // x_fwd, bluesteins.rs:51-61
    xtab_.upload(x);
    std::vector<double> wr(m_, 0.0), wi(m_, 0.0);
    for (size_t k = 0; k < n_; ++k) {
      wr[k] = cr[k]; wi[k] = -ci[k];
      if (k) { wr[m_ - k] = cr[k]; wi[m_ - k] = -ci[k]; }
    }
    host_fft_any(wr, wi);
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
  std::unique_ptr<BluTiledEngine<T>> blut_;       // Bluestein on a smooth M (then eng_ is empty)
  std::unique_ptr<BluRegEngine<T>> blur_;         // ... of a short transform: one launch, transforms in registers (then eng_ is empty)
  std::unique_ptr<BluRegEngine<T>> regf_;         // a length with factors 5 ... 13 as a direct transform on the same register stages
  bool regf_on_request_ = false;                  // ... of a 2^a 3^b length, by plan option "register_stages" (mix_ stays allocated)
  int smooth_m_mode_ = 1;                         // option "bluestein_smooth_m"
  std::unique_ptr<MixedEngine<T>> mix_;
  std::unique_ptr<TiledMixedEngine<T>> tiled_;  // 2^a*3^b, a < 12, beyond the LDS kernels: column tiles of mixed length
  std::unique_ptr<GenericEngine<T>> gen_;  // ... and what has no tile factorisation: one global pass per radix
  DevBuf xtab_, wtab_;
  DevBuf chirp_p_, chirp_u_, tn_lo_, tn_hi_;  // chirp-in pass computing the chirp (init_bluestein)
  uint32_t tn_bits_ = 0;
  bool reference_chirp_ = false, chirp_compute_saved_ = false;  // option "bluestein_reference_chirp"
  bool chirp_compute_ = false;  // option "bluestein_chirp_compute"
  mutable DevBuf scratch_, work_, hostio_;
  mutable PinnedBuf pinned_;
  mutable hipStream_t legacy_stream_ = nullptr;  // legacy host-buffer calls (exec_host)
  size_t chunk_bytes_ = 0;
  size_t host_chunk_bytes_ = HOST_CHUNK_BYTES;  // exec_host_batch: bytes of one streamed chunk
  bool force_scratch_ = false;
  size_t pipe_chunk_ = 0, pipe_slots_ = 0;  // option "stream_pipeline"
  bool pipe_one_stream_ = false;
  bool fused_ = false;  // Bluestein: chirp steps fused into the inner passes
  bool small_fused_ = false;  // Bluestein with M <= 2^15: everything in one launch
  bool conv_ = false, conv_ok_ = false;  // Bluestein: forward LAST + (.)w + inverse FIRST in one launch
  unsigned nxcd_ = 8, nxcd_last_ = 0;
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
