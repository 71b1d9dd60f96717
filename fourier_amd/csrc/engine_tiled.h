// engine_tiled.h -- 2^a * 3^b with a < 12 beyond one compute unit's LDS as two or three big-radix passes of mixed length
// on column tiles (kernels_tiled.h): N = L1 x L2 (x L3), every factor a pass length with a kernel.  The reference runs these
// lengths natively, one small radix per sweep of memory (autosort/mod.rs:104-116, 203-284).
#pragma once
#include "engine_common.h"
#include "mixed_schedule.h"

namespace fourier_hip {

// workgroup -> tile order of the mixed-length tile passes (xcd_chunked, kernels_common.h): runs of 8 neighbouring tiles per XCD where row
// segments straddle 128-byte lines (+3 ... 15 %; 2 / 4 / 8 / 32 within 2 % of each other), the plain order where they do not (chunks there:
// -3 ... +1 %) -- profiles/r06_s24_regtile_chunk_ab.jsonl
constexpr uint32_t FOURIER_TILE_CHUNK = 8, FOURIER_TILE_CHUNK_ALIGNED = 0;

template <typename T> class TiledMixedEngine {
 public:
  static constexpr size_t MAX_N = (size_t)1 << 26;
  // every length with an ahead-of-time kernel: prime factors up to 7 (kernels_regtile.cpp, kernels_tiled.cpp); without the register-tile kernels
  // (FOURIER_NO_REGTILE: experiments library, emulator) the LDS kernels' lengths alone, up to 512 points
  static const std::vector<uint32_t>& menu() {
    static const std::vector<uint32_t> with_reg = [] {
      std::vector<uint32_t> v;
      for (uint32_t L = 64; L <= max_len(false); ++L)
        if (get_regtile_kernel(Real<T>{}, L).fn || get_tiled_kernel(Real<T>{}, L).fn) v.push_back(L);
      return v;
    }();
    static const std::vector<uint32_t> lds_only = [] {
      std::vector<uint32_t> v;
      for (uint32_t L = 64; L <= 512; ++L)
        if (get_tiled_kernel(Real<T>{}, L).fn) v.push_back(L);
      return v;
    }();
    return dev_env("FOURIER_NO_REGTILE") ? lds_only : with_reg;
  }
  // longest tile pass: 512 points ahead of time; 1024 for a kernel compiled at run time (a 16-column f32 / 8-column f64 tile of 1024 rows is
  // 128 KiB of LDS, one 1024-thread workgroup per CU: 2.4 - 3.2 TB/s per pass against 4.2 - 4.9 for the short tiles -- level with a three-pass
  // plan of short tiles, but it reaches lengths that have no split into factors of 64 ... 512 at all: 5^8 = 625 x 625 15 % of the HBM peak
  // against 9 % as Bluestein, 500000 = 800 x 625 18 % against 11 %, profiles/r05_s21_long_tiles_ab.jsonl)
  // (ahead of time: 512 points for the LDS kernels; 28 lengths of 513 ... 1024 points run on register tiles of 64-byte rows, round 6;
  // f32: five more with a 40- or 35-point stage -- 875, 945, 972, 980, 1000 --, 10^6 = 1000 x 1000 0.20 -> 0.29, profiles/r06_s42_1000_point_tiles.jsonl)
  // 390625 = 625 x 625: 0.09 (Bluestein) -> 0.26; 500000 = 800 x 625: 0.11 -> 0.24; 640000, 729000 (three passes of 80 ... 100 points before):
  // +28 ... 52 % (profiles/r06_s40_long_tiles_ab.jsonl)
  static uint32_t max_len(bool rtc) { (void)rtc; return 1024u; }
  // every length a tile pass can have once it is compiled at run time (plan option "specialise"): prime factors up to 13
  static const std::vector<uint32_t>& menu_rtc() {
    static const std::vector<uint32_t> m = [] {
      std::vector<uint32_t> v;
      for (uint32_t L = 64; L <= max_len(true); ++L) {
        uint32_t r = L;
        for (uint32_t p : {2u, 3u, 5u, 7u, 11u, 13u})
          while (r % p == 0) r /= p;
        if (r == 1) v.push_back(L);
      }
      return v;
    }();
    return m;
  }
  // pass lengths, most balanced factorisation first: two factors if there is one, else three; empty: none
  static std::vector<uint32_t> factorise(size_t n, bool rtc = false) {
    std::vector<uint32_t> best;
    uint32_t best_max = 0xffffffffu;
    const auto& m = rtc ? menu_rtc() : menu();
    for (uint32_t a : m) {
      if (n % a) continue;
      const size_t r = n / a;
      if (r <= max_len(rtc) && r >= 64 && r <= a && std::find(m.begin(), m.end(), (uint32_t)r) != m.end() && a < best_max) {
        best = {a, (uint32_t)r};
        best_max = a;
      }
    }
    if (!best.empty()) return best;
    for (uint32_t a : m) {
      if (n % a) continue;
      for (uint32_t b : m) {
        if (b > a || (n / a) % b) continue;
        const size_t r = n / a / b;
        if (r > b || r < 64 || std::find(m.begin(), m.end(), (uint32_t)r) == m.end()) continue;
        if (a < best_max) { best = {a, b, (uint32_t)r}; best_max = a; }
      }
    }
    return best;
  }
  // any_a: also lengths with a >= 12, which otherwise run as power-of-two tiles + odd passes (A/B: FOURIER_TILED_FIRST)
  static bool handles(size_t n, bool any_a = false) {
    if (n < 4096 || n > MAX_N || dev_env("FOURIER_NO_TILED_MIXED")) return false;
    size_t p = n;
    while (p % 3 == 0) p /= 3;
    if (!is_pow2(p) || p == n || (p >= 4096 && !any_a)) return false;  // 2^a * 3^b, b >= 1, a < 12
    return !factorise(n).empty();
  }

  // Lengths with a factor 5 or 7 and no prime factor above 7 beyond the whole-transform LDS kernels (the reference: Bluestein,
  // fourier/src/lib.rs:38-42): two or three tile passes over the ahead-of-time menu -- 10^5 = 400 x 250, 44100 = 210 x 210, 10^6 =
  // 100 x 100 x 100 (round 5; until then only under the plan option "specialise")
  static bool handles_smooth(size_t n) {
    if (n < 4096 || n > MAX_N || dev_env("FOURIER_NO_TILED_MIXED") || dev_env("FOURIER_NO_TILED_SMOOTH")) return false;
    size_t p = n;
    for (size_t q : {2, 3, 5, 7})
      while (p % q == 0) p /= q;
    if (p != 1 || (n % 5 != 0 && n % 7 != 0)) return false;
    return !factorise(n).empty();
  }

  // the ahead-of-time kernel of a tile pass of length L: the register-resident one (kernels_regtile.h) where the length splits into two
  // factors of at most 32, else the LDS one (kernels_tiled.h); FOURIER_NO_REGTILE (experiments library, emulator): always the latter
  static TiledKernel pass_kernel(uint32_t L) {
    if (!dev_env("FOURIER_NO_REGTILE")) {
      const TiledKernel k = get_regtile_kernel(Real<T>{}, L);
      if (k.fn) return k;
    }
    return get_tiled_kernel(Real<T>{}, L);
  }
  // launch shape of a tile pass of length L for a kernel compiled at run time: tiled_shape (mixed_schedule.h), the function the
  // kernel's own TiledCfg is built from
  static TiledKernel shape_of(uint32_t L) {
    const TileShape t = tiled_shape(L, (uint32_t)sizeof(cpx<T>));
    TiledKernel k;
    k.L = L; k.cols = t.cols; k.threads = t.threads; k.smem = t.smem;
    return k;
  }
  // rtc: plan option "specialise" -- the tile lengths may have prime factors up to 13, and a length without an ahead-of-time
  // kernel is compiled with hipRTC (rtc.cpp); throws UNSUPPORTED where that is not possible
  // every tile length of the run-time factorisation of n has an ahead-of-time kernel or a cached specialised one
  static bool specialised_kernels_cached(size_t n) {
    const std::vector<uint32_t> lens = factorise(n, true);
    for (uint32_t L : lens)
      if (!get_tiled_kernel(Real<T>{}, L).fn && !rtc_cached(sizeof(T) == 8, L, shape_of(L).smem, true)) return false;
    return !lens.empty();
  }
  explicit TiledMixedEngine(size_t n, bool rtc = false, bool allow_compile = true) : n_(n) {
    const std::vector<uint32_t> lens = factorise(n, rtc);
    if (lens.empty()) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no tile factorisation");
    uint64_t s = 1, size = n;
    for (uint32_t L : lens) {
      Pass ps;
      ps.k = pass_kernel(L);
      if (!ps.k.fn) {
        if (!rtc) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no tile kernel of this length");
        ps.k = shape_of(L);
        std::string why;
        if (!rtc_mixed_kernel(sizeof(T) == 8, L, ps.k.smem, ps.rtc, why, true, allow_compile))
          throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "tile pass of length " + std::to_string(L) + ": " + why);
        ps.k.smem = 0;  // declared statically by the specialised kernel
        specialised_ = true;
      }
      ps.s = s; ps.m = size / L;
      {  // row segments that straddle 128-byte lines (a transform, an input row or an output row that does not start on one): neighbouring
         // tiles in one L2 (xcd_chunked)
        const uint64_t e = sizeof(cpx<T>);
        const bool straddle = (n * e) % 128 != 0 || (ps.s * ps.m * e) % 128 != 0 || (ps.s > 1 && (ps.s * e) % 128 != 0);
        ps.xcd_chunk = straddle ? FOURIER_TILE_CHUNK : FOURIER_TILE_CHUNK_ALIGNED;
      }
      if (ps.k.fn) raise_smem_limit((const void*)ps.k.fn, ps.k.smem);
      // tables of the in-tile transform: the reference's layout for a plan of length L (mod.rs:24-46), f64 trig then cast
      auto it = tables_.find(L);
      if (it == tables_.end()) {
        std::vector<cpx<T>> tw;
        size_t cur = L;
        if (ps.k.r1) {  // register-resident kernel: the twiddle between its two stages, W_L^{j2 * k1} as [k1 < r1][j2 < r2]
          for (size_t k1 = 0; k1 < ps.k.r1; ++k1)
            for (size_t j2 = 0; j2 < ps.k.r2; ++j2) {
              double re, im;
              unit_root(j2 * k1, L, re, im);
              tw.push_back({(T)re, (T)im});
            }
          cur = 1;
        }
        while (cur > 1) {
          const size_t R = mix_next_radix(L, (uint32_t)cur, cur == L);
          const size_t mm = cur / R;
          for (size_t i = 0; i < mm; ++i) {
            tw.push_back({(T)1, (T)0});
            for (size_t j = 1; j < R; ++j) tw.push_back(ref_twiddle(i * j, cur));
          }
          cur /= R;
        }
        auto buf = std::unique_ptr<DevBuf>(new DevBuf());
        buf->upload(tw);
        it = tables_.emplace(L, std::move(buf)).first;
      }
      ps.tw = it->second.get();
      if (ps.m > 1) {  // inter-pass twiddle W_size^{i*k}, two-level
        const int lb = (ilog2(size) + 1) / 2;
        ps.lo_bits = (uint32_t)lb;
        std::vector<cpx<T>> lo((size_t)1 << lb), hi((size_t)(size >> lb) + 1);
        for (size_t e = 0; e < lo.size(); ++e) { double re, im; unit_root(e, size, re, im); lo[e] = {(T)re, (T)im}; }
        for (size_t h = 0; h < hi.size(); ++h) { double re, im; unit_root((uint64_t)h << lb, size, re, im); hi[h] = {(T)re, (T)im}; }
        ps.tw_lo.reset(new DevBuf()); ps.tw_hi.reset(new DevBuf());
        ps.tw_lo->upload(lo); ps.tw_hi->upload(hi);
      }
      passes_.push_back(std::move(ps));
      s *= L; size /= L;
    }
  }
  size_t num_passes() const { return passes_.size(); }
  std::string describe() const {
    std::string d;
    for (const Pass& p : passes_) d += (d.empty() ? "" : "x") + std::to_string(p.k.L);
    if (specialised_) d += " specialised";
    return d;
  }
  bool specialised() const { return specialised_; }
  // scratch: batch * n elements; needed by in-place calls and by three-pass plans
  bool needs_scratch(bool in_place) const { return in_place || passes_.size() >= 3; }
  void run(const cpx<T>* in, cpx<T>* out, cpx<T>* scratch, size_t batch, bool inverse, double scale, hipStream_t stream, Profiler* prof) const {
    if (batch == 0) return;
    const size_t np = passes_.size();
    const bool in_place = ((const void*)in == (const void*)out);
    // every pass but the last is out of place; the last (same columns in and out) may run in place.  Two passes: in -> out ->
    // out, or in -> scratch -> out for an in-place call; three passes: in -> scratch -> out -> out (the input is dead after the first pass)
    const cpx<T>* src = in;
    for (size_t p = 0; p < np; ++p) {
      const Pass& ps = passes_[p];
      cpx<T>* dst = out;
      if (p + 1 < np) dst = (np == 2) ? (in_place ? scratch : out) : (p == 0 ? scratch : out);
      TiledArgs a;
      std::memset(&a, 0, sizeof(a));
      a.in = src; a.out = dst; a.tw = ps.tw->p;
      a.tw_lo = ps.m > 1 ? ps.tw_lo->p : nullptr; a.tw_hi = ps.m > 1 ? ps.tw_hi->p : nullptr; a.lo_bits = ps.lo_bits;
      a.n = n_; a.s = ps.s; a.m = ps.m;
      const uint64_t columns = ps.s == 1 ? ps.m : ps.s;
      a.tiles_per_row = (columns + ps.k.cols - 1) / ps.k.cols;
      a.swap_in = (p == 0) && inverse; a.swap_out = (p + 1 == np) && inverse;
      a.scale = (p + 1 == np) ? scale : 1.0;
      a.xcd_chunk = ps.xcd_chunk;
      const cpx<T> w3 = ref_twiddle(1, 3), w8 = ref_twiddle(1, 8);
      a.w3re = w3.re; a.w3im = w3.im; a.w8re = w8.re; a.w8im = w8.im;
      const uint64_t grid = (uint64_t)batch * a.tiles_per_row * (ps.s == 1 ? 1 : ps.m);
      if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
      PROF_BEGIN(prof, (int)p);
#ifndef FOURIER_EMU
      if (ps.rtc.fn) {  // a tile length compiled at run time: a module function
        void* params[] = {&a};
        HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)ps.rtc.fn, (unsigned)grid, 1, 1, ps.k.threads, 1, 1, 0, stream, params, nullptr));
      } else
#endif
      {
        FOURIER_LAUNCH(ps.k.fn, grid, ps.k.threads, ps.k.smem, stream, a);
      }
      PROF_END(prof);
      src = dst;
    }
  }

 private:
  // twiddle.rs:7-19: theta = (index*2) as f64 * PI / size as f64; (cos, -sin) cast to T
  static cpx<T> ref_twiddle(size_t index, size_t size) {
    const double theta = (double)(index * 2) * M_PI / (double)size;
    return {(T)std::cos(theta), (T)(-std::sin(theta))};
  }
  struct Pass {
    TiledKernel k;
    RtcKernel rtc;  // set instead of k.fn for a length compiled at run time
    uint64_t s = 1, m = 1;
    uint32_t lo_bits = 0, xcd_chunk = 0;
    DevBuf* tw = nullptr;
    std::unique_ptr<DevBuf> tw_lo, tw_hi;
  };
  size_t n_;
  bool specialised_ = false;  // at least one pass runs a kernel compiled at run time
  std::vector<Pass> passes_;
  std::map<uint32_t, std::unique_ptr<DevBuf>> tables_;
};

// Bluestein's inner transforms on a SMOOTH M = L1 x L2 (kernels_regtile.h): the reference pads to the next power of two
// (bluesteins.rs:110: M >= 2N - 1 is all the algorithm needs), up to 4N; here the smallest product of two register-tile lengths, three
// sweeps -- chirp-in first pass (L1), conv (last forward + (.) w + first inverse pass, L2), chirp-out last pass (L1).
template <typename T> class BluTiledEngine {
 public:
  // lengths with all three Bluestein kernels
  static const std::vector<uint32_t>& menu() {
    static const std::vector<uint32_t> m = [] {
      std::vector<uint32_t> v;
      for (uint32_t L = 64; L <= 512; ++L)
        if (get_regtile_kernel(Real<T>{}, L, 1).fn) v.push_back(L);
      return v;
    }();
    return m;
  }
  // M = L1 x L2 >= 2n - 1, L1 >= L2 (the conv kernel, the heavier one, at the shorter length): the smallest product -- or one up to 4 % longer
  // whose lengths split more evenly into their two register stages: the conv kernel at 243 = 27 x 9 takes 0.83 ms where 240 = 16 x 15 would
  // take 0.55 (profiles/r06_s31_smooth_m_pruned_ab.jsonl).  0: no product of two tile lengths reaches 2n - 1
  static uint64_t choose_m(size_t n, uint32_t& l1, uint32_t& l2) {
    const uint64_t need = 2 * (uint64_t)n - 1;
    uint64_t smallest = 0;
    for (uint32_t a : menu())
      for (uint32_t b : menu()) {
        if (b > a) break;
        const uint64_t m = (uint64_t)a * b;
        if (m < need) continue;
        if (smallest == 0 || m < smallest) smallest = m;
        break;  // larger b only grows m
      }
    if (smallest == 0) return 0;
    auto imbalance = [](uint32_t L) {  // r1 / r2 of the length's two register stages, >= 1
      const RegTileShape t = reg_tile_shape(L, (uint32_t)sizeof(cpx<T>));
      return (double)t.r1 / (double)t.r2;
    };
    uint64_t best = 0;
    double best_score = 0;
    for (int relaxed = 0; relaxed < 2 && best == 0; ++relaxed)  // the two lengths within a factor two of each other, if there is such a pair
      for (uint32_t a : menu())
        for (uint32_t b : menu()) {
          if (b > a) break;
          const uint64_t m = (uint64_t)a * b;
          if (m < need) continue;
          if (m * 100 > smallest * 104) break;
          if (a > 2 * b && !relaxed) continue;
          const double score = (double)m * (1.0 + 0.10 * (imbalance(b) - 1.0) + 0.03 * (imbalance(a) - 1.0));
          if (best == 0 || score < best_score) { best = m; best_score = score; l1 = a; l2 = b; }
        }
    return best;
  }
  BluTiledEngine(size_t n_user, uint32_t l1, uint32_t l2) : n_(n_user), m_((uint64_t)l1 * l2), l1_(l1), l2_(l2) {
    k_in_ = get_regtile_kernel(Real<T>{}, l1, 1);
    k_conv_ = get_regtile_kernel(Real<T>{}, l2, 2);
    k_out_ = get_regtile_kernel(Real<T>{}, l1, 3);
    if (!k_in_.fn || !k_conv_.fn || !k_out_.fn) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no register-tile Bluestein kernels of these lengths");
    for (const TiledKernel* k : {&k_in_, &k_conv_, &k_out_}) raise_smem_limit((const void*)k->fn, k->smem);
    tw1_.upload(stage_table(k_in_));
    tw2_.upload(stage_table(k_conv_));
    const int lb = (ilog2(m_) + 1) / 2;  // W_M^{i * k}, two-level (first forward pass and first inverse pass)
    lo_bits_ = (uint32_t)lb;
    std::vector<cpx<T>> lo((size_t)1 << lb), hi((size_t)(m_ >> lb) + 1);
    for (size_t e = 0; e < lo.size(); ++e) { double re, im; unit_root(e, m_, re, im); lo[e] = {(T)re, (T)im}; }
    for (size_t h = 0; h < hi.size(); ++h) { double re, im; unit_root((uint64_t)h << lb, m_, re, im); hi[h] = {(T)re, (T)im}; }
    tw_lo_.upload(lo); tw_hi_.upload(hi);
    const uint64_t e = sizeof(cpx<T>);
    chunk_in_ = ((n_ * e) % 128 != 0 || (l2_ * e) % 128 != 0) ? FOURIER_TILE_CHUNK : FOURIER_TILE_CHUNK_ALIGNED;
    chunk_conv_ = ((m_ * e) % 128 != 0 || (l1_ * e) % 128 != 0) ? FOURIER_TILE_CHUNK : FOURIER_TILE_CHUNK_ALIGNED;
    chunk_out_ = ((n_ * e) % 128 != 0 || (l2_ * e) % 128 != 0) ? FOURIER_TILE_CHUNK : FOURIER_TILE_CHUNK_ALIGNED;
  }
  uint64_t m() const { return m_; }
  std::string describe() const { return "mixed tiles " + std::to_string(l1_) + "x" + std::to_string(l2_); }
  // in: user array (n per transform), out: user array; work, scratch: m per transform each.  in == out is fine: the user array is read
  // completely by the first launch and written by the last.
  void run(const cpx<T>* in, cpx<T>* out, cpx<T>* work, cpx<T>* scratch, size_t batch, const void* xtab, const void* wtab, bool inverse,
           double scale, hipStream_t stream, Profiler* prof) const {
    if (batch == 0) return;
    TiledArgs a;
    auto base = [&]() {
      std::memset(&a, 0, sizeof(a));
      a.tw_lo = tw_lo_.p; a.tw_hi = tw_hi_.p; a.lo_bits = lo_bits_;
      a.n = m_; a.scale = 1.0;
      a.blu_x = xtab; a.blu_w = wtab; a.blu_n = n_; a.blu_swap = inverse ? 1 : 0;
    };
    auto launch = [&](const TiledKernel& k, uint64_t tiles, int slot) {
      const uint64_t grid = (uint64_t)batch * tiles;
      if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
      PROF_BEGIN(prof, slot);
      FOURIER_LAUNCH(k.fn, grid, k.threads, k.smem, stream, a);
      PROF_END(prof);
    };
    // chirp-in: first pass of the forward transform, length L1 at s = 1, m = L2 columns
    base();
    a.in = in; a.out = work; a.tw = tw1_.p; a.s = 1; a.m = l2_;
    a.tiles_per_row = (l2_ + k_in_.cols - 1) / k_in_.cols; a.xcd_chunk = chunk_in_;
    launch(k_in_, a.tiles_per_row, 0);
    // conv: last forward pass (length L2 at s = L1), (.) w, first inverse pass (length L2, m = L1)
    base();
    a.in = work; a.out = scratch; a.tw = tw2_.p; a.s = l1_; a.m = 1;
    a.tiles_per_row = (l1_ + k_conv_.cols - 1) / k_conv_.cols; a.xcd_chunk = chunk_conv_;
    launch(k_conv_, a.tiles_per_row, 1);
    // chirp-out: last pass of the inverse transform, length L1 at s = L2
    base();
    a.in = scratch; a.out = out; a.tw = tw1_.p; a.s = l2_; a.m = 1; a.swap_out = 1; a.scale = scale;
    a.tiles_per_row = (l2_ + k_out_.cols - 1) / k_out_.cols; a.xcd_chunk = chunk_out_;
    launch(k_out_, a.tiles_per_row, 2);
  }

 private:
  static std::vector<cpx<T>> stage_table(const TiledKernel& k) {  // W_L^{j2 * k1} as [k1 < r1][j2 < r2]
    std::vector<cpx<T>> tw;
    for (size_t k1 = 0; k1 < k.r1; ++k1)
      for (size_t j2 = 0; j2 < k.r2; ++j2) {
        double re, im;
        unit_root(j2 * k1, k.L, re, im);
        tw.push_back({(T)re, (T)im});
      }
    return tw;
  }
  size_t n_;
  uint64_t m_;
  uint32_t l1_, l2_, lo_bits_ = 0, chunk_in_ = 0, chunk_conv_ = 0, chunk_out_ = 0;
  TiledKernel k_in_, k_conv_, k_out_;
  DevBuf tw1_, tw2_, tw_lo_, tw_hi_;
};

// ---- Bluestein of a SHORT transform on a smooth M = R1 x R2 in one launch, both M-point transforms in registers (kernels_chirpz.h) ----
template <typename T> class BluRegEngine {
 public:
  static const std::vector<uint32_t>& menu() {
    static const std::vector<uint32_t> m = [] {
      std::vector<uint32_t> v;
      for (uint32_t mm = 32; mm <= 9261; ++mm)
        if (get_chirpz_kernel(Real<T>{}, mm).fn) v.push_back(mm);
      return v;
    }();
    return m;
  }
  // the smallest M of the menu that reaches 2n - 1; 0: none
  static uint32_t choose_m(size_t n) {
    for (uint32_t m : menu())
      if ((uint64_t)m >= 2 * (uint64_t)n - 1) return m;
    return 0;
  }
  // direct: the same register stages as a plain transform of n_user = m points (kernels_regfft.h) -- lengths with factors 5 ... 13
  // (A/B builds hold several variants of a three-stage length: FOURIER_REGFFT_VARIANT = 1 ... 6, kernels_regfft.cpp; experiments library)
  static int direct_variant() { const char* v = dev_env("FOURIER_REGFFT_VARIANT"); return v ? atoi(v) : 0; }
  static bool has_direct(size_t n) {
    return n <= 20480 && !dev_env("FOURIER_NO_REGFFT") && get_regfft_kernel(Real<T>{}, (uint32_t)n, direct_variant()).fn != nullptr;
  }
  // (variant 100: a 2^a 3^b length on request, plan option "register_stages")
  static bool has_direct_on_request(size_t n) { return n <= 20480 && get_regfft_kernel(Real<T>{}, (uint32_t)n, 100).fn != nullptr; }
  BluRegEngine(size_t n_user, uint32_t m, bool direct = false, int variant = -1)
      : n_(n_user), direct_(direct),
        k_(direct ? get_regfft_kernel(Real<T>{}, m, variant < 0 ? direct_variant() : variant) : get_chirpz_kernel(Real<T>{}, m)) {
    if (!k_.fn || (direct ? (uint64_t)m != n_user : (uint64_t)m < 2 * (uint64_t)n_user - 1))
      throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "no one-launch register kernel of this length");
    raise_smem_limit((const void*)k_.fn, k_.smem);
    std::vector<cpx<T>> tw;
    auto root = [&](uint64_t e, uint64_t size) {
      double re, im;
      unit_root(e % size, size, re, im);
      tw.push_back({(T)re, (T)im});
    };
    const uint64_t r1 = k_.r1, r2 = k_.r2, r3 = k_.r3;
    if (r3 == 0) {  // [j2][k1]: W_M^{j2 * k1}
      for (uint64_t j2 = 0; j2 < r2; ++j2)
        for (uint64_t k1 = 0; k1 < r1; ++k1) root(j2 * k1, k_.m);
    } else if (k_.fact) {  // regfft3_kernel<FACT>: W_M^{(j3 + R3 j2) k1} in two factors
      for (uint64_t j2 = 0; j2 < r2; ++j2)  // [j2][k1]: W_{R1 R2}^{j2 * k1}
        for (uint64_t k1 = 0; k1 < r1; ++k1) root(j2 * k1, r1 * r2);
      for (uint64_t k1 = 0; k1 < r1; ++k1)  // [k1 * R3 + j3]: W_M^{j3 * k1}
        for (uint64_t j3 = 0; j3 < r3; ++j3) root(j3 * k1, k_.m);
      for (uint64_t k2 = 0; k2 < r2; ++k2)  // [k2][j3]: W_{R2 R3}^{j3 * k2}
        for (uint64_t j3 = 0; j3 < r3; ++j3) root(j3 * k2, r2 * r3);
    } else {  // the four tables of chirpz_reg3_kernel, one after the other
      for (uint64_t j2 = 0; j2 < r2; ++j2)  // [j2][k1 * R3 + j3]: W_M^{(j3 + R3 * j2) * k1}
        for (uint64_t k1 = 0; k1 < r1; ++k1)
          for (uint64_t j3 = 0; j3 < r3; ++j3) root((j3 + r3 * j2) * k1, k_.m);
      for (uint64_t k2 = 0; k2 < r2; ++k2)  // [k2][j3]: W_{R2 R3}^{j3 * k2}
        for (uint64_t j3 = 0; j3 < r3; ++j3) root(j3 * k2, r2 * r3);
      for (uint64_t c3 = 0; c3 < r3; ++c3)  // [c3][a]: W_M^{a * c3}
        for (uint64_t a = 0; a < r1 * r2; ++a) root(a * c3, k_.m);
      for (uint64_t c2 = 0; c2 < r2; ++c2)  // [c2][k1]: W_{R1 R2}^{k1 * c2}
        for (uint64_t k1 = 0; k1 < r1; ++k1) root(k1 * c2, r1 * r2);
    }
    tw_.upload(tw);
  }
  uint64_t m() const { return k_.m; }
  bool three_stages() const { return k_.r3 != 0; }
  std::string describe() const {
    return "registers " + std::to_string(k_.r1) + "x" + std::to_string(k_.r2) + (k_.r3 ? "x" + std::to_string(k_.r3) : std::string()) + " one-launch";
  }
  bool direct() const { return direct_; }
  // in, out: user arrays (n per transform); in == out is fine: a wave reads its transforms completely before it writes them
  void run(const cpx<T>* in, cpx<T>* out, size_t batch, const void* xtab, const void* wtab, bool inverse, double scale, hipStream_t stream,
           Profiler* prof) const {
    if (batch == 0) return;
    ChirpzArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.chirp = xtab; a.w = wtab; a.tw = tw_.p;
    a.n = n_; a.batch = batch; a.swap = inverse ? 1 : 0; a.scale = scale;
    const uint64_t grid = ((uint64_t)batch + k_.tpw - 1) / k_.tpw;
    if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
    PROF_BEGIN(prof, 0);
    FOURIER_LAUNCH(k_.fn, grid, k_.threads, k_.smem, stream, a);
    PROF_END(prof);
  }

 private:
  size_t n_;
  bool direct_ = false;
  ChirpzKernel k_;
  DevBuf tw_;
};

}  // namespace fourier_hip
