// engine_pow2.h -- batched power-of-two FFT (and 2^a*3^b with a >= 12): the schedule of big-radix Stockham passes, its tables
// and launches.  Counterpart of Autosort::new / initialize_twiddles / apply_stages (autosort/mod.rs:24-46,104-134,313-404).
#pragma once
#include "engine_common.h"

namespace fourier_hip {

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
    KernelInfo k_skel;          // the pass without arithmetic and exchanges (experiments library: bench.py's streaming ceiling)
    KernelInfo k_pf, k_pf_blu;  // LAST pass: the persistent prefetching form (fft_last_prefetch_kernel), plain / chirp-out
    unsigned pf_grid = 0, pf_blu_grid = 0;  // resident workgroups of those kernels on this device
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
    // (2^10 as 32x32 only as a transform of its own: the one-launch chirp-z kernels of M = 1024 are built on the whole-row kernel)
    if (p3 == 1 && !dev_env("FOURIER_NO_TWOLEVEL") && (k > 10 || plain) && get_twolevel_kernel(Real<T>{}, k, tl, tl1, tl2)) {
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
    } else if (k == 20 && plain && p3 == 1 && dev_env("FOURIER_PLAN_2048x512")) {
      // experiment (round 5): 2^20 = 2048 x 512 instead of 1024 x 1024 -- row strides of 4 KiB (first pass) and 16 KiB (last pass) instead
      // of 8 KiB + 8 KiB, the stride the column-tile copy streams slowest at (profiles/r05_s6_stride_bench.jsonl)
      lens = {11, 9};
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
      pass->k = get_kernel(Real<T>{}, L, pass->mode, IO_PLAIN);
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
      if (pass->mode == MODE_LAST && !pass->k.split) {
        pass->k_pf = get_prefetch_kernel(Real<T>{}, L, IO_PLAIN);
        if (pass->k_pf.fn && pass->k_pf.COLS == pass->k.COLS) pass->pf_grid = resident_grid(pass->k_pf);
        else pass->k_pf = KernelInfo();
      }
      set_smem_attribute(pass->k);
      if (pass->mode == MODE_FIRST || pass->mode == MODE_LAST) {
        pass->k_skel = get_skeleton_kernel(Real<T>{}, L, pass->mode);
        if (pass->k_skel.fn && (pass->k_skel.COLS != pass->k.COLS || pass->k.split)) pass->k_skel = KernelInfo();
        if (pass->k_skel.fn) set_smem_attribute(pass->k_skel);
      }
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
      pass->odd_fn = get_odd_kernel(Real<T>{}, (int)r);
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
  // workgroups of a persistent kernel that are resident at once on this device, a multiple of 8 (one block sequence per XCD)
  static unsigned resident_grid(const KernelInfo& k) {
    set_smem_attribute(k);
    int per_cu = 0, cus = 0, dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn, k.NT, k.smem));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const unsigned g = (unsigned)std::max(1, per_cu) * (unsigned)std::max(1, cus);
    return g >= 8 ? g / 8 * 8 : g;
  }

  // Whole-Bluestein-in-one-launch (bluestein_small_kernel) is available when this (inner) plan is a
  // one-launch two-level plan: it additionally needs the inter-pass table of the role-swapped L2 x L1 problem.
  bool enable_bluestein_small() {
    if (tiny_ || passes_.size() != 1) return false;
    if (blu_small_.fn) return true;  // tables already uploaded (set_option may be called repeatedly)
    if (passes_[0]->mode == MODE_ROWS) {  // M <= 1024: row core twice, COLS transforms per workgroup
      if (!get_blu_small_kernel(Real<T>{}, ilog2(n_), blu_small_)) return false;
      set_smem_attribute(blu_small_);
      return true;
    }
    if (passes_[0]->mode != MODE_TWOLEVEL) return false;
    if (!get_blu_small_kernel(Real<T>{}, ilog2(n_), blu_small_)) return false;
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
    a.n = n_; a.scale = scale; a.nxcd = nxcd & 0xff; a.total_cols = batch;
    const uint64_t grid = rows ? (batch + blu_small_.COLS - 1) / blu_small_.COLS : batch;
    if (grid > 0x7fffffffull) throw EngineError(::fourier::c::FOURIER_HIP_UNSUPPORTED, "grid too large; lower chunk_bytes");
    PROF_BEGIN(prof, 0);
    FOURIER_LAUNCH(blu_small_.fn, grid, blu_small_.NT, blu_small_.smem, stream, a);
    PROF_END(prof);
  }

  // ---- XCD-fused two-pass plan (fft_l2fused_kernel): opt-in via the plan option "l2_fused"
  void init_l2fused(int k) {
    FusedInfo fi;
    if (!get_fused_kernel(Real<T>{}, k, fi)) return;
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
    f.k_blu = get_kernel(Real<T>{}, f.k.L, MODE_FIRST, IO_BLU_IN);
    l.k_blu = get_kernel(Real<T>{}, l.k.L, MODE_LAST, IO_BLU_OUT);
    f.has_blu = l.has_blu = true;
    for (Pass* p : {&f, &l}) set_smem_attribute(p->k_blu);
    if (l.mode == MODE_LAST && !l.k_blu.split) {
      l.k_pf_blu = get_prefetch_kernel(Real<T>{}, l.k.L, IO_BLU_OUT);
      if (l.k_pf_blu.fn && l.k_pf_blu.COLS == l.k_blu.COLS) l.pf_blu_grid = resident_grid(l.k_pf_blu);
      else l.k_pf_blu = KernelInfo();
    }
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
           BluIO blu = BluIO(), unsigned nxcd_last = 0) const {  // nxcd_last != 0: the tile order of the LAST pass alone
    if (batch == 0) return;
    // N = 16 (and f32 N = 32) also run one lane per transform; their ROWS pass only serves Bluestein M = 16 / 32
    const bool lane_per_transform = tiny_ || n_ == 16 || (n_ == 32 && sizeof(T) == 4);
    if (lane_per_transform && blu.io == IO_PLAIN) {
      TinyArgs a{in, out, (uint64_t)batch, (int)n_, inverse, inverse, scale};
      PROF_BEGIN(prof, slot0);
      const TinyKernel fn = get_tiny_kernel(Real<T>{}, n_);
      FOURIER_LAUNCH(fn, (batch + 255) / 256, 256, 0, stream, a);
      PROF_END(prof);
      apply_mul(out, batch, mul, inverse, scale, stream, prof, slot0);
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
      launch_pass(p, src[p], dst[p], batch, inverse, scale, stream, prof, slot0 + (int)p, (nxcd_last && p + 1 == np && np >= 2) ? nxcd_last : nxcd, blu);
    apply_mul(out, batch, mul, inverse, scale, stream, prof, slot0 + (int)np - 1);
  }

  // ---- the two passes of a two-pass plan software-pipelined over two internal streams (plan option "stream_pipeline")
  // The batch is cut into chunks of `chunk` transforms; the intermediate of a chunk lives in one of `slots` slots of a small
  // plan-owned ring instead of the output array.  Stream A runs pass 0 of chunk k+1 while stream B runs pass 1 of chunk k; the only
  // ordering is event to event (a slot is written after its previous tenant has been read, read after it has been written) -- no
  // in-kernel synchronisation.  The reference ping-pongs between two whole buffers (autosort/mod.rs:335-379); this is the same
  // ping-pong with a buffer of a few chunks.  In-place calls need no batch-sized scratch this way.  The caller's stream is
  // forked into A and B and joined again, so the call stays stream-ordered (and capturable).
  struct StreamPipe {
    hipStream_t sa = nullptr, sb = nullptr;
    hipEvent_t fork = nullptr, join_a = nullptr, join_b = nullptr;
    std::vector<hipEvent_t> written, drained;  // per slot: pass 0 has written it / pass 1 has read it
    DevBuf ring;
    void ensure(size_t slots, size_t slot_bytes) {
      if (!sa) {
        HIP_CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&join_a, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&join_b, hipEventDisableTiming));
      }
      while (written.size() < slots) {
        hipEvent_t w = nullptr, d = nullptr;
        HIP_CHECK(hipEventCreateWithFlags(&w, hipEventDisableTiming));
        written.push_back(w);
        HIP_CHECK(hipEventCreateWithFlags(&d, hipEventDisableTiming));
        drained.push_back(d);
      }
      ring.ensure(slots * slot_bytes);
    }
    ~StreamPipe() {
      for (hipEvent_t e : written) (void)hipEventDestroy(e);
      for (hipEvent_t e : drained) (void)hipEventDestroy(e);
      for (hipEvent_t e : {fork, join_a, join_b}) if (e) (void)hipEventDestroy(e);
      for (hipStream_t s : {sa, sb}) if (s) (void)hipStreamDestroy(s);
    }
  };
  bool can_pipeline() const { return !tiny_ && passes_.size() == 2 && passes_[0]->mode == MODE_FIRST && passes_[1]->mode == MODE_LAST && !last_is_split(); }
  void reserve_pipeline(size_t chunk, size_t slots) const { pipe_.ensure(slots, chunk * n_ * sizeof(cpx<T>)); }
  // one_stream (A/B control): the same chunked walk through the ring, both passes on ONE internal stream, nothing overlapped
  void run_pipelined(const cpx<T>* in, cpx<T>* out, size_t batch, size_t chunk, size_t slots, bool inverse, double scale,
                     hipStream_t stream, unsigned nxcd, unsigned nxcd_last, bool one_stream = false) const {
    if (batch == 0) return;
    chunk = std::max<size_t>(1, std::min(chunk, batch));
    slots = std::max<size_t>(2, slots);
    reserve_pipeline(chunk, slots);
    StreamPipe& sp = pipe_;
    const hipStream_t sa = sp.sa, sb = one_stream ? sp.sa : sp.sb;
    HIP_CHECK(hipEventRecord(sp.fork, stream));
    HIP_CHECK(hipStreamWaitEvent(sa, sp.fork, 0));
    if (sb != sa) HIP_CHECK(hipStreamWaitEvent(sb, sp.fork, 0));
    size_t k = 0;
    for (size_t b0 = 0; b0 < batch; b0 += chunk, ++k) {
      const size_t nb = std::min(chunk, batch - b0), slot = k % slots;
      cpx<T>* mid = (cpx<T>*)sp.ring.p + slot * chunk * n_;
      if (sb != sa && k >= slots) HIP_CHECK(hipStreamWaitEvent(sa, sp.drained[slot], 0));
      launch_pass(0, in + b0 * n_, mid, nb, inverse, scale, sa, nullptr, 0, nxcd);
      if (sb != sa) {
        HIP_CHECK(hipEventRecord(sp.written[slot], sa));
        HIP_CHECK(hipStreamWaitEvent(sb, sp.written[slot], 0));
      }
      launch_pass(1, mid, out + b0 * n_, nb, inverse, scale, sb, nullptr, 1, nxcd_last ? nxcd_last : nxcd);
      if (sb != sa) HIP_CHECK(hipEventRecord(sp.drained[slot], sb));
    }
    HIP_CHECK(hipEventRecord(sp.join_a, sa));
    HIP_CHECK(hipStreamWaitEvent(stream, sp.join_a, 0));
    if (sb != sa) {
      HIP_CHECK(hipEventRecord(sp.join_b, sb));
      HIP_CHECK(hipStreamWaitEvent(stream, sp.join_b, 0));
    }
  }

  // Pointwise multiplier on the M-point spectrum of a forward, unscaled transform (bluesteins.rs:236-239): its own sweep.
  // Only the unfused Bluestein options take it (bluestein_fusion = 0, bluestein_conv = 0); the default plans multiply
  // inside fft_conv_kernel / the one-launch kernels.  profile() adds its time to the slot of the last forward pass.
  void apply_mul(cpx<T>* out, size_t batch, const cpx<T>* mul, bool inverse, double scale, hipStream_t stream,
                 Profiler* prof = nullptr, int slot = 0) const {
    if (!mul) return;
    if (inverse || scale != 1.0) throw EngineError(::fourier::c::FOURIER_HIP_INVALID_ARGUMENT, "pointwise multiplier: forward unscaled only");
    BluArgs m{nullptr, out, mul, (uint64_t)n_, (uint64_t)n_, (uint64_t)batch, 0, 1.0};
    const size_t blocks = (batch * n_ + 255) / 256;
    PROF_BEGIN(prof, slot);  // counted in the slot of the pass it follows (the last forward pass)
    FOURIER_LAUNCH(get_blu_kernel(Real<T>{}, 2), std::min<size_t>(std::max<size_t>(blocks, 1), 256 * 32), 256, 0, stream, m);
    PROF_END(prof);
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
      a.xcd_interleave = (nxcd >> 8) & 7;
      a.walk_band = (nxcd >> 12) & 0xff; a.walk_group = (nxcd >> 20) & 0x3ff; a.walk_tf = nxcd >> 30;
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
      // LAST pass in its persistent prefetching form (plan option "last_pass_prefetch"): the same tiles, walked by as many
      // workgroups as are resident at once
      const KernelInfo& pf = blu_here ? ps.k_pf_blu : ps.k_pf;
      if (prefetch_last(ps) && ps.mode == MODE_LAST && pf.fn && !a.swap_in && (!blu_here || blu.io == IO_BLU_OUT)) {
        a.total_cols = grid;  // tiles of the whole launch
        const uint64_t resident = blu_here ? ps.pf_blu_grid : ps.pf_grid;
        PROF_BEGIN(prof, slot);
        FOURIER_LAUNCH(pf.fn, std::min<uint64_t>(grid, resident), pf.NT, pf.smem, stream, a);
        PROF_END(prof);
        return;
      }
      const KernelInfo& run = (skeleton_ && !blu_here && ps.k_skel.fn) ? ps.k_skel : kk;
      PROF_BEGIN(prof, slot);
      FOURIER_LAUNCH(run.fn, grid, run.NT, run.smem, stream, a);
      PROF_END(prof);
    }
  }
  // "skeleton" (experiments library): every pass that has one runs as its load / store skeleton -- timing only, wrong results
  bool has_skeleton() const {
    for (const auto& p : passes_) if (!p->k_skel.fn) return false;
    return !passes_.empty();
  }
  void set_skeleton(bool on) { skeleton_ = on; }
  // "last_pass_prefetch" (experiments library): 1 = wherever the kernel exists; off by default -- measured slower
  void set_prefetch_last(bool on) { prefetch_last_ = on; }
  bool has_prefetch_last() const { return !passes_.empty() && passes_.back()->k_pf.fn != nullptr; }
  bool prefetch_last(const Pass&) const { return prefetch_last_; }

  // Bluestein middle: this plan's LAST pass + (.) wtab + the FIRST pass of an inverse plan that starts with the
  // same length, in one launch (fft_conv_kernel).  src and dst are M-point work arrays, dst != src.
  bool can_conv() const { return !tiny_ && passes_.size() >= 2 && passes_.back()->mode == MODE_LAST; }
  void enable_conv() {
    if (!can_conv()) return;
    conv_ = get_conv_kernel(Real<T>{}, passes_.back()->k.L);
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
    a.xcd_interleave = (nxcd >> 8) & 7;
    a.walk_band = (nxcd >> 12) & 0xff; a.walk_group = (nxcd >> 20) & 0x3ff; a.walk_tf = nxcd >> 30;
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
  bool prefetch_last_ = false;
  bool skeleton_ = false;
  unsigned fused_grid_ = 0, fused_depth_ = 2;
  mutable DevBuf fused_window_, fused_ctrl_;
  mutable PinnedBuf fused_flag_;  // ctrl[1] (abort flag) of the last fused launch, read back before the call returns
  mutable StreamPipe pipe_;       // option "stream_pipeline": the two internal streams, their events and the ring
  std::string desc_override_;
  std::vector<std::unique_ptr<Pass>> passes_;
  std::map<int, std::unique_ptr<StageTables<T>>> stage_;
};

}  // namespace fourier_hip
