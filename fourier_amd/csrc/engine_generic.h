// engine_generic.h -- 2^a * 3^b with a < 12 beyond the LDS kernels' reach: the reference's Stockham autosort pass by pass in global
// memory (autosort/mod.rs:203-284).
#pragma once
#include "engine_common.h"

namespace fourier_hip {

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
      ps.tw_lo.reset(new DevBuf());
      ps.tw_hi.reset(new DevBuf());
      if (ps.m > 1) {  // W_size^{e} as a two-level table: lo[e & mask] * hi[e >> bits] (f64 trig, cast: twiddle.rs:7-19)
        const int lb = (ilog2(size) + 1) / 2;
        ps.lo_bits = (uint32_t)lb;
        std::vector<cpx<T>> lo((size_t)1 << lb), hi((size >> lb) + 1);
        for (size_t e = 0; e < lo.size(); ++e) { double re, im; unit_root(e, size, re, im); lo[e] = {(T)re, (T)im}; }
        for (size_t h = 0; h < hi.size(); ++h) { double re, im; unit_root((uint64_t)h << lb, size, re, im); hi[h] = {(T)re, (T)im}; }
        ps.tw_lo->upload(lo);
        ps.tw_hi->upload(hi);
      }
      ps.fn = get_stockham_pass_kernel(Real<T>{}, r);
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
      a.in = src; a.out = dst;
      a.tw_lo = ps.m > 1 ? ps.tw_lo->p : nullptr; a.tw_hi = ps.m > 1 ? ps.tw_hi->p : nullptr; a.lo_bits = ps.lo_bits;
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
  struct Pass { int r = 0; uint32_t s = 0, m = 0, lo_bits = 0; std::unique_ptr<DevBuf> tw_lo, tw_hi; void (*fn)(GenArgs) = nullptr; size_t smem = 0; };
  size_t n_;
  std::vector<Pass> passes_;
};

}  // namespace fourier_hip
