// engine.cpp -- host side of libfourier.so: plan factory, pass scheduling, C ABI.
//
// Mirrors the reference's plan layer one level up:
//   create_fft_f32/f64  (fourier/src/lib.rs:31-60)        -> Plan<T>::create      (Stockham, else Bluestein)
//   Autosort::new       (autosort/mod.rs:104-134)         -> Pow2Engine<T>        (big-radix pass schedule)
//   initialize_twiddles (autosort/mod.rs:24-46)           -> make_stage_tables / make_two_level (f64 trig, cast)
//   Bluesteins::new     (bluesteins.rs:109-130, :18-61)   -> Plan<T>::init_bluestein
//   apply_stages / apply (mod.rs:313-404, bluesteins.rs:215-259) -> Plan<T>::exec
//   fourier-ffi C ABI   (fourier-ffi/src/lib.rs:14-106)   -> extern "C" block at the end
// This translation unit holds the host logic only (engine_pow2.h, engine_mixed.h, engine_generic.h, plan.h) and the C
// ABI; the kernels are instantiated in kernels_*.cpp, one object per family and precision, and reached through the registry
// of engine_common.h.  Compiled with hipcc for gfx950; the same files build against tests/emu/hipemu.h (-DFOURIER_EMU)
// for CPU-side logic tests only.
#include "plan.h"

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

// ---- library-wide defaults for plans created afterwards
namespace fourier_hip {
namespace {
std::atomic<int> g_specialise_policy{-1};  // -1: not decided yet (the environment is read on first use)
}
int specialise_policy() {
  int v = g_specialise_policy.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("FOURIER_HIP_SPECIALISE");  // "0", "1", "2": for programs that cannot be changed to call the function
    v = (e && *e >= '0' && *e <= '2' && !e[1]) ? *e - '0' : 1;
    g_specialise_policy.store(v, std::memory_order_relaxed);
  }
  return v;
}
void set_specialise_policy(int v) { g_specialise_policy.store(v, std::memory_order_relaxed); }
namespace {
std::atomic<int> g_register_stages{-1};
}
int register_stages_default() {
  int v = g_register_stages.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("FOURIER_HIP_REGISTER_STAGES");
    v = (e && e[0] == '1' && !e[1]) ? 1 : 0;
    g_register_stages.store(v, std::memory_order_relaxed);
  }
  return v;
}
void set_register_stages_default(int v) { g_register_stages.store(v, std::memory_order_relaxed); }
}  // namespace fourier_hip

extern "C" int fourier_hip_set_default_option(const char* key, long long v) {
  if (!key) return fc::FOURIER_HIP_INVALID_ARGUMENT;
  if (std::string(key) == "specialise_at_create" && v >= 0 && v <= 2) { set_specialise_policy((int)v); return fc::FOURIER_HIP_OK; }
  if (std::string(key) == "register_stages_at_create" && (v == 0 || v == 1)) { set_register_stages_default((int)v); return fc::FOURIER_HIP_OK; }
  return fc::FOURIER_HIP_INVALID_ARGUMENT;
}
extern "C" long long fourier_hip_get_default_option(const char* key) {
  if (key && std::string(key) == "specialise_at_create") return specialise_policy();
  if (key && std::string(key) == "register_stages_at_create") return register_stages_default();
  return -1;
}

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
