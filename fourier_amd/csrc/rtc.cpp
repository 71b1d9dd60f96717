// rtc.cpp -- run-time specialisation: the per-length LDS mixed-radix kernel (kernels_mixed.h: mixed_radix_kernel_ct<T, N>) for a
// length that has no ahead-of-time instantiation, compiled with hipRTC when a plan asks for it (plan option "specialise").
//
// The ahead-of-time set (kernels_mixed_ct.cpp: every 2^a*3^b the reference runs natively, every 2^a*3^b*5^c, a selection with a
// factor 7) is what a build can afford -- 251 of the 875 lengths up to 20480 whose prime factors stop at 13; the other lengths
// run the runtime-parameterised kernel at 24-36 % of the HBM peak where a per-length kernel reaches 45-60 % (round 3).  hipRTC
// closes that gap without an instantiation per length in the library: the four device headers are embedded in the library
// (rtc_sources.inc, generated from the same files the build compiles), the program is an explicit instantiation of the one
// kernel, the code object is loaded as a module and cached per (precision, length) for the life of the process.  About one
// second per length, paid when the option is set, never on plan creation or in a transform call.  libhiprtc is loaded lazily
// (dlopen): the library keeps libamdhip64 as its only link-time dependency, and where hipRTC is missing the option reports
// FOURIER_HIP_UNSUPPORTED and the plan keeps its kernel.
#include "engine_common.h"
#include "mixed_schedule.h"

#ifndef FOURIER_EMU
#include <dlfcn.h>
#include <mutex>
#include "rtc_sources.inc"
#endif

namespace fourier_hip {

#ifdef FOURIER_EMU
bool rtc_mixed_kernel(bool, uint32_t, size_t, RtcKernel&, std::string& why, bool) { why = "no hipRTC under the CPU emulation"; return false; }
#else

namespace {
// the few hipRTC entry points, resolved once
struct Rtc {
  typedef struct _hiprtcProgram* Program;
  int (*create)(Program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*add_name)(Program, const char*) = nullptr;
  int (*compile)(Program, int, const char* const*) = nullptr;
  int (*lowered)(Program, const char*, const char**) = nullptr;
  int (*code_size)(Program, size_t*) = nullptr;
  int (*code)(Program, char*) = nullptr;
  int (*log_size)(Program, size_t*) = nullptr;
  int (*log)(Program, char*) = nullptr;
  int (*destroy)(Program*) = nullptr;
  bool ok = false;
  Rtc() {
    void* h = nullptr;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    create = (decltype(create))sym("hiprtcCreateProgram");
    add_name = (decltype(add_name))sym("hiprtcAddNameExpression");
    compile = (decltype(compile))sym("hiprtcCompileProgram");
    lowered = (decltype(lowered))sym("hiprtcGetLoweredName");
    code_size = (decltype(code_size))sym("hiprtcGetCodeSize");
    code = (decltype(code))sym("hiprtcGetCode");
    log_size = (decltype(log_size))sym("hiprtcGetProgramLogSize");
    log = (decltype(log))sym("hiprtcGetProgramLog");
    destroy = (decltype(destroy))sym("hiprtcDestroyProgram");
    ok = create && add_name && compile && lowered && code_size && code && log_size && log && destroy;
  }
};
std::mutex g_mu;
std::map<std::pair<int, uint32_t>, RtcKernel> g_cache;  // ((device, f64, tile pass), n) -> loaded kernel; modules live as long as the process
}  // namespace

bool rtc_mixed_kernel(bool f64, uint32_t n, size_t lds_bytes, RtcKernel& out, std::string& why, bool tile_pass) {
  static Rtc rtc;
  if (!rtc.ok) { why = "libhiprtc not available"; return false; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { why = "no device"; return false; }
  std::lock_guard<std::mutex> lock(g_mu);
  const auto key = std::make_pair((dev * 2 + (f64 ? 1 : 0)) * 2 + (tile_pass ? 1 : 0), n);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) { out = it->second; return true; }
  const std::string real = f64 ? "double" : "float";
  // the whole-transform kernel of length n, or the column-tile pass of length n (kernels_tiled.h)
  const std::string kernel = tile_pass ? "tiled_mixed_kernel_ct" : "mixed_radix_kernel_ct", args = tile_pass ? "TiledArgs" : "MixArgs";
  const std::string expr = "fourier_hip::" + kernel + "<" + real + ", " + std::to_string(n) + "u>";
  const std::string src = std::string("#include \"") + (tile_pass ? "kernels_tiled.h" : "kernels_mixed.h") + "\"\nnamespace fourier_hip { template __global__ void " +
                          kernel + "<" + real + ", " + std::to_string(n) + "u>(" + args + "); }\n";
  Rtc::Program prog = nullptr;
  if (rtc.create(&prog, src.c_str(), "fourier_rtc_mixed.hip", RTC_NUM_HEADERS, RTC_HEADER_SOURCES, RTC_HEADER_NAMES) != 0) { why = "hiprtcCreateProgram failed"; return false; }
  bool ok = false;
  do {
    if (rtc.add_name(prog, expr.c_str()) != 0) { why = "hiprtcAddNameExpression failed"; break; }
    // the flags of fourier_amd/build.py (SLP packing of f32 math doubles the butterflies' live registers)
    // and the kernel's LDS footprint, which it declares statically (kernels_common.h: FOURIER_RTC_LDS_BYTES)
    const std::string lds = "-DFOURIER_RTC_LDS_BYTES=" + std::to_string(lds_bytes);
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", lds.c_str()};
    if (rtc.compile(prog, 5, opts) != 0) {
      size_t ls = 0;
      rtc.log_size(prog, &ls);
      std::string log(ls, '\0');
      if (ls) rtc.log(prog, &log[0]);
      why = "hiprtcCompileProgram failed: " + log.substr(0, 2000);
      break;
    }
    const char* low = nullptr;
    if (rtc.lowered(prog, expr.c_str(), &low) != 0 || !low) { why = "hiprtcGetLoweredName failed"; break; }
    size_t cs = 0;
    if (rtc.code_size(prog, &cs) != 0 || cs == 0) { why = "hiprtcGetCodeSize failed"; break; }
    std::vector<char> code(cs);
    if (rtc.code(prog, code.data()) != 0) { why = "hiprtcGetCode failed"; break; }
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { why = "hipModuleLoadData failed"; (void)hipGetLastError(); break; }
    if (hipModuleGetFunction(&fn, mod, low) != hipSuccess) { why = "hipModuleGetFunction failed"; (void)hipGetLastError(); (void)hipModuleUnload(mod); break; }
    out.fn = (void*)fn;
    g_cache.emplace(key, out);
    ok = true;
  } while (false);
  rtc.destroy(&prog);
  return ok;
}
#endif

}  // namespace fourier_hip
