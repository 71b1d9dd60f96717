// rtc.cpp -- run-time specialisation: the per-length LDS mixed-radix kernel (kernels_mixed.h: mixed_radix_kernel_ct<T, N>) for a
// length that has no ahead-of-time instantiation, compiled with hipRTC when a plan asks for it (plan option "specialise").
//
// The ahead-of-time set (kernels_mixed_ct.cpp: every 2^a*3^b the reference runs natively, every 2^a*3^b*5^c, a selection with a
// factor 7) is what a build can afford -- 251 of the 875 lengths up to 20480 whose prime factors stop at 13; the other lengths
// run the runtime-parameterised kernel at 24-36 % of the HBM peak where a per-length kernel reaches 45-60 % (round 3).  hipRTC
// closes that gap without an instantiation per length in the library: the four device headers are embedded in the library
// (rtc_sources.inc, generated from the same files the build compiles), the program is an explicit instantiation of the one
// kernel, the code object is loaded as a module and cached per (precision, length) for the life of the process -- and, round 5,
// ON DISK: $FOURIER_HIP_CACHE_DIR, else $XDG_CACHE_HOME/fourier-hip, else $HOME/.cache/fourier-hip (an empty FOURIER_HIP_CACHE_DIR
// switches the disk cache off), one file per (device architecture, precision, kind, length, LDS bytes, hash of the embedded headers
// and compile options), written through a temporary file and rename(), read back only from a directory and a file that belong to this user and that nobody else may write (no symbolic links followed; the entry names its own key and payload length).  About one second per length the first time on a machine,
// a few milliseconds from the disk cache (no libhiprtc needed for that), nothing from the process cache.  libhiprtc is loaded
// lazily (dlopen): the library keeps libamdhip64 as its only link-time dependency, and where hipRTC is missing and the cache has no
// entry the caller gets FOURIER_HIP_UNSUPPORTED and the plan keeps its kernel.
#include "engine_common.h"
#include "mixed_schedule.h"

#ifndef FOURIER_EMU
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <mutex>
#ifdef FOURIER_RTC_SOURCES_INC  // packaging/CMakeLists.txt generates the file into its build directory
#include FOURIER_RTC_SOURCES_INC
#else
#include "rtc_sources.inc"
#endif
#endif

namespace fourier_hip {

#ifdef FOURIER_EMU
bool rtc_mixed_kernel(bool, uint32_t, size_t, RtcKernel&, std::string& why, bool, bool) { why = "no hipRTC under the CPU emulation"; return false; }
bool rtc_cached(bool, uint32_t, size_t, bool) { return false; }
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

const char* const RTC_OPTIONS[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"};
constexpr int RTC_NUM_OPTIONS = 4;

// ---- the on-disk cache
std::string cache_dir() {
  const char* e = getenv("FOURIER_HIP_CACHE_DIR");
  if (e) return std::string(e);  // (empty: no disk cache)
  if ((e = getenv("XDG_CACHE_HOME")) && *e) return std::string(e) + "/fourier-hip";
  if ((e = getenv("HOME")) && *e) return std::string(e) + "/.cache/fourier-hip";
  return std::string();
}
// FNV-1a over everything a code object depends on besides its key: the embedded headers, the compile options and the HIP runtime's
// version (the compiler comes with it; hipRTC's own version is only known once libhiprtc is loaded, which a cache hit never does)
uint64_t sources_hash() {
  static const uint64_t h = [] {
    uint64_t x = 1469598103934665603ull;
    auto eat = [&](const char* p) { for (; *p; ++p) { x ^= (unsigned char)*p; x *= 1099511628211ull; } x ^= 0xff; x *= 1099511628211ull; };
    for (int i = 0; i < RTC_NUM_HEADERS; ++i) { eat(RTC_HEADER_NAMES[i]); eat(RTC_HEADER_SOURCES[i]); }
    for (int i = 0; i < RTC_NUM_OPTIONS; ++i) eat(RTC_OPTIONS[i]);
    int ver = 0;
    if (hipRuntimeGetVersion(&ver) != hipSuccess) { (void)hipGetLastError(); ver = 0; }
    eat(("hip-runtime-" + std::to_string(ver)).c_str());
    return x;
  }();
  return h;
}
std::string cache_file(int dev, bool f64, uint32_t n, size_t lds_bytes, bool tile_pass) {
  const std::string dir = cache_dir();
  if (dir.empty()) return std::string();
  static std::mutex mu;
  static std::map<int, std::string> archs;  // device -> its architecture string (a property query costs about a millisecond)
  std::string arch;
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = archs.find(dev);
    if (it == archs.end()) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return std::string(); }
      std::string a = prop.gcnArchName;
      for (char& c : a) if (!isalnum((unsigned char)c)) c = '_';
      it = archs.emplace(dev, a).first;
    }
    arch = it->second;
  }
  char hash[24];
  snprintf(hash, sizeof hash, "%016llx", (unsigned long long)sources_hash());
  return dir + "/" + arch + "-" + (f64 ? "f64" : "f32") + "-" + (tile_pass ? "tile" : "whole") + "-n" + std::to_string(n) + "-lds" + std::to_string(lds_bytes) + "-" + hash + ".co";
}
void make_dirs(const std::string& dir) {
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), i == dir.size() ? 0700 : 0755);
}
const char CACHE_MAGIC[] = "FOURIER-HIP-CO-2\n";
// A code object is executed on the device, so a cache entry is trusted only when nobody else could have put it there: the directory
// belongs to this user and is not writable by group or others, the entry is opened without following a symbolic link, is a regular file
// of this user that nobody else may write, and says of itself what the caller asked for.
// file: magic line; "<tag> <payload bytes>\n" with tag = what the file name says (precision, kind, length, LDS bytes); the kernel's lowered
// name and a newline; the code object, exactly <payload bytes> long
std::string entry_tag(bool f64, uint32_t n, size_t lds_bytes, bool tile_pass) {
  return std::string(f64 ? "f64" : "f32") + "-" + (tile_pass ? "tile" : "whole") + "-n" + std::to_string(n) + "-lds" + std::to_string(lds_bytes);
}
bool trusted_dir(const std::string& dir) {
  struct stat st;
  return stat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && !(st.st_mode & (S_IWGRP | S_IWOTH));
}
// damaged (out): a file of THIS user in a trusted directory whose content is not a cache entry of this key (an older format, a truncated
// write, a renamed entry): the caller discards it
bool read_cache(const std::string& path, const std::string& tag, std::string& lowered, std::vector<char>& code, bool& damaged) {
  damaged = false;
  if (path.empty() || !trusted_dir(path.substr(0, path.rfind('/')))) return false;
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) { close(fd); return false; }
  FILE* f = fdopen(fd, "rb");
  if (!f) { close(fd); return false; }
  std::vector<char> all;
  char buf[65536];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + got);
  fclose(f);
  damaged = true;  // from here on the file is ours; anything but a well-formed entry of this key is discarded
  const size_t ml = sizeof(CACHE_MAGIC) - 1;
  if (all.size() <= ml || memcmp(all.data(), CACHE_MAGIC, ml) != 0) return false;
  const char* end = all.data() + all.size();
  const char* nl1 = (const char*)memchr(all.data() + ml, '\n', all.size() - ml);
  if (!nl1) return false;
  const std::string head((const char*)all.data() + ml, nl1);  // "<tag> <payload bytes>"
  const size_t sp = head.rfind(' ');
  if (sp == std::string::npos || head.substr(0, sp) != tag) return false;
  char* num_end = nullptr;
  const unsigned long long payload = strtoull(head.c_str() + sp + 1, &num_end, 10);
  if (!num_end || *num_end) return false;
  const char* nl2 = (const char*)memchr(nl1 + 1, '\n', (size_t)(end - (nl1 + 1)));
  if (!nl2 || (unsigned long long)(end - (nl2 + 1)) != payload || payload < 64) return false;  // truncated or padded
  lowered.assign(nl1 + 1, nl2);
  code.assign(nl2 + 1, end);
  // the code object must be an ELF image that ends inside the buffer (hipModuleLoadData takes no size)
  if (memcmp(code.data(), "\177ELF", 4) != 0) return false;
  uint64_t shoff = 0; uint16_t shentsize = 0, shnum = 0;
  memcpy(&shoff, code.data() + 0x28, 8); memcpy(&shentsize, code.data() + 0x3a, 2); memcpy(&shnum, code.data() + 0x3c, 2);
  if (shoff > code.size() || (uint64_t)shentsize * shnum > code.size() - shoff) return false;
  damaged = lowered.empty();
  return !damaged;
}
void write_cache(const std::string& path, const std::string& tag, const std::string& lowered, const std::vector<char>& code) {
  if (path.empty()) return;
  const std::string dir = path.substr(0, path.rfind('/'));
  make_dirs(dir);
  if (!trusted_dir(dir)) return;  // somebody else's (or a shared) directory: this process neither reads nor feeds it
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
  if (!f) { if (fd >= 0) close(fd); return; }
  const std::string head = tag + " " + std::to_string(code.size()) + "\n" + lowered + "\n";
  bool ok = fwrite(CACHE_MAGIC, 1, sizeof(CACHE_MAGIC) - 1, f) == sizeof(CACHE_MAGIC) - 1 && fwrite(head.data(), 1, head.size(), f) == head.size() &&
            fwrite(code.data(), 1, code.size(), f) == code.size();
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}
bool load_module(const std::vector<char>& code, const std::string& lowered, RtcKernel& out, std::string& why) {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { why = "hipModuleLoadData failed"; (void)hipGetLastError(); return false; }
  if (hipModuleGetFunction(&fn, mod, lowered.c_str()) != hipSuccess) { why = "hipModuleGetFunction failed"; (void)hipGetLastError(); (void)hipModuleUnload(mod); return false; }
  out.fn = (void*)fn;
  return true;
}
int cache_key(int dev, bool f64, bool tile_pass) { return (dev * 2 + (f64 ? 1 : 0)) * 2 + (tile_pass ? 1 : 0); }
}  // namespace

bool rtc_cached(bool f64, uint32_t n, size_t lds_bytes, bool tile_pass) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_cache.count(std::make_pair(cache_key(dev, f64, tile_pass), n))) return true;
  }
  const std::string path = cache_file(dev, f64, n, lds_bytes, tile_pass);
  return !path.empty() && trusted_dir(path.substr(0, path.rfind('/'))) && access(path.c_str(), R_OK) == 0;
}

bool rtc_mixed_kernel(bool f64, uint32_t n, size_t lds_bytes, RtcKernel& out, std::string& why, bool tile_pass, bool allow_compile) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { why = "no device"; return false; }
  const auto key = std::make_pair(cache_key(dev, f64, tile_pass), n);
  // the process cache under the lock; file I/O and the one-second compilation outside it (creates on other threads do not queue up behind a
  // compilation), the map re-checked before the insertion: two threads that compile the same kernel at once both succeed, one module stays unused
  auto cached = [&]() {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_cache.find(key);
    if (it == g_cache.end()) return false;
    out = it->second;
    return true;
  };
  auto publish = [&]() {
    std::lock_guard<std::mutex> lock(g_mu);
    auto ins = g_cache.emplace(key, out);
    if (!ins.second) out = ins.first->second;
  };
  if (cached()) return true;
  const std::string path = cache_file(dev, f64, n, lds_bytes, tile_pass), tag = entry_tag(f64, n, lds_bytes, tile_pass);
  {  // the disk cache: no compiler needed
    std::string lowered, err;
    std::vector<char> code;
    bool damaged = false;
    if (read_cache(path, tag, lowered, code, damaged)) {
      if (load_module(code, lowered, out, err)) { publish(); return true; }
      damaged = true;  // a file this runtime cannot load
    }
    if (damaged) (void)unlink(path.c_str());  // compile again (where asked to)
  }
  if (!allow_compile) { why = "not in the code-object cache (compilation not asked for)"; return false; }
  static Rtc rtc;  // (initialised once, thread-safe)
  if (!rtc.ok) { why = "libhiprtc not available"; return false; }
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
    const char* opts[RTC_NUM_OPTIONS + 1];
    for (int i = 0; i < RTC_NUM_OPTIONS; ++i) opts[i] = RTC_OPTIONS[i];
    opts[RTC_NUM_OPTIONS] = lds.c_str();
    if (rtc.compile(prog, RTC_NUM_OPTIONS + 1, opts) != 0) {
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
    if (!load_module(code, low, out, why)) break;
    write_cache(path, tag, low, code);
    publish();
    ok = true;
  } while (false);
  rtc.destroy(&prog);
  return ok;
}
#endif

}  // namespace fourier_hip
