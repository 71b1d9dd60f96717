// env_product.cpp -- what makes lib/libfourier.so the PRODUCT library: no development switches (dev_env() sees no
// environment) and none of the measured-slower designs (their registry entries report "not available").  The experiments
// library links env_experiments.cpp and kernels_experiments.cpp in place of this file (engine_common.h).
#include "engine_common.h"

namespace fourier_hip {

const char* dev_env(const char*) { return nullptr; }

// measured 30-45 % slower than the two-launch plan / 5-7 % slower than the 16-column last pass
bool get_fused_kernel(Real<float>, int, FusedInfo&) { return false; }
bool get_fused_kernel(Real<double>, int, FusedInfo&) { return false; }
KernelInfo get_split_kernel(Real<float>, int, int) { return KernelInfo(); }
KernelInfo get_split_kernel(Real<double>, int, int) { return KernelInfo(); }
// the persistent prefetching last pass: 15-30 % slower than fft_pass_kernel (kernels_experiments.h has the measurements)
KernelInfo get_prefetch_kernel(Real<float>, int, int) { return KernelInfo(); }
KernelInfo get_prefetch_kernel(Real<double>, int, int) { return KernelInfo(); }
// measurement tooling (the passes' load / store skeleton, bench.py's streaming ceiling)
KernelInfo get_skeleton_kernel(Real<float>, int, int) { return KernelInfo(); }
KernelInfo get_skeleton_kernel(Real<double>, int, int) { return KernelInfo(); }

}  // namespace fourier_hip
