// env_experiments.cpp -- development switches of lib/libfourier_experiments.so and of the CPU emulation build: environment
// variables read at plan creation (FOURIER_NO_TWOLEVEL, FOURIER_WIDE_2048, FOURIER_SPLIT_2048, FOURIER_PLAN_4096, ...;
// DESIGN.md section 6 lists them).  The product library links env_product.cpp instead.
#include "engine_common.h"

namespace fourier_hip {

const char* dev_env(const char* name) { return getenv(name); }

}  // namespace fourier_hip
