// fft_kernels.h -- umbrella over the gfx950 device code of the batched 1D c2c FFT engine (one header per kernel family;
// kernels_common.h describes the data layout and lists the families).
#pragma once
#include "kernels_common.h"
#include "kernels_pass.h"
#include "kernels_onelaunch.h"
#include "kernels_small.h"
#include "kernels_mixed.h"
#include "kernels_misc.h"
#ifdef FOURIER_EXPERIMENTS
#include "kernels_experiments.h"
#endif
