/* fourier_warm_cache -- fills the on-disk code-object cache of libfourier ahead of time (include/fourier.h, "run-time specialisation").
 *
 * The reference's create_fft_* picks its plan implicitly (fourier/src/lib.rs:38-42); so does this library, but a length whose prime factors
 * stop at 13 and that has no ahead-of-time kernel only gets its own kernel when one is in the cache -- a plain create never compiles
 * (policy 1).  Run this once per machine and user (an install step: `cmake --build <dir> --target warm_cache`, or
 * `python -m fourier_amd.warm_cache`), and every later fourier_create_float / _double of these lengths loads its kernel in milliseconds.
 *
 * usage: fourier_warm_cache [max_length | length ...]     default: every such length up to 4096; both precisions; needs a GPU and libhiprtc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fourier.h"

static int smooth13(size_t n) {
  static const size_t p[] = {2, 3, 5, 7, 11, 13};
  for (size_t i = 0; i < sizeof p / sizeof p[0]; ++i)
    while (n % p[i] == 0) n /= p[i];
  return n == 1;
}

static int warm(size_t n, int verbose) {
  int compiled = 0;
  struct fourier_fft_float* f = fourier_create_float(n);
  if (f) {
    const char* d = fourier_hip_describe_float(f);
    if (strstr(d, "specialised")) ++compiled;
    if (verbose) printf("%zu f32: %s\n", n, d);
    fourier_destroy_float(f);
  }
  struct fourier_fft_double* g = fourier_create_double(n);
  if (g) {
    const char* d = fourier_hip_describe_double(g);
    if (strstr(d, "specialised")) ++compiled;
    if (verbose) printf("%zu f64: %s\n", n, d);
    fourier_destroy_double(g);
  }
  return compiled;
}

int main(int argc, char** argv) {
  if (fourier_hip_set_default_option("specialise_at_create", 2) != 0) {
    fprintf(stderr, "fourier_warm_cache: this libfourier has no run-time specialisation\n");
    return 2;
  }
  size_t plans = 0, lengths = 0;
  if (argc > 2 || (argc == 2 && !smooth13(0 + strtoull(argv[1], NULL, 10)))) {
    /* an explicit list of lengths (a single argument that is not 13-smooth is taken as a list of one) */
    for (int i = 1; i < argc; ++i) { plans += (size_t)warm(strtoull(argv[i], NULL, 10), 1); ++lengths; }
  } else {
    const size_t max_n = argc == 2 ? strtoull(argv[1], NULL, 10) : 4096;
    for (size_t n = 2; n <= max_n; ++n)
      if (smooth13(n)) { plans += (size_t)warm(n, 0); ++lengths; }
  }
  printf("fourier_warm_cache: %zu lengths, %zu plans run on kernels of their own from the cache\n", lengths, plans);
  return 0;
}
