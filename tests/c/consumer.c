/* C consumer of include/fourier.h: the check the reference's own ctest program performs
 * (fourier-ffi/test.c:7-39: N=4 impulse, FFT out of place then IFFT in place, compare to the input
 * within 1e-10), written against this repo's header, plus the batched device extension is NOT used
 * here on purpose: this is what an existing C user relinks without source changes. */
#include "fourier.h"
#include <complex.h>
#include <stdio.h>

static int check_float(void) {
  float complex in[4] = {1, 0, 0, 0}, out[4];
  struct fourier_fft_float *fft = fourier_create_float(4);
  if (!fft) return 1;
  fourier_transform_float(fft, in, out, FOURIER_TRANSFORM_FFT);
  for (int i = 0; i < 4; i++)
    if (cabsf(out[i] - 1.0f) > 1e-6f) return 2;
  fourier_transform_in_place_float(fft, out, FOURIER_TRANSFORM_IFFT);
  fourier_destroy_float(fft);
  for (int i = 0; i < 4; i++)
    if (cabsf(in[i] - out[i]) > 1e-10f) return 3;
  return 0;
}

static int check_double(void) {
  double complex in[4] = {1, 0, 0, 0}, out[4];
  struct fourier_fft_double *fft = fourier_create_double(4);
  if (!fft) return 1;
  fourier_transform_double(fft, in, out, FOURIER_TRANSFORM_FFT);
  fourier_transform_in_place_double(fft, out, FOURIER_TRANSFORM_IFFT);
  fourier_destroy_double(fft);
  for (int i = 0; i < 4; i++)
    if (cabs(in[i] - out[i]) > 1e-10) return 3;
  return 0;
}

int main(void) {
  int a = check_float(), b = check_double();
  if (a || b) {
    fprintf(stderr, "consumer.c failed: float=%d double=%d\n", a, b);
    return 1;
  }
  if (fourier_create_float(0) != NULL) return 2; /* size 0 -> NULL */
  printf("Tests ran successfully.\n");
  return 0;
}
