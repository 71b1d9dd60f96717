/* The exact FFI call sequence the Rust shim makes (rust/fourier-hip/src/lib.rs), issued from C because the build
 * image has no Rust toolchain: for both precisions
 *   new():                 fourier_create_*            (NULL on failure; size 0 -> NULL)
 *   Fft::transform_in_place: fourier_transform_in_place_* then fourier_hip_last_status_*
 *   Fft::transform:        fourier_transform_*        then fourier_hip_last_status_*
 *   transform_batch():     fourier_hip_transform_batch_host_*  (several transforms in host memory)
 *   reserve():             fourier_hip_reserve_*
 *   Drop:                  fourier_destroy_*
 * with the transform codes of `code()` (= fourier-ffi/src/lib.rs:3-12 inverted), the unknown-code no-op and the
 * status reset after a failed call.  Known answers: the N = 4 impulse of fourier-ffi/test.c:7-39 and a tone. */
#include "fourier.h"
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(cond, code) do { if (!(cond)) { fprintf(stderr, "ffi_sequence: check %d failed (%s)\n", code, #cond); return code; } } while (0)

static int run_float(void) {
  struct fourier_fft_float *fft = fourier_create_float(4);
  CHECK(fft != NULL, 1);
  float complex x[4] = {1, 0, 0, 0}, y[4];
  fourier_transform_in_place_float(fft, x, 0 /* Transform::Fft */);
  CHECK(fourier_hip_last_status_float(fft) == FOURIER_HIP_OK, 2);
  for (int i = 0; i < 4; i++) CHECK(cabsf(x[i] - 1.0f) < 1e-6f, 3);
  fourier_transform_float(fft, x, y, 1 /* Transform::Ifft */);
  CHECK(fourier_hip_last_status_float(fft) == FOURIER_HIP_OK, 4);
  CHECK(cabsf(y[0] - 1.0f) < 1e-6f && cabsf(y[1]) < 1e-6f && cabsf(y[2]) < 1e-6f && cabsf(y[3]) < 1e-6f, 5);
  /* codes 2..4: UnscaledIfft, SqrtScaledFft, SqrtScaledIfft on a constant vector */
  float complex c[4] = {1, 1, 1, 1}, d[4];
  fourier_transform_float(fft, c, d, 2);
  CHECK(cabsf(d[0] - 4.0f) < 1e-5f && cabsf(d[1]) < 1e-5f, 6);
  fourier_transform_float(fft, c, d, 3);
  CHECK(cabsf(d[0] - 2.0f) < 1e-5f, 7);
  fourier_transform_float(fft, c, d, 4);
  CHECK(cabsf(d[0] - 2.0f) < 1e-5f, 8);
  d[0] = 42;
  fourier_transform_float(fft, c, d, 9); /* unknown code: silent no-op (lib.rs:10) */
  CHECK(crealf(d[0]) == 42.0f, 9);
  /* a failed call sets the status, the next successful call clears it */
  CHECK(fourier_hip_transform_batch_host_float(fft, c, d, 1, 77) == FOURIER_HIP_INVALID_ARGUMENT, 10);
  CHECK(fourier_hip_last_status_float(fft) == FOURIER_HIP_INVALID_ARGUMENT, 11);
  fourier_transform_in_place_float(fft, c, 0);
  CHECK(fourier_hip_last_status_float(fft) == FOURIER_HIP_OK, 12);
  fourier_destroy_float(fft);

  /* transform_batch(): 3 transforms of a non-power-of-two size held in host memory, tone in bin 5 */
  enum { N = 96, B = 3 };
  struct fourier_fft_float *big = fourier_create_float(N);
  CHECK(big != NULL, 13);
  CHECK(fourier_hip_reserve_float(big, B, 0) == FOURIER_HIP_OK, 14);
  float complex *in = malloc(sizeof(float complex) * N * B), *out = malloc(sizeof(float complex) * N * B);
  for (int b = 0; b < B; b++)
    for (int i = 0; i < N; i++) in[b * N + i] = (b + 1) * cexpf(2.0f * 3.14159265358979323846f * I * 5.0f * (float)i / (float)N);
  CHECK(fourier_hip_transform_batch_host_float(big, in, out, B, 0) == FOURIER_HIP_OK, 15);
  for (int b = 0; b < B; b++)
    for (int k = 0; k < N; k++) CHECK(cabsf(out[b * N + k] - (k == 5 ? (float)(N * (b + 1)) : 0.0f)) < 2e-3f, 16);
  free(in); free(out);
  fourier_destroy_float(big);
  return 0;
}

static int run_double(void) {
  struct fourier_fft_double *fft = fourier_create_double(4);
  CHECK(fft != NULL, 21);
  double complex x[4] = {1, 0, 0, 0}, y[4];
  fourier_transform_in_place_double(fft, x, 0);
  CHECK(fourier_hip_last_status_double(fft) == FOURIER_HIP_OK, 22);
  for (int i = 0; i < 4; i++) CHECK(cabs(x[i] - 1.0) < 1e-10, 23);
  fourier_transform_double(fft, x, y, 1);
  CHECK(fourier_hip_last_status_double(fft) == FOURIER_HIP_OK, 24);
  CHECK(cabs(y[0] - 1.0) < 1e-10 && cabs(y[1]) < 1e-10 && cabs(y[2]) < 1e-10 && cabs(y[3]) < 1e-10, 25);
  CHECK(fourier_hip_reserve_double(fft, 8, 1) == FOURIER_HIP_OK, 26);
  fourier_destroy_double(fft);
  return 0;
}

int main(void) {
  CHECK(fourier_create_float(0) == NULL && fourier_create_double(0) == NULL, 90); /* new(0) -> None */
  fourier_destroy_float(NULL);                                                    /* Drop of a moved-from handle: no-op */
  CHECK(fourier_hip_last_status_float(NULL) == FOURIER_HIP_INVALID_ARGUMENT, 91);
  int rc = run_float();
  if (rc) return rc;
  rc = run_double();
  if (rc) return rc;
  printf("Tests ran successfully.\n");
  return 0;
}
