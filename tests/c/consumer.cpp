// C++ consumer of include/fourier.h through the header-only RAII wrapper fourier::fft<T>
// (same shape as the reference's, fourier-ffi/include/fourier.h:64-128; behaviour checked as in
// fourier-ffi/test.cpp:17-48: N=4 impulse round trip).
#include "fourier.h"
#include <complex>
#include <cstdio>

template <typename T> static int check() {
  std::complex<T> in[4] = {1, 0, 0, 0}, out[4];
  fourier::fft<T> fft(4);
  if (!fft) return 1;
  fft.transform(in, out, fourier::transform::fft);
  fft.transform_in_place(out, fourier::transform::ifft);
  for (int i = 0; i < 4; i++)
    if (std::abs(in[i] - out[i]) > T(1e-10)) return 2;
  // all five scalings are reachable through the enum class
  fft.transform(in, out, fourier::transform::sqrt_scaled_fft);
  for (int i = 0; i < 4; i++)
    if (std::abs(out[i] - std::complex<T>(T(0.5), 0)) > T(1e-6)) return 3;
  fourier::fft<T> moved(std::move(fft));
  moved.transform(in, out, fourier::transform::unscaled_ifft);
  if (std::abs(out[0] - std::complex<T>(1, 0)) > T(1e-6)) return 4;
  // extension: several transforms held in host memory, one call (impulses at 0, 1, 2 -> 1, -i at bin 1, +1 at bin 2)
  std::complex<T> many[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, res[12];
  if (moved.transform_batch_host(many, res, 3, fourier::transform::fft) != 0) return 5;
  if (std::abs(res[0] - std::complex<T>(1, 0)) > T(1e-6) || std::abs(res[5] - std::complex<T>(0, -1)) > T(1e-6) ||
      std::abs(res[10] - std::complex<T>(1, 0)) > T(1e-6))
    return 6;
  return 0;
}

int main() {
  const int a = check<float>(), b = check<double>();
  if (a || b) {
    std::fprintf(stderr, "consumer.cpp failed: float=%d double=%d\n", a, b);
    return 1;
  }
  std::printf("Tests ran successfully.\n");
  return 0;
}
