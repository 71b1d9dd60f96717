"""fourier_amd: MI355X-native batched 1D c2c FFT engine behind calebzulawski/fourier's plan API.

Only the hot path lives here: csrc/ (HIP kernels + the C ABI of include/fourier.h) and the host-side
mirror of the reference's operator interface (fft.py).
"""
from .fft import Fft, FourierError, Transform, create_fft_f32, create_fft_f64, get_default_option, set_default_option  # noqa: F401
