"""Host-side mirror of the reference's operator interface for the FFT path.

Same names, argument meaning and error behaviour as the reference:
  * `Transform`                  -- fourier-algorithms/src/fft.rs:4-37
  * `Fft` (size / transform_in_place / transform / fft / ifft ...) -- fft.rs:40-82
  * `create_fft_f32`, `create_fft_f64` -- fourier/src/lib.rs:31-60
plus the batched, device-resident entry point the GPU path is measured on (the reference has no
batch API: fft.rs:48-61 takes one slice per call).

Buffers: numpy complex64/complex128 arrays go through the legacy host ABI (H2D + D2H inside the
library; arrays holding several transforms are streamed through the device in chunks); torch CUDA
tensors go through the device-resident batched ABI on the current stream.
All compute happens in libfourier.so (HIP); there is no CPU fallback.
"""
import enum

import numpy as np

from . import _lib


class Transform(enum.IntEnum):
    """fourier-algorithms/src/fft.rs:4-16; integer values = the C codes (fourier-ffi/src/lib.rs:3-12)."""

    Fft = 0
    Ifft = 1
    UnscaledIfft = 2
    SqrtScaledFft = 3
    SqrtScaledIfft = 4

    def is_forward(self):  # fft.rs:20-25
        return self in (Transform.Fft, Transform.SqrtScaledFft)

    def inverse(self):  # fft.rs:28-36
        return {Transform.Fft: Transform.Ifft, Transform.Ifft: Transform.Fft,
                Transform.SqrtScaledFft: Transform.SqrtScaledIfft,
                Transform.SqrtScaledIfft: Transform.SqrtScaledFft}.get(self)


class FourierError(RuntimeError):
    pass


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Fft:
    """The `Fft` trait (fft.rs:40-82) over a libfourier.so plan handle."""

    def __init__(self, size, real, device=-1):
        self._suffix = {"f32": "float", "f64": "double"}[real]
        self.real = real
        self.np_dtype = np.dtype(np.complex64 if real == "f32" else np.complex128)
        self._L = _lib.lib()
        self._h = getattr(self._L, f"fourier_hip_create_{self._suffix}")(int(size), int(device))
        if not self._h:
            # the reference's create panics -> NULL through the FFI (fourier-ffi/src/lib.rs:18-19)
            raise FourierError(f"cannot create FFT plan of size {size}")
        self._n = int(size)
        self.device = int(getattr(self._L, f"fourier_hip_device_{self._suffix}")(self._h))  # device=-1 binds the current one

    # -- trait surface ---------------------------------------------------------------------
    def size(self):
        return self._n

    def transform_in_place(self, input, transform):
        """fft.rs:48: in-place transform of exactly `size` elements (or batch*size, see below)."""
        self._dispatch(input, input, transform)

    def transform(self, input, output, transform):
        """fft.rs:51-61: out-of-place; asserts both lengths equal size()."""
        self._dispatch(input, output, transform)

    def fft_in_place(self, input):  # fft.rs:64-66
        self.transform_in_place(input, Transform.Fft)

    def ifft_in_place(self, input):  # fft.rs:69-71
        self.transform_in_place(input, Transform.Ifft)

    def fft(self, input, output):  # fft.rs:74-76
        self.transform(input, output, Transform.Fft)

    def ifft(self, input, output):  # fft.rs:79-81
        self.transform(input, output, Transform.Ifft)

    # -- batched device-resident extension -------------------------------------------------
    def transform_batch_ptr(self, d_in, d_out, batch, transform, stream=0):
        """Raw-pointer form: `batch` contiguous transforms on device memory, enqueued on `stream`."""
        st = getattr(self._L, f"fourier_hip_transform_batch_{self._suffix}")(
            self._h, d_in, d_out, int(batch), int(transform), stream)
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())

    def transform_batch_host(self, input, output, transform):
        """`batch` contiguous transforms in host (numpy) memory, streamed through the device in chunks with
        copies and kernels overlapped; synchronous.  input may be output (in place)."""
        for a in (input, output):
            if not (isinstance(a, np.ndarray) and a.dtype == self.np_dtype and a.flags.c_contiguous):
                raise TypeError(f"expected C-contiguous numpy {self.np_dtype} arrays")
        if input.size != output.size or input.size % self._n != 0:
            raise ValueError(f"buffers of {input.size}/{output.size} elements are not the same whole number of transforms")
        st = getattr(self._L, f"fourier_hip_transform_batch_host_{self._suffix}")(
            self._h, input.ctypes.data, output.ctypes.data, input.size // self._n, int(transform))
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())

    def synchronize(self, stream=0):
        """Blocks until everything queued on `stream` (a HIP stream handle, 0 = the NULL stream) of the plan's device has
        finished: the wait that follows a stream-ordered transform_batch_ptr when no other runtime owns the stream."""
        st = getattr(self._L, f"fourier_hip_synchronize_{self._suffix}")(self._h, stream)
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())

    def reserve(self, batch, in_place=False):
        """Pre-size the plan-owned device buffers so that later batched calls of up to `batch` transforms never
        allocate (hipMalloc synchronises the device; needed before HIP-graph capture)."""
        st = getattr(self._L, f"fourier_hip_reserve_{self._suffix}")(self._h, int(batch), int(bool(in_place)))
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())

    def profile_batch_ptr(self, d_in, d_out, batch, transform, stream=0, nslots=16):
        """One batched transform with a HIP event pair around every kernel launch.
        Returns [(slot_name, total_ms, launches), ...] in launch order."""
        import ctypes

        ms = (ctypes.c_float * nslots)()
        cnt = (ctypes.c_int * nslots)()
        st = getattr(self._L, f"fourier_hip_profile_{self._suffix}")(
            self._h, d_in, d_out, int(batch), int(transform), stream, nslots, ms, cnt)
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())
        names = getattr(self._L, f"fourier_hip_slot_names_{self._suffix}")(self._h).decode().split(",")
        return [(nm, float(ms[i]), int(cnt[i])) for i, nm in enumerate(names) if i < nslots]

    def set_option(self, key, value):
        st = getattr(self._L, f"fourier_hip_set_option_{self._suffix}")(self._h, key.encode(), int(value))
        if st != 0:
            raise FourierError(f"bad option {key}={value}")

    def describe(self):
        return getattr(self._L, f"fourier_hip_describe_{self._suffix}")(self._h).decode()

    def model_bytes(self):
        return getattr(self._L, f"fourier_hip_model_bytes_{self._suffix}")(self._h)

    # -- plumbing --------------------------------------------------------------------------
    def _dispatch(self, input, output, transform):
        code = int(transform)
        if _is_torch(input) or _is_torch(output):
            import torch

            want = torch.complex64 if self.real == "f32" else torch.complex128
            for t in (input, output):
                if not (_is_torch(t) and t.is_cuda and t.dtype == want and t.is_contiguous()):
                    raise TypeError(f"expected contiguous CUDA {want} tensors")
            # the reference asserts input.len() == output.len() == size (fft.rs:57-58); the batched
            # extension accepts any whole number of transforms
            if input.numel() != output.numel() or input.numel() % self._n != 0 or input.numel() == 0:
                raise ValueError(f"buffer of {input.numel()} elements is not a multiple of size {self._n}")
            # the plan's tables, scratch and kernels live on ONE device (fixed at creation)
            for t in (input, output):
                if t.device.index != self.device:
                    raise ValueError(f"tensor on cuda:{t.device.index}, plan on cuda:{self.device}")
            # same buffer = in place; anything else must not overlap (include/fourier.h)
            a0, b0 = input.data_ptr(), output.data_ptr()
            nbytes = input.numel() * input.element_size()
            if a0 != b0 and a0 < b0 + nbytes and b0 < a0 + nbytes:
                raise ValueError("input and output overlap partially")
            stream = torch.cuda.current_stream(input.device).cuda_stream
            self.transform_batch_ptr(input.data_ptr(), output.data_ptr(), input.numel() // self._n, code, stream)
            return
        for a in (input, output):
            if not (isinstance(a, np.ndarray) and a.dtype == self.np_dtype and a.flags.c_contiguous):
                raise TypeError(f"expected C-contiguous numpy {self.np_dtype} arrays")
        if not output.flags.writeable:
            raise ValueError("output is read-only")
        if input.size == output.size and input.size > self._n and input.size % self._n == 0:
            # several whole transforms in host memory: streamed through the device (extension)
            self.transform_batch_host(input, output, transform)
            return
        if input.size != self._n or output.size != self._n:
            raise ValueError(f"buffer length {input.size}/{output.size} != size {self._n}")  # fft.rs:57-58
        if input is output or input.ctypes.data == output.ctypes.data:
            getattr(self._L, f"fourier_transform_in_place_{self._suffix}")(self._h, output.ctypes.data, code)
        else:
            getattr(self._L, f"fourier_transform_{self._suffix}")(self._h, input.ctypes.data, output.ctypes.data, code)
        st = getattr(self._L, f"fourier_hip_last_status_{self._suffix}")(self._h)
        if st != 0:
            raise FourierError(self._L.fourier_hip_status_string(st).decode())

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                getattr(self._L, f"fourier_destroy_{self._suffix}")(h)
            except Exception:
                pass


def set_default_option(key, value):
    """Library-wide default for plans created afterwards (include/fourier.h: fourier_hip_set_default_option), e.g.
    ("specialise_at_create", 2): lengths whose prime factors stop at 13 get their own kernels compiled inside create_fft_*."""
    if _lib.lib().fourier_hip_set_default_option(key.encode(), int(value)) != 0:
        raise FourierError(f"bad default option {key}={value}")


def get_default_option(key):
    return int(_lib.lib().fourier_hip_get_default_option(key.encode()))


def create_fft_f32(size, device=-1):
    """fourier/src/lib.rs:31-43."""
    return Fft(size, "f32", device)


def create_fft_f64(size, device=-1):
    """fourier/src/lib.rs:49-60."""
    return Fft(size, "f64", device)
