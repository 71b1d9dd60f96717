"""ctypes loader for the CPU oracle (oracle/fourier_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (fourier_amd) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfourier_oracle.so")

FFT, IFFT, UNSCALED_IFFT, SQRT_SCALED_FFT, SQRT_SCALED_IFFT = range(5)


def build(force=False):
    src = os.path.join(_HERE, "fourier_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfourier_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        for s in ("float", "double"):
            getattr(L, f"oracle_fourier_create_{s}").restype = vp
            getattr(L, f"oracle_fourier_create_{s}").argtypes = [sz]
            getattr(L, f"oracle_fourier_destroy_{s}").restype = None
            getattr(L, f"oracle_fourier_destroy_{s}").argtypes = [vp]
            getattr(L, f"oracle_fourier_size_{s}").restype = sz
            getattr(L, f"oracle_fourier_size_{s}").argtypes = [vp]
            getattr(L, f"oracle_fourier_transform_in_place_{s}").restype = None
            getattr(L, f"oracle_fourier_transform_in_place_{s}").argtypes = [vp, vp, ci]
            getattr(L, f"oracle_fourier_transform_{s}").restype = None
            getattr(L, f"oracle_fourier_transform_{s}").argtypes = [vp, vp, vp, ci]
            getattr(L, f"oracle_fourier_batch_create_{s}").restype = vp
            getattr(L, f"oracle_fourier_batch_create_{s}").argtypes = [sz, ci]
            getattr(L, f"oracle_fourier_batch_destroy_{s}").restype = None
            getattr(L, f"oracle_fourier_batch_destroy_{s}").argtypes = [vp]
            getattr(L, f"oracle_fourier_batch_run_{s}").restype = None
            getattr(L, f"oracle_fourier_batch_run_{s}").argtypes = [vp, vp, vp, sz, ci]
            getattr(L, f"oracle_fourier_batch_stage_{s}").restype = ci
            getattr(L, f"oracle_fourier_batch_stage_{s}").argtypes = [vp, vp, sz]
            getattr(L, f"oracle_fourier_batch_cpu_{s}").restype = ci
            getattr(L, f"oracle_fourier_batch_cpu_{s}").argtypes = [vp, ci]
        L.oracle_fourier_counts.restype = ci
        L.oracle_fourier_counts.argtypes = [sz, ctypes.POINTER(sz)]
        L.oracle_fourier_table_len.restype = sz
        L.oracle_fourier_table_len.argtypes = [sz]
        L.oracle_fourier_radix_pass_double.restype = ci
        L.oracle_fourier_radix_pass_double.argtypes = [ci, vp, vp, ci, sz, sz, ci]
        L.oracle_fourier_set_clone.restype = ci
        L.oracle_fourier_set_clone.argtypes = [ci]
        L.oracle_fourier_have_avx.restype = ci
        _lib = L
    return _lib


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.complex64:
        return "float"
    if dtype == np.complex128:
        return "double"
    raise TypeError(f"oracle supports complex64/complex128, got {dtype}")


class OracleFft:
    """Mirror of the reference's `Fft` trait (fourier-algorithms/src/fft.rs:40-82) over the oracle."""

    def __init__(self, size, dtype=np.complex64):
        self.dtype = np.dtype(dtype)
        self._s = _suffix(dtype)
        self._h = getattr(lib(), f"oracle_fourier_create_{self._s}")(size)
        if not self._h:
            raise ValueError(f"oracle: cannot create plan of size {size}")
        self._n = size

    def size(self):
        return self._n

    def transform(self, x, transform=FFT):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.shape == (self._n,)
        out = np.empty_like(x)
        getattr(lib(), f"oracle_fourier_transform_{self._s}")(self._h, x.ctypes.data, out.ctypes.data, transform)
        return out

    def transform_in_place(self, x, transform=FFT):
        assert x.dtype == self.dtype and x.flags.c_contiguous and x.shape == (self._n,)
        getattr(lib(), f"oracle_fourier_transform_in_place_{self._s}")(self._h, x.ctypes.data, transform)
        return x

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            getattr(lib(), f"oracle_fourier_destroy_{self._s}")(h)


class OracleBatch:
    """`nthreads` independent plans of one size; run() splits the batch contiguously over them.

    The worker threads are persistent and pinned (spread over the process's affinity mask); each builds its own plan, so
    tables and work buffers sit on the worker's NUMA node.  Plan creation, stage() and any warm-up run happen outside the
    timed region (fourier-bench times `transform` only, fourier-bench/benches/fft_bench.rs:36).
    """

    def __init__(self, n, dtype=np.complex64, nthreads=1):
        self._s = _suffix(dtype)
        self.dtype = np.dtype(dtype)
        self.n = n
        self.nthreads = nthreads
        self._staged = 0
        self._c = getattr(lib(), f"oracle_fourier_batch_create_{self._s}")(n, nthreads)
        if not self._c:
            raise ValueError(f"oracle: cannot create plans of size {n}")

    def run(self, x, transform=FFT, out=None):
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.ndim == 2 and x.shape[1] == self.n
        if out is None:
            out = np.empty_like(x)
        getattr(lib(), f"oracle_fourier_batch_run_{self._s}")(self._c, x.ctypes.data, out.ctypes.data, x.shape[0], transform)
        return out

    def stage(self, x):
        """Copy x into a context-owned input buffer, every worker copying (first-touching) the slice it will transform."""
        x = np.ascontiguousarray(x, dtype=self.dtype)
        assert x.ndim == 2 and x.shape[1] == self.n
        if not getattr(lib(), f"oracle_fourier_batch_stage_{self._s}")(self._c, x.ctypes.data, x.shape[0]):
            raise MemoryError("oracle: cannot stage the input")
        self._staged = x.shape[0]

    def run_staged(self, out, transform=FFT):
        """Transform the staged input into `out` ((batch, n), C-contiguous)."""
        assert out.dtype == self.dtype and out.flags.c_contiguous and out.shape == (self._staged, self.n)
        getattr(lib(), f"oracle_fourier_batch_run_{self._s}")(self._c, None, out.ctypes.data, self._staged, transform)
        return out

    def cpus(self):
        """CPU every worker is pinned to (-1 = not pinned)."""
        return [getattr(lib(), f"oracle_fourier_batch_cpu_{self._s}")(self._c, t) for t in range(self.nthreads)]

    def __del__(self):
        c, self._c = getattr(self, "_c", None), None
        if c:
            getattr(lib(), f"oracle_fourier_batch_destroy_{self._s}")(c)


def transform_batch(x, transform=FFT, nthreads=1):
    """x: (batch, n) complex array -> oracle transform of every row."""
    x = np.ascontiguousarray(x)
    return OracleBatch(x.shape[1], x.dtype, nthreads).run(x, transform)


def radix_counts(size):
    c = (ctypes.c_size_t * 5)()
    ok = lib().oracle_fourier_counts(size, c)
    return list(c) if ok else None


def table_len(size):
    """Entries of one direction's twiddle table (autosort/mod.rs:24-46), None if the size does not factor."""
    if radix_counts(size) is None:
        return None
    return int(lib().oracle_fourier_table_len(size))


# which clone of the reference's pass functions the oracle runs (process-wide):
GENERIC, AVX_NO_FIRST_PASS, AVX = 0, 1, 2


def set_clone(clone):
    """0 = generic (scalar) functions, 1 = AVX clone without the hand-scheduled f32 first pass, 2 = AVX clone as the
    reference runs it on an AVX host (default).  Results are bit-identical; only the speed differs.  Returns the clone
    in effect (0 when the oracle was built without AVX)."""
    return int(lib().oracle_fourier_set_clone(int(clone)))


def have_avx():
    return bool(lib().oracle_fourier_have_avx())



def radix_pass(radix, x, forward, size, stride, clone=GENERIC):
    """One Stockham pass (autosort/mod.rs:203-284) of the restatement on a complex128 array of size*stride points."""
    x = np.ascontiguousarray(x, dtype=np.complex128)
    assert x.shape == (size * stride,)
    out = np.empty_like(x)
    ok = lib().oracle_fourier_radix_pass_double(radix, x.ctypes.data, out.ctypes.data, int(forward), size, stride, int(clone))
    assert ok
    return out
