"""TEST-ONLY: builds the engine sources against tests/emu/hipemu.h (CPU fibers) so kernel logic,
plan logic and the C ABI can be exercised without a GPU.  Not a product path."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(ROOT, "fourier_amd", "csrc", "engine.cpp")
DEPS = [SRC, os.path.join(ROOT, "fourier_amd", "csrc", "fft_kernels.h"), os.path.join(HERE, "hipemu.h"),
        os.path.join(ROOT, "include", "fourier.h")]
OUT = os.path.join(HERE, "libfourier_emu.so")


def build():
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS)  # noqa: E731
    if fresh():
        return OUT
    import fcntl

    with open(OUT + ".lock", "w") as lock:  # pytest-xdist workers: one builds, the others wait and find it fresh
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not fresh():
            tmp = OUT + f".{os.getpid()}.tmp"
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFOURIER_EMU", "-DFOURIER_EXPERIMENTS", "-include", os.path.join(HERE, "hipemu.h"),
                                   "-shared", "-fPIC", "-pthread", "-o", tmp, SRC])
            os.replace(tmp, OUT)
    return OUT


def load():
    from fourier_amd import _lib

    cdll = _lib.bind(ctypes.CDLL(build()))
    cdll.fourier_emu_lds_stats.restype = None
    cdll.fourier_emu_lds_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 3 + [ctypes.c_int]
    return cdll
