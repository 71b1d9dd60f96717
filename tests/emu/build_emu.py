"""TEST-ONLY: builds the engine sources against tests/emu/hipemu.h (CPU fibers) so kernel logic,
plan logic and the C ABI can be exercised without a GPU.  Not a product path.

The same translation units as fourier_amd/build.py (host logic + one object per kernel family and precision), compiled
with g++ -DFOURIER_EMU on all cores, linked with the experiments switches (env_experiments.cpp + kernels_experiments.cpp):
the CPU tests drive the alternative plans through environment variables."""
import concurrent.futures
import ctypes
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "fourier_amd", "csrc")
# FOURIER_EMU_ASAN=1: the same library instrumented with AddressSanitizer (tools/asan_emu.sh: every global, LDS and table access of the
# emulated kernels and every host-side buffer of the plan layer is bounds-checked while the CPU tests run)
ASAN = os.environ.get("FOURIER_EMU_ASAN") in ("1", "2")  # 2: + UndefinedBehaviorSanitizer (reports go to stderr, the run continues)
UBSAN = os.environ.get("FOURIER_EMU_ASAN") == "2"
OUT = os.path.join(HERE, ("libfourier_emu_asan_ubsan.so" if UBSAN else "libfourier_emu_asan.so") if ASAN else "libfourier_emu.so")
OBJDIR = os.path.join(HERE, ("obj_asan_ubsan" if UBSAN else "obj_asan") if ASAN else "obj")
SAN = ["-fsanitize=address" + (",undefined" if UBSAN else ""), "-fno-omit-frame-pointer", "-g1"] if ASAN else []


def deps():
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.cpp")) +
            [os.path.join(HERE, "hipemu.h"), os.path.join(ROOT, "include", "fourier.h")])


def compile_and_link(out):
    from fourier_amd import build as B

    os.makedirs(OBJDIR, exist_ok=True)
    newest_header = max(os.path.getmtime(d) for d in deps() if d.endswith(".h"))
    jobs, objs = [], []
    for name, src, defs, group in B.translation_units():
        if group in B.PRODUCT_ONLY:
            continue
        srcp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, name + ".o")
        objs.append(obj)
        if os.path.exists(obj) and os.path.getmtime(obj) >= max(newest_header, os.path.getmtime(srcp)):
            continue
        jobs.append(["g++", "-O1" if ASAN else "-O2", "-std=c++17", "-DFOURIER_EMU", "-include", os.path.join(HERE, "hipemu.h"), "-fPIC", "-pthread"] + SAN + defs +
                    ["-c", srcp, "-o", obj])
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1)) as ex:
        for rc in ex.map(lambda cmd: subprocess.call(cmd), jobs):
            if rc:
                raise RuntimeError("emulator build failed")
    subprocess.check_call(["g++", "-shared", "-pthread"] + SAN + ["-o", out] + objs)


def build():
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps())  # noqa: E731
    if fresh():
        return OUT
    import fcntl

    with open(OUT + ".lock", "w") as lock:  # pytest-xdist workers: one builds, the others wait and find it fresh
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not fresh():
            tmp = OUT + f".{os.getpid()}.tmp"
            compile_and_link(tmp)
            os.replace(tmp, OUT)
    return OUT


def load():
    from fourier_amd import _lib

    cdll = _lib.bind(ctypes.CDLL(build()))
    cdll.fourier_emu_lds_stats.restype = None
    cdll.fourier_emu_lds_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 3 + [ctypes.c_int]
    return cdll
