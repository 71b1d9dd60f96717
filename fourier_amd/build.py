"""Builds fourier_amd/lib/libfourier.so for gfx950 with hipcc (in-tree, so it travels to the GPU box)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "engine.cpp")
DEPS = [SRC, os.path.join(HERE, "csrc", "fft_kernels.h"), os.path.join(os.path.dirname(HERE), "include", "fourier.h")]
OUT = os.path.join(HERE, "lib", "libfourier.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
         "-Wl,-soname,libfourier.so.0", "-Wno-unused-result",
         # SLP-packing f32 math into v_pk_* ops doubles the live register set of the butterflies (222 vs 104
         # VGPRs on the 1024-point pass) and costs a workgroup per CU; keep scalar f32 VALU ops
         "-fno-slp-vectorize"]


def build(force=False, extra=()):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        link_soname()
        return OUT
    cmd = [HIPCC] + FLAGS + list(extra) + [SRC, "-o", OUT]
    subprocess.check_call(cmd)
    link_soname()
    return OUT


def link_soname():
    """libfourier.so.0 -> libfourier.so, the name consumers' DT_NEEDED carries (fourier-ffi/CMakeLists.txt:55)."""
    so0 = OUT + ".0"
    if not os.path.lexists(so0):
        os.symlink(os.path.basename(OUT), so0)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a != "--force"]))
