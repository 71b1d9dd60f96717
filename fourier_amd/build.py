"""Builds fourier_amd/lib/libfourier.so (+ the static archive libfourier.a, as the reference's CMake package ships
both: fourier-ffi/CMakeLists.txt:38-65) for gfx950 with hipcc, in-tree, so that they travel to the GPU box.
One compilation (engine.o), two link steps."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "engine.cpp")
DEPS = [SRC, os.path.join(HERE, "csrc", "fft_kernels.h"), os.path.join(os.path.dirname(HERE), "include", "fourier.h")]
OUT = os.path.join(HERE, "lib", "libfourier.so")
OBJ = os.path.join(HERE, "lib", "engine.o")
STATIC = os.path.join(HERE, "lib", "libfourier.a")
# The same sources with -DFOURIER_EXPERIMENTS: the measured-slower designs (XCD-fused one-launch plan, half-tile last pass)
# and the environment switches that select alternative plans.  Loaded only by the GPU tests of those designs and by A/B
# tools; never by the operator layer (fourier_amd/_lib.py binds libfourier.so).
OUT_EXPERIMENTS = os.path.join(HERE, "lib", "libfourier_experiments.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LINK_FLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,-soname,libfourier.so.0"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
         "-Wl,-soname,libfourier.so.0", "-Wno-unused-result",
         # SLP-packing f32 math into v_pk_* ops doubles the live register set of the butterflies (222 vs 104
         # VGPRs on the 1024-point pass) and costs a workgroup per CU; keep scalar f32 VALU ops
         "-fno-slp-vectorize"]


def build(force=False, extra=()):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    fresh = lambda f: os.path.exists(f) and all(os.path.getmtime(f) >= os.path.getmtime(d) for d in DEPS)  # noqa: E731
    if not force and fresh(OUT) and fresh(STATIC):
        link_soname()
        return OUT
    compile_flags = [f for f in FLAGS if f != "-shared" and not f.startswith("-Wl,")]
    subprocess.check_call([HIPCC] + compile_flags + list(extra) + ["-c", SRC, "-o", OBJ])
    subprocess.check_call([HIPCC] + LINK_FLAGS + [OBJ, "-o", OUT])
    if os.path.exists(STATIC):
        os.remove(STATIC)
    subprocess.check_call(["ar", "rcs", STATIC, OBJ])  # consumers link it with -lamdhip64 -lstdc++ (packaging/CMakeLists.txt)
    os.remove(OBJ)
    link_soname()
    return OUT


def build_experiments(force=False):
    fresh = os.path.exists(OUT_EXPERIMENTS) and all(os.path.getmtime(OUT_EXPERIMENTS) >= os.path.getmtime(d) for d in DEPS)
    if not force and fresh:
        return OUT_EXPERIMENTS
    os.makedirs(os.path.dirname(OUT_EXPERIMENTS), exist_ok=True)
    flags = [f for f in FLAGS if not f.startswith("-Wl,-soname")]
    subprocess.check_call([HIPCC] + flags + ["-DFOURIER_EXPERIMENTS", SRC, "-o", OUT_EXPERIMENTS])
    return OUT_EXPERIMENTS


def build_all(force=False):
    """Product library and experiments library side by side (two independent compilations, ~2 minutes together)."""
    import threading

    err = []

    def side():
        try:
            build_experiments(force)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=side)
    t.start()
    out = build(force)
    t.join()
    if err:
        raise err[0]
    return out


def link_soname():
    """libfourier.so.0 -> libfourier.so, the name consumers' DT_NEEDED carries (fourier-ffi/CMakeLists.txt:55)."""
    so0 = OUT + ".0"
    if not os.path.lexists(so0):
        os.symlink(os.path.basename(OUT), so0)


if __name__ == "__main__":
    if "--all" in sys.argv:
        print(build_all(force="--force" in sys.argv))
    else:
        print(build(force="--force" in sys.argv, extra=[a for a in sys.argv[1:] if a != "--force"]))
