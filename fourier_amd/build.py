"""Builds fourier_amd/lib/libfourier.so (+ the static archive libfourier.a, as the reference's CMake package ships
both: fourier-ffi/CMakeLists.txt:38-65) and lib/libfourier_experiments.so for gfx950 with hipcc, in-tree, so that they
travel to the GPU box.

The engine is a set of translation units (csrc/): the host logic + C ABI (engine.cpp) and one object per kernel family
and precision (kernels_*.cpp with -DFOURIER_TU_REAL=float|double; the per-length mixed-radix kernels in
FOURIER_MIX_SHARDS shards each).  They compile side by side on all cores; BOTH libraries link the same objects and differ
in two of them: the product links env_product.o (no development switches, no experiment kernels), the experiments
library links env_experiments.o + kernels_experiments_*.o (csrc/engine_common.h)."""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SRC = os.path.join(CSRC, "engine.cpp")  # the host translation unit (tools name it)
OUT = os.path.join(LIBDIR, "libfourier.so")
STATIC = os.path.join(LIBDIR, "libfourier.a")
# The same objects plus the measured-slower designs (XCD-fused one-launch plan, half-tile last pass) and the environment
# switches that select alternative plans.  Loaded only by the GPU tests of those designs and by A/B tools; never by the
# operator layer (fourier_amd/_lib.py binds libfourier.so).
OUT_EXPERIMENTS = os.path.join(LIBDIR, "libfourier_experiments.so")
OBJDIR = os.path.join(LIBDIR, "obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
LINK_FLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared"]
SONAME = "-Wl,-soname,libfourier.so.0"
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result",
          # SLP-packing f32 math into v_pk_* ops doubles the live register set of the butterflies (222 vs 104
          # VGPRs on the 1024-point pass) and costs a workgroup per CU; keep scalar f32 VALU ops
          "-fno-slp-vectorize"]
FLAGS = CFLAGS  # older tools import this name


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "fourier.h")]


def gen_rtc_sources():
    """csrc/rtc_sources.inc: the device headers of a run-time specialisation, embedded as string literals (tools/gen_rtc_sources.py)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import gen_rtc_sources

    return gen_rtc_sources.main()


def mix_shards(name="MIX"):
    with open(os.path.join(CSRC, "engine_common.h")) as f:
        return int(re.search(r"#define FOURIER_%s_SHARDS (\d+)" % name, f.read()).group(1))


def translation_units():
    """(object name, source file, extra -D flags, group) for every object of the two libraries.  Groups: 'host', 'env_product',
    'env_experiments', 'pass', 'onelaunch', 'misc', 'mixed', 'chirpz', 'experiments' (tools/build_variants.py rebuilds by group)."""
    tus = [("engine", "engine.cpp", [], "host"),
           ("rtc", "rtc.cpp", [], "host"),
           ("env_product", "env_product.cpp", [], "env_product"),
           ("env_experiments", "env_experiments.cpp", [], "env_experiments"),
           ("exp_copy_ceiling", "exp_copy_ceiling.cpp", [], "experiments")]
    for real, tag in (("float", "f32"), ("double", "f64")):
        d = [f"-DFOURIER_TU_REAL={real}"]
        for i in range(mix_shards()):  # the longest compilations first
            tus.append((f"kernels_mixed_ct_{tag}_{i}", "kernels_mixed_ct.cpp", d + [f"-DFOURIER_MIX_SHARD={i}"], "mixed"))
        for i in range(mix_shards("REGFFT")):
            tus.append((f"kernels_regfft_{tag}_{i}", "kernels_regfft.cpp", d + [f"-DFOURIER_REGFFT_SHARD={i}"], "chirpz"))
    for real, tag in (("float", "f32"), ("double", "f64")):
        d = [f"-DFOURIER_TU_REAL={real}"]
        tus.append((f"kernels_pass_{tag}", "kernels_pass.cpp", d, "pass"))
        tus.append((f"kernels_onelaunch_{tag}", "kernels_onelaunch.cpp", d, "onelaunch"))
        tus.append((f"kernels_mixed_rt_{tag}", "kernels_mixed_rt.cpp", d, "mixed"))
        tus.append((f"kernels_misc_{tag}", "kernels_misc.cpp", d, "misc"))
        for i in range(4):
            tus.append((f"kernels_tiled_{tag}_{i}", "kernels_tiled.cpp", d + [f"-DFOURIER_TILED_SHARD={i}"], "mixed"))
            tus.append((f"kernels_regtile_{tag}_{i}", "kernels_regtile.cpp", d + [f"-DFOURIER_TILED_SHARD={i}"], "mixed"))
            tus.append((f"kernels_chirpz_{tag}_{i}", "kernels_chirpz.cpp", d + [f"-DFOURIER_TILED_SHARD={i}"], "chirpz"))
        tus.append((f"kernels_experiments_{tag}", "kernels_experiments.cpp", d, "experiments"))
        tus.append((f"kernels_skeleton_{tag}", "kernels_skeleton.cpp", d, "experiments"))
    return tus


PRODUCT_ONLY, EXPERIMENTS_ONLY = {"env_product"}, {"env_experiments", "experiments"}


def deps_of(obj):
    """Files the object was compiled from, as hipcc's -MD dependency file beside it lists them (None: unknown)."""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    with open(d) as f:
        text = f.read().replace("\\\n", " ")
    files = [t for t in text.split(":", 1)[-1].split() if t.startswith(os.path.dirname(HERE))]
    return files or None


def compile_objects(objdir=OBJDIR, extra=(), force=False, groups=None, verbose=False):
    """Compiles every translation unit whose object is older than one of the files it was built from (its source and the
    headers it includes, from the compiler's own dependency list; every header when that list is missing) -- or all with
    force -- into objdir, on all cores.  Returns ({object name: path}, number compiled).  groups: restrict to these groups
    (the others must exist already)."""
    os.makedirs(objdir, exist_ok=True)
    gen_rtc_sources()
    all_headers = headers()
    jobs, objs = [], {}
    for name, src, defs, group in translation_units():
        obj = os.path.join(objdir, name + ".o")
        objs[name] = obj
        if groups is not None and group not in groups:
            continue
        srcp = os.path.join(CSRC, src)
        if not force and os.path.exists(obj):
            deps = deps_of(obj) or (all_headers + [srcp])
            if all(os.path.exists(d) for d in deps) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps):
                continue
        jobs.append((name, [HIPCC] + CFLAGS + list(extra) + defs + ["-MD", "-MF", obj[:-2] + ".d", "-c", srcp, "-o", obj]))

    def run(job):
        t0 = time.time()
        p = subprocess.run(job[1], stderr=subprocess.PIPE, text=True)
        return job[0], p.returncode, p.stderr, time.time() - t0

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, os.cpu_count() or 1)) as ex:
            for name, rc, err, dt in ex.map(run, jobs):
                if verbose or rc:
                    sys.stderr.write(f"[build] {name}: {dt:.1f} s rc={rc}\n")
                if rc:
                    sys.stderr.write(err)
                    raise subprocess.CalledProcessError(rc, name)
                if err.strip() and "-Rpass-analysis=kernel-resource-usage" in extra:
                    with open(os.path.join(objdir, name + ".remarks.txt"), "w") as f:
                        f.write(err)
    return objs, len(jobs)


def group_of():
    return {name: group for name, _, _, group in translation_units()}


def link(objs, out, experiments, soname=True):
    g = group_of()
    skip = PRODUCT_ONLY if experiments else EXPERIMENTS_ONLY
    members = [p for n, p in objs.items() if g[n] not in skip]
    subprocess.check_call([HIPCC] + LINK_FLAGS + ([SONAME] if soname else []) + members + ["-ldl", "-o", out])  # rtc.cpp: dlopen (libhiprtc, lazily)
    return members


def build(force=False, extra=(), verbose=False):
    """The product library + static archive.  Incremental by object unless force."""
    os.makedirs(LIBDIR, exist_ok=True)
    product_groups = set(group_of().values()) - EXPERIMENTS_ONLY
    objs, compiled = compile_objects(OBJDIR, extra, force, groups=product_groups, verbose=verbose)
    g = group_of()
    needed = [p for n, p in objs.items() if g[n] in product_groups]
    stale = not (os.path.exists(OUT) and os.path.exists(STATIC)) or any(os.path.getmtime(p) > os.path.getmtime(OUT) for p in needed)
    if compiled or stale or force:
        members = link(objs, OUT, experiments=False)
        if os.path.exists(STATIC):
            os.remove(STATIC)
        subprocess.check_call(["ar", "rcs", STATIC] + members)  # consumers link it with -lamdhip64 -lstdc++ (packaging/CMakeLists.txt)
    link_soname()
    return OUT


def build_experiments(force=False, extra=(), verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    groups = set(group_of().values()) - PRODUCT_ONLY
    objs, compiled = compile_objects(OBJDIR, extra, force, groups=groups, verbose=verbose)
    g = group_of()
    needed = [p for n, p in objs.items() if g[n] in groups]
    stale = not os.path.exists(OUT_EXPERIMENTS) or any(os.path.getmtime(p) > os.path.getmtime(OUT_EXPERIMENTS) for p in needed)
    if compiled or stale or force:
        link(objs, OUT_EXPERIMENTS, experiments=True, soname=False)
    return OUT_EXPERIMENTS


def build_all(force=False, verbose=False):
    """Product library and experiments library: ONE compilation of the shared objects (all cores), three link steps."""
    t0 = time.time()
    _, compiled = compile_objects(OBJDIR, (), force, verbose=verbose)
    out = build(False, verbose=verbose)
    build_experiments(False, verbose=verbose)
    if verbose:
        sys.stderr.write(f"[build] {compiled} translation units compiled, {time.time() - t0:.1f} s wall\n")
    return out


def link_soname():
    """libfourier.so.0 -> libfourier.so, the name consumers' DT_NEEDED carries (fourier-ffi/CMakeLists.txt:55)."""
    so0 = OUT + ".0"
    if not os.path.lexists(so0):
        os.symlink(os.path.basename(OUT), so0)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--all" in sys.argv:
        print(build_all(force=force, verbose=True))
    else:
        print(build(force=force, extra=[a for a in sys.argv[1:] if a not in ("--force",)], verbose=True))
