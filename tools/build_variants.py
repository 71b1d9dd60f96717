#!/usr/bin/env python3
"""Builds A/B variants of libfourier.so (different compile-time knobs) into fourier_amd/lib/variants/
so one GPU session can time them side by side (tools/gpu_sweep.py).  Development tool only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourier_amd import build as B  # noqa: E402

# The compile-time knobs of rounds 1 - 5 (cache policies, tile widths, split thresholds, launch bounds, the LDS mixed-radix schedule, ...) are
# settled and gone from the headers: profiles/r06_removed_ab_knobs.patch (apply with `git apply -R`) brings them back together with their
# variants below it in the history of this file.  What is left are builds that differ by a compiler flag or by one of the few live switches.
VARIANTS = {
    "base": [],
    "slp": ["-fslp-vectorize"],
    # ablations (timing only, wrong results): experiments translation units, their templates in the inline namespace `ablated`
    "abl1": ["-DFOURIER_EXPERIMENTS_TU=1", "-DFOURIER_ABLATE=1"],
    "abl2": ["-DFOURIER_EXPERIMENTS_TU=1", "-DFOURIER_ABLATE=2"],
    "abl3": ["-DFOURIER_EXPERIMENTS_TU=1", "-DFOURIER_ABLATE=3"],
    "abl4": ["-DFOURIER_EXPERIMENTS_TU=1", "-DFOURIER_ABLATE=4"],  # stage twiddles from a constant (no table loads inside the in-tile transform)
    "abl5": ["-DFOURIER_EXPERIMENTS_TU=1", "-DFOURIER_ABLATE=5"],  # per-thread inter-pass twiddle factor from a constant (no two-level look-up)
}


def groups_of(flags):
    """Translation-unit groups (fourier_amd/build.py) a variant's flags can reach: the LDS mixed-radix knobs touch only the
    'mixed' objects, everything else only the tile / one-launch / small kernels; the other objects come from the base build."""
    g = set()
    mixed = any("MIX" in f for f in flags)
    other = any("MIX" not in f and not f.startswith("-f") for f in flags)  # a code-generation flag follows the knobs it comes with
    if mixed:
        g |= {"mixed"}
    if other or not mixed:
        g |= {"pass", "onelaunch", "misc"}
    return g


def main(names):
    outdir = os.path.join(ROOT, "fourier_amd", "lib", "variants")
    os.makedirs(outdir, exist_ok=True)
    base_objs, _ = B.compile_objects(B.OBJDIR, (), False, groups=set(B.group_of().values()) - B.EXPERIMENTS_ONLY)  # the shared objects
    group = B.group_of()
    for name in names:
        flags = VARIANTS[name]
        saved = list(B.CFLAGS)
        if name == "slp":
            B.CFLAGS[:] = [f for f in B.CFLAGS if f != "-fno-slp-vectorize"]
        try:
            objdir = os.path.join(outdir, "obj_" + name)
            groups = groups_of(flags)
            objs, _ = B.compile_objects(objdir, flags, False, groups=groups)
            members = {n: (objs[n] if group[n] in groups else base_objs[n]) for n in objs}
            out = os.path.join(outdir, f"libfourier_{name}.so")
            B.link(members, out, experiments=False, soname=False)
            print(name, "ok", sorted(groups))
        except Exception as e:  # noqa: BLE001
            print(name, f"FAILED {e!r}")
        finally:
            B.CFLAGS[:] = saved


if __name__ == "__main__":
    main(sys.argv[1:] or list(VARIANTS))
