#!/usr/bin/env python3
"""Development tool: compiles one kernel translation unit of fourier_amd/csrc to gfx950 assembly and reports, per kernel,
  (a) loops that contain a global / buffer load -- a plain `for (u = tid; u < n; u += NT) lds[u] = global[u]` compiles to load,
      s_waitcnt vmcnt(0), ds_write PER ITERATION (round 4: the copy loops of the LDS mixed-radix kernels, +4 ... 22 % once batched);
  (b) the order of its memory events: L = global / buffer load, S = store, W = s_waitcnt with a vmcnt field, d = LDS access,
      B = barrier, run-length compressed ("L16W14L8W2d62 ..."): load phases that are split by waits show up as L..W..L..W.

usage: python tools/isa_scan.py <tu.cpp> [float|double] [--seq] [extra hipcc flags ...]
  e.g. python tools/isa_scan.py kernels_pass.cpp float --seq
       python tools/isa_scan.py kernels_mixed_ct.cpp double -DFOURIER_MIX_SHARD=3"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fourier_amd", "csrc")


def demangle(names):
    return subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True).stdout.strip().split("\n") if names else []


def main(argv):
    tu, real, seq, extra = argv[0], "float", False, []
    for a in argv[1:]:
        if a in ("float", "double"):
            real = a
        elif a == "--seq":
            seq = True
        else:
            extra.append(a)
    if not any(f.startswith("-DFOURIER_MIX_SHARD") for f in extra):
        extra.append("-DFOURIER_MIX_SHARD=0")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tu.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-fno-slp-vectorize",
                               f"-DFOURIER_TU_REAL={real}", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                               os.path.join(CSRC, tu), "-o", out] + extra, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    cur, loop, loops, events = None, None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, loop = m.group(1), None
            events[cur] = []
            continue
        if cur is None:
            continue
        t = l.strip()
        if "Loop Header" in l:
            loop = (cur, i)
            loops[loop] = 0
        is_load = re.match(r"(global_load|buffer_load|flat_load)", t) is not None
        if loop and is_load:
            loops[loop] += 1
        if is_load:
            events[cur].append("L")
        elif re.match(r"(global_store|buffer_store|flat_store)", t):
            events[cur].append("S")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            events[cur].append("W")
        elif t.startswith("s_barrier"):
            events[cur].append("B")
        elif t.startswith("ds_"):
            events[cur].append("d")
    per_kernel = {}
    for (k, _), n in loops.items():
        if n:
            per_kernel[k] = per_kernel.get(k, 0) + 1
    names = list(per_kernel)
    print(f"== {tu} ({real}): kernels with loops that contain global loads: {len(names)}")
    for d, k in zip(demangle(names), names):
        print(f"  {per_kernel[k]} loop(s): {d[:140]}")
    if seq:
        names = [k for k in events if "L" in events[k]]
        for d, k in zip(demangle(names), names):
            s = "".join(events[k])
            comp = re.sub(r"(.)\1*", lambda m: m.group(1) + (str(len(m.group(0))) if len(m.group(0)) > 1 else ""), s)
            print(f"  {d[:100]} :: {comp[:260]}")


if __name__ == "__main__":
    main(sys.argv[1:])
