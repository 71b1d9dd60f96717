#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.

usage: python tools/resource_usage.py [--from remarks.txt] [extra hipcc flags...]  > table
Used to check that a refactor of the kernel headers leaves the hot kernels' allocation unchanged."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourier_amd import build as B

PAT = re.compile(r"remark: (?:Function Name: (\S+)|\s*(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+))")

def main():
    argv = sys.argv[1:]
    if argv[:1] == ["--from"]:
        err = open(argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as td:  # every translation unit once more, remarks collected per object
            B.compile_objects(td, ["-Rpass-analysis=kernel-resource-usage"] + argv, force=True)
            err = "".join(open(os.path.join(td, f)).read() for f in sorted(os.listdir(td)) if f.endswith(".remarks.txt"))
    rows, cur = [], None
    for line in err.splitlines():
        m = PAT.search(line)
        if not m:
            continue
        if m.group(1):
            cur = {"name": m.group(1)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(2).split(" ")[0]] = int(m.group(3))
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], stdout=subprocess.PIPE, text=True).stdout.strip().split("\n")
    for r, name in zip(rows, names):
        name = name.replace("fourier_hip::", "").replace("(PassArgs)", "").replace("void ", "")
        print(f"{name:70s} vgpr={r.get('VGPRs',0):3d} agpr={r.get('AGPRs',0):3d} sgpr={r.get('SGPRs',0):3d} scratch={r.get('ScratchSize',0):4d} occ={r.get('Occupancy',0)}")

if __name__ == "__main__":
    main()
