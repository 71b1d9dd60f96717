#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of the last GPU session from gpurun_out/ (scratch) into profiles/
(tracked) under a per-round name, and derives profiles/traffic_latest.json (HBM bytes per launch from
the PMC passes, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 16 B/lane streaming reads at half their bytes)."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
SLOT = {"0>": "pass0", "1>": "pass1", "2>": "pass1"}


def main(tag):
    os.makedirs(P, exist_ok=True)
    for src, dst in (("prof_trace/trace_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"),
                     ("bench.json", f"{tag}_bench.json"), ("bench_f64.json", f"{tag}_bench_f64.json")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    raw = collections.defaultdict(dict)
    for name, f in (("FETCH_SIZE", "prof_fetch/fetch_counter_collection.csv"), ("WRITE_SIZE", "prof_write/write_counter_collection.csv")):
        path = os.path.join(G, f)
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "fft_pass_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            raw[k][name] = sum(v) / len(v)
            raw[k]["dispatches_" + name] = len(v)
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over `python bench.py --steps 3 "
                   "--warmup 1 --no-cpu` (batch 4096, N=2^20 f32). Counter values are KB per dispatch; "
                   "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halves 16 B/lane streams).",
           "raw_kb_per_dispatch": raw, "per_launch_bytes": {}}
    for k, d in raw.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            b = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
            out["per_launch_bytes"][k] = b
            # bench.py names kernels by slot: MODE template argument 0 = pass0 (FIRST), 2 = pass1 (LAST)
            targs = [t.strip() for t in k[k.index("<") + 1:k.index(">")].split(",")]  # <T, L, CG, MODE, IO>
            out["per_launch_bytes"]["pass0" if targs[3] == "0" else "pass1"] = b
    json.dump(out, open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(P, "traffic_latest.json"), "w"), indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
