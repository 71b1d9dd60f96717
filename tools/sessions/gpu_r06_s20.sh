#!/bin/bash
# Round 6, session 20: where do the mixed-length tile passes lose their time?  SQ counters (waits, VALU, LDS conflicts) and kernel stats of
# 44100 = 210 x 210, 100000 = 400 x 250 (f32), 13122 = 162 x 81 (f64), beside a power-of-two tile plan (2^16 = 256 x 256) of the same footprint.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() {  # n batch real
  local n=$1 b=$2 real=$3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_s20_stats_${real}_$n" -o s -- python "$R/tools/run_config.py" $n $b $real 5 > "$R/gpurun_out/r06_s20_stats_${real}_$n.log" 2>&1
  for cs in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
            "occ:SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    name=${cs%%:*}; ctrs=${cs#*:}
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/r06_s20_sq_${real}_${n}_$name" -o "$name" -- python "$R/tools/run_config.py" $n $b $real 2 > "$R/gpurun_out/r06_s20_sq_${real}_${n}_$name.log" 2>&1
    echo "sq $real $n $name rc=$?"
  done
}
run 44100 8192 f32
run 100000 4096 f32
run 65536 8192 f32
run 13122 16384 f64
run 16384 16384 f64
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/r06_s20_sq_*_*_*/")):
    cfg = d.split("/")[1][len("r06_s20_sq_"):].rsplit("_", 1)[0]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("fourier_hip::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            out.setdefault(cfg, {}).setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
for d in sorted(glob.glob("gpurun_out/r06_s20_stats_*/")):
    cfg = d.split("/")[1][len("r06_s20_stats_"):]
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Name"]:
                out.setdefault(cfg, {}).setdefault(r["Name"].replace("fourier_hip::", "").split("(")[0], {})["avg_us"] = float(r["AverageNs"]) / 1e3
for cfg, ks in out.items():
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if n in c:
                    c["frac_" + n] = round(c[n] / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "per-dispatch averages; frac_* = counter / SQ_WAVE_CYCLES", "configs": out}, open("gpurun_out/r06_s20_sq_tiled.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        print(cfg, k[:80], {n: (round(v, 4) if n.startswith("frac") or n.startswith("lds_c") or n == "avg_us" else int(v)) for n, v in c.items()})
PY
