#!/bin/bash
# Round 6, session 22: SQ counters and per-dispatch times of the register-resident tile passes (44100, 100000, 32000 f32; 15625, 100000 f64).
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
run() {  # n batch real
  local n=$1 b=$2 real=$3
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r06_s22_stats_${real}_$n" -o s -- python "$R/tools/run_config.py" $n $b $real 5 > "$R/gpurun_out/r06_s22_stats_${real}_$n.log" 2>&1
  for cs in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    name=${cs%%:*}; ctrs=${cs#*:}
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/r06_s22_sq_${real}_${n}_$name" -o "$name" -- python "$R/tools/run_config.py" $n $b $real 2 > "$R/gpurun_out/r06_s22_sq_${real}_${n}_$name.log" 2>&1
    echo "sq $real $n $name rc=$?"
  done
}
run 44100 8192 f32
run 100000 4096 f32
run 32000 8192 f32
run 15625 16384 f64
run 100000 2048 f64
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/r06_s22_sq_*_*_*/")):
    cfg = d.split("/")[1][len("r06_s22_sq_"):].rsplit("_", 1)[0]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("fourier_hip::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            out.setdefault(cfg, {}).setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
for d in sorted(glob.glob("gpurun_out/r06_s22_stats_*/")):
    cfg = d.split("/")[1][len("r06_s22_stats_"):]
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        seq = [(int(r["Start_Timestamp"]), r["Kernel_Name"].replace("fourier_hip::", "").split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
               for r in csv.DictReader(open(f)) if "fourier_hip" in r["Kernel_Name"]]
        seq.sort()
        out.setdefault(cfg, {})["dispatch_us_in_order"] = [(k, round(t, 1)) for _, k, t in seq]
for cfg, ks in out.items():
    for k, c in ks.items():
        if k == "dispatch_us_in_order":
            continue
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if n in c:
                    c["frac_" + n] = round(c[n] / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "per-dispatch averages; frac_* = counter / SQ_WAVE_CYCLES", "configs": out}, open("gpurun_out/r06_s22_sq_regtile.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        if k == "dispatch_us_in_order":
            print(cfg, k, c[-6:])
        else:
            print(cfg, k[:80], {n: (round(v, 4) if n.startswith("frac") or n.startswith("lds_c") else int(v)) for n, v in c.items()})
PY
