#!/bin/bash
# Round 3: HBM-side traffic counters for the BASELINE configurations that had none (C3, C4, C5 chunk), per kernel.
# Separate rocprofv3 --pmc passes (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; MI355X_MICROARCH.md), no trace
# domains next to --pmc.  Output: gpurun_out/pmc_<cfg>_<set>/..., summary gpurun_out/pmc_traffic_<cfg>.json
# (the correction of /opt/skills/guides/MI355X_MICROARCH.md: KB units, FETCH_SIZE x 2 on gfx950 for 16 B/lane streams).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
export TMPDIR=/tmp
cd /tmp
CFGS=${PMC_CFGS:-"c4 999983 512 f32 2;c5chunk 4194304 1024 f32 2;c3 1048576 4096 f64 2;c2 1048576 4096 f32 2"}
IFS=';' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  for cs in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "hit:TCC_HIT_sum TCC_MISS_sum" "req:TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum"; do
    name=${cs%%:*}; ctrs=${cs#*:}
    timeout 400 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc_$1_$name" -o "$name" -- python "$R/tools/run_config.py" $2 $3 $4 $5 > "$R/gpurun_out/pmc_$1_$name.log" 2>&1
    echo "$1 $name rc=$?"
  done
done
cd "$R"
python - <<'PY'
import csv, collections, glob, json, os
for d in sorted(set(p.split("/")[1].rsplit("_", 1)[0] for p in glob.glob("gpurun_out/pmc_*_*/"))):
    cfg = d[len("pmc_"):]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_{cfg}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("fourier_hip::", "").replace("(fourier_hip::PassArgs)", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"note": "rocprofv3 --pmc, one counter set per pass, averages per dispatch; FETCH_SIZE/WRITE_SIZE in KB; "
                   "hbm_side_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)", "kernels": {}}
    for k, c in acc.items():
        row = {n: sum(v) / len(v) for n, v in c.items()}
        row["dispatches"] = max(len(v) for v in c.values())
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            row["read_bytes"] = 2 * row["FETCH_SIZE"] * 1024
            row["write_bytes"] = row["WRITE_SIZE"] * 1024
            row["hbm_side_bytes"] = row["read_bytes"] + row["write_bytes"]
        if "TCC_HIT_sum" in row and "TCC_MISS_sum" in row and row["TCC_HIT_sum"] + row["TCC_MISS_sum"] > 0:
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
        out["kernels"][k] = row
    json.dump(out, open(f"gpurun_out/pmc_traffic_{cfg}.json", "w"), indent=1)
    print(cfg, json.dumps({k: {n: round(v, 4) if isinstance(v, float) and v < 10 else round(v) for n, v in r.items()} for k, r in out["kernels"].items()}, indent=1))
PY
