#!/bin/bash
# Round 6, session 65: rocprofv3 kernel-trace statistics and PMC traffic (FETCH_SIZE / WRITE_SIZE / L2 hit / requests, one counter set per pass) of
# the register-stage transforms: 5005 f64 (35 x 13 x 11, factored tables), 15625 f32 (25^3, one transform per workgroup), 1001 f32 (13 x 11 x 7,
# packed pairs), 700 f64 (28 x 25) -- average launch duration, achieved fraction of the HBM peak on the algorithmic bytes, HBM-side bytes per launch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
export TMPDIR=/tmp
CF="r5005f64 5005 26816 f64 6;r15625f32 15625 17179 f32 6;r1001f32 1001 268167 f32 6;r700f64 700 95869 f64 6"
cd /tmp
IFS=';' read -ra LIST <<< "$CF"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$1" -o "$1" -- python "$R/tools/run_config.py" $2 $3 $4 $5 > "$R/gpurun_out/prof_$1.log" 2>&1
  echo "$1 rc=$?"; f=$(find "$R/gpurun_out/prof_$1" -name "*kernel_stats.csv" | head -1); head -3 "$f" | cut -c1-170; cp "$f" "$R/gpurun_out/r06_s65_$1_kernel_stats.csv"
done
cd "$R"
PMC_CFGS="r5005f64 5005 26816 f64 2;r15625f32 15625 17179 f32 2;r1001f32 1001 268167 f32 2;r700f64 700 95869 f64 2" bash tools/gpu_r03_pmc.sh > gpurun_out/r06_s65_pmc.log 2>&1
grep "rc=" gpurun_out/r06_s65_pmc.log | tr '\n' ' '
for c in r5005f64 r15625f32 r1001f32 r700f64; do cp gpurun_out/pmc_traffic_$c.json gpurun_out/r06_s65_pmc_traffic_$c.json; done
python - <<'PY'
import csv, glob, json
alg = {"r5005f64": 2 * 5005 * 16 * 26816, "r15625f32": 2 * 15625 * 8 * 17179, "r1001f32": 2 * 1001 * 8 * 268167, "r700f64": 2 * 700 * 16 * 95869}
out = {}
for c, b in alg.items():
    rows = [r for r in csv.DictReader(open(f"gpurun_out/r06_s65_{c}_kernel_stats.csv")) if "regfft" in r["Name"]]
    t = json.load(open(f"gpurun_out/r06_s65_pmc_traffic_{c}.json"))["kernels"]
    k = next(v for n, v in t.items() if "regfft" in n)
    ns = float(rows[0]["AverageNs"])
    out[c] = {"kernel": rows[0]["Name"], "calls": int(rows[0]["Calls"]), "avg_us": round(ns / 1e3, 2), "algorithmic_bytes": b, "achieved_gbps": round(b / ns, 1),
              "frac_of_8tbps": round(b / ns / 8000, 4), "hbm_side_bytes": round(k.get("hbm_side_bytes", 0)), "traffic_over_algorithmic": round(k.get("hbm_side_bytes", 0) / b, 4),
              "l2_hit_rate": round(k.get("l2_hit_rate", 0), 4)}
json.dump(out, open("gpurun_out/r06_s65_regfft_roofline.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
