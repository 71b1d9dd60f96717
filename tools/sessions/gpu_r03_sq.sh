#!/bin/bash
# Round 3: where the waves of the kernels below the streaming ceiling spend their cycles -- SQ wave-cycle breakdown, LDS and
# VMEM instruction counters, GPU-active cycles (one rocprofv3 --pmc pass per group, no trace domain beside it) for C4
# (conv, chirp-in, chirp-out), the C5 chunk (narrow first pass, one-workgroup-per-CU last pass), C2 (reference point)
# and the one-launch 2^14 / 2^15 plans.  Output: gpurun_out/sq_breakdown.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
export TMPDIR=/tmp
cd /tmp
for cfg in "c4 999983 512 f32 2" "c5chunk 4194304 1024 f32 2" "c2 1048576 4096 f32 2" "p14 16384 65536 f32 2" "p15 32768 32768 f32 2"; do
  set -- $cfg
  for cs in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
            "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
    name=${cs%%:*}; ctrs=${cs#*:}
    timeout 400 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/sq_$1_$name" -o "$name" -- python "$R/tools/run_config.py" $2 $3 $4 $5 > "$R/gpurun_out/sq_$1_$name.log" 2>&1
    echo "$1 $name rc=$?"
  done
done
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/sq_*_*/")):
    cfg = d.split("/")[1][len("sq_"):].rsplit("_", 1)[0]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("fourier_hip::", "").replace("(fourier_hip::PassArgs)", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            out.setdefault(cfg, {}).setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
for cfg, ks in out.items():
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if n in c:
                    c["frac_" + n] = round(c[n] / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "per-dispatch averages; SQ_* cycle counters are quad-cycles summed over waves (MI355X_MICROARCH.md); frac_* = counter / SQ_WAVE_CYCLES",
           "configs": out}, open("gpurun_out/sq_breakdown.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        print(cfg, k[:60], {n: v for n, v in c.items() if n.startswith("frac_") or n == "lds_conflict_frac"})
PY
