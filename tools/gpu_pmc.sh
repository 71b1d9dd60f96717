#!/bin/bash
# PMC evidence for the two pass kernels: LDS bank conflicts, wave-cycle breakdown (one rocprofv3 --pmc pass each).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --output-format csv -d "$R/gpurun_out/prof_lds" -o lds -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --batch 1024 > "$R/gpurun_out/prof_lds.log" 2>&1; echo "lds rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d "$R/gpurun_out/prof_sq" -o sq -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --batch 1024 > "$R/gpurun_out/prof_sq.log" 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$R/gpurun_out/prof_grbm" -o grbm -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu --batch 1024 > "$R/gpurun_out/prof_grbm.log" 2>&1; echo "grbm rc=$?"
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/prof_*/*_counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fft_pass_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_extra.json", "w"), indent=1)
PY
