#!/bin/bash
# PMC evidence for the Bluestein kernels of C4 (conv, chirp-in first pass, chirp-out last pass): wave-cycle
# breakdown and LDS counters, one rocprofv3 --pmc pass per group.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d "$R/gpurun_out/pmc_conv_sq" -o sq -- python "$R/tools/run_config.py" 999983 512 f32 3 > "$R/gpurun_out/pmc_conv_sq.log" 2>&1; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM --output-format csv -d "$R/gpurun_out/pmc_conv_lds" -o lds -- python "$R/tools/run_config.py" 999983 512 f32 3 > "$R/gpurun_out/pmc_conv_lds.log" 2>&1; echo "lds rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$R/gpurun_out/pmc_conv_grbm" -o grbm -- python "$R/tools/run_config.py" 999983 512 f32 3 > "$R/gpurun_out/pmc_conv_grbm.log" 2>&1; echo "grbm rc=$?"
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_conv_*/*_counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fourier_hip" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_conv.json", "w"), indent=1)
PY
