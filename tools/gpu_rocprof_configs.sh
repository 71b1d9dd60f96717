#!/bin/bash
# rocprofv3 kernel-trace summaries for the other BASELINE configurations (C3 f64, C4 Bluestein, C5 chunk), next to
# the HIP-event figures of tools/gpu_c4c5.py.  Outputs: gpurun_out/prof_<cfg>/<cfg>_kernel_stats.csv
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
R="$PWD"
cd /tmp
for cfg in "c3 1048576 4096 f64 6" "c4 999983 512 f32 12" "c5chunk 4194304 1024 f32 6" "c4f64 999983 256 f64 12"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$1" -o "$1" -- python "$R/tools/run_config.py" $2 $3 $4 $5 > "$R/gpurun_out/prof_$1.log" 2>&1
  echo "$1 rc=$?"; f=$(find "$R/gpurun_out/prof_$1" -name "*kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-170
done
