#!/bin/bash
# Round 6, session 17: FIRST pass of length 2048 -- 8-column tiles (default) vs 16-column tiles (one workgroup per CU) with / without the streaming hint on the loads.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
timeout 900 python tools/gpu_r06_wide_first.py 2>&1 | grep '^{' | tee -a gpurun_out/r06_s17_first_pass_2048_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
"
done
