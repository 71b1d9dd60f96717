#!/bin/bash
# Round 6, session 5: the one-launch chirp-z kernels of M <= 1024 -- loads in batches of 8 rows, half the transforms per workgroup (finer
# workgroups, the same waves per CU) -- against the product, alternating on shared buffers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=37,61,97,127,191,222,251,331,439,509 timeout 1200 python tools/gpu_r06_chirpz_ab.py blu_rb8 blu_cg2 blu_cg2_rb8 2>&1 | grep '^{' | tee gpurun_out/r06_s5_chirpz_rows_tuning_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
