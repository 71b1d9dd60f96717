#!/bin/bash
# Round 6, session 6: the chirp-z kernels of M <= 1024 as ONE-WAVE workgroups (every barrier inside a wave), loads in batches of 4 / 8 rows,
# against the product and the half-width variant; the plain whole-row kernels (N = 256, 512, 1024) on narrower workgroups.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=37,61,97,127,191,222,251,331,439,509 timeout 1200 python tools/gpu_r06_chirpz_ab.py blu_cg2_rb8 blu_1wave_rb8 blu_1wave_rb4 2>&1 | grep '^{' | tee gpurun_out/r06_s6_chirpz_one_wave_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
CHIRPZ_SIZES=256,512,1024 timeout 1200 python tools/gpu_r06_chirpz_ab.py rows_cg2 rows_1wave 2>&1 | grep '^{' | tee gpurun_out/r06_s6_rows_width_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
