#!/bin/bash
# Round 6, session 26: register-resident tile passes, fifth version (row swizzle instead of padding in the exchange buffer: 5 instead of 4 workgroups per CU at
# L = 210 f32; chunks of 8 tiles per XCD; 125 = 25 x 5 back on the LDS kernel) against the LDS kernels (no_regtile, same tile order).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,62208,10368,13122,15625,18432,30000,32000,44100,48000,50000,88200,96000,100000,192000,250000,1000000 timeout 2400 python tools/gpu_r06_chirpz_ab.py no_regtile 2>&1 | grep '^{' | tee gpurun_out/r06_s26_regtile_v5_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
