#!/bin/bash
# Round 6, session 21: the mixed-length tile passes with the transform in registers (kernels_regtile.h, default) against the LDS kernels of
# rounds 4 - 5 (variant no_regtile), alternating on shared buffers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,55296,59049,62208,10368,13122,15625,18432,22050,30000,32000,44100,48000,50000,65610,88200,96000,100000,192000,250000,1000000,3188646 timeout 1800 python tools/gpu_r06_chirpz_ab.py no_regtile 2>&1 | grep '^{' | tee gpurun_out/r06_s21_regtile_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
