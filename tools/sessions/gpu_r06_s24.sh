#!/bin/bash
# Round 6, session 25 (= session 24 with fewer arms): register-resident tile passes, fourth version (no scratch in the f64 copy-out, branch-free loads), and the workgroup -> tile order:
# chunks of 0 / 2 / 4 (default) / 8 / 32 neighbouring tiles per XCD where row segments straddle 128-byte lines, 4 everywhere (chunk4_always);
# the LDS kernels of rounds 4 - 5 with (no_regtile) and without (old_chunk0) the order.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,62208,10368,13122,15625,18432,30000,32000,44100,48000,50000,88200,96000,100000,192000,250000,1000000 timeout 2400 python tools/gpu_r06_chirpz_ab.py chunk0 no_regtile old_chunk0 2>&1 | grep '^{' | tee gpurun_out/r06_s25_regtile_v4_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
