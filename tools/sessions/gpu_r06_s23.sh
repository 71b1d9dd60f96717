#!/bin/bash
# Round 6, session 23: register-resident tile passes, second version (stage twiddle on the stage-B side, element-wise copy-out in the first
# pass, XCD-contiguous tile order) against: the same without the tile order (mix_noremap), the LDS kernels with it (no_regtile) and without
# (old_noremap = rounds 4 - 5).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,62208,10368,13122,15625,18432,30000,32000,44100,48000,50000,88200,96000,100000,192000,250000,1000000 timeout 2000 python tools/gpu_r06_chirpz_ab.py mix_noremap no_regtile old_noremap 2>&1 | grep '^{' | tee gpurun_out/r06_s23_regtile_remap_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
