#!/bin/bash
# Round 6, session 27: Bluestein on a smooth M = L1 x L2 (three sweeps on register tiles, kernels_regtile.h) against the power-of-two M.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python tools/gpu_r06_smooth_m.py 2>&1 | grep '^{' | tee gpurun_out/r06_s27_smooth_m_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    if 'arm' in d: print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e %.1e' % (d['rel_l2_vs_torch_f64'], d['round_trip_rel_l2']), d['plan'][:48], d['kernels_ms'])
    else: print(d)
"
