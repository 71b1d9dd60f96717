#!/bin/bash
# Round 6, session 7: the one-launch chirp-z kernels of M = 2048 ... 8192 (one transform per workgroup of 64 ... 256 threads) under a
# 128-register cap (four waves per SIMD instead of three), packed and scalar arithmetic.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=722,1013,1418,2039,3001,4093,4097,10007 timeout 1200 python tools/gpu_r06_chirpz_ab.py blu_small_mw4 blu_small_mw4_scalar onelaunch_scalar 2>&1 | grep '^{' | tee gpurun_out/r06_s7_chirpz_small_occupancy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
