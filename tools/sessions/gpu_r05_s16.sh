#!/bin/bash
# Round 5, session 16: tile order of the LAST pass alone (tile_walk_last), C5 chunk / C2 / 2^21: strided bands looked 3.5 % faster on C5's
# last pass in session 15 while ruining its first pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['batch'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
W() { echo $(( $1 + ($2 << 8) + (${3:-0} << 19) )); }
ARMS="default="
for spec in "2 0 2" "4 0 2" "8 0 2" "16 0 2" "8 8 2" "2 8 2" "4 8 2" "2 64 2" "8 0 0" "16 0 0" "4 0 0" "32 0 0" "2 0 0" "8 8 0" "16 8 0" "1 0 0" "1 8 0"; do set -- $spec; ARMS="$ARMS l_b$1g$2m$3=tile_walk_last:$(W $1 $2 $3)"; done
timeout 900 python tools/gpu_ab_options.py 2^22:1024 2^22:512:f64 --arms $ARMS --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s16_last_pass_walk_ab.jsonl | summ
