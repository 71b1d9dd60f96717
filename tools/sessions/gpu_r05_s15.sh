#!/bin/bash
# Round 5, session 15: strided bands (a band = every (tiles / bw)-th tile of a transform instead of bw adjacent ones) against the
# adjacent-tile bands of session 3, C2 / C3 / C5 chunk.
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
ARMS="default= plain=tile_walk:0 b8=tile_walk:$(W 8 0) s8=tile_walk:$(W 8 0 2) s4=tile_walk:$(W 4 0 2) s16=tile_walk:$(W 16 0 2) s2=tile_walk:$(W 2 0 2) s8g8=tile_walk:$(W 8 8 2) s8tf=tile_walk:$(W 8 0 3) s32=tile_walk:$(W 32 0 2)"
timeout 900 python tools/gpu_ab_options.py 2^20:4096 2^20:4096:f64 2^22:1024 --arms $ARMS --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s15_strided_bands_ab.jsonl | summ
