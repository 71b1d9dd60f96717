#!/bin/bash
# Round 5, session 4: band walk (tile_walk) over the two-pass power-of-two sizes, f32 and f64 -- where does the 8-tile band of
# session 3 (f32 2^20: -1.7 ... -2.7 %) hold.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['batch'], d['arm'], d['plan'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
W() { echo $(( $1 + ($2 << 8) + (${3:-0} << 19) )); }
ARMS="default= b4=tile_walk:$(W 4 0) b8=tile_walk:$(W 8 0) b16=tile_walk:$(W 16 0) b8g16=tile_walk:$(W 8 16)"
echo "== f32"; timeout 900 python tools/gpu_ab_options.py 2^16:32768 2^17:16384 2^18:8192 2^19:4096 2^20:2048 2^21:1024 2^22:512 2^24:128 --arms $ARMS --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s4_tile_walk_sizes_f32_ab.jsonl | summ
echo "== f64"; timeout 900 python tools/gpu_ab_options.py 2^16:16384:f64 2^18:4096:f64 2^19:2048:f64 2^20:1024:f64 2^21:512:f64 2^22:256:f64 --arms $ARMS --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s4_tile_walk_sizes_f64_ab.jsonl | summ
