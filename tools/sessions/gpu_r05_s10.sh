#!/bin/bash
# Round 5, session 10: 2^20 f32 as 2048 x 512 (row strides 4 KiB / 16 KiB) against 1024 x 1024 (8 KiB / 8 KiB); tile walks on both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['plan'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
echo "== C2"; FOURIER_PLAN_2048x512=1 timeout 600 python tools/gpu_ab_options.py 2^20:4096 2^20:4096:f64 --arms default= plain=tile_walk:0 b8=tile_walk:8 b4=tile_walk:4 b16=tile_walk:16 --libs exp=fourier_amd/lib/libfourier_experiments.so --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s10_c2_2048x512_ab.jsonl | summ
