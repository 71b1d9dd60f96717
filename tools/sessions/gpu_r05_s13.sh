#!/bin/bash
# Round 5, session 13: the inter-pass twiddle of the two-pass plans applied by the LAST pass on its loads (variant tw_on_load) against by
# the FIRST pass before its stores (product): the same products on the same values -- bit-identical --, moved out of the pass that is
# furthest above its skeleton.
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
timeout 900 python tools/gpu_ab_options.py 2^20:4096 2^20:2048:f64 2^22:1024 2^18:8192 2^16:32768 2^21:1024 --libs tw_on_load=fourier_amd/lib/variants/libfourier_tw_on_load.so --reps 9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s13_twiddle_on_load_ab.jsonl | summ
