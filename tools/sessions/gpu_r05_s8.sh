#!/bin/bash
# Round 5, session 8: mixed-length tile passes whose first in-tile pass reads global memory and whose last one writes it
# (FOURIER_TILED_GIO) against gather -> LDS passes -> read-twiddle-store (arm gio_off); GPU tests of the tile routes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile or tiled or factors_5_and_7 or sweep or mixed" 2>&1 | tail -6
echo "== A/B"; timeout 900 python tools/gpu_ab_options.py 62208:4315 20736:12945 59049:4546 13122:20456 118098:2273 27648:9709 24576:10922 147456:1820 100000:2684 44100:6087 1000000:268 18432:7281:f64 10368:12945:f64 13122:10228:f64 12288:10922:f64 24576:5461:f64 100000:1342:f64 \
  --libs gio_off=fourier_amd/lib/variants/libfourier_tiled_gio_off.so --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s8_tiled_gio_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['plan'], d['ms'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"
