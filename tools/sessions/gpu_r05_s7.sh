#!/bin/bash
# Round 5, session 7: GPU tests of the ahead-of-time 5/7 tile passes, the code-object cache and the specialise policy; rates of the
# lengths that moved from Bluestein to tile passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "factors_5_and_7 or code_object_cache or specialise or cmake or product_library or sweep" 2>&1 | tail -15
echo "== rates"; timeout 900 python tools/gpu_ab_options.py 100000:2684 44100:6087 48000:5592 96000:2796 1000000:268 9800:27392 21000:12783 30870:8696 100000:1342:f64 44100:3043:f64 1000000:134:f64 --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s7_smooth_tile_lengths.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['plan'], d['ms'], d['frac8'], d['kernels_ms'])
    else: print(l.rstrip())
"
