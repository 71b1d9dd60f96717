#!/bin/bash
# Round 5, session 2: the last pass out of place (plan option "scratch": pass 0 -> scratch, pass 1 scratch -> out) against in
# place on the output; tile orders (xcd_swizzle) once more on this box; f64 the same.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
echo "== C2"; timeout 600 python tools/gpu_ab_options.py 2^20:4096 --arms default= scratch=scratch:1 swz0=xcd_swizzle:0 swz2=xcd_swizzle:2 swz3=xcd_swizzle:3 swz4=xcd_swizzle:4 scratch_chunk2g=scratch:1,chunk_bytes:2147483648 scratch_chunk256m=scratch:1,chunk_bytes:268435456 --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s2_c2_scratch_ab.jsonl | summ
echo "== C3"; timeout 600 python tools/gpu_ab_options.py 2^20:4096:f64 --arms default= scratch=scratch:1 --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s2_c3_scratch_ab.jsonl | summ
echo "== C5"; timeout 600 python tools/gpu_ab_options.py 2^22:1024 --arms default= scratch=scratch:1 --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s2_c5_scratch_ab.jsonl | summ
