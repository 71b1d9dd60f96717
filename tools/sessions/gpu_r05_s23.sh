#!/bin/bash
# Round 5, session 23: one-launch chirp-z kernels with the padding half of the work array as compile-time zeros (8 of 16 register rows
# loaded, multiplied by the chirp and stored; the first radix-16 stage of the forward and the last stage of the inverse transform fold)
# against all 16 rows (arm prune_off = the kernels until now): the reference's composite / prime benchmark lengths and M up to 2^15.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sweep or bluestein or blu or reference_chirp or non_finite or random_sizes" 2>&1 | tail -3
echo "== A/B"; timeout 900 python tools/gpu_ab_options.py 191:1405421 222:1209096 439:611480 722:371797 1013:264990 1418:189306 4097:65520 10007:26824 23:4194304 97:2767011 191:702710:f64 1013:132495:f64 4097:32760:f64 \
  --libs prune_off=fourier_amd/lib/variants/libfourier_blu_prune_off.so --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s23_chirpz_pruned_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['plan'], d['ms'], d['frac8'], d['equals_first_arm'])
    else: print(l.rstrip())
"
