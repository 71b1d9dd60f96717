#!/bin/bash
# Round 6, session 12: the 16-column last pass of length 2048 (C5, one workgroup per CU) with plain instead of streaming loads / loads + stores,
# on fresh allocations (session 11: both plain = level on most allocations, -8 ... -10 % on some); C5 x 3 processes, then 2^21, then C4 and C2 as controls.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
for k in c5 c5 c5; do
  timeout 900 python tools/gpu_r06_placement4.py $k l2048_ld_plain=$V/libfourier_l2048_ld_plain.so l2048_ldst_plain=$V/libfourier_l2048_ldst_plain.so 2>&1 | grep '^{' | tee -a gpurun_out/r06_s12_l2048_last_pass_policy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kind'], d['pid'], d['scenario'], d['passes'])
"
done
timeout 900 python tools/gpu_ab_options.py 2^21:1024 2^22:1024 999983:512 2^21:512:f64 2^22:512:f64 --libs l2048_ld_plain=$V/libfourier_l2048_ld_plain.so l2048_ldst_plain=$V/libfourier_l2048_ldst_plain.so --reps 7 2>&1 | grep '^{' | tee gpurun_out/r06_s12_l2048_policy_sizes_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
"
