#!/bin/bash
# Round 6, session 16: the Bluestein middle kernel (fft_conv_kernel) with plain instead of streaming loads / stores / both -- C4, its f64 twin, N = 65537
# (M = 2^18, conv at L = 512) and N = 40001 (M = 2^17); two processes (allocations).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
for i in 1 2; do
timeout 900 python tools/gpu_ab_options.py 999983:512 999983:256:f64 65537:4096 40001:8192 --libs conv_ld_plain=$V/libfourier_conv_ld_plain.so conv_st_plain=$V/libfourier_conv_st_plain.so conv_ldst_plain=$V/libfourier_conv_ldst_plain.so --reps 9 2>&1 | grep '^{' | tee -a gpurun_out/r06_s16_conv_policy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['ms_min'], d['equals_first_arm'], d['kernels_ms'])
"
done
