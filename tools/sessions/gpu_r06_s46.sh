#!/bin/bash
# Round 6, session 46: the one-launch chirp-z on M = R1 x R2 x R3 (1296 ... 9261 points, a workgroup per transform, three register stages each
# way): the GPU test of every kernel, then the A/B against the power-of-two one-launch kernels (M = 2048 ... 16384).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_chirpz_on_a_smooth_m or smooth_work_array_only" 2>&1 | tail -15 > gpurun_out/r06_s46_pytest_chirpz_reg.log
tail -3 gpurun_out/r06_s46_pytest_chirpz_reg.log
CHIRPZ_MENU=3 timeout 1800 python tools/gpu_r06_chirpz_reg.py 2>&1 | grep '^{' > gpurun_out/r06_s46_chirpz_reg3_ab.jsonl
wc -l gpurun_out/r06_s46_chirpz_reg3_ab.jsonl
