#!/bin/bash
# Round 6, session 34: register budget of the register-tile kernels -- launch bounds that aim at as many workgroups per CU as the LDS holds while
# a wave keeps at least 168 / 128 / 96 registers (default: no bound) -- on plain tile plans and on the smooth-M Bluestein sweeps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,44100,48000,100000,250000,1000000,16411,18221,32771,40001,65537,80021,10007 timeout 1800 python tools/gpu_r06_chirpz_ab.py rt_minv168 rt_minv128 rt_minv96 2>&1 | grep '^{' > gpurun_out/r06_s34_regtile_budget_ab.jsonl
wc -l gpurun_out/r06_s34_regtile_budget_ab.jsonl
