#!/bin/bash
# Round 6, session 63: the whole evidence set at HEAD of the round (register-stage transforms to 20480 points in both precisions) + the final table against the routes it replaced: tools/gpu_r06_final.sh.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
STRESS_SEED=63636 bash tools/gpu_r06_final.sh 2>&1 | tee gpurun_out/r06_s63_session.log | tail -60
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s63.XXXXXX)
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s63_regfft.err | grep '^{' > gpurun_out/r06_s63_regfft_final_table_ab.jsonl
wc -l gpurun_out/r06_s63_regfft_final_table_ab.jsonl
