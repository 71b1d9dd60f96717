#!/bin/bash
# Round 6, session 52: the whole evidence set at the register-stage transforms of kernels_regfft.h (tools/gpu_r06_final.sh).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
STRESS_SEED=52525 bash tools/gpu_r06_final.sh 2>&1 | tee gpurun_out/r06_s52_session.log | tail -60
