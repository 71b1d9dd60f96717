#!/bin/bash
# Round 6, session 47: the whole evidence set at the register chirp-z kernels (tools/gpu_r06_final.sh) + the reference's bench sizes table.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
STRESS_SEED=47474 bash tools/gpu_r06_final.sh 2>&1 | tee gpurun_out/r06_s47_session.log | tail -60
