#!/bin/bash
# Round 6, session 57: the whole evidence set at HEAD of the round (the register-stage transforms with their per-length variants): tools/gpu_r06_final.sh.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
STRESS_SEED=57575 bash tools/gpu_r06_final.sh 2>&1 | tee gpurun_out/r06_s57_session.log | tail -60
