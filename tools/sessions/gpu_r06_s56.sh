#!/bin/bash
# Round 6, session 56: SQ wave-cycle breakdown of the register-stage transforms (kernels_regfft.h): three stages at 8000 (f64) / 8008 (f32) / 2000
# points beside two stages at 700 points -- what bounds the long three-stage kernels at 0.45 - 0.55 (tools/gpu_r04_sq.sh: one --pmc pass per group).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export SQ_CONFIGS="r8000f64 8000 4194 f64 2;r8008f32 8008 8380 f32 2;r2000f64 2000 16777 f64 2;r700f64 700 47934 f64 2;r9009f64 9009 3724 f64 2"
bash tools/gpu_r04_sq.sh 2>&1 | tail -12
cp gpurun_out/sq_breakdown.json gpurun_out/r06_s56_sq_regfft.json
