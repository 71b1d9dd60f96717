#!/bin/bash
# Round 6, session 42: f32 register tiles with a 35- / 40-point stage (875, 945, 972, 980, 1000 points): 10^6 = 1000 x 1000 as two passes against
# 100 x 100 x 100 (the f64 file: the same stages in f64 spill into AGPRs and lose to three passes -- not adopted).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_ab_options.py 1000000:512:f32 765625:512:f32 980000:512:f32 972000:512:f32 945000:512:f32 900000:512:f32 2^20:512:f32 1000000:256:f64 \
  2>&1 | grep '^{' > gpurun_out/r06_s42_1000_point_tiles.jsonl
wc -l gpurun_out/r06_s42_1000_point_tiles.jsonl
