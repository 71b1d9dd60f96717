#!/bin/bash
# Round 4, session 5: the mixed-length column-tile passes (kernels_tiled.h) -- GPU parity, then A/B against the global-pass and
# Bluestein routes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beyond_the_lds_limit or prefetching_last or product_library or non_finite" 2>&1 | tail -5
echo "== A/B"; timeout 900 python tools/gpu_r04_tiled.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tiled_ab.jsonl
