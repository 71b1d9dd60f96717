#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/gpu_full.sh
echo "== c4c5"; python tools/gpu_c4c5.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c4c5.jsonl
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; tail -3 gpurun_out/reference_sizes.jsonl
echo "== sizes sweep"; timeout 600 python tools/gpu_sweep.py --what sizes 2>&1 | grep -v amdgpu.ids | tail -22
