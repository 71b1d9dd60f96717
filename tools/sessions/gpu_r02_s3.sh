#!/bin/bash
# Round 2, session 3: parity at HEAD, then A/Bs: 2^22 as 4096 x 1024 (narrow first pass), conv-kernel cache-policy
# variants + XCD-sliced tile order for C4, the XCD-fused plan with the window wait moved behind the tile's arithmetic.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== A/B product"; timeout 900 python tools/gpu_r02_ab.py all 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_ab3.jsonl; echo "ab rc=$?"
echo "== A/B conv variants"; timeout 600 python tools/gpu_r02_ab.py conv --only-variants conv_stnt conv_wnt conv_both 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02_ab3.jsonl; echo "ab rc=$?"
echo "== A/B fused mw3"; timeout 600 python tools/gpu_r02_ab.py fused --only-variants fused_mw3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02_ab3.jsonl; echo "ab rc=$?"
cut -c1-300 gpurun_out/r02_ab3.jsonl
