#!/bin/bash
# Round 2, session 2: parity at HEAD (new L=2048 kernels, XCD-fused plan), then the A/B of both against what they replace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
echo "== A/B"; timeout 1500 python tools/gpu_r02_ab.py all --variants 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_ab.jsonl; echo "ab rc=$?"; cut -c1-330 gpurun_out/r02_ab.jsonl
