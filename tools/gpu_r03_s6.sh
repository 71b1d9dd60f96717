#!/bin/bash
# Round 3, session 6: parity at HEAD, persistent last-pass workgroups A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== persistent A/B"; timeout 900 python tools/gpu_r03_persist.py 2>&1 | grep -v amdgpu.ids > gpurun_out/persist_ab.jsonl; wc -l gpurun_out/persist_ab.jsonl
