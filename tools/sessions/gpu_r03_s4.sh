#!/bin/bash
# Round 3, session 4: parity at HEAD (new tests included), stage-twiddle batching and chirp-out store policy A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/pytest_gpu.log
echo "== A/B"; timeout 900 python tools/gpu_variants_sizes.py "C4" "2^22" "2^20" "2^20 f64" "2^14" "2^15" "2^12" "2^18" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_s4.jsonl; wc -l gpurun_out/ab_s4.jsonl
