#!/bin/bash
# Round 3, session 2: the store-soffset hazard (repro + fix), parity at HEAD, plain-pass A/B (buffer vs pointer loads,
# with / without the laundered FIRST mapping) against round 2 on shared buffers, C4 under the band-major tile order.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== hazard"; timeout 600 python tools/gpu_r03_hazard.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/store_soffset_hazard.txt
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
echo "== A/B"; timeout 900 python tools/gpu_variants_sizes.py "2^20" "2^20 f64" "2^21" "2^22" "2^24" "2^18" "2^12" "2^14" "2^15" "2^13" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_s2.jsonl; wc -l gpurun_out/ab_s2.jsonl
echo "== C4 options"; timeout 600 python tools/gpu_r03_c4.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_options_s2.jsonl; wc -l gpurun_out/c4_options_s2.jsonl
