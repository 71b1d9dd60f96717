#!/bin/bash
# Round 3, session 1: parity at HEAD (the buffer-descriptor kernels), A/B against the round-2 library on shared
# buffers, C4 under the tile-order options, HBM-side counters for C3 / C4 / C5 chunk.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== A/B r02 vs HEAD"; timeout 900 python tools/gpu_variants_sizes.py "2^20" "2^20 f64" "2^21" "2^22" "C4" "2^18" "2^24" "2^12" "2^14" "2^15" "2^11" "2^13" "1000" "3125" "2^10" "2^11 f64" "1000 f64" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_r02_head.jsonl; wc -l gpurun_out/ab_r02_head.jsonl
echo "== C4 options"; timeout 600 python tools/gpu_r03_c4.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_options.jsonl; wc -l gpurun_out/c4_options.jsonl
echo "== pmc"; bash tools/gpu_r03_pmc.sh > gpurun_out/pmc.log 2>&1; tail -5 gpurun_out/pmc.log
