#!/bin/bash
# Round 6, session 54: GPU parity at HEAD (the final regfft_shapes.h with the split-plane variants) and the final table measured once more --
# every listed length on the product library against the route it had (FOURIER_NO_REGFFT=1, experiments library), both precisions.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s54.XXXXXX)
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s54_regfft.err | grep '^{' > gpurun_out/r06_s54_regfft_final_table_ab.jsonl
wc -l gpurun_out/r06_s54_regfft_final_table_ab.jsonl; tail -2 gpurun_out/r06_s54_regfft.err
