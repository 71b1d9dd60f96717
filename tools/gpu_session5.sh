#!/bin/bash
# Round-end evidence session: parity tests, smoke, bench (f32 + f64), rocprofv3 kernel trace + PMC traffic, C4/C5 configs,
# reference bench sizes, small sizes, size sweep.  Outputs under gpurun_out/ (copied into profiles/ by tools/collect_profiles.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash tools/gpu_full.sh
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== c4c5"; python tools/gpu_c4c5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4c5.jsonl; wc -l gpurun_out/c4c5.jsonl
echo "== bluestein conv"; python tools/gpu_blu.py 2>&1 | grep -v amdgpu.ids > gpurun_out/blu.jsonl; wc -l gpurun_out/blu.jsonl
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; wc -l gpurun_out/reference_sizes.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== sizes sweep"; timeout 600 python tools/gpu_sweep.py --what sizes 2>&1 | grep -v amdgpu.ids | grep "size:" > gpurun_out/sizes.jsonl; wc -l gpurun_out/sizes.jsonl
