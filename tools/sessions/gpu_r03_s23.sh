#!/bin/bash
# Round 3, session 23: state after the prime-radix (5 / 7 / 11 / 13) LDS kernels and the 1024-thread mixed-radix workgroups:
# full GPU parity suite, smoke, default bench line, the reference's benchmark sizes, the small-size table, one stress seed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench.json
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; wc -l gpurun_out/reference_sizes.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== stress"; STRESS_SEED=31337 timeout 900 python tools/gpu_r03_stress.py > gpurun_out/stress_31337.json 2> gpurun_out/stress.err; tail -c 600 gpurun_out/stress_31337.json
