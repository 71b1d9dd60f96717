#!/bin/bash
# Round 6, session 61: GPU parity and smoke at HEAD (regfft_shapes.h with the f32 lengths to 20480 points), the final table once more against
# the route each length had (product library against the experiments library under FOURIER_NO_REGFFT=1), one more stress seed.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s61.XXXXXX)
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s61_regfft.err | grep '^{' > gpurun_out/r06_s61_regfft_final_table_ab.jsonl
wc -l gpurun_out/r06_s61_regfft_final_table_ab.jsonl; tail -2 gpurun_out/r06_s61_regfft.err
echo "== stress"; STRESS_SEED=61616 timeout 1200 python tools/gpu_r03_stress.py > gpurun_out/stress.json 2> gpurun_out/stress.err; python -c "import json; d=json.loads(open(\"gpurun_out/stress.json\").read().strip().splitlines()[-1]); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")})"
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench.json
