#!/bin/bash
# Round 6, session 68: GPU parity, smoke and the default bench line at HEAD of the round (library-wide default register_stages_at_create).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_s68_smoke.log; tail -3 gpurun_out/r06_s68_smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/r06_s68_bench.json 2> gpurun_out/r06_s68_bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r06_s68_bench.json
