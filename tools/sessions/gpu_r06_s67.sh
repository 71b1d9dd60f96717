#!/bin/bash
# Round 6, session 67: GPU parity and smoke at HEAD of the round (plan option register_stages), one stress seed over lengths to 20480.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_s67_smoke.log; tail -3 gpurun_out/r06_s67_smoke.log
echo "== stress"; STRESS_SEED=67676 timeout 1200 python tools/gpu_r03_stress.py > gpurun_out/stress.json 2> gpurun_out/stress.err; python -c "import json; d=json.loads(open(\"gpurun_out/stress.json\").read().strip().splitlines()[-1]); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")})"
