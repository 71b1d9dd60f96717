#!/bin/bash
# Round 5 evidence session: parity at HEAD, smoke, bench (default line with the streaming ceiling and the flat per-config scalars;
# f64; C5 full job through a 1-rank RCCL group), rocprofv3 kernel trace over the default bench and the other BASELINE
# configurations, the PMC traffic passes (one counter set per run) over C2 / C3 / C4 / C5 chunk, the size sweeps, one stress seed.
# Everything lands in gpurun_out/ (tools/collect_session.sh <tag> copies it to profiles/<tag>_*).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py --details gpurun_out/bench_details.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench.json; grep -v amdgpu.ids gpurun_out/bench.err | grep -v reference_bench_sizes | tail -3
echo "== bench f64"; timeout 900 python bench.py --config c3 --no-cpu > gpurun_out/bench_f64.json 2>> gpurun_out/bench.err; tail -c 700 gpurun_out/bench_f64.json
echo "== bench c5 (full 65536-transform job) under a 1-rank RCCL group"
BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c5 --steps 3 --warmup 1 > gpurun_out/bench_c5_dist.json 2> gpurun_out/bench_c5_dist.err; echo "c5 rc=$?"; tail -c 600 gpurun_out/bench_c5_dist.json
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other --no-ceiling > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1); head -4 "$f" | cut -c1-200
echo "== rocprof other configs"; bash tools/gpu_rocprof_configs.sh
echo "== pmc"; bash tools/gpu_r03_pmc.sh > gpurun_out/pmc.log 2>&1; grep "rc=" gpurun_out/pmc.log | tr '\n' ' '
echo "== c4c5"; python tools/gpu_c4c5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4c5.jsonl; wc -l gpurun_out/c4c5.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== sizes sweep"; timeout 600 python tools/gpu_sweep.py --what sizes 2>&1 | grep -v amdgpu.ids | grep "size:" > gpurun_out/sizes.jsonl; wc -l gpurun_out/sizes.jsonl
echo "== stress"; STRESS_SEED=${STRESS_SEED:-50505} timeout 1200 python tools/gpu_r03_stress.py > gpurun_out/stress.json 2> gpurun_out/stress.err; python -c "import json; d=json.loads(open(\"gpurun_out/stress.json\").read().strip().splitlines()[-1]); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")})"
