#!/bin/bash
# Round 2 evidence session: parity at HEAD, smoke, bench (default line with other_configs; f64; C5 full job through a
# 1-rank RCCL group), rocprofv3 kernel trace + the two PMC traffic passes over the default bench, kernel traces of the
# other BASELINE configurations, the reference's bench sizes, the size sweeps.  Everything lands in gpurun_out/
# (tools/collect_profiles.py <tag> copies the summaries into profiles/).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/bench.json; grep -v amdgpu.ids gpurun_out/bench.err | tail -3
echo "== bench f64"; timeout 900 python bench.py --config c3 --no-cpu > gpurun_out/bench_f64.json 2>> gpurun_out/bench.err; cut -c1-400 gpurun_out/bench_f64.json
echo "== bench c5 (full 65536-transform job) under a 1-rank RCCL group"
BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c5 --steps 3 --warmup 1 > gpurun_out/bench_c5_dist.json 2> gpurun_out/bench_c5_dist.err; echo "c5 rc=$?"; cut -c1-700 gpurun_out/bench_c5_dist.json
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
echo "== rocprof pmc"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o fetch -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-other > "$R/gpurun_out/prof_fetch.log" 2>&1; echo "fetch rc=$?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o write -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-other > "$R/gpurun_out/prof_write.log" 2>&1; echo "write rc=$?")
head -4 gpurun_out/prof_trace/trace_kernel_stats.csv | cut -c1-200
echo "== rocprof other configs"; bash tools/gpu_rocprof_configs.sh
echo "== c4c5"; python tools/gpu_c4c5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4c5.jsonl; wc -l gpurun_out/c4c5.jsonl
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; wc -l gpurun_out/reference_sizes.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== sizes sweep"; timeout 600 python tools/gpu_sweep.py --what sizes 2>&1 | grep -v amdgpu.ids | grep "size:" > gpurun_out/sizes.jsonl; wc -l gpurun_out/sizes.jsonl
