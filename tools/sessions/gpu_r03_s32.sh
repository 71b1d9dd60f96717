#!/bin/bash
# Round 3, session 32 (sessions 25, 27 and 30 again at the final commit): evidence at HEAD after the prime-radix / thread-count / 160 KiB work on the LDS mixed-radix kernels: full GPU
# parity suite, smoke, default bench line, rocprofv3 kernel trace of the same command, the reference's benchmark sizes,
# the small-size table, the size sweep.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench.json
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -2
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; wc -l gpurun_out/reference_sizes.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== stress"; STRESS_SEED=990099 timeout 900 python tools/gpu_r03_stress.py > gpurun_out/stress_990099.json 2> gpurun_out/stress.err; python -c "import json; d=json.load(open(\"gpurun_out/stress_990099.json\")); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")})"
echo "== lengths with a factor 7: per-length kernel against the runtime kernel and Bluestein"; PRIME_RADIX_SIZES=840:f32,1260:f32,2520:f32,5040:f32,10080:f32,1792:f32,7168:f32,17920:f32,2520:f64,5040:f64 timeout 400 python tools/gpu_r03_prime_radix.py > gpurun_out/factor7_ab.jsonl 2>> gpurun_out/stress.err; wc -l gpurun_out/factor7_ab.jsonl
