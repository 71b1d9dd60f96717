#!/bin/bash
# Full GPU session: parity tests, bench, rocprofv3 kernel trace + PMC passes at the bench's own batch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-s}
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err | grep -v amdgpu.ids
echo "== bench f64"; timeout 900 python bench.py --dtype f64 --no-cpu > gpurun_out/bench_f64.json 2>> gpurun_out/bench.err; cat gpurun_out/bench_f64.json
R="$PWD"
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
echo "== rocprof pmc"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o fetch -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$R/gpurun_out/prof_fetch.log" 2>&1; echo "fetch rc=$?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o write -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$R/gpurun_out/prof_write.log" 2>&1; echo "write rc=$?")
head -4 gpurun_out/prof_trace/trace_kernel_stats.csv | cut -c1-200
