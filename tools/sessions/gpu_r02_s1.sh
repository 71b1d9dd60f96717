#!/bin/bash
# Round 2, session 1: parity tests at HEAD, the new bench.py (default line with other_configs; C5 through a single-rank
# RCCL group), every membench tag DESIGN.md quotes (appended to ONE file) plus the XCD-local exchange model,
# rocprofv3 kernel trace + the two PMC traffic passes over the default bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/bench.json; grep -v amdgpu.ids gpurun_out/bench.err | tail -3
echo "== bench c5 under a 1-rank RCCL group"
BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c5 --steps 2 --warmup 1 > gpurun_out/bench_c5_dist.json 2> gpurun_out/bench_c5_dist.err; echo "c5 rc=$?"; cut -c1-1200 gpurun_out/bench_c5_dist.json; grep -v amdgpu.ids gpurun_out/bench_c5_dist.err | tail -3
echo "== membench (all tags, one file)"
rm -f gpurun_out/membench.jsonl
for mode in "" --xcd --fused-only --sync --pipe --l2x; do
  timeout 600 python tools/membench.py $mode > gpurun_out/membench_${mode#--}.log 2>&1; echo "membench '$mode' rc=$?"
done
wc -l gpurun_out/membench.jsonl; grep l2x gpurun_out/membench.jsonl | cut -c1-260
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
echo "== rocprof pmc"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/prof_fetch" -o fetch -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-other > "$R/gpurun_out/prof_fetch.log" 2>&1; echo "fetch rc=$?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/prof_write" -o write -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-other > "$R/gpurun_out/prof_write.log" 2>&1; echo "write rc=$?")
head -4 gpurun_out/prof_trace/trace_kernel_stats.csv | cut -c1-200
