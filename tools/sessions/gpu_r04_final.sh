#!/bin/bash
# Round 4 evidence session: parity at HEAD, smoke, bench (default line with the in-run copy ceiling and other_configs; f64; C5
# full job through a 1-rank RCCL group), rocprofv3 kernel trace over the default bench and the other BASELINE
# configurations, the PMC traffic passes (one counter set per run) over C2 / C3 / C4 / C5 chunk, the size sweeps, the A/B of
# the LDS-staged twiddle tables of the per-length mixed-radix kernels, of the run-time specialisation and of the mixed-length tile passes.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/bench.json; grep -v amdgpu.ids gpurun_out/bench.err | tail -3
echo "== bench f64"; timeout 900 python bench.py --config c3 --no-cpu > gpurun_out/bench_f64.json 2>> gpurun_out/bench.err; cut -c1-300 gpurun_out/bench_f64.json
echo "== bench c5 (full 65536-transform job) under a 1-rank RCCL group"
BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c5 --steps 3 --warmup 1 > gpurun_out/bench_c5_dist.json 2> gpurun_out/bench_c5_dist.err; echo "c5 rc=$?"; cut -c1-500 gpurun_out/bench_c5_dist.json
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
head -4 gpurun_out/prof_trace/trace_kernel_stats.csv | cut -c1-200
echo "== rocprof other configs"; bash tools/gpu_rocprof_configs.sh
echo "== pmc"; bash tools/gpu_r03_pmc.sh > gpurun_out/pmc.log 2>&1; grep "rc=" gpurun_out/pmc.log | tr '\n' ' '
echo "== c4c5"; python tools/gpu_c4c5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/c4c5.jsonl; wc -l gpurun_out/c4c5.jsonl
echo "== reference sizes"; timeout 900 python tests/harness/bench_reference_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/reference_sizes.jsonl; wc -l gpurun_out/reference_sizes.jsonl
echo "== small sizes"; python tools/gpu_small_sizes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/small_sizes.jsonl; wc -l gpurun_out/small_sizes.jsonl
echo "== sizes sweep"; timeout 600 python tools/gpu_sweep.py --what sizes 2>&1 | grep -v amdgpu.ids | grep "size:" > gpurun_out/sizes.jsonl; wc -l gpurun_out/sizes.jsonl
echo "== LDS twiddle tables of the per-length mixed-radix kernels, A/B"
timeout 600 python tools/gpu_ab_options.py 6:8388608 12:4194304 24:4194304 48:2097152 96:1048576 192:524288 384:262144 243:524288 486:262144 100:1048576 125:1048576 250:524288 500:262144 320:262144 448:262144 96:524288:f64 243:262144:f64 500:131072:f64 \
  --libs twlds=fourier_amd/lib/variants/libfourier_mix_twlds.so --reps 7 2>&1 | grep -v amdgpu.ids > gpurun_out/mix_twlds_ab.jsonl; wc -l gpurun_out/mix_twlds_ab.jsonl
echo "== specialise (hipRTC) A/B"
timeout 600 python tools/gpu_ab_options.py 1001:262144 3003:131072 4095:65536 5005:53000 9009:29000 18018:14000 100000:2600 44100:6000 1001:131072:f64 9009:14000:f64 \
  --arms default= specialise=specialise:1 --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/specialise_ab.jsonl; wc -l gpurun_out/specialise_ab.jsonl
echo "== mixed tiles A/B"; timeout 600 python tools/gpu_r04_tiled.py 2>&1 | grep -v amdgpu.ids > gpurun_out/tiled_ab.jsonl; wc -l gpurun_out/tiled_ab.jsonl
echo "== stress"; STRESS_SEED=40404 timeout 900 python tools/gpu_r03_stress.py > gpurun_out/stress_40404.json 2> gpurun_out/stress.err; python -c "import json; d=json.load(open(\"gpurun_out/stress_40404.json\")); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")})"
