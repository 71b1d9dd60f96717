#!/bin/bash
# Copies what tools/gpu_r06_final.sh left in gpurun_out/ (scratch) into profiles/ (tracked) under <tag>_*, and folds the PMC
# summaries into profiles/traffic_latest.json.  usage: bash tools/collect_session.sh r05_s9
set -u
T=${1:?tag}; G=gpurun_out; P=profiles
cd "$(dirname "$0")/.."
cp $G/pytest_gpu.log $P/${T}_pytest_gpu.log; cp $G/bench.json $P/${T}_bench.json; cp $G/bench_details.json $P/${T}_bench_details.json
cp $G/bench_f64.json $P/${T}_bench_f64.json; cp $G/bench_c5_dist.json $P/${T}_bench_c5_full_job_1rank_rccl.json
cp "$(find $G/prof_trace -name '*kernel_stats.csv' | head -1)" $P/${T}_bench_kernel_stats.csv
for c in c3 c4 c5chunk c4f64; do cp "$(find $G/prof_$c -name '*kernel_stats.csv' | head -1)" $P/${T}_${c}_kernel_stats.csv; done
cp $G/c4c5.jsonl $P/${T}_c4c5.jsonl; cp $G/small_sizes.jsonl $P/${T}_small_sizes.jsonl; cp $G/sizes.jsonl $P/${T}_sizes.jsonl
cp $G/stress.json $P/${T}_stress_vs_oracle.json
python tools/pmc_to_traffic.py $T
