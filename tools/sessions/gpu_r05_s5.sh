#!/bin/bash
# Round 5, session 5: GPU parity at the band-walk default + ragged-tile twiddle clamp + shared tile shape; the bench line with
# the streaming ceiling (copies + skeletons) and the flat per-config scalars; rocprofv3 kernel stats of the same command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r05_s5_pytest_gpu.log
echo "== bench"; timeout 900 python bench.py --details gpurun_out/r05_s5_bench_details.json > gpurun_out/r05_s5_bench.json 2> gpurun_out/r05_s5_bench.err; echo rc=$?; tail -c 1800 gpurun_out/r05_s5_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s5_bench.json"))
r = d["roofline"]
print({k: v for k, v in r.items() if not isinstance(v, (dict, list))})
print(r.get("skeleton_ceiling")); print(r.get("copy_ceiling", {}).get("by_policy_gbps"))
PY
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-other --no-ceiling --steps 10 > /tmp/prof_bench.json 2>/tmp/prof.err; cd $GRAFT_REPO_ROOT; f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r05_s5_bench_kernel_stats.csv; head -5 gpurun_out/r05_s5_bench_kernel_stats.csv; cp /tmp/prof_bench.json gpurun_out/r05_s5_bench_under_rocprof.json
