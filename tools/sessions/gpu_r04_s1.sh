#!/bin/bash
# Round 4, session 1: the engine after the split into translation units (same kernels, same register allocation): full GPU
# parity suite, smoke, the default bench line (now with the in-run copy ceiling), rocprofv3 kernel trace of the same command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
echo "== pytest gpu"; timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
r = d["roofline"]
print({k: r.get(k) for k in ("frac", "copy_ceiling_gbps", "frac_of_copy_ceiling", "round_trips", "whole_path_frac", "whole_path_bound_frac", "whole_path_frac_of_bound")})
print(r.get("copy_ceiling"))
for k in ("c3", "c4", "c5"):
    o = d["other_configs"][k]
    print(k, o.get("ms_per_step"), o.get("hbm_frac_algorithmic"), o.get("kernels_ms"), o.get("error"))
PY
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_trace" -o trace -- python "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-other > "$R/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -2
