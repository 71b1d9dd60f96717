#!/bin/bash
# Round 4, session 2: the persistent prefetching last pass (fft_last_prefetch_kernel) -- bit-identity tests on the GPU, then
# A/B against the plain last pass on shared buffers; the in-run copy ceiling with the column-tile copy added.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "prefetching_last or non_finite" 2>&1 | tail -5
echo "== A/B"; timeout 900 python tools/gpu_ab_options.py 2^22:1024 2^21:1024 2^20:4096 999983:512 2^20:2048:f64 2^22:512:f64 --arms plain=last_pass_prefetch:0 prefetch=last_pass_prefetch:1 --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefetch_ab.jsonl
echo "== bench (ceiling)"; timeout 600 python bench.py --no-cpu --no-other --steps 10 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_quick.json"))
r = d["roofline"]
print(d["ms_per_step"], {k: r.get(k) for k in ("frac", "copy_ceiling_gbps", "frac_of_copy_ceiling", "round_trips", "whole_path_frac", "whole_path_bound_frac", "whole_path_frac_of_bound")})
print(r.get("copy_ceiling", {}).get("by_policy_gbps"))
PY
