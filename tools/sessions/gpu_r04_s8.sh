#!/bin/bash
# Round 4, session 8: run-time specialisation with hipRTC (plan option "specialise") -- GPU test and A/B against the default
# route; 2^a*3^b (a >= 12) as two mixed-length tile passes against the default; the copy ceiling with occupancy caps.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "specialise or product_library or cmake_package" 2>&1 | tail -5
echo "== specialise A/B"; timeout 900 python tools/gpu_ab_options.py 1001:262144 2002:131072 3003:131072 4095:65536 5005:53000 6006:43000 7007:37000 9009:29000 17017:15000 1001:131072:f64 3003:43000:f64 4095:32768:f64 5005:26000:f64 \
  --arms default= specialise=specialise:1 --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/specialise_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['plan'], d['ms'], d['frac8'])
    else: print(l.rstrip())
"
echo "== tiled first"; timeout 600 python tools/gpu_r04_tiled_first.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tiled_first_ab.jsonl | cut -c1-200
echo "== bench (ceiling)"; timeout 600 python bench.py --no-cpu --no-other --steps 10 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_quick.json"))
r = d["roofline"]
print(d["ms_per_step"], {k: r.get(k) for k in ("frac", "achieved", "copy_ceiling_gbps", "frac_of_copy_ceiling", "round_trips", "whole_path_frac", "whole_path_bound_frac", "whole_path_frac_of_bound")})
print(r.get("copy_ceiling", {}).get("by_policy_gbps"))
PY
