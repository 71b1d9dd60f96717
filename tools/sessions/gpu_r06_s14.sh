#!/bin/bash
# Round 6, session 14: the C5 chunk under the new cache policy of its last pass (no streaming hints) -- tile orders on fresh allocations again
# (two processes): is there an order that brings the allocations that stayed at 13.5 ms down as well?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2; do
  PLACEMENT_REPS=3 timeout 900 python tools/gpu_r06_placement.py c5 q$i 2>&1 | grep '^{' >> gpurun_out/r06_s14_placement_c5_plain_policy.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_s14_placement_c5_plain_policy.jsonl"):
    d = json.loads(l)
    if d["tag"] == "fresh_alloc":
        print(d["proc"], d["scenario"], "pass0", d["arms"]["default"].get("pass0"), {k: v.get("pass1") for k, v in d["arms"].items()})
PY
