#!/bin/bash
# Round 6, session 13: the one-launch plans 2^11 ... 2^15 without streaming hints on their loads and stores (all of them / only the 1024-thread
# workgroups that sit alone on a CU), after session 12's finding on the one-workgroup-per-CU last pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=2048,4096,8192,16384,32768 timeout 1200 python tools/gpu_r06_chirpz_ab.py twolevel_plain_1024 twolevel_plain_all 2>&1 | grep '^{' | tee gpurun_out/r06_s13_one_launch_policy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['plan'])
"
echo "== the same on fresh buffers per arm order (second process)"
CHIRPZ_SIZES=16384,32768 timeout 1200 python tools/gpu_r06_chirpz_ab.py twolevel_plain_1024 twolevel_plain_all 2>&1 | grep '^{' | tee -a gpurun_out/r06_s13_one_launch_policy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['plan'])
"
