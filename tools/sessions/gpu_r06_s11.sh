#!/bin/bash
# Round 6, session 11: the last pass's final stores (and loads) without the streaming hint, on fresh allocations (is the slow mode a property of
# streaming stores?), C3 x 3 processes, C2, C5.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
for k in c3 c3 c3 c2 c2 c5; do
  timeout 900 python tools/gpu_r06_placement4.py $k last_st_plain=$V/libfourier_last_st_plain.so last_ldst_plain=$V/libfourier_last_ldst_plain.so 2>&1 | grep '^{' | tee -a gpurun_out/r06_s11_last_pass_store_policy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kind'], d['pid'], d['scenario'], d['passes'])
"
done
