#!/bin/bash
# Round 6, session 18: streaming hints on the global <-> LDS copies of the LDS mixed-radix kernels and of the mixed-length tile passes (they are plain).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=96,243,729,2187,625,3125,1000,768,1536,3072,6561,9216,18432,10000,44100,48000,100000,15625,13122,1000000 timeout 1500 python tools/gpu_r06_chirpz_ab.py mix_nt_ld mix_nt_st mix_nt_both 2>&1 | grep '^{' | tee gpurun_out/r06_s18_mixed_copy_policy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
