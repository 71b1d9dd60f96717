#!/bin/bash
# Round 6, session 19: session 18 again with the hint on the LDS mixed-radix kernels' copies only (the mixed-length tile passes lost up to 40 % there),
# twice in one session, and the tiny register kernels (2 ... 32 points) with the same hint.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
CHIRPZ_SIZES=96,100,243,729,2187,625,3125,1000,1001,768,1536,3072,6561,9216,18432,10000,15625,13122,19683,5005,4096 timeout 1200 python tools/gpu_r06_chirpz_ab.py mix_lds_nt 2>&1 | grep '^{' | tee gpurun_out/r06_s19_mixed_lds_copy_policy_ab_$rep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
done
CHIRPZ_SIZES=2,3,4,5,7,8,11,13,16,17,25,32 timeout 600 python tools/gpu_r06_chirpz_ab.py tiny_nt 2>&1 | grep '^{' | tee gpurun_out/r06_s19_tiny_policy_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.1e' % d['rel_l2_vs_torch_f64'], d['plan'][:50])
"
