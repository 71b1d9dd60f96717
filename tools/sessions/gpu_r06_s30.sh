#!/bin/bash
# Round 6, session 30: cache policy of the register-tile passes (plain by default): streaming hints on the loads, the stores, both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,30000,44100,48000,100000,250000,1000000,16411,65537 timeout 1500 python tools/gpu_r06_chirpz_ab.py rt_ld_nt rt_st_nt rt_both_nt 2>&1 | grep '^{' > gpurun_out/r06_s30_regtile_policy_ab.jsonl
wc -l gpurun_out/r06_s30_regtile_policy_ab.jsonl
