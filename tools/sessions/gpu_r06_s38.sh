#!/bin/bash
# Round 6, session 38: 256-byte row segments (tiles of 32 f32 / 16 f64 columns) in the register-tile passes up to 256 / 512 points, against the
# 128-byte default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,30000,44100,48000,100000,250000,1000000,16411,65537 timeout 1500 python tools/gpu_r06_chirpz_ab.py rt_wide256 rt_wide512 2>&1 | grep '^{' > gpurun_out/r06_s38_regtile_wide_ab.jsonl
wc -l gpurun_out/r06_s38_regtile_wide_ab.jsonl
