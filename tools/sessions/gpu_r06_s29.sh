#!/bin/bash
# Round 6, session 29: register-resident tile passes on 16-byte units (f32: two columns per thread) -- the plain passes against the LDS
# kernels (no_regtile), and Bluestein on a smooth M against the power-of-two M.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=20736,59049,62208,10368,13122,15625,18432,30000,32000,44100,48000,50000,88200,96000,100000,192000,250000,1000000 timeout 1500 python tools/gpu_r06_chirpz_ab.py no_regtile 2>&1 | grep '^{' > gpurun_out/r06_s29_regtile_units_ab.jsonl
timeout 1500 python tools/gpu_r06_smooth_m.py 2>&1 | grep '^{' > gpurun_out/r06_s29_smooth_m_units_ab.jsonl
wc -l gpurun_out/r06_s29_*.jsonl
