#!/bin/bash
# Round 6, session 40: register tiles of 513 ... 1024 points on 64-byte rows -- two tile passes where three (or power-of-two tiles + odd passes)
# were needed -- against the plans of rounds 4 - 5 (tile_max_512) and against keeping 2^a 3^b, a >= 12, beyond 384 x 384 on the power-of-two route.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
CHIRPZ_SIZES=196608,221184,331776,390625,442368,500000,589824,640000,729000,786432,884736,250000 timeout 1800 python tools/gpu_r06_chirpz_ab.py tile_max_512 tiled_first_384 2>&1 | grep '^{' > gpurun_out/r06_s40_long_tiles_ab.jsonl
wc -l gpurun_out/r06_s40_long_tiles_ab.jsonl
