#!/bin/bash
# Round 6, session 50: the 22 lengths of 1715 ... 8085 points that need a register stage of 33 ... 40 points (a first run with stages of 49 points: 0.12 ... 0.2 of the peak, spilled) (both precisions: f64 spills there,
# its alternative is Bluestein) -- the same A/B as session 49.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=/tmp/fourier_cache_s50
export REGFFT_SIZES=1715,2695,3185,3430,3993,4235,4719,5005,5145,5390,5445,5577,5915,6125,6370,6435,6591,6860,7605,7623,7986,8085
export REGFFT_SPECIALISED=5005
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s50_regfft.err | grep '^{' > gpurun_out/r06_s50_regfft_long_stages_ab.jsonl
wc -l gpurun_out/r06_s50_regfft_long_stages_ab.jsonl; tail -3 gpurun_out/r06_s50_regfft.err
