#!/bin/bash
# Round 6, session 51: the 61 candidate lengths of 8193 ... 10240 points (three register stages; before: per-length LDS kernels, two tile passes or
# Bluestein) -- the same A/B as session 49.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=/tmp/fourier_cache_s51
export REGFFT_SIZES=8232,8250,8316,8320,8400,8424,8448,8450,8470,8505,8580,8624,8640,8712,8736,8750,8775,8788,8800,8820,8910,8960,9000,9009,9072,9075,9100,9126,9152,9240,9261,9360,9375,9408,9438,9450,9464,9477,9504,9555,9600,9625,9680,9702,9720,9750,9800,9801,9828,9856,9900,9984,10000,10010,10080,10125,10140,10164,10192,10206,10240
export REGFFT_SPECIALISED=9009
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s51_regfft.err | grep '^{' > gpurun_out/r06_s51_regfft_8193_10240_ab.jsonl
wc -l gpurun_out/r06_s51_regfft_8193_10240_ab.jsonl; tail -3 gpurun_out/r06_s51_regfft.err
