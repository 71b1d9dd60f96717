#!/bin/bash
# Round 6, session 49: every candidate length of regfft_shapes.h (592 lengths of 14 ... 8192 points with factors 5 ... 13, A/B build: all adopted)
# as a direct transform on register stages (kernels_regfft.h) against the route it had (experiments library, FOURIER_NO_REGFFT=1); a third arm
# for 20 lengths: the length's own LDS kernel compiled at run time.  The table decides regfft_shapes.h (tools/gen_regfft_shapes.py --ab).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=/tmp/fourier_cache_s49
export REGFFT_SPECIALISED=77,143,175,350,385,700,715,1001,1400,2002,2450,3003,4004,4900,5005,6006,1155,2310,3465,8008
timeout 1500 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s49_regfft.err | grep '^{' > gpurun_out/r06_s49_regfft_ab.jsonl
wc -l gpurun_out/r06_s49_regfft_ab.jsonl; tail -3 gpurun_out/r06_s49_regfft.err
