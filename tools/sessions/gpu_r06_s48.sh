#!/bin/bash
# Round 6, session 48: lengths with factors 5 ... 13 as a direct transform on the register stages of kernels_chirpz.h (regfft_kernel /
# regfft3_kernel, one launch) against the route they had (LDS mixed-radix kernels; FOURIER_NO_REGFFT=1 on the experiments library).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s48_regfft.err | grep '^{' > gpurun_out/r06_s48_regfft_ab.jsonl
wc -l gpurun_out/r06_s48_regfft_ab.jsonl; tail -3 gpurun_out/r06_s48_regfft.err
