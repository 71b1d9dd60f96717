#!/bin/bash
# Round 6, session 53: the three-stage lengths of kernels_regfft.h (494 candidates, an --ab-build of regfft_shapes.h) with whole exchanges against
# split-plane exchanges (real parts, then imaginary parts, through half the LDS: two workgroups per compute unit from 5000 points on) against
# the route each length had -- alternating on shared buffers, all three arms on the experiments library.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s53.XXXXXX)
export REGFFT_VARIANTS=1
timeout 1500 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s53_regfft.err | grep '^{' > gpurun_out/r06_s53_regfft_split_ab.jsonl
wc -l gpurun_out/r06_s53_regfft_split_ab.jsonl; tail -3 gpurun_out/r06_s53_regfft.err
