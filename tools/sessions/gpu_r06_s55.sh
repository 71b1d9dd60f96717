#!/bin/bash
# Round 6, session 55: the three-stage lengths of kernels_regfft.h (an --ab-build of regfft_shapes.h) in four variants -- whole / split-plane
# exchanges x whole / factored twiddle tables (W_N^{(j3 + R3 j2) k1} = W_{R1R2}^{j2 k1} W_N^{j3 k1}: tables that stay in the L1) -- against the
# route each length had; alternating on shared buffers, all arms on the experiments library.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s55.XXXXXX)
export REGFFT_VARIANTS=1
timeout 1500 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s55_regfft.err | grep '^{' > gpurun_out/r06_s55_regfft_variants_ab.jsonl
wc -l gpurun_out/r06_s55_regfft_variants_ab.jsonl; tail -3 gpurun_out/r06_s55_regfft.err
