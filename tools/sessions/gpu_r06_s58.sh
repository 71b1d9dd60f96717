#!/bin/bash
# Round 6, session 58: the f32 three-stage lengths of kernels_regfft.h with ONE transform per workgroup on scalar arithmetic (half the registers and
# LDS of the packed pair: two workgroups of seven ... nine waves per compute unit), without / with factored tables, against the listed variant
# and the route each length had (an --unpaired-build of regfft_shapes.h; all arms on the experiments library).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_CACHE_DIR=$(mktemp -d /tmp/fourier_cache_s58.XXXXXX)
export REGFFT_VARIANTS=2
timeout 1500 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s58_regfft.err | grep '^{' > gpurun_out/r06_s58_regfft_unpaired_ab.jsonl
wc -l gpurun_out/r06_s58_regfft_unpaired_ab.jsonl; tail -3 gpurun_out/r06_s58_regfft.err
