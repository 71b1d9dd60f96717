#!/bin/bash
# Round 6, session 43: the whole chirp-z of a short transform in one launch on a smooth M = R1 x R2 in registers (kernels_chirpz.h): the GPU test of
# every kernel of the menu, then the A/B against the power-of-two one-launch kernels and four builds of the new kernels (register caps, chirp kept
# or reloaded).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_chirpz_on_a_smooth_m" 2>&1 | tail -15 > gpurun_out/r06_s43_pytest_chirpz_reg.log
V=fourier_amd/lib/variants
timeout 1500 python tools/gpu_r06_chirpz_reg.py free=$V/libfourier_chirpz_free.so cap3=$V/libfourier_chirpz_cap3.so reload=$V/libfourier_chirpz_reload.so keep=$V/libfourier_chirpz_keep.so \
  2>&1 | grep '^{' > gpurun_out/r06_s43_chirpz_reg_ab.jsonl
wc -l gpurun_out/r06_s43_chirpz_reg_ab.jsonl
tail -5 gpurun_out/r06_s43_pytest_chirpz_reg.log
