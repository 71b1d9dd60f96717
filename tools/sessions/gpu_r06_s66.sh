#!/bin/bash
# Round 6, session 66: the 2^a 3^b lengths on register stages ON REQUEST (plan option "register_stages" = 1; an --optin-build of regfft_shapes.h: 57
# candidates) against their default plan (the LDS kernels on the reference's own schedule, bit-identical to the CPU restatement), both precisions.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export REGFFT_VARIANTS=3
timeout 900 python tools/gpu_r06_regfft_ab.py 2>gpurun_out/r06_s66_regfft.err | grep '^{' > gpurun_out/r06_s66_regfft_on_request_ab.jsonl
wc -l gpurun_out/r06_s66_regfft_on_request_ab.jsonl; tail -3 gpurun_out/r06_s66_regfft.err
