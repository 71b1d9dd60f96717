#!/bin/bash
# Round 4, session 3: variants of the persistent prefetching last pass against the plain last pass, shared buffers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
timeout 900 python tools/gpu_ab_options.py 2^22:1024 2^20:4096 --arms plain=last_pass_prefetch:0 prefetch=last_pass_prefetch:1 \
  --libs vm0=$V/libfourier_pf_vm0.so vm0_plainst=$V/libfourier_pf_vm0_plainst.so plainst=$V/libfourier_pf_plainst.so stores_first=$V/libfourier_pf_stores_first.so vm0_dma4=$V/libfourier_pf_vm0_dma4.so \
  --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefetch_variants_ab.jsonl
