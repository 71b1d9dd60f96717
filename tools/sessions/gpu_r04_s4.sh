#!/bin/bash
# Round 4, session 4: the prefetching last pass with a barrier between the vmcnt wait and the reads of the LDS-DMA rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
timeout 900 python tools/gpu_ab_options.py 2^22:1024 2^20:4096 999983:512 --arms plain=last_pass_prefetch:0 prefetch=last_pass_prefetch:1 \
  --libs vm0=$V/libfourier_pf_vm0.so vm0_plainst=$V/libfourier_pf_vm0_plainst.so stores_first=$V/libfourier_pf_stores_first.so \
  --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prefetch_variants_ab2.jsonl
