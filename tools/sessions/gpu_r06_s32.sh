#!/bin/bash
# Round 6, session 32: Bluestein on a smooth M chosen for evenly split tile lengths (up to 4 % above the smallest product), default adoption
# rule (power-of-two work array at least 1.6 x longer), against the power-of-two M.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/gpu_r06_smooth_m.py 8209 9001 10007 16411 17011 18221 19001 20011 32771 34003 36007 38011 40001 65537 70001 75011 80021 2>&1 | grep '^{' > gpurun_out/r06_s32_smooth_m_balanced_ab.jsonl
wc -l gpurun_out/r06_s32_smooth_m_balanced_ab.jsonl
