#!/bin/bash
# Round 6, session 41: Bluestein on a smooth M with end passes of 513 ... 1024 points (N = 131073 ... 262144: M up to 1024 x 512), against the
# power-of-two M = 2^19; default rule and forced.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/gpu_r06_smooth_m.py 131101 140009 150011 160001 65537 2>&1 | grep '^{' > gpurun_out/r06_s41_smooth_m_long_end_passes_ab.jsonl
SMOOTH_FORCE=1 timeout 1200 python tools/gpu_r06_smooth_m.py 170003 180001 200003 230003 2>&1 | grep '^{' > gpurun_out/r06_s41_smooth_m_long_end_passes_forced_ab.jsonl
wc -l gpurun_out/r06_s41_*.jsonl
