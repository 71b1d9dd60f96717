#!/bin/bash
# Round 6, session 35: Bluestein on a smooth M with the loads that are needed late (w at the conv kernel's outputs, the chirp at the chirp-out
# pass's outputs) issued before the barrier instead of with the data; default rule, against the power-of-two M.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/gpu_r06_smooth_m.py 8209 9001 10007 16411 17011 18221 19001 20011 32771 34003 36007 38011 40001 65537 70001 75011 80021 2>&1 | grep '^{' > gpurun_out/r06_s35_smooth_m_late_loads_ab.jsonl
SMOOTH_FORCE=1 timeout 900 python tools/gpu_r06_smooth_m.py 11003 22003 24001 44017 48017 90001 100003 2>&1 | grep '^{' > gpurun_out/r06_s35_smooth_m_late_loads_forced_ab.jsonl
wc -l gpurun_out/r06_s35_*.jsonl
