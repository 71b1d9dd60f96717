#!/bin/bash
# Round 6, session 31: Bluestein on a smooth M with the padding rows of the chirp-in / chirp-out passes as compile-time zeros, forced wherever a
# product of two tile lengths exists (option value 2), against the power-of-two M: where does it pay now?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
SMOOTH_FORCE=1 timeout 1500 python tools/gpu_r06_smooth_m.py 8209 9001 10007 11003 12289 14009 16411 18221 20011 22003 24001 26003 28001 32771 36007 40001 44017 48017 52009 56003 65537 70001 80021 90001 100003 110017 120011 2>&1 | grep '^{' > gpurun_out/r06_s31_smooth_m_pruned_ab.jsonl
wc -l gpurun_out/r06_s31_smooth_m_pruned_ab.jsonl
