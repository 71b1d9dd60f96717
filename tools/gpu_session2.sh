#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== membench"; timeout 900 python tools/membench.py > gpurun_out/membench.log 2>&1; echo "rc=$?"; tail -120 gpurun_out/membench.log
echo "== variants"; timeout 600 python tools/gpu_sweep.py --what variants > gpurun_out/sweep2.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/sweep2.log
