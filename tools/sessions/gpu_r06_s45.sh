#!/bin/bash
# Round 6, session 45: the one-launch chirp-z kernels in registers with pipelined table loads and -- f32 -- two transforms per lane on packed
# arithmetic: the GPU test of every kernel, then the A/B against the power-of-two kernels and seven builds (one f32 transform per lane, register
# bounds, batch sizes of the table loads, chirp kept / reloaded); SQ counters of four shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_chirpz_on_a_smooth_m" 2>&1 | tail -15 > gpurun_out/r06_s45_pytest_chirpz_reg.log
tail -3 gpurun_out/r06_s45_pytest_chirpz_reg.log
V=fourier_amd/lib/variants
timeout 1800 python tools/gpu_r06_chirpz_reg.py scalar=$V/libfourier_chirpz_scalar.so cap1=$V/libfourier_chirpz_cap1.so cap4=$V/libfourier_chirpz_cap4.so tb4=$V/libfourier_chirpz_tb4.so \
  tb16=$V/libfourier_chirpz_tb16.so reload=$V/libfourier_chirpz_reload.so keep=$V/libfourier_chirpz_keep.so 2>&1 | grep '^{' > gpurun_out/r06_s45_chirpz_reg_ab.jsonl
wc -l gpurun_out/r06_s45_chirpz_reg_ab.jsonl
export RUN_CONFIG_OPTIONS=bluestein_smooth_m:2
rm -rf gpurun_out/sq_*_*/
SQ_CONFIGS="reg191f32 191 1400000 f32 2;reg191f64 191 700000 f64 2;reg439f64 439 300000 f64 2;reg439f32 439 600000 f32 2" timeout 1500 bash tools/gpu_r04_sq.sh > gpurun_out/r06_s45_sq.log 2>&1
cp gpurun_out/sq_breakdown.json gpurun_out/r06_s45_sq_chirpz_reg.json
rm -rf gpurun_out/sq_*_*/ gpurun_out/sq_*.log gpurun_out/sq_breakdown.json
tail -4 gpurun_out/r06_s45_sq.log
