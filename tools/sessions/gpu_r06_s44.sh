#!/bin/bash
# Round 6, session 44: the one-launch chirp-z kernels in registers with their table loads in batches between scheduling fences (no spills under the
# register bounds) against the power-of-two kernels and five builds (no register bound, batch sizes, chirp kept / reloaded); SQ counters of three shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
timeout 1500 python tools/gpu_r06_chirpz_reg.py free=$V/libfourier_chirpz_free.so tb16=$V/libfourier_chirpz_tb16.so tb4_8=$V/libfourier_chirpz_tb4_8.so reload=$V/libfourier_chirpz_reload.so keep=$V/libfourier_chirpz_keep.so \
  2>&1 | grep '^{' > gpurun_out/r06_s44_chirpz_reg_ab.jsonl
wc -l gpurun_out/r06_s44_chirpz_reg_ab.jsonl
export RUN_CONFIG_OPTIONS=bluestein_smooth_m:2
SQ_CONFIGS="reg191f32 191 1400000 f32 2;reg191f64 191 700000 f64 2;reg439f64 439 300000 f64 2;reg439f32 439 600000 f32 2" timeout 1500 bash tools/gpu_r04_sq.sh > gpurun_out/r06_s44_sq.log 2>&1
cp gpurun_out/sq_breakdown.json gpurun_out/r06_s44_sq_chirpz_reg.json
export RUN_CONFIG_OPTIONS=bluestein_smooth_m:0
rm -rf gpurun_out/sq_*_*/
SQ_CONFIGS="pow191f32 191 1400000 f32 2;pow191f64 191 700000 f64 2;pow439f64 439 300000 f64 2" timeout 1500 bash tools/gpu_r04_sq.sh >> gpurun_out/r06_s44_sq.log 2>&1
cp gpurun_out/sq_breakdown.json gpurun_out/r06_s44_sq_chirpz_pow2.json
rm -rf gpurun_out/sq_*_*/ gpurun_out/sq_*.log
tail -12 gpurun_out/r06_s44_sq.log
