#!/bin/bash
# Round 5, session 1 (diagnostics): where do the f32 1024 x 1024 passes lose against their own
# skeleton and against the f64 passes -- ablation arms on shared buffers; does buffer placement move the passes; do wave
# priorities / a start-up phase shift of the co-resident workgroups help the conv kernel and the one-workgroup-per-CU pass.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
echo "== C2 arms"; timeout 600 python tools/gpu_ab_options.py 2^20:4096 --libs abl1=$V/libfourier_abl1.so abl2=$V/libfourier_abl2.so abl3=$V/libfourier_abl3.so abl4=$V/libfourier_abl4.so abl5=$V/libfourier_abl5.so row16=$V/libfourier_row_stores_16b.so setprio=$V/libfourier_setprio.so --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s1_c2_ablation_ab.jsonl | summ
echo "== C3 arms"; timeout 600 python tools/gpu_ab_options.py 2^20:4096:f64 --libs abl1=$V/libfourier_abl1.so abl2=$V/libfourier_abl2.so abl3=$V/libfourier_abl3.so abl4=$V/libfourier_abl4.so abl5=$V/libfourier_abl5.so setprio=$V/libfourier_setprio.so --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s1_c3_ablation_ab.jsonl | summ
echo "== C4 arms"; timeout 600 python tools/gpu_ab_options.py 999983:512 --libs setprio=$V/libfourier_setprio.so setprio2=$V/libfourier_setprio2.so dephase2=$V/libfourier_conv_dephase2.so dephase4=$V/libfourier_conv_dephase4.so dephase4_setprio=$V/libfourier_conv_dephase4_setprio.so --reps 9 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s1_c4_setprio_dephase_ab.jsonl | summ
echo "== C5 chunk arms"; timeout 600 python tools/gpu_ab_options.py 2^22:1024 --libs setprio=$V/libfourier_setprio.so abl2=$V/libfourier_abl2.so --reps 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s1_c5_setprio_ab.jsonl | summ
echo "== placement"; timeout 600 python tools/gpu_r05_placement.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s1_placement.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['tag'], d.get('dx'), d.get('dy'), d.get('filler_mb'), d['product_ms'], d['skeleton_ms'], d['copy_tiles_streaming_ms_scaled'])
    else: print(l.rstrip())
"
