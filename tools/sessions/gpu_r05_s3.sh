#!/bin/bash
# Round 5, session 3: the order in which an XCD walks its tiles (plan option "tile_walk": band width in tiles, transforms per
# group, tile- or transform-fastest inside a band) -- xcd_swizzle 4 (8-tile bands over the XCD's whole range) was 2.7 % faster than
# the default on the box of session 2.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['n'], d['real'], d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'], d['kernels_ms'])
    else: print(l.rstrip())
"; }
W() { echo $(( $1 + ($2 << 8) + (${3:-0} << 19) )); }
ARMS="default= swz4=xcd_swizzle:4"
for bw in 1 2 4 8 16 32; do for g in 0 8 64; do ARMS="$ARMS b${bw}g${g}=tile_walk:$(W $bw $g)"; done; done
ARMS="$ARMS b8g0tf=tile_walk:$(W 8 0 1) b8g64tf=tile_walk:$(W 8 64 1) b1g64tf=tile_walk:$(W 1 64 1) b4g16=tile_walk:$(W 4 16) b2g32=tile_walk:$(W 2 32) b16g4=tile_walk:$(W 16 4) b32g2=tile_walk:$(W 32 2)"
echo "== C2"; timeout 900 python tools/gpu_ab_options.py 2^20:4096 --arms $ARMS --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s3_c2_tile_walk_ab.jsonl | summ
echo "== C3"; timeout 900 python tools/gpu_ab_options.py 2^20:4096:f64 --arms default= swz4=xcd_swizzle:4 b8g8=tile_walk:$(W 8 8) b8g64=tile_walk:$(W 8 64) b16g0=tile_walk:$(W 16 0) b4g0=tile_walk:$(W 4 0) b32g0=tile_walk:$(W 32 0) b2g0=tile_walk:$(W 2 0) --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s3_c3_tile_walk_ab.jsonl | summ
echo "== C5"; timeout 900 python tools/gpu_ab_options.py 2^22:1024 --arms default= swz4=xcd_swizzle:4 b8g8=tile_walk:$(W 8 8) b16g0=tile_walk:$(W 16 0) b4g0=tile_walk:$(W 4 0) b32g0=tile_walk:$(W 32 0) b2g0=tile_walk:$(W 2 0) b1g0=tile_walk:$(W 1 0) --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s3_c5_tile_walk_ab.jsonl | summ
echo "== C4"; timeout 900 python tools/gpu_ab_options.py 999983:512 --arms default= swz4=xcd_swizzle:4 b8g8=tile_walk:$(W 8 8) b16g0=tile_walk:$(W 16 0) b4g0=tile_walk:$(W 4 0) b32g0=tile_walk:$(W 32 0) --reps 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s3_c4_tile_walk_ab.jsonl | summ
