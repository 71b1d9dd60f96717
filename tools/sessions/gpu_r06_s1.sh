#!/bin/bash
# Round 6, session 1: (a) which variable separates the two modes of the f64 1024 x 1024 / f32 2048 x 2048 last
# passes -- fresh allocations in several processes, tile orders, virtual offsets; (b) the two passes of C2 software-pipelined over two
# streams with the intermediate in a small ring (plan option "stream_pipeline"), alternating arms on shared buffers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
V=fourier_amd/lib/variants
echo "== pytest (pipeline option)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stream_pipeline or product_library" 2>&1 | tail -3
for i in 1 2 3; do
  echo "== placement c3 process $i"; timeout 600 python tools/gpu_r06_placement.py c3 p$i 2>&1 | grep '^{' >> gpurun_out/r06_s1_placement_f64.jsonl
done
for i in 1 2; do
  echo "== placement c5 process $i"; timeout 600 python tools/gpu_r06_placement.py c5 p$i 2>&1 | grep '^{' >> gpurun_out/r06_s1_placement_c5.jsonl
done
echo "== placement c2"; PLACEMENT_FULL=0 timeout 600 python tools/gpu_r06_placement.py c2 p1 2>&1 | grep '^{' >> gpurun_out/r06_s1_placement_c2.jsonl
python - <<'PY'
import json
for f in ("f64", "c5", "c2"):
    print("==", f)
    for l in open(f"gpurun_out/r06_s1_placement_{f}.jsonl"):
        d = json.loads(l)
        if d["tag"] == "fresh_alloc":
            print(d["proc"], d["scenario"], d["y_ptr"], {k: v.get("pass1") for k, v in d["arms"].items()}, "pass0", d["arms"]["default"].get("pass0"))
        elif d["tag"] in ("out_offset", "in_offset"):
            print(d["proc"], d["tag"], d["dx"], d["dy"], d["default"], d.get("walk8"))
        else:
            print(d)
PY
P() { echo "$1=stream_pipeline:$(( $2 | ($3 << 16) | (${4:-0} << 24) ))"; }
ARMS="default= $(P c1s2 1 2) $(P c1s4 1 4) $(P c1s8 1 8) $(P c2s2 2 2) $(P c2s4 2 4) $(P c2s8 2 8) $(P c4s2 4 2) $(P c4s3 4 3) $(P c4s4 4 4) $(P c8s2 8 2) $(P c8s3 8 3) $(P c8s4 8 4) $(P c8s8 8 8) $(P c16s2 16 2) $(P c16s4 16 4) $(P c32s2 32 2) $(P c32s4 32 4) $(P c64s2 64 2) $(P c128s2 128 2) $(P c256s2 256 2) $(P c512s2 512 2) $(P c8s2_one 8 2 1) $(P c64s2_one 64 2 1) $(P c512s2_one 512 2 1)"
echo "== C2 stream pipeline"; timeout 900 python tools/gpu_ab_options.py 2^20:4096 --arms $ARMS --reps 5 2>&1 | grep '^{' | tee gpurun_out/r06_s1_stream_pipeline_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'])
"
ARMS2="default= $(P c8s2 8 2) $(P c8s4 8 4) $(P c16s2 16 2) $(P c32s2 32 2) $(P c64s2 64 2) $(P c4s4 4 4)"
echo "== C2 stream pipeline x cache policies"; timeout 900 python tools/gpu_ab_options.py 2^20:4096 --arms $ARMS2 --libs st_mid_plain=$V/libfourier_nt_store.so ld_last_plain=$V/libfourier_nt_load.so both_plain=$V/libfourier_nt_both.so ld_last_sc1=$V/libfourier_ld_last_sc1.so st_plain_ld_sc1=$V/libfourier_st_mid_plain_ld_last_sc1.so --reps 5 2>&1 | grep '^{' | tee gpurun_out/r06_s1_stream_pipeline_policies_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'])
"
echo "== C3 stream pipeline"; timeout 900 python tools/gpu_ab_options.py 2^20:4096:f64 --arms default= $(P c4s2 4 2) $(P c8s2 8 2) $(P c16s2 16 2) $(P c64s2 64 2) $(P c256s2 256 2) --reps 5 2>&1 | grep '^{' | tee gpurun_out/r06_s1_stream_pipeline_c3_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'])
"
