#!/bin/bash
# Round 6, session 2: (a) the slow allocations chunk by chunk (gpu_r06_placement2.py); (b) the one-launch kernels with packed f32 arithmetic
# against the scalar build; (c) the stream pipeline's large-chunk arms once more, with more repetitions; (d) the GPU parity suite on the packed build.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== chirp-z packed vs scalar"; timeout 900 python tools/gpu_r06_chirpz_ab.py onelaunch_scalar 2>&1 | grep '^{' | tee gpurun_out/r06_s2_chirpz_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
for i in 1 2 3; do
  echo "== placement2 c3 process $i"; timeout 900 python tools/gpu_r06_placement2.py c3 p$i 2>&1 | grep '^{' >> gpurun_out/r06_s2_placement2_f64.jsonl
done
echo "== placement2 c2"; timeout 900 python tools/gpu_r06_placement2.py c2 p1 2>&1 | grep '^{' >> gpurun_out/r06_s2_placement2_c2.jsonl
python - <<'PY'
import json
for f in ("f64", "c2"):
    for l in open(f"gpurun_out/r06_s2_placement2_{f}.jsonl"):
        d = json.loads(l)
        if d["tag"] != "alloc": continue
        print(f, d["proc"], d["scenario"], d["y_ptr"], "whole", d["whole"], "swapped", d["swapped_y_to_x"])
        print("   pass1/chunk", [c["pass1"] for c in d["per_chunk"]])
        print("   pass0/chunk", [c["pass0"] for c in d["per_chunk"]])
        print("   fill_y", [c["fill_y_ms"] for c in d["per_chunk"]], "fill_x", [c["fill_x_ms"] for c in d["per_chunk"]])
        print("   copy", [c["copy_x_to_y_ms"] for c in d["per_chunk"]])
        print("   x0->y[j] pass1", d["x0_to_ychunk"], "pass0", d["x0_to_ychunk_pass0"])
        print("   x[j]->y0 pass0", d["xchunk_to_y0_pass0"])
        print("   arms slow", d["slow_chunk"], d["arms_on_slow_chunk"], "fast", d["fast_chunk"], d["arms_on_fast_chunk"])
PY
P() { echo "$1=stream_pipeline:$(( $2 | ($3 << 16) | (${4:-0} << 24) ))"; }
echo "== C2 stream pipeline, large chunks"; timeout 900 python tools/gpu_ab_options.py 2^20:4096 --arms default= $(P c256s2 256 2) $(P c512s2 512 2) $(P c1024s2 1024 2) $(P c2048s2 2048 2) $(P c512s3 512 3) $(P c512s4 512 4) $(P c256s4 256 4) $(P c512s2_one 512 2 1) $(P c1024s2_one 1024 2 1) default2= --reps 9 2>&1 | grep '^{' | tee gpurun_out/r06_s2_stream_pipeline_large_chunks_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['arm'], d['ms'], d['ms_min'], d['frac8'], d['equals_first_arm'])
"
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06_s2_pytest_gpu.log
