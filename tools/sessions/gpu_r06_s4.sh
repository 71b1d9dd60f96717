#!/bin/bash
# Round 6, session 4: the row-mode LDS swizzle (conflict-free under the bank model) against the skew layout, with and without packed arithmetic;
# SQ counters again; a per-XCD phase in the walk through its own transforms against the slow allocations (f64); the GPU parity suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rows / chirp-z: swizzle + packed (product) vs skew + packed vs swizzle + scalar"
CHIRPZ_SIZES=64,128,256,512,1024,37,97,191,222,331,439,722,1013,1418,2039,4097,10007 timeout 1200 python tools/gpu_r06_chirpz_ab.py rows_skew onelaunch_scalar 2>&1 | grep '^{' | tee gpurun_out/r06_s4_rows_swizzle_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
R="$PWD"; cd /tmp
for cs in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
          "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
  name=${cs%%:*}; ctrs=${cs#*:}
  for n in 97 191 256; do
    timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/r06_s4_sq_product_${n}_$name" -o "$name" -- python "$R/tools/run_config.py" $n 600000 f32 2 > "$R/gpurun_out/r06_s4_sq_product_${n}_$name.log" 2>&1
    echo "sq product $n $name rc=$?"
  done
done
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/r06_s4_sq_*_*_*/")):
    cfg = d.split("/")[1][len("r06_s4_sq_"):].rsplit("_", 1)[0]
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "fourier_hip" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].replace("fourier_hip::", "").replace("(fourier_hip::PassArgs)", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            out.setdefault(cfg, {}).setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
for cfg, ks in out.items():
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
                if n in c:
                    c["frac_" + n] = round(c[n] / wc, 4)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
json.dump({"note": "per-dispatch averages; frac_* = counter / SQ_WAVE_CYCLES", "configs": out}, open("gpurun_out/r06_s4_sq_rows.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        print(cfg, k[:70], {n: round(v, 4) for n, v in c.items() if n.startswith("frac") or n.startswith("lds_c")})
PY
for i in 1 2; do
  echo "== placement3 phase c3 $i"; PLACEMENT3_PHASE=1 timeout 900 python tools/gpu_r06_placement3.py c3 p$RANDOM 2>&1 | grep '^{' >> gpurun_out/r06_s4_placement3_phase_c3.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_s4_placement3_phase_c3.jsonl"):
    d = json.loads(l)
    print(d["proc"], d["scenario"], "whole p0", {k: v[0] for k, v in d["whole"].items()}, "p1", {k: v[1] for k, v in d["whole"].items()})
    for j, c in d["chunks"].items():
        print("   chunk", j, "p1", {k: v[1] for k, v in c.items()})
PY
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06_s4_pytest_gpu.log
