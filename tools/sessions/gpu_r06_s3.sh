#!/bin/bash
# Round 6, session 3: per-XCD tile rotation ("xcd_rotate") against the slow mode of the last passes; the packed one-launch kernels (fixed) again;
# SQ counters of a chirp-z kernel in both builds; the GPU parity suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for k in c3 c3 c2 c5; do
  echo "== placement3 $k"; timeout 900 python tools/gpu_r06_placement3.py $k p$RANDOM 2>&1 | grep '^{' >> gpurun_out/r06_s3_placement3_$k.jsonl
done
python - <<'PY'
import json
for f in ("c3", "c2", "c5"):
    for l in open(f"gpurun_out/r06_s3_placement3_{f}.jsonl"):
        d = json.loads(l)
        print(f, d["proc"], d["scenario"], d["y_ptr"])
        print("   whole pass1", {k: v[1] for k, v in d["whole"].items()})
        print("   whole pass0", {k: v[0] for k, v in d["whole"].items()})
        for j, c in d["chunks"].items():
            print("   chunk", j, "pass1", {k: v[1] for k, v in c.items()})
            print("   chunk", j, "pass0", {k: v[0] for k, v in c.items()})
PY
echo "== chirp-z packed vs scalar"; timeout 900 python tools/gpu_r06_chirpz_ab.py onelaunch_scalar 2>&1 | grep '^{' | tee gpurun_out/r06_s3_chirpz_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['real'], d['n'], d['arm'], d['ms'], d['frac8'], '%.2e' % d['rel_l2_vs_torch_f64'], d['plan'])
"
R="$PWD"; cd /tmp
for lib in product scalar; do
  LIBARG=""; [ $lib = scalar ] && LIBARG=$R/fourier_amd/lib/variants/libfourier_onelaunch_scalar.so
  for cs in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES" \
            "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    name=${cs%%:*}; ctrs=${cs#*:}
    for n in 191 1013; do
      timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d "$R/gpurun_out/r06_s3_sq_${lib}_${n}_$name" -o "$name" -- python "$R/tools/run_config.py" $n 600000 f32 2 $LIBARG > "$R/gpurun_out/r06_s3_sq_${lib}_${n}_$name.log" 2>&1
      echo "sq $lib $n $name rc=$?"
    done
  done
done
cd "$R"
python - <<'PY'
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob("gpurun_out/r06_s3_sq_*_*_*/")):
    cfg = d.split("/")[1][len("r06_s3_sq_"):].rsplit("_", 1)[0]
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
json.dump({"note": "per-dispatch averages; frac_* = counter / SQ_WAVE_CYCLES", "configs": out}, open("gpurun_out/r06_s3_sq_chirpz.json", "w"), indent=1)
for cfg, ks in out.items():
    for k, c in ks.items():
        print(cfg, k[:70], {n: (round(v, 4) if n.startswith("frac") or n.startswith("lds_c") else int(v)) for n, v in c.items()})
PY
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06_s3_pytest_gpu.log
