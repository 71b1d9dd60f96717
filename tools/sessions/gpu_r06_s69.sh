#!/bin/bash
# Round 6, session 69: one stress seed with the library-wide default FOURIER_HIP_REGISTER_STAGES=1 -- every 2^a 3^b length listed on request runs
# its register-stage kernel (random codes, batches, in / out of place, both precisions, against the oracle).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
export FOURIER_HIP_REGISTER_STAGES=1
python - <<'PY'
import fourier_amd as fa
print("default:", fa.get_default_option("register_stages_at_create"), fa.create_fft_f32(4608).describe(), "|", fa.create_fft_f64(2592).describe())
PY
STRESS_SEED=69696 timeout 1200 python tools/gpu_r03_stress.py > gpurun_out/r06_s69_stress_register_stages_default.json 2> gpurun_out/stress.err
python -c "import json; d=json.loads(open(\"gpurun_out/r06_s69_stress_register_stages_default.json\").read().strip().splitlines()[-1]); print({k: d[k] for k in (\"cases\", \"failures\", \"seconds\")}); print({k: v for k, v in d.get(\"worst\", {}).items() if \"registers\" in k})"
