#!/bin/bash
# One GPU session: parity tests, bench, sweep, rocprofv3 traces.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" ; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12
nproc
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
echo "== sweep"; timeout 1500 python tools/gpu_sweep.py > gpurun_out/sweep.log 2>&1; echo "sweep rc=$?"; tail -60 gpurun_out/sweep.log
echo "== rocprof kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof_trace" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu > "$OLDPWD/gpurun_out/prof_trace.log" 2>&1; echo "trace rc=$?")
echo "== rocprof pmc"
(cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/prof_fetch" -o fetch -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu --batch 1024 > "$OLDPWD/gpurun_out/prof_fetch.log" 2>&1; echo "fetch rc=$?")
(cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/prof_write" -o write -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu --batch 1024 > "$OLDPWD/gpurun_out/prof_write.log" 2>&1; echo "write rc=$?")
find gpurun_out -name "*.csv" | head -20
du -sh gpurun_out
