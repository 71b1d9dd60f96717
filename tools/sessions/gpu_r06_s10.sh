#!/bin/bash
# Round 6, session 10: robustness at the final kernels -- two more stress seeds over every plan family, the `specialise` stress, the C / C++
# consumers and the CMake package with the warm-cache target.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for seed in 71771 82882; do
  STRESS_SEED=$seed timeout 1200 python tools/gpu_r03_stress.py > gpurun_out/r06_s10_stress_vs_oracle_seed$seed.json 2> gpurun_out/r06_s10_stress_$seed.err
  python -c "import json; d=json.loads(open('gpurun_out/r06_s10_stress_vs_oracle_seed$seed.json').read().strip().splitlines()[-1]); print($seed, {k: d[k] for k in ('cases', 'failures', 'seconds')})"
done
STRESS_SEED=7474 timeout 1500 python tools/gpu_r04_stress_specialise.py > gpurun_out/r06_s10_stress_specialise_seed7474.json 2> gpurun_out/r06_s10_stress_specialise.err
python -c "import json; d=json.loads(open('gpurun_out/r06_s10_stress_specialise_seed7474.json').read().strip().splitlines()[-1]); print('specialise', {k: d[k] for k in ('cases', 'failures') if k in d})"
echo "== warm cache through the CMake package"
export FOURIER_HIP_CACHE_DIR=/tmp/fourier-warm-$$
rm -rf build/cmake_s10 && cmake -S packaging -B build/cmake_s10 -DCMAKE_HIP_COMPILER=/opt/rocm/lib/llvm/bin/clang++ -DCMAKE_PREFIX_PATH=/opt/rocm > gpurun_out/r06_s10_cmake.log 2>&1 && cmake --build build/cmake_s10 -j 32 >> gpurun_out/r06_s10_cmake.log 2>&1
echo "cmake build rc=$?"
(cd build/cmake_s10 && ctest 2>&1 | tail -8)
( time build/cmake_s10/fourier_warm_cache 1200 ) 2>&1 | tail -4
( time build/cmake_s10/fourier_warm_cache 1200 ) 2>&1 | tail -4
build/cmake_s10/fourier_warm_cache 5005 9009 1001
ls $FOURIER_HIP_CACHE_DIR | wc -l
