#!/bin/bash
# Development session: NT-load default, reversed pass order on 2^21 / C4 (FOURIER_REVERSE_LENS), variants.
mkdir -p gpurun_out
python tools/gpu_sweep.py --what variants 2>&1 | grep variant | cut -c1-330
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, json, math, time
sys.path.insert(0, os.getcwd())
import torch
from fourier_amd import fft as F
def t(n, batch, real="f32", rev=False, tag=""):
    if rev: os.environ["FOURIER_REVERSE_LENS"] = "1"
    else: os.environ.pop("FOURIER_REVERSE_LENS", None)
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2): plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    print(json.dumps(dict(tag=tag, rev=rev, plan=plan.describe(), ms=round(sorted(ts)[2] * 1e3, 3), kernels={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
    del x, y, plan; torch.cuda.empty_cache()
for rev in (False, True):
    t(1 << 21, 1024, rev=rev, tag="2^21")
    t(1 << 19, 4096, rev=rev, tag="2^19")
    t(1 << 17, 16384, rev=rev, tag="2^17")
    t(1 << 21, 512, "f64", rev=rev, tag="2^21 f64")
    t(999983, 512, rev=rev, tag="C4")
    t(999983, 256, "f64", rev=rev, tag="C4 f64")
    t(40000, 8192, rev=rev, tag="40000")
    t(1 << 23, 128, rev=rev, tag="2^23")
PY
