#!/usr/bin/env python3
"""Round 3 A/B: Bluestein inner plan order for C4 -- 2048 x 1024 forward (default: end passes of length 2048, conv kernel at 1024)
against 1024 x 2048 (experiments library, FOURIER_BLU_SHORT_FIRST=1: end passes at 1024, conv at 2048)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib, build

exp = _lib.bind(ctypes.CDLL(build.OUT_EXPERIMENTS)); base = _lib.lib()
for real, n, batch in (("f32", 999983, 512), ("f64", 999983, 256), ("f32", 65537, 8192), ("f32", 40000, 8192)):
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        for route in ("default", "short_first"):
            if route != "default":
                os.environ["FOURIER_BLU_SHORT_FIRST"] = "1"; _lib._lib = exp
            else:
                os.environ.pop("FOURIER_BLU_SHORT_FIRST", None); _lib._lib = base
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
            for _ in range(2): plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            torch.cuda.synchronize(); ts = []
            for _ in range(5):
                t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            ref = torch.fft.fft(x[:1].to(torch.complex128)); got = y[:1].to(torch.complex128)
            print(json.dumps(dict(real=real, n=n, route=route, plan=plan.describe(), ms=round(sorted(ts)[2] * 1e3, 3), kernels={k: round(m, 3) for k, m, c in prof if c},
                                  rel_l2=float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref)))), flush=True)
            del plan
    _lib._lib = base
    del x, y; torch.cuda.empty_cache()
