#!/usr/bin/env python3
"""Round 4 A/B: 2^a*3^b with a >= 12 that have a two-factor tile factorisation -- power-of-two tiles + odd passes (default)
against two mixed-length tile passes (experiments library, FOURIER_TILED_FIRST=1)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib, build

exp = _lib.bind(ctypes.CDLL(build.OUT_EXPERIMENTS))
base = _lib.lib()
for n, real in ((24576, "f32"), (36864, "f32"), (49152, "f32"), (73728, "f32"), (98304, "f32"), (110592, "f32"), (147456, "f32"), (196608, "f32"), (221184, "f32"),
                (12288, "f64"), (24576, "f64"), (49152, "f64"), (147456, "f64")):
    esz = 8 if real == "f32" else 16
    batch = max(2, (1 << 31) // (n * esz))
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for route in ("default", "tiled_first"):
        os.environ.pop("FOURIER_TILED_FIRST", None)
        _lib._lib = base
        if route == "tiled_first":
            os.environ["FOURIER_TILED_FIRST"] = "1"; _lib._lib = exp
        plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
        for _ in range(2):
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        ref = torch.fft.fft(x[:2].to(torch.complex128)); got = y[:2].to(torch.complex128)
        print(json.dumps(dict(n=n, real=real, batch=batch, route=route, plan=plan.describe(), ms=round(t * 1e3, 3),
                              frac8=round(batch * 2 * n * esz / t / 8e12, 4), rel_l2_vs_torch=float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref)))), flush=True)
        del plan
    os.environ.pop("FOURIER_TILED_FIRST", None)
    _lib._lib = base
    del x, y; torch.cuda.empty_cache()
