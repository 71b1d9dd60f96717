#!/usr/bin/env python3
"""Round 4 A/B: 2^a*3^b with a < 12 beyond the LDS limit -- column tiles of mixed length (default) against one global-memory pass
per radix (experiments library, FOURIER_NO_TILED_MIXED=1) and Bluestein (FOURIER_NO_GENERIC_MIXED=1 as well)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib, build

exp = _lib.bind(ctypes.CDLL(build.OUT_EXPERIMENTS))
base = _lib.lib()
for n, real in ((59049, "f32"), (62208, "f32"), (39366, "f32"), (20736, "f32"), (147456, "f32"), (2 * 3 ** 13, "f32"), (2048 * 3 ** 7, "f32"),
                (13122, "f64"), (18432, "f64"), (10368, "f64"), (62208, "f64")):
    esz = 8 if real == "f32" else 16
    batch = max(2, (1 << 31) // (n * esz))
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for route in ("tiles", "global-pass", "bluestein"):
        for k in ("FOURIER_NO_TILED_MIXED", "FOURIER_NO_GENERIC_MIXED"):
            os.environ.pop(k, None)
        _lib._lib = base
        if route != "tiles":
            os.environ["FOURIER_NO_TILED_MIXED"] = "1"; _lib._lib = exp
        if route == "bluestein":
            os.environ["FOURIER_NO_GENERIC_MIXED"] = "1"
        plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
        for _ in range(2):
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        ref = torch.fft.fft(x[:2].to(torch.complex128)); got = y[:2].to(torch.complex128)
        print(json.dumps(dict(n=n, real=real, batch=batch, route=route, plan=plan.describe(), ms=round(t * 1e3, 3),
                              frac8=round(batch * 2 * n * esz / t / 8e12, 4), rel_l2_vs_torch=float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref)),
                              kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
        del plan
    _lib._lib = base
    del x, y; torch.cuda.empty_cache()
