#!/usr/bin/env python3
"""Development tool: times a few sizes on the product library and on every A/B build under fourier_amd/lib/variants/."""
import ctypes, glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

CASES = [("2^20", 1 << 20, 2048, "f32"), ("2^20 f64", 1 << 20, 1024, "f64"), ("2^21", 1 << 21, 1024, "f32"), ("2^22", 1 << 22, 512, "f32"),
         ("2^21 f64", 1 << 21, 512, "f64"), ("C4", 999983, 512, "f32"), ("2^18", 1 << 18, 8192, "f32"), ("2^24", 1 << 24, 128, "f32"),
         ("2^6", 64, 1 << 22, "f32"), ("2^8", 256, 1 << 21, "f32"), ("2^10", 1024, 1 << 19, "f32"), ("2^12", 4096, 1 << 17, "f32"),
         ("2^14", 1 << 14, 1 << 15, "f32"), ("2^15", 1 << 15, 1 << 14, "f32"), ("1000", 1000, 1 << 18, "f32"),
         ("2^11", 2048, 1 << 18, "f32"), ("2^13", 8192, 1 << 16, "f32"), ("125", 125, 1 << 21, "f32"), ("3125", 3125, 1 << 16, "f32"),
         ("2^11 f64", 2048, 1 << 17, "f64"), ("1000 f64", 1000, 1 << 17, "f64"), ("2^8 f64", 256, 1 << 20, "f64"), ("2^6 f64", 64, 1 << 21, "f64"), ("2^5 f64", 32, 1 << 22, "f64"), ("2^7", 128, 1 << 21, "f32"),
         ("2^7 f64", 128, 1 << 20, "f64"), ("2^10 f64", 1024, 1 << 18, "f64"),
         ("2^14 big", 1 << 14, 1 << 16, "f32"), ("2^15 big", 1 << 15, 1 << 15, "f32"), ("2^12 big", 4096, 1 << 18, "f32"), ("2^14 f64", 1 << 14, 1 << 15, "f64"), ("2^13 f64", 1 << 13, 1 << 16, "f64")]


def run(lib, tag, n, batch, real, x, y):
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    print(json.dumps(dict(lib=lib, tag=tag, plan=plan.describe(), ms=round(sorted(ts)[2] * 1e3, 3),
                          kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
    del plan


if __name__ == "__main__":
    only = sys.argv[1:]
    libs = [("product", None)] + [(os.path.basename(p)[len("libfourier_"):-3], p)
                                  for p in sorted(glob.glob(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_*.so")))]
    base = _lib.lib()
    for tag, n, batch, real in CASES:
        if only and tag not in only:
            continue
        # the SAME buffers for every library (physical placement alone moves a pass by several percent), A B A B order
        cdt = torch.complex64 if real == "f32" else torch.complex128
        x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
        for rep in range(2):
            for name, path in libs:
                _lib._lib = base if path is None else _lib.bind(ctypes.CDLL(path), strict=False)
                try:
                    run(name, tag, n, batch, real, x, y)
                except Exception as e:
                    print(json.dumps(dict(lib=name, tag=tag, error=repr(e))), flush=True)
        del x, y; torch.cuda.empty_cache()
    _lib._lib = base
