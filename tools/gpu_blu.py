#!/usr/bin/env python3
"""Development tool: times the large-Bluestein plans (conv kernel on / off) on the product library and on every
A/B build under fourier_amd/lib/variants/ (tools/build_variants.py)."""
import ctypes, glob, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

CASES = [("C4 f32 999983x512", 999983, 512, "f32"), ("C4 f64 999983x256", 999983, 256, "f64"),
         ("prime 65537x8192", 65537, 8192, "f32"), ("40000x8192 (512x256)", 40000, 8192, "f32"),
         ("2200000x128 (3-pass)", 2200000, 128, "f32"), ("10007x16384 f64 (256x128)", 10007, 16384, "f64"),
         ("2^20 x2048", 1 << 20, 2048, "f32"), ("2^20 f64 x1024", 1 << 20, 1024, "f64")]


def run(lib, tag, n, batch, real, opts):
    cdt = torch.complex64 if real == "f32" else torch.complex128
    esz = 8 if real == "f32" else 16
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1)
    y = torch.empty_like(x)
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    for k, v in opts:
        plan.set_option(k, v)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    print(json.dumps(dict(lib=lib, tag=tag, plan=plan.describe(), opts=dict(opts), ms=round(t * 1e3, 3),
                          gflops=round(batch * 5 * n * math.log2(n) / t / 1e9, 1), frac8=round(batch * 2 * n * esz / t / 8e12, 4),
                          model_tbps=round(plan.model_bytes() * batch / t / 1e12, 3),
                          kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
    del x, y, plan; torch.cuda.empty_cache()


if __name__ == "__main__":
    libs = [("product", None)] + [(os.path.basename(p)[len("libfourier_"):-3], p)
                                  for p in sorted(glob.glob(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_*.so")))]
    base = _lib.lib()
    for name, path in libs:
        _lib._lib = base if path is None else _lib.bind(ctypes.CDLL(path))
        for tag, n, batch, real in CASES:
            for opts in ((), (("xcd_swizzle", 3),), (("bluestein_conv", 0),)):
                if path is not None and opts:
                    continue
                try:
                    run(name, tag, n, batch, real, opts)
                except Exception as e:
                    print(json.dumps(dict(lib=name, tag=tag, error=repr(e))), flush=True)
    _lib._lib = base
