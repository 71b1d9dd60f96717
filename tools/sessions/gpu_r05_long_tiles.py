import sys, os, json, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import ctypes, numpy as np, torch
from fourier_amd import _lib, fft as F, build as B
exp = _lib.bind(ctypes.CDLL(B.OUT_EXPERIMENTS), strict=False)
prod = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for n, real in ((1000000, "f32"), (1000000, "f64"), (640000, "f32"), (810000, "f32"), (500000, "f32"), (390625, "f32")):
    esz = 8 if real == "f32" else 16
    batch = max(1, (2 << 30) // (n * esz))
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    plans = {}
    plans["product"] = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    _lib._lib = exp
    try:
        p = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
        t0 = time.time(); p.set_option("specialise", 1); tc = time.time() - t0
        plans["long_tiles"] = p
    except Exception as e:
        print("long tiles failed", n, real, repr(e)); tc = None
    finally:
        _lib._lib = prod
    outs = {}
    for name, p in plans.items():
        for _ in range(2): p.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        torch.cuda.synchronize(); outs[name] = y.clone()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); p.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        prof = p.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        print(json.dumps(dict(n=n, real=real, arm=name, plan=p.describe(), batch=batch, ms=round(t * 1e3, 3), frac8=round(batch * 2 * n * esz / t / 8e12, 4),
                              kernels_ms={k: round(ms, 3) for k, ms, c in prof if c}, compile_s=tc if name == "long_tiles" else None)), flush=True)
    if len(outs) == 2:
        a, b = outs["product"].to(torch.complex128), outs["long_tiles"].to(torch.complex128)
        print(json.dumps(dict(n=n, real=real, rel_l2_between_arms=float((a - b).norm() / a.norm()))), flush=True)
    del x, y, plans, outs; torch.cuda.empty_cache()
