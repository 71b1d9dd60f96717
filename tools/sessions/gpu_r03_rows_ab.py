#!/usr/bin/env python3
"""Round 3 A/B: whole-transform (ROWS) kernels of length 32 .. 128 with their global I/O staged through LDS (default)
against the element-access form (variant library rows_unstaged); 256 as a control."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

base = _lib.lib()
var = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_rows_unstaged.so")), strict=False)
for real, esz, cdt, sizes in (("f32", 8, torch.complex64, (64, 128, 256)), ("f64", 16, torch.complex128, (32, 64, 128, 256))):
    for n in sizes:
        batch = (1 << 31) // (n * esz)
        x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
        st = torch.cuda.current_stream().cuda_stream
        keep = None
        for rnd in range(2):  # two rounds: the first measurement of a size runs on cold clocks
            for route, lib in (("staged", base), ("unstaged", var)):
                _lib._lib = lib
                plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
                for _ in range(3):
                    plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
                torch.cuda.synchronize(); ts = []
                for _ in range(7):
                    t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                t = sorted(ts)[3]
                if keep is None:
                    keep = y[:4096].clone()
                ref = torch.fft.fft(x[:64].to(torch.complex128)); got = y[:64].to(torch.complex128)
                err = [float(torch.linalg.norm(got[i::2] - ref[i::2]) / torch.linalg.norm(ref[i::2])) for i in (0, 1)]  # even / odd transforms
                print(json.dumps(dict(n=n, real=real, batch=batch, round=rnd, route=route, plan=plan.describe(), ms=round(t * 1e3, 3),
                                      frac8=round(batch * 2 * n * esz / t / 8e12, 4), same_bits=bool(torch.equal(y[:4096], keep)), rel_l2_even_odd=err)), flush=True)
                del plan
        _lib._lib = base
        del x, y; torch.cuda.empty_cache()
