#!/usr/bin/env python3
"""Development tool: mixed-radix lengths on the product library and on the A/B builds under fourier_amd/lib/variants/."""
import ctypes, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from fourier_amd import fft as F, _lib
from gpu_sweep import time_plan
libs = [("product", None)] + [(os.path.basename(p)[len("libfourier_"):-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_*.so")))]
base = _lib.lib()
for real, esz, cdt in (("f64", 16, torch.complex128),):
    for n in (27, 81, 243, 729, 1152, 2187, 2304, 4374, 6561, 144, 576, 9216):
        bb = (1 << 30) // (n * esz)
        xs = torch.empty((bb, n), dtype=cdt, device="cuda"); torch.view_as_real(xs).uniform_(0, 1); ys = torch.empty_like(xs)
        row = {}
        for name, path in libs:
            _lib._lib = base if path is None else _lib.bind(ctypes.CDLL(path))
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
            med, best = time_plan(plan, xs, ys, bb, reps=4, warm=1)
            row[name] = round(bb * 2.0 * n * esz / med / 8e12, 3)
            del plan
        print(real, n, row, flush=True)
        del xs, ys; torch.cuda.empty_cache()
_lib._lib = base
