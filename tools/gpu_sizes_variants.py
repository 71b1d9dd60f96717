#!/usr/bin/env python3
"""Development tool: size sweep (f32) for every built variant library."""
import ctypes, glob, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import _lib, fft as F
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_sweep import time_plan
dev = torch.device("cuda", 0)
for real, esz, cdt in (("f32", 8, torch.complex64), ("f64", 16, torch.complex128)):
    for lg in (8, 10, 11, 12, 13, 14, 15, 16, 18, 20):
        nn = 1 << lg
        bb = max(1, min((4 << 30) // (nn * esz), 1 << 20))
        xs = torch.empty((bb, nn), dtype=cdt, device=dev); torch.view_as_real(xs).uniform_(0, 1); ys = torch.empty_like(xs)
        row = {"size": f"{real}:2^{lg}"}
        for path in sorted(glob.glob(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_*.so"))):
            name = os.path.basename(path)[len("libfourier_"):-3]
            _lib._lib = _lib.bind(ctypes.CDLL(path))
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(nn, 0)
            med, best = time_plan(plan, xs, ys, bb, reps=4, warm=1)
            row[name] = round(bb * 2.0 * nn * esz / med / 8e12, 4)
            del plan
        print(json.dumps(row), flush=True)
        del xs, ys; torch.cuda.empty_cache()
