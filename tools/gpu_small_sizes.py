#!/usr/bin/env python3
"""Development tool: device-resident batched throughput for small / odd sizes (f32, f64)."""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from fourier_amd import fft as F
from gpu_sweep import time_plan
dev = torch.device("cuda", 0)
for real, esz, cdt in (("f32", 8, torch.complex64), ("f64", 16, torch.complex128)):
    for n in (4, 8, 16, 32, 64, 128, 256, 512, 12, 96, 243, 729, 768, 1536, 2187, 3072, 4374, 4608, 6561, 9216, 13122, 18432, 100, 125, 625, 1000, 3125, 5000, 10000, 15625, 343, 2401, 1001, 4095, 5005, 191, 1013, 4097, 6144, 10007):
        bb = max(1, min((2 << 30) // (n * esz), 1 << 22))
        xs = torch.empty((bb, n), dtype=cdt, device=dev); torch.view_as_real(xs).uniform_(0, 1); ys = torch.empty_like(xs)
        plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
        med, best = time_plan(plan, xs, ys, bb, reps=4, warm=1)
        print(json.dumps(dict(real=real, n=n, plan=plan.describe(), batch=bb, ms=round(med * 1e3, 3), us_per_transform=round(med / bb * 1e6, 4),
                              frac8=round(bb * 2.0 * n * esz / med / 8e12, 4))), flush=True)
        del xs, ys, plan; torch.cuda.empty_cache()
