#!/usr/bin/env python3
"""Runs `reps` device-resident batched forward transforms of one configuration (for rocprofv3 kernel traces):
python tools/run_config.py <n> <batch> <f32|f64> <reps> [library.so: an A/B build instead of the product library]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fourier_amd import fft as F

n, batch, real, reps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
if len(sys.argv) > 5:
    import ctypes
    from fourier_amd import _lib
    _lib._lib = _lib.bind(ctypes.CDLL(sys.argv[5]), strict=False)
cdt = torch.complex64 if real == "f32" else torch.complex128
x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
for kv in os.environ.get("RUN_CONFIG_OPTIONS", "").split(","):  # plan options, e.g. RUN_CONFIG_OPTIONS=bluestein_smooth_m:2
    if kv:
        plan.set_option(kv.split(":")[0], int(kv.split(":")[1]))
for _ in range(reps):
    plan.transform(x, y, F.Transform.Fft)
torch.cuda.synchronize()
print(plan.describe())
