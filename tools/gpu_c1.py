#!/usr/bin/env python3
"""BASELINE config C1 (N=4096 f32, one transform): the legacy host ABI call, the CPU port on one core, and the
device-resident batched rate, on the same box."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fourier_amd import fft as F
n = 4096
x = (np.random.default_rng(0).random(n) + 1j * np.random.default_rng(1).random(n)).astype(np.complex64); y = np.empty_like(x)
p = F.create_fft_f32(n, 0)
p.transform(x, y, 0)
t0 = time.perf_counter()
for _ in range(2000): p.transform(x, y, 0)
legacy_us = (time.perf_counter() - t0) / 2000 * 1e6
d = torch.from_numpy(np.tile(x, (1 << 16, 1))).cuda(); o = torch.empty_like(d)
p.transform(d, o, F.Transform.Fft); torch.cuda.synchronize()
t0 = time.perf_counter(); p.transform(d, o, F.Transform.Fft); torch.cuda.synchronize(); dev_ns = (time.perf_counter() - t0) / d.shape[0] * 1e9
print(json.dumps(dict(config="C1 N=4096 f32", legacy_host_call_us=round(legacy_us, 2), device_batched_ns_per_transform=round(dev_ns, 2),
                      rel_l2_vs_numpy=float(np.linalg.norm(y - np.fft.fft(x.astype(np.complex128))) / np.linalg.norm(np.fft.fft(x.astype(np.complex128)))))))
