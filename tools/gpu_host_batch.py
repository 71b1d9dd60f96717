#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-batched entry point (fourier_hip_transform_batch_host_*) next to a loop over the
legacy one-transform-per-call ABI and the device-resident batched call."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fourier_amd import fft as F

for n, batch, real in ((1 << 20, 256, "f32"), (4096, 1 << 16, "f32"), (256, 1 << 20, "f32"), (999983, 128, "f32"), (1 << 20, 128, "f64")):
    dt = np.complex64 if real == "f32" else np.complex128
    esz = np.dtype(dt).itemsize
    rng = np.random.default_rng(1)
    x = rng.random((batch, n * 2), dtype=np.float32 if real == "f32" else np.float64).view(dt)
    y = np.empty_like(x)
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    plan.transform_batch_host(x[: max(1, batch // 8)], y[: max(1, batch // 8)], 0)  # warm-up: staging, scratch
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); plan.transform_batch_host(x, y, 0); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    k = min(batch, 64)
    t0 = time.perf_counter()
    for b in range(k):
        plan.transform(x[b], y[b], 0)
    t_loop = (time.perf_counter() - t0) / k
    d = torch.from_numpy(x[: min(batch, 1 << 14)]).cuda(); o = torch.empty_like(d)
    plan.transform(d, o, F.Transform.Fft); torch.cuda.synchronize()
    t0 = time.perf_counter(); plan.transform(d, o, F.Transform.Fft); torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / d.shape[0]
    flops = 5.0 * n * math.log2(n)
    print(json.dumps(dict(n=n, batch=batch, real=real, plan=plan.describe(), host_batched_ms=round(t * 1e3, 2),
                          host_batched_us_per_transform=round(t / batch * 1e6, 3),
                          pcie_gbps_each_way=round(batch * n * esz / t / 1e9, 2), host_batched_gflops=round(batch * flops / t / 1e9, 1),
                          legacy_loop_us_per_transform=round(t_loop * 1e6, 2), legacy_loop_gflops=round(flops / t_loop / 1e9, 1),
                          device_resident_us_per_transform=round(t_dev * 1e6, 4))), flush=True)
    del d, o
