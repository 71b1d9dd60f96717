#!/usr/bin/env python3
"""the reference's own benchmark surface (fourier-bench/benches/fft_bench.rs:153-159:
pow2 / pow3 / pow5 / composite / prime sizes 125..3125, forward + inverse, f32 + f64, one out-of-place
`transform` per iteration, fft_bench.rs:36) re-expressed for this repo: the CPU restatement (oracle, 1 core)
next to the GPU engine (a) through the drop-in host-slice ABI (latency, PCIe-bound) and (b) device-resident
batched.  Informative only: these sizes are latency-bound and are not a BASELINE config."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import fourier_amd
from fourier_amd import Transform
from oracle import oracle as O

SCENARIOS = {"pow2": [256, 512, 1024], "pow3": [243, 729, 2187], "pow5": [125, 625, 3125],
             "composite": [222, 722, 1418], "prime": [191, 439, 1013]}


def main():
    O.build()
    rng = np.random.default_rng(0)
    for real, dt, cdt in (("f32", np.complex64, torch.complex64), ("f64", np.complex128, torch.complex128)):
        for name, sizes in SCENARIOS.items():
            for n in sizes:
                for tr, code in ((Transform.Fft, O.FFT), (Transform.Ifft, O.IFFT)):
                    x = (rng.random(n) + 1j * rng.random(n)).astype(dt)  # fft_bench.rs:18-23
                    y = np.empty_like(x)
                    cpu = O.OracleFft(n, dt)
                    reps = 2000
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        cpu.transform(x, code)
                    cpu_us = (time.perf_counter() - t0) / reps * 1e6
                    plan = (fourier_amd.create_fft_f32 if real == "f32" else fourier_amd.create_fft_f64)(n)
                    plan.transform(x, y, tr)
                    t0 = time.perf_counter()
                    for _ in range(200):
                        plan.transform(x, y, tr)
                    host_us = (time.perf_counter() - t0) / 200 * 1e6
                    batch = 16384
                    dx = torch.from_numpy(np.tile(x, (batch, 1))).cuda()
                    dy = torch.empty_like(dx)
                    for _ in range(3):
                        plan.transform(dx, dy, tr)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        plan.transform(dx, dy, tr)
                    torch.cuda.synchronize()
                    dev_us = (time.perf_counter() - t0) / 20 / batch * 1e6
                    err = float(np.abs(dy[0].cpu().numpy() - cpu.transform(x, code)).max())
                    print(json.dumps(dict(scenario=name, n=n, real=real, transform=tr.name, plan=plan.describe(),
                                          cpu_oracle_us=round(cpu_us, 2), gpu_host_abi_us=round(host_us, 1),
                                          gpu_device_batched_us=round(dev_us, 4), max_abs_diff_vs_oracle=err)), flush=True)


if __name__ == "__main__":
    main()
