#!/usr/bin/env python3
"""BASELINE config C1 on the CPU oracle (one core): N=4096 f32 forward, out of place, like fft_bench.rs:36."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
O.build()
n = 4096
x = (np.random.default_rng(0).random(n) + 1j * np.random.default_rng(1).random(n)).astype(np.complex64)
f = O.OracleFft(n, np.complex64)
f.transform(x, O.FFT)
t0 = time.perf_counter()
for _ in range(2000): f.transform(x, O.FFT)
print(json.dumps(dict(config="C1 N=4096 f32 CPU oracle, 1 core", us_per_transform=round((time.perf_counter() - t0) / 2000 * 1e6, 2))))
