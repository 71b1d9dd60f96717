#!/usr/bin/env python3
"""Round 3 A/B: last passes as persistent workgroups (plan option "persistent": 0 never, 1 = L 2048 only (default), 2 = 1024 too)
on shared buffers, A B C A B C order."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

CASES = [("C5 chunk 2^22", 1 << 22, 1024, "f32"), ("2^21", 1 << 21, 1024, "f32"), ("C4", 999983, 512, "f32"), ("C2 2^20", 1 << 20, 4096, "f32"),
         ("C3 2^20 f64", 1 << 20, 4096, "f64"), ("2^22 f64", 1 << 22, 512, "f64"), ("2^18", 1 << 18, 8192, "f32"), ("2^24", 1 << 24, 128, "f32"),
         ("3*2^20", 3 << 20, 512, "f32")]
for tag, n, batch, real in CASES:
    cdt = torch.complex64 if real == "f32" else torch.complex128
    esz = 8 if real == "f32" else 16
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        for mode in (0, 1, 2):
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
            plan.set_option("persistent", mode)
            for _ in range(2):
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            torch.cuda.synchronize(); ts = []
            for _ in range(5):
                t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = sorted(ts)[2]
            prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            print(json.dumps(dict(tag=tag, persistent=mode, plan=plan.describe(), ms=round(t * 1e3, 3), frac8=round(batch * 2 * n * esz / t / 8e12, 4),
                                  kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
            del plan
    del x, y; torch.cuda.empty_cache()
