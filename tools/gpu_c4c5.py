#!/usr/bin/env python3
"""Development tool: times BASELINE configs C4 (Bluestein N=999983 x512) and C5-chunk (N=2^22) variants."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

def timeit(plan, x, y, batch, reps=5, warm=2):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warm): plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]

def run(tag, n, batch, real="f32", opts=()):
    cdt = torch.complex64 if real == "f32" else torch.complex128
    esz = 8 if real == "f32" else 16
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1)
    y = torch.empty_like(x)
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    for k, v in opts: plan.set_option(k, v)
    t = timeit(plan, x, y, batch)
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, torch.cuda.current_stream().cuda_stream)
    print(json.dumps(dict(tag=tag, plan=plan.describe(), n=n, batch=batch, opts=dict(opts), ms=round(t * 1e3, 3),
                          gflops=round(batch * 5 * n * math.log2(n) / t / 1e9, 1), alg_gbps=round(batch * 2 * n * esz / t / 1e9, 1),
                          frac8=round(batch * 2 * n * esz / t / 8e12, 4), kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
    del x, y, plan; torch.cuda.empty_cache()

if __name__ == "__main__":
    run("C4 fused", 999983, 512)
    run("C4 unfused", 999983, 512, opts=(("bluestein_fusion", 0),))
    run("C4 fused chunk256M", 999983, 512, opts=(("chunk_bytes", 256 << 20),))
    run("C4 f64 fused", 999983, 256, "f64")
    run("prime 65537", 65537, 8192)
    run("mixed 3*2^18", 3 << 18, 1024)
    run("mixed 9*2^16", 9 << 16, 1024)
    run("mixed 27*2^14", 27 << 14, 2048)
    run("mixed 3*2^12", 3 << 12, 65536)
    run("mixed 81*2^12", 81 << 12, 2048)
    run("mixed 243*2^12", 243 << 12, 1024)
    run("mixed 729*2^13", 729 << 13, 128)
    run("C5 chunk 2^22", 1 << 22, 1024)
    run("2^21", 1 << 21, 1024)
    run("2^24 3-pass", 1 << 24, 128)
