#!/usr/bin/env python3
"""Round 3 development tool: C4 (Bluestein N=999983 x 512) and neighbours under the tile-order options, with a value check."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F


def run(tag, n, batch, real="f32", opts=(), check=True):
    cdt = torch.complex64 if real == "f32" else torch.complex128
    esz = 8 if real == "f32" else 16
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1)
    y = torch.empty_like(x)
    plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    for k, v in opts:
        plan.set_option(k, v)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    err = None
    if check:  # torch's own FFT on the first and last transform (independent of this repo)
        ref = torch.fft.fft(x[[0, batch - 1]].to(torch.complex128))
        got = y[[0, batch - 1]].to(torch.complex128)
        err = float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref))
    print(json.dumps(dict(tag=tag, plan=plan.describe(), n=n, batch=batch, opts=dict(opts), ms=round(t * 1e3, 3),
                          frac8=round(batch * 2 * n * esz / t / 8e12, 4), rel_l2_vs_torch=err,
                          kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
    del x, y, plan; torch.cuda.empty_cache()


if __name__ == "__main__":
    for rep in range(2):
        for cc in (1, 0):
            run("C4", 999983, 512, opts=(("bluestein_chirp_compute", cc),))
    for cc in (1, 0):
        run("C4 f64", 999983, 256, "f64", opts=(("bluestein_chirp_compute", cc),))
        run("N=65537", 65537, 8192, opts=(("bluestein_chirp_compute", cc),))
        run("N=40000", 40000, 8192, opts=(("bluestein_chirp_compute", cc),))
        run("N=2200000", 2200000, 128, opts=(("bluestein_chirp_compute", cc),))
    run("N=1021", 1021, 1 << 18)
    run("N=3125", 3125, 1 << 16)
    run("N=10007", 10007, 1 << 15)
    run("N=191", 191, 1 << 20)
