#!/usr/bin/env python3
"""Development tool: times build variants / plan options of the headline workload on one GPU and
writes JSON lines to gpurun_out/sweep.jsonl.  Not part of the product or of the driver contract."""
import argparse
import ctypes
import glob
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from fourier_amd import _lib, fft as F  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, "sweep.jsonl"), "a")


def emit(**kw):
    line = json.dumps(kw)
    print(line, flush=True)
    LOG.write(line + "\n")
    LOG.flush()


def time_plan(plan, x, y, batch, reps=5, warm=2, code=0):
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, code, stream)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, code, stream)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def report(tag, plan, n, batch, esz, med, best, extra=None):
    flops = 5.0 * n * math.log2(n) * batch
    bytes_ = 2.0 * n * esz * batch
    rec = dict(tag=tag, plan=plan.describe(), n=n, batch=batch, ms_med=round(med * 1e3, 3), ms_best=round(best * 1e3, 3),
               gflops=round(flops / med / 1e9, 1), alg_gbps=round(bytes_ / med / 1e9, 1), frac8=round(bytes_ / med / 8e12, 4))
    if extra:
        rec.update(extra)
    emit(**rec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--what", default="copy,variants,options,sizes")
    args = ap.parse_args()
    what = args.what.split(",")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    emit(tag="device", name=torch.cuda.get_device_name(0), mem_gb=round(torch.cuda.get_device_properties(0).total_memory / 2**30, 1))
    n = 1 << 20
    batch = args.batch
    x = torch.empty((batch, n), dtype=torch.complex64, device=dev)
    torch.view_as_real(x).uniform_(0, 1)
    y = torch.empty_like(x)

    if "copy" in what:
        # achievable-HBM reference: plain device copy, large (streams HBM) and small (fits Infinity Cache)
        for label, rows in (("copy_large", batch), ("copy_64MiB", 8), ("copy_16MiB", 2)):
            a, b = x[:rows], y[:rows]
            for _ in range(3):
                b.copy_(a)
            torch.cuda.synchronize()
            reps = 5 if rows == batch else 200
            t0 = time.perf_counter()
            for _ in range(reps):
                b.copy_(a)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            emit(tag=label, bytes=a.numel() * 8, ms=round(dt * 1e3, 4), gbps_rw=round(2 * a.numel() * 8 / dt / 1e9, 1))

    base_lib = _lib.lib()
    if "variants" in what:
        for path in sorted(glob.glob(os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_*.so"))):
            name = os.path.basename(path)[len("libfourier_"):-3]
            try:
                _lib._lib = _lib.bind(ctypes.CDLL(path))
                plan = F.create_fft_f32(n, 0)
                med, best = time_plan(plan, x, y, batch)
                prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, torch.cuda.current_stream().cuda_stream)
                report("variant:" + name, plan, n, batch, 8, med, best, {"kernels_ms": {k: round(ms, 3) for k, ms, _ in prof}})
                del plan
            except Exception as e:  # keep sweeping
                emit(tag="variant:" + name, error=repr(e))
        _lib._lib = base_lib

    if "xcd" in what:
        for xcd in (0, 1, 2):
            for scratch, chunk in ((0, 0), (1, 0)):
                plan = F.create_fft_f32(n, 0)
                plan.set_option("xcd_swizzle", xcd)
                plan.set_option("scratch", scratch)
                plan.set_option("chunk_bytes", chunk)
                med, best = time_plan(plan, x, y, batch)
                prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, torch.cuda.current_stream().cuda_stream)
                report(f"xcd={xcd},scratch={scratch},chunk={chunk >> 20}MiB", plan, n, batch, 8, med, best,
                       {"kernels_ms": {k: round(ms, 3) for k, ms, _ in prof}})
                del plan

    if "options" in what:
        for inplace in (0, 1):
            for scratch in (0, 1):
                for chunk in (0, 16 << 20, 32 << 20, 64 << 20, 128 << 20, 256 << 20, 1 << 30):
                    if inplace and not scratch:
                        continue  # in-place always routes through scratch
                    plan = F.create_fft_f32(n, 0)
                    plan.set_option("chunk_bytes", chunk)
                    plan.set_option("scratch", scratch)
                    med, best = time_plan(plan, x, x if inplace else y, batch, reps=4, warm=1)
                    report(f"opt:inplace={inplace},scratch={scratch},chunk={chunk >> 20}MiB", plan, n, batch, 8, med, best)
                    del plan
        torch.view_as_real(x).uniform_(0, 1)

    if "sizes" in what:
        del y
        torch.cuda.empty_cache()
        for real, esz in (("f32", 8), ("f64", 16)):
            for lg in (8, 10, 11, 12, 13, 14, 15, 16, 18, 20, 21, 22, 24):
                nn = 1 << lg
                bb = max(1, min((8 << 30) // (nn * esz), 1 << 20))
                cdt = torch.complex64 if real == "f32" else torch.complex128
                xs = torch.empty((bb, nn), dtype=cdt, device=dev)
                torch.view_as_real(xs).uniform_(0, 1)
                ys = torch.empty_like(xs)
                plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(nn, 0)
                med, best = time_plan(plan, xs, ys, bb, reps=4, warm=1)
                report(f"size:{real}:2^{lg}", plan, nn, bb, esz, med, best)
                del xs, ys, plan
                torch.cuda.empty_cache()
        nn, bb = 999983, 512
        xs = torch.empty((bb, nn), dtype=torch.complex64, device=dev)
        torch.view_as_real(xs).uniform_(0, 1)
        ys = torch.empty_like(xs)
        plan = F.create_fft_f32(nn, 0)
        for chunk in (0, 64 << 20, 256 << 20):
            plan.set_option("chunk_bytes", chunk)
            med, best = time_plan(plan, xs, ys, bb, reps=4, warm=1)
            prof = plan.profile_batch_ptr(xs.data_ptr(), ys.data_ptr(), bb, 0, torch.cuda.current_stream().cuda_stream)
            report(f"size:f32:999983:chunk={chunk >> 20}MiB", plan, nn, bb, 8, med, best, {"kernels_ms": {k: round(ms, 3) for k, ms, _ in prof}})


if __name__ == "__main__":
    main()
