#!/usr/bin/env python3
"""Development tool: A/B of plan options (and of library variants) on shared device buffers, alternating the arms.

usage: gpu_ab_options.py WORKLOAD... [--arms NAME=opt:val,opt:val ...] [--libs NAME=path ...] [--reps R]
  WORKLOAD = n:batch[:f32|f64]   (n may be written 2^k)
Every arm is one plan (library x option set); arms run round-robin R times on the same x -> y buffers; the line of an arm
carries its median time, the per-kernel times of one profiled call, the fraction of the 8 TB/s HBM peak on the algorithmic
bytes, and whether its output equals the first arm's bit for bit."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import _lib, fft as F


def parse_n(s):
    return (1 << int(s[2:])) if s.startswith("2^") else int(s)


def main(argv):
    workloads, arms, libs, reps = [], [], {"product": None}, 5
    it = iter(argv)
    for a in it:
        if a == "--arms":
            for spec in it:
                if spec.startswith("--"):
                    a = spec
                    break
                name, _, opts = spec.partition("=")
                arms.append((name, [(o.split(":")[0], int(o.split(":")[1])) for o in opts.split(",") if o]))
            else:
                continue
        if a == "--libs":
            for spec in it:
                if spec.startswith("--"):
                    a = spec
                    break
                name, _, path = spec.partition("=")
                libs[name] = path
            else:
                continue
        if a == "--reps":
            reps = int(next(it))
        elif not a.startswith("--"):
            f = a.split(":")
            workloads.append((parse_n(f[0]), int(f[1]), f[2] if len(f) > 2 else "f32"))
    arms = arms or [("default", [])]
    product = _lib.lib()
    handles = {"product": product}
    for name, path in libs.items():
        if path:
            handles[name] = _lib.bind(ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path), strict=False)
    st = torch.cuda.current_stream().cuda_stream
    for n, batch, real in workloads:
        cdt = torch.complex64 if real == "f32" else torch.complex128
        esz = 8 if real == "f32" else 16
        x = torch.empty((batch, n), dtype=cdt, device="cuda")
        torch.view_as_real(x).uniform_(0, 1)
        y = torch.empty_like(x)
        plans = []
        for lname, L in handles.items():
            for aname, opts in arms:
                _lib._lib = L
                plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
                try:
                    for k, v in opts:
                        plan.set_option(k, v)
                except Exception:  # an option this library does not have (experiments-only options on the product library)
                    continue
                plans.append((f"{lname}/{aname}", plan, []))
        _lib._lib = product
        first = None
        for name, plan, ts in plans:  # warm-up + value comparison
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            torch.cuda.synchronize()
            if first is None:
                first = y.clone() if y.numel() * y.element_size() <= (40 << 30) else None
                ts.append(None)
            else:
                ts.append(None if first is None else bool(torch.equal(torch.view_as_real(y), torch.view_as_real(first))))
        for _ in range(reps):
            for name, plan, ts in plans:
                t0 = time.perf_counter()
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        for name, plan, ts in plans:
            same, times = ts[0], sorted(ts[1:])
            t = times[len(times) // 2]
            prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            print(json.dumps(dict(arm=name, plan=plan.describe(), n=n, batch=batch, real=real, ms=round(t * 1e3, 3), ms_min=round(times[0] * 1e3, 3),
                                  frac8=round(batch * 2 * n * esz / t / 8e12, 4), equals_first_arm=same,
                                  kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
        del x, y, plans, first
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main(sys.argv[1:])
