#!/usr/bin/env python3
"""Development tool (round 6, session 17): the FIRST pass of length 2048 -- 8-column tiles on two workgroups per CU (the default since round 2) against
16-column tiles on one workgroup per CU (experiments switch FOURIER_WIDE_2048), the latter with and without the streaming hint on its loads, and the
default with plain stores; 2^21, 2^22 (f32, f64), C4.  Alternating arms on shared buffers, per-kernel HIP events of one profiled call."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import _lib, fft as F

V = os.path.join(ROOT, "fourier_amd", "lib", "variants")
ARMS = [("product", None, {}), ("wide_first", "libfourier_exp_base.so", {"FOURIER_WIDE_2048": "1"}),
        ("wide_first_plain_loads", "libfourier_exp_wide_first_plain.so", {"FOURIER_WIDE_2048": "1"}), ("narrow_first_plain_stores", "libfourier_narrow_first_st_plain.so", {})]


def main():
    base = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n, batch, real in ((1 << 22, 1024, "f32"), (1 << 21, 1024, "f32"), (999983, 512, "f32"), (1 << 22, 512, "f64")):
        cdt = torch.complex64 if real == "f32" else torch.complex128
        esz = 8 if real == "f32" else 16
        x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
        plans = []
        for name, lib, env in ARMS:
            L = base if lib is None else _lib.bind(ctypes.CDLL(os.path.join(V, lib)), strict=False)
            _lib._lib = L
            for k, v in env.items():
                os.environ[k] = v
            plans.append((name, (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0), []))
            for k in env:
                os.environ.pop(k)
        _lib._lib = base
        first = None
        for name, plan, ts in plans:
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
            if first is None:
                first = y.clone()
                ts.append(None)
            else:
                ts.append(bool(torch.equal(torch.view_as_real(y), torch.view_as_real(first))))
        for _ in range(7):
            for name, plan, ts in plans:
                t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
        for name, plan, ts in plans:
            t = sorted(ts[1:])[3]
            prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            print(json.dumps(dict(n=n, real=real, batch=batch, arm=name, plan=plan.describe(), ms=round(t * 1e3, 3), equals_first_arm=ts[0],
                                  frac8=round(batch * 2 * n * esz / t / 8e12, 4), kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
        del x, y, first, plans
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
