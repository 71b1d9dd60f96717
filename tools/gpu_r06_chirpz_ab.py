#!/usr/bin/env python3
"""Development tool (round 6): the one-launch kernels (both passes of 2^11..2^15; the whole chirp-z for M <= 2^15) with
packed f32 arithmetic (the product build) against the scalar build (lib/variants/libfourier_onelaunch_scalar.so), alternating on shared buffers:
median ms of 7, fraction of the 8 TB/s HBM peak on the algorithmic bytes, rel-L2 error against torch's f64 FFT.  One JSON line per (size, arm)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

SIZES = [int(v) for v in os.environ.get("CHIRPZ_SIZES", "").split(",") if v] or [191, 222, 439, 722, 1013, 1418, 37, 97, 331, 2039, 4097, 5003, 10007, 16381, 2048, 4096, 8192, 16384, 32768]
REPS = 7


def main():
    base = _lib.lib()
    libs = [("packed", base)]
    vdir = os.path.join(ROOT, "fourier_amd", "lib", "variants")
    for name in sys.argv[1:] or ["onelaunch_scalar"]:
        libs.append((name, _lib.bind(ctypes.CDLL(os.path.join(vdir, f"libfourier_{name}.so")), strict=False)))
    st = torch.cuda.current_stream().cuda_stream
    for real, cdt, esz in (("f32", torch.complex64, 8), ("f64", torch.complex128, 16)):
        for n in SIZES:
            batch = max(1, (1 << 30) // (n * esz))
            x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
            ref = torch.fft.fft(x[:64].to(torch.complex128), dim=1)
            plans = []
            for name, L in libs:
                _lib._lib = L
                plans.append((name, (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0), []))
            _lib._lib = base
            errs = {}
            for name, plan, ts in plans:
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                errs[name] = float((y[:64].to(torch.complex128) - ref).norm() / ref.norm())
            for _ in range(REPS):
                for name, plan, ts in plans:
                    t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
            for name, plan, ts in plans:
                t = sorted(ts)[len(ts) // 2]
                print(json.dumps(dict(real=real, n=n, arm=name, plan=plan.describe(), batch=batch, ms=round(t * 1e3, 3), ms_min=round(min(ts) * 1e3, 3),
                                      frac8=round(batch * 2.0 * n * esz / t / 8e12, 4), rel_l2_vs_torch_f64=errs[name])), flush=True)
            del x, y, plans, ref
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
