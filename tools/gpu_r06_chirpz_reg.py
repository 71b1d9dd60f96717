#!/usr/bin/env python3
"""Development tool (round 6, session 43): the one-launch chirp-z on a smooth M = R1 x R2 in registers (kernels_chirpz.h, plan option
bluestein_smooth_m = 2) against the power-of-two one-launch kernels (= 0) and the default rule (= 1), alternating on shared buffers: median ms of
7, fraction of the 8 TB/s HBM peak on the algorithmic bytes, rel-L2 error against torch's f64 FFT.  One JSON line per (precision, n, arm).
usage: gpu_r06_chirpz_reg.py [lib=path ...]   (CHIRPZ_SIZES=n,n,...: lengths; default: the longest Bluestein length under every M of the menu
and the reference's prime / composite bench sets)"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

MENU = [36, 49, 64, 81, 100, 120, 144, 168, 196, 225, 256, 288, 324, 360, 400, 441, 480, 525, 576, 625, 675, 729, 784, 840, 900, 960, 1024, 1152]
MENU3 = [1296, 1440, 1600, 2304, 2560, 3072, 8820, 9261]  # three stages
REPS = 7


def rough(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n > 1


def main():
    sizes = [int(v) for v in os.environ.get("CHIRPZ_SIZES", "").split(",") if v]
    if not sizes:
        menu = MENU3 if os.environ.get("CHIRPZ_MENU") == "3" else MENU
        sizes = sorted({next(v for v in range((m + 1) // 2, 0, -1) if rough(v)) for m in menu} | ({722, 1013, 1418, 4097, 2053, 1031} if menu is MENU3 else {191, 222, 439, 722, 37, 97, 331}))
    libs = [("product", _lib.lib())]
    for spec in sys.argv[1:]:
        name, _, path = spec.partition("=")
        libs.append((name, _lib.bind(ctypes.CDLL(path), strict=False)))
    base = libs[0][1]
    st = torch.cuda.current_stream().cuda_stream
    for real, cdt, esz in (("f32", torch.complex64, 8), ("f64", torch.complex128, 16)):
        for n in sizes:
            batch = max(1, (1 << 29) // (n * esz))
            x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
            ref = torch.fft.fft(x[:64].to(torch.complex128), dim=1)
            plans = []
            for lname, L in libs:
                _lib._lib = L
                for arm, opt in ((("pow2", 0), ("default", 1), ("registers", 2)) if lname == "product" else (("registers", 2),)):
                    p = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
                    try:
                        p.set_option("bluestein_smooth_m", opt)
                    except Exception:
                        continue
                    if arm == "registers" and "registers" not in p.describe():
                        continue
                    plans.append((arm if lname == "product" else lname, p, []))
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
