#!/usr/bin/env python3
"""Development tool (round 6, sessions 48 / 49): lengths with factors 5 ... 13 as a direct transform on register stages (kernels_regfft.h; the
product library's route for the lengths of regfft_shapes.h) against the route they had before (the LDS mixed-radix kernels, per length or
runtime-parameterised, or tile passes: the experiments library under FOURIER_NO_REGFFT=1), alternating on shared buffers: median ms of REPS by
HIP events on the launch stream, fraction of the 8 TB/s HBM peak, rel-L2 error of four transforms against numpy's f64 FFT.
REGFFT_SPECIALISED=n,n,...: a third arm for these lengths, the length's own LDS kernel compiled at run time (plan option "specialise").
REGFFT_VARIANTS=1 (an --ab-build of regfft_shapes.h, sessions 53 / 54): the three-stage lengths only, arms plain / split / fact / splitfact
(whole or split-plane exchanges x whole or factored twiddle tables, FOURIER_REGFFT_VARIANT = 1 ... 4) / before, all on the experiments library."""
import ctypes, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fourier_amd import fft as F, _lib, build as B


VARIANTS = os.environ.get("REGFFT_VARIANTS") in ("1", "2")
OPTIN = os.environ.get("REGFFT_VARIANTS") == "3"  # the 2^a 3^b lengths listed on request: plan option "register_stages" = 1 against the default plan
UNPAIRED = os.environ.get("REGFFT_VARIANTS") == "2"  # an --unpaired-build: f32 only, arms listed / unpaired / unpairedfact / before


def listed():
    with open(os.path.join(ROOT, "fourier_amd", "csrc", "regfft_shapes.h")) as f:
        if OPTIN:
            return [int(m.group(1)) for m in re.finditer(r"^FOURIER_REGFFT_OPT_ROW\((\d+),", f.read(), re.M)]
        return [int(m.group(1)) for m in re.finditer(r"^FOURIER_REGFFT_ROW\((\d+), \d+, \d+, (\d+),", f.read(), re.M) if not VARIANTS or int(m.group(2))]


SIZES = [int(v) for v in os.environ.get("REGFFT_SIZES", "").split(",") if v] or listed()
SPECIALISED = {int(v) for v in os.environ.get("REGFFT_SPECIALISED", "").split(",") if v}
REPS = int(os.environ.get("REGFFT_REPS", "7"))
BYTES = 1 << 29


def main():
    base = _lib.lib()
    exp = _lib.bind(ctypes.CDLL(B.OUT_EXPERIMENTS), strict=False)
    st = torch.cuda.current_stream().cuda_stream
    reals = [v for v in os.environ.get("REGFFT_REALS", "").split(",") if v] or (["f32"] if UNPAIRED else ["f32", "f64"])
    for real, cdt, esz in (("f32", torch.complex64, 8), ("f64", torch.complex128, 16)):
        if real not in reals:
            continue
        xbuf = torch.empty(BYTES // esz, dtype=cdt, device="cuda"); torch.view_as_real(xbuf).uniform_(-1, 1); ybuf = torch.empty_like(xbuf)
        mk = F.create_fft_f32 if real == "f32" else F.create_fft_f64
        for n in SIZES:
            batch = max(1, BYTES // (n * esz))
            x, y = xbuf[: batch * n].view(batch, n), ybuf[: batch * n].view(batch, n)
            ref = np.fft.fft(x[:4].cpu().numpy().astype(np.complex128), axis=1)
            def under(env, lib):
                os.environ.update(env)
                _lib._lib = lib
                try:
                    return mk(n, 0)
                finally:
                    _lib._lib = base
                    for k in env:
                        del os.environ[k]
            if OPTIN:
                p = mk(n, 0)
                try:
                    p.set_option("register_stages", 1)
                except Exception:  # noqa: BLE001 -- no kernel listed in this precision
                    continue
                plans = [("registers", p, []), ("before", mk(n, 0), [])]
            elif VARIANTS:
                arms = (("listed", 0), ("unpaired", 5), ("unpairedfact", 6)) if UNPAIRED else (("plain", 1), ("split", 2), ("fact", 3), ("splitfact", 4))
                plans = [(name, under({"FOURIER_REGFFT_VARIANT": str(v)}, exp), []) for name, v in arms]
            else:
                plans = [("registers", mk(n, 0), [])]
            if not OPTIN:
                plans.append(("before", under({"FOURIER_NO_REGFFT": "1"}, exp), []))
            if n in SPECIALISED:
                p = under({"FOURIER_NO_REGFFT": "1"}, exp)
                try:
                    p.set_option("specialise", 1)
                    plans.append(("specialised", p, []))
                except Exception as e:  # noqa: BLE001 -- a length the run-time route refuses stays out of the table
                    print(f"specialise({n}) refused: {e}", file=sys.stderr)
            plans = [p for p in plans if p[0] in ("before", "specialised") or "registers" in p[1].describe()]  # (an arm without a kernel in this precision)
            if plans[0][0] in ("before", "specialised"):
                continue
            errs = {}
            for name, plan, ts in plans:
                y[:4].zero_()
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                got = y[:4].cpu().numpy().astype(np.complex128)
                errs[name] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            for _ in range(REPS):
                for name, plan, ts in plans:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); e1.record(); e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e-3)
            for name, plan, ts in plans:
                t = sorted(ts)[len(ts) // 2]
                print(json.dumps(dict(real=real, n=n, arm=name, plan=plan.describe(), batch=batch, ms=round(t * 1e3, 4), ms_min=round(min(ts) * 1e3, 4),
                                      frac8=round(batch * 2.0 * n * esz / t / 8e12, 4), rel_l2_vs_numpy_f64=errs[name])), flush=True)
            del plans
        del xbuf, ybuf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
