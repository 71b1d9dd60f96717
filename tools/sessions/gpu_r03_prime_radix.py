#!/usr/bin/env python3
"""Round 3 A/B: lengths with prime factors 5..13 as LDS Stockham passes (default: per-length kernel where instantiated,
runtime-parameterised kernel otherwise) against (a) the runtime kernel forced (experiments library, FOURIER_MIX_GENERIC=1),
(b) the Bluestein route the reference takes (FOURIER_MIX_REFERENCE_RADICES=1) and (c) variant libraries given on the
command line (e.g. mix_ref_order: the reference's radix order continued instead of odd radices first)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib, build

exp = _lib.bind(ctypes.CDLL(build.OUT_EXPERIMENTS))
base = _lib.lib()
variants = {}
for name in sys.argv[1:]:
    path = os.path.join(ROOT, "fourier_amd", "lib", "variants", f"libfourier_{name}.so")
    variants[name] = _lib.bind(ctypes.CDLL(path), strict=False)
SIZES = [(125, "f32"), (625, "f32"), (3125, "f32"), (15625, "f32"), (100, "f32"), (1000, "f32"), (2500, "f32"), (5000, "f32"), (10000, "f32"), (16000, "f32"),
         (12500, "f32"), (16807, "f32"), (2401, "f32"), (1001, "f32"), (4095, "f32"), (5005, "f32"), (960, "f32"), (7680, "f32"),
         (1200, "f32"), (1920, "f32"), (3600, "f32"), (6000, "f32"), (7200, "f32"), (3000, "f32"), (4500, "f32"),
         (1920, "f64"), (6000, "f64"), (5005, "f64"), (125, "f64"), (625, "f64"), (3125, "f64"), (1000, "f64"), (2500, "f64"), (5000, "f64"),
         (8000, "f64"), (2401, "f64"), (4095, "f64")]
if os.environ.get("PRIME_RADIX_SIZES"):
    SIZES = [(int(t.split(":")[0]), t.split(":")[1]) for t in os.environ["PRIME_RADIX_SIZES"].split(",")]
for n, real in SIZES:
    esz = 8 if real == "f32" else 16
    batch = max(2, (1 << 30) // (n * esz))
    cdt = torch.complex64 if real == "f32" else torch.complex128
    x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(0, 1); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.fft.fft(x[:3].to(torch.complex128))
    routes = [("default", base, {})] + ([] if os.environ.get("PRIME_RADIX_ONLY_VARIANTS") else [("runtime-kernel", exp, {"FOURIER_MIX_GENERIC": "1"}), ("bluestein", exp, {"FOURIER_MIX_REFERENCE_RADICES": "1"})])
    routes += [(name, lib, {}) for name, lib in variants.items()]
    for route, lib, env in routes:
        for k in ("FOURIER_MIX_GENERIC", "FOURIER_MIX_REFERENCE_RADICES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        _lib._lib = lib
        try:
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
            for _ in range(2):
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
            torch.cuda.synchronize(); ts = []
            for _ in range(5):
                t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            t = sorted(ts)[2]
            got = y[:3].to(torch.complex128)
            if route == "default":
                keep = y[:64].clone()
            same = bool(torch.equal(y[:64], keep))
            print(json.dumps(dict(n=n, real=real, batch=batch, route=route, plan=plan.describe(), ms=round(t * 1e3, 3), us_per_transform=round(t / batch * 1e6, 4),
                                  frac8=round(batch * 2 * n * esz / t / 8e12, 4), rel_l2_vs_f64_truth=float(torch.linalg.norm(got - ref) / torch.linalg.norm(ref)), same_bits_as_default=same)), flush=True)
            del plan
        except Exception as e:  # e.g. the runtime kernel's two LDS buffers do not fit
            print(json.dumps(dict(n=n, real=real, route=route, error=str(e)[:200])), flush=True)
    _lib._lib = base
    del x, y; torch.cuda.empty_cache()
