#!/usr/bin/env python3
"""Round-2 A/B on one GPU (development tool): (1) the L = 2048 passes, 16-column one-workgroup-per-CU kernels
(FOURIER_WIDE_2048=1) against the narrow first pass + split last pass; (2) the XCD-fused one-launch plan for
2^16..2^18 (f32) / 2^15..2^17 (f64) against the two-launch plan, over window depths and builds."""
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from fourier_amd import _lib, fft as F  # noqa: E402


def timeit(plan, x, y, batch, reps=7, warm=2):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def emit(**kw):
    print(json.dumps(kw), flush=True)


def run(tag, n, batch, real="f32", opts=(), env=None, check=None):
    cdt = torch.complex64 if real == "f32" else torch.complex128
    esz = 8 if real == "f32" else 16
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
    finally:
        for k in (env or {}):
            del os.environ[k]
    for k, v in opts:
        plan.set_option(k, v)
    x = torch.empty((batch, n), dtype=cdt, device="cuda")
    torch.view_as_real(x).uniform_(-1, 1)
    y = torch.empty_like(x)
    t = timeit(plan, x, y, batch)
    prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, torch.cuda.current_stream().cuda_stream)
    rec = dict(tag=tag, plan=plan.describe(), n=n, batch=batch, opts=dict(opts), env=env or {}, ms=round(t * 1e3, 3),
               alg_tbps=round(batch * 2 * n * esz / t / 1e12, 3), frac8=round(batch * 2 * n * esz / t / 8e12, 4),
               kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})
    if check is not None:  # compare with a reference result (same input): max abs difference
        torch.manual_seed(7)
        xs = torch.randn((4, n), dtype=cdt, device="cuda")
        ys = torch.empty_like(xs)
        plan.transform_batch_ptr(xs.data_ptr(), ys.data_ptr(), 4, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref = check(xs)
        rec["rel_l2_vs_torch_fft_f64"] = float((ys.to(torch.complex128) - ref).norm() / ref.norm())
    emit(**rec)
    del x, y, plan
    torch.cuda.empty_cache()


def torch_ref(xs):
    return torch.fft.fft(xs.to(torch.complex128), dim=1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "l2048"):
        for env in ({"FOURIER_WIDE_2048": "1"}, None, {"FOURIER_WIDE_2048": "1"}, None):
            run("2^21", 1 << 21, 1024, env=env, check=torch_ref)
            run("2^22 (C5 chunk)", 1 << 22, 1024, env=env, check=torch_ref)
            run("C4", 999983, 512, env=env)
            run("2^21 f64", 1 << 21, 512, "f64", env=env)
    if which in ("all", "p4096"):
        for env in (None, {"FOURIER_PLAN_4096": "1"}, None, {"FOURIER_PLAN_4096": "1"}):
            run("2^22 (C5 chunk)", 1 << 22, 1024, env=env, check=torch_ref)
        for env in (None, {"FOURIER_PLAN_4096": "1"}):
            run("2^22 f64", 1 << 22, 512, "f64", env=env, check=torch_ref)
    if which in ("all", "conv"):
        for opts in ((), (("xcd_swizzle", 3),), (), (("xcd_swizzle", 3),)):
            run("C4", 999983, 512, opts=opts)
        run("N=65537", 65537, 8192)
    if which in ("p23",):
        for env in ({"FOURIER_THREE_PASS_2P23": "1"}, None, {"FOURIER_THREE_PASS_2P23": "1"}, None):
            run("2^23", 1 << 23, 512, env=env, check=torch_ref)
            run("2^23 f64", 1 << 23, 256, "f64", env=env, check=torch_ref)
    if which in ("blu_order",):
        for env in (None, {"FOURIER_BLU_SHORT_FIRST": "1"}, None, {"FOURIER_BLU_SHORT_FIRST": "1"}):
            run("C4", 999983, 512, env=env, check=torch_ref)
            run("N=40000", 40000, 8192, env=env)
            run("C4 f64", 999983, 256, "f64", env=env)
    if which in ("chirp",):
        for rep in range(2):
            for ev in (1, 0):
                run("C4", 999983, 512, opts=(("bluestein_chirp_eval", ev),), check=torch_ref)
                run("N=65537", 65537, 8192, opts=(("bluestein_chirp_eval", ev),))
                run("N=40000", 40000, 8192, opts=(("bluestein_chirp_eval", ev),))
                run("C4 f64", 999983, 256, "f64", opts=(("bluestein_chirp_eval", ev),), check=torch_ref)
    if which in ("all", "fused"):
        for real, ks in (("f32", (16, 17, 18)), ("f64", (15, 16, 17))):
            esz = 8 if real == "f32" else 16
            for k in ks:
                n = 1 << k
                batch = (8 << 30) // (n * esz)
                run(f"2^{k} {real} two-launch", n, batch, real, check=torch_ref)
                for depth in (2, 3, 4, 6):
                    run(f"2^{k} {real} fused d{depth}", n, batch, real, opts=(("l2_fused", 1), ("l2_fused_depth", depth)),
                        check=torch_ref if depth == 2 else None)
                for grid in (512, 768):
                    run(f"2^{k} {real} fused d4 grid{grid}", n, batch, real, opts=(("l2_fused", 1), ("l2_fused_depth", 4), ("l2_fused_grid", grid)))


if __name__ == "__main__":
    libs = [None]
    if "--only-variants" in sys.argv:
        sys.argv[sys.argv.index("--only-variants")] = "--variants"
        libs = []
    if "--variants" in sys.argv:
        sys.argv.remove("--variants")
        libs += [n for n in sys.argv[2:] if os.path.exists(os.path.join(ROOT, "fourier_amd/lib/variants", f"libfourier_{n}.so"))]
    for name in libs:
        if name:
            _lib._lib = _lib.bind(ctypes.CDLL(os.path.join(ROOT, "fourier_amd/lib/variants", f"libfourier_{name}.so")))
        emit(lib=name or "product")
        main()
