#!/usr/bin/env python3
"""Development tool (round 6, session 11): do the last pass's FINAL stores without the streaming hint (and its loads without it) change the
slow allocations?  Per fresh allocation of the C3 / C2 buffers: the passes of the product library and of the A/B builds, HIP events, median of 3.
usage: gpu_r06_placement4.py KIND lib=path ..."""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import _lib, fft as F

KIND = sys.argv[1]
N, BATCH, REAL, ESZ = {"c3": (1 << 20, 4096, "f64", 16), "c2": (1 << 20, 4096, "f32", 8), "c5": (1 << 22, 1024, "f32", 8)}[KIND]
BYTES = N * BATCH * ESZ


def main():
    st = torch.cuda.current_stream().cuda_stream
    base = _lib.lib()
    libs = [("product", base)] + [(a.split("=")[0], _lib.bind(ctypes.CDLL(os.path.join(ROOT, a.split("=")[1])), strict=False)) for a in sys.argv[2:]]
    plans = []
    for name, L in libs:
        _lib._lib = L
        plans.append((name, (F.create_fft_f32 if REAL == "f32" else F.create_fft_f64)(N, 0)))
    _lib._lib = base
    fdt = torch.float32 if REAL == "f32" else torch.float64

    def prof(plan, xp, yp):
        acc = {}
        for _ in range(3):
            for name, ms, cnt in plan.profile_batch_ptr(xp, yp, BATCH, 0, st):
                if cnt:
                    acc.setdefault(name, []).append(ms)
        return [round(statistics.median(acc[k]), 3) for k in ("pass0", "pass1")]

    for si, (name, filler_mb) in enumerate([("x_then_y", 0), ("y_then_x_filler1027m", 1027), ("x_then_y_filler20g", 20000), ("y_then_x", 0), ("y_then_x_filler40g", 40000), ("x_then_y_again", 0)]):
        fillt = torch.empty(filler_mb << 20, dtype=torch.uint8, device="cuda") if filler_mb else None
        if name.startswith("y_then_x"):
            Y = torch.empty(BYTES, dtype=torch.uint8, device="cuda"); X = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
        else:
            X = torch.empty(BYTES, dtype=torch.uint8, device="cuda"); Y = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
        X.view(fdt).uniform_(0, 1)
        row = {n: prof(p, X.data_ptr(), Y.data_ptr()) for n, p in plans}
        print(json.dumps(dict(kind=KIND, pid=os.getpid(), scenario=name, y_ptr=hex(Y.data_ptr()), passes=row)), flush=True)
        del X, Y, fillt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
