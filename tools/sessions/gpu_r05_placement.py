#!/usr/bin/env python3
"""Development tool (round 5): does the time of the C2 pass kernels depend on WHERE the buffers lie?

One binary's passes were seen at 10.9 ... 12.2 ms across sessions ("buffer placement") and nobody had looked.
Two sweeps on one box, one process:
  * virtual offsets: in = X + dx, out = Y + dy inside two over-allocated buffers, dx / dy from 0 to 1 GiB in steps that change
    the low address bits (4 KiB ... 2 MiB: page offset, channel interleave) and the high ones (16 MiB ... 1 GiB);
  * fresh allocations: the buffers freed and allocated again behind differently sized filler allocations, offset 0.
Per point: the product plan's two passes, their load / store skeletons (experiments library, option "skeleton"), and the
column-tile copy of exp_copy_ceiling.cpp -- HIP events per kernel, median of `reps` profiled calls.  One JSON line per point."""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import _lib, fft as F, build as B

N, BATCH = 1 << 20, int(os.environ.get("PLACEMENT_BATCH", "4096"))
BYTES = N * BATCH * 8
PAD = 1 << 30
REPS = int(os.environ.get("PLACEMENT_REPS", "5"))


def med_kernels(plan, xp, yp, st):
    acc = {}
    for _ in range(REPS):
        for name, ms, cnt in plan.profile_batch_ptr(xp, yp, BATCH, 0, st):
            if cnt:
                acc.setdefault(name, []).append(ms)
    return {k: round(statistics.median(v), 3) for k, v in acc.items()}


def main():
    product = _lib.lib()
    exp = _lib.bind(ctypes.CDLL(B.OUT_EXPERIMENTS), strict=False)
    craw = ctypes.CDLL(B.OUT_EXPERIMENTS)
    f = craw.fourier_exp_copy_ceiling
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.POINTER(ctypes.c_float)]
    st = torch.cuda.current_stream().cuda_stream
    plan = F.create_fft_f32(N, 0)
    _lib._lib = exp
    skel = F.create_fft_f32(N, 0)
    skel.set_option("skeleton", 1)
    _lib._lib = product

    def copy_ms(xp, yp):
        ms = ctypes.c_float(0)
        nb = min(BYTES, 16 << 30)
        rc = f(xp, yp, nb, 1 << 20, 3, 0, 3, st, ctypes.byref(ms))
        return round(ms.value * BYTES / nb, 3) if rc == 0 else None

    def point(tag, xp, yp, **kw):
        k = med_kernels(plan, xp, yp, st)
        s = med_kernels(skel, xp, yp, st)
        print(json.dumps(dict(tag=tag, in_mod_2m=xp % (2 << 20), out_mod_2m=yp % (2 << 20), diff_mod_1g=(yp - xp) % (1 << 30),
                              product_ms=k, skeleton_ms=s, copy_tiles_streaming_ms_scaled=copy_ms(xp, yp), **kw)), flush=True)

    X = torch.empty(BYTES + PAD, dtype=torch.uint8, device="cuda")
    Y = torch.empty(BYTES + PAD, dtype=torch.uint8, device="cuda")
    X[: BYTES].view(torch.float32).uniform_(0, 1)
    X[BYTES:].zero_()
    xp0, yp0 = X.data_ptr(), Y.data_ptr()
    K, M = 1 << 10, 1 << 20
    offs = [0, 4 * K, 8 * K, 16 * K, 64 * K, 128 * K, 512 * K, M, 2 * M, 2 * M + 4 * K, 8 * M, 16 * M, 64 * M, 256 * M, 512 * M, 1024 * M - 2 * M]
    for rep in range(2):  # the whole sweep twice: what repeats is placement, what does not is noise
        for d in offs:
            point("out_offset", xp0, yp0 + d, dx=0, dy=d, sweep=rep)
        for d in offs[1:]:
            point("in_offset", xp0 + d, yp0, dx=d, dy=0, sweep=rep)
        for d in (4 * K, M, 64 * M):
            point("both_offset", xp0 + d, yp0 + d, dx=d, dy=d, sweep=rep)
    del X, Y
    torch.cuda.empty_cache()
    for i, filler_mb in enumerate((0, 3, 1027, 40000, 5)):
        fill = torch.empty(filler_mb << 20, dtype=torch.uint8, device="cuda") if filler_mb else None
        Y = torch.empty(BYTES, dtype=torch.uint8, device="cuda")  # (the other order than above)
        X = torch.empty(BYTES, dtype=torch.uint8, device="cuda")
        X.view(torch.float32).uniform_(0, 1)
        point("fresh_alloc", X.data_ptr(), Y.data_ptr(), filler_mb=filler_mb, x_ptr=hex(X.data_ptr()), y_ptr=hex(Y.data_ptr()))
        del X, Y, fill
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
