#!/usr/bin/env python3
"""Development tool (round 6, session 3): does a per-XCD rotation of the tile index (plan option "xcd_rotate") take the slow mode away?

gpu_r06_placement2.py: a last pass over ONE 4 GiB chunk runs in the slow mode almost everywhere (f64: 1.55 ms = 24.8 ms per batch) and
fast (1.39) on a few chunks, while the whole batch is fast or slow by allocation -- physically contiguous buffers + XCD ranges a power of two
apart = eight address streams that agree in every low bit.  Arms: rotation of the last pass / of the first pass, with and without odd XCDs
walking backwards; per fresh allocation the whole batch and four single chunks, HIP events per kernel, median of 3.  One JSON line per allocation.
usage: gpu_r06_placement3.py KIND [tag]   KIND = c3 | c2 | c5"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

KIND = sys.argv[1] if len(sys.argv) > 1 else "c3"
TAG = sys.argv[2] if len(sys.argv) > 2 else ""
N, BATCH, REAL, ESZ, CH = {"c3": (1 << 20, 4096, "f64", 16, 256), "c2": (1 << 20, 4096, "f32", 8, 512), "c5": (1 << 22, 1024, "f32", 8, 128)}[KIND]
BYTES = N * BATCH * ESZ
NCH = BATCH // CH
st = None
L, F1, BL, BF = 0, 12, 24, 25
ARMS = [("default", 0), ("last1", 1), ("last2", 2), ("last3", 3), ("last4", 4), ("last5", 5), ("last8", 8), ("last11", 11), ("last16", 16), ("last32", 32),
        ("last0_back", 1 << BL), ("last1_back", 1 | 1 << BL), ("last8_back", 8 | 1 << BL),
        ("first1", 1 << F1), ("first4", 4 << F1), ("first8", 8 << F1), ("first16", 16 << F1), ("first3", 3 << F1), ("first0_back", 1 << BF),
        ("both8", 8 | 8 << F1), ("both1", 1 | 1 << F1), ("both3", 3 | 3 << F1)]
if os.environ.get("PLACEMENT3_PHASE"):  # session 4: a per-XCD phase inside its own range of transforms instead (both passes)
    ARMS = [("default", 0)] + [(f"lastphase{v}", ("xcd_phase", v)) for v in (1, 3, 8, 21, 37, 64, 101)]


def make(rot):
    p = (F.create_fft_f32 if REAL == "f32" else F.create_fft_f64)(N, 0)
    if isinstance(rot, tuple):  # ("xcd_phase", transforms)
        p.set_option(rot[0], rot[1])
    elif rot:
        p.set_option("xcd_rotate", rot)
    return p


def prof(plan, xp, yp, batch, reps=3):
    acc = {}
    for _ in range(reps):
        for name, ms, cnt in plan.profile_batch_ptr(xp, yp, batch, 0, st):
            if cnt:
                acc.setdefault(name, []).append(ms)
    return [round(statistics.median(acc[k]), 3) for k in ("pass0", "pass1")]


def emit(**kw):
    print(json.dumps(dict(kind=KIND, proc=TAG, pid=os.getpid(), **kw)), flush=True)


def main():
    global st
    st = torch.cuda.current_stream().cuda_stream
    plans = [(n, make(r)) for n, r in ARMS]
    fdt = torch.float32 if REAL == "f32" else torch.float64

    def alloc(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    for si, (name, filler_mb) in enumerate([("x_then_y", 0), ("y_then_x_filler1027m", 1027), ("x_then_y_filler20g", 20000), ("y_then_x", 0)]):
        fillt = alloc(filler_mb << 20) if filler_mb else None
        if name.startswith("y_then_x"):
            Y = alloc(BYTES); X = alloc(BYTES)
        else:
            X = alloc(BYTES); Y = alloc(BYTES)
        X.view(fdt).uniform_(0, 1)
        xp, yp = X.data_ptr(), Y.data_ptr()
        cb = CH * N * ESZ
        whole = {an: prof(p, xp, yp, BATCH) for an, p in plans}
        chunks = {}
        for j in (0, NCH // 3, 2 * NCH // 3, NCH - 1):
            chunks[j] = {an: prof(p, xp + j * cb, yp + j * cb, CH) for an, p in plans}
        emit(tag="alloc", scenario=name, idx=si, x_ptr=hex(xp), y_ptr=hex(yp), whole=whole, chunks=chunks)
        del X, Y, fillt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
