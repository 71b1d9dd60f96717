#!/usr/bin/env python3
"""Development tool (round 6, session 2): WHERE inside a "slow" allocation is the time lost?

gpu_r06_placement.py showed: the mode of the f64 1024 x 1024 last pass (21.9 / 24.4 ms) belongs to the ALLOCATION of the output (the same
virtual address is fast after one hipMalloc and slow after another; virtual offsets inside an allocation change nothing; every tile order is
slow on a slow allocation), and it is the WRITES into it that are slow (an in-place call -- pass 0 reads y, pass 1 writes y -- has a fast pass 0).
This tool times the passes CHUNK BY CHUNK (4 GiB of f64 = 256 transforms) over a fresh pair of buffers: is a slow allocation slow everywhere or in
a sub-range (a physical region)?  Also: the roles swapped (y -> x), a linear fill and a linear copy per chunk, several tile orders on the slowest chunk.
usage: gpu_r06_placement2.py KIND [tag]   KIND = c3 | c2"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

KIND = sys.argv[1] if len(sys.argv) > 1 else "c3"
TAG = sys.argv[2] if len(sys.argv) > 2 else ""
N, BATCH, REAL, ESZ = {"c3": (1 << 20, 4096, "f64", 16), "c2": (1 << 20, 4096, "f32", 8)}[KIND]
BYTES = N * BATCH * ESZ
CH = 256 if KIND == "c3" else 512  # transforms per chunk: 4 GiB
NCH = BATCH // CH
st = None


def make(opts=()):
    p = (F.create_fft_f32 if REAL == "f32" else F.create_fft_f64)(N, 0)
    for k, v in opts:
        p.set_option(k, v)
    return p


def prof(plan, xp, yp, batch, reps=3):
    acc = {}
    for _ in range(reps):
        for name, ms, cnt in plan.profile_batch_ptr(xp, yp, batch, 0, st):
            if cnt:
                acc.setdefault(name, []).append(ms)
    return {k: round(statistics.median(v), 3) for k, v in acc.items()}


def ev_ms(fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return round(statistics.median(ts), 3)


def emit(**kw):
    print(json.dumps(dict(kind=KIND, proc=TAG, pid=os.getpid(), **kw)), flush=True)


def main():
    global st
    st = torch.cuda.current_stream().cuda_stream
    base = make()
    arms = [("default", base), ("walk2", make([("tile_walk", 2)])), ("walk8", make([("tile_walk", 8)])), ("swz0", make([("xcd_swizzle", 0)])), ("swz2", make([("xcd_swizzle", 2)]))]
    emit(tag="start", free_gb=round(torch.cuda.mem_get_info()[0] / 2**30, 2), plan=base.describe())

    def alloc(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    fdt = torch.float32 if REAL == "f32" else torch.float64
    scenarios = [("x_then_y", 0), ("y_then_x_filler1027m", 1027), ("y_then_x_filler40g", 40000), ("y_then_x_filler1027m_again", 1027), ("x_then_y_filler20g", 20000), ("y_then_x", 0)]
    for si, (name, filler_mb) in enumerate(scenarios):
        fillt = alloc(filler_mb << 20) if filler_mb else None
        if name.startswith("y_then_x"):
            Y = alloc(BYTES); X = alloc(BYTES)
        else:
            X = alloc(BYTES); Y = alloc(BYTES)
        X.view(fdt).uniform_(0, 1)
        xp, yp = X.data_ptr(), Y.data_ptr()
        whole = prof(base, xp, yp, BATCH)
        swapped = prof(base, yp, xp, BATCH)  # y -> x: which of the two allocations is the slow one to WRITE
        X.view(fdt).uniform_(0, 1)
        cb = CH * N * ESZ
        per = []
        for j in range(NCH):
            o = j * cb
            k = prof(base, xp + o, yp + o, CH, reps=3)
            fill_ms = ev_ms(lambda: Y[o:o + cb].view(fdt).fill_(1.0))
            copy_ms = ev_ms(lambda: Y[o:o + cb].copy_(X[o:o + cb]))
            fillx_ms = ev_ms(lambda: X[o:o + cb].view(fdt).fill_(1.0))
            per.append(dict(chunk=j, pass0=k.get("pass0"), pass1=k.get("pass1"), fill_y_ms=fill_ms, copy_x_to_y_ms=copy_ms, fill_x_ms=fillx_ms))
        X.view(fdt).uniform_(0, 1)
        # x chunk 0 -> every y chunk (isolates y), and every x chunk -> y chunk 0 (isolates x)
        iso_y = [prof(base, xp, yp + j * cb, CH, reps=3) for j in range(NCH)]
        iso_x = [prof(base, xp + j * cb, yp, CH, reps=3) for j in range(NCH)]
        slow = max(range(NCH), key=lambda j: per[j]["pass1"])
        fast = min(range(NCH), key=lambda j: per[j]["pass1"])
        arms_slow = {an: prof(p, xp + slow * cb, yp + slow * cb, CH) for an, p in arms}
        arms_fast = {an: prof(p, xp + fast * cb, yp + fast * cb, CH) for an, p in arms}
        emit(tag="alloc", scenario=name, idx=si, x_ptr=hex(xp), y_ptr=hex(yp), whole=whole, swapped_y_to_x=swapped, per_chunk=per,
             x0_to_ychunk=[k.get("pass1") for k in iso_y], x0_to_ychunk_pass0=[k.get("pass0") for k in iso_y],
             xchunk_to_y0_pass0=[k.get("pass0") for k in iso_x], slow_chunk=slow, fast_chunk=fast, arms_on_slow_chunk=arms_slow, arms_on_fast_chunk=arms_fast)
        del X, Y, fillt
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
