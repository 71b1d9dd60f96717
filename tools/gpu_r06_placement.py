#!/usr/bin/env python3
"""Development tool (round 6): what separates the two modes of the f64 1024 x 1024 last pass (21.9 / 24.4 ms per
4096 transforms, both seen in ONE session of round 5) and of the f32 2048 x 2048 last pass -- and which tile order is immune.

One process = several FRESH allocations of the workload's buffers (both orders, behind fillers, after a freed C2-sized pair: what
`bench.py --config c3` and bench.py's quick_config produce); per allocation the passes under several tile orders, then virtual offsets
of the output inside an over-allocated buffer.  HIP events per kernel (fourier_hip_profile_*), median of REPS.  One JSON line per point.
usage: gpu_r06_placement.py KIND [tag]     KIND = c3 (f64 2^20 x 4096) | c5 (f32 2^22 x 1024) | c2 (f32 2^20 x 4096)"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

KIND = sys.argv[1] if len(sys.argv) > 1 else "c3"
TAG = sys.argv[2] if len(sys.argv) > 2 else ""
N, BATCH, REAL, ESZ = {"c3": (1 << 20, 4096, "f64", 16), "c5": (1 << 22, 1024, "f32", 8), "c2": (1 << 20, 4096, "f32", 8)}[KIND]
BATCH = int(os.environ.get("PLACEMENT_BATCH", BATCH))
BYTES = N * BATCH * ESZ
REPS = int(os.environ.get("PLACEMENT_REPS", "3"))
FULL = os.environ.get("PLACEMENT_FULL", "1") != "0"
st = None

# arms: plan options; the time that matters is pass1's
ARMS = [("default", []), ("walk2", [("tile_walk", 2)]), ("walk4", [("tile_walk", 4)]), ("walk8", [("tile_walk", 8)]), ("walk16", [("tile_walk", 16)]),
        ("walk8_tf", [("tile_walk", 8 | 1 << 19)]), ("walk8_g8", [("tile_walk", 8 | 8 << 8)]), ("walk8_strided", [("tile_walk", 8 | 1 << 20)]),
        ("swz0_noremap", [("xcd_swizzle", 0)]), ("swz1_plain", [("xcd_swizzle", 1)]), ("swz2_tf_rr", [("xcd_swizzle", 2)]), ("swz3_sliced", [("xcd_swizzle", 3)]),
        ("swz4_bandmajor", [("xcd_swizzle", 4)])]


def make(opts):
    p = (F.create_fft_f32 if REAL == "f32" else F.create_fft_f64)(N, 0)
    for k, v in opts:
        p.set_option(k, v)
    return p


def med(plan, xp, yp):
    acc = {}
    for _ in range(REPS):
        for name, ms, cnt in plan.profile_batch_ptr(xp, yp, BATCH, 0, st):
            if cnt:
                acc.setdefault(name, []).append(ms)
    return {k: round(statistics.median(v), 3) for k, v in acc.items()}


def emit(**kw):
    print(json.dumps(dict(kind=KIND, proc=TAG, pid=os.getpid(), **kw)), flush=True)


def main():
    global st
    st = torch.cuda.current_stream().cuda_stream
    plans = [(n, make(o)) for n, o in ARMS]
    base = plans[0][1]
    free0, total = torch.cuda.mem_get_info()
    emit(tag="start", free_gb=round(free0 / 2**30, 2), total_gb=round(total / 2**30, 2), plan=base.describe())

    def alloc(nbytes):
        return torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def fill(t):
        t[: BYTES].view(torch.float32 if REAL == "f32" else torch.float64).uniform_(0, 1)

    scenarios = [("x_then_y", 0), ("y_then_x", 0), ("x_then_y_filler3m", 3), ("y_then_x_filler1027m", 1027), ("after_freed_c2_pair", -1), ("x_then_y_again", 0),
                 ("y_then_x_filler40g", 40000), ("x_then_y_filler5m", 5)]
    for si, (name, filler_mb) in enumerate(scenarios):
        fillt = None
        if filler_mb < 0:  # what quick_config sees: a 32 GiB + 32 GiB pair was alive, used, and has just been freed
            a, b = alloc(32 << 30), alloc(32 << 30)
            a.view(torch.float32).uniform_(0, 1)
            b.copy_(a)
            torch.cuda.synchronize()
            del a, b
            torch.cuda.empty_cache()
        elif filler_mb:
            fillt = alloc(filler_mb << 20)
        if name.startswith("y_then_x"):
            Y = alloc(BYTES); X = alloc(BYTES)
        else:
            X = alloc(BYTES); Y = alloc(BYTES)
        fill(X)
        xp, yp = X.data_ptr(), Y.data_ptr()
        row = dict(tag="fresh_alloc", scenario=name, idx=si, x_ptr=hex(xp), y_ptr=hex(yp), x_mod_1g=xp % (1 << 30), y_mod_1g=yp % (1 << 30), y_minus_x=yp - xp)
        arms = {}
        for an, p in (plans if FULL else plans[:1]):
            arms[an] = med(p, xp, yp)
        row["arms"] = arms
        # in place on X (x -> x): pass 0 goes to the plan's scratch, pass 1 scratch -> x (out of place)
        emit(**row)
        del X, Y, fillt
        torch.cuda.empty_cache()

    # virtual offsets of the output (and input) inside over-allocated buffers
    PAD = 1 << 30
    X = alloc(BYTES + PAD); Y = alloc(BYTES + PAD)
    fill(X)
    xp0, yp0 = X.data_ptr(), Y.data_ptr()
    K, M = 1 << 10, 1 << 20
    offs = [0, 4 * K, 16 * K, 64 * K, 256 * K, M, 2 * M, 4 * M, 6 * M, 8 * M, 10 * M, 16 * M, 18 * M, 32 * M, 64 * M, 256 * M, 512 * M, 1024 * M - 2 * M]
    for rep in range(2):
        for d in offs:
            emit(tag="out_offset", dy=d, dx=0, sweep=rep, default=med(base, xp0, yp0 + d), walk8=med(plans[3][1], xp0, yp0 + d))
        for d in offs[1:10]:
            emit(tag="in_offset", dy=0, dx=d, sweep=rep, default=med(base, xp0 + d, yp0))
    # out-of-place last pass: in-place call on Y (pass 0 -> plan scratch, pass 1 scratch -> Y) -- is it the in-place-ness?
    try:
        Y[: BYTES].copy_(X[: BYTES])
        emit(tag="in_place_call", default=med(base, yp0, yp0))
    except Exception as e:  # the scratch may not fit
        emit(tag="in_place_call", error=repr(e))


if __name__ == "__main__":
    main()
