#!/usr/bin/env python3
"""Development microbenchmarks: HBM / Infinity-Cache bandwidth for the FFT passes' access patterns."""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "libmembench.so")


def build():
    src = os.path.join(ROOT, "tools", "membench.hip")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", SO])
    return SO


def main():
    import torch

    L = ctypes.CDLL(build())
    vp, u64, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int
    L.mb_lin.argtypes = [ci, ci, vp, vp, u64, ci, ci, vp]
    L.mb_slab.argtypes = [ci, vp, vp, u64, u64, vp]
    L.mb_tile.argtypes = [vp, vp, u64, ci, vp]
    out = open(os.path.join(ROOT, "gpurun_out", "membench.jsonl"), "a")

    def emit(**kw):
        s = json.dumps(kw)
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    dev = torch.device("cuda", 0)
    big = 16 << 30
    a = torch.empty(big, dtype=torch.uint8, device=dev)
    b = torch.empty(big, dtype=torch.uint8, device=dev)
    a.view(torch.float32).uniform_(0, 1)
    st = torch.cuda.current_stream().cuda_stream

    def timeit(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    L.mb_fused_model.argtypes = [vp, vp, vp, u64, u64, ci, ci, vp]
    L.mb_slab_x.argtypes = [vp, vp, u64, u64, ci, vp]
    L.mb_tile_x.argtypes = [vp, vp, u64, ci, vp]
    L.mb_fused_sync.argtypes = [vp, vp, vp, vp, u64, ci, ci, ci, vp]
    L.mb_fused_pipe.argtypes = [vp, vp, vp, vp, u64, ci, ci, ci, ci, vp]
    L.mb_tile_s.argtypes = [vp, vp, u64, u64, u64, ci, vp]
    L.mb_l2x.argtypes = [vp, vp, vp, vp, u64, u64, ci, ci, ci, ci, ci, ci, vp]
    if "--l2x" in sys.argv:
        # XCD-local exchange model (round 2): does an intermediate that round-trips through the XCD's own L2 cost HBM time?
        S = torch.empty(8 * (64 << 20), dtype=torch.uint8, device=dev)
        xcc = torch.zeros(8 * 64 * 2, dtype=torch.int32, device=dev)
        wpx = 64  # 2 workgroups per CU
        t = timeit(lambda: L.mb_l2x(a.data_ptr(), b.data_ptr(), S.data_ptr(), xcc.data_ptr(), big, 1 << 20, 0, 0, 0, 0, wpx, 69632, st), reps=3, warm=1)
        xs = xcc[:8 * wpx].cpu().numpy()
        import numpy as _np
        emit(tag="l2x", mode="copy_only", wgs_per_xcd=wpx, ms=round(t * 1e3, 3), alg_tbps=round(2 * big / t / 1e12, 3),
             xcc_mismatch=int((xs != (_np.arange(len(xs)) % 8)).sum()))
        for fp_kib in (256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 65536):
            for mode, mname in ((1, "copy+ring_write"), (2, "copy+ring_write+ring_read")):
                for sfl, lfl, xshift in ((0, 0, 0), (0, 1, 0), (2, 1, 0), (1, 1, 0), (0, 1, 3), (1, 1, 3)):
                    if mode == 1 and (lfl, xshift) != (0, 0) and not (sfl, lfl, xshift) in ((1, 1, 0), (2, 1, 0)):
                        continue
                    t = timeit(lambda: L.mb_l2x(a.data_ptr(), b.data_ptr(), S.data_ptr(), xcc.data_ptr(), big, fp_kib << 10, mode, sfl, lfl, xshift, wpx, 69632, st),
                               reps=3, warm=1)
                    emit(tag="l2x", mode=mname, ring_kib_per_xcd=fp_kib, store=("plain", "sc1", "nt")[sfl], load=("plain", "sc1", "nt")[lfl],
                         read_from_xcd_plus=xshift, ms=round(t * 1e3, 3), alg_tbps=round(2 * big / t / 1e12, 3), us_per_8MiB=round(t / (big / (8 << 20)) * 1e6, 3))
        return
    L.mb_regx.argtypes = [vp, vp, vp, vp, u64, ci, ci, vp]
    if "--regx" in sys.argv:
        # register-resident 2^20 model: one transform per XCD held in the registers of 64 workgroups, all-to-all through
        # a small double-buffered L2 window in `rounds` rounds with one XCD-wide barrier each, no arithmetic
        S = torch.empty(8 * (16 << 20), dtype=torch.uint8, device=dev)  # rounds = 1 needs 2 x 8 MiB per XCD
        ctrl = torch.zeros(1024, dtype=torch.int32, device=dev)
        for rounds in (1, 2, 4, 8, 16):
            t = timeit(lambda: L.mb_regx(a.data_ptr(), b.data_ptr(), S.data_ptr(), ctrl.data_ptr(), big, rounds, 69632, st), reps=3, warm=1)
            torch.cuda.synchronize()
            tick = ctrl.cpu().numpy()
            emit(tag="regx", rounds=rounds, window_kib_per_xcd=2 * 8192 // rounds, ms=round(t * 1e3, 3), alg_tbps=round(2 * big / t / 1e12, 3),
                 us_per_transform=round(t / (big / (8 << 20)) * 1e6, 3), abort_flag=int(tick[1023]), wgs_per_xcd=[int(tick[x * 32]) for x in range(8)])
        return
    if "--stride" in sys.argv:
        nt = 1024  # transforms of 8 MiB payload
        for sru, dru in ((512, 512), (520, 512), (512, 520), (520, 520), (528, 528), (576, 576), (1024, 1024), (1032, 1032), (2048, 2048), (640, 640)):
            t = timeit(lambda: L.mb_tile_s(a.data_ptr(), b.data_ptr(), nt, sru, dru, 1, st))
            emit(tag="tile_stride", src_row_bytes=sru * 16, dst_row_bytes=dru * 16, ms=round(t * 1e3, 3), payload_tbps=round(2 * nt * (8 << 20) / t / 1e12, 3))
        return
    if "--pipe" in sys.argv:
        S = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        ctrs = torch.zeros(8192 + 64 * 2 * 2048 + 1024, dtype=torch.int32, device=dev)
        nt = big // (8 << 20)
        for sc1 in (1, 1, 1 + 16, 0):
            for tpx, wpt in ((1, 64), (2, 32), (4, 16), (1, 32)):
                b.zero_()
                t = timeit(lambda: L.mb_fused_pipe(a.data_ptr(), b.data_ptr(), S.data_ptr(), ctrs.data_ptr(), big, tpx, wpt, 69632, sc1, st), reps=3, warm=1)
                torch.cuda.synchronize()
                flag = int(ctrs[4096].item())
                bad = 0
                for t0 in range(0, nt, 64):
                    xa = a[t0 * (8 << 20):(t0 + 64) * (8 << 20)].view(torch.int32).view(64, 1024, 512, 4)
                    xb = b[t0 * (8 << 20):(t0 + 64) * (8 << 20)].view(torch.int32).view(64, 512, 1024, 4)
                    bad += int((xb != xa.transpose(1, 2)).any(dim=-1).sum().item())
                xcc = ctrs[4200:4200 + 8 * tpx * wpt].cpu().numpy()
                import numpy as _np
                mism = int((xcc != (_np.arange(len(xcc)) % 8)).sum())
                emit(tag="fused_pipe", xcc_mismatch=mism, xcc_first16=[int(v) for v in xcc[:16]], sc1=sc1, teams_per_xcd=tpx, wgs_per_team=wpt, scratch_mib=8 * tpx * 16, ms=round(t * 1e3, 3),
                     alg_tbps=round(2 * big / t / 1e12, 3), us_per_transform=round(t / nt * 1e6, 3), abort_flag=flag, wrong_units=bad)
        return
    if "--sync" in sys.argv:
        S = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        ctrs = torch.zeros(8192, dtype=torch.int32, device=dev)
        b.zero_()
        for tpx, wpt in ((2, 32), (1, 64), (4, 16), (2, 16), (1, 32)):
            for lds in (69632,):
                t = timeit(lambda: L.mb_fused_sync(a.data_ptr(), b.data_ptr(), S.data_ptr(), ctrs.data_ptr(), big, tpx, wpt, lds, st), reps=3, warm=1)
                torch.cuda.synchronize()
                flag = int(ctrs[4096].item())
                emit(tag="fused_sync", teams_per_xcd=tpx, wgs_per_team=wpt, ms=round(t * 1e3, 3), alg_tbps=round(2 * big / t / 1e12, 3),
                     us_per_transform=round(t / (big / (8 << 20)) * 1e6, 3), abort_flag=flag)
        # data check of the last run (all transforms): B[u*1024 + row] == A[row*512 + u] in 16-byte units
        nt = big // (8 << 20)
        bad = 0
        for t0 in range(0, nt, 64):
            xa = a[t0 * (8 << 20):(t0 + 64) * (8 << 20)].view(torch.int32).view(64, 1024, 512, 4)
            xb = b[t0 * (8 << 20):(t0 + 64) * (8 << 20)].view(torch.int32).view(64, 512, 1024, 4)
            bad += int((xb != xa.transpose(1, 2)).any(dim=-1).sum().item())
        emit(tag="fused_sync_check", wrong_units=bad, transforms=nt)
        return
    if "--xcd" in sys.argv:
        for swz in (0, 1):
            for slab in (64 << 10, 128 << 10, 512 << 10, 2 << 20):
                t = timeit(lambda: L.mb_slab_x(a.data_ptr(), b.data_ptr(), big, slab, swz, st))
                emit(tag="slab_x", swz=swz, slab_kib=slab >> 10, ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
            t = timeit(lambda: L.mb_tile_x(a.data_ptr(), b.data_ptr(), big, swz, st))
            emit(tag="tile_x", swz=swz, ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
            t = timeit(lambda: L.mb_tile_x(a.data_ptr(), a.data_ptr(), big, swz, st))
            emit(tag="tile_x_inplace", swz=swz, ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
        return
    if "--fused-only" in sys.argv:
        ring = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
        for blocks in (256, 512, 1024):
            for mode, mname in ((0, "copy_only"), (1, "copy+ring_write"), (2, "copy+ring_write+ring_read")):
                for ring_mib in (16, 32, 64, 128, 192, 256, 1024, 4096):
                    if mode == 0 and ring_mib != 16:
                        continue
                    t = timeit(lambda: L.mb_fused_model(a.data_ptr(), b.data_ptr(), ring.data_ptr(), big, ring_mib << 20, mode, blocks, st), reps=3, warm=1)
                    emit(tag="fused_model", mode=mname, blocks=blocks, ring_mib=ring_mib, ms=round(t * 1e3, 3),
                         alg_tbps=round(2 * big / t / 1e12, 3), us_per_8MiB=round(t / (big / (8 << 20)) * 1e6, 3))
        return
    # 1. linear streaming, HBM-sized
    for mode, mname in ((0, "copy"), (1, "read"), (2, "write")):
        for U in (4, 8, 16):
            for blocks in (2048, 4096, 8192, 65536):
                t = timeit(lambda: L.mb_lin(mode, U, a.data_ptr(), b.data_ptr(), big, blocks, 1, st))
                traffic = big * (2 if mode == 0 else 1)
                emit(tag=f"lin_{mname}", U=U, blocks=blocks, ms=round(t * 1e3, 3), tbps=round(traffic / t / 1e12, 3))
    # 2. slab (one contiguous slab per block)
    for U in (4, 8, 16):
        for slab in (64 << 10, 128 << 10, 1 << 20):
            t = timeit(lambda: L.mb_slab(U, a.data_ptr(), b.data_ptr(), big, slab, st))
            emit(tag="slab_copy", U=U, slab_kib=slab >> 10, ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
    # 3. column-tile pattern of the FFT passes (no compute)
    for tr in (0, 1):
        t = timeit(lambda: L.mb_tile(a.data_ptr(), b.data_ptr(), big, tr, st))
        emit(tag="tile_copy", transposed=tr, ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
        t = timeit(lambda: L.mb_tile(a.data_ptr(), a.data_ptr(), big, 0, st)) if tr == 0 else None
        if t:
            emit(tag="tile_copy_inplace", ms=round(t * 1e3, 3), tbps=round(2 * big / t / 1e12, 3))
    # 4. footprint sweep: the same copy repeated inside one launch (persistent) -> cache-resident bandwidth
    for fp_mib in (8, 16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096):
        fp = fp_mib << 20
        iters = max(2, min(400, (32 << 30) // fp))
        for mode, mname in ((0, "copy"), (1, "read"), (2, "write")):
            half = fp // 2 if mode == 0 else fp
            t = timeit(lambda: L.mb_lin(mode, 8, a.data_ptr(), b.data_ptr(), half, 2048, iters, st), reps=2, warm=1)
            traffic = half * iters * (2 if mode == 0 else 1)
            emit(tag=f"footprint_{mname}", footprint_mib=fp_mib, iters=iters, ms=round(t * 1e3, 3), tbps=round(traffic / t / 1e12, 3))
    # 5. write-then-read across launches: is freshly written data served from the Infinity Cache?
    for fp_mib in (16, 32, 64, 128, 192, 256, 512, 2048):
        fp = fp_mib << 20
        def wr():
            L.mb_lin(2, 8, a.data_ptr(), b.data_ptr(), fp, 2048, 1, st)   # write b[0:fp]
        def rd():
            L.mb_lin(1, 8, b.data_ptr(), a.data_ptr(), fp, 2048, 1, st)   # read b[0:fp]
        tw = timeit(wr, reps=20, warm=2)
        trd = timeit(rd, reps=20, warm=2)
        both = timeit(lambda: (wr(), rd()), reps=20, warm=2)
        emit(tag="write_then_read", footprint_mib=fp_mib, write_us=round(tw * 1e6, 1), read_us=round(trd * 1e6, 1),
             pair_us=round(both * 1e6, 1), pair_tbps=round(2 * fp / both / 1e12, 3))


if __name__ == "__main__":
    main()
