#!/usr/bin/env python3
"""BASELINE config C5: batched 1D c2c f32, N=2^22, batch=65536 (2 TiB of input), batch-sharded across the
ranks of one node with no data-path collective.  The input does not fit anywhere at once, so every rank
walks its contiguous shard (fourier_amd.shard.batch_shard) in fixed chunks, regenerating each chunk on the
device (seeded by the chunk's first transform index) and transforming it (out of place by default); the first
transform of a few chunks is checked against the CPU oracle.  Launch: `python tests/harness/run_c5.py` (1 GPU) or
`python -m torch.distributed.run --nproc-per-node N tests/harness/run_c5.py`.  Prints one JSON line on rank 0."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=22)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--chunk", type=int, default=1024)
    ap.add_argument("--check-chunks", type=int, default=2)
    ap.add_argument("--inplace", action="store_true", help="transform each chunk in place (routes the intermediate through the plan's scratch)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import fourier_amd
    from fourier_amd import Transform, shard

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n = 1 << args.log2n
    lo, hi = shard.batch_shard(args.batch, world, rank)
    plan = fourier_amd.create_fft_f32(n, local)
    buf = torch.empty((args.chunk, n), dtype=torch.complex64, device=dev)
    res = buf if args.inplace else torch.empty_like(buf)  # out of place: the first pass writes its intermediate into `res`
    gen = torch.Generator(device=dev)
    checks = []
    # untimed warm-up on one chunk: first-touch of the buffers and, in place, the plan's 32 GiB scratch (one-off, like plan creation)
    buf.zero_()
    plan.transform(buf, res, Transform.Fft)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t_fft = 0.0
    t0 = time.perf_counter()
    nchunks = 0
    for b0 in range(lo, hi, args.chunk):
        nb = min(args.chunk, hi - b0)
        gen.manual_seed(0x5EED0C50 + b0)
        torch.view_as_real(buf[:nb]).uniform_(0.0, 1.0, generator=gen)
        keep = buf[0].cpu().numpy() if nchunks < args.check_chunks else None
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        plan.transform(buf[:nb], res[:nb], Transform.Fft)
        torch.cuda.synchronize(dev)
        t_fft += time.perf_counter() - t1
        if keep is not None:
            from oracle import oracle as O

            ref = O.OracleFft(n, np.complex64).transform(keep, O.FFT)
            got = res[0].cpu().numpy()
            checks.append(float(np.linalg.norm(got.astype(np.complex128) - ref) / np.linalg.norm(ref)))
        nchunks += 1
    wall = time.perf_counter() - t0
    t_fft = shard.reduce_max_seconds(t_fft, dist, dev)
    wall = shard.reduce_max_seconds(wall, dist, dev)
    if rank == 0:
        flops = args.batch * 5.0 * n * math.log2(n)
        print(json.dumps({
            "config": f"C5: 1D c2c f32 N=2^{args.log2n} batch={args.batch}, {world} GPU(s), chunks of {args.chunk}, "
                      f"{'in place' if args.inplace else 'out of place'}",
            "plan": plan.describe(), "n_gpus": world, "fft_seconds_max_rank": round(t_fft, 4), "wall_seconds_incl_regen": round(wall, 3),
            "gflops_fft_only": round(flops / t_fft / 1e9, 1), "alg_gbps_fft_only": round(args.batch * 2.0 * n * 8 / t_fft / 1e9, 1),
            "hbm_frac_per_gpu": round(args.batch * 2.0 * n * 8 / t_fft / 8e12 / world, 4),
            "parity_rel_l2_first_transforms": checks, "tolerance": 1e-6}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
