#!/usr/bin/env python3
"""Development tool: does the relative placement of the input and output buffers move the C2 pass times?
One allocation, x at its start, y at (x + 32 GiB + offset) for a few offsets; per-kernel HIP-event times."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

n, batch = 1 << 20, 4096
nbytes = batch * n * 8
plan = F.create_fft_f32(n, 0)
pool = torch.empty(2 * nbytes + (64 << 20), dtype=torch.uint8, device="cuda")
x = pool[:nbytes].view(torch.float32)
x.uniform_(0, 1)
st = torch.cuda.current_stream().cuda_stream
base = pool.data_ptr()
for rep in range(2):
    for off in (0, 256, 1024, 4096, 8192, 16384, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (8 << 20) + 128 * 1024, 32 << 20):
        yp = base + nbytes + off
        for _ in range(2):
            plan.transform_batch_ptr(base, yp, batch, 0, st)
        acc = {}
        for _ in range(3):
            for name, ms, cnt in plan.profile_batch_ptr(base, yp, batch, 0, st):
                if cnt: acc[name] = acc.get(name, 0) + ms / 3
        print(json.dumps({"tag": "offset", "y_minus_x_end": off, **{k: round(v, 3) for k, v in acc.items()}, "total": round(sum(acc.values()), 3)}), flush=True)
# in place (pass 0 into the plan's scratch)
plan.reserve(batch, True)
for _ in range(2):
    plan.transform_batch_ptr(base, base, batch, 0, st)
acc = {}
for _ in range(3):
    for name, ms, cnt in plan.profile_batch_ptr(base, base, batch, 0, st):
        if cnt: acc[name] = acc.get(name, 0) + ms / 3
print(json.dumps({"tag": "inplace", **{k: round(v, 3) for k, v in acc.items()}, "total": round(sum(acc.values()), 3)}), flush=True)
