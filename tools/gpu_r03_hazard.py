#!/usr/bin/env python3
"""Round 3: reproduces (variant library `tl_store_soff`) / rules out (product) the store-data corruption seen with
`buffer_store_dwordx4 ... sN offen` (scalar offset register) when the data registers are rewritten right after the store.
Runs the one-launch plans 2^14 / 2^15 repeatedly on fixed inputs and counts outputs that differ from torch's FFT."""
import ctypes, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F, _lib

libs = [("product", None)]
p = os.path.join(ROOT, "fourier_amd", "lib", "variants", "libfourier_tl_store_soff.so")
if os.path.exists(p):
    libs.append(("tl_store_soff", p))
base = _lib.lib()
for name, path in libs:
    _lib._lib = base if path is None else _lib.bind(ctypes.CDLL(path))
    for n in (1 << 12, 1 << 13, 1 << 14, 1 << 15):
        batch = (1 << 26) // n
        torch.manual_seed(n)
        x = torch.randn(batch, n, dtype=torch.complex64, device="cuda"); y = torch.empty_like(x)
        ref = torch.fft.fft(x)
        plan = F.create_fft_f32(n, 0)
        bad_runs = bad_elems = 0
        lanes = {}
        for rep in range(20):
            y.zero_()
            plan.transform(x, y, F.Transform.Fft); torch.cuda.synchronize()
            bad = (y - ref).abs() > 1e-3 * ref.abs().max()
            nb = int(bad.sum())
            bad_runs += nb > 0; bad_elems += nb
            if nb and len(lanes) < 64:
                cols = (bad.nonzero()[:, 1] % 32).unique().tolist()
                for c in cols: lanes[c] = lanes.get(c, 0) + 1
        print(json.dumps(dict(lib=name, n=n, batch=batch, plan=plan.describe(), runs=20, runs_with_wrong_outputs=bad_runs,
                              wrong_outputs=bad_elems, wrong_output_index_mod_32=sorted(lanes))), flush=True)
        del plan, x, y, ref; torch.cuda.empty_cache()
_lib._lib = base
