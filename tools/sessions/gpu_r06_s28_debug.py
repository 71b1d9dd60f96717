#!/usr/bin/env python3
"""Round 6, session 28 (debug): the smooth-M Bluestein route on the GPU at small and large batches, forward error against torch's f64 FFT."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fourier_amd import fft as F
st = torch.cuda.current_stream().cuda_stream
for real, cdt in (("f32", torch.complex64), ("f64", torch.complex128)):
    for n in (20011, 10007 if real == "f64" else 16411):
        for batch in (1, 3, 8, 64, 1000):
            x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1); y = torch.zeros_like(x)
            plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, 0)
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
            ref = torch.fft.fft(x.to(torch.complex128), dim=1)
            errs = ((y.to(torch.complex128) - ref).norm(dim=1) / ref.norm(dim=1)).cpu()
            print(real, n, batch, plan.describe(), "max err %.2e min err %.2e" % (float(errs.max()), float(errs.min())), "bad transforms:", int((errs > 1e-3).sum()), flush=True)
