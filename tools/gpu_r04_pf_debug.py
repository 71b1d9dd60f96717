#!/usr/bin/env python3
"""Development tool: where does the prefetching last pass differ from the plain one?  (2^22 = 2048 x 2048, one transform)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

for n, L2, batch in ((1 << 22, 2048, 3), (1 << 21, 1024, 3)):
    s = n // L2
    Q = L2 // 16
    x = torch.randn((batch, n), dtype=torch.complex64, device="cuda")
    a, b = torch.empty_like(x), torch.empty_like(x)
    on, off = F.create_fft_f32(n, 0), F.create_fft_f32(n, 0)
    on.set_option("last_pass_prefetch", 1); off.set_option("last_pass_prefetch", 0)
    on.transform(x, a, F.Transform.Fft); off.transform(x, b, F.Transform.Fft)
    torch.cuda.synchronize()
    bad = (torch.view_as_real(a) != torch.view_as_real(b)).any(dim=-1)  # (batch, n)
    print(n, "mismatch fraction", float(bad.float().mean()), "nan in a", bool(torch.isnan(torch.view_as_real(a)).any()))
    m = bad.view(batch, L2, s)  # [b][k][j]
    k_rate = m.float().mean(dim=(0, 2))  # per output row k = th + Q*r
    print(" per r:", [round(float(k_rate.view(16, Q)[r].mean()), 3) for r in range(16)])
    print(" per th (first 16):", [round(float(k_rate.view(16, Q)[:, th].mean()), 3) for th in range(16)])
    j_rate = m.float().mean(dim=(0, 1))  # per column j
    tiles = j_rate.view(-1, 16).mean(dim=1)
    print(" per tile (first 16):", [round(float(t), 3) for t in tiles[:16]], " ... tiles with any mismatch:", int((tiles > 0).sum()), "of", tiles.numel())
    print(" per column in tile:", [round(float(v), 3) for v in j_rate.view(-1, 16).mean(dim=0)])
    print(" per transform:", [round(float(v), 3) for v in m.float().mean(dim=(1, 2))])
    rel = float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
    print(" rel l2 diff", rel)
