import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import fourier_amd as fa
from fourier_amd import fft as F
for n, real in ((1<<14,"f32"),(1<<15,"f32"),(1<<14,"f64"),(1<<13,"f32")):
    cdt = torch.complex64 if real=="f32" else torch.complex128
    torch.manual_seed(0)
    x = torch.randn(4, n, dtype=cdt, device="cuda"); y = torch.zeros_like(x)
    plan = (F.create_fft_f32 if real=="f32" else F.create_fft_f64)(n, 0)
    plan.transform(x, y, F.Transform.Fft); torch.cuda.synchronize()
    ref = torch.fft.fft(x.to(torch.complex128))
    d = (y.to(torch.complex128)-ref).abs()
    tol = 1e-3 if real=="f32" else 1e-9
    bad = (d > tol*ref.abs().max())
    print(n, real, plan.describe(), "rel", float(torch.linalg.norm(y.to(torch.complex128)-ref)/torch.linalg.norm(ref)), "bad per transform", bad.sum(1).tolist())
    if bad.any():
        idx = bad[0].nonzero().flatten().cpu().numpy()
        print("  first bad idx", idx[:40], "count", len(idx), " idx%128 set", sorted(set((idx%128).tolist()))[:20], " idx//128 set", sorted(set((idx//128).tolist()))[:20])
        # which input elements matter? inverse-transform the error to locate an input-side fault
        e = torch.fft.ifft((y.to(torch.complex128)-ref)[0])
        top = torch.topk(e.abs(), 16).indices.cpu().numpy()
        print("  error back-projected to input idx", sorted(top.tolist()), float(e.abs().max()))
