import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
import fourier_amd as fa
from oracle import oracle as O
from helpers import hash_normal, rel_l2
O.build()
for n, dtype in ((999983, np.complex128), (65537, np.complex128), (10007, np.complex128), (1013, np.complex128), (250007, np.complex128), (999983, np.complex64), (10007, np.complex64)):
    x = np.stack([hash_normal(77 + b, n) for b in range(2)]).astype(dtype)
    truth = torch.fft.fft(torch.from_numpy(x).to(torch.complex128)).numpy()
    ref = O.transform_batch(x, 0)
    row = dict(n=n, dtype=np.dtype(dtype).name, oracle_vs_truth=rel_l2(ref, truth))
    for opt in (0, 1):
        plan = fa.create_fft_f32(n) if dtype == np.complex64 else fa.create_fft_f64(n)
        plan.set_option("bluestein_reference_chirp", opt)
        d = torch.from_numpy(x).cuda(); o = torch.empty_like(d)
        plan.transform(d, o, fa.Transform.Fft); torch.cuda.synchronize()
        got = o.cpu().numpy()
        row[f"opt{opt}_vs_oracle"] = rel_l2(got, ref); row[f"opt{opt}_vs_truth"] = rel_l2(got, truth)
        inv = torch.empty_like(d); plan.transform(o, inv, fa.Transform.Ifft); torch.cuda.synchronize()
        row[f"opt{opt}_roundtrip"] = rel_l2(inv.cpu().numpy(), x)
    print(json.dumps(row), flush=True)
