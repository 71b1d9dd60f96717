#!/usr/bin/env python3
"""Round 3 one-off stress: every 2^a*3^b up to 6e6 (LDS, global-pass and tiled routes), every length up to 20480 whose prime
factors stop at 13 (the radix-5/7/11/13 LDS kernels), random other sizes, random codes,
in and out of place, f32 and f64, against the oracle.  Prints one line per failure and a summary."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import fourier_amd as fa
from oracle import oracle as O

O.build()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "12345")))
smooth = sorted({(2 ** a) * (3 ** b) for a in range(0, 23) for b in range(0, 15) if (2 ** a) * (3 ** b) <= 6_000_000})
def _smooth(limit, primes):
    vals = {1}
    for p in primes:
        vals = {v * p ** e for v in vals for e in range(0, 16) if v * p ** e <= limit}
    return vals
smooth13 = sorted(v for v in _smooth(20480, [2, 3, 5, 7, 11, 13]) if any(v % p == 0 for p in (5, 7, 11, 13)))  # the prime-radix LDS kernels
# round 5: lengths with factors 5 / 7 beyond the LDS kernels (ahead-of-time tile passes): a random 120 of the 7-smooth lengths up to 3e6
smooth7_big = sorted(v for v in _smooth(3_000_000, [2, 3, 5, 7]) if v > 8192 and (v % 5 == 0 or v % 7 == 0))
smooth7_big = sorted({int(v) for v in rng.choice(smooth7_big, size=min(120, len(smooth7_big)), replace=False)})
# round 6: lengths whose Bluestein plan takes a product of two tile lengths for M (just above a power of two: M_pow2 / M >= 1.44 ... 1.6), 60 of them
smooth_m = sorted({int(v) for k in range(13, 17) for v in rng.integers((1 << k) + 1, int(1.27 * (1 << k)), 15)})
others = sorted({int(v) for v in np.concatenate([rng.integers(2, 5000, 150), rng.integers(5000, 200000, 80), rng.integers(200000, 3000000, 25)])})
worst = {}
fails = 0
t0 = time.time()
count = 0
for n in smooth + smooth13 + smooth7_big + smooth_m + others:
    for dtype, tol in ((np.complex64, 2e-6), (np.complex128, 1e-9 if n > 100000 else 5e-11)):
        if dtype == np.complex128 and rng.random() < 0.5:
            continue
        batch = int(rng.integers(1, 4)) if n > 4096 else int(rng.integers(1, 70))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(dtype)
        code = int(rng.integers(0, 5))
        inplace = bool(rng.integers(0, 2))
        plan = fa.create_fft_f32(n) if dtype == np.complex64 else fa.create_fft_f64(n)
        d = torch.from_numpy(x).cuda()
        o = d if inplace else torch.empty_like(d)
        plan.transform(d, o, fa.Transform(code))
        torch.cuda.synchronize()
        got = o.cpu().numpy()
        ref = O.transform_batch(x, code)
        err = float(np.linalg.norm(got.astype(np.complex128) - ref) / max(np.linalg.norm(ref), 1e-300))
        fam = plan.describe().split()[0] + " " + plan.describe().split()[1][:12] + (" smooth M" if "bluestein" in plan.describe() and "mixed tiles" in plan.describe() else "")
        worst[(fam, dtype.__name__)] = max(worst.get((fam, dtype.__name__), 0.0), err)
        count += 1
        if not (err <= tol):
            fails += 1
            print(json.dumps(dict(FAIL=True, n=n, dtype=dtype.__name__, batch=batch, code=code, inplace=inplace, plan=plan.describe(), rel_l2=err)), flush=True)
        del plan, d, o
print(json.dumps(dict(cases=count, failures=fails, seconds=round(time.time() - t0, 1), worst_rel_l2_by_family={f"{k[0]} {k[1]}": v for k, v in sorted(worst.items())})))
