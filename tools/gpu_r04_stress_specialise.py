#!/usr/bin/env python3
"""Round 4 stress of the run-time specialisation (plan option "specialise", rtc.cpp): random lengths whose prime factors stop at
13 and that have NO ahead-of-time per-length kernel -- whole-transform kernels up to a compute unit's LDS, column-tile passes
beyond it -- compiled with hipRTC, random batch / code / placement, f32 and f64, against the oracle.  One line per failure, a
summary at the end (cases, failures, compile seconds, worst error per route)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import fourier_amd as fa
from oracle import oracle as O

O.build()
rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "4040")))
def smooth(limit, primes):
    vals = {1}
    for p in primes:
        vals = {v * p ** e for v in vals for e in range(0, 24) if v * p ** e <= limit}
    return sorted(vals)
small = [v for v in smooth(20480, [2, 3, 5, 7, 11, 13]) if any(v % p == 0 for p in (7, 11, 13)) and v > 16]
large = [v for v in smooth(4_000_000, [2, 3, 5, 7, 11, 13]) if v > 20480 and any(v % p == 0 for p in (5, 7, 11, 13))]
pick = list(rng.choice(small, int(os.environ.get("STRESS_SMALL", "110")), replace=False)) + list(rng.choice(large, int(os.environ.get("STRESS_LARGE", "40")), replace=False))
worst, fails, count, refused, t_compile = {}, 0, 0, 0, 0.0
t0 = time.time()
for n in pick:
    n = int(n)
    for dtype, tol in ((np.complex64, 2e-6), (np.complex128, 1e-9 if n > 100000 else 5e-11)):
        if rng.random() < 0.35:
            continue
        plan = fa.create_fft_f32(n) if dtype == np.complex64 else fa.create_fft_f64(n)
        before = plan.describe()
        t1 = time.time()
        try:
            plan.set_option("specialise", 1)
        except fa.FourierError:
            refused += 1  # no tile factorisation (a large length), or the family does not apply: the plan keeps its route
        t_compile += time.time() - t1
        batch = int(rng.integers(1, 4)) if n > 4096 else int(rng.integers(1, 70))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(dtype)
        code = int(rng.integers(0, 5))
        inplace = bool(rng.integers(0, 2))
        d = torch.from_numpy(x).cuda()
        o = d if inplace else torch.empty_like(d)
        plan.transform(d, o, fa.Transform(code))
        torch.cuda.synchronize()
        got = o.cpu().numpy()
        ref = O.transform_batch(x, code)
        err = float(np.linalg.norm(got.astype(np.complex128) - ref) / max(np.linalg.norm(ref), 1e-300))
        after = plan.describe()
        fam = ("specialised " if "specialised" in after else "unchanged ") + after.split()[1] + " " + dtype.__name__
        worst[fam] = max(worst.get(fam, 0.0), err)
        count += 1
        if not (err <= tol):
            fails += 1
            print(json.dumps(dict(FAIL=True, n=n, dtype=dtype.__name__, batch=batch, code=code, inplace=inplace, before=before, plan=after, rel_l2=err)), flush=True)
        del plan, d, o
print(json.dumps(dict(cases=count, failures=fails, refused=refused, seconds=round(time.time() - t0, 1), set_option_seconds=round(t_compile, 1),
                      worst_rel_l2_by_route=dict(sorted(worst.items())))))
