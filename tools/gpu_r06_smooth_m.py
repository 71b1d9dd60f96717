#!/usr/bin/env python3
"""Development tool (round 6): Bluestein on a smooth M = L1 x L2 (default) against the reference's next power of two
(plan option bluestein_smooth_m = 0), alternating on shared buffers: median ms of 7, fraction of the 8 TB/s HBM peak on the algorithmic
bytes, rel-L2 error against torch's f64 FFT (forward) and of the scaled round trip, per-kernel times of one profiled call.
usage: gpu_r06_smooth_m.py [N ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fourier_amd import fft as F

SIZES = [int(v) for v in sys.argv[1:]] or [8209, 10007, 12289, 14000 + 9, 16381, 16411, 20011, 24001, 28001, 32003, 32771, 40001, 48017, 56003, 65537, 70001, 80021,
                                            90001, 100003, 120011, 131071, 18000 + 13 * 17]
REPS = 7


def main():
    st = torch.cuda.current_stream().cuda_stream
    for real, cdt, esz in (("f32", torch.complex64, 8), ("f64", torch.complex128, 16)):
        for n in SIZES:
            batch = max(1, (1 << 29) // (n * esz))
            x = torch.empty((batch, n), dtype=cdt, device="cuda"); torch.view_as_real(x).uniform_(-1, 1); y = torch.empty_like(x); z = torch.empty_like(x)
            ref = torch.fft.fft(x[:32].to(torch.complex128), dim=1)
            make = F.create_fft_f32 if real == "f32" else F.create_fft_f64
            plans = [("smooth", make(n, 0), []), ("pow2", make(n, 0), [])]
            plans[0][1].set_option("bluestein_smooth_m", 2 if os.environ.get("SMOOTH_FORCE") else 1)  # 2: wherever a product of two tile lengths exists
            plans[1][1].set_option("bluestein_smooth_m", 0)
            if plans[0][1].describe() == plans[1][1].describe():
                print(json.dumps(dict(real=real, n=n, plan=plans[0][1].describe(), note="same route")), flush=True)
                continue
            info = {}
            for name, plan, ts in plans:
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(F.Transform.Fft), st); torch.cuda.synchronize()
                err = float((y[:32].to(torch.complex128) - ref).norm() / ref.norm())
                plan.transform_batch_ptr(y.data_ptr(), z.data_ptr(), batch, int(F.Transform.Ifft), st); torch.cuda.synchronize()
                rt = float((z[:32].to(torch.complex128) - x[:32].to(torch.complex128)).norm() / x[:32].to(torch.complex128).norm())
                info[name] = (err, rt)
            for _ in range(REPS):
                for name, plan, ts in plans:
                    t0 = time.perf_counter(); plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st); torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
            for name, plan, ts in plans:
                t = sorted(ts)[len(ts) // 2]
                prof = plan.profile_batch_ptr(x.data_ptr(), y.data_ptr(), batch, 0, st)
                print(json.dumps(dict(real=real, n=n, arm=name, plan=plan.describe(), batch=batch, ms=round(t * 1e3, 3), frac8=round(batch * 2.0 * n * esz / t / 8e12, 4),
                                      rel_l2_vs_torch_f64=info[name][0], round_trip_rel_l2=info[name][1], kernels_ms={k: round(ms, 3) for k, ms, c in prof if c})), flush=True)
            del x, y, z, plans, ref
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
