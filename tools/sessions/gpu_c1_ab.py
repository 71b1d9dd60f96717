import ctypes, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from fourier_amd import fft as F, _lib
libs = [("product", None), ("r02", os.path.join(os.getcwd(), "fourier_amd/lib/variants/libfourier_r02.so"))]
base = _lib.lib()
for n in (4096, 256, 1000, 65536):
    x = (np.random.default_rng(0).random(n) + 1j * np.random.default_rng(1).random(n)).astype(np.complex64); y = np.empty_like(x)
    for rep in range(2):
        for name, path in libs:
            _lib._lib = base if path is None else _lib.bind(ctypes.CDLL(path), strict=False)
            p = F.create_fft_f32(n, 0)
            for _ in range(50): p.transform(x, y, 0)
            t0 = time.perf_counter()
            for _ in range(2000): p.transform(x, y, 0)
            us = (time.perf_counter() - t0) / 2000 * 1e6
            err = float(np.linalg.norm(y - np.fft.fft(x.astype(np.complex128))) / np.linalg.norm(np.fft.fft(x.astype(np.complex128))))
            print(json.dumps(dict(n=n, lib=name, legacy_host_call_us=round(us, 2), rel_l2=err)), flush=True)
            del p
_lib._lib = base
