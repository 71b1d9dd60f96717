"""Shared test helpers: reproducible inputs, naive DFT, the reference's float_cmp semantics."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)
from make_golden import hash_normal, hash_uniform, sample_bins  # noqa: E402,F401


def load_ref10():
    with open(os.path.join(GOLDEN, "ref_dft10.json")) as f:
        d = json.load(f)
    x = np.array([complex(*p) for p in d["x"]])
    y = np.array([complex(*p) for p in d["y"]])
    return x, y


def naive_dft(x, inverse=False):
    """The reference test's own oracle (fourier/tests/integrity.rs:6-40): twiddle angle in f64,
    cast to T, accumulate in T.  IDFT folds 1/N into the twiddle."""
    x = np.asarray(x)
    n = x.shape[0]
    k = np.arange(n, dtype=np.int64)
    f = np.pi * (2.0 * np.outer(k, k)) / n
    if inverse:
        w = (np.cos(f) / n + 1j * (np.sin(f) / n)).astype(x.dtype)
    else:
        w = (np.cos(f) - 1j * np.sin(f)).astype(x.dtype)
    out = np.zeros(n, dtype=x.dtype)
    for j in range(n):  # accumulate in T, in the reference's order over n
        out += w[:, j] * x[j]
    return out


def _ulps(a, b):
    a = np.asarray(a)
    it = np.int32 if a.dtype == np.float32 else np.int64
    ai = a.view(it).astype(np.int64 if it == np.int32 else object)
    bi = np.asarray(b).view(it).astype(np.int64 if it == np.int32 else object)
    # float_cmp ulps: distance in sign-magnitude integer space when signs agree
    return np.abs(ai - bi)


def near(actual, expected, eps, ulps=8):
    """float_cmp::approx_eq! semantics (integrity.rs:89-143): |a-b| <= eps OR within `ulps` ulps,
    applied to re and im separately.  Returns (ok, worst_abs_diff)."""
    actual = np.asarray(actual)
    expected = np.asarray(expected).astype(actual.dtype)
    ok = True
    worst = 0.0
    for a, b in ((actual.real, expected.real), (actual.imag, expected.imag)):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        close = d <= eps
        if not close.all():
            same_sign = np.signbit(a) == np.signbit(b)
            u = _ulps(a, b)
            close = close | (same_sign & (np.array(u, dtype=np.float64) <= ulps))
        ok = ok and bool(close.all())
        worst = max(worst, float(d.max()) if d.size else 0.0)
    return ok, worst


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def max_rel(a, b):
    a = np.asarray(a, dtype=np.complex128)
    b = np.asarray(b, dtype=np.complex128)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def regfft_shape(n, dtype, emu=False, on_request=False):
    """The register stages "R1xR2[xR3]" of a length that runs as a direct transform in one launch (fourier_amd/csrc/regfft_shapes.h, generated from
    the A/B tables of sessions 49 - 66 by tools/gen_regfft_shapes.py), or None where the length keeps another route in this precision
    (emu: in the CPU emulation build, which holds the subset its test names; on_request: the 2^a 3^b lengths that take register stages only
    under plan option "register_stages")."""
    import re

    global _REGFFT_ROWS
    try:
        rows = _REGFFT_ROWS
    except NameError:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fourier_amd", "csrc", "regfft_shapes.h")
        with open(path) as f:
            text = f.read()
        rows = _REGFFT_ROWS = {kind: {int(m.group(1)): tuple(int(v) for v in m.groups()[1:])
                                      for m in re.finditer(r"^FOURIER_REGFFT_%sROW\((\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d)\)" % kind, text, re.M)}
                               for kind in ("", "OPT_")}
    r = rows["OPT_" if on_request else ""].get(int(n))
    if r is None or not r[3 if np.dtype(dtype).itemsize == 8 else 4] or (emu and not r[5]):
        return None
    return f"{r[0]}x{r[1]}" + (f"x{r[2]}" if r[2] else "")
