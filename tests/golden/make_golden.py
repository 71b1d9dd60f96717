#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run once, in the build container).

Everything here is DATA (inputs + expected outputs):
  * ref_dft10.json    -- the 10-point x / y=DFT(x) literals held by the reference's own test
                         (fourier/tests/integrity.rs:48-72), verbatim.
  * sweep_1_255.npz   -- the reference sweep's procedure (integrity.rs:145-192) made reproducible:
                         one 256-sample complex vector per direction (sigma=1 forward, sigma=256
                         inverse), and numpy float64 FFT/IFFT of every prefix of length 1..255.
  * n4096.npz         -- full float64 spectrum at N=4096 (BASELINE config C1).
  * big_samples.npz   -- N=2^20, 2^22, 999983: 64 sampled bins + the spectrum's L2 norm, float64.
Inputs come from a counter-based hash (hash_uniform below) so that tests can regenerate them
bit-exactly on any machine without committing megabytes.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MASK = (1 << 64) - 1


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(MASK)
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(MASK)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(MASK)
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed, n, lo=-1.0, hi=1.0):
    """n complex128 samples, re/im i.i.d. uniform [lo,hi), pure function of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(2 * n, dtype=np.uint64)
        h = splitmix64(idx ^ splitmix64(np.uint64(seed) + np.zeros(1, dtype=np.uint64)))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    u = lo + (hi - lo) * u
    return u[0::2] + 1j * u[1::2]


def hash_normal(seed, n, sigma=1.0):
    """n complex128 samples, re/im i.i.d. N(0, sigma) via Box-Muller on hash_uniform."""
    u = hash_uniform(seed, n, 0.0, 1.0)
    r = np.sqrt(-2.0 * np.log(1.0 - u.real))
    return sigma * (r * np.cos(2 * np.pi * u.imag) + 1j * r * np.sin(2 * np.pi * u.imag))


REF_X = [
    (0.07984231300862901, 0.2912053597430635), (-0.3999645806965225, 0.5665336963535724),
    (-1.0278505586819058, 0.503591759111203), (-0.5847182112607883, 0.2852956847818571),
    (0.8165939265478418, 0.48428811274975), (-0.08194705182666534, 1.3634815124261457),
    (-0.3447660142546443, -0.781105283625392), (0.5282881452973941, -0.4680176663374855),
    (-1.0689887834801322, 1.2245743551261743), (-0.5118813091268151, -1.2811082751440426),
]
REF_Y = [
    (-2.5953921244736087, 2.188739255184846), (0.27239725684518834, -0.5487581762070741),
    (1.2911356591694985, 0.4115497080289079), (5.181762895312528, -4.330109311527908),
    (1.432856335350818, 4.664992454671986), (-0.4949461092468147, 1.2563693510247518),
    (-2.0558954508390226, 1.2359845182788503), (-0.7015751667411471, -0.6481366043854868),
    (1.9167718867021326, -0.22783157531854403), (-3.448692051993283, -1.0907460223196943),
]

BIG = [(1 << 20, 0x5EED0020), (1 << 22, 0x5EED0022), (999983, 0x5EED0099)]
NSAMP = 64


def sample_bins(n):
    return (np.arange(NSAMP, dtype=np.int64) * 2654435761 + 12345) % n


def main():
    with open(os.path.join(HERE, "ref_dft10.json"), "w") as f:
        json.dump({"source": "fourier/tests/integrity.rs:48-72", "x": REF_X, "y": REF_Y}, f, indent=1)

    xf = hash_normal(0xDEADBEEF, 256, 1.0)
    xi = hash_normal(0xDEADBEEF + 1, 256, 256.0)
    fwd = np.concatenate([np.fft.fft(xf[:n]) for n in range(1, 256)])
    inv = np.concatenate([np.fft.ifft(xi[:n]) for n in range(1, 256)])
    np.savez_compressed(os.path.join(HERE, "sweep_1_255.npz"), x_fwd=xf, x_inv=xi, y_fwd=fwd, y_inv=inv)

    x = hash_normal(0x5EED0012, 4096, 1.0)
    np.savez_compressed(os.path.join(HERE, "n4096.npz"), seed=np.uint64(0x5EED0012), y=np.fft.fft(x))

    out = {}
    for n, seed in BIG:
        x = hash_uniform(seed, n)
        y = np.fft.fft(x)
        bins = sample_bins(n)
        out[f"seed_{n}"] = np.uint64(seed)
        out[f"bins_{n}"] = bins
        out[f"y_{n}"] = y[bins]
        out[f"l2_{n}"] = np.linalg.norm(y)
        out[f"maxabs_{n}"] = np.abs(y).max()
    np.savez_compressed(os.path.join(HERE, "big_samples.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
