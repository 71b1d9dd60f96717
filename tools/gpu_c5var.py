import sys, os, ctypes, json, math, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fourier_amd import _lib, fft as F
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from gpu_c4c5 import run
base = _lib.lib()
for name in ("base", "cg2048x8"):
    path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "fourier_amd/lib/variants", f"libfourier_{name}.so")
    _lib._lib = _lib.bind(ctypes.CDLL(path))
    run(f"{name} 2^22", 1 << 22, 512)
    run(f"{name} 2^21", 1 << 21, 1024)
    run(f"{name} C4", 999983, 512)
