"""N>1 host logic on CPU: world_size-2 gloo ranks shard the batch index contiguously (no data-path
collective), each rank transforms its own range, only timings/rows are reduced.
Compute goes through the test-only emulation build here; on GPUs the same code runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
from emu import build_emu
from fourier_amd import _lib
_lib._lib = build_emu.load()  # test-side monkeypatch: route the operator layer to the emulation build
import fourier_amd as fa
from fourier_amd import shard
from helpers import hash_normal
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
N, B = 512, 7
lo, hi = shard.batch_shard(B, world, rank)
x = np.stack([hash_normal(1000 + b, N) for b in range(lo, hi)]).astype(np.complex64)
plan = fa.create_fft_f32(N)
y = np.empty_like(x)
plan.transform_batch_ptr(x.ctypes.data, y.ctypes.data, hi - lo, int(fa.Transform.Fft))
rows = shard.gather_rows((lo, hi, y), dist)
tmax = shard.reduce_max_seconds(0.25 * (rank + 1), dist)
every = shard.gather_seconds(0.25 * (rank + 1), dist)
assert every == [0.25 * (r + 1) for r in range(world)], every  # rank order, on every rank
if rank == 0:
    full = np.concatenate([r[2] for r in sorted(rows, key=lambda r: r[0])])
    covered = sorted((r[0], r[1]) for r in rows)
    np.save({out!r}, full)
    print("RESULT", covered, tmax, flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_batch_shard_ranges():
    from fourier_amd.shard import batch_shard, owner_of

    for B in (1, 7, 4096, 65536):
        for G in (1, 2, 4, 8):
            ranges = [batch_shard(B, G, r) for r in range(G)]
            assert ranges[0][0] == 0 and ranges[-1][1] == B
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(G - 1))
            assert max(h - l for l, h in ranges) - min(h - l for l, h in ranges) <= 1
    assert owner_of(3, 7, 2) == 1 and owner_of(2, 7, 2) == 0


def test_two_ranks_gloo_shard_and_reduce(tmp_path, oracle):
    from emu import build_emu

    build_emu.build()
    out = str(tmp_path / "full.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=out))
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HIPEMU_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT")][0]
    assert "[(0, 3), (3, 7)]" in line and line.rstrip().endswith("0.5")  # max over ranks of 0.25, 0.5
    from helpers import hash_normal

    x = np.stack([hash_normal(1000 + b, 512) for b in range(7)]).astype(np.complex64)
    ref = oracle.transform_batch(x, oracle.FFT)
    got = np.load(out)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 1e-6
