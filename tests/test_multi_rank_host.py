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


def test_bench_dry_run_validates_the_sharding_of_the_full_c5_job():
    """`bench.py --gpus N --dry-run` (no GPU, no process group): the shard ranges, chunk walk and byte counts every rank of a
    1 / 2 / 4 / 8-GPU run of BASELINE configs[4] (f32 N=2^22, GLOBAL batch 65536 = 2 TiB) would execute -- what the driver's
    scaling step launches at round end on hardware this container does not have."""
    import json

    for gpus in (1, 2, 4, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", "c5", "--dry-run"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1
        out = json.loads(lines[0])
        assert out["config_key"] == "c5" and out["scaling"] == "strong" and out["n_gpus"] == gpus and out["n"] == 1 << 22
        assert out["global_batch"] == 65536 and sum(out["transforms_per_rank"]) == 65536
        assert out["transforms_per_rank"] == [65536 // gpus] * gpus
        assert out["tiles_global_batch_exactly_once"] and out["fits_hbm_per_gpu"] and out["collectives_on_the_data_path"] == 0
        for rk in out["ranks"]:
            assert rk["chunk"] == 1024 and rk["chunks_per_step"] == 64 // gpus and rk["ragged_last_chunk"] is None
            assert rk["resident_bytes"] == 2 * 1024 * (1 << 22) * 8  # 64 GiB: input + output chunk
        assert out["algorithmic_bytes_per_step"] == 65536 * 2 * (1 << 22) * 8
    # the default under torch.distributed.run with more than one rank IS c5 (auto), a ragged split still tiles the batch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--batch", "1000", "--chunk", "128"],
                       capture_output=True, text=True, timeout=120)
    out = json.loads(r.stdout.strip())
    assert r.returncode == 0 and out["config_key"] == "c5" and out["transforms_per_rank"] == [333, 333, 334]
    assert out["tiles_global_batch_exactly_once"] and [rk["ragged_last_chunk"] for rk in out["ranks"]] == [77, 77, 78]
