"""Multi-GPU host logic: the batch index shards contiguously across ranks, no data-path collective.

SURVEY.md section 8(e): transforms are independent (the reference's `Fft::transform` takes one slice,
fourier-algorithms/src/fft.rs:51-61), so rank g of G owns transforms [floor(g*B/G), floor((g+1)*B/G)).
Only timings / checksums are reduced across ranks (torch.distributed; RCCL on GPUs, gloo in CPU tests).
"""


def batch_shard(global_batch, world, rank):
    """Contiguous [begin, end) range of transform indices owned by `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (global_batch * rank) // world, (global_batch * (rank + 1)) // world


def owner_of(b, global_batch, world):
    """Rank owning transform b (inverse of batch_shard)."""
    for r in range(world):
        lo, hi = batch_shard(global_batch, world, r)
        if lo <= b < hi:
            return r
    raise ValueError("transform index out of range")


def reduce_max_seconds(seconds, dist=None, device=None):
    """Whole-job time = max over ranks (each rank barrier+sync brackets its own timed region)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_seconds(seconds, dist=None, device=None):
    """Every rank's own figure, in rank order (the multi-GPU bench line carries them so that a straggler is visible)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(seconds)]
    import torch

    mine = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def gather_rows(local_rows, dist=None):
    """Gather per-rank numpy row blocks (used for checksums / parity samples, not in the data path)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_rows]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local_rows)
    return out


class DeviceShardedFft:
    """In-process form of SURVEY.md section 8(e): one plan, one host thread and one HIP stream per device.

    `devices` lists the device index of every shard (a device may appear more than once; handles are Send, not
    Sync, so every shard gets its own plan).  `transform(inputs, outputs, transform)` takes one device tensor per
    shard -- shard g holds transforms [floor(g*B/G), floor((g+1)*B/G)) of the global batch, see batch_shard -- and
    runs all shards concurrently, each on its own non-default stream from its own host thread, through the C ABI
    (`fourier_hip_create_*(size, device)` + `fourier_hip_transform_batch_*`).  No data moves between devices.
    """

    def __init__(self, size, real, devices):
        from . import fft as F

        self.devices = [int(d) for d in devices]
        make = F.create_fft_f32 if real == "f32" else F.create_fft_f64
        self.plans = [make(size, d) for d in self.devices]
        self._streams = None

    def _stream_handles(self, inputs):
        if _is_torch_tensor(inputs[0]):
            import torch

            if self._streams is None:
                self._streams = [torch.cuda.Stream(device=d) for d in self.devices]
            return [s.cuda_stream for s in self._streams]
        return [0] * len(self.devices)

    def transform(self, inputs, outputs, transform):
        """inputs/outputs: per-shard torch CUDA tensors (or raw (ptr, batch) pairs, which run on the NULL stream of
        their device).  Returns when every shard is done: each worker synchronises its stream before it exits."""
        import threading

        if not (len(inputs) == len(outputs) == len(self.plans)):
            raise ValueError("one input and one output per shard")
        handles = self._stream_handles(inputs)
        # torch's current stream is thread-local: the streams the CALLER produced the inputs on are looked up here, on
        # the calling thread, and handed to the workers (a worker thread would only ever see the default stream)
        producers = [None] * len(self.plans)
        if _is_torch_tensor(inputs[0]):
            import torch

            producers = [torch.cuda.current_stream(d) for d in self.devices]
        errors = [None] * len(self.plans)

        def work(g):
            try:
                plan, x, y = self.plans[g], inputs[g], outputs[g]
                if _is_torch_tensor(x):
                    import torch

                    if x.device.index != plan.device or y.device.index != plan.device:
                        raise ValueError(f"shard {g}: tensors on cuda:{x.device.index}/{y.device.index}, plan on cuda:{plan.device}")
                    if x.numel() != y.numel() or x.numel() % plan.size() != 0:
                        raise ValueError(f"shard {g}: not a whole number of transforms")
                    with torch.cuda.device(plan.device):
                        self._streams[g].wait_stream(producers[g])  # inputs produced on the caller's current stream
                        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), x.numel() // plan.size(), int(transform), handles[g])
                        self._streams[g].synchronize()
                else:
                    (xp, nb), (yp, _) = x, y
                    plan.transform_batch_ptr(xp, yp, nb, int(transform), 0)
                    plan.synchronize(0)  # fourier_hip_synchronize_*: the NULL stream of the plan's device
            except Exception as e:  # re-raised on the calling thread
                errors[g] = e

        threads = [threading.Thread(target=work, args=(g,)) for g in range(len(self.plans))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None:
                raise e


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch")
