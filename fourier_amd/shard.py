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
        """inputs/outputs: per-shard torch CUDA tensors (or raw (ptr, batch) pairs).  Returns when every shard is done."""
        import threading

        if not (len(inputs) == len(outputs) == len(self.plans)):
            raise ValueError("one input and one output per shard")
        handles = self._stream_handles(inputs)
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
                        self._streams[g].wait_stream(torch.cuda.current_stream(plan.device))  # inputs produced on the current stream
                        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), x.numel() // plan.size(), int(transform), handles[g])
                        self._streams[g].synchronize()
                else:
                    (xp, nb), (yp, _) = x, y
                    plan.transform_batch_ptr(xp, yp, nb, int(transform), 0)
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
