#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched 1D c2c FFT hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched under
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.
metric = nominal 5*N*log2(N) GFLOP/s, whole job (all ranks).

Workloads (`--config`, default `auto`):
  c2    BASELINE.json configs[1]: f32 N=2^20, batch 4096 per GPU, forward, out of place, inputs resident in
        HBM before the timed region.  A "step" = one pass of the hot path over the batch.  THE DEFAULT AT N=1.
  c5    BASELINE.json configs[4]: f32 N=2^22, GLOBAL batch 65536 (2 TiB of input) split contiguously over the
        ranks (fourier_amd.shard.batch_shard; no data-path collective) -- "scaling": "strong".  The shard does
        not fit in HBM, so a rank walks it in fixed chunks of 1024 transforms (32 GiB), each regenerated on the
        device (seeded by its global chunk index) right before it is transformed.  A "step" = one pass over the
        whole 65536-transform job; the step time is the FFT time (chunk generation is bracketed out with
        device syncs and reported separately as wall_ms_per_step_incl_regen).  THE DEFAULT AT N>1: this is the
        configuration BASELINE.json names for the 1/2/4/8-GPU scaling curve.
  c3 / c4   configs[2] (f64 N=2^20 x 4096) and configs[3] (Bluestein N=999983 f32 x 512), same form as c2.
`auto` = c2 on one GPU, c5 under torch.distributed.run with more than one rank.  The line names its configuration
(`config_key`); a c5 line carries its own single-GPU reference (`strong_scaling.single_gpu_reference`: rank 0 alone on
two chunks while the other ranks idle), every rank's time, the process group's backend and world size, and the
efficiency against that reference, so that one record is enough to judge batch-shard scaling.

Extra objects in the JSON line:
  roofline      -- dominant kernel, algorithmic bytes per launch / HIP-event duration (events on the
                   launch stream, inside this process), peak 8 TB/s; `traffic` from the committed
                   rocprofv3 PMC pass (profiles/traffic_latest.json) or null.  Also measured IN THIS RUN, on
                   the workload's own buffers: `stream_ceiling_gbps` = the best pure load -> store form on this
                   box -- eleven hand-written device copies (`copy_ceiling_gbps`, csrc/exp_copy_ceiling.cpp) AND the
                   pass kernels' own skeletons in the f32 and f64 shape (`skeleton_ceiling`, csrc/kernels_skeleton.cpp;
                   both from lib/libfourier_experiments.so, measurement tooling) --, `frac_of_stream_ceiling` for
                   the dominant kernel, `round_trips` of the plan through HBM and the bound they put on the whole
                   path (`whole_path_bound_frac` = stream ceiling / round trips / 8 TB/s).  Scalars `c3_*`, `c4_*`,
                   `c5_*` (ms per step, whole-path fraction, dominant kernel's fraction) repeat other_configs where a
                   driver-side record keeps them (also under `config`).
  cpu_baseline  -- the oracle (CPU restatement of the reference, kind "port") timed on this box's host
                   cores on a bounded sample of the same workload (rank 0, N=1 only).
  other_configs -- (N=1, default config only) a few seconds each on C1 (one N=4096 transform per call through the
                   host-slice API, beside the CPU port), C3, C4 and one C5 chunk after the headline timing: ms,
                   algorithmic fraction, dominant-kernel fraction.  The reference's 60-row criterion size grid is
                   summarised in the line (min / max fraction per size set) and printed in full on stderr / --details.
The line's members are ordered so that metric, value, ms_per_step and the scalar roofline block come LAST.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    "c2": dict(n=1 << 20, batch=4096, dtype="f32", name="BASELINE configs[1]"),
    "c3": dict(n=1 << 20, batch=4096, dtype="f64", name="BASELINE configs[2]"),
    "c4": dict(n=999983, batch=512, dtype="f32", name="BASELINE configs[3]"),
    "c5": dict(n=1 << 22, global_batch=65536, chunk=1024, dtype="f32", name="BASELINE configs[4]"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", default="auto", choices=["auto", "c2", "c3", "c4", "c5"])
    p.add_argument("--log2n", type=int, default=None, help="override the transform length (c2/c3 form)")
    p.add_argument("--batch", type=int, default=None, help="transforms per GPU per step (c2/c3/c4) or global batch (c5)")
    p.add_argument("--chunk", type=int, default=None, help="c5: transforms per resident chunk")
    p.add_argument("--dtype", default=None, choices=["f32", "f64"])
    p.add_argument("--chunk-bytes", type=int, default=None, help="override the plan's chunk_bytes option")
    p.add_argument("--scratch", type=int, default=None, help="override the plan's scratch option (0/1)")
    p.add_argument("--inplace", action="store_true")
    p.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--no-other", action="store_true", help="skip the other_configs leg")
    p.add_argument("--no-ceiling", action="store_true", help="skip the in-run streaming-ceiling measurement")
    p.add_argument("--details", default=None, help="also write the complete record (with the reference's 60-row size grid) to this file")
    p.add_argument("--cpu-sample", type=int, default=0, help="transforms in the CPU sample (0 = auto)")
    p.add_argument("--dry-run", action="store_true",
                   help="no GPU, no process group: print the shard ranges, chunk walk and byte counts a --gpus N run of this config "
                        "WOULD execute (every rank's view), and check that they tile the global batch exactly")
    return p.parse_args()


def emit(out, args):
    """Rank 0's ONE JSON line on stdout.  Bulky members (the reference's size grid) go to stderr and to --details; the
    members a reader wants first -- metric, value, ms_per_step, the scalar roofline block with every BASELINE config --
    come LAST in the line, because a truncated tail of stdout is what a driver-side record shows."""
    full = json.dumps(out)
    if args.details:
        try:
            with open(args.details, "w") as f:
                f.write(full + "\n")
        except OSError:
            pass
    line = dict(out)
    oc = line.get("other_configs")
    if isinstance(oc, dict) and "reference_bench_sizes" in oc:
        grid = oc["reference_bench_sizes"]
        sys.stderr.write(json.dumps({"reference_bench_sizes": grid}) + "\n")
        oc = {k: v for k, v in oc.items() if not k.startswith("reference_bench_sizes")}
        if isinstance(grid, list):  # per scenario / precision: the range of the algorithmic fraction of 8 TB/s (forward and inverse)
            summ = {}
            for r in grid:
                a = summ.setdefault(f"{r['scenario']}_{r['dtype']}", [1.0, 0.0])
                a[0], a[1] = min(a[0], r["hbm_frac_algorithmic"]), max(a[1], r["hbm_frac_algorithmic"])
            oc["reference_bench_sizes_summary"] = {k: [round(v[0], 3), round(v[1], 3)] for k, v in summ.items()}
            oc["reference_bench_sizes_rows"] = len(grid)
        line["other_configs"] = oc
    tail_keys = ["cpu_baseline", "config", "roofline", "metric", "unit", "ms_per_step", "value"]
    ordered = {k: v for k, v in line.items() if k not in tail_keys}
    r = line.get("roofline")
    if isinstance(r, dict):  # nested members first, scalars last
        line["roofline"] = {**{k: v for k, v in r.items() if isinstance(v, (dict, list))}, **{k: v for k, v in r.items() if not isinstance(v, (dict, list))}}
    for k in tail_keys:
        if k in line:
            ordered[k] = line[k]
    print(json.dumps(ordered), flush=True)


def nominal_flops(n):
    return 5.0 * n * math.log2(n)


def make_plan(fourier_amd, n, dtype, device, args=None):
    plan = (fourier_amd.create_fft_f32 if dtype == "f32" else fourier_amd.create_fft_f64)(n, device)
    if args is not None:
        if args.chunk_bytes is not None:
            plan.set_option("chunk_bytes", args.chunk_bytes)
        if args.scratch is not None:
            plan.set_option("scratch", args.scratch)
    return plan


def kernel_profile(plan, x_ptr, y_ptr, batch, stream, reps=3):
    """Per-kernel ms of one batched transform: HIP events on the launch stream (fourier_hip_profile_*)."""
    from fourier_amd import Transform

    acc = {}
    for _ in range(reps):
        for name, ms, cnt in plan.profile_batch_ptr(x_ptr, y_ptr, batch, int(Transform.Fft), stream):
            if cnt:
                a = acc.setdefault(name, [0.0, 0])
                a[0] += ms
                a[1] += cnt
    return {k: {"ms_per_step": v[0] / reps, "launches_per_step": v[1] // reps} for k, v in acc.items()}


def copy_ceiling(src_ptr, dst_ptr, nbytes, stream):
    """GB/s (read + written bytes) of a hand-written slab copy src -> dst on this box, this run: measurement tooling from
    lib/libfourier_experiments.so (fourier_exp_copy_ceiling), never the product library.  None when unavailable."""
    import ctypes

    from fourier_amd import build as B

    try:
        lib = ctypes.CDLL(B.OUT_EXPERIMENTS)
        f = lib.fourier_exp_copy_ceiling
    except (OSError, AttributeError):
        return None
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                  ctypes.POINTER(ctypes.c_float)]
    out = {}
    slab = 1 << 20  # bytes per workgroup
    nbytes = min(nbytes, 16 << 30) // (128 * slab) * (128 * slab)  # whole slabs / 8 and 16 MiB "transforms", a multiple of 8 per XCD
    if nbytes <= 0:
        return None
    # (label, access bits: 1 = streaming hint, 2 = the passes' column-tile shape, resident workgroups per CU: 0 = as many as fit)
    for label, nt, wgs in (("slab_plain", 0, 0), ("slab_streaming", 1, 0), ("column_tiles_plain", 2, 0), ("column_tiles_streaming", 3, 0),
                           ("column_tiles_streaming_2_per_cu", 3, 2), ("column_tiles_streaming_3_per_cu", 3, 3),
                           ("column_tiles_plain_2_per_cu", 2, 2), ("slab_streaming_4_per_cu", 1, 4),
                           ("column_tiles_16k_rows_streaming", 7, 0), ("column_tiles_16k_rows_streaming_2_per_cu", 7, 2),
                           ("column_tiles_16k_rows_plain", 6, 0)):
        ms = ctypes.c_float(0.0)
        rc = f(src_ptr, dst_ptr, nbytes, slab, nt, wgs, 5, stream, ctypes.byref(ms))
        if rc != 0 or ms.value <= 0:
            return None
        out[label] = 2.0 * nbytes / (ms.value * 1e-3) / 1e9
    best = max(out, key=out.get)
    return {"gbps": round(out[best], 1), "policy": best, "bytes_copied": nbytes, "by_policy_gbps": {k: round(v, 1) for k, v in out.items()},
            "how": "best of eleven hand-written device copies (16-byte accesses, XCD-aware block order): linear 1 MiB slabs per workgroup "
                   "with 8 loads in flight per thread, and the passes' own shape -- 128-byte row segments at an 8 KiB (f32 plan) or 16 KiB (f64 plan) row stride, 16 loads "
                   "in flight per thread, one 128 KiB column tile per 512-thread workgroup --, with plain and with streaming accesses, with "
                   "as many workgroups per CU as fit and capped at the pass kernels' two (three, four); 5 launches between HIP events on "
                   "the launch stream, this process, the workload's own buffers"}


def skeleton_ceiling(x_ptr, y_ptr, nbytes, stream):
    """GB/s (read + written bytes) of the pass kernels' own load-tile / store-tile SKELETONS -- the 1024 x 1024 plans of f32
    (8 KiB rows) and f64 (16 KiB rows) with butterflies, twiddles and LDS exchanges compiled out (csrc/kernels_skeleton.cpp,
    plan option "skeleton" of lib/libfourier_experiments.so; same grid, tile order, cache policy, two workgroups per CU) --
    on the workload's own buffers, HIP events per kernel, best of 3.  VERDICT round 4 item 1a: the product's f64 passes and
    these skeletons out-stream every hand-written copy form, so they belong in the ceiling.  None when unavailable."""
    import ctypes

    from fourier_amd import _lib, build as B, fft as F

    try:
        exp = _lib.bind(ctypes.CDLL(B.OUT_EXPERIMENTS), strict=False)
    except OSError:
        return None
    product = _lib.lib()
    out = {}
    try:
        for label, real, esz in (("skeleton_f32_8k_rows", "f32", 8), ("skeleton_f64_16k_rows", "f64", 16)):
            n = 1 << 20
            batch = nbytes // (n * esz) // 8 * 8
            if batch <= 0:
                continue
            _lib._lib = exp
            try:
                plan = (F.create_fft_f32 if real == "f32" else F.create_fft_f64)(n, -1)
                plan.set_option("skeleton", 1)
            finally:
                _lib._lib = product
            best = {}
            for _ in range(3):
                for name, ms, cnt in plan.profile_batch_ptr(x_ptr, y_ptr, batch, 0, stream):
                    if cnt:
                        best[name] = min(best.get(name, 1e30), ms)
            for name, ms in best.items():
                out[f"{label}_{name}"] = 2.0 * batch * n * esz / (ms * 1e-3) / 1e9
            del plan
    except Exception:
        _lib._lib = product
        return None
    if not out:
        return None
    best = max(out, key=out.get)
    return {"gbps": round(out[best], 1), "form": best, "by_form_gbps": {k: round(v, 1) for k, v in out.items()}}


def roofline_of(plan, kernels, batch, alg_bytes_per, dtype, whole_path_frac, traffic_ok=False, ceiling=None, skeleton=None):
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    dom_ms = kernels[dom]["ms_per_step"]
    achieved = batch * alg_bytes_per / (dom_ms * 1e-3) / 1e9  # all launches of that kernel in a step cover the batch
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if traffic_ok and os.path.exists(tpath):  # the committed PMC pass was taken on the default workload only
        try:
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj.get("per_launch_bytes", {}).get(dom)
        except Exception:
            traffic = None
    r = {
        "bound": "hbm", "kernel": dom,
        "rocprof_name": f"fourier_hip kernel behind slot '{dom}' of plan {plan.describe()} "
                        f"({'float' if dtype == 'f32' else 'double'})",
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "algorithmic_bytes_per_launch": batch * alg_bytes_per / max(kernels[dom]["launches_per_step"], 1),
        "kernels": {k: {"ms_per_step": round(v["ms_per_step"], 4), "launches_per_step": v["launches_per_step"]}
                    for k, v in kernels.items()},
        "whole_path_frac": round(whole_path_frac, 4),
    }
    trips = sum(1 for v in kernels.values() if v["launches_per_step"])  # launches that each move the whole batch through HBM
    r["round_trips"] = trips
    if ceiling:
        r["copy_ceiling_gbps"] = ceiling["gbps"]
        r["copy_ceiling"] = ceiling
        r["frac_of_copy_ceiling"] = round(achieved / ceiling["gbps"], 4)
    # the STREAMING ceiling of this box, this run: the best of every pure load -> store form we can write -- the hand-written
    # copies above AND the pass kernels' own skeletons (f32 and f64 shape).  Round 4's `copy_ceiling_gbps` alone under-reached what
    # the product's own f64 passes stream; everything "of ceiling / of bound" is priced against this.
    forms = {}
    if ceiling:
        forms.update({"copy_" + k: v for k, v in ceiling["by_policy_gbps"].items()})
    if skeleton:
        forms.update(skeleton["by_form_gbps"])
        r["skeleton_ceiling"] = skeleton
    if forms:
        best = max(forms, key=forms.get)
        sc = forms[best]
        r["stream_ceiling_gbps"] = round(sc, 1)
        r["stream_ceiling_form"] = best
        r["frac_of_stream_ceiling"] = round(achieved / sc, 4)
        # a plan that moves every point through HBM `trips` times cannot beat (stream ceiling / trips) on the algorithmic bytes
        r["whole_path_bound_frac"] = round(sc / trips / HBM_PEAK_GBPS, 4)
        r["whole_path_frac_of_bound"] = round(whole_path_frac / (sc / trips / HBM_PEAK_GBPS), 4)
    return r


def reference_bench_sizes(fourier_amd, torch, dev, bytes_per_size=1 << 29):
    """The reference's own criterion grid (fourier-bench/benches/fft_bench.rs:153-159: powers of two / three / five, composites,
    primes; forward and inverse; f32 and f64), device-resident and batched: plan, time per transform, fraction of the HBM
    peak on the algorithmic bytes.  A few seconds in total; informative (the reference times one host transform per call)."""
    from fourier_amd import Transform

    rows = []
    stream = torch.cuda.current_stream(dev).cuda_stream
    for dtype, cdt, esz in (("f32", torch.complex64, 8), ("f64", torch.complex128, 16)):
        for scenario, sizes in (("pow2", (256, 512, 1024)), ("pow3", (243, 729, 2187)), ("pow5", (125, 625, 3125)),
                                ("composite", (222, 722, 1418)), ("prime", (191, 439, 1013))):
            for n in sizes:
                batch = max(1, bytes_per_size // (n * esz))
                plan = make_plan(fourier_amd, n, dtype, dev.index)
                x = torch.empty((batch, n), dtype=cdt, device=dev)
                torch.view_as_real(x).uniform_(0.0, 1.0)
                y = torch.empty_like(x)
                for direction, code in (("forward", Transform.Fft), ("inverse", Transform.Ifft)):
                    ts = []
                    for it in range(5):
                        t0 = time.perf_counter()
                        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(code), stream)
                        torch.cuda.synchronize(dev)
                        if it >= 2:
                            ts.append(time.perf_counter() - t0)
                    t = sorted(ts)[len(ts) // 2]
                    rows.append({"scenario": scenario, "n": n, "dtype": dtype, "direction": direction, "plan": plan.describe(), "batch": batch,
                                 "ns_per_transform": round(t / batch * 1e9, 2),
                                 "hbm_frac_algorithmic": round(batch * 2.0 * esz * n / t / 1e9 / HBM_PEAK_GBPS, 4)})
                del x, y, plan
    return rows


def quick_config(fourier_amd, torch, dev, key, reps=3, allocations=3):
    """A few seconds on another BASELINE config (N=1 only).  The time of the strided tile passes depends on WHERE the driver put the buffers
    (round 6, DESIGN section 4: an allocation's physical placement moves the f64 last pass between 21.9 and 24.5 ms, whatever the tile order;
    virtual offsets inside an allocation change nothing), so the config is timed on `allocations` FRESH pairs of buffers -- both allocation orders,
    the cache emptied in between -- and the record carries the median allocation's numbers plus min / max over the allocations."""
    from fourier_amd import Transform

    c = CONFIGS[key]
    n, dtype = c["n"], c["dtype"]
    batch = c.get("batch", c.get("chunk"))
    esz = 8 if dtype == "f32" else 16
    cdt = torch.complex64 if dtype == "f32" else torch.complex128
    plan = make_plan(fourier_amd, n, dtype, dev.index)
    stream = torch.cuda.current_stream(dev).cuda_stream
    runs = []
    for a in range(max(1, allocations)):
        if a % 2 == 0:
            x = torch.empty((batch, n), dtype=cdt, device=dev)
            y = torch.empty_like(x)
        else:  # the other order: the driver hands out different physical ranges
            y = torch.empty((batch, n), dtype=cdt, device=dev)
            x = torch.empty_like(y)
        torch.view_as_real(x).uniform_(0.0, 1.0)
        plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(Transform.Fft), stream)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(Transform.Fft), stream)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        runs.append((sorted(ts)[len(ts) // 2], kernel_profile(plan, x.data_ptr(), y.data_ptr(), batch, stream, reps=1)))
        del x, y
        torch.cuda.empty_cache()
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    t, kernels = runs[order[len(order) // 2]]  # the median allocation
    alg = 2.0 * n * esz
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    # HBM-side bytes per launch from the committed PMC passes (tools/gpu_r03_pmc.sh -> profiles/traffic_latest.json,
    # "configs"): per kernel, with the ratio to the algorithmic bytes of one launch (batch x 2 x N x sizeof(complex))
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tk = json.load(f).get("configs", {}).get(key, {})
        traffic = {k: {"hbm_side_bytes": v["hbm_side_bytes"], "over_algorithmic": round(v["hbm_side_bytes"] / (batch * alg), 3),
                       "l2_hit_rate": None if v.get("l2_hit_rate") is None else round(v["l2_hit_rate"], 3)}
                   for k, v in tk.get("kernels", {}).items() if k in kernels}
        traffic = {"source": tk.get("source"), "kernels": traffic} if traffic else None
    except Exception:
        traffic = None
    out = {
        "workload": f"{c['name']}: {dtype} N={n} batch={batch}" + (" (one chunk of the 65536-transform job)" if key == "c5" else ""),
        "plan": plan.describe(), "ms_per_step": round(t * 1e3, 3),
        "ms_min": round(runs[order[0]][0] * 1e3, 3), "ms_max": round(runs[order[-1]][0] * 1e3, 3),
        "fresh_allocations": [{"ms_per_step": round(r[0] * 1e3, 3), "kernels_ms": {k: round(v["ms_per_step"], 3) for k, v in r[1].items()}} for r in runs],
        "gflops": round(batch * nominal_flops(n) / t / 1e9, 1),
        "hbm_frac_algorithmic": round(batch * alg / t / 1e9 / HBM_PEAK_GBPS, 4),
        "dominant_kernel": dom,
        "dominant_kernel_frac": round(batch * alg / (kernels[dom]["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        "kernels_ms": {k: round(v["ms_per_step"], 3) for k, v in kernels.items()},
        "traffic": traffic,
    }
    del plan
    torch.cuda.empty_cache()
    return out


def config_c1(fourier_amd):
    """BASELINE configs[0]: ONE f32 N=4096 forward transform through the reference-compatible host-slice API
    (`Fft::transform` on host memory = fourier_transform_float: H2D + kernels + D2H inside the library, synchronous),
    beside the oracle (the reference's CPU path restated, AVX clone) on one core.  Plumbing + latency, not throughput."""
    import numpy as np
    from fourier_amd import Transform
    from oracle import oracle as O

    n = 4096
    rng = np.random.default_rng(1)
    x = (rng.random(n) + 1j * rng.random(n)).astype(np.complex64)
    y = np.empty_like(x)
    plan = fourier_amd.create_fft_f32(n)
    for _ in range(20):
        plan.transform(x, y, Transform.Fft)
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.transform(x, y, Transform.Fft)
    gpu_us = (time.perf_counter() - t0) / reps * 1e6
    orc = O.OracleFft(n, np.complex64)
    ref = orc.transform(x, O.FFT)
    t0 = time.perf_counter()
    for _ in range(reps):
        orc.transform(x, O.FFT)
    cpu_us = (time.perf_counter() - t0) / reps * 1e6
    err = float(np.linalg.norm(y.astype(np.complex128) - ref) / np.linalg.norm(ref))
    return {"workload": "BASELINE configs[0]: f32 N=4096, one forward transform per call on HOST buffers (legacy ABI, PCIe inclusive)",
            "plan": plan.describe(), "gpu_us_per_call": round(gpu_us, 2), "cpu_port_one_core_us_per_call": round(cpu_us, 2),
            "rel_l2_vs_oracle": err, "tolerance": 1e-6}


def cpu_baseline(torch, plan, hx, n, dtype, dev, stream, cores):
    """Parity sample + cpu_baseline: the oracle on this box's host cores (bounded sample)."""
    import numpy as np
    from fourier_amd import Transform
    from oracle import oracle as O

    O.build()
    flops_per = nominal_flops(n)
    sample = hx.shape[0]
    xs = torch.from_numpy(hx).to(dev)
    ys = torch.empty_like(xs)
    plan.transform_batch_ptr(xs.data_ptr(), ys.data_ptr(), sample, int(Transform.Fft), stream)
    torch.cuda.synchronize(dev)
    got = ys.cpu().numpy()
    # one oracle plan per thread (plans are Send, not Sync).  Persistent workers pinned across the affinity mask, each
    # building its own plan and first-touching its slice of the staged input and of the output (oracle.OracleBatch); a
    # warm-up pass over the WHOLE sample precedes the timed one, so the timed region holds transforms only -- no thread
    # creation, no page faults, no remote-node tables.  Thread counts from all host threads down to 1/32 of them.
    ref = np.empty_like(hx)
    tried, pinned = {}, {}
    for nt in sorted({max(1, cores >> k) for k in range(6)}, reverse=True):
        ob = O.OracleBatch(n, hx.dtype, nthreads=nt)
        ob.stage(hx)
        ob.run_staged(ref, O.FFT)  # warm-up over everything
        t0 = time.perf_counter()
        ob.run_staged(ref, O.FFT)
        tried[nt] = time.perf_counter() - t0
        pinned[nt] = sum(1 for c in ob.cpus() if c >= 0)
        del ob
    used = min(tried, key=tried.get)
    cpu_s = tried[used]
    all_s = tried[max(tried)]
    # the reference itself is single-threaded (one plan, one slice per call): the same port on ONE core, as the
    # reference's AVX clone (vector/avx.rs wide passes + radix_4_stride_1_avx_f32, what an AVX host runs; the
    # multi-thread legs above use it too) and as the generic scalar functions (what round 1 timed)
    k1 = min(sample, 4)
    one = {}
    for label, clone in (("avx_clone", O.AVX), ("generic_scalar", O.GENERIC)):
        if O.set_clone(clone) != clone:
            continue
        ob1 = O.OracleBatch(n, hx.dtype, nthreads=1)
        ref1 = np.empty_like(hx[:k1])
        ob1.run(hx[:1], O.FFT, out=ref1[:1])
        t0 = time.perf_counter()
        ob1.run(hx[:k1], O.FFT, out=ref1)
        one[label] = (time.perf_counter() - t0) / k1
        del ob1
    O.set_clone(O.AVX)
    one_core_s = one.get("avx_clone", one["generic_scalar"])
    err = float(np.linalg.norm(got.astype(np.complex128) - ref) / np.linalg.norm(ref))
    parity = {"sample_transforms": sample, "rel_l2_vs_oracle": err, "tolerance": 1e-6 if dtype == "f32" else 5e-14}
    base = {
        "value": round(sample * flops_per / cpu_s / 1e9, 2), "unit": "GFLOP/s", "cores": used,
        "host_threads_available": cores,
        "all_threads": {"threads": max(tried), "value": round(sample * flops_per / all_s / 1e9, 2), "unit": "GFLOP/s",
                        "pinned_workers": pinned[max(tried)]},
        "best": {"threads": used, "value": round(sample * flops_per / cpu_s / 1e9, 2), "unit": "GFLOP/s"},
        "threads_tried_gflops": {str(k): round(sample * flops_per / v / 1e9, 2) for k, v in tried.items()},
        "harness": "persistent pinned workers, one plan per worker built on its own CPU, input staged and output first "
                   "touched slice by slice by the owning worker, one full warm-up pass before the timed pass",
        "kind": "port", "clone": "avx" if O.have_avx() else "generic",
        "sample": f"{sample} of the same transforms ({dtype} N={n}, out-of-place), "
                  f"one oracle plan per thread on {used} threads, {cpu_s:.2f} s wall",
        "ms_per_transform_aggregate": round(cpu_s / sample * 1e3, 3),
        "one_core": {"ms_per_transform": round(one_core_s * 1e3, 3), "value": round(flops_per / one_core_s / 1e9, 3),
                     "unit": "GFLOP/s", "sample": f"{k1} transforms on 1 thread",
                     "ms_per_transform_by_clone": {k: round(v * 1e3, 3) for k, v in one.items()}},
    }
    return parity, base


def dry_run(args):
    """What `bench.py --gpus N` would do, rank by rank, without touching a GPU or a process group: configuration, shard of
    every rank, its chunk walk (first transform and length of every resident chunk), bytes resident and bytes streamed per
    step.  Checks that the chunks of all ranks tile [0, global batch) exactly once.  One JSON line."""
    from fourier_amd import shard

    world = args.gpus
    key = args.config if args.config != "auto" else ("c5" if world > 1 else "c2")
    cfg = dict(CONFIGS[key])
    dtype = args.dtype or cfg["dtype"]
    n = (1 << args.log2n) if args.log2n is not None else cfg["n"]
    esz = 8 if dtype == "f32" else 16
    ranks, covered = [], []
    if key == "c5":
        gbatch = args.batch or cfg["global_batch"]
        for r in range(world):
            lo, hi = shard.batch_shard(gbatch, world, r)
            chunk = max(1, min(args.chunk or cfg["chunk"], hi - lo))
            walk = [(b0, min(chunk, hi - b0)) for b0 in range(lo, hi, chunk)]
            covered += walk
            ranks.append({"rank": r, "transforms": [lo, hi], "chunk": chunk, "chunks_per_step": len(walk),
                          "ragged_last_chunk": walk[-1][1] if walk and walk[-1][1] != chunk else None,
                          "resident_bytes": chunk * n * esz * (1 if args.inplace else 2),
                          "algorithmic_bytes_per_step": (hi - lo) * 2 * n * esz})
        scaling = "strong"
    else:
        batch = args.batch or cfg["batch"]
        gbatch = world * batch
        for r in range(world):
            covered.append((r * batch, batch))
            ranks.append({"rank": r, "transforms": [r * batch, (r + 1) * batch], "chunk": batch, "chunks_per_step": 1, "ragged_last_chunk": None,
                          "resident_bytes": batch * n * esz * (1 if args.inplace else 2), "algorithmic_bytes_per_step": batch * 2 * n * esz})
        scaling = "weak"
    covered.sort()
    pos, ok = 0, True
    for b0, nb in covered:
        ok = ok and b0 == pos and nb > 0
        pos = b0 + nb
    ok = ok and pos == gbatch
    hbm = 288e9
    print(json.dumps({
        "dry_run": True, "config_key": key, "scaling": scaling, "n_gpus": world, "n": n, "dtype": dtype, "global_batch": gbatch,
        "transforms_per_rank": [r["transforms"][1] - r["transforms"][0] for r in ranks], "ranks": ranks,
        "tiles_global_batch_exactly_once": ok, "fits_hbm_per_gpu": all(r["resident_bytes"] < hbm for r in ranks),
        "collectives_on_the_data_path": 0,
        "nominal_flop_per_step": gbatch * nominal_flops(n), "algorithmic_bytes_per_step": gbatch * 2 * n * esz,
    }), flush=True)
    return 0 if ok else 1


def main():
    args = parse()
    if args.dry_run:
        raise SystemExit(dry_run(args))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the hot path)")
    # test hook (tests/test_gpu_parity.py): several ranks on a box with fewer GPUs share devices; RCCL refuses two ranks
    # on one device, so such a run names BENCH_BACKEND=gloo (only the max-over-ranks time goes through the group)
    if os.environ.get("BENCH_SHARE_DEVICES"):
        local_rank %= torch.cuda.device_count()
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the timing scalars are reduced
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):  # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries exactly one JSON line
        # (the driver parses it), so the banner is sent to stderr: fd 1 -> fd 2 around the communicator's creation
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    import fourier_amd
    from fourier_amd import Transform, shard

    key = args.config if args.config != "auto" else ("c5" if world > 1 else "c2")
    cfg = dict(CONFIGS[key])
    if args.dtype:
        cfg["dtype"] = args.dtype
    if args.log2n is not None:
        cfg["n"] = 1 << args.log2n
    dtype, n = cfg["dtype"], cfg["n"]
    cdt = torch.complex64 if dtype == "f32" else torch.complex128
    esz = 8 if dtype == "f32" else 16
    flops_per = nominal_flops(n)
    alg_bytes_per = 2.0 * n * esz  # each point read once + written once
    plan = make_plan(fourier_amd, n, dtype, local_rank, args)
    stream = torch.cuda.current_stream(dev).cuda_stream
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    hx = None
    extra = {}
    if key == "c5":
        # ---- strong scaling: the global batch is split over the ranks, a rank walks its shard in resident chunks
        gbatch = args.batch or cfg["global_batch"]
        chunk = args.chunk or cfg["chunk"]
        lo, hi = shard.batch_shard(gbatch, world, rank)
        chunk = max(1, min(chunk, hi - lo))
        x = torch.empty((chunk, n), dtype=cdt, device=dev)
        y = x if args.inplace else torch.empty_like(x)
        if args.inplace:
            plan.reserve(chunk, in_place=True)
        gen = torch.Generator(device=dev)
        batch = chunk  # per launch (roofline leg)
        units_per_step = gbatch

        def step(step_index):
            """One pass over this rank's shard; returns the FFT-only seconds (generation bracketed out)."""
            t_fft = 0.0
            for b0 in range(lo, hi, chunk):
                nb = min(chunk, hi - b0)
                gen.manual_seed(0x5EED0C50 + 7919 * step_index + b0)  # every chunk of every step is new data
                torch.view_as_real(x[:nb]).uniform_(0.0, 1.0, generator=gen)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), nb, int(Transform.Fft), stream)
                torch.cuda.synchronize(dev)
                t_fft += time.perf_counter() - t1
            return t_fft

        # ---- in-run single-GPU reference: rank 0 alone times two full-size chunks (after one warm chunk) while every
        # other rank waits at the barrier below, so that the line holds what ONE GPU does per chunk with an idle node
        # around it.  ideal step time = reference x chunks of the largest shard; efficiency = ideal / measured.
        ref_ms_per_chunk = None
        if dist is not None:
            dist.barrier()
        if rank == 0:
            ts_ref = []
            for i in range(3):
                gen.manual_seed(0x5EED0AAA + i)
                torch.view_as_real(x).uniform_(0.0, 1.0, generator=gen)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), chunk, int(Transform.Fft), stream)
                torch.cuda.synchronize(dev)
                if i:
                    ts_ref.append(time.perf_counter() - t1)
            ref_ms_per_chunk = sum(ts_ref) / len(ts_ref) * 1e3
        if dist is not None:
            dist.barrier()  # nobody starts its warm-up while rank 0 is still timing the reference
        for w in range(args.warmup):
            step(-1 - w)
        sync_all()
        t0 = time.perf_counter()
        fft_s = 0.0
        for k in range(args.steps):
            fft_s += step(k)
        torch.cuda.synchronize(dev)
        sync_all()
        wall = time.perf_counter() - t0
        elapsed = shard.reduce_max_seconds(fft_s, dist, red_dev)
        per_rank = shard.gather_seconds(fft_s, dist, red_dev)
        wall = shard.reduce_max_seconds(wall, dist, red_dev)
        extra["wall_ms_per_step_incl_regen"] = round(wall / args.steps * 1e3, 3)
        if rank == 0:
            shards = [shard.batch_shard(gbatch, world, r) for r in range(world)]
            largest = max(h - l for l, h in shards)
            ideal_ms = ref_ms_per_chunk * largest / chunk
            extra["strong_scaling"] = {
                "per_rank_fft_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank],
                "transforms_per_rank": [h - l for l, h in shards],
                "single_gpu_reference": {"ms_per_chunk": round(ref_ms_per_chunk, 3), "chunk": chunk,
                                         "how": "rank 0 alone, two timed chunks after one warm chunk, every other rank waiting at a barrier until it is done"},
                "ideal_ms_per_step": round(ideal_ms, 3),  # reference x (largest shard / chunk): perfect batch-shard scaling
                "efficiency_vs_reference": round(ideal_ms / (elapsed / args.steps * 1e3), 4),
            }
        workload = (f"batched 1D c2c {dtype} N={n} GLOBAL batch={gbatch} batch-sharded over {world} GPU(s) "
                    f"({hi - lo} per GPU as resident chunks of {chunk}, regenerated on the device), forward "
                    f"{'in-place' if args.inplace else 'out-of-place'} ({cfg['name']})")
        config = {"workload": workload, "n": n, "global_batch": gbatch, "batch_per_gpu": hi - lo, "chunk": chunk,
                  "parallelism": f"batch-shard x{world}", "plan": plan.describe()}
        scaling = "strong"
        if rank == 0 and not args.no_cpu:
            hx = None  # parity for c5: two transforms of the resident chunk, below
    else:
        batch = args.batch or cfg["batch"]
        units_per_step = world * batch
        # synthetic input: re, im i.i.d. uniform [0,1) (the reference bench recipe, fft_bench.rs:18-23)
        torch.manual_seed(0x5EED0001 + rank)
        x = torch.empty((batch, n), dtype=cdt, device=dev)
        torch.view_as_real(x).uniform_(0.0, 1.0)
        y = x if args.inplace else torch.empty_like(x)
        if args.inplace:
            plan.reserve(batch, in_place=True)

        def step():
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(Transform.Fft), stream)

        # host copy of the parity / CPU-baseline sample, taken before anything can overwrite the input
        if rank == 0 and world == 1 and not args.no_cpu:
            # ~20-30 s of CPU work with the AVX clone (2.8 ms per transform on one core of the round-2 box, six thread counts)
            sample = min(batch, args.cpu_sample or 1024)
            hx = x[:sample].cpu().numpy()
        for _ in range(args.warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        sync_all()
        mine = time.perf_counter() - t0
        elapsed = shard.reduce_max_seconds(mine, dist, red_dev)
        if dist is not None:
            per_rank = shard.gather_seconds(mine, dist, red_dev)
            extra["weak_scaling"] = {"per_rank_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank]}
        workload = (f"batched 1D c2c {dtype} N={n} batch={batch}/GPU forward "
                    f"{'in-place' if args.inplace else 'out-of-place'} ({cfg['name']})")
        config = {"workload": workload, "n": n, "batch_per_gpu": batch, "global_batch": world * batch,
                  "parallelism": f"batch-shard x{world}", "plan": plan.describe()}
        scaling = "weak"

    total_units = units_per_step * args.steps
    gflops = total_units * flops_per / elapsed / 1e9
    alg_gbps = total_units * alg_bytes_per / elapsed / 1e9
    ms_per_step = elapsed / args.steps * 1e3
    log2n = math.log2(n)
    nlabel = f"2^{int(log2n)}" if log2n == int(log2n) else str(n)

    out = {
        "metric": f"batched 1D c2c FFT GFLOP/s (5N*log2N), {dtype} N={nlabel}",
        "value": round(gflops, 1),
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": config,
        "config_key": key,  # c2 (weak, per-GPU batch fixed) or c5 (strong, global batch fixed): a curve must not mix them
        "process_group": None if dist is None else {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                                    "collectives": "timing scalars only (max, all-gather of one double per rank)"},
        "hbm_gbps_algorithmic": round(alg_gbps, 1),
        "hbm_frac_algorithmic": round(alg_gbps / (HBM_PEAK_GBPS * world), 4),
    }
    out.update(extra)

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events on the launch stream, live
        kernels = kernel_profile(plan, x.data_ptr(), y.data_ptr(), batch, stream)
        ceiling = skeleton = None
        if x.data_ptr() != y.data_ptr() and not args.no_ceiling:  # same box, same run, same buffers: plain device copies and the passes' skeletons x -> y (y is rewritten below)
            try:
                ceiling = copy_ceiling(x.data_ptr(), y.data_ptr(), x.numel() * x.element_size(), stream)
                torch.cuda.synchronize(dev)
            except Exception:
                ceiling = None
            try:
                skeleton = skeleton_ceiling(x.data_ptr(), y.data_ptr(), x.numel() * x.element_size(), stream)
                torch.cuda.synchronize(dev)
            except Exception:
                skeleton = None
        out["roofline"] = roofline_of(plan, kernels, batch, alg_bytes_per, dtype, alg_gbps / world / HBM_PEAK_GBPS,
                                      traffic_ok=(key == "c2" and n == 1 << 20 and dtype == "f32" and batch == 4096), ceiling=ceiling,
                                      skeleton=skeleton)

        if key == "c5" and not args.no_cpu:
            # parity on the resident chunk: its first and last transform against the oracle (bounded: 2 transforms)
            import numpy as np
            from oracle import oracle as O

            O.build()
            torch.view_as_real(x).uniform_(0.0, 1.0)
            keep = [x[0].cpu().numpy(), x[batch - 1].cpu().numpy()]
            plan.transform_batch_ptr(x.data_ptr(), y.data_ptr(), batch, int(Transform.Fft), stream)
            torch.cuda.synchronize(dev)
            got = [y[0].cpu().numpy(), y[batch - 1].cpu().numpy()]
            orc = O.OracleFft(n, np.complex64 if dtype == "f32" else np.complex128)
            t0 = time.perf_counter()
            refs = [orc.transform(k, O.FFT) for k in keep]
            one_core_s = (time.perf_counter() - t0) / len(keep)
            errs = [float(np.linalg.norm(g.astype(np.complex128) - r) / np.linalg.norm(r)) for g, r in zip(got, refs)]
            out["parity"] = {"sample_transforms": 2, "rel_l2_vs_oracle": max(errs), "tolerance": 1e-6 if dtype == "f32" else 5e-14}
            if world == 1:
                out["cpu_baseline"] = {"value": round(flops_per / one_core_s / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
                                       "sample": f"2 of the same transforms ({dtype} N={n}) on one core, {one_core_s * 1e3:.1f} ms each"}
        elif hx is not None:
            out["parity"], out["cpu_baseline"] = cpu_baseline(torch, plan, hx, n, dtype, dev, stream, cores)

        if world == 1 and dist is None and args.config == "auto" and not args.no_other and args.batch is None and args.log2n is None:
            # ---- the other BASELINE configs, a few seconds each, so that a driver-run record holds them too
            del x, y
            torch.cuda.empty_cache()
            others = {}
            try:
                others["c1"] = config_c1(fourier_amd)
            except Exception as e:
                others["c1"] = {"error": repr(e)}
            for k in ("c3", "c4", "c5"):
                try:
                    others[k] = quick_config(fourier_amd, torch, dev, k)
                except Exception as e:  # never lose the headline line to a secondary config
                    others[k] = {"error": repr(e)}
                    torch.cuda.empty_cache()
            try:
                torch.cuda.empty_cache()
                grid = reference_bench_sizes(fourier_amd, torch, dev)
                others["reference_bench_sizes"] = grid  # forward + inverse, f32 + f64 (fft_bench.rs:153-159)
                others["reference_bench_sizes_f32"] = [r for r in grid if r["dtype"] == "f32" and r["direction"] == "forward"]
            except Exception as e:
                others["reference_bench_sizes"] = {"error": repr(e)}
            out["other_configs"] = others
            # The driver's record keeps the scalar members of `roofline` and `config` and the tail of stdout: every BASELINE
            # config goes there as flat scalars -- ms per step, whole-path algorithmic fraction of
            # 8 TB/s, dominant kernel's fraction --, and the 60-row size grid goes to stderr / --details, not into the line.
            for k in ("c3", "c4", "c5"):
                o = others.get(k, {})
                for dst in (out["roofline"], out["config"]):
                    dst[f"{k}_ms_per_step"] = o.get("ms_per_step")  # median over fresh allocations (quick_config)
                    dst[f"{k}_ms_min"] = o.get("ms_min")
                    dst[f"{k}_ms_max"] = o.get("ms_max")
                    dst[f"{k}_whole_path_frac"] = o.get("hbm_frac_algorithmic")
                    dst[f"{k}_dominant_kernel_frac"] = o.get("dominant_kernel_frac")
            out["roofline"]["c1_gpu_us_per_call"] = others.get("c1", {}).get("gpu_us_per_call")
        emit(out, args)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
