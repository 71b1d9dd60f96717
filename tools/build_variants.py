#!/usr/bin/env python3
"""Builds A/B variants of libfourier.so (different compile-time knobs) into fourier_amd/lib/variants/
so one GPU session can time them side by side (tools/gpu_sweep.py).  Development tool only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fourier_amd import build as B  # noqa: E402

VARIANTS = {
    "base": [],
    "slp": ["-fslp-vectorize"],
    "waves1": ["-DFOURIER_MIN_WAVES(NT)=1"],
    "small_mw4": ["-DFOURIER_MIN_WAVES(NT)=((NT)>=1024?4:((NT)>=256?(NT)/128:4))"],
    "small_mw5": ["-DFOURIER_MIN_WAVES(NT)=((NT)>=1024?4:((NT)>=256?(NT)/128:5))"],
    "mid_mw3": ["-DFOURIER_MIN_WAVES(NT)=((NT)>=1024?4:((NT)>=512?4:((NT)>=256?3:1)))"],
    "nt_first": ["-DFOURIER_NT_LOAD=1"],
    "nt_store_all": ["-DFOURIER_NT_STORE=2"],
    "nt_none": ["-DFOURIER_NT_LOAD=0", "-DFOURIER_NT_STORE=0"],
    "conv_1wg": ["-DFOURIER_CONV_MIN_WAVES(NT)=((NT)>=512?2:1)"],
    "nt_load": ["-DFOURIER_NT_LOAD=1"],
    "nt_store": ["-DFOURIER_NT_STORE=1"],
    "nt_both": ["-DFOURIER_NT_LOAD=1", "-DFOURIER_NT_STORE=1"],
    "cg4": ["-DFOURIER_CG_1024=4"],
    "cg2048_4": ["-DFOURIER_CG_2048=4"],
    "mix_pingpong_40k": ["-DFOURIER_MIX_INPLACE_BYTES=(40u*1024u)"],
    "mix64_pairs_all": ["-DFOURIER_MIX_PAIR_MIN_N_F64=0u"],
    "mix64_pairs_none": ["-DFOURIER_MIX_PAIR_MIN_N_F64=100000u"],
    "cg16": ["-DFOURIER_CG_1024=16"],
    "split8k": ["-DFOURIER_SPLIT_THRESHOLD=(8*1024)"],
    "split4k": ["-DFOURIER_SPLIT_THRESHOLD=(4*1024)"],
    "split16k": ["-DFOURIER_SPLIT_THRESHOLD=(16*1024)"],
    "split32k": ["-DFOURIER_SPLIT_THRESHOLD=(32*1024)"],
    "split0": ["-DFOURIER_SPLIT_THRESHOLD=0"],
    "fused_mw3": ["-DFOURIER_FUSED_MIN_WAVES=3"],
    "conv_cg4": ["-DFOURIER_CONV_CG_1024=4"],
    "conv_cg4_mw3": ["-DFOURIER_CONV_CG_1024=4", "-DFOURIER_CONV_MIN_WAVES(NT)=((NT)==256?3:FOURIER_MIN_WAVES(NT))"],
    "conv_cg4_mw4": ["-DFOURIER_CONV_CG_1024=4", "-DFOURIER_CONV_MIN_WAVES(NT)=((NT)==256?4:FOURIER_MIN_WAVES(NT))"],
    "conv_stnt": ["-DFOURIER_CONV_ST_NT=1"],
    "conv_wnt": ["-DFOURIER_CONV_W_NT=1"],
    "conv_both": ["-DFOURIER_CONV_ST_NT=1", "-DFOURIER_CONV_W_NT=1"],
    "cg4096_4": ["-DFOURIER_CG_4096=4"],
    "split_nt": ["-DFOURIER_SPLIT_LD=POL_NT"],
    "tl_store_soff": ["-DFOURIER_TWOLEVEL_STORE_SOFF=1"],
    "no_chirp": ["-DFOURIER_AB_NO_CHIRP=1"],
    "no_w": ["-DFOURIER_AB_NO_W=1"],
    "stage_tw0": ["-DFOURIER_STAGE_TW_BATCH=0"],
    "stage_tw4": ["-DFOURIER_STAGE_TW_BATCH=4"],
    "blu_out_nt": ["-DFOURIER_BLU_OUT_ST_NT=1"],
    "nt_narrow2048_off": ["-DFOURIER_NT_STORE_NARROW_2048=0"],
    "conv_wb4": ["-DFOURIER_CONV_W_BATCH=4"],
    "conv_wb16": ["-DFOURIER_CONV_W_BATCH=16"],
    "mix_ref_order": ["-DFOURIER_MIX_PRIMES_FIRST=0"],
    "mix_pair5_none": ["-DFOURIER_MIX_PAIR5_MIN_N_F32=1000000u", "-DFOURIER_MIX_PAIR5_MIN_N_F64=1000000u"],
    "mix_pair5_all": ["-DFOURIER_MIX_PAIR5_MIN_N_F32=25u", "-DFOURIER_MIX_PAIR5_MIN_N_F64=25u"],
    "mix_wide_16k": ["-DFOURIER_MIX_WIDE_MIN_BYTES=16384u"],
    "rows_unstaged": ["-DFOURIER_ROWS_STAGED=0"],
    "mix_mid512": ["-DFOURIER_MIX_MID_THREADS=512u", "-DFOURIER_MIX_MID_MIN_BYTES=16384u"],
    "mix_half_none": ["-DFOURIER_MIX_HALF_MAX_ITEMS=0u"],
    "mix_half_250": ["-DFOURIER_MIX_HALF_MAX_ITEMS=250u"],
    "mix_wide_none": ["-DFOURIER_MIX_WIDE_MIN_BYTES=0xffffffffu", "-DFOURIER_MIX_WIDE_MIN_N=0xffffffffu"],
    "row_stores_16b": ["-DFOURIER_PAIRED_ROW_STORES=1"],
    "mix_gio_off": ["-DFOURIER_MIX_GIO_MIN_RUN=0u"],
    "mix_gio_all": ["-DFOURIER_MIX_GIO_ALL=1"],
    "mix_twlds": ["-DFOURIER_MIX_TW_LDS=1"],
    "mix_swizzle_off": ["-DFOURIER_MIX_SWIZZLE=0"],
    "mix_loads_per_round": ["-DFOURIER_MIX_LOADS_FIRST=0"],
    "mix_loads_first_all": ["-DFOURIER_MIX_LOADS_FIRST=2"],
    "tabs_before_loads": ["-DFOURIER_TABS_AFTER_LOADS=0"],
    "rows128_cg16": ["-DFOURIER_CG_128_ROWS=16"],
    "rows_staged_per_half": ["-DFOURIER_ROWS_STAGED_LOADS_FIRST=0"],
    "mix_copy_loop": ["-DFOURIER_MIX_COPY_BATCHED=0"],
    "mix_slp": ["-fslp-vectorize", "-DFOURIER_MIX_SLP_BUILD=1"],  # packed f32 VALU ops in the LDS mixed-radix / mixed-tile kernels only
    "pf_nobar": ["-DFOURIER_PF_BARRIER_AFTER_WAIT=0"],
    "pf_vm0": ["-DFOURIER_PF_WAIT_ALL=1"],
    "pf_vm0_plainst": ["-DFOURIER_PF_WAIT_ALL=1", "-DFOURIER_PF_ST_PLAIN=1"],
    "pf_plainst": ["-DFOURIER_PF_ST_PLAIN=1"],
    "pf_stores_first": ["-DFOURIER_PF_STORES_FIRST=1"],
    "pf_vm0_dma4": ["-DFOURIER_PF_WAIT_ALL=1", "-DFOURIER_PF_DMA_ROWS=4"],
    "pf_vm0_dma12": ["-DFOURIER_PF_WAIT_ALL=1", "-DFOURIER_PF_DMA_ROWS=12"],
    "abl1": ["-DFOURIER_ABLATE=1"],
    "abl2": ["-DFOURIER_ABLATE=2"],
    "abl3": ["-DFOURIER_ABLATE=3"],
    "blu_prune_off": ["-DFOURIER_BLU_PRUNE=0"],
    "abl4": ["-DFOURIER_ABLATE=4"],  # stage twiddles from a constant (no table loads inside the in-tile transform)
    "abl5": ["-DFOURIER_ABLATE=5"],
    "onelaunch_scalar": ["-DFOURIER_ONELAUNCH_PK=0"],  # round 6: the one-launch kernels (2^11..2^15, chirp-z M <= 2^15) without packed f32 arithmetic
    "ld_last_sc1": ["-DFOURIER_NT_LOAD=3"],  # round 6, stream pipeline: pass-1 loads sc1 / intermediate stores plain
    "st_mid_plain_ld_last_sc1": ["-DFOURIER_NT_LOAD=3", "-DFOURIER_NT_STORE=1"],  # per-thread inter-pass twiddle factor from a constant (no two-level look-up)
}


def groups_of(flags):
    """Translation-unit groups (fourier_amd/build.py) a variant's flags can reach: the LDS mixed-radix knobs touch only the
    'mixed' objects, everything else only the tile / one-launch / small kernels; the other objects come from the base build."""
    mixed = any("MIX" in f for f in flags)
    other = any("MIX" not in f and not f.startswith("-f") for f in flags)  # a code-generation flag follows the knobs it comes with
    g = set()
    if mixed:
        g |= {"mixed"}
    if other or not mixed:
        g |= {"pass", "onelaunch", "misc"}
    return g


def main(names):
    outdir = os.path.join(ROOT, "fourier_amd", "lib", "variants")
    os.makedirs(outdir, exist_ok=True)
    base_objs, _ = B.compile_objects(B.OBJDIR, (), False, groups=set(B.group_of().values()) - B.EXPERIMENTS_ONLY)  # the shared objects
    group = B.group_of()
    for name in names:
        flags = VARIANTS[name]
        saved = list(B.CFLAGS)
        if name == "slp":
            B.CFLAGS[:] = [f for f in B.CFLAGS if f != "-fno-slp-vectorize"]
        try:
            objdir = os.path.join(outdir, "obj_" + name)
            groups = groups_of(flags)
            objs, _ = B.compile_objects(objdir, flags, False, groups=groups)
            members = {n: (objs[n] if group[n] in groups else base_objs[n]) for n in objs}
            out = os.path.join(outdir, f"libfourier_{name}.so")
            B.link(members, out, experiments=False, soname=False)
            print(name, "ok", sorted(groups))
        except Exception as e:  # noqa: BLE001
            print(name, f"FAILED {e!r}")
        finally:
            B.CFLAGS[:] = saved


if __name__ == "__main__":
    main(sys.argv[1:] or list(VARIANTS))
