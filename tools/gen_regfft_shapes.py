#!/usr/bin/env python3
"""Writes fourier_amd/csrc/regfft_shapes.h: the lengths that run as a direct transform on register stages (kernels_regfft.h) and the split
R1 x R2 (x R3) of each.

Candidates: every length of 14 ... 10240 points (a transform, in f32 a pair, stays within a compute unit's LDS) whose prime factors stop at 13 and that is not 2^a 3^b (those keep the reference's own
schedule, autosort/mod.rs:24-46), where it splits into two stages of at most 32 points -- the most balanced split: stage A runs on R2 of a
lane group's R1 lanes -- or, failing that, into three (descending, the smallest largest stage; stages of 33 ... 40 points where nothing shorter exists;
49-point stages spill to 0.12 ... 0.2 of the peak, profiles/r06_s50_*; at most 1024 lanes per stage).  Without --ab every candidate is adopted in both precisions (the A/B build); with --ab FILE (rows of
tools/gpu_r06_regfft_ab.py: real, n, arm = registers | before, ms) a length is adopted in a precision where the register kernel is at least
MARGIN faster than the route it had (median of 7, alternating on shared buffers).  --split-ab FILE (arms plain | split | fact | splitfact | before): a three-stage
length takes the fastest variant (whole / split-plane exchanges x whole / factored twiddle tables) where that is VARIANT_MARGIN faster than the
plain one, and is judged against the route it had with it.  --ab-build: every candidate, all four variants.  --unpaired-ab FILE (f32, arms listed | unpaired | unpairedfact | before): one transform per
workgroup where that is VARIANT_MARGIN faster than the listed variant; --unpaired-build: the listed variant and both unpaired ones.  The emulator build keeps the lengths its test names.

    python tools/gen_regfft_shapes.py [--ab profiles/r06_s49_regfft_ab.jsonl ...] [--split-ab profiles/r06_s53_regfft_split_ab.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "fourier_amd", "csrc", "regfft_shapes.h")
NMIN, NMAX, RMAX, RMAX_LONG = 14, 10240, 32, 40
NMAX_F32 = 20480  # N x 8 bytes of LDS: f32 with one transform per workgroup (the unpaired variants), f64 on split-plane exchanges
MARGIN = 1.04
VARIANT_MARGIN = 1.03
EMU_OPT = {729, 1536, 4608}
EMU = {22, 77, 143, 175, 200, 245, 350, 385, 400, 560, 700, 800, 1001, 2000, 2002, 2904, 4000, 5005, 8000, 8960, 9009, 12000}


def smooth(n, primes):
    for p in primes:
        while n % p == 0:
            n //= p
    return n == 1


def split2(n, rmax):
    best = None
    for r2 in range(2, int(n ** 0.5) + 1):
        if n % r2 == 0 and n // r2 <= rmax:
            best = (n // r2, r2, 0)
    return best


def split3(n, rmax):
    best = None
    for a in range(2, rmax + 1):
        if n % a:
            continue
        for b in range(2, a + 1):
            if (n // a) % b:
                continue
            c = n // a // b
            if 2 <= c <= b and a * b <= 1024 and (best is None or a < best[0]):
                best = (a, b, c)
    return best


def candidates():
    rows = []
    for n in range(NMIN, NMAX_F32 + 1):
        if not smooth(n, (2, 3, 5, 7, 11, 13)) or smooth(n, (2, 3)):
            continue
        s = split2(n, RMAX) or split3(n, RMAX)
        if s is None:
            s = split3(n, RMAX_LONG)
        if s:
            rows.append((n, s, 1, 1))  # (beyond NMAX: f32 unpaired, f64 on split-plane exchanges only -- N x 8 bytes of LDS)
    return rows


def optin_candidates():
    """2^a 3^b lengths that are not powers of two: on request only (plan option "register_stages"; by default they keep the reference's schedule).
    Up to NMAX the plain kernel; beyond, the variants that fit the LDS (f32 unpaired with factored tables, f64 split planes with factored tables)."""
    rows = []
    for n in range(NMIN, NMAX_F32 + 1):
        if not smooth(n, (2, 3)) or n & (n - 1) == 0:
            continue
        s = split2(n, RMAX) or split3(n, RMAX) or split3(n, RMAX_LONG)
        if s:
            rows.append((n, s, 1 if n <= NMAX else 6, 1 if n <= NMAX else 4))
    return rows


def read_ab(files):
    ab = {}
    for f in files:  # a later file replaces an earlier one's rows of a length
        rows_of = {}
        for line in open(f):
            r = json.loads(line)
            rows_of.setdefault((r["real"], r["n"]), {})[r["arm"]] = r
        for key, arms in rows_of.items():
            # (a "before" arm that found the length's run-time kernel in a code-object cache left behind on the box is not the default route:
            # f32 5005 in session 50, whose box had run the session's first attempt)
            if "before" in arms and "specialised" in arms["before"]["plan"]:
                continue
            ab[key] = {arm: r["ms"] for arm, r in arms.items()}
    return ab


def main(argv):
    files = [argv[i + 1] for i, a in enumerate(argv) if a == "--ab"]
    split_files = [argv[i + 1] for i, a in enumerate(argv) if a == "--split-ab"]
    ab, sab = read_ab(files), read_ab(split_files)
    up = read_ab([argv[i + 1] for i, a in enumerate(argv) if a == "--unpaired-ab"])
    rows, kept = [], {"f32": 0, "f64": 0, "split": 0}
    for n, s, f32, f64 in candidates():
        if "--ab-build" in argv:  # every candidate, both exchange variants of the three-stage ones
            f32, f64 = (9 if s[2] else 1) if n <= NMAX else 0, (9 if s[2] else 1) if f64 else 0
        elif files:
            def flag(real):
                t = dict(ab.get((real, n), {}))
                v = 1
                if s[2] and (real, n) in sab:  # three stages: arms plain / split / fact / splitfact / before of the variant sessions
                    u = sab[(real, n)]
                    arms = [(u[a], i + 1) for i, a in enumerate(("plain", "split", "fact", "splitfact")) if a in u]
                    if arms:
                        best, v = min(arms)
                        if "plain" in u and u["plain"] < VARIANT_MARGIN * best:
                            best, v = u["plain"], 1
                        if "before" in u:
                            t = {"registers": best, "before": u["before"]}
                return v if "registers" in t and "before" in t and t["before"] >= MARGIN * t["registers"] else 0
            f32, f64 = (f32 and flag("f32")) if n <= NMAX else 0, f64 and flag("f64")
        if up:  # f32, three stages: the unpaired variants against the listed one (arms listed / unpaired / unpairedfact / before)
            u = up.get(("f32", n), {})
            if s[2] and "before" in u:
                cur = u.get("listed") if f32 else None
                arms = [(u[a], v) for a, v in (("unpaired", 5), ("unpairedfact", 6)) if a in u]
                if arms:
                    best, v = min(arms)
                    if cur is None or cur >= VARIANT_MARGIN * best:
                        f32 = v if u["before"] >= MARGIN * best else f32
        if "--long-f64-build" in argv and s[2] and n > NMAX:  # f64 beyond NMAX: the split-plane variants are the ones that fit the LDS
            f64 = 9
        if "--unpaired-build" in argv and s[2]:  # the listed variant (or the plain kernel) and the two unpaired ones
            f32 = 10 + (f32 or 1)
        if f32 or f64:
            rows.append((n, s, f32, f64))
            kept["f32"] += bool(f32)
            kept["f64"] += bool(f64)
            kept["split"] += (f32 > 1) + (f64 > 1)
    opt = read_ab([argv[i + 1] for i, a in enumerate(argv) if a == "--optin-ab"])
    opt_rows = []
    for n, s3, f32, f64 in optin_candidates():
        if "--optin-build" not in argv:
            def won(real):
                t = opt.get((real, n), {})
                return "registers" in t and "before" in t and t["before"] >= MARGIN * t["registers"]
            f32, f64 = f32 if won("f32") else 0, f64 if won("f64") else 0
        if f32 or f64:
            opt_rows.append((n, s3, f32, f64))
    with open(OUT, "w") as f:
        f.write("// regfft_shapes.h -- GENERATED by tools/gen_regfft_shapes.py" + "".join(" --ab " + os.path.relpath(x, ROOT) for x in files) + "".join(" --split-ab " + os.path.relpath(x, ROOT) for x in split_files) + "".join(" --unpaired-ab " + os.path.relpath(argv[i + 1], ROOT) for i, a in enumerate(argv) if a == "--unpaired-ab") + (" --ab-build" if "--ab-build" in argv else "") + (" --unpaired-build" if "--unpaired-build" in argv else "") + (" --long-f64-build" if "--long-f64-build" in argv else "") + "".join(" --optin-ab " + os.path.relpath(argv[i + 1], ROOT) for i, a in enumerate(argv) if a == "--optin-ab") + (" --optin-build" if "--optin-build" in argv else "") + "\n")
        f.write("// the lengths of kernels_regfft.h: FOURIER_REGFFT_ROW(N, R1, R2, R3 (0: two stages), f32, f64, in the emulator build); a precision's flag:\n")
        f.write("// 0 = not adopted, 1 = adopted; three stages: 2 = split-plane exchanges, 3 = factored twiddle tables, 4 = both, f32 5 / 6 = one transform per\n")
        f.write("// workgroup (unpaired) without / with factored tables; A/B builds: 9 = 1 ... 4 built, 10 + F = the listed F and 5 / 6 built\n")
        f.write(f"// {len(rows)} lengths: {kept['f32']} in f32, {kept['f64']} in f64" + (f" (at least {MARGIN:.2f} x the route they had)" if files else " (every candidate: the A/B build)") + "\n")
        for n, (r1, r2, r3), f32, f64 in rows:
            assert r1 * r2 * (r3 or 1) == n
            f.write(f"FOURIER_REGFFT_ROW({n}, {r1}, {r2}, {r3}, {f32}, {f64}, {int(n in EMU)})\n")
        f.write(f"// on request (plan option \"register_stages\"): {len(opt_rows)} lengths 2^a 3^b, {sum(bool(r[2]) for r in opt_rows)} in f32, {sum(bool(r[3]) for r in opt_rows)} in f64\n")
        for n, (r1, r2, r3), f32, f64 in opt_rows:
            assert r1 * r2 * (r3 or 1) == n
            f.write(f"FOURIER_REGFFT_OPT_ROW({n}, {r1}, {r2}, {r3}, {f32}, {f64}, {int(n in EMU_OPT)})\n")
    print(f"{OUT}: {len(rows)} lengths, f32 {kept['f32']}, f64 {kept['f64']}, split {kept['split']}")


if __name__ == "__main__":
    main(sys.argv[1:])
