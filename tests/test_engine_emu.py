"""Host logic + kernel logic of the engine, exercised WITHOUT a GPU.

The engine sources (fourier_amd/csrc) are compiled against tests/emu/hipemu.h, a test-only fiber
emulation of the HIP subset they use, and driven through the same C ABI / Python operator layer as
the product.  This validates plan construction, pass scheduling, kernel index arithmetic, LDS layouts,
the legacy host ABI and error behaviour before any GPU minute is spent; the `-m gpu` tests repeat the
parity checks on the real library.  The emulation is never used by the product path.
"""
import ctypes
import os

import numpy as np
import pytest

from helpers import GOLDEN, hash_normal, load_ref10, naive_dft, near, regfft_shape, rel_l2, max_rel

F32_EPS, F64_EPS = 1e-4, 1e-11  # fourier/tests/integrity.rs:92,120


@pytest.fixture(scope="module")
def fa():
    from emu import build_emu
    from fourier_amd import _lib

    prev = _lib._lib
    _lib._lib = build_emu.load()  # test-side monkeypatch: route the operator layer to the emulation build
    import fourier_amd

    yield fourier_amd
    _lib._lib = prev


def make(fa, n, dtype):
    return fa.create_fft_f32(n) if np.dtype(dtype) == np.complex64 else fa.create_fft_f64(n)


def run_batch(plan, x, code, inplace=False):
    x = np.ascontiguousarray(x)
    y = x.copy() if inplace else np.empty_like(x)
    src = y if inplace else x
    plan.transform_batch_ptr(src.ctypes.data, y.ctypes.data, x.shape[0], int(code))
    return y


@pytest.mark.parametrize("dtype,eps", [(np.complex64, F32_EPS), (np.complex128, F64_EPS)])
@pytest.mark.parametrize("forward", [True, False])
def test_sweep_1_255_like_reference(fa, oracle, dtype, eps, forward):
    """integrity.rs:145-192 procedure through the engine: every size 1..255 (tiny, row kernels,
    Bluestein), vs the naive DFT at the reference tolerance, vs numpy-f64 and vs the oracle."""
    g = np.load(os.path.join(GOLDEN, "sweep_1_255.npz"))
    x = (g["x_fwd"] if forward else g["x_inv"]).astype(dtype)
    y64 = g["y_fwd"] if forward else g["y_inv"]
    code = fa.Transform.Fft if forward else fa.Transform.Ifft
    off = 0
    for n in range(1, 256):
        plan = make(fa, n, dtype)
        got = np.empty(n, dtype)
        plan.transform(np.ascontiguousarray(x[:n]), got, code)  # legacy host ABI, out of place
        ok, worst = near(naive_dft(x[:n], inverse=not forward), got, eps)
        assert ok, (n, worst)
        want = y64[off:off + n]
        off += n
        scale = max(np.abs(want).max(), 1.0)
        tol = (3e-6 if dtype == np.complex64 else 1e-12) * scale
        assert np.abs(got - want).max() <= tol, (n, float(np.abs(got - want).max()))
        orc = oracle.OracleFft(n, dtype).transform(x[:n], int(code))
        assert np.abs(got - orc).max() <= 2 * tol, n


@pytest.mark.parametrize("dtype,eps", [(np.complex64, F32_EPS), (np.complex128, F64_EPS)])
def test_reference_golden_vector_through_engine(fa, dtype, eps):
    x, y = load_ref10()  # integrity.rs:48-72; N=10: Bluestein (M=32) in the reference, radix 2.5 here
    plan = make(fa, 10, dtype)
    got = np.empty(10, dtype)
    plan.fft(x.astype(dtype), got)
    ok, worst = near(got, y, eps)
    assert ok, worst
    plan.ifft(y.astype(dtype), got)
    ok, worst = near(got, x, eps)
    assert ok, worst


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_ffi_impulse_roundtrip(fa, dtype):
    # fourier-ffi/test.c:7-39 through the legacy C ABI
    plan = make(fa, 4, dtype)
    x = np.array([1, 0, 0, 0], dtype=dtype)
    out = np.empty_like(x)
    plan.transform(x, out, fa.Transform.Fft)
    assert np.allclose(out, 1)
    plan.transform_in_place(out, fa.Transform.Ifft)
    assert np.abs(out - x).max() <= 1e-10


@pytest.mark.parametrize("n", [16, 64, 512, 2048, 4096, 8192, 1 << 14, 1 << 15, 1 << 16])
@pytest.mark.parametrize("dtype,tl2,tmax", [(np.complex64, 1e-6, 2e-6), (np.complex128, 5e-14, 1e-13)])
def test_pow2_sizes_all_codes_vs_oracle(fa, oracle, n, dtype, tl2, tmax):
    plan = make(fa, n, dtype)
    x = np.stack([hash_normal(900 + b, n) for b in range(3)]).astype(dtype)
    for code in range(5):
        ref = oracle.transform_batch(x, code)
        for inplace in (False, True):
            got = run_batch(plan, x, code, inplace)
            assert rel_l2(got, ref) <= tl2 and max_rel(got, ref) <= tmax, (n, code, inplace, rel_l2(got, ref))


def test_n4096_against_committed_spectrum(fa):
    g = np.load(os.path.join(GOLDEN, "n4096.npz"))
    x = hash_normal(int(g["seed"]), 4096)
    for dtype, tl2 in ((np.complex64, 1e-6), (np.complex128, 5e-14)):
        got = run_batch(make(fa, 4096, dtype), x.astype(dtype)[None, :], 0)[0]
        assert rel_l2(got, g["y"]) <= tl2


@pytest.mark.parametrize("n", [7, 17, 96, 100, 1000, 1003, 1025, 2500])
def test_bluestein_and_mixed_radix_sizes(fa, oracle, n):
    x = np.stack([hash_normal(40 + b, n) for b in range(2)])
    for dtype, tl2 in ((np.complex64, 2e-6), (np.complex128, 2e-12)):
        plan = make(fa, n, dtype)
        for code in range(5):
            ref = oracle.transform_batch(x.astype(dtype), code)
            assert rel_l2(run_batch(plan, x.astype(dtype), code), ref) <= tl2, (n, code)
            assert rel_l2(run_batch(plan, x.astype(dtype), code, inplace=True), ref) <= tl2, (n, code)


def test_lengths_with_prime_factors_5_to_13_run_as_stockham_passes(fa, oracle):
    """Beyond the reference (which sends them to Bluestein, fourier/src/lib.rs:38-42): lengths whose prime factors stop at
    13 run the Stockham pass with the radix list continued [4,8,4,3,2,5,7,11,13] -- per-length kernels for the reference's
    own 5^k benchmark lengths and round decimal lengths, the runtime-parameterised kernel for the rest.  Every transform
    code, in and out of place, ragged batches; within the Bluestein tolerance of the oracle and tighter against f64 truth."""
    for n, batch in ((5, 300), (35, 70), (125, 19), (143, 9), (625, 5), (1000, 3), (1001, 3), (3125, 2), (4095, 2)):
        x = np.stack([hash_normal(500 + b, n) for b in range(batch)])
        for dtype, tl2, ttruth in ((np.complex64, 2e-6, 4e-7), (np.complex128, 2e-12, 2e-15)):
            plan = make(fa, n, dtype)
            if dtype == np.complex128 and n > 2048 and (n % 11 == 0 or n % 13 == 0):
                assert "bluestein" in plan.describe()  # f64 radix 11 / 13 beyond the 256-thread kernels: does not fit the registers
                continue
            assert plan.describe().startswith("stockham registers" if regfft_shape(n, dtype, emu=True) else "stockham mixed-radix"), plan.describe()
            xs = x.astype(dtype)
            for code in range(5):
                ref = oracle.transform_batch(xs, code)
                got = run_batch(plan, xs, code)
                assert rel_l2(got, ref) <= tl2, (n, code, rel_l2(got, ref))
                assert np.array_equal(run_batch(plan, xs, code, inplace=True), got), (n, code)
            assert rel_l2(run_batch(plan, xs, 0), np.fft.fft(xs.astype(np.complex128), axis=1)) <= ttruth, n
    assert "bluestein" in make(fa, 17 * 64, np.complex64).describe()      # a factor above 13
    assert "bluestein" in make(fa, 11 * 13 * 160, np.complex64).describe()  # beyond the 160 KiB of LDS, and no ahead-of-time tile length has a factor 11 / 13
    # 2^3*3*5^3*7 beyond the LDS: two column-tile passes whose lengths have factors 5 / 7 (round 5; Bluestein until then)
    plan = make(fa, 21000, np.complex64)
    assert "mixed tiles 150x140" in plan.describe(), plan.describe()
    xs = np.stack([hash_normal(400 + b, 21000) for b in range(2)]).astype(np.complex64)
    for code in (0, 4):
        assert rel_l2(run_batch(plan, xs, code), oracle.transform_batch(xs, code)) <= 2e-6, code
    assert "bluestein" in make(fa, 9100, np.complex64).describe()         # no per-length kernel, beyond the runtime kernel's 8192 points


MIXED_SIZES = sorted({(2 ** a) * (3 ** b) for a in range(13) for b in range(1, 8) if (2 ** a) * (3 ** b) <= 4096})


def test_mixed_radix_sizes_are_bit_identical_to_the_oracle(fa, oracle):
    """N = 2^a*3^b <= 18432 (f64: 9216) runs the reference's own radix schedule [4,8,4,3,2], tables and operation order
    (autosort/mod.rs:20-46,203-284, butterfly.rs:3-65) natively: every transform code, both precisions,
    in and out of place must equal the CPU restatement bit for bit."""
    for n in MIXED_SIZES[::3] + [3, 243, 3072, 3888]:
        x = np.stack([hash_normal(11 + b, n) for b in range(3)])
        for dtype in (np.complex64, np.complex128):
            plan = make(fa, n, dtype)
            assert "mixed-radix" in plan.describe()
            for code in range(5):
                ref = oracle.transform_batch(x.astype(dtype), code)
                assert np.array_equal(run_batch(plan, x.astype(dtype), code), ref), (n, dtype, code)
                assert np.array_equal(run_batch(plan, x.astype(dtype), code, inplace=True), ref), (n, dtype, code)
    assert "mixed-radix" in make(fa, 18432, np.complex64).describe()  # one in-place LDS buffer of 144 KiB
    assert "mixed-radix" in make(fa, 19683, np.complex64).describe()  # 3^9: 154 of the 160 KiB
    assert "mixed-radix" in make(fa, 9216, np.complex128).describe()
    assert "mixed tiles 108x96" in make(fa, 10368, np.complex128).describe()   # f64: above the LDS-resident limit (9216): two passes on column tiles
    assert "mixed tiles 128x96" in make(fa, 12288, np.complex64).describe()   # 3*2^12: two mixed-length tile passes, not the LDS kernel
    for n, dtype in ((6144, np.complex64), (9216, np.complex64), (18432, np.complex64), (13122, np.complex64),
                     (4608, np.complex128), (9216, np.complex128), (6561, np.complex128)):
        xb = np.stack([hash_normal(21 + b, n) for b in range(2)]).astype(dtype)
        assert np.array_equal(run_batch(make(fa, n, dtype), xb, 0), oracle.transform_batch(xb, 0)), n


def test_length_specialised_mixed_kernels_equal_the_generic_one(fa, monkeypatch):
    """Every LDS-resident 2^a*3^b length has a kernel instantiation with the schedule, strides and table offsets as
    compile-time constants; the runtime-parameterised kernel (FOURIER_MIX_GENERIC) must give the same bits."""
    for n in (3, 54, 243, 768, 2187, 4374, 6561):
        x = np.stack([hash_normal(31 + b, n) for b in range(5)]).astype(np.complex64)
        fast = make(fa, n, np.complex64)
        monkeypatch.setenv("FOURIER_MIX_GENERIC", "1")
        slow = make(fa, n, np.complex64)
        monkeypatch.delenv("FOURIER_MIX_GENERIC")
        for code in (0, 1, 4):
            assert np.array_equal(run_batch(fast, x, code), run_batch(slow, x, code)), (n, code)


def test_large_mixed_radix_sizes_run_natively(fa, oracle, monkeypatch):
    """N = 2^a*3^b (a >= 12, any b): big-radix passes over the 2^a part, then the odd part as radix-27 Stockham passes
    plus one of radix 3 / 9 / 27 -- twiddled middle passes and a final one (the reference's order, RADICES =
    [4,8,4,3,2]) -- or, where the length splits into two tile lengths of at most 576 (round 6, register tiles: measured faster; round 4: 384),
    two mixed-length tile passes; every code, in and out of place, against the oracle, on both routes."""
    cases = ((3 * 4096, "64x64x3", "mixed tiles 128x96"), (9 * 8192, "128x64x9", "mixed tiles 288x256"), (27 * 4096, "64x64x27", "mixed tiles 384x288"),
             (81 * 4096, "64x64x27x3", "mixed tiles 576x576"), (243 * 4096, "64x64x27x9", None))
    for pow2_first in (False, True):
        if pow2_first:
            monkeypatch.setenv("FOURIER_POW2_TILES_FIRST", "1")
        for n, pow2_route, tile_route in cases:
            if pow2_first and tile_route is None:
                continue  # same plan as in the first round
            want = pow2_route if (pow2_first or tile_route is None) else tile_route
            x = np.stack([hash_normal(80 + b, n) for b in range(2)])
            for dtype, tl2 in ((np.complex64, 1e-6), (np.complex128, 5e-14)):
                plan = make(fa, n, dtype)
                assert plan.describe().startswith("stockham " + want), plan.describe()
                for code in (range(5) if n <= 27 * 4096 and not pow2_first else (0, 1)):  # the GPU test runs all five codes on all of them
                    ref = oracle.transform_batch(x.astype(dtype), code)
                    assert rel_l2(run_batch(plan, x.astype(dtype), code), ref) <= tl2, (n, code)
                    assert rel_l2(run_batch(plan, x.astype(dtype), code, inplace=True), ref) <= tl2, (n, code)
    monkeypatch.delenv("FOURIER_POW2_TILES_FIRST")
    assert make(fa, 729 * 4096, np.complex64).describe().startswith("stockham 64x64x27x27")
    assert "mixed-radix" in make(fa, 3 * 2048, np.complex128).describe()  # too little 2^a for two tiled passes: LDS kernel
    assert "mixed tiles 144x128" in make(fa, 9 * 2048, np.complex128).describe()   # f64 18432: beyond the LDS route, a < 12: column tiles of mixed length


def test_mixed_radix_sizes_beyond_the_lds_limit_with_a_small_power_of_two_run_as_tiled_passes(fa, oracle, monkeypatch):
    """2^a * 3^b with a < 12 above 19683 (f32) / 9216 (f64) points: the reference runs them in its Stockham path, one small
    radix per sweep (autosort/mod.rs:104-116).  Here: two or three big-radix passes of MIXED length on column tiles
    (kernels_tiled.h; round 4), and for the rare length without such a factorisation one global-memory pass per radix (27 / 9 /
    3, then 16 / 8 / 4 / 2; round 3).  All five codes, in and out of place, ragged batch and ragged tiles (lengths without a
    factor 16), against the oracle; the two routes against each other and against the Bluestein route they replaced."""
    for n, dtype, tol, desc in ((59049, np.complex64, 1e-6, "243x243"), (62208, np.complex64, 1e-6, "256x243"), (20736, np.complex64, 1e-6, "144x144"),
                                (10368, np.complex128, 5e-14, "108x96"), (13122, np.complex128, 5e-14, "162x81"),
                                (2 * 3 ** 13, np.complex64, 1e-6, "243x162x81")):
        plan = make(fa, n, dtype)
        assert "mixed tiles " + desc in plan.describe(), plan.describe()
        x = np.stack([hash_normal(900 + b, n) for b in range(2 if n < 10 ** 6 else 1)]).astype(dtype)
        for code in (range(5) if n < 30000 else (0, 1)):  # every code on the small ones (the GPU test runs all five on all)
            ref = oracle.transform_batch(x, code)
            assert rel_l2(run_batch(plan, x, code), ref) <= tol, (n, code)
            assert rel_l2(run_batch(plan, x, code, inplace=True), ref) <= tol, (n, code, "in place")
    monkeypatch.setenv("FOURIER_NO_TILED_MIXED", "1")
    for n, dtype, tol in ((59049, np.complex64, 1e-6), (10368, np.complex128, 5e-14)):
        gen = make(fa, n, dtype)
        assert "global-pass" in gen.describe(), gen.describe()
        x = np.stack([hash_normal(900 + b, n) for b in range(2)]).astype(dtype)
        for code in (0, 1, 3):
            ref = oracle.transform_batch(x, code)
            assert rel_l2(run_batch(gen, x, code), ref) <= tol, (n, code)
            assert rel_l2(run_batch(gen, x, code, inplace=True), ref) <= tol, (n, code, "in place")
    monkeypatch.setenv("FOURIER_NO_GENERIC_MIXED", "1")
    blu = make(fa, 62208, np.complex64)
    assert "bluestein" in blu.describe()
    monkeypatch.delenv("FOURIER_NO_TILED_MIXED")
    x = hash_normal(5, 62208).astype(np.complex64)[None, :]
    assert rel_l2(run_batch(blu, x, 0), run_batch(make(fa, 62208, np.complex64), x, 0)) <= 2e-6


def test_register_resident_tile_passes_against_the_lds_tile_passes(fa, oracle, monkeypatch):
    """Round 6 (kernels_regtile.h): a tile pass of mixed length L = R1 x R2 keeps a column's transform in registers -- stage A (DFT_R1 on rows
    loaded straight from global memory), one LDS exchange, stage B (DFT_R2, inter-pass twiddle, store) -- where L splits into two factors of
    at most 32 (and not worse than 4 : 1); the LDS kernels (kernels_tiled.h) stay for 125, 245, 343, 490 and as the A/B arm
    (FOURIER_NO_REGTILE).  Both against the oracle, all five codes, in place, ragged batch and ragged tiles (column counts without a factor
    16; f32 units of two columns: a last column by itself), two and three passes; and against each other."""
    for n, dtype, tol in ((44100, np.complex64, 2e-6), (30870, np.complex64, 2e-6), (20736, np.complex64, 1e-6), (59049, np.complex64, 1e-6),
                          (13122, np.complex128, 5e-14), (44100, np.complex128, 1e-9), (15625, np.complex128, 1e-9), (1000000, np.complex128, 1e-9)):
        reg = make(fa, n, dtype)
        monkeypatch.setenv("FOURIER_NO_REGTILE", "1")
        lds = make(fa, n, dtype)
        monkeypatch.delenv("FOURIER_NO_REGTILE")
        assert "mixed tiles" in reg.describe() and reg.describe() == lds.describe(), (reg.describe(), lds.describe())
        x = np.stack([hash_normal(800 + b, n) for b in range(3 if n < 100000 else 1)]).astype(dtype)
        for code in (range(5) if n < 50000 else (0, 1)):
            ref = oracle.transform_batch(x, code)
            a, b = run_batch(reg, x, code), run_batch(lds, x, code)
            assert rel_l2(a, ref) <= tol and rel_l2(b, ref) <= tol, (n, code, rel_l2(a, ref), rel_l2(b, ref))
            assert rel_l2(a, b) <= (4e-7 if dtype == np.complex64 else 2e-15), (n, code)
            assert np.array_equal(run_batch(reg, x, code, inplace=True), a), (n, code)
        if n != 15625:  # (125 = 25 x 5 stays on the LDS kernel: the same plan twice)
            assert not np.array_equal(run_batch(reg, x, 0), run_batch(lds, x, 0)), n


def test_tile_lengths_of_513_to_1024_points(fa, oracle, monkeypatch):
    """Round 6: 28 tile lengths above 512 points (two register stages of at most 32 points each) on 64-byte row segments, so that the tile stays
    within 64 KiB of LDS: two tile passes where three were needed (640000 = 800 x 800 instead of 100 x 80 x 80) or none existed (390625 = 625 x
    625 took Bluestein), and 2^a 3^b with a >= 12 up to 576 x 576 (81 * 4096: four round trips before).  Against the oracle, in place, a ragged
    tile (625 columns of 4 f64 / 8 f32 columns per tile); without the register-tile kernels the old plans are back."""
    for n, dtype, desc, tol in ((390625, np.complex64, "mixed tiles 625x625", 2e-6), (640000, np.complex128, "mixed tiles 800x800", 1e-9),
                                (81 * 4096, np.complex64, "mixed tiles 576x576", 1e-6),
                                # f32 only: a 40-point stage (1000 = 40 x 25); f64 keeps the three passes (spills at 40 points, r06_s42)
                                (1000000, np.complex64, "mixed tiles 1000x1000", 2e-6)):
        plan = make(fa, n, dtype)
        assert desc in plan.describe(), plan.describe()
        x = np.stack([hash_normal(70 + b, n) for b in range(2)]).astype(dtype)
        for code in (0, 1):
            ref = oracle.transform_batch(x, code)
            a = run_batch(plan, x, code)
            assert rel_l2(a, ref) <= tol, (n, code, rel_l2(a, ref))
            assert np.array_equal(run_batch(plan, x, code, inplace=True), a), (n, code)
    monkeypatch.setenv("FOURIER_NO_REGTILE", "1")
    assert "bluestein" in make(fa, 390625, np.complex64).describe() and "mixed tiles 100x80x80" in make(fa, 640000, np.complex128).describe()


def test_one_launch_chirpz_on_a_smooth_m_in_registers(fa, oracle):
    """Round 6 (kernels_chirpz.h): a short Bluestein length runs the whole chirp-z in one launch on M = R1 x R2 >= 2N - 1 (bluesteins.rs:110 asks
    for no more) with both M-point transforms in registers, 64 / R1 lane groups per one-wave workgroup, where that M is
    shorter than the reference's power of two (f64: wherever such an M exists, f32 -- two transforms per lane on packed arithmetic -- where it is a
    tenth shorter).  Against the oracle (its own chirp-z on the power of two) and the naive DFT, all five codes, in
    place, batches that do not fill the last wave; option bluestein_smooth_m = 0 brings the power-of-two kernels back, = 2 forces the
    register route wherever the menu reaches (every shape of it is run here)."""
    for dtype, tol in ((np.complex64, 2e-6), (np.complex128, 1e-12)):  # (f64: the ORACLE's unreduced chirp angle, bluesteins.rs:10,31,57)
        for n, want in ((17, "M=36 registers 6x6"), (191, "M=400 registers 20x20"), (331, "M=675 registers 27x25"), (307, "M=625 registers 25x25"),
                        (149, "M=324 registers 18x18"), (37, "M=81 registers 9x9"), (31, "M=64 registers 8x8")) + \
                       (((575, "M=1152 registers 36x32"),) if dtype == np.complex64 else ((222, "M=480 registers 24x20"), (511, "M=1024 registers 32x32"))):
            plan = make(fa, n, dtype)
            assert want in plan.describe() and "one-launch" in plan.describe(), plan.describe()
            for batch in (1, 7):
                x = np.stack([hash_normal(40 + b, n) for b in range(batch)]).astype(dtype)
                for code in range(5):
                    ref = oracle.transform_batch(x, code)
                    a = run_batch(plan, x, code)
                    assert rel_l2(a, ref) <= tol, (n, code, rel_l2(a, ref))
                    assert np.array_equal(run_batch(plan, x, code, inplace=True), a), (n, code)
            assert max_rel(run_batch(plan, x[:1], 0)[0], naive_dft(x[0])) <= (3e-6 if dtype == np.complex64 else 5e-13), n  # (the naive sum itself: 1e-13 at 511 points)
            plan.set_option("bluestein_smooth_m", 0)
            assert "registers" not in plan.describe() and "bluestein M=" in plan.describe(), plan.describe()
            assert rel_l2(run_batch(plan, x, 0), oracle.transform_batch(x, 0)) <= tol, n
        # every kernel of the menu, forced: the largest n its M reaches, a ragged batch
        menu = [36, 49, 64, 81, 100, 120, 144, 168, 196, 225, 256, 288, 324, 360, 400, 441, 480, 525, 576, 625, 675, 729, 784, 840, 900, 960, 1024]
        if dtype == np.complex64:
            menu += [1152]
        menu3 = [1296, 1440, 1600, 2304, 2560, 3072, 8820, 9261]  # M = R1 x R2 x R3: a workgroup per transform, three register stages each way
        full, menu = menu + menu3, menu + (menu3[::2] if dtype == np.complex64 else menu3[1::2])
        def rough(n):  # a prime factor above 13: a Bluestein length
            for p in (2, 3, 5, 7, 11, 13):
                while n % p == 0:
                    n //= p
            return n > 1
        for m in menu:
            n = next(v for v in range((m + 1) // 2, 0, -1) if rough(v))
            plan = make(fa, n, dtype)
            plan.set_option("bluestein_smooth_m", 2)
            assert f"bluestein M={min(v for v in full if v >= 2 * n - 1)} registers" in plan.describe(), (n, plan.describe())
            x = np.stack([hash_normal(90 + b, n) for b in range(5 if m <= 1152 else 3)]).astype(dtype)
            truth = np.fft.fft(x.astype(np.complex128), axis=1)
            assert rel_l2(run_batch(plan, x, 0), truth) <= (1e-6 if dtype == np.complex64 else 3e-15), (m, rel_l2(run_batch(plan, x, 0), truth))
            back = run_batch(plan, run_batch(plan, x, 0), 1)
            assert rel_l2(back, x) <= (2e-6 if dtype == np.complex64 else 6e-15), m


REGFFT_LENGTHS = (22, 77, 143, 175, 200, 245, 350, 385, 400, 560, 700, 800, 1001, 2000, 2002, 2904, 4000, 5005, 8000, 8960, 9009, 12000)  # gen_regfft_shapes.py: EMU


def test_lengths_with_factors_5_to_13_as_one_launch_on_register_stages(fa, oracle, monkeypatch):
    """Round 6, sessions 48 - 51 (kernels_regfft.h, regfft_kernel / regfft3_kernel): a length with factors 5 ... 13 that regfft_shapes.h lists in
    the precision runs as a direct transform in ONE launch -- n = j2 + R2 j1, DFT_R1, twiddle, DFT_R2 (, DFT_R3) with one LDS exchange per
    step -- instead of the LDS mixed-radix kernel's round trip per small radix (the reference: Bluestein, fourier/src/lib.rs:38-42; the oracle's
    chirp-z is the comparison, numpy's f64 transform the truth).  All five codes, in place, batches that do not fill the last wave.  The
    emulator build holds the lengths below of the table's 791 (those the A/B left on their earlier route keep it here too)."""
    seen = 0
    for dtype, tol, tol_truth in ((np.complex64, 2e-6, 4e-7), (np.complex128, 5e-12, 1e-15)):  # (f64: the ORACLE's unreduced chirp angle, 1.4e-12 at 8000)
        for n in REGFFT_LENGTHS:
            plan, shape = make(fa, n, dtype), regfft_shape(n, dtype, emu=True)
            if shape is None:  # a length the A/B left on its earlier route
                assert "registers" not in plan.describe(), plan.describe()
                continue
            seen += 1
            assert f"stockham registers {shape} one-launch" in plan.describe(), plan.describe()
            for batch in (1, 7) if n <= 1001 else (3,):
                x = np.stack([hash_normal(60 + b, n) for b in range(batch)]).astype(dtype)
                truth = np.fft.fft(x.astype(np.complex128), axis=1)
                for code in range(5):
                    a = run_batch(plan, x, code)
                    assert rel_l2(a, oracle.transform_batch(x, code)) <= tol, (n, code)
                    assert np.array_equal(run_batch(plan, x, code, inplace=True), a), (n, code)
                assert rel_l2(run_batch(plan, x, 0), truth) <= tol_truth, (n, rel_l2(run_batch(plan, x, 0), truth))
                assert rel_l2(run_batch(plan, run_batch(plan, x, 0), 1), x) <= 2 * tol_truth, n
    assert seen >= 30, seen  # two-stage and three-stage shapes in both precisions
    monkeypatch.setenv("FOURIER_NO_REGFFT", "1")
    assert "mixed-radix" in make(fa, 1001, np.complex64).describe() and "mixed-radix" in make(fa, 350, np.complex128).describe()


def test_plan_option_register_stages_moves_a_2a3b_length_off_the_reference_schedule_on_request(fa, oracle):
    """Round 6 (sessions 66 / 67): 2^a 3^b lengths keep the reference's own schedule by default -- bit-identical to the CPU restatement -- and
    take the register-stage kernel regfft_shapes.h lists for them (FOURIER_REGFFT_OPT_ROW) only under plan option "register_stages" = 1:
    the same values within rounding, every code, in place; 0 restores the bits; a length without such a kernel refuses, a plan that runs
    register stages by default returns OK unchanged."""
    seen = 0
    for n in (729, 1536, 4608):
        for dtype, tol, close in ((np.complex64, 1e-6, 4e-7), (np.complex128, 2e-14, 2e-15)):
            plan = make(fa, n, dtype)
            base = plan.describe()
            assert "mixed-radix" in base
            if regfft_shape(n, dtype, emu=True, on_request=True) is None:  # not listed in this precision (the A/B of session 66)
                with pytest.raises(fa.FourierError):
                    plan.set_option("register_stages", 1)
                assert plan.describe() == base
                continue
            seen += 1
            x = np.stack([hash_normal(70 + b, n) for b in range(3)]).astype(dtype)
            want = {code: run_batch(plan, x, code) for code in range(5)}
            assert all(np.array_equal(want[code], oracle.transform_batch(x, code)) for code in range(5)), n
            plan.set_option("register_stages", 1)
            assert plan.describe().startswith("stockham registers " + regfft_shape(n, dtype, emu=True, on_request=True) + " one-launch"), plan.describe()
            for code in range(5):
                got = run_batch(plan, x, code)
                assert rel_l2(got, oracle.transform_batch(x, code)) <= tol and rel_l2(got, want[code]) <= close, (n, code, rel_l2(got, want[code]))
                assert np.array_equal(run_batch(plan, x, code, inplace=True), got), (n, code)
            plan.set_option("register_stages", 1)  # twice: unchanged
            plan.set_option("register_stages", 0)
            assert plan.describe() == base and np.array_equal(run_batch(plan, x, 0), want[0]), n
    assert seen >= 5, seen
    for n in (1024, 1013, 3 * 4096):  # a power of two, a Bluestein length, 2^a 3^b on tile passes: no kernel on request
        plan = make(fa, n, np.complex64)
        desc = plan.describe()
        with pytest.raises(fa.FourierError):
            plan.set_option("register_stages", 1)
        assert plan.describe() == desc
    plan = make(fa, 1001, np.complex64)
    desc = plan.describe()
    plan.set_option("register_stages", 1)
    plan.set_option("register_stages", 0)
    assert plan.describe() == desc and "registers" in desc
    # the library-wide default (fourier_hip_set_default_option / FOURIER_HIP_REGISTER_STAGES=1): plans created afterwards take the kernel at create
    assert fa.get_default_option("register_stages_at_create") == 0
    fa.set_default_option("register_stages_at_create", 1)
    try:
        plan = make(fa, 729, np.complex128)
        assert plan.describe().startswith("stockham registers 27x27 one-launch"), plan.describe()
        x = np.stack([hash_normal(80 + b, 729) for b in range(2)]).astype(np.complex128)
        assert rel_l2(run_batch(plan, x, 0), oracle.transform_batch(x, 0)) <= 2e-14
        plan.set_option("register_stages", 0)  # a handle can still go back
        assert "mixed-radix" in plan.describe() and np.array_equal(run_batch(plan, x, 0), oracle.transform_batch(x, 0))
        assert "mixed-radix" in make(fa, 96, np.complex64).describe() and "registers" in make(fa, 1001, np.complex64).describe()
    finally:
        fa.set_default_option("register_stages_at_create", 0)
    assert "mixed-radix" in make(fa, 729, np.complex128).describe()


def test_bluestein_fusion_matches_unfused(fa):
    """The fused Bluestein forms (whole chirp-z in one launch for M <= 2^15; chirp steps fused into the
    inner passes above) give the same values, to rounding, as the separate blu_pre / blu_post sweeps
    (bluesteins.rs:229-258), for every transform code, in and out of place."""
    for n in (17, 102, 439, 1025, 3001, 40001):  # no prime factor below 17 (or too long for LDS): Bluestein
        x = np.stack([hash_normal(70 + b, n) for b in range(2)]).astype(np.complex64)
        fused, plain = make(fa, n, np.complex64), make(fa, n, np.complex64)
        plain.set_option("bluestein_fusion", 0)
        for code in range(5):
            a, b = run_batch(fused, x, code), run_batch(plain, x, code)
            assert rel_l2(a, b) <= 3e-7, (n, code, rel_l2(a, b))
            assert np.array_equal(run_batch(fused, x, code, inplace=True), a), (n, code)
    assert "fused" in make(fa, 3001, np.complex64).describe() and "fused" in make(fa, 1013, np.complex64).describe()


def test_lane_per_transform_and_small_row_kernels_with_ragged_batches(fa):
    """N = 2..16 (f32: ..32) run one lane per transform: the wave loads coalesced 16-byte units and transposes
    them across lanes with __shfl_xor; N = 32 (f64) / 64 are the smallest row kernels.  Batches that do not fill
    the last wave / workgroup, in and out of place, several scalings."""
    for dtype, tol in ((np.complex64, 3e-7), (np.complex128, 1e-15)):
        for n in (2, 4, 8, 16, 32, 64):
            plan = make(fa, n, dtype)
            for batch in ((1, 3, 64, 65, 257) if n <= 32 else (1, 2, 31, 64, 127, 129, 300)):
                x = np.stack([hash_normal(b * 7 + n, n) for b in range(batch)]).astype(dtype)
                for code in (0, 1, 4):
                    x128 = x.astype(np.complex128)
                    ref = np.fft.fft(x128, axis=1) if code == 0 else np.fft.ifft(x128, axis=1) * (1 if code == 1 else np.sqrt(n))
                    y = run_batch(plan, x, code)
                    assert rel_l2(y, ref) <= tol, (n, batch, code, rel_l2(y, ref))
                    assert np.array_equal(run_batch(plan, x, code, inplace=True), y), (n, batch, code)


def test_bluestein_conv_kernel_matches_separate_passes(fa, oracle):
    """Large Bluestein plans run the forward inner FFT's last pass, the multiply by the transformed chirp
    and the inverse inner FFT's first pass as ONE launch (bluesteins.rs:236-239 in a single sweep).
    Where the pass lengths are a palindrome (256x256) the same plan serves both directions and the values
    are bit-identical to the separate-pass form; otherwise the inverse runs the mirrored plan (512x256 forward,
    256x512 inverse) and agrees to rounding.  Both stay within the oracle tolerance."""
    for n, dtype, exact, tol in ((20002, np.complex64, True, 2e-6), (40001, np.complex64, False, 2e-6),
                                 (10001, np.complex128, False, 5e-11), (70001, np.complex128, True, 5e-11)):
        x = np.stack([hash_normal(7 + b, n) for b in range(2)]).astype(dtype)
        conv, plain = make(fa, n, dtype), make(fa, n, dtype)
        for p in (conv, plain):
            p.set_option("bluestein_smooth_m", 0)  # the power-of-two work array (10001 / 70001 f64 take a smooth M by default: round 6)
        plain.set_option("bluestein_conv", 0)
        y = np.empty_like(x)
        names = [p[0] for p in conv.profile_batch_ptr(x.ctypes.data, y.ctypes.data, 2, 0) if p[2] > 0]
        assert names == ["fwd_pass0", "conv_pass", "inv_pass1"], names
        assert [p[0] for p in plain.profile_batch_ptr(x.ctypes.data, y.ctypes.data, 2, 0) if p[2] > 0] == \
            ["fwd_pass0", "fwd_pass1", "inv_pass0", "inv_pass1"]
        for code in range(5):
            a, b, ref = run_batch(conv, x, code), run_batch(plain, x, code), oracle.transform_batch(x, code)
            if exact:
                assert np.array_equal(a, b), (n, code)
            else:
                assert rel_l2(a, b) <= (3e-7 if dtype == np.complex64 else 1e-15), (n, code, rel_l2(a, b))
            assert rel_l2(a, ref) <= tol, (n, code, rel_l2(a, ref))
            assert np.array_equal(run_batch(conv, x, code, inplace=True), a), (n, code)


def test_bluestein_reference_chirp_option(fa, oracle):
    """Plan option "bluestein_reference_chirp": tables from the reference's unreduced angle k^2 * pi / N (bluesteins.rs:10,31,57) -- the engine
    then equals the CPU restatement to f64 rounding where by default it equals the exact DFT (the GPU test runs the large lengths)."""
    for n in (73, 1013, 20011):
        x = np.stack([hash_normal(30 + b, n) for b in range(2)]).astype(np.complex128)
        plan = make(fa, n, np.complex128)
        base = run_batch(plan, x, 0)
        assert rel_l2(base, np.fft.fft(x, axis=1)) <= 5e-15
        plan.set_option("bluestein_reference_chirp", 1)
        for code in (0, 1):
            assert rel_l2(run_batch(plan, x, code), oracle.transform_batch(x, code)) <= 5e-15, (n, code)
        plan.set_option("bluestein_reference_chirp", 0)
        assert np.array_equal(run_batch(plan, x, 0), base)
    with pytest.raises(fa.FourierError):
        make(fa, 1024, np.complex128).set_option("bluestein_reference_chirp", 1)


def test_bluestein_chirp_in_pass_computes_the_chirp(fa, oracle):
    """The fused chirp-in first pass builds x[k] = exp(-i*pi*k^2/N) from a row table, a column table and an exact-exponent
    cross term (option bluestein_chirp_compute, default on) instead of reading the N-entry table: both against the
    oracle and against each other, forward and inverse, f32 and f64."""
    for n, dtype, tol in ((40001, np.complex64, 2e-6), (70001, np.complex64, 2e-6), (40001, np.complex128, 5e-11)):
        x = np.stack([hash_normal(600 + b, n) for b in range(2)]).astype(dtype)
        comp, read = make(fa, n, dtype), make(fa, n, dtype)
        for p in (comp, read):
            p.set_option("bluestein_smooth_m", 0)  # (the smooth-M route reads the chirp table)
        comp.set_option("bluestein_chirp_compute", 1)  # default: on only for long first passes and tables beyond the L2
        read.set_option("bluestein_chirp_compute", 0)
        for code in (0, 1):
            ref = oracle.transform_batch(x, code)
            a, b = run_batch(comp, x, code), run_batch(read, x, code)
            assert rel_l2(a, ref) <= tol and rel_l2(b, ref) <= tol, (n, code, rel_l2(a, ref), rel_l2(b, ref))
            assert rel_l2(a, b) <= (3e-7 if dtype == np.complex64 else 1e-12), (n, code, rel_l2(a, b))
            assert not np.array_equal(a, b) or dtype == np.complex128  # the two routes are really different code


def test_bluestein_on_a_smooth_work_array(fa, oracle):
    """Round 6: M need only reach 2N - 1 (bluesteins.rs:110; the reference rounds up to a power of two).  Where the power-of-two work array
    is at least 1.6 x longer than a product of two register-tile lengths (the smallest, or one up to 4 % longer whose lengths split more
    evenly into their register stages), the three sweeps run on that product (kernels_regtile.h: chirp-in first pass, conv, chirp-out last pass).  All five codes against the oracle, in place, a ragged batch; the plan
    option brings the power-of-two route back and both agree; the reference's chirp angle on request."""
    for n, dtype, desc, tol in ((16411, np.complex64, "bluestein M=32928 inner mixed tiles 196x168", 2e-6),
                                (10007, np.complex128, "bluestein M=20160 inner mixed tiles 168x120", 5e-11),
                                (32771, np.complex128, "bluestein M=65856 inner mixed tiles 336x196", 5e-11)):
        plan = make(fa, n, dtype)
        assert desc in plan.describe(), plan.describe()
        x = np.stack([hash_normal(40 + b, n) for b in range(3)]).astype(dtype)
        y = np.empty_like(x)
        names = [p[0] for p in plan.profile_batch_ptr(x.ctypes.data, y.ctypes.data, 3, 0) if p[2] > 0]
        assert names == ["chirp_in_pass", "conv_pass", "chirp_out_pass"], names
        pow2 = make(fa, n, dtype)
        pow2.set_option("bluestein_smooth_m", 0)
        assert "mixed tiles" not in pow2.describe() and "bluestein M=%d " % (1 << int(np.ceil(np.log2(2 * n - 1)))) in pow2.describe(), pow2.describe()
        for code in range(5):
            ref = oracle.transform_batch(x, code)
            a = run_batch(plan, x, code)
            assert rel_l2(a, ref) <= tol, (n, code, rel_l2(a, ref))
            assert np.array_equal(run_batch(plan, x, code, inplace=True), a), (n, code)
            assert rel_l2(a, run_batch(pow2, x, code)) <= (4e-7 if dtype == np.complex64 else 4e-15), (n, code)
        assert rel_l2(run_batch(plan, x, 0), np.fft.fft(x.astype(np.complex128), axis=1)) <= (4e-7 if dtype == np.complex64 else 4e-15)
        pow2.set_option("bluestein_smooth_m", 1)  # ... and back
        assert pow2.describe() == plan.describe()
        assert np.array_equal(run_batch(pow2, x, 0), run_batch(plan, x, 0))
        pow2.set_option("chunk_bytes", 2 * n * x.itemsize)  # chunks of one transform (work array and scratch of one chunk): the same bits
        assert np.array_equal(run_batch(pow2, x, 0), run_batch(plan, x, 0)) and np.array_equal(run_batch(pow2, x, 1, inplace=True), run_batch(plan, x, 1))
    plan = make(fa, 10007, np.complex128)
    plan.set_option("bluestein_reference_chirp", 1)
    x = np.stack([hash_normal(50 + b, 10007) for b in range(2)]).astype(np.complex128)
    for code in (0, 1):
        assert rel_l2(run_batch(plan, x, code), oracle.transform_batch(x, code)) <= 5e-15, code
    # lengths just below a power of two (M / M_smooth < 1.6) and every M within the one-launch kernels keep the power of two
    assert "mixed tiles" not in make(fa, 24001, np.complex64).describe() and "mixed tiles" not in make(fa, 5003, np.complex128).describe()
    with pytest.raises(fa.FourierError):
        make(fa, 4096, np.complex64).set_option("bluestein_smooth_m", 0)


def test_bluestein_conv_three_pass_inner_plan(fa):
    # M = 2^23: forward 256x256x128, inverse mirrored 128x256x256, middle launch = last forward + first inverse
    n = 2200000
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)[None, :]
    plan = make(fa, n, np.complex64)
    assert "M=8388608 inner 256x256x128" in plan.describe()
    y = np.empty_like(x)
    names = [p[0] for p in plan.profile_batch_ptr(x.ctypes.data, y.ctypes.data, 1, 0) if p[2] > 0]
    assert names == ["fwd_pass0", "fwd_pass1", "conv_pass", "inv_pass1", "inv_pass2"], names
    assert rel_l2(y[0], np.fft.fft(x[0].astype(np.complex128))) <= 2e-6


def test_one_launch_plans_match_the_two_launch_plans(fa, monkeypatch):
    """2^12..2^15 run both Stockham passes inside one workgroup (one HBM round trip); the two-launch
    plan of the same size (forced with FOURIER_NO_TWOLEVEL) must agree to rounding."""
    for n in (4096, 8192, 1 << 14, 1 << 15):
        x = np.stack([hash_normal(500 + b, n) for b in range(2)]).astype(np.complex64)
        one = make(fa, n, np.complex64)
        assert "one-launch" in one.describe()
        monkeypatch.setenv("FOURIER_NO_TWOLEVEL", "1")
        two = make(fa, n, np.complex64)
        monkeypatch.delenv("FOURIER_NO_TWOLEVEL")
        assert "one-launch" not in two.describe()
        for code in (0, 1, 3):
            a, b = run_batch(one, x, code), run_batch(two, x, code)
            assert rel_l2(a, b) <= 3e-7, (n, code, rel_l2(a, b))
            assert np.array_equal(run_batch(one, x, code, inplace=True), a)
    assert "one-launch" in make(fa, 1 << 14, np.complex128).describe()
    assert "one-launch" not in make(fa, 1 << 15, np.complex128).describe()  # f64 2^15 does not fit a workgroup


def test_random_sizes_batches_codes_vs_oracle(fa, oracle):
    """Seeded random sweep over every plan family (tiny, row, one-launch, mixed-radix, all Bluestein forms):
    random size, batch, transform code, precision, in/out of place -- against the CPU restatement."""
    rng = np.random.default_rng(20260926)
    sizes = [int(v) for v in rng.integers(1, 6000, 28)] + [4096 + 1, 8192, 12345]
    for n in sizes:
        dtype = np.complex64 if rng.integers(0, 2) else np.complex128
        batch = int(rng.integers(1, 5))
        code = int(rng.integers(0, 5))
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(dtype)
        plan = make(fa, n, dtype)
        ref = oracle.transform_batch(x, code)
        tol = 2e-6 if dtype == np.complex64 else 1e-11  # f64: the oracle's own unreduced-chirp error dominates
        for inplace in (False, True):
            got = run_batch(plan, x, code, inplace)
            assert rel_l2(got, ref) <= tol, (n, plan.describe(), batch, code, inplace, rel_l2(got, ref))


def test_three_pass_plan(fa, monkeypatch):
    """2^23: by default two passes, 4096 (32-byte-wide first-pass tiles) x 2048; FOURIER_THREE_PASS_2P23=1 keeps the
    256 x 256 x 128 plan, which exercises the middle (uniform-twiddle) pass every size from 2^24 up uses."""
    n = 1 << 23
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(n, np.float32) + 1j * rng.standard_normal(n, np.float32)).astype(np.complex64)[None, :]
    ref = np.fft.fft(x[0].astype(np.complex128))
    monkeypatch.setenv("FOURIER_THREE_PASS_2P23", "1")
    three = make(fa, n, np.complex64)
    monkeypatch.delenv("FOURIER_THREE_PASS_2P23")
    assert "256x256x128" in three.describe()
    assert rel_l2(run_batch(three, x, 0)[0], ref) <= 1e-6
    assert rel_l2(run_batch(three, x, 0, inplace=True)[0], ref) <= 1e-6
    two = make(fa, n, np.complex64)
    assert "4096x2048" in two.describe()
    assert rel_l2(run_batch(two, x, 0)[0], ref) <= 1e-6
    assert rel_l2(run_batch(two, x, 0, inplace=True)[0], ref) <= 1e-6


def test_chunking_and_scratch_options_do_not_change_results(fa):
    n, batch = 4096, 7
    x = np.stack([hash_normal(300 + b, n) for b in range(batch)]).astype(np.complex64)
    base = run_batch(make(fa, n, np.complex64), x, 0)
    for chunk_bytes, scratch in ((n * 8, 0), (3 * n * 8, 1), (0, 1)):
        plan = make(fa, n, np.complex64)
        plan.set_option("chunk_bytes", chunk_bytes)
        plan.set_option("scratch", scratch)
        assert np.array_equal(run_batch(plan, x, 0), base)
        assert np.array_equal(run_batch(plan, x, 0, inplace=True), base)
    with pytest.raises(fa.FourierError):
        make(fa, n, np.complex64).set_option("no_such_option", 1)
    # the workgroup -> tile mappings are bijections: same bits whatever the mapping (two-pass, Bluestein conv and
    # one-launch plans; batch sizes that do and do not divide by the XCD count)
    for n2, batch2 in ((1 << 16, 3), (1 << 16, 8), (1 << 16, 16), (40001, 2), (40001, 8), (4096, 5)):
        x2 = np.stack([hash_normal(400 + b, n2) for b in range(batch2)]).astype(np.complex64)
        base2 = run_batch(make(fa, n2, np.complex64), x2, 0)
        for mode in (0, 1, 2, 3, 4):  # 4 = band-major walk of each XCD's own transforms (batch a multiple of 8)
            plan = make(fa, n2, np.complex64)
            plan.set_option("xcd_swizzle", mode)
            assert np.array_equal(run_batch(plan, x2, 0), base2), (n2, batch2, mode)
        # the general band walk (round 5; the default order of f32 2^20): tiles per band | transforms per group << 8 | transform-fastest << 19
        for walk in (1, 4, 8, 16, 4 | 1 << 8, 2 | 2 << 8, 8 | 1 << 19, 4 | 2 << 8 | 1 << 19, 8 | 1 << 20, 4 | 3 << 19, 3, 0):  # (3 does not divide the tiles: the plain order)
            plan = make(fa, n2, np.complex64)
            plan.set_option("tile_walk", walk)
            assert np.array_equal(run_batch(plan, x2, 0), base2), (n2, batch2, walk)
    with pytest.raises(fa.FourierError):
        make(fa, n, np.complex64).set_option("xcd_swizzle", 5)
    with pytest.raises(fa.FourierError):
        make(fa, n, np.complex64).set_option("tile_walk", 1 << 21)


def test_stream_pipeline_option_is_bit_identical_and_needs_no_batch_sized_scratch(fa):
    """Plan option "stream_pipeline" (round 6): the two passes of a two-pass plan chunk by chunk over two internal streams with the
    intermediate in a small ring -- the same kernels on the same data, so the same bits, out of place and in place, for chunk
    sizes that do and do not divide the batch; refused where the plan is not a plain two-pass plan."""
    for n, batch, dtype in ((1 << 16, 11, np.complex64), (1 << 16, 5, np.complex128)):
        x = np.stack([hash_normal(900 + b, n) for b in range(batch)]).astype(dtype)
        base = run_batch(make(fa, n, dtype), x, 0)
        base_inv = run_batch(make(fa, n, dtype), x, 1)
        for chunk, slots, one in ((1, 2, 0), (2, 3, 0), (4, 2, 0), (3, 4, 1), (64, 2, 0)):
            plan = make(fa, n, dtype)
            plan.set_option("stream_pipeline", chunk | slots << 16 | one << 24)
            assert np.array_equal(run_batch(plan, x, 0), base), (n, chunk, slots)
            assert np.array_equal(run_batch(plan, x, 0, inplace=True), base), (n, chunk, slots)
            assert np.array_equal(run_batch(plan, x, 1, inplace=True), base_inv), (n, chunk, slots)
            plan.set_option("stream_pipeline", 0)
            assert np.array_equal(run_batch(plan, x, 0), base)
    for n in (4096, 1000, 40001, 1 << 24):  # one launch, LDS mixed radix, Bluestein, three passes
        with pytest.raises(fa.FourierError):
            make(fa, n, np.complex64).set_option("stream_pipeline", 2 | 2 << 16)


def test_out_of_memory_for_the_scratch_falls_back_to_smaller_chunks(fa, monkeypatch):
    """An in-place call needs a scratch of one chunk (default: the whole batch).  When the device cannot give that
    much the engine halves the chunk until the allocation fits instead of failing; same bits as the unchunked run."""
    for n, batch in ((1 << 16, 6), (40001, 5)):  # two-pass in place; Bluestein work + scratch
        x = np.stack([hash_normal(600 + b, n) for b in range(batch)]).astype(np.complex64)
        ref = run_batch(make(fa, n, np.complex64), x, 0, inplace=True)
        plan = make(fa, n, np.complex64)
        monkeypatch.setenv("HIPEMU_MAX_ALLOC", str(2 * (1 << 17 if n == 40001 else n) * 8 + 4096))  # room for two transforms
        got = run_batch(plan, x, 0, inplace=True)
        monkeypatch.delenv("HIPEMU_MAX_ALLOC")
        assert np.array_equal(got, ref), n


def test_linearity_and_roundtrip_properties(fa):
    n = 1 << 14
    plan = make(fa, n, np.complex64)
    a = hash_normal(1, n).astype(np.complex64)[None, :]
    b = hash_normal(2, n).astype(np.complex64)[None, :]
    fa_, fb, fab = run_batch(plan, a, 0), run_batch(plan, b, 0), run_batch(plan, a + 2 * b, 0)
    assert rel_l2(fab, fa_ + 2 * fb) <= 1e-6
    assert rel_l2(run_batch(plan, fa_, 1), a) <= 1e-6  # Ifft(Fft(x)) == x
    assert rel_l2(run_batch(plan, run_batch(plan, a, 3), 4), a) <= 1e-6  # unitary pair
    # Parseval
    assert abs(np.linalg.norm(fa_) ** 2 / n - np.linalg.norm(a) ** 2) <= 1e-5 * np.linalg.norm(a) ** 2


def test_error_behaviour_matches_reference_ffi(fa):
    from fourier_amd import _lib

    L = _lib.lib()
    assert not L.fourier_create_float(0)  # reference hangs; we return NULL
    assert not L.fourier_create_double(0)
    L.fourier_destroy_float(None)  # no-op
    x = hash_normal(3, 8).astype(np.complex64)
    buf = x.copy()
    L.fourier_transform_in_place_float(None, buf.ctypes.data, 0)  # NULL handle: no-op
    assert np.array_equal(buf, x)
    h = L.fourier_create_float(8)
    L.fourier_transform_in_place_float(h, buf.ctypes.data, 9)  # unknown code: silent no-op (lib.rs:10)
    assert np.array_equal(buf, x)
    assert L.fourier_hip_transform_batch_float(h, buf.ctypes.data, buf.ctypes.data, 1, 9, None) == 1
    assert L.fourier_hip_last_status_float(h) == 1
    assert L.fourier_hip_status_string(1) == b"invalid argument"
    assert L.fourier_hip_size_float(h) == 8 and L.fourier_hip_size_float(None) == 0
    L.fourier_destroy_float(h)
    with pytest.raises(fa.FourierError):
        fa.create_fft_f32(0)
    with pytest.raises(fa.FourierError):
        fa.create_fft_f32(1 << 31)  # beyond the engine's range
    plan = fa.create_fft_f32(8)
    with pytest.raises(ValueError):  # fft.rs:57-58 length asserts
        plan.transform(np.zeros(8, np.complex64), np.zeros(9, np.complex64), fa.Transform.Fft)
    with pytest.raises(TypeError):
        plan.fft_in_place(np.zeros(8, np.complex128))


def test_host_batched_entry_point_streams_chunks(fa):
    """fourier_hip_transform_batch_host_*: host arrays of many transforms go through the device in chunks
    (32 MiB by default, four staging slots, copies and kernels on separate streams).  Same bits as the device-resident batched call,
    for single- and multi-chunk batches, in and out of place, through the operator layer and directly."""
    for n, batch, chunk_bytes in ((1000, 1, None), (1000, 7, 8000), (4096, 33, 1 << 17), (1 << 16, 70, None), (64, 3000, 1 << 16)):
        rng = np.random.default_rng(n)
        x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
        plan = make(fa, n, np.complex64)
        if chunk_bytes:
            plan.set_option("host_chunk_bytes", chunk_bytes)  # many small chunks: every slot is reused several times
        ref = run_batch(plan, x, 1)
        y = np.empty_like(x)
        plan.transform_batch_host(x, y, fa.Transform.Ifft)
        assert np.array_equal(y, ref), (n, batch)
        z = x.copy()
        if batch > 1:
            plan.transform(z, z, fa.Transform.Ifft)  # numpy arrays holding several transforms take the same route
        else:
            plan.transform_batch_host(z, z, fa.Transform.Ifft)
        assert np.array_equal(z, ref), (n, batch)
    with pytest.raises(ValueError):
        make(fa, 16, np.complex64).transform_batch_host(np.zeros(40, np.complex64), np.zeros(40, np.complex64), 0)


def test_profile_hook_reports_every_kernel(fa):
    plan = make(fa, 1 << 16, np.complex64)  # 256 x 256: two launches
    x = hash_normal(1, 1 << 16).astype(np.complex64)[None, :]
    y = np.empty_like(x)
    prof = plan.profile_batch_ptr(x.ctypes.data, y.ctypes.data, 1, 0)
    assert [p[0] for p in prof] == ["pass0", "pass1"] and all(p[2] == 1 for p in prof)
    one = make(fa, 4096, np.complex64)  # 64 x 64 inside one workgroup: a single launch
    x1 = hash_normal(1, 4096).astype(np.complex64)[None, :]
    y1 = np.empty_like(x1)
    assert [p[0] for p in one.profile_batch_ptr(x1.ctypes.data, y1.ctypes.data, 1, 0)] == ["pass0"]
    assert rel_l2(y[0], np.fft.fft(x[0].astype(np.complex128))) <= 1e-6
    planb = make(fa, 102, np.complex64)  # M = 256; with fusion off: separate chirp kernels around the inner FFT
    planb.set_option("bluestein_fusion", 0)
    xb = hash_normal(1, 102).astype(np.complex64)[None, :]
    yb = np.empty_like(xb)
    names = [p[0] for p in planb.profile_batch_ptr(xb.ctypes.data, yb.ctypes.data, 1, 0)]
    assert names == ["blu_pre", "fwd_pass0", "inv_pass0", "blu_post"]
    planc = make(fa, 1003, np.complex64)  # M = 2048: the whole chirp-z in one launch
    xc = hash_normal(1, 1003).astype(np.complex64)[None, :]
    yc = np.empty_like(xc)
    assert [p[0] for p in planc.profile_batch_ptr(xc.ctypes.data, yc.ctypes.data, 1, 0)] == ["bluestein_one_launch"]
    assert rel_l2(yc[0], np.fft.fft(xc[0].astype(np.complex128))) <= 2e-6


def test_lds_layouts_are_bank_conflict_light(fa):
    """Bank-conflict model of MI355X_MICROARCH.md (LDS table) applied to the headline tile (1024-point f32 pass, split planes) and -- round 6 --
    to the whole-transform (row-mode) kernels, the one-launch chirp-z kernels built on them and the register-tile passes of mixed length
    (exchange planes with rows swapped in the odd planes, odd leading dimension in the staging): total LDS-array cycles within 1.1x of
    conflict-free for the pass, exactly conflict-free for the row-mode swizzle (tools/lds_rows_swizzle_search.py found one per shape; under
    the skew layout of rounds 1 - 5 these were 2.3x (L = 512, 1024) to 6.7x (L = 128), and SQ_LDS_BANK_CONFLICT agreed: 57 % at M = 512)."""
    import subprocess
    import sys

    code = (
        "import sys, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from emu import build_emu\n"
        "from fourier_amd import _lib\n"
        "c = build_emu.load(); _lib._lib = c\n"
        "import fourier_amd as fa\n"
        "a, b, d = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()\n"
        "for n, real, batch in ((1 << 20, 'f32', 1), (64, 'f32', 32), (128, 'f32', 64), (256, 'f32', 32), (512, 'f32', 32), (1024, 'f32', 16), (128, 'f64', 32), (512, 'f64', 16),\n"
        "                       (37, 'f32', 32), (97, 'f32', 16), (191, 'f32', 8), (439, 'f32', 4), (97, 'f64', 16), (191, 'f64', 8), (439, 'f64', 8),\n"
        "                       (44100, 'f32', 2), (48000, 'f32', 2), (20736, 'f64', 2), (100000, 'f64', 1), (16411, 'f32', 2), (10007, 'f64', 2)):\n"
        "    p = (fa.create_fft_f32 if real == 'f32' else fa.create_fft_f64)(n); x = np.ones((batch, n), np.complex64 if real == 'f32' else np.complex128); y = np.empty_like(x)\n"
        "    c.fourier_emu_lds_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), 1)\n"
        "    p.transform_batch_ptr(x.ctypes.data, y.ctypes.data, batch, 0)\n"
        "    c.fourier_emu_lds_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), 1)\n"
        "    print(n, real, b.value / d.value)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIPEMU_LDS_TRACE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split() for l in out.stdout.strip().splitlines() if len(l.split()) == 3]
    assert len(rows) == 21, out.stdout
    for n, real, ratio in rows[:8]:
        assert float(ratio) <= (1.1 if n == str(1 << 20) else 1.0), (n, real, ratio)
    # short Bluestein lengths: since session 45 the chirp-z kernels in registers (kernels_chirpz.h; 439 f32 stays on the row-mode kernels): padded
    # planes, 16-byte elements -- conflict-free at 9 x 9 and 20 x 20, 1.08 at 30 x 30 (SQ_LDS_BANK_CONFLICT on the GPU: 0 and 7.7 %,
    # profiles/r06_s45_sq_chirpz_reg.json), 1.25 at 14 x 14; the LDS instructions are 5 % of these kernels' wave cycles
    for n, real, ratio in rows[8:15]:
        assert float(ratio) <= 1.25, (n, real, ratio)
    for n, real, ratio in rows[15:]:  # round 6: the register-tile passes (two-pass plans and the three Bluestein sweeps on a smooth M)
        assert float(ratio) <= 1.05, (n, real, ratio)


def test_lds_mixed_radix_passes_are_bank_conflict_free():
    """The layout map between the first passes of the per-length mixed-radix kernels (mixed_schedule.h: mix_out_layout): under
    the bank model of MI355X_MICROARCH.md the in-pass LDS accesses of the power-of-two-heavy schedules cost 2.0-2.4 x their
    conflict-free cycles in the plain layout (hardware: SQ_LDS_BANK_CONFLICT 36-48 % of SQ_LDS_IDX_ACTIVE,
    profiles/r04_s14b_sq_breakdown_mixed.json) and 1.0 x with the map (hardware: 0.0 %, r04_s17b)."""
    import subprocess
    import sys

    code = (
        "import sys, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from emu import build_emu\n"
        "from fourier_amd import _lib\n"
        "c = build_emu.load(); _lib._lib = c\n"
        "import fourier_amd as fa\n"
        "for n, mk, dt in ((768, fa.create_fft_f32, np.complex64), (3072, fa.create_fft_f32, np.complex64), (3072, fa.create_fft_f64, np.complex128)):\n"
        "    p = mk(n); x = np.ones((2, n), dt); y = np.empty_like(x)\n"
        "    a, b, d = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()\n"
        "    c.fourier_emu_lds_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), 1)\n"
        "    p.transform_batch_ptr(x.ctypes.data, y.ctypes.data, 2, 0)\n"
        "    c.fourier_emu_lds_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(d), 1)\n"
        "    assert 'mixed-radix' in p.describe(), p.describe()\n"
        "    assert np.allclose(y[0, 0], n) and abs(y[0, 1:]).max() < 1e-3 * n\n"
        "    print(n, a.value, b.value / d.value)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIPEMU_LDS_TRACE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l.split() for l in out.stdout.strip().splitlines()]
    assert len(rows) == 3
    for n, instr, ratio in rows:
        assert int(instr) > 0          # the model saw the passes' LDS accesses
        assert float(ratio) <= 1.02, (n, ratio)


def test_host_staging_copies_cover_byte_counts_that_do_not_divide_over_the_copy_threads(fa):
    """Regression (round-1 advisor, high): parallel_copy split a job with floor(bytes / nt) and dropped the last
    r < nt bytes of jobs of the form nt*4096*k + r -- the last element of a large host batch was never staged in
    nor copied back.  n = 7, batch = 452023 is 12*4096*515 + 8 bytes when it is one chunk."""
    n, batch = 7, 452023
    assert (n * batch * 8) % (12 * 4096) == 8
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))).astype(np.complex64)
    plan = make(fa, n, np.complex64)
    plan.set_option("host_chunk_bytes", x.nbytes)  # the whole batch is one chunk, i.e. one copy job each way
    y = np.full_like(x, np.nan)
    plan.transform_batch_host(x, y, fa.Transform.Fft)
    ref = run_batch(plan, x[-3:], 0)
    assert not np.isnan(y.view(np.float32)).any()          # every output byte was written back
    assert np.array_equal(y[-3:], ref)                      # and the last transform saw its whole input
    # the legacy one-transform ABI stages through the same routine: a transform of 12*4096*k + 8 bytes
    n2 = (12 * 4096 * 40 + 8) // 8
    x2 = hash_normal(9, n2).astype(np.complex64)
    p2 = make(fa, n2, np.complex64)
    y2 = np.full_like(x2, np.nan)
    p2.transform(x2, y2, fa.Transform.Fft)
    ref2 = run_batch(p2, x2[None, :], 0)[0]
    assert np.array_equal(y2, ref2)


def test_last_status_is_the_status_of_the_last_call(fa):
    """Regression (round-1 advisor, medium): a failing call left the handle's status set for ever, so every later
    successful numpy transform raised.  Now every entry point resets it on entry (include/fourier.h)."""
    plan = make(fa, 64, np.complex64)
    x = hash_normal(2, 64).astype(np.complex64)
    y = np.empty_like(x)
    with pytest.raises(fa.FourierError):
        plan.transform_batch_ptr(x.ctypes.data, y.ctypes.data, 1, 9)  # unknown transform code
    plan.transform(x, y, fa.Transform.Fft)  # must not raise
    assert rel_l2(y, np.fft.fft(x.astype(np.complex128))) < 1e-6
    from fourier_amd import _lib

    L = _lib.lib()
    assert L.fourier_hip_last_status_float(plan._h) == 0
    # the contract of include/fourier.h, entry point by entry point (round-2 advisor): the ones that do work reset the
    # status on entry -- the legacy `void` transforms included, which report ONLY through this query (the Rust shim's
    # `assert!(status == 0)` after them depends on it) -- and the pure queries leave it untouched
    bad = lambda: L.fourier_hip_transform_batch_float(plan._h, x.ctypes.data, y.ctypes.data, 1, 9, None)  # noqa: E731
    assert bad() != 0 and L.fourier_hip_last_status_float(plan._h) != 0
    for query in (lambda: L.fourier_hip_size_float(plan._h), lambda: L.fourier_hip_device_float(plan._h),
                  lambda: L.fourier_hip_describe_float(plan._h), lambda: L.fourier_hip_model_bytes_float(plan._h),
                  lambda: L.fourier_hip_slot_names_float(plan._h), lambda: L.fourier_hip_last_status_float(plan._h)):
        query()
        assert L.fourier_hip_last_status_float(plan._h) != 0  # still the failed call's status
    for work in (lambda: L.fourier_transform_float(plan._h, x.ctypes.data, y.ctypes.data, 0),
                 lambda: L.fourier_transform_in_place_float(plan._h, y.ctypes.data, 1),
                 lambda: L.fourier_hip_reserve_float(plan._h, 4, 1),
                 lambda: L.fourier_hip_synchronize_float(plan._h, None),
                 lambda: L.fourier_hip_transform_batch_host_float(plan._h, x.ctypes.data, y.ctypes.data, 1, 0)):
        assert bad() != 0 and L.fourier_hip_last_status_float(plan._h) != 0
        work()
        assert L.fourier_hip_last_status_float(plan._h) == 0
    # an unknown code through a legacy `void` entry point is a silent no-op (fourier-ffi/src/lib.rs:10) and not an error
    y[:] = 7
    L.fourier_transform_float(plan._h, x.ctypes.data, y.ctypes.data, 9)
    assert L.fourier_hip_last_status_float(plan._h) == 0 and (y == 7).all()


def test_reserve_presizes_the_scratch_so_that_calls_do_not_allocate(fa):
    """fourier_hip_reserve_*: after reserve(batch, in_place) a batched call of at most that batch performs no
    device allocation (hipMalloc synchronises; needed for graph capture).  The emulator counts allocations."""
    from fourier_amd import _lib

    L = _lib.lib()
    L.fourier_emu_alloc_count.restype = ctypes.c_uint64
    for n in (1 << 16, 1000003 // 11, 3 << 12):  # two-pass in place, Bluestein with a work array, three launches
        plan = make(fa, n, np.complex64)
        x = hash_normal(4, 3 * n).astype(np.complex64).reshape(3, n)
        ref = run_batch(plan, x, 0, inplace=True)
        plan2 = make(fa, n, np.complex64)
        plan2.reserve(3, in_place=True)
        before = L.fourier_emu_alloc_count()
        got = run_batch(plan2, x, 0, inplace=True)
        got1 = run_batch(plan2, x[:2], 0, inplace=True)
        assert L.fourier_emu_alloc_count() == before, n
        assert np.array_equal(got, ref) and np.array_equal(got1, ref[:2])
    assert plan2.device == 0


def test_device_sharded_driver_runs_every_shard_on_its_own_thread(fa, oracle):
    """fourier_amd.shard.DeviceShardedFft (in-process form): one plan + one host thread per shard, the
    global batch split with batch_shard.  The emulator has one device, so both shards name device 0."""
    from fourier_amd import shard

    n, gbatch = 1000, 11
    x = hash_normal(31, gbatch * n).astype(np.complex64).reshape(gbatch, n)
    y = np.empty_like(x)
    drv = shard.DeviceShardedFft(n, "f32", [0, 0, 0])
    ins, outs = [], []
    for g in range(3):
        lo, hi = shard.batch_shard(gbatch, 3, g)
        ins.append((x[lo:hi].ctypes.data, hi - lo))
        outs.append((y[lo:hi].ctypes.data, hi - lo))
    drv.transform(ins, outs, fa.Transform.Fft)
    assert rel_l2(y, oracle.transform_batch(x, oracle.FFT)) <= 2e-6
    with pytest.raises(ValueError):
        drv.transform(ins[:2], outs, fa.Transform.Fft)


@pytest.mark.parametrize("dtype,k", [(np.complex64, 16), (np.complex64, 17), (np.complex128, 15), (np.complex128, 17)])
def test_xcd_fused_one_launch_plan_equals_the_two_launch_plan(fa, monkeypatch, dtype, k):
    """fft_l2fused_kernel (plan option l2_fused): persistent workgroups pulling (transform, pass, tile) items from the
    queue of the XCD they run on, inter-workgroup waits through counters.  The emulator runs the blocks concurrently on
    host threads with real atomics and hands out pretend XCC ids, so the scheduling protocol itself is exercised:
    bit-identical to the two-launch plan for 1, 3 and 8 pretend XCDs, in and out of place, ragged batches."""
    n = 1 << k
    x = hash_normal(k, 7 * n).astype(dtype).reshape(7, n)
    two = make(fa, n, dtype)
    ref = run_batch(two, x, 1)
    for xcds in ("1", "3", "8"):
        monkeypatch.setenv("HIPEMU_XCDS", xcds)
        one = make(fa, n, dtype)
        one.set_option("l2_fused", 1)
        one.set_option("l2_fused_depth", 1 + int(xcds) % 3)
        assert "xcd-l2" in one.describe()
        assert np.array_equal(run_batch(one, x, 1), ref), xcds
        assert np.array_equal(run_batch(one, x, 1, inplace=True), ref), xcds
        assert np.array_equal(run_batch(one, x[:1], 1), ref[:1]), xcds
    with pytest.raises(fa.FourierError):
        make(fa, 1 << 12, dtype).set_option("l2_fused", 1)


def test_l2048_narrow_first_pass_and_split_last_pass(fa, oracle, monkeypatch):
    """2^21 = 2048 x 1024 and 2^22 = 2048 x 2048.  Default plans run the first pass of length 2048 on 64-byte-wide
    tiles: bit-identical to the 16-column kernel (FOURIER_WIDE_2048=1).  The half-tile last pass (FOURIER_SPLIT_2048=1,
    radix-2 decimation in frequency + a 1024-point tile per workgroup, routed through the scratch because two
    workgroups read each column tile) is kept as an experiment: one extra rounding, checked against both."""
    for n in (1 << 21, 1 << 22):
        x = hash_normal(n % 1000, n).astype(np.complex64)[None, :]
        new = make(fa, n, np.complex64)
        monkeypatch.setenv("FOURIER_WIDE_2048", "1")
        old = make(fa, n, np.complex64)
        monkeypatch.delenv("FOURIER_WIDE_2048")
        monkeypatch.setenv("FOURIER_SPLIT_2048", "1")
        split = make(fa, n, np.complex64)
        monkeypatch.delenv("FOURIER_SPLIT_2048")
        yn, yo, ys = run_batch(new, x, 0), run_batch(old, x, 0), run_batch(split, x, 0)
        assert np.array_equal(yn, yo), n
        assert np.array_equal(run_batch(new, x, 0, inplace=True), yn), n
        assert rel_l2(yn, oracle.transform_batch(x, oracle.FFT)) <= 1e-6, n
        if n == 1 << 22:
            assert rel_l2(ys, yo) < 3e-7 and np.array_equal(run_batch(split, x, 0, inplace=True), ys)
        else:
            assert np.array_equal(ys, yo)  # no last pass of length 2048 in this plan


def test_empty_batch_is_a_successful_no_op_for_every_plan_family(fa):
    """batch = 0 through the batched entry points (device-resident, host-streamed, reserve): returns OK, launches
    nothing, touches no byte -- for every plan family, including the opt-in XCD-fused one."""
    from fourier_amd import _lib

    L = _lib.lib()
    for n, opts in ((8, ()), (1024, ()), (4096, ()), (1 << 16, ()), (1 << 16, (("l2_fused", 1),)), (96, ()), (3 * 4096, ()), (100, ()), (40001, ())):
        plan = make(fa, n, np.complex64)
        for k, v in opts:
            plan.set_option(k, v)
        buf = np.full(n, 7 + 7j, np.complex64)
        assert L.fourier_hip_transform_batch_float(plan._h, buf.ctypes.data, buf.ctypes.data, 0, 0, None) == 0, n
        assert L.fourier_hip_transform_batch_host_float(plan._h, buf.ctypes.data, buf.ctypes.data, 0, 0) == 0, n
        assert L.fourier_hip_reserve_float(plan._h, 0, 1) == 0, n
        assert L.fourier_hip_last_status_float(plan._h) == 0
        assert (buf == 7 + 7j).all(), n


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_a_non_finite_transform_does_not_reach_its_neighbours(fa, oracle, dtype):
    """Transforms of a batch are independent (the reference runs one plan call per transform, fft.rs:51-61): a row of NaN /
    Inf poisons its own output only.  Covers every plan family that puts several transforms into one workgroup or pads a
    transform (the one-launch chirp-z read its padding from the next transform's row and relied on 0 * x)."""
    for n in (8, 17, 64, 96, 127, 439, 625, 1000, 1013, 2048, 3001):
        batch = 6
        x = np.stack([hash_normal(900 + b, n) for b in range(batch)]).astype(dtype)
        ref = oracle.transform_batch(x, 0)
        for bad_row, bad in ((1, np.nan), (batch - 1, np.inf), (0, -np.inf)):
            xb = x.copy()
            xb[bad_row, n // 2] = bad
            got = run_batch(make(fa, n, dtype), xb, 0)
            keep = [b for b in range(batch) if b != bad_row]
            assert np.isfinite(got[keep]).all(), (n, bad_row)
            assert rel_l2(got[keep], ref[keep]) <= (2e-6 if dtype == np.complex64 else 2e-12), (n, bad_row)
            assert not np.isfinite(got[bad_row]).all(), (n, bad_row)


def test_prefetching_last_pass_is_bit_identical_to_the_plain_last_pass(fa, oracle):
    """Plan option last_pass_prefetch (fft_last_prefetch_kernel, experiments build): the LAST pass as persistent workgroups that
    fetch their next tile -- eight rows by LDS-DMA into the idle exchange buffer, eight into registers -- ahead of the current
    tile's stores.  Same in-tile arithmetic: under the emulator (one compiler, no FMA contraction) the same bits as fft_pass_kernel, for the plain last pass (L = 1024 and 2048, f32 and f64, with
    and without the stage twiddles in LDS) and for the chirp-out pass of a Bluestein plan; ragged tile counts per workgroup
    (the emulated device keeps 6 workgroups resident)."""
    # (the kernel is a measured-slower experiment, kept for A/B: one case per form -- L = 1024 f32 with a ragged tile count and both
    # directions, L = 1024 f64 behind a 2048-point first pass, the chirp-out form; L = 2048 runs on the GPU, tests/test_gpu_parity.py)
    for n, dtype, batch, codes in ((1 << 20, np.complex64, 3, (0, 4)), (1 << 21, np.complex128, 1, (0,)), (300007, np.complex64, 1, (0,))):
        x = np.stack([hash_normal(300 + b, n) for b in range(batch)]).astype(dtype)
        on, off = make(fa, n, dtype), make(fa, n, dtype)
        on.set_option("last_pass_prefetch", 1)
        off.set_option("last_pass_prefetch", 0)
        for code in codes:
            a, b = run_batch(on, x, code), run_batch(off, x, code)
            assert np.array_equal(a, b), (n, dtype, code)
        first = run_batch(on, x, 0)
        assert np.array_equal(run_batch(on, x, 0, inplace=True), first), (n, dtype)
        tol = (1e-6 if n & (n - 1) == 0 else 2e-6) if dtype == np.complex64 else (5e-14 if n & (n - 1) == 0 else 1e-9)
        assert rel_l2(first, oracle.transform_batch(x, 0)) <= tol, (n, dtype)


def test_plan_option_specialise_is_refused_without_hiprtc_and_leaves_the_plan_alone(fa, oracle):
    """Under the emulator there is no hipRTC: "specialise" reports UNSUPPORTED (as it does on a box without libhiprtc) for a
    length of the family and for one outside it, and the plan keeps its route and its results; a length that already runs a
    per-length kernel -- LDS (1000) or register stages (1001, round 6) -- returns OK."""
    for n in (3003, 11011, 1013):
        plan = make(fa, n, np.complex64)
        desc = plan.describe()
        with pytest.raises(fa.FourierError):
            plan.set_option("specialise", 1)
        assert plan.describe() == desc
        x = np.stack([hash_normal(3 + b, n) for b in range(3)]).astype(np.complex64)
        assert rel_l2(run_batch(plan, x, 0), oracle.transform_batch(x, 0)) <= 2e-6
    for n in (1000, 1001):
        plan = make(fa, n, np.complex64)
        desc = plan.describe()
        plan.set_option("specialise", 1)
        assert plan.describe() == desc and "specialised" not in desc and ("registers" in desc) == (n == 1001)


def test_every_route_string_describe_can_return_is_named_in_the_public_header(fa, monkeypatch):
    """include/fourier.h tells callers to match on fourier_hip_describe_* where the accuracy class of a route matters (VERDICT round 4
    item 7: the list there had fallen behind plan.h twice).  Every word of every description the plan factory returns -- one length
    per route of Plan::Plan, both precisions -- must occur in the header's route list."""
    import re

    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fourier.h")).read()
    routes = header[header.index("The routes, in the order they are tried"):header.index("struct fourier_fft_float *fourier_hip_create_float")]
    sizes = [4, 16, 64, 1024, 1 << 12, 1 << 14, 1 << 16, 1 << 23,  # tiny, whole rows, one-launch, two and three passes
             3 * 4096, 27 * 4096, 512 * 432, 3 << 18,              # mixed tiles (a >= 12), power-of-two passes + odd passes
             96, 1000, 3003, 18432,                                 # LDS mixed-radix: per-length and runtime-parameterised kernels
             1001, 350,                                             # register stages (three, two)
             62208, 3 ** 10 * 2, 3 ** 16, 625 * 625,                # mixed tiles (a < 12), three tile passes, tiles beyond 512 points
             100000, 44100,                                         # tile passes with factors 5 / 7
             17, 1013, 40001, 999983, 16411]                        # Bluestein: one-launch ("fused"), fused passes, smooth M
    seen = set()

    def visit(n):
        for dtype in (np.complex64, np.complex128):
            d = make(fa, n, dtype).describe()
            seen.add(re.sub(r"\d+", "#", d))
            for word in re.findall(r"[a-z][a-z\-]{2,}", d):
                assert word in routes, (word, d)

    for n in sizes:
        visit(n)
    # one Stockham pass per radix in global memory: since round 6 (tile lengths up to 1024) every 2^a 3^b, a < 12, the engine accepts has a tile
    # factorisation -- the route is what is left when that one is switched off (experiments library; the emulator build is one)
    monkeypatch.setenv("FOURIER_NO_TILED_MIXED", "1")
    visit(59049)
    monkeypatch.delenv("FOURIER_NO_TILED_MIXED")
    # every route family was actually visited
    for must in ("stockham tiny", "one-launch", "mixed tiles", "mixed-radix", "global-pass", "bluestein M=", "fused", "registers"):
        assert any(must.replace("M=", "M=") in s for s in seen), (must, sorted(seen))
