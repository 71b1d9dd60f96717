"""Pins the CPU oracle (oracle/fourier_oracle.cpp) against everything the reference's own tests
hold for this path and against independent numpy-float64 fixtures."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, hash_normal, hash_uniform, load_ref10, naive_dft, near, rel_l2, max_rel, sample_bins

F32_EPS, F64_EPS = 1e-4, 1e-11  # fourier/tests/integrity.rs:92,120


def test_radix_schedule_matches_reference_table(oracle):
    # autosort/mod.rs:104-116; SURVEY.md section 3.2 schedule table
    assert oracle.radix_counts(4096) == [1, 3, 0, 0, 1]
    assert oracle.radix_counts(1 << 20) == [1, 6, 0, 0, 0]
    assert oracle.radix_counts(1 << 21) == [1, 6, 0, 0, 1]
    assert oracle.radix_counts(1 << 22) == [1, 6, 1, 0, 0]
    assert oracle.radix_counts(1) == [0, 0, 0, 0, 0]
    assert oracle.radix_counts(6) == [0, 0, 0, 1, 1]
    assert oracle.radix_counts(999983) is None  # prime -> Bluestein
    assert oracle.radix_counts(10) is None


@pytest.mark.parametrize("dtype,eps", [(np.complex64, F32_EPS), (np.complex128, F64_EPS)])
def test_reference_golden_vector(oracle, dtype, eps):
    """integrity.rs:48-72: the only literal known-answer data in the reference.  N=10 is not
    {2,3}-smooth so this drives the Bluestein path (M=32)."""
    x, y = load_ref10()
    # the reference checks its naive DFT against the literals both ways (integrity.rs:77-81)
    ok, _ = near(naive_dft(x.astype(dtype)), y, eps)
    assert ok
    ok, _ = near(naive_dft(y.astype(dtype), inverse=True), x, eps)
    assert ok
    f = oracle.OracleFft(10, dtype)
    ok, worst = near(f.transform(x.astype(dtype), oracle.FFT), y, eps)
    assert ok, worst
    ok, worst = near(f.transform(y.astype(dtype), oracle.IFFT), x, eps)
    assert ok, worst


@pytest.mark.parametrize("dtype,eps", [(np.complex64, F32_EPS), (np.complex128, F64_EPS)])
@pytest.mark.parametrize("forward", [True, False])
def test_sweep_1_255_reference_procedure(oracle, dtype, eps, forward):
    """integrity.rs:145-192: sizes 1..255, prefix of a 256-sample vector (sigma 1 fwd / 256 inv),
    FFT vs naive DFT at the reference tolerance; plus the independent numpy-f64 fixture."""
    g = np.load(os.path.join(GOLDEN, "sweep_1_255.npz"))
    x = (g["x_fwd"] if forward else g["x_inv"]).astype(dtype)
    y64 = g["y_fwd"] if forward else g["y_inv"]
    code = oracle.FFT if forward else oracle.IFFT
    off = 0
    for n in range(1, 256):
        f = oracle.OracleFft(n, dtype)
        got = f.transform(x[:n], code)
        ok, worst = near(naive_dft(x[:n], inverse=not forward), got, eps)
        assert ok, (n, worst)
        want = y64[off:off + n]
        off += n
        scale = max(np.abs(want).max(), 1.0)
        tol = (3e-6 if dtype == np.complex64 else 1e-12) * scale
        assert np.abs(got - want).max() <= tol, (n, np.abs(got - want).max(), tol)


@pytest.mark.parametrize("n", [64, 73])
def test_static_fft_sizes(oracle, n):
    # integrity.rs:234-254
    for dtype, eps in ((np.complex64, F32_EPS), (np.complex128, F64_EPS)):
        x = hash_normal(77, n).astype(dtype)
        for code, inv in ((oracle.FFT, False), (oracle.IFFT, True)):
            ok, worst = near(naive_dft(x, inverse=inv), oracle.OracleFft(n, dtype).transform(x, code), eps)
            assert ok, (n, worst)


@pytest.mark.parametrize("dtype", [np.complex64, np.complex128])
def test_ffi_impulse_roundtrip(oracle, dtype):
    # fourier-ffi/test.c:7-39: [1,0,0,0] -> transform FFT -> transform_in_place IFFT == input (1e-10)
    f = oracle.OracleFft(4, dtype)
    x = np.array([1, 0, 0, 0], dtype=dtype)
    y = f.transform(x, oracle.FFT)
    assert np.allclose(y, np.ones(4))
    f.transform_in_place(y, oracle.IFFT)
    assert np.abs(y - x).max() <= 1e-10


@pytest.mark.parametrize("n", [1, 2, 6, 10, 96, 100, 256])
def test_all_transform_scalings(oracle, n):
    """fft.rs:4-16 definitions of the 5 Transform codes (the reference tests only 2 of them)."""
    x = hash_normal(5, n)
    F = np.fft.fft(x)
    Fi = np.fft.ifft(x)
    want = {
        oracle.FFT: F, oracle.IFFT: Fi, oracle.UNSCALED_IFFT: Fi * n,
        oracle.SQRT_SCALED_FFT: F / np.sqrt(n), oracle.SQRT_SCALED_IFFT: Fi * np.sqrt(n),
    }
    for dtype, tol in ((np.complex64, 2e-6), (np.complex128, 2e-13)):
        f = oracle.OracleFft(n, dtype)
        for code, w in want.items():
            got = f.transform(x.astype(dtype), code)
            assert max_rel(got, w) <= tol, (n, code, max_rel(got, w))
            buf = x.astype(dtype).copy()
            f.transform_in_place(buf, code)
            assert np.array_equal(buf, got)


def test_unknown_transform_code_is_noop(oracle):
    # fourier-ffi/src/lib.rs:10: unknown code panics before the call -> buffers untouched
    f = oracle.OracleFft(8, np.complex64)
    x = hash_normal(1, 8).astype(np.complex64)
    buf = x.copy()
    f.transform_in_place(buf, 7)
    assert np.array_equal(buf, x)


def test_size_zero_is_invalid(oracle):
    with pytest.raises(ValueError):
        oracle.OracleFft(0, np.complex64)


def test_n4096_full_spectrum(oracle):
    g = np.load(os.path.join(GOLDEN, "n4096.npz"))
    x = hash_normal(int(g["seed"]), 4096)
    for dtype, tl2, tmax in ((np.complex64, 1e-6, 2e-6), (np.complex128, 5e-14, 1e-13)):
        got = oracle.OracleFft(4096, dtype).transform(x.astype(dtype), oracle.FFT)
        assert rel_l2(got, g["y"]) <= tl2 and max_rel(got, g["y"]) <= tmax


@pytest.mark.parametrize("n,dtype,tl2,tmax", [
    (1 << 20, np.complex64, 1e-6, 2e-6),
    (1 << 20, np.complex128, 5e-14, 1e-13),
    (999983, np.complex64, 2e-6, 4e-6),
])
def test_baseline_sizes_sampled_bins(oracle, n, dtype, tl2, tmax):
    """BASELINE configs C2/C3/C4 sizes: oracle vs committed numpy-f64 sampled bins + L2 norm."""
    g = np.load(os.path.join(GOLDEN, "big_samples.npz"))
    x = hash_uniform(int(g[f"seed_{n}"]), n).astype(dtype)
    got = oracle.OracleFft(n, dtype).transform(x, oracle.FFT)
    bins = g[f"bins_{n}"]
    assert np.array_equal(bins, sample_bins(n))
    err = np.abs(got[bins] - g[f"y_{n}"]).max() / float(g[f"maxabs_{n}"])
    assert err <= tmax, err
    assert abs(np.linalg.norm(got.astype(np.complex128)) - float(g[f"l2_{n}"])) <= tl2 * float(g[f"l2_{n}"])


def test_batch_threads_match_single(oracle):
    x = np.stack([hash_normal(100 + b, 96) for b in range(7)]).astype(np.complex64)
    a = oracle.transform_batch(x, oracle.FFT, 1)
    b = oracle.transform_batch(x, oracle.FFT, 3)
    assert np.array_equal(a, b)
    f = oracle.OracleFft(96, np.complex64)
    assert np.array_equal(a[4], f.transform(x[4], oracle.FFT))


def test_twiddle_table_lengths_match_the_reference_layout(oracle):
    """autosort/mod.rs:24-46: one direction's table holds size_cur entries per pass (m rows of `radix`).  SURVEY.md
    section 3.2 derives 5 266 / 1 348 168 / 2 696 338 / 5 392 676 entries for N = 4096 / 2^20 / 2^21 / 2^22."""
    assert oracle.table_len(4096) == 5266
    assert oracle.table_len(1 << 20) == 1348168
    assert oracle.table_len(1 << 21) == 2696338
    assert oracle.table_len(1 << 22) == 5392676
    assert oracle.table_len(1) == 0 and oracle.table_len(2) == 2 and oracle.table_len(6) == 6 + 2
    assert oracle.table_len(999983) is None
    for n in (12, 96, 243, 1000 * 0 + 972, 4096, 18432):  # independent count: sum of size_cur over the schedule
        counts, cur, total = oracle.radix_counts(n), n, 0
        for radix, c in zip((4, 8, 4, 3, 2), counts):
            for _ in range(c):
                total += cur
                cur //= radix
        assert oracle.table_len(n) == total and cur == 1, n


def _numpy_stockham_pass(x, radix, size, stride, forward):
    """Independent restatement of ONE pass, autosort/mod.rs:203-284, from its index formula:
    out[j + R*s*i + s*k] = W_size^{i*k} * sum_k' omega_R^{k*k'} * in[j + s*i + s*m*k'],  i < m = size/R, j < s."""
    m = size // radix
    sgn = -1.0 if forward else 1.0
    X = x.reshape(radix, m, stride)                                    # [k'][i][j]
    kk = np.arange(radix)
    omega = np.exp(sgn * 2j * np.pi * np.outer(kk, kk) / radix)        # DFT_R
    Y = np.einsum("kq,qij->ikj", omega, X)                             # [i][k][j]
    if size != radix:                                                  # mod.rs:238,272: no twiddle on the last pass
        W = np.exp(sgn * 2j * np.pi * np.outer(np.arange(m), kk) / size)
        Y = Y * W[:, :, None]
    return Y.reshape(-1)


@pytest.mark.parametrize("radix,size,stride", [(2, 2, 96), (2, 16, 3), (3, 9, 4), (3, 3, 32), (4, 64, 1), (4, 4, 8),
                                               (8, 512, 1), (8, 64, 4), (8, 8, 16), (4, 16, 3), (8, 24 * 0 + 8, 5)])
def test_one_pass_against_an_independent_numpy_restatement(oracle, radix, size, stride):
    """Pass-by-pass pin of radix_pass (generic AND AVX clone, narrow and wide forms) against numpy in f64."""
    x = hash_normal(1000 + radix * size + stride, size * stride)
    for forward in (True, False):
        want = _numpy_stockham_pass(x, radix, size, stride, forward)
        for clone in (oracle.GENERIC, oracle.AVX):
            got = oracle.radix_pass(radix, x, forward, size, stride, clone)
            assert rel_l2(got, want) < 1e-14, (radix, size, stride, forward, clone)


def test_avx_clone_is_bit_identical_to_the_generic_functions(oracle):
    """vector/avx.rs + autosort/avx_optimization.rs:4-92 restated with intrinsics: same operations, same roundings.
    Values must be equal for every size class (pure pow2 with the f32 radix-4 stride-1 first pass, odd first
    passes, 2^a*3^b, Bluestein) in both precisions and directions."""
    if not oracle.have_avx():
        pytest.skip("oracle built without AVX")
    sizes = [4, 8, 16, 32, 64, 128, 256, 1024, 4096, 1 << 15, 6, 12, 24, 48, 96, 243, 768, 972, 7, 100, 1000]
    try:
        for dtype in (np.complex64, np.complex128):
            for n in sizes:
                x = hash_normal(n, n).astype(dtype)
                outs = []
                for clone in (oracle.GENERIC, oracle.AVX_NO_FIRST_PASS, oracle.AVX):
                    assert oracle.set_clone(clone) == clone
                    f = oracle.OracleFft(n, dtype)
                    outs.append((f.transform(x, oracle.FFT), f.transform(x, oracle.IFFT)))
                for o in outs[1:]:
                    assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]), (n, dtype)
    finally:
        oracle.set_clone(oracle.AVX)


REF = "/root/reference"


@pytest.mark.parametrize("src,cc,std", [("test.c", "gcc", "-std=c11"), ("test.cpp", "g++", "-std=c++14")])
def test_reference_consumers_compile_unchanged_against_our_header(tmp_path, src, cc, std):
    """The reference's own C and C++ consumers (fourier-ffi/test.c:7-46, test.cpp) must compile UNCHANGED against
    include/fourier.h with the reference's warning flags as errors.  Compile-only (no GPU here); the files are read
    where they lie -- nothing is copied into the repo.  Skipped where the reference tree does not exist (GPU box)."""
    import subprocess

    path = os.path.join(REF, "fourier-ffi", src)
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
