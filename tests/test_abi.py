"""The C-ABI library builds for gfx950 and exports every symbol include/fourier.h declares
(no compute calls: this runs without a GPU)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from fourier_amd import build

    return build.build()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "fourier.h")).read()
    text = text[: text.index("Header-only C++ RAII wrapper")]
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fourier_(?:hip_)?[a-z_]+_(?:float|double)|fourier_hip_status_string|fourier_hip_[sg]et_default_option)\s*\(", text)))


def test_header_declares_the_reference_abi():
    syms = declared_symbols()
    # the 8 entry points of the reference header, fourier-ffi/include/fourier.h:41-58
    for s in ("float", "double"):
        for op in ("create", "destroy", "transform_in_place", "transform"):
            assert f"fourier_{op}_{s}" in syms


def test_library_exports_every_declared_symbol(libpath):
    out = subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing


def test_library_loads_and_every_symbol_resolves_without_a_gpu(libpath):
    """dlopen + dlsym of every declared entry point (no compute: this container has no GPU).  Without a
    device, plan creation must fail the way the reference's FFI does on a panic: NULL, no abort
    (fourier-ffi/src/lib.rs:18-19), and the product layer must raise instead of falling back to the CPU."""
    import ctypes

    from fourier_amd import _lib

    cdll = _lib.bind(ctypes.CDLL(libpath))
    for sym in declared_symbols():
        assert getattr(cdll, sym) is not None
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert not cdll.fourier_create_float(8) and not cdll.fourier_create_double(8)
        assert cdll.fourier_hip_size_float(None) == 0
        cdll.fourier_destroy_float(None)
        import fourier_amd

        with pytest.raises(fourier_amd.FourierError):
            fourier_amd.create_fft_f32(8)


def test_python_binding_lists_the_same_symbols():
    from fourier_amd import _lib

    assert sorted(_lib.ALL_SYMBOLS) == declared_symbols()


def test_library_is_gfx950_only(libpath):
    # the fat binary embeds one code object per offload target: only gfx950 may be present
    blob = open(libpath, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_soname_matches_reference_packaging(libpath):
    # fourier-ffi/CMakeLists.txt:55 / config.toml:13-14: SONAME libfourier.so.0
    out = subprocess.run(["objdump", "-p", libpath], capture_output=True, text=True, check=True).stdout
    assert "libfourier.so.0" in out


def test_transform_enum_matches_reference():
    from fourier_amd import Transform

    # fourier-ffi/src/lib.rs:3-12
    assert [int(t) for t in (Transform.Fft, Transform.Ifft, Transform.UnscaledIfft, Transform.SqrtScaledFft,
                             Transform.SqrtScaledIfft)] == [0, 1, 2, 3, 4]
    # fft.rs:20-36
    assert Transform.Fft.is_forward() and Transform.SqrtScaledFft.is_forward()
    assert not Transform.Ifft.is_forward() and not Transform.UnscaledIfft.is_forward()
    assert Transform.Fft.inverse() is Transform.Ifft and Transform.UnscaledIfft.inverse() is None
    assert Transform.SqrtScaledIfft.inverse() is Transform.SqrtScaledFft


def test_rust_shim_names_only_exported_symbols(libpath):
    """rust/fourier-hip/src/lib.rs cannot be compiled here (no rustc): at least every `fn fourier_*` of its
    `extern "C"` block must be a symbol of the library and be declared in include/fourier.h."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rust", "fourier-hip", "src", "lib.rs")).read()
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n    }")]
    names = set(re.findall(r"fn (fourier_\w+)\(", block))
    assert len(names) == 22, sorted(names)
    exported = set(subprocess.run(["nm", "-D", "--defined-only", libpath], capture_output=True, text=True).stdout.split())
    header = open(os.path.join(root, "include", "fourier.h")).read()
    for n in sorted(names):
        assert n in exported, n
        assert re.search(r"\b%s\(" % n, header), n


def test_cmake_package_configures(tmp_path):
    """packaging/CMakeLists.txt (shared + static library from one object library, five ctest programs) must at least
    configure with the ROCm toolchain of this image.  Building it compiles the engine once more (~2 min here) and its
    programs need a GPU: the build + ctest run is tests/test_gpu_parity.py::test_cmake_package_builds_and_its_ctest_programs_pass."""
    import shutil
    import subprocess

    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("cmake / ROCm clang not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["cmake", "-S", os.path.join(root, "packaging"), "-B", str(tmp_path / "b"),
                        "-DCMAKE_HIP_COMPILER=/opt/rocm/lib/llvm/bin/clang++", "-DCMAKE_PREFIX_PATH=/opt/rocm"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    targets = subprocess.run(["cmake", "--build", str(tmp_path / "b"), "--target", "help"], capture_output=True, text=True).stdout
    for t in ("fourier", "fourier_static", "consumer_c", "consumer_cxx", "consumer_c_static", "consumer_cxx_static", "ffi_sequence"):
        assert t in targets, t
