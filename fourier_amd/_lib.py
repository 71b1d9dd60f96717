"""ctypes binding of the C ABI declared in include/fourier.h (libfourier.so).

There is deliberately no CPU fallback: if the HIP library has not been built, importing the
operator layer raises.  (The reference's FFI, fourier-ffi/src/lib.rs:14-106, is bound the other way
round; see INTEGRATION.md for the Rust shim.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfourier.so")

SUFFIXES = ("float", "double")

# every symbol include/fourier.h declares
LEGACY_SYMBOLS = [f"fourier_{op}_{s}" for s in SUFFIXES for op in ("create", "destroy", "transform_in_place", "transform")]
EXT_SYMBOLS = [f"fourier_hip_{op}_{s}" for s in SUFFIXES
               for op in ("create", "size", "transform_batch", "reserve", "device", "synchronize", "transform_batch_host", "last_status", "set_option", "describe", "model_bytes",
                          "profile", "slot_names")] + [
    "fourier_hip_status_string", "fourier_hip_set_default_option", "fourier_hip_get_default_option"]
ALL_SYMBOLS = LEGACY_SYMBOLS + EXT_SYMBOLS


def bind(cdll, strict=True):
    """Attach argtypes/restypes for every entry point of include/fourier.h to a loaded CDLL.  strict=False (A/B tools that
    load libraries built from older sources) tolerates entry points added since."""
    vp, sz, ci, ll, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_longlong, ctypes.c_char_p
    for s in SUFFIXES:
        f = getattr(cdll, f"fourier_create_{s}"); f.restype = vp; f.argtypes = [sz]
        f = getattr(cdll, f"fourier_destroy_{s}"); f.restype = None; f.argtypes = [vp]
        f = getattr(cdll, f"fourier_transform_in_place_{s}"); f.restype = None; f.argtypes = [vp, vp, ci]
        f = getattr(cdll, f"fourier_transform_{s}"); f.restype = None; f.argtypes = [vp, vp, vp, ci]
        f = getattr(cdll, f"fourier_hip_create_{s}"); f.restype = vp; f.argtypes = [sz, ci]
        f = getattr(cdll, f"fourier_hip_size_{s}"); f.restype = sz; f.argtypes = [vp]
        f = getattr(cdll, f"fourier_hip_transform_batch_{s}"); f.restype = ci; f.argtypes = [vp, vp, vp, sz, ci, vp]
        f = getattr(cdll, f"fourier_hip_transform_batch_host_{s}"); f.restype = ci; f.argtypes = [vp, vp, vp, sz, ci]
        f = getattr(cdll, f"fourier_hip_reserve_{s}"); f.restype = ci; f.argtypes = [vp, sz, ci]
        f = getattr(cdll, f"fourier_hip_device_{s}"); f.restype = ci; f.argtypes = [vp]
        if strict or hasattr(cdll, f"fourier_hip_synchronize_{s}"):
            f = getattr(cdll, f"fourier_hip_synchronize_{s}"); f.restype = ci; f.argtypes = [vp, vp]
        f = getattr(cdll, f"fourier_hip_last_status_{s}"); f.restype = ci; f.argtypes = [vp]
        f = getattr(cdll, f"fourier_hip_set_option_{s}"); f.restype = ci; f.argtypes = [vp, cp, ll]
        f = getattr(cdll, f"fourier_hip_describe_{s}"); f.restype = cp; f.argtypes = [vp]
        f = getattr(cdll, f"fourier_hip_model_bytes_{s}"); f.restype = ctypes.c_double; f.argtypes = [vp]
        f = getattr(cdll, f"fourier_hip_profile_{s}"); f.restype = ci
        f.argtypes = [vp, vp, vp, sz, ci, vp, ci, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ci)]
        f = getattr(cdll, f"fourier_hip_slot_names_{s}"); f.restype = cp; f.argtypes = [vp]
    cdll.fourier_hip_status_string.restype = cp
    cdll.fourier_hip_status_string.argtypes = [ci]
    if strict or hasattr(cdll, "fourier_hip_set_default_option"):
        cdll.fourier_hip_set_default_option.restype = ci
        cdll.fourier_hip_set_default_option.argtypes = [cp, ll]
        cdll.fourier_hip_get_default_option.restype = ll
        cdll.fourier_hip_get_default_option.argtypes = [cp]
    return cdll


_lib = None


def lib():
    """The product library.  Raises if the HIP build is missing (no fallback by design)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP engine first "
                "(python -c 'import __graft_entry__ as g; g.build()' or python -m fourier_amd.build)")
        try:  # when torch is around, load it first so both share one HIP runtime (same SONAME)
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = bind(ctypes.CDLL(LIB_PATH))
    return _lib
