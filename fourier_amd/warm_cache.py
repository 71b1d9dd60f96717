"""python -m fourier_amd.warm_cache [max_length | length ...]

Fills the on-disk code-object cache of libfourier.so ahead of time (include/fourier.h: run-time specialisation): every length up to
`max_length` (default 4096) whose prime factors stop at 13, both precisions, created once under the library-wide policy
"specialise_at_create" = 2, which compiles the length's own kernel where the cache has none.  Afterwards a plain create_fft_f32/f64 --
the reference's implicit plan choice, fourier/src/lib.rs:38-42 -- of such a length loads its kernel from the cache in milliseconds
instead of running the runtime-parameterised kernel (or Bluestein).  About a second per length and precision that has no ahead-of-time
kernel; needs a GPU and libhiprtc.  The same as the C program packaging/warm_cache.c (CMake target `warm_cache`)."""
import sys


def smooth13(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


def warm(lengths, verbose=False):
    from . import fft

    fft.set_default_option("specialise_at_create", 2)
    own = 0
    for n in lengths:
        for make in (fft.create_fft_f32, fft.create_fft_f64):
            d = make(n).describe()
            own += "specialised" in d
            if verbose:
                print(n, d)
    return own


def main(argv):
    args = [int(a) for a in argv]
    if len(args) > 1 or (len(args) == 1 and not smooth13(args[0])):
        lengths, verbose = args, True
    else:
        lengths, verbose = [n for n in range(2, (args[0] if args else 4096) + 1) if smooth13(n)], False
    own = warm(lengths, verbose)
    print(f"fourier_amd.warm_cache: {len(lengths)} lengths, {own} plans run on kernels of their own from the cache")


if __name__ == "__main__":
    main(sys.argv[1:])
