#!/bin/bash
# CPU only: runs the emulator-backed engine tests (tests/test_engine_emu.py) against an AddressSanitizer build of the emulator library
# (tests/emu/build_emu.py under FOURIER_EMU_ASAN=1: g++ -fsanitize=address over the same translation units).  Every global-memory,
# LDS and table access of every kernel the tests launch, and every host buffer of the plan layer, is bounds-checked: the out-of-bounds
# read of the inter-pass twiddle table that ADVICE round 4 found by hand (kernels_tiled.h, masked columns of a ragged tile) is reported
# here as "heap-buffer-overflow ... tiled_mixed_kernel_ct<float, 243u> kernels_tiled.h:75" when the clamp is taken out again.
# libstdc++ is preloaded next to libasan: python does not link it, and ASan's __cxa_throw interceptor needs the real one at start-up.
# About a minute of build on 8 cores and two to five minutes of tests on six workers.  UBSAN=1 adds -fsanitize=undefined (index arithmetic:
# shifts, signed overflow, misaligned accesses; nine minutes).  usage: [UBSAN=1] bash tools/asan_emu.sh [pytest args]
set -u
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
STDCXX=$(gcc -print-file-name=libstdc++.so.6)
[ -f "$STDCXX" ] || STDCXX=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
rm -f /tmp/fourier_asan.*
MODE=1; PRE="$ASAN $STDCXX"
if [ "${UBSAN:-0}" = "1" ]; then MODE=2; PRE="$ASAN $(gcc -print-file-name=libubsan.so) $STDCXX"; fi
FOURIER_EMU_ASAN=$MODE LD_PRELOAD="$PRE" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:log_path=/tmp/fourier_asan UBSAN_OPTIONS=log_path=/tmp/fourier_asan \
  python -m pytest tests/test_engine_emu.py -q -m "not gpu" -p no:cacheprovider "${@:--n 6}"
rc=$?
if ls /tmp/fourier_asan.* > /dev/null 2>&1; then echo "AddressSanitizer reports:"; head -30 /tmp/fourier_asan.*; exit 1; fi
# (the sanitizer objects must not stay in the tree: gpurun refuses a snapshot that holds -fsanitize=address objects under tests/)
rm -rf tests/emu/obj_asan* tests/emu/libfourier_emu_asan*
echo "AddressSanitizer${UBSAN:+ / UndefinedBehaviorSanitizer}: no report"; exit $rc
