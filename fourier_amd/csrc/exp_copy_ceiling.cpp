// exp_copy_ceiling.cpp -- what a plain device copy reaches on THIS box, for bench.py's roofline record (VERDICT round 3:
// the denominator that explains "each pass at the streaming ceiling" must be measured in the same run on the same
// buffers, not quoted from DESIGN.md).  Linked into lib/libfourier_experiments.so only: measurement tooling, not product.
//
// The copy has the pass kernels' memory shape without their arithmetic: every workgroup owns one contiguous slab, 16-byte
// accesses, eight loads in flight per thread, and the XCD-aware block mapping of the passes (block b runs on XCD b % 8;
// every XCD walks its own contiguous eighth of the buffer, so each 2 MiB page is touched by one XCD) -- the `slab_x` form of
// tools/membench.hip, the best of the copy forms measured in round 2 (profiles/r02_membench.jsonl: 5.87 TB/s).
#include "engine_common.h"
#include "kernels_common.h"

namespace fourier_hip {

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_slab_kernel(const void* __restrict__ src, void* __restrict__ dst, uint64_t units_per_block) {
  uint64_t b = blockIdx.x;
  const uint64_t per_xcd = gridDim.x / 8;
  if (per_xcd * 8 == gridDim.x) b = (b % 8) * per_xcd + b / 8;
  const Unit16<float>* s = (const Unit16<float>*)src + b * units_per_block;
  Unit16<float>* d = (Unit16<float>*)dst + b * units_per_block;
  for (uint64_t base = threadIdx.x; base < units_per_block; base += 256 * U) {
    Unit16<float> v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = load_unit<float, NT>(s + base + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) store_unit<float, NT>(d + base + u * 256, v[u]);
  }
}

// The same bytes in the passes' own access shape: the buffer is a sequence of 8 MiB "transforms" of 1024 rows x 8 KiB; a
// 512-thread workgroup copies one column tile -- sixteen 128-byte row segments per thread at a row stride of 8 KiB, all
// sixteen loads in flight, then the sixteen stores to the same positions of dst (the load and store side of the LAST pass of
// the 1024 x 1024 plan with its arithmetic and LDS exchanges removed); XCD-aware tile order as in xcd_remap mode 0.
// ROWU = 16-byte units per row: 512 (8 KiB rows, the f32 1024 x 1024 plan) or 1024 (16 KiB rows, 16 MiB "transforms": the f64 plan,
// whose pass kernels stream faster than the f32 ones -- VERDICT round 4, item 1a)
template <bool NT, int ROWU>
__global__ void __launch_bounds__(512) copy_tile_kernel(const void* __restrict__ src, void* __restrict__ dst) {
  uint64_t b = blockIdx.x;
  const uint64_t per_xcd = gridDim.x / 8;
  if (per_xcd * 8 == gridDim.x) b = (b % 8) * per_xcd + b / 8;
  constexpr uint64_t TILES = ROWU / 8;  // tiles of 8 units (128 bytes) per row
  const uint64_t transform = b / TILES, tile = b % TILES;
  const uint64_t base = transform * (1024 * (uint64_t)ROWU) + tile * 8 + (uint64_t)(threadIdx.x / 8) * ROWU + threadIdx.x % 8;
  const Unit16<float>* s = (const Unit16<float>*)src + base;
  Unit16<float>* d = (Unit16<float>*)dst + base;
  Unit16<float> v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = load_unit<float, NT>(s + (uint64_t)r * 64 * ROWU);
#pragma unroll
  for (int r = 0; r < 16; ++r) store_unit<float, NT>(d + (uint64_t)r * 64 * ROWU, v[r]);
}

}  // namespace fourier_hip

// Copies `bytes` (a multiple of bytes_per_block, itself a multiple of 32 KiB) from src to dst `reps` times on `stream` and
// reports the mean milliseconds per copy between two HIP events on that stream.  nt bit 0: streaming (non-temporal) loads
// and stores, the cache policy of the pass kernels; bit 1: the column-tile shape of the passes instead of linear slabs; bit 2 (with
// bit 1): rows of 16 KiB (the f64 plan's shape; bytes a multiple of 16 MiB) instead of 8 KiB.  Returns a fourier_hip status code.
// wgs_per_cu > 0 caps the resident workgroups per compute unit by giving every workgroup 160 KiB / wgs_per_cu of (unused)
// dynamic LDS -- the pass kernels hold two workgroups per CU, a bare copy would hold four to eight.
extern "C" int fourier_exp_copy_ceiling(const void* src, void* dst, uint64_t bytes, uint64_t bytes_per_block, int nt, int wgs_per_cu,
                                        int reps, void* stream, float* ms_per_copy) {
  using namespace fourier_hip;
  if (!src || !dst || !ms_per_copy || reps <= 0 || bytes_per_block == 0 || bytes_per_block % (256 * 8 * 16) || bytes % bytes_per_block ||
      bytes / bytes_per_block > 0x7fffffffull)
    return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
#ifdef FOURIER_EMU
  (void)nt; (void)stream;
  *ms_per_copy = 0.0f;
  return ::fourier::c::FOURIER_HIP_UNSUPPORTED;
#else
  try {
    const hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)(bytes / bytes_per_block);
    const bool tile_shape = (nt & 2) != 0;  // bit 1: the passes' column-tile shape (bytes must be a multiple of 8 / 16 MiB)
    const bool wide_rows = tile_shape && (nt & 4) != 0;
    if (tile_shape && bytes % ((uint64_t)(wide_rows ? 16 : 8) << 20)) return ::fourier::c::FOURIER_HIP_INVALID_ARGUMENT;
    struct Events {  // destroyed on every way out (the early return and a throwing HIP_CHECK leaked them)
      hipEvent_t a = nullptr, b = nullptr;
      ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_CHECK(hipEventCreate(&ev.a));
    HIP_CHECK(hipEventCreate(&ev.b));
    const hipEvent_t a = ev.a, b = ev.b;
    const unsigned tile_blocks = (unsigned)(bytes >> 17);  // 128 KiB per workgroup
    const size_t lds = wgs_per_cu > 0 ? ((size_t)160 * 1024 / (size_t)wgs_per_cu) & ~(size_t)1023 : 0;
    if (lds > 48 * 1024) {
      raise_smem_limit((const void*)&copy_tile_kernel<true, 512>, lds);
      raise_smem_limit((const void*)&copy_tile_kernel<false, 512>, lds);
      raise_smem_limit((const void*)&copy_tile_kernel<true, 1024>, lds);
      raise_smem_limit((const void*)&copy_tile_kernel<false, 1024>, lds);
      raise_smem_limit((const void*)&copy_slab_kernel<8, true>, lds);
      raise_smem_limit((const void*)&copy_slab_kernel<8, false>, lds);
    }
    auto launch = [&] {
      if (wide_rows) {
        if (nt & 1) copy_tile_kernel<true, 1024><<<tile_blocks, 512, lds, st>>>(src, dst);
        else copy_tile_kernel<false, 1024><<<tile_blocks, 512, lds, st>>>(src, dst);
      } else if (tile_shape) {
        if (nt & 1) copy_tile_kernel<true, 512><<<tile_blocks, 512, lds, st>>>(src, dst);
        else copy_tile_kernel<false, 512><<<tile_blocks, 512, lds, st>>>(src, dst);
      } else if (nt & 1) copy_slab_kernel<8, true><<<blocks, 256, lds, st>>>(src, dst, bytes_per_block / 16);
      else copy_slab_kernel<8, false><<<blocks, 256, lds, st>>>(src, dst, bytes_per_block / 16);
    };
    launch();  // warm-up (page tables, clocks)
    HIP_CHECK(hipEventRecord(a, st));
    for (int r = 0; r < reps; ++r) launch();
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_per_copy = ms / (float)reps;
    return ::fourier::c::FOURIER_HIP_OK;
  } catch (const EngineError& e) {
    return e.status;
  }
#endif
}
