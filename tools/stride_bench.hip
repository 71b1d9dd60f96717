// stride_bench.hip -- development tool (round 5): what bounds a column-tile copy on MI355X -- the row stride, the rows one wave
// instruction touches, in place or out of place.  The pass kernels' tile shape: a 512-thread workgroup moves 1024 rows x 128 bytes,
// sixteen 16-byte loads per thread in flight, then sixteen stores; XCD-aware tile order (every XCD a contiguous range of
// "transforms").  bench.py's column-tile copy at 16 KiB rows streams 6.1 TB/s where the 8 KiB-row form streams 5.8 (r05_s5).
// build: hipcc --offload-arch=gfx950 -O3 tools/stride_bench.hip -o tools/stride_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MAP: which 8 rows one wave instruction touches (lane / 8 = i, wave = w): 0: rows 8w + i (adjacent rows); 1: rows 8i + w (8 rows
// apart); 2: rows 2 apart; 3: 4 apart.  BAND: tiles per band of the XCD's walk (0 = tile-major inside a transform)
template <int MAP, bool NT>
__global__ void __launch_bounds__(512) tile_copy(const v4u* __restrict__ src, v4u* __restrict__ dst, uint32_t rowu_in, uint32_t rowu_out,
                                                 uint32_t tiles, uint64_t tru_in, uint64_t tru_out, uint32_t band) {
  uint32_t b = blockIdx.x;
  const uint32_t nwg = gridDim.x, per_xcd = nwg / 8;
  const uint32_t xcd = b % 8, slot = b / 8;
  uint32_t transform, tile;
  if (band && tiles % band == 0) {
    const uint32_t tpx = per_xcd / tiles, per_band = tpx * band, bd = slot / per_band, rem = slot % per_band;
    transform = xcd * tpx + rem / band; tile = bd * band + rem % band;
  } else {
    b = xcd * per_xcd + slot; transform = b / tiles; tile = b % tiles;
  }
  const uint32_t t = threadIdx.x, cg = t % 8, w = t / 64, i = (t / 8) % 8;
  uint32_t th;
  if (MAP == 0) th = 8 * w + i;
  else if (MAP == 1) th = 8 * i + w;
  else if (MAP == 2) th = 2 * i + (w & 1) + 16 * (w >> 1);
  else th = 4 * i + (w & 3) + 32 * (w >> 2);
  const v4u* s = src + transform * tru_in + (uint64_t)th * rowu_in + tile * 8 + cg;
  v4u* d = dst + transform * tru_out + (uint64_t)th * rowu_out + tile * 8 + cg;
  v4u v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = NT ? __builtin_nontemporal_load(s + (uint64_t)r * 64 * rowu_in) : s[(uint64_t)r * 64 * rowu_in];
#pragma unroll
  for (int r = 0; r < 16; ++r) { if (NT) __builtin_nontemporal_store(v[r], d + (uint64_t)r * 64 * rowu_out); else d[(uint64_t)r * 64 * rowu_out] = v[r]; }
}

int main(int argc, char** argv) {
  const uint64_t payload = (uint64_t)16 << 30;  // bytes moved each way
  struct Case { uint32_t rowu_in, rowu_out, payload_u; int map; int inplace; uint32_t band; };
  std::vector<Case> cases;
  for (int map = 0; map < 4; ++map)
    for (int inplace = 0; inplace < 2; ++inplace) {
      cases.push_back({512, 512, 512, map, inplace, 0});
      cases.push_back({1024, 1024, 1024, map, inplace, 0});
    }
  for (uint32_t ru : {256u, 264u, 520u, 528u, 544u, 576u, 640u, 768u, 1032u, 2048u, 4096u})
    for (int inplace = 0; inplace < 2; ++inplace) cases.push_back({ru, ru, ru >= 1024 ? (ru / 1024) * 1024 : (ru >= 512 ? 512u : 256u), 0, inplace, 0});
  for (uint32_t band : {8u, 16u}) { cases.push_back({512, 512, 512, 0, 0, band}); cases.push_back({512, 512, 512, 0, 1, band}); cases.push_back({512, 512, 512, 1, 1, band}); }
  // mixed: read side padded, write side natural and the other way round
  cases.push_back({520, 512, 512, 0, 0, 0}); cases.push_back({512, 520, 512, 0, 0, 0}); cases.push_back({1024, 512, 512, 0, 0, 0}); cases.push_back({512, 1024, 512, 0, 0, 0});
  uint64_t maxbytes = 0;
  for (auto& c : cases) {
    const uint64_t tr = payload / ((uint64_t)c.payload_u * 16 * 1024);
    maxbytes = std::max(maxbytes, tr * 1024 * (uint64_t)std::max(c.rowu_in, c.rowu_out) * 16);
  }
  v4u *a, *b;
  CK(hipMalloc(&a, maxbytes)); CK(hipMalloc(&b, maxbytes));
  CK(hipMemset(a, 1, maxbytes)); CK(hipMemset(b, 2, maxbytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  std::vector<std::vector<float>> ms(cases.size());
  for (int round = 0; round < rounds + 1; ++round)
    for (size_t ci = 0; ci < cases.size(); ++ci) {
      const Case& c = cases[ci];
      const uint32_t tiles = c.payload_u / 8;
      const uint64_t tr = payload / ((uint64_t)c.payload_u * 16 * 1024) / 8 * 8;
      const uint32_t grid = (uint32_t)(tr * tiles);
      v4u* dst = c.inplace ? a : b;
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 3; ++rep) {
        const uint64_t ti = (uint64_t)1024 * c.rowu_in, to = (uint64_t)1024 * c.rowu_out;
        switch (c.map) {
          case 0: tile_copy<0, true><<<grid, 512>>>(a, dst, c.rowu_in, c.rowu_out, tiles, ti, to, c.band); break;
          case 1: tile_copy<1, true><<<grid, 512>>>(a, dst, c.rowu_in, c.rowu_out, tiles, ti, to, c.band); break;
          case 2: tile_copy<2, true><<<grid, 512>>>(a, dst, c.rowu_in, c.rowu_out, tiles, ti, to, c.band); break;
          default: tile_copy<3, true><<<grid, 512>>>(a, dst, c.rowu_in, c.rowu_out, tiles, ti, to, c.band); break;
        }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (round) ms[ci].push_back(t / 3);
    }
  for (size_t ci = 0; ci < cases.size(); ++ci) {
    const Case& c = cases[ci];
    std::sort(ms[ci].begin(), ms[ci].end());
    const float med = ms[ci][ms[ci].size() / 2];
    const uint64_t tr = payload / ((uint64_t)c.payload_u * 16 * 1024) / 8 * 8;
    const double bytes = 2.0 * tr * 1024 * c.payload_u * 16;
    printf("{\"row_stride_in_bytes\": %u, \"row_stride_out_bytes\": %u, \"row_payload_bytes\": %u, \"lane_map\": %d, \"in_place\": %d, \"band\": %u, \"ms\": %.3f, \"ms_min\": %.3f, \"gbps\": %.1f}\n",
           c.rowu_in * 16, c.rowu_out * 16, c.payload_u * 16, c.map, c.inplace, c.band, med, ms[ci][0], bytes / (med * 1e-3) / 1e9);
  }
  return 0;
}
