// membench.hip -- development microbenchmarks (not product): achievable HBM / Infinity-Cache
// bandwidth for the access patterns the FFT passes use.  Built by tools/membench.py with hipcc.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// linear copy: each thread moves U units (16 B) per outer iteration, grid-stride; `iters` repeats
// the whole copy inside the launch (persistent form: no launch gaps -> cache-resident bandwidth)
template <int U, int MODE>  // MODE 0 copy, 1 read-only (sum), 2 write-only
__global__ void __launch_bounds__(256) k_lin(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t units, int iters) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  v4u acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (uint64_t base = (uint64_t)blockIdx.x * 256 + threadIdx.x; base < units; base += stride * U) {
      v4u v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * stride;
        if (MODE != 2) v[u] = (i < units) ? src[i] : v4u{0, 0, 0, 0};
        else v[u] = v4u{(unsigned)i, 1, 2, 3};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t i = base + (uint64_t)u * stride;
        if (MODE == 1) acc += v[u];
        else if (i < units) dst[i] = v[u];
      }
    }
  }
  if (MODE == 1 && acc.x == 0x12345678u) dst[0] = acc;
}

// block-contiguous copy: block b owns a contiguous slab; mirrors "one tile per workgroup"
template <int U>
__global__ void __launch_bounds__(256) k_slab(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t units_per_block) {
  const v4u* s = src + (uint64_t)blockIdx.x * units_per_block;
  v4u* d = dst + (uint64_t)blockIdx.x * units_per_block;
  for (uint64_t base = threadIdx.x; base < units_per_block; base += 256 * U) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) d[base + u * 256] = v[u];
  }
}

// column-tile copy: the FFT pass pattern without compute.  Matrix rows x rowunits (16-B units);
// a block of 512 threads moves a tile of `rows` x 8 units: thread (th = tid/8, cg = tid%8) loads 16
// rows th + 64*r.  transposed=1 writes like the FIRST pass (tile column c -> contiguous row c).
__global__ void __launch_bounds__(512) k_tile(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t rowunits,
                                              int transposed) {
  const int tid = threadIdx.x, cg = tid & 7, th = tid >> 3;  // 64 x 8
  const uint64_t tiles = rowunits / 8;
  const uint64_t b = blockIdx.x / tiles, t = blockIdx.x % tiles;
  const uint64_t xform = 1024 * rowunits;
  const v4u* s = src + b * xform + t * 8 + cg;
  v4u v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = s[(uint64_t)(th + 64 * r) * rowunits];
  if (!transposed) {
    v4u* d = dst + b * xform + t * 8 + cg;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[(uint64_t)(th + 64 * r) * rowunits] = v[r];
  } else {
    // tile holds 16 columns (2 per unit); column c of tile t -> output row (t*16 + c), 1024 elements
    // = 512 units; lane-contiguous 8-byte stores in the real kernel, modelled here as one 16-B unit
    // per lane pair: thread writes unit (th + 64*r)/2 ... keep it simple: each thread writes 16 units
    // of row (t*8 + cg) at unit offsets th + 64*r (rows are 1024 units apart here)
    v4u* d = dst + b * xform + (t * 8 + cg) * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) d[th + 64 * r] = v[r];
  }
}

// Model of a fused two-pass FFT kernel's traffic: per tile, stream 128 KiB in from A (HBM), park
// 128 KiB in a small reused ring S, read another 128 KiB back from S, stream 128 KiB out to B (HBM).
// No synchronisation (throughput model only).  ring_tiles*128 KiB = footprint of S.
__global__ void __launch_bounds__(512) k_fused_model(const v4u* __restrict__ A, v4u* __restrict__ B, v4u* __restrict__ S,
                                                     uint64_t tiles, uint64_t ring_tiles, int mode) {
  const int tid = threadIdx.x;
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const v4u* a = A + t * 8192 + tid;
    v4u v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = a[r * 512];
    if (mode >= 1) {
      v4u* s = S + (t % ring_tiles) * 8192 + tid;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r * 512] = v[r];
    }
    if (mode >= 2) {
      const v4u* s2 = S + ((t + ring_tiles / 2 + 7) % ring_tiles) * 8192 + tid;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = s2[r * 512];
    }
    v4u* b = B + t * 8192 + tid;
#pragma unroll
    for (int r = 0; r < 16; ++r) b[r * 512] = v[r];
  }
}
// XCD-aware slab copy: block b is observed to run on XCD b % 8.  With swz=1 the slab index is
// remapped so that every XCD walks its own contiguous 1/8 of the buffer (each 2 MiB page is then
// touched -- and its translation fetched -- by one XCD instead of all eight).
template <int U>
__global__ void __launch_bounds__(256) k_slab_x(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t units_per_block, int swz) {
  uint64_t b = blockIdx.x;
  if (swz) { const uint64_t cpx = gridDim.x / 8; b = (b % 8) * cpx + b / 8; }
  const v4u* s = src + b * units_per_block;
  v4u* d = dst + b * units_per_block;
  for (uint64_t base = threadIdx.x; base < units_per_block; base += 256 * U) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) d[base + u * 256] = v[u];
  }
}
extern "C" int mb_slab_x(const void* src, void* dst, uint64_t bytes, uint64_t bytes_per_block, int swz, void* stream) {
  const unsigned blocks = (unsigned)(bytes / bytes_per_block);
  k_slab_x<8><<<blocks, 256, 0, (hipStream_t)stream>>>((const v4u*)src, (v4u*)dst, bytes_per_block / 16, swz);
  return (int)hipGetLastError();
}
__global__ void __launch_bounds__(512) k_tile_x(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t rowunits, int swz) {
  const int tid = threadIdx.x, cg = tid & 7, th = tid >> 3;
  const uint64_t tiles = rowunits / 8;
  uint64_t blk = blockIdx.x;
  if (swz) { const uint64_t cpx = gridDim.x / 8; blk = (blk % 8) * cpx + blk / 8; }
  const uint64_t b = blk / tiles, t = blk % tiles;
  const uint64_t xform = 1024 * rowunits;
  const v4u* s = src + b * xform + t * 8 + cg;
  v4u v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = s[(uint64_t)(th + 64 * r) * rowunits];
  v4u* d = dst + b * xform + t * 8 + cg;
#pragma unroll
  for (int r = 0; r < 16; ++r) d[(uint64_t)(th + 64 * r) * rowunits] = v[r];
}
extern "C" int mb_tile_x(const void* src, void* dst, uint64_t bytes, int swz, void* stream) {
  const uint64_t rowunits = 512, xform_bytes = 1024 * rowunits * 16;
  const unsigned blocks = (unsigned)(bytes / xform_bytes * (rowunits / 8));
  k_tile_x<<<blocks, 512, 0, (hipStream_t)stream>>>((const v4u*)src, (v4u*)dst, rowunits, swz);
  return (int)hipGetLastError();
}

// ---- fused two-phase model WITH team barriers (what the real fused FFT kernel would do, minus math)
// grid = 8 XCDs x teams_per_xcd x wgs_per_team persistent workgroups (block b -> XCD b % 8, observed).
// A team processes transforms t = team, team + nteams, ...: phase A moves the transform's 64 tiles
// (1024 rows x 128 B, column-tile pattern) from A into the team's 8 MiB scratch (transposed rows),
// team barrier (release/acquire at agent scope), phase B moves 64 column tiles scratch -> B,
// second barrier (WAR on the scratch).  Spins are bounded; on timeout a flag is set and everybody bails.
__device__ __forceinline__ void team_arrive(unsigned* ctr, bool release) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (release) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ bool team_wait(unsigned* ctr, unsigned target, unsigned* abort_flag, bool acquire) {
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > (1u << 22) || ((spins & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    if (acquire) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // the L1 invalidate must have COMPLETED before the other waves pass the barrier below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}
__global__ void __launch_bounds__(512, 4) k_fused_sync(const v4u* __restrict__ A, v4u* __restrict__ B, v4u* __restrict__ S,
                                                       unsigned* ctrs, uint64_t ntransforms, int teams_per_xcd,
                                                       int wgs_per_team, int variant) {
  extern __shared__ unsigned char pad_lds[];  // occupy LDS like the real kernel (limits to 2 WG/CU)
  const int tid = threadIdx.x, cg = tid & 7, th = tid >> 3;
  const unsigned xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const unsigned team_in_xcd = slot / wgs_per_team, w = slot % wgs_per_team;
  const unsigned nteams = 8 * teams_per_xcd, team = xcd * teams_per_xcd + team_in_xcd;
  unsigned* ctr1 = ctrs + team * 64;       // arrivals of phase A (data release)
  unsigned* ctr2 = ctrs + team * 64 + 32;  // arrivals of phase B (scratch free again)
  unsigned* abort_flag = ctrs + 4096;
  v4u* Sg = S + (uint64_t)team * (512 * 1024);  // 8 MiB per team
  const uint64_t rowunits = 512, xform = 1024 * rowunits;
  unsigned phase = 0;
  for (uint64_t t = team; t < ntransforms; t += nteams, ++phase) {
    // ---- phase A: tiles w, w + wgs_per_team, ... of transform t: A -> Sg (transposed rows)
    for (unsigned tile = w; tile < 64; tile += wgs_per_team) {
      const v4u* s = A + t * xform + tile * 8 + cg;
      v4u v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_nontemporal_load(s + (uint64_t)(th + 64 * r) * rowunits);
      if (tile == w && phase > 0) {  // scratch must be free: all phase-B reads of the previous transform done
        if (!team_wait(ctr2, phase * wgs_per_team, abort_flag, false)) return;
      }
      v4u* d = Sg + (uint64_t)(tile * 8 + cg) * 1024;  // column c of the tile -> contiguous row
#pragma unroll
      for (int r = 0; r < 16; ++r) d[th + 64 * r] = v[r];
    }
    team_arrive(ctr1, true);
    if (!team_wait(ctr1, (phase + 1) * wgs_per_team, abort_flag, true)) return;
    // ---- phase B: column tiles of the scratch -> B
    for (unsigned tile = w; tile < 64; tile += wgs_per_team) {
      const v4u* s = Sg + tile * 8 + cg;
      v4u v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = s[(uint64_t)(th + 64 * r) * rowunits];
      if (tile + wgs_per_team >= 64) team_arrive(ctr2, false);  // last tile's loads have landed (vmcnt(0) inside)
      v4u* d = B + t * xform + tile * 8 + cg;
#pragma unroll
      for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], d + (uint64_t)(th + 64 * r) * rowunits);
    }
  }
  (void)variant; (void)pad_lds;
}
// ---- variant 2: software-pipelined (phase A of transform t+1 runs before phase B of t, scratch double
// buffered) and write-through (sc1) scratch stores instead of an L2 write-back fence.
__device__ __forceinline__ void store_sc1(v4u* p, v4u v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ v4u load_flavour(const v4u* p, int flavour) {
  v4u v;
  if (flavour == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (flavour == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else v = *p;
  return v;
}
__global__ void __launch_bounds__(512, 4) k_fused_pipe(const v4u* __restrict__ A, v4u* __restrict__ B, v4u* __restrict__ S,
                                                       unsigned* ctrs, uint64_t ntransforms, int teams_per_xcd,
                                                       int wgs_per_team, int use_sc1) {
  extern __shared__ unsigned char pad_lds[];
  const int tid = threadIdx.x, cg = tid & 7, th = tid >> 3;
  const int load_fl = use_sc1 >> 4;
  use_sc1 &= 15;
  if (tid == 0) ctrs[4200 + blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID[3:0]
  const unsigned xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const unsigned team_in_xcd = slot / wgs_per_team, w = slot % wgs_per_team;
  const unsigned nteams = 8 * teams_per_xcd, team = xcd * teams_per_xcd + team_in_xcd;
  // one counter PER PHASE (a single monotonic sum lets fast workgroups' later arrivals stand in for slow
  // workgroups' earlier ones -- the stale-data bug of the first version)
  const uint64_t mine_max = (ntransforms + nteams - 1) / nteams + 1;
  unsigned* ctr1 = ctrs + 8192 + (uint64_t)team * 2 * mine_max;
  unsigned* ctr2 = ctr1 + mine_max;
  unsigned* abort_flag = ctrs + 4096;
  v4u* S0 = S + (uint64_t)team * (2 * 512 * 1024);  // 2 x 8 MiB per team
  const uint64_t rowunits = 512, xform = 1024 * rowunits;
  const uint64_t mine = (ntransforms > team) ? (ntransforms - team + nteams - 1) / nteams : 0;  // transforms of this team
  // iteration i: phase A of local transform i (if any), then phase B of local transform i-1 (if any)
  for (uint64_t i = 0; i <= mine; ++i) {
    if (i < mine) {
      const uint64_t t = team + i * nteams;
      v4u* Sg = S0 + (i & 1) * (512 * 1024);
      for (unsigned tile = w; tile < 64; tile += wgs_per_team) {
        const v4u* s = A + t * xform + tile * 8 + cg;
        v4u v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_nontemporal_load(s + (uint64_t)(th + 64 * r) * rowunits);
        if (tile == w && i >= 2) {  // buffer (i&1) was last read by phase B of local transform i-2
          if (!team_wait(ctr2 + (i - 2), (unsigned)wgs_per_team, abort_flag, false)) return;
        }
        v4u* d = Sg + (uint64_t)(tile * 8 + cg) * 1024;
        if (use_sc1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) store_sc1(d + th + 64 * r, v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) d[th + 64 * r] = v[r];
        }
      }
      team_arrive(ctr1 + i, !use_sc1);
    }
    if (i >= 1) {
      const uint64_t t = team + (i - 1) * nteams;
      const v4u* Sg = S0 + ((i - 1) & 1) * (512 * 1024);
      if (!team_wait(ctr1 + (i - 1), (unsigned)wgs_per_team, abort_flag, true)) return;
      for (unsigned tile = w; tile < 64; tile += wgs_per_team) {
        const v4u* s = Sg + tile * 8 + cg;
        v4u v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = load_flavour(s + (uint64_t)(th + 64 * r) * rowunits, load_fl);
        if (tile + wgs_per_team >= 64) team_arrive(ctr2 + (i - 1), false);
        v4u* d = B + t * xform + tile * 8 + cg;
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], d + (uint64_t)(th + 64 * r) * rowunits);
      }
    }
  }
  (void)pad_lds;
}
extern "C" int mb_fused_pipe(const void* A, void* B, void* S, void* ctrs, uint64_t bytes, int teams_per_xcd, int wgs_per_team,
                             int lds_bytes, int use_sc1, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(ctrs, 0, (8192 + 64 * 2 * 2048) * 4, st);
  hipFuncSetAttribute((const void*)k_fused_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const unsigned blocks = 8 * teams_per_xcd * wgs_per_team;
  k_fused_pipe<<<blocks, 512, lds_bytes, st>>>((const v4u*)A, (v4u*)B, (v4u*)S, (unsigned*)ctrs, bytes / (8 << 20), teams_per_xcd, wgs_per_team, use_sc1);
  return (int)hipGetLastError();
}

extern "C" int mb_fused_sync(const void* A, void* B, void* S, void* ctrs, uint64_t bytes, int teams_per_xcd, int wgs_per_team,
                             int lds_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(ctrs, 0, 4097 * 4, st);
  hipFuncSetAttribute((const void*)k_fused_sync, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const unsigned blocks = 8 * teams_per_xcd * wgs_per_team;
  k_fused_sync<<<blocks, 512, lds_bytes, st>>>((const v4u*)A, (v4u*)B, (v4u*)S, (unsigned*)ctrs, bytes / (8 << 20), teams_per_xcd, wgs_per_team, 0);
  return (int)hipGetLastError();
}

// column-tile copy with independent source / destination row strides (in 16-B units): does an 8 KiB row
// stride (N=2^20: 1024 complex64 per row) cost DRAM bank/channel conflicts that a padded stride avoids?
__global__ void __launch_bounds__(512) k_tile_s(const v4u* __restrict__ src, v4u* __restrict__ dst, uint64_t src_ru, uint64_t dst_ru,
                                                int swz) {
  const int tid = threadIdx.x, cg = tid & 7, th = tid >> 3;
  uint64_t blk = blockIdx.x;
  if (swz) { const uint64_t cpx = gridDim.x / 8; blk = (blk % 8) * cpx + blk / 8; }
  const uint64_t b = blk / 64, t = blk % 64;   // 64 tiles of 8 units per transform, 1024 rows
  const v4u* s = src + b * 1024 * src_ru + t * 8 + cg;
  v4u v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = s[(uint64_t)(th + 64 * r) * src_ru];
  v4u* d = dst + b * 1024 * dst_ru + t * 8 + cg;
#pragma unroll
  for (int r = 0; r < 16; ++r) d[(uint64_t)(th + 64 * r) * dst_ru] = v[r];
}
extern "C" int mb_tile_s(const void* src, void* dst, uint64_t transforms, uint64_t src_ru, uint64_t dst_ru, int swz, void* stream) {
  k_tile_s<<<(unsigned)(transforms * 64), 512, 0, (hipStream_t)stream>>>((const v4u*)src, (v4u*)dst, src_ru, dst_ru, swz);
  return (int)hipGetLastError();
}

extern "C" int mb_fused_model(const void* A, void* B, void* S, uint64_t bytes, uint64_t ring_bytes, int mode, int blocks, void* stream) {
  k_fused_model<<<blocks, 512, 0, (hipStream_t)stream>>>((const v4u*)A, (v4u*)B, (v4u*)S, bytes / (128 << 10), ring_bytes / (128 << 10), mode);
  return (int)hipGetLastError();
}

extern "C" int mb_lin(int mode, int U, const void* src, void* dst, uint64_t bytes, int blocks, int iters, void* stream) {
  const uint64_t units = bytes / 16;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(UU, MM) k_lin<UU, MM><<<blocks, 256, 0, st>>>((const v4u*)src, (v4u*)dst, units, iters)
#define BYMODE(UU) do { if (mode == 0) LAUNCH(UU, 0); else if (mode == 1) LAUNCH(UU, 1); else LAUNCH(UU, 2); } while (0)
  if (U == 1) BYMODE(1); else if (U == 2) BYMODE(2); else if (U == 4) BYMODE(4); else if (U == 8) BYMODE(8); else BYMODE(16);
  return (int)hipGetLastError();
}
extern "C" int mb_slab(int U, const void* src, void* dst, uint64_t bytes, uint64_t bytes_per_block, void* stream) {
  const uint64_t upb = bytes_per_block / 16;
  const unsigned blocks = (unsigned)(bytes / bytes_per_block);
  hipStream_t st = (hipStream_t)stream;
  if (U == 4) k_slab<4><<<blocks, 256, 0, st>>>((const v4u*)src, (v4u*)dst, upb);
  else if (U == 8) k_slab<8><<<blocks, 256, 0, st>>>((const v4u*)src, (v4u*)dst, upb);
  else k_slab<16><<<blocks, 256, 0, st>>>((const v4u*)src, (v4u*)dst, upb);
  return (int)hipGetLastError();
}
extern "C" int mb_tile(const void* src, void* dst, uint64_t bytes, int transposed, void* stream) {
  const uint64_t rowunits = 512;  // 1024 complex64 per row = 8 KiB
  const uint64_t xform_bytes = 1024 * rowunits * 16;
  const unsigned blocks = (unsigned)(bytes / xform_bytes * (rowunits / 8));
  k_tile<<<blocks, 512, 0, (hipStream_t)stream>>>((const v4u*)src, (v4u*)dst, rowunits, transposed);
  return (int)hipGetLastError();
}

// ---- XCD-local exchange model: is an intermediate that round-trips through the XCD's OWN L2 free? ----
// Persistent workgroups (512 threads, LDS-padded to 2 per CU like the pass kernels).  Per 128 KiB tile: stream the
// tile in from A (nt), park it in a slot of THIS XCD's private ring (fp_bytes per XCD, so the eight rings together
// are 8*fp), read back another slot of the SAME ring that a neighbouring workgroup of this XCD wrote a moment ago,
// stream the tile out to B (nt).  No synchronisation (throughput model; the values read back are whatever is
// there).  xshift != 0 reads the ring of XCD (x + xshift) % 8 instead: the cross-XCD control.
// store_fl: 0 plain, 1 sc1 (write-through), 2 nt;  load_fl: 0 plain, 1 sc1 (L1 bypass), 2 nt
__device__ __forceinline__ void st_fl(v4u* p, v4u v, int fl) {
  if (fl == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (fl == 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}
__global__ void __launch_bounds__(512, 4) k_l2x(const v4u* __restrict__ A, v4u* __restrict__ B, v4u* __restrict__ S, unsigned* xcc_out,
                                                uint64_t tiles, uint64_t fp_units, int mode, int store_fl, int load_fl, int xshift) {
  extern __shared__ unsigned char pad_lds[];
  const int tid = threadIdx.x;
  const unsigned xcd = blockIdx.x % 8, slot = blockIdx.x / 8, wpx = gridDim.x / 8;
  if (tid == 0) xcc_out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID[3:0]
  const uint64_t nslots = fp_units / 8192;  // 128 KiB slots in one XCD's ring
  v4u* ring_w = S + (uint64_t)xcd * fp_units;
  const v4u* ring_r = S + (uint64_t)((xcd + xshift) % 8) * fp_units;
  uint64_t it = 0;
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x, ++it) {
    const v4u* a = A + t * 8192 + tid;
    v4u v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __builtin_nontemporal_load(a + r * 512);
    const uint64_t sw = (it * wpx + slot) % nslots;
    if (mode >= 1) {
      v4u* s = ring_w + sw * 8192 + tid;
#pragma unroll
      for (int r = 0; r < 16; ++r) st_fl(s + r * 512, v[r], store_fl);
    }
    if (mode >= 2) {
      // transposed read-back (the real kernel reads columns of what the team wrote): slot written ~nslots/2 tiles ago
      const v4u* s2 = ring_r + ((sw + nslots / 2 + 1) % nslots) * 8192;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const v4u* p = s2 + ((tid & 7) + 8 * (((tid >> 3) + 64 * r) % 1024));
        if (load_fl == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[r]) : "v"(p) : "memory");
        else if (load_fl == 2) v[r] = __builtin_nontemporal_load(p);
        else v[r] = *p;
      }
      if (load_fl == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    v4u* b = B + t * 8192 + tid;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], b + r * 512);
  }
  (void)pad_lds;
}
extern "C" int mb_l2x(const void* A, void* B, void* S, void* xcc, uint64_t bytes, uint64_t fp_bytes_per_xcd, int mode, int store_fl,
                      int load_fl, int xshift, int wgs_per_xcd, int lds_bytes, void* stream) {
  hipFuncSetAttribute((const void*)k_l2x, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  k_l2x<<<8 * wgs_per_xcd, 512, lds_bytes, (hipStream_t)stream>>>((const v4u*)A, (v4u*)B, (v4u*)S, (unsigned*)xcc, bytes / (128 << 10),
                                                                 fp_bytes_per_xcd / 16, mode, store_fl, load_fl, xshift);
  return (int)hipGetLastError();
}

// ---- register-resident transform model (the only on-chip home big enough for a 2^20-point f32 transform: an XCD's
// registers).  64 persistent workgroups per XCD (2 per CU, 512 threads, a 128 KiB tile each = one 8 MiB transform per
// XCD).  Per transform: every workgroup streams its tile in from A, the team does an all-to-all in `rounds` rounds
// through a double-buffered L2 window (per round: 128 KiB / rounds per workgroup written with plain stores, ONE
// XCD-wide barrier, the same amount read back from another workgroup's region with sc1 loads into the same
// registers), then streams the tile out to B.  No arithmetic, no LDS exchange: an upper bound for a fused 2^20 FFT
// that keeps the transform in registers.  Teams form by hardware XCC id (ticket per XCD); an XCD that does not get
// exactly 64 workgroups, or a barrier that does not complete, raises ctrl[1023] and everybody bails (bounded spins).
__global__ void __launch_bounds__(512, 4) k_regx(const v4u* __restrict__ A, v4u* __restrict__ B, v4u* __restrict__ S, unsigned* ctrl,
                                                 uint64_t ntransforms, int rounds) {
  extern __shared__ unsigned char pad_lds[];
  __shared__ unsigned bc[2];
  const int tid = threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;
  unsigned* abort_flag = ctrl + 1023;
  unsigned* ticket = ctrl + xcc * 32;
  unsigned* ctr = ctrl + xcc * 32 + 16;
  if (tid == 0) bc[0] = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned w = bc[0];
  if (w >= 64) { if (tid == 0) __hip_atomic_store(abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
  const int upr = 16 / rounds;                              // 16-byte units per thread per round
  v4u* W = S + (uint64_t)xcc * (2 * 64 * 8192 / rounds);    // two buffers of 64 regions of (8192 / rounds) units
  unsigned phase = 0;
  for (uint64_t t = xcc; t < ntransforms; t += 8) {
    const v4u* a = A + t * (64 * 8192) + (uint64_t)w * 8192 + tid;
    v4u v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __builtin_nontemporal_load(a + r * 512);
    for (int r = 0; r < rounds; ++r) {
      v4u* wr = W + (uint64_t)(phase & 1) * (64 * 8192 / rounds) + (uint64_t)w * (8192 / rounds) + tid;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u / upr == r) wr[(u % upr) * 512] = v[u];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = 64u * (phase + 1);
        unsigned spins = 0, ok = 1;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 21) || ((spins & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
        }
        bc[1] = ok;
      }
      __syncthreads();
      if (!bc[1]) return;
      const unsigned src = (w + 8u * (unsigned)r + 5u) & 63u;  // another workgroup's region of this round
      const v4u* rd = W + (uint64_t)(phase & 1) * (64 * 8192 / rounds) + (uint64_t)src * (8192 / rounds) + tid;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u / upr == r) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[u]) : "v"(rd + (u % upr) * 512) : "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ++phase;
    }
    v4u* b = B + t * (64 * 8192) + (uint64_t)w * 8192 + tid;
#pragma unroll
    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(v[r], b + r * 512);
  }
  (void)pad_lds;
}
extern "C" int mb_regx(const void* A, void* B, void* S, void* ctrl, uint64_t bytes, int rounds, int lds_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(ctrl, 0, 1024 * 4, st);
  hipFuncSetAttribute((const void*)k_regx, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  k_regx<<<512, 512, lds_bytes, st>>>((const v4u*)A, (v4u*)B, (v4u*)S, (unsigned*)ctrl, bytes / (8 << 20), rounds);
  return (int)hipGetLastError();
}
