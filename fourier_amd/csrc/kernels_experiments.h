// kernels_experiments.h -- the measured-slower designs: the half-tile last pass and the XCD-fused
// one-launch plan.  Compiled only into lib/libfourier_experiments.so and the CPU emulation build.
#pragma once
#include "kernels_pass.h"

namespace fourier_hip {

// last pass of length 2L on half tiles (pass_tile, SPLIT = 1): grid = 2 x batch x tiles
#ifndef FOURIER_SPLIT_LD
#define FOURIER_SPLIT_LD POL_PLAIN  // the second reader of a line must find it in the L2: no streaming hint on the loads
#endif
template <typename T, int L, int CG, int IO = IO_PLAIN>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_last_split_kernel(PassArgs a) {
  FOURIER_DYN_SMEM(smem);
  pass_tile<T, L, CG, MODE_LAST, IO, FOURIER_SPLIT_LD, PassPolicy<L, MODE_LAST>::ST, 1>(a, blockIdx.x, gridDim.x, smem, (int)threadIdx.x);
}

// ---- N = L1 x L2 with BOTH passes in one launch and the intermediate parked in the XCD's own L2 --------------------
// (2^16 .. 2^18 in f32, 2^15 .. 2^17 in f64: N * sizeof(complex) <= 2 MiB.)  The two-launch plan moves every point
// through HBM twice; here a transform is read from HBM once (pass A = the FIRST pass) and written once (pass B = the
// LAST pass), and the transposed intermediate between them lives in a small window that is written and read back by
// workgroups of ONE XCD, so it never leaves that XCD's 4 MiB L2 (measured with tools/membench.py --l2x: a window of
// <= 1 MiB per XCD that is written with plain stores and read back with sc1 loads costs nothing next to the HBM
// streams: 5.84 vs 5.85 TB/s; profiles/r02_membench.jsonl).
//
// Persistent workgroups, data-flow scheduling, no team barrier.  Every workgroup reads the id of the XCD it runs
// on (HW_REG_XCC_ID) and pulls work items from THAT XCD's queue, so all items of one transform are executed on one
// XCD whatever the dispatcher did (placement is observed, never assumed).  The queue of XCD x is the sequence, for
// step s = 0, 1, ...: the tiles of pass B of its local transform s - 1, then the tiles of pass A of local transform s
// (older work first: with depth = 1 pass A of s reuses the window pass B of s - 1 is reading).  The workgroup that draws (A, s, tile 0) claims the next global transform from one device-wide counter and
// publishes it (map[s]); XCDs therefore share the batch dynamically and any number of resident workgroups per XCD
// (even one) completes the job.  An item waits only for items drawn EARLIER from the same queue (its transform's
// claim; pass B: all tiles of pass A; pass A, just before its stores: the readers of the window slot's previous
// tenant, `depth` steps back), every drawn item is held by a running workgroup, hence no deadlock.  Waits are nevertheless bounded
// (spin_limit) and raise ctrl[1] instead of hanging the device.
// Visibility: producer = plain stores, every wave waits vmcnt(0) (the stores have reached the XCD's L2), workgroup
// barrier, then one relaxed agent-scope increment; consumer = one lane polls the counter (relaxed, sc1), workgroup
// barrier, then sc1 loads, which are served by that same L2.
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lane 0 only: wait until *p >= target (or *p != 0 when target == 0); false = gave up (abort flag raised)
__device__ __forceinline__ bool fused_wait(const uint32_t* p, uint32_t target, uint32_t* abort_flag, uint32_t limit, uint32_t* seen) {
  for (uint32_t spins = 0;; ++spins) {
    const uint32_t v = ld_relaxed(p);
    if (target ? v >= target : v != 0) { *seen = v; return true; }
    if (spins >= limit || ((spins & 63) == 63 && ld_relaxed(abort_flag))) {
      __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

#ifndef FOURIER_FUSED_MIN_WAVES
// three 256-thread workgroups per CU (<= 168 VGPRs): at four (<= 128) the two pass bodies spill 56-116 bytes per lane
// and every size measured slower (profiles/r02_s3_plan4096_conv_and_fused_ab.jsonl)
#define FOURIER_FUSED_MIN_WAVES 3
#endif
struct FusedWindowFree {
  const uint32_t* counter;  // done_b of the slot's previous tenant, or null when the slot has never been used
  uint32_t target;
  uint32_t* abort_flag;
  uint32_t limit;
  int tid;
  __device__ __forceinline__ void operator()() const {
    if (!counter) return;  // wave-uniform
    if (tid == 0) {
      uint32_t seen;
      (void)fused_wait(counter, target, abort_flag, limit, &seen);  // on give-up the abort flag is up: every later wait bails out
    }
    __syncthreads();
  }
};

template <typename T, int L1, int CG1, int L2, int CG2>
__global__ void __launch_bounds__((L1 / 16) * CG1, FOURIER_FUSED_MIN_WAVES) fft_l2fused_kernel(FusedArgs f) {
  using CA = TileCfg<T, L1, CG1>;
  using CB = TileCfg<T, L2, CG2>;
  static_assert(CA::NT == CB::NT, "both passes run on the same workgroup");
  constexpr size_t SMEM_A = CA::smem_bytes(MODE_FIRST), SMEM_B = CB::smem_bytes(MODE_LAST);
  constexpr size_t SLOT = ((SMEM_A > SMEM_B ? SMEM_A : SMEM_B) + 15) & ~(size_t)15;  // broadcast words behind the tiles' LDS
  FOURIER_DYN_SMEM(smem);
  volatile uint32_t* bc = (volatile uint32_t*)(smem + SLOT);
  const int tid = (int)threadIdx.x;
  const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & (FUSED_XCC_IDS - 1);  // HW_REG_XCC_ID[3:0]
  uint32_t* const q = f.ctrl + FUSED_CTRL_HDR + (uint64_t)xcc * fused_ctrl_stride(f.batch);
  const uint64_t cap = (uint64_t)f.batch + 2;
  uint32_t* const map = q + 16;
  uint32_t* const done_a = map + cap;
  uint32_t* const done_b = done_a + cap;
  uint32_t* const abort_flag = f.ctrl + 1;
  const uint32_t per_step = f.tiles_a + f.tiles_b;
  cpx<T>* const win0 = (cpx<T>*)f.window + (uint64_t)xcc * f.depth * f.a.n;

  for (;;) {
    // ---- draw an item; lane 0 resolves its transform and waits for what the item depends on
    if (tid == 0) {
      const uint32_t item = __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t s = item / per_step, r = item % per_step;
      const bool is_a = r >= f.tiles_b;  // within a step: pass B of the previous transform first, then pass A of this one
      const uint32_t tile = is_a ? r - f.tiles_b : r;
      uint32_t g = 0xffffffffu, j = is_a ? s : s - 1;
      int act = 0;  // 0 skip, 1 run, 2 exit
      if (!is_a && s == 0) {
        act = 0;  // there is no transform -1
      } else if (j >= cap) {
        act = 2;
      } else {
        uint32_t v = 0;
        bool ok = true;
        if (is_a && tile == 0) {
          const uint32_t t = __hip_atomic_fetch_add(f.ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = t < f.batch ? t + 1 : 0xffffffffu;
          __hip_atomic_store(map + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          ok = fused_wait(map + j, 0, abort_flag, f.spin_limit, &v);
        }
        if (!ok) act = 2;
        else if (v == 0xffffffffu) act = is_a ? 0 : 2;  // the batch is exhausted: nothing after this pass-B item exists
        else {
          uint32_t seen;
          if (!is_a) ok = fused_wait(done_a + j, f.tiles_a, abort_flag, f.spin_limit, &seen);  // pass A waits later, see WindowFree
          act = ok ? 1 : 2;
          g = v - 1;
        }
      }
      bc[0] = (uint32_t)act; bc[1] = g; bc[2] = j; bc[3] = (is_a ? 0u : 0x80000000u) | tile;
    }
    __syncthreads();
    const uint32_t act = bc[0], g = bc[1], j = bc[2], kt = bc[3];
    __syncthreads();  // everyone has read the slot before lane 0 of the next iteration rewrites it
    if (act == 2) return;
    if (act == 0) continue;
    const bool is_a = (kt >> 31) == 0;
    const uint32_t tile = kt & 0x7fffffffu;
    cpx<T>* const win = win0 + (uint64_t)(j % f.depth) * f.a.n;
    int tid_i = tid;
    FOURIER_LAUNDER(tid_i);
    if (is_a) {
      PassArgs a = f.a;
      a.in = (const cpx<T>*)f.in + (uint64_t)g * f.a.n;
      a.out = win;
      // the window slot's previous tenant (local transform j - depth) must have been read completely -- checked only
      // now, with this tile's data already loaded and transformed in registers
      const FusedWindowFree hook{j >= f.depth ? done_b + (j - f.depth) : nullptr, f.tiles_b, abort_flag, f.spin_limit, tid};
      pass_tile<T, L1, CG1, MODE_FIRST, IO_PLAIN, POL_NT, POL_PLAIN, 0, FusedWindowFree>(a, tile, f.tiles_a, smem, tid_i, hook);
      FOURIER_WAIT_VMEM();  // this wave's window stores have reached the L2
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(done_a + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      PassArgs b = f.b;
      b.in = win;
      b.out = (cpx<T>*)f.out + (uint64_t)g * f.a.n;
      pass_tile<T, L2, CG2, MODE_LAST, IO_PLAIN, POL_SC1, POL_NT>(b, tile, f.tiles_b, smem, tid_i);
      __syncthreads();  // every wave holds its window data in registers by now (the tile's LDS exchanges waited for it)
      if (tid == 0) __hip_atomic_fetch_add(done_b + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- LAST pass of a length whose tile fills a CU (L = 2048: a 256 KiB tile, ONE 1024-thread workgroup per CU), persistent,
// with the next tile on its way while the current one is finished ----
// fft_pass_kernel at one workgroup per CU runs load -> butterflies / exchanges -> store strictly one after the other: nothing
// overlaps a tile's arithmetic (about a third of its time), and a CU's load and store streams never overlap each other
// (C5's last pass: 63 % of the HBM peak against 74 % for the two-workgroups-per-CU kernels of length 1024).  Here a workgroup
// walks a range of tiles, and per tile:
//   [B] barrier: every wave is done with the exchange buffer (the reads of the last exchange)
//       rows 0..7 of the NEXT tile: LDS-DMA (buffer_load_dwordx4 ... lds) into the now idle exchange buffer -- no registers;
//       rows 8..15 of the next tile: ordinary loads into 32 free registers
//       stores of THIS tile's sixteen rows (issued behind the loads: vmcnt completes in order on gfx9, so the loads can
//       be waited for with the stores still in flight -- the order that lost in round 3 was stores first, loads behind them)
//       s_waitcnt vmcnt(#stores): the next tile has landed; ds_read its rows 0..7 back (lane-linear 1 KiB per wave and row)
//   [A] barrier: every wave has its rows out of the buffer before the first exchange overwrites it
//       in-tile FFT (tile_core), same arithmetic as fft_pass_kernel: bit-identical results.
// MEASURED SLOWER (round 4, profiles/r04_s2..s4_prefetch_last_pass*_ab.jsonl, shared buffers): C5's last pass 13.6 ms per 1024
// transforms as fft_pass_kernel, 17.4-18.0 ms in this form; with the stores issued AHEAD of the prefetch (persistence alone,
// nothing overlapped) 15.7-16.0 ms; four instead of eight LDS-DMA rows, plain instead of streaming stores, waiting for the
// stores as well: all within 3 % of 17.6 ms.  The length-1024 passes (two workgroups per CU) lose more: 11.5 -> 17.2 ms.  A CU
// that holds ONE 256 KiB tile has no room to keep a second tile's loads in flight during the butterflies (the exchange buffer
// is busy until the last exchange, the registers until the stores), so the next tile's latency stays exposed, and a fixed
// tile range per persistent workgroup gives up the dispatcher's dynamic balancing.  Kept in the experiments library only.
// Results agree with fft_pass_kernel to rounding (rel-L2 9e-8: the compiler contracts the same source into different FMAs in
// the two kernels), not bit for bit.
// The loop is rotated (prefetch and previous tile's stores first, the first iteration's stores go to a zero-sized descriptor
// and are dropped) so that the compiler sees the same outstanding-operation state on loop entry and on the back edge.
// IO_BLU_OUT (the chirp-out pass of a Bluestein plan): the chirp units are loaded ahead of the prefetch, so that waiting
// for them does not wait for the prefetch.
// The gfx950 builtins below live in __device__ functions, not in the kernel body: the HOST pass of hipcc analyses a
// __global__ function's body too, finds no such builtin for x86, and then silently emits no launch stub for the kernel.
#ifdef FOURIER_EMU
#define FOURIER_LDS_PTR(p) (p)
#else
#define FOURIER_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif
// LDS-DMA: sixteen bytes per lane from a buffer descriptor straight into LDS at lds + 16 * lane (lds wave-uniform), no
// registers; completion is counted on vmcnt like any load, and NOTHING ELSE orders a later ds_read behind it
template <int AUX> __device__ __forceinline__ void lds_dma_unit(BufRsrc r, unsigned char* lds, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, FOURIER_LDS_PTR(lds), 16, (int)voff, 0, 0, AUX);
}
// s_waitcnt vmcnt(N), N < 64, through the builtin (the compiler's own counter model sees it; inline asm it would not):
// simm16 = vmcnt[3:0] | expcnt (7 = no wait) << 4 | lgkmcnt (15 = no wait) << 8 | vmcnt[5:4] << 14
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#ifndef FOURIER_PF_DMA_ROWS
#define FOURIER_PF_DMA_ROWS 8
#endif
// A/B knobs of the prefetching last pass (tools/build_variants.py)
#ifndef FOURIER_PF_SIMPLE_LOOP
#define FOURIER_PF_SIMPLE_LOOP 0  // 1: persistent workgroups over the plain pass_tile, no prefetch (A/B)
#endif
#ifndef FOURIER_PF_WAIT_ALL
#define FOURIER_PF_WAIT_ALL 0     // 1: s_waitcnt vmcnt(0) -- also the previous tile's stores -- before the prefetched tile is used
#endif
#ifndef FOURIER_PF_ST_PLAIN
#define FOURIER_PF_ST_PLAIN 0     // 1: final stores without the streaming hint
#endif
#ifndef FOURIER_PF_BARRIER_AFTER_WAIT
#define FOURIER_PF_BARRIER_AFTER_WAIT 1
#endif
#ifndef FOURIER_PF_STORES_FIRST
#define FOURIER_PF_STORES_FIRST 0 // 1: the previous tile's stores are issued AHEAD of the prefetch (persistence alone, no overlap)
#endif
// LDS of the prefetching last pass: the exchange buffer (which doubles as the landing zone of the LDS-DMA rows) and, where a
// CU's 160 KiB hold them for every resident workgroup, the stage twiddle tables behind it
template <typename T, int L, int CG> struct PrefetchCfg {
  using C = TileCfg<T, L, CG>;
  static constexpr size_t TW_OFF = (C::EXCH_BYTES + 15) & ~(size_t)15;
  static constexpr size_t TW_BYTES = (size_t)(C::Q + (C::R3 > 1 ? C::R3 : 0)) * 16 * sizeof(cpx<T>);
  static constexpr int WGS_PER_CU = C::NT >= 1024 ? 1 : 2;
  static constexpr bool TW_IN_LDS = (TW_OFF + TW_BYTES) * WGS_PER_CU <= (size_t)160 * 1024;
  static constexpr size_t SMEM = TW_IN_LDS ? TW_OFF + TW_BYTES : C::EXCH_BYTES;
};
template <typename T, int L, int CG, int IO>
__global__ void __launch_bounds__((L / 16) * CG, FOURIER_MIN_WAVES((L / 16) * CG)) fft_last_prefetch_kernel(PassArgs a) {
  static_assert(IO == IO_PLAIN || IO == IO_BLU_OUT, "last pass: plain or chirp-out");
  using C = TileCfg<T, L, CG>;
  constexpr int VEC = C::VEC, Q = C::Q, COLS = C::COLS, NT = C::NT, WAVES = NT / 64;
  constexpr int NDMA = FOURIER_PF_DMA_ROWS, NREG = 16 - NDMA;  // rows by LDS-DMA / rows straight into registers
  static_assert(NT % 64 == 0 && (size_t)NDMA * NT * 16 <= C::EXCH_BYTES, "the DMA rows must fit the exchange buffer");
  constexpr int LDAUX = PassPolicy<L, MODE_LAST, CG>::LD == POL_NT ? BUF_NT : BUF_PLAIN;
  constexpr int STAUX = (PassPolicy<L, MODE_LAST, CG>::ST == POL_NT && !FOURIER_PF_ST_PLAIN) ? BUF_NT : BUF_PLAIN;
  constexpr int NSTORES = (FOURIER_PF_WAIT_ALL || FOURIER_PF_STORES_FIRST) ? 0 : (IO == IO_BLU_OUT ? 8 : 16);  // VMEM operations issued behind the prefetch in one iteration
  using PF = PrefetchCfg<T, L, CG>;
  FOURIER_DYN_SMEM(smem);
  const int tid = (int)threadIdx.x;
  // The stage twiddles of the in-tile FFT, staged in LDS once per workgroup (where they fit beside the exchange buffer): read
  // from global memory inside tile_core their s_waitcnt vmcnt would also wait -- vmcnt completes in order -- for the stores of
  // the previous tile, which are meant to drain under this tile's butterflies.
  const cpx<T>* tw1 = (const cpx<T>*)a.tw1;
  const cpx<T>* tw2 = (const cpx<T>*)a.tw2;
  if constexpr (PF::TW_IN_LDS) {
    cpx<T>* l1 = (cpx<T>*)(smem + PF::TW_OFF);
    cpx<T>* l2 = l1 + Q * 16;
    for (int i = tid; i < Q * 16; i += NT) l1[i] = tw1[i];
    if constexpr (C::R3 > 1)
      for (int i = tid; i < C::R3 * 16; i += NT) l2[i] = tw2[i];
    tw1 = l1; tw2 = l2;  // visible after the loop's first barrier
  }
#if FOURIER_PF_SIMPLE_LOOP
  // the plain last pass, tile after tile in persistent workgroups, NO prefetch: what a workgroup that is not torn down and
  // re-dispatched between tiles is worth by itself.  Measured (profiles/r04_s33_*, r04_s34_*): 30 % SLOWER than one workgroup per
  // tile -- half of that is the static tile assignment (tiles drawn from per-XCD atomic counters instead: 12 % slower), the rest
  // the in-order vmcnt of gfx9: the first use of tile t + 1's loads also waits for the acknowledgement of tile t's stores
  for (uint32_t vb = blockIdx.x; vb < (uint32_t)a.total_cols; vb += gridDim.x) {
    pass_tile<T, L, CG, MODE_LAST, IO, PassPolicy<L, MODE_LAST, CG>::LD, PassPolicy<L, MODE_LAST, CG>::ST>(a, vb, (uint32_t)a.total_cols, smem, tid);
    __syncthreads();
  }
  return;
#endif
  const cpx<T>* __restrict__ in = (const cpx<T>*)a.in;
  cpx<T>* __restrict__ out = (cpx<T>*)a.out;
  const uint32_t total = (uint32_t)a.total_cols, tiles = (uint32_t)a.tiles;  // total_cols: tiles of the whole launch
  const T scale = (T)a.scale;
  const uint32_t blu_bytes = (uint32_t)(a.blu_n * sizeof(cpx<T>));

  cpx<T> x[VEC][16];
#pragma unroll
  for (int v = 0; v < VEC; ++v)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[v][r] = cpx<T>{0, 0};
  uint64_t pb = 0, pc0 = 0;  // the tile whose results x holds
  uint32_t st_bytes = 0;     // 0: nothing to store yet (first iteration): the descriptor drops every store

  // Everything a phase derives from the thread index is derived from a laundered copy taken IN that phase (see tile_core): the
  // compiler otherwise computes the lane offsets of every phase once, ahead of the loop, and carries them through the in-tile
  // FFT -- 160 bytes of scratch per lane in the first version of this kernel.
  // stores of the tile (pb, pc0) held in x; IO_BLU_OUT: times the chirp units c (loaded ahead of the prefetch)
  auto store_tile = [&](int th, int cg, const Unit16<T>* c, uint32_t bytes) {
    if constexpr (IO == IO_BLU_OUT) {
      // out = work (.) x (.) scale, first blu_n points only (bluesteins.rs:240-258); rows 8..15 lie beyond the user array
      const BufRsrc ro = make_rsrc(out + pb * a.blu_n, bytes ? blu_bytes : 0u);
      const uint32_t voff = (uint32_t)((pc0 + (uint64_t)(cg * VEC) + a.s * (uint64_t)th) * sizeof(cpx<T>));
      const uint32_t rowb = (uint32_t)(a.s * (uint64_t)Q * sizeof(cpx<T>));
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        Unit16<T> u;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cpx<T> y = x[v][r];
          if (a.swap_out) y = {y.im, y.re};
          y = cmul(y, cpx<T>{c[r].a[2 * v], c[r].a[2 * v + 1]});
          if (a.blu_swap) y = {y.im, y.re};
          u.a[2 * v] = y.re * scale; u.a[2 * v + 1] = y.im * scale;
        }
        buf_store_unit<T, BUF_PLAIN>(ro, voff + (uint32_t)r * rowb, u);
      }
    } else {
      // output row of register r: j0 + s * (L*i + th + Q*r); the uniform part goes into the descriptor base
      const uint64_t i = pc0 >> a.s_shift, j0 = pc0 & (a.s - 1);
      const uint64_t base = pb * a.n + j0 + a.s * ((uint64_t)L * i);
      const uint32_t voff = (uint32_t)(((uint64_t)(cg * VEC) + a.s * (uint64_t)th) * sizeof(cpx<T>));
      const uint64_t rows = a.s * (uint64_t)Q;  // elements between a thread's consecutive output rows
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        Unit16<T> u;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          cpx<T> y = x[v][r];
          if (a.swap_out) y = {y.im, y.re};
          u.a[2 * v] = y.re * scale; u.a[2 * v + 1] = y.im * scale;
        }
        buf_store_unit<T, STAUX>(make_rsrc(out + base + rows * (uint64_t)r, bytes), voff, u);
      }
    }
  };
  auto chirp_load = [&](int th, int cg, Unit16<T>* c) {
    if constexpr (IO == IO_BLU_OUT) {
      const BufRsrc rc = make_rsrc(a.blu_x, blu_bytes);
      const uint32_t voff = (uint32_t)((pc0 + (uint64_t)(cg * VEC) + a.s * (uint64_t)th) * sizeof(cpx<T>));
      const uint32_t rowb = (uint32_t)(a.s * (uint64_t)Q * sizeof(cpx<T>));
#pragma unroll
      for (int r = 0; r < 8; ++r) c[r] = buf_load_unit<T>(rc, voff + (uint32_t)r * rowb);
    }
  };

  for (uint32_t vb = blockIdx.x; vb < total; vb += gridDim.x) {
    const uint32_t blk = xcd_remap(a, vb, total);
    const uint64_t b = blk / tiles, c0 = (uint64_t)(blk % tiles) * COLS;
    __syncthreads();  // [B] the exchange buffer is idle
    {
      int t = tid;
      FOURIER_LAUNDER(t);
      const int th = t / CG, cg = t % CG;  // cg-fastest mapping: a wave covers eight 128-byte row segments per access
      // LDS slot of this wave's row r: (r * WAVES + wave) KiB into the exchange buffer, lane-linear (LDS-DMA writes M0 + 16 * lane)
      unsigned char* const slot = smem + (size_t)wave_uniform(t >> 6) * 1024;
      Unit16<T> c[IO == IO_BLU_OUT ? 8 : 1];
      chirp_load(th, cg, c);
      if constexpr (FOURIER_PF_STORES_FIRST != 0) store_tile(th, cg, c, st_bytes);
      // ---- prefetch tile (b, c0): rows th + Q*r, r < NDMA through LDS-DMA, the others into registers
      const cpx<T>* p = in + b * a.n + c0;
      const uint32_t voff_ld = (uint32_t)(((uint64_t)th * a.cn + (uint64_t)(cg * VEC)) * sizeof(cpx<T>));
      Unit16<T> nx[NREG > 0 ? NREG : 1];
#pragma unroll
      for (int r = 0; r < NDMA; ++r)
        lds_dma_unit<LDAUX>(make_rsrc(p + (uint64_t)(Q * r) * a.cn), slot + (size_t)r * WAVES * 1024, voff_ld);
#pragma unroll
      for (int r = NDMA; r < 16; ++r) nx[r - NDMA] = buf_load_unit<T, LDAUX>(make_rsrc(p + (uint64_t)(Q * r) * a.cn), voff_ld);
      FOURIER_SCHED_FENCE();
      // ---- the previous tile's results leave behind the prefetch
      if constexpr (FOURIER_PF_STORES_FIRST == 0) store_tile(th, cg, c, st_bytes);
      FOURIER_SCHED_FENCE();
      // ---- the prefetched tile has landed once at most the stores above are outstanding (in-order completion)
      wait_vmcnt<NSTORES>();
#if FOURIER_PF_BARRIER_AFTER_WAIT
      // vmcnt retires an LDS-DMA when its data has come back, and a ds_read issued right behind the wait can still overtake
      // the LDS write itself (cdna_hip_programming.md: "read a staged buffer one phase AFTER the wait that retires it"): a
      // barrier between the wait and the reads
      __syncthreads();
#endif
      FOURIER_SCHED_FENCE();
      const unsigned char* const mine = slot + (size_t)(t & 63) * 16;
#pragma unroll
      for (int r = 0; r < NDMA; ++r) {
        const Unit16<T> u = *(const Unit16<T>*)(mine + (size_t)r * WAVES * 1024);
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[v][r] = {u.a[2 * v], u.a[2 * v + 1]};
      }
#pragma unroll
      for (int r = NDMA; r < 16; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[v][r] = {nx[r - NDMA].a[2 * v], nx[r - NDMA].a[2 * v + 1]};
    }
    __syncthreads();  // [A] every wave has read its rows back before the first exchange rewrites the buffer
    {
      int t = tid;
      FOURIER_LAUNDER(t);
      int th = t / CG, cg = t % CG;
      tile_core<T, L, CG, MODE_LAST>(x, th, cg, tid, smem, tw1, tw2);
    }
    pb = b; pc0 = c0; st_bytes = 0x7fffffffu;
  }
  if (st_bytes) {  // the last tile's results
    int t = tid;
    FOURIER_LAUNDER(t);
    Unit16<T> c[IO == IO_BLU_OUT ? 8 : 1];
    chirp_load(t / CG, t % CG, c);
    store_tile(t / CG, t % CG, c, st_bytes);
  }
}


}  // namespace fourier_hip
