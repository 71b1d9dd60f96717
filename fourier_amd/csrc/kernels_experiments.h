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

}  // namespace fourier_hip
