// mlp_common.h -- what the fused-MLP translation units share (mlp.hip: the tile kernels, backward, dW, the one-launch
// step; mlp_nb.hip: the 80-row N*B-row forward): vector types, activation helpers, the packed-weight fragment load,
// the k-walk rotation, the debug-build stamp macros.  Everything sits in an anonymous namespace: each unit gets its own
// copy, nothing is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/osrl_amd.h"
#include "argmem.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The in-launch exchanges of this library (seed statistics in mlp.hip, slab tiles in mlp_dwt_adam_body, the one-launch BC
// step's arrival counter) publish device-coherent words as: relaxed agent-scope stores (sc1) -> `s_waitcnt vmcnt(0)`
// (asm volatile with a memory clobber: neither the compiler nor the wave moves the arrival atomic above it) -> relaxed
// agent-scope fetch_add; readers: barrier -> relaxed agent-scope loads.  That is a release/acquire on gfx942/gfx950
// because sc1 stores are tracked by vmcnt and complete at the memory side; a target with a separate store counter
// (vscnt) would need `s_waitcnt_vscnt` there.  Pin the family instead of hoping.  gfx950 is the ONLY supported
// target of this library (README "Build"; every tile shape, LDS budget and ISA guard is calibrated on it) -- the protocol
// itself would also hold on gfx942, which is deliberately not admitted here:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "osrl_amd's in-launch exchanges assume gfx950 memory counters (sc1 stores tracked by vmcnt): build with --offload-arch=gfx950"
#endif

#ifdef OSRL_PHASE_TIMING  // tools/mlp_phase.hip: per-phase cycle stamps of workgroup 0 (debug builds only)
__device__ long long g_phase_t[4][64];
__device__ long long g_phase_all[8192][4][16];  // every workgroup (first 8192), for phase averages
#define PHASE_STAMP(i)                                                                                  \
  if ((threadIdx.x & 63) == 0) {                                                                        \
    const long long t_ = __builtin_readcyclecounter();                                                  \
    const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                                \
    if (wg_ < 8192 && (i) < 16 && threadIdx.x < 256) g_phase_all[wg_][threadIdx.x >> 6][i] = t_;                             \
    if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x < 256) g_phase_t[threadIdx.x >> 6][i] = t_;           \
  }
// residency log: (start, end) in 100 MHz ticks, HW_ID, XCC_ID of every workgroup
__device__ long long g_wg_log[16384][4];
#define WG_LOG(slot)                                                                                  \
  if (threadIdx.x == 0) {                                                                             \
    const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                              \
    if (wg_ < 16384) {                                                                                \
      g_wg_log[wg_][slot] = wall_clock64();                                                           \
      g_wg_log[wg_][2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                   \
      g_wg_log[wg_][3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                  \
    }                                                                                                 \
  }
#elif defined(OSRL_STEP_STAMPS)  // tools/step_stamps.py: per-layer wall-clock stamps inside the one-launch BC step's forward /
// backward bodies (thread 0 of every workgroup, 100 MHz counter); [0] = mlp_fwd_body's PHASE_STAMP sites, [1] = BWD_STAMP
__device__ long long g_step_phase[OSRL_STEP_MAX_WG][2][16];
#define PHASE_STAMP(i) \
  if (threadIdx.x == 0 && blockIdx.x < OSRL_STEP_MAX_WG && blockIdx.y == 0 && (i) < 16) g_step_phase[blockIdx.x][0][i] = wall_clock64();
#define BWD_STAMP(i) \
  if (threadIdx.x == 0 && blockIdx.x < OSRL_STEP_MAX_WG && blockIdx.y == 0 && (i) < 16) g_step_phase[blockIdx.x][1][i] = wall_clock64();
#define WG_LOG(slot)
#else
#define PHASE_STAMP(i)
#define WG_LOG(slot)
#endif
#ifndef BWD_STAMP
#define BWD_STAMP(i)
#endif

// Wave priority (s_setprio 0..3, default 0): launches on at most OSRL_CHAIN_PRIO rows -- the 2048-row latency chain of a
// train step: forwards with saved activations, backward-dz, dW -- raise theirs to 3, so that on a CU they share with
// the N*B-row inference launches (the step's filler work, priority 0) the instruction arbiter serves the chain first.
// Measured on the CPQ step: +1.3 % (1945 -> 1970 steps/s); 0 disables.
#ifndef OSRL_CHAIN_PRIO
#define OSRL_CHAIN_PRIO 4096
#endif

namespace {

// max(x, 0) as ONE v_max_f32: fmaxf() compiles to a canonicalising v_max x,x in front of the max (IEEE sNaN quieting),
// and every VALU instruction of an epilogue is paid in MFMA issue slots (4 cycles per wave each).  Same value for
// every non-NaN input.
__device__ __forceinline__ float relu1(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ float act_fwd(int act, float x) {
  if (act == OSRL_ACT_RELU) return relu1(x);
  if (act == OSRL_ACT_TANH) return tanhf(x);
  return x;
}
// derivative expressed with the activation OUTPUT y (relu: threshold_backward on the output)
__device__ __forceinline__ float act_bwd(int act, float y) {
  if (act == OSRL_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
  if (act == OSRL_ACT_TANH) return 1.0f - y * y;
  return 1.0f;
}
__device__ __forceinline__ int map_row(int r, int map, int div) {
  if (map == OSRL_MAP_MOD) return r % div;
  if (map == OSRL_MAP_DIV) return r / div;
  return r;
}
__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }
// balanced split of nblk column blocks over the NW waves: the first (nblk % NW) waves take one extra block
template <int NW = 4>
__device__ __forceinline__ void wave_blocks(int nblk, int wave, int* cb0, int* cnt) {
  const int base = nblk / NW, rem = nblk % NW;
  *cnt = base + (wave < rem ? 1 : 0);
  *cb0 = wave * base + (wave < rem ? wave : rem);
}

// B fragment (4 consecutive k of column n) from the packed layout P[q = k/4][n][4], Np columns, with the k-step
// part of the address kept scalar: P + kc*16*Np is wave-uniform (SGPR pair), the lane part (kq*Np + n)*16 bytes is a
// 32-bit VGPR offset computed once per layer -> global_load_dwordx4 saddr+voffset
__device__ __forceinline__ f32x4 load_bp_s(const float* __restrict__ Pk /*uniform*/, unsigned lane_off_bytes) {
#ifdef OSRL_EXP_NO_BLOAD
  const float v = (float)lane_off_bytes * 1e-6f;
  return f32x4{v, v + 1.f, v + 2.f, v + 3.f};
#else
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Pk) + lane_off_bytes);
#endif
}

// ---- the MFMA core -------------------------------------------------------------------------------------
// acc[rb][c] += A(lds tile rows rb*16.., k) * B(k, cols n0 + c*16..)   over nk 16-deep k steps.
// Software pipeline: a ring of STAGES B-fragment sets keeps STAGES-1 k-steps of weight loads in flight
// (global -> VGPR, straight from L2) while the MFMAs of the current step run; the A fragments
// (ds_read_b128 from the LDS activation tile) are prefetched one step ahead.
// The pipeline is split in two calls so that a layer's FIRST weight loads can be issued long before its
// k-loop starts -- before the previous layer's barrier + epilogue, or before the input tile is staged:
//   mm_prefetch  issues the loads of the first STAGES-1 k-steps into the ring (no waits);
//   mm_run       runs the k-loop assuming exactly that.
// Measured (tools/mlp_phase.hip, all workgroups): without this a wave spent 22k cycles in the 5-k-step first
// layer (5k cycles of MFMA work) and 13k cycles staging its input with nothing else in flight.
constexpr int kRing = 3;  // ring slots (STAGES <= ring depth) of the many-workgroups-per-CU kernels
// Ring depth of the 8-wave kernels (NW = 8: launches of at most ~2 workgroups per CU -- the 2048-row training launches,
// BC's 256 rows), an EXPERIMENT knob: nothing else on the CU hides a weight load's latency there, so a deeper ring
// (OSRL_RING_DEEP = 4 / 6: 3 / 5 k-steps of weights in flight) looked like the remedy for their 13k-cycle 16-k-step
// layers (8k of MFMA time).  Measured (tools/mlp_phase.hip variants, profiles/r3_phase_ring_warm.txt): depth 4 changes
// a layer by -4 % .. +2 %, depth 6 is 20-30 % SLOWER (registers: the ring is live across staging and epilogues), and
// it makes no difference whether the weights were just re-written from another XCD (COLD=1) or are L2-hot -- these
// layers are chains of ~700-cycle round trips (weights, LDS, barriers) of which the weight ring is only one.  Default 3.
#ifndef OSRL_RING_DEEP
#define OSRL_RING_DEEP 3
#endif
constexpr int kRingDeep = OSRL_RING_DEEP;
template <int NW>
constexpr int ring_depth() { return NW == 8 ? kRingDeep : kRing; }
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}
// L2 warm-up for the 8-wave kernels, the second EXPERIMENT of the same measurement (OSRL_L2_WARM=1): each thread
// touches kWarmLines 128-byte lines of the next layers' weights while the current layer computes.  It costs 10-25 %
// (the first layer's k-loop waits behind the touches: loads return in order) and buys nothing -- see above, the
// layers are not bound by where the weights come from.  Off; kept for the A/B build of tools/build_phase_variants.sh.
constexpr int kWarmLines = 4;  // x 512 threads x 128 B = 256 KB per layer (a 256 x 256 layer)
#ifndef OSRL_L2_WARM
#define OSRL_L2_WARM 0
#endif
template <int NT>
__device__ __forceinline__ void l2_warm(const float* __restrict__ P, int n_floats, float (&d)[kWarmLines]) {
  const int n_lines = n_floats >> 5;
#pragma unroll
  for (int j = 0; j < kWarmLines; ++j) {
    int i = (int)threadIdx.x + j * NT;
    i = i < n_lines ? i : n_lines - 1;  // (past the end: touch the last line again -- no branch around a load)
    d[j] = P[(size_t)i * 32];
  }
}
__device__ __forceinline__ void l2_warm_done(float (&d)[kWarmLines]) {
#pragma unroll
  for (int j = 0; j < kWarmLines; ++j) asm volatile("" ::"v"(d[j]));
}
#ifndef OSRL_PIN_ROWS
#define OSRL_PIN_ROWS 5
#endif
constexpr int kPinRows = OSRL_PIN_ROWS;  // row blocks per tile from which mm_run pins its in-step instruction order

// Every workgroup needs the SAME weight lines; each starts its k-walk at a different step so the
// request streams are decorrelated (fp32 sum order changes per workgroup; fixed per (grid, tile)).
// (wave index through readfirstlane: rot, every k index and the weight base address stay in SGPRs; a per-lane
// k costs two 64-bit VALU multiply-adds per weight load, and VALU issue time adds to -- does not hide behind --
// the MFMA time of the other waves on the SIMD: measured 2.7 VALU instructions per MFMA before this)
__device__ __forceinline__ int k_rot(int nk) {
  const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  return (int)((blockIdx.x * 5u + blockIdx.y * 3u + w) % (unsigned)nk);
}
__device__ __forceinline__ int k_at(int kc, int rot, int nk, int kc0) {  // nk steps starting at kc0 (split-K sub-range)
  const int k = kc + rot;
  return kc0 + (k >= nk ? k - nk : k);
}

// experiment switches of tools/mlp_phase.hip (never defined in the product build)
#ifdef OSRL_EXP_NO_MFMA
__device__ __forceinline__ f32x4 EXP_MFMA(float a, float b, f32x4 c) {
  c[0] += a * b;  // one VALU FMA keeps the operands alive; no matrix instruction
  return c;
}
#else
#define EXP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#endif
#ifdef OSRL_EXP_NO_AREAD
#define EXP_AREAD(p) (f32x4{(float)(size_t)(p), 1.f, 2.f, 3.f})
#else
#define EXP_AREAD(p) (*reinterpret_cast<const f32x4*>(p))
#endif

template <int NRB, int NCB>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[NRB][NCB]) {
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int c = 0; c < NCB; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// the 80-row forward lives in mlp_nb.hip; mlp.hip's osrl_mlp_forward asks it first (kNbNotTaken: the shape is not its)
constexpr int kNbNotTaken = -12345;
constexpr size_t kLdsMax = 160 * 1024;
}  // namespace
// kl / kl_L: the OSRL_TAIL_VAE_KL tail (per-row KL of net 0's (mean | log_std) output), NULL = none
__attribute__((visibility("hidden"))) int osrl_launch_fwd_nb(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out, hipStream_t stream,
                                                             float* kl = nullptr, int kl_L = 0);
// the same kernels on 64-row tiles (mlp_nb64.hip = mlp_nb.hip compiled with OSRL_NB_RB = 4)
__attribute__((visibility("hidden"))) int osrl_launch_fwd_nb64(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out, hipStream_t stream,
                                                             float* kl = nullptr, int kl_L = 0);
