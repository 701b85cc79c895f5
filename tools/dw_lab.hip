// Lab bench for the CDT weight-gradient GEMM (mlp_dw_big_kernel of osrl_amd/csrc/mlp_dw.hip): standalone, no torch.
//   hipcc -O3 --offload-arch=gfx950 tools/dw_lab.hip -o tools/_lab/dw_lab && tools/_lab/dw_lab
// dW[out,in] = dz^T a over M = 81920 rows for the twelve projection weights of the 3 CDT blocks, one wave per 128 x 64
// tile, 32 row splits (the product's launch: 72 x 32 workgroups).  Variant 0 = the product kernel (48 dword fragment
// loads per 16-row k-step, a row guard on every one); variant 1 = fragment registers filled by dwordx4 loads of 4
// CONSECUTIVE columns (register j of lane m = column 4 m + j: the MFMA's i index runs over a permuted column block, the
// store undoes it with 16-byte stores), guards only in the tail.  Same (row -> MFMA, k-lane) map, i.e. the same bits.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef struct {
  const float* dz;
  const float* a;
  int64_t w_off, b_off;
  int32_t out, in;
  int32_t ldz, lda;
} osrl_dw_entry_t;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kDwbO = 8, kDwbI = 4;

struct DwBigFrag {
  f32x4 a[kDwbO], b[kDwbI];
};

// ---- variant 0: the product kernel as of round 3 -----------------------------------------------------------------
__device__ __forceinline__ void dwb_load(DwBigFrag& f, const float* __restrict__ dz, const float* __restrict__ av,
                                         size_t ldz, size_t lda_g, int r0, int r_end, int m, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = r0 + 4 * kq + t;
    const bool rok = r < r_end;
    const size_t rc = (size_t)(rok ? r : r_end - 1);
    const float* __restrict__ pz = dz + rc * ldz + m;
    const float* __restrict__ pa = av + rc * lda_g + m;
#pragma unroll
    for (int ob = 0; ob < kDwbO; ++ob) {
      const float v = pz[ob * 16];
      f.a[ob][t] = rok ? v : 0.f;
    }
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib) {
      const float v = pa[ib * 16];
      f.b[ib][t] = rok ? v : 0.f;
    }
  }
}

__global__ __launch_bounds__(256, 1) void dw_big_v0(const osrl_dw_entry_t* __restrict__ entries,
                                                    const int32_t* __restrict__ items, int n_items, int rows,
                                                    int rows_per_split, float* __restrict__ slabs, int64_t slab_stride) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2];
  const osrl_dw_entry_t E = entries[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * 16 * kDwbO, i0 = it * 16 * kDwbI;
  const int s = blockIdx.y;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  const bool want_db = it == 0;
  const float* __restrict__ dz = E.dz + o0;
  const float* __restrict__ av = E.a + i0;
  f32x4 acc[kDwbO][kDwbI];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[kDwbO];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] = 0.f;
  if (r_begin < r_end) {
    DwBigFrag f[2];
    dwb_load(f[0], dz, av, ldz, lda_g, r_begin, r_end, m, kq);
    auto step = [&](auto s_c, int r0) {
      constexpr int c = decltype(s_c)::value;
      dwb_load(f[c ^ 1], dz, av, ldz, lda_g, r0 + 16, r_end, m, kq);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
          for (int ib = 0; ib < kDwbI; ++ib)
            acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[c].a[ob][t], f[c].b[ib][t], acc[ob][ib], 0, 0, 0);
      if (want_db) {
#pragma unroll
        for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] += (f[c].a[ob][0] + f[c].a[ob][1]) + (f[c].a[ob][2] + f[c].a[ob][3]);
      }
      __builtin_amdgcn_sched_group_barrier(0x020, 4 * (kDwbO + kDwbI), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kDwbO * kDwbI, 0);
    };
    using std::integral_constant;
    int r0 = r_begin;
    for (; r0 + 32 <= r_end; r0 += 32) {
      step(integral_constant<int, 0>{}, r0);
      step(integral_constant<int, 1>{}, r0 + 16);
    }
    if (r0 < r_end) {
      step(integral_constant<int, 0>{}, r0);
      if (r0 + 16 < r_end) step(integral_constant<int, 1>{}, r0 + 16);
    }
  }
  float* __restrict__ slab = slabs + (size_t)s * slab_stride;
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[E.w_off + (size_t)(o0 + ob * 16 + kq * 4 + r) * in + i0 + ib * 16 + m] = acc[ob][ib][r];
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < kDwbO; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) slab[E.b_off + o0 + ob * 16 + m] = v;
    }
  }
}

// ---- variant 1: dwordx4 fragment loads over permuted column blocks -------------------------------------------------
// Fragment register a[4 q + j] of lane (m, kq) = dz[row][o0 + 64 q + 4 m + j]: as an MFMA operand it is the 16-column
// block {o0 + 64 q + 4 i + j : i = 0..15}.  One dwordx4 load per (64 columns, t): 12 loads per k-step instead of 48, each
// 4 rows x 256 B contiguous.  NB = buffers in flight (2: as variant 0; 3: two k-steps ahead).
template <bool GUARD, bool WRAP = false>
__device__ __forceinline__ void dwp_load(DwBigFrag& f, const float* __restrict__ dz, const float* __restrict__ av,
                                         size_t ldz, size_t lda_g, int r0, int r_end, int m, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = r0 + 4 * kq + t;
    const bool rok = !GUARD || r < r_end;
    const size_t rc = WRAP ? (size_t)(r & 255) : (size_t)(rok ? r : r_end - 1);
    const float* __restrict__ pz = dz + rc * ldz + 4 * m;
    const float* __restrict__ pa = av + rc * lda_g + 4 * m;
#pragma unroll
    for (int q = 0; q < kDwbO / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pz + 64 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) f.a[4 * q + j][t] = rok ? v[j] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < kDwbI / 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pa + 64 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) f.b[4 * q + j][t] = rok ? v[j] : 0.f;
    }
  }
}

template <int NB, bool WRAP = false>
__global__ __launch_bounds__(256, 1) void dw_big_v1(const osrl_dw_entry_t* __restrict__ entries,
                                                    const int32_t* __restrict__ items, int n_items, int rows,
                                                    int rows_per_split, float* __restrict__ slabs, int64_t slab_stride) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2];
  const osrl_dw_entry_t E = entries[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * 16 * kDwbO, i0 = it * 16 * kDwbI;
  const int s = blockIdx.y;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  const bool want_db = it == 0;
  const float* __restrict__ dz = E.dz + o0;
  const float* __restrict__ av = E.a + i0;
  f32x4 acc[kDwbO][kDwbI];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[kDwbO];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] = 0.f;
  auto mma = [&](const DwBigFrag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
        for (int ib = 0; ib < kDwbI; ++ib)
          acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ob][t], f.b[ib][t], acc[ob][ib], 0, 0, 0);
    if (want_db) {
#pragma unroll
      for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] += (f.a[ob][0] + f.a[ob][1]) + (f.a[ob][2] + f.a[ob][3]);
    }
  };
  if (r_begin < r_end) {
    DwBigFrag f[NB];
    // whole 16-row steps whose prefetch (NB - 1 steps ahead) stays inside the split run unguarded
    const int n_steps = (r_end - r_begin + 15) >> 4;
    const int n_full = (r_end - r_begin) >> 4;  // steps with all 16 rows
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) dwp_load<true, WRAP>(f[i], dz, av, ldz, lda_g, r_begin + 16 * i, r_end, m, kq);
    int st = 0;
    auto step = [&](auto c_tag, auto g_tag) __attribute__((always_inline)) {
      constexpr int c = decltype(c_tag)::value;
      constexpr bool G = decltype(g_tag)::value;
      dwp_load<G, WRAP>(f[(c + NB - 1) % NB], dz, av, ldz, lda_g, r_begin + 16 * (st + NB - 1), r_end, m, kq);
      mma(f[c]);
      __builtin_amdgcn_sched_group_barrier(0x020, 4 * (kDwbO + kDwbI) / 4, 0);  // VMEM reads of a later step
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kDwbO * kDwbI, 0);          // this step's MFMAs
      ++st;
    };
    using std::integral_constant;
    using std::true_type;
    using std::false_type;
    // unguarded rounds of NB steps
    while (st + NB - 1 + NB <= n_full) {
      if constexpr (NB == 2) {
        step(integral_constant<int, 0>{}, false_type{});
        step(integral_constant<int, 1>{}, false_type{});
      } else {
        step(integral_constant<int, 0>{}, false_type{});
        step(integral_constant<int, 1>{}, false_type{});
        step(integral_constant<int, 2>{}, false_type{});
      }
    }
    // guarded tail (its loads past r_end read a clamped row and become zeros): continues the buffer rotation
    while (st < n_steps) {
      const int c = st % NB;
      if (c == 0) step(integral_constant<int, 0>{}, true_type{});
      else if (c == 1) step(integral_constant<int, 1>{}, true_type{});
      else step(integral_constant<int, (NB > 2 ? 2 : 0)>{}, true_type{});
    }
  }
  // acc[4 qa + ja][4 qb + jb][r] of lane (n = m, kq) = dW[o0 + 64 qa + 4 (4 kq + r) + ja][i0 + 64 qb + 4 n + jb]
  float* __restrict__ slab = slabs + (size_t)s * slab_stride;
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int qb = 0; qb < kDwbI / 4; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = o0 + 64 * (ob >> 2) + 4 * (4 * kq + r) + (ob & 3);
        const f32x4 v = {acc[ob][4 * qb + 0][r], acc[ob][4 * qb + 1][r], acc[ob][4 * qb + 2][r], acc[ob][4 * qb + 3][r]};
        *reinterpret_cast<f32x4*>(&slab[E.w_off + (size_t)orow * in + i0 + 64 * qb + 4 * m]) = v;
      }
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < kDwbO; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) slab[E.b_off + o0 + 64 * (ob >> 2) + 4 * m + (ob & 3)] = v;
    }
  }
}

// ---- variant 2 = variant 1 with running fragment pointers in the unguarded loop (8 64-bit adds per k-step instead of
// the row * stride multiplies: the f32 MFMA runs on the vector ALUs, every VALU instruction is time taken from it)
__global__ __launch_bounds__(256, 1) void dw_big_v2(const osrl_dw_entry_t* __restrict__ entries,
                                                    const int32_t* __restrict__ items, int n_items, int rows,
                                                    int rows_per_split, float* __restrict__ slabs, int64_t slab_stride) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x * 4 + wave;
  if (item >= n_items) return;
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2];
  const osrl_dw_entry_t E = entries[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * 16 * kDwbO, i0 = it * 16 * kDwbI;
  const int s = blockIdx.y;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  const bool want_db = it == 0;
  const float* __restrict__ dz = E.dz + o0;
  const float* __restrict__ av = E.a + i0;
  f32x4 acc[kDwbO][kDwbI];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[kDwbO];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] = 0.f;
  auto mma = [&](const DwBigFrag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
        for (int ib = 0; ib < kDwbI; ++ib)
          acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ob][t], f.b[ib][t], acc[ob][ib], 0, 0, 0);
    if (want_db) {
#pragma unroll
      for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] += (f.a[ob][0] + f.a[ob][1]) + (f.a[ob][2] + f.a[ob][3]);
    }
  };
  if (r_begin < r_end) {
    DwBigFrag f[2];
    const int n_steps = (r_end - r_begin + 15) >> 4;
    const int n_full = (r_end - r_begin) >> 4;
    // the rows lane (m, kq) reads in a step: r0 + 4 kq + t, t = 0..3 (consecutive rows: one pointer + t * stride)
    const float* pz = dz + (size_t)(r_begin + 4 * kq) * ldz + 4 * m;
    const float* pa = av + (size_t)(r_begin + 4 * kq) * lda_g + 4 * m;
    const size_t zstep = 16 * ldz, astep = 16 * lda_g;
    auto load_fast = [&](DwBigFrag& g) __attribute__((always_inline)) {  // the step the pointers stand on; advances them
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int q = 0; q < kDwbO / 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(pz + t * ldz + 64 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) g.a[4 * q + j][t] = v[j];
        }
#pragma unroll
        for (int q = 0; q < kDwbI / 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(pa + t * lda_g + 64 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) g.b[4 * q + j][t] = v[j];
        }
      }
      pz += zstep;
      pa += astep;
    };
    int st = 0;
    if (n_full >= 1) load_fast(f[0]);
    else dwp_load<true>(f[0], dz, av, ldz, lda_g, r_begin, r_end, m, kq);
    // unguarded pairs of steps: the prefetched step st + 1 / st + 2 must be whole
    while (st + 3 <= n_full) {
      load_fast(f[1]);
      mma(f[0]);
      __builtin_amdgcn_sched_group_barrier(0x020, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kDwbO * kDwbI, 0);
      load_fast(f[0]);
      mma(f[1]);
      __builtin_amdgcn_sched_group_barrier(0x020, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kDwbO * kDwbI, 0);
      st += 2;
    }
    // guarded tail: f[st & 1] holds step st
    while (st < n_steps) {
      if ((st & 1) == 0) {
        dwp_load<true>(f[1], dz, av, ldz, lda_g, r_begin + 16 * (st + 1), r_end, m, kq);
        mma(f[0]);
      } else {
        dwp_load<true>(f[0], dz, av, ldz, lda_g, r_begin + 16 * (st + 1), r_end, m, kq);
        mma(f[1]);
      }
      ++st;
    }
  }
  float* __restrict__ slab = slabs + (size_t)s * slab_stride;
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int qb = 0; qb < kDwbI / 4; ++qb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = o0 + 64 * (ob >> 2) + 4 * (4 * kq + r) + (ob & 3);
        const f32x4 v = {acc[ob][4 * qb + 0][r], acc[ob][4 * qb + 1][r], acc[ob][4 * qb + 2][r], acc[ob][4 * qb + 3][r]};
        *reinterpret_cast<f32x4*>(&slab[E.w_off + (size_t)orow * in + i0 + 64 * qb + 4 * m]) = v;
      }
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < kDwbO; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) slab[E.b_off + o0 + 64 * (ob >> 2) + 4 * m + (ob & 3)] = v;
    }
  }
}

// ---- variant 3: one 8-wave workgroup per 256 x 256 tile and row split, operands through LDS -----------------------------
// The waves of variants 0-2 stream private 2560 x 192 blocks (18 GB per launch through L2 / MALL).  Here a workgroup's
// eight waves (2 x 4: 128 out-columns x 64 in-columns each) share one DMA-fed slab ring: 16 rows x (256 + 256) columns
// = 32 KB per k-step, every piece one whole 1 KB row (lane l <- 16 bytes at column 4 l: the same permuted-column fragment
// registers as variant 1, now read from LDS with conflict-free ds_read_b128), 4 slots, fragments double-buffered in
// registers, the memory-instruction issue point staggered between the two waves of a SIMD (linear_pers_kernel's loop).
// 6 GB per launch, and 7 row splits instead of 32 (36 tiles x 7 = 252 workgroups: one round): the slab reduction that
// follows reads 7 slabs.  Requires out % 256 == 0, in % 256 == 0, rows_per_split % 16 == 0, rows % 16 == 0.
constexpr int kCoS = 4;  // slots
__device__ __forceinline__ void glds16(const void* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__global__ __launch_bounds__(512, 2) void dw_coop(const osrl_dw_entry_t* __restrict__ entries,
                                                  const int32_t* __restrict__ items, int n_items, int rows,
                                                  int rows_per_split, float* __restrict__ slabs, int64_t slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds_c[];
  constexpr int kSlot = 2 * 16 * 256;  // floats: [mat][row][256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wo = wave >> 2, wi = wave & 3;
  const int item = blockIdx.x;
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2];
  const osrl_dw_entry_t E = entries[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * 256, i0 = it * 256;
  const int s = blockIdx.y;
  const int r_begin = s * rows_per_split;
  int r_end = r_begin + rows_per_split;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  const bool want_db = it == 0 && wi == 0;
  const int G = (r_end - r_begin) >> 4;  // k-steps (whole: host)
  f32x4 acc[kDwbO][kDwbI];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int ib = 0; ib < kDwbI; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbacc[kDwbO];
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] = 0.f;
  if (G > 0) {
    // ---- issue side: pieces p = 0..3 of this wave: (matrix, row) = ((4 wave + p) >> 4, (4 wave + p) & 15)
    const char* src[4];
    size_t step_b[4];
    int dst_off[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = 4 * wave + p, mat = idx >> 4, row = idx & 15;
      const float* base = mat ? E.a + i0 : E.dz + o0;
      const size_t ld = mat ? lda_g : ldz;
      src[p] = reinterpret_cast<const char*>(base + (size_t)(r_begin + row) * ld + 4 * lane);
      step_b[p] = 16 * ld * sizeof(float);
      dst_off[p] = mat * 16 * 256 + row * 256;  // floats; the wave writes the whole 1 KB row (lane * 16 B)
    }
    int it_slot = 0, it_g = 0;
    auto dma = [&]() __attribute__((always_inline)) {
      if (it_g < G) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          glds16(src[p], lds_c + it_slot * kSlot + dst_off[p]);
          src[p] += step_b[p];
        }
      }
      ++it_g;
      it_slot = it_slot + 1 == kCoS ? 0 : it_slot + 1;
    };
    // ---- consume side
    const int z_off = (4 * kq) * 256 + wo * 128 + 4 * m;             // + t * 256 + 64 q
    const int a_off = 16 * 256 + (4 * kq) * 256 + wi * 64 + 4 * m;   // + t * 256
    DwBigFrag f;  // single-buffered: 128 accumulator + 48 fragment registers of the 256 a wave has at two per SIMD
    auto rd = [&](int slot, DwBigFrag& g) __attribute__((always_inline)) {
      const float* sl = lds_c + slot * kSlot;
#pragma unroll
      for (int t = 0; t < 4; ++t) {  // by t: the MFMAs of t = 0 need only the first three reads
#pragma unroll
        for (int q = 0; q < kDwbO / 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sl + z_off + t * 256 + 64 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) g.a[4 * q + j][t] = v[j];
        }
        const f32x4 v = *reinterpret_cast<const f32x4*>(sl + a_off + t * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j) g.b[j][t] = v[j];
      }
    };
    auto mma_t = [&](const DwBigFrag& g, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
        for (int ib = 0; ib < kDwbI; ++ib)
          acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.a[ob][t], g.b[ib][t], acc[ob][ib], 0, 0, 0);
    };
    constexpr int L = kCoS - 1, DMA_OPS = 4;
#pragma unroll
    for (int i = 0; i < L; ++i) dma();
    int c_slot = 0;
    for (int g = 0; g < G; ++g) {
      // top of k-step g: slab g landed for everyone (slabs g + 1, g + 2 may be in flight); everyone's MFMAs of step
      // g - 1 are issued, i.e. the slot of slab g - 1 is free for slab g + 3
      if (g + L <= G) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((L - 1) * DMA_OPS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      rd(c_slot, f);
      c_slot = c_slot + 1 == kCoS ? 0 : c_slot + 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if ((t == 0 && wo == 0) || (t == 2 && wo == 1)) dma();
        mma_t(f, t);
      }
      if (want_db) {
#pragma unroll
        for (int ob = 0; ob < kDwbO; ++ob) dbacc[ob] += (f.a[ob][0] + f.a[ob][1]) + (f.a[ob][2] + f.a[ob][3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float* __restrict__ slab = slabs + (size_t)s * slab_stride;
  const int ow = o0 + wo * 128, iw = i0 + wi * 64;
#pragma unroll
  for (int ob = 0; ob < kDwbO; ++ob)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int orow = ow + 64 * (ob >> 2) + 4 * (4 * kq + r) + (ob & 3);
      const f32x4 v = {acc[ob][0][r], acc[ob][1][r], acc[ob][2][r], acc[ob][3][r]};
      *reinterpret_cast<f32x4*>(&slab[E.w_off + (size_t)orow * in + iw + 4 * m]) = v;
    }
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < kDwbO; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) slab[E.b_off + ow + 64 * (ob >> 2) + 4 * m + (ob & 3)] = v;
    }
  }
}

template <class K>
float run(K k, const osrl_dw_entry_t* de, const int32_t* di, int n_items, int rows, int n_splits, float* slabs,
          int64_t stride, int reps) {
  int rps = (rows + n_splits - 1) / n_splits;
  rps = (rps + 15) & ~15;
  constexpr int kLds = 96 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  dim3 grid((n_items + 3) / 4, n_splits, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, grid, dim3(256), kLds, 0, de, di, n_items, rows, rps, slabs, stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(256), kLds, 0, de, di, n_items, rows, rps, slabs, stride);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 81920;
  const int n_splits = argc > 2 ? atoi(argv[2]) : 32;
  const int E = 256, NL = 3;
  struct Sh { int out, in; };
  const Sh shapes[4] = {{3 * E, E}, {E, E}, {4 * E, E}, {E, 4 * E}};
  // operands: per layer dz [M, out], a [M, in]: shared random buffers of the widest shapes
  std::vector<float> h((size_t)M * 1024);
  uint32_t s = 777;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  float *dZ[NL], *dA[NL];
  for (int l = 0; l < NL; ++l) {
    CK(hipMalloc(&dZ[l], h.size() * 4));
    CK(hipMalloc(&dA[l], h.size() * 4));
    CK(hipMemcpy(dZ[l], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dA[l], h.data() + 4096 * l + 128, (h.size() - 4096 * l - 128) * 4, hipMemcpyHostToDevice));
  }
  std::vector<osrl_dw_entry_t> ents;
  std::vector<int32_t> items;
  int64_t off = 0;
  double gf = 0;
  for (int l = 0; l < NL; ++l)
    for (const Sh& sh : shapes) {
      osrl_dw_entry_t e;
      e.dz = dZ[l]; e.a = dA[l];
      e.out = sh.out; e.in = sh.in; e.ldz = 0; e.lda = 0;
      e.w_off = off; off += (int64_t)sh.out * sh.in;
      e.b_off = off; off += sh.out;
      off = (off + 3) & ~3ll;
      const int ei = (int)ents.size();
      ents.push_back(e);
      for (int ot = 0; ot < sh.out / 128; ++ot)
        for (int it = 0; it < sh.in / 64; ++it) { items.push_back(ei); items.push_back(ot); items.push_back(it); items.push_back(0); }
      gf += 2.0 * M * sh.out * sh.in * 1e-9;
    }
  const int n_items = (int)items.size() / 4;
  const int64_t stride = off;
  osrl_dw_entry_t* de;
  int32_t* di;
  float *s0, *s1;
  CK(hipMalloc(&de, ents.size() * sizeof(osrl_dw_entry_t)));
  CK(hipMalloc(&di, items.size() * 4));
  CK(hipMemcpy(de, ents.data(), ents.size() * sizeof(osrl_dw_entry_t), hipMemcpyHostToDevice));
  CK(hipMemcpy(di, items.data(), items.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&s0, (size_t)stride * n_splits * 4));
  CK(hipMalloc(&s1, (size_t)stride * n_splits * 4));
  printf("M=%d items=%d splits=%d  %.1f GF = %.0f us at 157.3 TF/s\n", M, n_items, n_splits, gf, gf / 157.3 * 1e3);
  const int reps = 5;
  CK(hipMemset(s0, 0, (size_t)stride * n_splits * 4));
  const float t0 = run(dw_big_v0, de, di, n_items, M, n_splits, s0, stride, reps);
  printf("  v0      %8.1f us  %.3f of peak\n", t0, gf * 1e3 / t0 / 157.3);
  std::vector<float> r0((size_t)stride * n_splits), r1((size_t)stride * n_splits);
  CK(hipMemcpy(r0.data(), s0, r0.size() * 4, hipMemcpyDeviceToHost));
  auto check = [&](const char* name, float t) {
    CK(hipMemcpy(r1.data(), s1, r1.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    printf("  %-7s %8.1f us  %.3f of peak   elements that differ from v0: %zu\n", name, t, gf * 1e3 / t / 157.3, bad);
  };
  CK(hipMemset(s1, 0, (size_t)stride * n_splits * 4));
  check("v1 nb2", run(dw_big_v1<2>, de, di, n_items, M, n_splits, s1, stride, reps));
  CK(hipMemset(s1, 0, (size_t)stride * n_splits * 4));
  check("v1 nb3", run(dw_big_v1<3>, de, di, n_items, M, n_splits, s1, stride, reps));
  CK(hipMemset(s1, 0, (size_t)stride * n_splits * 4));
  check("v2", run(dw_big_v2, de, di, n_items, M, n_splits, s1, stride, reps));
  {
    std::vector<int32_t> it2;
    for (int e = 0; e < (int)ents.size(); ++e)
      for (int ot = 0; ot < ents[e].out / 256; ++ot)
        for (int itt = 0; itt < ents[e].in / 256; ++itt) { it2.push_back(e); it2.push_back(ot); it2.push_back(itt); it2.push_back(0); }
    const int n2 = (int)it2.size() / 4;
    int32_t* di2;
    CK(hipMalloc(&di2, it2.size() * 4));
    CK(hipMemcpy(di2, it2.data(), it2.size() * 4, hipMemcpyHostToDevice));
    // reference: the split sums of v0, in double
    std::vector<double> ref((size_t)stride, 0.0);
    for (int sp = 0; sp < n_splits; ++sp)
      for (int64_t i = 0; i < stride; ++i) ref[i] += r0[(size_t)sp * stride + i];
    for (int S2 : {7, 14}) {
      int rps = (M + S2 - 1) / S2;
      rps = (rps + 15) & ~15;
      const size_t lds = sizeof(float) * kCoS * 2 * 16 * 256;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(dw_coop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipMemset(s1, 0, (size_t)stride * n_splits * 4));
      auto launch = [&] { hipLaunchKernelGGL(dw_coop, dim3(n2, S2, 1), dim3(512), lds, 0, de, di2, n2, M, rps, s1, stride); };
      for (int i = 0; i < 2; ++i) launch();
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const float t = ms * 1000.f / reps;
      CK(hipMemcpy(r1.data(), s1, (size_t)stride * S2 * 4, hipMemcpyDeviceToHost));
      double maxerr = 0, maxref = 0;
      for (int64_t i = 0; i < stride; ++i) {
        double v = 0;
        for (int sp = 0; sp < S2; ++sp) v += r1[(size_t)sp * stride + i];
        maxerr = fmax(maxerr, fabs(v - ref[i]));
        maxref = fmax(maxref, fabs(ref[i]));
      }
      printf("  coop %2d splits (%d workgroups) %8.1f us  %.3f of peak   max |sum - sum_v0| %.3e (max |sum| %.3e)\n", S2, n2 * S2, t,
             gf * 1e3 / t / 157.3, maxerr, maxref);
    }
  }
  check("v1 wrap", run(dw_big_v1<2, true>, de, di, n_items, M, n_splits, s1, stride, reps));  // every row read = one of 256: cache-resident operands (results differ by construction)
  return 0;
}
