// vae_ns.hip -- the VAE's forward / backward on the training rows as ALL-CU layer launches (round 5).
//
// Replaces, for 400-wide VAEs on >= 1024 rows, the four fused 16-row-tile launches of the VAE phase (osrl_mlp_forward_tail
// enc, osrl_mlp_forward dec, osrl_mlp_backward_dz_seed dec, osrl_mlp_backward_dz enc; VAE.forward net.py:319-339 + the
// loss of cpq.py:125-135 and its autograd): at 2048 rows those occupy 128 of the 256 CUs for 25-27 us each and spend most
// of that walking the 400 x 400 layer on one row tile per CU (tools/chain_lab.hip: 21.9 us for that layer alone, 2.6x its
// MFMA time whatever the ring depth or the row-tile height).  Here (tools/nsplit_lab.hip: 11.4 us for the same layer):
//
//   * every H x H layer is ONE launch of [48 rows x 80 columns] output tiles: (rows / 48) x (H / 80) = 215 workgroups at
//     2048 x 400 -- the whole chip --, 4 waves split K (7/6/6/6 k-steps of 16), each wave keeps the tile's 3 x 5
//     accumulator blocks and streams its weight fragments through a 4-deep register ring; the four partial tiles meet in
//     LDS, all threads add them, apply bias / activation (or relu'), write coalesced rows;
//   * the narrow layers never get a launch of their own beyond the first: layer 0 of the encoder and the observation part
//     of the decoder's layer 0 (P = obs W0[:, :od]^T, independent of z) are one launch; the heads (mean | log_std, u) and
//     dL/dz are [rows x <= 32] products whose K runs over the wide layer's OUTPUT columns -- each 80-column workgroup adds
//     its share as a "slab" row (split-K over the column groups, MFMA on the LDS-resident output tile), and the NEXT
//     launch's prologue sums the H / 80 slabs of its 48 rows, finishes the row-local arithmetic (z = mean + sd eps; u, the
//     MSE's dY, the logged loss; dL/d(mean | log_std)) and turns the result into its own A operand;
//   * a generated A operand ( h0 = relu(P + Wz z + b);  dZ1 = (dZ2 W2) * relu'(h1) ) is a small-K MFMA product of the
//     prologue's [48 x <= 32] row tile with the matching weight columns, written by each wave into a PRIVATE LDS region for
//     its own K range (no workgroup barrier), fixed up element-wise against a row-major operand (P + bias / h1) and read
//     back as fragments; the rows a later dW launch needs (h0, dZ1) leave from there, each column group storing a fifth.
//
// Resource fit is part of the design: the launches run beside the N*B-row launch of the side branch
// (84 KB of LDS, 197 registers on one wave per SIMD), so a workgroup here holds <= 64.5 KB of LDS and <= 312 registers.
// Fragment conventions (mlp_common.h): A lane (m = lane & 15, kq = lane >> 4) holds A[m][k0 + 4 kq .. + 3], B lane holds
// P[(k0 / 4 + kq)][n0 + m][0 .. 3], MFMA t of a k-step uses element t of both; acc[r] = out[4 kq + r][m].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "argmem.h"
#include "trace.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "vae_ns.hip: the in-launch statistic exchange assumes gfx950 memory counters (see mlp_common.h)"
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 48, kBN = 80, kKS = 4;   // tile rows, tile columns, K-split waves
constexpr int kLD = kBN + 4;                 // row stride of a [48][80] tile in LDS
constexpr int kSW = 32;                      // slab row width (floats)
constexpr int kSL = 36;                      // row stride of the prologue's [48][<= 32] row tile
constexpr int kPL = 68;                      // row stride of a wave's private [48][64] A region
constexpr int kNKW = 7;                      // k-steps per wave, upper bound (H <= 448)
constexpr float kLsMin = -4.0f, kLsMax = 15.0f;  // net.py:325

// Wave priority (mlp_common.h OSRL_CHAIN_PRIO): these launches ARE the latency chain's VAE phase and run beside the N*B-row
// cost-critic launch of the side branch (priority 0) on every CU
#ifndef OSRL_VAE_NS_PRIO
#define OSRL_VAE_NS_PRIO 0
#endif
#if OSRL_VAE_NS_PRIO > 0
#define NS_PRIO() __builtin_amdgcn_s_setprio(OSRL_VAE_NS_PRIO)
#else
#define NS_PRIO()
#endif

#ifdef VAE_NS_STAMPS  // tools/vae_ns_lab.hip: per-phase 100 MHz stamps of wave 0 of every workgroup (lab builds only)
__device__ unsigned long long g_ns_stamp[512][12];
#define NS_STAMP(i)                                                                                     \
  if (threadIdx.x == 0 && blockIdx.x < 512) g_ns_stamp[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();
#else
#define NS_STAMP(i)
#endif

struct NsArgs {
  int rows, od, ad, L, H, row_tiles, col_groups;
  float max_action, beta, inv_rows;
  const float *obs, *act, *eps;
  const float *e0f, *e1f, *e2f, *e1b, *e2b, *eb0, *eb1, *eb2;
  const float *d0f, *d1f, *d2f, *d0b, *d1b, *d2b, *db0, *db1, *db2;
  float *enc_x, *enc_h0, *enc_h1, *enc_head, *z, *dec_x, *dec_h0, *dec_h1, *dec_u;
  float *enc_dz0, *enc_dz1, *enc_dz2, *dec_dz0, *dec_dz1, *dec_dz2;
  float *P, *slabE, *slabD, *slabX;
  float* partials;
  uint32_t* counter;
  float* stat;
};

__device__ __forceinline__ int r16(int x) { return (x + 15) & ~15; }
__device__ __forceinline__ void mfma(f32x4& acc, float a, float b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void coh_put(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coh_get(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// blockIdx -> (row tile, column group): ids with the same (id % 8) run on one XCD, so each XCD gets WHOLE row tiles (the
// column groups of a tile share its A rows through that XCD's L2).  false = an id beyond the last tile.
__device__ __forceinline__ bool tile_of(int row_tiles, int col_groups, int* tile, int* cg) {
  const int id = blockIdx.x, x = id & 7, s = id >> 3;
  *tile = x + 8 * (s / col_groups);
  *cg = s % col_groups;
  return *tile < row_tiles;
}

// contiguous K ranges over the 4 waves: the first (nk % 4) waves take one k-step more
__device__ __forceinline__ void k_range(int nk, int wave, int* ks0, int* cnt) {
  const int base = nk / kKS, extra = nk % kKS;
  *cnt = base + (wave < extra ? 1 : 0);
  *ks0 = wave * base + (wave < extra ? wave : extra);
}

// ---- layer 0 of the encoder | the observation part of the decoder's layer 0 -------------------------------------------
// one workgroup = 4 waves on one [32 rows x 80 columns] tile: waves 0, 1 -> enc_h0 = relu(x W0e^T + b), x = [obs | act],
// one 16-row block each; waves 2, 3 -> P = obs W0d[:, :od]^T (no bias).  The concatenated inputs of both nets (dW of
// layer 0 reads them) are written by the first column group.
template <int NKW, class AR>
__device__ __forceinline__ void l0_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  NS_PRIO();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int tile = blockIdx.x / a.col_groups, cg = blockIdx.x % a.col_groups;
  const int row0 = tile * 32, col0 = cg * kBN;
  const int od = a.od, Ke = a.od + a.ad, Kpe = r16(Ke), LDX = Kpe + 4, H = a.H;
  const int prob = wave >> 1, rb = wave & 1;
  const float* __restrict__ PF = prob == 0 ? a.e0f : a.d0f;
  const int nk = prob == 0 ? Kpe >> 4 : (od + 15) >> 4;
  f32x4 bf[NKW][5];
#pragma unroll
  for (int j = 0; j < NKW; ++j) {
    const int ks = j < nk ? j : 0;
#pragma unroll
    for (int c = 0; c < 5; ++c)
      bf[j][c] = *reinterpret_cast<const f32x4*>(PF + ((size_t)(ks * 4 + kq) * H + col0 + c * 16 + m) * 4);
  }
  const int Kd = od + a.L;
  // the [32 x Kpe] input tile: each wave stages 8 rows, lane = (row, eighth), element (row, 8 cc + eighth) for cc <
  // Kpe / 8 -- every load of the tile is requested before the first is used, and nothing divides by a run-time width
  {
    const int r = wave * 8 + (lane >> 3), q = lane & 7;
    const int gr = row0 + r, grc = gr < a.rows ? gr : a.rows - 1;
    const float* __restrict__ orow = a.obs + (size_t)grc * od;
    const float* __restrict__ arow = a.act + (size_t)grc * a.ad;
    float v[NKW * 2];
#pragma unroll
    for (int cc = 0; cc < NKW * 2; ++cc) {
      const int c = 8 * cc + q;
      const float* __restrict__ src = c < od ? orow + c : arow + (c < Ke ? c - od : 0);
      v[cc] = *src;
    }
#pragma unroll
    for (int cc = 0; cc < NKW * 2; ++cc) {
      const int c = 8 * cc + q;
      if (c < Kpe) {
        const float x = (gr < a.rows && c < Ke) ? v[cc] : 0.f;
        lds[r * LDX + c] = x;
        if (cg == 0 && gr < a.rows && c < Ke) {
          a.enc_x[(size_t)gr * Ke + c] = x;
          if (c < od) a.dec_x[(size_t)gr * Kd + c] = x;
        }
      }
    }
  }
  __syncthreads();
  f32x4 acc[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NKW; ++j) {
    if (j < nk) {
      f32x4 af = *reinterpret_cast<const f32x4*>(lds + (rb * 16 + m) * LDX + j * 16 + 4 * kq);
      if (prob == 1) {  // the decoder's share stops at the observation columns (the tile holds the action behind them)
#pragma unroll
        for (int t = 0; t < 4; ++t) af[t] = (j * 16 + 4 * kq + t) < od ? af[t] : 0.f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) mfma(acc[c], af[t], bf[j][c][t]);
    }
  }
  float* __restrict__ out = prob == 0 ? a.enc_h0 : a.P;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int col = col0 + c * 16 + m;
    const float bv = prob == 0 ? a.eb0[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = row0 + rb * 16 + kq * 4 + r;
      float v = acc[c][r] + bv;
      if (prob == 0) v = fmaxf(v, 0.f);
      if (gr < a.rows) out[(size_t)gr * H + col] = v;
    }
  }
}

// ---- shared tail of the wide launches: the 4 partial tiles -> LDS -> sum ------------------------------------------------
__device__ __forceinline__ void park_partials(float* lds, int wave, int m, int kq, const f32x4 (&acc)[3][5]) {
  float* pw = lds + (size_t)wave * kBM * kLD;
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(rb * 16 + kq * 4 + r) * kLD + c * 16 + m] = acc[rb][c][r];
}
__device__ __forceinline__ f32x4 sum_partials(const float* lds, int r, int c4) {
  f32x4 s = *reinterpret_cast<const f32x4*>(lds + r * kLD + 4 * c4);
#pragma unroll
  for (int w = 1; w < kKS; ++w) s += *reinterpret_cast<const f32x4*>(lds + (size_t)w * kBM * kLD + r * kLD + 4 * c4);
  return s;
}
// [48 x 80] tile in LDS (partial buffer 0) x a narrow weight block: waves 0-2 take one 16-row block each; NB 16-column
// output blocks; result rows go to slab[cg]
template <int NB>
__device__ __forceinline__ void slab_product(const float* lds, int wave, int m, int kq, const f32x4 (&wb)[5][NB],
                                             float* __restrict__ slab, int cg, int row0, int rows) {
  if (wave >= 3) return;
  f32x4 acc[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 5; ++ks) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(lds + (wave * 16 + m) * kLD + ks * 16 + 4 * kq);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NB; ++c) mfma(acc[c], a4[t], wb[ks][c][t]);
  }
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = row0 + wave * 16 + kq * 4 + r;
      if (gr < rows) slab[((size_t)cg * rows + gr) * kSW + c * 16 + m] = acc[c][r];
    }
}

// ---- encoder, wide layer: h1 = relu(h0 W1^T + b1); head slabs --------------------------------------------------------
template <int NHB, class AR>
__device__ __forceinline__ void fwd_enc_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  NS_PRIO();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  int tile, cg;
  if (!tile_of(a.row_tiles, a.col_groups, &tile, &cg)) return;
  const int row0 = tile * kBM, col0 = cg * kBN, H = a.H;
  int ks0, cnt;
  k_range(H >> 4, wave, &ks0, &cnt);
  const float* __restrict__ A = a.enc_h0;
  const float* __restrict__ W = a.e1f;
  int arow[3];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb) {
    const int r = row0 + rb * 16 + m;
    arow[rb] = r < a.rows ? r : a.rows - 1;
  }
  f32x4 af[4][3], bf[4][5];
  auto issue = [&](int j, int slot) __attribute__((always_inline)) {
    const int ks = ks0 + (j < cnt ? j : cnt - 1);
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
      af[slot][rb] = *reinterpret_cast<const f32x4*>(A + (size_t)arow[rb] * H + ks * 16 + kq * 4);
#pragma unroll
    for (int c = 0; c < 5; ++c)
      bf[slot][c] = *reinterpret_cast<const f32x4*>(W + ((size_t)(ks * 4 + kq) * H + col0 + c * 16 + m) * 4);
  };
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  // the head's weight columns for this tile's 80 k: e2f = PF[k/4][n][k%4], n < round16(2 L)
  f32x4 hb[5][NHB];
  const int Nh = r16(2 * a.L);
#pragma unroll
  for (int ks = 0; ks < 5; ++ks)
#pragma unroll
    for (int c = 0; c < NHB; ++c)
      hb[ks][c] = *reinterpret_cast<const f32x4*>(a.e2f + ((size_t)((col0 >> 2) + ks * 4 + kq) * Nh + c * 16 + m) * 4);
  f32x4 acc[3][5];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < kNKW; ++j) {
    if (j + 3 < kNKW) issue(j + 3, (j + 3) & 3);
    __builtin_amdgcn_sched_barrier(0);
    if (j < cnt) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int c = 0; c < 5; ++c) mfma(acc[rb][c], af[j & 3][rb][t], bf[j & 3][c][t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  park_partials(lds, wave, m, kq, acc);
  __syncthreads();
  for (int idx = tid; idx < kBM * 20; idx += 256) {
    const int r = idx / 20, c4 = idx - r * 20;
    f32x4 s = sum_partials(lds, r, c4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.eb1 + col0 + 4 * c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = fmaxf(s[j] + bv[j], 0.f);
    const int gr = row0 + r;
    if (gr < a.rows) *reinterpret_cast<f32x4*>(a.enc_h1 + (size_t)gr * H + col0 + 4 * c4) = s;
    *reinterpret_cast<f32x4*>(lds + r * kLD + 4 * c4) = s;  // (in place: this thread read exactly these words of buffer 0)
  }
  __syncthreads();
  slab_product<NHB>(lds, wave, m, kq, hb, a.slabE, cg, row0, a.rows);
}

// ---- the three launches whose A operand is GENERATED from the previous launch's slabs ---------------------------------
enum { MODE_DEC_FWD = 0, MODE_DEC_BWD = 1, MODE_ENC_BWD = 2 };

template <int MODE, int NKS, class AR>
__device__ __forceinline__ void gen_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  NS_PRIO();
  __shared__ float s_red[2][kKS];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  int tile, cg;
  if (!tile_of(a.row_tiles, a.col_groups, &tile, &cg)) return;
  const int row0 = tile * kBM, col0 = cg * kBN, H = a.H, rows = a.rows, L = a.L, od = a.od, ad = a.ad;
  const int n_cg = a.col_groups;
  const int cg_inv = 65536 / n_cg + 1;  // (r * cg_inv) >> 16 == r / n_cg for r < 48, n_cg <= 5
  int ks0, cnt;
  k_range(H >> 4, wave, &ks0, &cnt);
  NS_STAMP(0);
  float* S = lds;                                             // [48][kSL] row tile of the prologue
  float* priv = lds + kBM * kSL + (size_t)wave * kBM * kPL;   // [48][kPL] this wave's A columns
  // main-product weights: forward pack of layer 1 (row stride H) / backward pack (row stride H + 16)
  const float* __restrict__ W = MODE == MODE_DEC_FWD ? a.d1f : (MODE == MODE_DEC_BWD ? a.d1b : a.e1b);
  const int Wld = MODE == MODE_DEC_FWD ? H : H + 16;
  // small product: S [48 x 16 NKS] x these weights' columns [ (ks_lo + ks) * 4 + kq ][ n ]
  const float* __restrict__ Ws = MODE == MODE_DEC_FWD ? a.d0f : (MODE == MODE_DEC_BWD ? a.d2b : a.e2b);
  const int Wsld = MODE == MODE_DEC_FWD ? H : H + 16;
  const int ks_lo = MODE == MODE_DEC_FWD ? od >> 4 : 0;
  // element-wise operand of the fix-up (row-major [rows, H]) and where the generated rows go
  const float* __restrict__ aux = MODE == MODE_DEC_FWD ? a.P : (MODE == MODE_DEC_BWD ? a.dec_h1 : a.enc_h1);
  float* __restrict__ gen_out = MODE == MODE_DEC_FWD ? a.dec_h0 : (MODE == MODE_DEC_BWD ? a.dec_dz1 : a.enc_dz1);

  f32x4 bf[4][5];
  auto issue = [&](int j, int slot) __attribute__((always_inline)) {
    const int ks = ks0 + (j < cnt ? j : cnt - 1);
#pragma unroll
    for (int c = 0; c < 5; ++c)
      bf[slot][c] = *reinterpret_cast<const f32x4*>(W + ((size_t)(ks * 4 + kq) * Wld + col0 + c * 16 + m) * 4);
  };
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  const int ch0 = cnt < 4 ? cnt : 4, ch1 = cnt - ch0;  // k-steps of the two halves of this wave's range
  f32x4 bs[NKS][4];
  auto issue_small = [&](int jh0, int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = jh0 + (jj < ch ? jj : 0);
        bs[ks][jj] = *reinterpret_cast<const f32x4*>(Ws + ((size_t)((ks_lo + ks) * 4 + kq) * Wsld + (ks0 + j) * 16 + m) * 4);
      }
  };
  f32x4 ax[12];
  f32x4 bvh = f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue_aux = [&](int jh0, int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int idx = lane + 64 * i, r = idx >> 4, c4 = idx & 15;
      const int gr = row0 + r, grc = gr < rows ? gr : rows - 1;
      const int col = (ks0 + jh0) * 16 + (4 * c4 < 16 * ch ? 4 * c4 : 0);
      ax[i] = *reinterpret_cast<const f32x4*>(aux + (size_t)grc * H + col);
    }
    // (the fix-up's bias columns depend on the lane alone: requested here, not in the fix-up -- a load there is a round trip
    // in front of the half's k-steps, 0.8 us per half in the phase stamps of tools/vae_ns_lab.hip)
    if (MODE == MODE_DEC_FWD)
      bvh = *reinterpret_cast<const f32x4*>(a.db0 + (ks0 + jh0) * 16 + (4 * (lane & 15) < 16 * ch ? 4 * (lane & 15) : 0));
  };
  issue_small(0, ch0);
  issue_aux(0, ch0);
  NS_STAMP(1);
  // ---- prologue: the 48 rows' slabs -> row-local results -> S
  constexpr int SWD = 16 * NKS;
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int it = 0; it < kBM * SWD / 256; ++it) {  // (48 * 16 NKS / 256 = 3 NKS passes, all loads of all passes in flight)
    const int idx = tid + 256 * it;
    const int r = idx / SWD, kk = idx - r * SWD;
    const int gr = row0 + r, grc = gr < rows ? gr : rows - 1;
    const bool live = gr < rows, w0 = live && cg == 0;
    float sv = 0.f;
    if (MODE == MODE_DEC_FWD) {
      const int j = 16 * ks_lo + kk - od;
      if (j >= 0 && j < L) {  // z = mean + exp(clamp(log_std)) * eps  (net.py:319-331; == osrl_vae_latent)
        float mean = a.eb2[j], ls = a.eb2[L + j];
        float pm[5], pl[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) {  // (H <= 448: at most 5 column groups; absent ones re-read group 0)
          const float* __restrict__ sl = a.slabE + ((size_t)(g < n_cg ? g : 0) * rows + grc) * kSW;
          pm[g] = sl[j];
          pl[g] = sl[L + j];
        }
#pragma unroll
        for (int g = 0; g < 5; ++g) {
          mean += g < n_cg ? pm[g] : 0.f;
          ls += g < n_cg ? pl[g] : 0.f;
        }
        // (explicit fma: the by-value kernel and its descriptor-in-memory twin must take the SAME contraction decisions
        // -- graph replays and eager steps are compared bit for bit)
        const float zv = __builtin_fmaf(expf(fminf(fmaxf(ls, kLsMin), kLsMax)), a.eps[(size_t)grc * L + j], mean);
        sv = zv;
        if (w0) {
          a.enc_head[(size_t)gr * 2 * L + j] = mean;
          a.enc_head[(size_t)gr * 2 * L + L + j] = ls;
          a.z[(size_t)gr * L + j] = zv;
          a.dec_x[(size_t)gr * (od + L) + od + j] = zv;
        }
      }
    } else if (MODE == MODE_DEC_BWD) {
      if (kk < ad) {  // u = max_action tanh(.), dY of the MSE, dZ2 = dY max_action (1 - tanh^2)  (== OSRL_SEED_MSE)
        float up = a.db2[kk];
        float pu[5];
#pragma unroll
        for (int g = 0; g < 5; ++g) pu[g] = a.slabD[((size_t)(g < n_cg ? g : 0) * rows + grc) * kSW + kk];
#pragma unroll
        for (int g = 0; g < 5; ++g) up += g < n_cg ? pu[g] : 0.f;
        const float y = a.max_action * tanhf(up);
        const float d = y - a.act[(size_t)grc * ad + kk];
        const float tt = y * (1.0f / a.max_action);
        sv = 2.0f * d * (a.inv_rows / (float)ad) * a.max_action * __builtin_fmaf(-tt, tt, 1.0f);
        if (w0) {
          l0 = __builtin_fmaf(d, d, l0);
          a.dec_u[(size_t)gr * ad + kk] = y;
          a.dec_dz2[(size_t)gr * ad + kk] = sv;
        }
      }
      if (kk < L && w0) {  // the KL term of the logged loss (cpq.py:128)
        const float mean = a.enc_head[(size_t)gr * 2 * L + kk];
        const float sd = expf(fminf(fmaxf(a.enc_head[(size_t)gr * 2 * L + L + kk], kLsMin), kLsMax));
        l1 += -0.5f * (__builtin_fmaf(-sd, sd, __builtin_fmaf(-mean, mean, 1.0f + logf(sd * sd))));
      }
    } else {
      if (kk < 2 * L) {  // d(recon + beta KL)/d(mean | log_std) through z = mean + sd eps  (== osrl_vae_latent_bwd)
        const int k = kk < L ? kk : kk - L;
        float g = 0.f;
        float pg[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) pg[q] = a.slabX[((size_t)(q < n_cg ? q : 0) * rows + grc) * kSW + k];
#pragma unroll
        for (int q = 0; q < 5; ++q) g += q < n_cg ? pg[q] : 0.f;
        const float mean = a.enc_head[(size_t)grc * 2 * L + k];
        const float lsr = a.enc_head[(size_t)grc * 2 * L + L + k];
        const float ev = a.eps[(size_t)grc * L + k];
        const float sd = expf(fminf(fmaxf(lsr, kLsMin), kLsMax));
        const float c = a.beta * a.inv_rows / (float)L;
        const bool inside = lsr >= kLsMin && lsr <= kLsMax;
        const float t_kl = c * (sd - 1.0f / sd);
        sv = kk < L ? __builtin_fmaf(c, mean, g) : (inside ? __builtin_fmaf(g, ev, t_kl) * sd : 0.f);
        if (w0) a.enc_dz2[(size_t)gr * 2 * L + kk] = sv;
      }
    }
    S[r * kSL + kk] = live ? sv : 0.f;
  }
  if (MODE == MODE_DEC_BWD) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      l0 += __shfl_xor(l0, o);
      l1 += __shfl_xor(l1, o);
    }
    if (lane == 0) {
      s_red[0][wave] = l0;
      s_red[1][wave] = l1;
    }
  }
  __syncthreads();
  if (MODE == MODE_DEC_BWD && cg == 0 && tid == 0) {  // this tile's partials of the logged loss, waves in order
    float t0 = 0.f, t1 = 0.f;
    for (int w = 0; w < kKS; ++w) {
      t0 += s_red[0][w];
      t1 += s_red[1][w];
    }
    coh_put(a.partials + 2 * tile, t0);
    coh_put(a.partials + 2 * tile + 1, t1);
  }
  f32x4 acc[3][5];
#pragma unroll
  for (int rb = 0; rb < 3; ++rb)
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // epilogue operands: requested under the SECOND half's MFMAs (not up front: the launch has to fit beside a 197-register
  // wave of the N*B-row launch, i.e. in 312 registers, and the first half still holds its fix-up operands)
  f32x4 wb[5][1];  // DEC_FWD: the output head's columns (d2f, n < 16); DEC_BWD: layer 0's latent columns (d0b, i = od + m)
  f32x4 hv[4];     // backward: h0 of this thread's epilogue elements (relu')
  auto issue_epilogue = [&]() __attribute__((always_inline)) {
    if (MODE == MODE_DEC_FWD) {
      const int Nh = r16(ad);
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
        wb[ks][0] = *reinterpret_cast<const f32x4*>(a.d2f + ((size_t)((col0 >> 2) + ks * 4 + kq) * Nh + m) * 4);
    } else if (MODE == MODE_DEC_BWD) {
      const int Kb0 = r16(od + L) + 16;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
        wb[ks][0] = *reinterpret_cast<const f32x4*>(a.d0b + ((size_t)((col0 >> 2) + ks * 4 + kq) * Kb0 + od + m) * 4);
    }
    if (MODE != MODE_DEC_FWD) {
      const float* __restrict__ h0 = MODE == MODE_DEC_BWD ? a.dec_h0 : a.enc_h0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx < kBM * 20 ? idx / 20 : 0, c4 = idx < kBM * 20 ? idx - r * 20 : 0;
        const int gr = row0 + r, grc = gr < rows ? gr : rows - 1;
        hv[i] = *reinterpret_cast<const f32x4*>(h0 + (size_t)grc * H + col0 + 4 * c4);
      }
    }
  };

  // ---- the wave's K range in two halves: generate the A columns of a half in the private region, then run its k-steps
  auto half = [&](const int jh0, const int ch, const int hsel) __attribute__((always_inline)) {
    // small product -> private region (the row tile's fragments are re-read per half: 12-24 registers less in the k-loop)
    f32x4 sf[3][NKS];
#pragma unroll
    for (int rb = 0; rb < 3; ++rb)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
        sf[rb][ks] = *reinterpret_cast<const f32x4*>(S + (rb * 16 + m) * kSL + ks * 16 + 4 * kq);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (jj < ch) {
        f32x4 t3[3];
#pragma unroll
        for (int rb = 0; rb < 3; ++rb) t3[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) mfma(t3[rb], sf[rb][ks][t], bs[ks][jj][t]);
#pragma unroll
        for (int rb = 0; rb < 3; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) priv[(rb * 16 + kq * 4 + r) * kPL + jj * 16 + m] = t3[rb][r];
      }
    }
    NS_STAMP(3 + 3 * hsel);
    // fix-up, element-wise in row-major order; the generated rows leave from here (each column group a share of the rows)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int idx = lane + 64 * i, r = idx >> 4, c4 = idx & 15;
      if (4 * c4 < 16 * ch) {
        const int gk = (ks0 + jh0) * 16 + 4 * c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(priv + r * kPL + 4 * c4);
        if (MODE == MODE_DEC_FWD) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q] + ax[i][q] + bvh[q], 0.f);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ax[i][q] > 0.f ? v[q] : 0.f;
        }
        *reinterpret_cast<f32x4*>(priv + r * kPL + 4 * c4) = v;
        const int gr = row0 + r;
        const int share = r - ((r * cg_inv) >> 16) * n_cg;  // r % n_cg (r < 48)
        if (gr < rows && share == cg) *reinterpret_cast<f32x4*>(gen_out + (size_t)gr * H + gk) = v;
      }
    }
    if (hsel == 0 && ch1 > 0) {  // the second half's operands, requested under this half's MFMAs
      issue_small(ch0, ch1);
      issue_aux(ch0, ch1);
    }
    if (hsel == 1) issue_epilogue();
    NS_STAMP(4 + 3 * hsel);
    // main k-steps of this half
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = jh0 + jj;  // (jh0 is 0 or 4: the ring slot is static)
      if (hsel == 0 ? true : jj < 3) {
        if (j + 3 < kNKW) issue(j + 3, (j + 3) & 3);
        __builtin_amdgcn_sched_barrier(0);
        if (jj < ch) {
          f32x4 af[3];
#pragma unroll
          for (int rb = 0; rb < 3; ++rb)
            af[rb] = *reinterpret_cast<const f32x4*>(priv + (rb * 16 + m) * kPL + jj * 16 + 4 * kq);
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int rb = 0; rb < 3; ++rb)
#pragma unroll
              for (int c = 0; c < 5; ++c) mfma(acc[rb][c], af[rb][t], bf[j & 3][c][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  NS_STAMP(2);
  half(0, ch0, 0);
  NS_STAMP(5);
  half(4, ch1, 1);
  NS_STAMP(8);

  // ---- the four partial tiles -> sum -> epilogue
  __syncthreads();  // every wave is through with S and its private region: the partial buffers alias them
  park_partials(lds, wave, m, kq, acc);
  __syncthreads();
  NS_STAMP(9);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    if (idx < kBM * 20) {
      const int r = idx / 20, c4 = idx - r * 20;
      f32x4 s = sum_partials(lds, r, c4);
      const int gr = row0 + r;
      if (MODE == MODE_DEC_FWD) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.db1 + col0 + 4 * c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = fmaxf(s[q] + bv[q], 0.f);
        if (gr < rows) *reinterpret_cast<f32x4*>(a.dec_h1 + (size_t)gr * H + col0 + 4 * c4) = s;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = hv[i][q] > 0.f ? s[q] : 0.f;
        float* __restrict__ dz0 = MODE == MODE_DEC_BWD ? a.dec_dz0 : a.enc_dz0;
        if (gr < rows) *reinterpret_cast<f32x4*>(dz0 + (size_t)gr * H + col0 + 4 * c4) = s;
      }
      if (MODE != MODE_ENC_BWD) *reinterpret_cast<f32x4*>(lds + r * kLD + 4 * c4) = s;
    }
  }
  NS_STAMP(10);
  if (MODE != MODE_ENC_BWD) {
    __syncthreads();
    slab_product<1>(lds, wave, m, kq, wb, MODE == MODE_DEC_FWD ? a.slabD : a.slabX, cg, row0, rows);
  }
  NS_STAMP(11);
  if (MODE == MODE_DEC_BWD && cg == 0) {
    // the logged loss: the last first-column-group workgroup to get here sums every tile's partials in tile order
    // (wait-free: a workgroup is the last one or leaves; protocol of mlp.hip's seeded backward)
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned seen = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = seen == (unsigned)a.row_tiles - 1;
      if (last) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = last;
    }
    __syncthreads();
    if (s_last && wave == 0) {
      const int per = (a.row_tiles + 63) / 64;
      float t0 = 0.f, t1 = 0.f;
      for (int i = lane * per; i < (lane + 1) * per && i < a.row_tiles; ++i) {
        t0 += coh_get(a.partials + 2 * i);
        t1 += coh_get(a.partials + 2 * i + 1);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        t0 += __shfl_xor(t0, o);
        t1 += __shfl_xor(t1, o);
      }
      if (lane == 0 && a.stat)
        a.stat[0] = __builtin_fmaf(a.beta, t1 * (a.inv_rows / (float)L), t0 * (a.inv_rows / (float)ad));
    }
  }
}

// by-value kernels and their twins that read the descriptor from device memory (csrc/argmem.h: inside a captured step the
// descriptor lives in the step's argument arena -- where the runtime keeps kernel arguments in host memory every wave
// would otherwise fetch its part over PCIe, in several dependent round trips)
template <int NKW>
__global__ __launch_bounds__(256) void vae_ns_l0_kernel(const NsArgs a) { l0_body<NKW, const NsArgs&>(a); }
template <int NKW>
__global__ __launch_bounds__(256) void vae_ns_l0_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(2, p);
  l0_body<NKW, const OSRL_CAS NsArgs&>(*(const OSRL_CAS NsArgs*)p);
}
template <int NHB>
__global__ __launch_bounds__(256) void vae_ns_fwd_enc_kernel(const NsArgs a) { fwd_enc_body<NHB, const NsArgs&>(a); }
template <int NHB>
__global__ __launch_bounds__(256) void vae_ns_fwd_enc_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(3, p);
  fwd_enc_body<NHB, const OSRL_CAS NsArgs&>(*(const OSRL_CAS NsArgs*)p);
}
template <int MODE, int NKS>
__global__ __launch_bounds__(256) void vae_ns_gen_kernel(const NsArgs a) { gen_body<MODE, NKS, const NsArgs&>(a); }
template <int MODE, int NKS>
__global__ __launch_bounds__(256) void vae_ns_gen_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(40 + MODE, p);
  gen_body<MODE, NKS, const OSRL_CAS NsArgs&>(*(const OSRL_CAS NsArgs*)p);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool shape_ok(const osrl_vae_ns_t* v) {
  if (!v || !v->enc || !v->dec || v->rows < 1) return false;
  const osrl_mlp_t *e = v->enc, *d = v->dec;
  if (e->n_layers != 3 || d->n_layers != 3 || e->n_nets != 1 || d->n_nets != 1) return false;
  const int H = e->dims[1], od = v->od, ad = v->ad, L = v->L;
  if (H < 80 || H > 448 || H % 80 || e->dims[2] != H || d->dims[1] != H || d->dims[2] != H) return false;
  if (od < 1 || ad < 1 || L < 1 || ad > 16 || L > 16 || od + ad > 128 || od + L > 128) return false;
  if (e->dims[0] != od + ad || e->dims[3] != 2 * L || d->dims[0] != od + L || d->dims[3] != ad) return false;
  if (e->acts[0] != OSRL_ACT_RELU || e->acts[1] != OSRL_ACT_RELU || e->acts[2] != OSRL_ACT_ID) return false;
  if (d->acts[0] != OSRL_ACT_RELU || d->acts[1] != OSRL_ACT_RELU || d->acts[2] != OSRL_ACT_TANH) return false;
  if (e->out_scale != 1.0f || !(d->out_scale > 0.f)) return false;
  return true;
}

bool fill(const osrl_vae_ns_t* v, NsArgs* a, bool backward) {
  if (!shape_ok(v) || !v->obs || !v->act || !v->eps || !v->z || !v->P || !v->slabs) return false;
  const osrl_mlp_t *e = v->enc, *d = v->dec;
  a->rows = v->rows; a->od = v->od; a->ad = v->ad; a->L = v->L; a->H = e->dims[1];
  a->row_tiles = (v->rows + kBM - 1) / kBM;
  a->col_groups = a->H / kBN;
  a->max_action = d->out_scale;
  a->beta = v->beta;
  a->inv_rows = 1.0f / (float)(v->rows_global > 0 ? v->rows_global : v->rows);
  a->obs = v->obs; a->act = v->act; a->eps = v->eps;
  a->e0f = e->Wf[0][0]; a->e1f = e->Wf[0][1]; a->e2f = e->Wf[0][2]; a->e1b = e->Wb[0][1]; a->e2b = e->Wb[0][2];
  a->eb0 = e->b[0][0]; a->eb1 = e->b[0][1]; a->eb2 = e->b[0][2];
  a->d0f = d->Wf[0][0]; a->d1f = d->Wf[0][1]; a->d2f = d->Wf[0][2];
  a->d0b = d->Wb[0][0]; a->d1b = d->Wb[0][1]; a->d2b = d->Wb[0][2];
  a->db0 = d->b[0][0]; a->db1 = d->b[0][1]; a->db2 = d->b[0][2];
  a->enc_x = v->enc_acts.x; a->enc_h0 = v->enc_acts.h[0][0]; a->enc_h1 = v->enc_acts.h[0][1]; a->enc_head = v->enc_acts.h[0][2];
  a->z = v->z;
  a->dec_x = v->dec_acts.x; a->dec_h0 = v->dec_acts.h[0][0]; a->dec_h1 = v->dec_acts.h[0][1]; a->dec_u = v->dec_acts.h[0][2];
  a->enc_dz0 = v->enc_g.dz[0][0]; a->enc_dz1 = v->enc_g.dz[0][1]; a->enc_dz2 = v->enc_g.dz[0][2];
  a->dec_dz0 = v->dec_g.dz[0][0]; a->dec_dz1 = v->dec_g.dz[0][1]; a->dec_dz2 = v->dec_g.dz[0][2];
  a->P = v->P;
  const size_t slab = (size_t)a->col_groups * v->rows * kSW;
  a->slabE = v->slabs; a->slabD = v->slabs + slab; a->slabX = v->slabs + 2 * slab;
  a->partials = v->partials; a->counter = v->counter; a->stat = v->stat;
  if (!a->e0f || !a->e1f || !a->e2f || !a->eb0 || !a->eb1 || !a->eb2 || !a->d0f || !a->d1f || !a->d2f || !a->db0 ||
      !a->db1 || !a->db2)
    return false;
  if (!a->enc_x || !a->enc_h0 || !a->enc_h1 || !a->enc_head || !a->dec_x || !a->dec_h0 || !a->dec_h1 || !a->dec_u)
    return false;
  if (backward) {
    if (!a->e1b || !a->e2b || !a->d0b || !a->d1b || !a->d2b) return false;
    if (!a->enc_dz0 || !a->enc_dz1 || !a->enc_dz2 || !a->dec_dz0 || !a->dec_dz1 || !a->dec_dz2) return false;
    if (!a->partials || !a->counter) return false;
  }
  return true;
}

constexpr size_t kWideLds = sizeof(float) * kKS * kBM * kLD;  // 64512 B >= S + 4 private regions (59136 B)
static_assert(sizeof(float) * (kBM * kSL + kKS * kBM * kPL) <= kWideLds, "the partial buffers must cover the A regions");

template <class K, class KP>
int launch(K k, KP kp, int grid, int threads, size_t ldsb, hipStream_t st, const NsArgs& a) {
  const void* dev = osrl_argmem::slot(a);
  hipError_t e = dev ? hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)
                     : hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  if (dev) hipLaunchKernelGGL(kp, dim3(grid), dim3(threads), ldsb, st, dev);
  else hipLaunchKernelGGL(k, dim3(grid), dim3(threads), ldsb, st, a);
  return (int)hipGetLastError();
}
#define NS_LAUNCH(K, ...) launch(K<__VA_ARGS__>, K##_p<__VA_ARGS__>

int wide_grid(const NsArgs& a) { return ((a.row_tiles + 7) / 8) * 8 * a.col_groups; }

}  // namespace

extern "C" int osrl_vae_ns_supported(const osrl_vae_ns_t* v) { return shape_ok(v) ? 1 : 0; }

extern "C" int osrl_vae_ns_forward(const osrl_vae_ns_t* v, void* stream) {
  NsArgs a;
  if (!fill(v, &a, false)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int Kpe = (a.od + a.ad + 15) & ~15;
  const int g0 = ((a.rows + 31) / 32) * a.col_groups;
  const size_t l0 = sizeof(float) * 32 * (Kpe + 4);
  int rc;
  if (Kpe <= 48) rc = NS_LAUNCH(vae_ns_l0_kernel, 3), g0, 256, l0, st, a);
  else if (Kpe <= 80) rc = NS_LAUNCH(vae_ns_l0_kernel, 5), g0, 256, l0, st, a);
  else rc = NS_LAUNCH(vae_ns_l0_kernel, 8), g0, 256, l0, st, a);
  if (rc) return rc;
  rc = 2 * a.L <= 16 ? NS_LAUNCH(vae_ns_fwd_enc_kernel, 1), wide_grid(a), 256, kWideLds, st, a)
                     : NS_LAUNCH(vae_ns_fwd_enc_kernel, 2), wide_grid(a), 256, kWideLds, st, a);
  if (rc) return rc;
  const int nks = (a.od + a.L - 1) / 16 - a.od / 16 + 1;
  return nks == 1 ? NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 1), wide_grid(a), 256, kWideLds, st, a)
                  : NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 2), wide_grid(a), 256, kWideLds, st, a);
}

extern "C" int osrl_vae_ns_backward(const osrl_vae_ns_t* v, void* stream) {
  NsArgs a;
  if (!fill(v, &a, true)) return -1;
  hipStream_t st = (hipStream_t)stream;
  int rc = NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_BWD, 1), wide_grid(a), 256, kWideLds, st, a);
  if (rc) return rc;
  return 2 * a.L <= 16 ? NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 1), wide_grid(a), 256, kWideLds, st, a)
                       : NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 2), wide_grid(a), 256, kWideLds, st, a);
}
