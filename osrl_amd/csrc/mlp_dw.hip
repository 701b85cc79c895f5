// mlp_dw.hip -- the weight-gradient GEMMs of the fused MLP (dW = dZ^T A, db = colsum(dZ)) for gfx950: 64 x 64 tiles x a
// common split count (mlp_dw_kernel), a flat (tile, row split) work list on (16 T)^2 tiles (mlp_dwt_kernel), the same
// with the optimizer step applied by a tile's last split (mlp_dwt_adam_kernel), and one-wave-per-tile for token-matrix
// row counts (mlp_dw_big_kernel).  Split from mlp.hip in round 4 (compile time); design notes at the kernels.
#include "dwt.h"
#include "adam.h"
#include "trace.h"

namespace {

// ---- dW = dZ^T A, db = colsum(dZ) ---------------------------------------------------------------
// One workgroup = one 64x64 tile of one layer's dW for one split of the batch rows (blockIdx.y).  The 4 waves
// split that row range again (the contraction runs over rows), each accumulating a full 64x64 partial in
// registers with the k-slot trick (A^T / B fragments are 4 scalar loads of 64 B-contiguous lanes each);
// the fragments of k-step i+1 are fetched while the 64 MFMAs of step i run (double buffer).  The 4 partials
// are summed through LDS in a fixed order and ONE slab tile is written, coalesced: 4x fewer slab bytes for
// the Adam kernel to re-read than one slab per wave.
constexpr int kDwLd = 65;                                     // LDS row stride of a 64x64 partial
constexpr size_t kDwLds = sizeof(float) * (4 * 64 * kDwLd + 4 * 64);

struct DwFrag {
  f32x4 a[4], b[4];
};

__device__ __forceinline__ void dw_load(DwFrag& f, const float* __restrict__ dz, const float* __restrict__ av,
                                        size_t ldz, size_t lda_g, int r0, int r_end, int o0, int i0, int out, int in,
                                        int nob, int nib, int m, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = r0 + 4 * kq + t;
    const bool rok = r < r_end;
    const size_t rc = (size_t)(rok ? r : r_end - 1);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const int o = o0 + ob * 16 + m;
      const bool ok = rok && ob < nob && o < out;
      const float v = dz[rc * ldz + (o < out ? o : 0)];
      f.a[ob][t] = ok ? v : 0.f;
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
      const int i = i0 + ib * 16 + m;
      const bool ok = rok && ib < nib && i < in;
      const float v = av[rc * lda_g + (i < in ? i : 0)];
      f.b[ib][t] = ok ? v : 0.f;
    }
  }
}

__device__ __forceinline__ void dw_mma(f32x4 (&acc)[4][4], float (&dbacc)[4], const DwFrag& f, int nob, int nib,
                                       bool want_db) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      if (ob < nob) {
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
          if (ib < nib) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ob][t], f.b[ib][t], acc[ob][ib], 0, 0, 0);
      }
    }
  }
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) dbacc[ob] += (f.a[ob][0] + f.a[ob][1]) + (f.a[ob][2] + f.a[ob][3]);
  }
}

__global__ __launch_bounds__(256) void mlp_dw_kernel(const osrl_dw_entry_t* __restrict__ entries,
                                                      const int32_t* __restrict__ items, int n_items, int rows,
                                                      int rows_per_split, float* __restrict__ slabs,
                                                      int64_t slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][64][kDwLd] partials + [4][64] bias partials
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int item = blockIdx.x;
#if OSRL_CHAIN_PRIO > 0
  if (rows <= OSRL_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2];
  const osrl_dw_entry_t E = entries[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * 64, i0 = it * 64;
  const int s = blockIdx.y;
  const int rpw = rows_per_split >> 2;  // rows per wave (host keeps rows_per_split a multiple of 64)
  const int r_begin = s * rows_per_split + wave * rpw;
  int r_end = r_begin + rpw;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  int nob = (out - o0 + 15) >> 4;
  nob = nob > 4 ? 4 : nob;
  int nib = (in - i0 + 15) >> 4;
  nib = nib > 4 ? 4 : nib;
  const bool want_db = it == 0;

  f32x4 acc[4][4];
  zero_acc<4, 4>(acc);
  float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ dz = E.dz;
  const float* __restrict__ av = E.a;
  if (r_begin < r_end) {
    DwFrag f0, f1;
    dw_load(f0, dz, av, ldz, lda_g, r_begin, r_end, o0, i0, out, in, nob, nib, m, kq);
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
      if (r0 + 16 < r_end) dw_load(f1, dz, av, ldz, lda_g, r0 + 16, r_end, o0, i0, out, in, nob, nib, m, kq);
      dw_mma(acc, dbacc, f0, nob, nib, want_db);
      if (r0 + 16 < r_end) {
        if (r0 + 32 < r_end) dw_load(f0, dz, av, ldz, lda_g, r0 + 32, r_end, o0, i0, out, in, nob, nib, m, kq);
        dw_mma(acc, dbacc, f1, nob, nib, want_db);
      }
    }
  }
  // ---- 4 partials -> LDS -> fixed-order sum -> one coalesced slab tile
  float* mine = red + wave * 64 * kDwLd;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(ob * 16 + kq * 4 + r) * kDwLd + ib * 16 + m] = acc[ob][ib][r];
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) red[4 * 64 * kDwLd + wave * 64 + ob * 16 + m] = v;
    }
  }
  __syncthreads();
  float* __restrict__ slab = slabs + (size_t)s * slab_stride;
  const int il = tid & 63;
#pragma unroll 4
  for (int ol = tid >> 6; ol < 64; ol += 4) {
    const int off = ol * kDwLd + il;
    const float v = ((red[off] + red[64 * kDwLd + off]) + red[2 * 64 * kDwLd + off]) + red[3 * 64 * kDwLd + off];
    const int o = o0 + ol, i = i0 + il;
    if (o < out && i < in) slab[E.w_off + (size_t)o * in + i] = v;
  }
  if (want_db && tid < 64) {
    const float* db = red + 4 * 64 * kDwLd;
    const float v = ((db[tid] + db[64 + tid]) + db[128 + tid]) + db[192 + tid];
    if (o0 + tid < out) slab[E.b_off + o0 + tid] = v;
  }
}


template <int T>
__global__ __launch_bounds__(256, 1) void mlp_dwt_kernel(const osrl_dw_entry_t* __restrict__ entries,
                                                         const int32_t* __restrict__ items, int rows,
                                                         float* __restrict__ slabs, int64_t slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][16T][16T+1] partials + [4][16T] bias partials
  constexpr int TW = 16 * T, LD = TW + 1;
  const int tid = threadIdx.x;
  OSRL_TRACE_BEGIN(100 + T, entries);
#if OSRL_CHAIN_PRIO > 0
  if (rows <= OSRL_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
  dwt_tile<T>(entries, items, blockIdx.x, rows, red,
              [&](const OSRL_CAS osrl_dw_entry_t& E, const int o0, const int i0, const int sp, const bool want_db) {
    const int out = E.out, in = E.in;
    float* __restrict__ slab = slabs + (size_t)(sp & 0xffff) * slab_stride;
    for (int idx = tid; idx < TW * TW; idx += 256) {
      const int ol = idx / TW, il = idx - ol * TW;
      const int off = ol * LD + il;
      const float v = ((red[off] + red[TW * LD + off]) + red[2 * TW * LD + off]) + red[3 * TW * LD + off];
      const int o = o0 + ol, i = i0 + il;
      if (o < out && i < in) slab[E.w_off + (size_t)o * in + i] = v;
    }
    if (want_db && tid < TW) {
      const float* db = red + 4 * TW * LD;
      const float v = ((db[tid] + db[TW + tid]) + db[2 * TW + tid]) + db[3 * TW + tid];
      if (o0 + tid < out) slab[E.b_off + o0 + tid] = v;
    }
  });
}




// ---- dW + the optimizer step of its group in ONE launch (osrl_mlp_backward_dw_tiles_adam) ------------------------
// The (tile, row split) workgroups of mlp_dwt_kernel, and the LAST split of a tile to finish applies Adam (+ Polyak +
// the packed-copy refresh) to that tile's parameters right there: every split stores its slab tile (device-coherent
// stores), waits for their acknowledgement and signs in at the tile's arrival counter; the workgroup that finds all
// other splits signed in reads the tile's slabs back (device-coherent loads), sums them IN SLAB ORDER -- the order
// of optim.hip's adam_body -- and runs osrl_adam::update1 on the sum, i.e. parameters, moments, targets and packed
// copies get the same bits as from osrl_mlp_backward_dw_tiles + osrl_adam_step_packed.  What it removes from a train
// step: one launch per optimizer group (four on the CPQ step's chains, 6.6-10 us each plus what they lose beside an
// N*B-row launch), the slab re-read by a second grid, and the full-group pass over padding.  A tile with ONE split
// needs no counter and no slab: its gradient is complete in LDS (the BC one-launch step's epilogue).
// The wait-free form matters: nobody spins -- a workgroup either is the last one or leaves -- so the launch makes no
// assumption about co-residency or dispatch order (cf. mlp_step_kernel's bounded poll).
struct DwAdamArgs {
  const osrl_dw_entry_t* entries;
  const int32_t* items;
  const int32_t* tile_ids;  // [n_work] arrival counter of each item's tile
  uint32_t* counters;       // [n_tiles] zero before the first launch; the last arriver of a tile re-arms it
  float* slabs;
  int64_t slab_stride;
  float *p, *m, *v, *tgt;
  const int32_t *map_f, *map_b;
  float *pf, *pb, *tf;
  const osrl_step_state_t* st;
  float lr, b1, b2, eps, tau;
  int32_t rows, pad_;
};

// device-coherent slab words (see the exchange in mlp_dwt_adam_body)
__device__ __forceinline__ void slab_put(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float slab_get(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int T, class AR>
__device__ __forceinline__ void mlp_dwt_adam_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][16T][16T+1] partials + [4][16T] bias partials
  __shared__ int s_last;
  constexpr int TW = 16 * T, LD = TW + 1;
  constexpr int PER = (TW * TW + 255) / 256;  // elements of the tile per lane (T = 5: 25, 4: 16, 3: 9, 2: 4)
  const int tid = threadIdx.x;
  const int rows = a.rows;
#if OSRL_CHAIN_PRIO > 0
  if (rows <= OSRL_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
#endif
  const int item = blockIdx.x;
  dwt_tile<T>(a.entries, a.items, item, rows, red,
              [&](const OSRL_CAS osrl_dw_entry_t& E, const int o0, const int i0, const int sp, const bool want_db) {
    const int out = E.out, in = E.in;
    const int s = sp & 0xffff, nsp = sp >> 16;
    float* __restrict__ slabs = a.slabs;
    const int64_t stride = a.slab_stride;
    const int64_t w_off = E.w_off, b_off = E.b_off;
    float* __restrict__ P = a.p;
    float* __restrict__ M = a.m;
    float* __restrict__ V = a.v;
    float* TG = a.tgt;
    const int32_t* MF = a.map_f;
    const int32_t* MB = a.map_b;
    const float* TGr = TG ? TG : P;  // absent target / maps re-read p: no branch around a load
    const int32_t* MFr = MF ? MF : reinterpret_cast<const int32_t*>(P);
    const int32_t* MBr = MB ? MB : reinterpret_cast<const int32_t*>(P);
    // ---- (every split) the optimizer state of this tile's elements is REQUESTED before the slab tile is stored and the
    // arrival is counted: it does not depend on the other splits, and its round trip then runs under the store + release
    // + atomic sequence instead of behind the acquire, where only the slabs remain to be fetched.  (The splits that turn
    // out not to be the last one waste these loads: 24 B per element and split, L2 hits after the first.)
    unsigned w[PER];
    bool ok[PER];
    int offl[PER];
    float pv[PER], mv[PER], vv[PER], tv0[PER];
    int mf[PER], mb[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int idx = tid + j * 256;
      const int ol = idx / TW, il = idx - ol * TW;
      const int o = o0 + ol, i = i0 + il;
      ok[j] = idx < TW * TW && o < out && i < in;
      w[j] = (unsigned)(ok[j] ? w_off + (int64_t)o * in + i : w_off);  // (groups are < 2^32 floats)
      offl[j] = ok[j] ? ol * LD + il : 0;
      pv[j] = P[w[j]];
      mv[j] = M[w[j]];
      vv[j] = V[w[j]];
      tv0[j] = TGr[w[j]];
      mf[j] = MFr[w[j]];
      mb[j] = MBr[w[j]];
    }
    const bool has_b = want_db && tid < TW && o0 + tid < out;
    const unsigned wb = (unsigned)(b_off + (has_b ? o0 + tid : 0));
    float pb_ = P[wb], mb_ = M[wb], vb_ = V[wb], tb_ = TGr[wb];
    if (nsp > 1) {
      float* __restrict__ slab = slabs + (size_t)s * stride;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int off = offl[j];
        const float v = ((red[off] + red[TW * LD + off]) + red[2 * TW * LD + off]) + red[3 * TW * LD + off];
        if (ok[j]) slab_put(slab + w[j], v);
      }
      if (has_b) {
        const float* db = red + 4 * TW * LD;
        slab_put(slab + wb, ((db[tid] + db[TW + tid]) + db[2 * TW + tid]) + db[3 * TW + tid]);
      }
      // The slab tile is exchanged between workgroups on different XCDs (each with its own L2) INSIDE the launch.  The
      // textbook form -- release fence (write back the whole L2), count, acquire fence (invalidate the whole L2) -- cost
      // this launch 15-30 us (224-448 workgroups each writing back an L2 that the others are still filling, the last
      // arrivers' invalidates evicting everybody's operand lines).  Instead the slab words themselves are device-coherent
      // accesses (relaxed agent-scope atomics = sc1 stores / loads: written through to, and read from, the level all
      // XCDs share), the arrival is counted once every wave's stores are acknowledged, and no cache is touched.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        unsigned* c = a.counters + a.tile_ids[item];
        const unsigned seen = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = seen == (unsigned)nsp - 1;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every split signed in: re-arm
        s_last = last;
      }
      __syncthreads();
      if (!s_last) return;
    }
    // ---- the optimizer step on this tile (optim.hip adam_body, element for element)
    const osrl_step_state_t* __restrict__ st = a.st;
    const float lr_t = a.lr * st->lr_scale;
    const osrl_adam::Coef c{a.b1, a.b2, a.eps, lr_t / st->bc1, st->bc2_sqrt};
    const float tau = a.tau;
    auto apply = [&](const unsigned wi, const float g, float p_, float m_, float v_, const float t0, const int mfi,
                     const int mbi) {
      osrl_adam::update1(p_, m_, v_, g, c);
      P[wi] = p_;
      M[wi] = m_;
      V[wi] = v_;
      float tv = p_;
      if (TG) {
        tv = osrl_adam::polyak1(tau, p_, t0);
        TG[wi] = tv;
      }
      if (MF && mfi >= 0) {
        a.pf[mfi] = p_;
        if (TG && a.tf) a.tf[mfi] = tv;
      }
      if (MF && MB && mbi >= 0) a.pb[mbi] = p_;
    };
    float g[PER], gb = 0.f;
    if (nsp > 1) {
      // every slab value of the tile requested at once (KS slabs per element in flight, slabs past nsp re-read slab 0 and
      // are dropped by a select): ONE round trip behind the acquire; summed in slab order like adam_body
      auto gather = [&](auto ks_c) {
        constexpr int KS = decltype(ks_c)::value;
        float sl[PER][KS], sb[KS];
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
          for (int q = 0; q < KS; ++q) sl[j][q] = slab_get(slabs + (size_t)(q < nsp ? q : 0) * stride + w[j]);
#pragma unroll
        for (int q = 0; q < KS; ++q) sb[q] = slab_get(slabs + (size_t)(q < nsp ? q : 0) * stride + wb);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          g[j] = sl[j][0];
#pragma unroll
          for (int q = 1; q < KS; ++q)
            if (q < nsp) g[j] += sl[j][q];
          for (int q = KS; q < nsp; ++q) g[j] += slab_get(slabs + (size_t)q * stride + w[j]);
        }
        gb = sb[0];
#pragma unroll
        for (int q = 1; q < KS; ++q)
          if (q < nsp) gb += sb[q];
        for (int q = KS; q < nsp; ++q) gb += slab_get(slabs + (size_t)q * stride + wb);
      };
      // (80 x 80 tiles carry 25 elements per lane: 8 slabs each in flight on top of their state would spill)
      if (nsp <= 4 || PER > 16)
        gather(std::integral_constant<int, 4>{});
      else
        gather(std::integral_constant<int, 8>{});
    } else {
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int off = offl[j];
        g[j] = ((red[off] + red[TW * LD + off]) + red[2 * TW * LD + off]) + red[3 * TW * LD + off];
      }
      const float* db = red + 4 * TW * LD;
      const int tb = has_b ? tid : 0;
      gb = ((db[tb] + db[TW + tb]) + db[2 * TW + tb]) + db[3 * TW + tb];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (ok[j]) apply(w[j], g[j], pv[j], mv[j], vv[j], tv0[j], mf[j], mb[j]);
    if (has_b) apply(wb, gb, pb_, mb_, vb_, tb_, -1, -1);
  });
}
template <int T>
__global__ __launch_bounds__(256, 1) void mlp_dwt_adam_kernel(const DwAdamArgs a) {
  mlp_dwt_adam_body<T, const DwAdamArgs&>(a);
}
template <int T>
__global__ __launch_bounds__(256, 1) void mlp_dwt_adam_kernel_p(const void* p) {
  mlp_dwt_adam_body<T, const OSRL_CAS DwAdamArgs&>(*(const OSRL_CAS DwAdamArgs*)p);
}


// ---- dW for big row counts: one wave = one 128x128 tile, one wave per SIMD ------------------------------------
// mlp_dw_kernel above is built for B = 2048-4096 rows (many small tiles, 4 waves splitting a few hundred rows).  At
// token-matrix sizes (CDT: M = 81920 rows, dW tiles of 768x256 ... 256x1024) its k-loop is the regime tools/
// loop_probe2.hip diagnoses: 64x64 tiles per wave, 2 waves per SIMD, 32 fragment loads per 64 MFMAs -> 58 % of the roof.
// Here every wave owns a 128x64 output tile (128 accumulator registers) for one split of the rows: 48 fragment loads
// feed 128 MFMAs per 16-row k-step, the next step's loads are issued before this step's MFMAs (order pinned), four
// independent waves form a workgroup (no LDS traffic, no barriers; the LDS request only keeps it at one workgroup per
// CU = one wave per SIMD), and the host picks the split count so that tiles x splits fill the 256 CUs in one round.
// Takes the (entry, 128-row block, 64-column block) items whose tile lies fully inside dW; the rest of the plan
// (ragged edges, narrow layers, biases of layers with no full tile) stays with mlp_dw_kernel.
constexpr int kDwbO = 8, kDwbI = 4;  // 16-wide blocks per wave tile: 128 (out) x 64 (in); 8 x 8 (256 accumulators)
                                      // spills 373 registers around the loop even at 512 per lane

struct DwBigFrag {
  f32x4 a[kDwbO], b[kDwbI];
};

// Round 4 (tools/dw_lab.hip, bit-identical to the round-3 kernel, 3388 -> 3164 us for the 12 CDT projections): the
// fragment registers are filled by dwordx4 loads of 4 CONSECUTIVE columns -- register a[4 q + j] of lane (m, kq) =
// dz[row][o0 + 64 q + 4 m + j], i.e. as an MFMA operand the permuted 16-column block {o0 + 64 q + 4 i + j : i = 0..15};
// the store undoes the permutation with 16-byte stores.  12 loads of 4 rows x 256 B per k-step instead of 48 of
// 4 rows x 64 B, no row guard outside the tail, and running fragment pointers (8 64-bit adds per k-step instead of the
// row * stride multiplies: the f32 MFMA runs on the vector ALUs, so every VALU instruction is time taken from it).
// The same (row -> MFMA, k-lane) map as before: the same bits.  Needs 16-byte aligned operands and strides % 4 == 0
// (the plan builder, engine/core.py DwPlan, sends other layers to mlp_dw_kernel).  What is left (0.78 of the roof):
// with every operand row cache-resident the same loop runs at 0.82 -- each (tile, split) wave streams its own
// 2560 x 192 block, 18 GB per launch through L2 / MALL; a workgroup-wide 256 x 256 tile through LDS would halve it.
template <bool GUARD>
__device__ __forceinline__ void dwp_load(DwBigFrag& f, const float* __restrict__ dz, const float* __restrict__ av,
                                         size_t ldz, size_t lda_g, int r0, int r_end, int m, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = r0 + 4 * kq + t;
    const bool rok = !GUARD || r < r_end;
    const size_t rc = (size_t)(rok ? r : r_end - 1);
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

__global__ __launch_bounds__(256, 1) void mlp_dw_big_kernel(const osrl_dw_entry_t* __restrict__ entries,
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


// ---- dW for big row counts, 256 x 256 tiles through LDS (round 4; tools/dw_lab.hip) ---------------------------------
// The waves of mlp_dw_big_kernel stream private 2560 x 192 operand blocks: 18 GB per launch of the CDT projections through
// L2 / MALL (with the operand rows cache-resident the same loop runs at 0.82 of the roof instead of 0.78).  Here the eight
// waves of a workgroup (2 x 4: 128 out-columns x 64 in-columns each, the same 128-register accumulator tile and the same
// permuted-column fragment registers) share one DMA-fed slab ring: 16 rows x (256 + 256) columns = 32 KB per k-step, every
// DMA piece one whole 1 KB row (lane l <- 16 bytes at column 4 l), 4 slots, conflict-free ds_read_b128 (row pitch 1 KB,
// lanes of a service group on 16 different 16-byte slots), the DMA issue point staggered between the two waves of a
// SIMD.  6 GB per launch; 3390 (round 3) / 3170 (mlp_dw_big_kernel now) -> 2980 us, 0.83 of the fp32 MFMA roof, and as
// few row splits as fill the CUs once (36 tiles x 7).  Fragments are single-buffered: 128 + 48 registers of the 256 a
// wave has at two per SIMD (double-buffered they spill).
// Requires out % 256 == 0, in % 256 == 0, rows % 16 == 0, 16-byte aligned operands / strides % 4 == 0 (DwPlan checks).
constexpr int kCoS = 4;  // slots
__device__ __forceinline__ void dwc_glds16(const void* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__global__ __launch_bounds__(512, 2) void mlp_dw_coop_kernel(const osrl_dw_entry_t* __restrict__ entries,
                                                  const int32_t* __restrict__ items, int n_items, int rows,
                                                  int rows_per_split, float* __restrict__ slabs, int64_t slab_stride) {
  extern __shared__ __attribute__((aligned(16))) float lds_co[];
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
          dwc_glds16(src[p], lds_co + it_slot * kSlot + dst_off[p]);
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
      const float* sl = lds_co + slot * kSlot;
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

}  // namespace

extern "C" int osrl_mlp_backward_dw_big(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items,
                                        int32_t rows, int32_t n_splits, float* slabs, int64_t slab_stride,
                                        void* stream) {
  if (!d_entries || !d_items || n_items < 1 || rows < 1 || n_splits < 1 || !slabs) return -1;
  int rps = (rows + n_splits - 1) / n_splits;
  rps = (rps + 15) & ~15;  // whole 16-row k-steps
  constexpr int kLds = 96 * 1024;  // unused: more than half of the CU's LDS -> one workgroup (one wave per SIMD) per CU
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dw_big_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(mlp_dw_big_kernel, dim3((n_items + 3) / 4, n_splits, 1), dim3(256), kLds, (hipStream_t)stream,
                     d_entries, d_items, n_items, rows, rps, slabs, slab_stride);
  return (int)hipGetLastError();
}

extern "C" int osrl_mlp_backward_dw_coop(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items,
                                         int32_t rows, int32_t n_splits, float* slabs, int64_t slab_stride,
                                         void* stream) {
  if (!d_entries || !d_items || n_items < 1 || rows < 16 || (rows & 15) || n_splits < 1 || !slabs) return -1;
  int rps = (rows + n_splits - 1) / n_splits;
  rps = (rps + 15) & ~15;  // whole 16-row k-steps
  constexpr int kLds = sizeof(float) * kCoS * 2 * 16 * 256;  // 128 KB: one workgroup (two waves per SIMD) per CU
  // per device and stateless, like every sibling launcher: the attribute belongs to the CURRENT device
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dw_coop_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(mlp_dw_coop_kernel, dim3(n_items, n_splits, 1), dim3(512), kLds, (hipStream_t)stream, d_entries,
                     d_items, n_items, rows, rps, slabs, slab_stride);
  return (int)hipGetLastError();
}

template <int T>
static int launch_dwt(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_work, int32_t rows, float* slabs,
                      int64_t slab_stride, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dwt_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dwt_lds<T>());
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(mlp_dwt_kernel<T>, dim3(n_work), dim3(256), dwt_lds<T>(), stream, d_entries, d_items, rows, slabs,
                     slab_stride);
  return (int)hipGetLastError();
}

extern "C" int osrl_mlp_backward_dw_tiles(const osrl_dw_entry_t* d_entries, const int32_t* d_work, int32_t n_work,
                                          int32_t rows, int32_t tile_blocks, float* slabs, int64_t slab_stride,
                                          void* stream) {
  if (!d_entries || !d_work || n_work < 1 || rows < 1 || !slabs) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (tile_blocks == 5) return launch_dwt<5>(d_entries, d_work, n_work, rows, slabs, slab_stride, (hipStream_t)stream);
  if (tile_blocks == 4) return launch_dwt<4>(d_entries, d_work, n_work, rows, slabs, slab_stride, (hipStream_t)stream);
  // 48 x 48 / 32 x 32 tiles: 38 / 17 KB of LDS and < 100 registers per lane, i.e. workgroups that fit on a CU beside an
  // 80-row forward workgroup (which leaves a 64 x 64 tile's 68 KB no room)
  if (tile_blocks == 3) return launch_dwt<3>(d_entries, d_work, n_work, rows, slabs, slab_stride, (hipStream_t)stream);
  if (tile_blocks == 2) return launch_dwt<2>(d_entries, d_work, n_work, rows, slabs, slab_stride, (hipStream_t)stream);
  return -1;
}

template <int T>
static int launch_dwt_adam(const DwAdamArgs& a, int32_t n_work, hipStream_t stream) {
  const void* dev_args = osrl_argmem::slot(a);
  hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(mlp_dwt_adam_kernel_p<T>)
                                              : reinterpret_cast<const void*>(mlp_dwt_adam_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)dwt_lds<T>());
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  if (dev_args)
    hipLaunchKernelGGL(mlp_dwt_adam_kernel_p<T>, dim3(n_work), dim3(256), dwt_lds<T>(), stream, dev_args);
  else
    hipLaunchKernelGGL(mlp_dwt_adam_kernel<T>, dim3(n_work), dim3(256), dwt_lds<T>(), stream, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_mlp_backward_dw_tiles_adam(const osrl_dw_entry_t* d_entries, const int32_t* d_work,
                                               const int32_t* d_tile_ids, uint32_t* d_counters, int32_t n_work,
                                               int32_t rows, int32_t tile_blocks, float* slabs, int64_t slab_stride,
                                               const osrl_dw_adam_t* o, void* stream) {
  if (!d_entries || !d_work || !d_tile_ids || !d_counters || n_work < 1 || rows < 1 || !slabs || !o) return -1;
  if (!o->p || !o->m || !o->v || !o->st || (o->map_f && !o->pf) || (o->map_b && (!o->pb || !o->map_f))) return -1;
  DwAdamArgs a{};
  a.entries = d_entries;
  a.items = d_work;
  a.tile_ids = d_tile_ids;
  a.counters = d_counters;
  a.slabs = slabs;
  a.slab_stride = slab_stride;
  a.p = o->p; a.m = o->m; a.v = o->v; a.tgt = o->tgt;
  a.map_f = o->map_f; a.map_b = o->map_b;
  a.pf = o->pf; a.pb = o->pb; a.tf = o->tgt ? o->tf : nullptr;
  a.st = o->st;
  a.lr = o->lr; a.b1 = o->beta1; a.b2 = o->beta2; a.eps = o->eps; a.tau = o->tau;
  a.rows = rows;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipStream_t st = (hipStream_t)stream;
  if (tile_blocks == 5) return launch_dwt_adam<5>(a, n_work, st);
  if (tile_blocks == 4) return launch_dwt_adam<4>(a, n_work, st);
  if (tile_blocks == 3) return launch_dwt_adam<3>(a, n_work, st);
  if (tile_blocks == 2) return launch_dwt_adam<2>(a, n_work, st);
  return -1;
}

extern "C" int osrl_mlp_backward_dw(const osrl_dw_entry_t* d_entries, const int32_t* d_items, int32_t n_items,
                                    int32_t rows, int32_t n_splits, float* slabs, int64_t slab_stride,
                                    void* stream) {
  if (!d_entries || !d_items || n_items < 1 || rows < 1 || n_splits < 1 || !slabs) return -1;
  int rps = (rows + n_splits - 1) / n_splits;
  rps = (rps + 63) & ~63;  // 4 waves x whole 16-row k-steps
  dim3 grid(n_items, n_splits, 1);
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dw_kernel),  // 66 KB dynamic LDS: opt in
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDwLds);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  hipLaunchKernelGGL(mlp_dw_kernel, grid, dim3(256), kDwLds, (hipStream_t)stream, d_entries, d_items, n_items, rows,
                     rps, slabs, slab_stride);
  return (int)hipGetLastError();
}
