// dwt.h -- one (tile, row split) item of the dW GEMM  dW[out, in] = dZ^T A  on (16 T) x (16 T) tiles: fragment loads,
// the MFMA step and dwt_tile() (k-loop + the four waves' partials into LDS + a caller-given epilogue).  Shared by the dW
// launches (mlp_dw.hip: slab store, or the optimizer step) and the one-launch regression step (mlp.hip).
#pragma once
#include "mlp_common.h"

namespace {

// ---- dW on (16 T) x (16 T) tiles with a flat (tile, row split) work list ---------------------------------------
// mlp_dw_kernel deals 64 x 64 tiles x a common split count.  For the 400-wide VAE (25 column blocks) that is 140 tiles,
// 13 of every 49 ragged, x 2 splits = 280 workgroups on 256 CUs: the CUs that get two full tiles set the pace (49.6 us
// for 1.59 GFLOP = 0.20 of the fp32 roof, profiles/r2_bench_trace_summary.txt).  25 = 5 x 5: with T = 5 (80 x 80
// tiles, 25 accumulator tiles per wave) the VAE's six layers are 60 full tiles + 10 one-block-high strips, no ragged
// edge anywhere; the work list gives a full tile 4 row splits (each wave 128 rows = 800 MFMAs) and a strip 1 (each wave
// 512 rows = 640 MFMAs): 250 workgroups, one per CU, one round, even work.  Same arithmetic per output element as
// mlp_dw_kernel (a wave's k-ordered MFMA chain over its rows, four partials summed in wave order, slabs summed in
// split order by the consumer), so the sum order -- and the bits -- depend only on (rows per wave), as before.
// items[4 i ..] = (entry, out tile, in tile, split | n_splits << 16).
template <int T>
struct DwFragT {
  f32x4 a[T], b[T];
};

// One 16-row k-step of fragments, UNMASKED: the column offsets oa / ia are loop invariants, clamped once to stay in
// bounds -- a lane of an invalid column reads element 0 of the row and pollutes only output rows o >= out / columns
// i >= in, which are never stored (an MFMA's output element (o, i) depends on operand rows o and i alone).  With the
// select-per-load form of mlp_dw_kernel every load is consumed by a v_cndmask right behind it, so the loads of step
// k + 1 are waited for BEFORE the MFMAs of step k start and a k-step costs latency + MFMA time instead of the larger
// of the two (tools/dw_bench.py: 54 us for the VAE group whatever the tiling).  Only rows are masked, and only in a
// wave's last, partial k-step (dwt_load_tail).
template <int T>
__device__ __forceinline__ void dwt_load(DwFragT<T>& f, const float* __restrict__ pz, const float* __restrict__ pa,
                                         size_t ldz, size_t lda_g, const unsigned (&oa)[T], const unsigned (&ia)[T],
                                         int nob, int nib) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    // branch-free: blocks past nob / nib re-read column 0 (a guarded load gets its own wait)
#pragma unroll
    for (int ob = 0; ob < T; ++ob) f.a[ob][t] = pz[t * ldz + oa[ob]];
#pragma unroll
    for (int ib = 0; ib < T; ++ib) f.b[ib][t] = pa[t * lda_g + ia[ib]];
  }
}
// the partial last k-step: rows >= r_end contribute zeros (A operand zeroed; B then does not matter)
template <int T>
__device__ __forceinline__ void dwt_load_tail(DwFragT<T>& f, const float* __restrict__ dz, const float* __restrict__ av,
                                              size_t ldz, size_t lda_g, const unsigned (&oa)[T], const unsigned (&ia)[T],
                                              int nob, int nib, int r0, int r_end, int kq) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = r0 + 4 * kq + t;
    const bool rok = r < r_end;
    const size_t rc = (size_t)(rok ? r : r_end - 1);
#pragma unroll
    for (int ob = 0; ob < T; ++ob) {
      const float v = dz[rc * ldz + oa[ob]];
      f.a[ob][t] = rok ? v : 0.f;
    }
#pragma unroll
    for (int ib = 0; ib < T; ++ib) f.b[ib][t] = av[rc * lda_g + ia[ib]];
  }
}

// FULL: all T x T blocks of the tile exist -- straight-line MFMAs (a guard per block is a branch per block, and every
// branch target gets a conservative s_waitcnt vmcnt(0): the next step's loads would be waited for before this step's
// MFMAs start); ragged tiles and one-block strips take the guarded form
// SHAPE 2 = 1 x T (the narrow heads' strips: one output block), also straight-line; SHAPE 0 = anything else, guarded
template <int T, int SHAPE>
__device__ __forceinline__ void dwt_mma(f32x4 (&acc)[T][T], float (&dbacc)[T], const DwFragT<T>& f, int nob, int nib,
                                        bool want_db) {
  constexpr bool FULL = SHAPE == 1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int ob = 0; ob < (SHAPE == 2 ? 1 : T); ++ob) {
      if (SHAPE != 0 || ob < nob) {
#pragma unroll
        for (int ib = 0; ib < T; ++ib)
          if (SHAPE != 0 || ib < nib) acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ob][t], f.b[ib][t], acc[ob][ib], 0, 0, 0);
      }
    }
  }
  (void)FULL;
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < T; ++ob) dbacc[ob] += (f.a[ob][0] + f.a[ob][1]) + (f.a[ob][2] + f.a[ob][3]);
  }
}

template <int T>
constexpr size_t dwt_lds() { return sizeof(float) * (4 * (16 * T) * (16 * T + 1) + 4 * 16 * T); }

// One work item: the four waves' partials of a tile go to LDS (red: [4][16T][16T+1] + [4][16T] bias partials), then
// epi(E, o0, i0, split | n_splits << 16, want_db) consumes them after a barrier: the slab store of mlp_dwt_kernel, or the optimizer
// step itself when the item covers all rows (mlp_step_kernel).  Waves beyond the first four (a wider workgroup) only
// take part in the barrier and the epilogue.
// WARM (experiment, off: every lane first touches the 128-byte lines of its wave's row range -- measured 7.5 -> 9.7 us
// for the one-launch step's dW phase, tools/step_stamps.py).
// DEEP (the one-launch step at <= 64 rows per wave): the four k-steps' fragments are all requested before the first
// MFMA -- one round trip to operands that other XCDs wrote moments ago instead of three (same MFMA order, same bits).
template <int T, bool WARM = false, bool DEEP = false, class EPI>
__device__ __forceinline__ void dwt_tile(const osrl_dw_entry_t* __restrict__ entries, const int32_t* __restrict__ items,
                                         const int item, const int rows, float* __restrict__ red, EPI epi) {
  constexpr int TW = 16 * T, LD = TW + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ei = items[item * 4 + 0], ot = items[item * 4 + 1], it = items[item * 4 + 2], sp = items[item * 4 + 3];
  const int s = sp & 0xffff, nsp = sp >> 16;
  // read through the constant address space: pointers loaded from there are known-global (global_load with a scalar
  // base); loaded from a plain global struct they are generic and every fragment load becomes a flat_load with its own
  // 64-bit address add, counted on BOTH vmcnt and lgkmcnt
  const OSRL_CAS osrl_dw_entry_t& E = ((const OSRL_CAS osrl_dw_entry_t*)entries)[ei];
  const int out = E.out, in = E.in;
  const size_t ldz = E.ldz > 0 ? (size_t)E.ldz : (size_t)out, lda_g = E.lda > 0 ? (size_t)E.lda : (size_t)in;
  const int o0 = ot * TW, i0 = it * TW;
  int rps = (rows + nsp - 1) / nsp;
  rps = (rps + 63) & ~63;  // 4 waves x whole 16-row k-steps
  const int rpw = rps >> 2;
  const int r_begin = s * rps + wave * rpw;
  int r_end = r_begin + rpw;
  r_end = r_end > rows ? rows : r_end;
  const int m = lane & 15, kq = lane >> 4;
  int nob = (out - o0 + 15) >> 4;
  nob = nob > T ? T : nob;
  int nib = (in - i0 + 15) >> 4;
  nib = nib > T ? T : nib;
  const bool want_db = it == 0;

  if (wave < 4) {
  f32x4 acc[T][T];
  zero_acc<T, T>(acc);
  float dbacc[T];
#pragma unroll
  for (int ob = 0; ob < T; ++ob) dbacc[ob] = 0.f;
  const float* __restrict__ dz = E.dz;
  const float* __restrict__ av = E.a;
  unsigned oa[T], ia[T];  // this lane's column of each block, clamped into the row (see dwt_load)
#pragma unroll
  for (int b = 0; b < T; ++b) {
    const int o = o0 + b * 16 + m, i = i0 + b * 16 + m;
    oa[b] = (unsigned)(o < out ? o : 0);  // (also every block past nob / nib: o >= out, i >= in there)
    ia[b] = (unsigned)(i < in ? i : 0);
  }
  if (r_begin < r_end) {
    const int n_full = (r_end - r_begin) >> 4;  // whole 16-row k-steps
    const float* __restrict__ pz = dz + (size_t)(r_begin + 4 * kq) * ldz;
    const float* __restrict__ pa = av + (size_t)(r_begin + 4 * kq) * lda_g;
    float wt[WARM ? 4 * ((TW + 31) / 32) : 1];
    if constexpr (WARM) {
      constexpr int NL = (TW + 31) / 32;  // lines per row of a panel
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        int r = r_begin + pass * 64 + lane;
        r = r < r_end ? r : r_end - 1;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          const int oc = o0 + 32 * j < out ? o0 + 32 * j : out - 1, ic = i0 + 32 * j < in ? i0 + 32 * j : in - 1;
          wt[(pass * 2 + 0) * NL + j] = dz[(size_t)r * ldz + oc];
          wt[(pass * 2 + 1) * NL + j] = av[(size_t)r * lda_g + ic];
        }
      }
    }
    DwFragT<T> f0, f1;
    // no control flow inside the pair loop (the reload past the end re-reads the last step): the compiler can then
    // count the loads in flight (s_waitcnt vmcnt(n > 0)) and step k's MFMAs run under step k + 1's loads
    auto run = [&](auto full_c) {
      constexpr int FULL = decltype(full_c)::value;
      if constexpr (DEEP) {
        if (n_full == 4 && r_begin + 64 == r_end) {
          DwFragT<T> f2, f3;
          dwt_load<T>(f0, pz, pa, ldz, lda_g, oa, ia, nob, nib);
          dwt_load<T>(f1, pz + (size_t)16 * ldz, pa + (size_t)16 * lda_g, ldz, lda_g, oa, ia, nob, nib);
          dwt_load<T>(f2, pz + (size_t)32 * ldz, pa + (size_t)32 * lda_g, ldz, lda_g, oa, ia, nob, nib);
          dwt_load<T>(f3, pz + (size_t)48 * ldz, pa + (size_t)48 * lda_g, ldz, lda_g, oa, ia, nob, nib);
          dwt_mma<T, FULL>(acc, dbacc, f0, nob, nib, want_db);
          dwt_mma<T, FULL>(acc, dbacc, f1, nob, nib, want_db);
          dwt_mma<T, FULL>(acc, dbacc, f2, nob, nib, want_db);
          dwt_mma<T, FULL>(acc, dbacc, f3, nob, nib, want_db);
          return;
        }
      }
      // (round 4, measured and removed: a ring of four fragment sets = three k-steps of loads in flight.  The isolated
      // launches of all four CPQ groups moved by -4 .. +8 %, the step not at all (profiles/r4_dw_ring_ab.txt): the k-loop's
      // load latency is not what bounds these launches.)
      if (n_full > 0) dwt_load<T>(f0, pz, pa, ldz, lda_g, oa, ia, nob, nib);
      int k = 0;
      for (; k + 1 < n_full; k += 2) {
        dwt_load<T>(f1, pz + (size_t)(k + 1) * 16 * ldz, pa + (size_t)(k + 1) * 16 * lda_g, ldz, lda_g, oa, ia, nob, nib);
        dwt_mma<T, FULL>(acc, dbacc, f0, nob, nib, want_db);
        const int kn = k + 2 < n_full ? k + 2 : n_full - 1;
        dwt_load<T>(f0, pz + (size_t)kn * 16 * ldz, pa + (size_t)kn * 16 * lda_g, ldz, lda_g, oa, ia, nob, nib);
        dwt_mma<T, FULL>(acc, dbacc, f1, nob, nib, want_db);
      }
      if (k < n_full) dwt_mma<T, FULL>(acc, dbacc, f0, nob, nib, want_db);  // odd count: the last whole step is in f0
      if (r_begin + 16 * n_full < r_end) {
        dwt_load_tail<T>(f1, dz, av, ldz, lda_g, oa, ia, nob, nib, r_begin + 16 * n_full, r_end, kq);
        dwt_mma<T, FULL>(acc, dbacc, f1, nob, nib, want_db);
      }
    };
    if (nob == T && nib == T)
      run(std::integral_constant<int, 1>{});
    else if (nob == 1 && nib == T)
      run(std::integral_constant<int, 2>{});
    else
      run(std::integral_constant<int, 0>{});
    if constexpr (WARM) {
#pragma unroll
      for (int j = 0; j < 4 * ((TW + 31) / 32); ++j) asm volatile("" ::"v"(wt[j]));
    }
  }
  // ---- 4 partials -> LDS -> fixed-order sum -> one coalesced slab tile
  float* mine = red + wave * TW * LD;
#pragma unroll
  for (int ob = 0; ob < T; ++ob)
#pragma unroll
    for (int ib = 0; ib < T; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(ob * 16 + kq * 4 + r) * LD + ib * 16 + m] = acc[ob][ib][r];
  if (want_db) {
#pragma unroll
    for (int ob = 0; ob < T; ++ob) {
      float v = dbacc[ob];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) red[4 * TW * LD + wave * TW + ob * 16 + m] = v;
    }
  }
  }  // wave < 4
  __syncthreads();
  epi(E, o0, i0, sp, want_db);
}

}  // namespace
