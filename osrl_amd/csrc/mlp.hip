// mlp.hip -- fused multi-layer-perceptron forward / backward for gfx950 (MI355X, CDNA4).
//
// Replaces, for the OSRL hot path, every `nn.Sequential(Linear, act, ...)` forward and its
// autograd (osrl/common/net.py:12-30 mlp(); users: MLPActor :65-85, SquashedGaussianMLPActor
// :152-205 trunk+heads, EnsembleQCritic :208-242, EnsembleDoubleQCritic :245-287, VAE :290-339,
// MLPGaussianPerturbationActor :33-62).
//
// Design (fp32 everywhere: the parity budget is 1e-4 -> only v_mfma_f32_16x16x4_f32, which is an
// exact k-ordered fmaf chain, see /opt/skills/guides/cdna_hip_programming.md section 3):
//   * one workgroup = 4 wave64 = one tile of BM = 16*NRB rows of ONE ensemble member;
//   * the row tile's activations live in ONE LDS buffer [BM][lda] that every layer overwrites in
//     place (k-loop reads it -> barrier -> epilogue writes the next layer's activations);
//   * wave w owns a contiguous group of 16-wide output-column blocks and ALL NRB row blocks, so
//     the A fragments (activations, ds_read_b128) are shared by the 4 waves through LDS while each
//     weight element is fetched by exactly one wave, straight from L2 into its B fragment;
//   * k-slot trick: MFMA 16x16x4 sums over 4 k-slots; slot kq of the t-th MFMA is mapped to
//     k = k0 + 4*kq + t, so ONE 16-byte load per lane (A: ds_read_b128, B: global_load_dwordx4 of
//     4 consecutive k of a weight row) feeds 4 MFMAs.  Per 16-deep k step a wave issues
//     NRB + NCB wide loads for 4*NRB*NCB MFMAs (8 loads : 64 MFMAs at NRB=NCB=4);
//   * weights are read from a PACKED, fragment-ordered copy  P[k/16][(k%16)/4][n][k%4]  (zero
//     padded to 16 in both dims, refreshed by osrl_pack_weights after every optimizer step):
//     lanes m=0..15 of a fragment load then read 16 consecutive 16-byte words, i.e. every
//     global_load_dwordx4 is 4 x 256 B contiguous and guard-free.  (Loading fragments from the
//     canonical [out,in] layout puts consecutive lanes 1 KB apart -> 64 L1 requests per
//     instruction; measured: the TA/L1 pipe time per k-step then equals the MFMA time and the
//     kernel sits at ~30% of the fp32 MFMA roof.)  The backward pass uses the analogous packing
//     of W^T.
//   * lda = round16(max width) + 8 floats => (lda/4) mod 16 is 2 mod 4, which makes the
//     ds_read_b128 A-fragment pattern (16 rows x 4 k-quads per wave) bank-conflict free for the
//     four 16-lane service groups of ds_read_b128 (MI355X_MICROARCH.md LDS table).
//   * backward-dz walks the same structure with the transposed weight access; backward-dw is a
//     split-K (over rows) MFMA GEMM dW = dZ^T A writing per-split slabs that the Adam kernel sums
//     in a fixed order (deterministic, no atomics).
#include "dwt.h"
#include "adam.h"
#include "gather.h"
#include "step.h"
#include "trace.h"

namespace {


template <int RW, int CNT, int STAGES, int RD>
__device__ __forceinline__ void mm_prefetch(f32x4 (&b)[RD][RW], int nk, const float* __restrict__ P, int Np, int col0,
                                            int kc0 = 0) {
  static_assert(CNT <= RW && STAGES <= RD, "ring too small");
  const int lane = threadIdx.x & 63;
  const unsigned lane_off = (unsigned)(((lane >> 4) * Np + col0 + (lane & 15)) * 16);  // bytes
  const int rot = k_rot(nk);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
    const int kc = k_at(s < nk ? s : nk - 1, rot, nk, kc0);
    const float* __restrict__ Pk = P + (size_t)kc * 16 * Np;
#pragma unroll
    for (int c = 0; c < CNT; ++c) b[s][c] = load_bp_s(Pk, lane_off + c * 256);
  }
}

// Only the first CNT (<= AW) column blocks of acc are touched, so a wave with fewer blocks runs a dense loop
// on the same accumulator array.
template <int NRB, int AW, int RW, int CNT, int STAGES, int RD>
__device__ __forceinline__ void mm_run(const float* lds, int lda, int nk, const float* __restrict__ P, int Np, int col0,
                                       f32x4 (&acc)[NRB][AW], f32x4 (&b)[RD][RW], int kc0 = 0) {
  static_assert(CNT <= AW && CNT <= RW && STAGES <= RD, "tile shapes");
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  const float* arow = lds + m * lda + 4 * kq;
  const unsigned lane_off = (unsigned)((kq * Np + col0 + m) * 16);  // bytes
  const int rot = k_rot(nk);
  f32x4 a[2][NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
    a[0][rb] = EXP_AREAD(arow + rb * 16 * lda + k_at(0, rot, nk, kc0) * 16);
  // one k-step with compile-time ring / double-buffer slots (S = step index modulo U)
  auto step = [&](auto s_c, int kc) {
    constexpr int s = decltype(s_c)::value;
    {  // issue the loads of k-step kc + STAGES - 1 into the ring slot freed by step kc - 1
      int kl = kc + STAGES - 1;
      kl = k_at(kl < nk ? kl : nk - 1, rot, nk, kc0);
      const float* __restrict__ Pk = P + (size_t)kl * 16 * Np;
#pragma unroll
      for (int c = 0; c < CNT; ++c) b[(s + STAGES - 1) % STAGES][c] = load_bp_s(Pk, lane_off + c * 256);
      const int ka = k_at(kc + 1 < nk ? kc + 1 : kc, rot, nk, kc0);
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) a[(s + 1) & 1][rb] = EXP_AREAD(arow + rb * 16 * lda + ka * 16);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) acc[rb][c] = EXP_MFMA(a[s & 1][rb][t], b[s % STAGES][c][t], acc[rb][c]);
    if constexpr (NRB >= kPinRows) {
      // One wave per SIMD (80-row tiles): nothing else hides a load's latency, so the order inside a k-step is pinned:
      // the next ring slot's weight loads and the next step's ds_reads go out FIRST, then this step's MFMAs.  Left to
      // itself the machine scheduler sinks the loads to ~35 MFMAs (~1100 cycles) before their first use to save
      // registers, and the wave then sits in s_waitcnt vmcnt at the top of every k-step.
      __builtin_amdgcn_sched_group_barrier(0x020, CNT, 0);           // VMEM reads
      __builtin_amdgcn_sched_group_barrier(0x100, NRB, 0);           // DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NRB * CNT, 0); // MFMA
    }
  };
  // Main loop: groups of U = 2*STAGES steps with NO per-step control flow (every taken branch costs the wave an
  // instruction-buffer refill: with a conditional per unrolled step a lone wave reached 74% MFMA issue in a
  // load-free loop, tools/mlp_phase.hip); the < U leftover steps run in a branchy tail.
  using std::integral_constant;
  int kc = 0;
  if constexpr (STAGES > 3) {
    // deep ring: groups of STAGES steps (even: the A double buffer alternates with the step index), then the
    // < STAGES leftover steps with one uniform branch each
    static_assert(STAGES % 2 == 0, "the A double buffer needs an even group");
    for (; kc + STAGES <= nk; kc += STAGES) static_for<STAGES>([&](auto s_c) { step(s_c, kc + decltype(s_c)::value); });
    static_for<STAGES - 1>([&](auto s_c) {
      if (kc + decltype(s_c)::value < nk) step(s_c, kc + decltype(s_c)::value);
    });
    return;
  }
  constexpr int U = 2 * STAGES;
  for (; kc + U <= nk; kc += U) {
    step(integral_constant<int, 0>{}, kc);
    step(integral_constant<int, 1>{}, kc + 1);
    step(integral_constant<int, 2>{}, kc + 2);
    step(integral_constant<int, 3>{}, kc + 3);
    if constexpr (U > 4) {
      step(integral_constant<int, 4>{}, kc + 4);
      step(integral_constant<int, 5>{}, kc + 5);
    }
  }
  if (kc < nk) {
    step(integral_constant<int, 0>{}, kc);
    if (kc + 1 < nk) {
      step(integral_constant<int, 1>{}, kc + 1);
      if (kc + 2 < nk) {
        step(integral_constant<int, 2>{}, kc + 2);
        if constexpr (U > 4) {
          if (kc + 3 < nk) {
            step(integral_constant<int, 3>{}, kc + 3);
            if (kc + 4 < nk) step(integral_constant<int, 4>{}, kc + 4);
          }
        }
      }
    }
  }
}

template <int NRB, int NCB, int RD = kRing>
constexpr int mm_stages() {  // short k-steps need a deeper load ring
  return RD > 3 ? RD : (NRB * NCB >= 16 || NCB >= 7) ? 2 : 3;
}

// cnt (<= NCB) column blocks starting at column col0 (= first block * 16 [+ dx_col0]); 0 = idle wave
template <int NRB, int NCB, int RD>
__device__ __forceinline__ void layer_prefetch(f32x4 (&ring)[RD][NCB], int nk, const float* __restrict__ P, int Np,
                                               int col0, int cnt) {
  constexpr int ST = mm_stages<NRB, NCB, RD>();
  if (cnt == NCB) {
    mm_prefetch<NCB, NCB, ST>(ring, nk, P, Np, col0);
  } else if (NCB > 2 && cnt == NCB - 1) {
    mm_prefetch<NCB, (NCB > 2 ? NCB - 1 : 1), ST>(ring, nk, P, Np, col0);
  } else if (cnt > 0) {
    mm_prefetch<NCB, 1, 3>(ring, nk, P, Np, col0);  // ragged wave: first block only
  }
}

// requires layer_prefetch(ring, same arguments) to have been issued by this wave
template <int NRB, int NCB, int RD>
__device__ __forceinline__ void layer_run(const float* lds, int lda, int nk, const float* __restrict__ P, int Np,
                                          int col0, int cnt, f32x4 (&acc)[NRB][NCB], f32x4 (&ring)[RD][NCB]) {
  constexpr int ST = mm_stages<NRB, NCB, RD>();
  if (cnt == NCB) {
    mm_run<NRB, NCB, NCB, NCB, ST>(lds, lda, nk, P, Np, col0, acc, ring);
  } else if (NCB > 2 && cnt == NCB - 1) {
    // balanced split of e.g. 25 column blocks as 7/6/6/6: the 6-block waves keep a dense k-loop
    mm_run<NRB, NCB, NCB, (NCB > 2 ? NCB - 1 : 1), ST>(lds, lda, nk, P, Np, col0, acc, ring);
  } else {
    // ragged wave (few column blocks, e.g. the 1-wide Q head): one block at a time
    for (int c = 0; c < cnt; ++c) {
      f32x4 t[NRB][1];  // starts from acc (the caller's initial value, e.g. the bias)
#pragma unroll
      for (int cc = 0; cc < NCB; ++cc)
        if (cc == c) {
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) t[rb][0] = acc[rb][cc];
        }
      if (c > 0) mm_prefetch<NCB, 1, 3>(ring, nk, P, Np, col0 + c * 16);
      mm_run<NRB, 1, NCB, 1, 3>(lds, lda, nk, P, Np, col0 + c * 16, t, ring);
#pragma unroll
      for (int cc = 0; cc < NCB; ++cc)
        if (cc == c) {
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) acc[rb][cc] = t[rb][0];
        }
    }
  }
}

// Narrow layer (N <= 32, e.g. the 1-wide Q head or the 2*ad-wide policy head): instead of one wave
// walking all of K alone, the 4 waves split K; partial tiles are summed through the (free) LDS
// activation buffer.  On return lds[row][c], c < nblk*16, holds the raw sums (no bias/activation).
// Requires lda >= 16*NW and nk >= 4.  Contains the barriers that protect the in-place overwrite.
struct NarrowPart {
  int col, k_lo, k_n;
};
template <int NW = 4>
__device__ __forceinline__ NarrowPart narrow_part(int nk, int col_off, int nblk, int wave) {
  const int parts = NW / nblk;  // waves per column block (nblk is 1 or 2)
  const int blk = wave / parts, part = wave - blk * parts;
  const int k_lo = (nk * part) / parts, k_hi = (nk * (part + 1)) / parts;
  return NarrowPart{col_off + blk * 16, k_lo, k_hi - k_lo};
}
template <int NCB, int NW = 4, int RD = kRing>
__device__ __forceinline__ void narrow_prefetch(f32x4 (&ring)[RD][NCB], int nk, const float* __restrict__ P, int Np,
                                                int col_off, int nblk, int wave) {
  const NarrowPart np = narrow_part<NW>(nk, col_off, nblk, wave);
  if (np.k_n > 0) mm_prefetch<NCB, 1, 3>(ring, np.k_n, P, Np, np.col, np.k_lo);
}
// requires narrow_prefetch(ring, same arguments)
template <int NRB, int NCB, int NW = 4, int RD = kRing>
__device__ __forceinline__ void narrow_layer_splitk(float* lds, int lda, int nk, const float* __restrict__ P, int Np,
                                                    int col_off, int nblk, int wave, f32x4 (&ring)[RD][NCB]) {
  const int lane = threadIdx.x & 63;
  const int parts = NW / nblk;
  const NarrowPart np = narrow_part<NW>(nk, col_off, nblk, wave);
  f32x4 t[NRB][1];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) t[rb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (np.k_n > 0) mm_run<NRB, 1, NCB, 1, 3>(lds, lda, np.k_n, P, Np, np.col, t, ring, np.k_lo);
  __syncthreads();  // all reads of the input activations are done
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds[(rb * 16 + (lane >> 4) * 4 + r) * lda + wave * 16 + (lane & 15)] = t[rb][0][r];
  __syncthreads();
  constexpr int BM = 16 * NRB;
  const int ncol = nblk * 16;
  float v[(BM * 32 + 64 * NW - 1) / (64 * NW)];
#pragma unroll
  for (int j = 0; j < (BM * 32 + 64 * NW - 1) / (64 * NW); ++j) {
    const int e = j * 64 * NW + (int)threadIdx.x;
    float sacc = 0.f;
    if (e < BM * ncol) {
      const int r = e / ncol, c = e - r * ncol;
      const int b = c >> 4, cc = c & 15;
      for (int q = 0; q < parts; ++q) sacc += lds[r * lda + (b * parts + q) * 16 + cc];
    }
    v[j] = sacc;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < (BM * 32 + 64 * NW - 1) / (64 * NW); ++j) {
    const int e = j * 64 * NW + (int)threadIdx.x;
    if (e < BM * ncol) {
      const int r = e / ncol, c = e - r * ncol;
      lds[r * lda + c] = v[j];
    }
  }
  __syncthreads();
}


// copy the LDS tile [BM][N] (stride lda) to global dst[(row0+r)*N + c].  NT = the workgroup's thread count when the
// caller knows it at compile time: `blockDim` is an s_load from the hidden block of the kernarg segment in every wave
// (a PCIe round trip where the runtime keeps kernel arguments in host memory); 0 = read it
template <int NT = 0>
__device__ __forceinline__ void tile_to_global(const float* lds, int lda, int BM, int N, float* __restrict__ dst,
                                               int row0, int rows) {
  const int tid = threadIdx.x;
  if ((N & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    const int n4 = N >> 2;
    for (int idx = tid; idx < BM * n4; idx += (NT ? NT : (int)blockDim.x)) {
      const int r = idx / n4, c4 = idx - r * n4;
      if (row0 + r < rows)
        *reinterpret_cast<f32x4*>(dst + (size_t)(row0 + r) * N + 4 * c4) =
            *reinterpret_cast<const f32x4*>(lds + r * lda + 4 * c4);
    }
  } else {
    for (int idx = tid; idx < BM * N; idx += (NT ? NT : (int)blockDim.x)) {
      const int r = idx / N, c = idx - r * N;
      if (row0 + r < rows) dst[(size_t)(row0 + r) * N + c] = lds[r * lda + c];
    }
  }
}

// activation (+ scale) of one wave's accumulators into the LDS tile (in place: next layer's input).  The bias is
// already in the accumulators (they are initialised with it), the scale and the k-padding select are compiled
// out when not needed: VALU instructions do not co-execute with the other waves' MFMAs on a SIMD
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every PMC pass), so each one here is paid in MFMA issue slots.
template <int NRB, int NCB, int ACT, bool SCALE, bool RAGGED>
__device__ __forceinline__ void fwd_epilogue_core(float* lds, int lda, const f32x4 (&acc)[NRB][NCB], int cb0, int cnt,
                                                  int N, float oscale, int lane) {
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    if (c < cnt) {
      const int col = (cb0 + c) * 16 + (lane & 15);
      const bool live = col < N;
      float* dst = lds + ((lane >> 4) * 4) * lda + col;
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = act_fwd(ACT, acc[rb][c][r]);
          if (SCALE) v *= oscale;
          if (RAGGED) v = live ? v : 0.f;  // zero the k-padding of the next layer
          dst[(rb * 16 + r) * lda] = v;
        }
      }
    }
  }
}
template <int NRB, int NCB, int ACT>
__device__ __forceinline__ void fwd_epilogue(float* lds, int lda, const f32x4 (&acc)[NRB][NCB], int cb0, int cnt, int N,
                                             float oscale, int lane) {
  if (oscale == 1.0f && (N & 15) == 0)
    fwd_epilogue_core<NRB, NCB, ACT, false, false>(lds, lda, acc, cb0, cnt, N, oscale, lane);
  else
    fwd_epilogue_core<NRB, NCB, ACT, true, true>(lds, lda, acc, cb0, cnt, N, oscale, lane);
}

// ---- row-local tails on the LDS-resident last tile (osrl_mlp_tail_t; the arithmetic of csrc/glue.hip's
// vae_latent_kernel / vae_latent_bwd_kernel, expression for expression) ------------------------------------------
constexpr float kTailLsMin = -4.0f, kTailLsMax = 15.0f;  // net.py:325 (== kVaeLsMin / kVaeLsMax of glue.hip)

// tile = the encoder output [BM][2L] (mean | log_std): z = mean + exp(clamp(log_std)) * eps
template <class TR, int NT = 0>
__device__ __forceinline__ void tail_vae_latent(const float* lds, int lda, int BM, int row0, int rows, TR t) {
  const int Lz = t.L;
  for (int idx = threadIdx.x; idx < BM * Lz; idx += (NT ? NT : (int)blockDim.x)) {
    const int r = idx / Lz, k = idx - r * Lz;
    const int gr = row0 + r;
    if (gr < rows) {
      const float mean = lds[r * lda + k];
      const float ls = fminf(fmaxf(lds[r * lda + Lz + k], kTailLsMin), kTailLsMax);
      const size_t i = (size_t)gr * Lz + k;
      t.out[i] = mean + expf(ls) * t.eps[i];
    }
  }
}

// tile = dL/dz [BM][L] (the decoder's dX slice): d/d(mean | log_std) of recon + beta KL through z = mean + sd * eps
template <class TR, int NT = 0>
__device__ __forceinline__ void tail_vae_latent_bwd(const float* lds, int lda, int BM, int row0, int rows, TR t) {
  const int Lz = t.L;
  for (int idx = threadIdx.x; idx < BM * Lz; idx += (NT ? NT : (int)blockDim.x)) {
    const int r = idx / Lz, k = idx - r * Lz;
    const int gr = row0 + r;
    if (gr < rows) {
      const size_t i = (size_t)gr * Lz + k;
      const float mean = t.head[(size_t)gr * 2 * Lz + k];
      const float lsr = t.head[(size_t)gr * 2 * Lz + Lz + k];
      const float ev = t.eps[i];
      const float g = lds[r * lda + k];
      const float sd = expf(fminf(fmaxf(lsr, kTailLsMin), kTailLsMax));
      const float c = t.beta * t.inv_rows_ / (float)Lz;
      t.out[(size_t)gr * 2 * Lz + k] = g + c * mean;
      const bool inside = lsr >= kTailLsMin && lsr <= kTailLsMax;
      t.out[(size_t)gr * 2 * Lz + Lz + k] = inside ? (g * ev + c * (sd - 1.0f / sd)) * sd : 0.f;
    }
  }
}

// tile = the actor trunk's output [BM][2 ad] (mu | log_std): the action draws of the squashed-Gaussian head
// (glue.hip gauss_head_kernel / gauss_ood_kernel, expression for expression) while the tile is in LDS
constexpr float kTailLogStdMin = -20.0f, kTailLogStdMax = 2.0f;  // net.py:148-149 (== kLogStdMin / kLogStdMax of glue.hip)
template <class TR, int NT = 0>
__device__ __forceinline__ void tail_gauss(const float* lds, int lda, int BM, int row0, int rows, TR t) {
  const int ad = t.L;
  const float max_a = t.max_action;
  const float* __restrict__ e1 = t.eps;
  const float* __restrict__ e2 = t.eps2;
  float* __restrict__ a1 = t.out;
  float* __restrict__ a2 = t.out2;
  float* __restrict__ th2 = t.tanh2;
  for (int idx = threadIdx.x; idx < BM * ad; idx += (NT ? NT : (int)blockDim.x)) {
    const int r = idx / ad, j = idx - r * ad;
    const int gr = row0 + r;
    if (gr < rows) {
      const float mu = lds[r * lda + j];
      const float ls = fminf(fmaxf(lds[r * lda + ad + j], kTailLogStdMin), kTailLogStdMax);
      const float sd = expf(ls);
      const size_t i = (size_t)gr * ad + j;
      if (e1) {
        const float u = mu + sd * e1[i];
        a1[i] = max_a * tanhf(u);
      }
      if (e2) {
        const float u = mu + sd * e2[i];
        const float th = tanhf(u);
        a2[i] = max_a * th;
        if (th2) th2[i] = th;
      }
    }
  }
  const float* __restrict__ eo = t.eps_ood;
  if (eo) {
    float* __restrict__ so = t.out_ood;
    const int ns = t.n_samples;
    for (int idx = threadIdx.x; idx < ns * BM * ad; idx += (NT ? NT : (int)blockDim.x)) {
      const int jr = idx / ad, k = idx - jr * ad;
      const int smp = jr / BM, r = jr - smp * BM;
      const int gr = row0 + r;
      if (gr < rows) {
        const float mu = lds[r * lda + k];
        const float ls = fminf(fmaxf(lds[r * lda + ad + k], kTailLogStdMin), kTailLogStdMax);
        const size_t i = ((size_t)smp * rows + gr) * ad + k;
        so[i] = mu + expf(ls) * eo[i];
      }
    }
  }
}

struct FwdArgs {
  osrl_mlp_t net;
  osrl_rows_t in;
  osrl_mlp_acts_t out;
  int32_t lda;
  osrl_mlp_tail_t tail;
};

// waves per SIMD the register allocator must leave room for (512 VGPRs / waves): the accumulators need
// NRB*NCB*4 registers and the weight ring STAGES*NCB*4; without the hint the allocator drifts to 170-200
// registers (AGPR copies of loop invariants) and one fewer workgroup fits per CU
#ifndef OSRL_WPS_8
#define OSRL_WPS_8 3  // <2,4>: 3 waves (<= 168 VGPRs); 4 spills into scratch inside the k-loop
#endif
constexpr int waves_per_simd(int nrb, int ncb) {
  return nrb * ncb >= 14 ? 2 : nrb * ncb == 8 ? OSRL_WPS_8 : nrb * ncb >= 4 ? 3 : 4;
}
// the backward kernel also holds the prefetched activations of the epilogue: one wave less
constexpr int waves_per_simd_bwd(int nrb, int ncb) { return nrb * ncb >= 7 ? 2 : 3; }

// AR = how the descriptor is reached: `const FwdArgs&` (the by-value kernel argument, i.e. the kernarg segment) or
// `const OSRL_CAS FwdArgs&` (a device-resident block, argmem.h); same member accesses, same scalar loads
template <int NRB, int NCB, int NW, class AR>
__device__ __forceinline__ void mlp_fwd_body(AR a, const int e, const int tile) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 16 * NRB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = tile * BM;
  const int rows = a.in.rows, lda = a.lda;
  const int L = a.net.n_layers;
#if OSRL_CHAIN_PRIO > 0
  // latency-chain launches (training rows) outrank the N*B-row filler launches they share CUs with
  if (rows <= OSRL_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
#endif

  WG_LOG(0);
  PHASE_STAMP(0);
  // first weight loads of layer l (issued before the previous layer's epilogue / before the input is staged)
  f32x4 ring[ring_depth<NW>()][NCB];
  auto begin_layer = [&](int l) {
    const int K = a.net.dims[l], N = a.net.dims[l + 1];
    const int nblk = (N + 15) >> 4, nk = round16(K) >> 4;
    if (nblk <= 2 && nk >= 4 && lda >= 16 * NW) {
      narrow_prefetch<NCB, NW>(ring, nk, a.net.Wf[e][l], round16(N), 0, nblk, wave);
    } else {
      int cb0, cnt;
      wave_blocks<NW>(nblk, wave, &cb0, &cnt);
      layer_prefetch<NRB, NCB>(ring, nk, a.net.Wf[e][l], round16(N), cb0 * 16, cnt);
    }
  };
  begin_layer(0);
  {  // stage cat(src0[map0(r)], src1[map1(r)]) zero-padded to a multiple of 16 columns.
     // 16 lanes walk one row (64-byte segments), 16 rows per pass; every load of a pass is issued
     // before the first LDS store (branch-free clamped addresses), so the latencies overlap.
    const int K0 = a.net.dims[0], K0p = round16(K0);
    const int d0 = a.in.d0, d1 = a.in.d1;
    const int cl = tid & 15, rl = tid >> 4;
    const float* __restrict__ s0 = a.in.src0;
    const float* __restrict__ s1 = a.in.src1 ? a.in.src1 : a.in.src0;
#pragma unroll 1
    for (int r = rl; r < BM; r += 4 * NW) {
      const int gr = row0 + r;
      const bool rok = gr < rows;
      const int grc = rok ? gr : rows - 1;
      const float* p0 = s0 + (size_t)map_row(grc, a.in.map0, a.in.div0) * d0;
      const float* p1 = s1 + (size_t)map_row(grc, a.in.map1, a.in.div1) * d1 - d0;
      for (int cbase = 0; cbase < K0p; cbase += 16 * 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = cbase + j * 16 + cl;
          const bool in0 = c < d0, ok = rok && c < d0 + d1;
          const float* p = in0 ? p0 + c : p1 + c;
          v[j] = *(ok ? p : s0);
          v[j] = ok ? v[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = cbase + j * 16 + cl;
          if (c < K0p) lds[r * lda + c] = v[j];
        }
      }
    }
    __syncthreads();
    if (e == 0 && a.out.x) tile_to_global<64 * NW>(lds, lda, BM, K0, a.out.x, row0, rows);
  }
  PHASE_STAMP(1);
  constexpr bool kWarm = OSRL_L2_WARM && NW == 8;
  float warm[OSRL_MAX_LAYERS - 1][kWarmLines];
  if (kWarm) {
    for (int l = 1; l < L; ++l)
      l2_warm<64 * NW>(a.net.Wf[e][l], round16(a.net.dims[l]) * round16(a.net.dims[l + 1]), warm[l - 1]);
  }

  for (int l = 0; l < L; ++l) {
    const int K = a.net.dims[l], N = a.net.dims[l + 1];
    const int nblk = (N + 15) >> 4;
    const float* __restrict__ bias = a.net.b[e][l];
    const int act = a.net.acts[l];
    const float oscale = (l == L - 1) ? a.net.out_scale : 1.0f;
    const int nk = round16(K) >> 4;
    if (nblk <= 2 && nk >= 4 && lda >= 16 * NW) {
      // narrow layer: split K over the 4 waves, then bias/activation on the summed tile in LDS
      narrow_layer_splitk<NRB, NCB, NW>(lds, lda, nk, a.net.Wf[e][l], round16(N), 0, nblk, wave, ring);
      if (l + 1 < L) begin_layer(l + 1);
      PHASE_STAMP(2 + 4 * l);
      PHASE_STAMP(3 + 4 * l);
      const int ncol = nblk * 16;
      for (int idx = tid; idx < BM * ncol; idx += 64 * NW) {
        const int r = idx / ncol, c = idx - r * ncol;
        const float v = c < N ? act_fwd(act, lds[r * lda + c] + bias[c]) * oscale : 0.f;
        lds[r * lda + c] = v;
      }
    } else {
      int cb0, cnt;
      wave_blocks<NW>(nblk, wave, &cb0, &cnt);  // cnt may be 0 for idle waves
      float bv[NCB];  // bias fetched before the k-loop so its latency hides behind the MFMAs
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const int col = (cb0 + c) * 16 + (lane & 15);
        const bool ok = c < cnt && col < N;
        bv[c] = bias[ok ? col : 0];
        bv[c] = ok ? bv[c] : 0.f;
      }
      f32x4 acc[NRB][NCB];  // start from the bias: no add in the epilogue
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[rb][c] = f32x4{bv[c], bv[c], bv[c], bv[c]};
      if (cnt > 0) layer_run<NRB, NCB>(lds, lda, nk, a.net.Wf[e][l], round16(N), cb0 * 16, cnt, acc, ring);
      if (l + 1 < L) begin_layer(l + 1);  // the ring is free again: overlap the next layer's first fetch with the epilogue
      PHASE_STAMP(2 + 4 * l);
      __syncthreads();  // every wave finished reading the previous activations
      PHASE_STAMP(3 + 4 * l);
      // the activation switch is hoisted out of the element loop: with a per-element runtime switch every one of
      // the NRB*NCB*4 values jumped over an inlined tanhf body (4 scalar branches each, ~14 KB of sparse code):
      // measured 10-11k cycles per epilogue vs 17k for the whole first-layer k-loop (tools/mlp_phase.hip)
      if (act == OSRL_ACT_RELU)
        fwd_epilogue<NRB, NCB, OSRL_ACT_RELU>(lds, lda, acc, cb0, cnt, N, oscale, lane);
      else if (act == OSRL_ACT_TANH)
        fwd_epilogue<NRB, NCB, OSRL_ACT_TANH>(lds, lda, acc, cb0, cnt, N, oscale, lane);
      else
        fwd_epilogue<NRB, NCB, OSRL_ACT_ID>(lds, lda, acc, cb0, cnt, N, oscale, lane);
    }
    PHASE_STAMP(4 + 4 * l);
    __syncthreads();
    float* save = a.out.h[e][l];
    if (save) tile_to_global<64 * NW>(lds, lda, BM, N, save, row0, rows);
    PHASE_STAMP(5 + 4 * l);
    if (kWarm && l == 0) {
      for (int w = 1; w < L; ++w) l2_warm_done(warm[w - 1]);
    }
  }
  // the net's output tile [BM][dims[L]] is still in LDS (nothing wrote it since the last barrier)
  if (a.tail.kind == OSRL_TAIL_VAE_LATENT && e == 0) tail_vae_latent<decltype((a.tail)), 64 * NW>(lds, lda, BM, row0, rows, a.tail);
  if (a.tail.kind == OSRL_TAIL_GAUSS && e == 0) tail_gauss<decltype((a.tail)), 64 * NW>(lds, lda, BM, row0, rows, a.tail);
  WG_LOG(1);
}

template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd(NRB, NCB))) void mlp_fwd_kernel(const FwdArgs a) {
  mlp_fwd_body<NRB, NCB, NW, const FwdArgs&>(a, blockIdx.y, blockIdx.x);
}
// the same kernel with its descriptor in device memory (argmem.h)
template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd(NRB, NCB))) void mlp_fwd_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(5, p);
  mlp_fwd_body<NRB, NCB, NW, const OSRL_CAS FwdArgs&>(*(const OSRL_CAS FwdArgs*)p, blockIdx.y, blockIdx.x);
}

// gridDim.x capped below the tile count (osrl_mlp_t::wg_cap): a big launch that is NOT on the critical path then
// leaves CU slots and MFMA issue to the latency-critical 128-workgroup launches it runs beside.  (A separate
// kernel: the tile loop costs the one-tile kernel ~12 more spilled registers and 3 % of its speed.)
template <int NRB, int NCB>
__global__ __launch_bounds__(256, waves_per_simd(NRB, NCB)) void mlp_fwd_loop_kernel(const FwdArgs a) {
  const int n_tiles = (a.in.rows + 16 * NRB - 1) / (16 * NRB);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    mlp_fwd_body<NRB, NCB, 4, const FwdArgs&>(a, blockIdx.y, tile);
    __syncthreads();  // the LDS tile is reused
  }
}

// Two independent forward problems (different networks / inputs, same tile shape) in ONE launch: the 2048-row
// training launches of a step are 128 row tiles x 1-4 nets each, i.e. at most one workgroup per CU and a serial
// latency chain inside it; pairing two of them fills the idle CUs and removes a launch from the critical path.
template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd(NRB, NCB))) void mlp_fwd2_kernel(
    const FwdArgs a0, const FwdArgs a1, int nets0, int tiles0, int tiles1) {
  if ((int)blockIdx.y < nets0) {
    if ((int)blockIdx.x >= tiles0) return;  // whole workgroup leaves before any barrier
    mlp_fwd_body<NRB, NCB, NW, const FwdArgs&>(a0, blockIdx.y, blockIdx.x);
  } else {
    if ((int)blockIdx.x >= tiles1) return;
    mlp_fwd_body<NRB, NCB, NW, const FwdArgs&>(a1, blockIdx.y - nets0, blockIdx.x);
  }
}
struct Fwd2Args {  // device-resident form of the pair launch (argmem.h)
  FwdArgs a0, a1;
  int32_t nets0, tiles0, tiles1, pad_;
};
template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd(NRB, NCB))) void mlp_fwd2_kernel_p(
    const void* p) {
  OSRL_TRACE_BEGIN(6, p);
  const OSRL_CAS Fwd2Args& f = *(const OSRL_CAS Fwd2Args*)p;
  if ((int)blockIdx.y < f.nets0) {
    if ((int)blockIdx.x >= f.tiles0) return;
    mlp_fwd_body<NRB, NCB, NW, const OSRL_CAS FwdArgs&>(f.a0, blockIdx.y, blockIdx.x);
  } else {
    if ((int)blockIdx.x >= f.tiles1) return;
    mlp_fwd_body<NRB, NCB, NW, const OSRL_CAS FwdArgs&>(f.a1, blockIdx.y - f.nets0, blockIdx.x);
  }
}

struct BwdArgs {
  osrl_mlp_t net;
  osrl_mlp_acts_t saved;
  osrl_mlp_grads_t g;
  int32_t rows, lda;
  osrl_mlp_tail_t tail;
  osrl_mlp_seed_t seed;
};

// ---- loss seeds (osrl_mlp_seed_t): dL/d(output) of the rows a tile stages, from forward outputs of the same rows --
// the expressions of csrc/glue.hip's loss kernels, term for term (vae_loss_body, cpq_critic_loss_body,
// cpq_cost_loss_body, cpq_actor_loss_kernel, gauss_head_bwd_kernel), so that the gradient has their bits.
constexpr float kSeedLogStdMin = -20.0f, kSeedLogStdMax = 2.0f;  // net.py:148-149 (== kLogStdMin / kLogStdMax of glue.hip)
constexpr int kSeedEns = 4;                                       // == kEns of glue.hip (host-checked)
__device__ __forceinline__ void seed_members(const float* __restrict__ q, int n, int stride, int i, float (&v)[kSeedEns]) {
#pragma unroll
  for (int e = 0; e < kSeedEns; ++e) v[e] = q[(size_t)(e < n ? e : 0) * stride + i];  // members past n re-read member 0
}
__device__ __forceinline__ float seed_min(const float* __restrict__ q, int n, int stride, int i) {
  float v[kSeedEns];
  seed_members(q, n, stride, i, v);
  float m = v[0];
#pragma unroll
  for (int e = 1; e < kSeedEns; ++e) m = e < n ? fminf(m, v[e]) : m;
  return m;
}
__device__ __forceinline__ float seed_kl_elem(float mean, float ls_raw) {
  const float sd = expf(fminf(fmaxf(ls_raw, kTailLsMin), kTailLsMax));
  return -0.5f * (1.0f + logf(sd * sd) - mean * mean - sd * sd);
}
// element (r, c) of net e's dY; yv = that element of the net's output, yrow = the output row; l0 collects the
// statistic's terms of valid elements
template <class SR>
__device__ __forceinline__ float seed_dy(SR s, const int e, const int r, const int c, const int NL, const int rows,
                                         const float yv, const float* __restrict__ yrow, const bool ok, float& l0) {
  const int kind = s.kind;
  if (kind == OSRL_SEED_MSE) {
    const float d = yv - s.x0[(size_t)r * NL + c];
    if (ok) l0 += d * d;
    return 2.0f * d * s.scale;
  }
  if (kind == OSRL_SEED_CPQ_CRITIC) {
    const float qt = seed_min(s.a, s.n_a, rows, r);
    const float qct = seed_min(s.b, s.n_b, rows, r);
    const float backup = s.x0[r] + s.gamma * (1.0f - s.x1[r]) * (qct <= s.thres ? 1.0f : 0.0f) * qt;  // cpq.py:145-146
    const float d = yv - backup;
    if (ok) l0 += d * d;
    return 2.0f * d * s.scale;
  }
  if (kind == OSRL_SEED_CPQ_COST) {
    const float backup = s.x0[r] + s.gamma * seed_min(s.a, s.n_a, rows, r);  // cpq.py:161
    const float d = yv - backup;
    if (ok) l0 += d * d;
    return 2.0f * d * s.scale;
  }
  if (kind == OSRL_SEED_CPQ_ACTOR) {
    float qv[kSeedEns];
    seed_members(s.a, s.n_a, rows, r, qv);
    float qm = qv[0];
    int am = 0;
#pragma unroll
    for (int k = 1; k < kSeedEns; ++k) {
      const bool lt = k < s.n_a && qv[k] < qm;
      qm = lt ? qv[k] : qm;
      am = lt ? k : am;
    }
    const float mask = seed_min(s.b, s.n_b, rows, r) <= s.thres ? 1.0f : 0.0f;
    if (ok && e == 0) l0 -= mask * qm;
    return e == am ? -mask * s.scale : 0.f;
  }
  if (kind == OSRL_SEED_BCQ_CRITIC) {  // glue.hip bcq_critic_loss_kernel, term for term (bcql.py:138-150)
    const int ns = s.n_samples, nr = rows * ns;
    const float lmbda = s.thres;
    const float* __restrict__ q2p = s.a + (size_t)s.n_a * nr;
    float best = -INFINITY;
    for (int j = 0; j < ns; ++j) {
      const int i = r * ns + j;
      float q1 = s.a[i];
      for (int k = 1; k < s.n_a; ++k) q1 = fminf(q1, s.a[(size_t)k * nr + i]);
      float q2 = q2p[i];
      for (int k = 1; k < s.n_b; ++k) q2 = fminf(q2, q2p[(size_t)k * nr + i]);
      const float v = lmbda * fminf(q1, q2) + (1.0f - lmbda) * fmaxf(q1, q2);
      best = fmaxf(best, v);
    }
    const float nd = s.x1 ? (1.0f - s.x1[r]) : 1.0f;
    const float backup = s.x0[r] + s.gamma * nd * best;
    const float d = yv - backup;
    if (ok) l0 += d * d;
    return 2.0f * d * s.scale;
  }
  // OSRL_SEED_GAUSS_HEAD: yrow = (mu | log_std) of the row, NL = 2 ad
  const int ad = NL >> 1;
  const int j = c < ad ? c : c - ad;
  const float t = s.tanh_u[(size_t)r * ad + j];
  const float lsr = yrow[ad + j];
  const float ev = s.eps[(size_t)r * ad + j];
  float dv[kSeedEns];
#pragma unroll
  for (int k = 0; k < kSeedEns; ++k) dv[k] = s.a[((size_t)(k < s.n_a ? k : 0) * rows + r) * ad + j];
  float da = 0.f;
#pragma unroll
  for (int k = 0; k < kSeedEns; ++k) da += k < s.n_a ? dv[k] : 0.f;
  const float du = da * s.max_action * (1.0f - t * t);
  const float ls = fminf(fmaxf(lsr, kSeedLogStdMin), kSeedLogStdMax);
  const bool inside = lsr >= kSeedLogStdMin && lsr <= kSeedLogStdMax;
  return c < ad ? du : (inside ? du * ev * expf(ls) : 0.f);
}
// device-coherent words for the statistic's partials (workgroups on different XCDs exchange them inside the launch)
__device__ __forceinline__ void coh_put(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float coh_get(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NRB, int NCB, int NW, class AR>
__device__ __forceinline__ void mlp_bwd_dz_body(AR a, const int e, const int tile) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 16 * NRB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = tile * BM;
  const int rows = a.rows, lda = a.lda;
  const int L = a.net.n_layers;
  __shared__ float s_seed[2][8];  // per-wave partials of a seeded launch's statistic
  __shared__ int s_seed_last;
#if OSRL_CHAIN_PRIO > 0
  if (rows <= OSRL_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
#endif

  BWD_STAMP(0);
  // steps: l = L-1 .. 1 (dH_{l-1} = dZ_l W_l), then step 0 = the dX slice; begin_step issues a step's first weight loads
  f32x4 ring[ring_depth<NW>()][NCB];
  auto begin_step = [&](int l) {
    if (l >= 1) {
      const int K = a.net.dims[l + 1], N = a.net.dims[l];
      int cb0, cnt;
      wave_blocks<NW>((N + 15) >> 4, wave, &cb0, &cnt);
      layer_prefetch<NRB, NCB>(ring, round16(K) >> 4, a.net.Wb[e][l], round16(N) + 16, cb0 * 16, cnt);
    } else if (a.g.dx[e]) {
      const int nblk = (a.g.dx_cols + 15) >> 4, nk = round16(a.net.dims[1]) >> 4;
      const int Npb = round16(a.net.dims[0]) + 16;
      if (nblk <= 2 && nk >= 4 && lda >= 16 * NW) {
        narrow_prefetch<NCB, NW>(ring, nk, a.net.Wb[e][0], Npb, a.g.dx_col0, nblk, wave);
      } else {
        int cb0, cnt;
        wave_blocks<NW>(nblk, wave, &cb0, &cnt);
        layer_prefetch<NRB, NCB>(ring, nk, a.net.Wb[e][0], Npb, a.g.dx_col0 + cb0 * 16, cnt);
      }
    }
  };
  begin_step(L - 1);
  {  // dZ_{L-1} = dY * out_scale * act'(Y / out_scale)
    const float oscale = a.net.out_scale, inv_oscale = 1.0f / a.net.out_scale;
    const int NL = a.net.dims[L], NLp = round16(NL);
    const float* __restrict__ dy = a.g.dy[e];
    const float* __restrict__ y = a.saved.h[e][L - 1];
    const int act = a.net.acts[L - 1];
    const float* __restrict__ ysrc = act != OSRL_ACT_ID ? y : dy;
    if (a.seed.kind != OSRL_SEED_NONE) {
      // the launch computes its own dY (osrl_mlp_seed_t) + this tile's partial of the logged statistic
      float l0 = 0.f, l1 = 0.f;
      for (int idx = tid; idx < BM * NLp; idx += 64 * NW) {
        const int r = idx / NLp, c = idx - r * NLp;
        const int gr = row0 + r;
        const bool ok = gr < rows && c < NL;
        const int grc = gr < rows ? gr : rows - 1, cc = c < NL ? c : 0;
        const float* __restrict__ yrow = y + (size_t)grc * NL;
        const float yv = yrow[cc];
        const float dyv = seed_dy<decltype((a.seed))>(a.seed, e, grc, cc, NL, rows, yv, yrow, ok, l0);
        float v = dyv * oscale;
        if (act != OSRL_ACT_ID) v *= act_bwd(act, yv * inv_oscale);
        lds[r * lda + c] = ok ? v : 0.f;
      }
      if (a.seed.kl_head && e == 0) {  // the KL term of the VAE statistic (vae_loss_body) over this tile's rows
        const int Lz = a.seed.kl_L;
        const float* __restrict__ hd = a.seed.kl_head;
        for (int idx = tid; idx < BM * Lz; idx += 64 * NW) {
          const int r = idx / Lz, k = idx - r * Lz;
          const int gr = row0 + r;
          const int grc = gr < rows ? gr : rows - 1;
          const float kv = seed_kl_elem(hd[(size_t)grc * 2 * Lz + k], hd[(size_t)grc * 2 * Lz + Lz + k]);
          l1 += gr < rows ? kv : 0.f;
        }
      }
      if (a.seed.partials) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          l0 += __shfl_xor(l0, o);
          l1 += __shfl_xor(l1, o);
        }
        if (lane == 0) {
          s_seed[0][wave] = l0;
          s_seed[1][wave] = l1;
        }
      }
    } else
    for (int idx = tid; idx < BM * NLp; idx += 64 * NW) {
      const int r = idx / NLp, c = idx - r * NLp;
      const int gr = row0 + r;
      // branch-free (clamped address + select): dy and y are requested together; as "if (ok) { load dy; load y }"
      // each load sat in its own exec-masked block with a full s_waitcnt behind it
      const bool ok = gr < rows && c < NL;
      const size_t off = ok ? (size_t)gr * NL + c : 0;
      const float dyv = dy[off];
      float yv = ysrc[off];  // (the identity activation re-reads dy: no branch between the two requests)
      asm volatile("" : "+v"(yv));  // keeps the compiler from sinking this load into the activation branch below
      float v = dyv * oscale;
      if (act != OSRL_ACT_ID) v *= act_bwd(act, yv * inv_oscale);
      lds[r * lda + c] = ok ? v : 0.f;
    }
    __syncthreads();
    if (a.seed.kind != OSRL_SEED_NONE && a.seed.partials && tid == 0) {  // this tile's partials, waves in order
      float t0 = 0.f, t1 = 0.f;
      for (int w = 0; w < NW; ++w) {
        t0 += s_seed[0][w];
        t1 += s_seed[1][w];
      }
      float* pp = a.seed.partials + 2 * ((size_t)e * ((rows + BM - 1) / BM) + tile);
      coh_put(pp, t0);
      coh_put(pp + 1, t1);
    }
    if (a.g.dz[e][L - 1]) tile_to_global<64 * NW>(lds, lda, BM, NL, a.g.dz[e][L - 1], row0, rows);
  }
  BWD_STAMP(1);  // dZ_{L-1} staged
  constexpr bool kWarm = OSRL_L2_WARM && NW == 8;
  float warm[OSRL_MAX_LAYERS - 1][kWarmLines];
  if (kWarm) {  // the later steps' W^T packs (step L-1's first loads are already in flight)
    for (int l = L - 2; l >= (a.g.dx[e] ? 0 : 1); --l)
      l2_warm<64 * NW>(a.net.Wb[e][l], round16(a.net.dims[l + 1]) * (round16(a.net.dims[l]) + 16), warm[l]);
  }

  for (int l = L - 1; l >= 1; --l) {
    // dH_{l-1}[r][i] = sum_o dZ_l[r][o] * W_l[o][i]   (K = dims[l+1] (o), N = dims[l] (i))
    const int K = a.net.dims[l + 1], N = a.net.dims[l];
    const int nblk = (N + 15) >> 4;
    int cb0, cnt;
    wave_blocks<NW>(nblk, wave, &cb0, &cnt);
    const float* __restrict__ h = a.saved.h[e][l - 1];
    const int act = a.net.acts[l - 1];
    // activation outputs needed by act'(.) in the epilogue: fetched BEFORE the k-loop (small tiles)
    constexpr bool PREFETCH_H = NRB * NCB * 4 <= 32;
    float hv[PREFETCH_H ? NRB : 1][PREFETCH_H ? NCB : 1][4];
    if (PREFETCH_H) {
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        const int col = (cb0 + c) * 16 + (lane & 15);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int gr = row0 + rb * 16 + (lane >> 4) * 4 + r;
            const bool ok = c < cnt && col < N && gr < rows;
            const float x = h[ok ? (size_t)gr * N + col : 0];
            hv[PREFETCH_H ? rb : 0][PREFETCH_H ? c : 0][r] = ok ? x : 0.f;
          }
        }
      }
    }
    f32x4 acc[NRB][NCB];
    zero_acc<NRB, NCB>(acc);
    // packed W^T: contraction over the layer's outputs (K), columns = the layer's inputs (N) (+16 pad)
    if (cnt > 0) layer_run<NRB, NCB>(lds, lda, round16(K) >> 4, a.net.Wb[e][l], round16(N) + 16, cb0 * 16, cnt, acc, ring);
    begin_step(l - 1);
    BWD_STAMP(2 + 3 * (L - 1 - l));  // k-loop done
    __syncthreads();
    BWD_STAMP(3 + 3 * (L - 1 - l));  // barrier
    // activation switch hoisted out of the element loop (see fwd_epilogue)
    auto epilogue = [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        if (c < cnt) {
          const int col = (cb0 + c) * 16 + (lane & 15);
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = rb * 16 + (lane >> 4) * 4 + r;
              const int gr = row0 + row;
              float v = 0.f;
              if (PREFETCH_H) {
                v = (col < N && gr < rows) ? acc[rb][c][r] * act_bwd(ACT, hv[PREFETCH_H ? rb : 0][PREFETCH_H ? c : 0][r]) : 0.f;
              } else if (col < N && gr < rows) {
                v = acc[rb][c][r] * act_bwd(ACT, h[(size_t)gr * N + col]);
              }
              lds[row * lda + col] = v;
            }
          }
        }
      }
    };
    if (act == OSRL_ACT_RELU)
      epilogue(std::integral_constant<int, OSRL_ACT_RELU>{});
    else if (act == OSRL_ACT_TANH)
      epilogue(std::integral_constant<int, OSRL_ACT_TANH>{});
    else
      epilogue(std::integral_constant<int, OSRL_ACT_ID>{});
    __syncthreads();
    if (a.g.dz[e][l - 1]) tile_to_global<64 * NW>(lds, lda, BM, N, a.g.dz[e][l - 1], row0, rows);
    BWD_STAMP(4 + 3 * (L - 1 - l));  // epilogue + barrier + dZ stored
    if (kWarm && l == L - 1) {
      for (int w = L - 2; w >= (a.g.dx[e] ? 0 : 1); --w) l2_warm_done(warm[w]);
    }
  }

  if (a.g.dx[e]) {
    // dX[:, c0:c0+nc] = dZ_0 * W_0[:, c0:c0+nc]
    const int K = a.net.dims[1], nc = a.g.dx_cols;
    const int nblk = (nc + 15) >> 4;
    const int nk = round16(K) >> 4;
    const int Npb = round16(a.net.dims[0]) + 16;
    float* __restrict__ dx = a.g.dx[e];
    if (nblk <= 2 && nk >= 4 && lda >= 16 * NW) {
      narrow_layer_splitk<NRB, NCB, NW>(lds, lda, nk, a.net.Wb[e][0], Npb, a.g.dx_col0, nblk, wave, ring);
      tile_to_global<64 * NW>(lds, lda, BM, nc, dx, row0, rows);
      // (the host fuses the tail only when this branch is the one taken: bwd_tail_fusable)
      if (a.tail.kind == OSRL_TAIL_VAE_LATENT_BWD && e == 0) tail_vae_latent_bwd<decltype((a.tail)), 64 * NW>(lds, lda, BM, row0, rows, a.tail);
    } else {
      int cb0, cnt;
      wave_blocks<NW>(nblk, wave, &cb0, &cnt);
      f32x4 acc[NRB][NCB];
      zero_acc<NRB, NCB>(acc);
      if (cnt > 0) layer_run<NRB, NCB>(lds, lda, nk, a.net.Wb[e][0], Npb, a.g.dx_col0 + cb0 * 16, cnt, acc, ring);
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        if (c < cnt) {
          const int col = (cb0 + c) * 16 + (lane & 15);
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int gr = row0 + rb * 16 + (lane >> 4) * 4 + r;
              if (col < nc && gr < rows) dx[(size_t)gr * nc + col] = acc[rb][c][r];
            }
          }
        }
      }
    }
  }
  if (a.seed.kind != OSRL_SEED_NONE && a.seed.partials) {
    // the logged statistic: the last workgroup to get here sums every tile's partials in a fixed order (each lane a
    // contiguous chunk, lanes by the butterfly, waves in order).  Wait-free: a workgroup is the last one or leaves.
    const int n_part = a.net.n_nets * ((rows + BM - 1) / BM);
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's partials are acknowledged
      const unsigned seen = __hip_atomic_fetch_add(a.seed.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = seen == (unsigned)n_part - 1;
      if (last) __hip_atomic_store(a.seed.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_seed_last = last;
    }
    __syncthreads();
    if (s_seed_last) {
      const int per = (n_part + 64 * NW - 1) / (64 * NW);
      float t0 = 0.f, t1 = 0.f;
      const float* pp = a.seed.partials;
      for (int i = tid * per; i < (tid + 1) * per && i < n_part; ++i) {
        t0 += coh_get(pp + 2 * i);
        t1 += coh_get(pp + 2 * i + 1);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        t0 += __shfl_xor(t0, o);
        t1 += __shfl_xor(t1, o);
      }
      if (lane == 0) {
        s_seed[0][wave] = t0;
        s_seed[1][wave] = t1;
      }
      __syncthreads();
      if (tid == 0 && a.seed.stat) {
        float u0 = 0.f, u1 = 0.f;
        for (int w = 0; w < NW; ++w) {
          u0 += s_seed[0][w];
          u1 += s_seed[1][w];
        }
        a.seed.stat[0] = u0 * a.seed.stat_scale + a.seed.kl_beta * (u1 * a.seed.stat_scale2);
      }
    }
  }
}
template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd_bwd(NRB, NCB))) void mlp_bwd_dz_kernel(const BwdArgs a) {
  mlp_bwd_dz_body<NRB, NCB, NW, const BwdArgs&>(a, blockIdx.y, blockIdx.x);
}
template <int NRB, int NCB, int NW = 4>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? (NCB <= 2 ? 4 : 2) : waves_per_simd_bwd(NRB, NCB))) void mlp_bwd_dz_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(7, p);
  mlp_bwd_dz_body<NRB, NCB, NW, const OSRL_CAS BwdArgs&>(*(const OSRL_CAS BwdArgs*)p, blockIdx.y, blockIdx.x);
}

// ---- general linear layer  Y[M, N] = A[M, K] * P (+ bias) (+ resid)  ------------------------------
// The transformer-sized sibling of mlp_fwd_kernel (CDT: QKV / out-proj / MLP projections and their
// dX GEMMs, osrl/common/net.py:406-415,422-441): one launch = one GEMM, K up to 1024, N unbounded via
// column groups of 16*4*NCB columns on blockIdx.y.  A is staged through LDS once per workgroup, P is
// a packed weight (forward pack for y = x W^T, backward pack for dx = dy W), same MFMA core.
struct LinArgs {
  const float* A;
  const float* P;
  const float* bias;
  const float* resid;
  float* Y;
  int64_t lda_g, ldr, ldy;
  int32_t M, K, N, Np, col0, lda;
};

template <int NRB, int NCB>
__global__ __launch_bounds__(256) void linear_kernel(const LinArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 16 * NRB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * BM, M = a.M, K = a.K, N = a.N, lda = a.lda;
  const int Kp = round16(K);
  const int nk = Kp >> 4;
  const int nblk_tot = (N + 15) >> 4;
  constexpr int GB = 4 * NCB;  // column blocks per workgroup
  const int gb0 = blockIdx.y * GB;
  int nblk = nblk_tot - gb0;
  nblk = nblk > GB ? GB : nblk;
  const bool narrow = nblk_tot <= 2 && nk >= 4 && lda >= 64;
  int cb0 = 0, cnt = 0;
  f32x4 ring[kRing][NCB];
  if (narrow) {  // first weight loads go out before the A tile is staged
    narrow_prefetch<NCB>(ring, nk, a.P, a.Np, a.col0, nblk_tot, wave);
  } else {
    wave_blocks(nblk, wave, &cb0, &cnt);
    layer_prefetch<NRB, NCB>(ring, nk, a.P, a.Np, a.col0 + (gb0 + cb0) * 16, cnt);
  }
  {  // stage A[row0 : row0+BM, 0:K] zero padded; 16 lanes per row, float4 when the rows are 16-B aligned
    const int cl = tid & 15, rl = tid >> 4;
    const bool vec = ((K & 3) == 0) && ((a.lda_g & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
#pragma unroll 1
    for (int r = rl; r < BM; r += 16) {
      const int gr = row0 + r;
      const bool rok = gr < M;
      const float* __restrict__ src = a.A + (size_t)(rok ? gr : M - 1) * a.lda_g;
      if (vec) {
        for (int cbase = 0; cbase < Kp; cbase += 64 * 4) {
          f32x4 v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = cbase + j * 64 + cl * 4;
            const bool ok = rok && c < K;
            v[j] = *reinterpret_cast<const f32x4*>(src + (ok ? c : 0));
            if (!ok) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = cbase + j * 64 + cl * 4;
            if (c < Kp) *reinterpret_cast<f32x4*>(lds + r * lda + c) = v[j];
          }
        }
      } else {
        for (int cbase = 0; cbase < Kp; cbase += 16 * 8) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = cbase + j * 16 + cl;
            const bool ok = rok && c < K;
            v[j] = src[ok ? c : 0];
            v[j] = ok ? v[j] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = cbase + j * 16 + cl;
            if (c < Kp) lds[r * lda + c] = v[j];
          }
        }
      }
    }
    __syncthreads();
  }
  if (narrow) {
    narrow_layer_splitk<NRB, NCB>(lds, lda, nk, a.P, a.Np, a.col0, nblk_tot, wave, ring);
  } else {
    f32x4 acc[NRB][NCB];
    zero_acc<NRB, NCB>(acc);
    if (cnt > 0) layer_run<NRB, NCB>(lds, lda, nk, a.P, a.Np, a.col0 + (gb0 + cb0) * 16, cnt, acc, ring);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      if (c < cnt) {
        const int col = (cb0 + c) * 16 + (lane & 15);  // column inside this group's LDS tile
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) lds[(rb * 16 + (lane >> 4) * 4 + r) * lda + col] = acc[rb][c][r];
      }
    }
    __syncthreads();
  }
  // copy out: + bias + residual, coalesced
  const int ncols = (nblk * 16 < N - gb0 * 16) ? nblk * 16 : N - gb0 * 16;
  const int gcol0 = gb0 * 16;
  for (int idx = tid; idx < BM * ncols; idx += 256) {
    const int r = idx / ncols, c = idx - r * ncols;
    const int gr = row0 + r;
    if (gr < M) {
      float v = lds[r * lda + c];
      if (a.bias) v += a.bias[gcol0 + c];
      if (a.resid) v += a.resid[(size_t)gr * a.ldr + gcol0 + c];
      a.Y[(size_t)gr * a.ldy + gcol0 + c] = v;
    }
  }
}

// ---- weight packing ------------------------------------------------------------------------------
// ---- big-M linear layer: both operands staged through LDS ------------------------------------------------
// linear_kernel keeps the whole [BM, K] activation tile in LDS and streams the packed weights from L2 into
// registers per wave; at CDT sizes (M = 81920 tokens, K, N up to 1024) that is the L2-streaming-bound regime of
// mlp_fwd_kernel (~58 % of the fp32 roof).  This kernel is the classic LDS-tiled GEMM instead:
// one workgroup = 8 waves = 2 row groups x 4 column groups on a 128-row x 256-column output tile; per 16-deep k-step
// the [128 x 16] slice of A and the [16 x 256] slab of the packed weights are copied global -> registers -> LDS
// (double buffered, one barrier per k-step), every wave reads its fragments with ds_read_b128 and issues 64 MFMAs.
// L2 -> CU traffic per FLOP is 4x lower than with 32-row tiles.  Requires K % 16 == 0, N % 256 == 0 (per launch
// column group), 16-byte aligned A rows.
// (Round 2, measured and removed: (i) GELU fused into this kernel's epilogue -- forward writing pre-activation and
// gelu(pre-activation), backward multiplying the dX of mlp.2 by gelu'(hpre): the 64-byte-segment epilogue with erff /
// expf per element costs what the separate streaming GELU passes cost (CDT step 18.75 -> 18.84 ms); (ii) a
// one-wave-per-SIMD variant in the style of mlp_fwd_nb_kernel (80-row slice of A resident in LDS, 4 column blocks per
// wave): 380 us per call vs 311 us here -- with 16 k-steps per column group / K chunk the exposed stage-in, loop start
// and 80-store epilogues of a lone wave outweigh the better k-loop.)
struct LinBigArgs {
  const float* A;
  const float* P;
  const float* bias;
  const float* resid;
  float* Y;
  int64_t lda_g, ldr, ldy;
  int32_t M, K, N, Np, col0;
};

// One 32*RB-row x 256-column output tile (RB = 16-row blocks per wave: 4 -> 128 rows, 1 -> 32 rows).
// Both operands reach LDS by DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a THREE-slot
// ring, two k-steps ahead of the MFMAs; the one barrier per k-step waits for this wave's DMA of the step about to be
// read only (s_waitcnt vmcnt(n) with the newer batch left in flight -- __syncthreads() would carry a release fence =
// vmcnt(0) and end the prefetch at every barrier).  LDS image of a slot: As = [row][16 k] as float4 number
// row * 4 + k / 4 == the loading thread's id, Bs = [kq][col][4 k] as float4 number kq * 256 + col == thread id
// (+ 512 for kq + 2): both are "wave base + lane * 16 B", which is what the DMA writes.
constexpr int kLbSlots = 3;
constexpr int kLbA = 128 * 16, kLbB = 16 * 256;                    // floats per slot
constexpr size_t kLbLds = sizeof(float) * kLbSlots * (kLbA + kLbB);  // 72 KB: two workgroups per CU

__device__ __forceinline__ void glds16(const void* g, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int RB>
__device__ __forceinline__ void lin_big_tile(const LinBigArgs& a, const int row0, const int gcol0, float* As, float* Bs) {
  constexpr int BM = 32 * RB, BN = 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int M = a.M, nk = a.K >> 4;
  // DMA sources: A slice = BM rows x 4 float4 -> one per thread of the first 4*BM threads (whole waves: BM = 128 all
  // eight, BM = 32 the first two); B slab = 4 kq-rows x 256 cols float4 -> two per thread
  const int ar = tid >> 2, ac = tid & 3;
  const bool has_a = wave * 64 < 4 * BM;  // wave-uniform
  const int arow = row0 + ar < M ? row0 + ar : M - 1;
  const f32x4* __restrict__ Ag = reinterpret_cast<const f32x4*>(a.A + (size_t)arow * a.lda_g) + ac;
  const int bq0 = tid >> 8, bcol = tid & 255;  // chunks (bq0, bcol) and (bq0 + 2, bcol)
  const f32x4* __restrict__ Bg = reinterpret_cast<const f32x4*>(a.P) + (size_t)a.col0 + gcol0 + bcol;
  const int Np = a.Np;
  const int wbase = wave * 64 * 4;  // this wave's first float inside a slot image (float4 number = thread id)
  auto dma = [&](int ks) {
    float* as = As + (ks % kLbSlots) * kLbA;
    float* bs = Bs + (ks % kLbSlots) * kLbB;
    if (has_a) glds16(Ag + ks * 4, as + wbase);
    glds16(Bg + (size_t)(ks * 4 + bq0) * Np, bs + wbase);
    glds16(Bg + (size_t)(ks * 4 + bq0 + 2) * Np, bs + 512 * 4 + wbase);
  };
  f32x4 acc[RB][4];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  dma(0);
  if (nk > 1) dma(1);
  const int a_off = ((wr * 16 * RB + (lane & 15)) * 16 + 4 * (lane >> 4));  // floats inside an A slot
  const int b_off = ((lane >> 4) * BN + wc * 64 + (lane & 15)) * 4;          // floats inside a B slot
  for (int ks = 0; ks < nk; ++ks) {
    // batch ks landed (this wave's part), batch ks + 1 may stay in flight; then everyone's part landed and everyone
    // is done reading slot (ks + 2) % 3 (= the slot of step ks - 1)
    if (ks + 1 < nk) {
      if (has_a) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (ks + 2 < nk) dma(ks + 2);
    const float* as = As + (ks % kLbSlots) * kLbA;
    const float* bs = Bs + (ks % kLbSlots) * kLbB;
    f32x4 af[RB], bf[4];
#pragma unroll
    for (int r = 0; r < RB; ++r) af[r] = *reinterpret_cast<const f32x4*>(&as[a_off + r * 16 * 16]);
#pragma unroll
    for (int c = 0; c < 4; ++c) bf[c] = *reinterpret_cast<const f32x4*>(&bs[b_off + c * 64]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][t], bf[c][t], acc[r][c], 0, 0, 0);
  }
  // epilogue: + bias + residual, 16 lanes x 4 B contiguous per row
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = gcol0 + wc * 64 + c * 16 + (lane & 15);
    const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gr = row0 + wr * 16 * RB + r * 16 + (lane >> 4) * 4 + i;
        if (gr < M) {
          float v = acc[r][c][i] + bv;
          if (a.resid) v += a.resid[(size_t)gr * a.ldr + col];
          a.Y[(size_t)gr * a.ldy + col] = v;
        }
      }
    }
  }
}

// TAIL = 0: one 128-row tile per workgroup.  TAIL = 1: 160 rows per workgroup as a 128-row tile followed by a 32-row
// tile.  The host takes it when that turns a ragged last round of the 512 resident workgroups into whole rounds: at
// M = 81920, N = 256 (five of the eight GEMMs of a CDT block) 128-row tiles are 640 workgroups = 1.25 rounds, i.e. the
// launch pays two; 160 rows per workgroup are exactly 512.  (A single 160-row register tile needs 149 VGPRs: one
// workgroup per CU instead of two.)
template <int TAIL>
__global__ __launch_bounds__(512, 4) void linear_big_kernel(const LinBigArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lb_lds[];
  float* As = lb_lds;
  float* Bs = lb_lds + kLbSlots * kLbA;
  constexpr int kRows = TAIL ? 160 : 128;
  const int row0 = blockIdx.x * kRows, gcol0 = blockIdx.y * 256;
  lin_big_tile<4>(a, row0, gcol0, As, Bs);
  if (TAIL && row0 + 128 < a.M) {
    __syncthreads();  // every wave is done reading the first tile's last slots before the ring restarts
    lin_big_tile<1>(a, row0 + 128, gcol0, As, Bs);
  }
}


// ---- the persistent form of the same GEMM (round 4; tools/gemm_lab.hip is its bench, profiles/r4_gemm_lab.txt) -------
// What linear_big_kernel loses at M = 81920 (0.61-0.69 of the fp32 MFMA roof by shape): every workgroup of the chip
// reaches its epilogue at the same time -- 335 MB of stores in a burst at the chip's write rate with no MFMA running,
// then a DMA prologue with neither (the same kernel with the stores removed: 0.75-0.80).  Here ONE 8-wave workgroup
// per CU (256 registers per wave) walks a contiguous run of 128 x 128 tiles:
//  * an S-slot DMA ring that keeps running across tile boundaries (no prologue per tile): slab g + 1 is landed at the
//    top of k-step g and its fragments go to the OTHER fragment register set while the MFMAs of step g run;
//  * the finished tile's accumulators move to a second register set and are stored ("dripped") 2 x 64 lanes per k-step
//    during the first 16 k-steps of the next tile, the residual they need loaded one step earlier: the output leaves
//    the chip beside the MFMAs;
//  * A slot image XOR-swizzled (kq position = kq ^ perm[(row >> 2) & 3]) -> conflict-free ds_read_b128;
//  * 16 k-steps unrolled, branch-free except for the point where a wave issues its memory instructions: after the
//    first quarter of the step's MFMAs in waves 0-3, after the third in waves 4-7 (the two waves of a SIMD are w and
//    w + 4: one feeds the matrix pipe while the other issues ds_read / DMA, ~100+ cycles each, in order).
// Bit-identical to linear_big_kernel (same per-element k order).  Measured (M = 81920, us, old -> new):
// K256 N1024 427 -> 344, K256 N768 336 -> 264, K1024 N256 +resid 434 -> 332, K256 N256 +resid 141 -> 107,
// K768 N256 306 -> 245, K1024 N256 395 -> 319: 0.78-0.86 of the roof.
// Requires M % 128 == 0, K % 256 == 0, N % 128 == 0 (host: osrl_linear); past the last tile the "next" tile is the
// last one again (harmless reloads, drained before the kernel ends).
// a load the compiler does not track: no conservative vmcnt(0) at its use (which would also wait for the DMA issued
// since) -- the user waits with vm_wait<N>() naming the registers, N = vector memory operations issued after the load
__device__ __forceinline__ float gload_untracked(const float* ubase, unsigned voff_bytes) {
  float r;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(voff_bytes), "s"(ubase) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void vm_wait(float& r0, float& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N));
}

template <int RB, int CB, int S, bool RES>
__global__ __launch_bounds__(512, 2) void linear_pers_kernel(const LinBigArgs a, const int row_tiles, const int col_groups,
                                                          const int nwg) {
  constexpr int BM = 32 * RB, BN = 64 * CB;
  constexpr int kA = BM * 16, kB = 16 * BN;
  constexpr int PB = 4 * BN / 512;    // B pieces per thread and slab
  constexpr int NACC = RB * CB;       // accumulators (f32x4) per wave
  constexpr int SPI = 4 * NACC / 16;  // dripped stores per k-step
  constexpr int DMA_OPS = 1 + PB;
  constexpr int L = S - 1;            // slabs in flight ahead of the one being multiplied
  static_assert(BM == 128, "A slab = one float4 per thread");
  extern __shared__ __attribute__((aligned(16))) float lds_p[];
  float* As = lds_p;
  float* Bs = lds_p + S * kA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int M = a.M, Np = a.Np;
  const int ngrp = a.K >> 8;
  const int tiles = row_tiles * col_groups;
  const int wg = blockIdx.x;
  const int t0 = (int)((long)tiles * wg / nwg), t1 = (int)((long)tiles * (wg + 1) / nwg);
  if (t0 >= t1) return;
  const int wbase = wave * 64 * 4;
  // ---- issue side: this thread's float4 of the A slab (swizzled kq) and its PB float4 of the B slab ----
  const int a_row = tid >> 2;
  const int a_kq = (tid & 3) ^ ((0x78 >> (2 * ((tid >> 4) & 3))) & 3);
  int bkq[PB], bcol[PB];
#pragma unroll
  for (int p = 0; p < PB; ++p) {
    const int f = tid + 512 * p;
    bkq[p] = f / BN;
    bcol[p] = f % BN;
  }
  // running sources of the slab about to be issued: Ap = this thread's float4 of the A slab (per-thread pointer),
  // Bp = first float4 of the B slab (wave-uniform: scalar registers) + voffB[p] = this thread's float4 inside the slab
  // (never changes): loop-carried, so nothing about the 16 unrolled issues can be hoisted and spilled
  const f32x4 *Ap, *Agn;
  const f32x4 *Bp, *Bgn;
  unsigned voffB[PB];
#pragma unroll
  for (int p = 0; p < PB; ++p) voffB[p] = (unsigned)(bkq[p] * Np + bcol[p]);
  auto tile_ptrs = [&](int t, const f32x4*& ag, const f32x4*& bg) {
    const int rt = t / col_groups, cg = t - rt * col_groups;
    const int row = rt * BM + a_row;
    ag = reinterpret_cast<const f32x4*>(a.A + (size_t)(row < M ? row : M - 1) * a.lda_g) + a_kq;
    bg = reinterpret_cast<const f32x4*>(a.P) + (size_t)a.col0 + cg * BN;
  };
  tile_ptrs(t0, Ap, Bp);
  tile_ptrs(t0 + 1 < t1 ? t0 + 1 : t0, Agn, Bgn);
  int it_slot = 0;
  const size_t b_step = (size_t)4 * Np;
  auto dma = [&]() {  // the next slab of the tile being issued -> slot it_slot
    float* as = As + it_slot * kA;
    float* bs = Bs + it_slot * kB;
    glds16(Ap, as + wbase);
#pragma unroll
    for (int p = 0; p < PB; ++p) glds16(Bp + voffB[p], bs + p * 512 * 4 + wbase);
    Ap += 4;
    Bp += b_step;
    it_slot = it_slot + 1 == S ? 0 : it_slot + 1;
  };
  // ---- consume side ----
  const int a_off = (wr * 16 * RB + (lane & 15)) * 16 + 4 * ((lane >> 4) ^ ((0x78 >> (2 * ((lane >> 2) & 3))) & 3));
  const int b_off = ((lane >> 4) * BN + wc * 16 * CB + (lane & 15)) * 4;
  f32x4 acc[RB][CB], prev[RB][CB];
  f32x4 af[2][RB], bf[2][CB];
#pragma unroll
  for (int r = 0; r < RB; ++r)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  int c_slot = 0;  // slot of the slab being multiplied
  auto rd = [&](int slot, f32x4* afb, f32x4* bfb) {
    const float* as = As + slot * kA;
    const float* bs = Bs + slot * kB;
#pragma unroll
    for (int r = 0; r < RB; ++r) afb[r] = *reinterpret_cast<const f32x4*>(&as[a_off + r * 16 * 16]);
#pragma unroll
    for (int c = 0; c < CB; ++c) bfb[c] = *reinterpret_cast<const f32x4*>(&bs[b_off + c * 64]);
  };
  auto mm = [&](const f32x4* afb, const f32x4* bfb) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int r = 0; r < RB; ++r)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(afb[r][t], bfb[c][t], acc[r][c], 0, 0, 0);
  };
  // element e (0 .. 4 NACC) of a wave tile = accumulator e / 4 = (r, c), row i = e % 4.  Its address is a wave-uniform
  // base (scalar registers) + one per-lane 32-bit offset that never changes (saddr + voffset form).  The row stride is
  // laundered through an empty asm so that the 64 offsets of a tile stay inside the k-loop (not hoisted and spilled).
  const unsigned lane_row = (unsigned)(lane >> 4) * 4u;
  const unsigned loff_y = lane_row * (unsigned)a.ldy + (unsigned)(lane & 15);
  const unsigned loff_r = lane_row * (unsigned)a.ldr + (unsigned)(lane & 15);
  const float* rbase = a.resid;
  float* ybase = a.Y;
  auto el_off = [&](int e, int ld_in) -> unsigned {
    const int ai = e >> 2, i = e & 3, r = ai % RB;
    const int c = ai / RB;
    int ld = ld_in;
    asm volatile("" : "+s"(ld));
    return (unsigned)((r * 16 + i) * ld + c * 16);
  };
  float pbias[CB];

  // prologue: slabs 0 .. L-1 of the first tile in flight; slab 0 landed for everyone, its fragments on the way to set 0
#pragma unroll
  for (int i = 0; i < L; ++i) dma();
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((L - 1) * DMA_OPS) : "memory");
  rd(0, af[0], bf[0]);

  // ONE instance of the 16 unrolled k-steps: whether the previous tile is dripped during them (first group of a tile
  // that has a predecessor) and whether the slabs issued from step 16 - L on belong to the next tile (last group of a
  // tile) are wave-uniform run-time flags
  auto group = [&](auto drip_tag, const bool LAST) {
    constexpr bool DRIP = decltype(drip_tag)::value;
    // the residual of the elements stored at step j is loaded one step earlier (a full k-step of latency cover; the
    // first set waits once per tile)
    float rres[2][SPI];
    auto res_load = [&](int j, float* dst) {
#pragma unroll
      for (int s = 0; s < SPI; ++s) {
        dst[s] = gload_untracked(rbase + el_off(j * SPI + s, (int)a.ldr), loff_r * 4u);
      }
    };
    if (DRIP && RES) res_load(0, rres[0]);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // top of k-step g: slab g + 1 landed for everyone (slabs g + 2 .. g + S - 2 and the drips issued since may be in
      // flight); everyone's MFMAs of step g - 1 are issued, i.e. the slot of slab g - 1 is free
      constexpr int n_dma = (S - 3) * DMA_OPS;
      if (j >= S - 2 && DRIP) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n_dma + (S - 2) * SPI + (RES ? (S - 3) * SPI : 0)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(n_dma) : "memory");
      c_slot = c_slot + 1 == S ? 0 : c_slot + 1;
      if (DRIP && RES && j + 1 < 16) res_load(j + 1, rres[(j + 1) & 1]);
      if (j + L == 16 && LAST) {  // the slabs issued from here on belong to the next tile
        Ap = Agn;
        Bp = Bgn;
      }
      // this step's fragments are in registers already: MFMAs first.  The reads of the next slab and the DMA issue
      // (~100+ cycles each, in-order in this wave) go after the first quarter of the MFMAs in waves 0-3 and after the
      // third quarter in waves 4-7 -- the two waves of a SIMD are w and w + 4, so one feeds the matrix pipe while the
      // other issues memory instructions
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if ((t == 1 && wr == 0) || (t == 3 && wr == 1)) {
          rd(c_slot, af[(j + 1) & 1], bf[(j + 1) & 1]);
          dma();
        }
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
          for (int r = 0; r < RB; ++r)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j & 1][r][t], bf[j & 1][c][t], acc[r][c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (DRIP) {
        if (RES) {
          // issued after the loads of rres[j & 1]: this and the previous step's DMA, the previous step's stores and the
          // next set's loads (before the loop for j = 0: the next set's loads and this step's DMA)
          static_assert(SPI == 2, "vm_wait names two registers");
          if (j == 0) vm_wait<SPI + DMA_OPS>(rres[0][0], rres[0][1]);
          else if (j + 1 < 16) vm_wait<2 * DMA_OPS + 2 * SPI>(rres[j & 1][0], rres[j & 1][1]);
          else vm_wait<2 * DMA_OPS + SPI>(rres[j & 1][0], rres[j & 1][1]);
        }
#pragma unroll
        for (int s = 0; s < SPI; ++s) {
          const int e = j * SPI + s;
          float* yp = ybase + el_off(e, (int)a.ldy);
          const int ai = e >> 2, r = ai % RB, c = ai / RB;
          float pb = pbias[c];
          asm volatile("" : "+v"(pb));
          float v = prev[r][c][e & 3] + pb;
          if (RES) v += rres[j & 1][s];
          yp[loff_y] = v;
        }
      }
    }
  };

  for (int t = t0; t < t1; ++t) {
    for (int grp = 0; grp < ngrp; ++grp) {
      if (grp == 0 && t > t0) group(std::true_type{}, grp == ngrp - 1);
      else group(std::false_type{}, grp == ngrp - 1);
    }
    const int rt = t / col_groups, cg = t - rt * col_groups;
    const int urow0 = rt * BM + wr * 16 * RB;
    const int ucol0 = cg * BN + wc * 16 * CB;
    ybase = a.Y + (size_t)urow0 * a.ldy + ucol0;
    if (RES) rbase = a.resid + (size_t)urow0 * a.ldr + ucol0;
#pragma unroll
    for (int c = 0; c < CB; ++c) pbias[c] = a.bias ? a.bias[ucol0 + c * 16 + (lane & 15)] : 0.f;
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        prev[r][c] = acc[r][c];
        acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    tile_ptrs(t + 2 < t1 ? t + 2 : t1 - 1, Agn, Bgn);
  }
  // the last tile's results (and the reloads issued past the end must have landed before the workgroup's LDS goes)
#pragma unroll
  for (int e = 0; e < 4 * NACC; ++e) {
    float* yp = ybase + el_off(e, (int)a.ldy);
    const int ai = e >> 2, r = ai % RB, c = ai / RB;
    float v = prev[r][c][e & 3] + pbias[c];
    if (RES) v += (rbase + el_off(e, (int)a.ldr))[loff_r];
    yp[loff_y] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// forward pack  PF[q = k/4][n][k%4],  n < round16(N), q < round16(K)/4        (y = x W^T, W [N,K])
// backward pack PB[q = o/4][i][o%4],  i < round16(K)+16, q < round16(N)/4     (dx = dz W)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src_flat, float* __restrict__ pf,
                                                    float* __restrict__ pb, const osrl_pack_entry_t* __restrict__ ents) {
  const osrl_pack_entry_t E = ents[blockIdx.y];
  const float* __restrict__ W = src_flat + E.src_off;
  const int N = E.out, K = E.in;
  const int Np = round16(N), Kp = round16(K);
  if (pf && E.f_off >= 0) {
    float* __restrict__ d = pf + E.f_off;
    const int total = Kp * Np;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
      const int t = j & 3, n = (j >> 2) % Np, q = (j >> 2) / Np;
      const int k = q * 4 + t;
      d[j] = (n < N && k < K) ? W[(size_t)n * K + k] : 0.f;
    }
  }
  if (pb && E.b_off >= 0) {
    float* __restrict__ d = pb + E.b_off;
    const int Kb = Kp + 16;
    const int total = Np * Kb;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < total; j += gridDim.x * 256) {
      const int t = j & 3, i = (j >> 2) % Kb, q = (j >> 2) / Kb;
      const int o = q * 4 + t;
      d[j] = (o < N && i < K) ? W[(size_t)o * K + i] : 0.f;
    }
  }
}

inline int round16h(int x) { return (x + 15) & ~15; }

struct TileChoice {
  int nrb, ncb, lda, nw;  // row blocks per tile, column blocks per wave, LDS row stride, waves per workgroup
};

// tile rows: keep >= ~2 workgroups per CU in flight when the row count allows it, otherwise shrink
// the tile so that small batches still spread over the 256 CUs.
inline TileChoice choose_tile(const osrl_mlp_t* net, int rows, int extra_width) {
  int maxw = extra_width;
  for (int l = 0; l <= net->n_layers; ++l) maxw = net->dims[l] > maxw ? net->dims[l] : maxw;
  const int nblk = (maxw + 15) / 16;
  const int cpw = (nblk + 3) / 4;
  TileChoice t;
  t.ncb = cpw <= 1 ? 1 : cpw <= 2 ? 2 : cpw <= 4 ? 4 : 7;
  t.lda = round16h(maxw) + 8;
  // Measured on MI355X (tools/kbench.py, profiles/kbench_r1.txt): with the packed weight layout small
  // row tiles win -- more workgroups in flight hide the L2 latency of the weight stream better than
  // the 2-4x weight reuse of bigger tiles pays back.  16-row tiles everywhere, 32-row tiles once the
  // launch has >= 1024 of them (e.g. the N*B = 20480-row x 2-net OOD scoring of CPQ).
  const long wg32 = (long)((rows + 31) / 32) * net->n_nets;
  t.nrb = (t.ncb != 7 && wg32 >= 1024) ? 2 : 1;
  t.nw = 4;
  if (net->tile_rows == 16 || net->tile_rows == 32 || (net->tile_rows == 64 && t.ncb != 7)) {
    t.nrb = net->tile_rows / 16;
    return t;
  }
  // Launches with at most ~2 workgroups per CU (the 2048-row training launches: 128 row tiles x 1-4 nets) are
  // serial latency chains inside each workgroup: give the workgroup 8 waves (2 per SIMD, half the column blocks
  // each) so that two k-loops interleave on every SIMD and staging / epilogues use twice the lanes.
  const long wg16 = (long)((rows + 15) / 16) * net->n_nets;
  if (t.nrb == 1 && cpw > 2 && wg16 <= 2 * 256 && t.lda >= 128) {
    t.nw = 8;
    t.ncb = (nblk + 7) / 8 <= 2 ? 2 : 4;
  }
  return t;
}

// kernel_p (may be nullptr): the variant that reads its descriptor from device memory; taken when the calling
// thread's argument arena (argmem.h) holds an uploaded copy of `args`
template <typename Args, typename K>
int launch_tiles(K kernel, void (*kernel_p)(const void*), const Args& args, int rows, int nets, int nrb, int lda,
                 hipStream_t stream, int threads = 256, int wg_cap = 0) {
  const int BM = 16 * nrb;
  const size_t lds_bytes = (size_t)BM * lda * sizeof(float);
  int tiles = (rows + BM - 1) / BM;
  if (wg_cap > 0 && tiles * nets > wg_cap) tiles = (wg_cap + nets - 1) / nets;  // forward kernel loops over tiles
  dim3 grid(tiles, nets, 1);
  const void* dev_args = kernel_p ? osrl_argmem::slot(args) : nullptr;
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(kernel_p) : reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (dev_args)
    hipLaunchKernelGGL(kernel_p, grid, dim3(threads), lds_bytes, stream, dev_args);
  else
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds_bytes, stream, args);
  return (int)hipGetLastError();
}
template <typename Args, typename K>
int launch_tiles(K kernel, const Args& args, int rows, int nets, int nrb, int lda, hipStream_t stream,
                 int threads = 256, int wg_cap = 0) {
  return launch_tiles(kernel, (void (*)(const void*))nullptr, args, rows, nets, nrb, lda, stream, threads, wg_cap);
}

#define OSRL_DISPATCH_TILE(KERNEL, ARGS, ROWS, NETS, T, STREAM, CAP)                        \
  do {                                                                                  \
    if (T.nw == 8) {                                                                    \
      if (T.ncb == 2) return launch_tiles(KERNEL<1, 2, 8>, KERNEL##_p<1, 2, 8>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 512, CAP); \
      return launch_tiles(KERNEL<1, 4, 8>, KERNEL##_p<1, 4, 8>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 512, CAP);     \
    }                                                                                   \
    if (T.ncb == 1) {                                                                   \
      if (T.nrb == 4) return launch_tiles(KERNEL<4, 1>, KERNEL##_p<4, 1>, ARGS, ROWS, NETS, 4, T.lda, STREAM, 256, CAP); \
      if (T.nrb == 2) return launch_tiles(KERNEL<2, 1>, KERNEL##_p<2, 1>, ARGS, ROWS, NETS, 2, T.lda, STREAM, 256, CAP); \
      return launch_tiles(KERNEL<1, 1>, KERNEL##_p<1, 1>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 256, CAP);             \
    }                                                                                   \
    if (T.ncb == 2) {                                                                   \
      if (T.nrb == 4) return launch_tiles(KERNEL<4, 2>, KERNEL##_p<4, 2>, ARGS, ROWS, NETS, 4, T.lda, STREAM, 256, CAP); \
      if (T.nrb == 2) return launch_tiles(KERNEL<2, 2>, KERNEL##_p<2, 2>, ARGS, ROWS, NETS, 2, T.lda, STREAM, 256, CAP); \
      return launch_tiles(KERNEL<1, 2>, KERNEL##_p<1, 2>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 256, CAP);             \
    }                                                                                   \
    if (T.ncb == 4) {                                                                   \
      if (T.nrb == 4) return launch_tiles(KERNEL<4, 4>, KERNEL##_p<4, 4>, ARGS, ROWS, NETS, 4, T.lda, STREAM, 256, CAP); \
      if (T.nrb == 2) return launch_tiles(KERNEL<2, 4>, KERNEL##_p<2, 4>, ARGS, ROWS, NETS, 2, T.lda, STREAM, 256, CAP); \
      return launch_tiles(KERNEL<1, 4>, KERNEL##_p<1, 4>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 256, CAP);             \
    }                                                                                   \
    if (T.nrb == 2) return launch_tiles(KERNEL<2, 7>, KERNEL##_p<2, 7>, ARGS, ROWS, NETS, 2, T.lda, STREAM, 256, CAP); \
    return launch_tiles(KERNEL<1, 7>, KERNEL##_p<1, 7>, ARGS, ROWS, NETS, 1, T.lda, STREAM, 256, CAP);               \
  } while (0)

bool valid_net(const osrl_mlp_t* n) {
  if (!n || n->n_layers < 1 || n->n_layers > OSRL_MAX_LAYERS || n->n_nets < 1 || n->n_nets > OSRL_MAX_NETS)
    return false;
  for (int l = 0; l <= n->n_layers; ++l)
    if (n->dims[l] < 1 || n->dims[l] > OSRL_MAX_WIDTH) return false;
  return true;
}


// ---- one supervised regression step of one MLP in ONE launch (osrl_mlp_regress_step) ---------------------------
// BC at B = 256 (bc.py:45-55,103-109) is six dependent launches of 2-8 us of work each: the step is its launch gaps.
// Here workgroup w < n_tiles owns rows [16 w, 16 w + 16): it draws and gathers them (the replay sampler's indices are a
// pure function of (seed, step, row): gather.h), runs the forward, the MSE gradient and the backward chain through the
// same bodies as the separate launches (mlp_fwd_body / mlp_bwd_dz_body, 8 waves), and signs in at an arrival counter.
// Workgroup w < n_work owns item w of the 64 x 64 dW work list; every item covers ALL rows (one row split), so once
// the counter shows every row tile, the tile's gradient is complete in LDS (dwt_tile) and the workgroup applies Adam
// to it right there -- no gradient slab, no optimizer launch.  The last workgroup to finish ticks the step state and
// re-arms the counters.  One grid-wide dependency (the counter) instead of five launch boundaries.
// All <= OSRL_STEP_MAX_WG workgroups are resident at once (one per CU, 256 CUs), so the wait cannot starve; it is
// bounded anyway (kStepSpinMax polls, then the error word is set and the workgroup leaves).
constexpr int kStepThreads = 512;
#ifndef OSRL_STEP_FENCE_ALL
#define OSRL_STEP_FENCE_ALL 0  // 1: every wave executes the release fence (3.4 us vs 1.8 us, tools/step_stamps.py)
#endif
#ifdef OSRL_STEP_STAMPS  // tools/step_stamps.py: 100 MHz wall-clock stamps of thread 0 of every workgroup (debug builds only)
__device__ long long g_step_stamp[OSRL_STEP_MAX_WG][16];
#define STEP_STAMP(i) \
  if (threadIdx.x == 0) g_step_stamp[blockIdx.x][i] = wall_clock64();
#else
#define STEP_STAMP(i)
#endif
constexpr int kStepSpinMax = 1 << 22;  // x ~130 ns per poll: ~0.5 s
struct StepArgs {
  FwdArgs fwd;
  BwdArgs bwd;
  osrl_gather::GatherArgs gather;
  osrl_step_state_t* st;
  const float* stats_cur;
  float* ring;
  const float* target;
  float* du;
  float* stat;
  const osrl_dw_entry_t* entries;
  const int32_t* work;
  float *p, *m, *v;
  const int32_t *map_f, *map_b;
  float *pf, *pb;
  float* ws;  // [OSRL_STEP_MAX_WG] loss partials | [0] tiles arrived  [1] workgroups done  [2] error
  float beta1, beta2, lr, eps, inv_n;
  int32_t warmup, n_stats, ring_len, n_tiles, n_work, n_wg, rows, tile_blocks;
};

template <int NCB, class AR>
__device__ __forceinline__ void mlp_step_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int64_t s_t;
  __shared__ int s_flag;
  __shared__ float s_red[kStepThreads / 64];
  __shared__ float s_part[OSRL_STEP_MAX_WG];
  __shared__ float s_tick[3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  unsigned* __restrict__ ctr = reinterpret_cast<unsigned*>(a.ws + OSRL_STEP_MAX_WG);
  STEP_STAMP(0);
  if (tid == 0) s_t = __atomic_load_n(&a.st->step, __ATOMIC_RELAXED);
  __syncthreads();
  STEP_STAMP(1);
  const int64_t t_old = s_t;
  const uint32_t step = (uint32_t)(t_old + 1);
  // the previous step's statistics go to the ring before anything of this step can overwrite them (the loss is stored
  // by the last row tile to arrive, and workgroup 0 is a row tile: it arrives after this)
  if (wg == 0) osrl_step::commit_stats<kStepThreads>(t_old, a.stats_cur, a.ring, a.n_stats, a.ring_len);

  if (wg < a.n_tiles) {
    const int rows = a.rows, row0 = wg * 16;
    if (a.gather.n_fields > 0) {  // half a wave per row: this tile's 16 rows in one pass
      osrl_gather::gather_tile16<decltype((a.gather))>(a.gather, step, wg);
      __syncthreads();  // (workgroup scope: the rows were written through this CU's L1)
    }
    STEP_STAMP(2);
    mlp_fwd_body<1, NCB, 8, decltype((a.fwd))>(a.fwd, 0, wg);
    __syncthreads();
    STEP_STAMP(3);
    {  // F.mse_loss (bc.py:46-47) on this tile: du = 2 (u - target) / n, partial sum of squares
      const int L = a.fwd.net.n_layers, ad = a.fwd.net.dims[L];
      const float* __restrict__ u = a.fwd.out.h[0][L - 1];
      const float* __restrict__ tg = a.target;
      float* __restrict__ du = a.du;
      const float inv_n = a.inv_n;
      float loss = 0.f;
      int n_here = (rows - row0) * ad;
      n_here = n_here > 16 * ad ? 16 * ad : n_here;
      for (int idx = tid; idx < n_here; idx += kStepThreads) {
        const size_t i = (size_t)row0 * ad + idx;
        const float d = u[i] - tg[i];
        loss += d * d;
        du[i] = 2.0f * d * inv_n;
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) loss += __shfl_xor(loss, o);
      if (lane == 0) s_red[wave] = loss;
      __syncthreads();
      if (tid == 0)
        a.ws[wg] = (((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) + ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])));
    }
    STEP_STAMP(4);
    mlp_bwd_dz_body<1, NCB, 8, decltype((a.bwd))>(a.bwd, 0, wg);
    STEP_STAMP(5);
    // release: what this tile wrote (activations, dZ, the loss partial) leaves this XCD's L2 before the arrival is
    // counted.  A barrier does not wait for global stores (the compiler emits lgkmcnt only), so every wave first waits
    // for ITS stores to be acknowledged by the L2; then one write-back of the L2 covers them all.
#if OSRL_STEP_FENCE_ALL
    __threadfence();
    __syncthreads();
    if (tid == 0) {
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __threadfence();
#endif
      const unsigned seen = atomicAdd(ctr, 1u);
      s_flag = seen == (unsigned)a.n_tiles - 1;
      if (s_flag) __threadfence();
    }
    __syncthreads();
    STEP_STAMP(6);
    if (s_flag) {  // last row tile: the loss, summed in tile order (every partial fetched by its own lane)
      if (tid < OSRL_STEP_MAX_WG) s_part[tid] = tid < a.n_tiles ? reinterpret_cast<const volatile float*>(a.ws)[tid] : 0.f;
      __syncthreads();
      if (tid == 0 && a.stat) {
        float t = 0.f;
        for (int g = 0; g < a.n_tiles; ++g) t += s_part[g];
        a.stat[0] = t * a.inv_n;
      }
    }
  }

  STEP_STAMP(7);
  if (wg < a.n_work) {
    auto phase_b = [&](auto t_c) {
      constexpr int T = decltype(t_c)::value, TW = 16 * T, LD = TW + 1;
      constexpr int PER = (TW * TW + kStepThreads - 1) / kStepThreads;  // elements of the tile per lane (8 / 2)
      // The optimizer state of this workgroup's tile does not depend on the row tiles: p / m / v and the pack maps of
      // this lane's elements are requested BEFORE the wait and are in registers when the gradient is.
      const int ei = a.work[wg * 4 + 0], ot = a.work[wg * 4 + 1], it = a.work[wg * 4 + 2];
      const OSRL_CAS osrl_dw_entry_t& E0 = ((const OSRL_CAS osrl_dw_entry_t*)a.entries)[ei];
      const int out = E0.out, in = E0.in, o0 = ot * TW, i0 = it * TW;
      float* __restrict__ P = a.p;
      float* __restrict__ M = a.m;
      float* __restrict__ V = a.v;
      const int32_t* __restrict__ MF = a.map_f;
      const int32_t* __restrict__ MB = a.map_b;
      float pv[PER], mv[PER], vv[PER];
      int mf[PER], mb[PER];
      unsigned w[PER];
      bool ok[PER];
      const bool has_b = it == 0 && tid < TW && o0 + tid < out;
      const unsigned wb = (unsigned)(E0.b_off + (has_b ? o0 + tid : 0));
      float pb_, mb_, vb_;
      auto load_state = [&]() {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int idx = tid + j * kStepThreads;
          const int ol = idx / TW, il = idx - ol * TW;
          const int o = o0 + ol, i = i0 + il;
          ok[j] = idx < TW * TW && o < out && i < in;
          w[j] = ok[j] ? (unsigned)(E0.w_off + (int64_t)o * in + i) : (unsigned)E0.w_off;  // (groups are < 2^32 floats)
          pv[j] = P[w[j]];
          mv[j] = M[w[j]];
          vv[j] = V[w[j]];
          mf[j] = MF ? MF[w[j]] : -1;
          mb[j] = MB ? MB[w[j]] : -1;
        }
        pb_ = P[wb];
        mb_ = M[wb];
        vb_ = V[wb];
      };
      // (64 x 64 tiles: 8 elements per lane = 48 registers on top of the tile's 64 accumulators and 64 fragment
      // registers -- they would spill; there the state is read when the gradient is ready, as the optimizer launch does)
      constexpr bool PRE = T == 2;
      if constexpr (PRE) load_state();
      if (tid == 0) {
        int polls = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)a.n_tiles) {
          __builtin_amdgcn_s_sleep(4);
          if (++polls > kStepSpinMax) {
            ctr[2] = 1u;
            break;
          }
        }
        // acquire: drop this CU's L1 / this XCD's L2 copies of what the row tiles wrote.  A cache operation of the
        // CU, not of the wave: the other waves' loads come after the barrier behind it.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      STEP_STAMP(8);
      __syncthreads();
      STEP_STAMP(9);
      if (wave == 7 && lane < 3) {
        // this step's bias corrections (osrl_step::tick_values: two double-precision pow, ~2 us on one lane) on a wave
        // that idles while waves 0-3 run the tile's MFMAs; lanes 0 / 1 take beta1 / beta2 in lockstep
        const double t = (double)(t_old + 1);
        const double pw = pow((double)(lane == 0 ? a.beta1 : a.beta2), t);
        const float lrs = a.warmup > 0 ? (float)fmin(t / (double)a.warmup, 1.0) : 1.0f;
        s_tick[lane] = lane == 0 ? (float)(1.0 - pw) : lane == 1 ? (float)sqrt(1.0 - pw) : lrs;
      }
      dwt_tile<T, false, T == 2>(a.entries, a.work, wg, a.rows, lds,
                                 [&](const OSRL_CAS osrl_dw_entry_t&, const int, const int, const int /*split*/, const bool) {
        STEP_STAMP(10);
        if constexpr (!PRE) load_state();
        const float lr_t = a.lr * s_tick[2];
        const osrl_adam::Coef c{a.beta1, a.beta2, a.eps, lr_t / s_tick[0], s_tick[1]};
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int idx = tid + j * kStepThreads;
          const int ol = idx / TW, il = idx - ol * TW;
          const int off = ok[j] ? ol * LD + il : 0;
          const float g = ((lds[off] + lds[TW * LD + off]) + lds[2 * TW * LD + off]) + lds[3 * TW * LD + off];
          if (ok[j]) {
            osrl_adam::update1(pv[j], mv[j], vv[j], g, c);
            P[w[j]] = pv[j];
            M[w[j]] = mv[j];
            V[w[j]] = vv[j];
            if (mf[j] >= 0) a.pf[mf[j]] = pv[j];
            if (mb[j] >= 0) a.pb[mb[j]] = pv[j];
          }
        }
        if (has_b) {
          const float* db = lds + 4 * TW * LD;
          const float gb = ((db[tid] + db[TW + tid]) + db[2 * TW + tid]) + db[3 * TW + tid];
          osrl_adam::update1(pb_, mb_, vb_, gb, c);
          P[wb] = pb_;
          M[wb] = mb_;
          V[wb] = vb_;
        }
      });
    };
    if (a.tile_blocks == 2)
      phase_b(std::integral_constant<int, 2>{});
    else
      phase_b(std::integral_constant<int, 4>{});
  }

  STEP_STAMP(11);
  __syncthreads();
  if (tid == 0) {
    // (no fence: the count orders nothing but the tick below, and every workgroup read t_old before its first barrier)
    const unsigned done = atomicAdd(ctr + 1, 1u);
    if (done == (unsigned)a.n_wg - 1) {  // every workgroup read t_old long ago and is past the counter: tick, re-arm
      if (wg < a.n_work) {  // (wave 7 of a dW workgroup computed exactly osrl_step::tick_values(t_old + 1) already)
        a.st->step = t_old + 1;
        a.st->bc1 = s_tick[0];
        a.st->bc2_sqrt = s_tick[1];
        a.st->lr_scale = s_tick[2];
      } else {
        osrl_step::advance(a.st, t_old, a.beta1, a.beta2, a.warmup);
      }
      ctr[0] = 0u;
      ctr[1] = 0u;
    }
  }
  STEP_STAMP(12);
}

template <int NCB>
__global__ __launch_bounds__(kStepThreads, 2) void mlp_step_kernel(const StepArgs a) {
  mlp_step_body<NCB, const StepArgs&>(a);
}
template <int NCB>
__global__ __launch_bounds__(kStepThreads, 2) void mlp_step_kernel_p(const void* p) {
  mlp_step_body<NCB, const OSRL_CAS StepArgs&>(*(const OSRL_CAS StepArgs*)p);
}

}  // namespace

// forward tails: argument check, and the same arithmetic as separate launches (the fallback when a launch keeps no
// output tile in LDS)
static bool fwd_tail_ok(const osrl_mlp_tail_t* t, const osrl_mlp_t* net) {
  if (!t || t->kind == OSRL_TAIL_NONE) return true;
  if (t->L < 1 || 2 * t->L != net->dims[net->n_layers]) return false;
  if (t->kind == OSRL_TAIL_VAE_LATENT) return t->eps && t->out;
  if (t->kind == OSRL_TAIL_VAE_KL) return t->out != nullptr;
  if (t->kind == OSRL_TAIL_GAUSS)
    return (!t->eps || t->out) && (!t->eps2 || t->out2) && (!t->eps_ood || (t->out_ood && t->n_samples >= 1));
  return false;
}
static int fwd_tail_as_launches(const osrl_mlp_tail_t* t, const float* head, int rows, void* stream) {
  if (t->kind == OSRL_TAIL_VAE_LATENT) return osrl_vae_latent(head, t->eps, rows, t->L, t->out, stream);
  if (t->kind == OSRL_TAIL_VAE_KL) return osrl_vae_kl_rows(head, rows, t->L, t->out, stream);
  int rc = 0;
  if (t->eps) rc = osrl_gauss_head(head, t->eps, rows, t->L, t->max_action, t->out, nullptr, nullptr, stream);
  if (rc == 0 && t->eps2) rc = osrl_gauss_head(head, t->eps2, rows, t->L, t->max_action, t->out2, t->tanh2, nullptr, stream);
  if (rc == 0 && t->eps_ood) rc = osrl_gauss_ood_sample(head, t->eps_ood, t->n_samples, rows, t->L, t->out_ood, stream);
  return rc;
}

static int mlp_forward_impl(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out,
                            const osrl_mlp_tail_t* tail, void* stream) {
  if (!valid_net(net) || net->out_scale == 0.f || !in || !out || in->rows < 1 || in->d0 + in->d1 != net->dims[0]) return -1;
  for (int e = 0; e < net->n_nets; ++e) {
    if (!out->h[e][net->n_layers - 1]) return -1;
    for (int l = 0; l < net->n_layers; ++l)
      if (!net->Wf[e][l] || !net->b[e][l]) return -1;
  }
  const bool want_tail = tail && tail->kind != OSRL_TAIL_NONE;
  if (want_tail && !fwd_tail_ok(tail, net)) return -1;
  {
    const bool kl_tail = want_tail && tail->kind == OSRL_TAIL_VAE_KL;  // the one tail the 80-row kernel runs itself
    const int rc = osrl_launch_fwd_nb(net, in, out, (hipStream_t)stream, kl_tail ? tail->out : nullptr, kl_tail ? tail->L : 0);
    if (rc != kNbNotTaken) {  // the 80-row kernel keeps no output tile in LDS: any other tail is its own launch
      if (rc != 0 || !want_tail || kl_tail) return rc;
      if (in->row_list) return -3;
      return fwd_tail_as_launches(tail, out->h[0][net->n_layers - 1], in->rows, stream);
    }
  }
  if (in->row_list || in->n_rows_dev) return -3;  // a device-chosen row set: only the 80-row inference forward reads it (share0 is a hint)
  if (want_tail && tail->kind == OSRL_TAIL_VAE_KL) {  // tile kernels: the plain forward, then the rows kernel
    const int rc = mlp_forward_impl(net, in, out, nullptr, stream);
    return rc != 0 ? rc : fwd_tail_as_launches(tail, out->h[0][net->n_layers - 1], in->rows, stream);
  }
  FwdArgs a{};
  a.net = *net;
  a.in = *in;
  a.out = *out;
  a.tail = osrl_mlp_tail_t{};
  if (want_tail) a.tail = *tail;
  const TileChoice t = choose_tile(net, in->rows, 0);
  a.lda = t.lda;
  if (net->wg_cap > 0 && t.nw == 4 && (t.ncb == 4 || t.ncb == 7) && t.nrb <= 2 &&
      (long)((in->rows + 16 * t.nrb - 1) / (16 * t.nrb)) * net->n_nets > net->wg_cap) {
    hipStream_t st = (hipStream_t)stream;
    const int cap = net->wg_cap, R = in->rows, E = net->n_nets;
    if (t.ncb == 4) {
      if (t.nrb == 2) return launch_tiles(mlp_fwd_loop_kernel<2, 4>, a, R, E, 2, t.lda, st, 256, cap);
      return launch_tiles(mlp_fwd_loop_kernel<1, 4>, a, R, E, 1, t.lda, st, 256, cap);
    }
    if (t.nrb == 2) return launch_tiles(mlp_fwd_loop_kernel<2, 7>, a, R, E, 2, t.lda, st, 256, cap);
    return launch_tiles(mlp_fwd_loop_kernel<1, 7>, a, R, E, 1, t.lda, st, 256, cap);
  }
  OSRL_DISPATCH_TILE(mlp_fwd_kernel, a, in->rows, net->n_nets, t, (hipStream_t)stream, 0);
}

extern "C" int osrl_mlp_forward(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out,
                                void* stream) {
  return mlp_forward_impl(net, in, out, nullptr, stream);
}

extern "C" int osrl_mlp_forward_tail(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out,
                                     const osrl_mlp_tail_t* tail, void* stream) {
  return mlp_forward_impl(net, in, out, tail, stream);
}

template <int NRB, int NCB, int NW>
static int launch_fwd2(const FwdArgs& a0, const FwdArgs& a1, int nets0, int nets1, int rows0, int rows1, int lda,
                       hipStream_t stream) {
  const int BM = 16 * NRB;
  const int t0 = (rows0 + BM - 1) / BM, t1 = (rows1 + BM - 1) / BM;
  const size_t lds_bytes = (size_t)BM * lda * sizeof(float);
  const void* dev_args = nullptr;
  if (osrl_argmem::current()) {
    Fwd2Args f{};
    f.a0 = a0;
    f.a1 = a1;
    f.nets0 = nets0;
    f.tiles0 = t0;
    f.tiles1 = t1;
    dev_args = osrl_argmem::slot(f);
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(mlp_fwd2_kernel_p<NRB, NCB, NW>)
                                                : reinterpret_cast<const void*>(mlp_fwd2_kernel<NRB, NCB, NW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
  }
  (void)hipGetLastError();
  const dim3 grid(t0 > t1 ? t0 : t1, nets0 + nets1, 1);
  if (dev_args)
    hipLaunchKernelGGL((mlp_fwd2_kernel_p<NRB, NCB, NW>), grid, dim3(64 * NW), lds_bytes, stream, dev_args);
  else
    hipLaunchKernelGGL((mlp_fwd2_kernel<NRB, NCB, NW>), grid, dim3(64 * NW), lds_bytes, stream, a0, a1, nets0, t0, t1);
  return (int)hipGetLastError();
}

static int mlp_forward2_impl(const osrl_mlp_t* net0, const osrl_rows_t* in0, const osrl_mlp_acts_t* out0,
                             const osrl_mlp_tail_t* tail0, const osrl_mlp_t* net1, const osrl_rows_t* in1,
                             const osrl_mlp_acts_t* out1, const osrl_mlp_tail_t* tail1, void* stream) {
  if (!valid_net(net0) || !valid_net(net1) || !in0 || !in1 || !out0 || !out1) return -1;
  if (in0->row_list || in0->n_rows_dev || in1->row_list || in1->n_rows_dev) return -3;  // (osrl_mlp_forward only)
  if (!fwd_tail_ok(tail0, net0) || !fwd_tail_ok(tail1, net1)) return -1;
  TileChoice t0 = choose_tile(net0, in0->rows, 0), t1 = choose_tile(net1, in1->rows, 0);
  // pair only 16-row-tile launches of equal tile shape; anything else runs as two launches
  // a KL tail exists on the single-launch path only (80-row kernel or a follow-up osrl_vae_kl_rows launch)
  const bool kl_tail = (tail0 && tail0->kind == OSRL_TAIL_VAE_KL) || (tail1 && tail1->kind == OSRL_TAIL_VAE_KL);
  const bool pair = !kl_tail && t0.nrb == 1 && t1.nrb == 1 && t0.ncb == t1.ncb && t0.nw == t1.nw &&
                    ((t0.nw == 8 && (t0.ncb == 2 || t0.ncb == 4)) || (t0.nw == 4 && (t0.ncb == 4 || t0.ncb == 7)));
  if (!pair) {
    const int rc = mlp_forward_impl(net0, in0, out0, tail0, stream);
    return rc != 0 ? rc : mlp_forward_impl(net1, in1, out1, tail1, stream);
  }
  for (int p = 0; p < 2; ++p) {
    const osrl_mlp_t* net = p ? net1 : net0;
    const osrl_rows_t* in = p ? in1 : in0;
    const osrl_mlp_acts_t* out = p ? out1 : out0;
    if (net->out_scale == 0.f || in->rows < 1 || in->d0 + in->d1 != net->dims[0]) return -1;
    for (int e = 0; e < net->n_nets; ++e) {
      if (!out->h[e][net->n_layers - 1]) return -1;
      for (int l = 0; l < net->n_layers; ++l)
        if (!net->Wf[e][l] || !net->b[e][l]) return -1;
    }
  }
  FwdArgs a0{}, a1{};
  a0.net = *net0; a0.in = *in0; a0.out = *out0;
  a1.net = *net1; a1.in = *in1; a1.out = *out1;
  a0.tail = a1.tail = osrl_mlp_tail_t{};
  if (tail0) a0.tail = *tail0;
  if (tail1) a1.tail = *tail1;
  const int lda = t0.lda > t1.lda ? t0.lda : t1.lda;
  a0.lda = a1.lda = lda;
  const int n0 = net0->n_nets, n1 = net1->n_nets, r0 = in0->rows, r1 = in1->rows;
  hipStream_t st = (hipStream_t)stream;
  if (t0.nw == 8) {
    if (t0.ncb == 2) return launch_fwd2<1, 2, 8>(a0, a1, n0, n1, r0, r1, lda, st);
    return launch_fwd2<1, 4, 8>(a0, a1, n0, n1, r0, r1, lda, st);
  }
  if (t0.ncb == 4) return launch_fwd2<1, 4, 4>(a0, a1, n0, n1, r0, r1, lda, st);
  return launch_fwd2<1, 7, 4>(a0, a1, n0, n1, r0, r1, lda, st);
}

extern "C" int osrl_mlp_forward2(const osrl_mlp_t* net0, const osrl_rows_t* in0, const osrl_mlp_acts_t* out0,
                                 const osrl_mlp_t* net1, const osrl_rows_t* in1, const osrl_mlp_acts_t* out1,
                                 void* stream) {
  return mlp_forward2_impl(net0, in0, out0, nullptr, net1, in1, out1, nullptr, stream);
}

extern "C" int osrl_mlp_forward2_tail(const osrl_mlp_t* net0, const osrl_rows_t* in0, const osrl_mlp_acts_t* out0,
                                      const osrl_mlp_tail_t* tail0, const osrl_mlp_t* net1, const osrl_rows_t* in1,
                                      const osrl_mlp_acts_t* out1, const osrl_mlp_tail_t* tail1, void* stream) {
  return mlp_forward2_impl(net0, in0, out0, tail0, net1, in1, out1, tail1, stream);
}

static bool seed_ok(const osrl_mlp_seed_t* s, const osrl_mlp_t* net, const osrl_mlp_acts_t* saved) {
  if (!s || s->kind == OSRL_SEED_NONE) return true;
  const int L = net->n_layers, NL = net->dims[L];
  for (int e = 0; e < net->n_nets; ++e)
    if (!saved->h[e][L - 1]) return false;  // the seeds read the nets' outputs
  if ((s->partials != nullptr) != (s->counter != nullptr)) return false;
  switch (s->kind) {
    case OSRL_SEED_MSE: return s->x0 && (!s->kl_head || s->kl_L >= 1);
    case OSRL_SEED_CPQ_CRITIC:
      return NL == 1 && s->a && s->b && s->x0 && s->x1 && s->n_a >= 1 && s->n_a <= kSeedEns && s->n_b >= 1 && s->n_b <= kSeedEns;
    case OSRL_SEED_CPQ_COST: return NL == 1 && s->a && s->x0 && s->n_a >= 1 && s->n_a <= kSeedEns;
    case OSRL_SEED_CPQ_ACTOR:
      return NL == 1 && s->a && s->b && s->n_a >= 1 && s->n_a <= kSeedEns && s->n_b >= 1 && s->n_b <= kSeedEns;
    case OSRL_SEED_BCQ_CRITIC:
      return NL == 1 && s->a && s->x0 && s->n_a >= 1 && s->n_b >= 1 && s->n_samples >= 1;
    case OSRL_SEED_GAUSS_HEAD:
      return (NL & 1) == 0 && net->n_nets == 1 && s->a && s->eps && s->tanh_u && s->n_a >= 1 && s->n_a <= kSeedEns && !s->partials;
    default: return false;
  }
}

static int mlp_backward_dz_impl(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                                const osrl_mlp_grads_t* g, const osrl_mlp_tail_t* tail, void* stream,
                                const osrl_mlp_seed_t* seed = nullptr) {
  if (!valid_net(net) || !saved || !g || rows < 1) return -1;
  const bool seeded = seed && seed->kind != OSRL_SEED_NONE;
  if (seeded && !seed_ok(seed, net, saved)) return -1;
  for (int e = 0; e < net->n_nets; ++e) {
    if (!g->dy[e] && !seeded) return -1;
    for (int l = g->dx[e] ? 0 : 1; l < net->n_layers; ++l)
      if (!net->Wb[e][l]) return -1;
    for (int l = 0; l < net->n_layers; ++l)
      if (!saved->h[e][l] && (l < net->n_layers - 1 || net->acts[l] != OSRL_ACT_ID)) return -1;
    if (g->dx[e] && (g->dx_cols < 1 || g->dx_col0 < 0 || g->dx_col0 + g->dx_cols > net->dims[0])) return -1;
  }
  const bool want_tail = tail && tail->kind != OSRL_TAIL_NONE;
  if (want_tail && (tail->kind != OSRL_TAIL_VAE_LATENT_BWD || tail->L < 1 || !g->dx[0] || g->dx_cols != tail->L ||
                    !tail->eps || !tail->head || !tail->out))
    return -1;
  BwdArgs a{};
  a.net = *net;
  a.saved = *saved;
  a.g = *g;
  a.rows = rows;
  a.tail = osrl_mlp_tail_t{};
  a.seed = osrl_mlp_seed_t{};
  if (seeded) a.seed = *seed;
  const TileChoice t = choose_tile(net, rows, g->dx_cols);
  a.lda = t.lda;
  // the kernel's dX step leaves the slice in LDS only in its split-K form (mlp_bwd_dz_kernel: nblk <= 2 && nk >= 4 &&
  // lda >= 16 * NW); otherwise the tail is its own launch behind this one
  const bool fused = want_tail && (g->dx_cols + 15) / 16 <= 2 && round16h(net->dims[1]) / 16 >= 4 && t.lda >= 16 * t.nw;
  if (fused) {
    a.tail = *tail;
    a.tail.inv_rows_ = 1.0f / (float)(tail->rows_global > 0 ? tail->rows_global : rows);
  }
  if (want_tail && !fused) {
    const int rc = mlp_backward_dz_impl(net, rows, saved, g, nullptr, stream, seed);
    if (rc != 0) return rc;
    return osrl_vae_latent_bwd(tail->head, tail->eps, g->dx[0], rows, tail->L, tail->beta, tail->rows_global, tail->out,
                               stream);
  }
  OSRL_DISPATCH_TILE(mlp_bwd_dz_kernel, a, rows, net->n_nets, t, (hipStream_t)stream, 0);
}

extern "C" int osrl_mlp_backward_dz(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                                    const osrl_mlp_grads_t* g, void* stream) {
  return mlp_backward_dz_impl(net, rows, saved, g, nullptr, stream);
}

extern "C" int osrl_mlp_backward_dz_tail(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                                         const osrl_mlp_grads_t* g, const osrl_mlp_tail_t* tail, void* stream) {
  return mlp_backward_dz_impl(net, rows, saved, g, tail, stream);
}

extern "C" int osrl_mlp_backward_dz_seed(const osrl_mlp_t* net, int32_t rows, const osrl_mlp_acts_t* saved,
                                         const osrl_mlp_grads_t* g, const osrl_mlp_tail_t* tail,
                                         const osrl_mlp_seed_t* seed, void* stream) {
  return mlp_backward_dz_impl(net, rows, saved, g, tail, stream, seed);
}



// ---- host side of mlp_step_kernel ---------------------------------------------------------------------------------
template <int NCB>
static int launch_step(const StepArgs& k, size_t lds_bytes, hipStream_t stream) {
  const void* dev_args = osrl_argmem::slot(k);
  hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(mlp_step_kernel_p<NCB>)
                                              : reinterpret_cast<const void*>(mlp_step_kernel<NCB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  if (dev_args)
    hipLaunchKernelGGL(mlp_step_kernel_p<NCB>, dim3(k.n_wg), dim3(kStepThreads), lds_bytes, stream, dev_args);
  else
    hipLaunchKernelGGL(mlp_step_kernel<NCB>, dim3(k.n_wg), dim3(kStepThreads), lds_bytes, stream, k);
  return (int)hipGetLastError();
}

#ifdef OSRL_STEP_STAMPS
extern "C" int osrl_debug_step_stamps(long long* host_out /* [OSRL_STEP_MAX_WG][16] */) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_stamp), sizeof(long long) * OSRL_STEP_MAX_WG * 16);
}
extern "C" int osrl_debug_step_phases(long long* host_out /* [OSRL_STEP_MAX_WG][2][16] */) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_step_phase), sizeof(long long) * OSRL_STEP_MAX_WG * 2 * 16);
}
#endif

extern "C" int osrl_mlp_regress_step(const osrl_mlp_step_t* s, void* stream) {
  if (!s || !s->st || !valid_net(&s->net) || !s->target || !s->entries || !s->work || !s->p || !s->m || !s->v || !s->ws)
    return -1;
  const osrl_mlp_t* net = &s->net;
  const int L = net->n_layers, rows = s->in.rows;
  if (rows < 1 || s->in.d0 + s->in.d1 != net->dims[0] || !s->in.src0 || (s->in.d1 > 0 && !s->in.src1) || s->n_work < 1)
    return -1;
  if (!s->acts.x || !s->grads.dy[0] || s->grads.dx[0] || (s->tile_blocks != 2 && s->tile_blocks != 4)) return -1;
  for (int l = 0; l < L; ++l)
    if (!s->acts.h[0][l] || !s->grads.dz[0][l] || !net->Wf[0][l] || !net->b[0][l] || (l > 0 && !net->Wb[0][l])) return -1;
  if ((s->map_f && !s->pf) || (s->map_b && !s->pb)) return -1;
  osrl_gather::GatherArgs ga{};
  if (s->n_fields < 0 || s->n_fields > OSRL_MAX_FIELDS ||
      !osrl_gather::fill(ga, s->n_fields, s->src, s->dst, s->width, s->scale, s->n_rows, rows, s->gather_seed, s->gather_stream,
                         s->st) ||
      (s->n_fields > 0 && s->n_rows < 1))
    return -1;
  // the shapes the fused launch is built for: everything else keeps the separate launches
  const TileChoice t = choose_tile(net, rows, 0);
  const int n_tiles = (rows + 15) / 16;
  if (net->n_nets != 1 || t.nw != 8 || t.nrb != 1 || n_tiles > OSRL_STEP_MAX_WG || s->n_work > OSRL_STEP_MAX_WG)
    return OSRL_E_UNSUPPORTED;
  StepArgs k{};
  k.fwd.net = *net;
  k.fwd.in = s->in;
  k.fwd.out = s->acts;
  k.fwd.lda = t.lda;
  k.fwd.tail = osrl_mlp_tail_t{};
  k.bwd.net = *net;
  k.bwd.saved = s->acts;
  k.bwd.g = s->grads;
  k.bwd.rows = rows;
  k.bwd.lda = t.lda;
  k.bwd.tail = osrl_mlp_tail_t{};
  k.bwd.seed = osrl_mlp_seed_t{};
  k.gather = ga;
  k.st = s->st;
  k.stats_cur = s->stats_cur;
  k.ring = s->ring;
  k.target = s->target;
  k.du = const_cast<float*>(s->grads.dy[0]);
  k.stat = s->stat;
  k.entries = s->entries;
  k.work = s->work;
  k.p = s->p;
  k.m = s->m;
  k.v = s->v;
  k.map_f = s->map_f;
  k.map_b = s->map_b;
  k.pf = s->pf;
  k.pb = s->pb;
  k.ws = s->ws;
  k.beta1 = s->beta1;
  k.beta2 = s->beta2;
  k.lr = s->lr;
  k.eps = s->eps;
  const int64_t n = s->n_global > 0 ? s->n_global : (int64_t)rows * net->dims[L];
  k.inv_n = 1.0f / (float)n;
  k.warmup = s->warmup;
  k.n_stats = s->n_stats;
  k.ring_len = s->ring_len > 0 ? s->ring_len : 1;
  k.n_tiles = n_tiles;
  k.n_work = s->n_work;
  k.n_wg = n_tiles > s->n_work ? n_tiles : s->n_work;
  k.rows = rows;
  k.tile_blocks = s->tile_blocks == 2 ? 2 : 4;
  const size_t lds_rows = (size_t)16 * t.lda * sizeof(float);
  const size_t lds_bytes = lds_rows > dwt_lds<4>() ? lds_rows : dwt_lds<4>();
  if (t.ncb == 2) return launch_step<2>(k, lds_bytes, (hipStream_t)stream);
  return launch_step<4>(k, lds_bytes, (hipStream_t)stream);
}

// the register-streamed tile kernel (linear_kernel): any M, K <= 1024, any N
static int launch_linear_tiles(const float* A, int64_t lda, int32_t M, int32_t K, const float* P, int32_t Np, int32_t col0,
                               int32_t N, const float* bias, const float* resid, int64_t ldr, float* Y, int64_t ldy,
                               void* stream) {
  LinArgs a;
  a.A = A; a.P = P; a.bias = bias; a.resid = resid; a.Y = Y;
  a.lda_g = lda; a.ldr = ldr; a.ldy = ldy;
  a.M = M; a.K = K; a.N = N; a.Np = Np; a.col0 = col0;
  const int nblk = (N + 15) / 16;
  const int ncb = nblk >= 16 ? 4 : ((nblk + 3) / 4 <= 1 ? 1 : (nblk + 3) / 4 <= 2 ? 2 : (nblk + 3) / 4 <= 4 ? 4 : 7);
  const int gcols = nblk >= 16 ? 256 : nblk * 16;
  const int wmax = round16h(K) > gcols ? round16h(K) : gcols;
  a.lda = (wmax < 64 ? 64 : wmax) + 8;
  const int nrb = (K > 512 || ncb == 7) ? 1 : 2;
  const int BM = 16 * nrb;
  const size_t lds_bytes = (size_t)BM * a.lda * sizeof(float);
  dim3 grid((M + BM - 1) / BM, (nblk + 4 * ncb - 1) / (4 * ncb), 1);
  (void)hipGetLastError();
#define OSRL_LIN_LAUNCH(R, C)                                                                                   \
  do {                                                                                                          \
    if (lds_bytes > 64 * 1024)                                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_kernel<R, C>),                           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                    \
    hipLaunchKernelGGL((linear_kernel<R, C>), grid, dim3(256), lds_bytes, (hipStream_t)stream, a);              \
  } while (0)
  if (nrb == 2) {
    if (ncb == 1) OSRL_LIN_LAUNCH(2, 1); else if (ncb == 2) OSRL_LIN_LAUNCH(2, 2); else OSRL_LIN_LAUNCH(2, 4);
  } else {
    if (ncb == 1) OSRL_LIN_LAUNCH(1, 1); else if (ncb == 2) OSRL_LIN_LAUNCH(1, 2);
    else if (ncb == 4) OSRL_LIN_LAUNCH(1, 4); else OSRL_LIN_LAUNCH(1, 7);
  }
#undef OSRL_LIN_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int osrl_linear(const float* A, int64_t lda, int32_t M, int32_t K, const float* P, int32_t Np, int32_t col0,
                           int32_t N, const float* bias, const float* resid, int64_t ldr, float* Y, int64_t ldy,
                           void* stream) {
  if (!A || !P || !Y || M < 1 || K < 1 || K > 1024 || N < 1 || Np < 16) return -1;
  if (M >= 4096 && (K & 15) == 0 && (N & 255) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0) {
    LinBigArgs b;
    b.A = A; b.P = P; b.bias = bias; b.resid = resid; b.Y = Y;
    b.lda_g = lda; b.ldr = ldr; b.ldy = ldy;
    b.M = M; b.K = K; b.N = N; b.Np = Np; b.col0 = col0;
    (void)hipGetLastError();
    // 128-row x 256-column tiles on the 512 resident workgroups (2 per CU): a ragged last round costs a whole one.
    // At M = 81920, N = 256 (five of the eight GEMMs of a CDT block) that is 640 tiles = 1.25 rounds.
    // the persistent 128 x 128-tile kernel where its shape conditions hold and the tiles spread evenly over the CUs
    if ((M & 127) == 0 && (K & 255) == 0 && (N & 127) == 0) {
      // per CURRENT device and stateless (ADVICE r4 / VERDICT r5: no per-process latch -- a process that drives a second
      // device, e.g. the CPX partitions of one MI355X, must see that device's CU count and opt ITS copy of the kernel in)
      int n_cu = 256;
      {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
          n_cu = v;
        (void)hipGetLastError();
      }
      const int row_tiles = M / 128, col_groups = N / 128;
      const long tiles = (long)row_tiles * col_groups;
      if (tiles % n_cu == 0 || tiles >= 8L * n_cu) {
        constexpr int kPersSlots = 4;
        constexpr size_t kPersLds = sizeof(float) * kPersSlots * (128 * 16 + 16 * 128);
        {  // (the form that is launched: the attribute belongs to the current device's copy of the kernel)
          hipError_t e0 = resid ? hipFuncSetAttribute(reinterpret_cast<const void*>(linear_pers_kernel<4, 2, kPersSlots, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersLds)
                                : hipFuncSetAttribute(reinterpret_cast<const void*>(linear_pers_kernel<4, 2, kPersSlots, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersLds);
          if (e0 != hipSuccess) return (int)e0;
          (void)hipGetLastError();
        }
        if (resid)
          hipLaunchKernelGGL((linear_pers_kernel<4, 2, kPersSlots, true>), dim3(n_cu), dim3(512), kPersLds,
                             (hipStream_t)stream, b, row_tiles, col_groups, n_cu);
        else
          hipLaunchKernelGGL((linear_pers_kernel<4, 2, kPersSlots, false>), dim3(n_cu), dim3(512), kPersLds,
                             (hipStream_t)stream, b, row_tiles, col_groups, n_cu);
        return (int)hipGetLastError();
      }
    }
    const long cols = N / 256, t128 = (long)((M + 127) / 128) * cols;
    // the 128 + 32-row form costs 1.85 tile times (the 32-row pass pays the per-k-step staging + barrier of a 128-row
    // one for a quarter of the MFMAs): only where it replaces two rounds by one.  (Sending the leftover rows to the
    // register-streamed tile kernel in a second launch measured the same: 6.96 ms of projections per step either way,
    // 7.43 ms with the ragged round.)
    const long t160 = (long)((M + 159) / 160) * cols;
    const double c128 = (double)((t128 + 511) / 512), c160 = 1.85 * (double)((t160 + 511) / 512);
    {  // 72 KB of dynamic LDS: opt the launched form in, per call (stateless across devices, like every sibling launcher)
      hipError_t e0 = (c160 < c128 * 0.95)
                          ? hipFuncSetAttribute(reinterpret_cast<const void*>(linear_big_kernel<1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLbLds)
                          : hipFuncSetAttribute(reinterpret_cast<const void*>(linear_big_kernel<0>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLbLds);
      if (e0 != hipSuccess) return (int)e0;
      (void)hipGetLastError();
    }
    if (c160 < c128 * 0.95)
      hipLaunchKernelGGL(linear_big_kernel<1>, dim3((M + 159) / 160, N / 256, 1), dim3(512), kLbLds,
                         (hipStream_t)stream, b);
    else
      hipLaunchKernelGGL(linear_big_kernel<0>, dim3((M + 127) / 128, N / 256, 1), dim3(512), kLbLds,
                         (hipStream_t)stream, b);
    return (int)hipGetLastError();
  }
  return launch_linear_tiles(A, lda, M, K, P, Np, col0, N, bias, resid, ldr, Y, ldy, stream);
}

extern "C" int osrl_pack_weights(const float* src_flat, float* pf, float* pb, const osrl_pack_entry_t* d_entries,
                                 int32_t n_entries, int32_t max_elems, void* stream) {
  if (!src_flat || !d_entries || n_entries < 1 || (!pf && !pb)) return -1;
  int bx = (max_elems + 255) / 256;
  bx = bx < 1 ? 1 : bx > 64 ? 64 : bx;
  (void)hipGetLastError();
  hipLaunchKernelGGL(pack_kernel, dim3(bx, n_entries), dim3(256), 0, (hipStream_t)stream, src_flat, pf, pb, d_entries);
  return (int)hipGetLastError();
}
