// mlp_nb.hip -- the N*B-row forward of the fused MLP (80-row tiles, one workgroup per CU) for gfx950.
// Split from mlp.hip (round 4) so that the two units compile side by side; design notes at the kernel below and in
// mlp.hip's header.
#include "mlp_common.h"
#include "trace.h"

namespace {

// ---- N*B-row forward: 80-row tiles, ONE 4-wave workgroup per CU ------------------------------------------
// The inference-only launches of a step (CPQ: target cost critics and the VAE encoder on the N*B = 20480 sampled
// rows; BCQ-Lag / BEAR-Lag: decoder, actor, target critics on N*B rows) carry 69 % of the step's FLOPs.  With the tile
// kernel above, 2-3 workgroups share a CU and the k-loop sits at 53-58 % of the fp32 MFMA roof; tools/loop_probe2.hip
// shows why a different shape wins: ONE wave per SIMD with an 80-row x (64 | 112)-column register tile (80-140
// accumulator registers out of the wave's 512) runs the same loop at 93-96 % -- 5 ds_read_b128 + 4-7
// global_load_dwordx4 feed 80-140 MFMAs per k-step, weight traffic per FLOP is 2.5-5x lower than with 16/32-row tiles,
// and with the in-step order pinned (loads of the next step first) one k-step of MFMAs (2560-4480 cycles) covers the
// L2 latency with no second wave needed.
// This kernel is that loop plus the least it needs around it: the input tile is staged with every load in flight at
// once, each wide layer is  bias-initialised accumulators -> k-loop -> barrier -> activation into the LDS tile (in
// place), the narrow head (<= 32 outputs, always the last layer) splits K over the 4 waves with all of a wave's weight
// fragments requested together in front of its k-steps, partial tiles meet in LDS and go straight to global.
// The wide layers compute the TRANSPOSED tile: the packed weight fragment is the MFMA's A operand and the activation
// fragment its B operand (both are "16 lanes x 4 consecutive k", so loads and packing are those of the other kernels),
// which leaves a lane with out[row = lane & 15][4 consecutive columns] -- exactly the row-major float4 the next
// layer's fragment read wants.  The epilogue is then one ds_write_b128 per 16x16 tile in the (conflict-free) pattern
// of the fragment reads, instead of four ds_write_b32 down a column; with the one-instruction ReLU and no column select
// for widths that are multiples of 16 the epilogue of a 400-wide layer went from 4.85k to 2.7k cycles
// (profiles/r2_mlp_phase_nb.txt).  Same products, same accumulation order per output: same bits.
// Eligibility (host): no saved activations, hidden layers of 13-16 (NCB = 4) or 25-28 (NCB = 7) column blocks, narrow
// last layer with >= 4 k-steps; anything else takes mlp_fwd_kernel.
struct NbArgs {
  osrl_mlp_t net;
  osrl_rows_t in;
  float* y[OSRL_MAX_NETS];
  int32_t lda, kl_L;
  float* kl;  // OSRL_TAIL_VAE_KL: [rows] per-row KL of net 0's (mean | log_std) output, or NULL
  // round 6: a ONE-output head (every Q network: 256 -> 1) is not run as a layer -- the last wide layer's epilogue takes
  // the dot product of its activated accumulators with the head's weight column straight from the registers (nb_head_dot):
  // no activation write-back, no 16-column MFMA pass for one real column, no partial tiles, two barriers fewer
  int32_t fuse_head;
};
constexpr float kNbLsMin = -4.0f, kNbLsMax = 15.0f;  // net.py:325 (== kVaeLsMin / kVaeLsMax of glue.hip)

#ifndef OSRL_NB8_ADB
#define OSRL_NB8_ADB false  // 8-wave form: activation fragments single buffered (see nb_mm)
#endif
#ifndef OSRL_NB_INTERLEAVE
#define OSRL_NB_INTERLEAVE 1
#endif
// Row blocks per tile: 5 (80 rows) here; mlp_nb64.hip compiles this file a second time with OSRL_NB_RB = 4 (64-row tiles:
// the 67.6 KB activation tile of a 256-wide net then allows TWO workgroups per CU) and its own launcher name
#ifndef OSRL_NB_RB
#define OSRL_NB_RB 5
#endif
#ifndef OSRL_NB_LAUNCH
#define OSRL_NB_LAUNCH osrl_launch_fwd_nb
#endif
constexpr int kNbRb = OSRL_NB_RB;
constexpr int kNbRows = 16 * OSRL_NB_RB;

// Biases of a wave's column blocks, branch-free (clamped address + select): every load is in flight at once.  The
// obvious "col < N ? bias[col] : 0.f" compiles to one exec-masked global_load + s_waitcnt vmcnt(0) PER BLOCK, i.e.
// 7 serial L2 round trips in front of every wide layer: ~9k of the ~10k cycles a layer took beyond its MFMAs
// (profiles/r2_mlp_phase_nb.txt: the same excess for the 5-step and the 25-step layer).
// nb_bias only REQUESTS the values (columns past N read bias[0]); nb_bias_acc, called after the first weight /
// activation fragments are requested and fenced from them by a scheduling barrier, turns them into the accumulators'
// start values (acc = bias: no add in the epilogue), so the bias round trip and the first weight round trip overlap.
// (Accumulator tiles are TRANSPOSED, see nb_mm: a lane holds columns 4 * (lane >> 4) + 0..3 of column block c.)
template <int CNT>
__device__ __forceinline__ void nb_bias(const float* __restrict__ bias, int col0, int N, int lane, f32x4 (&braw)[CNT]) {
#pragma unroll
  for (int c = 0; c < CNT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = col0 + c * 16 + 4 * (lane >> 4) + r;
      braw[c][r] = bias[col < N ? col : 0];
    }
}
template <int CNT, int R>
__device__ __forceinline__ void nb_bias_acc(const f32x4 (&braw)[CNT], int col0, int N, int lane, f32x4 (&acc)[R][CNT]) {
#pragma unroll
  for (int c = 0; c < CNT; ++c) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = col0 + c * 16 + 4 * (lane >> 4) + r < N ? braw[c][r] : 0.f;
#pragma unroll
    for (int rb = 0; rb < R; ++rb) acc[rb][c] = bv;
  }
}

// SHARED SRC0 ROWS (osrl_rows_t.share0, second session of round 6): the N*B rows of CPQ's OOD launches are the B observations
// N times over beside N different sampled actions (cpq.py:164-176, row = n B + b), so the observation part of layer 0 is the
// same for the N copies of an observation.  A tile of this form is [kNbRb samples] x [16 observations] -- row block rb is
// sample s0 + rb of the SAME 16 observations -- and layer 0 runs in two phases: (A) the k-steps that lie wholly inside the
// observation columns (kc0 of them: 4 of 5 at (76, 2)) on ONE row block, bias-initialised, every weight fragment used for 16
// rows instead of 80; (B) every row block's accumulators start from that result and walk the remaining k-steps as before.
// Same products; a row's sum is formed as (bias + first 16 kc0 terms in k order) + rest instead of one rotated walk -- a few
// ulp from the plain form, which is why the plans use it for no_grad launches only.  (First built as a per-observation
// prefix matrix computed by osrl_linear launches in front of the launch: the kernels gained 8 us each, the three small
// launches cost 38 -- gpurun_out/r6prefix.)
template <int CNT, int S>
__device__ __forceinline__ void nb_share_step(const float* __restrict__ P, int Np, unsigned lane_off, const float* arow, int ks,
                                              int kc0, f32x4 (&b)[2][CNT], f32x4 (&a0)[2], f32x4 (&ao)[1][CNT]) {
  const int kn = ks + 1 < kc0 ? ks + 1 : ks;  // last step: a harmless re-load
  const float* __restrict__ Pk = P + (size_t)kn * 16 * Np;
#pragma unroll
  for (int c = 0; c < CNT; ++c) b[S ^ 1][c] = load_bp_s(Pk, lane_off + c * 256);
  a0[S ^ 1] = *reinterpret_cast<const f32x4*>(arow + kn * 16);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < CNT; ++c) ao[0][c] = EXP_MFMA(b[S][c][t], a0[S][t], ao[0][c]);
}
// phase A: ao = bias + sum over k-steps [0, kc0) of row block 0 (b / a0: scratch fragment sets)
template <int CNT>
__device__ __forceinline__ void nb_share_acc(const float* __restrict__ P, int Np, unsigned lane_off, const float* arow, int kc0,
                                             f32x4 (&b)[2][CNT], f32x4 (&ao)[1][CNT]) {
  f32x4 a0[2];
#pragma unroll
  for (int c = 0; c < CNT; ++c) b[0][c] = load_bp_s(P, lane_off + c * 256);
  a0[0] = *reinterpret_cast<const f32x4*>(arow);
  int ks = 0;
  for (; ks + 2 <= kc0; ks += 2) {
    nb_share_step<CNT, 0>(P, Np, lane_off, arow, ks, kc0, b, a0, ao);
    nb_share_step<CNT, 1>(P, Np, lane_off, arow, ks + 1, kc0, b, a0, ao);
  }
  if (ks < kc0) nb_share_step<CNT, 0>(P, Np, lane_off, arow, ks, kc0, b, a0, ao);
}

// ADB: the activation fragments are double buffered (next step's ds_reads issued under this step's MFMAs: one wave per
// SIMD has nothing else to hide them behind).  The 8-wave form reads them single buffered at the end of a step -- the
// SIMD's other wave covers the LDS round trip -- which is what brings it under 160 registers per lane.
template <int CNT, bool ADB = true, bool PRE = false>
__device__ __forceinline__ void nb_mm(const float* lds, int lda, int nk_all, const float* __restrict__ P, int Np, int col0,
                                      int N, const f32x4 (&braw)[CNT], f32x4 (&acc)[kNbRb][CNT], int pl, const int kc0_in) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  const float* arow = lds + m * lda + 4 * kq;
  const unsigned lane_off = (unsigned)((kq * Np + col0 + m) * 16);  // bytes
  const int kc0 = PRE ? kc0_in : 0, nk = nk_all - kc0;  // (shared src0 rows: phase B walks the k-steps from kc0 on)
  const int rot = k_rot(nk);
  f32x4 b[2][CNT], a[ADB ? 2 : 1][kNbRb];
  f32x4 ao[1][CNT];
  if constexpr (PRE) {  // phase A on row block 0 (every row block of the tile holds the same src0 rows)
    nb_bias_acc<CNT, 1>(braw, col0, N, lane, ao);
    nb_share_acc<CNT>(P, Np, lane_off, arow, kc0, b, ao);
  }
  {
    const int k0 = k_at(0, rot, nk, kc0);
    const float* __restrict__ Pk = P + (size_t)k0 * 16 * Np;
#pragma unroll
    for (int c = 0; c < CNT; ++c) b[0][c] = load_bp_s(Pk, lane_off + c * 256);
#pragma unroll
    for (int rb = 0; rb < kNbRb; ++rb) a[0][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + k0 * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PRE) {
#pragma unroll
    for (int rb = 0; rb < kNbRb; ++rb)
#pragma unroll
      for (int c = 0; c < CNT; ++c) acc[rb][c] = ao[0][c];
  } else {
    nb_bias_acc<CNT, kNbRb>(braw, col0, N, lane, acc);
  }
  if (pl == 1) { PHASE_STAMP(14); }  // debug build: layer 1's k-loop starts here
  auto step = [&](auto s_c, int kc) {
    constexpr int s = decltype(s_c)::value;
    constexpr int sa = ADB ? s : 0, sn = ADB ? (s ^ 1) : 0;
    const int kn = k_at(kc + 1 < nk ? kc + 1 : kc, rot, nk, kc0);  // last step: a harmless re-load
    const float* __restrict__ Pk = P + (size_t)kn * 16 * Np;
#pragma unroll
    for (int c = 0; c < CNT; ++c) b[s ^ 1][c] = load_bp_s(Pk, lane_off + c * 256);
    if constexpr (ADB) {
#pragma unroll
      for (int rb = 0; rb < kNbRb; ++rb) a[sn][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + kn * 16);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int rb = 0; rb < kNbRb; ++rb)
          acc[rb][c] = EXP_MFMA(b[s][c][t], a[sa][rb][t], acc[rb][c]);
    if constexpr (!ADB) {
#pragma unroll
      for (int rb = 0; rb < kNbRb; ++rb) a[0][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + kn * 16);
      constexpr int kPerB = (4 * kNbRb * CNT) / (CNT + 1);
#pragma unroll
      for (int i = 0; i < CNT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, kPerB, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kNbRb * CNT - kPerB * CNT, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, kNbRb, 0);
      return;
    }
#if OSRL_NB_INTERLEAVE
    // next step's loads one at a time, each followed by a few of THIS step's MFMAs: with one wave per SIMD nothing else
    // can fill the MFMA pipe while the ~35 address / load instructions of a step issue (540 cycles per 16-deep k-step
    // with all of them in front of the MFMAs)
    constexpr int kPer = (4 * kNbRb * CNT) / (CNT + kNbRb + 1);
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
#pragma unroll
    for (int i = 0; i < kNbRb; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kNbRb * CNT - kPer * (CNT + kNbRb), 0);
#else
    __builtin_amdgcn_sched_group_barrier(0x020, CNT, 0);                // VMEM reads of the next step first
    __builtin_amdgcn_sched_group_barrier(0x100, kNbRb, 0);              // its DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kNbRb * CNT, 0);    // then this step's MFMAs
#endif
  };
  using std::integral_constant;
  int kc = 0;
  for (; kc + 2 <= nk; kc += 2) {
    step(integral_constant<int, 0>{}, kc);
    step(integral_constant<int, 1>{}, kc + 1);
  }
  if (kc < nk) step(integral_constant<int, 0>{}, kc);
}

// R row blocks of CNT column blocks.  RAGGED (N not a multiple of 16): columns past N are written as zeros, the k
// padding of the next layer; otherwise the select is compiled out (the epilogue is VALU-bound: accumulator read +
// activation + select per element was ~36 cycles x 124 elements per wave and layer).
template <int CNT, int R, int ACT, bool RAGGED>
__device__ __forceinline__ void nb_epilogue_core(float* lds, int lda, const f32x4 (&acc)[R][CNT], int cb0, int rb0, int N,
                                                 int lane) {
#pragma unroll
  for (int c = 0; c < CNT; ++c) {
    const int col = (cb0 + c) * 16 + 4 * (lane >> 4);
    float* dst = lds + (rb0 * 16 + (lane & 15)) * lda + col;
#pragma unroll
    for (int rb = 0; rb < R; ++rb) {
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = act_fwd(ACT, acc[rb][c][r]);
        if (RAGGED) v[r] = col + r < N ? v[r] : 0.f;
      }
      *reinterpret_cast<f32x4*>(dst + rb * 16 * lda) = v;
    }
  }
}
template <int CNT, int R>
__device__ __forceinline__ void nb_epilogue(float* lds, int lda, const f32x4 (&acc)[R][CNT], int cb0, int rb0, int N,
                                            int act, int lane) {
  const bool ragged = (N & 15) != 0;
  if (act == OSRL_ACT_RELU) {
    if (ragged) nb_epilogue_core<CNT, R, OSRL_ACT_RELU, true>(lds, lda, acc, cb0, rb0, N, lane);
    else nb_epilogue_core<CNT, R, OSRL_ACT_RELU, false>(lds, lda, acc, cb0, rb0, N, lane);
  } else if (act == OSRL_ACT_TANH) {
    nb_epilogue_core<CNT, R, OSRL_ACT_TANH, true>(lds, lda, acc, cb0, rb0, N, lane);
  } else {
    nb_epilogue_core<CNT, R, OSRL_ACT_ID, true>(lds, lda, acc, cb0, rb0, N, lane);
  }
}

// The one-output head on the last wide layer's accumulators.  A lane holds out[row = lane & 15][4 consecutive columns] of
// each (row block, column block) register tile (nb_mm: transposed tiles), the head's packed weight column is
// PF[k / 4][n = 0][k % 4] with 16 padded outputs, i.e. the four weights of a lane's columns are ONE aligned float4 at
// hw + (col / 4) * 64: partial[rb] = sum over the lane's columns of act(acc) * w, summed over the four lanes of a row
// (xor 16, 32), one partial per wave and row into `part` [waves][rows]; the caller sums the waves in order.  (hwv: requested
// by the caller in front of the k-loop.)
template <int CNT>
__device__ __forceinline__ void nb_head_dot(const f32x4 (&acc)[kNbRb][CNT], const f32x4 (&hwv)[CNT], int act, int lane,
                                            float* __restrict__ part) {
  float rd[kNbRb];
#pragma unroll
  for (int rb = 0; rb < kNbRb; ++rb) rd[rb] = 0.f;
  auto run = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int c = 0; c < CNT; ++c)
#pragma unroll
      for (int rb = 0; rb < kNbRb; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) rd[rb] = __builtin_fmaf(act_fwd(ACT, acc[rb][c][r]), hwv[c][r], rd[rb]);
  };
  if (act == OSRL_ACT_RELU) run(std::integral_constant<int, OSRL_ACT_RELU>{});
  else if (act == OSRL_ACT_TANH) run(std::integral_constant<int, OSRL_ACT_TANH>{});
  else run(std::integral_constant<int, OSRL_ACT_ID>{});
#pragma unroll
  for (int rb = 0; rb < kNbRb; ++rb) {
    rd[rb] += __shfl_xor(rd[rb], 16);
    rd[rb] += __shfl_xor(rd[rb], 32);
    if (lane < 16) part[rb * 16 + lane] = rd[rb];
  }
}

template <int CNT, bool ADB = true, bool PRE = false>
__device__ __forceinline__ void nb_wide_layer(float* lds, int lda, int K, int N, const float* __restrict__ P,
                                              const float* __restrict__ bias, int act, int cb0, int lane, int pl,
                                              const int kc0, const float* __restrict__ hw = nullptr,
                                              float* __restrict__ part = nullptr) {
  (void)pl;  // layer number, for the debug build's phase stamps only
  f32x4 braw[CNT];
  nb_bias<CNT>(bias, cb0 * 16, N, lane, braw);
  f32x4 hwv[CNT];
  if (hw) {  // (wave-uniform) the head's weights for this lane's columns: in flight under the k-loop
#pragma unroll
    for (int c = 0; c < CNT; ++c)
      hwv[c] = *reinterpret_cast<const f32x4*>(hw + (size_t)((cb0 + c) * 4 + (lane >> 4)) * 64);
  }
  f32x4 acc[kNbRb][CNT];
  nb_mm<CNT, ADB, PRE>(lds, lda, round16(K) >> 4, P, round16(N), cb0 * 16, N, braw, acc, pl, kc0);
  PHASE_STAMP(2 + 4 * pl);
  if (hw) {  // the tile is not written again: no barrier in front, the caller's barrier behind
    nb_head_dot<CNT>(acc, hwv, act, lane, part);
    return;
  }
  __syncthreads();  // every wave finished reading the previous activations
  PHASE_STAMP(3 + 4 * pl);
  nb_epilogue<CNT, kNbRb>(lds, lda, acc, cb0, 0, N, act, lane);
  PHASE_STAMP(4 + 4 * pl);
  __syncthreads();
  PHASE_STAMP(5 + 4 * pl);
}

// ---- 4q + 1 column blocks (400-wide layers: 25): every wave owns q blocks, the last block is SHARED by rows ----------
// Dealing 25 blocks as 7 + 6 + 6 + 6 makes the 7-block wave the layer's pace: 12 % over the mean.  Here wave 0 takes
// row blocks {0, 1} of the shared block, waves 1..3 one row block each: 32 / 31 / 31 / 31 register tiles.
// NX = row blocks of the shared column this wave owns (2: wave 0, 1: the others), starting at rbx0.
template <int CNT, int NX, bool ADB = true, bool PRE = false>
__device__ __forceinline__ void nb_mm_x(const float* lds, int lda, int nk_all, const float* __restrict__ P, int Np, int col0,
                                        int colx, int rbx0, int N, const f32x4 (&braw)[CNT], const f32x4 (&brawx)[1],
                                        f32x4 (&acc)[kNbRb][CNT], f32x4 (&xacc)[NX], int pl, const int kc0_in) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  const float* arow = lds + m * lda + 4 * kq;
  const unsigned lane_off = (unsigned)((kq * Np + col0 + m) * 16);  // bytes
  const unsigned lane_offx = (unsigned)((kq * Np + colx + m) * 16);
  const int kc0 = PRE ? kc0_in : 0, nk = nk_all - kc0;  // (shared src0 rows: phase B walks the k-steps from kc0 on)
  const int rot = k_rot(nk);
  const float* arowx = arow + rbx0 * 16 * lda;  // the shared column's row blocks (wave-uniform start)
  f32x4 b[2][CNT + 1], a[ADB ? 2 : 1][kNbRb], ax[ADB ? 2 : 1][NX];
  f32x4 ao[1][CNT + 1];
  if constexpr (PRE) {  // phase A on row block 0, this wave's CNT column blocks and the row-shared one (column offsets differ)
    f32x4 b1[1][CNT], bx[1][1];
    nb_bias_acc<CNT, 1>(braw, col0, N, lane, b1);
    nb_bias_acc<1, 1>(brawx, colx, N, lane, bx);
#pragma unroll
    for (int c = 0; c < CNT; ++c) ao[0][c] = b1[0][c];
    ao[0][CNT] = bx[0][0];
    f32x4 a0[2];
    auto ldw = [&](int k, int sI) {
      const float* __restrict__ Pk = P + (size_t)k * 16 * Np;
#pragma unroll
      for (int c = 0; c < CNT; ++c) b[sI][c] = load_bp_s(Pk, lane_off + c * 256);
      b[sI][CNT] = load_bp_s(Pk, lane_offx);
      a0[sI] = *reinterpret_cast<const f32x4*>(arow + k * 16);
    };
    auto stepA = [&](auto s_c, int ks) {
      constexpr int sI = decltype(s_c)::value;
      ldw(ks + 1 < kc0 ? ks + 1 : ks, sI ^ 1);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < CNT + 1; ++c) ao[0][c] = EXP_MFMA(b[sI][c][t], a0[sI][t], ao[0][c]);
    };
    ldw(0, 0);
    int ks = 0;
    for (; ks + 2 <= kc0; ks += 2) {
      stepA(std::integral_constant<int, 0>{}, ks);
      stepA(std::integral_constant<int, 1>{}, ks + 1);
    }
    if (ks < kc0) stepA(std::integral_constant<int, 0>{}, ks);
  }
  {
    const int k0 = k_at(0, rot, nk, kc0);
    const float* __restrict__ Pk = P + (size_t)k0 * 16 * Np;
#pragma unroll
    for (int c = 0; c < CNT; ++c) b[0][c] = load_bp_s(Pk, lane_off + c * 256);
    b[0][CNT] = load_bp_s(Pk, lane_offx);
#pragma unroll
    for (int rb = 0; rb < kNbRb; ++rb) a[0][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + k0 * 16);
#pragma unroll
    for (int i = 0; i < NX; ++i) ax[0][i] = *reinterpret_cast<const f32x4*>(arowx + i * 16 * lda + k0 * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PRE) {
#pragma unroll
    for (int rb = 0; rb < kNbRb; ++rb)
#pragma unroll
      for (int c = 0; c < CNT; ++c) acc[rb][c] = ao[0][c];
#pragma unroll
    for (int i = 0; i < NX; ++i) xacc[i] = ao[0][CNT];
  } else {
    nb_bias_acc<CNT, kNbRb>(braw, col0, N, lane, acc);
    f32x4 xa[NX][1];
    nb_bias_acc<1, NX>(brawx, colx, N, lane, xa);
#pragma unroll
    for (int i = 0; i < NX; ++i) xacc[i] = xa[i][0];
  }
  if (pl == 1) { PHASE_STAMP(14); }  // debug build: layer 1's k-loop starts here
  auto step = [&](auto s_c, int kc) {
    constexpr int s = decltype(s_c)::value;
    constexpr int sa = ADB ? s : 0, sn = ADB ? (s ^ 1) : 0;
    const int kn = k_at(kc + 1 < nk ? kc + 1 : kc, rot, nk, kc0);  // last step: a harmless re-load
    const float* __restrict__ Pk = P + (size_t)kn * 16 * Np;
#pragma unroll
    for (int c = 0; c < CNT; ++c) b[s ^ 1][c] = load_bp_s(Pk, lane_off + c * 256);
    b[s ^ 1][CNT] = load_bp_s(Pk, lane_offx);
    if constexpr (ADB) {
#pragma unroll
      for (int rb = 0; rb < kNbRb; ++rb) a[sn][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + kn * 16);
#pragma unroll
      for (int i = 0; i < NX; ++i) ax[sn][i] = *reinterpret_cast<const f32x4*>(arowx + i * 16 * lda + kn * 16);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int rb = 0; rb < kNbRb; ++rb)
          acc[rb][c] = EXP_MFMA(b[s][c][t], a[sa][rb][t], acc[rb][c]);
#pragma unroll
      for (int i = 0; i < NX; ++i)
        xacc[i] = EXP_MFMA(b[s][CNT][t], ax[sa][i][t], xacc[i]);
    }
    if constexpr (!ADB) {
#pragma unroll
      for (int rb = 0; rb < kNbRb; ++rb) a[0][rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + kn * 16);
#pragma unroll
      for (int i = 0; i < NX; ++i) ax[0][i] = *reinterpret_cast<const f32x4*>(arowx + i * 16 * lda + kn * 16);
      constexpr int kTotB = 4 * (kNbRb * CNT + NX), kPerB = kTotB / (CNT + 2);
#pragma unroll
      for (int i = 0; i < CNT + 1; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, kPerB, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kTotB - kPerB * (CNT + 1), 0);
      __builtin_amdgcn_sched_group_barrier(0x100, kNbRb + NX, 0);
      return;
    }
#if OSRL_NB_INTERLEAVE
    constexpr int kTot = 4 * (kNbRb * CNT + NX), kPer = kTot / (CNT + 1 + kNbRb + NX + 1);
#pragma unroll
    for (int i = 0; i < CNT + 1; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
#pragma unroll
    for (int i = 0; i < kNbRb + NX; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, kPer, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, kTot - kPer * (CNT + 1 + kNbRb + NX), 0);
#else
    __builtin_amdgcn_sched_group_barrier(0x020, CNT + 1, 0);                    // VMEM reads of the next step first
    __builtin_amdgcn_sched_group_barrier(0x100, kNbRb + NX, 0);                 // its DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * (kNbRb * CNT + NX), 0);     // then this step's MFMAs
#endif
  };
  using std::integral_constant;
  int kc = 0;
  for (; kc + 2 <= nk; kc += 2) {
    step(integral_constant<int, 0>{}, kc);
    step(integral_constant<int, 1>{}, kc + 1);
  }
  if (kc < nk) step(integral_constant<int, 0>{}, kc);
}

template <int CNT, int NX, bool ADB = true, bool PRE = false>
__device__ __forceinline__ void nb_wide_layer_x(float* lds, int lda, int K, int N, const float* __restrict__ P,
                                                const float* __restrict__ bias, int act, int cb0, int cbx, int rbx0,
                                                int lane, int pl, const int kc0) {
  (void)pl;
  f32x4 acc[kNbRb][CNT], xacc[NX];
  f32x4 braw[CNT], brawx[1];
  nb_bias<CNT>(bias, cb0 * 16, N, lane, braw);
  nb_bias<1>(bias, cbx * 16, N, lane, brawx);
  nb_mm_x<CNT, NX, ADB, PRE>(lds, lda, round16(K) >> 4, P, round16(N), cb0 * 16, cbx * 16, rbx0, N, braw, brawx, acc, xacc, pl, kc0);
  PHASE_STAMP(2 + 4 * pl);
  __syncthreads();  // every wave finished reading the previous activations
  PHASE_STAMP(3 + 4 * pl);
  nb_epilogue<CNT, kNbRb>(lds, lda, acc, cb0, 0, N, act, lane);
  {
    f32x4 xa[NX][1];
#pragma unroll
    for (int i = 0; i < NX; ++i) xa[i][0] = xacc[i];
    nb_epilogue<1, NX>(lds, lda, xa, cbx, rbx0, N, act, lane);
  }
  PHASE_STAMP(4 + 4 * pl);
  __syncthreads();
  PHASE_STAMP(5 + 4 * pl);
}

// SHARED: every wide layer has 4 (NCB - 1) + 1 column blocks (the 400-wide VAE encoder / decoder): NCB - 1 blocks per
// wave + the row-shared last block (nb_wide_layer_x); a separate instantiation, so that neither form carries the
// other's register footprint
// NW = waves per workgroup: 4 (one wave per SIMD, the shapes above) or 8 (two per SIMD, 400-wide layers only: every wave
// owns 3 column blocks, the 25th is shared by rows over waves 0..4 -- 16 / 15 register tiles per wave instead of 32 / 31,
// i.e. <= 160 registers per lane instead of 339: the chain's 8-wave workgroups (2 x 96 registers per SIMD) then fit on
// the CU beside this one, which at one 339-register wave per SIMD they do not; DESIGN.md section 3 "round 4")
template <int NCB, bool SHARED, int NW, class AR, bool LIST = false, bool PREFIX = false>
__device__ __forceinline__ void mlp_fwd_nb_body(AR a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 16 * kNbRb;
  __shared__ float s_head[SHARED ? 1 : NW * BM];  // fuse_head: per-wave partial dot products of the one-output head
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e = blockIdx.y, row0 = blockIdx.x * BM;
  int rows = a.in.rows;
  // tile row r -> row of the launch.  Plain: 16 kNbRb consecutive rows.  PREFIX (osrl_rows_t.share0, rows = n B + b): row
  // block rb = sample s0 + rb of the 16 src0 rows b0 .. b0 + 15 -- tiles are dealt [sample group][src0 group], every tile whole
  int sh_s0 = 0, sh_b0 = 0;
  if constexpr (PREFIX) {
    const int ogn = a.in.div0 >> 4, sg = blockIdx.x / ogn;
    sh_s0 = sg * kNbRb;
    sh_b0 = (blockIdx.x - sg * ogn) * 16;
  }
  auto grow = [&](int r) -> int {
    if constexpr (PREFIX) return (sh_s0 + (r >> 4)) * a.in.div0 + sh_b0 + (r & 15);
    else return row0 + r;
  };
  const int lda = a.lda, L = a.net.n_layers;
  if constexpr (LIST) {  // a row set chosen on the device (osrl_rows_t.row_list): its size is a device word, the grid is sized
    const int nd = a.in.n_rows_dev[0];  // for the list's capacity -- workgroups past the count leave before any barrier
    rows = nd < rows ? nd : rows;
    if (row0 >= rows) return;
  }
  WG_LOG(0);
  PHASE_STAMP(0);
  {  // ---- stage cat(src0[map0(r)], src1[map1(r)]) zero padded to a multiple of 16 columns: every load of the tile
     // is issued before the first LDS store.  16 lanes walk one row (64-byte segments), 16 rows per pass, 5 passes;
     // no per-element division (80 x K0p / 256 of them cost 23k cycles in the first version of this kernel).
    const int K0 = a.net.dims[0], K0p = round16(K0);
    const int d0 = a.in.d0, d1 = a.in.d1;
    const int cl = tid & 15, rl = tid >> 4;
    const float* __restrict__ s0 = a.in.src0;
    const float* __restrict__ s1 = a.in.src1 ? a.in.src1 : a.in.src0;
    constexpr int kColChunks = 8;  // K0 <= 128 (host-checked)
    // the row maps as straight-line code on values read ONCE: one unsigned division per (row, source) whose
    // reciprocal set-up is common to the five passes, selects instead of the three-way branch of map_row().  (With
    // map_row() inlined per pass the compiler re-read the descriptor from the kernel arguments in every branch arm:
    // ~20 s_load + s_waitcnt lgkmcnt(0) round trips in front of the tile's loads.)
    // The values are parked in VECTOR registers (the opaque asm makes them non-rematerialisable): there are plenty
    // before the accumulators exist, while the scalar file is full of layer descriptors by now.
    unsigned dv0 = a.in.map0 == OSRL_MAP_ID ? 1u : (unsigned)a.in.div0;
    unsigned dv1 = a.in.map1 == OSRL_MAP_ID ? 1u : (unsigned)a.in.div1;
    unsigned mod0 = a.in.map0 == OSRL_MAP_MOD, idn0 = a.in.map0 == OSRL_MAP_ID;
    unsigned mod1 = a.in.map1 == OSRL_MAP_MOD, idn1 = a.in.map1 == OSRL_MAP_ID;
    int d0v = d0, d1v = d1, rows_v = rows, K0v = K0;
    asm volatile("" : "+v"(dv0), "+v"(dv1), "+v"(mod0), "+v"(idn0), "+v"(mod1), "+v"(idn1));
    asm volatile("" : "+v"(d0v), "+v"(d1v), "+v"(rows_v), "+v"(K0v));
    auto mapped = [](unsigned r, unsigned mod, unsigned idn, unsigned dv) -> unsigned {
      const unsigned q = r / dv, rem = r - q * dv;  // dv == 1 for the identity map
      return mod ? rem : (idn ? r : q);
    };
    constexpr int kRowsPass = 4 * NW, kPasses = (BM + kRowsPass - 1) / kRowsPass;  // 16 rows x 5 passes | 32 x 3 (the last half empty)
    float v[kPasses][kColChunks];
#pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int gr = grow(p * kRowsPass + rl);
      const bool rok = gr < rows_v && (BM % kRowsPass == 0 || p * kRowsPass + rl < BM);
      unsigned grc = (unsigned)(rok ? gr : rows_v - 1);
      if constexpr (LIST) grc = (unsigned)a.in.row_list[grc];  // (one more dependent load in front of the tile's: this form only)
      const float* p0 = s0 + (size_t)mapped(grc, mod0, idn0, dv0) * d0v;
      const float* p1 = s1 + (size_t)mapped(grc, mod1, idn1, dv1) * d1v - d0v;
#pragma unroll
      for (int j = 0; j < kColChunks; ++j) {
        const int c = j * 16 + cl;
        const bool ok = rok && c < K0;
        const float* q = c < d0 ? p0 + c : p1 + c;
        v[p][j] = *(ok ? q : s0);      // (round 6, measured and not kept, profiles/r6_stagein_ab.txt / r6_tilewalk_ab.txt: requesting
        v[p][j] = ok ? v[p][j] : 0.f;  // only the chunks below the input's padded width behind uniform branches: C3 / C2 -1.6 %, the
      }                                // branches break the one-batch issue of the loads; workgroups WALKING several tiles with
    }                                  // the next tile's rows requested under the last wide layer: +60-80 registers, -1.5 .. -4 %)
#pragma unroll
    for (int p = 0; p < kPasses; ++p)
#pragma unroll
      for (int j = 0; j < kColChunks; ++j) {
        const int c = j * 16 + cl;
        if (c < K0p && (BM % kRowsPass == 0 || p * kRowsPass + rl < BM)) lds[(p * kRowsPass + rl) * lda + c] = v[p][j];
      }
    __syncthreads();
  }
  PHASE_STAMP(1);
  auto wide = [&](const int l, auto pre_tag) {  // one wide layer; pre_tag: layer 0 of a shared-src0-rows tile (nb_share_acc)
    constexpr bool PRE = decltype(pre_tag)::value;
    const int K = a.net.dims[l], N = a.net.dims[l + 1];
    const int nblk = (N + 15) >> 4;
    const int pre = PRE ? a.in.share_k16 : 0;  // phase A's k-steps (nb_share_acc)
    if constexpr (SHARED && NW == 8) {  // 8q + 1 blocks (25): q = 3 each, the last one shared by rows over waves 0..4
      const int q = nblk >> 3;
      if (wave < kNbRb)
        nb_wide_layer_x<NCB - 1, 1, OSRL_NB8_ADB, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], wave * q,
                                                       8 * q, wave, lane, l, pre);
      else
        nb_wide_layer<NCB - 1, OSRL_NB8_ADB, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], wave * q, lane, l,
                                                  pre);
    } else if constexpr (SHARED) {  // 4q + 1 blocks (400-wide: 25): q each, the last one shared by rows
      const int q = nblk >> 2;
      if (wave == 0)
        nb_wide_layer_x<NCB - 1, 2, true, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], 0, 4 * q, 0, lane, l,
                                               pre);
      else
        nb_wide_layer_x<NCB - 1, 1, true, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], wave * q, 4 * q,
                                               wave + 1, lane, l, pre);
    } else {
      int cb0, cnt;
      wave_blocks<NW>(nblk, wave, &cb0, &cnt);
      const bool headl = a.fuse_head && l == L - 2;  // (uniform) the one-output head rides on this layer's accumulators
      const float* __restrict__ hw = headl ? a.net.Wf[e][L - 1] : nullptr;
      float* part = headl ? s_head + wave * BM : nullptr;
      if (cnt == NCB)
        nb_wide_layer<NCB, true, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], cb0, lane, l, pre, hw, part);
      else
        nb_wide_layer<NCB - 1, true, PRE>(lds, lda, K, N, a.net.Wf[e][l], a.net.b[e][l], a.net.acts[l], cb0, lane, l, pre, hw,
                                          part);
    }
  };
  if constexpr (PREFIX) wide(0, std::true_type{});
  for (int l = PREFIX ? 1 : 0; l + 1 < L; ++l) wide(l, std::false_type{});
  if (!SHARED && a.fuse_head) {  // ---- one-output head: the waves' partial dot products, summed in wave order
    __syncthreads();
    const int l = L - 1;
    if (tid < BM && grow(tid) < rows) {
      float sacc = s_head[tid];
#pragma unroll
      for (int w = 1; w < NW; ++w) sacc += s_head[w * BM + tid];
      a.y[e][grow(tid)] = act_fwd(a.net.acts[l], sacc + a.net.b[e][l][0]) * a.net.out_scale;
    }
    PHASE_STAMP(5 + 4 * l);
    WG_LOG(1);
    return;
  }
  {  // ---- narrow head: K split over the 4 waves, every weight fragment of a wave's share loaded up front
    const int l = L - 1;
    const int K = a.net.dims[l], N = a.net.dims[l + 1];
    const int Np = round16(N), nblk = Np >> 4, nk = round16(K) >> 4;
    const int k_lo = (nk * wave) / NW, k_hi = (nk * (wave + 1)) / NW;
    constexpr int kMaxSteps = 32 / NW;  // nk <= 32 (widths <= 448 -> nk <= 28 -> <= 7 | 4 steps per wave)
    const float* __restrict__ P = a.net.Wf[e][l];
    const int m = lane & 15, kq = lane >> 4;
    f32x4 t[kNbRb][2];
#pragma unroll
    for (int rb = 0; rb < kNbRb; ++rb) t[rb][0] = t[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bw[kMaxSteps][2];
#pragma unroll
    for (int sI = 0; sI < kMaxSteps; ++sI) {
      const int ks = k_lo + sI < k_hi ? k_lo + sI : k_hi - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c)
        bw[sI][c] = (c < nblk) ? *reinterpret_cast<const f32x4*>(P + ((size_t)(ks * 4 + kq) * Np + c * 16 + m) * 4)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* arow = lds + m * lda + 4 * kq;
#pragma unroll
    for (int sI = 0; sI < kMaxSteps; ++sI) {
      if (k_lo + sI < k_hi) {
        const int ks = k_lo + sI;
        f32x4 af[kNbRb];
#pragma unroll
        for (int rb = 0; rb < kNbRb; ++rb) af[rb] = *reinterpret_cast<const f32x4*>(arow + rb * 16 * lda + ks * 16);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int rb = 0; rb < kNbRb; ++rb) {
            t[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rb][tt], bw[sI][0][tt], t[rb][0], 0, 0, 0);
            if (nblk > 1) t[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[rb][tt], bw[sI][1][tt], t[rb][1], 0, 0, 0);
          }
      }
    }
    PHASE_STAMP(2 + 4 * l);
    __syncthreads();  // all reads of the activations are done: the tile's first 4 * Np columns take the partials
    PHASE_STAMP(3 + 4 * l);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (c < nblk) {
#pragma unroll
        for (int rb = 0; rb < kNbRb; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            lds[(rb * 16 + kq * 4 + r) * lda + wave * Np + c * 16 + m] = t[rb][c][r];
      }
    __syncthreads();
    PHASE_STAMP(4 + 4 * l);
    const float* __restrict__ bias = a.net.b[e][l];
    const int act = a.net.acts[l];
    const float oscale = a.net.out_scale;
    float* __restrict__ y = a.y[e];
    for (int idx = tid; idx < BM * N; idx += 64 * NW) {
      const int r = idx / N, c = idx - r * N;
      if (grow(r) < rows) {
        const float* p = lds + r * lda + c;
        float sacc = ((p[0] + p[Np]) + p[2 * Np]) + p[3 * Np];
        if constexpr (NW == 8) sacc = (((sacc + p[4 * Np]) + p[5 * Np]) + p[6 * Np]) + p[7 * Np];
        y[(size_t)grow(r) * N + c] = act_fwd(act, sacc + bias[c]) * oscale;
      }
    }
    if (a.kl && e == 0) {
      // the KL rows of cpq.py:178-182 (glue.hip vae_kl_rows_kernel, term for term) from the partial tiles still in LDS:
      // one launch and one read of the head off the side chain's tail
      const int Lz = a.kl_L;
      for (int r = tid; r < BM; r += 64 * NW) {
        if (grow(r) < rows) {
          auto outv = [&](int c) {
            const float* p = lds + r * lda + c;
            float sacc = ((p[0] + p[Np]) + p[2 * Np]) + p[3 * Np];
            if constexpr (NW == 8) sacc = (((sacc + p[4 * Np]) + p[5 * Np]) + p[6 * Np]) + p[7 * Np];
            return act_fwd(act, sacc + bias[c]) * oscale;
          };
          float s_ = 0.f;
          for (int k = 0; k < Lz; ++k) {
            const float mean = outv(k);
            const float sd = expf(fminf(fmaxf(outv(Lz + k), kNbLsMin), kNbLsMax));
            s_ += -0.5f * (1.0f + logf(sd * sd) - mean * mean - sd * sd);
          }
          a.kl[grow(r)] = s_ / (float)Lz;
        }
      }
    }
    PHASE_STAMP(5 + 4 * l);
  }
  WG_LOG(1);
}
// (NCB == 4: the 13..16-block form is held to 256 registers per lane although its LDS tile allows one workgroup per CU
// anyway -- with the one-output head fused in, the allocator took 205 VGPRs + 80 AGPRs where 512 are free, which left the
// chain kernels of the other graph branch two 96-register waves per SIMD beside it instead of three: C2 2215-2221 vs
// 2309-2322 steps/s with the cap (205 registers, accumulators in VGPRs), profiles/r6_nb_registers_ab.txt)
// (PREFIX: tiles of shared src0 rows, osrl_rows_t.share0 / nb_share_acc above -- instantiations of their own, so that the plain
// forms keep their register counts)
template <int NCB, bool SHARED = false, bool LIST = false, bool PREFIX = false>
__global__ __launch_bounds__(256, (kNbRb == 4 || NCB == 4) ? 2 : 1) void mlp_fwd_nb_kernel(const NbArgs a) {
  mlp_fwd_nb_body<NCB, SHARED, 4, const NbArgs&, LIST, PREFIX>(a);
}
template <int NCB, bool SHARED = false, bool LIST = false, bool PREFIX = false>
__global__ __launch_bounds__(256, (kNbRb == 4 || NCB == 4) ? 2 : 1) void mlp_fwd_nb_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(8, p);
  mlp_fwd_nb_body<NCB, SHARED, 4, const OSRL_CAS NbArgs&, LIST, PREFIX>(*(const OSRL_CAS NbArgs*)p);
}
// the 8-wave form (25-block layers): NCB - 1 = 3 column blocks per wave
#ifndef OSRL_NB8_WPE
#define OSRL_NB8_WPE 2
#endif
#define OSRL_NB8_ATTR
__global__ __launch_bounds__(512, OSRL_NB8_WPE) OSRL_NB8_ATTR void mlp_fwd_nb8_kernel(const NbArgs a) {
  mlp_fwd_nb_body<4, true, 8, const NbArgs&>(a);
}
__global__ __launch_bounds__(512, OSRL_NB8_WPE) OSRL_NB8_ATTR void mlp_fwd_nb8_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(9, p);
  mlp_fwd_nb_body<4, true, 8, const OSRL_CAS NbArgs&>(*(const OSRL_CAS NbArgs*)p);
}
__global__ __launch_bounds__(512, OSRL_NB8_WPE) OSRL_NB8_ATTR void mlp_fwd_nb8_pre_kernel(const NbArgs a) {
  mlp_fwd_nb_body<4, true, 8, const NbArgs&, false, true>(a);
}
__global__ __launch_bounds__(512, OSRL_NB8_WPE) OSRL_NB8_ATTR void mlp_fwd_nb8_pre_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(9, p);
  mlp_fwd_nb_body<4, true, 8, const OSRL_CAS NbArgs&, false, true>(*(const OSRL_CAS NbArgs*)p);
}

// the 8-wave form of the 13..16-block (<= 256-wide) nets: 2 column blocks per wave, two waves per SIMD (the 4-wave form
// runs ONE wave per SIMD -- its 84.5 KB activation tile allows one workgroup per CU -- so nothing covers a wave's
// barrier / weight-latency stalls)
__global__ __launch_bounds__(512, 2) void mlp_fwd_nb8n_kernel(const NbArgs a) {
  mlp_fwd_nb_body<2, false, 8, const NbArgs&>(a);
}
__global__ __launch_bounds__(512, 2) void mlp_fwd_nb8n_kernel_p(const void* p) {
  mlp_fwd_nb_body<2, false, 8, const OSRL_CAS NbArgs&>(*(const OSRL_CAS NbArgs*)p);
}

// ---- host side of mlp_fwd_nb_kernel: eligibility + launch (tile_rows = 80) ------------------------------------
template <int NCB, bool SHARED = false, bool LIST = false, bool PREFIX = false>
static int launch_nb(const NbArgs& a, int tiles, int nets, size_t lds_bytes, hipStream_t stream) {
  const void* dev_args = osrl_argmem::slot(a);
  hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(mlp_fwd_nb_kernel_p<NCB, SHARED, LIST, PREFIX>)
                                              : reinterpret_cast<const void*>(mlp_fwd_nb_kernel<NCB, SHARED, LIST, PREFIX>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return (int)e;
  (void)hipGetLastError();
  if (dev_args)
    hipLaunchKernelGGL((mlp_fwd_nb_kernel_p<NCB, SHARED, LIST, PREFIX>), dim3(tiles, nets, 1), dim3(256), lds_bytes, stream, dev_args);
  else
    hipLaunchKernelGGL((mlp_fwd_nb_kernel<NCB, SHARED, LIST, PREFIX>), dim3(tiles, nets, 1), dim3(256), lds_bytes, stream, a);
  return (int)hipGetLastError();
}

}  // namespace

__attribute__((visibility("hidden"))) int OSRL_NB_LAUNCH(const osrl_mlp_t* net, const osrl_rows_t* in, const osrl_mlp_acts_t* out, hipStream_t stream,
                                                             float* kl, int kl_L) {
  const int L = net->n_layers, nets = net->n_nets;
  if (net->tile_rows != 80 || L < 2 || out->x || net->dims[0] > 128) return kNbNotTaken;  // (80 = "the big-row inference form")
  // (a launch with osrl_rows_t.row_list that is not taken here is refused by the caller: mlp.hip mlp_forward_impl)
  for (int e = 0; e < nets; ++e)
    for (int l = 0; l + 1 < L; ++l)
      if (out->h[e][l]) return kNbNotTaken;  // training launches keep hidden activations: mlp_fwd_kernel
  int ncb = 0, wmax = net->dims[0];
  for (int l = 0; l + 1 < L; ++l) {  // wide layers: every wave owns NCB or NCB - 1 column blocks
    const int N = net->dims[l + 1], nblk = (N + 15) / 16;
    const int need = nblk >= 13 && nblk <= 16 ? 4 : nblk >= 25 && nblk <= 28 ? 7 : 0;
    if (!need || (ncb && need != ncb)) return kNbNotTaken;
    ncb = need;
    wmax = N > wmax ? N : wmax;
  }
  const int NL = net->dims[L], nkl = (((net->dims[L - 1] + 15) & ~15) >> 4);
  if (NL > 32 || nkl < 4 || nkl > 32) return kNbNotTaken;  // narrow head, K split over 4 waves (<= 8 steps each)
  const int lda = ((wmax + 15) & ~15) + 8;
  if (lda < 4 * ((NL + 15) & ~15)) return kNbNotTaken;  // the head's 4 partial tiles live in the activation tile
  size_t lds_bytes = (size_t)kNbRows * lda * sizeof(float);
  if (kNbRb == 5 && lds_bytes <= 80 * 1024) lds_bytes = 80 * 1024 + 256;  // more than half of the 160 KB: one workgroup per CU
  if (lds_bytes > kLdsMax) return kNbNotTaken;
  NbArgs a{};
  a.net = *net;
  a.in = *in;
  bool share = false;
  if (in->share0) {  // tiles of [kNbRb copies] x [16 src0 rows] (nb_share_acc): rows = n * div0 + b, every tile whole, phase A =
    // the whole 16-column k-steps inside src0's columns with at least one k-step left for phase B.  A HINT: the launch
    // computes the same function either way -- where the shape or the kernel form does not allow it, the plain tiles run
    const int nk0 = (((net->dims[0] + 15) & ~15) >> 4), B0 = in->div0;
    share = !in->row_list && in->map0 == OSRL_MAP_MOD && B0 >= 16 && !(B0 & 15) && in->rows % B0 == 0 &&
            (in->rows / B0) % kNbRb == 0 && in->share_k16 >= 1 && in->share_k16 < nk0 && 16 * in->share_k16 <= in->d0;
  }
  a.in.share0 = 0;  // (set again where a shared-row form takes the launch: the plain kernels' descriptors stay as they were)
  for (int e = 0; e < OSRL_MAX_NETS; ++e) a.y[e] = e < nets ? out->h[e][L - 1] : nullptr;
  a.lda = lda;
  a.kl = kl;
  a.kl_L = kl_L;
  {  // one-output heads (Q networks) on the 13..16-block forms: fused into the last wide layer's epilogue.  OSRL_NB_HEAD=0
    // restores the head as a layer (read per launch: A/B runs and the kernel-equality test flip it)
    const char* fh = getenv("OSRL_NB_HEAD");
    a.fuse_head = (NL == 1 && ncb == 4 && !kl && !(fh && atoi(fh) == 0)) ? 1 : 0;
  }
  const int tiles = (in->rows + kNbRows - 1) / kNbRows;
  if (in->row_list || in->n_rows_dev) {  // a device-chosen row set: the 4-wave 80-row form of the <= 256-wide nets only
#if OSRL_NB_RB == 5
    if (!in->row_list || !in->n_rows_dev) return -1;
    if (ncb != 4 || kl) return -3;
    return launch_nb<4, false, true>(a, tiles, nets, lds_bytes, stream);
#else
    return -3;
#endif
  }
  bool shared = ncb == 7;  // every wide layer 4*6 + 1 = 25 column blocks (400-wide): the balanced instantiation
  for (int l = 0; l + 1 < L; ++l) shared = shared && ((net->dims[l + 1] + 15) >> 4) == 25;
  // 8 waves for the 25-block layers (OSRL_NB_WAVES=4: the one-wave-per-SIMD form, for A/B runs); the head's 8 partial
  // tiles need 8 * round16(NL) columns of the activation tile
  const char* nb_env = getenv("OSRL_NB_WAVES");  // (read per launch: a test flips it between calls)
  const bool nb8 = !(nb_env && atoi(nb_env) == 4);
  if (shared && nb8 && lda >= 8 * ((NL + 15) & ~15)) {
    a.in.share0 = share ? 1 : 0;
    const void* dev_args = osrl_argmem::slot(a);
    const void* fn = share ? (dev_args ? reinterpret_cast<const void*>(mlp_fwd_nb8_pre_kernel_p)
                                         : reinterpret_cast<const void*>(mlp_fwd_nb8_pre_kernel))
                             : (dev_args ? reinterpret_cast<const void*>(mlp_fwd_nb8_kernel_p)
                                         : reinterpret_cast<const void*>(mlp_fwd_nb8_kernel));
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    (void)hipGetLastError();
    if (share) {
      if (dev_args)
        hipLaunchKernelGGL(mlp_fwd_nb8_pre_kernel_p, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, dev_args);
      else
        hipLaunchKernelGGL(mlp_fwd_nb8_pre_kernel, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, a);
    } else if (dev_args)
      hipLaunchKernelGGL(mlp_fwd_nb8_kernel_p, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, dev_args);
    else
      hipLaunchKernelGGL(mlp_fwd_nb8_kernel, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, a);
    return (int)hipGetLastError();
  }
  if (ncb == 4 && lda >= 8 * ((NL + 15) & ~15)) {
    // 13..16-block (<= 256-wide) nets.  The 4-wave 80-row form runs ONE wave per SIMD (its 84.5 KB activation tile allows one
    // workgroup per CU).  From 512 80-row tiles per net on (BCQ-Lag / BEAR-Lag at B = 4096, N = 10) the 64-row form of
    // mlp_nb64.hip takes the launch: two 4-wave workgroups per CU with independent barriers -- C3 604 -> 639 steps/s (the
    // 8-wave 80-row form below: 622).  At CPQ's 256 tiles per net, launched beside the 2048-row chain kernels, the 80-row
    // 4-wave form is the faster neighbour (C2 2237 vs 2210 with 64-row tiles, 2211 with 8 waves).
    // OSRL_NB64 = 0 / 1 and OSRL_NB256_WAVES = 4 / 8 force a form (read per launch: A/B runs and a test flip them)
    const char* w256 = getenv("OSRL_NB256_WAVES");
    const bool eight = OSRL_NB_RB == 5 && w256 && atoi(w256) == 8;
#if OSRL_NB_RB == 5
    {
      const char* e64 = getenv("OSRL_NB64");
      const bool use64 = e64 && e64[0] ? atoi(e64) == 1 : (tiles >= 512 && !eight && !(w256 && atoi(w256) == 4));
      if (use64) return osrl_launch_fwd_nb64(net, in, out, stream, kl, kl_L);
    }
#endif
    if (eight) {
      const void* dev_args = osrl_argmem::slot(a);
      hipError_t e = hipFuncSetAttribute(dev_args ? reinterpret_cast<const void*>(mlp_fwd_nb8n_kernel_p)
                                                  : reinterpret_cast<const void*>(mlp_fwd_nb8n_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return (int)e;
      (void)hipGetLastError();
      if (dev_args)
        hipLaunchKernelGGL(mlp_fwd_nb8n_kernel_p, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, dev_args);
      else
        hipLaunchKernelGGL(mlp_fwd_nb8n_kernel, dim3(tiles, nets, 1), dim3(512), lds_bytes, stream, a);
      return (int)hipGetLastError();
    }
  }
  if (share && ncb == 4 && !shared) {  // (the shared-row forms: 4-wave 13..16-block nets here, 8-wave 25-block nets above)
    a.in.share0 = 1;
    return launch_nb<4, false, false, true>(a, tiles, nets, lds_bytes, stream);
  }
  if (shared) return launch_nb<7, true>(a, tiles, nets, lds_bytes, stream);
  return ncb == 4 ? launch_nb<4>(a, tiles, nets, lds_bytes, stream) : launch_nb<7>(a, tiles, nets, lds_bytes, stream);
}
