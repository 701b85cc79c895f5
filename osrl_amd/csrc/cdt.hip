// cdt.hip -- the non-GEMM kernels of the Constrained Decision Transformer train step on gfx950.
//
// The transformer's projections run on the packed-weight MFMA kernels of mlp.hip (osrl_linear,
// osrl_mlp_backward_dw); this file holds everything around them:
//   * token embedding + timestep embedding + (r,c,s,a) interleave + emb LayerNorm   cdt.py:178-222
//   * LayerNorm forward / backward (+ fused residual add)                            net.py:402-403,427,440
//   * causal + key-padding softmax attention, forward and backward (80x80 score tiles at the
//     BASELINE config: 4*seq_len tokens, head_dim 32; scores never leave LDS)       net.py:406-409,428-435
//   * exact-erf GELU forward / backward                                              net.py:412
//   * heads' loss: Gaussian NLL + entropy over valid tokens, 2-class cost NLL, shifted state MSE,
//     accuracy, and the gradients that seed the backward pass                        cdt.py:357-394
//   * timestep-embedding gradient scatter, global grad-norm clip factor              cdt.py:398-399
//   * temperature (log_temperature) Adam step                                        cdt.py:402-407
// All fp32.  Row-wise kernels use one wave64 per token row (E <= 512) with shuffle reductions; these
// are HBM-streaming kernels (LayerNorm: 2 reads + 1-2 writes of [tokens, E] per call).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "philox.h"
#include "gelu.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kMaxEPL = 8;  // features per lane: E <= 512
constexpr float kLnEps = 1e-5f;

// 64-lane sum, result in every lane: four DPP steps inside each 16-lane row (VALU, no LDS round trip), then two
// cross-row exchanges -- instead of six ds_bpermute round trips (the LayerNorm kernels reduce twice per token row)
__device__ __forceinline__ float wsum(float v) {
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));  // row_half_mirror
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));  // row_mirror
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float block_sum1024(float v, float* sm) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sm[i];
    sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}

// LayerNorm of one row held as v[j] (feature lane + 64 j); returns mean / rstd, writes y
__device__ __forceinline__ void ln_row(const float (&v)[kMaxEPL], int E, int lane, const float* __restrict__ g,
                                       const float* __restrict__ b, float* __restrict__ y, float* mean_o,
                                       float* rstd_o) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j)
    if (lane + 64 * j < E) s += v[j];
  const float mean = wsum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j)
    if (lane + 64 * j < E) q += (v[j] - mean) * (v[j] - mean);
  const float rstd = 1.0f / sqrtf(wsum(q) / (float)E + kLnEps);
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j) {
    const int f = lane + 64 * j;
    if (f < E) y[f] = (v[j] - mean) * rstd * g[f] + b[f];
  }
  *mean_o = mean;
  *rstd_o = rstd;
}

struct EmbedArgs {
  const float *states, *actions, *returns, *ctg, *episode_cost;
  const int64_t* time_steps;
  const float *Ws, *bs, *Wa, *ba, *Wc, *bc, *Wr, *br, *Wp, *bp, *te, *g, *b;
  float *seq, *x0, *stats, *ctg_t;
  int32_t B, T, od, ad, E, cost_transform;
  int32_t R, prefix, use_rew, use_cost;  // tokens per timestep (2..4), prefix token in front, which tokens exist
};

// one wave per token.  Token order per timestep: [return] [cost] state action (cdt.py:185-200: costs are inserted in
// front of the state token, returns in front of those); with a cost prefix one more token leads each sequence
// (cdt.py:207-218: Linear(1, E) of the episode's cost budget, no timestep embedding).  te == NULL: time_emb = False.
__global__ __launch_bounds__(256) void embed_ln_kernel(const EmbedArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int S_ = a.R * a.T + a.prefix;
  if (row >= a.B * S_) return;
  const int b = row / S_, pos = row - b * S_;
  const bool is_prefix = a.prefix && pos == 0;
  const int tp = pos - a.prefix;
  const int t = is_prefix ? 0 : tp / a.R, slot = is_prefix ? 0 : tp - t * a.R;
  const int bt = b * a.T + t;
  // slot -> token kind (0 return, 1 cost, 2 state, 3 action)
  int which = slot + (4 - a.R);
  if (a.R == 3 && a.use_rew) which = slot == 0 ? 0 : slot + 1;
  const float* __restrict__ te = (a.te && !is_prefix) ? a.te + (size_t)a.time_steps[bt] * a.E : nullptr;
  float v[kMaxEPL];
  const float ret = a.use_rew ? a.returns[bt] : 0.f;
  const float ctg = a.use_cost ? (a.cost_transform ? 50.0f - a.ctg[bt] : a.ctg[bt]) : 0.f;  // cdt.py:78-81,187-188
  if (!is_prefix && which == 1 && lane == 0) a.ctg_t[bt] = ctg;
  const float ec = is_prefix ? a.episode_cost[b] : 0.f;
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j) {
    const int f = lane + 64 * j;
    float x = 0.f;
    if (f < a.E) {
      if (is_prefix) {
        x = ec * a.Wp[f] + a.bp[f];
      } else if (which == 0) {
        x = ret * a.Wr[f] + a.br[f];
      } else if (which == 1) {
        x = ctg * a.Wc[f] + a.bc[f];
      } else if (which == 2) {
        x = a.bs[f];
        for (int i = 0; i < a.od; ++i) x += a.states[(size_t)bt * a.od + i] * a.Ws[(size_t)f * a.od + i];
      } else {
        x = a.ba[f];
        for (int i = 0; i < a.ad; ++i) x += a.actions[(size_t)bt * a.ad + i] * a.Wa[(size_t)f * a.ad + i];
      }
      if (te) x += te[f];
      a.seq[(size_t)row * a.E + f] = x;
    }
    v[j] = x;
  }
  float mean, rstd;
  ln_row(v, a.E, lane, a.g, a.b, a.x0 + (size_t)row * a.E, &mean, &rstd);
  if (lane == 0) {
    a.stats[2 * row] = mean;
    a.stats[2 * row + 1] = rstd;
  }
}

// The same for E = 256 / 512 (round 4): lanes own four consecutive features per 256-feature chunk (16-byte accesses of
// seq / x0 / the timestep row), and the state / action embedding weights -- which the kernel above reads per element with
// a lane stride of od floats, ~90 scattered dword loads per token: 109 us for the 81920 tokens of C5, bound by the
// texture-address path -- are staged ONCE per workgroup, transposed ([input][feature]), into LDS; a workgroup walks many
// tokens.  The products are added in the same order (bias first, inputs ascending): the pre-LayerNorm sequence is
// bit-equal, the LayerNorm sums meet in a different order (as in ln_fwd_v4_kernel).
template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_v4_kernel(const EmbedArgs a, const int n_wg) {
  constexpr int E = 256 * NCH;
  extern __shared__ __attribute__((aligned(16))) float wt[];  // [od + ad][E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < (a.od + a.ad) * E; idx += 256) {
    const int i = idx / E, f = idx - i * E;
    wt[idx] = i < a.od ? a.Ws[(size_t)f * a.od + i] : a.Wa[(size_t)f * a.ad + (i - a.od)];
  }
  __syncthreads();
  const int S_ = a.R * a.T + a.prefix, rows = a.B * S_;
  f32x4 gg[NCH], bb[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    gg[c] = *reinterpret_cast<const f32x4*>(a.g + 4 * lane + 256 * c);
    bb[c] = *reinterpret_cast<const f32x4*>(a.b + 4 * lane + 256 * c);
  }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += n_wg * 4) {
    const int b = row / S_, pos = row - b * S_;
    const bool is_prefix = a.prefix && pos == 0;
    const int tp = pos - a.prefix;
    const int t = is_prefix ? 0 : tp / a.R, slot = is_prefix ? 0 : tp - t * a.R;
    const int bt = b * a.T + t;
    int which = slot + (4 - a.R);  // slot -> token kind (0 return, 1 cost, 2 state, 3 action)
    if (a.R == 3 && a.use_rew) which = slot == 0 ? 0 : slot + 1;
    const float* __restrict__ te = (a.te && !is_prefix) ? a.te + (size_t)a.time_steps[bt] * E : nullptr;
    const float ret = a.use_rew ? a.returns[bt] : 0.f;
    const float ctg = a.use_cost ? (a.cost_transform ? 50.0f - a.ctg[bt] : a.ctg[bt]) : 0.f;
    if (!is_prefix && which == 1 && lane == 0) a.ctg_t[bt] = ctg;
    const float ec = is_prefix ? a.episode_cost[b] : 0.f;
    const size_t base = (size_t)row * E + 4 * lane;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int f0 = 4 * lane + 256 * c;
      f32x4 x;
      if (is_prefix || which < 2) {  // rank-1 embeddings: scalar * W[:, 0] + bias
        const float sc = is_prefix ? ec : (which == 0 ? ret : ctg);
        const float* W = is_prefix ? a.Wp : (which == 0 ? a.Wr : a.Wc);
        const float* bv = is_prefix ? a.bp : (which == 0 ? a.br : a.bc);
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(W + f0), b4 = *reinterpret_cast<const f32x4*>(bv + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = sc * w4[j] + b4[j];
      } else {
        const bool st = which == 2;
        const int n_in = st ? a.od : a.ad;
        const float* __restrict__ in = st ? a.states + (size_t)bt * a.od : a.actions + (size_t)bt * a.ad;
        const float* __restrict__ w = wt + (size_t)(st ? 0 : a.od) * E + f0;
        x = *reinterpret_cast<const f32x4*>((st ? a.bs : a.ba) + f0);
        for (int i = 0; i < n_in; ++i) {
          const float xi = in[i];
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + (size_t)i * E);
#pragma unroll
          for (int j = 0; j < 4; ++j) x[j] += xi * w4[j];
        }
      }
      if (te) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(te + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] += t4[j];
      }
      *reinterpret_cast<f32x4*>(a.seq + base + 256 * c) = x;
      v[c] = x;
      s += (x[0] + x[1]) + (x[2] + x[3]);
    }
    const float mean = wsum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) q += (v[c][j] - mean) * (v[c][j] - mean);
    const float rstd = 1.0f / sqrtf(wsum(q) / (float)E + kLnEps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * gg[c][j] + bb[c][j];
      *reinterpret_cast<f32x4*>(a.x0 + base + 256 * c) = o;
    }
    if (lane == 0) {
      a.stats[2 * row] = mean;
      a.stats[2 * row + 1] = rstd;
    }
  }
}

// the dropout keep-multiplier of flat element i at a site (the mask of dropout_kernel: one Philox call per 4 consecutive
// elements, word i & 3) -- for kernels whose lanes do not own 4 consecutive elements
struct DropSite {
  uint32_t thresh, site, k0, k1;
  float scale;
  const osrl_step_state_t* st;
};
__device__ __forceinline__ float drop_keep(const DropSite& d, uint32_t step, uint64_t i) {
  const osrl_rng::U4 w = osrl_rng::drop_words(i >> 2, step, d.site, d.k0, d.k1);
  const uint32_t r = (i & 3) == 0 ? w.x : (i & 3) == 1 ? w.y : (i & 3) == 2 ? w.z : w.w;
  return r >= d.thresh ? d.scale : 0.f;
}

// y = LN(x + delta); xout = x + delta (optional).  DROP (round 3): delta is the residual branch BEFORE its nn.Dropout
// (net.py:414,439): the keep-multiplier is applied on the way in -- x + (delta * m), product rounded before the add,
// i.e. the bits of osrl_dropout followed by the plain kernel -- and one read + one write pass over [M, E] and a launch
// per residual branch disappear (0.36 ms of the 15 ms C5 step)
template <bool DROP>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ delta,
                                                     const float* __restrict__ g, const float* __restrict__ b,
                                                     float* __restrict__ xout, float* __restrict__ y,
                                                     float* __restrict__ stats, int M, int E, const DropSite ds) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  uint32_t step = 0;
  if constexpr (DROP) step = ds.st ? (uint32_t)ds.st->step : 0u;
  float v[kMaxEPL];
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j) {
    const int f = lane + 64 * j;
    float t = 0.f;
    if (f < E) {
      t = x[(size_t)row * E + f];
      if (delta) {
        float d = delta[(size_t)row * E + f];
        if constexpr (DROP) d = __fmul_rn(d, drop_keep(ds, step, (uint64_t)row * E + f));
        t = __fadd_rn(t, d);
      }
      if (xout) xout[(size_t)row * E + f] = t;
    }
    v[j] = t;
  }
  float mean, rstd;
  ln_row(v, E, lane, g, b, y + (size_t)row * E, &mean, &rstd);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)) (+ dres); per-workgroup partial dgamma/dbeta
// DROP (round 3): also dxd = dx * keep-multiplier of a dropout site -- the gradient entering the residual branch
// behind the dropout that follows this LayerNorm's input (what a separate osrl_dropout pass over dx produced)
template <bool DROP>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ g,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ partial, int M, int E,
                                                     float* __restrict__ dxd, const DropSite ds) {
  __shared__ float sm[4][2 * 64 * kMaxEPL];
  uint32_t step = 0;
  if constexpr (DROP) step = ds.st ? (uint32_t)ds.st->step : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[kMaxEPL], db[kMaxEPL];
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j) dg[j] = db[j] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[kMaxEPL], dxh[kMaxEPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxEPL; ++j) {
      const int f = lane + 64 * j;
      xh[j] = dxh[j] = 0.f;
      if (f < E) {
        const float d = dy[(size_t)row * E + f];
        xh[j] = (x[(size_t)row * E + f] - mean) * rstd;
        dxh[j] = d * g[f];
        dg[j] += d * xh[j];
        db[j] += d;
        s1 += dxh[j];
        s2 += dxh[j] * xh[j];
      }
    }
    s1 = wsum(s1) / (float)E;
    s2 = wsum(s2) / (float)E;
#pragma unroll
    for (int j = 0; j < kMaxEPL; ++j) {
      const int f = lane + 64 * j;
      if (f < E) {
        float v = rstd * (dxh[j] - s1 - xh[j] * s2);
        if (dres) v += dres[(size_t)row * E + f];
        dx[(size_t)row * E + f] = v;
        if constexpr (DROP) dxd[(size_t)row * E + f] = __fmul_rn(v, drop_keep(ds, step, (uint64_t)row * E + f));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kMaxEPL; ++j) {
    sm[wave][lane + 64 * j] = dg[j];
    sm[wave][64 * kMaxEPL + lane + 64 * j] = db[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * E; i += 256) {
    const int f = i < E ? i : 64 * kMaxEPL + (i - E);
    partial[(size_t)blockIdx.x * 2 * E + i] = (sm[0][f] + sm[1][f]) + (sm[2][f] + sm[3][f]);
  }
}

// ---- the same two kernels for E % 256 == 0 (round 4): a lane owns FOUR CONSECUTIVE features per 256-feature chunk --
// every tensor moves as one 16-byte access per lane and chunk (the kernels above issue four dword accesses 256 B apart),
// and the dropout site costs ONE Philox call per lane and chunk: the four words of a call are the masks of four
// consecutive elements, so the kernels above evaluated the generator four times per element group and kept one word of
// each (~100 VALU instructions per call; profiles/r4_cdt_kernel_stats.csv: ln_bwd<true> 131 us = 3.2 TB/s for its
// 420 MB).  Same arithmetic per element and the same masks; the row reductions add the lanes' partial sums in a different
// grouping (last-bit differences in mean / rstd against the kernels above: both are within the parity tolerance).
// The backward loads the next row of the wave while it works on the current one.
template <bool DROP>
__device__ __forceinline__ f32x4 drop_keep4(const DropSite& d, uint32_t step, uint64_t i4) {  // elements 4 i4 .. 4 i4 + 3
  if constexpr (!DROP) return f32x4{1.f, 1.f, 1.f, 1.f};
  const osrl_rng::U4 w = osrl_rng::drop_words(i4, step, d.site, d.k0, d.k1);
  return f32x4{w.x >= d.thresh ? d.scale : 0.f, w.y >= d.thresh ? d.scale : 0.f, w.z >= d.thresh ? d.scale : 0.f,
               w.w >= d.thresh ? d.scale : 0.f};
}

template <bool DROP, int NCH>
__global__ __launch_bounds__(256) void ln_fwd_v4_kernel(const float* __restrict__ x, const float* __restrict__ delta,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        float* __restrict__ xout, float* __restrict__ y,
                                                        float* __restrict__ stats, int M, const DropSite ds) {
  constexpr int E = 256 * NCH;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  uint32_t step = 0;
  if constexpr (DROP) step = ds.st ? (uint32_t)ds.st->step : 0u;
  const size_t base = (size_t)row * E + 4 * lane;
  f32x4 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    f32x4 t = *reinterpret_cast<const f32x4*>(x + base + 256 * c);
    if (delta) {
      f32x4 d = *reinterpret_cast<const f32x4*>(delta + base + 256 * c);
      if constexpr (DROP) {
        const f32x4 k = drop_keep4<DROP>(ds, step, (base + 256 * c) >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = __fmul_rn(d[j], k[j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = __fadd_rn(t[j], d[j]);
    }
    if (xout) *reinterpret_cast<f32x4*>(xout + base + 256 * c) = t;
    v[c] = t;
    s += (t[0] + t[1]) + (t[2] + t[3]);
  }
  const float mean = wsum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) q += (v[c][j] - mean) * (v[c][j] - mean);
  const float rstd = 1.0f / sqrtf(wsum(q) / (float)E + kLnEps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + 4 * lane + 256 * c);
    const f32x4 bb = *reinterpret_cast<const f32x4*>(b + 4 * lane + 256 * c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (v[c][j] - mean) * rstd * gg[j] + bb[j];
    *reinterpret_cast<f32x4*>(y + base + 256 * c) = o;
  }
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

template <bool DROP, int NCH>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const float* __restrict__ g,
                                                        const float* __restrict__ dres, float* __restrict__ dx,
                                                        float* __restrict__ partial, int M, float* __restrict__ dxd,
                                                        const DropSite ds, const int n_wg) {
  constexpr int E = 256 * NCH;
  __shared__ float sm[4][2 * E];
  uint32_t step = 0;
  if constexpr (DROP) step = ds.st ? (uint32_t)ds.st->step : 0u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 dg[NCH], db[NCH], gg[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    dg[c] = db[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    gg[c] = *reinterpret_cast<const f32x4*>(g + 4 * lane + 256 * c);
  }
  struct Row {
    f32x4 d[NCH], xv[NCH], r[NCH];
    float mean, rstd;
  };
  auto load = [&](int row, Row& R) __attribute__((always_inline)) {
    const size_t base = (size_t)row * E + 4 * lane;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      R.d[c] = *reinterpret_cast<const f32x4*>(dy + base + 256 * c);
      R.xv[c] = *reinterpret_cast<const f32x4*>(x + base + 256 * c);
      if (dres) R.r[c] = *reinterpret_cast<const f32x4*>(dres + base + 256 * c);
    }
    R.mean = stats[2 * row];
    R.rstd = stats[2 * row + 1];
  };
  auto work = [&](int row, const Row& R) __attribute__((always_inline)) {
    const size_t base = (size_t)row * E + 4 * lane;
    f32x4 xh[NCH], dxh[NCH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = R.d[c][j];
        xh[c][j] = (R.xv[c][j] - R.mean) * R.rstd;
        dxh[c][j] = d * gg[c][j];
        dg[c][j] += d * xh[c][j];
        db[c][j] += d;
        s1 += dxh[c][j];
        s2 += dxh[c][j] * xh[c][j];
      }
    s1 = wsum(s1) / (float)E;
    s2 = wsum(s2) / (float)E;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = R.rstd * (dxh[c][j] - s1 - xh[c][j] * s2);
        if (dres) v[j] += R.r[c][j];
      }
      *reinterpret_cast<f32x4*>(dx + base + 256 * c) = v;
      if constexpr (DROP) {
        const f32x4 k = drop_keep4<DROP>(ds, step, (base + 256 * c) >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __fmul_rn(v[j], k[j]);
        *reinterpret_cast<f32x4*>(dxd + base + 256 * c) = v;
      }
    }
  };
  const int stride = n_wg * 4;
  int row = blockIdx.x * 4 + wave;
  if (row < M) {
    Row A, B;
    load(row, A);
    for (;;) {
      const int r1 = row + stride;
      if (r1 < M) load(r1, B);
      work(row, A);
      if (r1 >= M) break;
      const int r2 = r1 + stride;
      if (r2 < M) load(r2, A);
      work(r1, B);
      if (r2 >= M) break;
      row = r2;
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sm[wave][256 * c + 4 * lane + j] = dg[c][j];
      sm[wave][E + 256 * c + 4 * lane + j] = db[c][j];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * E; i += 256)
    partial[(size_t)blockIdx.x * 2 * E + i] = (sm[0][i] + sm[1][i]) + (sm[2][i] + sm[3][i]);
}

// dgamma -> slab[g_off + f], dbeta -> slab[b_off + f]  (fixed-order sum of the per-workgroup partials;
// 64 columns per workgroup, the 4 waves split the partial rows, then a 4-way LDS reduction)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, int nparts, int E,
                                                               float* __restrict__ slab, int64_t g_off, int64_t b_off) {
  // 16 wave groups split the partial rows (8 independent loads in flight each; 4 groups x one dependent chain of 256
  // loads took 62 us for the 1024 x 512 partials of one LayerNorm), fixed-order sums throughout
  __shared__ float sm[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (i < 2 * E) {
    int p = grp;
    for (; p + 16 * 7 < nparts; p += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(p + 16 * u) * 2 * E + i];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; p < nparts; p += 16) s += partial[(size_t)p * 2 * E + i];
  }
  sm[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && i < 2 * E) {
    float t = 0.f;
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) t += sm[gq][cl];
    slab[(i < E ? g_off + i : b_off + (i - E))] = t;
  }
}

// the same reduction for up to 16 LayerNorms in one launch (blockIdx.y = site): the eight reductions of a 3-layer step
// were eight 5 us launches in the backward chain
struct LnSites {
  int64_t g_off[16], b_off[16];
};
__global__ __launch_bounds__(1024) void ln_param_reduce_many_kernel(const float* __restrict__ partial, int64_t ws_stride,
                                                                    int nparts, int E, float* __restrict__ slab,
                                                                    const LnSites sites) {
  __shared__ float sm[16][64];
  const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cl;
  partial += (size_t)blockIdx.y * ws_stride;
  float s = 0.f;
  if (i < 2 * E) {
    int p = grp;
    for (; p + 16 * 7 < nparts; p += 16 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(p + 16 * u) * 2 * E + i];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; p < nparts; p += 16) s += partial[(size_t)p * 2 * E + i];
  }
  sm[grp][cl] = s;
  __syncthreads();
  if (grp == 0 && i < 2 * E) {
    float t = 0.f;
#pragma unroll
    for (int gq = 0; gq < 16; ++gq) t += sm[gq][cl];
    slab[(i < E ? sites.g_off[blockIdx.y] + i : sites.b_off[blockIdx.y] + (i - E))] = t;
  }
}

// ---------------- attention: one workgroup (4 waves) per (batch, head), fp32 MFMA 16x16x4 -------------
// S tokens (<= 128) x head dim d (<= 64).  Q, K, (V | V^T), dO live in LDS as row-major tiles padded to
// multiples of 16 (row stride = width + 8 floats => conflict-free ds_read_b128 fragments, as in mlp.hip);
// the S x S score / probability tiles never leave LDS.  Only the lower-triangular 16x16 blocks are
// computed (causal mask, net.py:417-418); key padding (net.py:433) masks columns inside the blocks.
// Fragment conventions (k-slot trick of mlp.hip): a "row fragment" takes 4 consecutive k of one row with a
// single ds_read_b128, a "column fragment" takes them from 4 consecutive rows with 4 ds_read_b32.
struct AttnArgs {
  const float* qkv;   // [B, S, 3E]
  const float* mask;  // [B, T] (1 = valid); token j is a padded key iff mask[b, j/rep] <= 0
  float* o;           // fwd out [B, S, E]
  const float* dout;  // bwd in  [B, S, E]
  float* dqkv;        // bwd out [B, S, 3E]
  int32_t B, S, E, H, rep;
  int32_t prefix;  // 1: token 0 of every sequence is the cost-prefix token (masked like timestep 0, cdt.py:216-218)
  // attention-probability dropout (net.py:406-409): drop_scale = 1/(1-p), 0 thresh = off
  uint32_t drop_thresh, drop_site, k0, k1;
  float drop_scale;
  const osrl_step_state_t* st;
  // round 6 (head widths 16 / 32): the keep decisions of the probability dropout, one NIBBLE (four keys of one query) per
  // byte, written by the forward launch and read back by the backward launch instead of 15 Philox calls per lane --
  // layout [B*H][key block][query row (Sp)][key quad]: the 64 lanes of a (row block, key block) pair touch 64
  // consecutive bytes.  nullptr: every launch regenerates its masks (same decisions: the bytes ARE the Philox words' tests)
  unsigned char* keep;
};

// keep-multipliers of the four probabilities P[bh][i][j0 .. j0+3] (j0 a multiple of 4) one lane owns in the register
// layout of the attention kernels.  Logical mask layout [B*H, S, Sp] (Sp = S rounded up to 16): element
// (bh*S + i)*Sp + j, so one Philox call covers exactly the four keys of a lane's accumulator -- the earlier
// [.., 16, 8] layout spent 8 calls per 16-query block and lane whatever the number of key blocks, and the regenerated
// masks were most of the kernels' VALU time.
__device__ __forceinline__ f32x4 attn_drop_mult4(const AttnArgs& a, int bh, int i, int j0, int Sp) {
  const uint32_t step = a.st ? (uint32_t)a.st->step : 0u;
  const uint64_t e4 = (((uint64_t)bh * a.S + i) * Sp + j0) >> 2;
  const osrl_rng::U4 w = osrl_rng::drop_words(e4, step, a.drop_site, a.k0, a.k1);
  return f32x4{w.x >= a.drop_thresh ? a.drop_scale : 0.f, w.y >= a.drop_thresh ? a.drop_scale : 0.f,
               w.z >= a.drop_thresh ? a.drop_scale : 0.f, w.w >= a.drop_thresh ? a.drop_scale : 0.f};
}

__device__ __forceinline__ size_t attn_keep_at(int bh, int nb, int jb, int Sp, int row, int q4) {
  return (((size_t)bh * nb + jb) * Sp + row) * 4 + (q4 >> 2);
}
__device__ __forceinline__ unsigned attn_keep_bits(const f32x4& dm) {
  return (dm[0] != 0.f ? 1u : 0u) | (dm[1] != 0.f ? 2u : 0u) | (dm[2] != 0.f ? 4u : 0u) | (dm[3] != 0.f ? 8u : 0u);
}
__device__ __forceinline__ f32x4 attn_keep_mult4(unsigned bits, float scale) {
  return f32x4{(bits & 1u) ? scale : 0.f, (bits & 2u) ? scale : 0.f, (bits & 4u) ? scale : 0.f, (bits & 8u) ? scale : 0.f};
}

// 16-lane (one DPP row) butterfly reductions: 4 VALU DPP ops instead of 4-6 ds_bpermute round trips
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
  const int x = __builtin_bit_cast(int, v);
  int r;
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  v += dpp_f(v, 2);
  v += dpp_f(v, 3);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f(v, 0));
  v = fmaxf(v, dpp_f(v, 1));
  v = fmaxf(v, dpp_f(v, 2));
  v = fmaxf(v, dpp_f(v, 3));
  return v;
}

__device__ __forceinline__ f32x4 frag_row(const float* base, int ld, int row0, int k0, int lane) {
  return *reinterpret_cast<const f32x4*>(base + (row0 + (lane & 15)) * ld + k0 + 4 * (lane >> 4));
}
__device__ __forceinline__ f32x4 frag_col(const float* base, int ld, int k0, int col0, int lane) {
  const float* p = base + (k0 + 4 * (lane >> 4)) * ld + col0 + (lane & 15);
  return f32x4{p[0], p[ld], p[2 * ld], p[3 * ld]};
}
__device__ __forceinline__ void mfma4(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
}
// load one [S, d] slice of qkv / dout into a zero-padded row-major LDS tile (and optionally its transpose).
// 8 independent global loads per thread are issued before the first LDS store so their latencies overlap
// (one load per loop iteration serialises ~40 L2 round trips per workgroup: measured 3x slower kernels).
__device__ __forceinline__ void attn_load_tile_any(const float* __restrict__ src, size_t row_stride, int S_, int d, int Sp,
                                               int dp, float* dst, int ld, float* dst_t, int ld_t) {
  const int total = Sp * dp;
  for (int base = 0; base < total; base += 8 * 256) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = base + j * 256 + (int)threadIdx.x;
      const int i = idx / dp, c = idx - i * dp;
      const bool ok = idx < total && i < S_ && c < d;
      v[j] = src[ok ? (size_t)i * row_stride + c : 0];
      v[j] = ok ? v[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = base + j * 256 + (int)threadIdx.x;
      if (idx < total) {
        const int i = idx / dp, c = idx - i * dp;
        if (dst) dst[i * ld + c] = v[j];
        if (dst_t) dst_t[c * ld_t + i] = v[j];
      }
    }
  }
}

// The same copy for d == dp == DPC (a compile-time head width) with 16-byte aligned rows: float4 per thread, row /
// column of an element by shifts.  (Round 4, PMC: the generic copy above spends two integer divisions by the run-time
// dp, a 64-bit multiply and the bounds tests on every ELEMENT -- ~60 vector instructions each, and vector instructions
// are what the attention kernels are made of: 2100 per wave in the forward against 60 MFMAs, VALU + MFMA = 95 % of the
// vector ALUs' cycles.)
template <int DPC>
__device__ __forceinline__ void attn_load_tile_v(const float* __restrict__ src, size_t row_stride, int S_, int Sp,
                                                 float* dst, int ld, float* dst_t, int ld_t) {
  constexpr int C4 = DPC / 4;  // float4 per row
  const int total4 = Sp * C4;
  for (int base = 0; base < total4; base += 4 * 256) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = base + j * 256 + (int)threadIdx.x;
      const int i = idx / C4, c4 = idx % C4;
      v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < total4 && i < S_) v[j] = *reinterpret_cast<const f32x4*>(src + (size_t)i * row_stride + 4 * c4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = base + j * 256 + (int)threadIdx.x;
      if (idx < total4) {
        const int i = idx / C4, c4 = idx % C4;
        if (dst) *reinterpret_cast<f32x4*>(dst + i * ld + 4 * c4) = v[j];
        if (dst_t) {
#pragma unroll
          for (int t = 0; t < 4; ++t) dst_t[(4 * c4 + t) * ld_t + i] = v[j][t];
        }
      }
    }
  }
}
__device__ __forceinline__ void attn_load_tile(const float* __restrict__ src, size_t row_stride, int S_, int d, int Sp,
                                               int dp, float* dst, int ld, float* dst_t, int ld_t) {
  const bool vec = d == dp && (row_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (ld & 3) == 0;
  if (vec && dp == 32) attn_load_tile_v<32>(src, row_stride, S_, Sp, dst, ld, dst_t, ld_t);
  else if (vec && dp == 64) attn_load_tile_v<64>(src, row_stride, S_, Sp, dst, ld, dst_t, ld_t);
  else if (vec && dp == 16) attn_load_tile_v<16>(src, row_stride, S_, Sp, dst, ld, dst_t, ld_t);
  else attn_load_tile_any(src, row_stride, S_, d, Sp, dp, dst, ld, dst_t, ld_t);
}

// key padding (net.py:433) as an ADDITIVE bias of the scaled scores: 0 for a valid key, -inf for a padded one (and for
// the columns past S) -- one fma per score instead of a compare / select chain
__device__ __forceinline__ void attn_key_bias(const AttnArgs& a, int b, int Sp, float* kbias) {
  const int T = (a.S - a.prefix) / a.rep;
  for (int j = threadIdx.x; j < Sp; j += blockDim.x) {
    const int t = j < a.prefix ? 0 : (j - a.prefix) / a.rep;
    kbias[j] = (j < a.S && a.mask[(size_t)b * T + t] > 0.f) ? 0.f : -INFINITY;
  }
}
// softmax in base 2: the scores are scaled by log2(e) / sqrt(d) on the way out of the accumulators, so a probability is
// ONE v_exp_f32 (exp2) behind a subtract -- expf is ~15 vector instructions, and vector instructions are what these
// kernels are made of (profiles/r4_pmc_cdt_table.txt: 2970 per wave of the backward against 210 MFMAs)
constexpr float kLog2e = 1.4426950408889634f;
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// one [16, 4-column] fragment of a row-major global matrix in frag_row layout: X[row0 + (lane & 15)][k0 + 4*(lane>>4) ..+3],
// zero outside [0, rows) x [0, cols)
__device__ __forceinline__ f32x4 gfrag_row(const float* __restrict__ base, size_t ld, int row0, int rows, int k0,
                                           int cols, int lane) {
  const int r = row0 + (lane & 15), c = k0 + 4 * (lane >> 4);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (r < rows) {
    const float* p = base + (size_t)r * ld + c;
    if (c + 3 < cols && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
      v = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (c + t < cols) v[t] = p[t];
    }
  }
  return v;
}

// Forward of the attention core, one workgroup per (sample, head).  Round 2, second version: the probabilities never
// touch LDS.  A wave owns whole 16-query row blocks and forms the TRANSPOSED score blocks S^T = K Q^T, whose
// accumulator layout (lane: query i = lane & 15, keys 4*(lane>>4) .. +3 of the block) IS the A-operand fragment layout
// of P in O = P V -- so mask, softmax (row reductions = in-register over the blocks + two cross-row-group shuffles),
// dropout and the second product all run from registers.  LDS holds only K and V^T (24 KB at S = 80, d = 32, against
// 65 KB with the Q and [S, S] tiles): six workgroups per CU instead of two, one barrier instead of four, no idle waves
// behind a serial softmax.  Q fragments come straight from global.  Row blocks are dealt to the four waves largest
// first (block ib costs ib + 1 key blocks under the causal mask).
template <int NBMAX>  // S <= 16 * NBMAX (5: CDT's 4 x 20 tokens; 8: the 128-token limit) -- sizes the register tiles
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = a.E / a.H, dp = (d + 15) & ~15, Sp = (a.S + 15) & ~15;
  const int ldq = dp + 8, ldp = Sp + 8;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *Ks = sm, *Vt = Ks + Sp * ldq, *kbias = Vt + dp * ldp;
  const float* __restrict__ base = a.qkv + (size_t)b * a.S * 3 * a.E + h * d;
  attn_load_tile(base + a.E, 3 * a.E, a.S, d, Sp, dp, Ks, ldq, nullptr, 0);
  attn_load_tile(base + 2 * a.E, 3 * a.E, a.S, d, Sp, dp, nullptr, 0, Vt, ldp);
  attn_key_bias(a, b, Sp, kbias);
  const int nb = Sp >> 4, ncb = dp >> 4;
  // deal the row blocks, largest first, each to the least loaded wave (every wave computes the same deal)
  unsigned mine = 0;
  {
    int load[4] = {0, 0, 0, 0};
    for (int ib = nb - 1; ib >= 0; --ib) {
      int w = 0;
      for (int k = 1; k < 4; ++k)
        if (load[k] < load[w]) w = k;
      load[w] += ib + 1;
      if (w == wave) mine |= 1u << ib;
    }
  }
  __syncthreads();
  const float sc2 = kLog2e / sqrtf((float)d);
  const int q4 = 4 * (lane >> 4), m = lane & 15;
  // causal mask (net.py:417-418): only the diagonal block of a row block holds keys j > i -- there, key q4 + r of the
  // block against row m of the block
  const f32x4 dbias = {q4 + 0 <= m ? 0.f : -INFINITY, q4 + 1 <= m ? 0.f : -INFINITY, q4 + 2 <= m ? 0.f : -INFINITY,
                       q4 + 3 <= m ? 0.f : -INFINITY};
  for (int ib = nb - 1; ib >= 0; --ib) {
    if (!((mine >> ib) & 1u)) continue;
    f32x4 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qf[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < ncb) qf[c] = gfrag_row(base, 3 * (size_t)a.E, ib * 16, a.S, c * 16, d, lane);
    }
    const int i = ib * 16 + m;  // the query row this lane normalises
    f32x4 s[NBMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb) {
      s[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (jb <= ib) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < ncb) mfma4(s[jb], frag_row(Ks, ldq, jb * 16, c * 16, lane), qf[c]);
        f32x4 kb = *reinterpret_cast<const f32x4*>(kbias + jb * 16 + q4);
        if (jb == ib) kb += dbias;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[jb][r] = __builtin_fmaf(s[jb][r], sc2, kb[r]);
          mx = fmaxf(mx, s[jb][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const bool live = i < a.S && mx > -INFINITY;
    const float mxs = live ? mx : 0.f;  // (a row without a valid key: exp2(-inf - 0) = 0 everywhere, and 1 / sum is not used)
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb) {
      if (jb <= ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2_fast(s[jb][r] - mxs);
          s[jb][r] = e;
          sum += e;
        }
      }
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = live ? 1.0f / sum : 0.f;
    if (a.drop_thresh && i < a.S) {  // P' = P M / (1-p): one Philox call per key block and lane
#pragma unroll
      for (int jb = 0; jb < NBMAX; ++jb)
        if (jb <= ib) s[jb] *= attn_drop_mult4(a, blockIdx.x, i, jb * 16 + q4, Sp);
    }
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
      if (jb <= ib) s[jb] *= inv;
    // O[ib] = P[ib] V : A = the probability fragments in registers, B = row fragments of V^T
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      if (cb < ncb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb)
          if (jb <= ib) mfma4(acc, s[jb], frag_row(Vt, ldp, cb * 16, jb * 16, lane));
        const int c = cb * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int io = ib * 16 + q4 + r;
          if (io < a.S && c < d) a.o[((size_t)b * a.S + io) * a.E + h * d + c] = acc[r];
        }
      }
    }
  }
}

// deal blocks 0..nb-1 with costs cost(blk) to 4 waves, most expensive first, each to the least loaded wave
__device__ __forceinline__ unsigned attn_deal(int nb, int wave, bool rows) {
  unsigned mine = 0;
  int load[4] = {0, 0, 0, 0};
  for (int t = 0; t < nb; ++t) {
    const int blk = rows ? nb - 1 - t : t;  // row block ib costs ib + 1 key blocks, key block jb costs nb - jb row blocks
    int w = 0;
    for (int k = 1; k < 4; ++k)
      if (load[k] < load[w]) w = k;
    load[w] += rows ? blk + 1 : nb - blk;
    if (w == wave) mine |= 1u << blk;
  }
  return mine;
}

// Backward of the attention core (net.py:395-417 under autograd), one workgroup per (sample, head), everything
// [S, S]-shaped in registers (third version; round 1 kept two [S, S] tiles + four operand tiles in 108 KB of LDS, the
// second version one tile + two operand tiles in 62 KB).  Two passes over the causal block triangle:
//   A, by QUERY block (the wave that owns rows i): S^T = K Q^T and dP'^T = V dO^T in the transposed accumulator layout
//      (lane: row i = lane & 15, keys 4*(lane>>4) ..+3), so softmax, r_i = sum_j P'_ij dP'_ij and
//      dS = P' dP' - P r are lane-local plus two cross-row-group shuffles, and dS is already the A fragment of
//      dQ = dS K.  Leaves per-row (max, 1/sum, r) and the keep-mask bytes of the tile in LDS.
//   B, by KEY block (the wave that owns columns j): S = Q K^T and dP' = dO V^T in the standard layout (lane: rows
//      4*(lane>>4) ..+3, key j = lane & 15), P rebuilt from the row statistics, and P'^T / dS^T are then the A fragments
//      of dV = P'^T dO and dK = dS^T Q (the contraction runs over the query rows).
// The MFMA products are recomputed (about 840 instead of 600 MFMAs per workgroup: nothing next to the latency they
// replace); LDS holds two [S, d] tiles ((K, V) in pass A, (Q, dO) in pass B; the other two operands of each pass are
// the owning wave's own rows, read straight from global as fragments), 33 KB at S = 80, d = 32: four workgroups per CU,
// three barriers.
template <int NBMAX>
__global__ __launch_bounds__(256, NBMAX <= 5 ? 3 : 2) void attn_bwd_kernel(const AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int d = a.E / a.H, dp = (d + 15) & ~15, Sp = (a.S + 15) & ~15;
  const int ldq = dp + 8;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nb = Sp >> 4, ncb = dp >> 4;
  const int q4 = 4 * (lane >> 4), m = lane & 15;
  float *T0 = sm, *T1 = T0 + Sp * ldq, *rmax = T1 + Sp * ldq, *rinv = rmax + Sp, *rdot = rinv + Sp, *kbias = rdot + Sp;
  unsigned char* Mb = reinterpret_cast<unsigned char*>(kbias + Sp);  // [Sp][Sp] keep flags
  const size_t ldg = 3 * (size_t)a.E;
  const float* __restrict__ base = a.qkv + (size_t)b * a.S * 3 * a.E + h * d;
  const float* __restrict__ dob = a.dout + (size_t)b * a.S * a.E + h * d;
  float* __restrict__ dq_out = a.dqkv + (size_t)b * a.S * 3 * a.E + h * d;
  const float scale = 1.0f / sqrtf((float)d), sc2 = kLog2e * scale;
  // causal mask on the diagonal blocks: pass A holds (row m, keys q4 + r), pass B (rows q4 + r, key m) of the block
  const f32x4 dbiasA = {q4 + 0 <= m ? 0.f : -INFINITY, q4 + 1 <= m ? 0.f : -INFINITY, q4 + 2 <= m ? 0.f : -INFINITY,
                        q4 + 3 <= m ? 0.f : -INFINITY};
  const f32x4 dbiasB = {m <= q4 + 0 ? 0.f : -INFINITY, m <= q4 + 1 ? 0.f : -INFINITY, m <= q4 + 2 ? 0.f : -INFINITY,
                        m <= q4 + 3 ? 0.f : -INFINITY};

  // ---- pass A: T0 = K, T1 = V
  attn_load_tile(base + a.E, ldg, a.S, d, Sp, dp, T0, ldq, nullptr, 0);
  attn_load_tile(base + 2 * a.E, ldg, a.S, d, Sp, dp, T1, ldq, nullptr, 0);
  attn_key_bias(a, b, Sp, kbias);
  const unsigned my_rows = attn_deal(nb, wave, true), my_cols = attn_deal(nb, wave, false);
  __syncthreads();
  for (int ib = nb - 1; ib >= 0; --ib) {
    if (!((my_rows >> ib) & 1u)) continue;
    f32x4 qf[4], dof[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qf[c] = dof[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < ncb) {
        qf[c] = gfrag_row(base, ldg, ib * 16, a.S, c * 16, d, lane);
        dof[c] = gfrag_row(dob, (size_t)a.E, ib * 16, a.S, c * 16, d, lane);
      }
    }
    const int i = ib * 16 + m;
    f32x4 s[NBMAX], g[NBMAX], pk[NBMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb) {
      s[jb] = g[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (jb <= ib) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < ncb) {
            mfma4(s[jb], frag_row(T0, ldq, jb * 16, c * 16, lane), qf[c]);
            mfma4(g[jb], frag_row(T1, ldq, jb * 16, c * 16, lane), dof[c]);  // dP'[i][j] = sum_c dO[i][c] V[j][c]
          }
        f32x4 kb = *reinterpret_cast<const f32x4*>(kbias + jb * 16 + q4);
        if (jb == ib) kb += dbiasA;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[jb][r] = __builtin_fmaf(s[jb][r], sc2, kb[r]);
          mx = fmaxf(mx, s[jb][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const bool live = i < a.S && mx > -INFINITY;
    const float mxs = live ? mx : 0.f;
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
      if (jb <= ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2_fast(s[jb][r] - mxs);
          s[jb][r] = e;
          sum += e;
        }
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = live ? 1.0f / sum : 0.f;
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
      if (jb <= ib) {
        s[jb] *= inv;  // P
        pk[jb] = s[jb];
      }
    if (a.drop_thresh) {  // P' = P M / (1-p); the keep flags of the tile go to LDS for pass B
#pragma unroll
      for (int jb = 0; jb < NBMAX; ++jb)
        if (jb <= ib) {
          const f32x4 dm = i < a.S ? attn_drop_mult4(a, blockIdx.x, i, jb * 16 + q4, Sp) : f32x4{0.f, 0.f, 0.f, 0.f};
          pk[jb] *= dm;
          const uint32_t flags = (dm[0] != 0.f ? 1u : 0u) | (dm[1] != 0.f ? 0x100u : 0u) |
                                 (dm[2] != 0.f ? 0x10000u : 0u) | (dm[3] != 0.f ? 0x1000000u : 0u);
          *reinterpret_cast<uint32_t*>(Mb + i * Sp + jb * 16 + q4) = flags;  // 4 consecutive keys, 4-byte aligned
        }
    }
    float rd = 0.f;  // r_i = sum_j P'_ij dP'_ij
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
      if (jb <= ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) rd += pk[jb][r] * g[jb][r];
      }
    rd += __shfl_xor(rd, 16);
    rd += __shfl_xor(rd, 32);
    if (q4 == 0) {
      rmax[i] = mxs;  // (base-2 domain, like the scores pass B rebuilds)
      rinv[i] = inv;
      rdot[i] = rd;
    }
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
      if (jb <= ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) g[jb][r] = pk[jb][r] * g[jb][r] - s[jb][r] * rd;  // dS
      }
    // dQ[ib] = scale * dS[ib] K : A = dS fragments (registers), B[k = j][n = c] = K[j][c]
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      if (cb < ncb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb)
          if (jb <= ib) mfma4(acc, g[jb], frag_col(T0, ldq, jb * 16, cb * 16, lane));
        const int c = cb * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int io = ib * 16 + q4 + r;
          if (io < a.S && c < d) dq_out[(size_t)io * ldg + c] = acc[r] * scale;
        }
      }
  }
  __syncthreads();  // K, V tiles are done with; row statistics and keep flags are complete

  // ---- pass B: T0 = Q, T1 = dO
  attn_load_tile(base, ldg, a.S, d, Sp, dp, T0, ldq, nullptr, 0);
  attn_load_tile(dob, (size_t)a.E, a.S, d, Sp, dp, T1, ldq, nullptr, 0);
  __syncthreads();
  for (int jb = 0; jb < nb; ++jb) {
    if (!((my_cols >> jb) & 1u)) continue;
    f32x4 kf[4], vf[4], accK[4], accV[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kf[c] = vf[c] = accK[c] = accV[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < ncb) {
        kf[c] = gfrag_row(base + a.E, ldg, jb * 16, a.S, c * 16, d, lane);
        vf[c] = gfrag_row(base + 2 * a.E, ldg, jb * 16, a.S, c * 16, d, lane);
      }
    }
    const int j = jb * 16 + m;
    const float kbj = kbias[j];
    for (int ib = jb; ib < nb; ++ib) {
      f32x4 sv = {0.f, 0.f, 0.f, 0.f}, gp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < ncb) {
          mfma4(sv, frag_row(T0, ldq, ib * 16, c * 16, lane), kf[c]);  // S[i][j]
          mfma4(gp, frag_row(T1, ldq, ib * 16, c * 16, lane), vf[c]);  // dP'[i][j]
        }
      const f32x4 mxv = *reinterpret_cast<const f32x4*>(rmax + ib * 16 + q4);
      const f32x4 ivv = *reinterpret_cast<const f32x4*>(rinv + ib * 16 + q4);
      const f32x4 rdv = *reinterpret_cast<const f32x4*>(rdot + ib * 16 + q4);
      f32x4 pp, dsv;
      f32x4 kb = {kbj, kbj, kbj, kbj};
      if (ib == jb) kb += dbiasB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib * 16 + q4 + r;
        const float P = exp2_fast(__builtin_fmaf(sv[r], sc2, kb[r]) - mxv[r]) * ivv[r];
        const float km = a.drop_thresh ? (Mb[i * Sp + j] ? a.drop_scale : 0.f) : 1.0f;
        pp[r] = P * km;                        // P'
        dsv[r] = pp[r] * gp[r] - P * rdv[r];   // dS
      }
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        if (cb < ncb) {
          mfma4(accV[cb], pp, frag_col(T1, ldq, ib * 16, cb * 16, lane));   // dV[j][c] += P'[i][j] dO[i][c]
          mfma4(accK[cb], dsv, frag_col(T0, ldq, ib * 16, cb * 16, lane));  // dK[j][c] += dS[i][j] Q[i][c]
        }
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      if (cb < ncb) {
        const int c = cb * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jo = jb * 16 + q4 + r;
          if (jo < a.S && c < d) {
            dq_out[(size_t)jo * ldg + a.E + c] = accK[cb][r] * scale;
            dq_out[(size_t)jo * ldg + 2 * a.E + c] = accV[cb][r];
          }
        }
      }
  }
}

// ======================================================================================================================
// Round 5: the same two kernels for head widths of 16 / 32 floats with 16-byte aligned rows -- every CDT configuration of
// the reference (embedding_dim / num_heads = 128 / 8, examples/configs/cdt_configs.py:24-26; BASELINE C5: 256 / 8).  The
// algorithm and the register layouts are those of attn_fwd_kernel / attn_bwd_kernel above (which stay for other widths);
// what changed is everything AROUND the MFMAs, because that is what the kernels spend their time on
// (profiles/r4_pmc_cdt_table.txt: 2970 vector instructions per wave of the backward against 210 MFMAs, 2.6 waves per SIMD):
//   * the row block a wave works on is a TEMPLATE parameter: which key blocks exist, which one is the diagonal, every LDS
//     offset and every register-array index are compile-time facts -- no exec-mask bookkeeping around `jb <= ib`;
//   * 3 waves per (sample, head) at 5 row blocks: row block ib costs ib + 1 key blocks, so four waves carry 5 / 4 / 3 / 3
//     blocks and wait a quarter of the time at the barriers; three carry 5 / 4 + 1 / 3 + 2.  (2 waves up to 4 row blocks,
//     3 at 5-6, 4 at 7-8: the splits without a remainder);
//   * base-2 softmax with additive masks (attn_key_bias, exp2_fast), the causal test only on diagonal blocks;
//   * operand tiles by guard-free float4 copies (rows clamped, zeros selected afterwards), pointers formed once.
template <int NW>
__device__ __forceinline__ unsigned attn_deal_n(int nb, int wave, bool rows) {
  unsigned mine = 0;
  int load[NW];
#pragma unroll
  for (int k = 0; k < NW; ++k) load[k] = 0;
  for (int t = 0; t < nb; ++t) {
    const int blk = rows ? nb - 1 - t : t;
    int w = 0;
#pragma unroll
    for (int k = 1; k < NW; ++k)
      if (load[k] < load[w]) w = k;
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (k == w) load[k] += rows ? blk + 1 : nb - blk;
    if (w == wave) mine |= 1u << blk;
  }
  return mine;
}

// [S, 16 NCB] slice of a row-major global matrix -> zero-padded LDS tile (row pitch 16 NCB + 8) and / or its transpose, in
// two halves: every float4 of the slice is REQUESTED by attn_tile_ld (at most four per thread: S <= 16 NB, 64 NW threads,
// see attn_launch_v) and lands in LDS by attn_tile_st -- so a workgroup puts all the global loads of a phase in flight
// before it waits for the first one (two tiles + the key mask one after the other were three round trips, 5.5 us of a
// 29 us wave life in tools/attn_lab.hip's stamps), and the second pass's tiles travel under the first pass's arithmetic
template <int NCB, int NT>
__device__ __forceinline__ void attn_tile_ld(const float* __restrict__ src, int row_stride, int S_, f32x4 (&v)[4]) {
  constexpr int C4 = NCB * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = k * NT + (int)threadIdx.x;
    const int i = idx / C4, c4 = idx % C4;
    const int ic = i < S_ ? i : S_ - 1;
    v[k] = *reinterpret_cast<const f32x4*>(src + __mul24(ic, row_stride) + 4 * c4);
  }
}
template <int NCB, int NT>
__device__ __forceinline__ void attn_tile_st(const f32x4 (&v)[4], int S_, int Sp, float* dst, float* dst_t, int ld_t) {
  constexpr int C4 = NCB * 4, LD = NCB * 16 + 8;
  const int total4 = Sp * C4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = k * NT + (int)threadIdx.x;
    const int i = idx / C4, c4 = idx % C4;
    if (idx < total4) {
      const f32x4 w = i < S_ ? v[k] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (dst) *reinterpret_cast<f32x4*>(dst + i * LD + 4 * c4) = w;
      if (dst_t) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dst_t[(4 * c4 + t) * ld_t + i] = w[t];
      }
    }
  }
}
// the key-padding bias in the same two halves (thread j < Sp owns key j; the division by `rep` as a multiply: exact for
// j < 128)
__device__ __forceinline__ float attn_key_mask_ld(const AttnArgs& a, int b, int Sp) {
  const int T = (a.S - a.prefix) / a.rep, j = threadIdx.x;
  const int t = j < a.prefix ? 0 : (int)(((float)(j - a.prefix) + 0.5f) * (1.0f / (float)a.rep));
  return a.mask[(size_t)b * T + (j < a.S ? t : 0)];
}
__device__ __forceinline__ void attn_key_bias_st(const AttnArgs& a, float mv, int Sp, float* kbias) {
  const int j = threadIdx.x;
  if (j < Sp) kbias[j] = (j < a.S && mv > 0.f) ? 0.f : -INFINITY;
}

struct AttnCtx {
  const float *T0, *T1;        // the two operand tiles of the running pass
  float *rmax, *rinv, *rdot;   // per-row softmax statistics (pass A -> pass B)
  const float* kbias;
  unsigned char* Mb;           // keep flags [16 NB][16 NB]
  const float *q, *dob;        // this (sample, head)'s q columns of qkv (k at + E, v at + 2 E) / dout columns
  float* dq;                   // its dq columns of dqkv
  int ldg, Sp, lane, m, q4, bh;
  float scale, sc2;
  f32x4 dbiasA, dbiasB;
};

// one [16, 4]-fragment per 16 columns of row (row0 + m) of a global matrix, zero for rows >= S
template <int NCB>
__device__ __forceinline__ void attn_gfrags(const float* __restrict__ base, int ld, int row, int S_, int q4, f32x4 (&f)[NCB]) {
  const bool ok = row < S_;
  const float* __restrict__ p = base + (__mul24(ok ? row : S_ - 1, ld) + q4);
#pragma unroll
  for (int c = 0; c < NCB; ++c) f[c] = *reinterpret_cast<const f32x4*>(p + 16 * c);
#pragma unroll
  for (int c = 0; c < NCB; ++c) f[c] = ok ? f[c] : f32x4{0.f, 0.f, 0.f, 0.f};
}

// forward, row block IB of one (sample, head): O[IB] = dropout(softmax(Q[IB] K^T)) V
template <int NB, int NCB, int IB>
__device__ __forceinline__ void attn_fwd_rows(const AttnCtx& c, const AttnArgs& a, float* __restrict__ o_rows) {
  constexpr int LDQ = 16 * NCB + 8, LDP = 16 * NB + 8;
  const int lane = c.lane, m = c.m, q4 = c.q4;
  const int row = IB * 16 + m;
  f32x4 qf[NCB];
  attn_gfrags<NCB>(c.q, c.ldg, row, a.S, q4, qf);
  f32x4 s[IB + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb) {
    s[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCB; ++cc) mfma4(s[jb], frag_row(c.T0, LDQ, jb * 16, cc * 16, lane), qf[cc]);
    f32x4 kb = *reinterpret_cast<const f32x4*>(c.kbias + jb * 16 + q4);
    if (jb == IB) kb += c.dbiasA;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[jb][r] = __builtin_fmaf(s[jb][r], c.sc2, kb[r]);
      mx = fmaxf(mx, s[jb][r]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const bool live = row < a.S && mx > -INFINITY;
  const float mxs = live ? mx : 0.f;
  float sum = 0.f;
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[jb][r] = exp2_fast(s[jb][r] - mxs);
      sum += s[jb][r];
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = live ? 1.0f / sum : 0.f;
  if (a.drop_thresh && row < a.S) {
#pragma unroll
    for (int jb = 0; jb <= IB; ++jb) {
      const f32x4 dm = attn_drop_mult4(a, c.bh, row, jb * 16 + q4, c.Sp);
      if (a.keep) a.keep[attn_keep_at(c.bh, c.Sp >> 4, jb, c.Sp, row, q4)] = (unsigned char)attn_keep_bits(dm);
      s[jb] *= dm;
    }
  }
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb) s[jb] *= inv;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb <= IB; ++jb) mfma4(acc, s[jb], frag_row(c.T1, LDP, cb * 16, jb * 16, lane));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = IB * 16 + q4 + r;
      if (io < a.S) o_rows[__mul24(io, a.E) + cb * 16 + m] = acc[r];
    }
  }
}
template <int NB, int NCB, int IB>
__device__ __forceinline__ void attn_fwd_rows_from(const AttnCtx& c, const AttnArgs& a, float* __restrict__ o_rows, unsigned mine) {
  if constexpr (IB >= 0) {
    if ((mine >> IB) & 1u) attn_fwd_rows<NB, NCB, IB>(c, a, o_rows);
    attn_fwd_rows_from<NB, NCB, IB - 1>(c, a, o_rows, mine);
  }
}

template <int NB, int NCB, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_v_kernel(const AttnArgs a) {
  constexpr int NT = 64 * NW, D = 16 * NCB, LDQ = D + 8, LDP = 16 * NB + 8;
  __shared__ __attribute__((aligned(16))) float Ks[16 * NB * LDQ];
  __shared__ __attribute__((aligned(16))) float Vt[D * LDP];
  __shared__ __attribute__((aligned(16))) float kbias[16 * NB];
  const int Sp = (a.S + 15) & ~15, nb = Sp >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  AttnCtx c;
  c.lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  c.m = c.lane & 15;
  c.q4 = 4 * (c.lane >> 4);
  c.ldg = 3 * a.E;
  c.Sp = Sp;
  c.bh = blockIdx.x;
  c.q = a.qkv + (size_t)b * a.S * c.ldg + h * D;
  c.T0 = Ks;
  c.T1 = Vt;
  c.kbias = kbias;
  c.sc2 = kLog2e / sqrtf((float)D);
  c.dbiasA = f32x4{c.q4 + 0 <= c.m ? 0.f : -INFINITY, c.q4 + 1 <= c.m ? 0.f : -INFINITY,
                   c.q4 + 2 <= c.m ? 0.f : -INFINITY, c.q4 + 3 <= c.m ? 0.f : -INFINITY};
  {
    f32x4 tk[4], tv[4];
    attn_tile_ld<NCB, NT>(c.q + a.E, c.ldg, a.S, tk);
    attn_tile_ld<NCB, NT>(c.q + 2 * a.E, c.ldg, a.S, tv);
    const float mv = attn_key_mask_ld(a, b, Sp);
    attn_tile_st<NCB, NT>(tk, a.S, Sp, Ks, nullptr, 0);
    attn_tile_st<NCB, NT>(tv, a.S, Sp, nullptr, Vt, LDP);
    attn_key_bias_st(a, mv, Sp, kbias);
  }
  const unsigned mine = attn_deal_n<NW>(nb, wave, true);
  __syncthreads();
  attn_fwd_rows_from<NB, NCB, NB - 1>(c, a, a.o + (size_t)b * a.S * a.E + h * D, mine);
}

// backward, pass A for row block IB (statistics, keep flags, dQ)
template <int NB, int NCB, int IB>
__device__ __forceinline__ void attn_bwd_rows(const AttnCtx& c, const AttnArgs& a) {
  constexpr int LDQ = 16 * NCB + 8, MP = 16 * NB;
  const int lane = c.lane, m = c.m, q4 = c.q4;
  const int row = IB * 16 + m;
  f32x4 qf[NCB], dof[NCB];
  attn_gfrags<NCB>(c.q, c.ldg, row, a.S, q4, qf);
  attn_gfrags<NCB>(c.dob, a.E, row, a.S, q4, dof);
  unsigned kbits[IB + 1];  // the forward launch's keep decisions (requested here, used behind the softmax)
  if (a.drop_thresh && a.keep) {
#pragma unroll
    for (int jb = 0; jb <= IB; ++jb)
      kbits[jb] = row < a.S ? a.keep[attn_keep_at(c.bh, c.Sp >> 4, jb, c.Sp, row, q4)] : 0u;
  }
  f32x4 s[IB + 1], g[IB + 1], pk[IB + 1];
  float mx = -INFINITY;
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb) {
    s[jb] = g[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCB; ++cc) {
      mfma4(s[jb], frag_row(c.T0, LDQ, jb * 16, cc * 16, lane), qf[cc]);
      mfma4(g[jb], frag_row(c.T1, LDQ, jb * 16, cc * 16, lane), dof[cc]);  // dP'[i][j] = sum_c dO[i][c] V[j][c]
    }
    f32x4 kb = *reinterpret_cast<const f32x4*>(c.kbias + jb * 16 + q4);
    if (jb == IB) kb += c.dbiasA;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[jb][r] = __builtin_fmaf(s[jb][r], c.sc2, kb[r]);
      mx = fmaxf(mx, s[jb][r]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  const bool live = row < a.S && mx > -INFINITY;
  const float mxs = live ? mx : 0.f;
  float sum = 0.f;
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[jb][r] = exp2_fast(s[jb][r] - mxs);
      sum += s[jb][r];
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = live ? 1.0f / sum : 0.f;
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb) {
    s[jb] *= inv;  // P
    pk[jb] = s[jb];
  }
  if (a.drop_thresh) {  // P' = P M / (1-p); the keep flags of the tile go to LDS for pass B
#pragma unroll
    for (int jb = 0; jb <= IB; ++jb) {
      const f32x4 dm = a.keep ? attn_keep_mult4(kbits[jb], a.drop_scale)
                              : (row < a.S ? attn_drop_mult4(a, c.bh, row, jb * 16 + q4, c.Sp) : f32x4{0.f, 0.f, 0.f, 0.f});
      pk[jb] *= dm;
      const uint32_t flags = (dm[0] != 0.f ? 1u : 0u) | (dm[1] != 0.f ? 0x100u : 0u) | (dm[2] != 0.f ? 0x10000u : 0u) |
                             (dm[3] != 0.f ? 0x1000000u : 0u);
      *reinterpret_cast<uint32_t*>(c.Mb + row * MP + jb * 16 + q4) = flags;
    }
  }
  float rd = 0.f;  // r_i = sum_j P'_ij dP'_ij
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) rd = __builtin_fmaf(pk[jb][r], g[jb][r], rd);
  rd += __shfl_xor(rd, 16);
  rd += __shfl_xor(rd, 32);
  if (q4 == 0) {
    c.rmax[row] = mxs;
    c.rinv[row] = inv;
    c.rdot[row] = rd;
  }
#pragma unroll
  for (int jb = 0; jb <= IB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) g[jb][r] = pk[jb][r] * g[jb][r] - s[jb][r] * rd;  // dS
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {  // dQ[IB] = scale * dS[IB] K
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb <= IB; ++jb) mfma4(acc, g[jb], frag_col(c.T0, LDQ, jb * 16, cb * 16, lane));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = IB * 16 + q4 + r;
      if (io < a.S) c.dq[__mul24(io, c.ldg) + cb * 16 + m] = acc[r] * c.scale;
    }
  }
}
template <int NB, int NCB, int IB>
__device__ __forceinline__ void attn_bwd_rows_from(const AttnCtx& c, const AttnArgs& a, unsigned mine) {
  if constexpr (IB >= 0) {
    if ((mine >> IB) & 1u) attn_bwd_rows<NB, NCB, IB>(c, a);
    attn_bwd_rows_from<NB, NCB, IB - 1>(c, a, mine);
  }
}

// backward, pass B for key block jb (dK, dV): T0 = Q, T1 = dO
template <int NB, int NCB>
__device__ __forceinline__ void attn_bwd_cols(const AttnCtx& c, const AttnArgs& a, const int jb, const int nb) {
  constexpr int LDQ = 16 * NCB + 8, MP = 16 * NB;
  const int lane = c.lane, m = c.m, q4 = c.q4;
  const int col = jb * 16 + m;
  f32x4 kf[NCB], vf[NCB], accK[NCB], accV[NCB];
  attn_gfrags<NCB>(c.q + a.E, c.ldg, col, a.S, q4, kf);
  attn_gfrags<NCB>(c.q + 2 * a.E, c.ldg, col, a.S, q4, vf);
#pragma unroll
  for (int cc = 0; cc < NCB; ++cc) accK[cc] = accV[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float kbj = c.kbias[col];
  auto block = [&](const int ib, const bool diag) __attribute__((always_inline)) {
    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, gp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCB; ++cc) {
      mfma4(sv, frag_row(c.T0, LDQ, ib * 16, cc * 16, lane), kf[cc]);  // S[i][j]
      mfma4(gp, frag_row(c.T1, LDQ, ib * 16, cc * 16, lane), vf[cc]);  // dP'[i][j]
    }
    const f32x4 mxv = *reinterpret_cast<const f32x4*>(c.rmax + ib * 16 + q4);
    const f32x4 ivv = *reinterpret_cast<const f32x4*>(c.rinv + ib * 16 + q4);
    const f32x4 rdv = *reinterpret_cast<const f32x4*>(c.rdot + ib * 16 + q4);
    f32x4 kb = {kbj, kbj, kbj, kbj};
    if (diag) kb += c.dbiasB;
    const unsigned char* mb = c.Mb + (ib * 16 + q4) * MP + col;
    f32x4 pp, dsv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float P = exp2_fast(__builtin_fmaf(sv[r], c.sc2, kb[r]) - mxv[r]) * ivv[r];
      const float km = a.drop_thresh ? (mb[r * MP] ? a.drop_scale : 0.f) : 1.0f;
      pp[r] = P * km;                        // P'
      dsv[r] = pp[r] * gp[r] - P * rdv[r];   // dS
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      mfma4(accV[cb], pp, frag_col(c.T1, LDQ, ib * 16, cb * 16, lane));   // dV[j][c] += P'[i][j] dO[i][c]
      mfma4(accK[cb], dsv, frag_col(c.T0, LDQ, ib * 16, cb * 16, lane));  // dK[j][c] += dS[i][j] Q[i][c]
    }
  };
  block(jb, true);
  for (int ib = jb + 1; ib < nb; ++ib) block(ib, false);
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jo = jb * 16 + q4 + r;
      if (jo < a.S) {
        c.dq[__mul24(jo, c.ldg) + a.E + cb * 16 + m] = accK[cb][r] * c.scale;
        c.dq[__mul24(jo, c.ldg) + 2 * a.E + cb * 16 + m] = accV[cb][r];
      }
    }
}

#ifdef ATTN_STAMPS  // tools/attn_lab.hip: 100 MHz stamps per wave of the first 512 workgroups (lab builds only)
__device__ unsigned long long g_attn_stamp[512][4][8];
#define ATTN_STAMP(i) if ((threadIdx.x & 63) == 0 && blockIdx.x >= 4096 && blockIdx.x < 4608) g_attn_stamp[blockIdx.x - 4096][threadIdx.x >> 6][i] = __builtin_amdgcn_s_memrealtime();
#else
#define ATTN_STAMP(i)
#endif
template <int NB, int NCB, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_v_kernel(const AttnArgs a) {
  constexpr int NT = 64 * NW, D = 16 * NCB, LDQ = D + 8, SPM = 16 * NB;
  __shared__ __attribute__((aligned(16))) float T0[SPM * LDQ];
  __shared__ __attribute__((aligned(16))) float T1[SPM * LDQ];
  __shared__ __attribute__((aligned(16))) float stats[4 * SPM];  // row max | 1 / sum | r | key bias
  __shared__ __attribute__((aligned(16))) unsigned char Mb[SPM * SPM];
  const int Sp = (a.S + 15) & ~15, nb = Sp >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x - b * a.H;
  AttnCtx c;
  c.lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  c.m = c.lane & 15;
  c.q4 = 4 * (c.lane >> 4);
  c.ldg = 3 * a.E;
  c.Sp = Sp;
  c.bh = blockIdx.x;
  c.q = a.qkv + (size_t)b * a.S * c.ldg + h * D;
  c.dob = a.dout + (size_t)b * a.S * a.E + h * D;
  c.dq = a.dqkv + (size_t)b * a.S * c.ldg + h * D;
  c.T0 = T0;
  c.T1 = T1;
  c.rmax = stats;
  c.rinv = stats + SPM;
  c.rdot = stats + 2 * SPM;
  c.kbias = stats + 3 * SPM;
  c.Mb = Mb;
  c.scale = 1.0f / sqrtf((float)D);
  c.sc2 = kLog2e * c.scale;
  c.dbiasA = f32x4{c.q4 + 0 <= c.m ? 0.f : -INFINITY, c.q4 + 1 <= c.m ? 0.f : -INFINITY,
                   c.q4 + 2 <= c.m ? 0.f : -INFINITY, c.q4 + 3 <= c.m ? 0.f : -INFINITY};
  c.dbiasB = f32x4{c.m <= c.q4 + 0 ? 0.f : -INFINITY, c.m <= c.q4 + 1 ? 0.f : -INFINITY,
                   c.m <= c.q4 + 2 ? 0.f : -INFINITY, c.m <= c.q4 + 3 ? 0.f : -INFINITY};
  // ---- pass A: T0 = K, T1 = V
  ATTN_STAMP(0);
  constexpr bool kAhead = NB <= 5;  // pass B's tiles requested before pass A (32 registers; the 8-block forms keep theirs)
  f32x4 tq[4], tdo[4];
  {
    f32x4 tk[4], tv[4];
    attn_tile_ld<NCB, NT>(c.q + a.E, c.ldg, a.S, tk);
    attn_tile_ld<NCB, NT>(c.q + 2 * a.E, c.ldg, a.S, tv);
    const float mv = attn_key_mask_ld(a, b, Sp);
    attn_tile_st<NCB, NT>(tk, a.S, Sp, T0, nullptr, 0);
    attn_tile_st<NCB, NT>(tv, a.S, Sp, T1, nullptr, 0);
    attn_key_bias_st(a, mv, Sp, stats + 3 * SPM);
  }
  if (kAhead) {
    attn_tile_ld<NCB, NT>(c.q, c.ldg, a.S, tq);
    attn_tile_ld<NCB, NT>(c.dob, a.E, a.S, tdo);
  }
  const unsigned my_rows = attn_deal_n<NW>(nb, wave, true), my_cols = attn_deal_n<NW>(nb, wave, false);
  __syncthreads();
  ATTN_STAMP(1);
  attn_bwd_rows_from<NB, NCB, NB - 1>(c, a, my_rows);
  ATTN_STAMP(2);
  __syncthreads();  // K, V tiles are done with; row statistics and keep flags are complete
  ATTN_STAMP(3);
  // ---- pass B: T0 = Q, T1 = dO
  if (!kAhead) {
    attn_tile_ld<NCB, NT>(c.q, c.ldg, a.S, tq);
    attn_tile_ld<NCB, NT>(c.dob, a.E, a.S, tdo);
  }
  attn_tile_st<NCB, NT>(tq, a.S, Sp, T0, nullptr, 0);
  attn_tile_st<NCB, NT>(tdo, a.S, Sp, T1, nullptr, 0);
  __syncthreads();
  ATTN_STAMP(4);
  for (int jb = 0; jb < nb; ++jb)
    if ((my_cols >> jb) & 1u) attn_bwd_cols<NB, NCB>(c, a, jb, nb);
  ATTN_STAMP(5);
}

// ---------------- dropout (nn.Dropout: cdt.py:87,222; net.py:404,414,439)
// y = x * keep/(1-p); the same call on the incoming gradient is the backward pass (the mask is a pure function of
// (seed, step, site, element), philox.h).  HBM-streaming: one Philox call per float4.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                      uint32_t thresh, float scale, uint32_t site, uint32_t k0,
                                                      uint32_t k1, const osrl_step_state_t* __restrict__ st) {
  const uint32_t step = st ? (uint32_t)st->step : 0u;
  const int64_t n4 = (n + 3) >> 2;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const osrl_rng::U4 w = osrl_rng::drop_words((uint64_t)i, step, site, k0, k1);
    const float m[4] = {w.x >= thresh ? scale : 0.f, w.y >= thresh ? scale : 0.f, w.z >= thresh ? scale : 0.f,
                        w.w >= thresh ? scale : 0.f};
    const int64_t b = i * 4;
    if (vec && b + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + b);
      *reinterpret_cast<float4*>(y + b) = make_float4(v.x * m[0], v.y * m[1], v.z * m[2], v.w * m[3]);
    } else {
      for (int k = 0; k < 4; ++k)
        if (b + k < n) y[b + k] = x[b + k] * m[k];
    }
  }
}

// ---------------- GELU
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    reinterpret_cast<f32x4*>(y)[i] = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
  }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i], g = reinterpret_cast<const f32x4*>(dy)[i];
    reinterpret_cast<f32x4*>(dx)[i] = f32x4{g[0] * gelu_g(v[0]), g[1] * gelu_g(v[1]), g[2] * gelu_g(v[2]),
                                           g[3] * gelu_g(v[3])};
  }
}

// ---------------- heads' loss (cdt.py:357-394) ----------------
struct LossArgs {
  const float *head, *logits, *sp, *actions, *states, *mask, *costs;  // head = (mu|log_std) or the action prediction
  float *dhead, *dlogits, *dsp, *stat, *ent_out;
  const float* log_temp;
  const osrl_step_state_t* st;
  int32_t B, T, od, ad, stochastic, no_entropy, warmup;
  float cost_w, state_w, lr;
  const float* counts;  // data parallel: {global #valid tokens, global sum(mask)} (all-reduced); NULL = local
  int32_t world;        // ranks (equal per-rank batches): every 1/(B*T) uses B*world
  float stat_share;     // 1/world for the statistics that are already global
};
// stat layout: 0 nll, 1 ent, 2 ent_reg, 3 all_loss, 4 act_loss, 5 cost_loss, 6 cost_acc, 7 state_loss, 8 train_lr
// One workgroup: the whole loss incl. the statistics (small batches).  Several workgroups (``ws`` given, big batches):
// each handles a 1024-token slice (the normalisers come from ``counts``, computed before by mask_counts_kernel), writes
// the gradients of its tokens and its six partial sums to ws[block][8]; cdt_loss_finish_kernel adds the partials in
// block order (deterministic) and writes the statistics.  (Round 1: one workgroup walked all 20480 tokens of C5: 350 us.)
__device__ __forceinline__ void cdt_loss_stats(const LossArgs& a, float ll, float ent, float act_mse, float closs,
                                               float correct, float sloss, float inv_nv, float inv_bt, float inv_s,
                                               float ent_reg, float msum) {
  ll *= inv_nv;
  ent *= inv_nv;
  const float act_loss = a.stochastic ? -(ll + ent_reg * ent) : act_mse * inv_bt / (float)a.ad;
  const float cost_loss = closs * inv_bt, state_loss = sloss * inv_s;
  float* s = a.stat;
  s[0] = -ll;
  s[1] = ent;
  s[2] = ent_reg * a.stat_share;
  s[3] = act_loss + a.cost_w * cost_loss + a.state_w * state_loss;
  s[4] = act_loss;
  s[5] = cost_loss;
  s[6] = correct / msum;
  s[7] = state_loss;
  // scheduler.get_last_lr() AFTER scheduler.step(): the factor of the NEXT optimizer step  cdt.py:409,417
  const double tn = (double)(a.st->step + 1);
  s[8] = a.stat_share * a.lr * (a.warmup > 0 ? (float)fmin(tn / (double)a.warmup, 1.0) : 1.0f);
  if (a.ent_out) a.ent_out[0] = ent;
}

__global__ __launch_bounds__(1024) void cdt_loss_kernel(const LossArgs a, float* __restrict__ ws) {
  __shared__ float sm[20];
  const int BT = a.B * a.T, ad = a.ad, od = a.od;
  float nvalid = 0.f, msum = 0.f;
  if (a.counts) {
    nvalid = a.counts[0];
    msum = a.counts[1];
  } else {
    for (int i = threadIdx.x; i < BT; i += 1024) {
      nvalid += a.mask[i] > 0.f ? 1.f : 0.f;
      msum += a.mask[i];
    }
    nvalid = block_sum1024(nvalid, sm);
    msum = block_sum1024(msum, sm);
  }
  const float inv_nv = 1.0f / (fmaxf(nvalid, 1.f) * (float)ad);
  const float temp = expf(a.log_temp ? a.log_temp[0] : 0.f);
  const float ent_reg = (a.stochastic && !a.no_entropy) ? temp : 0.f;
  float ll = 0.f, ent = 0.f, act_mse = 0.f, closs = 0.f, correct = 0.f, sloss = 0.f;
  const float inv_bt = 1.0f / ((float)BT * (float)a.world);
  const float inv_s = (a.T > 1) ? 1.0f / ((float)a.B * (float)a.world * (float)(a.T - 1) * (float)od) : 0.f;
  for (int i = blockIdx.x * 1024 + threadIdx.x; i < BT; i += 1024 * gridDim.x) {
    const float m = a.mask[i];
    const bool valid = m > 0.f;
    if (a.stochastic) {  // Normal(mu, exp(ls)).log_prob / entropy, mean over valid tokens x action dims
      for (int k = 0; k < ad; ++k) {
        const float mu = a.head[(size_t)i * 2 * ad + k], ls = a.head[(size_t)i * 2 * ad + ad + k];
        const float sd = expf(ls), z = (a.actions[(size_t)i * ad + k] - mu) / sd;
        if (valid) {
          ll += -0.5f * z * z - ls - 0.9189385332046727f;
          ent += 1.4189385332046727f + ls;
        }
        a.dhead[(size_t)i * 2 * ad + k] = valid ? -(z / sd) * inv_nv : 0.f;
        a.dhead[(size_t)i * 2 * ad + ad + k] = valid ? (-(z * z - 1.0f) - ent_reg) * inv_nv : 0.f;
      }
    } else {  // F.mse_loss(...,'none') * mask, mean over ALL B*T*ad
      for (int k = 0; k < ad; ++k) {
        const float dlt = a.head[(size_t)i * ad + k] - a.actions[(size_t)i * ad + k];
        act_mse += dlt * dlt * m;
        a.dhead[(size_t)i * ad + k] = 2.0f * dlt * m * inv_bt / (float)ad;
      }
    }
    // cost head: log_softmax over 2 classes, nll * mask, mean over ALL B*T
    const float l0 = a.logits[2 * i], l1 = a.logits[2 * i + 1];
    const float mx = fmaxf(l0, l1), lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
    const int cls = a.costs[i] > 0.5f ? 1 : 0;
    closs += -((cls ? l1 : l0) - lse) * m;
    correct += (((l1 > l0) ? 1 : 0) == cls) ? m : 0.f;
    const float p0 = expf(l0 - lse), p1 = expf(l1 - lse);
    a.dlogits[2 * i] = (p0 - (cls == 0 ? 1.f : 0.f)) * m * inv_bt * a.cost_w;
    a.dlogits[2 * i + 1] = (p1 - (cls == 1 ? 1.f : 0.f)) * m * inv_bt * a.cost_w;
    // state head: predict next state, mask[:, :-1]
    const int t = i % a.T;
    for (int k = 0; k < od; ++k) {
      float gsp = 0.f;
      if (t < a.T - 1) {
        const float dlt = a.sp[(size_t)i * od + k] - a.states[(size_t)(i + 1) * od + k];
        sloss += dlt * dlt * m;
        gsp = 2.0f * dlt * m * inv_s * a.state_w;
      }
      a.dsp[(size_t)i * od + k] = gsp;
    }
  }
  ll = block_sum1024(ll, sm);
  ent = block_sum1024(ent, sm);
  act_mse = block_sum1024(act_mse, sm);
  closs = block_sum1024(closs, sm);
  correct = block_sum1024(correct, sm);
  sloss = block_sum1024(sloss, sm);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {
      cdt_loss_stats(a, ll, ent, act_mse, closs, correct, sloss, inv_nv, inv_bt, inv_s, ent_reg, msum);
    } else {
      float* w = ws + 8 * blockIdx.x;
      w[0] = ll; w[1] = ent; w[2] = act_mse; w[3] = closs; w[4] = correct; w[5] = sloss;
    }
  }
}

__global__ void cdt_loss_finish_kernel(const LossArgs a, const float* __restrict__ ws, int nblk) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nblk; ++b)
    for (int k = 0; k < 6; ++k) v[k] += ws[8 * b + k];
  const int BT = a.B * a.T;
  const float nvalid = a.counts[0], msum = a.counts[1];
  const float inv_nv = 1.0f / (fmaxf(nvalid, 1.f) * (float)a.ad);
  const float temp = expf(a.log_temp ? a.log_temp[0] : 0.f);
  const float ent_reg = (a.stochastic && !a.no_entropy) ? temp : 0.f;
  const float inv_bt = 1.0f / ((float)BT * (float)a.world);
  const float inv_s = (a.T > 1) ? 1.0f / ((float)a.B * (float)a.world * (float)(a.T - 1) * (float)a.od) : 0.f;
  cdt_loss_stats(a, v[0], v[1], v[2], v[3], v[4], v[5], inv_nv, inv_bt, inv_s, ent_reg, msum);
}

__global__ __launch_bounds__(1024) void mask_counts_kernel(const float* __restrict__ mask, int BT, float* out) {
  __shared__ float sm[20];
  float nvalid = 0.f, msum = 0.f;
  for (int i = threadIdx.x; i < BT; i += 1024) {
    nvalid += mask[i] > 0.f ? 1.f : 0.f;
    msum += mask[i];
  }
  nvalid = block_sum1024(nvalid, sm);
  msum = block_sum1024(msum, sm);
  if (threadIdx.x == 0) {
    out[0] = nvalid;
    out[1] = msum;
  }
}

// d timestep_emb[time[b,t]] += sum of the 4 token gradients of (b,t)   (scatter, fp32 atomics)
__global__ void te_scatter_kernel(const float* __restrict__ dseq, const int64_t* __restrict__ time_steps, int BT, int E,
                                  int T, int R, int prefix, float* __restrict__ dte) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)BT * E) return;
  const int bt = (int)(i / E), f = (int)(i - (int64_t)bt * E);
  const int b = bt / T, t = bt - b * T;
  const float* __restrict__ p = dseq + ((size_t)b * (R * T + prefix) + prefix + (size_t)t * R) * E + f;
  float v;
  if (R == 4) {
    v = (p[0] + p[E]) + (p[2 * E] + p[3 * E]);
  } else {
    v = p[0] + p[E];
    if (R == 3) v += p[2 * E];
  }
  atomicAdd(&dte[(size_t)time_steps[bt] * E + f], v);
}

// clip_grad_norm_: scale = min(1, clip / (||g||_2 + 1e-6))   two-stage deterministic reduction
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, int64_t n4,
                                                            float* __restrict__ partial) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// one wave: lane l sums partials l, l + 64, ... in fp64, then a fixed butterfly over the lanes (one thread walking 512
// partials was a 26 us chain of dependent loads and adds in the step's tail, profiles/r5_cdt_trace_summary.txt)
__global__ __launch_bounds__(64) void clip_scale_kernel(const float* __restrict__ partial, int n, float clip,
                                                        float* __restrict__ out) {
  double t = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) t += (double)partial[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o);
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(t);
    out[0] = clip > 0.f ? fminf(1.0f, clip / (norm + 1e-6f)) : 1.0f;
    out[1] = norm;
  }
}

// Adam(lr) on the scalar log_temperature with loss = exp(logT) * (entropy - target)   cdt.py:402-407
__global__ void temperature_step_kernel(float* logT, float* mv, const float* ent, float target, float lr, float b1,
                                        float b2, float eps, const osrl_step_state_t* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float g = expf(logT[0]) * (ent[0] - target);
    mv[0] = b1 * mv[0] + (1.0f - b1) * g;
    mv[1] = b2 * mv[1] + (1.0f - b2) * g * g;
    logT[0] -= (lr / st->bc1) * (mv[0] / (sqrtf(mv[1]) / st->bc2_sqrt + eps));
  }
}

// the round-5 kernels: head width 16 or 32, every row of every operand 16-byte aligned
static bool attn_vec_ok(int E, int H, const void* p0, const void* p1, const void* p2, const void* p3) {
  const int d = E / H;
  if (d != 16 && d != 32) return false;
  for (const void* p : {p0, p1, p2, p3})
    if (p && (reinterpret_cast<uintptr_t>(p) & 15)) return false;
  return true;
}
template <bool FWD, int NB, int NCB, int NW>
static void attn_launch_one(const AttnArgs& a, hipStream_t stream) {
  if (FWD)
    hipLaunchKernelGGL((attn_fwd_v_kernel<NB, NCB, NW>), dim3(a.B * a.H), dim3(64 * NW), 0, stream, a);
  else
    hipLaunchKernelGGL((attn_bwd_v_kernel<NB, NCB, NW>), dim3(a.B * a.H), dim3(64 * NW), 0, stream, a);
}
template <bool FWD>
static void attn_launch_v(const AttnArgs& a, hipStream_t stream) {
  const int nb = (a.S + 15) / 16, ncb = a.E / a.H / 16;
  // waves per (sample, head): the splits of 1 + 2 + .. + nb key blocks without a remainder (2 up to 4 row blocks, 3 at
  // 5-6, 4 at 7-8)
  if (nb <= 4) {
    if (ncb == 1) attn_launch_one<FWD, 5, 1, 2>(a, stream); else attn_launch_one<FWD, 5, 2, 2>(a, stream);
  } else if (nb == 5) {
    if (ncb == 1) attn_launch_one<FWD, 5, 1, 3>(a, stream); else attn_launch_one<FWD, 5, 2, 3>(a, stream);
  } else if (nb == 6) {
    if (ncb == 1) attn_launch_one<FWD, 8, 1, 3>(a, stream); else attn_launch_one<FWD, 8, 2, 3>(a, stream);
  } else {
    if (ncb == 1) attn_launch_one<FWD, 8, 1, 4>(a, stream); else attn_launch_one<FWD, 8, 2, 4>(a, stream);
  }
}

#define S ((hipStream_t)stream)
#define CLEAR() (void)hipGetLastError()
#define DONE() return (int)hipGetLastError()

}  // namespace

extern "C" {

int osrl_cdt_embed_ln(const float* states, const float* actions, const float* returns, const float* costs_to_go,
                      const float* episode_cost, const int64_t* time_steps, const float* Ws, const float* bs,
                      const float* Wa, const float* ba, const float* Wc, const float* bc, const float* Wr,
                      const float* br, const float* Wp, const float* bp, const float* timestep_emb, const float* ln_g,
                      const float* ln_b, int32_t B, int32_t T, int32_t od, int32_t ad, int32_t E,
                      int32_t cost_transform, int32_t use_rew, int32_t use_cost, int32_t prefix, float* seq, float* x0,
                      float* stats, float* ctg_t, void* stream) {
  if (!states || !actions || !time_steps || !seq || !x0 || !stats || B < 1 || T < 1 || E < 1 || E > 64 * kMaxEPL)
    return -1;
  if ((use_rew && (!returns || !Wr || !br)) || (use_cost && (!costs_to_go || !Wc || !bc || !ctg_t)) ||
      (prefix && (!episode_cost || !Wp || !bp)))
    return -1;
  const int R = 2 + (use_rew ? 1 : 0) + (use_cost ? 1 : 0);
  EmbedArgs a{states, actions, returns, costs_to_go, episode_cost, time_steps, Ws, bs, Wa, ba, Wc, bc, Wr, br, Wp, bp,
              timestep_emb, ln_g, ln_b, seq, x0, stats, ctg_t, B, T, od, ad, E, cost_transform, R, prefix ? 1 : 0,
              use_rew ? 1 : 0, use_cost ? 1 : 0};
  const int rows = B * (R * T + (prefix ? 1 : 0));
  CLEAR();
  {  // E = 256 / 512, aligned tensors, the transposed state / action weights within 48 KB of LDS: the 16-byte form
    const size_t wt_bytes = sizeof(float) * (size_t)(od + ad) * E;
    const uintptr_t al = reinterpret_cast<uintptr_t>(seq) | reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(bs) |
                         reinterpret_cast<uintptr_t>(ba) | reinterpret_cast<uintptr_t>(Wc) | reinterpret_cast<uintptr_t>(bc) |
                         reinterpret_cast<uintptr_t>(Wr) | reinterpret_cast<uintptr_t>(br) | reinterpret_cast<uintptr_t>(Wp) |
                         reinterpret_cast<uintptr_t>(bp) | reinterpret_cast<uintptr_t>(timestep_emb) |
                         reinterpret_cast<uintptr_t>(ln_g) | reinterpret_cast<uintptr_t>(ln_b);
    if ((E == 256 || E == 512) && wt_bytes <= 48 * 1024 && (al & 15) == 0 && rows >= 4096) {
      int n_wg = (rows + 3) / 4;
      n_wg = n_wg > 2048 ? 2048 : n_wg;
      if (E == 256)
        hipLaunchKernelGGL(embed_ln_v4_kernel<1>, dim3(n_wg), dim3(256), wt_bytes, S, a, n_wg);
      else
        hipLaunchKernelGGL(embed_ln_v4_kernel<2>, dim3(n_wg), dim3(256), wt_bytes, S, a, n_wg);
      DONE();
    }
  }
  hipLaunchKernelGGL(embed_ln_kernel, dim3((rows + 3) / 4), dim3(256), 0, S, a);
  DONE();
}

// the 16-byte-per-lane LayerNorm kernels: E = 256 or 512, every tensor 16-byte aligned (rows are then aligned too)
static bool ln_v4_ok(int E, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f) {
  if (E != 256 && E != 512) return false;
  const uintptr_t m = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                      reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(e) | reinterpret_cast<uintptr_t>(f);
  return (m & 15) == 0;
}

int osrl_layernorm_fwd(const float* x, const float* delta, const float* gamma, const float* beta, float* xout,
                       float* y, float* stats, int32_t M, int32_t E, void* stream) {
  if (!x || !gamma || !beta || !y || !stats || M < 1 || E < 1 || E > 64 * kMaxEPL) return -1;
  CLEAR();
  if (ln_v4_ok(E, x, delta, gamma, beta, xout, y)) {
    if (E == 256)
      hipLaunchKernelGGL((ln_fwd_v4_kernel<false, 1>), dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y,
                         stats, M, DropSite{});
    else
      hipLaunchKernelGGL((ln_fwd_v4_kernel<false, 2>), dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y,
                         stats, M, DropSite{});
    DONE();
  }
  hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y, stats, M,
                     E, DropSite{});
  DONE();
}

static bool drop_site(const osrl_dropout_t* dr, DropSite* d) {
  if (!dr || !(dr->p > 0.f) || !(dr->p < 1.0f)) return false;
  *d = DropSite{osrl_rng::drop_thresh(dr->p), dr->site, (uint32_t)dr->seed, (uint32_t)(dr->seed >> 32),
                1.0f / (1.0f - dr->p), dr->st};
  return true;
}

int osrl_layernorm_fwd_drop(const float* x, const float* delta, const osrl_dropout_t* drop, const float* gamma,
                            const float* beta, float* xout, float* y, float* stats, int32_t M, int32_t E,
                            void* stream) {
  if (!x || !delta || !gamma || !beta || !y || !stats || M < 1 || E < 1 || E > 64 * kMaxEPL) return -1;
  DropSite d{};
  // p = 0 (or no descriptor): the plain call.  p >= 1 would have to zero the branch (torch does) while the backward's
  // twin rejects it: both reject it -- no reference config drops everything, the engine validates p < 1
  if (drop && !(drop->p < 1.0f)) return -1;
  if (!drop_site(drop, &d)) return osrl_layernorm_fwd(x, delta, gamma, beta, xout, y, stats, M, E, stream);
  CLEAR();
  if (ln_v4_ok(E, x, delta, gamma, beta, xout, y)) {
    if (E == 256)
      hipLaunchKernelGGL((ln_fwd_v4_kernel<true, 1>), dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y,
                         stats, M, d);
    else
      hipLaunchKernelGGL((ln_fwd_v4_kernel<true, 2>), dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y,
                         stats, M, d);
    DONE();
  }
  hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, S, x, delta, gamma, beta, xout, y, stats, M, E,
                     d);
  DONE();
}

int osrl_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* dres,
                       float* dx, float* partial_ws, int32_t n_parts, int32_t M, int32_t E, float* slab,
                       int64_t g_off, int64_t b_off, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !partial_ws || n_parts < 1 || M < 1 || E > 64 * kMaxEPL)
    return -1;
  CLEAR();
  if (ln_v4_ok(E, dy, x, gamma, dres, dx, dx)) {
    if (E == 256)
      hipLaunchKernelGGL((ln_bwd_v4_kernel<false, 1>), dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx,
                         partial_ws, M, nullptr, DropSite{}, n_parts);
    else
      hipLaunchKernelGGL((ln_bwd_v4_kernel<false, 2>), dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx,
                         partial_ws, M, nullptr, DropSite{}, n_parts);
  } else
  hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx, partial_ws, M,
                     E, nullptr, DropSite{});
  if (slab)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * E + 63) / 64), dim3(1024), 0, S, partial_ws, n_parts, E, slab,
                       g_off, b_off);
  DONE();
}

int osrl_layernorm_param_reduce(const float* partial_ws, int64_t ws_stride, int32_t n_sites, int32_t n_parts, int32_t E,
                                float* slab, const int64_t* g_offs, const int64_t* b_offs, void* stream) {
  if (!partial_ws || !slab || !g_offs || !b_offs || n_sites < 1 || n_sites > 16 || n_parts < 1 || E < 1 ||
      ws_stride < (int64_t)n_parts * 2 * E)
    return -1;
  LnSites sites{};
  for (int k = 0; k < n_sites; ++k) {
    sites.g_off[k] = g_offs[k];
    sites.b_off[k] = b_offs[k];
  }
  CLEAR();
  hipLaunchKernelGGL(ln_param_reduce_many_kernel, dim3((2 * E + 63) / 64, n_sites), dim3(1024), 0, S, partial_ws, ws_stride,
                     n_parts, E, slab, sites);
  DONE();
}

int osrl_layernorm_bwd_drop(const float* dy, const float* x, const float* stats, const float* gamma, const float* dres,
                            float* dx, float* dx_dropped, const osrl_dropout_t* drop, float* partial_ws, int32_t n_parts,
                            int32_t M, int32_t E, float* slab, int64_t g_off, int64_t b_off, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dx_dropped || !partial_ws || n_parts < 1 || M < 1 ||
      E > 64 * kMaxEPL)
    return -1;
  DropSite d{};
  if (!drop_site(drop, &d)) return -1;
  CLEAR();
  if (ln_v4_ok(E, dy, x, gamma, dres, dx, dx_dropped)) {
    if (E == 256)
      hipLaunchKernelGGL((ln_bwd_v4_kernel<true, 1>), dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx,
                         partial_ws, M, dx_dropped, d, n_parts);
    else
      hipLaunchKernelGGL((ln_bwd_v4_kernel<true, 2>), dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx,
                         partial_ws, M, dx_dropped, d, n_parts);
  } else
  hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(n_parts), dim3(256), 0, S, dy, x, stats, gamma, dres, dx, partial_ws, M, E,
                     dx_dropped, d);
  if (slab)
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((2 * E + 63) / 64), dim3(1024), 0, S, partial_ws, n_parts, E, slab,
                       g_off, b_off);
  DONE();
}

constexpr size_t kMaxLds = 160 * 1024;  // LDS per workgroup on gfx950
static size_t attn_lds(int S_, int d, bool bwd) {
  const size_t Sp = (S_ + 15) & ~15, dp = (d + 15) & ~15, ldq = dp + 8, ldp = Sp + 8;
  if (bwd)  // two [S, d] operand tiles, per-row (max, 1/sum, r), key mask, keep-flag bytes of the [S, S] tile
    return sizeof(float) * (2 * Sp * ldq + 4 * Sp) + Sp * Sp;
  const size_t fl = Sp * ldq + dp * ldp + Sp;  // forward: K, V^T, key mask (scores stay in registers)
  return sizeof(float) * fl;
}

static bool attn_drop_args(const osrl_dropout_t* dr, AttnArgs* a) {
  a->drop_thresh = 0;
  a->drop_site = a->k0 = a->k1 = 0;
  a->drop_scale = 1.0f;
  a->st = nullptr;
  if (!dr || dr->p <= 0.f) return true;
  if (!(dr->p < 1.0f)) return false;
  a->drop_thresh = osrl_rng::drop_thresh(dr->p);
  a->drop_scale = 1.0f / (1.0f - dr->p);
  a->drop_site = dr->site;
  a->k0 = (uint32_t)dr->seed;
  a->k1 = (uint32_t)(dr->seed >> 32);
  a->st = dr->st;
  return true;
}

int osrl_dropout(const float* x, float* y, int64_t n, const osrl_dropout_t* dr, void* stream) {
  if (!x || !y || n < 1 || !dr || !(dr->p > 0.f) || !(dr->p < 1.0f)) return -1;
  int64_t blocks = ((n + 3) / 4 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  CLEAR();
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, S, x, y, n, osrl_rng::drop_thresh(dr->p),
                     1.0f / (1.0f - dr->p), dr->site, (uint32_t)dr->seed, (uint32_t)(dr->seed >> 32), dr->st);
  DONE();
}

int osrl_attention_fwd(const float* qkv, const float* mask, int32_t B, int32_t S_, int32_t E, int32_t H, int32_t rep,
                       int32_t prefix, const osrl_dropout_t* drop, float* o, void* stream) {
  return osrl_attention_fwd_keep(qkv, mask, B, S_, E, H, rep, prefix, drop, o, nullptr, stream);
}

int64_t osrl_attention_keep_bytes(int32_t B, int32_t S_, int32_t E, int32_t H) {
  if (B < 1 || S_ < 1 || S_ > 128 || H < 1 || E % H || !(E / H == 16 || E / H == 32)) return 0;
  const int64_t Sp = (S_ + 15) & ~15;
  return (int64_t)B * H * (Sp >> 4) * Sp * 4;
}

int osrl_attention_fwd_keep(const float* qkv, const float* mask, int32_t B, int32_t S_, int32_t E, int32_t H, int32_t rep,
                            int32_t prefix, const osrl_dropout_t* drop, float* o, unsigned char* keep, void* stream) {
  prefix = prefix ? 1 : 0;
  if (!qkv || !mask || !o || B < 1 || S_ < 1 || S_ > 128 || E % H || E / H > 64 || rep < 1 || (S_ - prefix) % rep ||
      S_ - prefix < rep)
    return -1;
  AttnArgs a{qkv, mask, o, nullptr, nullptr, B, S_, E, H, rep, prefix};
  if (!attn_drop_args(drop, &a)) return -1;
  a.keep = a.drop_thresh ? keep : nullptr;
  if (attn_vec_ok(E, H, qkv, o, nullptr, nullptr)) {
    CLEAR();
    attn_launch_v<true>(a, S);
    DONE();
  }
  if (keep) return -1;  // the keep hand-off exists for the head-width 16 / 32 kernels only (osrl_attention_keep_bytes == 0)
  const size_t lds = attn_lds(S_, E / H, false);
  if (lds > kMaxLds) return -1;
  CLEAR();
  if (S_ <= 80)
    hipLaunchKernelGGL(attn_fwd_kernel<5>, dim3(B * H), dim3(256), lds, S, a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<8>, dim3(B * H), dim3(256), lds, S, a);
  DONE();
}

int osrl_attention_bwd(const float* qkv, const float* mask, const float* dout, int32_t B, int32_t S_, int32_t E,
                       int32_t H, int32_t rep, int32_t prefix, const osrl_dropout_t* drop, float* dqkv, void* stream) {
  return osrl_attention_bwd_keep(qkv, mask, dout, B, S_, E, H, rep, prefix, drop, dqkv, nullptr, stream);
}

int osrl_attention_bwd_keep(const float* qkv, const float* mask, const float* dout, int32_t B, int32_t S_, int32_t E,
                            int32_t H, int32_t rep, int32_t prefix, const osrl_dropout_t* drop, float* dqkv,
                            const unsigned char* keep, void* stream) {
  prefix = prefix ? 1 : 0;
  if (!qkv || !mask || !dout || !dqkv || B < 1 || S_ < 1 || S_ > 128 || E % H || E / H > 64 || rep < 1 ||
      (S_ - prefix) % rep || S_ - prefix < rep)
    return -1;
  AttnArgs a{qkv, mask, nullptr, dout, dqkv, B, S_, E, H, rep, prefix};
  if (!attn_drop_args(drop, &a)) return -1;
  a.keep = a.drop_thresh ? const_cast<unsigned char*>(keep) : nullptr;
  if (attn_vec_ok(E, H, qkv, nullptr, dout, dqkv)) {
    CLEAR();
    attn_launch_v<false>(a, S);
    DONE();
  }
  if (keep) return -1;
  const size_t lds = attn_lds(S_, E / H, true);
  if (lds > kMaxLds) return -1;
  if (lds > 64 * 1024) {  // opt in to a large dynamic allocation (S = 128 with head_dim 64)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  CLEAR();
  if (S_ <= 80)
    hipLaunchKernelGGL(attn_bwd_kernel<5>, dim3(B * H), dim3(256), lds, S, a);
  else
    hipLaunchKernelGGL(attn_bwd_kernel<8>, dim3(B * H), dim3(256), lds, S, a);
  DONE();
}

int osrl_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n < 4 || (n & 3)) return -1;
  int64_t blocks = (n / 4 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  CLEAR();
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((int)blocks), dim3(256), 0, S, x, y, n / 4);
  DONE();
}

int osrl_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  if (!dy || !x || !dx || n < 4 || (n & 3)) return -1;
  int64_t blocks = (n / 4 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  CLEAR();
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((int)blocks), dim3(256), 0, S, dy, x, dx, n / 4);
  DONE();
}

int osrl_cdt_loss(const float* head, const float* logits, const float* state_pred, const float* actions,
                  const float* states, const float* mask, const float* costs, int32_t B, int32_t T, int32_t od,
                  int32_t ad, int32_t stochastic, int32_t no_entropy, const float* log_temperature, float cost_w,
                  float state_w, float lr, int32_t warmup, const osrl_step_state_t* st, const float* counts,
                  int32_t world, float* dhead, float* dlogits, float* dsp, float* stat, float* ent_out, float* ws,
                  void* stream) {
  if (!head || !logits || !state_pred || !actions || !states || !mask || !costs || !st || !dhead || !dlogits || !dsp ||
      !stat || B < 1 || T < 1)
    return -1;
  LossArgs a{head, logits, state_pred, actions, states, mask, costs, dhead, dlogits, dsp, stat, ent_out,
             log_temperature, st, B, T, od, ad, stochastic, no_entropy, warmup, cost_w, state_w, lr, counts,
             world > 0 ? world : 1, 1.0f / (float)(world > 0 ? world : 1)};
  CLEAR();
  const int nblk = (B * T + 1023) / 1024;
  if (ws && counts && nblk > 1) {  // big batch: one 1024-token slice per workgroup, then the ordered sum of the partials
    hipLaunchKernelGGL(cdt_loss_kernel, dim3(nblk), dim3(1024), 0, S, a, ws);
    hipLaunchKernelGGL(cdt_loss_finish_kernel, dim3(1), dim3(64), 0, S, a, ws, nblk);
  } else {
    hipLaunchKernelGGL(cdt_loss_kernel, dim3(1), dim3(1024), 0, S, a, (float*)nullptr);
  }
  DONE();
}

int osrl_cdt_mask_counts(const float* mask, int32_t BT, float* out, void* stream) {
  if (!mask || !out || BT < 1) return -1;
  CLEAR();
  hipLaunchKernelGGL(mask_counts_kernel, dim3(1), dim3(1024), 0, S, mask, BT, out);
  DONE();
}

int osrl_cdt_timestep_scatter(const float* dseq, const int64_t* time_steps, int32_t B, int32_t T, int32_t R,
                              int32_t prefix, int32_t E, float* dte, void* stream) {
  if (!dseq || !time_steps || !dte || B < 1 || T < 1 || E < 1 || R < 2 || R > 4) return -1;
  const int64_t n = (int64_t)B * T * E;
  CLEAR();
  hipLaunchKernelGGL(te_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S, dseq, time_steps, B * T, E,
                     T, R, prefix ? 1 : 0, dte);
  DONE();
}

int osrl_clip_grad_scale(const float* grad, int64_t n, float clip, float* partial_ws, int32_t n_parts, float* out,
                         void* stream) {
  if (!grad || !partial_ws || !out || n < 4 || (n & 3) || n_parts < 1 || n_parts > 4096) return -1;
  CLEAR();
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(n_parts), dim3(256), 0, S, grad, n / 4, partial_ws);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(1), dim3(64), 0, S, partial_ws, n_parts, clip, out);
  DONE();
}

int osrl_cdt_temperature_step(float* log_temperature, float* moments, const float* entropy, float target_entropy,
                              float lr, float beta1, float beta2, float eps, const osrl_step_state_t* st, void* stream) {
  if (!log_temperature || !moments || !entropy || !st) return -1;
  CLEAR();
  hipLaunchKernelGGL(temperature_step_kernel, dim3(1), dim3(64), 0, S, log_temperature, moments, entropy,
                     target_entropy, lr, beta1, beta2, eps, st);
  DONE();
}

}  // extern "C"
