// rng.hip -- on-device Gaussian noise and the on-device replay sampler for gfx950.
//
// * osrl_randn_fill replaces the host-side torch samplers of the reference's train step
//   (randn_like net.py:327, Normal.rsample net.py:187, Normal.sample cpq.py:166, torch.randn
//   net.py:334 -- the latter is even drawn on the CPU and copied to the device every call).
//   Philox4x32-10 (counter = {element/4, step, stream_id, 0}, key = seed) + Box-Muller; the step
//   comes from the device-resident osrl_step_state_t so a captured hipGraph draws fresh noise at
//   every replay.  Parity tests inject explicit noise tensors instead.
// * osrl_replay_gather replaces TransitionDataset.__iter__/__prepare_sample + DataLoader + H2D
//   (osrl/common/dataset.py:832-847, examples/train/train_cpq.py:122-142): uniform-with-replacement
//   row indices drawn on device, then a row gather of the resident transition tables.  HBM-bound:
//   one wave reads one row of each table with coalesced dword loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "philox.h"
#include "gather.h"
#include "step.h"
#include "argmem.h"
#include "trace.h"

namespace {

using namespace osrl_rng;

// NT: the workgroup size when it is a compile-time constant (0: read blockDim -- an s_load from the hidden kernarg block)
template <int NT = 0>
__device__ __forceinline__ void randn_body(float* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1,
                                           uint32_t stream_id, uint32_t step, int64_t block, int64_t n_blocks) {
  const int64_t n4 = (n + 3) >> 2;
  constexpr bool kConst = NT != 0;
  const int64_t nt = kConst ? NT : (int64_t)blockDim.x;
  for (int64_t i = block * nt + threadIdx.x; i < n4; i += n_blocks * nt) {
    const U4 r = philox4x32_10(U4{(uint32_t)i, (uint32_t)(i >> 32), step, stream_id}, k0, k1);
    const float r0 = sqrtf(-2.0f * __logf(u01(r.x))), r1 = sqrtf(-2.0f * __logf(u01(r.z)));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u01(r.y), &s0, &c0);
    __sincosf(6.283185307179586f * u01(r.w), &s1, &c1);
    const float z[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
    const int64_t base = i * 4;
    if (base + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15) == 0)) {
      *reinterpret_cast<float4*>(out + base) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
      for (int k = 0; k < 4; ++k)
        if (base + k < n) out[base + k] = z[k];
    }
  }
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1,
                                                    uint32_t stream_id, const osrl_step_state_t* __restrict__ st) {
  randn_body(out, n, k0, k1, stream_id, st ? (uint32_t)st->step : 0u, blockIdx.x, gridDim.x);
}

using osrl_gather::GatherArgs;
using osrl_gather::gather_body;

__global__ __launch_bounds__(256) void gather_kernel(const GatherArgs a) {
  gather_body<const GatherArgs&>(a, a.st ? (uint32_t)a.st->step : 0u, blockIdx.x);
}

// 1024-thread workgroups: every workgroup signs in with one atomic on ONE address (those serialise at ~20 ns each:
// 600 four-row workgroups made this kernel 17 us long), so few, fat workgroups
constexpr int kBeginThreads = 1024;
// The step prologue in one launch (osrl_step_begin): workgroups [0, g_blocks) gather, [g_blocks, g_blocks + r_blocks)
// fill the noise, every one of them with step = t_old + 1; the LAST workgroup to have read t_old ticks the state.
struct BeginArgs {
  osrl_step_state_t* st;
  float beta1, beta2;
  int32_t warmup, n_stats, ring_len;
  const float* stats_cur;
  float* ring;
  float* noise;
  int64_t noise_n;
  uint32_t nk0, nk1, noise_stream;
  int32_t g_blocks, r_blocks;
  // osrl_step_begin_peer: a SECOND step state that takes turns with `st` (software-pipelined steps, engine/pipeline.py):
  // the new step count is max(st->step, peer->step) + 1, `st` receives it; nullptr = `st` counts alone
  const osrl_step_state_t* peer;
};

template <class BR, class AR>
__device__ __forceinline__ void step_begin_body(BR b, AR a) {
  __shared__ int64_t s_t, s_own;
  __shared__ int s_last;
  __shared__ float s_tick[3];
  if (threadIdx.x == 0) {
    const int64_t own = __atomic_load_n(&b.st->step, __ATOMIC_RELAXED);
    const int64_t other = b.peer ? __atomic_load_n(&b.peer->step, __ATOMIC_RELAXED) : own;
    s_own = own;  // the step whose statistics `stats_cur` holds (this state's previous turn)
    s_t = other > own ? other : own;
  }
  __syncthreads();
  const int64_t t_old = s_t;
  // The tick's bias corrections (osrl_step::tick_values: two double-precision pow, ~2 us on one lane) used to run in the
  // last workgroup AFTER everything else -- serial time at the head of every step.  Every workgroup now computes them
  // up front on three lanes of its last wave (lanes 0 / 1 take beta1 / beta2 in lockstep), beside its gather / noise work;
  // whoever turns out to be last only stores them.  Same expressions, same bits.
  if (threadIdx.x >= kBeginThreads - 64 && threadIdx.x < kBeginThreads - 61) {
    const int l = threadIdx.x - (kBeginThreads - 64);
    const double t = (double)(t_old + 1);
    const double pw = pow((double)(l == 0 ? b.beta1 : b.beta2), t);
    const float lrs = b.warmup > 0 ? (float)fmin(t / (double)b.warmup, 1.0) : 1.0f;
    s_tick[l] = l == 0 ? (float)(1.0 - pw) : l == 1 ? (float)sqrt(1.0 - pw) : lrs;
  }
  if (threadIdx.x == 0) {
    // the old step has been READ by this workgroup (its value went through LDS): count the arrival
    __threadfence();
    const uint32_t seen = atomicAdd(&b.st->arrive_, 1u);
    const int nb = b.g_blocks + b.r_blocks;  // the launch's grid (host: max(g_blocks + r_blocks, 1))
    s_last = seen == (unsigned)(nb > 0 ? nb : 1) - 1;
  }
  const uint32_t step = (uint32_t)(t_old + 1);
  const int blk = blockIdx.x;
  if (blk < b.g_blocks) {
    gather_body<AR, kBeginThreads>(a, step, blk);
  } else if (b.noise) {
    randn_body<kBeginThreads>(b.noise, b.noise_n, b.nk0, b.nk1, b.noise_stream, step, blk - b.g_blocks, b.r_blocks);
  }
  __syncthreads();
  if (s_last) {  // every workgroup holds t_old in registers by now: the state may move
    osrl_step::commit_stats<kBeginThreads>(s_own, b.stats_cur, b.ring, b.n_stats, b.ring_len);
    if (threadIdx.x == 0) {
      b.st->step = t_old + 1;
      b.st->bc1 = s_tick[0];
      b.st->bc2_sqrt = s_tick[1];
      b.st->lr_scale = s_tick[2];
      b.st->arrive_ = 0;
    }
  }
}
__global__ __launch_bounds__(kBeginThreads) void step_begin_kernel(const BeginArgs b, const GatherArgs a) {
  step_begin_body<const BeginArgs&, const GatherArgs&>(b, a);
}
struct BeginPack {  // both descriptors as one device-resident block (argmem.h): the first launch of every step
  BeginArgs b;
  GatherArgs a;
};
__global__ __launch_bounds__(kBeginThreads) void step_begin_kernel_p(const void* p) {
  OSRL_TRACE_BEGIN(1, p);
  const OSRL_CAS BeginPack& k = *(const OSRL_CAS BeginPack*)p;
  step_begin_body<const OSRL_CAS BeginArgs&, const OSRL_CAS GatherArgs&>(k.b, k.a);
}

struct SeqArgs {
  const float *obs, *act, *ret, *cret, *cost;  // concatenated trajectories [Ntot, .]
  const int64_t* traj_start;
  const int32_t* traj_len;
  const float* cdf;  // inclusive cumulative trajectory-sampling probabilities, or NULL = uniform
  const float* start_cdf;  // [Ntot] inclusive cumulative start-index probabilities inside each trajectory, or NULL
  const int32_t* idx_in;   // optional [B,2] = (trajectory, start) given by the caller instead of drawn
  float *states, *actions, *returns, *cost_returns, *mask, *episode_cost, *costs;
  int64_t* time_steps;
  int32_t* idx_out;  // optional [B,2] = (trajectory, start)
  int32_t n_traj, B, T, od, ad;
  float reward_scale, cost_scale;
  uint32_t k0, k1, stream_id;
  const osrl_step_state_t* st;
};

// SequenceDataset.__iter__/__prepare_sample (dataset.py:749-787): trajectory ~ sample_prob, start ~ U{0..len-1},
// window [start, start+T) clipped to the trajectory, zero tail padding, mask, time_steps = start + arange(T).
__global__ __launch_bounds__(256) void seq_window_kernel(const SeqArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const uint32_t step = a.st ? (uint32_t)a.st->step : 0u;
  const U4 r = philox4x32_10(U4{(uint32_t)b, 0x5e9u, step, a.stream_id}, a.k0, a.k1);
  int traj;
  if (a.idx_in) {
    traj = a.idx_in[2 * b];
  } else if (a.cdf) {  // inverse-CDF draw: first index with cdf[i] > u
    const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
    int lo = 0, hi = a.n_traj - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (a.cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    traj = lo;
  } else {
    traj = (int)__umul64hi(((uint64_t)r.x << 32) | r.y, (uint64_t)a.n_traj);
  }
  const int len = a.traj_len[traj];
  const int64_t base = a.traj_start[traj];
  int start;
  if (a.idx_in) {
    start = a.idx_in[2 * b + 1];
    start = start < 0 ? 0 : start >= len ? len - 1 : start;
  } else if (a.start_cdf) {  // start_sampling (dataset.py:781-783): start ~ start_idx_sample_prob[traj]
    const float u = (float)(r.z >> 8) * (1.0f / 16777216.0f);
    const float* __restrict__ c = a.start_cdf + base;
    int lo = 0, hi = len - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (c[mid] > u) hi = mid; else lo = mid + 1;
    }
    start = lo;
  } else {
    start = (int)__umul64hi(((uint64_t)r.z << 32) | r.w, (uint64_t)len);
  }
  if (lane == 0) {
    a.episode_cost[b] = a.cret[base] * a.cost_scale;
    if (a.idx_out) {
      a.idx_out[2 * b] = traj;
      a.idx_out[2 * b + 1] = start;
    }
  }
  const int T = a.T;
  for (int t = lane; t < T; t += 64) {
    const bool ok = start + t < len;
    const int64_t src = base + start + t;
    const size_t o = (size_t)b * T + t;
    a.returns[o] = ok ? a.ret[src] * a.reward_scale : 0.f;
    a.cost_returns[o] = ok ? a.cret[src] * a.cost_scale : 0.f;
    a.costs[o] = ok ? a.cost[src] : 0.f;
    a.mask[o] = ok ? 1.f : 0.f;
    a.time_steps[o] = start + t;
  }
  for (int i = lane; i < T * a.od; i += 64) {
    const int t = i / a.od, c = i - t * a.od;
    a.states[(size_t)b * T * a.od + i] = (start + t < len) ? a.obs[(size_t)(base + start + t) * a.od + c] : 0.f;
  }
  for (int i = lane; i < T * a.ad; i += 64) {
    const int t = i / a.ad, c = i - t * a.ad;
    a.actions[(size_t)b * T * a.ad + i] = (start + t < len) ? a.act[(size_t)(base + start + t) * a.ad + c] : 0.f;
  }
}

}  // namespace

extern "C" int osrl_seq_window_gather(const float* obs, const float* act, const float* returns,
                                      const float* cost_returns, const float* costs, const int64_t* traj_start,
                                      const int32_t* traj_len, const float* cdf, const float* start_cdf,
                                      const int32_t* idx_in, int32_t n_traj, int32_t B, int32_t T,
                                      int32_t od, int32_t ad, float reward_scale, float cost_scale, float* o_states,
                                      float* o_actions, float* o_returns, float* o_cost_returns,
                                      int64_t* o_time_steps, float* o_mask, float* o_episode_cost, float* o_costs,
                                      int32_t* idx_out, uint64_t seed, uint32_t stream_id,
                                      const osrl_step_state_t* st, void* stream) {
  if (!obs || !act || !returns || !cost_returns || !costs || !traj_start || !traj_len || n_traj < 1 || B < 1 || T < 1 ||
      !o_states || !o_actions || !o_returns || !o_cost_returns || !o_time_steps || !o_mask || !o_episode_cost || !o_costs)
    return -1;
  SeqArgs a{obs, act, returns, cost_returns, costs, traj_start, traj_len, cdf, start_cdf, idx_in, o_states, o_actions, o_returns,
            o_cost_returns, o_mask, o_episode_cost, o_costs, o_time_steps, idx_out, n_traj, B, T, od, ad,
            reward_scale, cost_scale, (uint32_t)seed, (uint32_t)(seed >> 32), stream_id, st};
  (void)hipGetLastError();
  hipLaunchKernelGGL(seq_window_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_randn_fill(float* out, int64_t n, uint64_t seed, uint32_t stream_id,
                               const osrl_step_state_t* st, void* stream) {
  if (!out || n < 1) return -1;
  const int64_t n4 = (n + 3) / 4;
  int64_t blocks = (n4 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(randn_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, out, n, (uint32_t)seed,
                     (uint32_t)(seed >> 32), stream_id, st);
  return (int)hipGetLastError();
}

extern "C" int osrl_replay_gather(int32_t n_fields, const float* const* src, float* const* dst,
                                  const int32_t* width, const float* scale, int64_t n_rows, int32_t batch,
                                  int32_t* idx_out, uint64_t seed, uint32_t stream_id,
                                  const osrl_step_state_t* st, void* stream) {
  if (n_fields < 1 || n_fields > OSRL_MAX_FIELDS || !src || !dst || !width || n_rows < 1 || batch < 1) return -1;
  GatherArgs a;
  for (int f = 0; f < OSRL_MAX_FIELDS; ++f) {
    a.src[f] = f < n_fields ? src[f] : nullptr;
    a.dst[f] = f < n_fields ? dst[f] : nullptr;
    a.width[f] = f < n_fields ? width[f] : 0;
    a.scale[f] = (f < n_fields && scale) ? scale[f] : 1.0f;
    if (f < n_fields && (!a.src[f] || !a.dst[f] || a.width[f] < 1)) return -1;
  }
  a.n_fields = n_fields;
  a.batch = batch;
  a.n_rows = n_rows;
  a.idx_out = idx_out;
  a.k0 = (uint32_t)seed;
  a.k1 = (uint32_t)(seed >> 32);
  a.stream_id = stream_id;
  a.st = st;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(gather_kernel, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_step_begin(osrl_step_state_t* st, float beta1, float beta2, int32_t warmup, const float* stats_cur,
                               float* ring, int32_t n_stats, int32_t ring_len, float* noise, int64_t noise_n,
                               uint64_t noise_seed, uint32_t noise_stream, int32_t n_fields, const float* const* src,
                               float* const* dst, const int32_t* width, const float* scale, int64_t n_rows,
                               int32_t batch, uint64_t gather_seed, uint32_t gather_stream, void* stream) {
  return osrl_step_begin_peer(st, nullptr, beta1, beta2, warmup, stats_cur, ring, n_stats, ring_len, noise, noise_n,
                              noise_seed, noise_stream, n_fields, src, dst, width, scale, n_rows, batch, gather_seed,
                              gather_stream, stream);
}

extern "C" int osrl_step_begin_peer(osrl_step_state_t* st, const osrl_step_state_t* peer, float beta1, float beta2,
                                    int32_t warmup, const float* stats_cur, float* ring, int32_t n_stats,
                                    int32_t ring_len, float* noise, int64_t noise_n, uint64_t noise_seed,
                                    uint32_t noise_stream, int32_t n_fields, const float* const* src, float* const* dst,
                                    const int32_t* width, const float* scale, int64_t n_rows, int32_t batch,
                                    uint64_t gather_seed, uint32_t gather_stream, void* stream) {
  if (!st || peer == st || n_fields < 0 || n_fields > OSRL_MAX_FIELDS || (noise && noise_n < 1)) return -1;
  if (n_fields > 0 && (!src || !dst || !width || n_rows < 1 || batch < 1)) return -1;
  BeginPack k{};
  GatherArgs& a = k.a;
  BeginArgs& b = k.b;
  for (int f = 0; f < OSRL_MAX_FIELDS; ++f) {
    a.src[f] = f < n_fields ? src[f] : nullptr;
    a.dst[f] = f < n_fields ? dst[f] : nullptr;
    a.width[f] = f < n_fields ? width[f] : 0;
    a.scale[f] = (f < n_fields && scale) ? scale[f] : 1.0f;
    if (f < n_fields && (!a.src[f] || !a.dst[f] || a.width[f] < 1)) return -1;
  }
  a.n_fields = n_fields;
  a.batch = n_fields > 0 ? batch : 0;
  a.n_rows = n_rows;
  a.idx_out = nullptr;
  a.k0 = (uint32_t)gather_seed;
  a.k1 = (uint32_t)(gather_seed >> 32);
  a.stream_id = gather_stream;
  a.st = st;
  b.st = st;
  b.beta1 = beta1;
  b.beta2 = beta2;
  b.warmup = warmup;
  b.n_stats = n_stats;
  b.ring_len = ring_len > 0 ? ring_len : 1;
  b.stats_cur = stats_cur;
  b.ring = ring;
  b.noise = noise;
  b.noise_n = noise_n;
  b.nk0 = (uint32_t)noise_seed;
  b.nk1 = (uint32_t)(noise_seed >> 32);
  b.noise_stream = noise_stream;
  b.peer = peer;
  constexpr int kWaves = kBeginThreads / 64;
  b.g_blocks = n_fields > 0 ? (batch + kWaves - 1) / kWaves : 0;
  int64_t rb = noise ? ((noise_n + 3) / 4 + kBeginThreads - 1) / kBeginThreads : 0;
  b.r_blocks = (int32_t)(rb > 256 ? 256 : rb);
  const int grid = b.g_blocks + b.r_blocks > 0 ? b.g_blocks + b.r_blocks : 1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (const void* dev_args = osrl_argmem::slot(k))
    hipLaunchKernelGGL(step_begin_kernel_p, dim3(grid), dim3(kBeginThreads), 0, (hipStream_t)stream, dev_args);
  else
    hipLaunchKernelGGL(step_begin_kernel, dim3(grid), dim3(kBeginThreads), 0, (hipStream_t)stream, b, a);
  return (int)hipGetLastError();
}
