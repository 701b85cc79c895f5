// Vectorised step of the build-owned synthetic safe environment (osrl_amd/common/synthetic_env.py), used by the
// batched on-device evaluate() (the reference's rollout loops cpq.py:330-347, bcql.py:323-340,
// bc.py:125-145 cross host<->device once per env step of ONE episode).  One workgroup per episode (= one row of
// the policy's batch): s' = A s + Bm clip(a), reward = 1 - 0.1 |s' - goal|^2, cost = 1[s'.w > thr]; the episode
// accumulators (return, cost, length) live on device so a whole rollout needs one host sync at its end.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/osrl_amd.h"

namespace {

constexpr int kMaxDim = 256;

__global__ __launch_bounds__(256) void env_step_kernel(osrl_env_t e, const float* __restrict__ act,
                                                       float* __restrict__ state, float* __restrict__ obs,
                                                       int obs_ld, float* __restrict__ acc /*[E,4]*/) {
  __shared__ float s[kMaxDim];
  __shared__ float a[64];
  __shared__ float red[2][4];
  const int ep = blockIdx.x, t = threadIdx.x;
  const int od = e.state_dim, ad = e.action_dim;
  float* st = state + (size_t)ep * od;
  float* ac = acc + (size_t)ep * 4;
  const bool alive = ac[3] == 0.f;  // finished episodes keep their state and totals (uniform per workgroup)
  if (t < od) s[t] = st[t];
  if (t < ad) a[t] = fminf(fmaxf(act[(size_t)ep * ad + t], -e.max_action), e.max_action);
  __syncthreads();
  float sn = 0.f, d2 = 0.f, sw = 0.f;
  if (t < od) {
    // At / Bt are stored transposed: lane t reads column t of each row -> coalesced, L2-resident for all episodes
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int j = 0;
    for (; j + 4 <= od; j += 4) {
      p0 = fmaf(e.At[(size_t)(j + 0) * od + t], s[j + 0], p0);
      p1 = fmaf(e.At[(size_t)(j + 1) * od + t], s[j + 1], p1);
      p2 = fmaf(e.At[(size_t)(j + 2) * od + t], s[j + 2], p2);
      p3 = fmaf(e.At[(size_t)(j + 3) * od + t], s[j + 3], p3);
    }
    for (; j < od; ++j) p0 = fmaf(e.At[(size_t)j * od + t], s[j], p0);
    float sb = 0.f;
    for (int k = 0; k < ad; ++k) sb = fmaf(e.Bt[(size_t)k * od + t], a[k], sb);
    sn = ((p0 + p1) + (p2 + p3)) + sb;
    const float d = sn - e.goal[t];
    d2 = d * d;
    sw = sn * e.w[t];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    d2 += __shfl_xor(d2, o);
    sw += __shfl_xor(sw, o);
  }
  if ((t & 63) == 0) {
    red[0][t >> 6] = d2;
    red[1][t >> 6] = sw;
  }
  __syncthreads();
  if (!alive) return;
  if (t < od) {
    st[t] = sn;
    obs[(size_t)ep * obs_ld + t] = sn;
  }
  if (t == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    float D = 0.f, W = 0.f;
    for (int i = 0; i < nw; ++i) {
      D += red[0][i];
      W += red[1][i];
    }
    const float len = ac[2] + 1.f;
    ac[0] += 1.f - 0.1f * D;
    ac[1] += (W > e.cost_threshold ? 1.f : 0.f) * e.cost_scale;
    ac[2] = len;
    if (len >= (float)e.episode_len) ac[3] = 1.f;
  }
}

}  // namespace

extern "C" int osrl_env_step(const osrl_env_t* env, const float* act, float* state, float* obs, int32_t obs_ld,
                             float* acc, int32_t episodes, void* stream) {
  if (!env || !act || !state || !obs || !acc || episodes < 1) return -1;
  if (env->state_dim < 1 || env->state_dim > kMaxDim || env->action_dim < 1 || env->action_dim > 64 ||
      obs_ld < env->state_dim || !env->At || !env->Bt || !env->w || !env->goal)
    return -1;
  (void)hipGetLastError();
  const int threads = ((env->state_dim + 63) / 64) * 64;
  hipLaunchKernelGGL(env_step_kernel, dim3(episodes), dim3(threads), 0, (hipStream_t)stream, *env, act, state, obs,
                     obs_ld, acc);
  return (int)hipGetLastError();
}
