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
                                                       int obs_ld, float* __restrict__ acc /*[E,4]*/,
                                                       float* __restrict__ step_out /*[E,2] or null*/) {
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
    const float rew = 1.f - 0.1f * D, cost = W > e.cost_threshold ? 1.f : 0.f;
    ac[0] += rew;
    ac[1] += cost * e.cost_scale;
    if (step_out) {
      step_out[(size_t)ep * 2] = rew;
      step_out[(size_t)ep * 2 + 1] = cost;
    }
    ac[2] = len;
    if (len >= (float)e.episode_len) ac[3] = 1.f;
  }
}


// ---- CDT windowed autoregression (CDTTrainer.rollout, cdt.py:436-518) with E episodes as the batch rows ----
// The reference keeps the whole history [episode_len+1] and slices the last seq_len steps every env step; here the
// CDT engine's [E, T] batch buffers ARE the window: it grows left-aligned (mask = 1 on the filled prefix) until it
// holds T steps, then slides by one per env step.  `cursor` = number of env steps taken so far (device int32, the
// same for every episode: all start together).

// act[e] = clamp(mean action predicted at the window's last filled position, +-max_action)  (cdt.py:489-493)
__global__ void cdt_pick_kernel(const float* __restrict__ head, int nh, int ad, int E, int T,
                                const int* __restrict__ cursor, float max_action, float* __restrict__ act) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E * ad) return;
  const int e = i / ad, k = i - e * ad;
  const int n = min(*cursor + 1, T);
  const float v = head[((size_t)e * T + (n - 1)) * nh + k];
  act[i] = fminf(fmaxf(v, -max_action), max_action);
}

struct CdtWin {
  float* states;       // [E,T,od]
  float* actions;      // [E,T,ad]
  float* returns;      // [E,T]
  float* ctg;          // [E,T]
  int64_t* time_steps; // [E,T]
  float* mask;         // [E,T]
  int E, T, od, ad;
};

// after the env step: store the action taken, append (s', R - r, C - c, t+1) -- sliding the window when it is full
__global__ __launch_bounds__(256) void cdt_push_kernel(CdtWin w, const float* __restrict__ act,
                                                       const float* __restrict__ obs, int obs_ld,
                                                       const float* __restrict__ step_out, float cost_scale,
                                                       int cost_reverse, const int* __restrict__ cursor,
                                                       int episode_len) {
  extern __shared__ float buf[];  // (T-1)*(od+ad) staged rows for the in-place slide
  const int e = blockIdx.x, t = threadIdx.x, T = w.T, od = w.od, ad = w.ad;
  const int step = *cursor;
  if (step >= episode_len) return;  // replayed past the episode end (graph chunks): time steps stay inside the table
  const int n = min(step + 1, T);
  float* S = w.states + (size_t)e * T * od;
  float* A = w.actions + (size_t)e * T * ad;
  float* R = w.returns + (size_t)e * T;
  float* C = w.ctg + (size_t)e * T;
  int64_t* TS = w.time_steps + (size_t)e * T;
  float* M = w.mask + (size_t)e * T;
  const float rew = step_out[(size_t)e * 2];
  const float craw = step_out[(size_t)e * 2 + 1];
  const float cost = (cost_reverse ? 1.f - craw : craw) * cost_scale;
  for (int k = t; k < ad; k += blockDim.x) A[(size_t)(n - 1) * ad + k] = act[(size_t)e * ad + k];
  __syncthreads();
  int pos = n;  // where the new step goes
  float r_prev = R[n - 1], c_prev = C[n - 1];
  if (n == T) {  // full window: slide rows 1..T-1 to 0..T-2 (staged through LDS: source and target overlap)
    float* sb = buf;
    float* ab = buf + (size_t)(T - 1) * od;
    for (int i = t; i < (T - 1) * od; i += blockDim.x) sb[i] = S[od + i];
    for (int i = t; i < (T - 1) * ad; i += blockDim.x) ab[i] = A[ad + i];
    float rr = 0.f, cc = 0.f;
    int64_t ts = 0;
    if (t < T - 1) {
      rr = R[t + 1];
      cc = C[t + 1];
      ts = TS[t + 1];
    }
    __syncthreads();
    for (int i = t; i < (T - 1) * od; i += blockDim.x) S[i] = sb[i];
    for (int i = t; i < (T - 1) * ad; i += blockDim.x) A[i] = ab[i];
    if (t < T - 1) {
      R[t] = rr;
      C[t] = cc;
      TS[t] = ts;
    }
    pos = T - 1;
  }
  for (int k = t; k < od; k += blockDim.x) S[(size_t)pos * od + k] = obs[(size_t)e * obs_ld + k];
  for (int k = t; k < ad; k += blockDim.x) A[(size_t)pos * ad + k] = 0.f;  // "last action is dummy with zeros"
  if (t == 0) {
    R[pos] = r_prev - rew;   // cdt.py:507
    C[pos] = c_prev - cost;  // cdt.py:508
    TS[pos] = step + 1;
    M[pos] = 1.f;
  }
}

__global__ void cursor_inc_kernel(int* cursor, int episode_len) {
  if (*cursor < episode_len) *cursor += 1;
}

}  // namespace

extern "C" int osrl_env_step(const osrl_env_t* env, const float* act, float* state, float* obs, int32_t obs_ld,
                             float* acc, float* step_out, int32_t episodes, void* stream) {
  if (!env || !act || !state || !obs || !acc || episodes < 1) return -1;
  if (env->state_dim < 1 || env->state_dim > kMaxDim || env->action_dim < 1 || env->action_dim > 64 ||
      obs_ld < env->state_dim || !env->At || !env->Bt || !env->w || !env->goal)
    return -1;
  (void)hipGetLastError();
  const int threads = ((env->state_dim + 63) / 64) * 64;
  hipLaunchKernelGGL(env_step_kernel, dim3(episodes), dim3(threads), 0, (hipStream_t)stream, *env, act, state, obs,
                     obs_ld, acc, step_out);
  return (int)hipGetLastError();
}

extern "C" int osrl_cdt_rollout_pick(const float* head, int32_t head_width, int32_t action_dim, int32_t episodes,
                                     int32_t seq_len, const int32_t* cursor, float max_action, float* act,
                                     void* stream) {
  if (!head || !cursor || !act || episodes < 1 || seq_len < 1 || action_dim < 1 || head_width < action_dim) return -1;
  (void)hipGetLastError();
  const int n = episodes * action_dim;
  hipLaunchKernelGGL(cdt_pick_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, head, head_width,
                     action_dim, episodes, seq_len, cursor, max_action, act);
  return (int)hipGetLastError();
}

extern "C" int osrl_cdt_rollout_push(float* states, float* actions, float* returns, float* costs_to_go,
                                     int64_t* time_steps, float* mask, int32_t episodes, int32_t seq_len,
                                     int32_t state_dim, int32_t action_dim, const float* act, const float* obs,
                                     int32_t obs_ld, const float* step_out, float cost_scale, int32_t cost_reverse,
                                     int32_t* cursor, int32_t episode_len, void* stream) {
  if (!states || !actions || !returns || !costs_to_go || !time_steps || !mask || !act || !obs || !step_out || !cursor ||
      episodes < 1 || seq_len < 1 || seq_len > 256 || state_dim < 1 || action_dim < 1 || obs_ld < state_dim)
    return -1;
  const size_t lds = (size_t)(seq_len - 1) * (state_dim + action_dim) * sizeof(float);
  if (lds > 64 * 1024) return -1;
  (void)hipGetLastError();
  CdtWin w{states, actions, returns, costs_to_go, time_steps, mask, episodes, seq_len, state_dim, action_dim};
  hipLaunchKernelGGL(cdt_push_kernel, dim3(episodes), dim3(256), lds, (hipStream_t)stream, w, act, obs, obs_ld,
                     step_out, cost_scale, cost_reverse, cursor, episode_len);
  hipLaunchKernelGGL(cursor_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, cursor, episode_len);
  return (int)hipGetLastError();
}
