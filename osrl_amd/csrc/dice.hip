// COptiDICE glue kernels (osrl/algorithms/coptidice.py of the reference).  The nu / chi / actor
// networks run on the fused MLP kernels of mlp.hip over the stacked rows [obs; next_obs] (2B rows, so one saved
// forward serves both the s and the s' terms of every loss); what is specific to COptiDICE is here: the optimal
// importance weights w*(s,a), the chi / tau upper-bound estimator with its BATCH-global softmax, the nu and lambda
// losses, the weighted log-likelihood policy extraction, and Adam on the two scalar leaves.
//
// Scalars: `leaves` = {tau, m_tau, v_tau, lmbda, m_lmbda, v_lmbda} (raw, pre-softplus, as coptidice.py:96-97);
// `work` = {softplus(lmbda), softplus(tau), weighted_c} of THIS step: the policy extraction re-evaluates w* with the
// lambda' computed at the top of update(), before the lambda step (coptidice.py:138,211-213).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/osrl_amd.h"

namespace {

constexpr int kRed = 1024;
constexpr float kLogStdMin = -20.0f, kLogStdMax = 2.0f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
template <bool MAX>
__device__ float block_red(float v, float* sm /*>=17*/) {
  v = MAX ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = sm[0];
    for (int i = 1; i < (kRed >> 6); ++i) t = MAX ? fmaxf(t, sm[i]) : t + sm[i];  // (kRed-thread launches only)
    sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// get_f_div_fn (coptidice.py:15-38): f, f', g = f'^{-1}, g'
__device__ __forceinline__ float f_fn(int t, float w) {
  switch (t) {
    case OSRL_F_CHI2: return 0.5f * (w - 1.f) * (w - 1.f);
    case OSRL_F_SOFTCHI: return w < 1.f ? w * (logf(w + 1e-10f) - 1.f) + 1.f : 0.5f * (w - 1.f) * (w - 1.f);
    default: return w * logf(w + 1e-10f);
  }
}
__device__ __forceinline__ float f_prime(int t, float w) {
  switch (t) {
    case OSRL_F_CHI2: return w - 1.f;
    case OSRL_F_SOFTCHI: return w < 1.f ? logf(w + 1e-10f) + w / (w + 1e-10f) - 1.f : w - 1.f;
    default: return logf(w + 1e-10f) + w / (w + 1e-10f);
  }
}
__device__ __forceinline__ float g_fn(int t, float x) {
  switch (t) {
    case OSRL_F_CHI2: return x + 1.f;
    case OSRL_F_SOFTCHI: return x < 0.f ? expf(fminf(x, 0.f)) : x + 1.f;
    default: return expf(x - 1.f);
  }
}
__device__ __forceinline__ float g_prime(int t, float x) {
  switch (t) {
    case OSRL_F_CHI2: return 1.f;
    case OSRL_F_SOFTCHI: return x < 0.f ? expf(fminf(x, 0.f)) : 1.f;
    default: return expf(x - 1.f);
  }
}

// EnsembleQCritic.predict (net.py:236-238): min over the nets (first minimum wins, as torch.min(dim=0))
__device__ __forceinline__ float net_min(const float* __restrict__ y, int n, int rows2, int r, int* arg) {
  float m = y[r];
  int a = 0;
  for (int k = 1; k < n; ++k) {
    const float v = y[(size_t)k * rows2 + r];
    if (v < m) { m = v; a = k; }
  }
  *arg = a;
  return m;
}

// _optimal_w (coptidice.py:122-131): e = r - lambda' c + gamma (1-d) nu(s') - nu(s); w = relu(f'^{-1}(e / alpha))
__global__ void dice_w_kernel(const float* __restrict__ nu2, int n_nu, int B, const float* __restrict__ rew,
                              const float* __restrict__ cost, const float* __restrict__ done,
                              const float* __restrict__ leaves, float* __restrict__ work, int use_saved, float alpha,
                              float gamma, int f_type, float* __restrict__ e_out, float* __restrict__ w_out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const float lam = use_saved ? work[0] : softplus(leaves[3]);
  if (b == 0 && !use_saved) {
    work[0] = lam;
    work[1] = softplus(leaves[0]);
  }
  if (b >= B) return;
  int a;
  const float nu_s = net_min(nu2, n_nu, 2 * B, b, &a), nu_n = net_min(nu2, n_nu, 2 * B, B + b, &a);
  const float e = rew[b] - lam * cost[b] + gamma * (1.f - done[b]) * nu_n - nu_s;
  if (e_out) e_out[b] = e;
  w_out[b] = fmaxf(g_fn(f_type, e / alpha), 0.f);
}

__device__ __forceinline__ void scalar_adam(float* leaf /*p,m,v*/, float g, float lr, const osrl_step_state_t* st) {
  const float m = 0.9f * leaf[1] + 0.1f * g;
  const float v = 0.999f * leaf[2] + 0.001f * g * g;
  leaf[1] = m;
  leaf[2] = v;
  leaf[0] -= lr / st->bc1 * m / (sqrtf(v) / st->bc2_sqrt + 1e-8f);
}

struct ChiArgs {
  const float* chi2;  // [n_chi, 2B] or null when cost_ub_epsilon == 0
  const float* w;
  const float* cost;
  const float* done;
  const float* init;
  int n_chi, B;
  float gamma, p0, eps_ub, scalar_lr;
  const osrl_step_state_t* st;
  float* leaves;
  float* work;
  float* ell;   // [B] scratch (single GPU) / this rank's ell (data parallel, filled by dice_chi_ell_kernel)
  float* dchi;  // [n_chi, 2B]
  float* stat;  // chi_loss, tau_loss, D_kl
  // data parallel: the softmax runs over the GLOBAL batch -- ell_all = all-gathered ell [Bg], this rank's rows start
  // at row0; global scalars are written x stat_share (the statistics vector is all-reduced with SUM)
  const float* ell_all;
  int Bg, row0;
  float stat_share;
};

__device__ __forceinline__ float chi_ell(const ChiArgs& a, int b, int* i1, int* i2) {
  const int B = a.B;
  const float cs = net_min(a.chi2, a.n_chi, 2 * B, b, i1), cn = net_min(a.chi2, a.n_chi, 2 * B, B + b, i2);
  return (1.f - a.gamma) * cs * a.init[b] / a.p0 + a.w[b] * (a.cost[b] + a.gamma * (1.f - a.done[b]) * cn - cs);
}

// ell of this rank's rows (data parallel: all-gathered before dice_chi_kernel)
__global__ void dice_chi_ell_kernel(ChiArgs a) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= a.B) return;
  int i1, i2;
  a.ell[b] = chi_ell(a, b, &i1, &i2);
}

// coptidice.py:149-185 in one workgroup (the softmax runs over the whole batch: "dim=0")
__global__ __launch_bounds__(kRed) void dice_chi_kernel(ChiArgs a) {
  __shared__ float sm[20];
  const int B = a.B;
  const int Bg = a.Bg;  // global batch (== B on one GPU)
  const float invB = 1.f / (float)Bg;
  if (!a.chi2) {  // cost_ub_epsilon == 0: weighted_c = mean(w c) (this rank's share), no chi / tau update
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += kRed) s += a.w[b] * a.cost[b];
    s = block_red<false>(s, sm);
    if (threadIdx.x == 0) {
      a.work[2] = s * invB;
      a.stat[0] = a.stat[1] = a.stat[2] = 0.f;
    }
    return;
  }
  const float tau = a.work[1];
  if (!a.ell_all) {
    for (int b = threadIdx.x; b < B; b += kRed) {
      int i1, i2;
      a.ell[b] = chi_ell(a, b, &i1, &i2);
    }
    __syncthreads();
  }
  const float* __restrict__ ell = a.ell_all ? a.ell_all : a.ell;  // the GLOBAL batch
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < Bg; i += kRed) mx = fmaxf(mx, ell[i] / tau);
  mx = block_red<true>(mx, sm);
  float se = 0.f;
  for (int i = threadIdx.x; i < Bg; i += kRed) se += expf(ell[i] / tau - mx);
  se = block_red<false>(se, sm);
  const float lse = logf(se), logB = logf((float)Bg);
  float dkl = 0.f, cl = 0.f;
  for (int i = threadIdx.x; i < Bg; i += kRed) {
    const float lsm = ell[i] / tau - mx - lse;
    const float wt = expf(lsm) * (float)Bg;
    dkl += wt * (lsm + logB) - wt + 1.f;
    cl += wt * ell[i];
  }
  dkl = block_red<false>(dkl, sm) * invB;
  cl = block_red<false>(cl, sm) * invB;
  // this rank's rows: weighted_c share and d chi_loss / d ell_i = s_i (1 + (ell_i - chi_loss) / tau') -- `weights` is
  // not detached in the reference (coptidice.py:165,173)
  const int row0 = a.row0;
  float wc = 0.f;
  for (int b = threadIdx.x; b < B; b += kRed) {
    const float l = ell[row0 + b];
    const float sft = expf(l / tau - mx - lse);
    wc += sft * (float)Bg * a.w[b] * a.cost[b];
    const float dl = sft * (1.f + (l - cl) / tau);
    int i1, i2;
    net_min(a.chi2, a.n_chi, 2 * B, b, &i1);
    net_min(a.chi2, a.n_chi, 2 * B, B + b, &i2);
    const float ds = dl * ((1.f - a.gamma) * a.init[b] / a.p0 - a.w[b]);
    const float dn = dl * a.w[b] * a.gamma * (1.f - a.done[b]);
    for (int k = 0; k < a.n_chi; ++k) {
      a.dchi[(size_t)k * 2 * B + b] = k == i1 ? ds : 0.f;
      a.dchi[(size_t)k * 2 * B + B + b] = k == i2 ? dn : 0.f;
    }
  }
  wc = block_red<false>(wc, sm) * invB;
  if (threadIdx.x == 0) {
    a.work[2] = wc;  // data parallel: this rank's share (all-reduced by the caller)
    a.stat[0] = cl * a.stat_share;
    a.stat[1] = tau * (a.eps_ub - dkl) * a.stat_share;  // tau_loss (coptidice.py:180)
    a.stat[2] = dkl * a.stat_share;
    scalar_adam(a.leaves + 0, sigmoid(a.leaves[0]) * (a.eps_ub - dkl), a.scalar_lr, a.st);
  }
}

struct NuArgs {
  const float* nu2;
  const float* e;
  const float* w;
  const float* done;
  const float* init;
  int n_nu, B, f_type;
  float inv_rows, stat_share;  // 1 / global batch; share of the global scalars in the all-reduced statistics
  float gamma, alpha, p0, qc_thres, scalar_lr;
  const osrl_step_state_t* st;
  float* leaves;
  const float* work;
  float* dnu;   // [n_nu, 2B]
  float* stat;  // Df, td_error, nu_loss, lmbda_loss, (actor_loss), tau, lmbda  -> indices 0,1,2,3,5,6
};

// coptidice.py:147,188-201: nu loss (+ Df, td_error) and the lambda step
__global__ __launch_bounds__(kRed) void dice_nu_kernel(NuArgs a) {
  __shared__ float sm[20];
  const int B = a.B;
  const float invB = a.inv_rows;
  float df = 0.f, td = 0.f, nl = 0.f;
  for (int b = threadIdx.x; b < B; b += kRed) {
    int i1, i2;
    const float nu_s = net_min(a.nu2, a.n_nu, 2 * B, b, &i1);
    net_min(a.nu2, a.n_nu, 2 * B, B + b, &i2);
    const float e = a.e[b], w = a.w[b];
    const float fw = f_fn(a.f_type, w);
    df += fw;
    td += e * e;
    nl += (1.f - a.gamma) * nu_s * a.init[b] / a.p0 + w * e - a.alpha * fw;
    // d/de (w e - alpha f(w)) = w + (e - alpha f'(w)) dw/de,  dw/de = [g > 0] g'(e/alpha) / alpha
    const float x = e / a.alpha;
    const float dwde = g_fn(a.f_type, x) > 0.f ? g_prime(a.f_type, x) / a.alpha : 0.f;
    const float de = (w + (e - a.alpha * f_prime(a.f_type, w)) * dwde) * invB;
    const float ds = (1.f - a.gamma) * a.init[b] * invB / a.p0 - de;
    const float dn = de * a.gamma * (1.f - a.done[b]);
    for (int k = 0; k < a.n_nu; ++k) {
      a.dnu[(size_t)k * 2 * B + b] = k == i1 ? ds : 0.f;
      a.dnu[(size_t)k * 2 * B + B + b] = k == i2 ? dn : 0.f;
    }
  }
  df = block_red<false>(df, sm) * invB;
  td = block_red<false>(td, sm) * invB;
  nl = block_red<false>(nl, sm) * invB;
  if (threadIdx.x == 0) {
    const float lam = a.work[0], wc = a.work[2];
    a.stat[0] = df;
    a.stat[1] = td;
    a.stat[2] = nl;
    a.stat[3] = lam * (a.qc_thres - wc) * a.stat_share;  // lmbda_loss (coptidice.py:197)
    a.stat[5] = a.work[1] * a.stat_share;  // tau' and lambda' of the top of the step (coptidice.py:229-230)
    a.stat[6] = lam * a.stat_share;
    scalar_adam(a.leaves + 3, sigmoid(a.leaves[3]) * (a.qc_thres - wc), a.scalar_lr, a.st);
  }
}

// out[r, k] = x[r, k] + eps[r, k] * std[k] * scale   (coptidice.py:204-205)
__global__ void dice_perturb_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                    const float* __restrict__ std, int rows, int d, float scale,
                                    float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * d) return;
  out[i] = x[i] + eps[i] * std[i % d] * scale;
}

// actor_loss = -mean(w * sum_k Normal(mu, sigma).log_prob(a)) on the PRE-tanh Gaussian (coptidice.py:207-215)
__global__ __launch_bounds__(kRed) void dice_actor_kernel(const float* __restrict__ head, const float* __restrict__ act,
                                                          const float* __restrict__ w, int B, int ad, float invB,
                                                          float* __restrict__ dhead, float* __restrict__ stat) {
  __shared__ float sm[20];
  float loss = 0.f;
  for (int b = threadIdx.x; b < B; b += kRed) {
    float lp = 0.f;
    const float c = -w[b] * invB;
    for (int k = 0; k < ad; ++k) {
      const float mu = head[(size_t)b * 2 * ad + k];
      const float lsr = head[(size_t)b * 2 * ad + ad + k];
      const float ls = fminf(fmaxf(lsr, kLogStdMin), kLogStdMax);
      const float inv_var = expf(-2.f * ls);
      const float d = act[(size_t)b * ad + k] - mu;
      lp += -0.5f * d * d * inv_var - ls - 0.9189385332046727f;
      dhead[(size_t)b * 2 * ad + k] = c * d * inv_var;
      dhead[(size_t)b * 2 * ad + ad + k] = (lsr >= kLogStdMin && lsr <= kLogStdMax) ? c * (d * d * inv_var - 1.f) : 0.f;
    }
    loss += w[b] * lp;
  }
  loss = block_red<false>(loss, sm);
  if (threadIdx.x == 0) stat[0] = -loss * invB;
}

}  // namespace

#define S ((hipStream_t)stream)

extern "C" int osrl_dice_optimal_w(const float* nu2, int32_t n_nu, int32_t rows, const float* rew, const float* cost,
                                   const float* done, const float* leaves, float* work, int32_t use_saved_lambda,
                                   float alpha, float gamma, int32_t f_type, float* e, float* w, void* stream) {
  if (!nu2 || !rew || !cost || !done || !leaves || !work || !w || rows < 1 || n_nu < 1 || !(alpha > 0.f) ||
      f_type < OSRL_F_CHI2 || f_type > OSRL_F_KL)
    return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(dice_w_kernel, dim3((rows + 255) / 256), dim3(256), 0, S, nu2, n_nu, rows, rew, cost, done, leaves,
                     work, use_saved_lambda, alpha, gamma, f_type, e, w);
  return (int)hipGetLastError();
}

static bool chi_args(ChiArgs* a, const float* chi2, int32_t n_chi, int32_t rows, const float* w, const float* cost,
                     const float* done, const float* is_init, float gamma, float p0, float* ell) {
  if (!chi2 || !w || !cost || !done || !is_init || !ell || rows < 1 || n_chi < 1 || !(p0 > 0.f)) return false;
  *a = ChiArgs{};
  a->chi2 = chi2; a->w = w; a->cost = cost; a->done = done; a->init = is_init;
  a->n_chi = n_chi; a->B = rows; a->gamma = gamma; a->p0 = p0; a->ell = ell;
  return true;
}

extern "C" int osrl_dice_chi_ell(const float* chi2, int32_t n_chi, int32_t rows, const float* w, const float* cost,
                                 const float* done, const float* is_init, float gamma, float init_state_propotion,
                                 float* ell, void* stream) {
  ChiArgs a;
  if (!chi_args(&a, chi2, n_chi, rows, w, cost, done, is_init, gamma, init_state_propotion, ell)) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(dice_chi_ell_kernel, dim3((rows + 255) / 256), dim3(256), 0, S, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_dice_chi_step(const float* chi2, int32_t n_chi, int32_t rows, const float* w, const float* cost,
                                  const float* done, const float* is_init, float gamma, float init_state_propotion,
                                  float cost_ub_epsilon, float scalar_lr, const osrl_step_state_t* st, float* leaves,
                                  float* work, float* ell_ws, float* dchi, const float* ell_all, int32_t rows_global,
                                  int32_t row0, float stat_share, float* stat, void* stream) {
  if (!w || !cost || !done || !is_init || !st || !leaves || !work || !stat || rows < 1) return -1;
  if (chi2 && (!ell_ws || !dchi || n_chi < 1 || !(init_state_propotion > 0.f))) return -1;
  if (ell_all && (rows_global < rows || row0 < 0 || row0 + rows > rows_global)) return -1;
  (void)hipGetLastError();
  ChiArgs a{};
  a.chi2 = chi2; a.w = w; a.cost = cost; a.done = done; a.init = is_init;
  a.n_chi = n_chi; a.B = rows; a.gamma = gamma; a.p0 = init_state_propotion; a.eps_ub = cost_ub_epsilon;
  a.scalar_lr = scalar_lr; a.st = st; a.leaves = leaves; a.work = work; a.ell = ell_ws; a.dchi = dchi; a.stat = stat;
  a.ell_all = ell_all; a.row0 = ell_all ? row0 : 0; a.stat_share = stat_share;
  a.Bg = ((ell_all || !chi2) && rows_global > rows) ? rows_global : rows;  // (no chi: this rank's share of mean(w c))
  hipLaunchKernelGGL(dice_chi_kernel, dim3(1), dim3(kRed), 0, S, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_dice_nu_step(const float* nu2, int32_t n_nu, int32_t rows, const float* e, const float* w,
                                 const float* done, const float* is_init, int32_t f_type, float gamma, float alpha,
                                 float init_state_propotion, float qc_thres, float scalar_lr, int32_t rows_global,
                                 float stat_share, const osrl_step_state_t* st, float* leaves, const float* work,
                                 float* dnu, float* stat, void* stream) {
  if (!nu2 || !e || !w || !done || !is_init || !st || !leaves || !work || !dnu || !stat || rows < 1 || n_nu < 1 ||
      !(init_state_propotion > 0.f) || f_type < OSRL_F_CHI2 || f_type > OSRL_F_KL)
    return -1;
  (void)hipGetLastError();
  NuArgs a{nu2, e, w, done, is_init, n_nu, rows, f_type, 1.0f / (float)(rows_global > 0 ? rows_global : rows),
           stat_share, gamma, alpha, init_state_propotion, qc_thres, scalar_lr, st, leaves, work, dnu, stat};
  hipLaunchKernelGGL(dice_nu_kernel, dim3(1), dim3(kRed), 0, S, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_dice_perturb(const float* x, const float* eps, const float* std, int32_t rows, int32_t dim,
                                 float scale, float* out, void* stream) {
  if (!x || !eps || !std || !out || rows < 1 || dim < 1) return -1;
  (void)hipGetLastError();
  const int n = rows * dim;
  hipLaunchKernelGGL(dice_perturb_kernel, dim3((n + 255) / 256), dim3(256), 0, S, x, eps, std, rows, dim, scale, out);
  return (int)hipGetLastError();
}

extern "C" int osrl_dice_actor_loss(const float* head, const float* act, const float* w, int32_t rows, int32_t ad,
                                    int32_t rows_global, float* dhead, float* stat, void* stream) {
  if (!head || !act || !w || !dhead || !stat || rows < 1 || ad < 1) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(dice_actor_kernel, dim3(1), dim3(kRed), 0, S, head, act, w, rows, ad,
                     1.0f / (float)(rows_global > 0 ? rows_global : rows), dhead, stat);
  return (int)hipGetLastError();
}
