// BEAR-Lagrangian glue kernels (osrl/algorithms/bearl.py of the reference).  The MLP work of the
// step runs on the fused kernels of mlp.hip; what is specific to BEAR is here:
//   bear_mmd:        per batch row, MMD between M raw (pre-tanh) VAE decodes x and M raw actor samples
//                    u = mu + sigma*eps (bearl.py:277-312, gaussian or laplacian kernel) and d MMD / d u,
//   bear_actor_loss: the scalar side of actor_loss (bearl.py:243-275): twin-min Q / Qc, PID multiplier, the
//                    q-term gate on n_train_steps, the log_alpha dual step, the statistics, dL/dq for the critics,
//   bear_head_bwd:   d loss / d (mu, log_std) of all B*M actor rows from the MMD gradient (all samples) and the
//                    critics' action gradient (sample 0 only, bearl.py:243-245).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/osrl_amd.h"

namespace {

constexpr float kLogStdMin = -20.0f, kLogStdMax = 2.0f;  // osrl/common/net.py:148-149
constexpr int kRed = 1024;
constexpr int kMmdWaves = 4;  // waves (= batch rows) per workgroup of bear_mmd_kernel; launch geometry as compile-time
                              // constants: `blockDim` is a load from the hidden kernarg block in every wave
constexpr int kMaxAd = 16;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ float block_sum(float v, float* sm /*>=17*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (kRed >> 6); ++i) t += sm[i];  // (always launched with kRed threads: no hidden-kernarg read)
    sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}

// One wave per batch row b; lane j < M owns sample j.  x = raw_vae[b*M + i], y_j = u[b*M + j].
//   k(a,b) = exp(-|a-b|^2 / (2 sigma))  (gaussian)   or   exp(-|a-b|_1 / (2 sigma))  (laplacian)
//   mmd = sqrt(mean k(x,x) + mean k(y,y) - 2 mean k(x,y) + 1e-6)
//   d mmd / d y_j = [ (2/M^2) sum_i dk(y_i,y_j)/dy_j - (2/M^2) sum_i dk(x_i,y_j)/dy_j ] / (2 mmd)
template <bool GAUSS>
__global__ __launch_bounds__(256) void bear_mmd_kernel(const float* __restrict__ raw_vae,
                                                       const float* __restrict__ head,
                                                       const float* __restrict__ eps, int B, int M, int ad,
                                                       float sigma, float* __restrict__ mmd,
                                                       float* __restrict__ du, float* __restrict__ tanh_u,
                                                       float* __restrict__ a0) {
  extern __shared__ float lds[];  // per wave: x[M*ad], y[M*ad]
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * kMmdWaves + wv;
  float* xs = lds + (size_t)wv * 2 * M * ad;
  float* ys = xs + (size_t)M * ad;
  const bool row_ok = b < B;
  if (row_ok) {
    for (int i = lane; i < M * ad; i += 64) {
      const int r = b * M + i / ad, k = i % ad;
      xs[i] = raw_vae[(size_t)r * ad + k];
      const float mu = head[(size_t)r * 2 * ad + k];
      const float ls = fminf(fmaxf(head[(size_t)r * 2 * ad + ad + k], kLogStdMin), kLogStdMax);
      const float u = mu + expf(ls) * eps[(size_t)r * ad + k];
      ys[i] = u;
      const float t = tanhf(u);
      tanh_u[(size_t)r * ad + k] = t;
      if (i < ad) a0[(size_t)b * ad + k] = t;  // sample 0 feeds the critics, unscaled (bearl.py:243-245)
    }
  }
  __syncthreads();
  if (!row_ok) return;
  float sxx = 0.f, sxy = 0.f, syy = 0.f;
  float g[kMaxAd];
#pragma unroll
  for (int k = 0; k < kMaxAd; ++k) g[k] = 0.f;
  const float inv2s = 1.0f / (2.0f * sigma);
  if (lane < M) {
    const float* yj = ys + (size_t)lane * ad;
    const float* xj = xs + (size_t)lane * ad;
    for (int i = 0; i < M; ++i) {
      const float* xi = xs + (size_t)i * ad;
      const float* yi = ys + (size_t)i * ad;
      float dxx = 0.f, dxy = 0.f, dyy = 0.f;
      for (int k = 0; k < ad; ++k) {
        const float a = xi[k] - xj[k], c = xi[k] - yj[k], d = yi[k] - yj[k];
        if (GAUSS) {
          dxx += a * a; dxy += c * c; dyy += d * d;
        } else {
          dxx += fabsf(a); dxy += fabsf(c); dyy += fabsf(d);
        }
      }
      const float kxx = expf(-dxx * inv2s), kxy = expf(-dxy * inv2s), kyy = expf(-dyy * inv2s);
      sxx += kxx; sxy += kxy; syy += kyy;
      for (int k = 0; k < ad; ++k) {
        const float c = xi[k] - yj[k], d = yi[k] - yj[k];
        // d k(.,y_j)/d y_j ; the y-y term counts twice (y_j is row and column of the symmetric kernel matrix)
        const float gxy = GAUSS ? kxy * c * 2.0f * inv2s : kxy * (c > 0.f ? 1.f : (c < 0.f ? -1.f : 0.f)) * inv2s;
        const float gyy = GAUSS ? kyy * d * 2.0f * inv2s : kyy * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv2s;
        g[k] += 2.0f * gyy - 2.0f * gxy;
      }
    }
  }
  const float mm = 1.0f / (float)(M * M);
  const float inner = (wave_sum(sxx) + wave_sum(syy) - 2.0f * wave_sum(sxy)) * mm + 1e-6f;
  const float v = sqrtf(inner);
  if (lane == 0) mmd[b] = v;
  if (lane < M) {
    const float sc = mm / (2.0f * v);
    for (int k = 0; k < ad; ++k) du[((size_t)b * M + lane) * ad + k] = g[k] * sc;
  }
}

// min over the q1 group, min over the q2 group, then the binary min with torch's tie rule (as glue.hip minmin)
__device__ __forceinline__ float minmin(const float* __restrict__ q, int n1, int n2, int rows, int b, int* i1,
                                        int* i2, float* w1) {
  int a1 = 0, a2 = 0;
  float m1 = q[b], m2 = q[(size_t)n1 * rows + b];
  for (int e = 1; e < n1; ++e) {
    const float v = q[(size_t)e * rows + b];
    if (v < m1) { m1 = v; a1 = e; }
  }
  for (int e = 1; e < n2; ++e) {
    const float v = q[(size_t)(n1 + e) * rows + b];
    if (v < m2) { m2 = v; a2 = e; }
  }
  *i1 = a1;
  *i2 = a2;
  *w1 = m1 < m2 ? 1.0f : (m1 == m2 ? 0.5f : 0.0f);
  return fminf(m1, m2);
}

struct BearLoss {
  const float* q;
  const float* qc;
  const float* mmd;
  int nq1, nq2, nc1, nc2, rows;
  float qc_thres, KP, KI, KD, inv_rows, stat_share, alpha_lr, thresh;
  const float* means_in;  // all-reduced (mean q, mean qc, mean mmd) under data parallelism, else null
  const osrl_step_state_t* st;
  int64_t start_step;
  float* pid;
  float* log_alpha;
  float* dq;
  float* dqc;
  float* coef;  // out: exp(log_alpha_before) * inv_rows -- scales d mmd / d u in bear_head_bwd
  float* stat;  // out[5]: actor_loss, mmd_loss, qc_penalty, lagrangian, alpha_value
};

__global__ __launch_bounds__(kRed) void bear_actor_sums_kernel(BearLoss a, float* __restrict__ out) {
  __shared__ float sm[20];
  float sq = 0.f, sqc = 0.f, sm_ = 0.f;
  int i1, i2;
  float w1;
  for (int b = threadIdx.x; b < a.rows; b += kRed) {
    sq += minmin(a.q, a.nq1, a.nq2, a.rows, b, &i1, &i2, &w1);
    sqc += minmin(a.qc, a.nc1, a.nc2, a.rows, b, &i1, &i2, &w1);
    sm_ += a.mmd[b];
  }
  sq = block_sum(sq, sm);
  sqc = block_sum(sqc, sm);
  sm_ = block_sum(sm_, sm);
  if (threadIdx.x == 0) {
    out[0] = sq * a.inv_rows;
    out[1] = sqc * a.inv_rows;
    out[2] = sm_ * a.inv_rows;
  }
}

__global__ __launch_bounds__(kRed) void bear_actor_loss_kernel(BearLoss a) {
  __shared__ float sm[20];
  __shared__ float s_mult, s_useq;
  float sq = 0.f, sqc = 0.f, smm = 0.f;
  int i1, i2;
  float w1;
  if (!a.means_in) {
    for (int b = threadIdx.x; b < a.rows; b += kRed) {
      sq += minmin(a.q, a.nq1, a.nq2, a.rows, b, &i1, &i2, &w1);
      sqc += minmin(a.qc, a.nc1, a.nc2, a.rows, b, &i1, &i2, &w1);
      smm += a.mmd[b];
    }
    sq = block_sum(sq, sm) * a.inv_rows;
    sqc = block_sum(sqc, sm) * a.inv_rows;
    smm = block_sum(smm, sm) * a.inv_rows;
  } else {
    sq = a.means_in[0];
    sqc = a.means_in[1];
    smm = a.means_in[2];
  }
  if (threadIdx.x == 0) {
    // LagrangianPIDController.control  net.py:376-387
    const float e_new = sqc - a.qc_thres;
    const float e_old = a.pid[0], integ = a.pid[1];
    const float diff = fmaxf(e_new - e_old, 0.f);
    const float integ_new = fmaxf(integ + e_new, 0.f);
    a.pid[0] = e_new;
    a.pid[1] = integ_new;
    const float mult = fmaxf(a.KP * fmaxf(e_new, 0.f) + a.KI * integ_new + a.KD * diff, 0.f);
    s_mult = mult;
    const float penalty = (sqc - a.qc_thres) * mult;
    // n_train_steps counts completed actor updates = device step - 1 (the step ticks first)  bearl.py:254-259,268
    const bool use_q = (a.st->step - 1) >= a.start_step;
    s_useq = use_q ? 1.f : 0.f;
    const float la = a.log_alpha[0];
    const float alpha = expf(la);
    a.coef[0] = alpha * a.inv_rows;
    // log_alpha += alpha_lr * exp(log_alpha) * mean(mmd - thresh); clamp [-5, 5]   bearl.py:265-267
    const float la_new = fminf(fmaxf(la + a.alpha_lr * alpha * (smm - a.thresh), -5.0f), 5.0f);
    a.log_alpha[0] = la_new;
    if (a.stat) {
      a.stat[0] = ((use_q ? -sq : 0.f) + alpha * (smm - a.thresh) + penalty) * a.stat_share;
      a.stat[1] = smm * a.stat_share;
      a.stat[2] = penalty * a.stat_share;
      a.stat[3] = mult * a.stat_share;
      a.stat[4] = expf(la_new) * a.stat_share;
    }
  }
  __syncthreads();
  const float mult = s_mult, gq = s_useq;
  for (int b = threadIdx.x; b < a.rows; b += kRed) {
    minmin(a.q, a.nq1, a.nq2, a.rows, b, &i1, &i2, &w1);
    for (int e = 0; e < a.nq1; ++e) a.dq[(size_t)e * a.rows + b] = e == i1 ? -gq * w1 * a.inv_rows : 0.f;
    for (int e = 0; e < a.nq2; ++e)
      a.dq[(size_t)(a.nq1 + e) * a.rows + b] = e == i2 ? -gq * (1.0f - w1) * a.inv_rows : 0.f;
    minmin(a.qc, a.nc1, a.nc2, a.rows, b, &i1, &i2, &w1);
    for (int e = 0; e < a.nc1; ++e) a.dqc[(size_t)e * a.rows + b] = e == i1 ? w1 * mult * a.inv_rows : 0.f;
    for (int e = 0; e < a.nc2; ++e)
      a.dqc[(size_t)(a.nc1 + e) * a.rows + b] = e == i2 ? (1.0f - w1) * mult * a.inv_rows : 0.f;
  }
}

__global__ void bear_head_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                     const float* __restrict__ tanh_u, const float* __restrict__ du_mmd,
                                     const float* __restrict__ coef, const float* __restrict__ da_nets, int n_nets,
                                     int B, int M, int ad, float* __restrict__ dhead) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // (launched with 256 threads)
  if (i >= B * M * ad) return;
  const int r = i / ad, k = i - r * ad;
  const int b = r / M;
  float du = coef[0] * du_mmd[i];
  if (r - b * M == 0) {  // the critics saw sample 0 only
    float da = 0.f;
    for (int e = 0; e < n_nets; ++e) da += da_nets[((size_t)e * B + b) * ad + k];
    const float t = tanh_u[i];
    du += da * (1.0f - t * t);
  }
  const float lsr = head[(size_t)r * 2 * ad + ad + k];
  const float ls = fminf(fmaxf(lsr, kLogStdMin), kLogStdMax);
  const bool inside = lsr >= kLogStdMin && lsr <= kLogStdMax;
  dhead[(size_t)r * 2 * ad + k] = du;
  dhead[(size_t)r * 2 * ad + ad + k] = inside ? du * eps[i] * expf(ls) : 0.f;
}

BearLoss make_loss(const float* q, int nq1, int nq2, const float* qc, int nc1, int nc2, const float* mmd, int rows,
                   int rows_global) {
  BearLoss a{};
  a.q = q; a.qc = qc; a.mmd = mmd;
  a.nq1 = nq1; a.nq2 = nq2; a.nc1 = nc1; a.nc2 = nc2; a.rows = rows;
  a.inv_rows = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  return a;
}

}  // namespace

#define S ((hipStream_t)stream)

extern "C" int osrl_bear_mmd(const float* raw_vae, const float* head, const float* eps, int32_t rows,
                             int32_t n_samples, int32_t ad, float sigma, int32_t kernel, float* mmd, float* du,
                             float* tanh_u, float* a0, void* stream) {
  if (!raw_vae || !head || !eps || !mmd || !du || !tanh_u || !a0 || rows < 1 || n_samples < 1 || n_samples > 64 ||
      ad < 1 || ad > kMaxAd || !(sigma > 0.f) || (kernel != OSRL_MMD_GAUSSIAN && kernel != OSRL_MMD_LAPLACIAN))
    return -1;
  (void)hipGetLastError();
  const int waves = kMmdWaves;
  const size_t lds = (size_t)waves * 2 * n_samples * ad * sizeof(float);
  const dim3 grid((rows + waves - 1) / waves), block(64 * waves);
  if (kernel == OSRL_MMD_GAUSSIAN)
    hipLaunchKernelGGL(bear_mmd_kernel<true>, grid, block, lds, S, raw_vae, head, eps, rows, n_samples, ad, sigma, mmd,
                       du, tanh_u, a0);
  else
    hipLaunchKernelGGL(bear_mmd_kernel<false>, grid, block, lds, S, raw_vae, head, eps, rows, n_samples, ad, sigma,
                       mmd, du, tanh_u, a0);
  return (int)hipGetLastError();
}

extern "C" int osrl_bear_actor_sums(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1,
                                    int32_t nc2, const float* mmd, int32_t rows, int32_t rows_global, float* out,
                                    void* stream) {
  if (!q || !qc || !mmd || !out || rows < 1 || nq1 < 1 || nq2 < 1 || nc1 < 1 || nc2 < 1) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bear_actor_sums_kernel, dim3(1), dim3(kRed), 0, S,
                     make_loss(q, nq1, nq2, qc, nc1, nc2, mmd, rows, rows_global), out);
  return (int)hipGetLastError();
}

extern "C" int osrl_bear_actor_loss(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1,
                                    int32_t nc2, const float* mmd, int32_t rows, float qc_thres, float KP, float KI,
                                    float KD, float target_mmd_thresh, float alpha_lr, int64_t start_update_policy_step,
                                    int32_t rows_global, const float* global_means, float stat_share,
                                    const osrl_step_state_t* st, float* pid, float* log_alpha, float* dq, float* dqc,
                                    float* coef, float* stat, void* stream) {
  if (!q || !qc || !mmd || !st || !pid || !log_alpha || !dq || !dqc || !coef || rows < 1 || nq1 < 1 || nq2 < 1 ||
      nc1 < 1 || nc2 < 1)
    return -1;
  (void)hipGetLastError();
  BearLoss a = make_loss(q, nq1, nq2, qc, nc1, nc2, mmd, rows, rows_global);
  a.qc_thres = qc_thres; a.KP = KP; a.KI = KI; a.KD = KD;
  a.thresh = target_mmd_thresh; a.alpha_lr = alpha_lr; a.start_step = start_update_policy_step;
  a.means_in = global_means; a.stat_share = stat_share; a.st = st;
  a.pid = pid; a.log_alpha = log_alpha; a.dq = dq; a.dqc = dqc; a.coef = coef; a.stat = stat;
  hipLaunchKernelGGL(bear_actor_loss_kernel, dim3(1), dim3(kRed), 0, S, a);
  return (int)hipGetLastError();
}

extern "C" int osrl_bear_head_bwd(const float* head, const float* eps, const float* tanh_u, const float* du_mmd,
                                  const float* coef, const float* da_nets, int32_t n_nets, int32_t rows,
                                  int32_t n_samples, int32_t ad, float* dhead, void* stream) {
  if (!head || !eps || !tanh_u || !du_mmd || !coef || !da_nets || !dhead || rows < 1 || n_samples < 1 || ad < 1 ||
      n_nets < 1)
    return -1;
  (void)hipGetLastError();
  const int n = rows * n_samples * ad;
  hipLaunchKernelGGL(bear_head_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, S, head, eps, tanh_u, du_mmd, coef,
                     da_nets, n_nets, rows, n_samples, ad, dhead);
  return (int)hipGetLastError();
}
