// glue.hip -- the elementwise / reduction tails of the OSRL loss functions for gfx950.
//
// Everything between two fused-MLP launches of a train step: distribution heads, Bellman backups,
// loss values and the loss gradients d(loss)/d(net output) that seed the MLP backward kernels,
// the CPQ OOD quantile, the log_alpha ascent and the PID-Lagrangian controller.  All scalar state
// (log_alpha, PID integrators, logged statistics) stays device-resident: no .item() host sync
// (the reference syncs 5-6 times per step: cpq.py:134,152,198,199,216; bcql.py:131,154,178,205-207).
//
// These are latency-bound kernels on [rows, <=16] arrays (rows = batch or N*batch); the
// batch-global reductions run in ONE 1024-thread workgroup with wavefront shuffles + a 16-entry
// LDS stage, in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "argmem.h"
#include "trace.h"

namespace {

constexpr float kLogStdMin = -20.0f, kLogStdMax = 2.0f;  // net.py:148-149
constexpr float kVaeLsMin = -4.0f, kVaeLsMax = 15.0f;    // net.py:325
constexpr int kRed = 1024;
// Launch geometry as compile-time constants (round 3): `blockDim` / `gridDim` are read from the hidden block behind a
// kernel's explicit arguments, i.e. by an s_load from the kernarg segment in every wave -- a PCIe round trip at the head
// of every wave where the runtime keeps kernel arguments in host memory, and beyond the 14 dwords that kernarg preload
// hands over.  The element-wise kernels are always launched with kEw threads (GRID_1D), the batch-sum kernels with kRed.
constexpr int kEw = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// sum over a 1024-thread block; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sm /*>=17 floats*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(kRed >> 6); ++i) t += sm[i];
    sm[16] = t;
  }
  __syncthreads();
  return sm[16];
}
__device__ __forceinline__ float softplus(float x) {  // log(1+exp(x)), F.softplus
  return x > 20.0f ? x : log1pf(expf(x));
}

#define GRID_1D(n) dim3((unsigned)(((n) + kEw - 1) / kEw)), dim3(kEw)

// ---------------- squashed Gaussian head ----------------
__global__ void gauss_head_kernel(const float* __restrict__ head, const float* __restrict__ eps, int rows, int ad,
                                  float max_a, float* __restrict__ a, float* __restrict__ tanh_u,
                                  float* __restrict__ logp) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const int r = blockIdx.x * kEw + threadIdx.x;
  if (r >= rows) return;
  float lp = 0.f;
  for (int j = 0; j < ad; ++j) {
    const float mu = head[(size_t)r * 2 * ad + j];
    const float ls = fminf(fmaxf(head[(size_t)r * 2 * ad + ad + j], kLogStdMin), kLogStdMax);
    const float sd = expf(ls);
    const float e = eps ? eps[(size_t)r * ad + j] : 0.f;
    const float u = mu + sd * e;
    const float t = tanhf(u);
    if (a) a[(size_t)r * ad + j] = max_a * t;
    if (tanh_u) tanh_u[(size_t)r * ad + j] = t;
    // Normal(mu,sd).log_prob(u) - 2*(log2 - u - softplus(-2u))   (net.py:191-193)
    lp += -0.5f * e * e - ls - 0.9189385332046727f;
    lp -= 2.0f * (0.6931471805599453f - u - softplus(-2.0f * u));
  }
  if (logp) logp[r] = lp;
}

// A loaded value that is only consumed under a condition gets its load SUNK into that branch by the compiler (and a
// select on it turned into such a branch), where it waits alone: s_waitcnt vmcnt(0) per load, one memory round trip
// after the other.  pin() makes the value unconditionally live at the point of the call, so a group of loads issued
// back to back in front of a group of pin()s stays in flight together.
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }

__global__ void gauss_head_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                      const float* __restrict__ tanh_u, const float* __restrict__ da_nets,
                                      int n_nets, int rows, int ad, float max_a, float* __restrict__ dhead) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const int i = blockIdx.x * kEw + threadIdx.x;
  if (i >= rows * ad) return;
  const int r = i / ad, j = i - r * ad;
  float t = tanh_u[i];
  float lsr = head[(size_t)r * 2 * ad + ad + j];
  float ev = eps[i];
  float da = 0.f;
  if (n_nets <= 4) {  // the usual ensemble sizes: every member requested at once (members past n_nets re-read member 0)
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = da_nets[(size_t)(e < n_nets ? e : 0) * rows * ad + i];
#pragma unroll
    for (int e = 0; e < 4; ++e) pin(v[e]);
#pragma unroll
    for (int e = 0; e < 4; ++e) da += e < n_nets ? v[e] : 0.f;
  } else {
    for (int e = 0; e < n_nets; ++e) da += da_nets[(size_t)e * rows * ad + i];
  }
  pin(t);
  pin(lsr);
  pin(ev);
  const float du = da * max_a * (1.0f - t * t);
  const float ls = fminf(fmaxf(lsr, kLogStdMin), kLogStdMax);
  const bool inside = lsr >= kLogStdMin && lsr <= kLogStdMax;
  dhead[(size_t)r * 2 * ad + j] = du;
  dhead[(size_t)r * 2 * ad + ad + j] = inside ? du * ev * expf(ls) : 0.f;
}

__global__ void gauss_ood_kernel(const float* __restrict__ head, const float* __restrict__ eps, int n_samples,
                                 int rows, int ad, float* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const int64_t i = (int64_t)blockIdx.x * kEw + threadIdx.x;
  const int64_t total = (int64_t)n_samples * rows * ad;
  if (i >= total) return;
  const int k = (int)(i % ad);
  const int b = (int)((i / ad) % rows);
  const float mu = head[(size_t)b * 2 * ad + k];
  const float ls = fminf(fmaxf(head[(size_t)b * 2 * ad + ad + k], kLogStdMin), kLogStdMax);
  out[i] = mu + expf(ls) * eps[i];
}

// ---------------- VAE tails ----------------
__global__ void vae_latent_kernel(const float* __restrict__ head, const float* __restrict__ eps, int rows, int L,
                                  float* __restrict__ z) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const int i = blockIdx.x * kEw + threadIdx.x;
  if (i >= rows * L) return;
  const int r = i / L, k = i - r * L;
  const float mean = head[(size_t)r * 2 * L + k];
  const float ls = fminf(fmaxf(head[(size_t)r * 2 * L + L + k], kVaeLsMin), kVaeLsMax);
  z[i] = mean + expf(ls) * eps[i];
}

__device__ __forceinline__ float kl_elem(float mean, float ls_raw) {
  const float sd = expf(fminf(fmaxf(ls_raw, kVaeLsMin), kVaeLsMax));
  return -0.5f * (1.0f + logf(sd * sd) - mean * mean - sd * sd);
}

// (body + two entry kernels: by value, and "_p" = arguments in a device-resident block, csrc/argmem.h -- the three
// batch-sum kernels of the CPQ step whose 16-32 dwords of arguments exceed what kernarg preload hands a wave)
__device__ __forceinline__ void vae_loss_body(const float* __restrict__ u, const float* __restrict__ act,
                                              const float* __restrict__ head, int rows, int ad, int L, float beta,
                                              float inv_rows, float* __restrict__ du, float* __restrict__ stat) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  __shared__ float sm[20];
  float rec = 0.f, kl = 0.f;
  const float ia = inv_rows / (float)ad, il = inv_rows / (float)L;
  // four strides of the workgroup per pass, their loads in flight together (one element per pass = one memory round
  // trip per pass: 4 + 16 of them at C2's 2048 x (2 + 8)); per-thread accumulation order unchanged
  const int n_rec = rows * ad, n_kl = rows * L;
  for (int i0 = threadIdx.x; i0 < n_rec; i0 += 4 * kRed) {
    float uv[4], av[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * kRed;
      uv[k] = u[i < n_rec ? i : 0];
      av[k] = act[i < n_rec ? i : 0];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pin(uv[k]);
      pin(av[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * kRed;
      if (i < n_rec) {
        const float d = uv[k] - av[k];
        rec += d * d;
        du[i] = 2.0f * d * ia;
      }
    }
  }
  for (int i0 = threadIdx.x; i0 < n_kl; i0 += 4 * kRed) {
    float mv[4], lv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + k * kRed < n_kl ? i0 + k * kRed : 0;
      const int r = i / L, c = i - r * L;
      mv[k] = head[(size_t)r * 2 * L + c];
      lv[k] = head[(size_t)r * 2 * L + L + c];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pin(mv[k]);
      pin(lv[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k * kRed < n_kl) kl += kl_elem(mv[k], lv[k]);
  }
  rec = block_sum(rec, sm);
  kl = block_sum(kl, sm);
  if (threadIdx.x == 0 && stat) stat[0] = rec * ia + beta * (kl * il);
}
__global__ __launch_bounds__(kRed) void vae_loss_kernel(const float* __restrict__ u, const float* __restrict__ act,
                                                        const float* __restrict__ head, int rows, int ad, int L,
                                                        float beta, float inv_rows, float* __restrict__ du,
                                                        float* __restrict__ stat) {
  vae_loss_body(u, act, head, rows, ad, L, beta, inv_rows, du, stat);
}
struct VaeLossArgs {
  const float *u, *act, *head;
  float *du, *stat;
  int32_t rows, ad, L;
  float beta, inv_rows;
};
__global__ __launch_bounds__(kRed) void vae_loss_kernel_p(const void* p) {
  const OSRL_CAS VaeLossArgs& a = *(const OSRL_CAS VaeLossArgs*)p;
  vae_loss_body(a.u, a.act, a.head, a.rows, a.ad, a.L, a.beta, a.inv_rows, a.du, a.stat);
}

__global__ void vae_latent_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                      const float* __restrict__ dz, int rows, int L, float beta, float inv_rows,
                                      float* __restrict__ dhead) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const int i = blockIdx.x * kEw + threadIdx.x;
  if (i >= rows * L) return;
  const int r = i / L, k = i - r * L;
  const float mean = head[(size_t)r * 2 * L + k];
  const float lsr = head[(size_t)r * 2 * L + L + k];
  const float g = dz[i];
  float ev = eps[i];
  pin(ev);
  const float sd = expf(fminf(fmaxf(lsr, kVaeLsMin), kVaeLsMax));
  const float c = beta * inv_rows / (float)L;
  dhead[(size_t)r * 2 * L + k] = g + c * mean;
  const bool inside = lsr >= kVaeLsMin && lsr <= kVaeLsMax;
  dhead[(size_t)r * 2 * L + L + k] = inside ? (g * ev + c * (sd - 1.0f / sd)) * sd : 0.f;
}

__global__ void vae_kl_rows_kernel(const float* __restrict__ head, int rows, int L, float* __restrict__ kl) {
  const int r = blockIdx.x * kEw + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  const float* __restrict__ hr = head + (size_t)r * 2 * L;
  for (int k0 = 0; k0 < L; k0 += 4) {  // four latent dimensions per pass, their 8 loads in flight together
    float mv[4], lv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + u < L ? k0 + u : L - 1;
      mv[u] = hr[k];
      lv[u] = hr[L + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pin(mv[u]);
      pin(lv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + u < L) s += kl_elem(mv[u], lv[u]);
  }
  kl[r] = s / (float)L;
}

// ---------------- exact quantile: 4-pass 8-bit radix select in one workgroup ----------------
__device__ __forceinline__ uint32_t f2key(float f) {  // order-preserving float -> uint
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// Histogram add with wave-level aggregation of equal bins (per-lane LDS atomics on one address serialise).
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t bin, bool active) {
  // one aggregation round for the first active lane's bin (the high bytes of real-valued keys sit in one or two
  // bins), plain LDS atomics for the rest; more rounds measured slower (30 vs 27 us at n = 20480)
  const unsigned long long act = __ballot(active);
  if (act == 0) return;
  const int lead = __ffsll((long long)act) - 1;
  const uint32_t lb = __shfl(bin, lead);
  const bool same = active && bin == lb;
  const unsigned long long m = __ballot(same);
  if ((int)(threadIdx.x & 63) == lead) atomicAdd(&hist[lb], (uint32_t)__popcll(m));
  if (active && !same) atomicAdd(&hist[bin], 1u);
}

// k-th smallest (0-based) key among x[0..n), STREAMED from global (any n; the register-resident form for
// n <= 32 * 1024 is bit_select below); every thread returns it.  hist: 256 uints in LDS.
// bc[2] returns the number of elements <= the selected key (for the interpolation partner).
__device__ uint32_t radix_select(const float* __restrict__ x, int64_t n, int64_t k, uint32_t* hist,
                                 uint32_t* bc /*4 uints*/) {
  uint32_t prefix = 0, mask = 0;
  int64_t below = 0;  // elements strictly below the current prefix range
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += kRed) hist[i] = 0;
    __syncthreads();
    const int nround = (int)((n + kRed - 1) / kRed * kRed);
    for (int i = threadIdx.x; i < nround; i += kRed) {
      uint32_t key = 0;
      bool act = false;
      if (i < n) {
        key = f2key(x[i]);
        act = (key & mask) == prefix;
      }
      hist_add(hist, (key >> shift) & 255u, act);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      // wave-parallel bin search: lane l owns bins 4l..4l+3; shuffle prefix scan over the 64 lanes
      const int l = threadIdx.x;
      const uint32_t c0 = hist[4 * l], c1 = hist[4 * l + 1], c2 = hist[4 * l + 2], c3 = hist[4 * l + 3];
      const uint32_t sl = c0 + c1 + c2 + c3;
      uint32_t incl = sl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (l >= o) incl += t;
      }
      const uint32_t excl = incl - sl;
      const uint32_t kk0 = (uint32_t)k;
      if (kk0 >= excl && kk0 < incl) {  // exactly one lane
        uint32_t kk = kk0 - excl, d = 4 * l, below_d = excl, cd = c0;
        if (kk >= c0) {
          kk -= c0; below_d += c0; d++; cd = c1;
          if (kk >= c1) {
            kk -= c1; below_d += c1; d++; cd = c2;
            if (kk >= c2) { kk -= c2; below_d += c2; d++; cd = c3; }
          }
        }
        bc[0] = d;
        bc[1] = kk;
        bc[2] = below_d;  // elements in lower bins of this pass
        bc[3] = cd;       // elements in the chosen bin
      }
    }
    __syncthreads();
    prefix |= bc[0] << shift;
    mask |= 255u << shift;
    k = bc[1];
    below += bc[2];
    if (shift == 0) bc[2] = (uint32_t)(below + bc[3]);  // count of elements <= selected key
    __syncthreads();
  }
  return prefix;
}

// k-th smallest (0-based) of REGISTER-resident keys (element j*blockDim + tid in kreg[j]; slots past n hold
// 0xffffffff): the answer is built from the top bit down, one block-wide count of "key < candidate" per bit.  A
// round is ROWS compares whose lane masks are counted on the scalar unit (v_cmp -> s_bcnt1: no cross-lane traffic),
// one LDS atomic per wave into that round's own counter, ONE barrier, one broadcast read.  The histogram form spends
// its time in same-address LDS atomics because the KL values of one batch share their high bytes (27 us at
// n = 20480; this form: profiles/r2_kbench.txt).
// cnt: 34 uints of LDS, zeroed, with a barrier between the zeroing and this call.  *n_le = #keys <= the result.
template <int ROWS>
__device__ __forceinline__ uint32_t bit_select(const uint32_t (&kreg)[ROWS], int64_t k, uint32_t* cnt,
                                               uint32_t* n_le) {
  const bool lead = (threadIdx.x & 63) == 0;
  uint32_t prefix = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t t = prefix | (1u << bit);
    uint32_t c = 0;  // wave-uniform
#pragma unroll
    for (int j = 0; j < ROWS; ++j) c += (uint32_t)__popcll(__ballot(kreg[j] < t));
    if (lead) atomicAdd(&cnt[bit], c);
    __syncthreads();
    if ((int64_t)cnt[bit] <= k) prefix = t;  // largest v with #(keys < v) <= k  ==  the k-th smallest key
  }
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < ROWS; ++j) c += (uint32_t)__popcll(__ballot(kreg[j] <= prefix));
  if (lead) atomicAdd(&cnt[32], c);
  __syncthreads();
  *n_le = cnt[32];
  return prefix;
}

// torch.quantile(x, q) ('linear') of n <= ROWS * blockDim values, keys in registers; valid in every thread.
// cnt: 34 uints, s_min: 17 uints of LDS.
template <int ROWS>
__device__ __forceinline__ float quantile_rows(const float* __restrict__ x, int64_t n, float q, uint32_t* cnt,
                                               uint32_t* s_min) {
  const double pos = (double)q * (double)(n - 1);
  const int64_t lo = (int64_t)floor(pos);
  const int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
  const float w = (float)(pos - (double)lo);
  if (threadIdx.x < 34) cnt[threadIdx.x] = 0;
  uint32_t kreg[ROWS];
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    // all loads in flight at once: clamped address + select.  ("i < n ? f2key(x[i]) : ~0u" compiles to an exec-masked
    // load with its own s_waitcnt vmcnt(0) per row: ROWS serial round trips, most of this kernel's time.)
    const int64_t i = (int64_t)j * kRed + threadIdx.x;
    const bool ok = i < n;
    const float xv = x[ok ? i : 0];
    kreg[j] = f2key(xv) | (ok ? 0u : 0xffffffffu);  // (an OR, not a select: a select is turned back into a branch)
  }
  __syncthreads();
  uint32_t n_le;
  const uint32_t klo = bit_select<ROWS>(kreg, lo, cnt, &n_le);
  uint32_t khi = klo;
  if (hi != lo && (int64_t)n_le < hi + 1) {  // the next order statistic is the smallest key strictly above klo
    uint32_t mn = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const int64_t i = (int64_t)j * kRed + threadIdx.x;
      if (i < n && kreg[j] > klo && kreg[j] < mn) mn = kreg[j];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t other = __shfl_xor(mn, o);
      mn = other < mn ? other : mn;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mn;
    __syncthreads();
    uint32_t t = s_min[0];
    for (int i = 1; i < (int)(kRed >> 6); ++i) t = s_min[i] < t ? s_min[i] : t;
    khi = t;
  }
  const float vlo = key2f(klo), vhi = key2f(khi);
  return vlo + (vhi - vlo) * w;
}
constexpr int kQuantRegRows = 32;
__device__ __forceinline__ float quantile_regs(const float* __restrict__ x, int64_t n, float q, uint32_t* cnt,
                                               uint32_t* s_min) {
  const int rows = (int)((n + kRed - 1) / kRed);  // uniform
  if (rows <= 4) return quantile_rows<4>(x, n, q, cnt, s_min);
  if (rows <= 12) return quantile_rows<12>(x, n, q, cnt, s_min);
  if (rows <= 20) return quantile_rows<20>(x, n, q, cnt, s_min);
  return quantile_rows<kQuantRegRows>(x, n, q, cnt, s_min);
}

__global__ __launch_bounds__(kRed) void quantile_kernel(const float* __restrict__ x, int64_t n, float q,
                                                        float* __restrict__ out) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t bc[4];
  __shared__ uint32_t s_min[17];
  if (n <= (int64_t)32 * kRed) {  // register-resident keys, bitwise select
    const float v = quantile_regs(x, n, q, hist, s_min);
    if (threadIdx.x == 0) out[0] = v;
    return;
  }
  // torch.quantile 'linear': pos = q*(n-1); lo=floor(pos); result = x_lo + (x_hi-x_lo)*(pos-lo)
  const double pos = (double)q * (double)(n - 1);
  const int64_t lo = (int64_t)floor(pos);
  const int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
  const float w = (float)(pos - (double)lo);
  const uint32_t klo = radix_select(x, n, lo, hist, bc);
  const int64_t n_le = bc[2];
  uint32_t khi = klo;
  if (hi != lo && n_le < hi + 1) {
    // the (lo+1)-th order statistic is the smallest key strictly above klo: one min-reduction pass
    uint32_t mn = 0xffffffffu;
    for (int64_t i = threadIdx.x; i < n; i += kRed) {
      const uint32_t key = f2key(x[i]);
      if (key > klo && key < mn) mn = key;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const uint32_t other = __shfl_xor(mn, o);
      mn = other < mn ? other : mn;
    }
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t t = s_min[0];
      for (int i = 1; i < (int)(kRed >> 6); ++i) t = s_min[i] < t ? s_min[i] : t;
      s_min[0] = t;
    }
    __syncthreads();
    khi = s_min[0];
  }
  if (threadIdx.x == 0) {
    const float vlo = key2f(klo), vhi = key2f(khi);
    out[0] = vlo + (vhi - vlo) * w;
  }
}

// ---------------- exact quantile of MANY values: multi-workgroup radix select ----------------
// Data-parallel CPQ takes the 0.75-quantile over the all-gathered KL rows (world * N * B = 163840 values at 8 ranks);
// the single-workgroup streaming select above needs 200 us for that, on the step's critical chain.  Here every pass
// is a grid: workgroups histogram one key byte of their slice in LDS (wave-aggregated adds, then one global atomic per
// non-empty bin -- integer adds, order-independent: deterministic), and the NEXT launch walks the global histograms of
// the passes before it (256 bins each, redundantly per workgroup) to know its prefix.  Workspace (uint32): 4 x 256
// histogram bins, then succ (atomicMin target) at [1024]; the last launch leaves it cleared for the next call.
constexpr int kQselWs = 4 * 256 + 8;

// digits of passes 0..npass-1 of the k-th smallest key, from the global histograms: prefix / rank inside the prefix /
// number of keys below the prefix range / size of the last chosen bin.  One wave (64 threads) per call.
struct QselWalk {
  uint32_t prefix, k, below, cnt;
};
__device__ QselWalk qsel_walk(const uint32_t* __restrict__ ws, uint32_t k, int npass, uint32_t* sh /*>= 8 uints*/) {
  // executed by threads 0..63 of the workgroup; result broadcast through sh[0..3]
  if (threadIdx.x < 64) {
    const int l = threadIdx.x;
    uint32_t prefix = 0, below = 0, cnt = 0;
    for (int p = 0; p < npass; ++p) {
      const uint32_t* h = ws + p * 256;
      const uint32_t c0 = h[4 * l], c1 = h[4 * l + 1], c2 = h[4 * l + 2], c3 = h[4 * l + 3];
      const uint32_t sl = c0 + c1 + c2 + c3;
      uint32_t incl = sl;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (l >= o) incl += t;
      }
      const uint32_t excl = incl - sl;
      uint32_t d = 0, kk = 0, bd = 0, cd = 0;
      const bool mine = k >= excl && k < incl;  // exactly one lane
      if (mine) {
        kk = k - excl; d = 4 * l; bd = excl; cd = c0;
        if (kk >= c0) {
          kk -= c0; bd += c0; d++; cd = c1;
          if (kk >= c1) {
            kk -= c1; bd += c1; d++; cd = c2;
            if (kk >= c2) { kk -= c2; bd += c2; d++; cd = c3; }
          }
        }
      }
      const unsigned long long who = __ballot(mine);
      const int src = __ffsll((long long)who) - 1;
      d = __shfl(d, src); kk = __shfl(kk, src); bd = __shfl(bd, src); cd = __shfl(cd, src);
      prefix |= d << (24 - 8 * p);
      k = kk;
      below += bd;
      cnt = cd;
    }
    if (l == 0) { sh[0] = prefix; sh[1] = k; sh[2] = below; sh[3] = cnt; }
  }
  __syncthreads();
  return QselWalk{sh[0], sh[1], sh[2], sh[3]};
}

__global__ __launch_bounds__(1024) void qsel_hist_kernel(const float* __restrict__ x, int64_t n, int64_t k, int pass,
                                                         uint32_t* __restrict__ ws) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh[8];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
  if (pass == 0 && blockIdx.x == 0 && threadIdx.x == 0) ws[1024] = 0xffffffffu;  // successor search starts empty
  const QselWalk w = qsel_walk(ws, (uint32_t)k, pass, sh);  // (its barrier also covers the zeroing above)
  const int shift = 24 - 8 * pass;
  const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nround = (n + stride - 1) / stride * stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
    uint32_t key = 0;
    bool act = false;
    if (i < n) {
      key = f2key(x[i]);
      act = (key & mask) == w.prefix;
    }
    hist_add(hist, (key >> shift) & 255u, act);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x)
    if (hist[i]) atomicAdd(&ws[pass * 256 + i], hist[i]);
}

// the (lo+1)-th order statistic when it is not a duplicate of the lo-th: the smallest key strictly above it
__global__ __launch_bounds__(1024) void qsel_succ_kernel(const float* __restrict__ x, int64_t n, int64_t lo, int64_t hi,
                                                         uint32_t* __restrict__ ws) {
  __shared__ uint32_t sh[8];
  const QselWalk w = qsel_walk(ws, (uint32_t)lo, 4, sh);
  const int64_t n_le = (int64_t)w.below + w.cnt;  // keys <= the selected one
  if (hi == lo || n_le >= hi + 1) return;         // the partner is the same key
  uint32_t mn = 0xffffffffu;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t key = f2key(x[i]);
    if (key > w.prefix && key < mn) mn = key;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint32_t other = __shfl_xor(mn, o);
    mn = other < mn ? other : mn;
  }
  if ((threadIdx.x & 63) == 0 && mn != 0xffffffffu) atomicMin(&ws[1024], mn);
}

__global__ __launch_bounds__(256) void qsel_finish_kernel(int64_t n, float q, uint32_t* __restrict__ ws,
                                                          float* __restrict__ out) {
  __shared__ uint32_t sh[8];
  const double pos = (double)q * (double)(n - 1);
  const int64_t lo = (int64_t)floor(pos);
  const int64_t hi = lo + 1 < n ? lo + 1 : n - 1;
  const float wgt = (float)(pos - (double)lo);
  const QselWalk w = qsel_walk(ws, (uint32_t)lo, 4, sh);
  const int64_t n_le = (int64_t)w.below + w.cnt;
  const uint32_t klo = w.prefix;
  const uint32_t khi = (hi == lo || n_le >= hi + 1) ? klo : ws[1024];
  __syncthreads();  // everyone has read the histograms and the successor
  if (threadIdx.x == 0) {
    const float vlo = key2f(klo), vhi = key2f(khi);
    out[0] = vlo + (vhi - vlo) * wgt;
  }
  for (int i = threadIdx.x; i < 4 * 256; i += blockDim.x) ws[i] = 0;  // ready for the next call
}

// ---------------- CPQ ----------------
// Ensemble outputs are [n][stride].  SMALL (n <= kEns, host-checked: every reference config) requests all members of
// an element together as straight-line code -- members past n re-read member 0 and are dropped by a select -- so the
// loads of one batch row (targets, online nets, reward, done) overlap; the counted loops of the general form wait for
// each member in turn (cpq_critic_loss: 8 serial round trips per row before).  Same fminf / accumulation order, same
// bits.
constexpr int kEns = 4;
template <bool SMALL>
__device__ __forceinline__ void load_members(const float* __restrict__ q, int n, int stride, int i, float (&v)[kEns]) {
  static_assert(SMALL, "general form: counted loops at the call site");
#pragma unroll
  for (int e = 0; e < kEns; ++e) v[e] = q[(size_t)(e < n ? e : 0) * stride + i];
}
template <bool SMALL = false>
__device__ __forceinline__ float min_over(const float* __restrict__ q, int n, int stride, int i) {
  if constexpr (SMALL) {
    float v[kEns];
    load_members<true>(q, n, stride, i, v);
    float m = v[0];
#pragma unroll
    for (int e = 1; e < kEns; ++e) m = e < n ? fminf(m, v[e]) : m;
    return m;
  } else {
    float v = q[i];
    for (int e = 1; e < n; ++e) v = fminf(v, q[(size_t)e * stride + i]);
    return v;
  }
}

template <bool SMALL>
__device__ __forceinline__ void cpq_critic_loss_body(const float* __restrict__ q_old, int n_q_old,
                                                     const float* __restrict__ qc_old, int n_qc_old,
                                                     const float* __restrict__ q, int n_q,
                                                     const float* __restrict__ rew, const float* __restrict__ done,
                                                     int rows, float gamma, float q_thres, float inv_rows,
                                                     float* __restrict__ dq, float* __restrict__ stat) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  __shared__ float sm[20];
  float loss = 0.f;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    const float qt = min_over<SMALL>(q_old, n_q_old, rows, b);
    const float qct = min_over<SMALL>(qc_old, n_qc_old, rows, b);
    // backup = r + gamma*(1-done)*(qc_targ<=q_thres)*q_targ      cpq.py:145-146
    if constexpr (SMALL) {
      float qv[kEns];
      load_members<true>(q, n_q, rows, b, qv);
      const float backup = rew[b] + gamma * (1.0f - done[b]) * (qct <= q_thres ? 1.0f : 0.0f) * qt;
#pragma unroll
      for (int e = 0; e < kEns; ++e)
        if (e < n_q) {
          const float d = qv[e] - backup;
          loss += d * d;
          dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
        }
    } else {
      const float backup = rew[b] + gamma * (1.0f - done[b]) * (qct <= q_thres ? 1.0f : 0.0f) * qt;
      for (int e = 0; e < n_q; ++e) {
        const float d = q[(size_t)e * rows + b] - backup;
        loss += d * d;
        dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
      }
    }
  }
  loss = block_sum(loss, sm);
  if (threadIdx.x == 0 && stat) stat[0] = loss * inv_rows;
}
template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_critic_loss_kernel(const float* __restrict__ q_old, int n_q_old,
                                                               const float* __restrict__ qc_old, int n_qc_old,
                                                               const float* __restrict__ q, int n_q,
                                                               const float* __restrict__ rew,
                                                               const float* __restrict__ done, int rows,
                                                               float gamma, float q_thres, float inv_rows,
                                                               float* __restrict__ dq, float* __restrict__ stat) {
  cpq_critic_loss_body<SMALL>(q_old, n_q_old, qc_old, n_qc_old, q, n_q, rew, done, rows, gamma, q_thres, inv_rows, dq, stat);
}
struct CriticLossArgs {
  const float *q_old, *qc_old, *q, *rew, *done;
  float *dq, *stat;
  int32_t n_q_old, n_qc_old, n_q, rows;
  float gamma, q_thres, inv_rows;
};
template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_critic_loss_kernel_p(const void* p) {
  const OSRL_CAS CriticLossArgs& a = *(const OSRL_CAS CriticLossArgs*)p;
  cpq_critic_loss_body<SMALL>(a.q_old, a.n_q_old, a.qc_old, a.n_qc_old, a.q, a.n_q, a.rew, a.done, a.rows, a.gamma,
                              a.q_thres, a.inv_rows, a.dq, a.stat);
}

// mean over the (global) batch of qc_ood = ((KL >= quantile) * qc_sampled).mean(0)   cpq.py:184,187
template <bool SMALL>
__device__ __forceinline__ float cpq_ood_mean_block(const float* __restrict__ qc_sampled, int n_qc_old,
                                                    const float* __restrict__ kl, const float quant,
                                                    int n_samples, int rows, float inv_rows, float* sm) {
  float ood = 0.f;
  const int nr = n_samples * rows;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    float s = 0.f;
    // unconditional loads (no branch on kl) so the N sample loads of a row are all in flight at once
#pragma unroll 5
    for (int j = 0; j < n_samples; ++j) {
      const int i = j * rows + b;
      const float v = min_over<SMALL>(qc_sampled, n_qc_old, nr, i);
      s += kl[i] >= quant ? v : 0.f;
    }
    ood += s / (float)n_samples;
  }
  return block_sum(ood, sm) * inv_rows;
}

template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_ood_mean_kernel(const float* __restrict__ qc_sampled, int n_qc_old,
                                                            const float* __restrict__ kl,
                                                            const float* __restrict__ quantile, int n_samples,
                                                            int rows, float inv_rows, float* __restrict__ out) {
  __shared__ float sm[20];
  const float ood = cpq_ood_mean_block<SMALL>(qc_sampled, n_qc_old, kl, quantile[0], n_samples, rows, inv_rows, sm);
  if (threadIdx.x == 0) out[0] = ood;
}

// quantile + OOD mean in one launch (single-GPU step, n_samples * rows <= 32 * 1024): the KL rows are read once into
// registers for the select, the masked mean re-reads them from L2 in the order of cpq_ood_mean_kernel (same bits)
template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_ood_stat_kernel(const float* __restrict__ qc_sampled, int n_qc_old,
                                                            const float* __restrict__ kl, float q, int n_samples,
                                                            int rows, float inv_rows, float* __restrict__ quant_out,
                                                            float* __restrict__ out) {
  __shared__ uint32_t cnt[34];
  __shared__ uint32_t s_min[17];
  __shared__ float sm[20];
  OSRL_TRACE_BEGIN(12, kl);
  const float quant = quantile_regs(kl, (int64_t)n_samples * rows, q, cnt, s_min);
  const float ood = cpq_ood_mean_block<SMALL>(qc_sampled, n_qc_old, kl, quant, n_samples, rows, inv_rows, sm);
  if (threadIdx.x == 0) {
    quant_out[0] = quant;
    out[0] = ood;
  }
}

// The OOD rows as a SET (osrl_amd.h osrl_cpq_ood_select / _sum): quantile, then list = ascending indices with kl >= quantile.
// Every thread owns a contiguous run of <= 32 indices; block-wide exclusive scan of the per-thread counts (wave scans by
// shuffles, the 16 wave totals through LDS).
__global__ __launch_bounds__(kRed) void cpq_ood_select_kernel(const float* __restrict__ kl, const float* __restrict__ quantile_in,
                                                              float q, int n, float* __restrict__ quant_out,
                                                              int32_t* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ uint32_t cnt[34];
  __shared__ uint32_t s_min[17];
  __shared__ uint32_t s_tot[kRed / 64 + 1];
  OSRL_TRACE_BEGIN(14, list);
  const float quant = quantile_in ? quantile_in[0] : quantile_regs(kl, (int64_t)n, q, cnt, s_min);
  const int per = (n + kRed - 1) / kRed;  // <= 32 (host-checked)
  const int i0 = threadIdx.x * per;
  uint32_t bits = 0;
  for (int k = 0; k < per; ++k) {
    const int i = i0 + k;
    if (i < n && kl[i] >= quant) bits |= 1u << k;
  }
  const uint32_t c = __builtin_popcount(bits);
  uint32_t incl = c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();  // (quantile_regs is done with its scratch; s_tot is ours)
  if (lane == 63) s_tot[wave] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < kRed / 64; ++w) {
    const uint32_t t = s_tot[w];
    if (w < wave) base += t;
    total += t;
  }
  uint32_t pos = base + incl - c;
  for (int k = 0; k < per; ++k)
    if (bits & (1u << k)) list[pos++] = i0 + k;
  if (threadIdx.x == 0) {
    count[0] = (int32_t)total;
    if (quant_out) quant_out[0] = quant;
  }
}

__global__ __launch_bounds__(kRed) void cpq_ood_sum_kernel(const float* __restrict__ qc_sel, int n_qc, int cap,
                                                           const int32_t* __restrict__ count, float scale,
                                                           float* __restrict__ out) {
  __shared__ float sm[20];
  OSRL_TRACE_BEGIN(15, out);
  int n = count[0];
  n = n < cap ? n : cap;
  float s = 0.f;
  for (int r = threadIdx.x; r < n; r += kRed) s += min_over(qc_sel, n_qc, cap, r);
  s = block_sum(s, sm);
  if (threadIdx.x == 0) out[0] = s * scale;
}

struct OodArgs {  // non-NULL qc_sampled: compute the OOD mean here (single-GPU step: one launch less)
  const float* qc_sampled;
  const float* kl;
  const float* quantile;
  int32_t n_qc_old, n_samples;
};

template <bool SMALL>
__device__ __forceinline__ void cpq_cost_loss_body(
    const float* __restrict__ qc_old_next, int n_qc_old, const float* __restrict__ qc, int n_qc,
    float* __restrict__ ood_mean_p, const float* __restrict__ cost, int rows, float gamma, float qc_thres,
    float alpha_lr, float inv_rows, float stat_share, float* __restrict__ log_alpha, float* __restrict__ dq,
    float* __restrict__ stat, const OodArgs oa) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  __shared__ float sm[20];
  float ood_here = 0.f;
  if (oa.qc_sampled)
    ood_here =
        cpq_ood_mean_block<SMALL>(oa.qc_sampled, oa.n_qc_old, oa.kl, oa.quantile[0], oa.n_samples, rows, inv_rows, sm);
  float loss = 0.f;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    if constexpr (SMALL) {
      float qv[kEns];
      load_members<true>(qc, n_qc, rows, b, qv);
      const float backup = cost[b] + gamma * min_over<true>(qc_old_next, n_qc_old, rows, b);  // cpq.py:161
#pragma unroll
      for (int e = 0; e < kEns; ++e)
        if (e < n_qc) {
          const float d = qv[e] - backup;
          loss += d * d;
          dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
        }
    } else {
      const float backup = cost[b] + gamma * min_over(qc_old_next, n_qc_old, rows, b);  // cpq.py:161
      for (int e = 0; e < n_qc; ++e) {
        const float d = qc[(size_t)e * rows + b] - backup;
        loss += d * d;
        dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
      }
    }
  }
  loss = block_sum(loss, sm);
  if (threadIdx.x == 0 && !oa.qc_sampled && !ood_mean_p) {  // dual step deferred (osrl_cpq_alpha_step)
    if (stat) stat[0] = loss * inv_rows;
  } else if (threadIdx.x == 0) {
    const float ood_mean = oa.qc_sampled ? ood_here : ood_mean_p[0];
    if (oa.qc_sampled) ood_mean_p[0] = ood_here;
    float la = log_alpha[0];
    const float ea = expf(la);
    // cpq.py:186-187; under data parallelism the mse part is this rank's partial sum and the global
    // terms are pre-divided by the world size (stat_share) so that an all-reduce(SUM) restores them
    if (stat) stat[0] = loss * inv_rows - stat_share * ea * (ood_mean - qc_thres);
    la += alpha_lr * ea * (qc_thres - ood_mean);  // cpq.py:193-194
    la = fminf(fmaxf(la, -5.0f), 5.0f);
    log_alpha[0] = la;
    if (stat) stat[1] = stat_share * expf(la);
  }
}
template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_cost_loss_kernel(
    const float* __restrict__ qc_old_next, int n_qc_old, const float* __restrict__ qc, int n_qc,
    float* __restrict__ ood_mean_p, const float* __restrict__ cost, int rows, float gamma, float qc_thres,
    float alpha_lr, float inv_rows, float stat_share, float* __restrict__ log_alpha, float* __restrict__ dq,
    float* __restrict__ stat, const OodArgs oa) {
  cpq_cost_loss_body<SMALL>(qc_old_next, n_qc_old, qc, n_qc, ood_mean_p, cost, rows, gamma, qc_thres, alpha_lr, inv_rows,
                            stat_share, log_alpha, dq, stat, oa);
}
struct CostLossArgs {
  const float *qc_old_next, *qc, *cost;
  float *ood_mean_p, *log_alpha, *dq, *stat;
  OodArgs oa;
  int32_t n_qc_old, n_qc, rows;
  float gamma, qc_thres, alpha_lr, inv_rows, stat_share;
};
template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_cost_loss_kernel_p(const void* p) {
  const OSRL_CAS CostLossArgs& a = *(const OSRL_CAS CostLossArgs*)p;
  const OodArgs oa{a.oa.qc_sampled, a.oa.kl, a.oa.quantile, a.oa.n_qc_old, a.oa.n_samples};
  cpq_cost_loss_body<SMALL>(a.qc_old_next, a.n_qc_old, a.qc, a.n_qc, a.ood_mean_p, a.cost, a.rows, a.gamma, a.qc_thres,
                            a.alpha_lr, a.inv_rows, a.stat_share, a.log_alpha, a.dq, a.stat, oa);
}

// the dual step of cpq.py:186-195 on its own: stat[0] (the MSE part written by cpq_cost_loss_kernel) gets the
// -exp(log_alpha)*(ood_mean - thres) term, log_alpha ascends and is clamped, stat[1] = exp(log_alpha)
__global__ void cpq_alpha_step_kernel(const float* __restrict__ ood_mean, float qc_thres, float alpha_lr,
                                      float stat_share, float* __restrict__ log_alpha, float* __restrict__ stat) {
  OSRL_TRACE_BEGIN(13, ood_mean);
  float la = log_alpha[0];
  const float ea = expf(la);
  if (stat) stat[0] -= stat_share * ea * (ood_mean[0] - qc_thres);
  la += alpha_lr * ea * (qc_thres - ood_mean[0]);
  la = fminf(fmaxf(la, -5.0f), 5.0f);
  log_alpha[0] = la;
  if (stat) stat[1] = stat_share * expf(la);
}

template <bool SMALL>
__global__ __launch_bounds__(kRed) void cpq_actor_loss_kernel(const float* __restrict__ q, int n_q,
                                                              const float* __restrict__ qc, int n_qc, int rows,
                                                              float q_thres, float inv_rows,
                                                              float* __restrict__ dq, float* __restrict__ stat) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  __shared__ float sm[20];
  float loss = 0.f;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    int am = 0;
    float qm;
    if constexpr (SMALL) {
      float qv[kEns];
      load_members<true>(q, n_q, rows, b, qv);
      qm = qv[0];
#pragma unroll
      for (int e = 1; e < kEns; ++e) {
        const bool lt = e < n_q && qv[e] < qm;
        qm = lt ? qv[e] : qm;
        am = lt ? e : am;
      }
    } else {
      qm = q[b];
      for (int e = 1; e < n_q; ++e) {
        const float v = q[(size_t)e * rows + b];
        if (v < qm) { qm = v; am = e; }
      }
    }
    const float mask = min_over<SMALL>(qc, n_qc, rows, b) <= q_thres ? 1.0f : 0.0f;
    loss -= mask * qm;
    for (int e = 0; e < n_q; ++e) dq[(size_t)e * rows + b] = e == am ? -mask * inv_rows : 0.f;
  }
  loss = block_sum(loss, sm);
  if (threadIdx.x == 0 && stat) stat[0] = loss * inv_rows;
}

__global__ __launch_bounds__(kRed) void mse_loss_kernel(const float* __restrict__ u, const float* __restrict__ act,
                                                        int n, float inv_n, float* __restrict__ du,
                                                        float* __restrict__ stat) {
  __shared__ float sm[20];
  float loss = 0.f;
  for (int i = threadIdx.x; i < n; i += kRed) {
    const float d = u[i] - act[i];
    loss += d * d;
    du[i] = 2.0f * d * inv_n;
  }
  loss = block_sum(loss, sm);
  if (threadIdx.x == 0 && stat) stat[0] = loss * inv_n;
}

// ---------------- BCQ-Lag ----------------
__global__ void clamp_kernel(float* __restrict__ x, int64_t n, float lo, float hi) {
  const int64_t i = (int64_t)blockIdx.x * kEw + threadIdx.x;
  if (i < n) x[i] = fminf(fmaxf(x[i], lo), hi);
}

__global__ void bcq_perturb_kernel(const float* __restrict__ dec, const float* __restrict__ t, int n, float phi,
                                   float max_a, float* __restrict__ a) {
  const int i = blockIdx.x * kEw + threadIdx.x;
  if (i >= n) return;
  a[i] = fminf(fmaxf(dec[i] + phi * max_a * t[i], -max_a), max_a);  // net.py:61-62
}

__global__ void bcq_perturb_bwd_kernel(const float* __restrict__ dec, const float* __restrict__ t,
                                       const float* __restrict__ da_nets, int n_nets, int n, float phi, float max_a,
                                       float* __restrict__ dt) {
  const int i = blockIdx.x * kEw + threadIdx.x;
  if (i >= n) return;
  float da = 0.f;
  for (int e = 0; e < n_nets; ++e) da += da_nets[(size_t)e * n + i];
  const float pre = dec[i] + phi * max_a * t[i];
  dt[i] = (pre >= -max_a && pre <= max_a) ? da * phi * max_a : 0.f;
}

__global__ __launch_bounds__(kRed) void bcq_critic_loss_kernel(const float* __restrict__ q_t, int n1, int n2,
                                                               int n_samples, const float* __restrict__ q_on,
                                                               int n_on, const float* __restrict__ base,
                                                               const float* __restrict__ done, int rows,
                                                               float gamma, float lmbda, float inv_rows,
                                                               float* __restrict__ dq, float* __restrict__ stat) {
  __shared__ float sm[20];
  float loss = 0.f;
  const int nr = rows * n_samples;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    float best = -INFINITY;
    for (int j = 0; j < n_samples; ++j) {
      const int i = b * n_samples + j;  // repeat_interleave order, bcql.py:138,146
      const float q1 = min_over(q_t, n1, nr, i);
      const float q2 = min_over(q_t + (size_t)n1 * nr, n2, nr, i);
      const float v = lmbda * fminf(q1, q2) + (1.0f - lmbda) * fmaxf(q1, q2);
      best = fmaxf(best, v);
    }
    const float nd = done ? (1.0f - done[b]) : 1.0f;
    const float backup = base[b] + gamma * nd * best;
    for (int e = 0; e < n_on; ++e) {
      const float d = q_on[(size_t)e * rows + b] - backup;
      loss += d * d;
      dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
    }
  }
  loss = block_sum(loss, sm);
  if (threadIdx.x == 0 && stat) stat[0] = loss * inv_rows;
}

// ---- batch-sum kernels on a GRID (round 3) --------------------------------------------------------------------
// The single-workgroup loss kernels above are built for B = 2048 rows x a few columns; at BCQ-Lag's C3 shape (4096 rows,
// 10 target samples x 4 target nets per row; VAE with 8 + 16 columns) one CU walks 0.65-1 MB alone: 44 us
// (bcq_critic_loss, on BOTH branches of the step) and 34 us (vae_loss) of a 1.77 ms step.  Issuing more loads per
// thread did not help (70 us / 33 us: the one CU's address path is the limit, not its latency).  Here <= 64
// workgroups of 256 threads split the rows; every workgroup leaves its partial sum(s) in `ws`, signs in with one
// atomic, and the LAST one to arrive adds the partials IN WORKGROUP ORDER (deterministic: no floating-point atomics)
// and writes the statistic.  ws: >= OSRL_LOSS_WS floats, zero once before first use (the last arriver re-arms it).
constexpr int kGridWg = 64, kGridThreads = 256;
__device__ __forceinline__ float block_sum256(float v, float* sm /*>= 5 floats*/) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sm[w] = v;
  __syncthreads();
  return ((sm[0] + sm[1]) + sm[2]) + sm[3];
}
// true in EVERY thread of the last workgroup to call it; partials must have been written by thread 0 before
__device__ __forceinline__ bool grid_last_arriver(float* ws, int slot_counter) {
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned seen = atomicAdd(reinterpret_cast<unsigned*>(ws + slot_counter), 1u);
    s_last = seen == gridDim.x - 1;
    if (s_last) {
      __threadfence();
      reinterpret_cast<unsigned*>(ws + slot_counter)[0] = 0u;  // re-armed for the next launch
    }
  }
  __syncthreads();
  return s_last != 0;
}
// the last workgroup's sum of n_sets x gridDim.x partials, in workgroup order: every partial is fetched by its own
// thread (one round trip for all of them), thread 0 adds them from LDS; result in out[0 .. n_sets) of thread 0
template <int NSETS>
__device__ __forceinline__ void grid_ordered_sum(const float* ws, float (&out)[NSETS]) {
  __shared__ float part[NSETS * kGridWg];
  const int n = (int)gridDim.x;
  for (int i = threadIdx.x; i < NSETS * kGridWg; i += kGridThreads) {
    const int set = i / kGridWg, g = i - set * kGridWg;
    part[i] = g < n ? reinterpret_cast<const volatile float*>(ws)[set * kGridWg + g] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int set = 0; set < NSETS; ++set) {
      float t = 0.f;
      for (int g = 0; g < n; ++g) t += part[set * kGridWg + g];
      out[set] = t;
    }
  }
}

__global__ __launch_bounds__(kGridThreads) void bcq_critic_loss_grid_kernel(
    const float* __restrict__ q_t, int n1, int n2, int n_samples, const float* __restrict__ q_on, int n_on,
    const float* __restrict__ base, const float* __restrict__ done, int rows, float gamma, float lmbda, float inv_rows,
    float* __restrict__ dq, float* __restrict__ stat, float* __restrict__ ws) {
  __shared__ float sm[8];
  float loss = 0.f;
  const int nr = rows * n_samples;
  for (int b = blockIdx.x * kGridThreads + threadIdx.x; b < rows; b += gridDim.x * kGridThreads) {
    float best = -INFINITY;
    for (int j = 0; j < n_samples; ++j) {
      const int i = b * n_samples + j;  // repeat_interleave order, bcql.py:138,146
      const float q1 = min_over(q_t, n1, nr, i);
      const float q2 = min_over(q_t + (size_t)n1 * nr, n2, nr, i);
      const float v = lmbda * fminf(q1, q2) + (1.0f - lmbda) * fmaxf(q1, q2);
      best = fmaxf(best, v);
    }
    const float nd = done ? (1.0f - done[b]) : 1.0f;
    const float backup = base[b] + gamma * nd * best;
    for (int e = 0; e < n_on; ++e) {
      const float d = q_on[(size_t)e * rows + b] - backup;
      loss += d * d;
      dq[(size_t)e * rows + b] = 2.0f * d * inv_rows;
    }
  }
  loss = block_sum256(loss, sm);
  if (threadIdx.x == 0) ws[blockIdx.x] = loss;
  if (grid_last_arriver(ws, 2 * kGridWg)) {
    float t[1];
    grid_ordered_sum<1>(ws, t);
    if (threadIdx.x == 0 && stat) stat[0] = t[0] * inv_rows;
  }
}

__global__ __launch_bounds__(kGridThreads) void vae_loss_grid_kernel(const float* __restrict__ u,
                                                                     const float* __restrict__ act,
                                                                     const float* __restrict__ head, int rows, int ad,
                                                                     int L, float beta, float inv_rows,
                                                                     float* __restrict__ du, float* __restrict__ stat,
                                                                     float* __restrict__ ws) {
  __shared__ float sm[8];
  float rec = 0.f, kl = 0.f;
  const float ia = inv_rows / (float)ad, il = inv_rows / (float)L;
  const int n_rec = rows * ad, n_kl = rows * L;
  const int stride = gridDim.x * kGridThreads;
  for (int i = blockIdx.x * kGridThreads + threadIdx.x; i < n_rec; i += stride) {
    const float d = u[i] - act[i];
    rec += d * d;
    du[i] = 2.0f * d * ia;
  }
  for (int i = blockIdx.x * kGridThreads + threadIdx.x; i < n_kl; i += stride) {
    const int r = i / L, c = i - r * L;
    kl += kl_elem(head[(size_t)r * 2 * L + c], head[(size_t)r * 2 * L + L + c]);
  }
  rec = block_sum256(rec, sm);
  kl = block_sum256(kl, sm);
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = rec;
    ws[kGridWg + blockIdx.x] = kl;
  }
  if (grid_last_arriver(ws, 2 * kGridWg)) {
    float t[2];
    grid_ordered_sum<2>(ws, t);
    if (threadIdx.x == 0 && stat) stat[0] = t[0] * ia + beta * (t[1] * il);
  }
}

// min over the q1 group, min over the q2 group, then the binary min with torch's tie rule
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

// local sums of q_pi and qc_pi over the rows, pre-divided by the GLOBAL batch (data-parallel pre-pass of the
// actor loss: the PID controller needs the global mean of qc_pi before any gradient is formed)
__global__ __launch_bounds__(kRed) void bcq_actor_sums_kernel(const float* __restrict__ q, int nq1, int nq2,
                                                              const float* __restrict__ qc, int nc1, int nc2, int rows,
                                                              float inv_rows, float* __restrict__ out) {
  __shared__ float sm[20];
  float sq = 0.f, sqc = 0.f;
  int i1, i2;
  float w1;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    sq += minmin(q, nq1, nq2, rows, b, &i1, &i2, &w1);
    sqc += minmin(qc, nc1, nc2, rows, b, &i1, &i2, &w1);
  }
  sq = block_sum(sq, sm);
  sqc = block_sum(sqc, sm);
  if (threadIdx.x == 0) {
    out[0] = sq * inv_rows;
    out[1] = sqc * inv_rows;
  }
}

__global__ __launch_bounds__(kRed) void bcq_actor_loss_kernel(const float* __restrict__ q, int nq1, int nq2,
                                                              const float* __restrict__ qc, int nc1, int nc2,
                                                              int rows, float qc_thres, float KP, float KI, float KD,
                                                              float inv_rows, const float* __restrict__ means_in,
                                                              float stat_share, float* __restrict__ pid,
                                                              float* __restrict__ dq, float* __restrict__ dqc,
                                                              float* __restrict__ stat) {
  __shared__ float sm[20];
  __shared__ float s_mult;
  float sq = 0.f, sqc = 0.f;
  int i1, i2;
  float w1;
  if (!means_in) {
    for (int b = threadIdx.x; b < rows; b += kRed) {
      sq += minmin(q, nq1, nq2, rows, b, &i1, &i2, &w1);
      sqc += minmin(qc, nc1, nc2, rows, b, &i1, &i2, &w1);
    }
    sq = block_sum(sq, sm) * inv_rows;
    sqc = block_sum(sqc, sm) * inv_rows;
  } else {  // all-reduced global means
    sq = means_in[0];
    sqc = means_in[1];
  }
  if (threadIdx.x == 0) {
    // LagrangianPIDController.control  net.py:376-387
    const float e_new = sqc - qc_thres;
    const float e_old = pid[0], integ = pid[1];
    const float diff = fmaxf(e_new - e_old, 0.f);
    const float integ_new = fmaxf(integ + e_new, 0.f);
    pid[0] = e_new;
    pid[1] = integ_new;
    const float mult = fmaxf(KP * fmaxf(e_new, 0.f) + KI * integ_new + KD * diff, 0.f);
    s_mult = mult;
    const float penalty = (sqc - qc_thres) * mult;  // mean((qc_pi - thres)*mult)
    if (stat) {
      stat[0] = (-sq + penalty) * stat_share;
      stat[1] = penalty * stat_share;
      stat[2] = mult * stat_share;
    }
  }
  __syncthreads();
  const float mult = s_mult;
  for (int b = threadIdx.x; b < rows; b += kRed) {
    minmin(q, nq1, nq2, rows, b, &i1, &i2, &w1);
    for (int e = 0; e < nq1; ++e) dq[(size_t)e * rows + b] = e == i1 ? -w1 * inv_rows : 0.f;
    for (int e = 0; e < nq2; ++e) dq[(size_t)(nq1 + e) * rows + b] = e == i2 ? -(1.0f - w1) * inv_rows : 0.f;
    minmin(qc, nc1, nc2, rows, b, &i1, &i2, &w1);
    for (int e = 0; e < nc1; ++e) dqc[(size_t)e * rows + b] = e == i1 ? w1 * mult * inv_rows : 0.f;
    for (int e = 0; e < nc2; ++e) dqc[(size_t)(nc1 + e) * rows + b] = e == i2 ? (1.0f - w1) * mult * inv_rows : 0.f;
  }
}

#define S ((hipStream_t)stream)
#define LAUNCH_CHECK() return (int)hipGetLastError()

}  // namespace

extern "C" {

int osrl_gauss_head(const float* head, const float* eps, int32_t rows, int32_t ad, float max_action, float* a,
                    float* tanh_u, float* logp, void* stream) {
  if (!head || rows < 1 || ad < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(gauss_head_kernel, GRID_1D(rows), 0, S, head, eps, rows, ad, max_action, a, tanh_u, logp);
  LAUNCH_CHECK();
}

int osrl_gauss_head_bwd(const float* head, const float* eps, const float* tanh_u, const float* da_nets,
                        int32_t n_nets, int32_t rows, int32_t ad, float max_action, float* dhead, void* stream) {
  if (!head || !eps || !tanh_u || !da_nets || !dhead || rows < 1 || ad < 1 || n_nets < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(gauss_head_bwd_kernel, GRID_1D(rows * ad), 0, S, head, eps, tanh_u, da_nets, n_nets, rows, ad,
                     max_action, dhead);
  LAUNCH_CHECK();
}

int osrl_gauss_ood_sample(const float* head, const float* eps, int32_t n_samples, int32_t rows, int32_t ad,
                          float* out, void* stream) {
  if (!head || !eps || !out || n_samples < 1 || rows < 1 || ad < 1) return -1;
  const int64_t n = (int64_t)n_samples * rows * ad;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(gauss_ood_kernel, GRID_1D(n), 0, S, head, eps, n_samples, rows, ad, out);
  LAUNCH_CHECK();
}

int osrl_vae_latent(const float* head, const float* eps, int32_t rows, int32_t L, float* z, void* stream) {
  if (!head || !eps || !z || rows < 1 || L < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(vae_latent_kernel, GRID_1D(rows * L), 0, S, head, eps, rows, L, z);
  LAUNCH_CHECK();
}

int osrl_vae_loss(const float* u, const float* act, const float* head, int32_t rows, int32_t ad, int32_t L,
                  float beta, int32_t rows_global, float* du, float* stat, void* stream) {
  if (!u || !act || !head || !du || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  const void* dev_args = nullptr;
  if (osrl_argmem::current()) {
    VaeLossArgs a{};
    a.u = u; a.act = act; a.head = head; a.du = du; a.stat = stat;
    a.rows = rows; a.ad = ad; a.L = L; a.beta = beta; a.inv_rows = inv;
    dev_args = osrl_argmem::slot(a);
  }
  if (dev_args)
    hipLaunchKernelGGL(vae_loss_kernel_p, dim3(1), dim3(kRed), 0, S, dev_args);
  else
    hipLaunchKernelGGL(vae_loss_kernel, dim3(1), dim3(kRed), 0, S, u, act, head, rows, ad, L, beta, inv, du, stat);
  LAUNCH_CHECK();
}

int osrl_vae_latent_bwd(const float* head, const float* eps, const float* dz, int32_t rows, int32_t L, float beta,
                        int32_t rows_global, float* dhead, void* stream) {
  if (!head || !eps || !dz || !dhead || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(vae_latent_bwd_kernel, GRID_1D(rows * L), 0, S, head, eps, dz, rows, L, beta,
                     1.0f / (float)(rows_global > 0 ? rows_global : rows), dhead);
  LAUNCH_CHECK();
}

int osrl_vae_kl_rows(const float* head, int32_t rows, int32_t L, float* kl, void* stream) {
  if (!head || !kl || rows < 1 || L < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(vae_kl_rows_kernel, GRID_1D(rows), 0, S, head, rows, L, kl);
  LAUNCH_CHECK();
}

int osrl_quantile(const float* x, int64_t n, float q, float* out, void* stream) {
  if (!x || !out || n < 1 || q < 0.f || q > 1.f) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(quantile_kernel, dim3(1), dim3(kRed), 0, S, x, n, q, out);
  LAUNCH_CHECK();
}

int osrl_quantile_ws(const float* x, int64_t n, float q, uint32_t* ws, float* out, void* stream) {
  if (!x || !out || !ws || n < 1 || n > 0xffffffffll || q < 0.f || q > 1.f) return -1;
  const double pos = (double)q * (double)(n - 1);
  const int64_t lo = (int64_t)floor(pos), hi = lo + 1 < n ? lo + 1 : n - 1;
  int64_t blocks = (n + 1023) / 1024;
  blocks = blocks > 256 ? 256 : blocks;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  for (int pass = 0; pass < 4; ++pass)
    hipLaunchKernelGGL(qsel_hist_kernel, dim3((unsigned)blocks), dim3(1024), 0, S, x, n, lo, pass, ws);
  hipLaunchKernelGGL(qsel_succ_kernel, dim3((unsigned)blocks), dim3(1024), 0, S, x, n, lo, hi, ws);
  hipLaunchKernelGGL(qsel_finish_kernel, dim3(1), dim3(256), 0, S, n, q, ws, out);
  LAUNCH_CHECK();
}

int osrl_cpq_critic_loss(const float* q_old, int32_t n_q_old, const float* qc_old, int32_t n_qc_old,
                         const float* q, int32_t n_q, const float* rew, const float* done, int32_t rows,
                         float gamma, float q_thres, int32_t rows_global, float* dq, float* stat, void* stream) {
  if (!q_old || !qc_old || !q || !rew || !done || !dq || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  const bool small = n_q_old <= kEns && n_qc_old <= kEns && n_q <= kEns;
  if (osrl_argmem::current()) {
    CriticLossArgs a{};
    a.q_old = q_old; a.qc_old = qc_old; a.q = q; a.rew = rew; a.done = done; a.dq = dq; a.stat = stat;
    a.n_q_old = n_q_old; a.n_qc_old = n_qc_old; a.n_q = n_q; a.rows = rows;
    a.gamma = gamma; a.q_thres = q_thres; a.inv_rows = inv;
    if (const void* dev_args = osrl_argmem::slot(a)) {
      if (small)
        hipLaunchKernelGGL(cpq_critic_loss_kernel_p<true>, dim3(1), dim3(kRed), 0, S, dev_args);
      else
        hipLaunchKernelGGL(cpq_critic_loss_kernel_p<false>, dim3(1), dim3(kRed), 0, S, dev_args);
      LAUNCH_CHECK();
    }
  }
  if (small)
    hipLaunchKernelGGL(cpq_critic_loss_kernel<true>, dim3(1), dim3(kRed), 0, S, q_old, n_q_old, qc_old, n_qc_old, q, n_q,
                       rew, done, rows, gamma, q_thres, inv, dq, stat);
  else
    hipLaunchKernelGGL(cpq_critic_loss_kernel<false>, dim3(1), dim3(kRed), 0, S, q_old, n_q_old, qc_old, n_qc_old, q,
                       n_q, rew, done, rows, gamma, q_thres, inv, dq, stat);
  LAUNCH_CHECK();
}

int osrl_cpq_ood_mean(const float* qc_sampled, int32_t n_qc_old, const float* kl, const float* quantile,
                      int32_t n_samples, int32_t rows, int32_t rows_global, float* out, void* stream) {
  if (!qc_sampled || !kl || !quantile || !out || rows < 1 || n_samples < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  if (n_qc_old <= kEns)
    hipLaunchKernelGGL(cpq_ood_mean_kernel<true>, dim3(1), dim3(kRed), 0, S, qc_sampled, n_qc_old, kl, quantile,
                       n_samples, rows, inv, out);
  else
    hipLaunchKernelGGL(cpq_ood_mean_kernel<false>, dim3(1), dim3(kRed), 0, S, qc_sampled, n_qc_old, kl, quantile,
                       n_samples, rows, inv, out);
  LAUNCH_CHECK();
}

int osrl_cpq_ood_stat(const float* qc_sampled, int32_t n_qc_old, const float* kl, float q, int32_t n_samples,
                      int32_t rows, int32_t rows_global, float* quant_out, float* out, void* stream) {
  if (!qc_sampled || !kl || !quant_out || !out || rows < 1 || n_samples < 1 || q < 0.f || q > 1.f) return -1;
  if ((int64_t)n_samples * rows > (int64_t)32 * kRed) return -2;  // keys must fit the workgroup's registers
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  if (n_qc_old <= kEns)
    hipLaunchKernelGGL(cpq_ood_stat_kernel<true>, dim3(1), dim3(kRed), 0, S, qc_sampled, n_qc_old, kl, q, n_samples, rows,
                       inv, quant_out, out);
  else
    hipLaunchKernelGGL(cpq_ood_stat_kernel<false>, dim3(1), dim3(kRed), 0, S, qc_sampled, n_qc_old, kl, q, n_samples,
                       rows, inv, quant_out, out);
  LAUNCH_CHECK();
}

int osrl_cpq_ood_select(const float* kl, const float* quantile_in, float q, int32_t n, float* quant_out, int32_t* list,
                        int32_t* count, void* stream) {
  if (!kl || !list || !count || n < 1 || q < 0.f || q > 1.f) return -1;
  if ((int64_t)n > (int64_t)32 * kRed) return -2;  // keys in the workgroup's registers, <= 32 indices per thread
  (void)hipGetLastError();
  hipLaunchKernelGGL(cpq_ood_select_kernel, dim3(1), dim3(kRed), 0, S, kl, quantile_in, q, n, quant_out, list, count);
  LAUNCH_CHECK();
}

int osrl_cpq_ood_sum(const float* qc_sel, int32_t n_qc, int32_t cap, const int32_t* count, float scale, float* out,
                     void* stream) {
  if (!qc_sel || !count || !out || n_qc < 1 || cap < 1) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(cpq_ood_sum_kernel, dim3(1), dim3(kRed), 0, S, qc_sel, n_qc, cap, count, scale, out);
  LAUNCH_CHECK();
}

int osrl_cpq_cost_loss(const float* qc_old_next, int32_t n_qc_old, const float* qc, int32_t n_qc,
                       const float* ood_mean, const float* cost, int32_t rows, float gamma, float qc_thres,
                       float alpha_lr, int32_t rows_global, float stat_share, float* log_alpha, float* dq,
                       float* stat, void* stream) {
  if (!qc_old_next || !qc || !cost || (ood_mean && !log_alpha) || !dq || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  const bool small = n_qc_old <= kEns && n_qc <= kEns;
  if (osrl_argmem::current()) {
    CostLossArgs a{};
    a.qc_old_next = qc_old_next; a.qc = qc; a.cost = cost; a.ood_mean_p = const_cast<float*>(ood_mean);
    a.log_alpha = log_alpha; a.dq = dq; a.stat = stat; a.oa = OodArgs{nullptr, nullptr, nullptr, 0, 0};
    a.n_qc_old = n_qc_old; a.n_qc = n_qc; a.rows = rows; a.gamma = gamma; a.qc_thres = qc_thres;
    a.alpha_lr = alpha_lr; a.inv_rows = inv; a.stat_share = stat_share;
    if (const void* dev_args = osrl_argmem::slot(a)) {
      if (small)
        hipLaunchKernelGGL(cpq_cost_loss_kernel_p<true>, dim3(1), dim3(kRed), 0, S, dev_args);
      else
        hipLaunchKernelGGL(cpq_cost_loss_kernel_p<false>, dim3(1), dim3(kRed), 0, S, dev_args);
      LAUNCH_CHECK();
    }
  }
  if (small)
    hipLaunchKernelGGL(cpq_cost_loss_kernel<true>, dim3(1), dim3(kRed), 0, S, qc_old_next, n_qc_old, qc, n_qc,
                       const_cast<float*>(ood_mean), cost, rows, gamma, qc_thres, alpha_lr, inv, stat_share, log_alpha,
                       dq, stat, OodArgs{nullptr, nullptr, nullptr, 0, 0});
  else
    hipLaunchKernelGGL(cpq_cost_loss_kernel<false>, dim3(1), dim3(kRed), 0, S, qc_old_next, n_qc_old, qc, n_qc,
                       const_cast<float*>(ood_mean), cost, rows, gamma, qc_thres, alpha_lr, inv, stat_share, log_alpha,
                       dq, stat, OodArgs{nullptr, nullptr, nullptr, 0, 0});
  LAUNCH_CHECK();
}

int osrl_cpq_alpha_step(const float* ood_mean, float qc_thres, float alpha_lr, float stat_share, float* log_alpha,
                        float* stat, void* stream) {
  if (!ood_mean || !log_alpha) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(cpq_alpha_step_kernel, dim3(1), dim3(1), 0, S, ood_mean, qc_thres, alpha_lr, stat_share,
                     log_alpha, stat);
  LAUNCH_CHECK();
}

int osrl_cpq_cost_loss_ood(const float* qc_sampled, int32_t n_qc_sampled, const float* kl, const float* quantile,
                           int32_t n_samples, const float* qc_old_next, int32_t n_qc_old, const float* qc,
                           int32_t n_qc, float* ood_mean_out, const float* cost, int32_t rows, float gamma,
                           float qc_thres, float alpha_lr, float* log_alpha, float* dq, float* stat, void* stream) {
  if (!qc_sampled || !kl || !quantile || n_samples < 1 || !qc_old_next || !qc || !ood_mean_out || !cost ||
      !log_alpha || !dq || rows < 1)
    return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  if (n_qc_old <= kEns && n_qc <= kEns && n_qc_sampled <= kEns)
    hipLaunchKernelGGL(cpq_cost_loss_kernel<true>, dim3(1), dim3(kRed), 0, S, qc_old_next, n_qc_old, qc, n_qc,
                       ood_mean_out, cost, rows, gamma, qc_thres, alpha_lr, 1.0f / (float)rows, 1.0f, log_alpha, dq, stat,
                       OodArgs{qc_sampled, kl, quantile, n_qc_sampled, n_samples});
  else
    hipLaunchKernelGGL(cpq_cost_loss_kernel<false>, dim3(1), dim3(kRed), 0, S, qc_old_next, n_qc_old, qc, n_qc,
                       ood_mean_out, cost, rows, gamma, qc_thres, alpha_lr, 1.0f / (float)rows, 1.0f, log_alpha, dq, stat,
                       OodArgs{qc_sampled, kl, quantile, n_qc_sampled, n_samples});
  LAUNCH_CHECK();
}

int osrl_cpq_actor_loss(const float* q, int32_t n_q, const float* qc, int32_t n_qc, int32_t rows, float q_thres,
                        int32_t rows_global, float* dq, float* stat, void* stream) {
  if (!q || !qc || !dq || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const float inv = 1.0f / (float)(rows_global > 0 ? rows_global : rows);
  if (n_q <= kEns && n_qc <= kEns)
    hipLaunchKernelGGL(cpq_actor_loss_kernel<true>, dim3(1), dim3(kRed), 0, S, q, n_q, qc, n_qc, rows, q_thres, inv, dq,
                       stat);
  else
    hipLaunchKernelGGL(cpq_actor_loss_kernel<false>, dim3(1), dim3(kRed), 0, S, q, n_q, qc, n_qc, rows, q_thres, inv, dq,
                       stat);
  LAUNCH_CHECK();
}

int osrl_mse_loss(const float* u, const float* target, int64_t n, int64_t n_global, float* du, float* stat,
                  void* stream) {
  if (!u || !target || !du || n < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(kRed), 0, S, u, target, (int)n,
                     1.0f / (float)(n_global > 0 ? n_global : n), du, stat);
  LAUNCH_CHECK();
}

int osrl_clamp(float* x, int64_t n, float lo, float hi, void* stream) {
  if (!x || n < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(clamp_kernel, GRID_1D(n), 0, S, x, n, lo, hi);
  LAUNCH_CHECK();
}

int osrl_bcq_perturb(const float* dec, const float* t, int32_t rows, int32_t ad, float phi, float max_action,
                     float* a, void* stream) {
  if (!dec || !t || !a || rows < 1 || ad < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(bcq_perturb_kernel, GRID_1D((int64_t)rows * ad), 0, S, dec, t, rows * ad, phi, max_action, a);
  LAUNCH_CHECK();
}

int osrl_bcq_perturb_bwd(const float* dec, const float* t, const float* da_nets, int32_t n_nets, int32_t rows,
                         int32_t ad, float phi, float max_action, float* dt, void* stream) {
  if (!dec || !t || !da_nets || !dt || rows < 1 || ad < 1 || n_nets < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(bcq_perturb_bwd_kernel, GRID_1D((int64_t)rows * ad), 0, S, dec, t, da_nets, n_nets, rows * ad,
                     phi, max_action, dt);
  LAUNCH_CHECK();
}

int osrl_bcq_critic_loss(const float* q_t, int32_t n1, int32_t n2, int32_t n_samples, const float* q_on,
                         int32_t n_on, const float* base, const float* done, int32_t rows, float gamma,
                         float lmbda, int32_t rows_global, float* dq, float* stat, void* stream) {
  if (!q_t || !q_on || !base || !dq || rows < 1 || n1 < 1 || n2 < 1 || n_samples < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(bcq_critic_loss_kernel, dim3(1), dim3(kRed), 0, S, q_t, n1, n2, n_samples, q_on, n_on, base,
                     done, rows, gamma, lmbda, 1.0f / (float)(rows_global > 0 ? rows_global : rows), dq, stat);
  LAUNCH_CHECK();
}

static int loss_grid(int64_t n_items) {
  int64_t g = (n_items + kGridThreads - 1) / kGridThreads;
  return (int)(g < 1 ? 1 : g > kGridWg ? kGridWg : g);
}

int osrl_bcq_critic_loss_ws(const float* q_t, int32_t n1, int32_t n2, int32_t n_samples, const float* q_on, int32_t n_on,
                            const float* base, const float* done, int32_t rows, float gamma, float lmbda,
                            int32_t rows_global, float* dq, float* stat, float* ws, void* stream) {
  if (!q_t || !q_on || !base || !dq || !ws || rows < 1 || n1 < 1 || n2 < 1 || n_samples < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(bcq_critic_loss_grid_kernel, dim3(loss_grid(rows)), dim3(kGridThreads), 0, S, q_t, n1, n2,
                     n_samples, q_on, n_on, base, done, rows, gamma, lmbda,
                     1.0f / (float)(rows_global > 0 ? rows_global : rows), dq, stat, ws);
  LAUNCH_CHECK();
}

int osrl_vae_loss_ws(const float* u, const float* act, const float* head, int32_t rows, int32_t ad, int32_t L, float beta,
                     int32_t rows_global, float* du, float* stat, float* ws, void* stream) {
  if (!u || !act || !head || !du || !ws || rows < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const int64_t n = (int64_t)rows * (ad > L ? ad : L);
  hipLaunchKernelGGL(vae_loss_grid_kernel, dim3(loss_grid(n)), dim3(kGridThreads), 0, S, u, act, head, rows, ad, L,
                     beta, 1.0f / (float)(rows_global > 0 ? rows_global : rows), du, stat, ws);
  LAUNCH_CHECK();
}

int osrl_bcq_actor_sums(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                        int32_t rows, int32_t rows_global, float* out, void* stream) {
  if (!q || !qc || !out || rows < 1) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bcq_actor_sums_kernel, dim3(1), dim3(kRed), 0, S, q, nq1, nq2, qc, nc1, nc2, rows,
                     1.0f / (float)(rows_global > 0 ? rows_global : rows), out);
  LAUNCH_CHECK();
}

int osrl_bcq_actor_loss(const float* q, int32_t nq1, int32_t nq2, const float* qc, int32_t nc1, int32_t nc2,
                        int32_t rows, float qc_thres, float KP, float KI, float KD, int32_t rows_global,
                        const float* global_means, float stat_share, float* pid, float* dq, float* dqc, float* stat,
                        void* stream) {
  if (!q || !qc || !pid || !dq || !dqc || rows < 1) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(bcq_actor_loss_kernel, dim3(1), dim3(kRed), 0, S, q, nq1, nq2, qc, nc1, nc2, rows, qc_thres,
                     KP, KI, KD, 1.0f / (float)(rows_global > 0 ? rows_global : rows), global_means, stat_share, pid, dq,
                     dqc, stat);
  LAUNCH_CHECK();
}

const char* osrl_version(void) { return "osrl_amd 0.1 (gfx950)"; }

}  // extern "C"
