// act.hip -- the B = 1 (.. 4) act() latency path for gfx950 (MI355X).
//
// Replaces, for the reference's episode-by-episode evaluation loop (XTrainer.rollout: osrl/algorithms/cpq.py:330-347,
// bcql.py:322-340, bc.py:130-149), what one `model.act(obs)` call costs there: `torch.tensor(obs[None]).to(device)`
// (H2D), 3-6 aten kernels (Linear / ReLU / tanh / clamp ...), two `.cpu().numpy()` syncs (D2H) -- cpq.py:240-252,
// bcql.py:236-243, bc.py:66-76 -- with ONE kernel launch and no copy calls:
//   * the observation is written by the host straight into a pinned, device-mapped buffer; the kernel reads it over the
//     host link, runs the whole policy (1-2 chained MLPs + the distribution head) in one workgroup and writes the
//     action (+ log-prob) back into pinned memory, then publishes a sequence number the host spins on (system-scope
//     release / acquire) -- no hipMemcpy, no stream synchronise on the fast path;
//   * the layers are GEMVs: at 1-4 rows a 16-row MFMA tile would be >= 75 % padding and the work is a pure weight
//     stream (86 K parameters = 344 KB for the CPQ actor): LANES own output neurons and read the packed forward
//     weights PF[k/4][n][k%4] -- the copies the optimizer kernel keeps fresh -- so every wave load is 1 KB contiguous
//     and no dot product needs a cross-lane reduction; the 1024 threads split k as well, partials meet in LDS;
// HBM/L2-bound on one CU by design (latency, not throughput): ~345 KB at the per-CU L2 rate is ~3 us.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <new>

#include "../../include/osrl_amd.h"
#include "philox.h"

using osrl_rng::U4;
using osrl_rng::philox4x32_10;

namespace {

constexpr int kThreads = 1024, kWaves = 16;
constexpr int kMaxRows = OSRL_POLICY_MAX_ROWS;
constexpr int kW = 512;  // LDS row stride (floats) >= widest layer (OSRL_MAX_WIDTH = 448) and obs+act inputs
constexpr float kLogStdMin = -20.0f, kLogStdMax = 2.0f;  // net.py:148-149

__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float act_fwd(int act, float x) {
  if (act == OSRL_ACT_RELU) return fmaxf(x, 0.0f);
  if (act == OSRL_ACT_TANH) return tanhf(x);
  return x;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct Io {  // pinned + device-mapped; the host writes obs / noise, the kernel writes act / logp / seq
  float* obs;     // [kMaxRows, obs_dim]
  float* noise;   // [kMaxRows, noise_dim] explicit standard-normal noise (tests) -- else drawn in the kernel
  float* act;     // [kMaxRows, act_dim]
  float* logp;    // [kMaxRows]
  uint64_t* seq;  // completion counter
};

constexpr int kInline = 128;  // floats of observation carried in the kernel arguments themselves

struct ActArgs {
  osrl_policy_t p;
  Io io;
  int32_t rows, deterministic, host_noise, obs_inline;
  uint32_t k0, k1;
  uint64_t counter, seq;
  // rows * obs_dim <= kInline: the observation travels in the kernarg segment (written by the host with the launch
  // packet, read by scalar loads) instead of being fetched from pinned host memory by the kernel -- one host-link
  // round trip less on the critical path
  float obs[kInline];
};

// y[r][n] = act(b[n] + sum_k W[n][k] x[r][k]) * scale for r < R rows; x, y in LDS (stride kW), `red` = LDS scratch.
// The weights are read from the PACKED forward copy PF[k/4][n][k%4] (osrl_pack_weights: zero padded to multiples of 16
// in both dims) with LANES OWNING OUTPUT NEURONS: for a fixed k-quad consecutive n are consecutive 16-byte words, so
// every wave load is 1 KB contiguous and a dot product needs NO cross-lane reduction (a per-neuron wave reduction is a
// chain of six ds_bpermute: measured 5 us per layer).  The 1024 threads = KS k-splits x NL neuron lanes
// (NL = min(256, Np rounded up to a power of two)); each thread issues all its loads back to back; the KS partials meet
// in LDS.
__device__ __forceinline__ int round16(int x) { return (x + 15) & ~15; }

template <int R>
__device__ __forceinline__ void gemv_layer(const float* __restrict__ PF, const float* __restrict__ b, int in, int out,
                                           int act, float scale, const float* x, float* y, float* red) {
  const int tid = threadIdx.x;
  const int Np = round16(out), nq = round16(in) >> 2;
  int NL = 16;
  while (NL < Np && NL < 256) NL <<= 1;
  const int KS = kThreads / NL;
  const int ks = tid / NL, nl = tid - ks * NL;
  const int q0 = (nq * ks) / KS, q1 = (nq * (ks + 1)) / KS;
  const int rs = Np > 256 ? 512 : NL;  // row stride of `red`: KS * R * rs <= 2048 * R floats
  const float4* __restrict__ P4 = reinterpret_cast<const float4*>(PF);
  for (int n = nl; n < Np; n += 256) {
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int q = q0; q < q1; ++q) {
      const float4 w = P4[(size_t)q * Np + n];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 xv = *reinterpret_cast<const float4*>(x + r * kW + 4 * q);  // same address across the wave
        acc[r] = fmaf(w.x, xv.x, fmaf(w.y, xv.y, fmaf(w.z, xv.z, fmaf(w.w, xv.w, acc[r]))));
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) red[(ks * R + r) * rs + n] = acc[r];
  }
  __syncthreads();
  for (int i = tid; i < R * Np; i += kThreads) {
    const int r = i / Np, n = i - r * Np;
    float s = 0.f;
    for (int k = 0; k < KS; ++k) s += red[(k * R + r) * rs + n];
    y[r * kW + n] = n < out ? act_fwd(act, s + b[n]) * scale : 0.f;  // zero = the next layer's k padding
  }
  __syncthreads();
}

// one MLP: input (zero padded to a multiple of 16 columns) in buf[0]; returns the index of the buffer with the output
template <int R>
__device__ __forceinline__ int run_net(const osrl_gemv_net_t& n, float (*buf)[kMaxRows * kW], float* red) {
  int cur = 0;
  for (int l = 0; l < n.n_layers; ++l) {
    const float sc = l == n.n_layers - 1 ? n.out_scale : 1.0f;
    gemv_layer<R>(n.Wf[l], n.b[l], n.dims[l], n.dims[l + 1], n.acts[l], sc, buf[cur], buf[cur ^ 1], red);
    cur ^= 1;
  }
  return cur;
}

__device__ __forceinline__ float draw_normal(const ActArgs& a, int idx) {
  // Philox4x32-10 keyed like csrc/rng.hip (counter = (element/4, call counter lo, hi, stream 0xAC7)), Box-Muller
  const U4 r = philox4x32_10(U4{(uint32_t)(idx >> 2), (uint32_t)a.counter, (uint32_t)(a.counter >> 32), 0xAC7u}, a.k0,
                             a.k1);
  const uint32_t u[4] = {r.x, r.y, r.z, r.w};
  const int pair = (idx & 3) >> 1;
  const float u1 = ((float)(u[2 * pair] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(u[2 * pair + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.283185307179586f * u2, &s, &c);
  return (idx & 1) ? rad * s : rad * c;
}

#ifdef OSRL_ACT_STAMPS  // debug: 100 MHz wall-clock stamps of thread 0 into logp[1..7] (rows = 1 runs only)
#define ACT_STAMP(i) if (threadIdx.x == 0) stamp_[i] = wall_clock64();
#else
#define ACT_STAMP(i)
#endif

template <int R>
__global__ __launch_bounds__(kThreads) void policy_act_kernel(const ActArgs a) {
  __shared__ __attribute__((aligned(16))) float buf[2][kMaxRows * kW];
  __shared__ __attribute__((aligned(16))) float red[2048 * R];  // [KS][R][stride] partial sums of a layer
#ifdef OSRL_ACT_STAMPS
  long long stamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  ACT_STAMP(0);
  const int tid = threadIdx.x;
  const osrl_policy_t& p = a.p;
  const int od = p.obs_dim, ad = p.act_dim, rows = a.rows;
  // ---- stage the observation rows (host-mapped memory) + the second input segment of stage 0
  for (int i = tid; i < R * kW; i += kThreads) {
    const int r = i / kW, c = i - r * kW;
    float v = 0.f;
    if (r < rows) {
      if (c < od) {
        v = a.obs_inline ? a.obs[r * od + c] : a.io.obs[r * od + c];
      } else if (p.kind == OSRL_POLICY_BCQ && c < od + p.latent_dim) {
        // vae.decode(obs) draws z ~ clamp(N(0,1), +-0.5) (net.py:331-334); deterministic callers still sample there,
        // like the reference does
        const int j = r * p.latent_dim + (c - od);
        const float z = a.host_noise ? a.io.noise[j] : draw_normal(a, j);
        v = fminf(fmaxf(z, -0.5f), 0.5f);
      }
    }
    buf[0][i] = v;
  }
  __syncthreads();
  ACT_STAMP(1);
  int cur = run_net<R>(p.net[0], buf, red);
  ACT_STAMP(2);
  if (p.kind == OSRL_POLICY_MLP) {  // BC: act_limit * tanh(mlp(obs)) -- tanh + scale are the net's last layer
    for (int i = tid; i < rows * ad; i += kThreads) a.io.act[i] = buf[cur][(i / ad) * kW + (i % ad)];
  } else if (p.kind == OSRL_POLICY_GAUSS) {
    // SquashedGaussianMLPActor tail (net.py:176-201): head = (mu | log_std)
    if (tid < rows) {
      const float* h = buf[cur] + tid * kW;
      float lp = 0.f;
      for (int j = 0; j < ad; ++j) {
        const float mu = h[j];
        const float ls = fminf(fmaxf(h[ad + j], kLogStdMin), kLogStdMax);
        float e = 0.f;
        if (!a.deterministic) e = a.host_noise ? a.io.noise[tid * ad + j] : draw_normal(a, tid * ad + j);
        const float u = mu + expf(ls) * e;
        a.io.act[tid * ad + j] = p.max_action * tanhf(u);
        lp += -0.5f * e * e - ls - 0.9189385332046727f;
        lp -= 2.0f * (0.6931471805599453f - u - softplus(-2.0f * u));
      }
      a.io.logp[tid] = lp;
    }
  } else {  // OSRL_POLICY_BCQ: a0 = decoder([obs, z]); t = pi([obs, a0]); a = clamp(a0 + phi*max_a*t)  (net.py:58-62)
    float* nxt = buf[cur ^ 1];
    const float* dec = buf[cur];
    for (int i = tid; i < R * kW; i += kThreads) {
      const int r = i / kW, c = i - r * kW;
      float v = 0.f;
      if (r < rows)
        v = c < od ? (a.obs_inline ? a.obs[r * od + c] : a.io.obs[r * od + c]) : (c < od + ad ? dec[r * kW + (c - od)] : 0.f);
      nxt[i] = v;
    }
    __syncthreads();
    // keep a0 (rows x ad) in registers of the first threads across the second net
    float a0 = 0.f;
    if (tid < rows * ad) a0 = dec[(tid / ad) * kW + (tid % ad)];
    __syncthreads();
    if (cur == 0) {  // run_net expects its input in buf[0]
      for (int i = tid; i < R * kW; i += kThreads) buf[0][i] = buf[1][i];
      __syncthreads();
    }
    const int c2 = run_net<R>(p.net[1], buf, red);
    if (tid < rows * ad) {
      const float t = buf[c2][(tid / ad) * kW + (tid % ad)];
      a.io.act[tid] = fminf(fmaxf(a0 + p.phi * p.max_action * t, -p.max_action), p.max_action);
    }
  }
  // ---- publish: results must be visible to the host before the sequence number
  __syncthreads();
  ACT_STAMP(3);
  if (tid == 0) {
    __threadfence_system();
#ifdef OSRL_ACT_STAMPS
    stamp_[4] = wall_clock64();
    a.io.logp[1] = (float)(stamp_[1] - stamp_[0]) * 0.01f;  // us: stage-in
    a.io.logp[2] = (float)(stamp_[2] - stamp_[1]) * 0.01f;  // the network(s)
    a.io.logp[3] = (float)(stamp_[4] - stamp_[2]) * 0.01f;  // head + fence
#endif
    __hip_atomic_store(a.io.seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

struct Handle {
  osrl_policy_t p;
  Io host, dev;
  void* pinned;
  size_t bytes;
  uint64_t seq, calls;
  int noise_dim;
};

bool valid_gemv(const osrl_gemv_net_t& n) {
  if (n.n_layers < 1 || n.n_layers > OSRL_MAX_LAYERS || n.out_scale == 0.f) return false;
  for (int l = 0; l <= n.n_layers; ++l)
    if (n.dims[l] < 1 || n.dims[l] > kW) return false;
  for (int l = 0; l < n.n_layers; ++l)
    if (!n.Wf[l] || !n.b[l]) return false;
  return true;
}

}  // namespace

extern "C" int osrl_policy_create(const osrl_policy_t* desc, void** handle) {
  if (!desc || !handle) return -1;
  const osrl_policy_t& p = *desc;
  if (p.kind < OSRL_POLICY_MLP || p.kind > OSRL_POLICY_BCQ || p.obs_dim < 1 || p.act_dim < 1 || !valid_gemv(p.net[0]))
    return -1;
  int noise_dim = 0;
  if (p.kind == OSRL_POLICY_MLP) {
    if (p.net[0].dims[0] != p.obs_dim || p.net[0].dims[p.net[0].n_layers] != p.act_dim) return -1;
  } else if (p.kind == OSRL_POLICY_GAUSS) {
    if (p.net[0].dims[0] != p.obs_dim || p.net[0].dims[p.net[0].n_layers] != 2 * p.act_dim) return -1;
    noise_dim = p.act_dim;
  } else {
    if (!valid_gemv(p.net[1]) || p.latent_dim < 1 || p.net[0].dims[0] != p.obs_dim + p.latent_dim ||
        p.net[0].dims[p.net[0].n_layers] != p.act_dim || p.net[1].dims[0] != p.obs_dim + p.act_dim ||
        p.net[1].dims[p.net[1].n_layers] != p.act_dim || p.obs_dim + p.latent_dim > kW || p.obs_dim + p.act_dim > kW)
      return -1;
    noise_dim = p.latent_dim;
  }
  Handle* h = new (std::nothrow) Handle;
  if (!h) return -1;
  h->p = p;
  h->noise_dim = noise_dim;
  h->seq = h->calls = 0;
  auto r256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
  const size_t o_obs = 0, o_noise = o_obs + r256(sizeof(float) * kMaxRows * p.obs_dim),
               o_act = o_noise + r256(sizeof(float) * kMaxRows * (noise_dim > 0 ? noise_dim : 1)),
               o_logp = o_act + r256(sizeof(float) * kMaxRows * p.act_dim), o_seq = o_logp + r256(sizeof(float) * kMaxRows);
  h->bytes = o_seq + 256;
  hipError_t e = hipHostMalloc(&h->pinned, h->bytes, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) {
    delete h;
    return (int)e;
  }
  memset(h->pinned, 0, h->bytes);
  void* dptr = nullptr;
  e = hipHostGetDevicePointer(&dptr, h->pinned, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(h->pinned);
    delete h;
    return (int)e;
  }
  auto at = [](void* base, size_t off) { return reinterpret_cast<char*>(base) + off; };
  h->host = Io{(float*)at(h->pinned, o_obs), (float*)at(h->pinned, o_noise), (float*)at(h->pinned, o_act),
               (float*)at(h->pinned, o_logp), (uint64_t*)at(h->pinned, o_seq)};
  h->dev = Io{(float*)at(dptr, o_obs), (float*)at(dptr, o_noise), (float*)at(dptr, o_act), (float*)at(dptr, o_logp),
              (uint64_t*)at(dptr, o_seq)};
  *handle = h;
  return 0;
}

extern "C" int osrl_policy_io(void* handle, float** obs, float** noise, float** act, float** logp) {
  if (!handle) return -1;
  Handle* h = static_cast<Handle*>(handle);
  if (obs) *obs = h->host.obs;
  if (noise) *noise = h->host.noise;
  if (act) *act = h->host.act;
  if (logp) *logp = h->host.logp;
  return 0;
}

extern "C" int osrl_policy_act(void* handle, int32_t rows, int32_t deterministic, int32_t host_noise, uint64_t seed,
                               void* stream) {
  if (!handle || rows < 1 || rows > kMaxRows) return -1;
  Handle* h = static_cast<Handle*>(handle);
  ActArgs a;
  a.p = h->p;
  a.io = h->dev;
  a.rows = rows;
  a.deterministic = deterministic;
  a.host_noise = host_noise;
  a.k0 = (uint32_t)seed;
  a.k1 = (uint32_t)(seed >> 32);
  a.counter = ++h->calls;
  a.seq = ++h->seq;
  a.obs_inline = rows * h->p.obs_dim <= kInline;
  if (a.obs_inline) memcpy(a.obs, h->host.obs, sizeof(float) * rows * h->p.obs_dim);
  (void)hipGetLastError();
  if (rows == 1)
    hipLaunchKernelGGL(policy_act_kernel<1>, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(policy_act_kernel<kMaxRows>, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  // fast path: spin on the sequence number the kernel publishes (system-scope release) -- a stream synchronise costs
  // more than the kernel; after 2 ms fall back to it (also surfaces a faulted launch instead of spinning forever)
  volatile uint64_t* seq = h->host.seq;
  const auto t0 = std::chrono::steady_clock::now();
  for (uint32_t it = 0;; ++it) {
    if (*seq >= a.seq) break;
    if ((it & 1023) == 1023 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
      e = hipStreamSynchronize((hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
      if (*seq < a.seq) return -2;  // the kernel ran but did not publish: should be impossible
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

extern "C" int osrl_policy_destroy(void* handle) {
  if (!handle) return -1;
  Handle* h = static_cast<Handle*>(handle);
  (void)hipDeviceSynchronize();  // no launch of this handle may still be writing the pinned block
  const hipError_t e = hipHostFree(h->pinned);
  delete h;
  return (int)e;
}
