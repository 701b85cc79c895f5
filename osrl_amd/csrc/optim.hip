// optim.hip -- device-resident optimizer step for gfx950.
//
// Replaces torch.optim.Adam.step / AdamW.step (osrl/algorithms/cpq.py:232-238, bcql.py:218-226,
// bc.py:54-55, cdt.py:321-326) and the python-loop Polyak `_soft_update` (cpq.py:107-113,
// bcql.py:114-120), fused into ONE pass over a flat fp32 parameter group:
//     g  = sum_s slabs[s][i]                (fixed-order reduction of the split-K dW partials)
//     m  = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2
//     p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
//     tgt = tau p + (1-tau) tgt             (only for groups that own a target copy)
// HBM-bound streaming kernel: float4 per lane, grid-stride, traffic = (S + 7 [+2]) * 4 B / param.
// Fusing Polyak here is exact: in every algorithm the target of a group is not read again between
// that group's optimizer step and the end of train_one_step (CPQ/BCQL phase order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"
#include "step.h"
#include "argmem.h"
#include "adam.h"
#include "trace.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ void step_tick_kernel(osrl_step_state_t* st, const osrl_step_state_t* peer, float beta1, float beta2,
                                 int warmup, const float* __restrict__ stats_cur, float* __restrict__ ring, int n_stats,
                                 int ring_len) {
  const int64_t t_own = st->step;  // the step whose statistics stats_cur holds
  const int64_t t_peer = peer ? peer->step : t_own;
  osrl_step::commit_stats(t_own, stats_cur, ring, n_stats, ring_len);
  __syncthreads();
  if (threadIdx.x == 0) osrl_step::advance(st, t_peer > t_own ? t_peer : t_own, beta1, beta2, warmup);
}

// sum_s slabs[s][i] in slab order; 8 loads are issued before the first add so their latencies overlap
// (a one-load-per-iteration loop serialises up to 32 L2/HBM round trips: the Adam kernel sat at 12 us)
// g + sum_{s >= s0} slabs[s][i], in slab order
__device__ __forceinline__ f32x4 slab_sum_from(const float* __restrict__ slabs, int s0, int n_splits,
                                               int64_t slab_stride, int64_t i, f32x4 g) {
  int s = s0;
  for (; s + 8 <= n_splits; s += 8) {
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(slabs + (size_t)(s + j) * slab_stride)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) g += t[j];
  }
  if (s < n_splits) {
    f32x4 t[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int sj = s + j < n_splits ? s + j : s;
      t[j] = reinterpret_cast<const f32x4*>(slabs + (size_t)sj * slab_stride)[i];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (s + j < n_splits) g += t[j];
  }
  return g;
}
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ slabs, int n_splits, int64_t slab_stride,
                                          int64_t i) {
  return slab_sum_from(slabs, 1, n_splits, slab_stride, i, reinterpret_cast<const f32x4*>(slabs)[i]);
}
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pin4(f32x4& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin4(i32x4& x) { asm volatile("" : "+v"(x)); }

// packed-weight refresh fused into the optimizer step: map_f / map_b give, per flat parameter, its position in the
// fragment-ordered forward / backward copies (-1 = not a packed weight); the zero padding of those copies is
// never touched.  Replaces two pack_kernel launches per optimizer group and step.
struct PackMap {
  const int32_t* map_f;
  const int32_t* map_b;
  float* pf;
  float* pb;
  float* tf;
};

// i0 / stride: this thread's first float4 and the grid stride (the by-value kernel derives them from the launch
// geometry, the device-resident-descriptor twin from its descriptor)
__device__ __forceinline__ void adam_body(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                          float* __restrict__ tgt, const float* __restrict__ slabs, int n_splits,
                                          int64_t slab_stride, int64_t n4, float lr, float b1, float b2, float eps,
                                          float wd, float tau, const float* __restrict__ gscale,
                                          const osrl_step_state_t* __restrict__ st, const PackMap pk, int64_t i0,
                                          int64_t stride) {
  __builtin_amdgcn_s_setprio(3);  // latency-chain kernel: outrank the N*B-row filler launches (csrc/mlp.hip)
  const float lr_t = lr * st->lr_scale;
  const float step_size = lr_t / st->bc1;
  const float bc2s = st->bc2_sqrt;
  const float gs = gscale ? *gscale : 1.0f;
  const float decay = 1.0f - lr_t * wd;
  // Everything an element needs is REQUESTED before anything is consumed: the first eight slabs (slabs past n_splits
  // re-read slab 0), p, m, v, the target and the two pack maps (absent ones re-read p) -- one memory round trip instead
  // of five dependent ones (slabs -> p/m/v -> target -> map_f -> map_b), which is what a group of <= 200k parameters
  // (one element per thread, every wave resident at once) spends its 7-8 us on.  pin4() keeps the compiler from sinking
  // a load into the branch that consumes it.
  const f32x4* __restrict__ T4 = reinterpret_cast<const f32x4*>(tgt ? tgt : p);
  const i32x4* __restrict__ MF4 = reinterpret_cast<const i32x4*>(pk.map_f ? (const void*)pk.map_f : (const void*)p);
  const i32x4* __restrict__ MB4 = reinterpret_cast<const i32x4*>(pk.map_b ? (const void*)pk.map_b : (const void*)p);
  for (int64_t i = i0; i < n4; i += stride) {
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      t[j] = reinterpret_cast<const f32x4*>(slabs + (size_t)(j < n_splits ? j : 0) * slab_stride)[i];
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    f32x4 tv0 = T4[i];
    i32x4 mf = MF4[i], mb = MB4[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) pin4(t[j]);
    pin4(pv);
    pin4(mv);
    pin4(vv);
    pin4(tv0);
    pin4(mf);
    pin4(mb);
    f32x4 g = t[0];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < n_splits) g += t[j];
    if (n_splits > 8) g = slab_sum_from(slabs, 8, n_splits, slab_stride, i, g);
    g *= gs;
    if (wd != 0.0f) pv *= decay;
    osrl_adam::update4(pv, mv, vv, g, osrl_adam::Coef{b1, b2, eps, step_size, bc2s});
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    f32x4 tv = pv;
    if (tgt) {
#pragma unroll
      for (int k = 0; k < 4; ++k) tv[k] = osrl_adam::polyak1(tau, pv[k], tv0[k]);
      reinterpret_cast<f32x4*>(tgt)[i] = tv;
    }
    if (pk.map_f) {
      const int mfa[4] = {mf[0], mf[1], mf[2], mf[3]};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (mfa[k] >= 0) {
          pk.pf[mfa[k]] = pv[k];
          if (tgt && pk.tf) pk.tf[mfa[k]] = tv[k];
        }
      if (pk.map_b) {
        const int mba[4] = {mb[0], mb[1], mb[2], mb[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (mba[k] >= 0) pk.pb[mba[k]] = pv[k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ tgt,
                                                   const float* __restrict__ slabs, int n_splits,
                                                   int64_t slab_stride, int64_t n4, float lr, float b1, float b2,
                                                   float eps, float wd, float tau, const float* __restrict__ gscale,
                                                   const osrl_step_state_t* __restrict__ st, const PackMap pk) {
  adam_body(p, m, v, tgt, slabs, n_splits, slab_stride, n4, lr, b1, b2, eps, wd, tau, gscale, st, pk,
            (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}
// the same step with its 36 dwords of arguments in a device-resident block (argmem.h): no hidden launch-geometry
// arguments either, so a wave's only kernarg traffic is the preloaded pointer
struct AdamArgs {
  float *p, *m, *v, *tgt;
  const float* slabs;
  int64_t slab_stride, n4, stride;
  const float* gscale;
  const osrl_step_state_t* st;
  PackMap pk;
  int32_t n_splits;
  float lr, b1, b2, eps, wd, tau;
  int32_t pad_;  // (explicit: the arena looks descriptors up by memcmp over the whole struct -- no implicit padding bytes)
};
static_assert(sizeof(AdamArgs) == 8 * 4 + 8 + 3 * 8 + 2 * 8 + sizeof(PackMap) + 8 * 4, "AdamArgs has implicit padding");
__global__ __launch_bounds__(256) void adam_kernel_p(const void* ptr) {
  OSRL_TRACE_BEGIN(11, ptr);
  const OSRL_CAS AdamArgs& a = *(const OSRL_CAS AdamArgs*)ptr;
  const PackMap pk{a.pk.map_f, a.pk.map_b, a.pk.pf, a.pk.pb, a.pk.tf};
  adam_body(a.p, a.m, a.v, a.tgt, a.slabs, a.n_splits, a.slab_stride, a.n4, a.lr, a.b1, a.b2, a.eps, a.wd, a.tau,
            a.gscale, a.st, pk, (int64_t)blockIdx.x * 256 + threadIdx.x, a.stride);
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* __restrict__ flat, const float* __restrict__ slabs,
                                                           int n_splits, int64_t slab_stride, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    reinterpret_cast<f32x4*>(flat)[i] = slab_sum(slabs, n_splits, slab_stride, i);
  }
}

// the same sum with a split count per 1024-float chunk (a group whose ranges come from plans with different row splits:
// CDT's 81920-row projections take 2-8, its 20480-row heads 32 -- summing 32 slabs of everything reads 320 MB of zeros)
__global__ __launch_bounds__(256) void reduce_slabs_counts_kernel(float* __restrict__ flat, const float* __restrict__ slabs,
                                                                  const uint8_t* __restrict__ counts, int64_t slab_stride,
                                                                  int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int ns = __builtin_amdgcn_readfirstlane((int)counts[i >> 8]);  // (a wave's 64 float4 sit in one chunk)
    reinterpret_cast<f32x4*>(flat)[i] = slab_sum(slabs, ns, slab_stride, i);
  }
}

// The Polyak step of a group ALONE (cpq.py:299-303 sync_weight, common/net.py soft update): tgt = tau p + (1 - tau) tgt and
// the packed forward copy of the targets, with the bits adam_body leaves when it carries the step (same polyak1 on the same
// fp32 values).  For a plan whose LAST reader of the old targets runs after the group's optimizer step (CPQ's OOD rows).
__global__ __launch_bounds__(256) void polyak_kernel(const float* __restrict__ p, float* __restrict__ tgt, int64_t n4,
                                                     float tau, const int32_t* __restrict__ map_f, float* __restrict__ tf) {
  OSRL_TRACE_BEGIN(16, tgt);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
    const f32x4 tv0 = reinterpret_cast<const f32x4*>(tgt)[i];
    f32x4 tv;
#pragma unroll
    for (int k = 0; k < 4; ++k) tv[k] = osrl_adam::polyak1(tau, pv[k], tv0[k]);
    reinterpret_cast<f32x4*>(tgt)[i] = tv;
    if (map_f) {
      const i32x4 mf = reinterpret_cast<const i32x4*>(map_f)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (mf[k] >= 0) tf[mf[k]] = tv[k];
    }
  }
}

inline int stream_grid(int64_t n4) {
  int64_t b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

}  // namespace

extern "C" int osrl_step_tick_peer(osrl_step_state_t* st, const osrl_step_state_t* peer, float beta1, float beta2,
                                   int32_t warmup, const float* stats_cur, float* ring, int32_t n_stats,
                                   int32_t ring_len, void* stream) {
  if (!st || peer == st) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, st, peer, beta1, beta2, warmup,
                     stats_cur, ring, n_stats, ring_len > 0 ? ring_len : 1);
  return (int)hipGetLastError();
}

extern "C" int osrl_step_tick(osrl_step_state_t* st, float beta1, float beta2, int32_t warmup,
                              const float* stats_cur, float* ring, int32_t n_stats, int32_t ring_len,
                              void* stream) {
  return osrl_step_tick_peer(st, nullptr, beta1, beta2, warmup, stats_cur, ring, n_stats, ring_len, stream);
}

static int adam_launch(float* p, float* m, float* v, float* tgt, const float* slabs, int32_t n_splits,
                       int64_t slab_stride, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, float tau, const float* gscale, const osrl_step_state_t* st,
                       const PackMap& pk, void* stream) {
  if (!p || !m || !v || !slabs || !st || n < 4 || (n & 3) || (slab_stride & 3) || n_splits < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  const int grid = stream_grid(n / 4);
  const void* dev_args = nullptr;
  if (osrl_argmem::current()) {
    AdamArgs a{};
    a.p = p; a.m = m; a.v = v; a.tgt = tgt; a.slabs = slabs;
    a.slab_stride = slab_stride; a.n4 = n / 4; a.stride = (int64_t)grid * 256;
    a.gscale = gscale; a.st = st; a.pk = pk; a.n_splits = n_splits;
    a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.tau = tau;
    dev_args = osrl_argmem::slot(a);
  }
  if (dev_args)
    hipLaunchKernelGGL(adam_kernel_p, dim3(grid), dim3(256), 0, (hipStream_t)stream, dev_args);
  else
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, m, v, tgt, slabs,
                       n_splits, slab_stride, n / 4, lr, beta1, beta2, eps, weight_decay, tau, gscale, st, pk);
  return (int)hipGetLastError();
}

extern "C" int osrl_adam_step(float* p, float* m, float* v, float* tgt, const float* slabs, int32_t n_splits,
                              int64_t slab_stride, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float tau, const float* gscale, const osrl_step_state_t* st,
                              void* stream) {
  return adam_launch(p, m, v, tgt, slabs, n_splits, slab_stride, n, lr, beta1, beta2, eps, weight_decay, tau, gscale,
                     st, PackMap{nullptr, nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int osrl_adam_step_packed(float* p, float* m, float* v, float* tgt, const float* slabs, int32_t n_splits,
                                     int64_t slab_stride, int64_t n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, float tau, const float* gscale,
                                     const osrl_step_state_t* st, const int32_t* map_f, const int32_t* map_b,
                                     float* pf, float* pb, float* tf, void* stream) {
  if (!map_f || !pf || (map_b && !pb)) return -1;
  return adam_launch(p, m, v, tgt, slabs, n_splits, slab_stride, n, lr, beta1, beta2, eps, weight_decay, tau, gscale,
                     st, PackMap{map_f, map_b, pf, pb, tf}, stream);
}

extern "C" int osrl_polyak(const float* p, float* tgt, int64_t n, float tau, const int32_t* map_f, float* tf, void* stream) {
  if (!p || !tgt || n < 4 || (n & 3) || (map_f && !tf)) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, tgt, n / 4, tau, map_f, tf);
  return (int)hipGetLastError();
}

extern "C" int osrl_reduce_slabs(float* flat, const float* slabs, int32_t n_splits, int64_t slab_stride, int64_t n,
                                 void* stream) {
  if (!flat || !slabs || n < 4 || (n & 3) || (slab_stride & 3) || n_splits < 1) return -1;
  (void)hipGetLastError();  // drop stale errors of unrelated earlier runtime calls
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, flat, slabs,
                     n_splits, slab_stride, n / 4);
  return (int)hipGetLastError();
}

extern "C" int osrl_reduce_slabs_counts(float* flat, const float* slabs, const uint8_t* counts, int64_t slab_stride,
                                        int64_t n, void* stream) {
  if (!flat || !slabs || !counts || n < 4 || (n & 3) || (slab_stride & 3)) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(reduce_slabs_counts_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, flat, slabs,
                     counts, slab_stride, n / 4);
  return (int)hipGetLastError();
}

// ---- device-resident argument blocks (argmem.h): the calling thread's arena ------------------------------------

namespace osrl_argmem {
static thread_local Arena g_arena = {nullptr, nullptr, 0, 0, kOff, 0, 0, 0};
Arena* current() { return g_arena.mode == kOff ? nullptr : &g_arena; }
}  // namespace osrl_argmem

extern "C" int osrl_args_begin(void* host_staging, const void* dev_copy, int64_t capacity, int64_t used, int32_t mode) {
  using namespace osrl_argmem;
  if (!host_staging || capacity < 64 || used < 0 || used > capacity || (mode != kRecord && mode != kReplay)) return -1;
  if (mode == kReplay && !dev_copy) return -1;
  if (g_arena.mode != kOff) return -2;  // no nesting
  g_arena = Arena{(char*)host_staging, (const char*)dev_copy, capacity, used, mode, 0, 0, 0};
  return 0;
}

extern "C" int osrl_args_end(int64_t* used, int32_t* n_blocks, int32_t* n_hits, int32_t* n_misses) {
  using namespace osrl_argmem;
  if (g_arena.mode == kOff) return -1;
  if (used) *used = g_arena.used;
  if (n_blocks) *n_blocks = g_arena.n_blocks;
  if (n_hits) *n_hits = g_arena.n_hits;
  if (n_misses) *n_misses = g_arena.n_misses;
  g_arena.mode = kOff;
  return 0;
}
