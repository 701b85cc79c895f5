// Dataset ingestion on device: what the reference does once per run in host numpy/python loops
// before training -- episode split, return-to-go / cost-to-go (osrl/common/dataset.py:19-27,137-183), the BC
// trajectory filters (dataset.py:30-134) and CDT's cost-weighted trajectory sampling probabilities
// (dataset.py:439-459) -- so the flat DSRL arrays are uploaded once and everything that feeds the on-device
// samplers (osrl_replay_gather / osrl_seq_window_gather) is produced in HBM.
//
// All of it is integer / byte-shaped streaming work (flags, prefix sums, stable compaction, row gathers) plus one
// short fp32 recurrence per episode; nothing here is GEMM-shaped.  Index results are exact; the fp32 recurrences
// use the reference's operation order with explicitly rounded multiply and add (no FMA contraction), so returns
// are bit-identical to numpy's.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/osrl_amd.h"

namespace {

constexpr int kScanThreads = 1024;
constexpr int kScanItems = 4;  // per thread
constexpr int kScanTile = kScanThreads * kScanItems;

// exclusive prefix sum of one int per thread over a 1024-thread block; *total (optional) = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* wsum /*[16] LDS*/) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int base = 0, all = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
    const int s = wsum[i];
    if (i < wv) base += s;
    all += s;
  }
  __syncthreads();
  if (total) *total = all;
  return base + incl - v;
}

struct DoneFlag {  // an episode ends at i (dataset.py:60, :165)
  const float* terminals;
  const float* timeouts;
  __device__ __forceinline__ int operator()(int64_t i) const {
    return ((terminals && terminals[i] == 1.f) || (timeouts && timeouts[i] == 1.f)) ? 1 : 0;
  }
};

struct KeepFlag {  // process_bc_dataset's transition selection (dataset.py:108-124)
  const float* cr;
  int mode;
  float t0, t1;
  __device__ __forceinline__ int operator()(int64_t i) const {
    const float c = cr[i];
    switch (mode) {
      case OSRL_BC_ALL: return 1;
      case OSRL_BC_SAFE: return c <= t0;
      case OSRL_BC_RISKY: return c >= t0;
      default: return t0 < c && c <= t1;  // boundary
    }
  }
};

template <class F>
__global__ __launch_bounds__(kScanThreads) void scan_tile_sums(F f, int64_t n, int32_t* __restrict__ bsum) {
  __shared__ int wsum[16];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) v += f(base + j);
  int tot;
  (void)block_excl_scan(v, &tot, wsum);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// in-place exclusive scan of the tile sums by one block (carry across 1024-wide chunks); total -> *total
__global__ __launch_bounds__(kScanThreads) void scan_tile_offsets(int32_t* __restrict__ bsum, int nb,
                                                                  int32_t* __restrict__ total) {
  __shared__ int wsum[16];
  int carry = 0;
  for (int c0 = 0; c0 < nb; c0 += kScanThreads) {
    const int i = c0 + threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, &tot, wsum);
    if (i < nb) bsum[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total = carry;
}

// flagged positions, in order: pos[rank(i)] = i  (stable compaction)
template <class F>
__global__ __launch_bounds__(kScanThreads) void scan_scatter(F f, int64_t n, const int32_t* __restrict__ boff,
                                                             int64_t* __restrict__ pos) {
  __shared__ int wsum[16];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int fl[kScanItems];
  int v = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    fl[j] = base + j < n ? f(base + j) : 0;
    v += fl[j];
  }
  int r = boff[blockIdx.x] + block_excl_scan(v, nullptr, wsum);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (fl[j]) pos[r++] = base + j;
}

template <class F>
int flagged_positions(F f, int64_t n, int64_t* pos, int32_t* count, int32_t* ws, hipStream_t s) {
  const int nb = (int)((n + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_tile_sums<F>, dim3(nb), dim3(kScanThreads), 0, s, f, n, ws);
  hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(kScanThreads), 0, s, ws, nb, count);
  hipLaunchKernelGGL(scan_scatter<F>, dim3(nb), dim3(kScanThreads), 0, s, f, n, ws, pos);
  return (int)hipGetLastError();
}

// episode e = [end[e-1]+1 .. end[e]]
__global__ void episode_bounds_kernel(const int64_t* __restrict__ ep_end, const int32_t* __restrict__ n_ep,
                                      int64_t* __restrict__ ep_start, int32_t* __restrict__ ep_len) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= *n_ep) return;
  const int64_t s = e == 0 ? 0 : ep_end[e - 1] + 1;
  ep_start[e] = s;
  ep_len[e] = (int32_t)(ep_end[e] - s + 1);
}

// dataset.py:19-27 per episode, back to front: c[t] = x[t] + gamma * c[t+1] with one rounded multiply and one
// rounded add (numpy evaluates exactly that in fp32); one thread per episode keeps the order.
__global__ void episode_returns_kernel(const float* __restrict__ x, const int64_t* __restrict__ ep_start,
                                       const int32_t* __restrict__ ep_len, int n_ep, float gamma, int reverse,
                                       int broadcast_first, float* __restrict__ out, float* __restrict__ x_out) {
#pragma clang fp contract(off)  // hipcc contracts a*b+c into an FMA by default, even through __fmul_rn/__fadd_rn
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_ep) return;
  const int64_t s = ep_start[e];
  const int len = ep_len[e];
  float acc = 0.f;
  for (int t = len - 1; t >= 0; --t) {
    float v = x[s + t];
    if (reverse) v = __fsub_rn(1.f, v);  // cost_reverse: 1 - c (dataset.py:161-162)
    if (x_out) x_out[s + t] = v;
    if (t == len - 1) {
      acc = v;
    } else {
      const float ga = gamma * acc;
      acc = v + ga;
    }
    if (!broadcast_first) out[s + t] = acc;
  }
  if (broadcast_first)  // process_bc_dataset: every transition carries the episode's return (dataset.py:67-68)
    for (int t = 0; t < len; ++t) out[s + t] = acc;
}

// dataset.py:439-459: p_e = max(T(cost_returns[first step of e]), 0) / sum; cdf = running sum / sum (fp64 sums)
__global__ __launch_bounds__(kScanThreads) void cost_sample_prob_kernel(const float* __restrict__ cost_returns,
                                                                       const int64_t* __restrict__ ep_start, int n_ep,
                                                                       int kind, float a, float b,
                                                                       float* __restrict__ prob,
                                                                       float* __restrict__ cdf) {
  __shared__ double part[kScanThreads];
  const int chunk = (n_ep + kScanThreads - 1) / kScanThreads;
  const int e0 = threadIdx.x * chunk, e1 = min(n_ep, e0 + chunk);
  auto weight = [&](int e) {
    const float c = cost_returns[ep_start[e]];
    const float w = kind == OSRL_COST_AFFINE ? __fadd_rn(__fmul_rn(a, c), b) : __fdiv_rn(1.f, __fadd_rn(c, b));
    return w < 0.f ? 0.f : w;
  };
  double mine = 0.0;
  for (int e = e0; e < e1; ++e) mine += (double)weight(e);
  part[threadIdx.x] = mine;
  __syncthreads();
  for (int o = 1; o < kScanThreads; o <<= 1) {  // inclusive Hillis-Steele scan in LDS
    const double t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0.0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  const double total = part[kScanThreads - 1];
  double run = part[threadIdx.x] - mine;
  for (int e = e0; e < e1; ++e) {
    const float w = weight(e);
    run += (double)w;
    prob[e] = (float)((double)w / total);
    if (cdf) cdf[e] = (float)(run / total);
  }
}

// compute_start_index_sample_prob (dataset.py:472-494), one workgroup per trajectory: n = sum(costs), l = len,
// x = 100 if prob*l - n <= 0 else n(1-prob)/(prob*l - n), x = 1 if x <= 0; w[i] = sum_{|j|<=10} costs[i+j] *
// exp(-(j*j)/10) + x (np.convolve with gauss_kernel(10, 10), its 10-sample skirts cut off); p = w / sum(w).
// fp64 like numpy (costs are fp32 inputs, the kernel and all sums are fp64 there); cdf = inclusive running sum.
__global__ __launch_bounds__(kScanThreads) void start_index_prob_kernel(const float* __restrict__ costs,
                                                                       const int64_t* __restrict__ ep_start,
                                                                       const int32_t* __restrict__ ep_len, double prob,
                                                                       float* __restrict__ p_out,
                                                                       float* __restrict__ cdf_out) {
  __shared__ double part[kScanThreads];
  __shared__ double kern[21];
  const int e = blockIdx.x;
  const int64_t s = ep_start[e];
  const int len = ep_len[e];
  const float* __restrict__ c = costs + s;
  if (threadIdx.x < 21) {
    const double x = (double)((int)threadIdx.x - 10);
    kern[threadIdx.x] = exp(-(x * x / 10.0));
  }
  const int chunk = (len + kScanThreads - 1) / kScanThreads;
  const int i0 = threadIdx.x * chunk, i1 = min(len, i0 + chunk);
  auto block_sum = [&](double mine) {  // inclusive scan of the per-thread partials; returns the total
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < kScanThreads; o <<= 1) {
      const double t = (int)threadIdx.x >= o ? part[threadIdx.x - o] : 0.0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    return part[kScanThreads - 1];
  };
  double mine = 0.0;
  for (int i = i0; i < i1; ++i) mine += (double)c[i];
  const double n = block_sum(mine);
  __syncthreads();
  const double den = prob * (double)len - n;
  double x = den <= 0.0 ? 100.0 : n * (1.0 - prob) / den;
  if (x <= 0.0) x = 1.0;
  auto weight = [&](int i) {
    double w = 0.0;
    // np.convolve(costs, kernel)[10 + i] = sum_m costs[m] * kernel[10 + i - m]
    for (int j = -10; j <= 10; ++j) {
      const int m = i + j;
      if (m >= 0 && m < len) w += (double)c[m] * kern[10 - j];
    }
    return w + x;
  };
  mine = 0.0;
  for (int i = i0; i < i1; ++i) mine += weight(i);
  const double total = block_sum(mine);
  double run = part[threadIdx.x] - mine;
  for (int i = i0; i < i1; ++i) {
    const double w = weight(i);
    run += w;
    if (p_out) p_out[s + i] = (float)(w / total);
    if (cdf_out) cdf_out[s + i] = (float)(run / total);
  }
}

// dst[j, :width] = src[idx[j], :width] (+ one appended column extra[idx[j]]: BC multi-task, dataset.py:128-130)
__global__ void gather_rows_kernel(const float* __restrict__ src, int width, const int64_t* __restrict__ idx,
                                   int64_t n_rows, float* __restrict__ dst, int dst_ld,
                                   const float* __restrict__ extra) {
  const int cols = width + (extra ? 1 : 0);
  const int64_t total = n_rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const int64_t s = idx[r];
    dst[r * dst_ld + c] = c < width ? src[s * width + c] : extra[s];
  }
}

inline int grid_for(int64_t n, int threads) {
  const int64_t g = (n + threads - 1) / threads;
  return (int)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

}  // namespace

#define S ((hipStream_t)stream)

extern "C" int64_t osrl_ingest_ws_elems(int64_t n) { return (n + kScanTile - 1) / kScanTile + 8; }

extern "C" int osrl_episode_segments(const float* terminals, const float* timeouts, int64_t n, int64_t* ep_end,
                                     int64_t* ep_start, int32_t* ep_len, int32_t* n_episodes, int32_t* ws,
                                     void* stream) {
  if ((!terminals && !timeouts) || n < 1 || n >= (int64_t)1 << 31 || !ep_end || !ep_start || !ep_len || !n_episodes ||
      !ws)
    return -1;
  (void)hipGetLastError();
  const int rc = flagged_positions(DoneFlag{terminals, timeouts}, n, ep_end, n_episodes, ws, S);
  if (rc) return rc;
  // at most n episodes; the bound is read on device so no host sync is needed between the two launches
  hipLaunchKernelGGL(episode_bounds_kernel, dim3(grid_for(n, 256)), dim3(256), 0, S, ep_end, n_episodes, ep_start,
                     ep_len);
  return (int)hipGetLastError();
}

extern "C" int osrl_episode_returns(const float* x, const int64_t* ep_start, const int32_t* ep_len,
                                    int32_t n_episodes, float gamma, int32_t reverse, int32_t broadcast_first,
                                    float* out, float* x_out, void* stream) {
  if (n_episodes == 0) return 0;  // no complete episode: nothing is written (out keeps the caller's zeros)
  if (!x || !ep_start || !ep_len || !out || n_episodes < 0) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(episode_returns_kernel, dim3((n_episodes + 63) / 64), dim3(64), 0, S, x, ep_start, ep_len,
                     n_episodes, gamma, reverse, broadcast_first, out, x_out);
  return (int)hipGetLastError();
}

extern "C" int osrl_cost_sample_prob(const float* cost_returns, const int64_t* ep_start, int32_t n_episodes,
                                     int32_t kind, float a, float b, float* prob, float* cdf, void* stream) {
  if (!cost_returns || !ep_start || !prob || n_episodes < 1 || (kind != OSRL_COST_AFFINE && kind != OSRL_COST_RECIPROCAL))
    return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(cost_sample_prob_kernel, dim3(1), dim3(kScanThreads), 0, S, cost_returns, ep_start, n_episodes,
                     kind, a, b, prob, cdf);
  return (int)hipGetLastError();
}

extern "C" int osrl_start_index_prob(const float* costs, const int64_t* ep_start, const int32_t* ep_len,
                                     int32_t n_episodes, double prob, float* p_out, float* cdf_out, void* stream) {
  if (!costs || !ep_start || !ep_len || n_episodes < 1 || (!p_out && !cdf_out)) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(start_index_prob_kernel, dim3(n_episodes), dim3(kScanThreads), 0, S, costs, ep_start, ep_len,
                     prob, p_out, cdf_out);
  return (int)hipGetLastError();
}

extern "C" int osrl_bc_select(const float* cost_returns, int64_t n, int32_t mode, float t0, float t1, int64_t* idx,
                              int32_t* n_keep, int32_t* ws, void* stream) {
  if (!cost_returns || n < 1 || n >= (int64_t)1 << 31 || !idx || !n_keep || !ws || mode < OSRL_BC_ALL ||
      mode > OSRL_BC_BOUNDARY)
    return -1;
  (void)hipGetLastError();
  return flagged_positions(KeepFlag{cost_returns, mode, t0, t1}, n, idx, n_keep, ws, S);
}

extern "C" int osrl_gather_rows(const float* src, int32_t width, const int64_t* idx, int64_t n_rows, float* dst,
                                int32_t dst_ld, const float* extra, void* stream) {
  if (n_rows == 0) return 0;  // an empty selection (e.g. bc_mode="risky" on a safe dataset): nothing to move
  if (!src || !idx || !dst || width < 1 || n_rows < 0 || dst_ld < width + (extra ? 1 : 0)) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_rows * (width + (extra ? 1 : 0)), 256)), dim3(256), 0, S, src,
                     width, idx, n_rows, dst, dst_ld, extra);
  return (int)hipGetLastError();
}
