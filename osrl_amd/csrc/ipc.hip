// ipc.hip -- the data-parallel step's exchanges WITHOUT collective-library launches.
//
// The reference has no distributed code; what a data-parallel train step exchanges is what a single
// device would have reduced over the batch: per optimizer phase the flat gradient of a group (0.3-1.6 MB),
// CPQ's N*B KL values (80 KB per rank) and a few statistics -- every message latency-bound.  An RCCL launch costs 16-19 us
// on ONE rank (profiles/r5_bench_c2_forced_dp.json), four of them per CPQ step on the critical chain.  Here every rank owns
// a PUBLISHED buffer (two halves, used alternately) and a control block in its device memory, exported with
// hipIpcGetMemHandle and mapped by every peer.  One exchange = ONE kernel launch per rank:
//     P1  copy the local segments into the rank's published half (sequence number s = own flag + 1, half = s & 1);
//     rel every wave waits for its stores, one thread per workgroup writes the L2 back (system scope) and signs in at
//         the arrival counter; the LAST workgroup publishes flag = s;
//     P2  one thread per workgroup waits until EVERY rank's flag has reached s (bounded poll; system-scope acquire);
//     P3  all-reduce: dst[i] = sum over ranks IN RANK ORDER of their published values -- every rank adds the same numbers
//         in the same order, so replicas stay bit-identical; all-gather: dst[r * n + i] = rank r's value.
// Two halves suffice: a rank leaves exchange s only after every peer has published s, and a peer publishes s + 1 only after
// it finished reading s -- nobody can reach s + 2 (which reuses half s & 1) while somebody still reads half s & 1.
// The launch is an ordinary kernel: asynchronous on the caller's stream, hipGraph-capturable, no host state; all ranks must
// issue the same sequence of exchanges (as with any collective).  Measured with two PROCESSES on one MI355X
// (tools/ipc_slab_lab.hip, profiles/r6_ipc_slab_lab.txt): flag one way 0.45 us, "both published -> summed" 4.7-5.0 us at 64
// workgroups for 1.6 MB -- the mechanism, not xGMI.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/osrl_amd.h"

namespace {

constexpr int kThreads = 256;
constexpr int kSpinMax = 1 << 22;  // x ~0.5 us per system-scope poll: ~2 s, then the error word is set and the launch goes on

__device__ __forceinline__ unsigned ld_sys(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_sys(uint32_t* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct IpcArgs {
  float* pub[OSRL_IPC_MAX_WORLD];
  uint32_t* ctl[OSRL_IPC_MAX_WORLD];  // [0] flag = last published sequence number  [1] arrivals  [2] error  [3] exchanges done
  int64_t half;                       // floats per published half
  float* seg[OSRL_IPC_MAX_SEG];       // local tensors (all-reduce: in place; all-gather: seg[0] = source, seg[1] = destination)
  int64_t len[OSRL_IPC_MAX_SEG], off[OSRL_IPC_MAX_SEG];
  int64_t total;                      // floats this rank publishes
  int32_t world, rank, n_seg, gather;
  // a segment may be the first of n_slabs split-K gradient slabs `sstride` floats apart: the rank's own slab sum (slab
  // order, the fixed order of osrl_reduce_slabs / the Adam kernel) is formed WHILE publishing -- one launch fewer per group
  int32_t n_slabs[OSRL_IPC_MAX_SEG];
  int64_t sstride[OSRL_IPC_MAX_SEG];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// PUBLISHED data never rests in an L2: it is written with system-scope write-through stores (sc0 sc1: complete at the memory
// side, tracked by vmcnt) and read with system-scope loads (sc0 sc1: served from memory, not from this XCD's L2).  The first
// version of this kernel used plain accesses between an L2 write-back and an L2 invalidate per workgroup: the exchange
// itself was as fast (17 / 16 / 11 / 13 us for the CPQ step's four against RCCL's 16 / 17 / 19 / 17 on one rank), but 64
// write-back + invalidate pairs per exchange slowed the kernels of the OTHER graph branch -- the forced-data-parallel C2
// step ran 1860 steps/s against 2000 with RCCL (profiles/r6_ipc_wgs_sweep.txt).  The 16-byte forms are inline assembly
// (the atomic builtins stop at 8 bytes); their completion is waited for explicitly: the compiler's counter bookkeeping
// does not see them.
__device__ __forceinline__ void st16_sys(f32x4* p, const f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void ld16_sys_x4(const f32x4* p0, const f32x4* p1, const f32x4* p2, const f32x4* p3, f32x4& v0,
                                            f32x4& v1, f32x4& v2, f32x4& v3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
      "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
}
__device__ __forceinline__ float ld_sys_f(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ void st_sys_f(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// local -> published: pub[0..n) = src[0..n) (four float4 per lane in flight where both ends are 16-byte aligned)
// (n_slabs > 1: src is slab 0 of n_slabs gradient slabs sstride floats apart; what is published is their sum in slab order)
__device__ __forceinline__ void publish_seg(float* pub, const float* __restrict__ src, int64_t n, int64_t i0, int64_t stride,
                                            int n_slabs, int64_t sstride) {
  const bool vec = ((reinterpret_cast<uintptr_t>(pub) | reinterpret_cast<uintptr_t>(src)) & 15) == 0 && (sstride & 3) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  const f32x4* __restrict__ s4 = reinterpret_cast<const f32x4*>(src);
  f32x4* d4 = reinterpret_cast<f32x4*>(pub);
  const int64_t ss4 = sstride >> 2;
  for (int64_t i = i0; i < n4; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = i + k * stride;
      v[k] = s4[j < n4 ? j : i];
    }
    for (int sl = 1; sl < n_slabs; ++sl) {
      f32x4 w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t j = i + k * stride;
        w[k] = s4[(j < n4 ? j : i) + sl * ss4];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += w[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t j = i + k * stride;
      if (j < n4) st16_sys(d4 + j, v[k]);
    }
  }
  for (int64_t i = 4 * n4 + i0; i < n; i += stride) {
    float acc = src[i];
    for (int sl = 1; sl < n_slabs; ++sl) acc += src[i + sl * sstride];
    st_sys_f(pub + i, acc);
  }
}

// published (any rank's) -> local: dst[0..n) = pub[0..n)
__device__ __forceinline__ void fetch_seg(float* __restrict__ dst, const float* pub, int64_t n, int64_t i0, int64_t stride) {
  const bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(pub)) & 15) == 0;
  const int64_t n4 = vec ? n >> 2 : 0;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(pub);
  f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dst);
  for (int64_t i = i0; i < n4; i += 4 * stride) {
    int64_t j[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) j[k] = i + k * stride < n4 ? i + k * stride : i;
    f32x4 v0, v1, v2, v3;
    ld16_sys_x4(s4 + j[0], s4 + j[1], s4 + j[2], s4 + j[3], v0, v1, v2, v3);
    d4[j[0]] = v0;
    if (j[1] != i) d4[j[1]] = v1;
    if (j[2] != i) d4[j[2]] = v2;
    if (j[3] != i) d4[j[3]] = v3;
  }
  for (int64_t i = 4 * n4 + i0; i < n; i += stride) dst[i] = ld_sys_f(pub + i);
}

__global__ __launch_bounds__(kThreads) void ipc_exchange_kernel(const IpcArgs a) {
  __shared__ unsigned s_seq;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  uint32_t* my = a.ctl[a.rank];
  // every workgroup reads the OLD flag before anyone can publish the new one (publishing needs all arrivals)
  if (tid == 0) s_seq = ld_sys(&my[0]) + 1u;
  __syncthreads();
  const unsigned seq = s_seq;
  const int64_t h = (int64_t)(seq & 1u) * a.half;
  const int64_t stride = (int64_t)gridDim.x * kThreads, i0 = (int64_t)blockIdx.x * kThreads + tid;
  // ---- P1: publish (float4 x 4 in flight per lane where the segment is 16-byte aligned: a 1.56 MB gradient is two
  // rounds of the 64 workgroups; one float per lane and iteration made the exchange 36 us long inside the step)
  float* __restrict__ mine = a.pub[a.rank] + h;
  const int n_src = a.gather ? 1 : a.n_seg;
  for (int s = 0; s < n_src; ++s) publish_seg(mine + a.off[s], a.seg[s], a.len[s], i0, stride, a.n_slabs[s], a.sstride[s]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // release: this wave's write-through stores are complete at the memory side
  __syncthreads();
  if (tid == 0) {
    const unsigned seen = __hip_atomic_fetch_add(&my[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seen == gridDim.x - 1) {
      __hip_atomic_store(&my[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      st_sys(&my[0], seq);
    }
    // ---- P2: every rank (this one included) has published `seq`
    int ok = 1;
    for (int r = 0; r < a.world && ok; ++r) {
      int polls = 0;
      while ((int)(ld_sys(&a.ctl[r][0]) - seq) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++polls > kSpinMax) {
          ok = 0;
          st_sys(&my[2], 1u + (unsigned)r);  // which peer never arrived (osrl_ipc_status)
          break;
        }
      }
    }
    s_ok = ok;  // (no acquire fence: what follows reads the published halves with system-scope loads only)
  }
  __syncthreads();
  if (!s_ok) return;  // (the destination keeps its local values; the host reads the error word at its next sync point)
  // ---- P3
  if (a.gather) {
    float* __restrict__ dst = a.seg[1];
    const int64_t n = a.len[0];
    for (int r = 0; r < a.world; ++r) fetch_seg(dst + (int64_t)r * n, a.pub[r] + h + a.off[0], n, i0, stride);
  } else {
    for (int s = 0; s < a.n_seg; ++s) {
      float* __restrict__ dst = a.seg[s];
      const int64_t o = h + a.off[s], n = a.len[s];
      const bool vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;  // (published offsets are multiples of 4 floats)
      const int64_t n4 = vec ? n >> 2 : 0;
      if (a.world == 1) {  // (one rank: the sum is the published value)
        fetch_seg(dst, a.pub[0] + o, n, i0, stride);
        continue;
      }
      for (int64_t i = i0; i < n4; i += 4 * stride) {  // four float4 per lane and rank in flight
        int64_t j[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) j[k] = i + k * stride < n4 ? i + k * stride : i;
        f32x4 acc[4];
        {
          const f32x4* p = reinterpret_cast<const f32x4*>(a.pub[0] + o);
          ld16_sys_x4(p + j[0], p + j[1], p + j[2], p + j[3], acc[0], acc[1], acc[2], acc[3]);
        }
        for (int r = 1; r < a.world; ++r) {  // rank order: the same sum on every rank
          const f32x4* p = reinterpret_cast<const f32x4*>(a.pub[r] + o);
          f32x4 v0, v1, v2, v3;
          ld16_sys_x4(p + j[0], p + j[1], p + j[2], p + j[3], v0, v1, v2, v3);
          acc[0] += v0;
          acc[1] += v1;
          acc[2] += v2;
          acc[3] += v3;
        }
        f32x4* d4 = reinterpret_cast<f32x4*>(dst);
        d4[j[0]] = acc[0];
        if (j[1] != i) d4[j[1]] = acc[1];
        if (j[2] != i) d4[j[2]] = acc[2];
        if (j[3] != i) d4[j[3]] = acc[3];
      }
      for (int64_t i = 4 * n4 + i0; i < n; i += stride) {
        float acc = ld_sys_f(a.pub[0] + o + i);
        for (int r = 1; r < a.world; ++r) acc += ld_sys_f(a.pub[r] + o + i);
        dst[i] = acc;
      }
    }
  }
  if (blockIdx.x == 0 && tid == 0) my[3] = seq;
}

int launch(const osrl_ipc_t* x, float* const* seg, const int64_t* len, int n_seg, int gather, void* stream,
           const int32_t* n_slabs = nullptr, const int64_t* sstride = nullptr) {
  if (!x || x->world < 1 || x->world > OSRL_IPC_MAX_WORLD || x->rank < 0 || x->rank >= x->world || !seg || !len ||
      n_seg < 1 || n_seg > OSRL_IPC_MAX_SEG || x->half_floats < 4)
    return -1;
  IpcArgs a{};
  for (int r = 0; r < x->world; ++r) {
    if (!x->pub[r] || !x->ctl[r]) return -1;
    a.pub[r] = x->pub[r];
    a.ctl[r] = x->ctl[r];
  }
  a.half = x->half_floats;
  a.world = x->world;
  a.rank = x->rank;
  a.n_seg = n_seg;
  a.gather = gather;
  int64_t off = 0;
  const int n_pub = gather ? 1 : n_seg;
  for (int s = 0; s < n_seg; ++s) {
    if (!seg[s] || len[s] < 1) return -1;
    a.seg[s] = seg[s];
    a.len[s] = len[s];
    a.off[s] = off;
    a.n_slabs[s] = (n_slabs && !gather && n_slabs[s] > 1) ? n_slabs[s] : 1;
    a.sstride[s] = (sstride && a.n_slabs[s] > 1) ? sstride[s] : 0;
    if (a.n_slabs[s] > 1 && a.sstride[s] < len[s]) return -1;
    if (s < n_pub) off += (len[s] + 3) & ~(int64_t)3;
  }
  if (off > x->half_floats) return OSRL_E_UNSUPPORTED;  // the caller splits the message or builds a larger exchange
  a.total = off;
  // 64 workgroups: one L2 write-back each behind the publish (the fences, not the bytes, are what an exchange costs:
  // tools/ipc_slab_lab.hip -- 4.7 us at 64 workgroups, 9.5 us at 256 for the same 1.6 MB); small messages take fewer
  int64_t want = (off + 16 * kThreads - 1) / (16 * kThreads);  // ~4 float4 per lane
  int cap = 64;
  if (const char* e = getenv("OSRL_IPC_WGS")) {  // (lab: read per launch)
    const int v = atoi(e);
    if (v >= 1 && v <= 256) cap = v;
  }
  const int grid = (int)(want < 1 ? 1 : want > cap ? cap : want);
  (void)hipGetLastError();
  hipLaunchKernelGGL(ipc_exchange_kernel, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int osrl_ipc_alloc(int64_t bytes, void** dev_ptr, void* handle64) {
  if (bytes < 16 || !dev_ptr || !handle64) return -1;
  static_assert(sizeof(hipIpcMemHandle_t) == OSRL_IPC_HANDLE_BYTES, "handle size");
  void* p = nullptr;
  // FINE-GRAINED device memory: what a peer DEVICE maps over xGMI must not be held in the reader's L2 across the flag
  // handshake -- coarse-grained (plain hipMalloc) memory is only guaranteed coherent at kernel boundaries, and both the
  // flags and the published halves are read by peers INSIDE a running kernel.  (Same-device peers share one L2 per XCD
  // path and never saw the difference; the kernel's system-scope accesses stay as they are.  OSRL_IPC_COARSE=1: the plain
  // allocation of the first version, kept for the A/B in profiles/r6_ipc_finegrained_ab.txt.)
  static const bool coarse = [] { const char* v = getenv("OSRL_IPC_COARSE"); return v && v[0] == '1'; }();
  hipError_t e = coarse ? hipMalloc(&p, (size_t)bytes)
                        : hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return (int)e;
  }
  memcpy(handle64, &h, sizeof h);
  *dev_ptr = p;
  return 0;
}

extern "C" int osrl_ipc_open(const void* handle64, void** dev_ptr) {
  if (!handle64 || !dev_ptr) return -1;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof h);
  return (int)hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
}

extern "C" int osrl_ipc_close(void* mapped_ptr) { return mapped_ptr ? (int)hipIpcCloseMemHandle(mapped_ptr) : -1; }
extern "C" int osrl_ipc_free(void* dev_ptr) { return dev_ptr ? (int)hipFree(dev_ptr) : -1; }

extern "C" int osrl_ipc_all_reduce(const osrl_ipc_t* x, float* const* bufs, const int64_t* lens, int32_t n_bufs, void* stream) {
  return launch(x, bufs, lens, n_bufs, 0, stream);
}

extern "C" int osrl_ipc_all_reduce_slabs(const osrl_ipc_t* x, float* const* bufs, const int64_t* lens, const int32_t* n_slabs,
                                         const int64_t* slab_strides, int32_t n_bufs, void* stream) {
  return launch(x, bufs, lens, n_bufs, 0, stream, n_slabs, slab_strides);
}

extern "C" int osrl_ipc_all_gather(const osrl_ipc_t* x, const float* src, int64_t n, float* dst, void* stream) {
  if (!src || !dst || n < 1) return -1;
  float* seg[2] = {const_cast<float*>(src), dst};
  const int64_t len[2] = {n, n};
  return launch(x, seg, len, 2, 1, stream);
}

extern "C" int osrl_ipc_status(const osrl_ipc_t* x, uint32_t* words4) {
  if (!x || !words4 || x->rank < 0 || x->rank >= OSRL_IPC_MAX_WORLD || !x->ctl[x->rank]) return -1;
  return (int)hipMemcpy(words4, x->ctl[x->rank], 16, hipMemcpyDeviceToHost);  // synchronises: not for the hot path
}
