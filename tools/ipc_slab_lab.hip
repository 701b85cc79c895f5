// ipc_slab_lab.hip -- two PROCESSES, one GPU: the RCCL-free gradient exchange of DESIGN.md section 7 as a toy.
//
// Each rank owns a slab buffer (its dW partials), an arrival counter and a flag word in device memory, exports them with
// hipIpcGetMemHandle and maps the peer's.  Per "step": a grid of workgroups writes the rank's slab, every workgroup signs in at
// the arrival counter behind a system-scope release, the last one publishes flag = step; then the same launch waits for the
// PEER's flag (bounded poll), sums peer slab + own slab (the fixed-order sum the Adam kernel does over split-K slabs) and
// writes the result.  Measured per rank with the device's 100 MHz counter: (a) flag ping-pong = the one-way latency of a
// system-scope flag between two processes' kernels, (b) a full exchange step against the same launch without a peer (reads
// its own slab twice) = the exposed cost of exchanging through mapped memory instead of an RCCL launch (16-19 us at world 1,
// profiles/r5_bench_c2_forced_dp.json).  On ONE GPU both mappings are local HBM behind one L2: this prices the MECHANISM
// (flags, fences, polling, a second read stream), not xGMI.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ipc_slab_lab.hip -o /tmp/ipc_slab_lab ; run: HSA_ENABLE_IPC_MODE_LEGACY=0 /tmp/ipc_slab_lab
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static int g_rank = 0;

struct Shared {  // one per rank, in ITS device memory
  unsigned flag;      // last published step
  unsigned arrive;    // workgroups signed in this step
  long long t_pub;    // 100 MHz device time at which `flag` was published (one counter for every process on the device)
  unsigned pad[12];
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
constexpr int kSpinMax = 1 << 18;  // bounded: a peer that is not resident at the same time ends the run, it does not hang the GPU

// (a) ping-pong: rank 0 publishes i and waits for the echo; rank 1 echoes.  stamps[i] = 100 MHz ticks of the round trip.
__global__ void pingpong(Shared* mine, const Shared* peer, int rank, int iters, long long* stamps, unsigned* fail) {
  for (int i = 1; i <= iters; ++i) {
    long long t0 = __builtin_amdgcn_s_memrealtime();
    if (rank == 0) {
      st_sys(&mine->flag, (unsigned)i);
      int n = 0;
      while (ld_sys(&peer->flag) != (unsigned)i) if (++n > kSpinMax) { *fail = 1; return; }
    } else {
      int n = 0;
      while (ld_sys(&peer->flag) != (unsigned)i) if (++n > kSpinMax) { *fail = 1; return; }
      st_sys(&mine->flag, (unsigned)i);
    }
    stamps[i - 1] = __builtin_amdgcn_s_memrealtime() - t0;
  }
}

// (b) one exchange step.  use_peer = 0: the same launch reading its own slab as "the peer's" (no flags): the baseline.
__global__ __launch_bounds__(256) void exchange(float* my_slab, const float* peer_slab, Shared* mine, const Shared* peer,
                                                float* out, int64_t n4, unsigned step, int use_peer, float seed,
                                                long long* stamp, unsigned* fail) {
  __shared__ int s_ok;
  if (*reinterpret_cast<volatile unsigned*>(fail)) return;  // an earlier step gave up: drain the queue quickly
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  long long t0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
  float4* ms = reinterpret_cast<float4*>(my_slab);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    ms[i] = make_float4(seed, seed + 1.f, seed + 2.f, (float)(i & 1023));  // "dW": this rank's slab
  long long t_ready = 0;
  if (use_peer) {
    // release as in the one-launch BC step (csrc/mlp.hip): every wave waits for ITS stores, ONE thread per workgroup writes
    // the L2 back (a fence per thread made the 256-workgroup launch 67 us long: 65k L2 write-backs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      const unsigned seen = atomicAdd(&mine->arrive, 1u);
      if (seen == gridDim.x - 1) {
        mine->arrive = 0;
        mine->t_pub = __builtin_amdgcn_s_memrealtime();
        __threadfence_system();
        st_sys(&mine->flag, step);
      }
      int n = 0, ok = 1;
      while (ld_sys(&peer->flag) < step) if (++n > kSpinMax) { ok = 0; *fail = 1; break; }
      __threadfence_system();  // acquire
      if (blockIdx.x == 0) {
        int m = 0;
        while (ld_sys(&mine->flag) < step) if (++m > kSpinMax) break;  // (own publish time: the last workgroup's)
        const long long a = *reinterpret_cast<volatile long long*>(&mine->t_pub), b = *reinterpret_cast<const volatile long long*>(&peer->t_pub);
        t_ready = a > b ? a : b;  // both slabs published
      }
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
  } else {
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) t_ready = __builtin_amdgcn_s_memrealtime();
  }
  const float4* ps = reinterpret_cast<const float4*>(peer_slab);
  float4* o = reinterpret_cast<float4*>(out);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = ms[i], b = ps[i];  // own slab first, then the peer's: the fixed order of the split-K sum
    o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const long long t1 = __builtin_amdgcn_s_memrealtime();
    stamp[0] = t1 - t0;            // workgroup 0's life (with a peer: includes the two processes' launch skew)
    stamp[4096] = t1 - t_ready;    // from "both slabs published" to workgroup 0's share of the sum written
  }
}

static void xfer(int wfd, int rfd, const void* mine, void* theirs, size_t n) {
  if (write(wfd, mine, n) != (ssize_t)n || read(rfd, theirs, n) != (ssize_t)n) { fprintf(stderr, "pipe\n"); exit(3); }
}
static int cmp(const void* a, const void* b) { long long x = *(const long long*)a, y = *(const long long*)b; return x < y ? -1 : x > y; }

int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  const pid_t pid = fork();  // HIP is initialised AFTER the fork, in both processes
  g_rank = pid == 0 ? 1 : 0;
  const int wfd = g_rank == 0 ? p2c[1] : c2p[1], rfd = g_rank == 0 ? c2p[0] : p2c[0];
  CK(hipSetDevice(0));
  const int64_t n = 400 * 1024;  // floats per slab: 1.6 MB, the largest gradient group of the CPQ step (VAE at C2)
  float *slab, *out, *self2;
  Shared* sh;
  long long* stamps;
  unsigned* fail;
  CK(hipMalloc(&slab, n * 4)); CK(hipMalloc(&self2, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&sh, sizeof(Shared)));
  CK(hipMalloc(&stamps, 2 * 4096 * 8)); CK(hipMalloc(&fail, 4));
  CK(hipMemset(sh, 0, sizeof(Shared))); CK(hipMemset(fail, 0, 4)); CK(hipMemset(self2, 0, n * 4));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t hs, hf, ps, pf;
  CK(hipIpcGetMemHandle(&hs, slab)); CK(hipIpcGetMemHandle(&hf, sh));
  xfer(wfd, rfd, &hs, &ps, sizeof hs);
  xfer(wfd, rfd, &hf, &pf, sizeof hf);
  float* peer_slab; Shared* peer_sh;
  CK(hipIpcOpenMemHandle((void**)&peer_slab, ps, hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void**)&peer_sh, pf, hipIpcMemLazyEnablePeerAccess));
  char go = 1, got;
  xfer(wfd, rfd, &go, &got, 1);  // both mapped
  unsigned hfail = 0;
  // ---- (a) flag ping-pong
  const int iters = 2000;
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, sh, peer_sh, g_rank, iters, stamps, fail);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
  static long long h[4096];
  CK(hipMemcpy(h, stamps, iters * 8, hipMemcpyDeviceToHost));
  if (hfail) { printf("rank %d: ping-pong gave up waiting (the two processes' kernels were not resident together)\n", g_rank); }
  else if (g_rank == 0) {
    qsort(h + 100, iters - 100, 8, cmp);
    printf("flag ping-pong between two processes, system scope: round trip median %.2f us (p10 %.2f, p90 %.2f) -> one way %.2f us\n",
           h[100 + (iters - 100) / 2] * 0.01, h[100 + (iters - 100) / 10] * 0.01, h[100 + 9 * (iters - 100) / 10] * 0.01,
           h[100 + (iters - 100) / 2] * 0.005);
  }
  xfer(wfd, rfd, &go, &got, 1);
  CK(hipMemset(sh, 0, sizeof(Shared))); CK(hipMemset(fail, 0, 4)); CK(hipDeviceSynchronize());
  xfer(wfd, rfd, &go, &got, 1);
  // ---- (b) exchange steps: grids of 64 / 256 workgroups (what a 1.6 MB Adam launch uses), with and without a peer
  for (int grid : {64, 256}) {
    for (int use_peer = 0; use_peer <= 1; ++use_peer) {
      const int steps = 300;
      unsigned base = use_peer ? (grid == 64 ? 0u : 1000u) : 0u;
      for (int s = 1; s <= steps && !hfail; ++s) {
        hipLaunchKernelGGL(exchange, dim3(grid), dim3(256), 0, 0, slab, use_peer ? peer_slab : self2, sh, peer_sh, out, n / 4,
                           base + (unsigned)s, use_peer, (float)s, stamps + s - 1, fail);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h, stamps, steps * 8, hipMemcpyDeviceToHost));
      static long long h2[4096];
      CK(hipMemcpy(h2, stamps + 4096, steps * 8, hipMemcpyDeviceToHost));
      qsort(h2 + 20, steps - 20, 8, cmp);
      if (hfail) { printf("rank %d: exchange (grid %d) gave up waiting for the peer's flag\n", g_rank, grid); break; }
      // check: out = own + peer with the LAST step's seeds
      static float ho[8];
      CK(hipMemcpy(ho, out, 32, hipMemcpyDeviceToHost));
      qsort(h + 20, steps - 20, 8, cmp);
      printf("rank %d grid %3d %s: workgroup 0's life median %.2f us (p10 %.2f, p90 %.2f); both slabs published -> summed: median %.2f us (p10 %.2f, p90 %.2f); out[0..3] = %.0f %.0f %.0f %.0f (want %d %d %d 0)\n",
             g_rank, grid, use_peer ? "slab exchange with the peer process" : "no peer (own slab twice)    ",
             h[20 + (steps - 20) / 2] * 0.01, h[20 + (steps - 20) / 10] * 0.01, h[20 + 9 * (steps - 20) / 10] * 0.01,
             h2[20 + (steps - 20) / 2] * 0.01, h2[20 + (steps - 20) / 10] * 0.01, h2[20 + 9 * (steps - 20) / 10] * 0.01, ho[0], ho[1],
             ho[2], ho[3], use_peer ? 2 * steps : steps, use_peer ? 2 * steps + 2 : steps + 1, use_peer ? 2 * steps + 4 : steps + 2);
      xfer(wfd, rfd, &go, &got, 1);
    }
    if (hfail) break;
  }
  CK(hipIpcCloseMemHandle(peer_slab)); CK(hipIpcCloseMemHandle(peer_sh));
  if (g_rank == 0) { int st; waitpid(pid, &st, 0); }
  return 0;
}
