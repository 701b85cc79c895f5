// trace.h -- LAB ONLY (-DOSRL_TRACE: tools/build_trace_lib.sh -> lib/libosrl_trace.so; the product library never defines it).
// Kernel START times of an UN-PROFILED run: the first workgroup of every instrumented launch leaves (id | grid, the 100 MHz
// real-time counter) in a device ring.  Why: rocprofv3 intercepts the queues and rewrites packets, and what a cross-queue
// wait costs differs with and without it (DESIGN_LOG round 6, second session: a graph whose kernel timeline is 8 us per step
// shorter under the profiler runs 17 us per step longer without) -- timelines of the un-profiled executor need stamps
// written by the kernels themselves.  tools/trace_steps.py reads the ring.
#pragma once
#ifdef OSRL_TRACE
#include <hip/hip_runtime.h>
extern "C" void osrl_trace_register(void (*set)(unsigned long long*));  // csrc/diag.hip
namespace osrl_trace {
static __device__ unsigned long long* g_ring;  // THIS translation unit's copy of the ring pointer (no -fgpu-rdc in the build)
static void set_tu(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ring), &p, sizeof p); }
static struct Reg {
  Reg() { osrl_trace_register(&set_tu); }
} reg_;
// ring[0] = cursor, ring[1] = capacity in records, then (id | gridDim.x << 8 | gridDim.y << 28, ticks, site) triples;
// site = an address that names the launch SITE (the launch's device-resident descriptor, or its first pointer argument)
__device__ __forceinline__ void begin(unsigned id, const void* site) {
  if ((threadIdx.x | threadIdx.y | blockIdx.x | blockIdx.y | blockIdx.z) == 0) {
    unsigned long long* r = g_ring;
    if (r) {
      const unsigned long long i = atomicAdd(&r[0], 1ull);
      if (i < r[1]) {
        r[2 + 3 * i] = (unsigned long long)id | ((unsigned long long)gridDim.x << 8) | ((unsigned long long)gridDim.y << 28);
        r[3 + 3 * i] = __builtin_amdgcn_s_memrealtime();
        r[4 + 3 * i] = (unsigned long long)(uintptr_t)site;
      }
    }
  }
}
}  // namespace osrl_trace
#define OSRL_TRACE_BEGIN(id, site) osrl_trace::begin(id, (const void*)(site))
#else
#define OSRL_TRACE_BEGIN(id, site)
#endif
