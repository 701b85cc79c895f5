// diag.hip -- lease diagnostics (bench.py prints them beside the headline number).
//
// osrl_kernarg_probe: WHERE does this process's HIP runtime keep kernel arguments?  A one-lane kernel reports its own
// kernarg segment address; ROCr's pointer database says which agent owns that allocation.  Host-resident kernargs
// (no large BAR, or HIP_FORCE_DEV_KERNARG=0) cost the CPQ step 3-20 % depending on how many launches still read their
// arguments themselves (profiles/r3_kernarg_ab.txt), so a slow lease can be attributed.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdint.h>

#include "../../include/osrl_amd.h"

namespace {
__global__ void kernarg_probe_kernel(uint64_t* out) {
  if (threadIdx.x == 0) out[0] = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
}
__global__ void stamp_kernel(uint64_t* out) {
  if (threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime();  // the constant 100 MHz counter (10 ns ticks)
}
}  // namespace

// osrl_stamp_realtime: one lane writes the device's 100 MHz real-time counter to out[0].  Asynchronous and
// hipGraph-capturable: bench.py brackets a launch INSIDE the captured step with two of these (on the launch's own
// stream) and reads the difference after a replay -- how long the launch takes inside the replayed graph, which HIP
// events around eagerly issued launches do not tell (the host cannot keep two queues fed: the overlap differs).
extern "C" int osrl_stamp_realtime(uint64_t* out, void* stream) {
  if (!out) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  return (int)hipGetLastError();
}

// where: 1 = device memory (a GPU agent owns the allocation), 0 = host memory, -1 = unknown to the pointer database.
// dev_scratch: 8 bytes of device memory.  Synchronises the stream (diagnostic call, not for the hot path).
extern "C" int osrl_kernarg_probe(uint64_t* dev_scratch, int32_t* where, uint64_t* address, void* stream) {
  if (!dev_scratch || !where) return -1;
  (void)hipGetLastError();
  hipLaunchKernelGGL(kernarg_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  uint64_t addr = 0;
  e = hipMemcpy(&addr, dev_scratch, sizeof addr, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  if (address) *address = addr;
  *where = -1;
  hsa_amd_pointer_info_t info;
  info.size = sizeof info;
  if (hsa_amd_pointer_info((void*)(uintptr_t)addr, &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return 0;
  if (info.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return 0;
  hsa_device_type_t dt;
  if (hsa_agent_get_info(info.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS) return 0;
  *where = dt == HSA_DEVICE_TYPE_GPU ? 1 : 0;
  return 0;
}

#ifdef OSRL_TRACE
// LAB ONLY (csrc/trace.h, tools/build_trace_lib.sh): every translation unit keeps its own copy of the ring pointer (the
// build links plain objects, no relocatable device code) and registers its setter here from a static constructor.
namespace {
typedef void (*trace_setter_t)(unsigned long long*);
struct TraceSetters {
  trace_setter_t fn[32];
  int n;
};
TraceSetters& trace_setters() {
  static TraceSetters s{};  // (function-local: ready whichever translation unit's constructor runs first)
  return s;
}
}  // namespace
extern "C" void osrl_trace_register(void (*set)(unsigned long long*)) {
  TraceSetters& s = trace_setters();
  if (s.n < 32) s.fn[s.n++] = set;
}
// ring: device memory, uint64 [2 + 3 * capacity]: ring[0] = cursor (reset to 0 here), ring[1] = capacity; NULL switches off
extern "C" int osrl_debug_trace_set(uint64_t* ring, int64_t capacity) {
  if (ring) {
    const unsigned long long head[2] = {0ull, (unsigned long long)capacity};
    hipError_t e = hipMemcpy(ring, head, sizeof head, hipMemcpyHostToDevice);
    if (e != hipSuccess) return (int)e;
  }
  TraceSetters& s = trace_setters();
  for (int i = 0; i < s.n; ++i) s.fn[i]((unsigned long long*)ring);
  return s.n;
}
#endif
