// device side of tools/aql_probe.cpp:  hipcc --offload-arch=gfx950 --cuda-device-only -O2 -c tools/aql_probe_kernels.hip -o tools/aql_probe_kernels.hsaco
#include <hip/hip_runtime.h>
extern "C" __global__ void spin_kernel(long long ticks, int* flag, int val) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
  if (flag && threadIdx.x == 0 && blockIdx.x == 0) { __atomic_store_n(flag, val, __ATOMIC_RELEASE); }
}
extern "C" __global__ void dims_kernel(int* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) {
    out[0] = gridDim.x; out[1] = gridDim.y; out[2] = blockDim.x; out[3] = blockIdx.x; out[4] = lds[blockDim.x - 1];
  }
}
