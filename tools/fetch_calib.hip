// fetch_calib.hip -- calibration of the FETCH_SIZE / WRITE_SIZE counters on a known-byte stream.
// The N*B-row kernels stage their input rows with 4-byte-per-lane loads (64 lanes x 4 B = one 256-byte request per wave);
// MI355X_MICROARCH.md calibrates FETCH_SIZE on wide streaming reads only.  Three kernels read the SAME 1 GiB buffer once
// (larger than the 256 MB Infinity Cache) with 4 / 8 / 16 bytes per lane and write one float per wave; under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// counter / known bytes is the correction factor for that access width.  Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int W>  // floats per lane
__global__ __launch_bounds__(256) void read_w(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * W;
  float acc = 0.f;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W; i + W <= n; i += stride) {
    if constexpr (W == 1) acc += in[i];
    else if constexpr (W == 2) { const float2 v = *reinterpret_cast<const float2*>(in + i); acc += v.x + v.y; }
    else { const float4 v = *reinterpret_cast<const float4*>(in + i); acc += (v.x + v.y) + (v.z + v.w); }
  }
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = acc;
}
// a pure write stream of known size: 4 bytes per lane
__global__ __launch_bounds__(256) void write_1(float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = 1.0f;
}

int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB of floats
  float *in, *out;
  if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&out, 4 << 20) != hipSuccess) return 1;
  hipMemset(in, 0, n * 4);
  hipDeviceSynchronize();
  const int grid = 256 * 8;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_w<1>, dim3(grid), dim3(256), 0, 0, in, out, n);
    hipLaunchKernelGGL(read_w<2>, dim3(grid), dim3(256), 0, 0, in, out, n);
    hipLaunchKernelGGL(read_w<4>, dim3(grid), dim3(256), 0, 0, in, out, n);
    hipLaunchKernelGGL(write_1, dim3(grid), dim3(256), 0, 0, in, n);
  }
  hipDeviceSynchronize();
  printf("known bytes per launch: read %zu, write_1 %zu; out floats per read launch %d\n", n * 4, n * 4, grid * 4);
  return hipGetLastError() != hipSuccess;
}
