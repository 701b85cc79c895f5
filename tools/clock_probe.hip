// What shader clock does the GPU actually sustain?  Ratio of s_memtime (shader cycles) to s_memrealtime (100 MHz)
// measured inside (a) an idle-ish kernel, (b) a dense fp32-MFMA kernel on every CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void burn(int iters, int use_mfma, long long* out, float* sink) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (use_mfma) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 7) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
  }
}

int main() {
  long long* out;
  float* sink;
  (void)hipMalloc(&out, 64);
  (void)hipMalloc(&sink, 64);
  for (int mfma : {0, 1, 1, 1}) {
    for (int iters : {20000, 200000, 2000000}) {
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(burn, dim3(256 * 4), dim3(256), 0, 0, mfma ? iters : iters / 10, mfma, out, sink);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      long long h[2];
      (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double flops = mfma ? 256.0 * 4 * 4 * (double)iters * 8 * 2048 : 0;
      printf("mfma=%d iters=%8d: %9.3f ms  shader cycles=%lld  realtime ticks=%lld -> %.0f MHz   %.1f TF/s\n", mfma,
             iters, ms, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0), flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
