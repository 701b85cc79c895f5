// Micro-benchmark: issue rate of fp32-input MFMA on gfx950 (what is the practical roof for mlp.hip?).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 16x16x4 with distinct A/B registers per MFMA (as in the real kernel: 4 A x 4 B fragments)
__global__ __launch_bounds__(256) void k16_ab(float* out, int iters, const float* src) {
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  f32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4 + i * 1024);
    b[i] = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4 + i * 1024 + 4096);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[c][t], acc[r][c], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 4x4x1 (16 independent 4x4 blocks per instruction: with A broadcast over the blocks it is a 4-row x 64-column
// rank-1 update -- the candidate for <= 4096-row launches where 16-row tiles leave half the CUs idle)
template <int NACC>
__global__ __launch_bounds__(256) void k4(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float *out, *src;
  hipMalloc(&out, 256 * 8 * 256 * 4 * 4);
  hipMalloc(&src, 1 << 20);
  hipMemset(src, 0, 1 << 20);
  const int iters = 4000;
  for (int bpc : {1, 2, 4}) {
    const int grid = 256 * bpc;
    auto tf16 = [&](int nacc, float ms) { return 2.0 * 16 * 16 * 4 * 4.0 * nacc * iters * (grid * 4.0) / (ms * 1e-3) / 1e12; };
    auto tf32 = [&](int nacc, float ms) { return 2.0 * 32 * 32 * 2 * 4.0 * nacc * iters * (grid * 4.0) / (ms * 1e-3) / 1e12; };
    float ms;
    ms = run([&] { hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("16x16x4 acc=2  blocks/CU=%d: %7.2f TF/s\n", bpc, tf16(2, ms));
    ms = run([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("16x16x4 acc=4  blocks/CU=%d: %7.2f TF/s\n", bpc, tf16(4, ms));
    ms = run([&] { hipLaunchKernelGGL(k16<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("16x16x4 acc=16 blocks/CU=%d: %7.2f TF/s\n", bpc, tf16(16, ms));
    ms = run([&] { hipLaunchKernelGGL(k16_ab, dim3(grid), dim3(256), 0, 0, out, iters, src); });
    printf("16x16x4 4x4 frag blocks/CU=%d: %7.2f TF/s\n", bpc, tf16(16, ms));
    auto tf4 = [&](int nacc, float ms) { return 2.0 * 16 * 4 * 4 * 1 * 4.0 * nacc * iters * (grid * 4.0) / (ms * 1e-3) / 1e12; };
    ms = run([&] { hipLaunchKernelGGL(k4<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("4x4x1   acc=4  blocks/CU=%d: %7.2f TF/s\n", bpc, tf4(4, ms));
    ms = run([&] { hipLaunchKernelGGL(k4<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("4x4x1   acc=16 blocks/CU=%d: %7.2f TF/s\n", bpc, tf4(16, ms));
    ms = run([&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("32x32x2 acc=1  blocks/CU=%d: %7.2f TF/s\n", bpc, tf32(1, ms));
    ms = run([&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("32x32x2 acc=4  blocks/CU=%d: %7.2f TF/s\n", bpc, tf32(4, ms));
  }
  return 0;
}
