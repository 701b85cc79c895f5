// Probe: one Linear + ReLU layer on 4- / 8-row tiles with v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 x 4 x 1: one
// instruction = 4 rows x 64 columns x 1 k at the full fp32 MFMA rate), against the 16-row-tile kernels of mlp.hip
// whose 256 -> 256 layer takes 13.3k cycles on one CU (profiles/r3_phase_ring_warm.txt).
//   Y[r][n] = relu(sum_k X[r][k] W[n][k] + b[n]),  W packed as mlp.hip's forward pack PF[k/4][n][k%4]
//   workgroup = RG * 4 rows, 4 waves, wave w = columns [64 w, 64 w + 64) (N = 256)
// Checks the layout assumptions (A: lane 4b + i = row i; B: lane 4b + j = column j of block b; D: VGPR i, lane 4b + j)
// against a host GEMM, then times `reps` launches at 256 and 2048 rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma4_probe.hip -o tools/mfma4_probe.bin && tools/mfma4_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int RG, int DEPTH>
__global__ __launch_bounds__(256, 1) void layer4(const float* __restrict__ X, const float* __restrict__ PF,
                                                  const float* __restrict__ bias, float* __restrict__ Y, int rows, int K,
                                                  int N, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [RG * 4][lda]
  const int lda = K + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RG * 4;
  const long long t0 = __builtin_readcyclecounter();
  const int n = wave * 64 + lane;  // this lane's output column
  const int nchunks = K >> 4;      // 16 k per chunk = 4 dwordx4 per lane
  const float* __restrict__ pb = PF + (size_t)n * 4;
  const size_t qstride = (size_t)N * 4;  // floats per k-quad
  f32x4 ring[DEPTH][4];
#pragma unroll
  for (int c = 0; c < DEPTH; ++c) {
    const int cc = c < nchunks ? c : nchunks - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) ring[c][q] = *reinterpret_cast<const f32x4*>(pb + (size_t)(cc * 4 + q) * qstride);
  }
  const float bv = bias[n];
  // stage X rows (RG * 4 x K floats): 256 threads, float4 each
  for (int idx = tid; idx < RG * 4 * (K >> 2); idx += 256) {
    const int r = idx / (K >> 2), c4 = idx - r * (K >> 2);
    const int gr = row0 + r < rows ? row0 + r : rows - 1;
    *reinterpret_cast<f32x4*>(lds + r * lda + c4 * 4) = *reinterpret_cast<const f32x4*>(X + (size_t)gr * K + c4 * 4);
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  f32x4 acc[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) acc[g] = f32x4{bv, bv, bv, bv};
  const float* arow = lds + (lane & 3) * lda;  // A: lane 4b + i holds row i (replicated over the 16 blocks)
  auto chunk = [&](auto j_c, int c) {
    constexpr int j = decltype(j_c)::value;
    f32x4 a[RG][4];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) a[g][q] = *reinterpret_cast<const f32x4*>(arow + g * 4 * lda + c * 16 + q * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < RG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g][q][t], ring[j][q][t], acc[g], 0, 0, 0);
    int cn = c + DEPTH;
    cn = cn < nchunks ? cn : nchunks - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) ring[j][q] = *reinterpret_cast<const f32x4*>(pb + (size_t)(cn * 4 + q) * qstride);
  };
  int c = 0;
  for (; c + DEPTH <= nchunks; c += DEPTH) {
    if constexpr (DEPTH >= 1) chunk(std::integral_constant<int, 0>{}, c);
    if constexpr (DEPTH >= 2) chunk(std::integral_constant<int, 1>{}, c + 1);
    if constexpr (DEPTH >= 3) chunk(std::integral_constant<int, 2>{}, c + 2);
    if constexpr (DEPTH >= 4) chunk(std::integral_constant<int, 3>{}, c + 3);
    if constexpr (DEPTH >= 5) chunk(std::integral_constant<int, 4>{}, c + 4);
    if constexpr (DEPTH >= 6) chunk(std::integral_constant<int, 5>{}, c + 5);
    if constexpr (DEPTH >= 7) chunk(std::integral_constant<int, 6>{}, c + 6);
    if constexpr (DEPTH >= 8) chunk(std::integral_constant<int, 7>{}, c + 7);
  }
  if (c < nchunks) {  // (not taken when DEPTH divides K / 16)
    if (c + 0 < nchunks) chunk(std::integral_constant<int, 0>{}, c + 0);
    if constexpr (DEPTH >= 2) if (c + 1 < nchunks) chunk(std::integral_constant<int, 1>{}, c + 1);
    if constexpr (DEPTH >= 3) if (c + 2 < nchunks) chunk(std::integral_constant<int, 2>{}, c + 2);
    if constexpr (DEPTH >= 4) if (c + 3 < nchunks) chunk(std::integral_constant<int, 3>{}, c + 3);
    if constexpr (DEPTH >= 5) if (c + 4 < nchunks) chunk(std::integral_constant<int, 4>{}, c + 4);
    if constexpr (DEPTH >= 6) if (c + 5 < nchunks) chunk(std::integral_constant<int, 5>{}, c + 5);
    if constexpr (DEPTH >= 7) if (c + 6 < nchunks) chunk(std::integral_constant<int, 6>{}, c + 6);
  }
  const long long t2 = __builtin_readcyclecounter();
  // D: VGPR i = row i, lane = column
#pragma unroll
  for (int g = 0; g < RG; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gr = row0 + g * 4 + i;
      if (gr < rows) Y[(size_t)gr * N + n] = fmaxf(acc[g][i], 0.f);
    }
  const long long t3 = __builtin_readcyclecounter();
  if (stamps && lane == 0 && blockIdx.x < 64) {
    long long* s = stamps + (blockIdx.x * 4 + wave) * 4;
    s[0] = t1 - t0;
    s[1] = t2 - t1;
    s[2] = t3 - t2;
  }
}

template <int RG, int DEPTH>
static void run(const char* name, int rows, int K, int N, const float* dX, const float* dPF, const float* dB, float* dY,
                const std::vector<float>& ref, long long* dst) {
  const int grid = (rows + RG * 4 - 1) / (RG * 4);
  const size_t ldsb = sizeof(float) * RG * 4 * (K + 4);
  hipLaunchKernelGGL((layer4<RG, DEPTH>), dim3(grid), dim3(256), ldsb, 0, dX, dPF, dB, dY, rows, K, N, dst);
  (void)hipDeviceSynchronize();
  std::vector<float> y((size_t)rows * N);
  (void)hipMemcpy(y.data(), dY, y.size() * 4, hipMemcpyDeviceToHost);
  double err = 0;
  for (size_t i = 0; i < y.size(); ++i) err = fmax(err, fabs((double)y[i] - ref[i]));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int reps = 200;
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL((layer4<RG, DEPTH>), dim3(grid), dim3(256), ldsb, 0, dX, dPF, dB, dY, rows, K, N, dst);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long st[64 * 4 * 4];
  (void)hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost);
  const int nw = (grid < 64 ? grid : 64) * 4;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int i = 0; i < nw; ++i) {
    s0 += st[i * 4 + 0];
    s1 += st[i * 4 + 1];
    s2 += st[i * 4 + 2];
  }
  printf("%-14s rows=%5d grid=%4d  max|err|=%.2e  %.2f us/launch (back to back)  cycles: stage %.0f  k-loop %.0f  store %.0f\n", name,
         rows, grid, err, ms * 1000.0 / reps, s0 / nw, s1 / nw, s2 / nw);
}

int main() {
  const int K = 256, N = 256, R = 2048;
  std::vector<float> X((size_t)R * K), W((size_t)N * K), B(N), PF((size_t)K * N), ref((size_t)R * N);
  srand(1);
  for (auto& v : X) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : W) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : B) v = (float)rand() / RAND_MAX - 0.5f;
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) PF[((size_t)(k / 4) * N + n) * 4 + k % 4] = W[(size_t)n * K + k];
  for (int r = 0; r < R; ++r)
    for (int n = 0; n < N; ++n) {
      double s = B[n];
      for (int k = 0; k < K; ++k) s += (double)X[(size_t)r * K + k] * W[(size_t)n * K + k];
      ref[(size_t)r * N + n] = (float)(s > 0 ? s : 0);
    }
  float *dX, *dPF, *dB, *dY;
  long long* dst;
  (void)hipMalloc(&dX, X.size() * 4);
  (void)hipMalloc(&dPF, PF.size() * 4);
  (void)hipMalloc(&dB, B.size() * 4);
  (void)hipMalloc(&dY, ref.size() * 4);
  (void)hipMalloc(&dst, sizeof(long long) * 64 * 4 * 4);
  (void)hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dPF, PF.data(), PF.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  for (int rows : {256, 2048}) {
    run<1, 2>("4 rows depth2", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<1, 4>("4 rows depth4", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<1, 8>("4 rows depth8", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<2, 2>("8 rows depth2", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<2, 4>("8 rows depth4", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<2, 8>("8 rows depth8", rows, K, N, dX, dPF, dB, dY, ref, dst);
    run<4, 4>("16 rows depth4", rows, K, N, dX, dPF, dB, dY, ref, dst);
  }
  return 0;
}
