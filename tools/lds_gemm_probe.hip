// Prototype for the next mlp_fwd: can a CU keep its fp32 MFMAs fed when BOTH operands come from LDS and the weight
// k-slab is staged once per workgroup (global -> registers -> LDS, 2 barriers per 16-deep k-step)?
// One persistent workgroup per CU, 8 waves = 2 row groups x 4 column groups; wave tile = (16*NRB rows) x 64 cols.
// Layer: [BM rows, K=256] x [256, N=256]; activations "already in LDS" (as in the fused kernel); `tiles` repeats.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NRB>
__global__ __launch_bounds__(512) void lds_layer(const float* __restrict__ P, int tiles, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 32 * NRB, K = 256, N = 256, lda = K + 8;
  constexpr int SLAB = 16 * N;  // floats per k-slab, packed [kq 0..3][n 0..255][4]
  float* A = lds;
  float* W = lds + BM * lda;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  for (int i = tid; i < BM * lda; i += 512) A[i] = (float)(i & 15) * 0.01f;
  __syncthreads();
  const float* arow = A + (wr * 16 * NRB + (lane & 15)) * lda + 4 * (lane >> 4);
  const float* wfrag = W + ((lane >> 4) * N + wc * 64 + (lane & 15)) * 4;
  f32x4 acc[NRB][4];
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0, 0, 0, 0};
  // this thread's share of a slab: 16 KB / 512 threads = 2 x 16 B
  f32x4 stg[2];
  const f32x4* Pv = reinterpret_cast<const f32x4*>(P);
  for (int t = 0; t < tiles; ++t) {
    stg[0] = Pv[tid];
    stg[1] = Pv[tid + 512];
    for (int ks = 0; ks < K / 16; ++ks) {
      __syncthreads();  // everyone finished reading the previous slab
      reinterpret_cast<f32x4*>(W)[tid] = stg[0];
      reinterpret_cast<f32x4*>(W)[tid + 512] = stg[1];
      __syncthreads();
      const int kn = ks + 1 < K / 16 ? ks + 1 : 0;
      stg[0] = Pv[kn * (SLAB / 4) + tid];  // next slab: in flight during this step's MFMAs
      stg[1] = Pv[kn * (SLAB / 4) + tid + 512];
      f32x4 a[NRB], b[4];
#pragma unroll
      for (int r = 0; r < NRB; ++r) a[r] = *reinterpret_cast<const f32x4*>(arow + r * 16 * lda + ks * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const f32x4*>(wfrag + c * 64);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int r = 0; r < NRB; ++r) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][tt], b[c][tt], acc[r][c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  if (s == 12345.f) out[0] = s;
}

template <int NRB>
void run(const float* P, float* out) {
  constexpr int BM = 32 * NRB;
  const size_t ldsb = (size_t)(BM * 264 + 16 * 256) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_layer<NRB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  const int tiles = 200;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(lds_layer<NRB>, dim3(256), dim3(512), ldsb, 0, P, 2, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(lds_layer<NRB>, dim3(256), dim3(512), ldsb, 0, P, tiles, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * tiles * 2.0 * BM * 256 * 256;
  printf("BM=%3d (8 waves, wave tile %dx64), LDS %zu KB: %8.3f ms  %6.1f TF/s (%.0f%% of 157.3)\n", BM, 16 * NRB, ldsb / 1024, ms,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 1.573);
}

int main() {
  float *P, *out;
  (void)hipMalloc(&P, 256 * 256 * 4);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(P, 0, 256 * 256 * 4);
  run<1>(P, out);
  run<2>(P, out);
  run<3>(P, out);
  run<4>(P, out);
  return 0;
}
