// What does the MFMA issue rate look like for the k-loop shape of mlp.hip in isolation?
//   per k-step: NRB ds_read_b128 (A fragments from an LDS tile) + NCB global_load_dwordx4 (B fragments from an
//   L2-resident packed weight buffer) + NRB*NCB*4 v_mfma_f32_16x16x4_f32, ring of 3 stages, counted vmcnt.
// Variants switch the LDS reads / global loads off; grid = 256 * blocks/CU workgroups of 4 waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NRB, int NCB, bool LDSR, bool GLD>
__global__ __launch_bounds__(256) void loop_k(const float* __restrict__ P, int Np, int nk, int reps, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lda = 264;
  for (int i = threadIdx.x; i < 16 * NRB * lda; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  const float* arow = lds + (lane & 15) * lda + 4 * (lane >> 4);
  const unsigned lane_off = (unsigned)(((lane >> 4) * Np + wave * 16 * NCB + (lane & 15)) * 16);
  f32x4 acc[NRB][NCB];
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0, 0, 0, 0};
  f32x4 b[3][NCB], a[2][NRB];
  for (int s = 0; s < 2; ++s)
    for (int c = 0; c < NCB; ++c)
      b[s][c] = GLD ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P + (size_t)s * 16 * Np) + lane_off + c * 256)
                    : f32x4{1.f, 2.f, 3.f, (float)c};
  for (int r = 0; r < NRB; ++r) a[0][r] = LDSR ? *reinterpret_cast<const f32x4*>(arow + r * 16 * lda) : f32x4{1.f, 1.f, 2.f, (float)r};
  for (int rep = 0; rep < reps; ++rep) {
    for (int kb = 0; kb + 6 <= nk; kb += 6) {
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int kc = kb + s;
        int kl = kc + 2;
        kl = kl < nk ? kl : kl - nk;
        const float* __restrict__ Pk = P + (size_t)kl * 16 * Np;
        if (GLD) {
#pragma unroll
          for (int c = 0; c < NCB; ++c)
            b[(s + 2) % 3][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Pk) + lane_off + c * 256);
        }
        if (LDSR) {
          int ka = kc + 1;
          ka = ka < nk ? ka : 0;
#pragma unroll
          for (int r = 0; r < NRB; ++r) a[(s + 1) & 1][r] = *reinterpret_cast<const f32x4*>(arow + r * 16 * lda + ka * 16);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < NRB; ++r)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][r][t], b[s % 3][c][t], acc[r][c], 0, 0, 0);
      }
    }
  }
  float sres = 0;
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < NCB; ++c) sres += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  if (sres == 12345.f) out[0] = sres;
}

template <int NRB, int NCB, bool LDSR, bool GLD>
void run(const char* name, const float* P, float* out) {
  const int Np = 256, nk = 16 * 6 / 6 * 1;  // 16 -> use 18 (multiple of 6)
  const int nks = 18, reps = 400;
  (void)nk;
  for (int bpc : {1, 2, 3}) {
    const size_t ldsb = 16 * NRB * 264 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_k<NRB, NCB, LDSR, GLD>), dim3(256 * bpc), dim3(256), ldsb, 0, P, Np, nks, 4, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop_k<NRB, NCB, LDSR, GLD>), dim3(256 * bpc), dim3(256), ldsb, 0, P, Np, nks, reps, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * bpc * 4 * (double)reps * nks * NRB * NCB * 4 * 2048;
    printf("%-28s NRB=%d NCB=%d blocks/CU=%d: %8.3f ms  %6.1f TF/s (%.0f%% of 157.3)\n", name, NRB, NCB, bpc, ms,
           flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 1.573);
  }
}

int main() {
  float *P, *out;
  (void)hipMalloc(&P, 32 * 16 * 256 * 4 * 4);  // 32 k-steps x 4 quads x 256 cols x 4 floats = 512 KB
  (void)hipMalloc(&out, 64);
  (void)hipMemset(P, 0, 32 * 16 * 256 * 4 * 4);
  run<2, 4, false, false>("mfma only", P, out);
  run<2, 4, true, false>("mfma + ds_read", P, out);
  run<2, 4, false, true>("mfma + global_load", P, out);
  run<2, 4, true, true>("mfma + ds_read + global_load", P, out);
  run<1, 4, true, true>("mfma + ds_read + global_load", P, out);
  run<4, 4, true, true>("mfma + ds_read + global_load", P, out);
  return 0;
}
