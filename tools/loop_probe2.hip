// loop_probe2: the k-loop of mlp_fwd_kernel<5,7> / <5,4> (80-row tiles, one 4-wave workgroup per CU) in isolation:
//   per k-step  NCB global_load_dwordx4 (packed weights, L2 resident) + NRB ds_read_b128 + 4*NRB*NCB MFMAs,
//   2-slot ring, order pinned with sched_group_barrier.  Variants: ROT = every workgroup starts its k-walk at a different
//   step (as the product kernel does), PIN = pinned order, LB1 = launch_bounds(256,1) (512 registers, AGPR accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NRB, int NCB, bool ROT, bool PIN, int MINB>
__global__ __launch_bounds__(256, MINB) void loop_k(const float* __restrict__ P, int Np, int nk, int reps, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lda = 408;
  for (int i = threadIdx.x; i < 16 * NRB * lda; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  const float* arow = lds + (lane & 15) * lda + 4 * (lane >> 4);
  const unsigned lane_off = (unsigned)(((lane >> 4) * Np + wave * 16 * NCB + (lane & 15)) * 16);
  const int rot = ROT ? (int)((blockIdx.x * 5u + wave) % (unsigned)nk) : 0;
  auto kat = [&](int k) { k += rot; return k >= nk ? k - nk : k; };
  f32x4 acc[NRB][NCB];
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0, 0, 0, 0};
  f32x4 b[2][NCB], a[2][NRB];
  for (int c = 0; c < NCB; ++c)
    b[0][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(P + (size_t)kat(0) * 16 * Np) + lane_off + c * 256);
  for (int r = 0; r < NRB; ++r) a[0][r] = *reinterpret_cast<const f32x4*>(arow + r * 16 * lda + kat(0) * 16);
  for (int rep = 0; rep < reps; ++rep) {
    for (int kb = 0; kb + 4 <= nk; kb += 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int kc = kb + s;
        int kl = kc + 1;
        kl = kat(kl < nk ? kl : 0);
        const float* __restrict__ Pk = P + (size_t)kl * 16 * Np;
#pragma unroll
        for (int c = 0; c < NCB; ++c)
          b[(s + 1) & 1][c] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(Pk) + lane_off + c * 256);
#pragma unroll
        for (int r = 0; r < NRB; ++r) a[(s + 1) & 1][r] = *reinterpret_cast<const f32x4*>(arow + r * 16 * lda + kl * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < NRB; ++r)
              acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][r][t], b[s & 1][c][t], acc[r][c], 0, 0, 0);
        if (PIN) {
          __builtin_amdgcn_sched_group_barrier(0x020, NCB, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NRB, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NRB * NCB, 0);
        }
      }
    }
  }
  float sres = 0;
  for (int r = 0; r < NRB; ++r)
    for (int c = 0; c < NCB; ++c) sres += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
  if (sres == 12345.f) out[0] = sres;
}

template <int NRB, int NCB, bool ROT, bool PIN, int MINB>
void run(const char* name, const float* P, float* out, int Np, int nks, int grid) {
  const int reps = 200;
  const size_t ldsb = (size_t)16 * NRB * 408 * 4;
  auto k = loop_k<NRB, NCB, ROT, PIN, MINB>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, P, Np, nks, 4, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, 0, P, Np, nks, reps, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * (double)reps * (nks / 4 * 4) * NRB * NCB * 4 * 2048;
  printf("%-34s NRB=%d NCB=%d rot=%d pin=%d minb=%d grid=%d Np=%d nk=%d: %8.3f ms  %6.1f TF/s (%.0f%%) err=%d\n", name, NRB, NCB,
         (int)ROT, (int)PIN, MINB, grid, Np, nks, ms, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 1.573,
         (int)hipGetLastError());
}

int main() {
  float *P, *out;
  const size_t bytes = (size_t)32 * 16 * 512 * 4;
  (void)hipMalloc(&P, bytes);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(P, 0, bytes);
  // 400-wide layer (Np = 400, 25 k-steps -> 24 used), 80-row tiles
  run<5, 7, false, false, 1>("5x7", P, out, 448, 24, 256);
  run<5, 7, false, true, 1>("5x7 pin", P, out, 448, 24, 256);
  run<5, 7, true, true, 1>("5x7 pin rot", P, out, 448, 24, 256);
  run<5, 7, true, false, 1>("5x7 rot", P, out, 448, 24, 256);
  run<5, 4, false, true, 1>("5x4 pin", P, out, 256, 16, 256);
  run<5, 4, true, true, 1>("5x4 pin rot", P, out, 256, 16, 256);
  run<5, 4, true, true, 1>("5x4 pin rot 2 rounds", P, out, 256, 16, 512);
  run<4, 4, true, true, 1>("4x4 pin rot", P, out, 256, 16, 256);
  run<4, 4, true, true, 2>("4x4 pin rot minb2 grid512", P, out, 256, 16, 512);
  run<2, 7, true, true, 2>("2x7 pin rot minb2 grid512", P, out, 448, 24, 512);
  run<2, 7, true, false, 2>("2x7 rot minb2 grid512", P, out, 448, 24, 512);
  run<2, 4, true, true, 3>("2x4 pin rot minb3 grid768", P, out, 256, 16, 768);
  run<2, 4, true, false, 3>("2x4 rot minb3 grid768", P, out, 256, 16, 768);
  run<5, 7, true, true, 1>("5x7 pin rot short k (nk=4)", P, out, 448, 4, 256);
  return 0;
}
