// Probe (gfx950): sustained issue rate of the bf16 MFMAs against the f32-input MFMA, and whether ordinary vector
// instructions overlap with them.  Background: v_mfma_f32_16x16x4_f32 runs ON the vector ALUs at the packed-fp32 rate
// (DESIGN_LOG round 4): every kernel of this package is bound by that pipe.  A product of two fp32 numbers split EXACTLY
// into three bf16 pieces each (truncation splits: a = a1 + a2 + a3) is six bf16 MFMAs (the three dropped cross terms are
// <= 2^-23 |a||b|); at 16x the f32 rate that would be 2.67x -- if the rate holds under load and the split's vector
// instructions run beside the matrix unit.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_probe.hip -o /tmp/mfma_bf16_probe && /tmp/mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

union Frag { bf16x8 v; unsigned u[4]; };

// MODE 0: v_mfma_f32_16x16x4_f32   1: v_mfma_f32_16x16x32_bf16   2: v_mfma_f32_32x32x16_bf16
// VALU: ordinary vector instructions (v_fma_f32 on private registers) issued per MFMA
template <int MODE, int NACC, int VALU>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a0) {
  f32x4 acc4[NACC];
  f32x16 acc16[MODE == 2 ? NACC : 1];
  for (int i = 0; i < NACC; ++i) acc4[i] = f32x4{0, 0, 0, 0};
  if (MODE == 2)
    for (int i = 0; i < NACC; ++i)
      for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
  Frag fa, fb;
  for (int k = 0; k < 4; ++k) { fa.u[k] = 0x3f803f80u + threadIdx.x; fb.u[k] = 0x3f803f80u; }
  float a = a0 + threadIdx.x, b = 1.0f;
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = a0 * k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
      if (MODE == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc4[i], 0, 0, 0);
      if (MODE == 2) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc16[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < VALU; ++v) x[v & 7] = __builtin_fmaf(x[v & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc4[i][0] + acc4[i][3];
  if (MODE == 2)
    for (int i = 0; i < NACC; ++i) s += acc16[i][0] + acc16[i][15];
  for (int k = 0; k < 8; ++k) s += x[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NACC, int VALU>
static void run(const char* name, float* d, int wg_per_cu) {
  const int iters = 20000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<MODE, NACC, VALU>), dim3(grid), dim3(256), 0, 0, d, 100, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NACC, VALU>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double fl = MODE == 0 ? 2.0 * 16 * 16 * 4 : MODE == 1 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
  const double n_mfma = (double)grid * 4 * iters * NACC;
  const double cyc = best * 1e-3 * 2.4e9 / ((double)iters * NACC * wg_per_cu);  // SIMD cycles per MFMA at 2.4 GHz (nominal)
  printf("%-44s %8.3f ms  %8.1f TFLOP/s  %6.1f cycles/MFMA/SIMD (2.4 GHz nominal)\n", name, best, n_mfma * fl / best * 1e-9, cyc);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0, 8, 0>("f32 16x16x4, 1 wave/SIMD", d, 1);
  run<0, 8, 0>("f32 16x16x4, 2 waves/SIMD", d, 2);
  run<0, 8, 4>("f32 16x16x4 + 4 v_fma per MFMA, 1 wave/SIMD", d, 1);
  run<1, 8, 0>("bf16 16x16x32, 1 wave/SIMD", d, 1);
  run<1, 8, 0>("bf16 16x16x32, 2 waves/SIMD", d, 2);
  run<1, 8, 2>("bf16 16x16x32 + 2 v_fma per MFMA, 1 wave", d, 1);
  run<1, 8, 3>("bf16 16x16x32 + 3 v_fma per MFMA, 1 wave", d, 1);
  run<1, 8, 4>("bf16 16x16x32 + 4 v_fma per MFMA, 1 wave", d, 1);
  run<1, 8, 4>("bf16 16x16x32 + 4 v_fma per MFMA, 2 waves", d, 2);
  run<2, 4, 0>("bf16 32x32x16, 1 wave/SIMD", d, 1);
  run<2, 4, 0>("bf16 32x32x16, 2 waves/SIMD", d, 2);
  run<2, 4, 4>("bf16 32x32x16 + 4 v_fma per MFMA, 1 wave", d, 1);
  run<2, 4, 7>("bf16 32x32x16 + 7 v_fma per MFMA, 1 wave", d, 1);
  run<2, 4, 8>("bf16 32x32x16 + 8 v_fma per MFMA, 2 waves", d, 2);
  return 0;
}
