// Does the VGPR bank of the A / B source registers (and VGPR vs AGPR accumulators) change the issue rate of
// v_mfma_f32_16x16x4_f32?  One wave per SIMD, 16 independent accumulators, explicit registers through inline asm.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bank_probe.hip -o tools/mfma_bank_probe.bin && tools/mfma_bank_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define M16(ACC, A, B)                                                                                     \
  "v_mfma_f32_16x16x4_f32 " ACC "[0:3], " A ", " B ", " ACC "[0:3]\n"                                       \
  "v_mfma_f32_16x16x4_f32 " ACC "[4:7], " A ", " B ", " ACC "[4:7]\n"                                       \
  "v_mfma_f32_16x16x4_f32 " ACC "[8:11], " A ", " B ", " ACC "[8:11]\n"                                     \
  "v_mfma_f32_16x16x4_f32 " ACC "[12:15], " A ", " B ", " ACC "[12:15]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[16:19], " A ", " B ", " ACC "[16:19]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[20:23], " A ", " B ", " ACC "[20:23]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[24:27], " A ", " B ", " ACC "[24:27]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[28:31], " A ", " B ", " ACC "[28:31]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[32:35], " A ", " B ", " ACC "[32:35]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[36:39], " A ", " B ", " ACC "[36:39]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[40:43], " A ", " B ", " ACC "[40:43]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[44:47], " A ", " B ", " ACC "[44:47]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[48:51], " A ", " B ", " ACC "[48:51]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[52:55], " A ", " B ", " ACC "[52:55]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[56:59], " A ", " B ", " ACC "[56:59]\n"                                   \
  "v_mfma_f32_16x16x4_f32 " ACC "[60:63], " A ", " B ", " ACC "[60:63]\n"

// variant 0: acc in AGPRs, A = v100 (bank 0), B = v104 (bank 0)      1: AGPR acc, A = v100, B = v105 (bank 1)
// variant 2: acc in VGPRs v[0:63], A = v100, B = v104                  3: VGPR acc, A = v100, B = v105
// variant 4: VGPR acc, A = v101 (bank 1), B = v106 (bank 2): neither shares a bank with acc dword 0
template <int V>
__global__ __launch_bounds__(256, 1) void k(int iters, float* out) {
  for (int i = 0; i < iters; ++i) {
    if (V == 0) asm volatile(M16("a", "v100", "v104") ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v100","v104","v105");
    if (V == 1) asm volatile(M16("a", "v100", "v105") ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v100","v104","v105");
#define VCLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v100","v101","v104","v105","v106"
    if (V == 2) asm volatile(M16("v", "v100", "v104") ::: VCLOB);
    if (V == 3) asm volatile(M16("v", "v100", "v105") ::: VCLOB);
    if (V == 4) asm volatile(M16("v", "v101", "v106") ::: VCLOB);
  }
  if (out && threadIdx.x == 9999) out[0] = 1.f;
}

template <int V>
void run(const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, 100, (float*)nullptr);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, iters, (float*)nullptr);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * 4 * iters * 16 * 2048.0;  // CUs x waves x iters x MFMAs x flops
  printf("%-44s %7.2f TF/s  (%.1f cycles per MFMA at 2.4 GHz)\n", name, flops / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 16.0));
}

int main() {
  run<0>("AGPR acc, A v100 / B v104 (same bank)");
  run<1>("AGPR acc, A v100 / B v105 (different banks)");
  run<2>("VGPR acc, A v100 / B v104 (same bank)");
  run<3>("VGPR acc, A v100 / B v105");
  run<4>("VGPR acc, A v101 / B v106");
  return 0;
}
