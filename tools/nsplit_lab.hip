// Lab (round 5): ONE wide layer of the 2048-row chain, y = relu(x W^T + b), as an ALL-CU GEMM instead
// of inside a 16-row tile walk.  Standalone, no torch:
//   hipcc -O3 --offload-arch=gfx950 tools/nsplit_lab.hip -o tools/_lab/nsplit_lab && tools/_lab/nsplit_lab
// Form: one workgroup = one [16 RB rows x 16 CB columns] output tile and KS waves; wave w owns the k-steps of its K range
// for the WHOLE tile (RB x CB accumulator blocks), requests every operand fragment of its range up front (A rows straight
// from the row-major activation matrix as 16-byte fragments, W from the packed copy P[k/4][n][k%4]) -- ONE memory round
// trip per wave, no LDS staging, no barrier before the end -- and the KS partial tiles meet in LDS, where all threads add
// them, apply bias + activation and write coalesced rows.  blockIdx -> tile mapping keeps the column groups of one row
// tile on one XCD (they share the A rows through that XCD's L2).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

struct GArgs {
  const float* X;  // [rows][K] row-major (K % 4 == 0)
  const float* P;  // [Kp/4][Np][4]
  const float* b;  // [Np]
  float* Y;        // [rows][N]
  int rows, K, N, Kp, Np, row_tiles, col_groups;
};

// RB row blocks x CB column blocks per workgroup, KS waves (K split), NKW = k-steps per wave (upper bound)
template <int RB, int CB, int KS, int NKW, bool XCD>
__global__ __launch_bounds__(64 * KS, 1) void gemm_ns(const GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kq = lane >> 4;
  int tile, cg;
  if (XCD) {  // ids with the same (id % 8) run on one XCD: give each XCD whole row tiles
    const int id = blockIdx.x, x = id & 7, s = id >> 3;
    tile = x + 8 * (s / a.col_groups);
    cg = s % a.col_groups;
    if (tile >= a.row_tiles) return;
  } else {
    tile = blockIdx.x / a.col_groups;
    cg = blockIdx.x % a.col_groups;
  }
  const int row0 = tile * 16 * RB, col0 = cg * 16 * CB;
  const int nk = a.Kp >> 4;
  // contiguous K ranges: the first (nk % KS) waves take one k-step more
  const int base = nk / KS, extra = nk % KS;
  const int cnt = base + (wave < extra ? 1 : 0), ks0 = wave * base + (wave < extra ? wave : extra);
  f32x4 af[NKW][RB], bf[NKW][CB];
#pragma unroll
  for (int j = 0; j < NKW; ++j) {
    const int ks = ks0 + (j < cnt ? j : 0);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      int r = row0 + rb * 16 + m;
      r = r < a.rows ? r : a.rows - 1;
      af[j][rb] = *reinterpret_cast<const f32x4*>(a.X + (size_t)r * a.K + ks * 16 + kq * 4);
    }
#pragma unroll
    for (int c = 0; c < CB; ++c)
      bf[j][c] = *reinterpret_cast<const f32x4*>(a.P + ((size_t)(ks * 4 + kq) * a.Np + col0 + c * 16 + m) * 4);
    __builtin_amdgcn_sched_barrier(0);  // requests leave in k-step order: the MFMAs of step j wait for steps <= j only
  }
  f32x4 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NKW; ++j) {
    __builtin_amdgcn_sched_barrier(0);
    if (j < cnt) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int c = 0; c < CB; ++c)
            acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][rb][t], bf[j][c][t], acc[rb][c], 0, 0, 0);
    }
  }
  // partial tiles -> LDS [KS][16 RB][16 CB + 4]
  constexpr int LD = 16 * CB + 4, BM = 16 * RB;
  float* pw = lds + (size_t)wave * BM * LD;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(rb * 16 + kq * 4 + r) * LD + c * 16 + m] = acc[rb][c][r];
  __syncthreads();
  constexpr int C4 = 4 * CB;  // float4 per tile row
  for (int idx = threadIdx.x; idx < BM * C4; idx += 64 * KS) {
    const int r = idx / C4, c4 = idx - r * C4;
    f32x4 s = *reinterpret_cast<const f32x4*>(lds + r * LD + 4 * c4);
#pragma unroll
    for (int w = 1; w < KS; ++w) s += *reinterpret_cast<const f32x4*>(lds + (size_t)w * BM * LD + r * LD + 4 * c4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b + col0 + 4 * c4);
    const int gr = row0 + r;
    if (gr < a.rows) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(s[j] + bv[j], 0.f);
      *reinterpret_cast<f32x4*>(a.Y + (size_t)gr * a.N + col0 + 4 * c4) = o;
    }
  }
}

__global__ void null_kernel(const GArgs a) {
  if (a.rows < 0) a.Y[0] = 0.f;
}

template <class K>
float time_k(K k, dim3 grid, int threads, size_t ldsb, const GArgs& a, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, grid, dim3(threads), ldsb, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(threads), ldsb, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1000.f / reps;
}

template <int RB, int CB, int KS, int NKW, bool XCD>
void run(const char* name, GArgs a, const std::vector<float>& hX, const std::vector<float>& hW, const std::vector<float>& hb) {
  a.row_tiles = (a.rows + 16 * RB - 1) / (16 * RB);
  a.col_groups = a.Np / (16 * CB);
  if (a.col_groups * 16 * CB != a.Np) { printf("  %-28s skipped (N %% %d)\n", name, 16 * CB); return; }
  if ((a.Kp / 16 + KS - 1) / KS > NKW) { printf("  %-28s skipped (k-steps per wave > %d)\n", name, NKW); return; }
  const int n_wg = XCD ? ((a.row_tiles + 7) / 8) * 8 * a.col_groups : a.row_tiles * a.col_groups;
  const size_t ldsb = sizeof(float) * KS * 16 * RB * (16 * CB + 4);
  CK(hipMemset(a.Y, 0, (size_t)a.rows * a.N * 4));
  const float t = time_k(gemm_ns<RB, CB, KS, NKW, XCD>, dim3(n_wg), 64 * KS, ldsb, a, 100);
  std::vector<float> y((size_t)a.rows * a.N);
  CK(hipMemcpy(y.data(), a.Y, y.size() * 4, hipMemcpyDeviceToHost));
  double mc = 0;
  for (int t2 = 0; t2 < 256; ++t2) {
    const int r = (t2 * 977 + 13) % a.rows, n = (t2 * 131 + 7) % a.N;
    double acc = hb[n];
    for (int k = 0; k < a.K; ++k) acc += (double)hX[(size_t)r * a.K + k] * hW[(size_t)n * a.K + k];
    mc = fmax(mc, fabs(fmax(acc, 0.0) - y[(size_t)r * a.N + n]));
  }
  const double gf = 2.0 * a.rows * a.K * a.N * 1e-9;
  printf("  %-28s %4d wg x %3d thr  lds %5.1f KB  %7.2f us  %.2f of the fp32 roof  |y-cpu| %.1e\n", name, n_wg, 64 * KS,
         ldsb / 1024.0, t, gf * 1e3 / t / 157.3, mc);
}

int main() {
  const int rows = 2048;
  for (int shape = 0; shape < 3; ++shape) {
    const int K = shape == 0 ? 400 : shape == 1 ? 256 : 80, N = shape == 1 ? 256 : 400;
    const int Kp = (K + 15) & ~15, Np = (N + 15) & ~15;
    std::vector<float> hX((size_t)rows * K), hP((size_t)Kp * Np, 0.f), hb((size_t)Np, 0.f), hW((size_t)N * K);
    uint32_t s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hX) v = rnd();
    for (auto& v : hW) v = rnd() * 0.1f;
    for (int n = 0; n < N; ++n) hb[n] = rnd();
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) hP[((size_t)(k / 4) * Np + n) * 4 + (k & 3)] = hW[(size_t)n * K + k];
    float *dX, *dP, *db, *dY;
    CK(hipMalloc(&dX, hX.size() * 4));
    CK(hipMalloc(&dP, hP.size() * 4));
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dY, (size_t)rows * N * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    GArgs a{dX, dP, db, dY, rows, K, N, Kp, Np, 0, 0};
    printf("y[%d, %d] = relu(x[%d, %d] W^T + b): %.1f MFLOP, MFMA floor on 256 CUs %.2f us\n", rows, N, rows, K,
           2.0 * rows * K * N * 1e-6, 2.0 * rows * K * N / 157.3e6);
    if (shape == 0) printf("  empty kernel, 240 wg x 256 thr, same back-to-back launch loop: %.2f us\n",
                           time_k(null_kernel, dim3(240), 256, 0, a, 100));
    if (shape == 0) {
      run<3, 5, 4, 7, true>("48x80 KS4 xcd", a, hX, hW, hb);
      run<3, 5, 4, 7, false>("48x80 KS4", a, hX, hW, hb);
      run<3, 5, 8, 4, true>("48x80 KS8 xcd", a, hX, hW, hb);
      run<2, 5, 4, 7, true>("32x80 KS4 xcd", a, hX, hW, hb);
      run<2, 5, 8, 4, true>("32x80 KS8 xcd", a, hX, hW, hb);
      run<4, 5, 4, 7, true>("64x80 KS4 xcd", a, hX, hW, hb);
    } else if (shape == 1) {
      run<3, 4, 4, 4, true>("48x64 KS4 xcd", a, hX, hW, hb);
      run<2, 4, 4, 4, true>("32x64 KS4 xcd", a, hX, hW, hb);
      run<4, 4, 4, 4, true>("64x64 KS4 xcd", a, hX, hW, hb);
      run<2, 4, 8, 2, true>("32x64 KS8 xcd", a, hX, hW, hb);
      run<2, 2, 4, 4, true>("32x32 KS4 xcd", a, hX, hW, hb);
    } else {
      run<2, 5, 1, 5, true>("32x80 KS1 xcd", a, hX, hW, hb);
      run<2, 5, 2, 3, true>("32x80 KS2 xcd", a, hX, hW, hb);
      run<1, 5, 1, 5, true>("16x80 KS1 xcd", a, hX, hW, hb);
      run<3, 5, 1, 5, true>("48x80 KS1 xcd", a, hX, hW, hb);
    }
    CK(hipFree(dX)); CK(hipFree(dP)); CK(hipFree(db)); CK(hipFree(dY));
  }
  return 0;
}
