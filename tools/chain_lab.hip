// Lab: ONE wide layer of the 16-row chain kernels (mlp_fwd_kernel<1, NCB, 8>: y = relu(x W^T + b), 16 rows per
// workgroup, 8 waves, packed weights P[k/4][n][k%4] streamed from L2) in two decompositions, standalone, no torch:
//   hipcc -O3 --offload-arch=gfx950 tools/chain_lab.hip -o tools/_lab/chain_lab && tools/_lab/chain_lab
// A  (the product's): a wave owns a contiguous group of 16-column blocks and walks ALL k-steps with a DEPTH-deep register
//    ring of weight fragments -- a layer is (K / 16) / DEPTH dependent L2 round trips per wave;
// B  K-split: a wave owns the k-steps ks = wave (mod 8) for ALL column blocks, issues every weight fragment of a k-step
//    at once (16-25 KB in flight per wave), leaves its partial [16 x N] tile in LDS and the eight partials are summed
//    (+ bias, activation) by all threads -- (K / 16) / 8 round trips per wave + one LDS reduction.
// Reports us per layer launch for 128 row tiles x E nets (the C2 shapes: B = 2048; widths 256 and 400).
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

struct LArgs {
  const float* X;   // [rows][K]
  const float* P;   // [E][Kp/4][Np][4]
  const float* b;   // [E][Np]
  float* Y;         // [E][rows][N]
  int rows, K, N, Kp, Np;
};

__device__ __forceinline__ void stage_rows(const LArgs& a, float* lds, int lda, int row0) {
  // 16 rows x Kp columns, zero padded; 512 threads
  for (int idx = threadIdx.x; idx < 16 * a.Kp; idx += 512) {
    const int r = idx / a.Kp, c = idx - r * a.Kp;
    lds[r * lda + c] = (c < a.K && row0 + r < a.rows) ? a.X[(size_t)(row0 + r) * a.K + c] : 0.f;
  }
}

// ---- A: column groups, DEPTH-deep fragment ring -------------------------------------------------------------------
template <int NCB, int DEPTH>
__global__ __launch_bounds__(512, 2) void layer_cols(const LArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int e = blockIdx.y, row0 = blockIdx.x * 16;
  const int lda = a.Kp + 8, nk = a.Kp >> 4, nblk = a.Np >> 4;
  stage_rows(a, lds, lda, row0);
  __syncthreads();
  // balanced contiguous groups: the first (nblk % 8) waves take one block more
  const int base = nblk / 8, extra = nblk % 8;
  const int cnt = base + (wave < extra ? 1 : 0), cb0 = wave * base + (wave < extra ? wave : extra);
  const float* __restrict__ P = a.P + (size_t)e * a.Kp * a.Np;
  f32x4 acc[NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const float bv = c < cnt ? a.b[(size_t)e * a.Np + (cb0 + c) * 16 + m] : 0.f;
    acc[c] = f32x4{bv, bv, bv, bv};  // (bias per column: the accumulator layout has column = lane & 15)
  }
  f32x4 bf[DEPTH][NCB];
  auto loadb = [&](int ks, f32x4* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int cb = cb0 + (c < cnt ? c : 0);
      dst[c] = *reinterpret_cast<const f32x4*>(P + ((size_t)(ks * 4 + kq) * a.Np + cb * 16 + m) * 4);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) loadb(d < nk ? d : nk - 1, bf[d]);
  const float* arow = lds + m * lda + 4 * kq;
  for (int ks0 = 0; ks0 < nk; ks0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int ks = ks0 + d;
      if (ks < nk) {
        const int kn = ks + DEPTH - 1;
        loadb(kn < nk ? kn : nk - 1, bf[(d + DEPTH - 1) % DEPTH]);
        const f32x4 af = *reinterpret_cast<const f32x4*>(arow + ks * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < NCB; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf[d][c][t], acc[c], 0, 0, 0);
      }
    }
  }
  float* __restrict__ y = a.Y + (size_t)e * a.rows * a.N;
#pragma unroll
  for (int c = 0; c < NCB; ++c)
    if (c < cnt) {
      const int col = (cb0 + c) * 16 + m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = row0 + kq * 4 + r;
        if (gr < a.rows && col < a.N) y[(size_t)gr * a.N + col] = fmaxf(acc[c][r], 0.f);
      }
    }
}

// ---- A32: the same with TWO 16-row blocks per workgroup (32 rows): every weight fragment feeds two MFMAs ------------------
template <int NCB, int DEPTH>
__global__ __launch_bounds__(512, 2) void layer_cols32(const LArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int e = blockIdx.y, row0 = blockIdx.x * 32;
  const int lda = a.Kp + 8, nk = a.Kp >> 4, nblk = a.Np >> 4;
  for (int idx = threadIdx.x; idx < 32 * a.Kp; idx += 512) {
    const int r = idx / a.Kp, c = idx - r * a.Kp;
    lds[r * lda + c] = (c < a.K && row0 + r < a.rows) ? a.X[(size_t)(row0 + r) * a.K + c] : 0.f;
  }
  __syncthreads();
  const int base = nblk / 8, extra = nblk % 8;
  const int cnt = base + (wave < extra ? 1 : 0), cb0 = wave * base + (wave < extra ? wave : extra);
  const float* __restrict__ P = a.P + (size_t)e * a.Kp * a.Np;
  f32x4 acc[2][NCB];
#pragma unroll
  for (int c = 0; c < NCB; ++c) {
    const float bv = c < cnt ? a.b[(size_t)e * a.Np + (cb0 + c) * 16 + m] : 0.f;
    acc[0][c] = acc[1][c] = f32x4{bv, bv, bv, bv};
  }
  f32x4 bf[DEPTH][NCB];
  auto loadb = [&](int ks, f32x4* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
      const int cb = cb0 + (c < cnt ? c : 0);
      dst[c] = *reinterpret_cast<const f32x4*>(P + ((size_t)(ks * 4 + kq) * a.Np + cb * 16 + m) * 4);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) loadb(d < nk ? d : nk - 1, bf[d]);
  const float* arow = lds + m * lda + 4 * kq;
  for (int ks0 = 0; ks0 < nk; ks0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int ks = ks0 + d;
      if (ks < nk) {
        const int kn = ks + DEPTH - 1;
        loadb(kn < nk ? kn : nk - 1, bf[(d + DEPTH - 1) % DEPTH]);
        const f32x4 af0 = *reinterpret_cast<const f32x4*>(arow + ks * 16);
        const f32x4 af1 = *reinterpret_cast<const f32x4*>(arow + 16 * lda + ks * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int c = 0; c < NCB; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af0[t], bf[d][c][t], acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af1[t], bf[d][c][t], acc[1][c], 0, 0, 0);
          }
      }
    }
  }
  float* __restrict__ y = a.Y + (size_t)e * a.rows * a.N;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int c = 0; c < NCB; ++c)
      if (c < cnt) {
        const int col = (cb0 + c) * 16 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = row0 + rb * 16 + kq * 4 + r;
          if (gr < a.rows && col < a.N) y[(size_t)gr * a.N + col] = fmaxf(acc[rb][c][r], 0.f);
        }
      }
}

// ---- B: K-split over the 8 waves, partial tiles through LDS ---------------------------------------------------------
template <int NBLK>  // column blocks (Np / 16): 16 or 25
__global__ __launch_bounds__(512, 2) void layer_ksplit(const LArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int e = blockIdx.y, row0 = blockIdx.x * 16;
  const int lda = a.Kp + 8, nk = a.Kp >> 4;
  constexpr int Np = NBLK * 16;
  float* part = lds + 16 * lda;  // [4][16 rows][Np]
  stage_rows(a, lds, lda, row0);
  __syncthreads();
  const float* __restrict__ P = a.P + (size_t)e * a.Kp * Np;
  f32x4 acc[NBLK];
#pragma unroll
  for (int c = 0; c < NBLK; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* arow = lds + m * lda + 4 * kq;
  for (int ks = wave; ks < nk; ks += 8) {
    f32x4 bf[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
      bf[c] = *reinterpret_cast<const f32x4*>(P + ((size_t)(ks * 4 + kq) * Np + c * 16 + m) * 4);
    const f32x4 af = *reinterpret_cast<const f32x4*>(arow + ks * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < NBLK; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf[c][t], acc[c], 0, 0, 0);
  }
  // partial tiles: waves 4-7 park theirs in LDS, waves 0-3 add the one of wave + 4 to their registers and park the sums
  // (4 x 16 x Np floats: 64 KB at 256 columns, 100 KB at 400), then all threads add the four and finish the layer
  float* pw = part + (size_t)(wave & 3) * 16 * Np;
  if (wave >= 4) {
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(kq * 4 + r) * Np + c * 16 + m] = acc[c][r];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[c][r] += pw[(kq * 4 + r) * Np + c * 16 + m];
  }
  __syncthreads();
  if (wave < 4) {
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(kq * 4 + r) * Np + c * 16 + m] = acc[c][r];
  }
  __syncthreads();
  float* __restrict__ y = a.Y + (size_t)e * a.rows * a.N;
  for (int idx = threadIdx.x; idx < 16 * (Np / 4); idx += 512) {
    const int r = idx / (Np / 4), c4 = idx - r * (Np / 4);
    f32x4 s = *reinterpret_cast<const f32x4*>(part + r * Np + 4 * c4);
#pragma unroll
    for (int w = 1; w < 4; ++w) s += *reinterpret_cast<const f32x4*>(part + (size_t)w * 16 * Np + r * Np + 4 * c4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b + (size_t)e * Np + 4 * c4);
    const int gr = row0 + r;
    if (gr < a.rows) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * c4 + j < a.N) y[(size_t)gr * a.N + 4 * c4 + j] = fmaxf(s[j] + bv[j], 0.f);
    }
  }
}

template <class K>
float time_k(K k, dim3 grid, size_t ldsb, const LArgs& a, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, grid, dim3(512), ldsb, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(512), ldsb, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main() {
  const int rows = 2048, Emax = 6;
  for (int width : {256, 400}) {
    const int K = width, N = width, Kp = (K + 15) & ~15, Np = (N + 15) & ~15;
    std::vector<float> hX((size_t)rows * K), hP((size_t)Emax * Kp * Np, 0.f), hb((size_t)Emax * Np, 0.f), hW((size_t)Emax * N * K);
    uint32_t s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hX) v = rnd();
    for (auto& v : hW) v = rnd() * 0.1f;
    for (int e = 0; e < Emax; ++e) {
      for (int n = 0; n < N; ++n) hb[(size_t)e * Np + n] = rnd();
      for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) hP[(size_t)e * Kp * Np + ((size_t)(k / 4) * Np + n) * 4 + (k & 3)] = hW[((size_t)e * N + n) * K + k];
    }
    float *dX, *dP, *db, *dY0, *dY1;
    CK(hipMalloc(&dX, hX.size() * 4));
    CK(hipMalloc(&dP, hP.size() * 4));
    CK(hipMalloc(&db, hb.size() * 4));
    CK(hipMalloc(&dY0, (size_t)Emax * rows * N * 4));
    CK(hipMalloc(&dY1, (size_t)Emax * rows * N * 4));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    LArgs a{dX, dP, db, dY0, rows, K, N, Kp, Np};
    const size_t ldsA = sizeof(float) * 16 * (Kp + 8), ldsB = ldsA + sizeof(float) * 4 * 16 * Np;
    printf("width %d: one layer, 2048 rows (128 row tiles), %.2f MFLOP per net; weights %.0f KB per net\n", width,
           2.0 * rows * K * N * 1e-6, Kp * Np * 4 / 1024.0);
    for (int E : {1, 2, 4, 6}) {
      const dim3 grid(rows / 16, E, 1);
      const double gf = 2.0 * rows * K * N * E * 1e-9;
      float tA2, tA3, tA4, tB, tC;
      const dim3 grid32(rows / 32, E, 1);
      const size_t ldsC = sizeof(float) * 32 * (Kp + 8);
      if (width == 256) {
        a.Y = dY0; tA3 = time_k(layer_cols<2, 3>, grid, ldsA, a, 50);
        tA2 = time_k(layer_cols<2, 2>, grid, ldsA, a, 50);
        tA4 = time_k(layer_cols<2, 6>, grid, ldsA, a, 50);
        a.Y = dY1; tC = time_k(layer_cols32<2, 3>, grid32, ldsC, a, 50);
        a.Y = dY1; tB = time_k(layer_ksplit<16>, grid, ldsB, a, 50);
      } else {
        a.Y = dY0; tA3 = time_k(layer_cols<4, 3>, grid, ldsA, a, 50);
        tA2 = time_k(layer_cols<4, 2>, grid, ldsA, a, 50);
        tA4 = time_k(layer_cols<4, 6>, grid, ldsA, a, 50);
        a.Y = dY1; tC = time_k(layer_cols32<4, 3>, grid32, ldsC, a, 50);
        a.Y = dY1; tB = time_k(layer_ksplit<25>, grid, ldsB, a, 50);
      }
      std::vector<float> y0((size_t)E * rows * N), y1((size_t)E * rows * N);
      CK(hipMemcpy(y0.data(), dY0, y0.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y1.data(), dY1, y1.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, mc = 0;
      for (size_t i = 0; i < y0.size(); ++i) md = fmax(md, fabs((double)y0[i] - y1[i]));
      for (int t = 0; t < 32; ++t) {  // CPU spot check of A
        const int e = t % E, r = (t * 977) % rows, n = (t * 131) % N;
        double acc = hb[(size_t)e * Np + n];
        for (int k = 0; k < K; ++k) acc += (double)hX[(size_t)r * K + k] * hW[((size_t)e * N + n) * K + k];
        mc = fmax(mc, fabs(fmax(acc, 0.0) - y0[((size_t)e * rows + r) * N + n]));
      }
      printf("  %d nets: columns ring 2 / 3 / 6: %6.2f / %6.2f / %6.2f us (%.2f of the fp32 roof)   32-row tiles: %6.2f us   K-split: %6.2f us   |A-B| %.1e  |A-cpu| %.1e\n",
             E, tA2, tA3, tA4, gf * 1e3 / tA3 / 157.3, tC, tB, md, mc);
    }
    CK(hipFree(dX)); CK(hipFree(dP)); CK(hipFree(db)); CK(hipFree(dY0)); CK(hipFree(dY1));
  }
  return 0;
}
