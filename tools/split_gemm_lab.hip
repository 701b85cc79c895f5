// Lab: an fp32-accurate GEMM on the bf16 matrix cores of gfx950 (Y = A W^T + b, A [M,K] fp32 activations, W [N,K] fp32
// weights -- the CDT projections of csrc/mlp.hip linear_pers_kernel, nn.Linear of cdt.py's TransformerBlock).
//
// Why: every f32-input MFMA of this package issues at the packed-fp32 VECTOR rate (practical roof 135 TF/s,
// tools/mfma_bf16_probe.hip); v_mfma_f32_32x32x16_bf16 sustains 2.3-2.4 PF/s on the same chip.  An fp32 number splits
// EXACTLY into three bf16 pieces by truncation (a = a1 + a2 + a3: 8 + 8 + 8 significant bits), so
//     a * b = sum_{i,j} a_i b_j,   every a_i b_j exact in fp32 (16-bit product),
// and dropping the three terms with i + j >= 5 (<= 2^-23 |a||b|, below fp32's own half ulp of the product) leaves SIX
// bf16 MFMAs per fp32 one, accumulated in fp32 by the matrix unit: 18x / 6 = ~2.7x the f32 MFMA rate at fp32 accuracy
// (numpy emulation, K = 256 / 1024: rms error 1.4e-7 / 2.6e-7 against 6e-8 / 1.1e-7 for a chunked fp32 sum and 2.9e-7 /
// 3.4e-7 for the CPU BLAS sgemm the reference itself runs on).
//
// Kernel: 128 x 128 output tile per 4-wave workgroup (two per CU), wave tile 64 x 64 = 2 x 2 MFMA blocks of 32 x 32;
// per 32-deep slab every thread splits its 16 A values in registers (4 vector instructions per value + packing) and
// writes the three bf16 planes to LDS; W arrives pre-split (split_planes_kernel, once per optimizer step).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/split_gemm_lab.hip -o /tmp/split_gemm_lab && /tmp/split_gemm_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));      \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPitch = 80;             // bytes per row of a plane in LDS: 32 bf16 + 16 B pad (conflict-free b128 reads)
constexpr int kPlane = 128 * kPitch;   // one plane of a 128-row slab
constexpr int kLds = 6 * kPlane;       // A planes 0-2, B planes 3-5: 61440 B

// a = h + m + l exactly (h, m, l: fp32 bit patterns whose low 16 bits are zero = bf16 values)
__device__ __forceinline__ void split3(float a, unsigned& h, unsigned& m, unsigned& l) {
  h = __float_as_uint(a) & 0xffff0000u;
  const float r1 = a - __uint_as_float(h);
  m = __float_as_uint(r1) & 0xffff0000u;
  l = __float_as_uint(r1 - __uint_as_float(m));
}
// two bf16 (the high halves of e0, e1) in one dword, e0 in the low half
__device__ __forceinline__ unsigned pack_hi(unsigned e0, unsigned e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// W [n] fp32 -> planes[p][n] bf16 (same element order)
__global__ void split_planes_kernel(const float* __restrict__ w, uint16_t* __restrict__ planes, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h, m, l;
    split3(w[i], h, m, l);
    planes[i] = (uint16_t)(h >> 16);
    planes[n + i] = (uint16_t)(m >> 16);
    planes[2 * n + i] = (uint16_t)(l >> 16);
  }
}

struct SplitArgs {
  const float* A;
  const uint16_t* Wp;  // three planes of W [N, K], plane stride N * K
  const float* bias;
  const float* resid;
  float* Y;
  int64_t lda, ldr, ldy;
  int32_t M, K, N;
};

template <int NPROD, int ORDER = 0, bool SW = false>
__global__ __launch_bounds__(256, SW ? 3 : 2) void split_gemm_kernel(const SplitArgs a) {
  constexpr int kPitch = SW ? 64 : 80, kPlane = 128 * kPitch;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int row0, col0;
  if (ORDER == 0) {
    row0 = blockIdx.x * 128; col0 = blockIdx.y * 128;
  } else {
    // 1-D grid; workgroup i runs on XCD i % 8: give every XCD a contiguous run of tiles, column tiles fastest, so the
    // workgroups that share A rows (and the W planes) meet in ONE L2
    const int n = gridDim.x, ct = a.N >> 7;
    const int i = blockIdx.x, per = (n + 7) >> 3;
    int t = (i & 7) * per + (i >> 3);
    if (t >= n) return;
    row0 = (t / ct) * 128; col0 = (t % ct) * 128;
  }
  const int K = a.K, nk = K >> 5;
  const size_t pstride = (size_t)a.N * K;
  // ---- sources
  const int arow = tid >> 1, ahalf = tid & 1;
  const float* ap = a.A + (size_t)(row0 + arow) * a.lda + ahalf * 16;
  const uint16_t* bp[6];
  int boff[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int c = tid + 256 * j, p = c >> 9, rem = c & 511, col = rem >> 2, q = rem & 3;
    bp[j] = a.Wp + p * pstride + (size_t)(col0 + col) * K + q * 8;
    boff[j] = (3 + p) * kPlane + col * kPitch + (SW ? (q ^ ((col >> 2) & 3)) : q) * 16;
  }
  const int aoff = arow * kPitch;
  const int asw = SW ? ((arow >> 2) & 3) : 0;
  f32x4 pa[4];
  u32x4 pb[6];
  auto fetch = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = *reinterpret_cast<const f32x4*>(ap + ks * 32 + 4 * i);
#pragma unroll
    for (int j = 0; j < 6; ++j) pb[j] = *reinterpret_cast<const u32x4*>(bp[j] + ks * 32);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[r][c][v] = 0.f;
  // fragment addresses (bytes): plane p, K16 step s: + p * kPlane + s * 32
  const int fra = wr * 64 + (lane & 31), frb = wc * 64 + (lane & 31);  // (+ 32 per block: (row >> 2) & 3 unchanged)
  const int fa = fra * kPitch, fb = 3 * kPlane + frb * kPitch;
  const int swa = SW ? ((fra >> 2) & 3) : 0, swb = SW ? ((frb >> 2) & 3) : 0;
  fetch(0);
  for (int ks = 0; ks < nk; ++ks) {
    // ---- split this thread's 16 A values, write the planes; pass the pre-split W chunks through
#pragma unroll
    for (int h8 = 0; h8 < 2; ++h8) {
      unsigned hh[8], mm[8], ll[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split3(pa[2 * h8 + (e >> 2)][e & 3], hh[e], mm[e], ll[e]);
      u32x4 vh, vm, vl;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        vh[d] = pack_hi(hh[2 * d], hh[2 * d + 1]);
        vm[d] = pack_hi(mm[2 * d], mm[2 * d + 1]);
        vl[d] = pack_hi(ll[2 * d], ll[2 * d + 1]);
      }
      const int ao = aoff + (((ahalf * 2 + h8) ^ asw) * 16);
      *reinterpret_cast<u32x4*>(lds + 0 * kPlane + ao) = vh;
      *reinterpret_cast<u32x4*>(lds + 1 * kPlane + ao) = vm;
      *reinterpret_cast<u32x4*>(lds + 2 * kPlane + ao) = vl;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<u32x4*>(lds + boff[j]) = pb[j];
    __syncthreads();
    if (ks + 1 < nk) fetch(ks + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + fa + p * kPlane + r * 32 * kPitch + (((2 * s + (lane >> 5)) ^ swa) * 16)));
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + fb + p * kPlane + c * 32 * kPitch + (((2 * s + (lane >> 5)) ^ swb) * 16)));
      // smallest terms first; the four blocks between two products on the same accumulator
      constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}, PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};  // the last NPROD are used
#pragma unroll
      for (int t = 9 - NPROD; t < 9; ++t) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][PA[t]], bf[c][PB[t]], acc[r][c], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: 32 x 32 block layout: register v of lane l = row (v / 4) * 8 + (l / 32) * 4 + v % 4, column l % 32.
  // One lane-dependent base offset; everything else is wave-uniform (scalar) arithmetic.
  {
    const unsigned ldy = (unsigned)a.ldy, ldr = (unsigned)a.ldr;
    const unsigned lrow = (unsigned)(wr * 64 + (lane >> 5) * 4), lcol = (unsigned)(wc * 64 + (lane & 31));
    float* yb = a.Y + (size_t)row0 * a.ldy + col0;
    const float* rb_ = a.resid ? a.resid + (size_t)row0 * a.ldr + col0 : nullptr;
    const unsigned oy = lrow * ldy + lcol, orr = lrow * ldr + lcol;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float bv = a.bias ? a.bias[col0 + lcol + c * 32] : 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const unsigned ro = (unsigned)(r * 32 + (v >> 2) * 8 + (v & 3));
          float y = acc[r][c][v] + bv;
          if (rb_) y += rb_[orr + ro * ldr + c * 32];
          yb[oy + ro * ldy + c * 32] = y;
        }
    }
  }
}



// ---- v2: persistent, one 8-wave workgroup per CU; A fp32 slabs and pre-split W planes arrive by DMA (global_load_lds),
// every thread splits 8 values of slab g + 1 from the staging ring into the other A-plane buffer while the MFMAs of slab g
// run; one barrier per 32-deep slab; the plane rows are 64 B with the 16-byte chunk position XOR-swizzled by (row >> 2) & 3
// (conflict-free ds_read_b128 without padding; the DMA realises it by choosing which global chunk a lane fetches).
__device__ __forceinline__ void glds16(const void* g, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
constexpr int kAst = 128 * 128;       // one fp32 A slab: 128 rows x 32 k
constexpr int kPl = 128 * 64;         // one bf16 plane of 128 rows x 32 k
constexpr int kV2Lds = 2 * kAst + 2 * 3 * kPl + 3 * 3 * kPl;  // 155648 B

template <int NSETS, bool DO_SPLIT, bool DO_MMA, bool DO_DMA, bool NO_BAR = false, bool NO_LDS = false>
__global__ __launch_bounds__(512, 2) void split_gemm_v2_kernel(const SplitArgs a, const int row_tiles, const int col_tiles,
                                                               const int nwg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ast = lds;
  unsigned char* const Apl = lds + 2 * kAst;
  unsigned char* const Bpl = Apl + 2 * 3 * kPl;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int K = a.K, nk = K >> 5;
  const size_t pstride = (size_t)a.N * K;
  const int tiles = row_tiles * col_tiles;
  const int t0 = (int)((long)tiles * blockIdx.x / nwg), t1 = (int)((long)tiles * (blockIdx.x + 1) / nwg);
  if (t0 >= t1) return;
  const int nsteps = (t1 - t0) * nk;
  // ---- DMA issue state: the slab about to be issued
  int i_tile = t0, i_ks = 0, i_step = 0;
  const int a_lrow = lane >> 3, a_chunk = lane & 7;                      // A staging: 8 rows x 8 chunks per wave-instruction
  const int b_lrow = lane >> 2, b_chunk = (lane & 3) ^ ((lane >> 4) & 3);  // W planes: 16 rows x 4 chunk slots
  auto dma = [&]() {
    const int rt = i_tile / col_tiles, ct = i_tile - rt * col_tiles;
    const int sa = i_step & 1, sb = i_step % 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = rt * 128 + 8 * (2 * wave + j) + a_lrow;
      glds16(a.A + (size_t)row * a.lda + i_ks * 32 + a_chunk * 4, Ast + sa * kAst + (2 * wave + j) * 1024);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int col = ct * 128 + 16 * wave + b_lrow;
      glds16(a.Wp + p * pstride + (size_t)col * K + i_ks * 32 + b_chunk * 8, Bpl + (sb * 3 + p) * kPl + wave * 1024);
    }
    ++i_step;
    if (++i_ks == nk) { i_ks = 0; ++i_tile; if (i_tile >= t1) i_tile = t1 - 1; }  // (past the end: harmless reloads)
  };
  // ---- split: this thread's 8 values of a staged slab -> the three planes
  const int s_row = tid >> 2, s_q = tid & 3;
  const int s_src = s_row * 128 + s_q * 32;
  const int s_dst = s_row * 64 + ((s_q ^ ((s_row >> 2) & 3)) * 16);
  auto split = [&](int step) {
    const unsigned char* src = Ast + (step & 1) * kAst + s_src;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 16);
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split3(x0[e], hh[e], mm[e], ll[e]); split3(x1[e], hh[4 + e], mm[4 + e], ll[4 + e]); }
    u32x4 vh, vm, vl;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vh[d] = pack_hi(hh[2 * d], hh[2 * d + 1]);
      vm[d] = pack_hi(mm[2 * d], mm[2 * d + 1]);
      vl[d] = pack_hi(ll[2 * d], ll[2 * d + 1]);
    }
    unsigned char* dst = Apl + (step & 1) * 3 * kPl + s_dst;
    *reinterpret_cast<u32x4*>(dst) = vh;
    *reinterpret_cast<u32x4*>(dst + kPl) = vm;
    *reinterpret_cast<u32x4*>(dst + 2 * kPl) = vl;
  };
  // ---- fragments
  const int f_row_a = wr * 64 + (lane & 31), f_row_b = wc * 32 + (lane & 31);
  const int f_hi = lane >> 5;
  f32x16 acc[NSETS][2];
#pragma unroll
  for (int q = 0; q < NSETS; ++q)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[q][r][v] = 0.f;
  auto mma = [&](int step) {
    const unsigned char* ap = Apl + (step & 1) * 3 * kPl;
    const unsigned char* bp = Bpl + (step % 3) * 3 * kPl;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2][3], bf[3];
      if (NO_LDS) {
        u32x4 z = {0x3f803f80u + (unsigned)lane, 0x3f803f80u, 0x3f803f80u + (unsigned)step, 0x3f803f80u};
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[0][p] = __builtin_bit_cast(bf16x8, z); af[1][p] = __builtin_bit_cast(bf16x8, z); bf[p] = __builtin_bit_cast(bf16x8, z); }
      } else {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = f_row_a + r * 32;
        const int off = row * 64 + (((2 * s + f_hi) ^ ((row >> 2) & 3)) * 16);
#pragma unroll
        for (int p = 0; p < 3; ++p) af[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ap + p * kPl + off));
      }
      {
        const int off = f_row_b * 64 + (((2 * s + f_hi) ^ ((f_row_b >> 2) & 3)) * 16);
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + p * kPl + off));
      }
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 2; ++r)
          acc[t * NSETS / 6][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][PA[t]], bf[PB[t]], acc[t * NSETS / 6][r], 0, 0, 0);
    }
  };
  // ---- prologue: slabs 0 and 1 in flight; slab 0 landed for everyone; split(0)
  dma();
  dma();
  asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
  split(0);
  int tile = t0, ks = 0;
  bool stored = false;
  for (int g = 0; g < nsteps; ++g) {
    // slab g + 1 landed (this wave's part; the stores of a tile that just ended may stay in flight), planes of slab g written
    if (NO_BAR) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (stored) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stored = false;
    if (DO_DMA) dma();  // slab g + 2
    if (wr == 0) {
      if (DO_SPLIT && g + 1 < nsteps) split(g + 1);
      if (DO_MMA) mma(g);
    } else {
      if (DO_MMA) mma(g);
      if (DO_SPLIT && g + 1 < nsteps) split(g + 1);
    }
    if (++ks == nk) {  // the tile is complete
      ks = 0;
      const int rt = tile / col_tiles, ct = tile - rt * col_tiles;
      const int col = ct * 128 + wc * 32 + (lane & 31);
      const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = rt * 128 + wr * 64 + r * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
          float y = acc[0][r][v];
#pragma unroll
          for (int q = 1; q < NSETS; ++q) y += acc[q][r][v];
          y += bv;
          if (a.resid) y += a.resid[(size_t)row * a.ldr + col];
          a.Y[(size_t)row * a.ldy + col] = y;
#pragma unroll
          for (int q = 0; q < NSETS; ++q) acc[q][r][v] = 0.f;
        }
      stored = true;
      ++tile;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int NSETS, bool DO_SPLIT, bool DO_MMA, bool DO_DMA, bool NO_BAR = false, bool NO_LDS = false>
static float run_v2(const SplitArgs& a, int reps) {
  auto* kern = split_gemm_v2_kernel<NSETS, DO_SPLIT, DO_MMA, DO_DMA, NO_BAR, NO_LDS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kV2Lds));
  int n_cu = 256;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  const int rt = a.M / 128, ct = a.N / 128;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), kV2Lds, 0, a, rt, ct, n_cu);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), kV2Lds, 0, a, rt, ct, n_cu);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}


// ---- v3: 256-column tiles over 16-deep slabs (half the L2 -> LDS bytes per FLOP of v2), persistent 8-wave workgroup per CU.
// <RB, CB, WR> = 32 x 32 blocks per wave (rows, columns) and wave rows: <4, 2, 2> = 256 x 256 tile, 128 x 64 per wave;
// <5, 1, 1> = 160 x 256 tile (N = 256 at M = 81920: 512 tiles = two whole rounds of the 256 CUs), 160 x 32 per wave.
// Plane rows are 32 B (16 k); the 16-byte chunk position is XOR-swizzled by (row >> 3) & 1.
template <int RB, int CB, int WR, int NSETS>
__global__ __launch_bounds__(512, 2) void split_gemm_v3_kernel(const SplitArgs a, const int row_tiles, const int col_tiles,
                                                               const int nwg) {
  constexpr int TM = WR * RB * 32, WC = 8 / WR;
  static_assert(WC * CB * 32 == 256, "256-column tiles");
  constexpr int kSt = TM * 64, kPa = TM * 32, kPb = 256 * 32;
  constexpr int NA = TM / 16;  // A staging DMA instructions per slab (16 rows each)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const Ast = lds;
  unsigned char* const Apl = lds + 2 * kSt;
  unsigned char* const Bpl = Apl + 2 * 3 * kPa;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int K = a.K, nk = K >> 4;
  const size_t pstride = (size_t)a.N * K;
  const int tiles = row_tiles * col_tiles;
  const int t0 = (int)((long)tiles * blockIdx.x / nwg), t1 = (int)((long)tiles * (blockIdx.x + 1) / nwg);
  if (t0 >= t1) return;
  const int nsteps = (t1 - t0) * nk;
  int i_tile = t0, i_ks = 0, i_step = 0;
  const int a_lrow = lane >> 2, a_chunk = lane & 3;                        // A staging: 16 rows x 4 chunks
  const int b_lrow = lane >> 1, b_chunk = (lane & 1) ^ ((lane >> 4) & 1);  // W planes: 32 rows x 2 chunk slots, row >> 3 = lane >> 4
  auto dma = [&]() {
    const int rt = i_tile / col_tiles, ct = i_tile - rt * col_tiles;
    const int sa = i_step & 1, sb = i_step % 3;
#pragma unroll
    for (int j = 0; j < (NA + 7) / 8; ++j) {
      const int blk = wave + 8 * j;
      if (blk < NA) {
        const int row = rt * TM + 16 * blk + a_lrow;
        glds16(a.A + (size_t)row * a.lda + i_ks * 16 + a_chunk * 4, Ast + sa * kSt + blk * 1024);
      }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int col = ct * 256 + 32 * wave + b_lrow;
      glds16(a.Wp + p * pstride + (size_t)col * K + i_ks * 16 + b_chunk * 8, Bpl + (sb * 3 + p) * kPb + wave * 1024);
    }
    ++i_step;
    if (++i_ks == nk) { i_ks = 0; ++i_tile; if (i_tile >= t1) i_tile = t1 - 1; }
  };
  const int s_row = tid >> 1, s_h = tid & 1;
  const int s_src = s_row * 64 + s_h * 32;
  const int s_dst = s_row * 32 + ((s_h ^ ((s_row >> 3) & 1)) * 16);
  auto split = [&](int step) {
    if (TM * 2 < 512 && tid >= TM * 2) return;
    const unsigned char* src = Ast + (step & 1) * kSt + s_src;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 16);
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split3(x0[e], hh[e], mm[e], ll[e]); split3(x1[e], hh[4 + e], mm[4 + e], ll[4 + e]); }
    u32x4 vh, vm, vl;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      vh[d] = pack_hi(hh[2 * d], hh[2 * d + 1]);
      vm[d] = pack_hi(mm[2 * d], mm[2 * d + 1]);
      vl[d] = pack_hi(ll[2 * d], ll[2 * d + 1]);
    }
    unsigned char* dst = Apl + (step & 1) * 3 * kPa + s_dst;
    *reinterpret_cast<u32x4*>(dst) = vh;
    *reinterpret_cast<u32x4*>(dst + kPa) = vm;
    *reinterpret_cast<u32x4*>(dst + 2 * kPa) = vl;
  };
  const int f_hi = lane >> 5;
  int offa[RB], offb[CB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = wr * RB * 32 + r * 32 + (lane & 31);
    offa[r] = row * 32 + ((f_hi ^ ((row >> 3) & 1)) * 16);
  }
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int col = wc * CB * 32 + c * 32 + (lane & 31);
    offb[c] = col * 32 + ((f_hi ^ ((col >> 3) & 1)) * 16);
  }
  f32x16 acc[NSETS][RB][CB];
#pragma unroll
  for (int q = 0; q < NSETS; ++q)
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[q][r][c][v] = 0.f;
  auto mma = [&](int step) {
    const unsigned char* ap = Apl + (step & 1) * 3 * kPa;
    const unsigned char* bp = Bpl + (step % 3) * 3 * kPb;
    bf16x8 af[RB][3], bf[CB][3];
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + p * kPb + offb[c]));
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[r][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ap + p * kPa + offa[r]));
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int r0 = 0; r0 < RB; r0 += 2)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = r0; r < (r0 + 2 < RB ? r0 + 2 : RB); ++r)
#pragma unroll
          for (int c = 0; c < CB; ++c)
            acc[t * NSETS / 6][r][c] =
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][PA[t]], bf[c][PB[t]], acc[t * NSETS / 6][r][c], 0, 0, 0);
  };
  dma();
  dma();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  split(0);
  int tile = t0, ks = 0;
  bool stored = false;
  for (int g = 0; g < nsteps; ++g) {
    if (stored) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stored = false;
    dma();  // slab g + 2
    if ((wave & 1) == 0) {
      if (g + 1 < nsteps) split(g + 1);
      mma(g);
    } else {
      mma(g);
      if (g + 1 < nsteps) split(g + 1);
    }
    if (++ks == nk) {
      ks = 0;
      const int rt = tile / col_tiles, ct = tile - rt * col_tiles;
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int col = ct * 256 + wc * CB * 32 + c * 32 + (lane & 31);
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = rt * TM + wr * RB * 32 + r * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
            float y = acc[0][r][c][v];
#pragma unroll
            for (int q = 1; q < NSETS; ++q) y += acc[q][r][c][v];
            y += bv;
            if (a.resid) y += a.resid[(size_t)row * a.ldr + col];
            a.Y[(size_t)row * a.ldy + col] = y;
#pragma unroll
            for (int q = 0; q < NSETS; ++q) acc[q][r][c][v] = 0.f;
          }
      }
      stored = true;
      ++tile;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int RB, int CB, int WR, int NSETS>
static float run_v3(const SplitArgs& a, int reps) {
  constexpr int TM = WR * RB * 32;
  constexpr int kLdsV3 = 2 * TM * 64 + 2 * 3 * TM * 32 + 3 * 3 * 256 * 32;
  auto* kern = split_gemm_v3_kernel<RB, CB, WR, NSETS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsV3));
  int n_cu = 256;
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
  if (a.M % TM || a.N % 256) return -1.f;
  const int rt = a.M / TM, ct = a.N / 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), kLdsV3, 0, a, rt, ct, n_cu);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(n_cu), dim3(512), kLdsV3, 0, a, rt, ct, n_cu);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

template <int NPROD, int ORDER = 0, bool SW = false>
static float run(const SplitArgs& a, int reps) {
  const int kLds = 6 * 128 * (SW ? 64 : 80);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(split_gemm_kernel<NPROD, ORDER, SW>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
  dim3 grid(a.M / 128, a.N / 128);
  if (ORDER == 1) grid = dim3(((a.M / 128) * (a.N / 128) + 7) / 8 * 8, 1);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((split_gemm_kernel<NPROD, ORDER, SW>), grid, dim3(256), kLds, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((split_gemm_kernel<NPROD, ORDER, SW>), grid, dim3(256), kLds, 0, a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 81920;
  struct Shape { int K, N, res; };
  const Shape all_shapes[] = {{256, 1024, 0}, {256, 768, 0}, {1024, 256, 1}, {256, 256, 1}, {768, 256, 0}};
  std::vector<Shape> shapes(all_shapes, all_shapes + (argc > 2 ? 1 : 5));
  const int maxK = 1024, maxN = 1024;
  std::vector<float> hA((size_t)M * maxK), hW((size_t)maxK * maxN), hb(maxN), hR((size_t)M * 256);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };  // 24 random bits
  for (auto& v : hA) v = rnd() * 4.f;
  for (auto& v : hW) v = rnd() * 0.25f;
  for (auto& v : hb) v = rnd();
  for (auto& v : hR) v = rnd();
  float *dA, *dW, *db, *dY, *dR;
  uint16_t* dWp;
  CK(hipMalloc(&dA, hA.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dWp, hW.size() * 2 * 3));
  CK(hipMalloc(&db, hb.size() * 4));
  CK(hipMalloc(&dR, hR.size() * 4));
  CK(hipMalloc(&dY, (size_t)M * maxN * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> y((size_t)M * maxN);
  const int reps = 20;
  for (const Shape& sh : shapes) {
    const size_t nw = (size_t)sh.N * sh.K;
    hipLaunchKernelGGL(split_planes_kernel, dim3(512), dim3(256), 0, 0, dW, dWp, nw);
    SplitArgs a;
    a.A = dA; a.Wp = dWp; a.bias = db; a.resid = sh.res ? dR : nullptr; a.Y = dY;
    a.lda = sh.K; a.ldr = sh.N; a.ldy = sh.N; a.M = M; a.K = sh.K; a.N = sh.N;
    const double gf = 2.0 * M * sh.K * sh.N * 1e-9;
    printf("M=%d K=%d N=%d resid=%d  (%.1f GF; %.1f us at the 157.3 TF/s f32-MFMA roof)\n", M, sh.K, sh.N, sh.res, gf, gf / 157.3e3 * 1e6);
    auto check = [&](const char* name, float us) {
      CK(hipMemcpy(y.data(), dY, (size_t)M * sh.N * 4, hipMemcpyDeviceToHost));
      // fp64 reference and a sequential fp32 sum on 2048 sampled elements
      double e_max = 0, e_sq = 0, f_max = 0, f_sq = 0, scale = 0;
      const int ns = 2048;
      for (int i = 0; i < ns; ++i) {
        const int r = (int)(((uint64_t)i * 2654435761u) % M), c = (int)(((uint64_t)i * 40503u + 7) % sh.N);
        double ref = hb[c];
        float f32 = 0.f;
        for (int k = 0; k < sh.K; ++k) {
          ref += (double)hA[(size_t)r * sh.K + k] * hW[(size_t)c * sh.K + k];
          f32 = fmaf(hA[(size_t)r * sh.K + k], hW[(size_t)c * sh.K + k], f32);
        }
        f32 += hb[c];
        if (sh.res) { ref += hR[(size_t)r * sh.N + c]; f32 += hR[(size_t)r * sh.N + c]; }
        const double e = fabs(ref - y[(size_t)r * sh.N + c]), f = fabs(ref - f32);
        e_max = fmax(e_max, e); e_sq += e * e; f_max = fmax(f_max, f); f_sq += f * f; scale = fmax(scale, fabs(ref));
      }
      printf("  %-22s %8.1f us  %6.1f TF/s-equivalent (%.2fx the f32 roof)   err vs fp64: max %.2e rms %.2e   [sequential fp32 fma: max %.2e rms %.2e; |y| max %.1f]\n",
             name, us, gf * 1e3 / us, gf * 1e3 / us / 157.3, e_max, sqrt(e_sq / ns), f_max, sqrt(f_sq / ns), scale);
    };
    if (argc > 2) {
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v3 256x256", run_v3<4, 2, 2, 1>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v3 160x256", run_v3<5, 1, 1, 1>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v3 160x256, 2 acc sets", run_v3<5, 1, 1, 2>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 1 acc set", run_v2<1, true, true, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 3 acc sets", run_v2<3, true, true, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 no mma", run_v2<3, true, false, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 mma only", run_v2<3, false, true, false>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 dma only", run_v2<3, false, false, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 mma only, no barrier", run_v2<3, false, true, false, true, false>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 mma only, no lds", run_v2<3, false, true, false, false, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 mma only, neither", run_v2<3, false, true, false, true, true>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("v2 mma only 1 set, neither", run_v2<1, false, true, false, true, true>(a, reps));
    }
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("6 products", run<6>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("6 products, xcd order", run<6, 1>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("6 products, 3 wg/cu", run<6, 0, true>(a, reps));
    if (argc > 2) {
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("9 products", run<9>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("3 products (bf16x2)", run<3>(a, reps));
    CK(hipMemset(dY, 0, (size_t)M * sh.N * 4)); check("1 product (plain bf16)", run<1>(a, reps));
    }
  }
  return 0;
}
