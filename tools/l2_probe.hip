// How fast can every CU stream the SAME small (L2-resident) buffer?  (the weight-stream pattern of mlp.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ buf, int n_vec4, int passes, int rotate,
                                             float* out) {
  // each wave reads 1 KiB (64 lanes x 16 B) per load, DEPTH independent loads in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = n_vec4 / 64;  // 1-KiB chunks
  f32x4 acc = {0, 0, 0, 0};
  int start = rotate ? (int)((blockIdx.x * 37u + wave * 11u) % (unsigned)nchunk) : wave;
  for (int p = 0; p < passes; ++p) {
    for (int c = 0; c < nchunk; c += 4 * DEPTH) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int ch = start + c + d * 4;
        ch %= nchunk;
        v[d] = reinterpret_cast<const f32x4*>(buf)[(size_t)ch * 64 + lane];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[0] = acc[0];
}

int main() {
  float *buf, *out;
  const size_t bytes = 1 << 20;  // 1 MiB: fits every XCD's 4 MiB L2
  (void)hipMalloc(&buf, bytes);
  (void)hipMalloc(&out, 64);
  (void)hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int n_vec4 = bytes / 16, passes = 20;
  for (int rotate : {0, 1})
    for (int bpc : {1, 2, 4}) {
      auto run = [&](auto kern, int depth) {
        const int grid = 256 * bpc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, n_vec4, 2, rotate, out);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, n_vec4, passes, rotate, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        // every wave reads 1/4 of the buffer per pass (chunks strided by 4 over the waves' DEPTH lanes)
        const double tot = (double)grid * 4 * (double)(n_vec4 / 64 / 4) * 1024.0 * passes;
        printf("rotate=%d blocks/CU=%d depth=%d: %8.2f TB/s aggregate  %6.1f B/clk/CU@2.4GHz\n", rotate, bpc, depth,
               tot / (ms * 1e-3) / 1e12, tot / (ms * 1e-3) / 256 / 2.4e9);
      };
      run(probe<1>, 1);
      run(probe<2>, 2);
      run(probe<4>, 4);
      run(probe<8>, 8);
    }
  return 0;
}
