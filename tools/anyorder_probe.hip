// Does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two kernels of ONE stream overlap on gfx950,
// eagerly and inside a captured hipGraph?  hipcc --offload-arch=gfx950 -O2 tools/anyorder_probe.hip -o tools/anyorder_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); } } while (0)

__global__ void spin_kernel(long long ticks, int* flag, int val) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
  if (flag && threadIdx.x == 0 && blockIdx.x == 0) { __atomic_store_n(flag, val, __ATOMIC_RELEASE); }
}
__global__ void check_kernel(const int* f1, const int* f2, int* out) {
  if (threadIdx.x == 0) out[0] = (__atomic_load_n(f1, __ATOMIC_ACQUIRE) == 1) + 2 * (__atomic_load_n(f2, __ATOMIC_ACQUIRE) == 2);
}

static void launch(int wgs, long long ticks, int* flag, int val, hipStream_t s, int flags, size_t lds = 0) {
  void* args[] = {&ticks, &flag, &val};
  CK(hipExtLaunchKernel((const void*)spin_kernel, dim3(wgs), dim3(256), args, lds, s, nullptr, nullptr, flags));
}

int main() {
  int clk = 0;
  CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0));  // kHz
  printf("wall clock rate %d kHz\n", clk);
  const long long us30 = (long long)clk * 30 / 1000;
  hipStream_t s; CK(hipStreamCreate(&s));
  int *f; CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto run = [&](const char* name, int wg1, int wg2, int flag2, size_t lds, int n3 = 0) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(s));
      auto t0 = now();
      const int N = 50;
      for (int i = 0; i < N; ++i) {
        launch(wg1, us30, nullptr, 0, s, 0, lds);
        launch(wg2, us30, nullptr, 0, s, flag2, lds);
        for (int k = 0; k < n3; ++k) launch(wg2, us30, nullptr, 0, s, flag2, lds);
      }
      CK(hipStreamSynchronize(s));
      double us = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
      if (rep) printf("%-58s %7.1f us per group (sum of spins %d us)\n", name, us, 30 * (2 + n3));
    }
  };
  run("eager  K(128) ; K(128) in order", 128, 128, 0, 0);
  run("eager  K(128) ; K(128) any-order", 128, 128, hipExtAnyOrderLaunch, 0);
  run("eager  K(256,100KB LDS) ; K(256,100KB LDS) in order", 256, 256, 0, 100 * 1024);
  run("eager  K(256,100KB LDS) ; K(256,100KB LDS) any-order", 256, 256, hipExtAnyOrderLaunch, 100 * 1024);
  run("eager  K(128) ; 3 x K(64) any-order", 128, 64, hipExtAnyOrderLaunch, 0, 2);
  run("eager  K(1) ; K(1) any-order", 1, 1, hipExtAnyOrderLaunch, 0);
  // ordering: K1 in order, K2 any-order, K3 in order must see both flags
  int bad = 0;
  for (int i = 0; i < 200; ++i) {
    CK(hipMemsetAsync(f, 0, 64, s));
    launch(64, us30 / 3, f, 1, s, 0);
    launch(64, us30, f + 1, 2, s, hipExtAnyOrderLaunch);
    int* o = f + 2; const int* f1 = f; const int* f2 = f + 1;
    void* a[] = {&f1, &f2, &o};
    CK(hipExtLaunchKernel((const void*)check_kernel, dim3(1), dim3(64), a, 0, s, nullptr, nullptr, 0));
    int h = 0; CK(hipMemcpyAsync(&h, o, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    bad += (h != 3);
  }
  printf("ordering check: in-order kernel after (in-order, any-order) pair saw both results in %d / 200 runs\n", 200 - bad);
  // inside a captured graph
  for (int flag2 : {0, (int)hipExtAnyOrderLaunch}) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 10; ++i) { launch(128, us30, nullptr, 0, s, 0); launch(128, us30, nullptr, 0, s, flag2); }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(s));
      auto t0 = now();
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      double us = std::chrono::duration<double, std::micro>(now() - t0).count() / 100;
      if (rep) printf("graph  K(128) ; K(128) %s: %7.1f us per pair\n", flag2 ? "any-order" : "in order ", us);
    }
  }
  // host cost of a launch
  {
    CK(hipStreamSynchronize(s));
    auto t0 = now();
    for (int i = 0; i < 2000; ++i) launch(1, 0, nullptr, 0, s, hipExtAnyOrderLaunch);
    double host = std::chrono::duration<double, std::micro>(now() - t0).count() / 2000;
    CK(hipStreamSynchronize(s));
    double tot = std::chrono::duration<double, std::micro>(now() - t0).count() / 2000;
    printf("empty kernel, any-order: host %0.2f us per launch, %0.2f us per launch incl. drain\n", host, tot);
    t0 = now();
    for (int i = 0; i < 2000; ++i) launch(1, 0, nullptr, 0, s, 0);
    host = std::chrono::duration<double, std::micro>(now() - t0).count() / 2000;
    CK(hipStreamSynchronize(s));
    tot = std::chrono::duration<double, std::micro>(now() - t0).count() / 2000;
    printf("empty kernel, in order : host %0.2f us per launch, %0.2f us per launch incl. drain\n", host, tot);
  }
  return 0;
}
