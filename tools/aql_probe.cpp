// Own HSA queue + hand-written AQL packets: do two dispatches of ONE queue overlap on gfx950 when the second packet's
// barrier bit is clear?  (hipExtAnyOrderLaunch is ignored on GFX9, tools/anyorder_probe.hip.)
//   g++ -O2 -I/opt/rocm/include tools/aql_probe.cpp -o tools/aql_probe.bin -L/opt/rocm/lib -lhsa-runtime64
//   tools/aql_probe.bin tools/aql_probe_kernels.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); printf("%s -> %s (line %d)\n", #x, m ? m : "?", __LINE__); exit(1); } } while (0)

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool; static bool have_dev = false, have_karg = false;
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !have_dev) { g_dev_pool = p; have_dev = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_karg) { g_karg_pool = p; have_karg = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kern { uint64_t obj; uint32_t karg, lds, priv; };
static Kern get_kernel(hsa_executable_t exe, const char* name) {
  hsa_executable_symbol_t sym; CK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
  Kern k;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.obj));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  printf("%s: object %#lx kernarg %u lds %u private %u\n", name, (unsigned long)k.obj, k.karg, k.lds, k.priv);
  return k;
}

static hsa_queue_t* q;
static void dispatch(const Kern& k, void* kargs, uint32_t wgs_x, uint32_t wgs_y, uint32_t block, uint32_t dyn_lds, bool barrier,
                     hsa_signal_t done) {
  uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
  while (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
  hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
  p->setup = 2 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  p->workgroup_size_x = block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
  p->grid_size_x = wgs_x * block; p->grid_size_y = wgs_y; p->grid_size_z = 1;
  p->private_segment_size = k.priv; p->group_segment_size = k.lds + dyn_lds;
  p->kernel_object = k.obj; p->kernarg_address = kargs; p->reserved2 = 0; p->completion_signal = done;
  uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                    (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                    (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
  hsa_signal_store_screlease(q->doorbell_signal, idx);
}

int main(int argc, char** argv) {
  CK(hsa_init());
  CK(hsa_iterate_agents(on_agent, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));
  char nm[64]; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, nm); printf("gpu agent %s dev pool %d kernarg pool %d\n", nm, have_dev, have_karg);
  FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> co(n); if (fread(co.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
  hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(co.data(), n, &rd));
  hsa_executable_t exe; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CK(hsa_executable_load_agent_code_object(exe, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(exe, nullptr));
  Kern spin = get_kernel(exe, "spin_kernel.kd"), dims = get_kernel(exe, "dims_kernel.kd");
  CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  // kernargs + outputs: host kernarg pool (GPU-visible) for simplicity
  char* karg; CK(hsa_amd_memory_pool_allocate(g_karg_pool, 1 << 16, 0, (void**)&karg)); CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, karg));
  int* out; CK(hsa_amd_memory_pool_allocate(g_karg_pool, 4096, 0, (void**)&out)); CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, out));
  memset(out, 0, 4096);
  hsa_signal_t done; CK(hsa_signal_create(1, 0, nullptr, &done));
  hsa_signal_t none = {0};
  auto now = [] { return std::chrono::steady_clock::now(); };
  // --- hidden arguments check
  {
    struct { int* out; uint32_t bc[3]; uint16_t gs[3]; uint16_t rem[3]; char pad[16]; uint64_t off[3]; uint16_t dims; char pad2[46]; uint32_t dyn; } a;
    memset(&a, 0, sizeof a);
    a.out = out; a.bc[0] = 7; a.bc[1] = 3; a.bc[2] = 1; a.gs[0] = 128; a.gs[1] = 1; a.gs[2] = 1; a.dims = 2; a.dyn = 512;
    printf("offsets: bc %zu gs %zu off %zu dims %zu dyn %zu\n", (char*)a.bc - (char*)&a, (char*)a.gs - (char*)&a, (char*)a.off - (char*)&a, (char*)&a.dims - (char*)&a, (char*)&a.dyn - (char*)&a);
    memcpy(karg + 8192, &a, sizeof a);
    hsa_signal_store_relaxed(done, 1);
    dispatch(dims, karg + 8192, 7, 3, 128, 512, true, done);
    hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    printf("dims kernel saw gridDim (%d,%d) blockDim %d blockIdx.x %d lds[last] %d   (expect 7 3 128 6 127)\n", out[0], out[1], out[2], out[3], out[4]);
  }
  // --- overlap
  const long long us30 = 100 * 30;  // wall clock 100 MHz
  struct SpinArgs { long long ticks; int* flag; int val; };
  auto run = [&](const char* name, int wg1, int wg2, bool barrier2, uint32_t lds, int n_more) {
    SpinArgs a{us30, nullptr, 0}; memcpy(karg, &a, sizeof a);
    for (int rep = 0; rep < 2; ++rep) {
      const int N = 50;
      auto t0 = now();
      for (int i = 0; i < N; ++i) {
        dispatch(spin, karg, wg1, 1, 256, lds, true, none);
        for (int k = 0; k <= n_more; ++k) {
          bool last = (i == N - 1 && k == n_more);
          if (last) hsa_signal_store_relaxed(done, 1);
          dispatch(spin, karg, wg2, 1, 256, lds, barrier2, last ? done : none);
        }
      }
      // the last packet may finish before earlier ones when its barrier bit is clear: close with a barrier'd empty-ish dispatch
      hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      SpinArgs z{0, nullptr, 0}; memcpy(karg + 4096, &z, sizeof z);
      hsa_signal_store_relaxed(done, 1);
      dispatch(spin, karg + 4096, 1, 1, 64, 0, true, done);
      hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      double us = std::chrono::duration<double, std::micro>(now() - t0).count() / N;
      if (rep) printf("%-64s %7.1f us per group (sum of spins %d us)\n", name, us, 30 * (2 + n_more));
    }
  };
  run("K(128) ; K(128) barrier bit set", 128, 128, true, 0, 0);
  run("K(128) ; K(128) barrier bit CLEAR on the second", 128, 128, false, 0, 0);
  run("K(256,100KB) ; K(256,100KB) barrier bit set", 256, 256, true, 100 * 1024, 0);
  run("K(256,100KB) ; K(256,100KB) barrier bit clear (no room: serial)", 256, 256, false, 100 * 1024, 0);
  run("K(128) ; 3 x K(64) barrier bit clear", 128, 64, false, 0, 2);
  run("K(1) ; 7 x K(1) barrier bit clear", 1, 1, false, 0, 6);
  run("K(1024) ; K(1024) barrier bit clear (8 WG/CU fit)", 1024, 1024, false, 0, 0);
  // --- ordering: P1 (barrier) short, P2 (no barrier) long, P3 (barrier) must see both
  {
    int bad = 0;
    for (int i = 0; i < 200; ++i) {
      out[8] = 0; out[9] = 0;
      SpinArgs a1{us30 / 3, out + 8, 1}, a2{us30, out + 9, 2};
      memcpy(karg + 256, &a1, sizeof a1); memcpy(karg + 512, &a2, sizeof a2);
      SpinArgs z{0, nullptr, 0}; memcpy(karg + 4096, &z, sizeof z);
      dispatch(spin, karg + 256, 64, 1, 256, 0, true, none);
      dispatch(spin, karg + 512, 64, 1, 256, 0, false, none);
      hsa_signal_store_relaxed(done, 1);
      dispatch(spin, karg + 4096, 1, 1, 64, 0, true, done);
      hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
      bad += !(out[8] == 1 && out[9] == 2);
    }
    printf("ordering: a barrier packet after (barrier, no-barrier) completed after both in %d / 200 runs\n", 200 - bad);
  }
  // --- host cost per packet
  {
    SpinArgs z{0, nullptr, 0}; memcpy(karg + 4096, &z, sizeof z);
    auto t0 = now();
    for (int i = 0; i < 20000; ++i) dispatch(spin, karg + 4096, 1, 1, 64, 0, true, none);
    double host = std::chrono::duration<double, std::micro>(now() - t0).count() / 20000;
    hsa_signal_store_relaxed(done, 1);
    dispatch(spin, karg + 4096, 1, 1, 64, 0, true, done);
    hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
    double tot = std::chrono::duration<double, std::micro>(now() - t0).count() / 20000;
    printf("empty kernel, barrier packets: host %.2f us per packet, %.2f us per packet incl. drain\n", host, tot);
  }
  return 0;
}
