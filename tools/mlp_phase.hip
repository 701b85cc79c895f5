// Per-phase cycle breakdown of mlp_fwd_kernel for one workgroup (debug build of csrc/mlp.hip).
#define OSRL_PHASE_TIMING 1
#include "../osrl_amd/csrc/mlp.hip"
#include <cstdio>
#include <vector>

static int r16(int x) { return (x + 15) / 16 * 16; }

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 2048, tile = argc > 2 ? atoi(argv[2]) : 16;
  const int E = argc > 3 ? atoi(argv[3]) : 2, H = argc > 4 ? atoi(argv[4]) : 256, K0 = 78, NO = argc > 5 ? atoi(argv[5]) : 1;
  const int dims[4] = {K0, H, H, NO};
  osrl_mlp_t net{};
  net.n_layers = 3;
  net.n_nets = E;
  for (int i = 0; i < 4; ++i) net.dims[i] = dims[i];
  net.acts[0] = net.acts[1] = OSRL_ACT_RELU;
  net.acts[2] = OSRL_ACT_ID;
  net.out_scale = 1.f;
  net.tile_rows = tile;
  size_t ncan = 0, nf = 0;
  std::vector<osrl_pack_entry_t> ents;
  for (int e = 0; e < E; ++e)
    for (int l = 0; l < 3; ++l) {
      osrl_pack_entry_t pe{(int64_t)ncan, (int64_t)nf, -1, dims[l + 1], dims[l]};
      ents.push_back(pe);
      ncan += (size_t)dims[l] * dims[l + 1];
      nf += (size_t)r16(dims[l]) * r16(dims[l + 1]);
    }
  float *can, *pf, *bias, *x, *y;
  osrl_pack_entry_t* dents;
  (void)hipMalloc(&can, ncan * 4);
  (void)hipMalloc(&pf, nf * 4);
  (void)hipMalloc(&bias, 4096);
  (void)hipMalloc(&x, (size_t)rows * K0 * 4);
  (void)hipMalloc(&y, (size_t)E * rows * NO * 4);
  (void)hipMalloc(&dents, ents.size() * sizeof(osrl_pack_entry_t));
  (void)hipMemset(can, 0, ncan * 4);
  (void)hipMemset(bias, 0, 4096);
  (void)hipMemset(x, 0, (size_t)rows * K0 * 4);
  (void)hipMemcpy(dents, ents.data(), ents.size() * sizeof(osrl_pack_entry_t), hipMemcpyHostToDevice);
  osrl_pack_weights(can, pf, nullptr, dents, (int)ents.size(), 1 << 18, nullptr);
  for (int e = 0, i = 0; e < E; ++e)
    for (int l = 0; l < 3; ++l, ++i) {
      net.Wf[e][l] = pf + ents[i].f_off;
      net.b[e][l] = bias;
    }
  osrl_rows_t in{};
  in.rows = rows;
  in.d0 = K0;
  in.src0 = x;
  osrl_mlp_acts_t out{};
  for (int e = 0; e < E; ++e) out.h[e][2] = y + (size_t)e * rows * NO;
  for (int i = 0; i < 3; ++i) osrl_mlp_forward(&net, &in, &out, nullptr);
  (void)hipDeviceSynchronize();
  long long t[4][64];
  (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_t), sizeof(t));
  const char* names[] = {"stage-in", "L0 mm", "L0 barrier", "L0 epilogue", "L0 sync+save", "L1 mm", "L1 barrier", "L1 epilogue",
                         "L1 sync+save", "L2 mm", "L2 barrier", "L2 epilogue", "L2 sync+save"};
  printf("rows=%d tile=%d E=%d H=%d out=%d   (cycles of s_memtime @100MHz? raw deltas per wave)\n", rows, tile, E, H, NO);
  for (int w = 0; w < 4; ++w) {
    printf("wave %d:", w);
    for (int i = 0; i < 13; ++i) printf(" %s=%lld", names[i], t[w][i + 1] - t[w][i]);
    printf("  total=%lld\n", t[w][13] - t[w][0]);
  }
  return 0;
}
