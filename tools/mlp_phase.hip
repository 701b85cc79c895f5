// Per-phase cycle breakdown of mlp_fwd_kernel for one workgroup (debug build of csrc/mlp.hip).
#define OSRL_PHASE_TIMING 1
#include "../osrl_amd/csrc/mlp.hip"
#include "../osrl_amd/csrc/mlp_nb.hip"
// (mlp.hip's unfused tail fall-backs call into glue.hip, which this stand-alone build does not link)
extern "C" int osrl_vae_latent(const float*, const float*, int32_t, int32_t, float*, void*) { return -1; }
extern "C" int osrl_vae_latent_bwd(const float*, const float*, const float*, int32_t, int32_t, float, int32_t, float*,
                                   void*) { return -1; }
extern "C" int osrl_gauss_head(const float*, const float*, int32_t, int32_t, float, float*, float*, float*, void*) { return -1; }
extern "C" int osrl_gauss_ood_sample(const float*, const float*, int32_t, int32_t, int32_t, float*, void*) { return -1; }
extern "C" int osrl_vae_kl_rows(const float*, int32_t, int32_t, float*, void*) { return -1; }
// (the 64-row twin of the N*B-row kernels is a second compile of mlp_nb.hip, mlp_nb64.hip: not part of this build)
int osrl_launch_fwd_nb64(const osrl_mlp_t*, const osrl_rows_t*, const osrl_mlp_acts_t*, hipStream_t, float*, int) { return -1; }
namespace osrl_argmem {
Arena* current() { return nullptr; }  // (defined in optim.hip in the library)
}
#include <cstdio>
#include <vector>
#include <algorithm>

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
  net.tile_rows = tile;  // -1 = the LDS-staged-weights kernel (grid and tile heights chosen by the library)
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
  // COLD=1: the packed weights are re-written before every forward (by whatever CUs the pack kernel lands on), as the
  // optimizer step does inside a train step: the forward then finds them in no L2 it can reach
  const bool cold = getenv("COLD") && atoi(getenv("COLD")) != 0;
  for (int i = 0; i < 3; ++i) {
    if (cold) osrl_pack_weights(can, pf, nullptr, dents, (int)ents.size(), 1 << 18, nullptr);
    osrl_mlp_forward(&net, &in, &out, nullptr);
  }
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
  {  // phase durations averaged over every workgroup (wave 0..3)
    static long long pa[8192][4][16];
    (void)hipMemcpyFromSymbol(pa, HIP_SYMBOL(g_phase_all), sizeof(pa));
    const int tl = tile > 0 ? tile : 64;
    const int nw = ((rows + tl - 1) / tl) * E < 8192 ? ((rows + tl - 1) / tl) * E : 8192;
    printf("mean cycles over %d workgroups:", nw);
    double tot = 0;
    for (int i = 0; i < 13; ++i) {
      double sacc = 0;
      for (int g = 0; g < nw; ++g)
        for (int w = 0; w < 4; ++w) sacc += (double)(pa[g][w][i + 1] - pa[g][w][i]);
      sacc /= 4.0 * nw;
      tot += sacc;
      printf(" %s=%.0f", names[i], sacc);
    }
    printf("  total=%.0f\n", tot);
    if (tile == 80) {  // mlp_fwd_nb_kernel stamps the start of layer 1's k-loop (slot 14)
      double pro = 0;
      for (int g = 0; g < nw; ++g)
        for (int w = 0; w < 4; ++w) pro += (double)(pa[g][w][14] - pa[g][w][5]);
      printf("  of L1 mm: %.0f cycles from the layer's start to the first k-step (pointers, bias + first fragments, acc init)\n",
             pro / (4.0 * nw));
    }
  }
  // workgroup residency: how many workgroups does a CU really hold at once?
  static long long wl[16384][4];
  (void)hipMemcpyFromSymbol(wl, HIP_SYMBOL(g_wg_log), sizeof(wl));
  const int BMt = tile > 0 ? tile : 64, nwg = ((rows + BMt - 1) / BMt) * E;
  const int n = nwg < 16384 ? nwg : 16384;
  long long t0 = wl[0][0], t1 = 0;
  for (int i = 0; i < n; ++i) {
    if (wl[i][0] < t0) t0 = wl[i][0];
    if (wl[i][1] > t1) t1 = wl[i][1];
  }
  // CU key = xcc(4b) | se(3b) | sh(1b) | cu(4b)
  int maxc[4096] = {0}, cnt_at_mid[4096] = {0};
  const long long mid = t0 + (t1 - t0) / 3;
  double dur = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = (unsigned)wl[i][2], xcc = (unsigned)wl[i][3] & 15;
    const int key = (int)((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15));
    if (wl[i][0] <= mid && wl[i][1] > mid) cnt_at_mid[key]++;
    dur += (double)(wl[i][1] - wl[i][0]);
    (void)maxc;
  }
  int hist[16] = {0}, cus = 0;
  for (int k = 0; k < 4096; ++k)
    if (cnt_at_mid[k]) {
      hist[cnt_at_mid[k] < 15 ? cnt_at_mid[k] : 15]++;
      ++cus;
    }
  printf("kernel span %.2f us over %d workgroups; mean workgroup life %.2f us; at 1/3 of the span %d CUs hold:", (t1 - t0) / 100.0,
         n, dur / n / 100.0, cus);
  for (int c = 1; c < 16; ++c)
    if (hist[c]) printf("  %d WGs on %d CUs", c, hist[c]);
  printf("\n");
  {  // lockstep metric: distance from a workgroup's start to the nearest other start on the SAME CU, in units of
     // the mean life (0 = the CU's workgroups move in lockstep, 1/occupancy = evenly staggered)
    std::vector<std::vector<long long>> starts(4096);
    for (int i = 0; i < n; ++i) {
      const unsigned hw = (unsigned)wl[i][2], xcc = (unsigned)wl[i][3] & 15;
      starts[(xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)].push_back(wl[i][0]);
    }
    double acc = 0;
    long cntp = 0;
    for (auto& v : starts) {
      std::sort(v.begin(), v.end());
      for (size_t j = 0; j < v.size(); ++j) {
        long long d = 1LL << 60;
        if (j > 0) d = std::min(d, v[j] - v[j - 1]);
        if (j + 1 < v.size()) d = std::min(d, v[j + 1] - v[j]);
        if (d < (1LL << 59)) {
          acc += (double)d;
          ++cntp;
        }
      }
    }
    printf("lockstep metric: mean nearest-start distance on a CU = %.3f of the mean workgroup life\n",
           acc / cntp / (dur / n));
  }
  {  // lifetime distribution, per XCD and per third of the kernel span
    std::vector<double> life(n);
    double xs[16] = {0}, ts[3] = {0};
    int xn[16] = {0}, tn[3] = {0};
    for (int i = 0; i < n; ++i) {
      life[i] = (wl[i][1] - wl[i][0]) / 100.0;
      const int x = (int)(wl[i][3] & 15);
      xs[x] += life[i];
      xn[x]++;
      int th = (int)(3 * (wl[i][0] - t0) / (t1 - t0 + 1));
      ts[th] += life[i];
      tn[th]++;
    }
    std::vector<double> srt(life);
    std::sort(srt.begin(), srt.end());
    printf("workgroup life us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f |", srt[0], srt[n / 10], srt[n / 2],
           srt[n * 9 / 10], srt[n - 1]);
    printf(" by start third:");
    for (int k = 0; k < 3; ++k) printf(" %.1f(n=%d)", tn[k] ? ts[k] / tn[k] : 0.0, tn[k]);
    printf(" | by XCC:");
    for (int x = 0; x < 16; ++x)
      if (xn[x]) printf(" %.1f", xs[x] / xn[x]);
    printf("\n");
  }
  return 0;
}
