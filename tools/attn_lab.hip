// attn_lab.hip -- the attention core of csrc/cdt.hip (osrl_attention_fwd / _bwd) on its own: C5's shape (1024 x 8 heads,
// 80 tokens, head width 32, probability dropout 0.1, ragged key padding) against a double-precision CPU restatement of
// nn.MultiheadAttention's core (net.py:406-409,417-435) on a sample of (sample, head) pairs, + per-launch times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/attn_lab.hip -o tools/_lab/attn_lab
//   (the flag is cdt.hip's own in osrl_amd/build.py FILE_FLAGS)
#define ATTN_STAMPS 1
#include "../osrl_amd/csrc/cdt.hip"
#undef S
#undef CLEAR
#undef DONE

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 1024, SEQ = argc > 2 ? atoi(argv[2]) : 80, E = 256, H = argc > 4 ? atoi(argv[4]) : 8, REP = 4;
  const float pdrop = argc > 3 ? (float)atof(argv[3]) : 0.1f;
  const int d = E / H, T = SEQ / REP, Sp = (SEQ + 15) & ~15;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> qkv((size_t)B * SEQ * 3 * E), dout((size_t)B * SEQ * E), mask((size_t)B * T);
  for (auto& v : qkv) v = 0.7f * nd(rng);
  for (auto& v : dout) v = nd(rng);
  for (int b = 0; b < B; ++b) {
    const int pad = T > 2 ? (b * 7) % (T - 2) : 0;  // 0 .. T-3 padded timesteps
    for (int t = 0; t < T; ++t) {
      bool ok = true;
      if (b % 3 == 0) ok = t < T - pad;   // padded at the end (dataset.py pads the tail)
      if (b % 3 == 1) ok = t >= pad;      // padded at the front: whole query rows without a valid key
      mask[(size_t)b * T + t] = ok ? 1.f : 0.f;
    }
  }
  float *d_qkv, *d_dout, *d_mask, *d_o, *d_dqkv, *d_ones, *d_keep;
  osrl_step_state_t* d_st;
  CK(hipMalloc(&d_qkv, qkv.size() * 4));
  CK(hipMalloc(&d_dout, dout.size() * 4));
  CK(hipMalloc(&d_mask, mask.size() * 4));
  CK(hipMalloc(&d_o, dout.size() * 4));
  CK(hipMalloc(&d_dqkv, qkv.size() * 4));
  const size_t nmask = (size_t)B * H * SEQ * Sp;
  CK(hipMalloc(&d_ones, nmask * 4));
  CK(hipMalloc(&d_keep, nmask * 4));
  CK(hipMalloc(&d_st, sizeof(osrl_step_state_t)));
  osrl_step_state_t st{};
  st.step = 3;
  CK(hipMemcpy(d_st, &st, sizeof(st), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_dout, dout.data(), dout.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_mask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_dqkv, 0, qkv.size() * 4));
  std::vector<float> ones(nmask, 1.f), keep(nmask, 1.f);
  CK(hipMemcpy(d_ones, ones.data(), nmask * 4, hipMemcpyHostToDevice));
  osrl_dropout_t dr{pdrop, 4u, 0x1234567ull, d_st};
  const osrl_dropout_t* drp = pdrop > 0.f ? &dr : nullptr;
  if (drp) {
    if (osrl_dropout(d_ones, d_keep, (int64_t)nmask, drp, nullptr)) { printf("osrl_dropout failed\n"); return 1; }
    CK(hipMemcpy(keep.data(), d_keep, nmask * 4, hipMemcpyDeviceToHost));
  }
  if (osrl_attention_fwd(d_qkv, d_mask, B, SEQ, E, H, REP, 0, drp, d_o, nullptr)) { printf("fwd failed\n"); return 1; }
  if (osrl_attention_bwd(d_qkv, d_mask, d_dout, B, SEQ, E, H, REP, 0, drp, d_dqkv, nullptr)) { printf("bwd failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> o(dout.size()), dqkv(qkv.size());
  CK(hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(dqkv.data(), d_dqkv, dqkv.size() * 4, hipMemcpyDeviceToHost));

  // ---- CPU restatement on a sample of (sample, head) pairs
  double worst[4] = {0, 0, 0, 0}, scale_[4] = {0, 0, 0, 0};
  const double sc = 1.0 / std::sqrt((double)d);
  int npairs = 0;
  for (int b = 0; b < B; b += (B >= 64 ? B / 48 : 1)) {
    for (int h = (b % 2); h < H; h += 3) {
      ++npairs;
      std::vector<double> P((size_t)SEQ * SEQ, 0.0), Pk((size_t)SEQ * SEQ, 0.0), dPk((size_t)SEQ * SEQ, 0.0), dS((size_t)SEQ * SEQ, 0.0);
      auto Q = [&](int i, int c) { return (double)qkv[((size_t)b * SEQ + i) * 3 * E + h * d + c]; };
      auto K = [&](int i, int c) { return (double)qkv[((size_t)b * SEQ + i) * 3 * E + E + h * d + c]; };
      auto V = [&](int i, int c) { return (double)qkv[((size_t)b * SEQ + i) * 3 * E + 2 * E + h * d + c]; };
      auto dO = [&](int i, int c) { return (double)dout[((size_t)b * SEQ + i) * E + h * d + c]; };
      for (int i = 0; i < SEQ; ++i) {
        double mx = -INFINITY;
        std::vector<double> s(SEQ, -INFINITY);
        for (int j = 0; j <= i; ++j) {
          if (mask[(size_t)b * T + j / REP] <= 0.f) continue;
          double acc = 0;
          for (int c = 0; c < d; ++c) acc += Q(i, c) * K(j, c);
          s[j] = acc * sc;
          mx = std::max(mx, s[j]);
        }
        if (mx == -INFINITY) continue;  // no valid key: the kernels give a zero row
        double sum = 0;
        for (int j = 0; j <= i; ++j) if (s[j] > -INFINITY) sum += std::exp(s[j] - mx);
        for (int j = 0; j <= i; ++j) {
          if (s[j] == -INFINITY) continue;
          const double p = std::exp(s[j] - mx) / sum;
          const double km = drp ? (double)keep[(((size_t)(b * H + h) * SEQ + i) * Sp) + j] : 1.0;
          P[(size_t)i * SEQ + j] = p;
          Pk[(size_t)i * SEQ + j] = p * km;
          double g = 0;
          for (int c = 0; c < d; ++c) g += dO(i, c) * V(j, c);
          dPk[(size_t)i * SEQ + j] = g;
        }
        double rd = 0;
        for (int j = 0; j <= i; ++j) rd += Pk[(size_t)i * SEQ + j] * dPk[(size_t)i * SEQ + j];
        for (int j = 0; j <= i; ++j)
          dS[(size_t)i * SEQ + j] = Pk[(size_t)i * SEQ + j] * dPk[(size_t)i * SEQ + j] - P[(size_t)i * SEQ + j] * rd;
      }
      for (int i = 0; i < SEQ; ++i)
        for (int c = 0; c < d; ++c) {
          double ov = 0, dq = 0, dk = 0, dv = 0;
          for (int j = 0; j < SEQ; ++j) {
            ov += Pk[(size_t)i * SEQ + j] * V(j, c);
            dq += dS[(size_t)i * SEQ + j] * K(j, c);
            dk += dS[(size_t)j * SEQ + i] * Q(j, c);
            dv += Pk[(size_t)j * SEQ + i] * dO(j, c);
          }
          dq *= sc;
          dk *= sc;
          const size_t go = ((size_t)b * SEQ + i) * E + h * d + c, gq = ((size_t)b * SEQ + i) * 3 * E + h * d + c;
          const double ref[4] = {ov, dq, dk, dv};
          const double got[4] = {o[go], dqkv[gq], dqkv[gq + E], dqkv[gq + 2 * E]};
          for (int k = 0; k < 4; ++k) {
            worst[k] = std::max(worst[k], std::fabs(ref[k] - got[k]));
            scale_[k] = std::max(scale_[k], std::fabs(ref[k]));
          }
        }
    }
  }
  const char* nm[4] = {"o", "dq", "dk", "dv"};
  bool ok = true;
  printf("B %d  S %d  E %d  H %d  dropout %.2f  (%d (sample, head) pairs against the CPU restatement)\n", B, SEQ, E, H, pdrop, npairs);
  for (int k = 0; k < 4; ++k) {
    printf("  %-3s max|gpu-cpu| %.3e  scale %.3e  rel %.2e\n", nm[k], worst[k], scale_[k], worst[k] / scale_[k]);
    if (!(worst[k] <= 2e-5 * scale_[k])) ok = false;
  }
  // ---- times
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int pass = 0; pass < 2; ++pass) {
    const int iters = 30;
    for (int w = 0; w < 3; ++w)
      pass ? osrl_attention_bwd(d_qkv, d_mask, d_dout, B, SEQ, E, H, REP, 0, drp, d_dqkv, nullptr)
           : osrl_attention_fwd(d_qkv, d_mask, B, SEQ, E, H, REP, 0, drp, d_o, nullptr);
    CK(hipEventRecord(e0));
    for (int w = 0; w < iters; ++w)
      pass ? osrl_attention_bwd(d_qkv, d_mask, d_dout, B, SEQ, E, H, REP, 0, drp, d_dqkv, nullptr)
           : osrl_attention_fwd(d_qkv, d_mask, B, SEQ, E, H, REP, 0, drp, d_o, nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = pass ? (double)B * SEQ * E * 4 * 7 : (double)B * SEQ * E * 4 * 4;
    printf("  %s: %.1f us per launch  (%.2f TB/s of its %0.f MB)\n", pass ? "backward" : "forward ", ms * 1e3 / iters,
           bytes / (ms * 1e-3 / iters) / 1e12, bytes / 1e6);
  }
  if (B * H >= 4608) {  // phase stamps of the LAST timed backward launch: workgroups 4096 .. 4607 (mid-launch: every CU busy)
    static unsigned long long stp[512][4][8];
    CK(hipMemcpyFromSymbol(stp, HIP_SYMBOL(g_attn_stamp), sizeof(stp)));
    const char* ph[5] = {"tiles K,V + barrier", "pass A", "barrier", "tiles Q,dO + barrier", "pass B"};
    for (int w = 0; w < 4; ++w) {
      double acc[5] = {0, 0, 0, 0, 0};
      int n = 0;
      for (int g = 0; g < 512; ++g) {
        if (stp[g][w][5] <= stp[g][w][0]) continue;
        ++n;
        for (int i = 0; i < 5; ++i) acc[i] += (double)(stp[g][w][i + 1] - stp[g][w][i]) * 0.01;
      }
      if (!n) continue;
      printf("  backward, wave %d (us, mean of %d workgroups):", w, n);
      double tot = 0;
      for (int i = 0; i < 5; ++i) { printf("  %s %.2f", ph[i], acc[i] / n); tot += acc[i] / n; }
      printf("  | life %.2f\n", tot);
    }
  }
  printf(ok ? "PARITY OK\n" : "PARITY FAILED\n");
  return ok ? 0 : 1;
}
