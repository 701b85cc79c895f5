// Lab + standalone parity harness for osrl_amd/csrc/vae_ns.hip (round 5): builds a VAE at the BASELINE shapes, packs its
// weights the way osrl_pack_weights does, runs osrl_vae_ns_forward / _backward, compares EVERY buffer they fill with a
// double-precision CPU evaluation of VAE.forward + loss + autograd (net.py:319-339, cpq.py:125-135), and times the two
// calls.  No torch:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/vae_ns_lab.hip -o tools/_lab/vae_ns_lab && tools/_lab/vae_ns_lab
#define VAE_NS_STAMPS 1
#include "../osrl_amd/csrc/vae_ns.hip"

static osrl_argmem::Arena g_ar{nullptr, nullptr, 0, 0, 0, 0, 0, 0};
namespace osrl_argmem {
Arena* current() { return g_ar.mode == kOff ? nullptr : &g_ar; }  // (the library's lives in optim.hip)
}

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static uint32_t g_s = 12345;
static float rnd() {
  g_s = g_s * 1664525u + 1013904223u;
  return ((g_s >> 8) & 0xffff) / 65536.f - 0.5f;
}
static int R16(int x) { return (x + 15) & ~15; }

struct Lin {
  int out, in;
  std::vector<float> W, b;  // [out][in], [out]
  float *dF = nullptr, *dB = nullptr, *db = nullptr;
  void init(int o, int i) {
    out = o; in = i;
    W.resize((size_t)o * i); b.resize(o);
    const float k = 1.0f / std::sqrt((float)i);
    for (auto& v : W) v = 2 * k * rnd();
    for (auto& v : b) v = 2 * k * rnd();
    const int Np = R16(o), Kp = R16(i), Kb = Kp + 16;
    std::vector<float> pf((size_t)Kp * Np, 0.f), pb((size_t)Np * Kb, 0.f);
    for (int n = 0; n < o; ++n)
      for (int kk = 0; kk < i; ++kk) {
        pf[((size_t)(kk / 4) * Np + n) * 4 + (kk & 3)] = W[(size_t)n * i + kk];
        pb[((size_t)(n / 4) * Kb + kk) * 4 + (n & 3)] = W[(size_t)n * i + kk];
      }
    CK(hipMalloc(&dF, pf.size() * 4)); CK(hipMalloc(&dB, pb.size() * 4)); CK(hipMalloc(&db, b.size() * 4));
    CK(hipMemcpy(dF, pf.data(), pf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  }
};

static float* dalloc(size_t n) {
  float* p;
  CK(hipMalloc(&p, n * 4));
  CK(hipMemset(p, 0xff, n * 4));  // NaN pattern: an unwritten element shows
  return p;
}
static float* upload(const std::vector<float>& v) {
  float* p;
  CK(hipMalloc(&p, v.size() * 4));
  CK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return p;
}
static double cmp(const char* name, const float* dptr, const std::vector<double>& ref, double* worst_rel) {
  std::vector<float> h(ref.size());
  CK(hipMemcpy(h.data(), dptr, h.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, sc = 1e-30;
  size_t bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    if (!(h[i] == h[i])) ++bad;
    md = fmax(md, fabs((double)h[i] - ref[i]));
    sc = fmax(sc, fabs(ref[i]));
  }
  printf("    %-10s max|gpu-cpu| %.2e  scale %.2e  rel %.1e%s\n", name, md, sc, md / sc, bad ? "  <-- NaN / unwritten elements" : "");
  if (bad) *worst_rel = 1.0;
  *worst_rel = fmax(*worst_rel, md / sc);
  return md;
}

static int run_case(int rows, int od, int ad, int H, int rows_global) {
  const int L = 2 * ad;
  printf("rows %d  obs %d  act %d  latent %d  hidden %d  rows_global %d\n", rows, od, ad, L, H, rows_global);
  Lin e0, e1, e2, d0, d1, d2;
  e0.init(H, od + ad); e1.init(H, H); e2.init(2 * L, H);
  d0.init(H, od + L); d1.init(H, H); d2.init(ad, H);
  const float max_action = 1.5f, beta = 0.5f;
  std::vector<float> obs((size_t)rows * od), act((size_t)rows * ad), eps((size_t)rows * L);
  for (auto& v : obs) v = 2 * rnd();
  for (auto& v : act) v = 2 * rnd() * max_action;
  for (auto& v : eps) v = 3 * rnd();
  // ---- CPU, double
  const double inv = 1.0 / (rows_global > 0 ? rows_global : rows);
  auto lin = [&](const Lin& l, const std::vector<double>& x, int n) {
    std::vector<double> y((size_t)n * l.out);
    for (int r = 0; r < n; ++r)
      for (int o = 0; o < l.out; ++o) {
        double s = l.b[o];
        for (int i = 0; i < l.in; ++i) s += x[(size_t)r * l.in + i] * l.W[(size_t)o * l.in + i];
        y[(size_t)r * l.out + o] = s;
      }
    return y;
  };
  auto relu = [](std::vector<double> v) { for (auto& x : v) x = x > 0 ? x : 0; return v; };
  auto bwd = [&](const Lin& l, const std::vector<double>& dz, int n) {  // dX = dZ W
    std::vector<double> dx((size_t)n * l.in, 0.0);
    for (int r = 0; r < n; ++r)
      for (int o = 0; o < l.out; ++o) {
        const double g = dz[(size_t)r * l.out + o];
        if (g != 0.0)
          for (int i = 0; i < l.in; ++i) dx[(size_t)r * l.in + i] += g * l.W[(size_t)o * l.in + i];
      }
    return dx;
  };
  std::vector<double> ex((size_t)rows * (od + ad));
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < od; ++c) ex[(size_t)r * (od + ad) + c] = obs[(size_t)r * od + c];
    for (int c = 0; c < ad; ++c) ex[(size_t)r * (od + ad) + od + c] = act[(size_t)r * ad + c];
  }
  auto eh0 = relu(lin(e0, ex, rows)), eh1 = relu(lin(e1, eh0, rows)), head = lin(e2, eh1, rows);
  std::vector<double> z((size_t)rows * L), dx_((size_t)rows * (od + L));
  for (int r = 0; r < rows; ++r)
    for (int j = 0; j < L; ++j) {
      const double ls = fmin(fmax(head[(size_t)r * 2 * L + L + j], -4.0), 15.0);
      z[(size_t)r * L + j] = head[(size_t)r * 2 * L + j] + exp(ls) * eps[(size_t)r * L + j];
    }
  for (int r = 0; r < rows; ++r) {
    for (int c = 0; c < od; ++c) dx_[(size_t)r * (od + L) + c] = obs[(size_t)r * od + c];
    for (int j = 0; j < L; ++j) dx_[(size_t)r * (od + L) + od + j] = z[(size_t)r * L + j];
  }
  auto dh0 = relu(lin(d0, dx_, rows)), dh1 = relu(lin(d1, dh0, rows)), upre = lin(d2, dh1, rows);
  std::vector<double> u((size_t)rows * ad), dz2((size_t)rows * ad);
  double l0 = 0, l1 = 0;
  for (size_t i = 0; i < u.size(); ++i) {
    const double t = tanh(upre[i]);
    u[i] = max_action * t;
    const double d = u[i] - act[i];
    l0 += d * d;
    dz2[i] = 2 * d * (inv / ad) * max_action * (1 - t * t);
  }
  for (int r = 0; r < rows; ++r)
    for (int j = 0; j < L; ++j) {
      const double m = head[(size_t)r * 2 * L + j], sd = exp(fmin(fmax(head[(size_t)r * 2 * L + L + j], -4.0), 15.0));
      l1 += -0.5 * (1 + log(sd * sd) - m * m - sd * sd);
    }
  const double stat = l0 * (inv / ad) + beta * (l1 * (inv / L));
  // ---- device
  osrl_mlp_t enc{}, dec{};
  enc.n_layers = dec.n_layers = 3; enc.n_nets = dec.n_nets = 1;
  int ed[4] = {od + ad, H, H, 2 * L}, dd[4] = {od + L, H, H, ad};
  for (int i = 0; i < 4; ++i) { enc.dims[i] = ed[i]; dec.dims[i] = dd[i]; }
  enc.acts[0] = enc.acts[1] = dec.acts[0] = dec.acts[1] = OSRL_ACT_RELU; enc.acts[2] = OSRL_ACT_ID; dec.acts[2] = OSRL_ACT_TANH;
  enc.out_scale = 1.f; dec.out_scale = max_action;
  Lin* el[3] = {&e0, &e1, &e2}; Lin* dl[3] = {&d0, &d1, &d2};
  for (int l = 0; l < 3; ++l) {
    enc.Wf[0][l] = el[l]->dF; enc.Wb[0][l] = el[l]->dB; enc.b[0][l] = el[l]->db;
    dec.Wf[0][l] = dl[l]->dF; dec.Wb[0][l] = dl[l]->dB; dec.b[0][l] = dl[l]->db;
  }
  osrl_vae_ns_t v{};
  v.enc = &enc; v.dec = &dec; v.rows = rows; v.od = od; v.ad = ad; v.L = L; v.rows_global = rows_global; v.beta = beta;
  v.obs = upload(obs); v.act = upload(act); v.eps = upload(eps);
  v.enc_acts.x = dalloc((size_t)rows * (od + ad)); v.enc_acts.h[0][0] = dalloc((size_t)rows * H);
  v.enc_acts.h[0][1] = dalloc((size_t)rows * H); v.enc_acts.h[0][2] = dalloc((size_t)rows * 2 * L);
  v.dec_acts.x = dalloc((size_t)rows * (od + L)); v.dec_acts.h[0][0] = dalloc((size_t)rows * H);
  v.dec_acts.h[0][1] = dalloc((size_t)rows * H); v.dec_acts.h[0][2] = dalloc((size_t)rows * ad);
  v.z = dalloc((size_t)rows * L);
  v.enc_g.dz[0][0] = dalloc((size_t)rows * H); v.enc_g.dz[0][1] = dalloc((size_t)rows * H); v.enc_g.dz[0][2] = dalloc((size_t)rows * 2 * L);
  v.dec_g.dz[0][0] = dalloc((size_t)rows * H); v.dec_g.dz[0][1] = dalloc((size_t)rows * H); v.dec_g.dz[0][2] = dalloc((size_t)rows * ad);
  v.P = dalloc((size_t)rows * H);
  v.slabs = dalloc((size_t)3 * (H / 80) * rows * 32);
  v.partials = dalloc(2 * ((rows + 47) / 48));
  CK(hipMalloc((void**)&v.counter, 4)); CK(hipMemset(v.counter, 0, 4));
  v.stat = dalloc(1);
  if (!osrl_vae_ns_supported(&v)) { printf("  not supported\n"); return 1; }
  int rc = osrl_vae_ns_forward(&v, nullptr);
  if (rc) { printf("  forward rc %d\n", rc); return 1; }
  rc = osrl_vae_ns_backward(&v, nullptr);
  if (rc) { printf("  backward rc %d\n", rc); return 1; }
  CK(hipDeviceSynchronize());
  // relu' decisions are taken from the DEVICE's activations (compared with the CPU's below): an element within an ulp of
  // the kink may fall on either side in fp32 and fp64, which is not what this harness is after
  auto dl_ = [&](const float* d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; };
  const auto g_eh0 = dl_(v.enc_acts.h[0][0], (size_t)rows * H), g_eh1 = dl_(v.enc_acts.h[0][1], (size_t)rows * H);
  const auto g_dh0 = dl_(v.dec_acts.h[0][0], (size_t)rows * H), g_dh1 = dl_(v.dec_acts.h[0][1], (size_t)rows * H);
  auto mask = [](std::vector<double> g, const std::vector<float>& h) { for (size_t i = 0; i < g.size(); ++i) g[i] = h[i] > 0 ? g[i] : 0; return g; };
  auto ddz1 = mask(bwd(d2, dz2, rows), g_dh1), ddz0 = mask(bwd(d1, ddz1, rows), g_dh0);
  auto dxd = bwd(d0, ddz0, rows);
  std::vector<double> edz2((size_t)rows * 2 * L);
  for (int r = 0; r < rows; ++r)
    for (int j = 0; j < L; ++j) {
      const double g = dxd[(size_t)r * (od + L) + od + j];
      const double m = head[(size_t)r * 2 * L + j], lsr = head[(size_t)r * 2 * L + L + j];
      const double sd = exp(fmin(fmax(lsr, -4.0), 15.0)), c = beta * inv / L;
      edz2[(size_t)r * 2 * L + j] = g + c * m;
      edz2[(size_t)r * 2 * L + L + j] = (lsr >= -4 && lsr <= 15) ? (g * eps[(size_t)r * L + j] + c * (sd - 1 / sd)) * sd : 0.0;
    }
  auto edz1 = mask(bwd(e2, edz2, rows), g_eh1), edz0 = mask(bwd(e1, edz1, rows), g_eh0);
  double worst = 0;
  cmp("enc.x", v.enc_acts.x, ex, &worst); cmp("enc.h0", v.enc_acts.h[0][0], eh0, &worst); cmp("enc.h1", v.enc_acts.h[0][1], eh1, &worst);
  cmp("enc.head", v.enc_acts.h[0][2], head, &worst); cmp("z", v.z, z, &worst); cmp("dec.x", v.dec_acts.x, dx_, &worst);
  cmp("dec.h0", v.dec_acts.h[0][0], dh0, &worst); cmp("dec.h1", v.dec_acts.h[0][1], dh1, &worst); cmp("dec.u", v.dec_acts.h[0][2], u, &worst);
  cmp("dec.dz2", v.dec_g.dz[0][2], dz2, &worst); cmp("dec.dz1", v.dec_g.dz[0][1], ddz1, &worst); cmp("dec.dz0", v.dec_g.dz[0][0], ddz0, &worst);
  cmp("enc.dz2", v.enc_g.dz[0][2], edz2, &worst); cmp("enc.dz1", v.enc_g.dz[0][1], edz1, &worst); cmp("enc.dz0", v.enc_g.dz[0][0], edz0, &worst);
  cmp("loss", v.stat, std::vector<double>{stat}, &worst);
  // second pass: the re-armed counter gives the same statistic again
  CK(hipMemset(v.stat, 0, 4));
  osrl_vae_ns_forward(&v, nullptr); osrl_vae_ns_backward(&v, nullptr);
  CK(hipDeviceSynchronize());
  cmp("loss (2nd)", v.stat, std::vector<double>{stat}, &worst);
  // from here on the launches read their descriptors from device memory, as inside a captured step (csrc/argmem.h)
  {
    static char* host = (char*)malloc(1 << 16);
    g_ar = osrl_argmem::Arena{host, nullptr, 1 << 16, 0, osrl_argmem::kRecord, 0, 0, 0};
    osrl_vae_ns_forward(&v, nullptr); osrl_vae_ns_backward(&v, nullptr);
    CK(hipDeviceSynchronize());
    char* dev;
    CK(hipMalloc((void**)&dev, g_ar.used));
    CK(hipMemcpy(dev, host, g_ar.used, hipMemcpyHostToDevice));
    g_ar.dev = dev;
    g_ar.mode = osrl_argmem::kReplay;
    CK(hipMemset(v.stat, 0, 4));
    osrl_vae_ns_forward(&v, nullptr); osrl_vae_ns_backward(&v, nullptr);
    CK(hipDeviceSynchronize());
    cmp("loss (arena)", v.stat, std::vector<double>{stat}, &worst);
    printf("    argument arena: %d blocks, %d hits, %d misses\n", g_ar.n_blocks, g_ar.n_hits, g_ar.n_misses);
  }
  // timing
  hipEvent_t t0, t1, t2;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
  float f_ms = 0, b_ms = 0;
  const int reps = 50;
  for (int i = 0; i < reps + 5; ++i) {
    CK(hipEventRecord(t0));
    osrl_vae_ns_forward(&v, nullptr);
    CK(hipEventRecord(t1));
    osrl_vae_ns_backward(&v, nullptr);
    CK(hipEventRecord(t2));
    CK(hipEventSynchronize(t2));
    float a, b;
    CK(hipEventElapsedTime(&a, t0, t1)); CK(hipEventElapsedTime(&b, t1, t2));
    if (i >= 5) { f_ms += a; b_ms += b; }
  }
  // back-to-back throughput of the five launches (what a replayed graph sees)
  CK(hipEventRecord(t0));
  for (int i = 0; i < reps; ++i) { osrl_vae_ns_forward(&v, nullptr); osrl_vae_ns_backward(&v, nullptr); }
  CK(hipEventRecord(t2)); CK(hipEventSynchronize(t2));
  float all_ms;
  CK(hipEventElapsedTime(&all_ms, t0, t2));
  // the same five launches as a replayed hipGraph (what the engine does; no host-side launch cost in the figure)
  hipStream_t cs;
  CK(hipStreamCreate(&cs));
  hipGraph_t gr;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
  osrl_vae_ns_forward(&v, cs);
  osrl_vae_ns_backward(&v, cs);
  CK(hipStreamEndCapture(cs, &gr));
  CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, cs));
  CK(hipStreamSynchronize(cs));
  CK(hipEventRecord(t0, cs));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, cs));
  CK(hipEventRecord(t2, cs));
  CK(hipEventSynchronize(t2));
  float g_ms;
  CK(hipEventElapsedTime(&g_ms, t0, t2));
  // forward only / backward only graphs
  float gf_ms = 0, gb_ms = 0;
  for (int which = 0; which < 2; ++which) {
    hipGraph_t g2; hipGraphExec_t e2;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
    if (which == 0) osrl_vae_ns_forward(&v, cs); else osrl_vae_ns_backward(&v, cs);
    CK(hipStreamEndCapture(cs, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(e2, cs));
    CK(hipStreamSynchronize(cs));
    CK(hipEventRecord(t0, cs));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(e2, cs));
    CK(hipEventRecord(t2, cs));
    CK(hipEventSynchronize(t2));
    CK(hipEventElapsedTime(which == 0 ? &gf_ms : &gb_ms, t0, t2));
  }
  {  // each launch alone (a one-kernel graph, replayed)
    NsArgs na;
    fill(&v, &na, true);
    const int Kpe = (od + ad + 15) & ~15;
    const char* names[5] = {"l0", "enc wide", "dec wide", "dec wide^T", "enc wide^T"};
    printf("  per launch:");
    for (int k = 0; k < 5; ++k) {
      hipGraph_t g2; hipGraphExec_t e2;
      CK(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
      const int nks = (od + L - 1) / 16 - od / 16 + 1;
      if (k == 0) {
        const int g0 = ((rows + 31) / 32) * na.col_groups;
        const size_t l0b = sizeof(float) * 32 * (Kpe + 4);
        if (Kpe <= 48) NS_LAUNCH(vae_ns_l0_kernel, 3), g0, 256, l0b, cs, na);
        else if (Kpe <= 80) NS_LAUNCH(vae_ns_l0_kernel, 5), g0, 256, l0b, cs, na);
        else NS_LAUNCH(vae_ns_l0_kernel, 8), g0, 256, l0b, cs, na);
      } else if (k == 1) {
        if (2 * L <= 16) NS_LAUNCH(vae_ns_fwd_enc_kernel, 1), wide_grid(na), 256, kWideLds, cs, na);
        else NS_LAUNCH(vae_ns_fwd_enc_kernel, 2), wide_grid(na), 256, kWideLds, cs, na);
      } else if (k == 2) {
        if (nks == 1) NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 1), wide_grid(na), 256, kWideLds, cs, na);
        else NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 2), wide_grid(na), 256, kWideLds, cs, na);
      } else if (k == 3) {
        NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_BWD, 1), wide_grid(na), 256, kWideLds, cs, na);
      } else {
        if (2 * L <= 16) NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 1), wide_grid(na), 256, kWideLds, cs, na);
        else NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 2), wide_grid(na), 256, kWideLds, cs, na);
      }
      CK(hipStreamEndCapture(cs, &g2));
      CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
      for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(e2, cs));
      CK(hipStreamSynchronize(cs));
      CK(hipEventRecord(t0, cs));
      for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(e2, cs));
      CK(hipEventRecord(t2, cs));
      CK(hipEventSynchronize(t2));
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t2));
      printf("  %s %.2f us", names[k], ms * 1e3 / reps);
    }
    printf("\n");
  }
  {  // phase stamps (wave 0 of every workgroup, 10 ns ticks) of the three generated-operand launches, one eager pass each
    const char* ph[11] = {"issue", "prologue+barrier", "h0 small", "h0 fix-up", "h0 main", "h1 small", "h1 fix-up", "h1 main",
                          "barrier+park+barrier", "reduce+store", "slab product"};
    NsArgs na;
    fill(&v, &na, true);
    const int nks = (od + L - 1) / 16 - od / 16 + 1;
    for (int k = 0; k < 3; ++k) {
      if (k == 0) { if (nks == 1) NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 1), wide_grid(na), 256, kWideLds, nullptr, na); else NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_FWD, 2), wide_grid(na), 256, kWideLds, nullptr, na); }
      if (k == 1) NS_LAUNCH(vae_ns_gen_kernel, MODE_DEC_BWD, 1), wide_grid(na), 256, kWideLds, nullptr, na);
      if (k == 2) { if (2 * L <= 16) NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 1), wide_grid(na), 256, kWideLds, nullptr, na); else NS_LAUNCH(vae_ns_gen_kernel, MODE_ENC_BWD, 2), wide_grid(na), 256, kWideLds, nullptr, na); }
      CK(hipDeviceSynchronize());
      static unsigned long long st[512][12];
      CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ns_stamp), sizeof(st)));
      const int nwg = wide_grid(na) < 512 ? wide_grid(na) : 512;
      double acc[11] = {0};
      int live = 0;
      for (int g = 0; g < nwg; ++g) {
        if (st[g][11] <= st[g][0]) continue;  // (an id beyond the last tile returns before its first stamp)
        ++live;
        for (int i = 0; i < 11; ++i) acc[i] += (double)(st[g][i + 1] - st[g][i]) * 0.01;
      }
      printf("  phases of %s (us, mean of %d workgroups):", k == 0 ? "dec wide" : k == 1 ? "dec wide^T" : "enc wide^T", live);
      double tot = 0;
      for (int i = 0; i < 11; ++i) { printf(" %s %.2f", ph[i], acc[i] / (live ? live : 1)); tot += acc[i] / (live ? live : 1); }
      printf("  total %.2f\n", tot);
      CK(hipMemset(st, 0, 0));
    }
  }
  printf("  replayed graph: forward + backward %.2f us  (forward alone %.2f, backward alone %.2f)\n", g_ms * 1e3 / reps,
         gf_ms * 1e3 / reps, gb_ms * 1e3 / reps);
  printf("  forward (3 launches) %.2f us   backward (2 launches) %.2f us   [eager, event-bracketed];  5 launches back to back %.2f us\n",
         f_ms * 1e3 / reps, b_ms * 1e3 / reps, all_ms * 1e3 / reps);
  g_ar.mode = osrl_argmem::kOff;
  printf("  %s (worst relative error %.1e)\n", worst < 2e-5 ? "PARITY OK" : "PARITY FAILED", worst);
  return worst < 2e-5 ? 0 : 1;
}

int main() {
  int bad = 0;
  bad += run_case(100, 7, 3, 80, 0);        // ragged rows, odd dims, one column group
  bad += run_case(64, 76, 2, 400, 0);       // cpq_wide
  bad += run_case(2048, 76, 2, 400, 0);     // C2
  bad += run_case(2048, 17, 6, 400, 16384); // C4 (rows_global = the 8-GPU job's)
  bad += run_case(4096, 33, 8, 400, 0);     // C3: latent 16 straddles two k-steps of the decoder's layer 0
  printf(bad ? "FAILED (%d cases)\n" : "ALL CASES OK\n", bad);
  return bad;
}
