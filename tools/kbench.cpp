// kbench -- per-layer conv benchmark + on-device cross-check through the C ABI (no Python).
//
//   tools/kbench [filter] [--algo A] [--iters N] [--batch B] [--nocheck]
//
// For every conv of the 256x256 TwinGAN stage (E/D skeleton + G with UNet concat, SURVEY.md
// Appendix A) runs forward / backward-data / backward-weight with the MFMA kernels, checks each
// against the direct kernels (validated against the oracle by tests/test_gpu_ops.py) at a reduced
// batch, and times the full batch with HIP events.  Build: make -C twingan_amd/csrc kbench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../include/twingan_hip.h"

#define HC(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)
#define TC(x)                                                               \
  do {                                                                      \
    int r_ = (x);                                                           \
    if (r_) {                                                               \
      fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, tg_last_error()); \
      exit(3);                                                              \
    }                                                                       \
  } while (0)

struct Case {
  const char* name;
  int hw, cin, cout, k;
};

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t rng_state = 12345;
static float frand() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f;
}

static void* dev_bf16_random(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf(frand() * scale);
  void* d;
  HC(hipMalloc(&d, n * 2 + 64));
  HC(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static float* dev_f32_random(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = frand() * scale;
  float* d;
  HC(hipMalloc(&d, n * 4 + 64));
  HC(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
static double rel_l2_bf16(const void* a, const void* b, size_t n) {
  std::vector<uint16_t> ha(n), hb(n);
  HC(hipMemcpy(ha.data(), a, n * 2, hipMemcpyDeviceToHost));
  HC(hipMemcpy(hb.data(), b, n * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) {
    double x = bf2f(ha[i]), y = bf2f(hb[i]);
    num += (x - y) * (x - y);
    den += y * y;
  }
  return sqrt(num / (den + 1e-30));
}
static double rel_l2_f32(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  HC(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  HC(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) {
    double x = ha[i], y = hb[i];
    num += (x - y) * (x - y);
    den += y * y;
  }
  return sqrt(num / (den + 1e-30));
}

static TgConvDesc mk(int n, int hw, int cin, int cout, int k, int algo, int epi) {
  TgConvDesc d;
  memset(&d, 0, sizeof(d));
  d.n = n; d.hin = d.win = hw; d.cin = cin;
  d.hout = d.wout = hw; d.cout = cout;
  d.kh = d.kw = k;
  d.pad_t = d.pad_l = (k - 1) / 2;
  d.dtype = TG_BF16;
  d.algo = algo;
  d.epilogue = epi;
  d.lrelu_alpha = 0.2f;
  return d;
}

template <typename F>
static float time_us(F f, int iters) {
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));
  f();
  f();
  HC(hipDeviceSynchronize());
  HC(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  HC(hipEventRecord(e1, 0));
  HC(hipEventSynchronize(e1));
  float ms;
  HC(hipEventElapsedTime(&ms, e0, e1));
  HC(hipEventDestroy(e0));
  HC(hipEventDestroy(e1));
  return 1e3f * ms / iters;
}

int main(int argc, char** argv) {
  const char* filter = nullptr;
  const char* only_op = nullptr;      // --op fwd|dgrad|wgrad: time just that op (PMC runs)
  int algo = TG_ALGO_MFMA, iters = 20, batch = 16, check = 1;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--algo")) algo = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--batch")) batch = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--nocheck")) check = 0;
    else if (!strcmp(argv[i], "--op")) only_op = argv[++i];
    else filter = argv[i];
  }
  std::vector<Case> cases = {
      {"E256a", 256, 16, 16, 3},  {"E256b", 256, 16, 32, 3},  {"E128a", 128, 32, 32, 3},  {"E128b", 128, 32, 64, 3},
      {"E64a", 64, 64, 64, 3},    {"E64b", 64, 64, 128, 3},   {"E32a", 32, 128, 128, 3},  {"E32b", 32, 128, 256, 3},
      {"E16", 16, 256, 256, 3},   {"E8", 8, 256, 256, 3},     {"D4", 4, 264, 256, 3},     {"G4", 4, 256, 256, 3},
      {"G8a", 8, 512, 256, 3},    {"G16a", 16, 512, 256, 3},  {"G32a", 32, 512, 128, 3},  {"G32b", 32, 128, 128, 3},
      {"G64a", 64, 256, 64, 3},   {"P1x1", 4, 256, 256, 1},   {"G128a", 128, 128, 32, 3}, {"G256a", 256, 64, 16, 3},
  };
  printf("%-7s %-5s %4s %4s>%-4s | %9s %8s %8s %9s\n", "case", "op", "hw", "cin", "cout", "us", "GB/s", "TF/s", "relL2");
  for (const Case& c : cases) {
    if (filter && !strstr(c.name, filter)) continue;
    const size_t px = (size_t)batch * c.hw * c.hw;
    void* x = dev_bf16_random(px * c.cin, 1.0f);
    void* gy = dev_bf16_random(px * c.cout, 1.0f);
    float* w = dev_f32_random((size_t)c.k * c.k * c.cin * c.cout, sqrtf(2.0f / (c.k * c.k * c.cin)));
    float* bias = dev_f32_random(c.cout, 0.1f);
    void *y, *y2, *gx, *gx2;
    HC(hipMalloc(&y, px * c.cout * 2));
    HC(hipMalloc(&y2, px * c.cout * 2));
    HC(hipMalloc(&gx, px * c.cin * 2));
    HC(hipMalloc(&gx2, px * c.cin * 2));
    const size_t nw = (size_t)c.k * c.k * c.cin * c.cout;
    float *gw, *gw2;
    HC(hipMalloc(&gw, nw * 4));
    HC(hipMalloc(&gw2, nw * 4));
    TgConvDesc d = mk(batch, c.hw, c.cin, c.cout, c.k, algo, TG_EPI_BIAS | TG_EPI_LRELU);
    TgConvDesc d0 = d;
    d0.epilogue = 0;
    void *p0, *p1;
    HC(hipMalloc(&p0, tg_conv2d_pack_elems(&d, 0) * 2 + 64));
    HC(hipMalloc(&p1, tg_conv2d_pack_elems(&d, 1) * 2 + 64));
    TC(tg_conv2d_pack_weights(&d, w, 0, p0, nullptr));
    TC(tg_conv2d_pack_weights(&d, w, 1, p1, nullptr));
    const size_t wsb = tg_conv2d_bwd_weight_workspace(&d);
    void* ws = nullptr;
    if (wsb) HC(hipMalloc(&ws, wsb));
    // ---- correctness at batch nchk against the direct kernels
    double e_f = -1, e_d = -1, e_w = -1;
    if (check) {
      const int nchk = batch < 8 ? batch : 8;      // enough tiles to reach the multi-tile (weight-resident) variants
      TgConvDesc dm = mk(nchk, c.hw, c.cin, c.cout, c.k, algo, TG_EPI_BIAS | TG_EPI_LRELU);
      TgConvDesc dr = mk(nchk, c.hw, c.cin, c.cout, c.k, TG_ALGO_DIRECT, TG_EPI_BIAS | TG_EPI_LRELU);
      const size_t pc = (size_t)nchk * c.hw * c.hw;
      TC(tg_conv2d_fwd(&dm, x, p0, bias, y, nullptr));
      TC(tg_conv2d_fwd(&dr, x, w, bias, y2, nullptr));
      e_f = rel_l2_bf16(y, y2, pc * c.cout);
      TC(tg_conv2d_bwd_data(&dm, gy, p1, gx, nullptr));
      TC(tg_conv2d_bwd_data(&dr, gy, w, gx2, nullptr));
      e_d = rel_l2_bf16(gx, gx2, pc * c.cin);
      void* wsc = nullptr;
      const size_t wc = tg_conv2d_bwd_weight_workspace(&dm);
      if (wc) HC(hipMalloc(&wsc, wc));
      TC(tg_conv2d_bwd_weight(&dm, x, gy, gw, 0, wsc, wc, nullptr));
      TC(tg_conv2d_bwd_weight(&dr, x, gy, gw2, 0, nullptr, 0, nullptr));
      e_w = rel_l2_f32(gw, gw2, nw);
      if (wsc) HC(hipFree(wsc));
    }
    // ---- timing at the full batch
    const double flops = 2.0 * px * c.cout * c.k * c.k * c.cin;
    const double bytes = 2.0 * px * (c.cin + c.cout) + 2.0 * nw;
    float t;
    if (!only_op || !strcmp(only_op, "fwd")) {
    t = time_us([&] { TC(tg_conv2d_fwd(&d, x, p0, bias, y, nullptr)); }, iters);
    printf("%-7s %-5s %4d %4d>%-4d | %9.1f %8.0f %8.1f %9.2e\n", c.name, "fwd", c.hw, c.cin, c.cout, t, bytes / t * 1e-3,
           flops / t * 1e-6, e_f);
    }
    if (!only_op || !strcmp(only_op, "dgrad")) {
    t = time_us([&] { TC(tg_conv2d_bwd_data(&d0, gy, p1, gx, nullptr)); }, iters);
    printf("%-7s %-5s %4d %4d>%-4d | %9.1f %8.0f %8.1f %9.2e\n", c.name, "dgrad", c.hw, c.cin, c.cout, t, bytes / t * 1e-3,
           flops / t * 1e-6, e_d);
    }
    if (only_op && !strcmp(only_op, "mdgrad")) {      // backward-data with the producer's LeakyReLU mask in the epilogue
    t = time_us([&] { TC(tg_conv2d_bwd_data_masked(&d0, gy, p1, x, gx, nullptr)); }, iters);
    printf("%-7s %-5s %4d %4d>%-4d | %9.1f %8.0f %8.1f %9.2e\n", c.name, "mdgrd", c.hw, c.cin, c.cout, t,
           (bytes + 2.0 * px * c.cin) / t * 1e-3, flops / t * 1e-6, -1.0);
    }
    if (only_op && !strcmp(only_op, "stats")) {      // conv + statistics pass  vs  conv with the statistics epilogue
      const int chunks = tg_conv2d_fwd_stats_chunks(&d0);
      const int nch = tg_norm_chunks(batch, c.hw, c.hw);
      float* part;
      HC(hipMalloc(&part, (size_t)batch * (chunks > nch ? chunks : nch) * 2 * c.cout * 4 + 64));
      const float tp = time_us([&] { TC(tg_conv2d_fwd(&d0, x, p0, nullptr, y, nullptr)); }, iters);
      const float ts = time_us([&] {
        TC(tg_conv2d_fwd(&d0, x, p0, nullptr, y, nullptr));
        TC(tg_instance_norm_partials(y, part, batch, c.hw, c.hw, c.cout, TG_BF16, nullptr));
      }, iters);
      float tf = -1.f;
      double err = -1;
      if (chunks > 0) {
        tf = time_us([&] { TC(tg_conv2d_fwd_stats(&d0, x, p0, y2, part, chunks, nullptr)); }, iters);
        HC(hipDeviceSynchronize());
        std::vector<uint16_t> hy((size_t)px * c.cout), hy2((size_t)px * c.cout);
        std::vector<float> hp((size_t)batch * chunks * 2 * c.cout);
        HC(hipMemcpy(hy.data(), y, hy.size() * 2, hipMemcpyDeviceToHost));
        HC(hipMemcpy(hy2.data(), y2, hy2.size() * 2, hipMemcpyDeviceToHost));
        HC(hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < hy.size(); ++i) diff += hy[i] != hy2[i];
        err = 0;
        const size_t hw2 = (size_t)c.hw * c.hw;
        for (int img = 0; img < batch; ++img)
          for (int ch = 0; ch < c.cout; ++ch) {
            double s1 = 0, s2 = 0, q1 = 0, q2 = 0;
            for (size_t p = 0; p < hw2; ++p) {
              const double v = bf2f(hy2[((size_t)img * hw2 + p) * c.cout + ch]);
              s1 += v;
              s2 += v * v;
            }
            for (int k = 0; k < chunks; ++k) {
              q1 += hp[(((size_t)img * chunks + k) * 2 + 0) * c.cout + ch];
              q2 += hp[(((size_t)img * chunks + k) * 2 + 1) * c.cout + ch];
            }
            const double e1 = fabs(q1 - s1) / (sqrt(s2 * hw2) + 1e-30), e2 = fabs(q2 - s2) / (s2 + 1e-30);
            if (e1 > err) err = e1;
            if (e2 > err) err = e2;
          }
        if (diff) err = 1e9 + diff;      // the statistics variant must write the very same tensor
      }
      printf("%-7s %-5s %4d %4d>%-4d | conv %7.1f  conv+stats-pass %7.1f  conv-with-stats %7.1f us (chunks %d)  err %.2e\n", c.name,
             "stats", c.hw, c.cin, c.cout, tp, ts, tf, chunks, err);
      HC(hipFree(part));
    }
    if (!only_op || !strcmp(only_op, "wgrad")) {
    t = time_us([&] { TC(tg_conv2d_bwd_weight(&d0, x, gy, gw, 0, ws, wsb, nullptr)); }, iters);
    printf("%-7s %-5s %4d %4d>%-4d | %9.1f %8.0f %8.1f %9.2e\n", c.name, "wgrad", c.hw, c.cin, c.cout, t, bytes / t * 1e-3,
           flops / t * 1e-6, e_w);
    }
    if (only_op && !strcmp(only_op, "wgradpar")) {
      // K independent filter gradients (own outputs, own workspaces) of this layer: one after the other on ONE stream vs spread
      // over K streams -- the second is what a merged multi-job launch could at best approach (ramps and tails overlap)
      const int K = 8;
      std::vector<float*> gws(K);
      std::vector<void*> wss(K);
      std::vector<hipStream_t> st(K);
      for (int i = 0; i < K; ++i) {
        HC(hipMalloc(&gws[i], nw * 4));
        wss[i] = nullptr;
        if (wsb) HC(hipMalloc(&wss[i], wsb));
        HC(hipStreamCreate(&st[i]));
      }
      const float t1 = time_us([&] { for (int i = 0; i < K; ++i) TC(tg_conv2d_bwd_weight(&d0, x, gy, gws[i], 0, wss[i], wsb, nullptr)); }, iters);
      hipEvent_t e0, e1;
      HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
      float tot = 0.f;
      for (int it = 0; it < iters + 2; ++it) {
        HC(hipDeviceSynchronize());
        HC(hipEventRecord(e0, st[0]));
        for (int i = 1; i < K; ++i) HC(hipStreamWaitEvent(st[i], e0, 0));
        for (int i = 0; i < K; ++i) TC(tg_conv2d_bwd_weight(&d0, x, gy, gws[i], 0, wss[i], wsb, (void*)st[i]));
        for (int i = 1; i < K; ++i) { hipEvent_t ej; HC(hipEventCreate(&ej)); HC(hipEventRecord(ej, st[i])); HC(hipStreamWaitEvent(st[0], ej, 0)); HC(hipEventDestroy(ej)); }
        HC(hipEventRecord(e1, st[0]));
        HC(hipEventSynchronize(e1));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) tot += ms;
      }
      printf("%-7s %-8s %4d %4d>%-4d n%-3d | %d launches on one stream %8.1f us, on %d streams %8.1f us\n", c.name, "wgradpar", c.hw, c.cin,
             c.cout, batch, K, t1, K, tot / iters * 1e3);
      for (int i = 0; i < K; ++i) { HC(hipFree(gws[i])); if (wss[i]) HC(hipFree(wss[i])); HC(hipStreamDestroy(st[i])); }
    }
    if (only_op && !strcmp(only_op, "wgradb")) {      // the filter gradient with the fused bias gradient (timing only)
      t = time_us([&] { TC(tg_conv2d_bwd_weight_bias(&d0, x, gy, gw, bias, 0, ws, wsb, nullptr)); }, iters);
      printf("%-7s %-5s %4d %4d>%-4d | %9.1f %8.0f %8.1f\n", c.name, "wgradb", c.hw, c.cin, c.cout, t, bytes / t * 1e-3,
             flops / t * 1e-6);
    }
    fflush(stdout);
    HC(hipFree(x)); HC(hipFree(gy)); HC(hipFree(w)); HC(hipFree(bias)); HC(hipFree(y)); HC(hipFree(y2));
    HC(hipFree(gx)); HC(hipFree(gx2)); HC(hipFree(gw)); HC(hipFree(gw2)); HC(hipFree(p0)); HC(hipFree(p1));
    if (ws) HC(hipFree(ws));
  }
  return 0;
}
