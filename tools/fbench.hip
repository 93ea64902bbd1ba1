// Diagnostic micro-benchmark (not part of the library): the fused training step (ultr_dnn_train_fused with a parked
// DLA loss request) at the headline shape through the C ABI; per-kernel HIP-event durations (ultr_profile_*) and,
// when the library was built with -DULTR_STAMPS, the in-kernel phase stamps of train_chain_kernel.
//   tools/build_fbench.sh ; tools/fbench [B] [F]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "ultr_hip.h"

extern "C" int ultr_debug_tc_stamps(long long* host, int n) __attribute__((weak));
extern "C" int ultr_debug_tc_wstamps(long long* host, int n) __attribute__((weak));
extern "C" int ultr_debug_wg_stamps(long long* host, int n) __attribute__((weak));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define UK(x) do { int r_ = (x); if (r_ != 0) { printf("ultr error %d (%s) at %d\n", r_, ultr_last_error(), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, L = 10, F = argc > 2 ? atoi(argv[2]) : 136;
  const int iters = argc > 3 ? atoi(argv[3]) : 200;
  const long long M = (long long)B * L;
  const int n_rows = 54000;
  UltrDnnDesc d;
  memset(&d, 0, sizeof(d));
  d.n_layers = 4; d.in_features = F;
  d.sizes[0] = 512; d.sizes[1] = 256; d.sizes[2] = 128; d.sizes[3] = 1;
  d.activation = ULTR_ACT_ELU; d.norm = ULTR_NORM_LAYER; d.ln_eps = 1e-3f;
  long long off = 0;
  auto take = [&](long long n) { long long o = off; off += (n + 63) & ~63ll; return o; };
  int cur = F;
  for (int j = 0; j < 4; ++j) {
    const int N = d.sizes[j], ldn = (N + 3) & ~3;
    d.off_gamma[j] = take(cur); d.off_beta[j] = take(cur);
    d.off_W[j] = take(j == 3 ? cur : (long long)cur * ldn); d.off_b[j] = take(N);
    cur = N;
  }
  d.n_params = off;
  std::vector<float> hp(off), ht((size_t)n_rows * F);
  srand(1);
  for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  cur = F;
  for (int j = 0; j < 4; ++j) { for (int k = 0; k < cur; ++k) hp[d.off_gamma[j] + k] = 1.f; cur = d.sizes[j]; }
  for (auto& v : ht) v = rand() / (float)RAND_MAX;
  std::vector<int> hd(M);
  for (auto& v : hd) v = rand() % (n_rows + 1);
  std::vector<float> hc(M);
  for (auto& v : hc) v = (rand() % 4) == 0 ? 1.f : 0.f;
  std::vector<float> hprop(L + 1);
  for (auto& v : hprop) v = rand() / (float)RAND_MAX - 0.5f;
  float *params, *table, *stats, *scores, *dsc, *grads, *clicks, *prop, *pgrads, *lw, *lstats, *lws, *sumsq; int* docids; void* ws;
  CK(hipMalloc(&params, off * 4)); CK(hipMalloc(&table, ht.size() * 4)); CK(hipMalloc(&stats, (n_rows + 1) * 8));
  CK(hipMalloc(&scores, M * 4)); CK(hipMalloc(&dsc, M * 4)); CK(hipMalloc(&grads, off * 4)); CK(hipMalloc(&docids, M * 4));
  CK(hipMalloc(&clicks, M * 4)); CK(hipMalloc(&prop, (L + 1) * 4)); CK(hipMalloc(&pgrads, (L + 1) * 4)); CK(hipMalloc(&lw, B * 4));
  CK(hipMalloc(&lstats, 1024 * 4));
  const long long lwsn = ultr_loss_workspace_floats(B, L);
  CK(hipMalloc(&lws, lwsn * 4)); CK(hipMemset(lws, 0, lwsn * 4));
  const int nss = ultr_dnn_sumsq_count(&d);
  CK(hipMalloc(&sumsq, nss * 4));
  const long long wsb = ultr_dnn_fused_workspace_bytes(&d, B, L);
  if (wsb <= 0) { printf("shape not supported by the fused step\n"); return 1; }
  CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  CK(hipMemset(grads, 0, off * 4));
  CK(hipMemcpy(params, hp.data(), off * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(docids, hd.data(), M * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(clicks, hc.data(), M * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(prop, hprop.data(), (L + 1) * 4, hipMemcpyHostToDevice));
  UK(ultr_ln_row_stats(table, n_rows, F, F, 1e-3f, stats, nullptr));
  int wt_fresh = 0;  // the parameters never change here: the copies stay current after the first call
  auto step = [&]() {
    UK(ultr_loss_dla_deferred(scores, clicks, B, L, prop, ULTR_LOSS_SOFTMAX, ULTR_PROB_SOFTMAX, -1.f, dsc, pgrads, lw, lstats, lws, nullptr));
    UK(ultr_dnn_train_fused(&d, params, table, stats, n_rows, F, docids, B, L, ws, wt_fresh, scores, grads, sumsq, nullptr)); wt_fresh = 1;
  };
  for (int it = 0; it < 20; ++it) step();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) step();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("B=%d M=%lld F=%d: fused forward+backward %.2f us per call\n", B, M, F, 1e3 * ms / iters);
  ultr_profile_enable(1);
  for (int it = 0; it < 50; ++it) step();
  CK(hipDeviceSynchronize());
  static char buf[1 << 16];
  ultr_profile_report(buf, sizeof(buf));
  ultr_profile_enable(0);
  printf("%s\n", buf);
  if (ultr_debug_tc_stamps) {
    step();
    const int NS = 32, NB = std::min(1024, B);
    std::vector<long long> st((size_t)1024 * NS);
    ultr_debug_tc_stamps(st.data(), (int)st.size());
    long long t0 = -1;
    for (int b = 0; b < NB; ++b) if (st[(size_t)b * NS] && (t0 < 0 || st[(size_t)b * NS] < t0)) t0 = st[(size_t)b * NS];
    {
      double cyc = 0, wall = 0; int c = 0;
      for (int b = 0; b < NB; ++b) { const long long* s = &st[(size_t)b * NS]; if (s[28] && s[29]) { cyc += (double)(s[29] - s[28]); wall += (double)(s[21] - s[20]) * 10.0; c++; } }
      if (c) printf("   last dgrad GEMM: %.0f shader cycles in %.0f ns per workgroup = %.2f GHz\n", cyc / c, wall / c, cyc / wall);
    }
    if (ultr_debug_tc_wstamps) {
      // per-wave stamps: forward layer j: 3j start of GEMM, 3j+1 GEMM + epilogue done, 3j+2 next input stage done;
      // 8 output stage, 9 output backward, 10 / 11 dgrad_2 + row pass, 12 / 13 dgrad_1 + row pass
      std::vector<long long> ws((size_t)1024 * 128);
      ultr_debug_tc_wstamps(ws.data(), (int)ws.size());
      for (int i = 0; i < 14; ++i) {
        printf("   wstamp %2d:", i);
        for (int w = 0; w < 8; ++w) {
          double a = 0; int c = 0;
          for (int b = 0; b < NB; ++b) { const long long v = ws[((size_t)b * 8 + w) * 16 + i]; if (v) { a += (v - t0) / 100.0; c++; } }
          printf(" %6.2f", c ? a / c : 0.0);
        }
        printf("\n");
      }
    }
    double prev = 0;
    for (int i = 0; i < NS; ++i) {
      if (i == 28 || i == 29) continue;
      double sum = 0, mx = 0, mn = 1e18; int c = 0;
      for (int b = 0; b < NB; ++b) { const long long* s = &st[(size_t)b * NS]; if (s[0] && s[i]) { double v = (s[i] - t0) / 100.0; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); c++; } }
      if (c) { printf("   stamp %2d: min %6.2f mean %6.2f max %6.2f us  (+%.2f)\n", i, mn, sum / c, mx, sum / c - prev); prev = sum / c; }
    }
  }
  if (ultr_debug_wg_stamps) {
    std::vector<long long> st((size_t)1024 * 8);
    ultr_debug_wg_stamps(st.data(), (int)st.size());
    long long t0 = -1;
    for (int b = 0; b < 1024; ++b) if (st[(size_t)b * 8] && (t0 < 0 || st[(size_t)b * 8] < t0)) t0 = st[(size_t)b * 8];
    for (int half = 0; half < 2; ++half)
      for (int i = 0; i < 5; ++i) {
        double sum = 0, mx = 0, mn = 1e18; int c = 0;
        for (int b = half * 272; b < (half ? 1024 : 272); ++b) { const long long* s = &st[(size_t)b * 8]; if (s[0] && s[i]) { double v = (s[i] - t0) / 100.0; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); c++; } }
        if (c) printf("   wgrad blocks %s stamp %d: min %6.2f mean %6.2f max %6.2f us (n=%d)\n", half ? ">=272" : "<272 ", i, mn, sum / c, mx, c);
      }
  }
  return 0;
}
