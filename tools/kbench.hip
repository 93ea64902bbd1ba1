// Diagnostic micro-benchmark (not part of the library): times ultr_dnn_forward / ultr_dnn_backward at the
// headline shape through the C ABI and, when the library was built with -DULTR_STAMPS, prints the
// in-kernel phase stamps (100 MHz wall clock) per layer.
//   hipcc --offload-arch=gfx950 -O2 tools/kbench.hip -Iinclude -Lultra_amd/csrc -lultr_hip -o tools/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "ultr_hip.h"

extern "C" int ultr_debug_stamps(long long* host, int n) __attribute__((weak));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define UK(x) do { int r_ = (x); if (r_ != 0) { printf("ultr error %d (%s) at %d\n", r_, ultr_last_error(), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, L = 10, F = argc > 2 ? atoi(argv[2]) : 136;
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
  std::vector<float> hds(M);
  for (auto& v : hds) v = rand() / (float)RAND_MAX - 0.5f;
  float *params, *table, *stats, *scores, *dsc, *grads; int* docids; void* ws;
  CK(hipMalloc(&params, off * 4)); CK(hipMalloc(&table, ht.size() * 4)); CK(hipMalloc(&stats, (n_rows + 1) * 8));
  CK(hipMalloc(&scores, M * 4)); CK(hipMalloc(&dsc, M * 4)); CK(hipMalloc(&grads, off * 4)); CK(hipMalloc(&docids, M * 4));
  const long long wsb = ultr_dnn_workspace_bytes(&d, M);
  CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
  CK(hipMemcpy(params, hp.data(), off * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(docids, hd.data(), M * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsc, hds.data(), M * 4, hipMemcpyHostToDevice));
  UK(ultr_ln_row_stats(table, n_rows, F, F, 1e-3f, stats, nullptr));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int iters = 200;
  for (int it = 0; it < 20; ++it) {
    UK(ultr_dnn_forward(&d, params, table, stats, n_rows, F, docids, M, ws, scores, nullptr));
    UK(ultr_dnn_backward(&d, params, table, stats, n_rows, F, docids, M, ws, dsc, grads, nullptr));
  }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) UK(ultr_dnn_forward(&d, params, table, stats, n_rows, F, docids, M, ws, scores, nullptr));
  CK(hipEventRecord(e1));
  for (int it = 0; it < iters; ++it) UK(ultr_dnn_backward(&d, params, table, stats, n_rows, F, docids, M, ws, dsc, grads, nullptr));
  CK(hipEventRecord(e2));
  CK(hipEventSynchronize(e2));
  float f_ms, b_ms; CK(hipEventElapsedTime(&f_ms, e0, e1)); CK(hipEventElapsedTime(&b_ms, e1, e2));
  printf("B=%d M=%lld F=%d: forward %.2f us, backward %.2f us per call\n", B, M, F, 1e3 * f_ms / iters, 1e3 * b_ms / iters);
  if (ultr_debug_stamps) {
    const int NS = 8, NB = 1024, NK = 16;
    std::vector<long long> st((size_t)NK * NB * NS);
    ultr_debug_stamps(st.data(), (int)st.size());
    for (int k = 0; k < NK; ++k) {
      long long t0 = -1, tend = 0; int nb = 0;
      for (int b = 0; b < NB; ++b) { const long long* s = &st[((size_t)k * NB + b) * NS]; if (s[0]) { nb++; if (t0 < 0 || s[0] < t0) t0 = s[0]; for (int i = 0; i < NS; ++i) tend = std::max(tend, s[i]); } }
      if (!nb) continue;
      printf("kernel slot %d: %d blocks, first start -> last stamp %.2f us\n", k, nb, (tend - t0) / 100.0);
      // mean and max of each stamp relative to the kernel's first start, and per-block phase lengths
      for (int i = 0; i < NS; ++i) {
        double sum = 0, mx = 0, mn = 1e18; int c = 0;
        for (int b = 0; b < NB; ++b) { const long long* s = &st[((size_t)k * NB + b) * NS]; if (s[0] && s[i]) { double v = (s[i] - t0) / 100.0; sum += v; mx = std::max(mx, v); mn = std::min(mn, v); c++; } }
        if (c) printf("   stamp %d: min %.2f mean %.2f max %.2f us (n=%d)\n", i, mn, sum / c, mx, c);
      }
    }
  }
  return 0;
}
