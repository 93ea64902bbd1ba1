// Diagnostic (not part of the library): effective shader clock and dependent-load latency on the box,
// cold and right after a burst of work.  hipcc --offload-arch=gfx950 tools/clockprobe.hip -o tools/clockprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void spin(long long* out, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  float x = threadIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
__global__ void chase(const int* next, int n, long long* out) {
  int p = 0;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) p = next[p];
  long long c1 = clock64(), w1 = wall_clock64();
  out[0] = c1 - c0; out[1] = w1 - w0; out[2] = p;
}
__global__ void burn(float* y, int iters) {
  float x = threadIdx.x + blockIdx.x;
  for (int i = 0; i < iters; ++i) x = x * 1.0001f + 0.5f;
  y[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
  long long* out; hipMalloc(&out, 64); long long h[3];
  float* y; hipMalloc(&y, 1024 * 256 * 4);
  auto report = [&](const char* tag) {
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("%-28s shader cycles %lld, wall ticks(100MHz) %lld -> %.0f MHz, %.2f us\n", tag, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 100.0);
  };
  spin<<<1, 64>>>(out, 20000); hipDeviceSynchronize(); report("spin cold");
  for (int i = 0; i < 200; ++i) burn<<<1024, 256>>>(y, 20000);
  spin<<<1, 64>>>(out, 20000); hipDeviceSynchronize(); report("spin after 200 burns");
  for (int sz : {1 << 12, 1 << 20, 1 << 26}) {
    std::vector<int> perm(sz), nxt(sz);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    for (int i = 0; i < sz; ++i) nxt[perm[i]] = perm[(i + 1) % sz];
    int* d; hipMalloc(&d, (size_t)sz * 4); hipMemcpy(d, nxt.data(), (size_t)sz * 4, hipMemcpyHostToDevice);
    chase<<<1, 1>>>(d, 2000, out); hipDeviceSynchronize();
    chase<<<1, 1>>>(d, 2000, out); hipDeviceSynchronize();
    hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    printf("chase %9d ints: %.0f shader cycles / load, %.1f ns / load\n", sz, h[0] / 2000.0, 10.0 * h[1] / 2000.0);
    hipFree(d);
  }
  // launch cadence of back-to-back tiny kernels
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 1000; ++i) spin<<<1, 64>>>(out, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("1000 tiny launches: %.2f us each\n", ms);
  return 0;
}
