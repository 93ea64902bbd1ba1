// Diagnostic (not part of the library): does a small, latency-bound kernel get slower when the kernels before it
// touched a lot of memory (cold TLB / cold Infinity Cache)?   hipcc --offload-arch=gfx950 -O2 tools/tlbprobe.hip -o tools/tlbprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void thrash(float* buf, size_t n, size_t stride) {
  for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * stride; i < n; i += (size_t)gridDim.x * blockDim.x * stride) buf[i] += 1.f;
}
// 16 blocks; each thread does a chain of 3 dependent loads through small tables spread over `span` bytes
__global__ void __launch_bounds__(256) small(const int* idx, const float* a, const float* b, float* out, int n) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int i = idx[t % n];
  const float x = a[i];
  const float y = b[(int)x % n];
  out[t] = x + y;
}
int main() {
  const size_t big = (size_t)1 << 30;  // 1 GiB
  float* buf; hipMalloc(&buf, big); hipMemset(buf, 0, big);
  const int n = 1 << 16;
  int* idx; float *a, *b, *out;
  // the small kernel's operands live in four separate 64 MB allocations (four different 2 MB pages at least)
  hipMalloc(&idx, 64 << 20); hipMalloc(&a, 64 << 20); hipMalloc(&b, 64 << 20); hipMalloc(&out, 64 << 20);
  hipMemset(idx, 0, 64 << 20); hipMemset(a, 0, 64 << 20); hipMemset(b, 0, 64 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 30; ++rep) {
      if (mode == 1) hipLaunchKernelGGL(thrash, dim3(1024), dim3(256), 0, 0, buf, big / 4, (size_t)1024);          // one touch per 4 KB over 1 GiB
      if (mode == 2) hipLaunchKernelGGL(thrash, dim3(1024), dim3(256), 0, 0, buf, (size_t)(64 << 20) / 4, (size_t)16);  // 64 MB, one touch per 64 B
      if (mode == 3) hipLaunchKernelGGL(thrash, dim3(1024), dim3(256), 0, 0, buf, big / 4, (size_t)(512 * 1024));  // one touch per 2 MB over 1 GiB
      hipEventRecord(e0);
      hipLaunchKernelGGL(small, dim3(16), dim3(256), 0, 0, idx, a, b, out, n);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; sum += ms;
    }
    printf("%-44s small kernel: min %.2f us  mean %.2f us\n", mode == 0 ? "nothing before" : mode == 1 ? "after 1 GiB touched every 4 KB" : mode == 2 ? "after 64 MB streamed" : "after 1 GiB touched every 2 MB", best * 1e3, sum / 30 * 1e3);
  }
  return 0;
}
