// Diagnostic (not part of the library): cost of executing straight-line code once (instruction-cache cold)
// versus the same work as a compact loop.   hipcc --offload-arch=gfx950 -O2 tools/icprobe.hip -o tools/icprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

template <int N>
__global__ void straight(float* out, long long* stamps, float a, float b) {
  const long long t0 = wall_clock64();
  float x = threadIdx.x, y = blockIdx.x;
#pragma unroll
  for (int i = 0; i < N; ++i) { x = __builtin_fmaf(x, a, y); y = __builtin_fmaf(y, b, x); }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
  if (x == 1234.5f) out[0] = y;
}
template <int N>
__global__ void looped(float* out, long long* stamps, float a, float b, int reps) {
  const long long t0 = wall_clock64();
  float x = threadIdx.x, y = blockIdx.x;
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < N; ++i) { x = __builtin_fmaf(x, a, y); y = __builtin_fmaf(y, b, x); }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
  if (x == 1234.5f) out[0] = y;
}

int main() {
  float* out; long long* st; hipMalloc(&out, 64); hipMalloc(&st, 4096 * 16);
  std::vector<long long> h(4096 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char* tag, int blocks, float ms) {
    hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    long long first = h[0], last = 0; double dur = 0;
    for (int b = 0; b < blocks; ++b) { first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]); dur += h[2 * b + 1] - h[2 * b]; }
    printf("%-34s blocks %4d: event %.2f us, first start->last end %.2f us, mean in-kernel body %.2f us\n", tag, blocks, ms * 1e3,
           (last - first) / 100.0, dur / blocks / 100.0);
  };
  for (int blocks : {256, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      float ms;
      hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(straight<256>, dim3(blocks), dim3(256), 0, 0, out, st, 1.0001f, 0.9999f); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); report("straight 512 fma (4 KB code)", blocks, ms);
      hipEventRecord(e0); hipLaunchKernelGGL(straight<2048>, dim3(blocks), dim3(256), 0, 0, out, st, 1.0001f, 0.9999f); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); report("straight 4096 fma (32 KB code)", blocks, ms);
      hipEventRecord(e0); hipLaunchKernelGGL(looped<32>, dim3(blocks), dim3(256), 0, 0, out, st, 1.0001f, 0.9999f, 64); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); report("looped 64 x 64 fma (4096 fma)", blocks, ms);
      hipEventRecord(e0); hipLaunchKernelGGL(looped<32>, dim3(blocks), dim3(256), 0, 0, out, st, 1.0001f, 0.9999f, 8); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); report("looped 8 x 64 fma (512 fma)", blocks, ms);
      hipEventRecord(e0); hipLaunchKernelGGL(looped<32>, dim3(blocks), dim3(256), 0, 0, out, st, 1.0001f, 0.9999f, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); report("empty", blocks, ms);
    }
  }
  return 0;
}
