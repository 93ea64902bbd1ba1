// Diagnostic (not part of the library): floor for "every workgroup pulls its 32-row input panel into LDS" right
// after another kernel wrote those rows (the activations of the previous layer), at the headline shape.
//   hipcc --offload-arch=gfx950 -O2 tools/stageprobe.hip -o tools/stageprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void producer(float* act, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) act[i] = (float)(i & 1023) * 1e-3f;
}
// grid (rowblocks, colblocks): each WG stages rows [32 bx, 32 bx + 32) x K floats; NLD float4 loads per thread in flight
template <int NLD>
__global__ void __launch_bounds__(256) stage(const float* __restrict__ act, int K, float* sink, long long* stamps) {
  extern __shared__ float4 lds[];
  const long long t0 = wall_clock64();
  const float4* src = reinterpret_cast<const float4*>(act + (size_t)blockIdx.x * 32 * K);
  const int n4 = 32 * K / 4;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < n4; i += 256 * NLD) {
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = src[min(i + 256 * u, n4 - 1)];
#pragma unroll
    for (int u = 0; u < NLD; ++u) if (i + 256 * u < n4) lds[i + 256 * u] = v[u];
  }
  __syncthreads();
  const long long t1 = wall_clock64();
  acc = lds[(threadIdx.x * 7) % n4];
  if (acc.x == 123.456f) sink[0] = acc.y;
  if (threadIdx.x == 0) { const int b = blockIdx.y * gridDim.x + blockIdx.x; stamps[2 * b] = t0; stamps[2 * b + 1] = t1; }
}

int main() {
  const int M = 2560;
  float *act, *sink; long long* st;
  hipMalloc(&act, (size_t)M * 1024 * 4); hipMalloc(&sink, 64); hipMalloc(&st, 4096 * 16);
  std::vector<long long> h(4096 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode)
  for (int K : {256, 512}) {
    for (int cb : {1, 4}) {
      for (int nld = 0; nld < 3; ++nld) {
        float best = 1e9, ms; double body = 0, span = 0;
        for (int rep = 0; rep < 5; ++rep) {
          if (mode == 0) hipLaunchKernelGGL(producer, dim3(512), dim3(256), 0, 0, act, (size_t)M * K);
          hipEventRecord(e0);
          const size_t smem = (size_t)32 * K * 4;
          if (nld == 0) { hipFuncSetAttribute((const void*)stage<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(stage<4>, dim3(M / 32, cb), dim3(256), smem, 0, act, K, sink, st); }
          if (nld == 1) { hipFuncSetAttribute((const void*)stage<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(stage<8>, dim3(M / 32, cb), dim3(256), smem, 0, act, K, sink, st); }
          if (nld == 2) { hipFuncSetAttribute((const void*)stage<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); hipLaunchKernelGGL(stage<16>, dim3(M / 32, cb), dim3(256), smem, 0, act, K, sink, st); }
          hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
          const int nb = M / 32 * cb;
          hipMemcpy(h.data(), st, nb * 16, hipMemcpyDeviceToHost);
          long long first = h[0], last = 0; double d = 0;
          for (int b = 0; b < nb; ++b) { first = std::min(first, h[2 * b]); last = std::max(last, h[2 * b + 1]); d += h[2 * b + 1] - h[2 * b]; }
          if (ms < best) { best = ms; body = d / nb / 100.0; span = (last - first) / 100.0; }
        }
        printf("%s K %4d  col-blocks %d  loads in flight %2d : event %.2f us, span %.2f us, mean staging %.2f us\n", mode == 0 ? "after producer " : "no producer    ", K, cb, nld == 0 ? 4 : nld == 1 ? 8 : 16, best * 1e3, span, body);
      }
    }
  }
  return 0;
}
