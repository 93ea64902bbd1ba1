// Diagnostic (not part of the library): how fast can the CUs pull an L2-resident working set into registers?
// Every workgroup streams the same `bytes`-sized region (so after the first pass it lives in each XCD's L2),
// with 16-byte or 4-byte loads and a configurable number of waves per workgroup.
//   hipcc --offload-arch=gfx950 -O2 tools/bwprobe.hip -o tools/bwprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int UNROLL>
__global__ void rd16(const float4* __restrict__ src, size_t n16, int reps, float* sink) {
  float4 acc = make_float4(0, 0, 0, 0);
  const size_t stride = blockDim.x;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x + (size_t)(blockIdx.x % 7) * stride; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
      float4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
  }
  if (acc.x == 12345.f) sink[0] = acc.y + acc.z + acc.w;
}
template <int UNROLL>
__global__ void rd4(const float* __restrict__ src, size_t n, int reps, float* sink) {
  float acc = 0;
  const size_t stride = blockDim.x;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = threadIdx.x + (size_t)(blockIdx.x % 7) * stride; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
      float v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  }
  if (acc == 12345.f) sink[0] = acc;
}

int main() {
  float* buf; float* sink;
  const size_t maxb = 64u << 20;
  hipMalloc(&buf, maxb); hipMemset(buf, 0, maxb); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t bytes : {(size_t)256 << 10, (size_t)1 << 20, (size_t)8 << 20}) {
    for (int threads : {64, 256, 512, 1024}) {
      for (int blocks : {256, 512, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
          const int reps = (int)((64u << 20) / bytes) ;
          auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(rd16<8>, dim3(blocks), dim3(threads), 0, 0, (const float4*)buf, bytes / 16, reps, sink);
            if (mode == 1) hipLaunchKernelGGL(rd16<2>, dim3(blocks), dim3(threads), 0, 0, (const float4*)buf, bytes / 16, reps, sink);
            if (mode == 2) hipLaunchKernelGGL(rd4<16>, dim3(blocks), dim3(threads), 0, 0, buf, bytes / 4, reps, sink);
          };
          launch(); hipDeviceSynchronize();
          hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          const double total = (double)blocks * reps * bytes;
          printf("set %5zu KB  threads %4d  blocks %4d  %-8s : %8.1f us  %7.2f TB/s aggregate  %6.1f GB/s per block\n", bytes >> 10, threads,
                 blocks, mode == 0 ? "x4 u8" : mode == 1 ? "x4 u2" : "x1 u16", ms * 1e3, total / (ms * 1e-3) / 1e12,
                 total / blocks / (ms * 1e-3) / 1e9);
        }
      }
    }
  }
  return 0;
}
