// Diagnostic (not part of the library): instruction-cache residency across kernel launches.  A kernel with 32 KB
// of straight-line code is timed when it is relaunched back to back (its code stays in the 64 KB instruction
// cache) and when four such kernels alternate (128 KB of code: every launch starts cold), with 256 and 16 blocks.
//   hipcc --offload-arch=gfx950 -O2 tools/icprobe2.hip -o tools/icprobe2
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int SALT>
__global__ void straight(float* out, float a, float b) {
  float x = threadIdx.x + SALT, y = blockIdx.x;
#pragma unroll
  for (int i = 0; i < 2048; ++i) { x = __builtin_fmaf(x, a, y); y = __builtin_fmaf(y, b, x); }
  if (x == 1234.5f) out[0] = y;
}
template <int SALT>
__global__ void looped(float* out, float a, float b) {
  float x = threadIdx.x + SALT, y = blockIdx.x;
  for (int r = 0; r < 64; ++r) {
#pragma unroll
    for (int i = 0; i < 32; ++i) { x = __builtin_fmaf(x, a, y); y = __builtin_fmaf(y, b, x); }
  }
  if (x == 1234.5f) out[0] = y;
}
int main() {
  float* out; hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 16}) for (int mode = 0; mode < 4; ++mode) {
    float sum = 0; int cnt = 0;
    for (int rep = 0; rep < 40; ++rep) {
      hipEventRecord(e0);
      const int which = (mode == 1 || mode == 3) ? rep % 4 : 0;
      if (mode < 2) {
        if (which == 0) hipLaunchKernelGGL(straight<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 1) hipLaunchKernelGGL(straight<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 2) hipLaunchKernelGGL(straight<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 3) hipLaunchKernelGGL(straight<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
      } else {
        if (which == 0) hipLaunchKernelGGL(looped<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 1) hipLaunchKernelGGL(looped<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 2) hipLaunchKernelGGL(looped<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        if (which == 3) hipLaunchKernelGGL(looped<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 8) { sum += ms; cnt++; }
    }
    printf("blocks %3d  %-52s %.2f us\n", blocks, mode == 0 ? "32 KB straight-line, same kernel relaunched" : mode == 1 ? "32 KB straight-line, four kernels alternating" : mode == 2 ? "compact loop (same work), same kernel" : "compact loop, four kernels alternating", sum / cnt * 1e3);
  }
  return 0;
}
