// Diagnostic (not part of the library): the fp32 MFMA rate this box sustains, against the 157.3 TFLOP/s of the data
// sheet ("nominal values, builder must re-measure with a peak microbenchmark on the box").
// Register-only loops of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, 1-4 waves per SIMD, 1-4 independent
// accumulator chains per wave, for ~1 ms and ~20 ms (clock behaviour under sustained matrix load).
//   hipcc --offload-arch=gfx950 -O2 tools/mfmaprobe.hip -o tools/mfmaprobe
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma32(float* out, int iters) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void __launch_bounds__(256) mfma16(float* out, int iters) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 4; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* tag, K kernel, int chains, double flops_per_mfma, int blocks, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 16);  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 /* waves */ * iters * 8.0 * chains * flops_per_mfma;
  printf("%-14s chains %d  workgroups %5d (%d waves/SIMD)  %8.3f ms  %7.1f TFLOP/s  %5.1f %% of 157.3\n", tag, chains, blocks,
         blocks / 256, ms, flops / (ms * 1e-3) / 1e12, 100.0 * flops / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
  float* out;
  hipMalloc(&out, sizeof(float) * 256 * 4096);
  for (int wg : {256, 512, 1024}) {
    run("32x32x2 f32", mfma32<1>, 1, 4096.0, wg, 2000, out);
    run("32x32x2 f32", mfma32<2>, 2, 4096.0, wg, 2000, out);
    run("32x32x2 f32", mfma32<4>, 4, 4096.0, wg, 1000, out);
    run("16x16x4 f32", mfma16<1>, 1, 2048.0, wg, 4000, out);
    run("16x16x4 f32", mfma16<4>, 4, 2048.0, wg, 2000, out);
  }
  printf("-- sustained: 20x longer\n");
  run("32x32x2 f32", mfma32<4>, 4, 4096.0, 1024, 20000, out);
  run("16x16x4 f32", mfma16<4>, 4, 2048.0, 1024, 40000, out);
  return 0;
}
