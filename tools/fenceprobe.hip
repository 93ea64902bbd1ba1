// Diagnostic (not part of the library): what does the "last block reduces" hand-off (agent-scope release fence +
// atomic + acquire fence, common.hpp last_block_arrive) cost inside a small kernel that follows a kernel which left
// a few MB of dirty lines in L2?   hipcc --offload-arch=gfx950 -O2 -Iultra_amd/csrc tools/fenceprobe.hip -o tools/fenceprobe
#include "../ultra_amd/csrc/common.hpp"
#include <vector>
namespace ultr { void set_error(const char*, ...) {} int check_launch(const char*) { return 0; } bool g_profile_on = false;
void profile_begin(const char*, hipStream_t, void**) {} void profile_end(void*, hipStream_t) {} const char* profile_intern(const char* f, ...) { return f; } }
using namespace ultr;

__global__ void dirty(float* buf, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (float)i;
}
template <int MODE>
__global__ void __launch_bounds__(256) small(const float* in, float* rows, float* out, unsigned* counter) {
  float v = in[blockIdx.x * 256 + threadIdx.x];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) rows[blockIdx.x * 4 + (threadIdx.x >> 6)] = v;
  if (MODE == 0) return;                       // partial rows only
  if (MODE == 1) {                             // last block reduces (release/acquire fences)
    if (last_block_arrive(counter, gridDim.x)) {
      float s = threadIdx.x < gridDim.x * 4 ? rows[threadIdx.x] : 0.f;   // one load per thread, then a wave sum
      s = wave_sum(s);
      if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
    }
  }
  if (MODE == 3) {                             // sc1 stores + agent atomic + sc1 loads, no fence
    if ((threadIdx.x & 63) == 0) st_agent(&rows[blockIdx.x * 4 + (threadIdx.x >> 6)], v);
    if (last_block_arrive_sc1(counter, gridDim.x)) {
      float s = threadIdx.x < gridDim.x * 4 ? ld_agent(&rows[threadIdx.x]) : 0.f;
      s = wave_sum(s);
      if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
    }
  }
  if (MODE == 2) {                             // one float atomic per wave, no fence
    if ((threadIdx.x & 63) == 0) atomicAdd(out, v);
  }
}
int main() {
  float *big, *in, *rows, *out; unsigned* ctr;
  hipMalloc(&big, 16u << 20); hipMalloc(&in, 1 << 20); hipMalloc(&rows, 1 << 16); hipMalloc(&out, 64); hipMalloc(&ctr, 64);
  hipMemset(in, 0, 1 << 20); hipMemset(ctr, 0, 64); hipMemset(out, 0, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {16, 64}) for (int with_dirty = 0; with_dirty < 2; ++with_dirty) for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
      if (with_dirty) hipLaunchKernelGGL(dirty, dim3(512), dim3(256), 0, 0, big, (size_t)(10u << 20) / 4);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(small<0>, dim3(blocks), dim3(256), 0, 0, in, rows, out, ctr);
      if (mode == 1) hipLaunchKernelGGL(small<1>, dim3(blocks), dim3(256), 0, 0, in, rows, out, ctr);
      if (mode == 3) hipLaunchKernelGGL(small<3>, dim3(blocks), dim3(256), 0, 0, in, rows, out, ctr);
      if (mode == 2) hipLaunchKernelGGL(small<2>, dim3(blocks), dim3(256), 0, 0, in, rows, out, ctr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("blocks %3d  after 10MB dirty: %d  mode %s : %.2f us\n", blocks, with_dirty, mode == 0 ? "rows only      " : mode == 1 ? "last-block fence" : mode == 2 ? "float atomics   " : "sc1 hand-off    ", best * 1e3);
  }
  return 0;
}
