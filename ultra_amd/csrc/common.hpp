// Shared device helpers for libultr_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ultr_hip.h"

namespace ultr {

// ---- error plumbing (host) ---------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define ULTR_REQUIRE(cond, code, ...)          \
  do {                                         \
    if (!(cond)) {                             \
      ::ultr::set_error(__VA_ARGS__);          \
      return (code);                           \
    }                                          \
  } while (0)

// ---- opt-in per-kernel timing (ultr_profile_enable): HIP events around every launch, on the launch
// stream.  Off by default: one predictable branch per launch.
extern bool g_profile_on;
const char* profile_intern(const char* fmt, ...);
void profile_begin(const char* name, hipStream_t s, void** handle);
void profile_end(void* handle, hipStream_t s);
struct ProfScope {
  void* h;
  hipStream_t s;
  ProfScope(const char* name, hipStream_t st) : h(nullptr), s(st) {
    if (g_profile_on) profile_begin(name, st, &h);
  }
  ~ProfScope() {
    if (h) profile_end(h, s);
  }
};

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// A kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) is a PER-DEVICE attribute: a process that
// drives several GPUs must raise it on each.  One flag per (call site, device), relaxed atomics: the worst a race
// between host threads does is set the attribute twice.
struct PerDeviceOnce {
  unsigned char done[64] = {};
};
static inline int raise_lds_limit(PerDeviceOnce& once, const void* kernel, int bytes, const char* who) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev >= 0 && __atomic_load_n(&once.done[dev], __ATOMIC_RELAXED)) return ULTR_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot raise the dynamic LDS limit to %d bytes", who, bytes);
    return ULTR_EHIP;
  }
  if (dev >= 0) __atomic_store_n(&once.done[dev], (unsigned char)1, __ATOMIC_RELAXED);
  return ULTR_OK;
}
static inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// ---- wave / block reductions (wave = 64 lanes) -------------------------------------------
// Butterfly reductions at VALU speed: DPP lane permutes inside a 16-lane row (quad_perm, row_half_mirror,
// row_mirror), then gfx950's v_permlane16_swap / v_permlane32_swap across rows and wave halves.  A
// ds_bpermute-based __shfl_xor costs an LDS round trip (~100 cycles) per step, and these sums sit on
// the critical path of kernels that run ONE wave per SIMD.
typedef unsigned int ultr_u2 __attribute__((ext_vector_type(2)));
#define ULTR_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xF, 0xF, true))
__device__ __forceinline__ float xor16_partner(float v, float* other) {
  const ultr_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  *other = __uint_as_float(r[1]);
  return __uint_as_float(r[0]);
}
__device__ __forceinline__ float xor32_partner(float v, float* other) {
  const ultr_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  *other = __uint_as_float(r[1]);
  return __uint_as_float(r[0]);
}
// sum over groups of `width` consecutive lanes (width a power of two <= 64); every lane gets the result
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
  if (WIDTH > 1) v += ULTR_DPP_F(v, 0xB1);   // quad_perm [1,0,3,2]
  if (WIDTH > 2) v += ULTR_DPP_F(v, 0x4E);   // quad_perm [2,3,0,1]
  if (WIDTH > 4) v += ULTR_DPP_F(v, 0x141);  // row_half_mirror (quads already uniform)
  if (WIDTH > 8) v += ULTR_DPP_F(v, 0x140);  // row_mirror
  if (WIDTH > 16) { float o; const float a = xor16_partner(v, &o); v = a + o; }
  if (WIDTH > 32) { float o; const float a = xor32_partner(v, &o); v = a + o; }
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
  if (WIDTH > 1) v = fmaxf(v, ULTR_DPP_F(v, 0xB1));
  if (WIDTH > 2) v = fmaxf(v, ULTR_DPP_F(v, 0x4E));
  if (WIDTH > 4) v = fmaxf(v, ULTR_DPP_F(v, 0x141));
  if (WIDTH > 8) v = fmaxf(v, ULTR_DPP_F(v, 0x140));
  if (WIDTH > 16) { float o; const float a = xor16_partner(v, &o); v = fmaxf(a, o); }
  if (WIDTH > 32) { float o; const float a = xor32_partner(v, &o); v = fmaxf(a, o); }
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }
// sum over the 32 lanes of one wave half
__device__ __forceinline__ float half_sum(float v) { return group_sum<32>(v); }
// run-time width (power of two <= 64)
__device__ __forceinline__ float group_sum_rt(float v, int width) {
  switch (width) {
    case 1: return v;
    case 2: return group_sum<2>(v);
    case 4: return group_sum<4>(v);
    case 8: return group_sum<8>(v);
    case 16: return group_sum<16>(v);
    case 32: return group_sum<32>(v);
    default: return group_sum<64>(v);
  }
}
__device__ __forceinline__ float group_max_rt(float v, int width) {
  switch (width) {
    case 1: return v;
    case 2: return group_max<2>(v);
    case 4: return group_max<4>(v);
    case 8: return group_max<8>(v);
    case 16: return group_max<16>(v);
    case 32: return group_max<32>(v);
    default: return group_max<64>(v);
  }
}

// Block-wide sum of NV values per thread; result valid in every thread.  `red` is LDS scratch of
// at least NV * (blockDim.x/64) floats.  Deterministic (fixed order).
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * nw + wid] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[i * nw + w];
    v[i] = s;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, which makes
// the compiler drain every outstanding global load (s_waitcnt vmcnt(0)) -- fatal for kernels that keep
// several chunks of prefetch in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- cross-workgroup "last arriver reduces" hand-off (agent-scope release/acquire) --------
// Every block stores its partials with plain stores, then calls arrive(); the call returns true
// in exactly one block (the last to arrive), after which that block may read every block's
// partials with plain loads.  The counter is left at zero for the next launch.
__device__ __forceinline__ bool last_block_arrive(unsigned int* counter, unsigned int n_blocks) {
  __shared__ unsigned int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int last = (t == n_blocks - 1) ? 1u : 0u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0u;
}

// Column sums (max for columns >= max_from) of the partial rows [G][ns] the blocks of a launch left behind, by
// the last block, with ALL its threads: thread (rg, s) takes rows rg, rg + RG, ... of column s, four independent
// loads at a time, and the RG partials of a column meet in LDS in fixed order.  A thread that walks the rows of
// its column alone pays one L2 round trip per row (~0.3 us each: measured 5 us for 16 rows, tools/fenceprobe.hip).
// s_tmp: blockDim.x floats of LDS; the totals land in s_out[0..ns).  Requires ns <= blockDim.x.
template <bool SC1 = false>
__device__ __forceinline__ void reduce_rows(const float* __restrict__ rows, int G, int ns, int max_from, float* s_tmp,
                                            float* s_out) {
  const int NT = blockDim.x, RG = NT / ns;
  const int s = threadIdx.x % ns, rg = threadIdx.x / ns;
  const bool is_max = s >= max_from;
  float acc = is_max ? -INFINITY : 0.f;
  if (rg < RG) {
    int b0 = rg;
    // long walks (thousands of rows: the large-batch loss and metric kernels) keep 16 loads in flight per thread
    for (; b0 + 12 * RG < G; b0 += 16 * RG) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* src = rows + (long long)min(b0 + u * RG, G - 1) * ns + s;
        v[u] = SC1 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (b0 + u * RG < G) acc = is_max ? fmaxf(acc, v[u]) : acc + v[u];
    }
    for (; b0 < G; b0 += 4 * RG) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* src = rows + (long long)min(b0 + u * RG, G - 1) * ns + s;
        v[u] = SC1 ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (b0 + u * RG < G) acc = is_max ? fmaxf(acc, v[u]) : acc + v[u];
    }
  }
  __syncthreads();
  s_tmp[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < ns) {
    float r = s_tmp[threadIdx.x];
    for (int g = 1; g < RG; ++g) r = is_max ? fmaxf(r, s_tmp[g * ns + threadIdx.x]) : r + s_tmp[g * ns + threadIdx.x];
    s_out[threadIdx.x] = r;
  }
  __syncthreads();
}

// The same hand-off WITHOUT cache maintenance.  The release / acquire fences above write back and invalidate the
// whole L2 of the XCD; measured on MI355X (tools/fenceprobe.hip) they add 3 us to a 16-block kernel and 15 us to a
// 64-block one.  Here every handed-off word is written with st_agent() (an sc1 store: it bypasses the
// non-coherent L2 path) and read with ld_agent() (an sc1 load), the counter is an agent-scope atomic, and no
// fence is needed: the storing waves drain their stores (vmcnt(0)) before the workgroup barrier that precedes the
// add, and the last block loads only after its add has returned and its own barrier.
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_block_arrive_sc1(unsigned int* counter, unsigned int n_blocks) {
  __shared__ unsigned int s_last_sc1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned int last = (t == n_blocks - 1) ? 1u : 0u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last_sc1 = last;
  }
  __syncthreads();
  return s_last_sc1 != 0u;
}

// The same hand-off for MANY workgroups.  Returning atomics on one address -- or on one 128-byte line -- serialise at
// ~16 ns each (1024 arrivals: 16 us, most of it in the launch's tail, when the last wave of workgroups finishes together),
// so from kArriveTwoLevelMin workgroups on the arrival goes through 16 sub-counters, each on a line of its own, whose
// last arrivers meet at the top counter.  The workspace then keeps its partial rows behind the sub-counters:
// ws[0] top counter, ws[32 + 32 k] sub-counter k, rows from arrive_rows(ws, nb) -- writers and the reader apply the same
// rule, and the ultr_*_workspace_floats sizes include the room.  Counters return to zero by themselves.
constexpr unsigned int kArriveTwoLevelMin = 128;
constexpr int kArriveRowsFar = 576;  // floats in front of the rows when the arrival has two levels
__device__ __forceinline__ float* arrive_rows(float* ws, unsigned int nb) {
  return ws + (nb >= kArriveTwoLevelMin ? kArriveRowsFar : 8);
}
__device__ __forceinline__ bool last_block_arrive_ws(float* ws, unsigned int bid, unsigned int nb) {
  unsigned int* top = reinterpret_cast<unsigned int*>(ws);
  if (nb < kArriveTwoLevelMin) return last_block_arrive_sc1(top, nb);
  __shared__ unsigned int s_last_2l;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* sub = top + 32 + 32 * (bid & 15u);
    const unsigned int n_sub = (nb - (bid & 15u) + 15u) / 16u;  // workgroups that share this sub-counter
    unsigned int last = 0u;
    const unsigned int t = __hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == n_sub - 1u) {
      __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int t2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t2 == 15u) {
        __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1u;
      }
    }
    s_last_2l = last;
  }
  __syncthreads();
  return s_last_2l != 0u;
}

// ---- Philox4x32-10 -------------------------------------------------------------------------
struct Philox4 {
  uint32_t w0, w1, w2, w3;
};
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.w0 = c0; o.w1 = c1; o.w2 = c2; o.w3 = c3;
  return o;
}
// two 32-bit words -> double in [0,1), the genrand_res53 construction of CPython's random.random()
__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}
// word (r & 3) of a block, selected with compares between VALUES.  Left to itself the optimiser folds a select between
// the four member loads into one load at a computed offset, which keeps the block in private memory: promoted to LDS,
// indexed by the work-item's flat id, which is derived from the dispatch packet at the top of the kernel (measured:
// +13 us on a 12 us launch).  The empty asm makes the four words opaque register values first.
__host__ __device__ __forceinline__ uint32_t philox_word(const Philox4& b, int r) {
  uint32_t a0 = b.w0, a1 = b.w1, a2 = b.w2, a3 = b.w3;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#endif
  return (r & 2) ? ((r & 1) ? a3 : a2) : ((r & 1) ? a1 : a0);
}
// 32-bit word -> float in [0,1) (24 bits)
__host__ __device__ __forceinline__ float u24(uint32_t a) { return (float)(a >> 8) * (1.0f / 16777216.0f); }

// ---- activations (base_ranking_model.py:59-65), derivative through the OUTPUT value ---------
#define ULTR_SELU_ALPHA 1.6732632423543772848170429916717f
#define ULTR_SELU_SCALE 1.0507009873554804934193349852946f

// exp(z) - 1 for z <= 0 through the hardware exponential: 3 instructions instead of libm's ~45.  The
// absolute error is one ulp of exp(z) <= 6e-8, which is what matters downstream (the activations are O(1)
// and the derivative is taken from the output, a + 1 = exp(z)); the kernels run one wave per SIMD, where
// every instruction of the epilogue is on the critical path.
__device__ __forceinline__ float expm1_neg(float z) { return __expf(z) - 1.f; }
__device__ __forceinline__ float act_fwd(int kind, float z) {
  switch (kind) {
    case ULTR_ACT_ELU: return z > 0.f ? z : expm1_neg(z);
    case ULTR_ACT_RELU: return fmaxf(z, 0.f);
    case ULTR_ACT_SELU: return ULTR_SELU_SCALE * (z >= 0.f ? z : ULTR_SELU_ALPHA * expm1_neg(z));
    case ULTR_ACT_TANH: return tanhf(z);
    default: return 1.f / (1.f + expf(-z));
  }
}
__device__ __forceinline__ float act_grad_from_out(int kind, float a) {
  switch (kind) {
    case ULTR_ACT_ELU: return a > 0.f ? 1.f : a + 1.f;
    case ULTR_ACT_RELU: return a > 0.f ? 1.f : 0.f;
    case ULTR_ACT_SELU: return a > 0.f ? ULTR_SELU_SCALE : a + ULTR_SELU_SCALE * ULTR_SELU_ALPHA;
    case ULTR_ACT_TANH: return 1.f - a * a;
    default: return a * (1.f - a);
  }
}

}  // namespace ultr
