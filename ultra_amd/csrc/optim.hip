// Global-norm clip + Adagrad / SGD over a flat fp32 parameter buffer.
// Reference: tf.clip_by_global_norm + tf.train.AdagradOptimizer.apply_gradients as used in
//   dla.py:240-255, ipw_rank.py:155-165, regression_EM.py:188-194, pairwise_debias.py:185-191.
// TF semantics restated: clip: g * clip/max(||g||, clip); Adagrad: acc += g^2 (acc0 = 0.1),
// theta -= lr * g / sqrt(acc), no epsilon; tf.nn.l2_loss(p) = sum(p^2)/2 so its gradient is p.
//
// HBM-bound: 20 B per parameter per step (read g, acc, theta; write acc, theta) + 8 B for the
// norm pass (g, theta re-read, L2-resident for the DNN's <= 2 MB of parameters).
#include <new>

#include "optim_shared.hpp"

namespace ultr {

constexpr int kOptThreads = 256;
constexpr int kOptMaxBlocks = 1024;

__device__ __forceinline__ float eff_scale(const float* denom, float scale_const) {
  return denom ? scale_const / denom[0] : scale_const;
}

__device__ __forceinline__ void clip_adagrad_small(float* theta, float* accum, const float* grads, int n,
                                                   const float* denom, float scale_const, float l2, float clip_norm,
                                                   float lr, int strategy, float* out_norm) {
  __shared__ float red[4];
  const float sc = eff_scale(denom, scale_const);
  float ss[1] = {0.f};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float g = grads[i] * sc + l2 * theta[i];
    ss[0] += g * g;
  }
  block_sum<1>(ss, red);
  const float nrm = sqrtf(ss[0]);
  if (threadIdx.x == 0 && out_norm) out_norm[0] = nrm;
  const float f = clip_norm > 0.f ? clip_norm / fmaxf(nrm, clip_norm) : 1.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float g = (grads[i] * sc + l2 * theta[i]) * f;
    if (strategy == ULTR_OPT_ADAGRAD) {
      const float a = accum[i] + g * g;
      accum[i] = a;
      theta[i] -= lr * g / sqrtf(a);
    } else {
      theta[i] -= lr * g;
    }
  }
}
__global__ void __launch_bounds__(kOptThreads) gnorm_partial_kernel(const float* __restrict__ theta,
                                                                    const float* __restrict__ grads, long long n,
                                                                    const float* __restrict__ denom, float scale_const,
                                                                    float l2, float* __restrict__ ws, SmallGroup rider) {
  if (blockIdx.x == gridDim.x - 1 && rider.n > 0) {  // the extra block updates the small group completely
    clip_adagrad_small(rider.theta, rider.accum, rider.grads, rider.n, rider.denom, rider.scale_const, rider.l2,
                       rider.clip_norm, rider.lr, rider.strategy, rider.out_norm);
    return;
  }
  __shared__ float red[4];
  const float sc = eff_scale(denom, scale_const);
  float acc[1] = {0.f};
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(grads);
  const float4* t4 = reinterpret_cast<const float4*>(theta);
  const long long nb = gridDim.x - (rider.n > 0 ? 1 : 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += nb * blockDim.x) {
    const float4 g = g4[i];
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l2 != 0.f) t = t4[i];
    const float a = g.x * sc + l2 * t.x, b = g.y * sc + l2 * t.y, c = g.z * sc + l2 * t.z, d = g.w * sc + l2 * t.w;
    acc[0] += a * a + b * b + c * c + d * d;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float a = grads[i] * sc + (l2 != 0.f ? l2 * theta[i] : 0.f);
    acc[0] += a * a;
  }
  block_sum<1>(acc, red);
  if (threadIdx.x == 0) ws[8 + blockIdx.x] = acc[0];
}

__global__ void __launch_bounds__(kOptThreads) adagrad_apply_kernel(float* __restrict__ theta, float* __restrict__ accum,
                                                                    const float* __restrict__ grads, long long n,
                                                                    const float* __restrict__ denom, float scale_const,
                                                                    float l2, float clip_norm, float lr, int strategy,
                                                                    int n_partials, float* __restrict__ out_norm,
                                                                    const float* __restrict__ ws, TailOpts o,
                                                                    SmallGroup rider) {
  __shared__ float red[4];
  if ((int)blockIdx.x < o.a_pad) {  // the next batch's assembly rides along (the longer pole: first)
    if ((int)blockIdx.x < o.a_blocks) assemble_slots_body(o.a, blockIdx.x, o.a_blocks);
    return;
  }
  const unsigned int bid = blockIdx.x - o.a_pad, nbid = gridDim.x - o.a_pad;
  if (rider.n > 0 && bid == nbid - 1) {  // the extra block updates the small group completely
    clip_adagrad_small(rider.theta, rider.accum, rider.grads, rider.n, rider.denom, rider.scale_const, rider.l2,
                       rider.clip_norm, rider.lr, rider.strategy, rider.out_norm);
    // its norm goes out directly: workgroup 0 publishes the row at the same time and skips this entry
    const long long idx = rider.out_norm && o.pub_n > 0 ? rider.out_norm - o.pub_src : -1;
    if (threadIdx.x == 0 && idx >= 0 && idx < o.pub_n) o.pub_dst[idx] = rider.out_norm[0];
    return;
  }
  if (bid == 0 && o.mt.nb > 0) {  // the in-step metrics: partial rows of the launch before -> the values published below
    __shared__ float s_mt_tmp[kOptThreads];
    __shared__ float s_mt_acc[kMaxOut];
    metrics_tail_body(o.mt, s_mt_tmp, s_mt_acc);
    __threadfence_block();
    __syncthreads();
  }
  if (bid == 0 && o.em.kind != 0) {  // a handful of floats of EM state; its outputs are part of the row published below
    if (o.em.kind == ULTR_RIDE_REM) rem_update_body(o.em);
    else pd_update_body(o.em);
    __threadfence_block();
    __syncthreads();
  }
  const float sc = eff_scale(denom, scale_const);
  const long long nblk = nbid - (rider.n > 0 ? 1 : 0);
  // The block's first (usually only) batch of gradient / parameter / accumulator values is requested before the norm
  // is formed: the update needs the norm, its operands do not, and the two memory round trips overlap.
  const long long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(grads);
  float4* t4 = reinterpret_cast<float4*>(theta);
  float4* a4 = reinterpret_cast<float4*>(accum);
  const long long i0 = (long long)bid * blockDim.x + threadIdx.x;
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), t0 = g0, a0 = g0;
  if (i0 < n4) {
    g0 = g4[i0];
    t0 = t4[i0];
    if (strategy == ULTR_OPT_ADAGRAD) a0 = a4[i0];
  }
  // every block reduces the same partials in the same (fixed) order, so all blocks see one norm
  float ssq[1] = {0.f};
  float nrm;
  if (o.sumsq) {  // partials of the RAW gradient (the backward pass left them): ||g sc|| = |sc| ||g||
    for (int p = threadIdx.x; p < o.n_sumsq; p += blockDim.x) ssq[0] += o.sumsq[p];
    for (long long i = o.n_covered + threadIdx.x; i < n; i += blockDim.x) ssq[0] += grads[i] * grads[i];
    block_sum<1>(ssq, red);
    nrm = fabsf(sc) * sqrtf(ssq[0]);
  } else if (o.grid_norm) {
    // ---- the norm, formed inside this launch: partial of the elements this workgroup updates ...
    float part[1] = {0.f};
    for (long long i = i0; i < n4; i += nblk * blockDim.x) {
      const float4 g = i == i0 ? g0 : g4[i];
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l2 != 0.f) t = i == i0 ? t0 : t4[i];
      const float a = g.x * sc + l2 * t.x, b = g.y * sc + l2 * t.y, c = g.z * sc + l2 * t.z, d = g.w * sc + l2 * t.w;
      part[0] += a * a + b * b + c * c + d * d;
    }
    if (bid == 0 && threadIdx.x < (n & 3)) {
      const long long i = (n4 << 2) + threadIdx.x;
      const float a = grads[i] * sc + (l2 != 0.f ? l2 * theta[i] : 0.f);
      part[0] += a * a;
    }
    block_sum<1>(part, red);
    // ... published write-through, then the launch-wide rendezvous.  The counters only ever grow (never reset, so a
    // replayed graph needs no memset) and a stream runs its launches in order, so the arrivals of one launch share an
    // epoch: (old value) / (arrivals per launch).  Two levels, because returning atomics on one line serialise at
    // ~16 ns each (232 arrivals: 3.7 us): 16 sub-counters on lines of their own (ws[1040 + 32 k]), whose last arrivers
    // of the epoch step the top counter (ws[0], 16 per launch); everybody polls the top counter.  (2^32 arrivals, some
    // 10^8 steps, would wrap a counter: not handled.)
    float* wsw = const_cast<float*>(ws);
    if (threadIdx.x == 0) {
      st_agent(wsw + 8 + bid, part[0]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned int* top = reinterpret_cast<unsigned int*>(wsw);
      const unsigned int nb = (unsigned int)nblk;
      if (nb < 32u) {
        const unsigned int old = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = (old / nb + 1u) * nb;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
      } else {
        const unsigned int k = bid & 15u;
        unsigned int* sub = top + 1040 + 32 * k;
        const unsigned int n_sub = (nb - k + 15u) / 16u;  // update workgroups that share this sub-counter
        const unsigned int old = __hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int epoch = old / n_sub;
        if (old % n_sub == n_sub - 1u) __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = (epoch + 1u) * 16u;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n_partials; p += blockDim.x) ssq[0] += ld_agent(ws + 8 + p);
    block_sum<1>(ssq, red);
    nrm = sqrtf(ssq[0]);
  } else {
    for (int p = threadIdx.x; p < n_partials; p += blockDim.x) ssq[0] += ws[8 + p];
    block_sum<1>(ssq, red);
    nrm = sqrtf(ssq[0]);
  }
  if (bid == 0 && threadIdx.x == 0 && out_norm) out_norm[0] = nrm;
  if (bid == 0 && o.pub_n > 0) {  // the step's results row, straight into the caller's (pinned host) buffer
    for (int i = threadIdx.x; i < o.pub_n; i += blockDim.x) {
      const float* src = o.pub_src + i;
      if (rider.n > 0 && src == rider.out_norm) continue;
      o.pub_dst[i] = src == out_norm ? nrm : *src;
    }
  }
  const float f = clip_norm > 0.f ? clip_norm / fmaxf(nrm, clip_norm) : 1.f;
  for (long long i = i0; i < n4; i += nblk * blockDim.x) {
    const float4 g = i == i0 ? g0 : g4[i];
    float4 t = i == i0 ? t0 : t4[i];
    float gv[4] = {g.x, g.y, g.z, g.w};
    float tv[4] = {t.x, t.y, t.z, t.w};
    if (strategy == ULTR_OPT_ADAGRAD) {
      float4 a = i == i0 ? a0 : a4[i];
      float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gq = (gv[q] * sc + l2 * tv[q]) * f;
        av[q] += gq * gq;
        tv[q] -= lr * gq / sqrtf(av[q]);
      }
      a4[i] = make_float4(av[0], av[1], av[2], av[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[q] -= lr * (gv[q] * sc + l2 * tv[q]) * f;
    }
    t4[i] = make_float4(tv[0], tv[1], tv[2], tv[3]);
    if (o.tr.n > 0) {  // the updated values of a weight matrix also go to its transposed copy
      const long long e = i << 2;
#pragma unroll 1
      for (int j = 0; j < o.tr.n; ++j) {
        const long long rel = e - o.tr.off[j];
        if (rel >= 0 && rel < (long long)o.tr.K[j] * o.tr.ldn[j]) {
          const int k = (int)(rel / o.tr.ldn[j]), n0 = (int)(rel - (long long)k * o.tr.ldn[j]);  // ldn % 4 == 0: one row
          const int ldk = (o.tr.K[j] + 3) & ~3;
          if (o.tr.dst[j]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (n0 + q < o.tr.N[j]) o.tr.dst[j][(long long)(n0 + q) * ldk + k] = tv[q];
          }
          if (o.tr.pk[j] || o.tr.pkt[j]) pack_store(o.tr.pk[j], o.tr.pkt[j], k, n0, o.tr.ldn[j], ldk, tv[0], tv[1], tv[2], tv[3]);
        }
      }
    }
  }
  if (bid == 0 && threadIdx.x < (n & 3)) {
    const long long i = (n4 << 2) + threadIdx.x;
    const float gq = (grads[i] * sc + l2 * theta[i]) * f;
    if (strategy == ULTR_OPT_ADAGRAD) {
      const float a = accum[i] + gq * gq;
      accum[i] = a;
      theta[i] -= lr * gq / sqrtf(a);
    } else {
      theta[i] -= lr * gq;
    }
  }
}

// n <= 4096: norm and update in one single-block launch
__global__ void __launch_bounds__(kOptThreads) clip_adagrad_small_kernel(float* theta, float* accum, const float* grads,
                                                                         int n, const float* denom, float scale_const,
                                                                         float l2, float clip_norm, float lr,
                                                                         int strategy, float* out_norm) {
  clip_adagrad_small(theta, accum, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm);
}

}  // namespace ultr

using namespace ultr;

namespace ultr {
int clip_adagrad_impl(float* theta, float* acc, const float* grads, long long n, const float* denom,
                      float scale_const, float l2, float clip_norm, float lr, int strategy, float* out_norm,
                      float* ws, const SmallGroup& rider, TailOpts o, void* stream, bool take_assemble) {
  ULTR_REQUIRE(theta && grads && n > 0, ULTR_EINVAL, "ultr_clip_adagrad: bad argument");
  ULTR_REQUIRE(strategy == ULTR_OPT_ADAGRAD || strategy == ULTR_OPT_SGD, ULTR_EINVAL,
               "ultr_clip_adagrad: grad_strategy=%d", strategy);
  ULTR_REQUIRE(strategy == ULTR_OPT_SGD || acc, ULTR_EINVAL, "ultr_clip_adagrad: Adagrad needs an accumulator");
  // Every argument check comes before the parked requests (metric column sums, EM-state update, next batch's assembly)
  // are taken: a call that fails its checks leaves them parked for the caller's next, corrected call or flush.
  if (n <= 4096) {
    ULTR_REQUIRE(rider.n == 0 && !o.sumsq && o.pub_n == 0 && o.tr.n == 0 && !take_assemble, ULTR_EINVAL,
                 "ultr_clip_adagrad: the fused options need a main group of more than 4096 parameters");
  } else {
    ULTR_REQUIRE(ws, ULTR_EINVAL, "ultr_clip_adagrad: workspace required");
    ULTR_REQUIRE(((uintptr_t)theta % 16 == 0) && ((uintptr_t)grads % 16 == 0) && (!acc || (uintptr_t)acc % 16 == 0),
                 ULTR_EINVAL, "ultr_clip_adagrad: buffers must be 16-byte aligned");
    if (l2 != 0.f && o.sumsq) {  // the partials are of the raw gradient: with an l2 term the launch forms the norm itself
      o.sumsq = nullptr;
      o.grid_norm = 1;
    }
    ULTR_REQUIRE(!o.sumsq || (o.n_sumsq > 0 && o.n_covered >= 0 && o.n_covered <= n && n - o.n_covered <= 65536), ULTR_EINVAL,
                 "ultr_clip_adagrad3: bad sum-of-squares partials (n_sumsq=%d n_covered=%lld)", o.n_sumsq, o.n_covered);
    ULTR_REQUIRE(o.pub_n == 0 || (o.pub_src && o.pub_dst && o.pub_n > 0), ULTR_EINVAL, "ultr_clip_adagrad3: bad publish range");
  }
  if (g_pending_mtail.on) {  // the column sums of the fused step's in-step metrics: carried likewise
    g_pending_mtail.on = false;
    if (n > 4096) {
      o.mt = g_pending_mtail.t;
    } else {
      const int rc = launch_metrics_tail_now(g_pending_mtail.t, as_stream(stream));
      if (rc) return rc;
    }
  }
  if (g_pending_em.on) {  // a parked EM-state update: carried by the big update launch, else a launch of its own now
    g_pending_em.on = false;
    if (n > 4096) {
      o.em = g_pending_em.e;
    } else {
      const int rc = launch_em_now(g_pending_em.e, as_stream(stream));
      if (rc) return rc;
    }
  }
  if (n <= 4096) {
    {
      ultr::ProfScope prof_("clip_adagrad_small_kernel", as_stream(stream));
      hipLaunchKernelGGL(clip_adagrad_small_kernel, dim3(1), dim3(kOptThreads), 0, as_stream(stream), theta, acc, grads,
                       (int)n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm);
    }
    return check_launch("clip_adagrad_small_kernel");
  }
  if (take_assemble && g_pending_assemble.on && g_pending_assemble.nblocks <= 4096) {  // the next batch's assembly rides along
    o.a = g_pending_assemble.a;
    o.a_blocks = (int)g_pending_assemble.nblocks; o.a_pad = (o.a_blocks + 7) & ~7;
    g_pending_assemble.on = false;
  }
  long long blocks = cdiv(n >> 2, kOptThreads);
  if (blocks > kOptMaxBlocks) blocks = kOptMaxBlocks;
  if (blocks < 1) blocks = 1;
  SmallGroup none{};
  if (!o.sumsq && !o.grid_norm) {
    {
      ultr::ProfScope prof_("gnorm_partial_kernel", as_stream(stream));
      hipLaunchKernelGGL(gnorm_partial_kernel, dim3((unsigned)(blocks + (rider.n > 0 ? 1 : 0))), dim3(kOptThreads), 0,
                       as_stream(stream), theta, grads, n, denom, scale_const, l2, ws, rider);
    }
    int rc = check_launch("gnorm_partial_kernel");
    if (rc) return rc;
  }
  const SmallGroup& apply_rider = (o.sumsq || o.grid_norm) ? rider : none;  // no norm launch: the small group rides with the update
  {
    ultr::ProfScope prof_("adagrad_apply_kernel", as_stream(stream));
    hipLaunchKernelGGL(adagrad_apply_kernel, dim3((unsigned)(blocks + (apply_rider.n > 0 ? 1 : 0) + o.a_pad)), dim3(kOptThreads), 0,
                     as_stream(stream), theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, (int)blocks,
                     out_norm, ws, o, apply_rider);
  }
  return check_launch("adagrad_apply_kernel");
}
}  // namespace ultr

extern "C" int ultr_clip_adagrad(float* theta, float* acc, const float* grads, long long n, const float* denom,
                                 float scale_const, float l2, float clip_norm, float lr, int strategy,
                                 float* out_norm, float* ws, void* stream) {
  SmallGroup none{};
  return clip_adagrad_impl(theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm, ws, none, TailOpts{}, stream);
}

extern "C" int ultr_clip_adagrad2(float* theta, float* acc, const float* grads, long long n, const float* denom,
                                  float scale_const, float l2, float clip_norm, float lr, int strategy,
                                  float* out_norm, float* ws, float* theta_s, float* acc_s, const float* grads_s,
                                  int n_s, const float* denom_s, float scale_const_s, float l2_s, float clip_norm_s,
                                  float lr_s, float* out_norm_s, void* stream) {
  ULTR_REQUIRE(theta_s && grads_s && n_s > 0 && n_s <= 4096, ULTR_EINVAL, "ultr_clip_adagrad2: small group needs 1..4096 parameters");
  ULTR_REQUIRE(strategy == ULTR_OPT_SGD || acc_s, ULTR_EINVAL, "ultr_clip_adagrad2: Adagrad needs an accumulator");
  SmallGroup r{theta_s, acc_s, grads_s, n_s, denom_s, scale_const_s, l2_s, clip_norm_s, lr_s, strategy, out_norm_s};
  return clip_adagrad_impl(theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm, ws, r, TailOpts{}, stream);
}

extern "C" int ultr_clip_adagrad3(float* theta, float* acc, const float* grads, long long n, const float* denom,
                                  float scale_const, float l2, float clip_norm, float lr, int strategy,
                                  float* out_norm, float* ws, float* theta_s, float* acc_s, const float* grads_s,
                                  int n_s, const float* denom_s, float scale_const_s, float l2_s, float clip_norm_s,
                                  float lr_s, float* out_norm_s, const float* sumsq, int n_sumsq, long long n_covered,
                                  const float* publish_src, float* publish_dst, int publish_n, void* stream) {
  SmallGroup r{};
  if (theta_s) {
    ULTR_REQUIRE(grads_s && n_s > 0 && n_s <= 4096, ULTR_EINVAL, "ultr_clip_adagrad3: small group needs 1..4096 parameters");
    ULTR_REQUIRE(strategy == ULTR_OPT_SGD || acc_s, ULTR_EINVAL, "ultr_clip_adagrad3: Adagrad needs an accumulator");
    r = SmallGroup{theta_s, acc_s, grads_s, n_s, denom_s, scale_const_s, l2_s, clip_norm_s, lr_s, strategy, out_norm_s};
  }
  TailOpts o{sumsq, n_sumsq, n_covered, publish_src, publish_dst, publish_n, TransposeOpts{}, AssembleArgs{}, 0, 0};
  if (!sumsq && n_sumsq == -1) { o.n_sumsq = 0; o.grid_norm = 1; }  // no partials: the update launch forms the norm itself
  return clip_adagrad_impl(theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm, ws, r, o, stream);
}

// ---- parked-request contexts ---------------------------------------------------------------------------------------
// Every *_deferred call parks its request in the calling thread's BOUND context: the thread's own default context, or
// one the caller made (ultr_pending_create) and bound (ultr_pending_bind).  One context per model keeps the requests of
// models stepped alternately on one host thread apart (ultr_hip.h, Conventions).
namespace ultr {
struct PendingSet {
  PendingLoss loss;
  PendingEm em;
  PendingMetrics metrics;
  PendingMetricsTail mtail;
  PendingAssemble assemble;
};
static thread_local PendingSet t_default_set = {};
static thread_local PendingSet* t_bound = nullptr;  // nullptr: the default set
static inline PendingSet* cur_set() { return t_bound ? t_bound : &t_default_set; }
PendingLoss& pending_loss() { return cur_set()->loss; }
PendingEm& pending_em() { return cur_set()->em; }
PendingMetrics& pending_metrics() { return cur_set()->metrics; }
PendingMetricsTail& pending_mtail() { return cur_set()->mtail; }
PendingAssemble& pending_assemble() { return cur_set()->assemble; }
}  // namespace ultr

extern "C" void* ultr_pending_create(void) {
  return new (std::nothrow) ultr::PendingSet();
}

extern "C" void* ultr_pending_bind(void* ctx) {
  void* was = ultr::t_bound;
  ultr::t_bound = static_cast<ultr::PendingSet*>(ctx);
  return was;
}

extern "C" int ultr_pending_count(void* ctx) {
  const ultr::PendingSet* s = ctx ? static_cast<const ultr::PendingSet*>(ctx) : &ultr::t_default_set;
  return (s->loss.kind != ULTR_RIDE_NONE) + (s->em.on ? 1 : 0) + (s->metrics.on ? 1 : 0) + (s->mtail.on ? 1 : 0) +
         (s->assemble.on ? 1 : 0);
}

extern "C" void ultr_pending_destroy(void* ctx) {
  if (!ctx) return;
  if (ultr::t_bound == ctx) ultr::t_bound = nullptr;  // this thread falls back to its own; other threads: the caller's job
  delete static_cast<ultr::PendingSet*>(ctx);
}
