// Optimiser types shared by optim.hip and fused.hip (the fused step's update call also refreshes the transposed weights).
#pragma once
#include "feed_body.hpp"
#include "loss_body.hpp"
#include "metrics_body.hpp"

namespace ultr {

// A small parameter group (DLA's propensity tower: L+1 floats): norm, clip and update by ONE workgroup.
struct SmallGroup {
  float* theta; float* accum; const float* grads; int n; const float* denom;
  float scale_const, l2, clip_norm, lr; int strategy; float* out_norm;
};
// Transposed copies refreshed by the update launch: theta[off + k * ldn + n] (k < K, n < N) also goes to dst[n * ldk + k],
// ldk = K rounded up to 4.  The fused training step streams them in its dgrad GEMMs (fused.hip).
// pk / pkt (bf16 step): the same values rounded to bf16 into the packed copies the bf16 GEMMs stream -- pk[(k / 4)][n][k % 4]
// over all ldn columns, pkt[(n / 4)][k][n % 4] with row length ldk.  Any of dst / pk / pkt may be null.
struct TransposeOpts {
  int n;
  long long off[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  float* dst[ULTR_MAX_LAYERS];
  unsigned short* pk[ULTR_MAX_LAYERS];
  unsigned short* pkt[ULTR_MAX_LAYERS];
};

// fp32 -> bf16, round to nearest even (two values per instruction on gfx950)
// (through the compiler's vector conversion, which selects v_cvt_pk_bf16_f32: an inline-asm statement hides the VALU
// write from the hazard recogniser, and an MFMA that reads the result one instruction later gets stale operands --
// observed on hardware)
typedef __bf16 ultr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ultr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  const ultr_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ultr_bf16x2));
}
// Four consecutive columns n0 .. n0 + 3 (n0 % 4 == 0) of row k of a weight matrix into its packed bf16 copies.
__device__ __forceinline__ void pack_store(unsigned short* pk, unsigned short* pkt, int k, int n0, int ldn, int ldk,
                                           float v0, float v1, float v2, float v3) {
  const unsigned lo = pk_bf16(v0, v1), hi = pk_bf16(v2, v3);
  if (pk) {
    unsigned short* d = pk + ((long long)(k >> 2) * ldn + n0) * 4 + (k & 3);
    d[0] = (unsigned short)(lo & 0xFFFFu); d[4] = (unsigned short)(lo >> 16);
    d[8] = (unsigned short)(hi & 0xFFFFu); d[12] = (unsigned short)(hi >> 16);
  }
  if (pkt) *reinterpret_cast<uint2*>(pkt + ((long long)(n0 >> 2) * ldk + k) * 4) = make_uint2(lo, hi);
}

// Options of the fused optimiser tail (ultr_clip_adagrad3); all nullable.
struct TailOpts {
  const float* sumsq; int n_sumsq; long long n_covered;  // sum-of-squares partials of grads[0, n_covered) (raw, l2 == 0)
  const float* pub_src; float* pub_dst; int pub_n;       // results row copied out (pinned host memory) by workgroup 0
  TransposeOpts tr;                                      // tr.n > 0: transposed copies of updated weight matrices
  // a_blocks > 0: the assembly of the NEXT batch rides with the update launch as its first a_pad workgroups (nothing
  // reads the current batch any more, so it may land in the same buffers; ultr_batch_assemble_deferred)
  AssembleArgs a; int a_blocks, a_pad;
  // em.kind != 0: the EM-state update of regression-EM / pairwise debiasing, run by workgroup 0 before it publishes
  EmUpdate em;
  // mt.nb > 0: the column sums of a metric evaluation whose rider workgroups left only partial rows (the fused step's
  // slab-reduction launch), formed by workgroup 0 before it publishes
  MetricsTail mt;
  // grid_norm != 0 (no sumsq): the launch forms the gradient norm itself -- every update workgroup leaves the partial
  // of its own elements, arrives at a counter (ws[0]) and waits for the launch's other update workgroups before it reads
  // all partials in fixed order.  For gradients that are final only just before the update (data parallel: after the
  // collective; an l2 term): one launch instead of a norm launch + an update launch.  The update workgroups (at most
  // 1024 of 256 threads) are co-resident on the chip, so the wait cannot starve a workgroup it is waiting for.
  int grid_norm;
};

int clip_adagrad_impl(float* theta, float* acc, const float* grads, long long n, const float* denom, float scale_const,
                      float l2, float clip_norm, float lr, int strategy, float* out_norm, float* ws, const SmallGroup& rider,
                      TailOpts o, void* stream, bool take_assemble = false);

}  // namespace ultr
