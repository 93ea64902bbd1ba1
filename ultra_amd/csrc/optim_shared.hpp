// Optimiser types shared by optim.hip and fused.hip (the fused step's update call also refreshes the transposed weights).
#pragma once
#include "feed_body.hpp"

namespace ultr {

// A small parameter group (DLA's propensity tower: L+1 floats): norm, clip and update by ONE workgroup.
struct SmallGroup {
  float* theta; float* accum; const float* grads; int n; const float* denom;
  float scale_const, l2, clip_norm, lr; int strategy; float* out_norm;
};
// Transposed copies refreshed by the update launch: theta[off + k * ldn + n] (k < K, n < N) also goes to dst[n * ldk + k],
// ldk = K rounded up to 4.  The fused training step streams them in its dgrad GEMMs (fused.hip).
struct TransposeOpts {
  int n;
  long long off[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  float* dst[ULTR_MAX_LAYERS];
};

// Options of the fused optimiser tail (ultr_clip_adagrad3); all nullable.
struct TailOpts {
  const float* sumsq; int n_sumsq; long long n_covered;  // sum-of-squares partials of grads[0, n_covered) (raw, l2 == 0)
  const float* pub_src; float* pub_dst; int pub_n;       // results row copied out (pinned host memory) by workgroup 0
  TransposeOpts tr;                                      // tr.n > 0: transposed copies of updated weight matrices
  // a_blocks > 0: the assembly of the NEXT batch rides with the update launch as its first a_pad workgroups (nothing
  // reads the current batch any more, so it may land in the same buffers; ultr_batch_assemble_deferred)
  AssembleArgs a; int a_blocks, a_pad;
};

int clip_adagrad_impl(float* theta, float* acc, const float* grads, long long n, const float* denom, float scale_const,
                      float l2, float clip_norm, float lr, int strategy, float* out_norm, float* ws, const SmallGroup& rider,
                      TailOpts o, void* stream);

}  // namespace ultr
