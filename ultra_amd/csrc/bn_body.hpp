// Batch normalisation of the DNN scorer (norm = 'batch'), training-mode batch statistics.
// Reference: ultra/ranking_model/DNN.py:70-84 (one tf.keras.layers.BatchNormalization per layer, called with
// training=is_training), base_ranking_model.py:67-70.  keras defaults: axis -1, epsilon 1e-3, centre and scale; a
// [M, K] input takes the non-fused path: tf.nn.moments over the rows (biased variance), then
// (x - mean) * rsqrt(var + eps) * gamma + beta.  The moving-average update ops are never fetched by the reference's
// training step (dla.py:237-255 runs the optimiser's ops only), so evaluation normalises with the INITIAL moving
// statistics, mean 0 and variance 1 (ULTR_NORM_BATCH_EVAL).
//
// The statistics couple every row of the batch, so this path cannot ride in the list-owning fused step; it runs the
// per-layer plan with the normalisation pulled OUT of the GEMM kernels:
//   forward, per layer:  column sums over the rows -> mean ; column sums of (x - mean)^2 -> rstd ;
//                        a = gamma * xhat + beta and xhat written by an elementwise pass ; the GEMM kernel then runs
//                        as a plain layer on `a` (its norm = none mode)
//   backward, per layer: the weight-gradient GEMM takes xhat (dW = gamma G + beta S, dgamma, dbeta fall out of the
//                        slab reduction exactly as for LayerNorm: the algebra is the same, only the statistics run
//                        along the other axis) ; dgrad gives dh = dz W^T ; two more column sums
//                        c1 = mean_m(gamma dh), c2 = mean_m(gamma dh xhat) ; then
//                        dz_prev = rstd (gamma dh - c1 - xhat c2) * act'(act_prev)
// Column sums are deterministic: fixed row splits, partials added in split order.  Not a tuned path (the default
// scorer uses LayerNorm); every kernel here is a plain streaming pass.
#pragma once
#include "dnn_shared.hpp"

namespace ultr {

constexpr int kBnSplits = 64;  // row splits of a column sum (partials [split][K][2])

struct BnSumArgs {
  int mode;  // 0: sum x | 1: sum (x - mean)^2 | 2: sums of g dh and g dh xhat (g = gamma or 1)
  const float* x; int ldx; const int* docids; int n_rows;  // modes 0 / 1: rows, gathered through docids if given
  const float* cstat;                                      // mode 1: [K][2], mean in slot 0
  const float* dh; int lddh; const float* go; const float* w;  // mode 2: dense dh, or (dh == NULL) the rank-1 go[m] w[k]
  const float* xhat; int ldxh; const float* gamma;             // mode 2 (gamma / w may be NULL: 1)
  int K; long long M; int rows_per_split;
  float* part;  // [gridDim.y][K][2]
};

__global__ void __launch_bounds__(256) bn_colsum_kernel(BnSumArgs p) {
  __shared__ float s_red[2][4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + c;
  const bool kok = k < p.K;
  const long long m_begin = (long long)blockIdx.y * p.rows_per_split;
  long long m_end = m_begin + p.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const float mean = (p.mode == 1 && kok) ? p.cstat[2 * k] : 0.f;
  const float g = (p.mode == 2 && kok && p.gamma) ? p.gamma[k] : 1.f;
  const float wk = (p.mode == 2 && kok && p.w) ? p.w[k] : 1.f;
  float a1 = 0.f, a2 = 0.f;
  for (long long m = m_begin + rl; m < m_end; m += 4) {
    if (p.mode <= 1) {
      float x = 0.f;
      if (kok) {
        if (p.docids) {
          const int d = p.docids[m];
          x = d < p.n_rows ? p.x[(long long)d * p.ldx + k] : 0.f;  // the pad row is a zero row OF the batch
        } else {
          x = p.x[m * p.ldx + k];
        }
      }
      if (p.mode == 0) a1 += x;
      else a1 = fmaf(x - mean, x - mean, a1);
    } else if (kok) {
      const float dh = p.dh ? p.dh[m * p.lddh + k] : p.go[m] * wk;
      const float gd = g * dh;
      a1 += gd;
      a2 = fmaf(gd, p.xhat[m * p.ldxh + k], a2);
    }
  }
  s_red[0][rl][c] = a1;
  s_red[1][rl][c] = a2;
  __syncthreads();
  if (rl == 0 && kok) {
    float* o = p.part + ((long long)blockIdx.y * p.K + k) * 2;
    o[0] = (s_red[0][0][c] + s_red[0][1][c]) + (s_red[0][2][c] + s_red[0][3][c]);
    o[1] = (s_red[1][0][c] + s_red[1][1][c]) + (s_red[1][2][c] + s_red[1][3][c]);
  }
}

struct BnFinishArgs {
  int mode;  // 0: mean | 1: rstd | 2: c1, c2 | 3: output layer (parameter gradients + c1, c2) | 4: evaluation statistics
  const float* part; int nsplit; int K; long long M; float eps;
  float* cstat;   // [K][2] (mean, rstd)            modes 0, 1, 4
  float* c12;     // [K][2]                         modes 2, 3
  const float* gamma; const float* beta; const float* w;  // mode 3
  float* outrow;  // mode 3: [3 K + 4] = dW, dgamma, dbeta, db of the output layer (one partial row for finalize)
};

__global__ void __launch_bounds__(256) bn_finish_kernel(BnFinishArgs p) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= p.K) return;
  if (p.mode == 4) {  // moving_mean = 0, moving_variance = 1: never updated by the reference's training step
    p.cstat[2 * k] = 0.f;
    p.cstat[2 * k + 1] = 1.f / sqrtf(1.f + p.eps);
    return;
  }
  float s1 = 0.f, s2 = 0.f;
  for (int s = 0; s < p.nsplit; ++s) {
    s1 += p.part[((long long)s * p.K + k) * 2];
    s2 += p.part[((long long)s * p.K + k) * 2 + 1];
  }
  const float inv_m = 1.f / (float)p.M;
  if (p.mode == 0) {
    p.cstat[2 * k] = s1 * inv_m;
  } else if (p.mode == 1) {
    p.cstat[2 * k + 1] = 1.f / sqrtf(s1 * inv_m + p.eps);  // biased variance (tf.nn.moments)
  } else if (p.mode == 2) {
    p.c12[2 * k] = s1 * inv_m;
    p.c12[2 * k + 1] = s2 * inv_m;
  } else {  // s1 = sum_m go (the same for every k), s2 = sum_m go xhat[m][k]
    const float g = p.gamma[k], b = p.beta[k], wk = p.w[k];
    p.outrow[k] = fmaf(g, s2, b * s1);   // dW_out = sum_m (gamma xhat + beta) go
    p.outrow[p.K + k] = wk * s2;         // dgamma
    p.outrow[2 * p.K + k] = wk * s1;     // dbeta
    if (k == 0) p.outrow[3 * p.K] = s1;  // db
    p.c12[2 * k] = g * wk * s1 * inv_m;
    p.c12[2 * k + 1] = g * wk * s2 * inv_m;
  }
}

// xhat = (x - mean) rstd ; a = gamma xhat + beta, both [M][ld] with zeros beyond K
struct BnApplyArgs {
  const float* x; int ldx; const int* docids; int n_rows;
  const float* cstat; const float* gamma; const float* beta;
  int K, ld; long long M;
  float* xhat; float* a;
};
__global__ void __launch_bounds__(256) bn_apply_kernel(BnApplyArgs p) {
  const long long total = p.M * p.ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / p.ld;
    const int k = (int)(i - m * p.ld);
    float xh = 0.f, a = 0.f;
    if (k < p.K) {
      float x;
      if (p.docids) {
        const int d = p.docids[m];
        x = d < p.n_rows ? p.x[(long long)d * p.ldx + k] : 0.f;
      } else {
        x = p.x[m * p.ldx + k];
      }
      xh = (x - p.cstat[2 * k]) * p.cstat[2 * k + 1];
      a = fmaf(xh, p.gamma[k], p.beta[k]);
    }
    p.xhat[i] = xh;
    p.a[i] = a;
  }
}

// dz_prev[m][k] = rstd_k (gamma_k dh - c1_k - xhat c2_k) * act'(act_prev[m][k]) ; dh dense (in place allowed) or rank-1
struct BnDxArgs {
  const float* dh; const float* go; const float* w;
  const float* xhat; const float* cstat; const float* c12; const float* gamma; const float* act_prev;
  int activation, K, ld; long long M;
  float* dz;
};
__global__ void __launch_bounds__(256) bn_dx_kernel(BnDxArgs p) {
  const long long total = p.M * p.ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / p.ld;
    const int k = (int)(i - m * p.ld);
    float v = 0.f;
    if (k < p.K) {
      const float dh = p.dh ? p.dh[i] : p.go[m] * p.w[k];
      const float dx = p.cstat[2 * k + 1] * (p.gamma[k] * dh - p.c12[2 * k] - p.xhat[i] * p.c12[2 * k + 1]);
      v = dx * act_grad_from_out(p.activation, p.act_prev[i]);
    }
    p.dz[i] = v;
  }
}

// ---- host helpers ---------------------------------------------------------------------------------------------------
static inline int bn_rows_per_split(long long M) { return (int)cdiv(M, kBnSplits); }
static inline int bn_nsplit(long long M) { return (int)cdiv(M, bn_rows_per_split(M)); }

static int bn_launch_sum(BnSumArgs a, hipStream_t st) {
  a.rows_per_split = bn_rows_per_split(a.M);
  {
    ProfScope prof_("bn_colsum_kernel", st);
    hipLaunchKernelGGL(bn_colsum_kernel, dim3((unsigned)cdiv(a.K, 64), (unsigned)bn_nsplit(a.M)), dim3(256), 0, st, a);
  }
  return check_launch("bn_colsum_kernel");
}
static int bn_launch_finish(BnFinishArgs f, hipStream_t st) {
  f.nsplit = bn_nsplit(f.M);
  {
    ProfScope prof_("bn_finish_kernel", st);
    hipLaunchKernelGGL(bn_finish_kernel, dim3((unsigned)cdiv(f.K, 256)), dim3(256), 0, st, f);
  }
  return check_launch("bn_finish_kernel");
}
static inline unsigned bn_ew_blocks(long long total) { return (unsigned)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192); }

// (mean, rstd) of the columns of a layer's input -> cstat, then xhat and a
static int bn_forward_input(const float* x, int ldx, const int* docids, int n_rows, int K, int ld, long long M, float eps,
                            bool training, const float* gamma, const float* beta, float* cstat, float* part, float* xhat,
                            float* a, hipStream_t st) {
  int rc;
  if (training) {
    BnSumArgs s{};
    s.mode = 0; s.x = x; s.ldx = ldx; s.docids = docids; s.n_rows = n_rows; s.K = K; s.M = M; s.part = part;
    if ((rc = bn_launch_sum(s, st))) return rc;
    BnFinishArgs f{};
    f.mode = 0; f.part = part; f.K = K; f.M = M; f.eps = eps; f.cstat = cstat;
    if ((rc = bn_launch_finish(f, st))) return rc;
    s.mode = 1; s.cstat = cstat;
    if ((rc = bn_launch_sum(s, st))) return rc;
    f.mode = 1;
    if ((rc = bn_launch_finish(f, st))) return rc;
  } else {
    BnFinishArgs f{};
    f.mode = 4; f.K = K; f.M = M; f.eps = eps; f.cstat = cstat;
    if ((rc = bn_launch_finish(f, st))) return rc;
  }
  BnApplyArgs ap{x, ldx, docids, n_rows, cstat, gamma, beta, K, ld, M, xhat, a};
  {
    ProfScope prof_("bn_apply_kernel", st);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_ew_blocks(M * ld)), dim3(256), 0, st, ap);
  }
  return check_launch("bn_apply_kernel");
}

}  // namespace ultr
