// Propensity-weighted losses, forward + backward fused, one thread per query list.
// Reference: ultra/learning_algorithm/base_algorithm.py:136-251 (losses),
//   dla.py:119-180,257-306,343-364 ; ipw_rank.py:110-143 ; regression_EM.py:116-176 ;
//   pairwise_debias.py:139-171 ; ultra/utils/propensity_estimator.py:33-42.
//
// Layout: scores / clicks / d_scores are position-major [L][B] (element (b,l) at l*B+b), so the
// 64 lanes of a wave read 64 consecutive floats of one rank: every access is coalesced.
// Batch sums: per-wave shuffles -> per-block LDS (fixed order) -> per-block partial row in the
// workspace -> the last block to arrive sums the rows in block order (deterministic).
#include "loss_body.hpp"

namespace ultr {


// ------------------------------------------------------------------------------------------
// IPW: softmax_loss / sigmoid_loss_on_list with per-element weights
// stats: [0] numerator  [1] normaliser  [2+l] max_b pw
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLossThreads) ipw_loss_kernel(const float* __restrict__ scores,
                                                                const float* __restrict__ clicks,
                                                                const float* __restrict__ pw, int B, int L,
                                                                int loss_func, float* __restrict__ d_scores,
                                                                float* __restrict__ stats,
                                                                float* __restrict__ list_weights, float* ws) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  BlockStats bs{s_part};
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = b < B;
  float num = 0.f, den = 0.f, lw = 0.f;
  if (loss_func == ULTR_LOSS_SOFTMAX) {
    float m = -INFINITY;
    for (int l = 0; l < L; ++l) m = fmaxf(m, on ? scores[(long long)l * B + b] : 0.f);
    float se = 0.f, S = 0.f;
    for (int l = 0; l < L; ++l) {
      if (!on) break;
      se += expf(scores[(long long)l * B + b] - m);
      S += (clicks[(long long)l * B + b] + 0.0000001f) * pw[(long long)l * B + b];
    }
    const float lse = m + logf(se);
    for (int l = 0; l < L; ++l) {
      if (!on) break;
      const long long e = (long long)l * B + b;
      const float o = scores[e], c = clicks[e], w = pw[e];
      const float wl = (c + 0.0000001f) * w;
      num -= wl * (o - lse);
      d_scores[e] = expf(o - lse) * S - wl;
      lw += w * fminf(fmaxf(c, 0.f), 1.f);
    }
    den = on ? S : 0.f;
  } else {
    for (int l = 0; l < L; ++l) {
      if (!on) break;
      const long long e = (long long)l * B + b;
      const float o = scores[e], c = clicks[e], w = pw[e];
      const float z = fminf(c, 1.f);
      num += sig_ce(z, o) * w;
      d_scores[e] = (sigmoidf_(o) - z) * w;
      lw += w * fminf(fmaxf(c, 0.f), 1.f);
    }
    den = on ? 1.f : 0.f;
  }
  if (on && list_weights) list_weights[b] = lw / (float)L;
  bs.add(0, num);
  bs.add(1, den);
  for (int l = 0; l < L; ++l) bs.add(2 + l, on ? pw[(long long)l * B + b] : -INFINITY, true);
  bs.combine(s_acc, 2 + L, 2);
  finish_stats(s_acc, 2 + L, 2, ws, stats);
}

// IPW for short lists (L <= 16): one lane per (list, rank), 16 lists per block; the per-list softmax is a group
// reduction over 16 lanes, the per-rank maxima meet by lane exchange.  Same arithmetic as ipw_loss_kernel.
__global__ void __launch_bounds__(kLossThreads) ipw_loss_small_kernel(IpwArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  ipw_loss_small_body(a, blockIdx.x, gridDim.x, s_acc, s_part);
}

constexpr int kDlaListsMinB = 32768;
constexpr int kDlaListsMaxBlocks = 512;  // one partial row per workgroup for the last one to add
// Many short lists (B >= kDlaListsMinB, L <= 16, softmax loss): one lane per list, body in loss_body.hpp.
template <int LT>
__global__ void __launch_bounds__(kLossThreads) ipw_loss_lists_kernel(IpwArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  ipw_loss_lists_body<LT>(a, blockIdx.x, gridDim.x, s_acc, s_part);
}
template <int LT>
static void launch_ipw_lists(const IpwArgs& a, unsigned nb, hipStream_t st) {
  if (a.L == LT) hipLaunchKernelGGL(ipw_loss_lists_kernel<LT>, dim3(nb), dim3(kLossThreads), 0, st, a);
  else if constexpr (LT > 1) launch_ipw_lists<LT - 1>(a, nb, st);
}

template <int LT>
__global__ void __launch_bounds__(kLossThreads) rem_loss_lists_kernel(RemArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  rem_loss_lists_body<LT>(a, blockIdx.x, gridDim.x, s_acc, s_part);
}
template <int LT>
static void launch_rem_lists(const RemArgs& a, unsigned nb, hipStream_t st) {
  if (a.L == LT) hipLaunchKernelGGL(rem_loss_lists_kernel<LT>, dim3(nb), dim3(kLossThreads), 0, st, a);
  else if constexpr (LT > 1) launch_rem_lists<LT - 1>(a, nb, st);
}

__global__ void ipw_weights_kernel(const float* __restrict__ clicks, int B, int L, const float* __restrict__ ipw,
                                   int n_ipw, float* __restrict__ pw) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)B * L) return;
  const int l = (int)(e / B);
  pw[e] = clicks[e] > 0.f ? ipw[l < n_ipw ? l : n_ipw - 1] : 0.f;
}

// ------------------------------------------------------------------------------------------
// DLA
// stats: [0] rank num [1] rank den [2] exam num [3] exam den [4+l] pw_l [4+L+l] sum_b rw
// internal partial row adds [4+2L+l] = column sums needed for d exam/d logit
// ------------------------------------------------------------------------------------------
template <int LT>  // compile-time bound on L: a list's scores and clicks live in registers
__global__ void __launch_bounds__(kLossThreads) dla_loss_kernel(
    const float* __restrict__ scores, const float* __restrict__ clicks, int B, int L,
    const float* __restrict__ prop_params, int loss_func, int prob_kind, float max_pw,
    float* __restrict__ d_scores, float* __restrict__ prop_grads, float* __restrict__ list_weights,
    float* __restrict__ stats, float* ws) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  __shared__ float s_logit[ULTR_MAX_LIST], s_pre[ULTR_MAX_LIST], s_prob[ULTR_MAX_LIST], s_pw[ULTR_MAX_LIST],
      s_logp[ULTR_MAX_LIST];
  BlockStats bs{s_part};
  const int tid = threadIdx.x;
  bs.clear(4 + 3 * L);
  // propensity tower: one-hot(position) @ W_0 + b_0 -> ELU  (dla.py:264-304): logit_l = ELU(W[l]+b)
  if (tid < L) {
    const float pre = prop_params[tid] + prop_params[L];
    s_pre[tid] = pre;
    s_logit[tid] = eluf_(pre);
  }
  __syncthreads();
  if (tid == 0) {
    float m = -INFINITY, mean = 0.f;
    for (int l = 0; l < L; ++l) {
      m = fmaxf(m, s_logit[l]);
      mean += s_logit[l];
    }
    mean /= (float)L;
    float se = 0.f;
    for (int l = 0; l < L; ++l) se += expf(s_logit[l] - m);
    for (int l = 0; l < L; ++l) {
      s_logp[l] = s_logit[l] - m - logf(se);
      s_prob[l] = prob_kind == ULTR_PROB_SOFTMAX ? expf(s_logit[l] - m) / se : sigmoidf_(s_logit[l] - mean);
    }
    for (int l = 0; l < L; ++l) s_pw[l] = clip_w(s_prob[0] / s_prob[l], max_pw);  // dla.py:352-364
  }
  __syncthreads();

  // The grid is capped (the last workgroup adds one partial row per workgroup): a thread walks lists b, b + grid, ...
  for (long long b0 = (long long)blockIdx.x * blockDim.x; b0 < B; b0 += (long long)gridDim.x * blockDim.x) {
  const long long b = b0 + tid;
  const bool on = b < B;
  // issue every global load of this list up front (independent loads, one round trip)
  float o[LT], c[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    o[l] = 0.f;
    c[l] = 0.f;
    if (l < L && on) {
      o[l] = scores[(long long)l * B + b];
      c[l] = clicks[(long long)l * B + b];
    }
  }
  float rank_num = 0.f, rank_den = 0.f, exam_num = 0.f, exam_den = 0.f, lw = 0.f;
  float m = -INFINITY, mean = 0.f, se = 0.f, S = 0.f;
#pragma unroll
  for (int l = 0; l < LT; ++l)
    if (l < L) {
      m = fmaxf(m, o[l]);
      mean += o[l];
    }
  mean /= (float)L;
  float ex[LT];  // exp(o_l - max): evaluated once, shared by the softmax, the relevance weights and the gradient
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    ex[l] = 0.f;
    if (l < L) {
      ex[l] = expf(o[l] - m);
      se += ex[l];
      S += (c[l] + 0.0000001f) * s_pw[l];
    }
  }
  const float inv_se = 1.f / se;
  const float p0 = prob_kind == ULTR_PROB_SOFTMAX ? ex[0] * inv_se : sigmoidf_(o[0] - mean);
  const float lse = m + logf(se);
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    if (l < L) {  // L is block-uniform: every thread reaches the wave reductions below
      float rw = 0.f, col = 0.f;
      if (on) {
        const long long e = (long long)l * B + b;
        const float pl = prob_kind == ULTR_PROB_SOFTMAX ? ex[l] * inv_se : sigmoidf_(o[l] - mean);
        rw = clip_w(p0 / pl, max_pw);  // relevance weights (dla.py:159-160)
        lw += s_pw[l] * fminf(fmaxf(c[l], 0.f), 1.f);
        if (loss_func == ULTR_LOSS_SOFTMAX) {
          const float wl = (c[l] + 0.0000001f) * s_pw[l];
          rank_num -= wl * (o[l] - lse);
          d_scores[e] = ex[l] * inv_se * S - wl;
          const float wle = (c[l] + 0.0000001f) * rw;
          exam_num -= wle * s_logp[l];
          exam_den += wle;
          col = wle;  // d exam/d logit_l = softmax(logit)_l * sum_b S'_b - sum_b wle
        } else {
          const float z = fminf(c[l], 1.f);
          rank_num += sig_ce(z, o[l]) * s_pw[l];
          d_scores[e] = (sigmoidf_(o[l]) - z) * s_pw[l];
          exam_num += sig_ce(z, s_logit[l]) * rw;
          col = (sigmoidf_(s_logit[l]) - z) * rw;  // d exam/d logit_l directly
        }
      }
      bs.accumulate(4 + L + l, rw);
      bs.accumulate(4 + 2 * L + l, col);
    }
  }
  if (on) {
    rank_den = loss_func == ULTR_LOSS_SOFTMAX ? S : 1.f;
    if (loss_func != ULTR_LOSS_SOFTMAX) exam_den = 1.f;
    if (list_weights) list_weights[b] = lw / (float)L;
  }
  bs.accumulate(0, rank_num);
  bs.accumulate(1, rank_den);
  bs.accumulate(2, exam_num);
  bs.accumulate(3, exam_den);
  }
  bs.combine(s_acc, 4 + 3 * L, 1 << 30);
  __syncthreads();
  if (tid < L) s_acc[4 + tid] = 0.f;  // pw_l is batch independent: written once by the last block
  if (finish_stats(s_acc, 4 + 3 * L, 1 << 30, ws, stats)) {
    if (tid < L) stats[4 + tid] = s_pw[tid];
    // chain through the tower's ELU: d pre_l = d logit_l * ELU'(pre_l); d b = sum_l d pre_l
    if (tid == 0) {
      float db = 0.f;
      for (int l = 0; l < L; ++l) {
        float dlogit;
        if (loss_func == ULTR_LOSS_SOFTMAX) {
          const float sm = expf(s_logp[l]);
          dlogit = sm * s_acc[3] - s_acc[4 + 2 * L + l];
        } else {
          dlogit = s_acc[4 + 2 * L + l];
        }
        const float dpre = dlogit * (s_pre[l] > 0.f ? 1.f : s_logit[l] + 1.f);
        prop_grads[l] = dpre;
        db += dpre;
      }
      prop_grads[L] = db;
    }
  }
}

// DLA for short lists (L <= 16): body in loss_body.hpp, shared with the backward pass's first launch, which can
// carry it (ultr_loss_dla_deferred).
__global__ void __launch_bounds__(kLossThreads) dla_loss_small_kernel(DlaArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  __shared__ float s_tower[4 * 16];  // pre, logit, logp, pw for the last block's chain rule
  dla_loss_small_body(a, blockIdx.x, gridDim.x, s_acc, s_part, s_tower);
}

// Many short lists (B >= kDlaListsMinB, L <= 16): one lane per list, body in loss_body.hpp.
template <int LT>
__global__ void __launch_bounds__(kLossThreads) dla_loss_lists_kernel(DlaArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  __shared__ float s_tower[4 * 16];
  dla_loss_lists_body<LT>(a, blockIdx.x, gridDim.x, s_acc, s_part, s_tower);
}
template <int LT>
static void launch_dla_lists(const DlaArgs& a, unsigned nb, hipStream_t st) {
  if (a.L == LT) hipLaunchKernelGGL(dla_loss_lists_kernel<LT>, dim3(nb), dim3(kLossThreads), 0, st, a);
  else if constexpr (LT > 1) launch_dla_lists<LT - 1>(a, nb, st);
}

// ------------------------------------------------------------------------------------------
// Regression-EM
// stats: [0] sum loss_b [1] B [2+l] M-step numerator [2+L+l] sum ranker_label [2+2L+l] sum clicks
//        [2+3L] sum of d_scores (gradient of sigmoid_prob_b, un-normalised)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLossThreads) rem_loss_kernel(
    const float* __restrict__ scores, const float* __restrict__ clicks, int B, int L,
    const float* __restrict__ propensity, const float* __restrict__ bias_ptr, const float* __restrict__ uniforms,
    unsigned long long seed, unsigned int step, const unsigned int* step_ctr, unsigned int slot_base,
    float* __restrict__ d_scores, float* __restrict__ ranker_labels, float* __restrict__ list_weights,
    float* __restrict__ stats, float* d_bias, float* ws) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  BlockStats bs{s_part};
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = b < B;
  const float bias = bias_ptr ? bias_ptr[0] : 0.f;
  if (step_ctr) step += step_ctr[0];  // device-resident step counter (read-only here), see feed.hip
  float loss = 0.f, lw = 0.f, dsum = 0.f;
  Philox4 rnd;
  for (int l = 0; l < L; ++l) {
    float mstep = 0.f, y = 0.f, c = 0.f;
    if (on) {
      const long long e = (long long)l * B + b;
      const float x = scores[e] + bias;  // train_output + sigmoid_prob_b (regression_EM.py:119)
      c = clicks[e];
      const float p = propensity[l];
      float u;
      if (uniforms) {
        u = uniforms[e];
      } else {
        if ((l & 3) == 0) rnd = philox4x32_10(slot_base + (uint32_t)b, 0x52454Du, (uint32_t)(l >> 2), step,
                                             (uint32_t)seed, (uint32_t)(seed >> 32));
        u = u24(philox_word(rnd, l));
      }
      const RemElem v = rem_elem_u(bias, p, scores[e], c, u);  // the one definition of the element (loss_body.hpp)
      mstep = c + (1.f - c) * v.p_e1_r0_c0;  // :153
      y = v.y;
      loss += sig_ce_fast(y, x);  // :159-165
      d_scores[e] = v.d;
      dsum += v.d;
      if (ranker_labels) ranker_labels[e] = y;
      lw += fminf(fmaxf(c, 0.f), 1.f) * (1.f / p);  // propensity_weights = 1/propensity (:176)
    }
    bs.add(2 + l, mstep);
    bs.add(2 + L + l, y);
    bs.add(2 + 2 * L + l, c);
  }
  if (on && list_weights) list_weights[b] = lw / (float)L;
  bs.add(0, loss);
  bs.add(1, on ? 1.f : 0.f);
  bs.add(2 + 3 * L, dsum);
  bs.combine(s_acc, 3 + 3 * L, 1 << 30);
  if (finish_stats(s_acc, 3 + 3 * L, 1 << 30, ws, stats)) {
    if (threadIdx.x == 0 && d_bias) d_bias[0] = s_acc[2 + 3 * L];  // un-normalised d loss / d sigmoid_prob_b
  }
}

// Regression-EM for short lists (L <= 16): one lane per (list, rank), 16 lists per block -- the batch-256 step then
// runs on 16 CUs with one pass of per-rank exchanges instead of one CU walking the ranks with 3 L wave reductions.
// Same arithmetic and the same Philox words per (list, rank) as rem_loss_kernel.
__global__ void __launch_bounds__(kLossThreads) rem_loss_small_kernel(RemArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  rem_loss_small_body(a, blockIdx.x, gridDim.x, s_acc, s_part);
}

__global__ void rem_update_kernel(float* propensity, int L, const float* stats, float s, float* old_out) {
  const int l = threadIdx.x;
  if (l < L && old_out) old_out[l] = propensity[l];
  if (l < L) propensity[l] = (1.f - s) * propensity[l] + s * (stats[2 + l] / stats[1]);  // regression_EM.py:151-155
}

// ------------------------------------------------------------------------------------------
// Pairwise debiasing.  The block's [L][256] tile of scores and clicks is staged in LDS; every
// (i, j) pair is then evaluated from LDS.  stats: [i*L+j] pair loss l_ij, [L*L] loss.
// ------------------------------------------------------------------------------------------
constexpr int kPdMaxL = 32;

__global__ void __launch_bounds__(kLossThreads) pd_loss_kernel(const float* __restrict__ scores,
                                                               const float* __restrict__ clicks, int B, int L,
                                                               const float* __restrict__ t_plus,
                                                               const float* __restrict__ t_minus,
                                                               float* __restrict__ d_scores, float* __restrict__ stats,
                                                               float* ws) {
  extern __shared__ float smem[];
  float* s_o = smem;                         // [L][256]
  float* s_c = smem + L * kLossThreads;      // [L][256]
  float* s_acc = s_c + L * kLossThreads;     // [L*L+1]
  float* s_wave = s_acc + (L * L + 1);       // [4][L*L]
  __shared__ float s_tp[kPdMaxL], s_tm[kPdMaxL];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int b = blockIdx.x * blockDim.x + tid;
  const bool on = b < B;
  for (int l = 0; l < L; ++l) {
    s_o[l * kLossThreads + tid] = on ? scores[(long long)l * B + b] : 0.f;
    s_c[l * kLossThreads + tid] = on ? clicks[(long long)l * B + b] : 0.f;
  }
  if (tid < L) {
    s_tp[tid] = t_plus[tid];
    s_tm[tid] = t_minus[tid];
  }
  for (int s = tid; s < 4 * L * L; s += blockDim.x) s_wave[s] = 0.f;
  __syncthreads();
  // pair losses: l_ij = sum_b min(1, relu(c_i - c_j)) * softplus(o_j - o_i)   (:147-155)
  for (int i = 0; i < L; ++i) {
    const float ci = s_c[i * kLossThreads + tid], oi = s_o[i * kLossThreads + tid];
    if (!__any(ci > 0.f)) continue;  // wave-uniform skip: no clicked doc at rank i in these 64 lists
    for (int j = 0; j < L; ++j) {
      if (j == i) continue;
      const float mask = fminf(1.f, fmaxf(ci - s_c[j * kLossThreads + tid], 0.f));
      float v = 0.f;
      if (mask > 0.f) {
        const float diff = s_o[j * kLossThreads + tid] - oi;
        v = mask * (fmaxf(diff, 0.f) + log1pf(expf(-fabsf(diff))));
      }
      v = wave_sum(v);
      if (lane == 0) s_wave[wid * L * L + i * L + j] = v;
    }
  }
  // gradient wrt o_k: -sum_j w_kj sig(o_j-o_k) + sum_i w_ik sig(o_k-o_i), w_ij = mask_ij/(t+_i t-_j)
  if (on) {
    for (int k = 0; k < L; ++k) {
      const float ck = s_c[k * kLossThreads + tid], ok = s_o[k * kLossThreads + tid];
      float acc = 0.f;
      for (int j = 0; j < L; ++j) {
        if (j == k) continue;
        const float cj = s_c[j * kLossThreads + tid];
        const float m1 = fminf(1.f, fmaxf(ck - cj, 0.f));
        const float m2 = fminf(1.f, fmaxf(cj - ck, 0.f));
        if (m1 > 0.f || m2 > 0.f) {
          const float sg = sigmoidf_(s_o[j * kLossThreads + tid] - ok);  // d softplus(o_j - o_k)
          acc -= m1 / (s_tp[k] * s_tm[j]) * sg;
          acc += m2 / (s_tp[j] * s_tm[k]) * (1.f - sg);  // sig(o_k - o_j)
        }
      }
      d_scores[(long long)k * B + b] = acc;
    }
  }
  __syncthreads();
  for (int s = tid; s < L * L; s += blockDim.x)
    s_acc[s] = s_wave[s] + s_wave[L * L + s] + s_wave[2 * L * L + s] + s_wave[3 * L * L + s];
  if (tid == 0) s_acc[L * L] = 0.f;
  if (finish_stats(s_acc, L * L + 1, 1 << 30, ws, stats)) {
    if (tid == 0) {  // loss = sum_ij l_ij / t+_i / t-_j  (:158-159), pre-update ratios
      float loss = 0.f;
      for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j)
          if (i != j) loss += s_acc[i * L + j] / s_tp[i] / s_tm[j];
      stats[L * L] = loss;
    }
  }
}

// Pairwise debiasing for short lists (L <= 16): one lane per (list, rank k), a list = 16 consecutive lanes, so
// the L partners of a document are shuffles within the group and the L x L pair terms of a list cost L steps
// instead of L^2 (the one-thread-per-list kernel above ran the whole headline batch on ONE CU: 85 us).  Lane (b, k)
// forms the pair losses l_kj and the gradient wrt o_k in the same walk over j.  Batch sums: the four lists of a
// wave meet by lane exchange, the waves in LDS, the blocks in the workspace (fixed order, last block reduces).
__global__ void __launch_bounds__(kLossThreads) pd_loss_small_kernel(PdArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  __shared__ float s_t[32];
  pd_loss_small_body(a, blockIdx.x, gridDim.x, s_acc, s_part, s_t);
}

// t+_i <- (1-s) t+_i + s (A_i/A_0)^(1/(p+1)), A_i = sum_j l_ij/t-_j ; t-_j likewise (:162-171)
__global__ void pd_update_kernel(float* t_plus, float* t_minus, int L, float* stats, float s, float reg_p,
                                 float* old_out) {
  __shared__ float A[kPdMaxL], Am[kPdMaxL], tp[kPdMaxL], tm[kPdMaxL];
  __shared__ float s_l[kPdMaxL * kPdMaxL];  // the pair matrix: one round of independent loads instead of L^2 serial ones
  const int k = threadIdx.x;
  for (int i = k; i < L * L; i += blockDim.x) s_l[i] = stats[i];
  if (k < L) {
    tp[k] = t_plus[k];
    tm[k] = t_minus[k];
    if (old_out) {
      old_out[k] = tp[k];
      old_out[L + k] = tm[k];
    }
  }
  __syncthreads();
  float row = 0.f;
  if (k < L) {
    float a = 0.f, am = 0.f;
    for (int j = 0; j < L; ++j) {
      if (j == k) continue;
      a += s_l[k * L + j] / tm[j];   // t_plus_loss_list[k]
      am += s_l[j * L + k] / tp[j];  // t_minus_loss_list[k]
      row += s_l[k * L + j] / tp[k] / tm[j];
    }
    A[k] = a;
    Am[k] = am;
  }
  __syncthreads();
  if (k < L) {
    const float e = 1.f / (reg_p + 1.f);
    t_plus[k] = (1.f - s) * tp[k] + s * powf(A[k] / A[0], e);
    t_minus[k] = (1.f - s) * tm[k] + s * powf(Am[k] / Am[0], e);
  }
  {  // loss with the all-reduced pair matrix (data-parallel runs recompute it here): row sums, then one wave sum
    const float loss = wave_sum(row);
    if (k == 0) {
      stats[L * L] = loss;
    }
  }
}

}  // namespace ultr

using namespace ultr;

namespace ultr {

static inline unsigned small_blocks(int B) { return (unsigned)(cdiv(B, 16) < 1024 ? cdiv(B, 16) : 1024); }

int launch_loss_now(const PendingLoss& p, hipStream_t st) {
  const char* name = p.kind == ULTR_RIDE_DLA ? "dla_loss_kernel" : p.kind == ULTR_RIDE_IPW ? "ipw_loss_kernel"
                     : p.kind == ULTR_RIDE_REM ? "rem_loss_kernel" : "pd_loss_kernel";
  {
    ultr::ProfScope prof_(name, st);
    if (p.kind == ULTR_RIDE_DLA) hipLaunchKernelGGL(dla_loss_small_kernel, dim3(p.nblocks), dim3(kLossThreads), 0, st, p.dla);
    else if (p.kind == ULTR_RIDE_IPW) hipLaunchKernelGGL(ipw_loss_small_kernel, dim3(p.nblocks), dim3(kLossThreads), 0, st, p.ipw);
    else if (p.kind == ULTR_RIDE_REM) hipLaunchKernelGGL(rem_loss_small_kernel, dim3(p.nblocks), dim3(kLossThreads), 0, st, p.rem);
    else hipLaunchKernelGGL(pd_loss_small_kernel, dim3(p.nblocks), dim3(kLossThreads), 0, st, p.pd);
  }
  return check_launch(name);
}
}  // namespace ultr

extern "C" int ultr_loss_flush(void* stream) {
  if (g_pending_loss.kind == ULTR_RIDE_NONE) return ULTR_OK;
  const PendingLoss p = g_pending_loss;
  g_pending_loss.kind = ULTR_RIDE_NONE;
  return launch_loss_now(p, as_stream(stream));
}

extern "C" int ultr_loss_cancel(void) {
  const int was = g_pending_loss.kind != ULTR_RIDE_NONE ? 1 : 0;
  g_pending_loss.kind = ULTR_RIDE_NONE;
  return was;
}

extern "C" long long ultr_loss_workspace_floats(int B, int L) {
  const long long blocks = cdiv(B > 0 ? B : 1, 16);  // the short-list DLA kernel runs 16 lists per block
  long long ns = 4 + 4 * (long long)L;
  if ((long long)L * L + 1 > ns) ns = (long long)L * L + 1;
  return 8 + blocks * ns + kArriveRowsFar;  // (+ room for the two-level arrival's sub-counters, common.hpp)
}

static int loss_args_ok(const char* who, const void* scores, const void* clicks, int B, int L, const void* ws) {
  ULTR_REQUIRE(scores && clicks && ws, ULTR_EINVAL, "%s: null pointer", who);
  ULTR_REQUIRE(B > 0 && L > 0, ULTR_EINVAL, "%s: bad sizes B=%d L=%d", who, B, L);
  ULTR_REQUIRE(L <= ULTR_MAX_LIST, ULTR_ELIMIT, "%s: L=%d > %d", who, L, ULTR_MAX_LIST);
  return ULTR_OK;
}

static int ipw_args(const char* who, IpwArgs* a, const float* scores, const float* clicks, const float* pw, int B, int L,
                    int loss_func, float* d_scores, float* stats, float* list_weights, float* ws) {
  int rc = loss_args_ok(who, scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(pw && d_scores && stats, ULTR_EINVAL, "%s: null pointer", who);
  ULTR_REQUIRE(loss_func == ULTR_LOSS_SOFTMAX || loss_func == ULTR_LOSS_SIGMOID, ULTR_EINVAL,
               "%s: loss_func=%d (pairwise_loss raises NameError in the reference and is not supported)", who, loss_func);
  *a = IpwArgs{scores, clicks, pw, B, L, loss_func, d_scores, stats, list_weights, ws, nullptr, 0, nullptr};
  return ULTR_OK;
}

extern "C" int ultr_loss_ipw_fwdbwd(const float* scores, const float* clicks, const float* pw, int B, int L,
                                    int loss_func, float* d_scores, float* stats, float* list_weights, float* ws,
                                    void* stream) {
  IpwArgs a;
  int rc = ipw_args("ultr_loss_ipw_fwdbwd", &a, scores, clicks, pw, B, L, loss_func, d_scores, stats, list_weights, ws);
  if (rc) return rc;
  rc = ultr_loss_flush(stream);  // a parked request nobody carried runs first: launch order is call order
  if (rc) return rc;
  if (L <= 16 && B >= kDlaListsMinB && B <= (1 << 30) && loss_func == ULTR_LOSS_SOFTMAX) {  // one lane per list
    const unsigned nb = (unsigned)(cdiv(B, kLossThreads) < kDlaListsMaxBlocks ? cdiv(B, kLossThreads) : kDlaListsMaxBlocks);
    {
      ultr::ProfScope prof_("ipw_loss_lists_kernel", as_stream(stream));
      launch_ipw_lists<16>(a, nb, as_stream(stream));
    }
    return check_launch("ipw_loss_lists_kernel");
  }
  if (L <= 16) {
    PendingLoss p{};
    p.kind = ULTR_RIDE_IPW; p.ipw = a; p.nblocks = small_blocks(B);
    return launch_loss_now(p, as_stream(stream));
  }
  {
    ultr::ProfScope prof_("ipw_loss_kernel", as_stream(stream));
    hipLaunchKernelGGL(ipw_loss_kernel, dim3((unsigned)cdiv(B, kLossThreads)), dim3(kLossThreads), 0, as_stream(stream),
                     scores, clicks, pw, B, L, loss_func, d_scores, stats, list_weights, ws);
  }
  return check_launch("ipw_loss_kernel");
}

extern "C" int ultr_loss_ipw_deferred(const float* scores, const float* clicks, const float* pw, int B, int L,
                                      int loss_func, float* d_scores, float* stats, float* list_weights, float* ws,
                                      void* stream) {
  if (L > 16)
    return ultr_loss_ipw_fwdbwd(scores, clicks, pw, B, L, loss_func, d_scores, stats, list_weights, ws, stream);
  IpwArgs a;
  int rc = ipw_args("ultr_loss_ipw_deferred", &a, scores, clicks, pw, B, L, loss_func, d_scores, stats, list_weights, ws);
  if (rc) return rc;
  rc = ultr_loss_flush(stream);
  if (rc) return rc;
  g_pending_loss.kind = ULTR_RIDE_IPW;
  g_pending_loss.ipw = a;
  g_pending_loss.nblocks = small_blocks(B);
  return ULTR_OK;
}

extern "C" int ultr_ipw_weights(const float* clicks, int B, int L, const float* ipw, int n_ipw, float* pw, void* stream);

extern "C" int ultr_loss_ipw_table_deferred(const float* scores, const float* clicks, const float* ipw, int n_ipw,
                                            float* pw_out, int B, int L, int loss_func, float* d_scores, float* stats,
                                            float* list_weights, float* ws, void* stream) {
  ULTR_REQUIRE(ipw && n_ipw > 0 && pw_out, ULTR_EINVAL, "ultr_loss_ipw_table_deferred: needs the IPW table and pw_out");
  if (L > 16) {  // long lists: the two launches of their own
    int rc = ultr_ipw_weights(clicks, B, L, ipw, n_ipw, pw_out, stream);
    if (rc) return rc;
    return ultr_loss_ipw_fwdbwd(scores, clicks, pw_out, B, L, loss_func, d_scores, stats, list_weights, ws, stream);
  }
  IpwArgs a;
  int rc = ipw_args("ultr_loss_ipw_table_deferred", &a, scores, clicks, pw_out, B, L, loss_func, d_scores, stats,
                    list_weights, ws);
  if (rc) return rc;
  a.ipw_table = ipw; a.n_ipw = n_ipw; a.pw_out = pw_out;
  rc = ultr_loss_flush(stream);
  if (rc) return rc;
  g_pending_loss.kind = ULTR_RIDE_IPW;
  g_pending_loss.ipw = a;
  g_pending_loss.nblocks = small_blocks(B);
  return ULTR_OK;
}

extern "C" int ultr_ipw_weights(const float* clicks, int B, int L, const float* ipw, int n_ipw, float* pw,
                                void* stream) {
  ULTR_REQUIRE(clicks && ipw && pw && B > 0 && L > 0 && n_ipw > 0, ULTR_EINVAL, "ultr_ipw_weights: bad argument");
  {
    ultr::ProfScope prof_("ipw_weights_kernel", as_stream(stream));
    hipLaunchKernelGGL(ipw_weights_kernel, dim3((unsigned)cdiv((long long)B * L, 256)), dim3(256), 0, as_stream(stream),
                     clicks, B, L, ipw, n_ipw, pw);
  }
  return check_launch("ipw_weights_kernel");
}


static int dla_args(const char* who, DlaArgs* a, const float* scores, const float* clicks, int B, int L,
                    const float* prop_params, int loss_func, int logits_to_prob, float max_propensity_weight,
                    float* d_scores, float* prop_grads, float* list_weights, float* stats, float* ws) {
  int rc = loss_args_ok(who, scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(prop_params && d_scores && prop_grads && stats, ULTR_EINVAL, "%s: null pointer", who);
  ULTR_REQUIRE(loss_func == ULTR_LOSS_SOFTMAX || loss_func == ULTR_LOSS_SIGMOID, ULTR_EINVAL,
               "%s: loss_func=%d not supported", who, loss_func);
  ULTR_REQUIRE(logits_to_prob == ULTR_PROB_SOFTMAX || logits_to_prob == ULTR_PROB_SIGMOID, ULTR_EINVAL,
               "%s: logits_to_prob=%d", who, logits_to_prob);
  *a = DlaArgs{scores, clicks, B, L, prop_params, loss_func, logits_to_prob, max_propensity_weight,
               d_scores, prop_grads, list_weights, stats, ws};
  return ULTR_OK;
}


extern "C" int ultr_loss_dla_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* prop_params,
                                    int loss_func, int logits_to_prob, float max_propensity_weight, float* d_scores,
                                    float* prop_grads, float* list_weights, float* stats, float* ws, void* stream) {
  DlaArgs a;
  int rc = dla_args("ultr_loss_dla_fwdbwd", &a, scores, clicks, B, L, prop_params, loss_func, logits_to_prob,
                    max_propensity_weight, d_scores, prop_grads, list_weights, stats, ws);
  if (rc) return rc;
  rc = ultr_loss_flush(stream);  // a parked request nobody carried runs first: launch order is call order
  if (rc) return rc;
  // Short lists: a 16-lane group per list (every lane busy: one exp per lane instead of L per thread); it is also the body
  // other launches can carry.  Very many of them: one lane per list, compiled for the exact list length.
  if (L <= 16 && B >= kDlaListsMinB && B <= (1 << 30) && a.loss_func == ULTR_LOSS_SOFTMAX && a.prob_kind == ULTR_PROB_SOFTMAX) {  // one lane per list (measured at 2^20 lists of 10: see profiles/*hbm_kernels.txt)
    const unsigned nb = (unsigned)(cdiv(B, kLossThreads) < kDlaListsMaxBlocks ? cdiv(B, kLossThreads) : kDlaListsMaxBlocks);
    {
      ultr::ProfScope prof_("dla_loss_lists_kernel", as_stream(stream));
      launch_dla_lists<16>(a, nb, as_stream(stream));
    }
    return check_launch("dla_loss_lists_kernel");
  }
  if (L <= 16) {
    PendingLoss p{};
    p.kind = ULTR_RIDE_DLA; p.dla = a; p.nblocks = small_blocks(B);
    return launch_loss_now(p, as_stream(stream));
  }
  const unsigned dla_blocks = (unsigned)(cdiv(B, kLossThreads) < 1024 ? cdiv(B, kLossThreads) : 1024);
  {
    ultr::ProfScope prof_("dla_loss_kernel", as_stream(stream));
    // (lists of at most 16 have returned above: only the long-list build of the list-per-thread kernel is used)
    hipLaunchKernelGGL(dla_loss_kernel<64>, dim3(dla_blocks), dim3(kLossThreads), 0,
                       as_stream(stream), scores, clicks, B, L, prop_params, loss_func, logits_to_prob,
                       max_propensity_weight, d_scores, prop_grads, list_weights, stats, ws);
  }
  return check_launch("dla_loss_kernel");
}

extern "C" int ultr_loss_dla_deferred(const float* scores, const float* clicks, int B, int L, const float* prop_params,
                                      int loss_func, int logits_to_prob, float max_propensity_weight, float* d_scores,
                                      float* prop_grads, float* list_weights, float* stats, float* ws, void* stream) {
  if (L > 16)  // only the short-list kernel has a body another launch can carry
    return ultr_loss_dla_fwdbwd(scores, clicks, B, L, prop_params, loss_func, logits_to_prob, max_propensity_weight,
                                d_scores, prop_grads, list_weights, stats, ws, stream);
  DlaArgs a;
  int rc = dla_args("ultr_loss_dla_deferred", &a, scores, clicks, B, L, prop_params, loss_func, logits_to_prob,
                    max_propensity_weight, d_scores, prop_grads, list_weights, stats, ws);
  if (rc) return rc;
  rc = ultr_loss_flush(stream);
  if (rc) return rc;
  g_pending_loss.kind = ULTR_RIDE_DLA;
  g_pending_loss.dla = a;
  g_pending_loss.nblocks = small_blocks(B);
  return ULTR_OK;
}

extern "C" int ultr_loss_rem_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* propensity,
                                    const float* sigmoid_prob_b, const float* uniforms, unsigned long long seed,
                                    unsigned int step, const unsigned int* step_ctr, unsigned int slot_base,
                                    float* d_scores, float* ranker_labels, float* list_weights,
                                    float* stats, float* d_bias, float* ws, void* stream) {
  int rc = loss_args_ok("ultr_loss_rem_fwdbwd", scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(propensity && d_scores && stats, ULTR_EINVAL, "ultr_loss_rem_fwdbwd: null pointer");
  rc = ultr_loss_flush(stream);  // a parked request nobody carried runs first: launch order is call order
  if (rc) return rc;
  if (L <= 16 && B >= kDlaListsMinB && B <= (1 << 30) && !uniforms) {  // very many short lists: one lane per list
    const RemArgs a{scores, clicks, B, L, propensity, sigmoid_prob_b, uniforms, seed, step, step_ctr, slot_base, d_scores,
                    ranker_labels, list_weights, stats, d_bias, ws};
    const unsigned nb = (unsigned)(cdiv(B, kLossThreads) < kDlaListsMaxBlocks ? cdiv(B, kLossThreads) : kDlaListsMaxBlocks);
    {
      ultr::ProfScope prof_("rem_loss_lists_kernel", as_stream(stream));
      launch_rem_lists<16>(a, nb, as_stream(stream));
    }
    return check_launch("rem_loss_lists_kernel");
  }
  if (L <= 16) {
    PendingLoss p{};
    p.kind = ULTR_RIDE_REM; p.nblocks = small_blocks(B);
    p.rem = RemArgs{scores, clicks, B, L, propensity, sigmoid_prob_b, uniforms, seed, step, step_ctr, slot_base, d_scores,
                    ranker_labels, list_weights, stats, d_bias, ws};
    return launch_loss_now(p, as_stream(stream));
  }
  {
    ultr::ProfScope prof_("rem_loss_kernel", as_stream(stream));
    hipLaunchKernelGGL(rem_loss_kernel, dim3((unsigned)cdiv(B, kLossThreads)), dim3(kLossThreads), 0, as_stream(stream),
                     scores, clicks, B, L, propensity, sigmoid_prob_b, uniforms, seed, step, step_ctr, slot_base, d_scores,
                     ranker_labels, list_weights, stats, d_bias, ws);
  }
  return check_launch("rem_loss_kernel");
}

extern "C" int ultr_loss_rem_deferred(const float* scores, const float* clicks, int B, int L, const float* propensity,
                                      const float* sigmoid_prob_b, const float* uniforms, unsigned long long seed,
                                      unsigned int step, const unsigned int* step_ctr, unsigned int slot_base,
                                      float* d_scores, float* ranker_labels, float* list_weights,
                                      float* stats, float* d_bias, float* ws, void* stream) {
  if (L > 16)
    return ultr_loss_rem_fwdbwd(scores, clicks, B, L, propensity, sigmoid_prob_b, uniforms, seed, step, step_ctr, slot_base,
                                d_scores, ranker_labels, list_weights, stats, d_bias, ws, stream);
  int rc = loss_args_ok("ultr_loss_rem_deferred", scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(propensity && d_scores && stats, ULTR_EINVAL, "ultr_loss_rem_deferred: null pointer");
  rc = ultr_loss_flush(stream);
  if (rc) return rc;
  g_pending_loss.kind = ULTR_RIDE_REM;
  g_pending_loss.rem = RemArgs{scores, clicks, B, L, propensity, sigmoid_prob_b, uniforms, seed, step, step_ctr, slot_base,
                               d_scores, ranker_labels, list_weights, stats, d_bias, ws};
  g_pending_loss.nblocks = small_blocks(B);
  return ULTR_OK;
}

namespace ultr {
__global__ void __launch_bounds__(64) em_update_kernel(EmUpdate e) {
  if (e.kind == ULTR_RIDE_REM) rem_update_body(e);
  else pd_update_body(e);
}
int launch_em_now(const EmUpdate& e, hipStream_t st) {
  {
    ProfScope prof_(e.kind == ULTR_RIDE_REM ? "rem_update_kernel" : "pd_update_kernel", st);
    hipLaunchKernelGGL(em_update_kernel, dim3(1), dim3(64), 0, st, e);
  }
  return check_launch("em_update_kernel");
}
}  // namespace ultr

extern "C" int ultr_em_update_flush(void* stream) {
  if (!g_pending_em.on) return ULTR_OK;
  g_pending_em.on = false;
  return launch_em_now(g_pending_em.e, as_stream(stream));
}

extern "C" int ultr_rem_update_propensity_deferred(float* propensity, int L, const float* stats, float em_step_size,
                                                   float* old_out, void* stream) {
  ULTR_REQUIRE(propensity && stats && L > 0 && L <= ULTR_MAX_LIST, ULTR_EINVAL, "ultr_rem_update_propensity_deferred: bad args");
  int rc = ultr_em_update_flush(stream);  // an earlier request nobody carried runs first
  if (rc) return rc;
  const EmUpdate e{ULTR_RIDE_REM, propensity, nullptr, L, const_cast<float*>(stats), em_step_size, 0.f, old_out};
  g_pending_em.on = true;
  g_pending_em.e = e;
  return ULTR_OK;
}

extern "C" int ultr_pd_update_ratios(float* t_plus, float* t_minus, int L, float* stats, float em_step_size,
                                     float regulation_p, float* old_out, void* stream);

extern "C" int ultr_rem_update_propensity(float* propensity, int L, const float* stats, float em_step_size,
                                          float* old_out, void* stream) {
  ULTR_REQUIRE(propensity && stats && L > 0 && L <= ULTR_MAX_LIST, ULTR_EINVAL, "ultr_rem_update_propensity: bad args");
  {
    ultr::ProfScope prof_("rem_update_kernel", as_stream(stream));
    hipLaunchKernelGGL(rem_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), propensity, L, stats, em_step_size, old_out);
  }
  return check_launch("rem_update_kernel");
}

// Many short lists (B >= kDlaListsMinB, L <= kPdListsMaxL): one lane per list, body in loss_body.hpp.
constexpr int kPdListsMaxL = 12;  // LT * LT pair sums per lane in registers
// OCC workgroups per CU asked of the register allocator: left alone it takes 286 registers for lists of 10 (one wave
// per SIMD, every transcendental chain exposed); held to 256 (two waves per SIMD, a few spilled row pointers) the kernel
// runs 72.8 -> 64.6 us at 2^20 lists but 16.7 -> 18.6 us at 65536, where a CU has one workgroup either way -- so OCC = 2
// from 131072 lists on, for lists of at most 10 (lists of 12 would spill 160 values).
template <int LT, int OCC>
__global__ void __launch_bounds__(kLossThreads, OCC) pd_loss_lists_kernel(PdArgs a) {
  __shared__ float s_acc[kMaxStat];
  __shared__ float s_part[4 * kMaxStat];
  __shared__ float s_t[32 + LT * LT];
  pd_loss_lists_body<LT>(a, blockIdx.x, gridDim.x, s_acc, s_part, s_t);
}
template <int LT>
static void launch_pd_lists(const PdArgs& a, unsigned nb, hipStream_t st) {
  if (a.L == LT) {
    if (LT <= 10 && a.B >= 131072) hipLaunchKernelGGL((pd_loss_lists_kernel<LT, (LT <= 10 ? 2 : 1)>), dim3(nb), dim3(kLossThreads), 0, st, a);
    else hipLaunchKernelGGL((pd_loss_lists_kernel<LT, 1>), dim3(nb), dim3(kLossThreads), 0, st, a);
  } else if constexpr (LT > 2) launch_pd_lists<LT - 1>(a, nb, st);
}

extern "C" int ultr_loss_pd_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* t_plus,
                                   const float* t_minus, float* d_scores, float* stats, float* ws, void* stream) {
  int rc = loss_args_ok("ultr_loss_pd_fwdbwd", scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(t_plus && t_minus && d_scores && stats, ULTR_EINVAL, "ultr_loss_pd_fwdbwd: null pointer");
  ULTR_REQUIRE(L <= kPdMaxL, ULTR_ELIMIT, "ultr_loss_pd_fwdbwd: L=%d > %d", L, kPdMaxL);
  rc = ultr_loss_flush(stream);  // a parked request nobody carried runs first: launch order is call order
  if (rc) return rc;
  if (L >= 2 && L <= kPdListsMaxL && B >= kDlaListsMinB && B <= (1 << 30)) {  // very many short lists: one lane per list
    const unsigned nb = (unsigned)(cdiv(B, kLossThreads) < kDlaListsMaxBlocks ? cdiv(B, kLossThreads) : kDlaListsMaxBlocks);
    const PdArgs a{scores, clicks, B, L, t_plus, t_minus, d_scores, stats, ws};
    {
      ultr::ProfScope prof_("pd_loss_lists_kernel", as_stream(stream));
      launch_pd_lists<kPdListsMaxL>(a, nb, as_stream(stream));
    }
    return check_launch("pd_loss_lists_kernel");
  }
  if (L <= 16) {
    PendingLoss p{};
    p.kind = ULTR_RIDE_PD; p.nblocks = small_blocks(B);
    p.pd = PdArgs{scores, clicks, B, L, t_plus, t_minus, d_scores, stats, ws};
    return launch_loss_now(p, as_stream(stream));
  }
  const size_t smem = sizeof(float) * ((size_t)2 * L * kLossThreads + (size_t)L * L + 1 + (size_t)4 * L * L);
  if (smem > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pd_loss_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  {
    ultr::ProfScope prof_("pd_loss_kernel", as_stream(stream));
    hipLaunchKernelGGL(pd_loss_kernel, dim3((unsigned)cdiv(B, kLossThreads)), dim3(kLossThreads), smem,
                     as_stream(stream), scores, clicks, B, L, t_plus, t_minus, d_scores, stats, ws);
  }
  return check_launch("pd_loss_kernel");
}

extern "C" int ultr_loss_pd_deferred(const float* scores, const float* clicks, int B, int L, const float* t_plus,
                                     const float* t_minus, float* d_scores, float* stats, float* ws, void* stream) {
  if (L > 16) return ultr_loss_pd_fwdbwd(scores, clicks, B, L, t_plus, t_minus, d_scores, stats, ws, stream);
  int rc = loss_args_ok("ultr_loss_pd_deferred", scores, clicks, B, L, ws);
  if (rc) return rc;
  ULTR_REQUIRE(t_plus && t_minus && d_scores && stats, ULTR_EINVAL, "ultr_loss_pd_deferred: null pointer");
  rc = ultr_loss_flush(stream);
  if (rc) return rc;
  g_pending_loss.kind = ULTR_RIDE_PD;
  g_pending_loss.pd = PdArgs{scores, clicks, B, L, t_plus, t_minus, d_scores, stats, ws};
  g_pending_loss.nblocks = small_blocks(B);
  return ULTR_OK;
}

extern "C" int ultr_pd_update_ratios_deferred(float* t_plus, float* t_minus, int L, float* stats, float em_step_size,
                                              float regulation_p, float* old_out, void* stream) {
  ULTR_REQUIRE(t_plus && t_minus && stats && L > 0 && L <= kPdMaxL, ULTR_EINVAL, "ultr_pd_update_ratios_deferred: bad args");
  int rc = ultr_em_update_flush(stream);
  if (rc) return rc;
  if (L > kEmMaxL)  // only the short-list body can ride
    return ultr_pd_update_ratios(t_plus, t_minus, L, stats, em_step_size, regulation_p, old_out, stream);
  g_pending_em.on = true;
  g_pending_em.e = EmUpdate{ULTR_RIDE_PD, t_plus, t_minus, L, stats, em_step_size, regulation_p, old_out};
  return ULTR_OK;
}

extern "C" int ultr_pd_update_ratios(float* t_plus, float* t_minus, int L, float* stats, float em_step_size,
                                     float regulation_p, float* old_out, void* stream) {
  ULTR_REQUIRE(t_plus && t_minus && stats && L > 0 && L <= kPdMaxL, ULTR_EINVAL, "ultr_pd_update_ratios: bad args");
  {
    ultr::ProfScope prof_("pd_update_kernel", as_stream(stream));
    hipLaunchKernelGGL(pd_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), t_plus, t_minus, L, stats,
                     em_step_size, regulation_p, old_out);
  }
  return check_launch("pd_update_kernel");
}
