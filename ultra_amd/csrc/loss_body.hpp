// Device bodies of the loss kernels that other launches can carry (the DLA short-list loss rides with the first
// launch of the backward pass, ultr_loss_dla_deferred), plus the helpers every loss kernel shares.
// Reference: ultra/learning_algorithm/dla.py:119-180,257-306,343-364; base_algorithm.py:228-251,163-186.
#pragma once
#include "common.hpp"

namespace ultr {

constexpr int kLossThreads = 256;
constexpr int kMaxStat = 4 + 4 * ULTR_MAX_LIST;  // widest partial row (DLA / REM)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
__device__ __forceinline__ float sig_ce(float z, float x) { return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float eluf_(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float clip_w(float w, float maxw) { return maxw > 0.f ? fminf(fmaxf(w, 0.f), maxw) : w; }

// Per-block accumulation of batch sums (or maxima), fixed order and barrier-free until combine():
// add() reduces across the wave with shuffles and parks the result in the wave's LDS row.
struct BlockStats {
  float* s_part;  // [4 waves][kMaxStat]
  __device__ __forceinline__ void add(int slot, float v, bool is_max = false) {
    v = is_max ? wave_max(v) : wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[(threadIdx.x >> 6) * kMaxStat + slot] = v;
  }
  __device__ __forceinline__ void combine(float* s_acc, int ns, int max_from) {
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int s = threadIdx.x; s < ns; s += blockDim.x) {
      float r = s_part[s];
      for (int w = 1; w < nw; ++w) r = (s >= max_from) ? fmaxf(r, s_part[w * kMaxStat + s]) : r + s_part[w * kMaxStat + s];
      s_acc[s] = r;
    }
  }
};

// Publish this block's row, and in the last block reduce all rows into out[0..ns).  Slots >=
// max_from are reduced with max instead of +.  Returns true in the (single) last block, with the
// totals also left in s_acc.
__device__ __forceinline__ bool finish_stats(float* s_acc, int ns, int max_from, float* ws, float* out, unsigned int bid,
                                             unsigned int nb) {
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  float* rows = ws + 8;
  __syncthreads();
  for (int s = threadIdx.x; s < ns; s += blockDim.x) rows[(long long)bid * ns + s] = s_acc[s];
  const bool last = last_block_arrive(counter, nb);
  if (last) {
    __shared__ float s_tmp[kLossThreads];
    if (ns <= (int)blockDim.x) {
      reduce_rows(rows, (int)nb, ns, max_from, s_tmp, s_acc);
    } else {  // rows wider than the block (L = 64 with the widest statistics): plain column walk
      for (int s = threadIdx.x; s < ns; s += blockDim.x) {
        float r = rows[s];
        for (unsigned int b = 1; b < nb; ++b) {
          const float v = rows[(long long)b * ns + s];
          r = (s >= max_from) ? fmaxf(r, v) : r + v;
        }
        s_acc[s] = r;
      }
      __syncthreads();
    }
    for (int s = threadIdx.x; s < ns; s += blockDim.x) out[s] = s_acc[s];
    __syncthreads();
  }
  return last;
}
__device__ __forceinline__ bool finish_stats(float* s_acc, int ns, int max_from, float* ws, float* out) {
  return finish_stats(s_acc, ns, max_from, ws, out, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// DLA for short lists (L <= 16): one lane per (list, rank), a list = 16 consecutive lanes, so the
// per-list softmaxes are DPP group reductions and each lane evaluates ONE exp instead of L.
// stats: [0] rank numerator [1] rank normaliser [2] exam numerator [3] exam normaliser
//        [4+l] pw_l  [4+L+l] sum_b rw  [4+2L+l] column sums for the tower's gradient (scratch)
// ------------------------------------------------------------------------------------------
struct DlaArgs {
  const float* scores; const float* clicks; int B, L;
  const float* prop_params; int loss_func, prob_kind; float max_pw;
  float* d_scores; float* prop_grads; float* list_weights; float* stats; float* ws;
};

// The propensity tower at rank r = lane & 15 (dla.py:257-306; batch independent): logit_r = ELU(W[r] + b),
// its log-softmax, and the inverse propensity weight pw_r = p_0 / p_r (dla.py:352-364).
struct DlaTower { float pre, logit, logp, pw; };
__device__ __forceinline__ DlaTower dla_tower(const float* __restrict__ prop_params, int L, int r, int prob_kind,
                                              float max_pw) {
  DlaTower t;
  const bool in_list = r < L;
  t.pre = in_list ? prop_params[r] + prop_params[L] : 0.f;
  t.logit = eluf_(t.pre);
  const float tm = group_max<16>(in_list ? t.logit : -INFINITY);
  const float te = in_list ? expf(t.logit - tm) : 0.f;
  const float tse = group_sum<16>(te);
  t.logp = t.logit - tm - logf(tse);
  const float tmean = group_sum<16>(in_list ? t.logit : 0.f) / (float)L;
  const float tprob = prob_kind == ULTR_PROB_SOFTMAX ? te / tse : sigmoidf_(t.logit - tmean);
  const float tprob0 = group_sum<16>(r == 0 ? tprob : 0.f);
  t.pw = in_list ? clip_w(tprob0 / tprob, max_pw) : 0.f;
  return t;
}

// One (list, rank) lane of the rank loss (base_algorithm.py:228-251 / :163-186): the list's softmax pieces and
// ds = un-normalised d rank_loss / d score.  The 16 lanes of a group hold the 16 ranks of one list.  Shared by
// the loss kernel and by the backward pass's first launch, which evaluates ds in place of reading d_scores when
// it carries the loss (ultr_loss_dla_deferred): no implicit contraction, so both produce the same bits.
struct DlaLane { float m, ex, se, wl, S, lse, ds; };
__device__ __forceinline__ DlaLane dla_lane(float o, float c, float pw, bool live, int loss_func) {
#pragma clang fp contract(off)
  DlaLane v;
  v.m = group_max<16>(live ? o : -INFINITY);
  v.ex = live ? expf(o - v.m) : 0.f;
  v.se = group_sum<16>(v.ex);
  v.wl = live ? (c + 0.0000001f) * pw : 0.f;
  v.S = group_sum<16>(v.wl);
  v.lse = v.m + logf(v.se);
  if (loss_func == ULTR_LOSS_SOFTMAX) {
    v.ds = expf(o - v.lse) * v.S - v.wl;
  } else {
    v.ds = (sigmoidf_(o) - fminf(c, 1.f)) * pw;
  }
  return v;
}

// Block `bid` of `nb` (256 threads each) of the short-list DLA loss.  s_acc [kMaxStat], s_part [4 * kMaxStat],
// s_tower [4 * 16] are LDS.
__device__ __forceinline__ void dla_loss_small_body(const DlaArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part, float* s_tower) {
  const float* __restrict__ scores = a.scores;
  const float* __restrict__ clicks = a.clicks;
  const int B = a.B, L = a.L, loss_func = a.loss_func, prob_kind = a.prob_kind;
  const float max_pw = a.max_pw;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & 15, li = tid >> 4;
  const bool in_list = r < L;
  const DlaTower tw = dla_tower(a.prop_params, L, r, prob_kind, max_pw);
  const float pw = tw.pw, logit = tw.logit, logp = tw.logp;
  if (tid < 16) {
    s_tower[0 * 16 + tid] = tw.pre;
    s_tower[1 * 16 + tid] = tw.logit;
    s_tower[2 * 16 + tid] = tw.logp;
    s_tower[3 * 16 + tid] = tw.pw;
  }
  // The grid is capped (ultr_loss_workspace_floats): beyond 16 * nb lists a block walks several groups
  // of 16 lists and keeps its per-lane sums in registers, so the number of partial rows stays bounded.
  float rank_num = 0.f, rank_den = 0.f, exam_num = 0.f, exam_den = 0.f, col = 0.f, rw_sum = 0.f;
  for (long long b0 = (long long)bid * 16; b0 < B; b0 += (long long)nb * 16) {
    const long long b = b0 + li;
    const bool live = in_list && b < B;
    const long long e = (long long)r * B + (b < B ? b : 0);
    const float o = live ? scores[e] : 0.f, c = live ? clicks[e] : 0.f;
    const DlaLane v = dla_lane(o, c, pw, live, loss_func);
    // the list's relevance weights
    const float mean = group_sum<16>(live ? o : 0.f) / (float)L;
    const float pl = prob_kind == ULTR_PROB_SOFTMAX ? v.ex / v.se : sigmoidf_(o - mean);
    const float p0 = group_sum<16>(r == 0 ? pl : 0.f);
    const float rw = live ? clip_w(p0 / pl, max_pw) : 0.f;  // relevance weights (dla.py:159-160)
    const float cl = fminf(fmaxf(c, 0.f), 1.f);
    const float lw = group_sum<16>(live ? pw * cl : 0.f);
    rw_sum += rw;
    if (live) {
      a.d_scores[e] = v.ds;
      if (loss_func == ULTR_LOSS_SOFTMAX) {
        rank_num += -v.wl * (o - v.lse);
        const float wle = (c + 0.0000001f) * rw;
        exam_num += -wle * logp;
        exam_den += wle;
        col += wle;  // d exam/d logit_l = softmax(logit)_l * sum_b S'_b - sum_b wle
        rank_den += r == 0 ? v.S : 0.f;
      } else {
        const float z = fminf(c, 1.f);
        rank_num += sig_ce(z, o) * pw;
        exam_num += sig_ce(z, logit) * rw;
        col += (sigmoidf_(logit) - z) * rw;  // d exam/d logit_l directly
        rank_den += r == 0 ? 1.f : 0.f;
        exam_den += r == 0 ? 1.f : 0.f;
      }
      if (r == 0 && a.list_weights) a.list_weights[b] = lw / (float)L;
    }
  }
  const float rw = rw_sum;
  // batch sums: scalars over the whole wave; per-rank columns over the wave's 4 lists (xor 16, xor 32)
  rank_num = wave_sum(rank_num);
  rank_den = wave_sum(rank_den);
  exam_num = wave_sum(exam_num);
  exam_den = wave_sum(exam_den);
  float o1, o2;
  float rws = xor16_partner(rw, &o1);
  rws += o1;
  rws = xor32_partner(rws, &o2) + o2;
  float cols = xor16_partner(col, &o1);
  cols += o1;
  cols = xor32_partner(cols, &o2) + o2;
  float* mine = s_part + wid * kMaxStat;
  if (lane == 0) {
    mine[0] = rank_num;
    mine[1] = rank_den;
    mine[2] = exam_num;
    mine[3] = exam_den;
  }
  if (lane < L) {
    mine[4 + lane] = 0.f;  // pw_l is batch independent: written once by the last block
    mine[4 + L + lane] = rws;
    mine[4 + 2 * L + lane] = cols;
  }
  BlockStats bs{s_part};
  bs.combine(s_acc, 4 + 3 * L, 1 << 30);
  if (finish_stats(s_acc, 4 + 3 * L, 1 << 30, a.ws, a.stats, bid, nb)) {
    if (tid < L) a.stats[4 + tid] = s_tower[3 * 16 + tid];
    if (tid < 64) {  // chain through the tower's ELU: d pre_l = d logit_l * ELU'(pre_l); d b = sum_l d pre_l
      float dpre = 0.f;  // one rank per lane, the bias gradient is their wave sum
      if (tid < L) {
        const int l = tid;
        const float dlogit = loss_func == ULTR_LOSS_SOFTMAX ? expf(s_tower[2 * 16 + l]) * s_acc[3] - s_acc[4 + 2 * L + l]
                                                            : s_acc[4 + 2 * L + l];
        dpre = dlogit * (s_tower[0 * 16 + l] > 0.f ? 1.f : s_tower[1 * 16 + l] + 1.f);
        a.prop_grads[l] = dpre;
      }
      const float db = wave_sum(dpre);
      if (tid == 0) a.prop_grads[L] = db;
    }
  }
}

// A DLA loss request parked by ultr_loss_dla_deferred until the next ultr_dnn_backward of this host thread carries it.
struct PendingDla {
  bool on;
  DlaArgs a;
  unsigned nblocks;
};
extern thread_local PendingDla g_pending_dla;
int launch_dla_now(const DlaArgs& a, unsigned nblocks, hipStream_t st);  // loss.hip

}  // namespace ultr
