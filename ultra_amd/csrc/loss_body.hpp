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
  // the same for kernels whose threads walk several lists: the wave's row accumulates (clear() first)
  __device__ __forceinline__ void clear(int ns) {
    for (int s = threadIdx.x & 63; s < ns; s += 64) s_part[(threadIdx.x >> 6) * kMaxStat + s] = 0.f;
  }
  __device__ __forceinline__ void accumulate(int slot, float v) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) s_part[(threadIdx.x >> 6) * kMaxStat + slot] += v;
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
  // The hand-off to the last block carries no cache maintenance: every word of the partial rows is written with an
  // agent-scope (sc1, write-through) store and read with an agent-scope load, the storing waves drain their stores
  // before the workgroup barrier that precedes the counter's atomic add, and the last block loads only after its own
  // add has returned.  The fenced form (buffer_wbl2 + buffer_inv) costs 3-4 us here, more while other workgroups of
  // the launch keep the L2 dirty.
  float* rows = arrive_rows(ws, nb);
  __syncthreads();
  for (int s = threadIdx.x; s < ns; s += blockDim.x) st_agent(rows + (long long)bid * ns + s, s_acc[s]);
  const bool last = last_block_arrive_ws(ws, bid, nb);
  if (last) {
    __shared__ float s_tmp[kLossThreads];
    if (ns <= (int)blockDim.x) {
      reduce_rows<true>(rows, (int)nb, ns, max_from, s_tmp, s_acc);
    } else {  // rows wider than the block (L = 64 with the widest statistics): plain column walk
      for (int s = threadIdx.x; s < ns; s += blockDim.x) {
        float r = ld_agent(rows + s);
        for (unsigned int b = 1; b < nb; ++b) {
          const float v = ld_agent(rows + (long long)b * ns + s);
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
// w_r, b: prop_params[r] (any value for r >= L) and prop_params[L], already in registers (a caller that requests them
// early keeps the two loads off its critical path)
__device__ __forceinline__ DlaTower dla_tower_regs(float w_r, float b, int L, int r, int prob_kind, float max_pw) {
  DlaTower t;
  const bool in_list = r < L;
  t.pre = in_list ? w_r + b : 0.f;
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
__device__ __forceinline__ DlaTower dla_tower(const float* __restrict__ prop_params, int L, int r, int prob_kind,
                                              float max_pw) {
  return dla_tower_regs(r < L ? prop_params[r] : 0.f, prop_params[L], L, r, prob_kind, max_pw);
}

// One (list, rank) lane of the rank loss (base_algorithm.py:228-251 / :163-186): the list's softmax pieces and
// ds = un-normalised d rank_loss / d score.  The 16 lanes of a group hold the 16 ranks of one list.  Shared by
// the loss kernel and by the backward pass's first launch, which evaluates ds in place of reading d_scores when
// it carries the loss (ultr_loss_dla_deferred): no implicit contraction, so both produce the same bits.
// The kernel is bound by its arithmetic, not its bytes (measured: ~200 instructions per wave and 4 lists), so the
// softmax runs on the hardware exponential / logarithm / reciprocal (v_exp_f32, v_log_f32, v_rcp_f32: 1 ulp each on
// arguments that are <= 0 after the max is taken off -- far inside the 1e-5 parity bound) and evaluates ONE exponential
// per element: softmax_l = ex_l / se serves the gradient, the probabilities and the relevance weights.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_log(float x) { return __logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// sig_ce with the hardware exp / log: for loss STATISTICS (absolute error ~1e-7 per element), never for a gradient
__device__ __forceinline__ float sig_ce_fast(float z, float x) { return fmaxf(x, 0.f) - x * z + fast_log(1.f + fast_exp(-fabsf(x))); }
struct DlaLane { float m, ex, se, wl, S, lse, ds, sm; };
__device__ __forceinline__ DlaLane dla_lane(float o, float c, float pw, bool live, int loss_func) {
#pragma clang fp contract(off)
  DlaLane v;
  v.m = group_max<16>(live ? o : -INFINITY);
  v.ex = live ? fast_exp(o - v.m) : 0.f;
  v.se = group_sum<16>(v.ex);
  v.wl = live ? (c + 0.0000001f) * pw : 0.f;
  v.S = group_sum<16>(v.wl);
  v.lse = v.m + fast_log(v.se);
  v.sm = v.ex * fast_rcp(v.se);  // softmax(o)_l
  if (loss_func == ULTR_LOSS_SOFTMAX) {
    v.ds = v.sm * v.S - v.wl;
  } else {
    v.ds = (sigmoidf_(o) - fminf(c, 1.f)) * pw;
  }
  return v;
}

// The waves' partial rows (s_part) -> the block's row -> the last block's totals, and there the chain through the
// propensity tower's ELU: d pre_l = d logit_l * ELU'(pre_l); d b = sum_l d pre_l.  s_tower = [pre | logit | logp | pw][16].
__device__ __forceinline__ void dla_stats_tail(const DlaArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                               float* s_part, const float* s_tower) {
  const int L = a.L, loss_func = a.loss_func, tid = threadIdx.x;
  BlockStats bs{s_part};
  bs.combine(s_acc, 4 + 3 * L, 1 << 30);
  if (finish_stats(s_acc, 4 + 3 * L, 1 << 30, a.ws, a.stats, bid, nb)) {
    if (tid < L) a.stats[4 + tid] = s_tower[3 * 16 + tid];
    if (tid < 64) {
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
    const float pl = prob_kind == ULTR_PROB_SOFTMAX ? v.sm : sigmoidf_(o - mean);
    const float p0 = group_sum<16>(r == 0 ? pl : 0.f);
    const float rw = live ? clip_w(p0 * fast_rcp(pl), max_pw) : 0.f;  // relevance weights (dla.py:159-160)
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
  dla_stats_tail(a, bid, nb, s_acc, s_part, s_tower);
}

// DLA for MANY short lists (exactly LT <= 16 ranks; softmax loss with softmax probabilities, the defaults: the other
// variants keep the 16-lane body): one lane per LIST, its LT scores and clicks in registers, so every
// load and store is a full 256-byte row per wave and the per-list softmaxes are plain register arithmetic (the 16-lane
// body above spends ~200 wave instructions per 4 lists and runs out of issue slots long before HBM bandwidth).  The
// per-rank batch sums stay in registers over the grid-stride walk and meet once at the end.  Same statistics row, same
// hardware exp / log / rcp as dla_lane; sums over a list run rank 0..LT-1 in order (the 16-lane body adds them as a
// butterfly), so results agree with it to rounding, not to the bit.
template <int LT>
__device__ __forceinline__ void dla_loss_lists_body(const DlaArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part, float* s_tower) {
#pragma clang fp contract(off)
  const float* __restrict__ scores = a.scores;
  const float* __restrict__ clicks = a.clicks;
  const int B = a.B;
  const float max_pw = a.max_pw;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const DlaTower tw = dla_tower(a.prop_params, LT, tid & 15, ULTR_PROB_SOFTMAX, max_pw);
  if (tid < 16) {
    s_tower[0 * 16 + tid] = tw.pre;
    s_tower[1 * 16 + tid] = tw.logit;
    s_tower[2 * 16 + tid] = tw.logp;
    s_tower[3 * 16 + tid] = tw.pw;
  }
  __syncthreads();
  float pw[LT], rws[LT], col[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    pw[l] = s_tower[3 * 16 + l];
    rws[l] = 0.f;
    col[l] = 0.f;
  }
  float rank_num = 0.f, rank_den = 0.f, exam_num = 0.f, exam_den = 0.f;
  // Addresses: a uniform row base (rank l) plus one 32-bit byte offset per lane (the list), so the 3 LT accesses of a
  // list share one address register (B <= 2^30, checked by the launcher).
  auto at = [](const float* row, unsigned int off) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + off); };
  auto at_w = [](float* row, unsigned int off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(row) + off); };
#pragma unroll 1
  for (unsigned int b = bid * kLossThreads + tid; b < (unsigned int)B; b += nb * kLossThreads) {
    const unsigned int off = b * 4u;
    float o[LT], c[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {  // every load of the list up front: one round trip
      o[l] = *at(scores + (long long)l * B, off);
      c[l] = *at(clicks + (long long)l * B, off);
    }
    float m = o[0];
#pragma unroll
    for (int l = 1; l < LT; ++l) m = fmaxf(m, o[l]);
    float ex[LT], se = 0.f, S = 0.f, lw = 0.f;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      ex[l] = fast_exp(o[l] - m);
      se += ex[l];
      S += (c[l] + 0.0000001f) * pw[l];
      lw += pw[l] * fminf(fmaxf(c[l], 0.f), 1.f);
    }
    const float rse = fast_rcp(se), lse = m + fast_log(se);
    const float p0 = ex[0] * rse;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const float sm = ex[l] * rse;
      const float rw = clip_w(p0 * fast_rcp(sm), max_pw);  // relevance weights (dla.py:159-160)
      rws[l] += rw;
      const float wl = (c[l] + 0.0000001f) * pw[l];
      rank_num += -wl * (o[l] - lse);
      const float wle = (c[l] + 0.0000001f) * rw;
      exam_num += -wle * s_tower[2 * 16 + l];
      exam_den += wle;
      col[l] += wle;
      *at_w(a.d_scores + (long long)l * B, off) = sm * S - wl;
    }
    rank_den += S;
    if (a.list_weights) a.list_weights[b] = lw / (float)LT;
  }
  float* mine = s_part + wid * kMaxStat;
  rank_num = wave_sum(rank_num);
  rank_den = wave_sum(rank_den);
  exam_num = wave_sum(exam_num);
  exam_den = wave_sum(exam_den);
  if (lane == 0) {
    mine[0] = rank_num;
    mine[1] = rank_den;
    mine[2] = exam_num;
    mine[3] = exam_den;
  }
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const float r = wave_sum(rws[l]), cl = wave_sum(col[l]);
    if (lane == 0) {
      mine[4 + l] = 0.f;  // pw_l is batch independent: written once by the last block
      mine[4 + LT + l] = r;
      mine[4 + 2 * LT + l] = cl;
    }
  }
  dla_stats_tail(a, bid, nb, s_acc, s_part, s_tower);
}

// ------------------------------------------------------------------------------------------
// IPW for short lists (L <= 16): softmax_loss / sigmoid_loss_on_list with per-element weights (base_algorithm.py
// :228-251, :163-186); also the naive algorithm's loss (weights of one).  Same lane layout as the DLA kernel, and the
// same rank-loss lane function.
// stats: [0] numerator  [1] normaliser  [2+l] max_b pw
// ------------------------------------------------------------------------------------------
struct IpwArgs {
  const float* scores; const float* clicks; const float* pw; int B, L, loss_func;
  float* d_scores; float* stats; float* list_weights; float* ws;
  // ipw_table != NULL (ultr_loss_ipw_table_deferred): the weights are formed here from the clicks and the estimator's
  // table (propensity_estimator.py:33-42) instead of being read from pw, and written to pw_out for whoever wants them
  const float* ipw_table; int n_ipw; float* pw_out;
};
// weight of element (rank r, click c): supplied, or IPW_list[min(r, n - 1)] where clicked
__device__ __forceinline__ float ipw_weight(const IpwArgs& a, long long e, int r, float c) {
  if (!a.ipw_table) return a.pw[e];
  return c > 0.f ? a.ipw_table[r < a.n_ipw ? r : a.n_ipw - 1] : 0.f;
}

__device__ __forceinline__ void ipw_loss_small_body(const IpwArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part) {
  const int B = a.B, L = a.L, loss_func = a.loss_func;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & 15, li = tid >> 4;
  const bool in_list = r < L;
  float num = 0.f, den = 0.f, wmax = -INFINITY;
  for (long long b0 = (long long)bid * 16; b0 < B; b0 += (long long)nb * 16) {
    const long long b = b0 + li;
    const bool live = in_list && b < B;
    const long long e = (long long)r * B + (b < B ? b : 0);
    const float o = live ? a.scores[e] : 0.f, c = live ? a.clicks[e] : 0.f, w = live ? ipw_weight(a, e, r, c) : 0.f;
    if (live) wmax = fmaxf(wmax, w);
    if (live && a.pw_out) a.pw_out[e] = w;
    const float lw = group_sum<16>(live ? w * fminf(fmaxf(c, 0.f), 1.f) : 0.f);
    const DlaLane v = dla_lane(o, c, w, live, loss_func);
    if (live) {
      a.d_scores[e] = v.ds;
      if (loss_func == ULTR_LOSS_SOFTMAX) {
        num -= v.wl * (o - v.lse);
        den += r == 0 ? v.S : 0.f;
      } else {
        num += sig_ce(fminf(c, 1.f), o) * w;
        den += r == 0 ? 1.f : 0.f;
      }
      if (r == 0 && a.list_weights) a.list_weights[b] = lw / (float)L;
    }
  }
  num = wave_sum(num);
  den = wave_sum(den);
  float o1, o2;
  wmax = xor16_partner(wmax, &o1);
  wmax = fmaxf(wmax, o1);
  wmax = xor32_partner(wmax, &o2);
  wmax = fmaxf(wmax, o2);
  float* mine = s_part + wid * kMaxStat;
  if (lane == 0) {
    mine[0] = num;
    mine[1] = den;
  }
  if (lane < L) mine[2 + lane] = wmax;
  BlockStats bs{s_part};
  bs.combine(s_acc, 2 + L, 2);
  finish_stats(s_acc, 2 + L, 2, a.ws, a.stats, bid, nb);
}

// IPW / naive softmax loss for MANY short lists (exactly LT <= 16 ranks): one lane per list, as dla_loss_lists_body.
template <int LT>
__device__ __forceinline__ void ipw_loss_lists_body(const IpwArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part) {
#pragma clang fp contract(off)
  const int B = a.B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float num = 0.f, den = 0.f, wmax[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) wmax[l] = -INFINITY;
  auto at = [](const float* row, unsigned int off) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + off); };
  auto at_w = [](float* row, unsigned int off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(row) + off); };
#pragma unroll 1
  for (unsigned int b = bid * kLossThreads + tid; b < (unsigned int)B; b += nb * kLossThreads) {
    const unsigned int off = b * 4u;
    float o[LT], c[LT], w[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      o[l] = *at(a.scores + (long long)l * B, off);
      c[l] = *at(a.clicks + (long long)l * B, off);
      if (!a.ipw_table) w[l] = *at(a.pw + (long long)l * B, off);
    }
    if (a.ipw_table) {
#pragma unroll
      for (int l = 0; l < LT; ++l) w[l] = c[l] > 0.f ? a.ipw_table[l < a.n_ipw ? l : a.n_ipw - 1] : 0.f;
    }
    float m = o[0];
#pragma unroll
    for (int l = 1; l < LT; ++l) m = fmaxf(m, o[l]);
    float ex[LT], se = 0.f, S = 0.f, lw = 0.f;
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      wmax[l] = fmaxf(wmax[l], w[l]);
      if (a.pw_out) *at_w(a.pw_out + (long long)l * B, off) = w[l];
      ex[l] = fast_exp(o[l] - m);
      se += ex[l];
      S += (c[l] + 0.0000001f) * w[l];
      lw += w[l] * fminf(fmaxf(c[l], 0.f), 1.f);
    }
    const float rse = fast_rcp(se), lse = m + fast_log(se);
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const float wl = (c[l] + 0.0000001f) * w[l];
      num -= wl * (o[l] - lse);
      *at_w(a.d_scores + (long long)l * B, off) = ex[l] * rse * S - wl;
    }
    den += S;
    if (a.list_weights) a.list_weights[b] = lw / (float)LT;
  }
  num = wave_sum(num);
  den = wave_sum(den);
  float* mine = s_part + wid * kMaxStat;
  if (lane == 0) {
    mine[0] = num;
    mine[1] = den;
  }
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const float mx = wave_max(wmax[l]);
    if (lane == 0) mine[2 + l] = mx;
  }
  BlockStats bs{s_part};
  bs.combine(s_acc, 2 + LT, 2);
  finish_stats(s_acc, 2 + LT, 2, a.ws, a.stats, bid, nb);
}

// ------------------------------------------------------------------------------------------
// Regression-EM for short lists (regression_EM.py:116-176)
// stats: [0] sum loss_b [1] B [2+l] M-step numerator [2+L+l] sum ranker_label [2+2L+l] sum clicks
//        [2+3L] sum of d_scores (gradient of sigmoid_prob_b, un-normalised)
// ------------------------------------------------------------------------------------------
struct RemArgs {
  const float* scores; const float* clicks; int B, L;
  const float* propensity; const float* bias_ptr; const float* uniforms;
  unsigned long long seed; unsigned int step; const unsigned int* step_ctr; unsigned int slot_base;
  float* d_scores; float* ranker_labels; float* list_weights; float* stats; float* d_bias; float* ws;
};

// One (list b, rank r) element: E-step posteriors, the Bernoulli ranker label and d loss / d score = sigmoid - label.
// Elementwise, so the backward pass's first launch can evaluate it for its own rows (ultr_loss_rem_deferred).
struct RemElem { float x, g, y, d, p_e1_r0_c0; };
// rem_elem_u: the arithmetic, given the element's uniform draw.  ONE definition for every kernel that evaluates it (the
// 16-lane and lane-per-list bodies, the long-list kernel, the riders of the fused step), so that the same inputs give
// the same label everywhere.  The sigmoid and the shared 1 / (1 - p g) take the hardware exp and reciprocal (1 ulp each,
// like dla_lane): with libm's expf and three IEEE divides an element cost ~290 instructions and the kernel ran at a
// fifth of the memory rate; a label can differ from the float64 oracle's only where p_r1 - u is within rounding of 0,
// which was so before (test_rem_loss_kernel bounds the rate).
__device__ __forceinline__ RemElem rem_elem_u(float bias, float p, float score, float c, float u) {
#pragma clang fp contract(off)
  RemElem v;
  v.x = score + bias;  // train_output + sigmoid_prob_b (regression_EM.py:119)
  v.g = fast_rcp(1.f + fast_exp(-v.x));
  const float rden = fast_rcp(1.f - p * v.g);
  v.p_e1_r0_c0 = p * (1.f - v.g) * rden;              // :140-141
  const float p_e0_r1_c0 = (1.f - p) * v.g * rden;    // :142-143
  const float p_r1 = c + (1.f - c) * p_e0_r1_c0;     // :147-148
  v.y = ceilf(p_r1 - u);      // get_bernoulli_sample (:39)
  v.y = v.y > 0.f ? v.y : 0.f;  // ceil(-0.x) = -0 -> 0
  v.d = v.g - v.y;
  return v;
}
// rem_elem_v takes the element's score and click by value (the fused training step has them on chip) and finds its draw.
__device__ __forceinline__ RemElem rem_elem_v(const RemArgs& a, float bias, float p, unsigned int step, long long b, int r,
                                              long long e, float score, float c) {
  float u;
  if (a.uniforms) {
    u = a.uniforms[e];
  } else {
    const Philox4 rnd = philox4x32_10(a.slot_base + (uint32_t)b, 0x52454Du, (uint32_t)(r >> 2), step, (uint32_t)a.seed,
                                      (uint32_t)(a.seed >> 32));
    u = u24(philox_word(rnd, r));
  }
  return rem_elem_u(bias, p, score, c, u);
}
__device__ __forceinline__ RemElem rem_elem(const RemArgs& a, float bias, float p, unsigned int step, long long b, int r,
                                            long long e) {
  return rem_elem_v(a, bias, p, step, b, r, e, a.scores[e], a.clicks[e]);
}

__device__ __forceinline__ void rem_loss_small_body(const RemArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part) {
  const int B = a.B, L = a.L;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & 15, li = tid >> 4;
  const bool in_list = r < L;
  const float bias = a.bias_ptr ? a.bias_ptr[0] : 0.f;
  const float p = in_list ? a.propensity[r] : 1.f;
  const float rp = 1.f / p;
  const unsigned int step = a.step + (a.step_ctr ? a.step_ctr[0] : 0u);  // device-resident step counter (read-only here)
  float loss = 0.f, cnt = 0.f, dsum = 0.f, ms = 0.f, ys = 0.f, cs = 0.f;
  for (long long b0 = (long long)bid * 16; b0 < B; b0 += (long long)nb * 16) {
    const long long b = b0 + li;
    const bool live = in_list && b < B;
    const long long e = (long long)r * B + (b < B ? b : 0);
    float lwt = 0.f;
    if (live) {
      const RemElem v = rem_elem(a, bias, p, step, b, r, e);
      const float c = a.clicks[e];
      loss += sig_ce_fast(v.y, v.x);  // :159-165
      a.d_scores[e] = v.d;
      dsum += v.d;
      if (a.ranker_labels) a.ranker_labels[e] = v.y;
      ms += c + (1.f - c) * v.p_e1_r0_c0;  // :153
      ys += v.y;
      cs += c;
      lwt = fminf(fmaxf(c, 0.f), 1.f) * rp;  // propensity_weights = 1/propensity (:176)
      cnt += r == 0 ? 1.f : 0.f;
    }
    const float lw = group_sum<16>(lwt);
    if (live && r == 0 && a.list_weights) a.list_weights[b] = lw / (float)L;
  }
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  dsum = wave_sum(dsum);
  float o1, o2;
  ms = xor16_partner(ms, &o1); ms += o1; ms = xor32_partner(ms, &o2) + o2;
  ys = xor16_partner(ys, &o1); ys += o1; ys = xor32_partner(ys, &o2) + o2;
  cs = xor16_partner(cs, &o1); cs += o1; cs = xor32_partner(cs, &o2) + o2;
  float* mine = s_part + wid * kMaxStat;
  if (lane == 0) {
    mine[0] = loss;
    mine[1] = cnt;
    mine[2 + 3 * L] = dsum;
  }
  if (lane < L) {
    mine[2 + lane] = ms;
    mine[2 + L + lane] = ys;
    mine[2 + 2 * L + lane] = cs;
  }
  BlockStats bs{s_part};
  bs.combine(s_acc, 3 + 3 * L, 1 << 30);
  if (finish_stats(s_acc, 3 + 3 * L, 1 << 30, a.ws, a.stats, bid, nb)) {
    if (threadIdx.x == 0 && a.d_bias) a.d_bias[0] = s_acc[2 + 3 * L];  // un-normalised d loss / d sigmoid_prob_b
  }
}

// Regression-EM for MANY short lists (exactly LT <= 16 ranks, Philox draws): one lane per list, as dla_loss_lists_body;
// one Philox block serves four ranks.  Same draws and element arithmetic as rem_loss_small_body.
template <int LT>
__device__ __forceinline__ void rem_loss_lists_body(const RemArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                    float* s_part) {
  const int B = a.B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float bias = a.bias_ptr ? a.bias_ptr[0] : 0.f;
  const unsigned int step = a.step + (a.step_ctr ? a.step_ctr[0] : 0u);
  float p[LT], rp[LT], ms[LT], ys[LT], cs[LT];
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    p[l] = a.propensity[l];
    rp[l] = 1.f / p[l];
    ms[l] = ys[l] = cs[l] = 0.f;
  }
  float loss = 0.f, cnt = 0.f, dsum = 0.f;
  auto at = [](const float* row, unsigned int off) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + off); };
  auto at_w = [](float* row, unsigned int off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(row) + off); };
#pragma unroll 1
  for (unsigned int b = bid * kLossThreads + tid; b < (unsigned int)B; b += nb * kLossThreads) {
    const unsigned int off = b * 4u;
    float o[LT], c[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      o[l] = *at(a.scores + (long long)l * B, off);
      c[l] = *at(a.clicks + (long long)l * B, off);
    }
    float lwt = 0.f;
    Philox4 rnd{};
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      if ((l & 3) == 0)
        rnd = philox4x32_10(a.slot_base + b, 0x52454Du, (uint32_t)(l >> 2), step, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
      const uint32_t word = (l & 3) == 0 ? rnd.w0 : (l & 3) == 1 ? rnd.w1 : (l & 3) == 2 ? rnd.w2 : rnd.w3;
      const RemElem v = rem_elem_u(bias, p[l], o[l], c[l], u24(word));
      loss += sig_ce_fast(v.y, v.x);  // :159-165
      *at_w(a.d_scores + (long long)l * B, off) = v.d;
      dsum += v.d;
      if (a.ranker_labels) *at_w(a.ranker_labels + (long long)l * B, off) = v.y;
      ms[l] += c[l] + (1.f - c[l]) * v.p_e1_r0_c0;  // :153
      ys[l] += v.y;
      cs[l] += c[l];
      lwt += fminf(fmaxf(c[l], 0.f), 1.f) * rp[l];  // propensity_weights = 1/propensity (:176)
    }
    cnt += 1.f;
    if (a.list_weights) a.list_weights[b] = lwt / (float)LT;
  }
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  dsum = wave_sum(dsum);
  float* mine = s_part + wid * kMaxStat;
  if (lane == 0) {
    mine[0] = loss;
    mine[1] = cnt;
    mine[2 + 3 * LT] = dsum;
  }
#pragma unroll
  for (int l = 0; l < LT; ++l) {
    const float m1 = wave_sum(ms[l]), y1 = wave_sum(ys[l]), c1 = wave_sum(cs[l]);
    if (lane == 0) {
      mine[2 + l] = m1;
      mine[2 + LT + l] = y1;
      mine[2 + 2 * LT + l] = c1;
    }
  }
  BlockStats bs{s_part};
  bs.combine(s_acc, 3 + 3 * LT, 1 << 30);
  if (finish_stats(s_acc, 3 + 3 * LT, 1 << 30, a.ws, a.stats, bid, nb)) {
    if (threadIdx.x == 0 && a.d_bias) a.d_bias[0] = s_acc[2 + 3 * LT];  // un-normalised d loss / d sigmoid_prob_b
  }
}

// ------------------------------------------------------------------------------------------
// Pairwise debiasing for short lists (pairwise_debias.py:139-171, pair CE base_algorithm.py:151-161)
// stats: [0..L*L) pair-loss matrix l_ij (sum over lists)  [L*L] loss
// ------------------------------------------------------------------------------------------
struct PdArgs {
  const float* scores; const float* clicks; int B, L;
  const float* t_plus; const float* t_minus;
  float* d_scores; float* stats; float* ws;
};

// Lane (list, rank k) against the other ranks j of its 16-lane group: d loss / d score_k, and (WITH_PL) the pair
// losses l_kj added into pl[j].  Every lane of the wave must call it (group shuffles).
template <bool WITH_PL>
__device__ __forceinline__ float pd_lane(float o, float c, float tp_k, float tm_k, bool live, int r, int L, int gbase,
                                         float (&pl)[16]) {
#pragma clang fp contract(off)
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (j < L) {  // uniform
      const float oj = __shfl(o, gbase + j, 64), cj = __shfl(c, gbase + j, 64);
      const float tpj = __shfl(tp_k, gbase + j, 64), tmj = __shfl(tm_k, gbase + j, 64);
      const float m1 = fminf(1.f, fmaxf(c - cj, 0.f));  // min(1, relu(c_k - c_j))  (:147-150)
      const float m2 = fminf(1.f, fmaxf(cj - c, 0.f));
      if (live && j != r && (m1 > 0.f || m2 > 0.f)) {
        // one exponential serves both: softplus(d) = max(d, 0) + log1p(e), sigmoid(d) = (d >= 0 ? 1 : e) / (1 + e).
        // Hardware exp / rcp / log as in dla_lane (1 ulp each on e in (0, 1] and 1 + e in (1, 2]: an absolute error of
        // 1e-7 on a pair's loss and sigmoid, far inside the 1e-5 parity bound; the lanes run ~16 of these in a row on one
        // wave per SIMD, where libm's expf / log1pf / divide cost 2 us of a 10 us rider)
        const float diff = oj - o;
        const float ex = fast_exp(-fabsf(diff));
        const float inv = fast_rcp(1.f + ex);
        const float sg = diff >= 0.f ? inv : ex * inv;   // d softplus(o_j - o_k) / d o_j
        if (WITH_PL) pl[j] += m1 * (fmaxf(diff, 0.f) + fast_log(1.f + ex));   // softplus(o_j - o_k)  (:151-155)
        acc += m2 / (tpj * tm_k) * (1.f - sg) - m1 / (tp_k * tmj) * sg;  // pair (j, k): sig(o_k - o_j) ; pair (k, j)
      }
    }
  }
  return acc;
}

__device__ __forceinline__ void pd_loss_small_body(const PdArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                   float* s_part, float* s_t) {
  const int B = a.B, L = a.L;
  float* s_tp = s_t;
  float* s_tm = s_t + 16;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = tid & 15, li = tid >> 4, gbase = lane & 48;
  const bool in_list = r < L;
  const float tp_k = in_list ? a.t_plus[r] : 1.f, tm_k = in_list ? a.t_minus[r] : 1.f;
  if (tid < 16) {
    s_tp[tid] = tp_k;
    s_tm[tid] = tm_k;
  }
  float pl[16];  // l_kj summed over this lane's lists, j = 0..15
#pragma unroll
  for (int j = 0; j < 16; ++j) pl[j] = 0.f;
  for (long long b0 = (long long)bid * 16; b0 < B; b0 += (long long)nb * 16) {
    const long long b = b0 + li;
    const bool live = in_list && b < B;
    const long long e = (long long)r * B + (b < B ? b : 0);
    const float o = live ? a.scores[e] : 0.f, c = live ? a.clicks[e] : 0.f;
    const float acc = pd_lane<true>(o, c, tp_k, tm_k, live, r, L, gbase, pl);
    if (live) a.d_scores[e] = acc;
  }
  // the four lists of the wave (same rank k: lanes k, k+16, k+32, k+48), then the waves
  float* mine = s_part + wid * kMaxStat;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float o1, o2;
    float v = xor16_partner(pl[j], &o1);
    v += o1;
    v = xor32_partner(v, &o2) + o2;
    if (lane < L && j < L) mine[lane * L + j] = v;
  }
  if (lane == 0) mine[L * L] = 0.f;
  BlockStats bs{s_part};
  bs.combine(s_acc, L * L + 1, 1 << 30);
  if (finish_stats(s_acc, L * L + 1, 1 << 30, a.ws, a.stats, bid, nb)) {  // block-uniform
    // loss = sum_ij l_ij / t+_i / t-_j  (:158-159), pre-update ratios: one term per thread, then a block sum
    __shared__ float red[4];
    float term[1] = {0.f};
    if (tid < L * L && tid / L != tid % L) term[0] = s_acc[tid] / s_tp[tid / L] / s_tm[tid % L];
    block_sum<1>(term, red);
    if (tid == 0) a.stats[L * L] = term[0];
  }
}


// Pairwise debiasing for MANY short lists (exactly LT <= 12 ranks): one lane per LIST, as dla_loss_lists_body.  Its LT
// scores and clicks sit in registers, every unordered pair is visited once (one exponential, one reciprocal and one
// logarithm serve both orders), the LT * LT pair-loss sums stay in registers over the grid-stride walk and meet once at
// the end, and 1 / (t+_k t-_j) comes from an LDS table that every lane reads at the same address.  The 16-lane body
// spends four cross-lane moves per pair and one group of 16 lanes per list; this one issues nothing but arithmetic.
// Same pair arithmetic as pd_lane (hardware exp / rcp / log), the per-rank sums in the same rank order; the division by
// t+ t- is a multiplication by the table's reciprocal, so results agree with pd_lane to rounding, not to the bit.
template <int LT>
__device__ __forceinline__ void pd_loss_lists_body(const PdArgs& a, unsigned int bid, unsigned int nb, float* s_acc,
                                                   float* s_part, float* s_t /* [32 + LT * LT] */) {
  const int B = a.B;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float* s_tp = s_t;
  float* s_tm = s_t + 16;
  float* s_w = s_t + 32;  // [k][j] = 1 / (t+_k * t-_j)
  if (tid < 16) {
    s_tp[tid] = tid < LT ? a.t_plus[tid] : 1.f;
    s_tm[tid] = tid < LT ? a.t_minus[tid] : 1.f;
  }
  __syncthreads();
  if (tid < LT * LT) s_w[tid] = 1.f / (s_tp[tid / LT] * s_tm[tid % LT]);
  __syncthreads();
  float pl[LT][LT];  // l_kj summed over this lane's lists
#pragma unroll
  for (int k = 0; k < LT; ++k)
#pragma unroll
    for (int j = 0; j < LT; ++j) pl[k][j] = 0.f;
  auto at = [](const float* row, unsigned int off) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + off); };
  auto at_w = [](float* row, unsigned int off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(row) + off); };
#pragma unroll 1
  for (unsigned int b = bid * kLossThreads + tid; b < (unsigned int)B; b += nb * kLossThreads) {
    const unsigned int off = b * 4u;
    float o[LT], c[LT], acc[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      o[l] = *at(a.scores + (long long)l * B, off);
      c[l] = *at(a.clicks + (long long)l * B, off);
      acc[l] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < LT; ++k)
#pragma unroll
      for (int j = k + 1; j < LT; ++j) {
        // At most one order of the pair counts: (k, j) when c_k > c_j, (j, k) when c_j > c_k, with weight
        // m = min(1, |c_k - c_j|) (:147-150).  With x = the counting order's score difference (loser - winner),
        // its loss is m softplus(x) (:151-155) and d loss / d score is -+ m sigmoid(x) / (t+_winner t-_loser): what the
        // winner's gradient loses the loser's gains (pd_lane's 1 - sigmoid(-x) is sigmoid(x)).
        const float d = c[k] - c[j];
        const float m = fminf(1.f, fabsf(d));   // 0 for a pair that does not count: its terms vanish below
        const bool kwins = d > 0.f;
        const float diff = o[j] - o[k];
        const float x = kwins ? diff : -diff;
        const float ex = fast_exp(-fabsf(diff));
        const float inv = fast_rcp(1.f + ex);
        // log(1 + e), 1 + e in (1, 2]: the bare hardware log2 (1 ulp), no denormal scaling, no correction terms
        const float lg = __builtin_amdgcn_logf(1.f + ex) * 0.6931471805599453f;
        const float sg = x >= 0.f ? inv : ex * inv;                 // sigmoid(x)
        const float val = m * (fmaxf(x, 0.f) + lg);
        // (both table entries are read, at compile-time addresses: a read at a selected address made the compiler wrap
        // the rest of the pair in a branch -- "skip when no lane's pair counts", which never happens with 64 lists in
        // a wave -- and 45 branches cut the loop body into blocks of one dependent chain each, 13 cycles per instruction)
        const float wkj = s_w[k * LT + j], wjk = s_w[j * LT + k];
        const float t = m * (kwins ? wkj : wjk) * sg;
        const float ts = kwins ? t : -t;  // what rank j's gradient gains and rank k's loses
        pl[k][j] += kwins ? val : 0.f;
        pl[j][k] += kwins ? 0.f : val;
        acc[k] -= ts;
        acc[j] += ts;
      }
#pragma unroll
    for (int l = 0; l < LT; ++l) *at_w(a.d_scores + (long long)l * B, off) = acc[l];
  }
  float* mine = s_part + wid * kMaxStat;
#pragma unroll
  for (int k = 0; k < LT; ++k)
#pragma unroll
    for (int j = 0; j < LT; ++j) {
      const float v = k == j ? 0.f : wave_sum(pl[k][j]);
      if (lane == 0) mine[k * LT + j] = v;
    }
  if (lane == 0) mine[LT * LT] = 0.f;
  BlockStats bs{s_part};
  bs.combine(s_acc, LT * LT + 1, 1 << 30);
  if (finish_stats(s_acc, LT * LT + 1, 1 << 30, a.ws, a.stats, bid, nb)) {  // block-uniform
    // loss = sum_ij l_ij / t+_i / t-_j  (:158-159), pre-update ratios: one term per thread, then a block sum
    __shared__ float red[4];
    float term[1] = {0.f};
    if (tid < LT * LT && tid / LT != tid % LT) term[0] = s_acc[tid] / s_tp[tid / LT] / s_tm[tid % LT];
    block_sum<1>(term, red);
    if (tid == 0) a.stats[LT * LT] = term[0];
  }
}

// A loss request parked by ultr_loss_{dla,ipw,rem,pd}_deferred until the next ultr_dnn_backward under the same request context
// carries it in its first launch.
enum { ULTR_RIDE_NONE = 0, ULTR_RIDE_DLA = 1, ULTR_RIDE_IPW = 2, ULTR_RIDE_REM = 3, ULTR_RIDE_PD = 4 };

// The EM-state updates of regression-EM and pairwise debiasing (a handful of floats) as device functions of ONE
// workgroup: launches of their own (rem_update_kernel, pd_update_kernel), or carried by the optimiser's update launch
// (ultr_*_update_*_deferred): they need only the all-reduced statistics, which exist before that launch.
constexpr int kEmMaxL = 16;
struct EmUpdate {
  int kind;            // 0 none, ULTR_RIDE_REM, ULTR_RIDE_PD
  float* a; float* b;  // REM: propensity, - ; PD: t_plus, t_minus
  int L; float* stats; float step, reg_p; float* old_out;
};
// p <- (1-s) p + s * stats[2+l]/stats[1]  (regression_EM.py:151-155)
__device__ __forceinline__ void rem_update_body(const EmUpdate& e) {
  const int l = threadIdx.x;
  if (l < e.L && e.old_out) e.old_out[l] = e.a[l];
  if (l < e.L) e.a[l] = (1.f - e.step) * e.a[l] + e.step * (e.stats[2 + l] / e.stats[1]);
}
// t+_i <- (1-s) t+_i + s (A_i/A_0)^(1/(p+1)), A_i = sum_j l_ij/t-_j ; t-_j likewise (pairwise_debias.py:162-171); also the
// loss from the (all-reduced) pair matrix into stats[L*L].  L <= kEmMaxL here (longer lists: pd_update_kernel).
__device__ __forceinline__ void pd_update_body(const EmUpdate& e) {
  __shared__ float A[kEmMaxL], Am[kEmMaxL], tp[kEmMaxL], tm[kEmMaxL];
  __shared__ float s_l[kEmMaxL * kEmMaxL];
  const int k = threadIdx.x, L = e.L;
  for (int i = k; i < L * L; i += blockDim.x) s_l[i] = e.stats[i];
  if (k < L) {
    tp[k] = e.a[k];
    tm[k] = e.b[k];
    if (e.old_out) {
      e.old_out[k] = tp[k];
      e.old_out[L + k] = tm[k];
    }
  }
  __syncthreads();
  float row = 0.f;
  if (k < L) {
    float a = 0.f, am = 0.f;
    for (int j = 0; j < L; ++j) {
      if (j == k) continue;
      a += s_l[k * L + j] / tm[j];
      am += s_l[j * L + k] / tp[j];
      row += s_l[k * L + j] / tp[k] / tm[j];
    }
    A[k] = a;
    Am[k] = am;
  }
  __syncthreads();
  if (k < L) {
    const float ex = 1.f / (e.reg_p + 1.f);
    e.a[k] = (1.f - e.step) * tp[k] + e.step * powf(A[k] / A[0], ex);
    e.b[k] = (1.f - e.step) * tm[k] + e.step * powf(Am[k] / Am[0], ex);
  }
  if (k < 64) {
    const float loss = wave_sum(row);
    if (k == 0) e.stats[L * L] = loss;
  }
}
struct PendingEm { bool on; EmUpdate e; };
PendingEm& pending_em();  // of the calling thread's bound request context (optim.hip)
#define g_pending_em (::ultr::pending_em())
int launch_em_now(const EmUpdate& e, hipStream_t st);  // loss.hip

struct PendingLoss {
  int kind;
  DlaArgs dla;
  IpwArgs ipw;
  RemArgs rem;
  PdArgs pd;
  unsigned nblocks;
  const float* d_scores() const {
    return kind == ULTR_RIDE_DLA ? dla.d_scores : kind == ULTR_RIDE_IPW ? ipw.d_scores : kind == ULTR_RIDE_REM ? rem.d_scores : pd.d_scores;
  }
  long long rows() const {
    return kind == ULTR_RIDE_DLA ? (long long)dla.B * dla.L : kind == ULTR_RIDE_IPW ? (long long)ipw.B * ipw.L
           : kind == ULTR_RIDE_REM ? (long long)rem.B * rem.L : (long long)pd.B * pd.L;
  }
};
PendingLoss& pending_loss();
#define g_pending_loss (::ultr::pending_loss())
int launch_loss_now(const PendingLoss& p, hipStream_t st);  // loss.hip

}  // namespace ultr
