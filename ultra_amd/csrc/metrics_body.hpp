// The metric kernel's short-list path (lists of at most 64 entries) and its final reduction as device functions,
// shared by metrics_kernel (metrics.hip) and by the backward pass, which can carry the in-step metrics as extra
// workgroups of one of its own launches (dnn.hip: the metrics need only the scores and the list weights, both of
// which exist before the backward pass starts).
// Reference: ultra/utils/metrics.py:215-323,445-481 ; metric_utils.py:49-86 ; base_algorithm.py:64-87.
#pragma once
#include "common.hpp"

namespace ultr {

constexpr int kMetThreads = 256;
constexpr int kMaxTopn = 8;
constexpr int kMaxOut = 3 * kMaxTopn * 2;
constexpr int kMetMaxBlocks = 1024;  // cap on the grid of the short-list path (a block then walks several list groups)
constexpr float kPaddingScore = -100000.f;  // BaseAlgorithm.PADDING_SCORE (base_algorithm.py:25)

struct MetricsArgs {
  const float* scores; const float* labels; const int* docids; int n_rows;
  const float* list_weights; const float* u_pred;
  int B, Lx, P, LPB, n_topn;
  int topn[kMaxTopn];
  int metric_mask; float max_label;
  float* out; float* ws;
  // tie-break uniforms drawn in the kernel when u_pred is NULL and tie_seed != 0 (sort_by_scores shuffles ties with
  // fresh uniforms on every call, metric_utils.py:76-81): Philox4x32-10, counter (tie_slot_base + list, 'TIE', rank / 4,
  // tie_step + *tie_ctr), key = tie_seed; the last workgroup of the launch bumps *tie_ctr, so the next call -- or the
  // next replay of a captured graph -- draws new ones
  unsigned long long tie_seed; unsigned int tie_step; unsigned int* tie_ctr; unsigned int tie_slot_base;
  // != 0 (the fused training step only): the workgroups leave their partial rows in ws and stop; the optimiser's update
  // launch, which comes next on the stream, adds them up (MetricsTail).  Takes the arrival counter, the wait for it and
  // the reload of the rows -- about 4 us of dependent memory round trips -- off the launch that carries this rider.
  int defer_reduce;
};

// The deferred last step of a metric evaluation: rows [nb][nout] -> out[nout] = column sums / B, in the order the last
// workgroup of metrics_finish uses (same reduce_rows walk, same block size), and the tie counter's bump.
struct MetricsTail {
  const float* rows; int nb, nout, B, weighted;
  float* out; unsigned int* tie_ctr;
};
__device__ __forceinline__ void metrics_tail_body(const MetricsTail& t, float* s_tmp, float* s_acc) {
  reduce_rows<false>(t.rows, t.nb, t.nout, 1 << 30, s_tmp, s_acc);
  const int tid = threadIdx.x;
  if (tid < t.nout && ((tid & 1) == 0 || t.weighted)) t.out[tid] = s_acc[tid] / (float)t.B;
  if (tid == 0 && t.tie_ctr) __hip_atomic_fetch_add(t.tie_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct PendingMetricsTail { bool on; MetricsTail t; };
PendingMetricsTail& pending_mtail();  // of the calling thread's bound request context (optim.hip)
#define g_pending_mtail (::ultr::pending_mtail())
int launch_metrics_tail_now(const MetricsTail& t, hipStream_t st);  // metrics.hip

__device__ __forceinline__ unsigned int tie_step_of(const MetricsArgs& a) {
  return a.tie_step + (a.tie_ctr ? __hip_atomic_load(a.tie_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
}
__device__ __forceinline__ float tie_uniform(const MetricsArgs& a, unsigned int step, long long b, int r) {
  const Philox4 rnd = philox4x32_10(a.tie_slot_base + (uint32_t)b, 0x544945u, (uint32_t)(r >> 2), step,
                                    (uint32_t)a.tie_seed, (uint32_t)(a.tie_seed >> 32));
  return u24(philox_word(rnd, r));
}

__device__ __forceinline__ bool goes_before(float sa, float ua, int ia, float sb, float ub, int ib, int Lx) {
  const bool fa = ia >= Lx, fb = ib >= Lx;  // fillers always sort last
  if (fa != fb) return fb;
  if (fa) return ia < ib;
  if (sa != sb) return sa > sb;
  if (ua != ub) return ua < ub;
  return ia < ib;
}

// The value held by lane ^ J inside a 16-lane row (J = 1, 2, 4, 8), at VALU speed: quad permutes for 1 and 2, a half-row
// mirror followed by a quad reversal for 4 (i -> 7 - i -> i ^ 4 inside each half row), a row rotation by 8 for 8.
template <int J>
__device__ __forceinline__ int xor_lane_i(int v) {
  if (J == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  if (J == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  if (J == 4) return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
  return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);               // row_ror:8
}
template <int J> __device__ __forceinline__ float xor_lane_f(float v) { return __int_as_float(xor_lane_i<J>(__float_as_int(v))); }

// Short lists: the workgroup `bid` of `nb` walks its groups of LPB lists and leaves one row of nout partial sums per
// wave in s_vals [4][nout].  PT = 16: the common case (lists of 9 to 16) with everything about the group width known
// at compile time -- the ten exchange steps of the sort are DPP moves instead of LDS-routed shuffles, the group sums
// carry no width dispatch; PT = 0: any power of two up to 64, decided at run time.  Same arithmetic in the same order.
template <int PT>
__device__ __forceinline__ void metrics_short_lists_t(const MetricsArgs& a, int bid, int nb, float* s_vals) {
  const float* __restrict__ scores = a.scores; const float* __restrict__ labels = a.labels;
  const int* __restrict__ docids = a.docids; const float* __restrict__ list_weights = a.list_weights;
  const float* __restrict__ u_pred = a.u_pred;
  const int n_rows = a.n_rows, B = a.B, Lx = a.Lx, P = PT ? PT : a.P, LPB = PT ? kMetThreads / PT : a.LPB, n_topn = a.n_topn,
            metric_mask = a.metric_mask;
  const float max_label = a.max_label;
  const int topn[kMaxTopn] = {a.topn[0], a.topn[1], a.topn[2], a.topn[3], a.topn[4], a.topn[5], a.topn[6], a.topn[7]};
  const int n_met = ((metric_mask & ULTR_METRIC_MRR) ? 1 : 0) + ((metric_mask & ULTR_METRIC_NDCG) ? 1 : 0) +
                    ((metric_mask & ULTR_METRIC_ERR) ? 1 : 0);
  const int nout = n_met * n_topn * 2;
  const int tid = threadIdx.x;
  float r_lab = 0.f, r_ilab = 0.f, r_w = 1.f;
  auto gmax = [&](float v) { return PT == 16 ? group_max<16>(v) : group_max_rt(v, P); };
  const bool draw_ties = !u_pred && a.tie_seed != 0ull;
  const unsigned int t_step = draw_ties ? tie_step_of(a) : 0u;
    // Short lists: a list is a group of P lanes inside one wave and never touches LDS.  The grid is capped
    // (kMetMaxBlocks): beyond that a block walks several groups of LPB lists and keeps its outputs in registers,
    // so the number of partial rows the last block has to add stays bounded.
    const int r = tid & (P - 1), li = tid / P;
    constexpr int kAccSlots = PT == 16 ? kMaxOut / 16 : kMaxOut / 2;  // outputs per lane: nout / P
    float vacc[kAccSlots];
#pragma unroll
    for (int j = 0; j < kAccSlots; ++j) vacc[j] = 0.f;
    for (long long b0 = (long long)bid * LPB; b0 < B; b0 += (long long)nb * LPB) {
      const long long b = b0 + li;
      // the bitonic network exchanges elements between lanes (lane ^ j); 10 exchange steps for P = 16, no barrier
      float s = -INFINITY, uu = 0.f, lb = 0.f;
      int id = Lx + r;  // filler
      // the list weight was written by the kernel right before this one: ask for it now, it is needed last
      if (b < B && list_weights) r_w = list_weights[b];
      if (b < B && r < Lx) {
        const long long g = (long long)r * B + b;
        s = scores[g];
        if (docids && docids[g] == n_rows) s = kPaddingScore;
        lb = labels ? labels[g] : 0.f;
        uu = u_pred ? u_pred[(long long)b * Lx + r] : draw_ties ? tie_uniform(a, t_step, b, r) : 0.f;
        id = r;
      }
      // _prepare_and_validate_params: invalid labels (< 0) -> label 0, prediction -> rowmin - 1e-6
      const bool real = id < Lx;
      const float any_invalid = gmax((real && lb < 0.f) ? 1.f : 0.f);
      if (any_invalid != 0.f) {
        const float mn = -gmax(real ? -s : -INFINITY);
        if (real && lb < 0.f) {
          lb = 0.f;
          s = mn - 1e-6f;
        }
      }
      const float lab_orig = lb;  // label of the document at ORIGINAL position r
      float il = real ? lb : -INFINITY;
      const bool want_ideal = (metric_mask & ULTR_METRIC_NDCG) != 0;
      // one compare-exchange step of the network: partner values in, this lane's element out
      auto exchange = [&](int k, int j, float ps, float pu, int pi, float pl) {
        const bool lower = (r & j) == 0, first_low = (r & k) == 0;
        const bool want_first = lower == first_low;  // this lane keeps the element that goes first
        const bool mine_first = goes_before(s, uu, id, ps, pu, pi, Lx);
        if (mine_first != want_first) {
          s = ps; uu = pu; id = pi;
        }
        if (want_ideal) {
          // descending labels; on equal values either copy is the same number
          const bool keep = want_first ? (il >= pl) : (il <= pl);
          if (!keep) il = pl;
        }
      };
      if (PT == 16) {
#define ULTR_BITONIC_STEP(K, J) exchange(K, J, xor_lane_f<J>(s), xor_lane_f<J>(uu), xor_lane_i<J>(id), want_ideal ? xor_lane_f<J>(il) : 0.f)
        ULTR_BITONIC_STEP(2, 1);
        ULTR_BITONIC_STEP(4, 2); ULTR_BITONIC_STEP(4, 1);
        ULTR_BITONIC_STEP(8, 4); ULTR_BITONIC_STEP(8, 2); ULTR_BITONIC_STEP(8, 1);
        ULTR_BITONIC_STEP(16, 8); ULTR_BITONIC_STEP(16, 4); ULTR_BITONIC_STEP(16, 2); ULTR_BITONIC_STEP(16, 1);
#undef ULTR_BITONIC_STEP
      } else {
        for (int k = 2; k <= P; k <<= 1)
          for (int j = k >> 1; j > 0; j >>= 1)
            exchange(k, j, __shfl_xor(s, j, 64), __shfl_xor(uu, j, 64), __shfl_xor(id, j, 64),
                     want_ideal ? __shfl_xor(il, j, 64) : 0.f);
      }
      // label of the document now ranked r: fetch it from the lane of its original position
      const int src_lane = (tid & 63) - r + min(id, P - 1);
      r_lab = __shfl(lab_orig, src_lane, 64);
      if (id >= Lx) r_lab = 0.f;
      r_ilab = il;
      // short lists: one lane per (list, rank); a list is a P-lane group inside one wave, so sums, maxima and
      // the ERR prefix product are shuffles within the group -- no serial walk, every lane busy
      const bool live = r < Lx;
      const float lbr = live ? r_lab : 0.f;
      const float ilr = live ? r_ilab : 0.f;
      const float gain = live ? exp2f(lbr) - 1.f : 0.f, igain = live ? exp2f(ilr) - 1.f : 0.f;
      const float disc = logf((float)r + 2.f);
      const float gd = gain / disc, igd = igain / disc;
      auto gsum = [&](float v) { return PT == 16 ? group_sum<16>(v) : group_sum_rt(v, P); };
      const float rel_sum = gsum(gain);
      float mrr = (live && lbr >= 1.f) ? 1.f / (float)(r + 1) : 0.f;
      mrr = gmax(mrr);
      float e_term = 0.f;
      if (metric_mask & ULTR_METRIC_ERR) {
        const float rel = gain / powf(2.f, max_label);  // metrics.py:312-313
        float cum = live ? 1.f - rel : 1.f;             // inclusive prefix product within the group
        for (int o = 1; o < P; o <<= 1) {
          const float up = __shfl_up(cum, o, 64);
          if (r >= o) cum *= up;
        }
        e_term = live ? rel * (cum / (1.f - rel)) / (float)(r + 1) : 0.f;  // cumprod/(1-rel) (:314)
      }
      float dcg[kMaxTopn], idcg[kMaxTopn], err[kMaxTopn];
  #pragma unroll
      for (int q = 0; q < kMaxTopn; ++q) {
        dcg[q] = idcg[q] = err[q] = 0.f;
        if (q < n_topn) {
          const bool in = r < topn[q];
          if (metric_mask & ULTR_METRIC_NDCG) {
            dcg[q] = gsum(in ? gd : 0.f);
            idcg[q] = gsum(in ? igd : 0.f);
          }
          if (metric_mask & ULTR_METRIC_ERR) err[q] = gsum(in ? e_term : 0.f);
        }
      }
      // The list's sums are uniform over its P lanes, so lane r forms output o = r (r + P, ...) of the list instead of
      // lane 0 walking all outputs; the lists of a wave are then added lane-wise (same r), the waves meet in LDS.
      const float w = r_w;
      const float plw_plain = rel_sum != 0.f ? 1.f : 0.f, plw_w = rel_sum != 0.f ? w : 0.f;
      const int mi_ndcg = (metric_mask & ULTR_METRIC_MRR) ? 1 : 0;
      const int mi_err = mi_ndcg + ((metric_mask & ULTR_METRIC_NDCG) ? 1 : 0);
      for (int o = r; o < nout; o += P) {
        const int weighted = o & 1, q = (o >> 1) % n_topn, mi = (o >> 1) / n_topn;
        float d = 0.f, idn = 0.f, er = 0.f;
  #pragma unroll
        for (int qq = 0; qq < kMaxTopn; ++qq)
          if (qq == q) {
            d = dcg[qq];
            idn = idcg[qq];
            er = err[qq];
          }
        float val;
        if ((metric_mask & ULTR_METRIC_MRR) && mi == 0) {
          val = weighted ? mrr * w : mrr;
        } else if ((metric_mask & ULTR_METRIC_NDCG) && mi == mi_ndcg) {
          const float nd = idn != 0.f ? d / idn : 0.f;
          const float ndw = (w * idn) != 0.f ? (w * d) / (w * idn) : 0.f;
          val = weighted ? ndw * plw_w : nd * plw_plain;
        } else {
          (void)mi_err;
          val = weighted ? er * w : er;
        }
        if (b >= B) val = 0.f;
        const int slot = (o - r) / P;
#pragma unroll
        for (int j = 0; j < kAccSlots; ++j)
          if (j == slot) vacc[j] += val;
      }
    }
    float* s_w = s_vals;  // [4 waves][nout]
    for (int o = r; o < nout; o += P) {
      float val = 0.f;
      const int slot = (o - r) / P;
#pragma unroll
      for (int j = 0; j < kAccSlots; ++j)
        if (j == slot) val = vacc[j];
      if (PT == 16) {  // the other lists of this wave
        float o1, o2;
        val = xor16_partner(val, &o1);
        val += o1;
        val = xor32_partner(val, &o2) + o2;
      } else {
        for (int off = P; off < 64; off <<= 1) val += __shfl_xor(val, off, 64);
      }
      if ((tid & 63) < P) s_w[(tid >> 6) * nout + o] = val;
    }
}

__device__ __forceinline__ void metrics_short_lists(const MetricsArgs& a, int bid, int nb, float* s_vals) {
  if (a.P == 16) metrics_short_lists_t<16>(a, bid, nb, s_vals);  // uniform over the launch
  else metrics_short_lists_t<0>(a, bid, nb, s_vals);
}

// Block partial sums in fixed order; the last workgroup to arrive adds the rows and divides by B (reduce_mean over lists).
// nrow = rows of s_vals to add (short lists: one per wave, long lists: one per list).
__device__ __forceinline__ void metrics_finish(const MetricsArgs& a, int bid, int nb, const float* s_vals, int nrow) {
  const float* list_weights = a.list_weights;
  float* out = a.out; float* ws = a.ws;
  const int B = a.B, tid = threadIdx.x;
  const int n_met = ((a.metric_mask & ULTR_METRIC_MRR) ? 1 : 0) + ((a.metric_mask & ULTR_METRIC_NDCG) ? 1 : 0) +
                    ((a.metric_mask & ULTR_METRIC_ERR) ? 1 : 0);
  const int nout = n_met * a.n_topn * 2;
  // block partial sums in fixed order, then the last block divides by B (reduce_mean over lists)
  __shared__ float s_acc[kMaxOut];
  if (tid < nout) {
    float s = 0.f;
    for (int l = 0; l < nrow; ++l) s += s_vals[l * nout + tid];
    s_acc[tid] = s;
  }
  float* rows = a.defer_reduce ? ws + 8 : arrive_rows(ws, (unsigned int)nb);  // (deferred: no arrival, MetricsTail reads ws + 8)
  __syncthreads();
  if (a.defer_reduce) {  // the next launch on the stream adds the rows
    if (tid < nout) rows[(long long)bid * nout + tid] = s_acc[tid];
    return;
  }
  // hand-off without cache maintenance: agent-scope stores and loads around an atomic arrival counter (loss_body.hpp)
  if (tid < nout) st_agent(rows + (long long)bid * nout + tid, s_acc[tid]);
  if (last_block_arrive_ws(ws, (unsigned int)bid, (unsigned int)nb)) {
    __shared__ float s_tmp[kMetThreads];
    reduce_rows<true>(rows, (int)nb, nout, 1 << 30, s_tmp, s_acc);
    if (tid < nout && ((tid & 1) == 0 || list_weights)) out[tid] = s_acc[tid] / (float)B;
    // every workgroup has arrived, so every workgroup has read the tie counter: the next call draws fresh uniforms
    if (tid == 0 && a.tie_ctr && !a.u_pred && a.tie_seed != 0ull)
      __hip_atomic_fetch_add(a.tie_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// A metric request parked by ultr_metrics_deferred until the next ultr_dnn_backward under the same request context carries it.
struct PendingMetrics {
  bool on;
  MetricsArgs a;
  unsigned nblocks;
};
PendingMetrics& pending_metrics();
#define g_pending_metrics (::ultr::pending_metrics())
int launch_metrics_now(const MetricsArgs& a, unsigned nblocks, hipStream_t st);  // metrics.hip

}  // namespace ultr
