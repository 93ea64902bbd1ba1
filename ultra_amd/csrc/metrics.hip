// Ranking metrics with per-list bitonic sorts in LDS.
// Reference: ultra/utils/metrics.py:215-251 (_prepare_and_validate_params), :254-284 (MRR),
//   :287-323 (ERR), :445-481 (NDCG), :192-212 (DCG), :158-189 (safe_div, per-list weights);
//   ultra/utils/metric_utils.py:49-86 (sort_by_scores); padding mask base_algorithm.py:64-87.
//
// sort_by_scores composes stable-argsort(uniforms) with top_k (equal values -> lower index
// first), which is exactly a sort by the total order (score desc, uniform asc, index asc); a
// bitonic network under that comparator therefore yields the reference permutation bit for bit.
#include "metrics_body.hpp"
#include <mutex>

namespace ultr {

struct SortTile {
  float* key;  // [E] prediction (after masking)
  float* u;    // [E] tie-break uniform
  int* idx;    // [E] original position; >= Lx marks a filler element
  float* lab;  // [E] label in ORIGINAL order
  float* ilab; // [E] labels, sorted descending (ideal ranking)
};


// Bitonic network over lists of P (power of two) elements laid side by side in LDS; E = total
// elements of the block.  Every thread walks its elements e = tid, tid+256, ...
__device__ __forceinline__ void bitonic_sort_tile(const SortTile& t, int P, int E, int Lx, bool also_ideal) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int li = e & (P - 1);
        const int pj = li ^ j;
        if (pj > li) {
          const int a = e, b = e - li + pj;
          const bool first_low = (li & k) == 0;
          const float sa = t.key[a], sb = t.key[b], ua = t.u[a], ub = t.u[b];
          const int ia = t.idx[a], ib = t.idx[b];
          const bool a_first = goes_before(sa, ua, ia, sb, ub, ib, Lx);
          if (a_first != first_low) {
            t.key[a] = sb; t.key[b] = sa;
            t.u[a] = ub; t.u[b] = ua;
            t.idx[a] = ib; t.idx[b] = ia;
          }
          if (also_ideal) {
            const float la = t.ilab[a], lb = t.ilab[b];
            if ((la >= lb) != first_low) {
              t.ilab[a] = lb; t.ilab[b] = la;
            }
          }
        }
      }
      __syncthreads();
    }
  }
}

// Load a block's lists into LDS.  Element e -> list (blockIdx*LPB + e/P), position e%P.
__device__ __forceinline__ void load_tile(const SortTile& t, const float* __restrict__ scores,
                                          const float* __restrict__ labels, const int* __restrict__ docids,
                                          int n_rows, const float* __restrict__ u_pred, int B, int Lx, int P, int E,
                                          int LPB, const MetricsArgs* ties = nullptr) {
  const bool draw_ties = ties && !u_pred && ties->tie_seed != 0ull;
  const unsigned int t_step = draw_ties ? tie_step_of(*ties) : 0u;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const int li = e & (P - 1);
    const int b = blockIdx.x * LPB + e / P;
    float s = -INFINITY, uu = 0.f, lb = 0.f;
    int id = Lx + li;  // filler
    if (b < B && li < Lx) {
      const long long g = (long long)li * B + b;
      s = scores[g];
      if (docids && docids[g] == n_rows) s = kPaddingScore;
      lb = labels ? labels[g] : 0.f;
      uu = u_pred ? u_pred[(long long)b * Lx + li] : draw_ties ? tie_uniform(*ties, t_step, b, li) : 0.f;
      id = li;
    }
    t.key[e] = s;
    t.u[e] = uu;
    t.idx[e] = id;
    t.lab[e] = lb;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kMetThreads) metrics_kernel(MetricsArgs a) {
  extern __shared__ float smem[];
  const float* __restrict__ scores = a.scores; const float* __restrict__ labels = a.labels;
  const int* __restrict__ docids = a.docids; const float* __restrict__ list_weights = a.list_weights;
  const float* __restrict__ u_pred = a.u_pred;
  const int n_rows = a.n_rows, B = a.B, Lx = a.Lx, P = a.P, LPB = a.LPB, n_topn = a.n_topn, metric_mask = a.metric_mask;
  const float max_label = a.max_label;
  const int E = P * LPB;
  SortTile t;
  t.key = smem;
  t.u = smem + E;
  t.idx = reinterpret_cast<int*>(smem + 2 * E);
  t.lab = smem + 3 * E;
  t.ilab = smem + 4 * E;
  float* s_vals = smem + 5 * E;  // [LPB][nout]
  const int topn[kMaxTopn] = {a.topn[0], a.topn[1], a.topn[2], a.topn[3], a.topn[4], a.topn[5], a.topn[6], a.topn[7]};
  const int n_met = ((metric_mask & ULTR_METRIC_MRR) ? 1 : 0) + ((metric_mask & ULTR_METRIC_NDCG) ? 1 : 0) +
                    ((metric_mask & ULTR_METRIC_ERR) ? 1 : 0);
  const int nout = n_met * n_topn * 2;
  const int tid = threadIdx.x;
  if (P <= 64) {
    metrics_short_lists(a, blockIdx.x, gridDim.x, s_vals);
  } else {
  load_tile(t, scores, labels, docids, n_rows, u_pred, B, Lx, P, E, LPB, &a);

  // _prepare_and_validate_params: invalid labels (< 0) -> label 0, prediction -> rowmin - 1e-6
  if (tid < LPB) {
    const int base = tid * P;
    bool any_invalid = false;
    float mn = INFINITY;
    for (int i = 0; i < Lx; ++i) {
      any_invalid = any_invalid || t.lab[base + i] < 0.f;
      mn = fminf(mn, t.key[base + i]);
    }
    if (any_invalid) {
      for (int i = 0; i < Lx; ++i)
        if (t.lab[base + i] < 0.f) {
          t.lab[base + i] = 0.f;
          t.key[base + i] = mn - 1e-6f;
        }
    }
  }
  __syncthreads();
  for (int e = tid; e < E; e += blockDim.x) t.ilab[e] = ((e & (P - 1)) < Lx) ? t.lab[e] : -INFINITY;
  __syncthreads();

  bitonic_sort_tile(t, P, E, Lx, (metric_mask & ULTR_METRIC_NDCG) != 0);
  // long lists (evaluation over max_candidate_num): one thread per list walks the sorted list
  if (tid < LPB) {
    const int base = tid * P;
    const int b = blockIdx.x * LPB + tid;
    float* vals = s_vals + tid * nout;
    for (int o = 0; o < nout; ++o) vals[o] = 0.f;
    if (b < B) {
      const float w = list_weights ? list_weights[b] : 1.f;
      float mrr = 0.f, rel_sum = 0.f, cum = 1.f;
      float dcg[kMaxTopn], idcg[kMaxTopn], err[kMaxTopn];
#pragma unroll
      for (int q = 0; q < kMaxTopn; ++q) dcg[q] = idcg[q] = err[q] = 0.f;
      const float inv_pow_max = 1.f / powf(2.f, max_label);
      for (int r = 0; r < Lx; ++r) {
        const float lb = t.lab[base + t.idx[base + r]];
        const float il = t.ilab[base + r];
        const float gain = exp2f(lb) - 1.f, igain = exp2f(il) - 1.f;
        const float disc = logf((float)r + 2.f);  // log(1 + position), position = r+1
        rel_sum += gain;
        if (mrr == 0.f && lb >= 1.f) mrr = 1.f / (float)(r + 1);
        const float rel = gain * inv_pow_max;  // metrics.py:312-313
        cum *= (1.f - rel);
        const float non_rel = cum / (1.f - rel);  // cumprod/(1-rel) (:314)
        const float e_term = rel * non_rel / (float)(r + 1);
#pragma unroll
        for (int q = 0; q < kMaxTopn; ++q) {
          if (q < n_topn && r < topn[q]) {
            dcg[q] += gain / disc;
            idcg[q] += igain / disc;
            err[q] += e_term;
          }
        }
      }
      const float plw_plain = rel_sum != 0.f ? 1.f : 0.f;  // _per_example_weights_to_per_list_weights
      const float plw_w = rel_sum != 0.f ? w : 0.f;
      int mi = 0;
      if (metric_mask & ULTR_METRIC_MRR) {
#pragma unroll
        for (int q = 0; q < kMaxTopn; ++q)
          if (q < n_topn) {
            vals[(mi * n_topn + q) * 2 + 0] = mrr;
            vals[(mi * n_topn + q) * 2 + 1] = mrr * w;
          }
        ++mi;
      }
      if (metric_mask & ULTR_METRIC_NDCG) {
#pragma unroll
        for (int q = 0; q < kMaxTopn; ++q)
          if (q < n_topn) {
            const float nd = idcg[q] != 0.f ? dcg[q] / idcg[q] : 0.f;
            const float ndw = (w * idcg[q]) != 0.f ? (w * dcg[q]) / (w * idcg[q]) : 0.f;
            vals[(mi * n_topn + q) * 2 + 0] = nd * plw_plain;
            vals[(mi * n_topn + q) * 2 + 1] = ndw * plw_w;
          }
        ++mi;
      }
      if (metric_mask & ULTR_METRIC_ERR) {
#pragma unroll
        for (int q = 0; q < kMaxTopn; ++q)
          if (q < n_topn) {
            vals[(mi * n_topn + q) * 2 + 0] = err[q];
            vals[(mi * n_topn + q) * 2 + 1] = err[q] * w;
          }
        ++mi;
      }
    }
  }
  }
  __syncthreads();
  metrics_finish(a, blockIdx.x, gridDim.x, s_vals, P <= 64 ? kMetThreads / 64 : LPB);
}

// Large batches of short lists (Lx <= 16, B >= kLanesMinB): ONE LANE PER LIST.  Consecutive lanes read consecutive
// lists of the position-major arrays (fully coalesced, where the group-per-list mapping above touches 16 bytes of each
// 128-byte line per load), a list's scores / labels / uniforms live in registers, and the positions come from
// rank-by-counting over the list's pairs -- no lane ever talks to another before the final sums.
constexpr int kLanesMinB = 32768, kLanesMaxBlocks = 1024;  // (four waves per SIMD; every workgroup arrives at ONE counter: ~16 ns each)
// HAS_U: tie-break uniforms exist (supplied or drawn); without them ties fall to the lower index and no uniform is held.
template <int LM, bool WANT_ERR, bool HAS_U>
__global__ void __launch_bounds__(kMetThreads) metrics_lanes_kernel(MetricsArgs a) {
  __shared__ float s_vals[4 * kMaxOut];
  // 1 / log(rank + 2), the discount, and 1 / (rank + 1): two tables instead of 2 Lx logarithms and 3 Lx divisions a list
  // (the discount becomes a multiplication by the correctly rounded reciprocal: 1 ulp from the quotient)
  __shared__ float s_rlog[LM], s_rinv[LM];
  if (threadIdx.x < LM) {
    s_rlog[threadIdx.x] = 1.f / logf((float)threadIdx.x + 2.f);
    s_rinv[threadIdx.x] = 1.f / (float)(threadIdx.x + 1);
  }
  const float* __restrict__ scores = a.scores; const float* __restrict__ labels = a.labels;
  const int* __restrict__ docids = a.docids; const float* __restrict__ list_weights = a.list_weights;
  const float* __restrict__ u_pred = a.u_pred;
  const int n_rows = a.n_rows, B = a.B, n_topn = a.n_topn, metric_mask = a.metric_mask;
  const int Lx = LM == 10 ? 10 : a.Lx;  // (LM == 10 is launched for lists of exactly 10: no per-rank guards survive)
  const int topn[kMaxTopn] = {a.topn[0], a.topn[1], a.topn[2], a.topn[3], a.topn[4], a.topn[5], a.topn[6], a.topn[7]};
  const bool want_mrr = (metric_mask & ULTR_METRIC_MRR) != 0, want_ndcg = (metric_mask & ULTR_METRIC_NDCG) != 0;
  constexpr bool want_err = WANT_ERR;
  const int n_met = (want_mrr ? 1 : 0) + (want_ndcg ? 1 : 0) + (want_err ? 1 : 0);
  const int nout = n_met * n_topn * 2;
  const float inv_pow = 1.f / powf(2.f, a.max_label);
  const int tid = threadIdx.x;
  const bool draw_ties = !u_pred && a.tie_seed != 0ull;
  const unsigned int t_step = draw_ties ? tie_step_of(a) : 0u;
  // Per-lane sums over the lane's lists, [metric][topn][plain, weighted] at compile-time positions (only the selected
  // metrics' first n_topn entries are ever touched); the lanes meet once, after the walk.
  float acc_m[kMaxTopn][2], acc_n[kMaxTopn][2], acc_e[kMaxTopn][2];
#pragma unroll
  for (int q = 0; q < kMaxTopn; ++q) acc_m[q][0] = acc_m[q][1] = acc_n[q][0] = acc_n[q][1] = acc_e[q][0] = acc_e[q][1] = 0.f;
  __syncthreads();
  for (long long b0 = (long long)blockIdx.x * kMetThreads; b0 < B; b0 += (long long)gridDim.x * kMetThreads) {
    const bool on = b0 + tid < B;
    const long long b = on ? b0 + tid : B - 1;  // idle lanes redo the last list and contribute nothing
    float s[LM], lab[LM], u[LM];
    // every load of the list in one batch (a per-rank "is there a doc id array" test between them made each rank's
    // loads a basic block of its own, one memory round trip after the other)
#pragma unroll
    for (int l = 0; l < LM; ++l) {
      s[l] = -INFINITY; lab[l] = 0.f; u[l] = 0.f;
      if (l < Lx) {
        const long long g = (long long)l * B + b;
        s[l] = scores[g];
        lab[l] = labels[g];
      }
    }
    const float w = list_weights ? list_weights[b] : 1.f;
    if (docids) {
      int di[LM];
#pragma unroll
      for (int l = 0; l < LM; ++l) di[l] = l < Lx ? docids[(long long)l * B + b] : 0;
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < Lx) s[l] = di[l] == n_rows ? kPaddingScore : s[l];
    }
    if (HAS_U) {
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < Lx) u[l] = u_pred ? u_pred[b * Lx + l] : tie_uniform(a, t_step, b, l);
    }
    // _prepare_and_validate_params: invalid labels (< 0) -> label 0, prediction -> rowmin - 1e-6
    bool any_invalid = false;
    float mn = INFINITY;
#pragma unroll
    for (int l = 0; l < LM; ++l)
      if (l < Lx) {
        any_invalid = any_invalid || lab[l] < 0.f;
        mn = fminf(mn, s[l]);
      }
    if (any_invalid) {
#pragma unroll
      for (int l = 0; l < LM; ++l)
        if (l < Lx && lab[l] < 0.f) {
          lab[l] = 0.f;
          s[l] = mn - 1e-6f;
        }
    }
    // Positions by counting over the pairs j < i: j goes first iff it wins (score desc, uniform asc, index asc).  Both
    // counters of a pair step on the SAME mask (one add-with-carry each): before[i] counts the j < i that go before i,
    // after[j] the i > j that go after j, and rank = before + (the ranks above) - after.
    int before[LM], after[LM], ibefore[LM], iafter[LM];
    float above[LM], omr[LM], gain[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l) {
      before[l] = after[l] = ibefore[l] = iafter[l] = 0;
      above[l] = 1.f;
      gain[l] = __builtin_amdgcn_exp2f(lab[l]) - 1.f;  // (labels are small and non-negative: the bare v_exp_f32)
      omr[l] = 1.f - gain[l] * inv_pow;
    }
#pragma unroll
    for (int i = 1; i < LM; ++i) {
      if (i < Lx) {
#pragma unroll
        for (int j = 0; j < i; ++j) {
          const bool jb = HAS_U ? (s[j] > s[i] || (s[j] == s[i] && u[j] <= u[i])) : s[j] >= s[i];
          before[i] += jb ? 1 : 0;
          after[j] += jb ? 1 : 0;
          const bool jl = lab[j] >= lab[i];  // (the ideal order too, wanted or not: a uniform test per pair costs more)
          ibefore[i] += jl ? 1 : 0;
          iafter[j] += jl ? 1 : 0;
          if (want_err) {
            above[i] *= jb ? omr[j] : 1.f;
            above[j] *= jb ? 1.f : omr[i];
          }
        }
      }
      // one row of pairs at a time: left alone the scheduler hoists every comparison of the list to the front, and the
      // ~100 lane masks they produce spill from the scalar registers (measured: 2500 readlane / writelane per list walk)
      __builtin_amdgcn_sched_barrier(0);
    }
    float mrr = 0.f, rel_sum = 0.f;
    int rank[LM], irank[LM];
    float gd[LM], igd[LM], e_term[LM];
#pragma unroll
    for (int l = 0; l < LM; ++l) {
      rank[l] = irank[l] = LM;  // beyond every cutoff
      gd[l] = igd[l] = e_term[l] = 0.f;
      if (l < Lx) {
        rank[l] = before[l] + (Lx - 1 - l) - after[l];
        irank[l] = ibefore[l] + (Lx - 1 - l) - iafter[l];
        rel_sum += gain[l];
        // (table reads at unconditional places: a read under a lane condition becomes a branch with its own wait)
        const float rinv = s_rinv[rank[l]];
        mrr = fmaxf(mrr, lab[l] >= 1.f ? rinv : 0.f);
        gd[l] = gain[l] * s_rlog[rank[l]];
        igd[l] = gain[l] * s_rlog[irank[l]];
        // cumprod(1 - rel) / (1 - rel) as the reference writes it (metrics.py:312-314)
        if (want_err) e_term[l] = (1.f - omr[l]) * ((above[l] * omr[l]) / omr[l]) * rinv;
      }
    }
    const float plw_plain = rel_sum != 0.f ? 1.f : 0.f, plw_w = rel_sum != 0.f ? w : 0.f;
    const float keep = on ? 1.f : 0.f;
#pragma unroll
    for (int q = 0; q < kMaxTopn; ++q)
      if (q < n_topn) {  // uniform
        if (want_mrr) {
          acc_m[q][0] += keep * mrr;
          acc_m[q][1] += keep * (mrr * w);
        }
        if (want_ndcg) {
          float d = 0.f, idn = 0.f;
#pragma unroll
          for (int l = 0; l < LM; ++l) {
            d += rank[l] < topn[q] ? gd[l] : 0.f;
            idn += irank[l] < topn[q] ? igd[l] : 0.f;
          }
          const float nd = idn != 0.f ? __fdividef(d, idn) : 0.f;
          const float ndw = (w * idn) != 0.f ? __fdividef(w * d, w * idn) : 0.f;
          acc_n[q][0] += keep * (nd * plw_plain);
          acc_n[q][1] += keep * (ndw * plw_w);
        }
        if (want_err) {
          float e = 0.f;
#pragma unroll
          for (int l = 0; l < LM; ++l) e += rank[l] < topn[q] ? e_term[l] : 0.f;
          acc_e[q][0] += keep * e;
          acc_e[q][1] += keep * (e * w);
        }
      }
  }
  // outputs in the order [metric][topn][plain, weighted]: the slot index is a running (uniform) count
  float* my_row = s_vals + (tid >> 6) * nout;
  int o = 0;
  auto emit = [&](float plain, float weighted) {
    const float v0 = wave_sum(plain), v1 = wave_sum(weighted);
    if ((tid & 63) == 0) {
      my_row[o] = v0;
      my_row[o + 1] = v1;
    }
    o += 2;
  };
  if (want_mrr) {
#pragma unroll
    for (int q = 0; q < kMaxTopn; ++q)
      if (q < n_topn) emit(acc_m[q][0], acc_m[q][1]);
  }
  if (want_ndcg) {
#pragma unroll
    for (int q = 0; q < kMaxTopn; ++q)
      if (q < n_topn) emit(acc_n[q][0], acc_n[q][1]);
  }
  if (want_err) {
#pragma unroll
    for (int q = 0; q < kMaxTopn; ++q)
      if (q < n_topn) emit(acc_e[q][0], acc_e[q][1]);
  }
  __syncthreads();
  metrics_finish(a, blockIdx.x, gridDim.x, s_vals, kMetThreads / 64);
}

__global__ void __launch_bounds__(kMetThreads) sort_perm_kernel(const float* __restrict__ scores,
                                                                const float* __restrict__ u, int B, int Lx, int P,
                                                                int LPB, int topn, int* __restrict__ perm) {
  extern __shared__ float smem[];
  const int E = P * LPB;
  SortTile t;
  t.key = smem;
  t.u = smem + E;
  t.idx = reinterpret_cast<int*>(smem + 2 * E);
  t.lab = smem + 3 * E;
  t.ilab = smem + 4 * E;
  load_tile(t, scores, nullptr, nullptr, 0, u, B, Lx, P, E, LPB);
  bitonic_sort_tile(t, P, E, Lx, false);
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const int li = e & (P - 1);
    const int b = blockIdx.x * LPB + e / P;
    if (b < B && li < topn) perm[(long long)b * topn + li] = t.idx[e];
  }
}

static void sort_geometry(int Lx, int* P, int* LPB) {
  int p = 2;
  while (p < Lx) p <<= 1;
  *P = p;
  *LPB = p >= kMetThreads ? 1 : kMetThreads / p;
}

}  // namespace ultr

using namespace ultr;

extern "C" long long ultr_metrics_workspace_floats(int B, int n_topn) {
  // worst case: one list per block
  return 8 + (long long)(B > 0 ? B : 1) * 3 * (n_topn > 0 ? n_topn : 1) * 2 + kArriveRowsFar;  // (+ two-level arrival, common.hpp)
}

namespace ultr {

__global__ void __launch_bounds__(kMetThreads) metrics_tail_kernel(MetricsTail t) {
  __shared__ float s_tmp[kMetThreads];
  __shared__ float s_acc[kMaxOut];
  metrics_tail_body(t, s_tmp, s_acc);
}
int launch_metrics_tail_now(const MetricsTail& t, hipStream_t st) {
  {
    ultr::ProfScope prof_("metrics_tail_kernel", st);
    hipLaunchKernelGGL(metrics_tail_kernel, dim3(1), dim3(kMetThreads), 0, st, t);
  }
  return check_launch("metrics_tail_kernel");
}

static size_t metrics_smem(const MetricsArgs& a) {
  const int n_met = ((a.metric_mask & 1) ? 1 : 0) + ((a.metric_mask & 2) ? 1 : 0) + ((a.metric_mask & 4) ? 1 : 0);
  return sizeof(float) * ((size_t)5 * a.P * a.LPB + (size_t)a.LPB * n_met * a.n_topn * 2);
}

int launch_metrics_now(const MetricsArgs& a, unsigned nblocks, hipStream_t st) {
  static PerDeviceOnce attr_set;
  if (const int rc_ = raise_lds_limit(attr_set, reinterpret_cast<const void*>(metrics_kernel), 96 * 1024, "ultr_metrics")) return rc_;
  if (a.P <= 16 && a.B >= kLanesMinB) {  // one lane per list (nblocks was sized for it by prepare_metrics)
    ultr::ProfScope prof_("metrics_lanes_kernel", st);
    const bool err = (a.metric_mask & ULTR_METRIC_ERR) != 0, has_u = a.u_pred || a.tie_seed != 0ull;
#define ULTR_LANES(LM, E, U) hipLaunchKernelGGL((metrics_lanes_kernel<LM, E, U>), dim3(nblocks), dim3(kMetThreads), 0, st, a)
    if (a.Lx == 10) {  // the reference's rank_list_size: compiled for the exact length, no per-rank guards
      if (err) { if (has_u) ULTR_LANES(10, true, true); else ULTR_LANES(10, true, false); }
      else { if (has_u) ULTR_LANES(10, false, true); else ULTR_LANES(10, false, false); }
    } else if (a.Lx <= 8) {
      if (err) { if (has_u) ULTR_LANES(8, true, true); else ULTR_LANES(8, true, false); }
      else { if (has_u) ULTR_LANES(8, false, true); else ULTR_LANES(8, false, false); }
    } else {
      if (err) { if (has_u) ULTR_LANES(16, true, true); else ULTR_LANES(16, true, false); }
      else { if (has_u) ULTR_LANES(16, false, true); else ULTR_LANES(16, false, false); }
    }
#undef ULTR_LANES
    return check_launch("metrics_lanes_kernel");
  }
  {
    ultr::ProfScope prof_("metrics_kernel", st);
    hipLaunchKernelGGL(metrics_kernel, dim3(nblocks), dim3(kMetThreads), metrics_smem(a), st, a);
  }
  return check_launch("metrics_kernel");
}

// argument checks + launch geometry shared by ultr_metrics and ultr_metrics_deferred
struct TieSpec { unsigned long long seed; unsigned int step; unsigned int* ctr; unsigned int slot_base; };
static int prepare_metrics(const char* who, const float* scores, const float* labels, const int* docids, int n_rows,
                           const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h, int n_topn,
                           int metric_mask, float max_label, float* out, float* ws, MetricsArgs* a, unsigned* nblocks,
                           TieSpec ties = TieSpec{0ull, 0u, nullptr, 0u}) {
  ULTR_REQUIRE(scores && labels && out && ws && topn_h, ULTR_EINVAL, "%s: null pointer", who);
  ULTR_REQUIRE(B > 0 && Lx > 0, ULTR_EINVAL, "%s: bad sizes B=%d Lx=%d", who, B, Lx);
  ULTR_REQUIRE(Lx <= ULTR_MAX_SORT, ULTR_ELIMIT, "%s: list length %d > %d", who, Lx, ULTR_MAX_SORT);
  ULTR_REQUIRE(n_topn >= 1 && n_topn <= kMaxTopn, ULTR_ELIMIT, "%s: n_topn=%d not in [1,%d]", who, n_topn, kMaxTopn);
  ULTR_REQUIRE((metric_mask & ~7) == 0 && metric_mask != 0, ULTR_EINVAL, "%s: metric_mask=%d", who, metric_mask);
  *a = MetricsArgs{};
  for (int i = 0; i < kMaxTopn; ++i) {
    a->topn[i] = i < n_topn ? topn_h[i] : 0;
    if (i < n_topn) {
      ULTR_REQUIRE(a->topn[i] > 0, ULTR_EINVAL, "%s: topn[%d]=%d", who, i, a->topn[i]);
      if (a->topn[i] > Lx) a->topn[i] = Lx;  // topn is clamped to the list size (metric_utils.py:74)
    }
  }
  int P, LPB;
  sort_geometry(Lx, &P, &LPB);
  a->scores = scores; a->labels = labels; a->docids = docids; a->n_rows = n_rows; a->list_weights = list_weights;
  a->u_pred = u_pred; a->B = B; a->Lx = Lx; a->P = P; a->LPB = LPB; a->n_topn = n_topn;
  a->metric_mask = metric_mask; a->max_label = max_label; a->out = out; a->ws = ws;
  a->tie_seed = ties.seed; a->tie_step = ties.step; a->tie_ctr = ties.ctr; a->tie_slot_base = ties.slot_base;
  *nblocks = (unsigned)((P <= 64 && cdiv(B, LPB) > kMetMaxBlocks) ? kMetMaxBlocks : cdiv(B, LPB));
  if (P <= 16 && B >= kLanesMinB)
    *nblocks = (unsigned)(cdiv(B, kMetThreads) > kLanesMaxBlocks ? kLanesMaxBlocks : cdiv(B, kMetThreads));
  return ULTR_OK;
}
}  // namespace ultr

extern "C" int ultr_metrics(const float* scores, const float* labels, const int* docids, int n_rows,
                            const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                            int n_topn, int metric_mask, float max_label, float* out, float* ws, void* stream) {
  MetricsArgs a;
  unsigned nblocks;
  const int rc = prepare_metrics("ultr_metrics", scores, labels, docids, n_rows, list_weights, u_pred, B, Lx, topn_h,
                                 n_topn, metric_mask, max_label, out, ws, &a, &nblocks);
  if (rc) return rc;
  return launch_metrics_now(a, nblocks, as_stream(stream));
}

extern "C" int ultr_metrics_tail_flush(void* stream) {
  if (!g_pending_mtail.on) return ULTR_OK;
  g_pending_mtail.on = false;
  return launch_metrics_tail_now(g_pending_mtail.t, as_stream(stream));
}

// ---- the entry point that allocates nothing ------------------------------------------------------------------------
// For callers that cannot allocate device memory (HIP graph capture, serving loops, harnesses that forbid hipMalloc)
// and do not own a workspace: the partial rows and the arrival counter of metrics_kernel live in a file-scope device
// array.  Every GPU the library runs on holds its own copy of the array at its own address until the process ends;
// the address is looked up once per device and only that pointer is ever handed to a kernel or a memset.
namespace ultr {
constexpr long long kStaticWsFloats = 256 * 1024;  // 1 MiB per device
static __device__ __align__(256) float scratch[kStaticWsFloats];
constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mu;

// stream: the stream the caller is about to launch on, or nullptr when called from the set-up entry point.  The first
// look-up on a device loads the code object's symbol table: that must not happen inside a graph capture, so a capturing
// first call is refused (call ultr_metrics_static_prepare, or ultr_metrics_static itself once, before capturing).
static int static_scratch(float** out, hipStream_t stream, bool have_stream) {
  int dev = -1;
  ULTR_REQUIRE(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices, ULTR_EHIP,
               "ultr_metrics_static: no current device (or ordinal above %d)", kMaxDevices - 1);
  float* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);  // set once per device, never changed afterwards
  if (!p) {
    if (have_stream) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
      ULTR_REQUIRE(cs == hipStreamCaptureStatusNone, ULTR_EINVAL,
                   "ultr_metrics_static: first call on device %d inside a stream capture: call "
                   "ultr_metrics_static_prepare() (or this function once) on that device before capturing", dev);
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    p = g_scratch_ptr[dev];
    if (!p) {
      void* q = nullptr;
      const hipError_t e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
      ULTR_REQUIRE(e == hipSuccess && q, ULTR_EHIP, "ultr_metrics_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
      p = static_cast<float*>(q);
      __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);
    }
  }
  *out = p;
  return ULTR_OK;
}
}  // namespace ultr

extern "C" int ultr_metrics_static_prepare(void) {
  float* ws = nullptr;
  return static_scratch(&ws, nullptr, false);
}

extern "C" long long ultr_metrics_static_max_floats(void) { return kStaticWsFloats; }

extern "C" int ultr_metrics_static(const float* scores, const float* labels, const int* docids, int n_rows,
                                   const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                                   int n_topn, int metric_mask, float max_label, float* out, void* stream) {
  ULTR_REQUIRE(B > 0 && n_topn >= 1 && ultr_metrics_workspace_floats(B, n_topn) <= kStaticWsFloats, ULTR_ELIMIT,
               "ultr_metrics_static: B=%d lists with %d cutoffs need %lld workspace floats, the static array holds %lld "
               "(use ultr_metrics with a workspace of your own)",
               B, n_topn, ultr_metrics_workspace_floats(B > 0 ? B : 1, n_topn), kStaticWsFloats);
  float* ws = nullptr;
  int rc = static_scratch(&ws, as_stream(stream), true);
  if (rc) return rc;
  MetricsArgs a;
  unsigned nblocks;
  rc = prepare_metrics("ultr_metrics_static", scores, labels, docids, n_rows, list_weights, u_pred, B, Lx, topn_h, n_topn,
                       metric_mask, max_label, out, ws, &a, &nblocks);
  if (rc) return rc;
  // the array keeps the last call's contents: the arrival counters (the first kArriveRowsFar floats) must start at zero even if an
  // earlier launch on this device was torn down half-way; the partial rows are overwritten before they are read
  const hipError_t e = hipMemsetAsync(ws, 0, kArriveRowsFar * sizeof(float), as_stream(stream));
  ULTR_REQUIRE(e == hipSuccess, ULTR_EHIP, "ultr_metrics_static: hipMemsetAsync: %s", hipGetErrorString(e));
  return launch_metrics_now(a, nblocks, as_stream(stream));
}

extern "C" int ultr_metrics_ties(const float* scores, const float* labels, const int* docids, int n_rows,
                                 const float* list_weights, int B, int Lx, const int* topn_h, int n_topn, int metric_mask,
                                 float max_label, unsigned long long tie_seed, unsigned int tie_step,
                                 unsigned int* tie_ctr, unsigned int tie_slot_base, float* out, float* ws, void* stream) {
  ULTR_REQUIRE(tie_seed != 0ull, ULTR_EINVAL, "ultr_metrics_ties: tie_seed must not be 0 (0 means no tie shuffling)");
  MetricsArgs a;
  unsigned nblocks;
  const int rc = prepare_metrics("ultr_metrics_ties", scores, labels, docids, n_rows, list_weights, nullptr, B, Lx, topn_h,
                                 n_topn, metric_mask, max_label, out, ws, &a, &nblocks,
                                 TieSpec{tie_seed, tie_step, tie_ctr, tie_slot_base});
  if (rc) return rc;
  return launch_metrics_now(a, nblocks, as_stream(stream));
}

extern "C" int ultr_metrics_ties_deferred(const float* scores, const float* labels, const int* docids, int n_rows,
                                          const float* list_weights, int B, int Lx, const int* topn_h, int n_topn,
                                          int metric_mask, float max_label, unsigned long long tie_seed,
                                          unsigned int tie_step, unsigned int* tie_ctr, unsigned int tie_slot_base,
                                          float* out, float* ws, void* stream) {
  ULTR_REQUIRE(tie_seed != 0ull, ULTR_EINVAL, "ultr_metrics_ties_deferred: tie_seed must not be 0");
  MetricsArgs a;
  unsigned nblocks;
  int rc = prepare_metrics("ultr_metrics_ties_deferred", scores, labels, docids, n_rows, list_weights, nullptr, B, Lx,
                           topn_h, n_topn, metric_mask, max_label, out, ws, &a, &nblocks,
                           TieSpec{tie_seed, tie_step, tie_ctr, tie_slot_base});
  if (rc) return rc;
  if (g_pending_metrics.on) {  // an earlier request nobody carried: it runs now, on its own
    g_pending_metrics.on = false;
    rc = launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, as_stream(stream));
    if (rc) return rc;
  }
  if (a.P > 64) return launch_metrics_now(a, nblocks, as_stream(stream));
  g_pending_metrics.on = true;
  g_pending_metrics.a = a;
  g_pending_metrics.nblocks = nblocks;
  return ULTR_OK;
}

extern "C" int ultr_metrics_deferred(const float* scores, const float* labels, const int* docids, int n_rows,
                                     const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                                     int n_topn, int metric_mask, float max_label, float* out, float* ws, void* stream) {
  MetricsArgs a;
  unsigned nblocks;
  int rc = prepare_metrics("ultr_metrics_deferred", scores, labels, docids, n_rows, list_weights, u_pred, B, Lx, topn_h,
                           n_topn, metric_mask, max_label, out, ws, &a, &nblocks);
  if (rc) return rc;
  if (g_pending_metrics.on) {  // an earlier request nobody carried: it runs now, on its own
    g_pending_metrics.on = false;
    rc = launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, as_stream(stream));
    if (rc) return rc;
  }
  if (a.P > 64) return launch_metrics_now(a, nblocks, as_stream(stream));  // only the short-list path has a body to carry
  g_pending_metrics.on = true;
  g_pending_metrics.a = a;
  g_pending_metrics.nblocks = nblocks;
  return ULTR_OK;
}

extern "C" int ultr_metrics_cancel(void) {
  const int was = g_pending_metrics.on ? 1 : 0;
  g_pending_metrics.on = false;
  return was;
}

extern "C" int ultr_metrics_flush(void* stream) {
  if (!g_pending_metrics.on) return ULTR_OK;
  g_pending_metrics.on = false;
  return launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, as_stream(stream));
}

extern "C" int ultr_sort_by_scores(const float* scores, const float* u, int B, int Lx, int topn, int* perm,
                                   void* stream) {
  ULTR_REQUIRE(scores && perm, ULTR_EINVAL, "ultr_sort_by_scores: null pointer");
  ULTR_REQUIRE(B > 0 && Lx > 0 && topn > 0, ULTR_EINVAL, "ultr_sort_by_scores: bad sizes");
  ULTR_REQUIRE(Lx <= ULTR_MAX_SORT, ULTR_ELIMIT, "ultr_sort_by_scores: list length %d > %d", Lx, ULTR_MAX_SORT);
  if (topn > Lx) topn = Lx;
  int P, LPB;
  sort_geometry(Lx, &P, &LPB);
  const size_t smem = sizeof(float) * (size_t)5 * P * LPB;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_perm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                      96 * 1024);
  {
    ultr::ProfScope prof_("sort_perm_kernel", as_stream(stream));
    hipLaunchKernelGGL(sort_perm_kernel, dim3((unsigned)cdiv(B, LPB)), dim3(kMetThreads), smem, as_stream(stream),
                     scores, u, B, Lx, P, LPB, topn, perm);
  }
  return check_launch("sort_perm_kernel");
}
