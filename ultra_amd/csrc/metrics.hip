// Ranking metrics with per-list bitonic sorts in LDS.
// Reference: ultra/utils/metrics.py:215-251 (_prepare_and_validate_params), :254-284 (MRR),
//   :287-323 (ERR), :445-481 (NDCG), :192-212 (DCG), :158-189 (safe_div, per-list weights);
//   ultra/utils/metric_utils.py:49-86 (sort_by_scores); padding mask base_algorithm.py:64-87.
//
// sort_by_scores composes stable-argsort(uniforms) with top_k (equal values -> lower index
// first), which is exactly a sort by the total order (score desc, uniform asc, index asc); a
// bitonic network under that comparator therefore yields the reference permutation bit for bit.
#include "common.hpp"

namespace ultr {

constexpr int kMetThreads = 256;
constexpr int kMaxTopn = 8;
constexpr int kMaxOut = 3 * kMaxTopn * 2;
constexpr int kMetMaxBlocks = 1024;  // cap on the grid of the short-list path (a block then walks several list groups)
constexpr float kPaddingScore = -100000.f;  // BaseAlgorithm.PADDING_SCORE (base_algorithm.py:25)

struct SortTile {
  float* key;  // [E] prediction (after masking)
  float* u;    // [E] tie-break uniform
  int* idx;    // [E] original position; >= Lx marks a filler element
  float* lab;  // [E] label in ORIGINAL order
  float* ilab; // [E] labels, sorted descending (ideal ranking)
};

__device__ __forceinline__ bool goes_before(float sa, float ua, int ia, float sb, float ub, int ib, int Lx) {
  const bool fa = ia >= Lx, fb = ib >= Lx;  // fillers always sort last
  if (fa != fb) return fb;
  if (fa) return ia < ib;
  if (sa != sb) return sa > sb;
  if (ua != ub) return ua < ub;
  return ia < ib;
}

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
                                          int LPB) {
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
      uu = u_pred ? u_pred[(long long)b * Lx + li] : 0.f;
      id = li;
    }
    t.key[e] = s;
    t.u[e] = uu;
    t.idx[e] = id;
    t.lab[e] = lb;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kMetThreads) metrics_kernel(
    const float* __restrict__ scores, const float* __restrict__ labels, const int* __restrict__ docids, int n_rows,
    const float* __restrict__ list_weights, const float* __restrict__ u_pred, int B, int Lx, int P, int LPB,
    int n_topn, int t0, int t1, int t2, int t3, int t4, int t5, int t6, int t7, int metric_mask, float max_label,
    float* __restrict__ out, float* ws) {
  extern __shared__ float smem[];
  const int E = P * LPB;
  SortTile t;
  t.key = smem;
  t.u = smem + E;
  t.idx = reinterpret_cast<int*>(smem + 2 * E);
  t.lab = smem + 3 * E;
  t.ilab = smem + 4 * E;
  float* s_vals = smem + 5 * E;  // [LPB][nout]
  const int topn[kMaxTopn] = {t0, t1, t2, t3, t4, t5, t6, t7};
  const int n_met = ((metric_mask & ULTR_METRIC_MRR) ? 1 : 0) + ((metric_mask & ULTR_METRIC_NDCG) ? 1 : 0) +
                    ((metric_mask & ULTR_METRIC_ERR) ? 1 : 0);
  const int nout = n_met * n_topn * 2;

  const int tid = threadIdx.x;
  // registers of the short-list path: after the sort, lane (list li, rank r) holds the element ranked r
  float r_lab = 0.f, r_ilab = 0.f, r_w = 1.f;
  if (P <= 64) {
    // Short lists: a list is a group of P lanes inside one wave and never touches LDS.  The grid is capped
    // (kMetMaxBlocks): beyond that a block walks several groups of LPB lists and keeps its outputs in registers,
    // so the number of partial rows the last block has to add stays bounded.
    const int r = tid & (P - 1), li = tid / P;
    constexpr int kAccSlots = kMaxOut / 2;  // outputs per lane: nout / P <= 48 / 2
    float vacc[kAccSlots];
#pragma unroll
    for (int j = 0; j < kAccSlots; ++j) vacc[j] = 0.f;
    for (long long b0 = (long long)blockIdx.x * LPB; b0 < B; b0 += (long long)gridDim.x * LPB) {
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
        uu = u_pred ? u_pred[(long long)b * Lx + r] : 0.f;
        id = r;
      }
      // _prepare_and_validate_params: invalid labels (< 0) -> label 0, prediction -> rowmin - 1e-6
      const bool real = id < Lx;
      const float any_invalid = group_max_rt((real && lb < 0.f) ? 1.f : 0.f, P);
      if (any_invalid != 0.f) {
        const float mn = -group_max_rt(real ? -s : -INFINITY, P);
        if (real && lb < 0.f) {
          lb = 0.f;
          s = mn - 1e-6f;
        }
      }
      const float lab_orig = lb;  // label of the document at ORIGINAL position r
      float il = real ? lb : -INFINITY;
      const bool want_ideal = (metric_mask & ULTR_METRIC_NDCG) != 0;
      for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          const float ps = __shfl_xor(s, j, 64), pu = __shfl_xor(uu, j, 64);
          const int pi = __shfl_xor(id, j, 64);
          const bool lower = (r & j) == 0, first_low = (r & k) == 0;
          const bool want_first = lower == first_low;  // this lane keeps the element that goes first
          const bool mine_first = goes_before(s, uu, id, ps, pu, pi, Lx);
          if (mine_first != want_first) {
            s = ps; uu = pu; id = pi;
          }
          if (want_ideal) {
            const float pl = __shfl_xor(il, j, 64);
            // descending labels; on equal values either copy is the same number
            const bool keep = want_first ? (il >= pl) : (il <= pl);
            if (!keep) il = pl;
          }
        }
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
      auto gsum = [&](float v) { return group_sum_rt(v, P); };
      const float rel_sum = gsum(gain);
      float mrr = (live && lbr >= 1.f) ? 1.f / (float)(r + 1) : 0.f;
      mrr = group_max_rt(mrr, P);
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
      for (int off = P; off < 64; off <<= 1) val += __shfl_xor(val, off, 64);  // the other lists of this wave
      if ((tid & 63) < P) s_w[(tid >> 6) * nout + o] = val;
    }
  } else {
  load_tile(t, scores, labels, docids, n_rows, u_pred, B, Lx, P, E, LPB);

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
  // block partial sums in fixed order, then the last block divides by B (reduce_mean over lists)
  __shared__ float s_acc[kMaxOut];
  if (tid < nout) {
    float s = 0.f;
    const int nrow = P <= 64 ? kMetThreads / 64 : LPB;  // short lists: one row per wave, long lists: one per list
    for (int l = 0; l < nrow; ++l) s += s_vals[l * nout + tid];
    s_acc[tid] = s;
  }
  unsigned int* counter = reinterpret_cast<unsigned int*>(ws);
  float* rows = ws + 8;
  __syncthreads();
  if (tid < nout) rows[(long long)blockIdx.x * nout + tid] = s_acc[tid];
  if (last_block_arrive(counter, gridDim.x)) {
    __shared__ float s_tmp[kMetThreads];
    reduce_rows(rows, (int)gridDim.x, nout, 1 << 30, s_tmp, s_acc);
    if (tid < nout && ((tid & 1) == 0 || list_weights)) out[tid] = s_acc[tid] / (float)B;
  }
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
  return 8 + (long long)(B > 0 ? B : 1) * 3 * (n_topn > 0 ? n_topn : 1) * 2;
}

extern "C" int ultr_metrics(const float* scores, const float* labels, const int* docids, int n_rows,
                            const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                            int n_topn, int metric_mask, float max_label, float* out, float* ws, void* stream) {
  ULTR_REQUIRE(scores && labels && out && ws && topn_h, ULTR_EINVAL, "ultr_metrics: null pointer");
  ULTR_REQUIRE(B > 0 && Lx > 0, ULTR_EINVAL, "ultr_metrics: bad sizes B=%d Lx=%d", B, Lx);
  ULTR_REQUIRE(Lx <= ULTR_MAX_SORT, ULTR_ELIMIT, "ultr_metrics: list length %d > %d", Lx, ULTR_MAX_SORT);
  ULTR_REQUIRE(n_topn >= 1 && n_topn <= kMaxTopn, ULTR_ELIMIT, "ultr_metrics: n_topn=%d not in [1,%d]", n_topn,
               kMaxTopn);
  ULTR_REQUIRE((metric_mask & ~7) == 0 && metric_mask != 0, ULTR_EINVAL, "ultr_metrics: metric_mask=%d", metric_mask);
  int tn[kMaxTopn];
  for (int i = 0; i < kMaxTopn; ++i) {
    tn[i] = i < n_topn ? topn_h[i] : 0;
    if (i < n_topn) {
      ULTR_REQUIRE(tn[i] > 0, ULTR_EINVAL, "ultr_metrics: topn[%d]=%d", i, tn[i]);
      if (tn[i] > Lx) tn[i] = Lx;  // topn is clamped to the list size (metric_utils.py:74)
    }
  }
  int P, LPB;
  sort_geometry(Lx, &P, &LPB);
  const int n_met = ((metric_mask & 1) ? 1 : 0) + ((metric_mask & 2) ? 1 : 0) + ((metric_mask & 4) ? 1 : 0);
  const size_t smem = sizeof(float) * ((size_t)5 * P * LPB + (size_t)LPB * n_met * n_topn * 2);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(metrics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_perm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        96 * 1024);
    attr_set = true;
  }
  {
    ultr::ProfScope prof_("metrics_kernel", as_stream(stream));
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)((P <= 64 && cdiv(B, LPB) > kMetMaxBlocks) ? kMetMaxBlocks : cdiv(B, LPB))), dim3(kMetThreads), smem, as_stream(stream), scores,
                     labels, docids, n_rows, list_weights, u_pred, B, Lx, P, LPB, n_topn, tn[0], tn[1], tn[2], tn[3],
                     tn[4], tn[5], tn[6], tn[7], metric_mask, max_label, out, ws);
  }
  return check_launch("metrics_kernel");
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
