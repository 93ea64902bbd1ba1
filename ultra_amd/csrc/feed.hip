// Click simulation and batch assembly on the device.
// Reference: ultra/utils/click_models.py:86-116 (PBM), :204-241 (cascade);
//            ultra/input_layer/click_simulation_feed.py:75-162 (get_batch).
#include <type_traits>
#include "feed_body.hpp"

namespace ultr {

// draw d of the Philox stream (c0, c1, *, c3): words (2d, 2d+1) of the block sequence
struct PhiloxDraws {
  uint32_t c0, c1, c3, k0, k1;
  int cur_block;
  Philox4 blk;
  __device__ __forceinline__ void init(uint32_t a, uint32_t b, uint32_t d, unsigned long long seed) {
    c0 = a; c1 = b; c3 = d;
    k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    cur_block = -1;
  }
  __device__ __forceinline__ double draw(int d) {
    const int b = d >> 1;
    if (b != cur_block) {
      blk = philox4x32_10(c0, c1, (uint32_t)b, c3, k0, k1);
      cur_block = b;
    }
    return (d & 1) ? u53(blk.w2, blk.w3) : u53(blk.w0, blk.w1);
  }
};

// One thread per list.  labels/clicks/uniforms are list-major [n][L].
__global__ void __launch_bounds__(256) click_kernel(int model, const float* __restrict__ labels, long long n, int L,
                                                    ClickTables t, const double* __restrict__ uniforms,
                                                    unsigned long long seed, unsigned int step,
                                                    float* __restrict__ clicks) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PhiloxDraws ph;
  ph.init((uint32_t)i, 0u, step, seed);
  bool has_click = false;
  int last = -1;
  for (int r = 0; r < L; ++r) {
    const float lab = labels[i * L + r];
    const double u = uniforms ? uniforms[i * L + r] : ph.draw(r);  // a draw at EVERY rank (click_models.py:208)
    const bool c = u < click_threshold(t, model, r, lab, last);
    clicks[i * L + r] = (c && !(model == ULTR_CLICK_CASCADE && has_click)) ? 1.f : 0.f;
    has_click = has_click || c;
    if (c) last = r;
  }
}

// Short lists (L <= 16): one lane per (list, rank), sixteen lanes per list, so that a wave reads four whole lists
// with one coalesced load and every lane evaluates one Philox block and one float64 threshold instead of L of them.
// Same draws as click_kernel (draw r of the stream (list, 0, *, step): words (2r, 2r+1) of its block sequence).
__global__ void __launch_bounds__(256) click_small_kernel(int model, const float* __restrict__ labels, long long n, int L,
                                                          ClickTables t, const double* __restrict__ uniforms,
                                                          unsigned long long seed, unsigned int step,
                                                          float* __restrict__ clicks) {
  const int lane = threadIdx.x & 63, r = lane & 15, gbase = lane & 48;
  const long long nblk = ((long long)n + 15) / 16;
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long i = blk * 16 + (threadIdx.x >> 4);
    const bool live = i < n && r < L;
    const float lab = live ? labels[i * L + r] : 0.f;
    double u = 2.0;
    if (live) {
      if (uniforms) {
        u = uniforms[i * L + r];
      } else {
        const Philox4 bd = philox4x32_10((uint32_t)i, 0u, (uint32_t)(r >> 1), step, (uint32_t)seed, (uint32_t)(seed >> 32));
        u = (r & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
      }
    }
    bool c = false;
    if (model == ULTR_CLICK_UBM) {
      int last = -1;
      for (int q = 0; q < L; ++q) {  // the ranks of a list resolve in order: each needs the last click above it
        const bool cq = live && r == q && u < click_threshold(t, model, r, lab, last);
        if ((__ballot(cq) >> (gbase + q)) & 1ull) last = q;
        c = c || cq;
      }
    } else {
      c = live && u < click_threshold(t, model, r, lab, -1);
    }
    if (model == ULTR_CLICK_CASCADE) {  // only the first click of the list is kept (a draw was made at every rank)
      const unsigned int m = (unsigned int)((__ballot(c) >> gbase) & 0xFFFFull);
      c = c && (m & ((1u << r) - 1u)) == 0u;
    }
    if (live) clicks[i * L + r] = c ? 1.f : 0.f;
  }
}

// Very many short lists (exactly LT <= 16 ranks, Philox draws): one lane per LIST.  The 16-lane kernel above evaluates
// one whole Philox block per (list, rank) and uses half of it, and runs out of issue slots at ~1.4 TB/s; here a list
// costs LT / 2 blocks.  The list-major rows of 256 lists are one contiguous run: they cross LDS so that every global
// access is a full float4 row per lane group (row stride LT | 1 words in LDS: conflict-free for the per-list walk).
// Same draws and thresholds as click_kernel: bit-identical clicks.
constexpr long long kClickListsMinN = 32768;
template <int LT>
__global__ void __launch_bounds__(256) click_lists_kernel(int model, const float* __restrict__ labels, long long n,
                                                          ClickTables t, unsigned long long seed, unsigned int step,
                                                          float* __restrict__ clicks) {
  constexpr int PS = LT | 1;
  __shared__ float s_t[256 * PS];
  const int tid = threadIdx.x;
  const long long ntile = (n + 255) / 256;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const long long i0 = tile * 256;
    const int cnt = (int)(n - i0 < 256 ? n - i0 : 256);        // lists in this tile
    const float* __restrict__ src = labels + i0 * LT;           // 16-byte aligned: i0 is a multiple of 256
    float* __restrict__ dst = clicks + i0 * LT;
    const int nel = cnt * LT;
    for (int e4 = tid; e4 < (nel >> 2); e4 += 256) {
      const float4 v = reinterpret_cast<const float4*>(src)[e4];
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = 4 * e4 + q;
        s_t[(e / LT) * PS + e % LT] = vv[q];
      }
    }
    for (int e = (nel & ~3) + tid; e < nel; e += 256) s_t[(e / LT) * PS + e % LT] = src[e];
    __syncthreads();
    if (tid < cnt) {
      const long long i = i0 + tid;
      bool has_click = false;
      int last = -1;
      Philox4 bd{};
#pragma unroll
      for (int r = 0; r < LT; ++r) {
        if ((r & 1) == 0) bd = philox4x32_10((uint32_t)i, 0u, (uint32_t)(r >> 1), step, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double u = (r & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);  // a draw at EVERY rank (click_models.py:208)
        const float lab = s_t[tid * PS + r];
        const bool c = u < click_threshold(t, model, r, lab, last);
        s_t[tid * PS + r] = (c && !(model == ULTR_CLICK_CASCADE && has_click)) ? 1.f : 0.f;
        has_click = has_click || c;
        if (c) last = r;
      }
    }
    __syncthreads();
    for (int e4 = tid; e4 < (nel >> 2); e4 += 256) {
      float vv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = 4 * e4 + q;
        vv[q] = s_t[(e / LT) * PS + e % LT];
      }
      reinterpret_cast<float4*>(dst)[e4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int e = (nel & ~3) + tid; e < nel; e += 256) dst[e] = s_t[(e / LT) * PS + e % LT];
    __syncthreads();
  }
}
template <int LT>
static void launch_click_lists(int model, const float* labels, long long n, int L, const ClickTables& t,
                               unsigned long long seed, unsigned int step, float* clicks, unsigned nb, hipStream_t st) {
  if (L == LT) hipLaunchKernelGGL(click_lists_kernel<LT>, dim3(nb), dim3(256), 0, st, model, labels, n, t, seed, step, clicks);
  else if constexpr (LT > 1) launch_click_lists<LT - 1>(model, labels, n, L, t, seed, step, clicks, nb, st);
}

// Evaluate one candidate: pick the list, sample its clicks.  Returns the click bit mask.
template <typename DrawFn>
__device__ __forceinline__ unsigned long long eval_candidate(DrawFn&& draw, const int* __restrict__ initial_list,
                                                             const float* __restrict__ labels, int Q, int Lmax,
                                                             int L, int click_model, const ClickTables& t,
                                                             int* out_i) {
  int i = (int)(draw(0) * (double)Q);  // int(random.random() * length), click_simulation_feed.py:132
  i = i < Q ? i : Q - 1;
  *out_i = i;
  unsigned long long mask = 0ull;
  bool has_click = false;
  int last = -1;
  for (int r = 0; r < L; ++r) {
    const int doc = initial_list[(long long)i * Lmax + r];
    const float lab = doc < 0 ? 0.f : labels[(long long)i * Lmax + r];  // :80-82
    if (click_model == ULTR_CLICK_ORACLE) {  // oracle_mode: click_list = label_list, no click draws (:84-85)
      if (lab != 0.f) mask |= (1ull << r);
      continue;
    }
    const bool c = draw(1 + r) < click_threshold(t, click_model, r, lab, last);
    if (c && !(click_model == ULTR_CLICK_CASCADE && has_click)) mask |= (1ull << r);
    has_click = has_click || c;
    if (c) last = r;
  }
  return mask;
}

__device__ __forceinline__ void emit_slot(int slot, int i, unsigned long long mask, const int* __restrict__ initial_list,
                                          const float* __restrict__ labels, int click_model,
                                          int Lmax, int L, int B, int n_rows, int* __restrict__ out_docids,
                                          float* __restrict__ out_clicks, int* __restrict__ out_list_idx) {
  for (int r = 0; r < L; ++r) {
    const int doc = initial_list[(long long)i * Lmax + r];
    out_docids[(long long)r * B + slot] = doc < 0 ? n_rows : doc;  // pad -> the zero row (:140-143)
    float c = ((mask >> r) & 1ull) ? 1.f : 0.f;
    if (click_model == ULTR_CLICK_ORACLE) c = doc < 0 ? 0.f : labels[(long long)i * Lmax + r];
    out_clicks[(long long)r * B + slot] = c;
  }
  out_list_idx[slot] = i;
}

__global__ void __launch_bounds__(256) assemble_slots_kernel(AssembleArgs a) { assemble_slots_body(a, blockIdx.x, gridDim.x); }

// mode 1: a single workgroup consumes the supplied candidates IN ORDER (bit-exact with the
// reference's sequential random.random() stream).  Every loop bound is block-uniform.
__global__ void __launch_bounds__(1024) assemble_sequential_kernel(
    const int* __restrict__ initial_list, const float* __restrict__ labels, int Q, int Lmax, int L, int B,
    int click_model, ClickTables t, const double* __restrict__ uniforms, int n_cand, int slot_start,
    int check_validation, int n_rows, int* __restrict__ out_docids, float* __restrict__ out_clicks,
    int* __restrict__ out_list_idx, int* __restrict__ info) {
  __shared__ int s_wave_cnt[16];
  __shared__ int s_filled, s_consumed;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) {
    s_filled = slot_start;
    s_consumed = -1;
  }
  __syncthreads();
  for (int base = 0; base < n_cand; base += blockDim.x) {
    const int filled = s_filled;  // block-uniform
    if (filled >= B) break;
    const int c = base + tid;
    bool accept = false;
    int i = 0;
    unsigned long long mask = 0ull;
    if (c < n_cand) {
      const double* u = uniforms + (long long)c * (1 + L);
      mask = eval_candidate([&](int d) { return u[d]; }, initial_list, labels, Q, Lmax, L, click_model, t, &i);
      accept = !check_validation || mask != 0ull;
    }
    const unsigned long long bal = __ballot(accept);
    const int before_in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave_cnt[wid] = __popcll(bal);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < nw; ++w) {
      const int cw = s_wave_cnt[w];
      if (w < wid) before += cw;
      total += cw;
    }
    const int slot = filled + before + before_in_wave;
    if (accept && slot < B) {
      emit_slot(slot, i, mask, initial_list, labels, click_model, Lmax, L, B, n_rows, out_docids, out_clicks, out_list_idx);
      if (slot == B - 1) s_consumed = c + 1;
    }
    __syncthreads();
    if (tid == 0) s_filled = min(B, filled + total);
    __syncthreads();
  }
  if (tid == 0) {
    info[0] = s_consumed >= 0 ? s_consumed : n_cand;
    info[1] = s_filled;
  }
}


// ---------------------------------------------------------------------------------------------
// Online simulation (deterministic_online_simulation_feed.py:106-194): every list is re-ranked by the current
// model's scores and the click model runs on the result.  One workgroup per list; the list lives in LDS.
//   list_len = 1 + last position holding a real document (:131-137)
//   order    = positions < list_len by score, descending, equal scores keep their order (Python's stable
//              sorted(reverse=True), :156-161): rank(p) = #{q : s_q > s_p or (s_q == s_p and q < p)}
//   docids / labels of positions < list_len are permuted (:163-167, 182-183), the rest stay
//   clicks for ranks < min(L, list_len) on the re-ranked labels (:169-180), re-sampled while the list has no click
//   (check_validation, at most max_rounds more rounds, :176-180); labels of ranks >= L become 0 (:186-187)
// ---------------------------------------------------------------------------------------------
struct OnlineArgs {
  const int* docids; const float* labels; const float* scores;  // [Lmax][B]
  int B, Lmax, L, n_rows, click_model;
  ClickTables t;
  const double* uniforms;  // [B][L] or NULL
  int stochastic; float tau; const double* rank_uniforms;  // Plackett-Luce ranking (stochastic feed): [B][Lmax] or NULL
  unsigned long long seed; unsigned int step; const unsigned int* step_ctr; unsigned int slot_base;
  int check_validation, max_rounds;
  int* out_docids; float* out_clicks; int* out_perm;  // [Lmax][B]; out_perm may be NULL
};

__global__ void __launch_bounds__(256) online_rerank_click_kernel(OnlineArgs a) {
  extern __shared__ double smem_d[];  // sort keys (double) | labels | re-ranked labels | docids | order
  const int Lmax = a.Lmax, B = a.B, tid = threadIdx.x;
  double* s_score = smem_d;
  float* smem = reinterpret_cast<float*>(smem_d + Lmax);
  float* s_lab = smem;
  float* s_rlab = smem + Lmax;
  int* s_doc = reinterpret_cast<int*>(smem + 2 * Lmax);
  int* s_order = reinterpret_cast<int*>(smem + 3 * Lmax);
  __shared__ int s_len;
  const int b = blockIdx.x;
  if (tid == 0) s_len = 0;
  __syncthreads();
  for (int p = tid; p < Lmax; p += blockDim.x) {
    const long long e = (long long)p * B + b;
    const int d = a.docids[e];
    s_doc[p] = d;
    double key = (double)a.scores[e];
    if (a.stochastic) {
      // Plackett-Luce sample of the ranking (stochastic_online_simulation_feed.py:140-158: documents drawn without
      // replacement with p ~ exp(tau * score)) as a sort: Gumbel noise on tau * score, descending.  Same
      // distribution; np.random.choice's own stream is not reproduced.
      double u;
      if (a.rank_uniforms) {
        u = a.rank_uniforms[(long long)b * Lmax + p];
      } else {
        const unsigned int stp = a.step + (a.step_ctr ? *a.step_ctr : 0u);
        const Philox4 bd = philox4x32_10(a.slot_base + (uint32_t)b, 0xFFFFFFFFu, (uint32_t)(p >> 1), stp,
                                         (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        u = (p & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
      }
      u = u > 8.673617379884035e-19 ? u : 8.673617379884035e-19;  // 2^-60: keeps the noise finite
      key = (double)a.tau * key - log(-log(u));
    }
    s_score[p] = key;
    s_lab[p] = a.labels[e];
    if (d < a.n_rows) atomicMax(&s_len, p + 1);
  }
  __syncthreads();
  const int len = s_len;
  for (int p = tid; p < len; p += blockDim.x) {
    const double sp = s_score[p];
    int rank = 0;
    for (int q = 0; q < len; ++q) {
      const double sq = s_score[q];
      rank += (sq > sp || (sq == sp && q < p)) ? 1 : 0;
    }
    s_order[rank] = p;
  }
  __syncthreads();
  for (int r = tid; r < Lmax; r += blockDim.x) {
    const int src = r < len ? s_order[r] : r;
    const long long e = (long long)r * B + b;
    a.out_docids[e] = s_doc[src];
    s_rlab[r] = s_lab[src];
    if (a.out_perm) a.out_perm[e] = src;
    if (r >= a.L || r >= len) a.out_clicks[e] = r < len ? 0.f : s_lab[r];  // beyond the shown ranks / beyond the list
  }
  __syncthreads();
  if (tid >= 64) return;
  // wave 0: the click model on ranks < min(L, len), lane = rank
  const int Lc = a.L < len ? a.L : len, lane = tid;
  if (Lc == 0) return;
  const bool live = lane < Lc;
  const float lab = live ? s_rlab[lane] : 0.f;
  unsigned long long mask = 0ull;
  if (a.click_model == ULTR_CLICK_ORACLE) {
    if (live) a.out_clicks[(long long)lane * B + b] = lab;  // oracle_mode: the labels themselves (:170-171)
    return;
  }
  const unsigned int step = a.step + (a.step_ctr ? *a.step_ctr : 0u);
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), slot = a.slot_base + (uint32_t)b;
  const unsigned long long lmask = Lc >= 64 ? ~0ull : ((1ull << Lc) - 1ull);
  const int rounds = a.uniforms ? 1 : 1 + (a.check_validation ? a.max_rounds : 0);
  for (int round = 0; round < rounds; ++round) {
    double u = 2.0;
    if (live) {
      if (a.uniforms) {
        u = a.uniforms[(long long)b * a.L + lane];
      } else {  // draw `lane` of the stream (slot, round, *, step): words (2 lane, 2 lane + 1) of its block sequence
        const Philox4 bd = philox4x32_10(slot, (uint32_t)round, (uint32_t)(lane >> 1), step, k0, k1);
        u = (lane & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
      }
    }
    bool c = false;
    if (a.click_model == ULTR_CLICK_UBM) {
      int last_click = -1;
      for (int r = 0; r < Lc; ++r) {
        const bool cr = live && lane == r && u < click_threshold(a.t, a.click_model, lane, lab, last_click);
        if (__ballot(cr)) last_click = r;
        c = c || cr;
      }
    } else {
      c = live && u < click_threshold(a.t, a.click_model, lane, lab, -1);
    }
    mask = __ballot(c) & lmask;
    if (a.click_model == ULTR_CLICK_CASCADE) mask &= (~mask + 1ull);  // only the first click counts
    if (mask != 0ull || !a.check_validation) break;
  }
  if (live) a.out_clicks[(long long)lane * B + b] = ((mask >> lane) & 1ull) ? 1.f : 0.f;
}


// The same for short lists (Lmax <= 16): 16 lanes per list, four lists per wave, sixteen per workgroup -- the list
// lives in registers, ranks come from 16 group shuffles, and every group re-samples its clicks on its own (the wave
// loops until its last list is done).  Same draws and results as online_rerank_click_kernel.
__global__ void __launch_bounds__(256) online_rerank_click_small_kernel(OnlineArgs a) {
  const int Lmax = a.Lmax, B = a.B, tid = threadIdx.x, lane = tid & 63;
  const int r = lane & 15, gbase = lane & 48;
  const long long b = (long long)blockIdx.x * 16 + (tid >> 4);
  const bool in_batch = b < B, has_pos = in_batch && r < Lmax;
  const long long e = (long long)r * B + (in_batch ? b : 0);
  const int doc = has_pos ? a.docids[e] : a.n_rows;
  const float lab = has_pos ? a.labels[e] : 0.f;
  double key = has_pos ? (double)a.scores[e] : 0.0;
  const unsigned int step = a.step + (a.step_ctr ? *a.step_ctr : 0u);
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), slot = a.slot_base + (uint32_t)b;
  if (a.stochastic && has_pos) {
    double u;
    if (a.rank_uniforms) {
      u = a.rank_uniforms[b * Lmax + r];
    } else {
      const Philox4 bd = philox4x32_10(slot, 0xFFFFFFFFu, (uint32_t)(r >> 1), step, k0, k1);
      u = (r & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
    }
    u = u > 8.673617379884035e-19 ? u : 8.673617379884035e-19;
    key = (double)a.tau * key - log(-log(u));
  }
  const int len = (int)group_max<16>(has_pos && doc < a.n_rows ? (float)(r + 1) : 0.f);
  // rank of position r among the positions < len, then the position that lands on rank r
  int rank = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const double kq = __shfl(key, gbase + q, 64);
    rank += (q < len && (kq > key || (kq == key && q < r))) ? 1 : 0;
  }
  int src = r;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rq = __shfl(rank, gbase + q, 64);
    if (r < len && q < len && rq == r) src = q;
  }
  const int rdoc = __shfl(doc, gbase + src, 64);
  const float rlab = __shfl(lab, gbase + src, 64);
  if (has_pos) {
    a.out_docids[e] = rdoc;
    if (a.out_perm) a.out_perm[e] = src;
  }
  const int Lc = a.L < len ? a.L : len;
  const bool live = has_pos && r < Lc;
  float click = 0.f;
  if (a.click_model == ULTR_CLICK_ORACLE) {
    click = live ? rlab : 0.f;
  } else {
    const int rounds = a.uniforms ? 1 : 1 + (a.check_validation ? a.max_rounds : 0);
    bool done = !in_batch || Lc == 0;
    unsigned int mask = 0u;
    // the threshold of a (rank, label) does not change between re-sampling rounds, unless it depends on the last click
    const double thr = a.click_model != ULTR_CLICK_UBM ? click_threshold(a.t, a.click_model, r, rlab, -1) : 0.0;
    for (int round = 0; round < rounds; ++round) {
      if (__ballot(!done) == 0ull) break;
      double u = 2.0;
      if (live && !done) {
        if (a.uniforms) {
          u = a.uniforms[b * a.L + r];
        } else {
          const Philox4 bd = philox4x32_10(slot, (uint32_t)round, (uint32_t)(r >> 1), step, k0, k1);
          u = (r & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
        }
      }
      bool c = false;
      if (a.click_model == ULTR_CLICK_UBM) {
        int last_click = -1;
        for (int q = 0; q < 16; ++q) {  // wave-uniform bound: the ranks of every list resolve in order
          const bool cq = live && !done && r == q && u < click_threshold(a.t, a.click_model, r, rlab, last_click);
          const unsigned long long bq = __ballot(cq);
          if ((bq >> (gbase + q)) & 1ull) last_click = q;
          c = c || cq;
        }
      } else {
        c = live && !done && u < thr;
      }
      unsigned int m = (unsigned int)((__ballot(c) >> gbase) & 0xFFFFull);
      if (a.click_model == ULTR_CLICK_CASCADE) m &= (~m + 1u);  // only the first click counts
      if (!done) {
        mask = m;
        if (m != 0u || !a.check_validation) done = true;
      }
    }
    click = live && ((mask >> r) & 1u) ? 1.f : 0.f;
  }
  if (has_pos) a.out_clicks[e] = r < len ? click : lab;  // ranks beyond the shown ones are 0, padding keeps its label
}

// Very many short lists (exactly LT <= 16 positions, Philox draws): one lane per LIST.  The position-major arrays make
// every load and store a full row per wave; ranks come from counting over the list's pairs in registers, the click model
// walks the ranks in order (so the user browsing model needs no ballots), and every lane re-samples its own list until
// it has a click.  Same keys, draws and thresholds as the two kernels above: identical outputs.
// STOCH: Plackett-Luce keys (float64 Gumbel noise on tau * score); otherwise the scores themselves order the list (a
// float compares like its float64 image).
constexpr int kOnlineListsMinB = 131072;  // below, the 16-lane kernel wins: re-sampling rounds diverge less over 4 lists than over 64
template <int LT, bool STOCH>
__global__ void __launch_bounds__(256) online_rerank_click_lists_kernel(OnlineArgs a) {
  typedef typename std::conditional<STOCH, double, float>::type key_t;
  const int B = a.B;
  const unsigned int step = a.step + (a.step_ctr ? *a.step_ctr : 0u);
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const int rounds = 1 + (a.check_validation ? a.max_rounds : 0);
#pragma unroll 1
  for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < B; b += (long long)gridDim.x * 256) {
    const uint32_t slot = a.slot_base + (uint32_t)b;
    int doc[LT];
    float lab[LT];
    key_t key[LT];
#pragma unroll
    for (int p = 0; p < LT; ++p) {
      const long long e = (long long)p * B + b;
      doc[p] = a.docids[e];
      lab[p] = a.labels[e];
      key[p] = (key_t)a.scores[e];
    }
    if (STOCH) {
      Philox4 bd{};
#pragma unroll
      for (int p = 0; p < LT; ++p) {
        if ((p & 1) == 0) bd = philox4x32_10(slot, 0xFFFFFFFFu, (uint32_t)(p >> 1), step, k0, k1);
        double u = (p & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
        u = u > 8.673617379884035e-19 ? u : 8.673617379884035e-19;  // 2^-60: keeps the noise finite
        key[p] = (key_t)((double)a.tau * (double)key[p] - log(-log(u)));
      }
    }
    int len = 0;
#pragma unroll
    for (int p = 0; p < LT; ++p) len = doc[p] < a.n_rows ? p + 1 : len;
    // rank(p) = #{q < len : key_q > key_p or (key_q == key_p and q < p)}, for p < len
    int rank[LT];
#pragma unroll
    for (int p = 0; p < LT; ++p) rank[p] = 0;
#pragma unroll
    for (int p = 1; p < LT; ++p) {
#pragma unroll
      for (int q = 0; q < p; ++q) {  // q < p: on equal keys q goes first
        const bool both = p < len;  // q < p < len
        rank[p] += (both && key[q] >= key[p]) ? 1 : 0;
        rank[q] += (both && key[p] > key[q]) ? 1 : 0;
      }
      __builtin_amdgcn_sched_barrier(0);  // one row of pairs at a time (see metrics_lanes_kernel)
    }
    // the position that lands on rank r, its document and label
    float rlab[LT];
#pragma unroll
    for (int r = 0; r < LT; ++r) {
      int src = r, rd = doc[r];
      float rl = lab[r];
#pragma unroll
      for (int p = 0; p < LT; ++p) {
        const bool hit = r < len && p < len && rank[p] == r;
        src = hit ? p : src;
        rd = hit ? doc[p] : rd;
        rl = hit ? lab[p] : rl;
      }
      rlab[r] = rl;
      const long long e = (long long)r * B + b;
      a.out_docids[e] = rd;
      if (a.out_perm) a.out_perm[e] = src;
    }
    const int Lc = a.L < len ? a.L : len;
    unsigned int mask = 0u;
    if (a.click_model != ULTR_CLICK_ORACLE && Lc > 0) {  // (oracle mode: the labels themselves, written below)
      // the thresholds of a list do not change between re-sampling rounds, unless they depend on the last click
      const bool fixed_thr = a.click_model != ULTR_CLICK_UBM;
      double thr[LT];
#pragma unroll
      for (int r = 0; r < LT; ++r) thr[r] = fixed_thr ? click_threshold(a.t, a.click_model, r, rlab[r], -1) : 0.0;
      for (int round = 0; round < rounds; ++round) {
        unsigned int m = 0u;
        int last = -1;
        bool has_click = false;
        Philox4 bd{};
#pragma unroll
        for (int r = 0; r < LT; ++r) {
          if ((r & 1) == 0) bd = philox4x32_10(slot, (uint32_t)round, (uint32_t)(r >> 1), step, k0, k1);
          const double u = (r & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
          const bool c = r < Lc && u < (fixed_thr ? thr[r] : click_threshold(a.t, a.click_model, r, rlab[r], last));
          if (c && !(a.click_model == ULTR_CLICK_CASCADE && has_click)) m |= 1u << r;
          has_click = has_click || c;
          if (c) last = r;
        }
        mask = m;
        if (m != 0u || !a.check_validation) break;
      }
    }
#pragma unroll
    for (int r = 0; r < LT; ++r) {
      float click = ((mask >> r) & 1u) ? 1.f : 0.f;
      if (a.click_model == ULTR_CLICK_ORACLE) click = r < Lc ? rlab[r] : 0.f;
      a.out_clicks[(long long)r * B + b] = r < len ? click : lab[r];  // ranks beyond the shown ones are 0, padding keeps its label
    }
  }
}
template <int LT>
static void launch_online_lists(const OnlineArgs& a, unsigned nb, hipStream_t st) {
  if (a.Lmax == LT) {
    if (a.stochastic) hipLaunchKernelGGL((online_rerank_click_lists_kernel<LT, true>), dim3(nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((online_rerank_click_lists_kernel<LT, false>), dim3(nb), dim3(256), 0, st, a);
  } else if constexpr (LT > 1) {
    launch_online_lists<LT - 1>(a, nb, st);
  }
}

}  // namespace ultr

using namespace ultr;

static int click_model_ok(const char* who, int click_model, int n_exam, int n_cp) {
  ULTR_REQUIRE(click_model == ULTR_CLICK_PBM || click_model == ULTR_CLICK_CASCADE || click_model == ULTR_CLICK_ORACLE ||
                   click_model == ULTR_CLICK_TABLE || click_model == ULTR_CLICK_UBM,
               ULTR_EINVAL, "%s: click model %d is not on the device path", who, click_model);
  ULTR_REQUIRE(n_exam > 0 && n_cp > 0, ULTR_EINVAL, "%s: empty click tables (n_exam=%d, n_click_prob=%d)", who, n_exam, n_cp);
  ULTR_REQUIRE(click_model != ULTR_CLICK_UBM || n_exam >= 2, ULTR_EINVAL,
               "%s: the user browsing model needs at least two rows of examination probabilities", who);
  return ULTR_OK;
}

extern "C" int ultr_click_sample(int click_model, const float* labels, long long n, int L, const double* exam,
                                 int n_exam, const double* click_prob, int n_click_prob, const double* uniforms,
                                 unsigned long long seed, unsigned int stream_step, float* clicks, void* stream) {
  ULTR_REQUIRE(labels && clicks && exam && click_prob, ULTR_EINVAL, "ultr_click_sample: null pointer");
  ULTR_REQUIRE(n >= 0 && L > 0, ULTR_EINVAL, "ultr_click_sample: bad sizes n=%lld L=%d", n, L);
  int rc = click_model_ok("ultr_click_sample", click_model, n_exam, n_click_prob);
  if (rc) return rc;
  ULTR_REQUIRE(click_model != ULTR_CLICK_ORACLE, ULTR_EINVAL, "ultr_click_sample: oracle mode copies labels, nothing to sample");
  ULTR_REQUIRE(n < (1ll << 32), ULTR_ELIMIT, "ultr_click_sample: n=%lld exceeds the 32-bit Philox list counter", n);
  if (n == 0) return ULTR_OK;
  ClickTables t{exam, n_exam, click_prob, n_click_prob};
  if (L <= 16 && !uniforms && n >= kClickListsMinN && ((uintptr_t)labels % 16 == 0) && ((uintptr_t)clicks % 16 == 0)) {
    const long long blocks = cdiv(n, 256) < 8192 ? cdiv(n, 256) : 8192;
    ultr::ProfScope prof_("click_kernel", as_stream(stream));
    launch_click_lists<16>(click_model, labels, n, L, t, seed, stream_step, clicks, (unsigned)blocks, as_stream(stream));
    return check_launch("click_lists_kernel");
  }
  if (L <= 16) {
    const long long blocks = cdiv(n, 16) < 65536 ? cdiv(n, 16) : 65536;  // capped grid, block-strided beyond
    ultr::ProfScope prof_("click_kernel", as_stream(stream));
    hipLaunchKernelGGL(click_small_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), click_model, labels, n,
                       L, t, uniforms, seed, stream_step, clicks);
    return check_launch("click_small_kernel");
  }
  {
    ultr::ProfScope prof_("click_kernel", as_stream(stream));
    hipLaunchKernelGGL(click_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), click_model, labels, n,
                       L, t, uniforms, seed, stream_step, clicks);
  }
  return check_launch("click_kernel");
}

extern "C" int ultr_click_pbm(const float* labels, long long n, int L, const double* exam, int n_exam,
                              const double* click_prob, int n_click_prob, const double* uniforms,
                              unsigned long long seed, unsigned int stream_step, float* clicks, void* stream) {
  return ultr_click_sample(ULTR_CLICK_PBM, labels, n, L, exam, n_exam, click_prob, n_click_prob, uniforms, seed,
                           stream_step, clicks, stream);
}

extern "C" int ultr_click_cascade(const float* labels, long long n, int L, const double* exam, int n_exam,
                                  const double* click_prob, int n_click_prob, const double* uniforms,
                                  unsigned long long seed, unsigned int stream_step, float* clicks, void* stream) {
  return ultr_click_sample(ULTR_CLICK_CASCADE, labels, n, L, exam, n_exam, click_prob, n_click_prob, uniforms, seed,
                           stream_step, clicks, stream);
}

namespace ultr {

int launch_assemble_now(const AssembleArgs& a, unsigned nblocks, hipStream_t st) {
  {
    ultr::ProfScope prof_("assemble_slots_kernel", st);
    hipLaunchKernelGGL(assemble_slots_kernel, dim3(nblocks), dim3(256), 0, st, a);
  }
  return check_launch("assemble_slots_kernel");
}
}  // namespace ultr

static int assemble_checks(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B, int click_model,
                           const double* exam, int n_exam, const double* click_prob, int n_click_prob,
                           int* out_docids, float* out_clicks, int* out_list_idx, int* info) {
  ULTR_REQUIRE(initial_list && labels && exam && click_prob && out_docids && out_clicks && out_list_idx && info,
               ULTR_EINVAL, "ultr_batch_assemble: null pointer");
  ULTR_REQUIRE(Q > 0 && B > 0 && L > 0 && n_exam > 0 && n_click_prob > 0, ULTR_EINVAL,
               "ultr_batch_assemble: bad sizes Q=%d B=%d L=%d", Q, B, L);
  // click_simulation_feed.py:124-126 raises ValueError when the lists are shorter than rank_list_size
  ULTR_REQUIRE(Lmax >= L, ULTR_EINVAL,
               "Input ranklist length must be no less than the required list size, %d != %d.", Lmax, L);
  ULTR_REQUIRE(L <= ULTR_MAX_LIST, ULTR_ELIMIT, "ultr_batch_assemble: L=%d > %d", L, ULTR_MAX_LIST);
  return click_model_ok("ultr_batch_assemble", click_model, n_exam, n_click_prob);
}

static AssembleArgs assemble_args(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                                  int click_model, const double* exam, int n_exam, const double* click_prob,
                                  int n_click_prob, unsigned long long seed, unsigned int step, unsigned int* step_ctr,
                                  unsigned int slot_base, int check_validation, int max_attempts, int n_rows,
                                  int* out_docids, float* out_clicks, int* out_list_idx, int* info) {
  AssembleArgs a{};
  a.initial_list = initial_list; a.labels = labels; a.Q = Q; a.Lmax = Lmax; a.L = L; a.B = B; a.click_model = click_model;
  a.t = ClickTables{exam, n_exam, click_prob, n_click_prob};
  a.seed = seed; a.step = step; a.step_ctr = step_ctr; a.slot_base = slot_base;
  a.check_validation = check_validation; a.max_attempts = max_attempts; a.n_rows = n_rows;
  a.out_docids = out_docids; a.out_clicks = out_clicks; a.out_list_idx = out_list_idx; a.info = info;
  return a;
}

extern "C" int ultr_batch_assemble(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                                   int click_model, const double* exam, int n_exam, const double* click_prob,
                                   int n_click_prob, int mode, unsigned long long seed, unsigned int step,
                                   unsigned int* step_ctr,
                                   unsigned int slot_base, const double* uniforms, int n_cand, int slot_start,
                                   int check_validation, int max_attempts, int n_rows, int* out_docids,
                                   float* out_clicks, int* out_list_idx, int* info, void* stream) {
  int rc = assemble_checks(initial_list, labels, Q, Lmax, L, B, click_model, exam, n_exam, click_prob, n_click_prob,
                           out_docids, out_clicks, out_list_idx, info);
  if (rc) return rc;
  if (g_pending_assemble.on) {  // a parked request nobody carried runs first, on its own: launch order is call order
    g_pending_assemble.on = false;
    rc = launch_assemble_now(g_pending_assemble.a, g_pending_assemble.nblocks, as_stream(stream));
    if (rc) return rc;
  }
  ClickTables t{exam, n_exam, click_prob, n_click_prob};
  if (mode == 0) {
    ULTR_REQUIRE(max_attempts >= 1, ULTR_EINVAL, "ultr_batch_assemble: max_attempts=%d", max_attempts);
    return launch_assemble_now(assemble_args(initial_list, labels, Q, Lmax, L, B, click_model, exam, n_exam, click_prob,
                                             n_click_prob, seed, step, step_ctr, slot_base, check_validation,
                                             max_attempts, n_rows, out_docids, out_clicks, out_list_idx, info),
                               (unsigned)cdiv(B, 4), as_stream(stream));
  }
  ULTR_REQUIRE(mode == 1, ULTR_EINVAL, "ultr_batch_assemble: mode=%d", mode);
  ULTR_REQUIRE(uniforms && n_cand > 0 && slot_start >= 0 && slot_start < B, ULTR_EINVAL,
               "ultr_batch_assemble: sequential mode needs uniforms, n_cand=%d slot_start=%d", n_cand, slot_start);
  {
    ultr::ProfScope prof_("assemble_sequential_kernel", as_stream(stream));
    hipLaunchKernelGGL(assemble_sequential_kernel, dim3(1), dim3(1024), 0, as_stream(stream), initial_list, labels, Q,
                     Lmax, L, B, click_model, t, uniforms, n_cand, slot_start, check_validation, n_rows, out_docids,
                     out_clicks, out_list_idx, info);
  }
  return check_launch("assemble_sequential_kernel");
}

extern "C" int ultr_batch_assemble_deferred(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                                            int click_model, const double* exam, int n_exam, const double* click_prob,
                                            int n_click_prob, unsigned long long seed, unsigned int step,
                                            unsigned int* step_ctr, unsigned int slot_base, int check_validation,
                                            int max_attempts, int n_rows, int* out_docids, float* out_clicks,
                                            int* out_list_idx, int* info, void* stream) {
  int rc = assemble_checks(initial_list, labels, Q, Lmax, L, B, click_model, exam, n_exam, click_prob, n_click_prob,
                           out_docids, out_clicks, out_list_idx, info);
  if (rc) return rc;
  ULTR_REQUIRE(max_attempts >= 1, ULTR_EINVAL, "ultr_batch_assemble_deferred: max_attempts=%d", max_attempts);
  if (g_pending_assemble.on) {
    g_pending_assemble.on = false;
    rc = launch_assemble_now(g_pending_assemble.a, g_pending_assemble.nblocks, as_stream(stream));
    if (rc) return rc;
  }
  g_pending_assemble.a = assemble_args(initial_list, labels, Q, Lmax, L, B, click_model, exam, n_exam, click_prob,
                                       n_click_prob, seed, step, step_ctr, slot_base, check_validation, max_attempts,
                                       n_rows, out_docids, out_clicks, out_list_idx, info);
  g_pending_assemble.nblocks = (unsigned)cdiv(B, 4);
  g_pending_assemble.on = true;
  return ULTR_OK;
}

extern "C" int ultr_batch_assemble_flush(void* stream) {
  if (!g_pending_assemble.on) return ULTR_OK;
  g_pending_assemble.on = false;
  return launch_assemble_now(g_pending_assemble.a, g_pending_assemble.nblocks, as_stream(stream));
}

extern "C" int ultr_batch_assemble_cancel(void) {
  const int was = g_pending_assemble.on ? 1 : 0;
  g_pending_assemble.on = false;
  return was;
}

static int online_common(int stochastic, float tau, const double* rank_uniforms,
                         const int* docids, const float* labels, const float* scores, int B, int Lmax,
                                        int L, int n_rows, int click_model, const double* exam, int n_exam,
                                        const double* click_prob, int n_click_prob, const double* uniforms,
                                        unsigned long long seed, unsigned int step, const unsigned int* step_ctr,
                                        unsigned int slot_base, int check_validation, int max_rounds, int* out_docids,
                                        float* out_clicks, int* out_perm, void* stream) {
  ULTR_REQUIRE(docids && labels && scores && out_docids && out_clicks && exam && click_prob, ULTR_EINVAL,
               "ultr_online_rerank_click: null pointer");
  ULTR_REQUIRE(B > 0 && Lmax > 0 && L > 0, ULTR_EINVAL, "ultr_online_rerank_click: bad sizes B=%d Lmax=%d L=%d", B, Lmax, L);
  ULTR_REQUIRE(L <= Lmax, ULTR_EINVAL,
               "Input ranklist length must be no less than the required list size, %d != %d.", Lmax, L);
  ULTR_REQUIRE(L <= ULTR_MAX_LIST, ULTR_ELIMIT, "ultr_online_rerank_click: L=%d > %d", L, ULTR_MAX_LIST);
  ULTR_REQUIRE(Lmax <= ULTR_MAX_SORT, ULTR_ELIMIT, "ultr_online_rerank_click: Lmax=%d > %d", Lmax, ULTR_MAX_SORT);
  ULTR_REQUIRE(max_rounds >= 0, ULTR_EINVAL, "ultr_online_rerank_click: max_rounds=%d", max_rounds);
  ULTR_REQUIRE(!(uniforms && check_validation), ULTR_EINVAL,
               "ultr_online_rerank_click: supplied uniforms cover one round, check_validation needs the Philox stream");
  int rc = click_model_ok("ultr_online_rerank_click", click_model, n_exam, n_click_prob);
  if (rc) return rc;
  OnlineArgs a{};
  a.docids = docids; a.labels = labels; a.scores = scores; a.B = B; a.Lmax = Lmax; a.L = L; a.n_rows = n_rows;
  a.click_model = click_model; a.t = ClickTables{exam, n_exam, click_prob, n_click_prob};
  a.uniforms = uniforms; a.seed = seed; a.step = step; a.step_ctr = step_ctr; a.slot_base = slot_base;
  a.check_validation = check_validation; a.max_rounds = max_rounds;
  a.out_docids = out_docids; a.out_clicks = out_clicks; a.out_perm = out_perm;
  a.stochastic = stochastic; a.tau = tau; a.rank_uniforms = rank_uniforms;
  if (Lmax <= 16 && B >= kOnlineListsMinB && !uniforms && !rank_uniforms) {  // very many short lists: a lane per list
    const unsigned nb = (unsigned)(cdiv(B, 256) < 8192 ? cdiv(B, 256) : 8192);
    ultr::ProfScope prof_("online_rerank_click_kernel", as_stream(stream));
    launch_online_lists<16>(a, nb, as_stream(stream));
    return check_launch("online_rerank_click_lists_kernel");
  }
  if (Lmax <= 16) {  // short lists: sixteen lanes per list
    ultr::ProfScope prof_("online_rerank_click_kernel", as_stream(stream));
    hipLaunchKernelGGL(online_rerank_click_small_kernel, dim3((unsigned)cdiv(B, 16)), dim3(256), 0, as_stream(stream), a);
    return check_launch("online_rerank_click_small_kernel");
  }
  {
    ultr::ProfScope prof_("online_rerank_click_kernel", as_stream(stream));
    hipLaunchKernelGGL(online_rerank_click_kernel, dim3((unsigned)B), dim3(256),
                       (sizeof(double) + 4 * sizeof(float)) * (size_t)Lmax, as_stream(stream), a);
  }
  return check_launch("online_rerank_click_kernel");
}

extern "C" int ultr_online_rerank_click(const int* docids, const float* labels, const float* scores, int B, int Lmax,
                                        int L, int n_rows, int click_model, const double* exam, int n_exam,
                                        const double* click_prob, int n_click_prob, const double* uniforms,
                                        unsigned long long seed, unsigned int step, const unsigned int* step_ctr,
                                        unsigned int slot_base, int check_validation, int max_rounds, int* out_docids,
                                        float* out_clicks, int* out_perm, void* stream) {
  return online_common(0, 0.f, nullptr, docids, labels, scores, B, Lmax, L, n_rows, click_model, exam, n_exam, click_prob,
                       n_click_prob, uniforms, seed, step, step_ctr, slot_base, check_validation, max_rounds, out_docids,
                       out_clicks, out_perm, stream);
}

extern "C" int ultr_online_sample_click(float tau, const double* rank_uniforms, const int* docids, const float* labels,
                                        const float* scores, int B, int Lmax, int L, int n_rows, int click_model,
                                        const double* exam, int n_exam, const double* click_prob, int n_click_prob,
                                        const double* uniforms, unsigned long long seed, unsigned int step,
                                        const unsigned int* step_ctr, unsigned int slot_base, int check_validation,
                                        int max_rounds, int* out_docids, float* out_clicks, int* out_perm, void* stream) {
  return online_common(1, tau, rank_uniforms, docids, labels, scores, B, Lmax, L, n_rows, click_model, exam, n_exam,
                       click_prob, n_click_prob, uniforms, seed, step, step_ctr, slot_base, check_validation, max_rounds,
                       out_docids, out_clicks, out_perm, stream);
}
