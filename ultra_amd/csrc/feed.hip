// Click simulation and batch assembly on the device.
// Reference: ultra/utils/click_models.py:86-116 (PBM), :204-241 (cascade);
//            ultra/input_layer/click_simulation_feed.py:75-162 (get_batch).
#include "common.hpp"

namespace ultr {

struct ClickTables {
  const double* exam;
  int n_exam;
  const double* click_prob;
  int n_cp;
};

// click probability of (rank, label) in float64, click_models.py:105-116
__device__ __forceinline__ double click_threshold(const ClickTables& t, int rank, float label) {
  int rel = label > 0.f ? (int)label : 0;           // int() truncation, negatives -> 0
  rel = rel < t.n_cp ? rel : t.n_cp - 1;            // beyond the table -> last entry
  const int r = rank < t.n_exam ? rank : t.n_exam - 1;
  return t.exam[r] * t.click_prob[rel];
}

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
    return (d & 1) ? u53(blk.v[2], blk.v[3]) : u53(blk.v[0], blk.v[1]);
  }
};

// One thread per list.  labels/clicks/uniforms are list-major [n][L].
template <bool CASCADE>
__global__ void __launch_bounds__(256) click_kernel(const float* __restrict__ labels, long long n, int L,
                                                    ClickTables t, const double* __restrict__ uniforms,
                                                    unsigned long long seed, unsigned int step,
                                                    float* __restrict__ clicks) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PhiloxDraws ph;
  ph.init((uint32_t)i, 0u, step, seed);
  bool has_click = false;
  for (int r = 0; r < L; ++r) {
    const float lab = labels[i * L + r];
    const double u = uniforms ? uniforms[i * L + r] : ph.draw(r);  // a draw at EVERY rank (click_models.py:208)
    const bool c = u < click_threshold(t, r, lab);
    clicks[i * L + r] = (c && !(CASCADE && has_click)) ? 1.f : 0.f;
    has_click = has_click || c;
  }
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
  for (int r = 0; r < L; ++r) {
    const int doc = initial_list[(long long)i * Lmax + r];
    const float lab = doc < 0 ? 0.f : labels[(long long)i * Lmax + r];  // :80-82
    if (click_model == ULTR_CLICK_ORACLE) {  // oracle_mode: click_list = label_list, no click draws (:84-85)
      if (lab != 0.f) mask |= (1ull << r);
      continue;
    }
    const bool c = draw(1 + r) < click_threshold(t, r, lab);
    if (c && !(click_model == ULTR_CLICK_CASCADE && has_click)) mask |= (1ull << r);
    has_click = has_click || c;
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

// mode 0: one WAVE per slot (lane r = rank r), own Philox stream, bounded retry loop.  A thread-per-slot
// loop would chain ~2.5 attempts x L dependent gathers; here one attempt is one round trip.
__global__ void __launch_bounds__(256) assemble_slots_kernel(const int* __restrict__ initial_list,
                                                             const float* __restrict__ labels, int Q, int Lmax, int L,
                                                             int B, int click_model, ClickTables t,
                                                             unsigned long long seed, unsigned int step,
                                                             unsigned int* step_ctr,
                                                             unsigned int slot_base, int check_validation,
                                                             int max_attempts, int n_rows, int* __restrict__ out_docids,
                                                             float* __restrict__ out_clicks,
                                                             int* __restrict__ out_list_idx, int* __restrict__ info) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  // A hipGraph replays frozen kernel arguments, so the per-step Philox counter lives in device memory.  One thread
  // per block reads it and hands it to the others through LDS; the bump needs no hand-off of data, only "every block
  // has read the counter": the block arrives right after its read (the increment is made to depend on the value
  // read, so it cannot overtake the read) and the last arriver bumps -- at the START of the kernel, off the critical
  // path, with no release/acquire fence.
  if (step_ctr) {
    __shared__ unsigned int s_cur;
    if (threadIdx.x == 0) {
      const unsigned int cur = __hip_atomic_load(step_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_cur = cur;
      unsigned int* arrive = reinterpret_cast<unsigned int*>(&info[3]);
      const unsigned int t = __hip_atomic_fetch_add(arrive, 1u + (cur & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x - 1) {
        __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(step_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    step += s_cur;
  }
  if (b < B) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t slot = slot_base + (uint32_t)b;
    const unsigned long long lmask = L >= 64 ? ~0ull : ((1ull << L) - 1ull);
    // Rejection sampling is a serial chain of dependent loads per attempt (list row -> labels -> click tables),
    // and the step waits for the unluckiest slot.  Attempts are independent Philox streams (keyed by the attempt
    // index), so the wave evaluates G = 64 / P of them AT ONCE (P = lanes per attempt) and takes the first one
    // the sequential rule would have taken: same draws, same result, one round of loads in >99 % of the slots.
    const int P = L <= 16 ? 16 : L <= 32 ? 32 : 64, G = 64 / P;
    const int ai = lane / P, pos = lane % P;
    int i = 0, doc = -1, attempts = 0, gsel = 0;
    float lab = 0.f;
    unsigned long long mask = 0ull;
    for (int a0 = 0; a0 < max_attempts; a0 += G) {
      const int a = a0 + ai;
      // draw 0 (words 0,1 of block 0) picks the list: int(random.random() * length) (:132)
      const Philox4 b0 = philox4x32_10(slot, (uint32_t)a, 0u, step, k0, k1);
      i = (int)(u53(b0.v[0], b0.v[1]) * (double)Q);
      i = i < Q ? i : Q - 1;
      bool c = false;
      doc = -1;
      lab = 0.f;
      if (pos < L && a < max_attempts) {
        doc = initial_list[(long long)i * Lmax + pos];
        lab = doc < 0 ? 0.f : labels[(long long)i * Lmax + pos];  // :80-82
        if (click_model == ULTR_CLICK_ORACLE) {
          c = lab != 0.f;
        } else {
          const int d = 1 + pos;  // draws 1..L are the click draws (:87)
          const Philox4 bd = philox4x32_10(slot, (uint32_t)a, (uint32_t)(d >> 1), step, k0, k1);
          const double u = (d & 1) ? u53(bd.v[2], bd.v[3]) : u53(bd.v[0], bd.v[1]);
          c = u < click_threshold(t, pos, lab);
        }
      }
      const unsigned long long bal = __ballot(c);
      bool done = false;
      for (int g = 0; g < G && !done; ++g) {  // wave-uniform: the first attempt of this round that is accepted
        if (a0 + g >= max_attempts) break;
        unsigned long long mg = (P == 64 ? bal : (bal >> (g * P))) & lmask;
        if (click_model == ULTR_CLICK_CASCADE) mg &= (~mg + 1ull);  // only the first click counts
        if (!check_validation || mg != 0ull || a0 + g == max_attempts - 1) {
          done = true;
          gsel = g;
          mask = mg;
          attempts = a0 + g + 1;
        }
      }
      if (done) break;
    }
    if (ai == gsel && pos < L) {
      out_docids[(long long)pos * B + b] = doc < 0 ? n_rows : doc;  // pad -> the zero row (:140-143)
      float cv = ((mask >> pos) & 1ull) ? 1.f : 0.f;
      if (click_model == ULTR_CLICK_ORACLE) cv = lab;
      out_clicks[(long long)pos * B + b] = cv;
      if (pos == 0) {
        out_list_idx[b] = i;
        atomicAdd(&info[2], attempts);  // diagnostic: acceptance rate = B / total attempts
      }
    }
  }
}

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

}  // namespace ultr

using namespace ultr;

static int click_common(bool cascade, const float* labels, long long n, int L, const double* exam, int n_exam,
                        const double* click_prob, int n_cp, const double* uniforms, unsigned long long seed,
                        unsigned int step, float* clicks, void* stream) {
  ULTR_REQUIRE(labels && clicks && exam && click_prob, ULTR_EINVAL, "ultr_click: null pointer");
  ULTR_REQUIRE(n >= 0 && L > 0 && n_exam > 0 && n_cp > 0, ULTR_EINVAL, "ultr_click: bad sizes n=%lld L=%d", n, L);
  ULTR_REQUIRE(n < (1ll << 32), ULTR_ELIMIT, "ultr_click: n=%lld exceeds the 32-bit Philox list counter", n);
  if (n == 0) return ULTR_OK;
  ClickTables t{exam, n_exam, click_prob, n_cp};
  const dim3 grid((unsigned)cdiv(n, 256));
  if (cascade)
    {
      ultr::ProfScope prof_("click_kernel", as_stream(stream));
      hipLaunchKernelGGL(click_kernel<true>, grid, dim3(256), 0, as_stream(stream), labels, n, L, t, uniforms, seed, step,
                       clicks);
    }
  else
    {
      ultr::ProfScope prof_("click_kernel", as_stream(stream));
      hipLaunchKernelGGL(click_kernel<false>, grid, dim3(256), 0, as_stream(stream), labels, n, L, t, uniforms, seed,
                       step, clicks);
    }
  return check_launch("click_kernel");
}

extern "C" int ultr_click_pbm(const float* labels, long long n, int L, const double* exam, int n_exam,
                              const double* click_prob, int n_click_prob, const double* uniforms,
                              unsigned long long seed, unsigned int stream_step, float* clicks, void* stream) {
  return click_common(false, labels, n, L, exam, n_exam, click_prob, n_click_prob, uniforms, seed, stream_step, clicks,
                      stream);
}

extern "C" int ultr_click_cascade(const float* labels, long long n, int L, const double* exam, int n_exam,
                                  const double* click_prob, int n_click_prob, const double* uniforms,
                                  unsigned long long seed, unsigned int stream_step, float* clicks, void* stream) {
  return click_common(true, labels, n, L, exam, n_exam, click_prob, n_click_prob, uniforms, seed, stream_step, clicks,
                      stream);
}

extern "C" int ultr_batch_assemble(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                                   int click_model, const double* exam, int n_exam, const double* click_prob,
                                   int n_click_prob, int mode, unsigned long long seed, unsigned int step,
                                   unsigned int* step_ctr,
                                   unsigned int slot_base, const double* uniforms, int n_cand, int slot_start,
                                   int check_validation, int max_attempts, int n_rows, int* out_docids,
                                   float* out_clicks, int* out_list_idx, int* info, void* stream) {
  ULTR_REQUIRE(initial_list && labels && exam && click_prob && out_docids && out_clicks && out_list_idx && info,
               ULTR_EINVAL, "ultr_batch_assemble: null pointer");
  ULTR_REQUIRE(Q > 0 && B > 0 && L > 0 && n_exam > 0 && n_click_prob > 0, ULTR_EINVAL,
               "ultr_batch_assemble: bad sizes Q=%d B=%d L=%d", Q, B, L);
  // click_simulation_feed.py:124-126 raises ValueError when the lists are shorter than rank_list_size
  ULTR_REQUIRE(Lmax >= L, ULTR_EINVAL,
               "Input ranklist length must be no less than the required list size, %d != %d.", Lmax, L);
  ULTR_REQUIRE(L <= ULTR_MAX_LIST, ULTR_ELIMIT, "ultr_batch_assemble: L=%d > %d", L, ULTR_MAX_LIST);
  ULTR_REQUIRE(click_model == ULTR_CLICK_PBM || click_model == ULTR_CLICK_CASCADE || click_model == ULTR_CLICK_ORACLE,
               ULTR_EINVAL,
               "ultr_batch_assemble: click model %d is not on the device path", click_model);
  ClickTables t{exam, n_exam, click_prob, n_click_prob};
  if (mode == 0) {
    ULTR_REQUIRE(max_attempts >= 1, ULTR_EINVAL, "ultr_batch_assemble: max_attempts=%d", max_attempts);
    {
      ultr::ProfScope prof_("assemble_slots_kernel", as_stream(stream));
      hipLaunchKernelGGL(assemble_slots_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream),
                       initial_list, labels, Q, Lmax, L, B, click_model, t, seed, step, step_ctr, slot_base, check_validation,
                       max_attempts, n_rows, out_docids, out_clicks, out_list_idx, info);
    }
    return check_launch("assemble_slots_kernel");
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
