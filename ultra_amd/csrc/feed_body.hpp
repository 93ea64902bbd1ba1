// The batch-assembly kernel body (mode 0: one wave per slot) as a device function, shared by assemble_slots_kernel
// (feed.hip) and by the backward pass, whose last launch can carry the assembly of the NEXT batch as extra workgroups
// (ultr_batch_assemble_deferred: a prefetching feed).
// Reference: ultra/input_layer/click_simulation_feed.py:75-162 ; ultra/utils/click_models.py:86-116,204-241.
#pragma once
#include "common.hpp"

namespace ultr {

// exam / click_prob by click model:
//   PBM, cascade: exam [n_exam] per rank, click_prob [n_cp] per relevance grade (click_models.py:86-116, 204-241)
//   UBM:          exam [n_exam][n_exam], row = rank, column = distance to the last click - 1 (:125-142, 180-191)
//   TABLE:        exam [n_cp][n_exam] = P(click | grade, rank) formed on the host in float64 (the trust-bias model
//                 :304-322 and the click-matrix model :258-266 are functions of (grade, rank) only); click_prob unused
struct ClickTables {
  const double* exam;
  int n_exam;
  const double* click_prob;
  int n_cp;
};

// click probability of (rank, label) in float64, click_models.py:105-116; `last` = rank of the last click above
// this rank (-1: none), read by the user browsing model only (:170-191)
__device__ __forceinline__ double click_threshold(const ClickTables& t, int model, int rank, float label, int last) {
  int rel = label > 0.f ? (int)label : 0;           // int() truncation, negatives -> 0
  rel = rel < t.n_cp ? rel : t.n_cp - 1;            // beyond the table -> last entry
  if (model == ULTR_CLICK_TABLE) return t.exam[(long long)rel * t.n_exam + (rank < t.n_exam ? rank : t.n_exam - 1)];
  if (model == ULTR_CLICK_UBM) {
    const int n = t.n_exam, dist = rank - last;
    int row = n - 1, col;
    if (rank < n) {
      row = rank;
      col = dist - 1;
    } else if (dist > rank) {                        // no click so far: exam_prob[-1][-1]
      col = n - 1;
    } else {
      col = dist < n - 1 ? dist - 1 : n - 2;         // :188-190
    }
    return t.exam[row * n + col] * t.click_prob[rel];
  }
  const int r = rank < t.n_exam ? rank : t.n_exam - 1;
  return t.exam[r] * t.click_prob[rel];
}

struct AssembleArgs {
  const int* initial_list; const float* labels; int Q, Lmax, L, B, click_model;
  ClickTables t;
  unsigned long long seed; unsigned int step; unsigned int* step_ctr; unsigned int slot_base;
  int check_validation, max_attempts, n_rows;
  int* out_docids; float* out_clicks; int* out_list_idx; int* info;
};

// mode 0: one WAVE per slot (lane r = rank r), own Philox stream, bounded retry loop.  A thread-per-slot
// loop would chain ~2.5 attempts x L dependent gathers; here one attempt is one round trip.
__device__ __forceinline__ void assemble_slots_body(const AssembleArgs& a, int bid, int nb) {
  const int* __restrict__ initial_list = a.initial_list; const float* __restrict__ labels = a.labels;
  const int Q = a.Q, Lmax = a.Lmax, L = a.L, B = a.B, click_model = a.click_model;
  const ClickTables t = a.t;
  const unsigned long long seed = a.seed;
  unsigned int step = a.step;
  unsigned int* step_ctr = a.step_ctr;
  const unsigned int slot_base = a.slot_base;
  const int check_validation = a.check_validation, max_attempts = a.max_attempts, n_rows = a.n_rows;
  int* __restrict__ out_docids = a.out_docids; float* __restrict__ out_clicks = a.out_clicks;
  int* __restrict__ out_list_idx = a.out_list_idx; int* __restrict__ info = a.info;

  const int lane = threadIdx.x & 63;
  const int b = bid * 4 + (threadIdx.x >> 6);
  // diagnostic attempt count: the block's slots meet in LDS, and the blocks spread their sums over 16 counters --
  // one atomic per slot on ONE address serialises in L2 at ~16 ns each, 64 us for a batch of 4096
  __shared__ int s_attempts;
  if (threadIdx.x == 0) s_attempts = 0;
  __syncthreads();
  // A hipGraph replays frozen kernel arguments, so the per-step Philox counter lives in device memory.  One thread
  // per block reads it and hands it to the others through LDS; the bump needs no hand-off of data, only "every block
  // has read the counter": the block arrives right after its read has RETURNED (s_waitcnt between the two, so the
  // arrival cannot overtake the read) and the last arriver bumps -- at the START of the kernel, off the critical
  // path, with no release/acquire fence.
  if (step_ctr) {
    __shared__ unsigned int s_cur;
    if (threadIdx.x == 0) {
      const unsigned int cur = __hip_atomic_load(step_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_cur = cur;
      // two levels (16 sub-counters, then info[3]): a thousand returning atomics on one address take 16 us -- and on one
      // 128-byte LINE: sixteen counters side by side still ran the kernel 20 us at 4096 slots, so each has its own line
      unsigned int* sub = reinterpret_cast<unsigned int*>(&info[64 + 32 * (16 + (bid & 15))]);
      const unsigned int n_sub = ((unsigned)nb - (unsigned)(bid & 15) + 15u) / 16u;  // blocks that share this sub-counter
      // the arrival must not be issued before the counter value has come back: wait for the load, and keep the
      // compiler from moving either memory operation across this point
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int t = __hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == n_sub - 1) {
        __hip_atomic_store(sub, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int* arrive = reinterpret_cast<unsigned int*>(&info[3]);
        const unsigned int n_top = nb < 16 ? (unsigned)nb : 16u;
        const unsigned int t2 = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t2 == n_top - 1) {
          __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(step_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
    // PBM / cascade: this lane's examination probability and the (short) click-probability table are requested
    // before the first attempt, so that the threshold needs no memory access once the label has arrived
    const bool tables_in_regs = (click_model == ULTR_CLICK_PBM || click_model == ULTR_CLICK_CASCADE) && t.n_cp <= 8;
    double exam_r = 0.0, cpv[8];
    if (tables_in_regs) {
      exam_r = t.exam[pos < t.n_exam ? pos : t.n_exam - 1];
#pragma unroll
      for (int q = 0; q < 8; ++q) cpv[q] = t.click_prob[q < t.n_cp ? q : t.n_cp - 1];
    }
    for (int a0 = 0; a0 < max_attempts; a0 += G) {
      const int a = a0 + ai;
      // draw 0 (words 0,1 of block 0) picks the list: int(random.random() * length) (:132)
      const Philox4 b0 = philox4x32_10(slot, (uint32_t)a, 0u, step, k0, k1);
      i = (int)(u53(b0.w0, b0.w1) * (double)Q);
      i = i < Q ? i : Q - 1;
      bool c = false;
      doc = -1;
      lab = 0.f;
      double u = 2.0;
      const bool live = pos < L && a < max_attempts;
      if (live) {
        // both rows of the candidate are requested at once: a label load that waits for the doc id to turn out
        // non-negative costs the wave a second memory round trip per attempt
        doc = initial_list[(long long)i * Lmax + pos];
        const float lab_raw = labels[(long long)i * Lmax + pos];
        lab = doc < 0 ? 0.f : lab_raw;  // :80-82
        if (click_model == ULTR_CLICK_ORACLE) {
          c = lab != 0.f;
        } else {
          const int d = 1 + pos;  // draws 1..L are the click draws (:87)
          const Philox4 bd = philox4x32_10(slot, (uint32_t)a, (uint32_t)(d >> 1), step, k0, k1);
          u = (d & 1) ? u53(bd.w2, bd.w3) : u53(bd.w0, bd.w1);
          if (tables_in_regs) {
            int rel = lab > 0.f ? (int)lab : 0;     // click_threshold()'s rule: int() truncation, clamp to the table
            rel = rel < t.n_cp ? rel : t.n_cp - 1;
            double cp = cpv[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) cp = rel == q ? cpv[q] : cp;
            c = u < exam_r * cp;
          } else if (click_model != ULTR_CLICK_UBM) {
            c = u < click_threshold(t, click_model, pos, lab, -1);
          }
        }
      }
      if (click_model == ULTR_CLICK_UBM) {
        // the examination probability of a rank depends on where the last click above it was (:144-155): the ranks
        // of an attempt resolve in order, one wave-uniform round each, every lane keeping its own attempt's last click
        int last_click = -1;
        for (int r = 0; r < L; ++r) {
          const bool cr = live && pos == r && u < click_threshold(t, click_model, pos, lab, last_click);
          const unsigned long long br = __ballot(cr);
          if ((br >> (ai * P + r)) & 1ull) last_click = r;
          c = c || cr;
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
        atomicAdd(&s_attempts, attempts);  // diagnostic: acceptance rate = B / total attempts
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_attempts) atomicAdd(&info[64 + 32 * (bid & 15)], s_attempts);
}


// A request parked by ultr_batch_assemble_deferred until the next ultr_dnn_backward under the same request context carries it.
struct PendingAssemble {
  bool on;
  AssembleArgs a;
  unsigned nblocks;
};
PendingAssemble& pending_assemble();  // of the calling thread's bound request context (optim.hip)
#define g_pending_assemble (::ultr::pending_assemble())
int launch_assemble_now(const AssembleArgs& a, unsigned nblocks, hipStream_t st);  // feed.hip

}  // namespace ultr
