// Pieces of the DNN kernels shared by dnn.hip (per-launch plan) and fused.hip (fused training step):
// MFMA helpers, the weight-gradient body and the slab reduction.  See dnn.hip for the design notes.
#pragma once
#include "metrics_body.hpp"
#include "feed_body.hpp"
#include "loss_body.hpp"
#include <stdlib.h>
#include <string.h>
// No implicit FMA contraction in this file: the compiler contracts a*b+c differently in different unrolled copies
// of the same loop body, which made a row's result depend (in the last bit) on its position inside a 32-row block.
// Every fused multiply-add below is written as fmaf() explicitly, so a row's arithmetic is the same wherever it sits.
#pragma clang fp contract(off)

namespace ultr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;        // contraction chunk per LDS stage (16 MFMA steps of K=2)
constexpr int LDK = BK + 4;   // row stride of a k-contiguous LDS tile (conflict-free ds_read_b128)
constexpr int kThreads = 256; // 4 waves
using f32x4 = __attribute__((ext_vector_type(4))) float;
static inline int round4(int x) { return (x + 3) & ~3; }
// accumulator register r of lane l holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// Three-stage rotation: LOAD(chunk index, stage) must accept any index (it clamps), COMPUTE(stage, index).
#define ULTR_STREAM_GEMM(NCH, LOAD, COMPUTE)                    \
  do {                                                          \
    int i_ = 0;                                                 \
    for (; i_ + 3 <= (NCH); i_ += 3) {                          \
      LOAD(i_ + 2, S2); COMPUTE(S0, i_);                        \
      LOAD(i_ + 3, S0); COMPUTE(S1, i_ + 1);                    \
      LOAD(i_ + 4, S1); COMPUTE(S2, i_ + 2);                    \
    }                                                           \
    if (i_ < (NCH)) COMPUTE(S0, i_);                            \
    if (i_ + 1 < (NCH)) COMPUTE(S1, i_ + 1);                    \
  } while (0)
__device__ __forceinline__ float4 act4_rt(int kind, float4 z) {
  switch (kind) {
#define ULTR_ACT4(A) case A: return make_float4(act_fwd(A, z.x), act_fwd(A, z.y), act_fwd(A, z.z), act_fwd(A, z.w));
    ULTR_ACT4(ULTR_ACT_ELU)
    ULTR_ACT4(ULTR_ACT_RELU)
    ULTR_ACT4(ULTR_ACT_SELU)
    ULTR_ACT4(ULTR_ACT_TANH)
    default: return make_float4(act_fwd(ULTR_ACT_SIGMOID, z.x), act_fwd(ULTR_ACT_SIGMOID, z.y),
                                act_fwd(ULTR_ACT_SIGMOID, z.z), act_fwd(ULTR_ACT_SIGMOID, z.w));
#undef ULTR_ACT4
  }
}
// ---------------------------------------------------------------------------------------------
// wgrad:  G[k][n] = sum_m xhat[m][k] * dz[m][n]   (+ S[n] = sum_m dz[m][n])
// The contraction runs over rows, so BOTH operands are "k-strided" streams (lanes along the contiguous
// index, coalesced): no LDS, no barrier.  xhat is the normalised input the forward pass kept, which
// makes this kernel independent of doc ids and row statistics.  Workgroup = 2 x 2 waves (64 x 64 outputs,
// each stream shared by two waves through L1); rows are split over blockIdx.x into slabs -- blockIdx.x
// is the fastest-varying index so that an XCD (workgroup id mod 8) keeps to its own rows of both operands.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* xhat; int ldx;        // [M][ldx]
  const float* dz; int ldn; int N;   // [M][ldn]
  int K; long long M; int rows_per_split;
  float* gslab;            // [nsplit][K][ldn]
  float* sslab;            // [nsplit][ldn]
};

// Every layer's weight-gradient GEMM in one launch of the LDS-DMA ring kernel (fused.hip): layers[j] for j < n_layers,
// all with the same rows_per_split / nsplit; operand rows must come in whole 32-row chunks (M % 32 == 0 or zero rows
// behind M).  bf16: operands rounded to bf16 on their way to the registers.
int launch_wgrad_all(int n_layers, const WgradArgs* layers, int nsplit, bool bf16, hipStream_t st);

__device__ __forceinline__ void wgrad_body(const WgradArgs& p, int bx, int by, int bz) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int wr = w >> 1, wc = w & 1;
  const int k0 = by * 64 + wr * 32, n0 = bz * 64 + wc * 32;
  const long long m_begin = (long long)bx * p.rows_per_split;
  long long m_end = m_begin + p.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nch = (int)((m_end - m_begin + 31) >> 5);
  const int kc = min(k0 + l31, p.ldx - 1), nc = min(n0 + l31, p.ldn - 1);
  const long long m_last = p.M - 1;
  struct Frag { float a[16], b[16]; } S0, S1, S2;
  // full chunks: wave-uniform row base + one 32-bit per-lane offset (a vector add per load); the last, partial
  // chunk of a slab clamps its rows for the load and zeroes them afterwards
  const unsigned la = (unsigned)(16 * h * p.ldx + kc), lb = (unsigned)(16 * h * p.ldn + nc);
  auto load = [&](int i, Frag& S) {
    const long long mc = m_begin + 32 * (long long)min(i, nch - 1);
    if (mc + 32 <= m_end) {
      const float* ap = p.xhat + mc * p.ldx;
      const float* bp = p.dz + mc * p.ldn;
#pragma unroll
      for (int t = 0; t < 16; ++t) S.a[t] = ap[la + (unsigned)t * (unsigned)p.ldx];
#pragma unroll
      for (int t = 0; t < 16; ++t) S.b[t] = bp[lb + (unsigned)t * (unsigned)p.ldn];
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const long long m = min(mc + 16 * h + t, m_last);
        S.a[t] = p.xhat[m * p.ldx + kc];
        S.b[t] = p.dz[m * p.ldn + nc];
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;
  const bool do_colsum = by == 0 && wr == 0;
  auto compute = [&](Frag& S, int i) {
    const long long mc = m_begin + 32 * (long long)i;
    if (mc + 32 > m_end) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bool ok = mc + 16 * h + t < m_end;
        S.a[t] = ok ? S.a[t] : 0.f;
        S.b[t] = ok ? S.b[t] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(S.a[t], S.b[t], acc, 0, 0, 0);
    if (do_colsum) {
      float s4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q] = (S.b[4 * q] + S.b[4 * q + 1]) + (S.b[4 * q + 2] + S.b[4 * q + 3]);
      colsum += (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
  };
#define WG_LOAD(i, S) load(i, S)
#define WG_COMPUTE(S, i) compute(S, i)
  if (nch > 0) {
    load(0, S0);
    load(1, S1);
    ULTR_STREAM_GEMM(nch, WG_LOAD, WG_COMPUTE);
  }
  float* g = p.gslab + (long long)bx * p.K * p.ldn;
  const int col = n0 + l31;
  if (col < p.ldn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + acc_row(r, h);
      if (k < p.K) g[(long long)k * p.ldn + col] = acc[r];
    }
  }
  if (do_colsum) {
    float other;
    const float mine = xor32_partner(colsum, &other);  // the two wave halves hold alternate 16-row groups
    if (h == 0 && col < p.ldn) p.sslab[(long long)bx * p.ldn + col] = mine + other;
  }
}
// ---------------------------------------------------------------------------------------------
// finalize: reduce slabs in fixed order and form the parameter gradients
//   dW[k][n] = gamma_k G[k][n] + beta_k S[n] ; db[n] = S[n]
//   dgamma_k = sum_n W[k][n] G[k][n] ; dbeta_k = sum_n W[k][n] S[n]
// one block per (hidden layer, input row k); the last block reduces the output-layer partials
// ---------------------------------------------------------------------------------------------
// columns of the output layer's partial rows one slab-reduction workgroup takes (finalize_body); its other 256 / 16
// thread groups split the rows
constexpr int kFinOutCols = 16;
struct FinalizeArgs {
  int H;
  int row_start[ULTR_MAX_LAYERS + 1];  // block index ranges per layer
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS], nsplit[ULTR_MAX_LAYERS];
  const float* gslab[ULTR_MAX_LAYERS];
  const float* sslab[ULTR_MAX_LAYERS];
  long long off_W[ULTR_MAX_LAYERS], off_b[ULTR_MAX_LAYERS], off_gamma[ULTR_MAX_LAYERS], off_beta[ULTR_MAX_LAYERS];
  const float* params;
  float* grads;
  int has_ln;
  const float* outpart; int out_blocks; int Kout;
  float* sumsq;  // != NULL: [gridDim.x] sum of squares of what each workgroup stored into grads (norm partials)
};

__device__ __forceinline__ void finalize_body(const FinalizeArgs& p, const int blk) {
  __shared__ float red[12];
  if (blk >= p.row_start[p.H]) {  // output layer: column sums of the [out_blocks][3K+4] partial rows
    // A workgroup takes 16 columns and splits the partial rows over its 16 thread groups (64 columns x 4 groups until
    // round 3: at 4096 lists -- 1366 rows, 341 per thread -- that made these workgroups the launch's pole, 21 us against 8)
    __shared__ float s_col[kThreads];
    const int K = p.Kout, j = p.H;
    constexpr int cols = kFinOutCols;
    const int ng = (int)blockDim.x / cols;
    const int c = threadIdx.x % cols, g = threadIdx.x / cols;
    const int i = (blk - p.row_start[p.H]) * cols + c;
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < 3 * K + 1) {
      int b = g;
      for (; b + 15 * ng < p.out_blocks; b += 16 * ng) {  // 16 independent loads in flight per thread (one round trip each round)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p.outpart[(long long)(b + ng * q) * (3 * K + 4) + i];
#pragma unroll
        for (int q = 0; q < 16; ++q) a4[q & 3] += v[q];
      }
      for (; b + 3 * ng < p.out_blocks; b += 4 * ng) {  // 4 independent loads in flight per thread
#pragma unroll
        for (int q = 0; q < 4; ++q) a4[q] += p.outpart[(long long)(b + ng * q) * (3 * K + 4) + i];
      }
      for (; b < p.out_blocks; b += ng) a4[0] += p.outpart[(long long)b * (3 * K + 4) + i];
    }
    s_col[g * cols + c] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    __syncthreads();
    float q[1] = {0.f};
    if (g == 0 && i < 3 * K + 1) {
      float sum = 0.f;
      for (int gg = 0; gg < ng; ++gg) sum += s_col[gg * cols + c];
      bool stored = true;
      if (i < K) p.grads[p.off_W[j] + i] = sum;  // W_out is a plain [K] vector
      else if (i < 2 * K) { if (p.has_ln) p.grads[p.off_gamma[j] + (i - K)] = sum; else stored = false; }
      else if (i < 3 * K) { if (p.has_ln) p.grads[p.off_beta[j] + (i - 2 * K)] = sum; else stored = false; }
      else p.grads[p.off_b[j]] = sum;
      q[0] = stored ? sum * sum : 0.f;
    }
    if (p.sumsq) {
      block_sum<1>(q, red);
      if (threadIdx.x == 0) p.sumsq[blk] = q[0];
    }
    return;
  }
  int j = 0;
  while (blk >= p.row_start[j + 1]) ++j;
  const int k = blk - p.row_start[j];
  const int ldn = p.ldn[j], N = p.N[j], ns = p.nsplit[j];
  const float* W = p.params + p.off_W[j] + (long long)k * ldn;
  float* dW = p.grads + p.off_W[j] + (long long)k * ldn;
  const float gam = p.has_ln ? p.params[p.off_gamma[j] + k] : 1.f;
  const float bet = p.has_ln ? p.params[p.off_beta[j] + k] : 0.f;
  // Slab sums in a fixed order with every load of a round in flight at once (up to 8 slabs x 2 streams): a thread
  // that walks its slabs four at a time pays one L2 round trip per four slabs, and this launch sits at the end of the
  // backward pass with nothing to hide behind.  Narrow layers (N <= 128, a power of two) put the idle threads to work:
  // the block's 256 / N thread groups take alternate slabs and meet in LDS, again in fixed order.
  __shared__ float s_comb[2][kThreads];
  const float* __restrict__ gsl = p.gslab[j] + (long long)k * ldn;
  const float* __restrict__ ssl = p.sslab[j];
  const long long gstride = (long long)p.K[j] * ldn;
  const int groups = (N <= 128 && (N & (N - 1)) == 0) ? (int)blockDim.x / N : 1;
  auto slab_sums = [&](int n, int s0, int sstep, float* Gout, float* Sout) {
    float G4[4] = {0.f, 0.f, 0.f, 0.f}, S4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int base = s0; base < ns; base += 8 * sstep) {  // eight slabs a round: every plan in use has at most eight
      float g[8], sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = min(base + u * sstep, ns - 1);
        g[u] = gsl[(long long)sl * gstride + n];
        sv[u] = ssl[(long long)sl * ldn + n];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (base + u * sstep < ns) {
          G4[u & 3] += g[u];
          S4[u & 3] += sv[u];
        }
    }
    *Gout = (G4[0] + G4[1]) + (G4[2] + G4[3]);
    *Sout = (S4[0] + S4[1]) + (S4[2] + S4[3]);
  };
  float v[3] = {0.f, 0.f, 0.f};  // sum_n W G, sum_n W S, sum of squares of what this block stores
  auto emit = [&](int n, float G, float S) {
    const float dw = gam * G + bet * S;
    dW[n] = dw;
    v[2] += dw * dw;
    const float wv = W[n];
    v[0] += wv * G;
    v[1] += wv * S;
    if (k == 0) {
      p.grads[p.off_b[j] + n] = S;
      v[2] += S * S;
    }
  };
  if (groups > 1) {
    const int n = threadIdx.x & (N - 1), g = threadIdx.x / N;
    float G, S;
    slab_sums(n, g, groups, &G, &S);
    s_comb[0][threadIdx.x] = G;
    s_comb[1][threadIdx.x] = S;
    __syncthreads();
    if (g == 0) {
      for (int o = 1; o < groups; ++o) {
        G += s_comb[0][o * N + n];
        S += s_comb[1][o * N + n];
      }
      emit(n, G, S);
    }
  } else {
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
      float G, S;
      slab_sums(n, 0, 1, &G, &S);
      emit(n, G, S);
    }
  }
  if (p.has_ln || p.sumsq) {
    block_sum<3>(v, red);
    if (threadIdx.x == 0) {
      if (p.has_ln) {
        p.grads[p.off_gamma[j] + k] = v[0];
        p.grads[p.off_beta[j] + k] = v[1];
      }
      if (p.sumsq) p.sumsq[blk] = v[2] + (p.has_ln ? v[0] * v[0] + v[1] * v[1] : 0.f);
    }
  }
}
// The last launch of the backward pass can carry the assembly of the NEXT batch (ultr_batch_assemble_deferred): by
// then nothing reads the current batch any more (the metrics rode with layer 0's weight gradients), so the feed
// kernel's 10 us of Philox + rejection-sampling latency runs under the slab reduction instead of opening the next step.
// Rider workgroups first (the longer pole), padded to a multiple of 8.
struct FinalizeRiderArgs {
  FinalizeArgs f;
  AssembleArgs a;
  int a_blocks, a_pad;
};

}  // namespace ultr

// host-side argument checks (dnn.hip)
int ultr_check_desc(const UltrDnnDesc* d, const char* who);
int ultr_common_checks(const char* who, const UltrDnnDesc* desc, const float* params, const float* table,
                       const float* table_stats, int n_rows, int ldx, const int* docids, long long M, void* workspace);
