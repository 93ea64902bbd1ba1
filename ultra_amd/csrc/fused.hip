// The fused training step of the DNN list scorer for short lists (L <= 16), gfx950 (MI355X).
//
// Reference op sequence (one session.run of the training graph): gather + pad row base_algorithm.py:120-132,
// DNN.build DNN.py:63-96, the list losses dla.py:119-180 / base_algorithm.py:163-251 / regression_EM.py:116-176 /
// pairwise_debias.py:139-171, tf.gradients dla.py:237-239.
//
// A row's forward pass AND the dgrad / LayerNorm-backward chain of the backward pass never need another row, and the
// d_score of every supported loss needs only the row's own list (the batch normalisers are applied after the
// all-reduce, in the optimiser launch).  So ONE workgroup owns whole lists (nl = 16 / L of them = one 16-row MFMA tile) and
// runs   gather -> [LN -> GEMM -> bias -> act] x H -> LN -> score -> d_score -> output-layer backward ->
//        [dgrad GEMM -> LN backward * act'] x (H-1)
// with every activation in LDS from first to last; what goes to memory is what the weight gradients want: xhat_j
// and dz_j of every hidden layer (rows in list-major order: the contraction over rows does not care), the scores, and
// the output layer's parameter-gradient partials.  The step is then
//   train_chain_kernel | wgrad_all_kernel (every layer's weight-gradient GEMM + the loss statistics + the in-step
//   metrics as rider workgroups) | finalize (+ next batch's assembly) | clip + Adagrad
// i.e. 4 dependent launches instead of 10.  dgrad streams TRANSPOSED copies of W_1.. (WT_j [N_j][K_j]) so that its
// weight stream is as coalesced as the forward pass's; a small launch refreshes them from the parameters.
#include "dnn_shared.hpp"

namespace ultr {

constexpr int kTcThreads = 512, kTcRows = 16, kTcExch = 7 * 1024;
constexpr int kTcMiscFloats = (ULTR_MAX_LAYERS + 1) * 32 + 32;  // row statistics [layer][16][2], scores [16], pad

#ifdef ULTR_STAMPS
__device__ long long g_tc_stamps[1024 * 32];
#define TC_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_tc_stamps[blockIdx.x * 32 + (i)] = wall_clock64();       \
  } while (0)
#else
#define TC_STAMP(i) do { } while (0)
#endif

struct TcArgs {
  const float* table; int ldx; const int* docids; int n_rows; const float* row_stats;
  int has_ln; float eps; int activation; int H;
  int B, L, nl;                                   // nl lists of L ranks per workgroup, nl * L <= 16
  const float* W[ULTR_MAX_LAYERS]; const float* WT[ULTR_MAX_LAYERS];
  const float* bias[ULTR_MAX_LAYERS]; const float* gamma[ULTR_MAX_LAYERS]; const float* beta[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  float* xhat[ULTR_MAX_LAYERS];                   // [M][round4(K_j)]  normalised input rows of layer j, list-major rows
  float* dz[ULTR_MAX_LAYERS];                     // [M][ldn_j]        gradient wrt layer j's pre-activation
  float* scores;                                  // [L][B] position-major
  float* list_weights;                            // [B] or NULL
  float* outpart;                                 // [gridDim.x][3 K_H + 4] output-layer gradient partials
  int PS;                                         // row stride of the two LDS panels (== 4 mod 32)
  int panel_floats;                               // floats per panel
  int act_off[ULTR_MAX_LAYERS], act_ps[ULTR_MAX_LAYERS];  // LDS: activations of hidden layer j, [16][act_ps]
  int p1_off, p2_off, misc_off;
  DlaArgs dla; IpwArgs ipw; RemArgs rem; PdArgs pd;       // the one matching the kernel's RIDE is live
};

struct TcFrag { float4 b[4]; };

// Geometry of one 16-row GEMM  out[16][N] = A[16][K] . Wm[K][ld]  over the 8 waves of the workgroup: a wave owns units
// of 64 output columns; fewer than 8 units split the contraction over the idle waves (KS slices, meeting in LDS).
struct TcGeo {
  const float* W; int K, N, ld, U, Up, KS, ks, u_first, c_begin, c_last, nch;
};
__device__ __forceinline__ TcGeo tc_geo(const float* W, int K, int N, int ld, int w) {
  TcGeo g;
  g.W = W; g.K = K; g.N = N; g.ld = ld;
  g.U = (N + 63) >> 6;
  int Up = 1;
  while (Up < g.U && Up < 8) Up <<= 1;
  g.Up = Up;
  g.KS = g.U >= 8 ? 1 : 8 / Up;
  g.ks = g.U >= 8 ? 0 : w / Up;
  g.u_first = g.U >= 8 ? w : w % Up;
  const int nchunks = (K + 15) >> 4, cps = (nchunks + g.KS - 1) / g.KS;
  g.c_begin = g.ks * cps;
  const int c_end = min(g.c_begin + cps, nchunks);
  g.nch = max(c_end - g.c_begin, 0);
  g.c_last = max(c_end - 1, 0);
  return g;
}
__device__ __forceinline__ void tc_wload(const TcGeo& g, int colc, int i, int kk, TcFrag& S) {
  const int k0 = 16 * min(g.c_begin + i, g.c_last) + 4 * kk;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    S.b[s] = *reinterpret_cast<const float4*>(g.W + (size_t)min(k0 + s, g.K - 1) * g.ld + colc);
}
__device__ __forceinline__ void tc_prefetch(const TcGeo& g, int nn, int kk, TcFrag& S0, TcFrag& S1) {
  if (g.u_first < g.U) {
    const int colc = min(64 * g.u_first + 4 * nn, g.ld - 4);
    tc_wload(g, colc, 0, kk, S0);
    tc_wload(g, colc, 1, kk, S1);
  }
}

// The GEMM proper.  A: LDS panel [16][PS] (zero beyond K up to the next multiple of 16); once multiplied out it serves as
// the split-K exchange buffer.  epi(u, acc) receives the finished unit: acc[t][r] = out[row 4 kk + r][col 64 u + 4 nn + t].
// The first two weight chunks of the wave's first unit must already be in S0 / S1 (tc_prefetch).
template <class Epi>
__device__ __forceinline__ void tc_gemm(const TcGeo& g, float* A, int PS, int nn, int kk, int lane, TcFrag& S0,
                                        TcFrag& S1, TcFrag& S2, Epi epi) {
  for (int u = g.u_first; u < g.U; u += 8) {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int colc = min(64 * u + 4 * nn, g.ld - 4);
    if (u != g.u_first) {
      tc_wload(g, colc, 0, kk, S0);
      tc_wload(g, colc, 1, kk, S1);
    }
    auto compute = [&](TcFrag& S, int i) {
      const float4 a4 = *reinterpret_cast<const float4*>(A + nn * PS + 16 * (g.c_begin + i) + 4 * kk);
      const float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].w, acc[3], 0, 0, 0);
      }
    };
#define TC_LOAD(i, S) tc_wload(g, colc, (i), kk, S)
#define TC_COMPUTE(S, i) compute(S, (i))
    ULTR_STREAM_GEMM(g.nch, TC_LOAD, TC_COMPUTE);
#undef TC_LOAD
#undef TC_COMPUTE
    if (g.KS > 1) {  // uniform over the workgroup (U < 8: exactly one unit per wave)
      lds_barrier();  // the panel has been multiplied out: the exchange buffer takes its place
      if (g.ks > 0) {
        float* dst = A + ((g.ks - 1) * g.Up + u) * 1024 + lane * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          *reinterpret_cast<float4*>(dst + 256 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      }
      lds_barrier();
      if (g.ks == 0) {
        for (int sl = 1; sl < g.KS; ++sl) {
          const float* srcp = A + ((sl - 1) * g.Up + u) * 1024 + lane * 4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(srcp + 256 * t);
            acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
          }
        }
      }
    }
    if (g.ks == 0) epi(u, acc);
  }
  if (g.u_first >= g.U && g.KS > 1) {  // idle waves of a non-power-of-two unit count still meet the barriers
    lds_barrier();
    lds_barrier();
  }
}

// Two rows (2w, 2w+1) of a width-K matrix, lane l holding columns 4 l + 256 q .. + 3.
struct TcRows { float4 x[2][4]; };
__device__ __forceinline__ void tc_rows_from_lds(const float* src, int stride, int K16, int nq, int w, int lane, TcRows& R) {
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nq) R.x[r2][q] = *reinterpret_cast<const float4*>(src + (2 * w + r2) * stride + min(4 * lane + 256 * q, K16 - 4));
}
// two-pass (mean, rstd) of the two rows, interleaved (biased variance, keras LayerNormalization)
__device__ __forceinline__ void tc_row_stats(const TcRows& R, int K, int nq, int lane, float eps, float (&mean)[2], float (&rstd)[2]) {
  float sm[2], m2[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    sm[r2] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nq) {
        const int k = 4 * lane + 256 * q;
        sm[r2] += (k < K ? R.x[r2][q].x : 0.f) + (k + 1 < K ? R.x[r2][q].y : 0.f) + (k + 2 < K ? R.x[r2][q].z : 0.f) +
                  (k + 3 < K ? R.x[r2][q].w : 0.f);
      }
  }
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) mean[r2] = wave_sum(sm[r2]) / (float)K;
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    m2[r2] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nq) {
        const int k = 4 * lane + 256 * q;
        const float d0 = k < K ? R.x[r2][q].x - mean[r2] : 0.f, d1 = k + 1 < K ? R.x[r2][q].y - mean[r2] : 0.f;
        const float d2 = k + 2 < K ? R.x[r2][q].z - mean[r2] : 0.f, d3 = k + 3 < K ? R.x[r2][q].w - mean[r2] : 0.f;
        m2[r2] = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, m2[r2]))));
      }
  }
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) rstd[r2] = 1.f / sqrtf(wave_sum(m2[r2]) / (float)K + eps);
}

template <int RIDE>
__global__ void __launch_bounds__(kTcThreads) train_chain_kernel(TcArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, kk = lane >> 4;
  const int L = p.L, B = p.B, nl = p.nl, H = p.H, PS = p.PS;
  const long long list0 = (long long)blockIdx.x * nl;
  float* P1 = smem + p.p1_off;
  float* P2 = smem + p.p2_off;
  float* s_stat = smem + p.misc_off;                       // [layer][16][2]
  float* s_score = s_stat + (ULTR_MAX_LAYERS + 1) * 32;    // [16]
  TcFrag S0, S1, S2;
  TC_STAMP(0);

  // ---- the wave's two rows: local row rr = li * L + l  <->  list list0 + li, rank l
  bool rvalid[2];
  long long re[2], rm[2];  // position-major element index (scores, clicks, doc ids), list-major row (xhat, dz)
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int rr = 2 * w + r2, li = rr / L, l = rr - li * L;
    const long long b = list0 + li;
    rvalid[r2] = li < nl && b < B;
    re[r2] = rvalid[r2] ? (long long)l * B + b : 0;
    rm[r2] = rvalid[r2] ? b * L + l : 0;
  }
  // ---- the lane's element of the loss: list list0 + lane / 16, rank lane % 16
  const int li_l = lane >> 4, r_l = lane & 15;
  const long long b_l = list0 + li_l;
  const bool live = li_l < nl && r_l < L && b_l < B;
  const long long e_l = live ? (long long)r_l * B + b_l : 0;
  float c_l = 0.f, wgt_l = 0.f, pd_tp = 1.f, pd_tm = 1.f, rem_bias = 0.f, rem_p = 1.f;
  unsigned int rem_step = 0u;
  if (RIDE == ULTR_RIDE_DLA) {
    c_l = live ? p.dla.clicks[e_l] : 0.f;
    wgt_l = dla_tower(p.dla.prop_params, L, r_l, p.dla.prob_kind, p.dla.max_pw).pw;
  }
  if (RIDE == ULTR_RIDE_IPW) {
    c_l = live ? p.ipw.clicks[e_l] : 0.f;
    wgt_l = live ? ipw_weight(p.ipw, e_l, r_l, c_l) : 0.f;
  }
  if (RIDE == ULTR_RIDE_PD) {
    c_l = live ? p.pd.clicks[e_l] : 0.f;
    pd_tp = r_l < L ? p.pd.t_plus[r_l] : 1.f;
    pd_tm = r_l < L ? p.pd.t_minus[r_l] : 1.f;
  }
  if (RIDE == ULTR_RIDE_REM) {
    c_l = live ? p.rem.clicks[e_l] : 0.f;
    rem_bias = p.rem.bias_ptr ? p.rem.bias_ptr[0] : 0.f;
    rem_step = p.rem.step + (p.rem.step_ctr ? p.rem.step_ctr[0] : 0u);
    rem_p = r_l < L ? p.rem.propensity[r_l] : 1.f;
  }

  // ================================ forward: hidden layers ================================
  for (int j = 0; j < H; ++j) {
    const int K = p.K[j], K16 = (K + 15) & ~15, N = p.N[j], ldn = p.ldn[j];
    const TcGeo g = tc_geo(p.W[j], K, N, ldn, w);
    tc_prefetch(g, nn, kk, S0, S1);  // the weight stream starts before the rows are ready

    // ---- input stage: rows 2w, 2w+1; lanes run along the row
    {
      const int nq = (K16 + 255) >> 8;  // <= 4 (widths <= 1024)
      const int k_last4 = ((K + 3) & ~3) - 4, ldxh = (K + 3) & ~3;
      float4 gm[4], btq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        btq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq && p.has_ln) {
          const int kc = min(4 * lane + 256 * q, k_last4);
          gm[q] = *reinterpret_cast<const float4*>(p.gamma[j] + kc);
          btq[q] = *reinterpret_cast<const float4*>(p.beta[j] + kc);
        }
      }
      TcRows R;
      float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
      bool pad[2] = {false, false};
      if (j == 0) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const int d = rvalid[r2] ? p.docids[re[r2]] : p.n_rows;  // rows beyond the workgroup's lists: the zero row
          pad[r2] = d >= p.n_rows;
          const float* src = p.table + (long long)(pad[r2] ? 0 : d) * p.ldx;
          if (p.has_ln) {
            mean[r2] = p.row_stats[2 * (long long)d];
            rstd[r2] = p.row_stats[2 * (long long)d + 1];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) R.x[r2][q] = *reinterpret_cast<const float4*>(src + min(4 * lane + 256 * q, p.ldx - 4));
        }
      } else {
        tc_rows_from_lds(smem + p.act_off[j - 1], p.act_ps[j - 1], K16, nq, w, lane, R);
        if (p.has_ln) {
          tc_row_stats(R, K, nq, lane, p.eps, mean, rstd);
          if (lane == 0) {
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
              *reinterpret_cast<float2*>(s_stat + (j * 16 + 2 * w + r2) * 2) = make_float2(mean[r2], rstd[r2]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nq) {
          const int k = 4 * lane + 256 * q;
          const float4 gq = gm[q], bt = btq[q];
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 2 * w + r2;
            const float rs = pad[r2] ? 0.f : rstd[r2], mu = pad[r2] ? 0.f : mean[r2];
            float4 xh, a;  // (x - mean) * rstd * gamma + beta ; a zero feature row normalises to beta
            xh.x = k < K ? (R.x[r2][q].x - mu) * rs : 0.f;
            xh.y = k + 1 < K ? (R.x[r2][q].y - mu) * rs : 0.f;
            xh.z = k + 2 < K ? (R.x[r2][q].z - mu) * rs : 0.f;
            xh.w = k + 3 < K ? (R.x[r2][q].w - mu) * rs : 0.f;
            a.x = k < K ? fmaf(xh.x, gq.x, bt.x) : 0.f;
            a.y = k + 1 < K ? fmaf(xh.y, gq.y, bt.y) : 0.f;
            a.z = k + 2 < K ? fmaf(xh.z, gq.z, bt.z) : 0.f;
            a.w = k + 3 < K ? fmaf(xh.w, gq.w, bt.w) : 0.f;
            if (k < K16) {
              *reinterpret_cast<float4*>(P1 + row * PS + k) = a;
              if (rvalid[r2] && k < ldxh) *reinterpret_cast<float4*>(p.xhat[j] + rm[r2] * ldxh + k) = xh;
            }
          }
        }
    }
    TC_STAMP(1 + 3 * j);
    lds_barrier();

    // ---- GEMM + epilogue: bias, activation -> the layer's activations (LDS, kept for the backward pass)
    float* actj = smem + p.act_off[j];
    const int aps = p.act_ps[j];
    tc_gemm(g, P1, PS, nn, kk, lane, S0, S1, S2, [&](int u, f32x4 (&acc)[4]) {
      const int col0 = 64 * u + 4 * nn;
      const int n16 = (N + 15) & ~15;
      const float4 b4 = *reinterpret_cast<const float4*>(p.bias[j] + min(col0, ((N + 3) & ~3) - 4));
      if (col0 < n16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 z = make_float4(acc[0][r] + b4.x, acc[1][r] + b4.y, acc[2][r] + b4.z, acc[3][r] + b4.w);
          float4 v = act4_rt(p.activation, z);
          v.x = col0 < N ? v.x : 0.f;
          v.y = col0 + 1 < N ? v.y : 0.f;
          v.z = col0 + 2 < N ? v.z : 0.f;
          v.w = col0 + 3 < N ? v.w : 0.f;
          *reinterpret_cast<float4*>(actj + (4 * kk + r) * aps + col0) = v;
        }
      }
    });
    TC_STAMP(2 + 3 * j);
    lds_barrier();
    TC_STAMP(3 + 3 * j);
  }

  // ================================ output layer, loss, output-layer backward ================================
  const int Ko = p.K[H], Ko16 = (Ko + 15) & ~15, nqo = (Ko16 + 255) >> 8;
  TcRows R;
  float4 gm[4], btq[4], woq[4];
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
  {
    const int k_last4 = ((Ko + 3) & ~3) - 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      btq[q] = woq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nqo) {
        const int kc = min(4 * lane + 256 * q, k_last4);
        if (p.has_ln) {
          gm[q] = *reinterpret_cast<const float4*>(p.gamma[H] + kc);
          btq[q] = *reinterpret_cast<const float4*>(p.beta[H] + kc);
        }
        woq[q] = *reinterpret_cast<const float4*>(p.W[H] + kc);
      }
    }
    tc_rows_from_lds(smem + p.act_off[H - 1], p.act_ps[H - 1], Ko16, nqo, w, lane, R);
    if (p.has_ln) tc_row_stats(R, Ko, nqo, lane, p.eps, mean, rstd);
    float dot[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nqo) {
        const int k = 4 * lane + 256 * q;
        const float xs[2][4] = {{R.x[0][q].x, R.x[0][q].y, R.x[0][q].z, R.x[0][q].w},
                                {R.x[1][q].x, R.x[1][q].y, R.x[1][q].z, R.x[1][q].w}};
        const float gg[4] = {gm[q].x, gm[q].y, gm[q].z, gm[q].w}, bb[4] = {btq[q].x, btq[q].y, btq[q].z, btq[q].w};
        const float ww[4] = {woq[q].x, woq[q].y, woq[q].z, woq[q].w};
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = k + e < Ko ? fmaf((xs[r2][e] - mean[r2]) * rstd[r2], gg[e], bb[e]) : 0.f;
            dot[r2] = fmaf(a, ww[e], dot[r2]);
          }
      }
    const float ob = p.bias[H][0];
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const float sc = wave_sum(dot[r2]) + ob;
      if (lane == 0) {
        s_score[2 * w + r2] = sc;
        if (rvalid[r2]) p.scores[re[r2]] = sc;
      }
    }
  }
  TC_STAMP(16);
  lds_barrier();

  // ---- d_score of the lane's (list, rank), every wave for itself; the wave then picks the values of its two rows
  float go[2];
  {
    const float o = live ? s_score[li_l * L + r_l] : 0.f;
    const float cl = fminf(fmaxf(c_l, 0.f), 1.f);
    float ds = 0.f, lw = 0.f;
    if (RIDE == ULTR_RIDE_DLA) {
      ds = dla_lane(o, c_l, wgt_l, live, p.dla.loss_func).ds;
      lw = group_sum<16>(live ? wgt_l * cl : 0.f);
    }
    if (RIDE == ULTR_RIDE_IPW) {
      ds = dla_lane(o, c_l, wgt_l, live, p.ipw.loss_func).ds;
      lw = group_sum<16>(live ? wgt_l * cl : 0.f);
    }
    if (RIDE == ULTR_RIDE_PD) {
      float unused[16];
      ds = pd_lane<false>(o, c_l, pd_tp, pd_tm, live, r_l, L, lane & 48, unused);
    }
    if (RIDE == ULTR_RIDE_REM) {
      if (live) ds = rem_elem_v(p.rem, rem_bias, rem_p, rem_step, b_l, r_l, e_l, o, c_l).d;
      lw = group_sum<16>(live ? cl / rem_p : 0.f);
    }
    if (!live) ds = 0.f;
    if (w == 0 && live && r_l == 0 && p.list_weights && RIDE != ULTR_RIDE_PD) p.list_weights[b_l] = lw / (float)L;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int rr = 2 * w + r2, li = rr / L, l = rr - li * L;
      const int src = li < 4 ? 16 * li + l : 0;
      const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds), src));
      go[r2] = rvalid[r2] ? v : 0.f;
    }
  }

  // ---- output-layer backward: dz of the last hidden layer (-> P1 and memory) and the parameter-gradient partials
  {
    const int ldz = p.ldn[H - 1];
    float* s_out = P2;  // [8 waves][3 Ko + 1]
    float db = go[0] + go[1];
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (p.has_ln) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          const float gg[4] = {gm[q].x, gm[q].y, gm[q].z, gm[q].w}, ww[4] = {woq[q].x, woq[q].y, woq[q].z, woq[q].w};
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const float xs[4] = {R.x[r2][q].x, R.x[r2][q].y, R.x[r2][q].z, R.x[r2][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gw = k + e < Ko ? go[r2] * ww[e] : 0.f;
              const float xh = (xs[e] - mean[r2]) * rstd[r2];
              s1[r2] += gw * gg[e];
              s2[r2] += gw * gg[e] * xh;
            }
          }
        }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        s1[r2] = wave_sum(s1[r2]) / (float)Ko;
        s2[r2] = wave_sum(s2[r2]) / (float)Ko;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < nqo) {
        const int k = 4 * lane + 256 * q;
        const float gg[4] = {gm[q].x, gm[q].y, gm[q].z, gm[q].w}, bb[4] = {btq[q].x, btq[q].y, btq[q].z, btq[q].w};
        const float ww[4] = {woq[q].x, woq[q].y, woq[q].z, woq[q].w};
        float aW[4] = {0.f, 0.f, 0.f, 0.f}, aG[4] = {0.f, 0.f, 0.f, 0.f}, aB[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const float xs[4] = {R.x[r2][q].x, R.x[r2][q].y, R.x[r2][q].z, R.x[r2][q].w};
          float o4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = k + e < Ko;
            const float gw = ok ? go[r2] * ww[e] : 0.f;
            float dx = gw;
            if (p.has_ln) {
              const float xh = (xs[e] - mean[r2]) * rstd[r2];
              aW[e] += ok ? (gg[e] * xh + bb[e]) * go[r2] : 0.f;
              aG[e] += gw * xh;
              aB[e] += gw;
              dx = rstd[r2] * (gw * gg[e] - s1[r2] - xh * s2[r2]);
            } else {
              aW[e] += ok ? xs[e] * go[r2] : 0.f;
            }
            o4[e] = ok ? dx * act_grad_from_out(p.activation, xs[e]) : 0.f;
          }
          const float4 ov = make_float4(o4[0], o4[1], o4[2], o4[3]);
          if (k < Ko16) {
            *reinterpret_cast<float4*>(P1 + (2 * w + r2) * PS + k) = ov;
            if (rvalid[r2] && k < ldz) *reinterpret_cast<float4*>(p.dz[H - 1] + rm[r2] * ldz + k) = ov;
          }
        }
        float* mine = s_out + w * (3 * Ko + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < Ko) {
            mine[k + e] = aW[e];
            mine[Ko + k + e] = aG[e];
            mine[2 * Ko + k + e] = aB[e];
          }
      }
    if (lane == 0) s_out[w * (3 * Ko + 1) + 3 * Ko] = db;
  }

  // the transposed weights of the first dgrad GEMM stream in under the partial-row reduction
  int cur_is_p1 = 1;
  TcGeo gd = tc_geo(nullptr, 16, 16, 16, w);
  if (H >= 2) {
    gd = tc_geo(p.WT[H - 1], p.N[H - 1], p.K[H - 1], (p.K[H - 1] + 3) & ~3, w);
    tc_prefetch(gd, nn, kk, S0, S1);
  }
  TC_STAMP(17);
  lds_barrier();
  {
    float* out = p.outpart + (long long)blockIdx.x * (3 * Ko + 4);
    const int stride = 3 * Ko + 1;
    for (int i = tid; i < stride; i += kTcThreads) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) s += P2[ww * stride + i];
      out[i] = s;
    }
  }
  lds_barrier();  // P2 is free again: the dgrad epilogue writes it
  TC_STAMP(18);

  // ================================ backward: dgrad + row pass per hidden layer ================================
  for (int j = H - 1; j >= 1; --j) {
    // dh[16][K_j] = dz_j[16][N_j] . WT_j[N_j][K_j] ; then dz_{j-1} = LN-backward(dh) * act'(act_{j-1})
    float* cur = cur_is_p1 ? P1 : P2;
    float* nxt = cur_is_p1 ? P2 : P1;
    const int K = p.K[j], K16 = (K + 15) & ~15, nq = (K16 + 255) >> 8;
    const int ldo = p.ldn[j - 1];
    // what the row pass will need from memory, requested now
    float4 gq4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gq4[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (q < nq && p.has_ln) gq4[q] = *reinterpret_cast<const float4*>(p.gamma[j] + min(4 * lane + 256 * q, ((K + 3) & ~3) - 4));
    }
    tc_gemm(gd, cur, PS, nn, kk, lane, S0, S1, S2, [&](int u, f32x4 (&acc)[4]) {
      const int col0 = 64 * u + 4 * nn;
      if (col0 < K16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(nxt + (4 * kk + r) * PS + col0) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
      }
    });
    if (j >= 2) {  // the next layer's weight stream starts under the row pass
      gd = tc_geo(p.WT[j - 1], p.N[j - 1], p.K[j - 1], (p.K[j - 1] + 3) & ~3, w);
      tc_prefetch(gd, nn, kk, S0, S1);
    }
    TC_STAMP(19 + 2 * (H - 1 - j));
    lds_barrier();
    {
      TcRows D, A;
      tc_rows_from_lds(nxt, PS, K16, nq, w, lane, D);
      tc_rows_from_lds(smem + p.act_off[j - 1], p.act_ps[j - 1], K16, nq, w, lane, A);
      float mu[2] = {0.f, 0.f}, rs[2] = {1.f, 1.f}, c1[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
      if (p.has_ln) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const float2 st = *reinterpret_cast<const float2*>(s_stat + (j * 16 + 2 * w + r2) * 2);
          mu[r2] = st.x;
          rs[r2] = st.y;
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              const float dd[4] = {D.x[r2][q].x, D.x[r2][q].y, D.x[r2][q].z, D.x[r2][q].w};
              const float gg[4] = {gq4[q].x, gq4[q].y, gq4[q].z, gq4[q].w};
              const float hh[4] = {A.x[r2][q].x, A.x[r2][q].y, A.x[r2][q].z, A.x[r2][q].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float gv = k + e < K ? dd[e] * gg[e] : 0.f;
                c1[r2] += gv;
                c2[r2] += gv * (hh[e] - mu[r2]) * rs[r2];
              }
            }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          c1[r2] = wave_sum(c1[r2]) / (float)K;
          c2[r2] = wave_sum(c2[r2]) / (float)K;
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int row = 2 * w + r2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            const float dd[4] = {D.x[r2][q].x, D.x[r2][q].y, D.x[r2][q].z, D.x[r2][q].w};
            const float gg[4] = {gq4[q].x, gq4[q].y, gq4[q].z, gq4[q].w};
            const float hh[4] = {A.x[r2][q].x, A.x[r2][q].y, A.x[r2][q].z, A.x[r2][q].w};
            float o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float dx = dd[e];
              if (p.has_ln) dx = rs[r2] * (dx * gg[e] - c1[r2] - (hh[e] - mu[r2]) * rs[r2] * c2[r2]);
              o4[e] = k + e < K ? dx * act_grad_from_out(p.activation, hh[e]) : 0.f;
            }
            const float4 ov = make_float4(o4[0], o4[1], o4[2], o4[3]);
            if (k < K16) {
              if (j >= 2) *reinterpret_cast<float4*>(nxt + row * PS + k) = ov;
              if (rvalid[r2] && k < ldo) *reinterpret_cast<float4*>(p.dz[j - 1] + rm[r2] * ldo + k) = ov;
            }
          }
      }
    }
    TC_STAMP(20 + 2 * (H - 1 - j));
    if (j >= 2) lds_barrier();
    cur_is_p1 ^= 1;
  }
  TC_STAMP(31);
}

// WT_j[n][k] = W_j[k][n] for the hidden layers j >= 1 (32 x 32 tiles through LDS), one launch for all of them
struct TransposeArgs {
  int n; const float* W[ULTR_MAX_LAYERS]; float* WT[ULTR_MAX_LAYERS]; int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  int start[ULTR_MAX_LAYERS + 1];
};
__global__ void __launch_bounds__(256) transpose_weights_kernel(TransposeArgs p) {
  __shared__ float t[32][33];
  int j = 0;
  while (j + 1 < p.n && (int)blockIdx.x >= p.start[j + 1]) ++j;
  const int id = blockIdx.x - p.start[j];
  const int K = p.K[j], N = p.N[j], ldn = p.ldn[j], ldk = (K + 3) & ~3;
  const int tn = (N + 31) >> 5;
  const int k0 = (id / tn) * 32, n0 = (id % tn) * 32;
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
  for (int r = r0; r < 32; r += 8) t[r][c] = (k0 + r < K && n0 + c < N) ? p.W[j][(long long)(k0 + r) * ldn + n0 + c] : 0.f;
  __syncthreads();
  for (int r = r0; r < 32; r += 8)
    if (n0 + r < N && k0 + c < ldk) p.WT[j][(long long)(n0 + r) * ldk + k0 + c] = t[c][r];
}

// Every layer's weight-gradient GEMM in one launch, with the loss statistics and the in-step metrics as rider
// workgroups (they come first: the longer poles).
struct WgradAllArgs {
  int n_layers, nsplit;
  WgradArgs g[ULTR_MAX_LAYERS];
  int start[ULTR_MAX_LAYERS + 1], gy[ULTR_MAX_LAYERS];
  MetricsArgs m; int m_blocks, m_pad;
  int ride; DlaArgs dla; IpwArgs ipw; RemArgs rem; PdArgs pd; int l_blocks, l_pad;
};
__global__ void __launch_bounds__(kThreads, 2) wgrad_all_kernel(WgradAllArgs p) {
  int id = blockIdx.x;
  if (id < p.m_pad) {
    if (id >= p.m_blocks) return;
    __shared__ float s_vals[4 * kMaxOut];
    metrics_short_lists(p.m, id, p.m_blocks, s_vals);
    __syncthreads();
    metrics_finish(p.m, id, p.m_blocks, s_vals, kMetThreads / 64);
    return;
  }
  id -= p.m_pad;
  if (id < p.l_pad) {
    if (id >= p.l_blocks) return;
    __shared__ float s_acc[kMaxStat];
    __shared__ float s_part[4 * kMaxStat];
    __shared__ float s_tower[4 * 16];
    if (p.ride == ULTR_RIDE_DLA) dla_loss_small_body(p.dla, id, p.l_blocks, s_acc, s_part, s_tower);
    if (p.ride == ULTR_RIDE_IPW) ipw_loss_small_body(p.ipw, id, p.l_blocks, s_acc, s_part);
    if (p.ride == ULTR_RIDE_REM) rem_loss_small_body(p.rem, id, p.l_blocks, s_acc, s_part);
    if (p.ride == ULTR_RIDE_PD) pd_loss_small_body(p.pd, id, p.l_blocks, s_acc, s_part, s_tower);
    return;
  }
  id -= p.l_pad;
  int j = 0;
  while (j + 1 < p.n_layers && id >= p.start[j + 1]) ++j;
  const int local = id - p.start[j];
  const int bx = local % p.nsplit, t = local / p.nsplit;
  wgrad_body(p.g[j], bx, t % p.gy[j], t / p.gy[j]);
}

__global__ void __launch_bounds__(kThreads) finalize_fused_kernel(FinalizeRiderArgs p) {
  const int id = blockIdx.x;
  if (id >= p.a_pad) {
    finalize_body(p.f, id - p.a_pad);
    return;
  }
  if (id < p.a_blocks) assemble_slots_body(p.a, id, p.a_blocks);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct FusedPlan {
  int H, nl, nwg;
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  int PS, panel_floats, act_off[ULTR_MAX_LAYERS], act_ps[ULTR_MAX_LAYERS], p1_off, p2_off, misc_off;
  size_t smem_bytes;
  // workspace (floats)
  long long xhat[ULTR_MAX_LAYERS], dz[ULTR_MAX_LAYERS], wt[ULTR_MAX_LAYERS], gslab[ULTR_MAX_LAYERS], sslab[ULTR_MAX_LAYERS], outpart;
  int nsplit, rows_per_split;
  long long total;
  bool ok;
};

static FusedPlan make_fused_plan(const UltrDnnDesc* d, int B, int L) {
  FusedPlan P{};
  P.ok = false;
  P.H = d->n_layers - 1;
  if (P.H < 1 || L < 1 || L > 16 || B < 1) return P;
  P.nl = 16 / L;
  if (P.nl > 4) P.nl = 4;  // one 16-lane group per list in the loss lanes of a wave
  P.nwg = (int)cdiv(B, P.nl);
  const long long M = (long long)B * L;
  int cur = d->in_features, maxw = d->in_features;
  for (int j = 0; j <= P.H; ++j) {
    P.K[j] = cur;
    P.N[j] = d->sizes[j];
    P.ldn[j] = round4(d->sizes[j]);
    cur = d->sizes[j];
    if (j < P.H && P.N[j] > maxw) maxw = P.N[j];
  }
  if (maxw > 1024) return P;
  P.PS = ((maxw + 31) & ~31) + 4;
  P.panel_floats = kTcRows * P.PS > kTcExch ? kTcRows * P.PS : kTcExch;
  if (8 * (3 * P.K[P.H] + 1) > P.panel_floats) return P;
  int off = 0;
  for (int j = 0; j < P.H; ++j) {
    P.act_ps[j] = ((P.N[j] + 31) & ~31) + 4;
    P.act_off[j] = off;
    off += kTcRows * P.act_ps[j];
  }
  P.p1_off = off; off += P.panel_floats;
  P.p2_off = off; off += P.panel_floats;
  P.misc_off = off; off += kTcMiscFloats;
  P.smem_bytes = sizeof(float) * (size_t)off;
  if (P.smem_bytes > 160 * 1024) return P;
  long long o = 0;
  auto take = [&](long long n) {
    const long long at = o;
    o += (n + 63) & ~63ll;
    return at;
  };
  long long tiles = 0;
  for (int j = 0; j < P.H; ++j) tiles += cdiv(P.K[j], 64) * cdiv(P.N[j], 64);
  long long ns = 512 / tiles;
  if (ns < 1) ns = 1;
  if (ns > cdiv(M, 64)) ns = cdiv(M, 64);
  P.rows_per_split = (int)(cdiv(cdiv(M, ns), 32) * 32);
  P.nsplit = (int)cdiv(M, P.rows_per_split);
  for (int j = 0; j < P.H; ++j) {
    P.xhat[j] = take(M * round4(P.K[j]));
    P.dz[j] = take(M * P.ldn[j]);
    P.wt[j] = j >= 1 ? take((long long)P.N[j] * round4(P.K[j])) : 0;
    P.gslab[j] = take((long long)P.nsplit * P.K[j] * P.ldn[j]);
    P.sslab[j] = take((long long)P.nsplit * P.ldn[j]);
  }
  P.outpart = take((long long)P.nwg * (3 * P.K[P.H] + 4));
  P.total = o;
  P.ok = true;
  return P;
}

}  // namespace ultr

using namespace ultr;

#ifdef ULTR_STAMPS
extern "C" int ultr_debug_tc_stamps(long long* host, int n) {
  if (n > 1024 * 32) n = 1024 * 32;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tc_stamps), sizeof(long long) * (size_t)n);
  return n;
}
#endif

extern "C" long long ultr_dnn_fused_workspace_bytes(const UltrDnnDesc* desc, int B, int L) {
  if (ultr_check_desc(desc, "ultr_dnn_fused_workspace_bytes") != ULTR_OK) return -1;
  const FusedPlan P = make_fused_plan(desc, B, L);
  return P.ok ? P.total * (long long)sizeof(float) : 0;
}

template <int RIDE>
static int launch_chain(const TcArgs& a, int nwg, size_t smem, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(train_chain_kernel<RIDE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess) {
      set_error("ultr_dnn_train_fused: cannot raise the dynamic LDS limit");
      return ULTR_EHIP;
    }
    attr_done = true;
  }
  long long pw = 0;
  for (int j = 0; j < a.H; ++j) pw += (long long)a.K[j] * a.N[j] + (j >= 1 ? (long long)a.K[j] * a.N[j] : 0);
  {
    ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("train_chain_kernel/P%lld", pw) : "train_chain_kernel", st);
    hipLaunchKernelGGL(train_chain_kernel<RIDE>, dim3((unsigned)nwg), dim3(kTcThreads), smem, st, a);
  }
  return check_launch("train_chain_kernel");
}

extern "C" int ultr_dnn_train_fused(const UltrDnnDesc* desc, const float* params, const float* table,
                                    const float* table_stats, int n_rows, int ldx, const int* docids, int B, int L,
                                    void* workspace, float* scores, float* grads, float* sumsq, void* stream) {
  const long long M = (long long)B * L;
  int rc = ultr_common_checks("ultr_dnn_train_fused", desc, params, table, table_stats, n_rows, ldx, docids, M, workspace);
  if (rc) return rc;
  ULTR_REQUIRE(scores && grads && ((uintptr_t)grads & 15) == 0, ULTR_EINVAL, "ultr_dnn_train_fused: bad scores/grads");
  ULTR_REQUIRE((desc->n_params & 3) == 0, ULTR_EINVAL, "ultr_dnn_train_fused: n_params=%lld must be a multiple of 4", desc->n_params);
  const FusedPlan P = make_fused_plan(desc, B, L);
  ULTR_REQUIRE(P.ok, ULTR_ELIMIT, "ultr_dnn_train_fused: shape not supported (needs >= 1 hidden layer, lists of at most 16, "
               "widths that fit LDS): ask ultr_dnn_fused_workspace_bytes first");
  // the parked loss request describes what the step differentiates
  const PendingLoss pl = g_pending_loss;
  ULTR_REQUIRE(pl.kind != ULTR_RIDE_NONE, ULTR_EINVAL,
               "ultr_dnn_train_fused: no loss request parked (call ultr_loss_*_deferred with the same scores first)");
  const float* pl_scores = pl.kind == ULTR_RIDE_DLA ? pl.dla.scores : pl.kind == ULTR_RIDE_IPW ? pl.ipw.scores
                           : pl.kind == ULTR_RIDE_REM ? pl.rem.scores : pl.pd.scores;
  ULTR_REQUIRE(pl_scores == scores && pl.rows() == M, ULTR_EINVAL,
               "ultr_dnn_train_fused: the parked loss request is for another scores buffer or batch shape");
  g_pending_loss.kind = ULTR_RIDE_NONE;
  float* ws = reinterpret_cast<float*>(workspace);
  const int has_ln = desc->norm == ULTR_NORM_LAYER;
  hipStream_t st = as_stream(stream);
  const int H = P.H;

  // ---- transposed weights for the dgrad GEMMs
  if (H >= 2) {
    TransposeArgs t{};
    t.n = H - 1;
    int start = 0;
    for (int j = 1; j < H; ++j) {
      t.W[j - 1] = params + desc->off_W[j]; t.WT[j - 1] = ws + P.wt[j];
      t.K[j - 1] = P.K[j]; t.N[j - 1] = P.N[j]; t.ldn[j - 1] = P.ldn[j];
      t.start[j - 1] = start;
      start += (int)(cdiv(P.K[j], 32) * cdiv(P.N[j], 32));
    }
    t.start[H - 1] = start;
    {
      ultr::ProfScope prof_("transpose_weights_kernel", st);
      hipLaunchKernelGGL(transpose_weights_kernel, dim3((unsigned)start), dim3(256), 0, st, t);
    }
    rc = check_launch("transpose_weights_kernel");
    if (rc) return rc;
  }

  // ---- forward + loss gradient + dgrad chain
  TcArgs a{};
  a.table = table; a.ldx = ldx; a.docids = docids; a.n_rows = n_rows; a.row_stats = table_stats;
  a.has_ln = has_ln; a.eps = desc->ln_eps; a.activation = desc->activation; a.H = H;
  a.B = B; a.L = L; a.nl = P.nl;
  for (int j = 0; j <= H; ++j) {
    a.W[j] = params + desc->off_W[j]; a.bias[j] = params + desc->off_b[j];
    a.gamma[j] = params + desc->off_gamma[j]; a.beta[j] = params + desc->off_beta[j];
    a.K[j] = P.K[j]; a.N[j] = P.N[j]; a.ldn[j] = P.ldn[j];
    if (j < H) {
      a.xhat[j] = ws + P.xhat[j]; a.dz[j] = ws + P.dz[j];
      a.WT[j] = j >= 1 ? ws + P.wt[j] : nullptr;
      a.act_off[j] = P.act_off[j]; a.act_ps[j] = P.act_ps[j];
    }
  }
  a.scores = scores;
  a.outpart = ws + P.outpart;
  a.PS = P.PS; a.panel_floats = P.panel_floats; a.p1_off = P.p1_off; a.p2_off = P.p2_off; a.misc_off = P.misc_off;
  a.dla = pl.dla; a.ipw = pl.ipw; a.rem = pl.rem; a.pd = pl.pd;
  a.list_weights = pl.kind == ULTR_RIDE_DLA ? pl.dla.list_weights : pl.kind == ULTR_RIDE_IPW ? pl.ipw.list_weights
                   : pl.kind == ULTR_RIDE_REM ? pl.rem.list_weights : nullptr;
  if (pl.kind == ULTR_RIDE_DLA) rc = launch_chain<ULTR_RIDE_DLA>(a, P.nwg, P.smem_bytes, st);
  else if (pl.kind == ULTR_RIDE_IPW) rc = launch_chain<ULTR_RIDE_IPW>(a, P.nwg, P.smem_bytes, st);
  else if (pl.kind == ULTR_RIDE_REM) rc = launch_chain<ULTR_RIDE_REM>(a, P.nwg, P.smem_bytes, st);
  else rc = launch_chain<ULTR_RIDE_PD>(a, P.nwg, P.smem_bytes, st);
  if (rc) return rc;

  // ---- every weight-gradient GEMM + loss statistics + in-step metrics
  {
    WgradAllArgs g{};
    g.n_layers = H; g.nsplit = P.nsplit;
    int start = 0;
    for (int j = 0; j < H; ++j) {
      WgradArgs& q = g.g[j];
      q.xhat = ws + P.xhat[j]; q.ldx = round4(P.K[j]);
      q.dz = ws + P.dz[j]; q.ldn = P.ldn[j]; q.N = P.N[j];
      q.K = P.K[j]; q.M = M; q.rows_per_split = P.rows_per_split;
      q.gslab = ws + P.gslab[j]; q.sslab = ws + P.sslab[j];
      g.start[j] = start; g.gy[j] = (int)cdiv(P.K[j], 64);
      start += P.nsplit * g.gy[j] * (int)cdiv(P.ldn[j], 64);
    }
    g.start[H] = start;
    if (g_pending_metrics.on && g_pending_metrics.nblocks > 64) {  // too big to ride along: a launch of its own
      g_pending_metrics.on = false;
      rc = launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, st);
      if (rc) return rc;
    }
    if (g_pending_metrics.on) {
      g.m = g_pending_metrics.a; g.m_blocks = (int)g_pending_metrics.nblocks; g.m_pad = (g.m_blocks + 7) & ~7;
      g_pending_metrics.on = false;
    }
    g.ride = pl.kind; g.dla = pl.dla; g.ipw = pl.ipw; g.rem = pl.rem; g.pd = pl.pd;
    // the chain kernel has written the list weights: the riders must not race with the metrics that read them
    g.dla.list_weights = nullptr; g.ipw.list_weights = nullptr; g.rem.list_weights = nullptr;
    g.l_blocks = (int)pl.nblocks; g.l_pad = (g.l_blocks + 7) & ~7;
    {
      ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("wgrad_all_kernel/P%lld", [&] {
        long long s = 0; for (int j = 0; j < H; ++j) s += (long long)P.K[j] * P.N[j]; return s; }()) : "wgrad_all_kernel", st);
      hipLaunchKernelGGL(wgrad_all_kernel, dim3((unsigned)(start + g.m_pad + g.l_pad)), dim3(kThreads), 0, st, g);
    }
    rc = check_launch("wgrad_all_kernel");
    if (rc) return rc;
  }

  // ---- slab reduction -> flat gradient (+ the next batch's assembly)
  FinalizeRiderArgs fr{};
  FinalizeArgs& f = fr.f;
  f.H = H;
  int start = 0;
  for (int j = 0; j < H; ++j) {
    f.row_start[j] = start;
    start += P.K[j];
    f.K[j] = P.K[j]; f.N[j] = P.N[j]; f.ldn[j] = P.ldn[j]; f.nsplit[j] = P.nsplit;
    f.gslab[j] = ws + P.gslab[j]; f.sslab[j] = ws + P.sslab[j];
  }
  f.row_start[H] = start;
  for (int j = 0; j <= H; ++j) {
    f.off_W[j] = desc->off_W[j]; f.off_b[j] = desc->off_b[j];
    f.off_gamma[j] = desc->off_gamma[j]; f.off_beta[j] = desc->off_beta[j];
  }
  f.params = params; f.grads = grads; f.has_ln = has_ln;
  f.outpart = ws + P.outpart; f.out_blocks = P.nwg; f.Kout = P.K[H];
  f.sumsq = sumsq;
  const unsigned fblocks = (unsigned)(start + (3 * P.K[H] + 1 + 63) / 64);
  if (g_pending_assemble.on && g_pending_assemble.nblocks <= 4096) {
    fr.a = g_pending_assemble.a;
    fr.a_blocks = (int)g_pending_assemble.nblocks; fr.a_pad = (fr.a_blocks + 7) & ~7;
    g_pending_assemble.on = false;
  }
  {
    ultr::ProfScope prof_("finalize_fused_kernel", st);
    hipLaunchKernelGGL(finalize_fused_kernel, dim3(fblocks + (unsigned)fr.a_pad), dim3(kThreads), 0, st, fr);
  }
  return check_launch("finalize_fused_kernel");
}
