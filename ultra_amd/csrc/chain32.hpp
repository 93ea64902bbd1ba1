// The fused training chain for batches that do not fit the chip in one round: 32 rows per workgroup.
// Included by fused.hip (TcArgs, TcGeo, TC_STREAM, wave_sum_rows and the loss lane functions come from there).
//
// train_chain_kernel owns one 16-row MFMA tile per workgroup -- at lists of 10 that is ONE list and ten live rows of
// sixteen -- and every workgroup streams every weight.  That is the right shape while the batch is at most one
// workgroup per CU (256 lists): the time of the step is the time of one workgroup.  Beyond that the CUs work through
// several rounds of workgroups, the chip is throughput-bound, and both costs count: 37 % of the multiplies are padding
// and the weight stream is paid per list.  Here a workgroup owns min(4, 32 / L) whole lists (three lists of 10: 30 live
// rows of 32) and runs the same program
//   gather -> [LN -> GEMM -> bias -> ELU] x H -> LN -> score -> d_score -> output-layer backward ->
//   [dgrad GEMM -> LN backward * act'] x (H - 1)
// on v_mfma_f32_32x32x2_f32 tiles: a wave owns units of 64 output columns (two accumulators: column 2 l + t of the
// unit in accumulator t, so one 8-byte weight load per lane feeds two MFMAs), a streamed weight chunk is used by 32
// rows, and per list the multiply time and the stream are both about 0.6 of the 16-row kernel's.
// LDS holds two 32-row panels and nothing else (2 x 66 KB at the default widths), so the activations do NOT stay on
// chip for the backward pass: what LayerNorm-backward and the activation derivative need is the normalised input
// xhat_j, which the forward pass writes to memory for the weight-gradient launch anyway and the row pass reads back
// (L2-resident: written by the same CU a few microseconds earlier), and h = mean + xhat / rstd.
// Compiled for the reference's default scorer only (DNN.py:25-34: elu, layer normalisation; input width a multiple
// of 4, hidden widths multiples of 16, fp32); every other shape keeps the 16-row kernel or the per-launch plan.
#pragma once

namespace ultr {

constexpr int kC32Rows = 32;
constexpr int kC32MiscFloats = (ULTR_MAX_LAYERS + 1) * 64 + 64;  // row statistics [layer][32][2], scores [32], pad

struct C32Frag { float2 b[8]; };
struct C32Ring { C32Frag s[kTcStages]; };
typedef unsigned int c32_u32x2 __attribute__((ext_vector_type(2)));

// A GEMM input panel keeps the contraction index PERMUTED inside every group of 16: even k first, then odd k.  MFMA
// step s of a chunk takes k = 16 c + 2 s + h from lane half h, so a lane's eight values of a chunk are contiguous
// (two ds_read_b128) instead of strided by two.  Four consecutive k from k0 (k0 % 4 == 0) land as two pairs.
__device__ __forceinline__ void c32_store_panel(float* row, int k0, float a0, float a1, float a2, float a3) {
  const int base = (k0 & ~15) + ((k0 >> 1) & 7);
  *reinterpret_cast<float2*>(row + base) = make_float2(a0, a2);
  *reinterpret_cast<float2*>(row + base + 8) = make_float2(a1, a3);
}

// chunk i of the wave's stream: rows 16 c + 2 s + h (s = 0..7), two consecutive columns per lane
__device__ __forceinline__ void c32_wload(const TcGeo& g, unsigned voff, int i, C32Frag& S) {
  const unsigned vo = voff + (unsigned)(g.c_begin + i) * g.chunk_bytes;
#pragma unroll
  for (int s = 0; s < 8; ++s)
    S.b[s] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(g.rs, vo, (int)(2 * s * g.row_bytes), 0));
}
__device__ __forceinline__ unsigned c32_voff(const TcGeo& g, int u, int l31, int h) {
  const int colc = min(64 * u + 2 * l31, g.ld - 2);
  return 4u * (unsigned)(h * g.ld + colc);
}
template <bool ALWAYS = false>
__device__ __forceinline__ void c32_prefetch(const TcGeo& g, int l31, int h, C32Ring& R) {
  if (ALWAYS || g.u_first < g.U) {
    const unsigned voff = c32_voff(g, g.u_first, l31, h);
#pragma unroll
    for (int i = 0; i < kTcStages - 1; ++i) c32_wload(g, voff, i, R.s[i]);
  }
}

// out[32][N] = A[32][K] . Wm[K][ld]: A is a permuted LDS panel [32][PS]; once multiplied out it serves as the split-K
// exchange buffer.  epi(u, acc, b2): acc[t][r] = out[row (r & 3) + 8 (r >> 2) + 4 h][col 64 u + 2 l31 + t].
template <class Epi>
__device__ __forceinline__ void c32_gemm(const TcGeo& g, float* A, int PS, int l31, int h, int lane, C32Ring& R,
                                         const float* bias, Epi epi) {
  const float* arow = A + l31 * PS + 8 * h + 16 * g.c_begin;
  for (int u = g.u_first; u < g.U; u += 8) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const unsigned voff = c32_voff(g, u, l31, h);
    if (u != g.u_first) {
#pragma unroll
      for (int i = 0; i < kTcStages - 1; ++i) c32_wload(g, voff, i, R.s[i]);
    }
    float2 b2 = make_float2(0.f, 0.f);
    if (bias) b2 = *reinterpret_cast<const float2*>(bias + min(64 * u + 2 * l31, ((g.N + 1) & ~1) - 2));
    float4 a_cur0 = *reinterpret_cast<const float4*>(arow), a_cur1 = *reinterpret_cast<const float4*>(arow + 4);
    auto compute = [&](C32Frag& S, int i) {
      // the panel values of the NEXT chunk are requested before this chunk's MFMAs: their LDS latency runs under them
      const float* nx = arow + 16 * min(i + 1, g.nch - 1);
      const float4 a_nxt0 = *reinterpret_cast<const float4*>(nx), a_nxt1 = *reinterpret_cast<const float4*>(nx + 4);
      const float a[8] = {a_cur0.x, a_cur0.y, a_cur0.z, a_cur0.w, a_cur1.x, a_cur1.y, a_cur1.z, a_cur1.w};
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], S.b[s].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], S.b[s].y, acc[1], 0, 0, 0);
      }
      a_cur0 = a_nxt0;
      a_cur1 = a_nxt1;
    };
#define TC_LOAD(i, S) c32_wload(g, voff, (i), S)
#define TC_COMPUTE(S, i) compute(S, (i))
    TC_STREAM(g.nch, TC_LOAD, TC_COMPUTE);
#undef TC_LOAD
#undef TC_COMPUTE
    tc_drain_loads();
    if (g.KS > 1) {  // uniform over the workgroup (U < 8: exactly one unit per wave)
      lds_barrier();  // the panel has been multiplied out: the exchange buffer takes its place
      if (g.ks > 0) {
        float* dst = A + ((g.ks - 1) * g.Up + u) * 2048 + lane * 4;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(dst + 256 * (4 * t + q)) =
                make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
      }
      lds_barrier();
      if (g.ks == 0) {
        for (int sl = 1; sl < g.KS; ++sl) {
          const float* srcp = A + ((sl - 1) * g.Up + u) * 2048 + lane * 4;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 v = *reinterpret_cast<const float4*>(srcp + 256 * (4 * t + q));
              acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
            }
        }
      }
    }
    if (g.ks == 0) epi(u, acc, b2);
  }
  if (g.u_first >= g.U && g.KS > 1) {  // idle waves of a non-power-of-two unit count still meet the barriers
    lds_barrier();
    lds_barrier();
  }
  tc_drain_loads();
}

// TcArgs as for train_chain_kernel, with PS = row stride of the first panel (holds the widest GEMM input), PS2 = row
// stride of the second (hidden widths only); act_off / act_ps unused.
template <int RIDE, int NQ>
__global__ void __launch_bounds__(kTcThreads) train_chain32_kernel(TcArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr bool FAST = true;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int L = p.L, B = p.B, nl = p.nl, H = p.H;
  const long long list0 = p.list_base + (long long)blockIdx.x * nl;
  float* const Pa = smem + p.p1_off;
  float* const Pb = smem + p.p2_off;
  const int PSa = p.PS, PSb = p.PS2;
  float* s_stat = smem + p.misc_off;                       // [layer][32][2]
  float* s_score = s_stat + (ULTR_MAX_LAYERS + 1) * 64;    // [32]
  C32Ring R;
  TC_STAMP(0);

  // ---- the wave's four rows w, w + 8, w + 16, w + 24: local row rr = li * L + l  <->  list list0 + li, rank l.
  // Every stage walks all four in lockstep; rows beyond the workgroup's lists compute on whatever their LDS rows hold
  // (rows never mix) and store nothing.
  bool rvalid[4];
  long long re[4], rm[4];  // position-major element index (scores, clicks, doc ids), list-major row (xhat, dz)
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int rr = w + 8 * r4, li = rr / L, l = rr - li * L;
    const long long b = list0 + li;
    rvalid[r4] = li < nl && b < B;
    re[r4] = rvalid[r4] ? (long long)l * B + b : 0;
    rm[r4] = rvalid[r4] ? b * L + l : 0;
  }
  int docid[4];
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {  // (load, then select: see train_chain_kernel)
    const int dv = p.docids[re[r4]];
    docid[r4] = rvalid[r4] ? dv : p.n_rows;
  }
  // ---- the lane's element of the loss: list list0 + lane / 16, rank lane % 16 (loads now, arithmetic later)
  const int li_l = lane >> 4, r_l = lane & 15;
  const long long b_l = list0 + li_l;
  const bool live = li_l < nl && r_l < L && b_l < B;
  const long long e_l = live ? (long long)r_l * B + b_l : 0;
  float c_l = 0.f, wgt_l = 0.f, pd_tp = 1.f, pd_tm = 1.f, rem_bias = 0.f, rem_p = 1.f;
  unsigned int rem_step = 0u;
  float tower_w = 0.f, tower_b = 0.f, ipw_t = 0.f;
  if (RIDE == ULTR_RIDE_DLA) {
    c_l = live ? p.dla.clicks[e_l] : 0.f;
    tower_w = r_l < L ? p.dla.prop_params[r_l] : 0.f;
    tower_b = p.dla.prop_params[L];
  }
  if (RIDE == ULTR_RIDE_IPW) {
    c_l = live ? p.ipw.clicks[e_l] : 0.f;
    if (!p.ipw.ipw_table) wgt_l = live ? p.ipw.pw[e_l] : 0.f;
    else ipw_t = p.ipw.ipw_table[r_l < p.ipw.n_ipw ? r_l : p.ipw.n_ipw - 1];
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
  const float ob = p.bias[H][0];  // the output layer's bias
  float4 gm[NQ], btq[NQ];         // gamma / beta of the coming input stage, this lane's columns
  {
    const int K = p.K[0], nq = (((K + 15) & ~15) + 255) >> 8, k_last4 = ((K + 3) & ~3) - 4;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      btq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nq) {
        const int kc = min(4 * lane + 256 * q, k_last4);
        gm[q] = *reinterpret_cast<const float4*>(p.gamma[0] + kc);
        btq[q] = *reinterpret_cast<const float4*>(p.beta[0] + kc);
      }
    }
  }
  TcGeo g = tc_geo(p.W[0], p.K[0], p.N[0], p.ldn[0], w, false);

  // ---- input stage of layer 0: gathered rows, normalised with the table's statistics
  {
    const int K = p.K[0], K16 = (K + 15) & ~15, nq = (K16 + 255) >> 8, ldxh = (K + 3) & ~3;
    float4 x[4][NQ];
    float mean[4], rstd[4];
    bool pad[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int d = docid[r4];
      pad[r4] = d >= p.n_rows;
      const float* src = p.table + (long long)(pad[r4] ? 0 : d) * p.ldx;
      const float2 st = *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)d);
      mean[r4] = st.x;
      rstd[r4] = st.y;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nq) x[r4][q] = *reinterpret_cast<const float4*>(src + min(4 * lane + 256 * q, p.ldx - 4));
    }
    c32_prefetch<true>(g, l31, h, R);  // the first weight chunks, behind the rows in the load queue
    if (RIDE == ULTR_RIDE_DLA) wgt_l = dla_tower_regs(tower_w, tower_b, L, r_l, p.dla.prob_kind, p.dla.max_pw).pw;
    if (RIDE == ULTR_RIDE_IPW && p.ipw.ipw_table) wgt_l = live && c_l > 0.f ? ipw_t : 0.f;  // ipw_weight()
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int row = w + 8 * r4;
      const float rs = pad[r4] ? 0.f : rstd[r4], mu = pad[r4] ? 0.f : mean[r4];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nq) {
          const int k = 4 * lane + 256 * q;
          float xs[4], gg[4], bb[4], xh[4], a[4];
          tc_unpack(x[r4][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = k < K;
            xh[e] = ok ? (xs[e] - mu) * rs : 0.f;  // a zero feature row normalises to beta
            a[e] = ok ? fmaf(xh[e], gg[e], bb[e]) : 0.f;
          }
          if (k < K16) {
            c32_store_panel(Pa + row * PSa, k, a[0], a[1], a[2], a[3]);
            if (rvalid[r4] && k < ldxh)
              *reinterpret_cast<float4*>(p.xhat[0] + rm[r4] * ldxh + k) = make_float4(xh[0], xh[1], xh[2], xh[3]);
          }
        }
    }
  }

  // ================================ forward: hidden layers ================================
  float4 woq[NQ];   // the output layer's weights (loaded under the last hidden layer's GEMM)
  for (int j = 0; j < H; ++j) {
    const int N = p.N[j];
    lds_barrier();  // the input panel is complete
    {
      const int K = p.K[j + 1], nq = (((K + 15) & ~15) + 255) >> 8, k_last4 = ((K + 3) & ~3) - 4;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        btq[q] = woq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
          const int kc = min(4 * lane + 256 * q, k_last4);
          gm[q] = *reinterpret_cast<const float4*>(p.gamma[j + 1] + kc);
          btq[q] = *reinterpret_cast<const float4*>(p.beta[j + 1] + kc);
          if (j + 1 == H) woq[q] = *reinterpret_cast<const float4*>(p.W[H] + kc);
        }
      }
    }
    TC_WSTAMP(3 * j);
    c32_gemm(g, Pa, PSa, l31, h, lane, R, p.bias[j], [&](int u, f32x16 (&acc)[2], const float2& b2) {
      const int col0 = 64 * u + 2 * l31;
      if (col0 < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z0 = acc[0][r] + b2.x, z1 = acc[1][r] + b2.y;
          *reinterpret_cast<float2*>(Pb + acc_row(r, h) * PSb + col0) =
              make_float2(z0 > 0.f ? z0 : expm1_neg(z0), z1 > 0.f ? z1 : expm1_neg(z1));
        }
      }
    });
    TC_WSTAMP(3 * j + 1);
    if (j + 1 < H) {  // the next layer's weight stream starts before its rows are normalised
      g = tc_geo(p.W[j + 1], p.K[j + 1], p.N[j + 1], p.ldn[j + 1], w, false);
      c32_prefetch(g, l31, h, R);
    }
    lds_barrier();  // the layer's activations are complete
    if (j + 1 == H) break;

    // ---- input stage of layer j + 1: LayerNorm of the activations -> panel (and xhat to memory)
    {
      const int K = N, nq = (K + 255) >> 8;  // hidden widths are multiples of 16
      float4 x[4][NQ];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) x[r4][q] = *reinterpret_cast<const float4*>(Pb + (w + 8 * r4) * PSb + min(4 * lane + 256 * q, K - 4));
      float sm[4], m2[4], mean[4], rstd[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        sm[r4] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            float xs[4];
            tc_unpack(x[r4][q], xs);
#pragma unroll
            for (int e = 0; e < 4; ++e) sm[r4] += k < K ? xs[e] : 0.f;
          }
      }
      wave_sum_rows<4>(sm);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        mean[r4] = sm[r4] * (1.f / (float)K);
        m2[r4] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            float xs[4];
            tc_unpack(x[r4][q], xs);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dd = k < K ? xs[e] - mean[r4] : 0.f;
              m2[r4] = fmaf(dd, dd, m2[r4]);
            }
          }
      }
      wave_sum_rows<4>(m2);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        rstd[r4] = tc_rsqrt<FAST>(m2[r4] * (1.f / (float)K) + p.eps);
        if (lane == 0) *reinterpret_cast<float2*>(s_stat + ((j + 1) * 32 + w + 8 * r4) * 2) = make_float2(mean[r4], rstd[r4]);
      }
      const int ldxh = K;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = w + 8 * r4;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            if (k < K) {
              float xs[4], gg[4], bb[4], xh[4], a[4];
              tc_unpack(x[r4][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                xh[e] = (xs[e] - mean[r4]) * rstd[r4];
                a[e] = fmaf(xh[e], gg[e], bb[e]);
              }
              c32_store_panel(Pa + row * PSa, k, a[0], a[1], a[2], a[3]);
              if (rvalid[r4]) *reinterpret_cast<float4*>(p.xhat[j + 1] + rm[r4] * ldxh + k) = make_float4(xh[0], xh[1], xh[2], xh[3]);
            }
          }
      }
    }
    TC_WSTAMP(3 * j + 2);
  }

  // ================================ output layer, loss, output-layer backward ================================
  TC_WSTAMP(7);
  const int Ko = p.K[H], nqo = (Ko + 255) >> 8;
  float4 xo[4][NQ];
  float mean[4], rstd[4];
  {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) xo[r4][q] = *reinterpret_cast<const float4*>(Pb + (w + 8 * r4) * PSb + min(4 * lane + 256 * q, Ko - 4));
    float sm[4], m2[4], dot[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      sm[r4] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          float xs[4];
          tc_unpack(xo[r4][q], xs);
#pragma unroll
          for (int e = 0; e < 4; ++e) sm[r4] += k < Ko ? xs[e] : 0.f;
        }
    }
    wave_sum_rows<4>(sm);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      mean[r4] = sm[r4] * (1.f / (float)Ko);
      m2[r4] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          float xs[4];
          tc_unpack(xo[r4][q], xs);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dd = k < Ko ? xs[e] - mean[r4] : 0.f;
            m2[r4] = fmaf(dd, dd, m2[r4]);
          }
        }
    }
    wave_sum_rows<4>(m2);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      rstd[r4] = tc_rsqrt<FAST>(m2[r4] * (1.f / (float)Ko) + p.eps);
      dot[r4] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          float xs[4], gg[4], bb[4], ww[4];
          tc_unpack(xo[r4][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb); tc_unpack(woq[q], ww);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = k < Ko ? fmaf((xs[e] - mean[r4]) * rstd[r4], gg[e], bb[e]) : 0.f;
            dot[r4] = fmaf(a, ww[e], dot[r4]);
          }
        }
    }
    wave_sum_rows<4>(dot);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float sc = dot[r4] + ob;
      if (lane == 0) {
        s_score[w + 8 * r4] = sc;
        if (rvalid[r4]) p.scores[re[r4]] = sc;
      }
    }
  }
  TC_WSTAMP(8);
  lds_barrier();

  // ---- d_score of the lane's (list, rank), every wave for itself; the wave then picks the values of its four rows
  float go[4];
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
      lw = group_sum<16>(live ? cl * (1.f / rem_p) : 0.f);  // clicks x (1 / propensity), as every REM kernel forms it
    }
    if (!live) ds = 0.f;
    if (w == 0 && live && r_l == 0 && p.list_weights && RIDE != ULTR_RIDE_PD) p.list_weights[b_l] = lw / (float)L;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int rr = w + 8 * r4, li = rr / L, l = rr - li * L;
      const int src = li < 4 ? 16 * li + l : 0;
      const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds), src));
      go[r4] = rvalid[r4] ? v : 0.f;
    }
  }

  // the transposed weights of the first dgrad GEMM stream in under the output-layer backward stage and the partial-row
  // reduction
  if (H >= 2) {
    g = tc_geo(p.WT[H - 1], p.N[H - 1], p.K[H - 1], (p.K[H - 1] + 3) & ~3, w, false);
    c32_prefetch(g, l31, h, R);
  }
  // ---- output-layer backward: dz of the last hidden layer (-> first panel and memory) and the gradient partials
  {
    const int ldz = p.ldn[H - 1];
    float* mine = Pb + w * (3 * Ko + 1);  // [8 waves][3 Ko + 1]: the second panel is free (its rows sit in xo)
    float aW[NQ][4], aG[NQ][4], aB[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) aW[q][e] = aG[q][e] = aB[q][e] = 0.f;
    float s1[4], s2[4];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      s1[r4] = s2[r4] = 0.f;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          float xs[4], gg[4], ww[4];
          tc_unpack(xo[r4][q], xs); tc_unpack(gm[q], gg); tc_unpack(woq[q], ww);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gwg = k < Ko ? go[r4] * ww[e] * gg[e] : 0.f;
            s1[r4] += gwg;
            s2[r4] = fmaf(gwg, (xs[e] - mean[r4]) * rstd[r4], s2[r4]);
          }
        }
    }
    wave_sum_rows<4>(s1);
    wave_sum_rows<4>(s2);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      s1[r4] *= (1.f / (float)Ko);
      s2[r4] *= (1.f / (float)Ko);
      const int row = w + 8 * r4;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nqo) {
          const int k = 4 * lane + 256 * q;
          if (k < Ko) {
            float xs[4], gg[4], bb[4], ww[4], o4[4];
            tc_unpack(xo[r4][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb); tc_unpack(woq[q], ww);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gw = go[r4] * ww[e];
              const float xh = (xs[e] - mean[r4]) * rstd[r4];
              aW[q][e] += fmaf(gg[e], xh, bb[e]) * go[r4];
              aG[q][e] = fmaf(gw, xh, aG[q][e]);
              aB[q][e] += gw;
              const float dx = rstd[r4] * (gw * gg[e] - s1[r4] - xh * s2[r4]);
              o4[e] = dx * (xs[e] > 0.f ? 1.f : xs[e] + 1.f);
            }
            c32_store_panel(Pa + row * PSa, k, o4[0], o4[1], o4[2], o4[3]);
            if (rvalid[r4] && k < ldz) *reinterpret_cast<float4*>(p.dz[H - 1] + rm[r4] * ldz + k) = make_float4(o4[0], o4[1], o4[2], o4[3]);
          }
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q < nqo) {
        const int k = 4 * lane + 256 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k + e < Ko) {
            mine[k + e] = aW[q][e];
            mine[Ko + k + e] = aG[q][e];
            mine[2 * Ko + k + e] = aB[q][e];
          }
      }
    if (lane == 0) mine[3 * Ko] = (go[0] + go[1]) + (go[2] + go[3]);
  }

  TC_WSTAMP(9);
  lds_barrier();
  {
    float* out = p.outpart + (long long)(p.part_base + blockIdx.x) * (3 * Ko + 4);
    const int stride = 3 * Ko + 1;
    for (int i = tid; i < stride; i += kTcThreads) {
      float s = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) s += Pb[ww * stride + i];
      out[i] = s;
    }
  }
  lds_barrier();  // the second panel is free again: the dgrad epilogue writes it

  // ================================ backward: dgrad + row pass per hidden layer ================================
  float* cur = Pa; float* nxt = Pb;
  int PSc = PSa, PSn = PSb;
  for (int j = H - 1; j >= 1; --j) {
    // dh[32][K_j] = dz_j[32][N_j] . WT_j[N_j][K_j] ; then dz_{j-1} = LN-backward(dh) * act'(act_{j-1})
    const int K = p.K[j], nq = (K + 255) >> 8;
    const int ldo = p.ldn[j - 1];
    // what the row pass will need from memory, requested now: gamma_j and the rows' normalised inputs xhat_j
    float4 xh4[4][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (q < nq) {
        const int kc = min(4 * lane + 256 * q, K - 4);
        gm[q] = *reinterpret_cast<const float4*>(p.gamma[j] + kc);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) xh4[r4][q] = *reinterpret_cast<const float4*>(p.xhat[j] + rm[r4] * K + kc);
      }
    }
    c32_gemm(g, cur, PSc, l31, h, lane, R, nullptr, [&](int u, f32x16 (&acc)[2], const float2&) {
      const int col0 = 64 * u + 2 * l31;
      if (col0 < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          *reinterpret_cast<float2*>(nxt + acc_row(r, h) * PSn + col0) = make_float2(acc[0][r], acc[1][r]);
      }
    });
    TC_WSTAMP(j == 1 ? 12 : 10);
    if (j >= 2) {  // the next layer's weight stream starts under the row pass
      g = tc_geo(p.WT[j - 1], p.N[j - 1], p.K[j - 1], (p.K[j - 1] + 3) & ~3, w, false);
      c32_prefetch(g, l31, h, R);
    }
    lds_barrier();
    {
      float4 d4[4][NQ];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) d4[r4][q] = *reinterpret_cast<const float4*>(nxt + (w + 8 * r4) * PSn + min(4 * lane + 256 * q, K - 4));
      float mu[4], rs[4], sd[4], c1[4], c2[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float2 st = *reinterpret_cast<const float2*>(s_stat + (j * 32 + w + 8 * r4) * 2);
        mu[r4] = st.x;
        rs[r4] = st.y;
        sd[r4] = __builtin_amdgcn_rcpf(st.y);  // sqrt(var + eps): h = mean + xhat / rstd
        c1[r4] = c2[r4] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            float dd[4], gg[4], xx[4];
            tc_unpack(d4[r4][q], dd); tc_unpack(gm[q], gg); tc_unpack(xh4[r4][q], xx);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gv = k < K ? dd[e] * gg[e] : 0.f;
              c1[r4] += gv;
              c2[r4] = fmaf(gv, xx[e], c2[r4]);
            }
          }
      }
      wave_sum_rows<4>(c1);
      wave_sum_rows<4>(c2);
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        c1[r4] *= (1.f / (float)K);
        c2[r4] *= (1.f / (float)K);
        const int row = w + 8 * r4;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nq) {
            const int k = 4 * lane + 256 * q;
            if (k < K) {
              float dd[4], gg[4], xx[4], o4[4];
              tc_unpack(d4[r4][q], dd); tc_unpack(gm[q], gg); tc_unpack(xh4[r4][q], xx);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float dx = rs[r4] * (dd[e] * gg[e] - c1[r4] - xx[e] * c2[r4]);
                const float hv = fmaf(xx[e], sd[r4], mu[r4]);  // the activation this row's layer j - 1 produced
                o4[e] = dx * (hv > 0.f ? 1.f : hv + 1.f);
              }
              if (j >= 2) c32_store_panel(nxt + row * PSn, k, o4[0], o4[1], o4[2], o4[3]);
              if (rvalid[r4] && k < ldo) *reinterpret_cast<float4*>(p.dz[j - 1] + rm[r4] * ldo + k) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
          }
      }
    }
    TC_WSTAMP(j == 1 ? 13 : 11);
    if (j >= 2) lds_barrier();
    float* t = cur; cur = nxt; nxt = t;
    const int ts = PSc; PSc = PSn; PSn = ts;
  }
}

}  // namespace ultr
