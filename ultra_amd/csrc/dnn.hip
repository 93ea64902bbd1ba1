// DNN list scorer: forward and backward as LDS-tiled fp32-MFMA GEMMs with fused
// LayerNorm / bias / activation, for gfx950 (MI355X).
//
// Reference: ultra/ranking_model/DNN.py:63-96 (LN(input) -> matmul -> bias_add -> activation per
// layer; rows are the position-major concat of the list, :65), feature gather + zero pad row
// ultra/learning_algorithm/base_algorithm.py:120-132, gradients = tf.gradients (dla.py:237-239).
//
// Design notes:
//  * exact fp32: v_mfma_f32_32x32x2_f32 is bit-for-bit an fmaf chain, so the parity bound is the
//    fp32 one (1e-5); one wave owns one 32x32 accumulator tile (16 VGPRs).
//  * LayerNorm of layer j's input needs full-row statistics, but a GEMM tile only sees 32 columns
//    of a row.  The producing layer's epilogue writes per-(row, 32-column group) partial (sum,
//    M2); the consuming kernel merges them with Chan's formula (two-pass quality, fixed order,
//    no atomics).  Layer 0 normalises raw feature rows, whose statistics depend on the data
//    only: they are precomputed once per table (ultr_ln_row_stats).
//  * the feature gather (K1) is fused into the layer-0 A-tile load, forward and backward.
//  * backward per layer: wgrad G = xhat^T dz (split over rows into slabs, reduced in fixed order
//    by the finalize kernel; dW = gamma*G + beta x S, dgamma = sum_n W*G, dbeta = sum_n W*S, so
//    layer 0 needs no dgrad GEMM at all), dgrad dh = dz W^T, then a row pass applies the
//    LayerNorm backward and the activation derivative (from the stored activation output).
#include "dnn_shared.hpp"
#include <stdlib.h>
#include <string.h>

namespace ultr {


// Diagnostic build only (-DULTR_STAMPS, tools/kbench.hip): in-kernel phase stamps on the 100 MHz wall clock.
#ifdef ULTR_STAMPS
__device__ long long g_stamps[16 * 1024 * 8];
#define ULTR_STAMP(slot, i)                                                                              \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      const int b_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;                     \
      if (b_ < 1024) g_stamps[(((slot) * 1024) + b_) * 8 + (i)] = wall_clock64();                        \
    }                                                                                                    \
  } while (0)
#else
#define ULTR_STAMP(slot, i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// workspace layout (all offsets in floats)
// ---------------------------------------------------------------------------------------------
struct Layout {
  int H;                               // hidden layers
  int K[ULTR_MAX_LAYERS];              // input width of layer j
  int N[ULTR_MAX_LAYERS];              // output width of layer j
  int ldn[ULTR_MAX_LAYERS];            // padded output width (multiple of 4) = row stride of W_j, act_j, dz
  long long act[ULTR_MAX_LAYERS];      // [M][ldn_j]        activations of hidden layer j
  long long stat[ULTR_MAX_LAYERS + 1]; // [M][2]            (mean, rstd) of layer j's input, j >= 1
  long long dz[3];                     // [M][maxld]        rotating gradient buffers
  long long xhat[ULTR_MAX_LAYERS];     // [M][round4(K_j)]  normalised input rows of layer j (for wgrad)
  int nsplit[ULTR_MAX_LAYERS];
  int rows_per_split[ULTR_MAX_LAYERS];
  long long gslab[ULTR_MAX_LAYERS];    // [nsplit][K_j][ldn_j]
  long long sslab[ULTR_MAX_LAYERS];    // [nsplit][ldn_j]
  int out_rows_per_block, out_blocks;
  long long outpart;                   // [out_blocks][3*Kout + 4]
  int maxld;
  // batch normalisation only (bn_body.hpp): column statistics per layer, the scratch of the column sums, the
  // normalised-and-scaled input `a` of the layer being run, xhat of the output layer's input
  // large batches (wgall): every layer's weight gradient in ONE launch of the LDS-DMA ring kernel (fused.hip), after
  // the dgrad chain -- so each layer's dz keeps a buffer of its own, and all layers share one row split
  bool wgall;
  long long dzl[ULTR_MAX_LAYERS];      // [M][ldn_j]
  bool bn;
  long long cstat[ULTR_MAX_LAYERS + 1];  // [K_j][2] (mean, rstd)
  long long c12, bn_part, bn_a, xhat_out;
  long long total;
};


static Layout make_layout(const UltrDnnDesc* d, long long M) {
  Layout L;
  L.H = d->n_layers - 1;
  long long off = 0;
  auto take = [&](long long n) {
    long long o = off;
    off += (n + 63) & ~63ll;
    return o;
  };
  int cur = d->in_features;
  L.maxld = 4;
  for (int j = 0; j < d->n_layers; ++j) {
    L.K[j] = cur;
    L.N[j] = d->sizes[j];
    L.ldn[j] = round4(d->sizes[j]);
    cur = d->sizes[j];
  }
  for (int j = 0; j < L.H; ++j) {
    if (L.ldn[j] > L.maxld) L.maxld = L.ldn[j];
    L.act[j] = take(M * L.ldn[j]);
  }
  for (int j = 0; j <= L.H; ++j) L.stat[j] = take(M * 2);
  for (int i = 0; i < 3; ++i) L.dz[i] = take(M * L.maxld);
  for (int j = 0; j < L.H; ++j) {
    L.xhat[j] = take(M * round4(L.K[j]));
  }
  // Measured at M = 40960 (MSLR shape): the three register-streaming wgrad launches 140 + 106 + 42 us (54-77 TFLOP/s),
  // wgrad_all_kernel 199 us (96 TFLOP/s).  Needs whole 32-row chunks (no row masks in its ring).
  L.wgall = M >= 16384 && (M & 31) == 0 && L.H >= 1 && d->norm != ULTR_NORM_BATCH && d->norm != ULTR_NORM_BATCH_EVAL;
  long long all_tiles = 0;
  for (int j = 0; j < L.H; ++j) all_tiles += cdiv(L.K[j], 64) * cdiv(L.N[j], 64);
  for (int j = 0; j < L.H; ++j) {
    if (L.wgall) {  // two workgroups per CU over all layers together, one row split for all
      long long ns = 512 / all_tiles > 1 ? 512 / all_tiles : 1;
      if (ns > cdiv(M, 64)) ns = cdiv(M, 64);
      const long long rps = cdiv(cdiv(M, ns), 32) * 32;
      L.rows_per_split[j] = (int)rps;
      L.nsplit[j] = (int)cdiv(M, rps);
      L.gslab[j] = take((long long)L.nsplit[j] * L.K[j] * L.ldn[j]);
      L.sslab[j] = take((long long)L.nsplit[j] * L.ldn[j]);
      L.dzl[j] = take(M * L.ldn[j]);
      continue;
    }
    const long long tiles = cdiv(L.K[j], 64) * cdiv(L.N[j], 64);
    // Whole rounds of workgroups: at most one per CU (256) -- 24 tiles x 11 slabs = 264 workgroups would run a
    // second round for 8 of them and take twice as long -- or, when every slab still spans 512 rows, three per CU
    // (768): the GEMM streams its operands straight into registers, and three resident workgroups per CU hide
    // that latency (K136 N512 at M = 40960: 178 us with 264 workgroups, 139 with 504, 106 with 768, 119 with 1008).
    long long want = 256 / tiles > 1 ? 256 / tiles : 1;
    {
      const long long w3 = 768 / tiles > 1 ? 768 / tiles : 1;
      if (M / w3 >= 512) want = w3;
    }
    const long long max_split = cdiv(M, 64);
    if (want > max_split) want = max_split;
    if (want < 1) want = 1;
    long long rps = cdiv(cdiv(M, want), BK) * BK;
    L.rows_per_split[j] = (int)rps;
    L.nsplit[j] = (int)cdiv(M, rps);
    L.gslab[j] = take((long long)L.nsplit[j] * L.K[j] * L.ldn[j]);
    L.sslab[j] = take((long long)L.nsplit[j] * L.ldn[j]);
  }
  long long rpb = cdiv(cdiv(M, 256), 16) * 16;  // 16 rows = one batch of four rows for each of the four waves
  if (rpb < 16) rpb = 16;
  L.out_rows_per_block = (int)rpb;
  L.out_blocks = (int)cdiv(M, rpb);
  L.outpart = take((long long)L.out_blocks * (3 * L.K[L.H] + 4));
  L.bn = d->norm == ULTR_NORM_BATCH || d->norm == ULTR_NORM_BATCH_EVAL;
  L.c12 = L.bn_part = L.bn_a = L.xhat_out = 0;
  if (L.bn) {
    int maxk = 4;
    for (int j = 0; j <= L.H; ++j) {
      L.cstat[j] = take(2ll * L.K[j]);
      if (round4(L.K[j]) > maxk) maxk = round4(L.K[j]);
    }
    L.c12 = take(2ll * maxk);
    L.bn_part = take(2ll * 64 * maxk);  // kBnSplits row splits
    L.bn_a = take(M * maxk);
    L.xhat_out = take(M * round4(L.K[L.H]));
  }
  L.total = off;
  return L;
}

// ---------------------------------------------------------------------------------------------
// per-row LayerNorm statistics of a dense table (layer 0 input), one wave per row, two-pass
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) ln_row_stats_kernel(const float* __restrict__ x, long long n_rows, int F,
                                                                int ldx, float eps, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r > n_rows) return;
  if (r == n_rows) {  // the zero pad row (base_algorithm.py:120-123): mean 0, var 0
    if (lane == 0) {
      stats[2 * r] = 0.f;
      stats[2 * r + 1] = 1.f / sqrtf(eps);
    }
    return;
  }
  const float* row = x + r * ldx;
  float s = 0.f;
  for (int k = lane; k < F; k += 64) s += row[k];
  const float mean = wave_sum(s) / (float)F;
  float m2 = 0.f;
  for (int k = lane; k < F; k += 64) {
    const float d = row[k] - mean;
    m2 += d * d;
  }
  const float var = wave_sum(m2) / (float)F;  // biased variance (keras LayerNormalization)
  if (lane == 0) {
    stats[2 * r] = mean;
    stats[2 * r + 1] = 1.f / sqrtf(var + eps);
  }
}

// ---------------------------------------------------------------------------------------------
// MFMA micro-kernels.  v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; over the 16 steps of a BK=32 chunk, lane half h walks k = 16h + t.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_frag_kcontig(const float* __restrict__ tile, int row, int h, float (&f)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(tile + row * LDK + 16 * h + 4 * q);
    f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
  }
}
__device__ __forceinline__ void load_frag_kmajor(const float* __restrict__ tile, int ld, int col, int h,
                                                 float (&f)[16]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) f[t] = tile[(16 * h + t) * ld + col];
}
__device__ __forceinline__ void mma16(const float (&a)[16], const float (&b)[16], f32x16& acc) {
#pragma unroll
  for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
}
// Software pipeline shared by the three GEMM kernels: two register stages keep TWO chunks of global
// loads in flight while a third is being multiplied out of LDS (one workgroup per CU has no other
// waves to hide the L2 latency behind).  LOAD(chunk, regs), STORE(regs, buf), COMPUTE(buf).
#define ULTR_GEMM_PIPELINE(NCHUNK, LOAD, STORE, COMPUTE)  \
  do {                                                    \
    if ((NCHUNK) > 0) LOAD(0, R0);                        \
    if ((NCHUNK) > 1) LOAD(1, R1);                        \
    if ((NCHUNK) > 0) STORE(R0, 0);                       \
    lds_barrier();                                         \
    for (int c_ = 0; c_ < (NCHUNK); c_ += 2) {            \
      if (c_ + 2 < (NCHUNK)) LOAD(c_ + 2, R0);            \
      COMPUTE(0);                                         \
      if (c_ + 1 < (NCHUNK)) STORE(R1, 1);                \
      lds_barrier();                                       \
      if (c_ + 1 >= (NCHUNK)) break;                      \
      if (c_ + 3 < (NCHUNK)) LOAD(c_ + 3, R1);            \
      COMPUTE(1);                                         \
      if (c_ + 2 < (NCHUNK)) STORE(R0, 0);                \
      lds_barrier();                                       \
    }                                                     \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Streaming-B wave GEMM.  At the headline size (M = 2560 rows) a layer is a few hundred 32x32 tiles:
// far too few to hide latency behind occupancy, so the kernels are built around the length of the
// dependent chain instead.  What the probes under tools/ showed on MI355X:
//   * a wait on ANY global load costs a full round trip for everything issued before it (the memory
//     counter is in-order); L2 answers in ~130 ns, but a "lane owns a row" float4 load visits 32 cache
//     lines and keeps the CU's L1 busy for as many cycles, so a row-per-lane operand stream saturates L1;
//   * a workgroup-wide __syncthreads() drains every outstanding load (it fences global memory too).
// Hence: the A panel (32 rows x the whole contraction, LayerNorm already applied) is brought into LDS
// ONCE per workgroup with fully coalesced loads (a wave reads whole rows, which also yields the row
// statistics with two in-wave reductions), and the weights stream from L2 straight into MFMA operand
// registers as coalesced column reads, three 32-deep chunks ahead of the multiply, with no barrier in
// the main loop (lds_barrier() orders LDS only).  The contraction is split over the waves of a
// workgroup (WK) when a layer has fewer than ~1024 tiles; the partial accumulators meet in LDS.
//   "k-strided" operand: lane (j, h) owns column j and reads 16 rows k = kk .. kk+15 (coalesced over j).
// Addresses are clamped into the operand, values beyond the contraction length are zero in the panel.
// ---------------------------------------------------------------------------------------------
// full chunk: element index of row k = kk0 + 16h + t is (kk0 + t) * ld + (16h * ld + col): a wave-uniform
// term (scalar registers) plus one per-lane 32-bit offset, i.e. one vector add per load.
__device__ __forceinline__ void ld_kstrided(const float* __restrict__ W, int ld, int kk0, unsigned lane_off,
                                            float (&f)[16]) {
  const unsigned base = (unsigned)kk0 * (unsigned)ld + lane_off;
#pragma unroll
  for (int t = 0; t < 16; ++t) f[t] = W[base + (unsigned)t * (unsigned)ld];
}
// The same stream through buffer loads: descriptor and row offsets in scalar registers, ONE vector add per chunk.  The
// fp32 MFMA runs on the vector ALU, so the sixteen 64-bit address computations a chunk of flat loads costs came straight
// out of the multiply time (measured: 108 vector instructions per 32 MFMAs in the large-batch forward kernel).  The
// descriptor ends at row K: rows beyond the matrix read as zero, so the tail chunk needs no clamps.
constexpr int kRawBufFlags = 0x00020000;  // raw buffer, 32-bit data format (word 3 of a gfx9 buffer descriptor)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t weight_rsrc(const float* W, int K, int ld) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((unsigned)K * (unsigned)ld * 4u), kRawBufFlags);
}
__device__ __forceinline__ void ld_kstrided_buf(__amdgpu_buffer_rsrc_t rs, int ld, int kk0, unsigned lane_off, float (&f)[16]) {
  const unsigned vo = 4u * ((unsigned)kk0 * (unsigned)ld + lane_off);
#pragma unroll
  for (int t = 0; t < 16; ++t)
    f[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, (int)(4u * (unsigned)t * (unsigned)ld), 0));
}
// tail chunk (kk0 + 32 > K): row indices clamped to the last valid row (the panel holds zeros there)
__device__ __forceinline__ void ld_kstrided_tail(const float* __restrict__ W, int ld, int kk0, int h, int colc, int k_last,
                                                 float (&f)[16]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) f[t] = W[(size_t)min(kk0 + 16 * h + t, k_last) * ld + colc];
}
__device__ __forceinline__ void zero_tail(float (&f)[16], int kk, int K) {
#pragma unroll
  for (int t = 0; t < 16; ++t) f[t] = (kk + t < K) ? f[t] : 0.f;
}

template <int ACT>
__device__ __forceinline__ void act16(const f32x16& acc, float bias, bool ok, float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = ok ? act_fwd(ACT, acc[r] + bias) : 0.f;
}
__device__ __forceinline__ void act16_rt(int kind, const f32x16& acc, float bias, bool ok, float (&v)[16]) {
  switch (kind) {
    case ULTR_ACT_ELU: act16<ULTR_ACT_ELU>(acc, bias, ok, v); break;
    case ULTR_ACT_RELU: act16<ULTR_ACT_RELU>(acc, bias, ok, v); break;
    case ULTR_ACT_SELU: act16<ULTR_ACT_SELU>(acc, bias, ok, v); break;
    case ULTR_ACT_TANH: act16<ULTR_ACT_TANH>(acc, bias, ok, v); break;
    default: act16<ULTR_ACT_SIGMOID>(acc, bias, ok, v); break;
  }
}

// ---------------------------------------------------------------------------------------------
// forward hidden layer:  act[m][n] = f( LN(x_m) . W[:,n] + b[n] )
// workgroup = WN x WK waves: 32 rows x (32 WN) columns, contraction split WK ways
// ---------------------------------------------------------------------------------------------
constexpr int kActNone = -1;
constexpr int kPanelMaxK = 1024;  // contraction length held in LDS at a time (hidden widths fit: ULTR_MAX_WIDTH)

struct FwdArgs {
  // input rows: either gathered from the table (layer 0) or dense
  const float* x;          // table [n_rows][ldx]  or  act_{j-1} [M][ldx]
  int ldx;
  const int* docids;       // != NULL: gather rows x[docids[m]] ; docid == n_rows -> zero row
  int n_rows;
  const float* row_stats;  // gather: table_stats [(n_rows+1)][2] ; dense: computed here
  float* stat_out;         // dense + LN: (mean, rstd) of the input rows [M][2], written by the first column tile
  int has_ln;
  float eps;
  const float* gamma; const float* beta; const float* W; const float* bias;
  int K, N, ldn;
  long long M;
  int activation;          // ULTR_ACT_* or kActNone (plain GEMM)
  float* act_out;          // [M][ldn]
  float* xhat_out;         // != NULL: normalised input rows (x - mean) * rstd [M][ldxh], kept for wgrad
  int ldxh;
  // fused output layer (FUSE: the workgroup's WN column tiles cover complete rows, N <= 32 WN):
  //   score[m] = LN(act_m) . out_W + out_b ; out_stat[m] = (mean, rstd) of act_m
  const float* out_gamma; const float* out_beta; const float* out_W; const float* out_bias;
  float* out_stat; float* scores;
};

static inline int panel_stride(int K) { return ((K < kPanelMaxK ? K : kPanelMaxK) + 31) / 32 * 32 + 4; }
template <int WN, int WK, int RT = 1>
static inline size_t fwd_smem_floats(int K, bool fuse) {
  const size_t panel = 32 * RT * (size_t)panel_stride(K), exch = (size_t)(WK > 1 ? WK : 0) * WN * 1024;
  return (panel > exch ? panel : exch) + (fuse ? (size_t)WN * 32 * 36 + 32 * 3 * WN : 0);
}

// RT = 2 (large batches, WK == 1, no fused output layer): the workgroup stages 64 rows and every wave keeps two
// accumulator tiles, so that a streamed weight value feeds two MFMAs -- at 32 rows per workgroup the weight stream
// (one float per lane per MFMA) is what the L1/L2 path cannot deliver at more than about half the MFMA rate.
template <int WN, int WK, bool FUSE, bool GATHER, int QMAX, int RT = 1>
__global__ void __launch_bounds__(64 * WN * WK, RT == 1 ? 2 : 1) fwd_layer_kernel(FwdArgs p) {
  static_assert(RT == 1 || (WK == 1 && !FUSE), "two row tiles: plain layers without a contraction split only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = WN * WK;
  constexpr int ROWS = 32 * RT;
  const int PS = ((min(p.K, kPanelMaxK) + 31) & ~31) + 4;  // panel row stride: == 4 (mod 32), conflict-free b128 rows
  float* s_panel = smem;                             // [ROWS][PS] normalised input rows
  float* s_acc = smem;                               // [WK][WN][16][64] split-K exchange, reuses the panel (WK > 1)
  float* s_epi = smem + max(ROWS * PS, (WK > 1 ? WK : 0) * WN * 1024);       // FUSE: [WN][32][36] transpose tiles, then [32][WN][3]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int wk = w % WK, wn = w / WK;
  const long long m0 = (long long)blockIdx.x * ROWS;
  const int n0 = (blockIdx.y * WN + wn) * 32;
  ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 0);

  const int col = n0 + l31;
  const bool col_ok = col < p.N;
  const int colc = min(col, p.ldn - 1);
  const unsigned lane_off = (unsigned)(16 * h * p.ldn + colc);
  const __amdgpu_buffer_rsrc_t w_rs = weight_rsrc(p.W, p.K, p.ldn);
  struct Frag { float b[16]; } S0, S1, S2;
  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  const float bias = p.bias ? p.bias[min(col, p.N - 1)] : 0.f;

  for (int kp0 = 0; kp0 < p.K; kp0 += kPanelMaxK) {  // one pass unless the input is wider than the panel (layer 0 only)
    const int Kc = min(p.K - kp0, kPanelMaxK);
    // this wave's share of the pass, in chunks of 32
    const int nchunk_all = (Kc + 31) >> 5, cps = (nchunk_all + WK - 1) / WK;
    const int c_begin = wk * cps;
    const int c_end = min(c_begin + cps, nchunk_all);
    const int nch = max(c_end - c_begin, 0);
    const int c_last = max(c_end - 1, 0);
#define FWD_LOAD(i, S)                                                                        \
  do {                                                                                       \
    const int kk0_ = kp0 + 32 * min(c_begin + (i), c_last);                                  \
    ld_kstrided_buf(w_rs, p.ldn, kk0_, lane_off, S.b);                                       \
  } while (0)
    FWD_LOAD(0, S0);  // the weight stream starts before the panel is staged
    FWD_LOAD(1, S1);

    // ---- stage the panel: wave w takes rows w, w + NW, ...; lanes run along k (float4 each, coalesced)
    if (kp0 > 0) lds_barrier();  // the previous pass has been multiplied out
    constexpr int RPW = (ROWS + NW - 1) / NW;          // rows per wave
    // QMAX (template): float4 per lane and row the staging registers are sized for, 256 * QMAX >= min(K, panel)
    const int nq = (Kc + 255) >> 8;
    float4 gq[QMAX], bq[QMAX];                         // gamma / beta of this lane's columns, in flight with the rows
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
      gq[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      bq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nq && p.has_ln) {
        const int kc = min(kp0 + 4 * lane + 256 * q, ((p.K + 3) & ~3) - 4);
        gq[q] = *reinterpret_cast<const float4*>(p.gamma + kc);
        bq[q] = *reinterpret_cast<const float4*>(p.beta + kc);
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; rr += 4) {              // four rows at a time: 4 x nq independent loads in flight
      float4 xv[4][QMAX];
      float mean[4], rstd[4];
      bool pad[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = w + (rr + u) * NW;
        const long long m = min(m0 + row, p.M - 1);
        const float* src;
        mean[u] = 0.f; rstd[u] = 1.f; pad[u] = false;
        if (GATHER) {
          const int d = p.docids[m];                   // wave-uniform
          pad[u] = d >= p.n_rows;
          src = p.x + (long long)(pad[u] ? 0 : d) * p.ldx + kp0;
          if (p.has_ln) {
            mean[u] = p.row_stats[2 * (long long)d];
            rstd[u] = p.row_stats[2 * (long long)d + 1];
          }
        } else {
          src = p.x + m * p.ldx;
        }
        const int x_last4 = p.ldx - kp0 - 4;
#pragma unroll
        for (int q = 0; q < QMAX; ++q)
          if (q < nq) xv[u][q] = *reinterpret_cast<const float4*>(src + min(4 * lane + 256 * q, x_last4));
      }
      if (!GATHER && p.has_ln) {                       // two-pass row statistics, four rows interleaved
        float sm[4], m2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          sm[u] = 0.f;
#pragma unroll
          for (int q = 0; q < QMAX; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              sm[u] += (k < Kc ? xv[u][q].x : 0.f) + (k + 1 < Kc ? xv[u][q].y : 0.f) + (k + 2 < Kc ? xv[u][q].z : 0.f) +
                       (k + 3 < Kc ? xv[u][q].w : 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) mean[u] = wave_sum(sm[u]) / (float)Kc;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          m2[u] = 0.f;
#pragma unroll
          for (int q = 0; q < QMAX; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              const float d0 = k < Kc ? xv[u][q].x - mean[u] : 0.f, d1 = k + 1 < Kc ? xv[u][q].y - mean[u] : 0.f;
              const float d2 = k + 2 < Kc ? xv[u][q].z - mean[u] : 0.f, d3 = k + 3 < Kc ? xv[u][q].w - mean[u] : 0.f;
              m2[u] = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, m2[u]))));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          rstd[u] = 1.f / sqrtf(wave_sum(m2[u]) / (float)Kc + p.eps);  // biased variance (keras LayerNormalization)
          const int row = w + (rr + u) * NW;
          if (blockIdx.y == 0 && lane == 0 && row < ROWS && m0 + row < p.M)
            *reinterpret_cast<float2*>(p.stat_out + 2 * (m0 + row)) = make_float2(mean[u], rstd[u]);
        }
      }
#pragma unroll
      for (int q = 0; q < QMAX; ++q)
        if (q < nq) {
          const int k = 4 * lane + 256 * q;
          const float4 g = gq[q], bt = bq[q];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int row = w + (rr + u) * NW;
            if (row < ROWS && k < PS - 4) {
              float4 a, xh;  // (x - mean) * rstd * gamma + beta ; a zero feature row normalises to beta, 0 without LayerNorm
              const float rs = pad[u] ? 0.f : rstd[u], mu = pad[u] ? 0.f : mean[u];
              xh.x = k < Kc ? (xv[u][q].x - mu) * rs : 0.f;
              xh.y = k + 1 < Kc ? (xv[u][q].y - mu) * rs : 0.f;
              xh.z = k + 2 < Kc ? (xv[u][q].z - mu) * rs : 0.f;
              xh.w = k + 3 < Kc ? (xv[u][q].w - mu) * rs : 0.f;
              a.x = k < Kc ? fmaf(xh.x, g.x, bt.x) : 0.f;
              a.y = k + 1 < Kc ? fmaf(xh.y, g.y, bt.y) : 0.f;
              a.z = k + 2 < Kc ? fmaf(xh.z, g.z, bt.z) : 0.f;
              a.w = k + 3 < Kc ? fmaf(xh.w, g.w, bt.w) : 0.f;
              *reinterpret_cast<float4*>(s_panel + row * PS + k) = a;
              if (p.xhat_out && blockIdx.y == 0 && m0 + row < p.M && kp0 + k < p.ldxh)
                *reinterpret_cast<float4*>(p.xhat_out + (m0 + row) * p.ldxh + kp0 + k) = xh;
            }
          }
        }
    }
    lds_barrier();
    ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 1);

    auto compute = [&](Frag& S, int i) {
      const int kk = 32 * (c_begin + i) + 16 * h;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(s_panel + (32 * rt + l31) * PS + kk + 4 * q);
          a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], S.b[t], acc[rt], 0, 0, 0);
      }
    };
#define FWD_COMPUTE(S, i) compute(S, i)
    ULTR_STREAM_GEMM(nch, FWD_LOAD, FWD_COMPUTE);
  }
  ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 2);

  // --- split-K: every slice parks its accumulator in LDS; slice wk then owns registers [wk R, (wk+1) R) of the
  // tile (R = 16 / WK), adds them over the slices in slice order, and carries them through the epilogue, so
  // all waves of the workgroup share the activation math and the stores.
  constexpr int R = 16 / WK;
  float v[R];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {  // RT == 2: WK == 1 and no fused tail, the tiles go out one after the other
  if (WK > 1) {
    lds_barrier();  // every wave is done with the panel: the exchange buffer takes its place
    float* dst = s_acc + ((size_t)wk * WN + wn) * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[64 * r] = acc[rt][r];
    lds_barrier();
#pragma unroll
    for (int i = 0; i < R; ++i) {
      float sum = 0.f;
#pragma unroll
      for (int sl = 0; sl < WK; ++sl) sum += s_acc[((size_t)sl * WN + wn) * 1024 + 64 * (wk * R + i) + lane];
      v[i] = sum;
    }
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = acc[rt][i];
  }
  // --- epilogue: bias + activation in the accumulator layout (one column per lane)
  switch (p.activation) {
    case kActNone: _Pragma("unroll") for (int i = 0; i < R; ++i) v[i] = col_ok ? v[i] + bias : 0.f; break;
#define ULTR_ACT_CASE(A) case A: _Pragma("unroll") for (int i = 0; i < R; ++i) v[i] = col_ok ? act_fwd(A, v[i] + bias) : 0.f; break;
    ULTR_ACT_CASE(ULTR_ACT_ELU)
    ULTR_ACT_CASE(ULTR_ACT_RELU)
    ULTR_ACT_CASE(ULTR_ACT_SELU)
    ULTR_ACT_CASE(ULTR_ACT_TANH)
    default: _Pragma("unroll") for (int i = 0; i < R; ++i) v[i] = col_ok ? act_fwd(ULTR_ACT_SIGMOID, v[i] + bias) : 0.f; break;
#undef ULTR_ACT_CASE
  }
  if (!FUSE) {
    if (col < p.ldn) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const long long m = m0 + 32 * rt + acc_row(wk * R + i, h);
        if (m < p.M) p.act_out[m * p.ldn + col] = v[i];
      }
    }
  }
  }
  if (!FUSE) {
    ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 3);
    return;
  }

  // --- fused output layer: the WN column tiles of this workgroup hold complete rows (N <= 32 WN).  The tile
  // is turned through LDS so that a lane PAIR owns a row: 16 contiguous columns per lane give float4 stores
  // and in-lane row sums.
  float* s_tile = s_epi + wn * (32 * 36);
  float* s_part = s_epi + WN * 32 * 36;    // [32][WN][2] partial (sum, M2)
  float* s_sc = s_part + 32 * WN * 2;      // [32][WN]
  const int cnt = min(32, p.N - n0);
  const int ntile_out = (p.N + 31) >> 5;
#pragma unroll
  for (int i = 0; i < R; ++i) s_tile[acc_row(wk * R + i, h) * 36 + l31] = v[i];
  lds_barrier();
  const int e_row = lane >> 1, e_part = lane & 1;
  const long long e_m = m0 + e_row;
  const int e_c0 = n0 + 16 * e_part;
  float x[16];
  float rsum = 0.f, rm2 = 0.f;
  float4 og[4], ob[4], ow[4];
  if (wk == 0) {
    const int o_last4 = ((p.N + 3) & ~3) - 4;  // output-layer operands of this lane's 16 columns
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = min(e_c0 + 4 * q, o_last4);
      og[q] = *reinterpret_cast<const float4*>(p.out_gamma + c);
      ob[q] = *reinterpret_cast<const float4*>(p.out_beta + c);
      ow[q] = *reinterpret_cast<const float4*>(p.out_W + c);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(s_tile + e_row * 36 + 16 * e_part + 4 * q);
      x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
    }
    float s4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s4[q] = (x[4 * q] + x[4 * q + 1]) + (x[4 * q + 2] + x[4 * q + 3]);
    rsum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    rsum += ULTR_DPP_F(rsum, 0xB1);  // the partner lane holds the other 16 columns
    const float mu = cnt > 0 ? rsum / (float)cnt : 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float dv = (e_c0 + t < p.N) ? x[t] - mu : 0.f;
      rm2 += dv * dv;
    }
    rm2 += ULTR_DPP_F(rm2, 0xB1);
    if (cnt > 0 && e_m < p.M) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e_c0 + 4 * q < p.ldn)
          *reinterpret_cast<float4*>(p.act_out + e_m * p.ldn + e_c0 + 4 * q) = make_float4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
    }
    if (e_part == 0) *reinterpret_cast<float2*>(s_part + (e_row * WN + wn) * 2) = make_float2(rsum, rm2);
  }
  ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 3);
  lds_barrier();
  if (wk == 0) {  // every lane merges the column-group partials of its own row (Chan's formula)
    float total = 0.f;
#pragma unroll
    for (int g = 0; g < WN; ++g) total += g < ntile_out ? s_part[(e_row * WN + g) * 2] : 0.f;
    const float mu = total / (float)p.N;
    float m2 = 0.f;
#pragma unroll
    for (int g = 0; g < WN; ++g) {
      const int c = min(32, p.N - 32 * g);
      const float dd = s_part[(e_row * WN + g) * 2] / (float)max(c, 1) - mu;
      m2 += g < ntile_out ? s_part[(e_row * WN + g) * 2 + 1] + (float)c * dd * dd : 0.f;
    }
    float2 st = make_float2(0.f, 1.f);
    if (p.has_ln) st = make_float2(mu, 1.f / sqrtf(m2 / (float)p.N + p.eps));
    if (wn == 0 && e_part == 0 && e_m < p.M && p.has_ln) *reinterpret_cast<float2*>(p.out_stat + 2 * e_m) = st;
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float gq[4] = {og[q].x, og[q].y, og[q].z, og[q].w};
      const float bq[4] = {ob[q].x, ob[q].y, ob[q].z, ob[q].w};
      const float wq[4] = {ow[q].x, ow[q].y, ow[q].z, ow[q].w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xn = (x[4 * q + u] - st.x) * st.y;
        const float hk = p.has_ln ? xn * gq[u] + bq[u] : x[4 * q + u];
        dot += (e_c0 + 4 * q + u < p.N) ? hk * wq[u] : 0.f;
      }
    }
    dot += ULTR_DPP_F(dot, 0xB1);
    if (e_part == 0) s_sc[e_row * WN + wn] = dot;
  }
  lds_barrier();
  if (wk == 0 && wn == 0 && e_part == 0 && e_m < p.M) {
    float sc = p.out_bias[0];
    for (int g = 0; g < ntile_out; ++g) sc += s_sc[e_row * WN + g];
    p.scores[e_m] = sc;
  }
  ULTR_STAMP((p.K == 512 ? 1 : p.K == 256 ? 2 : 0), 4);
}

// ---------------------------------------------------------------------------------------------
// output layer (width 1):  score[m] = LN(x_m) . w + b          one wave per row
// ---------------------------------------------------------------------------------------------
struct OutArgs {
  const float* x; int ldx; const int* docids; int n_rows; const float* row_stats;
  float* stat_out; int has_ln; float eps;
  const float* gamma; const float* beta; const float* W; const float* bias;
  int K; long long M; float* scores;
};

// row source + (mean, rstd): layer-0 rows come with precomputed statistics, dense rows are reduced here
__device__ __forceinline__ float2 out_row_stats(const OutArgs& p, long long m, const float** src) {
  float2 st = make_float2(0.f, 1.f);
  if (p.docids) {
    const int d = p.docids[m];
    *src = d < p.n_rows ? p.x + (long long)d * p.ldx : nullptr;
    if (p.has_ln) st = make_float2(p.row_stats[2 * (long long)d], p.row_stats[2 * (long long)d + 1]);
  } else {
    *src = p.x + m * p.ldx;
    if (p.has_ln) {
      const int lane = threadIdx.x & 63;
      float sm = 0.f;
      for (int k = lane; k < p.K; k += 64) sm += (*src)[k];
      const float mean = wave_sum(sm) / (float)p.K;
      float m2 = 0.f;
      for (int k = lane; k < p.K; k += 64) {
        const float dd = (*src)[k] - mean;
        m2 += dd * dd;
      }
      st = make_float2(mean, 1.f / sqrtf(wave_sum(m2) / (float)p.K + p.eps));
    }
  }
  return st;
}

__global__ void __launch_bounds__(kThreads) fwd_out_kernel(OutArgs p) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  const float* src;
  const float2 st = out_row_stats(p, m, &src);
  if (p.has_ln && !p.docids && lane == 0) {
    p.stat_out[2 * m] = st.x;
    p.stat_out[2 * m + 1] = st.y;
  }
  float s = 0.f;
  for (int k = lane; k < p.K; k += 64) {
    const float x = src ? src[k] : 0.f;
    const float hk = p.has_ln ? (x - st.x) * st.y * p.gamma[k] + p.beta[k] : x;
    s += hk * p.W[k];
  }
  s = wave_sum(s);
  if (lane == 0) p.scores[m] = s + p.bias[0];
}

// ---------------------------------------------------------------------------------------------
// Forward chain: ALL layers of the scorer for 16 rows in one workgroup.
//
// A row's path through the network (gather -> LN -> GEMM -> bias -> act -> LN -> ... -> score) never needs another
// row, and 16 rows of every hidden width fit in LDS, so the per-layer kernels above -- each of which starts on a
// cold L2 and reads what the previous one has just written, the slow path of this memory system -- collapse into
// one launch whose activations never leave the CU between layers.  What still goes to memory is what the backward
// pass wants back: act_j, xhat_j, the row statistics.
//
//   * 8 waves, M / 16 workgroups (160 at the headline batch).  v_mfma_f32_16x16x4_f32; a wave owns a UNIT of 64
//     output columns = four 16x16 accumulators.  Column c of the unit sits in accumulator c % 4, slot c / 4, so one
//     float4 weight load per lane feeds four MFMAs and a lane ends up with 4 rows x 4 CONSECUTIVE columns
//     (float4 stores); along the contraction a lane's float4 of the LDS panel feeds four MFMA steps (step s of
//     16-chunk c contracts k = 16 c + 4 (lane / 16) + s on both operands).
//   * layers narrower than 8 units split the contraction over the idle waves (8 / units slices) and meet in LDS.
//   * weights stream from L2 (they are the same for every workgroup), three 16-deep chunks ahead, and the stream
//     of a layer starts before its input rows are normalised.
// ---------------------------------------------------------------------------------------------
#ifdef ULTR_STAMPS
#define CH_DBG(bit) (p.dbg & (bit))
#else
#define CH_DBG(bit) false
#endif
constexpr int kChainThreads = 512, kChainRows = 16, kChainExch = 7 * 1024;

struct ChainArgs {
  const float* table; int ldx; const int* docids; int n_rows; const float* row_stats;
  int has_ln; float eps; int activation; int H; long long M;
  const float* W[ULTR_MAX_LAYERS]; const float* bias[ULTR_MAX_LAYERS];
  const float* gamma[ULTR_MAX_LAYERS]; const float* beta[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  float* act_out[ULTR_MAX_LAYERS]; float* xhat_out[ULTR_MAX_LAYERS]; float* stat_out[ULTR_MAX_LAYERS];
  float* scores;
  int PS;          // LDS panel row stride in floats, == 4 (mod 32)
  int buf_floats;  // each of the two LDS buffers
  int dbg;         // diagnostic builds (-DULTR_STAMPS): 1 no weight loads, 2 no MFMAs, 4 no global stores
};


__global__ void __launch_bounds__(kChainThreads) fwd_chain_kernel(ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nn = lane & 15, kk = lane >> 4;
  const long long m0 = (long long)blockIdx.x * kChainRows;
  const int PS = p.PS;
  float* cur = smem;
  float* nxt = smem + p.buf_floats;
  struct Frag { float4 b[4]; } S0, S1, S2;
  ULTR_STAMP(5, 0);

  for (int j = 0; j <= p.H; ++j) {
    const int K = p.K[j], K16 = (K + 15) & ~15;
    const bool last = j == p.H;
    // ---- geometry of this layer's GEMM (unused for the output layer)
    const int N = last ? 1 : p.N[j], ldn = last ? 4 : p.ldn[j];
    const int U = (N + 63) >> 6;
    int Up = 1;
    while (Up < U && Up < 8) Up <<= 1;
    const int KS = U >= 8 ? 1 : 8 / Up;
    const int ks = U >= 8 ? 0 : w / Up;
    const int u_first = U >= 8 ? w : w % Up;
    const int nchunks = K16 >> 4, cps = (nchunks + KS - 1) / KS;
    const int c_begin = ks * cps, c_end = min(c_begin + cps, nchunks);
    const int nch = max(c_end - c_begin, 0), c_last = max(c_end - 1, 0);
    const float* __restrict__ Wj = p.W[j];
    int colc = min(64 * u_first + 4 * nn, ldn - 4);
#define CH_LOAD(i, S)                                                                              \
  do {                                                                                             \
    const int k0_ = 16 * min(c_begin + (i), c_last) + 4 * kk;                                      \
    if (CH_DBG(1)) { _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) S.b[s_] = make_float4(1.f, 2.f, 3.f, (float)k0_); } else \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                                               \
      S.b[s_] = *reinterpret_cast<const float4*>(Wj + (size_t)min(k0_ + s_, K - 1) * ldn + colc);  \
  } while (0)
    if (!last && u_first < U) {  // the weight stream starts before the rows are ready
      CH_LOAD(0, S0);
      CH_LOAD(1, S1);
    }

    // ---- input stage: wave w owns rows 2w and 2w+1; lanes run along the row (float4 each)
    if (j == 1) ULTR_STAMP(6, 0);
    {
      const int nq = (K16 + 255) >> 8;  // <= 4 (widths <= 1024)
      const int k_last4 = ((K + 3) & ~3) - 4, ldxh = (K + 3) & ~3;
      // gamma / beta (and the output layer's weights) are requested first: their latency runs under the statistics
      float4 gm[4], btq[4], woq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        btq[q] = woq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
          const int kc = min(4 * lane + 256 * q, k_last4);
          if (p.has_ln) {
            gm[q] = *reinterpret_cast<const float4*>(p.gamma[j] + kc);
            btq[q] = *reinterpret_cast<const float4*>(p.beta[j] + kc);
          }
          if (last) woq[q] = *reinterpret_cast<const float4*>(Wj + kc);
        }
      }
      float4 xv[2][4];
      float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
      bool pad[2] = {false, false};
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int row = 2 * w + r2;
        const long long m = min(m0 + row, p.M - 1);
        if (j == 0) {
          const int d = p.docids[m];
          pad[r2] = d >= p.n_rows;
          const float* src = p.table + (long long)(pad[r2] ? 0 : d) * p.ldx;
          if (p.has_ln) {
            mean[r2] = p.row_stats[2 * (long long)d];
            rstd[r2] = p.row_stats[2 * (long long)d + 1];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) xv[r2][q] = *reinterpret_cast<const float4*>(src + min(4 * lane + 256 * q, p.ldx - 4));
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) xv[r2][q] = *reinterpret_cast<const float4*>(cur + row * PS + min(4 * lane + 256 * q, K16 - 4));
        }
      }
      if (j > 0 && p.has_ln && !CH_DBG(8)) {  // two-pass statistics of the two rows, interleaved
        float sm[2], m2[2];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          sm[r2] = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              sm[r2] += (k < K ? xv[r2][q].x : 0.f) + (k + 1 < K ? xv[r2][q].y : 0.f) + (k + 2 < K ? xv[r2][q].z : 0.f) +
                        (k + 3 < K ? xv[r2][q].w : 0.f);
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
              const float d0 = k < K ? xv[r2][q].x - mean[r2] : 0.f, d1 = k + 1 < K ? xv[r2][q].y - mean[r2] : 0.f;
              const float d2 = k + 2 < K ? xv[r2][q].z - mean[r2] : 0.f, d3 = k + 3 < K ? xv[r2][q].w - mean[r2] : 0.f;
              m2[r2] = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, fmaf(d0, d0, m2[r2]))));
            }
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          rstd[r2] = 1.f / sqrtf(wave_sum(m2[r2]) / (float)K + p.eps);  // biased variance (keras LayerNormalization)
          const long long m = m0 + 2 * w + r2;
          if (lane == 0 && m < p.M) *reinterpret_cast<float2*>(p.stat_out[j] + 2 * m) = make_float2(mean[r2], rstd[r2]);
        }
      }
      if (j == 1) ULTR_STAMP(6, 1);
      float dot[2] = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < nq) {
          const int k = 4 * lane + 256 * q;
          const float4 g = gm[q], bt = btq[q], wo = woq[q];
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2) {
            const int row = 2 * w + r2;
            const float rs = pad[r2] ? 0.f : rstd[r2], mu = pad[r2] ? 0.f : mean[r2];
            float4 xh, a;  // (x - mean) * rstd * gamma + beta ; a zero feature row normalises to beta
            xh.x = k < K ? (xv[r2][q].x - mu) * rs : 0.f;
            xh.y = k + 1 < K ? (xv[r2][q].y - mu) * rs : 0.f;
            xh.z = k + 2 < K ? (xv[r2][q].z - mu) * rs : 0.f;
            xh.w = k + 3 < K ? (xv[r2][q].w - mu) * rs : 0.f;
            a.x = k < K ? fmaf(xh.x, g.x, bt.x) : 0.f;
            a.y = k + 1 < K ? fmaf(xh.y, g.y, bt.y) : 0.f;
            a.z = k + 2 < K ? fmaf(xh.z, g.z, bt.z) : 0.f;
            a.w = k + 3 < K ? fmaf(xh.w, g.w, bt.w) : 0.f;
            if (last) {
              dot[r2] = fmaf(a.w, wo.w, fmaf(a.z, wo.z, fmaf(a.y, wo.y, fmaf(a.x, wo.x, dot[r2]))));
            } else if (k < K16) {
              *reinterpret_cast<float4*>(cur + row * PS + k) = a;
              if (p.xhat_out[j] && m0 + row < p.M && k < ldxh && !CH_DBG(4))
                *reinterpret_cast<float4*>(p.xhat_out[j] + (m0 + row) * ldxh + k) = xh;
            }
          }
        }
      if (last) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const float sc = wave_sum(dot[r2]) + p.bias[j][0];
          const long long m = m0 + 2 * w + r2;
          if (lane == 0 && m < p.M) p.scores[m] = sc;
        }
        ULTR_STAMP(5, 7);
        break;
      }
    }
    ULTR_STAMP(5, min(1 + 2 * j, 6));
    if (j == 1) ULTR_STAMP(6, 2);
    lds_barrier();
    if (j == 1) ULTR_STAMP(6, 3);

    // ---- GEMM: this wave's units
    for (int u = u_first; u < U; u += 8) {
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u != u_first) {
        colc = min(64 * u + 4 * nn, ldn - 4);
        CH_LOAD(0, S0);
        CH_LOAD(1, S1);
      }
      // the bias of this lane's four columns rides along with the weight stream
      const float4 b4 = *reinterpret_cast<const float4*>(p.bias[j] + min(64 * u + 4 * nn, ((N + 3) & ~3) - 4));
      auto compute = [&](Frag& S, int i) {
        const float4 a4 = *reinterpret_cast<const float4*>(cur + nn * PS + 16 * (c_begin + i) + 4 * kk);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        if (CH_DBG(2)) {
#pragma unroll
          for (int s = 0; s < 4; ++s) acc[s][0] += a[s] * (S.b[s].x + S.b[s].y + S.b[s].z + S.b[s].w);
        } else
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].x, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].y, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].z, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].w, acc[3], 0, 0, 0);
        }
      };
#define CH_COMPUTE(S, i) compute(S, i)
      ULTR_STREAM_GEMM(nch, CH_LOAD, CH_COMPUTE);
      if (j == 1) ULTR_STAMP(6, 4);
#undef CH_COMPUTE

      if (KS > 1 && !CH_DBG(16)) {  // uniform over the workgroup (U < 8: exactly one unit per wave)
        lds_barrier();  // the panel has been multiplied out: the exchange buffer takes its place
        if (ks > 0) {
          float* dst = cur + ((ks - 1) * Up + u) * 1024 + lane * 4;
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(dst + 256 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        }
        lds_barrier();
        if (ks == 0) {
          for (int sl = 1; sl < KS; ++sl) {
            const float* srcp = cur + ((sl - 1) * Up + u) * 1024 + lane * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float4 v = *reinterpret_cast<const float4*>(srcp + 256 * t);
              acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
            }
          }
        }
      }
      if (j == 1) ULTR_STAMP(6, 5);
      if (ks == 0) {
        // epilogue: lane holds rows 4 kk + r (r = 0..3), columns col0 .. col0 + 3
        const int col0 = 64 * u + 4 * nn;
        const int n16 = (N + 15) & ~15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float4 z = make_float4(acc[0][r] + b4.x, acc[1][r] + b4.y, acc[2][r] + b4.z, acc[3][r] + b4.w);
          float4 v = CH_DBG(32) ? z : act4_rt(p.activation, z);
          v.x = col0 < N ? v.x : 0.f;
          v.y = col0 + 1 < N ? v.y : 0.f;
          v.z = col0 + 2 < N ? v.z : 0.f;
          v.w = col0 + 3 < N ? v.w : 0.f;
          const int row = 4 * kk + r;
          if (col0 < n16) *reinterpret_cast<float4*>(nxt + row * PS + col0) = v;
          if (col0 < ldn && m0 + row < p.M && !CH_DBG(4)) *reinterpret_cast<float4*>(p.act_out[j] + (m0 + row) * ldn + col0) = v;
        }
      }
    }
#undef CH_LOAD
    if (u_first >= U && KS > 1 && !CH_DBG(16)) {  // idle waves of a non-power-of-two unit count still meet the barriers
      lds_barrier();
      lds_barrier();
    }
    ULTR_STAMP(5, min(2 + 2 * j, 6));
    if (j == 1) ULTR_STAMP(6, 6);
    lds_barrier();
    if (j == 1) ULTR_STAMP(6, 7);
    float* t_ = cur; cur = nxt; nxt = t_;
  }
}

// ---------------------------------------------------------------------------------------------
// backward of the output layer: parameter-gradient partials + dz of the last hidden layer
// ---------------------------------------------------------------------------------------------
struct BwdOutArgs {
  OutArgs f;               // same operands as the forward (stat_out is read here)
  const float* stat_in;    // dense: merged stats written by fwd_out_kernel [M][2]
  const float* d_scores;   // [M]
  int activation;          // of the previous hidden layer
  int has_prev;            // 1: write dz for the previous hidden layer
  float* dz; int lddz;     // [M][lddz]
  int rows_per_block;
  float* partials;         // [gridDim.x][3K+4]: dW[K], dgamma[K], dbeta[K], db
  float* zero; long long zero_n4;  // the flat gradient buffer, cleared here (float4 units) instead of by a memset node
  // ultr_loss_dla_deferred: the launch carries the DLA loss.  Its first dla_pad workgroups (dla_blocks of them live)
  // are the loss kernel's own -- statistics, tower gradient, list weights, d_scores for whoever reads them later --
  // and the backward workgroups evaluate the d_score of each of their rows themselves (dla_lane) instead of
  // waiting for a loss launch to have written it: one launch less in the chain of dependent kernels.
  // The same for the IPW / naive loss (per-element weights) and the regression-EM loss (elementwise): `ride` names the kind.
  int ride;
  DlaArgs dla; IpwArgs ipw; RemArgs rem; PdArgs pd;
  int dla_blocks, dla_pad;
};

// generic fallback for very wide inputs (no hidden layer and more than 1024 features): LDS accumulators
__global__ void __launch_bounds__(kThreads) bwd_out_wide_kernel(BwdOutArgs p) {
  extern __shared__ float smem[];  // [4 waves][3K+1]
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < p.zero_n4; i += (long long)gridDim.x * kThreads)
    reinterpret_cast<float4*>(p.zero)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int K = p.f.K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float* acc = smem + w * (3 * K + 1);
  for (int i = lane; i < 3 * K + 1; i += 64) acc[i] = 0.f;
  const long long m_begin = (long long)blockIdx.x * p.rows_per_block;
  long long m_end = m_begin + p.rows_per_block;
  if (m_end > p.f.M) m_end = p.f.M;
  float db = 0.f;
  for (long long m = m_begin + w; m < m_end; m += 4) {
    const float go = p.d_scores[m];
    const float* src;
    float2 st = make_float2(0.f, 1.f);
    if (p.f.docids) {
      st = out_row_stats(p.f, m, &src);
    } else {
      src = p.f.x + m * p.f.ldx;
      if (p.f.has_ln) st = make_float2(p.stat_in[2 * m], p.stat_in[2 * m + 1]);
    }
    db += go;
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float x = src ? src[k] : 0.f;
      const float wk = p.f.W[k];
      if (p.f.has_ln) {
        const float xh = (x - st.x) * st.y;
        const float g = p.f.gamma[k];
        acc[k] += (g * xh + p.f.beta[k]) * go;  // dW
        acc[K + k] += go * wk * xh;             // dgamma
        acc[2 * K + k] += go * wk;              // dbeta
        s1 += go * wk * g;
        s2 += go * wk * g * xh;
      } else {
        acc[k] += x * go;
      }
    }
    if (p.has_prev) {
      if (p.f.has_ln) {
        s1 = wave_sum(s1) / (float)K;
        s2 = wave_sum(s2) / (float)K;
      }
      for (int k = lane; k < p.lddz; k += 64) {
        float v = 0.f;
        if (k < K) {
          const float a = src[k];
          float dx = go * p.f.W[k];
          if (p.f.has_ln) {
            const float xh = (a - st.x) * st.y;
            dx = st.y * (dx * p.f.gamma[k] - s1 - xh * s2);
          }
          v = dx * act_grad_from_out(p.activation, a);
        }
        p.dz[m * p.lddz + k] = v;
      }
    }
  }
  if (lane == 0) acc[3 * K] = db;
  __syncthreads();
  float* out = p.partials + (long long)blockIdx.x * (3 * K + 4);
  for (int i = threadIdx.x; i < 3 * K + 1; i += blockDim.x) {
    const int stride = 3 * K + 1;
    out[i] = smem[i] + smem[stride + i] + smem[2 * stride + i] + smem[3 * stride + i];
  }
}

// One wave per row, four rows in flight per wave (all their loads are issued before the first use); lane l
// owns columns k = l + 64 q and keeps the parameter-gradient sums of its columns in registers across the
// wave's rows.  The four waves of a workgroup meet in LDS and write one partial row; finalize adds the rows.
template <int KQ, int RIDE>
__global__ void __launch_bounds__(kThreads) bwd_out_kernel(BwdOutArgs p) {
  extern __shared__ float smem[];  // [4 waves][3K+1]
  constexpr bool DLA = RIDE == ULTR_RIDE_DLA;
  if (RIDE != ULTR_RIDE_NONE && (int)blockIdx.x < p.dla_pad) {
    __shared__ float s_acc[kMaxStat];
    __shared__ float s_part[4 * kMaxStat];
    __shared__ float s_tower[4 * 16];
    if ((int)blockIdx.x < p.dla_blocks) {
      if (RIDE == ULTR_RIDE_DLA) dla_loss_small_body(p.dla, blockIdx.x, p.dla_blocks, s_acc, s_part, s_tower);
      if (RIDE == ULTR_RIDE_IPW) ipw_loss_small_body(p.ipw, blockIdx.x, p.dla_blocks, s_acc, s_part);
      if (RIDE == ULTR_RIDE_REM) rem_loss_small_body(p.rem, blockIdx.x, p.dla_blocks, s_acc, s_part);
      if (RIDE == ULTR_RIDE_PD) pd_loss_small_body(p.pd, blockIdx.x, p.dla_blocks, s_acc, s_part, s_tower);
    }
    return;
  }
  const unsigned int bid = blockIdx.x - (RIDE != ULTR_RIDE_NONE ? p.dla_pad : 0);
  const unsigned int nbid = gridDim.x - (RIDE != ULTR_RIDE_NONE ? p.dla_pad : 0);
  const int K = p.f.K;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (long long i = (long long)bid * kThreads + threadIdx.x; i < p.zero_n4; i += (long long)nbid * kThreads)
    reinterpret_cast<float4*>(p.zero)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float tower_pw = 0.f;  // DLA: inverse propensity weight of rank lane & 15
  if (DLA) tower_pw = dla_tower(p.dla.prop_params, p.dla.L, lane & 15, p.dla.prob_kind, p.dla.max_pw).pw;
  float pd_tp = 1.f, pd_tm = 1.f;  // PD: t+ / t- of rank lane & 15
  if (RIDE == ULTR_RIDE_PD) {
    pd_tp = (lane & 15) < p.pd.L ? p.pd.t_plus[lane & 15] : 1.f;
    pd_tm = (lane & 15) < p.pd.L ? p.pd.t_minus[lane & 15] : 1.f;
  }
  float rem_bias = 0.f;
  unsigned int rem_step = 0u;
  if (RIDE == ULTR_RIDE_REM) {
    rem_bias = p.rem.bias_ptr ? p.rem.bias_ptr[0] : 0.f;
    rem_step = p.rem.step + (p.rem.step_ctr ? p.rem.step_ctr[0] : 0u);
  }
  float wv[KQ], gv[KQ], bv[KQ], aW[KQ], aG[KQ], aB[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = lane + 64 * q, kc = min(k, K - 1);
    const bool ok = k < K;
    wv[q] = ok ? p.f.W[kc] : 0.f;
    gv[q] = (ok && p.f.has_ln) ? p.f.gamma[kc] : (ok ? 1.f : 0.f);
    bv[q] = (ok && p.f.has_ln) ? p.f.beta[kc] : 0.f;
    aW[q] = aG[q] = aB[q] = 0.f;
  }
  float db = 0.f;
  const long long m_begin = (long long)bid * p.rows_per_block;
  long long m_end = m_begin + p.rows_per_block;
  if (m_end > p.f.M) m_end = p.f.M;
  const float inv_k = 1.f / (float)K;
  for (long long mb = m_begin + w; mb < m_end; mb += 16) {  // rows mb, mb + 4, mb + 8, mb + 12
    float x[4][KQ], go[4], mean[4], rstd[4];
    bool live[4];
    float go_group = 0.f;
    if (DLA) {  // lanes 16 u .. 16 u + 15 evaluate the list of row mb + 4 u, one rank each, and keep that row's d_score
      const long long mg = min(mb + 4 * (lane >> 4), m_end - 1);
      const int r = lane & 15, lg = (int)(mg / p.dla.B);
      const long long e = (long long)r * p.dla.B + (mg - (long long)lg * p.dla.B);
      const bool lv = r < p.dla.L;
      const float o = lv ? p.dla.scores[e] : 0.f, c = lv ? p.dla.clicks[e] : 0.f;
      const DlaLane v = dla_lane(o, c, tower_pw, lv, p.dla.loss_func);
      go_group = group_sum<16>(r == lg ? v.ds : 0.f);
    }
    if (RIDE == ULTR_RIDE_IPW) {  // the same with per-element weights
      const long long mg = min(mb + 4 * (lane >> 4), m_end - 1);
      const int r = lane & 15, lg = (int)(mg / p.ipw.B);
      const long long e = (long long)r * p.ipw.B + (mg - (long long)lg * p.ipw.B);
      const bool lv = r < p.ipw.L;
      const float o = lv ? p.ipw.scores[e] : 0.f, c = lv ? p.ipw.clicks[e] : 0.f, wgt = lv ? ipw_weight(p.ipw, e, r, c) : 0.f;
      const DlaLane v = dla_lane(o, c, wgt, lv, p.ipw.loss_func);
      go_group = group_sum<16>(r == lg ? v.ds : 0.f);
    }
    if (RIDE == ULTR_RIDE_PD) {  // every rank of the row's list against every other (group shuffles)
      const long long mg = min(mb + 4 * (lane >> 4), m_end - 1);
      const int r = lane & 15, lg = (int)(mg / p.pd.B);
      const long long e = (long long)r * p.pd.B + (mg - (long long)lg * p.pd.B);
      const bool lv = r < p.pd.L;
      const float o = lv ? p.pd.scores[e] : 0.f, c = lv ? p.pd.clicks[e] : 0.f;
      float unused[16];
      const float ds = pd_lane<false>(o, c, pd_tp, pd_tm, lv, r, p.pd.L, lane & 48, unused);
      go_group = group_sum<16>(r == lg ? ds : 0.f);
    }
    if (RIDE == ULTR_RIDE_REM) {  // elementwise: lane 16 u evaluates the row mb + 4 u
      const long long mg = min(mb + 4 * (lane >> 4), m_end - 1);
      const int lg = (int)(mg / p.rem.B);
      const long long bg = mg - (long long)lg * p.rem.B;
      go_group = rem_elem(p.rem, rem_bias, p.rem.propensity[lg], rem_step, bg, lg, mg).d;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long m = mb + 4 * u;
      live[u] = m < m_end;
      const long long mc = live[u] ? m : m_end - 1;
      go[u] = RIDE != ULTR_RIDE_NONE ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(go_group), 16 * u))
                                     : p.d_scores[mc];
      const float* src;
      mean[u] = 0.f; rstd[u] = 1.f;
      bool pad = false;
      if (p.f.docids) {
        const int d = p.f.docids[mc];
        pad = d >= p.f.n_rows;
        src = p.f.x + (long long)(pad ? 0 : d) * p.f.ldx;
        if (p.f.has_ln) {
          mean[u] = p.f.row_stats[2 * (long long)d];
          rstd[u] = p.f.row_stats[2 * (long long)d + 1];
        }
      } else {
        src = p.f.x + mc * p.f.ldx;
        if (p.f.has_ln) {
          mean[u] = p.stat_in[2 * mc];
          rstd[u] = p.stat_in[2 * mc + 1];
        }
      }
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const float t = src[min(lane + 64 * q, K - 1)];
        x[u][q] = pad ? 0.f : t;
      }
      if (!live[u]) go[u] = 0.f;
    }
    float s1[4], s2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s1[u] = s2[u] = 0.f;
      db += go[u];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const float gw = go[u] * wv[q];                 // zero beyond K (wv) and for rows beyond the block (go)
        if (p.f.has_ln) {
          const float xh = (x[u][q] - mean[u]) * rstd[u];
          aW[q] += (gv[q] * xh + bv[q]) * go[u];
          aG[q] += gw * xh;
          aB[q] += gw;
          s1[u] += gw * gv[q];
          s2[u] += gw * gv[q] * xh;
        } else {
          aW[q] += (lane + 64 * q < K ? x[u][q] : 0.f) * go[u];
        }
      }
    }
    if (p.has_prev) {
      if (p.f.has_ln) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s1[u] = wave_sum(s1[u]) * inv_k;
          s2[u] = wave_sum(s2[u]) * inv_k;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long m = mb + 4 * u;
        if (!live[u]) continue;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int k = lane + 64 * q;
          if (k < p.lddz) {
            float v = 0.f;
            if (k < K) {
              const float a = x[u][q];
              float dx = go[u] * wv[q];
              if (p.f.has_ln) dx = rstd[u] * (dx * gv[q] - s1[u] - (a - mean[u]) * rstd[u] * s2[u]);
              v = dx * act_grad_from_out(p.activation, a);
            }
            p.dz[m * p.lddz + k] = v;
          }
        }
      }
    }
  }
  float* acc = smem + w * (3 * K + 1);
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = lane + 64 * q;
    if (k < K) {
      acc[k] = aW[q];
      acc[K + k] = aG[q];
      acc[2 * K + k] = aB[q];
    }
  }
  if (lane == 0) acc[3 * K] = db;
  __syncthreads();
  float* out = p.partials + (long long)bid * (3 * K + 4);
  for (int i = threadIdx.x; i < 3 * K + 1; i += blockDim.x) {
    const int stride = 3 * K + 1;
    out[i] = smem[i] + smem[stride + i] + smem[2 * stride + i] + smem[3 * stride + i];
  }
}


__global__ void __launch_bounds__(kThreads) wgrad_kernel(WgradArgs p) { wgrad_body(p, blockIdx.x, blockIdx.y, blockIdx.z); }

// ---------------------------------------------------------------------------------------------
// dgrad:  dh[m][k] = sum_n dz[m][n] * W[k][n]        tile 32 rows x 128 cols (k)
// ---------------------------------------------------------------------------------------------
struct DgradArgs {
  const float* dz; int ldn; int N;   // [M][ldn] (contraction over n < N)
  const float* W;                    // [K][ldn]
  int K; int ldk;                    // output width and its padded stride
  long long M;
  float* dh;                         // [M][ldk]
};

__device__ __forceinline__ void dgrad_body(const DgradArgs& p, int bx, int by) {
  __shared__ __attribute__((aligned(16))) float As[2][32 * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][128 * LDK];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const long long m0 = (long long)bx * 32;
  const int k0 = by * 128;
  const int a_row = tid >> 3, a_q = tid & 7;  // A: 32 rows x 8 quads; B: 128 rows x 8 quads = 4 per thread
  struct Regs {
    float4 a, b[4];
    int n0;
  } R0, R1;
  const long long a_mc = (m0 + a_row < p.M) ? m0 + a_row : p.M - 1;
  const int n_last4 = p.ldn - 4;
  auto load_chunk = [&](int chunk, Regs& r) {  // unconditional loads on clamped addresses, masked at store time
    r.n0 = chunk * BK;
    r.a = *reinterpret_cast<const float4*>(p.dz + a_mc * p.ldn + min(r.n0 + 4 * a_q, n_last4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      const int row = min(k0 + (f >> 3), p.K - 1), nb = min(r.n0 + 4 * (f & 7), n_last4);
      r.b[i] = *reinterpret_cast<const float4*>(p.W + (long long)row * p.ldn + nb);
    }
  };
  auto store_chunk = [&](Regs& r, int buf) {
    float4 av = r.a;
    if (m0 + a_row >= p.M || r.n0 + 4 * a_q >= p.ldn) av = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(&As[buf][a_row * LDK + 4 * a_q]) = av;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      float4 v = r.b[i];
      if (k0 + (f >> 3) >= p.K || r.n0 + 4 * (f & 7) >= p.ldn) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&Bs[buf][(f >> 3) * LDK + 4 * (f & 7)]) = v;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto compute = [&](int buf) {
    float a[16], b[16];
    load_frag_kcontig(As[buf], l31, h, a);
    load_frag_kcontig(Bs[buf], w * 32 + l31, h, b);
    mma16(a, b, acc);
  };
  const int nchunk = (p.N + BK - 1) / BK;
#define DG_LOAD(c, R) load_chunk(c, R)
#define DG_STORE(R, b) store_chunk(R, b)
#define DG_COMPUTE(b) compute(b)
  ULTR_GEMM_PIPELINE(nchunk, DG_LOAD, DG_STORE, DG_COMPUTE);
  const int col = k0 + w * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long m = m0 + acc_row(r, h);
    if (m < p.M && col < p.ldk) p.dh[m * p.ldk + col] = acc[r];
  }
}

__global__ void __launch_bounds__(kThreads, 2) dgrad_kernel(DgradArgs p) { dgrad_body(p, blockIdx.x, blockIdx.y); }

// wgrad_j and dgrad_j both consume dz_j and nothing of each other: one launch runs the two sets of workgroups side
// by side (dgrad's come first: they run longer), so the pair costs max(wgrad, dgrad) instead of their sum plus a
// launch gap.  Both bodies fit 256 registers, i.e. two workgroups per CU, one of each kind where the mix allows.
struct BwdPairArgs {
  WgradArgs g;
  DgradArgs d;
  int d_gx, d_gy;          // dgrad grid
  int g_gx, g_gy, g_gz;    // wgrad grid
};
__global__ void __launch_bounds__(kThreads, 2) bwd_pair_kernel(BwdPairArgs p) {
  const int nd = p.d_gx * p.d_gy;
  int id = blockIdx.x;
  if (id < nd) {
    dgrad_body(p.d, id % p.d_gx, id / p.d_gx);
  } else {
    id -= nd;
    wgrad_body(p.g, id % p.g_gx, (id / p.g_gx) % p.g_gy, id / (p.g_gx * p.g_gy));
  }
}

// The in-step metrics ride along with a weight-gradient launch (ultr_metrics_deferred): they need only the scores
// and the list weights, so their 12 us of sort-and-sum latency on 16 CUs runs under a GEMM that is busy anyway
// instead of holding a place of its own in the chain of dependent launches.  The rider workgroups come FIRST
// (dispatched first, they are the longer pole), padded to a multiple of 8 so that the GEMM's workgroup-id -> XCD
// mapping is unchanged.
struct WgradRiderArgs {
  WgradArgs g;
  int g_gx, g_gy, g_gz;
  MetricsArgs m;
  int m_blocks, m_pad;
};
__global__ void __launch_bounds__(kThreads) wgrad_rider_kernel(WgradRiderArgs p) {
  int id = blockIdx.x;
  if (id >= p.m_pad) {
    id -= p.m_pad;
    wgrad_body(p.g, id % p.g_gx, (id / p.g_gx) % p.g_gy, id / (p.g_gx * p.g_gy));
    return;
  }
  if (id >= p.m_blocks) return;
  __shared__ float s_vals[4 * kMaxOut];
  metrics_short_lists(p.m, id, p.m_blocks, s_vals);
  __syncthreads();
  metrics_finish(p.m, id, p.m_blocks, s_vals, kMetThreads / 64);
}

// ---------------------------------------------------------------------------------------------
// row pass: dz = LayerNorm-backward(dh) * act'(a), in place.  One wave per row.
//   g = dh*gamma ; dx = rstd * (g - mean(g) - xhat*mean(g*xhat)) ; dz = dx * act'(a)
// ---------------------------------------------------------------------------------------------
struct RowpassArgs {
  float* dh; int ldk; int K; long long M;
  const float* a;          // activations of the layer whose output this is [M][ldk]
  const float* stat;       // (mean, rstd) of those rows [M][2] (LN of the NEXT layer's input)
  const float* gamma;
  int has_ln; int activation;
};

// Register-resident: every load of a row (dh, a, gamma, statistics) is issued before the first use, so the row
// costs one memory round trip instead of one per 64-column step; KQ = columns per lane.
template <int KQ>
__global__ void __launch_bounds__(kThreads) rowpass_kernel(RowpassArgs p) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= p.M) return;
  float* row = p.dh + m * p.ldk;
  const float* a = p.a + m * p.ldk;
  float dv[KQ], av[KQ], gv[KQ];
  float mean = 0.f, rstd = 1.f;
  if (p.has_ln) {
    mean = p.stat[2 * m];
    rstd = p.stat[2 * m + 1];
  }
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int kc = min(lane + 64 * q, p.K - 1);
    dv[q] = row[kc];
    av[q] = a[kc];
    gv[q] = p.has_ln ? p.gamma[kc] : 1.f;
  }
  float c1 = 0.f, c2 = 0.f;
  if (p.has_ln) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const bool ok = lane + 64 * q < p.K;
      const float g = ok ? dv[q] * gv[q] : 0.f;
      c1 += g;
      c2 += g * (av[q] - mean) * rstd;
    }
    c1 = wave_sum(c1) / (float)p.K;
    c2 = wave_sum(c2) / (float)p.K;
  }
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int k = lane + 64 * q;
    if (k < p.K) {
      float dx = dv[q];
      if (p.has_ln) dx = rstd * (dx * gv[q] - c1 - (av[q] - mean) * rstd * c2);
      row[k] = dx * act_grad_from_out(p.activation, av[q]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward chain: the dgrad GEMMs and the row passes of ALL hidden layers for 16 rows in one workgroup.
//
// dz_{j-1} = rowpass_j( dz_j . W_j^T ) is a per-row recurrence just like the forward pass, so for small batches the
// chain  dgrad_{H-1} -> row pass -> dgrad_{H-2} -> row pass -> ... -> dz_0  runs out of LDS in one launch
// (same workgroup shape, MFMA tiling and weight streaming as fwd_chain_kernel).  It starts from the dz_{H-1} that
// bwd_out_kernel wrote and leaves every dz_j in its own buffer for the weight-gradient GEMMs that follow.
//   B operand: W_j is row-major [K_j][ldn] and the contraction runs along its rows, so a lane's float4 holds four
//   contraction STEPS of one output column (forward: four columns of one step); the lane's four accumulators still
//   end up as 4 rows x 4 consecutive columns.
// ---------------------------------------------------------------------------------------------
struct BwdChainArgs {
  int has_ln, activation, H;
  long long M;
  const float* W[ULTR_MAX_LAYERS];      // W_j   [K_j][ldn_j]
  const float* gamma[ULTR_MAX_LAYERS];  // gamma of layer j's input LayerNorm [K_j]
  const float* act[ULTR_MAX_LAYERS];    // activations of hidden layer j [M][ldn_j]
  const float* stat[ULTR_MAX_LAYERS];   // (mean, rstd) of layer j's input rows [M][2], j >= 1
  float* dz[ULTR_MAX_LAYERS];           // dz_j [M][ldn_j]: dz_{H-1} is read, the others are written
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  int PS, buf_floats;
};

__global__ void __launch_bounds__(kChainThreads) bwd_chain_kernel(BwdChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nn = lane & 15, kk = lane >> 4;
  const long long m0 = (long long)blockIdx.x * kChainRows;
  const int PS = p.PS;
  float* cur = smem;
  float* nxt = smem + p.buf_floats;
  struct Frag { float4 b[4]; } S0, S1, S2;
  ULTR_STAMP(7, 0);

  {  // dz of the last hidden layer (written by bwd_out_kernel) -> LDS, zero beyond its width
    const int j = p.H - 1, N = p.N[j], ld = p.ldn[j], N16 = (N + 15) & ~15;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = 2 * w + r2;
      const long long m = min(m0 + row, p.M - 1);
      for (int k = 4 * lane; k < N16; k += 256) {
        float4 v = *reinterpret_cast<const float4*>(p.dz[j] + m * ld + min(k, ld - 4));
        v.x = k < N ? v.x : 0.f;
        v.y = k + 1 < N ? v.y : 0.f;
        v.z = k + 2 < N ? v.z : 0.f;
        v.w = k + 3 < N ? v.w : 0.f;
        *reinterpret_cast<float4*>(cur + row * PS + k) = v;
      }
    }
  }

  for (int j = p.H - 1; j >= 1; --j) {
    // dy[16][K] = dz_j[16][Nc] . W_j^T : output columns c < K (= width of layer j-1), contraction n < Nc
    const int K = p.K[j], Nc = p.N[j], ldw = p.ldn[j], ldo = p.ldn[j - 1];
    const int K16 = (K + 15) & ~15;
    const int U = (K + 63) >> 6;
    int Up = 1;
    while (Up < U && Up < 8) Up <<= 1;
    const int KS = U >= 8 ? 1 : 8 / Up;
    const int ks = U >= 8 ? 0 : w / Up;
    const int u_first = U >= 8 ? w : w % Up;
    const int nchunks = (Nc + 15) >> 4, cps = (nchunks + KS - 1) / KS;
    const int c_begin = ks * cps, c_end = min(c_begin + cps, nchunks);
    const int nch = max(c_end - c_begin, 0), c_last = max(c_end - 1, 0);
    const float* __restrict__ Wj = p.W[j];
    int crow = 64 * u_first + 4 * nn;  // first of this lane's four output columns = four rows of W_j
#define BC_LOAD(i, S)                                                                                  \
  do {                                                                                                 \
    const int n0_ = min(16 * min(c_begin + (i), c_last) + 4 * kk, ldw - 4);                            \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_)                                                   \
      S.b[t_] = *reinterpret_cast<const float4*>(Wj + (size_t)min(crow + t_, K - 1) * ldw + n0_);      \
  } while (0)
    if (u_first < U) {
      BC_LOAD(0, S0);
      BC_LOAD(1, S1);
    }
    // what the row pass will need, requested now: the activations of layer j-1 (two rows per wave), gamma, statistics
    const int nq = (K16 + 255) >> 8;  // <= 4
    const int k_last4 = ((K + 3) & ~3) - 4;
    float4 hv[2][4], gm[4];
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (q < nq) {
        const int kc = min(4 * lane + 256 * q, k_last4);
        if (p.has_ln) gm[q] = *reinterpret_cast<const float4*>(p.gamma[j] + kc);
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const long long m = min(m0 + 2 * w + r2, p.M - 1);
          hv[r2][q] = *reinterpret_cast<const float4*>(p.act[j - 1] + m * ldo + min(4 * lane + 256 * q, ldo - 4));
        }
      }
    }
    if (p.has_ln) {
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const long long m = min(m0 + 2 * w + r2, p.M - 1);
        const float2 st = *reinterpret_cast<const float2*>(p.stat[j] + 2 * m);
        mean[r2] = st.x;
        rstd[r2] = st.y;
      }
    }
    lds_barrier();  // the panel (first layer: the staged dz, later: the row pass output) is complete
    ULTR_STAMP(7, min(1 + 2 * (p.H - 1 - j), 6));

    for (int u = u_first; u < U; u += 8) {
      f32x4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u != u_first) {
        crow = 64 * u + 4 * nn;
        BC_LOAD(0, S0);
        BC_LOAD(1, S1);
      }
      auto compute = [&](Frag& S, int i) {
        const float4 a4 = *reinterpret_cast<const float4*>(cur + nn * PS + 16 * (c_begin + i) + 4 * kk);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, S.b[0].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, S.b[1].x, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, S.b[2].x, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, S.b[3].x, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, S.b[0].y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, S.b[1].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, S.b[2].y, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, S.b[3].y, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, S.b[0].z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, S.b[1].z, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, S.b[2].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, S.b[3].z, acc[3], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, S.b[0].w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, S.b[1].w, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, S.b[2].w, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, S.b[3].w, acc[3], 0, 0, 0);
      };
#define BC_COMPUTE(S, i) compute(S, i)
      ULTR_STREAM_GEMM(nch, BC_LOAD, BC_COMPUTE);
#undef BC_COMPUTE

      if (KS > 1) {  // uniform over the workgroup (U < 8: exactly one unit per wave)
        lds_barrier();  // the panel has been multiplied out: the exchange buffer takes its place
        if (ks > 0) {
          float* dst = cur + ((ks - 1) * Up + u) * 1024 + lane * 4;
#pragma unroll
          for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(dst + 256 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        }
        lds_barrier();
        if (ks == 0) {
          for (int sl = 1; sl < KS; ++sl) {
            const float* srcp = cur + ((sl - 1) * Up + u) * 1024 + lane * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const float4 v = *reinterpret_cast<const float4*>(srcp + 256 * t);
              acc[t][0] += v.x; acc[t][1] += v.y; acc[t][2] += v.z; acc[t][3] += v.w;
            }
          }
        }
      }
      if (ks == 0) {  // lane holds rows 4 kk + r (r = 0..3), columns col0 .. col0 + 3 of dy
        const int col0 = 64 * u + 4 * nn;
        if (col0 < K16) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(nxt + (4 * kk + r) * PS + col0) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
      }
    }
#undef BC_LOAD
    if (u_first >= U && KS > 1) {  // idle waves of a non-power-of-two unit count still meet the barriers
      lds_barrier();
      lds_barrier();
    }
    ULTR_STAMP(7, min(2 + 2 * (p.H - 1 - j), 6));
    lds_barrier();

    // ---- row pass on rows 2w, 2w+1 of dy:  g = dy*gamma ; dx = rstd (g - mean(g) - xhat mean(g xhat)) ; dz = dx act'(h)
    {
      float4 dv[2][4];
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < nq) dv[r2][q] = *reinterpret_cast<const float4*>(nxt + (2 * w + r2) * PS + min(4 * lane + 256 * q, K16 - 4));
      float c1[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
      if (p.has_ln) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              const float dd[4] = {dv[r2][q].x, dv[r2][q].y, dv[r2][q].z, dv[r2][q].w};
              const float gg[4] = {gm[q].x, gm[q].y, gm[q].z, gm[q].w};
              const float hh[4] = {hv[r2][q].x, hv[r2][q].y, hv[r2][q].z, hv[r2][q].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float g = k + e < K ? dd[e] * gg[e] : 0.f;
                c1[r2] += g;
                c2[r2] += g * (hh[e] - mean[r2]) * rstd[r2];
              }
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
            const float dd[4] = {dv[r2][q].x, dv[r2][q].y, dv[r2][q].z, dv[r2][q].w};
            const float gg[4] = {gm[q].x, gm[q].y, gm[q].z, gm[q].w};
            const float hh[4] = {hv[r2][q].x, hv[r2][q].y, hv[r2][q].z, hv[r2][q].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float dx = dd[e];
              if (p.has_ln) dx = rstd[r2] * (dx * gg[e] - c1[r2] - (hh[e] - mean[r2]) * rstd[r2] * c2[r2]);
              o[e] = k + e < K ? dx * act_grad_from_out(p.activation, hh[e]) : 0.f;
            }
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            if (k < K16) *reinterpret_cast<float4*>(nxt + row * PS + k) = ov;
            if (k < ldo && m0 + row < p.M) *reinterpret_cast<float4*>(p.dz[j - 1] + (m0 + row) * ldo + k) = ov;
          }
      }
    }
    float* t_ = cur; cur = nxt; nxt = t_;
    // the barrier at the top of the next layer (or the end of the kernel) orders these LDS writes
  }
  ULTR_STAMP(7, 7);
}

__global__ void __launch_bounds__(kThreads) finalize_kernel(FinalizeArgs p) { finalize_body(p, blockIdx.x); }

__global__ void __launch_bounds__(kThreads) finalize_rider_kernel(FinalizeRiderArgs p) {
  const int id = blockIdx.x;
  if (id >= p.a_pad) {
    finalize_body(p.f, id - p.a_pad);
    return;
  }
  if (id < p.a_blocks) assemble_slots_body(p.a, id, p.a_blocks);
}

}  // namespace ultr

using namespace ultr;

#ifdef ULTR_STAMPS
extern "C" int ultr_debug_stamps(long long* host, int n) {
  if (n > 16 * 1024 * 8) n = 16 * 1024 * 8;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * (size_t)n);
  return n;
}
#endif

template <int WN, int WK, bool FUSE, bool GATHER, int QMAX, int RT = 1>
static int launch_fwd_q(const FwdArgs& a, hipStream_t st, const char* name) {
  const size_t smem = sizeof(float) * fwd_smem_floats<WN, WK, RT>(a.K, FUSE);
  static PerDeviceOnce attr_set;  // the panel of a 512-wide layer is already past the 64 KB default
  if (const int rc_ = raise_lds_limit(attr_set, reinterpret_cast<const void*>(&fwd_layer_kernel<WN, WK, FUSE, GATHER, QMAX, RT>),
                                      160 * 1024, "ultr_dnn_forward")) return rc_;
  ULTR_REQUIRE(smem <= 160 * 1024, ULTR_ELIMIT, "ultr_dnn_forward: layer needs %zu bytes of LDS", smem);
  if (ultr::g_profile_on)  // name the instantiation and the GEMM shape the way rocprofv3 sees the kernel
    name = ultr::profile_intern("%s<%d, %d, %s, %s, %d, %d>/K%dN%d", name, WN, WK, FUSE ? "true" : "false", GATHER ? "true" : "false",
                                QMAX, RT, a.K, a.N);
  {
    ultr::ProfScope prof_(name, st);
    hipLaunchKernelGGL((fwd_layer_kernel<WN, WK, FUSE, GATHER, QMAX, RT>), dim3((unsigned)cdiv(a.M, 32 * RT), (unsigned)cdiv(a.N, 32 * WN)),
                       dim3(64 * WN * WK), smem, st, a);
  }
  return check_launch(name);
}
// ULTR_FWD_ROWTILES=1: one row tile per workgroup at every size
static bool two_row_tiles(const FwdArgs& a, size_t smem_floats) {
  static const bool on = [] { const char* e = getenv("ULTR_FWD_ROWTILES"); return !(e && e[0] == '1'); }();
  return on && a.M >= 32768 && smem_floats * sizeof(float) <= 160 * 1024;
}
template <int WN, int WK, bool FUSE, bool GATHER>
static int launch_fwd_g(const FwdArgs& a, hipStream_t st, const char* name) {
  if constexpr (WK == 1 && !FUSE) {  // large batches: 64 rows per workgroup, two accumulator tiles per wave
    if (two_row_tiles(a, fwd_smem_floats<WN, WK, 2>(a.K, FUSE)))
      return a.K <= 512 ? launch_fwd_q<WN, WK, FUSE, GATHER, 2, 2>(a, st, name) : launch_fwd_q<WN, WK, FUSE, GATHER, 4, 2>(a, st, name);
  }
  return a.K <= 512 ? launch_fwd_q<WN, WK, FUSE, GATHER, 2>(a, st, name) : launch_fwd_q<WN, WK, FUSE, GATHER, 4>(a, st, name);
}
template <int WN, int WK, bool FUSE>
static int launch_fwd(const FwdArgs& a, hipStream_t st, const char* name) {
  if constexpr (FUSE) return launch_fwd_g<WN, WK, FUSE, false>(a, st, name);  // the fused variants are built for dense inputs only
  else
  return a.docids ? launch_fwd_g<WN, WK, FUSE, true>(a, st, name) : launch_fwd_g<WN, WK, FUSE, false>(a, st, name);
}
// Pick the wave arrangement: about one 32x32 tile per SIMD-slot of the chip (1024 waves), splitting the
// contraction inside the workgroup when the layer has fewer tiles than that.
static int launch_fwd_layer(const FwdArgs& a, bool fuse, hipStream_t st, const char* name = "fwd_layer_kernel") {
  const long long tilesM = cdiv(a.M, 32), tilesN = cdiv(a.N, 32);
  const int chunks = (int)cdiv(a.K < kPanelMaxK ? a.K : kPanelMaxK, 32);
  if (fuse) {
    const int WN = tilesN <= 1 ? 1 : tilesN <= 2 ? 2 : 4;
    const int WK = (tilesM * WN < 1024 && chunks >= 4) ? 2 : 1;
    if (WN == 1) return WK == 1 ? launch_fwd<1, 1, true>(a, st, name) : launch_fwd<1, 2, true>(a, st, name);
    if (WN == 2) return WK == 1 ? launch_fwd<2, 1, true>(a, st, name) : launch_fwd<2, 2, true>(a, st, name);
    return WK == 1 ? launch_fwd<4, 1, true>(a, st, name) : launch_fwd<4, 2, true>(a, st, name);
  }
  const long long base = tilesM * tilesN;
  // measured at M = 2560 (tools/kbench, rocprofv3): layer 0 (K136 N512) 8x1 13.1 us, 4x1 14.7, 4x2 18.5, 2x2 20.7;
  // layer 1 (K512 N256) 4x2 18.7 us, 2x2 21.7, 4x1 22.8, 8x1 32.5 -- wider workgroups share one staged panel
  if (base >= 1024 || chunks < 4) return tilesN >= 8 ? launch_fwd<8, 1, false>(a, st, name) : launch_fwd<4, 1, false>(a, st, name);
  if (base * 2 >= 1024 || chunks < 8) return tilesN >= 4 ? launch_fwd<4, 2, false>(a, st, name) : launch_fwd<2, 2, false>(a, st, name);
  return launch_fwd<1, 4, false>(a, st, name);
}

#include "bn_body.hpp"

int ultr_check_desc(const UltrDnnDesc* d, const char* who) {
  ULTR_REQUIRE(d, ULTR_EINVAL, "%s: null descriptor", who);
  ULTR_REQUIRE(d->n_layers >= 1 && d->n_layers <= ULTR_MAX_LAYERS, ULTR_ELIMIT, "%s: n_layers=%d not in [1,%d]", who,
               d->n_layers, ULTR_MAX_LAYERS);
  ULTR_REQUIRE(d->in_features >= 1, ULTR_EINVAL, "%s: in_features=%d", who, d->in_features);
  ULTR_REQUIRE(d->sizes[d->n_layers - 1] == 1, ULTR_EINVAL, "%s: output_size must be 1 (scores are [B,1], DNN.py:33)",
               who);
  for (int j = 0; j + 1 < d->n_layers; ++j)
    ULTR_REQUIRE(d->sizes[j] >= 1 && d->sizes[j] <= ULTR_MAX_WIDTH, ULTR_ELIMIT, "%s: hidden width %d not in [1,%d]",
                 who, d->sizes[j], ULTR_MAX_WIDTH);
  ULTR_REQUIRE(d->activation >= ULTR_ACT_ELU && d->activation <= ULTR_ACT_SIGMOID, ULTR_EINVAL, "%s: activation=%d",
               who, d->activation);
  ULTR_REQUIRE(d->norm >= ULTR_NORM_NONE && d->norm <= ULTR_NORM_BATCH_EVAL, ULTR_EINVAL, "%s: norm=%d", who, d->norm);
  ULTR_REQUIRE(d->compute_dtype == ULTR_DTYPE_F32 || d->compute_dtype == ULTR_DTYPE_BF16, ULTR_EINVAL,
               "%s: compute_dtype=%d", who, d->compute_dtype);
  for (int j = 0; j < d->n_layers; ++j)
    ULTR_REQUIRE((d->off_W[j] & 3) == 0 && (d->off_b[j] & 3) == 0 && (d->off_gamma[j] & 3) == 0 &&
                     (d->off_beta[j] & 3) == 0,
                 ULTR_EINVAL, "%s: parameter offsets must be multiples of 4 floats", who);
  return ULTR_OK;
}

extern "C" long long ultr_dnn_workspace_bytes(const UltrDnnDesc* desc, long long M) {
  if (ultr_check_desc(desc, "ultr_dnn_workspace_bytes") != ULTR_OK || M <= 0) return -1;
  return make_layout(desc, M).total * (long long)sizeof(float);
}

extern "C" int ultr_ln_row_stats(const float* x, long long n_rows, int F, int ldx, float eps, float* stats,
                                 void* stream) {
  ULTR_REQUIRE(x && stats, ULTR_EINVAL, "ultr_ln_row_stats: null pointer");
  ULTR_REQUIRE(n_rows >= 0 && F > 0 && ldx >= F, ULTR_EINVAL, "ultr_ln_row_stats: bad sizes n_rows=%lld F=%d ldx=%d",
               n_rows, F, ldx);
  {
    ultr::ProfScope prof_("ln_row_stats_kernel", as_stream(stream));
    hipLaunchKernelGGL(ln_row_stats_kernel, dim3((unsigned)cdiv(n_rows + 1, 4)), dim3(kThreads), 0, as_stream(stream), x,
                     n_rows, F, ldx, eps, stats);
  }
  return check_launch("ln_row_stats_kernel");
}

int ultr_common_checks(const char* who, const UltrDnnDesc* desc, const float* params, const float* table,
                         const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                         void* workspace) {
  int rc = ultr_check_desc(desc, who);
  if (rc) return rc;
  ULTR_REQUIRE(params && table && docids && workspace, ULTR_EINVAL, "%s: null pointer", who);
  ULTR_REQUIRE(desc->norm != ULTR_NORM_LAYER || table_stats, ULTR_EINVAL, "%s: table_stats required with norm=layer",
               who);
  ULTR_REQUIRE(M > 0 && n_rows >= 0, ULTR_EINVAL, "%s: bad sizes M=%lld n_rows=%d", who, M, n_rows);
  ULTR_REQUIRE(ldx >= desc->in_features && (ldx & 3) == 0, ULTR_EINVAL,
               "%s: table row stride %d must be >= F=%d and a multiple of 4", who, ldx, desc->in_features);
  ULTR_REQUIRE(((uintptr_t)params & 15) == 0 && ((uintptr_t)table & 15) == 0 && ((uintptr_t)workspace & 255) == 0,
               ULTR_EINVAL, "%s: params/table must be 16-byte and workspace 256-byte aligned", who);
  ULTR_REQUIRE(M / 32 + 1 < 2147483647ll, ULTR_ELIMIT, "%s: M=%lld too large", who, M);
  return ULTR_OK;
}

extern "C" int ultr_dnn_forward(const UltrDnnDesc* desc, const float* params, const float* table,
                                const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                                void* workspace, float* scores, void* stream) {
  int rc = ultr_common_checks("ultr_dnn_forward", desc, params, table, table_stats, n_rows, ldx, docids, M, workspace);
  if (rc) return rc;
  ULTR_REQUIRE(scores, ULTR_EINVAL, "ultr_dnn_forward: null scores");
  const Layout L = make_layout(desc, M);
  float* ws = reinterpret_cast<float*>(workspace);
  const int has_ln = desc->norm == ULTR_NORM_LAYER;
  hipStream_t st = as_stream(stream);
  static const bool chain_on = [] { const char* e = getenv("ULTR_FWD_CHAIN"); return !(e && e[0] == '0'); }();
  size_t chain_smem = 0;
  int chain_ps = 0, chain_buf = 0;
  // The chain wins while its M / 16 workgroups are a few waves of the 256 CUs (measured, fwd kernels only, F = 136,
  // [512, 256, 128]: M = 2560 41.6 us against 47.6; 10240 117 against 125; 20480 193 against 191; 40960 378 against 335):
  // beyond that the per-layer kernels' 32-row tiles, which stream every weight half as often, are the better plan.
  // ULTR_FWD_CHAIN=1 forces the chain at any size.
  static const bool chain_force = [] { const char* e = getenv("ULTR_FWD_CHAIN"); return e && e[0] == '1'; }();
  const bool bn = L.bn, bn_train = desc->norm == ULTR_NORM_BATCH;  // batch normalisation: the per-layer plan (bn_body.hpp)
  if (!bn && chain_on && (M <= 16384 || chain_force) && L.H >= 1 && L.K[0] <= 1024) {  // hidden widths are <= 1024 (check_desc)
    int maxw = L.K[0];
    for (int j = 0; j < L.H; ++j) maxw = L.N[j] > maxw ? L.N[j] : maxw;
    chain_ps = ((maxw + 31) & ~31) + 4;
    chain_buf = kChainRows * chain_ps > kChainExch ? kChainRows * chain_ps : kChainExch;
    chain_smem = 2 * (size_t)chain_buf * sizeof(float);
  }
  if (chain_smem > 0 && chain_smem <= 160 * 1024) {  // else: the per-layer kernels below
    ChainArgs c{};
    c.table = table; c.ldx = ldx; c.docids = docids; c.n_rows = n_rows; c.row_stats = table_stats;
    c.has_ln = has_ln; c.eps = desc->ln_eps; c.activation = desc->activation; c.H = L.H; c.M = M;
    long long pw = 0;
    for (int j = 0; j <= L.H; ++j) {
      c.W[j] = params + desc->off_W[j]; c.bias[j] = params + desc->off_b[j];
      c.gamma[j] = params + desc->off_gamma[j]; c.beta[j] = params + desc->off_beta[j];
      c.K[j] = L.K[j]; c.N[j] = L.N[j]; c.ldn[j] = L.ldn[j];
      c.stat_out[j] = ws + L.stat[j];
      if (j < L.H) {
        c.act_out[j] = ws + L.act[j]; c.xhat_out[j] = ws + L.xhat[j];
        pw += (long long)L.K[j] * L.N[j];
      }
    }
    c.scores = scores;
    c.PS = chain_ps;
    c.buf_floats = chain_buf;
#ifdef ULTR_STAMPS
    { const char* e = getenv("ULTR_CHAIN_DBG"); c.dbg = e ? atoi(e) : 0; }
#endif
    static PerDeviceOnce attr_done;
    if (const int rc_ = raise_lds_limit(attr_done, reinterpret_cast<const void*>(fwd_chain_kernel), 160 * 1024, "ultr_dnn_forward")) return rc_;
    {
      ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("fwd_chain_kernel/P%lld", pw) : "fwd_chain_kernel", st);
      hipLaunchKernelGGL(fwd_chain_kernel, dim3((unsigned)cdiv(M, kChainRows)), dim3(kChainThreads), chain_smem, st, c);
    }
    return check_launch("fwd_chain_kernel");
  }
  for (int j = 0; j < L.H; ++j) {
    FwdArgs a{};
    if (j == 0) {
      a.x = table; a.ldx = ldx; a.docids = docids; a.n_rows = n_rows; a.row_stats = table_stats;
    } else {
      a.x = ws + L.act[j - 1]; a.ldx = L.ldn[j - 1]; a.docids = nullptr; a.n_rows = 0;
      a.stat_out = ws + L.stat[j];
    }
    a.has_ln = has_ln; a.eps = desc->ln_eps;
    a.gamma = params + desc->off_gamma[j]; a.beta = params + desc->off_beta[j];
    a.W = params + desc->off_W[j]; a.bias = params + desc->off_b[j];
    a.K = L.K[j]; a.N = L.N[j]; a.ldn = L.ldn[j]; a.M = M; a.activation = desc->activation;
    a.act_out = ws + L.act[j];
    a.xhat_out = ws + L.xhat[j]; a.ldxh = round4(L.K[j]);
    if (bn) {  // the normalised input comes from its own passes; the GEMM kernel runs as a plain layer on it
      rc = bn_forward_input(a.x, a.ldx, a.docids, a.n_rows, L.K[j], round4(L.K[j]), M, desc->ln_eps, bn_train, a.gamma,
                            a.beta, ws + L.cstat[j], ws + L.bn_part, ws + L.xhat[j], ws + L.bn_a, st);
      if (rc) return rc;
      a.x = ws + L.bn_a; a.ldx = round4(L.K[j]); a.docids = nullptr; a.n_rows = 0; a.row_stats = nullptr;
      a.stat_out = nullptr; a.has_ln = 0; a.xhat_out = nullptr;
    }
    const bool fuse = !bn && j == L.H - 1 && j > 0 && L.N[j] <= 128;  // the last hidden layer also produces the scores
    if (fuse) {
      a.out_gamma = params + desc->off_gamma[L.H]; a.out_beta = params + desc->off_beta[L.H];
      a.out_W = params + desc->off_W[L.H]; a.out_bias = params + desc->off_b[L.H];
      a.out_stat = ws + L.stat[L.H]; a.scores = scores;
    }
    rc = launch_fwd_layer(a, fuse, st);
    if (rc) return rc;
  }
  if (!bn && L.H > 1 && L.N[L.H - 1] <= 128) return ULTR_OK;  // scores came out of the fused epilogue
  OutArgs o{};
  const int j = L.H;
  if (j == 0) {
    o.x = table; o.ldx = ldx; o.docids = docids; o.n_rows = n_rows; o.row_stats = table_stats;
  } else {
    o.x = ws + L.act[j - 1]; o.ldx = L.ldn[j - 1]; o.stat_out = ws + L.stat[j];
  }
  o.has_ln = has_ln; o.eps = desc->ln_eps;
  o.gamma = params + desc->off_gamma[j]; o.beta = params + desc->off_beta[j];
  o.W = params + desc->off_W[j]; o.bias = params + desc->off_b[j];
  o.K = L.K[j]; o.M = M; o.scores = scores;
  if (bn) {
    rc = bn_forward_input(o.x, o.ldx, o.docids, o.n_rows, L.K[j], round4(L.K[j]), M, desc->ln_eps, bn_train, o.gamma,
                          o.beta, ws + L.cstat[j], ws + L.bn_part, ws + L.xhat_out, ws + L.bn_a, st);
    if (rc) return rc;
    o.x = ws + L.bn_a; o.ldx = round4(L.K[j]); o.docids = nullptr; o.n_rows = 0; o.row_stats = nullptr;
    o.stat_out = nullptr; o.has_ln = 0;
  }
  {
    ultr::ProfScope prof_("fwd_out_kernel", st);
    hipLaunchKernelGGL(fwd_out_kernel, dim3((unsigned)cdiv(M, 4)), dim3(kThreads), 0, st, o);
  }
  return check_launch("fwd_out_kernel");
}

static int dnn_backward_impl(const UltrDnnDesc* desc, const float* params, const float* table,
                             const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                             void* workspace, const float* d_scores, float* grads, float* sumsq, void* stream);

extern "C" int ultr_dnn_backward(const UltrDnnDesc* desc, const float* params, const float* table,
                                 const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                                 void* workspace, const float* d_scores, float* grads, void* stream) {
  return dnn_backward_impl(desc, params, table, table_stats, n_rows, ldx, docids, M, workspace, d_scores, grads, nullptr,
                           stream);
}

extern "C" int ultr_dnn_sumsq_count(const UltrDnnDesc* desc) {
  if (ultr_check_desc(desc, "ultr_dnn_sumsq_count") != ULTR_OK) return -1;
  const Layout L = make_layout(desc, 1);
  int n = (3 * L.K[L.H] + 1 + kFinOutCols - 1) / kFinOutCols;
  for (int j = 0; j < L.H; ++j) n += L.K[j];
  return n;
}

extern "C" int ultr_dnn_backward_sumsq(const UltrDnnDesc* desc, const float* params, const float* table,
                                       const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                                       void* workspace, const float* d_scores, float* grads, float* sumsq, void* stream) {
  ULTR_REQUIRE(sumsq, ULTR_EINVAL, "ultr_dnn_backward_sumsq: null sumsq");
  return dnn_backward_impl(desc, params, table, table_stats, n_rows, ldx, docids, M, workspace, d_scores, grads, sumsq,
                           stream);
}

static int dnn_backward_impl(const UltrDnnDesc* desc, const float* params, const float* table,
                             const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                             void* workspace, const float* d_scores, float* grads, float* sumsq, void* stream) {
  int rc = ultr_common_checks("ultr_dnn_backward", desc, params, table, table_stats, n_rows, ldx, docids, M, workspace);
  if (rc) return rc;
  ULTR_REQUIRE(d_scores && grads && ((uintptr_t)grads & 15) == 0, ULTR_EINVAL, "ultr_dnn_backward: bad d_scores/grads");
  const Layout L = make_layout(desc, M);
  float* ws = reinterpret_cast<float*>(workspace);
  const int has_ln = desc->norm == ULTR_NORM_LAYER;
  hipStream_t st = as_stream(stream);
  ULTR_REQUIRE((desc->n_params & 3) == 0, ULTR_EINVAL, "ultr_dnn_backward: n_params=%lld must be a multiple of 4", desc->n_params);
  const int H = L.H;
  int cur = 0;  // which dz buffer holds the gradient wrt the current layer's output
  const bool bn = L.bn;
  // where the gradient wrt layer j's output lives: its own buffer (wgall) or rotating slot `slot`
  auto dz_of = [&](int j, int slot) { return L.wgall ? ws + L.dzl[j] : ws + L.dz[slot]; };
  ULTR_REQUIRE(desc->norm != ULTR_NORM_BATCH_EVAL, ULTR_EINVAL,
               "ultr_dnn_backward: norm = batch (evaluation statistics) has no training pass");
  if (bn) {
    // Output layer under batch normalisation (bn_body.hpp): raw column sums of go and go * xhat give the layer's
    // parameter gradients (one partial row for finalize) and the two column means of the normalisation's backward
    // pass; dz of the last hidden layer follows elementwise.  No loss rides here: a parked one runs first.
    if (g_pending_loss.kind != ULTR_RIDE_NONE) {
      const PendingLoss pl = g_pending_loss;
      g_pending_loss.kind = ULTR_RIDE_NONE;
      rc = launch_loss_now(pl, st);
      if (rc) return rc;
    }
    if (hipMemsetAsync(grads, 0, sizeof(float) * (size_t)desc->n_params, st) != hipSuccess) {
      set_error("ultr_dnn_backward: hipMemsetAsync failed");
      return ULTR_EHIP;
    }
    const float* gamma = params + desc->off_gamma[H];
    const float* wout = params + desc->off_W[H];
    BnSumArgs s{};
    s.mode = 2; s.go = d_scores; s.xhat = ws + L.xhat_out; s.ldxh = round4(L.K[H]); s.K = L.K[H]; s.M = M;
    s.part = ws + L.bn_part;
    if ((rc = bn_launch_sum(s, st))) return rc;
    BnFinishArgs f{};
    f.mode = 3; f.part = ws + L.bn_part; f.K = L.K[H]; f.M = M; f.eps = desc->ln_eps; f.c12 = ws + L.c12;
    f.gamma = gamma; f.beta = params + desc->off_beta[H]; f.w = wout; f.outrow = ws + L.outpart;
    if ((rc = bn_launch_finish(f, st))) return rc;
    if (H > 0) {
      BnDxArgs x{nullptr, d_scores, wout, ws + L.xhat_out, ws + L.cstat[H], ws + L.c12, gamma, ws + L.act[H - 1],
                 desc->activation, L.K[H], L.ldn[H - 1], M, ws + L.dz[cur]};
      {
        ProfScope prof_("bn_dx_kernel", st);
        hipLaunchKernelGGL(bn_dx_kernel, dim3(bn_ew_blocks(M * L.ldn[H - 1])), dim3(256), 0, st, x);
      }
      if ((rc = check_launch("bn_dx_kernel"))) return rc;
    }
  } else {
    BwdOutArgs b{};
    OutArgs& o = b.f;
    if (H == 0) {
      o.x = table; o.ldx = ldx; o.docids = docids; o.n_rows = n_rows; o.row_stats = table_stats;
    } else {
      o.x = ws + L.act[H - 1]; o.ldx = L.ldn[H - 1];
      b.stat_in = ws + L.stat[H];
    }
    o.has_ln = has_ln; o.eps = desc->ln_eps;
    o.gamma = params + desc->off_gamma[H]; o.beta = params + desc->off_beta[H];
    o.W = params + desc->off_W[H]; o.bias = params + desc->off_b[H];
    o.K = L.K[H]; o.M = M;
    b.d_scores = d_scores; b.activation = desc->activation; b.has_prev = H > 0;
    b.dz = H > 0 ? dz_of(H - 1, cur) : ws + L.dz[cur]; b.lddz = H > 0 ? L.ldn[H - 1] : 0;
    b.rows_per_block = L.out_rows_per_block; b.partials = ws + L.outpart;
    b.zero = grads; b.zero_n4 = desc->n_params / 4;  // cleared by the first kernel of the pass, filled by the last
    const size_t smem = sizeof(float) * 4 * (3 * (size_t)L.K[H] + 1);
    ULTR_REQUIRE(smem <= 64 * 1024, ULTR_ELIMIT, "ultr_dnn_backward: output layer input width %d too large", L.K[H]);
    ULTR_REQUIRE(H == 0 || L.K[H] <= 1024, ULTR_ELIMIT, "ultr_dnn_backward: hidden width %d beyond %d", L.K[H], 1024);
    // A parked loss (ultr_loss_{dla,ipw,rem}_deferred) rides with this launch when it describes this very batch
    int ride = ULTR_RIDE_NONE;
    if (g_pending_loss.kind != ULTR_RIDE_NONE) {
      const PendingLoss pl = g_pending_loss;
      g_pending_loss.kind = ULTR_RIDE_NONE;
      if (L.K[H] <= 1024 && pl.d_scores() == d_scores && pl.rows() == M) {
        ride = pl.kind;
        b.ride = ride; b.dla = pl.dla; b.ipw = pl.ipw; b.rem = pl.rem; b.pd = pl.pd;
        b.dla_blocks = (int)pl.nblocks; b.dla_pad = (b.dla_blocks + 7) & ~7;
      } else {
        rc = launch_loss_now(pl, st);
        if (rc) return rc;
      }
    }
    {
      ultr::ProfScope prof_(ride == ULTR_RIDE_DLA ? "bwd_out_kernel+dla_loss" : ride == ULTR_RIDE_IPW ? "bwd_out_kernel+ipw_loss"
                            : ride == ULTR_RIDE_REM ? "bwd_out_kernel+rem_loss" : ride == ULTR_RIDE_PD ? "bwd_out_kernel+pd_loss"
                            : "bwd_out_kernel", st);
      const dim3 grid((unsigned)(L.out_blocks + b.dla_pad)), block(kThreads);
#define ULTR_BWD_OUT(KQ_)                                                                                      \
  do {                                                                                                         \
    if (ride == ULTR_RIDE_DLA) hipLaunchKernelGGL((bwd_out_kernel<KQ_, ULTR_RIDE_DLA>), grid, block, smem, st, b);      \
    else if (ride == ULTR_RIDE_IPW) hipLaunchKernelGGL((bwd_out_kernel<KQ_, ULTR_RIDE_IPW>), grid, block, smem, st, b); \
    else if (ride == ULTR_RIDE_REM) hipLaunchKernelGGL((bwd_out_kernel<KQ_, ULTR_RIDE_REM>), grid, block, smem, st, b); \
    else if (ride == ULTR_RIDE_PD) hipLaunchKernelGGL((bwd_out_kernel<KQ_, ULTR_RIDE_PD>), grid, block, smem, st, b);   \
    else hipLaunchKernelGGL((bwd_out_kernel<KQ_, ULTR_RIDE_NONE>), grid, block, smem, st, b);                            \
  } while (0)
      if (L.K[H] <= 128) ULTR_BWD_OUT(2);
      else if (L.K[H] <= 256) ULTR_BWD_OUT(4);
      else if (L.K[H] <= 1024) ULTR_BWD_OUT(16);
      else hipLaunchKernelGGL(bwd_out_wide_kernel, grid, block, smem, st, b);
#undef ULTR_BWD_OUT
    }
    rc = check_launch("bwd_out_kernel");
    if (rc) return rc;
  }
  // Per layer: wgrad_j, dgrad_j, then the row pass that turns dh into dz_{j-1}.  The two GEMMs of a layer can run
  // side by side in one launch (bwd_pair_kernel).  For the big layer the pair costs the SUM of its parts -- both
  // are bound by the memory side feeding just-written data, not by latency -- so only small layers are paired.
  // Measured at M = 2560: pairing layer 2 (K256 N128) 13.9 us against 9.4 + 8.8, pairing layer 1 (K512 N256) 34.0 us
  // against 14.9 + 17.5
  int pair_mask = 0;
  for (int j = 1; j < H; ++j)  // small GEMMs only: they leave most of the chip idle on their own
    if (2.0 * (double)M * L.K[j] * L.N[j] <= 0.25e9) pair_mask |= 1 << (j - 1);
  if (bn) pair_mask = 0;  // the normalisation's column sums sit between dgrad and the row pass
  if (L.wgall) pair_mask = 0;  // the weight gradients come after the dgrad chain, in one launch
  // A parked metric request (ultr_metrics_deferred) rides with layer 0's weight-gradient launch.  Measured at the
  // headline shape (wgrad alone -> with the rider): layer 0 (K136 N512) 12.4 -> 15.3 us, layer 2 (K256 N128) 9.3 -> 15.0,
  // layer 1 (K512 N256) 14.7 -> 24.1 against 14.4 us for a metric launch of its own: the heavier the carrier's operand
  // streams, the longer the rider's dependent loads queue behind them.
  // ULTR_BWD_CHAIN=1: every dgrad GEMM and row pass in ONE launch (bwd_chain_kernel), the weight gradients follow.
  // Opt-in: measured at M = 2560 the chain takes 37.9 us against 45 us for the launches it replaces, but the step
  // as a whole replays 8 us SLOWER (156 against 148 us) -- 160 workgroups of 16 rows stream every weight through
  // every CU and leave 96 CUs idle, and wgrad_2 loses its place beside dgrad_2.
  static const bool bchain_env = [] { const char* e = getenv("ULTR_BWD_CHAIN"); return e && e[0] == '1'; }();
  bool use_chain = bchain_env && H >= 2 && H <= 3 && !bn && !L.wgall;
  if (use_chain) {
    BwdChainArgs c{};
    c.has_ln = has_ln; c.activation = desc->activation; c.H = H; c.M = M;
    int maxw = L.N[H - 1];
    for (int j = 0; j < H; ++j) {
      c.W[j] = params + desc->off_W[j]; c.gamma[j] = params + desc->off_gamma[j];
      c.act[j] = ws + L.act[j]; c.stat[j] = ws + L.stat[j];
      c.dz[j] = ws + L.dz[(H - 1 - j) % 3];
      c.K[j] = L.K[j]; c.N[j] = L.N[j]; c.ldn[j] = L.ldn[j];
      if (j >= 1 && L.K[j] > maxw) maxw = L.K[j];
    }
    c.PS = ((maxw + 31) & ~31) + 4;
    c.buf_floats = kChainRows * c.PS > kChainExch ? kChainRows * c.PS : kChainExch;
    const size_t smem = 2 * (size_t)c.buf_floats * sizeof(float);
    if (smem > 160 * 1024) {
      use_chain = false;
    } else {
      static PerDeviceOnce attr_done;
      if (const int rc_ = raise_lds_limit(attr_done, reinterpret_cast<const void*>(bwd_chain_kernel), 160 * 1024, "ultr_dnn_backward")) return rc_;
      {
        ultr::ProfScope prof_("bwd_chain_kernel", st);
        hipLaunchKernelGGL(bwd_chain_kernel, dim3((unsigned)cdiv(M, kChainRows)), dim3(kChainThreads), smem, st, c);
      }
      rc = check_launch("bwd_chain_kernel");
      if (rc) return rc;
    }
  }
  int carrier = -1;
  if (g_pending_metrics.on) {
    if (H >= 1 && g_pending_metrics.nblocks <= 64 && !L.wgall) carrier = 0;  // a bigger request is no rider: a launch of its own
    if (carrier < 0) {
      g_pending_metrics.on = false;
      rc = launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, st);
      if (rc) return rc;
    }
  }
  WgradArgs wg_all[ULTR_MAX_LAYERS];
  for (int j = H - 1; j >= 0; --j) {
    WgradArgs g{};
    g.xhat = ws + L.xhat[j]; g.ldx = round4(L.K[j]);
    g.dz = dz_of(j, cur); g.ldn = L.ldn[j]; g.N = L.N[j];
    g.K = L.K[j]; g.M = M; g.rows_per_split = L.rows_per_split[j];
    g.gslab = ws + L.gslab[j]; g.sslab = ws + L.sslab[j];
    const dim3 ggrid((unsigned)L.nsplit[j], (unsigned)cdiv(L.K[j], 64), (unsigned)cdiv(L.ldn[j], 64));
    const bool pair = !use_chain && j > 0 && j != carrier && ((pair_mask >> (j - 1)) & 1);
    if (L.wgall) {
      wg_all[j] = g;  // launched together after the dgrad chain
    } else if (j == carrier) {
      WgradRiderArgs wr{};
      wr.g = g; wr.g_gx = (int)ggrid.x; wr.g_gy = (int)ggrid.y; wr.g_gz = (int)ggrid.z;
      wr.m = g_pending_metrics.a; wr.m_blocks = (int)g_pending_metrics.nblocks; wr.m_pad = (wr.m_blocks + 7) & ~7;
      g_pending_metrics.on = false;
      const long long nblk = (long long)ggrid.x * ggrid.y * ggrid.z + wr.m_pad;
      ULTR_REQUIRE(nblk < 2147483647ll, ULTR_ELIMIT, "ultr_dnn_backward: grid too large");
      {
        ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("wgrad_rider_kernel/K%dN%d", L.K[j], L.N[j]) : "wgrad_rider_kernel", st);
        hipLaunchKernelGGL(wgrad_rider_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, st, wr);
      }
      rc = check_launch("wgrad_rider_kernel");
      if (rc) return rc;
    } else if (j == 0 || !pair) {
      {
        ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("wgrad_kernel/K%dN%d", L.K[j], L.N[j]) : "wgrad_kernel", st);
        hipLaunchKernelGGL(wgrad_kernel, ggrid, dim3(kThreads), 0, st, g);
      }
      rc = check_launch("wgrad_kernel");
      if (rc) return rc;
    }
    if (j > 0 && use_chain) {
      cur = (cur + 1) % 3;  // dz_{j-1} is where the chain kernel left it
    } else if (j > 0) {
      const int nxt = (cur + 1) % 3;
      DgradArgs d{};
      d.dz = dz_of(j, cur); d.ldn = L.ldn[j]; d.N = L.N[j];
      d.W = params + desc->off_W[j];
      d.K = L.K[j]; d.ldk = L.ldn[j - 1]; d.M = M;
      d.dh = dz_of(j - 1, nxt);
      const dim3 dgrid((unsigned)cdiv(M, 32), (unsigned)cdiv(L.ldn[j - 1], 128));
      if (pair) {
        BwdPairArgs bp{g, d, (int)dgrid.x, (int)dgrid.y, (int)ggrid.x, (int)ggrid.y, (int)ggrid.z};
        const long long nblk = (long long)dgrid.x * dgrid.y + (long long)ggrid.x * ggrid.y * ggrid.z;
        ULTR_REQUIRE(nblk < 2147483647ll, ULTR_ELIMIT, "ultr_dnn_backward: grid too large");
        {
          ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("bwd_pair_kernel/K%dN%d", L.K[j], 2 * L.N[j]) : "bwd_pair_kernel", st);
          hipLaunchKernelGGL(bwd_pair_kernel, dim3((unsigned)nblk), dim3(kThreads), 0, st, bp);
        }
        rc = check_launch("bwd_pair_kernel");
      } else {
        {
          ultr::ProfScope prof_(ultr::g_profile_on ? ultr::profile_intern("dgrad_kernel/K%dN%d", L.K[j], L.N[j]) : "dgrad_kernel", st);
          hipLaunchKernelGGL(dgrad_kernel, dgrid, dim3(kThreads), 0, st, d);
        }
        rc = check_launch("dgrad_kernel");
      }
      if (rc) return rc;
      if (bn) {  // column means of gamma dh and gamma dh xhat, then dz_{j-1} elementwise, in place of the row pass
        const float* gamma = params + desc->off_gamma[j];
        BnSumArgs s{};
        s.mode = 2; s.dh = ws + L.dz[nxt]; s.lddh = L.ldn[j - 1]; s.xhat = ws + L.xhat[j]; s.ldxh = round4(L.K[j]);
        s.gamma = gamma; s.K = L.K[j]; s.M = M; s.part = ws + L.bn_part;
        if ((rc = bn_launch_sum(s, st))) return rc;
        BnFinishArgs bf{};
        bf.mode = 2; bf.part = ws + L.bn_part; bf.K = L.K[j]; bf.M = M; bf.c12 = ws + L.c12;
        if ((rc = bn_launch_finish(bf, st))) return rc;
        BnDxArgs x{ws + L.dz[nxt], nullptr, nullptr, ws + L.xhat[j], ws + L.cstat[j], ws + L.c12, gamma, ws + L.act[j - 1],
                   desc->activation, L.K[j], L.ldn[j - 1], M, ws + L.dz[nxt]};
        {
          ProfScope prof_("bn_dx_kernel", st);
          hipLaunchKernelGGL(bn_dx_kernel, dim3(bn_ew_blocks(M * L.ldn[j - 1])), dim3(256), 0, st, x);
        }
        if ((rc = check_launch("bn_dx_kernel"))) return rc;
        cur = nxt;
        continue;
      }
      RowpassArgs r{};
      r.dh = dz_of(j - 1, nxt); r.ldk = L.ldn[j - 1]; r.K = L.K[j]; r.M = M;
      r.a = ws + L.act[j - 1]; r.stat = ws + L.stat[j]; r.gamma = params + desc->off_gamma[j];
      r.has_ln = has_ln; r.activation = desc->activation;
      {
        ultr::ProfScope prof_("rowpass_kernel", st);
        const dim3 rgrid((unsigned)cdiv(M, 4)), rblock(kThreads);
        if (r.K <= 256) hipLaunchKernelGGL(rowpass_kernel<4>, rgrid, rblock, 0, st, r);
        else if (r.K <= 512) hipLaunchKernelGGL(rowpass_kernel<8>, rgrid, rblock, 0, st, r);
        else hipLaunchKernelGGL(rowpass_kernel<16>, rgrid, rblock, 0, st, r);
      }
      rc = check_launch("rowpass_kernel");
      if (rc) return rc;
      cur = nxt;
    }
  }
  if (L.wgall) {
    rc = launch_wgrad_all(H, wg_all, L.nsplit[0], false, st);
    if (rc) return rc;
  }
  FinalizeArgs f{};
  f.H = H;
  int start = 0;
  for (int j = 0; j < H; ++j) {
    f.row_start[j] = start;
    start += L.K[j];
    f.K[j] = L.K[j]; f.N[j] = L.N[j]; f.ldn[j] = L.ldn[j]; f.nsplit[j] = L.nsplit[j];
    f.gslab[j] = ws + L.gslab[j]; f.sslab[j] = ws + L.sslab[j];
  }
  f.row_start[H] = start;
  for (int j = 0; j <= H; ++j) {
    f.off_W[j] = desc->off_W[j]; f.off_b[j] = desc->off_b[j];
    f.off_gamma[j] = desc->off_gamma[j]; f.off_beta[j] = desc->off_beta[j];
  }
  // batch normalisation: xhat is column-normalised, the rest of the algebra (dW = gamma G + beta S, dgamma, dbeta) is
  // LayerNorm's; the output layer left ONE partial row
  f.params = params; f.grads = grads; f.has_ln = has_ln || bn;
  f.outpart = ws + L.outpart; f.out_blocks = bn ? 1 : L.out_blocks; f.Kout = L.K[H];
  f.sumsq = sumsq;
  const unsigned fblocks = (unsigned)(start + (3 * L.K[H] + 1 + kFinOutCols - 1) / kFinOutCols);
  if (g_pending_assemble.on && g_pending_assemble.nblocks <= 4096) {
    FinalizeRiderArgs fr{};
    fr.f = f; fr.a = g_pending_assemble.a;
    fr.a_blocks = (int)g_pending_assemble.nblocks; fr.a_pad = (fr.a_blocks + 7) & ~7;
    g_pending_assemble.on = false;
    {
      ultr::ProfScope prof_("finalize_rider_kernel", st);
      hipLaunchKernelGGL(finalize_rider_kernel, dim3(fblocks + (unsigned)fr.a_pad), dim3(kThreads), 0, st, fr);
    }
    return check_launch("finalize_rider_kernel");
  }
  {
    ultr::ProfScope prof_("finalize_kernel", st);
    hipLaunchKernelGGL(finalize_kernel, dim3(fblocks), dim3(kThreads), 0, st, f);
  }
  return check_launch("finalize_kernel");
}
