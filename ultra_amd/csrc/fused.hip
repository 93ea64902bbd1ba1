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
#include "optim_shared.hpp"

namespace ultr {

#ifndef ULTR_TC_STAMP_LAYER
#define ULTR_TC_STAMP_LAYER 2
#endif
constexpr int kTcThreads = 512, kTcRows = 16, kTcExch = 7 * 1024;
constexpr int kTcMiscFloats = (ULTR_MAX_LAYERS + 1) * 32 + 32;  // row statistics [layer][16][2], scores [16], pad

#ifdef ULTR_STAMPS
__device__ long long g_tc_stamps[1024 * 32];
__device__ long long g_tc_wstamps[1024 * 8 * 16];
#define TC_WSTAMP(i)                                                                                                  \
  do {                                                                                                                \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) g_tc_wstamps[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64(); \
  } while (0)
__device__ long long g_wg_stamps[1024 * 8];
#define WG_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_wg_stamps[blockIdx.x * 8 + (i)] = wall_clock64();         \
  } while (0)
#define TC_STAMP(i)                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_tc_stamps[blockIdx.x * 32 + (i)] = wall_clock64();       \
  } while (0)
#else
#define TC_STAMP(i) do { } while (0)
#define TC_WSTAMP(i) do { } while (0)
#define WG_STAMP(i) do { } while (0)
#endif

struct TcArgs {
  const float* table; int ldx; const int* docids; int n_rows; const float* row_stats;
  int has_ln; float eps; int activation; int H;
  int B, L, nl;                                   // nl lists of L ranks per workgroup, nl * L <= 16
  int list_base, part_base;                       // workgroup i owns lists list_base + i * nl ..., partial row part_base + i
  const float* W[ULTR_MAX_LAYERS]; const float* WT[ULTR_MAX_LAYERS];
  // bf16 instantiations stream these instead: W_j resp. W_j^T packed [rows / 4][ld][4] bf16 (pack_weights_kernel)
  const float* Wp[ULTR_MAX_LAYERS]; const float* WTp[ULTR_MAX_LAYERS];
  const float* bias[ULTR_MAX_LAYERS]; const float* gamma[ULTR_MAX_LAYERS]; const float* beta[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  float* xhat[ULTR_MAX_LAYERS];                   // [M][round4(K_j)]  normalised input rows of layer j, list-major rows
  float* dz[ULTR_MAX_LAYERS];                     // [M][ldn_j]        gradient wrt layer j's pre-activation
  float* scores;                                  // [L][B] position-major
  float* list_weights;                            // [B] or NULL
  float* outpart;                                 // [workgroups of the step][3 K_H + 4] output-layer gradient partials
  int PS;                                         // row stride of the two LDS panels (== 4 mod 32)
  int PS2;                                        // 32-row kernel: row stride of the second panel (hidden widths only)
  int panel_floats;                               // floats per panel
  int act_off[ULTR_MAX_LAYERS], act_ps[ULTR_MAX_LAYERS];  // LDS: activations of hidden layer j, [16][act_ps]
  int p1_off, p2_off, misc_off;
  int dbg;
  DlaArgs dla; IpwArgs ipw; RemArgs rem; PdArgs pd;       // the one matching the kernel's RIDE is live
};

#define ULTR_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

struct TcFrag { float4 b[4]; };

// The weight ring: kTcStages register stages per wave, kTcStages - 1 chunks of weights in flight while one is multiplied.
// Measured: six stages (five 4 KB chunks per wave = 160 KB per CU in flight) run the kernel in 58.2 us against 57.2 with
// four -- the stream is not short of loads in flight -- so four it is (-DULTR_TC_STAGES=6 builds the other).
#ifndef ULTR_TC_STAGES
#define ULTR_TC_STAGES 4
#endif
constexpr int kTcStages = ULTR_TC_STAGES;
struct TcRing { TcFrag s[kTcStages]; };

// Geometry of one 16-row GEMM  out[16][N] = A[16][K] . Wm[K][ld]  over the 8 waves of the workgroup: a wave owns units
// of 64 output columns; fewer than 8 units split the contraction over the idle waves (KS slices, meeting in LDS).
typedef unsigned int tc_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kTcBufFlags = 0x00020000;  // raw buffer, 32-bit data format (word 3 of a gfx9 buffer descriptor)
struct TcGeo {
  const float* W; int K, N, ld, U, Up, KS, ks, u_first, c_begin, c_last, nch;
  // The weight stream goes through buffer loads: descriptor in scalar registers, ONE 32-bit vector add per chunk for
  // the address (a flat load wants a 64-bit vector address per row: measured, the fp32 MFMA runs on the vector ALU, so
  // every vector instruction in the loop comes straight out of the multiply time), and the descriptor's size ends at
  // the wave's last chunk (or at row K): rows beyond the matrix and the loop's run-ahead loads past the end of the
  // stream are out of range -- they return zeros and move nothing.
  __amdgpu_buffer_rsrc_t rs;
  unsigned chunk_bytes, row_bytes;
  int dbg;  // diagnostic builds (-DULTR_STAMPS): 1 no weight loads
};
// bf: the matrix is the packed bf16 copy (below)
__device__ __forceinline__ TcGeo tc_geo(const float* W, int K, int N, int ld, int w, bool bf) {
  TcGeo g;
  g.W = W; g.K = K; g.N = N; g.ld = ld; g.dbg = 0;
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
  // fp32: rows of ld floats; packed bf16: quad-rows of ld * 4 values of 2 bytes, zero rows up to a multiple of 16
  g.row_bytes = (unsigned)ld * 4u;
  g.chunk_bytes = bf ? 32u * (unsigned)ld : 64u * (unsigned)ld;
  const unsigned bytes = bf ? (unsigned)c_end * g.chunk_bytes : (unsigned)min(16 * c_end, K) * g.row_bytes;
  g.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)bytes, kTcBufFlags);
  return g;
}
// Chunk i of the wave's weight stream: rows 16 c + 4 kk + s (s = 0..3) of Wm, four consecutive columns from colc.
// BF: the matrix is the packed bf16 copy -- element (k, n) at ((k / 4) * ld + n) * 4 + k % 4 (in bf16), rows padded with
// zeros to a multiple of 16 -- so a lane's 4 x 4 block of chunk i (rows 16 c + 4 kk .. + 3, columns colc .. colc + 3) is
// 32 contiguous bytes, half of what the fp32 stream moves.
typedef short tc_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tc_s16x4 tc_bf16x4(float a, float b, float c, float d) {
  const uint2 u = make_uint2(pk_bf16(a, b), pk_bf16(c, d));
  return __builtin_bit_cast(tc_s16x4, u);
}
__device__ __forceinline__ tc_s16x4 tc_as_bf16x4(float lo, float hi) {
  const float2 u = make_float2(lo, hi);
  return __builtin_bit_cast(tc_s16x4, u);
}
// voff: the lane's byte offset inside a chunk (tc_voff).  Loads are UNCONDITIONAL: a branch around them would leave the
// number of loads in flight unknown at the loop header, and the compiler would then drain every outstanding load there
// (s_waitcnt vmcnt(0)) instead of waiting for the oldest chunk only; past the end of the stream they are out of range.
template <bool FAST, bool BF>
__device__ __forceinline__ void tc_wload(const TcGeo& g, unsigned voff, int i, TcFrag& S) {
#ifdef ULTR_STAMPS
  if (!BF && (g.dbg & 1)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) S.b[s] = make_float4(0.f, 1.f, 2.f, 3.f);
    return;
  }
  if (g.dbg & 2) i = 0;  // every chunk re-reads the wave's first one: L1 hits, no L2 traffic (timing only)
#endif
  const unsigned vo = voff + (unsigned)(g.c_begin + i) * g.chunk_bytes;
  if (BF) {
    S.b[0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(g.rs, vo, 0, 0));
    S.b[1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(g.rs, vo, 16, 0));
    return;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s)
    S.b[s] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(g.rs, vo, (int)(s * g.row_bytes), 0));
}
// per-lane byte offset of the lane's block inside a chunk
template <bool BF> __device__ __forceinline__ unsigned tc_voff(const TcGeo& g, int colc, int kk) {
  return 4u * (BF ? (unsigned)(2 * (kk * g.ld + colc)) : (unsigned)(4 * kk * g.ld + colc));
}
// ALWAYS: waves without a unit of their own request (clamped, in-bounds) chunks as well, so that the number of loads
// this call puts in flight is the same on every path and a later wait for OLDER loads can be a counted one
template <bool FAST, bool BF, bool ALWAYS = false>
__device__ __forceinline__ void tc_prefetch(const TcGeo& g, int nn, int kk, TcRing& R) {
  if (ALWAYS || g.u_first < g.U) {
    const int colc = min(64 * g.u_first + 4 * nn, g.ld - 4);
    const unsigned voff = tc_voff<BF>(g, colc, kk);
#pragma unroll
    for (int i = 0; i < kTcStages - 1; ++i) tc_wload<FAST, BF>(g, voff, i, R.s[i]);
  }
}

#define TC_SB() __builtin_amdgcn_sched_barrier(0)
#if ULTR_TC_STAGES == 4
#define TC_STREAM(NCH, LOAD, COMPUTE)                                        \
  do {                                                                       \
    int i_ = 0;                                                              \
    for (; i_ + 4 <= (NCH); i_ += 4) {                                       \
      LOAD(i_ + 3, R.s[3]); TC_SB(); COMPUTE(R.s[0], i_); TC_SB();           \
      LOAD(i_ + 4, R.s[0]); TC_SB(); COMPUTE(R.s[1], i_ + 1); TC_SB();       \
      LOAD(i_ + 5, R.s[1]); TC_SB(); COMPUTE(R.s[2], i_ + 2); TC_SB();       \
      LOAD(i_ + 6, R.s[2]); TC_SB(); COMPUTE(R.s[3], i_ + 3); TC_SB();       \
    }                                                                        \
    if (i_ < (NCH)) COMPUTE(R.s[0], i_);                                     \
    if (i_ + 1 < (NCH)) COMPUTE(R.s[1], i_ + 1);                             \
    if (i_ + 2 < (NCH)) COMPUTE(R.s[2], i_ + 2);                             \
  } while (0)
#else
#define TC_STREAM(NCH, LOAD, COMPUTE)                                        \
  do {                                                                       \
    int i_ = 0;                                                              \
    for (; i_ + 6 <= (NCH); i_ += 6) {                                       \
      LOAD(i_ + 5, R.s[5]); TC_SB(); COMPUTE(R.s[0], i_); TC_SB();           \
      LOAD(i_ + 6, R.s[0]); TC_SB(); COMPUTE(R.s[1], i_ + 1); TC_SB();       \
      LOAD(i_ + 7, R.s[1]); TC_SB(); COMPUTE(R.s[2], i_ + 2); TC_SB();       \
      LOAD(i_ + 8, R.s[2]); TC_SB(); COMPUTE(R.s[3], i_ + 3); TC_SB();       \
      LOAD(i_ + 9, R.s[3]); TC_SB(); COMPUTE(R.s[4], i_ + 4); TC_SB();       \
      LOAD(i_ + 10, R.s[4]); TC_SB(); COMPUTE(R.s[5], i_ + 5); TC_SB();      \
    }                                                                        \
    if (i_ < (NCH)) COMPUTE(R.s[0], i_);                                     \
    if (i_ + 1 < (NCH)) COMPUTE(R.s[1], i_ + 1);                             \
    if (i_ + 2 < (NCH)) COMPUTE(R.s[2], i_ + 2);                             \
    if (i_ + 3 < (NCH)) COMPUTE(R.s[3], i_ + 3);                             \
    if (i_ + 4 < (NCH)) COMPUTE(R.s[4], i_ + 4);                             \
  } while (0)
#endif

// The GEMM proper.  A: LDS panel [16][PS] (finite beyond K up to the next multiple of 16 in rows that matter: zero); once
// multiplied out it serves as the split-K exchange buffer.  epi(u, acc) receives the finished unit:
// acc[t][r] = out[row 4 kk + r][col 64 u + 4 nn + t].  The first three weight chunks of the wave's first unit must
// already be in the ring (tc_prefetch).
// vmcnt(0) through the builtin, which the compiler's wait-count pass models (an asm statement it would not): afterwards
// it knows that every earlier load has landed.  Used where that is already true at run time -- at the end of a weight
// stream, whose last chunk is younger than everything the following stage reads (gamma / beta / bias requested before
// the GEMM) -- so that the stage's first use of those values does not get a conservative vmcnt(0) of its own, which would
// ALSO wait for the next layer's prefetch issued in between (the counter is in-order): measured, that wait was most of
// every stage (1 us each).
__device__ __forceinline__ void tc_drain_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// bias: the layer's bias vector (forward epilogues; requested with the unit's first chunks) or nullptr
template <bool FAST, bool BF, class Epi>
__device__ __forceinline__ void tc_gemm(const TcGeo& g, float* A, int PS, int nn, int kk, int lane, TcRing& R, const float* bias,
                                        Epi epi, int stamp = -1) {
  if (stamp >= 0) TC_STAMP(stamp);
  const float* arow = A + nn * PS + 4 * kk + 16 * g.c_begin;
  for (int u = g.u_first; u < g.U; u += 8) {
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int colc = min(64 * u + 4 * nn, g.ld - 4);
    const unsigned voff = tc_voff<BF>(g, colc, kk);
    if (u != g.u_first) {
#pragma unroll
      for (int i = 0; i < kTcStages - 1; ++i) tc_wload<FAST, BF>(g, voff, i, R.s[i]);
    }
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + min(64 * u + 4 * nn, ((g.N + 3) & ~3) - 4));
    float4 a_cur = *reinterpret_cast<const float4*>(arow);
    auto compute = [&](TcFrag& S, int i) {
      // the panel values of the NEXT chunk are requested before this chunk's MFMAs: their LDS latency runs under them
      const float4 a_nxt = *reinterpret_cast<const float4*>(arow + 16 * min(i + 1, g.nch - 1));
      if (BF) {  // one bf16 MFMA (16 x 16 x 16, fp32 accumulate) per column group takes the place of four fp32 ones
        const tc_s16x4 av = tc_bf16x4(a_cur.x, a_cur.y, a_cur.z, a_cur.w);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, tc_as_bf16x4(S.b[0].x, S.b[0].y), acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, tc_as_bf16x4(S.b[0].z, S.b[0].w), acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, tc_as_bf16x4(S.b[1].x, S.b[1].y), acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, tc_as_bf16x4(S.b[1].z, S.b[1].w), acc[3], 0, 0, 0);
        a_cur = a_nxt;
        return;
      }
      const float a[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#ifdef ULTR_STAMPS
      if (g.dbg & 4) {  // no MFMA: the stream alone (timing only)
#pragma unroll
        for (int s = 0; s < 4; ++s) { acc[0][0] += S.b[s].x * a[s]; acc[1][0] += S.b[s].y; acc[2][0] += S.b[s].z; acc[3][0] += S.b[s].w; }
        a_cur = a_nxt;
        return;
      }
#endif
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], S.b[s].w, acc[3], 0, 0, 0);
      }
      a_cur = a_nxt;
    };
#define TC_LOAD(i, S) tc_wload<FAST, BF>(g, voff, (i), S)
#define TC_COMPUTE(S, i) compute(S, (i))
    TC_STREAM(g.nch, TC_LOAD, TC_COMPUTE);
#undef TC_LOAD
#undef TC_COMPUTE
    tc_drain_loads();
    if (stamp >= 0) TC_STAMP(stamp + 1);
    if (g.KS > 1) {  // uniform over the workgroup (U < 8: exactly one unit per wave)
      lds_barrier();  // the panel has been multiplied out: the exchange buffer takes its place
      if (stamp >= 0) TC_STAMP(stamp + 2);
      if (g.ks > 0) {
        float* dst = A + ((g.ks - 1) * g.Up + u) * 1024 + lane * 4;
#pragma unroll
        for (int t = 0; t < 4; ++t)
          *reinterpret_cast<float4*>(dst + 256 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      }
      lds_barrier();
      if (stamp >= 0) TC_STAMP(stamp + 3);
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
    if (stamp >= 0) TC_STAMP(stamp + 4);
    if (g.ks == 0) epi(u, acc, b4);
  }
  if (g.u_first >= g.U && g.KS > 1) {  // idle waves of a non-power-of-two unit count still meet the barriers
    lds_barrier();
    lds_barrier();
  }
  tc_drain_loads();  // on every path out of the GEMM (a wave without a unit has nothing in flight either)
}

// FAST kernels are compiled for the reference's default scorer (DNN.py:25-34: elu, layer normalisation) with every width a
// multiple of 4: activation, LayerNorm switch and the per-element tail masks fold away.  The stages between the GEMMs
// are bound by the NUMBER of instructions they issue (two waves per SIMD, ~2.6 cycles per instruction measured), so this
// is worth 3-5x on them.  Everything else runs the generic instantiation.
template <bool FAST> __device__ __forceinline__ float tc_actf(int kind, float z) {
  if (FAST) return z > 0.f ? z : expm1_neg(z);
  return act_fwd(kind, z);
}
template <bool FAST> __device__ __forceinline__ float tc_actg(int kind, float a) {
  if (FAST) return a > 0.f ? 1.f : a + 1.f;
  return act_grad_from_out(kind, a);
}
// 1 / sqrt(v): the hardware reciprocal square root (1 ulp) in the fast kernels
template <bool FAST> __device__ __forceinline__ float tc_rsqrt(float v) {
  if (FAST) return __builtin_amdgcn_rsqf(v);
  return 1.f / sqrtf(v);
}
template <bool FAST> __device__ __forceinline__ bool tc_ok(int k, int e, int K) { return FAST ? k < K : k + e < K; }
__device__ __forceinline__ void tc_unpack(const float4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }

// The stages between the GEMMs are chains of dependent instructions (LDS read -> two wave reductions -> normalise ->
// store): a wave that owns TWO valid rows (w and w + 8; with lists of 10 that is waves 0 and 1) walks both chains in
// lockstep, step by step, so that one row's latencies hide behind the other's -- the workgroup waits for its slowest
// wave at the next barrier, and a wave that did its rows one after the other took twice as long as the rest.  Each
// row's arithmetic is what it was (same operations, same order).
template <int NR> struct TcRows { static constexpr int value = NR; };
template <int NR> __device__ __forceinline__ void wave_sum_rows(float (&v)[NR]) {
#define ULTR_TC_STEP(ctrl) { float t_[NR]; _Pragma("unroll") for (int r = 0; r < NR; ++r) t_[r] = ULTR_DPP_F(v[r], ctrl); \
                             _Pragma("unroll") for (int r = 0; r < NR; ++r) v[r] += t_[r]; }
  ULTR_TC_STEP(0xB1) ULTR_TC_STEP(0x4E) ULTR_TC_STEP(0x141) ULTR_TC_STEP(0x140)
#undef ULTR_TC_STEP
  {
    float a[NR], o[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = xor16_partner(v[r], &o[r]);
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = a[r] + o[r];
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = xor32_partner(v[r], &o[r]);
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = a[r] + o[r];
  }
}

// NQ: float4 per lane and row the stages are compiled for (256 NQ >= the widest layer input, rounded up to 16)
// BF: the hidden layers' GEMMs (forward and dgrad) multiply bf16-rounded activations with the packed bf16 weight copies,
// accumulating in fp32; everything between the GEMMs (LayerNorm, activations, loss, the output layer) stays fp32.
template <int RIDE, bool FAST, int NQ, bool BF>
__global__ void __launch_bounds__(kTcThreads) train_chain_kernel(TcArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, kk = lane >> 4;
  const int L = p.L, B = p.B, nl = p.nl, H = p.H, PS = p.PS;
  const bool has_ln = FAST ? true : p.has_ln != 0;
  const int act = p.activation;
  const long long list0 = p.list_base + (long long)blockIdx.x * nl;
  float* P1 = smem + p.p1_off;
  float* P2 = smem + p.p2_off;
  float* s_stat = smem + p.misc_off;                       // [layer][16][2]
  float* s_score = s_stat + (ULTR_MAX_LAYERS + 1) * 32;    // [16]
  TcRing R;
  TC_STAMP(0);
#ifdef ULTR_TC_PRIO
  // the two waves of a SIMD (w and w + 4) run the same program from the same barrier: left alone they reach their MFMA
  // blocks and their load sections together, and the matrix pipe idles during the latter.  A static priority for the
  // second half makes one of the pair win every arbitration, which staggers them
  if (w >= 4) __builtin_amdgcn_s_setprio(ULTR_TC_PRIO);
#endif

  // ---- the wave's two rows, w and w + 8 (valid rows spread evenly over the SIMDs): local row rr = li * L + l  <->
  // list list0 + li, rank l.  Rows beyond the workgroup's lists are skipped by every stage; their LDS rows only have
  // to hold finite numbers (the MFMA tiles carry them along, nothing reads their results).
  bool rvalid[2];
  long long re[2], rm[2];  // position-major element index (scores, clicks, doc ids), list-major row (xhat, dz)
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int rr = w + 8 * r2, li = rr / L, l = rr - li * L;
    const long long b = list0 + li;
    rvalid[r2] = li < nl && b < B;
    re[r2] = rvalid[r2] ? (long long)l * B + b : 0;
    rm[r2] = rvalid[r2] ? b * L + l : 0;
  }
  // ---- layer 0.  Everything the workgroup requests before its first GEMM, in the order the in-order load counter
  // wants it: what a later wait needs first is requested first.  doc ids (the gather's first round trip), the loss
  // operands and gamma / beta; then -- once the doc ids are here -- the gathered rows and their statistics (second round
  // trip); only then the first weight chunks and the L2 warm-up touches, which nothing waits for before the GEMM.
  int docid[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    // loaded on every path (re is 0 for a row that is not there) and selected afterwards: `valid ? docids[e] : n_rows`
    // becomes a load from a SELECTED address -- global memory or the kernel-argument segment, hence a flat load, and
    // while a flat load is in flight every wait of the wave has to be a wait for everything
    const int dv = p.docids[re[r2]];
    docid[r2] = rvalid[r2] ? dv : p.n_rows;
  }
  // ---- the lane's element of the loss: list list0 + lane / 16, rank lane % 16 (loads now, arithmetic later)
  const int li_l = lane >> 4, r_l = lane & 15;
  const long long b_l = list0 + li_l;
  const bool live = li_l < nl && r_l < L && b_l < B;
  const long long e_l = live ? (long long)r_l * B + b_l : 0;
  float c_l = 0.f, wgt_l = 0.f, pd_tp = 1.f, pd_tm = 1.f, rem_bias = 0.f, rem_p = 1.f;
  unsigned int rem_step = 0u;
  float tower_w = 0.f, tower_b = 0.f, ipw_t = 0.f;  // operands of the loss constants formed under the first GEMM
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
  float4 gm[NQ], btq[NQ];  // gamma / beta of the coming input stage, this lane's columns
  {
    const int K = p.K[0], nq = (((K + 15) & ~15) + 255) >> 8, k_last4 = ((K + 3) & ~3) - 4;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      btq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nq && has_ln) {
        const int kc = min(4 * lane + 256 * q, k_last4);
        gm[q] = *reinterpret_cast<const float4*>(p.gamma[0] + kc);
        btq[q] = *reinterpret_cast<const float4*>(p.beta[0] + kc);
      }
    }
  }
  TcGeo g = tc_geo(BF ? p.Wp[0] : p.W[0], p.K[0], p.N[0], p.ldn[0], w, BF);
  g.dbg = p.dbg;

  // ---- input stage of layer 0: gathered rows, normalised with the table's statistics
  float warm[2] = {0.f, 0.f};
  {
    const int K = p.K[0], K16 = (K + 15) & ~15, nq = (K16 + 255) >> 8, ldxh = (K + 3) & ~3;
    float4 x[2][NQ];
    float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
    bool pad[2];
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int d = docid[r2];
      pad[r2] = d >= p.n_rows;
      const float* src = p.table + (long long)(pad[r2] ? 0 : d) * p.ldx;
      if (has_ln) {
        const float2 st = *reinterpret_cast<const float2*>(p.row_stats + 2 * (long long)d);
        mean[r2] = st.x;
        rstd[r2] = st.y;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nq) x[r2][q] = *reinterpret_cast<const float4*>(src + min(4 * lane + 256 * q, p.ldx - 4));
    }
    // the first weight chunks, behind the rows in the load queue
    tc_prefetch<FAST, BF, true>(g, nn, kk, R);
    // ---- L2 warm-up: every workgroup starts at once, so the FIRST touch of each weight line would otherwise be paid
    // by all of them together, layer by layer, at HBM / Infinity-Cache latency in the middle of a GEMM.  The
    // workgroups that (by the round-robin placement; speed only) share an XCD each touch a slice of the streamed
    // matrices now: one or two lines per thread, no loop and no use of the values before the kernel's end, so that no
    // wait stands between these requests and the gather (a `sum += load` loop here waited for every touch in turn).
    {
      const int xi = blockIdx.x >> 3, nx = (gridDim.x + 7) >> 3;
      long long tot_lines = 0;
      for (int j = 0; j < H; ++j) {
        const long long tot = BF ? (long long)((p.K[j] + 15) & ~15) * p.ldn[j] / 2 : (long long)p.K[j] * p.ldn[j];
        const long long totT = j < 1 ? 0 : BF ? (long long)((p.N[j] + 15) & ~15) * ((p.K[j] + 3) & ~3) / 2
                                              : (long long)p.N[j] * ((p.K[j] + 3) & ~3);
        tot_lines += ((tot + 31) >> 5) + ((totT + 31) >> 5);
      }
      const long long per = (tot_lines + nx - 1) / nx;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const long long rel = tid + kTcThreads * r;
        long long line = xi * per + rel;
        const float* src = nullptr;
        if (rel >= per || line >= tot_lines) line = 0;  // nothing left to touch: the first line again (every lane issues)
        {
          for (int j = 0; j < H; ++j) {
            const long long tot = BF ? (long long)((p.K[j] + 15) & ~15) * p.ldn[j] / 2 : (long long)p.K[j] * p.ldn[j];
            const long long totT = j < 1 ? 0 : BF ? (long long)((p.N[j] + 15) & ~15) * ((p.K[j] + 3) & ~3) / 2
                                                  : (long long)p.N[j] * ((p.K[j] + 3) & ~3);
            const long long lines = (tot + 31) >> 5, linesT = (totT + 31) >> 5;
            if (!src && line < lines) src = (BF ? p.Wp[j] : p.W[j]) + line * 32;
            line -= lines;
            if (!src && line >= 0 && line < linesT) src = (BF ? p.WTp[j] : p.WT[j]) + line * 32;
            line -= linesT;
          }
        }
        // (an LDS-DMA touch would need no register, but the compiler orders every later LDS access behind it with a
        // vmcnt(0), which also waits for the weight chunks just requested)
        if (!src) src = BF ? p.Wp[0] : p.W[0];
        warm[r] = *src;
      }
    }
    // the loss constants' arithmetic, while the gathered rows are on their way (its operands came with the doc ids)
    if (RIDE == ULTR_RIDE_DLA) wgt_l = dla_tower_regs(tower_w, tower_b, L, r_l, p.dla.prob_kind, p.dla.max_pw).pw;
    if (RIDE == ULTR_RIDE_IPW && p.ipw.ipw_table) wgt_l = live && c_l > 0.f ? ipw_t : 0.f;  // ipw_weight()
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = w + 8 * r2;
      const float rs = pad[r2] ? 0.f : rstd[r2], mu = pad[r2] ? 0.f : mean[r2];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (q < nq) {
          const int k = 4 * lane + 256 * q;
          float xs[4], gg[4], bb[4], xh[4], a[4];
          tc_unpack(x[r2][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = tc_ok<FAST>(k, e, K);
            xh[e] = ok ? (xs[e] - mu) * rs : 0.f;  // a zero feature row normalises to beta
            a[e] = ok ? fmaf(xh[e], gg[e], bb[e]) : 0.f;
          }
          if (k < K16) {
            *reinterpret_cast<float4*>(P1 + row * PS + k) = make_float4(a[0], a[1], a[2], a[3]);
            if (rvalid[r2] && k < ldxh)
              *reinterpret_cast<float4*>(p.xhat[0] + rm[r2] * ldxh + k) = make_float4(xh[0], xh[1], xh[2], xh[3]);
          }
        }
    }
  }
  TC_STAMP(1);

  // ================================ forward: hidden layers ================================
  float4 woq[NQ];   // the output layer's weights (loaded under the last hidden layer's GEMM)
  for (int j = 0; j < H; ++j) {
    const int N = p.N[j];
    lds_barrier();  // the input panel is complete
    // gamma / beta of the NEXT input stage and this layer's bias ride along with the weight stream
    {
      const int K = p.K[j + 1], nq = (((K + 15) & ~15) + 255) >> 8, k_last4 = ((K + 3) & ~3) - 4;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
        btq[q] = woq[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
          const int kc = min(4 * lane + 256 * q, k_last4);
          if (has_ln) {
            gm[q] = *reinterpret_cast<const float4*>(p.gamma[j + 1] + kc);
            btq[q] = *reinterpret_cast<const float4*>(p.beta[j + 1] + kc);
          }
          if (j + 1 == H) woq[q] = *reinterpret_cast<const float4*>(p.W[H] + kc);
        }
      }
    }
    float* actj = smem + p.act_off[j];
    const int aps = p.act_ps[j];
    const float* biasj = p.bias[j];
    TC_WSTAMP(3 * j);
    tc_gemm<FAST, BF>(g, P1, PS, nn, kk, lane, R, biasj, [&](int u, f32x4 (&acc)[4], const float4& b4) {
      const int col0 = 64 * u + 4 * nn;
      const int n16 = (N + 15) & ~15;
      if (col0 < n16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float4 v;
          v.x = tc_ok<FAST>(col0, 0, N) ? tc_actf<FAST>(act, acc[0][r] + b4.x) : 0.f;
          v.y = tc_ok<FAST>(col0, 1, N) ? tc_actf<FAST>(act, acc[1][r] + b4.y) : 0.f;
          v.z = tc_ok<FAST>(col0, 2, N) ? tc_actf<FAST>(act, acc[2][r] + b4.z) : 0.f;
          v.w = tc_ok<FAST>(col0, 3, N) ? tc_actf<FAST>(act, acc[3][r] + b4.w) : 0.f;
          *reinterpret_cast<float4*>(actj + (4 * kk + r) * aps + col0) = v;
        }
      }
    }, j == ULTR_TC_STAMP_LAYER ? 10 : -1);
    TC_STAMP(2 + 3 * j);
    TC_WSTAMP(3 * j + 1);
    if (j + 1 < H) {  // the next layer's weight stream starts before its rows are normalised
      g = tc_geo(BF ? p.Wp[j + 1] : p.W[j + 1], p.K[j + 1], p.N[j + 1], p.ldn[j + 1], w, BF);
      g.dbg = p.dbg;
      tc_prefetch<FAST, BF>(g, nn, kk, R);
    }
    lds_barrier();  // the layer's activations are complete
    TC_STAMP(3 + 3 * j);
    if (j + 1 == H) break;

    // ---- input stage of layer j + 1: LayerNorm of the activations -> panel (and xhat to memory for the weight gradients)
    {
      const int K = N, K16 = (K + 15) & ~15, nq = (K16 + 255) >> 8, ldxh = (K + 3) & ~3;
      auto ln_rows = [&](auto nrc) {
        constexpr int NR = decltype(nrc)::value;
        float4 x[NR][NQ];
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nq) x[r2][q] = *reinterpret_cast<const float4*>(actj + (w + 8 * r2) * aps + min(4 * lane + 256 * q, K16 - 4));
        float mean[NR], rstd[NR];
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) { mean[r2] = 0.f; rstd[r2] = 1.f; }
        if (has_ln) {  // two-pass statistics (biased variance, keras LayerNormalization)
          float sm[NR], m2[NR];
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) {
            sm[r2] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if (q < nq) {
                const int k = 4 * lane + 256 * q;
                float xs[4];
                tc_unpack(x[r2][q], xs);
#pragma unroll
                for (int e = 0; e < 4; ++e) sm[r2] += tc_ok<FAST>(k, e, K) ? xs[e] : 0.f;
              }
          }
          wave_sum_rows<NR>(sm);
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) {
            mean[r2] = sm[r2] * (1.f / (float)K);
            m2[r2] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if (q < nq) {
                const int k = 4 * lane + 256 * q;
                float xs[4];
                tc_unpack(x[r2][q], xs);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float dd = tc_ok<FAST>(k, e, K) ? xs[e] - mean[r2] : 0.f;
                  m2[r2] = fmaf(dd, dd, m2[r2]);
                }
              }
          }
          wave_sum_rows<NR>(m2);
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) {
            rstd[r2] = tc_rsqrt<FAST>(m2[r2] * (1.f / (float)K) + p.eps);
            if (lane == 0) *reinterpret_cast<float2*>(s_stat + ((j + 1) * 16 + w + 8 * r2) * 2) = make_float2(mean[r2], rstd[r2]);
          }
        }
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) {
          const int row = w + 8 * r2;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              float xs[4], gg[4], bb[4], xh[4], a[4];
              tc_unpack(x[r2][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const bool ok = tc_ok<FAST>(k, e, K);
                xh[e] = ok ? (xs[e] - mean[r2]) * rstd[r2] : 0.f;
                a[e] = ok ? fmaf(xh[e], gg[e], bb[e]) : 0.f;
              }
              if (k < K16) {
                *reinterpret_cast<float4*>(P1 + row * PS + k) = make_float4(a[0], a[1], a[2], a[3]);
                if (k < ldxh) *reinterpret_cast<float4*>(p.xhat[j + 1] + rm[r2] * ldxh + k) = make_float4(xh[0], xh[1], xh[2], xh[3]);
              }
            }
        }
      };
      if (rvalid[1]) ln_rows(TcRows<2>{});       // wave-uniform (row w + 8 valid implies row w valid)
      else if (rvalid[0]) ln_rows(TcRows<1>{});
    }
    TC_STAMP(4 + 3 * j);
    TC_WSTAMP(3 * j + 2);
  }

  // ================================ output layer, loss, output-layer backward ================================
  const int Ko = p.K[H], Ko16 = (Ko + 15) & ~15, nqo = (Ko16 + 255) >> 8;
  float4 xo[2][NQ];
  float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
  {
    const float* actl = smem + p.act_off[H - 1];
    const int aps = p.act_ps[H - 1];
    auto out_rows = [&](auto nrc) {
      constexpr int NR = decltype(nrc)::value;
#pragma unroll
      for (int r2 = 0; r2 < NR; ++r2)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nqo) xo[r2][q] = *reinterpret_cast<const float4*>(actl + (w + 8 * r2) * aps + min(4 * lane + 256 * q, Ko16 - 4));
      if (has_ln) {
        float sm[NR], m2[NR];
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) {
          sm[r2] = 0.f;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nqo) {
              const int k = 4 * lane + 256 * q;
              float xs[4];
              tc_unpack(xo[r2][q], xs);
#pragma unroll
              for (int e = 0; e < 4; ++e) sm[r2] += tc_ok<FAST>(k, e, Ko) ? xs[e] : 0.f;
            }
        }
        wave_sum_rows<NR>(sm);
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) {
          mean[r2] = sm[r2] * (1.f / (float)Ko);
          m2[r2] = 0.f;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nqo) {
              const int k = 4 * lane + 256 * q;
              float xs[4];
              tc_unpack(xo[r2][q], xs);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float dd = tc_ok<FAST>(k, e, Ko) ? xs[e] - mean[r2] : 0.f;
                m2[r2] = fmaf(dd, dd, m2[r2]);
              }
            }
        }
        wave_sum_rows<NR>(m2);
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) rstd[r2] = tc_rsqrt<FAST>(m2[r2] * (1.f / (float)Ko) + p.eps);
      }
      float dot[NR];
#pragma unroll
      for (int r2 = 0; r2 < NR; ++r2) {
        dot[r2] = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nqo) {
            const int k = 4 * lane + 256 * q;
            float xs[4], gg[4], bb[4], ww[4];
            tc_unpack(xo[r2][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb); tc_unpack(woq[q], ww);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = tc_ok<FAST>(k, e, Ko) ? fmaf((xs[e] - mean[r2]) * rstd[r2], gg[e], bb[e]) : 0.f;
              dot[r2] = fmaf(a, ww[e], dot[r2]);
            }
          }
      }
      wave_sum_rows<NR>(dot);
#pragma unroll
      for (int r2 = 0; r2 < NR; ++r2) {
        const float sc = dot[r2] + ob;
        if (lane == 0) {
          s_score[w + 8 * r2] = sc;
          p.scores[re[r2]] = sc;
        }
      }
    };
    if (rvalid[1]) out_rows(TcRows<2>{});
    else if (rvalid[0]) out_rows(TcRows<1>{});
  }
  TC_STAMP(16);
  TC_WSTAMP(8);
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
      lw = group_sum<16>(live ? cl * (1.f / rem_p) : 0.f);  // clicks x (1 / propensity), as every REM kernel forms it
    }
    if (!live) ds = 0.f;
    if (w == 0 && live && r_l == 0 && p.list_weights && RIDE != ULTR_RIDE_PD) p.list_weights[b_l] = lw / (float)L;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int rr = w + 8 * r2, li = rr / L, l = rr - li * L;
      const int src = li < 4 ? 16 * li + l : 0;
      const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ds), src));
      go[r2] = rvalid[r2] ? v : 0.f;
    }
  }

  // the transposed weights of the first dgrad GEMM stream in under the output-layer backward stage and the partial-row
  // reduction (requested after them, the first chunk's latency was exposed: 0.7 us)
  if (H >= 2) {
    g = tc_geo(BF ? p.WTp[H - 1] : p.WT[H - 1], p.N[H - 1], p.K[H - 1], (p.K[H - 1] + 3) & ~3, w, BF);
    g.dbg = p.dbg;
    tc_prefetch<FAST, BF>(g, nn, kk, R);
  }
  // ---- output-layer backward: dz of the last hidden layer (-> P1 and memory) and the parameter-gradient partials
  {
    const int ldz = p.ldn[H - 1];
    float* mine = P2 + w * (3 * Ko + 1);  // [8 waves][3 Ko + 1]
    float aW[NQ][4], aG[NQ][4], aB[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) aW[q][e] = aG[q][e] = aB[q][e] = 0.f;
    auto outbwd_rows = [&](auto nrc) {
      constexpr int NR = decltype(nrc)::value;
      float s1[NR], s2[NR];
#pragma unroll
      for (int r2 = 0; r2 < NR; ++r2) s1[r2] = s2[r2] = 0.f;
      if (has_ln) {
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nqo) {
              const int k = 4 * lane + 256 * q;
              float xs[4], gg[4], ww[4];
              tc_unpack(xo[r2][q], xs); tc_unpack(gm[q], gg); tc_unpack(woq[q], ww);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float gwg = tc_ok<FAST>(k, e, Ko) ? go[r2] * ww[e] * gg[e] : 0.f;
                s1[r2] += gwg;
                s2[r2] = fmaf(gwg, (xs[e] - mean[r2]) * rstd[r2], s2[r2]);
              }
            }
        wave_sum_rows<NR>(s1);
        wave_sum_rows<NR>(s2);
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) {
          s1[r2] *= (1.f / (float)Ko);
          s2[r2] *= (1.f / (float)Ko);
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < NR; ++r2) {
        const int row = w + 8 * r2;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < nqo) {
            const int k = 4 * lane + 256 * q;
            float xs[4], gg[4], bb[4], ww[4], o4[4];
            tc_unpack(xo[r2][q], xs); tc_unpack(gm[q], gg); tc_unpack(btq[q], bb); tc_unpack(woq[q], ww);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool ok = tc_ok<FAST>(k, e, Ko);
              const float gw = ok ? go[r2] * ww[e] : 0.f;
              float dx = gw;
              if (has_ln) {
                const float xh = (xs[e] - mean[r2]) * rstd[r2];
                aW[q][e] += ok ? fmaf(gg[e], xh, bb[e]) * go[r2] : 0.f;
                aG[q][e] = fmaf(gw, xh, aG[q][e]);
                aB[q][e] += gw;
                dx = rstd[r2] * (gw * gg[e] - s1[r2] - xh * s2[r2]);
              } else {
                aW[q][e] += ok ? xs[e] * go[r2] : 0.f;
              }
              o4[e] = ok ? dx * tc_actg<FAST>(act, xs[e]) : 0.f;
            }
            if (k < Ko16) {
              const float4 ov = make_float4(o4[0], o4[1], o4[2], o4[3]);
              *reinterpret_cast<float4*>(P1 + row * PS + k) = ov;
              if (k < ldz) *reinterpret_cast<float4*>(p.dz[H - 1] + rm[r2] * ldz + k) = ov;
            }
          }
      }
    };
    if (rvalid[1]) outbwd_rows(TcRows<2>{});
    else if (rvalid[0]) outbwd_rows(TcRows<1>{});
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
    if (lane == 0) mine[3 * Ko] = go[0] + go[1];
  }

  int cur_is_p1 = 1;
  TC_STAMP(17);
  TC_WSTAMP(9);
  lds_barrier();
  {
    float* out = p.outpart + (long long)(p.part_base + blockIdx.x) * (3 * Ko + 4);
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
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      gm[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (q < nq && has_ln) gm[q] = *reinterpret_cast<const float4*>(p.gamma[j] + min(4 * lane + 256 * q, ((K + 3) & ~3) - 4));
    }
#ifdef ULTR_STAMPS
    if (j == 1 && threadIdx.x == 0 && blockIdx.x < 1024) g_tc_stamps[blockIdx.x * 32 + 28] = __builtin_amdgcn_s_memtime();
#endif
    tc_gemm<FAST, BF>(g, cur, PS, nn, kk, lane, R, nullptr, [&](int u, f32x4 (&acc)[4], const float4&) {
      const int col0 = 64 * u + 4 * nn;
      if (col0 < K16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float4*>(nxt + (4 * kk + r) * PS + col0) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
      }
    });
    if (j >= 2) {  // the next layer's weight stream starts under the row pass
      g = tc_geo(BF ? p.WTp[j - 1] : p.WT[j - 1], p.N[j - 1], p.K[j - 1], (p.K[j - 1] + 3) & ~3, w, BF);
      g.dbg = p.dbg;
      tc_prefetch<FAST, BF>(g, nn, kk, R);
    }
#ifdef ULTR_STAMPS
    if (j == 1 && threadIdx.x == 0 && blockIdx.x < 1024) g_tc_stamps[blockIdx.x * 32 + 29] = __builtin_amdgcn_s_memtime();
#endif
    TC_WSTAMP(j == 1 ? 12 : 10);
    TC_STAMP(19 + 2 * (H - 1 - j));
    lds_barrier();
    {
      const float* acts = smem + p.act_off[j - 1];
      const int aps = p.act_ps[j - 1];
      auto rowpass_rows = [&](auto nrc) {
        constexpr int NR = decltype(nrc)::value;
        float4 d4[NR][NQ], h4[NR][NQ];
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nq) {
              const int kc = min(4 * lane + 256 * q, K16 - 4);
              d4[r2][q] = *reinterpret_cast<const float4*>(nxt + (w + 8 * r2) * PS + kc);
              h4[r2][q] = *reinterpret_cast<const float4*>(acts + (w + 8 * r2) * aps + kc);
            }
        float mu[NR], rs[NR], c1[NR], c2[NR];
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) { mu[r2] = 0.f; rs[r2] = 1.f; c1[r2] = c2[r2] = 0.f; }
        if (has_ln) {
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) {
            const float2 st = *reinterpret_cast<const float2*>(s_stat + (j * 16 + w + 8 * r2) * 2);
            mu[r2] = st.x;
            rs[r2] = st.y;
          }
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
              if (q < nq) {
                const int k = 4 * lane + 256 * q;
                float dd[4], gg[4], hh[4];
                tc_unpack(d4[r2][q], dd); tc_unpack(gm[q], gg); tc_unpack(h4[r2][q], hh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float gv = tc_ok<FAST>(k, e, K) ? dd[e] * gg[e] : 0.f;
                  c1[r2] += gv;
                  c2[r2] = fmaf(gv, (hh[e] - mu[r2]) * rs[r2], c2[r2]);
                }
              }
          wave_sum_rows<NR>(c1);
          wave_sum_rows<NR>(c2);
#pragma unroll
          for (int r2 = 0; r2 < NR; ++r2) {
            c1[r2] *= (1.f / (float)K);
            c2[r2] *= (1.f / (float)K);
          }
        }
#pragma unroll
        for (int r2 = 0; r2 < NR; ++r2) {
          const int row = w + 8 * r2;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nq) {
              const int k = 4 * lane + 256 * q;
              float dd[4], gg[4], hh[4], o4[4];
              tc_unpack(d4[r2][q], dd); tc_unpack(gm[q], gg); tc_unpack(h4[r2][q], hh);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float dx = dd[e];
                if (has_ln) dx = rs[r2] * (dx * gg[e] - c1[r2] - (hh[e] - mu[r2]) * rs[r2] * c2[r2]);
                o4[e] = tc_ok<FAST>(k, e, K) ? dx * tc_actg<FAST>(act, hh[e]) : 0.f;
              }
              if (k < K16) {
                const float4 ov = make_float4(o4[0], o4[1], o4[2], o4[3]);
                if (j >= 2) *reinterpret_cast<float4*>(nxt + row * PS + k) = ov;
                if (k < ldo) *reinterpret_cast<float4*>(p.dz[j - 1] + rm[r2] * ldo + k) = ov;
              }
            }
        }
      };
      if (rvalid[1]) rowpass_rows(TcRows<2>{});
      else if (rvalid[0]) rowpass_rows(TcRows<1>{});
    }
    TC_WSTAMP(j == 1 ? 13 : 11);
    TC_STAMP(20 + 2 * (H - 1 - j));
    if (j >= 2) lds_barrier();
    cur_is_p1 ^= 1;
  }
  TC_STAMP(31);
  if (warm[0] + warm[1] == 1.2345e-30f && p.B < 0) p.outpart[0] = warm[0];  // never true: keeps the warm-up loads alive
}

}  // namespace ultr
#include "chain32.hpp"
namespace ultr {

// WT_j[n][k] = W_j[k][n] for the hidden layers j >= 1 (32 x 32 tiles through LDS), one launch for all of them: the
// step's preparation launch when the caller's update call has not refreshed the copies (ultr_clip_adagrad_fused does).
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

// The bf16 step's weight copies, one launch for all hidden layers (the step's preparation launch when the caller's
// update call has not refreshed them; ultr_clip_adagrad_fused does): W_j -> Wp_j[(k / 4)][n][k % 4] and, for j >= 1,
// W_j^T -> WTp_j[(n / 4)][k][n % 4], rounded to nearest even.  Rows beyond the matrix stay zero (zeroed workspace).
struct PackArgs {
  int n; const float* W[ULTR_MAX_LAYERS]; unsigned short* Wp[ULTR_MAX_LAYERS]; unsigned short* WTp[ULTR_MAX_LAYERS];
  int K[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  long long start[ULTR_MAX_LAYERS + 1];  // in float4 units
};
__global__ void __launch_bounds__(256) pack_weights_kernel(PackArgs p) {
  const long long total = p.start[p.n];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    int j = 0;
    while (j + 1 < p.n && i >= p.start[j + 1]) ++j;
    const long long rel = (i - p.start[j]) << 2;
    const int ldn = p.ldn[j], k = (int)(rel / ldn), n0 = (int)(rel - (long long)k * ldn);
    const float4 v = *reinterpret_cast<const float4*>(p.W[j] + rel);
    pack_store(p.Wp[j], p.WTp[j], k, n0, ldn, (p.K[j] + 3) & ~3, v.x, v.y, v.z, v.w);
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradients through an LDS ring filled by LDS-DMA.
//   G[k][n] = sum_m xhat[m][k] dz[m][n]  (+ S[n] = sum_m dz[m][n]),  64 x 64 outputs per workgroup (2 x 2 waves of
//   32 x 32), rows split into slabs over the grid.
// Both operands were written by the launch before: every first touch comes from the Infinity Cache / HBM at 1-2 us.
// The register-streaming wgrad_body keeps two 32-row chunks in flight per wave and ran latency-bound (measured: 1.3 to
// 1.5 us per chunk against 0.47 us of MFMA).  Here global_load_lds_dwordx4 moves the 32 x 64 tiles of both operands
// into a ring of kWgStages slots without touching a register, kWgStages - 1 chunks ahead of the multiply, and the
// waves wait with counted vmcnt: one raw barrier per chunk, no vmcnt(0) inside the loop.
//   Rows beyond M: the operand buffers hold zero rows up to the next multiple of 32 (the fused plan pads them; the
//   workspace is zero-initialised by the caller and nothing ever writes there), so no chunk needs a row mask.
// ---------------------------------------------------------------------------------------------
constexpr int kWgStages = 4;
constexpr int kWgTile = 32 * 64;               // floats per operand tile
constexpr int kWgSlot = 2 * kWgTile;           // A then B
constexpr size_t kWgRingBytes = sizeof(float) * kWgStages * kWgSlot;


// BF (the bf16 step): the same ring of fp32 tiles; the operands are rounded to bf16 on their way from LDS to the
// registers and multiplied by v_mfma_f32_32x32x16_bf16 (two per 32-row chunk instead of sixteen fp32 MFMAs), fp32 accumulate.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
template <bool BF>
__device__ __forceinline__ void wgrad_lds_body(const WgradArgs& p, int bx, int by, int bz, float* ring) {
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int k0 = by * 64, n0 = bz * 64;
  const long long m_begin = (long long)bx * p.rows_per_split;
  long long m_end = m_begin + p.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nch = (int)((m_end - m_begin + 31) >> 5);
  // this wave's DMA pieces: rows 8 w .. 8 w + 7 of each tile, 4 rows (1 KB) per instruction; lane -> row lane / 16,
  // 16 bytes at column 4 (lane % 16)
  const int prow = 8 * w + (lane >> 4), pcol = 4 * (lane & 15);
  const float* a_src = p.xhat + (m_begin + prow) * p.ldx + min(k0 + pcol, p.ldx - 4);
  const float* b_src = p.dz + (m_begin + prow) * p.ldn + min(n0 + pcol, p.ldn - 4);
  const size_t a_step = (size_t)32 * p.ldx, b_step = (size_t)32 * p.ldn;
  const size_t a_row4 = (size_t)4 * p.ldx, b_row4 = (size_t)4 * p.ldn;
  auto issue = [&](int c) {
    if (c < nch) {  // wave-uniform
      float* slot = ring + (c % kWgStages) * kWgSlot + 8 * w * 64;
      const float* ap = a_src + (size_t)c * a_step;
      const float* bp = b_src + (size_t)c * b_step;
      // Through asm, not the builtin: the compiler books an LDS-DMA load on the LDS counter as well (as if it were a flat
      // access that may complete out of order), and from then on every wait for an LDS READ in the loop is lgkmcnt(0)
      // -- the multiply loop below then waits for the operands it has just requested instead of the ones it needs.
      // The hardware counts these loads on vmcnt only, which the loop waits for by hand anyway.
      const unsigned l0 = (unsigned)(size_t)ULTR_LDS_PTR(slot);
      asm volatile(
          "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
          "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
          "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"
          :
          : "v"(ap), "v"(ap + a_row4), "v"(bp), "v"(bp + b_row4), "s"(l0), "s"(l0 + 0x400u), "s"(l0 + 0x2000u), "s"(l0 + 0x2400u)
          : "memory");  // (m0 is written; it cannot be named as a clobber, and nothing else in this kernel uses it)
    }
  };
  WG_STAMP(0);
#pragma unroll
  for (int c = 0; c < kWgStages - 1; ++c) issue(c);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;
  const bool do_colsum = by == 0 && wr == 0;
  const int a_off = h * 64 + wr * 32 + l31, b_off = kWgTile + h * 64 + wc * 32 + l31;
  const int a_off8 = 8 * h * 64 + wr * 32 + l31, b_off8 = kWgTile + 8 * h * 64 + wc * 32 + l31;
  for (int c = 0; c < nch; ++c) {
    // chunk c has landed once at most `ahead` younger chunks (4 DMA each) are still in flight
    const int ahead = min(kWgStages - 2, nch - 1 - c);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (c == 0) WG_STAMP(1);
    if (c == 1) WG_STAMP(2);
    issue(c + kWgStages - 1);  // into the slot of chunk c - 1: every wave has multiplied it out (the barrier above)
    const float* slot = ring + (c % kWgStages) * kWgSlot;
    float a[16], b[16];
    if (BF) {
      // lane (l31, h) of MFMA s holds rows 16 s + 8 h + e (e = 0..7) of its column: a[8 s + e]
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = 16 * (t >> 3) + (t & 7);
        a[t] = slot[a_off8 + 64 * row];
        b[t] = slot[b_off8 + 64 * row];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint4 ua = make_uint4(pk_bf16(a[8 * s2], a[8 * s2 + 1]), pk_bf16(a[8 * s2 + 2], a[8 * s2 + 3]),
                                    pk_bf16(a[8 * s2 + 4], a[8 * s2 + 5]), pk_bf16(a[8 * s2 + 6], a[8 * s2 + 7]));
        const uint4 ub = make_uint4(pk_bf16(b[8 * s2], b[8 * s2 + 1]), pk_bf16(b[8 * s2 + 2], b[8 * s2 + 3]),
                                    pk_bf16(b[8 * s2 + 4], b[8 * s2 + 5]), pk_bf16(b[8 * s2 + 6], b[8 * s2 + 7]));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, ua), __builtin_bit_cast(wg_bf16x8, ub), acc, 0, 0, 0);
      }
    } else {
      // pinned order: the operands of MFMA pair q + 1 are requested before pair q is issued, so that only the first
      // pair's LDS latency stands in front of the chunk's multiplies
      auto rd = [&](int q) {
#pragma unroll
        for (int t = 2 * q; t < 2 * q + 2; ++t) {
          a[t] = slot[a_off + 128 * t];
          b[t] = slot[b_off + 128 * t];
        }
      };
      rd(0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (q < 7) rd(q + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * q], b[2 * q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * q + 1], b[2 * q + 1], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_colsum) {
      float s4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s4[q] = (b[4 * q] + b[4 * q + 1]) + (b[4 * q + 2] + b[4 * q + 3]);
      colsum += (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
  }
  WG_STAMP(3);
  float* gs = p.gslab + (long long)bx * p.K * p.ldn;
  const int col = n0 + wc * 32 + l31;
  if (col < p.ldn) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = k0 + wr * 32 + acc_row(r, h);
      if (k < p.K) gs[(long long)k * p.ldn + col] = acc[r];
    }
  }
  if (do_colsum) {
    float other;
    const float mine = xor32_partner(colsum, &other);  // the two wave halves hold alternate rows
    if (h == 0 && col < p.ldn) p.sslab[(long long)bx * p.ldn + col] = mine + other;
  }
  WG_STAMP(4);
}

// Every layer's weight-gradient GEMM in one launch: exactly the workgroups that fit the chip at once (two per CU).
struct WgradAllArgs {
  int n_layers, nsplit, bf16;
  WgradArgs g[ULTR_MAX_LAYERS];
  int start[ULTR_MAX_LAYERS + 1], gy[ULTR_MAX_LAYERS];
};
__global__ void __launch_bounds__(kThreads, 2) wgrad_all_kernel(WgradAllArgs p) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int id = blockIdx.x;
  int j = 0;
  while (j + 1 < p.n_layers && id >= p.start[j + 1]) ++j;
  const int local = id - p.start[j];
  const int bx = local % p.nsplit, t = local / p.nsplit;
  if (p.bf16) wgrad_lds_body<true>(p.g[j], bx, t % p.gy[j], t / p.gy[j], ring);
  else wgrad_lds_body<false>(p.g[j], bx, t % p.gy[j], t / p.gy[j], ring);
}

// The slab reduction, with the step's latency-bound side jobs as rider workgroups (they come first): the in-step
// metrics, the loss statistics (both need only scores / clicks / list weights, which the chain kernel left two
// launches ago, and feed nothing before the update launch).  The reduction's own workgroups are short and leave most
// of the chip's workgroup slots free.  (The assembly of the NEXT batch overwrites what these riders read: it rides
// one launch later, with the update.)
struct FinalizeFusedArgs {
  FinalizeArgs f;
  MetricsArgs m; int m_blocks, m_pad;
  int ride; DlaArgs dla; IpwArgs ipw; RemArgs rem; PdArgs pd; int l_blocks, l_pad;
};
template <int RIDE>  // the loss rider compiled in (the kernel's register count is the maximum over its bodies)
__global__ void __launch_bounds__(kThreads) finalize_fused_kernel(FinalizeFusedArgs p) {
  int id = blockIdx.x;
  if (id < p.m_pad) {
    if (id >= p.m_blocks) return;
    __shared__ float s_vals[4 * kMaxOut];
    metrics_short_lists_t<16>(p.m, id, p.m_blocks, s_vals);  // (other group widths get a launch of their own)
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
    if (RIDE == ULTR_RIDE_DLA) dla_loss_small_body(p.dla, id, p.l_blocks, s_acc, s_part, s_tower);
    if (RIDE == ULTR_RIDE_IPW) ipw_loss_small_body(p.ipw, id, p.l_blocks, s_acc, s_part);
    if (RIDE == ULTR_RIDE_REM) rem_loss_small_body(p.rem, id, p.l_blocks, s_acc, s_part);
    if (RIDE == ULTR_RIDE_PD) pd_loss_small_body(p.pd, id, p.l_blocks, s_acc, s_part, s_tower);
    return;
  }
  finalize_body(p.f, id - p.l_pad);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// LDS geometry of the 16-row kernel
struct Chain16Geo {
  int PS, panel_floats, act_off[ULTR_MAX_LAYERS], act_ps[ULTR_MAX_LAYERS], p1_off, p2_off, misc_off;
  size_t smem_bytes;
  bool ok;
};
struct FusedPlan {
  int H, nl, nwg;
  int rows;  // 16: train_chain_kernel; 32: train_chain32_kernel (batches of more than one workgroup per CU)
  // rows == 32 only: the batch's last lists as a second launch of 16-row workgroups (tail_nwg of them, tail_nl lists each,
  // from list tail_base on) when that ends sooner than a last, mostly empty round of 32-row workgroups; 0: none
  int tail_nwg, tail_nl, tail_base;
  Chain16Geo tail;
  int PS2;
  bool fast;  // the reference's default scorer with aligned widths: the compile-time specialised kernels
  int K[ULTR_MAX_LAYERS], N[ULTR_MAX_LAYERS], ldn[ULTR_MAX_LAYERS];
  int PS, panel_floats, act_off[ULTR_MAX_LAYERS], act_ps[ULTR_MAX_LAYERS], p1_off, p2_off, misc_off;
  size_t smem_bytes;
  // workspace (floats)
  long long xhat[ULTR_MAX_LAYERS], dz[ULTR_MAX_LAYERS], wt[ULTR_MAX_LAYERS], gslab[ULTR_MAX_LAYERS], sslab[ULTR_MAX_LAYERS], outpart;
  long long wp[ULTR_MAX_LAYERS], wtp[ULTR_MAX_LAYERS];  // packed bf16 copies (compute_dtype bf16 only)
  bool bf16;
  int nsplit, rows_per_split;
  long long total;
  bool ok;
};

// The 32-row kernel pays while the 16-row one would need more than one workgroup per CU (256 CUs): ULTR_FUSED_ROWS=16|32
// forces one of them (diagnostics, tests).
static int fused_rows_override() {
  static const int v = [] { const char* e = getenv("ULTR_FUSED_ROWS"); return e ? atoi(e) : 0; }();
  return v;
}
static Chain16Geo chain16_geo(int H, const int* N, int KH, int maxw) {
  Chain16Geo G{};
  G.PS = ((maxw + 31) & ~31) + 4;
  G.panel_floats = kTcRows * G.PS > kTcExch ? kTcRows * G.PS : kTcExch;
  if (8 * (3 * KH + 1) > G.panel_floats) return G;
  int off = 0;
  for (int j = 0; j < H; ++j) {
    G.act_ps[j] = ((N[j] + 31) & ~31) + 4;
    G.act_off[j] = off;
    off += kTcRows * G.act_ps[j];
  }
  G.p1_off = off; off += G.panel_floats;
  G.p2_off = off; off += G.panel_floats;
  G.misc_off = off; off += kTcMiscFloats;
  G.smem_bytes = sizeof(float) * (size_t)off;
  G.ok = G.smem_bytes <= 160 * 1024;
  return G;
}
static FusedPlan make_fused_plan(const UltrDnnDesc* d, int B, int L) {
  FusedPlan P{};
  P.ok = false;
  P.H = d->n_layers - 1;
  P.bf16 = d->compute_dtype == ULTR_DTYPE_BF16;
  if (P.H < 1 || L < 1 || L > 16 || B < 1) return P;
  if (d->norm != ULTR_NORM_NONE && d->norm != ULTR_NORM_LAYER) return P;  // batch statistics couple the rows: per-layer plan
  const long long M = (long long)B * L;
  int cur = d->in_features, maxw = d->in_features, maxh = 0;
  for (int j = 0; j <= P.H; ++j) {
    P.K[j] = cur;
    P.N[j] = d->sizes[j];
    P.ldn[j] = round4(d->sizes[j]);
    cur = d->sizes[j];
    if (j < P.H && P.N[j] > maxw) maxw = P.N[j];
    if (j < P.H && P.N[j] > maxh) maxh = P.N[j];
  }
  if (maxw > 1024) return P;
  P.fast = d->activation == ULTR_ACT_ELU && d->norm == ULTR_NORM_LAYER && (P.K[0] & 3) == 0;
  for (int j = 1; j <= P.H; ++j) P.fast = P.fast && (P.K[j] & 15) == 0;
  const int nl16 = 16 / L > 4 ? 4 : 16 / L;  // one 16-lane group per list in the loss lanes of a wave
  const Chain16Geo g16 = chain16_geo(P.H, P.N, P.K[P.H], maxw);
  constexpr int kCUs = 256;
  // ---- 32 rows per workgroup?
  P.rows = 16;
  {
    const int nl32 = 32 / L > 4 ? 4 : 32 / L;
    const int PSa = ((maxw + 31) & ~31) + 4, PSb = ((maxh + 31) & ~31) + 4;
    // (either panel serves as the split-K exchange buffer of the GEMM that has just multiplied it out)
    const int pa = kC32Rows * PSa > 7 * 2048 ? kC32Rows * PSa : 7 * 2048, pb = kC32Rows * PSb > 7 * 2048 ? kC32Rows * PSb : 7 * 2048;
    const size_t bytes = sizeof(float) * ((size_t)pa + pb + kC32MiscFloats);
    const bool can32 = P.fast && !P.bf16 && maxw <= 768 && nl32 > nl16 && bytes <= 160 * 1024 && 8 * (3 * P.K[P.H] + 1) <= pb;
    const int ov = fused_rows_override();
    if (can32 && (ov == 32 || (ov != 16 && cdiv(B, nl16) > kCUs))) {
      P.rows = 32;
      P.nl = nl32;
      P.nwg = (int)cdiv(B, P.nl);
      P.PS = PSa; P.PS2 = PSb;
      P.panel_floats = pa;
      P.p1_off = 0; P.p2_off = pa; P.misc_off = pa + pb;
      P.smem_bytes = bytes;
      // One workgroup per CU at a time, so the launch runs in rounds of 256 workgroups and a last round of a few
      // workgroups costs as much as a full one.  A 16-row workgroup takes about 0.6 of the time of a 32-row one: when the
      // lists beyond the last full round fit ONE round of 16-row workgroups, they run as a second launch of those
      // (B = 4096, lists of 10: 5 rounds of 3 lists + 256 single lists instead of 6 rounds).
      const long long full = (long long)(P.nwg / kCUs) * kCUs, rest = B - full * nl32;
      if (ov != 32 && g16.ok && full > 0 && rest > 0 && rest <= (long long)kCUs * nl16) {
        P.nwg = (int)full;
        P.tail = g16;
        P.tail_nl = nl16;
        P.tail_base = (int)(full * nl32);
        P.tail_nwg = (int)cdiv(rest, nl16);
      }
    }
  }
  if (P.rows == 16) {
    if (!g16.ok) return P;
    P.nl = nl16;
    P.nwg = (int)cdiv(B, P.nl);
    P.PS = g16.PS; P.panel_floats = g16.panel_floats;
    for (int j = 0; j < P.H; ++j) { P.act_ps[j] = g16.act_ps[j]; P.act_off[j] = g16.act_off[j]; }
    P.p1_off = g16.p1_off; P.p2_off = g16.p2_off; P.misc_off = g16.misc_off;
    P.smem_bytes = g16.smem_bytes;
  }
  long long o = 0;
  auto take = [&](long long n) {
    const long long at = o;
    o += (n + 63) & ~63ll;
    return at;
  };
  long long tiles = 0;
  for (int j = 0; j < P.H; ++j) tiles += cdiv(P.K[j], 64) * cdiv(P.N[j], 64);
  constexpr int wg_target = 512;  // two workgroups per CU (measured against 256, 384 and 768 at the headline shape)
  long long ns = wg_target / tiles;
  if (ns < 1) ns = 1;
  if (ns > cdiv(M, 64)) ns = cdiv(M, 64);
  P.rows_per_split = (int)(cdiv(cdiv(M, ns), 32) * 32);
  P.nsplit = (int)cdiv(M, P.rows_per_split);
  const long long Mpad = (M + 31) & ~31ll;  // zero rows up to a whole 32-row chunk (wgrad_lds_body)
  for (int j = 0; j < P.H; ++j) {
    P.xhat[j] = take(Mpad * round4(P.K[j]));
    P.dz[j] = take(Mpad * P.ldn[j]);
    P.wt[j] = j >= 1 && !P.bf16 ? take((long long)P.N[j] * round4(P.K[j])) : 0;
    P.wp[j] = P.bf16 ? take((long long)((P.K[j] + 15) & ~15) * P.ldn[j] / 2) : 0;
    P.wtp[j] = P.bf16 && j >= 1 ? take((long long)((P.N[j] + 15) & ~15) * round4(P.K[j]) / 2) : 0;
    P.gslab[j] = take((long long)P.nsplit * P.K[j] * P.ldn[j]);
    P.sslab[j] = take((long long)P.nsplit * P.ldn[j]);
  }
  P.outpart = take((long long)(P.nwg + P.tail_nwg) * (3 * P.K[P.H] + 4));
  P.total = o;
  P.ok = true;
  return P;
}

}  // namespace ultr

namespace ultr {
int launch_wgrad_all(int n_layers, const WgradArgs* layers, int nsplit, bool bf16, hipStream_t st) {
  WgradAllArgs g{};
  g.n_layers = n_layers; g.nsplit = nsplit; g.bf16 = bf16 ? 1 : 0;
  int start = 0;
  long long pw = 0;
  for (int j = 0; j < n_layers; ++j) {
    g.g[j] = layers[j];
    g.start[j] = start; g.gy[j] = (int)cdiv(layers[j].K, 64);
    start += nsplit * g.gy[j] * (int)cdiv(layers[j].ldn, 64);
    pw += (long long)layers[j].K * layers[j].N;
  }
  g.start[n_layers] = start;
  static PerDeviceOnce wg_attr;
  if (const int rc_ = raise_lds_limit(wg_attr, reinterpret_cast<const void*>(wgrad_all_kernel), (int)kWgRingBytes, "wgrad_all_kernel")) return rc_;
  {
    ProfScope prof_(g_profile_on ? profile_intern(bf16 ? "wgrad_all_kernel_bf16/P%lld" : "wgrad_all_kernel/P%lld", pw)
                                 : "wgrad_all_kernel", st);
    hipLaunchKernelGGL(wgrad_all_kernel, dim3((unsigned)start), dim3(kThreads), kWgRingBytes, st, g);
  }
  return check_launch("wgrad_all_kernel");
}
}  // namespace ultr

using namespace ultr;

#ifdef ULTR_STAMPS
extern "C" int ultr_debug_wg_stamps(long long* host, int n) {
  if (n > 1024 * 8) n = 1024 * 8;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), sizeof(long long) * (size_t)n);
  return n;
}
extern "C" int ultr_debug_tc_wstamps(long long* host, int n) {
  if (n > 1024 * 128) n = 1024 * 128;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tc_wstamps), sizeof(long long) * (size_t)n);
  return n;
}
extern "C" int ultr_debug_tc_stamps(long long* host, int n) {
  if (n > 1024 * 32) n = 1024 * 32;
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tc_stamps), sizeof(long long) * (size_t)n);
  return n;
}
#endif

extern "C" int ultr_dnn_fused_rows(const UltrDnnDesc* desc, int B, int L) {
  if (ultr_check_desc(desc, "ultr_dnn_fused_rows") != ULTR_OK) return -1;
  const FusedPlan P = make_fused_plan(desc, B, L);
  return P.ok ? P.rows : 0;
}

extern "C" int ultr_dnn_fused_tail_lists(const UltrDnnDesc* desc, int B, int L) {
  if (ultr_check_desc(desc, "ultr_dnn_fused_tail_lists") != ULTR_OK) return -1;
  const FusedPlan P = make_fused_plan(desc, B, L);
  return P.ok && P.tail_nwg > 0 ? B - P.tail_base : 0;
}

extern "C" long long ultr_dnn_fused_workspace_bytes(const UltrDnnDesc* desc, int B, int L) {
  if (ultr_check_desc(desc, "ultr_dnn_fused_workspace_bytes") != ULTR_OK) return -1;
  const FusedPlan P = make_fused_plan(desc, B, L);
  return P.ok ? P.total * (long long)sizeof(float) : 0;
}

template <int RIDE, bool FAST, int NQ, bool BF>
static int launch_chain3(const TcArgs& a, int nwg, size_t smem, hipStream_t st) {
  static PerDeviceOnce attr_done;
  if (const int rc_ = raise_lds_limit(attr_done, reinterpret_cast<const void*>(train_chain_kernel<RIDE, FAST, NQ, BF>), 160 * 1024,
                                      "ultr_dnn_train_fused")) return rc_;
  long long pw = 0;
  for (int j = 0; j < a.H; ++j) pw += (long long)a.K[j] * a.N[j] + (j >= 1 ? (long long)a.K[j] * a.N[j] : 0);
  {
    // (a launch that covers part of the batch carries its row count: the tail launch beside train_chain32_kernel)
    const long long part_rows = (long long)a.L * ((long long)nwg * a.nl < a.B - a.list_base ? (long long)nwg * a.nl : a.B - a.list_base);
    const bool part = part_rows != (long long)a.B * a.L;
    ultr::ProfScope prof_(!ultr::g_profile_on ? "train_chain_kernel"
                          : part ? ultr::profile_intern("train_chain_kernel/P%lldM%lld", pw, part_rows)
                                 : ultr::profile_intern(BF ? "train_chain_kernel_bf16/P%lld" : "train_chain_kernel/P%lld", pw), st);
    hipLaunchKernelGGL((train_chain_kernel<RIDE, FAST, NQ, BF>), dim3((unsigned)nwg), dim3(kTcThreads), smem, st, a);
  }
  return check_launch("train_chain_kernel");
}
template <int RIDE, int NQ>
static int launch_chain32(const TcArgs& a, int nwg, size_t smem, hipStream_t st) {
  static PerDeviceOnce attr_done;
  if (const int rc_ = raise_lds_limit(attr_done, reinterpret_cast<const void*>(train_chain32_kernel<RIDE, NQ>), 160 * 1024,
                                      "ultr_dnn_train_fused")) return rc_;
  long long pw = 0;
  for (int j = 0; j < a.H; ++j) pw += (long long)a.K[j] * a.N[j] + (j >= 1 ? (long long)a.K[j] * a.N[j] : 0);
  {
    const long long part_rows = (long long)a.L * ((long long)nwg * a.nl < a.B ? (long long)nwg * a.nl : a.B);
    const bool part = part_rows != (long long)a.B * a.L;
    ultr::ProfScope prof_(!ultr::g_profile_on ? "train_chain32_kernel"
                          : part ? ultr::profile_intern("train_chain32_kernel/P%lldM%lld", pw, part_rows)
                                 : ultr::profile_intern("train_chain32_kernel/P%lld", pw), st);
    hipLaunchKernelGGL((train_chain32_kernel<RIDE, NQ>), dim3((unsigned)nwg), dim3(kTcThreads), smem, st, a);
  }
  return check_launch("train_chain32_kernel");
}
template <int RIDE>
static int launch_chain32_q(const TcArgs& a, int nwg, size_t smem, hipStream_t st) {
  int maxk = 0;
  for (int j = 0; j <= a.H; ++j) maxk = a.K[j] > maxk ? a.K[j] : maxk;
  if (maxk <= 512) return launch_chain32<RIDE, 2>(a, nwg, smem, st);
  return launch_chain32<RIDE, 3>(a, nwg, smem, st);
}
template <int RIDE, bool FAST, int NQ>
static int launch_chain2(const TcArgs& a, int nwg, size_t smem, hipStream_t st) {
  if (a.Wp[0]) return launch_chain3<RIDE, FAST, NQ, true>(a, nwg, smem, st);
  return launch_chain3<RIDE, FAST, NQ, false>(a, nwg, smem, st);
}
// `fast`: the reference's default scorer (elu, layer normalisation), input width a multiple of 4, hidden widths multiples
// of 16, and layer 0's last weight chunk readable up to the next multiple of 16 rows (decided by the caller)
template <int RIDE>
static int launch_chain(const TcArgs& a, int nwg, size_t smem, bool fast, hipStream_t st) {
  static const bool generic_only = [] { const char* e = getenv("ULTR_FUSED_GENERIC"); return e && e[0] == '1'; }();
  if (fast && !generic_only) {
    int maxk = 0;
    for (int j = 0; j <= a.H; ++j) maxk = a.K[j] > maxk ? a.K[j] : maxk;
    if (maxk <= 512) return launch_chain2<RIDE, true, 2>(a, nwg, smem, st);
    return launch_chain2<RIDE, true, 4>(a, nwg, smem, st);
  }
  return launch_chain2<RIDE, false, 4>(a, nwg, smem, st);
}

extern "C" int ultr_dnn_train_fused(const UltrDnnDesc* desc, const float* params, const float* table,
                                    const float* table_stats, int n_rows, int ldx, const int* docids, int B, int L,
                                    void* workspace, int wt_fresh, float* scores, float* grads, float* sumsq,
                                    void* stream) {
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
  g_pending_loss.kind = ULTR_RIDE_NONE;  // consumed, also when it does not match (ultr_hip.h): it must not outlive its buffers
  ULTR_REQUIRE(pl_scores == scores && pl.rows() == M, ULTR_EINVAL,
               "ultr_dnn_train_fused: the parked loss request is for another scores buffer or batch shape");
  float* ws = reinterpret_cast<float*>(workspace);
  const int has_ln = desc->norm == ULTR_NORM_LAYER;
  hipStream_t st = as_stream(stream);
  const int H = P.H;

  // ---- the streamed weight copies, unless the caller's last update call left them current: packed bf16 ...
  if (P.bf16 && !wt_fresh) {
    PackArgs t{};
    t.n = H;
    long long start = 0;
    for (int j = 0; j < H; ++j) {
      t.W[j] = params + desc->off_W[j];
      t.Wp[j] = reinterpret_cast<unsigned short*>(ws + P.wp[j]);
      t.WTp[j] = j >= 1 ? reinterpret_cast<unsigned short*>(ws + P.wtp[j]) : nullptr;
      t.K[j] = P.K[j]; t.ldn[j] = P.ldn[j];
      t.start[j] = start;
      start += (long long)P.K[j] * P.ldn[j] / 4;
    }
    t.start[H] = start;
    {
      ultr::ProfScope prof_("pack_weights_kernel", st);
      hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)cdiv(start, 256)), dim3(256), 0, st, t);
    }
    rc = check_launch("pack_weights_kernel");
    if (rc) return rc;
  }
  // ... or (fp32) transposed ones for the dgrad GEMMs
  if (!P.bf16 && H >= 2 && !wt_fresh) {
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
      a.WT[j] = j >= 1 && !P.bf16 ? ws + P.wt[j] : nullptr;
      a.Wp[j] = P.bf16 ? ws + P.wp[j] : nullptr;
      a.WTp[j] = P.bf16 && j >= 1 ? ws + P.wtp[j] : nullptr;
      a.act_off[j] = P.act_off[j]; a.act_ps[j] = P.act_ps[j];
    }
  }
  a.scores = scores;
  a.outpart = ws + P.outpart;
  a.PS = P.PS; a.PS2 = P.PS2; a.panel_floats = P.panel_floats; a.p1_off = P.p1_off; a.p2_off = P.p2_off; a.misc_off = P.misc_off;
#ifdef ULTR_STAMPS
  { const char* e = getenv("ULTR_TC_DBG"); a.dbg = e ? atoi(e) : 0; }
#endif
  a.dla = pl.dla; a.ipw = pl.ipw; a.rem = pl.rem; a.pd = pl.pd;
  a.list_weights = pl.kind == ULTR_RIDE_DLA ? pl.dla.list_weights : pl.kind == ULTR_RIDE_IPW ? pl.ipw.list_weights
                   : pl.kind == ULTR_RIDE_REM ? pl.rem.list_weights : nullptr;
  const bool fast = P.fast;
  if (P.rows == 32) {
    if (pl.kind == ULTR_RIDE_DLA) rc = launch_chain32_q<ULTR_RIDE_DLA>(a, P.nwg, P.smem_bytes, st);
    else if (pl.kind == ULTR_RIDE_IPW) rc = launch_chain32_q<ULTR_RIDE_IPW>(a, P.nwg, P.smem_bytes, st);
    else if (pl.kind == ULTR_RIDE_REM) rc = launch_chain32_q<ULTR_RIDE_REM>(a, P.nwg, P.smem_bytes, st);
    else rc = launch_chain32_q<ULTR_RIDE_PD>(a, P.nwg, P.smem_bytes, st);
  } else if (pl.kind == ULTR_RIDE_DLA) rc = launch_chain<ULTR_RIDE_DLA>(a, P.nwg, P.smem_bytes, fast, st);
  else if (pl.kind == ULTR_RIDE_IPW) rc = launch_chain<ULTR_RIDE_IPW>(a, P.nwg, P.smem_bytes, fast, st);
  else if (pl.kind == ULTR_RIDE_REM) rc = launch_chain<ULTR_RIDE_REM>(a, P.nwg, P.smem_bytes, fast, st);
  else rc = launch_chain<ULTR_RIDE_PD>(a, P.nwg, P.smem_bytes, fast, st);
  if (rc) return rc;
  if (P.tail_nwg > 0) {  // the lists beyond the last full round of 32-row workgroups
    TcArgs t = a;
    t.nl = P.tail_nl; t.list_base = P.tail_base; t.part_base = P.nwg;
    t.PS = P.tail.PS; t.PS2 = 0; t.panel_floats = P.tail.panel_floats;
    t.p1_off = P.tail.p1_off; t.p2_off = P.tail.p2_off; t.misc_off = P.tail.misc_off;
    for (int j = 0; j < H; ++j) { t.act_off[j] = P.tail.act_off[j]; t.act_ps[j] = P.tail.act_ps[j]; }
    if (pl.kind == ULTR_RIDE_DLA) rc = launch_chain<ULTR_RIDE_DLA>(t, P.tail_nwg, P.tail.smem_bytes, fast, st);
    else if (pl.kind == ULTR_RIDE_IPW) rc = launch_chain<ULTR_RIDE_IPW>(t, P.tail_nwg, P.tail.smem_bytes, fast, st);
    else if (pl.kind == ULTR_RIDE_REM) rc = launch_chain<ULTR_RIDE_REM>(t, P.tail_nwg, P.tail.smem_bytes, fast, st);
    else rc = launch_chain<ULTR_RIDE_PD>(t, P.tail_nwg, P.tail.smem_bytes, fast, st);
    if (rc) return rc;
  }

  // ---- every weight-gradient GEMM
  {
    WgradArgs layers[ULTR_MAX_LAYERS];
    for (int j = 0; j < H; ++j) {
      WgradArgs& q = layers[j];
      q.xhat = ws + P.xhat[j]; q.ldx = round4(P.K[j]);
      q.dz = ws + P.dz[j]; q.ldn = P.ldn[j]; q.N = P.N[j];
      q.K = P.K[j]; q.M = M; q.rows_per_split = P.rows_per_split;
      q.gslab = ws + P.gslab[j]; q.sslab = ws + P.sslab[j];
    }
    rc = launch_wgrad_all(H, layers, P.nsplit, P.bf16, st);
    if (rc) return rc;
  }

  // ---- slab reduction -> flat gradient, with the side jobs riding along
  FinalizeFusedArgs fr{};
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
  f.outpart = ws + P.outpart; f.out_blocks = P.nwg + P.tail_nwg; f.Kout = P.K[H];
  f.sumsq = sumsq;
  const unsigned fblocks = (unsigned)(start + (3 * P.K[H] + 1 + kFinOutCols - 1) / kFinOutCols);
  // too big to ride along, or lists of at most 8 (the rider is compiled for 16-lane groups): a launch of its own
  // (up to 512 rider workgroups = 8192 lists: beside the ~900 of the slab reduction they still fit the chip at once)
  constexpr unsigned kFusedMaxMetricRiders = 512;
  if (g_pending_metrics.on && (g_pending_metrics.nblocks > kFusedMaxMetricRiders || g_pending_metrics.a.P != 16)) {
    g_pending_metrics.on = false;
    rc = launch_metrics_now(g_pending_metrics.a, g_pending_metrics.nblocks, st);
    if (rc) return rc;
  }
  if (g_pending_metrics.on) {
    fr.m = g_pending_metrics.a; fr.m_blocks = (int)g_pending_metrics.nblocks; fr.m_pad = (fr.m_blocks + 7) & ~7;
    g_pending_metrics.on = false;
    // the rider leaves partial rows; the update launch (ultr_clip_adagrad*), next on this stream, adds them up
    rc = ultr_metrics_tail_flush(st);  // an earlier one nobody carried
    if (rc) return rc;
    const MetricsArgs& m = fr.m;
    const int n_met = ((m.metric_mask & ULTR_METRIC_MRR) ? 1 : 0) + ((m.metric_mask & ULTR_METRIC_NDCG) ? 1 : 0) +
                      ((m.metric_mask & ULTR_METRIC_ERR) ? 1 : 0);
    fr.m.defer_reduce = 1;
    g_pending_mtail.on = true;
    g_pending_mtail.t = MetricsTail{m.ws + 8, fr.m_blocks, n_met * m.n_topn * 2, m.B, m.list_weights ? 1 : 0, m.out,
                                    (m.tie_ctr && !m.u_pred && m.tie_seed != 0ull) ? m.tie_ctr : nullptr};
  }
  fr.ride = pl.kind; fr.dla = pl.dla; fr.ipw = pl.ipw; fr.rem = pl.rem; fr.pd = pl.pd;
  // the chain kernel has written the list weights: the loss rider must not race with the metrics that read them
  fr.dla.list_weights = nullptr; fr.ipw.list_weights = nullptr; fr.rem.list_weights = nullptr;
  fr.l_blocks = (int)pl.nblocks; fr.l_pad = (fr.l_blocks + 7) & ~7;
  // (a parked batch assembly stays parked: it rides with the update launch, ultr_clip_adagrad_fused -- this launch's
  // riders still read the current batch, which the assembly overwrites)
  {
    ultr::ProfScope prof_("finalize_fused_kernel", st);
    const dim3 grid(fblocks + (unsigned)(fr.m_pad + fr.l_pad));
    if (fr.ride == ULTR_RIDE_DLA) hipLaunchKernelGGL(finalize_fused_kernel<ULTR_RIDE_DLA>, grid, dim3(kThreads), 0, st, fr);
    else if (fr.ride == ULTR_RIDE_IPW) hipLaunchKernelGGL(finalize_fused_kernel<ULTR_RIDE_IPW>, grid, dim3(kThreads), 0, st, fr);
    else if (fr.ride == ULTR_RIDE_REM) hipLaunchKernelGGL(finalize_fused_kernel<ULTR_RIDE_REM>, grid, dim3(kThreads), 0, st, fr);
    else hipLaunchKernelGGL(finalize_fused_kernel<ULTR_RIDE_PD>, grid, dim3(kThreads), 0, st, fr);
  }
  rc = check_launch("finalize_fused_kernel");
  if (rc) g_pending_mtail.on = false;  // no rows were written: nothing for the update launch to add
  return rc;
}

extern "C" int ultr_clip_adagrad_fused(const UltrDnnDesc* desc, int B, int L, void* workspace, float* theta, float* acc,
                                       const float* grads, long long n, const float* denom, float scale_const, float l2,
                                       float clip_norm, float lr, int strategy, float* out_norm, float* ws, float* theta_s,
                                       float* acc_s, const float* grads_s, int n_s, const float* denom_s,
                                       float scale_const_s, float l2_s, float clip_norm_s, float lr_s, float* out_norm_s,
                                       const float* sumsq, int n_sumsq, long long n_covered, const float* publish_src,
                                       float* publish_dst, int publish_n, void* stream) {
  int rc = ultr_check_desc(desc, "ultr_clip_adagrad_fused");
  if (rc) return rc;
  const FusedPlan P = make_fused_plan(desc, B, L);
  ULTR_REQUIRE(P.ok && workspace, ULTR_EINVAL, "ultr_clip_adagrad_fused: shape not supported by the fused step, or no workspace");
  ULTR_REQUIRE(n >= desc->n_params && n > 4096, ULTR_EINVAL, "ultr_clip_adagrad_fused: theta must cover the scorer's parameters");
  SmallGroup r{};
  if (theta_s) {
    ULTR_REQUIRE(grads_s && n_s > 0 && n_s <= 4096, ULTR_EINVAL, "ultr_clip_adagrad_fused: small group needs 1..4096 parameters");
    ULTR_REQUIRE(strategy == ULTR_OPT_SGD || acc_s, ULTR_EINVAL, "ultr_clip_adagrad_fused: Adagrad needs an accumulator");
    r = SmallGroup{theta_s, acc_s, grads_s, n_s, denom_s, scale_const_s, l2_s, clip_norm_s, lr_s, strategy, out_norm_s};
  }
  TailOpts o{sumsq, n_sumsq, n_covered, publish_src, publish_dst, publish_n, TransposeOpts{}, AssembleArgs{}, 0, 0};
  if (!sumsq && n_sumsq == -1) { o.n_sumsq = 0; o.grid_norm = 1; }  // no partials: the update launch forms the norm itself
  // (a parked assembly of the next batch rides with the update: clip_adagrad_impl takes it once its checks have passed)
  float* wsf = reinterpret_cast<float*>(workspace);
  for (int j = P.bf16 ? 0 : 1; j < P.H; ++j) {
    const int t = o.tr.n++;
    o.tr.off[t] = desc->off_W[j]; o.tr.K[t] = P.K[j]; o.tr.N[t] = P.N[j]; o.tr.ldn[t] = P.ldn[j];
    o.tr.dst[t] = P.bf16 ? nullptr : wsf + P.wt[j];
    o.tr.pk[t] = P.bf16 ? reinterpret_cast<unsigned short*>(wsf + P.wp[j]) : nullptr;
    o.tr.pkt[t] = P.bf16 && j >= 1 ? reinterpret_cast<unsigned short*>(wsf + P.wtp[j]) : nullptr;
  }
  return clip_adagrad_impl(theta, acc, grads, n, denom, scale_const, l2, clip_norm, lr, strategy, out_norm, ws, r, o, stream, true);
}
