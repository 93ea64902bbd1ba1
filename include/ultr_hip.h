/*
 * ultr_hip.h -- C ABI of libultr_hip.so: the MI355X (gfx950) hot path of ULTRA's offline
 * unbiased-LTR training step, as hand-written HIP kernels.
 *
 * The reference (ULTR-Community/ULTRA) has no C/FFI boundary: its hot path is a TensorFlow-1
 * graph driven through Python plugin classes.  Each entry point below therefore cites the
 * reference Python/TF op sequence it replaces (file:line relative to the reference checkout);
 * the Python classes in ultra_amd/ mirror the reference's plugin API and bind these symbols
 * with ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _h (host);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 on success, negative ULTR_E* on error; ultr_last_error() gives the text;
 *   - no entry point allocates or frees device memory or synchronises, so calls are capturable
 *     in a hipGraph; two calls sharing one workspace must be ordered;
 *   - state between calls.  The plain entry points keep none beyond what the caller passes in
 *     (workspaces) and are re-entrant per stream.  Three families do keep state, all of it on the
 *     host side of the library:
 *       (1) the *_deferred calls (ultr_loss_*_deferred, ultr_metrics_deferred,
 *           ultr_batch_assemble_deferred, ultr_rem_update_propensity_deferred,
 *           ultr_pd_update_ratios_deferred -- and ultr_dnn_train_fused itself, which parks the last
 *           step of a metric evaluation it carried for the next ultr_clip_adagrad* call, see
 *           ultr_metrics_tail_flush) PARK one request per kind in a REQUEST CONTEXT: nothing is
 *           launched; a later carrying call made WITH THE SAME CONTEXT BOUND (ultr_dnn_backward*,
 *           ultr_dnn_train_fused, ultr_clip_adagrad_fused; each request's comment names its carriers)
 *           launches the parked work inside one of its own launches, on the CARRIER's stream.
 *           The context is the one bound to the calling host thread: ultr_pending_create() makes
 *           one, ultr_pending_bind(ctx) binds it to the calling thread (NULL = the thread's own
 *           default context, which is what a caller that never binds gets).  One context per model,
 *           bound before each group of calls made for that model, keeps models stepped alternately
 *           on one thread -- phases interleaved, feeds prefetching -- entirely apart: a request is
 *           only ever seen by calls made under its own context.  What follows:
 *           - every pointer of a parked request must stay valid until it has been carried, flushed
 *             (ultr_*_flush launches it on its own) or cancelled (ultr_*_cancel drops it);
 *           - parking a second request of a kind in one context launches the first on its own, at once;
 *           - a loss request is only differentiated in place by a carrier whose d_scores / scores
 *             buffer and B*L are the request's own; any other carrier launches it on its own first
 *             (ultr_dnn_train_fused consumes it and refuses).  Metric and assembly requests do not
 *             depend on the carrier's data and ride with whichever carrier of that context comes next;
 *           - a context is plain host memory and not locked: two threads must not use one context at
 *             the same time; a carrier under ANOTHER context (or another thread's default) does not see
 *             the request, and a carrier on another stream runs it on that stream, unordered against
 *             the stream the request's inputs were produced on;
 *           - the non-deferred twins of every call (ultr_loss_*_fwdbwd, ultr_metrics,
 *             ultr_batch_assemble) keep no state;
 *       (2) ultr_profile_* (process-wide event lists, off by default);
 *       (3) ultr_metrics_static (a per-device scratch array inside the library);
 *     and the launch-plan switches that remain (ULTR_* environment variables, DESIGN.md section 6:
 *     each selects a plan that the tests run) are read once per process, at first use;
 *   - "position-major" [L][B]: element (list b, rank l) is at l*B + b, i.e. the row order of
 *     DNN.build's axis-0 concat (DNN.py:65);
 *   - a doc id equal to n_rows denotes the zero pad row (base_algorithm.py:120-123).
 */
#ifndef ULTR_HIP_H_
#define ULTR_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define ULTR_VERSION 100 /* 0.1.0 */

#define ULTR_OK 0
#define ULTR_EINVAL (-1)  /* bad argument (shape, enum, null pointer)           */
#define ULTR_ELIMIT (-2)  /* size beyond a compiled-in limit                    */
#define ULTR_EHIP (-3)    /* a HIP runtime call failed (launch error)           */

#define ULTR_MAX_LAYERS 8
#define ULTR_MAX_LIST 64     /* max rank_list_size for the loss / feed kernels   */
#define ULTR_MAX_SORT 2048   /* max list length for the metric sort              */
#define ULTR_MAX_WIDTH 1024  /* max hidden width                                 */
#define ULTR_FEED_INFO_INTS 1088 /* int32 words of the feed kernels' `info` array  */

/* activation_func (base_ranking_model.py:59-65) */
enum { ULTR_ACT_ELU = 0, ULTR_ACT_RELU = 1, ULTR_ACT_SELU = 2, ULTR_ACT_TANH = 3, ULTR_ACT_SIGMOID = 4 };
/* norm hparam (DNN.py:32,70-84; base_ranking_model.py:67-70): "layer", "batch", or anything else = no normalisation.
 * BATCH: keras BatchNormalization called with training = true -- batch statistics over all M rows of the call
 * (biased variance, epsilon = ln_eps); the training passes use it.  BATCH_EVAL: training = false -- the moving
 * statistics, which the reference's training step never updates (mean 0, variance 1); forward only. */
enum { ULTR_NORM_NONE = 0, ULTR_NORM_LAYER = 1, ULTR_NORM_BATCH = 2, ULTR_NORM_BATCH_EVAL = 3 };
/* click models on the device path (click_models.py:74-116 PBM, :194-241 cascade, :119-191 user browsing model;
 * TABLE = any model whose click probability is a function of (relevance grade, rank) only, handed over as that
 * table: the trust-bias model :273-322 and the click-matrix model :244-270) */
enum { ULTR_CLICK_PBM = 0, ULTR_CLICK_CASCADE = 1, ULTR_CLICK_ORACLE = 2 /* oracle_mode: labels as clicks */,
       ULTR_CLICK_TABLE = 3, ULTR_CLICK_UBM = 4 };
/* loss_func hparam (dla.py:129-134, ipw_rank.py:135-143) */
enum { ULTR_LOSS_SOFTMAX = 0, ULTR_LOSS_SIGMOID = 1 };
/* logits_to_prob hparam (dla.py:99-101) */
enum { ULTR_PROB_SOFTMAX = 0, ULTR_PROB_SIGMOID = 1 };
/* grad_strategy hparam (dla.py:183-185) */
enum { ULTR_OPT_ADAGRAD = 0, ULTR_OPT_SGD = 1 };
/* compute_dtype hparam of the scorer (new; DNN.py:85-93 is the matmul it applies to).  F32: fp32 MFMA, the parity path.
 * BF16: opt-in fast path -- the forward, dgrad and weight-gradient GEMMs of the hidden layers multiply bf16-rounded
 * operands (activations and gradients rounded on the fly, weights from packed bf16 copies that the update launch keeps
 * current) and accumulate in fp32; LayerNorm, activations, losses, the output layer, the split-K reduction and the
 * optimiser stay fp32, what the step writes to memory stays fp32, and the parameters are fp32 master copies.
 * Stated error bound: DESIGN.md section 3. */
enum { ULTR_DTYPE_F32 = 0, ULTR_DTYPE_BF16 = 1 };
/* metric bits for ultr_metrics */
enum { ULTR_METRIC_MRR = 1, ULTR_METRIC_NDCG = 2, ULTR_METRIC_ERR = 4 };

/* Shape of the DNN scorer (DNN.py:25-34, :67-94) and where each variable lives in the flat
 * fp32 parameter buffer.  Layer j: LayerNorm(gamma_j, beta_j over its INPUT width) ->
 * W_j [in_j, sizes[j]] row-major -> + b_j -> activation (hidden layers only). */
typedef struct UltrDnnDesc {
  int n_layers;                 /* hidden layers + the output layer                  */
  int in_features;              /* F                                                 */
  int sizes[ULTR_MAX_LAYERS];   /* output width per layer; sizes[n_layers-1] must be 1 */
  int activation;               /* ULTR_ACT_*                                        */
  int norm;                     /* ULTR_NORM_*                                       */
  float ln_eps;                 /* keras LayerNormalization epsilon (1e-3)           */
  long long off_W[ULTR_MAX_LAYERS];     /* offsets in floats, each a multiple of 4   */
  long long off_b[ULTR_MAX_LAYERS];
  long long off_gamma[ULTR_MAX_LAYERS];
  long long off_beta[ULTR_MAX_LAYERS];
  long long n_params;           /* length of the flat buffer (padding included)      */
  int compute_dtype;            /* ULTR_DTYPE_*: arithmetic of the hidden layers' GEMMs in the fused training step
                                 * (ultr_dnn_train_fused); every other entry point computes in fp32 whatever it says */
} UltrDnnDesc;

int ultr_version(void);
const char* ultr_last_error(void);

/* Request contexts for the *_deferred calls (Conventions, "state between calls").  No counterpart in the
 * reference: there a tf.Session owns the graph's deferred work, one per model; here the caller owns a context
 * per model.  ultr_pending_create returns a new, empty context (NULL when out of host memory);
 * ultr_pending_bind makes ctx the calling thread's current context and returns the one bound before (NULL = the
 * thread's default context, also valid as an argument); ultr_pending_count returns how many requests are parked
 * in ctx (NULL = the calling thread's default context); ultr_pending_destroy drops the context together with
 * whatever is still parked in it (nothing is launched) and, if the calling thread had it bound, rebinds the
 * thread's default.  Host-only calls: no stream, no device work. */
void* ultr_pending_create(void);
void* ultr_pending_bind(void* ctx);
int ultr_pending_count(void* ctx);
void ultr_pending_destroy(void* ctx);

/* Opt-in per-kernel timing for bench.py's roofline line (no reference counterpart): when on, every
 * launch is bracketed by HIP events on its own stream.  ultr_profile_report synchronises and writes
 * "kernel_name count total_ms" lines into buf (host); returns the bytes written. */
int ultr_profile_enable(int on);
int ultr_profile_report(char* buf_h, int buflen);

/* ---- data set preparation -------------------------------------------------------------
 * Per-row LayerNorm statistics of the feature table: stats[r] = (mean, 1/sqrt(var+eps)) over
 * the F features of row r, for r < n_rows; stats[n_rows] = those of the zero pad row.
 * Layer 0 of DNN.build normalises the raw feature rows (DNN.py:78-81), and those statistics
 * depend only on the data, so they are computed once per table instead of every step.
 * x: [n_rows][ldx] fp32, ldx >= F, ldx % 4 == 0.  stats: [(n_rows+1)][2]. */
int ultr_ln_row_stats(const float* x, long long n_rows, int F, int ldx, float eps, float* stats,
                      void* stream);

/* ---- click simulation (click_models.py:86-116 PBM, :204-241 cascade) --------------------
 * labels [n][L] fp32 relevance grades -> clicks [n][L] fp32 in {0,1}.
 * click = [u < exam[min(rank,n_exam-1)] * click_prob[clamp(int(label))]] in float64.
 * uniforms != NULL: [n][L] float64 draws (bit-exact reproduction of a supplied stream);
 * uniforms == NULL: Philox4x32-10, counter (list, 0, block, stream_step), key = seed.
 * exam / click_prob are device float64 tables. */
int ultr_click_pbm(const float* labels, long long n, int L, const double* exam, int n_exam,
                   const double* click_prob, int n_click_prob, const double* uniforms,
                   unsigned long long seed, unsigned int stream_step, float* clicks, void* stream);
int ultr_click_cascade(const float* labels, long long n, int L, const double* exam, int n_exam,
                       const double* click_prob, int n_click_prob, const double* uniforms,
                       unsigned long long seed, unsigned int stream_step, float* clicks, void* stream);

/* Any device click model (sampleClicksForOneList of click_models.py:86-93, 144-155, 204-218, 258-266, 291-298).
 * Tables by model:
 *   PBM / CASCADE  exam [n_exam] per rank (ranks beyond -> last entry), click_prob [n_click_prob] per grade
 *   UBM            exam [n_exam][n_exam]: row = rank, column = (rank - rank of the last click above, -1 if none) - 1,
 *                  ranks >= n_exam read the last row as getExamProb does (:180-191); click_prob per grade
 *   TABLE          exam [n_click_prob][n_exam] = P(click | grade, rank), float64 products formed by the caller exactly
 *                  as the model's sampleClick forms them; click_prob is not read (pass exam)
 * click = [u < P] in float64; a uniform is consumed at every rank; cascade keeps only the first click. */
int ultr_click_sample(int click_model, const float* labels, long long n, int L, const double* exam, int n_exam,
                      const double* click_prob, int n_click_prob, const double* uniforms, unsigned long long seed,
                      unsigned int stream_step, float* clicks, void* stream);

/* ---- batch assembly: ClickSimulationFeed.get_batch (click_simulation_feed.py:107-162) ---
 * Rejection-samples B query lists with >= 1 click (when check_validation) from a data set
 * resident on the device and emits the feed tensors position-major.
 *   initial_list [Q][Lmax] int32 (doc row ids, -1 = pad), labels [Q][Lmax] fp32.
 *   out_docids [L][B] int32 (pad -> n_rows), out_clicks [L][B] fp32, out_list_idx [B] int32.
 * mode 0 (Philox, parallel): slot s = slot_base + b draws attempt t = 0.. on its own stream
 *   (counter (s, t, block, step)); draw 0 picks the list (int(u*Q), :132), draws 1..L are the
 *   clicks (:87).  After max_attempts rejected attempts the last candidate is kept.
 * mode 1 (sequential, bit-exact with a supplied random.random() stream): uniforms
 *   [n_cand][1+L] float64; candidates are consumed in order, accepted ones fill slots
 *   slot_start.. ; info[0] = candidates consumed, info[1] = slots filled so far.
 * step_ctr (mode 0, may be NULL): device uint32 added to `step` and incremented once per call by
 *   the last workgroup to finish -- kernel arguments are frozen in a hipGraph, this is not.
 * info: int32[ULTR_FEED_INFO_INTS] device, zero-initialised: {consumed, filled, unused, arrival counter}, then 32
 *   counters, each on a 128-byte line of its own (atomics on one line serialise at ~16 ns each whatever the word):
 *   info[64 + 32 k], k < 16: partial attempt counters (mode 0 diagnostic: their sum is the number of attempts made so
 *   far; B / attempts = acceptance rate); info[64 + 32 (16 + k)], k < 16: arrival sub-counters. */
int ultr_batch_assemble(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                        int click_model, const double* exam, int n_exam, const double* click_prob,
                        int n_click_prob, int mode, unsigned long long seed, unsigned int step,
                        unsigned int* step_ctr,
                        unsigned int slot_base, const double* uniforms, int n_cand, int slot_start,
                        int check_validation, int max_attempts, int n_rows, int* out_docids,
                        float* out_clicks, int* out_list_idx, int* info, void* stream);

/* ultr_batch_assemble (mode 0) with the launch deferred -- a prefetching feed: the request is parked and the NEXT
 * ultr_dnn_backward call under the same request context carries it as extra workgroups of its last launch, by which time
 * nothing reads the current batch any more, so the output buffers may be the ones the current step is using
 * (click_simulation_feed.py:107-162 assembles a batch per get_batch call; here the call for batch t+1 is made
 * during step t).  The buffers must stay valid until the request has run.  ultr_batch_assemble_flush launches a parked request on its own; so does any later
 * ultr_batch_assemble / ultr_batch_assemble_deferred call, so launch order is always call order. */
int ultr_batch_assemble_deferred(const int* initial_list, const float* labels, int Q, int Lmax, int L, int B,
                                 int click_model, const double* exam, int n_exam, const double* click_prob,
                                 int n_click_prob, unsigned long long seed, unsigned int step, unsigned int* step_ctr,
                                 unsigned int slot_base, int check_validation, int max_attempts, int n_rows,
                                 int* out_docids, float* out_clicks, int* out_list_idx, int* info, void* stream);
int ultr_batch_assemble_flush(void* stream);
/* Drops a parked request without running it (its output buffers are about to go away); returns 1 if there was one. */
int ultr_batch_assemble_cancel(void);

/* ---- online simulation: DeterministicOnlineSimulationFeed.simulate_clicks_online
 * (deterministic_online_simulation_feed.py:106-194) -----------------------------------------------
 * docids / labels / scores [Lmax][B] position-major: the sampled lists in their initial order (pad doc id = n_rows,
 * pad label 0), their true relevance labels, and the current model's scores for them.  Per list:
 *   list_len = 1 + last position holding a real document (:131-137);
 *   positions < list_len are re-ranked by score, descending, equal scores keeping their order (Python's stable
 *   sorted(reverse=True), :156-161); out_docids / the labels follow that order, positions >= list_len stay (:163-183);
 *   out_clicks: ranks < min(L, list_len) are sampled from the click model on the re-ranked labels (tables as for
 *   ultr_click_sample; ULTR_CLICK_ORACLE copies the labels, :170-171), other ranks < list_len are 0 (:186-187);
 *   check_validation: a list without a click is sampled again, at most max_rounds more times (:176-180).
 * uniforms != NULL: [B][L] float64, one round (check_validation must be 0); NULL: Philox4x32-10, counter
 *   (slot_base + b, round, block, step + *step_ctr), key = seed.  The caller advances *step_ctr (or step).
 * out_perm (may be NULL) [Lmax][B]: the initial position of the document now at each rank.
 * Limits: L <= ULTR_MAX_LIST, Lmax <= ULTR_MAX_SORT. */
int ultr_online_rerank_click(const int* docids, const float* labels, const float* scores, int B, int Lmax, int L,
                             int n_rows, int click_model, const double* exam, int n_exam, const double* click_prob,
                             int n_click_prob, const double* uniforms, unsigned long long seed, unsigned int step,
                             const unsigned int* step_ctr, unsigned int slot_base, int check_validation, int max_rounds,
                             int* out_docids, float* out_clicks, int* out_perm, void* stream);

/* The same with the ranking SAMPLED from the scores (StochasticOnlineSimulationFeed.simulate_clicks_online,
 * stochastic_online_simulation_feed.py:108-200): a Plackett-Luce draw, documents taken without replacement with
 * p ~ exp(tau * score) (:140-158), realised as a descending sort of tau * score + Gumbel noise (the same distribution;
 * np.random.choice's own stream is not reproduced, and documents whose probability underflows to 0 -- appended in
 * index order by :150-157 -- simply sort last).  rank_uniforms [B][Lmax] float64 in (0,1) or NULL (Philox, counter
 * (slot_base + b, 0xFFFFFFFF, block, step + *step_ctr)); noise = -log(-log(u)), keys in float64. */
int ultr_online_sample_click(float tau, const double* rank_uniforms, const int* docids, const float* labels,
                             const float* scores, int B, int Lmax, int L, int n_rows, int click_model,
                             const double* exam, int n_exam, const double* click_prob, int n_click_prob,
                             const double* uniforms, unsigned long long seed, unsigned int step,
                             const unsigned int* step_ctr, unsigned int slot_base, int check_validation, int max_rounds,
                             int* out_docids, float* out_clicks, int* out_perm, void* stream);

/* ---- DNN scorer (DNN.py:48-96 via base_algorithm.py:104-134) ---------------------------- */
/* Bytes of workspace ultr_dnn_forward/backward need for M rows (M = L*B). */
long long ultr_dnn_workspace_bytes(const UltrDnnDesc* desc, long long M);

/* scores[m] = DNN(table[docids[m]]) for m < M.  table [n_rows][ldx]; table_stats from
 * ultr_ln_row_stats (may be NULL when desc->norm == ULTR_NORM_NONE).  Keeps activations and
 * LayerNorm statistics in `workspace` for ultr_dnn_backward. */
int ultr_dnn_forward(const UltrDnnDesc* desc, const float* params, const float* table,
                     const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                     void* workspace, float* scores, void* stream);

/* grads[0..n_params) = d(sum_m d_scores[m] * scores[m]) / d params  (replaces tf.gradients,
 * dla.py:237-239 etc.).  Must follow ultr_dnn_forward with the same arguments/workspace. */
int ultr_dnn_backward(const UltrDnnDesc* desc, const float* params, const float* table,
                      const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                      void* workspace, const float* d_scores, float* grads, void* stream);

/* ultr_dnn_backward that also leaves the sum of squares of the gradient it wrote, as ultr_dnn_sumsq_count(desc)
 * partial sums (one per workgroup of its last launch; their total is ||grads[0..n_params)||^2).  ultr_clip_adagrad3
 * takes them in place of a norm pass over the gradient (tf.clip_by_global_norm, dla.py:240-245) -- valid as long as
 * the gradient is not changed in between, i.e. on a single GPU. */
int ultr_dnn_sumsq_count(const UltrDnnDesc* desc);
int ultr_dnn_backward_sumsq(const UltrDnnDesc* desc, const float* params, const float* table,
                            const float* table_stats, int n_rows, int ldx, const int* docids, long long M,
                            void* workspace, const float* d_scores, float* grads, float* sumsq, void* stream);

/* The fused training step for short lists (L <= 16): one call in place of ultr_dnn_forward + the loss + ultr_dnn_backward
 * (the reference runs all of them in ONE session.run: DNN.py:63-96, dla.py:119-180 + 237-239 and the other algorithms'
 * graphs).  A workgroup owns whole lists and carries them through forward, d_score, and the dgrad / LayerNorm-backward
 * chain with every activation in LDS; one launch then forms every layer's weight gradient, one reduces the partials.
 *   The loss comes from the request parked by ultr_loss_{dla,ipw,ipw_table,rem,pd}_deferred in the bound request context: its
 *   `scores` must be this call's `scores` and its B, L this call's (ULTR_EINVAL otherwise; the request is consumed).
 *   The chain kernel evaluates d_score with the loss kernels' own lane functions (same bits) and writes scores and the
 *   list weights; the request's statistics (and d_scores, prop_grads, pw_out) are produced by rider workgroups of the
 *   slab-reduction launch (the third of the step's launches).  A parked ultr_metrics_deferred request rides there too
 *   (lists of 9 to 16, at most 8192 lists; otherwise it gets a launch of its own), a parked
 *   ultr_batch_assemble_deferred request with the update launch (ultr_clip_adagrad_fused).
 *   Two row tiles: 16 rows per workgroup while the batch is at most one workgroup per CU (256 workgroups), 32 rows
 *   (chain32.hpp: the reference's default scorer -- elu, layer normalisation, hidden widths multiples of 16 -- in
 *   fp32, widths up to 768) beyond; ultr_dnn_fused_rows says which one a (desc, B, L) gets (0: not supported).
 *   A 32-row launch runs in rounds of 256 workgroups (one per CU); when the lists beyond its last full round fit one
 *   round of 16-row workgroups they run as a second launch of those instead of a last, mostly empty round
 *   (4096 lists of 10: 1280 x 3 lists, then 256 x 1).  ultr_dnn_fused_tail_lists: how many lists that launch takes (0:
 *   there is none; -1: bad descriptor).  Which kernel scores a list depends on (B, L) only; the two agree to rounding.
 *   grads[0..n_params) and sumsq (may be NULL) as ultr_dnn_backward_sumsq.  docids [L][B] position-major.
 * ultr_dnn_fused_workspace_bytes: bytes of workspace for (B, L); 0 when the shape is not supported (no hidden layer,
 * L > 16, widths whose 16-row panels do not fit the 160 KB of LDS) -- use the separate entry points then; -1 on a bad
 * descriptor. */
long long ultr_dnn_fused_workspace_bytes(const UltrDnnDesc* desc, int B, int L);
int ultr_dnn_fused_rows(const UltrDnnDesc* desc, int B, int L);
int ultr_dnn_fused_tail_lists(const UltrDnnDesc* desc, int B, int L);
/* workspace: zero-initialised by the caller before its first use and used for this (desc, B, L) only.
 * wt_fresh != 0: the transposed weight copies the workspace keeps for the dgrad GEMMs are current, i.e. the
 *   parameters have not changed since the last ultr_clip_adagrad_fused call on this workspace; 0: they are rebuilt
 *   first (one more small launch). */
int ultr_dnn_train_fused(const UltrDnnDesc* desc, const float* params, const float* table,
                         const float* table_stats, int n_rows, int ldx, const int* docids, int B, int L,
                         void* workspace, int wt_fresh, float* scores, float* grads, float* sumsq, void* stream);

/* ---- losses, forward + backward in one pass -------------------------------------------
 * All take scores/clicks position-major [L][B] and write d_scores [L][B] UN-NORMALISED: the
 * caller's optimiser call divides by the batch-global normaliser in `stats` (this is what
 * makes data-parallel runs exact: sums are all-reduced, then divided once).
 * `ws` is a float workspace of ultr_loss_workspace_floats(B, L) floats, zero-initialised once
 * (the kernels leave its arrival counter at zero). */
long long ultr_loss_workspace_floats(int B, int L);

/* softmax_loss with per-element weights (base_algorithm.py:228-251), IPWrank's default loss
 * (ipw_rank.py:141-143); pw [L][B].  loss_func SIGMOID = sigmoid_loss_on_list (:163-186).
 * stats[0] = sum_b loss_b (numerator), stats[1] = normaliser (sum wl, or B for sigmoid),
 * stats[2+l] = max_b pw[l][b]  ("Propensity weights %d" summary, ipw_rank.py:120-125).
 * list_weights [B] = mean_l(pw*clip(c,0,1))  (ipw_rank.py:180-181), may be NULL. */
int ultr_loss_ipw_fwdbwd(const float* scores, const float* clicks, const float* pw, int B, int L,
                         int loss_func, float* d_scores, float* stats, float* list_weights, float* ws,
                         void* stream);

/* IPW weights from a propensity table (propensity_estimator.py:33-42): pw = ipw[min(l,n-1)]
 * where clicks > 0 else 0.  ipw: device fp32 [n_ipw]. */
int ultr_ipw_weights(const float* clicks, int B, int L, const float* ipw, int n_ipw, float* pw,
                     void* stream);

/* DLA (dla.py:119-180, DenoisingNet :257-306, get_normalized_weights :343-364).
 * prop_params: [L+1] = W_0[0..L) then b_0.  Outputs:
 *   d_scores [L][B]   un-normalised d rank_loss / d train_output (divide by stats[1])
 *   prop_grads [L+1]  un-normalised d exam_loss / d (W_0, b_0)    (divide by stats[3])
 *   list_weights [B]  mean_l(pw_l * clip(c,0,1))                  (dla.py:207-208)
 *   stats: [0] rank numerator [1] rank normaliser [2] exam numerator [3] exam normaliser
 *          [4+l] pw_l   [4+L+l] sum_b rw[b][l]   (summaries dla.py:147-172) */
int ultr_loss_dla_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* prop_params,
                         int loss_func, int logits_to_prob, float max_propensity_weight, float* d_scores,
                         float* prop_grads, float* list_weights, float* stats, float* ws, void* stream);

/* ultr_loss_dla_fwdbwd with the launch deferred: same arguments and results, but for L <= 16 nothing is launched
 * here -- the request is parked and the NEXT ultr_dnn_backward / ultr_dnn_backward_sumsq call under the same request context
 * carries it inside its FIRST launch, provided that call's d_scores is this request's d_scores and its M equals
 * B * L: the loss kernel's workgroups ride along (statistics, tower gradient, list weights, d_scores), and the
 * backward workgroups evaluate the d_score of their own rows from scores / clicks / prop_params instead of reading
 * it, so the step's chain of dependent launches is one shorter (dla.py:119-180 is a sub-graph of the same
 * session.run as tf.gradients, :237-239).  All pointers must stay valid until that call.  Otherwise (longer lists,
 * another d_scores buffer, no hidden layer wider than 1024) the request is launched on its own, first.
 * ultr_loss_flush launches a parked request on its own (no-op when there is none). */
int ultr_loss_dla_deferred(const float* scores, const float* clicks, int B, int L, const float* prop_params,
                           int loss_func, int logits_to_prob, float max_propensity_weight, float* d_scores,
                           float* prop_grads, float* list_weights, float* stats, float* ws, void* stream);
int ultr_loss_flush(void* stream);
/* Drop a parked loss request without running it; returns 1 if there was one. */
int ultr_loss_cancel(void);
/* The same for the IPW / naive loss (ultr_loss_ipw_fwdbwd: pw must be ready when the backward pass runs), the pairwise
 * debiasing loss (ultr_loss_pd_fwdbwd) and the regression-EM loss (ultr_loss_rem_fwdbwd: its d_score is elementwise, sigmoid(score + b) - Bernoulli label, so the
 * backward workgroups draw the label of each of their rows themselves, from the same Philox counter or the same
 * supplied uniforms).  At most one request is parked per request context; parking a second one launches the first. */
int ultr_loss_ipw_deferred(const float* scores, const float* clicks, const float* pw, int B, int L, int loss_func,
                           float* d_scores, float* stats, float* list_weights, float* ws, void* stream);
/* ultr_ipw_weights + ultr_loss_ipw_deferred in one request: the weights are formed from the clicks and the estimator's
 * table (propensity_estimator.py:33-42) inside the carrying launch and also written to pw_out [L][B] (the
 * propensity_weights{l} feed entries of ipw_rank.py:116-119, 214-223). */
int ultr_loss_ipw_table_deferred(const float* scores, const float* clicks, const float* ipw, int n_ipw, float* pw_out,
                                 int B, int L, int loss_func, float* d_scores, float* stats, float* list_weights,
                                 float* ws, void* stream);
int ultr_loss_pd_deferred(const float* scores, const float* clicks, int B, int L, const float* t_plus,
                          const float* t_minus, float* d_scores, float* stats, float* ws, void* stream);
int ultr_loss_rem_deferred(const float* scores, const float* clicks, int B, int L, const float* propensity,
                           const float* sigmoid_prob_b, const float* uniforms, unsigned long long seed,
                           unsigned int step, const unsigned int* step_ctr, unsigned int slot_base,
                           float* d_scores, float* ranker_labels, float* list_weights,
                           float* stats, float* d_bias, float* ws, void* stream);

/* Regression-EM (regression_EM.py:116-176).  propensity [L] is the EM state (read, pre-update
 * value used for loss and gradient, q16).  uniforms [L][B] fp32 or NULL (Philox(seed, step)).
 *   d_scores un-normalised (divide by B_global); d_bias[0] (may be NULL) = sum of d_scores, the
 *   un-normalised gradient of sigmoid_prob_b;
 *   stats: [0] sum_b loss_b  [1] B  [2+l] sum_b (c + (1-c) P(e=1,r=0|c=0))  (M-step numerator)
 *          [2+L+l] sum_b ranker_label  [2+2L+l] sum_b clicks   ("Additional pseudo clicks")
 *          [2+3L] sum of d_scores = un-normalised gradient of sigmoid_prob_b
 *   Philox draws are keyed by (slot_base + b, l), so a data-parallel shard reproduces its
 *   slice of the one-big-batch run.
 *   list_weights [B] = mean_l(clip(c,0,1)/propensity_l)  (regression_EM.py:176,216-217). */
int ultr_loss_rem_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* propensity,
                         const float* sigmoid_prob_b, const float* uniforms, unsigned long long seed,
                         unsigned int step, const unsigned int* step_ctr, unsigned int slot_base,
                         float* d_scores, float* ranker_labels,
                         float* list_weights,
                         float* stats, float* d_bias, float* ws, void* stream);

/* EM state update from all-reduced statistics: p <- (1-s) p + s * stats[2+l]/stats[1]. */
int ultr_rem_update_propensity(float* propensity, int L, const float* stats, float em_step_size,
                               float* old_out /* [L] pre-update values, may be NULL */, void* stream);

/* Pairwise debiasing (pairwise_debias.py:139-171, pair CE base_algorithm.py:151-161).
 * t_plus/t_minus [L] state (pre-update values used).  d_scores is already final (the loss is a
 * batch SUM).  stats: [0..L*L) pair-loss matrix l_ij (sum_b), to be all-reduced before
 * ultr_pd_update_ratios; stats[L*L] = loss. */
int ultr_loss_pd_fwdbwd(const float* scores, const float* clicks, int B, int L, const float* t_plus,
                        const float* t_minus, float* d_scores, float* stats, float* ws, void* stream);
int ultr_pd_update_ratios(float* t_plus, float* t_minus, int L, float* stats, float em_step_size,
                          float regulation_p, float* old_out /* [2L] pre-update t+,t-; may be NULL */,
                          void* stream);

/* The same two EM-state updates (regression_EM.py:151-155, pairwise_debias.py:162-171), parked in
 * the bound request context instead of launched: the next ultr_clip_adagrad* call under it carries
 * the request in its update launch (workgroup 0 applies it before it publishes the step's row; a
 * parameter group of <= 4096 floats launches it first instead).  The pointers must stay valid
 * until that call.  A second request, or ultr_em_update_flush, launches a parked one on `stream`.
 * Pair lists too long for the riding body (PD with L > 16) run at once, as the call above. */
int ultr_rem_update_propensity_deferred(float* propensity, int L, const float* stats, float em_step_size,
                                        float* old_out, void* stream);
int ultr_pd_update_ratios_deferred(float* t_plus, float* t_minus, int L, float* stats, float em_step_size,
                                   float regulation_p, float* old_out, void* stream);
int ultr_em_update_flush(void* stream);

/* The fused training step (ultr_dnn_train_fused) carries a parked metric evaluation as rider workgroups that stop at
 * their partial rows; the column sums, the division by B and the tie counter's bump are parked in the bound request
 * context for the next ultr_clip_adagrad* call, whose update launch performs them before it publishes (a parameter group of
 * <= 4096 floats: a one-workgroup launch first).  The metric outputs are therefore valid after that call, or after
 * ultr_metrics_tail_flush, which performs a parked reduction on `stream` at once (no-op when nothing is parked). */
int ultr_metrics_tail_flush(void* stream);

/* ---- metrics (metrics.py:215-323,445-481; sort metric_utils.py:49-86) --------------------
 * scores/labels [Lx][B] position-major.  docids (may be NULL) [Lx][B]: entries == n_rows get
 * PADDING_SCORE first (base_algorithm.py:64-87).  list_weights [B] or NULL (per-list weights).
 * u_pred [B][Lx] fp32 tie-break uniforms or NULL (ties then fall back to lower index first).
 * out[(m*n_topn + t)*2 + w]: m in order MRR, NDCG, ERR (those selected in metric_mask),
 * w = 0 plain / 1 weighted (only written when list_weights != NULL).  topn_h: host ints.
 * ws: ultr_metrics_workspace_floats(B) floats, zero-initialised once. */
long long ultr_metrics_workspace_floats(int B, int n_topn);
int ultr_metrics(const float* scores, const float* labels, const int* docids, int n_rows,
                 const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                 int n_topn, int metric_mask, float max_label, float* out, float* ws, void* stream);

/* ultr_metrics with the launch deferred: same arguments and results, but for lists of at most 64 entries nothing
 * is launched here -- the request is parked and the NEXT ultr_dnn_backward call under the same request context carries it as
 * extra workgroups of its largest weight-gradient launch (the in-step metrics of dla.py:189-221 need only the
 * scores and the list weights, which exist before the backward pass; riding along takes their latency out of the
 * chain of dependent launches).  All pointers must stay valid until that call.  Longer lists are evaluated at once.
 * ultr_metrics_flush launches a parked request on its own (no-op when there is none). */
int ultr_metrics_deferred(const float* scores, const float* labels, const int* docids, int n_rows,
                 const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                 int n_topn, int metric_mask, float max_label, float* out, float* ws, void* stream);
/* ultr_metrics / ultr_metrics_deferred with the tie-break uniforms drawn inside the kernel instead of supplied:
 * sort_by_scores shuffles ties with fresh uniforms on every call (metric_utils.py:49-86, shuffle_ties=True, the default
 * of every metric in metrics.py).  Philox4x32-10 keyed by tie_seed (!= 0), counter (tie_slot_base + list, 'TIE',
 * rank / 4, tie_step + *tie_ctr); tie_ctr (device, may be NULL) is bumped by the launch's last workgroup, so a captured
 * graph draws new uniforms on every replay; tie_slot_base = rank * B makes N data-parallel ranks draw what one rank with
 * the whole batch would.  The uniforms only matter where scores tie exactly (pad documents, constant initialisers). */
int ultr_metrics_ties(const float* scores, const float* labels, const int* docids, int n_rows,
                      const float* list_weights, int B, int Lx, const int* topn_h, int n_topn, int metric_mask,
                      float max_label, unsigned long long tie_seed, unsigned int tie_step, unsigned int* tie_ctr,
                      unsigned int tie_slot_base, float* out, float* ws, void* stream);
int ultr_metrics_ties_deferred(const float* scores, const float* labels, const int* docids, int n_rows,
                               const float* list_weights, int B, int Lx, const int* topn_h, int n_topn, int metric_mask,
                               float max_label, unsigned long long tie_seed, unsigned int tie_step, unsigned int* tie_ctr,
                               unsigned int tie_slot_base, float* out, float* ws, void* stream);
int ultr_metrics_flush(void* stream);
/* Drop a parked request without running it (the scores it describes never came to be); returns 1 if there was one. */
int ultr_metrics_cancel(void);
/* ultr_metrics for callers that cannot allocate and own no workspace.  It has no counterpart in the reference: it is
 * for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments and
 * results as ultr_metrics minus `ws`: the partial rows and the arrival counter live in a device array inside the
 * library (ultr_metrics_static_max_floats() floats = 1 MiB, held by every GPU the library runs on until the process
 * ends); its address is looked up once per device.  The call makes no hipMalloc / hipFree and no allocating library
 * call.  ULTR_ELIMIT when ultr_metrics_workspace_floats(B, n_topn) exceeds the array.  Every call on a GPU shares
 * that GPU's array: two calls that overlap on one GPU (different streams, different threads) race -- serialise them
 * or use ultr_metrics with workspaces of your own.  The same holds for captured graphs: two graphs that each contain
 * the call must not be replayed concurrently on one GPU.
 * Graph capture: the FIRST call on a device resolves the array's address (hipGetSymbolAddress), which must happen
 * outside a capture -- call ultr_metrics_static_prepare() once per device (with that device current) at set-up, or
 * make one eager call first; a first call on a capturing stream returns ULTR_EINVAL and launches nothing.  After
 * that the call takes no lock and makes no runtime call besides hipMemsetAsync (32 bytes, captured as a memset node)
 * and the kernel launch. */
long long ultr_metrics_static_max_floats(void);
/* Set-up for ultr_metrics_static on the current device (idempotent, thread-safe); launches nothing. */
int ultr_metrics_static_prepare(void);
int ultr_metrics_static(const float* scores, const float* labels, const int* docids, int n_rows,
                        const float* list_weights, const float* u_pred, int B, int Lx, const int* topn_h,
                        int n_topn, int metric_mask, float max_label, float* out, void* stream);
/* The permutation itself: perm [B][topn] int32 = sort_by_scores(scores) indices. */
int ultr_sort_by_scores(const float* scores, const float* u, int B, int Lx, int topn, int* perm,
                        void* stream);

/* ---- optimiser (tf.clip_by_global_norm + Adagrad/SGD, dla.py:240-255 etc.) ---------------
 * g = grads * scale_const / (*denom or 1) + l2 * theta;  norm = ||g||_2;
 * if clip_norm > 0: g *= clip_norm / max(norm, clip_norm);
 * Adagrad: acc += g^2; theta -= lr * g / sqrt(acc)  (acc starts at 0.1, no epsilon); SGD:
 * theta -= lr*g.  out_norm (device float, may be NULL) receives the pre-clip norm.
 * ws: at least 2048 floats, zero-initialised once (1024 + 8 without the n_sumsq == -1 option of ultr_clip_adagrad3). */
int ultr_clip_adagrad(float* theta, float* acc, const float* grads, long long n, const float* denom,
                      float scale_const, float l2, float clip_norm, float lr, int strategy,
                      float* out_norm, float* ws, void* stream);

/* The same, plus a second, small parameter group (n_s <= 4096) with its own normaliser, clip norm and
 * learning rate, updated inside the first launch: DLA updates the ranker and the propensity tower with two
 * independent clip_by_global_norm + optimiser pairs (dla.py:240-251), which would otherwise cost a third
 * kernel per step for 11 floats.  The main group must have more than 4096 parameters. */
int ultr_clip_adagrad2(float* theta, float* acc, const float* grads, long long n, const float* denom,
                       float scale_const, float l2, float clip_norm, float lr, int strategy, float* out_norm,
                       float* ws, float* theta_s, float* acc_s, const float* grads_s, int n_s,
                       const float* denom_s, float scale_const_s, float l2_s, float clip_norm_s, float lr_s,
                       float* out_norm_s, void* stream);

/* The fused tail of a training step: ultr_clip_adagrad2 (small group optional: theta_s may be NULL) plus two
 * options, each of which removes one launch from the chain of dependent kernels.
 *   sumsq != NULL (and l2 == 0): the norm is formed from n_sumsq partial sums of squares of grads[0..n_covered)
 *     (ultr_dnn_backward_sumsq) plus the squares of grads[n_covered..n), n - n_covered <= 65536, instead of a
 *     pass over the gradient; the small group then rides with the update launch.
 *   sumsq == NULL and n_sumsq == -1: no partials exist (data parallel: the gradient is final only after the
 *     collective), and the update launch forms the norm ITSELF -- every update workgroup leaves the partial of its own
 *     elements in ws, arrives at a counter (two levels: ws[1040 + 32 k] and ws[0], which only ever grow: do not clear ws between calls) and waits
 *     for the launch's other update workgroups (at most 1024, co-resident) -- instead of a norm launch before it.
 *     With sumsq != NULL and l2 != 0 the same happens (the partials describe the raw gradient only).
 *   publish_n > 0: workgroup 0 copies publish_n floats publish_src -> publish_dst, with out_norm / out_norm_s
 *     taken fresh when they lie inside the source range.  publish_dst may be device memory or the device-side
 *     address of pinned host memory (ultr_host_device_pointer): the step's results row reaches the host without
 *     a copy of its own (the reference returns loss and summary from session.run, dla.py:308-341). */
int ultr_clip_adagrad3(float* theta, float* acc, const float* grads, long long n, const float* denom,
                       float scale_const, float l2, float clip_norm, float lr, int strategy, float* out_norm,
                       float* ws, float* theta_s, float* acc_s, const float* grads_s, int n_s,
                       const float* denom_s, float scale_const_s, float l2_s, float clip_norm_s, float lr_s,
                       float* out_norm_s, const float* sumsq, int n_sumsq, long long n_covered,
                       const float* publish_src, float* publish_dst, int publish_n, void* stream);

/* ultr_clip_adagrad3 for a scorer that trains through ultr_dnn_train_fused: the update launch also writes the updated
 * hidden-layer weight matrices, transposed, into the fused step's workspace (same desc, B, L), so the next
 * ultr_dnn_train_fused call may pass wt_fresh = 1.  theta must start with the scorer's flat parameter buffer. */
int ultr_clip_adagrad_fused(const UltrDnnDesc* desc, int B, int L, void* workspace, float* theta, float* acc,
                            const float* grads, long long n, const float* denom, float scale_const, float l2,
                            float clip_norm, float lr, int strategy, float* out_norm, float* ws, float* theta_s,
                            float* acc_s, const float* grads_s, int n_s, const float* denom_s, float scale_const_s,
                            float l2_s, float clip_norm_s, float lr_s, float* out_norm_s, const float* sumsq, int n_sumsq,
                            long long n_covered, const float* publish_src, float* publish_dst, int publish_n,
                            void* stream);

/* Device-side address of a pinned (page-locked, mapped) host allocation; ULTR_EINVAL if the memory is not mapped. */
int ultr_host_device_pointer(void* host, void** device);

#ifdef __cplusplus
}
#endif
#endif /* ULTR_HIP_H_ */
