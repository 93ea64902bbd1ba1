"""CPU oracle for the ULTRA training-step hot path -- TEST INFRASTRUCTURE ONLY.

This module is a NumPy (float64 by default) restatement of the arithmetic that
the reference performs for one training step.  It is the checker used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing under ``ultra_amd/`` (the product) may import it: the
product path runs on hand-written HIP kernels and fails loudly without them.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  TensorFlow 1.x is not installable here, so TF-op semantics
are restated from TF's documented behaviour (SURVEY.md section 8c lists them);
the pure-Python parts (click models, hparams, propensity lookup) are pinned by
golden vectors generated from the reference modules themselves
(``tests/golden/make_golden.py``).  The TF-graph parts have no golden vectors in
the reference (its only test asserts that checkpoint files exist), so for them
parity is pinned by two independent derivations that must agree: the closed-form
gradients below and the autograd restatement in ``oracle/torch_graph.py``.
"""
import math

import numpy as np

# --------------------------------------------------------------------------
# Click models                                   ultra/utils/click_models.py
# --------------------------------------------------------------------------

PBM_ORIGINAL_EXAM_PROB = [0.68, 0.61, 0.48, 0.34, 0.28, 0.20, 0.11, 0.10, 0.08, 0.06]  # click_models.py:82-83


def click_prob_table(neg_click_prob, pos_click_prob, relevance_grading_num):
    """click_models.py:51-57 (setClickProb)."""
    b = (pos_click_prob - neg_click_prob) / (pow(2, relevance_grading_num) - 1)
    a = neg_click_prob - b
    return [a + pow(2, i) * b for i in range(relevance_grading_num + 1)]


def pbm_exam_table(eta):
    """click_models.py:80-84 (PositionBiasedModel.setExamProb)."""
    return [pow(x, eta) for x in PBM_ORIGINAL_EXAM_PROB]


def _clamp_label(label, n):
    """click_models.py:108-111: int() truncation, negatives -> 0, >= len -> last."""
    rel = int(label) if label > 0 else 0
    return rel if rel < n else n - 1


def ubm_exam_prob(exam_prob, rank, last_click_rank):
    """UserBrowsingModel.getExamProb, click_models.py:180-191."""
    distance = rank - last_click_rank
    if rank < len(exam_prob):
        return exam_prob[rank][distance - 1]
    if distance > rank:
        return exam_prob[-1][-1]
    return exam_prob[-1][distance - 1 if distance < len(exam_prob[-1]) - 1 else -2]


def trust_bias_click_prob(click_prob, extra, rank, rel):
    """TrustBiasedModel.getClickProb, click_models.py:316-322."""
    primitive = click_prob[rel]
    ep_plus = 1 - (rank + 2) / 100
    ep_minus = extra['epsilon_1_neg'] / (rank + 1)
    return primitive * ep_plus + (1 - primitive) * ep_minus


def sample_clicks_one_list(model_name, click_prob, exam_prob, label_list, uniforms, extra=None):
    """One call of sampleClicksForOneList with the uniform draws supplied.

    PBM: click_models.py:86-116.  Cascade: click_models.py:204-241 (a uniform is
    consumed at every rank, the output is zeroed after the first click).  User browsing
    model: :144-191 (examination depends on the last click above).  Trust bias: :291-322.
    Click matrix: :258-266 (click_prob[grade][rank] is the probability itself).
    ``uniforms[r]`` stands for the r-th ``random.random()`` of the call.
    """
    clicks = []
    has_click = False
    last_click_rank = -1
    for rank, label in enumerate(label_list):
        if model_name == 'click_matrix_model':
            rel = int(label) if label > 0 else 0
            clicks.append(1 if uniforms[rank] < click_prob[rel][rank] else 0)
            continue
        rel = _clamp_label(label, len(click_prob))
        if model_name == 'user_browsing_model':
            exam_p = ubm_exam_prob(exam_prob, rank, last_click_rank)
        else:
            exam_p = exam_prob[rank if rank < len(exam_prob) else -1]
        click_p = trust_bias_click_prob(click_prob, extra, rank, rel) if model_name == 'trust_biased_model' \
            else click_prob[rel]
        click = 1 if uniforms[rank] < exam_p * click_p else 0
        if model_name == 'cascade_model':
            clicks.append(0 if has_click else click)
            if click > 0:
                has_click = True
        elif model_name in ('position_biased_model', 'trust_biased_model'):
            clicks.append(click)
        elif model_name == 'user_browsing_model':
            clicks.append(click)
            if click > 0:
                last_click_rank = rank
        else:
            raise ValueError('click model %s is outside the hot path' % model_name)
    return clicks


def get_batch_sequential(initial_list, labels, rank_list_size, batch_size, model_name,
                         click_prob, exam_prob, uniforms, check_validation=True, extra=None):
    """click_simulation_feed.py:107-143 with the Python RNG stream supplied.

    ``uniforms`` is a flat float64 sequence standing for successive
    ``random.random()`` results.  Per candidate: one draw for the list index
    (:132) then ``rank_list_size`` draws for the clicks (:87); rejected
    candidates (no click, :95-97) still consume their draws.

    Returns (rank_list_idxs, clicks [B][L] ints, n_candidates_consumed).
    """
    length = len(initial_list)
    L = rank_list_size
    pos = 0
    idxs, clicks = [], []
    n_cand = 0
    while len(idxs) < batch_size:
        i = int(uniforms[pos] * length)
        label_list = [0 if initial_list[i][x] < 0 else labels[i][x] for x in range(L)]  # :80-82
        c = sample_clicks_one_list(model_name, click_prob, exam_prob, label_list,
                                   uniforms[pos + 1:pos + 1 + L], extra)
        pos += 1 + L
        n_cand += 1
        if check_validation and sum(c) == 0:
            continue
        idxs.append(i)
        clicks.append(c)
    return idxs, clicks, n_cand


# ---- Philox4x32-10 (the device feed's counter-based stream; new, not in the reference)

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = 0x9E3779B9
_PHILOX_W1 = 0xBB67AE85


def philox4x32(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11).  ctr: [...,4] uint32, key: (k0,k1)."""
    c = np.array(ctr, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & mask, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & mask, lo0
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def u53(a, b):
    """Two uint32 words -> double in [0,1) exactly as MT19937's genrand_res53."""
    return ((a >> np.uint32(5)).astype(np.float64) * 67108864.0 +
            (b >> np.uint32(6)).astype(np.float64)) / 9007199254740992.0


def philox_uniforms(seed, step, slot, attempt, n):
    """The device feed's draw spec: draw d of (slot, attempt) in step ``step``.

    counter = (slot, attempt, block, step), key = (seed_lo, seed_hi); block b
    yields words 4b..4b+3; draw d uses words (2d, 2d+1) through ``u53``.
    """
    nblk = (2 * n + 3) // 4
    ctr = np.zeros((nblk, 4), dtype=np.uint32)
    ctr[:, 0] = slot
    ctr[:, 1] = attempt
    ctr[:, 2] = np.arange(nblk)
    ctr[:, 3] = step
    words = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)
    return u53(words[0:2 * n:2], words[1:2 * n:2])


def tie_uniforms(seed, step, slot_base, n_lists, list_len):
    """The metric kernels' in-kernel tie-break uniforms (new; stands in for the fresh tf.random_uniform of
    metric_utils.py:76-81): list b, rank r draws word r % 4 of Philox block counter (slot_base + b, 'TIE', r // 4, step)
    under key ``seed``, as a 24-bit float32 in [0, 1).  -> float32 [n_lists, list_len]."""
    nblk = (list_len + 3) // 4
    ctr = np.zeros((n_lists, nblk, 4), dtype=np.uint32)
    ctr[:, :, 0] = (slot_base + np.arange(n_lists, dtype=np.uint64))[:, None] & 0xFFFFFFFF
    ctr[:, :, 1] = 0x544945
    ctr[:, :, 2] = np.arange(nblk)[None, :]
    ctr[:, :, 3] = step
    words = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(n_lists, nblk * 4)[:, :list_len]
    return ((words >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def get_batch_slots(initial_list, labels, rank_list_size, batch_size, model_name, click_prob,
                    exam_prob, seed, step, slot_base=0, max_attempts=100, check_validation=True, extra=None):
    """Per-slot rejection sampling: slot s retries with attempt t = 0,1,.. on its own stream.

    Distributionally identical to click_simulation_feed.py:131-137 (i.i.d.
    rejection sampling) but order-free, so B lists sample in parallel and a
    DP run with B/N lists per rank equals one big batch bit for bit.
    """
    length = len(initial_list)
    L = rank_list_size
    idxs, clicks, attempts = [], [], []
    for s in range(batch_size):
        for t in range(max_attempts):
            u = philox_uniforms(seed, step, slot_base + s, t, 1 + L)
            i = int(u[0] * length)
            label_list = [0 if initial_list[i][x] < 0 else labels[i][x] for x in range(L)]
            c = sample_clicks_one_list(model_name, click_prob, exam_prob, label_list, u[1:], extra)
            if (not check_validation) or sum(c) > 0 or t == max_attempts - 1:
                break
        idxs.append(i)
        clicks.append(c)
        attempts.append(t + 1)
    return idxs, clicks, attempts


def online_rerank_click(docids, labels, scores, n_rows, rank_list_size, model_name, click_prob, exam_prob,
                        uniforms, extra=None, oracle_mode=False, tau=None, rank_uniforms=None):
    """deterministic_online_simulation_feed.py:129-187 for one batch, with the click draws supplied.

    docids / labels / scores are list-major [B][Lmax]; ``uniforms[i]`` stands for the ``random.random()`` results of
    list i's (single) sampleClicksForOneList call.  Returns (docids, clicks, order), list-major.
    """
    B, Lmax = len(docids), len(docids[0])
    out_docids = [list(row) for row in docids]
    out_clicks = [list(row) for row in labels]
    orders = []
    for i in range(B):
        valid_idx = Lmax - 1                                                   # :131-137
        while valid_idx > -1:
            if docids[i][valid_idx] < n_rows:
                break
            valid_idx -= 1
        list_len = valid_idx + 1
        s = [scores[i][k] for k in range(list_len)]
        if tau is not None:
            # stochastic_online_simulation_feed.py:140-158 draws the ranking from the Plackett-Luce model
            # p ~ exp(tau * score) with np.random.choice; the same distribution as a sort of Gumbel-perturbed scores
            # (the device formulation; the supplied uniforms stand for its noise draws)
            s = [float(tau) * float(s[k]) - math.log(-math.log(max(float(rank_uniforms[i][k]), 2.0 ** -60)))
                 for k in range(list_len)]
        rerank_list = sorted(range(len(s)), key=lambda k: s[k], reverse=True)  # :156-161 (stable)
        new_label_list = [labels[i][rerank_list[j]] for j in range(list_len)]
        if oracle_mode:
            click_list = new_label_list[:rank_list_size]                       # :170-171
        else:
            click_list = sample_clicks_one_list(model_name, click_prob, exam_prob, new_label_list[:rank_list_size],
                                                uniforms[i], extra)             # :173-174
        for j in range(list_len):                                              # :182-187
            out_docids[i][j] = docids[i][rerank_list[j]]
            out_clicks[i][j] = click_list[j] if j < rank_list_size else 0
        orders.append(rerank_list + list(range(list_len, Lmax)))
    return out_docids, out_clicks, orders


def propensity_for_one_list(ipw_list, click_list):
    """propensity_estimator.py:33-42 (getPropensityForOneList, clicked ranks only)."""
    return [(ipw_list[r] if r < len(ipw_list) else ipw_list[-1]) if click_list[r] > 0 else 0.0
            for r in range(len(click_list))]


# --------------------------------------------------------------------------
# DNN scorer                                       ultra/ranking_model/DNN.py
# --------------------------------------------------------------------------

LN_EPS = 1e-3  # tf.keras.layers.LayerNormalization default epsilon (DNN.py:73 passes none)
SELU_ALPHA = 1.6732632423543772848170429916717  # base_ranking_model.py:24-25
SELU_SCALE = 1.0507009873554804934193349852946


def act_fwd(name, z):
    """base_ranking_model.py:59-65 (ACT_FUNC_DIC); selu :13-26."""
    if name == 'elu':
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    if name == 'relu':
        return np.maximum(z, 0)
    if name == 'selu':
        return SELU_SCALE * np.where(z >= 0, z, SELU_ALPHA * np.expm1(np.minimum(z, 0)))
    if name == 'tanh':
        return np.tanh(z)
    if name == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-z))
    raise KeyError(name)


def act_grad_from_output(name, a):
    """d act / d z expressed through the activation OUTPUT a (what the device stores)."""
    if name == 'elu':
        return np.where(a > 0, 1.0, a + 1.0)
    if name == 'relu':
        return (a > 0).astype(a.dtype)
    if name == 'selu':
        return np.where(a > 0, SELU_SCALE, a + SELU_SCALE * SELU_ALPHA)
    if name == 'tanh':
        return 1.0 - a * a
    if name == 'sigmoid':
        return a * (1.0 - a)
    raise KeyError(name)


def glorot_uniform(rng, shape, dtype=np.float64):
    """tf.get_variable's default initializer (DNN.py:85-88 passes none): U(+-sqrt(6/(fi+fo)));
    a rank-1 shape [n] has fan_in = fan_out = n."""
    if len(shape) == 1:
        fi = fo = shape[0]
    else:
        fi, fo = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_dnn_params(rng, feature_size, hidden_layer_sizes, output_size=1, dtype=np.float64):
    """Variables of DNN.build (DNN.py:67-94): per layer j a LayerNormalization (gamma=1,
    beta=0) on the layer INPUT, dnn_W_j [in,out], dnn_b_j [out] (Glorot, biases too)."""
    sizes = list(hidden_layer_sizes) + [output_size]
    params = []
    cur = feature_size
    for out in sizes:
        params.append({
            'gamma': np.ones(cur, dtype=dtype), 'beta': np.zeros(cur, dtype=dtype),
            'W': glorot_uniform(rng, (cur, out), dtype), 'b': glorot_uniform(rng, (out,), dtype)})
        cur = out
    return params


def dnn_forward(params, x, activation='elu', norm='layer'):
    """DNN.py:65-96: per layer LN(input) -> matmul+bias -> activation (hidden layers only).

    x: [M, F] rows (position-major concat, DNN.py:65).  Returns scores [M, out] and a cache.
    """
    cache = []
    h = x
    n = len(params)
    for j, p in enumerate(params):
        ent = {'x': h}
        if norm == 'layer':
            mean = h.mean(axis=-1, keepdims=True)
            var = ((h - mean) ** 2).mean(axis=-1, keepdims=True)  # biased, keras LN
            rstd = 1.0 / np.sqrt(var + LN_EPS)
            xhat = (h - mean) * rstd
            hn = xhat * p['gamma'] + p['beta']
            ent.update(xhat=xhat, rstd=rstd)
        else:
            hn = h
        z = hn @ p['W'] + p['b']
        ent['hn'] = hn
        if j != n - 1:
            h = act_fwd(activation, z)
        else:
            h = z
        ent['out'] = h
        cache.append(ent)
    return h, cache


def dnn_backward(params, cache, d_out, activation='elu', norm='layer'):
    """Closed-form gradients of ``dnn_forward`` (what tf.gradients yields, dla.py:237-239)."""
    grads = [None] * len(params)
    n = len(params)
    dz = d_out
    for j in range(n - 1, -1, -1):
        p, ent = params[j], cache[j]
        if j != n - 1:
            dz = dz * act_grad_from_output(activation, ent['out'])
        g = {'W': ent['hn'].T @ dz, 'b': dz.sum(axis=0)}
        dhn = dz @ p['W'].T
        if norm == 'layer':
            g['gamma'] = (dhn * ent['xhat']).sum(axis=0)
            g['beta'] = dhn.sum(axis=0)
            gx = dhn * p['gamma']
            dx = ent['rstd'] * (gx - gx.mean(axis=-1, keepdims=True)
                                - ent['xhat'] * (gx * ent['xhat']).mean(axis=-1, keepdims=True))
        else:
            g['gamma'] = np.zeros_like(p['gamma'])
            g['beta'] = np.zeros_like(p['beta'])
            dx = dhn
        grads[j] = g
        dz = dx
    return grads


def gather_rows(table, idx):
    """base_algorithm.py:120-132: append a zero row, embedding_lookup; idx == len(table) -> zeros."""
    ext = np.concatenate([table, np.zeros((1, table.shape[1]), dtype=table.dtype)], axis=0)
    return ext[idx]


# --------------------------------------------------------------------------
# Losses                      ultra/learning_algorithm/{base_algorithm,dla,...}.py
# --------------------------------------------------------------------------

def _log_softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    s = e.sum(axis=-1, keepdims=True)
    return x - m - np.log(s), e / s


def softmax_loss(output, labels, pw):
    """base_algorithm.py:242-251.  Returns (loss, d loss / d output, sum_wl).

    weighted_labels = (labels+1e-7)*pw; label_dis = wl/sum_i wl; loss = sum_b CE(label_dis,
    output)*sum_i wl / sum_{b,i} wl.  TF1's softmax_cross_entropy_with_logits does not
    backprop into labels, and the weights do not depend on the parameters the loss is
    differentiated for (dla.py:237-239), so d/d output = (softmax*S_b - wl)/sum wl.
    """
    wl = (labels + 0.0000001) * pw
    Sb = wl.sum(axis=1, keepdims=True)
    logsm, sm = _log_softmax(output)
    ce = -(wl / Sb * logsm).sum(axis=1)
    total = wl.sum()
    loss = (ce * Sb[:, 0]).sum() / total
    d_out = (sm * Sb - wl) / total
    return loss, d_out, total


def sigmoid_ce(labels, logits):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
    return np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))


def sigmoid_loss_on_list(output, labels, pw):
    """base_algorithm.py:178-186: mean_b sum_i sigCE(min(labels,1), output)*pw."""
    z = np.minimum(labels, 1)
    loss = (sigmoid_ce(z, output) * pw).sum(axis=1).mean()
    sig = 1.0 / (1.0 + np.exp(-output))
    d_out = (sig - z) * pw / output.shape[0]
    return loss, d_out


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def normalized_weights(prob, max_weight=-1):
    """dla.py:343-364 (get_normalized_weights): w_i = p_0/p_i, clipped to [0,max] iff max>0."""
    w = prob[:, :1] / prob
    if max_weight > 0:
        w = np.clip(w, 0, max_weight)
    return w


def logits_to_prob(logits, kind='softmax'):
    """dla.py:99-101; sigmoid_prob dla.py:29-30."""
    if kind == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-(logits - logits.mean(axis=-1, keepdims=True))))
    return _log_softmax(logits)[1]


def dla_losses(train_output, clicks, prop_W, prop_b, loss_func='softmax_loss',
               logits_kind='softmax', max_propensity_weight=-1):
    """dla.py:119-180 + DenoisingNet :257-306.

    The propensity tower sees a one-hot of the position, so logit_i = ELU(W[i,0]+b) for every
    row.  Returns a dict with rank_loss, exam_loss, d rank_loss/d train_output,
    d exam_loss/d W, d exam_loss/d b, propensity_weights [B,L], relevance_weights [B,L].
    """
    B, L = train_output.shape
    pre = prop_W[:, 0] + prop_b[0]
    logit = elu(pre)
    propensity = np.tile(logit[None, :], (B, 1))
    pw = normalized_weights(logits_to_prob(propensity, logits_kind), max_propensity_weight)
    rw = normalized_weights(logits_to_prob(train_output, logits_kind), max_propensity_weight)
    if loss_func == 'sigmoid_loss':
        rank_loss, d_out = sigmoid_loss_on_list(train_output, clicks, pw)
        exam_loss, d_prop = sigmoid_loss_on_list(propensity, clicks, rw)
    else:
        rank_loss, d_out, _ = softmax_loss(train_output, clicks, pw)
        exam_loss, d_prop, _ = softmax_loss(propensity, clicks, rw)
    d_logit = d_prop.sum(axis=0)
    d_pre = d_logit * np.where(pre > 0, 1.0, logit + 1.0)
    return {'rank_loss': rank_loss, 'exam_loss': exam_loss, 'd_output': d_out,
            'd_prop_W': d_pre[:, None], 'd_prop_b': np.array([d_pre.sum()]),
            'propensity_weights': pw, 'relevance_weights': rw}


def rem_step(train_output_plus_b, clicks, propensity, em_step_size, uniforms):
    """regression_EM.py:134-165 with the Bernoulli uniforms supplied (TF's own RNG stream
    cannot be reproduced, q15).  ``train_output_plus_b`` already includes sigmoid_prob_b.

    Returns loss, d loss/d logits, new propensity [1,L], ranker_labels.  The loss and the
    gradient use the PRE-update propensity (q16).
    """
    gamma = 1.0 / (1.0 + np.exp(-train_output_plus_b))
    p = propensity
    denom = 1 - p * gamma
    p_e1_r0_c0 = p * (1 - gamma) / denom
    p_e0_r1_c0 = (1 - p) * gamma / denom
    p_r1 = clicks + (1 - clicks) * p_e0_r1_c0
    new_p = (1 - em_step_size) * p + em_step_size * np.mean(
        clicks + (1 - clicks) * p_e1_r0_c0, axis=0, keepdims=True)
    ranker_labels = np.ceil(p_r1 - uniforms)
    ce = sigmoid_ce(ranker_labels, train_output_plus_b)
    loss = ce.sum(axis=1).mean()
    d_out = (gamma - ranker_labels) / train_output_plus_b.shape[0]
    return loss, d_out, new_p, ranker_labels


def pd_step(train_output, clicks, t_plus, t_minus, em_step_size, regulation_p):
    """pairwise_debias.py:139-171; pair CE = softplus(o_j - o_i) (base_algorithm.py:155-160).

    Returns loss ([1,1] like the reference), d loss/d output, new t_plus, new t_minus.
    """
    B, L = train_output.shape
    pair = np.zeros((L, L), dtype=train_output.dtype)
    d_out = np.zeros_like(train_output)
    for i in range(L):
        for j in range(L):
            if i == j:
                continue
            mask = np.minimum(1.0, np.maximum(clicks[:, i] - clicks[:, j], 0.0))
            diff = train_output[:, j] - train_output[:, i]
            sp = np.maximum(diff, 0) + np.log1p(np.exp(-np.abs(diff)))
            pair[i, j] = (mask * sp).sum()
            w = mask / (t_plus[0, i] * t_minus[0, j])
            sg = 1.0 / (1.0 + np.exp(-diff))
            d_out[:, j] += w * sg
            d_out[:, i] -= w * sg
    loss = 0.0
    tpl = np.zeros(L, dtype=train_output.dtype)
    tml = np.zeros(L, dtype=train_output.dtype)
    for i in range(L):
        for j in range(L):
            if i == j:
                continue
            tpl[i] += pair[i, j] / t_minus[0, j]
            tml[j] += pair[i, j] / t_plus[0, i]
            loss += pair[i, j] / t_plus[0, i] / t_minus[0, j]
    with np.errstate(divide='ignore', invalid='ignore'):
        expo = 1.0 / (regulation_p + 1)
        new_tp = (1 - em_step_size) * t_plus + em_step_size * np.power(tpl / tpl[0], expo)[None, :]
        new_tm = (1 - em_step_size) * t_minus + em_step_size * np.power(tml / tml[0], expo)[None, :]
    return np.array([[loss]]), d_out, new_tp, new_tm, pair


# --------------------------------------------------------------------------
# Optimiser                                                dla.py:240-255 etc.
# --------------------------------------------------------------------------

def global_norm(tensors):
    """tf.global_norm: sqrt(sum_t sum t^2)."""
    return math.sqrt(sum(float((t.astype(np.float64) ** 2).sum()) for t in tensors))


def clip_by_global_norm(tensors, clip_norm):
    """tf.clip_by_global_norm: t * clip/max(norm, clip); returns (clipped, norm)."""
    nrm = global_norm(tensors)
    scale = clip_norm / max(nrm, clip_norm)
    return [t * scale for t in tensors], nrm


ADAGRAD_INIT_ACC = 0.1  # tf.train.AdagradOptimizer initial_accumulator_value default


def adagrad_update(theta, acc, grad, lr):
    """tf.train.AdagradOptimizer: acc += g^2; theta -= lr*g/sqrt(acc)  (no epsilon)."""
    acc = acc + grad * grad
    return theta - lr * grad / np.sqrt(acc), acc


def sgd_update(theta, grad, lr):
    return theta - lr * grad


# --------------------------------------------------------------------------
# Metrics                          ultra/utils/metrics.py, metric_utils.py
# --------------------------------------------------------------------------

PADDING_SCORE = -100000  # base_algorithm.py:25


def remove_padding(scores, docids, n_rows):
    """base_algorithm.py:64-87: score <- PADDING_SCORE where docid == n_rows (the pad row)."""
    return np.where(docids == n_rows, np.array(PADDING_SCORE, dtype=scores.dtype), scores)


def sort_permutation(scores, uniforms=None, topn=None):
    """metric_utils.py:49-86 (sort_by_scores) as an index permutation [B, topn].

    shuffle = stable argsort(uniforms) (:77-80); top_k on the shuffled scores, equal values ->
    lower (shuffled) index first (:82); map back through the shuffle (:84-85).
    """
    B, L = scores.shape
    if topn is None:
        topn = L
    topn = min(topn, L)
    perm = np.zeros((B, topn), dtype=np.int32)
    for b in range(B):
        if uniforms is not None:
            shuffle = np.argsort(uniforms[b], kind='stable')
        else:
            shuffle = np.arange(L)
        s = scores[b][shuffle]
        order = sorted(range(L), key=lambda k: (-s[k], k))[:topn]
        perm[b] = shuffle[order]
    return perm


def _prepare(labels, predictions, weights):
    """metrics.py:231-251 (_prepare_and_validate_params)."""
    w = 1.0 if weights is None else weights
    ew = np.ones_like(labels) * w
    valid = labels >= 0
    lab = np.where(valid, labels, 0)
    pred = np.where(valid, predictions, predictions.min(axis=1, keepdims=True) - 1e-6)
    return lab, pred, ew


def _safe_div(a, b):
    """metrics.py:158-173."""
    return np.where(b == 0, 0.0, a / np.where(b == 0, 1.0, b))


def _dcg(sorted_labels, sorted_weights):
    """metrics.py:207-212: gain 2^l-1, discount 1/ln(1+pos)."""
    n = sorted_labels.shape[1]
    disc = np.log(np.arange(1, n + 1, dtype=sorted_labels.dtype) + 1)
    return (sorted_weights * (np.power(2.0, sorted_labels) - 1.0) / disc).sum(axis=1, keepdims=True)


def ndcg(labels, predictions, weights=None, topn=None, u_pred=None, u_ideal=None):
    """metrics.py:445-481."""
    lab, pred, ew = _prepare(labels, predictions, weights)
    p = sort_permutation(pred, u_pred, topn)
    rows = np.arange(lab.shape[0])[:, None]
    dcg = _dcg(lab[rows, p], ew[rows, p])
    pi = sort_permutation(ew * lab, u_ideal, topn)
    idcg = _dcg(lab[rows, pi], ew[rows, pi])
    per_list = _safe_div(dcg, idcg)
    rel = np.power(2.0, lab) - 1.0
    plw = _safe_div((ew * rel).sum(axis=1, keepdims=True), rel.sum(axis=1, keepdims=True))
    return float((per_list * plw).mean())


def mrr(labels, predictions, weights=None, u_pred=None):
    """metrics.py:254-284 (topn is ignored by the reference: always the full list, q8)."""
    lab, pred, ew = _prepare(labels, predictions, weights)
    p = sort_permutation(pred, u_pred, None)
    rows = np.arange(lab.shape[0])[:, None]
    rel = (lab[rows, p] >= 1.0).astype(lab.dtype)
    rr = 1.0 / np.arange(1, lab.shape[1] + 1, dtype=lab.dtype)
    m = (rel * rr).max(axis=1, keepdims=True)
    return float((m * np.ones_like(ew) * ew).mean())


def err(labels, predictions, max_label, weights=None, topn=None, u_pred=None):
    """metrics.py:287-323."""
    lab, pred, ew = _prepare(labels, predictions, weights)
    p = sort_permutation(pred, u_pred, topn)
    rows = np.arange(lab.shape[0])[:, None]
    sl, sw = lab[rows, p], ew[rows, p]
    n = sl.shape[1]
    rel = (np.power(2.0, sl) - 1) / np.power(2.0, max_label)
    with np.errstate(divide='ignore', invalid='ignore'):
        non_rel = np.cumprod(1.0 - rel, axis=1) / (1.0 - rel)
    rr = 1.0 / np.arange(1, n + 1, dtype=lab.dtype)
    tn = n if topn is None else topn
    rr = rr * (rr >= 1.0 / (tn + 1))
    e = (rel * non_rel * rr * sw).sum(axis=1, keepdims=True)
    return float(e.mean())
