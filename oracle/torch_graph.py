"""Op-for-op PyTorch-CPU restatement of the reference's TF1 training graphs -- TEST
INFRASTRUCTURE ONLY (second, independent derivation next to ``ultr_oracle.py``; also the
``cpu_baseline`` of ``bench.py``, kind "port").  Never imported by ``ultra_amd/``.

Gradients come from autograd on the literal op sequence, so they check the closed-form
gradients of ``ultr_oracle.py`` and of the HIP kernels.  Citations are reference file:line.
"""
import math

import torch
import torch.nn.functional as F

PADDING_SCORE = -100000.0  # base_algorithm.py:25
LN_EPS = 1e-3


def _act(name):
    if name == 'elu':
        return F.elu
    if name == 'relu':
        return F.relu
    if name == 'selu':  # base_ranking_model.py:13-26
        alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
        return lambda x: scale * torch.where(x >= 0.0, x, alpha * F.elu(x))
    if name == 'tanh':
        return torch.tanh
    if name == 'sigmoid':
        return torch.sigmoid
    raise KeyError(name)


def params_from_numpy(np_params, dtype=torch.float64):
    out = []
    for p in np_params:
        out.append({k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in p.items()})
    return out


def dnn_build(params, input_list, activation='elu', norm='layer'):
    """DNN.py:63-96: concat positions on axis 0, per layer LN -> matmul -> bias -> act, split."""
    x = torch.cat(input_list, dim=0)
    act = _act(activation)
    n = len(params)
    for j, p in enumerate(params):
        if norm == 'layer':
            x = F.layer_norm(x, (x.shape[-1],), p['gamma'], p['beta'], LN_EPS)
        elif norm == 'batch':
            # keras BatchNormalization(training=True) on [M, K] (DNN.py:82-84): tf.nn.moments over the rows (biased
            # variance), epsilon 1e-3, then tf.nn.batch_normalization
            mean = x.mean(dim=0, keepdim=True)
            var = ((x - mean) ** 2).mean(dim=0, keepdim=True)
            x = (x - mean) * torch.rsqrt(var + LN_EPS) * p['gamma'] + p['beta']
        elif norm == 'batch_eval':
            # training=False: the moving statistics.  The reference's training step never runs the layer's update
            # ops (dla.py:237-255 fetches the optimiser's only), so they keep their initial values: mean 0, variance 1
            x = x * (1.0 / math.sqrt(1.0 + LN_EPS)) * p['gamma'] + p['beta']
        x = x @ p['W'] + p['b']
        if j != n - 1:
            x = act(x)
    return torch.split(x, input_list[0].shape[0], dim=0)


def get_ranking_scores(params, letor_features, docid_list, activation='elu', norm='layer'):
    """base_algorithm.py:104-134 + ranking_model :89-102 -> [B, len(docid_list)]."""
    table = torch.cat([letor_features, torch.zeros(1, letor_features.shape[1],
                                                   dtype=letor_features.dtype)], dim=0)
    inputs = [table[ids] for ids in docid_list]
    return torch.cat(dnn_build(params, inputs, activation, norm), dim=1)


def remove_padding(docid_list, output, n_rows):
    """base_algorithm.py:64-87."""
    cols = []
    for i in range(output.shape[1]):
        cols.append(torch.where(docid_list[i] != n_rows, output[:, i],
                                torch.full_like(output[:, i], PADDING_SCORE)))
    return torch.stack(cols, dim=1)


def softmax_loss(output, labels, pw):
    """base_algorithm.py:242-251 (labels of the CE are stop-gradiented as in TF1)."""
    wl = (labels + 0.0000001) * pw
    label_dis = (wl / wl.sum(1, keepdim=True)).detach()
    ce = -(label_dis * F.log_softmax(output, dim=1)).sum(1)
    return (ce * wl.sum(1)).sum() / wl.sum()


def sigmoid_loss_on_list(output, labels, pw):
    """base_algorithm.py:178-186."""
    z = torch.minimum(labels, torch.ones_like(labels))
    loss = F.binary_cross_entropy_with_logits(output, z, reduction='none') * pw
    return loss.sum(1).mean()


def get_normalized_weights(prob, max_w=-1):
    """dla.py:343-364."""
    cols = [prob[:, 0] / prob[:, i] for i in range(prob.shape[1])]
    w = torch.stack(cols, dim=1)
    if max_w > 0:
        w = torch.clamp(w, 0, max_w)
    return w


def sigmoid_prob(logits):
    """dla.py:29-30."""
    return torch.sigmoid(logits - logits.mean(-1, keepdim=True))


def denoising_net(prop_W, prop_b, labels_list):
    """dla.py:257-306: one-hot(position) -> W_0 [L,1] + b_0 -> ELU, per position."""
    L = len(labels_list)
    outs = []
    for i in range(L):
        feats = [torch.zeros_like(labels_list[i]).unsqueeze(-1) for _ in range(L)]
        feats[i] = torch.ones_like(labels_list[i]).unsqueeze(-1)
        outs.append(F.elu(torch.cat(feats, 1) @ prop_W + prop_b))
    return torch.cat(outs, 1)


def clip_by_global_norm(grads, clip):
    """tf.clip_by_global_norm."""
    nrm = torch.sqrt(sum((g * g).sum() for g in grads))
    scale = clip / torch.maximum(nrm, torch.tensor(clip, dtype=nrm.dtype))
    return [g * scale for g in grads], nrm


# ---- metrics (metrics.py / metric_utils.py), literal -------------------------------------

def sort_by_scores(scores, features_list, topn=None, uniforms=None):
    """metric_utils.py:49-86."""
    L = scores.shape[1]
    topn = L if topn is None else min(topn, L)
    if uniforms is None:
        uniforms = torch.rand(scores.shape, dtype=torch.float32)
    shuffle = torch.argsort(uniforms, dim=1, stable=True)
    s = torch.gather(scores, 1, shuffle)
    idx = torch.sort(-s, dim=1, stable=True).indices[:, :topn]  # top_k: ties -> lower index
    idx = torch.gather(shuffle, 1, idx)
    return [torch.gather(f, 1, idx) for f in features_list]


def _prepare(labels, predictions, weights):
    w = 1.0 if weights is None else weights
    ew = torch.ones_like(labels) * w
    valid = labels >= 0
    labels = torch.where(valid, labels, torch.zeros_like(labels))
    predictions = torch.where(valid, predictions,
                              -1e-6 * torch.ones_like(predictions) + predictions.min(1, keepdim=True).values)
    return labels, predictions, ew


def _safe_div(a, b):
    return torch.where(b == 0, torch.zeros_like(a), a / torch.where(b == 0, torch.ones_like(b), b))


def _dcg(labels, weights):
    n = labels.shape[1]
    den = torch.log(torch.arange(1, n + 1, dtype=labels.dtype) + 1)
    return (weights * (torch.pow(2.0, labels) - 1.0) / den).sum(1, keepdim=True)


def ndcg(labels, predictions, weights=None, topn=None, u_pred=None, u_ideal=None):
    """metrics.py:445-481."""
    labels, predictions, w = _prepare(labels, predictions, weights)
    sl, sw = sort_by_scores(predictions, [labels, w], topn, u_pred)
    dcg = _dcg(sl, sw)
    il, iw = sort_by_scores(w * labels, [labels, w], topn, u_ideal)
    per_list = _safe_div(dcg, _dcg(il, iw))
    rel = torch.pow(2.0, labels) - 1.0
    plw = _safe_div((w * rel).sum(1, keepdim=True), rel.sum(1, keepdim=True))
    return (per_list * plw).mean()


def mrr(labels, predictions, weights=None, u_pred=None):
    """metrics.py:254-284."""
    labels, predictions, w = _prepare(labels, predictions, weights)
    sl, = sort_by_scores(predictions, [labels], None, u_pred)
    rel = (sl >= 1.0).to(labels.dtype)
    rr = 1.0 / torch.arange(1, labels.shape[1] + 1, dtype=labels.dtype)
    m = (rel * rr).max(1, keepdim=True).values
    return (m * torch.ones_like(w) * w).mean()


def err(labels, predictions, max_label, weights=None, topn=None, u_pred=None):
    """metrics.py:287-323."""
    labels, predictions, w = _prepare(labels, predictions, weights)
    sl, sw = sort_by_scores(predictions, [labels, w], topn, u_pred)
    n = sl.shape[1]
    rel = (torch.pow(2.0, sl) - 1) / math.pow(2.0, max_label)
    non_rel = torch.cumprod(1.0 - rel, dim=1) / (1.0 - rel)
    rr = 1.0 / torch.arange(1, n + 1, dtype=labels.dtype)
    tn = n if topn is None else topn
    rr = rr * (rr >= 1.0 / (tn + 1)).to(labels.dtype)
    return (rel * non_rel * rr * sw).sum(1, keepdim=True).mean()


def metric_value(name, topn, labels, predictions, weights, max_label, u_pred=None):
    if name == 'ndcg':
        return ndcg(labels, predictions, weights, topn, u_pred)
    if name == 'mrr':
        return mrr(labels, predictions, weights, u_pred)
    if name == 'err':
        return err(labels, predictions, max_label, weights, topn, u_pred)
    raise KeyError(name)


# ---- one training step per algorithm ----------------------------------------------------

class TorchReferenceStep(object):
    """Holds variables + optimiser slots and runs one reference-equivalent training step.

    feed: letor_features [n,F] tensor, docids: list of L int64 [B], labels: list of L float [B].
    """

    def __init__(self, algo, np_params, rank_list_size, hparams=None, dtype=torch.float64,
                 activation='elu', norm='layer', metrics=('mrr', 'ndcg'), topn=(3, 5, 10),
                 max_label=2.0, prop_W=None, prop_b=None, with_metrics=True):
        self.algo = algo
        self.dtype = dtype
        self.L = rank_list_size
        self.act, self.norm = activation, norm
        self.params = params_from_numpy(np_params, dtype)
        self.hp = dict(learning_rate=0.05, max_gradient_norm=5.0, loss_func='softmax_loss',
                       logits_to_prob='softmax', propensity_learning_rate=-1.0,
                       ranker_loss_weight=1.0, l2_loss=0.0, max_propensity_weight=-1,
                       grad_strategy='ada', EM_step_size=0.05, regulation_p=1)
        if algo == 'pd':
            self.hp['learning_rate'] = 0.005  # pairwise_debias.py:64
        if hparams:
            self.hp.update(hparams)
        self.metrics, self.topn, self.max_label = metrics, topn, max_label
        self.with_metrics = with_metrics
        self.extra = []
        if algo == 'dla':
            self.prop_W = torch.tensor(prop_W, dtype=dtype, requires_grad=True)
            self.prop_b = torch.tensor(prop_b, dtype=dtype, requires_grad=True)
        if algo == 'rem':
            self.sigmoid_prob_b = torch.zeros(1, dtype=dtype, requires_grad=True)  # regression_EM.py:116
            self.propensity = torch.ones(1, self.L, dtype=dtype) * 0.9              # :121-122
            self.extra = [self.sigmoid_prob_b]
        if algo == 'pd':
            self.t_plus = torch.ones(1, self.L, dtype=dtype)    # pairwise_debias.py:116-119
            self.t_minus = torch.ones(1, self.L, dtype=dtype)
        self.acc = {}
        self.global_step = 0

    # -- helpers
    def ranker_vars(self):
        out = []
        for p in self.params:
            out += [p['W'], p['b']]
            if self.norm in ('layer', 'batch'):
                out += [p['gamma'], p['beta']]
        return out

    def _apply(self, variables, grads, lr):
        with torch.no_grad():
            for v, g in zip(variables, grads):
                if self.hp['grad_strategy'] == 'sgd':
                    v -= lr * g
                else:
                    a = self.acc.setdefault(id(v), torch.full_like(v, 0.1))
                    a += g * g
                    v -= lr * g / torch.sqrt(a)

    def _train_metrics(self, labels_bl, train_output, docids, n_rows, list_weights, u_pred=None):
        out = {}
        if not self.with_metrics:
            return out
        with torch.no_grad():
            pad_removed = remove_padding(docids, train_output.detach(), n_rows)
            for m in self.metrics:
                for k in self.topn:
                    out['%s_%d' % (m, k)] = float(metric_value(m, k, labels_bl, pad_removed, None,
                                                               self.max_label, u_pred))
                    if list_weights is not None:
                        out['Weighted_%s_%d' % (m, k)] = float(
                            metric_value(m, k, labels_bl, pad_removed, list_weights, self.max_label, u_pred))
        return out

    # -- steps
    def step(self, letor_features, docids, labels, pw_lists=None, uniforms=None, u_pred=None):
        hp = self.hp
        L = self.L
        n_rows = letor_features.shape[0]
        train_output = get_ranking_scores(self.params, letor_features, docids[:L], self.act, self.norm)
        labels_bl = torch.stack(labels[:L], dim=1)
        rv = self.ranker_vars()
        summary = {}
        if self.algo == 'dla':
            propensity = denoising_net(self.prop_W, self.prop_b, labels[:L])
            to_prob = sigmoid_prob if hp['logits_to_prob'] == 'sigmoid' else (lambda x: F.softmax(x, dim=1))
            loss_fn = sigmoid_loss_on_list if hp['loss_func'] == 'sigmoid_loss' else softmax_loss
            pw = get_normalized_weights(to_prob(propensity), hp['max_propensity_weight'])
            rank_loss = loss_fn(train_output, labels_bl, pw)
            rw = get_normalized_weights(to_prob(train_output), hp['max_propensity_weight'])
            exam_loss = loss_fn(propensity, labels_bl, rw)
            if hp['l2_loss'] > 0:  # dla.py:230-234
                for p in rv:
                    rank_loss = rank_loss + hp['l2_loss'] * (p * p).sum() / 2
            loss = exam_loss + hp['ranker_loss_weight'] * rank_loss
            dvars = [self.prop_W, self.prop_b]
            dg = torch.autograd.grad(exam_loss, dvars, retain_graph=True)
            rg = torch.autograd.grad(rank_loss, rv)
            if hp['max_gradient_norm'] > 0:
                dg, _ = clip_by_global_norm(list(dg), hp['max_gradient_norm'])
                rg, _ = clip_by_global_norm(list(rg), hp['max_gradient_norm'] * hp['ranker_loss_weight'])
            norm = torch.sqrt(sum((g * g).sum() for g in list(dg) + list(rg)))
            plr = hp['learning_rate'] if hp['propensity_learning_rate'] < 0 else hp['propensity_learning_rate']
            self.grads = {'ranker': [g.clone() for g in rg], 'denoise': [g.clone() for g in dg]}
            self._apply(dvars, dg, plr)
            self._apply(rv, rg, hp['learning_rate'])
            summary.update({'Rank Loss': float(rank_loss.detach()), 'Exam Loss': float(exam_loss.detach()),
                            'Gradient Norm': float(norm), 'Final Loss': float(loss.detach())})
            for i in range(L):
                summary['Inverse Propensity weights %d' % i] = float(pw[:, i].detach().mean())
                summary['Relevance weights %d' % i] = float(rw[:, i].detach().mean())
            lw = (pw.detach() * labels_bl.clamp(0, 1)).mean(1, keepdim=True)
            summary.update(self._train_metrics(labels_bl, train_output, docids, n_rows, lw, u_pred))
        elif self.algo == 'ipw':
            pwt = torch.stack(pw_lists[:L], dim=1)
            if hp['loss_func'] == 'sigmoid_loss':
                loss = sigmoid_loss_on_list(train_output, labels_bl, pwt)
            else:
                loss = softmax_loss(train_output, labels_bl, pwt)
            if hp['l2_loss'] > 0:
                for p in rv:
                    loss = loss + hp['l2_loss'] * (p * p).sum() / 2
            g = torch.autograd.grad(loss, rv)
            if hp['max_gradient_norm'] > 0:
                g, norm = clip_by_global_norm(list(g), hp['max_gradient_norm'])
            self.grads = {'ranker': [x.clone() for x in g]}
            self._apply(rv, g, hp['learning_rate'])
            summary['Loss'] = float(loss.detach().sum())
            lw = (pwt * labels_bl.clamp(0, 1)).mean(1, keepdim=True)
            summary.update(self._train_metrics(labels_bl, train_output, docids, n_rows, lw, u_pred))
        elif self.algo == 'naive':  # navie_algorithm.py:85-170: the same losses with propensity_weights = None (ones)
            ones = torch.ones_like(labels_bl)
            if hp['loss_func'] == 'sigmoid_cross_entropy':
                loss = sigmoid_loss_on_list(train_output, labels_bl, ones)
            else:
                loss = softmax_loss(train_output, labels_bl, ones)
            if hp['l2_loss'] > 0:
                l2 = sum((p * p).sum() / 2 for p in rv)
                summary['L2 Loss'] = float(l2.detach())
                loss = loss + hp['l2_loss'] * l2
            g = torch.autograd.grad(loss, rv)
            if hp['max_gradient_norm'] > 0:
                g, norm = clip_by_global_norm(list(g), hp['max_gradient_norm'])
                summary['Gradient Norm'] = float(norm)
            self.grads = {'ranker': [x.clone() for x in g]}
            self._apply(rv, g, hp['learning_rate'])
            pad_removed = remove_padding(docids, train_output.detach(), n_rows)
            summary.update({'Max_output_score': float(train_output.detach().max()),
                            'Min_output_score': float(train_output.detach().min()),
                            'Max_output_score_without_pad': float(pad_removed.max()),
                            'Min_output_score_without_pad': float(pad_removed.min()),
                            'Loss': float(loss.detach().sum())})
            summary.update(self._train_metrics(labels_bl, train_output, docids, n_rows, None, u_pred))
        elif self.algo == 'rem':
            train_output = train_output + self.sigmoid_prob_b
            gamma = torch.sigmoid(train_output)
            p = self.propensity
            p_e1_r0_c0 = p * (1 - gamma) / (1 - p * gamma)
            p_e0_r1_c0 = (1 - p) * gamma / (1 - p * gamma)
            p_r1 = labels_bl + (1 - labels_bl) * p_e0_r1_c0
            new_p = ((1 - hp['EM_step_size']) * p + hp['EM_step_size'] * torch.mean(
                labels_bl + (1 - labels_bl) * p_e1_r0_c0, dim=0, keepdim=True)).detach()
            if uniforms is None:
                uniforms = torch.rand(p_r1.shape, dtype=self.dtype)
            ranker_labels = torch.ceil(p_r1 - uniforms).detach()  # regression_EM.py:39 (Ceil has no gradient)
            loss = F.binary_cross_entropy_with_logits(train_output, ranker_labels, reduction='none').sum(1).mean()
            variables = rv + [self.sigmoid_prob_b]
            if hp['l2_loss'] > 0:
                for q in variables:
                    loss = loss + hp['l2_loss'] * (q * q).sum() / 2
            g = torch.autograd.grad(loss, variables)
            if hp['max_gradient_norm'] > 0:
                g, norm = clip_by_global_norm(list(g), hp['max_gradient_norm'])
                summary['Gradient Norm'] = float(norm)
            self.grads = {'ranker': [x.clone() for x in g]}
            self._apply(variables, g, hp['learning_rate'])
            self.ranker_labels = ranker_labels
            lw = ((1.0 / p) * labels_bl.clamp(0, 1)).mean(1, keepdim=True)
            self.propensity = new_p
            summary['Loss'] = float(loss.detach().sum())
            summary.update(self._train_metrics(labels_bl, train_output, docids, n_rows, lw, u_pred))
        elif self.algo == 'pd':
            out_list = torch.split(train_output, 1, dim=1)
            tpl = [0.0] * L
            tml = [0.0] * L
            loss = 0.0
            for i in range(L):  # pairwise_debias.py:143-159
                for j in range(L):
                    if i == j:
                        continue
                    mask = torch.minimum(torch.ones_like(labels[i]), F.relu(labels[i] - labels[j]))
                    logits = torch.cat([out_list[i], out_list[j]], dim=1)
                    ce = -F.log_softmax(logits, dim=1)[:, 0]  # base_algorithm.py:155-160
                    pair_loss = (mask * ce).sum()
                    tpl[i] = tpl[i] + pair_loss / self.t_minus[:, j:j + 1]
                    tml[j] = tml[j] + pair_loss / self.t_plus[:, i:i + 1]
                    loss = loss + pair_loss / self.t_plus[:, i:i + 1] / self.t_minus[:, j:j + 1]
            expo = 1.0 / (hp['regulation_p'] + 1)
            new_tp = ((1 - hp['EM_step_size']) * self.t_plus + hp['EM_step_size'] * torch.pow(
                torch.cat(tpl, dim=1) / tpl[0], expo)).detach()
            new_tm = ((1 - hp['EM_step_size']) * self.t_minus + hp['EM_step_size'] * torch.pow(
                torch.cat(tml, dim=1) / tml[0], expo)).detach()
            if hp['l2_loss'] > 0:
                for q in rv:
                    loss = loss + hp['l2_loss'] * (q * q).sum() / 2
            g = torch.autograd.grad(loss.sum(), rv)
            if hp['max_gradient_norm'] > 0:
                g, norm = clip_by_global_norm(list(g), hp['max_gradient_norm'])
                summary['Gradient Norm'] = float(norm)
            self.grads = {'ranker': [x.clone() for x in g]}
            self._apply(rv, g, hp['learning_rate'])
            self.t_plus, self.t_minus = new_tp, new_tm
            summary['Loss'] = float(loss.detach().sum())
            summary.update(self._train_metrics(labels_bl, train_output, docids, n_rows, None, u_pred))
        else:
            raise KeyError(self.algo)
        self.global_step += 1
        self.train_output = train_output.detach()
        return float(loss.detach().sum()), summary
