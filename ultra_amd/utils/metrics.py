"""Ranking metrics on the device (reference: ultra/utils/metrics.py, metric_utils.py).

``make_ranking_metric_fn(metric, topn)`` keeps the reference's factory signature
(metrics.py:64-155) and returns ``fn(labels, predictions, weights)`` over [batch, list] device
tensors; the work is done by the LDS bitonic-sort kernel behind ``ultr_metrics``.  The learning
algorithms call ``batch_metrics`` directly with position-major tensors, which evaluates every
(metric, topn) pair -- plain and weighted -- in ONE launch.
"""
import ctypes

import torch

from .. import _lib


class RankingMetricKey(object):
    """Metric key strings (metrics.py:36-61).  MAX_LABEL is set by the data loader (data_utils.py:92)."""
    MRR = 'mrr'
    ERR = 'err'
    MAX_LABEL = None
    ARP = 'arp'
    NDCG = 'ndcg'
    DCG = 'dcg'
    PRECISION = 'precision'
    MAP = 'map'
    ORDERED_PAIR_ACCURACY = 'ordered_pair_accuracy'


DEVICE_METRICS = ('mrr', 'ndcg', 'err')


class MetricPlan(object):
    """Which (metric, topn) pairs one ultr_metrics launch evaluates, and where each lands in `out`."""

    def __init__(self, metric_names, topn_list):
        for m in metric_names:
            if m not in DEVICE_METRICS:
                raise NotImplementedError(
                    "metric '%s' is not on the MI355X hot path (mrr, ndcg, err are)" % m)
        self.names = list(metric_names)
        self.topn = [int(t) for t in topn_list]
        self.mask = 0
        for m in self.names:
            self.mask |= _lib.METRIC_BITS[m]
        self.kernel_order = [m for m in _lib.METRIC_ORDER if m in self.names]
        self.n_out = len(self.kernel_order) * len(self.topn) * 2
        self.topn_c = (ctypes.c_int * len(self.topn))(*self.topn)

    def slot(self, metric, topn_index, weighted):
        return (self.kernel_order.index(metric) * len(self.topn) + topn_index) * 2 + (1 if weighted else 0)

    def tags(self, weighted):
        """[(tag, slot)] in the reference's summary order: for metric: for topn (dla.py:109-115,205-218)."""
        out = []
        for m in self.names:
            for i, k in enumerate(self.topn):
                out.append(('%s_%d' % (m, k), self.slot(m, i, False)))
                if weighted:
                    out.append(('Weighted_%s_%d' % (m, k), self.slot(m, i, True)))
        return out


def metrics_workspace(B, plan, device):
    n = _lib.lib().ultr_metrics_workspace_floats(int(B), len(plan.topn))
    return torch.zeros(n, dtype=torch.float32, device=device)


def batch_metrics(plan, scores_lb, labels_lb, docids_lb, n_rows, list_weights, out, ws, u_pred=None, max_label=None,
                  deferred=False, ties=None):
    """scores/labels/docids position-major [L][B] device tensors -> out[plan.n_out] (device, in place).

    deferred=True parks the request: the next ``ultr_dnn_backward`` under the same request context carries it inside one of its own
    launches (``ultr_metrics_deferred``); every tensor passed here must stay alive until then.
    ties=(seed, step, counter tensor | None, slot_base) with u_pred None: the kernel draws the tie-break uniforms itself
    (``ultr_metrics_ties``), fresh ones on every call like sort_by_scores (metric_utils.py:76-81)."""
    L, B = scores_lb.shape
    ml = max_label if max_label is not None else (RankingMetricKey.MAX_LABEL or 0.0)
    if u_pred is None and ties is not None:
        seed, step, ctr, slot_base = ties
        fn = _lib.lib().ultr_metrics_ties_deferred if deferred else _lib.lib().ultr_metrics_ties
        _lib.check(fn(
            _lib.ptr(scores_lb), _lib.ptr(labels_lb), _lib.ptr(docids_lb), int(n_rows), _lib.ptr(list_weights),
            int(B), int(L), plan.topn_c, len(plan.topn), plan.mask, float(ml), int(seed) & 0xFFFFFFFFFFFFFFFF,
            int(step) & 0xFFFFFFFF, _lib.ptr(ctr), int(slot_base) & 0xFFFFFFFF, _lib.ptr(out), _lib.ptr(ws),
            _lib.current_stream()), 'ultr_metrics_ties')
        return out
    fn = _lib.lib().ultr_metrics_deferred if deferred else _lib.lib().ultr_metrics
    _lib.check(fn(
        _lib.ptr(scores_lb), _lib.ptr(labels_lb), _lib.ptr(docids_lb), int(n_rows), _lib.ptr(list_weights),
        _lib.ptr(u_pred), int(B), int(L), plan.topn_c, len(plan.topn), plan.mask, float(ml), _lib.ptr(out),
        _lib.ptr(ws), _lib.current_stream()), 'ultr_metrics')
    return out


def make_ranking_metric_fn(metric_key, topn=None, name=None):
    """fn(labels [B,L], predictions [B,L], weights None | [B,1]) -> 0-dim device tensor."""
    def fn(labels, predictions, weights=None):
        B, L = predictions.shape
        plan = MetricPlan([metric_key], [topn if topn is not None else L])
        dev = predictions.device
        out = torch.zeros(plan.n_out, dtype=torch.float32, device=dev)
        ws = metrics_workspace(B, plan, dev)
        lw = None
        if weights is not None:
            lw = weights.reshape(-1).to(torch.float32).contiguous()
            if lw.numel() != B:
                raise NotImplementedError('only per-list weights [B,1] are on the device path')
        batch_metrics(plan, predictions.t().contiguous().float(), labels.t().contiguous().float(), None, 0, lw,
                      out, ws)
        return out[plan.slot(metric_key, 0, weights is not None)]
    return fn


def sort_by_scores_indices(scores_bl, uniforms_bl=None, topn=None):
    """The permutation of metric_utils.sort_by_scores (:49-86): [B, topn] int32 indices."""
    B, L = scores_bl.shape
    topn = L if topn is None else min(int(topn), L)
    perm = torch.empty((B, topn), dtype=torch.int32, device=scores_bl.device)
    u = None if uniforms_bl is None else uniforms_bl.contiguous().float()
    scores_lb = scores_bl.t().contiguous().float()
    _lib.check(_lib.lib().ultr_sort_by_scores(_lib.ptr(scores_lb), _lib.ptr(u), int(B),
                                              int(L), int(topn), _lib.ptr(perm), _lib.current_stream()),
               'ultr_sort_by_scores')
    return perm
